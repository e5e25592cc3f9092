#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r4l}; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for WL in c3 c5; do
for NC in 0 1; do
  [ $NC = 1 ] && export GSWT_NO_CHUNK_CULL=1 || unset GSWT_NO_CHUNK_CULL
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$WL$NC -- python3 tools/serial_frames.py $WL 20 > $O/serial_$WL$NC.log 2>&1
  python3 tools/pmc_summary.py stats $(find $O/serial_$WL$NC -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_${WL}_nocull$NC.csv; rm -rf $O/serial_$WL$NC
  echo "== $WL no_chunk_cull=$NC"; grep "k_project\|k_cull\|k_live\|k_emit\|k_totals" $O/kernel_stats_serial_${WL}_nocull$NC.csv | cut -c1-100
done; done
unset GSWT_NO_CHUNK_CULL
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_c3_cull_$i.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --no-cpu-baseline --no-chunk-cull > $O/bench_c3_nocull_$i.json 2>> $O/bench.err
done
timeout -k 10 400 python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5_cull.json 2>> $O/bench.err
timeout -k 10 400 python bench.py --workload c5 --no-cpu-baseline --no-chunk-cull > $O/bench_c5_nocull.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --workload c3d --no-cpu-baseline > $O/bench_c3d_cull.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --order depth --no-cpu-baseline > $O/bench_c3_depth_cull.json 2>> $O/bench.err
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'],1), round((d.get('static_camera') or {}).get('value',0),1), d['roofline']['frac'])" $f; done
