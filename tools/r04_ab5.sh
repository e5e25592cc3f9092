#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r4j}; mkdir -p $O; export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_depth_order_gpu.py tests/test_edge_cases_gpu.py tests/test_end_to_end_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for WL in c3 c5 c3d; do
  GSWT_ORDER=depth rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_d$WL -- python3 tools/serial_frames.py $WL 20 > $O/serial_depth_$WL.log 2>&1
  python3 tools/pmc_summary.py stats $(find $O/serial_d$WL -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_${WL}_depth.csv; rm -rf $O/serial_d$WL
  echo "== $WL depth"; grep "radix\|k_emit" $O/kernel_stats_serial_${WL}_depth.csv | cut -c1-100
done
timeout -k 10 400 python bench.py --order depth > $O/bench_c3_depth.json 2> $O/bench_c3_depth.err
timeout -k 10 300 python bench.py --order depth --graph --no-cpu-baseline > $O/bench_c3_depth_graph.json 2>> $O/bench_c3_depth.err
timeout -k 10 300 python bench.py --workload c3d --order depth --no-cpu-baseline > $O/bench_c3d_depth.json 2>> $O/bench_c3_depth.err
timeout -k 10 400 python bench.py --workload c5 --order depth --no-cpu-baseline > $O/bench_c5_depth.json 2>> $O/bench_c3_depth.err
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_c3.json 2>> $O/bench_c3_depth.err
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'],1), round((d.get('static_camera') or {}).get('value',0),1), d['roofline']['frac'])" $f; done
timeout -k 10 400 python tools/shard_emulation.py c3 100 > $O/shard_emulation_c3.txt 2>&1; tail -8 $O/shard_emulation_c3.txt | cut -c1-260
