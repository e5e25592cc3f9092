#!/bin/bash
# round 4: the tile-local depth sort (default) against the global depth passes: GPU suite, serial --stats lines, fly-path rates
set -o pipefail
O=gpurun_out/${1:-r4w}; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -1 $O/tests.log
for DS in 0 1; do for WL in c3 c3d; do
  GSWT_ORDER=depth GSWT_DEPTH_SORT=$DS rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$WL$DS -- python3 tools/serial_frames.py $WL 20 > $O/s_$WL$DS.log 2>&1
  python3 tools/pmc_summary.py stats $(find $O/s_$WL$DS -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_${WL}_depth_sort$DS.csv; rm -rf $O/s_$WL$DS
  echo "== $WL depth sort $DS"; head -14 $O/kernel_stats_serial_${WL}_depth_sort$DS.csv | cut -c1-110
done; done
for A in "" "--order depth --depth-sort 1" "--order depth" "--order depth --graph" "--workload c3d --order depth --depth-sort 1" "--workload c3d --order depth" "--workload c5 --order depth" "--workload c3d"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $A > $O/bench.json 2>> $O/bench.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2:], round(d['value'],1), round((d.get('static_camera') or {}).get('value',0),1), d.get('depth_sort_frames_tile_local_global_longest_list'))" $O/bench.json $A | tee -a $O/bench_lines.txt
done
