#!/bin/bash
# round 4: the compositor's work items in tile order against heaviest first (GSWT_OPT_ITEM_ORDER)
set -o pipefail
O=gpurun_out/${1:-r4y}; mkdir -p $O; export TMPDIR=/tmp
GSWT_SWEEP_CASES=56 timeout -k 10 900 python -m pytest tests/test_composite_dw_gpu.py tests/test_random_sweep_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -1 $O/tests.log
for IO in 0 1; do for WL in c3 c3d c3h c5; do
  GSWT_ITEM_ORDER=$IO rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$WL$IO -- python3 tools/serial_frames.py $WL 20 > $O/s_$WL$IO.log 2>&1
  python3 tools/pmc_summary.py stats $(find $O/s_$WL$IO -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_${WL}_item_order$IO.csv; rm -rf $O/s_$WL$IO
  echo "== $WL item order $IO"; grep "k_composite\|k_items\|k_combine" $O/kernel_stats_serial_${WL}_item_order$IO.csv | cut -c1-110
done; done
for A in "--item-order 0" "--item-order 1" "--workload c3d --item-order 0" "--workload c3d --item-order 1" "--workload c5 --item-order 0" "--workload c5 --item-order 1"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $A > $O/bench.json 2>> $O/bench.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2:], round(d['value'],1), round((d.get('static_camera') or {}).get('value',0),1))" $O/bench.json $A | tee -a $O/bench_lines.txt
done
