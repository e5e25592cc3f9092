#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r4k}; mkdir -p $O; export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_depth_order_gpu.py tests/test_end_to_end_gpu.py tests/test_random_sweep_gpu.py -x -q -s > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log; grep "longest tile" $O/tests.log
for DS in 1 2; do
  export GSWT_ORDER=depth GSWT_DEPTH_SORT=$DS
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_ds$DS -- python3 tools/serial_frames.py c3 20 > $O/serial_ds$DS.log 2>&1
  python3 tools/pmc_summary.py stats $(find $O/serial_ds$DS -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_c3_depth_sort$DS.csv; rm -rf $O/serial_ds$DS
  echo "== c3 depth, GSWT_OPT_DEPTH_SORT=$DS"; head -16 $O/kernel_stats_serial_c3_depth_sort$DS.csv | cut -c1-100
done
unset GSWT_ORDER GSWT_DEPTH_SORT
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_c3.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --order depth --depth-sort 1 --no-cpu-baseline > $O/bench_c3_depth_global.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --order depth --no-cpu-baseline > $O/bench_c3_depth_auto.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --order depth --graph --no-cpu-baseline > $O/bench_c3_depth_auto_graph.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --workload c3d --order depth --no-cpu-baseline > $O/bench_c3d_depth_auto.json 2>> $O/bench.err
timeout -k 10 400 python bench.py --workload c5 --order depth --no-cpu-baseline > $O/bench_c5_depth_auto.json 2>> $O/bench.err
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'],1), round((d.get('static_camera') or {}).get('value',0),1), d['roofline']['frac'], d['config'].get('depth_sort_frames_tile_local_global_longest_list'))" $f; done
