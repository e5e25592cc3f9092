#!/bin/bash
# round 4, first GPU call: the new tests, the whole GPU suite, bench lines (reference / depth / strict) and serial per-kernel stats
set -o pipefail
O=gpurun_out/${1:-r4b}; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_depth_order_gpu.py tests/test_render_parity_gpu.py tests/test_strict_gpu.py -x -q -s > $O/new_tests.log 2>&1; echo "new tests rc=$?"; tail -3 $O/new_tests.log
grep -h "L-inf\|vs strict" $O/new_tests.log | head -20
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -2 $O/gpu_tests.log
for V in ref depth v2; do
  [ $V = depth ] && export GSWT_ORDER=depth || unset GSWT_ORDER; [ $V = v2 ] && export GSWT_VS=v2 || unset GSWT_VS
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$V -- python3 tools/serial_frames.py c3 20 > $O/serial_$V.log 2>&1
  python3 tools/pmc_summary.py stats $(find $O/serial_$V -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_c3_$V.csv; cut -c1-100 $O/kernel_stats_serial_c3_$V.csv | head -22; rm -rf $O/serial_$V
done
unset GSWT_ORDER GSWT_VS
timeout -k 10 300 python bench.py > $O/bench_c3.json 2>$O/bench_c3.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --order depth > $O/bench_c3_depth.json 2>$O/bench_c3_depth.err; echo "bench depth rc=$?"
timeout -k 10 300 python bench.py --vertex-stage v2 --no-cpu-baseline > $O/bench_c3_v2.json 2>$O/bench_c3_v2.err; echo "bench v2 rc=$?"
timeout -k 10 300 python bench.py --order depth --graph --no-cpu-baseline > $O/bench_c3_depth_graph.json 2>$O/bench_c3_depth_graph.err; echo "bench depth graph rc=$?"
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'],1), round((d.get('static_camera') or {}).get('value',0),1), d['roofline']['frac'], d['roofline'].get('kernel_ms_isolated'), (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('max_abs_diff_vs_gpu'), d.get('stage_ms'))" $f; done
