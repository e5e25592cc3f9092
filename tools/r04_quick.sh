#!/bin/bash
# quick check on the GPU box: chosen tests, serial --stats of c3 (+ others), fly-path lines
set -o pipefail
O=gpurun_out/${1:-r4x}; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest ${TESTS:-tests/test_composite_dw_gpu.py tests/test_render_parity_gpu.py tests/test_baseline_configs_gpu.py tests/test_random_sweep_gpu.py tests/test_depth_order_gpu.py} -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -1 $O/tests.log
for WL in ${WLS:-c3}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s_$WL -- python3 tools/serial_frames.py $WL 20 > $O/s_$WL.log 2>&1
  python3 tools/pmc_summary.py stats $(find $O/s_$WL -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_${WL}.csv; rm -rf $O/s_$WL
  echo "== $WL"; head -${LINES:-8} $O/kernel_stats_serial_${WL}.csv | cut -c1-110
done
for A in "${BENCH_A:-}" "${BENCH_B:---order depth}" "${BENCH_C:---workload c3d}"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline $A > $O/bench.json 2>> $O/bench.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2:], round(d['value'],1), round((d.get('static_camera') or {}).get('value',0),1))" $O/bench.json $A | tee -a $O/bench_lines.txt
done
