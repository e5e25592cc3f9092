#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r4f}; mkdir -p $O; export TMPDIR=/tmp
timeout -k 5 300 python -m pytest tests/test_composite_dw_gpu.py tests/test_baseline_configs_gpu.py tests/test_render_parity_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for WL in c3 c3h c5; do
for HV in 2 1; do
  export GSWT_PROJECT_HALVES=$HV
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$WL$HV -- python3 tools/serial_frames.py $WL 20 > $O/serial_$WL$HV.log 2>&1
  python3 tools/pmc_summary.py stats $(find $O/serial_$WL$HV -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_${WL}_halves$HV.csv; echo "== $WL halves=$HV"; grep "k_project\|k_cull\|k_totals" $O/kernel_stats_serial_${WL}_halves$HV.csv | cut -c1-100; rm -rf $O/serial_$WL$HV
done; done
unset GSWT_PROJECT_HALVES
timeout -k 5 200 python tools/composite_ab.py c5 c3h 2>&1 | tee $O/composite_ab_c5.txt | tail -8
for HV in 2 1; do
GSWT_PROJECT_HALVES=$HV timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_c3_halves$HV.json 2>$O/bench_c3_halves$HV.err; echo "bench halves=$HV rc=$?"
done
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'],1), round((d.get('static_camera') or {}).get('value',0),1), d['roofline']['frac'], d.get('stage_ms'))" $f; done
