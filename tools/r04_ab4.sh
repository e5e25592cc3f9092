#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r4i}; mkdir -p $O; export TMPDIR=/tmp
timeout -k 5 600 python -m pytest tests/test_composite_dw_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for CV in 0 2; do
  export GSWT_COMPOSITE=$CV
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$CV -- python3 tools/serial_frames.py c3 20 > $O/serial_$CV.log 2>&1
  python3 tools/pmc_summary.py stats $(find $O/serial_$CV -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_c3_composite$CV.csv; echo "== composite variant $CV"; grep "k_composite\|k_combine\|k_items" $O/kernel_stats_serial_c3_composite$CV.csv | cut -c1-110; rm -rf $O/serial_$CV
done
unset GSWT_COMPOSITE
for CV in 0 2 0 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --composite $CV > $O/bench_c3_composite${CV}_$RANDOM.json 2>$O/bench.err; echo "bench composite=$CV rc=$?"
done
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'],1), round((d.get('static_camera') or {}).get('value',0),1), d['roofline']['frac'])" $f; done
