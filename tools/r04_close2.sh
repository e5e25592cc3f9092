#!/bin/bash
# Round-4 closing check after the last two changes (events bound to the k_composite dispatch, k_mg_init gone): the GPU suite, smoke, the
# two bench lines, and how the short bracket of the driver's command responds to more warm-up / more steps.   usage: bash tools/r04_close2.sh <tag>
set -o pipefail
TAG=${1:-r04w}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -rs > $O/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -2 $O/gpu_tests.log
timeout -k 10 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], 'value', round(d['value'],1), 'steady', round((d.get('steady_state') or {}).get('value',0),1), 'static', round((d.get('static_camera') or {}).get('value',0),1), 'kernel_ms', round(r['kernel_ms'],5), 'frac', round(r['frac'],4), 'cpu', (d.get('cpu_baseline') or {}).get('value'))" $1 "$2" | tee -a $O/lines.txt; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver_cmd.json 2> $O/bench.err; line $O/bench_c3_driver_cmd.json "--steps 20 --warmup 5:"
timeout -k 10 400 python bench.py > $O/bench_c3.json 2>> $O/bench.err; line $O/bench_c3.json "default:"
for A in "--steps 20 --warmup 5" "--steps 20 --warmup 100" "--steps 40 --warmup 5" "--steps 80 --warmup 5" "--steps 20 --warmup 5"; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --static-steps 0 $A > $O/b.json 2>> $O/bench.err; line $O/b.json "$A (no cpu baseline, no static run):"
done
echo close2 done
