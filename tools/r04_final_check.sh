#!/bin/bash
# Round-4 closing check on the GPU box: the GPU suite, smoke, the default / driver-command bench lines on the final kernels.
# usage: bash tools/r04_final_check.sh <tag>
set -o pipefail
TAG=${1:-r04z}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -rs > $O/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -2 $O/gpu_tests.log
timeout -k 10 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver_cmd.json 2> $O/bench.err; echo "driver cmd rc=$?"
timeout -k 10 400 python bench.py > $O/bench_c3.json 2>> $O/bench.err; echo "default rc=$?"
python3 - $O <<'P'
import json, sys
for n in ("bench_c3_driver_cmd", "bench_c3"):
    d = json.loads(open(f"{sys.argv[1]}/{n}.json").read().strip().splitlines()[-1])
    print(n, round(d["value"], 1), "static", round((d.get("static_camera") or {}).get("value", 0), 1), "steady", round((d.get("steady_state") or {}).get("value", 0), 1),
          "frac", round(d["roofline"]["frac"], 3), "cpu", (d.get("cpu_baseline") or {}).get("value"))
P
echo check done
