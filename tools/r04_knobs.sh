#!/bin/bash
# The run-time knobs re-swept on the round's final kernels (c3 fly path, bench.py --no-cpu-baseline --static-steps 0; two rounds, alternating).
# usage: bash tools/r04_knobs.sh <tag> [workload]
set -o pipefail
TAG=${1:-r04k}; WL=${2:-c3}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
run() { # label, env assignments (string), bench args
  env $2 timeout -k 10 300 python bench.py --no-cpu-baseline --static-steps 0 --workload $WL $3 > $O/b.json 2>> $O/bench.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2].ljust(34), 'value', round(d['value'],1), 'segment', d['segment'], 'kernel_ms', round(d['roofline']['kernel_ms'],5))" $O/b.json "$1" | tee -a $O/lines.txt; }
for i in 1 2; do
  run "default" "A=1" ""
  run "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=2" ""
  run "GPU_MAX_HW_QUEUES=6" "GPU_MAX_HW_QUEUES=6" ""
  run "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8" ""
  run "layout c012p3s" "GSWT_STREAM_LAYOUT=c012p3s" ""
  run "layout c01234s" "GSWT_STREAM_LAYOUT=c01234s" ""
  run "sort 256 threads" "GSWT_SORT_WIDE_MAX_M=0" ""
  run "--graph" "A=1" "--graph"
  run "--segment 1024" "A=1" "--segment 1024"
  run "--segment 2048" "A=1" "--segment 2048"
  run "--segment 3072" "A=1" "--segment 3072"
  run "--composite 1 (decoupled waves)" "A=1" "--composite 1"
  run "--composite 2 (no k_combine)" "A=1" "--composite 2"
  run "--item-order 1" "A=1" "--item-order 1"
  run "--in-flight 4" "A=1" "--in-flight 4"
  run "--vertex-stage v2" "A=1" "--vertex-stage v2"
  run "GSWT_EMIT_TAB=0" "GSWT_EMIT_TAB=0" ""
  run "--t-eps 0 (no early-out)" "A=1" "--t-eps 0"
done
echo knobs done
