#!/bin/bash
# Where k_project's VALU instructions go: one PMC pass per ablation flag of the kernel (GSWT_DBG_FLAGS: 256 = nothing behind the launch
# table, 128 = stop in front of the record gather, 16 = stop after the frustum cull, 8 = no stores / pairs, 0 = the whole kernel).
# Usage (GPU box): bash tools/pmc_project_ablation.sh <workload>
export GSWT_HIP_LIB=${GSWT_HIP_LIB:-$PWD/build_var/libgswt_hip_exp.so}   # ablation / variant bits live in the measurement build only (make variants)
set -o pipefail
WL=${1:-c3}
OUT=gpurun_out/pmc_project_$WL
mkdir -p $OUT
export TMPDIR=/tmp
for FL in 0 8 16 128 256; do
  export GSWT_DBG_FLAGS=$FL
  D=$OUT/f$FL
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d $D -- python3 tools/serial_frames.py $WL 8 > $D.log 2>&1 || { echo "pass $FL failed"; tail -3 $D.log; }
  python3 tools/pmc_summary.py pmc $D $D.json
  python3 - $D.json $FL <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "k_project" in k or "k_emit" in k:
        print("flags", sys.argv[2], k, {c: round(x) for c, x in sorted(v.items())})
PY
  rm -rf $D
done
