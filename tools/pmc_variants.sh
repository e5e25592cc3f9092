#!/bin/bash
# PMC passes (each in its own run, --kernel-trace only beside --pmc) over frames rendered one at a time, for a list of
# compositor variants given as "name:segment:debug-flags".  Usage (GPU box): bash tools/pmc_variants.sh <tag> <workload> name:seg:flags ...
export GSWT_HIP_LIB=${GSWT_HIP_LIB:-$PWD/build_var/libgswt_hip_exp.so}   # ablation / variant bits live in the measurement build only (make variants)
set -o pipefail
TAG=$1; WL=$2; shift 2
OUT=gpurun_out/pmcv_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for V in "$@"; do
  NAME=${V%%:*}; REST=${V#*:}; SEG=${REST%%:*}; FL=${REST#*:}
  export GSWT_SEGMENT=$SEG GSWT_DBG_FLAGS=$FL
  mkdir -p $OUT/$NAME
  for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA"; do
    D=$OUT/$NAME/$(echo $C | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 tools/serial_frames.py $WL 8 > $D.log 2>&1 || { echo "pmc pass $NAME $C failed"; tail -3 $D.log; }
  done
  python3 tools/pmc_summary.py pmc $OUT/$NAME $OUT/$NAME.json
  python3 - $OUT/$NAME.json $NAME <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "k_composite" in k:
        print(sys.argv[2], k, {c: round(x) for c, x in sorted(v.items())})
PY
  rm -rf $OUT/$NAME
  echo "variant $NAME done"
done
