#!/bin/bash
# k_project one frame at a time with each ablation flag (see pmc_project_ablation.sh): where its time goes.  GPU box: bash tools/project_ablation_times.sh
export GSWT_HIP_LIB=${GSWT_HIP_LIB:-$PWD/build_var/libgswt_hip_exp.so}   # ablation / variant bits live in the measurement build only (make variants)
mkdir -p gpurun_out/abl_t
export TMPDIR=/tmp
for FL in 0 8 16 128 256; do
  export GSWT_DBG_FLAGS=$FL
  D=gpurun_out/abl_t/f$FL
  rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/serial_frames.py c3 20 > $D.log 2>&1
  F=$(find $D -name "*kernel_stats.csv" | head -1)
  python3 tools/pmc_summary.py stats $F $D.csv
  echo "flags $FL: $(grep k_project $D.csv)"
  rm -rf $D
done
