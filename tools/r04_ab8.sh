#!/bin/bash
# k_emit with ONE barrier for its four chunks' scans (instead of a barrier pair per chunk): GPU suite on the new build, then the kernel's own
# time one frame at a time under rocprofv3 (tools/serial_frames.py) for the shipped build and build_var/libgswt_hip_prev.so, then the fly path A/B.
# usage: bash tools/r04_ab8.sh <tag>
set -o pipefail
TAG=${1:-r04ab8}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -q -rs > $O/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -1 $O/gpu_tests.log
timeout -k 10 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
for WL in c3 c5; do for L in shipped prev; do
  LIB=""; [ $L != shipped ] && LIB=$PWD/build_var/libgswt_hip_$L.so
  export GSWT_HIP_LIB=$LIB
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_${WL}_$L -- python3 tools/serial_frames.py $WL 20 > $O/serial_${WL}_$L.log 2>&1
  python3 tools/pmc_summary.py stats $(find $O/serial_${WL}_$L -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_${WL}_$L.csv; rm -rf $O/serial_${WL}_$L
  echo "== $WL $L"; grep "k_emit\|k_project\|k_composite" $O/kernel_stats_serial_${WL}_$L.csv | cut -c1-110
done; done
unset GSWT_HIP_LIB
AB_TESTS="" bash tools/r04_ab_lib.sh $TAG prev 2 c5
