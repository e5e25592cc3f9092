#!/bin/bash
# Frames in flight re-swept on the round's final kernels: the shipped library (five frame slots) against builds with six and seven
# (hipcc -DGSWT_FRAME_SLOTS=N ... -o build_var/libgswt_hip_slotsN.so; GSWT_HIP_LIB selects the library).  usage: bash tools/r04_slots.sh <tag> [workload]
set -o pipefail
TAG=${1:-r04s}; WL=${2:-c3}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2; do for N in 5 6 7; do
  LIB=""; [ $N != 5 ] && LIB=$PWD/build_var/libgswt_hip_slots$N.so
  GSWT_HIP_LIB=$LIB timeout -k 10 300 python bench.py --no-cpu-baseline --static-steps 0 --workload $WL > $O/b.json 2>> $O/bench.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[3], 'slots', sys.argv[2], 'in flight', d['frames_in_flight'], 'value', round(d['value'],1), 'host_submit', round(d['host_submit_ms_mean'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],5))" $O/b.json $N $WL | tee -a $O/lines.txt
done; done
echo slots done
