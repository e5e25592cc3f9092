#!/bin/bash
set -o pipefail
O=gpurun_out/${1:-r4g}; mkdir -p $O; export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests -m gpu -x -q -s > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log; grep "^frame " $O/tests.log | head -8
echo "== dw poll interval 1 (product) / 4"
timeout -k 5 200 python tools/composite_ab.py c3 c3d 2>&1 | grep seg | tee $O/composite_ab_sleep1.txt
GSWT_HIP_LIB=$PWD/build_var/libgswt_hip_dwsleep4.so timeout -k 5 200 python tools/composite_ab.py c3 c3d 2>&1 | grep seg | tee $O/composite_ab_sleep4.txt
for V in ref depth; do
  [ $V = depth ] && export GSWT_ORDER=depth || unset GSWT_ORDER
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$V -- python3 tools/serial_frames.py c3 20 > $O/serial_$V.log 2>&1
  python3 tools/pmc_summary.py stats $(find $O/serial_$V -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_c3_$V.csv; grep "radix\|Name" $O/kernel_stats_serial_c3_$V.csv | cut -c1-100; rm -rf $O/serial_$V
done
