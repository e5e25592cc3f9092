#!/bin/bash
# A/B of two builds of the device library on the fly path: the shipped one against build_var/libgswt_hip_<name>.so (GSWT_HIP_LIB), alternating.
# usage: bash tools/r04_ab_lib.sh <tag> <name> [rounds] [workloads...]
set -o pipefail
TAG=${1:-r04ab}; NAME=${2:-prev}; R=${3:-3}; shift 3; WLS=${@:-c3}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
if [ -n "$AB_TESTS" ]; then timeout -k 10 600 python -m pytest $AB_TESTS -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -1 $O/tests.log; fi
for WL in $WLS; do for i in $(seq $R); do for L in shipped $NAME; do
  LIB=""; [ $L != shipped ] && LIB=$PWD/build_var/libgswt_hip_$L.so
  GSWT_HIP_LIB=$LIB timeout -k 10 300 python bench.py --no-cpu-baseline --workload $WL $AB_ARGS > $O/b.json 2>> $O/bench.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[3].ljust(4), sys.argv[2].ljust(8), 'value', round(d['value'],1), 'static', round((d.get('static_camera') or {}).get('value',0),1), 'kernel_ms', round(r['kernel_ms'],5), 'isolated', round(r['kernel_ms_isolated'],5))" $O/b.json $L $WL | tee -a $O/lines.txt
done; done; done
echo ab done
