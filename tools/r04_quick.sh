#!/bin/bash
# quick GPU check: a test subset, serial per-kernel stats of c3 (reference / depth order), two bench lines.  usage: tools/r04_quick.sh <tag> [pytest args]
set -o pipefail
O=gpurun_out/${1:-r4q}; mkdir -p $O; export TMPDIR=/tmp
shift
timeout -k 10 900 python -m pytest ${@:-tests -m gpu} -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
for V in ref depth; do
  [ $V = depth ] && export GSWT_ORDER=depth || unset GSWT_ORDER
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$V -- python3 tools/serial_frames.py c3 20 > $O/serial_$V.log 2>&1
  python3 tools/pmc_summary.py stats $(find $O/serial_$V -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_c3_$V.csv; cut -c1-100 $O/kernel_stats_serial_c3_$V.csv | head -24; rm -rf $O/serial_$V
done
unset GSWT_ORDER
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_c3.json 2>$O/bench_c3.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --order depth --no-cpu-baseline > $O/bench_c3_depth.json 2>$O/bench_c3_depth.err; echo "bench depth rc=$?"
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'],1), round((d.get('static_camera') or {}).get('value',0),1), d['roofline']['frac'], d['roofline'].get('kernel_ms_isolated'), d.get('stage_ms'))" $f; done
