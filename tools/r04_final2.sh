#!/bin/bash
# Round-4 closing run on the final tree: GPU suite, smoke, the driver-command and default bench lines, the other workloads' lines, the
# per-kernel --stats one frame at a time and of the default command.   usage: bash tools/r04_final2.sh <tag>
set -o pipefail
TAG=${1:-r04u}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -rs > $O/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -2 $O/gpu_tests.log
timeout -k 10 200 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], 'value', round(d['value'],1), 'steady', round((d.get('steady_state') or {}).get('value',0),1), 'static', round((d.get('static_camera') or {}).get('value',0),1), 'kernel_ms', round(r['kernel_ms'],5), 'frac', round(r['frac'],4), 'isolated', round(r.get('kernel_ms_isolated') or 0,5), 'frac_isolated', round(r.get('frac_isolated') or 0,4), 'cpu', (d.get('cpu_baseline') or {}).get('value'))" $1 "$2" | tee -a $O/lines.txt; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver_cmd.json 2> $O/bench.err; line $O/bench_c3_driver_cmd.json "c3 --steps 20 --warmup 5:"
timeout -k 10 400 python bench.py > $O/bench_c3.json 2>> $O/bench.err; line $O/bench_c3.json "c3 default:"
timeout -k 10 400 python bench.py --order depth --no-cpu-baseline > $O/bench_c3_depth.json 2>> $O/bench.err; line $O/bench_c3_depth.json "c3 --order depth:"
for WL in c3h c3d c5; do timeout -k 10 400 python bench.py --no-cpu-baseline --workload $WL > $O/bench_$WL.json 2>> $O/bench.err; line $O/bench_$WL.json "$WL:"; done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 tools/serial_frames.py c3 30 > $O/serial.log 2>&1
python3 tools/pmc_summary.py stats $(find $O/serial -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_c3.csv; rm -rf $O/serial
rocprofv3 --kernel-trace --stats --output-format csv -d $O/fly -- python3 bench.py --no-cpu-baseline --static-steps 0 > $O/bench_flypath_under_rocprof.json 2> $O/fly.log
python3 tools/pmc_summary.py stats $(find $O/fly -name "*kernel_stats.csv" | head -1) $O/kernel_stats_flypath_c3.csv; rm -rf $O/fly
head -12 $O/kernel_stats_serial_c3.csv | cut -c1-110
echo final2 done
