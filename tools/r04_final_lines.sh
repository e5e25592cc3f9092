#!/bin/bash
# Round-4 closing evidence on the final kernels: what the sort events cost the fly path (--freeze-sort A/B), the fly-path --stats,
# the other workloads' lines, a long random parity sweep.   usage: bash tools/r04_final_lines.sh <tag>
set -o pipefail
TAG=${1:-r04y}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
line() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'],1), 'static', round((d.get('static_camera') or {}).get('value',0),1), 'swap-ins', (d.get('sort_events') or {}).get('swapped_in'))" $1 "$2" | tee -a $O/lines.txt; }
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --static-steps 0 > $O/b.json 2>> $O/bench.err && line $O/b.json "c3 default"
  timeout -k 10 300 python bench.py --no-cpu-baseline --static-steps 0 --freeze-sort > $O/b.json 2>> $O/bench.err && line $O/b.json "c3 freeze-sort"
done
GSWT_SWEEP_CASES=336 GSWT_SWEEP_SEED=4242 timeout -k 10 400 python -m pytest tests/test_random_sweep_gpu.py -q > $O/sweep.log 2>&1; echo "sweep rc=$?"; tail -1 $O/sweep.log | tee -a $O/lines.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fly -- python3 bench.py --no-cpu-baseline --static-steps 0 > $O/bench_flypath_under_rocprof.json 2> $O/stats_fly.log
python3 tools/pmc_summary.py stats $(find $O/stats_fly -name "*kernel_stats.csv" | head -1) $O/kernel_stats_flypath.csv; rm -rf $O/stats_fly
head -12 $O/kernel_stats_flypath.csv | cut -c1-120
for WL in c3h c3d c5; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --workload $WL > $O/bench_$WL.json 2>> $O/bench.err && line $O/bench_$WL.json "$WL"
done
echo lines done
