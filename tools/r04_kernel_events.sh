#!/bin/bash
# Do the events bound to the k_composite dispatch (GSWT_LAUNCH_TIMED) read what rocprofv3 reports?  The default bench command under
# rocprofv3 --kernel-trace --stats with both event methods, and the plain lines beside them (does the bound pair cost frames?).
# usage: bash tools/r04_kernel_events.sh <tag>
set -o pipefail
TAG=${1:-r04ke}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_edge_cases_gpu.py tests/test_graph_gpu.py tests/test_render_parity_gpu.py tests/test_composite_dw_gpu.py -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -1 $O/tests.log
cmp() { python3 - $1 $2 "$3" <<'P'
import csv, json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["roofline"]
row = next(x for x in csv.DictReader(open(sys.argv[2])) if "k_composite" in x["Name"])
print(sys.argv[3], "value", round(d["value"], 1), "| bench kernel_ms", round(r["kernel_ms"], 5), "all slots", round(r["kernel_ms_all_slots"], 5), "samples", r["kernel_ms_samples"],
      "| rocprofv3 average", round(float(row["AverageNs"]) / 1e6, 5), "over", row["Calls"], "| frac", round(r["frac"], 4), "isolated", r.get("kernel_ms_isolated"))
P
}
for KE in 1 0; do
  GSWT_KERNEL_EVENTS=$KE rocprofv3 --kernel-trace --stats --output-format csv -d $O/st_$KE -- python3 bench.py --no-cpu-baseline --static-steps 0 > $O/bench_prof_ke$KE.json 2> $O/st_$KE.log
  python3 tools/pmc_summary.py stats $(find $O/st_$KE -name "*kernel_stats.csv" | head -1) $O/kernel_stats_ke$KE.csv; rm -rf $O/st_$KE
  cmp $O/bench_prof_ke$KE.json $O/kernel_stats_ke$KE.csv "under rocprofv3, GSWT_KERNEL_EVENTS=$KE:" | tee -a $O/lines.txt
done
for i in 1 2; do for KE in 1 0; do
  GSWT_KERNEL_EVENTS=$KE timeout -k 10 300 python bench.py --no-cpu-baseline --static-steps 0 > $O/b.json 2>> $O/bench.err
  python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('plain, GSWT_KERNEL_EVENTS='+sys.argv[2]+':', 'value', round(d['value'],1), 'kernel_ms', round(r['kernel_ms'],5), 'all slots', round(r['kernel_ms_all_slots'],5), 'frac', round(r['frac'],4), 'isolated', round(r['kernel_ms_isolated'],5))" $O/b.json $KE | tee -a $O/lines.txt
done; done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>> $O/bench.err; python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print('driver command:', 'value', round(d['value'],1), 'steady', round(d['steady_state']['value'],1), 'kernel_ms', round(r['kernel_ms'],5), 'frac', round(r['frac'],4), r['kernel_ms_source'][:40])" $O/bench_driver_cmd.json | tee -a $O/lines.txt
echo events done
