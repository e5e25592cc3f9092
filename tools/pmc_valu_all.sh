#!/bin/bash
# VALU / SALU / LDS wave-instructions and waves of every kernel of a serial frame, plus the kernels' durations (two runs).  GPU box:
# bash tools/pmc_valu_all.sh <workload>
WL=${1:-c3}
export TMPDIR=/tmp
D=gpurun_out/pmc_valu_$WL
mkdir -p $D
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $D/p -- python3 tools/serial_frames.py $WL 8 > $D/p.log 2>&1 || tail -3 $D/p.log
python3 tools/pmc_summary.py pmc $D/p $D/valu.json
rocprofv3 --kernel-trace --stats --output-format csv -d $D/s -- python3 tools/serial_frames.py $WL 20 > $D/s.log 2>&1 || tail -3 $D/s.log
python3 tools/pmc_summary.py stats $(find $D/s -name "*kernel_stats.csv" | head -1) $D/stats.csv
python3 - $D <<'PY'
import json, sys, csv
d = json.load(open(sys.argv[1] + "/valu.json"))
t = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1] + "/stats.csv"))}
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0)):
    if "gswt" in k:
        print(f"{k[:48]:48s} VALU {v.get('SQ_INSTS_VALU', 0) / 1e6:7.2f} M  SALU {v.get('SQ_INSTS_SALU', 0) / 1e6:6.2f} M  LDS {v.get('SQ_INSTS_LDS', 0) / 1e6:5.2f} M  waves {v.get('SQ_WAVES', 0):8.0f}  avg {float(t.get(k, {}).get('AverageNs', 0)) / 1e3:7.1f} us")
PY
rm -rf $D/p $D/s
