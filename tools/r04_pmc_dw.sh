#!/bin/bash
# PMC passes for k_composite (GSWT_COMPOSITE=0) and k_composite_dw (=1), frames one at a time: VALU / SALU / LDS wave-instructions, wave-cycles,
# waiting share.  usage: bash tools/r04_pmc_dw.sh <workload>
WL=${1:-c3}
export TMPDIR=/tmp
O=gpurun_out/pmc_dw_$WL; mkdir -p $O
for CV in 0 1; do
  export GSWT_COMPOSITE=$CV; mkdir -p $O/v$CV
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32"; do
    D=$O/v$CV/$(echo $C | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 tools/serial_frames.py $WL 8 > $D.log 2>&1 || tail -2 $D.log
  done
  python3 tools/pmc_summary.py pmc $O/v$CV $O/pmc_v$CV.json
done
python3 - $O <<'PY'
import json, sys
for v in (0, 1):
    d = json.load(open(f"{sys.argv[1]}/pmc_v{v}.json"))
    for k, c in d.items():
        if "k_composite" in k:
            cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
            print(f"variant {v} {k[:40]:40s} VALU {c.get('SQ_INSTS_VALU',0)/1e6:6.2f} M  SALU {c.get('SQ_INSTS_SALU',0)/1e6:6.2f} M  LDS {c.get('SQ_INSTS_LDS',0)/1e6:5.2f} M  "
                  f"wave-cycles {c.get('SQ_WAVE_CYCLES',0)/1e6:7.1f} M  waiting {100*c.get('SQ_WAIT_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',1)):4.1f} %  issue-wait {100*c.get('SQ_WAIT_INST_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',1)):4.1f} %  "
                  f"cycles {cyc/1e3:6.1f} k  LDS conflicts {100*c.get('SQ_LDS_BANK_CONFLICT',0)/max(1,c.get('SQ_LDS_IDX_ACTIVE',1)):4.1f} %")
PY
rm -rf $O/v0 $O/v1
