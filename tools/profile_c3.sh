#!/bin/bash
# Evidence for profiles/: rocprofv3 kernel-trace stats of the default bench command, then PMC passes (each in its own
# run, --kernel-trace only beside --pmc) on a short serial-ish run.  Usage (on the GPU box): bash tools/profile_c3.sh <tag> [bench args]
set -o pipefail
TAG=${1:-s2}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT/pmc
export TMPDIR=/tmp
[ -n "$SKIP_STATS" ] || rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline "$@" > $OUT/bench_under_rocprof.json 2> $OUT/stats.log || exit 1
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32" \
         "GRBM_GUI_ACTIVE" "VALUBusy" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  D=$OUT/pmc/$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 4 "$@" > $D.json 2> $D.log || { echo "pmc pass $C failed"; tail -3 $D.log; }
  echo "pmc $C done"
done
python3 tools/pmc_summary.py pmc $OUT/pmc $OUT/pmc_summary.json
[ -n "$SKIP_STATS" ] || python3 tools/pmc_summary.py stats $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cat $OUT/pmc_summary.json | head -40
