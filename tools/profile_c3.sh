#!/bin/bash
# Evidence for profiles/: rocprofv3 kernel-trace stats of the default bench command (fly path), of the static-camera command and
# of frames run one at a time, then PMC passes (each in its own run, --kernel-trace only beside --pmc) on a short static-camera run.
# Usage (on the GPU box): bash tools/profile_c3.sh <tag> [workload]
set -o pipefail
TAG=${1:-r2}; WL=${2:-c3}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT/pmc
export TMPDIR=/tmp
if [ -z "$SKIP_STATS" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fly -- python3 bench.py --no-cpu-baseline --workload $WL --static-steps 0 > $OUT/bench_flypath_under_rocprof.json 2> $OUT/stats_fly.log || exit 1
  python3 tools/pmc_summary.py stats $(find $OUT/stats_fly -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_flypath.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_static -- python3 bench.py --no-cpu-baseline --workload $WL --mode static --steps 300 --warmup 30 > $OUT/bench_static_under_rocprof.json 2> $OUT/stats_static.log || exit 1
  python3 tools/pmc_summary.py stats $(find $OUT/stats_static -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_static.csv
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -- python3 tools/serial_frames.py $WL 30 > $OUT/serial.log 2>&1 || exit 1
  python3 tools/pmc_summary.py stats $(find $OUT/stats_serial -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_serial.csv
fi
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32" \
         "GRBM_GUI_ACTIVE" "VALUBusy" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  D=$OUT/pmc/$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $D -- python3 bench.py --no-cpu-baseline --workload $WL --mode static --steps 6 --warmup 4 > $D.json 2> $D.log || { echo "pmc pass $C failed"; tail -3 $D.log; }
  echo "pmc $C done"
done
python3 tools/pmc_summary.py pmc $OUT/pmc $OUT/pmc_summary.json
echo profile done
