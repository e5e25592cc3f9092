#!/bin/bash
# Round-4 evidence run on the GPU box: the whole GPU suite, the bench lines kept under profiles/, rocprofv3 --stats of the default /
# static / serial commands, PMC passes, serial per-kernel stats of the other workloads and the depth order, one rank's share of N.
# usage: bash tools/r04_campaign.sh <tag> [part ...]   parts: tests lines prof serial shard
set -o pipefail
TAG=${1:-r04}; shift
PARTS=${@:-tests lines prof serial shard}
O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has tests; then
  timeout -k 10 900 python -m pytest tests -m gpu -q -rs > $O/gpu_tests.log 2>&1; echo "suite rc=$?"; tail -2 $O/gpu_tests.log; grep -i "skip" $O/gpu_tests.log > $O/gpu_test_skips.txt
fi
line() { echo "$1 $(python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(round(d['value'],1), round((d.get('static_camera') or {}).get('value',0),1), 'steady', round((d.get('steady_state') or {}).get('value',0),1), 'frac', round(d['roofline']['frac'],3), 'iso', d['roofline'].get('kernel_ms_isolated'), 'cpu', (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('max_abs_diff_vs_gpu'), 'mem', d.get('device_memory_in_use_GB'))" $1 2>&1)"; }
if has lines; then
  timeout -k 10 400 python bench.py > $O/bench_c3.json 2> $O/bench_c3.err; line $O/bench_c3.json
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver_cmd.json 2>> $O/bench_c3.err; line $O/bench_c3_driver_cmd.json
  timeout -k 10 300 python bench.py --graph --no-cpu-baseline > $O/bench_c3_graph.json 2>> $O/bench_c3.err; line $O/bench_c3_graph.json
  timeout -k 10 400 python bench.py --order depth > $O/bench_c3_depth.json 2> $O/bench_c3_depth.err; line $O/bench_c3_depth.json
  timeout -k 10 300 python bench.py --order depth --graph --no-cpu-baseline > $O/bench_c3_depth_graph.json 2>> $O/bench_c3_depth.err; line $O/bench_c3_depth_graph.json
  timeout -k 10 300 python bench.py --order depth --depth-sort 1 --no-cpu-baseline > $O/bench_c3_depth_global_passes.json 2>> $O/bench_c3_depth.err; line $O/bench_c3_depth_global_passes.json
  timeout -k 10 300 python bench.py --no-chunk-cull --no-cpu-baseline > $O/bench_c3_no_chunk_cull.json 2>> $O/bench_c3.err; line $O/bench_c3_no_chunk_cull.json
  timeout -k 10 300 python bench.py --vertex-stage v2 --no-cpu-baseline > $O/bench_c3_v2.json 2>> $O/bench_c3.err; line $O/bench_c3_v2.json
  timeout -k 10 400 python bench.py --workload c3d > $O/bench_c3d.json 2> $O/bench_c3d.err; line $O/bench_c3d.json
  timeout -k 10 300 python bench.py --workload c3d --composite 1 --no-cpu-baseline > $O/bench_c3d_dw.json 2>> $O/bench_c3d.err; line $O/bench_c3d_dw.json
  timeout -k 10 300 python bench.py --workload c3d --order depth --no-cpu-baseline > $O/bench_c3d_depth.json 2>> $O/bench_c3d.err; line $O/bench_c3d_depth.json
  timeout -k 10 300 python bench.py --workload c3d --order depth --depth-sort 1 --no-cpu-baseline > $O/bench_c3d_depth_global_passes.json 2>> $O/bench_c3d.err; line $O/bench_c3d_depth_global_passes.json
  timeout -k 10 300 python bench.py --workload c3h --no-cpu-baseline > $O/bench_c3h.json 2> $O/bench_c3h.err; line $O/bench_c3h.json
  timeout -k 10 300 python bench.py --workload c3h --no-chunk-cull --no-cpu-baseline > $O/bench_c3h_no_chunk_cull.json 2>> $O/bench_c3h.err; line $O/bench_c3h_no_chunk_cull.json
  timeout -k 10 400 python bench.py --workload c5 --no-chunk-cull --no-cpu-baseline > $O/bench_c5_no_chunk_cull.json 2>> $O/bench_c5.err; line $O/bench_c5_no_chunk_cull.json
  timeout -k 10 400 python bench.py --workload c3s > $O/bench_c3s.json 2> $O/bench_c3s.err; line $O/bench_c3s.json
  timeout -k 10 600 python bench.py --workload c5 > $O/bench_c5_passes.json 2> $O/bench_c5.err; line $O/bench_c5_passes.json
  timeout -k 10 400 python bench.py --workload c5 --order depth --no-cpu-baseline > $O/bench_c5_depth.json 2>> $O/bench_c5.err; line $O/bench_c5_depth.json
  timeout -k 10 300 python bench.py --device-worker --no-cpu-baseline > $O/bench_c3_device_worker.json 2> $O/bench_dw.err; line $O/bench_c3_device_worker.json
  GSWT_BENCH_FAKE_WORLD=8 timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_c3_fake_world8.json 2> $O/bench_fw8.err; line $O/bench_c3_fake_world8.json
  timeout -k 10 200 python tools/pcie_rate.py c3 100 > $O/pcie_rate_c3.txt 2> $O/pcie.err; cat $O/pcie_rate_c3.txt
  timeout -k 10 200 python tools/host_cost.py > $O/host_cost.txt 2>> $O/pcie.err; cat $O/host_cost.txt
fi
if has prof; then
  timeout -k 10 1000 bash tools/profile_c3.sh $TAG c3 > $O/profile.log 2>&1; tail -3 $O/profile.log
  cp gpurun_out/prof_$TAG/kernel_stats_flypath.csv $O/ 2>/dev/null; cp gpurun_out/prof_$TAG/kernel_stats_static.csv $O/ 2>/dev/null; cp gpurun_out/prof_$TAG/kernel_stats_serial.csv $O/ 2>/dev/null
  cp gpurun_out/prof_$TAG/pmc_summary.json $O/ 2>/dev/null; cp gpurun_out/prof_$TAG/bench_flypath_under_rocprof.json $O/ 2>/dev/null
  rm -rf gpurun_out/prof_$TAG/stats_* gpurun_out/prof_$TAG/pmc
fi
if has serial; then
  for WL in c3h c5 c3d; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$WL -- python3 tools/serial_frames.py $WL 20 > $O/serial_$WL.log 2>&1
    python3 tools/pmc_summary.py stats $(find $O/serial_$WL -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_$WL.csv; rm -rf $O/serial_$WL
  done
  for WL in c3 c3h c5; do
    GSWT_NO_CHUNK_CULL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_n$WL -- python3 tools/serial_frames.py $WL 20 > $O/serial_nocull_$WL.log 2>&1
    python3 tools/pmc_summary.py stats $(find $O/serial_n$WL -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_${WL}_no_chunk_cull.csv; rm -rf $O/serial_n$WL
  done
  for WL in c3 c3d; do
    GSWT_ORDER=depth GSWT_DEPTH_SORT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_g$WL -- python3 tools/serial_frames.py $WL 20 > $O/serial_depth_global_$WL.log 2>&1
    python3 tools/pmc_summary.py stats $(find $O/serial_g$WL -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_${WL}_depth_global_passes.csv; rm -rf $O/serial_g$WL
  done
  for WL in c3 c3d c5; do
    GSWT_ORDER=depth rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_d$WL -- python3 tools/serial_frames.py $WL 20 > $O/serial_depth_$WL.log 2>&1
    python3 tools/pmc_summary.py stats $(find $O/serial_d$WL -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_${WL}_depth.csv; rm -rf $O/serial_d$WL
  done
  head -14 $O/kernel_stats_serial_c3_depth.csv | cut -c1-100
fi
if has shard; then
  timeout -k 10 300 python tools/shard_emulation.py c3 100 > $O/shard_emulation_c3.txt 2>&1; tail -12 $O/shard_emulation_c3.txt
  timeout -k 10 100 python3 tools/tile_lengths.py c3 c3d c3h > $O/tile_lengths.txt 2>&1
  GSWT_GRAPH=1 timeout -k 10 300 python tools/shard_emulation.py c3 100 > $O/shard_emulation_c3_graph.txt 2>&1; tail -8 $O/shard_emulation_c3_graph.txt
fi
echo campaign done
