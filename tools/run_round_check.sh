set -o pipefail
O=gpurun_out/$1; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -2 $O/gpu_tests.log
for WL in c3 c5; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial_$WL -- python3 tools/serial_frames.py $WL 20 > $O/serial_$WL.log 2>&1
python3 tools/pmc_summary.py stats $(find $O/serial_$WL -name "*kernel_stats.csv" | head -1) $O/kernel_stats_serial_$WL.csv; cut -c1-110 $O/kernel_stats_serial_$WL.csv | head -14; rm -rf $O/serial_$WL
timeout -k 10 250 python bench.py --workload $WL --no-cpu-baseline > $O/bench_$WL.json 2>$O/bench_$WL.err
python3 -c "
import json,sys;d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]);print(sys.argv[1],d['value'],d['static_camera']['value'],d['roofline']['kernel_ms_isolated'])" $O/bench_$WL.json
done
