#!/bin/bash
# The bench lines kept under profiles/ (one GPU): default c3 line (with the CPU baseline), dense / HeightMap / Sphere / c5 variants,
# graph mode, the device-side worker, the PCIe-inclusive rate and the GPU tests' skip reasons.  Usage: bash tools/run_bench_lines.sh <outdir>
O=${1:-gpurun_out/lines}; mkdir -p $O
python bench.py > $O/bench_c3.json 2> $O/bench_c3.err
python bench.py --graph --no-cpu-baseline > $O/bench_c3_graph.json 2>> $O/bench_c3.err
python bench.py --workload c3d > $O/bench_c3d.json 2> $O/bench_c3d.err
python bench.py --workload c3h --no-cpu-baseline > $O/bench_c3h.json 2> $O/bench_c3h.err
python bench.py --workload c3s > $O/bench_c3s.json 2> $O/bench_c3s.err
python bench.py --workload c5 > $O/bench_c5_passes.json 2> $O/bench_c5.err
python bench.py --device-worker --no-cpu-baseline > $O/bench_c3_device_worker.json 2> $O/bench_dw.err
python tools/pcie_rate.py c3 100 > $O/pcie_rate_c3.txt 2> $O/pcie.err
python tools/host_cost.py > $O/host_cost.txt 2>> $O/pcie.err
python -m pytest tests -m gpu -q -rs 2>&1 | grep -i "skip" > $O/gpu_test_skips.txt
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], round(d['value'],1), round((d.get('static_camera') or {}).get('value',0),1), d['roofline']['frac'], d['roofline'].get('frac_isolated'), (d.get('cpu_baseline') or {}).get('value'), (d.get('cpu_baseline') or {}).get('max_abs_diff_vs_gpu'))" $f; done
cat $O/pcie_rate_c3.txt $O/host_cost.txt $O/gpu_test_skips.txt
