#!/bin/bash
# frames/s of bench.py's fly path / static camera for several compositor work-item sizes (GPU box): bash tools/segment_sweep.sh <workload> seg ...
WL=$1; shift
for S in "$@"; do
  python3 bench.py --workload $WL --steps 480 --warmup 48 --no-cpu-baseline --segment $S 2> gpurun_out/seg_$WL_$S.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$WL segment', d['segment'], 'fly', round(d['value'],1), 'static', round((d.get('static_camera') or {}).get('value',0),1), 'k_composite ms', d['roofline']['kernel_ms_min_slot'], d['roofline'].get('kernel_ms_isolated'))"
done
