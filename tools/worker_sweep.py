"""Long random sweep of the device-side worker stages against libgswt_host (byte equality of cell states, order, records, groups).
Usage: python tools/worker_sweep.py [cases] [seed]   -- the CI-sized version of this is tests/test_worker_gpu.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gswt_renderer_amd import host  # noqa: E402
from gswt_renderer_amd.worker import DeviceWorker  # noqa: E402
from tests import test_worker_gpu as T  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    events = groups = 0
    for case in range(n_cases):
        surface = int(rng.choice([0, 0, 1, 1, 2]))
        sort_type = int(rng.choice([3, 3, 3, 0, 1, 2]))
        merge_type = int(rng.choice([2, 2, 2, 1, 0])) if sort_type == 3 else 2
        if surface == 2:
            k = int(rng.integers(1, 3))
            half = (5 * k, 2 * k)
            merge_type = 2 if merge_type == 1 else merge_type           # the reference's axis merge is documented as broken on the sphere
        else:
            half = (int(rng.integers(1, 9)), int(rng.integers(1, 9)))
        user = dict(surface_type=surface, tile_sort_type=sort_type, merge_type=merge_type, lod_blending=bool(rng.integers(0, 2)),
                    lod_bbox_check=bool(rng.integers(0, 2)), lod_transition_width_ratio=float(rng.uniform(0.02, 0.25)),
                    lod_dist_tolerance=float(rng.choice([0.0, 0.0, 0.3])), merge_topk=int(rng.integers(0, 80)),
                    merge_dot_threshold=float(rng.uniform(0.05, 0.95)), lod_max_dist=float(rng.uniform(3.0, 30.0)),
                    merge_tile_dist=(int(rng.integers(0, 2)), int(rng.integers(2, 5))), sphere_radius=float(rng.uniform(2.0, 8.0)),
                    height_map_scale=(1.0, 1.0, float(rng.uniform(0.2, 3.0))), height_map_wh=(int(rng.integers(4, 20)), int(rng.integers(4, 20))))
        if merge_type == 1 and min(half) * 2 + 1 < 2 * user["merge_tile_dist"][1] + 3:
            half = (max(half[0], 5), max(half[1], 5))
        pipe, _ = T._pipe(half, user, lod0=int(rng.integers(20, 200)), seed=case)
        dw = DeviceWorker(pipe.renderer, pipe.wang)
        for k in range(5):
            span = 4.0 * (2 * max(half) + 1) * 0.6
            pos = tuple(float(x) for x in rng.uniform((-span, -span, 0.3), (span, span, 12.0)))
            tgt = tuple(float(x) for x in rng.uniform((-span, -span, -1.0), (span, span, 2.0)))
            if surface == 2:
                pos = tuple(float(x) for x in rng.normal(size=3) * user["sphere_radius"] * 1.8)
                tgt = (0.0, 0.0, 0.0)
            cu, vp = T._cam(pos, tgt)
            rebuild = k == 0 or pipe.wang.check_update(pos)
            try:
                ref = T._compare(pipe, dw, pos, vp, rebuild=rebuild, tag=f"case {case} {user} half {half} cam {k} {pos} {tgt}")
            except host.GSWTHostError as e:          # reference panics (e.g. axis merge walking off a small map): both sides must refuse
                print(f"case {case}: host refused ({e}); skipped")
                break
            events += 1
            groups += ref["n"][1]
        dw.close()
        print(f"case {case}: surface {surface} sort {sort_type} merge {merge_type} half {half} ok", flush=True)
    print(f"{events} sort events, {groups} merged groups: device == host")


if __name__ == "__main__":
    main()
