#!/usr/bin/env python3
"""Per-stage times (hipEvents, one frame at a time) of rank r's share of an N-GPU column-band frame, on one GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, wang, cu, vp, sort = bench.build_workload(name)
W, H = w["width"], w["height"]
su = wang.scene_uniforms()
r = GSWTRenderer(0)
r.set_option(L.GSWT_OPT_TIMING, 2)
wang.upload_to(r)
r.configure(None)
r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
cols = r.shard_cols_padded(W, N) if N > 1 else W
out = torch.empty((H, cols, 4), dtype=torch.float32, device="cuda")
keys = ("ms_project", "ms_emit", "ms_sort", "ms_ranges", "ms_composite", "ms_composite_kernel", "ms_total")
for rank in range(N):
    acc = []
    for i in range(12):
        r.render_wait(r.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5, shard=(rank, N, "cols") if N > 1 else (0, 1)))
        t = r.timings()
        if i >= 4:
            acc.append([t[k] * 1e3 for k in keys])
    m = np.median(np.array(acc), axis=0)
    print(f"rank {rank}/{N}: " + " ".join(f"{k[3:]}={v:.0f}" for k, v in zip(keys, m)) + f" us; visible {t['n_visible']} pairs {t['n_pairs']}", flush=True)
