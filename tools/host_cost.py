#!/usr/bin/env python3
"""Host time of one gswt_render_async call (kernel launches, events, copies) vs the GPU frame time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L

w, wang, cu, vp, sort = bench.build_workload("c3")
W, H = w["width"], w["height"]
su = wang.scene_uniforms()
r = GSWTRenderer(0)
wang.upload_to(r); r.configure(None)
r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
for timing, graph in ((0, 0), (0, 1), (1, 0), (2, 0)):
    r.set_option(L.GSWT_OPT_TIMING, timing)
    r.set_option(L.GSWT_OPT_GRAPH, graph)
    for shard in ((0, 1), (0, 8, "cols")):
        bw = r.shard_cols_padded(W, shard[1]) if len(shard) > 2 else W
        outs = [torch.empty((H, bw, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        torch.cuda.synchronize()
        for i in range(6):
            r.render_wait(r.render_async(cu, su, W, H, outs[0].data_ptr(), transmittance_eps=1e-5, shard=shard))
        ts = []
        for i in range(40):
            t0 = time.perf_counter()
            tk = r.render_async(cu, su, W, H, outs[i % 2].data_ptr(), transmittance_eps=1e-5, shard=shard)
            ts.append(time.perf_counter() - t0)
            r.render_wait(tk)
        ts.sort()
        print(f"timing={timing} graph={graph} shard={shard}: host enqueue median {ts[len(ts)//2]*1e6:.0f} us, min {ts[0]*1e6:.0f} us", flush=True)
