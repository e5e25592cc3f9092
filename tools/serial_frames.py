#!/usr/bin/env python3
"""N frames of a workload one at a time (no overlap): for per-kernel isolated durations under rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w, wang, cu, vp, sort = bench.build_workload(name)
W, H = w["width"], w["height"]
su = wang.scene_uniforms()
r = GSWTRenderer(0)
r.set_option(L.GSWT_OPT_TIMING, 0)
if os.environ.get("GSWT_SEGMENT"):                 # kernel-variant sweeps: pairs per compositor work item, ablation / variant bits
    r.set_option(L.GSWT_OPT_SEGMENT, int(os.environ["GSWT_SEGMENT"], 0))
if os.environ.get("GSWT_DBG_FLAGS"):
    r.set_option(L.GSWT_OPT_DEBUG_FLAGS, int(os.environ["GSWT_DBG_FLAGS"], 0))
if os.environ.get("GSWT_VS", "") == "v2":
    r.set_option(L.GSWT_OPT_STRICT_VS, 0)
if os.environ.get("GSWT_NO_CHUNK_CULL"):
    r.set_option(L.GSWT_OPT_NO_CHUNK_CULL, 1)
if os.environ.get("GSWT_ITEM_ORDER"):
    r.set_option(L.GSWT_OPT_ITEM_ORDER, int(os.environ["GSWT_ITEM_ORDER"]))
if os.environ.get("GSWT_DEPTH_SORT"):
    r.set_option(L.GSWT_OPT_DEPTH_SORT, int(os.environ["GSWT_DEPTH_SORT"]))
if os.environ.get("GSWT_COMPOSITE"):
    r.set_option(L.GSWT_OPT_COMPOSITE, int(os.environ["GSWT_COMPOSITE"]))
order = L.GSWT_ORDER_DEPTH if os.environ.get("GSWT_ORDER", "") == "depth" else L.GSWT_ORDER_REFERENCE
wang.upload_to(r)
r.configure(wang.height_map() if int(wang.user.surface_type) == 1 else None)
r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
shard = None
if os.environ.get("GSWT_SHARD"):                   # "rank,world[,cols]": one rank's share of a sharded frame
    a = os.environ["GSWT_SHARD"].split(",")
    shard = (int(a[0]), int(a[1]), "cols") if len(a) > 2 else (int(a[0]), int(a[1]))
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
for i in range(n):
    r.render_wait(r.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5, order_mode=order, **({"shard": shard} if shard else {})))
print("done", r.timings()["n_pairs"])
