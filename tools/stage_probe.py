#!/usr/bin/env python3
"""Stage times (hipEvents, timing level 2) of N frames run one at a time: a quick A/B of kernel variants
(GSWT_HIP_LIB=<variant .so>).  Usage: stage_probe.py <workload> [frames]"""
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
r.set_option(L.GSWT_OPT_TIMING, 2)
if os.environ.get("GSWT_SEGMENT"):
    r.set_option(L.GSWT_OPT_SEGMENT, int(os.environ["GSWT_SEGMENT"]))      # pairs per compositor work item
if os.environ.get("GSWT_EXPANDED") is not None:
    r.set_option(L.GSWT_OPT_EXPANDED_LISTS, int(os.environ["GSWT_EXPANDED"]))
wang.upload_to(r)
r.configure(wang.height_map() if int(wang.user.surface_type) == 1 else None)
r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
acc = {}
for i in range(n + 3):
    r.render_wait(r.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5))
    if i >= 3:
        for k, v in r.timings().items():
            if k.startswith("ms_"):
                acc[k] = acc.get(k, 0.0) + v / n
print(os.path.basename(os.environ.get("GSWT_HIP_LIB", "default")), "segment=" + os.environ.get("GSWT_SEGMENT", "default"), name, " ".join(f"{k[3:]}={v * 1e3:.1f}" for k, v in acc.items()), "us; pairs", r.timings()["n_pairs"], "checksum", float(out.sum()))
