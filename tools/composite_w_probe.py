#!/usr/bin/env python3
"""k_composite (four-wave workgroups) against k_composite_w (one wave per item), one frame at a time: kernel time from its own
events for a list of (debug flags, segment) settings; image difference against the first setting, with and without the early-out."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the ablation / variant bits of GSWT_OPT_DEBUG_FLAGS exist only in the measurement build (`make -C gswt_renderer_amd/csrc variants`)
os.environ.setdefault("GSWT_HIP_LIB", os.path.join(ROOT, "build_var", "libgswt_hip_exp.so"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
settings = [(0, 1536), (0x1000, 1536), (0x1000, 512), (0x1000, 256), (0x1000, 1024), (0x3000, 1536), (0x3000, 512), (0x3000, 256)]
if len(sys.argv) > 2:
    settings = [tuple(int(x, 0) for x in a.split(":")) for a in sys.argv[2:]]
w, wang, cu, vp, sort = bench.build_workload(name)
W, H = w["width"], w["height"]
su = wang.scene_uniforms()
r = GSWTRenderer(0)
r.set_option(L.GSWT_OPT_TIMING, 2)
wang.upload_to(r)
r.configure(wang.height_map() if int(wang.user.surface_type) == 1 else None)
r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
torch.cuda.synchronize()
ref = {}
for eps in (1e-5, 0.0):
    for flags, seg in settings:
        r.set_option(L.GSWT_OPT_SEGMENT, seg)
        r.set_option(L.GSWT_OPT_DEBUG_FLAGS, flags)
        ts = []
        for i in range(12 if eps else 3):
            r.render_wait(r.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=eps))
            ts.append(r.timings())
        ts = ts[4:] if eps else ts[1:]
        img = out.cpu().numpy()
        if eps not in ref:
            ref[eps] = img
        print(f"{name} eps {eps:g} flags {flags:#x} seg {seg:4d}: k_composite {1e3 * np.median([t['ms_composite_kernel'] for t in ts]):.1f} us, "
              f"composite stage {1e3 * np.median([t['ms_composite'] for t in ts]):.1f} us, frame {1e3 * np.median([t['ms_total'] for t in ts]):.1f} us, "
              f"max|d| vs first {float(np.abs(img - ref[eps]).max()):.2e}", flush=True)
r.set_option(L.GSWT_OPT_DEBUG_FLAGS, 0)
r.set_option(L.GSWT_OPT_SEGMENT, L.GSWT_DEFAULT_SEGMENT)
