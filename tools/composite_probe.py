#!/usr/bin/env python3
"""k_composite alone, one frame at a time, for a list of (segment, debug flags) settings: kernel time from its own events."""
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
ref = None
for seg, flags in ((512, 0), (256, 0), (1024, 0), (512, 1), (512, 2), (512, 4), (512, 2 + 32), (512, 2 + 64), (512, 2 + 96)):
    r.set_option(L.GSWT_OPT_SEGMENT, seg)
    r.set_option(L.GSWT_OPT_DEBUG_FLAGS, flags)
    ts = []
    for i in range(12):
        r.render_wait(r.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5))
        ts.append(r.timings())
    ts = ts[4:]
    img = out.cpu().numpy()
    if ref is None:
        ref = img
    print(f"seg {seg:4d} flags {flags}: k_composite {1e3 * np.median([t['ms_composite_kernel'] for t in ts]):.1f} us, composite stage {1e3 * np.median([t['ms_composite'] for t in ts]):.1f} us, "
          f"frame {1e3 * np.median([t['ms_total'] for t in ts]):.1f} us, max|d| vs first {float(np.abs(img - ref).max()):.2e}")
r.set_option(L.GSWT_OPT_DEBUG_FLAGS, 0)
