#!/usr/bin/env python3
"""k_project under its ablation flags, one frame at a time: stage time (cull + project + totals) per flag set."""
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
for flags, what in ((0, "full"), (16, "stop after the frustum cull"), (128, "stop behind the list word (no record gather)"), (256, "return behind the launch-table entry"), (8, "no record / rect stores, no pairs"), (24, "both")):
    r.set_option(L.GSWT_OPT_DEBUG_FLAGS, flags)
    ts = []
    for i in range(12):
        r.render_wait(r.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5))
        ts.append(r.timings())
    ts = ts[4:]
    print(f"flags {flags:3d} ({what}): project stage {1e3 * np.median([t['ms_project'] for t in ts]):.1f} us, frame {1e3 * np.median([t['ms_total'] for t in ts]):.1f} us, "
          f"visible {ts[-1]['n_visible']}, pairs {ts[-1]['n_pairs']}")
r.set_option(L.GSWT_OPT_DEBUG_FLAGS, 0)
