#!/usr/bin/env python3
"""k_project stage time (k_cull + k_project + k_totals between the frame's stage events) of one library build, one frame at a time: A/B of kernel variants
is done by running this once per GSWT_HIP_LIB in the same job."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L
for name in sys.argv[1:] or ["c3"]:
    w, wang, cu, vp, sort = bench.build_workload(name)
    W, H = w["width"], w["height"]
    su = wang.scene_uniforms()
    r = GSWTRenderer(0)
    r.set_option(L.GSWT_OPT_TIMING, 2)
    if os.environ.get("GSWT_DBG_FLAGS"):             # kernel-variant bits (gswt_kernels.hip: 0x200 / 0x400 = 2 / 4 chunks per k_project workgroup)
        r.set_option(L.GSWT_OPT_DEBUG_FLAGS, int(os.environ["GSWT_DBG_FLAGS"], 0))
    wang.upload_to(r)
    r.configure(wang.height_map() if int(wang.user.surface_type) == 1 else None)
    r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    ts = []
    for i in range(40):
        r.render_wait(r.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5))
        ts.append(r.timings())
    ts = ts[8:]
    print(f"{os.path.basename(L.LIB_PATH)} flags {os.environ.get('GSWT_DBG_FLAGS', '0')} {name}: project stage {1e3 * np.median([t['ms_project'] for t in ts]):.1f} us (min {1e3 * min(t['ms_project'] for t in ts):.1f}), "
          f"composite kernel {1e3 * np.median([t['ms_composite_kernel'] for t in ts]):.1f} us, frame {1e3 * np.median([t['ms_total'] for t in ts]):.1f} us, checksum {float(out.double().sum().item()):.6f}", flush=True)
    r.close()
