#!/usr/bin/env python3
"""k_composite against k_composite_dw (GSWT_OPT_COMPOSITE), one frame at a time: the kernel's own hipEvent time (median of the last frames)
per workload and segment length.  usage: tools/composite_ab.py [workload ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L

r = GSWTRenderer(0)
for name in (sys.argv[1:] or ["c3", "c3d"]):
    w, wang, cu, vp, sort = bench.build_workload(name)
    W, H = w["width"], w["height"]
    su = wang.scene_uniforms()
    r.set_option(L.GSWT_OPT_TIMING, 1)
    wang.upload_to(r)
    r.configure(wang.height_map() if int(wang.user.surface_type) == 1 else None)
    r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    for seg in ((512, 1536, 4096) if name != "c3" else (512, 1024, 1536, 2048)):
        r.set_option(L.GSWT_OPT_SEGMENT, seg)
        row = []
        for variant in (0, 1, 2):
            r.set_option(L.GSWT_OPT_COMPOSITE, variant)
            ms = []
            for i in range(14):
                r.render_wait(r.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5))
                ms.append(r.timings()["ms_composite_kernel"])
            row.append(1e3 * float(np.median(ms[4:])))
        print(f"{name} seg {seg:5d}: k_composite {row[0]:7.1f} us   k_composite_dw {row[1]:7.1f} us   k_composite<FOLD> {row[2]:7.1f} us   pairs {r.timings()['n_pairs']}")
    r.set_option(L.GSWT_OPT_COMPOSITE, 0)
