#!/usr/bin/env python3
"""Lane / step statistics of k_composite's walk from a -DGSWT_STATS build of the library (GSWT_HIP_LIB=<that .so>)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
w, wang, cu, vp, sort = bench.build_workload(name)
W, H = w["width"], w["height"]
su = wang.scene_uniforms()
r = GSWTRenderer(0)
wang.upload_to(r); r.configure(None)
r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
r.render_wait(r.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5))
lib = L.load()
a = (C.c_ulonglong * 8)(); lib.gswt_debug_stats(a); a0 = list(a)
r.render_wait(r.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5))
lib.gswt_debug_stats(a); d = [x - y for x, y in zip(a, a0)]
t = r.timings()
steps, covered, zero_steps, batches, nmax_sum, cnt_sum, zero_groups = d[0], d[1], d[2], d[3], d[4], d[5], d[6]
print(f"pairs {t['n_pairs']}: wave-steps {steps}, covered lanes {covered} ({covered / (64.0 * steps):.3f} of lanes), steps with no covered lane {zero_steps} ({zero_steps / steps:.3f})")
print(f"  wave-batches {batches}, sum of longest lists {nmax_sum}, sum of all four lists {cnt_sum} (padding {1 - cnt_sum / (4.0 * nmax_sum):.3f} of group-steps)")
print(f"  16-lane group-steps with no covered pixel (padding included) {zero_groups} of {4 * steps} = {zero_groups / (4.0 * steps):.3f}; real entries {cnt_sum}: "
      f"zero-coverage real entries ~ {zero_groups - (4 * steps - cnt_sum)} ({(zero_groups - (4 * steps - cnt_sum)) / cnt_sum:.3f} of real entries)")
