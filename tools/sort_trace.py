#!/usr/bin/env python3
"""Phase stamps of k_radix_scatter (both passes of the frame's pair sort) from the -DGSWT_TRACE build (GSWT_HIP_LIB=build_var/libgswt_hip_trace.so):
[0] entry, [1] keys / values / count rows loaded, [2] digits counted, [3] scans done, [4] ranked into LDS, [5] copied out (100 MHz ticks)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
lib = L.load()
lib.gswt_debug_trace.argtypes = [C.c_void_p, C.c_uint]
N = 1 << 17
for i in range(4):
    r.render_wait(r.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5))
t = r.timings()
buf = np.zeros((N, 8), dtype=np.uint64)
assert lib.gswt_debug_trace(buf.ctypes.data, N) == 0
b = buf[N // 2: N // 2 + 8192].astype(np.int64)
print(f"{name}: pairs {t['n_pairs']}, sort stage {1e3 * t['ms_sort']:.1f} us")
for p, lbl in ((0, "pass 1 (low 8 bits)"), (1, "pass 2 (high bits)")):
    x = b[p * 4096:(p + 1) * 4096]
    x = x[x[:, 5] > 0]
    if not len(x):
        continue
    t0 = x[:, 0].min()
    ph = np.diff(x[:, :6], axis=1) / 100.0
    print(f"  {lbl}: {len(x)} blocks, entries within {(x[:, 0].max() - t0) / 100.0:.2f} us, last exit {(x[:, 5].max() - t0) / 100.0:.2f} us; "
          f"mean us per phase: loads {ph[:, 0].mean():.2f}, count {ph[:, 1].mean():.2f}, scans {ph[:, 2].mean():.2f}, rank {ph[:, 3].mean():.2f}, copy-out {ph[:, 4].mean():.2f}; "
          f"block lifetime mean {(x[:, 5] - x[:, 0]).mean() / 100.0:.2f} max {(x[:, 5] - x[:, 0]).max() / 100.0:.2f}")
