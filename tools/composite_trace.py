#!/usr/bin/env python3
"""Per-work-item phase stamps of the compositor from a -DGSWT_TRACE build of the library (GSWT_HIP_LIB=build_var/libgswt_hip_trace.so):
one frame per (debug flags, segment) setting, the raw stamps saved to gpurun_out/trace_<workload>_<flags>_<seg>.npy and a summary printed.
Build: hipcc <Makefile flags> -DGSWT_TRACE -shared gswt_kernels.hip gswt_passes.hip gswt_api.hip gswt_worker.hip -o build_var/libgswt_hip_trace.so"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
settings = [tuple(int(x, 0) for x in a.split(":")) for a in sys.argv[2:]] or [(0, 1536), (0x1000, 512)]
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
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for flags, seg in settings:
    r.set_option(L.GSWT_OPT_SEGMENT, seg)
    r.set_option(L.GSWT_OPT_DEBUG_FLAGS, flags)
    for i in range(4):
        r.render_wait(r.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5))
    t = r.timings()
    buf = np.zeros((N, 8), dtype=np.uint64)
    assert lib.gswt_debug_trace(buf.ctypes.data, N) == 0
    n_items = int(np.count_nonzero(buf[:, 1]))          # items past the real count never stamp [1]; stale rows of earlier settings are cut by the caller
    np.save(os.path.join(ROOT, "gpurun_out", f"trace_{name}_{flags:x}_{seg}.npy"), buf[:min(N, 40000)])
    print(f"{name} flags {flags:#x} seg {seg}: k_composite {1e3 * t['ms_composite_kernel']:.1f} us, pairs {t['n_pairs']}, stamped items {n_items}", flush=True)
r.set_option(L.GSWT_OPT_DEBUG_FLAGS, 0)
r.set_option(L.GSWT_OPT_SEGMENT, L.GSWT_DEFAULT_SEGMENT)
