#!/usr/bin/env python3
"""What bounds the two-frames-in-flight frame rate?  Throughput and one-frame latency with parts of the work ablated
(GSWT_OPT_DEBUG_FLAGS: 1 = compositor without the walk, 4 = without staging, 8 = k_project emits no pairs)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the ablation / variant bits of GSWT_OPT_DEBUG_FLAGS exist only in the measurement build (`make -C gswt_renderer_amd/csrc variants`)
os.environ.setdefault("GSWT_HIP_LIB", os.path.join(ROOT, "build_var", "libgswt_hip_exp.so"))
sys.path.insert(0, ROOT)
import torch
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L

w, wang, cu, vp, sort = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "c3")
W, H = w["width"], w["height"]
su = wang.scene_uniforms()
r = GSWTRenderer(0)
wang.upload_to(r)
r.configure(None)
r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
outs = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(r.frame_slots())]


def throughput(n):
    infl = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        infl.append(r.render_async(cu, su, W, H, outs[i % r.frame_slots()].data_ptr(), transmittance_eps=1e-5))
        if len(infl) == r.frame_slots():
            r.render_wait(infl.pop(0))
    while infl:
        r.render_wait(infl.pop(0))
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def latency(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        r.render_wait(r.render_async(cu, su, W, H, outs[0].data_ptr(), transmittance_eps=1e-5))
    return (time.perf_counter() - t0) / n * 1e6


r.set_option(L.GSWT_OPT_TIMING, 0)
for name, flags in (("full", 0), ("no walk", 1), ("no staging, no walk", 4), ("no pairs at all", 8)):
    r.set_option(L.GSWT_OPT_DEBUG_FLAGS, flags)
    throughput(20); latency(10)
    print(f"{name:22s}: all slots in flight {throughput(200):6.1f} us/frame, one at a time {latency(60):6.1f} us/frame", flush=True)
r.set_option(L.GSWT_OPT_DEBUG_FLAGS, 0)
