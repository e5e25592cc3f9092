#!/usr/bin/env python3
"""Experiment (not part of the product): (a) frames/s with K independent contexts, each on its own
stream, rendering the c3 frame concurrently; (b) k_composite ablations through GSWT_OPT_DEBUG_FLAGS."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the ablation / variant bits of GSWT_OPT_DEBUG_FLAGS exist only in the measurement build (`make -C gswt_renderer_amd/csrc variants`)
os.environ.setdefault("GSWT_HIP_LIB", os.path.join(ROOT, "build_var", "libgswt_hip_exp.so"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L

w, wang, cu, vp, sort = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "c3")
W, H = w["width"], w["height"]
su = wang.scene_uniforms()
dev = torch.device("cuda", 0)


def make_ctx():
    r = GSWTRenderer(0)
    st = torch.cuda.Stream(device=dev)
    r.set_stream(st.cuda_stream)
    r.set_option(L.GSWT_OPT_TIMING, 1)
    wang.upload_to(r)
    r.configure(None)
    r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
    outs = [torch.empty((H, W, 4), dtype=torch.float32, device=dev) for _ in range(r.frame_slots())]
    return r, st, outs


ctxs = [make_ctx() for _ in range(3)]
for r, st, outs in ctxs:           # warm (pair capacity converges)
    for i in range(3):
        r.render_wait(r.render_async(cu, su, W, H, outs[0].data_ptr(), transmittance_eps=1e-5))


def run(K, n):
    infl = [[] for _ in range(K)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        k = i % K
        r, st, outs = ctxs[k]
        infl[k].append(r.render_async(cu, su, W, H, outs[(i // K) % r.frame_slots()].data_ptr(), transmittance_eps=1e-5))
        if len(infl[k]) == r.frame_slots():
            r.render_wait(infl[k].pop(0))
    for k in range(K):
        while infl[k]:
            ctxs[k][0].render_wait(infl[k].pop(0))
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


for K in (1, 2, 3):
    run(K, 30)
    print(f"contexts={K}: {run(K, 300):.1f} frames/s", flush=True)

# ablations on one context (serial frames)
r, st, outs = ctxs[0]
r.set_option(L.GSWT_OPT_TIMING, 2)
acc = {}
for i in range(30):
    r.render_wait(r.render_async(cu, su, W, H, outs[0].data_ptr(), transmittance_eps=1e-5))
    if i >= 10:
        for k, v in r.timings().items():
            if k.startswith("ms_"):
                acc.setdefault(k, []).append(v)
print("serial stage times (us):", {k: round(float(np.median(v)) * 1e3, 1) for k, v in acc.items()}, flush=True)
r.set_option(L.GSWT_OPT_TIMING, 1)
for flags, name in [(0, "full"), (1, "no walk"), (2, "stage only"), (4, "no staging (fixed)")]:
    r.set_option(L.GSWT_OPT_DEBUG_FLAGS, flags)
    ts = []
    for i in range(20):
        r.render_wait(r.render_async(cu, su, W, H, outs[0].data_ptr(), transmittance_eps=1e-5))
        ts.append(r.timings()["ms_composite_kernel"])
    print(f"composite [{name}]: {np.median(ts[5:])*1e3:.1f} us", flush=True)
r.set_option(L.GSWT_OPT_DEBUG_FLAGS, 0)
for seg in (256, 512, 1024):
    r.set_option(L.GSWT_OPT_SEGMENT, seg)
    ts = []
    for i in range(20):
        r.render_wait(r.render_async(cu, su, W, H, outs[0].data_ptr(), transmittance_eps=1e-5))
        ts.append(r.timings()["ms_composite_kernel"])
    print(f"composite seg={seg}: {np.median(ts[5:])*1e3:.1f} us", flush=True)
