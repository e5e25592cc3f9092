#!/usr/bin/env python3
"""Phase stamps of k_emit from the -DGSWT_TRACE build (GSWT_HIP_LIB=build_var/libgswt_hip_trace.so): per workgroup (four consecutive chunks)
[0] entry, [1] every load has arrived, [2] exit, [3] chunks with pairs, [4] pairs (100 MHz ticks).  usage: tools/emit_trace.py [workload]"""
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
x = buf[81920:81920 + 8192].astype(np.int64)
x = x[x[:, 2] > 0]
t0 = x[:, 0].min()
life = (x[:, 2] - x[:, 0]) / 100.0
load = (x[:, 1] - x[:, 0]) / 100.0
live = x[:, 3] > 0
print(f"{name}: pairs {t['n_pairs']}, emit stage {1e3 * t['ms_emit']:.1f} us; {len(x)} workgroups stamped, {int(live.sum())} with pairs")
print(f"  entries: first 0, median {np.median(x[:, 0] - t0) / 100.0:.2f}, last {(x[:, 0].max() - t0) / 100.0:.2f} us; last exit {(x[:, 2].max() - t0) / 100.0:.2f} us")
print(f"  workgroups without pairs: lifetime mean {life[~live].mean():.2f} us (p90 {np.percentile(life[~live], 90):.2f})")
print(f"  workgroups with pairs: loads {load[live].mean():.2f} us, lifetime mean {life[live].mean():.2f} (p50 {np.median(life[live]):.2f}, p90 {np.percentile(life[live], 90):.2f}, max {life[live].max():.2f})")
for nl in (1, 2, 3, 4):
    m = x[:, 3] == nl
    if m.any():
        print(f"    {nl} chunks with pairs: {int(m.sum())} workgroups, lifetime mean {life[m].mean():.2f} us, pairs mean {x[m, 4].mean():.0f} max {x[m, 4].max()}")
big = np.argsort(-life)[:5]
print("  longest lives:", [(round(float(life[i]), 2), int(x[i, 3]), int(x[i, 4])) for i in big], "(us, chunks with pairs, pairs)")
print(f"  sum of lifetimes {life.sum() / 1e3:.1f} k workgroup-us = {life.sum() / 2048.0:.1f} us x 2048 slots")
for tt in (2, 4, 6, 8, 10, 12, 14, 16):
    res = int(((x[:, 0] - t0) <= tt * 100) .sum() - ((x[:, 2] - t0) <= tt * 100).sum())
    print(f"  t = {tt:3d} us: {res} workgroups resident, {int(((x[:, 0] - t0) <= tt * 100).sum())} started")
