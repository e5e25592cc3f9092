#!/usr/bin/env python3
"""Phase stamps of k_project's workgroups from the -DGSWT_TRACE build (GSWT_HIP_LIB=build_var/libgswt_hip_trace.so), one frame at a time:
[0] entry, [1] launch-table entry known, [2] list word arrived, [3] record arrived, [4] wave 0 through the projection, [5] behind the
workgroup barrier, [6] end, [7] pairs of the chunk (100 MHz ticks).  Prints the live workgroups' phase durations, how many are resident over
time and what the dispatcher does meanwhile."""
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
b = buf[8192:8192 + 49152].astype(np.int64)
seen = b[b[:, 0] > 0]
live = seen[(seen[:, 6] > seen[:, 0]) & (seen[:, 1] >= seen[:, 0])]      # (rows that were live in an earlier frame only keep stale stamps)
t0 = np.percentile(seen[:, 0], 0.1)
us = lambda x: x / 100.0
print(f"{name}: project stage {1e3 * t['ms_project']:.1f} us; workgroups stamped {len(seen)}, live {len(live)} (the others left behind the launch table)")
print(f"  entries: first {us(seen[:, 0].min() - t0):.2f}, last {us(seen[:, 0].max() - t0):.2f} us; last live exit {us(live[:, 6].max() - t0):.2f} us")
full = live[(live[:, 2] > 0) & (live[:, 3] > 0)]
ph = lambda a, bb: us((full[:, bb] - full[:, a]))
names = ["launch table", "list word", "record gather", "projection (wave 0)", "barrier", "sums + atomics"]
for k, nm in enumerate(names):
    d = ph(k, k + 1)
    print(f"  {nm:22s} mean {d.mean():6.2f} us  median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}")
life = us(live[:, 6] - live[:, 0])
print(f"  lifetime mean {life.mean():.2f} us, median {np.median(life):.2f}, p90 {np.percentile(life, 90):.2f}; sum {life.sum() / 1e3:.1f} k workgroup-us "
      f"= {life.sum() / 2048:.1f} us x 2048 slots")
# residency over time
ev = np.concatenate([np.stack([live[:, 0] - t0, np.ones(len(live))], 1), np.stack([live[:, 6] - t0, -np.ones(len(live))], 1)])
ev = ev[np.argsort(ev[:, 0], kind="stable")]
res = np.cumsum(ev[:, 1])
for q in (5, 10, 20, 30, 40, 50, 60, 70):
    i = np.searchsorted(ev[:, 0], q * 100)
    print(f"  t = {q:3d} us: {int(res[min(i, len(res) - 1)])} live workgroups resident, {int((live[:, 0] - t0 < q * 100).sum())} started")
hw = live[:, 7].astype(np.uint64)
has_pairs = (hw >> np.uint64(63)) != 0
hwid = (hw & np.uint64(0xFFFFFFFF)).astype(np.int64); xcc = ((hw >> np.uint64(32)) & np.uint64(0xFF)).astype(np.int64)
# HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (the XCC comes from XCC_ID)
cu_key = (xcc & 0xF) * 4096 + ((hwid >> 13) & 7) * 256 + ((hwid >> 12) & 1) * 16 + ((hwid >> 8) & 15)
print(f"  distinct CUs seen: {len(np.unique(cu_key))}; XCCs {sorted(set((xcc & 0xF).tolist()))}")
peak = []
for k in np.unique(cu_key):
    m = cu_key == k
    e = np.concatenate([np.stack([live[m, 0], np.ones(m.sum())], 1), np.stack([live[m, 6], -np.ones(m.sum())], 1)])
    e = e[np.argsort(e[:, 0], kind="stable")]
    peak.append(np.cumsum(e[:, 1]).max())
peak = np.array(peak)
print(f"  live workgroups resident on one CU at once: max {int(peak.max())}, median of the CUs' peaks {np.median(peak):.0f}, min {int(peak.min())}; workgroups per CU {np.bincount(np.unique(cu_key, return_inverse=True)[1]).mean():.1f}")
heavy = live[has_pairs]
live7 = np.where(has_pairs, 1, 0)
print(f"  chunks with pairs: {len(heavy)}; their lifetime mean {us(heavy[:, 6] - heavy[:, 0]).mean():.2f} us; without pairs {us((live[~has_pairs][:, 6] - live[~has_pairs][:, 0])).mean() if (~has_pairs).any() else 0:.2f} us")

# one CU's timeline: residency sampled every 2 us, and the gaps between a workgroup leaving and the next one arriving
for pick in (0, 100):
    k = np.unique(cu_key)[pick]
    m = cu_key == k
    st = np.sort(live[m, 0] - t0) / 100.0; en = np.sort(live[m, 6] - t0) / 100.0
    samples = [int((st <= q).sum() - (en <= q).sum()) for q in range(2, 66, 4)]
    print(f"  CU {pick}: {m.sum()} live workgroups; resident at t = 2, 6, .. us: {samples}")
    print(f"     first starts {np.round(st[:12], 1).tolist()}")
# how the start rate compares with what the free slots would allow
st_all = np.sort(live[:, 0] - t0) / 100.0
print("  starts per us in [10, 50) us:", round(((st_all >= 10) & (st_all < 50)).sum() / 40.0, 1), "; ends per us:", round((((live[:, 6] - t0) / 100.0 >= 10) & ((live[:, 6] - t0) / 100.0 < 50)).sum() / 40.0, 1))
dead = seen[~((seen[:, 6] > seen[:, 0]) & (seen[:, 1] >= seen[:, 0]))]
dst = np.sort(dead[:, 0] - t0) / 100.0
print(f"  workgroups that left behind the launch table: {len(dead)}; their entries: 1 % {np.percentile(dst, 1):.1f}, 50 % {np.percentile(dst, 50):.1f}, 99 % {np.percentile(dst, 99):.1f} us")
