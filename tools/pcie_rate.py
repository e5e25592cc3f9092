#!/usr/bin/env python3
"""PCIe-inclusive frame rate: the same static-camera frames with the framebuffer handed back in HOST memory
(gswt_render, out_on_device = 0: one 33 MB device-to-host copy per 1080p frame, frames one at a time) and with a pinned host
buffer filled by an overlapped asynchronous copy (three frames in flight).  Never bench.py's `value`; DESIGN.md section 8 quotes it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
w, wang, cu, vp, sort = bench.build_workload(name)
W, H = w["width"], w["height"]
su = wang.scene_uniforms()
r = GSWTRenderer(0)
r.set_option(L.GSWT_OPT_TIMING, 0)
wang.upload_to(r)
r.configure(wang.height_map() if int(wang.user.surface_type) == 1 else None)
r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
for _ in range(5):
    img = r.render(cu, su, W, H, transmittance_eps=1e-5)
t0 = time.perf_counter()
for _ in range(n):
    img = r.render(cu, su, W, H, transmittance_eps=1e-5)
t1 = time.perf_counter()
print(f"{name}: gswt_render into pageable host memory, one frame at a time: {n / (t1 - t0):.1f} frames/s ({(t1 - t0) / n * 1e3:.3f} ms/frame, {W * H * 16 / 1e6:.1f} MB per frame over PCIe)")
# overlapped: device frames + asynchronous copies into pinned host buffers on a copy stream
slots = 3
dev = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(slots)]
host = [torch.empty((H, W, 4), dtype=torch.float32).pin_memory() for _ in range(slots)]
copy_stream = torch.cuda.Stream()
done = [None] * slots
tickets = [None] * slots
def finish(k):
    r.render_wait(tickets[k])
    with torch.cuda.stream(copy_stream):
        host[k].copy_(dev[k], non_blocking=True)
        done[k] = torch.cuda.Event(); done[k].record(copy_stream)
    tickets[k] = None
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    k = i % slots
    if tickets[k] is not None:
        finish(k)
    if done[k] is not None:
        done[k].synchronize()
    tickets[k] = r.render_async(cu, su, W, H, dev[k].data_ptr(), transmittance_eps=1e-5)
for k in range(slots):
    if tickets[k] is not None:
        finish(k)
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"{name}: three frames in flight + asynchronous copies into pinned host buffers: {n / (t1 - t0):.1f} frames/s ({(t1 - t0) / n * 1e3:.3f} ms/frame)")
