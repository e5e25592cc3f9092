#!/usr/bin/env python3
"""What one rank of an N-GPU run does, measured on ONE GPU: shard r of N of the c3 frame (interleaved tile rows, and
contiguous tile-column bands with per-band draw culling), every frame slot in flight, no all-gather.  max over r of the
per-rank frame time bounds the N-GPU frame rate from above."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
w, wang, cu, vp, sort = bench.build_workload(name)
W, H = w["width"], w["height"]
su = wang.scene_uniforms()
r = GSWTRenderer(0)
r.set_option(L.GSWT_OPT_TIMING, 0)
r.set_option(L.GSWT_OPT_GRAPH, int(os.environ.get("GSWT_GRAPH", "0")))      # GSWT_GRAPH=1: one hipGraphLaunch per frame
if os.environ.get("GSWT_SEGMENT"):                 # pairs per compositor work item (default: the library's)
    r.set_option(L.GSWT_OPT_SEGMENT, int(os.environ["GSWT_SEGMENT"]))
print(f"workload {name}, GSWT_OPT_GRAPH = {os.environ.get('GSWT_GRAPH', '0')}, GSWT_OPT_SEGMENT = {os.environ.get('GSWT_SEGMENT', 'as bench.py picks it (auto_segment)')}", flush=True)
if os.environ.get("GSWT_SHARD_CONFIGS"):           # e.g. "cols8,cols4"
    only = os.environ["GSWT_SHARD_CONFIGS"].split(",")
wang.upload_to(r)
r.configure(wang.height_map() if int(wang.user.surface_type) == 1 else None)
r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
configs = [("rows", 1), ("rows", 2), ("rows", 4), ("rows", 8), ("cols", 2), ("cols", 4), ("cols", 8)]
if name == "c5":
    configs = [("rows", 1), ("rows", 8), ("cols", 2), ("cols", 4), ("cols", 8)]
if os.environ.get("GSWT_SHARD_CONFIGS"):
    configs = [(m, n) for m, n in configs if f"{m}{n}" in only]
for mode, N in configs:
    rows = r.shard_rows_padded(H, N) if (N > 1 and mode == "rows") else H
    cols = r.shard_cols_padded(W, N) if (N > 1 and mode == "cols") else W
    outs = [torch.empty((rows, cols, 4), dtype=torch.float32, device="cuda") for _ in range(r.frame_slots())]
    worst = 0.0
    vis = []
    per_rank = []
    for rank in range(N):
        def run(n):
            infl = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n):
                infl.append(r.render_async(cu, su, W, H, outs[i % r.frame_slots()].data_ptr(), transmittance_eps=1e-5, shard=(rank, N, mode) if mode == "cols" else (rank, N)))
                if len(infl) == r.frame_slots():
                    r.render_wait(infl.pop(0))
            while infl:
                r.render_wait(infl.pop(0))
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        if not os.environ.get("GSWT_SEGMENT"):         # as bench.py picks it: from the rank's own first frame
            r.render_wait(r.render_async(cu, su, W, H, outs[0].data_ptr(), transmittance_eps=1e-5, shard=(rank, N, mode) if mode == "cols" else (rank, N)))
            t0f = r.timings()
            r.set_option(L.GSWT_OPT_SEGMENT, bench.auto_segment(float(t0f["n_pairs"]) / max(1.0, float(t0f["n_tiles"])), float(t0f["n_pairs"])))
        run(10)
        t_rank = min(run(n_frames), run(n_frames))          # best of two: one allocation growth or clock ramp inside a 100-frame run is not the rank's rate
        per_rank.append(t_rank)
        worst = max(worst, t_rank)
        vis.append(r.timings()["n_visible"])
    print(f"{mode} N={N}: slowest rank {worst * 1e6:.0f} us/frame -> <= {1.0 / worst:.0f} frames/s before the all-gather "
          f"({rows * cols * 16 / 1e6:.1f} MB per rank; visible splats per rank {min(vis)}..{max(vis)}; us per rank {' '.join(f'{t * 1e6:.0f}' for t in per_rank)})", flush=True)
