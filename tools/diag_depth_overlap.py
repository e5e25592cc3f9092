#!/usr/bin/env python3
"""Depth-ordered c3 frames one at a time against the same frames with every slot in flight (launch by launch and as one hipGraph per frame),
after frames that leave the slots in other shapes (a dense scene, a small framebuffer with lists beyond the LDS buffer): prints every frame
that differs and where.  usage: tools/diag_depth_overlap.py [first workload]   (GSWT_DIAG_REPS: repetitions, default 3)"""
import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import bench
from gswt_renderer_amd.renderer import GSWTRenderer
from gswt_renderer_amd import _lib as L, host, workloads
w, wang, cu, vp, sort = bench.build_workload("c3")
W, Hh = w["width"], w["height"]
su = wang.scene_uniforms()
r = GSWTRenderer(0)
if len(sys.argv) > 1:                       # a dense frame first: the pair capacity (and with it the sort's payload path) stays large
    w2, wang2, cu2, vp2, sort2 = bench.build_workload(sys.argv[1])
    wang2.upload_to(r); r.configure(None)
    r.set_draws(sort2.draws, sort2.merged_gs_index, sort2.merged_map_id, sort2.merged_lod_id)
    img = r.render(cu2, wang2.scene_uniforms(), w2["width"], w2["height"], order_mode=L.GSWT_ORDER_DEPTH)
    print("first", sys.argv[1], r.timings()["n_pairs"], r.depth_stats())
wang.upload_to(r); r.configure(None)
r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
cam = workloads.camera_for("c3")
cams = []
for k in range(6):
    pos = (cam["pos"][0] + 0.15 * k, cam["pos"][1] + 0.4 * k, cam["pos"][2])
    tgt = (cam["target"][0] + 0.15 * k, cam["target"][1] + 0.4 * k, cam["target"][2] - 0.05 * k)
    cams.append(host.camera_uniforms(pos, tgt, cam["up"], cam["fovy"], cam["near"], cam["far"], W, Hh)[0])
# (a small framebuffer in front of the scene first: lists beyond the LDS buffer, as tests/test_depth_order_gpu.py leaves the slots)
cu_s, vp_s = host.camera_uniforms(cam["pos"], cam["target"], cam["up"], cam["fovy"], cam["near"], cam["far"], 256, 144)
for g in (0, 1, 0):
    r.set_option(L.GSWT_OPT_GRAPH, g)
    r.set_option(L.GSWT_OPT_TIMING, 0)
    r.render(cu_s, su, 256, 144, order_mode=L.GSWT_ORDER_DEPTH)
print("small frames done", r.depth_stats())
outs = [torch.empty((Hh, W, 4), dtype=torch.float32, device="cuda") for _ in cams]
r.set_option(L.GSWT_OPT_TIMING, 0)
REPS = int(os.environ.get('GSWT_DIAG_REPS', '3'))
for rep in range(REPS):
    want = []
    r.set_option(L.GSWT_OPT_GRAPH, 0)
    for c, o in zip(cams, outs):
        r.render_wait(r.render_async(c, su, W, Hh, o.data_ptr(), order_mode=L.GSWT_ORDER_DEPTH, transmittance_eps=1e-5))
        want.append(o.cpu().numpy().copy())
    for mode in (0, 1):
        r.set_option(L.GSWT_OPT_GRAPH, mode)
        tickets = []
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        for c, o in zip(cams, outs):
            if len(tickets) >= r.frame_slots():
                r.render_wait(tickets.pop(0))
            tickets.append(r.render_async(c, su, W, Hh, o.data_ptr(), order_mode=L.GSWT_ORDER_DEPTH, transmittance_eps=1e-5))
        for tk in tickets:
            r.render_wait(tk)
        for i, (a, o) in enumerate(zip(want, outs)):
            b = o.cpu().numpy()
            d = np.abs(a - b)
            if d.max() > 0:
                ys, xs = np.nonzero(d.max(axis=2))
                print(f"rep {rep} graph {mode} frame {i}: max diff {d.max():.3e}, {len(ys)} pixels, tiles {sorted(set(zip((ys//16).tolist(), (xs//16).tolist())))[:6]}")
print("diag done", r.depth_stats())
