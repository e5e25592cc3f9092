#!/usr/bin/env python3
"""bench.py -- frames/sec of the GSWT hot path on MI355X (BASELINE.json metric).

A "step" is one frame: gswt_render over the resident draw list of the workload (Wang-tile
instancing -> projection -> pair emit -> tile sort -> compositing), inputs already in HBM.
N=1 renders the whole frame on one GPU.  N>1 (launched by torch.distributed.run, one rank per
GPU) shards the frame by contiguous bands of 16-px screen-tile columns: every rank projects only the draws that can
reach its band, composites its band, and the final framebuffer is all-gathered over RCCL (torch.distributed backend
"nccl") and re-assembled on the device; total work is fixed, so scaling is "strong".

Prints ONE JSON line (rank 0) with `roofline` (composite kernel, algorithmic bytes / hipEvent
time on the kernel's own stream) and, at N=1, `cpu_baseline` (the CPU oracle timed on the host
cores for one frame of the same workload).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_workload(name: str, lod0_override: int | None = None):
    from gswt_renderer_amd import host, synth, workloads
    w = dict(workloads.WORKLOADS[name])
    if lod0_override:
        w["lod0"] = lod0_override
    verts = synth.make_tileset(n_lod=w["n_lod"], n_tile=16, lod0_count=w["lod0"])
    ts = host.TileSet.from_vertices(verts)
    wang = host.WangTile(ts)
    user = host.user_data(tile_map_half_wh=w["half"], **w["user"])
    wang.configure(user)
    cam = workloads.camera_for(name)
    cu, vp = host.camera_uniforms(cam["pos"], cam["target"], cam["up"], cam["fovy"], cam["near"], cam["far"], w["width"], w["height"])
    wang.build_tiles(cam["pos"])
    sort = wang.sort_tiles(cam["pos"], vp)
    return w, wang, cu, vp, sort


def cpu_baseline(wang, sort, cu, vp, su, W, H, culling_dist=1.0, passes=None, height_map=None):
    """Times the CPU oracle (oracle/gswt_oracle.c, OpenMP) on ONE frame of the same workload."""
    from oracle import gswt_oracle as orc
    tex, gi, li = wang.preload()
    draws = []
    f32 = np.float32
    for t, d in zip(sort.tiles, sort.draws):
        if t.key_len == 1:     # CPU viewport cull, renderer.rs:472-494 (the HIP path does this on the device)
            c = np.array(t.corners[:], dtype=f32).reshape(4, 3)
            mx = my = f32(np.finfo(np.float32).max)
            mz = -mx
            with np.errstate(all="ignore"):
                for ci in range(4):
                    p = orc.mat4_vec(vp, [c[ci, 0], c[ci, 1], c[ci, 2], f32(1.0)])
                    p = (p[:3] / p[3]).astype(f32)
                    mx, my, mz = min(mx, abs(p[0])), min(my, abs(p[1])), max(mz, p[2])
            if mz < -culling_dist or mx > culling_dist or my > culling_dist:
                continue
        tu = orc.Tile80.from_buffer_copy(bytes(d.tile))
        if d.merged:
            a, b = d.merged_offset, d.merged_offset + d.merged_count
            draws.append(orc.Draw(tu, sort.merged_gs_index[a:b], sort.merged_map_id[a:b], sort.merged_lod_id[a:b]))
        else:
            draws.append(orc.Draw(tu, gi[d.base_lod][d.base_tile][d.base_view], None, li[d.base_lod][d.base_tile][d.base_view]))
    ocu = orc.Camera176.from_buffer_copy(bytes(cu))
    osu = orc.Scene160.from_buffer_copy(bytes(su))
    # the GPU box gives one GPU's job a 16-core CPU share; the oracle's OpenMP team is sized to that
    n_threads = int(os.environ.get("GSWT_CPU_BASELINE_THREADS", str(min(16, os.cpu_count() or 1))))
    t0 = time.perf_counter()
    bg = bgd = None
    if passes is not None:          # the oracle's skybox + proxy restatements feed the splat pass (state.rs:384-402)
        faces, mips, pu = passes
        cam = type("Cam", (), {})()
        cam.view = np.array(ocu.view[:], dtype=np.float32); cam.projection = np.array(ocu.projection[:], dtype=np.float32)
        bg = orc.skybox_render(cam, faces, W, H)
        bgd = np.ones((H, W), np.float32)
        orc.proxy_render(orc.Proxy224.from_buffer_copy(bytes(pu)), W, H, bg, bgd, mips)
    img, st = orc.render(ocu, osu, tex, draws, W, H, n_threads=n_threads, bg_rgba=bg, bg_depth=bgd, height_map=height_map)
    dt = time.perf_counter() - t0
    return img, st, dt, n_threads


def make_passes(r, su, cu):
    """Synthetic inputs of the skybox and proxy passes (BASELINE config 5): a sky cube that is smooth in direction, a checker
    proxy texture with its mip chain, a `proxy_map` grid at z = -0.5.  Configures them on renderer `r`; returns
    (faces, mips, proxy uniforms) so that the oracle can be fed the same."""
    from gswt_renderer_amd import _lib as L
    # synthetic sky cube (smooth in direction) and proxy texture (checker mip chain); proxy_map grid at z = -0.5
    n = 256
    faces = np.zeros((6, n, n, 4), np.float32)
    tt, ss = np.meshgrid((np.arange(n) + .5) / n * 2 - 1, (np.arange(n) + .5) / n * 2 - 1, indexing="ij")
    one = np.ones_like(ss)
    dirs = {0: (one, -tt, -ss), 1: (-one, -tt, ss), 2: (ss, one, tt), 3: (ss, -one, -tt), 4: (ss, -tt, one), 5: (-ss, -tt, -one)}
    for fi in range(6):
        dv = np.stack(dirs[fi], -1)
        dv /= np.linalg.norm(dv, axis=-1, keepdims=True)
        faces[fi, ..., :3] = 0.5 + 0.4 * dv
        faces[fi, ..., 3] = 1.0
    ts = 512
    yy, xx = np.mgrid[0:ts, 0:ts]
    cur = np.zeros((ts, ts, 4), np.float32)
    cur[..., 0] = ((xx // 32 + yy // 32) % 2) * 0.6 + 0.2
    cur[..., 1] = 0.35; cur[..., 2] = 0.25; cur[..., 3] = 1.0
    mips = []
    while True:
        mips.append(cur.copy())
        if cur.shape[0] == 1:
            break
        cur = cur.reshape(cur.shape[0] // 2, 2, cur.shape[1] // 2, 2, 4).mean((1, 3)).astype(np.float32)
    r.skybox_configure(faces)
    r.proxy_configure(mips)
    pu = L.ProxyUniforms()
    pu.height_offset, pu.tile_width, pu.surface_type, pu.width_scale = -0.5, su.tile_width, 0, 4.0
    pu.map_proxy, pu.use_clip, pu.clip_height, pu.brightness, pu.black_background = 1, 0, 0.0, 1.0, 0
    pu.view[:] = cu.view[:]; pu.projection[:] = cu.projection[:]
    pu.map_half_wh[:] = su.map_half_wh[:]; pu.center_coord[:] = su.center_coord[:]
    pu.height_map_scale[:] = su.height_map_scale[:]; pu.cam_pos[:] = cu.cam_pos[:]
    return faces, mips, pu


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough for the two-frame pipeline and the clocks to settle (50 steps after 5 of warm-up read 4-5 % low)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--lod0", type=int, default=0, help="override LOD0 splats per tile (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--t-eps", type=float, default=1e-5, help="front-to-back early-out threshold")
    ap.add_argument("--passes", type=int, default=-1, help="1: run the skybox + proxy compute passes before the splats each frame "
                    "(BASELINE config 5); default: on for c5, off otherwise")
    ap.add_argument("--timing", type=int, default=1, help="hipEvent level: 1 = frame + k_composite (roofline), 2 = every stage")
    ap.add_argument("--in-flight", type=int, default=0, help="frames in flight (default: every frame slot of the library, two when a "
                    "slot's buffers exceed 2 GB)")
    ap.add_argument("--timing-every", type=int, default=8, help="frames between timed ones: the events around k_composite are recorded on "
                    "every N-th frame of the timed region (recording them on every frame costs ~4 %% of the frame rate)")
    args = ap.parse_args()

    # Only the final JSON line may reach stdout (RCCL prints a version banner there): park the real stdout and point
    # fd 1 at stderr for the rest of the run.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    fake_world = int(os.environ.get("GSWT_BENCH_FAKE_WORLD", "0"))     # test hook: rank 0 of an N-rank run without the other ranks
    force_dist = os.environ.get("GSWT_BENCH_FORCE_DIST") == "1"      # exercise the all-gather plumbing with one rank
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    import torch
    dist = None
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    elif n_gpus > 1:
        raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from gswt_renderer_amd.renderer import GSWTRenderer
    w, wang, cu, vp, sort = build_workload(args.workload, args.lod0 or None)
    W, H = w["width"], w["height"]
    r = GSWTRenderer(local_rank)                       # raises when libgswt_hip.so / the GPU is missing
    stream = torch.cuda.Stream(device=dev)
    r.set_stream(stream.cuda_stream)                    # ctx stream: fences, all-gather and unshard are ordered on it
    from gswt_renderer_amd import _lib as L
    r.set_option(L.GSWT_OPT_TIMING, args.timing)
    wang.upload_to(r)
    hmap = wang.height_map() if int(wang.user.surface_type) == 1 else None
    r.configure(hmap)
    r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
    su = wang.scene_uniforms()
    use_dist = world > 1 or force_dist
    if fake_world > 1 and world == 1:
        world, use_dist = fake_world, True
    shard = (rank, world, "cols") if world > 1 else (0, 1)
    rows = H
    band_w = r.shard_cols_padded(W, world) if world > 1 else W
    # as many frames in flight as the library has frame slots (gswt_render_async / gswt_render_wait): the next frames are
    # queued while the oldest executes; each frame in flight has its own output buffer
    slots = r.frame_slots()
    outs = [torch.empty((rows, band_w, 4), dtype=torch.float32, device=dev) for _ in range(slots)]
    out = outs[0]
    # ... unless a slot's per-frame buffers are large: rotating three multi-GB buffer sets costs more than the third frame in
    # flight gains (c5, ~5.6 GB per slot: 554 frames/s with two in flight, 543 with three).  The library reuses the lowest free
    # slot, so keeping fewer frames in flight also keeps fewer buffer sets in rotation.
    r.render_wait(r.render_async(cu, su, W, H, outs[0].data_ptr(), transmittance_eps=args.t_eps, shard=shard))
    per_slot_bytes = 60.0 * float(r.timings()["n_instanced"])           # rects + records per list entry, roughly
    if args.in_flight > 0:
        slots = max(1, min(slots, args.in_flight))
    elif per_slot_bytes > 2e9:
        slots = min(slots, 2)
    gathered = torch.empty((world * rows, band_w, 4), dtype=torch.float32, device=dev) if use_dist else None
    frame = torch.empty((H, W, 4), dtype=torch.float32, device=dev) if use_dist else None

    use_passes = args.passes == 1 or (args.passes < 0 and args.workload == "c5")
    bgs = depths = None
    pu = None
    if use_passes:
        faces, mips, pu = make_passes(r, su, cu)
        bgs = [torch.empty((H, W, 4), dtype=torch.float32, device=dev) for _ in range(slots)]
        depths = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(slots)]
        torch.cuda.synchronize()

    comp_ms, total_ms, pairs = [], [], []
    inflight = []
    last = [None]

    def submit(i):
        # the frame runs on its slot's own stream (the frames in flight overlap on the GPU); the all-gather of frame i is
        # queued on the ctx stream behind a device-side fence, AFTER frame i+1 has been submitted
        o = outs[i % slots]
        bgp = dpp = 0
        if use_passes:      # state.rs:384-392: skybox, then proxy (colour + depth), then the splats over them
            bgp, dpp = bgs[i % slots].data_ptr(), depths[i % slots].data_ptr()
            r.skybox_render(cu, W, H, bgp)
            r.proxy_render(pu, W, H, bgp, dpp, True)
        timed = args.timing > 0 and i % max(1, args.timing_every) == 0
        r.set_option(L.GSWT_OPT_TIMING, args.timing if timed else 0)      # per-frame: the slot remembers its own level
        ticket = r.render_async(cu, su, W, H, o.data_ptr(), transmittance_eps=args.t_eps, shard=shard, bg_rgba_ptr=bgp, bg_depth_ptr=dpp)
        inflight.append((ticket, o, timed))

    def collect():
        ticket, o, timed = inflight.pop(0)
        if use_dist:
            r.render_fence(ticket)
            with torch.cuda.stream(stream):
                if fake_world > 1:
                    gathered[:rows].copy_(o, non_blocking=True)      # stands in for the collective (own shard only)
                else:
                    dist.all_gather_into_tensor(gathered, o)
                r.unshard_mode(gathered.data_ptr(), W, H, world, "cols", frame.data_ptr())
        r.render_wait(ticket)
        t = r.timings()
        pairs.append(t["n_pairs"])
        if timed:
            comp_ms.append(t["ms_composite_kernel"]); total_ms.append(t["ms_total"])
            last[0] = t
        elif last[0] is None:
            last[0] = t

    def run(n):
        for i in range(n):
            submit(i)
            if len(inflight) == slots:
                collect()
        while inflight:
            collect()
        stream.synchronize()

    run(args.warmup)
    comp_ms.clear(); total_ms.clear(); pairs.clear()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    last = last[0]
    out = outs[(args.steps - 1) % slots]
    if dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # A few frames one at a time (outside the timed region): the compositing kernel without another frame's kernels
    # sharing the chip.  `roofline` itself comes from the timed region, where two frames overlap.
    iso = []
    r.set_option(L.GSWT_OPT_TIMING, max(1, args.timing))
    for i in range(6):
        r.render_wait(r.render_async(cu, su, W, H, outs[0].data_ptr(), transmittance_eps=args.t_eps, shard=shard,
                                     bg_rgba_ptr=bgs[0].data_ptr() if use_passes else 0, bg_depth_ptr=depths[0].data_ptr() if use_passes else 0))
        iso.append(r.timings()["ms_composite_kernel"])
    iso_ms = float(np.median(iso[2:]))
    if dist:
        dist.barrier()

    if rank == 0:
        fps = args.steps / dt
        P = float(np.mean(pairs))
        comp = float(np.mean(comp_ms)) * 1e-3
        n_px = rows * band_w
        algo_bytes = 52.0 * P + (36.0 if use_passes else 16.0) * n_px   # (4 + 48) B per pair + 16 B per pixel (+ 20 B read with bg colour + depth)
        achieved = algo_bytes / comp / 1e9 if comp > 0 else 0.0
        # HBM bytes per launch of k_composite from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE, separate passes; profiles/): raw counter bytes.  MI355X_MICROARCH's x2 rule is for wide coalesced
        # streams; these reads are 48-B gathers (uncalibrated), so the raw sum is reported and the x2 figure kept beside it.
        traffic, traffic_note = None, "no PMC summary for this workload under profiles/"
        valu_issue = None
        pmc_path = os.path.join(ROOT, "profiles", f"r01_pmc_{args.workload}.json")
        if world == 1 and os.path.exists(pmc_path):
            try:
                pmc = json.load(open(pmc_path))
                k = next(v for n, v in pmc.items() if "k_composite" in n)
                traffic = (k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
                if "SQ_INSTS_VALU" in k and iso_ms > 0:
                    # what actually bounds the kernel: a stream of dependent VALU instructions issues at ~4.8 cycles per
                    # wave-instruction per SIMD whatever the occupancy (tools/ubench/valu_issue.hip; 2 with ILP)
                    us = k["SQ_INSTS_VALU"] * 4.8 / (256 * 4 * 2.4e3)
                    valu_issue = {"wave_insts_per_launch": k["SQ_INSTS_VALU"], "cycles_per_wave_inst": 4.8, "simds": 1024, "clock_ghz": 2.4,
                                  "us_at_that_rate": us, "ratio_to_isolated_kernel": us / (iso_ms * 1e3),
                                  "note": "dependent-chain VALU issue rate measured by tools/ubench/valu_issue.hip; instruction count from the committed PMC pass"}
                traffic_note = (f"{os.path.basename(pmc_path)}: FETCH_SIZE {k['FETCH_SIZE'] / 1024:.1f} MiB + WRITE_SIZE {k['WRITE_SIZE'] / 1024:.1f} MiB raw per launch; "
                                f"with the gfx950 x2 wide-read rule the reads would be {2 * k['FETCH_SIZE'] / 1024:.1f} MiB (48-B gathers: uncalibrated)")
            except Exception as e:      # the summary is evidence, not a dependency
                traffic_note = f"could not read {pmc_path}: {e}"
        res = {
            "metric": "frames/sec @1920x1080, 32x32 Wang-tile grid" if args.workload == "c3" else f"frames/sec ({args.workload})",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {w['desc']}", "map": [2 * w["half"][0] + 1, 2 * w["half"][1] + 1],
                       "width": W, "height": H, "n_draws": int(last["n_draws"]), "n_instanced": int(last["n_instanced"]),
                       "n_visible": int(last["n_visible"]), "n_pairs": int(last["n_pairs"]), "order": "reference",
                       "transmittance_eps": args.t_eps, "skybox_proxy_passes": bool(use_passes),
                       "parallelism": f"screen-tile-column bands x{world} (projection culled per band) + RCCL all-gather" if world > 1 else "single GPU"},
            "stage_ms": {k: float(last[k]) for k in ("ms_project", "ms_emit", "ms_sort", "ms_ranges", "ms_composite", "ms_composite_kernel", "ms_total")},
            "frames_in_flight": slots,
            "roofline": {"bound": "hbm", "kernel": "k_composite", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms": comp * 1e3, "kernel_ms_samples": len(comp_ms), "timed_every": max(1, args.timing_every),
                         "kernel_ms_isolated": iso_ms, "frac_isolated": (algo_bytes / (iso_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if iso_ms > 0 else None,
                         "traffic_note": traffic_note, "valu_issue": valu_issue},
        }
        if use_dist:
            res["dist_check_max_abs_diff"] = float((frame - (out if world == 1 else frame)).abs().max().item())
        if world == 1 and not args.no_cpu_baseline:
            img_cpu, st, cdt, nthr = cpu_baseline(wang, sort, cu, vp, su, W, H, passes=(faces, mips, pu) if use_passes else None, height_map=hmap)
            gpu_img = out.cpu().numpy()
            res["cpu_baseline"] = {"value": 1.0 / cdt, "unit": "frames/s", "cores": nthr, "kind": "port",
                                   "sample": "1 frame of the same workload (oracle/gswt_oracle.c, OpenMP over 16-row bands)",
                                   "max_abs_diff_vs_gpu": float(np.max(np.abs(gpu_img.astype(np.float64) - img_cpu.astype(np.float64))))}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
