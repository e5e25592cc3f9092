#!/usr/bin/env python3
"""bench.py -- frames/sec of the GSWT hot path on MI355X (BASELINE.json metric).

A "step" is one frame of the hot path with every input already in HBM: by default the camera follows a fixed 240-frame
Catmull-Rom fly path (gswt_renderer_amd/flypaths/, the reference's benchmark mechanism: control.rs:383-527,
gui.rs:964-997) while the Wang-tile worker (check_update / build_tiles / sort_tiles, state.rs:478-561) runs on a second
host thread exactly as in the reference; every SortData it delivers is swapped in (gswt_set_draws_merge_groups: draw list
upload + merged-group lists rebuilt on the device) inside the timed region.  `--mode static` renders one camera over a
resident draw list instead (round 1's number; also reported beside the fly-path value as `static_camera`).
Each frame: Wang-tile instancing -> projection -> pair emit -> tile sort -> compositing.
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
import math
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
    verts = synth.make_tileset(n_lod=w["n_lod"], n_tile=16, lod0_count=w["lod0"], base_scale=w.get("base_scale", 0.02))
    ts = host.TileSet.from_vertices(verts)
    wang = host.WangTile(ts)
    user = host.user_data(tile_map_half_wh=w["half"], **w["user"])
    wang.configure(user)
    cam = workloads.camera_for(name)
    cu, vp = host.camera_uniforms(cam["pos"], cam["target"], cam["up"], cam["fovy"], cam["near"], cam["far"], w["width"], w["height"])
    wang.build_tiles(cam["pos"])
    sort = wang.sort_tiles(cam["pos"], vp)
    return w, wang, cu, vp, sort


def auto_segment(pairs_per_tile: float, pairs_frame: float) -> int:
    """GSWT_OPT_SEGMENT (pairs per compositor work item) for a frame with this many pairs per screen tile / in all: long for dense frames
    (a segment cannot skip what the segments in front of it saturated), short enough that the frame has ~2 048 work items."""
    from gswt_renderer_amd import _lib as L
    seg_dense = max(L.GSWT_DEFAULT_SEGMENT, 256 * math.ceil(4.0 * pairs_per_tile / 256.0))
    seg_fill = 256 * max(1, math.ceil(pairs_frame / 2048.0 / 256.0))
    return int(min(4096, seg_dense, seg_fill))


def oracle_draws(wang, sort, vp, culling_dist=1.0):
    """The draw list of a sort event as the CPU oracle takes it (tex, [orc.Draw]): the CPU viewport cull of renderer.rs:472-494
    applied on the host (the HIP path does it on the device).  Checker-side only (cpu_baseline leg and tests)."""
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
    return tex, draws


def cpu_baseline(wang, sort, cu, vp, su, W, H, culling_dist=1.0, passes=None, height_map=None, order_mode=0, v2=False):
    """Times the CPU oracle (oracle/gswt_oracle.c, OpenMP) on ONE frame of the same workload."""
    from oracle import gswt_oracle as orc
    tex, draws = oracle_draws(wang, sort, vp, culling_dist)
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
    with (orc.v2() if v2 else orc.strict(fragment=False)):      # the same vertex-stage sequence as the GPU frame it is compared with
        img, st = orc.render(ocu, osu, tex, draws, W, H, n_threads=n_threads, bg_rgba=bg, bg_depth=bgd, height_map=height_map, order_mode=order_mode)
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


class Worker:
    """The reference's worker thread (state.rs:478-561): takes the newest camera, rebuilds the tile map when the camera moved
    far enough (check_update / build_tiles), re-sorts when the view-projection changed by >= 0.01 (sum of |d VP|), and hands
    the newest SortData (+ the scene uniforms that belong to it) back.  libgswt_host calls release the GIL."""

    def __init__(self, wang, dev=None):
        import threading
        self.wang = wang
        self.dev = dev          # DeviceWorker: update_lod / merging / order / views / records on the GPU (--device-worker)
        self._dev_built = False
        self._lock = threading.Lock()
        self._wake = threading.Event()
        self._req = None
        self._res = None
        self._fifo = []
        self._done = {}
        self._stop = False
        self._prev_vp = None
        self.build_ms, self.sort_ms = [], []
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def submit(self, pos, vp):
        with self._lock:
            self._req = (pos, vp)
        self._wake.set()

    # -- lock-step mode (N > 1): every rank must swap the SAME SortData in at the SAME frame, or the gathered bands come from
    # different draw lists (and tile maps: spawning consumes the worker's RNG in the order of its build events).  The ranks
    # therefore hand the worker every `every`-th camera IN ORDER and apply its result a fixed number of frames later, waiting
    # for it if need be -- deterministic on every rank, unlike "newest camera wins".
    def submit_ordered(self, tag, pos, vp):
        with self._lock:
            self._fifo.append((tag, pos, vp))
        self._wake.set()

    def take(self, tag, timeout=30.0):
        t_end = time.perf_counter() + timeout
        while True:
            with self._lock:
                if tag in self._done:
                    return self._done.pop(tag)
            if time.perf_counter() > t_end:
                raise RuntimeError(f"worker result {tag} did not arrive")
            time.sleep(2e-5)

    def poll(self):
        with self._lock:
            r, self._res = self._res, None
        return r

    def close(self):
        self._stop = True
        self._wake.set()
        self._t.join(timeout=10)

    def step(self, pos, vp):
        """One worker iteration, synchronously (initial frame)."""
        rebuilt = False
        if self.wang.check_update(pos) or (self.dev is not None and not self._dev_built):
            self._dev_built = True                    # the device worker's first map comes with its first step
            t0 = time.perf_counter()
            self.wang.build_tiles(pos)
            if self.dev is not None:
                self.dev.build_tiles(pos)             # map upload + update_lod on the device
            self.build_ms.append(1e3 * (time.perf_counter() - t0))
            rebuilt = True
        moved = self._prev_vp is None or float(np.abs(vp - self._prev_vp).sum()) >= 0.01       # state.rs:527-548
        if not (rebuilt or moved or bool(self.wang.user.always_sort)):
            return None
        self._prev_vp = vp.copy()
        t0 = time.perf_counter()
        if self.dev is not None:
            self.dev.sort_tiles(pos, vp)
            self.dev.fetch()                          # waits for the worker's stream, publishes the event for the render thread
            raw = None
        else:
            raw = self.wang.sort_tiles_raw(pos, vp)
        self.sort_ms.append(1e3 * (time.perf_counter() - t0))
        return raw, self.wang.scene_uniforms()

    def _run(self):
        while True:
            self._wake.wait()
            self._wake.clear()
            if self._stop:
                return
            while True:
                with self._lock:
                    job = self._fifo.pop(0) if self._fifo else None
                if job is None:
                    break
                res = self.step(job[1], job[2])
                with self._lock:
                    self._done[job[0]] = res
            with self._lock:
                req, self._req = self._req, None
            if req is None:
                continue
            res = self.step(*req)
            if res is not None:
                with self._lock:
                    self._res = res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: long enough for the frame pipeline and the clocks to settle (50 steps after 5 of warm-up read 4-5 % low)
    ap.add_argument("--steps", type=int, default=480)
    ap.add_argument("--warmup", type=int, default=48)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--mode", default="flypath", choices=["flypath", "static"], help="flypath: 240-frame fly path + worker thread + "
                    "sort-event swap-ins in the timed region (the metric); static: one camera, resident draw list")
    ap.add_argument("--path", default="", help="fly path: a name under gswt_renderer_amd/flypaths/ or a JSON file in the reference's "
                    "schema (default: the workload's own path, else c3)")
    ap.add_argument("--path-frames", type=int, default=240)
    ap.add_argument("--lod0", type=int, default=0, help="override LOD0 splats per tile (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--t-eps", type=float, default=1e-5, help="front-to-back early-out threshold")
    ap.add_argument("--passes", type=int, default=-1, help="1: run the skybox + proxy compute passes before the splats each frame "
                    "(BASELINE config 5); default: on for c5, off otherwise")
    ap.add_argument("--timing", type=int, default=2, help="hipEvent level on the timed frames: 1 = frame + k_composite (roofline), 2 = every stage")
    ap.add_argument("--in-flight", type=int, default=0, help="frames in flight (default: every frame slot of the library, two when a "
                    "slot's buffers exceed 2 GB)")
    ap.add_argument("--timing-every", type=int, default=0, help="frames between timed ones: the events of a frame are recorded on "
                    "every N-th frame of the timed region (on every frame they cost ~3 %% of the frame rate)")
    ap.add_argument("--static-at", type=int, default=-1, help="static mode: use fly-path camera number K instead of the workload's own")
    ap.add_argument("--freeze-sort", action="store_true", help="fly path cameras without the worker: the first SortData stays (\"Lock (Sort)\" of the "
                    "reference GUI, gui.rs:599-605); separates the per-view workload from the cost of the sort events")
    ap.add_argument("--no-defer-swap", action="store_true", help="fly path: every swap-in is current at once (the next frame waits for its merged-list build on the device)")
    ap.add_argument("--device-worker", action="store_true", help="fly path: run the per-sort-event worker stages on the GPU (gswt_worker_*) instead of libgswt_host")
    ap.add_argument("--graph", action="store_true", help="GSWT_OPT_GRAPH: replay each frame's launch sequence as one hipGraphLaunch (frames that carry timing events still launch kernel by kernel); default with --gpus N > 1, where a rank's frame is short enough for the submitting thread to matter")
    ap.add_argument("--no-graph", action="store_true", help="never use GSWT_OPT_GRAPH")
    ap.add_argument("--segment", type=int, default=0, help="GSWT_OPT_SEGMENT, pairs per compositor work item (multiple of 256); 0 = from the first frame's pairs per screen tile: 4 x that, between the library's default and 4096 (dense scenes with the early-out on gain from long segments)")
    ap.add_argument("--static-steps", type=int, default=100, help="flypath mode: frames of the static-camera comparison run (0: skip)")
    ap.add_argument("--order", default="reference", choices=["reference", "depth"], help="reference: tiles back to front, presorted lists (wangtile.rs:489-499, "
                    "scene.rs:685-695); depth: GSWT_ORDER_DEPTH, every visible splat of the frame in true depth order (the global radix depth sort)")
    ap.add_argument("--vertex-stage", default="strict", choices=["strict", "v2"], help="strict (default, GSWT_OPT_STRICT_VS = 1): vs_main operator by operator as "
                    "gswt.wgsl:152-258 writes it; v2: the fma-chain / single-reciprocal rounding sequence (the default until round 3)")
    ap.add_argument("--composite", type=int, default=-1, help="GSWT_OPT_COMPOSITE: 0 k_composite + k_combine, 1 k_composite_dw (decoupled waves) + k_combine, "
                    "2 k_composite<FOLD> (no k_combine launch); default: the library's")
    ap.add_argument("--depth-sort", type=int, default=0, help="GSWT_OPT_DEPTH_SORT with --order depth: 0 auto, 1 global depth passes, 2 tile-local LDS sort")
    ap.add_argument("--item-order", type=int, default=-1, help="GSWT_OPT_ITEM_ORDER: 0 the compositor's work items in tile order, 1 heaviest first")
    ap.add_argument("--no-chunk-cull", action="store_true", help="GSWT_OPT_NO_CHUNK_CULL: project every chunk of the draws that survive the tile cull (A/B of the per-chunk frustum cull)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0, help="CPU baseline: fly-path frames are rendered by the oracle until this much time has gone (at most 24 frames)")
    args = ap.parse_args()
    if args.timing_every <= 0:
        # 0 = auto.  Every third frame of a long run carries hipEvents (3 is coprime with the five frame slots: every slot is sampled).  A short
        # fly-path run on one GPU (the driver's --steps 20) carries NONE inside its timed region: ten samples, two per slot, were never a
        # measurement, the frames that carried them cost the submitting thread 52 us instead of 34 (4 190-4 350 against
        # 4 500-4 570 frames/s, three runs each), and the roofline figures come from the 480-frame steady loop timed right behind the region
        # (roofline.kernel_ms_source).  Other short runs (static camera, N > 1: no steady loop follows) time every other frame.
        short_with_steady = args.steps < 64 and args.mode == "flypath" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.freeze_sort \
            and int(os.environ.get("GSWT_BENCH_FAKE_WORLD", "0")) <= 1
        args.timing_every = (1 << 30) if short_with_steady else (2 if args.steps < 64 else 3)

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

    from gswt_renderer_amd import flypath, host
    from gswt_renderer_amd import _lib as L
    from gswt_renderer_amd.renderer import GSWTRenderer
    w, wang, cu0, vp0, sort0 = build_workload(args.workload, args.lod0 or None)
    W, H = w["width"], w["height"]
    from gswt_renderer_amd import workloads
    cam0 = workloads.camera_for(args.workload)
    if world > 1 or fake_world > 1 or args.device_worker:
        # stream creation order of the library (hardware-queue sharing): the layout measured best for a rank's band frames + gather
        # (and for the device-side worker, whose kernels run on a stream of their own: 3 950 against 3 250 frames/s)
        os.environ.setdefault("GSWT_STREAM_LAYOUT", "c012p3s")
    r = GSWTRenderer(local_rank)                       # raises when libgswt_hip.so / the GPU is missing
    stream = torch.cuda.Stream(device=dev)
    r.set_stream(stream.cuda_stream)                    # ctx stream: fences, all-gather and unshard are ordered on it
    r.set_option(L.GSWT_OPT_TIMING, args.timing)
    r.set_option(L.GSWT_OPT_STRICT_VS, 1 if args.vertex_stage == "strict" else 0)
    if args.composite >= 0:
        r.set_option(L.GSWT_OPT_COMPOSITE, args.composite)
    if args.depth_sort:
        r.set_option(L.GSWT_OPT_DEPTH_SORT, args.depth_sort)
    if args.no_chunk_cull:
        r.set_option(L.GSWT_OPT_NO_CHUNK_CULL, 1)
    if args.item_order >= 0:
        r.set_option(L.GSWT_OPT_ITEM_ORDER, args.item_order)
    order_mode = L.GSWT_ORDER_DEPTH if args.order == "depth" else L.GSWT_ORDER_REFERENCE
    wang.upload_to(r)
    hmap = wang.height_map() if int(wang.user.surface_type) == 1 else None
    r.configure(hmap)
    r.set_draws(sort0.draws, sort0.merged_gs_index, sort0.merged_map_id, sort0.merged_lod_id)
    su0 = wang.scene_uniforms()
    use_dist = world > 1 or force_dist
    # The collective lives behind the C ABI (gswt_comm_init / gswt_render_gather: RCCL all-gather + re-assembly on the ctx
    # stream); torch.distributed only ships the 128-byte RCCL id.  GSWT_BENCH_TORCH_GATHER=1 (or a failed communicator
    # set-up) falls back to torch.distributed.all_gather_into_tensor + gswt_unshard_mode.
    abi_comm = False
    if use_dist and dist is not None and os.environ.get("GSWT_BENCH_TORCH_GATHER") != "1":
        box = [None]
        if rank == 0:
            try:
                box[0] = GSWTRenderer.comm_unique_id()
            except Exception as e:      # noqa: BLE001 - every rank learns it through the broadcast below and falls back together
                print(f"[bench] RCCL id unavailable ({e}); using torch.distributed for the gather", file=sys.stderr)
        dist.broadcast_object_list(box, src=0)
        if box[0] is not None:
            ok = torch.ones(1, dtype=torch.int32, device=dev)
            try:
                r.comm_init(box[0], rank, world)
            except Exception as e:      # noqa: BLE001
                ok.zero_()
                print(f"[bench] rank {rank}: gswt_comm_init failed ({e})", file=sys.stderr)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)           # all ranks use the ABI's gather, or none does
            abi_comm = bool(ok.item())
            if not abi_comm:
                try:
                    r.comm_destroy()
                except Exception:       # noqa: BLE001
                    pass
    if fake_world > 1 and world == 1:
        world, use_dist = fake_world, True
    shard = (rank, world, "cols") if world > 1 else (0, 1)
    rows = H
    band_w = r.shard_cols_padded(W, world) if world > 1 else W
    # as many frames in flight as the library has frame slots (gswt_render_async / gswt_render_wait): the next frames are
    # queued while the oldest executes; each frame in flight has its own output buffer
    slots = r.frame_slots()
    outs = [torch.empty((rows, band_w, 4), dtype=torch.float32, device=dev) for _ in range(slots)]
    # ... unless a slot's per-frame buffers are large: rotating three multi-GB buffer sets costs more than the third frame in
    # flight gains (c5, ~5.6 GB per slot: 554 frames/s with two in flight, 543 with three).  The library reuses the lowest free
    # slot, so keeping fewer frames in flight also keeps fewer buffer sets in rotation.
    r.render_wait(r.render_async(cu0, su0, W, H, outs[0].data_ptr(), transmittance_eps=args.t_eps, order_mode=order_mode, shard=shard))
    per_slot_bytes = 60.0 * float(r.timings()["n_instanced"])           # rects + records per list entry, roughly
    # compositor work-item size: a segment of a tile's list cannot skip what the segments in front of it already saturated, so a dense
    # frame (many pairs per screen tile) wants long segments; a sparse one is indifferent up to ~2 k (gswt_api.hip, opt_segment)
    t0f = r.timings()
    pairs_per_tile = float(t0f["n_pairs"]) / max(1.0, float(t0f["n_tiles"]))
    pairs_frame = float(t0f["n_pairs"])
    if world > 1 and dist is not None and fake_world <= 1:
        ppt = torch.tensor([pairs_per_tile, pairs_frame], dtype=torch.float64, device=dev)
        dist.all_reduce(ppt, op=dist.ReduceOp.MAX)             # every rank composites with the same segment length
        pairs_per_tile, pairs_frame = float(ppt[0].item()), float(ppt[1].item())
    # ... and a frame with few pairs (one rank's band of a sharded frame) wants SHORT ones, or its work items do not fill the chip: at least
    # ~2 048 of them (the compositor's resident workgroups).  c3: 1 536 either way; one of eight column bands (345 k pairs): 256 instead of
    # 1 536 -- k_composite 30 instead of 66 us one frame at a time, 61-75 instead of 73-90 us per frame with five in flight
    # (tools/shard_emulation.py, GSWT_SEGMENT; profiles/r04_shard_segment_c3.txt)
    segment = args.segment if args.segment > 0 else auto_segment(pairs_per_tile, pairs_frame)
    r.set_option(L.GSWT_OPT_SEGMENT, segment)
    # three frames in flight on a static camera (a fourth costs 6 %: four buffer sets in rotation), four on the fly path (the
    # fourth covers the bubble a SortData swap-in leaves in the frame stream: +5 %)
    slots_static = min(slots, 3)
    if world > 1 or args.device_worker:
        slots = min(slots, 4)           # a rank's band frames: the fifth frame in flight gains nothing there (rank 0 of 8: 7.8-8.9 k frames/s with four or five)
    if args.in_flight > 0:
        slots = slots_static = max(1, min(slots, args.in_flight))
    elif per_slot_bytes > 2e9:
        slots = slots_static = min(slots, 2)
    elif args.mode == "static":
        slots = slots_static
    gathered = torch.empty((world * rows, band_w, 4), dtype=torch.float32, device=dev) if use_dist else None
    frame = torch.empty((H, W, 4), dtype=torch.float32, device=dev) if use_dist else None

    use_passes = args.passes == 1 or (args.passes < 0 and args.workload == "c5")
    bgs = depths = None
    pu = None
    if use_passes:
        faces, mips, pu = make_passes(r, su0, cu0)
        bgs = [torch.empty((H, W, 4), dtype=torch.float32, device=dev) for _ in range(slots)]
        depths = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(slots)]
        torch.cuda.synchronize()

    # ---- cameras of the run -------------------------------------------------------------------------------------------
    path_name = args.path or (args.workload if os.path.exists(os.path.join(ROOT, "gswt_renderer_amd", "flypaths", args.workload + ".json")) else "c3")
    if args.mode == "flypath":
        cams = []
        for pos, tgt in flypath.sample(flypath.load(path_name), args.path_frames):
            cams.append((tuple(float(x) for x in pos),) + host.camera_uniforms(pos, tgt, cam0["up"], cam0["fovy"], cam0["near"], cam0["far"], W, H))
        # the fly path uses the device-side merged-list build: a sort event uploads O(#tiles), not 12 B per merged splat
        wang.upload_raw_depth_to(r)
        wang.set_device_merge(True)
    elif args.static_at >= 0:
        pos, tgt = flypath.sample(flypath.load(path_name), args.path_frames)[args.static_at % args.path_frames]
        cu_k, vp_k = host.camera_uniforms(pos, tgt, cam0["up"], cam0["fovy"], cam0["near"], cam0["far"], W, H)
        cams = [(tuple(float(x) for x in pos), cu_k, vp_k)]
        wk = Worker(wang); res_k = wk.step(cams[0][0], np.asarray(vp_k, dtype=np.float32)); wk.close()
        wang.upload_raw_depth_to(r); wang.set_device_merge(True)
        res_k = (wang.sort_tiles_raw(cams[0][0], vp_k), wang.scene_uniforms())
    else:
        cams = [(tuple(cam0["pos"]), cu0, vp0)]

    trace = [] if os.environ.get("GSWT_BENCH_TRACE") else None      # host-side timeline of submits and waits (stderr, after the run)
    state = {"su": su0, "swaps": 0, "swap_ms": [], "slots": slots, "every": 4}
    stats = {"comp_ms": [], "comp_slot": [], "pairs": [], "stage": [], "submit_ms": []}
    inflight = []

    def swap_in(res):
        raw, su = res
        t0 = time.perf_counter()
        if raw is None:
            dev_worker.swap_in(fetch=False)                                      # the event the device-side worker published last
        else:
            draws, nd, groups, ng, members, nm = raw
            r.set_draws_merge_groups_raw(draws, nd, groups, ng, members, nm)     # SortData swap-in, state.rs:361-376
        state["swap_ms"].append(1e3 * (time.perf_counter() - t0))
        if trace is not None:
            trace.append(("swap", state["swaps"] + 1, t0, time.perf_counter(), state["swaps"] + 1))
        state["su"] = su
        state["swaps"] += 1

    def submit(i, cu, timed):
        # the frame runs on its slot's own stream (the frames in flight overlap on the GPU); the all-gather of frame i is
        # queued on the ctx stream behind a fence, AFTER frame i+1 has been submitted
        slots = state["slots"]
        o = outs[i % slots]
        bgp = dpp = 0
        t0 = time.perf_counter()
        if use_passes:      # state.rs:384-392: skybox, then proxy (colour + depth), then the splats over them
            bgp, dpp = bgs[i % slots].data_ptr(), depths[i % slots].data_ptr()
            pu.view[:] = cu.view[:]; pu.projection[:] = cu.projection[:]; pu.cam_pos[:] = cu.cam_pos[:]
            pu.center_coord[:] = state["su"].center_coord[:]
            r.skybox_render(cu, W, H, bgp)
            r.proxy_render(pu, W, H, bgp, dpp, True)
        r.set_option(L.GSWT_OPT_TIMING, args.timing if timed else 0)      # per frame: the slot remembers its own level
        ticket = r.render_async(cu, state["su"], W, H, o.data_ptr(), transmittance_eps=args.t_eps, order_mode=order_mode, shard=shard, bg_rgba_ptr=bgp, bg_depth_ptr=dpp)
        stats["submit_ms"].append(1e3 * (time.perf_counter() - t0))
        if trace is not None:
            trace.append(("submit", i, t0, time.perf_counter(), state["swaps"]))
        inflight.append((ticket, o, timed))
        state["ticket"] = ticket

    def collect():
        ticket, o, timed = inflight.pop(0)
        if use_dist and abi_comm:
            r.render_gather(ticket, frame.data_ptr())
        elif use_dist:
            r.render_fence(ticket)
            with torch.cuda.stream(stream):
                if fake_world > 1:
                    gathered[:rows].copy_(o, non_blocking=True)      # stands in for the collective (own shard only)
                else:
                    dist.all_gather_into_tensor(gathered, o)
                r.unshard_mode(gathered.data_ptr(), W, H, world, "cols", frame.data_ptr())
        tw0 = time.perf_counter()
        r.render_wait(ticket)
        if trace is not None:
            trace.append(("wait", int(ticket), tw0, time.perf_counter(), state["swaps"]))
        t = r.timings()
        stats["pairs"].append(t["n_pairs"])
        stats["last"] = t
        if timed:
            stats["comp_ms"].append(t["ms_composite_kernel"])
            stats["comp_slot"].append(int(ticket))
            stats["stage"].append(t)

    def run(n, worker, first=0):
        slots = state["slots"]
        for k in range(n):
            i = first + k
            pos, cu, vp = cams[i % len(cams)]
            if worker is not None and lockstep:
                # N > 1: every E-th camera goes to the worker in order; its SortData is swapped in E frames later on every rank (E: below)
                E = state["every"]
                if i % E == 0:
                    worker.submit_ordered(i, pos, vp)
                if i % E == 0 and i >= E + first_tag[0]:
                    res = worker.take(i - E)
                    if res is not None:
                        swap_in(res)
            elif worker is not None:
                worker.submit(pos, vp)             # state.rs:323-334: camera to the worker, newest wins
                res = worker.poll()
                if res is not None:
                    swap_in(res)
            submit(i, cu, args.timing > 0 and k % max(1, args.timing_every) == 0)
            if len(inflight) == slots:
                collect()
        while inflight:
            collect()
        stream.synchronize()

    lockstep = world > 1            # (also with GSWT_BENCH_FAKE_WORLD)
    first_tag = [0]

    def timed_run(n, worker, warm):
        run(warm, worker)
        for v in stats.values():
            if isinstance(v, list):
                v.clear()
        state["swaps"] = 0; state["swap_ms"].clear()
        if worker is not None:
            worker.build_ms.clear(); worker.sort_ms.clear()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n, worker, first=warm)
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    use_graph = (args.graph or world > 1) and not args.no_graph
    if use_graph:
        r.set_option(L.GSWT_OPT_GRAPH, 1)
    worker = None
    if args.mode == "static" and args.static_at >= 0:
        swap_in(res_k)
    dev_worker = None
    defer_swap = args.mode == "flypath" and world == 1 and not args.no_defer_swap
    if defer_swap:
        # a swap-in takes effect with the first frame submitted after its device-side list build has finished; frames submitted
        # meanwhile keep the previous SortData (the reference's swap-in also lands one frame after the worker's message)
        r.set_option(L.GSWT_OPT_DEFER_SWAP, 1)
    elif args.mode == "flypath" and world > 1 and not args.no_defer_swap:
        # N > 1: every rank must switch at the same frame, and "finished" is a per-GPU event: the swap-in lands with the third
        # frame submitted after it on every rank (the build has two frames' time; a late one is waited for on the device)
        r.set_option(L.GSWT_OPT_DEFER_SWAP, 3)
    if args.mode == "flypath":
        if args.device_worker:
            from gswt_renderer_amd.worker import DeviceWorker
            dev_worker = DeviceWorker(r, wang)
        worker = Worker(wang, dev_worker)
        res = worker.step(cams[0][0], cams[0][2])            # the first SortData, synchronously (State::new + first frames)
        if res is not None:
            swap_in(res)
    if args.freeze_sort and worker is not None:
        worker.close()
    if lockstep and worker is not None and not args.freeze_sort:
        # The reference's worker takes the newest camera whenever it is free (state.rs:323-334); the lock-step stand-in for it hands over every
        # E-th camera, and E has to leave the worker the time of one event or the render thread waits for it: with E = 4, a 0.28-0.56 ms event
        # (sort_tiles, at times + build_tiles, at c3) holds a rank of eight, whose frames take ~75 us, at ~120 us per frame.  E = the event time over the
        # frame time of a short untimed run of this rank's frames (x 1.5, rounded up to a power of two, 4 .. 64), the largest over the ranks.
        # (an event = sort_tiles; a quarter of them also rebuild the tile map on the fly path: 3 build_tiles in 10 events)
        ev_ms = (float(np.mean(worker.sort_ms)) if worker.sort_ms else 0.3) + 0.25 * (float(np.mean(worker.build_ms)) if worker.build_ms else 0.3)
        n_cal = 24
        torch.cuda.synchronize()
        t0c = time.perf_counter()
        run(n_cal, None)
        torch.cuda.synchronize()
        fr_ms = 1e3 * (time.perf_counter() - t0c) / n_cal
        e = 4
        while e < 64 and e * fr_ms < 1.5 * ev_ms:
            e *= 2
        if dist is not None and fake_world <= 1:
            et = torch.tensor([float(e)], dtype=torch.float64, device=dev)
            dist.all_reduce(et, op=dist.ReduceOp.MAX)
            e = int(et.item())
        state["every"] = int(os.environ.get("GSWT_BENCH_LOCKSTEP_EVERY", e))
    dt = timed_run(args.steps, None if args.freeze_sort else worker, args.warmup)
    main_stats = {k: (list(v) if isinstance(v, list) else v) for k, v in stats.items()}
    if trace is not None:
        tb = trace[0][2]
        for kind, idx, a, b, sw in trace:
            print(f"trace {kind:6s} {idx:4d} start {1e6 * (a - tb):9.1f} us  dur {1e6 * (b - a):8.1f} us  swaps {sw}", file=sys.stderr)
        trace = None
    mg_built, mg_reused = r.merge_stats()
    mg_deep = r.merge_stats_deep()
    swaps, swap_ms = state["swaps"], list(state["swap_ms"])
    w_build, w_sort = (list(worker.build_ms), list(worker.sort_ms)) if worker is not None else ([], [])
    # A short timed region (the driver's --steps 20) is mostly pipeline fill and drain: the bracket starts on an idle GPU and ends
    # when the last of the four frames in flight has left it (~0.9 ms of a 5.6 ms region).  `value` stays what the contract says --
    # exactly K steps inside the bracket -- and the same loop over two laps of the path is reported beside it.
    steady = None
    steady_stats = None
    if args.mode == "flypath" and world == 1 and args.steps < 240 and worker is not None and not args.freeze_sort:
        n_long = 2 * len(cams)
        te_save, args.timing_every = args.timing_every, 3            # as the default run: every third frame carries timing events
        dt_long = timed_run(n_long, worker, 0)
        args.timing_every = te_save
        steady_stats = {k: (list(v) if isinstance(v, list) else v) for k, v in stats.items()}
        steady = {"value": n_long / dt_long, "unit": "frames/s", "steps": n_long, "ms_per_step": 1e3 * dt_long / n_long, "sort_events_swapped_in": state["swaps"],
                  "note": f"the same fly-path loop timed over {n_long} frames right after the {args.steps}-step region: what `value` converges to when fill and drain of the four-frame pipeline stop mattering (the default `python bench.py` times 480 frames)"}
    worker_ms = None
    if worker is not None:
        if not args.freeze_sort:
            worker.close()
        worker_ms = {"build_tiles_ms_mean": float(np.mean(w_build)) if w_build else None, "build_tiles_events": len(w_build),
                     "sort_tiles_ms_mean": float(np.mean(w_sort)) if w_sort else None, "sort_tiles_events": len(w_sort),
                     "threads": 1, "note": "libgswt_host (C++ WangTile) on one host thread beside the render thread, as state.rs:478-561; "
                     "sort_tiles in device-merge mode (group descriptions only; the merged lists are built on the GPU at swap-in)"}
        if dev_worker is not None:
            worker_ms["note"] = ("--device-worker: update_tile_map on the host thread, then update_lod, selective merging, tile order, views and "
                                 "records as HIP kernels on the worker's own stream (gswt_worker_*); build = host map + upload + update_lod, "
                                 "sort = device sort event + record read-back")

    # the last fly-path camera again, one frame at a time: k_composite without another frame's kernels sharing the chip, and the
    # image the CPU baseline is compared with.  `roofline` itself comes from the timed region, where the frames overlap.
    last_i = (args.warmup + args.steps - 1) % len(cams) if steady is None else (steady["steps"] - 1) % len(cams)      # the camera the draw list in place belongs to
    pos_l, cu_l, vp_l = cams[last_i]
    su_l = state["su"]
    iso = []
    r.set_option(L.GSWT_OPT_TIMING, max(1, args.timing))
    for i in range(6):
        if use_passes:
            r.skybox_render(cu_l, W, H, bgs[0].data_ptr()); r.proxy_render(pu, W, H, bgs[0].data_ptr(), depths[0].data_ptr(), True)
        r.render_wait(r.render_async(cu_l, su_l, W, H, outs[0].data_ptr(), transmittance_eps=args.t_eps, order_mode=order_mode, shard=shard,
                                     bg_rgba_ptr=bgs[0].data_ptr() if use_passes else 0, bg_depth_ptr=depths[0].data_ptr() if use_passes else 0))
        iso.append(r.timings()["ms_composite_kernel"])
    iso_ms = float(np.median(iso[2:]))
    out_last = outs[0]
    dist_check = None
    if use_dist:
        # the all-gathered frame against the same camera rendered unsharded on this rank, bit for bit
        tk = r.render_async(cu_l, su_l, W, H, outs[0].data_ptr(), transmittance_eps=args.t_eps, order_mode=order_mode, shard=shard,
                            bg_rgba_ptr=bgs[0].data_ptr() if use_passes else 0, bg_depth_ptr=depths[0].data_ptr() if use_passes else 0)
        if abi_comm:
            r.render_gather(tk, frame.data_ptr())
        else:
            r.render_fence(tk)
            with torch.cuda.stream(stream):
                if fake_world > 1:
                    gathered.zero_(); gathered[:rows].copy_(outs[0], non_blocking=True)
                else:
                    dist.all_gather_into_tensor(gathered, outs[0])
                r.unshard_mode(gathered.data_ptr(), W, H, world, "cols", frame.data_ptr())
        r.render_wait(tk)
        stream.synchronize()
        full = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
        r.render_wait(r.render_async(cu_l, su_l, W, H, full.data_ptr(), transmittance_eps=args.t_eps, order_mode=order_mode, shard=(0, 1),
                                     bg_rgba_ptr=bgs[0].data_ptr() if use_passes else 0, bg_depth_ptr=depths[0].data_ptr() if use_passes else 0))
        torch.cuda.synchronize()
        x1 = min(W, band_w) if fake_world > 1 else W          # fake world: only this rank's band was "gathered"
        dist_check = float((frame[:, :x1] - full[:, :x1]).abs().max().item())
        out_last = full

    # static-camera comparison (round 1's measurement): the workload's own camera over a resident draw list
    static = None
    if args.mode == "flypath" and args.static_steps > 0:
        pos_s, cu_s, vp_s = tuple(cam0["pos"]), cu0, vp0
        wk = Worker(wang); res = wk.step(pos_s, np.asarray(vp_s, dtype=np.float32)); wk.close()
        if res is None:
            wang.build_tiles(pos_s); res = (wang.sort_tiles_raw(pos_s, vp_s), wang.scene_uniforms())
        swap_in(res)
        cams_save, cams = cams, [(pos_s, cu_s, vp_s)]
        state["slots"] = slots_static
        dts = timed_run(args.static_steps, None, 30)
        static = {"value": args.static_steps / dts, "unit": "frames/s", "steps": args.static_steps, "n_pairs": int(np.mean(stats["pairs"])),
                  "k_composite_ms": float(np.mean(stats["comp_ms"])) if stats["comp_ms"] else None, "frames_in_flight": slots_static,
                  "note": "one camera (the workload's own), resident draw list, no worker, no swap-ins: what round 1 reported as value"}
        cams = cams_save
    if dist:
        dist.barrier()

    if rank == 0:
        st = main_stats
        fps = args.steps / dt
        # k_composite's hipEvent samples: a short timed region (the driver's --steps 20) holds ~10 of them, two per frame slot -- not a
        # measurement.  The steady-state loop right behind it runs the same frames with events on every third one (>= 150 samples):
        # when the region has fewer than 64 samples the roofline figures are taken from that loop (and say so).
        roof_src = "timed region"
        if steady_stats is not None and len(st["comp_ms"]) < 64 and len(steady_stats["comp_ms"]) >= 64:
            st = dict(st)
            for k in ("comp_ms", "comp_slot", "pairs", "stage"):
                st[k] = steady_stats[k]
            roof_src = f"steady_state loop ({steady['steps']} frames right behind the {args.steps}-step region: the region itself holds {len(main_stats['comp_ms'])} event samples)"
        P = float(np.mean(st["pairs"]))
        # HIP events bracket k_composite on its frame slot's stream.  The runtime multiplexes the streams onto 4 hardware queues, so
        # a slot whose queue is shared with another slot also times that slot's kernels in front of its own.  Per-slot means are
        # reported; `kernel_ms` is the mean over the slot with the lowest one (the one that measures the kernel, frames of the other
        # slots still overlapping it on the chip), `kernel_ms_all_slots` the mean over all samples.
        comp_all = float(np.mean(st["comp_ms"])) * 1e-3 if st["comp_ms"] else 0.0
        by_slot = {}
        for sl_i, ms in zip(st["comp_slot"], st["comp_ms"]):
            by_slot.setdefault(sl_i, []).append(ms)
        slot_means = {k: float(np.mean(v)) for k, v in sorted(by_slot.items()) if len(v) >= 2}
        best_slot = min(slot_means, key=slot_means.get) if slot_means else None
        comp = slot_means[best_slot] * 1e-3 if best_slot is not None else comp_all
        n_px = rows * band_w
        algo_bytes = 52.0 * P + (36.0 if use_passes else 16.0) * n_px   # (4 + 48) B per pair + 16 B per pixel (+ 20 B read with bg colour + depth)
        achieved = algo_bytes / comp / 1e9 if comp > 0 else 0.0
        achieved_all = algo_bytes / comp_all / 1e9 if comp_all > 0 else 0.0
        # what the compositing stage moves by this design: 32-B records + 4-B slot indices per pair and 16 B per output pixel (the tiles
        # without pairs are written by k_combine, so k_composite's own share is smaller still); SURVEY 8(d)'s 52 B per pair assumed 48-B records
        moved_bytes = 36.0 * P + 16.0 * n_px + (20.0 * n_px if use_passes else 0.0)
        last = st["last"]
        # HBM bytes per launch of k_composite and its VALU busy share from the committed PMC passes of the static-camera command
        # (rocprofv3 --pmc, separate passes; profiles/): raw counter bytes.  MI355X_MICROARCH's x2 rule is for wide coalesced
        # streams; these reads are scattered record gathers (uncalibrated), so the raw sum is reported and the x2 figure kept beside it.
        traffic, traffic_note, valu = None, "no PMC summary for this workload under profiles/", None
        pmc_path = next((q for q in (os.path.join(ROOT, "profiles", f"r04_pmc_{args.workload}.json"), os.path.join(ROOT, "profiles", f"r03_pmc_{args.workload}.json"), os.path.join(ROOT, "profiles", f"r02_pmc_{args.workload}.json"), os.path.join(ROOT, "profiles", f"r01_pmc_{args.workload}.json")) if os.path.exists(q)), None)
        if world == 1 and pmc_path:
            try:
                pmc = json.load(open(pmc_path))
                k = next(v for n, v in pmc.items() if "k_composite" in n)
                traffic = (k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024.0
                traffic_note = (f"{os.path.basename(pmc_path)}: FETCH_SIZE {k['FETCH_SIZE'] / 1024:.1f} MiB + WRITE_SIZE {k['WRITE_SIZE'] / 1024:.1f} MiB raw per launch; "
                                f"with the gfx950 x2 wide-read rule the reads would be {2 * k['FETCH_SIZE'] / 1024:.1f} MiB (record gathers: uncalibrated)")
                if "SQ_INSTS_VALU" in k and "GRBM_GUI_ACTIVE" in k:
                    cyc = k["GRBM_GUI_ACTIVE"] / 8.0                       # the counter sums the 8 XCDs
                    valu = {"valu_busy_pct": k.get("VALUBusy"), "wave_insts_valu": k["SQ_INSTS_VALU"], "wave_insts_salu": k.get("SQ_INSTS_SALU"), "wave_insts_lds": k.get("SQ_INSTS_LDS"),
                            "cycles_per_valu_inst_per_simd": cyc * 1024.0 / k["SQ_INSTS_VALU"],
                            "peak_cycles_per_valu_inst_per_simd": 2.0,
                            "valu_issue_frac": 2.0 / (cyc * 1024.0 / k["SQ_INSTS_VALU"]),
                            "lds_bank_conflict_cycle_share": (k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"]) if k.get("SQ_LDS_IDX_ACTIVE") else None,
                            "wave_cycles_waiting_share": (k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"]) if k.get("SQ_WAVE_CYCLES") else None,
                            "source": f"profiles/{os.path.basename(pmc_path)} (committed rocprofv3 --pmc passes of the static-camera command: NOT measured in this run)",
                            "note": "peak issue rate = tools/ubench/valu_issue2.hip: 2.0 cycles @2.4 GHz per wave64 "
                                    "instruction per SIMD with >= 4 waves issuing, ~7 cycles per instruction for one wave alone whatever its ILP"}
            except Exception as e:      # the summary is evidence, not a dependency
                traffic_note = f"could not read {pmc_path}: {e}"
        stage_keys = ("ms_project", "ms_emit", "ms_sort", "ms_ranges", "ms_composite", "ms_composite_kernel", "ms_total")
        stage_ms = {k: float(np.mean([t[k] for t in st["stage"]])) for k in stage_keys} if st["stage"] and args.timing >= 2 else None
        res = {
            "metric": "frames/sec @1920x1080, 32x32 Wang-tile grid" if args.workload == "c3" else f"frames/sec ({args.workload})",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {w['desc']}", "mode": args.mode,
                       "camera": (f"fly path '{path_name}' ({args.path_frames} frames, Catmull-Rom, reference JSON schema), worker thread + swap-ins in the timed region"
                                  if args.mode == "flypath" else "static (the workload's own camera)"),
                       "map": [2 * w["half"][0] + 1, 2 * w["half"][1] + 1],
                       "width": W, "height": H, "n_draws": int(last["n_draws"]), "n_instanced": int(last["n_instanced"]),
                       "n_visible": int(last["n_visible"]), "n_pairs_mean": int(P), "order": args.order,
                       "depth_sort_frames_tile_local_global_longest_list": list(r.depth_stats()) if args.order == "depth" else None,
                       "vertex_stage": "strict (gswt.wgsl:152-258 operator by operator; the default)" if args.vertex_stage == "strict" else "rounding sequence v2 (GSWT_OPT_STRICT_VS = 0)",
                       "transmittance_eps": args.t_eps, "skybox_proxy_passes": bool(use_passes),
                       "parallelism": (f"screen-tile-column bands x{world} (projection culled per band) + RCCL all-gather; no hardware 1 -> N curve has been "
                                       "measured by the builder (one-GPU boxes only): this line is the first" if world > 1 else "single GPU")},
            "frames_in_flight": slots,
            "segment": segment,
            "device_memory_in_use_GB": (lambda fr_to: round((fr_to[1] - fr_to[0]) / 1e9, 2))(torch.cuda.mem_get_info(dev)),
            "sort_events": {"swapped_in": swaps, "swap_in_ms_mean": float(np.mean(swap_ms)) if swap_ms else None,
                            "merged_groups_sorted": mg_built, "merged_groups_copied": mg_reused, "merged_groups_copied_from_older_than_previous_event": mg_deep,
                            "merged_groups_sorted_share": (mg_built / float(mg_built + mg_reused)) if (mg_built + mg_reused) else None,
                            "lockstep_every": (state["every"] if lockstep else None),
                            "takes_effect": "first frame submitted after the event's device-side list build has finished (GSWT_OPT_DEFER_SWAP = 1)" if defer_swap else ("third frame submitted after the swap-in, on every rank (GSWT_OPT_DEFER_SWAP = 3)" if (world > 1 and args.mode == "flypath" and not args.no_defer_swap) else "next frame (which waits for the build on the device)"),
                            "note": "SortData swap-ins inside the timed region (gswt_set_draws_merge_groups: draw-list upload into the spare draw set + merged lists built on the device, on a stream of their own)"},
            "worker_ms": worker_ms,
            "host_submit_ms_mean": float(np.mean(st["submit_ms"])) if st["submit_ms"] else None,
            "graph": ({"launches_rebuilds_node_updates": r.graph_stats(), "note": "GSWT_OPT_GRAPH: one hipGraphLaunch per frame; frames that carry timing events (every --timing-every-th) launch kernel by kernel"} if use_graph else None),
            "stage_ms": stage_ms,
            "steady_state": steady,
            "static_camera": static,
            "roofline": {"bound": "hbm", "kernel": "k_composite", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms_source": roof_src,
                         "valu_issue_frac": valu["valu_issue_frac"] if valu else None,
                         "traffic_source": (f"profiles/{os.path.basename(pmc_path)} (committed PMC passes: NOT measured in this run)" if traffic is not None else None),
                         "algorithmic_bytes_per_launch": algo_bytes,
                         "kernel_ms_min_slot": comp * 1e3, "kernel_ms": comp * 1e3, "kernel_ms_samples": len(by_slot.get(best_slot, st["comp_ms"])), "timed_every": (None if args.timing_every >= (1 << 30) else max(1, args.timing_every)),
                         "kernel_ms_by_slot": {str(k): round(v, 5) for k, v in slot_means.items()}, "kernel_ms_slot": best_slot, "kernel_ms_all_slots": comp_all * 1e3,
                         "frac_all_slots": achieved_all / HBM_PEAK_GBS,
                         "kernel_ms_method": ("start / stop events bound to the k_composite dispatch itself (hipExtLaunchKernelGGL): the kernel's own begin -> end on the device, "
                                              "the interval rocprofv3 --kernel-trace reports for it" if os.environ.get("GSWT_KERNEL_EVENTS", "1") != "0" else
                                              "hipEventRecord in front of and behind the launch (GSWT_KERNEL_EVENTS=0): previous command done -> kernel done, which also "
                                              "holds the time the dispatch waited for other frames' workgroups"),
                         "frac_note": "`achieved` / `frac` use kernel_ms_min_slot, the mean over the frame slot with the lowest one; frac_all_slots uses the mean over every sample "
                                      "(with events bound to the dispatch the slots read alike; with GSWT_KERNEL_EVENTS=0 a slot that shares its hardware queue also times its neighbours' kernels)",
                         "bytes_moved_by_design": moved_bytes, "frac_bytes_moved": (moved_bytes / comp / 1e9) / HBM_PEAK_GBS if comp > 0 else None,
                         "kernel_ms_isolated": iso_ms, "frac_isolated": (algo_bytes / (iso_ms * 1e-3) / 1e9) / HBM_PEAK_GBS if iso_ms > 0 else None,
                         "limiter": "instruction issue + latency, not HBM: BASELINE.json asks for the HBM fraction of the compositing kernel, so that is what `frac` is; "
                                    "what the kernel actually runs against is in `valu`",
                         "traffic_note": traffic_note, "valu": valu},
        }
        # the fraction of a MEASURED device-copy kernel beside the nominal peak (512 MB copied device to device: bytes
        # read + bytes written over the best of five copies)
        try:
            ca = torch.empty(128 << 20, dtype=torch.float32, device=dev); cb = torch.empty_like(ca)
            ca.fill_(1.0); cb.copy_(ca); torch.cuda.synchronize()
            best = None
            for _ in range(5):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record(); cb.copy_(ca); e1.record(); e1.synchronize()
                msc = e0.elapsed_time(e1)
                best = msc if best is None else min(best, msc)
            copy_gbs = 2.0 * ca.numel() * 4 / (best * 1e-3) / 1e9
            res["roofline"]["measured_copy_GBps"] = copy_gbs
            res["roofline"]["frac_of_measured_copy"] = achieved / copy_gbs
            res["roofline"]["frac_isolated_of_measured_copy"] = (algo_bytes / (iso_ms * 1e-3) / 1e9) / copy_gbs if iso_ms > 0 else None
            del ca, cb
        except Exception as e:      # noqa: BLE001 - the copy is a side measurement
            res["roofline"]["measured_copy_GBps"] = None
            print(f"[bench] device-copy measurement failed: {e}", file=sys.stderr)
        if use_dist:
            res["dist_check_max_abs_diff"] = dist_check
            res["collective"] = "gswt_render_gather (ncclAllGather behind the C ABI)" if abi_comm else "torch.distributed.all_gather_into_tensor + gswt_unshard_mode"
        if world == 1 and not args.no_cpu_baseline:
            # The oracle (oracle/gswt_oracle.c, OpenMP) renders fly-path cameras from the product host's draw lists: the LAST camera first
            # (its image is compared with the GPU's frame of the same draw list), then further cameras spread over the path until
            # --cpu-baseline-seconds have gone (at most 24 frames): a bounded sample of the same workload, not one frame.
            wang.set_device_merge(False)
            cam_ids = [last_i] + [int(k * len(cams) / 23.0) % len(cams) for k in range(23)] if len(cams) > 1 else [last_i]
            times, cmp_diff, nthr = [], None, None
            t_budget = time.perf_counter() + max(0.0, args.cpu_baseline_seconds)
            for n_done, ci in enumerate(cam_ids):
                if n_done > 0 and time.perf_counter() >= t_budget:
                    break
                pos_c, cu_c, vp_c = cams[ci]
                if wang.check_update(pos_c):
                    wang.build_tiles(pos_c)
                sort_c = wang.sort_tiles(pos_c, vp_c)
                su_c = wang.scene_uniforms()
                if use_passes:
                    pu.view[:] = cu_c.view[:]; pu.projection[:] = cu_c.projection[:]; pu.cam_pos[:] = cu_c.cam_pos[:]; pu.center_coord[:] = su_c.center_coord[:]
                img_cpu, stc, cdt, nthr = cpu_baseline(wang, sort_c, cu_c, vp_c, su_c, W, H, passes=(faces, mips, pu) if use_passes else None, height_map=hmap,
                                                       order_mode=1 if args.order == "depth" else 0, v2=args.vertex_stage == "v2")
                times.append(cdt)
                if n_done == 0:
                    # ... and the GPU renders exactly that draw list once more for the comparison
                    r.set_draws(sort_c.draws, sort_c.merged_gs_index, sort_c.merged_map_id, sort_c.merged_lod_id)
                    if use_passes:
                        r.skybox_render(cu_c, W, H, bgs[0].data_ptr()); r.proxy_render(pu, W, H, bgs[0].data_ptr(), depths[0].data_ptr(), True)
                    cmp_t = torch.empty((H, W, 4), dtype=torch.float32, device=dev)
                    r.render_wait(r.render_async(cu_c, su_c, W, H, cmp_t.data_ptr(), transmittance_eps=args.t_eps, order_mode=order_mode,
                                                 bg_rgba_ptr=bgs[0].data_ptr() if use_passes else 0, bg_depth_ptr=depths[0].data_ptr() if use_passes else 0))
                    cmp_diff = float(np.max(np.abs(cmp_t.cpu().numpy().astype(np.float64) - img_cpu.astype(np.float64))))
            res["cpu_baseline"] = {"value": len(times) / float(np.sum(times)), "unit": "frames/s", "cores": nthr, "kind": "port",
                                   "sample": f"{len(times)} frames of the same workload (fly-path cameras spread over the path, {float(np.sum(times)):.1f} s of oracle time; "
                                             f"order = {args.order}): oracle/gswt_oracle.c, OpenMP over 16-row bands; the reference itself (Rust + wgpu) cannot run here",
                                   "frames_per_s_min_max": [1.0 / max(times), 1.0 / min(times)],
                                   "max_abs_diff_vs_gpu": cmp_diff}
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(res) + "\n").encode())
    if abi_comm:
        r.comm_destroy()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
