"""N>1 path on CPU: two gloo ranks all-gather their screen-tile-row shards and rebuild the frame.
The shard layout is the one the HIP compositor writes (GPU test test_shard_union_equals_unsharded
checks the kernel against the same layout)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gswt_renderer_amd import dist as gd


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, H, W, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(7)
        frame = torch.rand((H, W, 4), generator=g)
        shard = gd.shard_of_frame(frame, rank, world)
        fg = gd.FrameGather(H, W, torch.device("cpu"))
        full = fg(shard)
        q.put((rank, bool(torch.equal(full, frame)), fg.rows_padded))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W", [(2, 1080, 64), (2, 150, 40), (3, 100, 24)])
def test_all_gather_rebuilds_frame(world, H, W):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, rp in res:
        assert ok, rank
        assert rp == gd.rows_padded(H, world)


def test_layout_matches_c_abi_helpers():
    """gswt_shard_rows_padded / row ownership in the C ABI agree with the dist layer."""
    import ctypes as C
    from gswt_renderer_amd import _lib as L
    lib = L.load()
    for H in (1, 15, 16, 17, 150, 1080, 2160):
        for world in (1, 2, 3, 4, 8):
            assert lib.gswt_shard_rows_padded(H, world) == gd.rows_padded(H, world)
            owned = sum(lib.gswt_shard_rows(H, r, world) for r in range(world))
            assert owned == H


def test_unshard_single_process():
    H, W, world = 100, 8, 4
    frame = torch.arange(H * W * 4, dtype=torch.float32).reshape(H, W, 4)
    gathered = torch.cat([gd.shard_of_frame(frame, r, world) for r in range(world)], dim=0)
    assert torch.equal(gd.unshard(gathered, H, world), frame)


def _worker_cols(rank, world, port, H, W, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(11)
        frame = torch.rand((H, W, 4), generator=g)
        shard = gd.shard_of_frame_cols(frame, rank, world)
        gathered = torch.empty((world * H, shard.shape[1], 4), dtype=torch.float32)
        dist.all_gather_into_tensor(gathered, shard.contiguous())
        full = gd.unshard_cols(gathered, W, world)
        q.put((rank, bool(torch.equal(full, frame)), shard.shape[1]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W", [(2, 64, 1920), (2, 40, 150), (3, 24, 100)])
def test_all_gather_rebuilds_frame_column_bands(world, H, W):
    """The column-band layout bench.py uses for N > 1 (GSWT_SHARD_COLUMNS): gloo all-gather + re-assembly."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cols, args=(r, world, port, H, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, bw in res:
        assert ok, rank
        assert bw == gd.cols_padded(W, world)


def test_column_layout_matches_c_abi_helpers():
    from gswt_renderer_amd import _lib as L
    lib = L.load()
    for W in (1, 15, 16, 17, 150, 1920, 3840):
        for world in (1, 2, 3, 4, 8):
            assert lib.gswt_shard_cols_padded(W, world) == gd.cols_padded(W, world)
    H, W, world = 20, 100, 4
    frame = torch.arange(H * W * 4, dtype=torch.float32).reshape(H, W, 4)
    gathered = torch.cat([gd.shard_of_frame_cols(frame, r, world) for r in range(world)], dim=0)
    assert torch.equal(gd.unshard_cols(gathered, W, world), frame)


def _lockstep_worker(rank, world, port, q):
    """bench.py's lock-step worker (N > 1 ranks must swap the SAME SortData in at the SAME frame): every rank feeds its own
    Worker the same ordered cameras and takes the results by tag; the digests are all-gathered over gloo."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from gswt_renderer_amd import flypath, host, workloads
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, wang, cu0, vp0, sort0 = bench.build_workload("tiny")
        W, H = w["width"], w["height"]
        cam0 = workloads.camera_for("tiny")
        # a short path that crosses several map cells: build events (tile ids from the worker's RNG) and sort events
        cams = []
        for k in range(24):
            pos = (cam0["pos"][0] + 0.9 * k, cam0["pos"][1] + 0.5 * k, cam0["pos"][2])
            tgt = (pos[0] + 1.0, pos[1] + 2.0, pos[2] - 0.5)
            cu, vp = host.camera_uniforms(pos, tgt, (0, 0, 1), 45.0, 0.1, 2400.0, W, H)
            cams.append((pos, np.asarray(vp, dtype=np.float32)))
        wk = bench.Worker(wang)
        digests = []
        import time
        for i, (pos, vp) in enumerate(cams):
            if i % 4 == 0:
                wk.submit_ordered(i, pos, vp)
            if rank == 1 and i % 3 == 0:
                time.sleep(0.003)                   # the ranks' render loops run at different speeds
            if i % 4 == 0 and i >= 4:
                res = wk.take(i - 4)
                h = hashlib.sha256()
                if res is not None:
                    raw, su = res                   # (draws, n_draws, groups, n_groups, members, n_members) + the scene uniforms
                    import ctypes as C
                    draws, n_d, groups, n_g, members, n_m = raw
                    h.update(bytes(su))
                    h.update(C.string_at(draws, n_d * C.sizeof(draws._type_)))
                    h.update(C.string_at(groups, n_g * C.sizeof(groups._type_)))
                    h.update(C.string_at(members, n_m * C.sizeof(members._type_)))
                digests.append(h.hexdigest() if res is not None else "none")
        wk.close()
        got = [None] * world
        dist.all_gather_object(got, digests)
        q.put((rank, got, sum(d != "none" for d in digests)))
    finally:
        dist.destroy_process_group()


def test_lockstep_worker_is_deterministic_across_ranks():
    """Worker.submit_ordered / take give every rank the same SortData sequence (same tile maps, draw lists and
    scene uniforms at the same tags) although the ranks' loops run at different speeds."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lockstep_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, n_events in res:
        assert n_events >= 3, n_events                         # the path really produced sort events
        assert all(g == got[0] for g in got), rank              # every rank saw the same sequence of digests
