"""Edge cases and error behaviour of the device C ABI (reference panics -> status codes)."""
import ctypes as C

import numpy as np
import pytest

from gswt_renderer_amd import _lib as L
from gswt_renderer_amd.renderer import GSWTError, make_draw
from oracle import gswt_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _upload(renderer, pp):
    renderer.upload_scene(pp.tex, pp.gs_index, pp.gs_lod_id)
    renderer.configure(None)


def test_empty_draw_list_returns_background(renderer):
    pp = H.tileset()
    _upload(renderer, pp)
    renderer.set_draws([])
    W, Hh = 70, 37                                     # not multiples of 16
    cam = orc.default_camera(W, Hh).uniforms()
    su = orc.scene_uniforms(num_lod=pp.n_lod)
    img = renderer.render(cam, su, W, Hh)
    assert img.shape == (Hh, W, 4) and not img.any()
    bg = np.random.default_rng(0).uniform(0, 1, (Hh, W, 4)).astype(np.float32)
    img = renderer.render(cam, su, W, Hh, bg_rgba=bg)
    assert np.array_equal(img, bg)
    assert renderer.timings()["n_pairs"] == 0


def test_nothing_visible_and_all_culled(renderer):
    pp = H.tileset()
    _upload(renderer, pp)
    case = H.grid_case(pp, center=(400, 400))          # far outside the frustum
    renderer.set_draws(case.draws)
    W, Hh = 96, 64
    cam = orc.default_camera(W, Hh).uniforms()
    su = orc.scene_uniforms(num_lod=pp.n_lod, map_half_wh=(1, 2))
    img = renderer.render(cam, su, W, Hh)
    t = renderer.timings()
    assert t["n_visible"] == 0 and t["n_pairs"] == 0 and not img.any()
    # lod_enable mask that disables every draw (renderer.rs:495)
    case = H.grid_case(pp)
    renderer.set_draws(case.draws)
    img = renderer.render(cam, su, W, Hh, lod_enable_mask=0)
    assert renderer.timings()["n_visible"] == 0 and not img.any()


@pytest.mark.parametrize("W,Hh", [(1, 1), (17, 5), (33, 100), (300, 16)])
def test_odd_target_sizes(renderer, W, Hh):
    pp = H.tileset()
    _upload(renderer, pp)
    case = H.grid_case(pp)
    renderer.set_draws(case.draws)
    cam = orc.default_camera(W, Hh).uniforms()
    su = orc.scene_uniforms(num_lod=pp.n_lod, map_half_wh=(1, 2))
    ref, st = orc.render(cam, su, pp.tex, case.orc_draws, W, Hh)
    img = renderer.render(cam, su, W, Hh)
    assert renderer.timings()["n_visible"] == st["n_visible"]
    assert H.max_abs_diff(img, ref) <= TOL


def test_huge_splats_cover_many_tiles(renderer):
    """splat_scale blows every splat up to hundreds of pixels: long per-tile lists, multi-segment tiles,
    k_combine, and early termination all in play."""
    pp = H.tileset(lod0_count=300)
    _upload(renderer, pp)
    case = H.grid_case(pp, half=(1, 2), center=(0, 4))
    renderer.set_draws(case.draws)
    W, Hh = 256, 160
    cam = orc.default_camera(W, Hh).uniforms()
    su = orc.scene_uniforms(num_lod=pp.n_lod, map_half_wh=(1, 2), splat_scale=80.0)
    ref, st = orc.render(cam, su, pp.tex, case.orc_draws, W, Hh)
    for seg in (256, 512, 1536, 4096):
        renderer.set_option(L.GSWT_OPT_SEGMENT, seg)
        img = renderer.render(cam, su, W, Hh)
        assert renderer.timings()["n_pairs"] == st["n_pairs16"] > 100000
        assert H.max_abs_diff(img, ref) <= TOL
        img = renderer.render(cam, su, W, Hh, transmittance_eps=1e-5)
        assert H.max_abs_diff(img, ref) <= TOL
    renderer.set_option(L.GSWT_OPT_SEGMENT, L.GSWT_DEFAULT_SEGMENT)


def test_point_cloud_clip_and_scene_scale(renderer):
    pp = H.tileset()
    _upload(renderer, pp)
    case = H.grid_case(pp)
    renderer.set_draws(case.draws)
    W, Hh = 160, 120
    cam = orc.default_camera(W, Hh).uniforms()
    for kw in (dict(point_cloud_radius=0.01), dict(use_clip=1, clip_height=-0.1), dict(use_clip=1, clip_height=0.1),
               dict(scene_scale=(1.5, 0.75, 2.0)), dict(splat_scale=0.5)):
        su = orc.scene_uniforms(num_lod=pp.n_lod, map_half_wh=(1, 2), **kw)
        ref, st = orc.render(cam, su, pp.tex, case.orc_draws, W, Hh)
        img = renderer.render(cam, su, W, Hh)
        assert renderer.timings()["n_visible"] == st["n_visible"], kw
        assert H.max_abs_diff(img, ref) <= TOL, kw


def test_error_codes(renderer):
    pp = H.tileset()
    lib = L.load()
    # a fresh ctx: calls out of order
    h = C.c_void_p()
    assert lib.gswt_create(0, C.byref(h)) == L.GSWT_OK
    try:
        d = (L.Draw * 1)()
        assert lib.gswt_set_draws(h, d, 1, None, None, None, 0) == L.GSWT_ERR_STATE            # before upload_scene
        cam = orc.default_camera(32, 32).uniforms()
        su = orc.scene_uniforms(num_lod=1)
        cfg = L.RenderConfig(); cfg.culling_dist = 1.0; cfg.lod_enable_mask = 0xFFFFFFFF
        out = np.zeros((32, 32, 4), np.float32)
        assert lib.gswt_render(h, C.byref(cam), C.byref(su), C.byref(cfg), 32, 32, None, None, 0, out.ctypes.data, 0) == L.GSWT_ERR_STATE
        assert b"gswt_set_draws" in lib.gswt_last_error(h)
        assert lib.gswt_create(9999, C.byref(C.c_void_p())) == L.GSWT_ERR_BAD_ARG
    finally:
        lib.gswt_destroy(h)
    _upload(renderer, pp)
    case = H.grid_case(pp)
    bad = make_draw(H.to_product_tile(orc.tile_uniforms(valid_lod_id=0)), base=(7, 0, 0))                # lod out of range
    with pytest.raises(GSWTError) as e:
        renderer.set_draws([bad])
    assert e.value.code == L.GSWT_ERR_BAD_ARG
    bad = make_draw(H.to_product_tile(orc.tile_uniforms(single_draw=1)), merged_range=(0, 10))           # merged range beyond the arrays
    with pytest.raises(GSWTError):
        renderer.set_draws([bad], np.zeros(4, np.uint32), np.zeros(4, np.uint32), None)
    renderer.set_draws(case.draws)
    cam = orc.default_camera(64, 48).uniforms()
    with pytest.raises(GSWTError):                                                                       # viewport mismatch
        renderer.render(cam, orc.scene_uniforms(num_lod=2), 80, 48)
    with pytest.raises(GSWTError):                                                                       # sphere surface not built yet
        renderer.render(cam, orc.scene_uniforms(num_lod=2, surface_type=2), 64, 48)
    with pytest.raises(GSWTError):                                                                       # height map surface without a map
        renderer.render(cam, orc.scene_uniforms(num_lod=2, surface_type=1), 64, 48)
    with pytest.raises(GSWTError):
        renderer.render(cam, orc.scene_uniforms(num_lod=2), 64, 48, shard=(3, 2))
    with pytest.raises(GSWTError):
        renderer.set_option(L.GSWT_OPT_SEGMENT, 100)
    # the profiling ablations (wrong images by design) are not in the product library: it cannot be switched into them
    with pytest.raises(GSWTError) as e:
        renderer.set_option(L.GSWT_OPT_DEBUG_FLAGS, 4)
    assert e.value.code == L.GSWT_ERR_BAD_ARG
    renderer.set_option(L.GSWT_OPT_DEBUG_FLAGS, 0)
    for key, bad_value in ((L.GSWT_OPT_COMPOSITE, 3), (L.GSWT_OPT_DEPTH_PASSES, 0), (L.GSWT_OPT_DEPTH_PASSES, 5)):
        with pytest.raises(GSWTError):
            renderer.set_option(key, bad_value)
    # the ctx is still usable afterwards
    img = renderer.render(cam, orc.scene_uniforms(num_lod=2, map_half_wh=(1, 2)), 64, 48)
    assert img.shape == (48, 64, 4)


def test_fenced_consumer_never_sees_an_overflowed_frame(renderer):
    """ADVICE r1 / VERDICT r1: with the pair capacity pinned far below the frame's pair count, work ordered behind
    gswt_render_fence on the ctx stream (here: a device-to-device copy, standing in for the RCCL all-gather) must still read
    the complete frame -- the fence re-runs the overflowed frame before it releases the ctx stream."""
    import torch
    from gswt_renderer_amd import host, synth
    from gswt_renderer_amd.pipeline import GSWTPipeline
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=800)
    pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer)
    W, Hh = 256, 160
    pos = (4.2, 1.0, 2.0)
    cu, vp = host.camera_uniforms(pos, (5.0, 3.0, 1.5), (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh)
    pipe.update(pos, vp)
    su = pipe.wang.scene_uniforms()
    want = renderer.render(cu, su, W, Hh)
    n_pairs = renderer.timings()["n_pairs"]
    assert n_pairs > 1000
    stream = torch.cuda.Stream()
    renderer.set_stream(stream.cuda_stream)
    try:
        out = torch.zeros((Hh, W, 4), dtype=torch.float32, device="cuda")
        seen = torch.zeros_like(out)
        torch.cuda.synchronize()
        renderer.set_option(L.GSWT_OPT_PAIR_CAP, 256)             # every frame overflows until the capacity has grown
        ticket = renderer.render_async(cu, su, W, Hh, out.data_ptr())
        renderer.render_fence(ticket)
        with torch.cuda.stream(stream):
            seen.copy_(out, non_blocking=True)                     # the fenced consumer
        renderer.render_wait(ticket)
        stream.synchronize()
        assert renderer.timings()["n_pairs"] == n_pairs
        assert np.array_equal(seen.cpu().numpy(), want)
        # the same with a sort event between submit and wait: the pending frame finishes against ITS draw list
        renderer.set_option(L.GSWT_OPT_PAIR_CAP, 256)
        out.zero_()
        torch.cuda.synchronize()
        ticket = renderer.render_async(cu, su, W, Hh, out.data_ptr())
        renderer.set_draws([])
        renderer.render_wait(ticket)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
    finally:
        renderer.set_option(L.GSWT_OPT_PAIR_CAP, 0)
        renderer.set_stream(0)


def test_more_shards_than_tile_columns_or_rows(renderer):
    """a rank whose band holds no screen tile (more ranks than 16-px tile columns / rows) renders an empty shard
    frame after frame with every timing level, and the non-empty shards still assemble the unsharded image bitwise."""
    pp = H.tileset()
    _upload(renderer, pp)
    case = H.grid_case(pp)
    renderer.set_draws(case.draws)
    W, Hh = 40, 24                                     # 3 tile columns, 2 tile rows
    cam = orc.default_camera(W, Hh).uniforms()
    su = orc.scene_uniforms(num_lod=pp.n_lod)
    full = renderer.render(cam, su, W, Hh)
    n = 8
    bw = renderer.shard_cols_padded(W, n)
    for level in (2, 1, 0):
        renderer.set_option(L.GSWT_OPT_TIMING, level)
        uni = np.zeros_like(full)
        for _ in range(2):                              # twice: a sticky error of the first frame would fail the second
            for r in range(n):
                part = renderer.render(cam, su, W, Hh, shard=(r, n, "cols"))
                x0, x1 = r * bw, min(W, (r + 1) * bw)
                if x1 > x0:
                    uni[:, x0:x1] = part[:, :x1 - x0]
                else:
                    assert renderer.timings()["n_tiles"] == 0 and not part.any()
        assert np.array_equal(uni, full)
        rows_p = renderer.shard_rows_padded(Hh, n)
        uni = np.zeros_like(full)
        for r in range(n):
            part = renderer.render(cam, su, W, Hh, shard=(r, n))
            assert part.shape == (rows_p, W, 4)
            for y in range(Hh):
                if (y // 16) % n == r:
                    uni[y] = part[((y // 16) // n) * 16 + (y % 16)]
        assert np.array_equal(uni, full)
    renderer.set_option(L.GSWT_OPT_TIMING, 2)


def test_pair_total_is_carried_in_64_bits(renderer):
    """k_totals folds the super-group sums in 64 bits -- a frame past 2^32 pairs reports its true count and the
    overflow flag instead of wrapping."""
    lib = L.load()
    rng = np.random.default_rng(3)
    for n_super, big in ((7, False), (1500, True), (2049, True)):
        pairs = rng.integers(0, 3_000_000 if big else 1000, size=n_super, dtype=np.uint32)
        if big:
            pairs[5] = 0xFFFFFFF0
        vis = rng.integers(0, 65536, size=n_super, dtype=np.uint32)
        cnt = (C.c_ulonglong * 4)()
        excl = np.zeros(n_super, dtype=np.uint32)
        cap = 1 << 30
        rc = lib.gswt_debug_totals(renderer._h, pairs.ctypes.data, vis.ctypes.data, n_super, cap, cnt, excl.ctypes.data)
        assert rc == 0
        total = int(pairs.astype(np.uint64).sum())
        assert cnt[1] == total and cnt[0] == int(vis.astype(np.uint64).sum())
        assert (cnt[3] != 0) == (total > cap)
        want = np.concatenate([np.zeros(1, np.uint64), np.cumsum(pairs.astype(np.uint64))[:-1]]) & np.uint64(0xFFFFFFFF)
        assert np.array_equal(excl.astype(np.uint64), want)


@pytest.mark.gpu
@pytest.mark.parametrize("n,key_bits", [(1, 13), (4095, 13), (4097, 5), (1 << 20, 13), (5_000_003, 15), (12_345_678, 15), (9_000_000, 32)])
def test_radix_sort_is_stable_at_every_size(renderer, n, key_bits):
    """The frame's radix sort alone (gswt_debug_sort) against numpy's stable sort: one workgroup, a ragged last block, the
    direct group sums (<= 32 groups of 32 workgroups), k_radix_supscan's group prefixes (more), the 256-thread build (> 8 M
    items) and a four-pass 32-bit sort (the depth order's).  Keys are skewed the way pair keys are (runs of equal tiles)."""
    rng = np.random.default_rng(n ^ key_bits)
    runs = rng.integers(0, 1 << min(key_bits, 31), size=max(1, n // 3), dtype=np.uint64).astype(np.uint32)
    keys = np.repeat(runs, 3)[:n].copy() if n >= 3 else runs[:n].copy()
    if len(keys) < n:
        keys = np.concatenate([keys, rng.integers(0, 1 << min(key_bits, 31), size=n - len(keys), dtype=np.uint64).astype(np.uint32)])
    if key_bits == 32:
        keys ^= rng.integers(0, 2, size=n, dtype=np.uint64).astype(np.uint32) << np.uint32(31)
    vals = np.arange(n, dtype=np.uint32)
    order = np.argsort(keys, kind="stable")
    k2, v2 = keys.copy(), vals.copy()
    lib = L.load()
    rc = lib.gswt_debug_sort(renderer._h, k2.ctypes.data, v2.ctypes.data, n, key_bits)
    assert rc == 0, renderer.last_error() if hasattr(renderer, "last_error") else rc
    assert np.array_equal(k2, keys[order])
    assert np.array_equal(v2, vals[order])          # equal keys keep their input order


@pytest.mark.gpu
def test_tile_depth_sort_is_stable_at_every_class_boundary(renderer):
    """k_tile_depth_sort alone (gswt_debug_tile_depth_sort) against numpy's stable sort, slice by slice: empty tiles, one pair, the
    boundaries of its four size classes (a wave up to 512 pairs, a 256-thread workgroup up to 4 096, the long-list grid up to 16 384 in LDS, beyond
    that through global memory),
    ragged last rounds, one depth for a whole slice, spans of 1 .. 32 bits (one to four passes, digits of 1 .. 8 bits), many equal keys
    (ties keep the list's order: the order contract of scene.rs:685-695 inside a depth tie)."""
    import ctypes as C
    rng = np.random.default_rng(20261005)
    lens = [0, 1, 2, 63, 64, 65, 0, 127, 129, 511, 512, 513, 514, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 0, 4095, 4096, 4097, 4098, 5000, 8191, 8192,
            8193, 12000, 16383, 16384, 3, 0, 700, 300, 4096, 513, 512, 16384, 1]
    keys, want = [], []
    for i, n in enumerate(lens):
        bits = (1, 3, 8, 9, 13, 16, 17, 21, 24, 25, 32, 0)[i % 12]         # span of the slice's keys (0: one depth)
        if bits == 0:
            k = np.full(n, 0x3F7F1234, dtype=np.uint32)
        else:
            base = np.uint64(rng.integers(0, (1 << 32) - (1 << bits) + 1)) if bits < 32 else np.uint64(0)
            k = (base + rng.integers(0, 1 << bits, size=n, dtype=np.uint64)).astype(np.uint32)
            if n > 8 and i % 3 == 0:                                          # runs of equal depths
                k = np.repeat(k[: (n + 3) // 4], 4)[:n].copy()
                k = k[rng.permutation(n)]
        keys.append(k)
    dk = np.concatenate(keys).astype(np.uint32)
    n = int(dk.size)
    vals = rng.permutation(n).astype(np.uint32)                               # any payload: the sort must not look at it
    out = vals.copy()
    la = np.asarray(lens, dtype=np.uint32)
    flagged = C.c_int(-1)
    lib = L.load()
    rc = lib.gswt_debug_tile_depth_sort(renderer._h, la.ctypes.data, la.size, out.ctypes.data, dk.ctypes.data, n, C.byref(flagged))
    assert rc == 0 and flagged.value == 0
    at = 0
    for i, m in enumerate(lens):
        order = np.argsort(dk[at:at + m], kind="stable")
        assert np.array_equal(out[at:at + m], vals[at:at + m][order]), (i, m)
        at += m
    # slices beyond the LDS buffer go through k_tile_depth_sort_xl (passes through global memory): odd and even pass counts, several per grid
    lens2 = [100, 16385, 50, 40000, 0, 16384, 123457, 20000, 7]
    spans = [12, 8, 5, 16, 1, 9, 24, 32, 3]
    ks = []
    for m, bits in zip(lens2, spans):
        base = np.uint64(rng.integers(0, (1 << 32) - (1 << bits) + 1)) if bits < 32 else np.uint64(0)
        ks.append((base + rng.integers(0, 1 << bits, size=m, dtype=np.uint64)).astype(np.uint32))
    dk2 = np.concatenate(ks).astype(np.uint32)
    n2 = int(dk2.size)
    v2 = rng.permutation(n2).astype(np.uint32)
    o2 = v2.copy()
    la2 = np.asarray(lens2, dtype=np.uint32)
    rc = lib.gswt_debug_tile_depth_sort(renderer._h, la2.ctypes.data, la2.size, o2.ctypes.data, dk2.ctypes.data, n2, C.byref(flagged))
    assert rc == 0 and flagged.value == 0
    at = 0
    for i, m in enumerate(lens2):
        order = np.argsort(dk2[at:at + m], kind="stable")
        assert np.array_equal(o2[at:at + m], v2[at:at + m][order]), (i, m)
        at += m


@pytest.mark.gpu
def test_frame_whose_live_chunks_outgrow_its_launch_grid_is_rerun(renderer):
    """The launch grids of k_project / k_emit follow the longest live-chunk list of the last finished frame (+ 50 % + 256, and only where that
    removes at least half of the launch table).  A camera that sees almost nothing, then one that sees the scene: the second frame's lists
    outgrow its grid, k_totals flags it, the host re-runs it over the whole table -- the image and the counts are the full frame's."""
    import bench
    from gswt_renderer_amd import host, workloads
    w, wang, cu, vp, sort = bench.build_workload("c3")
    W, Hh = w["width"], w["height"]
    su = wang.scene_uniforms()
    wang.upload_to(renderer)
    renderer.configure(None)
    renderer.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
    cam = workloads.camera_for("c3")
    pos = cam["pos"]
    up_cu, _ = host.camera_uniforms(pos, (pos[0] + 0.3, pos[1] + 0.2, pos[2] + 10.0), cam["up"], cam["fovy"], cam["near"], cam["far"], W, Hh)      # at the sky
    want = renderer.render(cu, su, W, Hh)
    t0 = renderer.timings()
    assert t0["n_visible"] > 1_000_000
    for order in (L.GSWT_ORDER_REFERENCE, L.GSWT_ORDER_DEPTH):
        ref = want if order == L.GSWT_ORDER_REFERENCE else renderer.render(cu, su, W, Hh, order_mode=order)
        sky = renderer.render(up_cu, su, W, Hh, order_mode=order)
        ts = renderer.timings()
        assert ts["n_visible"] < t0["n_visible"] // 20, ts["n_visible"]                 # a short live list: the next frame's grid is cut
        img = renderer.render(cu, su, W, Hh, order_mode=order)
        t1 = renderer.timings()
        assert (t1["n_visible"], t1["n_pairs"]) == (t0["n_visible"], t0["n_pairs"])
        assert np.array_equal(img, ref)
        # ... and with several frames in flight behind the short one
        import torch
        outs = [torch.empty((Hh, W, 4), dtype=torch.float32, device="cuda") for _ in range(4)]
        renderer.render(up_cu, su, W, Hh, order_mode=order)
        tickets = [renderer.render_async(cu, su, W, Hh, o.data_ptr(), order_mode=order) for o in outs]
        for tk in tickets:
            renderer.render_wait(tk)
        for o in outs:
            assert np.array_equal(o.cpu().numpy(), ref)
        assert sky[..., 3].max() <= want[..., 3].max()
