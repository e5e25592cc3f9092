"""GSWT_ORDER_DEPTH -- the "global radix depth sort" BASELINE.json's north_star names -- at BASELINE sizes.

The reference orders TILES back to front and each tile's presorted list (wangtile.rs:489-499, scene.rs:685-695): that is
GSWT_ORDER_REFERENCE.  GSWT_ORDER_DEPTH composites every visible splat of the frame in true depth order (stable: equal depths keep
the reference order); the CPU checker's order_mode 1 does the same with a stable merge sort (oracle/gswt_oracle.c orc_render).
Checked here: c3 and the dense c3d at 1920x1080 against the checker (image <= 1e-4, visible / pair counts equal), with and without
the early-out; the 8 column bands of c4's layout (bitwise union); the frame as one hipGraph; the depth sort's adaptive pass count."""
import numpy as np
import pytest

from gswt_renderer_amd import _lib as L
from oracle import gswt_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _setup(renderer, name):
    import bench
    w, wang, cu, vp, sort = bench.build_workload(name)
    W, Hh = w["width"], w["height"]
    su = wang.scene_uniforms()
    hm = wang.height_map() if int(wang.user.surface_type) == 1 else None
    wang.upload_to(renderer)
    renderer.configure(hm)
    renderer.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
    tex, draws = bench.oracle_draws(wang, sort, vp)
    ocu = orc.Camera176.from_buffer_copy(bytes(cu))
    osu = orc.Scene160.from_buffer_copy(bytes(su))
    # (wang owns the memory tex and the draws' index arrays are views of: it has to outlive them)
    return dict(W=W, H=Hh, cu=cu, su=su, hm=hm, tex=tex, draws=draws, ocu=ocu, osu=osu, wang=wang, sort=sort)


@pytest.mark.parametrize("name", ["c3", "c3d"])
def test_depth_order_at_baseline_size(renderer, name):
    s = _setup(renderer, name)
    W, Hh = s["W"], s["H"]
    ref_d, st = orc.render(s["ocu"], s["osu"], s["tex"], s["draws"], W, Hh, height_map=s["hm"], order_mode=1)
    ref_r, _ = orc.render(s["ocu"], s["osu"], s["tex"], s["draws"], W, Hh, height_map=s["hm"], order_mode=0)
    assert st["n_visible"] > 1_000_000
    img = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH)
    t = renderer.timings()
    assert t["n_visible"] == st["n_visible"] and t["n_pairs"] == st["n_pairs16"]
    assert H.max_abs_diff(img, ref_d) <= TOL
    # what bench.py --order depth times: the early-out at 1e-5
    img_e = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH, transmittance_eps=1e-5)
    assert H.max_abs_diff(img_e, ref_d) <= TOL
    # the two ways to get there -- global radix passes on the depth bits in front of the tile passes (the default), or tile passes first and
    # every tile's slice depth-sorted in LDS (the default; GSWT_OPT_DEPTH_SORT = 0 / 2; lists beyond 16 384 pairs through global memory, see
    # the test below) -- give the same bits
    try:
        renderer.set_option(L.GSWT_OPT_DEPTH_SORT, 1)
        l0, g0, _ = renderer.depth_stats()
        img_g = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH)
        l1, g1, max_len = renderer.depth_stats()
        assert (l1 - l0, g1 - g0) == (0, 1) and max_len > 256
        renderer.set_option(L.GSWT_OPT_DEPTH_SORT, 2)
        img_l = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH)
        l2, g2, _ = renderer.depth_stats()
        print(f"{name}: longest tile list {max_len} pairs; tile-local frames {l2 - l1}, global {g2 - g1}")
        assert (l2 - l1, g2 - g1) == (1, 0)
        assert np.array_equal(img_g, img) and np.array_equal(img_l, img)
    finally:
        renderer.set_option(L.GSWT_OPT_DEPTH_SORT, 0)
    # the mode is not a no-op: neighbouring Wang tiles interpenetrate, and a tile's presorted list is only ordered along one of nine
    # directions, so the reference order and the true depth order give different images
    d = H.max_abs_diff(ref_d, ref_r)
    print(f"{name}: depth order vs reference order L-inf {d:.3e}; GPU vs checker (depth) {H.max_abs_diff(img, ref_d):.3e}")
    assert d > 1e-3
    # the reference order on the same context right after: the two modes share every buffer but the depth sort's
    img_r = renderer.render(s["cu"], s["su"], W, Hh)
    assert H.max_abs_diff(img_r, ref_r) <= TOL
    if name == "c3":
        # c4's layout in depth order: 8 column bands, each with its own cull and its own (smaller) depth sort; bitwise the frame
        n = 8
        bw = renderer.shard_cols_padded(W, n)
        uni = np.zeros_like(img)
        for r in range(n):
            part = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH, shard=(r, n, "cols"))
            x0, x1 = r * bw, min(W, (r + 1) * bw)
            uni[:, x0:x1] = part[:, :x1 - x0]
        assert np.array_equal(uni, img)
        # interleaved rows, three ranks
        rows = renderer.shard_rows_padded(Hh, 3)
        uni = np.zeros_like(img)
        for r in range(3):
            part = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH, shard=(r, 3))
            for ty in range(r, (Hh + 15) // 16, 3):
                y0, y1 = ty * 16, min(Hh, ty * 16 + 16)
                uni[y0:y1] = part[(ty // 3) * 16:(ty // 3) * 16 + (y1 - y0)]
        assert rows >= 16 and np.array_equal(uni, img)


def test_tile_local_depth_sort_long_lists(renderer):
    """c3's scene on a small framebuffer: few screen tiles, so the pair lists are long.  At 800x448 (1 400 tiles) the longest lists go through
    the long-list workgroups of k_tile_depth_sort (4 096 < pairs <= 16 384, sorted inside LDS); at 256x144 (144 tiles) lists exceed the LDS
    buffer and go through k_tile_depth_sort_xl (passes through global memory, one workgroup per list).  No frame is re-run for the length of a
    list.  Same bits as the global depth passes either way, within 1e-4 of the checker, launch by launch and as one hipGraph per frame."""
    import bench
    from gswt_renderer_amd import host, workloads
    s = _setup(renderer, "c3")
    cam = workloads.camera_for("c3")
    try:
        for (W, Hh), lo, hi in (((800, 448), 4096, 16384), ((256, 144), 16384, 1 << 30)):
            cu, vp = host.camera_uniforms(cam["pos"], cam["target"], cam["up"], cam["fovy"], cam["near"], cam["far"], W, Hh)
            s["cu"], s["ocu"] = cu, orc.Camera176.from_buffer_copy(bytes(cu))
            s["tex"], s["draws"] = bench.oracle_draws(s["wang"], s["sort"], vp)
            renderer.set_option(L.GSWT_OPT_DEPTH_SORT, 1)
            img_g = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH)
            _, _, max_len = renderer.depth_stats()
            renderer.set_option(L.GSWT_OPT_DEPTH_SORT, 0)
            l0, g0, _ = renderer.depth_stats()
            img_l = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH)
            l1, g1, _ = renderer.depth_stats()
            print(f"{W}x{Hh}: longest tile list {max_len} pairs; tile-local frames {l1 - l0}, global {g1 - g0}")
            assert lo < max_len <= hi
            assert (l1 - l0, g1 - g0) == (1, 0)
            assert np.array_equal(img_l, img_g)
            ref_d, st = orc.render(s["ocu"], s["osu"], s["tex"], s["draws"], W, Hh, height_map=s["hm"], order_mode=1)
            assert renderer.timings()["n_pairs"] == st["n_pairs16"]
            assert H.max_abs_diff(img_l, ref_d) <= TOL
            renderer.set_option(L.GSWT_OPT_GRAPH, 1)
            renderer.set_option(L.GSWT_OPT_TIMING, 0)
            try:
                img_q = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH)
                img_q2 = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH, transmittance_eps=1e-5)
            finally:
                renderer.set_option(L.GSWT_OPT_GRAPH, 0)
                renderer.set_option(L.GSWT_OPT_TIMING, 2)
            assert np.array_equal(img_q, img_g) and H.max_abs_diff(img_q2, ref_d) <= TOL
    finally:
        renderer.set_option(L.GSWT_OPT_DEPTH_SORT, 0)


@pytest.mark.parametrize("depth_sort", [0, 1])
def test_depth_order_frames_replay_as_one_graph(renderer, depth_sort):
    """(depth_sort: the tile-local LDS sort -- the default -- and the global depth passes.)  No host word enters the depth-ordered chain any more (the sort reads its item count and key range on the device), so
    GSWT_OPT_GRAPH covers it: same bits as launch by launch, over a moving camera, all frame slots in flight."""
    import torch
    import bench
    from gswt_renderer_amd import host, workloads
    s = _setup(renderer, "c3")
    W, Hh = s["W"], s["H"]
    cam = workloads.camera_for("c3")
    cams = []
    for k in range(6):
        pos = (cam["pos"][0] + 0.15 * k, cam["pos"][1] + 0.4 * k, cam["pos"][2])
        tgt = (cam["target"][0] + 0.15 * k, cam["target"][1] + 0.4 * k, cam["target"][2] - 0.05 * k)
        cams.append(host.camera_uniforms(pos, tgt, cam["up"], cam["fovy"], cam["near"], cam["far"], W, Hh)[0])
    outs = [torch.empty((Hh, W, 4), dtype=torch.float32, device="cuda") for _ in cams]
    renderer.set_option(L.GSWT_OPT_TIMING, 0)
    renderer.set_option(L.GSWT_OPT_DEPTH_SORT, depth_sort)
    try:
        want = []
        for cu, o in zip(cams, outs):
            renderer.render_wait(renderer.render_async(cu, s["su"], W, Hh, o.data_ptr(), order_mode=L.GSWT_ORDER_DEPTH, transmittance_eps=1e-5))
            want.append(o.cpu().numpy().copy())
        g0 = renderer.graph_stats()
        renderer.set_option(L.GSWT_OPT_GRAPH, 1)
        tickets = []
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()                        # (torch's fills run on ITS stream: nothing orders them against the frame slots' streams)
        for cu, o in zip(cams, outs):
            if len(tickets) >= renderer.frame_slots():
                renderer.render_wait(tickets.pop(0))
            tickets.append(renderer.render_async(cu, s["su"], W, Hh, o.data_ptr(), order_mode=L.GSWT_ORDER_DEPTH, transmittance_eps=1e-5))
        for tk in tickets:
            renderer.render_wait(tk)
        g1 = renderer.graph_stats()
    finally:
        renderer.set_option(L.GSWT_OPT_GRAPH, 0)
        renderer.set_option(L.GSWT_OPT_TIMING, 2)
        renderer.set_option(L.GSWT_OPT_DEPTH_SORT, 0)
    assert g1[0] - g0[0] == len(cams)                   # every frame went through hipGraphLaunch
    for i, (a, o) in enumerate(zip(want, outs)):
        b = o.cpu().numpy()
        if not np.array_equal(a, b):
            dd = np.abs(a - b).max(axis=2)
            ys, xs = np.nonzero(dd)
            raise AssertionError(f"frame {i}: max diff {dd.max():.3e} at {len(ys)} pixels, screen tiles {sorted(set(zip((ys // 16).tolist(), (xs // 16).tolist())))[:8]}")
    assert not np.array_equal(want[0], want[-1])


def test_depth_sort_pass_count_follows_the_depth_range(renderer):
    """(GSWT_OPT_DEPTH_SORT = 1: the global depth passes.)  The depth sort launches as many 8-bit passes as the depth ranges of recent frames needed; a frame whose visible depths span more bits
    is flagged on the device (k_items) and re-run with more, like a pair-buffer overflow: the image never shows a partially sorted frame.
    More passes than needed sort all the same."""
    s = _setup(renderer, "c1")
    W, Hh = s["W"], s["H"]
    ref_d, st = orc.render(s["ocu"], s["osu"], s["tex"], s["draws"], W, Hh, order_mode=1)
    img = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH)
    t = renderer.timings()
    assert st["n_visible"] > 10000 and t["n_visible"] == st["n_visible"] and t["n_pairs"] == st["n_pairs16"]
    assert H.max_abs_diff(img, ref_d) <= TOL
    try:
        renderer.set_option(L.GSWT_OPT_DEPTH_SORT, 1)
        l0, g0, _ = renderer.depth_stats()
        for passes in (1, 2, 4):
            renderer.set_option(L.GSWT_OPT_DEPTH_PASSES, passes)
            img_p = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH)
            assert np.array_equal(img, img_p), passes
        l1, g1, _ = renderer.depth_stats()
        assert l1 == l0 and g1 - g0 >= 3
    finally:
        renderer.set_option(L.GSWT_OPT_DEPTH_PASSES, 3)
        renderer.set_option(L.GSWT_OPT_DEPTH_SORT, 0)


@pytest.mark.parametrize("depth_sort", [0, 1])
def test_depth_order_survives_a_pair_overflow_with_and_without_the_graph(renderer, depth_sort):
    """The pair buffers (and in depth order the tile-id payload buffers beside them) grow when a frame outgrows them; the overflowed frame is
    re-run by the fence / wait.  With the capacity pinned far below the frame's pair count the depth-ordered image must still be the
    complete one, launch by launch and as a graph."""
    s = _setup(renderer, "c1")
    W, Hh = s["W"], s["H"]
    want = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH)
    n_pairs = renderer.timings()["n_pairs"]
    assert n_pairs > 10000
    try:
        renderer.set_option(L.GSWT_OPT_DEPTH_SORT, depth_sort)
        for graph in (0, 1):
            renderer.set_option(L.GSWT_OPT_GRAPH, graph)
            renderer.set_option(L.GSWT_OPT_TIMING, 0 if graph else 2)
            renderer.set_option(L.GSWT_OPT_PAIR_CAP, 256)         # the next frame overflows; the capacity then grows
            img = renderer.render(s["cu"], s["su"], W, Hh, order_mode=L.GSWT_ORDER_DEPTH)
            assert renderer.timings()["n_pairs"] == n_pairs
            assert np.array_equal(img, want), graph
    finally:
        renderer.set_option(L.GSWT_OPT_PAIR_CAP, 0)
        renderer.set_option(L.GSWT_OPT_GRAPH, 0)
        renderer.set_option(L.GSWT_OPT_TIMING, 2)
        renderer.set_option(L.GSWT_OPT_DEPTH_SORT, 0)
