"""End-to-end GPU parity: product host (C++ WangTile) -> draw list -> HIP render, against the
oracle's WangTile restatement -> oracle draw list -> oracle render, same tile set, same camera.
Covers the three draw classes (plain / LOD-blending / merged), HeightMap surface, background
colour + proxy depth, early termination and screen-tile sharding."""
import numpy as np
import pytest

from gswt_renderer_amd import host, synth
from gswt_renderer_amd.pipeline import GSWTPipeline
from oracle import gswt_oracle as orc
from oracle import wangtile_oracle as wo
from tests import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _run_case(renderer, cfg, cam, W, Hh, *, lod0=600, n_lod=3, bg=False, t_eps=0.0, shard=None, culling_dist=1.0, order_mode=0,
              render_config=None, shard_cols=False, stats=None):
    """render_config: RenderConfig fields of SceneUniforms (draw_mode, point_cloud_radius, use_clip, clip_height)."""
    rc = dict(render_config or {})
    verts = synth.make_tileset(n_lod=n_lod, n_tile=16, lod0_count=lod0)
    pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer)
    cu, vp = host.camera_uniforms(cam[0], cam[1], (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh)
    pipe.update(cam[0], vp)
    # oracle side
    pp = orc.preprocess([[orc.scene_load(v) for v in lod] for lod in verts])
    ow = wo.WangTile(pp)
    ou = ow.configure(wo.UserData(**cfg))
    ocam = orc.Camera(W, Hh, cam[0], cam[1], [0, 0, 1])
    osd = ow.build_tiles(cam[0])
    osort = ow.sort_tiles(cam[0], ocam.view_proj())
    odraws = wo.renderer_draws(pp, osort, ocam.view_proj(), culling_dist=culling_dist)
    osu = wo.scene_uniforms_from_data(ou, osd["center_coord"], **rc)
    hm = ou.height_map.reshape(ou.height_map_wh[1], ou.height_map_wh[0]) if ou.surface_type == 1 else None
    bg_rgba = bg_depth = None
    if bg:
        rng = np.random.default_rng(5)
        bg_rgba = rng.uniform(0, 1, size=(Hh, W, 4)).astype(np.float32)
        bg_depth = rng.uniform(0.97, 1.0, size=(Hh, W)).astype(np.float32)
    ref, st = orc.render(ocam.uniforms(), osu, pp.tex, odraws, W, Hh, height_map=hm, bg_rgba=bg_rgba, bg_depth=bg_depth,
                         order_mode=order_mode)
    kinds = {"plain": 0, "blend": 0, "merged": 0}
    for d in odraws:
        kinds["merged" if d.tile.single_draw else ("blend" if d.tile.changing else "plain")] += 1
    if shard is None:
        img = pipe.render(cu, W, Hh, bg_rgba=bg_rgba, bg_depth=bg_depth, transmittance_eps=t_eps, culling_dist=culling_dist,
                          order_mode=order_mode, **rc)
        t = renderer.timings()
        assert t["n_visible"] == st["n_visible"]
        assert t["n_pairs"] == st["n_pairs16"]
    elif shard_cols:
        n = shard
        bw = renderer.shard_cols_padded(W, n)
        img = np.zeros((Hh, W, 4), dtype=np.float32)
        for r in range(n):
            part = pipe.render(cu, W, Hh, bg_rgba=bg_rgba, bg_depth=bg_depth, shard=(r, n, "cols"), culling_dist=culling_dist,
                               order_mode=order_mode, transmittance_eps=t_eps, **rc)
            assert part.shape == (Hh, bw, 4)
            x0, x1 = r * bw, min(W, (r + 1) * bw)
            if x1 > x0:
                img[:, x0:x1] = part[:, :x1 - x0]
                assert not part[:, x1 - x0:].any()                  # padding columns stay zero
            if stats is not None:
                stats.append(renderer.timings()["n_visible"])
    else:
        n = shard
        rows_p = renderer.shard_rows_padded(Hh, n)
        img = np.zeros((Hh, W, 4), dtype=np.float32)
        for r in range(n):
            part = pipe.render(cu, W, Hh, bg_rgba=bg_rgba, bg_depth=bg_depth, shard=(r, n), culling_dist=culling_dist,
                               order_mode=order_mode, **rc)
            assert part.shape == (rows_p, W, 4)
            for y in range(Hh):
                ty = y // 16
                if ty % n == r:
                    img[y] = part[(ty // n) * 16 + (y % 16)]
    return img, ref, kinds, st


def test_graph_edge_none_surface(renderer):
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    img, ref, kinds, st = _run_case(renderer, cfg, ((4.2, 1.0, 3.0), (5.0, 3.0, 2.5)), 384, 256)
    assert kinds["plain"] > 0 and kinds["blend"] > 0 and kinds["merged"] > 0, kinds
    assert st["n_visible"] > 500
    assert H.max_abs_diff(img, ref) <= TOL


def test_more_than_8192_screen_tiles(renderer):
    """2304 x 1296 = 144 x 81 = 11 664 screen tiles: k_items runs as two workgroups, the second summing the segments of the
    first one's tiles itself; a small segment size gives many tiles several work items (k_combine)."""
    from gswt_renderer_amd import _lib as L
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    renderer.set_option(L.GSWT_OPT_SEGMENT, 256)
    try:
        img, ref, kinds, st = _run_case(renderer, cfg, ((4.2, 1.0, 3.0), (5.0, 3.0, 2.5)), 2304, 1296, bg=True)
    finally:
        renderer.set_option(L.GSWT_OPT_SEGMENT, L.GSWT_DEFAULT_SEGMENT)
    assert st["n_visible"] > 500
    assert H.max_abs_diff(img, ref) <= TOL


def test_more_than_65536_screen_tiles(renderer):
    """5120x3472 = 320 x 217 = 69 440 screen tiles: 17 tile bits = three passes of the tile-bit sort, and tile ids above 2^16 take the plain
    division in tile_xy (k_composite / k_combine map ids below 2^16 to (column, row) by multiply-high)."""
    cfg = dict(tile_map_half_wh=(2, 2), surface_type=0, lod_max_dist=40.0, tile_sort_type=3, merge_type=2)
    cam = ((2.0, -9.0, 7.0), (2.0, 2.0, 0.0))
    img, ref, _, st = _run_case(renderer, cfg, cam, 5120, 3472, lod0=500)
    assert st["n_visible"] > 500
    assert img[3280:, :, 3].max() > 0.05                  # splats in the rows whose tiles have ids >= 65 600
    assert H.max_abs_diff(img, ref) <= TOL


def test_heightmap_surface_with_background_and_depth(renderer):
    cfg = dict(tile_map_half_wh=(3, 4), surface_type=1, lod_max_dist=24.0, tile_sort_type=3, merge_type=2,
               height_map_wh=(4, 4), height_map_scale=(1.0, 1.0, 0.3))
    img, ref, kinds, st = _run_case(renderer, cfg, ((0.5, 0.3, 5.0), (1.0, 1.0, 4.5)), 320, 240, bg=True)
    assert st["n_visible"] > 300
    assert H.max_abs_diff(img, ref) <= TOL


def test_distance_sort_no_blending(renderer):
    cfg = dict(tile_map_half_wh=(2, 2), surface_type=0, lod_max_dist=16.0, tile_sort_type=0, merge_type=2, lod_blending=False)
    img, ref, kinds, st = _run_case(renderer, cfg, ((0, 0, 5.0), (0, 1, 5.0)), 320, 240)
    assert H.max_abs_diff(img, ref) <= TOL


def test_early_termination_within_tolerance(renderer):
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    img, ref, kinds, st = _run_case(renderer, cfg, ((4.2, 1.0, 1.0), (5.0, 3.0, 0.8)), 320, 240, lod0=3000, t_eps=1e-5)
    assert H.max_abs_diff(img, ref) <= TOL


@pytest.mark.parametrize("n", [2, 3, 8])
def test_shard_union_equals_unsharded(renderer, n):
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    cam = ((4.2, 1.0, 3.0), (5.0, 3.0, 2.5))
    full, ref, _, _ = _run_case(renderer, cfg, cam, 200, 150)
    img, _, _, _ = _run_case(renderer, cfg, cam, 200, 150, shard=n)
    assert np.array_equal(img, full)          # bitwise
    assert H.max_abs_diff(img, ref) <= TOL


def test_moving_camera_sequence(renderer):
    """Map shift + LRU-cached merged lists over a short path (wangtile.rs:575-593,1682-1720)."""
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=400)
    pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer)
    pp = orc.preprocess([[orc.scene_load(v) for v in lod] for lod in verts])
    ow = wo.WangTile(pp)
    ou = ow.configure(wo.UserData(**cfg))
    W, Hh = 256, 160
    osd = None
    for k in range(5):
        pos = (0.3 + 1.7 * k, 0.2 + 0.9 * k, 3.0)
        tgt = (pos[0] + 1.0, pos[1] + 2.0, 2.4)
        cu, vp = host.camera_uniforms(pos, tgt, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh)
        pipe.update(pos, vp, force_sort=True)
        ocam = orc.Camera(W, Hh, pos, tgt, [0, 0, 1])
        if ow.check_update(pos):
            osd = ow.build_tiles(pos)
        osort = ow.sort_tiles(pos, ocam.view_proj())
        odraws = wo.renderer_draws(pp, osort, ocam.view_proj())
        ref, st = orc.render(ocam.uniforms(), wo.scene_uniforms_from_data(ou, osd["center_coord"]), pp.tex, odraws, W, Hh)
        img = pipe.render(cu, W, Hh)
        assert renderer.timings()["n_visible"] == st["n_visible"]
        assert H.max_abs_diff(img, ref) <= TOL, k


def test_depth_order_mode(renderer):
    """GSWT_ORDER_DEPTH (true global per-splat depth sort) against the oracle's depth-order mode, and
    it really differs from the reference order on interpenetrating tiles."""
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    cam = ((4.2, 1.0, 1.2), (5.0, 3.0, 0.9))
    img_d, ref_d, _, st = _run_case(renderer, cfg, cam, 320, 240, lod0=2500, order_mode=1)
    assert st["n_visible"] > 2000
    assert H.max_abs_diff(img_d, ref_d) <= TOL
    img_r, ref_r, _, _ = _run_case(renderer, cfg, cam, 320, 240, lod0=2500, order_mode=0)
    assert H.max_abs_diff(img_r, ref_r) <= TOL
    assert H.max_abs_diff(ref_d, ref_r) > 1e-3


def test_depth_order_sharded_and_background(renderer):
    cfg = dict(tile_map_half_wh=(2, 2), surface_type=0, lod_max_dist=16.0, tile_sort_type=3, merge_type=2)
    cam = ((0.5, 0.3, 2.0), (1.0, 3.0, 1.2))
    full, ref, _, _ = _run_case(renderer, cfg, cam, 200, 150, lod0=1500, order_mode=1, bg=True)
    assert H.max_abs_diff(full, ref) <= TOL
    img, _, _, _ = _run_case(renderer, cfg, cam, 200, 150, lod0=1500, order_mode=1, bg=True, shard=3)
    assert np.array_equal(img, full)


def test_async_pipelined_frames_match_sync(renderer):
    """gswt_render_async / gswt_render_wait: three frames queued back to back (two in flight) give exactly
    the images of the synchronous call, including a frame that forces the pair buffers to grow."""
    import torch
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=800)
    pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer)
    W, Hh = 256, 160
    pos = (4.2, 1.0, 2.0)
    cams = [host.camera_uniforms(pos, t, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh) for t in ((5.0, 3.0, 1.5), (3.0, 4.0, 1.0), (6.0, 1.5, 1.8))]
    pipe.update(pos, cams[0][1])
    su = pipe.wang.scene_uniforms()
    want = [renderer.render(cu, su, W, Hh) for cu, _ in cams]
    outs = [torch.zeros((Hh, W, 4), dtype=torch.float32, device="cuda") for _ in cams]
    torch.cuda.synchronize()          # the zero fills ran on torch's stream, the frames run on the ctx's slot streams
    tickets = []
    for (cu, _), o in zip(cams, outs):
        tickets.append(renderer.render_async(cu, su, W, Hh, o.data_ptr()))
        if len(tickets) == 2:
            renderer.render_wait(tickets.pop(0))
    while tickets:
        renderer.render_wait(tickets.pop(0))
    for o, wimg in zip(outs, want):
        assert np.array_equal(o.cpu().numpy(), wimg)
    assert renderer.timings()["n_pairs"] > 0


def test_async_all_slots_in_flight_and_slot_reuse(renderer):
    """Every frame slot in flight at once, then more frames than slots without a wait in between (the library collects
    the oldest frame itself and reuses its slot), then waits in submission order: every image equals the synchronous one."""
    import torch
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=700)
    pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer)
    W, Hh = 240, 160
    pos = (4.2, 1.0, 2.0)
    slots = renderer.frame_slots()
    assert slots >= 2
    n = 2 * slots + 1
    tgts = [(5.0 - 0.3 * k, 3.0 + 0.2 * k, 1.5 - 0.05 * k) for k in range(n)]
    cams = [host.camera_uniforms(pos, t, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh) for t in tgts]
    pipe.update(pos, cams[0][1])
    su = pipe.wang.scene_uniforms()
    want = [renderer.render(cu, su, W, Hh) for cu, _ in cams]
    outs = [torch.zeros((Hh, W, 4), dtype=torch.float32, device="cuda") for _ in cams]
    torch.cuda.synchronize()
    tickets = [renderer.render_async(cu, su, W, Hh, o.data_ptr()) for (cu, _), o in zip(cams, outs)]
    assert set(tickets) == set(range(slots))              # tickets are slot indices; the slots were reused
    for t in tickets[-slots:]:                            # only the last `slots` frames are still in flight
        renderer.render_wait(t)
    torch.cuda.synchronize()
    for o, wimg in zip(outs, want):
        assert np.array_equal(o.cpu().numpy(), wimg)


def test_device_side_merged_lists_bit_exact(renderer):
    """gswt_set_draws_merge_groups: the merged-group lists built on the GPU (segmented stable radix sort on
    (group, 16-bit depth bucket)) equal the host's Scene::sort_raw_depth_vec lists entry for entry, and the
    rendered image is unchanged."""
    from gswt_renderer_amd import _lib as L
    cfg = dict(tile_map_half_wh=(4, 4), surface_type=0, lod_max_dist=22.0, tile_sort_type=3, merge_type=2, merge_topk=40)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=900)
    W, Hh = 320, 200
    for pos, tgt in (((4.2, 1.0, 1.5), (5.0, 4.0, 1.0)), ((-3.0, 2.5, 2.5), (-1.0, 7.0, 1.0))):
        cu, vp = host.camera_uniforms(pos, tgt, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh)
        ph = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer)
        ph.update(pos, vp)
        s = ph.sort
        assert any(t.merged for t in s.tiles) and len(s.groups) >= 1
        img_host = ph.render(cu, W, Hh)
        want_list = s.merged_gs_index.copy()
        has_lod = np.zeros(len(want_list), dtype=bool)
        for t in s.tiles:
            if t.merged and t.single_lod_id == -1:
                has_lod[t.merged_offset:t.merged_offset + t.merged_count] = True
        want_map = s.merged_map_id.copy()
        want_lod = s.merged_lod_id.copy()
        pd = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer, device_merge=True)
        pd.update(pos, vp)
        assert [t.map_index for t in pd.sort.tiles] == [t.map_index for t in s.tiles]
        got_packed, got_map = renderer.read_merged()
        assert got_packed.shape == want_list.shape
        assert np.array_equal(got_packed & ((1 << 28) - 1), want_list)
        assert np.array_equal(got_map, want_map)
        assert np.array_equal((got_packed >> 28)[has_lod], want_lod[has_lod])
        img_dev = pd.render(cu, W, Hh)
        assert np.array_equal(img_dev, img_host)


@pytest.mark.parametrize("mode", [1, 2, 3, 4])
def test_debug_draw_modes(renderer, mode):
    """draw_mode 1..4 (TileID / TileLOD / LOD / View recolouring, gswt.wgsl:268-399) on a frame with plain, blending
    and merged draws; mode 1 hashes the merged head's offset through the canonical sin (rand(), :502-512)."""
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    img, ref, kinds, st = _run_case(renderer, cfg, ((4.2, 1.0, 3.0), (5.0, 3.0, 2.5)), 320, 208, render_config=dict(draw_mode=mode))
    assert kinds["plain"] > 0 and kinds["blend"] > 0 and kinds["merged"] > 0, kinds
    assert H.max_abs_diff(img, ref) <= TOL
    plain, _, _, _ = _run_case(renderer, cfg, ((4.2, 1.0, 3.0), (5.0, 3.0, 2.5)), 320, 208)
    assert H.max_abs_diff(img, plain) > 0.05           # the mode really recolours


def test_debug_draw_mode_with_point_cloud_and_depth(renderer):
    """point-cloud covariance scales with 2^lod when a debug mode is on (gswt.wgsl:182-192); float colours with a depth buffer."""
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    img, ref, kinds, st = _run_case(renderer, cfg, ((4.2, 1.0, 3.0), (5.0, 3.0, 2.5)), 256, 160, bg=True,
                                    render_config=dict(draw_mode=3, point_cloud_radius=0.002))
    assert H.max_abs_diff(img, ref) <= TOL


@pytest.mark.parametrize("cam", [((3.0, -19.0, 6.0), (0.0, 0.0, 0.0)), ((-14.0, 9.0, -8.0), (0.0, 0.0, 1.0))])
def test_sphere_surface_end_to_end(renderer, cam):
    """Sphere surface through the whole path: sphere topology + CPU mapping in the worker (tile centres, corners, LOD
    rings, Edge merging across block seams), sphere mapping per splat on the device."""
    cfg = dict(tile_map_half_wh=(5, 2), surface_type=2, sphere_radius=6.5, lod_max_dist=60.0, tile_sort_type=3, merge_type=2)
    img, ref, kinds, st = _run_case(renderer, cfg, cam, 320, 240, lod0=700)
    assert kinds["plain"] > 0 and kinds["blend"] + kinds["merged"] > 0, kinds
    assert st["n_visible"] > 3000
    assert H.max_abs_diff(img, ref) <= TOL


@pytest.mark.parametrize("n", [2, 3, 8])
def test_column_band_shards_equal_unsharded(renderer, n):
    """GSWT_SHARD_COLUMNS: contiguous tile-column bands with per-rank draw culling (bounds of every draw's splat centres +
    a conservative splat radius).  The union of the bands is the unsharded image BITWISE -- a draw dropped wrongly, or a
    pair emitted for the wrong band, would show -- and the culling really removes work."""
    cfg = dict(tile_map_half_wh=(4, 4), surface_type=0, lod_max_dist=24.0, tile_sort_type=3, merge_type=2)
    for cam, kw in ((((4.2, 1.0, 3.0), (5.0, 3.0, 2.5)), {}), (((-6.0, -2.0, 1.2), (2.0, 6.0, 0.6)), dict(bg=True)),
                    (((0.5, 0.3, 6.0), (1.5, 4.0, 0.0)), dict(render_config=dict(draw_mode=3)))):
        full, ref, kinds, st = _run_case(renderer, cfg, cam, 360, 200, lod0=900, **kw)
        assert H.max_abs_diff(full, ref) <= TOL
        vis = []
        img, _, _, _ = _run_case(renderer, cfg, cam, 360, 200, lod0=900, shard=n, shard_cols=True, stats=vis, **kw)
        assert np.array_equal(img, full), (cam, n)
        assert sum(vis) > 0 and min(vis) < st["n_visible"]          # some rank projected fewer splats than the whole frame has
    # the HeightMap surface (the reference's default, structure.rs:75) with the band cull: a cell's splats are bounded through the height
    # map's extremes and slopes; a steep map and a camera close to the ground make the bound work
    for hscale, cam in (((1.0, 1.0, 0.3), ((0.5, 0.3, 5.0), (1.0, 1.0, 4.5))), ((0.7, 1.3, 1.5), ((-5.0, -3.0, 2.2), (2.0, 6.0, 0.4))),
                        ((1.0, 1.0, -0.8), ((4.2, 1.0, 3.0), (5.0, 3.0, 2.5)))):
        cfgh = dict(tile_map_half_wh=(4, 4), surface_type=1, lod_max_dist=24.0, tile_sort_type=3, merge_type=2,
                    height_map_type=4, height_map_wh=(6, 6), height_map_scale=hscale)
        full, ref, _, st = _run_case(renderer, cfgh, cam, 360, 200, lod0=900)
        assert H.max_abs_diff(full, ref) <= TOL
        vis = []
        img, _, _, _ = _run_case(renderer, cfgh, cam, 360, 200, lod0=900, shard=n, shard_cols=True, stats=vis)
        assert np.array_equal(img, full), (hscale, cam, n)
        assert sum(vis) > 0
        if n >= 3:                                                  # the inflated boxes still let some band drop something
            assert min(vis) < st["n_visible"], (vis, st)
    # the Sphere surface: a cell's centres are bounded through the Lipschitz constant of the strip parametrisation (sphere_cell_box);
    # whole sphere in view, the north pole under the camera, the south pole from close by.  (The splat-radius bound carries |F|^2 <=
    # 2 (2.5 R / block_w)^2 + 1 and the coarsest LOD's covariance: at 320 px it only lets go of cells with splat_scale < 1.)
    # (splat_scale, camera, smallest number of bands at which some band must project less than the whole frame; None: union only)
    for ss, cam, cull_from in ((0.3, ((3.0, -19.0, 6.0), (0.0, 0.0, 0.0)), 3), (0.3, ((9.0, -30.0, 8.0), (0.0, 0.0, 0.0)), 3),
                               (0.2, ((0.6, 0.4, 16.0), (0.0, 0.0, 0.0)), 8), (0.3, ((0.6, 0.4, 11.0), (0.0, 0.0, 0.0)), None),
                               (0.25, ((2.0, -3.0, -9.5), (0.0, 0.0, -6.5)), None), (1.0, ((3.0, -19.0, 6.0), (0.0, 0.0, 0.0)), None)):
        cfgs = dict(tile_map_half_wh=(5, 2), surface_type=2, sphere_radius=6.5, lod_max_dist=60.0, tile_sort_type=3, merge_type=2)
        rc = dict(splat_scale=ss)
        full, ref, _, st = _run_case(renderer, cfgs, cam, 320, 240, lod0=700, render_config=rc)
        assert H.max_abs_diff(full, ref) <= TOL
        vis = []
        img, _, _, _ = _run_case(renderer, cfgs, cam, 320, 240, lod0=700, shard=n, shard_cols=True, stats=vis, render_config=rc)
        assert np.array_equal(img, full), (ss, cam, n)
        assert sum(vis) > 0
        if cull_from is not None and n >= cull_from:
            assert min(vis) < st["n_visible"], (ss, cam, vis, st)


def test_column_bands_follow_a_moving_camera(renderer):
    """Column-band frames over a camera path that shifts the tile map (new center_coord -> merged members move -> the draw
    bounds of the band cull must be rebuilt) and re-sorts the draws every step: band union == unsharded, bitwise, each step."""
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=400)
    pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer)
    W, Hh, n = 272, 160, 3
    bw = renderer.shard_cols_padded(W, n)
    shifted = 0
    last_center = None
    for k in range(6):
        pos = (0.3 + 2.3 * k, 0.2 + 1.1 * k, 3.0)
        tgt = (pos[0] + 1.0, pos[1] + 2.0, 2.4)
        cu, vp = host.camera_uniforms(pos, tgt, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh)
        pipe.update(pos, vp, force_sort=True)
        center = tuple(pipe.wang.scene_uniforms().center_coord[:])
        shifted += int(last_center is not None and center != last_center)
        last_center = center
        full = pipe.render(cu, W, Hh)
        img = np.zeros_like(full)
        for r in range(n):
            part = pipe.render(cu, W, Hh, shard=(r, n, "cols"))
            x0, x1 = r * bw, min(W, (r + 1) * bw)
            img[:, x0:x1] = part[:, :x1 - x0]
        assert np.array_equal(img, full), k
        assert full[..., 3].max() > 0.2
    assert shifted >= 2


def test_merged_group_reuse_across_sort_events(renderer):
    """gswt_set_draws_merge_groups copies the merged groups that did not change since the previous sort event (the reference's
    LRU hit, wangtile.rs:575-593) instead of re-sorting them: along a camera path the device lists and the images must equal
    the ones built with every group re-sorted, entry for entry, and some groups must actually have been reused."""
    from gswt_renderer_amd import _lib as L
    cfg = dict(tile_map_half_wh=(4, 4), surface_type=0, lod_max_dist=22.0, tile_sort_type=3, merge_type=2, merge_topk=40)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=900)
    W, Hh = 320, 200
    path = [((4.2 + 0.35 * k, 1.0 + 0.5 * k, 1.5), (5.0 + 0.3 * k, 4.0 + 0.55 * k, 1.0)) for k in range(9)]
    results = {}
    for reuse in (False, True):
        renderer.set_option(L.GSWT_OPT_NO_MERGE_REUSE, 0 if reuse else 1)
        pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer, device_merge=True)
        built0, reused0 = renderer.merge_stats()
        out = []
        for pos, tgt in path:
            cu, vp = host.camera_uniforms(pos, tgt, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh)
            pipe.update(pos, vp, force_sort=True)
            img = pipe.render(cu, W, Hh)
            lst, mp = renderer.read_merged()
            out.append((img, lst.copy(), mp.copy()))
        built, reused = renderer.merge_stats()
        results[reuse] = (out, built - built0, reused - reused0)
    renderer.set_option(L.GSWT_OPT_NO_MERGE_REUSE, 0)
    assert results[False][2] == 0 and results[True][2] > 0                       # groups were reused
    assert results[True][1] + results[True][2] == results[False][1]              # ... instead of being sorted
    for (img_a, l_a, m_a), (img_b, l_b, m_b) in zip(results[False][0], results[True][0]):
        assert np.array_equal(l_a, l_b) and np.array_equal(m_a, m_b)
        assert np.array_equal(img_a, img_b)


def test_merged_group_reuse_reaches_back_several_events(renderer):
    """the lists of the last 9 sort events stay addressable by (view, member tile ids, transition states), so a camera
    that oscillates between three places re-sorts nothing once each place has been seen -- the reference's 1 024-entry LRU of merged
    lists (wangtile.rs:427,575-593); with the previous event alone every return trip sorted again.  Lists and images equal the ones
    built with every group re-sorted."""
    from gswt_renderer_amd import _lib as L
    cfg = dict(tile_map_half_wh=(4, 4), surface_type=0, lod_max_dist=22.0, tile_sort_type=3, merge_type=2, merge_topk=40)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=900)
    W, Hh = 320, 200
    # one place, three viewing directions: the tile map (and with it the tile ids of the groups' members) stays, the merged groups and their
    # presort views change with the direction
    places = [((4.2, 1.0, 1.5), (5.0, 4.0, 1.0)), ((4.3, 1.1, 1.5), (8.0, 0.5, 1.0)), ((4.1, 0.9, 1.5), (1.0, 3.5, 1.0))]
    path = [places[k % 3] for k in range(12)]
    results = {}
    for reuse in (False, True):
        renderer.set_option(L.GSWT_OPT_NO_MERGE_REUSE, 0 if reuse else 1)
        pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer, device_merge=True)
        built0, reused0 = renderer.merge_stats()
        deep0 = renderer.merge_stats_deep()
        out, built_after_first_lap = [], None
        for k, (pos, tgt) in enumerate(path):
            cu, vp = host.camera_uniforms(pos, tgt, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh)
            pipe.update(pos, vp, force_sort=True)
            img = pipe.render(cu, W, Hh)
            lst, mp = renderer.read_merged()
            out.append((img, lst.copy(), mp.copy()))
            if k == 2:
                built_after_first_lap = renderer.merge_stats()[0] - built0
        built, reused = renderer.merge_stats()
        results[reuse] = (out, built - built0, reused - reused0, renderer.merge_stats_deep() - deep0, built_after_first_lap)
    renderer.set_option(L.GSWT_OPT_NO_MERGE_REUSE, 0)
    out_a, built_a, reused_a, deep_a, _ = results[False]
    out_b, built_b, reused_b, deep_b, first_lap_b = results[True]
    assert reused_a == 0 and deep_a == 0
    assert first_lap_b > 0 and built_b == first_lap_b            # after the first lap nothing is sorted any more
    assert deep_b > 0 and built_b + reused_b == built_a          # ... the return trips copy from events two and three back
    for (img_a, l_a, m_a), (img_b, l_b, m_b) in zip(out_a, out_b):
        assert np.array_equal(l_a, l_b) and np.array_equal(m_a, m_b)
        assert np.array_equal(img_a, img_b)


def test_merged_group_reuse_does_not_survive_a_scene_upload(renderer):
    """the merged lists the draw sets retain as copy sources belong to ONE scene.  Two tile sets of the same shape (same counts,
    other splats: another seed) uploaded one after the other on one ctx -- the second scene's sort events must not copy lists sorted
    for the first (they match by view, member tile ids and length): lists and images equal the ones built with every group re-sorted."""
    from gswt_renderer_amd import _lib as L
    cfg = dict(tile_map_half_wh=(4, 4), surface_type=0, lod_max_dist=22.0, tile_sort_type=3, merge_type=2, merge_topk=40)
    W, Hh = 320, 200
    path = [((4.2 + 0.35 * k, 1.0 + 0.5 * k, 1.5), (5.0 + 0.3 * k, 4.0 + 0.55 * k, 1.0)) for k in range(4)]

    def run(seed, reuse):
        renderer.set_option(L.GSWT_OPT_NO_MERGE_REUSE, 0 if reuse else 1)
        verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=900, seed_offset=seed)
        pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer, device_merge=True)      # uploads the scene + its raw depths
        out = []
        for pos, tgt in path:
            cu, vp = host.camera_uniforms(pos, tgt, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh)
            pipe.update(pos, vp, force_sort=True)
            img = pipe.render(cu, W, Hh)
            lst, mp = renderer.read_merged()
            out.append((img, lst.copy(), mp.copy()))
        return out

    try:
        want = run(7, False)                 # scene B alone, every group sorted
        run(0, True)                         # scene A fills the retained draw sets ...
        got = run(7, True)                   # ... then scene B of the same shape on the same ctx, reuse on
    finally:
        renderer.set_option(L.GSWT_OPT_NO_MERGE_REUSE, 0)
    assert not np.array_equal(run(0, True)[0][1], want[0][1])                    # the two scenes really sort differently
    for (img_a, l_a, m_a), (img_b, l_b, m_b) in zip(want, got):
        assert np.array_equal(l_a, l_b) and np.array_equal(m_a, m_b)
        assert np.array_equal(img_a, img_b)


def test_one_row_map_with_edge_merging(renderer):
    """a (2 N + 1) x 1 map (map_half_wh[1] == 0): the merged-member offset of gswt.wgsl:52-63 divides the map id by a map height
    of ONE -- the multiply-high shortcut of k_project has no magic number for that (0xFFFFFFFF / 1 + 1 wraps) and must divide."""
    cfg = dict(tile_map_half_wh=(4, 0), surface_type=0, lod_max_dist=30.0, tile_sort_type=0, merge_type=2, merge_topk=40, merge_dot_threshold=1.0)
    cam = ((2.0, 0.5, 7.0), (2.0, 2.0, 0.0))             # from above: the three middle cells (map ids 3, 4, 5) form one merged group
    img, ref, kinds, st = _run_case(renderer, cfg, cam, 320, 240, lod0=1500)
    assert kinds["merged"] >= 1 and st["n_visible"] > 1000, (kinds, st)
    assert H.max_abs_diff(img, ref) <= TOL


def test_deferred_swap_in_takes_effect_once_built(renderer):
    """GSWT_OPT_DEFER_SWAP: a sort event is read by the first frame submitted after its device-side build has finished; until then
    frames keep the previous draw list, and the next event makes a still-pending one current.  After gswt_synchronize (which
    also waits for the build stream) every event must have landed, in order, with the images of the immediate mode."""
    from gswt_renderer_amd import _lib as L
    cfg = dict(tile_map_half_wh=(4, 4), surface_type=0, lod_max_dist=22.0, tile_sort_type=3, merge_type=2, merge_topk=40)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=900)
    W, Hh = 320, 200
    path = [((4.2 + 0.35 * k, 1.0 + 0.5 * k, 1.5), (5.0 + 0.3 * k, 4.0 + 0.55 * k, 1.0)) for k in range(7)]
    cams = [host.camera_uniforms(pos, tgt, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh) for pos, tgt in path]
    cu_fix = cams[3][0]                                  # one camera for every image: only the draw list changes
    ref = []
    renderer.set_option(L.GSWT_OPT_DEFER_SWAP, 0)
    pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer, device_merge=True)
    for (pos, _), (cu, vp) in zip(path, cams):
        pipe.update(pos, vp, force_sort=True)
        ref.append(pipe.render(cu_fix, W, Hh))
    pipe.update(path[1][0], cams[1][1], force_sort=True)
    pipe.update(path[5][0], cams[5][1], force_sort=True)
    ref_bb = pipe.render(cu_fix, W, Hh)
    assert any(not np.array_equal(ref[0], r) for r in ref[1:])          # the events do change the image
    renderer.set_option(L.GSWT_OPT_DEFER_SWAP, 1)
    pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer, device_merge=True)
    try:
        for k, ((pos, _), (cu, vp)) in enumerate(zip(path, cams)):
            pipe.update(pos, vp, force_sort=True)
            early = pipe.render(cu_fix, W, Hh)          # may still be the previous event's list (never anything else)
            assert np.array_equal(early, ref[k]) or (k > 0 and np.array_equal(early, ref[k - 1])), k
            renderer.synchronize()
            assert np.array_equal(pipe.render(cu_fix, W, Hh), ref[k]), k
        # two events back to back: the second makes the first current, then lands itself
        pipe.update(path[1][0], cams[1][1], force_sort=True)
        pipe.update(path[5][0], cams[5][1], force_sort=True)
        renderer.synchronize()
        assert np.array_equal(pipe.render(cu_fix, W, Hh), ref_bb)
        # value n >= 2: the n-th frame submitted after the call reads the new list, whatever the device is doing
        renderer.set_option(L.GSWT_OPT_DEFER_SWAP, 3)
        pipe.update(path[2][0], cams[2][1], force_sort=True)
        before = pipe.render(cu_fix, W, Hh)
        assert np.array_equal(before, ref_bb) and np.array_equal(pipe.render(cu_fix, W, Hh), ref_bb)
        third = pipe.render(cu_fix, W, Hh)
        assert not np.array_equal(third, ref_bb)
        renderer.set_option(L.GSWT_OPT_DEFER_SWAP, 0)
        pipe.update(path[2][0], cams[2][1], force_sort=True)
        assert np.array_equal(pipe.render(cu_fix, W, Hh), third)
    finally:
        renderer.set_option(L.GSWT_OPT_DEFER_SWAP, 0)


def test_first_frames_of_fresh_contexts_fill_every_slot():
    """The FIRST frame of every frame slot of a new context, all in flight at once (first use allocates and clears the slot's
    buffers: a clear that is not ordered on the slot's non-blocking stream can land after the frame's own kernels -- the slot's
    first frame then came back as background).  Several fresh contexts, each compared with its own synchronous frames."""
    import torch
    from gswt_renderer_amd.renderer import GSWTRenderer
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=700)
    W, Hh = 240, 160
    pos = (4.2, 1.0, 2.0)
    for rep in range(4):
        r = GSWTRenderer(0)
        try:
            pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=r)
            slots = r.frame_slots()
            tgts = [(5.0 - 0.3 * k, 3.0 + 0.2 * k, 1.5 - 0.05 * k) for k in range(slots)]
            cams = [host.camera_uniforms(pos, t, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh) for t in tgts]
            pipe.update(pos, cams[0][1])
            su = pipe.wang.scene_uniforms()
            outs = [torch.zeros((Hh, W, 4), dtype=torch.float32, device="cuda") for _ in cams]
            torch.cuda.synchronize()
            tickets = [r.render_async(cu, su, W, Hh, o.data_ptr()) for (cu, _), o in zip(cams, outs)]      # every slot's first frame
            assert sorted(tickets) == list(range(slots))
            for t in tickets:
                r.render_wait(t)
            torch.cuda.synchronize()
            got = [o.cpu().numpy() for o in outs]
            want = [r.render(cu, su, W, Hh) for cu, _ in cams]
            for k in range(slots):
                assert got[k].any(), (rep, k)
                assert np.array_equal(got[k], want[k]), (rep, k)
        finally:
            r.close()
