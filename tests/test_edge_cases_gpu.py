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
    for seg in (256, 512, 2048):
        renderer.set_option(L.GSWT_OPT_SEGMENT, seg)
        img = renderer.render(cam, su, W, Hh)
        assert renderer.timings()["n_pairs"] == st["n_pairs16"] > 100000
        assert H.max_abs_diff(img, ref) <= TOL
        img = renderer.render(cam, su, W, Hh, transmittance_eps=1e-5)
        assert H.max_abs_diff(img, ref) <= TOL
    renderer.set_option(L.GSWT_OPT_SEGMENT, 512)


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
    # the ctx is still usable afterwards
    img = renderer.render(cam, orc.scene_uniforms(num_lod=2, map_half_wh=(1, 2)), 64, 48)
    assert img.shape == (48, 64, 4)
