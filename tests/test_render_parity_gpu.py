"""GPU parity: HIP hot path (through the C ABI) vs the CPU oracle on identical inputs.
Float tolerance stated by BASELINE.json north_star: 1e-4 per-channel L-inf on the float
framebuffer.  Per-splat vertex-stage outputs are compared bit-for-bit."""
import numpy as np
import pytest

from tests import helpers as H
from oracle import gswt_oracle as orc
from gswt_renderer_amd import _lib as L

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _scene(pp, half=(1, 2), **kw):
    return orc.scene_uniforms(num_lod=pp.n_lod, map_half_wh=half, **kw)


def test_plain_tiles_image_parity(renderer):
    pp = H.tileset()
    W, Hh = 320, 240
    cam = orc.default_camera(W, Hh).uniforms()
    su = _scene(pp)
    case = H.grid_case(pp)
    case.upload(renderer)
    ref, st = orc.render(cam, su, pp.tex, case.orc_draws, W, Hh)
    img = renderer.render(cam, su, W, Hh)
    t = renderer.timings()
    assert t["n_visible"] == st["n_visible"]
    assert t["n_pairs"] == st["n_pairs16"]
    assert H.max_abs_diff(img, ref) <= TOL
    assert ref[..., 3].max() > 0.5      # the case actually draws something


def test_vertex_stage_bit_exact(renderer):
    pp = H.tileset()
    W, Hh = 320, 240
    cam = orc.default_camera(W, Hh).uniforms()
    su = _scene(pp)
    case = H.grid_case(pp)
    renderer.set_option(L.GSWT_OPT_NO_LOD_PREFILTER, 1)
    renderer.set_option(L.GSWT_OPT_DEBUG_VARYINGS, 1)
    try:
        case.upload(renderer)
        renderer.render(cam, su, W, Hh)
        got = renderer.read_projected()
    finally:
        renderer.set_option(L.GSWT_OPT_NO_LOD_PREFILTER, 0)
        renderer.set_option(L.GSWT_OPT_DEBUG_VARYINGS, 0)
    want = orc.project_draws(cam, su, pp.tex, case.orc_draws)
    assert got.shape == want.shape
    assert np.array_equal(got["visible"], want["visible"])
    vis = want["visible"] == 1
    assert vis.sum() > 1000
    for fld in ("ndc", "depth", "major", "minor", "rgba"):
        a = got[fld][vis].view(np.uint32)
        b = want[fld][vis].view(np.uint32)
        assert np.array_equal(a, b), fld


@pytest.mark.parametrize("surface", ["plane", "hmap", "sphere"])
def test_both_vertex_stage_sequences_bit_exact(renderer, surface):
    """k_project<., ., STRICT> (the default) evaluates gswt.wgsl:152-258,260-265,402-419 operator by operator, k_project<., ., false>
    (GSWT_OPT_STRICT_VS = 0) the rounding sequence v2; each per splat bit for bit what the CPU checker computes in the same mode
    (oracle/gswt_oracle.c project_impl), on all three surfaces, with a non-unit scene scale (the full scene_scale_mat products) and
    debug colours on the sphere."""
    pp = H.tileset()
    W, Hh = 320, 240
    hm = None
    if surface == "sphere":
        su = orc.scene_uniforms(num_lod=pp.n_lod, map_half_wh=(5, 2), surface_type=2, sphere_radius=6.5, draw_mode=1)
        case = _sphere_case(pp)
        cam = orc.Camera(W, Hh, (3.0, -19.0, 6.0), (0.0, 0.0, 0.0), [0, 0, 1]).uniforms()
    else:
        kw = dict(scene_scale=(0.9, 1.2, 1.3))       # (the default camera sees the ground from y = 12 on: y must not shrink)
        if surface == "hmap":
            rng = np.random.default_rng(5)
            hm = rng.random((16, 16), dtype=np.float32)
            kw.update(surface_type=1, height_map_scale=(1.0, 1.0, 0.4))
        su = _scene(pp, **kw)
        case = H.grid_case(pp)
        cam = orc.default_camera(W, Hh).uniforms()
    wants = {}
    for strict_vs in (1, 0):
        renderer.set_option(L.GSWT_OPT_NO_LOD_PREFILTER, 1)
        renderer.set_option(L.GSWT_OPT_DEBUG_VARYINGS, 1)
        renderer.set_option(L.GSWT_OPT_STRICT_VS, strict_vs)
        try:
            renderer.configure(hm)
            case.upload(renderer)
            img = renderer.render(cam, su, W, Hh)
            got = renderer.read_projected()
            t = renderer.timings()
        finally:
            renderer.set_option(L.GSWT_OPT_NO_LOD_PREFILTER, 0)
            renderer.set_option(L.GSWT_OPT_DEBUG_VARYINGS, 0)
            renderer.set_option(L.GSWT_OPT_STRICT_VS, 1)
            renderer.configure(None)
        with (orc.strict(fragment=False) if strict_vs else orc.v2()):
            want = orc.project_draws(cam, su, pp.tex, case.orc_draws, height_map=hm)
            ref, st = orc.render(cam, su, pp.tex, case.orc_draws, W, Hh, height_map=hm)
        wants[strict_vs] = want
        assert np.array_equal(got["visible"], want["visible"])
        vis = want["visible"] == 1
        assert vis.sum() > 1000
        for fld in ("ndc", "depth", "major", "minor", "rgba"):
            assert np.array_equal(got[fld][vis].view(np.uint32), want[fld][vis].view(np.uint32)), (strict_vs, fld)
        # image: the same vertex stage + the compositor's F1..F4 on both sides
        assert t["n_visible"] == st["n_visible"] and t["n_pairs"] == st["n_pairs16"]
        assert H.max_abs_diff(img, ref) <= TOL
    # ... and the two sequences really differ (else this test would pass with the option ignored)
    vis = (wants[0]["visible"] == 1) & (wants[1]["visible"] == 1)
    assert not np.array_equal(wants[1]["major"][vis].view(np.uint32), wants[0]["major"][vis].view(np.uint32))


def _sphere_case(pp, half=(5, 2), tile_width=4.0, merged_cells=((2, 1), (3, 1))):
    """Every cell of a 10 x 4 sphere map (icosahedral strip, 5 x 2 blocks) as a plain draw, except
    `merged_cells`, which form one merged (single_draw) draw with per-splat map ids."""
    case = H.Case(pp)
    w, h = 2 * half[0], 2 * half[1]
    for mx in range(w):
        for my in range(h):
            if (mx, my) in merged_cells:
                continue
            tile = (mx * 5 + my * 3) % pp.n_tile
            case.add_static(lod=0, tile=tile, view=8, offset=((mx - half[0]) * tile_width, (my - half[1]) * tile_width, 0.0),
                            valid_lod_id=0, map_index=mx * h + my, map_coord=(mx, my))
    members = [(mx * h + my, 0, (mx * 5 + my * 3) % pp.n_tile, None) for mx, my in merged_cells]
    case.add_merged(members=members, view=8, head_lod=0, head_tile=members[0][2], head_map_index=members[0][0])
    return case


@pytest.mark.parametrize("draw_mode", [0, 1])
def test_sphere_surface_vertex_stage_and_image(renderer, draw_mode):
    """surface_type 2 (gswt.wgsl:600-623): sphere_get_uv / sphere_uv_to_pos through the canonical sin / cos,
    plain and merged (map-id) draws; per-splat outputs bit-exact, image within 1e-4."""
    pp = H.tileset()
    W, Hh = 320, 240
    R = 6.5
    su = orc.scene_uniforms(num_lod=pp.n_lod, map_half_wh=(5, 2), surface_type=2, sphere_radius=R, draw_mode=draw_mode)
    case = _sphere_case(pp)
    for pos, tgt in (((3.0, -19.0, 6.0), (0.0, 0.0, 0.0)), ((-14.0, 9.0, -8.0), (0.0, 0.0, 1.0))):
        cam = orc.Camera(W, Hh, pos, tgt, [0, 0, 1]).uniforms()
        renderer.set_option(L.GSWT_OPT_NO_LOD_PREFILTER, 1)
        renderer.set_option(L.GSWT_OPT_DEBUG_VARYINGS, 1)
        try:
            case.upload(renderer)
            img = renderer.render(cam, su, W, Hh)
            got = renderer.read_projected()
        finally:
            renderer.set_option(L.GSWT_OPT_NO_LOD_PREFILTER, 0)
            renderer.set_option(L.GSWT_OPT_DEBUG_VARYINGS, 0)
        want = orc.project_draws(cam, su, pp.tex, case.orc_draws)
        assert np.array_equal(got["visible"], want["visible"])
        vis = want["visible"] == 1
        assert vis.sum() > 2000
        for fld in ("ndc", "depth", "major", "minor", "rgba"):
            assert np.array_equal(got[fld][vis].view(np.uint32), want[fld][vis].view(np.uint32)), fld
        ref, st = orc.render(cam, su, pp.tex, case.orc_draws, W, Hh)
        assert renderer.timings()["n_visible"] == st["n_visible"]
        assert H.max_abs_diff(img, ref) <= TOL
        assert ref[..., 3].max() > 0.5
