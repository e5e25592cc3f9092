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
