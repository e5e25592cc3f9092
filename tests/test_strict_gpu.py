"""The HIP path against the oracle's STRICT mode (the shader text operator by operator, oracle/gswt_oracle.c) at c3 and on the
reference's default HeightMap surface (c3h), with bench.py's early-out threshold 1e-5: the whole error stack in one place.

  term                                              bound used here        measured (c3 / c3h)
  early-out cut (transmittance_eps)                 1.0e-5                 <= 1.0e-5
  blend order, exp2 / log2, colour scaling          1e-6                   5e-7
  v2 vs strict, continuous part (unmarked pixels)   1e-3, <= 1e-4 of the   2.8e-4 / 4.6e-4 at 3 / 52 pixels, below 1e-4
                                                    pixels above 1e-4      elsewhere
  v2 vs strict, flipped |p|^2 <= 4 decisions        alpha e^-4 = 0.018     719 / 775 of 2 073 600 pixels

Round 4: the product's DEFAULT vertex stage is the strict one (k_project<., ., STRICT>: +1 us of 71 at c3); the second test below shows
that the default image then meets 1e-4 (+ the early-out cut) against the strict image on every pixel that holds no flipped coverage
decision, at c3, c3h and c5.  The first test keeps the numbers of the v2 option (GSWT_OPT_STRICT_VS = 0), whose thin-ellipse term the
table above lists.  /root/reference/src/gswt.wgsl:152-258,402-435."""
import numpy as np
import pytest

from gswt_renderer_amd import _lib as L
from oracle import gswt_oracle as orc
from tests.test_strict_oracle import both_modes, check_bounds

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["c3", "c3h", "c5"])
def test_gpu_v2_option_vs_strict_full_error_stack(renderer, name):
    import bench
    import torch
    d = both_modes(name, varyings=False)
    w, wang, cu, vp, sort = bench.build_workload(name)
    W, H = w["width"], w["height"]
    su = wang.scene_uniforms()
    wang.upload_to(renderer)
    renderer.configure(wang.height_map() if int(wang.user.surface_type) == 1 else None)
    renderer.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    renderer.set_option(L.GSWT_OPT_STRICT_VS, 0)           # this test is about the rounding sequence v2 (the default until round 3)
    try:
        renderer.render_wait(renderer.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5))
        t = renderer.timings()
    finally:
        renderer.set_option(L.GSWT_OPT_STRICT_VS, 1)
    img = out.cpu().numpy()
    del out
    torch.cuda.empty_cache()
    assert t["n_visible"] == d["st2"]["n_visible"] == d["sts"]["n_visible"]
    # against v2: the product's contract
    assert np.abs(img.astype(np.float64) - d["v2"]).max() <= 1e-4
    # against strict: flipped decisions apart, the continuous differences of v2 + the early-out cut + the blend order
    # (c5: 4K, 18 M visible splats, up to ~40 splats deep per pixel at the horizon -- more decisions to flip and more thin ellipses per
    # pixel: measured 7 970 marked pixels of 8.29 M and 1.5e-3 on the unmarked ones)
    lim = dict(flip_frac=2e-3, cont_max=3e-3, cont_over_frac=1e-3) if name == "c5" else {}
    r = check_bounds(d, img, d["strict"], extra=1.1e-5, **lim)
    print(f"{name}: GPU (eps 1e-5) vs strict L-inf {r['linf']:.3e}, unmarked {r['linf_unmarked']:.3e}, "
          f"{r['over_1e4_unmarked']} unmarked pixels above 1e-4, {r['marked']} marked")


@pytest.mark.parametrize("name", ["c3", "c3h", "c5"])
def test_gpu_strict_vertex_stage_meets_1e4_off_the_ellipse_borders(renderer, name):
    """The default (GSWT_OPT_STRICT_VS = 1 since round 4): with the vertex stage evaluated as the shader text writes it, what is left between the HIP image and the strict
    image is (a) the fragment stage's |p|^2 <= 4 decisions that fall the other way (inherent to two rasterisations: F4 evaluates the
    inverse affine map tile-locally in binary32, the strict image interpolates the quad exactly), and (b) off those pixels the
    continuous terms: the early-out cut, blend order, exp2 / log2.  (b) meets north_star's 1e-4 (+ the 1e-5 cut) at c3, c3h AND c5 --
    the thin-ellipse term of the default sequence v2 (2.8e-4 / 4.6e-4 / 1.5e-3) came from the vertex stage."""
    import bench
    import torch
    w, wang, cu, vp, sort = bench.build_workload(name)
    W, H = w["width"], w["height"]
    su = wang.scene_uniforms()
    hm = wang.height_map() if int(wang.user.surface_type) == 1 else None
    tex, draws = bench.oracle_draws(wang, sort, vp)
    ocu = orc.Camera176.from_buffer_copy(bytes(cu))
    osu = orc.Scene160.from_buffer_copy(bytes(su))
    with orc.strict():
        strict_img, sts = orc.render(ocu, osu, tex, draws, W, H, height_map=hm)
    mixed_img, stm = orc.render(ocu, osu, tex, draws, W, H, height_map=hm)          # the checker's default: strict vertex stage + F1..F4
    mask, counts = orc.compare_modes(ocu, osu, tex, draws, W, H, height_map=hm, strict_vs=True)
    wang.upload_to(renderer)
    renderer.configure(hm)
    renderer.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    renderer.render_wait(renderer.render_async(cu, su, W, H, out.data_ptr(), transmittance_eps=1e-5))
    t = renderer.timings()
    img = out.cpu().numpy()
    del out
    torch.cuda.empty_cache()
    assert t["n_visible"] == sts["n_visible"] == stm["n_visible"] and t["n_pairs"] == stm["n_pairs16"]
    assert counts["visible_in_one_mode"] == 0
    # the product's contract in this mode: the same operation sequence on both sides (strict vertex stage, F1..F4)
    assert np.abs(img.astype(np.float64) - mixed_img).max() <= 1e-4
    diff = np.abs(img.astype(np.float64) - strict_img).max(axis=2)
    n = diff.size
    off = diff[~mask]
    over = int((off > 1e-4 + 1.1e-5).sum())
    print(f"{name}: GPU (strict VS, eps 1e-5) vs strict image: L-inf {diff.max():.3e}; {int(mask.sum())} of {n} pixels hold a flipped coverage "
          f"decision ({counts['decision_flips']} decisions); elsewhere L-inf {off.max():.3e}, {over} pixels above 1e-4 + cut")
    assert mask.sum() <= 2e-3 * n
    assert off.max() <= 1e-4 + 1.1e-5, off.max()
    if mask.any():
        assert diff[mask].max() <= 2 * float(np.exp(-4.0)) + 1e-3
