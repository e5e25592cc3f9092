"""k_composite_dw (GSWT_OPT_COMPOSITE = 1): the compositor with the four waves of a work item decoupled -- 128-pair batches through a ring
of three LDS buffers with ready / consumed counters instead of two workgroup barriers per batch.  Same F3 / F4 / blend order per pixel, so
the image has to be the default compositor's BIT FOR BIT when nothing is cut (transmittance_eps = 0); with the early-out on, a wave stops
accumulating weights below eps at a 128-pair boundary instead of a 256-pair one, so the two images may differ by less than eps (the
per-wave exits are the delicate part of the protocol: a saturated wave keeps staging for the others).  Checked with a background colour
+ depth buffer, debug colours, shards, short and long work items, at c3 and on the dense c3d.  Per-pixel math: /root/reference/src/gswt.wgsl:425-435; blend state renderer.rs:118-129."""
import numpy as np
import pytest

from gswt_renderer_amd import _lib as L
from tests.test_end_to_end_gpu import _run_case

pytestmark = pytest.mark.gpu


def _both(renderer, fn):
    a = fn()
    renderer.set_option(L.GSWT_OPT_COMPOSITE, 1)
    try:
        b = fn()
    finally:
        renderer.set_option(L.GSWT_OPT_COMPOSITE, 0)
    return a, b


@pytest.mark.parametrize("name,seg", [("c3", 1536), ("c3", 256), ("c3d", 4096)])
def test_decoupled_waves_bit_identical_at_baseline_size(renderer, name, seg):
    import bench
    w, wang, cu, vp, sort = bench.build_workload(name)
    W, Hh = w["width"], w["height"]
    su = wang.scene_uniforms()
    wang.upload_to(renderer)
    renderer.configure(None)
    renderer.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
    renderer.set_option(L.GSWT_OPT_SEGMENT, seg)
    try:
        for eps in (0.0, 1e-5):
            a, b = _both(renderer, lambda: renderer.render(cu, su, W, Hh, transmittance_eps=eps))
            assert a[..., 3].max() > 0.5
            if eps == 0.0:
                assert np.array_equal(a, b), (name, seg, float(np.abs(a - b).max()))
            else:
                assert float(np.abs(a - b).max()) <= eps, (name, seg, eps, float(np.abs(a - b).max()))
    finally:
        renderer.set_option(L.GSWT_OPT_SEGMENT, L.GSWT_DEFAULT_SEGMENT)


@pytest.mark.parametrize("kw", [dict(bg=True), dict(bg=True, t_eps=1e-4), dict(render_config=dict(draw_mode=1)), dict(shard=3), dict(shard=2, shard_cols=True),
                                dict(order_mode=1, t_eps=1e-5)])
def test_decoupled_waves_variants(renderer, kw):
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    cam = ((4.2, 1.0, 1.2), (5.0, 3.0, 0.9))
    (a, ref, _, _), (b, _, _, _) = _both(renderer, lambda: _run_case(renderer, cfg, cam, 320, 240, lod0=2500, **kw))
    if kw.get("t_eps", 0.0) == 0.0:
        assert np.array_equal(a, b)
    else:
        assert float(np.abs(a - b).max()) <= kw["t_eps"]
    assert np.abs(a.astype(np.float64) - ref).max() <= 1e-4 + kw.get("t_eps", 0.0)
