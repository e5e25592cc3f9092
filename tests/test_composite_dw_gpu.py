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


# ---- GSWT_OPT_COMPOSITE = 2: k_composite<FOLD> -- no k_combine launch: the last segment of a tile to finish folds the partials (agent-scope
# stores / loads + a ticket per tile), tiles without pairs are (empty) work items, workgroup 0 publishes the counters.  The fold order is
# k_combine's, so the image is bit-identical WHICHEVER workgroup finishes last: any lost update across the XCDs' L2 caches would show here.
def _variant(renderer, v, fn):
    renderer.set_option(L.GSWT_OPT_COMPOSITE, v)
    try:
        return fn()
    finally:
        renderer.set_option(L.GSWT_OPT_COMPOSITE, 0)


@pytest.mark.parametrize("name,seg", [("c3", 1536), ("c3", 256), ("c3d", 512), ("c5", 512)])
def test_folded_combine_bit_identical_at_baseline_size(renderer, name, seg):
    import bench
    import torch
    w, wang, cu, vp, sort = bench.build_workload(name)
    W, Hh = w["width"], w["height"]
    su = wang.scene_uniforms()
    wang.upload_to(renderer)
    renderer.configure(None)
    renderer.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
    renderer.set_option(L.GSWT_OPT_SEGMENT, seg)
    out = torch.empty((Hh, W, 4), dtype=torch.float32, device="cuda")
    try:
        for eps in (0.0, 1e-5):
            def frame():
                out.fill_(-1.0)
                torch.cuda.synchronize()           # (torch's stream: nothing orders its fill against the frame slot's stream)
                renderer.render_wait(renderer.render_async(cu, su, W, Hh, out.data_ptr(), transmittance_eps=eps))
                return out.cpu().numpy().copy(), renderer.timings()
            (a, ta) = _variant(renderer, 0, frame)
            assert a[..., 3].max() > 0.5 and a.min() >= 0.0
            for rep in range(6):                         # the finishing order of the segments differs from launch to launch
                (b, tb) = _variant(renderer, 2, frame)
                assert np.array_equal(a, b), (name, seg, eps, rep, float(np.abs(a - b).max()))
                assert (tb["n_visible"], tb["n_pairs"]) == (ta["n_visible"], ta["n_pairs"])
    finally:
        renderer.set_option(L.GSWT_OPT_SEGMENT, L.GSWT_DEFAULT_SEGMENT)


@pytest.mark.parametrize("kw", [dict(bg=True), dict(bg=True, t_eps=1e-4), dict(render_config=dict(draw_mode=1)), dict(shard=3), dict(shard=2, shard_cols=True),
                                dict(order_mode=1, t_eps=1e-5)])
def test_folded_combine_variants(renderer, kw):
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    cam = ((4.2, 1.0, 1.2), (5.0, 3.0, 0.9))
    a, ref, _, _ = _variant(renderer, 0, lambda: _run_case(renderer, cfg, cam, 320, 240, lod0=2500, **kw))
    renderer.set_option(L.GSWT_OPT_SEGMENT, 256)         # many multi-segment tiles
    try:
        b, _, _, _ = _variant(renderer, 2, lambda: _run_case(renderer, cfg, cam, 320, 240, lod0=2500, **kw))
        c, _, _, _ = _variant(renderer, 0, lambda: _run_case(renderer, cfg, cam, 320, 240, lod0=2500, **kw))
    finally:
        renderer.set_option(L.GSWT_OPT_SEGMENT, L.GSWT_DEFAULT_SEGMENT)
    assert np.array_equal(b, c)                          # same segment length: bit-identical with and without k_combine
    assert np.abs(b.astype(np.float64) - ref).max() <= 1e-4 + kw.get("t_eps", 0.0)
    assert np.abs(a.astype(np.float64) - b).max() <= 2e-6 + kw.get("t_eps", 0.0)      # another segment length regroups the fold


@pytest.mark.parametrize("name,seg", [("c3", 1536), ("c3", 256), ("c3d", 4096)])
def test_heaviest_first_item_order_bit_identical(renderer, name, seg):
    """GSWT_OPT_ITEM_ORDER = 1: k_items lists the compositor's work items by falling length (full segments first) instead of in tile order;
    a segment's partial still goes to its tile's slot + its number, so the image is the same bit for bit under all three compositors, with
    and without the early-out, in both orders of the pairs."""
    import bench
    w, wang, cu, vp, sort = bench.build_workload(name)
    W, Hh = w["width"], w["height"]
    su = wang.scene_uniforms()
    wang.upload_to(renderer)
    renderer.configure(None)
    renderer.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
    renderer.set_option(L.GSWT_OPT_SEGMENT, seg)
    try:
        for variant in (0, 1, 2):
            renderer.set_option(L.GSWT_OPT_COMPOSITE, variant)
            for kw in (dict(), dict(transmittance_eps=1e-5), dict(order_mode=L.GSWT_ORDER_DEPTH)):
                renderer.set_option(L.GSWT_OPT_ITEM_ORDER, 0)
                a = renderer.render(cu, su, W, Hh, **kw)
                renderer.set_option(L.GSWT_OPT_ITEM_ORDER, 1)
                b = renderer.render(cu, su, W, Hh, **kw)
                assert a[..., 3].max() > 0.5
                assert np.array_equal(a, b), (name, seg, variant, kw, float(np.abs(a - b).max()))
    finally:
        renderer.set_option(L.GSWT_OPT_SEGMENT, L.GSWT_DEFAULT_SEGMENT)
        renderer.set_option(L.GSWT_OPT_COMPOSITE, 0)
        renderer.set_option(L.GSWT_OPT_ITEM_ORDER, 0)
