"""Seeded random sweep: random cameras x surfaces x orderings x draw modes through the whole path
(host worker -> draw list -> HIP render) against the oracle.  The discontinuous decisions (culls, |p|^2 <= 4, depth
test, LOD drops) must agree everywhere, so one flipped pixel fails the 1e-4 bound."""
import os

import numpy as np
import pytest

from tests import helpers as H
from tests.test_end_to_end_gpu import _run_case

pytestmark = pytest.mark.gpu
TOL = 1e-4

# GSWT_SWEEP_SEED / GSWT_SWEEP_CASES: a longer one-off sweep on the GPU box (the defaults are what the suite runs)
_RNG = np.random.default_rng(int(os.environ.get("GSWT_SWEEP_SEED", "20261004")))
_N_CASES = int(os.environ.get("GSWT_SWEEP_CASES", "14"))


def _cases():
    out = []
    for k in range(_N_CASES):
        surface = (0, 1, 2, 0, 1, 0, 1)[k % 7]
        sort_t = (3, 3, 3, 0, 2, 1, 3)[k % 7]
        merge = (2, 2, 2, 2, 2, 1, 0)[k % 7]
        if surface == 2:
            ang, el = _RNG.uniform(0, 2 * np.pi), _RNG.uniform(-1.0, 1.0)
            dist = _RNG.uniform(14.0, 26.0)
            pos = (dist * np.cos(el) * np.cos(ang), dist * np.cos(el) * np.sin(ang), dist * np.sin(el))
            tgt = tuple(_RNG.uniform(-2.0, 2.0, 3))
            cfg = dict(tile_map_half_wh=(5, 2), surface_type=2, sphere_radius=float(_RNG.uniform(5.0, 8.0)), lod_max_dist=60.0,
                       tile_sort_type=3, merge_type=2)
        else:
            pos = (float(_RNG.uniform(-6, 6)), float(_RNG.uniform(-6, 6)), float(_RNG.uniform(0.8, 7.0)))
            yaw = _RNG.uniform(0, 2 * np.pi)
            tgt = (pos[0] + 4.0 * np.cos(yaw), pos[1] + 4.0 * np.sin(yaw), pos[2] - float(_RNG.uniform(0.2, 3.0)))
            cfg = dict(tile_map_half_wh=(3, 3), surface_type=surface, lod_max_dist=float(_RNG.uniform(14.0, 26.0)), tile_sort_type=sort_t,
                       merge_type=merge)
            if surface == 1:
                cfg.update(height_map_wh=(6, 5), height_map_scale=(1.0, 1.0, float(_RNG.uniform(0.1, 0.5))))
            if merge == 1:
                cfg.update(merge_tile_dist=(1, 3))
            if sort_t in (0, 1, 2) and merge != 2:
                cfg.update(tile_sort_type=3)            # corner data only exists for Graph sort or Edge merge (renderer.rs:476)
        rc = {}
        if k % 5 == 4:
            rc["draw_mode"] = int(_RNG.integers(1, 5))
        if k % 6 == 5:
            rc["point_cloud_radius"] = 0.003
        # round 4: the order mode (reference / global depth sort) and the compositor variant (k_composite + k_combine, decoupled waves,
        # folded combine) rotate through the cases too
        mode = dict(order_mode=int(k % 4 == 1), composite=(0, 1, 2)[k % 3], depth_sort=(0, 1)[(k // 4) % 2], item_order=(k // 2) % 2)
        out.append(pytest.param(cfg, (tuple(float(x) for x in pos), tuple(float(x) for x in tgt)), rc, bool(k % 3 == 0), mode, id=f"case{k}"))
    return out


@pytest.mark.parametrize("cfg,cam,rc,bg,mode", _cases())
def test_random_sweep(renderer, cfg, cam, rc, bg, mode):
    from gswt_renderer_amd import _lib as L
    W, Hh = (272, 176)
    t_eps = 1e-5 if cfg["surface_type"] == 0 else 0.0
    renderer.set_option(L.GSWT_OPT_COMPOSITE, mode["composite"])
    renderer.set_option(L.GSWT_OPT_DEPTH_SORT, mode["depth_sort"])     # (depth-ordered cases: global passes / tile-local LDS sort)
    renderer.set_option(L.GSWT_OPT_ITEM_ORDER, mode["item_order"])
    try:
        with np.errstate(all="ignore"):
            img, ref, kinds, st = _run_case(renderer, cfg, cam, W, Hh, lod0=500, bg=bg, render_config=rc, t_eps=t_eps, order_mode=mode["order_mode"])
    finally:
        renderer.set_option(L.GSWT_OPT_COMPOSITE, 0)
        renderer.set_option(L.GSWT_OPT_DEPTH_SORT, 0)
        renderer.set_option(L.GSWT_OPT_ITEM_ORDER, 0)
    assert H.max_abs_diff(img, ref) <= TOL + t_eps, (cfg, cam, rc, mode, kinds, st)
