"""GSWT_OPT_GRAPH: the frame's launch sequence replayed as one hipGraphLaunch per slot must give bit-identical frames --
across a moving camera (kernel-node argument updates), sort events that change the draw list and its sizes (grid updates),
pair-buffer growth (re-run + rebuilt graph), background / depth variants (another kernel in the chain: rebuilt graph),
column-band shards, and all slots in flight."""
import ctypes as C

import numpy as np
import pytest

from gswt_renderer_amd import _lib as L
from gswt_renderer_amd import host, synth
from gswt_renderer_amd.pipeline import GSWTPipeline

pytestmark = pytest.mark.gpu


def _graph_stats(renderer):
    a = (C.c_ulonglong * 3)()
    assert L.load().gswt_debug_graph_stats(renderer._h, a) == 0
    return list(a)


@pytest.fixture()
def graph_mode(renderer):
    renderer.set_option(L.GSWT_OPT_TIMING, 0)          # frames that carry timing events do not go through the graph
    yield renderer
    renderer.set_option(L.GSWT_OPT_GRAPH, 0)
    renderer.set_option(L.GSWT_OPT_TIMING, 2)
    renderer.set_option(L.GSWT_OPT_PAIR_CAP, 0)


def test_graph_replay_is_bit_identical_over_a_moving_camera_and_sort_events(graph_mode):
    import torch
    renderer = graph_mode
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=700)
    W, Hh = 272, 176
    slots = renderer.frame_slots()
    n = 3 * slots + 2
    frames = []
    for k in range(n):
        pos = (0.3 + 0.9 * k, 0.2 + 0.5 * k, 3.0 - 0.05 * k)
        tgt = (pos[0] + 1.0, pos[1] + 2.0, 2.2)
        frames.append((pos,) + host.camera_uniforms(pos, tgt, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh))

    def run(graph):
        pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer)      # fresh worker state: the same tile maps in both runs
        renderer.set_option(L.GSWT_OPT_GRAPH, 1 if graph else 0)
        outs = [torch.zeros((Hh, W, 4), dtype=torch.float32, device="cuda") for _ in range(n)]
        torch.cuda.synchronize()
        tickets, pairs = [], []
        for k, (pos, cu, vp) in enumerate(frames):
            if k % 3 == 0:
                pipe.update(pos, vp, force_sort=True)         # sort event: new draw list (sizes change with the map shift)
            su = pipe.wang.scene_uniforms()
            tickets.append(renderer.render_async(cu, su, W, Hh, outs[k].data_ptr(), transmittance_eps=1e-5))
            if len(tickets) == slots:
                renderer.render_wait(tickets.pop(0)); pairs.append(renderer.timings()["n_pairs"])
        while tickets:
            renderer.render_wait(tickets.pop(0)); pairs.append(renderer.timings()["n_pairs"])
        torch.cuda.synchronize()
        return [o.cpu().numpy() for o in outs], pairs

    s0 = _graph_stats(renderer)
    want, pairs_want = run(False)
    assert _graph_stats(renderer) == s0                 # option off: nothing goes through a graph
    got, pairs_got = run(True)
    s1 = _graph_stats(renderer)
    assert s1[0] - s0[0] >= n                           # every frame was a graph launch (a re-run after overflow adds one)
    assert 1 <= s1[1] - s0[1] <= 3 * slots              # graphs built: once per slot, rebuilt only when the kernel chain changes
    assert s1[2] - s0[2] >= n                           # the camera moves: kernel nodes were updated in place
    assert pairs_got == pairs_want and max(pairs_want) > 0
    for k in range(n):
        assert np.array_equal(got[k], want[k]), k


def test_graph_replay_with_background_depth_shards_and_buffer_growth(graph_mode):
    renderer = graph_mode
    cfg = dict(tile_map_half_wh=(3, 3), surface_type=0, lod_max_dist=20.0, tile_sort_type=3, merge_type=2)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=900)
    pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer)
    W, Hh = 320, 208
    pos, tgt = (4.2, 1.0, 2.0), (5.0, 3.0, 1.5)
    cu, vp = host.camera_uniforms(pos, tgt, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh)
    pipe.update(pos, vp)
    rng = np.random.default_rng(11)
    bg = rng.uniform(0, 1, size=(Hh, W, 4)).astype(np.float32)
    bgd = rng.uniform(0.97, 1.0, size=(Hh, W)).astype(np.float32)
    variants = [dict(), dict(bg_rgba=bg), dict(bg_rgba=bg, bg_depth=bgd), dict(transmittance_eps=1e-4), dict(shard=(1, 3, "cols")), dict(shard=(2, 3))]
    renderer.set_option(L.GSWT_OPT_GRAPH, 0)
    want = [pipe.render(cu, W, Hh, **v) for v in variants]
    renderer.set_option(L.GSWT_OPT_GRAPH, 1)
    s0 = _graph_stats(renderer)
    for rep in range(2):                                 # second round: the same variants again on graphs that exist
        for v, wimg in zip(variants, want):
            assert np.array_equal(pipe.render(cu, W, Hh, **v), wimg), (rep, sorted(v))
    # pair buffers pinned far too small: the frame overflows, is re-run with grown buffers (other grids: same chain, updated nodes)
    renderer.set_option(L.GSWT_OPT_PAIR_CAP, 256)
    assert np.array_equal(pipe.render(cu, W, Hh), want[0])
    assert renderer.timings()["n_pairs"] > 256
    assert _graph_stats(renderer)[0] - s0[0] >= 2 * len(variants) + 2
