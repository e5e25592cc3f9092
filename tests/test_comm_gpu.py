"""The framebuffer all-gather behind the C ABI: RCCL communicator with one rank (all a one-GPU box can
run: RCCL refuses two ranks on one device) and the hipMemcpyPeerAsync group with several contexts on one device.  In both
the gathered frame must equal the unsharded render bit for bit."""
import numpy as np
import pytest

from gswt_renderer_amd import _lib as L
from gswt_renderer_amd.renderer import GSWTRenderer, GSWTError
from oracle import gswt_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


def _scene(r, pp):
    r.upload_scene(pp.tex, pp.gs_index, pp.gs_lod_id)
    r.configure(None)
    case = H.grid_case(pp)
    r.set_draws(case.draws)


def test_rccl_communicator_single_rank(renderer):
    import torch
    pp = H.tileset()
    _scene(renderer, pp)
    W, Hh = 200, 120
    cam = orc.default_camera(W, Hh).uniforms()
    su = orc.scene_uniforms(num_lod=pp.n_lod)
    want = renderer.render(cam, su, W, Hh)
    uid = GSWTRenderer.comm_unique_id()
    assert len(uid) == L.GSWT_COMM_ID_BYTES
    renderer.comm_init(uid, 0, 1)
    try:
        with pytest.raises(GSWTError):
            renderer.comm_init(uid, 0, 1)                       # one communicator per ctx
        out = torch.zeros((Hh, W, 4), dtype=torch.float32, device="cuda")
        frame = torch.zeros_like(out)
        torch.cuda.synchronize()
        t = renderer.render_async(cam, su, W, Hh, out.data_ptr())
        renderer.render_gather(t, frame.data_ptr())
        renderer.render_wait(t)
        renderer.synchronize()
        assert np.array_equal(frame.cpu().numpy(), want)
        # a frame rendered as a shard of a different world is refused
        t = renderer.render_async(cam, su, W, Hh, out.data_ptr(), shard=(0, 2, "cols"))
        with pytest.raises(GSWTError):
            renderer.render_gather(t, frame.data_ptr())
        renderer.render_wait(t)
    finally:
        renderer.comm_destroy()


@pytest.mark.parametrize("mode", ["cols", "rows"])
def test_peer_copy_group_three_ranks_on_one_device(mode):
    import torch
    pp = H.tileset()
    n = 3
    rs = [GSWTRenderer(0) for _ in range(n)]
    try:
        for r in rs:
            _scene(r, pp)
        W, Hh = 200, 120
        cam = orc.default_camera(W, Hh).uniforms()
        su = orc.scene_uniforms(num_lod=pp.n_lod)
        want = rs[0].render(cam, su, W, Hh)
        GSWTRenderer.group_init(rs)
        if mode == "cols":
            shard_shape = (Hh, rs[0].shard_cols_padded(W, n), 4)
        else:
            shard_shape = (rs[0].shard_rows_padded(Hh, n), W, 4)
        for rep in range(2):                                        # twice: the group's buffers and events are reused
            outs = [torch.zeros(shard_shape, dtype=torch.float32, device="cuda") for _ in range(n)]
            frames = [torch.zeros((Hh, W, 4), dtype=torch.float32, device="cuda") for _ in range(n)]
            torch.cuda.synchronize()
            tickets = [r.render_async(cam, su, W, Hh, o.data_ptr(), shard=(k, n, "cols") if mode == "cols" else (k, n))
                       for k, (r, o) in enumerate(zip(rs, outs))]
            GSWTRenderer.group_render_gather(rs, tickets, [f.data_ptr() for f in frames])
            for r, t in zip(rs, tickets):
                r.render_wait(t)
                r.synchronize()
            for f in frames:
                assert np.array_equal(f.cpu().numpy(), want)
    finally:
        for r in rs:
            r.comm_destroy()
            r.close()


def test_back_to_back_group_gathers_without_a_sync():
    """Two gathers of DIFFERENT frames issued back to back (frames in flight, no synchronize between them): the second gather's
    pushes into a peer's gather buffer must wait for that peer's re-assembly of the first (write-after-read across
    streams).  Each gathered frame must be its own camera's unsharded render, bit for bit."""
    import torch
    pp = H.tileset()
    n = 3
    rs = [GSWTRenderer(0) for _ in range(n)]
    try:
        for r in rs:
            _scene(r, pp)
        W, Hh = 640, 360
        cams = [orc.default_camera(W, Hh).uniforms(), orc.Camera(W, Hh, (0.5, -1.0, 4.0), (1.0, 3.0, 1.0), [0, 0, 1]).uniforms()]
        su = orc.scene_uniforms(num_lod=pp.n_lod)
        wants = [rs[0].render(c, su, W, Hh) for c in cams]
        assert not np.array_equal(wants[0], wants[1])
        GSWTRenderer.group_init(rs)
        shard_shape = (Hh, rs[0].shard_cols_padded(W, n), 4)
        for rep in range(3):
            outs = [[torch.zeros(shard_shape, dtype=torch.float32, device="cuda") for _ in range(n)] for _ in cams]
            frames = [[torch.zeros((Hh, W, 4), dtype=torch.float32, device="cuda") for _ in range(n)] for _ in cams]
            torch.cuda.synchronize()
            tickets = []
            for ci, cam in enumerate(cams):                         # both frames in flight on every rank before any gather
                tickets.append([r.render_async(cam, su, W, Hh, o.data_ptr(), shard=(k, n, "cols")) for k, (r, o) in enumerate(zip(rs, outs[ci]))])
            for ci in range(len(cams)):                             # ... and the gathers one behind the other, no host sync
                GSWTRenderer.group_render_gather(rs, tickets[ci], [f.data_ptr() for f in frames[ci]])
            for ci in range(len(cams)):
                for r, t in zip(rs, tickets[ci]):
                    r.render_wait(t)
            for r in rs:
                r.synchronize()
            for ci in range(len(cams)):
                for f in frames[ci]:
                    assert np.array_equal(f.cpu().numpy(), wants[ci])
    finally:
        for r in rs:
            r.comm_destroy()
            r.close()
