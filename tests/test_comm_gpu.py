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


def test_frames_overlap_the_previous_frames_gather():
    """frame i + 1 is submitted before frame i is gathered, so its kernels run while frame i's fence, peer copies and
    re-assembly are in flight.  Three ranks (contexts) on one device, c3-sized column bands, three frames in flight per rank, event
    timestamps from the device (gswt_debug_frame_times): on every rank frame i + 1 STARTS before gather i has finished, for every i,
    and the gathered frames are the unsharded renders bit for bit."""
    import torch
    import bench
    from gswt_renderer_amd import host, workloads
    w, wang, cu0, vp0, sort = bench.build_workload("c3")
    W, Hh = w["width"], w["height"]
    su = wang.scene_uniforms()
    cam = workloads.camera_for("c3")
    n, n_frames = 3, 6
    cams = [host.camera_uniforms((cam["pos"][0] + 0.2 * k, cam["pos"][1] + 0.3 * k, cam["pos"][2]),
                                 (cam["target"][0] + 0.2 * k, cam["target"][1] + 0.3 * k, cam["target"][2]), cam["up"], cam["fovy"], cam["near"], cam["far"], W, Hh)[0]
            for k in range(n_frames)]
    rs = [GSWTRenderer(0) for _ in range(n)]
    try:
        for r in rs:
            wang.upload_to(r)
            r.configure(None)
            r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
            r.set_option(L.GSWT_OPT_TIMING, 1)
        wants = [rs[0].render(c, su, W, Hh, transmittance_eps=1e-5) for c in cams]
        GSWTRenderer.group_init(rs)
        bw = rs[0].shard_cols_padded(W, n)
        outs = [[torch.zeros((Hh, bw, 4), dtype=torch.float32, device="cuda") for _ in range(n)] for _ in cams]
        frames = [[torch.zeros((Hh, W, 4), dtype=torch.float32, device="cuda") for _ in range(n)] for _ in cams]
        torch.cuda.synchronize()
        depth = 3                                                    # frames in flight per rank
        tickets = {}

        def submit(i):
            tickets[i] = [r.render_async(cams[i], su, W, Hh, o.data_ptr(), shard=(k, n, "cols"), transmittance_eps=1e-5) for k, (r, o) in enumerate(zip(rs, outs[i]))]

        for i in range(depth):
            submit(i)
        times = {}
        for i in range(n_frames):
            GSWTRenderer.group_render_gather(rs, tickets[i], [f.data_ptr() for f in frames[i]])
            if i + 1 < n_frames:
                # device timeline of rank 0: frame i + 1 (already submitted) against the gather of frame i (just issued); both complete at the sync
                times[i] = rs[0].frame_times(tickets[i][0], tickets[i + 1][0]), rs[0].frame_times(tickets[i][0], tickets[i][0])
            for r, t in zip(rs, tickets[i]):
                r.render_wait(t)
            if i + depth < n_frames:
                submit(i + depth)
        for r in rs:
            r.synchronize()
        for i in range(n_frames):
            for f in frames[i]:
                assert np.array_equal(f.cpu().numpy(), wants[i]), i
        for i, ((s_next, e_next, _), (_, e_this, g_this)) in times.items():
            print(f"frame {i}: kernels end {e_this * 1e3:7.1f} us, gather done {g_this * 1e3:7.1f} us; frame {i + 1}: kernels {s_next * 1e3:7.1f} .. {e_next * 1e3:7.1f} us")
            assert g_this == g_this and s_next < g_this, (i, s_next, g_this)           # frame i + 1 was running before gather i finished
    finally:
        for r in rs:
            r.comm_destroy()
            r.close()
