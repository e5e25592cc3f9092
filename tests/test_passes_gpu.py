"""Background passes (skybox, proxy) as compute kernels vs the CPU oracle's restatement, and the full pass chain of
State::render (state.rs:384-402): skybox -> proxy -> splats composited over the proxy colour with the proxy depth test.
The reference rasterises these passes on a WebGPU device that does not exist here: parity is against our own oracle
("parity unpinned")."""
import numpy as np
import pytest

from gswt_renderer_amd import host, synth
from gswt_renderer_amd.pipeline import GSWTPipeline
from oracle import gswt_oracle as orc
from oracle import wangtile_oracle as wo
from tests import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _cube(n=48):
    """A cube map that is a smooth function of direction (continuous across the seams)."""
    faces = np.zeros((6, n, n, 4), np.float32)
    t, s = np.meshgrid((np.arange(n) + .5) / n * 2 - 1, (np.arange(n) + .5) / n * 2 - 1, indexing="ij")
    one = np.ones_like(s)
    dirs = {0: (one, -t, -s), 1: (-one, -t, s), 2: (s, one, t), 3: (s, -one, -t), 4: (s, -t, one), 5: (-s, -t, -one)}
    for f in range(6):
        d = np.stack(dirs[f], -1)
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        faces[f, ..., :3] = 0.5 + 0.4 * d + 0.1 * np.sin(7.0 * d[..., ::-1])
        faces[f, ..., 3] = 1.0
    return faces


def _mips(ts=64):
    yy, xx = np.mgrid[0:ts, 0:ts]
    cur = np.zeros((ts, ts, 4), np.float32)
    cur[..., 0] = ((xx // 8 + yy // 8) % 2) * 0.8 + 0.1
    cur[..., 1] = xx / ts
    cur[..., 2] = yy / ts
    cur[..., 3] = 1.0
    out = []
    while True:
        out.append(cur.copy())
        if cur.shape[0] == 1:
            return out
        cur = cur.reshape(cur.shape[0] // 2, 2, cur.shape[1] // 2, 2, 4).mean((1, 3)).astype(np.float32)


@pytest.mark.parametrize("equi", [0, 1])
def test_skybox_pass(renderer, equi):
    import torch
    W, Hh = 333, 201
    faces = _cube()
    renderer.skybox_configure(faces, bool(equi))
    for pos, tgt in (((0.5, 0.3, 5.0), (1.0, 6.0, 2.5)), ((0.0, 0.0, 1.0), (0.3, -0.2, 9.0)), ((2.0, 2.0, 2.0), (-5.0, 1.0, 1.5))):
        cam = orc.Camera(W, Hh, pos, tgt, [0, 0, 1])
        out = torch.zeros((Hh, W, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        renderer.skybox_render(cam.uniforms(), W, Hh, out.data_ptr())
        renderer.synchronize()
        ref = orc.skybox_render(cam, faces, W, Hh, equi)
        assert H.max_abs_diff(out.cpu().numpy(), ref) <= 1e-5
        assert ref[..., :3].std() > 0.05 and np.all(ref[..., 3] == 1.0)


@pytest.mark.parametrize("surface", [0, 1])
def test_proxy_pass_two_draws(renderer, surface):
    """proxy_full (GRID_DIM grid) then proxy_map (tile-map grid) into the same colour / depth targets: depth bit-exact
    (same canonical float sequence on both sides), colour within 1e-4, with clip plane and black-background variants."""
    import torch
    W, Hh = 320, 208
    mips = _mips()
    hm = np.random.default_rng(0).uniform(-1, 1, (8, 8)).astype(np.float32)
    renderer.configure(hm if surface == 1 else None)
    renderer.proxy_configure(mips, grid_dim=48)
    sky = np.random.default_rng(1).uniform(0, 1, (Hh, W, 4)).astype(np.float32)
    for pos, tgt, extra in (((0.5, 0.3, 5.0), (1.0, 6.0, 2.5), {}), ((-3.0, 2.0, 1.2), (4.0, 9.0, 0.2), dict(use_clip=1, clip_height=-0.1)),
                            ((1.0, 1.0, 9.0), (1.2, 1.1, 0.0), dict(black_background=1))):
        cam = orc.Camera(W, Hh, pos, tgt, [0, 0, 1])
        rgba = torch.from_numpy(sky.copy()).cuda()
        depth = torch.zeros((Hh, W), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ref_rgba, ref_depth = sky.copy(), np.ones((Hh, W), np.float32)
        common = dict(surface_type=surface, map_half_wh=(3, 4), center_coord=(1, -1), height_map_scale=(1.0, 1.0, 0.6), **extra)
        for k, u in enumerate((orc.proxy_uniforms(cam, map_proxy=0, height_offset=-0.5, width_scale=4.0, **common),
                               orc.proxy_uniforms(cam, map_proxy=1, height_offset=-0.45, **common))):
            renderer.proxy_render(u, W, Hh, rgba.data_ptr(), depth.data_ptr(), clear_depth=(k == 0))
            orc.proxy_render(u, W, Hh, ref_rgba, ref_depth, mips, height_map=hm if surface == 1 else None, grid_dim=48)
        renderer.synchronize()
        got_d, got_c = depth.cpu().numpy(), rgba.cpu().numpy()
        covered = ref_depth < 1.0
        assert 0.2 < covered.mean() <= 1.0
        assert np.array_equal(got_d.view(np.uint32), ref_depth.view(np.uint32))
        assert H.max_abs_diff(got_c, ref_rgba) <= TOL
    renderer.configure(None)


def test_full_pass_chain_skybox_proxy_splats(renderer):
    """BASELINE config 5's pass structure on a small frame: skybox, proxy (map grid on a HeightMap surface), then the
    splats over the proxy colour, depth-tested against the proxy depth -- everything stays on the device."""
    import torch
    cfg = dict(tile_map_half_wh=(3, 4), surface_type=1, lod_max_dist=24.0, tile_sort_type=3, merge_type=2,
               height_map_wh=(4, 4), height_map_scale=(1.0, 1.0, 0.3))
    W, Hh = 320, 240
    pos, tgt = (0.5, 0.3, 5.0), (1.0, 6.0, 2.5)
    verts = synth.make_tileset(n_lod=3, n_tile=16, lod0_count=600)
    pipe = GSWTPipeline(verts, host.user_data(**cfg), renderer=renderer)
    cu, vp = host.camera_uniforms(pos, tgt, (0, 0, 1), 45.0, 0.1, 2400.0, W, Hh)
    pipe.update(pos, vp)
    faces, mips = _cube(), _mips()
    renderer.skybox_configure(faces)
    renderer.proxy_configure(mips)
    # oracle side
    pp = orc.preprocess([[orc.scene_load(v) for v in lod] for lod in verts])
    ow = wo.WangTile(pp)
    ou = ow.configure(wo.UserData(**cfg))
    ocam = orc.Camera(W, Hh, pos, tgt, [0, 0, 1])
    osd = ow.build_tiles(pos)
    osort = ow.sort_tiles(pos, ocam.view_proj())
    odraws = wo.renderer_draws(pp, osort, ocam.view_proj())
    osu = wo.scene_uniforms_from_data(ou, osd["center_coord"])
    hm = ou.height_map.reshape(ou.height_map_wh[1], ou.height_map_wh[0])
    pu = orc.proxy_uniforms(ocam, map_proxy=1, height_offset=-0.5, surface_type=1, map_half_wh=cfg["tile_map_half_wh"],
                            center_coord=osd["center_coord"], height_map_scale=tuple(osu.height_map_scale[:3]))
    ref_bg = orc.skybox_render(ocam, faces, W, Hh)
    ref_depth = np.ones((Hh, W), np.float32)
    orc.proxy_render(pu, W, Hh, ref_bg, ref_depth, mips, height_map=hm)
    ref, st = orc.render(ocam.uniforms(), osu, pp.tex, odraws, W, Hh, height_map=hm, bg_rgba=ref_bg, bg_depth=ref_depth)
    # device side
    bg = torch.zeros((Hh, W, 4), dtype=torch.float32, device="cuda")
    depth = torch.zeros((Hh, W), dtype=torch.float32, device="cuda")
    out = torch.zeros((Hh, W, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    renderer.skybox_render(cu, W, Hh, bg.data_ptr())
    renderer.proxy_render(pu, W, Hh, bg.data_ptr(), depth.data_ptr(), clear_depth=True)
    su = pipe.wang.scene_uniforms()
    renderer.render(cu, su, W, Hh, bg_rgba=bg.data_ptr(), bg_depth=depth.data_ptr(), bg_on_device=True, out_device_ptr=out.data_ptr())
    renderer.synchronize()
    assert np.array_equal(depth.cpu().numpy().view(np.uint32), ref_depth.view(np.uint32))
    assert renderer.timings()["n_visible"] == st["n_visible"]
    assert H.max_abs_diff(out.cpu().numpy(), ref) <= TOL
    assert (ref_depth < 1.0).mean() > 0.3 and st["n_visible"] > 300
