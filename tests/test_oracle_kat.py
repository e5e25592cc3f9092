"""Known-answer tests that pin the CPU oracle.  The reference ships no
tests or golden vectors, so every expected value here is derived from the reference's source text
in closed form (file:line cited per test)."""
import math

import numpy as np
import pytest

from oracle import gswt_oracle as orc
from oracle import wangtile_oracle as wo

f32 = np.float32


# ---- K1 halves (utils.rs:66-73, gswt.wgsl:478-494) -------------------------------------------
def test_k1_half_roundtrip_normals():
    for x in [0.0, 1.0, -1.0, 0.5, 1.5, 1024.0, 65504.0, 6.103515625e-05, -3.140625]:
        h = orc.float_to_half(x)
        assert orc.half_to_float(h) == x
    assert orc.float_to_half(1.0) == 0x3C00 and orc.float_to_half(-2.0) == 0xC000
    assert orc.pack_half_2x16(1.0, -2.0) == 0xC0003C00


def test_k1_half_rounding_is_nearest_even():
    assert orc.float_to_half(1.0 + 2.0 ** -11) == 0x3C00          # exactly half way -> even (down)
    assert orc.float_to_half(1.0 + 3 * 2.0 ** -11) == 0x3C02      # half way -> even (up)
    assert orc.float_to_half(1.0 + 2.0 ** -11 + 2.0 ** -20) == 0x3C01
    assert orc.float_to_half(70000.0) == 0x7C00                    # overflow -> inf


def test_k1_custom_decode_quirks():
    # subnormal halves decode with 2^-15, not the IEEE 2^-14 (gswt.wgsl:483-485)
    assert orc.half_to_float(0x0001) == 2.0 ** -15 * (1.0 / 1024.0)
    assert orc.half_to_float(0x03FF) == 2.0 ** -15 * (1023.0 / 1024.0)
    assert orc.half_to_float(0x8200) == -(2.0 ** -15) * 0.5
    # Inf / NaN decode to 0 (gswt.wgsl:486-489)
    assert orc.half_to_float(0x7C00) == 0.0 and orc.half_to_float(0xFC00) == 0.0 and orc.half_to_float(0x7E00) == 0.0


# ---- K2/K4 record layout + covariance (scene.rs:306-411) --------------------------------------
def _row(pos, scale, rgba, rot_bytes):
    r = np.zeros(32, dtype=np.uint8)
    r[:12] = np.array(pos, dtype="<f4").view(np.uint8)
    r[12:24] = np.array(scale, dtype="<f4").view(np.uint8)
    r[24:28] = rgba
    r[28:32] = rot_bytes
    return r


def test_k2_k4_generate_texture_axis_aligned():
    # quaternion bytes (255,128,128,128): w = 1, x=y=z = 128/255*2-1 = 1/255 (not renormalised)
    rows = np.stack([_row((1, 2, 3), (0.5, 0.25, 0.125), (10, 20, 30, 40), (255, 128, 128, 128))])
    tex = orc.generate_texture(rows)
    assert tex.shape == (1, 8)
    assert tex[0, :3].view(np.float32).tolist() == [1.0, 2.0, 3.0] and tex[0, 3] == 0
    assert tex[0, 7] == (10 | 20 << 8 | 30 << 16 | 40 << 24)
    q = np.array([1.0, 1 / 255, 1 / 255, 1 / 255])
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    M = R @ np.diag([0.5, 0.25, 0.125])
    S = 4.0 * (M @ M.T)
    got = [orc.half_to_float(tex[0, 4] & 0xFFFF), orc.half_to_float(tex[0, 4] >> 16),
           orc.half_to_float(tex[0, 5] & 0xFFFF), orc.half_to_float(tex[0, 5] >> 16),
           orc.half_to_float(tex[0, 6] & 0xFFFF), orc.half_to_float(tex[0, 6] >> 16)]
    want = [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]]
    assert np.allclose(got, want, rtol=2e-3, atol=1e-6)           # half precision


def test_k4_rotation_90deg_about_z_swaps_axes():
    # q = (cos45, 0, 0, sin45) -> bytes (218, 128->127.5?, ..): use exact byte values and check symmetry
    b = int((math.cos(math.pi / 4) + 1) * 0.5 * 255)
    rows = np.stack([_row((0, 0, 0), (1.0, 0.1, 0.1), (0, 0, 0, 255), (b, 127, 127, b))])
    tex = orc.generate_texture(rows)
    sxx = orc.half_to_float(tex[0, 4] & 0xFFFF)
    syy = orc.half_to_float(tex[0, 5] >> 16)
    assert syy > 10 * sxx                                          # long axis now along y


# ---- K3 Scene::load packing (scene.rs:126-208) ---------------------------------------------------
def test_k3_scene_load_packing_and_importance_order():
    v = np.zeros((3, 62), dtype=np.float32)
    v[:, 58] = 1.0                                    # identity quaternion (w first)
    v[0, 55:58] = math.log(0.1); v[0, 54] = 0.0       # small
    v[1, 55:58] = math.log(1.0); v[1, 54] = 10.0      # big + opaque -> first
    v[2, 55:58] = math.log(0.5); v[2, 54] = -10.0     # mid size, nearly transparent -> last
    v[:, 0] = [1, 2, 3]
    v[1, 6:9] = [100.0, -100.0, float("nan")]         # saturating u8 casts: 255, 0, NaN -> 0
    rows = orc.scene_load(v)
    order = orc.rows_positions(rows)[:, 0].tolist()
    assert order == [2.0, 1.0, 3.0]
    assert rows[0, 24:27].tolist() == [255, 0, 0]
    assert rows[0, 27] == int(255 * (1 / (1 + math.exp(-10.0))))
    assert rows[1, 24:27].tolist() == [127, 127, 127]             # (0.5 + C0*0)*255 = 127.5 -> 127
    assert rows[1, 27] == 127                                      # sigmoid(0)*255 = 127.5 -> 127
    assert rows[0, 28:32].tolist() == [255, 127, 127, 127]        # ((1+1)/2*255, (0+1)/2*255 -> 127)
    assert np.allclose(orc.rows_scales(rows)[0], 1.0)


def test_k3_ply_body_behind_a_header_of_any_length():
    """scene.rs:72-212: the PLY body starts wherever `end_header\n` ends, so a body viewed in place is 4-byte aligned only by luck;
    the rows must not depend on it (and the C oracle must be handed an aligned array: tools/host_sanitizers.sh watches that)."""
    rng = np.random.default_rng(3)
    v = rng.normal(size=(5, 62)).astype(np.float32)
    want = orc.scene_load(v)
    for pad in range(4):
        head = b"ply\nformat binary_little_endian 1.0\ncomment " + b"x" * pad + b"\nelement vertex 5\nend_header\n"
        assert np.array_equal(orc.scene_from_ply(head + v.tobytes()), want)


# ---- K5 counting sort (scene.rs:655-698) ----------------------------------------------------------
def test_k5_counting_sort_descending_reverse_stable():
    seg, idx = orc.sort_raw_depth_vec([np.array([5, 1, 9, 5, 1], dtype=np.int32)])
    assert idx.tolist() == [2, 3, 0, 4, 1]            # descending; ties in REVERSE input order


def test_k5_all_equal_maps_nan_to_bucket_zero():
    seg, idx = orc.sort_raw_depth_vec([np.full(4, 7, dtype=np.int32)])
    assert idx.tolist() == [3, 2, 1, 0]               # inv = inf, 0*inf = NaN -> 0; stable then reversed


def test_k5_two_segments_and_range_mapping():
    a = np.array([0, 65535 * 4], dtype=np.int32)
    b = np.array([65535 * 2, -4], dtype=np.int32)
    seg, idx = orc.sort_raw_depth_vec([a, b])
    # range 262144 -> inv = 65535/262144: depths 0 and -4 both land in bucket 0 (16-bit quantisation),
    # and ties come out in reverse input order, so the later segment's -4 precedes the 0
    assert list(zip(seg.tolist(), idx.tolist())) == [(0, 1), (1, 0), (1, 1), (0, 0)]


def test_k5_raw_depth_truncates_toward_zero():
    rows = np.stack([_row((0.0003, 0, 0), (1, 1, 1), (0, 0, 0, 0), (255, 128, 128, 128)),
                     _row((-0.0003, 0, 0), (1, 1, 1), (0, 0, 0, 0), (255, 128, 128, 128))])
    vp = np.zeros(16, dtype=np.float32); vp[2] = 1.0
    assert orc.raw_depth(rows, vp).tolist() == [1, -1]            # 1.2288 -> 1, -1.2288 -> -1


# ---- K6/K7 Wang ids and neighbours (wangtile.rs:257-338,1830-1846) ---------------------------------
def test_k6_color_id_roundtrip():
    for tid in range(64):
        c = wo.WangTile.tile_id_to_color(tid)
        assert wo.WangTile.color_to_tile_id(c, tid // 16) == tid
    assert wo.WangTile.tile_id_to_color(0b1010) == (1, 0, 1, 0)   # W,N,E,S = bits 8,4,2,1


def test_k7_plane_neighbours_3x3():
    pp = _tiny_pp()
    w = wo.WangTile(pp)
    w.configure(wo.UserData(tile_map_half_wh=(1, 1), surface_type=0, lod_max_dist=10.0))
    nb = w.neighbor_map[1][1]
    assert nb[0] == ((0, 1), 2) and nb[1] == ((1, 2), 3) and nb[2] == ((2, 1), 0) and nb[3] == ((1, 0), 1)
    assert w.neighbor_map[0][0][0] is None and w.neighbor_map[0][0][3] is None
    assert w.neighbor_map[2][2][1] is None and w.neighbor_map[2][2][2] is None


# ---- K8 default camera (camera.rs:169-188, state.rs:114-122) -----------------
@pytest.mark.parametrize("W,H,p00,focal,htx", [(640, 480, 1.8106602, 579.41125, 0.55228475),
                                                (1920, 1080, 1.3579951, 1303.67532, 0.73637967)])
def test_k8_default_camera_uniforms(W, H, p00, focal, htx):
    cam = orc.default_camera(W, H)
    cu = cam.uniforms()
    view = np.array(cu.view[:]).reshape(4, 4).T                  # row-major math view
    assert np.allclose(view, [[1, 0, 0, 0], [0, 0, 1, -5], [0, -1, 0, 0], [0, 0, 0, 1]], atol=1e-6)
    assert abs(cu.projection[0] - p00) < 1e-6 and abs(cu.projection[5] - 2.4142136) < 1e-6
    assert abs(cu.projection[10] + 1.0000833) < 1e-6 and abs(cu.projection[14] + 0.20000833) < 1e-6
    assert abs(cu.focal[0] - focal) < 1e-2 and abs(cu.focal[1] - focal) < 1e-2
    assert abs(cu.htan_fov[0] - htx) < 1e-6 and abs(cu.htan_fov[1] - 0.41421356) < 1e-6
    assert cu.cam_pos[:3] == [0.0, 0.0, 5.0] and cu.viewport[:] == [float(W), float(H)]


# ---- helpers for the render KATs ------------------------------------------------------------------
def _tiny_pp():
    from gswt_renderer_amd import synth
    verts = synth.make_tileset(n_lod=2, n_tile=16, lod0_count=40)
    return orc.preprocess([[orc.scene_load(v) for v in lod] for lod in verts])


def _one_splat_tex(pos, sigma_diag, rgba=(255, 128, 0, 255)):
    """Record with covariance diag(sigma^2) (stored x4 in f16, scene.rs:403-405)."""
    tex = np.zeros((1, 8), dtype=np.uint32)
    tex[0, :3] = np.array(pos, dtype="<f4").view(np.uint32)
    sx, sy, sz = [4.0 * s * s for s in sigma_diag]
    tex[0, 4] = orc.pack_half_2x16(sx, 0.0)
    tex[0, 5] = orc.pack_half_2x16(0.0, sy)
    tex[0, 6] = orc.pack_half_2x16(0.0, sz)
    tex[0, 7] = rgba[0] | rgba[1] << 8 | rgba[2] << 16 | rgba[3] << 24
    return tex


def _draw1(**kw):
    return [orc.Draw(orc.tile_uniforms(**kw), np.array([0], dtype=np.uint32), np.array([0], dtype=np.uint32),
                     np.array([kw.pop("_lod", 0)], dtype=np.uint32))]


# ---- K9 single splat: pixel = alpha * rgb * exp(-|p|^2) with analytic p ------------------------------
def test_k9_single_splat_analytic():
    W, H = 64, 48
    cam = orc.default_camera(W, H)
    cu = cam.uniforms()
    D = 4.0
    # slightly anisotropic so the eigen-basis is defined (an exactly isotropic, centred splat makes
    # normalize((0,0)) = NaN in gswt.wgsl:256 and nothing is drawn -- covered below)
    sig = (0.05, 0.05, 0.1)
    tex = _one_splat_tex((0.0, D, 5.0), sig)
    su = orc.scene_uniforms(num_lod=1)
    img, st = orc.render(cu, su, tex, _draw1(valid_lod_id=0), W, H)
    assert st["n_visible"] == 1
    fx, fy = cu.focal[0], cu.focal[1]
    sx_px, sy_px = sig[0] * fx / D, sig[2] * fy / D              # world x -> screen x, world z -> screen y
    ys, xs = np.mgrid[0:H, 0:W]
    dx, dy = xs + 0.5 - W / 2, ys + 0.5 - H / 2
    r2 = dx * dx / (2 * sx_px * sx_px) + dy * dy / (2 * sy_px * sy_px)
    want_a = np.where(r2 <= 4.0, np.exp(-r2), 0.0)               # alpha byte 255 -> 1.0
    assert np.allclose(img[..., 3], want_a, atol=2e-3)           # f16 covariance storage
    assert np.allclose(img[..., 0], want_a * 1.0, atol=2e-3)
    assert np.allclose(img[..., 1], want_a * (128 / 255), atol=2e-3)
    assert img[..., 2].max() == 0.0
    assert img[H // 2, W // 2, 3] > 0.5


def test_k9_exactly_isotropic_centred_splat_is_degenerate():
    W, H = 64, 64                                                 # fx == fy
    cu = orc.default_camera(W, H).uniforms()
    tex = _one_splat_tex((0.0, 4.0, 5.0), (0.05, 0.05, 0.05))
    img, st = orc.render(cu, orc.scene_uniforms(num_lod=1), tex, _draw1(valid_lod_id=0), W, H)
    assert st["n_visible"] == 0 and img.max() == 0.0             # normalize(vec2(0,0)) -> NaN axes


# ---- K10 two overlapping splats: "over" back-to-front ---------------------------------------------
def test_k10_two_splats_over_blending():
    W, H = 32, 32
    cu = orc.default_camera(W, H).uniforms()
    t_far = _one_splat_tex((0.0, 6.0, 5.0), (0.4, 0.4, 0.5), (255, 0, 0, 128))
    t_near = _one_splat_tex((0.0, 3.0, 5.0), (0.2, 0.2, 0.25), (0, 0, 255, 128))
    tex = np.concatenate([t_far, t_near])
    su = orc.scene_uniforms(num_lod=1)
    d = [orc.Draw(orc.tile_uniforms(valid_lod_id=0), np.array([0, 1], dtype=np.uint32), None, np.zeros(2, np.uint32))]
    both, _ = orc.render(cu, su, tex, d, W, H)
    far, _ = orc.render(cu, su, t_far, _draw1(valid_lod_id=0), W, H)
    near, _ = orc.render(cu, su, t_near, _draw1(valid_lod_id=0), W, H)
    want = near + far * (1.0 - near[..., 3:4])                    # src + dst * (1 - src.a), renderer.rs:118-129
    assert np.allclose(both, want, atol=1e-6)
    # reversed draw order gives a different image (order matters)
    d2 = [orc.Draw(orc.tile_uniforms(valid_lod_id=0), np.array([1, 0], dtype=np.uint32), None, np.zeros(2, np.uint32))]
    rev, _ = orc.render(cu, su, tex, d2, W, H)
    assert np.abs(rev - both).max() > 0.05


# ---- K11 LOD blend ratio and discard rules (gswt.wgsl:134-141,402-408) -------------------------------
@pytest.mark.parametrize("dist,t_expected", [(9.5, 0.0), (9.75, 0.0), (10.0, 0.5), (10.125, 0.75), (10.25, 1.0), (10.5, 1.0)])
def test_k11_lod_transition_ratio(dist, t_expected):
    W, H = 32, 32
    cu = orc.default_camera(W, H).uniforms()
    tex = _one_splat_tex((0.0, dist, 5.0), (0.3, 0.3, 0.35), (255, 255, 255, 255))
    su = orc.scene_uniforms(num_lod=2, transition_width_ratio=0.05, transition_dist=(10.0, 20.0))
    # blending tile on lod 0 going to lower: higher_lod = 0; t = clamp((d - 10)/0.5 + 0.5)
    outs = {}
    for lod_id in (0, 1):
        d = [orc.Draw(orc.tile_uniforms(changing=1, changing_to_lower=1, tile_id=(0, 0, 0)),
                      np.array([0], dtype=np.uint32), None, np.array([lod_id], dtype=np.uint32))]
        sp = orc.project_draws(cu, su, tex, d)[0]
        outs[lod_id] = sp
    hi, lo = outs[0], outs[1]
    if t_expected == 1.0:
        assert hi["visible"] == 0                                 # higher-LOD splat dropped at t == 1
    else:
        assert hi["visible"] == 1 and abs(hi["rgba"][3] - (1.0 - t_expected)) < 1e-5
    if t_expected == 0.0:
        assert lo["visible"] == 0                                 # lower-LOD splat dropped at t == 0
    else:
        assert lo["visible"] == 1 and abs(lo["rgba"][3] - t_expected) < 1e-5


def test_k11_plain_tile_discards_other_lod():
    cu = orc.default_camera(32, 32).uniforms()
    tex = _one_splat_tex((0.0, 5.0, 5.0), (0.3, 0.3, 0.35))
    su = orc.scene_uniforms(num_lod=2)
    for lod_id, vis in ((0, 1), (1, 0)):
        d = [orc.Draw(orc.tile_uniforms(valid_lod_id=0), np.array([0], dtype=np.uint32), None, np.array([lod_id], dtype=np.uint32))]
        assert orc.project_draws(cu, su, tex, d)[0]["visible"] == vis


# ---- K12 merged-tile offset from map_id (gswt.wgsl:52-63) ---------------------------------------------
def test_k12_merged_offset_from_map_id():
    W, H = 64, 64
    cu = orc.default_camera(W, H).uniforms()
    tex = _one_splat_tex((0.3, 0.7, 4.0), (0.2, 0.2, 0.25))
    half, cc, tw = (2, 3), (1, -2), 4.0
    su = orc.scene_uniforms(num_lod=1, map_half_wh=half, center_coord=cc, tile_width=tw)
    map_h = 2 * half[1] + 1
    for mx, my in [(0, 0), (2, 3), (4, 6), (3, 5)]:
        map_id = mx * map_h + my
        ox = ((mx - half[0]) + cc[0]) * tw
        oy = ((my - half[1]) + cc[1]) * tw
        dm = [orc.Draw(orc.tile_uniforms(single_draw=1, single_lod_id=0), np.array([0], dtype=np.uint32),
                       np.array([map_id], dtype=np.uint32), None)]
        ds = [orc.Draw(orc.tile_uniforms(valid_lod_id=0, offset=(ox, oy, 0.0)), np.array([0], dtype=np.uint32), None,
                       np.zeros(1, np.uint32))]
        a = orc.project_draws(cu, su, tex, dm)[0]
        b = orc.project_draws(cu, su, tex, ds)[0]
        assert a["visible"] == b["visible"]
        for fld in ("ndc", "depth", "major", "minor"):
            assert np.array_equal(a[fld], b[fld])


# ---- K13 tile corner cull decisions (renderer.rs:472-494) ------------------------------------------------
def test_k13_tile_cull_min_abs_quirk():
    pp = _tiny_pp()
    cam = orc.default_camera(64, 48)
    vp = cam.view_proj()

    def ti_with(corners):
        z3 = np.zeros(3, dtype=np.float32)
        return wo.TileInstance((0, 0), 0, z3, 0, (0, 0), z3, ("none",), ("none",), wo.mat3_identity(),
                               [(np.array(c, dtype=np.float32), wo.mat3_identity()) for c in corners], None)
    def n_draws(corners):
        sd = {"tile_instance_vec": [ti_with(corners)], "render_data_vec": [((0, ((0, 0),), (("none",),)), None)]}
        return len(wo.renderer_draws(pp, sd, vp, culling_dist=1.0))
    assert n_draws([(-1, 4, 4), (-1, 8, 4), (1, 8, 4), (1, 4, 4)]) == 1           # in front, on screen
    assert n_draws([(50, 4, 4), (50, 8, 4), (54, 8, 4), (54, 4, 4)]) == 0         # far to the right
    # straddles the screen centre with every corner off-screen: min|x| > 1 -> culled (reference quirk)
    assert n_draws([(-30, 4, 4), (-30, 5, 4), (30, 5, 4), (30, 4, 4)]) == 0


# ---- K14 depth test against a constant proxy depth (renderer.rs:179-185,430-441) ---------------------------
def test_k14_depth_test_vs_background():
    W, H = 32, 32
    cu = orc.default_camera(W, H).uniforms()
    tex = _one_splat_tex((0.0, 5.0, 5.0), (0.5, 0.5, 0.6), (255, 255, 255, 255))
    su = orc.scene_uniforms(num_lod=1)
    depth = float(orc.project_draws(cu, su, tex, _draw1(valid_lod_id=0))[0]["depth"])
    bg = np.full((H, W, 4), 0.25, dtype=np.float32)
    behind, _ = orc.render(cu, su, tex, _draw1(valid_lod_id=0), W, H, bg_rgba=bg, bg_depth=np.full((H, W), depth, np.float32))
    front, _ = orc.render(cu, su, tex, _draw1(valid_lod_id=0), W, H, bg_rgba=bg,
                          bg_depth=np.full((H, W), np.nextafter(np.float32(depth), np.float32(2.0)), np.float32))
    assert np.array_equal(behind, bg)                              # depth < depthbuf fails on equality
    assert front[H // 2, W // 2, 0] > 0.9


# ---- K15 z clip: outside [0,1] is invisible (gswt.wgsl:152-160,415-419) ----------------------------------
def test_k15_near_far_clip():
    cu = orc.default_camera(32, 32).uniforms()
    su = orc.scene_uniforms(num_lod=1)
    vis = lambda y: orc.project_draws(cu, su, _one_splat_tex((0.0, y, 5.0), (0.01, 0.01, 0.012)), _draw1(valid_lod_id=0))[0]["visible"]
    assert vis(0.05) == 0       # in front of the near plane (0.1)
    assert vis(0.2) == 1
    assert vis(2399.0) == 1
    assert vis(2500.0) == 0     # beyond the far plane (2400): depth 1.0000017 > 1
    # just past the far plane the f32 depth rounds to exactly 1.0: it survives the clip but fails the
    # `Less` depth test against the 1.0 clear value (renderer.rs:182,436), so nothing is drawn
    tex = _one_splat_tex((0.0, 2401.0, 5.0), (30.0, 30.0, 36.0))
    sp = orc.project_draws(cu, su, tex, _draw1(valid_lod_id=0))[0]
    assert sp["visible"] == 1 and sp["depth"] == 1.0
    img, _ = orc.render(cu, su, tex, _draw1(valid_lod_id=0), 32, 32)
    assert img.max() == 0.0
    assert vis(-1.0) == 0       # behind the camera


def test_k16_canonical_sincos_matches_libm():
    """The canonical sin / cos both sides evaluate is a real sin / cos: within 2e-7 of libm on the ranges the sphere
    mapping and the debug hash use."""
    import ctypes as C
    lib = orc.lib()
    lib.orc_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.orc_sincosf.restype = None
    rng = np.random.default_rng(3)
    xs = np.concatenate([rng.uniform(-8, 8, 2000), rng.uniform(-6000, 6000, 2000), [0.0, np.pi / 2, -np.pi, 1e-8]]).astype(np.float32)
    worst = 0.0
    for x in xs:
        s, c = C.c_float(), C.c_float()
        lib.orc_sincosf(float(x), C.byref(s), C.byref(c))
        worst = max(worst, abs(s.value - np.sin(np.float64(x))), abs(c.value - np.cos(np.float64(x))))
    assert worst < 2e-7, worst
