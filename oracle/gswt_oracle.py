"""CPU ORACLE for the GSWT hot path -- TEST INFRASTRUCTURE ONLY.

Python/numpy half of the oracle: ctypes bindings to ``gswt_oracle.c`` plus a
restatement of the reference's *host* stages that produce the render inputs
(zengyf131/gswt_renderer: src/camera.rs, src/scene.rs, src/wangtile.rs,
src/renderer.rs).  Every function cites the reference file:line it follows.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product package ``gswt_renderer_amd`` never
does.

PARITY STATUS: "parity unpinned by the reference" -- the reference cannot be
built or run here and ships no tests/goldens.  Pins are the
known-answer tests in tests/test_oracle_kat.py and tests/golden/.

All float arithmetic is numpy float32, one rounding per operator, sequential
accumulation where the reference accumulates sequentially.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32

# --------------------------------------------------------------------------
# C library
# --------------------------------------------------------------------------


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def build(force: bool = False) -> None:
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    out = os.path.join(_HERE, "_build", "libgswt_oracle.so")
    src = os.path.join(_HERE, "gswt_oracle.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)


_lib = None


class Camera176(C.Structure):
    _fields_ = [("projection", C.c_float * 16), ("view", C.c_float * 16), ("focal", C.c_float * 2),
                ("viewport", C.c_float * 2), ("htan_fov", C.c_float * 4), ("cam_pos", C.c_float * 4)]


class Scene160(C.Structure):
    _fields_ = [("splat_scale", C.c_float), ("tile_width", C.c_float), ("use_clip", C.c_uint32),
                ("clip_height", C.c_float), ("surface_type", C.c_uint32), ("sphere_radius", C.c_float),
                ("point_cloud_radius", C.c_float), ("transition_width_ratio", C.c_float),
                ("num_lod", C.c_uint32), ("draw_mode", C.c_uint32), ("map_half_wh", C.c_uint32 * 2),
                ("center_coord", C.c_int32 * 2), ("_pad0", C.c_uint32 * 2),
                ("transition_dist", C.c_float * 16), ("height_map_scale", C.c_float * 4),
                ("scene_scale", C.c_float * 4)]


class Tile80(C.Structure):
    _fields_ = [("single_draw", C.c_uint32), ("map_index", C.c_uint32), ("single_lod_id", C.c_int32),
                ("valid_lod_id", C.c_int32), ("changing", C.c_uint32), ("changing_to_lower", C.c_int32),
                ("_pad0", C.c_uint32 * 2), ("tile_id", C.c_uint32 * 4), ("offset", C.c_float * 4),
                ("map_coord", C.c_uint32 * 4)]


class OrcDraw(C.Structure):
    _fields_ = [("tile", Tile80), ("gs_index", C.c_void_p), ("map_id", C.c_void_p),
                ("lod_id", C.c_void_p), ("count", C.c_uint32), ("_pad", C.c_uint32)]


class OrcSplat(C.Structure):
    _fields_ = [("visible", C.c_int32), ("ndc", C.c_float * 2), ("depth", C.c_float),
                ("major", C.c_float * 2), ("minor", C.c_float * 2), ("rgba", C.c_float * 4)]


SPLAT_DTYPE = np.dtype([("visible", "<i4"), ("ndc", "<f4", 2), ("depth", "<f4"), ("major", "<f4", 2),
                        ("minor", "<f4", 2), ("rgba", "<f4", 4)])


class OrcStats(C.Structure):
    _fields_ = [("n_instanced", C.c_uint64), ("n_visible", C.c_uint64), ("n_pairs16", C.c_uint64)]


assert C.sizeof(Camera176) == 176 and C.sizeof(Scene160) == 160 and C.sizeof(Tile80) == 80
assert C.sizeof(OrcSplat) == SPLAT_DTYPE.itemsize == 48


def lib():
    global _lib
    if _lib is None:
        build()
        name = "libgswt_oracle_fma.so" if _cpu_has_fma() else "libgswt_oracle.so"
        _lib = C.CDLL(os.path.join(_HERE, "_build", name))
        _lib.orc_half_to_float.restype = C.c_float
        _lib.orc_half_to_float.argtypes = [C.c_uint32]
        _lib.orc_float_to_half.restype = C.c_uint32
        _lib.orc_float_to_half.argtypes = [C.c_float]
        _lib.orc_pack_half_2x16.restype = C.c_uint32
        _lib.orc_pack_half_2x16.argtypes = [C.c_float, C.c_float]
        _lib.orc_generate_texture.restype = None
        _lib.orc_generate_texture.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.orc_sort_raw_depth.restype = None
        _lib.orc_sort_raw_depth.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.orc_raw_depth.restype = None
        _lib.orc_raw_depth.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        _lib.orc_scene_load.restype = None
        _lib.orc_scene_load.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        _lib.orc_project.restype = C.c_int
        _lib.orc_project.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                     C.c_uint32, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_render.restype = C.c_int
        _lib.orc_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_void_p, C.c_void_p]
        _lib.orc_project_draws.restype = C.c_int
        _lib.orc_project_draws.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_set_strict.restype = None
        _lib.orc_set_strict.argtypes = [C.c_int]
        _lib.orc_get_strict.restype = C.c_int
        _lib.orc_compare_modes.restype = C.c_int
        _lib.orc_compare_modes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _lib.orc_compare_modes2.restype = C.c_int
        _lib.orc_compare_modes2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return _lib


class strict:
    """Context manager, process-wide switch of the evaluation mode (gswt_oracle.c, "STRICT mode"):
    default (mode 2, no context needed): the vertex stage gswt.wgsl:152-258,260-265,402-419 operator by operator (IEEE `/`, no fused
        multiply-add, full matrix products) + the fragment sequence F1..F4 -- what the HIP path computes by default;
    strict() (mode 1): the fragment stage too -- exact quad interpolation, fs_main as written: the anchor image;
    strict(fragment=False): mode 2 explicitly;
    v2() (mode 0): the rounding sequence v2 (fma chains, one reciprocal per quotient) = the HIP path with GSWT_OPT_STRICT_VS = 0."""

    def __init__(self, fragment: bool = True):
        self.mode = 1 if fragment else 2

    def __enter__(self):
        self.prev = lib().orc_get_strict()
        lib().orc_set_strict(self.mode)
        return self

    def __exit__(self, *exc):
        lib().orc_set_strict(self.prev)
        return False


class v2(strict):
    """Context manager: the rounding sequence v2 everywhere (see `strict`)."""

    def __init__(self):
        self.mode = 0


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# --------------------------------------------------------------------------
# K1: halves
# --------------------------------------------------------------------------
def half_to_float(h: int) -> float:
    """halfToFloat, gswt.wgsl:478-494 (custom decode)."""
    return float(lib().orc_half_to_float(int(h) & 0xFFFF))


def float_to_half(x: float) -> int:
    """half::f16::from_f32 (RNE), call site utils.rs:69-70."""
    return int(lib().orc_float_to_half(C.c_float(x)))


def pack_half_2x16(x: float, y: float) -> int:
    """utils.rs:66-73"""
    return int(lib().orc_pack_half_2x16(C.c_float(x), C.c_float(y)))


# --------------------------------------------------------------------------
# cgmath 0.18 restatements (f32).  Column-major 4x4 as flat [4*c + r].
# --------------------------------------------------------------------------
def _dot3(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _normalize3(v):
    """cgmath InnerSpace::normalize = v * (1 / magnitude)."""
    v = np.asarray(v, dtype=f32)
    inv = f32(1.0) / f32(np.sqrt(_dot3(v, v)))
    return (v * inv).astype(f32)


def _cross3(a, b):
    return np.array([f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]),
                     f32(a[0] * b[1]) - f32(a[1] * b[0])], dtype=f32)


def perspective(fovy_deg: float, aspect: float, near: float, far: float) -> np.ndarray:
    """cgmath::perspective (camera.rs:115-120, wangtile.rs:145).  cot(fovy/2) is
    evaluated in double and rounded once (Rust's f32 tan is libm-dependent)."""
    fovy = f32(f32(fovy_deg) * f32(math.pi / 180.0))
    f = f32(1.0 / math.tan(float(fovy) / 2.0))
    near, far, aspect = f32(near), f32(far), f32(aspect)
    m = np.zeros(16, dtype=f32)
    m[0] = f / aspect
    m[5] = f
    m[10] = (far + near) / (near - far)
    m[11] = f32(-1.0)
    m[14] = (f32(2.0) * far * near) / (near - far)
    return m


def look_at_rh(eye, center, up) -> np.ndarray:
    """cgmath Matrix4::look_at_rh -> look_to_rh(eye, center - eye, up) (camera.rs:94-98)."""
    eye = np.asarray(eye, dtype=f32)
    fdir = _normalize3(np.asarray(center, dtype=f32) - eye)
    s = _normalize3(_cross3(fdir, np.asarray(up, dtype=f32)))
    u = _cross3(s, fdir)
    m = np.zeros(16, dtype=f32)
    m[0], m[1], m[2] = s[0], u[0], -fdir[0]
    m[4], m[5], m[6] = s[1], u[1], -fdir[1]
    m[8], m[9], m[10] = s[2], u[2], -fdir[2]
    m[12], m[13], m[14] = -_dot3(eye, s), -_dot3(eye, u), _dot3(eye, fdir)
    m[15] = f32(1.0)
    return m


def mat4_mul(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """cgmath Matrix4 * Matrix4: out[c][r] = sum_k a[k][r] * b[c][k], left to right."""
    out = np.zeros(16, dtype=f32)
    for c in range(4):
        for r in range(4):
            acc = f32(a[r] * b[4 * c])
            for k in range(1, 4):
                acc = f32(acc + f32(a[4 * k + r] * b[4 * c + k]))
            out[4 * c + r] = acc
    return out


def mat4_vec(m: np.ndarray, v) -> np.ndarray:
    """cgmath Matrix4 * Vector4 = col0*x + col1*y + col2*z + col3*w."""
    out = np.zeros(4, dtype=f32)
    for r in range(4):
        acc = f32(m[r] * v[0])
        for k in range(1, 4):
            acc = f32(acc + f32(m[4 * k + r] * v[k]))
        out[r] = acc
    return out


@dataclass
class Camera:
    """camera.rs:6-131 (perspective camera)."""
    width: int
    height: int
    position: np.ndarray
    target: np.ndarray
    up: np.ndarray
    fovy_deg: float = 45.0
    z_near: float = 0.1
    z_far: float = 2400.0

    def __post_init__(self):
        self.position = np.asarray(self.position, dtype=f32)
        self.target = np.asarray(self.target, dtype=f32)
        self.up = np.asarray(self.up, dtype=f32)
        self.view = look_at_rh(self.position, self.target, self.up)
        self.projection = perspective(self.fovy_deg, f32(self.width) / f32(self.height), self.z_near, self.z_far)

    def view_proj(self) -> np.ndarray:
        return mat4_mul(self.projection, self.view)          # camera.rs:86-88

    def uniforms(self) -> Camera176:
        """CameraUniforms::from_camera, camera.rs:169-188."""
        w, h = f32(self.width), f32(self.height)
        fx = f32(0.5) * self.projection[0] * w
        fy = f32(-0.5) * self.projection[5] * h
        fovy = f32(f32(self.fovy_deg) * f32(math.pi / 180.0))
        htany = f32(math.tan(float(fovy / f32(2.0))))
        htanx = (htany / h) * w
        cu = Camera176()
        cu.projection[:] = [float(x) for x in self.projection]
        cu.view[:] = [float(x) for x in self.view]
        cu.focal[:] = [abs(float(fx)), abs(float(fy))]
        cu.viewport[:] = [float(w), float(h)]
        cu.htan_fov[:] = [float(htanx), float(htany), 0.0, 0.0]
        cu.cam_pos[:] = [float(self.position[0]), float(self.position[1]), float(self.position[2]), 0.0]
        return cu


def default_camera(width: int, height: int) -> Camera:
    """state.rs:114-122"""
    return Camera(width, height, [0, 0, 5], [0, 1, 5], [0, 0, 1], 45.0, 0.1, 2400.0)


# --------------------------------------------------------------------------
# Scene (scene.rs)
# --------------------------------------------------------------------------
def scene_load(verts62: np.ndarray) -> np.ndarray:
    """Scene::load, scene.rs:115-212 -> [n, 32] uint8 rows."""
    # "A": a PLY body viewed in place (scene_from_ply: np.frombuffer behind an odd-length header) is contiguous but not 4-byte aligned
    verts62 = np.require(verts62, dtype=np.float32, requirements=["C", "A"]).reshape(-1, 62)
    rows = np.zeros((verts62.shape[0], 32), dtype=np.uint8)
    lib().orc_scene_load(_ptr(verts62), verts62.shape[0], _ptr(rows))
    return rows


def parse_ply(data: bytes):
    """Scene::parse_file_header, scene.rs:72-112 -> (header_size, splat_count)."""
    pos = 0
    count = 0
    for i in range(66 + 1):
        nl = data.find(b"\n", pos)
        if nl < 0:
            break
        line = data[pos:nl + 1]
        pos = nl + 1
        if line == b"end_header\n":
            return pos, count
        if line.startswith(b"element vertex "):
            count = int(line[15:-1])
        if i + 1 > 65:
            break
    raise ValueError("Scene::parse_file_header(): ERROR: the file is not correctly formatted.")


def scene_from_ply(data: bytes) -> np.ndarray:
    hs, n = parse_ply(data)
    verts = np.frombuffer(data, dtype="<f4", count=62 * n, offset=hs).reshape(n, 62)
    return scene_load(verts)


def load_scene_zip(path: str):
    """load_scene_zip, scene.rs:1030-1141 -> rows[lod][tile]."""
    import re
    import zipfile
    pat = re.compile(r"lod(\d+)_tile_(\d+)")
    entries = []
    with zipfile.ZipFile(path) as zf:
        for i, info in enumerate(zf.infolist()):
            fname = os.path.basename(info.filename)
            m = pat.search(fname)
            if m:
                entries.append((int(m.group(1)), int(m.group(2)), i, fname, info))
        entries.sort(key=lambda e: (e[0], e[1]))
        n_lod = entries[-1][0] - entries[0][0] + 1
        n_tile = entries[-1][1] + 1
        out = []
        for i in range(n_lod):
            lod_vec = []
            for j in range(n_tile):
                e = entries[i * n_tile + j]
                if ".ply" in e[3]:
                    lod_vec.append(scene_from_ply(zf.read(e[4])))
                elif ".splat" in e[3]:
                    lod_vec.append(np.zeros((0, 32), dtype=np.uint8))   # scene.rs:1120-1125: read, never stored
                else:
                    raise RuntimeError("unreachable")
            out.append(lod_vec)
    return out


def generate_texture(rows: np.ndarray) -> np.ndarray:
    """Scene::generate_texture, scene.rs:306-411 -> [n, 8] uint32."""
    rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1, 32)
    tex = np.zeros((rows.shape[0], 8), dtype=np.uint32)
    lib().orc_generate_texture(_ptr(rows), rows.shape[0], _ptr(tex))
    return tex


def raw_depth(rows: np.ndarray, view_proj: np.ndarray) -> np.ndarray:
    """scene.rs:537-552"""
    rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1, 32)
    vp = np.ascontiguousarray(view_proj, dtype=np.float32)
    out = np.zeros(rows.shape[0], dtype=np.int32)
    lib().orc_raw_depth(_ptr(rows), rows.shape[0], _ptr(vp), _ptr(out))
    return out


def sort_raw_depth_vec(depth_vec):
    """Scene::sort_raw_depth_vec, scene.rs:655-698 -> (segment, index) arrays."""
    lens = [len(d) for d in depth_vec]
    cat = np.ascontiguousarray(np.concatenate(depth_vec).astype(np.int32)) if lens else np.zeros(0, np.int32)
    order = np.zeros(cat.shape[0], dtype=np.uint32)
    lib().orc_sort_raw_depth(_ptr(cat), cat.shape[0], _ptr(order))
    displ = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    seg = (np.searchsorted(displ, order, side="right") - 1).astype(np.int64)
    idx = order.astype(np.int64) - displ[seg]
    return seg, idx


def rows_positions(rows: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(rows[:, :12]).view("<f4").reshape(-1, 3)


def rows_scales(rows: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(rows[:, 12:24]).view("<f4").reshape(-1, 3)


def _seq_sum_f32(a: np.ndarray) -> np.float32:
    """Sequential f32 accumulation (the reference's `+=` loops)."""
    a = np.asarray(a, dtype=f32).ravel()
    if a.size == 0:
        return f32(0.0)
    return f32(np.cumsum(a, dtype=f32)[-1])


PRESORT_DIRS_RAW = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (1, 0, -1), (-1, 0, -1), (0, 1, -1),
                    (0, -1, -1), (0, 0, -1)]                                   # wangtile.rs:146-156


@dataclass
class Preprocessed:
    """Outputs of WangTile::preprocess, wangtile.rs:71-255."""
    n_lod: int
    n_tile: int
    n_view: int
    rows: list                      # [lod][tile] height-normalised rows
    tex: np.ndarray                 # merged [U, 8] uint32
    merge_offset: np.ndarray        # [lod][tile]
    lod_avg_scale: np.ndarray
    presort_dirs: np.ndarray        # [9, 3] f32 normalised
    tile_center: np.ndarray         # [tile, 3]
    aabb: np.ndarray                # [tile, 2, 3]
    raw_depth: list                 # [lod][tile][view] int32 arrays
    gs_index: list                  # [lod][tile][view] uint32
    gs_lod_id: list                 # [lod][tile][view] uint32


def preprocess(tile_rows) -> Preprocessed:
    """WangTile::preprocess, wangtile.rs:71-255."""
    n_lod, n_tile = len(tile_rows), len(tile_rows[0])
    rows = [[np.array(r, dtype=np.uint8, copy=True) for r in lod] for lod in tile_rows]
    aabb = np.zeros((n_tile, 2, 3), dtype=f32)
    centers = np.zeros((n_tile, 3), dtype=f32)
    for t in range(n_tile):
        p = rows_positions(rows[0][t])
        lo, hi = p.min(axis=0), p.max(axis=0)                               # scene.rs:830-861
        avg = np.array([_seq_sum_f32(p[:, k]) for k in range(3)], dtype=f32) / f32(p.shape[0])
        for l in range(n_lod):                                                # :84-87 translate z
            pl = rows_positions(rows[l][t]).copy()
            pl[:, 2] = pl[:, 2] + (-avg[2])
            rows[l][t][:, :12] = pl.view(np.uint8).reshape(-1, 12)
        lo[2] -= avg[2]
        hi[2] -= avg[2]
        avg[2] = f32(0.0)
        aabb[t, 0], aabb[t, 1] = lo, hi
        centers[t] = avg / f32(n_lod)                                         # :106-107
    merge_offset = np.zeros((n_lod, n_tile), dtype=np.uint32)
    acc = 0
    for l in range(n_lod):
        for t in range(n_tile):
            merge_offset[l, t] = acc
            acc += rows[l][t].shape[0]
    merged = np.concatenate([rows[l][t] for l in range(n_lod) for t in range(n_tile)], axis=0)
    tex = generate_texture(merged)
    avg_scale = np.zeros(n_lod, dtype=f32)
    for l in range(n_lod):                                                    # :128-142
        ssum, snum = f32(0.0), 0
        for t in range(n_tile):
            ssum = f32(ssum + _seq_sum_f32(rows_scales(rows[l][t])))
            snum += rows[l][t].shape[0] * 3
        avg_scale[l] = ssum / f32(snum)
        if l > 0:
            assert avg_scale[l] > avg_scale[l - 1]
    dirs = np.stack([_normalize3(np.array(d, dtype=f32)) for d in PRESORT_DIRS_RAW])
    sort_proj = perspective(90.0, 1.0, 0.1, 10.0)                             # :145
    views = []
    for d in dirs:                                                            # :160-174
        up = [0, 0, 1] if (d[0] != 0 or d[1] != 0) else [0, 1, 0]
        views.append(mat4_mul(sort_proj, look_at_rh([0, 0, 0], d, up)))
    n_view = len(views)
    rd = [[[raw_depth(rows[l][t], views[k]) for k in range(n_view)] for t in range(n_tile)]
          for l in range(n_lod)]
    gs_index = [[[None] * n_view for _ in range(n_tile)] for _ in range(n_lod)]
    gs_lod = [[[None] * n_view for _ in range(n_tile)] for _ in range(n_lod)]
    for l in range(n_lod):                                                    # :221-252
        for t in range(n_tile):
            for k in range(n_view):
                dv, lods, offs = [rd[l][t][k]], [l], [merge_offset[l, t]]
                if l < n_lod - 1:
                    dv.append(rd[l + 1][t][k]); lods.append(l + 1); offs.append(merge_offset[l + 1, t])
                seg, idx = sort_raw_depth_vec(dv)
                gs_index[l][t][k] = (idx + np.asarray(offs, dtype=np.int64)[seg]).astype(np.uint32)
                gs_lod[l][t][k] = np.asarray(lods, dtype=np.uint32)[seg]
    return Preprocessed(n_lod, n_tile, n_view, rows, tex, merge_offset, avg_scale, dirs, centers, aabb,
                        rd, gs_index, gs_lod)


# --------------------------------------------------------------------------
# Uniform blocks (renderer.rs:602-726)
# --------------------------------------------------------------------------
def scene_uniforms(*, splat_scale=1.0, tile_width=4.0, use_clip=0, clip_height=0.0, surface_type=0,
                   sphere_radius=0.0, point_cloud_radius=0.0, transition_width_ratio=0.05, num_lod=1,
                   draw_mode=0, map_half_wh=(0, 0), center_coord=(0, 0), transition_dist=(),
                   height_map_scale=(1.0, 1.0, 0.0), scene_scale=(1.0, 1.0, 1.0)) -> Scene160:
    """SceneUniforms::from_data, renderer.rs:631-672."""
    s = Scene160()
    s.splat_scale, s.tile_width, s.use_clip, s.clip_height = splat_scale, tile_width, int(use_clip), clip_height
    s.surface_type, s.sphere_radius, s.point_cloud_radius = int(surface_type), sphere_radius, point_cloud_radius
    s.transition_width_ratio, s.num_lod, s.draw_mode = transition_width_ratio, int(num_lod), int(draw_mode)
    s.map_half_wh[:] = [int(map_half_wh[0]), int(map_half_wh[1])]
    s.center_coord[:] = [int(center_coord[0]), int(center_coord[1])]
    td = [float(x) for x in transition_dist][:16]
    s.transition_dist[:] = td + [0.0] * (16 - len(td))
    s.height_map_scale[:] = [float(height_map_scale[0]), float(height_map_scale[1]), float(height_map_scale[2]), 0.0]
    s.scene_scale[:] = [float(scene_scale[0]), float(scene_scale[1]), float(scene_scale[2]), 0.0]
    return s


def tile_uniforms(*, single_draw=0, map_index=0, single_lod_id=-1, valid_lod_id=-1, changing=0,
                  changing_to_lower=-1, tile_id=(0, 0, 0), offset=(0.0, 0.0, 0.0), map_coord=(0, 0)) -> Tile80:
    """TileUniforms, renderer.rs:675-726."""
    t = Tile80()
    t.single_draw, t.map_index, t.single_lod_id, t.valid_lod_id = int(single_draw), int(map_index), int(single_lod_id), int(valid_lod_id)
    t.changing, t.changing_to_lower = int(changing), int(changing_to_lower)
    t.tile_id[:] = [int(tile_id[0]), int(tile_id[1]), int(tile_id[2]), 0]
    t.offset[:] = [float(offset[0]), float(offset[1]), float(offset[2]), 0.0]
    t.map_coord[:] = [int(map_coord[0]), int(map_coord[1]), 0, 0]
    return t


@dataclass
class Draw:
    """One `render_pass.draw(0..6, 0..splat_count)` of renderer.rs:466-590."""
    tile: Tile80
    gs_index: np.ndarray
    map_id: np.ndarray | None = None
    lod_id: np.ndarray | None = None
    # bookkeeping for the product-side tests (which static list / merged range this is)
    base: tuple | None = None           # (lod, tile, view) of the static base list, or None if merged
    corners: np.ndarray | None = None   # [4,3] tile corners (renderer.rs:472-494) for non-merged draws
    keep: list = field(default_factory=list)


def _pack_draws(draws):
    arr = (OrcDraw * max(1, len(draws)))()
    keep = []
    for i, d in enumerate(draws):
        gi = np.ascontiguousarray(d.gs_index, dtype=np.uint32)
        mi = np.ascontiguousarray(d.map_id, dtype=np.uint32) if d.map_id is not None else None
        li = np.ascontiguousarray(d.lod_id, dtype=np.uint32) if d.lod_id is not None else None
        keep += [gi, mi, li]
        arr[i].tile = d.tile
        arr[i].gs_index = gi.ctypes.data
        arr[i].map_id = mi.ctypes.data if mi is not None else None
        arr[i].lod_id = li.ctypes.data if li is not None else None
        arr[i].count = gi.shape[0]
    return arr, keep


def render(cam: Camera176, scene: Scene160, tex: np.ndarray, draws, width: int, height: int, *,
           height_map: np.ndarray | None = None, bg_rgba: np.ndarray | None = None,
           bg_depth: np.ndarray | None = None, order_mode: int = 0, n_threads: int = 0):
    """GSWTRenderer::render (renderer.rs:407-592) on an already culled draw list.
    Returns (image [H, W, 4] f32, stats dict)."""
    tex = np.ascontiguousarray(tex, dtype=np.uint32)
    arr, keep = _pack_draws(draws)
    out = np.zeros((height, width, 4), dtype=np.float32)
    hm = np.ascontiguousarray(height_map, dtype=np.float32) if height_map is not None else None
    bgc = np.ascontiguousarray(bg_rgba, dtype=np.float32) if bg_rgba is not None else None
    bgd = np.ascontiguousarray(bg_depth, dtype=np.float32) if bg_depth is not None else None
    st = OrcStats()
    rc = lib().orc_render(C.byref(cam), C.byref(scene), _ptr(tex), arr, len(draws), _ptr(hm),
                          hm.shape[1] if hm is not None else 0, hm.shape[0] if hm is not None else 0,
                          width, height, _ptr(bgc), _ptr(bgd), order_mode, n_threads, _ptr(out), C.byref(st))
    if rc != 0:
        raise RuntimeError(f"orc_render failed: {rc}")
    return out, {"n_instanced": st.n_instanced, "n_visible": st.n_visible, "n_pairs16": st.n_pairs16}


def compare_modes(cam: Camera176, scene: Scene160, tex: np.ndarray, draws, width: int, height: int, *, height_map=None,
                  bg_depth=None, n_threads: int = 0, strict_vs: bool = False):
    """Where the canonical sequence v2 and the strict (shader-text) evaluation DECIDE differently (orc_compare_modes):
    returns (mask [H, W] bool of the pixels holding at least one flipped coverage / cull decision, counts dict).
    strict_vs: the first side's vertex stage is the strict one too (the HIP path with GSWT_OPT_STRICT_VS); what is left to flip is
    the fragment stage's coverage test F4 against the exact quad interpolation."""
    tex = np.ascontiguousarray(tex, dtype=np.uint32)
    arr, keep = _pack_draws(draws)
    hm = np.ascontiguousarray(height_map, dtype=np.float32) if height_map is not None else None
    bgd = np.ascontiguousarray(bg_depth, dtype=np.float32) if bg_depth is not None else None
    mask = np.zeros((height, width), dtype=np.uint8)
    counts = np.zeros(4, dtype=np.uint64)
    rc = lib().orc_compare_modes2(C.byref(cam), C.byref(scene), _ptr(tex), arr, len(draws), _ptr(hm),
                                  hm.shape[1] if hm is not None else 0, hm.shape[0] if hm is not None else 0,
                                  width, height, _ptr(bgd), n_threads, 1 if strict_vs else 0, _ptr(mask), _ptr(counts))
    if rc != 0:
        raise RuntimeError(f"orc_compare_modes failed: {rc}")
    return mask.astype(bool), {"visible_in_one_mode": int(counts[0]), "decision_flips": int(counts[1]),
                               "marked_pixels": int(counts[2]), "visible_in_both": int(counts[3])}


def project_draws(cam: Camera176, scene: Scene160, tex: np.ndarray, draws, *, height_map=None) -> np.ndarray:
    """vs_main for every instance of every draw -> structured array SPLAT_DTYPE."""
    tex = np.ascontiguousarray(tex, dtype=np.uint32)
    arr, keep = _pack_draws(draws)
    n = sum(int(np.asarray(d.gs_index).shape[0]) for d in draws)
    out = np.zeros(max(n, 1), dtype=SPLAT_DTYPE)
    hm = np.ascontiguousarray(height_map, dtype=np.float32) if height_map is not None else None
    lib().orc_project_draws(C.byref(cam), C.byref(scene), _ptr(tex), arr, len(draws), _ptr(hm),
                            hm.shape[1] if hm is not None else 0, hm.shape[0] if hm is not None else 0, _ptr(out))
    return out[:n]


class Proxy224(C.Structure):
    """proxy.wgsl Uniforms / proxy.rs:470-511 (224 B)."""
    _fields_ = [("height_offset", C.c_float), ("tile_width", C.c_float), ("surface_type", C.c_uint32), ("width_scale", C.c_float),
                ("map_proxy", C.c_uint32), ("use_clip", C.c_uint32), ("clip_height", C.c_float), ("brightness", C.c_float),
                ("black_background", C.c_uint32), ("_pad0", C.c_uint32 * 3), ("view", C.c_float * 16), ("projection", C.c_float * 16),
                ("map_half_wh", C.c_uint32 * 2), ("center_coord", C.c_int32 * 2), ("height_map_scale", C.c_float * 4),
                ("cam_pos", C.c_float * 4)]


assert C.sizeof(Proxy224) == 224


def proxy_uniforms(cam: "Camera", *, map_proxy=1, height_offset=-0.5, tile_width=4.0, surface_type=0, width_scale=4.0, use_clip=0,
                   clip_height=0.0, brightness=1.0, black_background=0, map_half_wh=(0, 0), center_coord=(0, 0),
                   height_map_scale=(1.0, 1.0, 0.0)) -> Proxy224:
    """Uniforms::new, proxy.rs:489-511"""
    u = Proxy224()
    u.height_offset, u.tile_width, u.surface_type, u.width_scale = height_offset, tile_width, int(surface_type), width_scale
    u.map_proxy, u.use_clip, u.clip_height, u.brightness = int(map_proxy), int(use_clip), clip_height, brightness
    u.black_background = int(black_background)
    u.view[:] = [float(x) for x in cam.view]
    u.projection[:] = [float(x) for x in cam.projection]
    u.map_half_wh[:] = [int(map_half_wh[0]), int(map_half_wh[1])]
    u.center_coord[:] = [int(center_coord[0]), int(center_coord[1])]
    u.height_map_scale[:] = [float(height_map_scale[0]), float(height_map_scale[1]), float(height_map_scale[2]), 0.0]
    u.cam_pos[:] = [float(cam.position[0]), float(cam.position[1]), float(cam.position[2]), 0.0]
    return u


def skybox_render(cam: "Camera", faces: np.ndarray, width: int, height: int, equirectangular: int = 0) -> np.ndarray:
    """Skybox::render, skybox.rs:457-488.  faces [6, n, n, 4] f32 (+X -X +Y -Y +Z -Z)."""
    faces = np.ascontiguousarray(faces, dtype=np.float32)
    out = np.zeros((height, width, 4), dtype=np.float32)
    view = np.ascontiguousarray(cam.view, dtype=np.float32)
    L = lib()
    L.orc_skybox.restype = None
    L.orc_skybox.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.orc_skybox(_ptr(view), float(cam.projection[0]), float(cam.projection[5]), int(equirectangular), _ptr(faces), faces.shape[1],
                 width, height, _ptr(out))
    return out


def proxy_render(u: Proxy224, width: int, height: int, rgba: np.ndarray, depth: np.ndarray, mips, *, height_map=None, grid_dim=2048):
    """Proxy::render for ONE draw, proxy.rs:366-447; rgba / depth are updated in place (depth must start at 1.0)."""
    L = lib()
    L.orc_proxy.restype = None
    L.orc_proxy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_void_p, C.c_void_p]
    mips = [np.ascontiguousarray(m, dtype=np.float32) for m in mips]
    arr = (C.c_void_p * len(mips))(*[m.ctypes.data for m in mips])
    hm = np.ascontiguousarray(height_map, dtype=np.float32) if height_map is not None else None
    assert rgba.dtype == np.float32 and depth.dtype == np.float32 and rgba.flags.c_contiguous and depth.flags.c_contiguous
    L.orc_proxy(C.byref(u), int(grid_dim), _ptr(hm), hm.shape[1] if hm is not None else 0, hm.shape[0] if hm is not None else 0,
                arr, mips[0].shape[0], len(mips), width, height, _ptr(rgba), _ptr(depth))


def num_threads() -> int:
    return int(lib().orc_num_threads())


# --------------------------------------------------------------------------
# Merged-group list building (the per-sort-event hot loop of sort_tiles)
# --------------------------------------------------------------------------
def build_merged_value(pp: Preprocessed, members, view_id: int, head_lod: int):
    """wangtile.rs:595-670.  members: list of (map_index, lod, tile, status) in `from_vec` order,
    status in {None, ('changing', to_lower), ('spawning', f)}.  Returns dict with gs_index,
    gs_map_id, gs_lod_id (or None), single_lod_id, splat_count."""
    do_transition = any(m[3] is not None for m in members)                      # :599-609
    depth_vec, lod_ids, map_idx, offs = [], [], [], []
    for (m_mi, m_lod, m_tile, status) in members:                               # :615-641
        depth_vec.append(pp.raw_depth[m_lod][m_tile][view_id])
        lod_ids.append(m_lod)
        map_idx.append(m_mi)
        offs.append(int(pp.merge_offset[m_lod, m_tile]))
        if status is not None and status[0] == "changing":
            other = m_lod + 1 if status[1] else m_lod - 1
            depth_vec.append(pp.raw_depth[other][m_tile][view_id])
            lod_ids.append(other)
            map_idx.append(m_mi)
            offs.append(int(pp.merge_offset[other, m_tile]))
    seg, idx = sort_raw_depth_vec(depth_vec)                                    # :642
    gs_index = (idx + np.asarray(offs, dtype=np.int64)[seg]).astype(np.uint32)
    gs_map_id = np.asarray(map_idx, dtype=np.uint32)[seg]
    gs_lod_id = np.asarray(lod_ids, dtype=np.uint32)[seg] if do_transition else None
    return {"splat_count": int(gs_index.shape[0]), "gs_index": gs_index, "gs_map_id": gs_map_id,
            "gs_lod_id": gs_lod_id, "single_lod_id": -1 if do_transition else int(head_lod),
            "merge_from_vec": [m[0] for m in members]}
