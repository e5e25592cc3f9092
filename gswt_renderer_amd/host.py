"""Host-side mirror of the reference's ``scene::Scene`` loader and ``wangtile::WangTile`` worker
(ctypes over libgswt_host.so, include/gswt_host.h).  Names and argument meaning follow the
reference (wangtile.rs:41,340,349,434,476,692); reference panics surface as ``GSWTHostError``.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _lib as L

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libgswt_host.so")

SORT_DISTANCE, SORT_VIEWPORT, SORT_OBJECT, SORT_GRAPH = 0, 1, 2, 3
MERGE_NONE, MERGE_AXIS, MERGE_EDGE = 0, 1, 2
SURFACE_NONE, SURFACE_HEIGHTMAP, SURFACE_SPHERE = 0, 1, 2
HMAP_TEXTURE, HMAP_RANDOM, HMAP_SLOPEX, HMAP_SLOPEY, HMAP_DUALSLOPE = 0, 1, 2, 3, 4
TR_NONE, TR_SPAWNING, TR_CHANGING_HIGHER, TR_CHANGING_LOWER = 0, 1, 2, 3


class GSWTHostError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gswt host error {code}: {msg}")
        self.code = code


class UserData(C.Structure):
    """structure.rs:15-65 (fields the worker reads).  Defaults = GUI defaults, structure.rs:121-137."""
    _fields_ = [("tile_map_half_wh", C.c_uint32 * 2), ("center_option", C.c_uint32), ("update_distance2", C.c_float),
                ("tile_width", C.c_float), ("tile_sort_type", C.c_uint32), ("surface_type", C.c_uint32),
                ("height_map_wh", C.c_uint32 * 2), ("height_map_type", C.c_uint32), ("height_map_scale", C.c_float * 3),
                ("height_tex", C.c_void_p), ("height_tex_wh", C.c_uint32 * 2), ("sphere_radius", C.c_float),
                ("lod_max_dist", C.c_float), ("lod_blending", C.c_uint32), ("lod_transition_width_ratio", C.c_float),
                ("lod_bbox_check", C.c_uint32), ("lod_dist_tolerance", C.c_float), ("merge_type", C.c_uint32),
                ("merge_tile_dist", C.c_int32 * 2), ("merge_dot_threshold", C.c_float), ("merge_topk", C.c_uint32),
                ("use_cache", C.c_uint32), ("cache_size", C.c_uint32), ("reset_rng", C.c_uint32), ("always_sort", C.c_uint32)]


def user_data(**kw) -> UserData:
    u = UserData()
    u.tile_map_half_wh[:] = kw.pop("tile_map_half_wh", (48, 48))
    u.center_option = kw.pop("center_option", 1)
    u.update_distance2 = kw.pop("update_distance2", 1.0)
    u.tile_width = kw.pop("tile_width", 4.0)
    u.tile_sort_type = kw.pop("tile_sort_type", SORT_GRAPH)
    u.surface_type = kw.pop("surface_type", SURFACE_HEIGHTMAP)
    u.height_map_wh[:] = kw.pop("height_map_wh", (10, 10))
    u.height_map_type = kw.pop("height_map_type", HMAP_RANDOM)
    u.height_map_scale[:] = kw.pop("height_map_scale", (1.0, 1.0, 1.0))
    u.sphere_radius = kw.pop("sphere_radius", 20.0)
    u.lod_max_dist = kw.pop("lod_max_dist", 96.0 * 4.0)
    u.lod_blending = int(kw.pop("lod_blending", True))
    u.lod_transition_width_ratio = kw.pop("lod_transition_width_ratio", 0.05)
    u.lod_bbox_check = int(kw.pop("lod_bbox_check", True))
    u.lod_dist_tolerance = kw.pop("lod_dist_tolerance", 0.0)
    u.merge_type = kw.pop("merge_type", MERGE_EDGE)
    u.merge_tile_dist[:] = kw.pop("merge_tile_dist", (3, 10))
    u.merge_dot_threshold = kw.pop("merge_dot_threshold", 0.2)
    u.merge_topk = kw.pop("merge_topk", 100)
    u.use_cache = int(kw.pop("use_cache", True))
    u.cache_size = kw.pop("cache_size", 1024)
    u.reset_rng = int(kw.pop("reset_rng", True))
    u.always_sort = int(kw.pop("always_sort", False))
    if kw:
        raise TypeError(f"unknown UserData fields {sorted(kw)}")
    return u


class Configured(C.Structure):
    _fields_ = [("tile_map_wh", C.c_uint32 * 2), ("height_map_wh", C.c_uint32 * 2), ("height_map", C.c_void_p),
                ("lod_transition_dist", C.c_float * 16), ("n_lod", C.c_uint32), ("n_tile", C.c_uint32),
                ("n_view", C.c_uint32)]


class SceneData(C.Structure):
    """structure.rs:466-474"""
    _fields_ = [("scene_id", C.c_uint32), ("splat_count", C.c_uint64), ("blending_splat_count", C.c_uint64),
                ("center_coord", C.c_int32 * 2), ("lod_splat_count", C.c_uint64 * 16),
                ("lod_instance_count", C.c_uint64 * 16)]


SortedTile = L.SortedTile        # SortData records live in gswt_hip.h: the device-side worker stages produce them too
SortDataC = L.SortDataC


class Preload(C.Structure):
    _fields_ = [("tex_data", C.c_void_p), ("n_splats", C.c_size_t), ("n_lod", C.c_int), ("n_tile", C.c_int),
                ("n_view", C.c_int), ("lists", C.POINTER(L.BaseList))]


_P = C.c_void_p
HOST_SYMBOLS = {
    "gswt_host_last_error": (C.c_char_p, []),
    "gswt_tileset_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_P)]),
    "gswt_tileset_destroy": (None, [_P]),
    "gswt_tileset_set_ply": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t]),
    "gswt_tileset_set_vertices": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t]),
    "gswt_tileset_set_rows": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t]),
    "gswt_load_scene_zip": (C.c_int, [C.c_char_p, C.POINTER(_P)]),
    "gswt_load_scene_zip_mem": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "gswt_tileset_dims": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gswt_tileset_splat_count": (C.c_size_t, [_P, C.c_int, C.c_int]),
    "gswt_tileset_rows": (_P, [_P, C.c_int, C.c_int]),
    "gswt_generate_texture": (C.c_int, [_P, C.c_size_t, _P]),
    "gswt_sort_raw_depth": (C.c_int, [_P, C.c_size_t, _P]),
    "gswt_pack_half_2x16": (C.c_uint32, [C.c_float, C.c_float]),
    "gswt_camera_uniforms_from_camera": (C.c_int, [_P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _P, _P]),
    "gswt_wang_new": (C.c_int, [_P, C.POINTER(_P)]),
    "gswt_wang_destroy": (None, [_P]),
    "gswt_wang_preload": (C.c_int, [_P, _P]),
    "gswt_wang_tile_base": (C.c_int, [_P, C.c_int, _P, _P]),
    "gswt_wang_lod_avg_scale": (C.c_int, [_P, _P, C.c_int]),
    "gswt_wang_raw_depth": (_P, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_size_t)]),
    "gswt_wang_merge_offset": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_uint32)]),
    "gswt_wang_configure": (C.c_int, [_P, _P, _P]),
    "gswt_wang_check_update": (C.c_int, [_P, _P]),
    "gswt_wang_build_tiles": (C.c_int, [_P, _P, _P]),
    "gswt_wang_sort_tiles": (C.c_int, [_P, _P, _P, _P]),
    "gswt_wang_set_device_merge": (C.c_int, [_P, C.c_int]),
    "gswt_wang_raw_depth_tables": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    "gswt_wang_worker_config": (C.c_int, [_P, _P]),
    "gswt_wang_export_cells": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "gswt_wang_export_cell_state": (C.c_int, [_P, _P, C.c_size_t]),
    "gswt_wang_get_tile_ids": (C.c_int, [_P, _P, C.c_size_t]),
    "gswt_wang_set_tile_ids": (C.c_int, [_P, _P, C.c_size_t]),
    "gswt_renderer_build_draws": (C.c_int, [_P, _P]),
    "gswt_scene_uniforms_from_data": (C.c_int, [_P, _P, _P, C.c_float, _P, C.c_float, _P]),
}

_host = None


def load():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError(f"{HOST_LIB_PATH} not found; run __graft_entry__.build() or make -C gswt_renderer_amd/csrc")
        lib = C.CDLL(HOST_LIB_PATH)
        for name, (res, args) in HOST_SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _host = lib
    return _host


def _check(rc):
    if rc < 0:
        raise GSWTHostError(rc, load().gswt_host_last_error().decode("utf-8", "replace"))
    return rc


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def pack_half_2x16(x: float, y: float) -> int:
    return int(load().gswt_pack_half_2x16(x, y))


def generate_texture(rows: np.ndarray) -> np.ndarray:
    rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1, 32)
    tex = np.zeros((rows.shape[0], 8), dtype=np.uint32)
    _check(load().gswt_generate_texture(_ptr(rows), rows.shape[0], _ptr(tex)))
    return tex


def sort_raw_depth(depths: np.ndarray) -> np.ndarray:
    d = np.ascontiguousarray(depths, dtype=np.int32)
    out = np.zeros(d.shape[0], dtype=np.uint32)
    _check(load().gswt_sort_raw_depth(_ptr(d), d.shape[0], _ptr(out)))
    return out


def camera_uniforms(pos, target, up, fovy_deg, z_near, z_far, width, height):
    """Camera::new_perspective + CameraUniforms::from_camera -> (CameraUniforms, view_proj[16])."""
    cu = L.CameraUniforms()
    vp = np.zeros(16, dtype=np.float32)
    _check(load().gswt_camera_uniforms_from_camera(_f3(pos), _f3(target), _f3(up), fovy_deg, z_near, z_far, width, height,
                                                   C.byref(cu), _ptr(vp)))
    return cu, vp


def default_camera(width, height):
    """state.rs:114-122"""
    return camera_uniforms((0, 0, 5), (0, 1, 5), (0, 0, 1), 45.0, 0.1, 2400.0, width, height)


class TileSet:
    """Vec<Vec<Scene>> [n_lod][n_tile]."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_vertices(cls, verts):
        """verts[lod][tile] = [n, 62] f32 PLY vertex records (Scene::load on each)."""
        lib = load()
        h = C.c_void_p()
        _check(lib.gswt_tileset_create(len(verts), len(verts[0]), C.byref(h)))
        ts = cls(h)
        for l, lod in enumerate(verts):
            for t, v in enumerate(lod):
                v = np.ascontiguousarray(v, dtype=np.float32).reshape(-1, 62)
                _check(lib.gswt_tileset_set_vertices(h, l, t, _ptr(v), v.shape[0]))
        return ts

    @classmethod
    def from_zip(cls, path_or_bytes):
        """load_scene_zip, scene.rs:1030-1141"""
        lib = load()
        h = C.c_void_p()
        if isinstance(path_or_bytes, (bytes, bytearray)):
            buf = np.frombuffer(bytes(path_or_bytes), dtype=np.uint8)
            _check(lib.gswt_load_scene_zip_mem(_ptr(buf), buf.shape[0], C.byref(h)))
        else:
            _check(lib.gswt_load_scene_zip(str(path_or_bytes).encode(), C.byref(h)))
        return cls(h)

    def dims(self):
        a, b = C.c_int(), C.c_int()
        _check(load().gswt_tileset_dims(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def rows(self, lod, tile) -> np.ndarray:
        n = load().gswt_tileset_splat_count(self._h, lod, tile)
        p = load().gswt_tileset_rows(self._h, lod, tile)
        if n == 0:
            return np.zeros((0, 32), dtype=np.uint8)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n, 32)).copy()

    def close(self):
        if self._h:
            load().gswt_tileset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class SortData:
    """structure.rs:488-493, flattened."""
    tiles: list            # list[SortedTile] copies, back-to-front
    merged_gs_index: np.ndarray
    merged_map_id: np.ndarray
    merged_lod_id: np.ndarray
    draws: list            # list[L.Draw] built by gswt_renderer_build_draws
    groups: list = None    # list[L.MergeGroup] (copies)
    members: list = None   # list[L.MergeMember] (copies)


class WangTile:
    """wangtile.rs:18-39"""

    def __init__(self, tileset: TileSet):
        """WangTile::new (takes ownership of the tile set) -> preprocess."""
        lib = load()
        h = C.c_void_p()
        th, tileset._h = tileset._h, None      # gswt_wang_new consumes the tile set, also when it fails
        _check(lib.gswt_wang_new(th, C.byref(h)))
        self._h = h
        self._lib = lib
        self.user = None
        self.conf = None
        self.scene_data = None
        p = Preload()
        _check(lib.gswt_wang_preload(h, C.byref(p)))
        self.n_tiles = (p.n_lod, p.n_tile, p.n_view)
        self._preload = p

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gswt_wang_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- preload() ------------------------------------------------------------------
    def preload(self):
        """PreloadData: (tex_data [U,8] u32 view, gs_index[lod][tile][view], gs_lod_id[...])."""
        p = self._preload
        tex = np.ctypeslib.as_array(C.cast(p.tex_data, C.POINTER(C.c_uint32)), shape=(p.n_splats, 8))
        gi, li = [], []
        k = 0
        for l in range(p.n_lod):
            gi.append([]); li.append([])
            for t in range(p.n_tile):
                gi[l].append([]); li[l].append([])
                for v in range(p.n_view):
                    b = p.lists[k]
                    k += 1
                    n = b.splat_count
                    gi[l][t].append(np.ctypeslib.as_array(C.cast(b.gs_index, C.POINTER(C.c_uint32)), shape=(n,)))
                    li[l][t].append(np.ctypeslib.as_array(C.cast(b.gs_lod_id, C.POINTER(C.c_uint32)), shape=(n,)))
        return tex, gi, li

    def set_device_merge(self, enable: bool):
        """sort_tiles then only describes merged groups; libgswt_hip builds their lists (gswt_set_draws_merge_groups)."""
        _check(self._lib.gswt_wang_set_device_merge(self._h, 1 if enable else 0))

    def upload_raw_depth_to(self, renderer):
        """gswt_upload_raw_depth with this WangTile's raw-depth tables (needed for device-side merged lists)."""
        ptrs, cnts, offs = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(self._lib.gswt_wang_raw_depth_tables(self._h, C.byref(ptrs), C.byref(cnts), C.byref(offs)))
        renderer._check(renderer._lib.gswt_upload_raw_depth(renderer._h, ptrs, cnts, offs))

    def upload_to(self, renderer):
        """GSWTRenderer::new(preload_data): hand the PreloadData pointers straight to libgswt_hip."""
        p = self._preload
        renderer._check(renderer._lib.gswt_upload_scene(renderer._h, p.tex_data, p.n_splats, p.lists, p.n_lod, p.n_tile, p.n_view))
        renderer.n_lists = self.n_tiles

    def tile_base(self, tile):
        c = (C.c_float * 3)()
        a = (C.c_float * 6)()
        _check(self._lib.gswt_wang_tile_base(self._h, tile, c, a))
        return np.array(c[:], dtype=np.float32), np.array(a[:], dtype=np.float32).reshape(2, 3)

    def lod_avg_scale(self):
        out = (C.c_float * 16)()
        n = _check(self._lib.gswt_wang_lod_avg_scale(self._h, out, 16))
        return np.array(out[:n], dtype=np.float32)

    def raw_depth(self, lod, tile, view):
        n = C.c_size_t()
        p = self._lib.gswt_wang_raw_depth(self._h, lod, tile, view, C.byref(n))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), shape=(n.value,)).copy()

    def merge_offset(self, lod, tile):
        o = C.c_uint32()
        _check(self._lib.gswt_wang_merge_offset(self._h, lod, tile, C.byref(o)))
        return o.value

    # -- configure / build_tiles / sort_tiles ---------------------------------------------
    def configure(self, user: UserData, height_tex: np.ndarray | None = None) -> Configured:
        self._tex_keep = None
        if height_tex is not None:
            ht = np.ascontiguousarray(height_tex, dtype=np.float32)
            self._tex_keep = ht
            user.height_tex = ht.ctypes.data
            user.height_tex_wh[:] = (ht.shape[1], ht.shape[0])
        conf = Configured()
        _check(self._lib.gswt_wang_configure(self._h, C.byref(user), C.byref(conf)))
        self.user, self.conf = user, conf
        return conf

    def height_map(self) -> np.ndarray | None:
        c = self.conf
        if not c or not c.height_map:
            return None
        return np.ctypeslib.as_array(C.cast(c.height_map, C.POINTER(C.c_float)), shape=(c.height_map_wh[1], c.height_map_wh[0])).copy()

    def check_update(self, cam_pos) -> bool:
        return bool(_check(self._lib.gswt_wang_check_update(self._h, _f3(cam_pos))))

    def build_tiles(self, cam_pos) -> SceneData:
        sd = SceneData()
        _check(self._lib.gswt_wang_build_tiles(self._h, _f3(cam_pos), C.byref(sd)))
        self.scene_data = sd
        return sd

    def tile_ids(self) -> np.ndarray:
        n = self.conf.tile_map_wh[0] * self.conf.tile_map_wh[1]
        ids = np.zeros(n, dtype=np.uint32)
        _check(self._lib.gswt_wang_get_tile_ids(self._h, _ptr(ids), n))
        return ids

    def set_tile_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        _check(self._lib.gswt_wang_set_tile_ids(self._h, _ptr(ids), ids.shape[0]))

    def sort_tiles(self, cam_pos, view_proj) -> SortData:
        vp = np.ascontiguousarray(view_proj, dtype=np.float32)
        sd = SortDataC()
        _check(self._lib.gswt_wang_sort_tiles(self._h, _f3(cam_pos), _ptr(vp), C.byref(sd)))
        draws = (L.Draw * max(1, sd.n_tiles))()
        _check(self._lib.gswt_renderer_build_draws(C.byref(sd), draws))
        tiles = [SortedTile.from_buffer_copy(bytes(sd.tiles[i])) for i in range(sd.n_tiles)]

        def arr(p):
            if sd.n_merged == 0:
                return np.zeros(0, dtype=np.uint32)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(sd.n_merged,)).copy()
        groups = [L.MergeGroup.from_buffer_copy(C.string_at(sd.groups + i * C.sizeof(L.MergeGroup), C.sizeof(L.MergeGroup)))
                  for i in range(sd.n_groups)]
        members = [L.MergeMember.from_buffer_copy(C.string_at(sd.members + i * C.sizeof(L.MergeMember), C.sizeof(L.MergeMember)))
                   for i in range(sd.n_members)]

        def arr(p):
            if sd.n_merged == 0 or not p:
                return np.zeros(0, dtype=np.uint32)
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(sd.n_merged,)).copy()
        return SortData(tiles, arr(sd.merged_gs_index), arr(sd.merged_map_id), arr(sd.merged_lod_id),
                        [draws[i] for i in range(sd.n_tiles)], groups, members)

    # -- hand-over to the device-side worker stages (gswt_worker_* of libgswt_hip) -------------
    def worker_config(self) -> L.WorkerConfig:
        cfg = L.WorkerConfig()
        _check(self._lib.gswt_wang_worker_config(self._h, C.byref(cfg)))
        return cfg

    def export_cells(self):
        """-> (gswt_cell C array, n, center_coord int32[2]) of the map build_tiles left."""
        n = self.conf.tile_map_wh[0] * self.conf.tile_map_wh[1]
        cells = (L.Cell * n)()
        cc = (C.c_int32 * 2)()
        _check(self._lib.gswt_wang_export_cells(self._h, cells, n, cc))
        return cells, n, cc

    def export_cell_state(self) -> np.ndarray:
        """Per cell (lod, transition, spawning bits, merge, merged_to) as a [n, 5] uint32 array (floats as bit patterns)."""
        n = self.conf.tile_map_wh[0] * self.conf.tile_map_wh[1]
        st = (L.CellState * n)()
        _check(self._lib.gswt_wang_export_cell_state(self._h, st, n))
        return np.frombuffer(bytes(st), dtype=np.uint32).reshape(n, 5).copy()

    def sort_tiles_raw(self, cam_pos, view_proj):
        """sort_tiles + gswt_renderer_build_draws without per-tile Python objects (the worker thread of a frame loop calls
        this once per sort event): -> (draws C array, n_draws, groups C array, n_groups, members C array, n_members); the
        arrays are copies owned by the caller, so the next sort_tiles may run while they are being uploaded."""
        vp = np.ascontiguousarray(view_proj, dtype=np.float32)
        sd = SortDataC()
        _check(self._lib.gswt_wang_sort_tiles(self._h, _f3(cam_pos), _ptr(vp), C.byref(sd)))
        draws = (L.Draw * max(1, sd.n_tiles))()
        _check(self._lib.gswt_renderer_build_draws(C.byref(sd), draws))
        groups = (L.MergeGroup * max(1, sd.n_groups))()
        members = (L.MergeMember * max(1, sd.n_members))()
        if sd.n_groups:
            C.memmove(groups, sd.groups, sd.n_groups * C.sizeof(L.MergeGroup))
        if sd.n_members:
            C.memmove(members, sd.members, sd.n_members * C.sizeof(L.MergeMember))
        return draws, int(sd.n_tiles), groups, int(sd.n_groups), members, int(sd.n_members)

    def scene_uniforms(self, *, splat_scale=1.0, scene_scale=(1.0, 1.0, 1.0), height_map_scale_v=1.0) -> L.SceneUniforms:
        """SceneUniforms::from_data, renderer.rs:631-672"""
        su = L.SceneUniforms()
        _check(self._lib.gswt_scene_uniforms_from_data(C.byref(self.user), C.byref(self.conf), C.byref(self.scene_data),
                                                       splat_scale, _f3(scene_scale), height_map_scale_v, C.byref(su)))
        return su
