"""ctypes bindings of libgswt_hip.so (include/gswt_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to
load, importing the renderer raises.  Build it with ``__graft_entry__.build()`` or
``make -C gswt_renderer_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSWT_HIP_LIB") or os.path.join(_HERE, "lib", "libgswt_hip.so")   # env override: kernel-variant sweeps

GSWT_OK = 0
GSWT_ERR_BAD_ARG = -1
GSWT_ERR_CAPACITY = -2
GSWT_ERR_HIP = -3
GSWT_ERR_STATE = -4
GSWT_ERR_RCCL = -6
GSWT_COMM_ID_BYTES = 128
GSWT_ORDER_REFERENCE = 0
GSWT_ORDER_DEPTH = 1
GSWT_OPT_NO_LOD_PREFILTER = 1
GSWT_OPT_DEBUG_VARYINGS = 2
GSWT_OPT_SEGMENT = 3
GSWT_DEFAULT_SEGMENT = 1536      # the library's default for GSWT_OPT_SEGMENT (tests that change it put this back)
GSWT_OPT_DEBUG_FLAGS = 4
GSWT_OPT_TIMING = 5
GSWT_OPT_PAIR_CAP = 6
GSWT_OPT_NO_MERGE_REUSE = 7
GSWT_OPT_DEFER_SWAP = 8
GSWT_OPT_GRAPH = 9
GSWT_OPT_STRICT_VS = 10
GSWT_OPT_DEPTH_PASSES = 12
GSWT_OPT_COMPOSITE = 13
GSWT_OPT_DEPTH_SORT = 14
GSWT_OPT_NO_CHUNK_CULL = 15
GSWT_OPT_ITEM_ORDER = 16
GSWT_SHARD_ROWS = 0
GSWT_SHARD_COLUMNS = 1


class CameraUniforms(C.Structure):
    """camera.rs:158-167"""
    _fields_ = [("projection", C.c_float * 16), ("view", C.c_float * 16), ("focal", C.c_float * 2),
                ("viewport", C.c_float * 2), ("htan_fov", C.c_float * 4), ("cam_pos", C.c_float * 4)]


class SceneUniforms(C.Structure):
    """renderer.rs:602-622"""
    _fields_ = [("splat_scale", C.c_float), ("tile_width", C.c_float), ("use_clip", C.c_uint32),
                ("clip_height", C.c_float), ("surface_type", C.c_uint32), ("sphere_radius", C.c_float),
                ("point_cloud_radius", C.c_float), ("transition_width_ratio", C.c_float),
                ("num_lod", C.c_uint32), ("draw_mode", C.c_uint32), ("map_half_wh", C.c_uint32 * 2),
                ("center_coord", C.c_int32 * 2), ("_pad0", C.c_uint32 * 2),
                ("transition_dist_vec", C.c_float * 16), ("height_map_scale", C.c_float * 4),
                ("scene_scale", C.c_float * 4)]


class TileUniforms(C.Structure):
    """renderer.rs:675-689"""
    _fields_ = [("single_draw", C.c_uint32), ("map_index", C.c_uint32), ("single_lod_id", C.c_int32),
                ("valid_lod_id", C.c_int32), ("changing", C.c_uint32), ("changing_to_lower", C.c_int32),
                ("_pad0", C.c_uint32 * 2), ("tile_id", C.c_uint32 * 4), ("offset", C.c_float * 4),
                ("map_coord", C.c_uint32 * 4)]


class BaseList(C.Structure):
    _fields_ = [("gs_index", C.c_void_p), ("gs_lod_id", C.c_void_p), ("splat_count", C.c_uint32),
                ("_pad", C.c_uint32)]


class Draw(C.Structure):
    _fields_ = [("tile", TileUniforms), ("merged", C.c_uint32), ("base_lod", C.c_uint32),
                ("base_tile", C.c_uint32), ("base_view", C.c_uint32), ("merged_offset", C.c_uint32),
                ("merged_count", C.c_uint32), ("merged_has_lod", C.c_uint32), ("cull_enable", C.c_uint32),
                ("corners", C.c_float * 12), ("lod", C.c_uint32), ("merged_group", C.c_uint32), ("_pad", C.c_uint32 * 2)]


class MergeGroup(C.Structure):
    _fields_ = [("view_id", C.c_uint32), ("first_member", C.c_uint32), ("n_members", C.c_uint32), ("_pad", C.c_uint32)]


class MergeMember(C.Structure):
    _fields_ = [("map_index", C.c_uint32), ("lod", C.c_uint32), ("tile", C.c_uint32), ("other_lod", C.c_int32)]


class RenderConfig(C.Structure):
    _fields_ = [("culling_dist", C.c_float), ("lod_enable_mask", C.c_uint32), ("order_mode", C.c_int32),
                ("transmittance_eps", C.c_float), ("shard_index", C.c_int32), ("shard_count", C.c_int32),
                ("shard_mode", C.c_int32), ("_pad", C.c_uint32)]


class Timings(C.Structure):
    _fields_ = [("ms_project", C.c_float), ("ms_scan", C.c_float), ("ms_emit", C.c_float),
                ("ms_sort", C.c_float), ("ms_ranges", C.c_float), ("ms_composite", C.c_float),
                ("ms_total", C.c_float), ("n_draws", C.c_uint32), ("n_instanced", C.c_uint64),
                ("n_visible", C.c_uint64), ("n_pairs", C.c_uint64), ("n_tiles", C.c_uint32),
                ("_pad", C.c_uint32), ("ms_composite_kernel", C.c_float), ("_pad2", C.c_float)]


class SortedTile(C.Structure):
    _fields_ = [("lod", C.c_uint32), ("tile", C.c_uint32), ("view_id", C.c_uint32), ("tile_offset", C.c_float * 3),
                ("map_index", C.c_uint32), ("map_coord", C.c_uint32 * 2), ("tile_center", C.c_float * 3),
                ("transition", C.c_int32), ("spawning_factor", C.c_float), ("has_corners", C.c_uint32),
                ("corners", C.c_float * 12), ("key_len", C.c_uint32), ("merged", C.c_uint32),
                ("merged_offset", C.c_uint32), ("merged_count", C.c_uint32), ("single_lod_id", C.c_int32),
                ("cache_hit", C.c_uint32), ("merged_group", C.c_uint32)]


class SortDataC(C.Structure):
    _fields_ = [("scene_id", C.c_uint32), ("n_tiles", C.c_uint32), ("tiles", C.POINTER(SortedTile)),
                ("n_merged", C.c_size_t), ("merged_gs_index", C.c_void_p), ("merged_map_id", C.c_void_p),
                ("merged_lod_id", C.c_void_p), ("n_groups", C.c_uint32), ("n_members", C.c_uint32),
                ("groups", C.c_void_p), ("members", C.c_void_p)]


class Cell(C.Structure):
    """gswt_cell: one map cell as update_tile_map leaves it (structure.rs:495-509)."""
    _fields_ = [("tile", C.c_uint32), ("has_corner", C.c_uint32), ("tile_offset", C.c_float * 3), ("tile_center", C.c_float * 3),
                ("to_local", C.c_float * 9), ("corner_pos", C.c_float * 12), ("corner_up", C.c_float * 12),
                ("edge_pos", C.c_float * 12), ("edge_normal", C.c_float * 12)]


class CellState(C.Structure):
    _fields_ = [("lod", C.c_uint32), ("transition", C.c_int32), ("spawning_factor", C.c_float), ("merge", C.c_uint32),
                ("merged_to", C.c_uint32)]


class WorkerConfig(C.Structure):
    _fields_ = [("map_w", C.c_uint32), ("map_h", C.c_uint32), ("half_w", C.c_uint32), ("half_h", C.c_uint32),
                ("n_lod", C.c_uint32), ("n_tile", C.c_uint32), ("n_view", C.c_uint32), ("tile_width", C.c_float),
                ("surface_type", C.c_uint32), ("tile_sort_type", C.c_uint32), ("merge_type", C.c_uint32),
                ("height_map_scale", C.c_float * 3), ("sphere_radius", C.c_float), ("lod_blending", C.c_uint32),
                ("lod_bbox_check", C.c_uint32), ("lod_transition_width_ratio", C.c_float), ("lod_dist_tolerance", C.c_float),
                ("merge_tile_dist", C.c_int32 * 2), ("merge_dot_threshold", C.c_float), ("merge_topk", C.c_uint32),
                ("hm_w", C.c_uint32), ("hm_h", C.c_uint32), ("height_map", C.c_void_p), ("lod_transition_dist", C.c_void_p),
                ("tile_center", C.c_void_p), ("tile_aabb", C.c_void_p), ("splat_count", C.c_void_p), ("presort_dirs", C.c_void_p),
                ("neighbors", C.c_void_p)]


assert C.sizeof(Cell) == 260 and C.sizeof(CellState) == 20

assert C.sizeof(CameraUniforms) == 176 and C.sizeof(SceneUniforms) == 160 and C.sizeof(TileUniforms) == 80

# every symbol include/gswt_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
class ProxyUniforms(C.Structure):
    """proxy.wgsl Uniforms / proxy.rs:470-511 (224 B)."""
    _fields_ = [("height_offset", C.c_float), ("tile_width", C.c_float), ("surface_type", C.c_uint32), ("width_scale", C.c_float),
                ("map_proxy", C.c_uint32), ("use_clip", C.c_uint32), ("clip_height", C.c_float), ("brightness", C.c_float),
                ("black_background", C.c_uint32), ("_pad0", C.c_uint32 * 3), ("view", C.c_float * 16), ("projection", C.c_float * 16),
                ("map_half_wh", C.c_uint32 * 2), ("center_coord", C.c_int32 * 2), ("height_map_scale", C.c_float * 4),
                ("cam_pos", C.c_float * 4)]


SYMBOLS = {
    "gswt_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "gswt_destroy": (None, [_P]),
    "gswt_last_error": (C.c_char_p, [_P]),
    "gswt_set_stream": (C.c_int, [_P, _P]),
    "gswt_set_option": (C.c_int, [_P, C.c_int, C.c_int]),
    "gswt_upload_scene": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_int, C.c_int, C.c_int]),
    "gswt_configure": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "gswt_set_draws": (C.c_int, [_P, _P, C.c_int, _P, _P, _P, C.c_size_t]),
    "gswt_upload_raw_depth": (C.c_int, [_P, _P, _P, _P]),
    "gswt_set_draws_merge_groups": (C.c_int, [_P, _P, C.c_int, _P, C.c_int, _P, C.c_int]),
    "gswt_debug_read_merged": (C.c_int, [_P, _P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gswt_render": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, C.c_int, _P, C.c_int]),
    "gswt_render_async": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, _P, _P, _P, C.POINTER(C.c_int)]),
    "gswt_render_wait": (C.c_int, [_P, C.c_int]),
    "gswt_render_fence": (C.c_int, [_P, C.c_int]),
    "gswt_frame_slots": (C.c_int, []),
    "gswt_skybox_configure": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "gswt_skybox_render": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "gswt_proxy_configure": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int]),
    "gswt_proxy_render": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, C.c_int]),
    "gswt_shard_rows": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "gswt_shard_rows_padded": (C.c_int, [C.c_int, C.c_int]),
    "gswt_unshard": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "gswt_unshard_mode": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "gswt_shard_cols_padded": (C.c_int, [C.c_int, C.c_int]),
    "gswt_comm_unique_id": (C.c_int, [_P]),
    "gswt_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "gswt_comm_destroy": (C.c_int, [_P]),
    "gswt_render_gather": (C.c_int, [_P, C.c_int, _P]),
    "gswt_group_init": (C.c_int, [_P, C.c_int]),
    "gswt_group_render_gather": (C.c_int, [_P, _P, _P, C.c_int]),
    "gswt_worker_create": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "gswt_worker_destroy": (None, [_P]),
    "gswt_worker_last_error": (C.c_char_p, [_P]),
    "gswt_worker_set_cells": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "gswt_worker_update_lod": (C.c_int, [_P, _P]),
    "gswt_worker_sort_tiles": (C.c_int, [_P, _P, _P]),
    "gswt_worker_fetch": (C.c_int, [_P]),
    "gswt_worker_read_cell_state": (C.c_int, [_P, _P, C.c_size_t]),
    "gswt_worker_read_sort": (C.c_int, [_P, _P]),
    "gswt_set_draws_from_worker": (C.c_int, [_P, _P]),
    "gswt_synchronize": (C.c_int, [_P]),
    "gswt_last_timings": (C.c_int, [_P, _P]),
    "gswt_debug_read_projected": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gswt_debug_read_ranges": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "gswt_debug_frame_times": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "gswt_debug_depth_stats": (C.c_int, [_P, C.POINTER(C.c_ulonglong)]),
    "gswt_debug_merge_stats": (C.c_int, [_P, _P]),
    "gswt_debug_merge_stats_deep": (C.c_int, [_P, _P]),
    "gswt_debug_totals": (C.c_int, [_P, _P, _P, C.c_uint32, C.c_uint32, _P, _P]),
    "gswt_debug_sort": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int]),
    "gswt_debug_tile_depth_sort": (C.c_int, [_P, _P, C.c_size_t, _P, _P, C.c_size_t, C.POINTER(C.c_int)]),
    "gswt_debug_graph_stats": (C.c_int, [_P, _P]),
}

_lib = None


def load():
    """Load libgswt_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the GSWT hot path has no CPU fallback. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C gswt_renderer_amd/csrc`.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)      # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
