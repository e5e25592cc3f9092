"""Host-side mirror of the reference's ``renderer::GSWTRenderer`` over the C ABI.

``GSWTRenderer.new / configure / render`` keep the reference's names and argument
meaning (renderer.rs:31,351,407); device work is done by libgswt_hip.so.  There is no
CPU fallback: constructing a renderer without the HIP library or a GPU raises.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


class GSWTError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"gswt error {code}: {msg}")
        self.code = code


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _shard_args(shard):
    """shard = (index, count) -> interleaved tile rows; (index, count, "cols") -> contiguous tile-column bands."""
    mode = L.GSWT_SHARD_COLUMNS if len(shard) > 2 and shard[2] in ("cols", "columns", L.GSWT_SHARD_COLUMNS) and shard[2] != 0 else L.GSWT_SHARD_ROWS
    return int(shard[0]), int(shard[1]), mode


def make_draw(tile: L.TileUniforms, *, base=None, merged_range=None, merged_has_lod=False, corners=None,
              lod=None) -> L.Draw:
    """One draw of the loop renderer.rs:466-591.  base = (lod, tile, view) of a static list or
    merged_range = (offset, count) into the merged arrays."""
    d = L.Draw()
    d.tile = tile
    if merged_range is not None:
        d.merged = 1
        d.merged_offset, d.merged_count = int(merged_range[0]), int(merged_range[1])
        d.merged_has_lod = 1 if merged_has_lod else 0
    else:
        d.merged = 0
        d.base_lod, d.base_tile, d.base_view = int(base[0]), int(base[1]), int(base[2])
    if corners is not None:
        d.cull_enable = 1
        d.corners[:] = [float(x) for x in np.asarray(corners, dtype=np.float32).reshape(12)]
    d.lod = int(tile.tile_id[0] if lod is None else lod)
    return d


class GSWTRenderer:
    """renderer.rs:10-29.  Owns the device context and all HBM buffers."""

    def __init__(self, device_id: int = 0):
        self._lib = L.load()
        h = C.c_void_p()
        rc = self._lib.gswt_create(device_id, C.byref(h))
        if rc != L.GSWT_OK:
            raise GSWTError(rc, "gswt_create failed (no HIP device?)")
        self._h = h
        self.n_lists = (0, 0, 0)

    # -- lifecycle ---------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.gswt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != L.GSWT_OK:
            raise GSWTError(rc, self._lib.gswt_last_error(self._h).decode())

    def set_option(self, key: int, value: int):
        self._check(self._lib.gswt_set_option(self._h, key, value))

    def set_stream(self, hip_stream: int):
        self._check(self._lib.gswt_set_stream(self._h, C.c_void_p(hip_stream)))

    # -- GSWTRenderer::new (renderer.rs:31): PreloadData upload ------------------------
    def upload_scene(self, tex_data: np.ndarray, gs_index, gs_lod_id):
        """tex_data [U, 8] u32 (Scene.tex_data); gs_index / gs_lod_id nested [lod][tile][view]."""
        tex = np.ascontiguousarray(tex_data, dtype=np.uint32).reshape(-1, 8)
        n_lod, n_tile, n_view = len(gs_index), len(gs_index[0]), len(gs_index[0][0])
        arr = (L.BaseList * (n_lod * n_tile * n_view))()
        keep = []
        i = 0
        for l in range(n_lod):
            for t in range(n_tile):
                for v in range(n_view):
                    gi = np.ascontiguousarray(gs_index[l][t][v], dtype=np.uint32)
                    li = np.ascontiguousarray(gs_lod_id[l][t][v], dtype=np.uint32)
                    keep += [gi, li]
                    arr[i].gs_index, arr[i].gs_lod_id, arr[i].splat_count = gi.ctypes.data, li.ctypes.data, gi.shape[0]
                    i += 1
        self._check(self._lib.gswt_upload_scene(self._h, _ptr(tex), tex.shape[0], arr, n_lod, n_tile, n_view))
        self.n_lists = (n_lod, n_tile, n_view)

    # -- GSWTRenderer::configure (renderer.rs:351) -----------------------------------
    def configure(self, height_map: np.ndarray | None):
        if height_map is None:
            self._check(self._lib.gswt_configure(self._h, None, 0, 0))
        else:
            hm = np.ascontiguousarray(height_map, dtype=np.float32)
            self._check(self._lib.gswt_configure(self._h, _ptr(hm), hm.shape[1], hm.shape[0]))

    # -- SortData swap-in (state.rs:361-376) -------------------------------------------
    def set_draws(self, draws, merged_gs_index=None, merged_map_id=None, merged_lod_id=None):
        arr = (L.Draw * max(1, len(draws)))(*draws)
        gi = np.ascontiguousarray(merged_gs_index, dtype=np.uint32) if merged_gs_index is not None else None
        mi = np.ascontiguousarray(merged_map_id, dtype=np.uint32) if merged_map_id is not None else None
        li = np.ascontiguousarray(merged_lod_id, dtype=np.uint32) if merged_lod_id is not None else None
        n = 0 if gi is None else gi.shape[0]
        self._check(self._lib.gswt_set_draws(self._h, arr, len(draws), _ptr(gi), _ptr(mi), _ptr(li), n))

    def set_draws_merge_groups(self, draws, groups_ptr, n_groups: int, members_ptr, n_members: int):
        """SortData swap-in with the merged lists built on the device (groups / members: C arrays of
        gswt_merge_group / gswt_merge_member, e.g. straight from libgswt_host's sort data)."""
        arr = (L.Draw * max(1, len(draws)))(*draws)
        self._check(self._lib.gswt_set_draws_merge_groups(self._h, arr, len(draws), groups_ptr, n_groups, members_ptr, n_members))

    def set_draws_merge_groups_raw(self, draws_arr, n_draws: int, groups_arr, n_groups: int, members_arr, n_members: int):
        """The same from C arrays (WangTile.sort_tiles_raw): no per-draw Python work on the render thread."""
        self._check(self._lib.gswt_set_draws_merge_groups(self._h, draws_arr, n_draws, groups_arr, n_groups, members_arr, n_members))

    def graph_stats(self):
        """GSWT_OPT_GRAPH bookkeeping: [frames replayed through hipGraphLaunch, graphs (re)built, kernel nodes updated]."""
        a = (C.c_ulonglong * 3)()
        self._check(self._lib.gswt_debug_graph_stats(self._h, a))
        return [int(x) for x in a]

    def merge_stats(self):
        """(merged groups sorted, merged groups copied from the previous sort event) since the ctx was created."""
        out = (C.c_ulonglong * 2)()
        self._check(self._lib.gswt_debug_merge_stats(self._h, out))
        return int(out[0]), int(out[1])

    def merge_stats_deep(self):
        """Of the copied groups, how many came from a sort event older than the previous one (the lists of the last 9 events are kept)."""
        out = C.c_ulonglong(0)
        self._check(self._lib.gswt_debug_merge_stats_deep(self._h, C.byref(out)))
        return int(out.value)

    def read_merged(self):
        n = C.c_size_t(0)
        self._check(self._lib.gswt_debug_read_merged(self._h, None, None, 0, C.byref(n)))
        a = np.zeros(max(1, n.value), dtype=np.uint32)
        b = np.zeros(max(1, n.value), dtype=np.uint32)
        self._check(self._lib.gswt_debug_read_merged(self._h, _ptr(a), _ptr(b), a.shape[0], C.byref(n)))
        return a[:n.value], b[:n.value]

    # -- GSWTRenderer::render (renderer.rs:407) ---------------------------------------
    def render(self, camera, scene, width: int, height: int, *, culling_dist: float = 1.0,
               lod_enable_mask: int = 0xFFFFFFFF, order_mode: int = L.GSWT_ORDER_REFERENCE,
               transmittance_eps: float = 0.0, shard=(0, 1), bg_rgba=None, bg_depth=None,
               out_device_ptr: int | None = None, bg_on_device: bool = False):
        """camera / scene: 176 / 160-byte uniform blocks (any ctypes struct or bytes of that layout).
        Returns the image [rows, W, 4] f32 on the host, or None when out_device_ptr is given."""
        cam = (C.c_char * 176).from_buffer_copy(bytes(camera))
        sc = (C.c_char * 160).from_buffer_copy(bytes(scene))
        cfg = L.RenderConfig()
        cfg.culling_dist, cfg.lod_enable_mask, cfg.order_mode = culling_dist, lod_enable_mask & 0xFFFFFFFF, order_mode
        cfg.transmittance_eps = transmittance_eps
        cfg.shard_index, cfg.shard_count, cfg.shard_mode = _shard_args(shard)
        rows, out_w = height, width
        if cfg.shard_count > 1 and cfg.shard_mode == L.GSWT_SHARD_COLUMNS:
            out_w = self._lib.gswt_shard_cols_padded(width, cfg.shard_count)
        elif cfg.shard_count > 1:
            rows = self._lib.gswt_shard_rows_padded(height, cfg.shard_count)
        if bg_on_device:
            bgc = C.c_void_p(bg_rgba) if bg_rgba else None
            bgd = C.c_void_p(bg_depth) if bg_depth else None
        else:
            bgc_a = np.ascontiguousarray(bg_rgba, dtype=np.float32) if bg_rgba is not None else None
            bgd_a = np.ascontiguousarray(bg_depth, dtype=np.float32) if bg_depth is not None else None
            bgc, bgd = _ptr(bgc_a), _ptr(bgd_a)
        if out_device_ptr is not None:
            self._check(self._lib.gswt_render(self._h, cam, sc, C.byref(cfg), width, height, bgc, bgd,
                                              1 if bg_on_device else 0, C.c_void_p(out_device_ptr), 1))
            return None
        out = np.empty((rows, out_w, 4), dtype=np.float32)
        self._check(self._lib.gswt_render(self._h, cam, sc, C.byref(cfg), width, height, bgc, bgd,
                                          1 if bg_on_device else 0, _ptr(out), 0))
        return out

    def render_async(self, camera, scene, width: int, height: int, out_device_ptr: int, *, culling_dist: float = 1.0,
                     lod_enable_mask: int = 0xFFFFFFFF, order_mode: int = L.GSWT_ORDER_REFERENCE,
                     transmittance_eps: float = 0.0, shard=(0, 1), bg_rgba_ptr: int = 0, bg_depth_ptr: int = 0) -> int:
        """Queues a frame (device pointers only) and returns a ticket for render_wait."""
        cam = (C.c_char * 176).from_buffer_copy(bytes(camera))
        sc = (C.c_char * 160).from_buffer_copy(bytes(scene))
        cfg = L.RenderConfig()
        cfg.culling_dist, cfg.lod_enable_mask, cfg.order_mode = culling_dist, lod_enable_mask & 0xFFFFFFFF, order_mode
        cfg.transmittance_eps = transmittance_eps
        cfg.shard_index, cfg.shard_count, cfg.shard_mode = _shard_args(shard)
        ticket = C.c_int(-1)
        self._check(self._lib.gswt_render_async(self._h, cam, sc, C.byref(cfg), width, height,
                                                C.c_void_p(bg_rgba_ptr) if bg_rgba_ptr else None,
                                                C.c_void_p(bg_depth_ptr) if bg_depth_ptr else None,
                                                C.c_void_p(out_device_ptr), C.byref(ticket)))
        return ticket.value

    def frame_slots(self) -> int:
        """Frames that may be in flight at once (render_async tickets)."""
        return int(self._lib.gswt_frame_slots())

    def render_wait(self, ticket: int):
        self._check(self._lib.gswt_render_wait(self._h, ticket))

    def render_fence(self, ticket: int):
        """Device-side: work submitted to the ctx stream afterwards waits for this frame."""
        self._check(self._lib.gswt_render_fence(self._h, ticket))

    # -- background passes (state.rs:384-392) ------------------------------------------
    def skybox_configure(self, faces: np.ndarray, equirectangular: bool = False):
        """Skybox::configure (skybox.rs:341): faces [6, n, n, 4] f32, +X -X +Y -Y +Z -Z."""
        f = np.ascontiguousarray(faces, dtype=np.float32)
        assert f.ndim == 4 and f.shape[0] == 6 and f.shape[1] == f.shape[2] and f.shape[3] == 4
        self._check(self._lib.gswt_skybox_configure(self._h, _ptr(f), f.shape[1], 1 if equirectangular else 0))

    def skybox_render(self, camera, width: int, height: int, out_device_ptr: int):
        """Skybox::render (skybox.rs:457) into a device RGBA f32 buffer."""
        cam = (C.c_char * 176).from_buffer_copy(bytes(camera))
        self._check(self._lib.gswt_skybox_render(self._h, cam, width, height, C.c_void_p(out_device_ptr)))

    def proxy_configure(self, mips, grid_dim: int = 2048):
        """Proxy::configure (proxy.rs:208): mip chain [level][n >> level, n >> level, 4] f32."""
        ms = [np.ascontiguousarray(m, dtype=np.float32) for m in mips]
        arr = (C.c_void_p * len(ms))(*[m.ctypes.data for m in ms])
        self._check(self._lib.gswt_proxy_configure(self._h, arr, ms[0].shape[0], len(ms), grid_dim))

    def proxy_render(self, uniforms, width: int, height: int, rgba_device_ptr: int, depth_device_ptr: int, clear_depth: bool):
        """One draw of Proxy::render (proxy.rs:366); uniforms: 224-byte proxy.wgsl Uniforms block."""
        u = (C.c_char * 224).from_buffer_copy(bytes(uniforms))
        self._check(self._lib.gswt_proxy_render(self._h, u, width, height, C.c_void_p(rgba_device_ptr), C.c_void_p(depth_device_ptr),
                                                1 if clear_depth else 0))

    def shard_rows_padded(self, height: int, shard_count: int) -> int:
        return int(self._lib.gswt_shard_rows_padded(height, shard_count))

    def shard_cols_padded(self, width: int, shard_count: int) -> int:
        return int(self._lib.gswt_shard_cols_padded(width, shard_count))

    def unshard_mode(self, gathered_device_ptr: int, width: int, height: int, shard_count: int, mode, out_device_ptr: int):
        m = L.GSWT_SHARD_COLUMNS if mode in ("cols", "columns", L.GSWT_SHARD_COLUMNS) and mode != 0 else L.GSWT_SHARD_ROWS
        self._check(self._lib.gswt_unshard_mode(self._h, C.c_void_p(gathered_device_ptr), width, height, shard_count, m,
                                                C.c_void_p(out_device_ptr)))

    def unshard(self, gathered_device_ptr: int, width: int, height: int, shard_count: int, out_device_ptr: int):
        self._check(self._lib.gswt_unshard(self._h, C.c_void_p(gathered_device_ptr), width, height, shard_count,
                                           C.c_void_p(out_device_ptr)))

    # -- multi-GPU gather behind the ABI -------------------------------------------------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0: the 128-byte RCCL id every rank passes to comm_init (ship it by any means)."""
        buf = (C.c_char * L.GSWT_COMM_ID_BYTES)()
        rc = L.load().gswt_comm_unique_id(buf)
        if rc != 0:
            raise GSWTError(rc, "gswt_comm_unique_id: RCCL is not available")
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        self._check(self._lib.gswt_comm_init(self._h, C.c_char_p(unique_id), rank, world))

    def comm_destroy(self):
        self._check(self._lib.gswt_comm_destroy(self._h))

    def render_gather(self, ticket: int, frame_device_ptr: int):
        """Overflow-safe fence + ncclAllGather of the shard images + re-assembly, on the ctx stream."""
        self._check(self._lib.gswt_render_gather(self._h, ticket, C.c_void_p(frame_device_ptr)))

    @staticmethod
    def group_init(renderers):
        """All ranks in one process (peer copies instead of RCCL): rank r = renderers[r]."""
        arr = (C.c_void_p * len(renderers))(*[r._h for r in renderers])
        rc = L.load().gswt_group_init(arr, len(renderers))
        if rc != 0:
            raise GSWTError(rc, "gswt_group_init failed")

    @staticmethod
    def group_render_gather(renderers, tickets, frame_device_ptrs):
        n = len(renderers)
        arr = (C.c_void_p * n)(*[r._h for r in renderers])
        tk = (C.c_int * n)(*tickets)
        fr = (C.c_void_p * n)(*frame_device_ptrs)
        rc = L.load().gswt_group_render_gather(arr, tk, fr, n)
        if rc != 0:
            raise GSWTError(rc, renderers[0]._lib.gswt_last_error(renderers[0]._h).decode())

    def synchronize(self):
        self._check(self._lib.gswt_synchronize(self._h))

    def timings(self) -> dict:
        t = L.Timings()
        self._check(self._lib.gswt_last_timings(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in L.Timings._fields_ if not k.startswith("_pad")}

    VARYINGS_DTYPE = np.dtype([("visible", "<i4"), ("ndc", "<f4", 2), ("depth", "<f4"), ("major", "<f4", 2),
                               ("minor", "<f4", 2), ("rgba", "<f4", 4)])

    def read_projected(self) -> np.ndarray:
        n = C.c_size_t(0)
        self._check(self._lib.gswt_debug_read_projected(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(1, n.value), dtype=self.VARYINGS_DTYPE)
        self._check(self._lib.gswt_debug_read_projected(self._h, _ptr(out), out.shape[0], C.byref(n)))
        return out[:n.value]

    def depth_stats(self):
        """GSWT_ORDER_DEPTH: (frames on the tile-local path, frames on the global depth passes, longest tile list of the last depth-ordered frame)."""
        a = (C.c_ulonglong * 3)()
        self._check(self._lib.gswt_debug_depth_stats(self._h, a))
        return int(a[0]), int(a[1]), int(a[2])

    def frame_times(self, ticket_ref: int, ticket: int):
        """(start, end, gather end) of slot `ticket`'s frame in ms after the start of slot `ticket_ref`'s frame (device timeline)."""
        out = (C.c_float * 3)()
        self._check(self._lib.gswt_debug_frame_times(self._h, ticket_ref, ticket, out))
        return float(out[0]), float(out[1]), float(out[2])

    def read_ranges(self) -> np.ndarray:
        """[n_tiles, 2] (start, end) of each screen tile's slice of the sorted pair list."""
        n = C.c_size_t(0)
        self._check(self._lib.gswt_debug_read_ranges(self._h, None, 0, C.byref(n)))
        out = np.zeros((max(1, n.value), 2), dtype=np.uint32)
        self._check(self._lib.gswt_debug_read_ranges(self._h, _ptr(out), out.shape[0], C.byref(n)))
        return out[:n.value]
