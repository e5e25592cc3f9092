"""Device-side worker stages (gswt_worker_* of include/gswt_hip.h): update_lod, selective merging, tile order, view choice and
SortData records as HIP kernels.  The tile map itself is built by libgswt_host (WangTile.build_tiles) and handed over per
build event; there is no CPU fallback -- without libgswt_hip / a GPU the constructor raises."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class DeviceWorker:
    def __init__(self, renderer, wang):
        """renderer: GSWTRenderer (the worker lives on its device); wang: a configured host.WangTile."""
        self._lib = L.load()
        self._r = renderer
        self._wang = wang
        cfg = wang.worker_config()
        h = C.c_void_p()
        rc = self._lib.gswt_worker_create(renderer._h, C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"gswt_worker_create failed ({rc})")
        self._h = h
        self.n_cells = cfg.map_w * cfg.map_h

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"gswt_worker error {rc}: {self._lib.gswt_worker_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.gswt_worker_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def build_tiles(self, cam_pos):
        """After WangTile.build_tiles on the host: upload the map, then update_lod on the device."""
        cells, n, cc = self._wang.export_cells()
        self._check(self._lib.gswt_worker_set_cells(self._h, cells, n, cc))
        self._check(self._lib.gswt_worker_update_lod(self._h, _f3(cam_pos)))

    def update_lod(self, cam_pos):
        self._check(self._lib.gswt_worker_update_lod(self._h, _f3(cam_pos)))

    def sort_tiles(self, cam_pos, view_proj):
        vp = np.ascontiguousarray(view_proj, dtype=np.float32)
        self._check(self._lib.gswt_worker_sort_tiles(self._h, _f3(cam_pos), vp.ctypes.data_as(C.c_void_p)))

    def cell_state(self) -> np.ndarray:
        st = (L.CellState * self.n_cells)()
        self._check(self._lib.gswt_worker_read_cell_state(self._h, st, self.n_cells))
        return np.frombuffer(bytes(st), dtype=np.uint32).reshape(self.n_cells, 5).copy()

    def read_sort(self):
        """-> (tiles bytes, groups bytes, members bytes, n_tiles, n_groups, n_members, n_merged)"""
        sd = L.SortDataC()
        self._check(self._lib.gswt_worker_read_sort(self._h, C.byref(sd)))
        tiles = C.string_at(C.cast(sd.tiles, C.c_void_p), sd.n_tiles * C.sizeof(L.SortedTile)) if sd.n_tiles else b""
        groups = C.string_at(sd.groups, sd.n_groups * C.sizeof(L.MergeGroup)) if sd.n_groups else b""
        members = C.string_at(sd.members, sd.n_members * C.sizeof(L.MergeMember)) if sd.n_members else b""
        return tiles, groups, members, int(sd.n_tiles), int(sd.n_groups), int(sd.n_members), int(sd.n_merged)

    def fetch(self):
        """Worker thread: wait for the sort event and publish its records, groups and draws in host memory."""
        self._check(self._lib.gswt_worker_fetch(self._h))

    def swap_in(self, fetch: bool = True):
        """gswt_set_draws_from_worker (render thread): the renderer's next frames use the sort event published last.
        fetch=True is the single-threaded form (fetch, then swap in)."""
        if fetch:
            self.fetch()
        self._r._check(self._lib.gswt_set_draws_from_worker(self._r._h, self._h))
