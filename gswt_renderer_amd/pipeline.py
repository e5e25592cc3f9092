"""Frame harness: the order of calls State::render makes on the hot path (state.rs:237-437) --
worker `build_tiles` / `sort_tiles` when the camera moved, SortData swap-in, then
`GSWTRenderer::render`.  Only the hot-path calls are mirrored; windowing, GUI, channels and the
skybox / proxy passes are out of scope (their outputs enter as bg_rgba / bg_depth).
"""
from __future__ import annotations

import numpy as np

from . import host
from .renderer import GSWTRenderer


class GSWTPipeline:
    def __init__(self, verts_or_zip, user: host.UserData, device_id: int = 0, renderer: GSWTRenderer | None = None,
                 device_merge: bool = False):
        if isinstance(verts_or_zip, (str, bytes, bytearray)):
            ts = host.TileSet.from_zip(verts_or_zip)
        else:
            ts = host.TileSet.from_vertices(verts_or_zip)
        self.wang = host.WangTile(ts)                      # State::new: WangTile::new(scene_vec)
        self.renderer = renderer or GSWTRenderer(device_id)
        self.wang.upload_to(self.renderer)                 # GSWTRenderer::new(.., wang.preload())
        self.device_merge = device_merge
        if device_merge:                                   # merged-group lists are then built on the GPU per sort event
            self.wang.upload_raw_depth_to(self.renderer)
            self.wang.set_device_merge(True)
        self.configure(user)
        self.sort = None
        self._last_vp = None

    def configure(self, user: host.UserData, height_tex=None):
        self.conf = self.wang.configure(user, height_tex)  # worker configure, then renderer configure
        self.renderer.configure(self.wang.height_map() if user.surface_type == host.SURFACE_HEIGHTMAP else None)
        self.sort = None
        self._last_vp = None

    def update(self, cam_pos, view_proj, force_sort: bool = False) -> bool:
        """Worker half of a frame (state.rs:483-560).  Returns True when a new SortData was swapped in."""
        rebuilt = False
        if self.wang.check_update(cam_pos):
            self.wang.build_tiles(cam_pos)
            rebuilt = True
        vp = np.asarray(view_proj, dtype=np.float32)
        moved = self._last_vp is None or float(np.abs(vp - self._last_vp).sum()) >= 0.01   # state.rs:527-548
        if rebuilt or moved or force_sort or self.wang.user.always_sort or self.sort is None:
            self.sort = self.wang.sort_tiles(cam_pos, vp)
            self._last_vp = vp.copy()
            s = self.sort
            if self.device_merge:
                import ctypes as C
                from . import _lib as L
                g = (L.MergeGroup * max(1, len(s.groups)))(*s.groups)
                m = (L.MergeMember * max(1, len(s.members)))(*s.members)
                self.renderer.set_draws_merge_groups(s.draws, g, len(s.groups), m, len(s.members))
            else:
                self.renderer.set_draws(s.draws, s.merged_gs_index, s.merged_map_id, s.merged_lod_id)
            return True
        return False

    def render(self, camera_uniforms, width, height, **kw):
        su = self.wang.scene_uniforms(splat_scale=kw.pop("splat_scale", 1.0), scene_scale=kw.pop("scene_scale", (1.0, 1.0, 1.0)),
                                      height_map_scale_v=kw.pop("height_map_scale_v", 1.0))
        # the other RenderConfig fields of SceneUniforms::from_data (renderer.rs:631-672)
        su.draw_mode = int(kw.pop("draw_mode", 0))
        su.point_cloud_radius = float(kw.pop("point_cloud_radius", 0.0))
        su.use_clip = int(kw.pop("use_clip", 0))
        su.clip_height = float(kw.pop("clip_height", 0.0))
        return self.renderer.render(camera_uniforms, su, width, height, **kw)
