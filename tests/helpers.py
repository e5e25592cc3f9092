"""Shared builders for the parity tests: the same synthetic inputs go to the CPU oracle
(oracle/) and to the HIP path (gswt_renderer_amd over the C ABI)."""
from __future__ import annotations

import functools

import numpy as np

from gswt_renderer_amd import synth
from gswt_renderer_amd import _lib as L
from gswt_renderer_amd.renderer import make_draw
from oracle import gswt_oracle as orc


@functools.lru_cache(maxsize=8)
def tileset(n_lod=2, n_tile=16, lod0_count=1500, base_scale=0.02, seed_offset=0):
    verts = synth.make_tileset(n_lod=n_lod, n_tile=n_tile, lod0_count=lod0_count, base_scale=base_scale,
                               seed_offset=seed_offset)
    rows = [[orc.scene_load(v) for v in lod] for lod in verts]
    return orc.preprocess(rows)


def to_product_tile(t: orc.Tile80) -> L.TileUniforms:
    return L.TileUniforms.from_buffer_copy(bytes(t))


class Case:
    """A draw list in both representations."""

    def __init__(self, pp):
        self.pp = pp
        self.orc_draws = []
        self.draws = []
        self.m_gs, self.m_map, self.m_lod = [], [], []
        self.m_off = 0

    def add_static(self, *, lod, tile, view, offset, valid_lod_id=-1, changing=0, changing_to_lower=-1,
                   base_lod=None, corners=None, map_index=0, map_coord=(0, 0)):
        base_lod = lod if base_lod is None else base_lod
        tu = orc.tile_uniforms(single_draw=0, map_index=map_index, valid_lod_id=valid_lod_id, changing=changing,
                               changing_to_lower=changing_to_lower, tile_id=(lod, tile, view), offset=offset,
                               map_coord=map_coord)
        pp = self.pp
        self.orc_draws.append(orc.Draw(tu, pp.gs_index[base_lod][tile][view], None, pp.gs_lod_id[base_lod][tile][view]))
        self.draws.append(make_draw(to_product_tile(tu), base=(base_lod, tile, view), corners=corners, lod=lod))

    def add_merged(self, *, members, view, head_lod, head_tile, head_map_index, offset=(0, 0, 0)):
        val = orc.build_merged_value(self.pp, members, view, head_lod)
        single_lod = val["single_lod_id"]
        tu = orc.tile_uniforms(single_draw=1, map_index=head_map_index, single_lod_id=single_lod, valid_lod_id=-1,
                               changing=1 if single_lod == -1 else 0, changing_to_lower=-1,
                               tile_id=(head_lod, head_tile, view), offset=offset)
        self.orc_draws.append(orc.Draw(tu, val["gs_index"], val["gs_map_id"], val["gs_lod_id"]))
        n = val["splat_count"]
        self.draws.append(make_draw(to_product_tile(tu), merged_range=(self.m_off, n), merged_has_lod=single_lod == -1,
                                    lod=head_lod))
        self.m_gs.append(val["gs_index"])
        self.m_map.append(val["gs_map_id"])
        self.m_lod.append(val["gs_lod_id"] if val["gs_lod_id"] is not None else np.zeros(n, dtype=np.uint32))
        self.m_off += n

    def merged_arrays(self):
        if not self.m_gs:
            return None, None, None
        return np.concatenate(self.m_gs), np.concatenate(self.m_map), np.concatenate(self.m_lod)

    def upload(self, renderer):
        pp = self.pp
        renderer.upload_scene(pp.tex, pp.gs_index, pp.gs_lod_id)
        g, m, l = self.merged_arrays()
        renderer.set_draws(self.draws, g, m, l)


def grid_case(pp, *, half=(1, 2), center=(0, 0), tile_width=4.0, lod_of=lambda ix, iy: 0, view=2, mode="plain"):
    """A (2hx+1) x (2hy+1) grid of plain class-A draws, far rows first (back-to-front for +y view)."""
    case = Case(pp)
    hx, hy = half
    wy = 2 * hy + 1
    for my in range(2 * hy, -1, -1):
        for mx in range(2 * hx + 1):
            ix, iy = mx - hx + center[0], my - hy + center[1]
            tile = (mx * 7 + my * 3) % pp.n_tile
            lod = lod_of(ix, iy)
            case.add_static(lod=lod, tile=tile, view=view, offset=(ix * tile_width, iy * tile_width, 0.0),
                            valid_lod_id=lod, map_index=mx * wy + my, map_coord=(mx, my))
    return case


def max_abs_diff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))
