#!/usr/bin/env python3
"""Histogram of the screen tiles' pair-list lengths of one frame (what k_tile_depth_sort's size classes are cut for).
usage: tools/tile_lengths.py [workload ...]"""
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from gswt_renderer_amd.renderer import GSWTRenderer  # noqa: E402

for name in (sys.argv[1:] or ["c3"]):
    w, wang, cu, vp, sort = bench.build_workload(name)
    r = GSWTRenderer(0)
    wang.upload_to(r)
    r.configure(wang.height_map() if int(wang.user.surface_type) == 1 else None)
    r.set_draws(sort.draws, sort.merged_gs_index, sort.merged_map_id, sort.merged_lod_id)
    r.render(cu, wang.scene_uniforms(), w["width"], w["height"])
    rg = r.read_ranges()
    ln = (rg[:, 1] - rg[:, 0]).astype(np.int64)
    edges = [0, 1, 65, 129, 257, 513, 1025, 2049, 4097, 8193, 16385, 1 << 40]
    print(f"{name}: {ln.size} tiles, {int(ln.sum())} pairs, longest {int(ln.max())}")
    for a, b in zip(edges[:-1], edges[1:]):
        m = (ln >= a) & (ln < b)
        print(f"  {a:>6} .. {min(b - 1, int(ln.max())):>6}: {int(m.sum()):>6} tiles  {int(ln[m].sum()):>9} pairs ({100.0 * ln[m].sum() / max(1, ln.sum()):.1f} %)")
    r.close()
