#!/usr/bin/env python3
"""Generates tests/golden/case_*.npz from the CPU oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference cannot be run here and ships no fixtures
(SURVEY.md 8c), so these vectors pin the ORACLE's behaviour at the time they were made: inputs
(packed 32 B/splat rows per (lod, tile), tile-id map, camera, config) and expected outputs
(tile draw order, presort views, draw classes, float image, per-splat vertex-stage outputs)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from gswt_renderer_amd import synth  # noqa: E402
from oracle import gswt_oracle as orc  # noqa: E402
from oracle import wangtile_oracle as wo  # noqa: E402

CASES = {
    "case_plane": dict(cfg=dict(tile_map_half_wh=(2, 2), surface_type=0, lod_max_dist=14.0, tile_sort_type=3, merge_type=2),
                       pos=(1.3, 0.4, 2.5), tgt=(2.0, 4.0, 1.5), W=64, H=48, lod0=60, n_lod=2),
    "case_hmap": dict(cfg=dict(tile_map_half_wh=(2, 2), surface_type=1, lod_max_dist=14.0, tile_sort_type=3, merge_type=2,
                               height_map_type=4, height_map_wh=(6, 6), height_map_scale=(1.0, 1.0, 0.4)),
                      pos=(0.7, -0.6, 3.0), tgt=(1.5, 3.0, 1.0), W=64, H=64, lod0=60, n_lod=2),
    "case_sphere": dict(cfg=dict(tile_map_half_wh=(5, 2), surface_type=2, sphere_radius=6.5, lod_max_dist=60.0, tile_sort_type=3,
                                 merge_type=2),
                        pos=(3.0, -19.0, 6.0), tgt=(0.0, 0.0, 0.0), W=64, H=48, lod0=60, n_lod=2),
    "case_plane_mode1": dict(cfg=dict(tile_map_half_wh=(2, 2), surface_type=0, lod_max_dist=14.0, tile_sort_type=3, merge_type=2),
                             pos=(1.3, 0.4, 2.5), tgt=(2.0, 4.0, 1.5), W=64, H=48, lod0=60, n_lod=2, render_config=dict(draw_mode=1)),
}


def make_passes(out_dir):
    """Small skybox + proxy case (inputs and the oracle's outputs): passes_small.npz"""
    rng = np.random.default_rng(42)
    W, H = 48, 32
    cam = orc.Camera(W, H, (0.5, 0.3, 5.0), (1.0, 6.0, 2.5), [0, 0, 1])
    faces = rng.uniform(0, 1, (6, 8, 8, 4)).astype(np.float32)
    faces[..., 3] = 1.0
    mips, cur = [], rng.uniform(0, 1, (16, 16, 4)).astype(np.float32)
    while True:
        mips.append(cur.copy())
        if cur.shape[0] == 1:
            break
        cur = cur.reshape(cur.shape[0] // 2, 2, cur.shape[1] // 2, 2, 4).mean((1, 3)).astype(np.float32)
    hm = rng.uniform(-1, 1, (4, 4)).astype(np.float32)
    sky = orc.skybox_render(cam, faces, W, H)
    rgba, depth = sky.copy(), np.ones((H, W), np.float32)
    common = dict(surface_type=1, map_half_wh=(3, 4), center_coord=(1, -1), height_map_scale=(1.0, 1.0, 0.6))
    us = [orc.proxy_uniforms(cam, map_proxy=0, height_offset=-0.5, width_scale=4.0, **common),
          orc.proxy_uniforms(cam, map_proxy=1, height_offset=-0.45, **common)]
    for u in us:
        orc.proxy_render(u, W, H, rgba, depth, mips, height_map=hm, grid_dim=24)
    np.savez_compressed(os.path.join(out_dir, "passes_small.npz"), faces=faces, mips=np.concatenate([m.reshape(-1) for m in mips]),
                        tex_size=np.int32(16), hm=hm, camera=np.array([cam.position, cam.target], dtype=np.float32),
                        size=np.array([W, H], dtype=np.int32), grid_dim=np.int32(24),
                        uniforms=np.stack([np.frombuffer(bytes(u), dtype=np.uint8) for u in us]), sky=sky, rgba=rgba, depth=depth)
    print("passes_small", "proxy coverage", float((depth < 1).mean()))


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    only = set(sys.argv[1:])
    if not only or "passes_small" in only:
        make_passes(out_dir)
    for name, c in CASES.items():
        if only and name not in only:
            continue
        verts = synth.make_tileset(n_lod=c["n_lod"], n_tile=16, lod0_count=c["lod0"])
        rows = [[orc.scene_load(v) for v in lod] for lod in verts]
        pp = orc.preprocess(rows)
        ow = wo.WangTile(pp)
        ou = ow.configure(wo.UserData(**c["cfg"]))
        cam = orc.Camera(c["W"], c["H"], c["pos"], c["tgt"], [0, 0, 1])
        with np.errstate(all="ignore"):
            osd = ow.build_tiles(c["pos"])
            ids = np.array([ow.tile_map[i][j].tid[1] for i in range(ou.tile_map_wh[0]) for j in range(ou.tile_map_wh[1])], dtype=np.uint32)
            osort = ow.sort_tiles(c["pos"], cam.view_proj())
            draws = wo.renderer_draws(pp, osort, cam.view_proj())
        su = wo.scene_uniforms_from_data(ou, osd["center_coord"], **c.get("render_config", {}))
        hm = ou.height_map.reshape(ou.height_map_wh[1], ou.height_map_wh[0]) if ou.surface_type == 1 else None
        with orc.v2():          # the rounding sequence v2 (GSWT_OPT_STRICT_VS = 0): `image`, `varyings`, `stats` as in rounds 2 and 3
            img, st = orc.render(cam.uniforms(), su, pp.tex, draws, c["W"], c["H"], height_map=hm)
            var = orc.project_draws(cam.uniforms(), su, pp.tex, draws, height_map=hm)
        # the default since round 4: strict vertex stage (= `varyings_strict`) + the fragment sequence F1..F4
        img_d, st_d = orc.render(cam.uniforms(), su, pp.tex, draws, c["W"], c["H"], height_map=hm)
        with orc.strict():      # the shader text operator by operator (oracle/gswt_oracle.c, "STRICT mode")
            img_s, st_s = orc.render(cam.uniforms(), su, pp.tex, draws, c["W"], c["H"], height_map=hm)
            var_s = orc.project_draws(cam.uniforms(), su, pp.tex, draws, height_map=hm)
        classes = [2 if d.tile.single_draw else (1 if d.tile.changing else 0) for d in draws]
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"),
            rows=np.concatenate([rows[l][t] for l in range(c["n_lod"]) for t in range(16)]),
            row_counts=np.array([[rows[l][t].shape[0] for t in range(16)] for l in range(c["n_lod"])], dtype=np.int32),
            tile_ids=ids, config=json.dumps(c["cfg"]), render_config=json.dumps(c.get("render_config", {})), camera=np.array([c["pos"], c["tgt"]], dtype=np.float32),
            size=np.array([c["W"], c["H"]], dtype=np.int32),
            order=np.array([t.map_index for t in osort["tile_instance_vec"]], dtype=np.int32),
            views=np.array([t.view_id for t in osort["tile_instance_vec"]], dtype=np.int32),
            lods=np.array([t.tid[0] for t in osort["tile_instance_vec"]], dtype=np.int32),
            draw_classes=np.array(classes, dtype=np.int32),
            image=img, varyings=var, image_strict=img_s, varyings_strict=var_s, stats=np.array([st["n_instanced"], st["n_visible"], st["n_pairs16"]], dtype=np.int64),
            image_default=img_d, stats_default=np.array([st_d["n_instanced"], st_d["n_visible"], st_d["n_pairs16"]], dtype=np.int64))
        print(name, "strict vs v2 image", float(np.abs(img - img_s).max()))
        print(name, "draws", len(draws), "classes", np.bincount(classes, minlength=3).tolist(), st, "img max", float(img.max()))


if __name__ == "__main__":
    main()
