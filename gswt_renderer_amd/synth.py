"""Seeded synthetic GSWT tile sets (no dataset is available offline).

Follows the measurement spec in SURVEY.md section 8(d) / BASELINE.md section 3:
``n_lod`` LODs x ``n_tile`` Wang tiles, LOD l has ``lod0_count / 4**l`` splats per tile,
positions x,y ~ U[0, tile_width), z ~ N(0, 0.15) clipped to +-0.6, log-scales
~ N(ln 0.02 + l ln 2, 0.35) per axis (so the average scale strictly increases with
LOD, which wangtile.rs:138-140 asserts), quaternions uniform on S^3, opacity logit
~ N(1.0, 1.5), f_dc ~ N(0, 1); numpy PCG64(seed = 1000*lod + tile).

The output is the reference's own input format: per (lod, tile) an ``[n, 62]``
float32 array of 3DGS PLY vertex records (scene.rs:19-26: pos3, normal3, f_dc3 +
f_rest45, opacity, scale3, rot4), which ``write_ply`` / ``write_tile_zip`` serialise
to the GSWT tile-zip layout ``lod{L}_tile_{T}.ply`` that load_scene_zip
(scene.rs:1030-1141) reads.
"""
from __future__ import annotations

import io
import zipfile

import numpy as np

PLY_PROPS = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)]
             + [f"f_rest_{i}" for i in range(45)] + ["opacity"] + [f"scale_{i}" for i in range(3)]
             + [f"rot_{i}" for i in range(4)])
assert len(PLY_PROPS) == 62


def make_tile(lod: int, tile: int, count: int, tile_width: float = 4.0, base_scale: float = 0.02,
              seed_offset: int = 0) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(1000 * lod + tile + seed_offset))
    v = np.zeros((count, 62), dtype=np.float32)
    v[:, 0:2] = rng.uniform(0.0, tile_width, size=(count, 2))
    v[:, 2] = np.clip(rng.normal(0.0, 0.15, size=count), -0.6, 0.6)
    v[:, 6:9] = rng.normal(0.0, 1.0, size=(count, 3))
    v[:, 54] = rng.normal(1.0, 1.5, size=count)
    v[:, 55:58] = rng.normal(np.log(base_scale) + lod * np.log(2.0), 0.35, size=(count, 3))
    q = rng.normal(0.0, 1.0, size=(count, 4))
    v[:, 58:62] = q / np.linalg.norm(q, axis=1, keepdims=True)
    return v


def make_tileset(n_lod: int = 3, n_tile: int = 16, lod0_count: int = 9800, tile_width: float = 4.0,
                 base_scale: float = 0.02, seed_offset: int = 0):
    """-> verts[lod][tile] = [n, 62] float32."""
    return [[make_tile(l, t, max(1, lod0_count // (4 ** l)), tile_width, base_scale, seed_offset)
             for t in range(n_tile)] for l in range(n_lod)]


def write_ply(verts62: np.ndarray) -> bytes:
    """Binary little-endian 3DGS PLY (header <= 65 lines, ends exactly 'end_header\\n')."""
    v = np.ascontiguousarray(verts62, dtype="<f4").reshape(-1, 62)
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {v.shape[0]}"]
    head += [f"property float {p}" for p in PLY_PROPS]
    head += ["end_header"]
    return ("\n".join(head) + "\n").encode("ascii") + v.tobytes()


def write_tile_zip(path_or_file, verts) -> None:
    """GSWT tile-zip: entries lod{L}_tile_{T}.ply (scene.rs:1057)."""
    with zipfile.ZipFile(path_or_file, "w", compression=zipfile.ZIP_DEFLATED) as zf:
        for l, lod in enumerate(verts):
            for t, v in enumerate(lod):
                zf.writestr(f"tiles/lod{l}_tile_{t}.ply", write_ply(v))


def tile_zip_bytes(verts) -> bytes:
    bio = io.BytesIO()
    write_tile_zip(bio, verts)
    return bio.getvalue()
