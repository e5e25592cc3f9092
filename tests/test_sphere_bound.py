"""The Sphere surface's band-cull bound (sphere_cell_box, gswt_kernels.hip): numerical check, on the CPU, of the two facts it rests on.
The strip parametrisation is restated here in float64 from gswt.wgsl:515-564 (sphere_get_uv + sphere_uv_to_pos); it is test-side only."""
import numpy as np
import pytest

K_KERNEL = 2.5          # the constant sphere_cell_box uses (analytic bound 2.363, see its header)


def sphere_point(bw, bidx, bidy, bx, by):
    bx = np.asarray(bx, dtype=np.float64); by = np.asarray(by, dtype=np.float64)
    t = bx - by
    with np.errstate(all="ignore"):
        if bidy == 0:
            u1 = np.where(t == bw, 0.0, (by / (bw - t) + bidx) / 5); v1 = ((bw - t) / bw) / 3
            u2 = (bx / bw + bidx) / 5 + ((by - bx) / bw) * 0.1; v2 = ((by - bx) / bw) / 3 + 1 / 3
        else:
            u1 = (bx / bw + bidx) / 5 + ((bw - t) / bw) * 0.1; v1 = ((bw - t) / bw) / 3 + 1 / 3
            u2 = np.where(-t == bw, 0.0, (bx / (bw + t) + bidx) / 5 + 0.1); v2 = (-t / bw) / 3 + 2 / 3
    u = np.where(by < bx, u1, u2); v = np.where(by < bx, v1, v2)
    u = (u + 0.5 * np.floor(v)) * 2 * np.pi; v = (v - 0.5) * np.pi
    return np.stack([np.cos(v) * np.cos(u), np.cos(v) * np.sin(u), np.sin(v)], -1)


@pytest.mark.parametrize("bidy", [0, 1])
def test_strip_map_is_lipschitz_inside_a_block(bidy):
    """|lz(a) - lz(b)| <= K (|dbx| + |dby|) / block_w for a, b inside one block -- across the diagonal and next to the pole too."""
    rng = np.random.default_rng(7 + bidy)
    bw = 8.0
    worst = 0.0
    for bidx in range(5):
        for scale in (1e-3, 1e-2, 0.1, 1.0, 8.0):
            a = rng.uniform(0, bw, size=(60000, 2))
            b = np.clip(a + rng.normal(size=a.shape) * scale, 0, bw)
            corner = np.array([bw, 0.0]) if bidy == 0 else np.array([0.0, bw])           # the pole of this block row
            a2 = np.clip(corner + rng.normal(size=a.shape) * 0.05 * bw, 0, bw)
            b2 = np.clip(a2 + rng.normal(size=a.shape) * scale * 0.01, 0, bw)
            for p, q in ((a, b), (a2, b2)):
                l1 = np.abs(p - q).sum(1)
                ok = l1 > 1e-9
                d = np.linalg.norm(sphere_point(bw, bidx, bidy, p[:, 0], p[:, 1]) - sphere_point(bw, bidx, bidy, q[:, 0], q[:, 1]), axis=1)
                worst = max(worst, float((d[ok] / l1[ok] * bw).max()))
    assert 1.0 < worst < 2.363 + 1e-6 < K_KERNEL, worst


def test_cell_box_contains_every_centre():
    """The box sphere_cell_box builds (n x n samples of lz R, +- rho) holds lz (R + z) for every (bx, by, z) of the footprint."""
    rng = np.random.default_rng(11)
    bw, R, n = 8.0, 6.5, 3
    for _ in range(300):
        bidx, bidy = int(rng.integers(0, 5)), int(rng.integers(0, 2))
        w, h = rng.uniform(0.2, bw, size=2)
        bx0, by0 = rng.uniform(0, bw - w), rng.uniform(0, bw - h)
        zmax = rng.uniform(0, 1.0)
        rho = (R * K_KERNEL * (w + h) / (2 * n * bw) + zmax) * 1.02 + 1e-4 * R
        gx, gy = np.meshgrid(bx0 + w * (np.arange(n) + 0.5) / n, by0 + h * (np.arange(n) + 0.5) / n)
        pts = sphere_point(bw, bidx, bidy, gx.ravel(), gy.ravel()) * R
        blo, bhi = pts.min(0) - rho, pts.max(0) + rho
        q = np.stack([rng.uniform(bx0, bx0 + w, 4000), rng.uniform(by0, by0 + h, 4000)], 1)
        z = rng.uniform(-zmax, zmax, 4000)
        c = sphere_point(bw, bidx, bidy, q[:, 0], q[:, 1]) * (R + z)[:, None]
        assert (c >= blo).all() and (c <= bhi).all()
