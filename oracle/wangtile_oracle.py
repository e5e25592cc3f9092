"""CPU ORACLE -- restatement of the reference's worker stages (src/wangtile.rs) and of the
host half of GSWTRenderer::render (src/renderer.rs:466-591).  TEST INFRASTRUCTURE ONLY
(same rules as gswt_oracle.py: only tests/, smoke() and bench.py's cpu_baseline may import it).

Pure Python + numpy float32 scalars; every float expression keeps the reference's operand
order, one rounding per operator.  Meant for small maps (pure-Python loops).

PARITY STATUS: "parity unpinned".  Two third-party behaviours are restated from their
published algorithms without the crate sources at hand and cannot be pinned here:
  * rand 0.9.2 StdRng (ChaCha12) + rand_core seed_from_u64 (PCG32 expansion) + random_range
    (wangtile.rs:55,353,385,1746-1752)  -> class StdRng below;
  * petgraph 0.8.3 toposort / DiGraph neighbor order / remove_node (wangtile.rs:1119-1213)
    -> class DiGraph below.
Tests therefore check the *constraints* (Wang edge colours match, the order respects the edge
DAG) and product-vs-oracle equality, and treat tile-id maps as explicit inputs where needed.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import gswt_oracle as orc

f32 = np.float32
_M32 = 0xFFFFFFFF
_M64 = 0xFFFFFFFFFFFFFFFF


# --------------------------------------------------------------------------
# rand 0.9 StdRng restatement (ChaCha12, 64-bit counter, stream 0)
# --------------------------------------------------------------------------
def _rotl32(x, n):
    return ((x << n) | (x >> (32 - n))) & _M32


def _chacha_block(key_words, counter, rounds=12):
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + \
         [counter & _M32, (counter >> 32) & _M32, 0, 0]
    w = list(st)

    def qr(a, b, c, d):
        w[a] = (w[a] + w[b]) & _M32; w[d] = _rotl32(w[d] ^ w[a], 16)
        w[c] = (w[c] + w[d]) & _M32; w[b] = _rotl32(w[b] ^ w[c], 12)
        w[a] = (w[a] + w[b]) & _M32; w[d] = _rotl32(w[d] ^ w[a], 8)
        w[c] = (w[c] + w[d]) & _M32; w[b] = _rotl32(w[b] ^ w[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(w[i] + st[i]) & _M32 for i in range(16)]


class StdRng:
    """StdRng::seed_from_u64 + next_u32 + random_range (unpinned restatement)."""

    def __init__(self, seed_u64: int = 0):
        # rand_core SeedableRng::seed_from_u64: PCG32 stream fills the 32-byte seed
        state = seed_u64 & _M64
        words = []
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & _M64
            xorshifted = (((state >> 18) ^ state) >> 27) & _M32
            rot = (state >> 59) & 31
            words.append(((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & _M32)
        self.key = words
        self.counter = 0
        self.buf = []

    def next_u32(self) -> int:
        if not self.buf:
            self.buf = _chacha_block(self.key, self.counter)
            self.counter += 1
        return self.buf.pop(0)

    def random_range_u32(self, n: int) -> int:
        """random_range(0..n) for a 32-bit usize (wasm32): widening multiply with one
        bias-correction draw (rand 0.9 UniformInt::sample_single_inclusive)."""
        assert n > 0
        m = self.next_u32() * n
        hi, lo = m >> 32, m & _M32
        if lo > ((-n) & _M32):
            hi2 = (self.next_u32() * n) >> 32
            if lo + hi2 > _M32:
                hi += 1
        return hi

    def random_range_f32_inclusive(self, low: float, high: float) -> np.float32:
        """random_range(low..=high) for f32 (rand 0.9 UniformFloat::sample_single_inclusive)."""
        low, high = f32(low), f32(high)
        max_rand = f32(1.0) - f32(2.0 ** -23)
        scale = f32(f32(high - low) / max_rand)
        while f32(f32(scale * max_rand) + low) > high:
            scale = np.nextafter(scale, f32(-np.inf), dtype=f32)
        bits = (self.next_u32() >> 9) | 0x3F800000
        v12 = np.array([bits], dtype=np.uint32).view(np.float32)[0]
        v01 = f32(v12 - f32(1.0))
        return f32(f32(v01 * scale) + low)


# --------------------------------------------------------------------------
# petgraph DiGraph restatement (adjacency as intrusive linked lists, newest edge first)
# --------------------------------------------------------------------------
class DiGraph:
    def __init__(self):
        self.weights = []        # node weight (map index)
        self.out = []            # per node: list of edge ids, newest first
        self.inc = []
        self.edges = []          # (src, dst) or None when removed

    def add_node(self, w):
        self.weights.append(w); self.out.append([]); self.inc.append([])
        return len(self.weights) - 1

    def add_edge(self, a, b):
        self.edges.append((a, b))
        e = len(self.edges) - 1
        self.out[a].insert(0, e)
        self.inc[b].insert(0, e)
        return e

    def neighbors_out(self, n):
        return [self.edges[e][1] for e in self.out[n]]

    def neighbors_in(self, n):
        return [self.edges[e][0] for e in self.inc[n]]

    def remove_node(self, n):
        """petgraph Graph::remove_node: drop incident edges, then swap_remove the node (the last
        node takes index n)."""
        for e in list(self.out[n]) + list(self.inc[n]):
            if self.edges[e] is None:
                continue
            a, b = self.edges[e]
            self.out[a].remove(e)
            self.inc[b].remove(e)
            self.edges[e] = None
        last = len(self.weights) - 1
        if n != last:
            self.weights[n] = self.weights[last]
            self.out[n] = self.out[last]
            self.inc[n] = self.inc[last]
            for e in self.out[n]:
                self.edges[e] = (n, self.edges[e][1])
            for e in self.inc[n]:
                self.edges[e] = (self.edges[e][0], n)
        self.weights.pop(); self.out.pop(); self.inc.pop()

    def toposort(self):
        """petgraph::algo::toposort (DFS finish order, reverse post-order, then a reversed-graph
        walk to detect cycles).  Returns (order, None) or (None, cycle_node)."""
        n = len(self.weights)
        discovered = [False] * n
        finished = [False] * n
        finish_stack = []
        for i in reversed(range(n)):
            if discovered[i]:
                continue
            stack = [i]
            while stack:
                nx = stack[-1]
                if not discovered[nx]:
                    discovered[nx] = True
                    for succ in self.neighbors_out(nx):
                        if succ == nx:
                            return None, nx
                        if not discovered[succ]:
                            stack.append(succ)
                else:
                    stack.pop()
                    if not finished[nx]:
                        finished[nx] = True
                        finish_stack.append(nx)
        finish_stack.reverse()
        discovered = [False] * n
        for i in finish_stack:
            # dfs.move_to(i); walk the reversed graph; any second node reached is on a cycle
            stack = [i]
            cycle = False
            while stack:
                node = stack.pop()
                if discovered[node]:
                    continue
                discovered[node] = True
                for pred in self.neighbors_in(node):
                    if not discovered[pred]:
                        stack.append(pred)
                if cycle:
                    return None, node
                cycle = True
        return finish_stack, None


# --------------------------------------------------------------------------
# small f32 vector helpers (cgmath operand order)
# --------------------------------------------------------------------------
def v3(x, y, z):
    return np.array([x, y, z], dtype=f32)


def dot3(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def mag3(a):
    return f32(np.sqrt(dot3(a, a)))


def dist3(a, b):
    return mag3((b - a).astype(f32))


def dist2_3(a, b):
    d = (b - a).astype(f32)
    return dot3(d, d)


def normalize3(a):
    return (a * f32(f32(1.0) / mag3(a))).astype(f32)


def cross3(a, b):
    return np.array([f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]),
                     f32(a[0] * b[1]) - f32(a[1] * b[0])], dtype=f32)


def mat3_cols(c0, c1, c2):
    return np.concatenate([c0, c1, c2]).astype(f32)       # column-major flat [3*c + r]


def mat3_identity():
    return np.array([1, 0, 0, 0, 1, 0, 0, 0, 1], dtype=f32)


def mat3_vec(m, v):
    return (m[0:3] * v[0] + m[3:6] * v[1] + m[6:9] * v[2]).astype(f32)


def mat3_invert(m):
    """cgmath Matrix3::invert."""
    c0, c1, c2 = m[0:3], m[3:6], m[6:9]
    det = f32(f32(f32(c0[0] * f32(f32(c1[1] * c2[2]) - f32(c2[1] * c1[2])))
                  - f32(c1[0] * f32(f32(c0[1] * c2[2]) - f32(c2[1] * c0[2]))))
              + f32(c2[0] * f32(f32(c0[1] * c1[2]) - f32(c1[1] * c0[2]))))
    r0 = (cross3(c1, c2) / det).astype(f32)
    r1 = (cross3(c2, c0) / det).astype(f32)
    r2 = (cross3(c0, c1) / det).astype(f32)
    # from_cols(r0, r1, r2).transpose()
    return np.array([r0[0], r1[0], r2[0], r0[1], r1[1], r2[1], r0[2], r1[2], r2[2]], dtype=f32)


def quat_from_mat3(m):
    """cgmath From<Matrix3> for Quaternion -> (s, x, y, z)."""
    def M(c, r):
        return m[3 * c + r]
    trace = f32(f32(M(0, 0) + M(1, 1)) + M(2, 2))
    half = f32(0.5)
    if trace >= 0:
        s = f32(np.sqrt(f32(f32(1.0) + trace)))
        w = f32(half * s)
        s = f32(half / s)
        return np.array([w, f32(f32(M(1, 2) - M(2, 1)) * s), f32(f32(M(2, 0) - M(0, 2)) * s),
                         f32(f32(M(0, 1) - M(1, 0)) * s)], dtype=f32)
    if M(0, 0) > M(1, 1) and M(0, 0) > M(2, 2):
        s = f32(np.sqrt(f32(f32(f32(M(0, 0) - M(1, 1)) - M(2, 2)) + f32(1.0))))
        x = f32(half * s)
        s = f32(half / s)
        return np.array([f32(f32(M(1, 2) - M(2, 1)) * s), x, f32(f32(M(1, 0) + M(0, 1)) * s),
                         f32(f32(M(0, 2) + M(2, 0)) * s)], dtype=f32)
    if M(1, 1) > M(2, 2):
        s = f32(np.sqrt(f32(f32(f32(M(1, 1) - M(0, 0)) - M(2, 2)) + f32(1.0))))
        y = f32(half * s)
        s = f32(half / s)
        return np.array([f32(f32(M(2, 0) - M(0, 2)) * s), f32(f32(M(1, 0) + M(0, 1)) * s), y,
                         f32(f32(M(2, 1) + M(1, 2)) * s)], dtype=f32)
    s = f32(np.sqrt(f32(f32(f32(M(2, 2) - M(0, 0)) - M(1, 1)) + f32(1.0))))
    z = f32(half * s)
    s = f32(half / s)
    return np.array([f32(f32(M(0, 1) - M(1, 0)) * s), f32(f32(M(0, 2) + M(2, 0)) * s),
                     f32(f32(M(2, 1) + M(1, 2)) * s), z], dtype=f32)


def mat3_from_quat(q):
    """cgmath From<Quaternion> for Matrix3; q = (s, x, y, z)."""
    s, x, y, z = q
    x2, y2, z2 = f32(x + x), f32(y + y), f32(z + z)
    xx2, xy2, xz2 = f32(x2 * x), f32(x2 * y), f32(x2 * z)
    yy2, yz2, zz2 = f32(y2 * y), f32(y2 * z), f32(z2 * z)
    sy2, sz2, sx2 = f32(y2 * s), f32(z2 * s), f32(x2 * s)
    one = f32(1.0)
    return np.array([f32(f32(one - yy2) - zz2), f32(xy2 + sz2), f32(xz2 - sy2),
                     f32(xy2 - sz2), f32(f32(one - xx2) - zz2), f32(yz2 + sx2),
                     f32(xz2 + sy2), f32(yz2 - sx2), f32(f32(one - xx2) - yy2)], dtype=f32)


# --------------------------------------------------------------------------
# Data contracts (structure.rs)
# --------------------------------------------------------------------------
SORT_DISTANCE, SORT_VIEWPORT, SORT_OBJECT, SORT_GRAPH = 0, 1, 2, 3        # structure.rs:451-457
MERGE_NONE, MERGE_AXIS, MERGE_EDGE = 0, 1, 2                              # structure.rs:459-464
SURFACE_NONE, SURFACE_HEIGHTMAP, SURFACE_SPHERE = 0, 1, 2                 # structure.rs:435-440
HMAP_TEXTURE, HMAP_RANDOM, HMAP_SLOPEX, HMAP_SLOPEY, HMAP_DUALSLOPE = 0, 1, 2, 3, 4   # :442-449


@dataclass
class UserData:
    """structure.rs:15-65; defaults = UserData::new (:67-100) except the fields the GUI always
    overwrites, which take the GUI defaults of UserDataString::new (:121-137)."""
    tile_map_half_wh: tuple = (48, 48)
    center_option: int = 1
    update_distance2: float = 1.0
    tile_width: float = 4.0
    tile_sort_type: int = SORT_GRAPH
    surface_type: int = SURFACE_HEIGHTMAP
    height_map_wh: tuple = (10, 10)
    height_map_type: int = HMAP_RANDOM
    height_map_scale: tuple = (1.0, 1.0, 1.0)
    height_tex: Optional[tuple] = None
    sphere_radius: float = 20.0
    lod_max_dist: float = 96.0 * 4.0
    lod_blending: bool = True
    lod_transition_width_ratio: float = 0.05
    lod_bbox_check: bool = True
    lod_dist_tolerance: float = 0.0
    merge_type: int = MERGE_EDGE
    merge_tile_dist: tuple = (3, 10)
    merge_dot_threshold: float = 0.2
    merge_topk: int = 100
    use_cache: bool = True
    cache_size: int = 1024
    reset_rng: bool = True
    always_sort: bool = False
    # from the worker
    tile_map_wh: tuple = (0, 0)
    height_map: Optional[np.ndarray] = None
    lod_transition_dist: list = field(default_factory=list)
    n_tiles: tuple = (0, 0, 0)


@dataclass
class TileInstance:
    """structure.rs:495-509"""
    tid: tuple
    view_id: int
    tile_offset: np.ndarray
    map_index: int
    map_coord: tuple
    tile_center: np.ndarray
    merge_status: tuple            # ('none',) | ('from', [idx...]) | ('to', idx)
    transition_status: tuple       # ('none',) | ('spawning', f) | ('changing', to_lower)
    to_local: np.ndarray           # mat3 column-major
    corner_data: Optional[list]    # 4 x (pos vec3, to_world mat3)  SW, NW, NE, SE
    edge_data: Optional[list]      # 4 x (pos vec3, normal vec3)    W, N, E, S

    def clone(self):
        return TileInstance(self.tid, self.view_id, self.tile_offset.copy(), self.map_index, self.map_coord,
                            self.tile_center.copy(), self.merge_status, self.transition_status, self.to_local.copy(),
                            None if self.corner_data is None else [(p.copy(), m.copy()) for p, m in self.corner_data],
                            None if self.edge_data is None else [(p.copy(), n.copy()) for p, n in self.edge_data])


def status_hash(st):
    """TileTransitionStatusHash::from_status, structure.rs:576-584"""
    if st[0] == "spawning":
        return ("spawning",)
    return st


class WangTile:
    """wangtile.rs:18-1847"""

    def __init__(self, pp: orc.Preprocessed):
        self.pp = pp                                   # preprocess() output (wangtile.rs:71-255)
        self.n_tiles = (pp.n_lod, pp.n_tile, pp.n_view)
        self.user = UserData()
        self.initialized = False
        self.tile_map = None
        self.neighbor_map = None
        self.center_coord = (0, 0)
        self.camera_pos = v3(0, 0, 0)
        self.rng = StdRng(0)
        self.cache = {}
        self.cache_order = []

    # ---- topology --------------------------------------------------------------
    def compute_map_neighbors(self, mc):
        """wangtile.rs:257-338 (plane / height-map branch); slots 0=W,1=N,2=E,3=S, each
        (neighbor map coord, which slot this tile is for that neighbor)."""
        w, h = self.user.tile_map_wh
        x, y = mc
        nb = [None, None, None, None]
        if self.user.surface_type == SURFACE_SPHERE:                     # :260-319, 5 x 2 icosahedral-strip blocks
            block_w = w // 5
            bidx, bidy = 5 * x // w, 2 * y // h
            bx, by = x - bidx * block_w, y - bidy * block_w
            if bx > 0:
                nb[0] = ((x - 1, y), 2)
            elif bidy == 0:
                nb[0] = (((w + x - 1) % w, y + block_w), 2)
            else:
                nb[0] = (((w + x - by - 1) % w, h - 1), 1)
            if bx < block_w - 1:
                nb[2] = ((x + 1, y), 0)
            elif bidy == 0:
                nb[2] = (((x + block_w - by) % w, 0), 3)
            else:
                nb[2] = (((x + 1) % w, y - block_w), 0)
            if y > 0:
                nb[3] = ((x, y - 1), 1)
            else:
                nb[3] = (((w + bidx * block_w - 1) % w, block_w - 1 - bx), 2)
            if y < h - 1:
                nb[1] = ((x, y + 1), 3)
            else:
                nb[1] = (((bidx * block_w + block_w) % w, 2 * block_w - 1 - bx), 0)
            return nb
        if x > 0:
            nb[0] = ((x - 1, y), 2)
        if x < w - 1:
            nb[2] = ((x + 1, y), 0)
        if y > 0:
            nb[3] = ((x, y - 1), 1)
        if y < h - 1:
            nb[1] = ((x, y + 1), 3)
        return nb

    # ---- configure -------------------------------------------------------------
    def configure(self, user: UserData) -> UserData:
        """wangtile.rs:349-432"""
        self.initialized = False
        self.user = user
        u = self.user
        if u.reset_rng:
            self.rng = StdRng(0)
        if u.surface_type == SURFACE_SPHERE:
            u.tile_map_wh = (u.tile_map_half_wh[0] * 2, u.tile_map_half_wh[1] * 2)
            assert u.tile_map_wh[0] * 2 == u.tile_map_wh[1] * 5
        else:
            u.tile_map_wh = (u.tile_map_half_wh[0] * 2 + 1, u.tile_map_half_wh[1] * 2 + 1)
        w, h = u.tile_map_wh
        self.tile_map = [[None] * h for _ in range(w)]
        assert self.n_tiles[1] // 16 >= u.center_option
        self.neighbor_map = [[self.compute_map_neighbors((i, j)) for j in range(h)] for i in range(w)]
        hw, hh = u.height_map_wh
        hm = []
        for i in range(hh):
            for j in range(hw):
                t = u.height_map_type
                if t == HMAP_TEXTURE:
                    v = f32(0.0)
                elif t == HMAP_RANDOM:
                    v = self.rng.random_range_f32_inclusive(-1.0, 1.0)
                elif t == HMAP_SLOPEX:
                    v = f32(f32(f32(j) / f32(hh)) * f32(2.0)) - f32(1.0)
                elif t == HMAP_SLOPEY:
                    v = f32(f32(f32(i) / f32(hh)) * f32(2.0)) - f32(1.0)
                else:
                    v = f32(f32(f32(i) / f32(hw)) + f32(f32(j) / f32(hh))) - f32(1.0)
                hm.append(f32(v))
        hm = np.array(hm, dtype=f32)
        if u.height_map_type == HMAP_TEXTURE and u.height_tex is not None:
            u.height_map_wh = tuple(u.height_tex[1])
            hm = np.array(u.height_tex[0], dtype=f32).ravel().copy()
        hm = (hm * f32(f32(u.tile_width) * f32(u.height_map_scale[2]))).astype(f32)
        if u.height_map_type == HMAP_RANDOM:
            hm = self.map_resize(hm, u.height_map_wh, (1024, 1024))
            u.height_map_wh = (1024, 1024)
        u.height_map = hm
        s_n = self.pp.lod_avg_scale[-1]
        u.lod_transition_dist = [f32(f32(f32(u.lod_max_dist) * s) / s_n) for s in self.pp.lod_avg_scale]
        self.cache = {}
        self.cache_order = []
        u.n_tiles = self.n_tiles
        return u

    # ---- height map helpers ----------------------------------------------------
    @staticmethod
    def _cubic_weight(t):
        t = f32(t)
        return [f32(f32(f32(f32(f32(-0.5) * t) + f32(1.0)) * t - f32(0.5)) * t),
                f32(f32(f32(f32(f32(1.5) * t) - f32(2.5)) * t) * t + f32(1.0)),
                f32(f32(f32(f32(f32(-1.5) * t) + f32(2.0)) * t + f32(0.5)) * t),
                f32(f32(f32(f32(f32(0.5) * t) - f32(0.5)) * t) * t)]

    def map_resize(self, hm, from_wh, to_wh):
        """wangtile.rs:1292-1349 (bicubic, vectorised over output pixels; identical op order)."""
        fw, fh = from_wh
        tw, th = to_wh
        src = np.asarray(hm, dtype=f32).reshape(fh, fw)
        ii = np.arange(tw, dtype=f32)
        jj = np.arange(th, dtype=f32)
        ux = (ii / f32(tw)).astype(f32)
        uy = (jj / f32(th)).astype(f32)
        x = (ux * f32(fw) - f32(0.5)).astype(f32)
        y = (uy * f32(fh) - f32(0.5)).astype(f32)
        x0 = np.floor(x).astype(np.int64)
        y0 = np.floor(y).astype(np.int64)
        dx = (x - x0.astype(f32)).astype(f32)
        dy = (y - y0.astype(f32)).astype(f32)

        def cw(t):
            return [(((f32(-0.5) * t + f32(1.0)) * t - f32(0.5)) * t).astype(f32),
                    (((f32(1.5) * t - f32(2.5)) * t) * t + f32(1.0)).astype(f32),
                    (((f32(-1.5) * t + f32(2.0)) * t + f32(0.5)) * t).astype(f32),
                    (((f32(0.5) * t - f32(0.5)) * t) * t).astype(f32)]
        wx, wy = cw(dx), cw(dy)
        out = np.zeros((th, tw), dtype=f32)
        for j in range(4):
            yi = ((y0 + j - 1) % fh + fh) % fh
            for i in range(4):
                xi = ((x0 + i - 1) % fw + fw) % fw
                val = src[yi[:, None], xi[None, :]]
                out = (out + ((val * wx[i][None, :]).astype(f32) * wy[j][:, None]).astype(f32)).astype(f32)
        return out.ravel()

    def map_fetch_bilinear_with_auxiliary(self, uv, dt):
        """wangtile.rs:1220-1290"""
        hm, (width, height) = self.user.height_map, self.user.height_map_wh

        def texel(xx, yy):
            return hm[(((yy % height) + height) % height) * width + (((xx % width) + width) % width)]
        x = f32(f32(uv[0] * f32(width)) - f32(0.5))
        y = f32(f32(uv[1] * f32(height)) - f32(0.5))
        dx, dy = f32(f32(dt) * f32(width)), f32(f32(dt) * f32(height))
        x0, y0 = int(math.floor(x)), int(math.floor(y))
        tx, ty = f32(x - f32(x0)), f32(y - f32(y0))
        i00, i10, i01, i11 = texel(x0, y0), texel(x0 + 1, y0), texel(x0, y0 + 1), texel(x0 + 1, y0 + 1)
        one = f32(1.0)

        def bil(ax, ay):
            i0 = f32(f32(i00 * f32(one - ax)) + f32(i10 * ax))
            i1 = f32(f32(i01 * f32(one - ax)) + f32(i11 * ax))
            return f32(f32(i0 * f32(one - ay)) + f32(i1 * ay))
        return [bil(tx, ty), bil(f32(tx + dx), ty), bil(f32(tx - dx), ty), bil(tx, f32(ty + dy)), bil(tx, f32(ty - dy))]

    def surface_mapping(self, map_coord, pos, to_world):
        """wangtile.rs:1352-1494 -> (new_pos, transform)"""
        u = self.user
        if u.surface_type == SURFACE_NONE:
            return pos.copy(), mat3_identity()
        if u.surface_type == SURFACE_SPHERE:
            return self._surface_mapping_sphere(map_coord, pos, to_world)
        DELTA = f32(0.001)
        tw = f32(u.tile_width)
        xr = f32(f32(f32(u.tile_map_wh[0]) * tw) * f32(u.height_map_scale[0]))
        yr = f32(f32(f32(u.tile_map_wh[1]) * tw) * f32(u.height_map_scale[1]))
        uu = f32(f32(pos[0] + f32(f32(u.tile_map_half_wh[0]) * tw)) / xr)
        vv = f32(f32(pos[1] + f32(f32(u.tile_map_half_wh[1]) * tw)) / yr)
        hv = self.map_fetch_bilinear_with_auxiliary((uu, vv), DELTA)
        hz = f32(u.height_map_scale[2])
        new_pos = pos.copy()
        new_pos[2] = f32(hv[0] * hz)
        h_r, h_l, h_u, h_d = f32(hv[1] * hz), f32(hv[2] * hz), f32(hv[3] * hz), f32(hv[4] * hz)
        lx = v3(1.0, 0.0, f32(f32(h_r - h_l) / f32(f32(f32(2.0) * DELTA) * xr)))
        ly = v3(0.0, 1.0, f32(f32(h_u - h_d) / f32(f32(f32(2.0) * DELTA) * yr)))
        lz = normalize3(cross3(lx, ly))
        l2w = mat3_cols(lx, ly, lz)
        new_pos = (new_pos + mat3_vec(l2w, v3(0.0, 0.0, pos[2]))).astype(f32)
        return new_pos, (l2w if to_world else mat3_invert(l2w))

    @staticmethod
    def _sincos(x):
        """The canonical sin / cos of DESIGN.md section 4 (Rust's f32::sin / cos are platform libm calls whose
        last bits are unpinnable); evaluated by the C oracle so that every side shares one implementation."""
        import ctypes as C
        lib = orc.lib()
        lib.orc_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.orc_sincosf.restype = None
        sn, cs = C.c_float(), C.c_float()
        lib.orc_sincosf(C.c_float(float(x)), C.byref(sn), C.byref(cs))
        return f32(sn.value), f32(cs.value)

    def _sphere_point(self, block_w, bidx, bidy, bx, by):
        """get_uv + uv_to_pos closures, wangtile.rs:1410-1461"""
        PI = f32(math.pi)
        five, three = f32(5.0), f32(3.0)
        if bidy == 0.0:
            if by < bx:
                if f32(bx - by) == block_w:
                    u = f32(0.0)
                else:
                    u = f32(f32(f32(by / f32(block_w - f32(bx - by))) + bidx) / five)
                v = f32(f32(f32(block_w - f32(bx - by)) / block_w) / three)
            else:
                u = f32(f32(f32(f32(bx / block_w) + bidx) / five) + f32(f32(f32(by - bx) / block_w) * f32(0.1)))
                v = f32(f32(f32(f32(by - bx) / block_w) / three) + f32(f32(1.0) / three))
        else:
            if by < bx:
                u = f32(f32(f32(f32(bx / block_w) + bidx) / five) + f32(f32(f32(block_w - f32(bx - by)) / block_w) * f32(0.1)))
                v = f32(f32(f32(f32(block_w - f32(bx - by)) / block_w) / three) + f32(f32(1.0) / three))
            else:
                if f32(by - bx) == block_w:
                    u = f32(0.0)
                else:
                    u = f32(f32(f32(f32(bx / f32(block_w - f32(by - bx))) + bidx) / five) + f32(0.1))
                v = f32(f32(f32(f32(by - bx) / block_w) / three) + f32(f32(2.0) / three))
        u = f32(u + f32(f32(0.5) * f32(np.floor(v))))
        u = f32(u * f32(f32(2.0) * PI))
        v = f32(f32(v - f32(0.5)) * PI)
        su, cu = self._sincos(u)
        sv, cv = self._sincos(v)
        return v3(f32(cv * cu), f32(cv * su), sv)

    def _surface_mapping_sphere(self, map_coord, pos, to_world):
        """wangtile.rs:1406-1488"""
        u = self.user
        DELTA = f32(0.001)
        tw = f32(u.tile_width)
        xmax, ymax = f32(f32(u.tile_map_wh[0]) * tw), f32(f32(u.tile_map_wh[1]) * tw)
        block_w = f32(xmax / f32(5.0))
        new_pos = (pos - self.coord_to_pos(self.map_to_coord((0, 0)))).astype(f32)
        bidx = f32(5 * map_coord[0] // u.tile_map_wh[0])
        bidy = f32(2 * map_coord[1] // u.tile_map_wh[1])
        bx = f32(new_pos[0] - f32(bidx * block_w))
        by = f32(new_pos[1] - f32(bidy * block_w))
        lz = self._sphere_point(block_w, bidx, bidy, bx, by)
        r = f32(u.sphere_radius)
        new_pos = (lz * r).astype(f32)
        dt = f32(DELTA * ymax)
        pr = (self._sphere_point(block_w, bidx, bidy, f32(bx + dt), by) * r).astype(f32)
        pl = (self._sphere_point(block_w, bidx, bidy, f32(bx - dt), by) * r).astype(f32)
        pu = (self._sphere_point(block_w, bidx, bidy, bx, f32(by + dt)) * r).astype(f32)
        pd = (self._sphere_point(block_w, bidx, bidy, bx, f32(by - dt)) * r).astype(f32)
        two_dt = f32(f32(2.0) * dt)
        lx = ((pr - pl).astype(f32) / two_dt).astype(f32)
        ly = ((pu - pd).astype(f32) / two_dt).astype(f32)
        l2w = mat3_cols(lx, ly, lz)
        new_pos = (new_pos + mat3_vec(l2w, v3(0.0, 0.0, pos[2]))).astype(f32)
        return new_pos, (l2w if to_world else mat3_invert(l2w))

    # ---- coordinates -----------------------------------------------------------
    def coord_to_pos(self, c):
        tw = f32(self.user.tile_width)
        return v3(f32(f32(c[0]) * tw), f32(f32(c[1]) * tw), 0.0)

    def pos_to_coord(self, p):
        tw = f32(self.user.tile_width)
        return (int(math.floor(f32(p[0] / tw))), int(math.floor(f32(p[1] / tw))))

    def index_to_map(self, idx):
        h = self.user.tile_map_wh[1]
        return (idx // h, idx % h)

    def map_to_index(self, mc):
        return mc[0] * self.user.tile_map_wh[1] + mc[1]

    def map_to_coord(self, mc):
        return (mc[0] + self.center_coord[0] - self.user.tile_map_half_wh[0],
                mc[1] + self.center_coord[1] - self.user.tile_map_half_wh[1])

    def coord_to_map(self, c):
        return (c[0] - self.center_coord[0] + self.user.tile_map_half_wh[0],
                c[1] - self.center_coord[1] + self.user.tile_map_half_wh[1])

    @staticmethod
    def tile_id_to_color(tid):
        return (tid % 16 // 8 % 2, tid % 16 // 4 % 2, tid % 16 // 2 % 2, tid % 16 % 2)

    @staticmethod
    def color_to_tile_id(color, center_idx):
        return color[0] * 8 + color[1] * 4 + color[2] * 2 + color[3] + 16 * center_idx

    # ---- tile map --------------------------------------------------------------
    def compute_corner_edge(self, mc, tile_center_z):
        """wangtile.rs:1609-1669"""
        u = self.user
        if u.tile_sort_type != SORT_GRAPH and u.merge_type != MERGE_EDGE:
            return None, None
        d_coords = [(0, 0), (0, 1), (1, 1), (1, 0)]
        corners = [None] * 4
        for ci in range(4):
            done = False
            nb = self.neighbor_map[mc[0]][mc[1]][ci]
            if nb is not None:
                n_inst = self.tile_map[nb[0][0]][nb[0][1]]
                if n_inst is not None and n_inst.corner_data is not None:
                    p, m = n_inst.corner_data[(nb[1] + 1) % 4]
                    corners[ci] = (p.copy(), m.copy())
                    done = True
            if not done:
                nb = self.neighbor_map[mc[0]][mc[1]][(ci + 3) % 4]
                if nb is not None:
                    n_inst = self.tile_map[nb[0][0]][nb[0][1]]
                    if n_inst is not None and n_inst.corner_data is not None:
                        p, m = n_inst.corner_data[nb[1]]
                        corners[ci] = (p.copy(), m.copy())
                        done = True
            if not done:
                cmc = (mc[0] + d_coords[ci][0], mc[1] + d_coords[ci][1])
                cpos = (self.coord_to_pos(self.map_to_coord(cmc)) + v3(0, 0, 1) * f32(tile_center_z)).astype(f32)
                corners[ci] = self.surface_mapping(mc, cpos, True)
        edges = [None] * 4
        for ei in range(4):
            c1p, c1m = corners[ei]
            c2p, c2m = corners[(ei + 1) % 4]
            epos = ((c1p + c2p).astype(f32) / f32(2.0)).astype(f32)
            cdir = (c2p - c1p).astype(f32)
            n1 = mat3_vec(c1m, v3(0, 0, 1))
            n2 = mat3_vec(c2m, v3(0, 0, 1))
            normal = ((n1 + n2).astype(f32) / f32(2.0)).astype(f32)
            edges[ei] = (epos, normalize3(cross3(normal, cdir)))
        return corners, edges

    def update_tile_map(self, camera_pos):
        """wangtile.rs:1671-1781"""
        u = self.user
        xmax, ymax = u.tile_map_wh
        self.camera_pos = np.asarray(camera_pos, dtype=f32)
        prev_center = self.center_coord
        if u.surface_type == SURFACE_SPHERE:                              # the sphere map never shifts, :1721-1723
            self.center_coord = (0, 0)
        else:
            self.center_coord = self.pos_to_coord(self.camera_pos)
        new_map = [[None] * ymax for _ in range(xmax)]
        for i in range(xmax if u.surface_type != SURFACE_SPHERE else 0):
            for j in range(ymax):
                px = i + self.center_coord[0] - prev_center[0]
                py = j + self.center_coord[1] - prev_center[1]
                if 0 <= px < xmax and 0 <= py < ymax:
                    prev = self.tile_map[px][py]
                    if prev is not None:
                        new_map[i][j] = TileInstance((0, prev.tid[1]), 0, prev.tile_offset, self.map_to_index((i, j)),
                                                     (i, j), prev.tile_center, ("none",), ("none",), prev.to_local,
                                                     None if prev.corner_data is None else [(p.copy(), m.copy()) for p, m in prev.corner_data],
                                                     None if prev.edge_data is None else [(p.copy(), n.copy()) for p, n in prev.edge_data])
        if u.surface_type != SURFACE_SPHERE:
            self.tile_map = new_map
        for i in range(xmax):
            for j in range(ymax):
                if self.tile_map[i][j] is not None:
                    continue
                mc = (i, j)
                tile_offset = self.coord_to_pos(self.map_to_coord(mc))
                color = [0, 0, 0, 0]
                for idx in range(4):
                    nb = self.neighbor_map[i][j][idx]
                    if nb is not None:
                        nt = self.tile_map[nb[0][0]][nb[0][1]]
                        if nt is not None:
                            color[idx] = self.tile_id_to_color(nt.tid[1])[nb[1]]
                        else:
                            color[idx] = self.rng.random_range_u32(2)
                    else:
                        color[idx] = self.rng.random_range_u32(2)
                center_option = self.rng.random_range_u32(u.center_option)
                tile_id = self.color_to_tile_id(color, center_option)
                base_center = self.pp.tile_center[tile_id]
                tile_center = (base_center + tile_offset).astype(f32)
                tile_center, to_local = self.surface_mapping(mc, tile_center, False)
                corner, edge = self.compute_corner_edge(mc, base_center[2])
                self.tile_map[i][j] = TileInstance((0, tile_id), 0, tile_offset, self.map_to_index(mc), mc, tile_center,
                                                   ("none",), ("none",), to_local, corner, edge)
        self.update_lod(self.camera_pos)

    def lod_select_spatial(self, mc, cam_pos):
        """wangtile.rs:1496-1569"""
        u = self.user
        pos_offset = self.coord_to_pos(self.map_to_coord(mc))
        ti = self.tile_map[mc[0]][mc[1]]
        tile = ti.tid[1]
        D = u.lod_transition_dist
        center_dist = dist3(ti.tile_center, cam_pos)
        sel = len(D) - 1
        for l, td in enumerate(D):
            if center_dist <= td:
                sel = l
                break
        status = ("none",)
        if u.lod_blending:
            lo, hi = self.pp.aabb[tile, 0], self.pp.aabb[tile, 1]
            if u.lod_bbox_check:
                pts = [lo, v3(lo[0], lo[1], hi[2]), v3(lo[0], hi[1], lo[2]), v3(lo[0], hi[1], hi[2]),
                       v3(hi[0], lo[1], lo[2]), v3(hi[0], lo[1], hi[2]), v3(hi[0], hi[1], lo[2]), hi]
            else:
                pts = [self.pp.tile_center[tile]]
            mn, mx = f32(-1.0), f32(-1.0)
            for p in pts:
                q, _ = self.surface_mapping(mc, (p + pos_offset).astype(f32), True)
                d = dist3(q, cam_pos)
                if mn < 0 or d < mn:
                    mn = d
                if mx < 0 or d > mx:
                    mx = d
            r, tol = f32(u.lod_transition_width_ratio), f32(u.lod_dist_tolerance)
            if sel > 0:
                if mn < f32(f32(D[sel - 1] * f32(f32(1.0) + r)) + tol):
                    status = ("changing", False)
            if sel < len(D) - 1:
                if mx > f32(f32(D[sel] * f32(f32(1.0) - r)) - tol):
                    status = ("changing", True)
        return sel, status

    def update_lod(self, camera_pos):
        """wangtile.rs:1571-1607"""
        u = self.user
        xmax, ymax = u.tile_map_wh
        cc = self.coord_to_pos(self.center_coord)
        cam_u = f32(f32(camera_pos[0] - cc[0]) / f32(u.tile_width))
        cam_v = f32(f32(camera_pos[1] - cc[1]) / f32(u.tile_width))
        one = f32(1.0)
        for i in range(xmax):
            for j in range(ymax):
                lod, status = self.lod_select_spatial((i, j), camera_pos)
                ti = self.tile_map[i][j]
                ti.tid = (lod, ti.tid[1])
                ti.transition_status = status
                if u.lod_blending and u.surface_type != SURFACE_SPHERE:
                    bf = one
                    if i == 0:
                        bf = f32(bf * f32(one - cam_u))
                    elif i == xmax - 1:
                        bf = f32(bf * cam_u)
                    if j == 0:
                        bf = f32(bf * f32(one - cam_v))
                    elif j == ymax - 1:
                        bf = f32(bf * cam_v)
                    if bf != one:
                        ti.transition_status = ("spawning", bf)

    def check_update(self, camera_pos):
        """wangtile.rs:692-699"""
        if not self.initialized:
            return True
        return dist2_3(np.asarray(camera_pos, dtype=f32), self.camera_pos) >= f32(self.user.update_distance2)

    def build_tiles(self, camera_pos):
        """wangtile.rs:434-474 -> SceneData dict"""
        self.initialized = True
        self.update_tile_map(np.asarray(camera_pos, dtype=f32))
        n_lod = self.n_tiles[0]
        sd = {"scene_id": 0, "splat_count": 0, "blending_splat_count": 0, "center_coord": self.center_coord,
              "lod_splat_count": [0] * n_lod, "lod_instance_count": [0] * n_lod}
        cnt = lambda l, t: int(self.pp.gs_index[l][t][0].shape[0])
        for i in range(self.user.tile_map_wh[0]):
            for j in range(self.user.tile_map_wh[1]):
                ti = self.tile_map[i][j]
                l, t = ti.tid
                sd["splat_count"] += cnt(l, t)
                sd["blending_splat_count"] += cnt(l, t)
                sd["lod_splat_count"][l] += cnt(l, t)
                sd["lod_instance_count"][l] += 1
                blend_lower = l < n_lod - 1
                if ti.transition_status[0] == "changing" and not ti.transition_status[1]:
                    sd["blending_splat_count"] += cnt(l - 1, t)
                    blend_lower = False
                if blend_lower:
                    sd["blending_splat_count"] += cnt(l + 1, t)
        return sd

    # ---- selective merging -----------------------------------------------------
    def selective_merge_edge(self, camera_pos, view_proj):
        """wangtile.rs:827-1027"""
        u = self.user
        xmax, ymax = u.tile_map_wh
        edge_vec = []
        check = [[False] * ymax for _ in range(xmax)]
        for i in range(xmax):
            for j in range(ymax):
                mi = self.map_to_index((i, j))
                check[i][j] = True
                ti = self.tile_map[i][j]
                ti.merge_status = ("none",)
                for n_i in range(4):
                    nb = self.neighbor_map[i][j][n_i]
                    if nb is None:
                        continue
                    if check[nb[0][0]][nb[0][1]]:
                        continue
                    epos, enorm = ti.edge_data[n_i]
                    c1p, c1m = ti.corner_data[n_i]
                    c2p, c2m = ti.corner_data[(n_i + 1) % 4]
                    vd = (epos - camera_pos).astype(f32)
                    vlen = mag3(vd)
                    if vd[0] == 0 and vd[1] == 0 and vd[2] == 0:
                        continue
                    if dot3(vd, c1m[6:9]) > 0 or dot3(vd, c2m[6:9]) > 0:
                        continue
                    p1 = orc.mat4_vec(view_proj, [c1p[0], c1p[1], c1p[2], f32(1.0)])
                    p1 = (p1[:3] / p1[3]).astype(f32)
                    p2 = orc.mat4_vec(view_proj, [c2p[0], c2p[1], c2p[2], f32(1.0)])
                    p2 = (p2[:3] / p2[3]).astype(f32)
                    clip = f32(1.0)

                    def outside(p):
                        return p[2] < -clip or p[0] < -clip or p[0] > clip or p[1] < -clip or p[1] > clip
                    if outside(p1) and outside(p2):
                        continue
                    dabs = f32(abs(dot3(enorm, vd)))
                    edge_vec.append((mi, n_i, dabs, f32(dabs / vlen)))
        edge_vec.sort(key=lambda e: float(e[2]))      # stable, ascending |n.v|
        topk = 0
        merge_map = [[None] * ymax for _ in range(xmax)]
        groups = []
        for (mi, ei, _, ndot) in edge_vec:
            if topk >= u.merge_topk:
                break
            if ndot > f32(u.merge_dot_threshold):
                continue
            mc = self.index_to_map(mi)
            nmc = self.neighbor_map[mc[0]][mc[1]][ei][0]
            ni = self.map_to_index(nmc)
            a, b = merge_map[mc[0]][mc[1]], merge_map[nmc[0]][nmc[1]]
            if a is None and b is None:
                groups.append([mi, ni])
                merge_map[mc[0]][mc[1]] = merge_map[nmc[0]][nmc[1]] = len(groups) - 1
            elif a is not None and b is None:
                groups[a].append(ni)
                merge_map[nmc[0]][nmc[1]] = a
            elif a is None and b is not None:
                groups[b].append(mi)
                merge_map[mc[0]][mc[1]] = b
            elif a != b:
                for g in groups[b]:
                    gm = self.index_to_map(g)
                    merge_map[gm[0]][gm[1]] = a
                groups[a].extend(groups[b])
                groups[b] = []
            topk += 1
        for i in range(len(groups)):                   # fix non-convex groups, :959-990
            seen = set()
            j = 0
            while j < len(groups[i]):
                tmc = self.index_to_map(groups[i][j])
                for nb in self.neighbor_map[tmc[0]][tmc[1]]:
                    if nb is None:
                        continue
                    nmc = nb[0]
                    nidx = self.map_to_index(nmc)
                    if nidx not in groups[i]:
                        if nidx in seen:
                            other = merge_map[nmc[0]][nmc[1]]
                            if other is not None:
                                for g in groups[other]:
                                    gm = self.index_to_map(g)
                                    merge_map[gm[0]][gm[1]] = i
                                groups[i].extend(groups[other])
                                groups[other] = []
                            else:
                                groups[i].append(nidx)
                                merge_map[nmc[0]][nmc[1]] = i
                        else:
                            seen.add(nidx)
                j += 1
        for g in groups:
            if not g:
                continue
            g = sorted(g)
            mind, mini = f32(np.finfo(np.float32).max), 0
            for k, idx in enumerate(g):
                mc = self.index_to_map(idx)
                d2 = dist2_3(self.tile_map[mc[0]][mc[1]].tile_center, camera_pos)
                if mind > d2:
                    mind, mini = d2, k
            for k, idx in enumerate(g):
                if k != mini:
                    mc = self.index_to_map(idx)
                    self.tile_map[mc[0]][mc[1]].merge_status = ("to", g[mini])
            mc = self.index_to_map(g[mini])
            self.tile_map[mc[0]][mc[1]].merge_status = ("from", list(g))

    def selective_merge_axis(self, camera_pos, view_proj):
        """wangtile.rs:722-825 (plane / height map)"""
        u = self.user
        center_mc = self.coord_to_map(self.center_coord)
        if u.surface_type == SURFACE_SPHERE:                               # nearest not-MergedTo tile, :725-740
            min_dist, center_mc = f32(-1.0), (0, 0)
            for idx in range(u.tile_map_wh[0] * u.tile_map_wh[1]):
                mc = self.index_to_map(idx)
                ti = self.tile_map[mc[0]][mc[1]]
                if ti.merge_status[0] == "to":
                    continue
                d = dist2_3(ti.tile_center, np.asarray(camera_pos, dtype=f32))
                if min_dist < 0.0 or d < min_dist:
                    min_dist, center_mc = d, mc
        nbs = self.neighbor_map[center_mc[0]][center_mc[1]]
        best, merge_dir = f32(0.0), -1
        cam_dir = normalize3(v3(view_proj[2], view_proj[6], view_proj[10]))
        for ci in range(4):
            if nbs[ci] is not None:
                mc = nbs[ci][0]
                tp = self.tile_map[mc[0]][mc[1]].tile_center
                dp = dot3(normalize3((tp - camera_pos).astype(f32)), cam_dir)
                if best < dp:
                    best, merge_dir = dp, ci
        if merge_dir < 0:
            return
        merge_neighbors = [(3, 1), (0, 2), (1, 3), (2, 0)]
        mcc = center_mc
        for _ in range(u.merge_tile_dist[0]):
            mcc = self.neighbor_map[mcc[0]][mcc[1]][merge_dir][0]
        for _ in range(u.merge_tile_dist[0], u.merge_tile_dist[1]):
            cidx = self.map_to_index(mcc)
            nb = self.neighbor_map[mcc[0]][mcc[1]]
            n1, n2 = nb[merge_neighbors[merge_dir][0]][0], nb[merge_neighbors[merge_dir][1]][0]
            vec = [self.map_to_index(n1), cidx, self.map_to_index(n2)]
            if (self.tile_map[mcc[0]][mcc[1]].merge_status != ("none",) or self.tile_map[n1[0]][n1[1]].merge_status != ("none",)
                    or self.tile_map[n2[0]][n2[1]].merge_status != ("none",)):
                break
            self.tile_map[mcc[0]][mcc[1]].merge_status = ("from", vec)
            self.tile_map[n1[0]][n1[1]].merge_status = ("to", cidx)
            self.tile_map[n2[0]][n2[1]].merge_status = ("to", cidx)
            mcc = nb[merge_dir][0]

    # ---- tile ordering ---------------------------------------------------------
    def _n_instance(self):
        return self.user.tile_map_wh[0] * self.user.tile_map_wh[1]

    def sort_tiles_object_pos(self, camera_pos):
        """wangtile.rs:1029-1047"""
        sv = []
        for idx in range(self._n_instance()):
            mc = self.index_to_map(idx)
            ti = self.tile_map[mc[0]][mc[1]]
            if ti.merge_status[0] == "to":
                continue
            sv.append((idx, dist2_3(camera_pos, ti.tile_center)))
        sv.sort(key=lambda e: float(e[1]))
        sv.reverse()
        return [e[0] for e in sv]

    def sort_tiles_object_vp(self, vp):
        """wangtile.rs:1049-1070"""
        sv = []
        for idx in range(self._n_instance()):
            mc = self.index_to_map(idx)
            ti = self.tile_map[mc[0]][mc[1]]
            if ti.merge_status[0] == "to":
                continue
            p = ti.tile_center
            d = f32(f32(f32(vp[2] * p[0]) + f32(vp[6] * p[1])) + f32(vp[10] * p[2]))
            sv.append((idx, d))
        sv.sort(key=lambda e: float(e[1]))
        sv.reverse()
        return [e[0] for e in sv]

    def sort_tiles_object_bfs(self, camera_pos):
        """wangtile.rs:1072-1113"""
        min_mc, min_d = (0, 0), f32(-1.0)
        for idx in range(self._n_instance()):
            mc = self.index_to_map(idx)
            ti = self.tile_map[mc[0]][mc[1]]
            if ti.merge_status[0] == "to":
                continue
            d = dist2_3(camera_pos, ti.tile_center)
            if min_d < 0 or d < min_d:
                min_d, min_mc = d, mc
        xmax, ymax = self.user.tile_map_wh
        check = [[False] * ymax for _ in range(xmax)]
        out, queue = [], [min_mc]
        check[min_mc[0]][min_mc[1]] = True
        while queue:
            mc = queue.pop(0)
            out.append(self.map_to_index(mc))
            for n_i in range(4):
                nb = self.neighbor_map[mc[0]][mc[1]][n_i]
                if nb is not None and not check[nb[0][0]][nb[0][1]]:
                    queue.append(nb[0])
                    check[nb[0][0]][nb[0][1]] = True
        out.reverse()
        return out

    def sort_tiles_object_graph(self, camera_pos):
        """wangtile.rs:1115-1218"""
        xmax, ymax = self.user.tile_map_wh
        g = DiGraph()
        node_map = [[None] * ymax for _ in range(xmax)]
        for i in range(xmax):
            for j in range(ymax):
                if self.tile_map[i][j].merge_status[0] != "to":
                    node_map[i][j] = g.add_node(self.map_to_index((i, j)))
        check = [[False] * ymax for _ in range(xmax)]

        def node_of(mc):
            ti = self.tile_map[mc[0]][mc[1]]
            if ti.merge_status[0] == "to":
                t = self.index_to_map(ti.merge_status[1])
                return node_map[t[0]][t[1]]
            return node_map[mc[0]][mc[1]]
        for i in range(xmax):
            for j in range(ymax):
                this = self.tile_map[i][j]
                this_node = node_of((i, j))
                check[i][j] = True
                for n_i in range(4):
                    nb = self.neighbor_map[i][j][n_i]
                    if nb is None or check[nb[0][0]][nb[0][1]]:
                        continue
                    nnode = node_of(nb[0])
                    if this_node == nnode:
                        continue
                    epos, enorm = this.edge_data[n_i]
                    vd = (epos - camera_pos).astype(f32)
                    if vd[0] == 0 and vd[1] == 0 and vd[2] == 0:
                        continue
                    dr = dot3(enorm, vd)
                    if dr > 0:
                        g.add_edge(this_node, nnode)
                    elif dr < 0:
                        g.add_edge(nnode, this_node)
        out, removed = [], []
        while True:
            order, cyc = g.toposort()
            if order is not None:
                for node in order:
                    if g.inc[node] or g.out[node]:
                        out.append(g.weights[node])
                break
            removed.append(g.weights[cyc])
            g.remove_node(cyc)
        out.extend(removed)
        out.reverse()
        return out

    def choose_presort_view(self, transform, pos, cam_pos):
        """wangtile.rs:701-718"""
        dl = mat3_vec(transform, normalize3((pos - cam_pos).astype(f32)))
        best, best_err = 0, f32(1000.0)
        for i, pd in enumerate(self.pp.presort_dirs):
            ex, ey, ez = f32(dl[0] - pd[0]), f32(dl[1] - pd[1]), f32(dl[2] - pd[2])
            err = f32(f32(f32(ex * ex) + f32(ey * ey)) + f32(ez * ez))
            if err < best_err:
                best, best_err = i, err
        return best

    # ---- sort_tiles --------------------------------------------------------------
    def sort_tiles(self, camera_pos, view_proj):
        """wangtile.rs:476-690 -> SortData dict {tile_instance_vec, render_data_vec[(key, value|None)]}"""
        u = self.user
        camera_pos = np.asarray(camera_pos, dtype=f32)
        view_proj = np.asarray(view_proj, dtype=f32)
        if u.merge_type == MERGE_AXIS:
            self.selective_merge_axis(camera_pos, view_proj)
        elif u.merge_type == MERGE_EDGE:
            self.selective_merge_edge(camera_pos, view_proj)
        order = {SORT_DISTANCE: lambda: self.sort_tiles_object_pos(camera_pos),
                 SORT_VIEWPORT: lambda: self.sort_tiles_object_vp(view_proj),
                 SORT_OBJECT: lambda: self.sort_tiles_object_bfs(camera_pos),
                 SORT_GRAPH: lambda: self.sort_tiles_object_graph(camera_pos)}[u.tile_sort_type]()
        inst_vec, rd_vec = [], []
        for mi in order:
            mc = self.index_to_map(mi)
            ti = self.tile_map[mc[0]][mc[1]]
            if ti.merge_status[0] == "from":
                from_vec = ti.merge_status[1]
                merge_x = merge_y = True
                avg_c = v3(0, 0, 0)
                avg_q = np.zeros(4, dtype=f32)
                tids, sts = [], []
                for m_mi in from_vec:
                    m_mc = self.index_to_map(m_mi)
                    if m_mc[0] != mc[0]:
                        merge_x = False
                    if m_mc[1] != mc[1]:
                        merge_y = False
                    mt = self.tile_map[m_mc[0]][m_mc[1]]
                    tids.append(mt.tid)
                    sts.append(status_hash(mt.transition_status))
                    avg_c = (avg_c + mt.tile_center).astype(f32)
                    avg_q = (avg_q + quat_from_mat3(mt.to_local)).astype(f32)
                if not merge_x and not merge_y:
                    view_id = len(self.pp.presort_dirs) - 1
                else:
                    n = f32(len(from_vec))
                    view_id = self.choose_presort_view(mat3_from_quat((avg_q / n).astype(f32)), (avg_c / n).astype(f32), camera_pos)
                key = (view_id, tuple(tids), tuple(sts))
            else:
                view_id = self.choose_presort_view(ti.to_local, ti.tile_center, camera_pos)
                key = (view_id, (ti.tid,), (status_hash(ti.transition_status),))
            new_ti = ti.clone()
            new_ti.view_id = view_id
            inst_vec.append(new_ti)
            value = None
            if ti.merge_status[0] == "from":
                from_vec = ti.merge_status[1]
                if u.use_cache and key in self.cache:
                    cv = self.cache[key]
                    self.cache_order.remove(key); self.cache_order.append(key)
                    new = dict(cv)
                    gmap = cv["gs_map_id"].copy()
                    old = cv["merge_from_vec"]
                    remap = {}                           # first matching member wins, :581-588
                    for jj in range(len(old)):
                        remap.setdefault(int(old[jj]), int(from_vec[jj]))
                    gmap = np.array([remap.get(int(x), int(x)) for x in cv["gs_map_id"]], dtype=np.uint32)
                    new["gs_map_id"] = gmap
                    rd_vec.append((key, new))
                    continue
                members = []
                for m_mi in from_vec:
                    m_mc = self.index_to_map(m_mi)
                    mt = self.tile_map[m_mc[0]][m_mc[1]]
                    st = mt.transition_status
                    members.append((m_mi, mt.tid[0], mt.tid[1], None if st[0] == "none" else st))
                value = orc.build_merged_value(self.pp, members, view_id, ti.tid[0])
                if u.use_cache:
                    self.cache[key] = dict(value)
                    self.cache_order.append(key)
                    while len(self.cache_order) > u.cache_size:
                        del self.cache[self.cache_order.pop(0)]
            rd_vec.append((key, value))
        return {"scene_id": 0, "tile_instance_vec": inst_vec, "render_data_vec": rd_vec}


# --------------------------------------------------------------------------
# Host half of GSWTRenderer::render: draw list from SortData (renderer.rs:466-591)
# --------------------------------------------------------------------------
def tile_uniforms_from_tile(ti: TileInstance, value):
    """TileUniforms::from_tile, renderer.rs:691-725 (+ :501-504)."""
    kw = dict(single_draw=0, map_index=ti.map_index, single_lod_id=-1, valid_lod_id=-1, changing=0,
              changing_to_lower=-1, tile_id=(ti.tid[0], ti.tid[1], ti.view_id),
              offset=(ti.tile_offset[0], ti.tile_offset[1], ti.tile_offset[2]), map_coord=ti.map_coord)
    if value is not None:
        kw["single_draw"] = 1
        kw["single_lod_id"] = value["single_lod_id"]
        kw["changing"] = 1 if value["single_lod_id"] == -1 else 0
    elif ti.transition_status[0] == "changing":
        kw["changing"] = 1
        kw["changing_to_lower"] = 1 if ti.transition_status[1] else 0
    else:
        kw["valid_lod_id"] = ti.tid[0]
    return orc.tile_uniforms(**kw)


def renderer_draws(pp: orc.Preprocessed, sort_data, view_proj, *, culling_dist=1.0, lod_enable=None):
    """The draw loop of renderer.rs:466-591 including the CPU tile cull (:472-494) and the
    lod_enable skip (:495).  Returns a list of orc.Draw in draw order."""
    draws = []
    vp = np.asarray(view_proj, dtype=f32)
    for ti, (key, value) in zip(sort_data["tile_instance_vec"], sort_data["render_data_vec"]):
        if len(key[1]) == 1:
            if ti.corner_data is None:
                raise RuntimeError("called `Option::unwrap()` on a `None` value (renderer.rs:476)")
            mx, my, mz = f32(np.finfo(np.float32).max), f32(np.finfo(np.float32).max), f32(-np.finfo(np.float32).max)
            for ci in range(4):
                p = ti.corner_data[ci][0]
                c = orc.mat4_vec(vp, [p[0], p[1], p[2], f32(1.0)])
                c = (c[:3] / c[3]).astype(f32)
                if abs(c[0]) < mx:
                    mx = f32(abs(c[0]))
                if abs(c[1]) < my:
                    my = f32(abs(c[1]))
                if c[2] > mz:
                    mz = c[2]
            clip = f32(culling_dist)
            if mz < -clip or mx > clip or my > clip:
                continue
        if lod_enable is not None and not lod_enable[ti.tid[0]]:
            continue
        tu = tile_uniforms_from_tile(ti, value)
        if value is not None:
            draws.append(orc.Draw(tu, value["gs_index"], value["gs_map_id"], value["gs_lod_id"]))
        else:
            l, t = ti.tid
            if ti.transition_status[0] == "changing" and not ti.transition_status[1]:
                bl = l - 1
            else:
                bl = l
            draws.append(orc.Draw(tu, pp.gs_index[bl][t][ti.view_id], None, pp.gs_lod_id[bl][t][ti.view_id],
                                  base=(bl, t, ti.view_id),
                                  corners=np.stack([ti.corner_data[ci][0] for ci in range(4)]) if ti.corner_data else None))
    return draws


def scene_uniforms_from_data(user: UserData, center_coord, *, splat_scale=1.0, scene_scale=(1.0, 1.0, 1.0),
                             height_map_scale_v=1.0, use_clip=0, clip_height=0.0, point_cloud_radius=0.0, draw_mode=0):
    """SceneUniforms::from_data, renderer.rs:631-672.  NB :646 reads `user_data.n_tiles.1`, i.e. the
    uniform named num_lod carries the TILE count (reference behaviour, kept)."""
    return orc.scene_uniforms(
        splat_scale=splat_scale, tile_width=user.tile_width, use_clip=use_clip, clip_height=clip_height,
        surface_type=user.surface_type, sphere_radius=user.sphere_radius, point_cloud_radius=point_cloud_radius,
        transition_width_ratio=user.lod_transition_width_ratio, num_lod=user.n_tiles[1], draw_mode=draw_mode,
        map_half_wh=user.tile_map_half_wh, center_coord=center_coord, transition_dist=user.lod_transition_dist,
        height_map_scale=(user.height_map_scale[0], user.height_map_scale[1],
                          float(f32(f32(user.height_map_scale[2]) * f32(height_map_scale_v)))),
        scene_scale=scene_scale)
