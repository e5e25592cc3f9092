// gswt_worker.hip -- the per-sort-event worker stages of WangTile on the device: update_lod (wangtile.rs:1496-1607),
// selective merging (:720-1027), the four tile orders (:1029-1218), the presort-view choice (:700-716) and the SortData records
// (:476-690, device-merge form) plus the host half of the draw loop (renderer.rs:466-591, TileUniforms::from_tile :691-725).
// Part of libgswt_hip.so; C ABI: the gswt_worker_* block of include/gswt_hip.h.  file:line are into zengyf131/gswt_renderer.
//
// What is parallel and what is not.  LOD selection, edge-candidate tests, order keys, view choice and record building are one
// thread per cell / cell side / record.  The merge union over the top-k edges, its convexity closure, the breadth-first order and
// petgraph's depth-first toposort are ORDER-DEFINED sequential algorithms (the result depends on the visiting order, and the
// contract is the reference's exact draw list), so they run on one lane of a single workgroup out of LDS, with every
// data-parallel part around them (edge numbering, adjacency lists, group finalisation) spread over the workgroup.
//
// Float discipline: the vector / matrix helpers and surface_mapping are the SAME sources libgswt_host compiles
// (host/gswt_math.h, host/gswt_surface.h under GSWT_HD), built with -ffp-contract=off: one rounding per operator, IEEE
// division and square root (hipcc's default for fp32), so every stage is bit-identical to gswt_wang_sort_tiles.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "../../include/gswt_hip.h"
#define GSWT_HD __host__ __device__
#include "host/gswt_math.h"
#include "host/gswt_surface.h"

namespace gswt {
int launch_sort(hipStream_t, uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, const unsigned long long*, int, uint32_t*, uint2* = nullptr, const uint32_t* = nullptr,
                uint32_t* = nullptr, uint32_t* = nullptr);
size_t radix_ws_words(uint32_t, int);
int ctx_device(const gswt_ctx*);
}  // namespace gswt

using namespace gswt_host;

namespace {

enum { SORT_DISTANCE = 0, SORT_VIEWPORT = 1, SORT_OBJECT = 2, SORT_GRAPH = 3 };
enum { MERGE_NONE = 0, MERGE_AXIS = 1, MERGE_EDGE = 2 };
enum { TR_NONE = 0, TR_SPAWNING = 1, TR_CHANGING_HIGHER = 2, TR_CHANGING_LOWER = 3 };
enum { MS_NONE = 0, MS_FROM = 1, MS_TO = 2 };
enum { ERR_LOD0_HIGHER = 1, ERR_NO_CORNERS = 2, ERR_AXIS_UNWRAP = 4 };
constexpr uint32_t kNoKey = 0xFFFFFFFFu;
constexpr uint16_t kNone16 = 0xFFFFu;

// counters of one sort event (device words)
enum { C_N_ORDER = 0, C_N_GROUPS, C_N_MEMBERS, C_ERR, C_N_NODES, C_POOL, C_MERGED_LO, C_MERGED_HI, C_COUNT };

struct WDev {
    SurfaceParams sp;
    int cells, n_lod, n_tile, n_view;
    int sort_type, merge_type;
    int lod_blending, lod_bbox_check;
    float lod_ratio, lod_tol;
    int merge_dist0, merge_dist1;
    float merge_dot_thr;
    uint32_t merge_topk;
    const float* lod_dist;
    const float* tile_center;
    const float* tile_aabb;
    const float* presort_dirs;
    const uint32_t* splat_count;
    const int32_t* nb;
    const gswt_cell* cell;
    gswt_cell_state* st;
    uint32_t* head_off;     // per cell: first member of the group it heads (into pool)
    uint32_t* head_len;
    uint32_t* pool;         // members of all groups, merged_from order
    uint32_t* counts;       // C_*
};

struct Mat4 { float m[16]; };

__device__ __forceinline__ V3 v3(const float* p) { return V3{p[0], p[1], p[2]}; }

// neighbour of cell c on side s: map index << 2 | the slot c occupies for it, or -1 (wangtile.rs:257-338; the planar case inline)
__device__ __forceinline__ int nbr(const WDev& w, int c, int s)
{
    if (w.sp.surface_type == 2) return w.nb[c * 4 + s];
    const int h = w.sp.map_h, x = c / h, y = c - x * h;
    if (s == 0) return x > 0 ? ((c - h) << 2) | 2 : -1;
    if (s == 2) return x < w.sp.map_w - 1 ? ((c + h) << 2) | 0 : -1;
    if (s == 3) return y > 0 ? ((c - 1) << 2) | 1 : -1;
    return y < h - 1 ? ((c + 1) << 2) | 3 : -1;
}

// ---- update_lod, wangtile.rs:1496-1607: one thread per cell ---------------------------------------------------------
__global__ __launch_bounds__(256) void k_w_lod(const WDev w, const V3 cam)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= w.cells) return;
    const int x = idx / w.sp.map_h, y = idx - x * w.sp.map_h;
    int ccx, ccy;
    sp_map_to_coord(w.sp, x, y, ccx, ccy);
    const V3 pos_offset = sp_coord_to_pos(w.sp, ccx, ccy);
    const gswt_cell& c = w.cell[idx];
    const float* D = w.lod_dist;
    const int nD = w.n_lod;
    const float center_dist = distance(v3(c.tile_center), cam);
    int sel = nD - 1;
    for (int l = 0; l < nD; l++)
        if (center_dist <= D[l]) { sel = l; break; }
    int status = TR_NONE;
    if (w.lod_blending) {
        const V3 lo = v3(w.tile_aabb + 6 * c.tile), hi = v3(w.tile_aabb + 6 * c.tile + 3);
        float mn = -1.0f, mx = -1.0f;
        const int npts = w.lod_bbox_check ? 8 : 1;
        for (int k = 0; k < npts; k++) {
            V3 p;
            if (w.lod_bbox_check) p = V3{(k & 4) ? hi.x : lo.x, (k & 2) ? hi.y : lo.y, (k & 1) ? hi.z : lo.z};
            else p = v3(w.tile_center + 3 * c.tile);
            V3 q; M3 tr;
            surface_mapping(w.sp, x, y, p + pos_offset, true, q, tr);
            const float d = distance(q, cam);
            if (mn < 0.0f || d < mn) mn = d;
            if (mx < 0.0f || d > mx) mx = d;
        }
        const float r = w.lod_ratio, tol = w.lod_tol;
        if (sel > 0 && mn < D[sel - 1] * (1.0f + r) + tol) status = TR_CHANGING_HIGHER;
        if (sel < nD - 1 && mx > D[sel] * (1.0f - r) - tol) status = TR_CHANGING_LOWER;
    }
    float spawning = 0.0f;
    if (w.lod_blending && w.sp.surface_type != 2) {
        const V3 cc = sp_coord_to_pos(w.sp, w.sp.center_x, w.sp.center_y);
        const float cam_u = (cam.x - cc.x) / w.sp.tile_width, cam_v = (cam.y - cc.y) / w.sp.tile_width;
        float bf = 1.0f;
        if (x == 0) bf *= 1.0f - cam_u;
        else if (x == w.sp.map_w - 1) bf *= cam_u;
        if (y == 0) bf *= 1.0f - cam_v;
        else if (y == w.sp.map_h - 1) bf *= cam_v;
        if (bf != 1.0f) { status = TR_SPAWNING; spawning = bf; }
    }
    gswt_cell_state& s = w.st[idx];
    s.lod = (uint32_t)sel; s.transition = status; s.spawning_factor = spawning;
}

// ---- edge candidates of selective_merge_edge (:827-957) and the edge orientations of the graph order (:1128-1170): one thread
// per (cell, side).  A pair of neighbours is owned by the cell visited first in the reference's scan (the lower map index).
__global__ __launch_bounds__(256) void k_w_edges(const WDev w, const V3 cam, const Mat4 vp, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                 int8_t* __restrict__ gdir)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= w.cells * 4) return;
    const int c = t >> 2, s = t & 3;
    if (s == 0 && w.merge_type == MERGE_EDGE) { w.st[c].merge = MS_NONE; w.st[c].merged_to = 0; w.head_len[c] = 0; }
    uint32_t key = kNoKey;
    int dir = 0;
    const int nbv = nbr(w, c, s);
    if (nbv >= 0 && (nbv >> 2) > c) {
        const gswt_cell& cl = w.cell[c];
        const V3 epos = v3(cl.edge_pos + 3 * s), enormal = v3(cl.edge_normal + 3 * s);
        const V3 vd = epos - cam;
        if (!is_zero(vd)) {
            const float dr = dot(enormal, vd);
            dir = dr > 0.0f ? 1 : (dr < 0.0f ? -1 : 0);
            if (w.merge_type == MERGE_EDGE) {
                const float vlen = magnitude(vd);
                const int s2 = (s + 1) & 3;
                const V3 u1 = v3(cl.corner_up + 3 * s), u2 = v3(cl.corner_up + 3 * s2);
                if (!(dot(vd, u1) > 0.0f || dot(vd, u2) > 0.0f)) {
                    const float a4[4] = {cl.corner_pos[3 * s], cl.corner_pos[3 * s + 1], cl.corner_pos[3 * s + 2], 1.0f};
                    const float b4[4] = {cl.corner_pos[3 * s2], cl.corner_pos[3 * s2 + 1], cl.corner_pos[3 * s2 + 2], 1.0f};
                    float p1[4], p2[4];
                    mat4_vec(vp.m, a4, p1);
                    mat4_vec(vp.m, b4, p2);
                    const V3 q1 = V3{p1[0], p1[1], p1[2]} / p1[3], q2 = V3{p2[0], p2[1], p2[2]} / p2[3];
                    const float clip = 1.0f;
                    const bool o1 = q1.z < -clip || q1.x < -clip || q1.x > clip || q1.y < -clip || q1.y > clip;
                    const bool o2 = q2.z < -clip || q2.x < -clip || q2.x > clip || q2.y < -clip || q2.y > clip;
                    if (!(o1 && o2)) {
                        const float dabs = fabsf(dot(enormal, vd));
                        const float ndot = dabs / vlen;
                        // the reference skips edges over the threshold without counting them towards top-k: dropping them here is the same
                        if (!(ndot > w.merge_dot_thr)) key = __float_as_uint(dabs);
                    }
                }
            }
        }
    }
    keys[t] = key;
    vals[t] = (uint32_t)t;
    gdir[t] = (int8_t)dir;
}

// ---- the sequential core of selective merging: one wave, LDS tables ---------------------------------------------------
// LDS (u16 each): mm[cells] group + 1 of a cell, nx[cells] next member + 1, seen[cells] closure stamp, then per group
// head / tail / len / offset (at most cells / 2 groups: a new group takes two fresh cells).
struct MergeLds {
    uint16_t *mm, *nx, *seen, *ghead, *gtail, *glen;
};

__device__ __forceinline__ void grp_append_cell(const MergeLds& L, int g, int c)
{
    L.nx[L.gtail[g]] = (uint16_t)(c + 1);
    L.nx[c] = 0;
    L.gtail[g] = (uint16_t)c;
    L.glen[g]++;
    L.mm[c] = (uint16_t)(g + 1);
}
__device__ __forceinline__ void grp_absorb(const MergeLds& L, int a, int b)       // members of b, in order, behind a's
{
    for (int c = L.ghead[b]; c != kNone16; c = L.nx[c] ? L.nx[c] - 1 : kNone16) L.mm[c] = (uint16_t)(a + 1);
    L.nx[L.gtail[a]] = (uint16_t)(L.ghead[b] + 1);
    L.gtail[a] = L.gtail[b];
    L.glen[a] += L.glen[b];
    L.ghead[b] = kNone16; L.glen[b] = 0;
}

__global__ __launch_bounds__(64) void k_w_merge(const WDev w, const V3 cam, const Mat4 vp, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals)
{
    extern __shared__ uint16_t lds[];
    const int cells = w.cells, lane = threadIdx.x;
    const int gcap = cells / 2 + 1;
    MergeLds L;
    L.mm = lds; L.nx = L.mm + cells; L.seen = L.nx + cells; L.ghead = L.seen + cells; L.gtail = L.ghead + gcap; L.glen = L.gtail + gcap;
    if (w.merge_type == MERGE_AXIS) {
        // selective_merge_axis, :720-825 (statuses are NOT reset between sort events: the reference keeps them until the next build)
        if (lane != 0) return;
        int cm = w.sp.half_w * w.sp.map_h + w.sp.half_h;          // coord_to_map(center_coord)
        if (w.sp.surface_type == 2) {
            float min_dist = -1.0f;
            cm = 0;
            for (int idx = 0; idx < cells; idx++) {
                if (w.st[idx].merge == MS_TO) continue;
                const V3 dv = cam - v3(w.cell[idx].tile_center);
                const float d2 = dot(dv, dv);
                if (min_dist < 0.0f || d2 < min_dist) { min_dist = d2; cm = idx; }
            }
        }
        float best = 0.0f;
        int merge_dir = -1;
        const V3 cam_dir = normalize(V3{vp.m[2], vp.m[6], vp.m[10]});
        for (int ci = 0; ci < 4; ci++) {
            const int n = nbr(w, cm, ci);
            if (n < 0) continue;
            const float dp = dot(normalize(v3(w.cell[n >> 2].tile_center) - cam), cam_dir);
            if (best < dp) { best = dp; merge_dir = ci; }
        }
        if (merge_dir < 0) return;
        const int mn[4][2] = {{3, 1}, {0, 2}, {1, 3}, {2, 0}};
        int m = cm;
        for (int i = 0; i < w.merge_dist0; i++) {
            const int n = nbr(w, m, merge_dir);
            if (n < 0) { atomicOr(&w.counts[C_ERR], ERR_AXIS_UNWRAP); return; }
            m = n >> 2;
        }
        for (int i = w.merge_dist0; i < w.merge_dist1; i++) {
            const int a = nbr(w, m, mn[merge_dir][0]), b = nbr(w, m, mn[merge_dir][1]), f = nbr(w, m, merge_dir);
            if (a < 0 || b < 0) { atomicOr(&w.counts[C_ERR], ERR_AXIS_UNWRAP); return; }
            const int ca = a >> 2, cb = b >> 2;
            if (w.st[m].merge != MS_NONE || w.st[ca].merge != MS_NONE || w.st[cb].merge != MS_NONE) break;
            const uint32_t at = w.counts[C_POOL];
            w.counts[C_POOL] = at + 3;
            w.pool[at] = (uint32_t)ca; w.pool[at + 1] = (uint32_t)m; w.pool[at + 2] = (uint32_t)cb;
            w.st[m].merge = MS_FROM; w.head_off[m] = at; w.head_len[m] = 3;
            w.st[ca].merge = MS_TO; w.st[ca].merged_to = (uint32_t)m;
            w.st[cb].merge = MS_TO; w.st[cb].merged_to = (uint32_t)m;
            if (f < 0) { atomicOr(&w.counts[C_ERR], ERR_AXIS_UNWRAP); return; }
            m = f >> 2;
        }
        return;
    }
    // ---- selective_merge_edge, :827-1027, behind the candidate sort -------------------------------------------------------
    for (int i = lane; i < 3 * cells; i += 64) lds[i] = 0;
    for (int i = lane; i < gcap; i += 64) { L.ghead[i] = kNone16; L.gtail[i] = kNone16; L.glen[i] = 0; }
    __syncthreads();
    __shared__ uint32_t s_mi[64], s_ni[64];
    __shared__ int s_ngroups, s_stop;
    if (lane == 0) { s_ngroups = 0; s_stop = 0; }
    const uint32_t n_cand = (uint32_t)cells * 4u;
    uint32_t accepted = 0;
    for (uint32_t base = 0; base < n_cand; base += 64) {
        __syncthreads();
        if (s_stop) break;
        {   // the wave fetches 64 sorted candidates and their neighbours, lane 0 consumes them in order
            const uint32_t i = base + lane;
            uint32_t mi = kNoKey, ni = 0;
            if (i < n_cand && keys[i] != kNoKey) {
                const uint32_t seq = vals[i];
                mi = seq >> 2;
                ni = (uint32_t)(nbr(w, (int)mi, (int)(seq & 3u)) >> 2);
            }
            s_mi[lane] = mi; s_ni[lane] = ni;
        }
        __syncthreads();
        if (lane == 0) {
            int ng = s_ngroups;
            for (int k = 0; k < 64; k++) {
                if (accepted >= w.merge_topk || s_mi[k] == kNoKey) { s_stop = 1; break; }
                const int mi = (int)s_mi[k], ni = (int)s_ni[k];
                const int a = (int)L.mm[mi] - 1, b = (int)L.mm[ni] - 1;
                if (a < 0 && b < 0) {
                    const int g = ng++;
                    L.ghead[g] = (uint16_t)mi; L.gtail[g] = (uint16_t)ni; L.glen[g] = 2;
                    L.nx[mi] = (uint16_t)(ni + 1); L.nx[ni] = 0;
                    L.mm[mi] = L.mm[ni] = (uint16_t)(g + 1);
                } else if (a >= 0 && b < 0) grp_append_cell(L, a, ni);
                else if (a < 0 && b >= 0) grp_append_cell(L, b, mi);
                else if (a != b) grp_absorb(L, a, b);
                accepted++;
            }
            s_ngroups = ng;
        }
    }
    __syncthreads();
    const int ngroups = s_ngroups;
    if (lane == 0) {
        // closure of non-convex groups, :959-990: a tile outside group i that is met a second time as a neighbour of i's members
        // joins i, with its own group if it has one; members appended meanwhile are visited too
        for (int i = 0; i < ngroups; i++) {
            const uint16_t stamp = (uint16_t)(i + 1);
            for (int j = L.ghead[i]; j != kNone16; j = L.nx[j] ? L.nx[j] - 1 : kNone16) {
                for (int s = 0; s < 4; s++) {
                    const int n = nbr(w, j, s);
                    if (n < 0) continue;
                    const int nidx = n >> 2;
                    if ((int)L.mm[nidx] - 1 == i) continue;
                    if (L.seen[nidx] == stamp) {
                        const int other = (int)L.mm[nidx] - 1;
                        if (other >= 0) grp_absorb(L, i, other);
                        else grp_append_cell(L, i, nidx);
                    } else L.seen[nidx] = stamp;
                }
            }
        }
        // member ranges in the pool, group order
        uint32_t at = 0;
        for (int g = 0; g < ngroups; g++) {
            // offsets can pass 65535 only with more than 65535 cells, which gswt_worker_create refuses
            L.gtail[g] = (uint16_t)at;          // tail is no longer needed: reuse as the pool offset
            at += L.glen[g];
        }
        w.counts[C_POOL] = at;
    }
    __syncthreads();
    // finalisation, :992-1026: members ascending, head = the member nearest to the camera (first minimum in ascending order)
    for (int g = lane; g < ngroups; g += 64) {
        const int len = L.glen[g];
        if (len == 0) continue;
        uint32_t* mem = w.pool + L.gtail[g];
        int n = 0;
        for (int c = L.ghead[g]; c != kNone16; c = L.nx[c] ? L.nx[c] - 1 : kNone16) {      // insertion sort into the pool
            int p = n++;
            while (p > 0 && mem[p - 1] > (uint32_t)c) { mem[p] = mem[p - 1]; p--; }
            mem[p] = (uint32_t)c;
        }
        float mind = 3.402823466e+38f;
        int mini = 0;
        for (int k = 0; k < len; k++) {
            const float d2 = distance2(v3(w.cell[mem[k]].tile_center), cam);
            if (mind > d2) { mind = d2; mini = k; }
        }
        const uint32_t head = mem[mini];
        for (int k = 0; k < len; k++) {
            gswt_cell_state& s = w.st[mem[k]];
            if (k != mini) { s.merge = MS_TO; s.merged_to = head; }
            else { s.merge = MS_FROM; s.merged_to = 0; }
        }
        w.head_off[head] = (uint32_t)L.gtail[g];
        w.head_len[head] = (uint32_t)len;
    }
}

// ---- Distance / Viewport order keys, :1029-1060: descending key, ties in descending map index = ascending stable sort reversed
__global__ __launch_bounds__(256) void k_w_order_keys(const WDev w, const V3 cam, const Mat4 vp, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= w.cells) return;
    uint32_t key = kNoKey;
    if (w.st[idx].merge != MS_TO) {
        const V3 p = v3(w.cell[idx].tile_center);
        float k = w.sort_type == SORT_DISTANCE ? distance2(cam, p) : (vp.m[2] * p.x + vp.m[6] * p.y) + vp.m[10] * p.z;
        k = k + 0.0f;                                    // -0 and +0 compare equal in the reference's partial_cmp
        const uint32_t u = __float_as_uint(k);
        key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        atomicAdd(&w.counts[C_N_NODES], 1u);
    }
    keys[idx] = key;
    vals[idx] = (uint32_t)idx;
}
__global__ __launch_bounds__(256) void k_w_order_reverse(const WDev w, const uint32_t* __restrict__ sorted_vals, uint32_t* __restrict__ order)
{
    const uint32_t n = w.counts[C_N_NODES];
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k == 0) w.counts[C_N_ORDER] = n;
    if (k < n) order[k] = sorted_vals[n - 1u - k];
}

// ---- Object (breadth-first, :1062-1126) and Graph (:1128-1218) orders: one workgroup, the traversals on thread 0 ----------
// Tables carved from one block: LDS when they fit (kLds: the pointers then stay in the LDS address space), global scratch
// otherwise.  An edge is ONE 64-bit record (src | dst << 16 | next-out << 32 | next-in << 48) and a node's list heads one
// 32-bit word, so a traversal step costs one record read plus one flag read instead of four dependent u16 reads.
constexpr int kSeqThreads = 1024;
struct GraphTabs {
    unsigned long long* erec;      // 2 x cells
    uint32_t* nrec;                // cells: ohead | ihead << 16
    uint16_t *node_of_cell, *weight, *row_of_index, *index_of_row, *deg, *pair_eid, *stack, *finish, *removed, *pos;
    uint16_t* flags;               // per node: bit 0 discovered, bit 1 finished (one read for both)
};
__host__ __device__ inline size_t graph_tab_bytes(int cells)
{
    // erec 16 B, nrec 4 B, five node tables 10 B, pair ids 8 B, stack 6 B, finish + removed + pos 6 B, flags 2 B per cell
    return (size_t)cells * (16 + 4 + 10 + 8 + 6 + 6 + 2) + 64;
}
template <typename P>
__device__ __forceinline__ GraphTabs graph_tabs(P* tab, int cells)
{
    GraphTabs T;
    T.erec = reinterpret_cast<unsigned long long*>(tab);
    T.nrec = reinterpret_cast<uint32_t*>(T.erec + 2 * cells);
    uint16_t* p = reinterpret_cast<uint16_t*>(T.nrec + cells);
    T.node_of_cell = p; p += cells; T.weight = p; p += cells; T.row_of_index = p; p += cells; T.index_of_row = p; p += cells; T.deg = p; p += cells;
    T.pair_eid = p; p += 4 * cells; T.stack = p; p += 3 * cells; T.finish = p; p += cells; T.removed = p; p += cells; T.pos = p; p += cells;
    T.flags = p;
    return T;
}
__device__ __forceinline__ uint32_t e_src(unsigned long long r) { return (uint32_t)r & 0xFFFFu; }
__device__ __forceinline__ uint32_t e_dst(unsigned long long r) { return (uint32_t)(r >> 16) & 0xFFFFu; }
__device__ __forceinline__ uint32_t e_onext(unsigned long long r) { return (uint32_t)(r >> 32) & 0xFFFFu; }
__device__ __forceinline__ uint32_t e_inext(unsigned long long r) { return (uint32_t)(r >> 48); }

// petgraph::algo::toposort over the live part of the graph (rows are node identities, indices are petgraph's NodeIndex values,
// which swap_remove renumbers), in its two halves.  Pass 1: depth-first finish order from the roots n-1..0, reversed ->
// T.finish, returns its length.  Pass 2 (petgraph's cycle check: a reverse-graph traversal per node in that order that must
// not reach a second node) returns the row petgraph reports, or -1.  Pass 2 can only find something if some live edge points
// backwards in T.finish, which the workgroup tests in parallel first: on an acyclic graph the second traversal is skipped.
__device__ __forceinline__ int toposort_pass1(const GraphTabs& T, int n_index)       // T.flags of every live row cleared by the caller
{
    int nfin = 0, sp = 0;
    for (int i = n_index - 1; i >= 0; i--) {
        const int root = T.row_of_index[i];
        if (T.flags[root] & 1u) continue;
        T.stack[sp++] = (uint16_t)root;
        int top = root;
        bool top_known = true;                       // the element just pushed is the top: no need to read it back
        while (sp > 0) {
            const int nx = top_known ? top : (int)T.stack[sp - 1];
            const uint32_t f = T.flags[nx];
            if (!(f & 1u)) {
                T.flags[nx] = (uint16_t)(f | 1u);
                uint32_t e = T.nrec[nx] & 0xFFFFu;
                top = nx; top_known = true;
                if (e != kNone16) {
                    unsigned long long rec = T.erec[e];
                    for (;;) {
                        // the successor's flag and the next record are fetched together (both addresses come from `rec`), so a
                        // list step costs one LDS round trip instead of two; dead records (src = none) fetch harmless addresses
                        const uint32_t next_e = e_onext(rec);
                        const bool live = e_src(rec) != kNone16;
                        const int succ = (int)e_dst(rec);             // (never nx itself: an edge joins two different nodes)
                        const uint32_t fl = T.flags[live ? succ : nx];
                        const unsigned long long rec_n = T.erec[next_e != kNone16 ? next_e : e];
                        if (live && !(fl & 1u)) { T.stack[sp++] = (uint16_t)succ; top = succ; }
                        if (next_e == kNone16) break;
                        e = next_e; rec = rec_n;
                    }
                }
            } else {
                sp--;
                if (!(f & 2u)) { T.flags[nx] = (uint16_t)(f | 2u); T.finish[nfin++] = (uint16_t)nx; }
                top_known = false;
            }
        }
    }
    for (int a = 0, b = nfin - 1; a < b; a++, b--) { const uint16_t t = T.finish[a]; T.finish[a] = T.finish[b]; T.finish[b] = t; }
    return nfin;
}
__device__ __forceinline__ int toposort_pass2(const GraphTabs& T, int n_index, int nfin)
{
    for (int i = 0; i < n_index; i++) T.flags[T.row_of_index[i]] = 0;
    for (int k = 0; k < nfin; k++) {
        int sp = 0;
        T.stack[sp++] = T.finish[k];
        bool cycle = false;
        while (sp > 0) {
            const int node = T.stack[--sp];
            if (T.flags[node]) continue;
            T.flags[node] = 1;
            for (uint32_t e = T.nrec[node] >> 16; e != kNone16;) {
                const unsigned long long r = T.erec[e];
                e = e_inext(r);
                if (e_src(r) == kNone16) continue;
                const int pred = (int)e_src(r);
                if (!T.flags[pred]) T.stack[sp++] = (uint16_t)pred;
            }
            if (cycle) return node;
            cycle = true;
        }
    }
    return -1;
}

// exclusive prefix of a 0/1 flag over the workgroup's threads (ballot + popcount per wave, wave totals through LDS); returns the
// prefix and adds the total to `carry` (read it before the next call: two barriers inside)
__device__ __forceinline__ uint32_t flag_scan(bool flag, uint32_t* s_wave, uint32_t& carry)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const uint32_t in_wave = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) s_wave[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t k = 0; k < kSeqThreads / 64; k++) { const uint32_t c = s_wave[k]; total += c; if (k < wv) before += c; }
    __syncthreads();
    const uint32_t r = carry + before + in_wave;
    carry += total;
    return r;
}

template <bool kLds>
__global__ __launch_bounds__(kSeqThreads) void k_w_order_seq(const WDev w, const V3 cam, const int8_t* __restrict__ gdir, unsigned long long* __restrict__ gtab,
                                                             uint32_t* __restrict__ order)
{
    extern __shared__ unsigned long long lds64[];
    const int cells = w.cells, tid = threadIdx.x;
    const GraphTabs T = kLds ? graph_tabs(lds64, cells) : graph_tabs(gtab, cells);
    __shared__ uint32_t s_wave[kSeqThreads / 64];
    if (w.sort_type == SORT_OBJECT) {
        // start: the not-MergedTo cell nearest to the camera, first minimum in index order
        __shared__ float s_d[kSeqThreads];
        __shared__ int s_i[kSeqThreads];
        float bd = -1.0f; int bi = 0;
        for (int idx = tid; idx < cells; idx += kSeqThreads) {
            if (w.st[idx].merge == MS_TO) continue;
            const float d = distance2(cam, v3(w.cell[idx].tile_center));
            if (bd < 0.0f || d < bd) { bd = d; bi = idx; }
        }
        s_d[tid] = bd; s_i[tid] = bi;
        for (int idx = tid; idx < cells; idx += kSeqThreads) T.flags[idx] = 0;
        __syncthreads();
        if (tid != 0) return;
        bd = -1.0f; bi = 0;
        for (int k = 0; k < kSeqThreads; k++)     // a thread's own candidates ascend in index: (d, index) order keeps the first minimum overall
            if (s_d[k] >= 0.0f && (bd < 0.0f || s_d[k] < bd || (s_d[k] == bd && s_i[k] < bi))) { bd = s_d[k]; bi = s_i[k]; }
        uint16_t* queue = T.stack;                 // 3 x cells entries: one per cell is enough
        int qh = 0, qt = 0, n = 0;
        queue[qt++] = (uint16_t)bi;
        T.flags[bi] = 1;
        while (qh < qt) {
            const int c = queue[qh++];
            T.finish[n++] = (uint16_t)c;
            for (int s = 0; s < 4; s++) {
                const int nb = nbr(w, c, s);
                if (nb >= 0 && !T.flags[nb >> 2]) { queue[qt++] = (uint16_t)(nb >> 2); T.flags[nb >> 2] = 1; }
            }
        }
        for (int k = 0; k < n; k++) order[k] = T.finish[n - 1 - k];
        w.counts[C_N_ORDER] = (uint32_t)n;
        return;
    }
    // ---- Graph ------------------------------------------------------------------------------------------------------------
    // nodes: not-MergedTo cells in index order (add_node in the scan of :1136-1147)
    uint32_t carry = 0;
    for (int base = 0; base < cells; base += kSeqThreads) {
        const int idx = base + tid;
        const bool flag = idx < cells && w.st[idx].merge != MS_TO;
        const uint32_t r = flag_scan(flag, s_wave, carry);
        if (idx < cells) {
            T.node_of_cell[idx] = flag ? (uint16_t)r : kNone16;
            if (flag) { T.weight[r] = (uint16_t)idx; T.row_of_index[r] = (uint16_t)r; T.index_of_row[r] = (uint16_t)r; T.nrec[r] = 0xFFFFFFFFu; T.deg[r] = 0; }
        }
    }
    const int n_nodes = (int)carry;
    __syncthreads();
    // edges: one per owned pair of different nodes with a nonzero orientation, numbered in scan order (cell, side)
    carry = 0;
    auto node_of = [&](int c) -> int { return w.st[c].merge == MS_TO ? T.node_of_cell[w.st[c].merged_to] : T.node_of_cell[c]; };
    for (int base = 0; base < cells * 4; base += kSeqThreads) {
        const int t = base + tid;
        bool flag = false;
        int a = 0, b = 0;
        if (t < cells * 4) {
            const int c = t >> 2, s = t & 3;
            const int dir = gdir[t];                  // nonzero only on the owner's side of a pair (k_w_edges)
            if (dir != 0) {
                const int nb = nbr(w, c, s);
                const int tn = node_of(c), nn = node_of(nb >> 2);
                if (tn != nn) { flag = true; a = dir > 0 ? tn : nn; b = dir > 0 ? nn : tn; }
            }
        }
        const uint32_t e = flag_scan(flag, s_wave, carry);
        if (t < cells * 4) {
            T.pair_eid[t] = flag ? (uint16_t)e : kNone16;
            if (flag) T.erec[e] = (unsigned long long)(uint32_t)a | ((unsigned long long)(uint32_t)b << 16) | (0xFFFFFFFFull << 32);
        }
    }
    __syncthreads();
    // adjacency lists, newest edge first (petgraph links a new edge at the head of both lists): a node's incident edges are the
    // pairs of its cells' sides; linking them in ascending id leaves the newest at the head
    for (int r = tid; r < n_nodes; r += kSeqThreads) {
        const int head = T.weight[r];
        const bool grp = w.st[head].merge == MS_FROM;
        const int len = grp ? (int)w.head_len[head] : 1;
        const uint32_t* mem = w.pool + (grp ? w.head_off[head] : 0u);
        uint32_t oh = kNone16, ih = kNone16;
        int last = -1, deg = 0;
        for (;;) {                                    // selection in ascending edge id: degrees are small (4 per member cell)
            int best = 0x7FFFFFFF;
            for (int m = 0; m < len; m++) {
                const int c = grp ? (int)mem[m] : head;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const int nb = nbr(w, c, s);
                    if (nb < 0) continue;
                    const int o = nb >> 2;
                    const int pe = o > c ? T.pair_eid[c * 4 + s] : T.pair_eid[o * 4 + (nb & 3)];
                    if (pe != kNone16 && pe > last && pe < best) best = pe;
                }
            }
            if (best == 0x7FFFFFFF) break;
            // an edge inside this node does not exist (tn == nn above); an edge to another node is seen from exactly one cell side
            // here, and multi-edges to the same neighbour node have ids of their own
            // the record's two link fields are written by the two end nodes' threads: 16-bit stores, so neither clobbers the other
            uint16_t* half = reinterpret_cast<uint16_t*>(&T.erec[best]);
            if (half[0] == (uint16_t)r) { half[2] = (uint16_t)oh; oh = (uint32_t)best; }       // this node is the source: out-list
            else { half[3] = (uint16_t)ih; ih = (uint32_t)best; }
            deg++;
            last = best;
        }
        T.nrec[r] = oh | (ih << 16);
        T.deg[r] = (uint16_t)deg;
    }
    __syncthreads();
    const int n_edges = (int)carry;
    __shared__ int s_nfin, s_cyc;
    int n_index = n_nodes, n_removed = 0;
#ifdef GSWT_W_SKIP_DFS                                        // timing ablation (tools only): everything but the traversals
    if (tid == 0) w.counts[C_N_ORDER] = 0;
    return;
#endif
    for (;;) {                                              // uniform over the workgroup
        for (int k = tid; k < n_nodes; k += kSeqThreads) T.flags[k] = 0;
        __syncthreads();
        if (tid == 0) s_nfin = toposort_pass1(T, n_index);
        __syncthreads();
        const int nfin = s_nfin;
        for (int k = tid; k < nfin; k += kSeqThreads) T.pos[T.finish[k]] = (uint16_t)k;
        __syncthreads();
        int back = 0;
        for (int e = tid; e < n_edges; e += kSeqThreads) {
            const unsigned long long rec = T.erec[e];
            if (e_src(rec) != kNone16 && T.pos[e_src(rec)] >= T.pos[e_dst(rec)]) back = 1;
        }
        const int cyclic = __syncthreads_or(back);
        if (tid == 0) s_cyc = cyclic ? toposort_pass2(T, n_index, nfin) : -1;
        __syncthreads();
        const int r = s_cyc;
        if (r < 0) break;
        // a node on a cycle is taken out (Graph::remove_node: its edges go, the last node takes its index) and listed behind the rest
        if (tid == 0) {
            T.removed[n_removed] = T.weight[r];
            for (int pass = 0; pass < 2; pass++)
                for (uint32_t e = pass ? T.nrec[r] >> 16 : T.nrec[r] & 0xFFFFu; e != kNone16;) {
                    const unsigned long long rec = T.erec[e];
                    const uint32_t cur = e;
                    e = pass ? e_inext(rec) : e_onext(rec);
                    if (e_src(rec) == kNone16) continue;
                    const int other = pass ? (int)e_src(rec) : (int)e_dst(rec);
                    T.deg[other]--;
                    T.deg[r]--;
                    T.erec[cur] = rec | 0xFFFFull;
                }
            const int idx = T.index_of_row[r], last = n_index - 1;
            if (idx != last) { const int lr = T.row_of_index[last]; T.row_of_index[idx] = (uint16_t)lr; T.index_of_row[lr] = (uint16_t)idx; }
        }
        n_removed++;
        n_index--;
        __syncthreads();
    }
    if (tid != 0) return;
    int n_out = 0;
    for (int k = 0; k < s_nfin; k++) {
        const int r = T.finish[k];
        if (T.deg[r] != 0) order[n_out++] = T.weight[r];
    }
    for (int k = 0; k < n_removed; k++) order[n_out++] = T.removed[k];
    for (int a = 0, b = n_out - 1; a < b; a++, b--) { const uint32_t t = order[a]; order[a] = order[b]; order[b] = t; }
    w.counts[C_N_ORDER] = (uint32_t)n_out;
}

// ---- SortData records ---------------------------------------------------------------------------------------------------
// exclusive prefixes over the ordered tiles: group index, first member, offset in the concatenated merged lists
__global__ __launch_bounds__(1024) void k_w_scan(const WDev w, const uint32_t* __restrict__ order, uint32_t* __restrict__ pre_group, uint32_t* __restrict__ pre_member,
                                                 uint32_t* __restrict__ pre_len, uint32_t* __restrict__ ent_len)
{
    __shared__ uint32_t s_a[1024], s_b[1024], s_c[1024];
    __shared__ uint32_t s_ca, s_cb;
    __shared__ unsigned long long s_cc;
    const int tid = threadIdx.x;
    const uint32_t n = w.counts[C_N_ORDER];
    if (tid == 0) { s_ca = 0; s_cb = 0; s_cc = 0; }
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t k = base + tid;
        uint32_t fa = 0, fb = 0, fc = 0;
        if (k < n) {
            const uint32_t mi = order[k];
            if (w.st[mi].merge == MS_FROM) {
                fa = 1; fb = w.head_len[mi];
                const uint32_t* mem = w.pool + w.head_off[mi];
                for (uint32_t m = 0; m < fb; m++) {
                    const gswt_cell_state& ms = w.st[mem[m]];
                    const uint32_t tile = w.cell[mem[m]].tile;
                    fc += w.splat_count[ms.lod * w.n_tile + tile];
                    if (ms.transition == TR_CHANGING_LOWER) fc += w.splat_count[(ms.lod + 1) * w.n_tile + tile];
                    else if (ms.transition == TR_CHANGING_HIGHER && ms.lod > 0) fc += w.splat_count[(ms.lod - 1) * w.n_tile + tile];
                }
            }
        }
        s_a[tid] = fa; s_b[tid] = fb; s_c[tid] = fc;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const uint32_t va = tid >= o ? s_a[tid - o] : 0u, vb = tid >= o ? s_b[tid - o] : 0u, vc = tid >= o ? s_c[tid - o] : 0u;
            __syncthreads();
            s_a[tid] += va; s_b[tid] += vb; s_c[tid] += vc;
            __syncthreads();
        }
        if (k < n) {
            pre_group[k] = s_ca + s_a[tid] - fa;
            pre_member[k] = s_cb + s_b[tid] - fb;
            pre_len[k] = (uint32_t)(s_cc + s_c[tid] - fc);       // merged_offset is a u32 in the reference's record too
            ent_len[k] = fc;
        }
        __syncthreads();
        if (tid == 1023) { s_ca += s_a[1023]; s_cb += s_b[1023]; s_cc += s_c[1023]; }
        __syncthreads();
    }
    if (tid == 0) {
        w.counts[C_N_GROUPS] = s_ca; w.counts[C_N_MEMBERS] = s_cb;
        w.counts[C_MERGED_LO] = (uint32_t)s_cc; w.counts[C_MERGED_HI] = (uint32_t)(s_cc >> 32);
    }
}

__device__ inline uint32_t choose_presort_view(const WDev& w, const M3& transform, V3 pos, V3 cam)      // :700-716
{
    const V3 dl = transform * normalize(pos - cam);
    uint32_t best = 0;
    float best_err = 1000.0f;
    for (int i = 0; i < w.n_view; i++) {
        const V3 pd = v3(w.presort_dirs + 3 * i);
        const float ex = dl.x - pd.x, ey = dl.y - pd.y, ez = dl.z - pd.z;
        const float err = (ex * ex + ey * ey) + ez * ez;
        if (err < best_err) { best = (uint32_t)i; best_err = err; }
    }
    return best;
}

// one thread per ordered tile: the record (:500-690), its group description, and the draw the render loop makes of it
// (renderer.rs:466-591 host half)
__global__ __launch_bounds__(256) void k_w_records(const WDev w, const V3 cam, const uint32_t* __restrict__ order, const uint32_t* __restrict__ pre_group,
                                                   const uint32_t* __restrict__ pre_member, const uint32_t* __restrict__ pre_len, const uint32_t* __restrict__ ent_len,
                                                   gswt_sorted_tile* __restrict__ tiles, gswt_merge_group* __restrict__ groups, gswt_merge_member* __restrict__ members,
                                                   gswt_draw* __restrict__ draws)
{
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k >= w.counts[C_N_ORDER]) return;
    const uint32_t mi = order[k];
    const gswt_cell& c = w.cell[mi];
    const gswt_cell_state& s = w.st[mi];
    const int mx = (int)mi / w.sp.map_h, my = (int)mi - mx * w.sp.map_h;
    uint32_t view_id, key_len = 1;
    bool do_transition = false;
    if (s.merge == MS_FROM) {
        const uint32_t len = w.head_len[mi];
        const uint32_t* mem = w.pool + w.head_off[mi];
        bool merge_x = true, merge_y = true;
        V3 avg_c;
        Quat avg_q;
        gswt_merge_member* mo = members + pre_member[k];
        for (uint32_t m = 0; m < len; m++) {
            const uint32_t m_mi = mem[m];
            const int ax = (int)m_mi / w.sp.map_h, ay = (int)m_mi - ax * w.sp.map_h;
            if (ax != mx) merge_x = false;
            if (ay != my) merge_y = false;
            const gswt_cell& mc = w.cell[m_mi];
            const gswt_cell_state& ms = w.st[m_mi];
            avg_c = avg_c + v3(mc.tile_center);
            M3 tl;
            for (int i = 0; i < 9; i++) tl.m[i] = mc.to_local[i];
            const Quat q = quat_from_mat3(tl);
            avg_q.s += q.s; avg_q.x += q.x; avg_q.y += q.y; avg_q.z += q.z;
            gswt_merge_member mm;
            mm.map_index = m_mi; mm.lod = ms.lod; mm.tile = mc.tile; mm.other_lod = -1;
            if (ms.transition == TR_CHANGING_LOWER) mm.other_lod = (int32_t)ms.lod + 1;
            else if (ms.transition == TR_CHANGING_HIGHER) mm.other_lod = (int32_t)ms.lod - 1;
            if (ms.transition != TR_NONE) do_transition = true;
            mo[m] = mm;
        }
        if (!merge_x && !merge_y) view_id = (uint32_t)w.n_view - 1u;
        else {
            const float n = (float)len;
            const Quat q{avg_q.s / n, avg_q.x / n, avg_q.y / n, avg_q.z / n};
            view_id = choose_presort_view(w, mat3_from_quat(q), avg_c / n, cam);
        }
        key_len = len;
        gswt_merge_group g;
        g.view_id = view_id; g.first_member = pre_member[k]; g.n_members = len; g._pad = 0;
        groups[pre_group[k]] = g;
    } else {
        M3 tl;
        for (int i = 0; i < 9; i++) tl.m[i] = c.to_local[i];
        view_id = choose_presort_view(w, tl, v3(c.tile_center), cam);
    }
    gswt_sorted_tile st;
    memset(&st, 0, sizeof(st));
    st.lod = s.lod; st.tile = c.tile; st.view_id = view_id;
    for (int i = 0; i < 3; i++) { st.tile_offset[i] = c.tile_offset[i]; st.tile_center[i] = c.tile_center[i]; }
    st.map_index = mi; st.map_coord[0] = (uint32_t)mx; st.map_coord[1] = (uint32_t)my;
    st.transition = s.transition; st.spawning_factor = s.spawning_factor;
    st.has_corners = c.has_corner;
    for (int i = 0; i < 12; i++) st.corners[i] = c.has_corner ? c.corner_pos[i] : 0.0f;
    st.key_len = key_len;
    st.single_lod_id = -1;
    if (s.merge == MS_FROM) {
        st.merged = 1; st.merged_group = pre_group[k]; st.merged_offset = pre_len[k]; st.merged_count = ent_len[k];
        st.single_lod_id = do_transition ? -1 : (int32_t)s.lod;
    }
    tiles[k] = st;
    // the draw, renderer.rs:499-590
    gswt_draw d;
    memset(&d, 0, sizeof(d));
    gswt_tile_uniforms& u = d.tile;
    u.single_draw = 0; u.map_index = mi; u.single_lod_id = -1; u.valid_lod_id = -1; u.changing = 0; u.changing_to_lower = -1;
    u.tile_id[0] = s.lod; u.tile_id[1] = c.tile; u.tile_id[2] = view_id; u.tile_id[3] = 0;
    u.offset[0] = c.tile_offset[0]; u.offset[1] = c.tile_offset[1]; u.offset[2] = c.tile_offset[2]; u.offset[3] = 0.0f;
    u.map_coord[0] = (uint32_t)mx; u.map_coord[1] = (uint32_t)my;
    d.lod = s.lod;
    if (st.merged) {
        u.single_draw = 1;
        u.single_lod_id = st.single_lod_id;
        u.changing = st.single_lod_id == -1 ? 1u : 0u;
        d.merged = 1; d.merged_offset = st.merged_offset; d.merged_count = st.merged_count; d.merged_group = st.merged_group;
        d.merged_has_lod = st.single_lod_id == -1 ? 1u : 0u;
    } else {
        d.base_lod = s.lod;
        if (s.transition == TR_CHANGING_LOWER) { u.changing = 1; u.changing_to_lower = 1; }
        else if (s.transition == TR_CHANGING_HIGHER) {
            u.changing = 1; u.changing_to_lower = 0;
            if (s.lod == 0) atomicOr(&w.counts[C_ERR], ERR_LOD0_HIGHER);
            d.base_lod = s.lod - 1;
        } else u.valid_lod_id = (int32_t)s.lod;
        d.base_tile = c.tile; d.base_view = view_id;
    }
    if (key_len == 1) {
        if (!c.has_corner) atomicOr(&w.counts[C_ERR], ERR_NO_CORNERS);
        d.cull_enable = 1;
        for (int i = 0; i < 12; i++) d.corners[i] = st.corners[i];
    }
    draws[k] = d;
}

template <typename T>
struct Dev {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) { release(); n = count; return hipMalloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T)); }
    void release() { if (p) hipFree(p); p = nullptr; n = 0; }
};
template <typename T>
struct Pinned {
    T* p = nullptr;
    hipError_t alloc(size_t count) { release(); return hipHostMalloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T), hipHostMallocDefault); }
    void release() { if (p) hipHostFree(p); p = nullptr; }
};

}  // namespace

struct gswt_worker {
    gswt_ctx* ctx = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    WDev dev{};
    int cells = 0;
    bool have_cells = false, have_lod = false, have_sort = false;
    size_t lds_merge = 0, lds_graph = 0;
    int graph_in_lds = 0;
    Dev<float> height_map, lod_dist, tile_center, tile_aabb, presort_dirs;
    Dev<uint32_t> splat_count, head_off, head_len, pool, counts, order, keys_a, vals_a, keys_b, vals_b, radix_ws, pre_group, pre_member, pre_len, ent_len;
    Dev<int32_t> nb;
    Dev<gswt_cell> cell;
    Dev<gswt_cell_state> st;
    Dev<int8_t> gdir;
    Dev<unsigned long long> gtab;
    Dev<unsigned long long> n64;
    Dev<gswt_sorted_tile> tiles;
    Dev<gswt_merge_group> groups;
    Dev<gswt_merge_member> members;
    Dev<gswt_draw> draws;
    // Results in host memory, three sets: one published (what gswt_set_draws_from_worker takes), one possibly in use by the render
    // thread, one being written by gswt_worker_fetch -- the worker thread never waits for the render thread and vice versa.
    struct HostSet {
        Pinned<gswt_sorted_tile> tiles;
        Pinned<gswt_merge_group> groups;
        Pinned<gswt_merge_member> members;
        Pinned<gswt_draw> draws;
        uint32_t counts[C_COUNT] = {};
    } hs[3];
    std::mutex mu;
    int published = -1, in_use = -1;
    bool sort_pending = false;
    Pinned<uint32_t> h_counts;
    size_t radix_words = 0;
};

namespace {

int wfail(gswt_worker* w, int code, const char* what, hipError_t e = hipSuccess)
{
    char buf[384];
    if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof(buf), "%s", what);
    w->err = buf;
    return code;
}

#define WHIP(call)                                                                  \
    do {                                                                            \
        const hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) return wfail(w, GSWT_ERR_HIP, #call, e_);             \
    } while (0)

template <typename T>
int upload(gswt_worker* w, Dev<T>& d, const T* src, size_t n)
{
    WHIP(d.alloc(n));
    if (n) { WHIP(hipMemcpy(d.p, src, n * sizeof(T), hipMemcpyHostToDevice)); WHIP(hipStreamSynchronize(nullptr)); }      // (the worker's stream is non-blocking: not ordered behind the null stream)
    return GSWT_OK;
}

Mat4 mat4_of(const float* vp) { Mat4 m; memcpy(m.m, vp, sizeof(m.m)); return m; }

int read_counts(gswt_worker* w)
{
    WHIP(hipMemcpyAsync(w->h_counts.p, w->counts.p, C_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, w->stream));
    WHIP(hipStreamSynchronize(w->stream));
    const uint32_t err = w->h_counts.p[C_ERR];
    if (err & ERR_LOD0_HIGHER) return wfail(w, GSWT_ERR_BAD_ARG, "render: Changing(false) on lod 0 (index underflow in the reference)");
    if (err & ERR_NO_CORNERS) return wfail(w, GSWT_ERR_STATE, "render: corner_data is None (called `Option::unwrap()` on a `None` value, renderer.rs:476)");
    if (err & ERR_AXIS_UNWRAP) return wfail(w, GSWT_ERR_BAD_ARG, "selective_merge_axis: the map is too small for merge_tile_dist (unwrap on a missing neighbour, wangtile.rs:770-780)");
    return GSWT_OK;
}

}  // namespace

// No C++ exception may unwind through the C ABI (std::string / std::mutex / std::lock_guard can throw, and these entry points are
// called from a second host thread): every int-returning entry point below is a function-try-block.
#define GSWT_WCATCH                                             \
    catch (const std::bad_alloc&) { return GSWT_ERR_CAPACITY; } \
    catch (...) { return GSWT_ERR_HIP; }

extern "C" {

int gswt_worker_create(gswt_ctx* ctx, const gswt_worker_config* cfg, gswt_worker** out)
try {
    if (!ctx || !cfg || !out) return GSWT_ERR_BAD_ARG;
    *out = nullptr;
    gswt_worker* w = new (std::nothrow) gswt_worker();
    if (!w) return GSWT_ERR_CAPACITY;
    // (an exception thrown below -- std::string, std::mutex, allocation -- unwinds into GSWT_WCATCH: the guard destroys the half-built worker)
    struct Guard { gswt_worker* w; ~Guard() { if (w) gswt_worker_destroy(w); } } guard{w};
    w->ctx = ctx;
    w->device = gswt::ctx_device(ctx);
    auto bail = [&](int code) { guard.w = nullptr; gswt_worker_destroy(w); return code; };
    const size_t cells = (size_t)cfg->map_w * cfg->map_h;
    if (cells == 0 || cfg->n_lod == 0 || cfg->n_lod > 16 || cfg->n_tile == 0 || cfg->n_view == 0 || !cfg->lod_transition_dist || !cfg->tile_center || !cfg->tile_aabb ||
        !cfg->splat_count || !cfg->presort_dirs || !cfg->neighbors || cfg->surface_type > 2 || cfg->tile_sort_type > 3 || cfg->merge_type > 2 ||
        (cfg->surface_type == 1 && (!cfg->height_map || cfg->hm_w == 0 || cfg->hm_h == 0)) || !(cfg->tile_width > 0.0f))
        return bail(GSWT_ERR_BAD_ARG);
    // u16 tables (map indices, node and edge ids up to 2 x cells) and an edge-merge LDS block of 9 bytes per cell
    if (cells > 16000 * 2 || 2 * cells >= 65535) return bail(GSWT_ERR_CAPACITY);
    if (hipSetDevice(w->device) != hipSuccess) return bail(GSWT_ERR_HIP);
    if (hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking) != hipSuccess) return bail(GSWT_ERR_HIP);
    w->cells = (int)cells;
    WDev& d = w->dev;
    d.sp.map_w = (int)cfg->map_w; d.sp.map_h = (int)cfg->map_h; d.sp.half_w = (int)cfg->half_w; d.sp.half_h = (int)cfg->half_h;
    d.sp.tile_width = cfg->tile_width; d.sp.surface_type = (int)cfg->surface_type;
    for (int k = 0; k < 3; k++) d.sp.hm_scale[k] = cfg->height_map_scale[k];
    d.sp.sphere_radius = cfg->sphere_radius;
    d.sp.hm_w = (int)cfg->hm_w; d.sp.hm_h = (int)cfg->hm_h;
    d.cells = (int)cells; d.n_lod = (int)cfg->n_lod; d.n_tile = (int)cfg->n_tile; d.n_view = (int)cfg->n_view;
    d.sort_type = (int)cfg->tile_sort_type; d.merge_type = (int)cfg->merge_type;
    d.lod_blending = (int)cfg->lod_blending; d.lod_bbox_check = (int)cfg->lod_bbox_check;
    d.lod_ratio = cfg->lod_transition_width_ratio; d.lod_tol = cfg->lod_dist_tolerance;
    d.merge_dist0 = cfg->merge_tile_dist[0]; d.merge_dist1 = cfg->merge_tile_dist[1];
    d.merge_dot_thr = cfg->merge_dot_threshold; d.merge_topk = cfg->merge_topk;
    int rc;
#define WTRY(x) do { rc = (x); if (rc != GSWT_OK) return bail(rc); } while (0)
    if (cfg->height_map && cfg->hm_w && cfg->hm_h) WTRY(upload(w, w->height_map, cfg->height_map, (size_t)cfg->hm_w * cfg->hm_h));
    WTRY(upload(w, w->lod_dist, cfg->lod_transition_dist, cfg->n_lod));
    WTRY(upload(w, w->tile_center, cfg->tile_center, (size_t)cfg->n_tile * 3));
    WTRY(upload(w, w->tile_aabb, cfg->tile_aabb, (size_t)cfg->n_tile * 6));
    WTRY(upload(w, w->presort_dirs, cfg->presort_dirs, (size_t)cfg->n_view * 3));
    WTRY(upload(w, w->splat_count, cfg->splat_count, (size_t)cfg->n_lod * cfg->n_tile));
    WTRY(upload(w, w->nb, cfg->neighbors, cells * 4));
    const uint32_t n_cand = (uint32_t)cells * 4u;
    w->radix_words = gswt::radix_ws_words(n_cand, 32);
    hipError_t e = hipSuccess;
    auto A = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    A(w->cell.alloc(cells)); A(w->st.alloc(cells)); A(w->head_off.alloc(cells)); A(w->head_len.alloc(cells)); A(w->pool.alloc(cells + 8));
    A(w->counts.alloc(C_COUNT)); A(w->order.alloc(cells)); A(w->keys_a.alloc(n_cand)); A(w->vals_a.alloc(n_cand)); A(w->keys_b.alloc(n_cand));
    A(w->vals_b.alloc(n_cand)); A(w->radix_ws.alloc(w->radix_words)); A(w->pre_group.alloc(cells)); A(w->pre_member.alloc(cells)); A(w->pre_len.alloc(cells));
    A(w->ent_len.alloc(cells)); A(w->gdir.alloc(n_cand)); A(w->gtab.alloc(graph_tab_bytes((int)cells) / 8 + 8)); A(w->n64.alloc(8));
    A(w->tiles.alloc(cells)); A(w->groups.alloc(cells)); A(w->members.alloc(cells)); A(w->draws.alloc(cells));
    for (auto& h : w->hs) { A(h.tiles.alloc(cells)); A(h.groups.alloc(cells)); A(h.members.alloc(cells)); A(h.draws.alloc(cells)); }
    A(w->h_counts.alloc(C_COUNT));
    if (e != hipSuccess) return bail(GSWT_ERR_HIP);
    // (stream-ordered: the worker's stream is non-blocking and does not wait for null-stream memsets)
    A(hipMemsetAsync(w->st.p, 0, cells * sizeof(gswt_cell_state), w->stream)); A(hipMemsetAsync(w->head_len.p, 0, cells * 4, w->stream));
    A(hipMemsetAsync(w->counts.p, 0, C_COUNT * 4, w->stream));
    {
        // item counts of the two sorts a sort event may run (edge candidates, cells): fixed for the worker's life, so they are
        // written here, once, from pinned memory and before the synchronize -- not per event from a stack array that an
        // asynchronous copy could outlive
        Pinned<unsigned long long> h;
        A(h.alloc(8));
        if (e == hipSuccess) {
            for (int k = 0; k < 8; k++) h.p[k] = 0ull;
            h.p[0] = n_cand; h.p[4] = cells;
            A(hipMemcpyAsync(w->n64.p, h.p, 8 * sizeof(unsigned long long), hipMemcpyHostToDevice, w->stream));
            A(hipStreamSynchronize(w->stream));
        }
        h.release();
    }
    A(hipStreamSynchronize(w->stream));
    if (e != hipSuccess) return bail(GSWT_ERR_HIP);
    d.sp.height_map = w->height_map.p; d.lod_dist = w->lod_dist.p; d.tile_center = w->tile_center.p; d.tile_aabb = w->tile_aabb.p;
    d.presort_dirs = w->presort_dirs.p; d.splat_count = w->splat_count.p; d.nb = w->nb.p; d.cell = w->cell.p; d.st = w->st.p;
    d.head_off = w->head_off.p; d.head_len = w->head_len.p; d.pool = w->pool.p; d.counts = w->counts.p;
    // LDS budgets: the edge-merge tables (9 B per cell) must fit; the graph tables do for small maps and live in global memory otherwise
    w->lds_merge = (3 * cells + 3 * (cells / 2 + 1)) * sizeof(uint16_t);
    if (cfg->merge_type == MERGE_EDGE && w->lds_merge > 156u * 1024u) return bail(GSWT_ERR_CAPACITY);
    w->lds_graph = graph_tab_bytes((int)cells);
    w->graph_in_lds = w->lds_graph <= 140u * 1024u;     // beside the kernel's 8 KB of static LDS
    if (w->lds_merge > 48u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(k_w_merge), hipFuncAttributeMaxDynamicSharedMemorySize, (int)w->lds_merge) != hipSuccess)
        return bail(GSWT_ERR_HIP);
    if (w->graph_in_lds && w->lds_graph > 48u * 1024u &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_w_order_seq<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)w->lds_graph) != hipSuccess)
        return bail(GSWT_ERR_HIP);
#undef WTRY
    guard.w = nullptr;
    *out = w;
    return GSWT_OK;
} GSWT_WCATCH

void gswt_worker_destroy(gswt_worker* w)
{
    if (!w) return;
    hipSetDevice(w->device);
    if (w->stream) { hipStreamSynchronize(w->stream); hipStreamDestroy(w->stream); }
    w->height_map.release(); w->lod_dist.release(); w->tile_center.release(); w->tile_aabb.release(); w->presort_dirs.release();
    w->splat_count.release(); w->head_off.release(); w->head_len.release(); w->pool.release(); w->counts.release(); w->order.release();
    w->keys_a.release(); w->vals_a.release(); w->keys_b.release(); w->vals_b.release(); w->radix_ws.release(); w->pre_group.release();
    w->pre_member.release(); w->pre_len.release(); w->ent_len.release(); w->nb.release(); w->cell.release(); w->st.release(); w->gdir.release();
    w->gtab.release(); w->n64.release(); w->tiles.release(); w->groups.release(); w->members.release(); w->draws.release();
    for (auto& h : w->hs) { h.tiles.release(); h.groups.release(); h.members.release(); h.draws.release(); }
    w->h_counts.release();
    delete w;
}

const char* gswt_worker_last_error(const gswt_worker* w) { return w ? w->err.c_str() : "null worker"; }

int gswt_worker_set_cells(gswt_worker* w, const gswt_cell* cells, size_t n_cells, const int32_t center_coord[2])
try {
    if (!w) return GSWT_ERR_BAD_ARG;
    if (!cells || !center_coord || n_cells != (size_t)w->cells) return wfail(w, GSWT_ERR_BAD_ARG, "gswt_worker_set_cells: expected the whole map");
    for (size_t i = 0; i < n_cells; i++)
        if (cells[i].tile >= (uint32_t)w->dev.n_tile) return wfail(w, GSWT_ERR_BAD_ARG, "gswt_worker_set_cells: tile id out of range");
    WHIP(hipSetDevice(w->device));
    WHIP(hipStreamSynchronize(w->stream));
    WHIP(hipMemcpy(w->cell.p, cells, n_cells * sizeof(gswt_cell), hipMemcpyHostToDevice));
    WHIP(hipStreamSynchronize(nullptr));
    w->dev.sp.center_x = center_coord[0]; w->dev.sp.center_y = center_coord[1];
    if (w->dev.sp.surface_type != 2) {
        // update_tile_map re-creates every instance with merge_status None (:1745-1760); the sphere map is never rebuilt
        WHIP(hipMemsetAsync(w->st.p, 0, n_cells * sizeof(gswt_cell_state), w->stream));
        WHIP(hipMemsetAsync(w->head_len.p, 0, n_cells * 4, w->stream));
        WHIP(hipMemsetAsync(w->counts.p, 0, C_COUNT * 4, w->stream));
    }
    w->have_cells = true; w->have_lod = false; w->have_sort = false; w->sort_pending = false;
    return GSWT_OK;
} GSWT_WCATCH

int gswt_worker_update_lod(gswt_worker* w, const float cam_pos[3])
try {
    if (!w) return GSWT_ERR_BAD_ARG;
    if (!cam_pos) return wfail(w, GSWT_ERR_BAD_ARG, "gswt_worker_update_lod: null argument");
    if (!w->have_cells) return wfail(w, GSWT_ERR_STATE, "gswt_worker_update_lod before gswt_worker_set_cells");
    WHIP(hipSetDevice(w->device));
    const V3 cam{cam_pos[0], cam_pos[1], cam_pos[2]};
    hipLaunchKernelGGL(k_w_lod, dim3((w->cells + 255) / 256), dim3(256), 0, w->stream, w->dev, cam);
    WHIP(hipGetLastError());
    w->have_lod = true;
    return GSWT_OK;
} GSWT_WCATCH

int gswt_worker_sort_tiles(gswt_worker* w, const float cam_pos[3], const float vp16[16])
try {
    if (!w) return GSWT_ERR_BAD_ARG;
    if (!cam_pos || !vp16) return wfail(w, GSWT_ERR_BAD_ARG, "gswt_worker_sort_tiles: null argument");
    if (!w->have_lod) return wfail(w, GSWT_ERR_STATE, "WangTile::sort_tiles before build_tiles (gswt_worker_set_cells + gswt_worker_update_lod)");
    WHIP(hipSetDevice(w->device));
    const V3 cam{cam_pos[0], cam_pos[1], cam_pos[2]};
    const Mat4 vp = mat4_of(vp16);
    const WDev& d = w->dev;
    hipStream_t s = w->stream;
    const uint32_t cells = (uint32_t)w->cells, n_cand = cells * 4u;
    // counters of this event (the member pool restarts with an edge merge; an axis merge keeps what it has)
    WHIP(hipMemsetAsync(w->counts.p, 0, (d.merge_type == MERGE_AXIS ? C_POOL : C_COUNT) * sizeof(uint32_t), s));
    if (d.merge_type == MERGE_AXIS) WHIP(hipMemsetAsync(w->counts.p + C_POOL + 1, 0, (C_COUNT - C_POOL - 1) * sizeof(uint32_t), s));
    const bool need_edges = d.merge_type == MERGE_EDGE || d.sort_type == SORT_GRAPH;
    if (need_edges) hipLaunchKernelGGL(k_w_edges, dim3((n_cand + 255) / 256), dim3(256), 0, s, d, cam, vp, w->keys_a.p, w->vals_a.p, w->gdir.p);
    const uint32_t* sk = w->keys_a.p; const uint32_t* sv = w->vals_a.p;
    if (d.merge_type == MERGE_EDGE) {
        // (w->n64[0] = n_cand, w->n64[4] = cells: written once by gswt_worker_create)
        WHIP(hipMemsetAsync(w->radix_ws.p, 0, w->radix_words * sizeof(uint32_t), s));
        const int where = gswt::launch_sort(s, w->keys_a.p, w->vals_a.p, w->keys_b.p, w->vals_b.p, n_cand, w->n64.p, 32, w->radix_ws.p);
        if (where) { sk = w->keys_b.p; sv = w->vals_b.p; }
    }
    if (d.merge_type != MERGE_NONE) hipLaunchKernelGGL(k_w_merge, dim3(1), dim3(64), w->lds_merge, s, d, cam, vp, sk, sv);
    if (d.sort_type == SORT_DISTANCE || d.sort_type == SORT_VIEWPORT) {
        hipLaunchKernelGGL(k_w_order_keys, dim3((cells + 255) / 256), dim3(256), 0, s, d, cam, vp, w->keys_a.p, w->vals_a.p);
        WHIP(hipMemsetAsync(w->radix_ws.p, 0, w->radix_words * sizeof(uint32_t), s));
        const int where = gswt::launch_sort(s, w->keys_a.p, w->vals_a.p, w->keys_b.p, w->vals_b.p, cells, w->n64.p + 4, 32, w->radix_ws.p);
        hipLaunchKernelGGL(k_w_order_reverse, dim3((cells + 255) / 256), dim3(256), 0, s, d, where ? w->vals_b.p : w->vals_a.p, w->order.p);
    } else {
        if (w->graph_in_lds) hipLaunchKernelGGL(k_w_order_seq<true>, dim3(1), dim3(kSeqThreads), w->lds_graph, s, d, cam, w->gdir.p, w->gtab.p, w->order.p);
        else hipLaunchKernelGGL(k_w_order_seq<false>, dim3(1), dim3(kSeqThreads), 0, s, d, cam, w->gdir.p, w->gtab.p, w->order.p);
    }
    hipLaunchKernelGGL(k_w_scan, dim3(1), dim3(1024), 0, s, d, w->order.p, w->pre_group.p, w->pre_member.p, w->pre_len.p, w->ent_len.p);
    hipLaunchKernelGGL(k_w_records, dim3((cells + 255) / 256), dim3(256), 0, s, d, cam, w->order.p, w->pre_group.p, w->pre_member.p, w->pre_len.p, w->ent_len.p,
                       w->tiles.p, w->groups.p, w->members.p, w->draws.p);
    WHIP(hipGetLastError());
    w->have_sort = true; w->sort_pending = true;
    return GSWT_OK;
} GSWT_WCATCH

int gswt_worker_read_cell_state(gswt_worker* w, gswt_cell_state* out, size_t capacity)
try {
    if (!w) return GSWT_ERR_BAD_ARG;
    if (!out || capacity < (size_t)w->cells) return wfail(w, GSWT_ERR_BAD_ARG, "gswt_worker_read_cell_state: capacity");
    WHIP(hipSetDevice(w->device));
    WHIP(hipStreamSynchronize(w->stream));
    WHIP(hipMemcpy(out, w->st.p, (size_t)w->cells * sizeof(gswt_cell_state), hipMemcpyDeviceToHost));
    for (int i = 0; i < w->cells; i++) if (out[i].merge != MS_TO) out[i].merged_to = 0;
    return GSWT_OK;
} GSWT_WCATCH

int gswt_worker_fetch(gswt_worker* w)
try {
    if (!w) return GSWT_ERR_BAD_ARG;
    if (!w->have_sort) return wfail(w, GSWT_ERR_STATE, "gswt_worker_fetch before gswt_worker_sort_tiles");
    WHIP(hipSetDevice(w->device));
    const int rc = read_counts(w);
    if (rc != GSWT_OK) return rc;
    int b = 0;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        while (b == w->published || b == w->in_use) b++;
    }
    gswt_worker::HostSet& h = w->hs[b];
    memcpy(h.counts, w->h_counts.p, sizeof(h.counts));
    const uint32_t* c = h.counts;
    hipStream_t s = w->stream;
    if (c[C_N_ORDER]) {
        WHIP(hipMemcpyAsync(h.tiles.p, w->tiles.p, c[C_N_ORDER] * sizeof(gswt_sorted_tile), hipMemcpyDeviceToHost, s));
        WHIP(hipMemcpyAsync(h.draws.p, w->draws.p, c[C_N_ORDER] * sizeof(gswt_draw), hipMemcpyDeviceToHost, s));
    }
    if (c[C_N_GROUPS]) WHIP(hipMemcpyAsync(h.groups.p, w->groups.p, c[C_N_GROUPS] * sizeof(gswt_merge_group), hipMemcpyDeviceToHost, s));
    if (c[C_N_MEMBERS]) WHIP(hipMemcpyAsync(h.members.p, w->members.p, c[C_N_MEMBERS] * sizeof(gswt_merge_member), hipMemcpyDeviceToHost, s));
    WHIP(hipStreamSynchronize(s));
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->published = b;
    }
    w->sort_pending = false;
    return GSWT_OK;
} GSWT_WCATCH

int gswt_worker_read_sort(gswt_worker* w, gswt_sort_data* out)
try {
    if (!w) return GSWT_ERR_BAD_ARG;
    if (!out) return wfail(w, GSWT_ERR_BAD_ARG, "gswt_worker_read_sort: null argument");
    if (!w->have_sort) return wfail(w, GSWT_ERR_STATE, "gswt_worker_read_sort before gswt_worker_sort_tiles");
    if (w->sort_pending) {
        const int rc = gswt_worker_fetch(w);
        if (rc != GSWT_OK) return rc;
    }
    const gswt_worker::HostSet& h = w->hs[w->published];
    const uint32_t* c = h.counts;
    memset(out, 0, sizeof(*out));
    out->n_tiles = c[C_N_ORDER]; out->tiles = h.tiles.p;
    out->n_merged = (size_t)c[C_MERGED_LO] | ((size_t)c[C_MERGED_HI] << 32);
    out->n_groups = c[C_N_GROUPS]; out->n_members = c[C_N_MEMBERS];
    out->groups = h.groups.p; out->members = h.members.p;
    return GSWT_OK;
} GSWT_WCATCH

int gswt_set_draws_from_worker(gswt_ctx* ctx, gswt_worker* w)
try {
    if (!ctx || !w) return GSWT_ERR_BAD_ARG;
    if (w->ctx != ctx) return GSWT_ERR_BAD_ARG;
    int b;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        b = w->published;
        if (b < 0) return GSWT_ERR_STATE;              // nothing fetched yet (gswt_worker_fetch publishes a sort event)
        w->in_use = b;
    }
    const gswt_worker::HostSet& h = w->hs[b];
    const uint32_t* c = h.counts;
    const int r = gswt_set_draws_merge_groups(ctx, h.draws.p, (int)c[C_N_ORDER], h.groups.p, (int)c[C_N_GROUPS], h.members.p, (int)c[C_N_MEMBERS]);
    {
        std::lock_guard<std::mutex> lk(w->mu);
        w->in_use = -1;
    }
    return r;                                          // failure text: gswt_last_error(ctx)
} GSWT_WCATCH

}  // extern "C"
