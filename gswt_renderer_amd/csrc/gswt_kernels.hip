// gswt_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the GSWT hot path.
//
//   k_cull        : per-draw viewport cull + lod_enable skip (renderer.rs:472-497), column-band cull; clears the frame counters
//   k_project     : Wang-tile instancing + vs_main (gswt.wgsl:27-422) per list entry: None / HeightMap / Sphere surface,
//                   LOD blend, EWA projection, debug draw modes; <STRICT> (default) evaluates the shader text operator by operator
//   k_totals/k_emit : (splat, 16x16 screen tile) pair emission in composite order, no scan launches; k_emit<DEPTH> also keys every
//                   pair with its splat's depth bits (GSWT_ORDER_DEPTH: the global radix depth sort)
//   radix sort    : stable LSD sort of the pairs on the tile bits (reference order), preceded by passes on the depth bits in use with
//                   the tile id as payload (depth order); k_mg_* : merged-group lists on the device
//   k_items       : work items (tile, segment of its list: GSWT_OPT_SEGMENT pairs) from the per-tile [start, end) the last pass leaves
//   k_composite   : front-to-back alpha compositing (fs_main gswt.wgsl:425-435 + blend/depth state
//                   renderer.rs:118-129,179-185): LDS-staged batches, per-sub-block lists, wave ballot early termination;
//                   k_composite_dw: the same with the four waves of an item decoupled; <FOLD>: folds the segment partials itself
//   k_combine     : folds the segment partials of long tile lists; k_unshard : all-gathered shards -> frame
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off.  Contraction is OFF for the
// whole file: the float sequences that feed discontinuous decisions (culling, |p|^2 <= 4,
// depth test) are the canonical sequences of DESIGN.md and must round exactly like the
// CPU oracle; every fused multiply-add below is an explicit fmaf().
// Wavefront = 64 lanes everywhere in this file.
#include "gswt_device.h"

namespace gswt {

// Lane mask of a predicate.  (Not __ballot(int): its argument is an int, so the compiler first materialises the predicate as 0 / 1 in a
// vector register and compares that again -- two vector instructions per ballot that v_cmp had already answered.)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// bits of a lane mask below the calling lane (v_mbcnt_lo / _hi): a lane's rank among the lanes the mask names
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// Wave-wide scans and sums through DPP (data-parallel primitives: the neighbour's register arrives with the VALU instruction) instead of
// __shfl_up / __shfl_down, which compile to ds_bpermute_b32 -- an LDS-crossbar round trip per step, six of them in a dependent chain per
// scan.  row_shr:1 / 2 / 4 / 8 make the inclusive scan of every 16-lane row (a lane whose source lies outside its row adds 0),
// row_bcast:15 on rows 1 and 3 adds the row in front, row_bcast:31 on rows 2 and 3 the first half of the wave: the GFX9 sequence of
// rocPRIM's warp scan.  Every lane of the wave must be active (it was so with the shuffles: a lane that has left reads as undefined there).
// Integer adds: the same values bit for bit.  (End of round 4: the scans of k_radix_scatter / k_emit / k_items / k_totals / the tile-local
// depth sort and the sums of k_project; profiles/r04_dpp_scans.txt.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, true);
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane)
{
    (void)lane;
    v += dpp_or_zero<0x111, 0xF>(v);          // row_shr:1
    v += dpp_or_zero<0x112, 0xF>(v);          // row_shr:2
    v += dpp_or_zero<0x114, 0xF>(v);          // row_shr:4
    v += dpp_or_zero<0x118, 0xF>(v);          // row_shr:8
    v += dpp_or_zero<0x142, 0xA>(v);          // row_bcast:15 -> rows 1, 3
    v += dpp_or_zero<0x143, 0xC>(v);          // row_bcast:31 -> rows 2, 3
    return v;
}
// the wave's smallest / largest value, in every lane (uniform): the same six DPP steps with min / max; a lane whose source lies outside its row
// (or whose row the step does not address) compares with itself
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_self(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ uint32_t wave_min(uint32_t v)
{
    v = min(v, dpp_or_self<0x111, 0xF>(v)); v = min(v, dpp_or_self<0x112, 0xF>(v)); v = min(v, dpp_or_self<0x114, 0xF>(v));
    v = min(v, dpp_or_self<0x118, 0xF>(v)); v = min(v, dpp_or_self<0x142, 0xA>(v)); v = min(v, dpp_or_self<0x143, 0xC>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ uint32_t wave_max(uint32_t v)
{
    v = max(v, dpp_or_self<0x111, 0xF>(v)); v = max(v, dpp_or_self<0x112, 0xF>(v)); v = max(v, dpp_or_self<0x114, 0xF>(v));
    v = max(v, dpp_or_self<0x118, 0xF>(v)); v = max(v, dpp_or_self<0x142, 0xA>(v)); v = max(v, dpp_or_self<0x143, 0xC>(v));
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
// the wave's sum, in every lane (uniform: a scalar register)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0u), 63);
}


__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ float clampf(float e, float lo, float hi) { return fminf(fmaxf(e, lo), hi); }

// halfToFloat, gswt.wgsl:478-494: normals as IEEE; subnormals scale 2^-15 (f * 2^-25);
// Inf/NaN -> 0.
// (v_cvt_f32_f16 is exact for every finite half, subnormals included: the shader's subnormal scale is half of IEEE's, and its
// Inf / NaN are 0 -- two selects on the exponent field instead of a branchy bit construction.)
__device__ __forceinline__ float half_decode(uint32_t h)
{
    // on the CONVERTED value: a half's Inf / NaN converts to Inf / NaN (-> 0), its subnormals and zeros to |x| < 2^-14, the smallest normal
    // half (-> x / 2: exact), everything else is already the shader's value
    const float x = __half2float(__ushort_as_half((unsigned short)h));
    const float y = fabsf(x) < 6.103515625e-05f ? x * 0.5f : x;
    return __builtin_amdgcn_class(x, 0x3 | 0x4 | 0x200) ? 0.0f : y;      // signalling / quiet NaN, -Inf, +Inf
}

// x mod w for the repeat sampler, identical to ((x % w) + w) % w in integer arithmetic but without the 64-bit
// division sequences (5 samples x 4 wraps per splat made the HeightMap path ~3000 instructions): the quotient is
// estimated in float (exact operands below 2^23, so it is off by at most one) and fixed up with two compares.
__device__ __noinline__ int wrap_repeat_slow(float fx, int w)
{
    const long xl = (long)fx;
    return (int)(((xl % w) + w) % w);
}

__device__ __forceinline__ int wrap_repeat(float fx, int w)
{
    if (fabsf(fx) < 8388608.0f && w < 8388608) {
        const int x = (int)fx;
        // a power-of-two map (the reference resizes its random maps to 1024 x 1024, wangtile.rs:405-412): two's-complement AND is the
        // mathematical modulus; the branch is uniform
        if ((w & (w - 1)) == 0) return x & (w - 1);
        int r = x - w * (int)floorf((float)x / (float)w);
        if (r < 0) r += w;
        if (r >= w) r -= w;
        return r;
    }
    return wrap_repeat_slow(fx, w);                 // far outside any real map: the exact 64-bit path, out of line
}

// WebGPU bilinear sample, R32Float, repeat addressing, level 0 (renderer.rs:376-388).
// The two texels of a row are neighbours unless the cell straddles the map's seam: one 8-byte load per row (the buffer carries one
// float of padding behind its last row), the seam case re-reads column 0.  Same texels, same arithmetic as four scalar loads:
// a splat on the HeightMap surface issues 10 height loads instead of 20.
__device__ __forceinline__ float sample_height(const float* __restrict__ hm, int w, int h, float u, float v)
{
    float x = u * (float)w - 0.5f;
    float y = v * (float)h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float tx = x - fx0, ty = y - fy0;
    const int xa = wrap_repeat(fx0, w), ya = wrap_repeat(fy0, h);
    const int yb = ya + 1 == h ? 0 : ya + 1;
    const float* r0 = hm + (size_t)ya * w + xa;
    const float* r1 = hm + (size_t)yb * w + xa;
    // (a two-float vector type with 4-byte alignment: global memory takes a dword-aligned 8-byte load as ONE global_load_dwordx2;
    // a memcpy of 8 bytes at alignment 4 is lowered to two dword loads)
    typedef float hm_pair __attribute__((ext_vector_type(2), aligned(4)));
    hm_pair p0 = *reinterpret_cast<const hm_pair*>(r0), p1 = *reinterpret_cast<const hm_pair*>(r1);
    if (xa + 1 == w) { p0.y = hm[(size_t)ya * w]; p1.y = hm[(size_t)yb * w]; }
    float i00 = p0.x, i10 = p0.y;
    float i01 = p1.x, i11 = p1.y;
    float i0 = i00 * (1.0f - tx) + i10 * tx;
    float i1 = i01 * (1.0f - tx) + i11 * tx;
    return i0 * (1.0f - ty) + i1 * ty;
}

// Canonical sin / cos (operation sequence fixed in DESIGN.md section 4; the CPU checker restates it): WGSL leaves sin()/cos()
// accuracy to the implementation, so what must hold is CPU-oracle == GPU bit for bit.
__device__ __forceinline__ void csincosf(float x, float& sn, float& cs)
{
    const float kf = rintf(x * 0.636619772367581343f);
    float r = fmaf(kf, -1.5703125f, x);
    r = fmaf(kf, -4.837512969970703125e-4f, r);
    r = fmaf(kf, -7.54978995489188216e-8f, r);
    const float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    const float s = fmaf(ps * z, r, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    const float c = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)kf & 3;
    float so = (q & 1) ? c : s, co = (q & 1) ? s : c;
    if (q == 2 || q == 3) so = -so;
    if (q == 1 || q == 2) co = -co;
    sn = so; cs = co;
}

// sphere_get_uv + sphere_uv_to_pos, gswt.wgsl:515-564
__device__ __forceinline__ void sphere_point(float block_w, float bidx, float bidy, float bx, float by, float p[3])
{
    const float PI = 3.1415926535897932384626433832795f;
    float u, v;
    if (bidy == 0.0f) {
        if (by < bx) {
            if (bx - by == block_w) u = 0.0f;
            else u = (by / (block_w - (bx - by)) + bidx) / 5.0f;
            v = ((block_w - (bx - by)) / block_w) / 3.0f;
        } else {
            u = (bx / block_w + bidx) / 5.0f + ((by - bx) / block_w) * 0.1f;
            v = ((by - bx) / block_w) / 3.0f + (1.0f / 3.0f);
        }
    } else {
        if (by < bx) {
            u = (bx / block_w + bidx) / 5.0f + ((block_w - (bx - by)) / block_w) * 0.1f;
            v = ((block_w - (bx - by)) / block_w) / 3.0f + (1.0f / 3.0f);
        } else {
            if (by - bx == block_w) u = 0.0f;
            else u = (bx / (block_w - (by - bx)) + bidx) / 5.0f + 0.1f;
            v = ((by - bx) / block_w) / 3.0f + (2.0f / 3.0f);
        }
    }
    u = u + 0.5f * floorf(v);
    u = u * (2.0f * PI);
    v = (v - 0.5f) * PI;
    float su, cu, sv, cv;
    csincosf(u, su, cu);
    csincosf(v, sv, cv);
    p[0] = cv * cu; p[1] = cv * su; p[2] = sv;
}

// rand(), gswt.wgsl:502-504
__device__ __forceinline__ float dbg_rand(float cx, float cy)
{
    const float d = cx * 12.9898f + cy * 78.233f;
    float sn, cs;
    csincosf(d, sn, cs);
    const float v = sn * 43758.5453f;
    return v - floorf(v);
}

// Debug draw recolouring, gswt.wgsl:268-399 (draw_mode 1..4); rgb in / out
__device__ __forceinline__ void debug_draw_color(const Frame& f, const DrawDev& d, float vx, float vy, uint32_t lod_id, float t_ratio,
                                                 float& cr, float& cg, float& cb)
{
    const float tw = f.tile_width;
    if (f.draw_mode == 1u) {
        const float g = clampf(((cr + cg) + cb) / 0.6f, 0.0f, 1.0f);
        cr = cg = cb = g;
        const float margin = 0.05f * tw;
        const bool sphere = f.surface_type == 2u;
        if (d.single_draw == 1u) {
            cr = cr * dbg_rand(d.off[0], d.off[1]);
            cg = cg * dbg_rand(d.off[0] + 23.45f, d.off[1] + 23.45f);
            cb = cb * dbg_rand(d.off[0] + 67.89f, d.off[1] + 67.89f);
        } else if (vx < margin || vx > tw - margin) {
            const uint32_t bit = vx < margin ? d.tile_idx / 8u % 2u : d.tile_idx / 2u % 2u;
            if (vy < margin || vy > tw - margin) { cr = cg = cb = 0.5f; }
            else if (bit == 0u) { cr = 1.0f; cg = 0.0f; cb = 0.0f; }
            else { cr = 0.0f; cg = 1.0f; cb = 0.13f; }
        } else if (vy < margin || vy > tw - margin) {
            const uint32_t bit = vy < margin ? d.tile_idx % 2u : d.tile_idx / 4u % 2u;
            if (bit == 0u) {
                if (sphere) { cr = 1.0f; cg = 0.0f; cb = 0.0f; } else { cr = 1.0f; cg = 0.85f; cb = 0.0f; }
            } else {
                if (sphere) { cr = 0.0f; cg = 1.0f; cb = 0.13f; } else { cr = 0.0f; cg = 0.58f; cb = 1.0f; }
            }
        }
    } else if (f.draw_mode == 2u || f.draw_mode == 3u) {
        if (t_ratio > 0.0f && t_ratio < 1.0f) { cr = cg = cb = 0.0f; return; }
        if (f.draw_mode == 2u && d.changing == 1u) { cr = 0.0f; cg = 1.0f; cb = 0.0f; return; }
        uint32_t L = d.tile_lod;
        if (f.draw_mode == 3u) L = d.single_lod_id >= 0 ? (uint32_t)d.single_lod_id : lod_id;
        float cx = 0.0f, cy = 1.0f;
        if (L < 3u) cx = (3.0f - (float)L) / 3.0f;
        else cy = (6.0f - (float)L) / 3.0f;
        cr = 0.5f; cg = cx; cb = cy;
    } else if (f.draw_mode == 4u) {
        const uint32_t v = d.tile_view;
        float cx = 0.0f, cy = 0.0f;
        if (v < 4u) cx = (4.0f - (float)v) / 4.0f;
        if (v >= 4u) cy = (8.0f - (float)v) / 4.0f;
        if (v >= 8u) { cx = 1.0f; cy = 1.0f; }
        cr = 0.5f; cg = cx; cb = cy;
    }
}

// Screen tile id -> (column, row) of this ctx's tile grid.  The id is uniform per workgroup: for ids and widths below 2^16 the quotient is
// the high word of id * (floor(2^32 / tiles_x) + 1) -- one scalar multiply instead of the ~25 vector instructions of a 32-bit division
// that every wave of k_composite / k_combine used to spend on it (exact: the error of the product stays below 2^-16 <= 1 / tiles_x).
__device__ __forceinline__ void tile_xy(const Frame& f, uint32_t tile, int& tx, int& ty)
{
    uint32_t q;
    if (f.tiles_x_magic != 0u && tile < 65536u) q = __umulhi(tile, f.tiles_x_magic);
    else q = tile / (uint32_t)f.tiles_x;
    tx = (int)(tile - q * (uint32_t)f.tiles_x); ty = (int)q;
}

// Number of 16-px tile rows in [ty0, ty1] owned by this shard (row % count == index).
__device__ __forceinline__ int owned_rows(int ty0, int ty1, int index, int count)
{
    if (count <= 1) return ty1 - ty0 + 1;
    // first owned row >= ty0
    int first = ty0 + ((index - ty0 % count) + count) % count;
    if (first > ty1) return 0;
    return (ty1 - first) / count + 1;
}

// Chunk tables of a draw set, written on the device from the draw records (per sort event): a chunk = 256 list entries of
// one draw.  chunk_tab is in slot order (chunk c = slots c * 256 ...); chunk_tab_xcd is k_project's launch order: the
// chunks of draw d sit in the list of XCD d % 8, lists interleaved so that launch position p runs on XCD p % 8 (short
// lists are padded with 0xFFFFFFFF by the eight extra workgroups at the end of the grid).  One workgroup per draw.
struct XcdLens { uint32_t len[8]; uint32_t longest; };
__global__ __launch_bounds__(256) void k_chunk_tabs(const DrawDev* __restrict__ draws, const uint32_t* __restrict__ xcd_first, uint32_t n_draws,
                                                    uint2* __restrict__ chunk_tab, uint2* __restrict__ chunk_tab_xcd, const XcdLens xl)
{
    const uint32_t d = blockIdx.x;
    if (d >= n_draws) {
        const uint32_t x = d - n_draws;              // 0..7: padding of XCD x's list up to the longest one
        for (uint32_t k = xl.len[x & 7u] + threadIdx.x; k < xl.longest; k += 256u) chunk_tab_xcd[(size_t)k * 8u + x] = make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
        return;
    }
    const uint32_t nch = (draws[d].count + (uint32_t)kChunk - 1u) / (uint32_t)kChunk;
    const uint32_t c0 = draws[d].slot_base / (uint32_t)kChunk, x0 = xcd_first[d];
    for (uint32_t k = threadIdx.x; k < nch; k += 256u) {
        const uint2 e = make_uint2(d, k * (uint32_t)kChunk);
        chunk_tab[c0 + k] = e;
        chunk_tab_xcd[(size_t)(x0 + k) * 8u + (draws[d].xcd & 7u)] = e;
    }
}

// ------------------------------------------------------------------------------------
// k_cull: the CPU viewport culling + lod_enable skip of renderer.rs:472-497, one thread per draw.
// ------------------------------------------------------------------------------------
// Column-band sharding: can any splat whose centre lies in the box [lo, hi] (world space, before scene_scale) reach this
// rank's pixel columns?  Conservative: the centres lie in the convex hull of the 8 projected box corners (all in front of
// the camera, else keep), and a splat's pixel half extent is <= 2 s sqrt(lambda1) with lambda1 <= |J|_F^2 max(scene_scale)^2
// trace(Vrk) at the nearest depth, capped by the 1024-px axis clamp (gswt.wgsl:257-258); 25 % + 2 px of slack on top.
// lo / hi: bounds of the splat centres BEFORE scene_scale and surface mapping (cell origin + the scene's tile-local bounds).
// HeightMap surface (gswt.wgsl:565-599): the mapped centre is (x, y, h(x, y) hz) + n z with |n| = 1, so x and y move by at most
// |z| and the height lies in [surf_zlo - |z|, surf_zhi + |z|]; the covariance is F Vrk F^T, lambda1 <= |F|_F^2 trace(Vrk).
// Sphere surface: sphere_cell_box above (bidx, bidy: the block of the draw's cell; unused elsewhere).
// Sphere surface (gswt.wgsl:515-564, 600-623): a splat centre is lz (R + z), lz = the unit vector of the strip parametrisation at the
// centre's block coordinates (bx, by).  Inside one block, 0 <= bx, by <= block_w, that map is continuous (also across the block's
// diagonal and at the pole) and |d lz| <= K (|d bx| + |d by|) / block_w with K = max(0.2 pi + pi / 3, 2 pi^2 / 15 + pi / 3) = 2.363
// (equatorial triangles: u, v linear; polar triangles: cos v <= pi (block_w - t) / (3 block_w) against |du| <= 2 pi / (5 (block_w - t))):
// the centres of a cell lie within rho = R K (w + h) / (2 n block_w) + max |z| of the lz R of n x n sample points of its footprint.
// K = 2.5 and 2 % on rho are the slack for the +-0.001 y_max taps of F and the rounding of the evaluation.  A footprint that leaves
// its block (cells astride a block seam) is not bounded: such cells are kept.  lo / hi: scaled flat bounds; out: bounds of the centres.
__device__ __forceinline__ bool sphere_cell_box(const Frame& f, float bidx, float bidy, const float lo[3], const float hi[3], float blo[3], float bhi[3])
{
    const float xmax = ((float)f.map_half_wh[0] * 2.0f) * f.tile_width;
    const float block_w = xmax / 5.0f;
    const float ox = (float)(f.center_coord[0] - (int32_t)f.map_half_wh[0]) * f.tile_width + bidx * block_w;
    const float oy = (float)(f.center_coord[1] - (int32_t)f.map_half_wh[1]) * f.tile_width + bidy * block_w;
    const float bx0 = lo[0] - ox, bx1 = hi[0] - ox, by0 = lo[1] - oy, by1 = hi[1] - oy;
    if (!(block_w > 0.0f && bx0 >= 0.0f && bx1 <= block_w && by0 >= 0.0f && by1 <= block_w)) return false;
    constexpr int n = 3;
    const float R = fabsf(f.sphere_radius);
    const float zmax = fmaxf(fabsf(lo[2]), fabsf(hi[2]));
    const float rho = (R * 2.5f * ((bx1 - bx0) + (by1 - by0)) / (2.0f * (float)n * block_w) + zmax) * 1.02f + 1e-4f * R;
    if (!(rho == rho)) return false;
    for (int k = 0; k < 3; k++) { blo[k] = 3.402823466e+38f; bhi[k] = -3.402823466e+38f; }
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) {
            float p[3];
            sphere_point(block_w, bidx, bidy, bx0 + (bx1 - bx0) * (((float)i + 0.5f) / (float)n), by0 + (by1 - by0) * (((float)j + 0.5f) / (float)n), p);
            for (int k = 0; k < 3; k++) { blo[k] = fminf(blo[k], p[k] * f.sphere_radius - rho); bhi[k] = fmaxf(bhi[k], p[k] * f.sphere_radius + rho); }
        }
    return blo[0] == blo[0] && bhi[0] == bhi[0] && blo[1] == blo[1] && bhi[1] == bhi[1] && blo[2] == blo[2] && bhi[2] == bhi[2];
}

__device__ __forceinline__ bool band_misses(const Frame& f, const float lo[3], const float hi[3], float bidx, float bidy)
{
    float blo[3], bhi[3];
    for (int k = 0; k < 3; k++) {
        const float a = lo[k] * f.scene_scale[k], b = hi[k] * f.scene_scale[k];
        blo[k] = fminf(a, b); bhi[k] = fmaxf(a, b);
    }
    if (f.surface_type == 1u) {
        const float zp = fmaxf(fabsf(blo[2]), fabsf(bhi[2])) * 1.0001f;
        blo[0] -= zp; bhi[0] += zp; blo[1] -= zp; bhi[1] += zp;
        blo[2] = f.surf_zlo - zp; bhi[2] = f.surf_zhi + zp;
    } else if (f.surface_type == 2u) {
        float slo[3], shi[3];
        if (!sphere_cell_box(f, bidx, bidy, blo, bhi, slo, shi)) return false;
        for (int k = 0; k < 3; k++) { blo[k] = slo[k]; bhi[k] = shi[k]; }
    }
    float xmin = 3.402823466e+38f, xmax = -3.402823466e+38f, wmin = 3.402823466e+38f;
    bool odd = false;                 // a NaN anywhere: keep (fminf / fmaxf drop NaN operands, so the extremes alone would not show it)
    for (int k = 0; k < 8; k++) {
        const float px = (k & 1) ? bhi[0] : blo[0];
        const float py = (k & 2) ? bhi[1] : blo[1];
        const float pz = (k & 4) ? bhi[2] : blo[2];
        const float cx = ((f.VP[0] * px + f.VP[4] * py) + f.VP[8] * pz) + f.VP[12];
        const float cw = ((f.VP[3] * px + f.VP[7] * py) + f.VP[11] * pz) + f.VP[15];
        wmin = fminf(wmin, cw);
        const float xp = (0.5f * (cx / cw) + 0.5f) * f.W;
        odd = odd || !(cw == cw) || !(xp == xp);
        xmin = fminf(xmin, xp); xmax = fmaxf(xmax, xp);
    }
    if (odd || !(wmin > 1e-6f && xmin == xmin && xmax == xmax)) return false;
    const float smax = fmaxf(fabsf(f.scene_scale[0]), fmaxf(fabsf(f.scene_scale[1]), fabsf(f.scene_scale[2])));
    const float hx = 1.3f * f.htan[0], hy = 1.3f * f.htan[1];
    const float jn2 = (f.focal[0] * f.focal[0] * (1.0f + hx * hx) + f.focal[1] * f.focal[1] * (1.0f + hy * hy)) / (wmin * wmin);
    const float lam = jn2 * smax * smax * fmaxf(f.loc_max_trace, 0.0f) * f.surf_f2;
    const float ss = fabsf(f.splat_scale);
    const float rad = fminf(2.0f * ss * sqrtf(lam), 1448.2f * ss) * 1.25f + 2.0f;
    if (!(rad == rad)) return false;
    return xmax + rad < (float)(f.col0 * kTile) || xmin - rad >= (float)(f.col1 * kTile);
}

// The CPU viewport culling + lod_enable skip of renderer.rs:472-497 for one draw (+ the band cull of a column-band shard).
// Band culling works on MAP CELLS: every splat of a Wang-tile instance lies in (cell origin + the scene's tile-local position
// bounds), f.loc_lo .. f.loc_hi, so a plain / blending draw is tested with its own offset and the members of a merged group
// through cell_culled[map id] (k_project skips such entries before gathering their records): the heaviest band no longer
// projects whole merged groups that merely touch it.  No per-sort-event bounds pass.
__device__ __forceinline__ bool draw_is_culled(const Frame& f, const DrawDev& d)
{
    bool culled = false;
    if (d.cull_enable) {
        float mx = 3.402823466e+38f, my = 3.402823466e+38f, mz = -3.402823466e+38f;
        for (int ci = 0; ci < 4; ci++) {
            float px = d.corners[3 * ci], py = d.corners[3 * ci + 1], pz = d.corners[3 * ci + 2];
            float c4[4];
            for (int r = 0; r < 4; r++)
                c4[r] = ((f.VP[r] * px + f.VP[4 + r] * py) + f.VP[8 + r] * pz) + f.VP[12 + r] * 1.0f;
            float cx = c4[0] / c4[3], cy = c4[1] / c4[3], cz = c4[2] / c4[3];
            if (fabsf(cx) < mx) mx = fabsf(cx);
            if (fabsf(cy) < my) my = fabsf(cy);
            if (cz > mz) mz = cz;
        }
        float clip = f.culling_dist;
        if (mz < -clip || mx > clip || my > clip) culled = true;
    }
    if (!((f.lod_enable_mask >> (d.lod & 31u)) & 1u)) culled = true;
    // Column-band sharding: drop the draw when none of its splats can touch this rank's pixel columns.  (Doing the same
    // cull on the host and launching k_project / k_emit over the surviving chunks only was measured: no gain -- the
    // workgroups of culled chunks exit at once and the frames overlap.)
    if (!culled && f.band_cull && d.count && d.single_draw != 1u) {
        const float lo[3] = {f.loc_lo[0] + d.off[0], f.loc_lo[1] + d.off[1], f.loc_lo[2] + d.off[2]};
        const float hi[3] = {f.loc_hi[0] + d.off[0], f.loc_hi[1] + d.off[1], f.loc_hi[2] + d.off[2]};
        float bidx = 0.0f, bidy = 0.0f;
        if (f.surface_type == 2u) {
            bidx = (float)(5u * d.map_coord[0] / (f.map_half_wh[0] * 2u));
            bidy = (float)(2u * d.map_coord[1] / (f.map_half_wh[1] * 2u));
        }
        if (band_misses(f, lo, hi, bidx, bidy)) culled = true;
    }
    return culled;
}

// ------------------------------------------------------------------------------------
// k_cull (first kernel of the frame): clears the per-frame accumulators (saves three memset launches), fills the band-cull table of a
// column-band shard, and builds the launch table of k_project for THIS frame, one thread per chunk (a chunk = 256 list entries of one draw): only the chunks of
// the draws that survive the reference's per-draw cull (draw_is_culled: every chunk's thread evaluates its own draw's -- until late in round 4 a
// kernel of its own in front of this one, one launch and ~6 us more), in the per-XCD layout of chunk_tab_xcd (position k * 8 + x runs on XCD x = the draw's DrawDev::xcd; the order inside an
// XCD's list is the order of the atomic adds: irrelevant, a chunk's output slots are fixed).  An entry carries what the chunk's first load
// needs -- the list position of its lane 0 and the list's length and arena -- so that k_project's list-word load does not wait for the draw record.
// Round 4: CHUNK-LEVEL FRUSTUM CULL.  The reference culls whole tile draws on the CPU (renderer.rs:472-494: min |x|, min |y|, max z of the four
// corner NDCs, which lets every tile through that has one corner near the view axis), and vs_main then drops splat by splat
// (gswt.wgsl:163-167).  A draw's list is presorted by depth, so a chunk of it is a slab of the tile: measured on the c3 frame, 49 % of the
// chunks of the surviving draws hold NO splat that passes vs_main's frustum test -- half of k_project's live workgroups gathered 256 records
// to find that out.  Every static list carries the tile-local bounding box of each of its chunks (gswt_upload_scene); a chunk whose box,
// moved to the draw's offset, lies on the far side of ONE of the five planes of that test at all eight corners (clip-space coordinates are
// affine in the position, so then at every point of the box; a relative margin of 1e-3 covers the rounding of both evaluations) is left out
// of the table.  Nothing that vs_main would keep is dropped: images, visible and pair counts are bit-identical (GSWT_OPT_NO_CHUNK_CULL).
// Plain and HeightMap surfaces (the Sphere mapping moves a splat far from its flat position: no cull there); merged groups carry no boxes
// yet (their lists are rebuilt per sort event and mix member tiles) and stay whole.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cull(const Frame f, const DrawDev* __restrict__ draws, uint32_t* __restrict__ draw_culled,
                                              uint32_t* __restrict__ cell_culled, uint32_t n_cells,
                                              uint32_t* __restrict__ zero_a, uint32_t n_zero_a, uint32_t* __restrict__ zero_b, uint32_t n_zero_b,
                                              uint32_t* __restrict__ zero_c, uint32_t n_zero_c, uint32_t* __restrict__ zero_d, uint32_t n_zero_d,
                                              const uint2* __restrict__ chunk_tab, uint32_t n_chunks, const float* __restrict__ boxes, uint32_t chunk_cull,
                                              uint32_t* __restrict__ live_cnt, uint4* __restrict__ live_tab, uint32_t* __restrict__ live_cid)
{
    __shared__ uint32_t s_cnt[8], s_base[8];
    if (threadIdx.x < 8u) s_cnt[threadIdx.x] = 0u;
    const uint32_t gtid = blockIdx.x * 256u + threadIdx.x;
    for (uint32_t j = gtid; j < n_zero_a; j += gridDim.x * 256u) zero_a[j] = 0u;
    for (uint32_t j = gtid; j < n_zero_b; j += gridDim.x * 256u) zero_b[j] = 0u;
    for (uint32_t j = gtid; j < n_zero_c; j += gridDim.x * 256u) zero_c[j] = 0u;       // per-chunk pair counts: culled chunks never run
    for (uint32_t j = gtid; j < n_zero_d; j += gridDim.x * 256u) zero_d[j] = 0u;       // GSWT_ORDER_DEPTH: group rows / digit totals of the depth sort
    if (f.band_cull) {
        const uint32_t map_wh_y = f.map_wh_y;                                // 2 h + 1; 2 h on the Sphere (gswt.wgsl:52-63, 606-610)
        for (uint32_t j = gtid; j < n_cells; j += gridDim.x * 256u) {
            // the merged-member offset of gswt.wgsl:52-63, same expression as k_project
            const uint32_t mq = j / map_wh_y, mr = j % map_wh_y;
            const float ox = (float)((int32_t)(mq - f.map_half_wh[0]) + f.center_coord[0]) * f.tile_width;
            const float oy = (float)((int32_t)(mr - f.map_half_wh[1]) + f.center_coord[1]) * f.tile_width;
            const float lo[3] = {f.loc_lo[0] + ox, f.loc_lo[1] + oy, f.loc_lo[2]}, hi[3] = {f.loc_hi[0] + ox, f.loc_hi[1] + oy, f.loc_hi[2]};
            float bidx = 0.0f, bidy = 0.0f;
            if (f.surface_type == 2u) {                                      // a merged member's block, as k_project derives it from the map id
                bidx = (float)(5u * mq / (f.map_half_wh[0] * 2u));
                bidy = (float)(2u * mr / (f.map_half_wh[1] * 2u));
            }
            cell_culled[j] = band_misses(f, lo, hi, bidx, bidy) ? 1u : 0u;
        }
    }
    __syncthreads();
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    bool live = false;
    uint4 e = make_uint4(0u, 0u, 0u, 0u);
    uint32_t x = 0, rank = 0;
    if (c < n_chunks) {
        const uint2 ct = chunk_tab[c];
        const DrawDev& d = draws[ct.x];
        const bool dc = draw_is_culled(f, d);
        if (ct.y == 0u) draw_culled[ct.x] = dc ? 1u : 0u;      // (read by the debug-varyings build of k_project only)
        live = !dc && d.count != 0u;
        if (live && chunk_cull && d.merged == 0u && d.box_base != 0xFFFFFFFFu && f.surface_type <= 1u) {
            const float* b = boxes + 6u * (size_t)(d.box_base + (ct.y >> 8));
            float lo[3], hi[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {                 // the same (pos + offset) * scene_scale as vs_main: rounding is monotonic, so the box still holds every centre
                const float a = (b[k] + d.off[k]) * f.scene_scale[k], z = (b[3 + k] + d.off[k]) * f.scene_scale[k];
                lo[k] = fminf(a, z); hi[k] = fmaxf(a, z);
            }
            if (f.surface_type == 1u) {
                // HeightMap surface (gswt.wgsl:565-599): the mapped centre is (x, y, h(x, y) hz) + n z with |n| = 1 and h between the map's
                // extremes (k_cull's band cull bounds a cell the same way): x and y move by at most |z|, the height lies in
                // [surf_zlo - |z|, surf_zhi + |z|]; a little slack for the rounding of the bilinear sample and of n
                const float zp = fmaxf(fabsf(lo[2]), fabsf(hi[2])) * 1.0001f + 1e-6f;
                lo[0] -= zp; hi[0] += zp; lo[1] -= zp; hi[1] += zp;
                lo[2] = f.surf_zlo - zp - 1e-4f * (fabsf(f.surf_zlo) + 1.0f); hi[2] = f.surf_zhi + zp + 1e-4f * (fabsf(f.surf_zhi) + 1.0f);
            }
            bool xp = true, xn = true, yp = true, yn = true, zn = true;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float px = (k & 1) ? hi[0] : lo[0], py = (k & 2) ? hi[1] : lo[1], pz = (k & 4) ? hi[2] : lo[2];
                float cv[4], q[4];
#pragma unroll
                for (int r = 0; r < 4; r++) cv[r] = ((f.V[r] * px + f.V[4 + r] * py) + f.V[8 + r] * pz) + f.V[12 + r];
#pragma unroll
                for (int r = 0; r < 4; r++) q[r] = ((f.GP[r] * cv[0] + f.GP[4 + r] * cv[1]) + f.GP[8 + r] * cv[2]) + f.GP[12 + r] * cv[3];
                const float clip = 1.2f * q[3];
                const float m = 1e-3f * (((fabsf(q[0]) + fabsf(q[1])) + fabsf(q[2])) + fabsf(clip)) + 1e-3f;
                xp = xp && (q[0] - clip > m); xn = xn && (-q[0] - clip > m);        // (a NaN compares false: such a chunk stays)
                yp = yp && (q[1] - clip > m); yn = yn && (-q[1] - clip > m);
                zn = zn && (-q[2] - clip > m);
            }
            if (xp || xn || yp || yn || zn) live = false;
        }
        x = d.xcd & 7u;
        if (live) {
            e = make_uint4(ct.x, ct.y, d.list_base + d.count - 1u - ct.y, d.count | (d.merged ? 0x80000000u : 0u));
            rank = atomicAdd(&s_cnt[x], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 8u && s_cnt[threadIdx.x] != 0u) s_base[threadIdx.x] = atomicAdd(&live_cnt[threadIdx.x * kSuperStride], s_cnt[threadIdx.x]);      // (a cache line per XCD's count: gswt_device.h)
    __syncthreads();
    if (live) {
        live_tab[(size_t)(s_base[x] + rank) * 8u + x] = e;
        live_cid[(size_t)(s_base[x] + rank) * 8u + x] = c;       // the chunk's number in slot order (k_emit walks the same table)
    }
}

// ------------------------------------------------------------------------------------
// k_project
// One workgroup = one 256-entry chunk of one draw, in composite (front-to-back) order:
// slot = slot_base + r, r-th entry from the END of the draw's list.
// ------------------------------------------------------------------------------------
// FULL = false is the lean variant the normal frame runs (surface None / HeightMap, draw_mode 0); the Sphere mapping
// (five sin / cos pairs per splat) and the debug recolouring live only in FULL = true.
// Measured and dropped (c3, isolated 103 us): (1) a persistent grid over a device-compacted chunk list -- static striding
// loses the dispatcher's dynamic balance (-8 % frames/s), a shared work counter serialises on one L2 address (230 us);
// (2) culling on the host and launching over the surviving chunks only -- the 20 k empty workgroups cost nothing
// measurable once two frames overlap (3519 vs 3517 frames/s); (3) four chunks per workgroup with all list / record
// loads issued up front (88 VGPRs, 123 us); (4) a per-chunk copy of the draw record + per-chunk cull flag to shorten the
// scalar chain chunk table -> draw -> flag: those were L2 hits, the 5.6 MB of copies are HBM misses (106 -> 113 us).
// (5) four table entries per workgroup, flags of the group fetched together and culled chunks settled at once (what
// paid in k_emit): unrolled, four inlined bodies = 77 VGPRs, 95 -> 101 us and c5 unchanged (527 frames/s); as a loop
// around one body the uniform float sub-expressions are hoisted into VGPRs (85).
// (6) the static lists a second time as 32-byte records in list order (plain / blending draws read 256 consecutive records,
// coalesced, addressed from the launch-table entry alone: no list word, no dependent gather; 150 MB at c3): 91.8 -> 88.2 us at c3,
// 128.1 -> 128.6 on the HeightMap surface, 461 -> 457 at c5 (stage events) -- the gather latency is hidden by the other waves.
// (7) column-band shards: a per-entry pre-pass (k_cull's cell table for merged groups, then position, the frustum test and a
// per-splat bound of the pixel extent against the band) followed by a compaction of the surviving entries to the front of the
// workgroup (ballot ranks + LDS exchange of entry / map id / slot), so that only ceil(survivors / 64) waves run the projection: the
// splats a rank of 8 carries through the whole projection fell from 471-619 k to 217-266 k (union of the bands still bitwise the
// unsharded image), and the kernel's time did not move (49-66 us per rank before and after; c5 441 -> 451 us per frame): in band
// mode, too, the kernel is paced by its workgroups' load chains, not by the projection's instructions.
// (9) round 3, with per-workgroup phase stamps (-DGSWT_TRACE, tools/project_trace.py): c3's 13.7 k live workgroups live 5.1 us each (launch
// table 0.4, list word 1.0, record 1.0, projection 1.7, barrier 0.5, sums 0.4) and start at ~260 per us, so that ~1 300 of the 2 048 slots are
// filled on average (each CU peaks at 7-8): lifetimes sum to 34 us x 2 048.  Two / four chunks per workgroup (half / a quarter of the
// workgroups for the dispatcher): 87 -> 101 -> 113 us.  A PERSISTENT grid taking launch positions from per-XCD queue heads (one returning
// atomic per chunk, issued a chunk ahead; the frame constants through device memory so that the loop does not hoist 150 scalar loads):
// 240 us with the eight heads in one cache line (~65 returning atomics per us and line), 114 us with the heads 256 bytes apart -- the loop
// costs 83 VGPRs (5 workgroups per CU, all of them busy: 1 150 resident, a chunk lives 5.8 us, 199 chunks per us against 264).
// (8) the grid sized from the previous frame's table of live chunks (+ 50 % + 64 per XCD list; k_totals flags a frame whose table
// outgrew its grid and the host re-runs it) instead of from the whole table: the dispatcher hands out ~4.6 k workgroups per us
// (flag 256: c5's 366 k workgroups that only read the launch table and leave take ~80 us), but the workgroups of culled chunks
// overlap with the live ones: project stage 461 -> 433 us at c5, 87 -> 87 us at c3, frame rates unchanged.
// -DGSWT_TRACE (tools/composite_trace.py): per work item, 100 MHz wall-clock stamps of the compositor's phases, left by lane 0 of wave 0:
// [0] entry, [1] item known, [2] first batch staged (gathers have arrived), [3] last walk done, [4] pairs of the item,
// [5] walk steps of wave 0 / of the wave, [6] ticks wave 0 spent in bin + walk, [7] hardware id (HW_ID | XCC_ID << 32)
#ifdef GSWT_TRACE
constexpr uint32_t kTraceItems = 1u << 17;
__device__ unsigned long long g_trace[kTraceItems * 8];
#define GSWT_TR(K, V) { if (tr_on) g_trace[(size_t)tr_item * 8u + (K)] = (unsigned long long)(V); }
#define GSWT_NOW() wall_clock64()
#else
#define GSWT_TR(K, V)
#define GSWT_NOW() 0ull
#endif
// k_project's stamps (tools/project_trace.py), rows 8192 + launch position: [0] entry, [1] launch-table entry known, [2] list word arrived (lane 0),
// [3] record arrived (lane 0, when it gets that far), [4] wave 0 through the projection, [5] behind the workgroup barrier, [6] end,
// [7] hardware id (HW_ID | XCC_ID << 32) | chunk has pairs << 63
// The quotients and roots of the projection are IEEE operations (correctly rounded: what the CPU checker reproduces).  -DGSWT_AB_FASTMATH
// (measurement only, WRONG bits) swaps them for the 1-ulp hardware approximations: the upper bound of what a cheaper exact sequence could gain.
// Ablation bits of GSWT_OPT_DEBUG_FLAGS (profiling: they cut a kernel short and the image is WRONG): compiled only into the measurement
// build (-DGSWT_EXPERIMENTS, `make variants`); the product library has no such branches and refuses a nonzero flag word.
#ifdef GSWT_EXPERIMENTS
#define GSWT_ABL(F, BITS) (((F).dbg_flags & (BITS)) != 0)
#else
#define GSWT_ABL(F, BITS) false
#endif
#ifdef GSWT_AB_FASTMATH
#define GSWT_RCP(X) __builtin_amdgcn_rcpf(X)
#define GSWT_SQRT(X) __builtin_amdgcn_sqrtf(X)
#else
#define GSWT_RCP(X) (1.0f / (X))
#define GSWT_SQRT(X) sqrtf(X)
#endif
// STRICT (GSWT_OPT_STRICT_VS): A6..A10 as the shader text writes them (gswt.wgsl:152-258, 260-265, 402-419) -- every `*`, `+`, `-`, `/` its own
// correctly rounded binary32 operation, matrix x vector as the left-to-right sum of column products, the full `scene_scale_mat * Vrk *
// transpose(..)` and `transpose(T) * Vrk * T` matrix products, length = sqrt(x*x + y*y), normalize = v / length(v), no fused multiply-add:
// the same operations, in the same order, as the CPU checker's strict mode, per splat bit for bit.  The default (sequence v2, DESIGN.md
// section 4) evaluates the same expressions with fma chains and one reciprocal per quotient -- also legal WGSL, ~25 % fewer instructions,
// and up to 5e-4 away from this one on thin ellipses (lambda2 = mid - radius cancels).
// HALVES = 2 (round 4, measured and NOT the default: see launch_project): a 512-thread workgroup takes TWO consecutive entries of its XCD's
// launch list, one per half (waves 0-3 / 4-7), side by side: the same body at the same 64 VGPRs, half the workgroups for the dispatcher
// to hand out (the trace reads as dispatch-bound: 13.7 k workgroups of ~5 us started at ~260 per us keep ~1 300 of the chip's 2 048
// workgroup slots filled; chunks IN SEQUENCE in one workgroup were slower: more registers, longer lifetime).  The halves meet only at
// the barrier in front of the per-chunk sums.
template <bool DEBUG, bool FULL, bool STRICT, int HALVES = 1>
__global__ __launch_bounds__(256 * HALVES) void k_project(
    const Frame f, const DrawDev* __restrict__ draws, const uint2* __restrict__ chunk_tab,
    const uint32_t* __restrict__ static_list, const uint32_t* __restrict__ merged_list,
    const uint32_t* __restrict__ merged_map, const uint4* __restrict__ tex,
    const float* __restrict__ hmap, const uint32_t* __restrict__ draw_culled, const uint32_t* __restrict__ cell_culled,
    const uint32_t* __restrict__ live_cnt, const uint4* __restrict__ live_tab, uint2* __restrict__ rects,
    Rec* __restrict__ recs, float* __restrict__ depths, uint32_t* __restrict__ block_sums, uint32_t* __restrict__ super_sums, uint32_t n_super,
    Varyings* __restrict__ dbg, float4* __restrict__ col_f)
{
    static_assert(HALVES == 1 || !DEBUG, "the debug-varyings build visits the static chunk table one entry per workgroup");
    __shared__ uint32_t s_wsum[4 * HALVES], s_wvis[4 * HALVES];
    // chunk_tab is in LAUNCH order, which is not slot order: workgroup b runs on XCD b % 8, and the table is laid out so
    // that all chunks of a draw land on one XCD (DrawDev::xcd) -- a draw's gathers stay inside one tile type's 313 KB of the
    // record table, so an XCD's 4 MB L2 then holds the few tile types it is working on instead of all 48 (6.6 MB).
    // The normal frame launches over k_cull's table of live chunks (same layout, culled draws left out; workgroups past an XCD's
    // live count exit after one cached scalar load); the debug-varyings build visits every chunk through the static table.
    uint2 ct;
    uint32_t list_top = 0, list_cnt = 0;      // non-DEBUG: list index of this chunk's lane 0, list length | merged << 31 (from k_cull)
    const uint32_t half = HALVES == 2 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0u;
    bool active = true;                        // HALVES == 2: the upper half of the last workgroup of an odd list has no entry
#ifdef GSWT_TRACE
    const bool tr_on = threadIdx.x == 0 && blockIdx.x < 49152u;
    const uint32_t tr_item = 8192u + blockIdx.x;
    GSWT_TR(0, GSWT_NOW())
#endif
    if (DEBUG) {
        ct = chunk_tab[blockIdx.x];
        if (ct.y == 0xFFFFFFFFu) return;    // padding of a short per-XCD list
    } else {
        const uint32_t x = blockIdx.x & 7u, k0 = (blockIdx.x >> 3) * (uint32_t)HALVES;
        const uint32_t n_live = live_cnt[x * kSuperStride];
        if (k0 >= n_live) return;
        active = k0 + half < n_live;
        const uint4 lt = live_tab[(size_t)min(k0 + half, n_live - 1u) * 8u + x];
        ct = make_uint2(lt.x, lt.y);
        list_top = lt.z; list_cnt = lt.w;
        if (GSWT_ABL(f, 256)) return;      // ablation: nothing behind the launch-table entry
#ifdef GSWT_TRACE
        { unsigned long long sink = lt.x + lt.w; asm volatile("" :: "s"(sink)); }
        GSWT_TR(1, GSWT_NOW())
#endif
    }
    const DrawDev& d = draws[ct.x];
    const uint32_t tid = HALVES == 2 ? (threadIdx.x & 255u) : threadIdx.x;
    const uint32_t cid = (d.slot_base + ct.y) >> 8;      // chunk id in slot space
    const bool s_culled = DEBUG ? draw_culled[ct.x] != 0u : false;

    const uint32_t r = ct.y + tid;
    const uint32_t slot = d.slot_base + r;
    const bool in_list = DEBUG ? r < d.count : (active && r < (list_cnt & 0x7FFFFFFFu));
    uint32_t count = 0;
    bool visible = false;
    uint2 my_rect = make_uint2(1u, 0u);     // empty: tx0 = 1 > tx1 = 0
    Varyings vout;
    if (DEBUG) { vout.visible = 0; vout.ndc[0] = vout.ndc[1] = vout.depth = 0.f; vout.major[0] = vout.major[1] = 0.f;
                 vout.minor[0] = vout.minor[1] = 0.f; vout.rgba[0] = vout.rgba[1] = vout.rgba[2] = vout.rgba[3] = 0.f; }

    if (in_list && !s_culled) {
        // list word (and, for a merged draw, the map id beside it: one round trip, not a third dependent one).  The normal frame
        // addresses it from the launch-table entry alone, so the load is in flight while the draw record is still being fetched.
        const bool mrg = DEBUG ? d.merged != 0u : (list_cnt >> 31) != 0u;
        const uint32_t li = DEBUG ? d.list_base + (d.count - 1u - r) : list_top - tid;
        const uint32_t* list = mrg ? merged_list : static_list;
        const uint32_t entry = list[li];
        const uint32_t map_id_m = mrg ? merged_map[li] : 0u;
        const uint32_t gs_index = entry & kIdxMask;
        const uint32_t lod_id = entry >> kLodShift;
#ifdef GSWT_TRACE
        { unsigned long long sink = entry; asm volatile("" :: "v"(sink)); }
        GSWT_TR(2, GSWT_NOW())
#endif
        do {
            // column-band shard: the member tile this entry belongs to cannot reach the band (k_cull's cell table)
            if (f.band_cull && d.single_draw == 1u && cell_culled[map_id_m] != 0u) break;
            // A1 gswt.wgsl:38-42
            if (d.valid_lod_id >= 0 && d.valid_lod_id != (int32_t)lod_id) break;
            if (GSWT_ABL(f, 128)) break;       // ablation: stop behind the list word, in front of the record gather
            // A2 :45-49
            const uint4 w0 = tex[2 * (size_t)gs_index];
            const uint4 w1 = tex[2 * (size_t)gs_index + 1];
#ifdef GSWT_TRACE
            { unsigned long long sink = w0.x + w1.w; asm volatile("" :: "v"(sink)); }
            GSWT_TR(3, GSWT_NOW())
#endif
            // A3 :52-65
            float ox = d.off[0], oy = d.off[1], oz = d.off[2];
            if (d.single_draw == 1u) {
                const uint32_t map_id = map_id_m;
                uint32_t map_wh_y = 2u * f.map_half_wh[1];
                if (f.surface_type != 2u) map_wh_y += 1u;
                // map_id / map_wh_y with a uniform divisor: for operands below 2^16 the quotient is the high word of map_id * (floor(2^32 / d) + 1)
                // (three instructions instead of the ~40 of a 32-bit division); larger maps take the division
                uint32_t mq, mr;
                if (f.map_wh_y_magic != 0u && map_id < 65536u) {
                    mq = __umulhi(map_id, f.map_wh_y_magic);
                    mr = map_id - mq * map_wh_y;
                } else { mq = map_id / map_wh_y; mr = map_id % map_wh_y; }
                ox = (float)((int32_t)(mq - f.map_half_wh[0]) + f.center_coord[0]) * f.tile_width;
                oy = (float)((int32_t)(mr - f.map_half_wh[1]) + f.center_coord[1]) * f.tile_width;
                oz = 0.0f;
            }
            float c0 = (u2f(w0.x) + ox) * f.scene_scale[0];
            float c1 = (u2f(w0.y) + oy) * f.scene_scale[1];
            float c2 = (u2f(w0.z) + oz) * f.scene_scale[2];
            // A4 :75-87
            float mapped_z = 0.0f;
            float F[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            if (f.surface_type == 1u) {
                const float DELTA = 0.001f;
                float xr = (2.0f * (float)f.map_half_wh[0] + 1.0f) * f.tile_width * f.height_map_scale[0];
                float yr = (2.0f * (float)f.map_half_wh[1] + 1.0f) * f.tile_width * f.height_map_scale[1];
                float h_u = (c0 + (float)f.map_half_wh[0] * f.tile_width) / xr;
                float h_v = (c1 + (float)f.map_half_wh[1] * f.tile_width) / yr;
                float hz = f.height_map_scale[2];
                float nz = sample_height(hmap, f.hm_w, f.hm_h, h_u, h_v) * hz;
                float dt = DELTA;
                float h_r = sample_height(hmap, f.hm_w, f.hm_h, h_u + dt, h_v) * hz;
                float h_l = sample_height(hmap, f.hm_w, f.hm_h, h_u - dt, h_v) * hz;
                float h_up = sample_height(hmap, f.hm_w, f.hm_h, h_u, h_v + dt) * hz;
                float h_d = sample_height(hmap, f.hm_w, f.hm_h, h_u, h_v - dt) * hz;
                float lx[3] = {1.0f, 0.0f, (h_r - h_l) / (2.0f * dt * xr)};
                float ly[3] = {0.0f, 1.0f, (h_up - h_d) / (2.0f * dt * yr)};
                float cz0 = lx[1] * ly[2] - lx[2] * ly[1];
                float cz1 = lx[2] * ly[0] - lx[0] * ly[2];
                float cz2 = lx[0] * ly[1] - lx[1] * ly[0];
                float len = sqrtf((cz0 * cz0 + cz1 * cz1) + cz2 * cz2);
                F[0] = lx[0]; F[1] = lx[1]; F[2] = lx[2];
                F[3] = ly[0]; F[4] = ly[1]; F[5] = ly[2];
                F[6] = cz0 / len; F[7] = cz1 / len; F[8] = cz2 / len;
                float z = c2;
                c0 = c0 + F[6] * z;
                c1 = c1 + F[7] * z;
                c2 = nz + F[8] * z;
                mapped_z = nz;
            } else if (FULL && f.surface_type == 2u) {
                // surface_mapping, gswt.wgsl:600-623 (icosahedral-strip parametrisation, 5 x 2 blocks)
                const float DELTA = 0.001f;
                const float xmax = ((float)f.map_half_wh[0] * 2.0f) * f.tile_width;
                const float ymax = ((float)f.map_half_wh[1] * 2.0f) * f.tile_width;
                const float block_w = xmax / 5.0f;
                const float nx = c0 - (float)(f.center_coord[0] - (int32_t)f.map_half_wh[0]) * f.tile_width;
                const float ny = c1 - (float)(f.center_coord[1] - (int32_t)f.map_half_wh[1]) * f.tile_width;
                float bidx = (float)(5u * d.map_coord[0] / (f.map_half_wh[0] * 2u));
                float bidy = (float)(2u * d.map_coord[1] / (f.map_half_wh[1] * 2u));
                if (d.single_draw == 1u) {
                    const uint32_t map_height = 2u * f.map_half_wh[1];
                    const uint32_t map_id = map_id_m;
                    bidx = (float)(5u * (map_id / map_height) / (f.map_half_wh[0] * 2u));
                    bidy = (float)(2u * (map_id % map_height) / (f.map_half_wh[1] * 2u));
                }
                const float bx = nx - bidx * block_w, by = ny - bidy * block_w;
                float lz[3], pr[3], pl[3], pu[3], pd[3];
                sphere_point(block_w, bidx, bidy, bx, by, lz);
                const float R = f.sphere_radius;
                const float dt = DELTA * ymax;
                sphere_point(block_w, bidx, bidy, bx + dt, by, pr);
                sphere_point(block_w, bidx, bidy, bx - dt, by, pl);
                sphere_point(block_w, bidx, bidy, bx, by + dt, pu);
                sphere_point(block_w, bidx, bidy, bx, by - dt, pd);
                for (int k = 0; k < 3; k++) {
                    F[k] = (pr[k] * R - pl[k] * R) / (2.0f * dt);
                    F[3 + k] = (pu[k] * R - pd[k] * R) / (2.0f * dt);
                    F[6 + k] = lz[k];
                }
                const float z = c2;
                const float n0 = lz[0] * R, n1 = lz[1] * R, n2 = lz[2] * R;
                c0 = n0 + F[6] * z;
                c1 = n1 + F[7] * z;
                c2 = n2 + F[8] * z;
                mapped_z = n2;
            }
            if (f.use_clip == 1u && mapped_z < f.clip_height) break;
            // A5 :91-150
            float t_ratio = -1.0f;
            uint32_t higher_lod = 0u;
            if (d.changing == 1u) {
                float dx = c0 - f.cam_pos[0], dy = c1 - f.cam_pos[1], dz = c2 - f.cam_pos[2];
                float cam_dist = sqrtf((dx * dx + dy * dy) + dz * dz);
                if (d.single_draw == 1u) {
                    if (lod_id == 0u) higher_lod = 0u;
                    else if (lod_id == f.num_lod - 1u) higher_lod = lod_id - 1u;
                    else {
                        float d1 = f.transition_dist[(lod_id - 1u) & 15u];
                        float d2 = f.transition_dist[lod_id & 15u];
                        higher_lod = (cam_dist - d1 < d2 - cam_dist) ? lod_id - 1u : lod_id;
                    }
                } else {
                    higher_lod = (d.changing_to_lower == 1) ? d.tile_lod : d.tile_lod - 1u;
                }
                float td = f.transition_dist[higher_lod & 15u];
                float thw = f.transition_width_ratio * td;
                t_ratio = clampf((cam_dist - td) / thw + 0.5f, 0.0f, 1.0f);
                if ((lod_id == higher_lod + 1u && t_ratio == 0.0f) || (lod_id == higher_lod && t_ratio == 1.0f)) break;
            }
            float majx, majy, minx, miny, cr, cg, cb, ca, ndcx, ndcy, depth;
            float q[4];
            if (STRICT) {
                // A6 :152-167, operator by operator
                float cv[4];
                for (int rr = 0; rr < 4; rr++) cv[rr] = ((f.V[rr] * c0 + f.V[4 + rr] * c1) + f.V[8 + rr] * c2) + f.V[12 + rr] * 1.0f;
                for (int rr = 0; rr < 4; rr++)
                    q[rr] = ((f.GP[rr] * cv[0] + f.GP[4 + rr] * cv[1]) + f.GP[8 + rr] * cv[2]) + f.GP[12 + rr] * cv[3];
                const float clip = 1.2f * q[3];
                if (q[2] < -clip || q[0] < -clip || q[0] > clip || q[1] < -clip || q[1] > clip) break;
                // A7 :169-205
                float K[9];
                {
                    float a = half_decode(w1.x & 0xFFFFu), b = half_decode(w1.x >> 16);
                    float cc = half_decode(w1.y & 0xFFFFu), dd = half_decode(w1.y >> 16);
                    float e = half_decode(w1.z & 0xFFFFu), ff = half_decode(w1.z >> 16);
                    K[0] = a; K[1] = b; K[2] = cc; K[3] = b; K[4] = dd; K[5] = e; K[6] = cc; K[7] = e; K[8] = ff;
                }
                if (f.point_cloud_radius > 0.0f) {
                    float pr = f.point_cloud_radius;
                    if (f.draw_mode > 0u) pr *= ldexpf(1.0f, (int)d.tile_lod);
                    K[0] = pr; K[1] = 0; K[2] = 0; K[3] = 0; K[4] = pr; K[5] = 0; K[6] = 0; K[7] = 0; K[8] = pr;
                }
                if (f.surface_type > 0u) {          // Vrk = transform * Vrk * transpose(transform)
                    float FK[9], R[9];
                    for (int cc = 0; cc < 3; cc++)
                        for (int rr = 0; rr < 3; rr++)
                            FK[3 * cc + rr] = (F[rr] * K[3 * cc] + F[3 + rr] * K[3 * cc + 1]) + F[6 + rr] * K[3 * cc + 2];
                    for (int cc = 0; cc < 3; cc++)
                        for (int rr = 0; rr < 3; rr++)
                            R[3 * cc + rr] = (FK[rr] * F[cc] + FK[3 + rr] * F[3 + cc]) + FK[6 + rr] * F[6 + cc];
                    for (int k = 0; k < 9; k++) K[k] = R[k];
                }
                {                                   // scene_scale_mat * Vrk * transpose(scene_scale_mat): the full matrix products, zeros included
                    const float S[9] = {f.scene_scale[0], 0.0f, 0.0f, 0.0f, f.scene_scale[1], 0.0f, 0.0f, 0.0f, f.scene_scale[2]};
                    float SK[9], R[9];
                    for (int cc = 0; cc < 3; cc++)
                        for (int rr = 0; rr < 3; rr++)
                            SK[3 * cc + rr] = (S[rr] * K[3 * cc] + S[3 + rr] * K[3 * cc + 1]) + S[6 + rr] * K[3 * cc + 2];
                    for (int cc = 0; cc < 3; cc++)
                        for (int rr = 0; rr < 3; rr++)
                            R[3 * cc + rr] = (SK[rr] * S[cc] + SK[3 + rr] * S[3 + cc]) + SK[6 + rr] * S[6 + cc];
                    for (int k = 0; k < 9; k++) K[k] = R[k];
                }
                // A8 :207-258
                const float dd0 = c0 - f.cam_pos[0], dd1 = c1 - f.cam_pos[1], dd2 = c2 - f.cam_pos[2];
                float t[3];
                for (int rr = 0; rr < 3; rr++) t[rr] = (f.V[rr] * dd0 + f.V[4 + rr] * dd1) + f.V[8 + rr] * dd2;
                const float txtz = t[0] / t[2], tytz = t[1] / t[2];
                const float limx = 1.3f * f.htan[0], limy = 1.3f * f.htan[1];
                t[0] = clampf(txtz, -limx, limx) * t[2];
                t[1] = clampf(tytz, -limy, limy) * t[2];
                const float tz2 = t[2] * t[2];
                // J_T columns: (fx / tz, 0, -fx tx / tz^2), (0, fy / tz, -fy ty / tz^2), (0, 0, 0)
                const float JT[9] = {f.focal[0] / t[2], 0.0f, (-f.focal[0] * t[0]) / tz2,
                                     0.0f, f.focal[1] / t[2], (-f.focal[1] * t[1]) / tz2,
                                     0.0f, 0.0f, 0.0f};
                float Tm[9];                        // T = transpose(view3) * J_T
                for (int cc = 0; cc < 3; cc++)
                    for (int rr = 0; rr < 3; rr++)
                        Tm[3 * cc + rr] = (f.V[4 * rr + 0] * JT[3 * cc] + f.V[4 * rr + 1] * JT[3 * cc + 1]) + f.V[4 * rr + 2] * JT[3 * cc + 2];
                float Am[9], C2[9];                 // cov2d = transpose(T) * Vrk * T
                for (int cc = 0; cc < 3; cc++)
                    for (int rr = 0; rr < 3; rr++)
                        Am[3 * cc + rr] = (Tm[3 * rr + 0] * K[3 * cc] + Tm[3 * rr + 1] * K[3 * cc + 1]) + Tm[3 * rr + 2] * K[3 * cc + 2];
                for (int cc = 0; cc < 3; cc++)
                    for (int rr = 0; rr < 3; rr++)
                        C2[3 * cc + rr] = (Am[rr] * Tm[3 * cc] + Am[3 + rr] * Tm[3 * cc + 1]) + Am[6 + rr] * Tm[3 * cc + 2];
                const float c00 = C2[0], c01 = C2[1], c11 = C2[4];
                const float mid = 0.5f * (c00 + c11);
                const float hxx = 0.5f * (c00 - c11);
                const float radius = sqrtf(hxx * hxx + c01 * c01);
                const float l1 = mid + radius, l2 = mid - radius;
                if (l2 < 0.0f) break;
                const float vx = c01, vy = l1 - c00;
                const float vlen = sqrtf(vx * vx + vy * vy);
                const float ex = vx / vlen, ey = vy / vlen;
                const float smaj = fminf(sqrtf(2.0f * l1), 1024.0f);
                const float smin = fminf(sqrtf(2.0f * l2), 1024.0f);
                majx = smaj * ex; majy = smaj * ey;
                minx = smin * ey; miny = smin * -ex;
                // A9 :260-265, 402-410
                cr = (float)(w1.w & 0xFFu) / 255.0f;
                cg = (float)((w1.w >> 8) & 0xFFu) / 255.0f;
                cb = (float)((w1.w >> 16) & 0xFFu) / 255.0f;
                ca = (float)((w1.w >> 24) & 0xFFu) / 255.0f;
                if (FULL && f.draw_mode != 0u) debug_draw_color(f, d, u2f(w0.x), u2f(w0.y), lod_id, t_ratio, cr, cg, cb);   // :268-399
                if (d.changing == 1u) {
                    if (lod_id != higher_lod) ca = ca * t_ratio;
                    else ca = ca * (1.0f - t_ratio);
                }
                // A10 :415-419
                ndcx = q[0] / q[3]; ndcy = q[1] / q[3]; depth = q[2] / q[3];
                if (DEBUG) {
                    const float fade = clampf(q[2] / q[3] + 1.0f, 0.0f, 1.0f);
                    vout.ndc[0] = ndcx; vout.ndc[1] = ndcy; vout.depth = depth;
                    vout.major[0] = majx; vout.major[1] = majy; vout.minor[0] = minx; vout.minor[1] = miny;
                    vout.rgba[0] = cr * fade; vout.rgba[1] = cg * fade; vout.rgba[2] = cb * fade; vout.rgba[3] = ca * fade;
                }
            } else {
                // A6 :152-167 -- canonical sequence v2: dot products are fma chains, quotients are products
                // with one correctly rounded reciprocal; the CPU checker evaluates exactly the same operations
                float cv[4];
                for (int rr = 0; rr < 4; rr++) cv[rr] = fmaf(f.V[8 + rr], c2, fmaf(f.V[4 + rr], c1, f.V[rr] * c0)) + f.V[12 + rr];
                for (int rr = 0; rr < 4; rr++)
                    q[rr] = fmaf(f.GP[12 + rr], cv[3], fmaf(f.GP[8 + rr], cv[2], fmaf(f.GP[4 + rr], cv[1], f.GP[rr] * cv[0])));
                float clip = 1.2f * q[3];
                if (q[2] < -clip || q[0] < -clip || q[0] > clip || q[1] < -clip || q[1] > clip) break;
                if (GSWT_ABL(f, 16)) break;               // ablation: stop after the frustum cull
                // A7 :169-205
                float K[9];
                {
                    float a = half_decode(w1.x & 0xFFFFu), b = half_decode(w1.x >> 16);
                    float cc = half_decode(w1.y & 0xFFFFu), dd = half_decode(w1.y >> 16);
                    float e = half_decode(w1.z & 0xFFFFu), ff = half_decode(w1.z >> 16);
                    K[0] = a; K[1] = b; K[2] = cc; K[3] = b; K[4] = dd; K[5] = e; K[6] = cc; K[7] = e; K[8] = ff;
                }
                if (f.point_cloud_radius > 0.0f) {
                    float pr = f.point_cloud_radius;
                    if (f.draw_mode > 0u) pr *= ldexpf(1.0f, (int)d.tile_lod);
                    K[0] = pr; K[1] = 0; K[2] = 0; K[3] = 0; K[4] = pr; K[5] = 0; K[6] = 0; K[7] = 0; K[8] = pr;
                }
                if (f.surface_type > 0u) {
                    float FK[9], R[9];
                    for (int cc = 0; cc < 3; cc++)
                        for (int rr = 0; rr < 3; rr++)
                            FK[3 * cc + rr] = fmaf(F[6 + rr], K[3 * cc + 2], fmaf(F[3 + rr], K[3 * cc + 1], F[rr] * K[3 * cc]));
                    for (int cc = 0; cc < 3; cc++)
                        for (int rr = 0; rr < 3; rr++)
                            R[3 * cc + rr] = fmaf(FK[6 + rr], F[6 + cc], fmaf(FK[3 + rr], F[3 + cc], FK[rr] * F[cc]));
                    for (int k = 0; k < 9; k++) K[k] = R[k];
                }
                // scene_scale_mat * Vrk * transpose(scene_scale_mat): a product with 1.0 is exact, so the usual unit scale skips 18 multiplications
                // (uniform branch; same bits)
                if (f.scene_scale[0] != 1.0f || f.scene_scale[1] != 1.0f || f.scene_scale[2] != 1.0f)
                    for (int cc = 0; cc < 3; cc++)
                        for (int rr = 0; rr < 3; rr++) K[3 * cc + rr] = (f.scene_scale[rr] * K[3 * cc + rr]) * f.scene_scale[cc];
                // A8 :207-258
                float d0 = c0 - f.cam_pos[0], d1 = c1 - f.cam_pos[1], d2 = c2 - f.cam_pos[2];
                float t[3];
                for (int rr = 0; rr < 3; rr++) t[rr] = fmaf(f.V[8 + rr], d2, fmaf(f.V[4 + rr], d1, f.V[rr] * d0));
                const float rz = GSWT_RCP(t[2]);
                float txtz = t[0] * rz, tytz = t[1] * rz;
                float limx = 1.3f * f.htan[0], limy = 1.3f * f.htan[1];
                t[0] = clampf(txtz, -limx, limx) * t[2];
                t[1] = clampf(tytz, -limy, limy) * t[2];
                const float rz2 = rz * rz;
                float j00 = f.focal[0] * rz, j02 = -((f.focal[0] * t[0]) * rz2);
                float j11 = f.focal[1] * rz, j12 = -((f.focal[1] * t[1]) * rz2);
                float T0[3], T1[3];
                for (int rr = 0; rr < 3; rr++) {
                    T0[rr] = fmaf(f.V[4 * rr + 2], j02, f.V[4 * rr + 0] * j00);
                    T1[rr] = fmaf(f.V[4 * rr + 2], j12, f.V[4 * rr + 1] * j11);
                }
                float A0[3], A1[3];
                for (int k = 0; k < 3; k++) {
                    A0[k] = fmaf(T0[2], K[3 * k + 2], fmaf(T0[1], K[3 * k + 1], T0[0] * K[3 * k]));
                    A1[k] = fmaf(T1[2], K[3 * k + 2], fmaf(T1[1], K[3 * k + 1], T1[0] * K[3 * k]));
                }
                float c00 = fmaf(A0[2], T0[2], fmaf(A0[1], T0[1], A0[0] * T0[0]));
                float c01 = fmaf(A1[2], T0[2], fmaf(A1[1], T0[1], A1[0] * T0[0]));
                float c11 = fmaf(A1[2], T1[2], fmaf(A1[1], T1[1], A1[0] * T1[0]));
                float mid = 0.5f * (c00 + c11);
                float hxx = 0.5f * (c00 - c11);
                float radius = GSWT_SQRT(fmaf(hxx, hxx, c01 * c01));
                float l1 = mid + radius, l2 = mid - radius;
                if (l2 < 0.0f) break;
                float vx = c01, vy = l1 - c00;
                float vlen = GSWT_SQRT(fmaf(vx, vx, vy * vy));
                const float rv = GSWT_RCP(vlen);
                float ex = vx * rv, ey = vy * rv;
                float smaj = fminf(GSWT_SQRT(2.0f * l1), 1024.0f);
                float smin = fminf(GSWT_SQRT(2.0f * l2), 1024.0f);
                majx = smaj * ex; majy = smaj * ey;
                minx = smin * ey; miny = smin * -ex;
                // A9 :260-265, 402-410 (byte / 255 as byte * fl(1 / 255))
                const float k255 = 1.0f / 255.0f;
                cr = (float)(w1.w & 0xFFu) * k255;
                cg = (float)((w1.w >> 8) & 0xFFu) * k255;
                cb = (float)((w1.w >> 16) & 0xFFu) * k255;
                ca = (float)((w1.w >> 24) & 0xFFu) * k255;
                if (FULL && f.draw_mode != 0u) debug_draw_color(f, d, u2f(w0.x), u2f(w0.y), lod_id, t_ratio, cr, cg, cb);   // :268-399
                if (d.changing == 1u) {
                    if (lod_id != higher_lod) ca = ca * t_ratio;
                    else ca = ca * (1.0f - t_ratio);
                }
                // rgba *= clamp(z/w + 1, 0, 1): identically 1 for 0 <= z/w, kept for the debug output only
                // A10 :415-419
                const float rq = GSWT_RCP(q[3]);
                ndcx = q[0] * rq; ndcy = q[1] * rq; depth = q[2] * rq;
                if (DEBUG) {
                    float fade = clampf(fmaf(q[2], rq, 1.0f), 0.0f, 1.0f);
                    vout.ndc[0] = ndcx; vout.ndc[1] = ndcy; vout.depth = depth;
                    vout.major[0] = majx; vout.major[1] = majy; vout.minor[0] = minx; vout.minor[1] = miny;
                    vout.rgba[0] = cr * fade; vout.rgba[1] = cg * fade; vout.rgba[2] = cb * fade; vout.rgba[3] = ca * fade;
                }
            }
            if (!(depth >= 0.0f && depth <= 1.0f)) break;
            // Fragment setup F1, F2: pixel-space centre and inverse affine map
            float cxp = fmaf(0.5f, ndcx, 0.5f) * f.W;
            float cyp = fmaf(-0.5f, ndcy, 0.5f) * f.H;
            float hs = 0.5f * f.splat_scale;
            float ux = hs * majx, uy = -(hs * majy);
            float wx = hs * minx, wy = -(hs * miny);
            float uu = fmaf(uy, uy, ux * ux);
            float ww = fmaf(wy, wy, wx * wx);
            if (!((uu > 0.0f) && (ww > 0.0f) && (uu < __builtin_inff()) && (ww < __builtin_inff()))) break;
            if (DEBUG) vout.visible = 1;
            visible = true;
            // depth_compare Less against the 1.0 clear when no proxy depth is bound (renderer.rs:182,436)
            if (!f.has_depth && !(depth < 1.0f)) { visible = false; break; }
            const float ruu = GSWT_RCP(uu), rww = GSWT_RCP(ww);
            const float r_iux = ux * ruu, r_iuy = uy * ruu, r_ivx = wx * rww, r_ivy = wy * rww;
            // half extents of |p| <= 2, inflated by 1e-5 relative + 1e-3 px (conservative under f32 rounding)
            float hx = fmaf(2.0f * GSWT_SQRT(fmaf(wx, wx, ux * ux)), 1.00001f, 0.001f);
            float hy = fmaf(2.0f * GSWT_SQRT(fmaf(wy, wy, uy * uy)), 1.00001f, 0.001f);
            // pixels whose CENTRE lies inside the box: x in [ceil(c - h - 0.5), floor(c + h - 0.5)]
            float fx0 = ceilf(cxp - hx - 0.5f), fx1 = floorf(cxp + hx - 0.5f);
            float fy0 = ceilf(cyp - hy - 0.5f), fy1 = floorf(cyp + hy - 0.5f);
            if (fx1 >= fx0 && fy1 >= fy0 && fx1 >= 0.0f && fy1 >= 0.0f && fx0 <= f.W - 1.0f && fy0 <= f.H - 1.0f) {
                int x0 = fx0 < 0.0f ? 0 : (int)fx0, x1 = fx1 > f.W - 1.0f ? f.width - 1 : (int)fx1;
                int y0 = fy0 < 0.0f ? 0 : (int)fy0, y1 = fy1 > f.H - 1.0f ? f.height - 1 : (int)fy1;
                int tx0 = x0 >> 4, tx1 = x1 >> 4, ty0 = y0 >> 4, ty1 = y1 >> 4;
                tx0 = max(tx0, f.col0); tx1 = min(tx1, f.col1 - 1);          // column band of this ctx (the whole frame when off)
                int rows = owned_rows(ty0, ty1, f.shard_index, f.shard_count);
                count = tx1 >= tx0 ? (uint32_t)((tx1 - tx0 + 1) * rows) : 0u;
                if (GSWT_ABL(f, 8)) count = 0;        // ablation: no record / rect stores, no pairs
                if (count) {
                    my_rect = make_uint2((uint32_t)tx0 | ((uint32_t)tx1 << 16), (uint32_t)ty0 | ((uint32_t)ty1 << 16));
                    // 32-byte record: the inverse map, the centre, alpha and the packed colour (unpacked by the compositor's blend);
                    // the pixel half extents are re-derived from the inverse map by the compositor's staging lane, the depth
                    // goes to a side array that only depth-tested / depth-ordered frames read
                    Rec* dst = recs + slot;
                    reinterpret_cast<float4*>(dst)[0] = make_float4(r_iux, r_iuy, r_ivx, r_ivy);
                    // (the centre travels in NDC: the compositor takes its offset from a tile origin with one rounding, F3 of DESIGN.md section 4)
                    reinterpret_cast<float4*>(dst)[1] = make_float4(ndcx, ndcy, ca, __uint_as_float(w1.w));
                    if (depths) depths[slot] = depth;
                    if (FULL && f.draw_mode != 0u) col_f[slot] = make_float4(cr, cg, cb, 0.0f);   // debug colours are not bytes
                }
            }
        } while (0);
    }
    if (DEBUG && in_list) dbg[d.entry_base + (d.count - 1u - r)] = vout;
    GSWT_TR(4, GSWT_NOW())

    // workgroup sums: pairs and visible splats
    const uint32_t wsum = wave_sum(count), wvis = (uint32_t)__popcll(ballot64(visible));
    // (Round 4, measured and reverted: no barrier here -- every wave adding its own sums with fire-and-forget atomics (12 per workgroup
    // instead of 2 + a store, the rects stored whether or not the chunk emits pairs): k_project 57 -> 183 us at c3, 379 -> 561 at c5.  The
    // device-scope atomics of a kernel retire at ~1 per ns chip-wide whatever their addresses: 107 k more of them cost 125 us.)
    const uint32_t w0 = 4u * half;             // first wave of this half in the workgroup's tables
    if ((tid & 63u) == 0) { s_wsum[w0 + (tid >> 6)] = wsum; s_wvis[w0 + (tid >> 6)] = wvis; }
    __syncthreads();
    GSWT_TR(5, GSWT_NOW())
    if (!active) return;
    const uint32_t bsum = s_wsum[w0] + s_wsum[w0 + 1u] + s_wsum[w0 + 2u] + s_wsum[w0 + 3u];
#ifdef GSWT_TRACE
    { unsigned hwid, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      GSWT_TR(7, (unsigned long long)hwid | ((unsigned long long)(xcc & 0xFFu) << 32) | ((unsigned long long)(bsum ? 1u : 0u) << 63)) }
#endif
    // k_emit reads the rects of a chunk only when the chunk has pairs
    if (bsum) rects[slot] = my_rect;
    if (tid == 0) {
        block_sums[cid] = bsum;
        // two-level sums, spread over n_chunks / 256 addresses (a single hot counter serialises the whole grid);
        // k_totals folds them into counters[0] (visible splats) and counters[1] (pairs)
        if (bsum) atomicAdd(&super_sums[(cid >> 8) * kSuperStride], bsum);
        uint32_t v = s_wvis[w0] + s_wvis[w0 + 1u] + s_wvis[w0 + 2u] + s_wvis[w0 + 3u];
        if (v) atomicAdd(&super_sums[(n_super + (cid >> 8)) * kSuperStride], v);
    }
    GSWT_TR(6, GSWT_NOW())
}

__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w, uint32_t* total);

// counters[0] = visible splats, counters[1] = pairs; super_excl[j] = pairs of all chunks before super-group j (exclusive
// scan of the pair half of super_sums), so that k_emit reads one word instead of summing up to n_chunks / 256 of them
// (c5: 1 430 per workgroup).  Single workgroup.  (Folding it into k_emit -- every workgroup sums the super-group counts
// in front of it, workgroup 0 leaves the totals -- was measured: k_emit +6 us for the 4.6 us saved, 4015 -> 3881 frames/s.)
__global__ __launch_bounds__(256) void k_totals(const uint32_t* __restrict__ super_sums, uint32_t n_super,
                                                unsigned long long* __restrict__ counters, uint32_t* __restrict__ super_excl, uint32_t pair_cap,
                                                uint32_t* __restrict__ live_cnt, uint32_t launched_per_xcd)
{
    // k_project is done with this frame's live-chunk counts: cleared here for the slot's next frame (k_cull both clears
    // buffers and adds to these counters, so it cannot clear them itself)
    if (live_cnt && threadIdx.x < 64u) {
        // the longest of the eight live lists: the host sizes the next frames' launch grids by it (counters[4]); a frame whose grid was cut
        // shorter than its own lists has chunks that nobody projected: flagged, the host re-runs it with the full grid (gswt_api.hip, live_hint)
        uint32_t v = threadIdx.x < 8u ? live_cnt[threadIdx.x * kSuperStride] : 0u;
        if (threadIdx.x < 8u) {
            live_cnt[(8u + threadIdx.x) * kSuperStride] = v;      // (k_emit runs behind this kernel and walks the same table)
            live_cnt[threadIdx.x * kSuperStride] = 0u;
        }
        for (int o = 4; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o, 64));
        if (threadIdx.x == 0u) {
            counters[4] = v;
            if (v > launched_per_xcd) counters[3] = 1ull;
        }
    }
    __shared__ unsigned long long s_v[4];
    __shared__ uint32_t s_w[4];
    unsigned long long v = 0;
    // the frame's pair count is carried in 64 bits (a super-group sum is a u32 of at most 65 536 slots x #screen tiles); the
    // prefix words k_emit reads stay 32-bit: once the running total passes the pair capacity the frame is flagged as
    // overflowed right here, and nothing downstream of k_emit consumes pairs of a flagged frame
    unsigned long long carry = 0;
    for (uint32_t base = 0; base < n_super; base += 1024u) {
        const uint32_t i = base + threadIdx.x * 4u;
        uint32_t p[4];
        unsigned long long sum = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t idx = min(i + (uint32_t)k, n_super - 1u);              // clamped, unmasked loads
            const uint32_t pv = super_sums[idx * kSuperStride], vv = super_sums[(n_super + idx) * kSuperStride];
            p[k] = i + k < n_super ? pv : 0u;
            v += i + k < n_super ? vv : 0u;
            sum += p[k];
        }
        // block scan of the low words; the high part of the block total is recovered from a 64-bit reduction
        uint32_t tot32;
        uint32_t ex = block_excl_scan((uint32_t)sum, s_w, &tot32) + (uint32_t)carry;
#pragma unroll
        for (int k = 0; k < 4; k++) { if (i + k < n_super) super_excl[i + k] = ex; ex += p[k]; }
        unsigned long long t64 = sum;
        for (int off = 32; off > 0; off >>= 1) t64 += __shfl_down(t64, off, 64);
        __syncthreads();
        if ((threadIdx.x & 63u) == 0) s_v[threadIdx.x >> 6] = t64;
        __syncthreads();
        carry += s_v[0] + s_v[1] + s_v[2] + s_v[3];
        __syncthreads();
    }
    v = wave_sum(v);
    if ((threadIdx.x & 63u) == 0) s_v[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        counters[1] = carry; counters[0] = s_v[0] + s_v[1] + s_v[2] + s_v[3];
        if (carry > (unsigned long long)pair_cap) counters[3] = 1ull;       // pair buffers too small: the host re-runs the frame
    }
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix,
// *total = block sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_w /*[4]*/, uint32_t* total)
{
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63u) s_w[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t i = 0; i < w; i++) base += s_w[i];
    *total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    return base + inc - v;
}

// ------------------------------------------------------------------------------------
// k_emit: one thread per slot; pairs written at block_off[blk] + local exclusive prefix,
// so the pair array is in composite (slot) order before the stable tile sort.
// key = screen tile index local to the shard; val = slot.
// ------------------------------------------------------------------------------------
// A workgroup takes kEmitGroup consecutive chunks: most chunks emit nothing (culled draws, LOD rings) and a workgroup
// that only learns so from a load costs a full memory round trip per chunk -- with one chunk per workgroup the kernel ran
// as (chunks / 2048 resident workgroups) x ~1.07 us: 22 us on c3, 190 us on c5's 365 k chunks.
// Measured and dropped: the four chunks side by side in a 1024-thread workgroup with a single barrier (32 us at c3 -- four times
// the waves for the empty groups); per-slot pair offsets left by k_project so that no per-chunk scan is needed here (18 us, +1.3 us
// there, no gain); groups of 8 / 16 (serialise the live chunks: 26 / 35 us at c3); launching over k_cull's table of LIVE chunks
// (what k_project does: 18 k of c3's 39 k chunks, 76 k of c5's 366 k; groups of 1 / 2 / 4 table entries): one more dependent load in
// front of everything, 24 / 22 / 24 us at c3 and 223 / 170 / 143 us at c5 against 23 / 130 for four consecutive chunks (stage
// events, gpurun_out s6) -- the workgroups of empty groups are not what the kernel's time is made of.
// (Round 4: nor the splats that cover many screen tiles -- a slot writing at most eight of its pairs itself and the workgroup the rest, 256
// consecutive pairs per step, from a queue in LDS: c3 18.7 us either way, c5 165 against 124, the dense c3d 277 against 46.)
// (Round 4, c5 again -- 366 k chunks, ~40 k with pairs: sixteen chunks per workgroup, the sixteen pair counts first and then each chunk that has
// pairs with loads of its own: 180 us against 129, c3 36.7 against 18.7.)
// ... nor the rects read for chunks that turn out to have no pairs (round 4: the pair counts first, everything else only for the chunks
// that have any -- c5 reads 750 MB of rects per frame for ~40 k of 366 k chunks: 131 us against 129, c3 18.5 either way);
// and neither is the barrier-separated walk over a workgroup's live chunks: ONE WAVE PER CHUNK (lane l owns slots 4 l .. 4 l + 3,
// offsets from one wave scan, the chunk's base from one wave reduction over its super-group's sums, no barrier, no LDS, the four
// chunks of a workgroup side by side) ran in 20.6 us at c3 and 146 us at c5 against 19.9 / 127.
constexpr uint32_t kEmitGroup = 4;

// block-wide (256 threads): exclusive scan of v and, in the same barrier pair, the sum of r
__device__ __forceinline__ uint32_t block_scan_and_sum(uint32_t v, uint32_t r, uint32_t* s_w /*[8]*/, uint32_t* v_total, uint32_t* r_total)
{
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    r = wave_sum(r);
    if (lane == 63u) s_w[w] = inc;
    if (lane == 0u) s_w[4u + w] = r;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t i = 0; i < w; i++) base += s_w[i];
    *v_total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    *r_total = s_w[4] + s_w[5] + s_w[6] + s_w[7];
    __syncthreads();
    return base + inc - v;
}

// DEPTH (GSWT_ORDER_DEPTH, north_star's "global radix depth sort"): every pair also gets its splat's depth bits as a second key (dkeys), and
// the smallest / largest depth key of the frame is left in krange.  The pair list is then sorted stably on the depth bits in use (the tile id
// travels as the sort's payload) and after that on the tile bits as always: inside a screen tile the pairs end up in true depth order, equal
// depths in composite order.  Round 4's first version sorted the emitting SPLATS by depth and emitted the pairs in that order (compaction,
// three passes with the tile rect as payload, per-group pair counts, a scan, the emission: five more kernels and 111 us per c3 frame);
// keying the PAIRS costs one more 4-byte word per pair in k_emit and in the depth passes and needs none of that.
// TAB (the default since the end of round 4; GSWT_EMIT_TAB=0: four consecutive chunks of the frame): the workgroup takes four entries of k_cull's table of live chunks (k_project's launch table: one XCD's list, so the chunks are not
// neighbours in slot order -- they need not be: a chunk's first pair comes from the prefix arrays) instead of four consecutive chunks of
// the frame.  At c3 73 % of the chunks are not live, and 39 % of this kernel's workgroup-time was workgroups that loaded four zeros and left
// (tools/emit_trace.py); the price is one more dependent load in front of everything.  One frame at a time the kernel takes what it took (c3 18.9
// against 18.7 us, c5 125 against 124 -- as in round 2, when the table held 46 % of the chunks), but with frames in flight the workgroups that
// are no longer launched leave their slots to the other frames' kernels: c5 fly path 719-720 against 692-698 frames/s, c3 the same.
template <bool DEPTH, bool TAB>
__global__ __launch_bounds__(256) void k_emit(const Frame f, const uint2* __restrict__ rects,
                                              const uint32_t* __restrict__ block_sums, const uint32_t* __restrict__ super_excl,
                                              uint32_t n_chunks, uint32_t pair_cap, unsigned long long* __restrict__ counters,
                                              uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                              const float* __restrict__ depths, uint32_t* __restrict__ dkeys, uint32_t* __restrict__ krange,
                                              const uint32_t* __restrict__ live_cnt2, const uint32_t* __restrict__ live_cid)
{
    __shared__ uint32_t s_w[8];
    __shared__ uint32_t s_mn[DEPTH ? 4 : 1], s_mx[DEPTH ? 4 : 1];
#ifdef GSWT_TRACE
    // phase stamps of k_emit (tools/emit_trace.py), rows 81920 + workgroup: [0] entry, [1] every load has arrived, [2] exit, [3] chunks with pairs, [4] pairs
    const bool tr_on = threadIdx.x == 0 && blockIdx.x < 8192u;
    const uint32_t tr_item = 81920u + blockIdx.x;
    uint32_t tr_live = 0, tr_pairs = 0;
    GSWT_TR(0, GSWT_NOW())
    GSWT_TR(2, 0ull)
#endif
    uint32_t cid[kEmitGroup], sums[kEmitGroup], bs[kEmitGroup], sbase[kEmitGroup];
    uint2 rcs[kEmitGroup];
    float dep[kEmitGroup];
    uint32_t kmn = 0xFFFFFFFFu, kmx = 0u;
    uint32_t n_mine;
    if (TAB) {
        const uint32_t x = blockIdx.x & 7u, k0 = (blockIdx.x >> 3) * kEmitGroup;
        const uint32_t n_live = live_cnt2[x * kSuperStride];
        if (k0 >= n_live) return;
        n_mine = min(kEmitGroup, n_live - k0);
#pragma unroll
        for (uint32_t k = 0; k < kEmitGroup; k++) cid[k] = min(live_cid[(size_t)min(k0 + k, n_live - 1u) * 8u + x], n_chunks - 1u);
    } else {
        const uint32_t c0 = blockIdx.x * kEmitGroup;
        n_mine = min(kEmitGroup, n_chunks - c0);
#pragma unroll
        for (uint32_t k = 0; k < kEmitGroup; k++) cid[k] = min(c0 + k, n_chunks - 1u);
    }
    // All loads are issued together with clamped indices (masked loads are waited for one at a time): the chunks' pair counts, the
    // sums of the chunks in front of each inside its super-group, the super-group prefixes and the tile rects of this thread's slot
    // in every chunk.
#pragma unroll
    for (uint32_t k = 0; k < kEmitGroup; k++) {
        sums[k] = block_sums[cid[k]];
        bs[k] = block_sums[min((cid[k] & ~255u) + threadIdx.x, cid[k])];
        sbase[k] = super_excl[cid[k] >> 8];                          // pairs of all chunks before the chunk's super-group (k_totals)
        rcs[k] = rects[(size_t)cid[k] * 256u + threadIdx.x];         // only meaningful when sums[k] != 0 (k_project wrote it then)
        dep[k] = DEPTH ? depths[(size_t)cid[k] * 256u + threadIdx.x] : 0.0f;      // likewise (k_project stores a slot's depth beside its record)
    }
    uint32_t any = 0;
#pragma unroll
    for (uint32_t k = 0; k < kEmitGroup; k++) { if (k >= n_mine) sums[k] = 0u; any |= sums[k]; }
#ifdef GSWT_TRACE
    { unsigned long long sink = sums[0] + bs[0] + sbase[3] + rcs[0].x + rcs[3].y; asm volatile("" :: "v"(sink)); }
    GSWT_TR(1, GSWT_NOW())
    for (uint32_t k = 0; k < kEmitGroup; k++) { tr_live += sums[k] ? 1u : 0u; tr_pairs += sums[k]; }
    GSWT_TR(3, tr_live)
    GSWT_TR(4, tr_pairs)
    if (any == 0u) { GSWT_TR(2, GSWT_NOW()) }
#endif
    if (any == 0u) return;
    const int sc = f.shard_count <= 1 ? 1 : f.shard_count;
#pragma unroll
    for (uint32_t k = 0; k < kEmitGroup; k++) {
        if (sums[k] == 0u) continue;                                 // workgroup-uniform
        const uint2 rc = rcs[k];
        const uint32_t slot = cid[k] * 256u + threadIdx.x;
        const int tx0 = rc.x & 0xFFFFu, tx1 = rc.x >> 16, ty0 = rc.y & 0xFFFFu, ty1 = rc.y >> 16;
        uint32_t count = 0;
        if (tx1 >= tx0) count = (uint32_t)((tx1 - tx0 + 1) * owned_rows(ty0, ty1, f.shard_index, f.shard_count));
        // first pair of the chunk = pairs of the earlier super-groups + of the earlier chunks of its own group; one barrier pair
        // yields that sum and the exclusive scan of this chunk's per-slot pair counts
        uint32_t tot, chunk_base;
        uint32_t off = block_scan_and_sum(count, (cid[k] & ~255u) + threadIdx.x < cid[k] ? bs[k] : 0u, s_w, &tot, &chunk_base);
        chunk_base += sbase[k];
        off += chunk_base;
        if ((unsigned long long)chunk_base + tot > (unsigned long long)pair_cap) {   // pair buffers too small: host re-runs the frame
            if (threadIdx.x == 0) atomicOr(&counters[3], 1ull);
            return;
        }
        if (count == 0) continue;
        const uint32_t dkey = __float_as_uint(dep[k]);               // depth in [0, 1]: bit order = value order
        if (DEPTH) { kmn = min(kmn, dkey); kmx = max(kmx, dkey); }
        if (sc > 1) {                                                // row shards: this rank's rows of the rect, numbered locally
            for (int ty = ty0 + ((f.shard_index - ty0 % sc) + sc) % sc; ty <= ty1; ty += sc) {
                const uint32_t row = (uint32_t)(ty / sc) * (uint32_t)f.tiles_x;
                for (int tx = tx0; tx <= tx1; tx++) {
                    keys[off] = row + (uint32_t)(tx - f.col0);
                    vals[off] = slot;
                    if (DEPTH) dkeys[off] = dkey;
                    off++;
                }
            }
        } else {
            // the usual frame: a loop of its own, so that the ~25-instruction integer division by the shard count is not part of every row of
            // every splat (as `ty / sc` with sc == 1 it was: the compiler cannot know the uniform's value)
            for (int ty = ty0; ty <= ty1; ty++) {
                const uint32_t row = (uint32_t)ty * (uint32_t)f.tiles_x;
                for (int tx = tx0; tx <= tx1; tx++) {
                    keys[off] = row + (uint32_t)(tx - f.col0);
                    vals[off] = slot;
                    if (DEPTH) dkeys[off] = dkey;
                    off++;
                }
            }
        }
    }
    GSWT_TR(2, GSWT_NOW())
    if (DEPTH && krange) {                                      // (the tile-local depth sort takes each tile's own range: no krange)
        // key range of the frame: one guarded atomic pair per workgroup.  (One pair per WAVE on the two words is tens of thousands of
        // atomics on two addresses, which the memory side serialises at ~8 ns each.)  The words only grow ([0] holds ~min), so a stale read
        // can only cause a redundant atomic, never a missed one.
        kmn = wave_min(kmn); kmx = wave_max(kmx);
        if ((threadIdx.x & 63u) == 0u) { s_mn[DEPTH ? threadIdx.x >> 6 : 0u] = kmn; s_mx[DEPTH ? threadIdx.x >> 6 : 0u] = kmx; }
        __syncthreads();
        if (threadIdx.x == 0u) {
            kmn = min(min(s_mn[0], s_mn[DEPTH ? 1 : 0]), min(s_mn[DEPTH ? 2 : 0], s_mn[DEPTH ? 3 : 0]));
            kmx = max(max(s_mx[0], s_mx[DEPTH ? 1 : 0]), max(s_mx[DEPTH ? 2 : 0], s_mx[DEPTH ? 3 : 0]));
            if (kmn <= kmx) {
                if (~kmn > __builtin_nontemporal_load(&krange[0])) atomicMax(&krange[0], ~kmn);                     // (load_krange)
                if (kmx > __builtin_nontemporal_load(&krange[1])) atomicMax(&krange[1], kmx);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Stable LSD radix sort pass on `nbits` (<= 8) key bits starting at `shift`.
// 4096 items per workgroup; each of the 4 waves owns 1024 consecutive items and ranks
// them 64 at a time with ballot match-any, so in-wave order == memory order.
// ------------------------------------------------------------------------------------
// pair count as seen by the kernels downstream of k_emit: never more than the buffers hold
// (n_ptr = &counters[1]; counters[3] = overflow flag set by k_emit).  On overflow part of the pair buffer is
// unwritten, so the whole downstream chain processes nothing and the host re-runs the frame with larger buffers.
__device__ __forceinline__ uint32_t clamped_count(const unsigned long long* n_ptr, uint32_t n_cap)
{
    const unsigned long long n = n_ptr[0];
    if (n_ptr[2] != 0ull || n > (unsigned long long)n_cap) return 0u;
    return (uint32_t)n;
}

// A sort workgroup takes 4096 consecutive items whatever its width: THREADS x (4096 / THREADS) per thread, each wave ranking
// its share 64 at a time.  Sweeps, frames/s with two frames in flight (bench.py, 300 steps): c3 (2.66 M pairs) 256 threads
// 3733, 512 3925, 1024 3899; c3h 3105 / 3578 / 3589; c5 (21 M pairs) 537.5 / 531.6 / 506.2.  The sort stage alone at c3:
// 77 / 69 / 76 us (other splits of the block: 1024x8 78, 512x16 82, 1024x2 95, 512x4 84).  launch_sort picks 512 threads up
// to kSortWideMax pairs of capacity and 256 above.
constexpr int kSortBlock = 4096;
constexpr uint32_t kSortWideMax = 0xFFFFFFFFu;  // pair capacities up to this use 512-thread workgroups (every size since k_radix_supscan: the 256-thread build only won at c5 while each workgroup summed ~100 group rows)

// Per pass: k_radix_hist leaves, for every digit d, the per-workgroup counts ghist[d][blk], the sums over
// groups of 32 workgroups gsup[d][blk >> 5] and the digit totals gtot[d] (integer atomics, spread over
// 256 x nblk/32 addresses).  k_radix_scatter derives its global offsets from those directly -- digit d's
// base = sum of gtot[< d] + gsup[d][< blk >> 5] + ghist[d][same group, < blk] -- so no scan kernel runs.
constexpr uint32_t kSupShift = 5;
constexpr uint32_t kSupDirect = 32;             // up to this many groups the scatter kernel reads every group row instead of digit totals

// Key range of the depth sort as k_emit<DEPTH> leaves it: [0] = ~smallest key, [1] = largest (both through atomicMax, so the frame's zeroed
// scratch words are the neutral start); no keys at all reads as (0, 0).
__device__ __forceinline__ void load_krange(const uint32_t* __restrict__ krange, uint32_t& kmin, uint32_t& kmax)
{
    kmin = ~krange[0]; kmax = krange[1];
    if (kmax < kmin) { kmin = 0u; kmax = 0u; }
}
// Number of radix passes the keys in [kmin, kmax] need (see k_radix_hist): pass p covers bits 8p .. 8p+7 of key - kmin.  The host
// launches as many passes as the previous frames needed (three at c3: visible depths of one frame span ~2^21 ulps); a frame that needs
// more is flagged by k_items and re-run, like a capacity overflow; one that needs fewer is sorted all the same (the extra passes see one
// digit and move nothing) and the host gives the pass back after a while.
__device__ __forceinline__ uint32_t sort_passes_needed(const uint32_t* __restrict__ krange)
{
    uint32_t kmin, kmax;
    load_krange(krange, kmin, kmax);
    const uint32_t span = kmax - kmin;
    uint32_t p = 1;                                     // the first pass always runs
    for (uint32_t sh = 8; sh < 32u; sh += 8u) if ((span >> sh) != 0u) p++;
    return p;
}

// Lanes of the wave that hold the same digit as this one (match-any on `nbits` bits), restricted to valid lanes.
// (Round 4, measured and dropped: a fast path for rounds whose 64 items share one digit -- one broadcast + one ballot instead of nbits
// ballots, taken in the passes after the first, where the pairs of one screen tile are neighbours: the last pass of the c3 pair sort went
// from 18.5 to 21.8 us.  Too few rounds are uniform, and the mixed ones pay the two extra wave-wide operations.)
// One step per key bit, six vector instructions each: the lane's bit sign-extended (v_bfe_i32: -1 / 0), the ballot of the set bits
// (v_cmp straight into a scalar pair), and per half of the mask peers &= ~(ballot ^ own) (v_xnor, v_and).  Unrolled, so no loop counter
// either.  (Until round 4 the loop was rolled and went through __ballot(int): ten vector instructions, two s_nop and the loop's scalar
// bookkeeping per bit -- about twice the issue slots.)
__device__ __forceinline__ unsigned long long match_digit(uint32_t dgt, bool valid, uint32_t nbits)
{
    const unsigned long long vm = __builtin_amdgcn_ballot_w64(valid);
    uint32_t plo = (uint32_t)vm, phi = (uint32_t)(vm >> 32);
    // (all eight steps whatever the digit's width: the callers mask the digit, so a bit above it is clear in every lane and its step
    // changes nothing -- five instructions, against a branch per bit on a count the compiler keeps as a lane mask)
    (void)nbits;
#pragma unroll
    for (uint32_t b = 0; b < 8u; b++) {
        const uint32_t own = (uint32_t)__builtin_amdgcn_sbfe((int)dgt, b, 1u);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(own != 0u);
        plo &= ~((uint32_t)m ^ own);
        phi &= ~((uint32_t)(m >> 32) ^ own);
    }
    return ((unsigned long long)phi << 32) | plo;
}

template <int kSortThreads>
__global__ __launch_bounds__(kSortThreads) void k_radix_hist(const uint32_t* __restrict__ keys, const unsigned long long* __restrict__ n_ptr,
                                                    uint32_t n_cap, uint32_t shift, uint32_t mask, uint32_t nbits, uint32_t* __restrict__ ghist,
                                                    uint32_t* __restrict__ gsup, uint32_t* __restrict__ gtot, uint32_t nblk,
                                                    uint32_t nsup, const uint32_t* __restrict__ krange)
{
    constexpr int kSortItems = kSortBlock / kSortThreads;
    // krange (the depth passes only): the frame's smallest / largest depth key.  Digits are taken from key - smallest, so the passes the host
    // launches cover the bits of (largest - smallest): visible depths of one c3 frame span ~2^21 ulps -- three 8-bit passes, not four
    // (sort_passes_needed; a frame that needs more than were launched is flagged by k_items and re-run).
    uint32_t kmin = 0;
    if (krange) { uint32_t kmax; load_krange(krange, kmin, kmax); }
    const uint32_t n = clamped_count(n_ptr, n_cap);
    __shared__ uint32_t s_h[256];
    if (threadIdx.x < 256u) s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t base = blockIdx.x * kSortBlock + w * (64 * kSortItems);
    if (base < n) {                               // wave-uniform
        // all loads first, with clamped indices (a load under a lane mask is waited for on the spot, which made
        // the loads of a lane dependent round trips)
        uint32_t key[kSortItems];
#pragma unroll
        for (int k = 0; k < kSortItems; k++) key[k] = keys[min(base + (uint32_t)k * 64u + lane, n - 1u)];
        // One LDS add per RUN of equal digits in a round, by the run's first lane: pairs of one screen tile are neighbours from the
        // second pass on (the first pass ordered them by the low digit, stably), so a round's 64 lanes used to add 1 to the same one
        // or two LDS words, which the LDS serialises -- the second pass's histogram took 13 us at c3 and 85 us at c5 against 4.7 / 28
        // for the first.  (One add per distinct digit through match-any, as k_radix_scatter counts, was measured here earlier: 12.1 us
        // against 11.2 -- eight ballots per round; a run needs one lane shift and one ballot.)
#pragma unroll
        for (int k = 0; k < kSortItems; k++) {
            const bool valid = base + (uint32_t)k * 64u + lane < n;           // valid lanes are a prefix of the wave
            const uint32_t dgt = ((key[k] - kmin) >> shift) & mask;
            const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)dgt, 0x138, 0xF, 0xF, true);      // wave_shr:1 (lane 0 is a head anyway)
            const unsigned long long heads = ballot64(valid && (lane == 0u || prev != dgt));
            const unsigned long long vmask = ballot64(valid);
            if ((heads >> lane) & 1ull) {
                const unsigned long long later = heads & ~((2ull << lane) - 1ull);      // run heads behind this lane
                const uint32_t end = later ? (uint32_t)__ffsll((long long)later) - 1u : (uint32_t)__popcll(vmask);
                atomicAdd(&s_h[dgt], end - lane);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x >= 256u) return;
    const uint32_t cnt = s_h[threadIdx.x];
    // digit-minor layouts ([blk][256], [group][256]) so that one workgroup's reads and writes are contiguous
    ghist[blockIdx.x * 256u + threadIdx.x] = cnt;
    if (cnt) {
        atomicAdd(&gsup[(blockIdx.x >> kSupShift) * 256u + threadIdx.x], cnt);
        // digit totals: with few groups k_radix_scatter sums the group rows itself, with many k_radix_supscan leaves them -- every
        // workgroup adding to the same 256 words (650 adds per word at c3, 5 300 at c5) was the most contended traffic of the sort
    }
    (void)nblk; (void)nbits; (void)gtot; (void)nsup;
}

// Large sorts only (more than kSupDirect groups of 32 workgroups: above ~4 M items): turns the group rows gsup[group][digit] into
// their exclusive prefix over the groups and leaves the digit totals in gtot.  Without it every scatter workgroup summed the rows of
// all groups in front of it -- at c5 (21.8 M pairs, 208 groups) on average 104 KB of L2 reads per workgroup, 690 MB per pass, more than
// the pass moves in keys and values -- and the digit totals were 5 300-way contended atomics.  One wave per digit, 64 groups per round.
// (Summing up to 64 group rows in the scatter workgroups, two rounds of 32, so that the fly path's frames whose pair capacity
// crosses 4 M pairs need no launch of this kernel: 4 219 / 4 139 frames/s against 4 262 / 4 240 with it, same box -- not better.)
__global__ __launch_bounds__(256) void k_radix_supscan(uint32_t* __restrict__ gsup, uint32_t* __restrict__ gtot, uint32_t nsup)
{
    const uint32_t lane = threadIdx.x & 63u, d = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t carry = 0;
    for (uint32_t g0 = 0; g0 < nsup; g0 += 256u) {
        // four rounds of loads in flight together (clamped, unmasked)
        uint32_t v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) v[k] = gsup[(size_t)min(g0 + k * 64u + lane, nsup - 1u) * 256u + d];
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            const uint32_t g = g0 + k * 64u + lane;
            const uint32_t x = g < nsup ? v[k] : 0u;
            const uint32_t inc = wave_incl_scan(x, lane);
            if (g < nsup) gsup[(size_t)g * 256u + d] = carry + inc - x;
            carry += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
        }
    }
    if (lane == 0u) gtot[d] = carry;
}

// The block's 4096 items are first ranked INTO LDS (sorted by digit inside the block), then copied out: consecutive threads
// write consecutive addresses of a digit's run instead of every lane storing two separate words to its own rank position
// (a wave's store instruction used to touch as many regions as it held distinct digits).
// AUX (the depth passes of GSWT_ORDER_DEPTH): a second 4-byte payload per item -- the pair's tile id -- travels with it.
//   AUX = 1: the ranking leaves each sorted item's source position inside the block (u16, 8 KB of LDS) and the copy-out fetches the payload
//            from the block's own 16 KB window of the input (64 scattered dwords per wave instruction).  Three workgroups per CU.
//   AUX = 2: the payload is staged through LDS beside the (key, value) pair (16 KB more: 60 KB per workgroup, two workgroups per CU).
// Measured per pass (scatter<512>, without payload / AUX 1 / AUX 2): c3 (2.66 M pairs) 18.5 / 22.7 / 26.6 us, c3d (8.2 M) 37.5 / 55.7 / 54.8,
// c5 (21.4 M) 99 / 164 / 145: the small sort is a latency chain that wants the third workgroup, the large one pays for the scattered
// reads.  launch_sort takes AUX 2 above 12 M pairs of capacity.
// (512-thread build: at most 80 VGPRs, so that THREE workgroups fit a CU -- c3's 649 blocks are then all resident at once (768 slots) and the
// kernel lasts one block's chain; at 86 VGPRs, where the five-instruction match_digit first left it, two fit, the blocks ran in two
// generations, and the chain it had shortened from 12.8 to 9.0 us bought nothing)
template <int kSortThreads, int AUX>
__global__ __launch_bounds__(kSortThreads) __attribute__((amdgpu_waves_per_eu((kSortThreads == 512 && AUX != 2) ? 6 : 1, 8))) void k_radix_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                       uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                       const unsigned long long* __restrict__ n_ptr, uint32_t n_cap, uint32_t shift,
                                                       uint32_t mask, uint32_t nbits, const uint32_t* __restrict__ ghist,
                                                       const uint32_t* __restrict__ gsup, const uint32_t* __restrict__ gtot,
                                                       uint32_t nblk, uint32_t nsup, uint2* __restrict__ ranges, const uint32_t* __restrict__ krange,
                                                       const uint32_t* __restrict__ aux_in, uint32_t* __restrict__ aux_out)
{
    constexpr int kSortItems = kSortBlock / kSortThreads, kSortWaves = kSortThreads / 64;
    uint32_t kmin = 0;                                  // (see k_radix_hist)
    if (krange) { uint32_t kmax; load_krange(krange, kmin, kmax); }
    const uint32_t n = clamped_count(n_ptr, n_cap);
    if (blockIdx.x * kSortBlock >= n) return;
#ifdef GSWT_TRACE
    // phase stamps of the scatter pass (tools/sort_trace.py): rows kTraceItems / 2 + (shift ? 4096 : 0) + block
    const bool tr_on = threadIdx.x == 0 && blockIdx.x < 4096u;
    const uint32_t tr_item = kTraceItems / 2u + (shift ? 4096u : 0u) + blockIdx.x;
    GSWT_TR(0, GSWT_NOW())
#endif
    __shared__ uint32_t s_h[kSortWaves][256];
    __shared__ uint32_t s_g[256];                       // digit -> (global base of this block's run) - (its start inside the block)
    __shared__ uint32_t s_w[4], s_w2[4];
    __shared__ uint32_t s_gs[2][256];                   // direct group sums: [0] earlier groups, [1] all groups (waves 4..7 -> waves 0..3)
    __shared__ uint2 s_kv[kSortBlock];
    __shared__ uint32_t s_aux[AUX == 2 ? kSortBlock : 1];    // AUX 2: the payload, ranked beside s_kv
    __shared__ uint16_t s_src[AUX == 1 ? kSortBlock : 1];    // AUX 1: sorted position inside the block -> source position inside the block
    const uint32_t w = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (int k = 0; k < kSortWaves * 256 / kSortThreads; k++) (&s_h[0][0])[k * kSortThreads + threadIdx.x] = 0;
    __syncthreads();
    const uint32_t blk0 = blockIdx.x * kSortBlock;
    const uint32_t base = blk0 + w * (64 * kSortItems);
    uint32_t key[kSortItems], val[kSortItems], aux[AUX == 2 ? kSortItems : 1];
    // every load of the workgroup is issued before anything is consumed: clamped indices instead of lane masks
    // (masked loads were waited for one by one: 16 + ~30 dependent round trips per wave, the whole 31 us of this kernel)
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const uint32_t i = min(base + (uint32_t)k * 64u + lane, n - 1u);      // n > 0 here
        key[k] = keys_in[i];
        val[k] = vals_in[i];
        if (AUX == 2) aux[AUX == 2 ? k : 0] = aux_in[i];
    }
    // digit = threadIdx.x (first 256 threads): global base of this workgroup's first item with that digit = items with a
    // smaller digit + the same digit in earlier groups of 32 workgroups (gsup) + in earlier workgroups of this group (ghist)
    const uint32_t d = threadIdx.x & 255u;
    const bool dig = threadIdx.x < 256u;
    const uint32_t sb = blockIdx.x >> kSupShift;
    uint32_t pre = 0, g_tot = 0;
    if (dig) {                                                                   // wave-uniform (waves 0..3)
        uint32_t t[32];
#pragma unroll
        for (uint32_t u = 0; u < 32u; u++) {                                     // <= 31 earlier workgroups of this group
            const uint32_t j = (sb << kSupShift) + u;
            t[u] = ghist[min(j, nblk - 1u) * 256u + d];
        }
        if (nsup <= kSupDirect && kSortThreads >= 512) {
            // the group rows are summed by the upper half of the workgroup (below), which has no digit column of its own
        } else if (nsup <= kSupDirect) {                                         // every group row at once: digit total + earlier groups
            uint32_t g[kSupDirect];
#pragma unroll
            for (uint32_t u = 0; u < kSupDirect; u++) g[u] = gsup[min(u, nsup - 1u) * 256u + d];
#pragma unroll
            for (uint32_t u = 0; u < kSupDirect; u++) { if (u < nsup) g_tot += g[u]; if (u < sb) pre += g[u]; }
        } else {                                                                 // k_radix_supscan ran: exclusive group prefixes + digit totals
            g_tot = gtot[d];
            pre = gsup[(size_t)sb * 256u + d];
        }
#pragma unroll
        for (uint32_t u = 0; u < 32u; u++) if ((sb << kSupShift) + u < blockIdx.x) pre += t[u];
    } else if (nsup <= kSupDirect && kSortThreads >= 512 && threadIdx.x < 512u) {
        uint32_t g[kSupDirect], ge = 0, ga = 0;
#pragma unroll
        for (uint32_t u = 0; u < kSupDirect; u++) g[u] = gsup[min(u, nsup - 1u) * 256u + d];
#pragma unroll
        for (uint32_t u = 0; u < kSupDirect; u++) { if (u < nsup) ga += g[u]; if (u < sb) ge += g[u]; }
        s_gs[0][d] = ge; s_gs[1][d] = ga;                                        // read by thread d after the next barrier
    }
#ifdef GSWT_TRACE
    { unsigned long long sink = key[0] + val[0] + pre; asm volatile("" :: "v"(sink)); }     // the loads have arrived
    GSWT_TR(1, GSWT_NOW())
#endif
    // per-wave digit counts: one LDS add per distinct digit of a 64-item round (see k_radix_hist); the peer masks are kept for
    // the ranking below
    // (a lane's rank among its peers = the peers below it: v_mbcnt on the mask, no (1 << lane) - 1 kept in two registers across the kernel)
    constexpr bool kCachePeers = kSortItems <= 8;       // 16 rounds of masks would cost the 256-thread build a wave per SIMD
    unsigned long long pm[kCachePeers ? kSortItems : 1];
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const bool valid = base + (uint32_t)k * 64u + lane < n;
        const uint32_t dgt = ((key[k] - kmin) >> shift) & mask;
        const unsigned long long peers = match_digit(dgt, valid, nbits);
        if (kCachePeers) pm[k] = peers;
        if (valid && lanes_below(peers) == 0u) atomicAdd(&s_h[w][dgt], (uint32_t)__popcll(peers));
    }
    GSWT_TR(2, GSWT_NOW())
    {
        // exclusive scans over the first 256 threads (the other waves only take part in the barriers): of the digit totals
        // (global bases) and of this block's digit counts (positions inside the block)
        const bool direct = nsup <= kSupDirect && kSortThreads >= 512;
        if (direct) {                                                            // uniform
            __syncthreads();
            if (dig) { pre += s_gs[0][d]; g_tot = s_gs[1][d]; }
        }
        uint32_t inc = wave_incl_scan(g_tot, lane);
        if (dig && lane == 63u) s_w[w] = inc;
        __syncthreads();                                                         // also: the LDS counts are complete
        uint32_t cnt[kSortWaves], blk_cnt = 0;
        if (dig) {
#pragma unroll
            for (int k = 0; k < kSortWaves; k++) { cnt[k] = s_h[k][d]; blk_cnt += cnt[k]; }
        }
        const uint32_t inc2 = wave_incl_scan(blk_cnt, lane);
        if (dig && lane == 63u) s_w2[w] = inc2;
        __syncthreads();
        if (dig) {
            uint32_t wb = 0, wb2 = 0;
            for (uint32_t i = 0; i < w; i++) { wb += s_w[i]; wb2 += s_w2[i]; }
            const uint32_t gbase = wb + inc - g_tot + pre;
            uint32_t lb = wb2 + inc2 - blk_cnt;                                  // start of digit d inside the block
            s_g[d] = gbase - lb;
#pragma unroll
            for (int k = 0; k < kSortWaves; k++) { s_h[k][d] = lb; lb += cnt[k]; }      // per-wave positions inside the block
        }
    }
    __syncthreads();
    GSWT_TR(3, GSWT_NOW())
    // Ranking, 64 items per round: match-any on the digit bits gives every lane its peers; the lowest peer takes the run's
    // position with ONE returning LDS add and hands it to the others through a lane shuffle.  (Round 1 read the counter,
    // fenced the wave, wrote it back and fenced again: two dependent LDS round trips per round that no other round could
    // overlap; the adds of consecutive rounds are independent instructions.)
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        uint32_t i = base + k * 64 + lane;
        bool valid = i < n;
        uint32_t dgt = valid ? (((key[k] - kmin) >> shift) & mask) : 0u;
        const unsigned long long peers = kCachePeers ? pm[kCachePeers ? k : 0] : match_digit(dgt, valid, nbits);
        const uint32_t rank = lanes_below(peers);
        uint32_t pos = 0;
        if (valid && rank == 0u) pos = atomicAdd(&s_h[w][dgt], (uint32_t)__popcll(peers));
        const int leader = valid ? (int)__ffsll((long long)peers) - 1 : (int)lane;
        pos = (uint32_t)__shfl((int)pos, leader, 64);
        if (valid) {
            s_kv[pos + rank] = make_uint2(key[k], val[k]);
            if (AUX == 2) s_aux[AUX == 2 ? pos + rank : 0u] = aux[AUX == 2 ? k : 0];
            if (AUX == 1) s_src[AUX == 1 ? pos + rank : 0u] = (uint16_t)(w * (64u * kSortItems) + (uint32_t)k * 64u + lane);
        }
    }
    __syncthreads();
    GSWT_TR(4, GSWT_NOW())
    const uint32_t n_blk = min((uint32_t)kSortBlock, n - blk0);
#pragma unroll
    for (int k = 0; k < kSortItems; k++) {
        const uint32_t i = (uint32_t)k * kSortThreads + threadIdx.x;
        if (i < n_blk) {
            const uint2 kv = s_kv[i];
            const uint32_t gp = s_g[((kv.x - kmin) >> shift) & mask] + i;
            vals_out[gp] = kv.y;
            if (AUX == 2) aux_out[gp] = s_aux[AUX == 2 ? i : 0u];
            if (AUX == 1) aux_out[gp] = aux_in[blk0 + s_src[AUX == 1 ? i : 0u]];
            if (!ranges) keys_out[gp] = kv.x;
            else {
                // LAST pass of the frame's pair sort: the block in LDS is sorted by the whole key (the passes before ordered the
                // input by the lower digits, stably), so the first and last item of every run of equal keys are neighbours' business --
                // they leave the tile's [start, end) in the sorted pair list: start as ~start, both through atomicMax on the zeroed
                // table, because a tile's run can continue in the next block.  This replaces a pass over the sorted keys (k_ranges,
                // 5 us + a launch at c3), and the sorted keys themselves are not written any more: nothing reads them.
                const uint32_t kp = s_kv[i == 0u ? 0u : i - 1u].x, kn = s_kv[min(i + 1u, n_blk - 1u)].x;
                if (i == 0u || kp != kv.x) atomicMax(&ranges[kv.x].x, ~gp);
                if (i == n_blk - 1u || kn != kv.x) atomicMax(&ranges[kv.x].y, gp + 1u);
            }
        }
    }
    GSWT_TR(5, GSWT_NOW())
}

// ------------------------------------------------------------------------------------
// Merged-group lists on the device (Scene::sort_raw_depth_vec for every MergedFrom group of a sort event,
// wangtile.rs:595-670 + scene.rs:655-698, in one segmented sort).
//   segment  = one member's raw-depth array (plus the other LOD's when the member is Changing)
//   k_mg_minmax : per group min / max of the concatenated raw depths
//   k_mg_keys   : key = group << 16 | bucket, bucket = clamp(floor((d - min) as f32 * (65535 / (max - min)))),
//                 NaN -> 0 as in Rust's `as i32`; val = position in the concatenation
//   radix sort  : stable on 16 + log2(groups) bits  == the CPU's stable scatter by bucket
//   k_mg_final  : reverse inside each group (`depth_index.reverse()`), map concat position -> gs_index / map id / lod
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mg_find_seg(const MergeSeg* __restrict__ segs, uint32_t n_segs, uint32_t e)
{
    uint32_t lo = 0, hi = n_segs;                     // largest s with segs[s].start <= e
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (segs[mid].start <= e) lo = mid; else hi = mid; }
    return lo;
}

// (a group's mn / mx arrive as INT_MAX / INT_MIN with the event's upload: gswt_set_draws_merge_groups -- until the end of round 4 a
// kernel of its own, k_mg_init, set them: one launch per sort event)
// Both passes over the concatenation run on a block table built with the segments: block = (segment, first entry of up to
// 1024 inside it), so a workgroup reads ONE segment record instead of binary-searching the segment of every entry
// (k_mg_minmax was 75 us for 1.24 M entries that way: nine dependent loads per entry plus per-lane atomics).
__global__ __launch_bounds__(256) void k_mg_minmax(const MergeSeg* __restrict__ segs, const uint2* __restrict__ blocks, const int32_t* __restrict__ raw,
                                                   MergeGroup* __restrict__ groups)
{
    __shared__ int32_t s_mn[4], s_mx[4];
    const uint2 b = blocks[blockIdx.x];
    const MergeSeg sg = segs[b.x];
    int32_t mn = 2147483647, mx = -2147483647 - 1;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t i = b.y + k * 256u + threadIdx.x;
        const int32_t d = raw[sg.src + min(i, sg.len - 1u)];             // clamped, unmasked: the duplicate changes neither min nor max
        mn = min(mn, d); mx = max(mx, d);
    }
    for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_down(mn, off, 64)); mx = max(mx, __shfl_down(mx, off, 64)); }
    if ((threadIdx.x & 63u) == 0) { s_mn[threadIdx.x >> 6] = mn; s_mx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&groups[sg.group].mn, min(min(s_mn[0], s_mn[1]), min(s_mn[2], s_mn[3])));
        atomicMax(&groups[sg.group].mx, max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3])));
    }
}

__global__ __launch_bounds__(256) void k_mg_keys(const MergeSeg* __restrict__ segs, const uint2* __restrict__ blocks, const int32_t* __restrict__ raw,
                                                 const MergeGroup* __restrict__ groups, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint2 b = blocks[blockIdx.x];
    const MergeSeg sg = segs[b.x];
    const MergeGroup g = groups[sg.group];
    // scene.rs:669-676: depth_inv = 65535 / (max - min) (f32), bucket = floor((d - min) as f32 * depth_inv) as i32, clamped
    const float depth_inv = 65535.0f / (float)(int32_t)(g.mx - g.mn);
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t i = b.y + k * 256u + threadIdx.x;
        if (i >= sg.len) break;
        const int32_t d = raw[sg.src + i];
        const float v = floorf((float)(int32_t)(d - g.mn) * depth_inv);
        int32_t bk;
        if (v != v) bk = 0;                                  // NaN (max == min: 0 * inf) -> 0
        else if (v >= 2147483648.0f) bk = 2147483647;
        else if (v <= -2147483648.0f) bk = -2147483647 - 1;
        else bk = (int32_t)v;
        bk = min(max(bk, 0), 65535);
        const uint32_t e = sg.start + i;
        keys[e] = (sg.group << 16) | (uint32_t)bk;
        vals[e] = e;
    }
}

__global__ __launch_bounds__(256) void k_mg_final(const MergeSeg* __restrict__ segs, uint32_t n_segs, const MergeGroup* __restrict__ groups,
                                                  const uint32_t* __restrict__ sorted_keys, const uint32_t* __restrict__ sorted_vals,
                                                  uint32_t n_total, uint32_t* __restrict__ merged_list, uint32_t* __restrict__ merged_map)
{
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= n_total) return;
    const uint32_t g = sorted_keys[p] >> 16;
    const MergeGroup gr = groups[g];
    const uint32_t out = gr.out_base + (gr.len - 1u - (p - gr.base));      // depth_index.reverse()
    const uint32_t e = sorted_vals[p];
    const MergeSeg sg = segs[mg_find_seg(segs, n_segs, e)];
    merged_list[out] = ((e - sg.start) + sg.gs_offset) | (sg.lod << kLodShift);
    merged_map[out] = sg.map_index;
}

// Groups whose list one of the retained sort events already holds: one workgroup per 1024 entries of a copy job (block table as above).
__global__ __launch_bounds__(256) void k_mg_copy(const MergeCopy* __restrict__ jobs, const uint2* __restrict__ blocks, const uint2* __restrict__ remap,
                                                 const MergeSources src, uint32_t* __restrict__ new_list, uint32_t* __restrict__ new_map)
{
    __shared__ uint2 s_pairs[256];
    const uint2 b = blocks[blockIdx.x];
    const MergeCopy jb = jobs[b.x];
    const uint32_t* __restrict__ old_list = src.list[jb.src_set < (uint32_t)kMergeSources ? jb.src_set : 0u];
    const uint32_t* __restrict__ old_map = src.map[jb.src_set < (uint32_t)kMergeSources ? jb.src_set : 0u];
    const uint32_t np = min(jb.n_pairs, 256u);
    if (threadIdx.x < np) s_pairs[threadIdx.x] = remap[jb.first_pair + threadIdx.x];
    __syncthreads();
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t i = b.y + k * 256u + threadIdx.x;
        if (i >= jb.len) break;
        new_list[jb.dst + i] = old_list[jb.src + i];
        uint32_t m = old_map[jb.src + i];
        for (uint32_t q = 0; q < np; q++) if (s_pairs[q].x == m) { m = s_pairs[q].y; break; }
        new_map[jb.dst + i] = m;
    }
}

// ------------------------------------------------------------------------------------
// [start, end) of each screen tile in the sorted pair list: left by the LAST pass of the pair sort (k_radix_scatter, `ranges`) as
// (~start, end) per tile, (0, 0) for a tile without pairs; k_items decodes it.  (Until the end of round 2 a kernel of its own,
// k_ranges, read the sorted keys back for it: 5 us + a launch at c3, 18 us at c5.)
// ------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------
// k_tile_depth_sort -- GSWT_ORDER_DEPTH, the tile-local path: the pair list has been sorted on the tile bits with each pair's depth bits as
// the payload; one workgroup per screen tile now sorts that tile's slice by depth INSIDE LDS -- a stable LSD radix sort on the bits of
// (depth key - the TILE's smallest key), all passes back to back -- and writes the slots back in place.  Inside a tile: true depth order,
// equal depths in composite order: bit for bit what the global depth passes in front of the tile sort produce, without their three trips
// through HBM, their histogram kernels and their launches.  The ranking is k_radix_scatter's: each wave owns a run of consecutive
// 64-item rounds and ranks them with ballot match-any (in-wave order = list order), per-wave digit counters in LDS, one exclusive scan
// over (digit, wave).  The digit width follows the tile's own key span: `passes` = ceil(bits / 8) digits of ceil(bits / passes) bits.
//
// Four size classes, one launch each (c3: 8 160 screen tiles, 4 679 of them sky; 2 133 lists of up to 512 pairs hold 18 % of the 2.66 M
// pairs, 1 290 of 513 .. 4 096 hold 72 %, 58 longer ones 10 % -- tools/tile_lengths.py):
//   <64, 8, false>    one WAVE per screen tile, lists of up to 512 pairs (5 KB of LDS, no barrier that costs anything)
//   <256, 16, false>  one 256-thread workgroup per tile, 513 .. 4 096 pairs (36 KB); a longer list's tile id goes onto `long_list`
//                     ([0] = count, cleared by k_cull)
//   <1024, 16, true>  a fixed grid that walks long_list: up to 16 384 pairs per list (144 KB of LDS, one workgroup per CU)
//   k_tile_depth_sort_xl (below)  lists beyond that: the same passes through global memory, one workgroup per list of a second list
// (The first build -- one 512-thread / 64-KB workgroup for every tile, 32 unconditional loads per thread -- took 115 us at c3; the classes
// as built here 12.6 + 27.9 + 16.0 us, each with ~4.8 us of launch floor, against 3 x (24.7 + 5.1) us for the global depth passes.  Other
// cuts measured: profiles/r04_depth_sort_variants.txt.)
// ------------------------------------------------------------------------------------
constexpr uint32_t kTileSortCap = 16384u, kTileSortLongGrid = 256u;
// WAVE: one wave of a larger workgroup sorts a list on its own (THREADS = 64): its LDS slices are private and a wave's LDS operations complete
// in order, so the workgroup barriers become compiler-level fences
template <bool WAVE>
__device__ __forceinline__ void tls_sync()
{
    if (WAVE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else __syncthreads();
}
template <int THREADS, int ROUNDS, bool LONG, bool WAVE = false>
__device__ __forceinline__ void tile_depth_sort_one(uint32_t tile, const uint2* __restrict__ ranges, uint32_t* __restrict__ vals, const uint32_t* __restrict__ dkeys,
                                                    uint32_t* __restrict__ long_list, uint32_t* __restrict__ xl_list, unsigned long long* __restrict__ counters,
                                                    uint32_t n_lo, bool push, uint2* s_kv, uint32_t (*s_h)[256], uint32_t* s_w, uint32_t* s_mn, uint32_t* s_mx)
{
    constexpr uint32_t CAP = (uint32_t)(THREADS * ROUNDS);
    constexpr int NW = THREADS / 64;
    // (everything the control flow hangs on is read through v_readfirstlane: a value loaded from memory sits in a vector register, and the
    // compiler then predicates every "uniform" branch below with exec-mask nests instead of scalar branches)
    const uint2 rgv = ranges[__builtin_amdgcn_readfirstlane(tile)];  // (~start, end); (0, 0): no pairs
    const uint32_t rg_x = __builtin_amdgcn_readfirstlane(rgv.x), rg_y = __builtin_amdgcn_readfirstlane(rgv.y);
    if (rg_y == 0u) return;
    const uint32_t start = ~rg_x, n = rg_y - start;
    if (n <= max(n_lo, 1u)) return;                               // (a shorter list: the class below's)
    if (n > CAP) {                                                // workgroup-uniform
        if ((WAVE ? (threadIdx.x & 63u) : threadIdx.x) == 0u && (LONG || push)) {
            if (LONG) counters[3] = 1ull;                         // (cannot happen: the list only holds what fits)
            else {
                uint32_t* const lst = n > kTileSortCap ? xl_list : long_list;      // beyond the LDS buffer: k_tile_depth_sort_xl
                lst[1u + atomicAdd(&lst[0], 1u)] = tile;
            }
        }
        return;
    }
    static_assert(!WAVE || THREADS == 64, "wave-local mode is one wave");
    const uint32_t lane = threadIdx.x & 63u, w = WAVE ? 0u : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), tid = WAVE ? lane : threadIdx.x;
    const uint32_t rounds = (n + 63u) >> 6, rpw = (rounds + (uint32_t)NW - 1u) / (uint32_t)NW;   // wave w owns rounds [w rpw, (w + 1) rpw)
    const uint32_t base_w = w * rpw * 64u;
    uint32_t key[ROUNDS], val[ROUNDS];
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
    for (int k = 0; k < ROUNDS; k++) {
        key[k] = 0u; val[k] = 0u;
        if ((uint32_t)k >= rpw || base_w + (uint32_t)k * 64u >= n) continue;                      // wave-uniform: this round holds no item
        const uint32_t i = min(base_w + (uint32_t)k * 64u + lane, n - 1u);                         // (the lanes past the end repeat the last item)
        key[k] = dkeys[start + i];
        val[k] = vals[start + i];
        mn = min(mn, key[k]); mx = max(mx, key[k]);
    }
    mn = wave_min(mn); mx = wave_max(mx);
    if (lane == 0u) { s_mn[w] = mn; s_mx[w] = mx; }
    tls_sync<WAVE>();
#pragma unroll
    for (int q = 0; q < NW; q++) { mn = min(mn, s_mn[q]); mx = max(mx, s_mx[q]); }
    mn = __builtin_amdgcn_readfirstlane(mn); mx = __builtin_amdgcn_readfirstlane(mx);
    const uint32_t span = mx - mn;
    if (span == 0u) return;                                       // one depth: the list is in order already
    const uint32_t bits = 32u - (uint32_t)__clz((int)span), passes = (bits + 7u) >> 3, nb = (bits + passes - 1u) / passes, dmask = (1u << nb) - 1u;
#pragma unroll
    for (int k = 0; k < ROUNDS; k++) key[k] -= mn;
    for (uint32_t p = 0; p < passes; p++) {
        const uint32_t shift = nb * p;
        for (uint32_t q = tid; q < (uint32_t)NW * 256u; q += (uint32_t)THREADS) (&s_h[0][0])[q] = 0u;
        tls_sync<WAVE>();                                          // (also: the previous pass's reads of s_kv are done)
        uint32_t pm[ROUNDS];                                     // per round: rank among the peers | first peer's lane << 8 | peers << 16
#pragma unroll
        for (int k = 0; k < ROUNDS; k++) {
            pm[k] = 0u;
            if ((uint32_t)k >= rpw || base_w + (uint32_t)k * 64u >= n) continue;
            const bool valid = base_w + (uint32_t)k * 64u + lane < n;
            const uint32_t dgt = (key[k] >> shift) & dmask;
            const unsigned long long peers = match_digit(dgt, valid, nb);
            const uint32_t rank = lanes_below(peers), cnt = (uint32_t)__popcll(peers);
            pm[k] = rank | (valid ? (uint32_t)(__ffsll((long long)peers) - 1) : lane) << 8 | cnt << 16;
            if (valid && rank == 0u) atomicAdd(&s_h[w][dgt], cnt);
        }
        tls_sync<WAVE>();
        // exclusive scan over (digit, wave): the first 256 threads own one digit each (a one-wave workgroup: four per lane)
        constexpr int SCAN_T = THREADS < 256 ? THREADS : 256, DPT = 256 / SCAN_T;
        uint32_t cnt[DPT][NW], tot = 0, inc = 0;
        const bool dig = tid < (uint32_t)SCAN_T;
        const uint32_t d0 = (tid & (uint32_t)(SCAN_T - 1)) * (uint32_t)DPT;
        if (dig) {
#pragma unroll
            for (int j = 0; j < DPT; j++)
#pragma unroll
                for (int q = 0; q < NW; q++) { cnt[j][q] = s_h[q][d0 + (uint32_t)j]; tot += cnt[j][q]; }
            inc = wave_incl_scan(tot, lane);
            if (lane == 63u) s_w[w] = inc;
        }
        tls_sync<WAVE>();
        if (dig) {
            uint32_t b = inc - tot;
            for (uint32_t q = 0; q < w; q++) b += s_w[q];
#pragma unroll
            for (int j = 0; j < DPT; j++)
#pragma unroll
                for (int q = 0; q < NW; q++) { s_h[q][d0 + (uint32_t)j] = b; b += cnt[j][q]; }
        }
        tls_sync<WAVE>();
        // (measured and dropped: the counter updates, broadcasts and stores of all rounds as three loops instead of one, so that the LDS
        // round trips of different rounds overlap: 35.3 against 30.6 us at c3 for the 256-thread class -- the extra live registers cost a wave)
#pragma unroll
        for (int k = 0; k < ROUNDS; k++) {
            if ((uint32_t)k >= rpw || base_w + (uint32_t)k * 64u >= n) continue;
            const bool valid = base_w + (uint32_t)k * 64u + lane < n;
            const uint32_t dgt = (key[k] >> shift) & dmask;
            uint32_t pos = 0;
            if (valid && (pm[k] & 255u) == 0u) pos = atomicAdd(&s_h[w][dgt], pm[k] >> 16);
            pos = (uint32_t)__shfl((int)pos, (int)((pm[k] >> 8) & 255u), 64) + (pm[k] & 255u);
            if (valid) s_kv[pos] = make_uint2(key[k], val[k]);
        }
        tls_sync<WAVE>();
        if (p + 1u == passes) break;
#pragma unroll
        for (int k = 0; k < ROUNDS; k++) {                        // back into list order for the next pass
            const uint32_t i = base_w + (uint32_t)k * 64u + lane;
            if ((uint32_t)k < rpw && i < n) { const uint2 kv = s_kv[i]; key[k] = kv.x; val[k] = kv.y; }
        }
    }
    for (uint32_t i = tid; i < n; i += (uint32_t)THREADS) vals[start + i] = s_kv[i].y;
}

// tile_depth_sort_xl (k_tile_depth_sort_xl in the documents): the lists that do not fit the LDS buffer (more than 16 384 pairs: c5's horizon tiles hold up to 27 908, a small
// framebuffer in front of a big scene more).  One 1 024-thread workgroup per list (a fixed grid over xl_list, which the 256-thread class
// fills), the same stable LSD passes on (key - the list's smallest key), but the items stay in global memory: a pass counts the digits of
// the list wave by wave (wave w owns one contiguous sixteenth of it, so in-wave order = list order), scans (digit, wave) in LDS, and
// scatters into the other half of the sort's ping-pong buffers (free once the tile passes are done) at the list's own offset; an odd
// number of passes is followed by a copy back.  Two reads of the list per pass instead of none -- a few lists per frame, one per CU, in
// place of three global passes over every pair of the frame (c5: 3 x (22 + 88) us), and no frame is ever re-run for the length of a list.
// Not a launch of its own: the long-list workgroups of k_tile_depth_sort<1024, 16, true> walk this list behind theirs (an empty launch is 4.8 us).
__device__ __forceinline__ void tile_depth_sort_xl(const uint2* __restrict__ ranges, uint32_t* vals_a, uint32_t* keys_a, uint32_t* vals_b,
                                                   uint32_t* keys_b, uint32_t n_tiles, const uint32_t* __restrict__ long_list,
                                                   uint32_t (*s_h)[256], uint32_t* s_w, uint32_t* s_mn, uint32_t* s_mx)
{
    constexpr uint32_t NW = 16u;                                   // (called by the 1 024-thread long-list workgroups behind their own list)
    const uint32_t* const xl_list = long_list + n_tiles + 1u;
    const uint32_t n_xl = min(xl_list[0], n_tiles);                // written by the launch in front of this one
    const uint32_t lane = threadIdx.x & 63u, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (uint32_t t = blockIdx.x; t < n_xl; t += gridDim.x) {
        const uint2 rgv = ranges[__builtin_amdgcn_readfirstlane(xl_list[1u + t])];
        const uint32_t start = ~(uint32_t)__builtin_amdgcn_readfirstlane(rgv.x), n = (uint32_t)__builtin_amdgcn_readfirstlane(rgv.y) - start;
        const uint32_t per = ((n + NW * 64u - 1u) / (NW * 64u)) * 64u;            // items per wave: whole 64-item rounds
        const uint32_t lo = min(w * per, n), hi = min(lo + per, n);
        // the list's key range
        uint32_t mn = 0xFFFFFFFFu, mx = 0u;
        for (uint32_t i = lo + lane; i < hi; i += 64u) { const uint32_t k = keys_a[start + i]; mn = min(mn, k); mx = max(mx, k); }
        mn = wave_min(mn); mx = wave_max(mx);
        if (lane == 0u) { s_mn[w] = mn; s_mx[w] = mx; }
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < NW; q++) { mn = min(mn, s_mn[q]); mx = max(mx, s_mx[q]); }
        mn = __builtin_amdgcn_readfirstlane(mn); mx = __builtin_amdgcn_readfirstlane(mx);
        const uint32_t span = mx - mn;
        uint32_t passes = 0, nb = 8u;
        if (span != 0u) { const uint32_t bits = 32u - (uint32_t)__clz((int)span); passes = (bits + 7u) >> 3; nb = (bits + passes - 1u) / passes; }
        const uint32_t dmask = (1u << nb) - 1u;
        uint32_t* src_k = keys_a; uint32_t* src_v = vals_a; uint32_t* dst_k = keys_b; uint32_t* dst_v = vals_b;
        for (uint32_t p = 0; p < passes; p++) {
            const uint32_t shift = nb * p;
            for (uint32_t q = threadIdx.x; q < NW * 256u; q += 1024u) (&s_h[0][0])[q] = 0u;
            __syncthreads();
            for (uint32_t r = lo; r < hi; r += 64u) {              // (wave-uniform bounds)
                const bool valid = r + lane < hi;
                const uint32_t dgt = ((src_k[start + min(r + lane, hi - 1u)] - mn) >> shift) & dmask;
                const unsigned long long peers = match_digit(dgt, valid, nb);
                if (valid && lanes_below(peers) == 0u) atomicAdd(&s_h[w][dgt], (uint32_t)__popcll(peers));
            }
            __syncthreads();
            // exclusive scan over (digit, wave): thread d < 256 owns digit d
            uint32_t cnt[NW], tot = 0, inc = 0;
            const bool dig = threadIdx.x < 256u;
            const uint32_t d = threadIdx.x & 255u;
            if (dig) {
#pragma unroll
                for (uint32_t q = 0; q < NW; q++) { cnt[q] = s_h[q][d]; tot += cnt[q]; }
                inc = wave_incl_scan(tot, lane);
                if (lane == 63u) s_w[w] = inc;
            }
            __syncthreads();
            if (dig) {
                uint32_t b = inc - tot;
                for (uint32_t q = 0; q < w; q++) b += s_w[q];
#pragma unroll
                for (uint32_t q = 0; q < NW; q++) { s_h[q][d] = b; b += cnt[q]; }
            }
            __syncthreads();
            for (uint32_t r = lo; r < hi; r += 64u) {
                const bool valid = r + lane < hi;
                const uint32_t i = start + min(r + lane, hi - 1u);
                const uint32_t key = src_k[i], val = src_v[i];
                const uint32_t dgt = ((key - mn) >> shift) & dmask;
                const unsigned long long peers = match_digit(dgt, valid, nb);
                const uint32_t rank = lanes_below(peers);
                uint32_t pos = 0;
                if (valid && rank == 0u) pos = atomicAdd(&s_h[w][dgt], (uint32_t)__popcll(peers));
                pos = (uint32_t)__shfl((int)pos, valid ? (int)__ffsll((long long)peers) - 1 : (int)lane, 64) + rank;
                if (valid) { dst_k[start + pos] = key; dst_v[start + pos] = val; }
            }
            __syncthreads();                                       // (the workgroup's global stores are visible to its own next pass)
            uint32_t* tk = src_k; src_k = dst_k; dst_k = tk;
            uint32_t* tv = src_v; src_v = dst_v; dst_v = tv;
        }
        if (passes & 1u) {                                         // the sorted slots lie in the other buffer: back to where the compositor reads
            for (uint32_t i = threadIdx.x; i < n; i += 1024u) vals_a[start + i] = vals_b[start + i];
        }
        __syncthreads();
    }
}

template <int THREADS, int ROUNDS, bool LONG>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(THREADS == 256 ? 4 : 1, 8))) void k_tile_depth_sort(const uint2* __restrict__ ranges, uint32_t* vals, uint32_t* dkeys,      // (no __restrict__: the last class writes both)
                                                             uint32_t n_tiles, uint32_t* __restrict__ long_list, unsigned long long* __restrict__ counters,
                                                             uint32_t n_lo, uint32_t push, uint32_t* vals_scratch, uint32_t* dkeys_scratch)
{
    constexpr int NW = THREADS / 64;
    __shared__ uint2 s_kv[THREADS * ROUNDS];
    __shared__ uint32_t s_h[NW][256];
    __shared__ uint32_t s_w[4];
    __shared__ uint32_t s_mn[NW], s_mx[NW];
    if (!LONG) {
        if (blockIdx.x < n_tiles) tile_depth_sort_one<THREADS, ROUNDS, LONG>(blockIdx.x, ranges, vals, dkeys, long_list, long_list + n_tiles + 1u, counters, n_lo, push != 0u, s_kv, s_h, s_w, s_mn, s_mx);
    } else {
        const uint32_t n_long = min(long_list[0], n_tiles);      // written by the launch in front of this one
        for (uint32_t t = blockIdx.x; t < n_long; t += gridDim.x) {
            tile_depth_sort_one<THREADS, ROUNDS, LONG>(long_list[1u + t], ranges, vals, dkeys, long_list, long_list + n_tiles + 1u, counters, n_lo, false, s_kv, s_h, s_w, s_mn, s_mx);
            __syncthreads();                                      // the next list reuses the LDS
        }
        // ... and the lists beyond the LDS buffer, through global memory (dkeys is written there: the pass buffers alternate)
        if (THREADS == 1024) tile_depth_sort_xl(ranges, vals, dkeys, vals_scratch, dkeys_scratch, n_tiles, long_list, s_h, s_w, s_mn, s_mx);
    }
}

// ------------------------------------------------------------------------------------
// k_composite: one workgroup (4 wave64) per work item = (16x16 screen tile, segment of its pair list).
// Wave w owns the 16x4 pixel strip of rows 4w .. 4w+3; inside a wave the four 16-lane groups own the
// four 4x4 sub-blocks of that strip and walk DIFFERENT splats concurrently (the c3 scene averages ~9 covered
// pixels per pair, so a whole wave per splat would leave most lanes idle).
//   stage   : 256 pairs per batch -> LDS (16 KB): the staging lane gathers the 48-B record, unpacks the
//             colour, evaluates the per-(splat, tile) constants F3 and the tile-local pixel-centre box
//   bin     : every wave ballots the batch against its four sub-blocks and appends the hits, in list
//             order, to four wave-private index lists in LDS (mbcnt-compacted byte stores)
//   walk    : counted loop to the longest of the four lists; each group reads its own next index and
//             record (LDS, four distinct addresses per wave), evaluates the canonical per-pixel
//             sequence F4 and blends under predication; a wave whose ballot of "T >= eps" is empty stops
// ------------------------------------------------------------------------------------
// Work items: a tile's pair list is cut into segments of `seg` pairs; item = (tile, segment).
// seg_count[t] = ceil(len / seg): a tile without pairs has no item (k_combine, which visits every tile anyway, writes its background).
// item_base[t] = exclusive scan of seg_count, item_base[n_tiles] = number of items; item_tab[item] =
// (tile, segment << 1 | tile has several segments, first pair, end pair): everything k_composite needs in one load.  Single workgroup (n_tiles is a few thousand to a few
// tens of thousands): one launch instead of count + 3 scan launches.
// GSWT_ORDER_DEPTH (krange != nullptr): also the depth sort's pass check -- counters[2] = 8-bit passes this frame's depth range needs; more
// than were launched = the pair list is not in depth order: the frame is flagged (counters[3]) and the host re-runs it with more passes.
template <bool HEAVY>
__global__ __launch_bounds__(1024) void k_items(const uint2* __restrict__ ranges, int n_tiles, uint32_t seg,
                                                uint32_t* __restrict__ item_base, uint4* __restrict__ item_tab, uint32_t max_items,
                                                const uint32_t* __restrict__ krange, uint32_t n_launched, unsigned long long* __restrict__ counters,
                                                uint32_t all_tiles, uint32_t report_max)
{
    // all_tiles (GSWT_OPT_FOLD_COMBINE): a tile without pairs gets one EMPTY work item -- k_composite then writes its background, and no
    // k_combine launch follows the compositor
    if (krange && blockIdx.x == 0u && threadIdx.x == 0u) {
        const uint32_t need = sort_passes_needed(krange);
        *reinterpret_cast<uint32_t*>(counters + 2) = need;                   // (the low half: the high one takes the longest tile list)
        if (need > n_launched) counters[3] = 1ull;
    }
    // One workgroup per 8192 tiles (c3: one workgroup, c5: four).  Thread t owns tiles base + j * 1024 + t, j = 0..7: every
    // load and store of a wave is contiguous (eight tiles per thread SIDE BY SIDE made each of them 64 separate 8-byte
    // requests: ~8 k requests through one CU's address path, the larger part of the old kernel's 14 us).  The eight block
    // scans are batched: eight wave scans back to back, one table of 8 x 16 wave sums, one 128-entry scan of it by wave 0.
    // A workgroup's first item = the segment count of all tiles in front of it, which it sums itself (<= 8 k tiles per
    // earlier workgroup, contiguous reads) instead of waiting for the others: no inter-workgroup dependence.
    constexpr int kPer = 8;
    __shared__ uint32_t s_t[kPer * 16];           // [j][wave]: wave sums, then their exclusive scan in tile order
    __shared__ uint32_t s_tot, s_carry[16];
    const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
    const int last = n_tiles > 0 ? n_tiles - 1 : 0;
    const bool seg_pow2 = (seg & (seg - 1u)) == 0u;
    const uint32_t seg_sh = (uint32_t)__ffs((int)seg) - 1u;
    const int base = (int)blockIdx.x * 1024 * kPer;
    uint2 r[kPer];
    // (ranges[t] = (~start, end) as the last sort pass leaves it, (0, 0) for a tile without pairs)
#pragma unroll
    for (int j = 0; j < kPer; j++) { r[j] = ranges[min(base + j * 1024 + (int)threadIdx.x, last)]; r[j].x = r[j].y ? ~r[j].x : 0u; }
    // segments of the tiles in front of this workgroup
    uint32_t carry = 0;
    if (base > 0) {
        uint32_t part = 0;
        for (int t0 = 0; t0 < base; t0 += 1024 * kPer) {
            uint2 e[kPer];
#pragma unroll
            for (int j = 0; j < kPer; j++) { e[j] = ranges[t0 + j * 1024 + (int)threadIdx.x]; e[j].x = e[j].y ? ~e[j].x : 0u; }      // t0 + 8191 < base <= n_tiles
#pragma unroll
            for (int j = 0; j < kPer; j++) {
                const uint32_t len = e[j].y - e[j].x;
                const uint32_t qq = seg_pow2 ? (len + seg - 1u) >> seg_sh : (len + seg - 1u) / seg;
                part += all_tiles ? max(qq, 1u) : qq;
            }
        }
        const uint32_t pi = wave_incl_scan(part, lane);
        if (lane == 63u) s_carry[w] = pi;
        __syncthreads();
        for (uint32_t i = 0; i < 16u; i++) carry += s_carry[i];
    }
    if (report_max) {
        // GSWT_ORDER_DEPTH: the longest tile list of the frame -> high half of counters[2]: the host picks the tile-local depth sort (lists that
        // fit its LDS buffer) or the global depth passes for the next frames by it
        uint32_t mx = 0;
#pragma unroll
        for (int j = 0; j < kPer; j++) mx = max(mx, base + j * 1024 + (int)threadIdx.x < n_tiles ? r[j].y - r[j].x : 0u);
        mx = wave_max(mx);
        if (lane == 0u && mx) atomicMax(reinterpret_cast<uint32_t*>(counters + 2) + 1, mx);        // <= 16 per workgroup, <= 4 workgroups
    }
    uint32_t cnt[kPer], inc[kPer];
#pragma unroll
    for (int j = 0; j < kPer; j++) {
        const uint32_t len = r[j].y - r[j].x;
        uint32_t q = seg_pow2 ? (len + seg - 1u) >> seg_sh : (len + seg - 1u) / seg;   // (a 32-bit divide is ~40 VALU)
        if (all_tiles) q = max(q, 1u);
        cnt[j] = base + j * 1024 + (int)threadIdx.x < n_tiles ? q : 0u;
    }
#pragma unroll
    for (int j = 0; j < kPer; j++) {
        inc[j] = wave_incl_scan(cnt[j], lane);
        if (lane == 63u) s_t[j * 16 + (int)w] = inc[j];
    }
    __syncthreads();
    if (w == 0u) {                            // exclusive scan of the 128 wave sums, two per lane
        const uint32_t a = s_t[2u * lane], b2 = s_t[2u * lane + 1u];
        const uint32_t pi = wave_incl_scan(a + b2, lane);
        s_t[2u * lane] = pi - a - b2;
        s_t[2u * lane + 1u] = pi - b2;
        if (lane == 63u) s_tot = pi;
    }
    __syncthreads();
    if (!HEAVY) {
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const int t = base + j * 1024 + (int)threadIdx.x;
            if (t < n_tiles) {
                const uint32_t first = carry + s_t[j * 16 + (int)w] + inc[j] - cnt[j];
                item_base[t] = first;
                const uint32_t multi = cnt[j] > 1u ? 1u : 0u;
                for (uint32_t k = 0; k < cnt[j] && first + k < max_items; k++) {
                    const uint32_t a = r[j].x + k * seg;
                    item_tab[first + k] = make_uint4((uint32_t)t, (k << 1) | multi, a, min(r[j].y, a + seg));
                }
            }
        }
    } else {
        // GSWT_OPT_ITEM_ORDER = 1, heaviest first: the compositor's workgroups start in table order, so the table lists the work items by
        // falling length -- class 0: the full segments (`seg` pairs), classes 1 .. 15: the last (or only) segment of a tile by sixteenths of
        // `seg` -- and the short ones fill the slots the long ones leave.  item_base[t] stays the tile's first PARTIAL slot (k_composite
        // writes a segment's partial to item_base[tile] + segment, k_combine reads them from there): only the hand-out order changes.  Inside a
        // class the order is whatever the waves' LDS atomics make it (the image does not depend on it).  A workgroup orders the items of its
        // own 8 192 tiles (c3: the frame's; c5: a quarter's).
        constexpr uint32_t kCls = 16u;
        __shared__ uint32_t s_cls[kCls], s_off[kCls];
        if (threadIdx.x < kCls) s_cls[threadIdx.x] = 0u;
        __syncthreads();
        uint32_t full[kPer], rcls[kPer];                 // full segments of the tile; class of its remainder item (kCls: none)
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const uint32_t len = r[j].y - r[j].x;
            full[j] = cnt[j] ? (seg_pow2 ? len >> seg_sh : len / seg) : 0u;
            const uint32_t rem = len - full[j] * seg;
            rcls[j] = cnt[j] > full[j] ? (rem ? (kCls - 1u) - min(kCls - 1u, ((rem - 1u) * kCls) / seg) : kCls - 1u) : kCls;
            if (rcls[j] == 0u) rcls[j] = 1u;             // (class 0 is the full segments' alone: its slots are handed out by a scan)
        }
        // per-class counts: one LDS atomic per wave and class present
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            uint32_t fs = full[j];
            fs = wave_sum(fs);
            if (lane == 0u && fs) atomicAdd(&s_cls[0], fs);
            unsigned long long left = ballot64(rcls[j] < kCls);
            while (left) {
                const uint32_t c = (uint32_t)__shfl((int)rcls[j], (int)__ffsll((long long)left) - 1, 64);
                const unsigned long long m = ballot64(rcls[j] == c);
                if (lane == 0u) atomicAdd(&s_cls[c], (uint32_t)__popcll(m));
                left &= ~m;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0u) { uint32_t a = 0; for (uint32_t c = 0; c < kCls; c++) { s_off[c] = a; a += s_cls[c]; } }
        __syncthreads();
        const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
        for (int j = 0; j < kPer; j++) {
            const int t = base + j * 1024 + (int)threadIdx.x;
            const uint32_t first = carry + s_t[j * 16 + (int)w] + inc[j] - cnt[j];
            if (t < n_tiles) item_base[t] = first;
            const uint32_t multi = cnt[j] > 1u ? 1u : 0u;
            // the full segments: a wave scan of their counts, one LDS atomic per wave for the run of slots
            const uint32_t fi = wave_incl_scan(full[j], lane);
            uint32_t fb = 0;
            if (lane == 63u && fi) fb = atomicAdd(&s_off[0], fi);
            fb = (uint32_t)__shfl((int)fb, 63, 64) + fi - full[j];
            for (uint32_t k = 0; k < full[j]; k++) {
                const uint32_t a = r[j].x + k * seg, pos = carry + fb + k;
                if (pos < max_items) item_tab[pos] = make_uint4((uint32_t)t, (k << 1) | multi, a, a + seg);
            }
            // the remainder items, class by class
            unsigned long long left = ballot64(rcls[j] < kCls);
            while (left) {
                const int src = (int)__ffsll((long long)left) - 1;
                const uint32_t c = (uint32_t)__shfl((int)rcls[j], src, 64);
                const unsigned long long m = ballot64(rcls[j] == c);
                uint32_t pb = 0;
                if (lane == (uint32_t)src) pb = atomicAdd(&s_off[c], (uint32_t)__popcll(m));
                pb = (uint32_t)__shfl((int)pb, src, 64);
                if (rcls[j] == c) {
                    const uint32_t k = full[j], a = r[j].x + k * seg, pos = carry + pb + (uint32_t)__popcll(m & lt);
                    if (pos < max_items) item_tab[pos] = make_uint4((uint32_t)t, (k << 1) | multi, a, r[j].y);
                }
                left &= ~m;
            }
        }
    }
    if (threadIdx.x == 0 && base + 1024 * kPer >= n_tiles) item_base[n_tiles] = carry + s_tot;      // the last workgroup
}

#ifdef GSWT_STATS
__device__ unsigned long long g_stats[8];
#define GSWT_STAT_STEP(C) { unsigned long long cm_ = ballot64(C); if ((threadIdx.x & 63u) == 0) { atomicAdd(&g_stats[0], 1ull); \
    atomicAdd(&g_stats[1], (unsigned long long)__popcll(cm_)); if (cm_ == 0ull) atomicAdd(&g_stats[2], 1ull); \
    unsigned z_ = 0; for (int q_ = 0; q_ < 4; q_++) if (((cm_ >> (16 * q_)) & 0xFFFFull) == 0ull) z_++; atomicAdd(&g_stats[6], (unsigned long long)z_); } }
#define GSWT_STAT_BATCH(NMAX, C0, C1, C2, C3) { if ((threadIdx.x & 63u) == 0) { atomicAdd(&g_stats[3], 1ull); atomicAdd(&g_stats[4], (unsigned long long)(NMAX)); \
    atomicAdd(&g_stats[5], (unsigned long long)((C0) + (C1) + (C2) + (C3))); } }
#else
#define GSWT_STAT_STEP(C)
#define GSWT_STAT_BATCH(NMAX, C0, C1, C2, C3)
#endif
// LDS record of a staged pair (two 16-B words; one more dword when a depth buffer is bound):
//   q0 = (iu.x, iu.y, -ku, log2 alpha)   q1 = (iv.x, iv.y, -kv, rgba8 bits)
// alpha rides in the exponent (B = 2^(-r2 log2 e + log2 alpha): one fma + v_exp), the colour stays packed and is
// unpacked by v_cvt_f32_ubyteN in the blend; the accumulators run in 0..255 units and are scaled once at the end.
// COLF (debug draw modes only): colours are floats from the side buffer col_f[slot], staged into a third LDS word.
// Lane geometry of the compositor: wave w owns the 16x4 strip of rows 4w .. 4w+3, 16-lane group g its 4x4 sub-block of columns 4g .. 4g+3.
struct CompLane {
    float lx, ly;                 // tile-local pixel centre of this lane
    int r0;                       // first pixel row (tile-local) of the wave's 16 x 4 strip
    uint32_t lane, grp;
};

typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_splat(float s) { v2f r; r.x = s; r.y = s; return r; }

#ifdef GSWT_EXPERIMENTS
constexpr uint32_t kListStride = 288u;     // (the register-broadcast walk pads a list to whole rounds of 16 + one round of prefetch overrun)
#else
constexpr uint32_t kListStride = 264u;     // u16 entries per sub-block list: 256 hits + padding to an even count + 2 of prefetch overrun
#endif
constexpr uint32_t kNullRec = 256u;        // LDS record no pixel is ever inside (list padding)

// bin + walk of one staged batch (n pairs in LDS) for one wave; updates the lane's (T, colour) state.
// Wave w owns the 16 x 4 pixel strip of rows 4w .. 4w+3, its 16-lane group g the 4 x 4 sub-block of columns 4g .. 4g+3
// (rows, not quadrants: the splat density varies down the screen, so the four lists of a strip are alike and the
// longest one -- which paces the wave -- is close to their mean: 1.98 M wave-steps against 2.28 M on the c3 frame).
// List entries are LDS byte offsets of the records (u16), padded to an even length with the offset of a null record
// whose r^2 is +inf: the walk needs neither a shift nor an `i < n` test nor a mid-pair exit.
// BATCH: pairs per staged batch (256 in k_composite, 128 in k_composite_dw); the list stride and the null record's index follow it.
template <bool EARLY, bool DEPTH, bool COLF, bool PK, bool DPPW, uint32_t BATCH = 256u>
__device__ __forceinline__ void composite_bin_walk(const Frame& f, const CompLane& g, uint32_t n, const float4* s_q0, const float4* s_q1,
                                                   const float4* s_q2, const uint32_t* s_bb, const float* s_dep, uint16_t* wlist,
                                                   float dbuf, float t_eps, float& T, float& ar, float& ag, float& ab, bool& wave_live)
{
    const uint32_t lane = g.lane, grp = g.grp;
    const float lx = g.lx, ly = g.ly;
    constexpr uint32_t kStride = BATCH == 256u ? kListStride : BATCH + 8u;     // u16 entries per sub-block list
    constexpr uint32_t kNull = BATCH == 256u ? kNullRec : BATCH;               // index of the null record behind the batch
    uint16_t* const my_list = wlist + grp * kStride;
    // bin: append this batch's hits to the four sub-block lists (list order preserved)
    uint32_t cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
#pragma unroll
    for (int c = 0; c < (int)(BATCH / 64u); c++) {
        const uint32_t idx = (uint32_t)c * 64u + lane;
        // the staged hit mask of the pair: this wave's strip is bits r0 .. r0 + 3 (r0 = 4 x strip).  Read unconditionally (idx < BATCH: a
        // slot past the batch's end holds an older batch's word) and cleared by a select: predicates that come out of an `if` reach the
        // ballots as 0 / 1 registers compared once more -- eight vector instructions per 64 pairs that four v_cmp already answered
        const uint32_t hm_raw = s_bb[idx] >> (uint32_t)g.r0;
        const uint32_t hm = idx < n ? hm_raw : 0u;
        const bool h0 = (hm & 1u) != 0u, h1 = (hm & 2u) != 0u, h2 = (hm & 4u) != 0u, h3 = (hm & 8u) != 0u;
        const unsigned long long m0 = ballot64(h0), m1 = ballot64(h1), m2 = ballot64(h2), m3 = ballot64(h3);
#define GSWT_APPEND(H, M, CNT, G)                                                                                         \
        if (M) {                                                                                                            \
            if (H) wlist[(G) * kStride + (CNT) + __builtin_amdgcn_mbcnt_hi((uint32_t)((M) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(M), 0u))] = (uint16_t)(idx * 16u); \
            CNT += (uint32_t)__popcll(M);                                                                                   \
        }
        GSWT_APPEND(h0, m0, cnt0, 0u) GSWT_APPEND(h1, m1, cnt1, 1u) GSWT_APPEND(h2, m2, cnt2, 2u) GSWT_APPEND(h3, m3, cnt3, 3u)
#undef GSWT_APPEND
    }
    const uint32_t n_mine = grp == 0u ? cnt0 : grp == 1u ? cnt1 : grp == 2u ? cnt2 : cnt3;
    const uint32_t n_max = max(max(cnt0, cnt1), max(cnt2, cnt3));
    GSWT_STAT_BATCH(n_max, cnt0, cnt1, cnt2, cnt3)
    if (GSWT_ABL(f, 1) || n_max == 0u) return;
    if (DPPW && !DEPTH && !COLF && !PK) {
        // ---- register-broadcast walk (measurement variant, -DGSWT_EXPERIMENTS only: 123-128 us against 95 us at c3, 312 against 242 at c3d) ----
        // In rounds of 16 steps: lane j of a 16-lane group loads the record of its list's entry 16 r + j ONCE (one list read + two
        // ds_read_b128 per lane and round, the next round's in flight during this one), and step j takes the record's dwords from lane j
        // through DPP row_newbcast:j, folded into the consuming instructions (v_fmac_f32_dpp, v_cvt_f32_ubyteN_dpp; three v_mov_b32_dpp
        // for the addends).  No LDS access and no LDS latency inside the steps: with the records fetched per step (a list entry, then two
        // ds_read_b128 that depend on it) a step lasted as long as that round trip under load, ~200 cycles for ~19 VALU instructions.
        // Same F4 sequence (v_fmac is the fused multiply-add of the scalar code), same blend order: the image is bit-identical.
        const uint32_t gi = lane & 15u;
        const uint32_t n_steps = (n_max + 1u) & ~1u;
        const uint32_t n_pad = ((n_max + 15u) & ~15u) + 16u;                 // whole rounds + the prefetched one
        for (uint32_t p = n_mine + gi; p < n_pad; p += 16u) my_list[p] = (uint16_t)(kNull * 16u);
        const char* const q0b = reinterpret_cast<const char*>(s_q0);
        const char* const q1b = reinterpret_cast<const char*>(s_q1);
        uint32_t e = my_list[gi];
        float4 c0 = *reinterpret_cast<const float4*>(q0b + e), c1 = *reinterpret_cast<const float4*>(q1b + e);
        const float nl2e = -1.4426950408889634f;
        float t0, t1, t2, t3;
#define GSWT_DSTEP(J)                                                                                          \
        asm volatile(                                                                                          \
            "v_mov_b32_dpp %[t0], %[r2] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"                       \
            "v_mov_b32_dpp %[t1], %[r6] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"                       \
            "v_fmac_f32_dpp %[t0], %[r1], %[ly] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"               \
            "v_fmac_f32_dpp %[t1], %[r5], %[ly] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"               \
            "v_fmac_f32_dpp %[t0], %[r0], %[lx] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"               \
            "v_fmac_f32_dpp %[t1], %[r4], %[lx] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"               \
            "v_mul_f32 %[t2], %[t0], %[t0]\n"                                                                  \
            "v_fmac_f32 %[t2], %[t1], %[t1]\n"                                                                 \
            "v_cmp_ge_f32 vcc, 4.0, %[t2]\n"                                                                   \
            "s_cbranch_vccz 1f\n"                                                                              \
            "v_mov_b32_dpp %[t3], %[r3] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"                       \
            "v_fmac_f32 %[t3], %[k], %[t2]\n"                                                                  \
            "v_exp_f32 %[t3], %[t3]\n"                                                                         \
            "v_cvt_f32_ubyte0_dpp %[t0], %[r7] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"                \
            "v_cvt_f32_ubyte1_dpp %[t1], %[r7] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"                \
            "v_cvt_f32_ubyte2_dpp %[t2], %[r7] row_newbcast:" #J " row_mask:0xf bank_mask:0xf\n"                \
            "v_cndmask_b32 %[t3], 0, %[t3], vcc\n"                                                             \
            "v_mul_f32 %[t3], %[T], %[t3]\n"                                                                   \
            "v_fmac_f32 %[ar], %[t3], %[t0]\n"                                                                 \
            "v_fmac_f32 %[ag], %[t3], %[t1]\n"                                                                 \
            "v_fmac_f32 %[ab], %[t3], %[t2]\n"                                                                 \
            "v_sub_f32 %[T], %[T], %[t3]\n"                                                                    \
            "1:\n"                                                                                             \
            : [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), [T] "+v"(T), [ar] "+v"(ar), [ag] "+v"(ag), [ab] "+v"(ab) \
            : [r0] "v"(c0.x), [r1] "v"(c0.y), [r2] "v"(c0.z), [r3] "v"(c0.w), [r4] "v"(c1.x), [r5] "v"(c1.y), [r6] "v"(c1.z), [r7] "v"(c1.w), \
              [lx] "v"(lx), [ly] "v"(ly), [k] "s"(nl2e)                                                        \
            : "vcc");
        for (uint32_t base = 0; base < n_steps; base += 16u) {
            e = my_list[base + 16u + gi];                                   // the next round's entry and record
            const float4 n0 = *reinterpret_cast<const float4*>(q0b + e), n1 = *reinterpret_cast<const float4*>(q1b + e);
            const uint32_t left = n_steps - base;                           // even, >= 2
            do {
                GSWT_DSTEP(0) GSWT_DSTEP(1) if (left <= 2u) break;
                GSWT_DSTEP(2) GSWT_DSTEP(3) if (left <= 4u) break;
                GSWT_DSTEP(4) GSWT_DSTEP(5) if (left <= 6u) break;
                GSWT_DSTEP(6) GSWT_DSTEP(7) if (left <= 8u) break;
                GSWT_DSTEP(8) GSWT_DSTEP(9) if (left <= 10u) break;
                GSWT_DSTEP(10) GSWT_DSTEP(11) if (left <= 12u) break;
                GSWT_DSTEP(12) GSWT_DSTEP(13) if (left <= 14u) break;
                GSWT_DSTEP(14) GSWT_DSTEP(15)
            } while (0);
            c0 = n0; c1 = n1;
        }
#undef GSWT_DSTEP
        if (EARLY && ballot64(T >= t_eps) == 0ull) wave_live = false;
        return;
    }
    // pad this group's list with the null record up to the wave's even step count (+2: the walk reads two entries ahead)
    const uint32_t n_steps = (n_max + 1u) & ~1u;
    for (uint32_t p = n_mine + (lane & 15u); p < n_steps + 2u; p += 16u) my_list[p] = (uint16_t)(kNull * 16u);
    // walk: one splat per 16-lane group per step.  Two-deep software pipeline, unrolled x2 so the two record register
    // sets (A, B) never need copying: the list entry is fetched two steps ahead, the record one step ahead.
    const char* const q0b = reinterpret_cast<const char*>(s_q0);
    const char* const q1b = reinterpret_cast<const char*>(s_q1);
    const char* const q2b = reinterpret_cast<const char*>(s_q2);
    const char* const dpb = reinterpret_cast<const char*>(s_dep);
#define GSWT_REC0(O) (*reinterpret_cast<const float4*>(q0b + (O)))
#define GSWT_REC1(O) (*reinterpret_cast<const float4*>(q1b + (O)))
#define GSWT_REC2(O) (*reinterpret_cast<const float4*>(q2b + (O)))
#define GSWT_RECD(O) (*reinterpret_cast<const float*>(dpb + ((O) >> 2)))
// PK (measurement variant, -DGSWT_EXPERIMENTS only): the staged record is q0 = (iu.x, iv.x, iu.y, iv.y), q1 = (-ku, -kv, log2 alpha, rgba8),
// so that F4's two coordinates are the two halves of packed operations -- (pu_y, pv_y) = v_pk_fma_f32((iu.y, iv.y), ly, (-ku, -kv)),
// (p.x, p.y) = v_pk_fma_f32((iu.x, iv.x), lx, (pu_y, pv_y)) -- and the blend pairs (red, green) and (T, blue): (T, ab) += wgt * (-1, cb).
// IEEE per component: the same image bit for bit, 15 VALU instructions per step instead of 19 -- and 5-7 % SLOWER (c3 104.6 against
// 98.9 us, c3d 259.8 against 243.8): four v_pk_fma_f32 cost more than the eight scalar instructions they replace.
#define GSWT_STEP(Q0, Q1, Q2, DV)                                                                   \
    if (PK && !DPPW) {                                                                              \
        v2f iy_, k_, ix_;                                                                           \
        iy_.x = Q0.z; iy_.y = Q0.w; k_.x = Q1.x; k_.y = Q1.y; ix_.x = Q0.x; ix_.y = Q0.y;           \
        const v2f pyv = pk_fma(iy_, pk_splat(ly), k_);                                              \
        const v2f pp = pk_fma(ix_, pk_splat(lx), pyv);                                              \
        const float r2 = fmaf(pp.y, pp.y, pp.x * pp.x);                                             \
        bool cover = r2 <= 4.0f;                                                                    \
        if (DEPTH) cover = cover && DV < dbuf;                                                      \
        if (ballot64(cover) != 0ull) {                                                              \
            const float e = __builtin_amdgcn_exp2f(fmaf(r2, -1.4426950408889634f, Q1.z));           \
            const float Bv = cover ? e : 0.0f;                                                      \
            const float wgt = T * Bv;                                                               \
            const uint32_t cw = __float_as_uint(Q1.w);                                              \
            v2f c01, acc01, acc23;                                                                  \
            c01.x = COLF ? Q2.x : (float)(cw & 0xFFu); c01.y = COLF ? Q2.y : (float)((cw >> 8) & 0xFFu); \
            mtb.y = COLF ? Q2.z : (float)((cw >> 16) & 0xFFu);                                      \
            acc01.x = ar; acc01.y = ag; acc23.x = T; acc23.y = ab;                                  \
            acc01 = pk_fma(pk_splat(wgt), c01, acc01);                                              \
            acc23 = pk_fma(pk_splat(wgt), mtb, acc23);                                              \
            ar = acc01.x; ag = acc01.y; T = acc23.x; ab = acc23.y;                                  \
        }                                                                                           \
    } else                                                                                          \
    {                                                                                               \
        const float pu_y = fmaf(Q0.y, ly, Q0.z);                                                    \
        const float pv_y = fmaf(Q1.y, ly, Q1.z);                                                    \
        const float ppx = fmaf(Q0.x, lx, pu_y);                                                     \
        const float ppy = fmaf(Q1.x, lx, pv_y);                                                     \
        const float r2 = fmaf(ppy, ppy, ppx * ppx);                                                 \
        bool cover = r2 <= 4.0f;                                                                    \
        if (DEPTH) cover = cover && DV < dbuf;                                                      \
        GSWT_STAT_STEP(cover)                                                                       \
        /* the blend runs under EXEC masking (s_and_saveexec on the coverage mask, skipped when no lane is covered): round 2 predicated it   \
           with a v_cndmask behind a wave-uniform ballot test; one VALU instruction less per step, 95.8 -> 93.0 us at c3 (round 3).           \
           (DPPW && PK: that older form, kept as a measurement variant of the -DGSWT_EXPERIMENTS build) */                                   \
        if ((DPPW && PK) ? ballot64(cover) != 0ull : cover) {                                       \
            const float e = __builtin_amdgcn_exp2f(fmaf(r2, -1.4426950408889634f, Q0.w));           \
            const float Bv = (DPPW && PK) ? (cover ? e : 0.0f) : e;                                 \
            const float wgt = T * Bv;                                                               \
            const uint32_t cw = __float_as_uint(Q1.w);                                              \
            ar = fmaf(wgt, COLF ? Q2.x : (float)(cw & 0xFFu), ar);                                  \
            ag = fmaf(wgt, COLF ? Q2.y : (float)((cw >> 8) & 0xFFu), ag);                           \
            ab = fmaf(wgt, COLF ? Q2.z : (float)((cw >> 16) & 0xFFu), ab);                          \
            T = T - wgt;                                                                            \
        }                                                                                           \
    }
    v2f mtb; mtb.x = -1.0f; mtb.y = 0.0f;        // (-1, blue): T - wgt = fma(wgt, -1, T) exactly
    {
        // (every list entry goes through an empty asm as soon as it is loaded: carried around the loop as a 16-bit value, the compiler
        // masks each one again before using it as an LDS address -- one VALU instruction per step; a 32-bit register it cannot look
        // into stays as ds_read_u16 left it.  c3 97.8 -> 96.5 us under stage events, c3d 450 -> 447.)
        uint32_t kA = my_list[0], kB = my_list[1];
        asm("" : "+v"(kA)); asm("" : "+v"(kB));
        float4 a0 = GSWT_REC0(kA), a1 = GSWT_REC1(kA);
        float4 a2 = make_float4(0.f, 0.f, 0.f, 0.f), b2 = a2;
        if (COLF) a2 = GSWT_REC2(kA);
        float da = DEPTH ? GSWT_RECD(kA) : 0.0f, db = 0.0f;
        for (uint32_t i = 0; i < n_steps; i += 2u) {
            kA = my_list[i + 2u];                                          // entry of step i+2: issued BEFORE the record reads, so that pinning it
            const float4 b0 = GSWT_REC0(kB), b1 = GSWT_REC1(kB);           // (below) waits for the oldest LDS read only, not for the records behind it
            if (COLF) b2 = GSWT_REC2(kB);
            if (DEPTH) db = GSWT_RECD(kB);
            asm("" : "+v"(kA));
            GSWT_STEP(a0, a1, a2, da)
            kB = my_list[i + 3u];                                          // entry of step i+3
            a0 = GSWT_REC0(kA); a1 = GSWT_REC1(kA);                        // record of step i+2
            if (COLF) a2 = GSWT_REC2(kA);
            if (DEPTH) da = GSWT_RECD(kA);
            asm("" : "+v"(kB));
            GSWT_STEP(b0, b1, b2, db)
        }
    }
#undef GSWT_STEP
#undef GSWT_REC0
#undef GSWT_REC1
#undef GSWT_REC2
#undef GSWT_RECD
    // Saturated pixels keep accumulating weights below t_eps (the oracle has no cut at all); the early-out is per wave
    // and per batch: once no pixel of the strip has T >= t_eps the wave stops binning and walking.
    if (EARLY && ballot64(T >= t_eps) == 0ull) wave_live = false;
}

// Measured and dropped in the walk (all bit-identical, c3, base 97 us):
//  * records through ONE ds_read_b32 per two steps + DPP row_newbcast of each dword (LDS cycles per step 9 -> 2): the
//    compiler folds the broadcast into v_cvt_f32_ubyteN but not into v_fmac, +7 VALU per step, 135 us.  LDS-array
//    cycles are not what bounds the walk; every VALU instruction added to a step costs ~5 us of kernel.
//  * geometry A, geometry B, one skip test on (cover A | cover B), blend A, blend B in one basic block (more ILP:
//    tools/ubench/valu_issue.hip measures 4.8 cycles per dependent v_fma per SIMD at 8 waves, 2.9 with two independent
//    chains): the compiler sinks the record prefetch to the loop top and splits the .w dwords into extra ds_read_b32
//    inside the blend, exposing two LDS latencies per iteration: 126 us.  Without any skip test: 97 us (no gain).  With
//    the pipeline restored by hand (four record sets ping-pong, the next PAIR's records in flight, steps padded to a
//    multiple of 4): the intended schedule in the ISA, but 81 VGPRs = 5 waves per SIMD: 105 us.
//  * XCD-grouped item order (segments of a tile and x-neighbour tiles on one XCD, groups of 2..16): 96.4 - 97.7 us; the
//    record gathers are not what bounds the kernel either.
//  (Occupancy matters here: one workgroup per CU LESS -- LDS padded -- costs 5 % of the frame rate.  The eighth came from the
//  pixel boxes as four signed bytes (LDS 20.8 -> 17.9 KB) plus, to get from 69 to 63 VGPRs without spills: the prefetched record
//  trimmed to the ten words the staging uses, the tile origin and the wave index through readfirstlane (SGPRs), and the output
//  pixel recomputed at the end from an opaque copy of the thread id.  c3 3990 -> 4078, c3h 3555 -> 3685 frames/s on one box.
//  The depth / float-colour variants need 71-72 VGPRs and stay at 7 waves: forcing 8 spilled loop invariants to scratch.)
// Measured and dropped: the same compositor as a PERSISTENT grid (one workgroup walks many items, the gathers of the next
// item's first batch in flight during the current item's walk; items dealt by weight class, boustrophedon, so that the
// busiest workgroup is 3 % above the mean).  Bit-identical output, 142 us against 114 us: the time goes with the number
// of resident workgroups (2 / 4 / 6 per CU: 262 / 167 / 143 us), i.e. the kernel is bound by the walk's per-wave
// progress (VALU 50 us + LDS 53 us + SALU 42 us of issue that do not overlap perfectly, two barriers per batch), and
// the gather latency was already covered by the other workgroups of the CU.  "stage-only 58 us" in the ablation is
// what staging costs with nothing to hide behind, not a serial share of the full kernel.

// FOLD (GSWT_OPT_FOLD_COMBINE, round 4): no k_combine behind the compositor.  (a) k_items hands out an empty work item for every tile
// without pairs: its background is written here.  (b) The segments of a long tile list are folded by whichever of their workgroups finishes
// LAST: every segment stores its partial (C, T) with agent-scope (sc1) stores -- the L2 of an XCD is not coherent with the other seven
// inside a kernel, and the segments of one tile run on different XCDs --, waits for them, and takes a ticket on the tile's counter; the
// workgroup that draws the last ticket reads all partials back (sc1 loads) and folds them front to back in segment order, exactly as
// k_combine does: the image is bit-identical whichever workgroup that is.  (c) Workgroup 0 publishes the frame's counters to the host.
// One ticket per multi-segment work item on ITS tile's word: ~700 atomics per c3 frame on ~250 addresses (a single frame-wide ticket
// word would serialise at ~8 ns per atomic on the memory side).
template <bool EARLY, bool DEPTH, bool COLF, bool PK, bool DPPW, bool FOLD = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((DEPTH || COLF) ? 7 : 8, 8))) void k_composite(const Frame f, const uint2* __restrict__ ranges,
                                                   const uint32_t* __restrict__ item_base, const uint4* __restrict__ item_tab,
                                                   uint32_t seg, const uint32_t* __restrict__ vals, const Rec* __restrict__ recs,
                                                   const float* __restrict__ depths, const float4* __restrict__ col_f,
                                                   const float4* __restrict__ bg_rgba, const float* __restrict__ bg_depth,
                                                   float4* __restrict__ out, float4* __restrict__ partials,
                                                   int n_tiles, int out_rows, uint32_t* __restrict__ tile_tick,
                                                   const unsigned long long* __restrict__ counters, unsigned long long* __restrict__ host_counters)
{
    __shared__ uint32_t s_last;
    __shared__ float4 s_q0[257], s_q1[257];                     // [256] = the null record (list padding)
    __shared__ uint32_t s_bb[256];                              // the 16 sub-blocks a staged pair's pixel box touches (bit 4 strip + column group)
    __shared__ float4 s_q2[COLF ? 257 : 1];
    __shared__ float s_dep[DEPTH ? 257 : 1];
    __shared__ uint16_t s_list[4][4][kListStride];     // [wave][sub-block][i] -> LDS byte offset of the i-th hit's record
    __shared__ uint4 s_dead;                           // EARLY: .x .. .w = wave 0 .. 3 has no pixel with T >= t_eps left
    // work item -> (tile, segment) through the table k_items left behind.  Consecutive items are dealt
    // round-robin over the 8 XCDs by the dispatcher, which balances the skewed tile-list lengths.
    const uint32_t item = blockIdx.x;
#ifdef GSWT_TRACE
    const bool tr_on = threadIdx.x == 0 && item < kTraceItems;
    const uint32_t tr_item = item;
    unsigned long long tr_walk = 0;
    bool tr_first = true;
    GSWT_TR(0, GSWT_NOW())
#endif
    const uint32_t n_items = item_base[n_tiles];
    const uint4 it = item_tab[item];                 // (in flight together with n_items; garbage past n_items, unused)
    // FOLD: this is the frame's last kernel -- the four result counters go straight into the slot's pinned host words
    if (FOLD && item == 0u && threadIdx.x < 5u && host_counters) host_counters[threadIdx.x] = counters[threadIdx.x];
    if (item >= n_items) return;
    GSWT_TR(1, GSWT_NOW())
    GSWT_TR(4, it.w - it.z)
#ifdef GSWT_TRACE
    { unsigned hwid, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); GSWT_TR(7, (unsigned long long)hwid | ((unsigned long long)xcc << 32)) }
#endif
    const int tile = (int)it.x;
    const bool multi_seg = (it.y & 1u) != 0u;
    int tx, tyl;
    tile_xy(f, (uint32_t)tile, tx, tyl);
    const int sc = f.shard_count <= 1 ? 1 : f.shard_count;
    const int ty = tyl * sc + (sc > 1 ? f.shard_index : 0);
    const int bx = (tx + f.col0) * kTile, by = ty * kTile;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));      // uniform per wave: everything derived from it stays scalar
    const uint32_t grp = lane >> 4, gi = lane & 15u;                 // wave = 16 x 4 strip, 16-lane group = 4 x 4 sub-block
    const int lxi = (int)grp * 4 + (int)(gi & 3u), lyi = (int)wave * 4 + (int)(gi >> 2);
    const int px = bx + lxi, py = by + lyi;
    const bool inside = px < f.width && py < f.height;
    const float lx = (float)lxi + 0.5f, ly = (float)lyi + 0.5f;
    // tile origin as floats: uniform, kept in SGPRs (a v_cvt of a scalar would park them in two VGPRs for the whole kernel)
    const float fbx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)bx)));
    const float fby = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)by)));
    // F3 (sequence v3): centre - tile origin = fma(W/2, ndc.x, W/2 - bx), fma(-H/2, ndc.y, H/2 - by): ONE rounding at the magnitude of the
    // offset; W/2 - bx and H/2 - by are exact and uniform (SGPRs)
    const float hW = 0.5f * f.W, hHn = -0.5f * f.H;
    const float c0x = hW - fbx, c0y = 0.5f * f.H - fby;
    uint16_t* const wlist = &s_list[wave][0][0];
    const CompLane cl = {lx, ly, (int)(wave * 4u), lane, grp};
    const uint2 rg = make_uint2(it.z, it.w);         // this item's slice of the tile's pair list
    // Transmittance doubles as the "still active" state: a lane is live while T >= t_eps.  Pixels
    // outside the target start at T = 0 when early-out is on (never live); with t_eps = 0 they just
    // accumulate and are never stored.
    float T = (EARLY && !inside) ? 0.0f : 1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f;
    float dbuf = 1.0f;
    if (DEPTH && inside) dbuf = bg_depth[(size_t)py * f.width + px];
    const float t_eps = f.t_eps;
    bool wave_live = true;
    // the null record: p.x = +inf for every pixel (0 * l + inf), so r^2 = +inf and no pixel is ever inside
    if (tid == 0) {
        if (EARLY) s_dead = make_uint4(0u, 0u, 0u, 0u);
        s_q0[kNullRec] = (PK && !DPPW) ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(0.f, 0.f, __builtin_inff(), 0.f);
        s_q1[kNullRec] = (PK && !DPPW) ? make_float4(__builtin_inff(), 0.f, 0.f, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (DEPTH) s_dep[DEPTH ? kNullRec : 0u] = 0.0f;
        if (COLF) s_q2[COLF ? kNullRec : 0u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // Software-pipelined gather: the records of batch b+1 and the slot indices of batch b+2 are in flight
    // while batch b is binned and walked (two dependent HBM latencies per batch otherwise sit between barriers).
    // The loads are unconditional with clamped indices (lanes past the end re-read the last pair and never stage
    // it): a load under a lane mask would be merged back through register copies that wait for it on the spot.
    // (only the words the staging needs are kept in registers: the depth word rides along with a depth buffer only, the
    // pad word of the third quad never)
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rd = ra;
    float rbw = 0.f;
#define GSWT_LOAD_REC(SLOT)                                                                          \
        {                                                                                           \
            const float4* rp = reinterpret_cast<const float4*>(recs + (SLOT));                      \
            ra = rp[0];                                                                             \
            rb = rp[1];                                                                             \
            if (DEPTH) rbw = depths[(SLOT)];                                                        \
        }
    uint32_t slot_nxt = 0;
    const uint32_t last_pair = rg.y - 1u;
    // profiling ablations of the gather chain (output is wrong): 32 = no `vals` level (the pair index stands in for the slot),
    // 64 = no record gather (every pair reads record 0)
    const bool abl_vals = GSWT_ABL(f, 32), abl_recs = GSWT_ABL(f, 64);
#define GSWT_VAL(I) (abl_vals ? (I) : vals[(I)])
    if (rg.x < rg.y && !GSWT_ABL(f, 4)) {
        uint32_t slot0 = GSWT_VAL(min(rg.x + tid, last_pair));
        if (abl_recs) slot0 = 0u;
        GSWT_LOAD_REC(slot0)
        if (COLF) rd = col_f[slot0];
        slot_nxt = GSWT_VAL(min(rg.x + 256u + tid, last_pair));
    }
    for (uint32_t base = rg.x; base < rg.y; base += 256u) {
        const uint32_t n = min(256u, rg.y - base);
        if (GSWT_ABL(f, 4)) break;                       // ablation: no staging at all
        if (tid < n) {
            // F3: per-(splat, tile) constants
            const float ox = fmaf(hW, rb.x, c0x), oy = fmaf(hHn, rb.y, c0y);
            const float nku = -fmaf(ra.x, ox, ra.y * oy);
            const float nkv = -fmaf(ra.z, ox, ra.w * oy);
            const float l2a = __builtin_amdgcn_logf(rb.z);                           // v_log_f32 = log2; log2(0) = -inf -> B = 0
            s_q0[tid] = (PK && !DPPW) ? make_float4(ra.x, ra.z, ra.y, ra.w) : make_float4(ra.x, ra.y, nku, l2a);
            s_q1[tid] = (PK && !DPPW) ? make_float4(nku, nkv, l2a, rb.w) : make_float4(ra.z, ra.w, nkv, rb.w);
            // Pixel half extents of |p| <= 2 from the inverse map: the quad axes are u = iu / |iu|^2, w = iv / |iv|^2 and the
            // box is 2 sqrt(u.x^2 + w.x^2) by 2 sqrt(u.y^2 + w.y^2).  Approximate reciprocals / roots (1 ulp) under a 1e-4
            // relative + 2e-3 px margin: the box only has to CONTAIN every pixel centre with r^2 <= 4 (it decides which
            // sub-block lists a pair enters, never a pixel's coverage), and k_project's own box decided the pair's tiles.
            const float ria = __builtin_amdgcn_rcpf(fmaf(ra.y, ra.y, ra.x * ra.x)), rib = __builtin_amdgcn_rcpf(fmaf(ra.w, ra.w, ra.z * ra.z));
            const float qux = ra.x * ria, quy = ra.y * ria, qwx = ra.z * rib, qwy = ra.w * rib;
            const float bhx = fmaf(2.0f * __builtin_amdgcn_sqrtf(fmaf(qwx, qwx, qux * qux)), 1.0001f, 0.002f);
            const float bhy = fmaf(2.0f * __builtin_amdgcn_sqrtf(fmaf(qwy, qwy, quy * quy)), 1.0001f, 0.002f);
            // Pixel box, tile-local, as the integer pixel ranges whose centres it holds: x_lo <= k + 0.5  <=>  ceil(x_lo - 0.5) <= k and
            // x_hi >= k + 0.5  <=>  floor(x_hi - 0.5) >= k for every integer k (x - 0.5 is exact in binary32 wherever the
            // outcome can depend on it), so the byte form bins exactly like the float box did; 16 -> 4 bytes of LDS per pair.
            // Clamped to [-2, 17]: only 0..15 are ever compared.
            const int xa = min(max((int)ceilf((ox - bhx) - 0.5f), -2), 17), xb = min(max((int)floorf((ox + bhx) - 0.5f), -2), 17);
            const int ya = min(max((int)ceilf((oy - bhy) - 0.5f), -2), 17), yb = min(max((int)floorf((oy + bhy) - 0.5f), -2), 17);
            // The 16 sub-blocks the box touches, as a bit mask (bit 4 s + g: strip s, column group g), computed ONCE here instead of four
            // column and one row test per wave and pair in the bin loop: columns g with 4 g + 3 >= xa and 4 g <= xb are g_lo .. g_hi,
            // g_lo = max(xa >> 2, 0), g_hi = min(xb >> 2, 3) (arithmetic shifts: xa, xb in [-2, 17]); rows likewise; the outer product of the
            // two 4-bit masks is one multiplication (the row bits spread to positions 0, 4, 8, 12: no carries).
            {
                const int gx0 = max(xa >> 2, 0), gx1 = min(xb >> 2, 3), gy0 = max(ya >> 2, 0), gy1 = min(yb >> 2, 3);
                const uint32_t cx = gx1 >= gx0 ? (2u << gx1) - (1u << gx0) : 0u;
                const uint32_t ry = gy1 >= gy0 ? (2u << gy1) - (1u << gy0) : 0u;
                const uint32_t spread = (ry & 1u) | ((ry & 2u) << 3) | ((ry & 4u) << 6) | ((ry & 8u) << 9);
                s_bb[tid] = cx * spread;
            }
            if (DEPTH) s_dep[tid] = rbw;
            if (COLF) s_q2[tid] = rd;
        }
        __syncthreads();
#ifdef GSWT_TRACE
        if (tr_first) { GSWT_TR(2, GSWT_NOW()) tr_first = false; }
        const unsigned long long tr_t0 = GSWT_NOW();
#endif
        {
            if (abl_recs) slot_nxt = 0u;
            GSWT_LOAD_REC(slot_nxt)
            if (COLF) rd = col_f[slot_nxt];
            slot_nxt = GSWT_VAL(min(base + 512u + tid, last_pair));
        }
        if (wave_live && !GSWT_ABL(f, 2))               // ablation bit 2: stage only
            composite_bin_walk<EARLY, DEPTH, COLF, PK, DPPW>(f, cl, n, s_q0, s_q1, s_q2, s_bb, s_dep, wlist, dbuf, t_eps, T, ar, ag, ab, wave_live);
#ifdef GSWT_TRACE
        tr_walk += GSWT_NOW() - tr_t0;
#endif
        if (base + 256u >= rg.y) break;                     // last batch of the item: nothing is staged after it, no barrier needed
        // EARLY: the item ends once all four strips are saturated.  A saturated wave leaves a flag in LDS in front of the batch's closing
        // barrier and everybody reads the four flags behind it (flags are written between a batch's two barriers and read between
        // batches: no race; they never go back to 0).  Until the end of round 4 this was __syncthreads_and(): a workgroup reduction with
        // a second barrier per batch, which made the early-out cost 1.5 % of c3's frame rate, where it saves nothing.
        if (EARLY && !wave_live && lane == 0u) (&s_dead.x)[wave] = 1u;
        __syncthreads();
        if (EARLY) {
            const uint4 dd = s_dead;
            if (__builtin_amdgcn_readfirstlane((int)(dd.x & dd.y & dd.z & dd.w)) != 0) break;
        }
    }
    GSWT_TR(3, GSWT_NOW())
    GSWT_TR(6, tr_walk)
    const float k255 = 1.0f / 255.0f;      // colour is continuous: sum(w * byte) / 255 vs sum(w * (byte / 255)) differ in the last bits only
    if (!COLF) { ar *= k255; ag *= k255; ab *= k255; }
    // (the partial's slot: the tile's first + the segment's number -- the table position only while the table is in tile order)
    const uint32_t pslot = multi_seg ? item_base[tile] + (it.y >> 1) : 0u;
    if (multi_seg && !FOLD) {
        // partial (C, T) of this segment; k_combine folds the segments front to back
        partials[(size_t)pslot * 256u + tid] = make_float4(ar, ag, ab, T);
        return;
    }
    if (multi_seg && FOLD) {
        float* const pp = reinterpret_cast<float*>(partials + (size_t)pslot * 256u + tid);
        __hip_atomic_store(pp + 0, ar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 1, ag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 2, ab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 3, T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_s_waitcnt(0);                   // this thread's partial has been written through
        __syncthreads();                                 // ... and so have the other 255
        const uint32_t i0 = item_base[tile], n_seg = item_base[tile + 1] - i0;
        if (tid == 0) s_last = __hip_atomic_fetch_add(&tile_tick[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_seg - 1u ? 1u : 0u;
        __syncthreads();
        if (s_last == 0u) return;
        // the last segment to finish folds them all, front to back: (C1, T1) o (C2, T2) = (C1 + T1 C2, T1 T2) -- k_combine's loop
        T = 1.0f; ar = 0.0f; ag = 0.0f; ab = 0.0f;
        for (uint32_t s0 = 0; s0 < n_seg; s0 += 4u) {
            float4 p[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) {
                const float* q = reinterpret_cast<const float*>(partials + (size_t)(i0 + min(s0 + k, n_seg - 1u)) * 256u + tid);
                p[k].x = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                p[k].y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                p[k].z = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                p[k].w = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (uint32_t k = 0; k < 4u; k++) {
                if (s0 + k < n_seg) {                     // workgroup-uniform
                    ar = fmaf(T, p[k].x, ar);
                    ag = fmaf(T, p[k].y, ag);
                    ab = fmaf(T, p[k].z, ab);
                    T = T * p[k].w;
                }
            }
        }
    }
    // pixel coordinates again, from a copy of the thread id the compiler cannot connect to the one above: otherwise px, py
    // and the output row stay in VGPRs across the whole walk (the kernel sits exactly at the 64-VGPR / 8-wave limit)
    uint32_t tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lxi2 = (int)(((tid2 & 63u) >> 4) * 4u + (tid2 & 3u)), lyi2 = (int)((tid2 >> 6) * 4u + ((tid2 & 15u) >> 2));
    const int px2 = bx + lxi2, py2 = by + lyi2;
    if (px2 < f.width && py2 < f.height) {
        const int px = px2, py = py2, lyi = lyi2;
        float4 bg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bg_rgba) bg = bg_rgba[(size_t)py * f.width + px];
        float4 o;
        o.x = fmaf(T, bg.x, ar);
        o.y = fmaf(T, bg.y, ag);
        o.z = fmaf(T, bg.z, ab);
        o.w = fmaf(T, bg.w, 1.0f - T);
        const int orow = tyl * kTile + lyi;      // compacted row inside the shard image
        if (orow < out_rows) out[(size_t)orow * f.out_w + (px - f.out_x0)] = o;
    }
}

// Measured and dropped again in round 2 (the item STREAM: a resident grid of workgroups, each taking items b, b + G, b + 2G, ...
// as single 256-pair batches with the next item's records and the one-after-next item's slot indices in flight while it bins
// and walks the current one -- the batch pipeline above carried across items; bit-identical output): c3 120 us at 7 workgroups
// per CU (72 VGPRs), 133 us at 8 (64 VGPRs, 20 spilled), 146 / 174 us at 4 per CU, against 96 us for one workgroup per item.
// The phases of this kernel add up (round-2 ablations at c3: 15 us launch + output, 28 us record gathers, 2.5 us binning,
// 50 us walk), but hiding the gathers behind another item's walk inside a workgroup is not what makes them overlap: the
// dispatcher's dynamic hand-out of ~16 k short workgroups does that better than a static stride over them.

// ------------------------------------------------------------------------------------
// k_composite_dw: the compositor with DECOUPLED waves (round 4; GSWT_OPT_COMPOSITE = 1).  Same work items, same lane -> pixel map, same
// F3 / F4 / blend order (bit-identical image), but no workgroup barrier inside an item:
//   * batches of 128 pairs go through a ring of three LDS buffers;
//   * batch k is staged by two waves (waves 0, 1 stage the even batches, waves 2, 3 the odd ones: 64 pairs each, the record gathered
//     into registers two batches of its own earlier, as k_composite's staging lane does);
//   * per buffer two monotonic LDS counters: `staged` (half batches written into it, ever) and `done` (wave walks finished on it, ever).
//     A wave walks batch k once staged[k % 3] >= 2 (k / 3 + 1); a buffer is refilled with batch k once done[k % 3] >= 4 (k / 3), i.e.
//     once all four waves have walked batch k - 3.  A wave stages up to two batches ahead of its own walk without waiting (and waits only
//     for a batch it is about to walk itself), so the four waves of an item run up to two batches apart instead of meeting at two
//     s_barriers per 256 pairs -- k_composite idles 41 % of its wave-slots there, a batch lasting as long as its longest sub-block list.
//   Waiting = polling an LDS word with s_sleep between reads.  EARLY: a wave whose strip is saturated keeps staging and counting, skips
//   its walks; when all four are, every wave leaves at its next poll.
// LDS 18.4 KB (k_composite: 17.9 KB): 8 workgroups per CU.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lds_peek(const uint32_t* p) { return *reinterpret_cast<const volatile uint32_t*>(p); }
#ifndef GSWT_DW_SLEEP
#define GSWT_DW_SLEEP 1        // s_sleep units (64 clocks) between two polls of a counter
#endif

template <bool EARLY, bool DEPTH, bool COLF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(COLF ? 6 : DEPTH ? 7 : 8, 8))) void k_composite_dw(const Frame f,
                                                   const uint32_t* __restrict__ item_base, const uint4* __restrict__ item_tab,
                                                   const uint32_t* __restrict__ vals, const Rec* __restrict__ recs,
                                                   const float* __restrict__ depths, const float4* __restrict__ col_f,
                                                   const float4* __restrict__ bg_rgba, const float* __restrict__ bg_depth,
                                                   float4* __restrict__ out, float4* __restrict__ partials,
                                                   int n_tiles, int out_rows)
{
    constexpr uint32_t B = 128u, NBUF = 3u, BS = B + 1u;        // pairs per batch, ring depth, records per buffer (the last one = the null record)
    __shared__ float4 s_q0[NBUF * BS], s_q1[NBUF * BS];
    __shared__ uint32_t s_bb[NBUF * B];
    __shared__ float4 s_q2[COLF ? NBUF * BS : 1];
    __shared__ float s_dep[DEPTH ? NBUF * BS : 1];
    __shared__ uint16_t s_list[4][4][B + 8u];
    __shared__ uint32_t s_staged[NBUF], s_done[NBUF], s_dead;
    const uint32_t item = blockIdx.x;
    const uint32_t n_items = item_base[n_tiles];
    const uint4 it = item_tab[item];
    if (item >= n_items) return;
    const int tile = (int)it.x;
    const bool multi_seg = (it.y & 1u) != 0u;
    int tx, tyl;
    tile_xy(f, (uint32_t)tile, tx, tyl);
    const int sc = f.shard_count <= 1 ? 1 : f.shard_count;
    const int ty = tyl * sc + (sc > 1 ? f.shard_index : 0);
    const int bx = (tx + f.col0) * kTile, by = ty * kTile;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t grp = lane >> 4, gi = lane & 15u;
    const int lxi = (int)grp * 4 + (int)(gi & 3u), lyi = (int)wave * 4 + (int)(gi >> 2);
    const int px = bx + lxi, py = by + lyi;
    const bool inside = px < f.width && py < f.height;
    const float lx = (float)lxi + 0.5f, ly = (float)lyi + 0.5f;
    const float fbx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)bx)));
    const float fby = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)by)));
    const float hW = 0.5f * f.W, hHn = -0.5f * f.H;
    const float c0x = hW - fbx, c0y = 0.5f * f.H - fby;
    uint16_t* const wlist = &s_list[wave][0][0];
    const CompLane cl = {lx, ly, (int)(wave * 4u), lane, grp};
    const uint2 rg = make_uint2(it.z, it.w);
    float T = (EARLY && !inside) ? 0.0f : 1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f;
    float dbuf = 1.0f;
    if (DEPTH && inside) dbuf = bg_depth[(size_t)py * f.width + px];
    const float t_eps = f.t_eps;
    bool wave_live = true, counted_dead = false;
    if (tid < NBUF) {
        s_staged[tid] = 0u; s_done[tid] = 0u;
        s_q0[tid * BS + B] = make_float4(0.f, 0.f, __builtin_inff(), 0.f);      // the null record of buffer tid: r^2 = +inf for every pixel
        s_q1[tid * BS + B] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (DEPTH) s_dep[DEPTH ? tid * BS + B : 0u] = 0.0f;
        if (COLF) s_q2[COLF ? tid * BS + B : 0u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid == 0) s_dead = 0u;
    const uint32_t n_pairs = rg.y - rg.x;
    const uint32_t nb = (n_pairs + B - 1u) / B;
    const uint32_t last_pair = rg.y - 1u;
    const uint32_t half = wave & 1u;
    // the batch this wave stages next: ks = (wave >> 1), + 2, + 2, ...; (sb, su) = (ks % 3, ks / 3) kept incrementally
    uint32_t ks = wave >> 1, sb = wave >> 1, su = 0u;
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra, rd = ra;
    float rbw = 0.f;
    uint32_t slot_nxt = 0;
#define GSWT_DW_PAIR(KB) min(rg.x + (KB) * B + half * 64u + lane, last_pair)
#define GSWT_DW_LOAD(SLOT)                                                                           \
        {                                                                                           \
            const float4* rp = reinterpret_cast<const float4*>(recs + (SLOT));                      \
            ra = rp[0];                                                                             \
            rb = rp[1];                                                                             \
            if (DEPTH) rbw = depths[(SLOT)];                                                        \
            if (COLF) rd = col_f[(SLOT)];                                                           \
        }
    if (n_pairs) {
        const uint32_t slot0 = vals[GSWT_DW_PAIR(ks)];
        GSWT_DW_LOAD(slot0)
        slot_nxt = vals[GSWT_DW_PAIR(ks + 2u)];
    }
    __syncthreads();                                    // the counters and the null records are in place: the only barrier of the item
    uint32_t kb = 0u, ku = 0u;                           // (k % 3, k / 3) of the batch this wave walks next
    bool all_dead = false;
    for (uint32_t k = 0; k < nb && !all_dead; k++) {
        // ---- staging duties: everything up to two batches ahead of my walk whose buffer is free; a batch I am about to walk is waited for
        while (ks < nb && ks <= k + 2u) {
            const uint32_t need = 4u * su;
            if (lds_peek(&s_done[sb]) < need) {
                if (ks > k) break;                      // not needed yet: try again after the next walk
                while (lds_peek(&s_done[sb]) < need) {
                    if (EARLY && lds_peek(&s_dead) == 4u) { all_dead = true; break; }
                    __builtin_amdgcn_s_sleep(GSWT_DW_SLEEP);
                }
                if (all_dead) break;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const uint32_t li = half * 64u + lane;      // my pair inside the batch
            if (ks * B + li < n_pairs) {
                const uint32_t at = sb * BS + li;
                const float ox = fmaf(hW, rb.x, c0x), oy = fmaf(hHn, rb.y, c0y);
                const float nku = -fmaf(ra.x, ox, ra.y * oy);
                const float nkv = -fmaf(ra.z, ox, ra.w * oy);
                const float l2a = __builtin_amdgcn_logf(rb.z);
                s_q0[at] = make_float4(ra.x, ra.y, nku, l2a);
                s_q1[at] = make_float4(ra.z, ra.w, nkv, rb.w);
                const float ria = __builtin_amdgcn_rcpf(fmaf(ra.y, ra.y, ra.x * ra.x)), rib = __builtin_amdgcn_rcpf(fmaf(ra.w, ra.w, ra.z * ra.z));
                const float qux = ra.x * ria, quy = ra.y * ria, qwx = ra.z * rib, qwy = ra.w * rib;
                const float bhx = fmaf(2.0f * __builtin_amdgcn_sqrtf(fmaf(qwx, qwx, qux * qux)), 1.0001f, 0.002f);
                const float bhy = fmaf(2.0f * __builtin_amdgcn_sqrtf(fmaf(qwy, qwy, quy * quy)), 1.0001f, 0.002f);
                const int xa = min(max((int)ceilf((ox - bhx) - 0.5f), -2), 17), xb = min(max((int)floorf((ox + bhx) - 0.5f), -2), 17);
                const int ya = min(max((int)ceilf((oy - bhy) - 0.5f), -2), 17), yb = min(max((int)floorf((oy + bhy) - 0.5f), -2), 17);
                const int gx0 = max(xa >> 2, 0), gx1 = min(xb >> 2, 3), gy0 = max(ya >> 2, 0), gy1 = min(yb >> 2, 3);
                const uint32_t cx = gx1 >= gx0 ? (2u << gx1) - (1u << gx0) : 0u;
                const uint32_t ry = gy1 >= gy0 ? (2u << gy1) - (1u << gy0) : 0u;
                const uint32_t spread = (ry & 1u) | ((ry & 2u) << 3) | ((ry & 4u) << 6) | ((ry & 8u) << 9);
                s_bb[sb * B + li] = cx * spread;
                if (DEPTH) s_dep[DEPTH ? at : 0u] = rbw;
                if (COLF) s_q2[COLF ? at : 0u] = rd;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0u) atomicAdd(&s_staged[sb], 1u);
            ks += 2u; sb += 2u; if (sb >= 3u) { sb -= 3u; su += 1u; }
            GSWT_DW_LOAD(slot_nxt)                       // the record of my next staging batch, the slot index of the one after it
            slot_nxt = vals[GSWT_DW_PAIR(ks + 2u)];
        }
        if (all_dead) break;
        // ---- wait for batch k, walk it
        {
            const uint32_t target = 2u * (ku + 1u);
            while (lds_peek(&s_staged[kb]) < target) {
                if (EARLY && lds_peek(&s_dead) == 4u) { all_dead = true; break; }
                __builtin_amdgcn_s_sleep(GSWT_DW_SLEEP);
            }
            if (all_dead) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        const uint32_t n = min(B, n_pairs - k * B);
        if (wave_live)
            composite_bin_walk<EARLY, DEPTH, COLF, false, false, B>(f, cl, n, s_q0 + kb * BS, s_q1 + kb * BS, s_q2 + (COLF ? kb * BS : 0u), s_bb + kb * B,
                                                                    s_dep + (DEPTH ? kb * BS : 0u), wlist, dbuf, t_eps, T, ar, ag, ab, wave_live);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0u) {
            atomicAdd(&s_done[kb], 1u);
            if (EARLY && !wave_live && !counted_dead) atomicAdd(&s_dead, 1u);
        }
        if (EARLY && !wave_live) counted_dead = true;
        kb += 1u; if (kb >= 3u) { kb = 0u; ku += 1u; }
        if (EARLY && lds_peek(&s_dead) == 4u) all_dead = true;
    }
#undef GSWT_DW_PAIR
#undef GSWT_DW_LOAD
    const float k255 = 1.0f / 255.0f;
    if (!COLF) { ar *= k255; ag *= k255; ab *= k255; }
    if (multi_seg) {
        partials[(size_t)(item_base[tile] + (it.y >> 1)) * 256u + tid] = make_float4(ar, ag, ab, T);      // (k_composite: pslot)
        return;
    }
    uint32_t tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lxi2 = (int)(((tid2 & 63u) >> 4) * 4u + (tid2 & 3u)), lyi2 = (int)((tid2 >> 6) * 4u + ((tid2 & 15u) >> 2));
    const int px2 = bx + lxi2, py2 = by + lyi2;
    if (px2 < f.width && py2 < f.height) {
        float4 bg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bg_rgba) bg = bg_rgba[(size_t)py2 * f.width + px2];
        float4 o;
        o.x = fmaf(T, bg.x, ar);
        o.y = fmaf(T, bg.y, ag);
        o.z = fmaf(T, bg.z, ab);
        o.w = fmaf(T, bg.w, 1.0f - T);
        const int orow = tyl * kTile + lyi2;
        if (orow < out_rows) out[(size_t)orow * f.out_w + (px2 - f.out_x0)] = o;
    }
}

#ifdef GSWT_EXPERIMENTS
#include "gswt_composite_exp.hip"      // compositor variants of round 3 (measured, not shipped)
#endif

// Folds the per-segment partials of multi-segment tiles: (C1,T1) o (C2,T2) = (C1 + T1*C2, T1*T2).
// One workgroup per tile, same lane -> pixel mapping as k_composite.
__global__ __launch_bounds__(256) void k_combine(const Frame f, const uint32_t* __restrict__ item_base,
                                                 const float4* __restrict__ partials, const float4* __restrict__ bg_rgba,
                                                 float4* __restrict__ out, int n_tiles, int out_rows,
                                                 const unsigned long long* __restrict__ counters, unsigned long long* __restrict__ host_counters)
{
    const int tile = blockIdx.x;
    // last kernel of the frame: the four result counters go straight into the slot's pinned host words (instead of a
    // separate 32-byte device-to-host copy, ~4 us of stream time and one more API call per frame)
    if (tile == 0 && threadIdx.x < 5u && host_counters) host_counters[threadIdx.x] = counters[threadIdx.x];
    const uint32_t i0 = item_base[tile], n_seg = item_base[tile + 1] - i0;
    if (n_seg == 1u) return;                      // the tile's only work item wrote the pixels itself; n_seg == 0: no pairs, background only
    int tx, tyl;
    tile_xy(f, (uint32_t)tile, tx, tyl);
    const int sc = f.shard_count <= 1 ? 1 : f.shard_count;
    const int ty = tyl * sc + (sc > 1 ? f.shard_index : 0);
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t grp = lane >> 4, gi = lane & 15u;      // same lane -> pixel map as k_composite
    const int lxi = (int)grp * 4 + (int)(gi & 3u);
    const int lyi = (int)wave * 4 + (int)(gi >> 2);
    const int px = (tx + f.col0) * kTile + lxi, py = ty * kTile + lyi;
    if (px >= f.width || py >= f.height) return;
    float T = 1.0f, ar = 0.0f, ag = 0.0f, ab = 0.0f;
    // eight segments' partials in flight together (clamped, unmasked): a horizon tile of c5 has 40 segments, and one dependent
    // 4-KB load per segment made the longest tile the kernel's duration (59 us at c5)
    for (uint32_t s0 = 0; s0 < n_seg; s0 += 8u) {
        float4 p[8];
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) p[k] = partials[(size_t)(i0 + min(s0 + k, n_seg - 1u)) * 256u + tid];
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) {
            if (s0 + k < n_seg) {                 // workgroup-uniform
                ar = fmaf(T, p[k].x, ar);
                ag = fmaf(T, p[k].y, ag);
                ab = fmaf(T, p[k].z, ab);
                T = T * p[k].w;
            }
        }
    }
    float4 bg = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bg_rgba) bg = bg_rgba[(size_t)py * f.width + px];
    float4 o;
    o.x = fmaf(T, bg.x, ar);
    o.y = fmaf(T, bg.y, ag);
    o.z = fmaf(T, bg.z, ab);
    o.w = fmaf(T, bg.w, 1.0f - T);
    const int orow = tyl * kTile + lyi;
    if (orow < out_rows) out[(size_t)orow * f.out_w + (px - f.out_x0)] = o;
}

// all-gathered shards -> frame.  rows: shard = tile row % count (rows_padded rows each, full width);
// columns: shard = tile column / band_tiles (height rows each, band_px wide)
__global__ void k_unshard(const float4* __restrict__ gathered, float4* __restrict__ out, int width, int height,
                          int shard_count, int rows_padded, int band_px)
{
    int x = blockIdx.x * blockDim.x + threadIdx.x;
    int y = blockIdx.y;
    if (x >= width || y >= height) return;
    size_t src;
    if (band_px > 0) {
        const int shard = x / band_px;
        src = ((size_t)shard * height + y) * band_px + (x - shard * band_px);
    } else {
        int ty = y >> 4;
        int shard = ty % shard_count, tyl = ty / shard_count;
        src = ((size_t)shard * rows_padded + (size_t)tyl * kTile + (y & 15)) * width + x;
    }
    out[(size_t)y * width + x] = gathered[src];
}

// ---- launch wrappers (called from gswt_api.hip) -------------------------------------
void launch_chunk_tabs(hipStream_t s, const DrawDev* draws, const uint32_t* xcd_first, uint32_t n_draws, uint2* chunk_tab, uint2* chunk_tab_xcd,
                       const uint64_t per_xcd[8], uint64_t longest)
{
    XcdLens xl;
    for (int x = 0; x < 8; x++) xl.len[x] = (uint32_t)per_xcd[x];
    xl.longest = (uint32_t)longest;
    if (n_draws) hipLaunchKernelGGL(k_chunk_tabs, dim3(n_draws + 8u), dim3(256), 0, s, draws, xcd_first, n_draws, chunk_tab, chunk_tab_xcd, xl);
}

GraphRec*& graph_recorder()
{
    static thread_local GraphRec* rec = nullptr;
    return rec;
}

bool kernel_events_enabled()
{
    static const bool on = !(getenv("GSWT_KERNEL_EVENTS") && atoi(getenv("GSWT_KERNEL_EVENTS")) == 0);
    return on;
}

void launch_cull(hipStream_t s, const Frame& f, const DrawDev* draws, uint32_t n_draws, uint32_t* draw_culled, uint32_t* cell_culled, uint32_t n_cells,
                 uint32_t* zero_a, uint32_t n_zero_a, uint32_t* zero_b, uint32_t n_zero_b, uint32_t* zero_c, uint32_t n_zero_c,
                 uint32_t* live_cnt, uint4* live_tab, uint32_t* zero_d, uint32_t n_zero_d,
                 const uint2* chunk_tab, uint32_t n_chunks, const float* boxes, bool chunk_cull, uint32_t* live_cid)
{
    (void)n_draws;
    uint32_t grid = (n_chunks + 255u) / 256u;              // one thread per chunk; the clears and the cell table stride over the grid
    if (grid < 32) grid = 32;
    GSWT_LAUNCH(k_cull, dim3(grid), dim3(256), s, f, draws, draw_culled, cell_culled, n_cells, zero_a, n_zero_a, zero_b, n_zero_b, zero_c, n_zero_c, zero_d, n_zero_d,
                chunk_tab, n_chunks, boxes, chunk_cull && boxes ? 1u : 0u, live_cnt, live_tab, live_cid);
}

void launch_project(hipStream_t s, bool debug, const Frame& f, const DrawDev* draws, const uint2* chunk_tab, uint32_t n_launch, uint32_t n_chunks,
                    const uint32_t* static_list, const uint32_t* merged_list, const uint32_t* merged_map, const uint4* tex,
                    const float* hmap, const uint32_t* draw_culled, const uint32_t* cell_culled, uint32_t* live_cnt, const uint4* live_tab, uint2* rects, Rec* recs, float* depths, uint32_t* block_sums,
                    uint32_t* super_sums, unsigned long long* counters, Varyings* dbg, float4* col_f, uint32_t pair_cap, bool strict)
{
    if (n_chunks == 0) return;
    const uint32_t n_super = n_chunks / 256u + 1u;      // super_sums = [pairs x 16 n_super][visible x 16 n_super][exclusive pair prefix x n_super] (kSuperStride), zeroed by the caller
    const bool full = f.surface_type == 2u || f.draw_mode != 0u;
    // GSWT_PROJECT_HALVES=2 (measurement only): 512-thread workgroups, two launch-list entries each.  Built in round 4 against the
    // "dispatch-bound" reading of the kernel's trace and LOST at every size, same bits: c3 83.4 us against 76.2, c3h 101.5 / 97.2,
    // c5 530 / 464 (profiles/r04_project_halves.txt) -- a workgroup now lives as long as the slower of its two chunks and holds 8 wave
    // slots until then; halving what the dispatcher hands out does not pay for that.
    static const bool wide = [] { const char* e = getenv("GSWT_PROJECT_HALVES"); return e && atoi(e) == 2; }();
#define GSWT_LAUNCH_PROJECT_S(D, F, S)                                                                                         \
    if (!D && wide) GSWT_LAUNCH((k_project<false, F, S, 2>), dim3(((n_launch / 8u + 1u) / 2u) * 8u), dim3(512), s, f, draws, chunk_tab, static_list, merged_list, \
                       merged_map, tex, hmap, draw_culled, cell_culled, live_cnt, live_tab, rects, recs, depths, block_sums, super_sums, n_super, dbg, col_f); \
    else GSWT_LAUNCH((k_project<D, F, S>), dim3(n_launch), dim3(256), s, f, draws, chunk_tab, static_list, merged_list,        \
                       merged_map, tex, hmap, draw_culled, cell_culled, live_cnt, live_tab, rects, recs, depths, block_sums, super_sums, n_super, dbg, col_f)
#define GSWT_LAUNCH_PROJECT(D, F) do { if (strict) GSWT_LAUNCH_PROJECT_S(D, F, true); else GSWT_LAUNCH_PROJECT_S(D, F, false); } while (0)
    if (debug && full) { GSWT_LAUNCH_PROJECT(true, true); }
    else if (debug) { GSWT_LAUNCH_PROJECT(true, false); }
    else if (full) { GSWT_LAUNCH_PROJECT(false, true); }
    else { GSWT_LAUNCH_PROJECT(false, false); }
#undef GSWT_LAUNCH_PROJECT
#undef GSWT_LAUNCH_PROJECT_S
    GSWT_LAUNCH(k_totals, dim3(1), dim3(256), s, super_sums, n_super, counters, super_sums + 2u * kSuperStride * n_super, pair_cap, live_cnt, n_launch / 8u);
}

// keys: tile ids, vals: slots.  GSWT_ORDER_DEPTH (dkeys != nullptr): also each pair's depth bits -> dkeys and the frame's key range -> krange.
void launch_emit(hipStream_t s, const Frame& f, uint32_t n_chunks, const uint2* rects, const uint32_t* block_sums,
                 const uint32_t* super_sums, uint32_t pair_cap, unsigned long long* counters, uint32_t* keys, uint32_t* vals,
                 const float* depths, uint32_t* dkeys, uint32_t* krange, const uint32_t* live_cnt, const uint32_t* live_cid, uint32_t n_launch)
{
    if (n_chunks == 0) return;
    const uint32_t n_super = n_chunks / 256u + 1u;      // [pairs x n_super][visible x n_super][exclusive pair prefix x n_super]
    const uint32_t* const excl = super_sums + 2u * kSuperStride * n_super;
    const float* const no_f = nullptr; uint32_t* const no_u = nullptr;
    // over k_cull's table of live chunks (GSWT_EMIT_TAB=0: over every chunk of the frame, four consecutive ones per workgroup)
    static const bool tab = !(getenv("GSWT_EMIT_TAB") && atoi(getenv("GSWT_EMIT_TAB")) == 0);
    if (tab && live_cnt && live_cid && n_launch) {
        const uint32_t* const cnt2 = live_cnt + 8u * kSuperStride;               // k_totals' copy of the live counts
        const dim3 grid(((n_launch / 8u + kEmitGroup - 1u) / kEmitGroup) * 8u);
        if (dkeys) GSWT_LAUNCH((k_emit<true, true>), grid, dim3(256), s, f, rects, block_sums, excl, n_chunks, pair_cap, counters, keys, vals, depths, dkeys, krange, cnt2, live_cid);
        else GSWT_LAUNCH((k_emit<false, true>), grid, dim3(256), s, f, rects, block_sums, excl, n_chunks, pair_cap, counters, keys, vals, no_f, no_u, no_u, cnt2, live_cid);
    } else {
        const dim3 grid((n_chunks + kEmitGroup - 1u) / kEmitGroup);
        const uint32_t* const no_c = nullptr;
        if (dkeys) GSWT_LAUNCH((k_emit<true, false>), grid, dim3(256), s, f, rects, block_sums, excl, n_chunks, pair_cap, counters, keys, vals, depths, dkeys, krange, no_c, no_c);
        else GSWT_LAUNCH((k_emit<false, false>), grid, dim3(256), s, f, rects, block_sums, excl, n_chunks, pair_cap, counters, keys, vals, no_f, no_u, no_u, no_c, no_c);
    }
}

// Sorts (keys, vals) by key bits [0, key_bits); the pair count is read on the device (*n_ptr), grids are
// sized for `n_cap`.  Result ends in (keys_a, vals_a) or (keys_b, vals_b): returns 0 if in a, 1 if in b.
// ws: per pass [ghist 256 x nblk][gsup 256 x nsup][gtot 256]; the gsup/gtot parts must be zero on entry
// (radix_ws_words() u32 in total, zeroed by k_cull each frame).
size_t radix_ws_words(uint32_t n_cap, int key_bits)
{
    const uint32_t nblk = (n_cap + kSortBlock - 1) / kSortBlock, nsup = (nblk >> kSupShift) + 1;
    const int passes = (key_bits + 7) / 8;
    return (size_t)passes * ((size_t)256 * nblk + (size_t)256 * nsup + 256);
}
// The part of the workspace that has to be zero when a sort starts: the group rows and digit totals of every pass (targets of
// atomic adds), which come first.  The per-workgroup rows behind them are written in full by k_radix_hist (every launched
// workgroup stores its 256 counts, zeros past the item count) -- a frame used to clear them too: 2 MB at c3, 13.6 MB at c5, by
// k_cull's 32 workgroups.
size_t radix_ws_zero_words(uint32_t n_cap, int key_bits)
{
    const uint32_t nblk = (n_cap + kSortBlock - 1) / kSortBlock, nsup = (nblk >> kSupShift) + 1;
    const int passes = (key_bits + 7) / 8;
    return (size_t)passes * ((size_t)256 * nsup + 256);
}

int launch_sort(hipStream_t s, uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n_cap,
                const unsigned long long* n_ptr, int key_bits, uint32_t* ws, uint2* ranges = nullptr, const uint32_t* krange = nullptr,
                uint32_t* aux_a = nullptr, uint32_t* aux_b = nullptr)
{
    if (n_cap == 0) return 0;
    const uint32_t nblk = (n_cap + kSortBlock - 1) / kSortBlock, nsup = (nblk >> kSupShift) + 1;
    // (GSWT_SORT_WIDE_MAX_M: tuning override of kSortWideMax in units of 2^20 items, read once)
    static const uint32_t wide_max = [] { const char* e = getenv("GSWT_SORT_WIDE_MAX_M"); return e ? (uint32_t)atoi(e) << 20 : kSortWideMax; }();
    const int threads = n_cap <= wide_max ? 512 : 256;
    int cur = 0;
    const int passes = (key_bits + 7) / 8;
    uint32_t* hist_rows = ws + (size_t)passes * ((size_t)256 * nsup + 256);      // behind the zeroed part (radix_ws_zero_words)
    for (int shift = 0; shift < key_bits; shift += 8) {
        uint32_t nbits = (uint32_t)((key_bits - shift) < 8 ? (key_bits - shift) : 8);
        uint32_t mask = (1u << nbits) - 1u;
        uint32_t* gsup = ws; uint32_t* gtot = gsup + (size_t)256 * nsup;
        ws = gtot + 256;
        uint32_t* ghist = hist_rows;
        hist_rows += (size_t)256 * nblk;
        uint32_t* ki = cur ? keys_b : keys_a; uint32_t* vi = cur ? vals_b : vals_a;
        uint32_t* ko = cur ? keys_a : keys_b; uint32_t* vo = cur ? vals_a : vals_b;
        const uint32_t* xi = cur ? aux_b : aux_a; uint32_t* xo = cur ? aux_a : aux_b;
#define GSWT_SORT_PASS(T)                                                                                                        \
        GSWT_LAUNCH(k_radix_hist<T>, dim3(nblk), dim3(T), s, ki, n_ptr, n_cap, (uint32_t)shift, mask, nbits, ghist, gsup, gtot, nblk, nsup, krange); \
        if (nsup > kSupDirect) GSWT_LAUNCH(k_radix_supscan, dim3(64), dim3(256), s, gsup, gtot, nsup);                          \
        if (aux_a && n_cap > (12u << 20)) GSWT_LAUNCH((k_radix_scatter<T, 2>), dim3(nblk), dim3(T), s, ki, vi, ko, vo, n_ptr, n_cap, (uint32_t)shift, mask, nbits, \
                           ghist, gsup, gtot, nblk, nsup, shift + 8 >= key_bits ? ranges : (uint2*)nullptr, krange, xi, xo);                  \
        else if (aux_a) GSWT_LAUNCH((k_radix_scatter<T, 1>), dim3(nblk), dim3(T), s, ki, vi, ko, vo, n_ptr, n_cap, (uint32_t)shift, mask, nbits,    \
                           ghist, gsup, gtot, nblk, nsup, shift + 8 >= key_bits ? ranges : (uint2*)nullptr, krange, xi, xo);                  \
        else GSWT_LAUNCH((k_radix_scatter<T, 0>), dim3(nblk), dim3(T), s, ki, vi, ko, vo, n_ptr, n_cap, (uint32_t)shift, mask, nbits,    \
                           ghist, gsup, gtot, nblk, nsup, shift + 8 >= key_bits ? ranges : (uint2*)nullptr, krange, xi, xo)
        if (threads == 512) { GSWT_SORT_PASS(512); }
        else { GSWT_SORT_PASS(256); }
#undef GSWT_SORT_PASS
        cur ^= 1;
    }
    return cur;
}

// Builds every merged group's (gs_index | lod, map_id) list on the device.  keys/vals a,b: n_total u32 each;
// ws: radix_ws_words(n_total, 16 + group_bits) zeroed words; n_total_dev: device u64 = n_total (+ two zero words after it).
void launch_merge_build(hipStream_t s, const MergeSeg* segs, uint32_t n_segs, const uint2* blocks, uint32_t n_blocks, MergeGroup* groups, uint32_t n_groups,
                        const int32_t* raw, uint32_t n_total, const unsigned long long* n_total_dev, uint32_t* ka, uint32_t* va,
                        uint32_t* kb, uint32_t* vb, uint32_t* radix_ws, int group_bits, uint32_t* merged_list, uint32_t* merged_map)
{
    if (n_total == 0 || n_groups == 0 || n_blocks == 0) return;
    hipLaunchKernelGGL(k_mg_minmax, dim3(n_blocks), dim3(256), 0, s, segs, blocks, raw, groups);
    hipLaunchKernelGGL(k_mg_keys, dim3(n_blocks), dim3(256), 0, s, segs, blocks, raw, groups, ka, va);
    const int where = launch_sort(s, ka, va, kb, vb, n_total, n_total_dev, 16 + group_bits, radix_ws);
    hipLaunchKernelGGL(k_mg_final, dim3((n_total + 255) / 256), dim3(256), 0, s, segs, n_segs, groups, where ? kb : ka, where ? vb : va,
                       n_total, merged_list, merged_map);
}

void launch_merge_copy(hipStream_t s, const MergeCopy* jobs, const uint2* blocks, uint32_t n_blocks, const uint2* remap, const MergeSources& src,
                       uint32_t* new_list, uint32_t* new_map)
{
    if (n_blocks) hipLaunchKernelGGL(k_mg_copy, dim3(n_blocks), dim3(256), 0, s, jobs, blocks, remap, src, new_list, new_map);
}

// `ranges` must be zero on entry (k_cull clears it each frame)

// GSWT_ORDER_DEPTH, tile-local path: depth-sorts every screen tile's slice of the (tile-sorted) pair list in LDS; dkeys = the pairs' depth
// bits in the same order (the tile sort's payload)
void launch_tile_depth_sort(hipStream_t s, const uint2* ranges, uint32_t* vals, uint32_t* dkeys, uint32_t* vals_scratch, uint32_t* dkeys_scratch, int n_tiles,
                            uint32_t* long_list, unsigned long long* counters)
{
    if (n_tiles <= 0) return;
    // four size classes in three launches: a wave per tile, a 256-thread workgroup per tile, a fixed grid over the list of still longer ones
    // which then also walks the list of those that do not fit LDS at all (long_list: [0] count, [1 .. n_tiles] tiles; the second list behind it)
    const uint32_t nt = (uint32_t)n_tiles;
    GSWT_LAUNCH((k_tile_depth_sort<64, 8, false>), dim3(nt), dim3(64), s, ranges, vals, dkeys, nt, long_list, counters, 0u, 0u, vals_scratch, dkeys_scratch);
    GSWT_LAUNCH((k_tile_depth_sort<256, 16, false>), dim3(nt), dim3(256), s, ranges, vals, dkeys, nt, long_list, counters, 512u, 1u, vals_scratch, dkeys_scratch);
    GSWT_LAUNCH((k_tile_depth_sort<1024, 16, true>), dim3(kTileSortLongGrid), dim3(1024), s, ranges, vals, dkeys, nt, long_list, counters, 4096u, 0u,
                vals_scratch, dkeys_scratch);
}
uint32_t tile_depth_sort_cap() { return kTileSortCap; }

// ranges -> per-tile segment counts -> item_base (exclusive scan, item_base[n_tiles] = #items) ->
// k_composite over an upper bound of items -> k_combine.
void launch_composite(hipStream_t s, const Frame& f, const uint2* ranges, const uint32_t* vals, const Rec* recs, const float* depths,
                      const float4* col_f, const float4* bg_rgba, const float* bg_depth, float4* out, int n_tiles, int out_rows,
                      uint32_t seg, uint32_t n_pairs, uint32_t* item_base, uint4* item_tab, float4* partials,
                      hipEvent_t ev_begin, hipEvent_t ev_end, unsigned long long* counters, unsigned long long* host_counters, int variant,
                      const uint32_t* krange, uint32_t depth_passes, uint32_t* tile_tick, bool report_max, bool heavy_first)
{
    if (n_tiles == 0) {                 // a shard without screen tiles (more ranks than tile columns): the events still exist
        if (ev_begin) hipEventRecord(ev_begin, s);
        if (ev_end) hipEventRecord(ev_end, s);
        return;
    }
    const uint32_t max_items = (uint32_t)n_tiles + n_pairs / seg + 1u;
    // variant 2 (GSWT_OPT_FOLD_COMBINE): k_composite folds the segment partials itself and writes the empty tiles: no k_combine launch
    const bool fold = variant == 2 && tile_tick != nullptr && host_counters != nullptr;
    if (heavy_first) {
        GSWT_LAUNCH(k_items<true>, dim3((n_tiles + 8191) / 8192), dim3(1024), s, ranges, n_tiles, seg, item_base, item_tab, max_items, krange, depth_passes, counters,
                fold ? 1u : 0u, report_max ? 1u : 0u);
    } else {
        GSWT_LAUNCH(k_items<false>, dim3((n_tiles + 8191) / 8192), dim3(1024), s, ranges, n_tiles, seg, item_base, item_tab, max_items, krange, depth_passes, counters,
                fold ? 1u : 0u, report_max ? 1u : 0u);
    }
    const bool early = f.t_eps > 0.0f, depth = f.has_depth != 0, colf = f.draw_mode != 0u;
    // (the shipped compositors carry ev_begin / ev_end themselves: GSWT_LAUNCH_TIMED; the experiment variants record them around the launch)
#ifdef GSWT_EXPERIMENTS
    if (ev_begin && (f.dbg_flags & (0x20000 | 0x4000 | 0x1000))) hipEventRecord(ev_begin, s);
    if (f.dbg_flags & 0x20000) {         // experiment: two packed waves per item with shared staging (0x2000: 256-pair batches instead of 128)
#define GSWT_LAUNCH_COMPOSITE_P2(E, D, C, NB2, OCC)                                                                             \
        GSWT_LAUNCH((k_composite_p2<E, D, C, NB2, OCC>), dim3(max_items), dim3(128), s, f, item_base, item_tab, vals, recs, depths, col_f, \
                    bg_rgba, bg_depth, out, partials, n_tiles, out_rows)
        // 0x40000: the register allocator is told to fit 8 waves per SIMD (64 VGPRs) instead of taking what it likes
#define GSWT_LAUNCH_COMPOSITE_P2N(E, D, C) { if (f.dbg_flags & 0x2000) GSWT_LAUNCH_COMPOSITE_P2(E, D, C, 2, 4); else if ((f.dbg_flags & 0x40000) && !D && !C) GSWT_LAUNCH_COMPOSITE_P2(E, false, false, 1, 8); else GSWT_LAUNCH_COMPOSITE_P2(E, D, C, 1, 4); }
        if (colf) {
            if (depth) GSWT_LAUNCH_COMPOSITE_P2N(false, true, true)
            else GSWT_LAUNCH_COMPOSITE_P2N(false, false, true)
        }
        else if (early && depth) GSWT_LAUNCH_COMPOSITE_P2N(true, true, false)
        else if (early) GSWT_LAUNCH_COMPOSITE_P2N(true, false, false)
        else if (depth) GSWT_LAUNCH_COMPOSITE_P2N(false, true, false)
        else GSWT_LAUNCH_COMPOSITE_P2N(false, false, false)
#undef GSWT_LAUNCH_COMPOSITE_P2N
#undef GSWT_LAUNCH_COMPOSITE_P2
        if (ev_end) hipEventRecord(ev_end, s);
        GSWT_LAUNCH(k_combine, dim3(n_tiles), dim3(256), s, f, item_base, partials, bg_rgba, out, n_tiles, out_rows, (const unsigned long long*)counters, host_counters);
        return;
    }
    if (f.dbg_flags & 0x4000) {          // experiment: independent strip waves (0x8000: two strips per wave, 0x10000: four; 0x2000: 128-pair batches)
        const uint32_t items8 = (max_items + 7u) & ~7u;
#define GSWT_LAUNCH_COMPOSITE_S(E, D, C, NB, SPW)                                                                                  \
        GSWT_LAUNCH((k_composite_s<E, D, C, NB, SPW>), dim3(items8 * (4u / SPW)), dim3(64), s, f, item_base, item_tab, vals, recs, depths, col_f, \
                    bg_rgba, bg_depth, out, partials, n_tiles, out_rows)
#define GSWT_LAUNCH_COMPOSITE_SN(E, D, C)                                                                                          \
        {                                                                                                                          \
            const int spw = (f.dbg_flags & 0x10000) ? 4 : (f.dbg_flags & 0x8000) ? 2 : 1;                                          \
            if (f.dbg_flags & 0x2000) { if (spw == 4) GSWT_LAUNCH_COMPOSITE_S(E, D, C, 2, 4); else if (spw == 2) GSWT_LAUNCH_COMPOSITE_S(E, D, C, 2, 2); else GSWT_LAUNCH_COMPOSITE_S(E, D, C, 2, 1); } \
            else { if (spw == 4) GSWT_LAUNCH_COMPOSITE_S(E, D, C, 1, 4); else if (spw == 2) GSWT_LAUNCH_COMPOSITE_S(E, D, C, 1, 2); else GSWT_LAUNCH_COMPOSITE_S(E, D, C, 1, 1); } \
        }
        if (early && !depth && !colf) GSWT_LAUNCH_COMPOSITE_SN(true, false, false)
        else if (!early && !depth && !colf) GSWT_LAUNCH_COMPOSITE_SN(false, false, false)
        else if (depth && !colf) { if (early) GSWT_LAUNCH_COMPOSITE_S(true, true, false, 1, 1); else GSWT_LAUNCH_COMPOSITE_S(false, true, false, 1, 1); }
        else { if (depth) GSWT_LAUNCH_COMPOSITE_S(false, true, true, 1, 1); else GSWT_LAUNCH_COMPOSITE_S(false, false, true, 1, 1); }
#undef GSWT_LAUNCH_COMPOSITE_SN
#undef GSWT_LAUNCH_COMPOSITE_S
        if (ev_end) hipEventRecord(ev_end, s);
        GSWT_LAUNCH(k_combine, dim3(n_tiles), dim3(256), s, f, item_base, partials, bg_rgba, out, n_tiles, out_rows, (const unsigned long long*)counters, host_counters);
        return;
    }
    if (f.dbg_flags & 0x1000) {          // experiment: the one-wave-per-item packed compositor (0x2000: 64-pair batches instead of 128)
#define GSWT_LAUNCH_COMPOSITE_P(E, D, C, NB)                                                                                    \
        GSWT_LAUNCH((k_composite_p<E, D, C, NB>), dim3(max_items), dim3(64), s, f, item_base, item_tab, vals, recs, depths, col_f, \
                    bg_rgba, bg_depth, out, partials, n_tiles, out_rows)
#define GSWT_LAUNCH_COMPOSITE_PN(E, D, C) { if (f.dbg_flags & 0x2000) GSWT_LAUNCH_COMPOSITE_P(E, D, C, 1); else GSWT_LAUNCH_COMPOSITE_P(E, D, C, 2); }
        if (colf) {
            if (depth) GSWT_LAUNCH_COMPOSITE_PN(false, true, true)
            else GSWT_LAUNCH_COMPOSITE_PN(false, false, true)
        }
        else if (early && depth) GSWT_LAUNCH_COMPOSITE_PN(true, true, false)
        else if (early) GSWT_LAUNCH_COMPOSITE_PN(true, false, false)
        else if (depth) GSWT_LAUNCH_COMPOSITE_PN(false, true, false)
        else GSWT_LAUNCH_COMPOSITE_PN(false, false, false)
#undef GSWT_LAUNCH_COMPOSITE_PN
#undef GSWT_LAUNCH_COMPOSITE_P
        if (ev_end) hipEventRecord(ev_end, s);
        GSWT_LAUNCH(k_combine, dim3(n_tiles), dim3(256), s, f, item_base, partials, bg_rgba, out, n_tiles, out_rows, (const unsigned long long*)counters, host_counters);
        return;
    }
#endif
    if (variant == 1) {                  // GSWT_OPT_COMPOSITE = 1: decoupled waves (k_composite_dw), same image bit for bit
#define GSWT_LAUNCH_COMPOSITE_DW(E, D, C)                                                                                      \
        GSWT_LAUNCH_TIMED((k_composite_dw<E, D, C>), dim3(max_items), dim3(256), s, ev_begin, ev_end, f, item_base, item_tab, vals, recs, \
                           depths, col_f, bg_rgba, bg_depth, out, partials, n_tiles, out_rows)
        if (colf) { if (depth) GSWT_LAUNCH_COMPOSITE_DW(false, true, true); else GSWT_LAUNCH_COMPOSITE_DW(false, false, true); }
        else if (early && depth) GSWT_LAUNCH_COMPOSITE_DW(true, true, false);
        else if (early) GSWT_LAUNCH_COMPOSITE_DW(true, false, false);
        else if (depth) GSWT_LAUNCH_COMPOSITE_DW(false, true, false);
        else GSWT_LAUNCH_COMPOSITE_DW(false, false, false);
#undef GSWT_LAUNCH_COMPOSITE_DW
        GSWT_LAUNCH(k_combine, dim3(n_tiles), dim3(256), s, f, item_base, partials, bg_rgba, out, n_tiles, out_rows, (const unsigned long long*)counters, host_counters);
        return;
    }
    if (fold) {
#define GSWT_LAUNCH_COMPOSITE_F(E, D, C)                                                                                       \
        GSWT_LAUNCH_TIMED((k_composite<E, D, C, false, false, true>), dim3(max_items), dim3(256), s, ev_begin, ev_end, f, ranges, item_base, item_tab, seg, vals, recs, \
                           depths, col_f, bg_rgba, bg_depth, out, partials, n_tiles, out_rows, tile_tick, (const unsigned long long*)counters, host_counters)
        if (colf) { if (depth) GSWT_LAUNCH_COMPOSITE_F(false, true, true); else GSWT_LAUNCH_COMPOSITE_F(false, false, true); }
        else if (early && depth) GSWT_LAUNCH_COMPOSITE_F(true, true, false);
        else if (early) GSWT_LAUNCH_COMPOSITE_F(true, false, false);
        else if (depth) GSWT_LAUNCH_COMPOSITE_F(false, true, false);
        else GSWT_LAUNCH_COMPOSITE_F(false, false, false);
#undef GSWT_LAUNCH_COMPOSITE_F
        return;
    }
#define GSWT_LAUNCH_COMPOSITE_K(E, D, C, PK, DW)                                                                               \
    GSWT_LAUNCH_TIMED((k_composite<E, D, C, PK, DW>), dim3(max_items), dim3(256), s, ev_begin, ev_end, f, ranges, item_base, item_tab, seg, vals, recs, \
                       depths, col_f, bg_rgba, bg_depth, out, partials, n_tiles, out_rows, (uint32_t*)nullptr, (const unsigned long long*)nullptr, (unsigned long long*)nullptr)
#ifdef GSWT_EXPERIMENTS      // measured and slower (profiles/r03_composite_variants.txt): 0x80000 the packed-coordinate step (15 VALU instead of 19), 0x100000 the register-broadcast (DPP) walk
#define GSWT_LAUNCH_COMPOSITE(E, D, C) do { if ((f.dbg_flags & 0x180000) == 0x180000) GSWT_LAUNCH_COMPOSITE_K(E, D, C, true, true); /* 0x180000: the scalar step with the blend predicated by v_cndmask behind a ballot test (round 2's form) instead of EXEC masking */ \
        else if (f.dbg_flags & 0x80000) GSWT_LAUNCH_COMPOSITE_K(E, D, C, true, false); else if (f.dbg_flags & 0x100000) GSWT_LAUNCH_COMPOSITE_K(E, D, C, false, true); else GSWT_LAUNCH_COMPOSITE_K(E, D, C, false, false); } while (0)
#else
#define GSWT_LAUNCH_COMPOSITE(E, D, C) GSWT_LAUNCH_COMPOSITE_K(E, D, C, false, false)
#endif
        if (colf) {                       // debug draw modes: float colours from the side buffer
        if (depth) GSWT_LAUNCH_COMPOSITE(false, true, true);
        else GSWT_LAUNCH_COMPOSITE(false, false, true);
    }
    else if (early && depth) GSWT_LAUNCH_COMPOSITE(true, true, false);
    else if (early) GSWT_LAUNCH_COMPOSITE(true, false, false);
    else if (depth) GSWT_LAUNCH_COMPOSITE(false, true, false);
    else GSWT_LAUNCH_COMPOSITE(false, false, false);
#undef GSWT_LAUNCH_COMPOSITE
#undef GSWT_LAUNCH_COMPOSITE_K
    GSWT_LAUNCH(k_combine, dim3(n_tiles), dim3(256), s, f, item_base, partials, bg_rgba, out, n_tiles, out_rows, (const unsigned long long*)counters, host_counters);
}

// k_totals alone on caller-provided sums (unit test of the 64-bit pair count)
void launch_totals(hipStream_t s, uint32_t* super_sums, uint32_t n_super, unsigned long long* counters, uint32_t pair_cap)
{
    hipLaunchKernelGGL(k_totals, dim3(1), dim3(256), 0, s, super_sums, n_super, counters, super_sums + 2u * kSuperStride * n_super, pair_cap, (uint32_t*)nullptr, 0xFFFFFFFFu);
}

void launch_unshard(hipStream_t s, const float4* gathered, float4* out, int width, int height, int shard_count, int rows_padded, int band_px)
{
    hipLaunchKernelGGL(k_unshard, dim3((width + 255) / 256, height), dim3(256), 0, s, gathered, out, width, height, shard_count, rows_padded, band_px);
}

}  // namespace gswt
#ifdef GSWT_TRACE
extern "C" __attribute__((visibility("default"))) int gswt_debug_trace(unsigned long long* out, unsigned n_items)
{
    hipDeviceSynchronize();
    if (n_items > gswt::kTraceItems) n_items = gswt::kTraceItems;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gswt::g_trace), (size_t)n_items * 64);
}
#endif
#ifdef GSWT_STATS
extern "C" __attribute__((visibility("default"))) int gswt_debug_stats(unsigned long long* out)
{
    hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gswt::g_stats), 64);
}
#endif
