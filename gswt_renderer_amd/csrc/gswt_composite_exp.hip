// gswt_composite_exp.hip -- compositor variants that were BUILT AND MEASURED in round 3 and lost against k_composite
// (gswt_kernels.hip).  Not part of the product library: included by gswt_kernels.hip only under -DGSWT_EXPERIMENTS
// (`make variants` -> build_var/libgswt_hip_exp.so), selected at run time through GSWT_OPT_DEBUG_FLAGS bits
// (tools/composite_w_probe.py, tools/composite_trace.py).  All of them produce k_composite's image bit for bit (same F3 / F4,
// same blend order per pixel, same partials layout).  Measurements: profiles/r03_composite_variants.txt, DESIGN.md section 6.
//   k_composite_p  (0x1000)  one wave per item, two pixels per lane (v_pk_fma_f32), the tile as two 16x8 halves in turn
//   k_composite_p2 (0x20000) two such waves per item with k_composite's shared staging and barriers
//   k_composite_s  (0x4000)  independent strip waves: 4 / SPW single-wave workgroups per item, each staging the item's pairs itself
//   (k_composite_w, one wave per item with one pixel per lane and four strips in turn, and its v_fma_mix_f32 colour variant,
//    were the first of the series: 111-114 us; git history, commit 55b2fcd)
// ------------------------------------------------------------------------------------
// k_composite_p (round 3): ONE WAVE per work item, TWO pixels per lane, packed FP32.
// The wave owns the whole 16x16 tile and composites it as two 16x8 halves, one after the other.  Inside a half, the
// eight 8-lane groups own its eight 4x4 sub-blocks (group g: columns 4 (g & 3) .., rows 4 (g >> 2) ..) and walk EIGHT
// different splats concurrently; lane l of a group holds the two x-neighbours (2 (l & 1), 2 (l & 1) + 1) of row l >> 1.
// F4 for the pair of pixels is v_pk_fma_f32 / v_pk_mul_f32 (IEEE per component: the canonical sequence, bit for bit the
// image of k_composite); pu_y / pv_y and the colour conversions are shared by the two pixels.
// Why (measured on the c3 frame, tools/composite_trace.py + PMC): a wave issues one VALU instruction every ~7-8 cycles
// whatever its instruction-level parallelism, and the LDS array is ~70 % busy while the walks run (one list entry and
// two ds_read_b128 per step and wave).  Two pixels per lane need 21 instructions (28 issue slots) and the same 10 LDS
// cycles per step for 128 pixel evaluations, against 18 (19.5) and 10 for 64; the eight lists of a half tile are walked in
// 1.22 M wave-steps on that frame against 1.98 M for four lists of a 16x4 strip (tools sim of the real boxes).
// One wave per item: nothing waits at a workgroup barrier for the slowest of four strips (41 % of the wave-slots of the
// four-wave kernel), and a CU holds ~24 independent gather -> stage -> walk chains instead of 8.
// Measured and dropped on the way (profiles/r03_composite_variants.txt): the same one-wave item with one pixel per lane
// and four strips walked in turn (k_composite_w: 111-114 us against 96 for k_composite -- the same walk, four times the
// serial chain per item), with v_fma_mix_f32 colours from halves (three VALU less, one ds_read_b32 more per step: +-0).
// ------------------------------------------------------------------------------------

// bin + walk of one staged batch for one 16x8 half tile (rows r0 .. r0 + 7).
template <bool EARLY, bool DEPTH, bool COLF, int NB>
__device__ __forceinline__ void packed_bin_walk(const Frame& f, const int r0, const v2f lx2, const float ly, const uint32_t lane, const uint32_t grp,
                                                const uint32_t n, const char* const q0b, const char* const q1b, const uint32_t* const s_bb,
                                                const char* const dpb, const char* const q2b, uint16_t* const wlist, const uint32_t list_stride,
                                                const v2f dbuf2, const float t_eps, v2f& T2, v2f& ar2, v2f& ag2, v2f& ab2, bool& live)
{
    constexpr uint32_t PB = (uint32_t)NB * 64u;
    uint16_t* const my_list = wlist + grp * list_stride;
    uint32_t cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int c = 0; c < NB; c++) {
        const uint32_t idx = (uint32_t)c * 64u + lane;
        bool hx[4] = {false, false, false, false}, hy0 = false, hy1 = false;
        if (idx < n) {
            const uint32_t bb = s_bb[idx];
            const int xa = __builtin_amdgcn_sbfe((int)bb, 0, 8), xb2 = __builtin_amdgcn_sbfe((int)bb, 8, 8);
            const int ya = __builtin_amdgcn_sbfe((int)bb, 16, 8), yb = __builtin_amdgcn_sbfe((int)bb, 24, 8);
            hy0 = yb >= r0 && ya <= r0 + 3;
            hy1 = yb >= r0 + 4 && ya <= r0 + 7;
            hx[0] = xa <= 3 && xb2 >= 0;
            hx[1] = xb2 >= 4 && xa <= 7;
            hx[2] = xb2 >= 8 && xa <= 11;
            hx[3] = xb2 >= 12 && xa <= 15;
        }
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const bool h = (g < 4 ? hy0 : hy1) && hx[g & 3];
            const unsigned long long m = __ballot(h);
            if (m) {
                if (h) wlist[(uint32_t)g * list_stride + cnt[g] + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = (uint16_t)(idx * 16u);
                cnt[g] += (uint32_t)__popcll(m);
            }
        }
    }
    uint32_t n_max = 0, n_mine = 0;
#pragma unroll
    for (int g = 0; g < 8; g++) { n_max = max(n_max, cnt[g]); n_mine = grp == (uint32_t)g ? cnt[g] : n_mine; }
    if ((f.dbg_flags & 1) || n_max == 0u) return;
    const uint32_t n_steps = (n_max + 1u) & ~1u;
    for (uint32_t p = n_mine + (lane & 7u); p < n_steps + 2u; p += 8u) my_list[p] = (uint16_t)(PB * 16u);      // the null record
#define GSWT_PREC0(O) (*reinterpret_cast<const float4*>(q0b + (O)))
#define GSWT_PREC1(O) (*reinterpret_cast<const float4*>(q1b + (O)))
#define GSWT_PREC2(O) (*reinterpret_cast<const float4*>(q2b + (O)))
#define GSWT_PRECD(O) (*reinterpret_cast<const float*>(dpb + ((O) >> 2)))
#define GSWT_PSTEP(Q0, Q1, Q2, DV)                                                                  \
    {                                                                                               \
        const float pu_y = fmaf(Q0.y, ly, Q0.z);                                                    \
        const float pv_y = fmaf(Q1.y, ly, Q1.z);                                                    \
        const v2f ppx = pk_fma(pk_splat(Q0.x), lx2, pk_splat(pu_y));                                \
        const v2f ppy = pk_fma(pk_splat(Q1.x), lx2, pk_splat(pv_y));                                \
        const v2f r2 = pk_fma(ppy, ppy, ppx * ppx);                                                 \
        bool c0 = r2.x <= 4.0f, c1 = r2.y <= 4.0f;                                                  \
        if (DEPTH) { c0 = c0 && DV < dbuf2.x; c1 = c1 && DV < dbuf2.y; }                            \
        if (__ballot(c0 || c1) != 0ull) {                                                           \
            const v2f arg = pk_fma(r2, pk_splat(-1.4426950408889634f), pk_splat(Q0.w));             \
            v2f Bv;                                                                                 \
            Bv.x = c0 ? __builtin_amdgcn_exp2f(arg.x) : 0.0f;                                       \
            Bv.y = c1 ? __builtin_amdgcn_exp2f(arg.y) : 0.0f;                                       \
            const v2f wgt = T2 * Bv;                                                                \
            const uint32_t cw = __float_as_uint(Q1.w);                                              \
            ar2 = pk_fma(wgt, pk_splat(COLF ? Q2.x : (float)(cw & 0xFFu)), ar2);                    \
            ag2 = pk_fma(wgt, pk_splat(COLF ? Q2.y : (float)((cw >> 8) & 0xFFu)), ag2);             \
            ab2 = pk_fma(wgt, pk_splat(COLF ? Q2.z : (float)((cw >> 16) & 0xFFu)), ab2);            \
            T2 = T2 - wgt;                                                                          \
        }                                                                                           \
    }
    {
        // Software pipeline of a step pair (i, i + 1): a list entry is read two steps ahead and a record set one step ahead.  The
        // entry only becomes an LDS address (the empty asm: a 32-bit register the compiler cannot look into, so it does not mask the
        // 16-bit load again) AFTER the step that was issued behind it: nothing is waited for right behind its own issue.
        // (k_composite pins the entry where it is loaded; with 8 waves per SIMD the others cover that wait, with 5-6 independent
        // waves it was one exposed LDS latency per step.)
        uint32_t kA = my_list[0], kB = my_list[1];
        asm("" : "+v"(kA)); asm("" : "+v"(kB));
        float4 a0 = GSWT_PREC0(kA), a1 = GSWT_PREC1(kA);
        float4 a2 = make_float4(0.f, 0.f, 0.f, 0.f), b2 = a2;
        if (COLF) a2 = GSWT_PREC2(kA);
        float da = DEPTH ? GSWT_PRECD(kA) : 0.0f, db = 0.0f;
        for (uint32_t i = 0; i < n_steps; i += 2u) {
            const float4 b0 = GSWT_PREC0(kB), b1 = GSWT_PREC1(kB);           // record of step i + 1
            if (COLF) b2 = GSWT_PREC2(kB);
            if (DEPTH) db = GSWT_PRECD(kB);
            kA = my_list[i + 2u];                                          // entry of step i + 2
            GSWT_PSTEP(a0, a1, a2, da)
            asm("" : "+v"(kA));
            a0 = GSWT_PREC0(kA); a1 = GSWT_PREC1(kA);                        // record of step i + 2
            if (COLF) a2 = GSWT_PREC2(kA);
            if (DEPTH) da = GSWT_PRECD(kA);
            kB = my_list[i + 3u];                                          // entry of step i + 3
            GSWT_PSTEP(b0, b1, b2, db)
            asm("" : "+v"(kB));
        }
    }
#undef GSWT_PSTEP
#undef GSWT_PREC0
#undef GSWT_PREC1
#undef GSWT_PREC2
#undef GSWT_PRECD
    if (EARLY && __ballot(T2.x >= t_eps || T2.y >= t_eps) == 0ull) live = false;
}

template <bool EARLY, bool DEPTH, bool COLF, int NB>
__global__ __launch_bounds__(64) void k_composite_p(const Frame f, const uint32_t* __restrict__ item_base, const uint4* __restrict__ item_tab,
                                                     const uint32_t* __restrict__ vals, const Rec* __restrict__ recs,
                                                     const float* __restrict__ depths, const float4* __restrict__ col_f,
                                                     const float4* __restrict__ bg_rgba, const float* __restrict__ bg_depth,
                                                     float4* __restrict__ out, float4* __restrict__ partials, int n_tiles, int out_rows)
{
    constexpr uint32_t PB = (uint32_t)NB * 64u;          // pairs per batch
    constexpr uint32_t kStride = PB + 8u;                // u16 entries per sub-block list (hits + even padding + 4 of prefetch overrun); even: lists are read as u32
    __shared__ float4 s_q0[PB + 1], s_q1[PB + 1];        // [PB] = the null record
    __shared__ uint32_t s_bb[PB];
    __shared__ float4 s_q2[COLF ? PB + 1 : 1];
    __shared__ float s_dep[DEPTH ? PB + 1 : 1];
    __shared__ __attribute__((aligned(4))) uint16_t s_list[8][kStride];
    const uint32_t item = blockIdx.x;
#ifdef GSWT_TRACE
    const bool tr_on = threadIdx.x == 0 && item < kTraceItems;
    const uint32_t tr_item = item;
    unsigned long long tr_walk = 0;
    bool tr_first = true;
    GSWT_TR(0, GSWT_NOW())
#endif
    const uint32_t n_items = item_base[n_tiles];
    const uint4 it = item_tab[item];
    if (item >= n_items) return;
    GSWT_TR(1, GSWT_NOW())
    GSWT_TR(4, it.w - it.z)
#ifdef GSWT_TRACE
    { unsigned hwid, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); GSWT_TR(7, (unsigned long long)hwid | ((unsigned long long)xcc << 32)) }
#endif
    const int tile = (int)it.x;
    const bool multi_seg = (it.y & 1u) != 0u;
    const int tx = tile % f.tiles_x, tyl = tile / f.tiles_x;
    const int sc = f.shard_count <= 1 ? 1 : f.shard_count;
    const int ty = tyl * sc + (sc > 1 ? f.shard_index : 0);
    const int bx = (tx + f.col0) * kTile, by = ty * kTile;
    const uint32_t lane = threadIdx.x;
    const uint32_t grp = lane >> 3, li = lane & 7u;
    // pixel pair of this lane inside a half tile: columns lxi, lxi + 1, row lyi0 (+ 8 in the lower half)
    const int lxi = (int)(grp & 3u) * 4 + (int)(li & 1u) * 2, lyi0 = (int)(grp >> 2) * 4 + (int)(li >> 1);
    v2f lx2; lx2.x = (float)lxi + 0.5f; lx2.y = (float)lxi + 1.5f;
    const float ly0 = (float)lyi0 + 0.5f;
    const float fbx = (float)bx, fby = (float)by;
    const uint2 rg = make_uint2(it.z, it.w);
    v2f T2[2], ar2[2], ag2[2], ab2[2], dbuf2[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int px = bx + lxi, py = by + 8 * h + lyi0;
        const bool in0 = px < f.width && py < f.height, in1 = px + 1 < f.width && py < f.height;
        T2[h].x = (EARLY && !in0) ? 0.0f : 1.0f; T2[h].y = (EARLY && !in1) ? 0.0f : 1.0f;
        ar2[h] = ag2[h] = ab2[h] = pk_splat(0.0f);
        dbuf2[h] = pk_splat(1.0f);
        if (DEPTH && in0) dbuf2[h].x = bg_depth[(size_t)py * f.width + px];
        if (DEPTH && in1) dbuf2[h].y = bg_depth[(size_t)py * f.width + px + 1];
    }
    const float t_eps = f.t_eps;
    bool live0 = true, live1 = true;
    if (lane == 0) {
        s_q0[PB] = make_float4(0.f, 0.f, __builtin_inff(), 0.f);       // r^2 = +inf for every pixel
        s_q1[PB] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (DEPTH) s_dep[PB] = 0.0f;
        if (COLF) s_q2[PB] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 ra[NB], rb[NB], rd[NB];
    float rbw[NB];
    uint32_t slot_nxt[NB];
    const uint32_t last_pair = rg.y - 1u;
#pragma unroll
    for (int k = 0; k < NB; k++) { ra[k] = rb[k] = rd[k] = make_float4(0.f, 0.f, 0.f, 0.f); rbw[k] = 0.f; slot_nxt[k] = 0u; }
    if (rg.x < rg.y) {
        uint32_t s0[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) s0[k] = vals[min(rg.x + (uint32_t)k * 64u + lane, last_pair)];
#pragma unroll
        for (int k = 0; k < NB; k++) slot_nxt[k] = vals[min(rg.x + PB + (uint32_t)k * 64u + lane, last_pair)];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const float4* rp = reinterpret_cast<const float4*>(recs + s0[k]);
            ra[k] = rp[0]; rb[k] = rp[1];
            if (DEPTH) rbw[k] = depths[s0[k]];
            if (COLF) rd[k] = col_f[s0[k]];
        }
    }
    const char* const q0b = reinterpret_cast<const char*>(s_q0);
    const char* const q1b = reinterpret_cast<const char*>(s_q1);
    const char* const q2b = reinterpret_cast<const char*>(s_q2);
    const char* const dpb = reinterpret_cast<const char*>(s_dep);
    for (uint32_t base = rg.x; base < rg.y; base += PB) {
        const uint32_t n = min(PB, rg.y - base);
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const uint32_t idx = (uint32_t)k * 64u + lane;
            if (idx < n) {
                // F3 and the tile-local pixel-centre box: as in k_composite
                const float ox = fmaf(0.5f * f.W, rb[k].x, 0.5f * f.W - fbx), oy = fmaf(-0.5f * f.H, rb[k].y, 0.5f * f.H - fby);   // F3, sequence v3
                const float nku = -fmaf(ra[k].x, ox, ra[k].y * oy);
                const float nkv = -fmaf(ra[k].z, ox, ra[k].w * oy);
                s_q0[idx] = make_float4(ra[k].x, ra[k].y, nku, __builtin_amdgcn_logf(rb[k].z));
                s_q1[idx] = make_float4(ra[k].z, ra[k].w, nkv, rb[k].w);
                const float ria = __builtin_amdgcn_rcpf(fmaf(ra[k].y, ra[k].y, ra[k].x * ra[k].x)), rib = __builtin_amdgcn_rcpf(fmaf(ra[k].w, ra[k].w, ra[k].z * ra[k].z));
                const float qux = ra[k].x * ria, quy = ra[k].y * ria, qwx = ra[k].z * rib, qwy = ra[k].w * rib;
                const float bhx = fmaf(2.0f * __builtin_amdgcn_sqrtf(fmaf(qwx, qwx, qux * qux)), 1.0001f, 0.002f);
                const float bhy = fmaf(2.0f * __builtin_amdgcn_sqrtf(fmaf(qwy, qwy, quy * quy)), 1.0001f, 0.002f);
                const int xa = min(max((int)ceilf((ox - bhx) - 0.5f), -2), 17), xb2 = min(max((int)floorf((ox + bhx) - 0.5f), -2), 17);
                const int ya = min(max((int)ceilf((oy - bhy) - 0.5f), -2), 17), yb = min(max((int)floorf((oy + bhy) - 0.5f), -2), 17);
                s_bb[idx] = (uint32_t)(xa & 0xFF) | ((uint32_t)(xb2 & 0xFF) << 8) | ((uint32_t)(ya & 0xFF) << 16) | ((uint32_t)(yb & 0xFF) << 24);
                if (DEPTH) s_dep[idx] = rbw[k];
                if (COLF) s_q2[idx] = rd[k];
            }
        }
        // one wave: its LDS operations complete in program order; the fences only keep the compiler from moving them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef GSWT_TRACE
        if (tr_first) { GSWT_TR(2, GSWT_NOW()) tr_first = false; }
        const unsigned long long tr_t0 = GSWT_NOW();
#endif
        // the next batch's records and the one-after-next's slot indices are in flight during the walk
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const float4* rp = reinterpret_cast<const float4*>(recs + slot_nxt[k]);
            ra[k] = rp[0]; rb[k] = rp[1];
            if (DEPTH) rbw[k] = depths[slot_nxt[k]];
            if (COLF) rd[k] = col_f[slot_nxt[k]];
        }
#pragma unroll
        for (int k = 0; k < NB; k++) slot_nxt[k] = vals[min(base + 2u * PB + (uint32_t)k * 64u + lane, last_pair)];
        if (!(f.dbg_flags & 2)) {
            if (live0) packed_bin_walk<EARLY, DEPTH, COLF, NB>(f, 0, lx2, ly0, lane, grp, n, q0b, q1b, s_bb, dpb, q2b, &s_list[0][0], kStride, dbuf2[0], t_eps, T2[0], ar2[0], ag2[0], ab2[0], live0);
            if (live1) packed_bin_walk<EARLY, DEPTH, COLF, NB>(f, 8, lx2, ly0 + 8.0f, lane, grp, n, q0b, q1b, s_bb, dpb, q2b, &s_list[0][0], kStride, dbuf2[1], t_eps, T2[1], ar2[1], ag2[1], ab2[1], live1);
        }
#ifdef GSWT_TRACE
        tr_walk += GSWT_NOW() - tr_t0;
#endif
        if (EARLY && !(live0 || live1)) break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    GSWT_TR(3, GSWT_NOW())
    GSWT_TR(6, tr_walk)
    const float k255 = 1.0f / 255.0f;
    // partials / output in k_composite's pixel order (k_combine folds partials[item * 256 + its own thread id]): thread id of
    // pixel (x, y) there = (y >> 2) * 64 + (x >> 2) * 16 + (y & 3) * 4 + (x & 3)
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int lyi = 8 * h + lyi0;
#pragma unroll
        for (int e = 0; e < 2; e++) {
            const int lx_ = lxi + e;
            float cr = e ? ar2[h].y : ar2[h].x, cg = e ? ag2[h].y : ag2[h].x, cb = e ? ab2[h].y : ab2[h].x;
            const float T = e ? T2[h].y : T2[h].x;
            if (!COLF) { cr *= k255; cg *= k255; cb *= k255; }
            if (multi_seg) {
                const uint32_t ptid = (uint32_t)((lyi >> 2) * 64 + (lx_ >> 2) * 16 + (lyi & 3) * 4 + (lx_ & 3));
                partials[(size_t)item * 256u + ptid] = make_float4(cr, cg, cb, T);
                continue;
            }
            const int px = bx + lx_, py = by + lyi;
            if (px < f.width && py < f.height) {
                float4 bg = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bg_rgba) bg = bg_rgba[(size_t)py * f.width + px];
                float4 o;
                o.x = fmaf(T, bg.x, cr);
                o.y = fmaf(T, bg.y, cg);
                o.z = fmaf(T, bg.z, cb);
                o.w = fmaf(T, bg.w, 1.0f - T);
                const int orow = tyl * kTile + lyi;
                if (orow < out_rows) out[(size_t)orow * f.out_w + (px - f.out_x0)] = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// k_composite_p2 (round 3): k_composite's workgroup (shared staging, two barriers per batch) with TWO waves of two-pixel
// lanes instead of four waves of one-pixel lanes: wave h composites the 16x8 half h of the tile with packed_bin_walk.
// 128 threads, NB2 pairs per thread and batch.
// ------------------------------------------------------------------------------------
template <bool EARLY, bool DEPTH, bool COLF, int NB2, int OCC>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(OCC, 8))) void k_composite_p2(const Frame f, const uint32_t* __restrict__ item_base, const uint4* __restrict__ item_tab,
                                                      const uint32_t* __restrict__ vals, const Rec* __restrict__ recs,
                                                      const float* __restrict__ depths, const float4* __restrict__ col_f,
                                                      const float4* __restrict__ bg_rgba, const float* __restrict__ bg_depth,
                                                      float4* __restrict__ out, float4* __restrict__ partials, int n_tiles, int out_rows)
{
    constexpr uint32_t PB = (uint32_t)NB2 * 128u;        // pairs per batch
    constexpr int NBW = NB2 * 2;                         // 64-pair chunks per batch (what packed_bin_walk bins)
    constexpr uint32_t kStride = PB + 8u;
    __shared__ float4 s_q0[PB + 1], s_q1[PB + 1];        // [PB] = the null record
    __shared__ uint32_t s_bb[PB];
    __shared__ float4 s_q2[COLF ? PB + 1 : 1];
    __shared__ float s_dep[DEPTH ? PB + 1 : 1];
    __shared__ uint16_t s_list[2][8][kStride];
    const uint32_t item = blockIdx.x;
    const uint32_t n_items = item_base[n_tiles];
    const uint4 it = item_tab[item];
    if (item >= n_items) return;
    const int tile = (int)it.x;
    const bool multi_seg = (it.y & 1u) != 0u;
    const int tx = tile % f.tiles_x, tyl = tile / f.tiles_x;
    const int sc = f.shard_count <= 1 ? 1 : f.shard_count;
    const int ty = tyl * sc + (sc > 1 ? f.shard_index : 0);
    const int bx = (tx + f.col0) * kTile, by = ty * kTile;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint32_t grp = lane >> 3, li = lane & 7u;
    const int lxi = (int)(grp & 3u) * 4 + (int)(li & 1u) * 2, lyi = (int)wave * 8 + (int)(grp >> 2) * 4 + (int)(li >> 1);
    v2f lx2; lx2.x = (float)lxi + 0.5f; lx2.y = (float)lxi + 1.5f;
    const float ly = (float)lyi + 0.5f;
    const float fbx = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)bx)));
    const float fby = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)by)));
    const uint2 rg = make_uint2(it.z, it.w);
    v2f T2, ar2 = pk_splat(0.0f), ag2 = ar2, ab2 = ar2, dbuf2 = pk_splat(1.0f);
    {
        const int px = bx + lxi, py = by + lyi;
        const bool in0 = px < f.width && py < f.height, in1 = px + 1 < f.width && py < f.height;
        T2.x = (EARLY && !in0) ? 0.0f : 1.0f; T2.y = (EARLY && !in1) ? 0.0f : 1.0f;
        if (DEPTH && in0) dbuf2.x = bg_depth[(size_t)py * f.width + px];
        if (DEPTH && in1) dbuf2.y = bg_depth[(size_t)py * f.width + px + 1];
    }
    const float t_eps = f.t_eps;
    bool wave_live = true;
    if (tid == 0) {
        s_q0[PB] = make_float4(0.f, 0.f, __builtin_inff(), 0.f);
        s_q1[PB] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (DEPTH) s_dep[PB] = 0.0f;
        if (COLF) s_q2[PB] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 ra[NB2], rb[NB2], rd[NB2];
    float rbw[NB2];
    uint32_t slot_nxt[NB2];
    const uint32_t last_pair = rg.y - 1u;
#pragma unroll
    for (int k = 0; k < NB2; k++) { ra[k] = rb[k] = rd[k] = make_float4(0.f, 0.f, 0.f, 0.f); rbw[k] = 0.f; slot_nxt[k] = 0u; }
    if (rg.x < rg.y) {
        uint32_t s0[NB2];
#pragma unroll
        for (int k = 0; k < NB2; k++) s0[k] = vals[min(rg.x + (uint32_t)k * 128u + tid, last_pair)];
#pragma unroll
        for (int k = 0; k < NB2; k++) slot_nxt[k] = vals[min(rg.x + PB + (uint32_t)k * 128u + tid, last_pair)];
#pragma unroll
        for (int k = 0; k < NB2; k++) {
            const float4* rp = reinterpret_cast<const float4*>(recs + s0[k]);
            ra[k] = rp[0]; rb[k] = rp[1];
            if (DEPTH) rbw[k] = depths[s0[k]];
            if (COLF) rd[k] = col_f[s0[k]];
        }
    }
    const char* const q0b = reinterpret_cast<const char*>(s_q0);
    const char* const q1b = reinterpret_cast<const char*>(s_q1);
    const char* const q2b = reinterpret_cast<const char*>(s_q2);
    const char* const dpb = reinterpret_cast<const char*>(s_dep);
    for (uint32_t base = rg.x; base < rg.y; base += PB) {
        const uint32_t n = min(PB, rg.y - base);
#pragma unroll
        for (int k = 0; k < NB2; k++) {
            const uint32_t idx = (uint32_t)k * 128u + tid;
            if (idx < n) {
                const float ox = fmaf(0.5f * f.W, rb[k].x, 0.5f * f.W - fbx), oy = fmaf(-0.5f * f.H, rb[k].y, 0.5f * f.H - fby);   // F3, sequence v3
                const float nku = -fmaf(ra[k].x, ox, ra[k].y * oy);
                const float nkv = -fmaf(ra[k].z, ox, ra[k].w * oy);
                s_q0[idx] = make_float4(ra[k].x, ra[k].y, nku, __builtin_amdgcn_logf(rb[k].z));
                s_q1[idx] = make_float4(ra[k].z, ra[k].w, nkv, rb[k].w);
                const float ria = __builtin_amdgcn_rcpf(fmaf(ra[k].y, ra[k].y, ra[k].x * ra[k].x)), rib = __builtin_amdgcn_rcpf(fmaf(ra[k].w, ra[k].w, ra[k].z * ra[k].z));
                const float qux = ra[k].x * ria, quy = ra[k].y * ria, qwx = ra[k].z * rib, qwy = ra[k].w * rib;
                const float bhx = fmaf(2.0f * __builtin_amdgcn_sqrtf(fmaf(qwx, qwx, qux * qux)), 1.0001f, 0.002f);
                const float bhy = fmaf(2.0f * __builtin_amdgcn_sqrtf(fmaf(qwy, qwy, quy * quy)), 1.0001f, 0.002f);
                const int xa = min(max((int)ceilf((ox - bhx) - 0.5f), -2), 17), xb2 = min(max((int)floorf((ox + bhx) - 0.5f), -2), 17);
                const int ya = min(max((int)ceilf((oy - bhy) - 0.5f), -2), 17), yb = min(max((int)floorf((oy + bhy) - 0.5f), -2), 17);
                s_bb[idx] = (uint32_t)(xa & 0xFF) | ((uint32_t)(xb2 & 0xFF) << 8) | ((uint32_t)(ya & 0xFF) << 16) | ((uint32_t)(yb & 0xFF) << 24);
                if (DEPTH) s_dep[idx] = rbw[k];
                if (COLF) s_q2[idx] = rd[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NB2; k++) {
            const float4* rp = reinterpret_cast<const float4*>(recs + slot_nxt[k]);
            ra[k] = rp[0]; rb[k] = rp[1];
            if (DEPTH) rbw[k] = depths[slot_nxt[k]];
            if (COLF) rd[k] = col_f[slot_nxt[k]];
        }
#pragma unroll
        for (int k = 0; k < NB2; k++) slot_nxt[k] = vals[min(base + 2u * PB + (uint32_t)k * 128u + tid, last_pair)];
        if (wave_live && !(f.dbg_flags & 2))
            packed_bin_walk<EARLY, DEPTH, COLF, NBW>(f, (int)wave * 8, lx2, ly, lane, grp, n, q0b, q1b, s_bb, dpb, q2b, &s_list[wave][0][0], kStride, dbuf2, t_eps, T2, ar2, ag2, ab2, wave_live);
        if (base + PB >= rg.y) break;
        if (EARLY) { if (__syncthreads_and(wave_live ? 0 : 1)) break; }
        else __syncthreads();
    }
    const float k255 = 1.0f / 255.0f;
#pragma unroll
    for (int e = 0; e < 2; e++) {
        const int lx_ = lxi + e;
        float cr = e ? ar2.y : ar2.x, cg = e ? ag2.y : ag2.x, cb = e ? ab2.y : ab2.x;
        const float T = e ? T2.y : T2.x;
        if (!COLF) { cr *= k255; cg *= k255; cb *= k255; }
        if (multi_seg) {
            const uint32_t ptid = (uint32_t)((lyi >> 2) * 64 + (lx_ >> 2) * 16 + (lyi & 3) * 4 + (lx_ & 3));
            partials[(size_t)item * 256u + ptid] = make_float4(cr, cg, cb, T);
            continue;
        }
        const int px = bx + lx_, py = by + lyi;
        if (px < f.width && py < f.height) {
            float4 bg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bg_rgba) bg = bg_rgba[(size_t)py * f.width + px];
            float4 o;
            o.x = fmaf(T, bg.x, cr);
            o.y = fmaf(T, bg.y, cg);
            o.z = fmaf(T, bg.z, cb);
            o.w = fmaf(T, bg.w, 1.0f - T);
            const int orow = tyl * kTile + lyi;
            if (orow < out_rows) out[(size_t)orow * f.out_w + (px - f.out_x0)] = o;
        }
    }
}

// ------------------------------------------------------------------------------------
// k_composite_s (round 3): INDEPENDENT STRIP WAVES.  A work item (tile, segment) is composited by 4 / SPW single-wave
// workgroups, each owning SPW of the tile's four 16x4 strips (one pixel per lane, four 16-lane groups = four 4x4
// sub-blocks per strip, the bin + walk of k_composite).  Every wave gathers and stages the item's pairs ITSELF, 64 x NB
// at a time; nothing is shared between the waves of an item, so there is no workgroup barrier and a wave never waits for
// the slowest strip of its tile (tools/composite_trace.py: in k_composite a batch lasts as long as its longest of 16
// lists -- 51 % of the group-slots of a workgroup are idle or padding).  The price is the staging arithmetic and the
// record gathers 4 / SPW times over; the waves of an item are launched on ONE XCD (blocks b, b + 8, ...) so that the
// repeated gathers meet in that XCD's L2.
// ------------------------------------------------------------------------------------
template <bool EARLY, bool DEPTH, bool COLF, int NB>
__device__ __forceinline__ void strip_bin_walk(const Frame& f, const int r0, const float lx, const float ly, const uint32_t lane, const uint32_t grp,
                                               const uint32_t n, const char* const q0b, const char* const q1b, const uint32_t* const s_bb,
                                               const char* const dpb, const char* const q2b, uint16_t* const wlist, const uint32_t list_stride,
                                               const float dbuf, const float t_eps, float& T, float& ar, float& ag, float& ab, bool& live)
{
    constexpr uint32_t PB = (uint32_t)NB * 64u;
    uint16_t* const my_list = wlist + grp * list_stride;
    uint32_t cnt0 = 0, cnt1 = 0, cnt2 = 0, cnt3 = 0;
#pragma unroll
    for (int c = 0; c < NB; c++) {
        const uint32_t idx = (uint32_t)c * 64u + lane;
        bool h0 = false, h1 = false, h2 = false, h3 = false;
        if (idx < n) {
            const uint32_t bb = s_bb[idx];
            const int xa = __builtin_amdgcn_sbfe((int)bb, 0, 8), xb2 = __builtin_amdgcn_sbfe((int)bb, 8, 8);
            const int ya = __builtin_amdgcn_sbfe((int)bb, 16, 8), yb = __builtin_amdgcn_sbfe((int)bb, 24, 8);
            const bool hy = yb >= r0 && ya <= r0 + 3;
            h0 = hy && xa <= 3 && xb2 >= 0;
            h1 = hy && xb2 >= 4 && xa <= 7;
            h2 = hy && xb2 >= 8 && xa <= 11;
            h3 = hy && xb2 >= 12 && xa <= 15;
        }
        const unsigned long long m0 = __ballot(h0), m1 = __ballot(h1), m2 = __ballot(h2), m3 = __ballot(h3);
#define GSWT_APPEND(H, M, CNT, G)                                                                                         \
        if (M) {                                                                                                            \
            if (H) wlist[(G) * list_stride + (CNT) + __builtin_amdgcn_mbcnt_hi((uint32_t)((M) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(M), 0u))] = (uint16_t)(idx * 16u); \
            CNT += (uint32_t)__popcll(M);                                                                                   \
        }
        GSWT_APPEND(h0, m0, cnt0, 0u) GSWT_APPEND(h1, m1, cnt1, 1u) GSWT_APPEND(h2, m2, cnt2, 2u) GSWT_APPEND(h3, m3, cnt3, 3u)
#undef GSWT_APPEND
    }
    const uint32_t n_mine = grp == 0u ? cnt0 : grp == 1u ? cnt1 : grp == 2u ? cnt2 : cnt3;
    const uint32_t n_max = max(max(cnt0, cnt1), max(cnt2, cnt3));
    if ((f.dbg_flags & 1) || n_max == 0u) return;
    const uint32_t n_steps = (n_max + 1u) & ~1u;
    for (uint32_t p = n_mine + (lane & 15u); p < n_steps + 2u; p += 16u) my_list[p] = (uint16_t)(PB * 16u);      // the null record
#define GSWT_SREC0(O) (*reinterpret_cast<const float4*>(q0b + (O)))
#define GSWT_SREC1(O) (*reinterpret_cast<const float4*>(q1b + (O)))
#define GSWT_SREC2(O) (*reinterpret_cast<const float4*>(q2b + (O)))
#define GSWT_SRECD(O) (*reinterpret_cast<const float*>(dpb + ((O) >> 2)))
#define GSWT_SSTEP(Q0, Q1, Q2, DV)                                                                  \
    {                                                                                               \
        const float pu_y = fmaf(Q0.y, ly, Q0.z);                                                    \
        const float pv_y = fmaf(Q1.y, ly, Q1.z);                                                    \
        const float ppx = fmaf(Q0.x, lx, pu_y);                                                     \
        const float ppy = fmaf(Q1.x, lx, pv_y);                                                     \
        const float r2 = fmaf(ppy, ppy, ppx * ppx);                                                 \
        bool cover = r2 <= 4.0f;                                                                    \
        if (DEPTH) cover = cover && DV < dbuf;                                                      \
        if (__ballot(cover) != 0ull) {                                                              \
            const float e = __builtin_amdgcn_exp2f(fmaf(r2, -1.4426950408889634f, Q0.w));           \
            const float Bv = cover ? e : 0.0f;                                                      \
            const float wgt = T * Bv;                                                               \
            const uint32_t cw = __float_as_uint(Q1.w);                                              \
            ar = fmaf(wgt, COLF ? Q2.x : (float)(cw & 0xFFu), ar);                                  \
            ag = fmaf(wgt, COLF ? Q2.y : (float)((cw >> 8) & 0xFFu), ag);                           \
            ab = fmaf(wgt, COLF ? Q2.z : (float)((cw >> 16) & 0xFFu), ab);                          \
            T = T - wgt;                                                                            \
        }                                                                                           \
    }
    {
        uint32_t kA = my_list[0], kB = my_list[1];
        asm("" : "+v"(kA)); asm("" : "+v"(kB));
        float4 a0 = GSWT_SREC0(kA), a1 = GSWT_SREC1(kA);
        float4 a2 = make_float4(0.f, 0.f, 0.f, 0.f), b2 = a2;
        if (COLF) a2 = GSWT_SREC2(kA);
        float da = DEPTH ? GSWT_SRECD(kA) : 0.0f, db = 0.0f;
        for (uint32_t i = 0; i < n_steps; i += 2u) {
            const float4 b0 = GSWT_SREC0(kB), b1 = GSWT_SREC1(kB);
            if (COLF) b2 = GSWT_SREC2(kB);
            if (DEPTH) db = GSWT_SRECD(kB);
            kA = my_list[i + 2u];
            asm("" : "+v"(kA));
            GSWT_SSTEP(a0, a1, a2, da)
            a0 = GSWT_SREC0(kA); a1 = GSWT_SREC1(kA);
            if (COLF) a2 = GSWT_SREC2(kA);
            if (DEPTH) da = GSWT_SRECD(kA);
            kB = my_list[i + 3u];
            asm("" : "+v"(kB));
            GSWT_SSTEP(b0, b1, b2, db)
        }
    }
#undef GSWT_SSTEP
#undef GSWT_SREC0
#undef GSWT_SREC1
#undef GSWT_SREC2
#undef GSWT_SRECD
    if (EARLY && __ballot(T >= t_eps) == 0ull) live = false;
}

template <bool EARLY, bool DEPTH, bool COLF, int NB, int SPW>
__global__ __launch_bounds__(64) void k_composite_s(const Frame f, const uint32_t* __restrict__ item_base, const uint4* __restrict__ item_tab,
                                                     const uint32_t* __restrict__ vals, const Rec* __restrict__ recs,
                                                     const float* __restrict__ depths, const float4* __restrict__ col_f,
                                                     const float4* __restrict__ bg_rgba, const float* __restrict__ bg_depth,
                                                     float4* __restrict__ out, float4* __restrict__ partials, int n_tiles, int out_rows)
{
    constexpr uint32_t PB = (uint32_t)NB * 64u;          // pairs per batch
    constexpr uint32_t kStride = PB + 8u;
    constexpr uint32_t kParts = 4u / (uint32_t)SPW;      // waves per item
    __shared__ float4 s_q0[PB + 1], s_q1[PB + 1];        // [PB] = the null record
    __shared__ uint32_t s_bb[PB];
    __shared__ float4 s_q2[COLF ? PB + 1 : 1];
    __shared__ float s_dep[DEPTH ? PB + 1 : 1];
    __shared__ uint16_t s_list[4][kStride];
    // blocks b, b + 8, ..., b + 8 (kParts - 1) are the waves of one item: the dispatcher deals consecutive blocks round-robin over the 8 XCDs
    const uint32_t q = blockIdx.x >> 3;
    const uint32_t part = q % kParts;
    const uint32_t item = (q / kParts) * 8u + (blockIdx.x & 7u);
#ifdef GSWT_TRACE
    const bool tr_on = threadIdx.x == 0 && part == 0u && item < kTraceItems;
    const uint32_t tr_item = item;
    unsigned long long tr_walk = 0;
    bool tr_first = true;
    GSWT_TR(0, GSWT_NOW())
#endif
    const uint32_t n_items = item_base[n_tiles];
    if (item >= n_items) return;
    const uint4 it = item_tab[item];
    GSWT_TR(1, GSWT_NOW())
    GSWT_TR(4, it.w - it.z)
    const int tile = (int)it.x;
    const bool multi_seg = (it.y & 1u) != 0u;
    const int tx = tile % f.tiles_x, tyl = tile / f.tiles_x;
    const int sc = f.shard_count <= 1 ? 1 : f.shard_count;
    const int ty = tyl * sc + (sc > 1 ? f.shard_index : 0);
    const int bx = (tx + f.col0) * kTile, by = ty * kTile;
    const uint32_t lane = threadIdx.x;
    const uint32_t grp = lane >> 4, gi = lane & 15u;
    const int s_first = (int)part * SPW;                 // first strip of this wave
    const int lxi = (int)grp * 4 + (int)(gi & 3u), lyi0 = s_first * 4 + (int)(gi >> 2);
    const float lx = (float)lxi + 0.5f, ly0 = (float)lyi0 + 0.5f;
    const float fbx = (float)bx, fby = (float)by;
    const int row_lo = s_first * 4, row_hi = row_lo + 4 * SPW - 1;      // tile-local pixel rows of this wave
    const uint2 rg = make_uint2(it.z, it.w);
    float T[SPW], ar[SPW], ag[SPW], ab[SPW], dbuf[SPW];
    bool live[SPW];
#pragma unroll
    for (int s = 0; s < SPW; s++) {
        const int px = bx + lxi, py = by + 4 * s + lyi0;
        const bool inside = px < f.width && py < f.height;
        T[s] = (EARLY && !inside) ? 0.0f : 1.0f; ar[s] = ag[s] = ab[s] = 0.0f;
        dbuf[s] = 1.0f;
        live[s] = true;
        if (DEPTH && inside) dbuf[s] = bg_depth[(size_t)py * f.width + px];
    }
    const float t_eps = f.t_eps;
    if (lane == 0) {
        s_q0[PB] = make_float4(0.f, 0.f, __builtin_inff(), 0.f);       // r^2 = +inf for every pixel
        s_q1[PB] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (DEPTH) s_dep[PB] = 0.0f;
        if (COLF) s_q2[PB] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 ra[NB], rb[NB], rd[NB];
    float rbw[NB];
    uint32_t slot_nxt[NB];
    const uint32_t last_pair = rg.y - 1u;
#pragma unroll
    for (int k = 0; k < NB; k++) { ra[k] = rb[k] = rd[k] = make_float4(0.f, 0.f, 0.f, 0.f); rbw[k] = 0.f; slot_nxt[k] = 0u; }
    if (rg.x < rg.y) {
        uint32_t s0[NB];
#pragma unroll
        for (int k = 0; k < NB; k++) s0[k] = vals[min(rg.x + (uint32_t)k * 64u + lane, last_pair)];
#pragma unroll
        for (int k = 0; k < NB; k++) slot_nxt[k] = vals[min(rg.x + PB + (uint32_t)k * 64u + lane, last_pair)];
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const float4* rp = reinterpret_cast<const float4*>(recs + s0[k]);
            ra[k] = rp[0]; rb[k] = rp[1];
            if (DEPTH) rbw[k] = depths[s0[k]];
            if (COLF) rd[k] = col_f[s0[k]];
        }
    }
    const char* const q0b = reinterpret_cast<const char*>(s_q0);
    const char* const q1b = reinterpret_cast<const char*>(s_q1);
    const char* const q2b = reinterpret_cast<const char*>(s_q2);
    const char* const dpb = reinterpret_cast<const char*>(s_dep);
    for (uint32_t base = rg.x; base < rg.y; base += PB) {
        const uint32_t n = min(PB, rg.y - base);
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const uint32_t idx = (uint32_t)k * 64u + lane;
            if (idx < n) {
                // the tile-local pixel-centre box first (k_composite's): a pair that misses this wave's rows is not staged
                const float ox = fmaf(0.5f * f.W, rb[k].x, 0.5f * f.W - fbx), oy = fmaf(-0.5f * f.H, rb[k].y, 0.5f * f.H - fby);   // F3, sequence v3
                const float ria = __builtin_amdgcn_rcpf(fmaf(ra[k].y, ra[k].y, ra[k].x * ra[k].x)), rib = __builtin_amdgcn_rcpf(fmaf(ra[k].w, ra[k].w, ra[k].z * ra[k].z));
                const float qux = ra[k].x * ria, quy = ra[k].y * ria, qwx = ra[k].z * rib, qwy = ra[k].w * rib;
                const float bhx = fmaf(2.0f * __builtin_amdgcn_sqrtf(fmaf(qwx, qwx, qux * qux)), 1.0001f, 0.002f);
                const float bhy = fmaf(2.0f * __builtin_amdgcn_sqrtf(fmaf(qwy, qwy, quy * quy)), 1.0001f, 0.002f);
                const int xa = min(max((int)ceilf((ox - bhx) - 0.5f), -2), 17), xb2 = min(max((int)floorf((ox + bhx) - 0.5f), -2), 17);
                const int ya = min(max((int)ceilf((oy - bhy) - 0.5f), -2), 17), yb = min(max((int)floorf((oy + bhy) - 0.5f), -2), 17);
                s_bb[idx] = (uint32_t)(xa & 0xFF) | ((uint32_t)(xb2 & 0xFF) << 8) | ((uint32_t)(ya & 0xFF) << 16) | ((uint32_t)(yb & 0xFF) << 24);
                if (SPW == 4 || (yb >= row_lo && ya <= row_hi)) {
                    const float nku = -fmaf(ra[k].x, ox, ra[k].y * oy);      // F3
                    const float nkv = -fmaf(ra[k].z, ox, ra[k].w * oy);
                    s_q0[idx] = make_float4(ra[k].x, ra[k].y, nku, __builtin_amdgcn_logf(rb[k].z));
                    s_q1[idx] = make_float4(ra[k].z, ra[k].w, nkv, rb[k].w);
                    if (DEPTH) s_dep[idx] = rbw[k];
                    if (COLF) s_q2[idx] = rd[k];
                }
            }
        }
        // one wave: its LDS operations complete in program order; the fences only keep the compiler from moving them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef GSWT_TRACE
        if (tr_first) { GSWT_TR(2, GSWT_NOW()) tr_first = false; }
        const unsigned long long tr_t0 = GSWT_NOW();
#endif
#pragma unroll
        for (int k = 0; k < NB; k++) {
            const float4* rp = reinterpret_cast<const float4*>(recs + slot_nxt[k]);
            ra[k] = rp[0]; rb[k] = rp[1];
            if (DEPTH) rbw[k] = depths[slot_nxt[k]];
            if (COLF) rd[k] = col_f[slot_nxt[k]];
        }
#pragma unroll
        for (int k = 0; k < NB; k++) slot_nxt[k] = vals[min(base + 2u * PB + (uint32_t)k * 64u + lane, last_pair)];
        bool any_live = false;
        if (!(f.dbg_flags & 2)) {
#pragma unroll
            for (int s = 0; s < SPW; s++) {
                if (live[s]) strip_bin_walk<EARLY, DEPTH, COLF, NB>(f, row_lo + 4 * s, lx, ly0 + 4.0f * (float)s, lane, grp, n, q0b, q1b, s_bb, dpb, q2b, &s_list[0][0], kStride,
                                                                  dbuf[s], t_eps, T[s], ar[s], ag[s], ab[s], live[s]);
                any_live = any_live || live[s];
            }
        }
#ifdef GSWT_TRACE
        tr_walk += GSWT_NOW() - tr_t0;
#endif
        if (EARLY && !any_live && !(f.dbg_flags & 2)) break;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    GSWT_TR(3, GSWT_NOW())
    GSWT_TR(6, tr_walk)
    const float k255 = 1.0f / 255.0f;
#pragma unroll
    for (int s = 0; s < SPW; s++) {
        float cr = ar[s], cg = ag[s], cb = ab[s];
        if (!COLF) { cr *= k255; cg *= k255; cb *= k255; }
        const int lyi = 4 * s + lyi0;
        if (multi_seg) {
            partials[(size_t)item * 256u + (uint32_t)(s_first + s) * 64u + lane] = make_float4(cr, cg, cb, T[s]);
            continue;
        }
        const int px = bx + lxi, py = by + lyi;
        if (px < f.width && py < f.height) {
            float4 bg = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bg_rgba) bg = bg_rgba[(size_t)py * f.width + px];
            float4 o;
            o.x = fmaf(T[s], bg.x, cr);
            o.y = fmaf(T[s], bg.y, cg);
            o.z = fmaf(T[s], bg.z, cb);
            o.w = fmaf(T[s], bg.w, 1.0f - T[s]);
            const int orow = tyl * kTile + lyi;
            if (orow < out_rows) out[(size_t)orow * f.out_w + (px - f.out_x0)] = o;
        }
    }
}

