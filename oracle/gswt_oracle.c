/*
 * gswt_oracle.c -- CPU ORACLE for the GSWT hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the reference's algorithm
 * (zengyf131/gswt_renderer: src/gswt.wgsl, src/renderer.rs, src/scene.rs,
 * src/utils.rs), written from the source text.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's cpu_baseline leg may load it.  The product
 * (gswt_renderer_amd/) never links, imports or calls anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned by the reference".  The reference (Rust ->
 * wasm32 + WGSL on wgpu) cannot be compiled or run in this environment and it
 * ships no tests, golden images or known-answer vectors.  This restatement is
 * pinned instead by known-answer tests derived from the source text (tests/
 * test_oracle_kat.py: K1..K15 of SURVEY.md section 8c) and by committed golden
 * fixtures generated from it (tests/golden/).
 *
 * Arithmetic: every float operation below is IEEE-754 binary32, one rounding per
 * written operator, no contraction (build with -ffp-contract=off).  Where a fused
 * multiply-add is part of the canonical sequence it is written fmaf() explicitly.
 * DESIGN.md "Canonical float sequences" lists the sequences the HIP kernels must
 * reproduce bit-for-bit (everything that feeds a discontinuous decision: culling,
 * the |p|^2 <= 4 coverage test, the depth test).
 *
 * Conventions: matrices are column-major, m[4*c + r] (cgmath / WGSL layout).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Bit-level helpers                                                          */
/* ------------------------------------------------------------------------- */

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* halfToFloat, gswt.wgsl:478-494.  NOT IEEE: subnormals scale by 2^-15 (not
 * 2^-14) and Inf/NaN decode to 0.  pow(2, k) with integer k is exact -> ldexpf. */
ORC_API float orc_half_to_float(uint32_t h)
{
    float s = (float)((h >> 15) & 0x1u);
    uint32_t e = (h >> 10) & 0x1Fu;
    float f = (float)(h & 0x3FFu);
    float sign = 1.0f - 2.0f * s;
    if (e == 0u) {
        return ldexpf(1.0f, -15) * (f / 1024.0f) * sign;
    } else if (e == 31u) {
        return 0.0f;
    }
    return ldexpf(1.0f, (int)e - 15) * (1.0f + f / 1024.0f) * sign;
}

/* half::f16::from_f32 (half 2.7.1): IEEE binary16, round-to-nearest-even.
 * Call site utils.rs:66-73. */
ORC_API uint32_t orc_float_to_half(float value)
{
    uint32_t x = f2u(value);
    uint32_t sign = x & 0x80000000u;
    uint32_t exp = x & 0x7F800000u;
    uint32_t man = x & 0x007FFFFFu;
    if (exp == 0x7F800000u) {                       /* Inf / NaN */
        uint32_t nan_bit = man == 0 ? 0 : 0x0200u;
        return (sign >> 16) | 0x7C00u | nan_bit | (man >> 13);
    }
    uint32_t half_sign = sign >> 16;
    int32_t unbiased = (int32_t)(exp >> 23) - 127;
    int32_t half_exp = unbiased + 15;
    if (half_exp >= 0x1F) return half_sign | 0x7C00u;  /* overflow -> Inf */
    if (half_exp <= 0) {                            /* subnormal or zero */
        if (14 - half_exp > 24) return half_sign;
        man |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - half_exp);
        uint32_t half_man = man >> shift;
        uint32_t round_bit = 1u << (shift - 1);
        if ((man & round_bit) != 0 && (man & (3 * round_bit - 1)) != 0) half_man += 1;
        return half_sign | half_man;
    }
    uint32_t half_e = (uint32_t)half_exp << 10;
    uint32_t half_man = man >> 13;
    uint32_t round_bit = 0x00001000u;
    if ((man & round_bit) != 0 && (man & (3 * round_bit - 1)) != 0)
        return (half_sign | half_e | half_man) + 1;
    return half_sign | half_e | half_man;
}

/* pack_half_2x16, utils.rs:66-73 */
ORC_API uint32_t orc_pack_half_2x16(float x, float y)
{
    return orc_float_to_half(x) | (orc_float_to_half(y) << 16);
}

/* ------------------------------------------------------------------------- */
/* Scene::generate_texture, scene.rs:306-411                                  */
/* rows32: n rows of [pos f32x3][scale f32x3][rgba u8x4][rot u8x4]            */
/* tex   : 8 u32 per splat [pos x3][0][ab][cd][ef][rgba]                      */
/* ------------------------------------------------------------------------- */
ORC_API void orc_generate_texture(const uint8_t *rows32, size_t n, uint32_t *tex)
{
    for (size_t i = 0; i < n; i++) {
        const uint8_t *row = rows32 + 32 * i;
        float fb[6];
        memcpy(fb, row, 24);
        uint32_t *t = tex + 8 * i;
        memcpy(t, row, 12);                 /* position, :330-336 */
        t[3] = 0;
        memcpy(&t[7], row + 24, 4);         /* rgba, :341-349 */
        float scale[3] = { fb[3], fb[4], fb[5] };
        float rot[4];
        for (int k = 0; k < 4; k++)          /* :361-367, no renormalisation */
            rot[k] = ((float)row[28 + k] / 255.0f) * 2.0f - 1.0f;
        /* R column-major, :369-380 */
        float r[9];
        r[0] = 1.0f - 2.0f * (rot[2] * rot[2] + rot[3] * rot[3]);
        r[1] = 2.0f * (rot[1] * rot[2] + rot[0] * rot[3]);
        r[2] = 2.0f * (rot[1] * rot[3] - rot[0] * rot[2]);
        r[3] = 2.0f * (rot[1] * rot[2] - rot[0] * rot[3]);
        r[4] = 1.0f - 2.0f * (rot[1] * rot[1] + rot[3] * rot[3]);
        r[5] = 2.0f * (rot[2] * rot[3] + rot[0] * rot[1]);
        r[6] = 2.0f * (rot[1] * rot[3] + rot[0] * rot[2]);
        r[7] = 2.0f * (rot[2] * rot[3] - rot[0] * rot[1]);
        r[8] = 1.0f - 2.0f * (rot[1] * rot[1] + rot[2] * rot[2]);
        /* m = r * diag(scale), cgmath Matrix3 * Matrix3: each element is
         * r[0][row]*s[c][0] + r[1][row]*s[c][1] + r[2][row]*s[c][2] with the two
         * off-diagonal zeros of s -> exactly r[c][row]*scale[c] (x*0 adds +-0). */
        float m[9];
        for (int c = 0; c < 3; c++)
            for (int rr = 0; rr < 3; rr++) {
                float acc = 0.0f;
                for (int k = 0; k < 3; k++) {
                    float sk = (k == c) ? scale[c] : 0.0f;
                    float term = r[3 * k + rr] * sk;
                    acc = (k == 0) ? term : acc + term;
                }
                m[3 * c + rr] = acc;
            }
        /* sigma, :391-398 */
        float sigma[6];
        sigma[0] = m[0] * m[0] + m[3] * m[3] + m[6] * m[6];
        sigma[1] = m[0] * m[1] + m[3] * m[4] + m[6] * m[7];
        sigma[2] = m[0] * m[2] + m[3] * m[5] + m[6] * m[8];
        sigma[3] = m[1] * m[1] + m[4] * m[4] + m[7] * m[7];
        sigma[4] = m[1] * m[2] + m[4] * m[5] + m[7] * m[8];
        sigma[5] = m[2] * m[2] + m[5] * m[5] + m[8] * m[8];
        t[4] = orc_pack_half_2x16(4.0f * sigma[0], 4.0f * sigma[1]);  /* :403-405 */
        t[5] = orc_pack_half_2x16(4.0f * sigma[2], 4.0f * sigma[3]);
        t[6] = orc_pack_half_2x16(4.0f * sigma[4], 4.0f * sigma[5]);
    }
}

/* ------------------------------------------------------------------------- */
/* Scene::sort_raw_depth_vec, scene.rs:655-698 (same kernel as sort_self      */
/* :557-583).  depths: concatenated segments; order_out[j] = index into the    */
/* concatenation (caller maps back to (segment, index-in-segment)).            */
/* Rust `as i32` on f32 saturates and maps NaN to 0.                           */
/* ------------------------------------------------------------------------- */
static inline int32_t rust_f32_as_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}

ORC_API void orc_sort_raw_depth(const int32_t *depths, size_t n, uint32_t *order_out)
{
    if (n == 0) return;
    int32_t mn = depths[0], mx = depths[0];
    for (size_t i = 1; i < n; i++) {
        if (depths[i] < mn) mn = depths[i];
        if (depths[i] > mx) mx = depths[i];
    }
    const int32_t size16 = 65536;
    /* (max - min) is an i32 subtraction in Rust (debug builds would panic on
     * overflow; release wraps).  Raw depths are bounded by 4096*|pos| so no wrap. */
    float depth_inv = (float)(size16 - 1) / (float)(int32_t)(mx - mn);
    uint32_t *counts = (uint32_t *)calloc((size_t)size16, 4);
    uint32_t *starts = (uint32_t *)calloc((size_t)size16, 4);
    int32_t *bucket = (int32_t *)malloc(n * 4);
    for (size_t i = 0; i < n; i++) {
        float v = floorf((float)(int32_t)(depths[i] - mn) * depth_inv);
        int32_t d = rust_f32_as_i32(v);
        if (d < 0) d = 0;
        if (d > size16 - 1) d = size16 - 1;
        bucket[i] = d;
        counts[d]++;
    }
    for (int32_t i = 1; i < size16; i++) starts[i] = starts[i - 1] + counts[i - 1];
    for (size_t i = 0; i < n; i++) {
        uint32_t j = starts[bucket[i]]++;
        order_out[j] = (uint32_t)i;
    }
    for (size_t a = 0, b = n - 1; a < b; a++, b--) {   /* depth_index.reverse() */
        uint32_t t = order_out[a]; order_out[a] = order_out[b]; order_out[b] = t;
    }
    free(counts); free(starts); free(bucket);
}

/* raw depth of sort_self, scene.rs:537-552: ((vp[2]x + vp[6]y + vp[10]z)*4096) as i32 */
ORC_API void orc_raw_depth(const uint8_t *rows32, size_t n, const float *view_proj16, int32_t *out)
{
    for (size_t i = 0; i < n; i++) {
        float p[3];
        memcpy(p, rows32 + 32 * i, 12);
        float d = (view_proj16[2] * p[0] + view_proj16[6] * p[1] + view_proj16[10] * p[2]) * 4096.0f;
        out[i] = rust_f32_as_i32(d);
    }
}

/* ------------------------------------------------------------------------- */
/* Uniform blocks (byte layouts: renderer.rs:602-726, camera.rs:158-189,       */
/* gswt.wgsl:437-476)                                                          */
/* ------------------------------------------------------------------------- */
typedef struct {
    float projection[16];   /*   0 */
    float view[16];         /*  64 */
    float focal[2];         /* 128 */
    float viewport[2];      /* 136 */
    float htan_fov[4];      /* 144 */
    float cam_pos[4];       /* 160 */
} orc_camera;               /* 176 B */

typedef struct {
    float splat_scale;      /*  0 */
    float tile_width;       /*  4 */
    uint32_t use_clip;      /*  8 */
    float clip_height;      /* 12 */
    uint32_t surface_type;  /* 16 */
    float sphere_radius;    /* 20 */
    float point_cloud_radius; /* 24 */
    float transition_width_ratio; /* 28 */
    uint32_t num_lod;       /* 32 */
    uint32_t draw_mode;     /* 36 */
    uint32_t map_half_wh[2];/* 40 */
    int32_t center_coord[2];/* 48 */
    uint32_t _pad0[2];      /* 56 */
    float transition_dist[16]; /* 64 */
    float height_map_scale[4]; /* 128 */
    float scene_scale[4];   /* 144 */
} orc_scene;                /* 160 B */

typedef struct {
    uint32_t single_draw;   /*  0 */
    uint32_t map_index;     /*  4 */
    int32_t single_lod_id;  /*  8 */
    int32_t valid_lod_id;   /* 12 */
    uint32_t changing;      /* 16 */
    int32_t changing_to_lower; /* 20 */
    uint32_t _pad0[2];      /* 24 */
    uint32_t tile_id[4];    /* 32 */
    float offset[4];        /* 48 */
    uint32_t map_coord[4];  /* 64 */
} orc_tile;                 /* 80 B */

/* One draw call of GSWTRenderer::render (renderer.rs:466-590): tile uniforms +
 * the three instance-rate vertex buffers (gs_index, map_id, lod_id) and the
 * instance count.  map_id / lod_id may be NULL where the shader never reads them
 * (renderer.rs:558,576 bind stale buffers in those cases). */
typedef struct {
    orc_tile tile;
    const uint32_t *gs_index;
    const uint32_t *map_id;
    const uint32_t *lod_id;
    uint32_t count;
    uint32_t _pad;
} orc_draw;

/* Per-splat vertex-stage result (gswt.wgsl:27-422) */
typedef struct {
    int32_t visible;     /* 0 = discarded / clipped */
    float ndc[2];        /* vCenter.xy */
    float depth;         /* vCenter.z  */
    float major[2];      /* majorAxis  */
    float minor[2];      /* minorAxis  */
    float rgba[4];       /* v_color    */
} orc_splat;

static inline float clampf(float e, float lo, float hi) { return fminf(fmaxf(e, lo), hi); }

/* WebGPU textureSampleLevel on an R32Float texture, FilterMode::Linear,
 * AddressMode::Repeat, level 0 (renderer.rs:376-388).  Texel centres at +0.5. */
static float sample_height(const float *hm, int w, int h, float u, float v)
{
    float x = u * (float)w - 0.5f;
    float y = v * (float)h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float tx = x - fx0, ty = y - fy0;
    long x0 = (long)fx0, y0 = (long)fy0;
    long xa = ((x0 % w) + w) % w, xb = (((x0 + 1) % w) + w) % w;
    long ya = ((y0 % h) + h) % h, yb = (((y0 + 1) % h) + h) % h;
    float i00 = hm[ya * w + xa], i10 = hm[ya * w + xb];
    float i01 = hm[yb * w + xa], i11 = hm[yb * w + xb];
    float i0 = i00 * (1.0f - tx) + i10 * tx;
    float i1 = i01 * (1.0f - tx) + i11 * tx;
    return i0 * (1.0f - ty) + i1 * ty;
}

/* surface_mapping, gswt.wgsl:565-624, HeightMap branch (surface_type == 1).
 * transform is column-major 3x3: cols local_x, local_y, local_z. */
static void surface_mapping_hmap(const orc_scene *s, const float *hm, int hw, int hh,
                                 float px, float py, float new_pos[3], float tr[9])
{
    const float DELTA = 0.001f;
    float xr = (2.0f * (float)s->map_half_wh[0] + 1.0f) * s->tile_width * s->height_map_scale[0];
    float yr = (2.0f * (float)s->map_half_wh[1] + 1.0f) * s->tile_width * s->height_map_scale[1];
    float h_u = (px + (float)s->map_half_wh[0] * s->tile_width) / xr;
    float h_v = (py + (float)s->map_half_wh[1] * s->tile_width) / yr;
    float hz = s->height_map_scale[2];
    new_pos[0] = px; new_pos[1] = py;
    new_pos[2] = sample_height(hm, hw, hh, h_u, h_v) * hz;
    float dt = DELTA;
    float h_r = sample_height(hm, hw, hh, h_u + dt, h_v) * hz;
    float h_l = sample_height(hm, hw, hh, h_u - dt, h_v) * hz;
    float h_up = sample_height(hm, hw, hh, h_u, h_v + dt) * hz;
    float h_d = sample_height(hm, hw, hh, h_u, h_v - dt) * hz;
    float lx[3] = { 1.0f, 0.0f, (h_r - h_l) / (2.0f * dt * xr) };
    float ly[3] = { 0.0f, 1.0f, (h_up - h_d) / (2.0f * dt * yr) };
    /* cross(lx, ly) */
    float cz[3] = { lx[1] * ly[2] - lx[2] * ly[1],
                    lx[2] * ly[0] - lx[0] * ly[2],
                    lx[0] * ly[1] - lx[1] * ly[0] };
    float len = sqrtf((cz[0] * cz[0] + cz[1] * cz[1]) + cz[2] * cz[2]);
    tr[0] = lx[0]; tr[1] = lx[1]; tr[2] = lx[2];
    tr[3] = ly[0]; tr[4] = ly[1]; tr[5] = ly[2];
    tr[6] = cz[0] / len; tr[7] = cz[1] / len; tr[8] = cz[2] / len;
}

/* Canonical sin / cos.  WGSL leaves the accuracy of sin()/cos() to the implementation, so the
 * reference's own values are unpinnable; what must hold here is that the CPU oracle and the HIP
 * kernels produce the SAME bits (the sphere mapping feeds discontinuous decisions, and the debug
 * colour hash rand() amplifies one ulp of sin by 43758).  Both sides therefore evaluate this
 * sequence: k = rint(x * 2/pi); three-term Cody-Waite reduction with fmaf; Cephes sinf/cosf
 * minimax polynomials on [-pi/4, pi/4]; quadrant fix-up.  |x| < ~1e5. */
ORC_API void orc_sincosf(float x, float *sn, float *cs)
{
    float kf = rintf(x * 0.636619772367581343f);
    float r = fmaf(kf, -1.5703125f, x);
    r = fmaf(kf, -4.837512969970703125e-4f, r);
    r = fmaf(kf, -7.54978995489188216e-8f, r);
    float z = r * r;
    float ps = fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = fmaf(ps, z, -1.6666654611e-1f);
    float s = fmaf(ps * z, r, r);
    float pc = fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = fmaf(pc, z, 4.166664568298827e-2f);
    float c = fmaf(pc * z, z, fmaf(-0.5f, z, 1.0f));
    int q = (int)kf & 3;
    float so = (q & 1) ? c : s, co = (q & 1) ? s : c;
    if (q == 2 || q == 3) so = -so;
    if (q == 1 || q == 2) co = -co;
    *sn = so; *cs = co;
}

/* sphere_get_uv, gswt.wgsl:515-555 */
static void sphere_get_uv(const orc_scene *s, float bidx, float bidy, float bx, float by, float uv[2])
{
    const float PI = 3.1415926535897932384626433832795f;
    float xmax = ((float)s->map_half_wh[0] * 2.0f) * s->tile_width;
    float block_w = xmax / 5.0f;
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
    uv[0] = u; uv[1] = v;
}

/* sphere_uv_to_pos, gswt.wgsl:558-564 */
static void sphere_uv_to_pos(const float uv[2], float p[3])
{
    float su, cu, sv, cv;
    orc_sincosf(uv[0], &su, &cu);
    orc_sincosf(uv[1], &sv, &cv);
    p[0] = cv * cu; p[1] = cv * su; p[2] = sv;
}

/* surface_mapping, gswt.wgsl:565-624, Sphere branch (surface_type == 2). */
static void surface_mapping_sphere(const orc_scene *s, const orc_tile *u, uint32_t map_id,
                                   float px, float py, float new_pos[3], float tr[9])
{
    const float DELTA = 0.001f;
    float xmax = ((float)s->map_half_wh[0] * 2.0f) * s->tile_width;
    float ymax = ((float)s->map_half_wh[1] * 2.0f) * s->tile_width;
    float block_w = xmax / 5.0f;
    float nx = px - (float)(s->center_coord[0] - (int32_t)s->map_half_wh[0]) * s->tile_width;
    float ny = py - (float)(s->center_coord[1] - (int32_t)s->map_half_wh[1]) * s->tile_width;
    float bidx = (float)(5u * u->map_coord[0] / (s->map_half_wh[0] * 2u));
    float bidy = (float)(2u * u->map_coord[1] / (s->map_half_wh[1] * 2u));
    if (u->single_draw == 1u) {
        uint32_t map_height = 2u * s->map_half_wh[1];
        uint32_t mcx = map_id / map_height, mcy = map_id % map_height;
        bidx = (float)(5u * mcx / (s->map_half_wh[0] * 2u));
        bidy = (float)(2u * mcy / (s->map_half_wh[1] * 2u));
    }
    float bx = nx - bidx * block_w;
    float by = ny - bidy * block_w;
    float uv[2], lz[3];
    sphere_get_uv(s, bidx, bidy, bx, by, uv);
    sphere_uv_to_pos(uv, lz);
    float R = s->sphere_radius;
    for (int k = 0; k < 3; k++) new_pos[k] = lz[k] * R;
    float dt = DELTA * ymax;
    float pr[3], pl[3], pu[3], pd[3];
    sphere_get_uv(s, bidx, bidy, bx + dt, by, uv); sphere_uv_to_pos(uv, pr);
    sphere_get_uv(s, bidx, bidy, bx - dt, by, uv); sphere_uv_to_pos(uv, pl);
    sphere_get_uv(s, bidx, bidy, bx, by + dt, uv); sphere_uv_to_pos(uv, pu);
    sphere_get_uv(s, bidx, bidy, bx, by - dt, uv); sphere_uv_to_pos(uv, pd);
    for (int k = 0; k < 3; k++) {
        tr[k] = (pr[k] * R - pl[k] * R) / (2.0f * dt);
        tr[3 + k] = (pu[k] * R - pd[k] * R) / (2.0f * dt);
        tr[6 + k] = lz[k];
    }
}

/* rand / randomVec3, gswt.wgsl:502-512 (debug draw mode 1) */
static float dbg_rand(float cx, float cy)
{
    float d = cx * 12.9898f + cy * 78.233f;
    float sn, cs;
    orc_sincosf(d, &sn, &cs);
    float v = sn * 43758.5453f;
    return v - floorf(v);
}

/* Debug draw recolouring, gswt.wgsl:268-399 (draw_mode 1..4).  rgb in / out, alpha untouched. */
static void debug_draw_color(const orc_scene *s, const orc_tile *u, const float vpos[3], uint32_t lod_id,
                             float t_ratio, float rgb[3])
{
    const float tw = s->tile_width;
    switch (s->draw_mode) {
    case 1u: {
        float g = clampf(((rgb[0] + rgb[1]) + rgb[2]) / 0.6f, 0.0f, 1.0f);
        rgb[0] = rgb[1] = rgb[2] = g;
        float margin = 0.05f * tw;
        const int sphere = s->surface_type == 2u;
        if (u->single_draw == 1u) {
            float ox = u->offset[0], oy = u->offset[1];
            rgb[0] = rgb[0] * dbg_rand(ox, oy);
            rgb[1] = rgb[1] * dbg_rand(ox + 23.45f, oy + 23.45f);
            rgb[2] = rgb[2] * dbg_rand(ox + 67.89f, oy + 67.89f);
        } else if (vpos[0] < margin) {
            if (vpos[1] < margin || vpos[1] > tw - margin) { rgb[0] = rgb[1] = rgb[2] = 0.5f; }
            else if (u->tile_id[1] / 8u % 2u == 0u) { rgb[0] = 1.0f; rgb[1] = 0.0f; rgb[2] = 0.0f; }
            else { rgb[0] = 0.0f; rgb[1] = 1.0f; rgb[2] = 0.13f; }
        } else if (vpos[0] > tw - margin) {
            if (vpos[1] < margin || vpos[1] > tw - margin) { rgb[0] = rgb[1] = rgb[2] = 0.5f; }
            else if (u->tile_id[1] / 2u % 2u == 0u) { rgb[0] = 1.0f; rgb[1] = 0.0f; rgb[2] = 0.0f; }
            else { rgb[0] = 0.0f; rgb[1] = 1.0f; rgb[2] = 0.13f; }
        } else if (vpos[1] < margin || vpos[1] > tw - margin) {
            const uint32_t bit = vpos[1] < margin ? u->tile_id[1] % 2u : u->tile_id[1] / 4u % 2u;
            if (bit == 0u) {
                if (sphere) { rgb[0] = 1.0f; rgb[1] = 0.0f; rgb[2] = 0.0f; }
                else { rgb[0] = 1.0f; rgb[1] = 0.85f; rgb[2] = 0.0f; }
            } else {
                if (sphere) { rgb[0] = 0.0f; rgb[1] = 1.0f; rgb[2] = 0.13f; }
                else { rgb[0] = 0.0f; rgb[1] = 0.58f; rgb[2] = 1.0f; }
            }
        }
        break;
    }
    case 2u:
    case 3u: {
        if (t_ratio > 0.0f && t_ratio < 1.0f) { rgb[0] = rgb[1] = rgb[2] = 0.0f; break; }
        if (s->draw_mode == 2u && u->changing == 1u) { rgb[0] = 0.0f; rgb[1] = 1.0f; rgb[2] = 0.0f; break; }
        uint32_t L = u->tile_id[0];
        if (s->draw_mode == 3u) L = u->single_lod_id >= 0 ? (uint32_t)u->single_lod_id : lod_id;
        float cx = 0.0f, cy = 1.0f;
        if (L < 3u) cx = (3.0f - (float)L) / 3.0f;
        else cy = (6.0f - (float)L) / 3.0f;
        rgb[0] = 0.5f; rgb[1] = cx; rgb[2] = cy;
        break;
    }
    case 4u: {
        uint32_t v = u->tile_id[2];
        float cx = 0.0f, cy = 0.0f;
        if (v < 4u) cx = (4.0f - (float)v) / 4.0f;
        if (v >= 4u) cy = (8.0f - (float)v) / 4.0f;
        if (v >= 8u) { cx = 1.0f; cy = 1.0f; }
        rgb[0] = 0.5f; rgb[1] = cx; rgb[2] = cy;
        break;
    }
    default: break;
    }
}

/* ------------------------------------------------------------------------- */
/* STRICT mode: the shader text, operator by operator.                         */
/*                                                                             */
/* The default sequence ("v2", what the HIP kernels reproduce bit for bit)      */
/* evaluates A6..A10 and F1/F2 with fma chains and ONE reciprocal per quotient. */
/* WGSL permits that, but it is a CHOSEN rounding sequence.  With strict mode   */
/* on, orc_project evaluates gswt.wgsl:152-258,402-419 as written -- every `*`, */
/* `+`, `-`, `/` its own correctly rounded binary32 operation, matrix * vector  */
/* as the left-to-right sum of column products, length() = sqrt(x*x + y*y),     */
/* normalize() = v / length(v), no fused multiply-add anywhere -- and the       */
/* fragment stage takes v_position at a pixel centre from the exact (double     */
/* precision) affine inverse of the quad (the hardware interpolator is not in   */
/* the shader text), rounds it to binary32 and evaluates fs_main                */
/* (gswt.wgsl:425-435) as written: A = -dot(p, p); discard if A < -4;           */
/* B = exp(A) * alpha.  The strict image is the anchor the v2 image and the GPU */
/* image are both measured against (tests/test_strict_oracle*.py).             */
/* Process-wide switch; set it before orc_render / orc_project_draws.          */
/* ------------------------------------------------------------------------- */
/* 2 (DEFAULT since round 4) = strict vertex stage + the fragment sequence F1..F4: what the HIP path computes by default (its      */
/* compositor always evaluates F1..F4); 1 = strict vertex AND fragment stage (the anchor image: exact quad interpolation);        */
/* 0 = the rounding sequence v2 everywhere (fma chains, one reciprocal per quotient: the HIP path with GSWT_OPT_STRICT_VS = 0).   */
static int g_strict = 2;
ORC_API void orc_set_strict(int on) { g_strict = on == 2 ? 2 : (on ? 1 : 0); }
ORC_API int orc_get_strict(void) { return g_strict; }

/* vs_main, gswt.wgsl:27-422, every draw_mode.  Canonical float sequence "A1..A10"
 * of DESIGN.md.  Returns out->visible. */
static int project_impl(const orc_camera *cam, const orc_scene *s, const orc_tile *u,
                        const uint32_t *tex, uint32_t gs_index, uint32_t map_id, uint32_t lod_id,
                        const float *hmap, int hm_w, int hm_h, orc_splat *out, int strict_mode)
{
    memset(out, 0, sizeof(*out));
    /* A1 :38-42 */
    if (u->valid_lod_id >= 0 && u->valid_lod_id != (int32_t)lod_id) return 0;
    /* A2 :45-49  texel (u,v) == linear u32[8*i .. 8*i+8] */
    const uint32_t *rec = tex + 8 * (size_t)gs_index;
    float pos[3] = { u2f(rec[0]), u2f(rec[1]), u2f(rec[2]) };
    /* A3 :52-65 */
    float off[3] = { u->offset[0], u->offset[1], u->offset[2] };
    uint32_t map_wh_y = 2u * s->map_half_wh[1];
    if (s->surface_type != 2u) map_wh_y += 1u;
    if (u->single_draw == 1u) {
        off[0] = (float)((int32_t)(map_id / map_wh_y - s->map_half_wh[0]) + s->center_coord[0]) * s->tile_width;
        off[1] = (float)((int32_t)(map_id % map_wh_y - s->map_half_wh[1]) + s->center_coord[1]) * s->tile_width;
        off[2] = 0.0f;
    }
    float c[3];
    for (int k = 0; k < 3; k++) c[k] = (pos[k] + off[k]) * s->scene_scale[k];
    /* A4 :75-87 */
    float mapped_z = 0.0f;
    float F[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 };
    if (s->surface_type == 1u) {
        float np[3];
        surface_mapping_hmap(s, hmap, hm_w, hm_h, c[0], c[1], np, F);
        /* center = mapped + transform * (0,0,z): col2 * z (the two zero terms add +-0) */
        float z = c[2];
        c[0] = np[0] + F[6] * z;
        c[1] = np[1] + F[7] * z;
        c[2] = np[2] + F[8] * z;
        mapped_z = np[2];
    } else if (s->surface_type == 2u) {
        float np[3];
        surface_mapping_sphere(s, u, map_id, c[0], c[1], np, F);
        float z = c[2];
        c[0] = np[0] + F[6] * z;
        c[1] = np[1] + F[7] * z;
        c[2] = np[2] + F[8] * z;
        mapped_z = np[2];
    }
    if (s->use_clip == 1u && mapped_z < s->clip_height) return 0;
    /* A5 :91-150 */
    float t_ratio = -1.0f;
    uint32_t higher_lod = 0u;
    if (u->changing == 1u) {
        float dx = c[0] - cam->cam_pos[0], dy = c[1] - cam->cam_pos[1], dz = c[2] - cam->cam_pos[2];
        float cam_dist = sqrtf((dx * dx + dy * dy) + dz * dz);
        if (u->single_draw == 1u) {
            if (lod_id == 0u) higher_lod = 0u;
            else if (lod_id == s->num_lod - 1u) higher_lod = lod_id - 1u;
            else {
                float d1 = s->transition_dist[lod_id - 1u];
                float d2 = s->transition_dist[lod_id];
                higher_lod = (cam_dist - d1 < d2 - cam_dist) ? lod_id - 1u : lod_id;
            }
        } else {
            higher_lod = (u->changing_to_lower == 1) ? u->tile_id[0] : u->tile_id[0] - 1u;
        }
        float td = s->transition_dist[higher_lod & 15u];
        float thw = s->transition_width_ratio * td;
        t_ratio = clampf((cam_dist - td) / thw + 0.5f, 0.0f, 1.0f);
        if ((lod_id == higher_lod + 1u && t_ratio == 0.0f) || (lod_id == higher_lod && t_ratio == 1.0f))
            return 0;
    }
    if (strict_mode) {
        /* A6..A10 as the shader writes them (gswt.wgsl:152-258, 260-265, 402-419) */
        const float *V = cam->view, *P = cam->projection;
        float cv[4];                        /* cam = view * vec4(center, 1.0) */
        for (int r = 0; r < 4; r++) cv[r] = ((V[r] * c[0] + V[4 + r] * c[1]) + V[8 + r] * c[2]) + V[12 + r] * 1.0f;
        /* opengl_to_wgpu * projection (left-associative): rows 0, 1, 3 of P unchanged (x*1 + 0*.. exact), row 2 = 0.5 P2 + 0.5 P3 */
        float GP[16];
        for (int cc = 0; cc < 4; cc++) {
            GP[4 * cc + 0] = P[4 * cc + 0];
            GP[4 * cc + 1] = P[4 * cc + 1];
            GP[4 * cc + 2] = 0.5f * P[4 * cc + 2] + 0.5f * P[4 * cc + 3];
            GP[4 * cc + 3] = P[4 * cc + 3];
        }
        float q[4];
        for (int r = 0; r < 4; r++)
            q[r] = ((GP[r] * cv[0] + GP[4 + r] * cv[1]) + GP[8 + r] * cv[2]) + GP[12 + r] * cv[3];
        float clip = 1.2f * q[3];
        if (q[2] < -clip || q[0] < -clip || q[0] > clip || q[1] < -clip || q[1] > clip) return 0;
        float a = orc_half_to_float(rec[4] & 0xFFFFu), b = orc_half_to_float(rec[4] >> 16);
        float c2 = orc_half_to_float(rec[5] & 0xFFFFu), d = orc_half_to_float(rec[5] >> 16);
        float e = orc_half_to_float(rec[6] & 0xFFFFu), f = orc_half_to_float(rec[6] >> 16);
        float K[9] = { a, b, c2, b, d, e, c2, e, f };
        if (s->point_cloud_radius > 0.0f) {
            float pr = s->point_cloud_radius;
            if (s->draw_mode > 0u) pr *= ldexpf(1.0f, (int)u->tile_id[0]);
            K[0] = pr; K[1] = 0; K[2] = 0; K[3] = 0; K[4] = pr; K[5] = 0; K[6] = 0; K[7] = 0; K[8] = pr;
        }
        if (s->surface_type > 0u) {         /* Vrk = transform * Vrk * transpose(transform) */
            float FK[9], R[9];
            for (int cc = 0; cc < 3; cc++)
                for (int r = 0; r < 3; r++)
                    FK[3 * cc + r] = (F[r] * K[3 * cc] + F[3 + r] * K[3 * cc + 1]) + F[6 + r] * K[3 * cc + 2];
            for (int cc = 0; cc < 3; cc++)
                for (int r = 0; r < 3; r++)
                    R[3 * cc + r] = (FK[r] * F[cc] + FK[3 + r] * F[3 + cc]) + FK[6 + r] * F[6 + cc];
            memcpy(K, R, sizeof(K));
        }
        {                                   /* scene_scale_mat * Vrk * transpose(scene_scale_mat), full matrix products */
            const float S[9] = { s->scene_scale[0], 0, 0, 0, s->scene_scale[1], 0, 0, 0, s->scene_scale[2] };
            float SK[9], R[9];
            for (int cc = 0; cc < 3; cc++)
                for (int r = 0; r < 3; r++)
                    SK[3 * cc + r] = (S[r] * K[3 * cc] + S[3 + r] * K[3 * cc + 1]) + S[6 + r] * K[3 * cc + 2];
            for (int cc = 0; cc < 3; cc++)
                for (int r = 0; r < 3; r++)
                    R[3 * cc + r] = (SK[r] * S[cc] + SK[3 + r] * S[3 + cc]) + SK[6 + r] * S[6 + cc];
            memcpy(K, R, sizeof(K));
        }
        float dd[3] = { c[0] - cam->cam_pos[0], c[1] - cam->cam_pos[1], c[2] - cam->cam_pos[2] };
        float t[3];
        for (int r = 0; r < 3; r++) t[r] = (V[r] * dd[0] + V[4 + r] * dd[1]) + V[8 + r] * dd[2];
        float txtz = t[0] / t[2], tytz = t[1] / t[2];
        float limx = 1.3f * cam->htan_fov[0], limy = 1.3f * cam->htan_fov[1];
        t[0] = clampf(txtz, -limx, limx) * t[2];
        t[1] = clampf(tytz, -limy, limy) * t[2];
        float tz2 = t[2] * t[2];
        /* J_T columns: (fx/tz, 0, -fx*tx/tz2), (0, fy/tz, -fy*ty/tz2), (0, 0, 0) */
        float JT[9] = { cam->focal[0] / t[2], 0.0f, (-cam->focal[0] * t[0]) / tz2,
                        0.0f, cam->focal[1] / t[2], (-cam->focal[1] * t[1]) / tz2,
                        0.0f, 0.0f, 0.0f };
        /* T = transpose(view3) * J_T: T[c][r] = sum_k view3T[k][r] * JT[c][k], view3T[k][r] = V[4 r + k] */
        float Tm[9];
        for (int cc = 0; cc < 3; cc++)
            for (int r = 0; r < 3; r++)
                Tm[3 * cc + r] = (V[4 * r + 0] * JT[3 * cc] + V[4 * r + 1] * JT[3 * cc + 1]) + V[4 * r + 2] * JT[3 * cc + 2];
        /* cov2d = transpose(T) * Vrk * T: first A = transpose(T) * Vrk, A[c][r] = sum_k Tt[k][r] * K[c][k], Tt[k][r] = Tm[3 r + k] */
        float Am[9], C2[9];
        for (int cc = 0; cc < 3; cc++)
            for (int r = 0; r < 3; r++)
                Am[3 * cc + r] = (Tm[3 * r + 0] * K[3 * cc] + Tm[3 * r + 1] * K[3 * cc + 1]) + Tm[3 * r + 2] * K[3 * cc + 2];
        for (int cc = 0; cc < 3; cc++)
            for (int r = 0; r < 3; r++)
                C2[3 * cc + r] = (Am[r] * Tm[3 * cc] + Am[3 + r] * Tm[3 * cc + 1]) + Am[6 + r] * Tm[3 * cc + 2];
        float c00 = C2[0], c01 = C2[1], c11 = C2[4];       /* cov2d[0][0], cov2d[0][1], cov2d[1][1] */
        float mid = 0.5f * (c00 + c11);
        float hx = 0.5f * (c00 - c11);
        float radius = sqrtf(hx * hx + c01 * c01);          /* length(vec2) */
        float l1 = mid + radius, l2 = mid - radius;
        if (l2 < 0.0f) return 0;
        float vx = c01, vy = l1 - c00;
        float vlen = sqrtf(vx * vx + vy * vy);
        float ex = vx / vlen, ey = vy / vlen;               /* normalize(): 0/0 -> NaN -> nothing drawn */
        float smaj = fminf(sqrtf(2.0f * l1), 1024.0f);
        float smin = fminf(sqrtf(2.0f * l2), 1024.0f);
        out->major[0] = smaj * ex; out->major[1] = smaj * ey;
        out->minor[0] = smin * ey; out->minor[1] = smin * -ex;
        uint32_t cw = rec[7];
        out->rgba[0] = (float)(cw & 0xFFu) / 255.0f;
        out->rgba[1] = (float)((cw >> 8) & 0xFFu) / 255.0f;
        out->rgba[2] = (float)((cw >> 16) & 0xFFu) / 255.0f;
        out->rgba[3] = (float)((cw >> 24) & 0xFFu) / 255.0f;
        if (s->draw_mode != 0u) debug_draw_color(s, u, pos, lod_id, t_ratio, out->rgba);
        if (u->changing == 1u) {
            if (lod_id != higher_lod) out->rgba[3] = out->rgba[3] * t_ratio;
            else out->rgba[3] = out->rgba[3] * (1.0f - t_ratio);
        }
        float fade = clampf(q[2] / q[3] + 1.0f, 0.0f, 1.0f);
        for (int k = 0; k < 4; k++) out->rgba[k] = out->rgba[k] * fade;
        out->ndc[0] = q[0] / q[3];
        out->ndc[1] = q[1] / q[3];
        out->depth = q[2] / q[3];
        if (!(out->depth >= 0.0f && out->depth <= 1.0f)) return 0;
        out->visible = 1;
        return 1;
    }
    /* A6 :152-167.  pos2d = (opengl_to_wgpu * projection) * (view * center).
     * Canonical sequence v2 (round 2): dot products are fma chains (first product rounded, then one fma per further
     * term, the translation column last), quotients are products with ONE correctly rounded reciprocal (1/t.z, 1/q.w,
     * 1/|v|, 1/(s*|major|), 1/(s*|minor|)) -- WGSL allows both (fused multiply-add; division to 2.5 ULP).  What
     * matters is that this file and k_project evaluate the same IEEE binary32 operations in the same order. */
    const float *V = cam->view, *P = cam->projection;
    float cv[4];
    for (int r = 0; r < 4; r++) cv[r] = fmaf(V[8 + r], c[2], fmaf(V[4 + r], c[1], V[r] * c[0])) + V[12 + r];
    float GP[16];
    for (int cc = 0; cc < 4; cc++) {
        GP[4 * cc + 0] = P[4 * cc + 0];
        GP[4 * cc + 1] = P[4 * cc + 1];
        GP[4 * cc + 2] = 0.5f * P[4 * cc + 2] + 0.5f * P[4 * cc + 3];
        GP[4 * cc + 3] = P[4 * cc + 3];
    }
    float q[4];
    for (int r = 0; r < 4; r++)
        q[r] = fmaf(GP[12 + r], cv[3], fmaf(GP[8 + r], cv[2], fmaf(GP[4 + r], cv[1], GP[r] * cv[0])));
    float clip = 1.2f * q[3];
    if (q[2] < -clip || q[0] < -clip || q[0] > clip || q[1] < -clip || q[1] > clip) return 0;
    /* A7 :169-205 */
    float a = orc_half_to_float(rec[4] & 0xFFFFu), b = orc_half_to_float(rec[4] >> 16);
    float c2 = orc_half_to_float(rec[5] & 0xFFFFu), d = orc_half_to_float(rec[5] >> 16);
    float e = orc_half_to_float(rec[6] & 0xFFFFu), f = orc_half_to_float(rec[6] >> 16);
    /* Vrk column-major symmetric */
    float K[9] = { a, b, c2, b, d, e, c2, e, f };
    if (s->point_cloud_radius > 0.0f) {
        float pr = s->point_cloud_radius;
        if (s->draw_mode > 0u) pr *= ldexpf(1.0f, (int)u->tile_id[0]);
        K[0] = pr; K[1] = 0; K[2] = 0; K[3] = 0; K[4] = pr; K[5] = 0; K[6] = 0; K[7] = 0; K[8] = pr;
    }
    if (s->surface_type > 0u) {
        /* Vrk = F * Vrk * F^T : (F*K)[c][r] = sum_k F[k][r]*K[c][k]; then * F^T */
        float FK[9], R[9];
        for (int cc = 0; cc < 3; cc++)
            for (int r = 0; r < 3; r++)
                FK[3 * cc + r] = fmaf(F[6 + r], K[3 * cc + 2], fmaf(F[3 + r], K[3 * cc + 1], F[r] * K[3 * cc]));
        /* (FK * F^T)[c][r] = sum_k FK[k][r] * F^T[c][k] = sum_k FK[k][r] * F[k][c] */
        for (int cc = 0; cc < 3; cc++)
            for (int r = 0; r < 3; r++)
                R[3 * cc + r] = fmaf(FK[6 + r], F[6 + cc], fmaf(FK[3 + r], F[3 + cc], FK[r] * F[cc]));
        memcpy(K, R, sizeof(K));
    }
    for (int cc = 0; cc < 3; cc++)          /* S * Vrk * S^T, S = diag(scene_scale) */
        for (int r = 0; r < 3; r++)
            K[3 * cc + r] = (s->scene_scale[r] * K[3 * cc + r]) * s->scene_scale[cc];
    /* A8 :207-258 */
    float dd[3] = { c[0] - cam->cam_pos[0], c[1] - cam->cam_pos[1], c[2] - cam->cam_pos[2] };
    float t[3];
    for (int r = 0; r < 3; r++) t[r] = fmaf(V[8 + r], dd[2], fmaf(V[4 + r], dd[1], V[r] * dd[0]));
    const float rz = 1.0f / t[2];
    float txtz = t[0] * rz, tytz = t[1] * rz;
    float limx = 1.3f * cam->htan_fov[0], limy = 1.3f * cam->htan_fov[1];
    t[0] = clampf(txtz, -limx, limx) * t[2];
    t[1] = clampf(tytz, -limy, limy) * t[2];
    const float rz2 = rz * rz;
    float j00 = cam->focal[0] * rz, j02 = -((cam->focal[0] * t[0]) * rz2);
    float j11 = cam->focal[1] * rz, j12 = -((cam->focal[1] * t[1]) * rz2);
    /* T = transpose(view3) * J_T ; T col0 = (dot(view3 col r, J col0))_r etc. */
    float T0[3], T1[3];
    for (int r = 0; r < 3; r++) {
        T0[r] = fmaf(V[4 * r + 2], j02, V[4 * r + 0] * j00);
        T1[r] = fmaf(V[4 * r + 2], j12, V[4 * r + 1] * j11);
    }
    /* A = T^T * Vrk : A[k][r] = dot(T_r, Vrk col k) ; cov2d[c][r] = sum_k A[k][r]*T_c[k] */
    float A0[3], A1[3];
    for (int k = 0; k < 3; k++) {
        A0[k] = fmaf(T0[2], K[3 * k + 2], fmaf(T0[1], K[3 * k + 1], T0[0] * K[3 * k]));
        A1[k] = fmaf(T1[2], K[3 * k + 2], fmaf(T1[1], K[3 * k + 1], T1[0] * K[3 * k]));
    }
    float c00 = fmaf(A0[2], T0[2], fmaf(A0[1], T0[1], A0[0] * T0[0]));
    float c01 = fmaf(A1[2], T0[2], fmaf(A1[1], T0[1], A1[0] * T0[0]));   /* cov2d[0][1] */
    float c11 = fmaf(A1[2], T1[2], fmaf(A1[1], T1[1], A1[0] * T1[0]));
    float mid = 0.5f * (c00 + c11);
    float hx = 0.5f * (c00 - c11);
    float radius = sqrtf(fmaf(hx, hx, c01 * c01));
    float l1 = mid + radius, l2 = mid - radius;
    if (l2 < 0.0f) return 0;
    float vx = c01, vy = l1 - c00;
    float vlen = sqrtf(fmaf(vx, vx, vy * vy));
    const float rv = 1.0f / vlen;
    float ex = vx * rv, ey = vy * rv;           /* normalize(): 0 * (1/0) -> NaN -> nothing drawn */
    float smaj = fminf(sqrtf(2.0f * l1), 1024.0f);
    float smin = fminf(sqrtf(2.0f * l2), 1024.0f);
    out->major[0] = smaj * ex; out->major[1] = smaj * ey;
    out->minor[0] = smin * ey; out->minor[1] = smin * -ex;
    /* A9 :260-265, 402-410 (byte / 255 as byte * fl(1/255)) */
    const float k255 = 1.0f / 255.0f;
    uint32_t cw = rec[7];
    out->rgba[0] = (float)(cw & 0xFFu) * k255;
    out->rgba[1] = (float)((cw >> 8) & 0xFFu) * k255;
    out->rgba[2] = (float)((cw >> 16) & 0xFFu) * k255;
    out->rgba[3] = (float)((cw >> 24) & 0xFFu) * k255;
    if (s->draw_mode != 0u) debug_draw_color(s, u, pos, lod_id, t_ratio, out->rgba);   /* :268-399 */
    if (u->changing == 1u) {
        if (lod_id != higher_lod) out->rgba[3] = out->rgba[3] * t_ratio;
        else out->rgba[3] = out->rgba[3] * (1.0f - t_ratio);
    }
    const float rq = 1.0f / q[3];
    float fade = clampf(fmaf(q[2], rq, 1.0f), 0.0f, 1.0f);
    for (int k = 0; k < 4; k++) out->rgba[k] = out->rgba[k] * fade;
    /* A10 :415-419.  Hardware clip keeps 0 <= z <= w with w_out = 1. */
    out->ndc[0] = q[0] * rq;
    out->ndc[1] = q[1] * rq;
    out->depth = q[2] * rq;
    if (!(out->depth >= 0.0f && out->depth <= 1.0f)) return 0;
    out->visible = 1;
    return 1;
}

ORC_API int orc_project(const orc_camera *cam, const orc_scene *s, const orc_tile *u,
                        const uint32_t *tex, uint32_t gs_index, uint32_t map_id, uint32_t lod_id,
                        const float *hmap, int hm_w, int hm_h, orc_splat *out)
{
    return project_impl(cam, s, u, tex, gs_index, map_id, lod_id, hmap, hm_w, hm_h, out, g_strict != 0);
}

/* ------------------------------------------------------------------------- */
/* Fragment stage (fs_main gswt.wgsl:425-435 + blend renderer.rs:118-129 +     */
/* depth state :179-185), restated analytically.   */
/*                                                                             */
/* Canonical sequence "F1..F4" (DESIGN.md).  The +-2 quad is an affine image of */
/* quad space, w_out = 1, so v_position at a pixel centre is the affine inverse */
/* evaluated there.  It is evaluated relative to the origin of the 16x16 pixel  */
/* block containing the pixel (well conditioned, and what the HIP compositor    */
/* does per screen tile).                                                       */
/* ------------------------------------------------------------------------- */
typedef struct {
    float cxp, cyp;       /* pixel-space centre (conservative box only) */
    float ndcx, ndcy;     /* the centre in NDC: F3 takes the block-local offset from it with ONE rounding */
    float iux, iuy;       /* quad-x row of the inverse affine map   */
    float ivx, ivy;       /* quad-y row                             */
    float hx, hy;         /* conservative half extents in pixels    */
    int ok;
} orc_frag_setup;

static void frag_setup(const orc_splat *sp, float splat_scale, float W, float H, orc_frag_setup *fs)
{
    /* F1: pixel-space centre */
    fs->cxp = fmaf(0.5f, sp->ndc[0], 0.5f) * W;
    fs->cyp = fmaf(-0.5f, sp->ndc[1], 0.5f) * H;
    fs->ndcx = sp->ndc[0]; fs->ndcy = sp->ndc[1];
    /* F2: pixel-space image of the unit quad axes (framebuffer y is down); the rows of the inverse affine map are
     * iu = u / |u|^2, iv = w / |w|^2, each with ONE reciprocal */
    float hs = 0.5f * splat_scale;
    float ux = hs * sp->major[0], uy = -(hs * sp->major[1]);
    float vx = hs * sp->minor[0], vy = -(hs * sp->minor[1]);
    float uu = fmaf(uy, uy, ux * ux);
    float vv = fmaf(vy, vy, vx * vx);
    fs->ok = (uu > 0.0f) && (vv > 0.0f) && (uu < INFINITY) && (vv < INFINITY);  /* false for NaN */
    if (!fs->ok) return;
    const float ruu = 1.0f / uu, rvv = 1.0f / vv;
    fs->iux = ux * ruu; fs->iuy = uy * ruu;
    fs->ivx = vx * rvv; fs->ivy = vy * rvv;
    /* extent of |p| <= 2 : |dX| <= 2 sqrt(ux^2 + vx^2), inflated by 1e-3 px + 1e-5 relative so the
     * box is conservative under f32 rounding of F3/F4.  A pixel can be covered only if its CENTRE
     * lies inside [c - h, c + h]. */
    fs->hx = fmaf(2.0f * sqrtf(fmaf(vx, vx, ux * ux)), 1.00001f, 0.001f);
    fs->hy = fmaf(2.0f * sqrtf(fmaf(vy, vy, uy * uy)), 1.00001f, 0.001f);
}

/* Rasterise one projected splat into rows [y_lo, y_hi) with "over" blending
 * dst = src + dst * (1 - src.a), back-to-front (renderer.rs:118-129). */
static void raster_over(const orc_splat *sp, const orc_frag_setup *fs, int W, int H,
                        int y_lo, int y_hi, const float *bg_depth, float *img)
{
    /* pixels whose centre x + 0.5 lies in [c - h, c + h] */
    float fx0 = ceilf(fs->cxp - fs->hx - 0.5f), fx1 = floorf(fs->cxp + fs->hx - 0.5f);
    float fy0 = ceilf(fs->cyp - fs->hy - 0.5f), fy1 = floorf(fs->cyp + fs->hy - 0.5f);
    if (!(fx1 >= fx0) || !(fy1 >= fy0)) return;
    if (!(fx1 >= 0.0f) || !(fy1 >= 0.0f) || !(fx0 <= (float)(W - 1)) || !(fy0 <= (float)(H - 1))) return;
    int x0 = fx0 < 0.0f ? 0 : (int)fx0, x1 = fx1 > (float)(W - 1) ? W - 1 : (int)fx1;
    int y0 = fy0 < (float)y_lo ? y_lo : (int)fy0, y1 = fy1 > (float)(y_hi - 1) ? y_hi - 1 : (int)fy1;
    if (y1 < y0) return;
    for (int by = y0 & ~15; by <= y1; by += 16) {
        for (int bx = x0 & ~15; bx <= x1; bx += 16) {
            /* F3: per-block constants.  Sequence v3 (round 4): the centre's offset from the block origin straight from NDC with ONE
             * rounding, o = fma(W/2, ndc.x, W/2 - bx) (W/2 - bx is exact), instead of (pixel-space centre) - bx: the pixel-space centre
             * carries two roundings at magnitude ~W (ulp 1.2e-4 px at x = 1900), which a thin ellipse (|iv| ~ 20 / px) turns into
             * ~2e-3 in p -- the continuous term that kept single pixels 1.2e-4 away from the strict image. */
            float ox = fmaf(0.5f * (float)W, fs->ndcx, 0.5f * (float)W - (float)bx);
            float oy = fmaf(-0.5f * (float)H, fs->ndcy, 0.5f * (float)H - (float)by);
            float nku = -fmaf(fs->iux, ox, fs->iuy * oy);
            float nkv = -fmaf(fs->ivx, ox, fs->ivy * oy);
            int ya = by < y0 ? y0 : by, yb = by + 15 > y1 ? y1 : by + 15;
            int xa = bx < x0 ? x0 : bx, xb = bx + 15 > x1 ? x1 : bx + 15;
            for (int y = ya; y <= yb; y++) {
                float ly = (float)(y - by) + 0.5f;
                float pu_y = fmaf(fs->iuy, ly, nku);
                float pv_y = fmaf(fs->ivy, ly, nkv);
                for (int x = xa; x <= xb; x++) {
                    /* F4: per-pixel */
                    float lx = (float)(x - bx) + 0.5f;
                    float px = fmaf(fs->iux, lx, pu_y);
                    float py = fmaf(fs->ivx, lx, pv_y);
                    float r2 = fmaf(py, py, px * px);
                    if (!(r2 <= 4.0f)) continue;                 /* discard if A < -4 */
                    /* depth_compare Less against the proxy depth, or the 1.0 clear value */
                    float dbuf = bg_depth ? bg_depth[(size_t)y * W + x] : 1.0f;
                    if (!(sp->depth < dbuf)) continue;
                    float Bv = expf(-r2) * sp->rgba[3];
                    float om = 1.0f - Bv;
                    float *dst = img + 4 * ((size_t)y * W + x);
                    dst[0] = Bv * sp->rgba[0] + dst[0] * om;
                    dst[1] = Bv * sp->rgba[1] + dst[1] * om;
                    dst[2] = Bv * sp->rgba[2] + dst[2] * om;
                    dst[3] = Bv + dst[3] * om;
                }
            }
        }
    }
}

/* STRICT mode fragment stage (see the mode's header above): per pixel centre the exact affine inverse of the +-2 quad in
 * double precision, rounded to binary32 = v_position; fs_main as written; "over" blend as above. */
static void raster_over_strict(const orc_splat *sp, float splat_scale, int W, int H, int y_lo, int y_hi,
                               const float *bg_depth, float *img)
{
    const double s = (double)splat_scale;
    const double Mx = sp->major[0], My = sp->major[1], Nx = sp->minor[0], Ny = sp->minor[1];
    const double mm = Mx * Mx + My * My, nn = Nx * Nx + Ny * Ny;
    if (!(mm > 0.0) || !(nn > 0.0) || !(mm < INFINITY) || !(nn < INFINITY)) return;
    /* pixel-space centre and the pixel-space images of the quad axes (framebuffer y is down) */
    const double cx = ((double)sp->ndc[0] * 0.5 + 0.5) * W, cy = (0.5 - (double)sp->ndc[1] * 0.5) * H;
    const double ux = 0.5 * s * Mx, uy = -0.5 * s * My, vx = 0.5 * s * Nx, vy = -0.5 * s * Ny;
    const double hx = 2.0 * sqrt(ux * ux + vx * vx) * 1.000001 + 1e-3, hy = 2.0 * sqrt(uy * uy + vy * vy) * 1.000001 + 1e-3;
    double fx0 = ceil(cx - hx - 0.5), fx1 = floor(cx + hx - 0.5), fy0 = ceil(cy - hy - 0.5), fy1 = floor(cy + hy - 0.5);
    if (!(fx1 >= fx0) || !(fy1 >= fy0) || !(fx1 >= 0.0) || !(fy1 >= 0.0) || !(fx0 <= W - 1) || !(fy0 <= H - 1)) return;
    int x0 = fx0 < 0.0 ? 0 : (int)fx0, x1 = fx1 > W - 1 ? W - 1 : (int)fx1;
    int y0 = fy0 < y_lo ? y_lo : (int)fy0, y1 = fy1 > y_hi - 1 ? y_hi - 1 : (int)fy1;
    for (int y = y0; y <= y1; y++) {
        /* ndc of the pixel centre; d = (ndc_px - vCenter.xy) * viewport / splat_scale = p.x major + p.y minor */
        const double ny = 1.0 - (y + 0.5) / H * 2.0;
        const double dy = (ny - (double)sp->ndc[1]) * H / s;
        for (int x = x0; x <= x1; x++) {
            const double nx = (x + 0.5) / W * 2.0 - 1.0;
            const double dx = (nx - (double)sp->ndc[0]) * W / s;
            const float px = (float)((dx * Mx + dy * My) / mm), py = (float)((dx * Nx + dy * Ny) / nn);
            const float A = -(px * px + py * py);                 /* -dot(v_position, v_position) */
            if (A < -4.0f || A != A) continue;                     /* discard (a NaN position draws nothing sensible either) */
            float dbuf = bg_depth ? bg_depth[(size_t)y * W + x] : 1.0f;
            if (!(sp->depth < dbuf)) continue;
            float Bv = expf(A) * sp->rgba[3];
            float om = 1.0f - Bv;
            float *dst = img + 4 * ((size_t)y * W + x);
            dst[0] = Bv * sp->rgba[0] + dst[0] * om;
            dst[1] = Bv * sp->rgba[1] + dst[1] * om;
            dst[2] = Bv * sp->rgba[2] + dst[2] * om;
            dst[3] = Bv + dst[3] * om;
        }
    }
}

/* Render statistics filled by orc_render */
typedef struct {
    uint64_t n_instanced;   /* sum of draw counts                */
    uint64_t n_visible;     /* survivors of the vertex stage     */
    uint64_t n_pairs16;     /* (splat, 16x16 block) pairs: blocks holding >= 1 pixel centre inside the
                               splat's axis-aligned bounding box of |p| <= 2 (frag_setup's hx, hy) */
} orc_stats;

/*
 * GSWTRenderer::render (renderer.rs:407-592) for an already culled draw list:
 * colour LoadOp::Load over `bg_rgba` (NULL = transparent black), depth buffer =
 * `bg_depth` (NULL = cleared to 1.0, which every visible splat passes because
 * depth <= 1 ... strictly: depth < 1.0 is required, renderer.rs:182), draws in
 * order, each draw's instances in order.
 *
 * order_mode 0 = REFERENCE (draw rank, list position); 1 = DEPTH (all visible
 * splats stably re-sorted by descending depth before blending).
 */
ORC_API int orc_render(const orc_camera *cam, const orc_scene *scene, const uint32_t *tex,
                       const orc_draw *draws, int n_draws,
                       const float *hmap, int hm_w, int hm_h,
                       int W, int H, const float *bg_rgba, const float *bg_depth,
                       int order_mode, int n_threads, float *out_rgba, orc_stats *stats)
{
    if (W <= 0 || H <= 0) return -1;
    if ((float)W != cam->viewport[0] || (float)H != cam->viewport[1]) return -2;
    uint64_t n_total = 0;
    uint64_t *prefix = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n_draws + 1));
    for (int d = 0; d < n_draws; d++) { prefix[d] = n_total; n_total += draws[d].count; }
    prefix[n_draws] = n_total;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    /* vertex stage for every instance, in draw order */
    orc_splat *sp = (orc_splat *)malloc(sizeof(orc_splat) * (size_t)(n_total ? n_total : 1));
    orc_frag_setup *fs = (orc_frag_setup *)malloc(sizeof(orc_frag_setup) * (size_t)(n_total ? n_total : 1));
    uint64_t n_vis = 0, n_pairs = 0;
    for (int d = 0; d < n_draws; d++) {
        const orc_draw *dr = &draws[d];
        long cnt = (long)dr->count;
        uint64_t vis_d = 0, pairs_d = 0;
#pragma omp parallel for schedule(static) reduction(+ : vis_d, pairs_d)
        for (long j = 0; j < cnt; j++) {
            uint64_t k = prefix[d] + (uint64_t)j;
            uint32_t mid = dr->map_id ? dr->map_id[j] : 0u;
            uint32_t lid = dr->lod_id ? dr->lod_id[j] : 0u;
            orc_project(cam, scene, &dr->tile, tex, dr->gs_index[j], mid, lid, hmap, hm_w, hm_h, &sp[k]);
            fs[k].ok = 0;
            if (sp[k].visible) {
                frag_setup(&sp[k], scene->splat_scale, (float)W, (float)H, &fs[k]);
                if (!fs[k].ok) sp[k].visible = 0;
            }
            if (sp[k].visible) {
                vis_d++;
                float fx0 = ceilf(fs[k].cxp - fs[k].hx - 0.5f), fx1 = floorf(fs[k].cxp + fs[k].hx - 0.5f);
                float fy0 = ceilf(fs[k].cyp - fs[k].hy - 0.5f), fy1 = floorf(fs[k].cyp + fs[k].hy - 0.5f);
                if (fx1 >= fx0 && fy1 >= fy0 && fx1 >= 0.0f && fy1 >= 0.0f && fx0 <= (float)(W - 1) && fy0 <= (float)(H - 1)) {
                    int x0 = fx0 < 0 ? 0 : (int)fx0, x1 = fx1 > (float)(W - 1) ? W - 1 : (int)fx1;
                    int y0 = fy0 < 0 ? 0 : (int)fy0, y1 = fy1 > (float)(H - 1) ? H - 1 : (int)fy1;
                    pairs_d += (uint64_t)((x1 >> 4) - (x0 >> 4) + 1) * (uint64_t)((y1 >> 4) - (y0 >> 4) + 1);
                }
            }
        }
        n_vis += vis_d; n_pairs += pairs_d;
    }
    /* composite order */
    uint64_t *order = NULL;
    uint64_t n_order = 0;
    if (order_mode == 1) {
        /* stable sort by descending depth (back-to-front); ties keep draw order */
        order = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n_vis ? n_vis : 1));
        for (uint64_t k = 0; k < n_total; k++) if (sp[k].visible) order[n_order++] = k;
        /* bottom-up merge sort, stable */
        uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n_order ? n_order : 1));
        for (uint64_t width = 1; width < n_order; width *= 2) {
            for (uint64_t lo = 0; lo < n_order; lo += 2 * width) {
                uint64_t mid = lo + width < n_order ? lo + width : n_order;
                uint64_t hi = lo + 2 * width < n_order ? lo + 2 * width : n_order;
                uint64_t i = lo, j = mid, o = lo;
                while (i < mid && j < hi) {
                    if (sp[order[j]].depth > sp[order[i]].depth) tmp[o++] = order[j++];
                    else tmp[o++] = order[i++];
                }
                while (i < mid) tmp[o++] = order[i++];
                while (j < hi) tmp[o++] = order[j++];
            }
            uint64_t *sw = order; order = tmp; tmp = sw;
        }
        free(tmp);
    }
    /* framebuffer: LoadOp::Load */
    size_t npx = (size_t)W * (size_t)H;
    if (bg_rgba) memcpy(out_rgba, bg_rgba, npx * 16);
    else memset(out_rgba, 0, npx * 16);
    /* fragment stage: horizontal bands of 16 rows are independent.  Bin the
     * ordered visible list per band first (order inside a band is preserved). */
    if (order_mode != 1) {
        order = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(n_vis ? n_vis : 1));
        n_order = 0;
        for (uint64_t k = 0; k < n_total; k++) if (sp[k].visible) order[n_order++] = k;
    }
    int n_bands = (H + 15) / 16;
    uint64_t *band_off = (uint64_t *)calloc((size_t)n_bands + 1, sizeof(uint64_t));
    for (int pass = 0; pass < 2; pass++) {
        uint64_t *fill = NULL, *blist = NULL;
        if (pass == 1) {
            uint64_t acc = 0;
            for (int b = 0; b < n_bands; b++) { uint64_t c = band_off[b]; band_off[b] = acc; acc += c; }
            band_off[n_bands] = acc;
            blist = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(acc ? acc : 1));
            fill = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)n_bands);
            memcpy(fill, band_off, sizeof(uint64_t) * (size_t)n_bands);
        }
        for (uint64_t i = 0; i < n_order; i++) {
            uint64_t k = order[i];
            float fy0 = ceilf(fs[k].cyp - fs[k].hy - 0.5f), fy1 = floorf(fs[k].cyp + fs[k].hy - 0.5f);
            if (!(fy1 >= fy0) || !(fy1 >= 0.0f) || !(fy0 <= (float)(H - 1))) continue;
            int b0 = fy0 < 0.0f ? 0 : ((int)fy0) >> 4;
            int b1 = fy1 > (float)(H - 1) ? n_bands - 1 : ((int)fy1) >> 4;
            if (b1 > n_bands - 1) b1 = n_bands - 1;
            for (int b = b0; b <= b1; b++) {
                if (pass == 0) band_off[b]++;
                else blist[fill[b]++] = k;
            }
        }
        if (pass == 1) {
#pragma omp parallel for schedule(dynamic, 1)
            for (int band = 0; band < n_bands; band++) {
                int y_lo = band * 16, y_hi = y_lo + 16 > H ? H : y_lo + 16;
                for (uint64_t i = band_off[band]; i < band_off[band + 1]; i++) {
                    uint64_t k = blist[i];
                    if (g_strict == 1) raster_over_strict(&sp[k], scene->splat_scale, W, H, y_lo, y_hi, bg_depth, out_rgba);
                    else raster_over(&sp[k], &fs[k], W, H, y_lo, y_hi, bg_depth, out_rgba);
                }
            }
            free(fill); free(blist);
        }
    }
    free(band_off);
    if (stats) { stats->n_instanced = n_total; stats->n_visible = n_vis; stats->n_pairs16 = n_pairs; }
    free(order); free(sp); free(fs); free(prefix);
    return 0;
}

/* v2 coverage decision of one pixel (F3 / F4 of raster_over, the sequence the HIP compositor evaluates) */
static inline int cover_v2(const orc_frag_setup *fs, int W, int H, int x, int y)
{
    int bx = x & ~15, by = y & ~15;
    float ox = fmaf(0.5f * (float)W, fs->ndcx, 0.5f * (float)W - (float)bx);
    float oy = fmaf(-0.5f * (float)H, fs->ndcy, 0.5f * (float)H - (float)by);
    float nku = -fmaf(fs->iux, ox, fs->iuy * oy);
    float nkv = -fmaf(fs->ivx, ox, fs->ivy * oy);
    float ly = (float)(y - by) + 0.5f, lx = (float)(x - bx) + 0.5f;
    float px = fmaf(fs->iux, lx, fmaf(fs->iuy, ly, nku));
    float py = fmaf(fs->ivx, lx, fmaf(fs->ivy, ly, nkv));
    float r2 = fmaf(py, py, px * px);
    return r2 <= 4.0f;
}

/* strict coverage decision of one pixel (raster_over_strict) */
static inline int cover_strict(const orc_splat *sp, float splat_scale, int W, int H, int x, int y)
{
    const double s = (double)splat_scale;
    const double Mx = sp->major[0], My = sp->major[1], Nx = sp->minor[0], Ny = sp->minor[1];
    const double mm = Mx * Mx + My * My, nn = Nx * Nx + Ny * Ny;
    if (!(mm > 0.0) || !(nn > 0.0) || !(mm < INFINITY) || !(nn < INFINITY)) return 0;
    const double ny = 1.0 - (y + 0.5) / H * 2.0, nx = (x + 0.5) / W * 2.0 - 1.0;
    const double dy = (ny - (double)sp->ndc[1]) * H / s, dx = (nx - (double)sp->ndc[0]) * W / s;
    const float px = (float)((dx * Mx + dy * My) / mm), py = (float)((dx * Nx + dy * Ny) / nn);
    const float A = -(px * px + py * py);
    return !(A < -4.0f) && A == A;
}

/* Where do the two evaluation modes DECIDE differently?  For every instance of every draw: vertex stage in both modes; a
 * splat visible in only one of them marks every pixel it covers there; for a splat visible in both, every pixel (of the
 * union of the two conservative boxes) whose coverage-and-depth decision differs is marked.  mask: W*H bytes (cleared here).
 * counts[0] = splats visible in exactly one mode, [1] = (pixel, splat) decisions that differ, [2] = marked pixels,
 * [3] = splats visible in both modes.  A marked pixel may differ by up to alpha * e^-4 per flipped splat; an unmarked pixel
 * blends the same splats in both modes and differs only by the continuous part of the arithmetic. */
/* vs_a_strict: side A's vertex stage (0 = v2, 1 = strict); its fragment stage is always F1..F4.  Side B is strict throughout.
 * (0: the product's default against the shader text; 1: the product with GSWT_OPT_STRICT_VS against the shader text.) */
ORC_API int orc_compare_modes2(const orc_camera *cam, const orc_scene *scene, const uint32_t *tex,
                               const orc_draw *draws, int n_draws, const float *hmap, int hm_w, int hm_h,
                               int W, int H, const float *bg_depth, int n_threads, int vs_a_strict, uint8_t *mask, uint64_t counts[4])
{
    if (W <= 0 || H <= 0) return -1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
    memset(mask, 0, (size_t)W * (size_t)H);
    uint64_t one_mode = 0, flips = 0, both = 0;
    for (int d = 0; d < n_draws; d++) {
        const orc_draw *dr = &draws[d];
        long cnt = (long)dr->count;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : one_mode, flips, both)
        for (long j = 0; j < cnt; j++) {
            uint32_t mid = dr->map_id ? dr->map_id[j] : 0u;
            uint32_t lid = dr->lod_id ? dr->lod_id[j] : 0u;
            orc_splat a, b;
            orc_frag_setup fa, fb;
            project_impl(cam, scene, &dr->tile, tex, dr->gs_index[j], mid, lid, hmap, hm_w, hm_h, &a, vs_a_strict != 0);
            project_impl(cam, scene, &dr->tile, tex, dr->gs_index[j], mid, lid, hmap, hm_w, hm_h, &b, 1);
            fa.ok = fb.ok = 0;
            if (a.visible) { frag_setup(&a, scene->splat_scale, (float)W, (float)H, &fa); if (!fa.ok) a.visible = 0; }
            if (b.visible) { frag_setup(&b, scene->splat_scale, (float)W, (float)H, &fb); if (!fb.ok) b.visible = 0; }
            if (!a.visible && !b.visible) continue;
            if (a.visible != b.visible) one_mode++; else both++;
            /* union of the two conservative boxes, one pixel of slack */
            float x_lo = 3e38f, x_hi = -3e38f, y_lo = 3e38f, y_hi = -3e38f;
            if (a.visible) { x_lo = fminf(x_lo, fa.cxp - fa.hx); x_hi = fmaxf(x_hi, fa.cxp + fa.hx); y_lo = fminf(y_lo, fa.cyp - fa.hy); y_hi = fmaxf(y_hi, fa.cyp + fa.hy); }
            if (b.visible) { x_lo = fminf(x_lo, fb.cxp - fb.hx); x_hi = fmaxf(x_hi, fb.cxp + fb.hx); y_lo = fminf(y_lo, fb.cyp - fb.hy); y_hi = fmaxf(y_hi, fb.cyp + fb.hy); }
            if (!(x_hi >= x_lo) || !(y_hi >= y_lo)) continue;
            float fx0 = floorf(x_lo) - 1.0f, fx1 = ceilf(x_hi) + 1.0f, fy0 = floorf(y_lo) - 1.0f, fy1 = ceilf(y_hi) + 1.0f;
            if (fx1 < 0.0f || fy1 < 0.0f || fx0 > (float)(W - 1) || fy0 > (float)(H - 1)) continue;
            int x0 = fx0 < 0.0f ? 0 : (int)fx0, x1 = fx1 > (float)(W - 1) ? W - 1 : (int)fx1;
            int y0 = fy0 < 0.0f ? 0 : (int)fy0, y1 = fy1 > (float)(H - 1) ? H - 1 : (int)fy1;
            for (int y = y0; y <= y1; y++)
                for (int x = x0; x <= x1; x++) {
                    float dbuf = bg_depth ? bg_depth[(size_t)y * W + x] : 1.0f;
                    int ca = a.visible && cover_v2(&fa, W, H, x, y) && a.depth < dbuf;
                    int cb = b.visible && cover_strict(&b, scene->splat_scale, W, H, x, y) && b.depth < dbuf;
                    if (ca != cb) { flips++; mask[(size_t)y * W + x] = 1; }
                }
        }
    }
    uint64_t marked = 0;
    for (size_t i = 0; i < (size_t)W * (size_t)H; i++) marked += mask[i];
    counts[0] = one_mode; counts[1] = flips; counts[2] = marked; counts[3] = both;
    return 0;
}

ORC_API int orc_compare_modes(const orc_camera *cam, const orc_scene *scene, const uint32_t *tex,
                              const orc_draw *draws, int n_draws, const float *hmap, int hm_w, int hm_h,
                              int W, int H, const float *bg_depth, int n_threads, uint8_t *mask, uint64_t counts[4])
{
    return orc_compare_modes2(cam, scene, tex, draws, n_draws, hmap, hm_w, hm_h, W, H, bg_depth, n_threads, 0, mask, counts);
}

/* Vertex stage only, for per-splat parity checks: out[k] for every instance. */
ORC_API int orc_project_draws(const orc_camera *cam, const orc_scene *scene, const uint32_t *tex,
                              const orc_draw *draws, int n_draws,
                              const float *hmap, int hm_w, int hm_h, orc_splat *out)
{
    uint64_t k = 0;
    for (int d = 0; d < n_draws; d++) {
        const orc_draw *dr = &draws[d];
        for (uint32_t j = 0; j < dr->count; j++, k++) {
            uint32_t mid = dr->map_id ? dr->map_id[j] : 0u;
            uint32_t lid = dr->lod_id ? dr->lod_id[j] : 0u;
            orc_project(cam, scene, &dr->tile, tex, dr->gs_index[j], mid, lid, hmap, hm_w, hm_h, &out[k]);
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Background passes State::render runs before the splats (state.rs:384-392),  */
/* restated per pixel: skybox (skybox.wgsl + skybox.rs:457-488) and proxy        */
/* (proxy.wgsl + proxy.rs:366-447).  A rasterised fragment's interpolated        */
/* attributes are functions of the pixel's view ray alone, so each pixel is      */
/* evaluated by casting that ray: the cube-map lookup vector is the ray          */
/* direction, and the visible proxy fragment (depth test Less, depth write on,   */
/* no culling, blend None) is the nearest ray / height-field-mesh intersection   */
/* whose depth lies in [0, 1].                                                   */
/* ------------------------------------------------------------------------- */
static void pixel_ray(const float *V, float p00, float p11, int x, int y, int W, int H, float d[3])
{
    float nx = ((float)x + 0.5f) / (float)W * 2.0f - 1.0f;
    float ny = 1.0f - ((float)y + 0.5f) / (float)H * 2.0f;
    float vx = nx / p00, vy = ny / p11, vz = -1.0f;
    d[0] = (V[0] * vx + V[1] * vy) + V[2] * vz;
    d[1] = (V[4] * vx + V[5] * vy) + V[6] * vz;
    d[2] = (V[8] * vx + V[9] * vy) + V[10] * vz;
}

/* cube face order +X -X +Y -Y +Z -Z, bilinear inside the face, clamp to edge */
static void sample_cube(const float *faces, int n, float tx, float ty, float tz, float out[4])
{
    float ax = fabsf(tx), ay = fabsf(ty), az = fabsf(tz);
    int face; float sc, tc, ma;
    if (az >= ax && az >= ay) { face = tz < 0.0f ? 5 : 4; sc = tz < 0.0f ? -tx : tx; tc = -ty; ma = az; }
    else if (ay >= ax) { face = ty < 0.0f ? 3 : 2; sc = tx; tc = ty < 0.0f ? -tz : tz; ma = ay; }
    else { face = tx < 0.0f ? 1 : 0; sc = tx < 0.0f ? tz : -tz; tc = -ty; ma = ax; }
    float s = 0.5f * (sc / ma + 1.0f), t = 0.5f * (tc / ma + 1.0f);
    float x = s * (float)n - 0.5f, y = t * (float)n - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float wx = x - fx0, wy = y - fy0;
    int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    x0 = x0 < 0 ? 0 : (x0 > n - 1 ? n - 1 : x0); x1 = x1 < 0 ? 0 : (x1 > n - 1 ? n - 1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > n - 1 ? n - 1 : y0); y1 = y1 < 0 ? 0 : (y1 > n - 1 ? n - 1 : y1);
    const float *f = faces + (size_t)face * n * n * 4;
    for (int k = 0; k < 3; k++) {
        float c00 = f[((size_t)y0 * n + x0) * 4 + k], c10 = f[((size_t)y0 * n + x1) * 4 + k];
        float c01 = f[((size_t)y1 * n + x0) * 4 + k], c11 = f[((size_t)y1 * n + x1) * 4 + k];
        out[k] = (c00 * (1.0f - wx) + c10 * wx) * (1.0f - wy) + (c01 * (1.0f - wx) + c11 * wx) * wy;
    }
    out[3] = 1.0f;
}

/* skybox.wgsl vs_main / fs_main: lookup vector = (x, -z, y) of the world ray, y negated for a cube map */
ORC_API void orc_skybox(const float *view16, float p00, float p11, int equirectangular, const float *faces, int face_size,
                        int W, int H, float *out_rgba)
{
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float d[3];
            pixel_ray(view16, p00, p11, x, y, W, H, d);
            float tx = d[0], ty = -d[2], tz = d[1];
            if (!equirectangular) ty = -ty;
            sample_cube(faces, face_size, tx, ty, tz, out_rgba + ((size_t)y * W + x) * 4);
        }
}

/* proxy.wgsl Uniforms (224 B) */
typedef struct {
    float height_offset, tile_width; uint32_t surface_type; float width_scale;
    uint32_t map_proxy, use_clip; float clip_height, brightness;
    uint32_t black_background, _pad0[3];
    float view[16], projection[16];
    uint32_t map_half_wh[2]; int32_t center_coord[2];
    float height_map_scale[4], cam_pos[4];
} orc_proxy_uniforms;

typedef struct {
    const orc_proxy_uniforms *u;
    const float *hmap; int hw, hh;
    const float *const *mips; int tex_size, n_mips;
    int nx, ny; float cs, gx0, gy0;          /* grid: cells, cell size, world xy of vertex (0, 0) */
    float GP[16];
} proxy_ctx;

static float proxy_vertex_mapped_height(const proxy_ctx *c, float rx, float ry)
{
    const orc_proxy_uniforms *u = c->u;
    if (u->surface_type != 1u) return 0.0f;
    float xr = (2.0f * (float)u->map_half_wh[0] + 1.0f) * u->tile_width * u->height_map_scale[0];
    float yr = (2.0f * (float)u->map_half_wh[1] + 1.0f) * u->tile_width * u->height_map_scale[1];
    float h_u = (rx + (float)u->map_half_wh[0] * u->tile_width) / xr;
    float h_v = (ry + (float)u->map_half_wh[1] * u->tile_width) / yr;
    return sample_height(c->hmap, c->hw, c->hh, h_u, h_v) * u->height_map_scale[2];
}

static void proxy_tex_bilinear(const float *lvl, int n, float u, float v, float out[3])
{
    float x = u * (float)n - 0.5f, y = v * (float)n - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float wx = x - fx0, wy = y - fy0;
    long x0 = (long)fx0, y0 = (long)fy0;
    long xa = ((x0 % n) + n) % n, xb = (((x0 + 1) % n) + n) % n;
    long ya = ((y0 % n) + n) % n, yb = (((y0 + 1) % n) + n) % n;
    for (int k = 0; k < 3; k++) {
        float c00 = lvl[(ya * n + xa) * 4 + k], c10 = lvl[(ya * n + xb) * 4 + k];
        float c01 = lvl[(yb * n + xa) * 4 + k], c11 = lvl[(yb * n + xb) * 4 + k];
        out[k] = (c00 * (1.0f - wx) + c10 * wx) * (1.0f - wy) + (c01 * (1.0f - wx) + c11 * wx) * wy;
    }
}

/* One triangle (a, b, c with mapped heights ma, mb, mc): Moeller-Trumbore, then the fragment tests.
 * Returns 1 and fills t / depth / mapped height / plane normal when the fragment exists and t < *best_t. */
static int proxy_tri(const proxy_ctx *c, const float o[3], const float d[3], const float a[3], const float b[3], const float cc[3],
                     float ma, float mb, float mc, float *best_t, float *depth_out, float nrm[3], float a_out[3])
{
    float e1[3] = { b[0] - a[0], b[1] - a[1], b[2] - a[2] }, e2[3] = { cc[0] - a[0], cc[1] - a[1], cc[2] - a[2] };
    float pv[3] = { d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0] };
    float det = (e1[0] * pv[0] + e1[1] * pv[1]) + e1[2] * pv[2];
    if (det == 0.0f) return 0;
    float tv[3] = { o[0] - a[0], o[1] - a[1], o[2] - a[2] };
    float bu = ((tv[0] * pv[0] + tv[1] * pv[1]) + tv[2] * pv[2]) / det;
    if (!(bu >= 0.0f && bu <= 1.0f)) return 0;
    float qv[3] = { tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0] };
    float bv = ((d[0] * qv[0] + d[1] * qv[1]) + d[2] * qv[2]) / det;
    if (!(bv >= 0.0f && bu + bv <= 1.0f)) return 0;
    float t = ((e2[0] * qv[0] + e2[1] * qv[1]) + e2[2] * qv[2]) / det;
    if (!(t > 0.0f && t < *best_t)) return 0;
    float mh = (ma + bu * (mb - ma)) + bv * (mc - ma);
    if (c->u->use_clip == 1u && mh < c->u->clip_height) return 0;                 /* fs_main discard */
    float hp[3] = { o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2] };
    const float *V = c->u->view, *GP = c->GP;
    float cv[4], q[4];
    for (int r = 0; r < 4; r++) cv[r] = ((V[r] * hp[0] + V[4 + r] * hp[1]) + V[8 + r] * hp[2]) + V[12 + r];
    for (int r = 0; r < 4; r++) q[r] = ((GP[r] * cv[0] + GP[4 + r] * cv[1]) + GP[8 + r] * cv[2]) + GP[12 + r] * cv[3];
    float depth = q[2] / q[3];
    if (!(q[3] > 0.0f && depth >= 0.0f && depth <= 1.0f)) return 0;             /* near / far clip */
    *best_t = t; *depth_out = depth;
    nrm[0] = e1[1] * e2[2] - e1[2] * e2[1]; nrm[1] = e1[2] * e2[0] - e1[0] * e2[2]; nrm[2] = e1[0] * e2[1] - e1[1] * e2[0];
    a_out[0] = a[0]; a_out[1] = a[1]; a_out[2] = a[2];
    return 1;
}

static void proxy_cell_vertices(const proxy_ctx *c, int i, int j, float v[4][3], float m[4])
{
    /* vertex k = (i + (k & 1), j + (k >> 1)) */
    for (int k = 0; k < 4; k++) {
        int vi = i + (k & 1), vj = j + (k >> 1);
        float rx = c->gx0 + (float)vi * c->cs, ry = c->gy0 + (float)vj * c->cs;
        m[k] = proxy_vertex_mapped_height(c, rx, ry);
        v[k][0] = rx; v[k][1] = ry; v[k][2] = c->u->height_offset + m[k];
    }
}

/* uv of the point where the ray of pixel (x, y) meets the plane (a, nrm) */
static void proxy_plane_uv(const proxy_ctx *c, const float o[3], const float d[3], const float a[3], const float nrm[3], float uv[2])
{
    float num = (nrm[0] * (a[0] - o[0]) + nrm[1] * (a[1] - o[1])) + nrm[2] * (a[2] - o[2]);
    float den = (nrm[0] * d[0] + nrm[1] * d[1]) + nrm[2] * d[2];
    float t = num / den;
    uv[0] = (o[0] + t * d[0]) / c->u->tile_width / 4.0f;
    uv[1] = (o[1] + t * d[1]) / c->u->tile_width / 4.0f;
}

/* proxy.rs:366-447 for one draw (map_proxy = 0: the GRID_DIM grid, 1: the tile-map grid).  rgba / depth are read-modify-write:
 * colour LoadOp::Load, depth test Less against what is there (the caller clears depth to 1.0 before the first draw). */
ORC_API void orc_proxy(const orc_proxy_uniforms *u, int grid_dim, const float *hmap, int hw, int hh,
                       const float *const *mips, int tex_size, int n_mips, int W, int H, float *rgba, float *depth)
{
    proxy_ctx c;
    memset(&c, 0, sizeof(c));
    c.u = u; c.hmap = hmap; c.hw = hw; c.hh = hh; c.mips = mips; c.tex_size = tex_size; c.n_mips = n_mips;
    const float tw = u->tile_width;
    if (u->map_proxy == 1u) {
        c.nx = 2 * (int)u->map_half_wh[0] + 1; c.ny = 2 * (int)u->map_half_wh[1] + 1; c.cs = tw;
        c.gx0 = (float)(-(int)u->map_half_wh[0]) * tw + (float)u->center_coord[0] * tw;
        c.gy0 = (float)(-(int)u->map_half_wh[1]) * tw + (float)u->center_coord[1] * tw;
    } else {
        c.nx = c.ny = grid_dim; c.cs = u->width_scale;
        c.gx0 = (float)(-(grid_dim / 2)) * u->width_scale + floorf((float)u->center_coord[0] * tw / u->width_scale) * u->width_scale;
        c.gy0 = (float)(-(grid_dim / 2)) * u->width_scale + floorf((float)u->center_coord[1] * tw / u->width_scale) * u->width_scale;
    }
    for (int cc = 0; cc < 4; cc++) {
        const float *P = u->projection;
        c.GP[4 * cc + 0] = P[4 * cc + 0]; c.GP[4 * cc + 1] = P[4 * cc + 1];
        c.GP[4 * cc + 2] = 0.5f * P[4 * cc + 2] + 0.5f * P[4 * cc + 3]; c.GP[4 * cc + 3] = P[4 * cc + 3];
    }
    const float p00 = u->projection[0], p11 = u->projection[5];
    const float o[3] = { u->cam_pos[0], u->cam_pos[1], u->cam_pos[2] };
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float d[3];
            pixel_ray(u->view, p00, p11, x, y, W, H, d);
            float best_t = 3.0e38f, dep = 1.0f, nrm[3] = { 0, 0, 1 }, pa[3] = { 0, 0, 0 };
            int hit = 0;
            /* the ray against the grid's xy rectangle, in cell units */
            float ogx = (o[0] - c.gx0) / c.cs, ogy = (o[1] - c.gy0) / c.cs;
            float dgx = d[0] / c.cs, dgy = d[1] / c.cs;
            float t0 = 0.0f, t1 = 3.0e38f;
            int ok = 1;
            if (dgx != 0.0f) {
                float ta = (0.0f - ogx) / dgx, tb = ((float)c.nx - ogx) / dgx;
                t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
            } else if (!(ogx >= 0.0f && ogx <= (float)c.nx)) ok = 0;
            if (dgy != 0.0f) {
                float ta = (0.0f - ogy) / dgy, tb = ((float)c.ny - ogy) / dgy;
                t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
            } else if (!(ogy >= 0.0f && ogy <= (float)c.ny)) ok = 0;
            if (ok && t0 <= t1) {
                if (u->surface_type != 1u) {
                    /* flat grid: every triangle lies in z = height_offset; one plane hit inside the rectangle */
                    float tz = (u->height_offset - o[2]) / d[2];
                    float hx = o[0] + tz * d[0], hy = o[1] + tz * d[1];
                    float cxf = (hx - c.gx0) / c.cs, cyf = (hy - c.gy0) / c.cs;
                    if (d[2] != 0.0f && tz > 0.0f && cxf >= 0.0f && cxf <= (float)c.nx && cyf >= 0.0f && cyf <= (float)c.ny &&
                        !(u->use_clip == 1u && 0.0f < u->clip_height)) {
                        float hp[3] = { hx, hy, o[2] + tz * d[2] };
                        const float *V = u->view, *GP = c.GP;
                        float cv[4], q[4];
                        for (int r = 0; r < 4; r++) cv[r] = ((V[r] * hp[0] + V[4 + r] * hp[1]) + V[8 + r] * hp[2]) + V[12 + r];
                        for (int r = 0; r < 4; r++) q[r] = ((GP[r] * cv[0] + GP[4 + r] * cv[1]) + GP[8 + r] * cv[2]) + GP[12 + r] * cv[3];
                        float dz = q[2] / q[3];
                        if (q[3] > 0.0f && dz >= 0.0f && dz <= 1.0f) {
                            hit = 1; dep = dz; best_t = tz;
                            nrm[0] = 0.0f; nrm[1] = 0.0f; nrm[2] = 1.0f;
                            pa[0] = c.gx0; pa[1] = c.gy0; pa[2] = u->height_offset;
                        }
                    }
                } else {
                    /* 2-D DDA over the cells in ray order; the first cell with a fragment holds the nearest one */
                    float ex = ogx + t0 * dgx, ey = ogy + t0 * dgy;
                    int ci = (int)floorf(ex), cj = (int)floorf(ey);
                    ci = ci < 0 ? 0 : (ci > c.nx - 1 ? c.nx - 1 : ci); cj = cj < 0 ? 0 : (cj > c.ny - 1 ? c.ny - 1 : cj);
                    int sx = dgx > 0.0f ? 1 : -1, sy = dgy > 0.0f ? 1 : -1;
                    float tmx = dgx != 0.0f ? ((float)(ci + (sx > 0 ? 1 : 0)) - ogx) / dgx : 3.0e38f;
                    float tmy = dgy != 0.0f ? ((float)(cj + (sy > 0 ? 1 : 0)) - ogy) / dgy : 3.0e38f;
                    float tdx = dgx != 0.0f ? fabsf(1.0f / dgx) : 3.0e38f, tdy = dgy != 0.0f ? fabsf(1.0f / dgy) : 3.0e38f;
                    for (int step = 0; step < c.nx + c.ny + 2; step++) {
                        float v[4][3], m[4];
                        proxy_cell_vertices(&c, ci, cj, v, m);
                        hit |= proxy_tri(&c, o, d, v[0], v[1], v[2], m[0], m[1], m[2], &best_t, &dep, nrm, pa);
                        hit |= proxy_tri(&c, o, d, v[1], v[3], v[2], m[1], m[3], m[2], &best_t, &dep, nrm, pa);
                        if (hit) break;
                        if (tmx < tmy) { ci += sx; tmx += tdx; } else { cj += sy; tmy += tdy; }
                        if (ci < 0 || ci >= c.nx || cj < 0 || cj >= c.ny) break;
                    }
                }
            }
            size_t pi = (size_t)y * W + x;
            if (!hit || !(dep < depth[pi])) continue;                               /* CompareFunction::Less */
            depth[pi] = dep;
            float *px = rgba + pi * 4;
            if (u->black_background == 1u) { px[0] = 0.0f; px[1] = 0.0f; px[2] = 0.0f; px[3] = 1.0f; continue; }
            /* textureSample: implicit LOD from the uv differences to the right / lower pixel on the fragment's plane */
            float uv[2], uvx[2], uvy[2], dxr[3], dyr[3];
            proxy_plane_uv(&c, o, d, pa, nrm, uv);
            pixel_ray(u->view, p00, p11, x + 1, y, W, H, dxr);
            pixel_ray(u->view, p00, p11, x, y + 1, W, H, dyr);
            proxy_plane_uv(&c, o, dxr, pa, nrm, uvx);
            proxy_plane_uv(&c, o, dyr, pa, nrm, uvy);
            float sz = (float)c.tex_size;
            float ax = (uvx[0] - uv[0]) * sz, ay = (uvx[1] - uv[1]) * sz, bx = (uvy[0] - uv[0]) * sz, by = (uvy[1] - uv[1]) * sz;
            float rho = fmaxf(sqrtf(ax * ax + ay * ay), sqrtf(bx * bx + by * by));
            float lod = log2f(rho);
            if (!(lod > 0.0f)) lod = 0.0f;                                          /* also catches NaN */
            if (lod > (float)(c.n_mips - 1)) lod = (float)(c.n_mips - 1);
            int l0 = (int)floorf(lod), l1 = l0 + 1 > c.n_mips - 1 ? c.n_mips - 1 : l0 + 1;
            float fl = lod - (float)l0;
            float c0[3], c1[3];
            proxy_tex_bilinear(c.mips[l0], c.tex_size >> l0, uv[0], uv[1], c0);
            proxy_tex_bilinear(c.mips[l1], c.tex_size >> l1, uv[0], uv[1], c1);
            for (int k = 0; k < 3; k++) px[k] = (c0[k] * (1.0f - fl) + c1[k] * fl) * u->brightness;
            px[3] = 1.0f;
        }
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------- */
/* Scene::load, scene.rs:115-212: 62-float PLY vertex records -> 32 B rows     */
/* sorted by descending importance exp(s0)exp(s1)exp(s2)*sigmoid(opacity).     */
/* verts: n x 62 f32 = pos3, normal3, f_dc3 + f_rest45, opacity, scale3, rot4  */
/* (scene.rs:19-26).  Rust `as u8` saturates, NaN -> 0.                         */
/* ------------------------------------------------------------------------- */
static inline uint8_t rust_f32_as_u8(float v)
{
    if (v != v) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

typedef struct { float key; uint32_t idx; } orc_imp;

/* stable merge sort, descending key; comparator = partial_cmp(...).unwrap_or(Equal) */
static void imp_sort(orc_imp *a, orc_imp *tmp, size_t n)
{
    for (size_t width = 1; width < n; width *= 2) {
        for (size_t lo = 0; lo < n; lo += 2 * width) {
            size_t mid = lo + width < n ? lo + width : n;
            size_t hi = lo + 2 * width < n ? lo + 2 * width : n;
            size_t i = lo, j = mid, o = lo;
            while (i < mid && j < hi) {
                if (a[j].key > a[i].key) tmp[o++] = a[j++];
                else tmp[o++] = a[i++];
            }
            while (i < mid) tmp[o++] = a[i++];
            while (j < hi) tmp[o++] = a[j++];
        }
        memcpy(a, tmp, n * sizeof(orc_imp));
    }
}

ORC_API void orc_scene_load(const float *verts62, size_t n, uint8_t *rows32)
{
    const float SH_C0 = 0.28209479177387814f;    /* scene.rs:15 */
    orc_imp *imp = (orc_imp *)malloc(sizeof(orc_imp) * (n ? n : 1));
    orc_imp *tmp = (orc_imp *)malloc(sizeof(orc_imp) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) {
        const float *s = verts62 + 62 * i;
        float size = expf(s[55]) * expf(s[56]) * expf(s[57]);       /* :132 */
        float opacity = 1.0f / (1.0f + expf(-s[54]));               /* :133 */
        imp[i].key = size * opacity;
        imp[i].idx = (uint32_t)i;
    }
    imp_sort(imp, tmp, n);
    for (size_t i = 0; i < n; i++) {
        const float *s = verts62 + 62 * (size_t)imp[i].idx;
        uint8_t *row = rows32 + 32 * i;
        float f[6] = { s[0], s[1], s[2], expf(s[55]), expf(s[56]), expf(s[57]) };
        memcpy(row, f, 24);
        row[24] = rust_f32_as_u8((0.5f + SH_C0 * s[6]) * 255.0f);   /* :188-191 */
        row[25] = rust_f32_as_u8((0.5f + SH_C0 * s[7]) * 255.0f);
        row[26] = rust_f32_as_u8((0.5f + SH_C0 * s[8]) * 255.0f);
        row[27] = rust_f32_as_u8((1.0f / (1.0f + expf(-s[54]))) * 255.0f);
        /* powi(2) = x*x ; sum left to right, :199-203 */
        float qlen = sqrtf(((s[58] * s[58] + s[59] * s[59]) + s[60] * s[60]) + s[61] * s[61]);
        for (int k = 0; k < 4; k++)
            row[28 + k] = rust_f32_as_u8(((s[58 + k] / qlen) + 1.0f) * 0.5f * 255.0f);  /* :205-208 */
    }
    free(imp); free(tmp);
}
