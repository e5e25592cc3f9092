// gswt_passes.hip -- the two background passes State::render runs before the splats (state.rs:384-392), as
// per-pixel compute kernels: their outputs are the bg_rgba / bg_depth inputs of gswt_render.
//
//   k_skybox : skybox.wgsl vs_main + fs_main (cube map lookup along the pixel's view ray), skybox.rs:457-488
//   k_proxy  : proxy.wgsl vs_main + fs_main + depth state (proxy.rs:96-134,366-447): the height-mapped ground grid.  Instead of
//              pushing up to 2 x 2048^2 triangles through a rasteriser, every pixel casts its view ray at the height field
//              (2-D DDA over the grid cells, two triangles per cell); the nearest fragment with depth in [0, 1] is exactly
//              what depth-test-Less rasterisation keeps.  Implicit-LOD trilinear texturing from the uv differences to the
//              right / lower pixel on the fragment's plane.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (same flags as gswt_kernels.hip).
#include "gswt_device.h"

namespace gswt {

// Cube-map face selection (WebGPU / Vulkan major-axis rule) + bilinear filter inside the face, clamp to edge
// (skybox.rs:420-426: ClampToEdge, Linear, level 0).  faces: [6][n][n] float4, +X -X +Y -Y +Z -Z.
__device__ __forceinline__ float4 sample_cube(const float4* __restrict__ faces, int n, float tx, float ty, float tz)
{
    const float ax = fabsf(tx), ay = fabsf(ty), az = fabsf(tz);
    int face;
    float sc, tc, ma;
    if (az >= ax && az >= ay) { face = tz < 0.0f ? 5 : 4; sc = tz < 0.0f ? -tx : tx; tc = -ty; ma = az; }
    else if (ay >= ax) { face = ty < 0.0f ? 3 : 2; sc = tx; tc = ty < 0.0f ? -tz : tz; ma = ay; }
    else { face = tx < 0.0f ? 1 : 0; sc = tx < 0.0f ? tz : -tz; tc = -ty; ma = ax; }
    const float s = 0.5f * (sc / ma + 1.0f), t = 0.5f * (tc / ma + 1.0f);
    const float x = s * (float)n - 0.5f, y = t * (float)n - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float wx = x - fx0, wy = y - fy0;
    int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), n - 1); x1 = min(max(x1, 0), n - 1);
    y0 = min(max(y0, 0), n - 1); y1 = min(max(y1, 0), n - 1);
    const float4* f = faces + (size_t)face * n * n;
    const float4 c00 = f[(size_t)y0 * n + x0], c10 = f[(size_t)y0 * n + x1];
    const float4 c01 = f[(size_t)y1 * n + x0], c11 = f[(size_t)y1 * n + x1];
    float4 o;
    o.x = (c00.x * (1.0f - wx) + c10.x * wx) * (1.0f - wy) + (c01.x * (1.0f - wx) + c11.x * wx) * wy;
    o.y = (c00.y * (1.0f - wx) + c10.y * wx) * (1.0f - wy) + (c01.y * (1.0f - wx) + c11.y * wx) * wy;
    o.z = (c00.z * (1.0f - wx) + c10.z * wx) * (1.0f - wy) + (c01.z * (1.0f - wx) + c11.z * wx) * wy;
    o.w = 1.0f;
    return o;
}



// One thread per pixel.  The rasterised cube's interpolated attribute `position` at a pixel is a positive multiple
// of the pixel's world-space view direction, so the lookup vector is that direction, re-ordered as the vertex
// shader does (skybox.wgsl:31-38): (x, -z, y), y negated again for a cube map.
__global__ __launch_bounds__(256) void k_skybox(const SkyArgs a, const float4* __restrict__ faces, float4* __restrict__ out)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= a.width || y >= a.height) return;
    const float nx = ((float)x + 0.5f) / (float)a.width * 2.0f - 1.0f;
    const float ny = 1.0f - ((float)y + 0.5f) / (float)a.height * 2.0f;
    const float vx = nx / a.p00, vy = ny / a.p11, vz = -1.0f;
    // d = R^T v, R = upper 3x3 of view (column-major)
    const float dx = (a.V[0] * vx + a.V[1] * vy) + a.V[2] * vz;
    const float dy = (a.V[4] * vx + a.V[5] * vy) + a.V[6] * vz;
    const float dz = (a.V[8] * vx + a.V[9] * vy) + a.V[10] * vz;
    float tx = dx, ty = -dz, tz = dy;
    if (a.equirectangular == 0) ty = -ty;
    out[(size_t)y * a.width + x] = sample_cube(faces, a.face_size, tx, ty, tz);
}

void launch_skybox(hipStream_t s, const float* view16, float p00, float p11, int width, int height, int face_size, int equirect,
                   const float4* faces, float4* out)
{
    SkyArgs a;
    for (int i = 0; i < 16; i++) a.V[i] = view16[i];
    a.p00 = p00; a.p11 = p11; a.width = width; a.height = height; a.face_size = face_size; a.equirectangular = equirect;
    hipLaunchKernelGGL(k_skybox, dim3((width + 63) / 64, (height + 3) / 4), dim3(256), 0, s, a, faces, out);
}


// ---- proxy -----------------------------------------------------------------------------------------


// x mod w of the repeat samplers without 64-bit division (see gswt_kernels.hip): float quotient + two fix-ups
__device__ __noinline__ int p_wrap_repeat_slow(float fx, int w)
{
    const long xl = (long)fx;
    return (int)(((xl % w) + w) % w);
}

__device__ __forceinline__ int p_wrap_repeat(float fx, int w)
{
    if (fabsf(fx) < 8388608.0f && w < 8388608) {
        const int x = (int)fx;
        int r = x - w * (int)floorf((float)x / (float)w);
        if (r < 0) r += w;
        if (r >= w) r -= w;
        return r;
    }
    return p_wrap_repeat_slow(fx, w);
}

// WebGPU bilinear, R32Float, repeat (same sampler as the splat kernel's height map)
__device__ __forceinline__ float p_sample_height(const float* __restrict__ hm, int w, int h, float u, float v)
{
    float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
    float fx0 = floorf(x), fy0 = floorf(y);
    float tx = x - fx0, ty = y - fy0;
    const int xa = p_wrap_repeat(fx0, w), ya = p_wrap_repeat(fy0, h);
    const int xb = xa + 1 == w ? 0 : xa + 1, yb = ya + 1 == h ? 0 : ya + 1;
    float i00 = hm[ya * w + xa], i10 = hm[ya * w + xb];
    float i01 = hm[yb * w + xa], i11 = hm[yb * w + xb];
    float i0 = i00 * (1.0f - tx) + i10 * tx;
    float i1 = i01 * (1.0f - tx) + i11 * tx;
    return i0 * (1.0f - ty) + i1 * ty;
}

__device__ __forceinline__ void pixel_ray(const float* V, float p00, float p11, int x, int y, int W, int H, float d[3])
{
    const float nx = ((float)x + 0.5f) / (float)W * 2.0f - 1.0f;
    const float ny = 1.0f - ((float)y + 0.5f) / (float)H * 2.0f;
    const float vx = nx / p00, vy = ny / p11, vz = -1.0f;
    d[0] = (V[0] * vx + V[1] * vy) + V[2] * vz;
    d[1] = (V[4] * vx + V[5] * vy) + V[6] * vz;
    d[2] = (V[8] * vx + V[9] * vy) + V[10] * vz;
}

__device__ __forceinline__ float proxy_mapped_height(const ProxyArgs& a, const float* __restrict__ hm, float rx, float ry)
{
    if (a.surface_type != 1u) return 0.0f;
    const float xr = (2.0f * (float)a.map_half_wh[0] + 1.0f) * a.tile_width * a.height_map_scale[0];
    const float yr = (2.0f * (float)a.map_half_wh[1] + 1.0f) * a.tile_width * a.height_map_scale[1];
    const float h_u = (rx + (float)a.map_half_wh[0] * a.tile_width) / xr;
    const float h_v = (ry + (float)a.map_half_wh[1] * a.tile_width) / yr;
    return p_sample_height(hm, a.hm_w, a.hm_h, h_u, h_v) * a.height_map_scale[2];
}

__device__ __forceinline__ bool proxy_depth(const ProxyArgs& a, const float hp[3], float& depth)
{
    float cv[4], q[4];
    for (int r = 0; r < 4; r++) cv[r] = ((a.V[r] * hp[0] + a.V[4 + r] * hp[1]) + a.V[8 + r] * hp[2]) + a.V[12 + r];
    for (int r = 0; r < 4; r++) q[r] = ((a.GP[r] * cv[0] + a.GP[4 + r] * cv[1]) + a.GP[8 + r] * cv[2]) + a.GP[12 + r] * cv[3];
    depth = q[2] / q[3];
    return q[3] > 0.0f && depth >= 0.0f && depth <= 1.0f;
}

// Moeller-Trumbore + the fragment tests (fs_main discard, near / far clip); keeps the nearest
__device__ __forceinline__ bool proxy_tri(const ProxyArgs& a, const float o[3], const float d[3], const float* va, const float* vb,
                                          const float* vc, float ma, float mb, float mc, float& best_t, float& depth_out,
                                          float nrm[3], float pa[3])
{
    const float e1[3] = {vb[0] - va[0], vb[1] - va[1], vb[2] - va[2]}, e2[3] = {vc[0] - va[0], vc[1] - va[1], vc[2] - va[2]};
    const float pv[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
    const float det = (e1[0] * pv[0] + e1[1] * pv[1]) + e1[2] * pv[2];
    if (det == 0.0f) return false;
    const float tv[3] = {o[0] - va[0], o[1] - va[1], o[2] - va[2]};
    const float bu = ((tv[0] * pv[0] + tv[1] * pv[1]) + tv[2] * pv[2]) / det;
    if (!(bu >= 0.0f && bu <= 1.0f)) return false;
    const float qv[3] = {tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0]};
    const float bv = ((d[0] * qv[0] + d[1] * qv[1]) + d[2] * qv[2]) / det;
    if (!(bv >= 0.0f && bu + bv <= 1.0f)) return false;
    const float t = ((e2[0] * qv[0] + e2[1] * qv[1]) + e2[2] * qv[2]) / det;
    if (!(t > 0.0f && t < best_t)) return false;
    const float mh = (ma + bu * (mb - ma)) + bv * (mc - ma);
    if (a.use_clip == 1u && mh < a.clip_height) return false;
    const float hp[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
    float depth;
    if (!proxy_depth(a, hp, depth)) return false;
    best_t = t; depth_out = depth;
    nrm[0] = e1[1] * e2[2] - e1[2] * e2[1]; nrm[1] = e1[2] * e2[0] - e1[0] * e2[2]; nrm[2] = e1[0] * e2[1] - e1[1] * e2[0];
    pa[0] = va[0]; pa[1] = va[1]; pa[2] = va[2];
    return true;
}

__device__ __forceinline__ void proxy_plane_uv(const ProxyArgs& a, const float o[3], const float d[3], const float pa[3], const float nrm[3],
                                               float uv[2])
{
    const float num = (nrm[0] * (pa[0] - o[0]) + nrm[1] * (pa[1] - o[1])) + nrm[2] * (pa[2] - o[2]);
    const float den = (nrm[0] * d[0] + nrm[1] * d[1]) + nrm[2] * d[2];
    const float t = num / den;
    uv[0] = (o[0] + t * d[0]) / a.tile_width / 4.0f;
    uv[1] = (o[1] + t * d[1]) / a.tile_width / 4.0f;
}

__device__ __forceinline__ void proxy_tex_bilinear(const float4* __restrict__ lvl, int n, float u, float v, float out[3])
{
    const float x = u * (float)n - 0.5f, y = v * (float)n - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float wx = x - fx0, wy = y - fy0;
    const int xa = p_wrap_repeat(fx0, n), ya = p_wrap_repeat(fy0, n);
    const int xb = xa + 1 == n ? 0 : xa + 1, yb = ya + 1 == n ? 0 : ya + 1;
    const float4 c00 = lvl[ya * n + xa], c10 = lvl[ya * n + xb], c01 = lvl[yb * n + xa], c11 = lvl[yb * n + xb];
    out[0] = (c00.x * (1.0f - wx) + c10.x * wx) * (1.0f - wy) + (c01.x * (1.0f - wx) + c11.x * wx) * wy;
    out[1] = (c00.y * (1.0f - wx) + c10.y * wx) * (1.0f - wy) + (c01.y * (1.0f - wx) + c11.y * wx) * wy;
    out[2] = (c00.z * (1.0f - wx) + c10.z * wx) * (1.0f - wy) + (c01.z * (1.0f - wx) + c11.z * wx) * wy;
}

__global__ __launch_bounds__(256) void k_proxy(const ProxyArgs a, const float* __restrict__ hm, const float4* __restrict__ tex,
                                               float4* __restrict__ rgba, float* __restrict__ depth_buf)
{
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (x >= a.width || y >= a.height) return;
    const float o[3] = {a.cam[0], a.cam[1], a.cam[2]};
    float d[3];
    pixel_ray(a.V, a.p00, a.p11, x, y, a.width, a.height, d);
    float best_t = 3.0e38f, dep = 1.0f, nrm[3] = {0.0f, 0.0f, 1.0f}, pa[3] = {0.0f, 0.0f, 0.0f};
    bool hit = false;
    const float ogx = (o[0] - a.gx0) / a.cs, ogy = (o[1] - a.gy0) / a.cs;
    const float dgx = d[0] / a.cs, dgy = d[1] / a.cs;
    float t0 = 0.0f, t1 = 3.0e38f;
    bool ok = true;
    if (dgx != 0.0f) {
        const float ta = (0.0f - ogx) / dgx, tb = ((float)a.nx - ogx) / dgx;
        t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
    } else if (!(ogx >= 0.0f && ogx <= (float)a.nx)) ok = false;
    if (dgy != 0.0f) {
        const float ta = (0.0f - ogy) / dgy, tb = ((float)a.ny - ogy) / dgy;
        t0 = fmaxf(t0, fminf(ta, tb)); t1 = fminf(t1, fmaxf(ta, tb));
    } else if (!(ogy >= 0.0f && ogy <= (float)a.ny)) ok = false;
    if (ok && t0 <= t1) {
        if (a.surface_type != 1u) {
            const float tz = (a.height_offset - o[2]) / d[2];
            const float hx = o[0] + tz * d[0], hy = o[1] + tz * d[1];
            const float cxf = (hx - a.gx0) / a.cs, cyf = (hy - a.gy0) / a.cs;
            if (d[2] != 0.0f && tz > 0.0f && cxf >= 0.0f && cxf <= (float)a.nx && cyf >= 0.0f && cyf <= (float)a.ny &&
                !(a.use_clip == 1u && 0.0f < a.clip_height)) {
                const float hp[3] = {hx, hy, o[2] + tz * d[2]};
                float dz;
                if (proxy_depth(a, hp, dz)) {
                    hit = true; dep = dz; best_t = tz;
                    pa[0] = a.gx0; pa[1] = a.gy0; pa[2] = a.height_offset;
                }
            }
        } else {
            const float ex = ogx + t0 * dgx, ey = ogy + t0 * dgy;
            int ci = (int)floorf(ex), cj = (int)floorf(ey);
            ci = ci < 0 ? 0 : (ci > a.nx - 1 ? a.nx - 1 : ci); cj = cj < 0 ? 0 : (cj > a.ny - 1 ? a.ny - 1 : cj);
            const int sx = dgx > 0.0f ? 1 : -1, sy = dgy > 0.0f ? 1 : -1;
            float tmx = dgx != 0.0f ? ((float)(ci + (sx > 0 ? 1 : 0)) - ogx) / dgx : 3.0e38f;
            float tmy = dgy != 0.0f ? ((float)(cj + (sy > 0 ? 1 : 0)) - ogy) / dgy : 3.0e38f;
            const float tdx = dgx != 0.0f ? fabsf(1.0f / dgx) : 3.0e38f, tdy = dgy != 0.0f ? fabsf(1.0f / dgy) : 3.0e38f;
            const int max_steps = a.nx + a.ny + 2;          // every wave leaves the loop after at most this many cells
            for (int step = 0; step < max_steps; step++) {
                float v[4][3], m[4];
                for (int k = 0; k < 4; k++) {
                    const int vi = ci + (k & 1), vj = cj + (k >> 1);
                    const float rx = a.gx0 + (float)vi * a.cs, ry = a.gy0 + (float)vj * a.cs;
                    m[k] = proxy_mapped_height(a, hm, rx, ry);
                    v[k][0] = rx; v[k][1] = ry; v[k][2] = a.height_offset + m[k];
                }
                const bool h1 = proxy_tri(a, o, d, v[0], v[1], v[2], m[0], m[1], m[2], best_t, dep, nrm, pa);
                const bool h2 = proxy_tri(a, o, d, v[1], v[3], v[2], m[1], m[3], m[2], best_t, dep, nrm, pa);
                hit = hit || h1 || h2;
                if (hit) break;
                if (tmx < tmy) { ci += sx; tmx += tdx; } else { cj += sy; tmy += tdy; }
                if (ci < 0 || ci >= a.nx || cj < 0 || cj >= a.ny) break;
            }
        }
    }
    const size_t pi = (size_t)y * a.width + x;
    if (!hit || !(dep < depth_buf[pi])) return;                  // CompareFunction::Less, depth write on
    depth_buf[pi] = dep;
    if (a.black_background == 1u) { rgba[pi] = make_float4(0.0f, 0.0f, 0.0f, 1.0f); return; }
    float uv[2], uvx[2], uvy[2], dxr[3], dyr[3];
    proxy_plane_uv(a, o, d, pa, nrm, uv);
    pixel_ray(a.V, a.p00, a.p11, x + 1, y, a.width, a.height, dxr);
    pixel_ray(a.V, a.p00, a.p11, x, y + 1, a.width, a.height, dyr);
    proxy_plane_uv(a, o, dxr, pa, nrm, uvx);
    proxy_plane_uv(a, o, dyr, pa, nrm, uvy);
    const float sz = (float)a.tex_size;
    const float ax = (uvx[0] - uv[0]) * sz, ay = (uvx[1] - uv[1]) * sz, bx = (uvy[0] - uv[0]) * sz, by = (uvy[1] - uv[1]) * sz;
    const float rho = fmaxf(sqrtf(ax * ax + ay * ay), sqrtf(bx * bx + by * by));
    float lod = log2f(rho);
    if (!(lod > 0.0f)) lod = 0.0f;
    if (lod > (float)(a.n_mips - 1)) lod = (float)(a.n_mips - 1);
    const int l0 = (int)floorf(lod), l1 = l0 + 1 > a.n_mips - 1 ? a.n_mips - 1 : l0 + 1;
    const float fl = lod - (float)l0;
    float c0[3], c1[3];
    proxy_tex_bilinear(tex + a.mip_off[l0], a.tex_size >> l0, uv[0], uv[1], c0);
    proxy_tex_bilinear(tex + a.mip_off[l1], a.tex_size >> l1, uv[0], uv[1], c1);
    rgba[pi] = make_float4((c0[0] * (1.0f - fl) + c1[0] * fl) * a.brightness, (c0[1] * (1.0f - fl) + c1[1] * fl) * a.brightness,
                           (c0[2] * (1.0f - fl) + c1[2] * fl) * a.brightness, 1.0f);
}

__global__ __launch_bounds__(256) void k_fill_f32(float* __restrict__ p, size_t n, float v)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

void launch_fill_f32(hipStream_t s, float* p, size_t n, float v)
{
    if (n) hipLaunchKernelGGL(k_fill_f32, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, p, n, v);
}

void launch_proxy(hipStream_t s, const ProxyArgs& a, const float* hm, const float4* tex, float4* rgba, float* depth)
{
    hipLaunchKernelGGL(k_proxy, dim3((a.width + 15) / 16, (a.height + 15) / 16), dim3(256), 0, s, a, hm, tex, rgba, depth);
}

}  // namespace gswt
