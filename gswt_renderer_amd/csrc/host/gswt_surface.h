// gswt_surface.h -- WangTile::surface_mapping (wangtile.rs:1352-1494) and its helpers (height-map fetch :1220-1349, the
// sphere unfolding :1410-1461), shared by libgswt_host (g++) and the device-side worker stages of libgswt_hip (hipcc, GSWT_HD =
// __host__ __device__): one source, one operator sequence, so tile centres, corners and LOD distances agree bit for bit.
#pragma once
#include "gswt_math.h"

namespace gswt_host {

struct SurfaceParams {
    int map_w = 0, map_h = 0, half_w = 0, half_h = 0, center_x = 0, center_y = 0;
    float tile_width = 1.0f;
    int surface_type = 0;               // 0 None, 1 HeightMap, 2 Sphere
    float hm_scale[3] = {1.0f, 1.0f, 1.0f};
    float sphere_radius = 1.0f;
    const float* height_map = nullptr;
    int hm_w = 0, hm_h = 0;
};

GSWT_HD inline void sp_map_to_coord(const SurfaceParams& w, int x, int y, int& cx, int& cy)
{
    cx = x + w.center_x - w.half_w;
    cy = y + w.center_y - w.half_h;
}
GSWT_HD inline V3 sp_coord_to_pos(const SurfaceParams& w, int cx, int cy) { return {(float)cx * w.tile_width, (float)cy * w.tile_width, 0.0f}; }

GSWT_HD inline float hm_texel(const float* hm, int width, int height, long x, long y)
{
    long xi = ((x % width) + width) % width, yi = ((y % height) + height) % height;
    return hm[(size_t)yi * width + xi];
}

GSWT_HD inline void map_fetch_bilinear_aux(const SurfaceParams& w, float u, float v, float dt, float res[5])
{
    const int width = w.hm_w, height = w.hm_h;
    float x = u * (float)width - 0.5f, y = v * (float)height - 0.5f;
    float dx = dt * (float)width, dy = dt * (float)height;
    long x0 = (long)std::floor(x), y0 = (long)std::floor(y);
    float tx = x - (float)x0, ty = y - (float)y0;
    float i00 = hm_texel(w.height_map, width, height, x0, y0), i10 = hm_texel(w.height_map, width, height, x0 + 1, y0);
    float i01 = hm_texel(w.height_map, width, height, x0, y0 + 1), i11 = hm_texel(w.height_map, width, height, x0 + 1, y0 + 1);
    auto bil = [&](float ax, float ay) -> float {
        float i0 = i00 * (1.0f - ax) + i10 * ax;
        float i1 = i01 * (1.0f - ax) + i11 * ax;
        return i0 * (1.0f - ay) + i1 * ay;
    };
    res[0] = bil(tx, ty);
    res[1] = bil(tx + dx, ty);
    res[2] = bil(tx - dx, ty);
    res[3] = bil(tx, ty + dy);
    res[4] = bil(tx, ty - dy);
}

// Canonical sin / cos: Rust's f32::sin / cos are platform libm calls, so their last bits are
// unpinnable; the host uses the one sequence the device kernels use (k = rint(x 2/pi), three-term Cody-Waite with
// fmaf, Cephes minimax polynomials), so tile centres / corners agree with the GPU's sphere mapping.
GSWT_HD inline void csincosf(float x, float& sn, float& cs)
{
    const float kf = std::rint(x * 0.636619772367581343f);
    float r = std::fmaf(kf, -1.5703125f, x);
    r = std::fmaf(kf, -4.837512969970703125e-4f, r);
    r = std::fmaf(kf, -7.54978995489188216e-8f, r);
    const float z = r * r;
    float ps = std::fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f);
    ps = std::fmaf(ps, z, -1.6666654611e-1f);
    const float s = std::fmaf(ps * z, r, r);
    float pc = std::fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f);
    pc = std::fmaf(pc, z, 4.166664568298827e-2f);
    const float c = std::fmaf(pc * z, z, std::fmaf(-0.5f, z, 1.0f));
    const int q = (int)kf & 3;
    float so = (q & 1) ? c : s, co = (q & 1) ? s : c;
    if (q == 2 || q == 3) so = -so;
    if (q == 1 || q == 2) co = -co;
    sn = so; cs = co;
}

// get_uv + uv_to_pos closures of surface_mapping, wangtile.rs:1410-1461
GSWT_HD inline V3 sphere_point(float block_w, float bidx, float bidy, float bx, float by)
{
    const float PI = 3.14159265358979323846f;
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
    u = u + 0.5f * std::floor(v);
    u = u * (2.0f * PI);
    v = (v - 0.5f) * PI;
    float su, cu, sv, cv;
    csincosf(u, su, cu);
    csincosf(v, sv, cv);
    return V3{cv * cu, cv * su, sv};
}

// ---- WangTile::surface_mapping, wangtile.rs:1352-1494 --------------------------------------
GSWT_HD inline void surface_mapping(const SurfaceParams& w, int mcx, int mcy, V3 pos, bool to_world, V3& new_pos, M3& transform)
{
    new_pos = pos;
    transform = M3{};
    const float DELTA = 0.001f;
    const float tw = w.tile_width;
    if (w.surface_type == 2) {
        const float xmax = (float)w.map_w * tw, ymax = (float)w.map_h * tw;
        const float block_w = xmax / 5.0f;
        int c0x, c0y;
        sp_map_to_coord(w, 0, 0, c0x, c0y);
        new_pos = new_pos - sp_coord_to_pos(w, c0x, c0y);
        const float bidx = (float)(5 * mcx / w.map_w), bidy = (float)(2 * mcy / w.map_h);
        const float bx = new_pos.x - bidx * block_w, by = new_pos.y - bidy * block_w;
        const V3 lz = sphere_point(block_w, bidx, bidy, bx, by);
        const float r = w.sphere_radius;
        new_pos = lz * r;
        const float dt = DELTA * ymax;
        const V3 pr = sphere_point(block_w, bidx, bidy, bx + dt, by) * r;
        const V3 pl = sphere_point(block_w, bidx, bidy, bx - dt, by) * r;
        const V3 pu = sphere_point(block_w, bidx, bidy, bx, by + dt) * r;
        const V3 pd = sphere_point(block_w, bidx, bidy, bx, by - dt) * r;
        const V3 lx = (pr - pl) / (2.0f * dt), ly = (pu - pd) / (2.0f * dt);
        M3 l2w = from_cols(lx, ly, lz);
        new_pos = new_pos + l2w * V3{0.0f, 0.0f, pos.z};
        transform = to_world ? l2w : invert(l2w);
        return;
    }
    if (w.surface_type != 1) return;
    float xr = ((float)w.map_w * tw) * w.hm_scale[0];
    float yr = ((float)w.map_h * tw) * w.hm_scale[1];
    float u = (pos.x + (float)(unsigned)w.half_w * tw) / xr;
    float v = (pos.y + (float)(unsigned)w.half_h * tw) / yr;
    float hv[5];
    map_fetch_bilinear_aux(w, u, v, DELTA, hv);
    const float hz = w.hm_scale[2];
    new_pos.z = hv[0] * hz;
    float h_r = hv[1] * hz, h_l = hv[2] * hz, h_u = hv[3] * hz, h_d = hv[4] * hz;
    V3 lx{1.0f, 0.0f, (h_r - h_l) / ((2.0f * DELTA) * xr)};
    V3 ly{0.0f, 1.0f, (h_u - h_d) / ((2.0f * DELTA) * yr)};
    V3 lz = normalize(cross(lx, ly));
    M3 l2w = from_cols(lx, ly, lz);
    new_pos = new_pos + l2w * V3{0.0f, 0.0f, pos.z};
    transform = to_world ? l2w : invert(l2w);
}

}  // namespace gswt_host
