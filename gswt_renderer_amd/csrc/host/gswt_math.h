// gswt_math.h -- f32 vector / matrix helpers with cgmath 0.18's operand order.
// One rounding per operator (build with -ffp-contract=off); column-major matrices.
#pragma once
// GSWT_HD: empty for g++ (libgswt_host); gswt_worker.hip defines it as __host__ __device__ so the device-side worker stages
// run the very same operator sequences.
#ifndef GSWT_HD
#define GSWT_HD
#endif
#include <cmath>
#include <cstdint>
#include <cstring>

namespace gswt_host {

struct V3 {
    float x = 0, y = 0, z = 0;
};
GSWT_HD inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
GSWT_HD inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
GSWT_HD inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
GSWT_HD inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
GSWT_HD inline bool is_zero(V3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
GSWT_HD inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
GSWT_HD inline float magnitude(V3 a) { return std::sqrt(dot(a, a)); }
GSWT_HD inline float distance2(V3 a, V3 b) { V3 d = b - a; return dot(d, d); }     // MetricSpace: (other - self).magnitude2()
GSWT_HD inline float distance(V3 a, V3 b) { return magnitude(b - a); }
GSWT_HD inline V3 normalize(V3 a) { return a * (1.0f / magnitude(a)); }            // InnerSpace::normalize
GSWT_HD inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct M3 {
    float m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // [3*c + r]
    GSWT_HD V3 col(int c) const { return {m[3 * c], m[3 * c + 1], m[3 * c + 2]}; }
    GSWT_HD float at(int c, int r) const { return m[3 * c + r]; }
};
GSWT_HD inline M3 from_cols(V3 a, V3 b, V3 c) { M3 r; r.m[0] = a.x; r.m[1] = a.y; r.m[2] = a.z; r.m[3] = b.x; r.m[4] = b.y; r.m[5] = b.z; r.m[6] = c.x; r.m[7] = c.y; r.m[8] = c.z; return r; }
GSWT_HD inline V3 operator*(const M3& m, V3 v) { return (m.col(0) * v.x + m.col(1) * v.y) + m.col(2) * v.z; }
GSWT_HD inline M3 invert(const M3& m)    // cgmath Matrix3::invert (determinant nonzero assumed by the reference's unwrap)
{
    V3 c0 = m.col(0), c1 = m.col(1), c2 = m.col(2);
    float det = (c0.x * (c1.y * c2.z - c2.y * c1.z) - c1.x * (c0.y * c2.z - c2.y * c0.z)) + c2.x * (c0.y * c1.z - c1.y * c0.z);
    V3 r0 = cross(c1, c2) / det, r1 = cross(c2, c0) / det, r2 = cross(c0, c1) / det;
    M3 o;   // from_cols(r0, r1, r2).transpose()
    o.m[0] = r0.x; o.m[1] = r1.x; o.m[2] = r2.x;
    o.m[3] = r0.y; o.m[4] = r1.y; o.m[5] = r2.y;
    o.m[6] = r0.z; o.m[7] = r1.z; o.m[8] = r2.z;
    return o;
}

struct Quat {
    float s = 0, x = 0, y = 0, z = 0;
};
GSWT_HD inline Quat quat_from_mat3(const M3& m)     // cgmath From<Matrix3> for Quaternion
{
    float trace = (m.at(0, 0) + m.at(1, 1)) + m.at(2, 2);
    const float half = 0.5f;
    Quat q;
    if (trace >= 0.0f) {
        float s = std::sqrt(1.0f + trace);
        q.s = half * s;
        s = half / s;
        q.x = (m.at(1, 2) - m.at(2, 1)) * s;
        q.y = (m.at(2, 0) - m.at(0, 2)) * s;
        q.z = (m.at(0, 1) - m.at(1, 0)) * s;
    } else if (m.at(0, 0) > m.at(1, 1) && m.at(0, 0) > m.at(2, 2)) {
        float s = std::sqrt(((m.at(0, 0) - m.at(1, 1)) - m.at(2, 2)) + 1.0f);
        q.x = half * s;
        s = half / s;
        q.y = (m.at(1, 0) + m.at(0, 1)) * s;
        q.z = (m.at(0, 2) + m.at(2, 0)) * s;
        q.s = (m.at(1, 2) - m.at(2, 1)) * s;
    } else if (m.at(1, 1) > m.at(2, 2)) {
        float s = std::sqrt(((m.at(1, 1) - m.at(0, 0)) - m.at(2, 2)) + 1.0f);
        q.y = half * s;
        s = half / s;
        q.z = (m.at(2, 1) + m.at(1, 2)) * s;
        q.x = (m.at(1, 0) + m.at(0, 1)) * s;
        q.s = (m.at(2, 0) - m.at(0, 2)) * s;
    } else {
        float s = std::sqrt(((m.at(2, 2) - m.at(0, 0)) - m.at(1, 1)) + 1.0f);
        q.z = half * s;
        s = half / s;
        q.x = (m.at(0, 2) + m.at(2, 0)) * s;
        q.y = (m.at(2, 1) + m.at(1, 2)) * s;
        q.s = (m.at(0, 1) - m.at(1, 0)) * s;
    }
    return q;
}
GSWT_HD inline M3 mat3_from_quat(Quat q)            // cgmath From<Quaternion> for Matrix3
{
    float x2 = q.x + q.x, y2 = q.y + q.y, z2 = q.z + q.z;
    float xx2 = x2 * q.x, xy2 = x2 * q.y, xz2 = x2 * q.z;
    float yy2 = y2 * q.y, yz2 = y2 * q.z, zz2 = z2 * q.z;
    float sy2 = y2 * q.s, sz2 = z2 * q.s, sx2 = x2 * q.s;
    M3 r;
    r.m[0] = (1.0f - yy2) - zz2; r.m[1] = xy2 + sz2; r.m[2] = xz2 - sy2;
    r.m[3] = xy2 - sz2; r.m[4] = (1.0f - xx2) - zz2; r.m[5] = yz2 + sx2;
    r.m[6] = xz2 + sy2; r.m[7] = yz2 - sx2; r.m[8] = (1.0f - xx2) - yy2;
    return r;
}

// Matrix4 * Matrix4 and Matrix4 * (x, y, z, w), flat column-major [4*c + r]
GSWT_HD inline void mat4_mul(const float* a, const float* b, float* out)
{
    for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) {
            float acc = a[r] * b[4 * c];
            for (int k = 1; k < 4; k++) acc = acc + a[4 * k + r] * b[4 * c + k];
            out[4 * c + r] = acc;
        }
}
GSWT_HD inline void mat4_vec(const float* m, const float v[4], float out[4])
{
    for (int r = 0; r < 4; r++) {
        float acc = m[r] * v[0];
        for (int k = 1; k < 4; k++) acc = acc + m[4 * k + r] * v[k];
        out[r] = acc;
    }
}
// cgmath::perspective; cot(fovy/2) evaluated in double and rounded once
GSWT_HD inline void perspective(float fovy_deg, float aspect, float near_, float far_, float* m)
{
    float fovy = fovy_deg * (float)(3.14159265358979323846 / 180.0);
    float f = (float)(1.0 / std::tan((double)fovy / 2.0));
    for (int i = 0; i < 16; i++) m[i] = 0.0f;
    m[0] = f / aspect;
    m[5] = f;
    m[10] = (far_ + near_) / (near_ - far_);
    m[11] = -1.0f;
    m[14] = (2.0f * far_ * near_) / (near_ - far_);
}
// Matrix4::look_at_rh -> look_to_rh(eye, center - eye, up)
GSWT_HD inline void look_at_rh(V3 eye, V3 center, V3 up, float* m)
{
    V3 f = normalize(center - eye);
    V3 s = normalize(cross(f, up));
    V3 u = cross(s, f);
    m[0] = s.x; m[1] = u.x; m[2] = -f.x; m[3] = 0.0f;
    m[4] = s.y; m[5] = u.y; m[6] = -f.y; m[7] = 0.0f;
    m[8] = s.z; m[9] = u.z; m[10] = -f.z; m[11] = 0.0f;
    m[12] = -dot(eye, s); m[13] = -dot(eye, u); m[14] = dot(eye, f); m[15] = 1.0f;
}

// Rust `as i32` / `as u8` on f32: saturating, NaN -> 0
GSWT_HD inline int32_t rust_as_i32(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int32_t)v;
}
GSWT_HD inline uint8_t rust_as_u8(float v)
{
    if (v != v) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

}  // namespace gswt_host
