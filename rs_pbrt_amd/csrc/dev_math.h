// Device-side leaf math for the gfx950 wavefront path tracer.
//
// Everything here is f32 with NO fused multiply-add (the translation unit is built with
// -ffp-contract=off) so that results are bit-identical to rs_pbrt's scalar Rust code, which
// LLVM never contracts; the libm functions Rust calls are restated from glibc (glibc_libm.h).  Each helper names the rs_pbrt function whose arithmetic it has to
// reproduce (paths relative to the rs_pbrt tree).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RDEV __device__ __forceinline__
#define RDEVN __device__ inline
// A kernel that is not a template.  librspt.so is several translation units (tu_decl.h): the units that exist only to carry
// instantiations of the heavy kernel templates (RSPT_TU_TEMPLATES_ONLY) see the plain kernels as templates nobody instantiates,
// so that each of those has exactly one definition — in librspt.hip, which launches them.
#ifdef RSPT_TU_TEMPLATES_ONLY
#define RSPT_PLAIN_KERNEL template <int RSPT_NEVER_INSTANTIATED = 0> __global__
#else
#define RSPT_PLAIN_KERNEL __global__
#endif

namespace rspt {

// src/core/pbrt.rs:16-23, src/core/rng.rs:13
#define RSPT_MACHINE_EPS 5.9604644775390625e-8f /* f32::EPSILON * 0.5 */
#define RSPT_SHADOW_EPS 0.0001f
#define RSPT_PI 3.14159265358979323846f
#define RSPT_INV_PI 0.31830988618379067154f
#define RSPT_PI_OVER_2 1.57079632679489661923f
#define RSPT_PI_OVER_4 0.78539816339744830961f
#define RSPT_TAU 6.28318530717958647692f
#define RSPT_ONE_MINUS_EPS 0x1.fffffep-1f
#define RSPT_INF __builtin_huge_valf()
#define RSPT_FLT_MAX 3.402823466e+38f

struct f3 {
    float x, y, z;
};
struct f2 {
    float x, y;
};

RDEV f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
RDEV f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RDEV f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RDEV f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
RDEV f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
// Vector / Float multiplies by the reciprocal (geometry.rs:1261-1297)
RDEV f3 vdiv(f3 a, float s) {
    float inv = 1.0f / s;
    return f3{a.x * inv, a.y * inv, a.z * inv};
}
RDEV float comp(f3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
RDEV f3 vabs(f3 a) { return f3{fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
// ---- libm as the reference's host computes it: sinf, cosf, logf, log2f, expf, acosf, atanf, atan2f (glibc_libm.h) ----
#define GL_FN RDEV
#define GL_FN_COLD __device__ __noinline__
#define GL_TABLE static __device__ const
#define GL_F2U(x) __float_as_uint(x)
#define GL_U2F(x) __uint_as_float(x)
#define GL_D2U(x) ((uint64_t)__double_as_longlong(x))
#define GL_U2D(x) __longlong_as_double((long long)(x))
#include "glibc_libm.h"

RDEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // geometry.rs:630
RDEV float absdot(f3 a, f3 b) { return fabsf(dot(a, b)); }
RDEV float len2(f3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
RDEV float len(f3 a) { return sqrtf(len2(a)); }
RDEV f3 normalize(f3 a) { return vdiv(a, len(a)); }  // geometry.rs:412
RDEV float dist2(f3 a, f3 b) { return len2(a - b); }
// cross products are evaluated in f64 and rounded once (geometry.rs:680-709)
RDEV f3 cross(f3 a, f3 b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return f3{(float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx))};
}
RDEV float max3(float a, float b, float c) { return fmaxf(a, fmaxf(b, c)); }
RDEV f3 faceforward(f3 n, f3 v) { return dot(n, v) < 0.0f ? -n : n; }  // geometry.rs:1852-1858
// geometry.rs:779-794
RDEV void coordinate_system(f3 v1, f3* v2, f3* v3) {
    if (fabsf(v1.x) > fabsf(v1.y))
        *v2 = vdiv(f3{-v1.z, 0.0f, v1.x}, sqrtf(v1.x * v1.x + v1.z * v1.z));
    else
        *v2 = vdiv(f3{0.0f, v1.z, -v1.y}, sqrtf(v1.y * v1.y + v1.z * v1.z));
    *v3 = cross(v1, *v2);
}

// pbrt.rs:61-91: one ulp up / down through the bit pattern
RDEV float next_up(float v) {
    if (__builtin_isinf(v) && v > 0.0f) return v;
    if (v == 0.0f) v = 0.0f;  // -0 -> +0
    uint32_t u = __float_as_uint(v);
    u = (v >= 0.0f) ? u + 1u : u - 1u;
    return __uint_as_float(u);
}
RDEV float next_down(float v) {
    if (__builtin_isinf(v) && v < 0.0f) return v;
    if (v == 0.0f) v = -0.0f;
    uint32_t u = __float_as_uint(v);
    u = (v > 0.0f) ? u - 1u : u + 1u;
    return __uint_as_float(u);
}
// pbrt.rs:94-96
RDEV constexpr float gamma_n(int n) { return ((float)n * RSPT_MACHINE_EPS) / (1.0f - (float)n * RSPT_MACHINE_EPS); }
RDEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
RDEV float lerpf(float t, float a, float b) { return a * (1.0f - t) + b * t; }  // pbrt.rs:231-241

// Rust `as i32` on f32: saturating, NaN -> 0
RDEV int32_t f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)x;
}

// geometry.rs:1535-1557 pnt3_offset_ray_origin
RDEV f3 offset_ray_origin(f3 p, f3 p_error, f3 n, f3 w) {
    float d = dot(vabs(n), p_error);
    f3 off = n * d;
    if (dot(w, n) < 0.0f) off = -off;
    f3 po = p + off;
    if (off.x > 0.0f) po.x = next_up(po.x); else if (off.x < 0.0f) po.x = next_down(po.x);
    if (off.y > 0.0f) po.y = next_up(po.y); else if (off.y < 0.0f) po.y = next_down(po.y);
    if (off.z > 0.0f) po.z = next_up(po.z); else if (off.z < 0.0f) po.z = next_down(po.z);
    return po;
}

// RGBSpectrum (spectrum.rs:1528-1835) as a 3-float value
struct rgb {
    float r, g, b;
};
RDEV rgb mkrgb(float v) { return rgb{v, v, v}; }
RDEV rgb operator+(rgb a, rgb b) { return rgb{a.r + b.r, a.g + b.g, a.b + b.b}; }
RDEV rgb operator-(rgb a, rgb b) { return rgb{a.r - b.r, a.g - b.g, a.b - b.b}; }
RDEV rgb operator*(rgb a, rgb b) { return rgb{a.r * b.r, a.g * b.g, a.b * b.b}; }
RDEV rgb operator*(rgb a, float s) { return rgb{a.r * s, a.g * s, a.b * s}; }
RDEV rgb operator/(rgb a, rgb b) { return rgb{a.r / b.r, a.g / b.g, a.b / b.b}; }
RDEV rgb operator/(rgb a, float s) { return rgb{a.r / s, a.g / s, a.b / s}; }  // three true divisions (:1752)
RDEV rgb rsqrt3(rgb a) { return rgb{sqrtf(a.r), sqrtf(a.g), sqrtf(a.b)}; }
RDEV bool is_black(rgb a) { return !(a.r != 0.0f) && !(a.g != 0.0f) && !(a.b != 0.0f); }
RDEV bool has_nans(rgb a) { return a.r != a.r || a.g != a.g || a.b != a.b; }
RDEV float lum(rgb a) { return 0.212671f * a.r + 0.715160f * a.g + 0.072169f * a.b; }  // :1581
RDEV float maxc(rgb a) { return fmaxf(fmaxf(a.r, a.g), a.b); }
RDEV rgb ldrgb(const float* p) { return rgb{p[0], p[1], p[2]}; }

// transform.rs:490-527, 662-708 (row-major m[16])
RDEV f3 xf_point(const float* m, f3 p) {
    float xp = m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3];
    float yp = m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7];
    float zp = m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11];
    float wp = m[12] * p.x + m[13] * p.y + m[14] * p.z + m[15];
    if (wp == 1.0f) return f3{xp, yp, zp};
    float inv = 1.0f / wp;
    return f3{inv * xp, inv * yp, inv * zp};
}
RDEV f3 xf_vector(const float* m, f3 v) {
    return f3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
RDEV f3 xf_point_err(const float* m, f3 p, f3* err) {
    float xs = fabsf(m[0] * p.x) + fabsf(m[1] * p.y) + fabsf(m[2] * p.z) + fabsf(m[3]);
    float ys = fabsf(m[4] * p.x) + fabsf(m[5] * p.y) + fabsf(m[6] * p.z) + fabsf(m[7]);
    float zs = fabsf(m[8] * p.x) + fabsf(m[9] * p.y) + fabsf(m[10] * p.z) + fabsf(m[11]);
    *err = f3{xs, ys, zs} * gamma_n(3);
    return xf_point(m, p);
}

}  // namespace rspt
