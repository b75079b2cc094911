// TEST INFRASTRUCTURE — CPU oracle for the rs_pbrt path-tracing hot path.
//
// A C++17 *restatement* of wahn/rs_pbrt v0.9.12's arithmetic (f32, no FMA contraction:
// build with -ffp-contract=off), function by function, each citing the reference
// file:line it follows (paths relative to the rs_pbrt tree).  PARITY UNPINNED: the
// reference has no Rust toolchain here, no golden vectors and no numeric tests
// (SURVEY.md §8c); this oracle is pinned by first-principles known-answer tests in
// tests/ instead.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may use anything under oracle/.  The product (rs_pbrt_amd/) never links or calls it.
#pragma once
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace orc {

typedef float Float; // src/core/pbrt.rs:14

// src/core/pbrt.rs:16-23
static const Float MACHINE_EPSILON = FLT_EPSILON * 0.5f;
static const Float SHADOW_EPSILON = 0.0001f;
static const Float PI = 3.14159265358979323846f; // std::f32::consts::PI
static const Float INV_PI = 0.31830988618379067154f;
static const Float INV_2_PI = 0.15915494309189533577f;
static const Float INV_4_PI = 0.07957747154594766788f; // pbrt.rs:20
static const Float PI_OVER_2 = 1.57079632679489661923f;
static const Float PI_OVER_4 = 0.78539816339744830961f;
// src/core/rng.rs:13
static const Float FLOAT_ONE_MINUS_EPSILON = 0x1.fffffep-1f;
static const Float INF = std::numeric_limits<float>::infinity();

// Rust `x as i32` from f32: saturating, NaN -> 0, truncation
static inline int32_t f2i(Float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
// Rust `x as isize` from f32
static inline int64_t f2i64(Float x) {
    if (x != x) return 0;
    if (x >= 9223372036854775808.0f) return INT64_MAX;
    if (x <= -9223372036854775808.0f) return INT64_MIN;
    return (int64_t)x;
}
// Rust `x as usize` from f32
static inline uint64_t f2usize(Float x) {
    if (x != x || x <= 0.0f) return 0;
    if (x >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)x;
}
// Rust `x as u32` from f32
static inline uint32_t f2u32(Float x) {
    if (x != x || x <= 0.0f) return 0;
    if (x >= 4294967296.0f) return UINT32_MAX;
    return (uint32_t)x;
}
// Rust `x as u8` from f32
static inline uint8_t f2u8(Float x) {
    if (x != x || x <= 0.0f) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}

// src/core/pbrt.rs:29-56
static inline uint32_t float_to_bits(Float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline Float bits_to_float(uint32_t u) { Float f; std::memcpy(&f, &u, 4); return f; }

// src/core/pbrt.rs:61-74
static inline Float next_float_up(Float v) {
    if (std::isinf(v) && v > 0.0f) return v;
    Float nv = (v == -0.0f) ? 0.0f : v;
    uint32_t ui = float_to_bits(nv);
    if (nv >= 0.0f) ui += 1; else ui -= 1;
    return bits_to_float(ui);
}
// src/core/pbrt.rs:78-91
static inline Float next_float_down(Float v) {
    if (std::isinf(v) && v < 0.0f) return v;
    Float nv = (v == 0.0f) ? -0.0f : v;
    uint32_t ui = float_to_bits(nv);
    if (nv > 0.0f) ui -= 1; else ui += 1;
    return bits_to_float(ui);
}
// src/core/pbrt.rs:94-96
static inline Float gamma(int n) {
    return ((Float)n * MACHINE_EPSILON) / (1.0f - (Float)n * MACHINE_EPSILON);
}
// src/core/pbrt.rs:108-122
template <class T> static inline T clamp_t(T v, T lo, T hi) { return v < lo ? lo : (v > hi ? hi : v); }
// src/core/pbrt.rs:231-241  a*(1-t) + b*t
static inline Float lerp(Float t, Float a, Float b) { return a * (1.0f - t) + b * t; }

// One 3-float type for Point3f / Vector3f / Normal3f (src/core/geometry.rs:387,1018,1599)
struct V3 {
    Float x, y, z;
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
struct P2 { Float x, y; };

static inline V3 v3(Float x, Float y, Float z) { return V3{x, y, z}; }
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
static inline V3 operator*(V3 a, Float b) { return V3{a.x * b, a.y * b, a.z * b}; } // geometry.rs:1237-1259
// geometry.rs:1261-1297: division multiplies by the reciprocal
static inline V3 operator/(V3 a, Float b) { Float inv = 1.0f / b; return V3{a.x * inv, a.y * inv, a.z * inv}; }
static inline V3 vabs(V3 a) { return V3{std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }
static inline Float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // geometry.rs:630
static inline Float abs_dot(V3 a, V3 b) { return std::fabs(dot(a, b)); }
static inline Float length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; } // geometry.rs:404
static inline Float length(V3 a) { return std::sqrt(length_squared(a)); }
static inline V3 normalize(V3 a) { return a / length(a); } // geometry.rs:412
// geometry.rs:680-709: cross product in f64, rounded once
static inline V3 cross(V3 a, V3 b) {
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return V3{(Float)((ay * bz) - (az * by)), (Float)((az * bx) - (ax * bz)), (Float)((ax * by) - (ay * bx))};
}
static inline Float max_component(V3 v) { return std::fmax(v.x, std::fmax(v.y, v.z)); } // geometry.rs:711
// geometry.rs:721-733
static inline int max_dimension(V3 v) { return v.x > v.y ? (v.x > v.z ? 0 : 2) : (v.y > v.z ? 1 : 2); }
static inline V3 permute(V3 v, int x, int y, int z) { return V3{v[x], v[y], v[z]}; }
// geometry.rs:779-794
static inline void coordinate_system(V3 v1, V3* v2, V3* v3_) {
    if (std::fabs(v1.x) > std::fabs(v1.y))
        *v2 = V3{-v1.z, 0.0f, v1.x} / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else
        *v2 = V3{0.0f, v1.z, -v1.y} / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    *v3_ = cross(v1, *v2);
}
// geometry.rs:1852-1858
static inline V3 faceforward(V3 n, V3 v) { return dot(n, v) < 0.0f ? -n : n; }
static inline Float distance_squared(V3 a, V3 b) { return length_squared(a - b); }

// geometry.rs:1535-1557
static inline V3 offset_ray_origin(V3 p, V3 p_error, V3 n, V3 w) {
    Float d = dot(vabs(n), p_error);
    V3 offset = n * d;
    if (dot(w, n) < 0.0f) offset = -offset;
    V3 po = p + offset;
    for (int i = 0; i < 3; i++) {
        if (offset[i] > 0.0f) po.at(i) = next_float_up(po[i]);
        else if (offset[i] < 0.0f) po.at(i) = next_float_down(po[i]);
    }
    return po;
}

// geometry.rs:1980-2090
struct Bounds3 {
    V3 p_min{FLT_MAX, FLT_MAX, FLT_MAX};    // Default: inverted box of +-f32::MAX (geometry.rs:1993)
    V3 p_max{-FLT_MAX, -FLT_MAX, -FLT_MAX};
    V3 diagonal() const { return p_max - p_min; }
    Float surface_area() const {
        V3 d = diagonal();
        Float r = d.x * d.y + d.x * d.z + d.y * d.z;
        return r + r;
    }
    int maximum_extent() const {
        V3 d = diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        else if (d.y > d.z) return 1;
        return 2;
    }
    V3 offset(V3 p) const {
        V3 o = p - p_min;
        if (p_max.x > p_min.x) o.x /= p_max.x - p_min.x;
        if (p_max.y > p_min.y) o.y /= p_max.y - p_min.y;
        if (p_max.z > p_min.z) o.z /= p_max.z - p_min.z;
        return o;
    }
    V3 lerp3(V3 t) const { return V3{lerp(t.x, p_min.x, p_max.x), lerp(t.y, p_min.y, p_max.y), lerp(t.z, p_min.z, p_max.z)}; }
};
static inline Bounds3 bounds_from(V3 a, V3 b) { // geometry.rs:2014
    Bounds3 r;
    r.p_min = V3{std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)};
    r.p_max = V3{std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)};
    return r;
}
static inline Bounds3 bunion(const Bounds3& b, V3 p) { // geometry.rs:2299
    Bounds3 r;
    r.p_min = V3{std::fmin(b.p_min.x, p.x), std::fmin(b.p_min.y, p.y), std::fmin(b.p_min.z, p.z)};
    r.p_max = V3{std::fmax(b.p_max.x, p.x), std::fmax(b.p_max.y, p.y), std::fmax(b.p_max.z, p.z)};
    return r;
}
static inline Bounds3 bunion(const Bounds3& a, const Bounds3& b) { // geometry.rs:2315
    Bounds3 r;
    r.p_min = V3{std::fmin(a.p_min.x, b.p_min.x), std::fmin(a.p_min.y, b.p_min.y), std::fmin(a.p_min.z, b.p_min.z)};
    r.p_max = V3{std::fmax(a.p_max.x, b.p_max.x), std::fmax(a.p_max.y, b.p_max.y), std::fmax(a.p_max.z, b.p_max.z)};
    return r;
}

// src/core/geometry.rs:2378-2390 (medium / differential omitted: the path integrator with
// constant textures never reads them, interaction.rs:388-474)
struct Ray {
    V3 o, d;
    mutable Float t_max; // Cell<Float>
    Float time;
    // Option<RayDifferential> (geometry.rs:2408-2414): camera rays only
    bool has_diff = false;
    V3 rx_o{0, 0, 0}, ry_o{0, 0, 0}, rx_d{0, 0, 0}, ry_d{0, 0, 0};
    uint32_t medium = 0; // Option<Arc<Medium>>: 0 = None, else 1 + index into rspt_scene_desc.media
    void scale_differentials(Float s) { // geometry.rs:2398-2405
        if (!has_diff) return;
        rx_o = o + (rx_o - o) * s; ry_o = o + (ry_o - o) * s;
        rx_d = d + (rx_d - d) * s; ry_d = d + (ry_d - d) * s;
    }
};

// src/core/spectrum.rs:1528-1835
struct Spec {
    Float c[3];
    Spec() : c{0, 0, 0} {}
    explicit Spec(Float v) : c{v, v, v} {}
    Spec(Float r, Float g, Float b) : c{r, g, b} {}
    bool is_black() const { return !(c[0] != 0.0f) && !(c[1] != 0.0f) && !(c[2] != 0.0f); }
    bool has_nans() const { return c[0] != c[0] || c[1] != c[1] || c[2] != c[2]; }
    Float y() const { return 0.212671f * c[0] + 0.715160f * c[1] + 0.072169f * c[2]; } // :1581
    Float max_component_value() const { return std::fmax(std::fmax(c[0], c[1]), c[2]); } // :1635
};
static inline Spec operator+(Spec a, Spec b) { return Spec(a.c[0] + b.c[0], a.c[1] + b.c[1], a.c[2] + b.c[2]); }
static inline Spec operator-(Spec a, Spec b) { return Spec(a.c[0] - b.c[0], a.c[1] - b.c[1], a.c[2] - b.c[2]); }
static inline Spec operator*(Spec a, Spec b) { return Spec(a.c[0] * b.c[0], a.c[1] * b.c[1], a.c[2] * b.c[2]); }
static inline Spec operator*(Spec a, Float b) { return Spec(a.c[0] * b, a.c[1] * b, a.c[2] * b); }
static inline Spec operator/(Spec a, Spec b) { return Spec(a.c[0] / b.c[0], a.c[1] / b.c[1], a.c[2] / b.c[2]); }
// spectrum.rs:1752-1763: three true divisions
static inline Spec operator/(Spec a, Float b) { return Spec(a.c[0] / b, a.c[1] / b, a.c[2] / b); }
static inline Spec ssqrt(Spec a) { return Spec(std::sqrt(a.c[0]), std::sqrt(a.c[1]), std::sqrt(a.c[2])); }
// spectrum.rs:1829-1835
static inline void rgb_to_xyz(const Float rgb[3], Float xyz[3]) {
    xyz[0] = 0.412453f * rgb[0] + 0.357580f * rgb[1] + 0.180423f * rgb[2];
    xyz[1] = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
    xyz[2] = 0.019334f * rgb[0] + 0.119193f * rgb[1] + 0.950227f * rgb[2];
}
// spectrum.rs:1822-1826
static inline void xyz_to_rgb(const Float xyz[3], Float rgb[3]) {
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}

// src/core/transform.rs:490-516 (row-major m[16])
static inline V3 transform_point(const Float* m, V3 p) {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    Float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    Float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    Float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    if (wp == 1.0f) return V3{xp, yp, zp};
    Float inv = 1.0f / wp;
    return V3{inv * xp, inv * yp, inv * zp};
}
// transform.rs:518-527
static inline V3 transform_vector(const Float* m, V3 v) {
    return V3{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z,
              m[8] * v.x + m[9] * v.y + m[10] * v.z};
}
// transform.rs:662-708
static inline V3 transform_point_with_error(const Float* m, V3 p, V3* p_error) {
    Float x = p.x, y = p.y, z = p.z;
    Float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    Float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    Float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    Float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    Float xs = std::fabs(m[0] * x) + std::fabs(m[1] * y) + std::fabs(m[2] * z) + std::fabs(m[3]);
    Float ys = std::fabs(m[4] * x) + std::fabs(m[5] * y) + std::fabs(m[6] * z) + std::fabs(m[7]);
    Float zs = std::fabs(m[8] * x) + std::fabs(m[9] * y) + std::fabs(m[10] * z) + std::fabs(m[11]);
    *p_error = V3{xs, ys, zs} * gamma(3);
    if (wp == 1.0f) return V3{xp, yp, zp};
    Float inv = 1.0f / wp;
    return V3{inv * xp, inv * yp, inv * zp};
}
// transform.rs:538-595
static inline Ray transform_ray(const Float* m, const Ray& r) {
    V3 o_error;
    V3 o = transform_point_with_error(m, r.o, &o_error);
    V3 d = transform_vector(m, r.d);
    Float ls = length_squared(d);
    Float t_max = r.t_max;
    if (ls > 0.0f) {
        Float dt = dot(vabs(d), o_error) / ls;
        o = o + d * dt;
        t_max -= dt;
    }
    Ray out{o, d, t_max, r.time};
    out.medium = r.medium; // :591
    if (r.has_diff) { // :550-556: plain transform_point / transform_vector
        out.has_diff = true;
        out.rx_o = transform_point(m, r.rx_o); out.ry_o = transform_point(m, r.ry_o);
        out.rx_d = transform_vector(m, r.rx_d); out.ry_d = transform_vector(m, r.ry_d);
    }
    return out;
}

} // namespace orc
