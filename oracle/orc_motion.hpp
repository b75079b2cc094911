// TEST INFRASTRUCTURE — AnimatedTransform::motion_bounds / bound_point_motion / interval_find_zeros and the Interval type
// (src/core/transform.rs:2147-2350), restated; the checker of librspt's rspt_motion_bounds (tests/test_motion_bounds.py).
//
// The DerivativeTerm coefficients c1..c5 (:944-2030) are 1060 lines of machine-expanded polynomials in the reference (the derivative
// of T(t) + R(q(t)) S(t) p written out entry by entry).  This restatement does NOT repeat that text: it evaluates the same five vectors
// from the closed form they expand — R(q(t)) = A + B cos(2 theta t) + C sin(2 theta t) for q(t) = q1 cos(theta t) + qperp sin(theta t) —
// in plain f32.  What ties it to the reference's own expressions is tests/golden/motion_bounds.npz: oracle/make_motion_fixture.py
// machine-converts lines 944-2030 of transform.rs (read where they lie, compiled into oracle/_ref/, never committed) and records, for
// seeded key pairs, the coefficients and the boxes those literal expressions give; motion_bounds() below takes an optional coefficient
// table so that the fixture's boxes are produced by THIS bound_point_motion / interval_find_zeros around the reference's literal terms.
#pragma once
#include "orc_animated.hpp"

namespace orc {

struct Interval { // transform.rs:2207-2250
    Float low, high;
    Interval(Float v0, Float v1) : low(std::fmin(v0, v1)), high(std::fmax(v0, v1)) {}
};
static inline Interval operator+(Interval a, Interval b) { Interval r(0, 0); r.low = a.low + b.low; r.high = a.high + b.high; return r; }
static inline Interval operator*(Interval a, Interval b) {
    Float min_rhs_low = std::fmin(a.low * b.low, a.high * b.low), min_rhs_high = std::fmin(a.low * b.high, a.high * b.high);
    Float max_rhs_low = std::fmax(a.low * b.low, a.high * b.low), max_rhs_high = std::fmax(a.low * b.high, a.high * b.high);
    Interval r(0, 0);
    r.low = std::fmin(min_rhs_low, min_rhs_high); r.high = std::fmax(max_rhs_low, max_rhs_high);
    return r;
}
static const Float PI_F = 3.14159265358979323846f;
static inline Interval interval_sin(Interval i) { // :2236-2255 (asserts: i.low >= 0, i.high <= 2.0001 pi)
    Float sin_low = std::sin(i.low), sin_high = std::sin(i.high);
    if (sin_low > sin_high) std::swap(sin_low, sin_high);
    if (i.low < PI_F / 2.0f && i.high > PI_F / 2.0f) sin_high = 1.0f;
    if (i.low < (3.0f / 2.0f) * PI_F && i.high > (3.0f / 2.0f) * PI_F) sin_low = -1.0f;
    Interval r(0, 0); r.low = sin_low; r.high = sin_high; return r;
}
static inline Interval interval_cos(Interval i) { // :2257-2274
    Float cos_low = std::cos(i.low), cos_high = std::cos(i.high);
    if (cos_low > cos_high) std::swap(cos_low, cos_high);
    if (i.low < PI_F && i.high > PI_F) cos_low = -1.0f;
    Interval r(0, 0); r.low = cos_low; r.high = cos_high; return r;
}
// :2281-2350.  Returns false where the reference would panic (zeros[8] is indexed with 8)
static bool interval_find_zeros(Float c1, Float c2, Float c3, Float c4, Float c5, Float theta, Interval t_interval, Float zeros[8], int* zero_count, int depth) {
    Float two_theta = 2.0f * theta;
    Interval range = Interval(c1, c1) + (Interval(c2, c2) + Interval(c3, c3) * t_interval) * interval_cos(Interval(two_theta, two_theta) * t_interval) +
                     (Interval(c4, c4) + Interval(c5, c5) * t_interval) * interval_sin(Interval(two_theta, two_theta) * t_interval);
    if (range.low > 0.0f || range.high < 0.0f || range.low == range.high) return true;
    if (depth > 0) {
        Float mid = (t_interval.low + t_interval.high) * 0.5f;
        return interval_find_zeros(c1, c2, c3, c4, c5, theta, Interval(t_interval.low, mid), zeros, zero_count, depth - 1) &&
               interval_find_zeros(c1, c2, c3, c4, c5, theta, Interval(mid, t_interval.high), zeros, zero_count, depth - 1);
    }
    Float t_newton = (t_interval.low + t_interval.high) * 0.5f;
    for (int i = 0; i < 4; i++) {
        Float f_newton = c1 + (c2 + c3 * t_newton) * std::cos(2.0f * theta * t_newton) + (c4 + c5 * t_newton) * std::sin(2.0f * theta * t_newton);
        Float f_prime_newton = (c3 + 2.0f * (c4 + c5 * t_newton) * theta) * std::cos(2.0f * t_newton * theta) +
                               (c5 - 2.0f * (c2 + c3 * t_newton) * theta) * std::sin(2.0f * t_newton * theta);
        if (f_newton == 0.0f || f_prime_newton == 0.0f) break;
        t_newton -= f_newton / f_prime_newton;
    }
    if (t_newton >= t_interval.low - 1e-3f && t_newton < t_interval.high + 1e-3f) {
        if (*zero_count >= 8) return false;
        zeros[(*zero_count)++] = t_newton;
    }
    return true;
}

static inline Bounds3 bnd_union_pnt(const Bounds3& b, const V3& p) { return bunion(b, p); } // geometry.rs:2298-2313
static inline Bounds3 bnd_union(const Bounds3& a, const Bounds3& b) { return bunion(a, b); } // :2315-2327
static inline V3 m44_point(const M44& m, const V3& p) { return transform_point(&m.m[0][0], p); } // Transform::transform_point, transform.rs:490-516
static inline Bounds3 m44_bounds(const M44& m, const Bounds3& b) { // Transform::transform_bounds, :596-660 (corner order min, x, y, z, yz, xy, xz, max)
    V3 p = m44_point(m, V3{b.p_min.x, b.p_min.y, b.p_min.z});
    Bounds3 ret = bounds_from(p, p);
    ret = bnd_union_pnt(ret, m44_point(m, V3{b.p_max.x, b.p_min.y, b.p_min.z}));
    ret = bnd_union_pnt(ret, m44_point(m, V3{b.p_min.x, b.p_max.y, b.p_min.z}));
    ret = bnd_union_pnt(ret, m44_point(m, V3{b.p_min.x, b.p_min.y, b.p_max.z}));
    ret = bnd_union_pnt(ret, m44_point(m, V3{b.p_min.x, b.p_max.y, b.p_max.z}));
    ret = bnd_union_pnt(ret, m44_point(m, V3{b.p_max.x, b.p_max.y, b.p_min.z}));
    ret = bnd_union_pnt(ret, m44_point(m, V3{b.p_max.x, b.p_min.y, b.p_max.z}));
    ret = bnd_union_pnt(ret, m44_point(m, V3{b.p_max.x, b.p_max.y, b.p_max.z}));
    return ret;
}

struct MotionTerms { Float c[5][3][4]; Float theta; }; // c[n][component] = (kc, kx, ky, kz) of DerivativeTerm c(n+1)[component] (:880-891)

// The five coefficient vectors from the closed form (see the header).  All f32, left to right.
static MotionTerms motion_terms(const AnimatedTransform& at) {
    MotionTerms mt{};
    Float cos_theta = quat_dot(at.r[0], at.r[1]);
    mt.theta = std::acos(clamp_t(cos_theta, -1.0f, 1.0f)); // :936
    Quat qperp = quat_normalize(quat_sub(at.r[1], quat_scale(at.r[0], cos_theta))); // :937
    const Float a[4] = {at.r[0].v.x, at.r[0].v.y, at.r[0].v.z, at.r[0].w}, b[4] = {qperp.v.x, qperp.v.y, qperp.v.z, qperp.w};
    // a quaternion's rotation matrix is I + lin(products q_i q_j) (quat_to_matrix above); with q = a cos + b sin the products are
    // (a_i a_j + b_i b_j) / 2 + (a_i a_j - b_i b_j) / 2 cos(2 phi) + (a_i b_j + a_j b_i) / 2 sin(2 phi)
    Float A[3][3], B[3][3], C[3][3];
    for (int which = 0; which < 3; which++) {
        Float P[4][4];
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                P[i][j] = which == 0 ? 0.5f * (a[i] * a[j] + b[i] * b[j]) : which == 1 ? 0.5f * (a[i] * a[j] - b[i] * b[j]) : 0.5f * (a[i] * b[j] + a[j] * b[i]);
        Float (*M)[3] = which == 0 ? A : which == 1 ? B : C;
        const Float one = which == 0 ? 1.0f : 0.0f;
        M[0][0] = one - 2.0f * (P[1][1] + P[2][2]); M[0][1] = 2.0f * (P[0][1] - P[3][2]); M[0][2] = 2.0f * (P[0][2] + P[3][1]);
        M[1][0] = 2.0f * (P[0][1] + P[3][2]); M[1][1] = one - 2.0f * (P[0][0] + P[2][2]); M[1][2] = 2.0f * (P[1][2] - P[3][0]);
        M[2][0] = 2.0f * (P[0][2] - P[3][1]); M[2][1] = 2.0f * (P[1][2] + P[3][0]); M[2][2] = one - 2.0f * (P[0][0] + P[1][1]);
    }
    Float S0[3][3], dS[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { S0[i][j] = at.s[0].m[i][j]; dS[i][j] = at.s[1].m[i][j] - at.s[0].m[i][j]; }
    auto mm = [](const Float x[3][3], const Float y[3][3], int i, int j) { return x[i][0] * y[0][j] + x[i][1] * y[1][j] + x[i][2] * y[2][j]; };
    const Float t0[3] = {at.t[0].x, at.t[0].y, at.t[0].z}, t1[3] = {at.t[1].x, at.t[1].y, at.t[1].z};
    const Float th2 = 2.0f * mt.theta;
    for (int c = 0; c < 3; c++) {
        mt.c[0][c][0] = -t0[c] + t1[c];
        for (int j = 0; j < 3; j++) {
            mt.c[0][c][1 + j] = mm(A, dS, c, j);
            mt.c[1][c][1 + j] = mm(B, dS, c, j) + th2 * mm(C, S0, c, j);
            mt.c[2][c][1 + j] = th2 * mm(C, dS, c, j);
            mt.c[3][c][1 + j] = mm(C, dS, c, j) - th2 * mm(B, S0, c, j);
            mt.c[4][c][1 + j] = -th2 * mm(B, dS, c, j);
        }
    }
    return mt;
}

// AnimatedTransform::transform_point (:2125-2135)
static inline V3 animated_point(const AnimatedTransform& at, Float time, const V3& p) {
    if (!at.actually_animated || time <= at.start_time) return m44_point(at.start, p);
    if (time >= at.end_time) return m44_point(at.end, p);
    return m44_point(at.interpolate(time), p);
}
// :2164-2210
static bool bound_point_motion(const AnimatedTransform& at, const MotionTerms& mt, const V3& p, Bounds3* out) {
    if (!at.actually_animated) { V3 q = m44_point(at.start, p); *out = bounds_from(q, q); return true; }
    V3 ps = m44_point(at.start, p), pe = m44_point(at.end, p);
    Bounds3 bounds = bounds_from(ps, pe); // Bounds3f::new, geometry.rs:2014-2026
    Float cos_theta = quat_dot(at.r[0], at.r[1]);
    Float theta = std::acos(clamp_t(cos_theta, -1.0f, 1.0f));
    for (int c = 0; c < 3; c++) {
        Float zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int n_zeros = 0;
        Float cn[5];
        for (int n = 0; n < 5; n++) cn[n] = mt.c[n][c][0] + mt.c[n][c][1] * p.x + mt.c[n][c][2] * p.y + mt.c[n][c][3] * p.z; // DerivativeTerm::eval
        if (!interval_find_zeros(cn[0], cn[1], cn[2], cn[3], cn[4], theta, Interval(0.0f, 1.0f), zeros, &n_zeros, 8)) return false;
        for (int i = 0; i < 8; i++) { // `for item in &zeros`: all eight slots, the unused ones hold 0.0 (= the start point)
            V3 pz = animated_point(at, lerp(zeros[i], at.start_time, at.end_time), p);
            bounds = bnd_union_pnt(bounds, pz);
        }
    }
    *out = bounds;
    return true;
}
// :2147-2163.  terms: nullptr = the closed form above; else a coefficient table (the fixture's, from the reference's literal expressions)
static bool motion_bounds(const AnimatedTransform& at, const Bounds3& b, const MotionTerms* terms, Bounds3* out) {
    if (!at.actually_animated) { *out = m44_bounds(at.start, b); return true; }
    if (!at.has_rotation) { *out = bnd_union(m44_bounds(at.start, b), m44_bounds(at.end, b)); return true; }
    const MotionTerms mt = terms ? *terms : motion_terms(at);
    const Float big = std::numeric_limits<Float>::max();
    Bounds3 bounds; bounds.p_min = V3{big, big, big}; bounds.p_max = V3{-big, -big, -big}; // Bounds3f::default, geometry.rs:1993-2011
    for (int corner = 0; corner < 8; corner++) {
        V3 p{(corner & 1) ? b.p_max.x : b.p_min.x, (corner & 2) ? b.p_max.y : b.p_min.y, (corner & 4) ? b.p_max.z : b.p_min.z}; // Bounds3f::corner, :2027-2046
        Bounds3 pb;
        if (!bound_point_motion(at, mt, p, &pb)) return false;
        bounds = bnd_union(bounds, pb);
    }
    *out = bounds;
    return true;
}

} // namespace orc
