// Host side of a moving instance's world bound: AnimatedTransform::motion_bounds / bound_point_motion
// (src/core/transform.rs:2147-2210) — the box a TransformedPrimitive hands the top-level BVH (primitive.rs:212-215).
//
// A point p under the interpolated transform is x(t) = T(t) + R(q(t)) S(t) p, t in [0, 1], with T and S interpolated linearly and
// q(t) = q0 cos(theta t) + qperp sin(theta t) (quat_slerp).  R is quadratic in q, so R(q(t)) = A + B cos(2 theta t) + C sin(2 theta t)
// and every component of the velocity has the shape the reference's interval_find_zeros (:2281-2350) looks for zeros of:
//     x'(t) = c1 + (c2 + c3 t) cos(2 theta t) + (c4 + c5 t) sin(2 theta t),
//     c1 = dT + A dS p,  c2 = B dS p + 2 theta C S0 p,  c3 = 2 theta C dS p,  c4 = C dS p - 2 theta B S0 p,  c5 = -2 theta B dS p.
// The reference carries these five as fully expanded polynomials in the quaternion and scale entries (DerivativeTerm c1..c5, :944-2030);
// here they are formed from the three matrices A, B, C in double precision and rounded once — the same numbers up to the rounding of the
// expanded f32 sums.  The zeros only decide WHERE the motion is sampled: the bound itself is the union of transform_point at those times
// (:2199-2207), and a velocity zero is where the position is least sensitive to its time, so the boxes agree with the expanded form's to
// the last bits (tests/test_motion_bounds.py holds both against fixtures made from the reference's own expressions).
// Root isolation (interval arithmetic, eight bisection levels, four Newton steps, the +-1e-3 acceptance window, the union over all eight
// slots of the zeros array whether filled or not) follows the reference operation by operation in f32.
#pragma once
#include <cmath>

#include "camera_anim.h"

namespace rspt {
#define RSPT_MOTION_PAD 1e-6f   // outward slack of the velocity-zero edges, relative to the box's largest extent (motion_bounds below)
namespace motion {

struct Box { float lo[3], hi[3]; };
struct Iv { float lo, hi; };                                       // Interval (:2207-2250)
inline Iv iv(float a, float b) { return Iv{fminf(a, b), fmaxf(a, b)}; }
inline Iv iv_add(Iv a, Iv b) { return Iv{a.lo + b.lo, a.hi + b.hi}; }
inline Iv iv_mul(Iv a, Iv b) {
    const float ll = a.lo * b.lo, hl = a.hi * b.lo, lh = a.lo * b.hi, hh = a.hi * b.hi;
    return Iv{fminf(fminf(ll, hl), fminf(lh, hh)), fmaxf(fmaxf(ll, hl), fmaxf(lh, hh))};
}
constexpr float kPi = 3.14159265358979323846f;
inline Iv iv_sin(Iv i) {                                           // interval_sin (:2236-2255); its asserts (0 <= i <= 2.0001 pi) hold for 2 theta t
    float lo = sinf(i.lo), hi = sinf(i.hi);
    if (lo > hi) { const float t = lo; lo = hi; hi = t; }
    if (i.lo < kPi / 2.0f && i.hi > kPi / 2.0f) hi = 1.0f;
    if (i.lo < (3.0f / 2.0f) * kPi && i.hi > (3.0f / 2.0f) * kPi) lo = -1.0f;
    return Iv{lo, hi};
}
inline Iv iv_cos(Iv i) {                                           // interval_cos (:2257-2274)
    float lo = cosf(i.lo), hi = cosf(i.hi);
    if (lo > hi) { const float t = lo; lo = hi; hi = t; }
    if (i.lo < kPi && i.hi > kPi) lo = -1.0f;
    return Iv{lo, hi};
}
// interval_find_zeros (:2281-2350).  zeros has 8 slots as in the reference; a ninth zero is where the reference would index out of
// bounds and panic — reported through *overflow instead (rspt_motion_bounds turns it into RSPT_E_UNSUPPORTED).
inline void find_zeros(float c1, float c2, float c3, float c4, float c5, float theta, Iv t, float zeros[8], int* n, int depth, bool* overflow) {
    const float two_theta = 2.0f * theta;
    const Iv ang = iv_mul(iv(two_theta, two_theta), t);
    const Iv range = iv_add(iv_add(iv(c1, c1), iv_mul(iv_add(iv(c2, c2), iv_mul(iv(c3, c3), t)), iv_cos(ang))),
                            iv_mul(iv_add(iv(c4, c4), iv_mul(iv(c5, c5), t)), iv_sin(ang)));
    if (range.lo > 0.0f || range.hi < 0.0f || range.lo == range.hi) return;
    if (depth > 0) {
        const float mid = (t.lo + t.hi) * 0.5f;
        find_zeros(c1, c2, c3, c4, c5, theta, iv(t.lo, mid), zeros, n, depth - 1, overflow);
        find_zeros(c1, c2, c3, c4, c5, theta, iv(mid, t.hi), zeros, n, depth - 1, overflow);
        return;
    }
    float tn = (t.lo + t.hi) * 0.5f;
    for (int i = 0; i < 4; i++) {
        const float f = c1 + (c2 + c3 * tn) * cosf(2.0f * theta * tn) + (c4 + c5 * tn) * sinf(2.0f * theta * tn);
        const float fp = (c3 + 2.0f * (c4 + c5 * tn) * theta) * cosf(2.0f * tn * theta) + (c5 - 2.0f * (c2 + c3 * tn) * theta) * sinf(2.0f * tn * theta);
        if (f == 0.0f || fp == 0.0f) break;
        tn -= f / fp;
    }
    if (tn >= t.lo - 1e-3f && tn < t.hi + 1e-3f) {
        if (*n >= 8) { *overflow = true; return; }
        zeros[(*n)++] = tn;
    }
}

// Transform::transform_point (:490-516)
inline void xf_point(const camanim::M4& m, const float p[3], float out[3]) {
    const float x = p[0], y = p[1], z = p[2];
    const float xp = m.m[0][0] * x + m.m[0][1] * y + m.m[0][2] * z + m.m[0][3];
    const float yp = m.m[1][0] * x + m.m[1][1] * y + m.m[1][2] * z + m.m[1][3];
    const float zp = m.m[2][0] * x + m.m[2][1] * y + m.m[2][2] * z + m.m[2][3];
    const float wp = m.m[3][0] * x + m.m[3][1] * y + m.m[3][2] * z + m.m[3][3];
    if (wp == 1.0f) { out[0] = xp; out[1] = yp; out[2] = zp; return; }
    const float inv = 1.0f / wp;
    out[0] = inv * xp; out[1] = inv * yp; out[2] = inv * zp;
}
inline void box_add(Box* b, const float p[3]) { for (int i = 0; i < 3; i++) { b->lo[i] = fminf(b->lo[i], p[i]); b->hi[i] = fmaxf(b->hi[i], p[i]); } }
inline void box_join(Box* b, const Box& o) { for (int i = 0; i < 3; i++) { b->lo[i] = fminf(b->lo[i], o.lo[i]); b->hi[i] = fmaxf(b->hi[i], o.hi[i]); } }
// Transform::transform_bounds (:596-660): the corners in the reference's order (min, x, y, z, yz, xy, xz, max)
inline Box xf_bounds(const camanim::M4& m, const Box& b) {
    static const int order[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 1, 1}, {1, 1, 0}, {1, 0, 1}, {1, 1, 1}};
    Box r{};
    for (int c = 0; c < 8; c++) {
        const float p[3] = {order[c][0] ? b.hi[0] : b.lo[0], order[c][1] ? b.hi[1] : b.lo[1], order[c][2] ? b.hi[2] : b.lo[2]};
        float q[3];
        xf_point(m, p, q);
        if (c == 0) for (int i = 0; i < 3; i++) r.lo[i] = r.hi[i] = q[i];
        else box_add(&r, q);
    }
    return r;
}

struct Keys {                       // AnimatedTransform::new (:912-943)
    camanim::M4 start, end;
    float t0, t1;
    bool animated, has_rotation;
    CamAnim trs;                    // t, r (second one on the shorter arc), s
    float theta;
    float kc[5][3], k[5][3][3];     // c_n[component]: kc + k . p
};

// AnimatedTransform::interpolate (:2081-2113), the forward matrix only: translate(trans) * rotate.to_transform() * scale
inline camanim::M4 interpolate(const Keys& a, float time) {
    if (!a.animated || time <= a.t0) return a.start;
    if (time >= a.t1) return a.end;
    const float dt = (time - a.t0) / (a.t1 - a.t0);
    float trans[3];
    for (int i = 0; i < 3; i++) trans[i] = a.trs.t[0][i] * (1.0f - dt) + a.trs.t[1][i] * dt;
    const float* q1 = a.trs.r[0];
    const float* q2 = a.trs.r[1];
    const float cos_theta = (q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2]) + q1[3] * q2[3];
    float q[4], u[4];
    if (cos_theta > 0.9995f) {      // quat_slerp (quaternion.rs:168-180)
        for (int i = 0; i < 4; i++) u[i] = q1[i] * (1.0f - dt) + q2[i] * dt;
        const float n = sqrtf((u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) + u[3] * u[3]), inv = 1.0f / n;
        q[0] = u[0] * inv; q[1] = u[1] * inv; q[2] = u[2] * inv; q[3] = u[3] / n;
    } else {
        const float theta = acosf(cos_theta < -1.0f ? -1.0f : (cos_theta > 1.0f ? 1.0f : cos_theta));
        const float thetap = theta * dt;
        for (int i = 0; i < 4; i++) u[i] = q2[i] - q1[i] * cos_theta;
        const float n = sqrtf((u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) + u[3] * u[3]), inv = 1.0f / n;
        const float qp[4] = {u[0] * inv, u[1] * inv, u[2] * inv, u[3] / n};
        const float c = cosf(thetap), sn = sinf(thetap);
        for (int i = 0; i < 4; i++) q[i] = q1[i] * c + qp[i] * sn;
    }
    const float xx = q[0] * q[0], yy = q[1] * q[1], zz = q[2] * q[2], xy = q[0] * q[1], xz = q[0] * q[2], yz = q[1] * q[2];
    const float wx = q[0] * q[3], wy = q[1] * q[3], wz = q[2] * q[3];
    camanim::M4 rot = camanim::identity(), tr = camanim::identity(), sc = camanim::identity();
    rot.m[0][0] = 1.0f - 2.0f * (yy + zz); rot.m[0][1] = 2.0f * (xy - wz); rot.m[0][2] = 2.0f * (xz + wy);
    rot.m[1][0] = 2.0f * (xy + wz); rot.m[1][1] = 1.0f - 2.0f * (xx + zz); rot.m[1][2] = 2.0f * (yz - wx);
    rot.m[2][0] = 2.0f * (xz - wy); rot.m[2][1] = 2.0f * (yz + wx); rot.m[2][2] = 1.0f - 2.0f * (xx + yy);
    for (int i = 0; i < 3; i++) {
        tr.m[i][3] = trans[i];
        for (int j = 0; j < 3; j++) sc.m[i][j] = (1.0f - dt) * a.trs.s[0][4 * i + j] + dt * a.trs.s[1][4 * i + j];   // pbrt.rs lerp
    }
    return camanim::mul(camanim::mul(tr, rot), sc);
}

// the rotation part of R(q) = I + lin(P) for the symmetrised products P_ij = (a_i b_j + a_j b_i) / 2 of two quaternions (x, y, z, w)
inline void rot_linear(const double a[4], const double b[4], double out[3][3]) {
    auto P = [&](int i, int j) { return 0.5 * (a[i] * b[j] + a[j] * b[i]); };
    enum { X, Y, Z, W };
    out[0][0] = -2.0 * (P(Y, Y) + P(Z, Z)); out[0][1] = 2.0 * (P(X, Y) - P(W, Z)); out[0][2] = 2.0 * (P(X, Z) + P(W, Y));
    out[1][0] = 2.0 * (P(X, Y) + P(W, Z)); out[1][1] = -2.0 * (P(X, X) + P(Z, Z)); out[1][2] = 2.0 * (P(Y, Z) - P(W, X));
    out[2][0] = 2.0 * (P(X, Z) - P(W, Y)); out[2][1] = 2.0 * (P(Y, Z) + P(W, X)); out[2][2] = -2.0 * (P(X, X) + P(Y, Y));
}

inline void make_keys(const float start[16], float t0, const float end[16], float t1, Keys* k) {
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { k->start.m[i][j] = start[4 * i + j]; k->end.m[i][j] = end[4 * i + j]; }
    k->t0 = t0; k->t1 = t1;
    k->has_rotation = false; k->theta = 0.0f;
    k->animated = camanim::camera_keys(start, t0, end, t1, &k->trs);      // decompose both keys, flip r[1] onto the shorter arc
    if (!k->animated) return;
    const float* r0 = k->trs.r[0];
    const float* r1 = k->trs.r[1];
    const float cos_theta = (r0[0] * r1[0] + r0[1] * r1[1] + r0[2] * r1[2]) + r0[3] * r1[3];
    k->has_rotation = cos_theta < 0.9995f;                                // (:932)
    if (!k->has_rotation) return;
    k->theta = acosf(cos_theta < -1.0f ? -1.0f : (cos_theta > 1.0f ? 1.0f : cos_theta));
    // qperp = normalize(r1 - r0 cos_theta) in f32, as quat_slerp and the reference's terms (:938) form it
    float u[4];
    for (int i = 0; i < 4; i++) u[i] = r1[i] - r0[i] * cos_theta;
    const float n = sqrtf((u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) + u[3] * u[3]), inv = 1.0f / n;
    const double a[4] = {r0[0], r0[1], r0[2], r0[3]}, b[4] = {u[0] * inv, u[1] * inv, u[2] * inv, u[3] / n};
    double Laa[3][3], Lbb[3][3], Lab[3][3], A[3][3], B[3][3], Cm[3][3], S0[3][3], dS[3][3];
    rot_linear(a, a, Laa); rot_linear(b, b, Lbb); rot_linear(a, b, Lab);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            A[i][j] = (i == j ? 1.0 : 0.0) + 0.5 * (Laa[i][j] + Lbb[i][j]);
            B[i][j] = 0.5 * (Laa[i][j] - Lbb[i][j]);
            Cm[i][j] = Lab[i][j];
            S0[i][j] = k->trs.s[0][4 * i + j];
            dS[i][j] = (double)k->trs.s[1][4 * i + j] - (double)k->trs.s[0][4 * i + j];
        }
    const double th2 = 2.0 * (double)k->theta;
    auto mm = [](const double x[3][3], const double y[3][3], int i, int j) { return x[i][0] * y[0][j] + x[i][1] * y[1][j] + x[i][2] * y[2][j]; };
    for (int c = 0; c < 3; c++) {
        k->kc[0][c] = (float)((double)k->trs.t[1][c] - (double)k->trs.t[0][c]);
        for (int n5 = 1; n5 < 5; n5++) k->kc[n5][c] = 0.0f;
        for (int j = 0; j < 3; j++) {
            k->k[0][c][j] = (float)mm(A, dS, c, j);
            k->k[1][c][j] = (float)(mm(B, dS, c, j) + th2 * mm(Cm, S0, c, j));
            k->k[2][c][j] = (float)(th2 * mm(Cm, dS, c, j));
            k->k[3][c][j] = (float)(mm(Cm, dS, c, j) - th2 * mm(B, S0, c, j));
            k->k[4][c][j] = (float)(-th2 * mm(B, dS, c, j));
        }
    }
}

// AnimatedTransform::bound_point_motion (:2164-2210)
// ends (optional): the box of the point at the two keys alone — plain f32 transform_point, bit-identical to the reference's; what the velocity zeros add beyond it is
// located from coefficient tables that differ from the reference's expanded sums in their last bits (motion_bounds below pads exactly those edges)
inline Box bound_point_motion(const Keys& a, const float p[3], bool* overflow, Box* ends = nullptr) {
    Box b;
    float ps[3], pe[3];
    xf_point(a.start, p, ps);
    if (!a.animated) { for (int i = 0; i < 3; i++) b.lo[i] = b.hi[i] = ps[i]; if (ends) *ends = b; return b; }
    xf_point(a.end, p, pe);
    for (int i = 0; i < 3; i++) { b.lo[i] = fminf(ps[i], pe[i]); b.hi[i] = fmaxf(ps[i], pe[i]); }    // Bounds3f::new orders its corners
    if (ends) *ends = b;
    for (int c = 0; c < 3; c++) {
        float cn[5];
        for (int n5 = 0; n5 < 5; n5++) cn[n5] = a.kc[n5][c] + a.k[n5][c][0] * p[0] + a.k[n5][c][1] * p[1] + a.k[n5][c][2] * p[2];   // DerivativeTerm::eval (:888-890)
        float zeros[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int nz = 0;
        find_zeros(cn[0], cn[1], cn[2], cn[3], cn[4], a.theta, iv(0.0f, 1.0f), zeros, &nz, 8, overflow);
        for (int z = 0; z < 8; z++) {       // every slot, filled or not (`for item in &zeros`, :2199): an empty one is t = 0, the start point again
            const float time = (1.0f - zeros[z]) * a.t0 + zeros[z] * a.t1;
            float pz[3];
            xf_point(interpolate(a, time), p, pz);
            box_add(&b, pz);
        }
    }
    return b;
}

// AnimatedTransform::motion_bounds (:2147-2163)
inline Box motion_bounds(const Keys& a, const Box& b, bool* overflow) {
    if (!a.animated) return xf_bounds(a.start, b);
    if (!a.has_rotation) { Box r = xf_bounds(a.start, b); box_join(&r, xf_bounds(a.end, b)); return r; }
    Box r;                                   // Bounds3f::default() (geometry.rs:1993-2011): p_min = f32::MAX, p_max = f32::MIN
    for (int i = 0; i < 3; i++) { r.lo[i] = 3.40282347e38f; r.hi[i] = -3.40282347e38f; }
    Box e = r;                               // the same union over the eight corners at the two keys only
    for (int corner = 0; corner < 8; corner++) {
        const float p[3] = {(corner & 1) ? b.hi[0] : b.lo[0], (corner & 2) ? b.hi[1] : b.lo[1], (corner & 4) ? b.hi[2] : b.lo[2]};
        Box ends;
        box_join(&r, bound_point_motion(a, p, overflow, &ends));
        box_join(&e, ends);
    }
    // A BOUND must err outward.  Edges set by a key position are the reference's bit for bit.  An edge pushed out by a velocity zero
    // is transform_point at a time found by Newton steps on c1..c5, which this file forms in double precision and rounds once where the reference sums expanded f32
    // terms (transform.rs:944-2030): the two agree within 5e-7 of the box's extent (tests/test_motion_bounds.py, 96 reference-derived cases), either way round.  Those
    // edges move OUT by RSPT_MOTION_PAD x the largest extent, so the result contains the reference's box; it differs from it by at most 1.5e-6 of that extent.
    float ext = 0.0f;
    for (int i = 0; i < 3; i++) ext = fmaxf(ext, r.hi[i] - r.lo[i]);
    const float pad = RSPT_MOTION_PAD * ext;
    for (int i = 0; i < 3; i++) {
        if (r.lo[i] < e.lo[i]) r.lo[i] = nextafterf(r.lo[i] - pad, -3.40282347e38f);
        if (r.hi[i] > e.hi[i]) r.hi[i] = nextafterf(r.hi[i] + pad, 3.40282347e38f);
    }
    return r;
}

}  // namespace motion
}  // namespace rspt
