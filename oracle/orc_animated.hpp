// TEST INFRASTRUCTURE — AnimatedTransform of the camera (src/core/transform.rs:894-2124, src/core/quaternion.rs), restated.
// Only what a camera ray needs: new() up to the rotation test (the DerivativeTerm coefficients c1..c5, :944-2030, feed motion_bounds
// of moving primitives, which this path does not have), decompose, interpolate, transform_ray.  Matrices are row-major Float[16].
#pragma once
#include "orc_math.hpp"

namespace orc {

struct M44 { Float m[4][4]; };
static inline M44 m44_identity() { M44 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = i == j ? 1.0f : 0.0f; return r; }
static inline M44 m44_from(const Float* a) { M44 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = a[4 * i + j]; return r; }
static inline M44 m44_transpose(const M44& a) { M44 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i]; return r; } // transform.rs:118-127
static inline M44 m44_mul(const M44& a, const M44& b) { // mtx_mul, transform.rs:238-249
    M44 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
static inline M44 m44_inverse(const M44& m) { // Matrix4x4::inverse, transform.rs:128-200: Gauss-Jordan with full pivoting
    int indxc[4] = {0, 0, 0, 0}, indxr[4] = {0, 0, 0, 0}, ipiv[4] = {0, 0, 0, 0};
    M44 minv = m;
    for (int i = 0; i < 4; i++) {
        int irow = 0, icol = 0;
        Float big = 0.0f;
        for (int j = 0; j < 4; j++) {
            if (ipiv[j] != 1) {
                for (int k = 0; k < 4; k++) {
                    if (ipiv[k] == 0) {
                        Float a = std::fabs(minv.m[j][k]);
                        if (a >= big) { big = a; irow = j; icol = k; }
                    }
                }
            }
        }
        ipiv[icol] += 1;
        if (irow != icol) for (int k = 0; k < 4; k++) { Float s = minv.m[irow][k]; minv.m[irow][k] = minv.m[icol][k]; minv.m[icol][k] = s; }
        indxr[i] = irow; indxc[i] = icol;
        Float pivinv = 1.0f / minv.m[icol][icol];
        minv.m[icol][icol] = 1.0f;
        for (int j = 0; j < 4; j++) minv.m[icol][j] *= pivinv;
        for (int j = 0; j < 4; j++) {
            if (j != icol) {
                Float save = minv.m[j][icol];
                minv.m[j][icol] = 0.0f;
                for (int k = 0; k < 4; k++) minv.m[j][k] -= minv.m[icol][k] * save;
            }
        }
    }
    for (int i = 0; i < 4; i++) {
        int j = 3 - i;
        if (indxr[j] != indxc[j]) for (int k = 0; k < 4; k++) { Float s = minv.m[k][indxr[j]]; minv.m[k][indxr[j]] = minv.m[k][indxc[j]]; minv.m[k][indxc[j]] = s; }
    }
    return minv;
}

struct Quat { V3 v; Float w; };
static inline Float quat_dot(const Quat& a, const Quat& b) { return dot(a.v, b.v) + a.w * b.w; } // quaternion.rs:181-183
static inline Quat quat_scale(const Quat& q, Float s) { return Quat{q.v * s, q.w * s}; }
static inline Quat quat_add(const Quat& a, const Quat& b) { return Quat{a.v + b.v, a.w + b.w}; }
static inline Quat quat_sub(const Quat& a, const Quat& b) { return Quat{a.v - b.v, a.w - b.w}; }
static inline Quat quat_normalize(const Quat& q) { Float n = std::sqrt(quat_dot(q, q)); return Quat{q.v / n, q.w / n}; } // :186-188 (Vector3f / Float multiplies by 1 / n, w is divided)
static inline Quat quat_from_matrix(const M44& m) { // Quaternion::new(Transform), quaternion.rs:34-79
    Float trace = m.m[0][0] + m.m[1][1] + m.m[2][2];
    if (trace > 0.0f) {
        Float s = std::sqrt(trace + 1.0f);
        Float w = s / 2.0f;
        s = 0.5f / s;
        return Quat{V3{(m.m[2][1] - m.m[1][2]) * s, (m.m[0][2] - m.m[2][0]) * s, (m.m[1][0] - m.m[0][1]) * s}, w};
    }
    const int nxt[3] = {1, 2, 0};
    Float q[3] = {0.0f, 0.0f, 0.0f};
    int i = m.m[1][1] > m.m[0][0] ? 1 : 0;
    if (m.m[2][2] > m.m[i][i]) i = 2;
    int j = nxt[i], k = nxt[j];
    Float s = std::sqrt((m.m[i][i] - (m.m[j][j] + m.m[k][k])) + 1.0f);
    q[i] = s * 0.5f;
    if (s != 0.0f) s = 0.5f / s;
    Float w = (m.m[k][j] - m.m[j][k]) * s;
    q[j] = (m.m[j][i] + m.m[i][j]) * s;
    q[k] = (m.m[k][i] + m.m[i][k]) * s;
    return Quat{V3{q[0], q[1], q[2]}, w};
}
static inline M44 quat_to_matrix(const Quat& q) { // Quaternion::to_transform().m, quaternion.rs:80-109 (the transpose: "we are left-handed")
    Float xx = q.v.x * q.v.x, yy = q.v.y * q.v.y, zz = q.v.z * q.v.z;
    Float xy = q.v.x * q.v.y, xz = q.v.x * q.v.z, yz = q.v.y * q.v.z;
    Float wx = q.v.x * q.w, wy = q.v.y * q.w, wz = q.v.z * q.w;
    M44 m = m44_identity();
    m.m[0][0] = 1.0f - 2.0f * (yy + zz); m.m[0][1] = 2.0f * (xy + wz); m.m[0][2] = 2.0f * (xz - wy);
    m.m[1][0] = 2.0f * (xy - wz); m.m[1][1] = 1.0f - 2.0f * (xx + zz); m.m[1][2] = 2.0f * (yz + wx);
    m.m[2][0] = 2.0f * (xz + wy); m.m[2][1] = 2.0f * (yz - wx); m.m[2][2] = 1.0f - 2.0f * (xx + yy);
    return m44_transpose(m);
}
static inline Quat quat_slerp(Float t, const Quat& q1, const Quat& q2) { // quaternion.rs:168-180
    Float cos_theta = quat_dot(q1, q2);
    if (cos_theta > 0.9995f) return quat_normalize(quat_add(quat_scale(q1, 1.0f - t), quat_scale(q2, t)));
    Float theta = std::acos(clamp_t(cos_theta, -1.0f, 1.0f));
    Float thetap = theta * t;
    Quat qperp = quat_normalize(quat_sub(q2, quat_scale(q1, cos_theta)));
    return quat_add(quat_scale(q1, std::cos(thetap)), quat_scale(qperp, std::sin(thetap)));
}

struct AnimatedTransform {
    M44 start, end;
    Float start_time = 0.0f, end_time = 1.0f;
    bool actually_animated = false;
    V3 t[2];
    Quat r[2];
    M44 s[2];
    bool has_rotation = false;

    static void decompose(const M44& m, V3* t, Quat* rquat, M44* s) { // transform.rs:2032-2080
        *t = V3{m.m[0][3], m.m[1][3], m.m[2][3]};
        M44 matrix = m;
        for (int i = 0; i < 3; i++) { matrix.m[i][3] = 0.0f; matrix.m[3][i] = 0.0f; }
        matrix.m[3][3] = 1.0f;
        Float norm;
        int count = 0;
        M44 r = matrix;
        do { // polar decomposition: r <- (r + (r^T)^-1) / 2
            M44 rnext = m44_identity();
            M44 rit = m44_inverse(m44_transpose(r));
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) rnext.m[i][j] = 0.5f * (r.m[i][j] + rit.m[i][j]);
            norm = 0.0f;
            for (int i = 0; i < 3; i++) {
                Float n = std::fabs(r.m[i][0] - rnext.m[i][0]) + std::fabs(r.m[i][1] - rnext.m[i][1]) + std::fabs(r.m[i][2] - rnext.m[i][2]);
                norm = std::fmax(norm, n); // f32::max
            }
            r = rnext;
            count++;
        } while (!(count >= 100 || norm <= 0.0001f));
        *rquat = quat_from_matrix(r);
        *s = m44_mul(m44_inverse(r), m);
    }
    AnimatedTransform() : start(m44_identity()), end(m44_identity()) { t[0] = t[1] = V3{0, 0, 0}; r[0] = r[1] = Quat{V3{0, 0, 0}, 1.0f}; s[0] = s[1] = m44_identity(); }
    AnimatedTransform(const Float* start_m, Float t0, const Float* end_m, Float t1) { // AnimatedTransform::new, transform.rs:912-943
        start = m44_from(start_m); end = m44_from(end_m);
        start_time = t0; end_time = t1;
        actually_animated = false; // *start_transform != *end_transform (Matrix4x4::eq, transform.rs:203-214: element by element)
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) if (start.m[i][j] != end.m[i][j]) actually_animated = true;
        decompose(start, &t[0], &r[0], &s[0]);
        decompose(end, &t[1], &r[1], &s[1]);
        if (quat_dot(r[0], r[1]) < 0.0f) r[1] = Quat{-r[1].v, -r[1].w}; // the shorter arc
        has_rotation = quat_dot(r[0], r[1]) < 0.9995f;
    }
    M44 interpolate(Float time) const { // transform.rs:2081-2113 (the matrix m of the result)
        if (!actually_animated || time <= start_time) return start;
        if (time >= end_time) return end;
        Float dt = (time - start_time) / (end_time - start_time);
        V3 trans = t[0] * (1.0f - dt) + t[1] * dt;
        Quat rotate = quat_slerp(dt, r[0], r[1]);
        M44 scale = m44_identity();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) scale.m[i][j] = lerp(dt, s[0].m[i][j], s[1].m[i][j]);
        M44 tr = m44_identity();
        tr.m[0][3] = trans.x; tr.m[1][3] = trans.y; tr.m[2][3] = trans.z;
        return m44_mul(m44_mul(tr, quat_to_matrix(rotate)), scale); // Transform::translate(&trans) * rotate.to_transform() * Transform { m: scale, .. }
    }
    // the whole Transform interpolate leaves (:2106-2112): m as above, m_inv as the reverse product of the factors' inverses
    // (Transform * Transform = { mtx_mul(a.m, b.m), mtx_mul(b.m_inv, a.m_inv) }, transform.rs:869-877): translate's is translate(-t)
    // (:316-327), the quaternion's the transpose of its m (quaternion.rs:102-106), the scale's Matrix4x4::inverse(&scale) (:2111).
    // Outside the interval the key Transforms themselves come back, with the inverses they were built with (start_inv / end_inv).
    void interpolate_full(Float time, const M44& start_inv, const M44& end_inv, M44* m, M44* m_inv) const {
        if (!actually_animated || time <= start_time) { *m = start; *m_inv = start_inv; return; }
        if (time >= end_time) { *m = end; *m_inv = end_inv; return; }
        Float dt = (time - start_time) / (end_time - start_time);
        V3 trans = t[0] * (1.0f - dt) + t[1] * dt;
        Quat rotate = quat_slerp(dt, r[0], r[1]);
        M44 scale = m44_identity();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) scale.m[i][j] = lerp(dt, s[0].m[i][j], s[1].m[i][j]);
        M44 tr = m44_identity(), tr_inv = m44_identity();
        tr.m[0][3] = trans.x; tr.m[1][3] = trans.y; tr.m[2][3] = trans.z;
        tr_inv.m[0][3] = -trans.x; tr_inv.m[1][3] = -trans.y; tr_inv.m[2][3] = -trans.z;
        const M44 rot = quat_to_matrix(rotate), rot_inv = m44_transpose(rot);
        *m = m44_mul(m44_mul(tr, rot), scale);
        *m_inv = m44_mul(m44_inverse(scale), m44_mul(rot_inv, tr_inv));
    }
    Ray transform_ray(const Ray& r) const { // :2114-2124
        M44 m = interpolate(r.time);
        return orc::transform_ray(&m.m[0][0], r);
    }
};

} // namespace orc
