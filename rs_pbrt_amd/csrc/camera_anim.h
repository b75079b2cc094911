// Host side of a moving camera: AnimatedTransform::new up to the rotation test (src/core/transform.rs:912-943) — the two key matrices
// decomposed into translation, rotation and scale (AnimatedTransform::decompose, :2032-2080) for the per-ray interpolation in
// dev_scene.h camera_to_world_at.  Plain f32, operation by operation as the reference writes it (this translation unit is compiled
// without FMA contraction), because the quaternions and scale matrices that come out are inputs of every camera ray.
// The derivative terms c1..c5 (:944-2030) belong to motion_bounds of moving primitives and are not needed for a camera.
#pragma once
#include <cmath>
#include <cstring>

#include "dev_scene.h"

namespace rspt {
namespace camanim {

struct M4 { float m[4][4]; };

inline M4 identity() { M4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = i == j ? 1.0f : 0.0f; return r; }
inline M4 transpose(const M4& a) { M4 r; for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) r.m[i][j] = a.m[j][i]; return r; }
inline M4 mul(const M4& a, const M4& b) {  // mtx_mul (:238-249)
    M4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j] + a.m[i][3] * b.m[3][j];
    return r;
}
// Matrix4x4::inverse (:128-200): Gauss-Jordan elimination, the pivot is the largest remaining element (later candidates win ties)
inline M4 inverse(const M4& src) {
    int col_of[4] = {0, 0, 0, 0}, row_of[4] = {0, 0, 0, 0}, used[4] = {0, 0, 0, 0};
    M4 a = src;
    for (int step = 0; step < 4; step++) {
        int pr = 0, pc = 0;
        float best = 0.0f;
        for (int r = 0; r < 4; r++) {
            if (used[r] == 1) continue;
            for (int c = 0; c < 4; c++) {
                if (used[c] != 0) continue;
                const float v = fabsf(a.m[r][c]);
                if (v >= best) { best = v; pr = r; pc = c; }
            }
        }
        used[pc] += 1;
        if (pr != pc) for (int k = 0; k < 4; k++) { const float t = a.m[pr][k]; a.m[pr][k] = a.m[pc][k]; a.m[pc][k] = t; }
        row_of[step] = pr; col_of[step] = pc;
        const float pivinv = 1.0f / a.m[pc][pc];
        a.m[pc][pc] = 1.0f;
        for (int k = 0; k < 4; k++) a.m[pc][k] *= pivinv;
        for (int r = 0; r < 4; r++) {
            if (r == pc) continue;
            const float save = a.m[r][pc];
            a.m[r][pc] = 0.0f;
            for (int k = 0; k < 4; k++) a.m[r][k] -= a.m[pc][k] * save;
        }
    }
    for (int step = 3; step >= 0; step--)
        if (row_of[step] != col_of[step])
            for (int k = 0; k < 4; k++) { const float t = a.m[k][row_of[step]]; a.m[k][row_of[step]] = a.m[k][col_of[step]]; a.m[k][col_of[step]] = t; }
    return a;
}
// Quaternion::new(Transform) (quaternion.rs:34-79) -> (x, y, z, w)
inline void quat_of(const M4& m, float q[4]) {
    const float trace = m.m[0][0] + m.m[1][1] + m.m[2][2];
    if (trace > 0.0f) {
        float s = sqrtf(trace + 1.0f);
        q[3] = s / 2.0f;
        s = 0.5f / s;
        q[0] = (m.m[2][1] - m.m[1][2]) * s; q[1] = (m.m[0][2] - m.m[2][0]) * s; q[2] = (m.m[1][0] - m.m[0][1]) * s;
        return;
    }
    const int next[3] = {1, 2, 0};
    int i = m.m[1][1] > m.m[0][0] ? 1 : 0;
    if (m.m[2][2] > m.m[i][i]) i = 2;
    const int j = next[i], k = next[j];
    float s = sqrtf((m.m[i][i] - (m.m[j][j] + m.m[k][k])) + 1.0f);
    float v[3] = {0.0f, 0.0f, 0.0f};
    v[i] = s * 0.5f;
    if (s != 0.0f) s = 0.5f / s;
    q[3] = (m.m[k][j] - m.m[j][k]) * s;
    v[j] = (m.m[j][i] + m.m[i][j]) * s;
    v[k] = (m.m[k][i] + m.m[i][k]) * s;
    q[0] = v[0]; q[1] = v[1]; q[2] = v[2];
}
// AnimatedTransform::decompose (:2032-2080): translation, then the rotation by polar decomposition (R <- (R + R^-T) / 2 until it moves
// by at most 1e-4 in the max row-sum norm, 100 rounds at most), then S = R^-1 M
inline void decompose(const M4& m, float t[3], float q[4], float s[16]) {
    t[0] = m.m[0][3]; t[1] = m.m[1][3]; t[2] = m.m[2][3];
    M4 r = m;
    for (int i = 0; i < 3; i++) { r.m[i][3] = 0.0f; r.m[3][i] = 0.0f; }
    r.m[3][3] = 1.0f;
    for (int count = 1;; count++) {
        const M4 rit = inverse(transpose(r));
        M4 rnext;
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) rnext.m[i][j] = 0.5f * (r.m[i][j] + rit.m[i][j]);
        float norm = 0.0f;
        for (int i = 0; i < 3; i++) {
            const float n = fabsf(r.m[i][0] - rnext.m[i][0]) + fabsf(r.m[i][1] - rnext.m[i][1]) + fabsf(r.m[i][2] - rnext.m[i][2]);
            norm = fmaxf(norm, n);
        }
        r = rnext;
        if (count >= 100 || norm <= 0.0001f) break;
    }
    quat_of(r, q);
    const M4 sm = mul(inverse(r), m);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) s[4 * i + j] = sm.m[i][j];
}
// false: the two matrices are equal (actually_animated = false, :923): the start matrix serves every ray
inline bool camera_keys(const float start[16], float t_start, const float end[16], float t_end, CamAnim* out) {
    bool differ = false;
    for (int i = 0; i < 16; i++) if (start[i] != end[i]) differ = true;
    if (!differ) return false;
    M4 a, b;
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { a.m[i][j] = start[4 * i + j]; b.m[i][j] = end[4 * i + j]; }
    memcpy(out->end, end, sizeof out->end);
    decompose(a, out->t[0], out->r[0], out->s[0]);
    decompose(b, out->t[1], out->r[1], out->s[1]);
    const float d = (out->r[0][0] * out->r[1][0] + out->r[0][1] * out->r[1][1] + out->r[0][2] * out->r[1][2]) + out->r[0][3] * out->r[1][3];
    if (d < 0.0f) for (int i = 0; i < 4; i++) out->r[1][i] = -out->r[1][i];  // the shorter arc (:934-936)
    out->time[0] = t_start; out->time[1] = t_end;
    return true;
}

}  // namespace camanim
}  // namespace rspt
