// TEST INFRASTRUCTURE — CPU oracle (see orc_math.hpp header).  Textures: MipMap lookups,
// 2-D mappings, screen-space differentials, bump mapping.
#pragma once
#include "orc_scene.hpp"

namespace orc {

static inline Spec S3(const float* p) { return Spec(p[0], p[1], p[2]); }

// ---- SurfaceInteraction::compute_differentials: src/core/interaction.rs:388-479 ----
static inline bool solve_linear_system_2x2(const Float a[2][2], const Float b[2], Float* x0, Float* x1) { // transform.rs:219-235
    Float det = a[0][0] * a[1][1] - a[0][1] * a[1][0];
    if (std::fabs(det) < 1e-10f) return false;
    *x0 = (a[1][1] * b[0] - a[0][1] * b[1]) / det;
    *x1 = (a[0][0] * b[1] - a[1][0] * b[0]) / det;
    if (std::isnan(*x0) || std::isnan(*x1)) return false;
    return true;
}
static inline Float vcomp(V3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }
static inline void compute_differentials(Interaction* si, const Ray& ray) {
    si->dudx = si->dvdx = si->dudy = si->dvdy = 0.0f;
    si->dpdx = si->dpdy = V3{0, 0, 0};
    if (!ray.has_diff) return;
    Float d = dot(si->n, si->p);
    Float tx = -(dot(si->n, ray.rx_o) - d) / dot(si->n, ray.rx_d);
    if (std::isinf(tx) || std::isnan(tx)) return;
    V3 px = ray.rx_o + ray.rx_d * tx;
    Float ty = -(dot(si->n, ray.ry_o) - d) / dot(si->n, ray.ry_d);
    if (std::isinf(ty) || std::isnan(ty)) return;
    V3 py = ray.ry_o + ray.ry_d * ty;
    si->dpdx = px - si->p;
    si->dpdy = py - si->p;
    int dim[2];
    if (std::fabs(si->n.x) > std::fabs(si->n.y) && std::fabs(si->n.x) > std::fabs(si->n.z)) { dim[0] = 1; dim[1] = 2; }
    else if (std::fabs(si->n.y) > std::fabs(si->n.z)) { dim[0] = 0; dim[1] = 2; }
    else { dim[0] = 0; dim[1] = 1; }
    const Float a[2][2] = {{vcomp(si->dpdu, dim[0]), vcomp(si->dpdv, dim[0])}, {vcomp(si->dpdu, dim[1]), vcomp(si->dpdv, dim[1])}};
    const Float bx[2] = {vcomp(px, dim[0]) - vcomp(si->p, dim[0]), vcomp(px, dim[1]) - vcomp(si->p, dim[1])};
    const Float by[2] = {vcomp(py, dim[0]) - vcomp(si->p, dim[0]), vcomp(py, dim[1]) - vcomp(si->p, dim[1])};
    if (!solve_linear_system_2x2(a, bx, &si->dudx, &si->dvdx)) { si->dudx = 0.0f; si->dvdx = 0.0f; }
    if (!solve_linear_system_2x2(a, by, &si->dudy, &si->dvdy)) { si->dudy = 0.0f; si->dvdy = 0.0f; }
}

// ---- MipMap<T>: src/core/mipmap.rs:197-400 ----
static const int WEIGHT_LUT_SIZE = 128; // mipmap.rs:21
static inline const Float* ewa_weight_lut() { // :186-192
    static Float lut[WEIGHT_LUT_SIZE];
    static bool init = false;
    if (!init) {
        for (int i = 0; i < WEIGHT_LUT_SIZE; i++) {
            Float alpha = 2.0f;
            Float r2 = (Float)i / (Float)(WEIGHT_LUT_SIZE - 1);
            lut[i] = std::exp(-alpha * r2) - std::exp(-alpha);
        }
        init = true;
    }
    return lut;
}
static inline const float* img_level(const rspt_image& m, uint32_t level, uint32_t* w, uint32_t* h) {
    const float* p = m.texels;
    uint32_t lw = m.width, lh = m.height;
    for (uint32_t i = 0; i < level; i++) { p += (size_t)m.channels * lw * lh; lw = std::max(1u, lw / 2); lh = std::max(1u, lh / 2); }
    *w = lw; *h = lh;
    return p;
}
static inline Spec img_texel(const rspt_image& m, uint32_t wrap, uint32_t level, int64_t s, int64_t t) { // :206-232
    uint32_t w, h;
    const float* p = img_level(m, level, &w, &h);
    uint64_t ss, tt;
    if (wrap == RSPT_WRAP_REPEAT) { ss = (uint64_t)s % (uint64_t)w; tt = (uint64_t)t % (uint64_t)h; } // mod_t(s as usize, u_size)
    else { ss = (uint64_t)clamp_t<int64_t>(s, 0, (int64_t)w - 1); tt = (uint64_t)clamp_t<int64_t>(t, 0, (int64_t)h - 1); } // Clamp, and Black's "TMP" branch
    const float* q = p + (size_t)m.channels * (tt * w + ss);
    return m.channels == 1 ? Spec(q[0]) : S3(q);
}
static inline Spec img_triangle(const rspt_image& m, uint32_t wrap, uint32_t level, P2 st) { // :323-336
    if (level > m.n_levels - 1) level = m.n_levels - 1;
    uint32_t w, h;
    img_level(m, level, &w, &h);
    Float s = st.x * (Float)w - 0.5f, t = st.y * (Float)h - 0.5f;
    int64_t s0 = f2i64(std::floor(s)), t0 = f2i64(std::floor(t));
    Float ds = s - (Float)s0, dt = t - (Float)t0;
    Spec tmp1 = img_texel(m, wrap, level, s0 + 1, t0 + 1) * (ds * dt);
    Spec tmp2 = img_texel(m, wrap, level, s0 + 1, t0) * (ds * (1.0f - dt));
    Spec tmp3 = img_texel(m, wrap, level, s0, t0 + 1) * ((1.0f - ds) * dt);
    Spec tmp4 = img_texel(m, wrap, level, s0, t0) * ((1.0f - ds) * (1.0f - dt));
    return tmp4 + tmp3 + tmp2 + tmp1;
}
static inline Spec img_lookup_width(const rspt_image& m, uint32_t wrap, P2 st, Float width) { // lookup_pnt_flt :233-252
    Float level = (Float)m.n_levels - 1.0f + std::log2(std::fmax(width, 1e-8f));
    if (level < 0.0f) return img_triangle(m, wrap, 0, st);
    if (level >= (Float)m.n_levels - 1.0f) return img_texel(m, wrap, m.n_levels - 1, 0, 0);
    uint32_t il = (uint32_t)f2usize(std::floor(level));
    Float delta = level - (Float)il;
    Spec a = img_triangle(m, wrap, il, st), b = img_triangle(m, wrap, il + 1, st);
    return a * (1.0f - delta) + b * delta; // lerp(delta, a, b)
}
static inline Spec img_ewa(const rspt_image& m, uint32_t wrap, uint32_t level, P2 st, P2 dst0, P2 dst1) { // :337-400
    if (level >= m.n_levels) return img_texel(m, wrap, m.n_levels - 1, 0, 0);
    uint32_t w, h;
    img_level(m, level, &w, &h);
    Float sx = st.x * (Float)w - 0.5f, sy = st.y * (Float)h - 0.5f;
    Float d0x = dst0.x * (Float)w, d0y = dst0.y * (Float)h, d1x = dst1.x * (Float)w, d1y = dst1.y * (Float)h;
    Float a = d0y * d0y + d1y * d1y + 1.0f;
    Float b = -2.0f * (d0x * d0y + d1x * d1y);
    Float c = d0x * d0x + d1x * d1x + 1.0f;
    Float inv_f = 1.0f / (a * c - b * b * 0.25f);
    a *= inv_f; b *= inv_f; c *= inv_f;
    Float det = -b * b + 4.0f * a * c;
    Float inv_det = 1.0f / det;
    Float u_sqrt = std::sqrt(det * c), v_sqrt = std::sqrt(a * det);
    int64_t s0 = f2i64(std::ceil(sx - 2.0f * inv_det * u_sqrt)), s1 = f2i64(std::floor(sx + 2.0f * inv_det * u_sqrt));
    int64_t t0 = f2i64(std::ceil(sy - 2.0f * inv_det * v_sqrt)), t1 = f2i64(std::floor(sy + 2.0f * inv_det * v_sqrt));
    // guard shared with the device code (the reference scans the ellipse bound unconditionally; a normal
    // footprint is a few texels wide after the eccentricity clamp)
    if (s1 - s0 > 256) s1 = s0 + 256;
    if (t1 - t0 > 256) t1 = t0 + 256;
    const Float* lut = ewa_weight_lut();
    Spec sum;
    Float sum_wts = 0.0f;
    for (int64_t it = t0; it <= t1; it++) {
        Float tt = (Float)it - sy;
        for (int64_t is = s0; is <= s1; is++) {
            Float ss = (Float)is - sx;
            Float r2 = a * ss * ss + b * ss * tt + c * tt * tt;
            if (r2 < 1.0f) {
                size_t index = std::min<size_t>((size_t)f2usize(r2 * (Float)WEIGHT_LUT_SIZE), WEIGHT_LUT_SIZE - 1);
                Float weight = lut[index];
                sum = sum + img_texel(m, wrap, level, is, it) * weight;
                sum_wts += weight;
            }
        }
    }
    return sum / sum_wts;
}
static inline Spec img_lookup(const rspt_image& m, const rspt_texture& tx, P2 st, P2 dst0, P2 dst1) { // lookup_pnt_vec_vec :253-297
    if (tx.trilinear) {
        Float width = std::fmax(std::fmax(std::fabs(dst0.x), std::fabs(dst0.y)), std::fmax(std::fabs(dst1.x), std::fabs(dst1.y)));
        return img_lookup_width(m, tx.wrap, st, width);
    }
    if (dst0.x * dst0.x + dst0.y * dst0.y < dst1.x * dst1.x + dst1.y * dst1.y) std::swap(dst0, dst1);
    Float major_length = std::sqrt(dst0.x * dst0.x + dst0.y * dst0.y);
    Float minor_length = std::sqrt(dst1.x * dst1.x + dst1.y * dst1.y);
    if (minor_length * tx.max_aniso < major_length && minor_length > 0.0f) {
        Float scale = major_length / (minor_length * tx.max_aniso);
        dst1.x *= scale; dst1.y *= scale;
        minor_length *= scale;
    }
    if (minor_length == 0.0f) return img_triangle(m, tx.wrap, 0, st);
    Float lod = std::fmax(0.0f, (Float)m.n_levels - 1.0f + std::log2(minor_length));
    uint32_t ilod = (uint32_t)f2usize(std::floor(lod));
    Spec col2 = img_ewa(m, tx.wrap, ilod + 1, st, dst0, dst1);
    Spec col1 = img_ewa(m, tx.wrap, ilod, st, dst0, dst1);
    Float t = lod - (Float)ilod;
    return col1 * (1.0f - t) + col2 * t;
}

// ---- geometry.rs:1584-1596 ----
static inline Float spherical_theta(V3 v) { return std::acos(clamp_t(v.z, -1.0f, 1.0f)); }
static inline Float spherical_phi(V3 v) { Float p = std::atan2(v.y, v.x); return p < 0.0f ? p + 2.0f * PI : p; }

// ---- Perlin noise: src/core/texture.rs:21-48 (Ken Perlin's reference permutation, twice), 289-425 ----
static const uint8_t NOISE_PERM[512] = {
#define RSPT_PERLIN_PERM \
    151, 160, 137, 91, 90, 15, 131, 13, 201, 95, 96, 53, 194, 233, 7, 225, 140, 36, 103, 30, 69, 142, 8, 99, 37, 240, 21, 10, 23, 190, 6, 148, 247, 120, \
    234, 75, 0, 26, 197, 62, 94, 252, 219, 203, 117, 35, 11, 32, 57, 177, 33, 88, 237, 149, 56, 87, 174, 20, 125, 136, 171, 168, 68, 175, 74, 165, 71, \
    134, 139, 48, 27, 166, 77, 146, 158, 231, 83, 111, 229, 122, 60, 211, 133, 230, 220, 105, 92, 41, 55, 46, 245, 40, 244, 102, 143, 54, 65, 25, 63, \
    161, 1, 216, 80, 73, 209, 76, 132, 187, 208, 89, 18, 169, 200, 196, 135, 130, 116, 188, 159, 86, 164, 100, 109, 198, 173, 186, 3, 64, 52, 217, 226, \
    250, 124, 123, 5, 202, 38, 147, 118, 126, 255, 82, 85, 212, 207, 206, 59, 227, 47, 16, 58, 17, 182, 189, 28, 42, 223, 183, 170, 213, 119, 248, 152, \
    2, 44, 154, 163, 70, 221, 153, 101, 155, 167, 43, 172, 9, 129, 22, 39, 253, 19, 98, 108, 110, 79, 113, 224, 232, 178, 185, 112, 104, 218, 246, 97, \
    228, 251, 34, 242, 193, 238, 210, 144, 12, 191, 179, 162, 241, 81, 51, 145, 235, 249, 14, 239, 107, 49, 192, 214, 31, 181, 199, 106, 157, 184, 84, \
    204, 176, 115, 121, 50, 45, 127, 4, 150, 254, 138, 236, 205, 93, 222, 114, 67, 29, 24, 72, 243, 141, 128, 195, 78, 66, 215, 61, 156, 180
    RSPT_PERLIN_PERM, RSPT_PERLIN_PERM
#undef RSPT_PERLIN_PERM
};
static inline Float smooth_step(Float mn, Float mx, Float value) { // texture.rs:289-292
    Float v = clamp_t((value - mn) / (mx - mn), 0.0f, 1.0f);
    return v * v * (-2.0f * v + 3.0f);
}
static inline Float noise_grad(int32_t x, int32_t y, int32_t z, Float dx, Float dy, Float dz) { // :342-358
    uint8_t h = NOISE_PERM[NOISE_PERM[NOISE_PERM[x] + y] + z];
    h &= 15;
    Float u = (h < 8 || h == 12 || h == 13) ? dx : dy;
    Float v = (h < 4 || h == 12 || h == 13) ? dy : dz;
    return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
static inline Float noise_weight(Float t) { Float t3 = t * t * t, t4 = t3 * t; return 6.0f * t4 * t - 15.0f * t4 + 10.0f * t3; } // :360-364
static inline Float noise_flt(Float x, Float y, Float z) { // :294-336
    int32_t ix = f2i(std::floor(x)), iy = f2i(std::floor(y)), iz = f2i(std::floor(z));
    Float dx = x - (Float)ix, dy = y - (Float)iy, dz = z - (Float)iz;
    ix &= 255; iy &= 255; iz &= 255;
    Float w000 = noise_grad(ix, iy, iz, dx, dy, dz), w100 = noise_grad(ix + 1, iy, iz, dx - 1.0f, dy, dz);
    Float w010 = noise_grad(ix, iy + 1, iz, dx, dy - 1.0f, dz), w110 = noise_grad(ix + 1, iy + 1, iz, dx - 1.0f, dy - 1.0f, dz);
    Float w001 = noise_grad(ix, iy, iz + 1, dx, dy, dz - 1.0f), w101 = noise_grad(ix + 1, iy, iz + 1, dx - 1.0f, dy, dz - 1.0f);
    Float w011 = noise_grad(ix, iy + 1, iz + 1, dx, dy - 1.0f, dz - 1.0f), w111 = noise_grad(ix + 1, iy + 1, iz + 1, dx - 1.0f, dy - 1.0f, dz - 1.0f);
    Float wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
    Float x00 = lerp(wx, w000, w100), x10 = lerp(wx, w010, w110), x01 = lerp(wx, w001, w101), x11 = lerp(wx, w011, w111);
    Float y0 = lerp(wy, x00, x10), y1 = lerp(wy, x01, x11);
    return lerp(wz, y0, y1);
}
static inline Float log_2(Float x) { return std::log(x) * 1.44269504088896340736f; } // pbrt.rs:153-156 (ln * LOG2_E)
static inline Float fbm(V3 p, V3 dpdx, V3 dpdy, Float omega, int32_t max_octaves) { // texture.rs:366-386
    Float len2 = std::fmax(length_squared(dpdx), length_squared(dpdy));
    Float n = clamp_t(-1.0f - 0.5f * log_2(len2), 0.0f, (Float)max_octaves);
    int32_t n_int = f2i(std::floor(n));
    Float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (int32_t i = 0; i < n_int; i++) {
        sum += o * noise_flt(p.x * lambda, p.y * lambda, p.z * lambda);
        lambda *= 1.99f;
        o *= omega;
    }
    Float n_partial = n - (Float)n_int;
    sum += o * smooth_step(0.3f, 0.7f, n_partial) * noise_flt(p.x * lambda, p.y * lambda, p.z * lambda);
    return sum;
}
static inline Float turbulence(V3 p, V3 dpdx, V3 dpdy, Float omega, int32_t max_octaves) { // texture.rs:388-424
    Float len2 = std::fmax(length_squared(dpdx), length_squared(dpdy));
    Float n = clamp_t(-1.0f - 0.5f * log_2(len2), 0.0f, (Float)max_octaves);
    uint64_t n_int = f2usize(std::floor(n));
    Float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (uint64_t i = 0; i < n_int; i++) {
        sum += o * std::fabs(noise_flt(p.x * lambda, p.y * lambda, p.z * lambda));
        lambda *= 1.99f;
        o *= omega;
    }
    Float n_partial = n - (Float)n_int;
    sum += o * lerp(smooth_step(0.3f, 0.7f, n_partial), 0.2f, std::fabs(noise_flt(p.x * lambda, p.y * lambda, p.z * lambda)));
    for (uint64_t i = n_int; i < (uint64_t)std::max(max_octaves, 0); i++) {
        sum += o * 0.2f;
        o *= omega;
    }
    return sum;
}

// ---- TextureMapping2D / 3D: src/core/texture.rs:51-283 ----
static inline P2 map_sphere(const rspt_texture& tx, V3 p) { // :135-144
    V3 v = normalize(transform_point(tx.world_to_texture, p) - V3{0, 0, 0});
    return P2{spherical_theta(v) * INV_PI, spherical_phi(v) * INV_2_PI};
}
static inline P2 map_cylinder(const rspt_texture& tx, V3 p) { // :184-191
    V3 v = normalize(transform_point(tx.world_to_texture, p) - V3{0, 0, 0});
    return P2{PI + std::atan2(v.y, v.x) * INV_2_PI, v.z};
}
static inline void wrap_dt(P2* d) { // the `if dstdx[1] > 0.5 ... else if < -0.5` fix-ups (:158-167, :203-217)
    if (d->y > 0.5f) d->y = 1.0f - d->y;
    else if (d->y < -0.5f) d->y = -(d->y + 1.0f);
}
static inline P2 tex_map2d(const rspt_texture& tx, const Interaction& si, P2* dstdx, P2* dstdy) {
    switch (tx.mapping) {
    case RSPT_MAP_PLANAR: {
        V3 vs{tx.map[0], tx.map[1], tx.map[2]}, vt{tx.map[3], tx.map[4], tx.map[5]};
        *dstdx = P2{dot(si.dpdx, vs), dot(si.dpdx, vt)};
        *dstdy = P2{dot(si.dpdy, vs), dot(si.dpdy, vt)};
        return P2{tx.map[6] + dot(si.p, vs), tx.map[7] + dot(si.p, vt)};
    }
    case RSPT_MAP_SPHERICAL: {
        P2 st = map_sphere(tx, si.p);
        const Float delta = 0.1f;
        const Float inv = 1.0f / delta; // `Vector2f / Float` multiplies by the reciprocal (geometry.rs:1281-1288): not the same bits as a division (found by the text pin, round 6)
        P2 sx = map_sphere(tx, si.p + si.dpdx * delta);
        *dstdx = P2{(sx.x - st.x) * inv, (sx.y - st.y) * inv};
        P2 sy = map_sphere(tx, si.p + si.dpdy * delta);
        *dstdy = P2{(sy.x - st.x) * inv, (sy.y - st.y) * inv};
        wrap_dt(dstdx); wrap_dt(dstdy);
        return st;
    }
    case RSPT_MAP_CYLINDRICAL: {
        P2 st = map_cylinder(tx, si.p);
        const Float delta = 0.01f;
        const Float inv = 1.0f / delta; // (as above)
        P2 sx = map_cylinder(tx, si.p + si.dpdx * delta);
        *dstdx = P2{(sx.x - st.x) * inv, (sx.y - st.y) * inv};
        wrap_dt(dstdx);
        P2 sy = map_cylinder(tx, si.p + si.dpdy * delta);
        *dstdy = P2{(sy.x - st.x) * inv, (sy.y - st.y) * inv};
        wrap_dt(dstdy);
        return st;
    }
    default:
        *dstdx = P2{si.dudx * tx.map[0], si.dvdx * tx.map[1]};
        *dstdy = P2{si.dudy * tx.map[0], si.dvdy * tx.map[1]};
        return P2{si.uv.x * tx.map[0] + tx.map[2], si.uv.y * tx.map[1] + tx.map[3]};
    }
}
static inline V3 tex_map3d(const rspt_texture& tx, const Interaction& si, V3* dpdx, V3* dpdy) { // IdentityMapping3D :270-282
    *dpdx = transform_vector(tx.world_to_texture, si.dpdx);
    *dpdy = transform_vector(tx.world_to_texture, si.dpdy);
    return transform_point(tx.world_to_texture, si.p);
}

// ---- Texture::evaluate: src/textures/*.rs ----
static inline Spec tex_eval(const Scene& sc, uint32_t ti, const Interaction& si, int depth = 0) {
    const rspt_texture& tx = sc.d.textures[ti];
    if (depth > 4) return Spec();
    switch (tx.kind) {
    case RSPT_TEX_CONSTANT: return S3(tx.value);
    case RSPT_TEX_SCALE: return tex_eval(sc, tx.tex1, si, depth + 1) * tex_eval(sc, tx.tex2, si, depth + 1); // scale.rs:24-27
    case RSPT_TEX_MIX: { // mix.rs:30-35
        Spec t1 = tex_eval(sc, tx.tex1, si, depth + 1), t2 = tex_eval(sc, tx.tex2, si, depth + 1);
        Float amt = tex_eval(sc, tx.tex3, si, depth + 1).c[0];
        return t1 * Spec(1.0f - amt) + t2 * Spec(amt);
    }
    case RSPT_TEX_IMAGE: {
        P2 dstdx, dstdy;
        P2 st = tex_map2d(tx, si, &dstdx, &dstdy);
        return img_lookup(sc.d.images[tx.image], tx, st, dstdx, dstdy);
    }
    case RSPT_TEX_CHECKERBOARD: { // checkerboard.rs:32-42: `(floor(s) as u32 + floor(t) as u32) % 2`
        P2 dstdx, dstdy;
        P2 st = tex_map2d(tx, si, &dstdx, &dstdy);
        uint32_t a = f2u32(std::floor(st.x)), b = f2u32(std::floor(st.y));
        return ((a + b) % 2u == 0u) ? tex_eval(sc, tx.tex1, si, depth + 1) : tex_eval(sc, tx.tex2, si, depth + 1);
    }
    case RSPT_TEX_DOTS: { // dots.rs:31-72
        P2 dstdx, dstdy;
        P2 st = tex_map2d(tx, si, &dstdx, &dstdy);
        int32_t s_cell = f2i(std::floor(st.x + 0.5f)), t_cell = f2i(std::floor(st.y + 0.5f));
        if (noise_flt((Float)s_cell + 0.5f, (Float)t_cell + 0.5f, 0.5f) > 0.0f) {
            const Float radius = 0.35f, max_shift = 0.5f - radius;
            Float s_center = (Float)s_cell + max_shift * noise_flt((Float)s_cell + 1.5f, (Float)t_cell + 2.8f, 0.5f);
            Float t_center = (Float)t_cell + max_shift * noise_flt((Float)s_cell + 4.5f, (Float)t_cell + 9.8f, 0.5f);
            Float dx = st.x - s_center, dy = st.y - t_center;
            if (dx * dx + dy * dy < radius * radius) return tex_eval(sc, tx.tex2, si, depth + 1); // inside_dot
        }
        return tex_eval(sc, tx.tex1, si, depth + 1); // outside_dot
    }
    case RSPT_TEX_FBM: { V3 dpdx, dpdy; V3 p = tex_map3d(tx, si, &dpdx, &dpdy); return Spec(fbm(p, dpdx, dpdy, tx.omega, tx.octaves)); }
    case RSPT_TEX_WRINKLED: { V3 dpdx, dpdy; V3 p = tex_map3d(tx, si, &dpdx, &dpdy); return Spec(turbulence(p, dpdx, dpdy, tx.omega, tx.octaves)); }
    case RSPT_TEX_WINDY: { // windy.rs:23-37
        V3 dpdx, dpdy; V3 p = tex_map3d(tx, si, &dpdx, &dpdy);
        Float wind_strength = fbm(p * 0.1f, dpdx * 0.1f, dpdy * 0.1f, 0.5f, 3);
        Float wave_height = fbm(p, dpdx, dpdy, 0.5f, 6);
        return Spec(std::fabs(wind_strength) * wave_height);
    }
    case RSPT_TEX_MARBLE: { // marble.rs:43-92
        V3 dpdx, dpdy; V3 p = tex_map3d(tx, si, &dpdx, &dpdy);
        p = p * tx.scale;
        Float marble = p.y + tx.variation * fbm(p, dpdx * tx.scale, dpdy * tx.scale, tx.omega, tx.octaves);
        Float t = 0.5f + 0.5f * std::sin(marble);
        static const Float c[9][3] = {{0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.5f, 0.5f, 0.5f}, {0.6f, 0.59f, 0.58f},
                                      {0.58f, 0.58f, 0.6f}, {0.58f, 0.58f, 0.6f}, {0.2f, 0.2f, 0.33f}, {0.58f, 0.58f, 0.6f}};
        uint64_t first = f2usize(std::floor(t * 6.0f));
        if (first > 5) first = 5;
        t = t * 6.0f - (Float)first;
        Spec c0 = S3(c[first]), c1 = S3(c[first + 1]), c2 = S3(c[first + 2]), c3 = S3(c[first + 3]);
        Spec s0 = c0 * (1.0f - t) + c1 * t, s1 = c1 * (1.0f - t) + c2 * t, s2 = c2 * (1.0f - t) + c3 * t;
        s0 = s0 * (1.0f - t) + s1 * t;
        s1 = s1 * (1.0f - t) + s2 * t;
        return (s0 * (1.0f - t) + s1 * t) * 1.5f;
    }
    }
    return Spec();
}
static inline Spec sclamp0(Spec s) { // Spectrum::clamp(0, inf) (spectrum.rs clamp_t per channel)
    return Spec(clamp_t(s.c[0], 0.0f, INF), clamp_t(s.c[1], 0.0f, INF), clamp_t(s.c[2], 0.0f, INF));
}

// ---- Material::bump: src/core/material.rs:116-219; set_shading_geometry interaction.rs:345-370 (si.shape is None for triangles) ----
static inline void bump(const Scene& sc, uint32_t ti, Interaction* si) {
    Interaction ev = *si;
    Float du = 0.5f * (std::fabs(si->dudx) + std::fabs(si->dudy));
    if (du == 0.0f) du = 0.0005f;
    ev.p = si->p + si->sh_dpdu * du;
    ev.uv = P2{si->uv.x + du, si->uv.y + 0.0f};
    ev.n = normalize(cross(si->sh_dpdu, si->sh_dpdv) + V3{0, 0, 0} * du); // si.dndu is the zero normal
    Float u_displace = tex_eval(sc, ti, ev).c[0];
    Float dv = 0.5f * (std::fabs(si->dvdx) + std::fabs(si->dvdy));
    if (dv == 0.0f) dv = 0.0005f;
    ev.p = si->p + si->sh_dpdv * dv;
    ev.uv = P2{si->uv.x + 0.0f, si->uv.y + dv};
    ev.n = normalize(cross(si->sh_dpdu, si->sh_dpdv) + V3{0, 0, 0} * dv);
    Float v_displace = tex_eval(sc, ti, ev).c[0];
    Float displace = tex_eval(sc, ti, *si).c[0];
    V3 dpdu = si->sh_dpdu + si->sh_n * ((u_displace - displace) / du) + si->sh_dndu * displace;
    V3 dpdv = si->sh_dpdv + si->sh_n * ((v_displace - displace) / dv) + si->sh_dndv * displace;
    // set_shading_geometry(dpdu, dpdv, dndu, dndv, false)
    si->sh_n = normalize(cross(dpdu, dpdv));
    si->sh_n = faceforward(si->sh_n, si->n);
    si->sh_dpdu = dpdu; si->sh_dpdv = dpdv;
}

static inline Float alpha_texture_value(const Scene& sc, uint32_t ti, const Interaction& si) { return tex_eval(sc, ti, si).c[0]; }

} // namespace orc
