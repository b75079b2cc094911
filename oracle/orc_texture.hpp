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

// ---- Texture::evaluate: src/textures/{constant,imagemap,scale}.rs, mappings src/core/texture.rs:91-121,222-257 ----
static inline Spec tex_eval(const Scene& sc, uint32_t ti, const Interaction& si, int depth = 0) {
    const rspt_texture& tx = sc.d.textures[ti];
    switch (tx.kind) {
    case RSPT_TEX_CONSTANT: return S3(tx.value);
    case RSPT_TEX_SCALE:
        if (depth > 4) return Spec();
        return tex_eval(sc, tx.tex1, si, depth + 1) * tex_eval(sc, tx.tex2, si, depth + 1); // scale.rs:24-27
    case RSPT_TEX_IMAGE: {
        P2 st, dstdx, dstdy;
        if (tx.mapping == RSPT_MAP_PLANAR) {
            V3 vs{tx.map[0], tx.map[1], tx.map[2]}, vt{tx.map[3], tx.map[4], tx.map[5]};
            dstdx = P2{dot(si.dpdx, vs), dot(si.dpdx, vt)};
            dstdy = P2{dot(si.dpdy, vs), dot(si.dpdy, vt)};
            st = P2{tx.map[6] + dot(si.p, vs), tx.map[7] + dot(si.p, vt)};
        } else {
            dstdx = P2{si.dudx * tx.map[0], si.dvdx * tx.map[1]};
            dstdy = P2{si.dudy * tx.map[0], si.dvdy * tx.map[1]};
            st = P2{si.uv.x * tx.map[0] + tx.map[2], si.uv.y * tx.map[1] + tx.map[3]};
        }
        return img_lookup(sc.d.images[tx.image], tx, st, dstdx, dstdy);
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

} // namespace orc
