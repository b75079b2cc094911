// Textures on the device: MipMap lookups (trilinear + EWA), UV / planar mappings,
// screen-space differentials of the camera hit, bump mapping — evaluated by k_texture, a stage of
// its own in front of k_shade, so that scenes without textures pay nothing (k_shade only reads the
// per-path results when the hit material is flagged as textured).
#pragma once
#include "dev_scene.h"

namespace rspt {

struct ImageDev {  // MipMap<T> pyramid (mipmap.rs:38-45)
    uint64_t texel_base;        // first float of this image in TexTables::texel_pool; `channels` floats per texel, levels concatenated
    uint32_t level_offset[16];  // in texels
    uint32_t width, height, n_levels, channels;
};
#define RSPT_TEX_SLOTS 4        // distinct textures one material may bind to lobe colours
#define RSPT_EWA_LUT 128        // WEIGHT_LUT_SIZE (mipmap.rs:21)
#define RSPT_EWA_MAX_SPAN 256   // guard on the EWA footprint scan (reference: unbounded)

struct TexTables {
    const rspt_texture* textures;
    const ImageDev* images;
    const float* texel_pool;       // all pyramids in one allocation: texel addresses derive from this kernel argument, so the
                                   // fetches are global loads (a pointer read from the image table would make them flat loads)
    const float* ewa_lut;          // host-built with expf as MipMap::new does (mipmap.rs:186-192)
    const uint32_t* mat_slots;     // [material][RSPT_TEX_SLOTS]: texture index or 0xffffffff
    const uint8_t* mat_flags;      // bit 0: some lobe colour is textured; bit 1: bump map; bit 2: dynamic
    const rspt_mat::DynMaterial* dyn;  // [material], where bit 2 is set
};
#define RSPT_SLOT_ALPHA 0x80000000u     // slot descriptor flags: the texture drives a lobe alpha ...
#define RSPT_SLOT_REMAP 0x40000000u     // ... through roughness_to_alpha
#define RSPT_SLOT_NODIFF 0x20000000u    // the m2 side of a MixMaterial: evaluated at an interaction without ray differentials (mixmat.rs:58-69)
#define RSPT_SLOT_TEX_MASK 0x1fffffffu
#define RSPT_MAT_TEXTURED 1u
#define RSPT_MAT_BUMP 2u
#define RSPT_MAT_DYNAMIC 4u             // the lobe list is built per hit from the raw parameter values in rows RSPT_TEX_ROWS .. (material_assembly.h)
// per-path results of k_texture, SoA [row][path]: rows 0..3 = clamp(texture value) of the material's
// slots; row 4 = (bumped shading.n, flags: bit 0 = bump applied, bits 8..15 = lobes dropped as black);
// row 5 = bumped shading.dpdu
#define RSPT_TEX_ROWS 6

// what Texture::evaluate reads of the SurfaceInteraction
struct TexSurf {
    f3 p;
    f2 uv;
    float dudx, dvdx, dudy, dvdy;
    f3 dpdx, dpdy;
};

RDEV int64_t f2i64(float x) {  // Rust `as isize`: saturating, NaN -> 0
    if (x != x) return 0;
    if (x >= 9223372036854775808.0f) return INT64_MAX;
    if (x <= -9223372036854775808.0f) return INT64_MIN;
    return (int64_t)x;
}

// ---- MipMap<T>::texel / triangle / lookup / ewa: src/core/mipmap.rs:206-400 ----
RDEV rgb img_texel(const ImageDev& m, const float* texels, uint32_t wrap, uint32_t level, int64_t s, int64_t t) {
    uint32_t w = m.width >> level, h = m.height >> level;
    w = w ? w : 1u; h = h ? h : 1u;
    uint64_t ss, tt;
    if (wrap == RSPT_WRAP_REPEAT) { ss = (uint64_t)s % (uint64_t)w; tt = (uint64_t)t % (uint64_t)h; }  // mod_t(s as usize, u_size)
    else {  // Clamp, and Black's clamp-like branch (mipmap.rs:217-227)
        ss = (uint64_t)(s < 0 ? 0 : (s > (int64_t)w - 1 ? (int64_t)w - 1 : s));
        tt = (uint64_t)(t < 0 ? 0 : (t > (int64_t)h - 1 ? (int64_t)h - 1 : t));
    }
    const float* q = texels + (size_t)m.channels * ((size_t)m.level_offset[level] + tt * w + ss);
    return m.channels == 1 ? mkrgb(q[0]) : ldrgb(q);
}
RDEVN rgb img_triangle(const ImageDev& m, const float* texels, uint32_t wrap, uint32_t level, f2 st) {
    if (level > m.n_levels - 1) level = m.n_levels - 1;
    uint32_t w = m.width >> level, h = m.height >> level;
    w = w ? w : 1u; h = h ? h : 1u;
    float s = st.x * (float)w - 0.5f, t = st.y * (float)h - 0.5f;
    int64_t s0 = f2i64(floorf(s)), t0 = f2i64(floorf(t));
    float ds = s - (float)s0, dt = t - (float)t0;
    rgb tmp1 = img_texel(m, texels, wrap, level, s0 + 1, t0 + 1) * (ds * dt);
    rgb tmp2 = img_texel(m, texels, wrap, level, s0 + 1, t0) * (ds * (1.0f - dt));
    rgb tmp3 = img_texel(m, texels, wrap, level, s0, t0 + 1) * ((1.0f - ds) * dt);
    rgb tmp4 = img_texel(m, texels, wrap, level, s0, t0) * ((1.0f - ds) * (1.0f - dt));
    return tmp4 + tmp3 + tmp2 + tmp1;
}
RDEV rgb img_lookup_width(const ImageDev& m, const float* texels, uint32_t wrap, f2 st, float width) {  // lookup_pnt_flt :233-252
    float level = (float)m.n_levels - 1.0f + rspt_log2f(fmaxf(width, 1e-8f));
    if (level < 0.0f) return img_triangle(m, texels, wrap, 0, st);
    if (level >= (float)m.n_levels - 1.0f) return img_texel(m, texels, wrap, m.n_levels - 1, 0, 0);
    uint32_t il = (uint32_t)floorf(level);
    float delta = level - (float)il;
    rgb a = img_triangle(m, texels, wrap, il, st), b = img_triangle(m, texels, wrap, il + 1, st);
    return a * (1.0f - delta) + b * delta;
}
RDEVN rgb img_ewa(const ImageDev& m, const float* texels, const float* lut, uint32_t wrap, uint32_t level, f2 st, f2 dst0, f2 dst1) {  // :337-400
    if (level >= m.n_levels) return img_texel(m, texels, wrap, m.n_levels - 1, 0, 0);
    uint32_t w = m.width >> level, h = m.height >> level;
    w = w ? w : 1u; h = h ? h : 1u;
    float sx = st.x * (float)w - 0.5f, sy = st.y * (float)h - 0.5f;
    float d0x = dst0.x * (float)w, d0y = dst0.y * (float)h, d1x = dst1.x * (float)w, d1y = dst1.y * (float)h;
    float a = d0y * d0y + d1y * d1y + 1.0f;
    float b = -2.0f * (d0x * d0y + d1x * d1y);
    float c = d0x * d0x + d1x * d1x + 1.0f;
    float inv_f = 1.0f / (a * c - b * b * 0.25f);
    a *= inv_f; b *= inv_f; c *= inv_f;
    float det = -b * b + 4.0f * a * c;
    float inv_det = 1.0f / det;
    float u_sqrt = sqrtf(det * c), v_sqrt = sqrtf(a * det);
    int64_t s0 = f2i64(ceilf(sx - 2.0f * inv_det * u_sqrt)), s1 = f2i64(floorf(sx + 2.0f * inv_det * u_sqrt));
    int64_t t0 = f2i64(ceilf(sy - 2.0f * inv_det * v_sqrt)), t1 = f2i64(floorf(sy + 2.0f * inv_det * v_sqrt));
    if (s1 - s0 > RSPT_EWA_MAX_SPAN) s1 = s0 + RSPT_EWA_MAX_SPAN;
    if (t1 - t0 > RSPT_EWA_MAX_SPAN) t1 = t0 + RSPT_EWA_MAX_SPAN;
    rgb sum = mkrgb(0.0f);
    float sum_wts = 0.0f;
    for (int64_t it = t0; it <= t1; it++) {
        float tt = (float)it - sy;
        for (int64_t is = s0; is <= s1; is++) {
            float ss = (float)is - sx;
            float r2 = a * ss * ss + b * ss * tt + c * tt * tt;
            if (r2 < 1.0f) {
                float fi = r2 * (float)RSPT_EWA_LUT;
                uint32_t index = (fi != fi || fi <= 0.0f) ? 0u : (fi >= (float)(RSPT_EWA_LUT - 1) ? (uint32_t)(RSPT_EWA_LUT - 1) : (uint32_t)fi);
                float weight = lut[index];
                sum = sum + img_texel(m, texels, wrap, level, is, it) * weight;
                sum_wts += weight;
            }
        }
    }
    return sum / sum_wts;
}
RDEV rgb img_lookup(const ImageDev& m, const float* texels, const float* lut, const rspt_texture& tx, f2 st, f2 dst0, f2 dst1) {  // lookup_pnt_vec_vec :253-297
    if (tx.trilinear) {
        float width = fmaxf(fmaxf(fabsf(dst0.x), fabsf(dst0.y)), fmaxf(fabsf(dst1.x), fabsf(dst1.y)));
        return img_lookup_width(m, texels, tx.wrap, st, width);
    }
    if (dst0.x * dst0.x + dst0.y * dst0.y < dst1.x * dst1.x + dst1.y * dst1.y) { f2 tmp = dst0; dst0 = dst1; dst1 = tmp; }
    float major_length = sqrtf(dst0.x * dst0.x + dst0.y * dst0.y);
    float minor_length = sqrtf(dst1.x * dst1.x + dst1.y * dst1.y);
    if (minor_length * tx.max_aniso < major_length && minor_length > 0.0f) {
        float scale = major_length / (minor_length * tx.max_aniso);
        dst1.x *= scale; dst1.y *= scale;
        minor_length *= scale;
    }
    if (minor_length == 0.0f) return img_triangle(m, texels, tx.wrap, 0, st);
    float lod = fmaxf(0.0f, (float)m.n_levels - 1.0f + rspt_log2f(minor_length));
    uint32_t ilod = (uint32_t)floorf(lod);
    rgb col2 = img_ewa(m, texels, lut, tx.wrap, ilod + 1, st, dst0, dst1);
    rgb col1 = img_ewa(m, texels, lut, tx.wrap, ilod, st, dst0, dst1);
    float t = lod - (float)ilod;
    return col1 * (1.0f - t) + col2 * t;
}

// ---- Perlin noise (texture.rs:21-48 permutation = Ken Perlin's reference table, twice; 289-425) ----
__device__ const uint8_t NOISE_PERM[512] = {
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
RDEV int32_t f2i32(float x) {  // Rust `as i32`
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
RDEV float smooth_step(float mn, float mx, float value) {
    float v = (value - mn) / (mx - mn);
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    return v * v * (-2.0f * v + 3.0f);
}
RDEV float noise_grad(int32_t x, int32_t y, int32_t z, float dx, float dy, float dz) {
    uint32_t h = NOISE_PERM[NOISE_PERM[NOISE_PERM[x] + y] + z] & 15u;
    float u = (h < 8u || h == 12u || h == 13u) ? dx : dy;
    float v = (h < 4u || h == 12u || h == 13u) ? dy : dz;
    return ((h & 1u) ? -u : u) + ((h & 2u) ? -v : v);
}
RDEV float noise_weight(float t) { float t3 = t * t * t, t4 = t3 * t; return 6.0f * t4 * t - 15.0f * t4 + 10.0f * t3; }
__device__ __noinline__ float noise_flt(float x, float y, float z) {
    int32_t ix = f2i32(floorf(x)), iy = f2i32(floorf(y)), iz = f2i32(floorf(z));
    float dx = x - (float)ix, dy = y - (float)iy, dz = z - (float)iz;
    ix &= 255; iy &= 255; iz &= 255;
    float w000 = noise_grad(ix, iy, iz, dx, dy, dz), w100 = noise_grad(ix + 1, iy, iz, dx - 1.0f, dy, dz);
    float w010 = noise_grad(ix, iy + 1, iz, dx, dy - 1.0f, dz), w110 = noise_grad(ix + 1, iy + 1, iz, dx - 1.0f, dy - 1.0f, dz);
    float w001 = noise_grad(ix, iy, iz + 1, dx, dy, dz - 1.0f), w101 = noise_grad(ix + 1, iy, iz + 1, dx - 1.0f, dy, dz - 1.0f);
    float w011 = noise_grad(ix, iy + 1, iz + 1, dx, dy - 1.0f, dz - 1.0f), w111 = noise_grad(ix + 1, iy + 1, iz + 1, dx - 1.0f, dy - 1.0f, dz - 1.0f);
    float wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
    float x00 = lerpf(wx, w000, w100), x10 = lerpf(wx, w010, w110), x01 = lerpf(wx, w001, w101), x11 = lerpf(wx, w011, w111);
    float y0 = lerpf(wy, x00, x10), y1 = lerpf(wy, x01, x11);
    return lerpf(wz, y0, y1);
}
RDEV float log_2(float x) { return rspt_logf(x) * 1.44269504088896340736f; }  // pbrt.rs:153-156
__device__ __noinline__ float fbm(f3 p, f3 dpdx, f3 dpdy, float omega, int32_t max_octaves) {
    float l2 = fmaxf(len2(dpdx), len2(dpdy));
    float n = -1.0f - 0.5f * log_2(l2);
    n = n < 0.0f ? 0.0f : (n > (float)max_octaves ? (float)max_octaves : n);  // clamp_t
    int32_t n_int = f2i32(floorf(n));
    float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (int32_t i = 0; i < n_int; i++) {
        sum += o * noise_flt(p.x * lambda, p.y * lambda, p.z * lambda);
        lambda *= 1.99f;
        o *= omega;
    }
    float n_partial = n - (float)n_int;
    sum += o * smooth_step(0.3f, 0.7f, n_partial) * noise_flt(p.x * lambda, p.y * lambda, p.z * lambda);
    return sum;
}
__device__ __noinline__ float turbulence(f3 p, f3 dpdx, f3 dpdy, float omega, int32_t max_octaves) {
    float l2 = fmaxf(len2(dpdx), len2(dpdy));
    float n = -1.0f - 0.5f * log_2(l2);
    n = n < 0.0f ? 0.0f : (n > (float)max_octaves ? (float)max_octaves : n);
    float fn = floorf(n);
    uint32_t n_int = (fn != fn || fn <= 0.0f) ? 0u : (uint32_t)fn;  // `as usize` of a value in [0, max_octaves]
    float sum = 0.0f, lambda = 1.0f, o = 1.0f;
    for (uint32_t i = 0; i < n_int; i++) {
        sum += o * fabsf(noise_flt(p.x * lambda, p.y * lambda, p.z * lambda));
        lambda *= 1.99f;
        o *= omega;
    }
    float n_partial = n - (float)n_int;
    sum += o * lerpf(smooth_step(0.3f, 0.7f, n_partial), 0.2f, fabsf(noise_flt(p.x * lambda, p.y * lambda, p.z * lambda)));
    for (uint32_t i = n_int; i < (uint32_t)(max_octaves > 0 ? max_octaves : 0); i++) {
        sum += o * 0.2f;
        o *= omega;
    }
    return sum;
}

// ---- TextureMapping2D / 3D (texture.rs:51-283) ----
RDEV f2 map_sphere(const rspt_texture& tx, f3 p) {
    f3 v = normalize(xf_point(tx.world_to_texture, p) - f3{0.0f, 0.0f, 0.0f});
    return f2{spherical_theta(v) * RSPT_INV_PI, spherical_phi(v) * 0.15915494309189533577f};
}
RDEV f2 map_cylinder(const rspt_texture& tx, f3 p) {
    f3 v = normalize(xf_point(tx.world_to_texture, p) - f3{0.0f, 0.0f, 0.0f});
    return f2{RSPT_PI + rspt_atan2f(v.y, v.x) * 0.15915494309189533577f, v.z};
}
RDEV void wrap_dt(f2* d) {
    if (d->y > 0.5f) d->y = 1.0f - d->y;
    else if (d->y < -0.5f) d->y = -(d->y + 1.0f);
}
__device__ __noinline__ f2 tex_map2d_round(const rspt_texture& tx, const TexSurf& si, f2* dstdx, f2* dstdy) {
    if (tx.mapping == RSPT_MAP_SPHERICAL) {
        f2 st = map_sphere(tx, si.p);
        const float delta = 0.1f;
        const float inv = 1.0f / delta;   // `Vector2f / Float` multiplies by the reciprocal (geometry.rs:1281-1288): not the same bits as a division
        f2 sx = map_sphere(tx, si.p + si.dpdx * delta);
        *dstdx = f2{(sx.x - st.x) * inv, (sx.y - st.y) * inv};
        f2 sy = map_sphere(tx, si.p + si.dpdy * delta);
        *dstdy = f2{(sy.x - st.x) * inv, (sy.y - st.y) * inv};
        wrap_dt(dstdx); wrap_dt(dstdy);
        return st;
    }
    if (tx.mapping == RSPT_MAP_CYLINDRICAL) {
        f2 st = map_cylinder(tx, si.p);
        const float delta = 0.01f;
        const float inv = 1.0f / delta;   // (as above)
        f2 sx = map_cylinder(tx, si.p + si.dpdx * delta);
        *dstdx = f2{(sx.x - st.x) * inv, (sx.y - st.y) * inv};
        wrap_dt(dstdx);
        f2 sy = map_cylinder(tx, si.p + si.dpdy * delta);
        *dstdy = f2{(sy.x - st.x) * inv, (sy.y - st.y) * inv};
        wrap_dt(dstdy);
        return st;
    }
    *dstdx = *dstdy = f2{0.0f, 0.0f};
    return f2{0.0f, 0.0f};
}
RDEV f2 tex_map2d(const rspt_texture& tx, const TexSurf& si, f2* dstdx, f2* dstdy) {
    if (tx.mapping == RSPT_MAP_UV) {
        *dstdx = f2{si.dudx * tx.map[0], si.dvdx * tx.map[1]};
        *dstdy = f2{si.dudy * tx.map[0], si.dvdy * tx.map[1]};
        return f2{si.uv.x * tx.map[0] + tx.map[2], si.uv.y * tx.map[1] + tx.map[3]};
    }
    if (tx.mapping == RSPT_MAP_PLANAR) {
        f3 vs{tx.map[0], tx.map[1], tx.map[2]}, vt{tx.map[3], tx.map[4], tx.map[5]};
        *dstdx = f2{dot(si.dpdx, vs), dot(si.dpdx, vt)};
        *dstdy = f2{dot(si.dpdy, vs), dot(si.dpdy, vt)};
        return f2{tx.map[6] + dot(si.p, vs), tx.map[7] + dot(si.p, vt)};
    }
    return tex_map2d_round(tx, si, dstdx, dstdy);
}
RDEV f3 tex_map3d(const rspt_texture& tx, const TexSurf& si, f3* dpdx, f3* dpdy) {
    *dpdx = xf_vector(tx.world_to_texture, si.dpdx);
    *dpdy = xf_vector(tx.world_to_texture, si.dpdy);
    return xf_point(tx.world_to_texture, si.p);
}

// ---- Texture::evaluate (src/textures/*.rs).  DEPTH = levels of children still allowed below this node; the host
// rejects deeper graphs (rspt_scene_create), so the recursion is a fixed three-deep chain of functions ----
template <int DEPTH>
__device__ __noinline__ rgb tex_eval_d(const TexTables& tt, uint32_t ti, const TexSurf& si) {
    const rspt_texture& tx = tt.textures[ti];
    // which children this node needs (one call site below, so the code of a level exists once)
    uint32_t kid[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    uint32_t n_kids = 0;
    if (tx.kind == RSPT_TEX_SCALE) { kid[0] = tx.tex1; kid[1] = tx.tex2; n_kids = 2; }
    else if (tx.kind == RSPT_TEX_MIX) { kid[0] = tx.tex1; kid[1] = tx.tex2; kid[2] = tx.tex3; n_kids = 3; }
    else if (tx.kind == RSPT_TEX_CHECKERBOARD) {  // checkerboard.rs:32-42
        f2 dstdx, dstdy;
        f2 st = tex_map2d(tx, si, &dstdx, &dstdy);
        float fa = floorf(st.x), fb = floorf(st.y);
        uint32_t a = (fa != fa || fa <= 0.0f) ? 0u : (fa >= 4294967296.0f ? 0xffffffffu : (uint32_t)fa);  // `as u32`
        uint32_t b = (fb != fb || fb <= 0.0f) ? 0u : (fb >= 4294967296.0f ? 0xffffffffu : (uint32_t)fb);
        kid[0] = ((a + b) % 2u == 0u) ? tx.tex1 : tx.tex2;
        n_kids = 1;
    } else if (tx.kind == RSPT_TEX_DOTS) {  // dots.rs:31-72: tex1 = outside, tex2 = inside
        f2 dstdx, dstdy;
        f2 st = tex_map2d(tx, si, &dstdx, &dstdy);
        int32_t s_cell = f2i32(floorf(st.x + 0.5f)), t_cell = f2i32(floorf(st.y + 0.5f));
        kid[0] = tx.tex1;
        if (noise_flt((float)s_cell + 0.5f, (float)t_cell + 0.5f, 0.5f) > 0.0f) {
            const float radius = 0.35f, max_shift = 0.5f - radius;
            float s_center = (float)s_cell + max_shift * noise_flt((float)s_cell + 1.5f, (float)t_cell + 2.8f, 0.5f);
            float t_center = (float)t_cell + max_shift * noise_flt((float)s_cell + 4.5f, (float)t_cell + 9.8f, 0.5f);
            float dx = st.x - s_center, dy = st.y - t_center;
            if (dx * dx + dy * dy < radius * radius) kid[0] = tx.tex2;
        }
        n_kids = 1;
    }
    if (n_kids) {
        rgb v[3] = {mkrgb(0.0f), mkrgb(0.0f), mkrgb(0.0f)};
        if constexpr (DEPTH > 0) {
#pragma unroll 1
            for (uint32_t k = 0; k < n_kids; k++) {
                rgb r = tex_eval_d<DEPTH - 1>(tt, kid[k], si);
                if (k == 0) v[0] = r; else if (k == 1) v[1] = r; else v[2] = r;
            }
        }
        if (tx.kind == RSPT_TEX_SCALE) return v[0] * v[1];                                      // scale.rs:24-27
        if (tx.kind == RSPT_TEX_MIX) return v[0] * mkrgb(1.0f - v[2].r) + v[1] * mkrgb(v[2].r);  // mix.rs:30-35
        return v[0];
    }
    switch (tx.kind) {
    case RSPT_TEX_CONSTANT: return ldrgb(tx.value);
    case RSPT_TEX_IMAGE: {
        f2 dstdx, dstdy;
        f2 st = tex_map2d(tx, si, &dstdx, &dstdy);
        const ImageDev& im = tt.images[tx.image];
        return img_lookup(im, tt.texel_pool + im.texel_base, tt.ewa_lut, tx, st, dstdx, dstdy);
    }
    case RSPT_TEX_FBM: { f3 dpdx, dpdy; f3 p = tex_map3d(tx, si, &dpdx, &dpdy); return mkrgb(fbm(p, dpdx, dpdy, tx.omega, tx.octaves)); }
    case RSPT_TEX_WRINKLED: { f3 dpdx, dpdy; f3 p = tex_map3d(tx, si, &dpdx, &dpdy); return mkrgb(turbulence(p, dpdx, dpdy, tx.omega, tx.octaves)); }
    case RSPT_TEX_WINDY: {  // windy.rs:23-37
        f3 dpdx, dpdy; f3 p = tex_map3d(tx, si, &dpdx, &dpdy);
        float wind_strength = fbm(p * 0.1f, dpdx * 0.1f, dpdy * 0.1f, 0.5f, 3);
        float wave_height = fbm(p, dpdx, dpdy, 0.5f, 6);
        return mkrgb(fabsf(wind_strength) * wave_height);
    }
    case RSPT_TEX_MARBLE: {  // marble.rs:43-92
        f3 dpdx, dpdy; f3 p = tex_map3d(tx, si, &dpdx, &dpdy);
        p = p * tx.scale;
        float marble = p.y + tx.variation * fbm(p, dpdx * tx.scale, dpdy * tx.scale, tx.omega, tx.octaves);
        float t = 0.5f + 0.5f * rspt_sinf(marble);
        float ff = floorf(t * 6.0f);
        uint32_t first = (ff != ff || ff <= 0.0f) ? 0u : (ff >= 5.0f ? 5u : (uint32_t)ff);
        t = t * 6.0f - (float)first;
        // control points c[first .. first + 3] of {A A A G W A A B A}: A = (.58 .58 .6), G = (.5 .5 .5), W = (.6 .59 .58), B = (.2 .2 .33)
        auto col = [](uint32_t k) { return k == 3u ? rgb{0.5f, 0.5f, 0.5f} : (k == 4u ? rgb{0.6f, 0.59f, 0.58f} : (k == 7u ? rgb{0.2f, 0.2f, 0.33f} : rgb{0.58f, 0.58f, 0.6f})); };
        rgb c0 = col(first), c1 = col(first + 1), c2 = col(first + 2), c3 = col(first + 3);
        rgb s0 = c0 * (1.0f - t) + c1 * t, s1 = c1 * (1.0f - t) + c2 * t, s2 = c2 * (1.0f - t) + c3 * t;
        s0 = s0 * (1.0f - t) + s1 * t;
        s1 = s1 * (1.0f - t) + s2 * t;
        return (s0 * (1.0f - t) + s1 * t) * 1.5f;
    }
    default: return mkrgb(0.0f);
    }
}
#define RSPT_TEX_MAX_DEPTH 3  // nodes on the longest path of a texture graph
RDEV rgb tex_eval(const TexTables& tt, uint32_t ti, const TexSurf& si) {
    const rspt_texture& tx = tt.textures[ti];
    if (tx.kind == RSPT_TEX_CONSTANT) return ldrgb(tx.value);
    if (tx.kind == RSPT_TEX_IMAGE && tx.mapping <= RSPT_MAP_PLANAR) {  // the common case stays inline: no call, no scratch frame
        f2 dstdx, dstdy;
        f2 st = tex_map2d(tx, si, &dstdx, &dstdy);
        const ImageDev& im = tt.images[tx.image];
        return img_lookup(im, tt.texel_pool + im.texel_base, tt.ewa_lut, tx, st, dstdx, dstdy);
    }
    // the callee takes references: hand it copies, so that the caller's own tables and surface record do not escape (an
    // escaped struct lives in scratch, and table pointers read back from scratch turn every access into a flat instruction)
    const TexTables tt_arg = tt;
    const TexSurf si_arg = si;
    return tex_eval_d<RSPT_TEX_MAX_DEPTH - 1>(tt_arg, ti, si_arg);
}

// Triangle::intersect's interaction (triangle.rs:274-448) with everything textures and bump mapping read
struct TexHit {
    f3 p, n;
    f2 uv;
    f3 dpdu, dpdv;                        // isect.dpdu / dpdv
    f3 sh_n, sh_dpdu, sh_dpdv, sh_dndu, sh_dndv;  // isect.shading
};
// Transform::transform_surface_interaction (transform.rs:815-860) on the full interaction of the texture stage
RDEV void inst_texhit(const InstDev& in, TexHit* h) {
    f3 p, pe;
    inst_point(in.m, in.m3, h->p, f3{0.0f, 0.0f, 0.0f}, &p, &pe);
    h->p = p;
    h->n = normalize(xf_normal(in.mi, h->n));
    h->dpdu = xf_vector(in.m, h->dpdu); h->dpdv = xf_vector(in.m, h->dpdv);
    f3 sn = normalize(xf_normal(in.mi, h->sh_n));
    h->sh_dpdu = xf_vector(in.m, h->sh_dpdu); h->sh_dpdv = xf_vector(in.m, h->sh_dpdv);
    h->sh_dndu = xf_normal(in.mi, h->sh_dndu); h->sh_dndv = xf_normal(in.mi, h->sh_dndv);
    h->sh_n = dot(sn, h->n) < 0.0f ? -sn : sn;
}
RDEVN void tri_fill_tex(const SceneDev& sc, uint32_t prim, const TriRec& t, float b0, float b1, float b2, TexHit* h) {
    f3 p0 = t.p0, p1 = t.p1, p2 = t.p2;
    f2 uv0{0.0f, 0.0f}, uv1{1.0f, 0.0f}, uv2{1.0f, 1.0f};  // triangle.rs:97-112
    const bool has_uv = (t.flags & MF_HAS_UV) && sc.UV;
    const bool has_n = (t.flags & MF_HAS_N) && sc.N, has_s = (t.flags & MF_HAS_S) && sc.S;
    uint32_t v0 = 0, v1 = 0, v2 = 0;
    f3 pn0{0.0f, 0.0f, 0.0f}, pn1 = pn0, pn2 = pn0;
    const bool packed = sc.tri_nuv != nullptr && (has_uv || has_n);   // the primitive's own copy of its normals / uvs (SceneDev::tri_nuv)
    if (packed) {
        const float4* q = sc.tri_nuv + 5 * (size_t)prim;
        if (has_n) { const float4 a = q[0], b = q[1]; const float c = q[2].x; pn0 = f3{a.x, a.y, a.z}; pn1 = f3{a.w, b.x, b.y}; pn2 = f3{b.z, b.w, c}; }
        if (has_uv) { const float4 c = q[2], d = q[3]; uv0 = f2{c.y, c.z}; uv1 = f2{c.w, d.x}; uv2 = f2{d.y, d.z}; }
    }
    if (((has_uv || has_n) && !packed) || has_s) {
        rspt_prim pr = sc.prims[prim];
        v0 = pr.v[0]; v1 = pr.v[1]; v2 = pr.v[2];
    }
    if (has_uv && !packed) {
        uv0 = f2{sc.UV[2 * (size_t)v0], sc.UV[2 * (size_t)v0 + 1]};
        uv1 = f2{sc.UV[2 * (size_t)v1], sc.UV[2 * (size_t)v1 + 1]};
        uv2 = f2{sc.UV[2 * (size_t)v2], sc.UV[2 * (size_t)v2 + 1]};
    }
    if (has_n && !packed) { pn0 = ld3(sc.N, v0); pn1 = ld3(sc.N, v1); pn2 = ld3(sc.N, v2); }
    f2 duv02{uv0.x - uv2.x, uv0.y - uv2.y}, duv12{uv1.x - uv2.x, uv1.y - uv2.y};
    f3 dp02 = p0 - p2, dp12 = p1 - p2;
    float det = duv02.x * duv12.y - duv02.y * duv12.x;
    bool degenerate = fabsf(det) < 1e-8f;
    f3 dpdu{0.0f, 0.0f, 0.0f}, dpdv{0.0f, 0.0f, 0.0f};
    if (!degenerate) {
        float invdet = 1.0f / det;
        dpdu = (dp02 * duv12.y - dp12 * duv02.y) * invdet;
        dpdv = (dp02 * -duv12.x + dp12 * duv02.x) * invdet;
    }
    if (degenerate || len2(cross(dpdu, dpdv)) == 0.0f) coordinate_system(normalize(cross(p2 - p0, p1 - p0)), &dpdu, &dpdv);
    h->p = p0 * b0 + p1 * b1 + p2 * b2;
    h->uv = f2{uv0.x * b0 + uv1.x * b1 + uv2.x * b2, uv0.y * b0 + uv1.y * b1 + uv2.y * b2};
    f3 n = normalize(cross(dp02, dp12));
    if (t.flags & MF_FLIP) n = -n;
    f3 sh_n = n, sh_dpdu = dpdu, sh_dpdv = dpdv, dndu{0.0f, 0.0f, 0.0f}, dndv{0.0f, 0.0f, 0.0f};
    if (has_n || has_s) {
        f3 ns = n;
        f3 n0{0.0f, 0.0f, 0.0f}, n1 = n0, n2 = n0;
        if (has_n) {
            n0 = pn0; n1 = pn1; n2 = pn2;
            ns = n0 * b0 + n1 * b1 + n2 * b2;
            ns = len2(ns) > 0.0f ? normalize(ns) : n;
        }
        f3 ss;
        if (has_s) {
            ss = ld3(sc.S, v0) * b0 + ld3(sc.S, v1) * b1 + ld3(sc.S, v2) * b2;
            ss = len2(ss) > 0.0f ? normalize(ss) : normalize(dpdu);
        } else
            ss = normalize(dpdu);
        f3 ts = cross(ss, ns);
        if (len2(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
        else coordinate_system(ns, &ss, &ts);
        if (has_n && !degenerate) {  // triangle.rs:389-416
            f3 dn1 = n0 - n2, dn2 = n1 - n2;
            float inv_det = 1.0f / det;
            dndu = (dn1 * duv12.y - dn2 * duv02.y) * inv_det;
            dndv = (dn1 * -duv12.x + dn2 * duv02.x) * inv_det;
        }
        sh_n = normalize(cross(ss, ts));
        n = faceforward(n, sh_n);
        sh_dpdu = ss; sh_dpdv = ts;
    }
    h->n = n; h->dpdu = dpdu; h->dpdv = dpdv;
    h->sh_n = sh_n; h->sh_dpdu = sh_dpdu; h->sh_dpdv = sh_dpdv; h->sh_dndu = dndu; h->sh_dndv = dndv;
}

// SurfaceInteraction::compute_differentials (interaction.rs:388-479) for a ray with differentials
RDEV bool solve_2x2(float a00, float a01, float a10, float a11, float b0, float b1, float* x0, float* x1) {  // transform.rs:219-235
    float det = a00 * a11 - a01 * a10;
    if (fabsf(det) < 1e-10f) return false;
    *x0 = (a11 * b0 - a01 * b1) / det;
    *x1 = (a00 * b1 - a10 * b0) / det;
    return !(*x0 != *x0 || *x1 != *x1);
}
RDEVN void compute_differentials(const TexHit& h, f3 rx_o, f3 rx_d, f3 ry_o, f3 ry_d, TexSurf* s) {
    s->dudx = s->dvdx = s->dudy = s->dvdy = 0.0f;
    s->dpdx = s->dpdy = f3{0.0f, 0.0f, 0.0f};
    float d = dot(h.n, h.p);
    float tx = -(dot(h.n, rx_o) - d) / dot(h.n, rx_d);
    if (__builtin_isinf(tx) || tx != tx) return;
    f3 px = rx_o + rx_d * tx;
    float ty = -(dot(h.n, ry_o) - d) / dot(h.n, ry_d);
    if (__builtin_isinf(ty) || ty != ty) return;
    f3 py = ry_o + ry_d * ty;
    s->dpdx = px - h.p;
    s->dpdy = py - h.p;
    int d0, d1;
    if (fabsf(h.n.x) > fabsf(h.n.y) && fabsf(h.n.x) > fabsf(h.n.z)) { d0 = 1; d1 = 2; }
    else if (fabsf(h.n.y) > fabsf(h.n.z)) { d0 = 0; d1 = 2; }
    else { d0 = 0; d1 = 1; }
    float a00 = comp(h.dpdu, d0), a01 = comp(h.dpdv, d0), a10 = comp(h.dpdu, d1), a11 = comp(h.dpdv, d1);
    float bx0 = comp(px, d0) - comp(h.p, d0), bx1 = comp(px, d1) - comp(h.p, d1);
    float by0 = comp(py, d0) - comp(h.p, d0), by1 = comp(py, d1) - comp(h.p, d1);
    if (!solve_2x2(a00, a01, a10, a11, bx0, bx1, &s->dudx, &s->dvdx)) { s->dudx = 0.0f; s->dvdx = 0.0f; }
    if (!solve_2x2(a00, a01, a10, a11, by0, by1, &s->dudy, &s->dvdy)) { s->dudy = 0.0f; s->dvdy = 0.0f; }
}

// PerspectiveCamera::generate_ray_differential's offset rays (perspective.rs:205-220, 245-271), transformed
// (transform.rs:550-556) and scaled by 1 / sqrt(spp) (integrator.rs:140-144; geometry.rs:2398-2405)
RDEVN void camera_differentials(const RenderDev& rd, f2 p_film, f3 p_lens, f3 ray_o, f3 ray_d, f3* rx_o, f3* rx_d, f3* ry_o, f3* ry_d) {
    float c2w[16];
    camera_to_world_at(rd, p_lens.z, c2w);   // the matrix the ray itself was transformed with (Transform::transform_ray, transform.rs:550-556)
    f3 p_camera = xf_point(rd.raster_to_camera, f3{p_film.x, p_film.y, 0.0f});
    f3 c0 = xf_point(rd.raster_to_camera, f3{0.0f, 0.0f, 0.0f});
    f3 dx_camera = xf_point(rd.raster_to_camera, f3{1.0f, 0.0f, 0.0f}) - c0;
    f3 dy_camera = xf_point(rd.raster_to_camera, f3{0.0f, 1.0f, 0.0f}) - c0;
    f3 ox{0.0f, 0.0f, 0.0f}, oy = ox;
    f3 dx = normalize(p_camera + dx_camera), dy = normalize(p_camera + dy_camera);
    if (rd.lens_radius > 0.0f) {
        f2 pl = concentric_disk(f2{p_lens.x, p_lens.y});
        pl = f2{pl.x * rd.lens_radius, pl.y * rd.lens_radius};
        float ftx = rd.focal_distance / dx.z;
        f3 pfx = f3{0.0f, 0.0f, 0.0f} + dx * ftx;
        ox = f3{pl.x, pl.y, 0.0f};
        dx = normalize(pfx - ox);
        float fty = rd.focal_distance / dy.z;
        f3 pfy = f3{0.0f, 0.0f, 0.0f} + dy * fty;
        oy = f3{pl.x, pl.y, 0.0f};
        dy = normalize(pfy - oy);
    }
    f3 wox = xf_point(c2w, ox), woy = xf_point(c2w, oy);
    f3 wdx = xf_vector(c2w, dx), wdy = xf_vector(c2w, dy);
    float s = 1.0f / sqrtf((float)rd.spp);
    *rx_o = ray_o + (wox - ray_o) * s; *ry_o = ray_o + (woy - ray_o) * s;
    *rx_d = ray_d + (wdx - ray_d) * s; *ry_d = ray_d + (wdy - ray_d) * s;
}

// Material::bump (material.rs:116-219) followed by set_shading_geometry(.., false) (interaction.rs:345-370; si.shape is
// None for triangles, so no orientation flip)
RDEVN void bump_map(const TexTables& tt, uint32_t ti, const TexHit& h, const TexSurf& s, f3* sh_n_out, f3* sh_dpdu_out) {
    TexSurf ev = s;
    float du = 0.5f * (fabsf(s.dudx) + fabsf(s.dudy));
    if (du == 0.0f) du = 0.0005f;
    ev.p = h.p + h.sh_dpdu * du;
    ev.uv = f2{s.uv.x + du, s.uv.y + 0.0f};
    float u_displace = tex_eval(tt, ti, ev).r;
    float dv = 0.5f * (fabsf(s.dvdx) + fabsf(s.dvdy));
    if (dv == 0.0f) dv = 0.0005f;
    ev.p = h.p + h.sh_dpdv * dv;
    ev.uv = f2{s.uv.x + 0.0f, s.uv.y + dv};
    float v_displace = tex_eval(tt, ti, ev).r;
    float displace = tex_eval(tt, ti, s).r;
    f3 dpdu = h.sh_dpdu + h.sh_n * ((u_displace - displace) / du) + h.sh_dndu * displace;
    f3 dpdv = h.sh_dpdv + h.sh_n * ((v_displace - displace) / dv) + h.sh_dndv * displace;
    f3 n = normalize(cross(dpdu, dpdv));
    *sh_n_out = faceforward(n, h.n);
    *sh_dpdu_out = dpdu;
}

}  // namespace rspt
