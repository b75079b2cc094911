// Device BSDF evaluation for the shade stage: Bsdf::{f, sample_f, pdf, num_components}
// (src/core/reflection.rs:223-446) over the pre-assembled lobe list of a material
// (include/rspt.h: rspt_bxdf), the BxDFs themselves (reflection.rs:711-1209) and the
// Trowbridge-Reitz microfacet distribution with visible-area sampling
// (src/core/microfacet.rs:225-353, 475-569).  All directions are in the shading frame
// (ss, ts, ns) unless suffixed _w.
#pragma once
#include "../../include/rspt.h"
#include "dev_math.h"

namespace rspt {

enum : uint32_t { BX_REFL = 1, BX_TRANS = 2, BX_DIFFUSE = 4, BX_GLOSSY = 8, BX_SPEC = 16, BX_ALL = 31 };

// Feature set of a shade-stage instantiation (kernels.h k_shade<F>): what the scene being rendered can put in front of the stage.
// rspt_scene_create knows every lobe type, light kind and mesh attribute of the scene; the host launches the narrowest compiled
// instantiation that covers them (librspt.hip shade_variant), and code for anything outside F folds away at compile time — the
// arithmetic of what remains is untouched, so the results are bit for bit those of the generic instantiation (SF_ALL).
enum : uint32_t {
    // bits 1 .. 9: 1 << RSPT_BXDF_*
    SF_CONDUCTOR = 1u << 10,   // a MicrofacetReflection / SpecularReflection with a conductor Fresnel (metal)
    SF_SC = 1u << 11,          // MixMaterial lobes (sc_opt)
    SF_L_AREA = 1u << 12, SF_L_POINT = 1u << 13, SF_L_SPOT = 1u << 14, SF_L_DISTANT = 1u << 15, SF_L_INFINITE = 1u << 16,
    SF_TEX = 1u << 17,         // textured materials (the texture stage ran)
    SF_INST = 1u << 18,        // object instances
    SF_NULL = 1u << 19,        // surfaces without a BSDF
    SF_HALTON = 1u << 20,      // the Halton sampler (else Sobol' from the LDS block)
    SF_VERTEX = 1u << 21,      // meshes with per-vertex normals / tangents / uvs
    SF_DYNAMIC = 1u << 22,     // materials whose lobe list is built per hit (material_assembly.h build_part)
    SF_SOBOL = 1u << 24,       // the Sobol' sampler; an instantiation with SF_HALTON and without this bit serves Halton renders only (no Sobol' block, no LDS tables)
    SF_ANIM = 1u << 23,        // moving object instances (dev_scene.h inst_at): only the all-features instantiation carries the interpolation
    SF_ALL = 0xffffffffu
};
#define RSPT_SF_LOBE(T) (1u << (T))

// trigonometry in the shading frame, reflection.rs:1801-1886
RDEV float cos2_t(f3 w) { return w.z * w.z; }
RDEV float sin2_t(f3 w) { return fmaxf(0.0f, 1.0f - cos2_t(w)); }
RDEV float sin_t(f3 w) { return sqrtf(sin2_t(w)); }
RDEV float tan_t(f3 w) { return sin_t(w) / w.z; }
RDEV float tan2_t(f3 w) { return sin2_t(w) / cos2_t(w); }
RDEV float cos_p(f3 w) {
    float s = sin_t(w);
    return s == 0.0f ? 1.0f : clampf(w.x / s, -1.0f, 1.0f);
}
RDEV float sin_p(f3 w) {
    float s = sin_t(w);
    return s == 0.0f ? 0.0f : clampf(w.y / s, -1.0f, 1.0f);
}
RDEV bool same_hemi(f3 a, f3 b) { return a.z * b.z > 0.0f; }

// reflection.rs:1897-1909
RDEV bool refract(f3 wi, f3 n, float eta, f3* wt) {
    float ci = dot(n, wi);
    float s2i = fmaxf(0.0f, 1.0f - ci * ci);
    float s2t = eta * eta * s2i;
    if (s2t >= 1.0f) return false;
    float ct = sqrtf(1.0f - s2t);
    *wt = (-wi) * eta + n * (eta * ci - ct);
    return true;
}
// reflection.rs:1920-1949
RDEV float fr_dielectric(float ci, float eta_i, float eta_t) {
    ci = clampf(ci, -1.0f, 1.0f);
    if (!(ci > 0.0f)) {
        float t = eta_i; eta_i = eta_t; eta_t = t;
        ci = fabsf(ci);
    }
    float si = sqrtf(fmaxf(0.0f, 1.0f - ci * ci));
    float st = eta_i / eta_t * si;
    if (st >= 1.0f) return 1.0f;
    float ct = sqrtf(fmaxf(0.0f, 1.0f - st * st));
    float r_parl = ((eta_t * ci) - (eta_i * ct)) / ((eta_t * ci) + (eta_i * ct));
    float r_perp = ((eta_i * ci) - (eta_t * ct)) / ((eta_i * ci) + (eta_t * ct));
    return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
}
// reflection.rs:1953-1972 with eta_i = 1
RDEV rgb fr_conductor(float ci, rgb eta_t, rgb k) {
    ci = clampf(ci, -1.0f, 1.0f);
    rgb one = mkrgb(1.0f);
    rgb eta = eta_t / one, eta_k = k / one;
    float c2 = ci * ci, s2 = 1.0f - c2;
    rgb eta2 = eta * eta, etak2 = eta_k * eta_k;
    rgb t0 = eta2 - etak2 - mkrgb(s2);
    rgb a2b2 = rsqrt3(t0 * t0 + eta2 * etak2 * mkrgb(4.0f));
    rgb t1 = a2b2 + mkrgb(c2);
    rgb a = rsqrt3((a2b2 + t0) * 0.5f);
    rgb t2 = a * 2.0f * ci;
    rgb rs = (t1 - t2) / (t1 + t2);
    rgb t3 = a2b2 * c2 + mkrgb(s2 * s2);
    rgb t4 = t2 * s2;
    rgb rp = rs * (t3 - t4) / (t3 + t4);
    return (rp + rs) * mkrgb(0.5f);
}

// sampling.rs:360-382, 214-221, 229-233
RDEV f2 concentric_disk(f2 u) {
    float ox = u.x * 2.0f - 1.0f, oy = u.y * 2.0f - 1.0f;
    if (ox == 0.0f && oy == 0.0f) return f2{0.0f, 0.0f};
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) {
        r = ox;
        theta = RSPT_PI_OVER_4 * (oy / ox);
    } else {
        r = oy;
        theta = RSPT_PI_OVER_2 - RSPT_PI_OVER_4 * (ox / oy);
    }
    return f2{rspt_cosf(theta) * r, rspt_sinf(theta) * r};
}
RDEV f3 cosine_hemisphere(f2 u) {
    f2 d = concentric_disk(u);
    float z = sqrtf(fmaxf(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return f3{d.x, d.y, z};
}
RDEV float pow5(float v) { return (v * v) * (v * v) * v; }  // reflection.rs:1974-1976
RDEV float power_heuristic(float f_pdf, float g_pdf) {  // nf = ng = 1
    float f = 1.0f * f_pdf, g = 1.0f * g_pdf;
    return (f * f) / (f * f + g * g);
}

// ---- TrowbridgeReitzDistribution, sample_visible_area = true ----
RDEV float tr_d(float ax, float ay, f3 wh) {
    float t2 = tan2_t(wh);
    if (__builtin_isinf(t2)) return 0.0f;
    float c4 = cos2_t(wh) * cos2_t(wh);
    float cp = cos_p(wh), sp = sin_p(wh);
    float e = ((cp * cp) / (ax * ax) + (sp * sp) / (ay * ay)) * t2;
    return 1.0f / (RSPT_PI * ax * ay * c4 * (1.0f + e) * (1.0f + e));
}
RDEV float tr_lambda(float ax, float ay, f3 w) {
    float att = fabsf(tan_t(w));
    if (__builtin_isinf(att)) return 0.0f;
    float cp = cos_p(w), sp = sin_p(w);
    float alpha = sqrtf((cp * cp) * ax * ax + (sp * sp) * ay * ay);
    float a2t2 = (alpha * att) * (alpha * att);
    return (-1.0f + sqrtf(1.0f + a2t2)) / 2.0f;
}
RDEV float tr_g1(float ax, float ay, f3 w) { return 1.0f / (1.0f + tr_lambda(ax, ay, w)); }
RDEV float tr_g(float ax, float ay, f3 wo, f3 wi) { return 1.0f / (1.0f + tr_lambda(ax, ay, wo) + tr_lambda(ax, ay, wi)); }
RDEV float tr_pdf(float ax, float ay, f3 wo, f3 wh) { return tr_d(ax, ay, wh) * tr_g1(ax, ay, wo) * absdot(wo, wh) / fabsf(wo.z); }
// microfacet.rs:475-531
RDEV void tr_sample11(float cos_th, float u1, float u2, float* sx, float* sy) {
    if (cos_th > 0.9999f) {
        float r = sqrtf(u1 / (1.0f - u1));
        float phi = RSPT_TAU * u2;
        *sx = r * rspt_cosf(phi);
        *sy = r * rspt_sinf(phi);
        return;
    }
    float sin_th = sqrtf(fmaxf(0.0f, 1.0f - cos_th * cos_th));
    float tan_th = sin_th / cos_th;
    float a = 1.0f / tan_th;
    float g1 = 2.0f / (1.0f + sqrtf(1.0f + 1.0f / (a * a)));
    a = 2.0f * u1 / g1 - 1.0f;
    float tmp = 1.0f / (a * a - 1.0f);
    if (tmp > 1e10f) tmp = 1e10f;
    float b = tan_th;
    float dd = sqrtf(fmaxf(b * b * tmp * tmp - (a * a - b * b) * tmp, 0.0f));
    float s1 = b * tmp - dd, s2 = b * tmp + dd;
    *sx = (a < 0.0f || s2 > 1.0f / tan_th) ? s1 : s2;
    float s, nu;
    if (u2 > 0.5f) { s = 1.0f; nu = 2.0f * (u2 - 0.5f); }
    else { s = -1.0f; nu = 2.0f * (0.5f - u2); }
    float z = (nu * (nu * (nu * 0.27385f - 0.73369f) + 0.46341f)) /
              (nu * (nu * (nu * 0.093073f + 0.309420f) - 1.0f) + 0.597999f);
    *sy = s * z * sqrtf(1.0f + *sx * *sx);
}
// microfacet.rs:533-569
RDEV f3 tr_sample(f3 wi, float ax, float ay, float u1, float u2) {
    f3 ws = normalize(f3{ax * wi.x, ay * wi.y, wi.z});
    float sx, sy;
    tr_sample11(ws.z, u1, u2, &sx, &sy);
    float cp = cos_p(ws), sp = sin_p(ws);
    float tmp = cp * sx - sp * sy;
    sy = sp * sx + cp * sy;
    sx = tmp;
    sx *= ax;
    sy *= ay;
    return normalize(f3{-sx, -sy, 1.0f});
}
RDEV f3 tr_sample_wh(float ax, float ay, f3 wo, f2 u) {
    if (wo.z < 0.0f) return -tr_sample(-wo, ax, ay, u.x, u.y);
    return tr_sample(wo, ax, ay, u.x, u.y);
}

// ---- one lobe (Bxdf enum, reflection.rs:462-633) ----
RDEV uint32_t lobe_type(uint32_t t) {
    switch (t) {
    case RSPT_BXDF_LAMBERT_R:
    case RSPT_BXDF_OREN_NAYAR: return BX_DIFFUSE | BX_REFL;
    case RSPT_BXDF_LAMBERT_T: return BX_DIFFUSE | BX_TRANS;
    case RSPT_BXDF_SPECULAR_R: return BX_REFL | BX_SPEC;
    case RSPT_BXDF_SPECULAR_T: return BX_TRANS | BX_SPEC;
    case RSPT_BXDF_FRESNEL_SPEC: return BX_REFL | BX_TRANS | BX_SPEC;
    case RSPT_BXDF_MICROFACET_R: return BX_REFL | BX_GLOSSY;
    case RSPT_BXDF_MICROFACET_T: return BX_TRANS | BX_GLOSSY;
    case RSPT_BXDF_FRESNEL_BLEND: return BX_REFL | BX_GLOSSY;
    }
    return 0;
}
RDEV bool lobe_matches(uint32_t type_bits, uint32_t flags) { return (type_bits & flags) == type_bits; }

template <uint32_t F = SF_ALL>
RDEV rgb lobe_fresnel(const rspt_bxdf& b, float ci) {  // Fresnel::evaluate :651-705
    if (b.fresnel == RSPT_FRESNEL_DIELECTRIC) return mkrgb(fr_dielectric(ci, b.eta_a, b.eta_b));
    if ((F & SF_CONDUCTOR) && b.fresnel == RSPT_FRESNEL_CONDUCTOR) return fr_conductor(ci, ldrgb(b.c1), ldrgb(b.c2));
    return mkrgb(1.0f);
}

// Lobe colours bound to textures (rspt_bxdf.tex_r / tex_t): k_texture left clamp(texture value) of the
// material's texture slots in per-path rows; the lobe's own r / t is the constant factor in front.
struct LobeTex {
    const float4* base;  // this path's entry of row 0 (nullptr: the material has no textured lobe)
    size_t stride;       // paths per row
};
template <uint32_t F = SF_ALL>
RDEV rgb lobe_r(const rspt_bxdf& b, const LobeTex& lt) {
    rgb r = ldrgb(b.r);
    if ((F & SF_TEX) && lt.base && b.tex_r) { float4 v = lt.base[(size_t)(b.tex_r - 1u) * lt.stride]; r = r * rgb{v.x, v.y, v.z}; }
    return r;
}
// alphas bound to roughness textures: k_texture left the final value (remapped, clamped to >= 0.001) in the slot
template <uint32_t F = SF_ALL>
RDEV float lobe_ax(const rspt_bxdf& b, const LobeTex& lt) { return ((F & SF_TEX) && lt.base && b.tex_ax) ? lt.base[(size_t)(b.tex_ax - 1u) * lt.stride].x : b.alpha_x; }
template <uint32_t F = SF_ALL>
RDEV float lobe_ay(const rspt_bxdf& b, const LobeTex& lt) { return ((F & SF_TEX) && lt.base && b.tex_ay) ? lt.base[(size_t)(b.tex_ay - 1u) * lt.stride].x : b.alpha_y; }
template <uint32_t F = SF_ALL>
RDEV rgb lobe_t(const rspt_bxdf& b, const LobeTex& lt) {
    rgb t = ldrgb(b.t);
    if ((F & SF_TEX) && lt.base && b.tex_t) { float4 v = lt.base[(size_t)(b.tex_t - 1u) * lt.stride]; t = t * rgb{v.x, v.y, v.z}; }
    return t;
}

// sc_opt of MixMaterial lobes: the reference writes `sc * A * B ...`, i.e. ((sc * A) * B) ...
template <uint32_t F = SF_ALL>
RDEV rgb lobe_scaled(const rspt_bxdf& b, rgb a) { return ((F & SF_SC) && b.has_sc) ? ldrgb(b.sc) * a : a; }

template <uint32_t F = SF_ALL>
RDEVN rgb lobe_f(const rspt_bxdf& b, const LobeTex& lt, f3 wo, f3 wi) {
    switch (b.type) {
    case RSPT_BXDF_LAMBERT_R: return lobe_scaled<F>(b, lobe_r<F>(b, lt)) * mkrgb(RSPT_INV_PI);
    case RSPT_BXDF_LAMBERT_T: return lobe_scaled<F>(b, lobe_r<F>(b, lt)) * RSPT_INV_PI;
    case RSPT_BXDF_OREN_NAYAR: {
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_OREN_NAYAR))) break;
        float sti = sin_t(wi), sto = sin_t(wo);
        float max_cos = 0.0f;
        if (sti > 1.0e-4f && sto > 1.0e-4f) {
            float d_cos = cos_p(wi) * cos_p(wo) + sin_p(wi) * sin_p(wo);
            max_cos = fmaxf(d_cos, 0.0f);
        }
        float sin_alpha, tan_beta;
        if (fabsf(wi.z) > fabsf(wo.z)) { sin_alpha = sto; tan_beta = sti / fabsf(wi.z); }
        else { sin_alpha = sti; tan_beta = sto / fabsf(wo.z); }
        return lobe_scaled<F>(b, lobe_r<F>(b, lt)) * mkrgb(RSPT_INV_PI * (b.on_a + b.on_b * max_cos * sin_alpha * tan_beta));
    }
    case RSPT_BXDF_MICROFACET_R: {
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_MICROFACET_R))) break;
        float cto = fabsf(wo.z), cti = fabsf(wi.z);
        f3 wh = wi + wo;
        if (cti == 0.0f || cto == 0.0f) return mkrgb(0.0f);
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return mkrgb(0.0f);
        wh = normalize(wh);
        rgb fr = lobe_fresnel<F>(b, dot(wi, wh));
        return lobe_scaled<F>(b, lobe_r<F>(b, lt)) * tr_d(lobe_ax<F>(b, lt), lobe_ay<F>(b, lt), wh) * tr_g(lobe_ax<F>(b, lt), lobe_ay<F>(b, lt), wo, wi) * fr / (4.0f * cti * cto);
    }
    case RSPT_BXDF_MICROFACET_T: {  // MicrofacetTransmission::f, TransportMode::Radiance (reflection.rs:1246-1317)
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_MICROFACET_T))) break;
        if (same_hemi(wo, wi)) return mkrgb(0.0f);
        float cto = wo.z, cti = wi.z;
        if (cto == 0.0f || cti == 0.0f) return mkrgb(0.0f);
        float eta = cto > 0.0f ? b.eta_b / b.eta_a : b.eta_a / b.eta_b;
        f3 wh = normalize(wo + wi * eta);
        if (wh.z < 0.0f) wh = -wh;
        if (dot(wo, wh) * dot(wi, wh) > 0.0f) return mkrgb(0.0f);
        rgb fr = mkrgb(fr_dielectric(dot(wo, wh), b.eta_a, b.eta_b));
        float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
        float factor = 1.0f / eta;
        return lobe_scaled<F>(b, mkrgb(1.0f) - fr) * lobe_r<F>(b, lt) *
               fabsf(tr_d(lobe_ax<F>(b, lt), lobe_ay<F>(b, lt), wh) * tr_g(lobe_ax<F>(b, lt), lobe_ay<F>(b, lt), wo, wi) * eta * eta * absdot(wi, wh) * absdot(wo, wh) * factor * factor /
                     (cti * cto * sqrt_denom * sqrt_denom));
    }
    case RSPT_BXDF_FRESNEL_BLEND: {  // FresnelBlend::f (reflection.rs:1398-1431): r = Rd, t = Rs
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_FRESNEL_BLEND))) break;
        rgb rd = lobe_r<F>(b, lt), rs = lobe_t<F>(b, lt);
        rgb diffuse = rd * (mkrgb(1.0f) - rs) * (28.0f / (23.0f * RSPT_PI)) * (1.0f - pow5(1.0f - 0.5f * fabsf(wi.z))) * (1.0f - pow5(1.0f - 0.5f * fabsf(wo.z)));
        f3 wh = wi + wo;
        if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return mkrgb(0.0f);
        wh = normalize(wh);
        rgb schlick = rs + (mkrgb(1.0f) - rs) * pow5(1.0f - dot(wi, wh));
        rgb specular = schlick * (tr_d(lobe_ax<F>(b, lt), lobe_ay<F>(b, lt), wh) / (4.0f * fabsf(dot(wi, wh)) * fmaxf(fabsf(wi.z), fabsf(wo.z))));
        return b.has_sc ? ldrgb(b.sc) * (diffuse + specular) : diffuse + specular;
    }
    default: break;
    }
    return mkrgb(0.0f);
}
template <uint32_t F = SF_ALL>
RDEVN float lobe_pdf(const rspt_bxdf& b, const LobeTex& lt, f3 wo, f3 wi) {
    switch (b.type) {
    case RSPT_BXDF_LAMBERT_R:
    case RSPT_BXDF_OREN_NAYAR:
    case RSPT_BXDF_SPECULAR_T:   // reflection.rs:828-834: cosine pdf, not 0
    case RSPT_BXDF_FRESNEL_SPEC: // reflection.rs:938-944: ditto
        return same_hemi(wo, wi) ? fabsf(wi.z) * RSPT_INV_PI : 0.0f;
    case RSPT_BXDF_LAMBERT_T: return !same_hemi(wo, wi) ? fabsf(wi.z) * RSPT_INV_PI : 0.0f;
    case RSPT_BXDF_MICROFACET_R: {
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_MICROFACET_R))) break;
        if (!same_hemi(wo, wi)) return 0.0f;
        f3 wh = normalize(wo + wi);
        return tr_pdf(lobe_ax<F>(b, lt), lobe_ay<F>(b, lt), wo, wh) / (4.0f * dot(wo, wh));
    }
    case RSPT_BXDF_MICROFACET_T: {  // reflection.rs:1350-1370
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_MICROFACET_T))) break;
        if (same_hemi(wo, wi)) return 0.0f;
        float eta = wo.z > 0.0f ? b.eta_b / b.eta_a : b.eta_a / b.eta_b;
        f3 wh = normalize(wo + wi * eta);
        float wo_wh = dot(wo, wh), wi_wh = dot(wi, wh);
        if (wo_wh * wi_wh > 0.0f) return 0.0f;
        float sqrt_denom = wo_wh + eta * wi_wh;
        float dwh_dwi = fabsf((eta * eta * wi_wh) / (sqrt_denom * sqrt_denom));
        return tr_pdf(lobe_ax<F>(b, lt), lobe_ay<F>(b, lt), wo, wh) * dwh_dwi;
    }
    case RSPT_BXDF_FRESNEL_BLEND: {  // reflection.rs:1462-1474
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_FRESNEL_BLEND))) break;
        if (!same_hemi(wo, wi)) return 0.0f;
        f3 wh = normalize(wo + wi);
        float pdf_wh = tr_pdf(lobe_ax<F>(b, lt), lobe_ay<F>(b, lt), wo, wh);
        return 0.5f * (fabsf(wi.z) * RSPT_INV_PI + pdf_wh / (4.0f * dot(wo, wh)));
    }
    default: break;
    }
    return 0.0f;
}
// sampled_type follows the reference's in/out sentinel convention (only written when non-zero).
// want_f = false: Bsdf::sample_f replaces the value of a non-specular lobe by the sum over all matching lobes
// (reflection.rs:393-410), so its own f() need not be evaluated (Q7: nothing else reads it)
template <uint32_t F = SF_ALL>
RDEVN rgb lobe_sample_f(const rspt_bxdf& b, const LobeTex& lt, f3 wo, f3* wi, f2 u, float* pdf, uint32_t* sampled_type, bool want_f) {
    const rgb black = mkrgb(0.0f);
    switch (b.type) {
    case RSPT_BXDF_LAMBERT_R:
    case RSPT_BXDF_OREN_NAYAR: {
        *wi = cosine_hemisphere(u);
        if (wo.z < 0.0f) wi->z *= -1.0f;
        *pdf = lobe_pdf<F>(b, lt, wo, *wi);
        return want_f ? lobe_f<F>(b, lt, wo, *wi) : mkrgb(0.0f);
    }
    case RSPT_BXDF_LAMBERT_T: {
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_LAMBERT_T))) break;
        *wi = cosine_hemisphere(u);
        if (wo.z > 0.0f) wi->z *= -1.0f;
        *pdf = lobe_pdf<F>(b, lt, wo, *wi);
        return want_f ? lobe_f<F>(b, lt, wo, *wi) : mkrgb(0.0f);
    }
    case RSPT_BXDF_SPECULAR_R: {
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_SPECULAR_R))) break;
        *wi = f3{-wo.x, -wo.y, wo.z};
        *pdf = 1.0f;
        return lobe_scaled<F>(b, lobe_fresnel<F>(b, wi->z)) * lobe_r<F>(b, lt) / fabsf(wi->z);
    }
    case RSPT_BXDF_SPECULAR_T: {
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_SPECULAR_T))) break;
        bool entering = wo.z > 0.0f;
        float ei = entering ? b.eta_a : b.eta_b, et = entering ? b.eta_b : b.eta_a;
        if (!refract(wo, faceforward(f3{0.0f, 0.0f, 1.0f}, wo), ei / et, wi)) return black;
        *pdf = 1.0f;
        rgb ft = lobe_r<F>(b, lt) * (mkrgb(1.0f) - mkrgb(fr_dielectric(wi->z, b.eta_a, b.eta_b)));
        ft = ft * mkrgb((ei * ei) / (et * et));
        return lobe_scaled<F>(b, ft) / fabsf(wi->z);
    }
    case RSPT_BXDF_FRESNEL_SPEC: {
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_FRESNEL_SPEC))) break;
        float fr = fr_dielectric(wo.z, b.eta_a, b.eta_b);
        if (u.x < fr) {
            *wi = f3{-wo.x, -wo.y, wo.z};
            if (*sampled_type != 0) *sampled_type = BX_REFL | BX_SPEC;
            *pdf = fr;
            return lobe_scaled<F>(b, lobe_r<F>(b, lt)) * fr / fabsf(wi->z);
        }
        bool entering = wo.z > 0.0f;
        float ei = entering ? b.eta_a : b.eta_b, et = entering ? b.eta_b : b.eta_a;
        if (!refract(wo, faceforward(f3{0.0f, 0.0f, 1.0f}, wo), ei / et, wi)) return black;
        rgb ft = lobe_t<F>(b, lt) * (1.0f - fr);
        ft = ft * mkrgb((ei * ei) / (et * et));
        if (*sampled_type != 0) *sampled_type = BX_TRANS | BX_SPEC;
        *pdf = 1.0f - fr;
        return lobe_scaled<F>(b, ft) / fabsf(wi->z);
    }
    case RSPT_BXDF_MICROFACET_R: {
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_MICROFACET_R))) break;
        if (wo.z == 0.0f) return black;
        f3 wh = tr_sample_wh(lobe_ax<F>(b, lt), lobe_ay<F>(b, lt), wo, u);
        *wi = (-wo) + wh * 2.0f * dot(wo, wh);  // reflect, reflection.rs:1889
        if (!same_hemi(wo, *wi)) return black;
        *pdf = tr_pdf(lobe_ax<F>(b, lt), lobe_ay<F>(b, lt), wo, wh) / (4.0f * dot(wo, wh));
        return want_f ? lobe_f<F>(b, lt, wo, *wi) : mkrgb(0.0f);
    }
    case RSPT_BXDF_MICROFACET_T: {  // reflection.rs:1322-1349
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_MICROFACET_T))) break;
        if (wo.z == 0.0f) return black;
        f3 wh = tr_sample_wh(lobe_ax<F>(b, lt), lobe_ay<F>(b, lt), wo, u);
        float eta = wo.z > 0.0f ? b.eta_a / b.eta_b : b.eta_b / b.eta_a;
        if (!refract(wo, wh, eta, wi)) return black;
        *pdf = lobe_pdf<F>(b, lt, wo, *wi);
        return want_f ? lobe_f<F>(b, lt, wo, *wi) : mkrgb(0.0f);
    }
    case RSPT_BXDF_FRESNEL_BLEND: {  // reflection.rs:1432-1461
        if (!(F & RSPT_SF_LOBE(RSPT_BXDF_FRESNEL_BLEND))) break;
        f2 uu = u;
        if (uu.x < 0.5f) {
            uu.x = fminf(2.0f * uu.x, RSPT_ONE_MINUS_EPS);
            *wi = cosine_hemisphere(uu);
            if (wo.z < 0.0f) wi->z *= -1.0f;
        } else {
            uu.x = fminf(2.0f * (uu.x - 0.5f), RSPT_ONE_MINUS_EPS);
            f3 wh = tr_sample_wh(lobe_ax<F>(b, lt), lobe_ay<F>(b, lt), wo, uu);
            *wi = (-wo) + wh * 2.0f * dot(wo, wh);
            if (!same_hemi(wo, *wi)) return black;
        }
        *pdf = lobe_pdf<F>(b, lt, wo, *wi);
        return want_f ? lobe_f<F>(b, lt, wo, *wi) : mkrgb(0.0f);
    }
    default: break;
    }
    return black;
}

// Bsdf (reflection.rs:223-446): frame + lobe slice
struct Bsdf {
    f3 ns, ng, ss, ts;
    float eta;
    const rspt_bxdf* lobes;
    uint32_t n;
    LobeTex lt;        // texture values of this hit (k_texture), or {nullptr, 0}
    uint32_t dropped;  // bit i: lobe i's textured colour came out black, the reference never added it

    // BxdfType of lobe i; a dropped lobe matches no flag set (bit 8 lies outside BSDF_ALL)
    RDEV uint32_t ltype(uint32_t i) const { return ((dropped >> i) & 1u) ? 0x100u : lobe_type(lobes[i].type); }

    RDEV f3 to_local(f3 v) const { return f3{dot(v, ss), dot(v, ts), dot(v, ns)}; }
    RDEV f3 to_world(f3 v) const {
        return f3{ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z,
                  ss.z * v.x + ts.z * v.y + ns.z * v.z};
    }
    RDEV int num_components(uint32_t flags) const {
        int c = 0;
        for (uint32_t i = 0; i < n; i++) c += lobe_matches(ltype(i), flags) ? 1 : 0;
        return c;
    }
    template <uint32_t F = SF_ALL>
    RDEVN rgb sum_f(f3 wo, f3 wi, bool refl, uint32_t flags) const {
        rgb f = mkrgb(0.0f);
        for (uint32_t i = 0; i < n; i++) {
            uint32_t t = ltype(i);
            if (lobe_matches(t, flags) && ((refl && (t & BX_REFL)) || (!refl && (t & BX_TRANS)))) f = f + lobe_f<F>(lobes[i], lt, wo, wi);
        }
        return f;
    }
    template <uint32_t F = SF_ALL>
    RDEVN rgb f(f3 wo_w, f3 wi_w, uint32_t flags) const {  // :274-297
        f3 wi = to_local(wi_w), wo = to_local(wo_w);
        if (wo.z == 0.0f) return mkrgb(0.0f);
        bool refl = (dot(wi_w, ng) * dot(wo_w, ng)) > 0.0f;
        return sum_f<F>(wo, wi, refl, flags);
    }
    template <uint32_t F = SF_ALL>
    RDEVN float pdf(f3 wo_w, f3 wi_w, uint32_t flags) const {  // :421-446
        if (n == 0) return 0.0f;
        f3 wo = to_local(wo_w), wi = to_local(wi_w);
        if (wo.z == 0.0f) return 0.0f;
        float p = 0.0f;
        int matching = 0;
        for (uint32_t i = 0; i < n; i++)
            if (lobe_matches(ltype(i), flags)) { matching++; p += lobe_pdf<F>(lobes[i], lt, wo, wi); }
        return matching > 0 ? p / (float)matching : 0.0f;
    }
    template <uint32_t F = SF_ALL>
    RDEVN rgb sample_f(f3 wo_w, f3* wi_w, f2 u, float* pdf_out, uint32_t flags, uint32_t* sampled_type) const {  // :298-420
        return sample_f_if<F>(wo_w, wi_w, u, pdf_out, flags, sampled_type, [](f3) { return false; });
    }
    // sample_f with an early way out: once the sampled direction is known, `useless(wi)` may declare the sample without any effect on the caller — the value and the
    // pdf of the OTHER lobes are then not evaluated and the function answers as for a zero pdf (black, *pdf_out = 0).  estimate_direct's BSDF-sampled term
    // (integrator.rs:480-568) is such a caller: it counts only when the direction meets the light's own triangle, a few times in 10^4 (round 6).
    template <uint32_t F = SF_ALL, class Useless>
    RDEVN rgb sample_f_if(f3 wo_w, f3* wi_w, f2 u, float* pdf_out, uint32_t flags, uint32_t* sampled_type, Useless&& useless) const {
        const rgb black = mkrgb(0.0f);
        int matching = num_components(flags);
        if (matching == 0) { *pdf_out = 0.0f; *sampled_type = 0; return black; }
        float fc = floorf(u.x * (float)matching);
        int ci = (fc != fc || fc <= 0.0f) ? 0 : (fc >= 255.0f ? 255 : (int)fc);  // `as u8`
        int comp = ci < matching - 1 ? ci : matching - 1;
        int idx = -1, count = comp;
        for (uint32_t i = 0; i < n; i++) {
            bool m = lobe_matches(ltype(i), flags);
            if (m && count == 0) { idx = (int)i; break; }
            else if (m) count -= 1;
        }
        if (idx < 0) return black;
        const rspt_bxdf& bx = lobes[idx];
        uint32_t bt = ltype((uint32_t)idx);
        f2 ur{fminf(u.x * (float)matching - (float)comp, RSPT_ONE_MINUS_EPS), u.y};
        f3 wi{0.0f, 0.0f, 0.0f};
        f3 wo = to_local(wo_w);
        if (wo.z == 0.0f) return black;
        *pdf_out = 0.0f;
        if (*sampled_type != 0) *sampled_type = bt;
        rgb f = lobe_sample_f<F>(bx, lt, wo, &wi, ur, pdf_out, sampled_type, false);
        if (*pdf_out == 0.0f) { if (*sampled_type != 0) *sampled_type = 0; return black; }
        *wi_w = to_world(wi);
        if (useless(*wi_w)) { *pdf_out = 0.0f; return black; }
        if (!(bt & BX_SPEC) && matching > 1)
            for (uint32_t i = 0; i < n; i++)
                if ((int)i != idx && lobe_matches(ltype(i), flags)) *pdf_out += lobe_pdf<F>(lobes[i], lt, wo, wi);
        if (matching > 1) *pdf_out /= (float)matching;
        if (!(bt & BX_SPEC)) {
            bool refl = dot(*wi_w, ng) * dot(wo_w, ng) > 0.0f;
            f = sum_f<F>(wo, wi, refl, flags);
        }
        return f;
    }
};

}  // namespace rspt
