// TEST INFRASTRUCTURE — CPU oracle (see orc_math.hpp header).  Material::compute_scattering_functions
// (src/core/material.rs:63-113) restated material by material from src/materials/*.rs: every parameter texture is evaluated at
// the hit, clamped, and the BxDFs are pushed in the reference's order behind the reference's `is_black` guards.  Parity unpinned
// (the reference output that exists here, tests/test_reference_pin.py, reaches MatteMaterial with sigma = 0 only); what this file pins is the LIBRARY's host-side assembly (rs_pbrt_amd/csrc/
// material_assembly.h), which folds constant parameters once per material: the two are written independently from the same Rust
// and compared lobe by lobe in tests/test_materials.py and sample by sample in the GPU render tests.
#pragma once
#include "orc_bsdf.hpp"

namespace orc {

// Option<Spectrum> scale handed down by MixMaterial (mixmat.rs:70-75)
struct ScaleOpt {
    bool some = false;
    Spec sc;
};

// every material parameter is a texture (TextureParams::get_spectrum_texture / get_float_texture wrap literals in a
// ConstantTexture, paramset.rs:622-735); ref = 1 + texture index
static inline Spec param_spectrum(const Scene& sc, uint32_t ref, const Interaction& si) { return tex_eval(sc, ref - 1u, si); }
static inline Float param_float(const Scene& sc, uint32_t ref, const Interaction& si) { return tex_eval(sc, ref - 1u, si).c[0]; }

// ---- the BxDF constructors (`::new`) of reflection.rs, as records ----
static inline rspt_bxdf bxdf_blank(uint32_t type, const ScaleOpt& s) {
    rspt_bxdf b;
    std::memset(&b, 0, sizeof b);
    b.type = type;
    if (s.some) { b.has_sc = 1; b.sc[0] = s.sc.c[0]; b.sc[1] = s.sc.c[1]; b.sc[2] = s.sc.c[2]; }
    return b;
}
static inline void put3(float dst[3], const Spec& v) { dst[0] = v.c[0]; dst[1] = v.c[1]; dst[2] = v.c[2]; }
// TrowbridgeReitzDistribution::new (microfacet.rs:233-239)
static inline void put_distribution(rspt_bxdf* b, Float alpha_x, Float alpha_y) {
    b->alpha_x = std::fmax(alpha_x, 0.001f);
    b->alpha_y = std::fmax(alpha_y, 0.001f);
}
// TrowbridgeReitzDistribution::roughness_to_alpha (microfacet.rs:243-254)
static inline Float tr_roughness_to_alpha(Float roughness) {
    const Float limit = 1e-3f;
    if (limit > roughness) roughness = limit;
    const Float x = std::log(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
static inline rspt_bxdf lambertian_reflection(const Spec& r, const ScaleOpt& s) { // reflection.rs:960-963
    rspt_bxdf b = bxdf_blank(RSPT_BXDF_LAMBERT_R, s); put3(b.r, r); return b;
}
static inline rspt_bxdf lambertian_transmission(const Spec& t, const ScaleOpt& s) { // reflection.rs:1008-1011
    rspt_bxdf b = bxdf_blank(RSPT_BXDF_LAMBERT_T, s); put3(b.r, t); return b;
}
static inline rspt_bxdf oren_nayar(const Spec& r, Float sigma, const ScaleOpt& s) { // OrenNayar::new reflection.rs:1057-1066
    rspt_bxdf b = bxdf_blank(RSPT_BXDF_OREN_NAYAR, s);
    put3(b.r, r);
    sigma = (PI / 180.0f) * sigma; // radians(), pbrt.rs:144-146
    const Float sigma2 = sigma * sigma;
    b.on_a = 1.0f - (sigma2 / (2.0f * (sigma2 + 0.33f)));
    b.on_b = 0.45f * sigma2 / (sigma2 + 0.09f);
    return b;
}
struct FresnelRec { uint32_t kind; Float eta_i, eta_t; Spec c_eta, c_k; };
static inline FresnelRec fresnel_noop() { return FresnelRec{RSPT_FRESNEL_NOOP, 0, 0, Spec(), Spec()}; }
static inline FresnelRec fresnel_dielectric(Float eta_i, Float eta_t) { return FresnelRec{RSPT_FRESNEL_DIELECTRIC, eta_i, eta_t, Spec(), Spec()}; }
static inline FresnelRec fresnel_conductor(const Spec& eta_t, const Spec& k) { return FresnelRec{RSPT_FRESNEL_CONDUCTOR, 1.0f, 0, eta_t, k}; } // eta_i = Spectrum::new(1.0)
static inline void put_fresnel(rspt_bxdf* b, const FresnelRec& f) {
    b->fresnel = f.kind;
    if (f.kind == RSPT_FRESNEL_DIELECTRIC) { b->eta_a = f.eta_i; b->eta_b = f.eta_t; }
    if (f.kind == RSPT_FRESNEL_CONDUCTOR) { put3(b->c1, f.c_eta); put3(b->c2, f.c_k); }
}
static inline rspt_bxdf specular_reflection(const Spec& r, const FresnelRec& f, const ScaleOpt& s) { // reflection.rs:717-723
    rspt_bxdf b = bxdf_blank(RSPT_BXDF_SPECULAR_R, s); put3(b.r, r); put_fresnel(&b, f); return b;
}
static inline rspt_bxdf specular_transmission(const Spec& t, Float eta_a, Float eta_b, const ScaleOpt& s) { // reflection.rs:764-779 (mode: radiance)
    rspt_bxdf b = bxdf_blank(RSPT_BXDF_SPECULAR_T, s); put3(b.r, t); b.eta_a = eta_a; b.eta_b = eta_b; return b;
}
static inline rspt_bxdf fresnel_specular(const Spec& r, const Spec& t, Float eta_a, Float eta_b, const ScaleOpt& s) { // reflection.rs:851-867
    rspt_bxdf b = bxdf_blank(RSPT_BXDF_FRESNEL_SPEC, s); put3(b.r, r); put3(b.t, t); b.eta_a = eta_a; b.eta_b = eta_b; return b;
}
static inline rspt_bxdf microfacet_reflection(const Spec& r, Float ax, Float ay, const FresnelRec& f, const ScaleOpt& s) { // reflection.rs:1136-1148
    rspt_bxdf b = bxdf_blank(RSPT_BXDF_MICROFACET_R, s); put3(b.r, r); put_distribution(&b, ax, ay); put_fresnel(&b, f); return b;
}
static inline rspt_bxdf microfacet_transmission(const Spec& t, Float ax, Float ay, Float eta_a, Float eta_b, const ScaleOpt& s) { // reflection.rs:1225-1244
    rspt_bxdf b = bxdf_blank(RSPT_BXDF_MICROFACET_T, s); put3(b.r, t); put_distribution(&b, ax, ay); b.eta_a = eta_a; b.eta_b = eta_b; return b;
}
static inline rspt_bxdf fresnel_blend(const Spec& rd, const Spec& rs, Float ax, Float ay, const ScaleOpt& s) { // reflection.rs:1383-1395
    rspt_bxdf b = bxdf_blank(RSPT_BXDF_FRESNEL_BLEND, s); put3(b.r, rd); put3(b.t, rs); put_distribution(&b, ax, ay); return b;
}

static inline void compute_scattering_functions(const Scene& sc, Interaction& si, uint32_t mi, bool allow_multiple_lobes, const ScaleOpt& scale, Bsdf* bsdf);

// src/materials/matte.rs:43-86
static inline void matte_csf(const Scene& sc, const rspt_material_desc& m, Interaction& si, const ScaleOpt& scale, Bsdf* bsdf) {
    if (m.bumpmap) bump(sc, m.bumpmap - 1u, &si);
    const Spec r = sclamp0(param_spectrum(sc, m.kd, si));
    const Float sig = clamp_t(param_float(sc, m.sigma, si), 0.0f, 90.0f);
    bsdf->init(si, 1.0f);
    if (!r.is_black()) {
        if (sig == 0.0f) bsdf->add(lambertian_reflection(r, scale));
        else bsdf->add(oren_nayar(r, sig, scale));
    }
}
// src/materials/plastic.rs:57-125
static inline void plastic_csf(const Scene& sc, const rspt_material_desc& m, Interaction& si, const ScaleOpt& scale, Bsdf* bsdf) {
    if (m.bumpmap) bump(sc, m.bumpmap - 1u, &si);
    const Spec kd = sclamp0(param_spectrum(sc, m.kd, si));
    const Spec ks = sclamp0(param_spectrum(sc, m.ks, si));
    Float rough = param_float(sc, m.roughness, si);
    bsdf->init(si, 1.0f);
    if (!kd.is_black()) bsdf->add(lambertian_reflection(kd, scale));
    if (!ks.is_black()) {
        const FresnelRec fresnel = fresnel_dielectric(1.5f, 1.0f);
        if (m.remap_roughness) rough = tr_roughness_to_alpha(rough);
        bsdf->add(microfacet_reflection(ks, rough, rough, fresnel, scale));
    }
}
// src/materials/mirror.rs:34-70
static inline void mirror_csf(const Scene& sc, const rspt_material_desc& m, Interaction& si, const ScaleOpt& scale, Bsdf* bsdf) {
    if (m.bumpmap) bump(sc, m.bumpmap - 1u, &si);
    const Spec r = sclamp0(param_spectrum(sc, m.kr, si));
    bsdf->init(si, 1.0f);
    bsdf->add(specular_reflection(r, fresnel_noop(), scale));
}
// src/materials/glass.rs:83-211
static inline void glass_csf(const Scene& sc, const rspt_material_desc& m, Interaction& si, bool allow_multiple_lobes, const ScaleOpt& scale, Bsdf* bsdf) {
    if (m.bumpmap) bump(sc, m.bumpmap - 1u, &si);
    Float urough = param_float(sc, m.uroughness, si);
    Float vrough = param_float(sc, m.vroughness, si);
    const Spec r = sclamp0(param_spectrum(sc, m.kr, si));
    const Spec t = sclamp0(param_spectrum(sc, m.kt, si));
    const bool is_specular = urough == 0.0f && vrough == 0.0f;
    const Float eta = param_float(sc, m.index, si);
    bsdf->init(si, eta);
    if (is_specular && allow_multiple_lobes) {
        bsdf->add(fresnel_specular(r, t, 1.0f, eta, scale));
    } else {
        if (m.remap_roughness) {
            urough = tr_roughness_to_alpha(urough);
            vrough = tr_roughness_to_alpha(vrough);
        }
        if (!r.is_black()) {
            const FresnelRec fresnel = fresnel_dielectric(1.0f, eta);
            if (is_specular) bsdf->add(specular_reflection(r, fresnel, scale));
            else bsdf->add(microfacet_reflection(r, urough, vrough, fresnel, scale));
        }
        if (!t.is_black()) {
            if (is_specular) bsdf->add(specular_transmission(t, 1.0f, eta, scale));
            else bsdf->add(microfacet_transmission(t, urough, vrough, 1.0f, eta, scale));
        }
    }
}
// src/materials/metal.rs:144-205
static inline void metal_csf(const Scene& sc, const rspt_material_desc& m, Interaction& si, const ScaleOpt& scale, Bsdf* bsdf) {
    if (m.bumpmap) bump(sc, m.bumpmap - 1u, &si);
    Float u_rough = m.uroughness ? param_float(sc, m.uroughness, si) : param_float(sc, m.roughness, si);
    Float v_rough = m.vroughness ? param_float(sc, m.vroughness, si) : param_float(sc, m.roughness, si);
    if (m.remap_roughness) {
        u_rough = tr_roughness_to_alpha(u_rough);
        v_rough = tr_roughness_to_alpha(v_rough);
    }
    const FresnelRec fr_mf = fresnel_conductor(param_spectrum(sc, m.eta, si), param_spectrum(sc, m.k, si));
    bsdf->init(si, 1.0f);
    bsdf->add(microfacet_reflection(Spec(1.0f), u_rough, v_rough, fr_mf, scale));
}
// src/materials/substrate.rs:62-114
static inline void substrate_csf(const Scene& sc, const rspt_material_desc& m, Interaction& si, const ScaleOpt& scale, Bsdf* bsdf) {
    if (m.bumpmap) bump(sc, m.bumpmap - 1u, &si);
    const Spec d = sclamp0(param_spectrum(sc, m.kd, si));
    const Spec s = sclamp0(param_spectrum(sc, m.ks, si));
    Float roughu = param_float(sc, m.uroughness, si);
    Float roughv = param_float(sc, m.vroughness, si);
    bsdf->init(si, 1.0f);
    if (!d.is_black() || !s.is_black()) {
        if (m.remap_roughness) {
            roughu = tr_roughness_to_alpha(roughu);
            roughv = tr_roughness_to_alpha(roughv);
        }
        bsdf->add(fresnel_blend(d, s, roughu, roughv, scale));
    }
}
// src/materials/uber.rs:114-259
static inline void uber_csf(const Scene& sc, const rspt_material_desc& m, Interaction& si, const ScaleOpt& scale, Bsdf* bsdf) {
    if (m.bumpmap) bump(sc, m.bumpmap - 1u, &si);
    const Float e = param_float(sc, m.index, si);
    const Spec op = sclamp0(param_spectrum(sc, m.opacity, si));
    const Spec t = sclamp0(Spec(1.0f) - op);
    const Spec kd = op * sclamp0(param_spectrum(sc, m.kd, si));
    const Spec ks = op * sclamp0(param_spectrum(sc, m.ks, si));
    Float u_rough = m.uroughness ? param_float(sc, m.uroughness, si) : param_float(sc, m.roughness, si);
    Float v_rough = m.vroughness ? param_float(sc, m.vroughness, si) : param_float(sc, m.roughness, si);
    const Spec kr = op * sclamp0(param_spectrum(sc, m.kr, si));
    const Spec kt = op * sclamp0(param_spectrum(sc, m.kt, si));
    if (!t.is_black()) bsdf->init(si, 1.0f);
    else bsdf->init(si, e);
    if (!t.is_black()) bsdf->add(specular_transmission(t, 1.0f, 1.0f, scale));
    if (!kd.is_black()) bsdf->add(lambertian_reflection(kd, scale));
    if (!ks.is_black()) {
        const FresnelRec fresnel = fresnel_dielectric(1.0f, e);
        if (m.remap_roughness) {
            u_rough = tr_roughness_to_alpha(u_rough);
            v_rough = tr_roughness_to_alpha(v_rough);
        }
        bsdf->add(microfacet_reflection(ks, u_rough, v_rough, fresnel, scale));
    }
    if (!kr.is_black()) bsdf->add(specular_reflection(kr, fresnel_dielectric(1.0f, e), scale));
    if (!kt.is_black()) bsdf->add(specular_transmission(kt, 1.0f, e, scale));
}
// src/materials/translucent.rs:64-189
static inline void translucent_csf(const Scene& sc, const rspt_material_desc& m, Interaction& si, const ScaleOpt& scale, Bsdf* bsdf) {
    if (m.bumpmap) bump(sc, m.bumpmap - 1u, &si);
    const Float eta = 1.5f;
    const Spec r = sclamp0(param_spectrum(sc, m.reflect, si));
    const Spec t = sclamp0(param_spectrum(sc, m.transmit, si));
    if (r.is_black() && t.is_black()) { bsdf->init(si, eta); return; }
    const Spec kd = sclamp0(param_spectrum(sc, m.kd, si));
    const Spec ks = sclamp0(param_spectrum(sc, m.ks, si));
    Float rough = param_float(sc, m.roughness, si);
    bsdf->init(si, eta);
    if (!kd.is_black()) {
        if (!r.is_black()) bsdf->add(lambertian_reflection(r * kd, scale));
        if (!t.is_black()) bsdf->add(lambertian_transmission(t * kd, scale));
    }
    if (!ks.is_black() && (!r.is_black() || !t.is_black())) {
        if (m.remap_roughness) rough = tr_roughness_to_alpha(rough);
        if (!r.is_black()) bsdf->add(microfacet_reflection(r * ks, rough, rough, fresnel_dielectric(1.0f, eta), scale));
        if (!t.is_black()) bsdf->add(microfacet_transmission(t * ks, rough, rough, 1.0f, eta, scale));
    }
}
// src/materials/mixmat.rs:43-305: m1 builds the Bsdf of `si` (its bump, its eta) under the scale s1; m2 builds one on a fresh
// SurfaceInteraction::new(p, p_error, uv, wo, dpdu, dpdv, dndu, dndv, time, shape) — no ray differentials, shading = geometry —
// under s2, and its BxDFs are re-created on the first Bsdf one by one (the long match of :76-300 copies every field).  The `_scale`
// this call is handed itself is ignored (:50).
static inline void mix_csf(const Scene& sc, const rspt_material_desc& m, Interaction& si, bool allow_multiple_lobes, Bsdf* bsdf) {
    const Spec s1 = sclamp0(param_spectrum(sc, m.amount, si));
    const Spec s2 = sclamp0(Spec(1.0f) - s1);
    Interaction si2 = si; // SurfaceInteraction::new (interaction.rs:249-330)
    si2.dudx = si2.dvdx = si2.dudy = si2.dvdy = 0.0f;
    si2.dpdx = si2.dpdy = V3{0, 0, 0};
    si2.sh_n = si2.n = normalize(cross(si.dpdu, si.dpdv)); // (flipped for reversed shapes; nothing reads it: m2's Bsdf frame is dropped below)
    si2.sh_dpdu = si.dpdu; si2.sh_dpdv = si.dpdv;
    si2.sh_dndu = V3{0, 0, 0}; si2.sh_dndv = V3{0, 0, 0}; // si.dndu / dndv stay zero for triangles (orc_scene.hpp Interaction)
    compute_scattering_functions(sc, si, m.m1, allow_multiple_lobes, ScaleOpt{true, s1}, bsdf);
    Bsdf bsdf2;
    compute_scattering_functions(sc, si2, m.m2, allow_multiple_lobes, ScaleOpt{true, s2}, &bsdf2);
    for (int i = 0; i < bsdf2.n; i++) bsdf->add(*bsdf2.lobes[i].b);
}

// Material::compute_scattering_functions (material.rs:63-113): dispatch on the material's kind
static inline void compute_scattering_functions(const Scene& sc, Interaction& si, uint32_t mi, bool allow_multiple_lobes, const ScaleOpt& scale, Bsdf* bsdf) {
    const rspt_material_desc& m = sc.d.materials[mi];
    switch (m.kind) {
    case RSPT_MAT_MATTE: matte_csf(sc, m, si, scale, bsdf); break;
    case RSPT_MAT_PLASTIC: plastic_csf(sc, m, si, scale, bsdf); break;
    case RSPT_MAT_MIRROR: mirror_csf(sc, m, si, scale, bsdf); break;
    case RSPT_MAT_GLASS: glass_csf(sc, m, si, allow_multiple_lobes, scale, bsdf); break;
    case RSPT_MAT_METAL: metal_csf(sc, m, si, scale, bsdf); break;
    case RSPT_MAT_SUBSTRATE: substrate_csf(sc, m, si, scale, bsdf); break;
    case RSPT_MAT_UBER: uber_csf(sc, m, si, scale, bsdf); break;
    case RSPT_MAT_TRANSLUCENT: translucent_csf(sc, m, si, scale, bsdf); break;
    case RSPT_MAT_MIX: mix_csf(sc, m, si, allow_multiple_lobes, bsdf); break;
    default: std::fprintf(stderr, "oracle: unknown material kind %u\n", m.kind); std::abort();
    }
}

// the Bsdf of a hit: what SurfaceInteraction::compute_scattering_functions leaves in isect.bsdf (interaction.rs:371-387)
static inline void make_bsdf(const Scene& sc, Interaction& si, uint32_t material, bool allow_multiple_lobes, Bsdf* bsdf) {
    compute_scattering_functions(sc, si, material, allow_multiple_lobes, ScaleOpt{}, bsdf);
}

} // namespace orc
