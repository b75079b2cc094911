// Material::compute_scattering_functions (src/core/material.rs:63-113 and src/materials/*.rs) for the library: turns the material
// records of the ABI (rspt_material_desc: a kind and one texture reference per parameter) into the lobe lists the shade stage runs.
//
// The reference evaluates every parameter texture at every hit and then decides which BxDFs to push.  Here
//   * a parameter bound to a ConstantTexture is folded once per material on the host (clamp, black guards, roughness remapping,
//     OrenNayar's A / B, the uber opacity products);
//   * Kd / Ks / roughness bound to any other texture leave a reference on the lobe (tex_r / tex_t / tex_ax / tex_ay) that the texture
//     stage resolves per hit — the lobe list keeps its shape, only factors change (kernels.h texture_path: the fast path);
//   * a varying parameter that decides the SHAPE of the lobe list — which lobes exist, their type, Bsdf.eta: sigma, index, opacity,
//     Kr / Kt, reflect / transmit, a mix amount, the conductor's eta / k, a glass roughness — makes the material DYNAMIC: the texture
//     stage leaves the raw value of every varying parameter in the path's rows and the shade stage runs build_part — the same function
//     the host folds constants with — per hit (kernels.h dynamic_lobes).
// Plain C++ where the host uses it (rspt_material_lobes runs without a device: tests/test_materials.py compares every recipe with the
// oracle's line-by-line restatement of the reference); build_part is also compiled for the device.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rspt.h"

#if defined(__HIPCC__)
#include "dev_math.h"  // rspt_logf: the device's restatement of the host libm's logf
#define RSPT_HD __host__ __device__ inline
#else
#define RSPT_HD inline
#endif

namespace rspt_mat {

// parameters of the non-mix materials (the mix's amount and every material's bump map are handled beside them)
enum ParamId : uint32_t { P_KD, P_KS, P_KR, P_KT, P_REFLECT, P_TRANSMIT, P_OPACITY, P_ETA, P_K, P_SIGMA, P_ROUGH, P_UROUGH, P_VROUGH, P_INDEX, P_COUNT };
// how a parameter's texture value enters the recipe: colours are clamped to [0, inf) right after evaluate (every recipe), the conductor's
// eta / k are not (metal.rs:183-187), float parameters read the first channel
enum ParamMode : uint32_t { PM_COLOUR, PM_RAW3, PM_FLOAT };
RSPT_HD ParamMode param_mode(uint32_t id) { return id <= P_OPACITY ? PM_COLOUR : (id <= P_K ? PM_RAW3 : PM_FLOAT); }

// a material parameter: its value, or (host only, the fast path) a texture that completes the lobe per hit
struct Param {
    uint32_t present;  // only the *_or_null parameters may be absent
    uint32_t tex;      // 1 + texture index when the parameter varies over the surface and stays symbolic, 0 when `v` is the whole story
    float v[3];
};

RSPT_HD float clamp0(float x) { return x < 0.0f ? 0.0f : x; }  // clamp_t(x, 0, inf) (pbrt.rs:108-121): a NaN stays a NaN
RSPT_HD bool black(const float c[3]) { return c[0] == 0.0f && c[1] == 0.0f && c[2] == 0.0f; }
RSPT_HD Param param_value(uint32_t id, float x, float y, float z) {  // a parameter from its texture's value at a hit (or a ConstantTexture's value)
    Param p;
    p.present = 1u; p.tex = 0u;
    const ParamMode m = param_mode(id);
    p.v[0] = m == PM_COLOUR ? clamp0(x) : x;
    p.v[1] = m == PM_COLOUR ? clamp0(y) : (m == PM_FLOAT ? x : y);
    p.v[2] = m == PM_COLOUR ? clamp0(z) : (m == PM_FLOAT ? x : z);
    return p;
}

// TrowbridgeReitzDistribution::roughness_to_alpha (microfacet.rs:243-254).  The reference's f32::ln is the host libm's logf; the device
// evaluates the same function through its restatement of glibc (glibc_libm.h rspt_logf)
RSPT_HD float roughness_to_alpha(float roughness) {
    const float r = roughness < 1e-3f ? 1e-3f : roughness;
#ifdef __HIP_DEVICE_COMPILE__
    const float x = rspt::rspt_logf(r);
#else
    const float x = std::log(r);
#endif
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}

// what build_part produces: the lobes one non-mix material pushes, Bsdf.eta, and which shape-deciding parameters were symbolic
struct Built {
    rspt_bxdf l[8];
    uint32_t n;
    uint32_t overflow;      // more than 8 pushes (Bsdf::add asserts, reflection.rs:247)
    uint32_t shape_varies;  // bit i: parameter i decides the shape of the list but varies over the surface (host: the material is dynamic)
    float eta;
};

namespace detail {
RSPT_HD rspt_bxdf lobe(uint32_t type) {
    rspt_bxdf b;
    b.type = type; b.fresnel = 0u;
    for (int i = 0; i < 3; i++) { b.r[i] = 0.0f; b.t[i] = 0.0f; b.c1[i] = 0.0f; b.c2[i] = 0.0f; b.sc[i] = 0.0f; }
    b.eta_a = b.eta_b = b.alpha_x = b.alpha_y = b.on_a = b.on_b = 0.0f;
    b.has_sc = b.tex_r = b.tex_t = b.tex_ax = b.tex_ay = b.remap = 0u;
    return b;
}
RSPT_HD void set3(float dst[3], const float src[3]) { dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; }
RSPT_HD void mul3(float dst[3], const float a[3], const float b[3]) { dst[0] = a[0] * b[0]; dst[1] = a[1] * b[1]; dst[2] = a[2] * b[2]; }
// the microfacet alphas of a (u, v) roughness pair: TrowbridgeReitzDistribution::new(remap ? roughness_to_alpha(r) : r, ..)
RSPT_HD void alphas(rspt_bxdf* b, const Param& ru, const Param& rv, bool remap) {
    for (int k = 0; k < 2; k++) {
        const Param& r = k ? rv : ru;
        float* alpha = k ? &b->alpha_y : &b->alpha_x;
        uint32_t* tex = k ? &b->tex_ay : &b->tex_ax;
        if (r.tex) { *alpha = 0.001f; *tex = r.tex; continue; }
        const float a = remap ? roughness_to_alpha(r.v[0]) : r.v[0];
        *alpha = a > 0.001f ? a : 0.001f;  // f32::max(0.001): a NaN alpha becomes 0.001
    }
    if (remap) b->remap |= RSPT_LOBE_REMAP;
}
RSPT_HD void push(Built* out, rspt_bxdf b, const float* sc, bool second) {
    if (sc) { set3(b.sc, sc); b.has_sc = 1u; }
    if (second) b.remap |= RSPT_LOBE_NODIFF;
    if (out->n < 8u) out->l[out->n++] = b;
    else out->overflow = 1u;
}
// a colour whose factor is known and whose texture (if any) completes it per hit: it can be non-black only if the factor is
RSPT_HD void colour(const Param& p, const float* scale, float c[3], uint32_t* tex) {
    *tex = p.tex;
    for (int i = 0; i < 3; i++) c[i] = p.tex ? (scale ? scale[i] : 1.0f) : (scale ? scale[i] * p.v[i] : p.v[i]);
}
}  // namespace detail

// One non-mix material: the lobes compute_scattering_functions pushes, in its order, behind its guards.  p[P_COUNT]: the parameters
// (absent ones: present = 0); sc: the MixMaterial scale handed down (scale_opt) or nullptr; second: the m2 side of a mix.  Parameters
// that may stay symbolic (tex != 0) are Kd / Ks / the roughnesses where the list keeps its shape; any other symbolic parameter is
// reported in out->shape_varies and the lobes built from its placeholder value are meaningless.
RSPT_HD void build_part(uint32_t kind, const Param* p, bool remap, bool multi, const float* sc, bool second, Built* out) {
    using namespace detail;
    float eta = 1.0f;
    auto shape = [&](uint32_t id) -> const Param& { if (p[id].tex) out->shape_varies |= 1u << id; return p[id]; };
    switch (kind) {
    case RSPT_MAT_MATTE: {  // Lambert, or OrenNayar when sigma != 0; nothing when Kd is black
        const Param& sg = shape(P_SIGMA);
        float c[3]; uint32_t tex;
        colour(p[P_KD], nullptr, c, &tex);
        if (black(c)) break;
        const float sig = sg.v[0] < 0.0f ? 0.0f : (sg.v[0] > 90.0f ? 90.0f : sg.v[0]);
        rspt_bxdf b = lobe(sig == 0.0f ? RSPT_BXDF_LAMBERT_R : RSPT_BXDF_OREN_NAYAR);
        set3(b.r, c); b.tex_r = tex;
        if (sig != 0.0f) {  // OrenNayar::new (reflection.rs:1057-1065)
            const float s = (3.14159265358979323846f / 180.0f) * sig, s2 = s * s;
            b.on_a = 1.0f - (s2 / (2.0f * (s2 + 0.33f)));
            b.on_b = 0.45f * s2 / (s2 + 0.09f);
        }
        push(out, b, sc, second);
        break;
    }
    case RSPT_MAT_PLASTIC: {  // diffuse + dielectric (1.5 -> 1.0) microfacet gloss
        float d[3], s[3]; uint32_t td, ts;
        colour(p[P_KD], nullptr, d, &td); colour(p[P_KS], nullptr, s, &ts);
        if (!black(d)) { rspt_bxdf b = lobe(RSPT_BXDF_LAMBERT_R); set3(b.r, d); b.tex_r = td; push(out, b, sc, second); }
        if (!black(s)) {
            rspt_bxdf b = lobe(RSPT_BXDF_MICROFACET_R);
            b.fresnel = RSPT_FRESNEL_DIELECTRIC; b.eta_a = 1.5f; b.eta_b = 1.0f;
            set3(b.r, s); b.tex_r = ts;
            alphas(&b, p[P_ROUGH], p[P_ROUGH], remap);
            push(out, b, sc, second);
        }
        break;
    }
    case RSPT_MAT_MIRROR: {  // pushed whatever Kr is
        const Param& kr = shape(P_KR);
        rspt_bxdf b = lobe(RSPT_BXDF_SPECULAR_R);
        b.fresnel = RSPT_FRESNEL_NOOP;
        set3(b.r, kr.v);
        push(out, b, sc, second);
        break;
    }
    case RSPT_MAT_GLASS: {
        const Param &kr = shape(P_KR), &kt = shape(P_KT), &ru = shape(P_UROUGH), &rv = shape(P_VROUGH), &ix = shape(P_INDEX);
        eta = ix.v[0];
        const bool specular = ru.v[0] == 0.0f && rv.v[0] == 0.0f;  // tested on the raw roughness values
        if (specular && multi) {  // one lobe that chooses between reflection and refraction itself; no black guard
            rspt_bxdf b = lobe(RSPT_BXDF_FRESNEL_SPEC);
            set3(b.r, kr.v); set3(b.t, kt.v); b.eta_a = 1.0f; b.eta_b = eta;
            push(out, b, sc, second);
            break;
        }
        if (!black(kr.v)) {
            rspt_bxdf b = lobe(specular ? RSPT_BXDF_SPECULAR_R : RSPT_BXDF_MICROFACET_R);
            b.fresnel = RSPT_FRESNEL_DIELECTRIC; b.eta_a = 1.0f; b.eta_b = eta;
            set3(b.r, kr.v);
            if (!specular) alphas(&b, ru, rv, remap);
            push(out, b, sc, second);
        }
        if (!black(kt.v)) {
            rspt_bxdf b = lobe(specular ? RSPT_BXDF_SPECULAR_T : RSPT_BXDF_MICROFACET_T);
            b.eta_a = 1.0f; b.eta_b = eta;
            set3(b.r, kt.v);
            if (!specular) alphas(&b, ru, rv, remap);
            push(out, b, sc, second);
        }
        break;
    }
    case RSPT_MAT_METAL: {  // one conductor microfacet lobe, R = 1; uroughness / vroughness fall back to roughness
        const Param &et = shape(P_ETA), &kk = shape(P_K);
        rspt_bxdf b = lobe(RSPT_BXDF_MICROFACET_R);
        b.fresnel = RSPT_FRESNEL_CONDUCTOR;
        b.r[0] = b.r[1] = b.r[2] = 1.0f;
        set3(b.c1, et.v); set3(b.c2, kk.v);
        alphas(&b, p[P_UROUGH].present ? p[P_UROUGH] : p[P_ROUGH], p[P_VROUGH].present ? p[P_VROUGH] : p[P_ROUGH], remap);
        push(out, b, sc, second);
        break;
    }
    case RSPT_MAT_SUBSTRATE: {  // FresnelBlend(Kd, Ks) unless both are black
        float d[3], s[3]; uint32_t td, ts;
        colour(p[P_KD], nullptr, d, &td); colour(p[P_KS], nullptr, s, &ts);
        if (black(d) && black(s)) break;
        rspt_bxdf b = lobe(RSPT_BXDF_FRESNEL_BLEND);
        set3(b.r, d); b.tex_r = td; set3(b.t, s); b.tex_t = ts;
        alphas(&b, p[P_UROUGH], p[P_VROUGH], remap);
        push(out, b, sc, second);
        break;
    }
    case RSPT_MAT_UBER: {  // (1 - opacity) pass-through + opacity * (diffuse, gloss, mirror, refraction)
        const Param &op = shape(P_OPACITY), &ix = shape(P_INDEX), &kr = shape(P_KR), &kt = shape(P_KT);
        const float e = ix.v[0];
        float through[3];
        for (int c = 0; c < 3; c++) through[c] = clamp0(1.0f - op.v[c]);
        if (!black(through)) {  // Bsdf eta stays 1 while anything passes straight through
            rspt_bxdf b = lobe(RSPT_BXDF_SPECULAR_T);
            set3(b.r, through); b.eta_a = 1.0f; b.eta_b = 1.0f;
            push(out, b, sc, second);
        } else eta = e;
        float d[3], s[3]; uint32_t td, ts;
        colour(p[P_KD], op.v, d, &td); colour(p[P_KS], op.v, s, &ts);
        if (!black(d)) { rspt_bxdf b = lobe(RSPT_BXDF_LAMBERT_R); set3(b.r, d); b.tex_r = td; push(out, b, sc, second); }
        if (!black(s)) {
            rspt_bxdf b = lobe(RSPT_BXDF_MICROFACET_R);
            b.fresnel = RSPT_FRESNEL_DIELECTRIC; b.eta_a = 1.0f; b.eta_b = e;
            set3(b.r, s); b.tex_r = ts;
            alphas(&b, p[P_UROUGH].present ? p[P_UROUGH] : p[P_ROUGH], p[P_VROUGH].present ? p[P_VROUGH] : p[P_ROUGH], remap);
            push(out, b, sc, second);
        }
        float r[3], t[3];
        mul3(r, op.v, kr.v); mul3(t, op.v, kt.v);
        if (!black(r)) { rspt_bxdf b = lobe(RSPT_BXDF_SPECULAR_R); b.fresnel = RSPT_FRESNEL_DIELECTRIC; b.eta_a = 1.0f; b.eta_b = e; set3(b.r, r); push(out, b, sc, second); }
        if (!black(t)) { rspt_bxdf b = lobe(RSPT_BXDF_SPECULAR_T); b.eta_a = 1.0f; b.eta_b = e; set3(b.r, t); push(out, b, sc, second); }
        break;
    }
    case RSPT_MAT_TRANSLUCENT: {  // (reflect, transmit) x (diffuse, gloss); Bsdf eta 1.5 always
        const Param &rf = shape(P_REFLECT), &tm = shape(P_TRANSMIT), &kd = shape(P_KD), &ks = shape(P_KS);
        eta = 1.5f;
        const bool has_r = !black(rf.v), has_t = !black(tm.v);
        if (!has_r && !has_t) break;
        float c[3];
        if (!black(kd.v)) {
            if (has_r) { rspt_bxdf b = lobe(RSPT_BXDF_LAMBERT_R); mul3(c, rf.v, kd.v); set3(b.r, c); push(out, b, sc, second); }
            if (has_t) { rspt_bxdf b = lobe(RSPT_BXDF_LAMBERT_T); mul3(c, tm.v, kd.v); set3(b.r, c); push(out, b, sc, second); }
        }
        if (!black(ks.v)) {
            if (has_r) {
                rspt_bxdf b = lobe(RSPT_BXDF_MICROFACET_R);
                b.fresnel = RSPT_FRESNEL_DIELECTRIC; b.eta_a = 1.0f; b.eta_b = eta;
                mul3(c, rf.v, ks.v); set3(b.r, c);
                alphas(&b, p[P_ROUGH], p[P_ROUGH], remap);
                push(out, b, sc, second);
            }
            if (has_t) {
                rspt_bxdf b = lobe(RSPT_BXDF_MICROFACET_T);
                b.eta_a = 1.0f; b.eta_b = eta;
                mul3(c, tm.v, ks.v); set3(b.r, c);
                alphas(&b, p[P_ROUGH], p[P_ROUGH], remap);
                push(out, b, sc, second);
            }
        }
        break;
    }
    default: break;
    }
    if (!second) out->eta = eta;  // the Bsdf that survives a mix is the one m1 built (mixmat.rs:70-75)
}

// ---- dynamic materials: what the device needs to run build_part per hit ---------------------------------------------------------
#define RSPT_DYN_ROWS 12  // varying parameter textures one (top-level) material may bind
struct DynParam {         // a parameter of a dynamic material: a constant, or the row of the path's texture rows that holds its raw value at the hit
    uint32_t present;
    uint32_t row;         // 0: constant v; else 1 + row
    float v[3];
};
#define RSPT_MIX_PARTS 8  // non-mix materials one (possibly nested) mix may bring together: Bsdf.bxdfs holds 8 lobes (reflection.rs:243-248)
struct DynPart {          // one non-mix material of the (flattened) tree, in the order its lobes arrive on the surviving Bsdf
    uint32_t kind, remap;
    uint32_t side;        // 0: no scale (the material is not under a mix); 1: its parent mix's s1 = clamp(amount); 2: s2 = clamp(1 - s1)
    uint32_t second;      // reached through some m2 edge: built on a SurfaceInteraction::new without differentials (mixmat.rs:58-69)
    DynParam amount;      // the PARENT mix's MixMaterial.scale (the only scale a material sees: a mix ignores the one it is handed, mixmat.rs:50)
    DynParam p[P_COUNT];
};
struct DynMaterial {      // one per material (valid where the material is flagged RSPT_MAT_DYNAMIC)
    uint32_t n_parts;
    DynPart part[RSPT_MIX_PARTS];
    uint32_t n_rows;
    uint32_t row_tex[RSPT_DYN_ROWS];  // texture index | RSPT_SLOT_NODIFF of every row
};

// ---- host: binding the ABI's texture references, static assembly, the records of dynamic materials ----
struct Error {
    int code = RSPT_OK;
    std::string text;
    explicit operator bool() const { return code != RSPT_OK; }
};
struct Lobes {
    rspt_material mat{};
    std::vector<rspt_bxdf> lobes;
    bool dynamic = false;   // lobes / mat.eta are then a placeholder (empty); dyn describes the material
    DynMaterial dyn{};
};

class Assembler {
public:
    Assembler(const rspt_scene_desc* d, bool allow_multiple_lobes) : d_(d), multi_(allow_multiple_lobes) {}

    // the lobe list of material `index`; lobes carry texture indices (1 + index), not slots
    Error assemble(uint32_t index, Lobes* out) {
        err_ = Error{};
        out->lobes.clear();
        out->mat = rspt_material{1.0f, 0u, 0u, 0u};
        out->dynamic = false;
        memset(&out->dyn, 0, sizeof out->dyn);
        if (!d_ || index >= d_->n_materials || !d_->materials) return fail(RSPT_E_INVALID, index, "material index out of range");
        // MixMaterial (mixmat.rs:43-305): m1 builds the Bsdf of `si` under Some(s1), m2 builds one on a SurfaceInteraction::new (no ray
        // differentials) under Some(s2), and m2's BxDFs are re-created on m1's Bsdf in order.  A mix that is itself handed a scale ignores it
        // (`_scale`, :50) and hands its OWN s1 / s2 down — so a tree of mixes flattens into its non-mix leaves, left to right, each scaled by
        // its immediate parent's amount only; the surviving Bsdf (eta, bump-mapped shading frame) is the leftmost leaf's.
        std::vector<Leaf> leaves;
        flatten(index, 0, Param{0u, 0u, {0, 0, 0}}, false, false, 0, &leaves);
        if (err_) return err_;
        const int n_parts = (int)leaves.size();
        Param p[RSPT_MIX_PARTS][P_COUNT];
        for (int k = 0; k < n_parts; k++) {
            const rspt_material_desc& q = *leaves[k].m;
            if (q.kind < RSPT_MAT_MATTE || q.kind > RSPT_MAT_TRANSLUCENT)
                return fail(RSPT_E_UNSUPPORTED, leaves[k].index, "material kind " + std::to_string(q.kind) + " (matte, plastic, mirror, glass, metal, substrate, uber, translucent, mix)");
            const uint32_t refs[P_COUNT] = {q.kd, q.ks, q.kr, q.kt, q.reflect, q.transmit, q.opacity, q.eta, q.k, q.sigma, q.roughness, q.uroughness, q.vroughness, q.index};
            static const char* const names[P_COUNT] = {"Kd", "Ks", "Kr", "Kt", "reflect", "transmit", "opacity", "eta", "k", "sigma", "roughness", "uroughness", "vroughness", "index"};
            const uint32_t used = params_of(q.kind), optional = (1u << P_UROUGH) | (1u << P_VROUGH);
            for (uint32_t i = 0; i < P_COUNT; i++) {
                p[k][i] = Param{0u, 0u, {0, 0, 0}};
                if (!((used >> i) & 1u)) continue;
                const bool opt = ((optional >> i) & 1u) && (q.kind == RSPT_MAT_METAL || q.kind == RSPT_MAT_UBER);
                p[k][i] = bind(leaves[k].index, refs[i], names[i], param_mode(i), opt);
            }
            if (q.bumpmap > d_->n_textures) return fail(RSPT_E_INVALID, leaves[k].index, "parameter \"bumpmap\": texture index out of range");
            if (err_) return err_;
        }
        // static attempt: everything that shapes the list constant
        Built bl;
        memset(&bl, 0, sizeof bl);
        bl.eta = 1.0f;
        bool amount_varies = false;
        for (int k = 0; k < n_parts; k++) {
            float sc[3] = {0, 0, 0};
            for (int c = 0; c < 3; c++) sc[c] = leaves[k].side == 2 ? clamp0(1.0f - leaves[k].amount.v[c]) : leaves[k].amount.v[c];
            if (leaves[k].side && leaves[k].amount.tex) amount_varies = true;
            build_part(leaves[k].m->kind, p[k], leaves[k].m->remap_roughness != 0, multi_, leaves[k].side ? sc : nullptr, leaves[k].second, &bl);
        }
        out->mat.bump_tex = leaves[0].m->bumpmap;  // the leftmost leaf's (also a ConstantTexture displaces: displace * shading.dndu, material.rs:150-170); every m2 bumps a copy of the interaction that is dropped
        if (!bl.shape_varies && !amount_varies) {
            if (bl.overflow) return fail(RSPT_E_UNSUPPORTED, index, "more than 8 BxDFs (Bsdf::add asserts, reflection.rs:247)");
            out->mat.eta = bl.eta;
            out->lobes.assign(bl.l, bl.l + bl.n);
            out->mat.n_bxdfs = bl.n;
            return err_;
        }
        // dynamic: every varying parameter gets a row of the path's texture rows; build_part runs per hit on the device.  Its Built record
        // holds 8 lobes like Bsdf.bxdfs (reflection.rs:243-248: Bsdf::add asserts on the ninth), and nothing on the device could report a
        // ninth push — so the list is bounded here from what can be non-black at SOME hit, and a material that could pass 8 is refused
        // like its static counterpart above
        uint32_t worst = 0;
        for (int k = 0; k < n_parts; k++) worst += max_lobes(leaves[k].m->kind, p[k], multi_);
        if (worst > 8u)
            return fail(RSPT_E_UNSUPPORTED, index, "up to " + std::to_string(worst) + " BxDFs where the varying parameters are all non-black (Bsdf::add asserts past 8, reflection.rs:247)");
        out->dynamic = true;
        DynMaterial& dm = out->dyn;
        dm.n_parts = (uint32_t)n_parts;
        auto row_of = [&](uint32_t tex_plus_1, bool second) -> uint32_t {
            const uint32_t desc = (tex_plus_1 - 1u) | (second ? 0x20000000u /* RSPT_SLOT_NODIFF */ : 0u);
            for (uint32_t r = 0; r < dm.n_rows; r++) if (dm.row_tex[r] == desc) return r + 1u;
            if (dm.n_rows == RSPT_DYN_ROWS) { fail(RSPT_E_UNSUPPORTED, index, "more than " + std::to_string(RSPT_DYN_ROWS) + " varying parameter textures on a material whose lobe list they shape"); return 1u; }
            dm.row_tex[dm.n_rows++] = desc;
            return dm.n_rows;
        };
        auto dyn_param = [&](const Param& q, bool second) {
            DynParam r;
            r.present = q.present; r.row = q.tex ? row_of(q.tex, second) : 0u;
            r.v[0] = q.v[0]; r.v[1] = q.v[1]; r.v[2] = q.v[2];
            return r;
        };
        for (int k = 0; k < n_parts; k++) {
            dm.part[k].kind = leaves[k].m->kind; dm.part[k].remap = leaves[k].m->remap_roughness != 0;
            dm.part[k].side = (uint32_t)leaves[k].side; dm.part[k].second = leaves[k].second ? 1u : 0u;
            dm.part[k].amount = dyn_param(leaves[k].amount, leaves[k].amount_second);   // evaluated on the interaction the parent mix was handed
            for (uint32_t i = 0; i < P_COUNT; i++) dm.part[k].p[i] = dyn_param(p[k][i], leaves[k].second);
        }
        return err_;
    }

private:
    const rspt_scene_desc* d_;
    bool multi_;
    Error err_;

    // a non-mix material of the flattened tree (see assemble)
    struct Leaf {
        const rspt_material_desc* m;
        uint32_t index;
        int side;             // 0: not under a mix; 1 / 2: the m1 / m2 side of its parent
        bool second;          // some m2 edge on the way down: no ray differentials
        Param amount;         // the parent's MixMaterial.scale ...
        bool amount_second;   // ... evaluated on the interaction the PARENT was handed
    };
    void flatten(uint32_t index, int side, const Param& amount, bool amount_second, bool second, int depth, std::vector<Leaf>* out) {
        if (err_) return;
        if (!d_ || index >= d_->n_materials || !d_->materials) { fail(RSPT_E_INVALID, index, "material index out of range"); return; }
        const rspt_material_desc& m = d_->materials[index];
        if (m.kind != RSPT_MAT_MIX) {
            if (out->size() == RSPT_MIX_PARTS) { fail(RSPT_E_UNSUPPORTED, index, "a tree of mixes with more than " + std::to_string(RSPT_MIX_PARTS) + " non-mix materials"); return; }
            out->push_back(Leaf{&m, index, side, second, amount, amount_second});
            return;
        }
        if (depth >= RSPT_MIX_PARTS) { fail(RSPT_E_INVALID, index, "mixes nested deeper than " + std::to_string(RSPT_MIX_PARTS) + " (a mix that contains itself?)"); return; }
        if (m.m1 >= d_->n_materials || m.m2 >= d_->n_materials) { fail(RSPT_E_INVALID, index, "mix: material index out of range"); return; }
        const Param a = bind(index, m.amount, "amount", PM_COLOUR, false);
        if (err_) return;
        flatten(m.m1, 1, a, second, second, depth + 1, out);
        flatten(m.m2, 2, a, second, true, depth + 1, out);
    }

    Error fail(int code, uint32_t index, const std::string& what) {
        if (!err_) { err_.code = code; err_.text = "material " + std::to_string(index) + ": " + what; }
        return err_;
    }
    // upper bound of the lobes build_part can push for one part at any hit: a factor counts as possibly non-black when it is a texture
    // or a non-black constant (the guards of build_part, evaluated on what is known at scene creation)
    static uint32_t max_lobes(uint32_t kind, const Param* p, bool multi) {
        auto may = [](const Param& q) { return q.tex != 0u || !black(q.v); };
        auto may_be_nonzero = [](const Param& q) { return q.tex != 0u || q.v[0] != 0.0f; };
        switch (kind) {
        case RSPT_MAT_MATTE: return may(p[P_KD]) ? 1u : 0u;
        case RSPT_MAT_PLASTIC: return (may(p[P_KD]) ? 1u : 0u) + (may(p[P_KS]) ? 1u : 0u);
        case RSPT_MAT_MIRROR: case RSPT_MAT_METAL: return 1u;
        case RSPT_MAT_GLASS:
            if (multi && !may_be_nonzero(p[P_UROUGH]) && !may_be_nonzero(p[P_VROUGH])) return 1u;
            return std::max(multi ? 1u : 0u, (may(p[P_KR]) ? 1u : 0u) + (may(p[P_KT]) ? 1u : 0u));
        case RSPT_MAT_SUBSTRATE: return (may(p[P_KD]) || may(p[P_KS])) ? 1u : 0u;
        case RSPT_MAT_UBER: {
            const Param& op = p[P_OPACITY];
            float through[3] = {clamp0(1.0f - op.v[0]), clamp0(1.0f - op.v[1]), clamp0(1.0f - op.v[2])};
            const uint32_t pass = (op.tex != 0u || !black(through)) ? 1u : 0u;
            if (!may(op)) return pass;
            return pass + (may(p[P_KD]) ? 1u : 0u) + (may(p[P_KS]) ? 1u : 0u) + (may(p[P_KR]) ? 1u : 0u) + (may(p[P_KT]) ? 1u : 0u);
        }
        case RSPT_MAT_TRANSLUCENT: {
            const uint32_t sides = (may(p[P_REFLECT]) ? 1u : 0u) + (may(p[P_TRANSMIT]) ? 1u : 0u);
            return sides * ((may(p[P_KD]) ? 1u : 0u) + (may(p[P_KS]) ? 1u : 0u));
        }
        }
        return 0u;
    }
    // which parameters a material kind reads
    static uint32_t params_of(uint32_t kind) {
        auto b = [](uint32_t i) { return 1u << i; };
        switch (kind) {
        case RSPT_MAT_MATTE: return b(P_KD) | b(P_SIGMA);
        case RSPT_MAT_PLASTIC: return b(P_KD) | b(P_KS) | b(P_ROUGH);
        case RSPT_MAT_MIRROR: return b(P_KR);
        case RSPT_MAT_GLASS: return b(P_KR) | b(P_KT) | b(P_UROUGH) | b(P_VROUGH) | b(P_INDEX);
        case RSPT_MAT_METAL: return b(P_ETA) | b(P_K) | b(P_ROUGH) | b(P_UROUGH) | b(P_VROUGH);
        case RSPT_MAT_SUBSTRATE: return b(P_KD) | b(P_KS) | b(P_UROUGH) | b(P_VROUGH);
        case RSPT_MAT_UBER: return b(P_KD) | b(P_KS) | b(P_KR) | b(P_KT) | b(P_OPACITY) | b(P_ROUGH) | b(P_UROUGH) | b(P_VROUGH) | b(P_INDEX);
        case RSPT_MAT_TRANSLUCENT: return b(P_KD) | b(P_KS) | b(P_REFLECT) | b(P_TRANSMIT) | b(P_ROUGH);
        }
        return 0u;
    }
    // parameter -> folded constant, or a reference to its (non-constant) texture
    Param bind(uint32_t index, uint32_t ref, const char* name, ParamMode mode, bool optional) {
        Param p = Param{0u, 0u, {0, 0, 0}};
        if (ref == 0) {
            if (!optional) fail(RSPT_E_INVALID, index, std::string("parameter \"") + name + "\" is missing (0 is valid for bumpmap and for uroughness / vroughness of metal and uber only)");
            return p;
        }
        if (ref > d_->n_textures || !d_->textures) { fail(RSPT_E_INVALID, index, std::string("parameter \"") + name + "\": texture index out of range"); return p; }
        const rspt_texture& t = d_->textures[ref - 1u];
        if (t.kind == RSPT_TEX_CONSTANT) {
            const uint32_t id = mode == PM_COLOUR ? P_KD : (mode == PM_RAW3 ? P_ETA : P_SIGMA);
            return param_value(id, t.value[0], t.value[1], t.value[2]);
        }
        p.present = 1u; p.tex = ref;
        p.v[0] = p.v[1] = p.v[2] = 1.0f;
        return p;
    }
};
}  // namespace rspt_mat
