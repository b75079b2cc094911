// Host side of Material::compute_scattering_functions (src/core/material.rs:63-113 and src/materials/*.rs): turns the material
// records of the ABI (rspt_material_desc: a kind and one texture reference per parameter) into the lobe lists the shade stage runs.
//
// The reference evaluates every parameter texture at every hit and then decides which BxDFs to push.  Here a parameter bound to a
// ConstantTexture is folded once per material (clamp, black guards, roughness remapping, OrenNayar's A / B, the uber opacity
// products), a parameter bound to any other texture leaves a reference on the lobe (tex_r / tex_t / tex_ax / tex_ay) that the
// texture stage resolves per hit (kernels.h texture_path).  Parameters that decide the SHAPE of the lobe list per hit — which
// lobes exist, their type, Bsdf.eta — must be constant; a scene that binds one of those to a varying texture is refused with
// RSPT_E_UNSUPPORTED (the caller keeps its CPU loop).
//
// Plain C++ (no HIP): rspt_material_lobes runs without a device so that tests/test_materials.py can compare every recipe with the
// oracle's line-by-line restatement of the reference on a CPU-only box.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rspt.h"

namespace rspt_mat {

struct Error {
    int code = RSPT_OK;
    std::string text;
    explicit operator bool() const { return code != RSPT_OK; }
};

// a material parameter after constant folding
struct Param {
    bool present = false;   // only the *_or_null parameters may be absent
    uint32_t tex = 0;       // 1 + texture index when the parameter varies over the surface, 0 when `v` is the whole story
    float v[3] = {0, 0, 0};
    bool varying() const { return tex != 0; }
};

struct Lobes {
    rspt_material mat{};
    std::vector<rspt_bxdf> lobes;
};

inline float clamp0(float x) { return x < 0.0f ? 0.0f : x; }  // clamp_t(x, 0, inf) (pbrt.rs:108-121): a NaN stays a NaN
inline bool black(const float c[3]) { return c[0] == 0.0f && c[1] == 0.0f && c[2] == 0.0f; }

// TrowbridgeReitzDistribution::roughness_to_alpha (microfacet.rs:243-254); the host's logf is the reference's f32::ln
inline float roughness_to_alpha(float roughness) {
    const float r = roughness < 1e-3f ? 1e-3f : roughness;
    const float x = std::log(r);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}

class Assembler {
public:
    Assembler(const rspt_scene_desc* d, bool allow_multiple_lobes) : d_(d), multi_(allow_multiple_lobes) {}

    // the lobe list of material `index`; lobes carry texture indices (1 + index), not slots
    Error assemble(uint32_t index, Lobes* out) {
        err_ = Error{};
        out->lobes.clear();
        out->mat = rspt_material{1.0f, 0u, 0u, 0u};
        if (!d_ || index >= d_->n_materials || !d_->materials) return fail(RSPT_E_INVALID, index, "material index out of range");
        const rspt_material_desc& m = d_->materials[index];
        if (m.kind == RSPT_MAT_MIX) mix(index, m, out);
        else single(index, m, nullptr, false, out);
        if (!err_ && out->lobes.size() > 8) fail(RSPT_E_UNSUPPORTED, index, "more than 8 BxDFs (Bsdf::add asserts, reflection.rs:247)");
        out->mat.n_bxdfs = (uint32_t)out->lobes.size();
        return err_;
    }

private:
    const rspt_scene_desc* d_;
    bool multi_;
    Error err_;

    Error fail(int code, uint32_t index, const std::string& what) {
        if (!err_) { err_.code = code; err_.text = "material " + std::to_string(index) + ": " + what; }
        return err_;
    }

    // parameter -> folded constant or texture reference.  `channels` 3: spectrum (clamped to [0, inf) as every recipe does right
    // after evaluate), 1: float (raw)
    Param bind(uint32_t index, uint32_t ref, const char* name, int channels, bool optional = false) {
        Param p;
        if (ref == 0) {
            if (!optional) fail(RSPT_E_INVALID, index, std::string("parameter \"") + name + "\" is missing (0 is valid for bumpmap / uroughness / vroughness of metal and uber only)");
            return p;
        }
        if (ref > d_->n_textures || !d_->textures) { fail(RSPT_E_INVALID, index, std::string("parameter \"") + name + "\": texture index out of range"); return p; }
        p.present = true;
        const rspt_texture& t = d_->textures[ref - 1u];
        if (t.kind == RSPT_TEX_CONSTANT) {
            for (int c = 0; c < 3; c++) p.v[c] = channels == 3 ? clamp0(t.value[c]) : t.value[0];
        } else {
            p.tex = ref;
            p.v[0] = p.v[1] = p.v[2] = 1.0f;  // the factor the texture value is multiplied with
        }
        return p;
    }
    // a parameter that shapes the lobe list: it has to be constant
    Param bind_const(uint32_t index, uint32_t ref, const char* name, int channels) {
        Param p = bind(index, ref, name, channels);
        if (p.varying()) fail(RSPT_E_UNSUPPORTED, index, std::string("parameter \"") + name + "\" is bound to a non-constant texture (it decides which lobes exist; constant only)");
        return p;
    }

    static rspt_bxdf lobe(uint32_t type) {
        rspt_bxdf b;
        memset(&b, 0, sizeof b);
        b.type = type;
        return b;
    }
    static void set3(float dst[3], const float src[3]) { dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; }
    static void mul3(float dst[3], const float a[3], const float b[3]) { dst[0] = a[0] * b[0]; dst[1] = a[1] * b[1]; dst[2] = a[2] * b[2]; }
    // the microfacet alphas of a (u, v) roughness pair: TrowbridgeReitzDistribution::new(remap ? roughness_to_alpha(r) : r, ..)
    static void alphas(rspt_bxdf* b, const Param& ru, const Param& rv, bool remap) {
        auto one = [&](const Param& r, float* alpha, uint32_t* tex) {
            if (r.varying()) { *alpha = 0.001f; *tex = r.tex; return; }
            const float a = remap ? roughness_to_alpha(r.v[0]) : r.v[0];
            *alpha = a > 0.001f ? a : 0.001f;  // f32::max(0.001): a NaN alpha becomes 0.001
        };
        one(ru, &b->alpha_x, &b->tex_ax);
        one(rv, &b->alpha_y, &b->tex_ay);
        if (remap) b->remap |= RSPT_LOBE_REMAP;
    }
    // colour of a lobe: factor * (texture or 1); exists = it can be non-black at some hit
    struct Colour {
        float c[3];
        uint32_t tex;
        bool exists() const { return !black(c); }  // a black factor stays black whatever the texture says
    };
    static Colour colour(const Param& p, const float* scale = nullptr) {
        Colour k;
        k.tex = p.tex;
        for (int i = 0; i < 3; i++) k.c[i] = scale ? scale[i] * p.v[i] : p.v[i];
        return k;
    }

    void push(Lobes* out, rspt_bxdf b, const float* sc, bool second) {
        if (sc) { set3(b.sc, sc); b.has_sc = 1u; }
        if (second) b.remap |= RSPT_LOBE_NODIFF;
        out->lobes.push_back(b);
    }

    // one non-mix material; sc: the MixMaterial scale handed down (scale_opt), second: the m2 side of a mix
    void single(uint32_t index, const rspt_material_desc& m, const float* sc, bool second, Lobes* out) {
        const bool remap = m.remap_roughness != 0;
        float eta = 1.0f;
        const Param bump = bind(index, m.bumpmap, "bumpmap", 1, true);
        if (err_) return;
        switch (m.kind) {
        case RSPT_MAT_MATTE: {  // Lambert, or OrenNayar when sigma != 0; nothing when Kd is black
            const Param kd = bind(index, m.kd, "Kd", 3);
            const Param sg = bind_const(index, m.sigma, "sigma", 1);
            if (err_) return;
            const Colour r = colour(kd);
            if (!r.exists()) break;
            const float sig = sg.v[0] < 0.0f ? 0.0f : (sg.v[0] > 90.0f ? 90.0f : sg.v[0]);
            rspt_bxdf b = lobe(sig == 0.0f ? RSPT_BXDF_LAMBERT_R : RSPT_BXDF_OREN_NAYAR);
            set3(b.r, r.c); b.tex_r = r.tex;
            if (sig != 0.0f) {  // OrenNayar::new (reflection.rs:1057-1065)
                const float s = (3.14159265358979323846f / 180.0f) * sig, s2 = s * s;
                b.on_a = 1.0f - (s2 / (2.0f * (s2 + 0.33f)));
                b.on_b = 0.45f * s2 / (s2 + 0.09f);
            }
            push(out, b, sc, second);
            break;
        }
        case RSPT_MAT_PLASTIC: {  // diffuse + dielectric (1.5 -> 1.0) microfacet gloss
            const Param kd = bind(index, m.kd, "Kd", 3), ks = bind(index, m.ks, "Ks", 3), ro = bind(index, m.roughness, "roughness", 1);
            if (err_) return;
            const Colour d = colour(kd), s = colour(ks);
            if (d.exists()) { rspt_bxdf b = lobe(RSPT_BXDF_LAMBERT_R); set3(b.r, d.c); b.tex_r = d.tex; push(out, b, sc, second); }
            if (s.exists()) {
                rspt_bxdf b = lobe(RSPT_BXDF_MICROFACET_R);
                b.fresnel = RSPT_FRESNEL_DIELECTRIC; b.eta_a = 1.5f; b.eta_b = 1.0f;
                set3(b.r, s.c); b.tex_r = s.tex;
                alphas(&b, ro, ro, remap);
                push(out, b, sc, second);
            }
            break;
        }
        case RSPT_MAT_MIRROR: {  // pushed whatever Kr is
            const Param kr = bind_const(index, m.kr, "Kr", 3);
            if (err_) return;
            rspt_bxdf b = lobe(RSPT_BXDF_SPECULAR_R);
            b.fresnel = RSPT_FRESNEL_NOOP;
            set3(b.r, kr.v);
            push(out, b, sc, second);
            break;
        }
        case RSPT_MAT_GLASS: {
            const Param kr = bind_const(index, m.kr, "Kr", 3), kt = bind_const(index, m.kt, "Kt", 3);
            const Param ru = bind_const(index, m.uroughness, "uroughness", 1), rv = bind_const(index, m.vroughness, "vroughness", 1);
            const Param ix = bind_const(index, m.index, "index", 1);
            if (err_) return;
            eta = ix.v[0];
            const bool specular = ru.v[0] == 0.0f && rv.v[0] == 0.0f;  // tested on the raw roughness values
            if (specular && multi_) {  // one lobe that chooses between reflection and refraction itself; no black guard
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
            const Param et = bind_const(index, m.eta, "eta", 3), kk = bind_const(index, m.k, "k", 3);
            const Param ro = bind(index, m.roughness, "roughness", 1);
            const Param ru = bind(index, m.uroughness, "uroughness", 1, true), rv = bind(index, m.vroughness, "vroughness", 1, true);
            if (err_) return;
            rspt_bxdf b = lobe(RSPT_BXDF_MICROFACET_R);
            b.fresnel = RSPT_FRESNEL_CONDUCTOR;
            b.r[0] = b.r[1] = b.r[2] = 1.0f;
            // the conductor's eta / k are evaluated without the clamp the colour parameters get (metal.rs:183-187)
            raw3(index, m.eta, b.c1); raw3(index, m.k, b.c2);
            (void)et; (void)kk;
            alphas(&b, ru.present ? ru : ro, rv.present ? rv : ro, remap);
            push(out, b, sc, second);
            break;
        }
        case RSPT_MAT_SUBSTRATE: {  // FresnelBlend(Kd, Ks) unless both are black
            const Param kd = bind(index, m.kd, "Kd", 3), ks = bind(index, m.ks, "Ks", 3);
            const Param ru = bind(index, m.uroughness, "uroughness", 1), rv = bind(index, m.vroughness, "vroughness", 1);
            if (err_) return;
            const Colour d = colour(kd), s = colour(ks);
            if (!d.exists() && !s.exists()) break;
            rspt_bxdf b = lobe(RSPT_BXDF_FRESNEL_BLEND);
            set3(b.r, d.c); b.tex_r = d.tex; set3(b.t, s.c); b.tex_t = s.tex;
            alphas(&b, ru, rv, remap);
            push(out, b, sc, second);
            break;
        }
        case RSPT_MAT_UBER: {  // (1 - opacity) pass-through + opacity * (diffuse, gloss, mirror, refraction)
            const Param op = bind_const(index, m.opacity, "opacity", 3), ix = bind_const(index, m.index, "index", 1);
            const Param kd = bind(index, m.kd, "Kd", 3), ks = bind(index, m.ks, "Ks", 3);
            const Param kr = bind_const(index, m.kr, "Kr", 3), kt = bind_const(index, m.kt, "Kt", 3);
            const Param ro = bind(index, m.roughness, "roughness", 1);
            const Param ru = bind(index, m.uroughness, "uroughness", 1, true), rv = bind(index, m.vroughness, "vroughness", 1, true);
            if (err_) return;
            const float e = ix.v[0];
            float through[3];
            for (int c = 0; c < 3; c++) through[c] = clamp0(1.0f - op.v[c]);
            if (!black(through)) {  // Bsdf eta stays 1 while anything passes straight through
                rspt_bxdf b = lobe(RSPT_BXDF_SPECULAR_T);
                set3(b.r, through); b.eta_a = 1.0f; b.eta_b = 1.0f;
                push(out, b, sc, second);
            } else eta = e;
            const Colour d = colour(kd, op.v), s = colour(ks, op.v);
            if (d.exists()) { rspt_bxdf b = lobe(RSPT_BXDF_LAMBERT_R); set3(b.r, d.c); b.tex_r = d.tex; push(out, b, sc, second); }
            if (s.exists()) {
                rspt_bxdf b = lobe(RSPT_BXDF_MICROFACET_R);
                b.fresnel = RSPT_FRESNEL_DIELECTRIC; b.eta_a = 1.0f; b.eta_b = e;
                set3(b.r, s.c); b.tex_r = s.tex;
                alphas(&b, ru.present ? ru : ro, rv.present ? rv : ro, remap);
                push(out, b, sc, second);
            }
            float r[3], t[3];
            mul3(r, op.v, kr.v); mul3(t, op.v, kt.v);
            if (!black(r)) { rspt_bxdf b = lobe(RSPT_BXDF_SPECULAR_R); b.fresnel = RSPT_FRESNEL_DIELECTRIC; b.eta_a = 1.0f; b.eta_b = e; set3(b.r, r); push(out, b, sc, second); }
            if (!black(t)) { rspt_bxdf b = lobe(RSPT_BXDF_SPECULAR_T); b.eta_a = 1.0f; b.eta_b = e; set3(b.r, t); push(out, b, sc, second); }
            break;
        }
        case RSPT_MAT_TRANSLUCENT: {  // (reflect, transmit) x (diffuse, gloss); Bsdf eta 1.5 always
            const Param rf = bind_const(index, m.reflect, "reflect", 3), tm = bind_const(index, m.transmit, "transmit", 3);
            const Param kd = bind_const(index, m.kd, "Kd", 3), ks = bind_const(index, m.ks, "Ks", 3);
            const Param ro = bind(index, m.roughness, "roughness", 1);
            if (err_) return;
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
                    alphas(&b, ro, ro, remap);
                    push(out, b, sc, second);
                }
                if (has_t) {
                    rspt_bxdf b = lobe(RSPT_BXDF_MICROFACET_T);
                    b.eta_a = 1.0f; b.eta_b = eta;
                    mul3(c, tm.v, ks.v); set3(b.r, c);
                    alphas(&b, ro, ro, remap);
                    push(out, b, sc, second);
                }
            }
            break;
        }
        default:
            fail(RSPT_E_UNSUPPORTED, index, "material kind " + std::to_string(m.kind) + " (matte, plastic, mirror, glass, metal, substrate, uber, translucent, mix)");
            return;
        }
        if (!second) {  // the Bsdf that survives a mix is the one m1 built: its eta, its bumped shading frame (mixmat.rs:70-75)
            out->mat.eta = eta;
            out->mat.bump_tex = bump.present ? m.bumpmap : 0u;  // also a ConstantTexture displaces (displace * shading.dndu, material.rs:150-170)
        }
    }

    // a spectrum parameter as the texture holds it (no clamp)
    void raw3(uint32_t index, uint32_t ref, float out[3]) {
        if (ref == 0 || ref > d_->n_textures) { fail(RSPT_E_INVALID, index, "texture index out of range"); return; }
        set3(out, d_->textures[ref - 1u].value);
    }

    void mix(uint32_t index, const rspt_material_desc& m, Lobes* out) {  // m1 scaled by `amount`, m2 by 1 - amount, lobes concatenated on m1's Bsdf
        const Param am = bind_const(index, m.amount, "amount", 3);
        if (err_) return;
        if (m.m1 >= d_->n_materials || m.m2 >= d_->n_materials) { fail(RSPT_E_INVALID, index, "mix: material index out of range"); return; }
        const rspt_material_desc &a = d_->materials[m.m1], &b = d_->materials[m.m2];
        if (a.kind == RSPT_MAT_MIX || b.kind == RSPT_MAT_MIX) { fail(RSPT_E_UNSUPPORTED, index, "mix of a mix (MixMaterial ignores the scale it is handed, mixmat.rs:50)"); return; }
        float s1[3], s2[3];
        for (int c = 0; c < 3; c++) { s1[c] = am.v[c]; s2[c] = clamp0(1.0f - s1[c]); }
        single(m.m1, a, s1, false, out);
        if (err_) return;
        single(m.m2, b, s2, true, out);
    }
};

}  // namespace rspt_mat
