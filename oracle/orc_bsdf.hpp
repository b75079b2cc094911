// TEST INFRASTRUCTURE — CPU oracle (see orc_math.hpp header).  BSDF / BxDFs / microfacets.
#pragma once
#include "orc_texture.hpp"

namespace orc {

enum { BSDF_REFLECTION = 1, BSDF_TRANSMISSION = 2, BSDF_DIFFUSE = 4, BSDF_GLOSSY = 8, BSDF_SPECULAR = 16, BSDF_ALL = 31 };

// src/core/reflection.rs:1801-1886
static inline Float cos_theta(V3 w) { return w.z; }
static inline Float cos_2_theta(V3 w) { return w.z * w.z; }
static inline Float abs_cos_theta(V3 w) { return std::fabs(w.z); }
static inline Float sin_2_theta(V3 w) { return std::fmax(0.0f, 1.0f - cos_2_theta(w)); }
static inline Float sin_theta(V3 w) { return std::sqrt(sin_2_theta(w)); }
static inline Float tan_theta(V3 w) { return sin_theta(w) / cos_theta(w); }
static inline Float tan_2_theta(V3 w) { return sin_2_theta(w) / cos_2_theta(w); }
static inline Float cos_phi(V3 w) { Float s = sin_theta(w); return s == 0.0f ? 1.0f : clamp_t(w.x / s, -1.0f, 1.0f); }
static inline Float sin_phi(V3 w) { Float s = sin_theta(w); return s == 0.0f ? 0.0f : clamp_t(w.y / s, -1.0f, 1.0f); }
static inline Float cos_2_phi(V3 w) { return cos_phi(w) * cos_phi(w); }
static inline Float sin_2_phi(V3 w) { return sin_phi(w) * sin_phi(w); }
static inline bool same_hemisphere(V3 w, V3 wp) { return w.z * wp.z > 0.0f; } // :1911
static inline V3 reflect(V3 wo, V3 n) { return -wo + n * 2.0f * dot(wo, n); } // :1889
// reflection.rs:1897-1909
static inline bool refract(V3 wi, V3 n, Float eta, V3* wt) {
    Float cos_theta_i = dot(n, wi);
    Float sin2_theta_i = std::fmax(0.0f, 1.0f - cos_theta_i * cos_theta_i);
    Float sin2_theta_t = eta * eta * sin2_theta_i;
    if (sin2_theta_t >= 1.0f) return false;
    Float cos_theta_t = std::sqrt(1.0f - sin2_theta_t);
    *wt = -wi * eta + n * (eta * cos_theta_i - cos_theta_t);
    return true;
}
// reflection.rs:1920-1949
static inline Float fr_dielectric(Float cos_theta_i, Float eta_i, Float eta_t) {
    cos_theta_i = clamp_t(cos_theta_i, -1.0f, 1.0f);
    bool entering = cos_theta_i > 0.0f;
    if (!entering) { std::swap(eta_i, eta_t); cos_theta_i = std::fabs(cos_theta_i); }
    Float sin_theta_i = std::sqrt(std::fmax(0.0f, 1.0f - cos_theta_i * cos_theta_i));
    Float sin_theta_t = eta_i / eta_t * sin_theta_i;
    if (sin_theta_t >= 1.0f) return 1.0f;
    Float cos_theta_t = std::sqrt(std::fmax(0.0f, 1.0f - sin_theta_t * sin_theta_t));
    Float r_parl = ((eta_t * cos_theta_i) - (eta_i * cos_theta_t)) / ((eta_t * cos_theta_i) + (eta_i * cos_theta_t));
    Float r_perp = ((eta_i * cos_theta_i) - (eta_t * cos_theta_t)) / ((eta_i * cos_theta_i) + (eta_t * cos_theta_t));
    return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
}
// reflection.rs:1953-1972
static inline Spec fr_conductor(Float cos_theta_i, Spec eta_i, Spec eta_t, Spec k) {
    cos_theta_i = clamp_t(cos_theta_i, -1.0f, 1.0f);
    Spec eta = eta_t / eta_i;
    Spec eta_k = k / eta_i;
    Float cos2 = cos_theta_i * cos_theta_i;
    Float sin2 = 1.0f - cos2;
    Spec eta_2 = eta * eta;
    Spec eta_k2 = eta_k * eta_k;
    Spec t0 = eta_2 - eta_k2 - Spec(sin2);
    Spec a2_plus_b2 = ssqrt(t0 * t0 + eta_2 * eta_k2 * Spec(4.0f));
    Spec t1 = a2_plus_b2 + Spec(cos2);
    Spec a = ssqrt((a2_plus_b2 + t0) * 0.5f);
    Spec t2 = a * 2.0f * cos_theta_i;
    Spec rs = (t1 - t2) / (t1 + t2);
    Spec t3 = a2_plus_b2 * cos2 + Spec(sin2 * sin2);
    Spec t4 = t2 * sin2;
    Spec rp = rs * (t3 - t4) / (t3 + t4);
    return (rp + rs) * Spec(0.5f);
}

// src/core/sampling.rs:360-382
static inline P2 concentric_sample_disk(P2 u) {
    P2 uo{u.x * 2.0f - 1.0f, u.y * 2.0f - 1.0f};
    if (uo.x == 0.0f && uo.y == 0.0f) return P2{0, 0};
    Float theta, r;
    if (std::fabs(uo.x) > std::fabs(uo.y)) { r = uo.x; theta = PI_OVER_4 * (uo.y / uo.x); }
    else { r = uo.y; theta = PI_OVER_2 - PI_OVER_4 * (uo.x / uo.y); }
    return P2{std::cos(theta) * r, std::sin(theta) * r};
}
// sampling.rs:214-221
static inline V3 cosine_sample_hemisphere(P2 u) {
    P2 d = concentric_sample_disk(u);
    Float z = std::sqrt(std::fmax(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return V3{d.x, d.y, z};
}
// sampling.rs:229-233
static inline Float power_heuristic(int nf, Float f_pdf, int ng, Float g_pdf) {
    Float f = (Float)nf * f_pdf, g = (Float)ng * g_pdf;
    return (f * f) / (f * f + g * g);
}

// ---- TrowbridgeReitzDistribution (sample_visible_area = true): src/core/microfacet.rs:225-353,475-569 ----
struct TR {
    Float ax, ay;
    Float d(V3 wh) const {
        Float t2 = tan_2_theta(wh);
        if (std::isinf(t2)) return 0.0f;
        Float cos4 = cos_2_theta(wh) * cos_2_theta(wh);
        Float e = (cos_2_phi(wh) / (ax * ax) + sin_2_phi(wh) / (ay * ay)) * t2;
        return 1.0f / (PI * ax * ay * cos4 * (1.0f + e) * (1.0f + e));
    }
    Float lambda(V3 w) const {
        Float att = std::fabs(tan_theta(w));
        if (std::isinf(att)) return 0.0f;
        Float alpha = std::sqrt(cos_2_phi(w) * ax * ax + sin_2_phi(w) * ay * ay);
        Float a2t2 = (alpha * att) * (alpha * att);
        return (-1.0f + std::sqrt(1.0f + a2t2)) / 2.0f;
    }
    Float g1(V3 w) const { return 1.0f / (1.0f + lambda(w)); }
    Float g(V3 wo, V3 wi) const { return 1.0f / (1.0f + lambda(wo) + lambda(wi)); }
    Float pdf(V3 wo, V3 wh) const { return d(wh) * g1(wo) * abs_dot(wo, wh) / abs_cos_theta(wo); }
    static void sample_11(Float cos_th, Float u1, Float u2, Float* slope_x, Float* slope_y) { // :475-531
        if (cos_th > 0.9999f) {
            Float r = std::sqrt(u1 / (1.0f - u1));
            Float phi = 6.28318530717958647692f * u2; // std::f32::consts::TAU
            *slope_x = r * std::cos(phi);
            *slope_y = r * std::sin(phi);
            return;
        }
        Float sin_th = std::sqrt(std::fmax(0.0f, 1.0f - cos_th * cos_th));
        Float tan_th = sin_th / cos_th;
        Float a = 1.0f / tan_th;
        Float g1 = 2.0f / (1.0f + std::sqrt(1.0f + 1.0f / (a * a)));
        a = 2.0f * u1 / g1 - 1.0f;
        Float tmp = 1.0f / (a * a - 1.0f);
        if (tmp > 1e10f) tmp = 1e10f;
        Float b = tan_th;
        Float dd = std::sqrt(std::fmax(b * b * tmp * tmp - (a * a - b * b) * tmp, 0.0f));
        Float sx1 = b * tmp - dd, sx2 = b * tmp + dd;
        if (a < 0.0f || sx2 > 1.0f / tan_th) *slope_x = sx1; else *slope_x = sx2;
        Float s, nu2;
        if (u2 > 0.5f) { s = 1.0f; nu2 = 2.0f * (u2 - 0.5f); }
        else { s = -1.0f; nu2 = 2.0f * (0.5f - u2); }
        Float z = (nu2 * (nu2 * (nu2 * 0.27385f - 0.73369f) + 0.46341f)) /
                  (nu2 * (nu2 * (nu2 * 0.093073f + 0.309420f) - 1.0f) + 0.597999f);
        *slope_y = s * z * std::sqrt(1.0f + *slope_x * *slope_x);
    }
    static V3 sample(V3 wi, Float ax, Float ay, Float u1, Float u2) { // :533-569
        V3 ws = normalize(V3{ax * wi.x, ay * wi.y, wi.z});
        Float sx = 0, sy = 0;
        sample_11(cos_theta(ws), u1, u2, &sx, &sy);
        Float tmp = cos_phi(ws) * sx - sin_phi(ws) * sy;
        sy = sin_phi(ws) * sx + cos_phi(ws) * sy;
        sx = tmp;
        sx *= ax; sy *= ay;
        return normalize(V3{-sx, -sy, 1.0f});
    }
    V3 sample_wh(V3 wo, P2 u) const { // :295-349, visible-area branch
        if (wo.z < 0.0f) return -sample(-wo, ax, ay, u.x, u.y);
        return sample(wo, ax, ay, u.x, u.y);
    }
};

static inline Float pow5(Float v) { return (v * v) * (v * v) * v; } // reflection.rs:1974-1976

// One lobe = one Bxdf enum value (reflection.rs:462-633)
struct Lobe {
    const rspt_bxdf* b;
    uint8_t get_type() const {
        switch (b->type) {
        case RSPT_BXDF_LAMBERT_R: case RSPT_BXDF_OREN_NAYAR: return BSDF_DIFFUSE | BSDF_REFLECTION;
        case RSPT_BXDF_LAMBERT_T: return BSDF_DIFFUSE | BSDF_TRANSMISSION;
        case RSPT_BXDF_SPECULAR_R: return BSDF_REFLECTION | BSDF_SPECULAR;
        case RSPT_BXDF_SPECULAR_T: return BSDF_TRANSMISSION | BSDF_SPECULAR;
        case RSPT_BXDF_FRESNEL_SPEC: return BSDF_REFLECTION | BSDF_TRANSMISSION | BSDF_SPECULAR;
        case RSPT_BXDF_MICROFACET_R: return BSDF_REFLECTION | BSDF_GLOSSY;
        case RSPT_BXDF_MICROFACET_T: return BSDF_TRANSMISSION | BSDF_GLOSSY; // :1319
        case RSPT_BXDF_FRESNEL_BLEND: return BSDF_REFLECTION | BSDF_GLOSSY;  // :1475
        }
        return 0;
    }
    bool matches_flags(uint8_t t) const { return (get_type() & t) == get_type(); } // :487-508
    Spec fresnel(Float cos_i) const { // Fresnel::evaluate :651-705
        switch (b->fresnel) {
        case RSPT_FRESNEL_DIELECTRIC: return Spec(fr_dielectric(cos_i, b->eta_a, b->eta_b));
        case RSPT_FRESNEL_CONDUCTOR: return fr_conductor(cos_i, Spec(1.0f), S3(b->c1), S3(b->c2));
        default: return Spec(1.0f);
        }
    }
    // sc_opt (MixMaterial): `sc * A * B ...` evaluates as ((sc * A) * B) ...; scaled(A) is that first product
    Spec scaled(Spec a) const { return b->has_sc ? S3(b->sc) * a : a; }
    Spec f(V3 wo, V3 wi) const {
        switch (b->type) {
        case RSPT_BXDF_LAMBERT_R: return scaled(S3(b->r)) * Spec(INV_PI); // :960-966
        case RSPT_BXDF_LAMBERT_T: return scaled(S3(b->r)) * INV_PI;       // :1011-1017
        case RSPT_BXDF_OREN_NAYAR: {                               // :1067-1096
            Float sti = sin_theta(wi), sto = sin_theta(wo);
            Float max_cos = 0.0f;
            if (sti > 1.0e-4f && sto > 1.0e-4f) {
                Float spi = sin_phi(wi), cpi = cos_phi(wi), spo = sin_phi(wo), cpo = cos_phi(wo);
                Float d_cos = cpi * cpo + spi * spo;
                max_cos = std::fmax(d_cos, 0.0f);
            }
            Float sin_alpha, tan_beta;
            if (abs_cos_theta(wi) > abs_cos_theta(wo)) { sin_alpha = sto; tan_beta = sti / abs_cos_theta(wi); }
            else { sin_alpha = sti; tan_beta = sto / abs_cos_theta(wo); }
            return scaled(S3(b->r)) * Spec(INV_PI * (b->on_a + b->on_b * max_cos * sin_alpha * tan_beta));
        }
        case RSPT_BXDF_MICROFACET_R: { // :1147-1170
            Float cto = abs_cos_theta(wo), cti = abs_cos_theta(wi);
            V3 wh = wi + wo;
            if (cti == 0.0f || cto == 0.0f) return Spec(0.0f);
            if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return Spec(0.0f);
            wh = normalize(wh);
            Float dt = dot(wi, wh);
            Spec fr = fresnel(dt);
            TR tr{b->alpha_x, b->alpha_y};
            return scaled(S3(b->r)) * tr.d(wh) * tr.g(wo, wi) * fr / (4.0f * cti * cto);
        }
        case RSPT_BXDF_MICROFACET_T: { // :1246-1317 (TransportMode::Radiance)
            if (same_hemisphere(wo, wi)) return Spec(0.0f);
            Float cto = cos_theta(wo), cti = cos_theta(wi);
            if (cto == 0.0f || cti == 0.0f) return Spec(0.0f);
            Float eta = cto > 0.0f ? b->eta_b / b->eta_a : b->eta_a / b->eta_b;
            V3 wh = normalize(wo + wi * eta);
            if (wh.z < 0.0f) wh = -wh;
            if (dot(wo, wh) * dot(wi, wh) > 0.0f) return Spec(0.0f);
            Spec f(fr_dielectric(dot(wo, wh), b->eta_a, b->eta_b));
            Float sqrt_denom = dot(wo, wh) + eta * dot(wi, wh);
            Float factor = 1.0f / eta;
            TR tr{b->alpha_x, b->alpha_y};
            return scaled(Spec(1.0f) - f) * S3(b->r) *
                   std::fabs(tr.d(wh) * tr.g(wo, wi) * eta * eta * abs_dot(wi, wh) * abs_dot(wo, wh) * factor * factor / (cti * cto * sqrt_denom * sqrt_denom));
        }
        case RSPT_BXDF_FRESNEL_BLEND: { // :1398-1431; rd = r, rs = t
            Spec rd = S3(b->r), rs = S3(b->t);
            Spec diffuse = rd * (Spec(1.0f) - rs) * (28.0f / (23.0f * PI)) * (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wi))) *
                           (1.0f - pow5(1.0f - 0.5f * abs_cos_theta(wo)));
            V3 wh = wi + wo;
            if (wh.x == 0.0f && wh.y == 0.0f && wh.z == 0.0f) return Spec(0.0f);
            wh = normalize(wh);
            TR tr{b->alpha_x, b->alpha_y};
            Spec schlick = rs + (Spec(1.0f) - rs) * pow5(1.0f - dot(wi, wh));
            Spec specular = schlick * (tr.d(wh) / (4.0f * std::fabs(dot(wi, wh)) * std::fmax(abs_cos_theta(wi), abs_cos_theta(wo))));
            return b->has_sc ? S3(b->sc) * (diffuse + specular) : diffuse + specular;
        }
        default: return Spec(0.0f); // specular lobes :721,782,866
        }
    }
    Float pdf(V3 wo, V3 wi) const {
        switch (b->type) {
        case RSPT_BXDF_LAMBERT_R: case RSPT_BXDF_OREN_NAYAR:
            return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * INV_PI : 0.0f; // :988,1115
        case RSPT_BXDF_LAMBERT_T:
            return !same_hemisphere(wo, wi) ? abs_cos_theta(wi) * INV_PI : 0.0f; // :1036
        case RSPT_BXDF_SPECULAR_R: return 0.0f; // :746
        case RSPT_BXDF_SPECULAR_T: case RSPT_BXDF_FRESNEL_SPEC: // Q5: cosine pdf, not 0 (:828-834, :938-944)
            return same_hemisphere(wo, wi) ? abs_cos_theta(wi) * INV_PI : 0.0f;
        case RSPT_BXDF_MICROFACET_R: { // :1197-1203
            if (!same_hemisphere(wo, wi)) return 0.0f;
            V3 wh = normalize(wo + wi);
            TR tr{b->alpha_x, b->alpha_y};
            return tr.pdf(wo, wh) / (4.0f * dot(wo, wh));
        }
        case RSPT_BXDF_MICROFACET_T: { // :1350-1370
            if (same_hemisphere(wo, wi)) return 0.0f;
            Float eta = cos_theta(wo) > 0.0f ? b->eta_b / b->eta_a : b->eta_a / b->eta_b;
            V3 wh = normalize(wo + wi * eta);
            Float wo_dot_wh = dot(wo, wh), wi_dot_wh = dot(wi, wh);
            if (wo_dot_wh * wi_dot_wh > 0.0f) return 0.0f;
            Float sqrt_denom = wo_dot_wh + eta * wi_dot_wh;
            Float dwh_dwi = std::fabs((eta * eta * wi_dot_wh) / (sqrt_denom * sqrt_denom));
            TR tr{b->alpha_x, b->alpha_y};
            return tr.pdf(wo, wh) * dwh_dwi;
        }
        case RSPT_BXDF_FRESNEL_BLEND: { // :1462-1474
            if (!same_hemisphere(wo, wi)) return 0.0f;
            V3 wh = normalize(wo + wi);
            TR tr{b->alpha_x, b->alpha_y};
            Float pdf_wh = tr.pdf(wo, wh);
            return 0.5f * (abs_cos_theta(wi) * INV_PI + pdf_wh / (4.0f * dot(wo, wh)));
        }
        }
        return 0.0f;
    }
    // The non-specular lobes end their sample_f with `if let Some(sc) = self.sc_opt { sc * self.f(wo, wi) } else { self.f(wo, wi) }` (reflection.rs:982-986, 1030-1034,
    // 1109-1113, 1190-1194, 1339-1343, 1456-1460) while f already carries sc: inside a MixMaterial the scale enters the value of a lobe's OWN sample_f twice.  Bsdf::sample_f
    // discards that value for every non-specular lobe (it re-sums f over the matching lobes, :393-412), so no radiance depends on it; kept because the lobe-level pin
    // (oracle/make_flow_fixtures.py flow_lobes, found by it) holds this function to the reference's text bit for bit.
    Spec sampled_value(V3 wo, V3 wi) const { return b->has_sc ? S3(b->sc) * f(wo, wi) : f(wo, wi); }
    Spec sample_f(V3 wo, V3* wi, P2 u, Float* pdf_out, uint8_t* sampled_type) const {
        switch (b->type) {
        case RSPT_BXDF_LAMBERT_R: case RSPT_BXDF_OREN_NAYAR: { // :968-987, :1097-1114
            *wi = cosine_sample_hemisphere(u);
            if (wo.z < 0.0f) wi->z *= -1.0f;
            *pdf_out = pdf(wo, *wi);
            return sampled_value(wo, *wi);
        }
        case RSPT_BXDF_LAMBERT_T: { // :1018-1035
            *wi = cosine_sample_hemisphere(u);
            if (wo.z > 0.0f) wi->z *= -1.0f;
            *pdf_out = pdf(wo, *wi);
            return sampled_value(wo, *wi);
        }
        case RSPT_BXDF_SPECULAR_R: { // :724-745
            *wi = V3{-wo.x, -wo.y, wo.z};
            *pdf_out = 1.0f;
            return scaled(fresnel(cos_theta(*wi))) * S3(b->r) / abs_cos_theta(*wi);
        }
        case RSPT_BXDF_SPECULAR_T: { // :785-826
            bool entering = cos_theta(wo) > 0.0f;
            Float eta_i = entering ? b->eta_a : b->eta_b, eta_t = entering ? b->eta_b : b->eta_a;
            if (!refract(wo, faceforward(V3{0, 0, 1}, wo), eta_i / eta_t, wi)) return Spec();
            *pdf_out = 1.0f;
            Spec ft = S3(b->r) * (Spec(1.0f) - Spec(fr_dielectric(cos_theta(*wi), b->eta_a, b->eta_b)));
            ft = ft * Spec((eta_i * eta_i) / (eta_t * eta_t)); // TransportMode::Radiance
            return scaled(ft) / abs_cos_theta(*wi);
        }
        case RSPT_BXDF_FRESNEL_SPEC: { // :869-936
            Float ct = cos_theta(wo);
            Float fr = fr_dielectric(ct, b->eta_a, b->eta_b);
            if (u.x < fr) {
                *wi = V3{-wo.x, -wo.y, wo.z};
                if (*sampled_type != 0) *sampled_type = BSDF_REFLECTION | BSDF_SPECULAR;
                *pdf_out = fr;
                return scaled(S3(b->r)) * fr / abs_cos_theta(*wi);
            } else {
                bool entering = cos_theta(wo) > 0.0f;
                Float eta_i = entering ? b->eta_a : b->eta_b, eta_t = entering ? b->eta_b : b->eta_a;
                if (!refract(wo, faceforward(V3{0, 0, 1}, wo), eta_i / eta_t, wi)) return Spec();
                Spec ft = S3(b->t) * (1.0f - fr);
                ft = ft * Spec((eta_i * eta_i) / (eta_t * eta_t));
                if (*sampled_type != 0) *sampled_type = BSDF_TRANSMISSION | BSDF_SPECULAR;
                *pdf_out = 1.0f - fr;
                return scaled(ft) / abs_cos_theta(*wi);
            }
        }
        case RSPT_BXDF_MICROFACET_R: { // :1172-1195
            if (wo.z == 0.0f) return Spec();
            TR tr{b->alpha_x, b->alpha_y};
            V3 wh = tr.sample_wh(wo, u);
            *wi = reflect(wo, wh);
            if (!same_hemisphere(wo, *wi)) return Spec();
            *pdf_out = tr.pdf(wo, wh) / (4.0f * dot(wo, wh));
            return sampled_value(wo, *wi);
        }
        case RSPT_BXDF_MICROFACET_T: { // :1322-1349
            if (wo.z == 0.0f) return Spec();
            TR tr{b->alpha_x, b->alpha_y};
            V3 wh = tr.sample_wh(wo, u);
            Float eta = cos_theta(wo) > 0.0f ? b->eta_a / b->eta_b : b->eta_b / b->eta_a;
            if (!refract(wo, wh, eta, wi)) return Spec();
            *pdf_out = pdf(wo, *wi);
            return sampled_value(wo, *wi);
        }
        case RSPT_BXDF_FRESNEL_BLEND: { // :1432-1461
            P2 uu = u;
            if (uu.x < 0.5f) {
                uu.x = std::fmin(2.0f * uu.x, FLOAT_ONE_MINUS_EPSILON);
                *wi = cosine_sample_hemisphere(uu);
                if (wo.z < 0.0f) wi->z *= -1.0f;
            } else {
                uu.x = std::fmin(2.0f * (uu.x - 0.5f), FLOAT_ONE_MINUS_EPSILON);
                TR tr{b->alpha_x, b->alpha_y};
                V3 wh = tr.sample_wh(wo, uu);
                *wi = reflect(wo, wh);
                if (!same_hemisphere(wo, *wi)) return Spec(0.0f);
            }
            *pdf_out = pdf(wo, *wi);
            return sampled_value(wo, *wi);
        }
        }
        return Spec();
    }
};

// src/core/reflection.rs:223-446
struct Bsdf {
    Float eta;
    V3 ns, ng, ss, ts;
    Lobe lobes[8];
    rspt_bxdf local[8]; // Bsdf.bxdfs: Vec<Bxdf> with capacity 8
    int n = 0;
    Bsdf() = default;
    Bsdf(const Bsdf&) = delete;
    Bsdf& operator=(const Bsdf&) = delete;
    // Bsdf::new (reflection.rs:235-245)
    void init(const Interaction& si, Float eta_) {
        eta = eta_;
        ss = normalize(si.sh_dpdu);
        ns = si.sh_n; ng = si.n;
        ts = cross(si.sh_n, ss); // nrm_cross_vec3
        n = 0;
    }
    // Bsdf::add (reflection.rs:246-249)
    void add(const rspt_bxdf& b) {
        if (n >= 8) { std::fprintf(stderr, "oracle: assertion failed: self.bxdfs.len() < MAX_BXDFS (reflection.rs:247)\n"); std::abort(); }
        local[n] = b;
        lobes[n] = Lobe{&local[n]};
        n++;
    }
    int num_components(uint8_t flags) const { int c = 0; for (int i = 0; i < n; i++) if (lobes[i].matches_flags(flags)) c++; return c; }
    V3 world_to_local(V3 v) const { return V3{dot(v, ss), dot(v, ts), dot(v, ns)}; }
    V3 local_to_world(V3 v) const {
        return V3{ss.x * v.x + ts.x * v.y + ns.x * v.z, ss.y * v.x + ts.y * v.y + ns.y * v.z, ss.z * v.x + ts.z * v.y + ns.z * v.z};
    }
    Spec f(V3 wo_w, V3 wi_w, uint8_t flags) const { // :274-297
        V3 wi = world_to_local(wi_w), wo = world_to_local(wo_w);
        if (wo.z == 0.0f) return Spec(0.0f);
        bool refl = (dot(wi_w, ng) * dot(wo_w, ng)) > 0.0f;
        Spec f(0.0f);
        for (int i = 0; i < n; i++)
            if (lobes[i].matches_flags(flags) && ((refl && (lobes[i].get_type() & BSDF_REFLECTION)) || (!refl && (lobes[i].get_type() & BSDF_TRANSMISSION))))
                f = f + lobes[i].f(wo, wi);
        return f;
    }
    Spec sample_f(V3 wo_world, V3* wi_world, P2 u, Float* pdf, uint8_t flags, uint8_t* sampled_type) const { // :298-420
        int matching = num_components(flags);
        if (matching == 0) { *pdf = 0.0f; *sampled_type = 0; return Spec(); }
        int comp = std::min((int)f2u8(std::floor(u.x * (Float)matching)), matching - 1);
        int idx = -1, count = comp;
        for (int i = 0; i < n; i++) {
            bool m = lobes[i].matches_flags(flags);
            if (m && count == 0) { idx = i; break; }
            else if (m) count -= 1;
        }
        if (idx < 0) return Spec();
        const Lobe& bx = lobes[idx];
        P2 ur{std::fmin(u.x * (Float)matching - (Float)comp, FLOAT_ONE_MINUS_EPSILON), u.y};
        V3 wi{0, 0, 0};
        V3 wo = world_to_local(wo_world);
        if (wo.z == 0.0f) return Spec();
        *pdf = 0.0f;
        if (*sampled_type != 0) *sampled_type = bx.get_type();
        Spec f = bx.sample_f(wo, &wi, ur, pdf, sampled_type);
        if (*pdf == 0.0f) { if (*sampled_type != 0) *sampled_type = 0; return Spec(); }
        *wi_world = local_to_world(wi);
        if (!(bx.get_type() & BSDF_SPECULAR) && matching > 1)
            for (int i = 0; i < n; i++)
                if (i != idx && lobes[i].matches_flags(flags)) *pdf += lobes[i].pdf(wo, wi);
        if (matching > 1) *pdf /= (Float)matching;
        if (!(bx.get_type() & BSDF_SPECULAR)) {
            bool refl = dot(*wi_world, ng) * dot(wo_world, ng) > 0.0f;
            f = Spec();
            for (int i = 0; i < n; i++)
                if (lobes[i].matches_flags(flags) && ((refl && (lobes[i].get_type() & BSDF_REFLECTION)) || (!refl && (lobes[i].get_type() & BSDF_TRANSMISSION))))
                    f = f + lobes[i].f(wo, wi);
        }
        return f;
    }
    Float pdf(V3 wo_world, V3 wi_world, uint8_t flags) const { // :421-446
        if (n == 0) return 0.0f;
        V3 wo = world_to_local(wo_world), wi = world_to_local(wi_world);
        if (wo.z == 0.0f) return 0.0f;
        Float p = 0.0f;
        int matching = 0;
        for (int i = 0; i < n; i++)
            if (lobes[i].matches_flags(flags)) { matching++; p += lobes[i].pdf(wo, wi); }
        return matching > 0 ? p / (Float)matching : 0.0f;
    }
};

} // namespace orc
