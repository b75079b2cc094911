// DirectLightingIntegrator::li for ONE camera sample on ONE lane (src/integrators/directlighting.rs:71-258) — the form the pixel samplers
// need (tile_serial.h).  A lane owns the whole sample, so the recursion into specular reflection AND transmission runs in the
// reference's own depth-first order on an explicit stack of at most max_depth frames: no 2^max_depth tree of node slots (direct.h, the
// wavefront form under Sobol' / Halton, is bounded by max_depth 8 for that reason; this one by RSPT_DL_SERIAL_DEPTH), and the sampler
// — dimensions and the 2-D sample arrays of uniform_sample_all_lights alike — is simply read in program order.
// The sample source is a template parameter read in program order: a pixel sampler's stream (tile_serial.h) or, one lane per camera
// sample, the Sobol' / Halton dimensions (lane_serial.h) — the form rspt_render takes under those samplers when the wavefront form
// cannot serve the render (textured materials, max_depth > 8, a material with several specular lobes of one kind).
// Textured materials: every activation carries its ray's differentials — the camera's at the root, specular_reflect's / _transmit's
// (directlighting.rs:150-191, :215-250) below it, none behind a BSDF-less surface (isect.spawn_ray) — through
// SurfaceInteraction::compute_differentials into the texture stage (kernels.h texture_hit), whose rows this activation keeps for as
// long as its BSDF lives (level L of the lane: tex + L * tex_rows * tex_stride).
#pragma once
#include "vol_serial.h"

namespace rspt {

#define RSPT_DL_SERIAL_DEPTH 32

template <bool INST, bool ALPHA, class SMP, bool ANIM = false>
struct DlSerial {
    VolSerial<INST, ALPHA, ANIM> base;      // closest(), surface()
    SMP* px;                          // get_1d / get_2d / get_2d_array / va
    const int32_t* n_light_samples;   // strategy all: Light::get_n_samples after round_count, per light (nullptr: 1 each)
    bool sample_all;
    float4* tex;                      // this lane's column of texture-stage rows (nullptr: the scene has no textured material)
    uint32_t tex_stride, tex_rows;    // lanes per row; rows per activation
    f2 p_film; f3 p_lens;             // the camera sample: film position, (lens x, lens y, time) — differentials of the camera ray
    rspt_mat::Built* dyn;             // dynamic materials: this lane's column of per-activation lobe records (level L: dyn + L * dyn_stride); nullptr = none
    uint32_t dyn_stride;

    const SceneDev& sc() const { return base.sc; }
    RDEV bool occluded(f3 o, f3 d, float t_max) { return serial_trace<true, INST, ALPHA, ANIM>(base.sc, base.tt, o, d, t_max, base.lds, base.time).prim != RSPT_MISS; }

    // estimate_direct (integrator.rs:406-570), handle_media = false, specular = false
    RDEVN rgb estimate_direct(const SerialHit& it, const Bsdf& bsdf, f2 u_scattering, uint32_t light_num, f2 u_light) {
        const SceneDev& S = base.sc;
        const rspt_light lt = S.lights[light_num];
        const uint32_t nonspec = BX_ALL & ~BX_SPEC;
        rgb l_d = mkrgb(0.0f);
        f3 wi{0.0f, 0.0f, 0.0f};
        float light_pdf = 0.0f, scattering_pdf = 0.0f;
        LightSample ls;
        rgb li = light_sample_li(S, lt, it.h.p, u_light, &wi, &light_pdf, &ls);
        if (light_pdf > 0.0f && !is_black(li)) {
            const rgb f = bsdf.f(it.wo, wi, nonspec) * mkrgb(absdot(wi, it.h.sh_n));
            scattering_pdf = bsdf.pdf(it.wo, wi, nonspec);
            if (!is_black(f)) {
                // VisibilityTester::unoccluded -> spawn_ray_to (interaction.rs:81-94)
                const f3 origin = offset_ray_origin(it.h.p, it.h.p_err, it.h.n, ls.p - it.h.p);
                const f3 target = offset_ray_origin(ls.p, ls.p_err, ls.n, origin - ls.p);
                if (occluded(origin, target - origin, 1.0f - RSPT_SHADOW_EPS)) li = mkrgb(0.0f);
                if (!is_black(li)) {
                    if (light_is_delta(lt)) l_d = l_d + f * li / light_pdf;
                    else l_d = l_d + f * li * mkrgb(power_heuristic(light_pdf, scattering_pdf)) / light_pdf;
                }
            }
        }
        if (!light_is_delta(lt)) {  // BSDF sample with MIS (:480-568); sampled_type sentinel 0 (Q6)
            uint32_t sampled_type = 0;
            rgb f = bsdf.sample_f(it.wo, &wi, u_scattering, &scattering_pdf, nonspec, &sampled_type);
            f = f * mkrgb(absdot(wi, it.h.sh_n));
            if (!is_black(f) && scattering_pdf > 0.0f) {
                const f3 ro = offset_ray_origin(it.h.p, it.h.p_err, it.h.n, wi);   // it.spawn_ray(&wi)
                float lpdf = 0.0f;
                if (lt.kind == RSPT_LIGHT_INFINITE) lpdf = infinite_pdf_li(S, lt, wi);
                else {  // DiffuseAreaLight::pdf_li -> Triangle::pdf_with_ref_point (triangle.rs:745-764)
                    const TriRec lt_tri = load_tri(S, lt.prim);
                    float t_l, lb0, lb1, lb2;
                    if (tri_test(lt_tri.p0, lt_tri.p1, lt_tri.p2, ro, ray_shear(wi), RSPT_INF, &t_l, &lb0, &lb1, &lb2)) {
                        Hit lh;
                        tri_fill(S, lt.prim, lt_tri, lb0, lb1, lb2, &lh);
                        lpdf = dist2(it.h.p, lh.p) / (absdot(lh.n, -wi) * tri_area(lt_tri));
                        if (__builtin_isinf(lpdf)) lpdf = 0.0f;
                    }
                }
                if (lpdf == 0.0f) return l_d;
                const float weight = power_heuristic(scattering_pdf, lpdf);
                const TraceResult r = base.closest(ro, wi, RSPT_INF);
                rgb li2 = mkrgb(0.0f);
                if (r.prim != RSPT_MISS) {
                    SerialHit lh;
                    base.surface(r, wi, 0u, &lh);
                    if (lt.kind == RSPT_LIGHT_DIFFUSE_AREA && lh.h.area_light >= 0 && (uint32_t)lh.h.area_light == light_num) li2 = light_l(lt, lh.h.n, -wi);
                } else if (lt.kind == RSPT_LIGHT_INFINITE) li2 = infinite_le(S, lt, wi);
                if (!is_black(li2)) l_d = l_d + f * li2 * mkrgb(1.0f) * weight / scattering_pdf;
            }
        }
        return l_d;
    }
    // uniform_sample_all_lights (integrator.rs:300-355) / uniform_sample_one_light without a distribution (:359-403)
    RDEVN rgb direct(const SerialHit& it, const Bsdf& bsdf) {
        const SceneDev& S = base.sc;
        rgb l = mkrgb(0.0f);
        if (S.n_lights == 0u) return l;
        if (!sample_all) {
            const float fl = px->get_1d() * (float)S.n_lights;
            uint32_t light_num = (fl != fl || fl <= 0.0f) ? 0u : (fl >= 4294967296.0f ? 0xffffffffu : (uint32_t)fl);   // `as usize`
            light_num = light_num < S.n_lights - 1u ? light_num : S.n_lights - 1u;
            const float light_pdf = 1.0f / (float)S.n_lights;
            const f2 u_light = px->get_2d();
            const f2 u_scattering = px->get_2d();
            return estimate_direct(it, bsdf, u_scattering, light_num, u_light) / light_pdf;
        }
        for (uint32_t j = 0; j < S.n_lights; j++) {
            const uint32_t n_samples = n_light_samples ? (uint32_t)n_light_samples[j] : 1u;
            uint32_t fa = 0, fb = 0, ca = 0, cb = 0;
            const bool have_a = px->get_2d_array(&fa, &ca);
            const bool have_b = px->get_2d_array(&fb, &cb);
            if (!have_a || !have_b) {   // the arrays are used up: one sample from the regular stream
                const f2 u_light = px->get_2d();
                const f2 u_scattering = px->get_2d();
                l = l + estimate_direct(it, bsdf, u_scattering, j, u_light);
            } else {
                rgb ld = mkrgb(0.0f);
                for (uint32_t k = 0; k < n_samples; k++) {
                    const float2 us = px->va(fb + k), ul = px->va(fa + k);
                    ld = ld + estimate_direct(it, bsdf, f2{us.x, us.y}, j, f2{ul.x, ul.y});
                }
                l = l + ld / (float)n_samples;
            }
        }
        return l;
    }

    struct Frame {   // one activation of `li` that is waiting for a specular child
        rgb l, f;    // radiance so far; the BSDF value of the child in flight
        float s;     // |wi . ns| / pdf of that child
        uint32_t stage;   // 1: the reflection child is in flight, 2: the transmission child
        SerialHit it;
        Bsdf bsdf;
        // the ray that reached this activation had differentials (only kept when the scene has textures): its offset directions, and
        // what compute_differentials made of them here — dpdx / dpdy, dndx = shading.dndu * dudx + shading.dndv * dvdx, dndy
        bool has_diff;
        f3 in_rx_d, in_ry_d, dpdx, dpdy, dndx, dndy;
    };
    struct RayDiff { bool has; f3 rx_o, rx_d, ry_o, ry_d; };
    // specular_reflect / specular_transmit (:133-258) up to the recursive call: draws its get_2d, returns whether a child ray was spawned
    RDEV bool specular(const Frame& fr, bool transmit, f3* o, f3* d, rgb* f_out, float* s_out, RayDiff* rdiff) {
        f3 wi{0.0f, 0.0f, 0.0f};
        float pdf = 0.0f;
        uint32_t st = 0;
        const rgb f = fr.bsdf.sample_f(fr.it.wo, &wi, px->get_2d(), &pdf, (transmit ? BX_TRANS : BX_REFL) | BX_SPEC, &st);
        const f3 ns = fr.it.h.sh_n;
        if (!(pdf > 0.0f && !is_black(f) && absdot(wi, ns) != 0.0f)) return false;
        *o = offset_ray_origin(fr.it.h.p, fr.it.h.p_err, fr.it.h.n, wi);
        *d = wi;
        *f_out = f; *s_out = absdot(wi, ns) / pdf;
        rdiff->has = false;
        if (tex && fr.has_diff) {   // the child's differentials (directlighting.rs:150-191 reflection, :215-250 transmission)
            const f3 wo = fr.it.wo;
            const f3 dwodx = -fr.in_rx_d - wo, dwody = -fr.in_ry_d - wo;
            const float ddndx = dot(dwodx, ns) + dot(wo, fr.dndx);
            const float ddndy = dot(dwody, ns) + dot(wo, fr.dndy);
            rdiff->has = true;
            rdiff->rx_o = fr.it.h.p + fr.dpdx; rdiff->ry_o = fr.it.h.p + fr.dpdy;
            if (!transmit) {
                rdiff->rx_d = wi - dwodx + (fr.dndx * dot(wo, ns) + ns * ddndx) * 2.0f;
                rdiff->ry_d = wi - dwody + (fr.dndy * dot(wo, ns) + ns * ddndy) * 2.0f;
            } else {
                float eta = fr.bsdf.eta;
                const f3 w = -wo;
                if (dot(wo, ns) < 0.0f) eta = 1.0f / eta;
                const float mu = eta * dot(w, ns) - dot(wi, ns);
                const float dmudx = (eta - (eta * eta * dot(w, ns)) / dot(wi, ns)) * ddndx;
                const float dmudy = (eta - (eta * eta * dot(w, ns)) / dot(wi, ns)) * ddndy;
                rdiff->rx_d = wi + dwodx * eta - (fr.dndx * mu + ns * dmudx);
                rdiff->ry_d = wi + dwody * eta - (fr.dndy * mu + ns * dmudy);
            }
        }
        return true;
    }

    RDEVN rgb li(f3 ray_o, f3 ray_d, float ray_tmax) {
        const SceneDev& S = base.sc;
        Frame stack[RSPT_DL_SERIAL_DEPTH];
        uint32_t sp = 0;          // = depth of the activation being entered
        rgb ret = mkrgb(0.0f);
        uint32_t walked = 0;
        RayDiff cur{false, f3{0.0f, 0.0f, 0.0f}, f3{0.0f, 0.0f, 0.0f}, f3{0.0f, 0.0f, 0.0f}, f3{0.0f, 0.0f, 0.0f}};   // the differentials of the ray being traced
        if (tex) { cur.has = true; camera_differentials(base.rd, p_film, p_lens, ray_o, ray_d, &cur.rx_o, &cur.rx_d, &cur.ry_o, &cur.ry_d); }
        for (;;) {
            // ---- enter li(ray, depth = sp) ----
            rgb l = mkrgb(0.0f);
            bool spawned = false;
            {
                const TraceResult r = base.closest(ray_o, ray_d, ray_tmax);
                ray_tmax = RSPT_INF;
                if (r.prim == RSPT_MISS) {
                    for (uint32_t k = 0; k < S.n_infinite; k++) l = l + infinite_le(S, S.lights[S.infinite_lights[k]], ray_d);   // every light's le(ray): only the infinite lights'
                } else {
                    Frame& fr = stack[sp];
                    base.surface(r, ray_d, 0u, &fr.it);
                    if (fr.it.h.material == 0xffffffffu) {   // no BSDF: li(isect.spawn_ray(ray.d), depth) (:87-89)
                        if (++walked > base.max_walk) { base.truncated = true; }
                        else {
                            ray_o = offset_ray_origin(fr.it.h.p, fr.it.h.p_err, fr.it.h.n, ray_d);
                            cur.has = false;   // isect.spawn_ray(&ray.d): no differentials
                            continue;
                        }
                    } else {
                        const rspt_material mat = S.materials[fr.it.h.material];
                        Bsdf& b = fr.bsdf;
                        b.eta = mat.eta; b.lt = LobeTex{nullptr, 0}; b.dropped = 0u;
                        const rspt_bxdf* dyn_l = nullptr;
                        uint32_t dyn_n = 0u;
                        fr.has_diff = false;
                        if (tex) {   // compute_scattering_functions: compute_differentials(ray), then the material's textures / bump map
                            TexHit th;
                            tri_fill_tex(S, r.prim, load_tri(S, r.prim), r.b0, r.b1, r.b2, &th);
                            if (INST && r.inst) {
                                const bool moving = ANIM && S.inst[r.inst - 1u].anim != RSPT_MISS;
                                InstDev moved;
                                if (moving) moved = inst_at(S, r.inst - 1u, base.time);
                                const InstDev& in = moving ? moved : S.inst[r.inst - 1u];
                                if (!in.identity) inst_texhit(in, &th);
                            }
                            TexSurf ts;
                            ts.p = th.p; ts.uv = th.uv;
                            ts.dudx = ts.dvdx = ts.dudy = ts.dvdy = 0.0f;
                            ts.dpdx = ts.dpdy = f3{0.0f, 0.0f, 0.0f};
                            if (cur.has) compute_differentials(th, cur.rx_o, cur.rx_d, cur.ry_o, cur.ry_d, &ts);
                            fr.has_diff = cur.has;
                            fr.in_rx_d = cur.rx_d; fr.in_ry_d = cur.ry_d;
                            fr.dpdx = ts.dpdx; fr.dpdy = ts.dpdy;
                            fr.dndx = th.sh_dndu * ts.dudx + th.sh_dndv * ts.dvdx;
                            fr.dndy = th.sh_dndu * ts.dudy + th.sh_dndv * ts.dvdy;
                            if (S.mat_flags && S.mat_flags[fr.it.h.material]) {
                                float4* rows = tex + (size_t)sp * tex_rows * tex_stride;
                                texture_hit_call(S, base.tt, th, ts, fr.it.h.material, rows, tex_stride);
                                if (S.mat_flags[fr.it.h.material] & RSPT_MAT_DYNAMIC) {   // the lobe list is built from this hit's texture values; the activation keeps its own record
                                    const rspt_mat::Built* bl = dynamic_lobes(S.dyn[fr.it.h.material], rows, tex_stride, false /* directlighting.rs:86 */, dyn + (size_t)sp * dyn_stride);
                                    dyn_l = bl->l; dyn_n = bl->n;
                                    b.eta = bl->eta;
                                }
                                b.lt = LobeTex{rows, tex_stride};
                                const float4 m4 = rows[4 * (size_t)tex_stride];
                                const uint32_t tf = __float_as_uint(m4.w);
                                b.dropped = (tf >> 8) & 0xffu;
                                if (tf & 1u) {   // Material::bump replaced the shading geometry
                                    const float4 d4 = rows[5 * (size_t)tex_stride];
                                    fr.it.h.sh_n = f3{m4.x, m4.y, m4.z};
                                    fr.it.h.sh_dpdu = f3{d4.x, d4.y, d4.z};
                                }
                            }
                        }
                        b.ss = normalize(fr.it.h.sh_dpdu); b.ns = fr.it.h.sh_n; b.ng = fr.it.h.n; b.ts = cross(fr.it.h.sh_n, b.ss);
                        b.lobes = dyn_l ? dyn_l : S.bxdfs + mat.first_bxdf;
                        b.n = dyn_l ? (dyn_n < 8u ? dyn_n : 8u) : (mat.n_bxdfs < 8u ? mat.n_bxdfs : 8u);
                        if (fr.it.h.area_light >= 0) l = l + light_l(S.lights[fr.it.h.area_light], fr.it.h.n, fr.it.wo);   // isect.le(&wo)
                        l = l + direct(fr.it, b);
                        if (sp + 1u < base.rd.max_depth && sp + 1u < RSPT_DL_SERIAL_DEPTH) {
                            fr.l = l;
                            f3 co, cd;
                            if (specular(fr, false, &co, &cd, &fr.f, &fr.s, &cur)) { fr.stage = 1u; ray_o = co; ray_d = cd; sp++; spawned = true; }
                            else if (specular(fr, true, &co, &cd, &fr.f, &fr.s, &cur)) { fr.stage = 2u; ray_o = co; ray_d = cd; sp++; spawned = true; }
                        }
                    }
                }
            }
            if (spawned) continue;
            // ---- return `l` to the activation that waits for it ----
            ret = l;
            for (;;) {
                if (sp == 0u) return ret;
                Frame& fr = stack[sp - 1u];
                fr.l = fr.l + fr.f * ret * mkrgb(fr.s);   // f * li(child) * Spectrum(|wi . ns| / pdf)
                if (fr.stage == 1u) {   // the reflection subtree is done: now the transmission side draws its sample
                    f3 co, cd;
                    if (specular(fr, true, &co, &cd, &fr.f, &fr.s, &cur)) { fr.stage = 2u; ray_o = co; ray_d = cd; break; }
                }
                ret = fr.l;
                sp--;
            }
        }
    }
};

}  // namespace rspt
