// VolPathIntegrator::li on the GPU (src/integrators/volpath.rs:60-347) with homogeneous media
// (src/media/homogeneous.rs, src/core/medium.rs).
//
// One wavefront iteration = one pass of the reference's loop body for every live path:
//   trace the continuation rays (closest hit)                                                 launch_trace
//   k_vol_shade   Medium::sample on the segment up to the hit (or to infinity), then either the medium interaction (light
//                 estimate with the phase function, HenyeyGreenstein::sample_p) or the surface interaction (Le, null-BSDF pass,
//                 light estimate, BSDF sample), Russian roulette, next ray with the medium Interaction::get_medium picks
//   [k_vol_tr]*   VisibilityTester::tr (light.rs:207-239) for the estimates' shadow rays: segment by segment through surfaces
//                 without material (closest-hit traces, transmittance of the medium each segment crosses), until every ray has
//                 reached its light or an opaque surface; the estimate is added to the path's radiance there, before the next
//                 iteration adds anything else (the reference's order of additions).
// As in v0.9.12, the BSDF- / phase-sampled half of estimate_direct (integrator.rs:480-568) adds nothing with handle_media: the
// transmittance Scene::intersect_tr multiplies into starts at Spectrum::default() (integrator.rs:531, scene.rs:86).  Homogeneous
// media draw no sample there, so the half is left out altogether; the two sample dimensions it would use are consumed.
#pragma once
#include "direct.h"

namespace rspt {

struct VolBuf {         // per path slot, next to PathBuf
    uint32_t* medium;   // the continuation ray's medium: 0 = none, else 1 + index (Ray.medium)
    float4* sh;         // shadow ray in flight: (transmittance so far .rgb, medium of the current segment)
    float4* p1_p;       // the light sample it aims at: InteractionCommon p / p_error / n (VisibilityTester.p1)
    float4* p1_e;
    float4* p1_n;
    float4* post;       // radiance to add after the estimate (volpath.rs:332-337 runs after :304-330), w != 0: present
    uint32_t* truncated;  // paths cut because the sampler ran out of dimensions (the reference panics there)
    const uint32_t* hit_inst_tr;  // scenes with object instances: 0 or 1 + instance of the shadow-ray segment's hit, by slot (the continuation ray's is pb.hit_inst)
};
// pb.nee_c1 = (f.rgb, light_pdf)   pb.nee_c2 = (li.rgb, MIS weight | < 0 for a delta light)   pb.nee_beta = (beta.rgb, light choice pdf)
// pb.ray_mis / pb.hit_mis = the shadow ray's current segment and its closest hit

struct VolSampler {  // GlobalSampler::get_1d / get_2d (sobol.rs:180-201, halton.rs) without sample arrays: dimensions in order.
                     // Sobol': eight dimensions at a time from the LDS copy of the generator matrices (SobolBlock, as k_shade), refilled
                     // when a pass needs more (a pass can draw 2 + 5 + 2 + 1)
    SobolBlock blk;
    const uint32_t* tab;
    uint32_t nd;
    uint64_t index;
    uint32_t hdim;
    bool halton;
    RDEV void start(const RenderDev& rd, const uint32_t* t, uint32_t n, uint64_t idx, uint32_t first_dim) {
        halton = rd.sampler_kind == RSPT_SAMPLER_HALTON;
        tab = t; nd = n; index = idx; hdim = first_dim;
        if (!halton) blk.fill(tab, nd, idx, first_dim);
    }
    RDEV uint32_t dim() const { return halton ? hdim : blk.dim; }
    RDEV float get_1d(const RenderDev& rd) {
        if (halton) return halton_dim(rd, index, hdim++);
        if (blk.dim + 1u > blk.base + 8u) blk.fill(tab, nd, index, blk.dim);
        return blk.get_1d();
    }
    RDEV f2 get_2d(const RenderDev& rd) {
        if (halton) { const f2 v = dl_dims(rd, index, hdim); hdim += 2u; return v; }
        if (blk.dim + 2u > blk.base + 8u) blk.fill(tab, nd, index, blk.dim);
        return blk.get_2d();
    }
};

RDEV rgb med_sigma_t(const rspt_medium& m) { return ldrgb(m.sigma_s) + ldrgb(m.sigma_a); }  // HomogeneousMedium::new (homogeneous.rs:24-31)
RDEV rgb rgb_exp(rgb a) { return rgb{rspt_expf(a.r), rspt_expf(a.g), rspt_expf(a.b)}; }
// HomogeneousMedium::tr (homogeneous.rs:33-36) over a ray of parametric length t_max and direction length len
RDEV rgb med_tr(const rspt_medium& m, float t_max, float len) {
    const rgb st = med_sigma_t(m);
    return rgb_exp(rgb{-st.r, -st.g, -st.b} * fminf(t_max * len, 3.402823466e+38f));
}
RDEV float phase_hg(float cos_theta, float g) {  // medium.rs:389-392
    const float denom = 1.0f + g * g + 2.0f * g * cos_theta;
    return 0.07957747154594766788f * (1.0f - g * g) / (denom * sqrtf(denom));
}
RDEV float hg_sample_p(float g, f3 wo, f3* wi, f2 u) {  // HenyeyGreenstein::sample_p (medium.rs:306-331)
    float cos_theta;
    if (fabsf(g) < 1e-3f) cos_theta = 1.0f - 2.0f * u.x;
    else {
        const float sqr_term = (1.0f - g * g) / (1.0f + g - 2.0f * g * u.x);
        cos_theta = -(1.0f + g * g - sqr_term * sqr_term) / (2.0f * g);
    }
    const float sin_theta = sqrtf(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
    const float phi = 2.0f * RSPT_PI * u.y;
    f3 v1, v2;
    coordinate_system(wo, &v1, &v2);
    *wi = v1 * (sin_theta * rspt_cosf(phi)) + v2 * (sin_theta * rspt_sinf(phi)) + wo * cos_theta;
    return phase_hg(cos_theta, g);
}
// GeometricPrimitive::intersect's medium interface (primitive.rs:160-170) + Interaction::get_medium (interaction.rs:95-107)
RDEV uint32_t surface_medium(const SceneDev& sc, uint32_t prim, uint32_t ray_medium, f3 n, f3 w) {
    const rspt_mesh me = sc.meshes[sc.prims[prim].mesh];
    uint32_t in = ray_medium, out = ray_medium;
    if (me.medium_inside != me.medium_outside) { in = me.medium_inside; out = me.medium_outside; }
    return dot(w, n) > 0.0f ? out : in;
}
// Triangle::intersect's t for a hit the trace kernel reported (it hands on the barycentrics only): the same arithmetic again
RDEV float hit_distance(const TriRec& t, f3 o, f3 d) {
    float th = 0.0f, b0, b1, b2;
    (void)tri_test(t.p0, t.p1, t.p2, o, ray_shear(d), RSPT_INF, &th, &b0, &b1, &b2);
    return th;
}

// the light estimate both interaction kinds share (uniform_sample_one_light integrator.rs:359-403 + the first half of estimate_direct
// :424-477 with handle_media): draws its five dimensions, leaves the shadow ray's first segment and the terms the resolve needs
struct VolRef {        // Interaction::get_common of the reference point
    f3 p, p_err, n, wo;
    uint32_t med_in, med_out;   // its medium interface
};
template <typename F>
RDEV bool vol_estimate(const SceneDev& sc, const LightDistDev& ld, const RenderDev& rd, const PathBuf& pb, const VolBuf& vb, uint32_t p, VolSampler& smp,
                       const VolRef& it, rgb beta, bool* retry, F&& scatter /* (wi, &pdf) -> f */) {
    // on-demand spatial distribution: a voxel without a row yet is claimed (light_row_try) and the path is put back, untouched, for the run
    // that follows the build of the claimed rows (librspt.hip batch_volpath) — nothing of this pass has been written for it at this point
    const int32_t row = light_row_try(ld, light_voxel(sc, ld, it.p));
    if (row < 0 && ld.lazy) { *retry = true; return false; }
    const uint32_t vox = row < 0 ? 0u : (uint32_t)row;
    float pdf_choice = 0.0f;
    const uint32_t light_num = sample_discrete(ld.func + (size_t)vox * sc.n_lights, ld.cdf + (size_t)vox * (sc.n_lights + 1), ld.func_int[vox], sc.n_lights, smp.get_1d(rd), &pdf_choice);
    if (pdf_choice == 0.0f) return false;
    const f2 u_light = smp.get_2d(rd);
    (void)smp.get_2d(rd);  // u_scattering: drawn (integrator.rs:392), used only by the half that adds nothing
    const rspt_light lt = sc.lights[light_num];
    f3 wi{0.0f, 0.0f, 0.0f};
    float light_pdf = 0.0f;
    LightSample ls;
    const rgb li = light_sample_li(sc, lt, it.p, u_light, &wi, &light_pdf, &ls);
    if (!(light_pdf > 0.0f) || is_black(li)) return false;
    float scattering_pdf = 0.0f;
    const rgb f = scatter(wi, &scattering_pdf);
    if (is_black(f)) return false;
    // VisibilityTester::tr: p0.spawn_ray_to(p1) (interaction.rs:81-94)
    const f3 origin = offset_ray_origin(it.p, it.p_err, it.n, ls.p - it.p);
    const f3 target = offset_ray_origin(ls.p, ls.p_err, ls.n, origin - ls.p);
    const f3 d = target - origin;
    store_ray(pb.ray_mis + p, origin, d, 1.0f - RSPT_SHADOW_EPS, p);
    const uint32_t seg_medium = dot(d, it.n) > 0.0f ? it.med_out : it.med_in;
    vb.sh[p] = make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(seg_medium));
    vb.p1_p[p] = make_float4(ls.p.x, ls.p.y, ls.p.z, 0.0f);
    vb.p1_e[p] = make_float4(ls.p_err.x, ls.p_err.y, ls.p_err.z, 0.0f);
    vb.p1_n[p] = make_float4(ls.n.x, ls.n.y, ls.n.z, 0.0f);
    pb.nee_c1[p] = make_float4(f.r, f.g, f.b, light_pdf);
    pb.nee_c2[p] = make_float4(li.r, li.g, li.b, light_is_delta(lt) ? -1.0f : power_heuristic(light_pdf, scattering_pdf));
    pb.nee_beta[p] = make_float4(beta.r, beta.g, beta.b, pdf_choice);
    return true;
}

// one pass of the loop body of VolPathIntegrator::li for every live path
// DYN: the scene has a material whose lobe list is built per hit (material_assembly.h; shade_path<.., SF_DYNAMIC> is the `path` counterpart)
template <bool DYN>
__global__ __launch_bounds__(256) void k_vol_shade(SceneDev sc, LightDistDev ld, RenderDev rd, PathBuf pb, VolBuf vb, const uint32_t* __restrict__ queue,
                                                   const uint32_t* __restrict__ count_in, uint32_t* __restrict__ q_next, uint32_t* cnt_next,
                                                   uint32_t* __restrict__ q_tr, uint32_t* cnt_tr, uint32_t dim_limit, uint32_t sob_nd, uint32_t sob_bits,
                                                   uint32_t* __restrict__ q_retry, uint32_t* cnt_retry) {
    extern __shared__ uint32_t sob_tab[];  // Sobol' generator matrices of the dimensions a path can reach, transposed to [bit][dim] (as k_shade)
    for (uint32_t t = threadIdx.x; rd.sampler_kind == RSPT_SAMPLER_SOBOL && t < sob_nd * sob_bits; t += 256u) {
        const uint32_t dd = t % sob_nd;
        sob_tab[t] = rd.sobol32[(dd < 1024u ? dd : 1023u) * 52u + (t / sob_nd)];
    }
    __syncthreads();
    const uint32_t n = *count_in;
    for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
        const uint32_t i = base + threadIdx.x;
        bool go_on = false, shadow = false, retry = false;
        uint32_t p = 0;
        if (i < n) {
            p = queue[i];
            uint32_t st = pb.state[p];
            const float4 le = pb.L_eta[p];
            rgb L{le.x, le.y, le.z};
            float eta_scale = le.w;
            const float4 bb = pb.beta[p];
            rgb beta{bb.x, bb.y, bb.z};
            const float4* rp = reinterpret_cast<const float4*>(pb.ray_cont + p);
            const float4 r0 = rp[0], r1 = rp[1];
            const f3 ray_o{r0.x, r0.y, r0.z}, ray_d{r0.w, r1.x, r1.y};
            const float4 hc = pb.hit_cont[p];
            const uint32_t prim = __float_as_uint(hc.x);
            const bool hit = prim != RSPT_MISS;
            uint32_t bounces = (st >> ST_BOUNCE_SHIFT) & 0xffu;
            const uint32_t medium = vb.medium[p];
            VolSampler smp;
            smp.start(rd, sob_tab, sob_nd, pb.sobol_index[p], st & ST_DIM_MASK);
            bool specular = (st & ST_SPECULAR) != 0;
            rgb post = mkrgb(0.0f);
            bool have_post = false, counted = true;   // counted: this pass ends with `bounces += 1`
            f3 new_o = ray_o, new_d = ray_d;
            uint32_t new_medium = medium;

            if (smp.dim() + 12u > dim_limit) {  // the reference's sampler panics past its last dimension (sobol.rs:119-124): cut and report
                atomicAdd(vb.truncated, 1u);
            } else {
                TriRec tri{};
                float t_hit = RSPT_INF;
                // an instanced hit: the triangle lives in object space and ray.t_max is the OBJECT ray's parameter (TransformedPrimitive::intersect:
                // r.t_max.set(ray.t_max), primitive.rs:224; Transform::transform_ray has moved the origin by its error bound)
                const uint32_t hi = (hit && pb.hit_inst) ? pb.hit_inst[p] : 0u;
                if (!hit && pb.hit_inst) t_hit = hc.y;   // no hit reported, but an identity instance may have shortened ray.t_max (Q10): the medium is sampled up to there
                if (hit) {
                    tri = load_tri(sc, prim);
                    if (hi) {   // (inst_at: the instance's own Transform, or for a moving instance the one interpolated at the path's time — as the traversal used it)
                        f3 oo, od; float ot;
                        inst_ray(inst_at(sc, hi - 1u, sc.ray_time ? sc.ray_time[p] : 0.0f), ray_o, ray_d, RSPT_INF, &oo, &od, &ot);
                        t_hit = hit_distance(tri, oo, od);
                    } else t_hit = hit_distance(tri, ray_o, ray_d);
                }
                // ---- medium.sample(&ray, sampler) (volpath.rs:96-101 / :289-294; homogeneous.rs:37-91) ----
                bool have_mi = false;
                f3 mi_p{0.0f, 0.0f, 0.0f};
                float g = 0.0f;
                if (medium) {
                    const rspt_medium m = sc.media[medium - 1u];
                    const rgb sigma_t = med_sigma_t(m);
                    uint32_t channel = (uint32_t)(smp.get_1d(rd) * 3.0f);
                    channel = channel < 2u ? channel : 2u;
                    const float dist = -rspt_logf(1.0f - smp.get_1d(rd)) / (channel == 0u ? sigma_t.r : (channel == 1u ? sigma_t.g : sigma_t.b));
                    const float dlen = len(ray_d);
                    const float t = fminf(dist / dlen, t_hit);
                    have_mi = t < t_hit;
                    const rgb tr = rgb_exp(rgb{-sigma_t.r, -sigma_t.g, -sigma_t.b} * fminf(t, 3.402823466e+38f) * dlen);
                    const rgb density = have_mi ? sigma_t * tr : tr;
                    float pdf = 0.0f;
                    pdf += density.r; pdf += density.g; pdf += density.b;
                    pdf *= 1.0f / 3.0f;
                    if (pdf == 0.0f) pdf = 1.0f;
                    beta = beta * (have_mi ? tr * ldrgb(m.sigma_s) / pdf : tr / pdf);
                    mi_p = ray_o + ray_d * t;
                    g = m.g;
                }
                if (!is_black(beta)) {
                    if (have_mi) {
                        if (bounces < rd.max_depth) {
                            // ---- scattering at a point in the medium (:108-127 / :311-330) ----
                            const f3 wo = -ray_d;
                            if (sc.n_lights) {
                                const VolRef it{mi_p, f3{0.0f, 0.0f, 0.0f}, f3{0.0f, 0.0f, 0.0f}, wo, medium, medium};
                                shadow = vol_estimate(sc, ld, rd, pb, vb, p, smp, it, beta, &retry, [&](f3 wi, float* pdf) {
                                    const float ph = phase_hg(dot(wo, wi), g);  // HenyeyGreenstein::p (medium.rs:302-305)
                                    *pdf = ph;
                                    return mkrgb(ph);
                                });
                            }
                            f3 wi{0.0f, 0.0f, 0.0f};
                            (void)hg_sample_p(g, wo, &wi, smp.get_2d(rd));
                            new_o = mi_p; new_d = wi;   // mi.spawn_ray(&wi): n = 0 and p_error = 0 leave the origin where it is; the medium stays
                            specular = false;
                            if (hit) go_on = true;
                            else if (bounces == 0 && sc.n_infinite) {  // :332-337 with the SCATTERED ray; then the path ends (:338-339)
                                for (uint32_t k = 0; k < sc.n_infinite; k++) post = post + beta * infinite_le(sc, sc.lights[sc.infinite_lights[k]], new_d);
                                have_post = true;
                            }
                        }
                    } else if (hit) {
                        Hit h;
                        tri_fill(sc, prim, tri, hc.y, hc.z, hc.w, &h);
                        const f3 wo_ray = -ray_d;  // what `li` itself passes on: isect.le(&-ray.d), bsdf.sample_f(&-ray.d, ..) (volpath.rs:133, :170)
                        f3 wo = wo_ray;            // isect.common.wo, what estimate_direct reads (they differ for a transformed hit only)
                        // Transform::transform_surface_interaction (transform.rs:815-860) starts from SurfaceInteraction::default(): the transformed
                        // hit has no medium interface (get_medium gives None on either side) and, in v0.9.12, no primitive (Q11)
                        InstDev in{};
                        if (hi) in = inst_at(sc, hi - 1u, sc.ray_time ? sc.ray_time[p] : 0.0f);
                        const bool transformed = hi && !in.identity;
                        if (transformed) {
                            inst_hit(in, &h);
                            wo = normalize(xf_vector(in.m, -xf_vector(in.mi, ray_d)));
                            if (!sc.inst_fixed) { h.material = 0xffffffffu; h.area_light = -1; }
                        }
                        if (bounces == 0 || specular) {  // :133-136
                            const rgb e = h.area_light >= 0 ? light_l(sc.lights[h.area_light], h.n, wo_ray) : mkrgb(0.0f);
                            L = L + beta * e;
                        }
                        if (bounces < rd.max_depth) {
                            if (h.material == 0xffffffffu) {  // no BSDF: isect.spawn_ray(&ray.d); `continue` skips the bounce count and the roulette (:141-145)
                                new_o = offset_ray_origin(h.p, h.p_err, h.n, ray_d);
                                new_medium = transformed ? 0u : surface_medium(sc, prim, medium, h.n, ray_d);
                                go_on = true; counted = false;
                                st |= ST_NO_DIFF;   // the re-spawned ray carries no differentials (k_texture)
                            } else {
                                const rspt_material mat = sc.materials[h.material];
                                Bsdf b;  // Bsdf::new (reflection.rs:235-245)
                                b.eta = mat.eta; b.lt = LobeTex{nullptr, 0}; b.dropped = 0u;
                                const rspt_bxdf* lobes = sc.bxdfs + mat.first_bxdf;
                                uint32_t n_lobes = mat.n_bxdfs;
                                if (sc.mat_flags && sc.mat_flags[h.material]) {  // textured material: k_texture ran for this hit (as in shade_path)
                                    const float4* tb = pb.tex + p;
                                    if (DYN && (sc.mat_flags[h.material] & RSPT_MAT_DYNAMIC)) {
                                        const rspt_mat::Built* bl = dynamic_lobes(sc.dyn[h.material], tb, pb.tex_stride, true /* volpath.rs:146 */, pb.dyn_built + (blockIdx.x * blockDim.x + threadIdx.x));
                                        lobes = bl->l; n_lobes = bl->n;
                                        b.eta = bl->eta;
                                    }
                                    b.lt = LobeTex{tb, pb.tex_stride};
                                    const float4 m4 = tb[4 * (size_t)pb.tex_stride];
                                    const uint32_t tf = __float_as_uint(m4.w);
                                    b.dropped = (tf >> 8) & 0xffu;
                                    if (tf & 1u) {
                                        const float4 d4 = tb[5 * (size_t)pb.tex_stride];
                                        h.sh_n = f3{m4.x, m4.y, m4.z};
                                        h.sh_dpdu = f3{d4.x, d4.y, d4.z};
                                    }
                                }
                                b.ss = normalize(h.sh_dpdu); b.ns = h.sh_n; b.ng = h.n; b.ts = cross(h.sh_n, b.ss);
                                b.lobes = lobes;
                                b.n = n_lobes < 8u ? n_lobes : 8u;
                                const rspt_mesh me = sc.meshes[sc.prims[prim].mesh];
                                uint32_t m_in = medium, m_out = medium;
                                if (me.medium_inside != me.medium_outside) { m_in = me.medium_inside; m_out = me.medium_outside; }
                                if (transformed) m_in = m_out = 0u;
                                if (sc.n_lights) {  // no non-specular-lobe test in front of the estimate here (:146-161)
                                    const VolRef it{h.p, h.p_err, h.n, wo, m_in, m_out};
                                    const uint32_t nonspec = BX_ALL & ~BX_SPEC;
                                    shadow = vol_estimate(sc, ld, rd, pb, vb, p, smp, it, beta, &retry, [&](f3 wi, float* pdf) {
                                        const rgb f = b.f(wo, wi, nonspec) * mkrgb(absdot(wi, h.sh_n));
                                        *pdf = b.pdf(wo, wi, nonspec);
                                        return f;
                                    });
                                }
                                f3 wi{0.0f, 0.0f, 0.0f};
                                float pdf = 0.0f;
                                uint32_t sampled_type = 255;
                                const rgb f = b.sample_f(wo_ray, &wi, smp.get_2d(rd), &pdf, BX_ALL, &sampled_type);
                                if (!(is_black(f) || pdf == 0.0f)) {
                                    beta = beta * ((f * absdot(wi, h.sh_n)) / pdf);
                                    specular = (sampled_type & BX_SPEC) != 0;
                                    if ((sampled_type & BX_SPEC) && (sampled_type & BX_TRANS)) {
                                        const float eta = b.eta;
                                        if (dot(wo_ray, h.n) > 0.0f) eta_scale *= eta * eta;
                                        else eta_scale *= 1.0f / (eta * eta);
                                    }
                                    new_o = offset_ray_origin(h.p, h.p_err, h.n, wi);
                                    new_d = wi;
                                    new_medium = dot(wi, h.n) > 0.0f ? m_out : m_in;
                                    go_on = true;
                                }
                            }
                        }
                    } else if (sc.n_infinite && (bounces == 0 || specular)) {  // escaped without scattering (:332-337)
                        for (uint32_t k = 0; k < sc.n_infinite; k++) L = L + beta * infinite_le(sc, sc.lights[sc.infinite_lights[k]], ray_d);
                    }
                    // ---- Russian roulette (:275-285): inside the found-intersection branch, also after scattering in the medium ----
                    if (go_on && counted) {
                        const rgb rr = beta * eta_scale;
                        if (maxc(rr) < rd.rr_threshold && bounces > 3) {
                            const float q = fmaxf(0.05f, 1.0f - maxc(rr));
                            if (smp.get_1d(rd) < q) go_on = false;
                            else beta = beta / (1.0f - q);
                        }
                        bounces += 1;
                    }
                }
            }
            if (retry) { go_on = false; shadow = false; }   // the voxel of this path's light estimate is being built: the path runs again, from the state it came with
            else {
            if (go_on) {
                store_ray(pb.ray_cont + p, new_o, new_d, RSPT_INF, p);
                pb.beta[p] = make_float4(beta.r, beta.g, beta.b, 0.0f);
                vb.medium[p] = new_medium;
            }
            vb.post[p] = make_float4(post.r, post.g, post.b, have_post ? 1.0f : 0.0f);
            if (!shadow && have_post) L = L + post;   // nothing in flight: the estimate added 0
            pb.L_eta[p] = make_float4(L.r, L.g, L.b, eta_scale);
            st = (st & ~(ST_DIM_MASK | (0xffu << ST_BOUNCE_SHIFT) | ST_SPECULAR)) | (smp.dim() & ST_DIM_MASK) | ((bounces & 0xffu) << ST_BOUNCE_SHIFT) | (specular ? ST_SPECULAR : 0u);
            pb.state[p] = st;
            }
        }
        dl_push(go_on, p, q_next, cnt_next);
        dl_push(shadow, p, q_tr, cnt_tr);
        if (q_retry) dl_push(retry, p, q_retry, cnt_retry);
    }
}

// one segment of VisibilityTester::tr (light.rs:207-239) for every shadow ray in flight
template <bool ANIM>   // ANIM: the scene has moving instances (inst_at's interpolation costs this kernel two waves of occupancy: 92 -> 147 VGPRs)
__global__ __launch_bounds__(256) void k_vol_tr(SceneDev sc, PathBuf pb, VolBuf vb, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_in,
                                                uint32_t* __restrict__ q_next, uint32_t* cnt_next) {
    const uint32_t n = *count_in;
    for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
        const uint32_t i = base + threadIdx.x;
        bool again = false;
        uint32_t p = 0;
        if (i < n) {
            p = queue[i] & ~RSPT_Q_MIS;
            const float4* rp = reinterpret_cast<const float4*>(pb.ray_mis + p);
            const float4 r0 = rp[0], r1 = rp[1];
            const f3 o{r0.x, r0.y, r0.z}, d{r0.w, r1.x, r1.y};
            const float t_max = r1.z;
            const float4 sh = vb.sh[p];
            rgb tr{sh.x, sh.y, sh.z};
            const uint32_t medium = __float_as_uint(sh.w);
            const float4 hm = pb.hit_mis[p];
            const uint32_t prim = __float_as_uint(hm.x);
            bool done = false, blocked = false;
            if (prim != RSPT_MISS) {
                const TriRec tri = load_tri(sc, prim);
                const uint32_t hi = vb.hit_inst_tr ? vb.hit_inst_tr[p] : 0u;
                InstDev moved{};
                if (ANIM && hi) moved = inst_at(sc, hi - 1u, sc.ray_time ? sc.ray_time[p] : 0.0f);
                const InstDev& in = ANIM ? moved : sc.inst[hi ? hi - 1u : 0u];   // (a reference, not a copy: the static case keeps reading the record's fields from memory as it needs them)
                const bool transformed = hi && !in.identity;
                const bool no_primitive = transformed && !sc.inst_fixed;   // Q11: isect.primitive is None, neither branch of :216-229 runs
                if (!no_primitive && tri.material != 0xffffffffu) { blocked = true; done = true; }  // an opaque surface: Spectrum::default() (:218-222)
                else {
                    if (medium && !no_primitive) {   // ray.t_max is the hit distance now (an instanced hit: the object ray's)
                        float th;
                        if (hi) { f3 oo, od; float ot; inst_ray(in, o, d, t_max, &oo, &od, &ot); th = hit_distance(tri, oo, od); }
                        else th = hit_distance(tri, o, d);
                        tr = tr * med_tr(sc.media[medium - 1u], th, len(d));
                    }
                    Hit h;
                    tri_fill(sc, prim, tri, hm.y, hm.z, hm.w, &h);
                    if (transformed) inst_hit(in, &h);
                    const float4 pp = vb.p1_p[p], pe = vb.p1_e[p], pn = vb.p1_n[p];
                    const f3 lp{pp.x, pp.y, pp.z};
                    // isect.common.spawn_ray_to(p1) (:236)
                    const f3 origin = offset_ray_origin(h.p, h.p_err, h.n, lp - h.p);
                    const f3 target = offset_ray_origin(lp, f3{pe.x, pe.y, pe.z}, f3{pn.x, pn.y, pn.z}, origin - lp);
                    const f3 nd = target - origin;
                    store_ray(pb.ray_mis + p, origin, nd, 1.0f - RSPT_SHADOW_EPS, p);
                    vb.sh[p] = make_float4(tr.r, tr.g, tr.b, __uint_as_float(transformed ? 0u : surface_medium(sc, prim, medium, h.n, nd)));
                    again = true;
                }
            } else {
                if (medium) tr = tr * med_tr(sc.media[medium - 1u], vb.hit_inst_tr ? hm.y : t_max, len(d));   // (with instances: ray.t_max as the traversal left it, Q10)
                done = true;
            }
            if (done) {  // the tail of estimate_direct's first half (integrator.rs:461-476), uniform_sample_one_light's / pdf, l += beta * ..
                const float4 le = pb.L_eta[p];
                rgb L{le.x, le.y, le.z};
                if (!blocked) {
                    const float4 c1 = pb.nee_c1[p], c2 = pb.nee_c2[p], nb = pb.nee_beta[p];
                    const rgb f{c1.x, c1.y, c1.z};
                    const rgb li = rgb{c2.x, c2.y, c2.z} * tr;
                    if (!is_black(li)) {
                        const rgb ldir = c2.w < 0.0f ? f * li / c1.w : f * li * mkrgb(c2.w) / c1.w;
                        L = L + rgb{nb.x, nb.y, nb.z} * (ldir / nb.w);
                    }
                }
                const float4 post = vb.post[p];
                if (post.w != 0.0f) L = L + rgb{post.x, post.y, post.z};
                pb.L_eta[p] = make_float4(L.r, L.g, L.b, le.w);
            }
        }
        dl_push(again, p, q_next, cnt_next);
    }
}

// raygen leaves the camera rays outside every medium (make_camera: MediumInterface::default().outside, api.rs:1638-1645)
RSPT_PLAIN_KERNEL void k_vol_init(VolBuf vb, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) vb.medium[i] = 0u;
}

}  // namespace rspt
