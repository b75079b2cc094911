// VolPathIntegrator::li for ONE camera sample on ONE lane (src/integrators/volpath.rs:60-347) — the form the pixel samplers need
// (tile_serial.h: a tile is one serial chain of its PCG stream, so the lane that owns the tile runs every sample from camera ray to
// radiance before it starts the next) — and GridDensityMedium (src/media/grid.rs), whose tr / sample draw a data-dependent number of
// sampler values per call, also in the middle of estimate_direct: only a sampler that is read in program order can follow that.
// The control flow is the reference's, statement by statement; the wavefront form of the same integrator is vol.h.
//
// Sampler draws, in order, per turn of the loop: [Medium::sample: homogeneous 2 x get_1d; grid: 1 or 2 per delta-tracking step] then
// uniform_sample_one_light: get_1d, get_2d, get_2d; VisibilityTester::tr: per segment in a grid medium 1 or 2 per ratio-tracking step;
// the BSDF- / phase-sampled half of estimate_direct: its Scene::intersect_tr walk draws the same way (the half adds nothing: the
// transmittance it multiplies into starts at Spectrum::default() = 0, integrator.rs:531 — but its draws move the stream, so with grid
// media in the scene the walk is made); then the direction sample get_2d and the roulette's get_1d.
#pragma once
#include "vol.h"
#include "trace_serial.h"

namespace rspt {

// ---- GridDensityMedium (grid.rs:57-270).  rspt_medium.pad carries 1 / max(density) (GridDensityMedium::new, :44-55; set by rspt_scene_create) ----
RDEV float grid_sigma_t(const rspt_medium& m) { return m.sigma_s[0] + m.sigma_a[0]; }   // (sigma_s + sigma_a)[Red]
RDEV float grid_inv_max(const rspt_medium& m) { return __uint_as_float(m.pad); }
RDEV float grid_d(const rspt_medium& m, int32_t x, int32_t y, int32_t z) {  // :57-75
    if (!(x >= 0 && x < m.nx && y >= 0 && y < m.ny && z >= 0 && z < m.nz)) return 0.0f;
    return m.density[((size_t)z * (size_t)m.ny + (size_t)y) * (size_t)m.nx + (size_t)x];
}
RDEV float vs_lerp(float t, float a, float b) { return a * (1.0f - t) + b * t; }  // pbrt.rs lerp
RDEVN float grid_density(const rspt_medium& m, f3 p) {  // :76-153: trilinear, sample points at voxel centres
    const f3 ps{p.x * (float)m.nx - 0.5f, p.y * (float)m.ny - 0.5f, p.z * (float)m.nz - 0.5f};
    const int32_t ix = f2i32(floorf(ps.x)), iy = f2i32(floorf(ps.y)), iz = f2i32(floorf(ps.z));
    const f3 d{ps.x - (float)ix, ps.y - (float)iy, ps.z - (float)iz};
    const float d00 = vs_lerp(d.x, grid_d(m, ix, iy, iz), grid_d(m, ix + 1, iy, iz));
    const float d10 = vs_lerp(d.x, grid_d(m, ix, iy + 1, iz), grid_d(m, ix + 1, iy + 1, iz));
    const float d01 = vs_lerp(d.x, grid_d(m, ix, iy, iz + 1), grid_d(m, ix + 1, iy, iz + 1));
    const float d11 = vs_lerp(d.x, grid_d(m, ix, iy + 1, iz + 1), grid_d(m, ix + 1, iy + 1, iz + 1));
    return vs_lerp(d.z, vs_lerp(d.y, d00, d10), vs_lerp(d.y, d01, d11));
}
// Bounds3f::intersect_b (geometry.rs:2183-2210) of the unit cube
RDEV bool unit_cube_intersect_b(f3 o, f3 d, float ray_tmax, float* hitt0, float* hitt1) {
    float t0 = 0.0f, t1 = ray_tmax;
    const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float inv_ray_dir = 1.0f / dd[i];
        float t_near = (0.0f - oo[i]) * inv_ray_dir, t_far = (1.0f - oo[i]) * inv_ray_dir;
        if (t_near > t_far) { const float t = t_near; t_near = t_far; t_far = t; }
        t_far *= 1.0f + 2.0f * gamma_n(3);
        if (t_near > t0) t0 = t_near;
        if (t_far < t1) t1 = t_far;
        if (t0 > t1) return false;
    }
    *hitt0 = t0; *hitt1 = t1;
    return true;
}
// the prelude tr and sample share (:158-176, :216-235): the world ray normalised (t_max scaled by its length), then to medium space
RDEV void grid_ray(const rspt_medium& m, f3 o, f3 d, float t_max, f3* mo, f3* md, float* mt) {
    xf_ray(m.world_to_medium, nullptr, o, normalize(d), t_max * len(d), mo, md, mt);   // (rspt_scene_create refuses a world_to_medium whose row 3 is not (0 0 0 1))
}
template <class S>
RDEVN rgb grid_tr(const rspt_medium& m, f3 o, f3 d, float ray_tmax, S& smp) {  // GridDensityMedium::tr :155-208: ratio tracking
    f3 mo, md; float mt;
    grid_ray(m, o, d, ray_tmax, &mo, &md, &mt);
    float t_min = 0.0f, t_max = 0.0f;
    if (!unit_cube_intersect_b(mo, md, mt, &t_min, &t_max)) return mkrgb(1.0f);
    const float inv_max = grid_inv_max(m), sigma_t = grid_sigma_t(m);
    float tr = 1.0f, t = t_min;
    for (;;) {
        t -= rspt_logf(1.0f - smp.get_1d()) * inv_max / sigma_t;
        if (t >= t_max) break;
        const float density = grid_density(m, mo + md * t);
        tr *= 1.0f - fmaxf(0.0f, density * inv_max);
        const float rr_threshold = 0.1f;  // "added after book publication"
        if (tr < rr_threshold) {
            const float q = fmaxf(0.05f, 1.0f - tr);
            if (smp.get_1d() < q) return mkrgb(0.0f);
            tr /= 1.0f - q;
        }
    }
    return mkrgb(tr);
}
template <class S>
RDEVN rgb grid_sample(const rspt_medium& m, f3 o, f3 d, float ray_tmax, S& smp, bool* sampled, f3* mi_p) {  // ::sample :209-270: delta tracking
    *sampled = false;
    f3 mo, md; float mt;
    grid_ray(m, o, d, ray_tmax, &mo, &md, &mt);
    float t_min = 0.0f, t_max = 0.0f;
    if (!unit_cube_intersect_b(mo, md, mt, &t_min, &t_max)) return mkrgb(1.0f);
    const float inv_max = grid_inv_max(m), sigma_t = grid_sigma_t(m);
    float t = t_min;
    for (;;) {
        t -= rspt_logf(1.0f - smp.get_1d()) * inv_max / sigma_t;
        if (t >= t_max) break;
        if (grid_density(m, mo + md * t) * inv_max > smp.get_1d()) {
            *mi_p = o + d * t;  // r_world.position(t): the world ray AS GIVEN (not normalised) at the normalised ray's parameter (:243)
            *sampled = true;
            return ldrgb(m.sigma_s) / sigma_t;
        }
    }
    return mkrgb(1.0f);
}

// what PixSampler hands out, under the names the code above and below uses
struct SerialSampler {
    PixSampler* px;
    RDEV float get_1d() { return px->get_1d(); }
    RDEV f2 get_2d() { return px->get_2d(); }
};

// a surface hit as VolPathIntegrator::li sees it
struct SerialHit {
    Hit h;
    f3 wo;               // isect.common.wo (estimate_direct reads it; `li` itself passes -ray.d on)
    bool has_primitive;  // false: a transformed instance hit in v0.9.12 behaviour (Q11)
    uint32_t m_in, m_out;  // the interaction's medium interface (0 = None)
};

// ANIM (round 6): the scene has moving instances — every traversal interpolates the instances it enters at `time`, the camera sample's ray time, and so does the hit's interaction
template <bool INST, bool ALPHA, bool ANIM = false>
struct VolSerial {
    const SceneDev& sc; const TexTables& tt; const LightDistDev& ld; const RenderDev& rd; const PathBuf& pb;
    uint32_t slot;
    SerialSampler smp;
    uint32_t* lds;
    uint32_t max_walk;     // cap on passes through BSDF-less surfaces / shadow-ray segments (the reference has none)
    bool truncated;
    float time;            // Ray.time of the camera sample (ANIM)

    RDEV TraceResult closest(f3 o, f3 d, float t_max) { return serial_trace<false, INST, ALPHA, ANIM>(sc, tt, o, d, t_max, lds, time); }
    // Medium::tr over a ray whose t_max is where its traversal left it
    RDEV rgb medium_tr(uint32_t medium, f3 o, f3 d, float t_max) {
        const rspt_medium& m = sc.media[medium - 1u];
        if (m.kind == RSPT_MEDIUM_GRID) return grid_tr(m, o, d, t_max, smp);
        return med_tr(m, t_max, len(d));
    }
    // GeometricPrimitive::intersect's interaction (+ TransformedPrimitive's transform) for a reported hit
    RDEV void surface(const TraceResult& r, f3 ray_d, uint32_t ray_medium, SerialHit* s) {
        const TriRec tri = load_tri(sc, r.prim);
        tri_fill(sc, r.prim, tri, r.b0, r.b1, r.b2, &s->h);
        s->wo = -ray_d;
        s->has_primitive = true;
        const rspt_mesh me = sc.meshes[sc.prims[r.prim].mesh];
        s->m_in = s->m_out = ray_medium;
        if (me.medium_inside != me.medium_outside) { s->m_in = me.medium_inside; s->m_out = me.medium_outside; }
        const bool moving = INST && ANIM && r.inst && sc.inst[r.inst - 1u].anim != RSPT_MISS;
        InstDev moved;
        if (moving) moved = inst_at(sc, r.inst - 1u, time);   // primitive_to_world.interpolate(ray.time) (primitive.rs:218-222)
        if (INST && r.inst && !(moving ? moved : sc.inst[r.inst - 1u]).identity) {  // transform_surface_interaction: no medium interface, (v0.9.12) no primitive
            const InstDev& in = moving ? moved : sc.inst[r.inst - 1u];
            inst_hit(in, &s->h);
            s->wo = normalize(xf_vector(in.m, -xf_vector(in.mi, ray_d)));
            s->m_in = s->m_out = 0u;
            if (!sc.inst_fixed) { s->h.material = 0xffffffffu; s->h.area_light = -1; s->has_primitive = false; }
        }
    }
    // VisibilityTester::tr (light.rs:207-239)
    RDEVN rgb visibility_tr(f3 p0, f3 p0_err, f3 p0_n, uint32_t p0_in, uint32_t p0_out, const LightSample& p1) {
        f3 origin = offset_ray_origin(p0, p0_err, p0_n, p1.p - p0);
        f3 target = offset_ray_origin(p1.p, p1.p_err, p1.n, origin - p1.p);
        f3 d = target - origin;
        uint32_t medium = dot(d, p0_n) > 0.0f ? p0_out : p0_in;
        rgb tr = mkrgb(1.0f);
        for (uint32_t seg = 0;; seg++) {
            if (seg > max_walk) { truncated = true; break; }
            const TraceResult r = closest(origin, d, 1.0f - RSPT_SHADOW_EPS);
            if (r.prim == RSPT_MISS) {
                if (medium) tr = tr * medium_tr(medium, origin, d, r.t_end);
                break;
            }
            SerialHit s;
            surface(r, d, medium, &s);
            if (s.has_primitive) {
                if (s.h.material != 0xffffffffu) return mkrgb(0.0f);
                if (medium) tr = tr * medium_tr(medium, origin, d, r.t);
            }
            origin = offset_ray_origin(s.h.p, s.h.p_err, s.h.n, p1.p - s.h.p);   // isect.common.spawn_ray_to(p1)
            target = offset_ray_origin(p1.p, p1.p_err, p1.n, origin - p1.p);
            d = target - origin;
            medium = dot(d, s.h.n) > 0.0f ? s.m_out : s.m_in;
        }
        return tr;
    }
    // the sampler side of Scene::intersect_tr (scene.rs:79-106) along a ray that starts in `medium`: nothing it computes is used (see the header)
    RDEVN void intersect_tr_draws(f3 o, f3 d, uint32_t medium) {
        for (uint32_t seg = 0;; seg++) {
            if (seg > max_walk) { truncated = true; break; }
            const TraceResult r = closest(o, d, RSPT_INF);
            if (medium) (void)medium_tr(medium, o, d, r.prim == RSPT_MISS ? r.t_end : r.t);
            if (r.prim == RSPT_MISS) break;
            SerialHit s;
            surface(r, d, medium, &s);
            if (s.has_primitive && s.h.material != 0xffffffffu) break;
            o = offset_ray_origin(s.h.p, s.h.p_err, s.h.n, d);   // isect.spawn_ray(&ray.d)
            medium = dot(d, s.h.n) > 0.0f ? s.m_out : s.m_in;
        }
    }
    // uniform_sample_one_light + estimate_direct (integrator.rs:359-570, handle_media) at a surface (bsdf != nullptr) or medium interaction
    RDEVN rgb one_light(f3 p, f3 p_err, f3 n, f3 wo, uint32_t m_in, uint32_t m_out, const Bsdf* bsdf, f3 sh_n, float g) {
        if (sc.n_lights == 0u) return mkrgb(0.0f);
        const uint32_t vox = light_row(ld, light_voxel(sc, ld, p));
        float pdf_choice = 0.0f;
        const uint32_t light_num = sample_discrete(ld.func + (size_t)vox * sc.n_lights, ld.cdf + (size_t)vox * (sc.n_lights + 1), ld.func_int[vox], sc.n_lights, smp.get_1d(), &pdf_choice);
        if (pdf_choice == 0.0f) return mkrgb(0.0f);
        const f2 u_light = smp.get_2d();
        const f2 u_scattering = smp.get_2d();
        const rspt_light lt = sc.lights[light_num];
        const uint32_t nonspec = BX_ALL & ~BX_SPEC;
        rgb l_d = mkrgb(0.0f);
        f3 wi{0.0f, 0.0f, 0.0f};
        float light_pdf = 0.0f, scattering_pdf = 0.0f;
        LightSample ls;
        rgb li = light_sample_li(sc, lt, p, u_light, &wi, &light_pdf, &ls);
        if (light_pdf > 0.0f && !is_black(li)) {
            rgb f;
            if (bsdf) { f = bsdf->f(wo, wi, nonspec) * mkrgb(absdot(wi, sh_n)); scattering_pdf = bsdf->pdf(wo, wi, nonspec); }
            else { const float ph = phase_hg(dot(wo, wi), g); f = mkrgb(ph); scattering_pdf = ph; }
            if (!is_black(f)) {
                li = li * visibility_tr(p, p_err, n, m_in, m_out, ls);
                if (!is_black(li)) {
                    if (light_is_delta(lt)) l_d = l_d + f * li / light_pdf;
                    else l_d = l_d + f * li * mkrgb(power_heuristic(light_pdf, scattering_pdf)) / light_pdf;
                }
            }
        }
        if (sc.n_grid_media && !light_is_delta(lt)) {  // the second half, for its draws only
            rgb f;
            if (bsdf) {
                uint32_t sampled_type = 0;
                f = bsdf->sample_f(wo, &wi, u_scattering, &scattering_pdf, nonspec, &sampled_type);
                f = f * mkrgb(absdot(wi, sh_n));
            } else {
                const float ph = hg_sample_p(g, wo, &wi, u_scattering);
                f = mkrgb(ph); scattering_pdf = ph;
            }
            if (!is_black(f) && scattering_pdf > 0.0f) {
                const f3 ro = offset_ray_origin(p, p_err, n, wi);   // it.spawn_ray(&wi)
                float lpdf = 0.0f;
                if (lt.kind == RSPT_LIGHT_INFINITE) lpdf = infinite_pdf_li(sc, lt, wi);
                else {  // Triangle::pdf_with_ref_point (triangle.rs:745-764)
                    const TriRec lt_tri = load_tri(sc, lt.prim);
                    float t_l, lb0, lb1, lb2;
                    if (tri_test(lt_tri.p0, lt_tri.p1, lt_tri.p2, ro, ray_shear(wi), RSPT_INF, &t_l, &lb0, &lb1, &lb2)) {
                        Hit lh;
                        tri_fill(sc, lt.prim, lt_tri, lb0, lb1, lb2, &lh);
                        lpdf = dist2(p, lh.p) / (absdot(lh.n, -wi) * tri_area(lt_tri));
                        if (__builtin_isinf(lpdf)) lpdf = 0.0f;
                    }
                }
                if (lpdf != 0.0f) intersect_tr_draws(ro, wi, dot(wi, n) > 0.0f ? m_out : m_in);
            }
        }
        return l_d / pdf_choice;
    }

    // VolPathIntegrator::li; p_film / p_lens: the camera sample (the texture stage rebuilds the camera ray's differentials from them)
    RDEVN rgb li(f3 ray_o, f3 ray_d, float ray_tmax0, f2 p_film, f3 p_lens) {
        rgb L = mkrgb(0.0f), beta = mkrgb(1.0f);
        uint32_t medium = 0u;   // camera rays start outside every medium (api.rs:1638-1645)
        bool specular = false, no_diff = false;
        uint32_t bounces = 0u;
        float eta_scale = 1.0f, t_max = ray_tmax0;
        for (uint32_t turn = 0;; turn++) {
            if (turn > max_walk + rd.max_depth + 2u) { truncated = true; break; }
            const TraceResult r = closest(ray_o, ray_d, t_max);
            t_max = RSPT_INF;   // every later ray is a spawn_ray
            const bool found = r.prim != RSPT_MISS;
            const float seg_tmax = found ? r.t : r.t_end;   // ray.t_max after Scene::intersect (an instanced hit: the object ray's parameter; a miss: Q10)
            // ---- medium.sample(&ray, sampler) (:96-101 / :289-294) ----
            bool have_mi = false;
            f3 mi_p{0.0f, 0.0f, 0.0f};
            float g = 0.0f;
            if (medium) {
                const rspt_medium& m = sc.media[medium - 1u];
                g = m.g;
                if (m.kind == RSPT_MEDIUM_GRID) beta = beta * grid_sample(m, ray_o, ray_d, seg_tmax, smp, &have_mi, &mi_p);
                else {  // HomogeneousMedium::sample (homogeneous.rs:37-91)
                    const rgb sigma_t = med_sigma_t(m);
                    uint32_t channel = (uint32_t)(smp.get_1d() * 3.0f);
                    channel = channel < 2u ? channel : 2u;
                    const float dist = -rspt_logf(1.0f - smp.get_1d()) / (channel == 0u ? sigma_t.r : (channel == 1u ? sigma_t.g : sigma_t.b));
                    const float dlen = len(ray_d);
                    const float t = fminf(dist / dlen, seg_tmax);
                    have_mi = t < seg_tmax;
                    const rgb tr = rgb_exp(rgb{-sigma_t.r, -sigma_t.g, -sigma_t.b} * fminf(t, 3.402823466e+38f) * dlen);
                    const rgb density = have_mi ? sigma_t * tr : tr;
                    float pdf = 0.0f;
                    pdf += density.r; pdf += density.g; pdf += density.b;
                    pdf *= 1.0f / 3.0f;
                    if (pdf == 0.0f) pdf = 1.0f;
                    beta = beta * (have_mi ? tr * ldrgb(m.sigma_s) / pdf : tr / pdf);
                    mi_p = ray_o + ray_d * t;
                }
            }
            if (is_black(beta)) break;
            if (have_mi) {  // scattering at a point in the medium (:101-127 / :304-330)
                if (bounces >= rd.max_depth) break;
                const f3 wo = -ray_d;
                L = L + beta * one_light(mi_p, f3{0.0f, 0.0f, 0.0f}, f3{0.0f, 0.0f, 0.0f}, wo, medium, medium, nullptr, f3{0.0f, 0.0f, 0.0f}, g);
                f3 wi{0.0f, 0.0f, 0.0f};
                (void)hg_sample_p(g, wo, &wi, smp.get_2d());
                ray_o = mi_p; ray_d = wi;   // mi.spawn_ray(&wi): n = 0 and p_error = 0 leave the origin where it is; the medium stays
                specular = false;
                if (!found) {  // :332-339 with the SCATTERED ray, then the path ends
                    if (bounces == 0u)
                        for (uint32_t k = 0; k < sc.n_infinite; k++) L = L + beta * infinite_le(sc, sc.lights[sc.infinite_lights[k]], ray_d);
                    break;
                }
            } else if (found) {
                SerialHit s;
                surface(r, ray_d, medium, &s);
                const f3 wo_ray = -ray_d;
                if (bounces == 0u || specular) {  // :133-136
                    const rgb e = s.h.area_light >= 0 ? light_l(sc.lights[s.h.area_light], s.h.n, wo_ray) : mkrgb(0.0f);
                    L = L + beta * e;
                }
                if (bounces >= rd.max_depth) break;
                if (s.h.material == 0xffffffffu) {  // no BSDF: isect.spawn_ray(&ray.d); `continue` skips the bounce count and the roulette (:141-145)
                    ray_o = offset_ray_origin(s.h.p, s.h.p_err, s.h.n, ray_d);
                    medium = dot(ray_d, s.h.n) > 0.0f ? s.m_out : s.m_in;
                    no_diff = true;
                    continue;
                }
                const rspt_material mat = sc.materials[s.h.material];
                Bsdf b;  // Bsdf::new (reflection.rs:235-245)
                b.eta = mat.eta; b.lt = LobeTex{nullptr, 0}; b.dropped = 0u;
                const rspt_bxdf* dyn_l = nullptr;
                uint32_t dyn_n = 0u;
                if (sc.mat_flags && sc.mat_flags[s.h.material]) {  // textured material: the texture stage, called for this hit (texture_path reads the path slot)
                    store_ray(pb.ray_cont + slot, ray_o, ray_d, RSPT_INF, slot);
                    pb.hit_cont[slot] = make_float4(__uint_as_float(r.prim), r.b0, r.b1, r.b2);
                    if (INST && pb.hit_inst) pb.hit_inst[slot] = r.inst;
                    pb.state[slot] = ST_ALIVE | ((bounces & 0xffu) << ST_BOUNCE_SHIFT) | (no_diff ? ST_NO_DIFF : 0u);
                    pb.p_film[slot] = make_float2(p_film.x, p_film.y);
                    texture_path(sc, tt, rd, pb, slot, &p_lens);
                    const float4* tb = pb.tex + slot;
                    if (sc.mat_flags[s.h.material] & RSPT_MAT_DYNAMIC) {   // the lobe list itself is built from the hit's texture values (material_assembly.h)
                        const rspt_mat::Built* bl = dynamic_lobes(sc.dyn[s.h.material], tb, pb.tex_stride, true /* volpath.rs:146 */, pb.dyn_built + (blockIdx.x * blockDim.x + threadIdx.x));
                        dyn_l = bl->l; dyn_n = bl->n;
                        b.eta = bl->eta;
                    }
                    b.lt = LobeTex{tb, pb.tex_stride};
                    const float4 m4 = tb[4 * (size_t)pb.tex_stride];
                    const uint32_t tf = __float_as_uint(m4.w);
                    b.dropped = (tf >> 8) & 0xffu;
                    if (tf & 1u) {
                        const float4 d4 = tb[5 * (size_t)pb.tex_stride];
                        s.h.sh_n = f3{m4.x, m4.y, m4.z};
                        s.h.sh_dpdu = f3{d4.x, d4.y, d4.z};
                    }
                }
                b.ss = normalize(s.h.sh_dpdu); b.ns = s.h.sh_n; b.ng = s.h.n; b.ts = cross(s.h.sh_n, b.ss);
                b.lobes = dyn_l ? dyn_l : sc.bxdfs + mat.first_bxdf;
                b.n = dyn_l ? (dyn_n < 8u ? dyn_n : 8u) : (mat.n_bxdfs < 8u ? mat.n_bxdfs : 8u);
                L = L + beta * one_light(s.h.p, s.h.p_err, s.h.n, s.wo, s.m_in, s.m_out, &b, s.h.sh_n, 0.0f);   // no non-specular-lobe test in front (:146-161)
                f3 wi{0.0f, 0.0f, 0.0f};
                float pdf = 0.0f;
                uint32_t sampled_type = 255;
                const rgb f = b.sample_f(wo_ray, &wi, smp.get_2d(), &pdf, BX_ALL, &sampled_type);
                if (is_black(f) || pdf == 0.0f) break;
                beta = beta * ((f * absdot(wi, s.h.sh_n)) / pdf);
                specular = (sampled_type & BX_SPEC) != 0;
                if ((sampled_type & BX_SPEC) && (sampled_type & BX_TRANS)) {
                    const float eta = b.eta;
                    if (dot(wo_ray, s.h.n) > 0.0f) eta_scale *= eta * eta;
                    else eta_scale *= 1.0f / (eta * eta);
                }
                ray_o = offset_ray_origin(s.h.p, s.h.p_err, s.h.n, wi);
                ray_d = wi;
                medium = dot(wi, s.h.n) > 0.0f ? s.m_out : s.m_in;
            } else {  // escaped without scattering (:332-339)
                if (bounces == 0u || specular)
                    for (uint32_t k = 0; k < sc.n_infinite; k++) L = L + beta * infinite_le(sc, sc.lights[sc.infinite_lights[k]], ray_d);
                break;
            }
            // Russian roulette (:275-285): inside the found-intersection branch, also after scattering in the medium
            const rgb rr = beta * eta_scale;
            if (maxc(rr) < rd.rr_threshold && bounces > 3u) {
                const float q = fmaxf(0.05f, 1.0f - maxc(rr));
                if (smp.get_1d() < q) break;
                beta = beta / (1.0f - q);
            }
            bounces += 1u;
        }
        return L;
    }
};

}  // namespace rspt
