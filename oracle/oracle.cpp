// TEST INFRASTRUCTURE — C entry points of the CPU oracle (liboracle.so), loaded with ctypes
// by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
// Parity: pinned by the reference's own output on ONE scene family — the two Cornell-box renders of its documentation (path + Sobol' +
// matte + area light + spatial light distribution + box filter: tests/test_reference_pin.py, 94 % of the 8-spp PNG's pixels byte for
// byte).  No other OUTPUT of the reference exists (the rest is held to its TEXT, below):
// rs_pbrt ships no tests / golden vectors and cannot be built here (no Rust toolchain), so those parts rest on first-principles
// known-answer tests (tests/test_oracle_*.py) until a dump of real rs_pbrt is committed (oracle/REFERENCE_FIXTURES.md).
// Pinned by the reference's own TEXT since round 6 (compiled from the Rust sources by committed rewrite rules, oracle/make_leaf_fixtures.py; bit for bit):
// fr_dielectric, fr_conductor, trowbridge_reitz_sample_11 / _sample, sobol_sample_float, concentric_sample_disk, Matrix4x4::inverse; since round 5
// AnimatedTransform's derivative polynomials (oracle/make_motion_fixture.py); third session of round 6 (oracle/make_geom_fixtures.py): Bounds3f::intersect_p, the
// watertight test of Triangle::intersect / intersect_p, pnt3_offset_ray_origin (+ next_float_up / _down, gamma), vec3_cross_vec3, vec3_coordinate_system, reflect,
// refract, power_heuristic, cosine_ / uniform_sample_hemisphere, TrowbridgeReitzDistribution::{roughness_to_alpha, d, lambda, g1, g, pdf}, phase_hg, RGBSpectrum::y,
// Rng::{set_sequence, uniform_uint32, uniform_uint32_bounded, uniform_float}.  Later sessions of round 6 (oracle/make_geom_fixtures.py, make_flow_fixtures.py; DESIGN.md §3a holds
// the whole list, ~420 functions, blocks and tables): the whole Triangle::intersect, BVHAccel::intersect / _p and the builder, all six samplers, the camera, light sampling incl. the infinite
// light, the film, all nine lobes and Bsdf's loop, eight material recipes, estimate_direct, PathIntegrator / DirectLighting / AO ::li, SamplerIntegrator::render's tile loop run over
// the text of every stage it calls, the MIP map, the texture mappings and procedural textures, bump mapping, the homogeneous medium, AnimatedTransform::decompose / interpolate,
// TransformedPrimitive::intersect.  Unpinned: MixMaterial's lobe copy, VolPathIntegrator::li's control flow, the grid medium, MipMap::new's resampling.
#include "orc_render.hpp"
#include "orc_motion.hpp"

using namespace orc;

extern "C" {

// ---- leaf known-answer hooks ----
float orc_next_float_up(float v) { return next_float_up(v); }
float orc_next_float_down(float v) { return next_float_down(v); }
float orc_gamma(int n) { return orc::gamma(n); }
float orc_radical_inverse(int base_index, uint64_t a) { return radical_inverse(base_index, a); }
uint64_t orc_sobol_index(const rspt_sampler_tables* t, uint32_t m, uint64_t frame, int32_t px, int32_t py) {
    return sobol_interval_to_index(SobolTables{t->sobol32, t->vdc, t->vdc_inv}, m, frame, px, py);
}
float orc_sobol_sample(const rspt_sampler_tables* t, int64_t index, int dim) {
    return sobol_sample_float(SobolTables{t->sobol32, t->vdc, t->vdc_inv}, index, dim, 0);
}
// RADICAL_INVERSE_PERMUTATIONS for the first n_dims primes; returns the entry count (out may be NULL)
uint64_t orc_halton_permutations(int n_dims, uint16_t* out) {
    std::vector<uint16_t> p = radical_inverse_permutations(n_dims);
    if (out) std::memcpy(out, p.data(), p.size() * sizeof(uint16_t));
    return p.size();
}
uint32_t orc_pcg32_next(uint64_t* state, uint64_t inc) { Rng r; r.state = *state; r.inc = inc; uint32_t v = r.uniform_uint32(); *state = r.state; return v; }
// HaltonSampler::{get_index_for_sample, sample_dimension} for pixel (px, py)
uint64_t orc_halton_index(const rspt_render_desc* rd, int32_t px, int32_t py, uint64_t sample_num) {
    HaltonSampler h(rd->spp, rd->sample_bounds, rd->sample_at_pixel_center != 0, rd->tables.halton_perms, rd->tables.n_halton_perms);
    h.px = px; h.py = py;
    return h.get_index_for_sample(sample_num);
}
float orc_halton_sample(const rspt_render_desc* rd, uint64_t index, int dim) {
    HaltonSampler h(rd->spp, rd->sample_bounds, rd->sample_at_pixel_center != 0, rd->tables.halton_perms, rd->tables.n_halton_perms);
    return h.sample_dimension(index, dim);
}
// camera samples of one pixel sample: out = p_film.xy, time, p_lens.xy
void orc_camera_sample(const rspt_render_desc* rd, int32_t px, int32_t py, int64_t s, float out[5]) {
    Sampler sp(*rd);
    sp.start_pixel(px, py);
    for (int64_t i = 0; i < s; i++) sp.start_next_sample();
    P2 f = sp.get_2d(); out[0] = (float)px + f.x; out[1] = (float)py + f.y;
    out[2] = sp.get_1d();
    P2 l = sp.get_2d(); out[3] = l.x; out[4] = l.y;
}
void orc_camera_ray(const rspt_render_desc* rd, const float cs[5], float out[7]) {
    Ray r = camera_ray(*rd, P2{cs[0], cs[1]}, cs[2], P2{cs[3], cs[4]});
    out[0] = r.o.x; out[1] = r.o.y; out[2] = r.o.z; out[3] = r.d.x; out[4] = r.d.y; out[5] = r.d.z; out[6] = r.t_max;
}
// AnimatedTransform of the camera: the decomposition (t[2][3], r[2][4] xyzw, s[2][16]) and the matrix interpolated at `time`
void orc_camera_matrix(const rspt_render_desc* rd, float time, float m_out[16], float* trs_out) {
    AnimatedTransform at = camera_animation(*rd);
    M44 m = at.interpolate(time);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) m_out[4 * i + j] = m.m[i][j];
    if (trs_out) {
        for (int k = 0; k < 2; k++) {
            trs_out[3 * k] = at.t[k].x; trs_out[3 * k + 1] = at.t[k].y; trs_out[3 * k + 2] = at.t[k].z;
            trs_out[6 + 4 * k] = at.r[k].v.x; trs_out[6 + 4 * k + 1] = at.r[k].v.y; trs_out[6 + 4 * k + 2] = at.r[k].v.z; trs_out[6 + 4 * k + 3] = at.r[k].w;
            for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) trs_out[14 + 16 * k + 4 * i + j] = at.s[k].m[i][j];
        }
    }
}
void orc_offset_ray_origin(const float p[3], const float pe[3], const float n[3], const float w[3], float out[3]) {
    V3 r = offset_ray_origin(V3{p[0], p[1], p[2]}, V3{pe[0], pe[1], pe[2]}, V3{n[0], n[1], n[2]}, V3{w[0], w[1], w[2]});
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_concentric_sample_disk(float ux, float uy, float out[2]) { P2 d = concentric_sample_disk(P2{ux, uy}); out[0] = d.x; out[1] = d.y; }
float orc_fr_dielectric(float c, float ei, float et) { return fr_dielectric(c, ei, et); }
// Distribution1D: returns sampled offset, writes pdf and the cdf (n+1 floats)
int64_t orc_distribution1d(const float* f, int n, float u, float* pdf, float* cdf_out, float* func_int) {
    Distribution1D d(std::vector<Float>(f, f + n));
    if (cdf_out) for (int i = 0; i <= n; i++) cdf_out[i] = d.cdf[i];
    if (func_int) *func_int = d.func_int;
    return (int64_t)d.sample_discrete(u, pdf);
}

// ---- BSDF hooks: evaluate a material in a given shading frame (n = (0,0,1), dpdu = (1,0,0)) ----
static Interaction unit_frame() {
    Interaction it; it.n = V3{0, 0, 1}; it.sh_n = V3{0, 0, 1}; it.sh_dpdu = V3{1, 0, 0}; it.sh_dpdv = V3{0, 1, 0};
    it.p = V3{0, 0, 0}; it.p_error = V3{0, 0, 0};
    return it;
}
// surf (may be NULL = the unit frame at the origin): p[3], uv[2], dudx, dvdx, dudy, dvdy, dpdx[3], dpdy[3]
static Interaction surf_interaction(const float* surf);
void orc_bsdf_f(const rspt_scene_desc* d, uint32_t material, int allow_multiple_lobes, const float wo[3], const float wi[3], uint32_t flags, float f_out[3], float* pdf_out) {
    Interaction it = unit_frame();
    Scene sc{*d};
    Bsdf b;
    make_bsdf(sc, it, material, allow_multiple_lobes != 0, &b);
    Spec f = b.f(V3{wo[0], wo[1], wo[2]}, V3{wi[0], wi[1], wi[2]}, (uint8_t)flags);
    f_out[0] = f.c[0]; f_out[1] = f.c[1]; f_out[2] = f.c[2];
    *pdf_out = b.pdf(V3{wo[0], wo[1], wo[2]}, V3{wi[0], wi[1], wi[2]}, (uint8_t)flags);
}
void orc_bsdf_sample_f(const rspt_scene_desc* d, uint32_t material, int allow_multiple_lobes, const float wo[3], float ux, float uy, uint32_t flags,
                       float f_out[3], float wi_out[3], float* pdf_out, uint32_t* sampled_type) {
    Interaction it = unit_frame();
    Scene sc{*d};
    Bsdf b;
    make_bsdf(sc, it, material, allow_multiple_lobes != 0, &b);
    V3 wi{0, 0, 0}; Float pdf = 0; uint8_t st = 255;
    Spec f = b.sample_f(V3{wo[0], wo[1], wo[2]}, &wi, P2{ux, uy}, &pdf, (uint8_t)flags, &st);
    f_out[0] = f.c[0]; f_out[1] = f.c[1]; f_out[2] = f.c[2];
    wi_out[0] = wi.x; wi_out[1] = wi.y; wi_out[2] = wi.z; *pdf_out = pdf; *sampled_type = st;
}
// Material::compute_scattering_functions at one surface point: Bsdf.eta and Bsdf.bxdfs (values only: every texture evaluated).
// Returns the number of BxDFs; out_bxdfs has room for 8.
int orc_material_lobes(const rspt_scene_desc* d, uint32_t material, int allow_multiple_lobes, const float* surf, float* eta_out, rspt_bxdf* out_bxdfs) {
    Interaction it = surf ? surf_interaction(surf) : unit_frame();
    Scene sc{*d};
    Bsdf b;
    make_bsdf(sc, it, material, allow_multiple_lobes != 0, &b);
    *eta_out = b.eta;
    for (int i = 0; i < b.n; i++) out_bxdfs[i] = *b.lobes[i].b;
    return b.n;
}

// ---- texture hooks ----
static Interaction surf_interaction(const float* surf) {
    Interaction it = unit_frame();
    it.p = V3{surf[0], surf[1], surf[2]}; it.uv = P2{surf[3], surf[4]};
    it.dudx = surf[5]; it.dvdx = surf[6]; it.dudy = surf[7]; it.dvdy = surf[8];
    it.dpdx = V3{surf[9], surf[10], surf[11]}; it.dpdy = V3{surf[12], surf[13], surf[14]};
    return it;
}
void orc_tex_eval(const rspt_scene_desc* d, uint32_t ti, const float* surf, float out[3]) {
    Scene sc{*d};
    Spec v = tex_eval(sc, ti, surf_interaction(surf));
    out[0] = v.c[0]; out[1] = v.c[1]; out[2] = v.c[2];
}
// camera ray with its (scaled) differentials: out = o[3], d[3], rx_o[3], rx_d[3], ry_o[3], ry_d[3]
// AnimatedTransform::interpolate (transform.rs:2081-2113) of a moving TransformedPrimitive at `time`: m and m_inv of the resulting Transform
void orc_interpolate_transform(const float start_m[16], const float start_inv[16], float t0, const float end_m[16], const float end_inv[16], float t1, float time,
                               float m_out[16], float inv_out[16]) {
    const AnimatedTransform a(start_m, t0, end_m, t1);
    M44 m, mi;
    a.interpolate_full(time, m44_from(start_inv), m44_from(end_inv), &m, &mi);
    std::memcpy(m_out, &m.m[0][0], 64); std::memcpy(inv_out, &mi.m[0][0], 64);
}
// AnimatedTransform::motion_bounds (transform.rs:2147-2210) of box [lo, hi] under the keys (start_m, t0) / (end_m, t1).
// terms_in: NULL = this oracle's own coefficients, else 60 floats c[5][3][4] (kc, kx, ky, kz) to use instead (the fixture generator passes the
// reference's literal expressions); terms_out (may be NULL): the 60 coefficients + theta that were used.  Returns 0, or 1 where the reference panics.
int orc_motion_bounds(const float start_m[16], float t0, const float end_m[16], float t1, const float lo[3], const float hi[3], const float* terms_in,
                      float out_lo[3], float out_hi[3], float* terms_out, int32_t* flags_out) {
    const AnimatedTransform a(start_m, t0, end_m, t1);
    MotionTerms mt{};
    if (a.actually_animated && a.has_rotation) {
        mt = motion_terms(a);
        if (terms_in) std::memcpy(mt.c, terms_in, sizeof mt.c);
    }
    Bounds3 b, r;
    b.p_min = V3{lo[0], lo[1], lo[2]}; b.p_max = V3{hi[0], hi[1], hi[2]};
    if (!motion_bounds(a, b, &mt, &r)) return 1;
    out_lo[0] = r.p_min.x; out_lo[1] = r.p_min.y; out_lo[2] = r.p_min.z; out_hi[0] = r.p_max.x; out_hi[1] = r.p_max.y; out_hi[2] = r.p_max.z;
    if (terms_out) { std::memcpy(terms_out, mt.c, sizeof mt.c); terms_out[60] = mt.theta; }
    if (flags_out) *flags_out = (a.actually_animated ? 1 : 0) | (a.has_rotation ? 2 : 0);
    return 0;
}
// what AnimatedTransform::new leaves (transform.rs:912-943): t[2][3], r[2][4] (x, y, z, w; second on the shorter arc), s[2][16] — the inputs of the
// reference's derivative-term expressions (the fixture generator feeds them to the machine-converted text)
void orc_animated_keys(const float start_m[16], float t0, const float end_m[16], float t1, float trs_out[46]) {
    const AnimatedTransform a(start_m, t0, end_m, t1);
    for (int k = 0; k < 2; k++) {
        trs_out[3 * k] = a.t[k].x; trs_out[3 * k + 1] = a.t[k].y; trs_out[3 * k + 2] = a.t[k].z;
        trs_out[6 + 4 * k] = a.r[k].v.x; trs_out[6 + 4 * k + 1] = a.r[k].v.y; trs_out[6 + 4 * k + 2] = a.r[k].v.z; trs_out[6 + 4 * k + 3] = a.r[k].w;
        std::memcpy(trs_out + 14 + 16 * k, &a.s[k].m[0][0], 64);
    }
}
void orc_camera_ray_diff(const rspt_render_desc* rd, const float cs[5], float out[18]) {
    Ray r = camera_ray(*rd, P2{cs[0], cs[1]}, cs[2], P2{cs[3], cs[4]});
    r.scale_differentials(1.0f / std::sqrt((Float)rd->spp));
    const V3 v[6] = {r.o, r.d, r.rx_o, r.rx_d, r.ry_o, r.ry_d};
    for (int i = 0; i < 6; i++) { out[3 * i] = v[i].x; out[3 * i + 1] = v[i].y; out[3 * i + 2] = v[i].z; }
}
// differentials of a hit: in = p[3], n[3], dpdu[3], dpdv[3], then the 18 floats of orc_camera_ray_diff; out = dudx, dvdx, dudy, dvdy, dpdx[3], dpdy[3]
void orc_compute_differentials(const float* in, float out[10]) {
    Interaction it = unit_frame();
    it.p = V3{in[0], in[1], in[2]}; it.n = V3{in[3], in[4], in[5]}; it.dpdu = V3{in[6], in[7], in[8]}; it.dpdv = V3{in[9], in[10], in[11]};
    const float* r = in + 12;
    Ray ray{V3{r[0], r[1], r[2]}, V3{r[3], r[4], r[5]}, INF, 0.0f};
    ray.has_diff = true;
    ray.rx_o = V3{r[6], r[7], r[8]}; ray.rx_d = V3{r[9], r[10], r[11]}; ray.ry_o = V3{r[12], r[13], r[14]}; ray.ry_d = V3{r[15], r[16], r[17]};
    compute_differentials(&it, ray);
    out[0] = it.dudx; out[1] = it.dvdx; out[2] = it.dudy; out[3] = it.dvdy;
    out[4] = it.dpdx.x; out[5] = it.dpdx.y; out[6] = it.dpdx.z; out[7] = it.dpdy.x; out[8] = it.dpdy.y; out[9] = it.dpdy.z;
}
// Material::bump in the unit frame (n = +z, dpdu = +x, dpdv = +y): out = shading n[3], dpdu[3]
void orc_bump(const rspt_scene_desc* d, uint32_t ti, const float* surf, float out[6]) {
    Scene sc{*d};
    Interaction it = surf_interaction(surf);
    bump(sc, ti, &it);
    out[0] = it.sh_n.x; out[1] = it.sh_n.y; out[2] = it.sh_n.z; out[3] = it.sh_dpdu.x; out[4] = it.sh_dpdu.y; out[5] = it.sh_dpdu.z;
}

// ---- BVH build (BVHAccel::new over triangles given by global vertex indices) ----
// tri_idx: n*3 vertex indices; ordered_out: n entries (input index of the primitive at each
// BVH-ordered slot).  Returns node count, or -(needed) if nodes_cap is too small.
int64_t orc_bvh_build(const float* P, const uint32_t* tri_idx, uint64_t n, uint32_t max_prims_in_node,
                      rspt_bvh_node* nodes_out, uint64_t nodes_cap, uint32_t* ordered_out) {
    std::vector<Bounds3> bounds(n);
    for (uint64_t i = 0; i < n; i++) { // Triangle::world_bound triangle.rs:126-133
        const uint32_t* v = tri_idx + 3 * i;
        V3 p0{P[3 * v[0]], P[3 * v[0] + 1], P[3 * v[0] + 2]}, p1{P[3 * v[1]], P[3 * v[1] + 1], P[3 * v[1] + 2]}, p2{P[3 * v[2]], P[3 * v[2] + 1], P[3 * v[2] + 2]};
        bounds[i] = bunion(bounds_from(p0, p1), p2);
    }
    std::vector<rspt_bvh_node> nodes; std::vector<uint32_t> ordered;
    bvh_build(bounds.data(), n, max_prims_in_node, nodes, ordered);
    if (nodes.size() > nodes_cap) return -(int64_t)nodes.size();
    std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(rspt_bvh_node));
    std::memcpy(ordered_out, ordered.data(), ordered.size() * sizeof(uint32_t));
    return (int64_t)nodes.size();
}

// the same over primitives given by their world bounds (n x (min xyz, max xyz)): a top-level aggregate that holds TransformedPrimitives
int64_t orc_bvh_build_bounds(const float* b6, uint64_t n, uint32_t max_prims_in_node, rspt_bvh_node* nodes_out, uint64_t nodes_cap, uint32_t* ordered_out) {
    std::vector<Bounds3> bounds(n);
    for (uint64_t i = 0; i < n; i++) {
        bounds[i].p_min = V3{b6[6 * i], b6[6 * i + 1], b6[6 * i + 2]};
        bounds[i].p_max = V3{b6[6 * i + 3], b6[6 * i + 4], b6[6 * i + 5]};
    }
    std::vector<rspt_bvh_node> nodes; std::vector<uint32_t> ordered;
    bvh_build(bounds.data(), n, max_prims_in_node, nodes, ordered);
    if (nodes.size() > nodes_cap) return -(int64_t)nodes.size();
    std::memcpy(nodes_out, nodes.data(), nodes.size() * sizeof(rspt_bvh_node));
    std::memcpy(ordered_out, ordered.data(), ordered.size() * sizeof(uint32_t));
    return (int64_t)nodes.size();
}
// Transform::transform_bounds (transform.rs:596-660): out = min xyz, max xyz
void orc_transform_bounds(const float m[16], const float lo[3], const float hi[3], float out[6]) {
    Bounds3 b;
    bool first = true;
    const int order[8][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 1, 1}, {1, 1, 0}, {1, 0, 1}, {1, 1, 1}};
    for (auto& c : order) {
        V3 p = transform_point(m, V3{c[0] ? hi[0] : lo[0], c[1] ? hi[1] : lo[1], c[2] ? hi[2] : lo[2]});
        if (first) { b.p_min = b.p_max = p; first = false; }
        else b = bunion(b, p);
    }
    out[0] = b.p_min.x; out[1] = b.p_min.y; out[2] = b.p_min.z; out[3] = b.p_max.x; out[4] = b.p_max.y; out[5] = b.p_max.z;
}

// ---- stage hook: Scene::intersect / intersect_p over a batch ----
// counters_out (optional): nodes_visited, tris_tested
void orc_trace(const rspt_scene_desc* sd, const rspt_ray* rays, uint64_t n, rspt_hit* out, int any_hit, int brute, uint64_t* counters_out) {
    Scene sc{*sd};
    Counters c;
    for (uint64_t i = 0; i < n; i++) {
        Ray r{V3{rays[i].o[0], rays[i].o[1], rays[i].o[2]}, V3{rays[i].d[0], rays[i].d[1], rays[i].d[2]}, rays[i].t_max, 0.0f};
        rspt_hit h; h.prim = 0xffffffffu; h.t = 0; h.b0 = h.b1 = h.b2 = 0;
        if (any_hit) { if (sc.intersect_p(r, &c)) h.prim = 0; }
        else if (brute) { uint32_t p = 0xffffffffu; Float t = 0.0f; if (sc.intersect_brute(r, &p, &t)) { h.prim = p; h.t = t; } }
        else {
            Interaction isect; Float t = 0, b[3] = {0, 0, 0};
            if (sc.intersect(r, &isect, &c, &t, b)) { h.prim = (uint32_t)isect.geo_prim; h.t = t; h.b0 = b[0]; h.b1 = b[1]; h.b2 = b[2]; }
        }
        out[i] = h;
    }
    if (counters_out) { counters_out[0] = c.nodes_visited; counters_out[1] = c.tris_tested; }
}

// debugging aid: follow a ray through surfaces without a BSDF the way PathIntegrator::li does (path.rs:109-116: ray = isect.spawn_ray(&ray.d))
// for at most n steps; out[i] = (prim, t, origin xyz of the step's ray); returns the number of hits
int orc_null_walk(const rspt_scene_desc* sd, const rspt_ray* r0, int n, rspt_hit* out) {
    Scene sc{*sd};
    Ray r{V3{r0->o[0], r0->o[1], r0->o[2]}, V3{r0->d[0], r0->d[1], r0->d[2]}, r0->t_max, 0.0f};
    int k = 0;
    for (; k < n; k++) {
        Interaction isect; Float t = 0, b[3] = {0, 0, 0};
        if (!sc.intersect(r, &isect, nullptr, &t, b)) break;
        out[k].prim = (uint32_t)isect.geo_prim; out[k].t = t; out[k].b0 = r.o.x; out[k].b1 = r.o.y; out[k].b2 = r.o.z;
        r = isect.spawn_ray(r.d);
    }
    return k;
}

// ---- the whole path: SamplerIntegrator::render ----
// counters_out[8]: nodes_visited, tris_tested, rays_closest, rays_any, bounces, samples, nan_samples, mis_rays
int orc_render(const rspt_scene_desc* sd, const rspt_render_desc* rd, int num_threads, float* film_xyzw, float* li_rgb,
               uint64_t* counters_out, double* seconds_out) {
    if (!sd || !rd) return -1;
    Scene sc{*sd};
    RenderOut out;
    render(sc, *rd, num_threads, film_xyzw, li_rgb, &out);
    if (counters_out) {
        const Counters& c = out.counters;
        uint64_t v[8] = {c.nodes_visited, c.tris_tested, c.rays_closest, c.rays_any, c.bounces, c.samples, c.nan_samples, c.mis_rays};
        std::memcpy(counters_out, v, sizeof v);
    }
    if (seconds_out) *seconds_out = out.seconds;
    return 0;
}

// DirectLightingIntegrator (kind 2; strategy 0 = UniformSampleAll, 1 = UniformSampleOne) and WhittedIntegrator (kind 3)
// through the same tile loop; n_light_samples: one entry per light or NULL (all 1).  No GPU counterpart yet.
int orc_render_integrator(const rspt_scene_desc* sd, const rspt_render_desc* rd, int num_threads, float* film_xyzw, float* li_rgb,
                          uint64_t* counters_out, int kind, int strategy, const int32_t* n_light_samples) {
    if (!sd || !rd || (kind != ORC_INTEGRATOR_DIRECT && kind != ORC_INTEGRATOR_WHITTED)) return -1;
    Scene sc{*sd};
    RenderOut out;
    render(sc, *rd, num_threads, film_xyzw, li_rgb, &out, kind, strategy, n_light_samples);
    if (counters_out) {
        const Counters& c = out.counters;
        uint64_t v[8] = {c.nodes_visited, c.tris_tested, c.rays_closest, c.rays_any, c.bounces, c.samples, c.nan_samples, c.mis_rays};
        std::memcpy(counters_out, v, sizeof v);
    }
    return 0;
}

void orc_pixel_sampler_arrays(const rspt_render_desc* rd, uint64_t seed, int n_pixels, float* out1d, float* out2d, float* draws, const int32_t* array_sizes, int n_arrays,
                              float* out_arrays);
// the pixel samplers: reseed(seed), then n_pixels x start_pixel (+ a full round of start_next_sample, as a rendered pixel has);
// out1d [dims][spp], out2d [dims][spp][2] = the vectors of the last pixel; draws[0..3] = get_1d, get_2d.x, get_2d.y, get_1d taken
// after every precomputed dimension of sample 0 has been handed out (the on-demand stream)
void orc_pixel_sampler(const rspt_render_desc* rd, uint64_t seed, int n_pixels, float* out1d, float* out2d, float* draws) {
    orc_pixel_sampler_arrays(rd, seed, n_pixels, out1d, out2d, draws, nullptr, 0, nullptr);
}
// the same with 2-D sample arrays requested first (what ao / directlighting's preprocess does): array_sizes[n_arrays];
// out_arrays = the arrays of the last pixel, concatenated, [size * spp][2] each
void orc_pixel_sampler_arrays(const rspt_render_desc* rd, uint64_t seed, int n_pixels, float* out1d, float* out2d, float* draws, const int32_t* array_sizes, int n_arrays,
                              float* out_arrays) {
    Sampler s(*rd);
    for (int i = 0; i < n_arrays; i++) s.request_2d_array(array_sizes[i]);
    s.reseed(seed);
    for (int k = 0; k < n_pixels; k++) {
        s.start_pixel(0, 0);
        if (k + 1 < n_pixels) while (s.start_next_sample()) {}
    }
    const PixelSampler& p = s.pix;
    for (size_t d = 0; d < p.samples_1d.size(); d++)
        for (int64_t i = 0; i < p.spp; i++) out1d[d * (size_t)p.spp + (size_t)i] = p.samples_1d[d][(size_t)i];
    for (size_t d = 0; d < p.samples_2d.size(); d++)
        for (int64_t i = 0; i < p.spp; i++) { out2d[(d * (size_t)p.spp + (size_t)i) * 2] = p.samples_2d[d][(size_t)i].x; out2d[(d * (size_t)p.spp + (size_t)i) * 2 + 1] = p.samples_2d[d][(size_t)i].y; }
    for (size_t d = 0; d < p.samples_1d.size(); d++) (void)s.get_1d();
    for (size_t d = 0; d < p.samples_2d.size(); d++) (void)s.get_2d();
    draws[0] = s.get_1d();
    P2 v = s.get_2d();
    draws[1] = v.x; draws[2] = v.y;
    draws[3] = s.get_1d();
    size_t k = 0;
    for (int i = 0; i < n_arrays; i++)
        for (const P2& q : p.sample_array_2d[(size_t)i]) { out_arrays[k++] = q.x; out_arrays[k++] = q.y; }
}
int32_t orc_round_count(const rspt_render_desc* rd, int32_t n) { Sampler s(*rd); return s.round_count(n); }

// f32::sin / cos / ln / log2 / exp / acos / atan2 = the host libm's functions, over an array (to compare the device's restatements with, bit for bit)
void orc_libm(uint32_t fn, const float* x, const float* y, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        const float v = x[i];
        out[i] = fn == 0 ? std::sin(v) : fn == 1 ? std::cos(v) : fn == 2 ? std::log(v) : fn == 3 ? std::log2(v) : fn == 4 ? std::exp(v) : fn == 5 ? std::acos(v) : std::atan2(v, y[i]);
    }
}

// media (VolPathIntegrator): leaf functions for known-answer tests
float orc_phase_hg(float cos_theta, float g) { return phase_hg(cos_theta, g); }
float orc_hg_sample_p(float g, const float wo[3], float ux, float uy, float wi_out[3]) {
    V3 wi{0, 0, 0};
    float p = hg_sample_p(g, V3{wo[0], wo[1], wo[2]}, &wi, P2{ux, uy});
    wi_out[0] = wi.x; wi_out[1] = wi.y; wi_out[2] = wi.z;
    return p;
}
// HomogeneousMedium::sample for the ray (o, d, t_max): out = (beta factor rgb, sampled 0/1, p xyz)
void orc_homogeneous_sample(const rspt_medium* m, const float o[3], const float d[3], float t_max, float u_channel, float u_dist, float out[7]) {
    Ray ray{V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, t_max, 0.0f};
    ray.medium = 1;
    Interaction mi = Interaction{}; mi.p = V3{0, 0, 0}; bool sampled = false;
    Spec b = homogeneous_sample(*m, 1, ray, u_channel, u_dist, &mi, &sampled);
    out[0] = b.c[0]; out[1] = b.c[1]; out[2] = b.c[2]; out[3] = sampled ? 1.0f : 0.0f;
    out[4] = mi.p.x; out[5] = mi.p.y; out[6] = mi.p.z;
}
// GridDensityMedium leaf functions (orc_render.hpp; not part of the render path yet).  grid = (sigma_a[3], sigma_s[3], g, n[3], world_to_medium[16], density);
// `u` is the stream sampler.get_1d() would deliver; *used = how many values were drawn; returns -1 if the stream ran out.
void orc_grid_density(const int32_t n[3], const float* density, const float* pts, uint64_t npts, float* out) {
    const float zero[3] = {0, 0, 0}, ident[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    GridMedium m = grid_medium_new(zero, zero, 0.0f, n[0], n[1], n[2], ident, density);
    for (uint64_t i = 0; i < npts; i++) out[i] = grid_density(m, V3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]});
}
int orc_grid_tr(const float sigma_a[3], const float sigma_s[3], const int32_t n[3], const float w2m[16], const float* density, const float o[3], const float d[3], float t_max,
                const float* u, uint64_t nu, float out[3], uint64_t* used) {
    GridMedium m = grid_medium_new(sigma_a, sigma_s, 0.0f, n[0], n[1], n[2], w2m, density);
    Ray ray{V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, t_max, 0.0f};
    uint64_t k = 0; bool dry = false;
    Spec tr = grid_tr(m, ray, [&]() -> Float { if (k >= nu) { dry = true; return 0.5f; } return u[k++]; });
    out[0] = tr.c[0]; out[1] = tr.c[1]; out[2] = tr.c[2]; *used = k;
    return dry ? -1 : 0;
}
// out = (beta rgb, sampled 0/1, p xyz, wo xyz)
int orc_grid_sample(const float sigma_a[3], const float sigma_s[3], float g, const int32_t n[3], const float w2m[16], const float* density, const float o[3], const float d[3], float t_max,
                    const float* u, uint64_t nu, float out[10], uint64_t* used) {
    GridMedium m = grid_medium_new(sigma_a, sigma_s, g, n[0], n[1], n[2], w2m, density);
    Ray ray{V3{o[0], o[1], o[2]}, V3{d[0], d[1], d[2]}, t_max, 0.0f};
    uint64_t k = 0; bool dry = false, sampled = false;
    Interaction mi = Interaction{}; mi.p = mi.wo = V3{0, 0, 0};
    Spec b = grid_sample(m, 1, ray, [&]() -> Float { if (k >= nu) { dry = true; return 0.5f; } return u[k++]; }, &mi, &sampled);
    out[0] = b.c[0]; out[1] = b.c[1]; out[2] = b.c[2]; out[3] = sampled ? 1.0f : 0.0f;
    out[4] = mi.p.x; out[5] = mi.p.y; out[6] = mi.p.z; out[7] = mi.wo.x; out[8] = mi.wo.y; out[9] = mi.wo.z; *used = k;
    return dry ? -1 : 0;
}
// VisibilityTester::tr between two free points (MediumInteraction-like ends: n = 0, p_error = 0) that start in medium `medium0`
void orc_visibility_tr(const rspt_scene_desc* sd, const float p0[3], uint32_t medium0, const float p1[3], float tr_out[3]) {
    Scene sc{*sd};
    Interaction a = Interaction{}, b = Interaction{};
    a.p_error = a.n = a.wo = b.p_error = b.n = b.wo = V3{0, 0, 0};
    a.p = V3{p0[0], p0[1], p0[2]}; a.med_in = a.med_out = medium0; a.is_medium = true;
    b.p = V3{p1[0], p1[1], p1[2]};
    sc.prepare_media();
    rspt_render_desc rd0{};            // (homogeneous media draw nothing; a grid medium would draw from this Sobol' sampler's dimension stream)
    rd0.sampler_kind = RSPT_SAMPLER_SOBOL; rd0.spp = 1; rd0.sample_bounds[2] = rd0.sample_bounds[3] = 1;
    Sampler smp(rd0);
    Spec tr = visibility_tr(sc, a, b, smp, nullptr);
    tr_out[0] = tr.c[0]; tr_out[1] = tr.c[1]; tr_out[2] = tr.c[2];
}

// spatial light distribution of one voxel (for differential tests of the device builder):
// writes n_lights func values and n_lights+1 cdf values.
void orc_spatial_voxel(const rspt_scene_desc* sd, const rspt_render_desc* rd, const int32_t pi[3], float* func_out, float* cdf_out, int32_t n_voxels_out[3]) {
    Scene sc{*sd};
    RenderCtx cx; cx.scene = &sc; cx.rd = rd;
    light_distrib_init(cx);
    if (n_voxels_out) for (int i = 0; i < 3; i++) n_voxels_out[i] = cx.strategy == RSPT_LIGHTS_SPATIAL ? cx.n_voxels[i] : 0;
    if (cx.strategy != RSPT_LIGHTS_SPATIAL) return;
    int p[3] = {pi[0], pi[1], pi[2]};
    std::unique_ptr<Distribution1D> d(spatial_compute(cx, p));
    for (size_t i = 0; i < d->func.size(); i++) func_out[i] = d->func[i];
    for (size_t i = 0; i < d->cdf.size(); i++) cdf_out[i] = d->cdf[i];
}

// ---- the oracle's restatements of the leaf functions that tests/golden/leaf_functions.npz pins by the reference's own text (oracle/make_leaf_fixtures.py) ----
// fn: 0 fr_dielectric (c, eta_i, eta_t -> 1)   1 fr_conductor (c, eta_i[3], eta_t[3], k[3] -> 3)   2 TrowbridgeReitz sample_11 (cos, u1, u2 -> 2)
//     3 trowbridge_reitz_sample (wi[3], ax, ay, u1, u2 -> 3)   4 sobol_sample_float (index, dim; words -> 1)   5 concentric_sample_disk (u[2] -> 2)   6 Matrix4x4::inverse (16 -> 16)
void orc_leaf(int fn, const float* a, const float* b, const float* c, const float* d, const float* e, const int64_t* ia, const int32_t* ib, const uint32_t* words, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = fr_dielectric(a[i], b[i], c[i]); break;
        case 1: { Spec r = fr_conductor(a[i], Spec(b[3 * i], b[3 * i + 1], b[3 * i + 2]), Spec(c[3 * i], c[3 * i + 1], c[3 * i + 2]), Spec(d[3 * i], d[3 * i + 1], d[3 * i + 2]));
                  out[3 * i] = r.c[0]; out[3 * i + 1] = r.c[1]; out[3 * i + 2] = r.c[2]; break; }
        case 2: { Float sx = 0, sy = 0; TR::sample_11(a[i], b[i], c[i], &sx, &sy); out[2 * i] = sx; out[2 * i + 1] = sy; break; }
        case 3: { V3 r = TR::sample(V3{a[3 * i], a[3 * i + 1], a[3 * i + 2]}, b[i], c[i], d[i], e[i]); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
        case 4: out[i] = sobol_sample_float(SobolTables{words, nullptr, nullptr}, ia[i], ib[i], 0); break;
        case 5: { P2 r = concentric_sample_disk(P2{a[2 * i], a[2 * i + 1]}); out[2 * i] = r.x; out[2 * i + 1] = r.y; break; }
        case 6: { M44 m; for (int r = 0; r < 4; r++) for (int k = 0; k < 4; k++) m.m[r][k] = a[16 * i + 4 * r + k];
                  const M44 v = m44_inverse(m); for (int r = 0; r < 4; r++) for (int k = 0; k < 4; k++) out[16 * i + 4 * r + k] = v.m[r][k]; break; }
        default: break;
        }
    }
}

// ---- the oracle's restatements of what tests/golden/geom_functions.npz pins by the reference's own text (oracle/make_geom_fixtures.py): same argument layout as
// that script's g_* wrappers ----
void orc_geom_scalar(int fn, const float* a, const float* b, const float* c, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) switch (fn) {
        case 0: out[i] = orc::gamma((int)a[i]); break;
        case 1: out[i] = next_float_up(a[i]); break;
        case 2: out[i] = next_float_down(a[i]); break;
        case 3: out[i] = power_heuristic((int)a[i], b[i], 1, c[i]); break;
        case 5: out[i] = tr_roughness_to_alpha(a[i]); break;
        case 6: out[i] = phase_hg(a[i], b[i]); break;
        case 7: out[i] = Spec(a[i], b[i], c[i]).y(); break;
        default: break;
    }
}
void orc_geom_sample(int fn, const float* u, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        const P2 p{u[2 * i], u[2 * i + 1]};
        const V3 r = fn == 0 ? cosine_sample_hemisphere(p) : uniform_sample_hemisphere(p);
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
void orc_geom_vec(int fn, const float* a, const float* b, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        const V3 x{a[3 * i], a[3 * i + 1], a[3 * i + 2]}, y{b[3 * i], b[3 * i + 1], b[3 * i + 2]};
        switch (fn) {
            case 0: { const V3 r = cross(x, y); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
            case 1: { V3 v2{0, 0, 0}, v3{0, 0, 0}; coordinate_system(x, &v2, &v3); out[6 * i] = v2.x; out[6 * i + 1] = v2.y; out[6 * i + 2] = v2.z;
                      out[6 * i + 3] = v3.x; out[6 * i + 4] = v3.y; out[6 * i + 5] = v3.z; break; }
            case 2: { const V3 r = reflect(x, y); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; break; }
            case 3: { V3 wt{0, 0, 0}; const bool ok = refract(x, y, b[3 * n + i], &wt); out[4 * i] = wt.x; out[4 * i + 1] = wt.y; out[4 * i + 2] = wt.z; out[4 * i + 3] = ok ? 1.0f : 0.0f; break; }
            case 4: out[i] = abs_dot(x, y); break;
            default: break;
        }
    }
}
void orc_geom_offset_ray_origin(const float* p, const float* e, const float* nn, const float* w, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        const V3 r = offset_ray_origin(V3{p[3 * i], p[3 * i + 1], p[3 * i + 2]}, V3{e[3 * i], e[3 * i + 1], e[3 * i + 2]}, V3{nn[3 * i], nn[3 * i + 1], nn[3 * i + 2]},
                                       V3{w[3 * i], w[3 * i + 1], w[3 * i + 2]});
        out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z;
    }
}
void orc_geom_box(const float* box, const float* o, const float* inv, const uint8_t* neg, const float* tmax, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        rspt_bvh_node nd{};
        for (int k = 0; k < 3; k++) { nd.bmin[k] = box[6 * i + k]; nd.bmax[k] = box[6 * i + 3 + k]; }
        Ray r{}; r.o = V3{o[3 * i], o[3 * i + 1], o[3 * i + 2]}; r.d = V3{0, 0, 0}; r.t_max = tmax[i];
        out[i] = Scene::box_hit(nd, r, V3{inv[3 * i], inv[3 * i + 1], inv[3 * i + 2]}, neg + 3 * i) ? 1.0f : 0.0f;
    }
}
void orc_geom_triangle(const float* tri, const float* o, const float* d, const float* tmax, uint64_t n, float* out) {   // the watertight test shared by intersect / intersect_p
    Scene sc{};
    rspt_prim pr{}; pr.v[0] = 0; pr.v[1] = 1; pr.v[2] = 2;
    for (uint64_t i = 0; i < n; i++) {
        sc.d.P = tri + 9 * i;
        Ray r{}; r.o = V3{o[3 * i], o[3 * i + 1], o[3 * i + 2]}; r.d = V3{d[3 * i], d[3 * i + 1], d[3 * i + 2]}; r.t_max = tmax[i];
        Float t = 0.0f, b[3] = {0.0f, 0.0f, 0.0f};
        const bool hit = sc.tri_hit_test(pr, r, &t, b);
        out[5 * i] = hit ? 1.0f : 0.0f; out[5 * i + 1] = hit ? t : 0.0f;
        for (int k = 0; k < 3; k++) out[5 * i + 2 + k] = hit ? b[k] : 0.0f;
    }
}
void orc_geom_microfacet(const float* wo, const float* wh, const float* ax, const float* ay, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        const TR t{ax[i], ay[i]};
        const V3 a{wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]}, h{wh[3 * i], wh[3 * i + 1], wh[3 * i + 2]};
        out[5 * i] = t.d(h); out[5 * i + 1] = t.lambda(a); out[5 * i + 2] = t.g1(a); out[5 * i + 3] = t.g(a, h); out[5 * i + 4] = t.pdf(a, h);
    }
}
void orc_geom_sobol(const rspt_sampler_tables* t, const int64_t* spp, const int32_t* bounds, const int32_t* pixel, uint64_t n, float* out) {   // the tile loop's draws (orc_render.hpp render)
    for (uint64_t i = 0; i < n; i++) {
        SobolSampler s(SobolTables{t->sobol32, t->vdc, t->vdc_inv}, spp[i], bounds + 4 * i);
        const int32_t px = pixel[2 * i], py = pixel[2 * i + 1];
        s.start_pixel(px, py);
        for (int k = 0; k < 4; k++) {
            float* o = out + (4 * i + k) * 26;
            const P2 f2 = s.get_2d();
            o[0] = (Float)px + f2.x; o[1] = (Float)py + f2.y;   // sampler.rs:85-95
            o[2] = s.get_1d();
            const P2 lens = s.get_2d();
            o[3] = lens.x; o[4] = lens.y;
            for (int b = 0; b < 4; b++) {
                o[5 + 5 * b] = s.get_1d();
                const P2 u = s.get_2d(), w = s.get_2d();
                o[6 + 5 * b] = u.x; o[7 + 5 * b] = u.y; o[8 + 5 * b] = w.x; o[9 + 5 * b] = w.y;
            }
            o[25] = s.start_next_sample() ? 1.0f : 0.0f;
        }
    }
}
// the radical inverses and the Halton sampler as the tile loop drives it (same layout as the reference text's g_radical / g_halton, oracle/make_geom_fixtures.py)
void orc_geom_radical(const uint16_t* perms, const uint16_t* bi, const uint64_t* a, uint64_t n, float* out, uint64_t* iout) {
    for (uint64_t i = 0; i < n; i++) {
        out[2 * i] = radical_inverse((int)bi[i], a[i]);
        out[2 * i + 1] = scrambled_radical_inverse((int)bi[i], a[i], perms + prime_tables().sums[bi[i]]);
        iout[4 * i] = reverse_bits_32((uint32_t)a[i]); iout[4 * i + 1] = reverse_bits_64(a[i]);
        iout[4 * i + 2] = inverse_radical_inverse(2, a[i] % 128, 7); iout[4 * i + 3] = inverse_radical_inverse(3, a[i] % 243, 5);
    }
}
void orc_geom_halton(const uint16_t* perms, uint64_t n_perms, const int64_t* spp, const int32_t* bounds, const int32_t* pixel, const uint8_t* center, const int32_t* arrays, uint64_t n, float* out, uint64_t* meta) {
    for (uint64_t i = 0; i < n; i++) {
        rspt_render_desc rd{};
        rd.sampler_kind = RSPT_SAMPLER_HALTON; rd.spp = spp[i]; rd.sample_at_pixel_center = center[i];
        for (int k = 0; k < 4; k++) rd.sample_bounds[k] = bounds[4 * i + k];
        rd.tables.halton_perms = perms; rd.tables.n_halton_perms = n_perms;
        Sampler s(rd);
        for (int k = 0; k < 2; k++) if (arrays[2 * i + k] > 0) s.request_2d_array(arrays[2 * i + k]);
        uint64_t* mt = meta + 12 * i;
        mt[0] = (uint64_t)s.halton.base_scales[0]; mt[1] = (uint64_t)s.halton.base_scales[1]; mt[2] = (uint64_t)s.halton.base_exponents[0]; mt[3] = (uint64_t)s.halton.base_exponents[1];
        mt[4] = s.halton.sample_stride; mt[5] = s.halton.mult_inverse[0]; mt[6] = s.halton.mult_inverse[1]; mt[7] = 0;
        const int32_t px = pixel[2 * i], py = pixel[2 * i + 1];
        s.start_pixel(px, py);
        for (int k = 0; k < 4; k++) {
            float* o = out + (4 * i + k) * 34;
            mt[8 + k] = s.halton.interval_sample_index;
            const P2 f2 = s.get_2d();
            o[0] = (Float)px + f2.x; o[1] = (Float)py + f2.y;   // sampler.rs:85-95
            o[2] = s.get_1d();
            const P2 lens = s.get_2d();
            o[3] = lens.x; o[4] = lens.y;
            for (int b = 0; b < 4; b++) {
                o[5 + 5 * b] = s.get_1d();
                const P2 u = s.get_2d(), w = s.get_2d();
                o[6 + 5 * b] = u.x; o[7 + 5 * b] = u.y; o[8 + 5 * b] = w.x; o[9 + 5 * b] = w.y;
            }
            for (int a = 0; a < 2; a++) {
                o[25 + 4 * a] = o[26 + 4 * a] = o[27 + 4 * a] = o[28 + 4 * a] = -1.0f;
                const int32_t na = arrays[2 * i + a];
                if (na <= 0 || k >= spp[i]) continue;
                size_t idx; uint64_t start;
                if (!s.get_2d_array(na, &idx, &start)) continue;
                const P2 f = s.get_2d_sample(idx, start), l = s.get_2d_sample(idx, start + (uint64_t)na - 1);
                o[25 + 4 * a] = f.x; o[26 + 4 * a] = f.y; o[27 + 4 * a] = l.x; o[28 + 4 * a] = l.y;
            }
            o[33] = s.start_next_sample() ? 1.0f : 0.0f;
        }
    }
}
// the pixel samplers (same layout as the reference text's g_pixel); c_rows: C_MAX_MIN_DIST (17 x 32 words), spp already as MaxMinDistSampler::new leaves it
void orc_geom_pixel(const uint32_t* c_rows, const int32_t* kind, const int64_t* par, const uint64_t* seed, const int32_t* pixel, const int32_t* arrays, uint64_t n, float* out, uint64_t* meta) {
    static const int kinds[4] = {RSPT_SAMPLER_ZEROTWO, RSPT_SAMPLER_MAXMINDIST, RSPT_SAMPLER_STRATIFIED, RSPT_SAMPLER_RANDOM};
    for (uint64_t i = 0; i < n; i++) {
        const int64_t* q = par + 6 * i;
        rspt_render_desc rd{};
        rd.sampler_kind = (uint32_t)kinds[kind[i]]; rd.pixel_dimensions = (uint32_t)q[1];
        rd.strat_x = (uint32_t)q[2]; rd.strat_y = (uint32_t)q[3]; rd.strat_jitter = (uint32_t)q[4];
        int64_t spp = kind[i] == 2 ? q[2] * q[3] : q[0];
        if (kind[i] == 1) {                                   // maxmin.rs:36-59 (the host's part: scenes.py make_sampler does the same): round up to a power of two, pick the matrix
            int64_t r = 1; while (r < spp) r <<= 1;
            spp = r;
            int lg = 0; while ((1ll << lg) < spp) lg++;
            rd.maxmin_c_pixel = c_rows + 32 * lg;
        }
        rd.spp = spp;
        Sampler s(rd);
        for (int k = 0; k < 2; k++) if (arrays[2 * i + k] > 0) s.request_2d_array(arrays[2 * i + k]);
        meta[2 * i] = (uint64_t)spp; meta[2 * i + 1] = (uint64_t)s.round_count(3);
        s.reseed(seed[i]);
        const int32_t px = pixel[2 * i], py = pixel[2 * i + 1];
        s.start_pixel(px, py);
        for (int k = 0; k < 4 && k < spp; k++) {
            float* o = out + (4 * i + k) * 34;
            const P2 f2 = s.get_2d();
            o[0] = (Float)px + f2.x; o[1] = (Float)py + f2.y;   // sampler.rs:85-95
            o[2] = s.get_1d();
            const P2 lens = s.get_2d();
            o[3] = lens.x; o[4] = lens.y;
            for (int b = 0; b < 4; b++) {
                o[5 + 5 * b] = s.get_1d();
                const P2 u = s.get_2d(), w = s.get_2d();
                o[6 + 5 * b] = u.x; o[7 + 5 * b] = u.y; o[8 + 5 * b] = w.x; o[9 + 5 * b] = w.y;
            }
            for (int a = 0; a < 2; a++) {
                o[25 + 4 * a] = o[26 + 4 * a] = o[27 + 4 * a] = o[28 + 4 * a] = -1.0f;
                const int32_t na = arrays[2 * i + a];
                if (na <= 0) continue;
                size_t idx; uint64_t start;
                if (!s.get_2d_array(na, &idx, &start)) continue;
                const P2 f = s.get_2d_sample(idx, start), l = s.get_2d_sample(idx, start + (uint64_t)na - 1);
                o[25 + 4 * a] = f.x; o[26 + 4 * a] = f.y; o[27 + 4 * a] = l.x; o[28 + 4 * a] = l.y;
            }
            o[33] = s.start_next_sample() ? 1.0f : 0.0f;
        }
    }
}
void orc_geom_triangle_full(const float* tri, const float* nrm, const float* tan, const float* uvs, const int32_t* flags, const float* o, const float* d, const float* tmax, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        Scene sc{};
        rspt_prim pr{}; pr.v[0] = 0; pr.v[1] = 1; pr.v[2] = 2; pr.mesh = 0; pr.area_light = -1;
        rspt_mesh m{}; m.has_n = (flags[i] & 1) ? 1u : 0u; m.flip = (flags[i] & 2) ? 1u : 0u; m.has_s = (flags[i] & 4) ? 1u : 0u; m.has_uv = (flags[i] & 8) ? 1u : 0u;
        sc.d.P = tri + 9 * i; sc.d.N = nrm + 9 * i; sc.d.S = tan + 9 * i; sc.d.UV = uvs + 6 * i; sc.d.prims = &pr; sc.d.n_prims = 1; sc.d.meshes = &m; sc.d.n_meshes = 1;
        Ray r{}; r.o = V3{o[3 * i], o[3 * i + 1], o[3 * i + 2]}; r.d = V3{d[3 * i], d[3 * i + 1], d[3 * i + 2]}; r.t_max = tmax[i]; r.time = 0.0f;
        float* q = out + 48 * i;
        for (int k = 0; k < 48; k++) q[k] = 0.0f;
        Float t = 0.0f; Interaction si;
        if (!sc.tri_intersect(pr, r, &t, &si)) continue;
        auto put = [&](int k, V3 v) { q[k] = v.x; q[k + 1] = v.y; q[k + 2] = v.z; };
        q[0] = 1.0f; q[1] = t; put(2, si.p); put(5, si.p_error); put(8, si.wo); put(11, si.n); q[14] = si.uv.x; q[15] = si.uv.y; put(16, si.dpdu); put(19, si.dpdv);
        // q[22 .. 27]: isect.dndu / dndv stay zero in the reference (triangle.rs:322-323, 433-434: the shading block's values shadow them)
        put(28, si.sh_n); put(31, si.sh_dpdu); put(34, si.sh_dpdv); put(37, si.sh_dndu); put(40, si.sh_dndv);
    }
}
// Film::get_film_tile + FilmTile::add_sample + Film::merge_film_tile: per case a 16 x 16 film (crop window = the frame), one tile of samples.  geo: n x 8 ints = tile x0 y0 x1 y1, n_samples, 3 unused;
// flt: n x 260 = filter radius x / y, max_sample_luminance, unused, the 16 x 16 filter table;  smp: n x 64 x 5 = p_film, L;  out: n x 256 x 4
void orc_geom_film(const int32_t* geo, const float* flt, const float* smp, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        rspt_render_desc rd{};
        rd.crop_px[0] = rd.crop_px[1] = 0; rd.crop_px[2] = rd.crop_px[3] = 16;
        rd.filter_radius[0] = flt[260 * i]; rd.filter_radius[1] = flt[260 * i + 1]; rd.max_sample_luminance = flt[260 * i + 2];
        for (int k = 0; k < 256; k++) rd.filter_table[k] = flt[260 * i + 4 + k];
        FilmTile t = get_film_tile(rd, geo + 8 * i);
        for (int k = 0; k < geo[8 * i + 4]; k++) {
            const float* q = smp + (64 * i + k) * 5;
            t.add_sample(rd, P2{q[0], q[1]}, Spec(q[2], q[3], q[4]), 1.0f);
        }
        float* film = out + 1024 * i;
        for (int k = 0; k < 1024; k++) film[k] = 0.0f;
        merge_film_tile(rd, t, film);
    }
}
void orc_geom_differentials(const float* x, uint64_t n, float* out) {
    for (uint64_t i = 0; i < n; i++) {
        const float* q = x + 25 * i;
        Interaction si; si.p = V3{q[0], q[1], q[2]}; si.n = V3{q[3], q[4], q[5]}; si.dpdu = V3{q[6], q[7], q[8]}; si.dpdv = V3{q[9], q[10], q[11]};
        Ray r{}; r.has_diff = q[24] != 0.0f; r.rx_o = V3{q[12], q[13], q[14]}; r.ry_o = V3{q[15], q[16], q[17]}; r.rx_d = V3{q[18], q[19], q[20]}; r.ry_d = V3{q[21], q[22], q[23]};
        compute_differentials(&si, r);
        float* o = out + 10 * i;
        o[0] = si.dudx; o[1] = si.dvdx; o[2] = si.dudy; o[3] = si.dvdy; o[4] = si.dpdx.x; o[5] = si.dpdx.y; o[6] = si.dpdx.z; o[7] = si.dpdy.x; o[8] = si.dpdy.y; o[9] = si.dpdy.z;
    }
}
void orc_geom_morton(const uint32_t* xy, uint64_t n, uint32_t* out) { for (uint64_t i = 0; i < n; i++) out[i] = morton2(xy[2 * i], xy[2 * i + 1]); }
void orc_geom_area_light(const float* tri, const float* nrm, const int32_t* flags, const float* L, const float* ref_p, const float* u, uint64_t n, float* out) {   // light_sample_li on one emitting triangle
    for (uint64_t i = 0; i < n; i++) {
        Scene sc{};
        rspt_prim pr{}; pr.v[0] = 0; pr.v[1] = 1; pr.v[2] = 2; pr.mesh = 0; pr.area_light = 0;
        rspt_mesh m{}; m.has_n = (flags[i] & 1) ? 1u : 0u; m.flip = (flags[i] & 2) ? 1u : 0u;
        rspt_light lt{}; lt.kind = RSPT_LIGHT_DIFFUSE_AREA; lt.prim = 0; lt.two_sided = (flags[i] & 4) ? 1u : 0u;
        for (int k = 0; k < 3; k++) lt.L[k] = L[3 * i + k];
        sc.d.P = tri + 9 * i; sc.d.N = nrm + 9 * i; sc.d.prims = &pr; sc.d.n_prims = 1; sc.d.meshes = &m; sc.d.n_meshes = 1; sc.d.lights = &lt; sc.d.n_lights = 1;
        Interaction iref; iref.p = V3{ref_p[3 * i], ref_p[3 * i + 1], ref_p[3 * i + 2]}; iref.p_error = V3{0, 0, 0}; iref.n = V3{0, 0, 0}; iref.wo = V3{0, 0, 0};
        Interaction li = iref; V3 wi{0, 0, 0}; Float pdf = 0.0f;
        const Spec s = light_sample_li(sc, lt, iref, P2{u[2 * i], u[2 * i + 1]}, &wi, &pdf, &li);
        float* o = out + 16 * i;
        o[0] = pdf; o[1] = wi.x; o[2] = wi.y; o[3] = wi.z; o[4] = s.c[0]; o[5] = s.c[1]; o[6] = s.c[2];
        o[7] = li.p.x; o[8] = li.p.y; o[9] = li.p.z; o[10] = li.n.x; o[11] = li.n.y; o[12] = li.n.z; o[13] = li.p_error.x; o[14] = li.p_error.y; o[15] = li.p_error.z;
        if (pdf == 0.0f) for (int k = 1; k < 7; k++) o[k] = 0.0f;
    }
}
void orc_geom_rng(const uint64_t* seq, const uint32_t* bound, uint64_t n, uint32_t* out_u, float* out_f) {
    for (uint64_t i = 0; i < n; i++) {
        Rng r; r.set_sequence(seq[i]);
        for (int k = 0; k < 4; k++) out_u[6 * i + k] = r.uniform_uint32();
        for (int k = 0; k < 2; k++) out_f[2 * i + k] = r.uniform_float();
        for (int k = 0; k < 2; k++) out_u[6 * i + 4 + k] = r.uniform_uint32_bounded(bound[i]);
    }
}

} // extern "C"
