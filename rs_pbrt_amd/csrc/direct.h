// DirectLightingIntegrator::li on the GPU (src/integrators/directlighting.rs:71-258).
//
// `li` recurses into specular reflection AND transmission, and the sampler's dimensions / 2-D sample arrays are consumed in
// that depth-first order, so a node's dimensions depend on how many nodes its earlier siblings' subtrees hold.  The wavefront
// form rests on one observation: the direction of a specular bounce does not depend on a sample value (a material with more
// than one specular-reflective or more than one specular-transmissive lobe is refused), so
//   1. the whole specular tree of every camera sample is traced first, level by level (k_dl_hit), in heap order: node h of
//      sample s lives in slot s * H + h, H = 2^max_depth, children 2h (reflection) and 2h + 1 (transmission);
//   2. one lane per camera sample walks its tree in the reference's order and gives every shading node its index among the
//      sample's shading nodes (which sample arrays it gets, sampler.rs get_2d_array) and its position in the dimension stream
//      (k_dl_assign);
//   3. the light estimates of all nodes run as wavefront rounds — one (light, array element) per round, shadow + MIS rays through
//      the same trace kernels as the path integrator — and are added in the reference's order (k_dl_nee / k_dl_nee_resolve);
//   4. one lane per camera sample folds the tree bottom-up: l = le + lights + f_r * li(reflected) * |cos| / pdf + ... (k_dl_gather).
// Null-BSDF hits continue inside their node (directlighting.rs:90-92).  Textured materials are not handled here yet
// (rspt_render refuses the combination).
#pragma once
#include "kernels.h"

namespace rspt {

enum : uint32_t { DL_EMPTY = 0, DL_SHADING = 1, DL_LEAF = 2, DL_PENDING = 3 };
enum : uint32_t { DLF_HAS_C1 = 1, DLF_HAS_C2 = 2, DLF_C2_ON_MISS = 4 };

struct DlBuf {          // per node slot
    float4* le_kind;    // (emitted radiance of a shading node | radiance of an escaped ray, kind)
    float4* w_r;        // edge to the reflection child: (f.rgb, |wi . ns| / pdf); all zero = no child
    float4* w_t;        // edge to the transmission child
    float4* l_all;      // running sum of uniform_sample_all_lights / _one_light; after k_dl_gather: li of the node
    float4* ld_acc;     // running `ld` of the current light (sample arrays)
    uint32_t* dim;      // dimension at which the node's lighting starts in the regular stream
    uint32_t* kidx;     // index among the camera sample's shading nodes, depth first
    uint32_t* nflags;   // DLF_* of the estimate in flight
    uint32_t vs, vr;    // k_dl_nee_all: estimate r of node slot n lives in the virtual slot v = n * vs + r * vr.  Round 6: planes — (vs, vr) = (1, node slots of the batch):
                        // the lanes of a wave, which hold neighbouring node slots, then store their rays / terms / flags of estimate r side by side (at stride R every 16-byte
                        // store of the wave met its own line: profiles/r06_pmc_calibration.md, "scattered stores are slow beyond their bytes") and the trace kernels and the
                        // resolve read them back the same way.  (R, 1) = round 5's interleaved layout, kept where a ray's slot must give its camera sample by ONE division
                        // (moving instances: SceneDev::time_div)
    uint32_t H;         // slots per camera sample = 2^levels
    uint32_t levels;    // levels of the tree that can hold nodes: max_depth, or 1 for a scene without specular lobes (no node ever has a child: the
                        // recursion's two sample_f calls return black and only consume their dimensions) — the slots, the level loops and the
                        // batch size follow it (untextured directlighting at depth 8 ran 1024 batches of 2^18 samples before)
    uint32_t* error;    // 1: a material with several specular lobes of one kind was met; 2: a camera sample ran out of sampler dimensions;
                        // 3: a specular bounce below the last level (the host's `levels` was wrong: never expected)
};

// per-wave aggregated queue append
RDEV void dl_push(bool want, uint32_t value, uint32_t* __restrict__ queue, uint32_t* counter) {
    const uint64_t m = __ballot(want);
    if (!m) return;
    const uint32_t lane = __lane_id();
    uint32_t base = 0;
    if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(counter, (uint32_t)__popcll(m));
    base = __shfl(base, __builtin_ctzll(m));
    if (want) queue[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = value;
}

// the interaction of a node's hit and its BSDF (Triangle::intersect second half + compute_scattering_functions with constant textures)
struct DlHit {
    Hit h;
    f3 wo;      // isect.common.wo
    Bsdf bsdf;
};
// TEX: the node's texture results are in pb.tex (k_dl_texture ran for it): the material's lobes are bound to them and Material::bump's shading geometry replaces the
// interpolated one, as in shade_path (kernels.h).  k_dl_hit, which runs before the texture stage, passes false: what it reads of the BSDF — specular lobes — does not exist in
// the scenes the wavefront form serves with textures (the host routes textures + specular lobes to the per-lane form).
template <uint32_t F = SF_ALL, bool TEX = false>
RDEV void dl_interaction(const SceneDev& sc, const PathBuf& pb, uint32_t slot, uint32_t prim, float4 hc, f3 ray_d, DlHit* o) {
    TriRec tri = load_tri(sc, prim);
    tri_fill<(F & SF_VERTEX) != 0>(sc, prim, tri, hc.y, hc.z, hc.w, &o->h);
    o->wo = -ray_d;
    if ((F & SF_INST) && pb.hit_inst) {
        const uint32_t hi = pb.hit_inst[slot];
        InstDev in{};   // the instance's Transform, a moving one's at the camera sample's time (node slots per sample = sc.time_div while the tree is traced)
        if (hi) in = inst_at(sc, hi - 1u, sc.ray_time ? sc.ray_time[slot / sc.time_div] : 0.0f);
        if (hi && !in.identity) {
            inst_hit(in, &o->h);
            o->wo = normalize(xf_vector(in.m, -xf_vector(in.mi, ray_d)));
            if (!sc.inst_fixed) { o->h.material = 0xffffffffu; o->h.area_light = -1; }
        }
    }
    if (o->h.material != 0xffffffffu) {
        const rspt_material mat = sc.materials[o->h.material];
        Bsdf& b = o->bsdf;
        b.eta = mat.eta; b.lt = LobeTex{nullptr, 0}; b.dropped = 0u;
        if (TEX && (F & SF_TEX) && pb.tex && sc.mat_flags && sc.mat_flags[o->h.material]) {   // (shade_path's textured assembly; dynamic lobe lists take the per-lane form)
            const float4* tb = pb.tex + slot;
            b.lt = LobeTex{tb, pb.tex_stride};
            const float4 m4 = tb[4 * (size_t)pb.tex_stride];
            const uint32_t tf = __float_as_uint(m4.w);
            b.dropped = (tf >> 8) & 0xffu;
            if (tf & 1u) {   // Material::bump replaced the shading geometry (material.rs:116-219)
                const float4 d4 = tb[5 * (size_t)pb.tex_stride];
                o->h.sh_n = f3{m4.x, m4.y, m4.z};
                o->h.sh_dpdu = f3{d4.x, d4.y, d4.z};
            }
        }
        b.ss = normalize(o->h.sh_dpdu); b.ns = o->h.sh_n; b.ng = o->h.n; b.ts = cross(o->h.sh_n, b.ss);
        b.lobes = sc.bxdfs + mat.first_bxdf;
        b.n = mat.n_bxdfs < 8u ? mat.n_bxdfs : 8u;
    }
}

// roots: the camera rays (k_raygen left them in src_rays by sample slot)
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_dl_init(Batch bt, PathBuf pb, DlBuf dl, const rspt_ray* __restrict__ src_rays, uint32_t* __restrict__ q_level0, uint32_t* cnt_level0) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0) *cnt_level0 = bt.n;
    if (i >= bt.n) return;
    const uint32_t slot = i * dl.H + 1u;
    pb.ray_cont[slot] = src_rays[i];
    dl.le_kind[slot] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float((uint32_t)DL_PENDING));
    q_level0[i] = slot;
}

// one level of the specular tree: classify every traced node, spawn its children into the next level's queue; null-BSDF hits
// re-enter the same level (q_retrace)
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_dl_hit(SceneDev sc, RenderDev rd, PathBuf pb, DlBuf dl, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_in,
                                                uint32_t* __restrict__ q_retrace, uint32_t* cnt_retrace, uint32_t* __restrict__ q_next, uint32_t* cnt_next, uint32_t level) {
    const uint32_t n = *count_in;
    for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
        const uint32_t i = base + threadIdx.x;
        bool retrace = false, kid_r = false, kid_t = false;
        uint32_t slot = 0;
        if (i < n) {
            slot = queue[i];
            const float4 hc = pb.hit_cont[slot];
            const uint32_t prim = __float_as_uint(hc.x);
            const float4* rp = reinterpret_cast<const float4*>(pb.ray_cont + slot);
            const float4 r0 = rp[0], r1 = rp[1];
            const f3 ray_d{r0.w, r1.x, r1.y};
            if (prim == RSPT_MISS) {  // for light in &scene.lights { l += light.le(ray) } (directlighting.rs:117-121): only infinite lights emit there
                rgb l = mkrgb(0.0f);
                for (uint32_t k = 0; k < sc.n_infinite; k++) l = l + infinite_le(sc, sc.lights[sc.infinite_lights[k]], ray_d);
                dl.le_kind[slot] = make_float4(l.r, l.g, l.b, __uint_as_float((uint32_t)DL_LEAF));
            } else {
                DlHit d;
                dl_interaction(sc, pb, slot, prim, hc, ray_d, &d);
                if (d.h.material == 0xffffffffu) {  // isect.bsdf.is_none(): return self.li(&isect.spawn_ray(&ray.d), ..., depth) (:90-92)
                    store_ray(pb.ray_cont + slot, offset_ray_origin(d.h.p, d.h.p_err, d.h.n, ray_d), ray_d, RSPT_INF, slot);
                    pb.state[slot] = ST_NO_DIFF;   // isect.spawn_ray: no differentials (k_dl_texture; the host zeroes the node slots' words per batch)
                    retrace = true;
                } else {
                    const rgb le = d.h.area_light >= 0 ? light_l(sc.lights[d.h.area_light], d.h.n, d.wo) : mkrgb(0.0f);  // l += isect.le(&wo)
                    dl.le_kind[slot] = make_float4(le.r, le.g, le.b, __uint_as_float((uint32_t)DL_SHADING));
                    float4 wr = make_float4(0.0f, 0.0f, 0.0f, 0.0f), wt = wr;
                    if (level + 1u < rd.max_depth) {  // specular_reflect / specular_transmit (:124-258), ray differentials left out
                        if (d.bsdf.num_components(BX_REFL | BX_SPEC) > 1 || d.bsdf.num_components(BX_TRANS | BX_SPEC) > 1) atomicMax(dl.error, 1u);
                        const uint32_t h = slot % dl.H, s = slot / dl.H;
                        for (int side = 0; side < 2; side++) {
                            f3 wi{0.0f, 0.0f, 0.0f};
                            float pdf = 0.0f;
                            uint32_t st = 0;
                            const rgb f = d.bsdf.sample_f(d.wo, &wi, f2{0.0f, 0.0f}, &pdf, (side ? BX_TRANS : BX_REFL) | BX_SPEC, &st);
                            if (pdf > 0.0f && !is_black(f) && absdot(wi, d.h.sh_n) != 0.0f) {
                                if (level + 1u >= dl.levels) { atomicMax(dl.error, 3u); continue; }
                                const uint32_t child = s * dl.H + 2u * h + (uint32_t)side;
                                store_ray(pb.ray_cont + child, offset_ray_origin(d.h.p, d.h.p_err, d.h.n, wi), wi, RSPT_INF, child);
                                dl.le_kind[child] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float((uint32_t)DL_PENDING));
                                const float4 w = make_float4(f.r, f.g, f.b, absdot(wi, d.h.sh_n) / pdf);
                                if (side) { wt = w; kid_t = true; } else { wr = w; kid_r = true; }
                            }
                        }
                    }
                    dl.w_r[slot] = wr; dl.w_t[slot] = wt;
                }
            }
        }
        dl_push(retrace, slot, q_retrace, cnt_retrace);
        const uint32_t h = slot % dl.H, s = slot / dl.H;
        dl_push(kid_r, s * dl.H + 2u * h, q_next, cnt_next);
        dl_push(kid_t, s * dl.H + 2u * h + 1u, q_next, cnt_next);
    }
}

// round 6: the texture stage of the wavefront form — scenes with textured materials and NO specular lobes (the tree is its roots: every node's ray is a camera ray, whose
// differentials compute_differentials takes from the camera, directlighting.rs:86 -> compute_scattering_functions).  Textures next to specular lobes keep the per-lane form,
// which carries the reflected / refracted differentials down the tree (dl_serial.h).
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_dl_texture(SceneDev sc, TexTables tt, RenderDev rd, PathBuf pb, DlBuf dl, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_in) {
    const uint32_t n = *count_in;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t slot = queue[i];
        if (__float_as_uint(dl.le_kind[slot].w) != DL_SHADING) continue;
        [[clang::always_inline]] texture_slot(sc, tt, rd, pb, slot, slot / dl.H, slot / sc.time_div, pb.state[slot] != ST_NO_DIFF, nullptr);
    }
}

// the reference's depth-first order: which sample arrays and which dimensions every shading node gets
// n_arrays = 2 * max_depth * n_lights with LightStrategy::UniformSampleAll (preprocess, directlighting.rs:54-70), else 0
// dim_limit: the sampler's dimension count (NUM_SOBOL_DIMENSIONS, or what the Halton permutation table covers); the reference panics when a
// dimension past it is asked for (sobol.rs:119-124), so a camera sample whose stream ends beyond it is reported (dl.error = 2), not rendered
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_dl_assign(Batch bt, DlBuf dl, uint32_t n_lights, uint32_t n_arrays, uint32_t sample_all, uint32_t max_depth, uint32_t dim_limit) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= bt.n) return;
    uint32_t k = 0;
    uint32_t dim = 5u + 2u * n_arrays;  // get_1d / get_2d jump over the array dimensions, two per array (sobol.rs:180-201): the regular stream starts behind them
    uint32_t stack[32];            // (heap index << 1) | stage
    int sp = 0;
    stack[sp++] = 1u << 1;
    while (sp > 0) {
        const uint32_t e = stack[--sp], h = e >> 1, stage = e & 1u;
        const uint32_t slot = s * dl.H + h;
        if (stage == 0u) {
            if (__float_as_uint(dl.le_kind[slot].w) != DL_SHADING) continue;
            dl.kidx[slot] = k; dl.dim[slot] = dim;
            if (n_lights) {
                if (sample_all) {  // lights whose array pair is used up fall back to get_2d() x 2 (integrator.rs:316-329)
                    const uint32_t pairs = n_arrays / 2u, first = k * n_lights;
                    const uint32_t with_arrays = first >= pairs ? 0u : (pairs - first < n_lights ? pairs - first : n_lights);
                    dim += 4u * (n_lights - with_arrays);
                } else dim += 5u;  // get_1d (light choice), get_2d, get_2d (integrator.rs:378-392)
            }
            k++;
            const uint32_t depth = 31u - (uint32_t)__builtin_clz(h);
            if (depth + 1u < max_depth) {
                dim += 2u;                                            // specular_reflect's sampler.get_2d()
                stack[sp++] = (h << 1) | 1u;                          // after the reflection subtree: the transmission side
                if (depth + 1u < dl.levels && __float_as_uint(dl.le_kind[s * dl.H + 2u * h].w) != DL_EMPTY) stack[sp++] = (2u * h) << 1;
            }
        } else {
            dim += 2u;                                                // specular_transmit's sampler.get_2d()
            const uint32_t depth = 31u - (uint32_t)__builtin_clz(h);
            if (depth + 1u < dl.levels && __float_as_uint(dl.le_kind[s * dl.H + 2u * h + 1u].w) != DL_EMPTY) stack[sp++] = (2u * h + 1u) << 1;
        }
    }
    if (dim > dim_limit) atomicMax(dl.error, 2u);
}

RDEV f2 dl_dims(const RenderDev& rd, uint64_t index, uint32_t d) {
    return rd.sampler_kind == RSPT_SAMPLER_HALTON ? f2{halton_dim(rd, index, d), halton_dim(rd, index, d + 1u)} : f2{sobol_dim(rd, index, d), sobol_dim(rd, index, d + 1u)};
}

// one round of estimate_direct (integrator.rs:406-570) for every shading node of a level: light j, element kk of its sample arrays
// (sample_all), or the one light uniform_sample_one_light picks (j, kk = 0)
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_dl_nee(SceneDev sc, RenderDev rd, Batch bt, PathBuf pb, DlBuf dl, const uint32_t* __restrict__ pix_list,
                                                const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_in, uint32_t j, uint32_t kk, uint32_t n_j,
                                                uint32_t n_arrays, uint32_t sample_all, uint32_t* __restrict__ q_any, uint32_t* cnt_any,
                                                uint32_t* __restrict__ q_mis, uint32_t* cnt_mis) {
    const uint32_t n = *count_in;
    for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
        const uint32_t i = base + threadIdx.x;
        bool want_sh = false, want_mis = false;
        uint32_t slot = 0;
        if (i < n) {
            slot = queue[i];
            uint32_t fl = 0;
            if (__float_as_uint(dl.le_kind[slot].w) == DL_SHADING) {
                const uint32_t s = slot / dl.H;
                const uint32_t nl = sc.n_lights;
                // ---- the sample values of this estimate ----
                f2 u_light, u_scatter;
                uint32_t light_num = j;
                float choice_pdf = 1.0f;
                bool active = true;
                const uint64_t index = pb.sobol_index[s];
                if (sample_all) {
                    const uint32_t pair = dl.kidx[slot] * nl + j;
                    if (pair < n_arrays / 2u) {  // get_2d_array_idxs -> get_2d_sample (sobol.rs:214-236): element cur_sample * n + kk of arrays 2 pair, 2 pair + 1
                        const uint32_t pk = pix_list[bt.pix0 + s / bt.ns];
                        const int32_t px = (int32_t)(int16_t)(pk & 0xffffu), py = (int32_t)(int16_t)(pk >> 16);
                        const uint64_t elem = (uint64_t)(bt.s0 + s % bt.ns) * n_j + kk;
                        const uint64_t ei = rd.sampler_kind == RSPT_SAMPLER_HALTON ? halton_index(rd, px, py, elem)
                                                                                  : sobol_interval_to_index(rd, (uint32_t)rd.log2_res, elem, px - rd.sample_bounds[0], py - rd.sample_bounds[1]);
                        u_light = dl_dims(rd, ei, 5u + 4u * pair);
                        u_scatter = dl_dims(rd, ei, 5u + 4u * pair + 2u);
                    } else if (kk == 0u) {       // a single estimate from the regular stream
                        const uint32_t pairs = n_arrays / 2u, first = dl.kidx[slot] * nl;
                        const uint32_t j0 = first >= pairs ? 0u : pairs - first;  // first light of this node without arrays
                        const uint32_t d = dl.dim[slot] + 4u * (j - j0);
                        u_light = dl_dims(rd, index, d); u_scatter = dl_dims(rd, index, d + 2u);
                    } else active = false;
                } else {                          // uniform_sample_one_light with light_distrib = None (integrator.rs:359-403)
                    const uint32_t d = dl.dim[slot];
                    const float u1 = rd.sampler_kind == RSPT_SAMPLER_HALTON ? halton_dim(rd, index, d) : sobol_dim(rd, index, d);
                    const uint32_t pick = (uint32_t)(u1 * (float)nl);
                    light_num = pick < nl - 1u ? pick : nl - 1u;
                    choice_pdf = 1.0f / (float)nl;
                    u_light = dl_dims(rd, index, d + 1u); u_scatter = dl_dims(rd, index, d + 3u);
                }
                if (active) {
                    const float4 hc = pb.hit_cont[slot];
                    const float4* rp = reinterpret_cast<const float4*>(pb.ray_cont + slot);
                    const float4 r0 = rp[0], r1 = rp[1];
                    DlHit d;
                    dl_interaction(sc, pb, slot, __float_as_uint(hc.x), hc, f3{r0.w, r1.x, r1.y}, &d);
                    const Hit& h = d.h;
                    const uint32_t flags = BX_ALL & ~BX_SPEC;  // estimate_direct(.., specular = false): bsdf_flags = BsdfAll & !BsdfSpecular (integrator.rs:419-423)
                    const rspt_light lt = sc.lights[light_num];
                    rgb c1 = mkrgb(0.0f), c2 = mkrgb(0.0f);
                    f3 wi{0.0f, 0.0f, 0.0f};
                    float light_pdf = 0.0f, scattering_pdf = 0.0f;
                    LightSample ls;
                    const rgb li = light_sample_li(sc, lt, h.p, u_light, &wi, &light_pdf, &ls);
                    if (light_pdf > 0.0f && !is_black(li)) {
                        const rgb f = d.bsdf.f(d.wo, wi, flags) * mkrgb(absdot(wi, h.sh_n));
                        scattering_pdf = d.bsdf.pdf(d.wo, wi, flags);
                        if (!is_black(f)) {
                            const f3 origin = offset_ray_origin(h.p, h.p_err, h.n, ls.p - h.p);
                            const f3 target = offset_ray_origin(ls.p, ls.p_err, ls.n, origin - ls.p);
                            store_ray(pb.ray_sh + slot, origin, target - origin, 1.0f - RSPT_SHADOW_EPS, slot);
                            want_sh = true;
                            if (light_is_delta(lt)) c1 = f * li / light_pdf;
                            else c1 = f * li * mkrgb(power_heuristic(light_pdf, scattering_pdf)) / light_pdf;
                            fl |= DLF_HAS_C1;
                        }
                    }
                    if (!light_is_delta(lt)) {
                        uint32_t sampled_type = 0;
                        rgb f = d.bsdf.sample_f(d.wo, &wi, u_scatter, &scattering_pdf, flags, &sampled_type);
                        f = f * mkrgb(absdot(wi, h.sh_n));
                        if (!is_black(f) && scattering_pdf > 0.0f) {
                            const f3 ro = offset_ray_origin(h.p, h.p_err, h.n, wi);
                            float lpdf = 0.0f;
                            rgb le_mis = ldrgb(lt.L);
                            if (lt.kind == RSPT_LIGHT_INFINITE) {
                                lpdf = infinite_pdf_li(sc, lt, wi);
                                if (lpdf != 0.0f) le_mis = infinite_le(sc, lt, wi);
                            } else {
                                const TriRec lt_tri = load_tri(sc, lt.prim);
                                float t_l, lb0, lb1, lb2;
                                if (tri_test(lt_tri.p0, lt_tri.p1, lt_tri.p2, ro, ray_shear(wi), RSPT_INF, &t_l, &lb0, &lb1, &lb2)) {
                                    Hit lh;
                                    tri_fill(sc, lt.prim, lt_tri, lb0, lb1, lb2, &lh);
                                    lpdf = dist2(h.p, lh.p) / (absdot(lh.n, -wi) * tri_area(lt_tri));
                                    if (__builtin_isinf(lpdf)) lpdf = 0.0f;
                                }
                            }
                            if (lpdf != 0.0f) {
                                c2 = f * le_mis * mkrgb(1.0f) * power_heuristic(scattering_pdf, lpdf) / scattering_pdf;
                                if (lt.kind != RSPT_LIGHT_INFINITE || !is_black(le_mis)) {
                                    store_ray(pb.ray_mis + slot, ro, wi, RSPT_INF, slot);
                                    want_mis = true;
                                    fl |= DLF_HAS_C2 | (lt.kind == RSPT_LIGHT_INFINITE ? DLF_C2_ON_MISS : 0u);
                                }
                            }
                        }
                    }
                    pb.nee_c1[slot] = make_float4(c1.r, c1.g, c1.b, choice_pdf);
                    pb.nee_c2[slot] = make_float4(c2.r, c2.g, c2.b, __uint_as_float(light_num));
                    fl |= 0x100u;  // an estimate is in flight
                }
            }
            dl.nflags[slot] = fl;
        }
        dl_push(want_sh, slot, q_any, cnt_any);
        dl_push(want_mis, slot | RSPT_Q_MIS, q_mis, cnt_mis);
    }
}

// fold the round's estimate into the node, in the reference's order (integrator.rs:309-353)
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_dl_nee_resolve(SceneDev sc, PathBuf pb, DlBuf dl, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_in,
                                                        uint32_t j, uint32_t kk, uint32_t n_j, uint32_t n_arrays, uint32_t sample_all) {
    const uint32_t n = *count_in;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t slot = queue[i];
        const uint32_t fl = dl.nflags[slot];
        if (!(fl & 0x100u)) continue;
        const float4 c1 = pb.nee_c1[slot], c2 = pb.nee_c2[slot];
        rgb ld = mkrgb(0.0f);
        if ((fl & DLF_HAS_C1) && pb.occluded[slot] == 0u) ld = ld + rgb{c1.x, c1.y, c1.z};
        if (fl & DLF_HAS_C2) {
            const float4 hm = pb.hit_mis[slot];
            const uint32_t hp = __float_as_uint(hm.x), light_num = __float_as_uint(c2.w);
            if (fl & DLF_C2_ON_MISS) {
                if (hp == RSPT_MISS) ld = ld + rgb{c2.x, c2.y, c2.z};
            } else if (hp != RSPT_MISS) {
                const TriRec t = load_tri(sc, hp);
                if (t.area_light >= 0 && (uint32_t)t.area_light == light_num) {
                    Hit h;
                    tri_fill(sc, hp, t, hm.y, hm.z, hm.w, &h);
                    const float4* mr = reinterpret_cast<const float4*>(pb.ray_mis + slot);
                    const float4 m0 = mr[0], m1 = mr[1];
                    if (!is_black(light_l(sc.lights[light_num], h.n, -f3{m0.w, m1.x, m1.y}))) ld = ld + rgb{c2.x, c2.y, c2.z};
                }
            }
        }
        const float4 la = dl.l_all[slot];
        rgb l = rgb{la.x, la.y, la.z};
        if (!sample_all) l = l + ld / c1.w;  // estimate_direct(..) / light_pdf
        else if (dl.kidx[slot] * sc.n_lights + j < n_arrays / 2u) {
            const float4 a4 = dl.ld_acc[slot];
            const rgb acc = rgb{a4.x, a4.y, a4.z} + ld;
            if (kk + 1u == n_j) { l = l + acc / (float)n_j; dl.ld_acc[slot] = make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
            else dl.ld_acc[slot] = make_float4(acc.r, acc.g, acc.b, 0.0f);
        } else l = l + ld;
        dl.l_all[slot] = make_float4(l.r, l.g, l.b, 0.0f);
    }
}

// ---- round 5: ALL light estimates of a node in ONE round ---------------------------------------------------------------------------------------
// uniform_sample_all_lights (integrator.rs:300-357) makes R = sum_j n_j estimates per shading node.  The round-per-estimate form above launches
// (k_dl_nee, two traces, k_dl_nee_resolve) R times per tree level and rebuilds the node's interaction in each — on the C3 stand-in (six light triangles)
// k_dl_nee alone was 48 % of the step.  Here the interaction is built once, estimate r of node slot n lives in the VIRTUAL slot v = n * R + r of the ray /
// result arrays (ray_sh, ray_mis, occluded, hit_mis, nee_c1, nee_c2, nflags: the trace kernels index by queue entry, so nothing changes for them), one
// any-hit launch and one closest-hit launch serve all of them, and k_dl_nee_resolve_all adds them in the reference's order (`ld += ..` over a light's
// elements, `l += ld / n`).  Same operations on the same values as the rounds: bit-identical radiance; the batch shrinks by R.

// Sobol' tables of the estimate kernel in LDS (round 6).  Every estimate draws FOUR dimensions (u_light, u_scatter) at an index of its own — the sample arrays of
// uniform_sample_all_lights have one element per (pixel sample, kk), sampler.rs request_2d_array / sobol.rs:110-140 — and sobol_dim walks the set bits of that index once
// per dimension through the 213 KB generator table in global memory (~25 dependent L2 round trips each), after sobol_interval_to_index's two walks through the van der
// Corput matrices: ~130 serial cache misses per estimate, six estimates per node on the C3 stand-in — k_dl_nee_all spent its time there, not in estimate_direct.  The
// kernel now keeps, per workgroup, the generator columns of the dimensions this render can reach, transposed to [bit][dimension] as the shade stage's (kernels.h
// shade_kernel), and the two van der Corput matrices of the film's resolution: ONE walk over the index's bits yields the four values.  Same XORs, same values.
struct DlSob {
    const uint32_t* tab;   // [bits][nd] in LDS; nullptr: Halton, or the tables do not fit — the global walks
    uint32_t nd, bits;
    const uint64_t* m;     // vdc rows of log2_res [52], then vdc_inv rows [52], in LDS
    // The scene's lights and the triangle records of the area lights, in LDS (nullptr: more lights than DL_LDS_LIGHTS).  Not for their bytes — every lane of a wave reads
    // the SAME record, an L1 hit — but for the counter they wait on: gfx950 has one vmcnt for loads and stores, so a global load issued behind an estimate's five stores
    // (shadow ray, two term vectors, flags) is waited for with vmcnt(0), i.e. until those stores are acknowledged by L2 — once per estimate, at two waves per SIMD:
    // SQ_WAIT_ANY 0.69 of the kernel's wave cycles, VALU busy 0.20 (profiles/r06_directlighting_pmc.txt).  LDS reads count on lgkmcnt and pass the stores.
    const uint32_t* lights;   // rspt_light[n] as words
    const float4* tris;       // [3 * n]: the triangle record of light j (area lights; zeros for the others)
};
#define DL_LDS_LIGHTS 64u
#ifndef RSPT_DL_EXP
#define RSPT_DL_EXP 0   // timing experiments on k_dl_nee_all (tools/ab_build.sh with AB_DEFS=-DRSPT_DL_EXP=n; WRONG PICTURES, never the shipped build): 1 = no result stores,
                        // 2 = no queue appends, 3 = no BSDF-sampled half of estimate_direct
#endif
RDEV rspt_light dl_light(const SceneDev& sc, const DlSob& sb, uint32_t j) {
    if (!sb.lights) return sc.lights[j];
    rspt_light lt;
    uint32_t* w = reinterpret_cast<uint32_t*>(&lt);
    const uint32_t* src = sb.lights + j * (uint32_t)(sizeof(rspt_light) / 4);
#pragma unroll
    for (uint32_t k = 0; k < sizeof(rspt_light) / 4; k++) w[k] = src[k];
    return lt;
}
RDEV TriRec dl_light_tri(const SceneDev& sc, const DlSob& sb, const rspt_light& lt, uint32_t j) {
    if (!sb.lights) return load_tri(sc, lt.prim);
    const float4 a = sb.tris[3u * j], b = sb.tris[3u * j + 1u], c = sb.tris[3u * j + 2u];
    TriRec t;
    t.p0 = f3{a.x, a.y, a.z}; t.p1 = f3{a.w, b.x, b.y}; t.p2 = f3{b.z, b.w, c.x};
    t.material = __float_as_uint(c.y); t.area_light = (int32_t)__float_as_uint(c.z); t.flags = __float_as_uint(c.w);
    return t;
}
RDEV uint64_t dl_interval_to_index(const RenderDev& rd, const DlSob& sb, uint32_t m, uint64_t frame, int32_t px, int32_t py) {
    if (!sb.tab) return sobol_interval_to_index(rd, m, frame, px, py);
    if (m == 0) return 0;
    uint64_t index = frame << (m << 1);
    uint64_t delta = 0;
    for (uint64_t f = frame; f != 0; f &= f - 1) delta ^= sb.m[__builtin_ctzll(f)];
    uint64_t b = ((uint64_t)((uint32_t)px << m) | (uint64_t)(int64_t)py) ^ delta;
    for (; b != 0; b &= b - 1) index ^= sb.m[52 + __builtin_ctzll(b)];
    return index;
}
// dimensions d .. d + 3 at `index`
RDEV void dl_dims4(const RenderDev& rd, const DlSob& sb, uint64_t index, uint32_t d, f2* a, f2* b) {
    if (sb.tab && d + 4u <= sb.nd && (index >> sb.bits) == 0) {
        uint32_t x0 = 0, x1 = 0, x2 = 0, x3 = 0;
        for (uint64_t i = index; i != 0; i &= i - 1) {
            const uint32_t* row = sb.tab + (uint32_t)__builtin_ctzll(i) * sb.nd + d;
            x0 ^= row[0]; x1 ^= row[1]; x2 ^= row[2]; x3 ^= row[3];
        }
        *a = f2{fminf((float)x0 * 0x1.0p-32f, RSPT_ONE_MINUS_EPS), fminf((float)x1 * 0x1.0p-32f, RSPT_ONE_MINUS_EPS)};
        *b = f2{fminf((float)x2 * 0x1.0p-32f, RSPT_ONE_MINUS_EPS), fminf((float)x3 * 0x1.0p-32f, RSPT_ONE_MINUS_EPS)};
        return;
    }
    *a = dl_dims(rd, index, d); *b = dl_dims(rd, index, d + 2u);
}

// the (u_light, u_scatter, light, choice pdf) of estimate (j, kk) of a node — the sample-value part of k_dl_nee, unchanged
RDEV bool dl_estimate_samples(const RenderDev& rd, const DlSob& sb, const Batch& bt, const PathBuf& pb, const DlBuf& dl, const uint32_t* __restrict__ pix_list, uint32_t slot, uint32_t nl,
                              uint32_t j, uint32_t kk, uint32_t n_j, uint32_t n_arrays, uint32_t sample_all, f2* u_light, f2* u_scatter, uint32_t* light_num, float* choice_pdf) {
    const uint32_t s = slot / dl.H;
    const uint64_t index = pb.sobol_index[s];
    *light_num = j; *choice_pdf = 1.0f;
    if (sample_all) {
        const uint32_t pair = dl.kidx[slot] * nl + j;
        if (pair < n_arrays / 2u) {
            const uint32_t pk = pix_list[bt.pix0 + s / bt.ns];
            const int32_t px = (int32_t)(int16_t)(pk & 0xffffu), py = (int32_t)(int16_t)(pk >> 16);
            const uint64_t elem = (uint64_t)(bt.s0 + s % bt.ns) * n_j + kk;
            const uint64_t ei = rd.sampler_kind == RSPT_SAMPLER_HALTON ? halton_index(rd, px, py, elem)
                                                                      : dl_interval_to_index(rd, sb, (uint32_t)rd.log2_res, elem, px - rd.sample_bounds[0], py - rd.sample_bounds[1]);
            dl_dims4(rd, sb, ei, 5u + 4u * pair, u_light, u_scatter);
            return true;
        }
        if (kk != 0u) return false;
        const uint32_t pairs = n_arrays / 2u, first = dl.kidx[slot] * nl;
        const uint32_t j0 = first >= pairs ? 0u : pairs - first;
        const uint32_t d = dl.dim[slot] + 4u * (j - j0);
        dl_dims4(rd, sb, index, d, u_light, u_scatter);
        return true;
    }
    const uint32_t d = dl.dim[slot];
    const float u1 = rd.sampler_kind == RSPT_SAMPLER_HALTON ? halton_dim(rd, index, d) : sobol_dim(rd, index, d);
    const uint32_t pick = (uint32_t)(u1 * (float)nl);
    *light_num = pick < nl - 1u ? pick : nl - 1u;
    *choice_pdf = 1.0f / (float)nl;
    dl_dims4(rd, sb, index, d + 1u, u_light, u_scatter);
    return true;
}

// estimate_direct (integrator.rs:406-570) of one light for a built interaction; rays and terms go to virtual slot v.  Returns the DLF_* flags | 0x100.
template <uint32_t F>   // the feature set the instantiation is compiled for (dev_bsdf.h SF_*: lobe kinds, light kinds, per-vertex normals), as k_shade<F>
RDEVN uint32_t dl_estimate(const SceneDev& sc, const DlSob& sb, const PathBuf& pb, const DlHit& d, uint32_t light_num, float choice_pdf, f2 u_light, f2 u_scatter, uint32_t v, bool* want_sh, bool* want_mis) {
    const Hit& h = d.h;
    uint32_t fl = 0;
    const uint32_t flags = BX_ALL & ~BX_SPEC;
    const rspt_light lt = dl_light(sc, sb, light_num);
    const bool is_area = !(F & (SF_L_POINT | SF_L_SPOT | SF_L_DISTANT | SF_L_INFINITE)) || lt.kind == RSPT_LIGHT_DIFFUSE_AREA;
    TriRec lt_tri{};
    if (is_area) lt_tri = dl_light_tri(sc, sb, lt, light_num);
    rgb c1 = mkrgb(0.0f), c2 = mkrgb(0.0f);
    f3 wi{0.0f, 0.0f, 0.0f};
    float light_pdf = 0.0f, scattering_pdf = 0.0f;
    LightSample ls;
    const rgb li = light_sample_li<F>(sc, lt, h.p, u_light, &wi, &light_pdf, &ls, is_area ? &lt_tri : nullptr);
    if (light_pdf > 0.0f && !is_black(li)) {
        const rgb f = d.bsdf.template f<F>(d.wo, wi, flags) * mkrgb(absdot(wi, h.sh_n));
        scattering_pdf = d.bsdf.template pdf<F>(d.wo, wi, flags);
        if (!is_black(f)) {
            const f3 origin = offset_ray_origin(h.p, h.p_err, h.n, ls.p - h.p);
            const f3 target = offset_ray_origin(ls.p, ls.p_err, ls.n, origin - ls.p);
            if (RSPT_DL_EXP != 1) store_ray(pb.ray_sh + v, origin, target - origin, 1.0f - RSPT_SHADOW_EPS, v);
            *want_sh = true;
            if (light_is_delta<F>(lt)) c1 = f * li / light_pdf;
            else c1 = f * li * mkrgb(power_heuristic(light_pdf, scattering_pdf)) / light_pdf;
            fl |= DLF_HAS_C1;
        }
    }
    if (RSPT_DL_EXP != 3 && !light_is_delta<F>(lt)) {
        uint32_t sampled_type = 0;
        // (round 6, as kernels.h shade_path: an area light's BSDF-sampled term needs the sampled direction to meet the light's own triangle — tested as soon as the
        //  lobe has chosen the direction; a miss skips the other lobes' pdfs and the lobes' values)
        const bool area_mis = !((F & SF_L_INFINITE) && lt.kind == RSPT_LIGHT_INFINITE) && RSPT_MIS_EARLY_OUT;
        float t_l = 0.0f, lb0 = 0.0f, lb1 = 0.0f, lb2 = 0.0f;
        bool on_light = false;
        rgb f = d.bsdf.template sample_f_if<F>(d.wo, &wi, u_scatter, &scattering_pdf, flags, &sampled_type, [&](f3 w) {
            if (!area_mis) return false;
            on_light = tri_test(lt_tri.p0, lt_tri.p1, lt_tri.p2, offset_ray_origin(h.p, h.p_err, h.n, w), ray_shear(w), RSPT_INF, &t_l, &lb0, &lb1, &lb2);
            return !on_light;
        });
        f = f * mkrgb(absdot(wi, h.sh_n));
        if (!is_black(f) && scattering_pdf > 0.0f) {
            const f3 ro = offset_ray_origin(h.p, h.p_err, h.n, wi);
            float lpdf = 0.0f;
            rgb le_mis = ldrgb(lt.L);
            if ((F & SF_L_INFINITE) && lt.kind == RSPT_LIGHT_INFINITE) {
                lpdf = infinite_pdf_li(sc, lt, wi);
                if (lpdf != 0.0f) le_mis = infinite_le(sc, lt, wi);
            } else {
                if (!area_mis) on_light = tri_test(lt_tri.p0, lt_tri.p1, lt_tri.p2, ro, ray_shear(wi), RSPT_INF, &t_l, &lb0, &lb1, &lb2);
                if (on_light) {
                    Hit lh;
                    tri_fill<(F & SF_VERTEX) != 0>(sc, lt.prim, lt_tri, lb0, lb1, lb2, &lh);
                    lpdf = dist2(h.p, lh.p) / (absdot(lh.n, -wi) * tri_area(lt_tri));
                    if (__builtin_isinf(lpdf)) lpdf = 0.0f;
                }
            }
            if (lpdf != 0.0f) {
                c2 = f * le_mis * mkrgb(1.0f) * power_heuristic(scattering_pdf, lpdf) / scattering_pdf;
                if (!((F & SF_L_INFINITE) && lt.kind == RSPT_LIGHT_INFINITE) || !is_black(le_mis)) {
                    store_ray(pb.ray_mis + v, ro, wi, RSPT_INF, v);
                    *want_mis = true;
                    fl |= DLF_HAS_C2 | (((F & SF_L_INFINITE) && lt.kind == RSPT_LIGHT_INFINITE) ? DLF_C2_ON_MISS : 0u);
                }
            }
        }
    }
    if (RSPT_DL_EXP == 1) { if (c1.r + c2.r == 1e30f) pb.nee_c1[v] = make_float4(c1.r, c1.g, c1.b, choice_pdf); return fl | 0x100u; }
    pb.nee_c1[v] = make_float4(c1.r, c1.g, c1.b, choice_pdf);
    pb.nee_c2[v] = make_float4(c2.r, c2.g, c2.b, __uint_as_float(light_num));
    return fl | 0x100u;
}

// nls: n_light_samples per light on the device (nullptr: one each); R = the number of estimates per node = sum_j n_j (sample_all) or 1
#define RSPT_DL_NEE_ARGS SceneDev sc, RenderDev rd, Batch bt, PathBuf pb, DlBuf dl, const uint32_t* __restrict__ pix_list, const uint32_t* __restrict__ queue, \
                         const uint32_t* __restrict__ count_in, const int32_t* __restrict__ nls, uint32_t R, uint32_t n_arrays, uint32_t sample_all,            \
                         uint32_t* __restrict__ q_any, uint32_t* cnt_any, uint32_t* __restrict__ q_mis, uint32_t* cnt_mis, uint32_t sob_nd, uint32_t sob_bits
template <uint32_t F>
__device__ __forceinline__ void dl_nee_all(const SceneDev& sc, const RenderDev& rd, const Batch& bt, const PathBuf& pb, const DlBuf& dl, const uint32_t* __restrict__ pix_list,
                                           const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_in, const int32_t* __restrict__ nls, uint32_t R,
                                           uint32_t n_arrays, uint32_t sample_all, uint32_t* __restrict__ q_any, uint32_t* cnt_any,
                                           uint32_t* __restrict__ q_mis, uint32_t* cnt_mis, uint32_t sob_nd, uint32_t sob_bits) {
    const uint32_t n = *count_in;
    const uint32_t nl = sc.n_lights, n_lights_round = sample_all ? nl : 1u;
    // the Sobol' tables of this render's dimensions in LDS (DlSob; sob_nd = 0: Halton, or they do not fit)
    extern __shared__ __attribute__((aligned(16))) uint64_t dl_lds[];
    DlSob sb{nullptr, 0u, 0u, nullptr, nullptr, nullptr};
    // [lights: 3 x float4 of triangles + rspt_light words per light][vdc rows 104 x u64][Sobol' columns]; the host sizes the launch's dynamic LDS to match (lds_lights = the
    // number of lights staged, 0 = none)
    const uint32_t lds_lights = sob_bits >> 16;
    sob_bits &= 0xffffu;
    uint64_t* after_lights = dl_lds;
    if (lds_lights) {
        float4* lt_tris = reinterpret_cast<float4*>(dl_lds);
        uint32_t* lt_words = reinterpret_cast<uint32_t*>(lt_tris + 3u * lds_lights);
        constexpr uint32_t LW = (uint32_t)(sizeof(rspt_light) / 4);
        for (uint32_t t = threadIdx.x; t < lds_lights * LW; t += 256u) lt_words[t] = reinterpret_cast<const uint32_t*>(sc.lights)[t];
        for (uint32_t t = threadIdx.x; t < 3u * lds_lights; t += 256u) {
            const rspt_light& l = sc.lights[t / 3u];
            lt_tris[t] = l.kind == RSPT_LIGHT_DIFFUSE_AREA ? sc.tris[3 * (size_t)l.prim + t % 3u] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        sb.lights = lt_words; sb.tris = lt_tris;
        after_lights = dl_lds + (lds_lights * (48u + 4u * LW) + 7u) / 8u;
        if (!sob_nd) __syncthreads();
    }
    if (sob_nd) {
        uint64_t* vm = after_lights;
        uint32_t* tab = reinterpret_cast<uint32_t*>(after_lights + 104);
        const uint32_t m = (uint32_t)rd.log2_res;
        for (uint32_t t = threadIdx.x; t < 104u; t += 256u) vm[t] = m == 0 ? 0ull : (t < 52u ? rd.vdc[(m - 1u) * 52u + t] : rd.vdc_inv[(m - 1u) * 52u + (t - 52u)]);
        for (uint32_t t = threadIdx.x; t < sob_nd * sob_bits; t += 256u) {
            const uint32_t dd = t % sob_nd;
            tab[t] = rd.sobol32[(dd < 1024u ? dd : 1023u) * 52u + (t / sob_nd)];
        }
        __syncthreads();
        sb.tab = tab; sb.nd = sob_nd; sb.bits = sob_bits; sb.m = vm;
    }
    for (uint32_t base = blockIdx.x * 256u; base < n; base += gridDim.x * 256u) {
        const uint32_t i = base + threadIdx.x;
        uint32_t slot = 0;
        bool shading = false;
        DlHit d;
        if (i < n) {
            slot = queue[i];
            shading = __float_as_uint(dl.le_kind[slot].w) == DL_SHADING;
            if (shading) {
                const float4 hc = pb.hit_cont[slot];
                const float4* rp = reinterpret_cast<const float4*>(pb.ray_cont + slot);
                const float4 r0 = rp[0], r1 = rp[1];
                dl_interaction<F, true>(sc, pb, slot, __float_as_uint(hc.x), hc, f3{r0.w, r1.x, r1.y}, &d);
            }
        }
        // Queue appends, gathered over up to 32 estimates: one atomicAdd per wave, queue and chunk instead of one per estimate.  With an append after every estimate (dl_push)
        // the kernel spent 43 % of its time on them — 10.6 -> 6.1 ms per launch with the appends compiled out (RSPT_DL_EXP=2), against 10.4 without the result stores
        // (profiles/r06_directlighting_experiments.txt): every append is an atomic round trip that the wave waits for with its stores in flight, on one address for the
        // whole chip.  A lane notes what its estimates want in two bit masks; flush() counts the chunk's entries by ballots, claims them at once and writes them estimate by
        // estimate, so that the rays of one estimate — neighbouring virtual slots of a plane — stay neighbours in the queue.
        uint32_t sh_bits = 0, mis_bits = 0, r0 = 0;
        auto flush_one = [&](uint32_t bits, uint32_t cnt, uint32_t tag, uint32_t* __restrict__ q, uint32_t* counter) {
            uint32_t total = 0;   // wave-uniform
            for (uint32_t k = 0; k < cnt; k++) total += (uint32_t)__popcll(__ballot((bits >> k) & 1u));
            if (!total) return;
            const uint32_t lane = __lane_id();
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(counter, total);
            base = __builtin_amdgcn_readfirstlane(base);
            for (uint32_t k = 0; k < cnt; k++) {
                const bool mine = (bits >> k) & 1u;
                const uint64_t m = __ballot(mine);
                if (mine) q[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (slot * dl.vs + (r0 + k) * dl.vr) | tag;
                base += (uint32_t)__popcll(m);
            }
        };
        auto flush = [&](uint32_t cnt) {
            if (RSPT_DL_EXP != 2) {
                flush_one(sh_bits, cnt, 0u, q_any, cnt_any);
                flush_one(mis_bits, cnt, RSPT_Q_MIS, q_mis, cnt_mis);
            }
            sh_bits = mis_bits = 0;
            r0 += cnt;
        };
        uint32_t r = 0;
        for (uint32_t j = 0; j < n_lights_round; j++) {          // (wave-uniform trip counts: flush() ballots)
            const uint32_t n_j = sample_all ? (uint32_t)(nls ? nls[j] : 1) : 1u;
            for (uint32_t kk = 0; kk < n_j; kk++, r++) {
                bool want_sh = false, want_mis = false;
                const uint32_t v = slot * dl.vs + r * dl.vr;
                if (i < n) {
                    uint32_t fl = 0;
                    if (shading) {
                        f2 u_light, u_scatter;
                        uint32_t light_num;
                        float choice_pdf;
                        if (dl_estimate_samples(rd, sb, bt, pb, dl, pix_list, slot, nl, j, kk, n_j, n_arrays, sample_all, &u_light, &u_scatter, &light_num, &choice_pdf))
                            fl = dl_estimate<F>(sc, sb, pb, d, light_num, choice_pdf, u_light, u_scatter, v, &want_sh, &want_mis);
                    }
                    if (RSPT_DL_EXP != 1 || fl == 0x7fffffffu) dl.nflags[v] = fl;
                }
                if (want_sh) sh_bits |= 1u << (r - r0);
                if (want_mis) mis_bits |= 1u << (r - r0);
                if (r - r0 == 31u) flush(32u);
            }
        }
        if (r != r0) flush(r - r0);
    }
}

template <uint32_t F>
__global__ __launch_bounds__(256) void k_dl_nee_all(RSPT_DL_NEE_ARGS) { dl_nee_all<F>(sc, rd, bt, pb, dl, pix_list, queue, count_in, nls, R, n_arrays, sample_all, q_any, cnt_any, q_mis, cnt_mis, sob_nd, sob_bits); }
template <uint32_t F, int W>   // the same built for W waves per SIMD (a register budget of 512 / W; the compiler spills what does not fit)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void k_dl_nee_all_w(RSPT_DL_NEE_ARGS) {
    dl_nee_all<F>(sc, rd, bt, pb, dl, pix_list, queue, count_in, nls, R, n_arrays, sample_all, q_any, cnt_any, q_mis, cnt_mis, sob_nd, sob_bits);
}

RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_dl_nee_resolve_all(SceneDev sc, PathBuf pb, DlBuf dl, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_in,
                                                            const int32_t* __restrict__ nls, uint32_t R, uint32_t n_arrays, uint32_t sample_all) {
    const uint32_t n = *count_in;
    const uint32_t nl = sc.n_lights, n_lights_round = sample_all ? nl : 1u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t slot = queue[i];
        const float4 la = dl.l_all[slot];
        rgb l = rgb{la.x, la.y, la.z};
        const float4 a4 = dl.ld_acc[slot];
        rgb acc = rgb{a4.x, a4.y, a4.z};
        uint32_t r = 0;
        for (uint32_t j = 0; j < n_lights_round; j++) {
            const uint32_t n_j = sample_all ? (uint32_t)(nls ? nls[j] : 1) : 1u;
            for (uint32_t kk = 0; kk < n_j; kk++, r++) {
                const uint32_t v = slot * dl.vs + r * dl.vr;
                const uint32_t fl = dl.nflags[v];
                if (!(fl & 0x100u)) continue;
                const float4 c1 = pb.nee_c1[v], c2 = pb.nee_c2[v];
                rgb ld = mkrgb(0.0f);
                if ((fl & DLF_HAS_C1) && pb.occluded[v] == 0u) ld = ld + rgb{c1.x, c1.y, c1.z};
                if (fl & DLF_HAS_C2) {
                    const float4 hm = pb.hit_mis[v];
                    const uint32_t hp = __float_as_uint(hm.x), light_num = __float_as_uint(c2.w);
                    if (fl & DLF_C2_ON_MISS) {
                        if (hp == RSPT_MISS) ld = ld + rgb{c2.x, c2.y, c2.z};
                    } else if (hp != RSPT_MISS) {
                        const TriRec t = load_tri(sc, hp);
                        if (t.area_light >= 0 && (uint32_t)t.area_light == light_num) {
                            Hit h;
                            tri_fill(sc, hp, t, hm.y, hm.z, hm.w, &h);
                            const float4* mr = reinterpret_cast<const float4*>(pb.ray_mis + v);
                            const float4 m0 = mr[0], m1 = mr[1];
                            if (!is_black(light_l(sc.lights[light_num], h.n, -f3{m0.w, m1.x, m1.y}))) ld = ld + rgb{c2.x, c2.y, c2.z};
                        }
                    }
                }
                if (!sample_all) l = l + ld / c1.w;  // estimate_direct(..) / light_pdf
                else if (dl.kidx[slot] * nl + j < n_arrays / 2u) {
                    acc = acc + ld;
                    if (kk + 1u == n_j) { l = l + acc / (float)n_j; acc = mkrgb(0.0f); }
                } else l = l + ld;
            }
        }
        dl.l_all[slot] = make_float4(l.r, l.g, l.b, 0.0f);
        dl.ld_acc[slot] = make_float4(acc.r, acc.g, acc.b, 0.0f);
    }
}

// li of every node, bottom-up; the root's goes to the camera sample
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_dl_gather(Batch bt, PathBuf pb, DlBuf dl, uint32_t n_lights, uint32_t max_depth) {
    const uint32_t s = blockIdx.x * 256u + threadIdx.x;
    if (s >= bt.n) return;
    for (uint32_t h = dl.H - 1u; h >= 1u; h--) {
        const uint32_t slot = s * dl.H + h;
        const float4 lk = dl.le_kind[slot];
        const uint32_t kind = __float_as_uint(lk.w);
        if (kind == DL_EMPTY) continue;
        rgb l = mkrgb(0.0f);
        if (kind == DL_LEAF) l = rgb{lk.x, lk.y, lk.z};
        else if (kind == DL_SHADING) {
            l = l + rgb{lk.x, lk.y, lk.z};                                   // l += isect.le(&wo)
            if (n_lights) { const float4 la = dl.l_all[slot]; l = l + rgb{la.x, la.y, la.z}; }
            const uint32_t depth = 31u - (uint32_t)__builtin_clz(h);
            if (depth + 1u < max_depth) {
                for (int side = 0; side < 2; side++) {                        // l += specular_reflect(..); l += specular_transmit(..)
                    const float4 w = side ? dl.w_t[slot] : dl.w_r[slot];
                    rgb term = mkrgb(0.0f);
                    const uint32_t child = s * dl.H + 2u * h + (uint32_t)side;
                    if (depth + 1u < dl.levels && __float_as_uint(dl.le_kind[child].w) != DL_EMPTY) {
                        const float4 lc = dl.l_all[child];
                        term = rgb{w.x, w.y, w.z} * rgb{lc.x, lc.y, lc.z} * mkrgb(w.w);  // f * self.li(..) * Spectrum::new(|cos| / pdf)
                    }
                    l = l + term;
                }
            }
        }
        dl.l_all[slot] = make_float4(l.r, l.g, l.b, 0.0f);
    }
    const float4 root = dl.l_all[s * dl.H + 1u];
    pb.L_eta[s] = make_float4(root.x, root.y, root.z, 1.0f);
}

}  // namespace rspt
