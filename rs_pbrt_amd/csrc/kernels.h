// gfx950 kernels of the wavefront path tracer.  One lane = one ray (trace stages) or one path
// (raygen / shade) or one pixel (film); wave64 throughout, no MFMA (pointer chasing + scalar
// f32/f64 arithmetic, HBM/L2-latency bound).  Stage inventory:
//   k_raygen        K1  Sampler::get_camera_sample + PerspectiveCamera::generate_ray_differential
//   k_trace<ANY>    K2/K3  BVHAccel::intersect / intersect_p + Triangle hit test, LDS-staged stack
//   k_shade         K4+K6  PathIntegrator::li body between two intersections, incl. NEE resolve
//   k_film          K8  FilmTile::add_sample / merge_film_tile
//   k_ld_*          K9  light sampling distributions
// Queues are arrays of path slots compacted with wave64 ballot + prefix popcount, one atomic per
// workgroup and queue (K7, in k_shade).
#pragma once
#include "dev_texture.h"
#include "pixel_sampler.h"

namespace rspt {

// ---- per-path state, SoA by path slot ------------------------------------------------------
enum : uint32_t {
    ST_DIM_MASK = 0xffffu,       // next Sobol' dimension
    ST_BOUNCE_SHIFT = 16,        // bounces (8 bits)
    ST_SPECULAR = 1u << 24,      // specular_bounce
    ST_PENDING = 1u << 25,       // a next-event estimate waits for its visibility / MIS ray
    ST_ALIVE = 1u << 26,         // a continuation ray is in flight
    ST_HAS_C1 = 1u << 27,        // pending light-sample term
    ST_HAS_C2 = 1u << 28,        // pending BSDF-sample (MIS) term
    ST_C2_ON_MISS = 1u << 29,    // ... of an infinite light: it counts when the MIS ray escapes
    ST_NO_DIFF = 1u << 30,       // the camera ray went through a null material: it was re-spawned without differentials (path.rs:109-116)
    ST_COMPACT = 1u << 31,       // the pending estimate is kept as ONE vector: what the resolve adds if the shadow ray arrives (shade_path)
};
#define RSPT_Q_MIS 0x80000000u   // closest-hit queue entry flag: this is the path's MIS ray

struct PathBuf {
    rspt_ray* ray_cont;   // continuation ray (o, d, t_max, slot)
    rspt_ray* ray_mis;    // BSDF-sampled MIS ray of estimate_direct (closest hit)
    rspt_ray* ray_sh;     // shadow ray of estimate_direct (any hit)
    float4* hit_cont;     // (prim bits, b0, b1, b2)
    float4* hit_mis;
    uint32_t* occluded;   // shadow ray result
    float4* L_eta;        // (L.rgb, eta_scale)
    float4* beta;         // (beta.rgb, -)
    float4* nee_c1;       // (f*Li*w/light_pdf .rgb, light choice pdf)
    float4* nee_c2;       // (f*L*w/scattering_pdf .rgb, light index bits)
    float4* nee_beta;     // (beta at the time of the estimate .rgb, -)
    uint64_t* sobol_index;
    uint32_t* state;
    float2* p_film;
    float4* tex;          // k_texture results, RSPT_TEX_ROWS rows of tex_stride paths (nullptr: scene without textures)
    uint32_t tex_stride;
    uint32_t* hit_inst;   // continuation ray's hit: 0 or 1 + instance (nullptr: scene without object instances)
    rspt_mat::Built* dyn_built;  // one per thread of the shade launch: where a dynamic material's lobes are built (nullptr: no dynamic material)
    uint32_t dyn_threads;
    float* time;          // Ray.time of the path (perspective.rs:226), kept only while a scene with moving instances is rendered (nullptr otherwise)
    uint32_t fresh;       // 1 in the path integrator's k_raygen and its FIRST shade launch of a batch: every slot's (L, eta_scale) is (0, 0, 0, 1) and beta (1, 1, 1)
                          // by construction, so k_raygen does not write the two 16-byte fields and shade_path does not read them (64 of ~1200 bytes per C3 sample)
    // ---- round 6: MOVING path state (the path integrator's schedule, k_shade<F, MOVE>) ----
    // A path no longer keeps its slot for life.  The index of a live path is its POSITION in the current iteration's active queue: k_shade reads the state at position p
    // and writes what the path carries on (ray, beta, L, state, Sobol' index, its pending estimate, its shadow ray) at the position it gets in the NEXT queue — dense,
    // unit-stride stores, and next iteration's loads are unit-stride too (a 16-byte field read from a sparse slot costs a whole 128-byte line from the fabric:
    // profiles/r06_pmc_calibration.md).  The fields above are the set the iteration READS (the host swaps the two sets between iterations); the o_* fields the set it
    // WRITES.  What stays indexed by the path's ORIGINAL slot: p_film, the final radiance (L_final, what k_film reads) and the general (BSDF-sampled / MIS) form of a
    // pending estimate — ray_mis, hit_mis, nee_c2, nee_beta — which only light-hitting samples and infinite lights use.
    uint32_t move;        // 1: positions (MOVE instantiations); 0: slots for life (every other schedule)
    uint32_t orig_is_p;   // 1 in the first MOVE launch of a batch: the queue still holds original slots (position = slot), pb.orig is not read
    uint32_t* orig;       // original slot of the path at position p (not read in the fresh launch: position = slot there)
    rspt_ray* o_ray_cont; float4* o_L_eta; float4* o_beta; float4* o_nee_c1; uint64_t* o_sobol_index; uint32_t* o_state; uint32_t* o_orig;
    float4* L_final;      // (L.rgb, -) by original slot, written once when the path ends
};

struct QueueCounts {  // one per wavefront iteration
    uint32_t active, closest, any;          // queue lengths
    uint32_t cursor_closest, cursor_any;    // dynamic-fetch cursors of the persistent trace kernel
    uint32_t overflow_closest, overflow_any; // rays handed to k_trace_fixup; each sits 2 words after its cursor
    uint32_t active_tail;                   // paths that only wait for their last next-event estimate: stored from the END of the
                                            // active queue, so that the waves of k_shade are either all-alive or all-short
    uint32_t xcd_closest[8], xcd_any[8];    // k_trace_w4's per-XCD fetch cursors (XCD-affine dealing, trace_w4.h); zeroed with the rest
};

struct Batch {
    uint32_t pix0, n_pix;  // range of the shard's pixel list
    uint32_t s0, ns;       // range of sample indices
    uint32_t n;            // n_pix * ns paths
};

// XCD-aware virtual block index: the dispatcher places block b on XCD b % 8, so give each XCD a
// contiguous eighth of every grid-stride stripe (neighbouring rays share BVH nodes -> same L2).
RDEV uint32_t virtual_block() {
    uint32_t g = gridDim.x, b = blockIdx.x;
    return (g & 7u) ? b : (b & 7u) * (g >> 3) + (b >> 3);
}

// ---- K1 -------------------------------------------------------------------------------------
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_raygen(RenderDev rd, Batch bt, PathBuf pb, const uint32_t* __restrict__ pix_list,
                                                uint32_t* __restrict__ q_active, uint32_t* __restrict__ q_closest, QueueCounts* cnt) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0) { cnt->active = bt.n; cnt->closest = bt.n; cnt->any = 0; }
    // Sobol': the block's copy of the tables a camera sample reads — the two van der Corput matrices of this resolution (index of the
    // sample in its pixel, lowdiscrepancy.rs:1014-1043) and generator-matrix rows 0..4 — so that the ~50 dependent table reads per sample
    // are LDS reads (the kernel took 8 % of a C3 stand-in step reading them through L1 / L2)
    __shared__ uint64_t s_vdc[2][52];
    __shared__ uint32_t s_gen[5][52];
    const bool sobol = rd.sampler_kind != RSPT_SAMPLER_HALTON;
    const uint32_t m = (uint32_t)rd.log2_res;
    if (sobol) {
        for (uint32_t t = threadIdx.x; t < 5u * 52u; t += 256u) s_gen[t / 52u][t % 52u] = rd.sobol32[t];
        if (m > 0u && threadIdx.x < 104u) s_vdc[threadIdx.x / 52u][threadIdx.x % 52u] = (threadIdx.x < 52u ? rd.vdc : rd.vdc_inv)[(m - 1u) * 52u + threadIdx.x % 52u];
        __syncthreads();
    }
    if (i >= bt.n) return;
    // path slot i = pixel-major (slot = pixel * ns + sample): the lanes of a wave start with rays through the same or
    // neighbouring pixels (sample-major slots cost 2 % on C2: less coherent first hits and shadow rays)
    uint32_t k = bt.pix0 + i / bt.ns, s = bt.s0 + i % bt.ns;
    uint32_t pk = pix_list[k];
    int32_t px = (int32_t)(int16_t)(pk & 0xffffu), py = (int32_t)(int16_t)(pk >> 16);
    // get_index_for_sample (sobol.rs:110-117 / halton.rs:173-214), then
    // Sampler::get_camera_sample (sampler.rs:85-95): film 2-D, time 1-D, lens 2-D
    uint64_t index;
    float fx, fy;
    f3 p_lens{0.0f, 0.0f, 0.0f};   // (lens x, lens y, time): the time value only for a moving camera
    if (!sobol) {
        index = halton_index(rd, px, py, (uint64_t)s);
        fy = halton_dim(rd, index, 1); fx = halton_dim(rd, index, 0);
        if (rd.cam_anim || pb.time) p_lens.z = halton_dim(rd, index, 2);
        if (rd.lens_radius > 0.0f) { p_lens.x = halton_dim(rd, index, 3); p_lens.y = halton_dim(rd, index, 4); }
    } else {
        // sobol_interval_to_index / sobol_dim / sobol_pixel_dim (dev_scene.h) over the LDS copies: the same XORs
        index = 0;
        if (m > 0u) {
            index = (uint64_t)s << (m << 1);
            uint64_t delta = 0;
            for (uint64_t f = (uint64_t)s; f != 0; f &= f - 1) delta ^= s_vdc[0][__builtin_ctzll(f)];
            uint64_t b = ((uint64_t)((uint32_t)(px - rd.sample_bounds[0]) << m) | (uint64_t)(int64_t)(py - rd.sample_bounds[1])) ^ delta;
            for (; b != 0; b &= b - 1) index ^= s_vdc[1][__builtin_ctzll(b)];
        }
        auto dim = [&](uint32_t dnum) {
            uint32_t v = 0;
            for (uint64_t a = index; a != 0; a &= a - 1) v ^= s_gen[dnum][__builtin_ctzll(a)];
            return fminf((float)v * 0x1.0p-32f, RSPT_ONE_MINUS_EPS);
        };
        auto pixel_dim = [&](uint32_t dnum, int32_t pix) {
            const float sv = dim(dnum) * (float)rd.resolution + (float)rd.sample_bounds[dnum];
            return clampf(sv - (float)pix, 0.0f, RSPT_ONE_MINUS_EPS);
        };
        fy = pixel_dim(1, py); fx = pixel_dim(0, px);
        if (rd.cam_anim || pb.time) p_lens.z = dim(2);
        if (rd.lens_radius > 0.0f) { p_lens.x = dim(3); p_lens.y = dim(4); }
    }
    f2 p_film{(float)px + fx, (float)py + fy};
    f3 o, d;
    float t_max;
    camera_ray(rd, p_film, p_lens, &o, &d, &t_max);
    rspt_ray r;
    r.o[0] = o.x; r.o[1] = o.y; r.o[2] = o.z; r.d[0] = d.x; r.d[1] = d.y; r.d[2] = d.z; r.t_max = t_max; r.id = i;
    pb.ray_cont[i] = r;
    if (!pb.fresh) {
        pb.L_eta[i] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
        pb.beta[i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
    }
    pb.sobol_index[i] = index;
    pb.state[i] = 5u | ST_ALIVE;  // dimensions 0..4 consumed by the camera sample
    pb.p_film[i] = make_float2(p_film.x, p_film.y);
    if (pb.time) pb.time[i] = rd.shutter_open * (1.0f - p_lens.z) + rd.shutter_close * p_lens.z;   // lerp(sample.time, shutter_open, shutter_close)
    q_active[i] = i;
    q_closest[i] = i;
}

// ---- K2 / K3 --------------------------------------------------------------------------------
#define RSPT_TRACE_BLOCK 256
#define RSPT_LDS_STACK 32  // entries per lane kept in LDS; deeper levels spill to scratch

// Bounds3f::intersect_p with precomputed reciprocal direction (geometry.rs:2211-2269)
RDEV bool box_hit(float4 n0, float4 n1, f3 o, f3 inv, bool ng0, bool ng1, bool ng2, float ray_tmax) {
    const float widen = 1.0f + 2.0f * gamma_n(3);
    float t_min = ((ng0 ? n0.w : n0.x) - o.x) * inv.x;
    float t_max = ((ng0 ? n0.x : n0.w) - o.x) * inv.x;
    float ty_min = ((ng1 ? n1.x : n0.y) - o.y) * inv.y;
    float ty_max = ((ng1 ? n0.y : n1.x) - o.y) * inv.y;
    t_max *= widen;
    ty_max *= widen;
    if (t_min > ty_max || ty_min > t_max) return false;
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_max) t_max = ty_max;
    float tz_min = ((ng2 ? n1.y : n0.z) - o.z) * inv.z;
    float tz_max = ((ng2 ? n0.z : n1.y) - o.z) * inv.z;
    tz_max *= widen;
    if (t_min > tz_max || tz_min > t_max) return false;
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_max) t_max = tz_max;
    return (t_min < ray_tmax) && (t_max > 0.0f);
}

// The alpha tests of Triangle::intersect (triangle.rs:313-330: alpha_mask) and Triangle::intersect_p (:593-655: alpha_mask and
// shadow_alpha_mask, plus the degenerate-triangle rejection that exists only on that branch).  Called for a candidate that passed the
// watertight test on a mesh with MF_ALPHA; the texture is evaluated at the hit's p / uv without ray differentials
// (SurfaceInteraction::new).  false = no hit, and the ray's t_max stays as it was.
template <bool SHADOW>
__device__ __noinline__ bool alpha_pass(const SceneDev& sc, const TexTables& tt, uint32_t pi, f3 p0, f3 p1, f3 p2, float b0, float b1, float b2) {
    const rspt_prim pr = sc.prims[pi];
    const rspt_mesh m = sc.meshes[pr.mesh];
    if (!m.alpha_tex && !(SHADOW && m.shadow_alpha_tex)) return true;
    f2 uv0{0.0f, 0.0f}, uv1{1.0f, 0.0f}, uv2{1.0f, 1.0f};  // triangle.rs:97-112
    if (m.has_uv && sc.UV) {
        uv0 = f2{sc.UV[2 * (size_t)pr.v[0]], sc.UV[2 * (size_t)pr.v[0] + 1]};
        uv1 = f2{sc.UV[2 * (size_t)pr.v[1]], sc.UV[2 * (size_t)pr.v[1] + 1]};
        uv2 = f2{sc.UV[2 * (size_t)pr.v[2]], sc.UV[2 * (size_t)pr.v[2] + 1]};
    }
    if (SHADOW) {  // "the triangle is actually degenerate; the intersection is bogus" (triangle.rs:611-621)
        const f2 duv02{uv0.x - uv2.x, uv0.y - uv2.y}, duv12{uv1.x - uv2.x, uv1.y - uv2.y};
        const f3 dp02 = p0 - p2, dp12 = p1 - p2;
        const float det = duv02.x * duv12.y - duv02.y * duv12.x;
        const bool degenerate = fabsf(det) < 1e-8f;
        f3 dpdu{0.0f, 0.0f, 0.0f}, dpdv{0.0f, 0.0f, 0.0f};
        if (!degenerate) {
            const float invdet = 1.0f / det;
            dpdu = (dp02 * duv12.y - dp12 * duv02.y) * invdet;
            dpdv = (dp02 * -duv12.x + dp12 * duv02.x) * invdet;
        }
        if ((degenerate || len2(cross(dpdu, dpdv)) == 0.0f) && len2(cross(p2 - p0, p1 - p0)) == 0.0f) return false;
    }
    TexSurf su;
    su.p = p0 * b0 + p1 * b1 + p2 * b2;
    su.uv = f2{uv0.x * b0 + uv1.x * b1 + uv2.x * b2, uv0.y * b0 + uv1.y * b1 + uv2.y * b2};
    su.dudx = su.dvdx = su.dudy = su.dvdy = 0.0f;
    su.dpdx = su.dpdy = f3{0.0f, 0.0f, 0.0f};
    if (m.alpha_tex && tex_eval(tt, m.alpha_tex - 1u, su).r == 0.0f) return false;
    if (SHADOW && m.shadow_alpha_tex && tex_eval(tt, m.shadow_alpha_tex - 1u, su).r == 0.0f) return false;
    return true;
}

// The same tests for scenes whose masks all have AlphaMask's form (SceneDev::alpha_masks), in line: no call, no texture-graph interpreter in the
// traversal kernel's register budget (k_trace_w4<.., ALPHA = 1> carries tex_eval's: 178 - 200 VGPRs + 624 B of scratch = 2 waves / SIMD).
// Operation for operation what alpha_pass -> tex_eval -> img_lookup -> img_triangle(level 0) computes for the first channel.
RDEV float alpha_mask_value(const AlphaMask& m, const float* __restrict__ texel_pool, f2 uv) {
    if (m.kind == 1u) return m.value;
    const f2 st{uv.x * m.su + m.du, uv.y * m.sv + m.dv};
    const uint32_t w = m.width ? m.width : 1u, h = m.height ? m.height : 1u;
    const float s = st.x * (float)w - 0.5f, t = st.y * (float)h - 0.5f;
    const int64_t s0 = f2i64(floorf(s)), t0 = f2i64(floorf(t));
    const float ds = s - (float)s0, dt = t - (float)t0;
    auto texel = [&](int64_t si, int64_t ti) {
        uint64_t ss, tt;
        if (m.wrap == RSPT_WRAP_REPEAT) { ss = (uint64_t)si % (uint64_t)w; tt = (uint64_t)ti % (uint64_t)h; }
        else {
            ss = (uint64_t)(si < 0 ? 0 : (si > (int64_t)w - 1 ? (int64_t)w - 1 : si));
            tt = (uint64_t)(ti < 0 ? 0 : (ti > (int64_t)h - 1 ? (int64_t)h - 1 : ti));
        }
        return texel_pool[m.base + (size_t)m.channels * (tt * w + ss)];
    };
    const float tmp1 = texel(s0 + 1, t0 + 1) * (ds * dt);
    const float tmp2 = texel(s0 + 1, t0) * (ds * (1.0f - dt));
    const float tmp3 = texel(s0, t0 + 1) * ((1.0f - ds) * dt);
    const float tmp4 = texel(s0, t0) * ((1.0f - ds) * (1.0f - dt));
    return tmp4 + tmp3 + tmp2 + tmp1;
}
template <bool SHADOW>
RDEV bool alpha_simple(const SceneDev& sc, const TexTables& tt, uint32_t pi, uint32_t flags, f3 p0, f3 p1, f3 p2, float b0, float b1, float b2) {
    const AlphaEntry& e = sc.alpha_masks[flags >> MF_MASK_SHIFT];
    const AlphaMask ma = e.alpha;
    const uint32_t shadow_kind = SHADOW ? e.shadow.kind : 0u;
    if (!ma.kind && !shadow_kind) return true;
    f2 uv0{0.0f, 0.0f}, uv1{1.0f, 0.0f}, uv2{1.0f, 1.0f};  // triangle.rs:97-112
    if ((flags & MF_HAS_UV) && sc.UV) {   // (the scene has its per-primitive copies: rspt_scene_create builds alpha_masks only then)
        const float4* q = sc.tri_nuv + 5 * (size_t)pi;
        const float4 q2 = q[2], q3 = q[3];
        uv0 = f2{q2.y, q2.z}; uv1 = f2{q2.w, q3.x}; uv2 = f2{q3.y, q3.z};
    }
    if (SHADOW) {  // "the triangle is actually degenerate; the intersection is bogus" (triangle.rs:611-621)
        const f2 duv02{uv0.x - uv2.x, uv0.y - uv2.y}, duv12{uv1.x - uv2.x, uv1.y - uv2.y};
        const f3 dp02 = p0 - p2, dp12 = p1 - p2;
        const float det = duv02.x * duv12.y - duv02.y * duv12.x;
        const bool degenerate = fabsf(det) < 1e-8f;
        f3 dpdu{0.0f, 0.0f, 0.0f}, dpdv{0.0f, 0.0f, 0.0f};
        if (!degenerate) {
            const float invdet = 1.0f / det;
            dpdu = (dp02 * duv12.y - dp12 * duv02.y) * invdet;
            dpdv = (dp02 * -duv12.x + dp12 * duv02.x) * invdet;
        }
        if ((degenerate || len2(cross(dpdu, dpdv)) == 0.0f) && len2(cross(p2 - p0, p1 - p0)) == 0.0f) return false;
    }
    const f2 uv{uv0.x * b0 + uv1.x * b1 + uv2.x * b2, uv0.y * b0 + uv1.y * b1 + uv2.y * b2};
    if (ma.kind && alpha_mask_value(ma, tt.texel_pool, uv) == 0.0f) return false;
    if (SHADOW && shadow_kind) { const AlphaMask ms = e.shadow; if (alpha_mask_value(ms, tt.texel_pool, uv) == 0.0f) return false; }
    return true;
}

struct TraceResult {
    uint32_t prim;
    float t, b0, b1, b2;
    uint32_t nodes, tris;
    uint32_t inst;  // 0: a primitive of the scene's own aggregate; k + 1: inside instance k (prim = the object's primitive, t in the object ray's parameter)
    float t_end;    // INST, closest hit: ray.t_max when the traversal ended — on a miss it can be shorter than it started: an identity instance
                    // shrinks it without reporting its hit (Q10), and VolPathIntegrator::li samples the medium up to ray.t_max (volpath.rs:96-101)
};

// BVHAccel::intersect (bvh.rs:401-462) / intersect_p (:463-514): ordered depth-first traversal,
// near child first by dir_is_neg[axis], far child pushed.  Visit order is exactly the
// reference's, so the winning primitive and (t, b) are bit-identical, ties included.
// INST: a leaf primitive may be a TransformedPrimitive (primitive.rs:216-265): the ray goes through the instance's
// world-to-object transform, the object's own aggregate (or single primitive) is traversed with the same loop and stack, and
// the walk returns to the remaining primitives of the leaf.  Quirks Q10 / Q11 are reproduced unless
// sc.inst_fixed: an identity instance shrinks t_max without reporting its hit, and its interaction survives only if some
// other primitive of the top-level aggregate reports a hit (`hit` below is BVHAccel::intersect's flag, `res` its isect).
// ANIM: the scene has moving instances (their Transform is interpolated at the ray's time, dev_scene.h inst_at) — its own instantiation, so that the kernels
// of every other instanced scene keep their register budget (with the interpolation inlined k_trace_fixup<.., INST> went from 97 to 172 VGPRs)
template <bool ANY, bool INST, bool ALPHA, int STRIDE = RSPT_TRACE_BLOCK /* words between two levels of the LDS stack = columns (threads per block) */, bool ANIM = false>
RDEV TraceResult traverse(const SceneDev& sc, const TexTables& tt, f3 o, f3 d, float t_max, uint32_t* lds_stack /* this lane's column */, float time = 0.0f /* Ray.time: moving instances */) {
    TraceResult res;
    res.prim = RSPT_MISS; res.t = 0.0f; res.b0 = res.b1 = res.b2 = 0.0f; res.nodes = 0; res.tris = 0; res.inst = 0; res.t_end = t_max;
    if (sc.n_nodes == 0) return res;
    f3 inv{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
    bool ng0 = inv.x < 0.0f, ng1 = inv.y < 0.0f, ng2 = inv.z < 0.0f;
    RayShear rs = ray_shear(d);
    uint32_t spill[64 - RSPT_LDS_STACK];
    uint32_t sp = 0, cur = 0;
    uint32_t leaf_i = 0, leaf_end = 0;  // primitives of the current leaf still to test
    // instance state (INST only)
    const f3 w_o = o, w_d = d;
    uint32_t inst = RSPT_MISS, sp_base = 0, w_leaf_i = 0, w_leaf_end = 0;
    float w_tmax = 0.0f;
    bool hit = false, inst_hit = false, inst_ident = false;
    for (;;) {
        if (leaf_i < leaf_end) {
            const uint32_t pi = leaf_i++;
            float4 a = sc.tris[3 * (size_t)pi], b = sc.tris[3 * (size_t)pi + 1], c = sc.tris[3 * (size_t)pi + 2];
            res.tris++;
            if (INST && (__float_as_uint(c.w) & MF_INSTANCE)) {  // TransformedPrimitive::intersect / intersect_p
                inst = __float_as_uint(a.x);
                const InstDev& in = sc.inst[inst];
                w_leaf_i = leaf_i; w_leaf_end = leaf_end; w_tmax = t_max; sp_base = sp; inst_hit = false;
                if (ANIM && in.anim != RSPT_MISS) {   // a moving instance: primitive_to_world.interpolate(r.time) and its inverse (primitive.rs:218-222)
                    const InstDev at = inst_at(sc, inst, time);
                    inst_ident = at.identity != 0u;
                    inst_ray(at, w_o, w_d, t_max, &o, &d, &t_max);
                } else {
                    inst_ident = in.identity != 0u;
                    inst_ray(in, w_o, w_d, t_max, &o, &d, &t_max);
                }
                inv = f3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
                ng0 = inv.x < 0.0f; ng1 = inv.y < 0.0f; ng2 = inv.z < 0.0f;
                rs = ray_shear(d);
                if (in.root_node != RSPT_MISS) { cur = in.root_node; leaf_i = leaf_end = 0; }
                else { cur = RSPT_MISS; leaf_i = in.first_prim; leaf_end = leaf_i + 1u; }  // a lone GeometricPrimitive: no box test
                continue;
            }
            float t, b0, b1, b2;
            if (tri_test(f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c.x}, o, rs, t_max, &t, &b0, &b1, &b2)) {
                if (ALPHA && (__float_as_uint(c.w) & MF_ALPHA) && !alpha_pass<ANY>(sc, tt, pi, f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c.x}, b0, b1, b2)) continue;
                if (ANY) { res.prim = 0; return res; }
                t_max = t;  // GeometricPrimitive::intersect shrinks the ray (primitive.rs:155)
                res.prim = pi; res.t = t; res.b0 = b0; res.b1 = b1; res.b2 = b2;
                if (INST && inst != RSPT_MISS) { res.inst = inst + 1u; inst_hit = true; }
                else { res.inst = 0; hit = true; }
            }
            continue;
        }
        if (cur == RSPT_MISS) {  // pop
            if (INST && inst != RSPT_MISS && sp == sp_base) {  // the object's traversal is over: back to world space
                if (inst_hit) { if (sc.inst_fixed || !inst_ident) hit = true; }  // primitive.rs:226-253: r.t_max.set(ray.t_max) either way
                else t_max = w_tmax;
                o = w_o; d = w_d;
                inv = f3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
                ng0 = inv.x < 0.0f; ng1 = inv.y < 0.0f; ng2 = inv.z < 0.0f;
                rs = ray_shear(d);
                inst = RSPT_MISS; sp_base = 0;
                leaf_i = w_leaf_i; leaf_end = w_leaf_end;
                continue;
            }
            if (sp == 0) break;
            sp--;
            cur = sp < RSPT_LDS_STACK ? lds_stack[sp * STRIDE] : spill[sp - RSPT_LDS_STACK];
        }
        float4 n0 = sc.nodes[2 * (size_t)cur], n1 = sc.nodes[2 * (size_t)cur + 1];
        res.nodes++;
        const uint32_t here = cur;
        cur = RSPT_MISS;
        if (box_hit(n0, n1, o, inv, ng0, ng1, ng2, t_max)) {
            uint32_t w = __float_as_uint(n1.w);
            uint32_t n_prims = w & 0xffffu, axis = (w >> 16) & 0xffu;
            uint32_t offset = __float_as_uint(n1.z);
            if (n_prims > 0) {
                leaf_i = offset; leaf_end = offset + n_prims;
            } else {
                bool neg = axis == 0 ? ng0 : (axis == 1 ? ng1 : ng2);
                uint32_t far_child = neg ? here + 1 : offset;
                cur = neg ? offset : here + 1;
                if (sp < RSPT_LDS_STACK) lds_stack[sp * STRIDE] = far_child;
                else spill[sp - RSPT_LDS_STACK] = far_child;
                sp++;
            }
        }
    }
    if (INST && !ANY && !hit) { res.prim = RSPT_MISS; res.inst = 0; res.t = res.b0 = res.b1 = res.b2 = 0.0f; }  // BVHAccel::intersect returns `hit`, not "isect was written"
    res.t_end = t_max;
    return res;
}

// out_mode 0: float4 (prim, b0, b1, b2) into hit_cont/hit_mis by slot; 1: rspt_hit AoS by queue
// position (stage hook); ANY: occluded[slot] (mode 0) or rspt_hit.prim (mode 1).
// out_inst (INST, closest hit, continuation rays only): 0 or 1 + instance of the hit, by slot
template <bool ANY, int OUT_MODE, bool COUNT, bool INST, bool ALPHA, bool ANIM = false>
__global__ __launch_bounds__(RSPT_TRACE_BLOCK) void k_trace(SceneDev sc, TexTables tt, const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_ptr,
                                                            uint32_t count_imm, const rspt_ray* __restrict__ rays_a, const rspt_ray* __restrict__ rays_b,
                                                            float4* __restrict__ out_a, float4* __restrict__ out_b, uint32_t* __restrict__ out_occ,
                                                            rspt_hit* __restrict__ out_hits, unsigned long long* __restrict__ counters, uint32_t* __restrict__ out_inst) {
    __shared__ uint32_t stack[RSPT_LDS_STACK * RSPT_TRACE_BLOCK];
    const uint32_t n = count_ptr ? *count_ptr : count_imm;
    const uint32_t stride = gridDim.x * RSPT_TRACE_BLOCK;
    unsigned long long c_nodes = 0, c_tris = 0;
    for (uint32_t base = virtual_block() * RSPT_TRACE_BLOCK; base < n; base += stride) {
        uint32_t i = base + threadIdx.x;
        if (i >= n) continue;
        uint32_t e = queue ? queue[i] : i;
        uint32_t slot = e & ~RSPT_Q_MIS;
        bool mis = (e & RSPT_Q_MIS) != 0;
        const float4* rp = reinterpret_cast<const float4*>((mis ? rays_b : rays_a) + slot);
        float4 r0 = rp[0], r1 = rp[1];
        const float time = (INST && OUT_MODE == 0 && sc.ray_time) ? sc.ray_time[slot / sc.time_div] : 0.0f;
        TraceResult res = traverse<ANY, INST, ALPHA, RSPT_TRACE_BLOCK, ANIM>(sc, tt, f3{r0.x, r0.y, r0.z}, f3{r0.w, r1.x, r1.y}, r1.z, stack + threadIdx.x, time);
        if (OUT_MODE == 0) {
            if (ANY) out_occ[slot] = res.prim != RSPT_MISS ? 1u : 0u;
            else {
                (mis ? out_b : out_a)[slot] = make_float4(__uint_as_float(res.prim), (INST && res.prim == RSPT_MISS) ? res.t_end : res.b0, res.b1, res.b2);  // a miss: .y = the ray's final t_max
                if (INST && !mis && out_inst) out_inst[slot] = res.inst;
            }
        } else {
            rspt_hit h;
            h.prim = res.prim; h.t = res.t; h.b0 = res.b0; h.b1 = res.b1; h.b2 = res.b2;
            out_hits[i] = h;
        }
        if (COUNT) { c_nodes += res.nodes; c_tris += res.tris; }
    }
    if (COUNT) {
        atomicAdd(&counters[0], c_nodes);
        atomicAdd(&counters[1], c_tris);
    }
}

// ---- K4 + K6 --------------------------------------------------------------------------------
#ifndef RSPT_MIS_EARLY_OUT
#define RSPT_MIS_EARLY_OUT 1   // A/B knob (tools/ab_build.sh AB_DEFS=-DRSPT_MIS_EARLY_OUT=0): the light-triangle test of estimate_direct's BSDF-sampled term before the lobes' values
#endif
struct ShadeOut {
    bool active, cont, mis, shadow;
};
// MOVE: what shade_path hands back instead of storing it at the path's own index — the kernel stores it at the path's next position (shade_kernel)
struct ShadeMove {
    f3 o, d;              // continuation ray (cont)
    rgb beta;
    float4 L;             // (L.rgb, eta_scale)
    float4* s_c1;         // this thread's LDS cells: the pending estimate's nee_c1 record (ST_PENDING in st) and the shadow ray (o.xyz, d.x | d.y, d.z) — staged there, not in
    float4* s_sh0;        // registers: both are made in the next-event block and would stay live through the continuation sampling, the kernel's register peak
    float2* s_sh1;
    uint32_t st;          // packed state
    bool valid;           // the lane held a path
    // (the Sobol' index and the original slot are NOT carried here: shade_kernel reads them again at the commit — two L1 / L2 hits against three registers held
    //  through the whole of shade_path, which runs at its register ceiling)
};

RDEV void store_ray(rspt_ray* dst, f3 o, f3 d, float t_max, uint32_t id) {
    float4* p = reinterpret_cast<float4*>(dst);
    p[0] = make_float4(o.x, o.y, o.z, d.x);
    p[1] = make_float4(d.y, d.z, t_max, __uint_as_float(id));
}

// A dynamic material's Bsdf.bxdfs at this hit: the raw values the texture stage left in the path's rows go through the parameter clamps
// and build_part — the function the host folds constant materials with (material_assembly.h).  Returns the thread's Built record.
RDEVN const rspt_mat::Built* dynamic_lobes(const rspt_mat::DynMaterial& dm, const float4* rows, size_t stride, bool allow_multiple_lobes, rspt_mat::Built* out) {
    using namespace rspt_mat;
    out->n = 0u; out->overflow = 0u; out->shape_varies = 0u; out->eta = 1.0f;
    auto fetch = [&](const DynParam& q, uint32_t id) {
        Param r;
        r.present = q.present; r.tex = 0u; r.v[0] = q.v[0]; r.v[1] = q.v[1]; r.v[2] = q.v[2];
        if (q.row) { const float4 v = rows[(size_t)(RSPT_TEX_ROWS + q.row - 1u) * stride]; r = param_value(id, v.x, v.y, v.z); }
        return r;
    };
    // a part under a mix is scaled by its parent's s1 = clamp(amount) or s2 = clamp(1 - s1) (mixmat.rs:52-56).  The scale is computed into
    // scalars and copied into one array: a choice between two arrays by pointer — `k ? s2 : s1` — came out as s2 for BOTH sides in the
    // per-lane kernels (every m1 lobe scaled by 1 - amount; the wavefront instantiation of the same source was right), found by the linearity
    // of the frame in the scales
    for (uint32_t k = 0; k < dm.n_parts; k++) {
        const DynPart& part = dm.part[k];
        Param p[P_COUNT];
#pragma unroll 1
        for (uint32_t i = 0; i < P_COUNT; i++) p[i] = fetch(part.p[i], i);
        float side[3] = {0.0f, 0.0f, 0.0f};
        if (part.side) {
            const Param am = fetch(part.amount, P_KD);  // a colour: clamp(0, inf)
            const bool s2 = part.side == 2u;
            side[0] = s2 ? clamp0(1.0f - am.v[0]) : am.v[0]; side[1] = s2 ? clamp0(1.0f - am.v[1]) : am.v[1]; side[2] = s2 ? clamp0(1.0f - am.v[2]) : am.v[2];
        }
        build_part(part.kind, p, part.remap != 0u, allow_multiple_lobes, part.side ? side : nullptr, part.second != 0u, out);
    }
    return out;
}

// One step of PathIntegrator::li (path.rs:91-280) for path slot p: fold in the previous
// bounce's next-event estimate, then process the hit of the continuation ray.
// F: the feature set the instantiation is compiled for (dev_bsdf.h SF_*): everything a scene outside F could bring folds away
// MOVE (round 6, PathBuf::move): p is the path's POSITION in this iteration's queue; nothing the path carries on is stored here — it comes back in *mv and
// shade_kernel stores it at the path's next position.  Indexed by the original slot `og` instead: the general form of a pending estimate.
template <bool PIX, uint32_t F = SF_ALL, bool MOVE = false>
RDEVN ShadeOut shade_path(const SceneDev& sc, const LightDistDev& ld, const RenderDev& rd, const PathBuf& pb, uint32_t p, unsigned long long* stats,
                          const uint32_t* __restrict__ sob_tab, uint32_t sob_nd, PixSampler* px, ShadeMove* mv = nullptr) {
    ShadeOut out{false, false, false, false};
    uint32_t st = pb.state[p];
    const bool fresh = !PIX && pb.fresh != 0u;   // the first shade launch of a path-integrator batch (PathBuf::fresh)
    float4 le = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
    if (!fresh) le = pb.L_eta[p];
    rgb L{le.x, le.y, le.z};
    float eta_scale = le.w;
    uint32_t og = p;      // where the slot-for-life arrays of this path live
    if (MOVE) {
        if (!pb.orig_is_p) og = pb.orig[p];
        mv->valid = true;
    }

    if ((st & ST_PENDING) && (st & ST_COMPACT)) {   // the estimate without a BSDF-sampled term (almost all of them): nee_c1 holds beta * ((0 + c1) / pdf), already
        const float4 k = pb.nee_c1[p];                // formed with the operations of the general resolve below; a blocked shadow ray adds beta * (0 / pdf) = +-0
        if (pb.occluded[p] == 0u) L = L + rgb{k.x, k.y, k.z};
        st &= ~(ST_PENDING | ST_HAS_C1 | ST_COMPACT);
    }
    if (st & ST_PENDING) {  // tail of estimate_direct (integrator.rs:461-568) + path.rs:126-139
        float4 c1 = pb.nee_c1[p], c2 = pb.nee_c2[og], nb = pb.nee_beta[og];
        rgb ldir = mkrgb(0.0f);
        if ((st & ST_HAS_C1) && pb.occluded[p] == 0u) ldir = ldir + rgb{c1.x, c1.y, c1.z};
        if (st & ST_HAS_C2) {
            float4 hm = pb.hit_mis[og];
            uint32_t hp = __float_as_uint(hm.x);
            uint32_t light_num = __float_as_uint(c2.w);
            if ((F & SF_L_INFINITE) && (st & ST_C2_ON_MISS)) {  // InfiniteAreaLight: li = light.le(ray) when nothing was hit (integrator.rs:561-563)
                if (hp == RSPT_MISS) ldir = ldir + rgb{c2.x, c2.y, c2.z};
            } else if (hp != RSPT_MISS) {
                TriRec t = load_tri(sc, hp);
                if (t.area_light >= 0 && (uint32_t)t.area_light == light_num) {
                    Hit h;
                    tri_fill<(F & SF_VERTEX) != 0>(sc, hp, t, hm.y, hm.z, hm.w, &h);
                    const float4* mr = reinterpret_cast<const float4*>(pb.ray_mis + og);
                    float4 m0 = mr[0], m1 = mr[1];
                    f3 wi{m0.w, m1.x, m1.y};
                    rgb li = light_l(sc.lights[light_num], h.n, -wi);
                    if (!is_black(li)) ldir = ldir + rgb{c2.x, c2.y, c2.z};
                }
            }
        }
        L = L + rgb{nb.x, nb.y, nb.z} * (ldir / c1.w);
        st &= ~(ST_PENDING | ST_HAS_C1 | ST_HAS_C2 | ST_C2_ON_MISS);
    }
    if (!(st & ST_ALIVE)) {
        if (MOVE) { mv->L = make_float4(L.r, L.g, L.b, eta_scale); mv->st = st; return out; }
        pb.L_eta[p] = make_float4(L.r, L.g, L.b, eta_scale);
        pb.state[p] = st;
        return out;
    }
    st &= ~ST_ALIVE;

    float4 hc = pb.hit_cont[p];
    uint32_t prim = __float_as_uint(hc.x);
    uint32_t bounces = (st >> ST_BOUNCE_SHIFT) & 0xffu;
    if (prim == RSPT_MISS) {  // K5: escaped path picks up the infinite lights (path.rs:267-277)
        if ((F & SF_L_INFINITE) && sc.n_infinite && (bounces == 0 || (st & ST_SPECULAR))) {
            const float4* rp = reinterpret_cast<const float4*>(pb.ray_cont + p);
            float4 r0 = rp[0], r1 = rp[1];
            f3 ray_d{r0.w, r1.x, r1.y};
            float4 bb = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
            if (!fresh) bb = pb.beta[p];
            rgb beta{bb.x, bb.y, bb.z};
            for (uint32_t i = 0; i < sc.n_infinite; i++) L = L + beta * infinite_le(sc, sc.lights[sc.infinite_lights[i]], ray_d);
        }
    } else {
        const float4* rp = reinterpret_cast<const float4*>(pb.ray_cont + p);
        float4 r0 = rp[0], r1 = rp[1];
        f3 ray_d{r0.w, r1.x, r1.y};
        float4 bb = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        if (!fresh) bb = pb.beta[p];
        rgb beta{bb.x, bb.y, bb.z};
        TriRec tri = load_tri(sc, prim);
        Hit h;
        tri_fill<(F & SF_VERTEX) != 0>(sc, prim, tri, hc.y, hc.z, hc.w, &h);
        f3 wo = -ray_d;  // SurfaceInteraction.wo, not normalised (triangle.rs:334)
        if ((F & SF_INST) && pb.hit_inst) {  // the hit lies inside an object instance: TransformedPrimitive::intersect (primitive.rs:216-253)
            const uint32_t hi = pb.hit_inst[p];
            const bool moving = (F & SF_ANIM) && hi && sc.inst[hi - 1u].anim != RSPT_MISS;   // (rare: the interpolated Transform of the path's time, as the traversal used it)
            InstDev moved;
            if (moving) moved = inst_at(sc, hi - 1u, sc.ray_time ? sc.ray_time[p] : 0.0f);
            const InstDev& in_ref = moving ? moved : sc.inst[hi ? hi - 1u : 0u];
            if (hi && !in_ref.identity) {
                const InstDev& in = in_ref;
                inst_hit(in, &h);  // transform_surface_interaction (transform.rs:815-860)
                wo = normalize(xf_vector(in.m, -xf_vector(in.mi, ray_d)));  // wo = -(object ray).d, transformed back and normalised
                if (!sc.inst_fixed) { h.material = 0xffffffffu; h.area_light = -1; }  // ret.primitive = None (Q11): no material, no Le
            }
        }
        if (bounces == 0 || (st & ST_SPECULAR)) {  // path.rs:97-101
            rgb e = h.area_light >= 0 ? light_l(sc.lights[h.area_light], h.n, wo) : mkrgb(0.0f);
            L = L + beta * e;
        }
        if (bounces < rd.max_depth) {  // path.rs:103
            if ((F & SF_NULL) && h.material == 0xffffffffu) {  // null BSDF: pass straight through (path.rs:109-116)
                f3 o = offset_ray_origin(h.p, h.p_err, h.n, ray_d);
                if (MOVE) { mv->o = o; mv->d = ray_d; mv->beta = beta; }
                else {
                    store_ray(pb.ray_cont + p, o, ray_d, RSPT_INF, p);
                    if (fresh) pb.beta[p] = make_float4(beta.r, beta.g, beta.b, 0.0f);   // (k_raygen left it unwritten, and the path goes on)
                }
                st |= ST_ALIVE | ST_NO_DIFF;
                out.cont = true;
            } else {
                rspt_material mat = sc.materials[h.material];
                Bsdf bsdf;  // Bsdf::new (reflection.rs:235-245)
                bsdf.eta = mat.eta;
                bsdf.lt = LobeTex{nullptr, 0};
                bsdf.dropped = 0u;
                const rspt_bxdf* lobes = sc.bxdfs + mat.first_bxdf;
                uint32_t n_lobes = mat.n_bxdfs;
                if ((F & SF_TEX) && sc.mat_flags && sc.mat_flags[h.material]) {  // textured material: k_texture ran for this hit
                    const float4* tb = pb.tex + p;
                    if ((F & SF_DYNAMIC) && (sc.mat_flags[h.material] & RSPT_MAT_DYNAMIC)) {  // the lobe list itself depends on texture values at the hit
                        const rspt_mat::Built* bl = dynamic_lobes(sc.dyn[h.material], tb, pb.tex_stride, true /* path.rs:108 */, pb.dyn_built + (blockIdx.x * blockDim.x + threadIdx.x));
                        lobes = bl->l; n_lobes = bl->n;
                        bsdf.eta = bl->eta;
                    }
                    bsdf.lt = LobeTex{tb, pb.tex_stride};
                    const float4 m4 = tb[4 * (size_t)pb.tex_stride];
                    const uint32_t tf = __float_as_uint(m4.w);
                    bsdf.dropped = (tf >> 8) & 0xffu;
                    if (tf & 1u) {  // Material::bump replaced the shading geometry (material.rs:116-219)
                        const float4 d4 = tb[5 * (size_t)pb.tex_stride];
                        h.sh_n = f3{m4.x, m4.y, m4.z};
                        h.sh_dpdu = f3{d4.x, d4.y, d4.z};
                    }
                }
                bsdf.ss = normalize(h.sh_dpdu);
                bsdf.ns = h.sh_n;
                bsdf.ng = h.n;
                bsdf.ts = cross(h.sh_n, bsdf.ss);
                bsdf.lobes = lobes;
                bsdf.n = n_lobes < 8u ? n_lobes : 8u;
                typename ShadeSamplerFor<PIX, F>::type smp;
                smp.bind(px);
                smp.start(rd, sob_tab, sob_nd, pb.sobol_index[p], st & ST_DIM_MASK);
                if (stats) atomicAdd(&stats[0], 1ull);

                // ---- uniform_sample_one_light (integrator.rs:359-403) ----
                const uint32_t nonspec = BX_ALL & ~BX_SPEC;
                if (sc.n_lights > 0 && bsdf.num_components(nonspec) > 0) {
                    uint32_t vox = light_row(ld, light_voxel(sc, ld, h.p));
                    float pdf_choice = 0.0f;
                    uint32_t light_num = sample_discrete(ld.func + (size_t)vox * sc.n_lights, ld.cdf + (size_t)vox * (sc.n_lights + 1),
                                                         ld.func_int[vox], sc.n_lights, smp.get_1d(rd), &pdf_choice);
                    if (pdf_choice != 0.0f) {
                        f2 u_light = smp.get_2d(rd);
                        f2 u_scatter = smp.get_2d(rd);
                        const rspt_light lt = sc.lights[light_num];
                        rgb c1 = mkrgb(0.0f), c2 = mkrgb(0.0f);
                        // light sample (integrator.rs:424-477)
                        f3 wi{0.0f, 0.0f, 0.0f};
                        float light_pdf = 0.0f, scattering_pdf = 0.0f;
                        LightSample ls;
                        rgb li = light_sample_li<F>(sc, lt, h.p, u_light, &wi, &light_pdf, &ls);
                        if (light_pdf > 0.0f && !is_black(li)) {
                            rgb f = bsdf.template f<F>(wo, wi, nonspec) * mkrgb(absdot(wi, h.sh_n));
                            scattering_pdf = bsdf.template pdf<F>(wo, wi, nonspec);
                            if (!is_black(f)) {
                                // VisibilityTester::unoccluded -> spawn_ray_to (interaction.rs:81-94)
                                f3 origin = offset_ray_origin(h.p, h.p_err, h.n, ls.p - h.p);
                                f3 target = offset_ray_origin(ls.p, ls.p_err, ls.n, origin - ls.p);
                                if (MOVE) { const f3 sd = target - origin; *mv->s_sh0 = make_float4(origin.x, origin.y, origin.z, sd.x); *mv->s_sh1 = make_float2(sd.y, sd.z); }
                                else store_ray(pb.ray_sh + p, origin, target - origin, 1.0f - RSPT_SHADOW_EPS, p);
                                out.shadow = true;
                                if (light_is_delta<F>(lt)) c1 = f * li / light_pdf;  // integrator.rs:470-471
                                else c1 = f * li * mkrgb(power_heuristic(light_pdf, scattering_pdf)) / light_pdf;
                                st |= ST_HAS_C1;
                            }
                        }
                        // BSDF sample with MIS (integrator.rs:480-568), area lights only; sampled_type sentinel 0 (Q6)
                        if (!light_is_delta<F>(lt)) {
                            uint32_t sampled_type = 0;
                            // An area light's BSDF-sampled term exists only when the sampled direction meets the light's own triangle (pdf_li = 0 otherwise, and with it the
                            // whole term: integrator.rs:520-566): that test needs the direction alone, so it runs as soon as the lobe has chosen one, and a miss — all but a
                            // few in 10^4 samples — skips the other lobes' pdfs and the sum of the lobes' values.  Nothing a miss would have computed is read afterwards.
                            const bool area_mis = !((F & SF_L_INFINITE) && lt.kind == RSPT_LIGHT_INFINITE) && RSPT_MIS_EARLY_OUT;
                            TriRec lt_tri;
                            float t_l = 0.0f, lb0 = 0.0f, lb1 = 0.0f, lb2 = 0.0f;
                            bool on_light = false;
                            if (area_mis) lt_tri = load_tri(sc, lt.prim);
                            rgb f = bsdf.template sample_f_if<F>(wo, &wi, u_scatter, &scattering_pdf, nonspec, &sampled_type, [&](f3 w) {
                                if (!area_mis) return false;
                                on_light = tri_test(lt_tri.p0, lt_tri.p1, lt_tri.p2, offset_ray_origin(h.p, h.p_err, h.n, w), ray_shear(w), RSPT_INF, &t_l, &lb0, &lb1, &lb2);
                                return !on_light;
                            });
                            f = f * mkrgb(absdot(wi, h.sh_n));
                            if (!is_black(f) && scattering_pdf > 0.0f) {
                                f3 ro = offset_ray_origin(h.p, h.p_err, h.n, wi);
                                float lpdf = 0.0f;
                                rgb le_mis = ldrgb(lt.L);
                                if ((F & SF_L_INFINITE) && lt.kind == RSPT_LIGHT_INFINITE) {  // InfiniteAreaLight::pdf_li; Le is known from the direction alone
                                    lpdf = infinite_pdf_li(sc, lt, wi);
                                    if (lpdf != 0.0f) le_mis = infinite_le(sc, lt, wi);
                                } else {  // DiffuseAreaLight::pdf_li -> Triangle::pdf_with_ref_point (triangle.rs:745-764)
                                    if (!area_mis) {
                                        lt_tri = load_tri(sc, lt.prim);
                                        on_light = tri_test(lt_tri.p0, lt_tri.p1, lt_tri.p2, ro, ray_shear(wi), RSPT_INF, &t_l, &lb0, &lb1, &lb2);
                                    }
                                    if (on_light) {
                                        Hit lh;
                                        tri_fill<(F & SF_VERTEX) != 0>(sc, lt.prim, lt_tri, lb0, lb1, lb2, &lh);
                                        lpdf = dist2(h.p, lh.p) / (absdot(lh.n, -wi) * tri_area(lt_tri));
                                        if (__builtin_isinf(lpdf)) lpdf = 0.0f;
                                    }
                                }
                                if (lpdf != 0.0f) {
                                    float weight = power_heuristic(scattering_pdf, lpdf);
                                    c2 = f * le_mis * mkrgb(1.0f) * weight / scattering_pdf;
                                    if (!((F & SF_L_INFINITE) && lt.kind == RSPT_LIGHT_INFINITE) || !is_black(le_mis)) {
                                        store_ray(pb.ray_mis + og, ro, wi, RSPT_INF, og);
                                        out.mis = true;
                                        st |= ST_HAS_C2 | (((F & SF_L_INFINITE) && lt.kind == RSPT_LIGHT_INFINITE) ? ST_C2_ON_MISS : 0u);
                                    }
                                }
                            }
                        }
                        // What the resolve at the top of this function will add (path.rs:126-139): beta * (ldir / pdf) with ldir = 0 [+ c1 if the shadow ray
                        // arrives] [+ c2 if the MIS ray ends on the light].  Without a BSDF-sampled term — the rule: that term exists only when the sampled
                        // direction meets the light's triangle — the two possible results are known now: beta * (0 / pdf), which is +-0 unless beta or
                        // pdf is not finite, and beta * ((0 + c1) / pdf).  Then only the second is kept (16 bytes written and read instead of 48 + 48:
                        // the stage is bound by its slot bytes, experiments/README.md round 4), formed by the very operations of the resolve, and a path
                        // without a light term at all has nothing pending.  Anything else takes the general form.
                        const rgb k_occ = beta * (mkrgb(0.0f) / pdf_choice);
                        if (!(st & ST_HAS_C2) && k_occ.r == 0.0f && k_occ.g == 0.0f && k_occ.b == 0.0f) {
                            if (st & ST_HAS_C1) {
                                const rgb k = beta * ((mkrgb(0.0f) + c1) / pdf_choice);
                                if (MOVE) *mv->s_c1 = make_float4(k.r, k.g, k.b, 0.0f);
                                else pb.nee_c1[p] = make_float4(k.r, k.g, k.b, 0.0f);
                                st |= ST_PENDING | ST_COMPACT;
                            }
                        } else {
                            if (MOVE) *mv->s_c1 = make_float4(c1.r, c1.g, c1.b, pdf_choice);
                            else pb.nee_c1[p] = make_float4(c1.r, c1.g, c1.b, pdf_choice);
                            pb.nee_c2[og] = make_float4(c2.r, c2.g, c2.b, __uint_as_float(light_num));
                            pb.nee_beta[og] = make_float4(beta.r, beta.g, beta.b, 0.0f);
                            st |= ST_PENDING;  // even an all-zero estimate is added (l + beta*0 == l)
                        }
                    }
                }

                // ---- continuation (path.rs:141-188) ----
                f3 wi{0.0f, 0.0f, 0.0f};
                float pdf = 0.0f;
                uint32_t sampled_type = 255;
                const f3 wo_ray = -ray_d;  // path.rs:141 takes -ray.d here; estimate_direct above takes isect.wo (they differ inside instances only)
                rgb f = bsdf.template sample_f<F>(wo_ray, &wi, smp.get_2d(rd), &pdf, BX_ALL, &sampled_type);
                bool go_on = !(is_black(f) || pdf == 0.0f);
                if (go_on) {
                    beta = beta * ((f * absdot(wi, h.sh_n)) / pdf);
                    st = (sampled_type & BX_SPEC) ? (st | ST_SPECULAR) : (st & ~ST_SPECULAR);
                    if ((sampled_type & BX_SPEC) && (sampled_type & BX_TRANS)) {
                        float eta = bsdf.eta;
                        if (dot(wo_ray, h.n) > 0.0f) eta_scale *= eta * eta;
                        else eta_scale *= 1.0f / (eta * eta);
                    }
                    f3 o = offset_ray_origin(h.p, h.p_err, h.n, wi);
                    // Russian roulette (path.rs:251-262)
                    rgb rr = beta * eta_scale;
                    if (maxc(rr) < rd.rr_threshold && bounces > 3) {
                        float q = fmaxf(0.05f, 1.0f - maxc(rr));
                        if (smp.get_1d(rd) < q) go_on = false;
                        else beta = beta / (1.0f - q);
                    }
                    if (go_on) {
                        if (MOVE) { mv->o = o; mv->d = wi; mv->beta = beta; }
                        else {
                            store_ray(pb.ray_cont + p, o, wi, RSPT_INF, p);
                            pb.beta[p] = make_float4(beta.r, beta.g, beta.b, 0.0f);
                        }
                        st |= ST_ALIVE;
                        out.cont = true;
                    }
                }
                bounces += 1;
                st = (st & ~(ST_DIM_MASK | (0xffu << ST_BOUNCE_SHIFT))) | (smp.dim() & ST_DIM_MASK) | ((bounces & 0xffu) << ST_BOUNCE_SHIFT);
            }
        }
    }
    if (MOVE) { mv->L = make_float4(L.r, L.g, L.b, eta_scale); mv->st = st; }
    else {
        pb.L_eta[p] = make_float4(L.r, L.g, L.b, eta_scale);
        pb.state[p] = st;
    }
    out.active = (st & (ST_ALIVE | ST_PENDING)) != 0;
    return out;
}

// The part of the texture stage that does not depend on where the hit and its differentials come from: Material::bump, the material's
// texture slots, the raw rows of a dynamic material, the mask of lobes whose colour came out black.  `h` is the interaction in world
// space, `s` carries compute_differentials' results (zeros for a ray without differentials); rows go to out[row * stride].
// Always inlined: k_texture's launch time depends on it (see the comment at its call site); the per-lane integrators call it through
// texture_path / texture_hit_call.
__device__ __attribute__((always_inline)) inline void texture_hit(const SceneDev& sc, const TexTables& tt, TexHit& h, const TexSurf& s, uint32_t material, float4* out, size_t stride) {
    const rspt_material mat = sc.materials[material];
    const uint32_t mf = tt.mat_flags[material];
    uint32_t flags = 0;
    if (mat.bump_tex) {
        f3 bn, bdpdu;
        bump_map(tt, mat.bump_tex - 1u, h, s, &bn, &bdpdu);
        out[5 * stride] = make_float4(bdpdu.x, bdpdu.y, bdpdu.z, 0.0f);
        flags |= 1u;
        out[4 * stride] = make_float4(bn.x, bn.y, bn.z, 0.0f);  // (.w is completed below)
        h.sh_n = bn;
    }
    rgb tv[RSPT_TEX_SLOTS];
#pragma unroll
    for (int k = 0; k < RSPT_TEX_SLOTS; k++) {
        tv[k] = mkrgb(0.0f);
        const uint32_t sd = tt.mat_slots[(size_t)material * RSPT_TEX_SLOTS + k];
        if (sd != 0xffffffffu) {
            const uint32_t ti = sd & RSPT_SLOT_TEX_MASK;
            rgb v;
            if (sd & RSPT_SLOT_NODIFF) {  // MixMaterial hands m2 a SurfaceInteraction::new(p, uv, ..): no dudx .. dpdy (mixmat.rs:58-69)
                TexSurf s2 = s;
                s2.dudx = s2.dvdx = s2.dudy = s2.dvdy = 0.0f;
                s2.dpdx = s2.dpdy = f3{0.0f, 0.0f, 0.0f};
                v = tex_eval(tt, ti, s2);
            } else v = tex_eval(tt, ti, s);
            if (sd & RSPT_SLOT_ALPHA) {  // a roughness texture: the slot carries the lobe's alpha (plastic.rs:86-92, microfacet.rs:233-254)
                float a = v.r;
                if (sd & RSPT_SLOT_REMAP) {
                    const float r = fmaxf(a, 1e-3f), x = rspt_logf(r);
                    a = 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
                }
                a = fmaxf(0.001f, a);
                out[k * stride] = make_float4(a, a, a, 0.0f);
                continue;
            }
            tv[k] = rgb{v.r < 0.0f ? 0.0f : v.r, v.g < 0.0f ? 0.0f : v.g, v.b < 0.0f ? 0.0f : v.b};  // Spectrum::clamp(0, inf) = clamp_t per channel (pbrt.rs:108-123)
            out[k * stride] = make_float4(tv[k].r, tv[k].g, tv[k].b, 0.0f);
        }
    }
    if (mf & RSPT_MAT_DYNAMIC) {  // raw values of every varying parameter: the shade stage builds the lobe list from them (dynamic_lobes)
        const rspt_mat::DynMaterial& dm = tt.dyn[material];
        for (uint32_t r = 0; r < dm.n_rows; r++) {
            const uint32_t sd = dm.row_tex[r];
            rgb v;
            if (sd & RSPT_SLOT_NODIFF) {
                TexSurf s2 = s;
                s2.dudx = s2.dvdx = s2.dudy = s2.dvdy = 0.0f;
                s2.dpdx = s2.dpdy = f3{0.0f, 0.0f, 0.0f};
                v = tex_eval(tt, sd & RSPT_SLOT_TEX_MASK, s2);
            } else v = tex_eval(tt, sd & RSPT_SLOT_TEX_MASK, s);
            out[(size_t)(RSPT_TEX_ROWS + r) * stride] = make_float4(v.r, v.g, v.b, 0.0f);
        }
    }
    // the reference's `if !colour.is_black()` guards around bsdf.add (matte.rs:70, plastic.rs:70,84, substrate.rs:72, uber.rs)
    const uint32_t nl = mat.n_bxdfs < 8u ? mat.n_bxdfs : 8u;
    for (uint32_t l = 0; l < nl; l++) {
        const rspt_bxdf& b = sc.bxdfs[mat.first_bxdf + l];
        if (!b.tex_r && !b.tex_t) continue;
        rgb r = ldrgb(b.r), t = ldrgb(b.t);
        if (b.tex_r) r = r * (b.tex_r == 1 ? tv[0] : (b.tex_r == 2 ? tv[1] : (b.tex_r == 3 ? tv[2] : tv[3])));
        if (b.tex_t) t = t * (b.tex_t == 1 ? tv[0] : (b.tex_t == 2 ? tv[1] : (b.tex_t == 3 ? tv[2] : tv[3])));
        const bool two = b.type == RSPT_BXDF_FRESNEL_SPEC || b.type == RSPT_BXDF_FRESNEL_BLEND;
        if (two ? (is_black(r) && is_black(t)) : is_black(r)) flags |= 1u << (8 + l);
    }
    float4 m4 = (flags & 1u) ? out[4 * stride] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    m4.w = __uint_as_float(flags);
    out[4 * stride] = m4;
}
RDEVN void texture_hit_call(const SceneDev& sc, const TexTables& tt, TexHit& h, const TexSurf& s, uint32_t material, float4* out, size_t stride) { texture_hit(sc, tt, h, s, material, out, stride); }
// ---- texture stage: runs in front of k_shade when the scene has textures -------------
// For every path whose continuation ray hit a textured material: compute_differentials (camera rays
// only; bounce rays carry none, interaction.rs:388-479), Material::bump, and the clamped value of each
// texture the material's lobes are bound to; k_shade picks the results up from pb.tex.
// the texture stage for path slot p (see above); lens: the camera sample's lens position when it cannot be recomputed from the
// sample index (pixel samplers), else nullptr
// The stage for ONE hit: p = the slot that holds the hit record, its ray, the instance word and the result rows; smp = the camera sample's slot (film position, sampler
// index: the same slot for the path integrator, slot / H for directlighting's node slots); tix = index of the ray's time (moving instances); with_diff: the ray is the camera
// ray itself (bounce rays and rays re-spawned behind a null material carry no differentials)
RDEV void texture_slot(const SceneDev& sc, const TexTables& tt, const RenderDev& rd, const PathBuf& pb, uint32_t p, uint32_t smp, uint32_t tix, bool with_diff, const f3* lens) {
    const float4 hc = pb.hit_cont[p];
    const uint32_t prim = __float_as_uint(hc.x);
    const TriRec tri = load_tri(sc, prim);
    if (tri.material == 0xffffffffu) return;
    const uint32_t mf = tt.mat_flags[tri.material];
    if (!mf) return;
    TexHit h;
    tri_fill_tex(sc, prim, tri, hc.y, hc.z, hc.w, &h);
    if (pb.hit_inst) {
        const uint32_t hi = pb.hit_inst[p];
        if (hi && sc.inst[hi - 1u].anim != RSPT_MISS) {   // a moving instance: its Transform at the path's time (inst_at)
            const InstDev at = inst_at(sc, hi - 1u, sc.ray_time ? sc.ray_time[tix] : 0.0f);
            if (!at.identity) {
                if (!sc.inst_fixed) return;
                inst_texhit(at, &h);
            }
        } else if (hi && !sc.inst[hi - 1u].identity) {
            if (!sc.inst_fixed) return;  // reference behaviour: the hit has lost its primitive, nothing to texture
            inst_texhit(sc.inst[hi - 1u], &h);
        }
    }
    TexSurf s;
    s.p = h.p; s.uv = h.uv;
    s.dudx = s.dvdx = s.dudy = s.dvdy = 0.0f;
    s.dpdx = s.dpdy = f3{0.0f, 0.0f, 0.0f};
    if (with_diff) {
        const float4* rp = reinterpret_cast<const float4*>(pb.ray_cont + p);
        const float4 r0 = rp[0], r1 = rp[1];
        const float2 pf = pb.p_film[smp];
        f3 p_lens{0.0f, 0.0f, 0.0f};
        if (lens) p_lens = *lens;   // a pixel sampler's lens / time sample (tile_serial.h): not a function of (index, dimension)
        else {
            const uint64_t index = pb.sobol_index[smp];
            const bool hal = rd.sampler_kind == RSPT_SAMPLER_HALTON;
            if (rd.lens_radius > 0.0f) { p_lens.x = hal ? halton_dim(rd, index, 3) : sobol_dim(rd, index, 3); p_lens.y = hal ? halton_dim(rd, index, 4) : sobol_dim(rd, index, 4); }
            if (rd.cam_anim) p_lens.z = hal ? halton_dim(rd, index, 2) : sobol_dim(rd, index, 2);
        }
        f3 rx_o, rx_d, ry_o, ry_d;
        camera_differentials(rd, f2{pf.x, pf.y}, p_lens, f3{r0.x, r0.y, r0.z}, f3{r0.w, r1.x, r1.y}, &rx_o, &rx_d, &ry_o, &ry_d);
        compute_differentials(h, rx_o, rx_d, ry_o, ry_d, &s);
    }
    texture_hit(sc, tt, h, s, tri.material, pb.tex + p, pb.tex_stride);
}
RDEVN void texture_path(const SceneDev& sc, const TexTables& tt, const RenderDev& rd, const PathBuf& pb, uint32_t p, const f3* lens) {
    const uint32_t st = pb.state[p];
    if (!(st & ST_ALIVE)) return;
    const uint32_t prim = __float_as_uint(pb.hit_cont[p].x);
    const uint32_t bounces = (st >> ST_BOUNCE_SHIFT) & 0xffu;
    if (prim == RSPT_MISS || bounces >= rd.max_depth) return;
    texture_slot(sc, tt, rd, pb, p, p, p, bounces == 0 && !(st & ST_NO_DIFF) /* the camera ray itself (a null-material pass-through re-spawns without differentials) */, lens);
}

// ---- K7b: bin the active queue by what the shade stage will do with each path -------------------------------------------
// The queue leaves k_shade in path-slot order; after the trace stage a wave's 64 paths have hit different things: nothing (an
// escaped path costs a handful of instructions), the depth limit, surfaces of different materials (different lobe lists).  A
// wave pays for the longest of them.  Three small kernels sort the queue into RSPT_BIN_K classes, each class padded to whole
// waves, so that k_shade's waves are (nearly) uniform.  The order inside a class keeps the queue's.
#define RSPT_BIN_K 16
#define RSPT_BIN_INVALID 0xffffffffu
struct BinInfo {  // one per wavefront iteration, zeroed with the queue counters
    uint32_t count[RSPT_BIN_K], cursor[RSPT_BIN_K], start[RSPT_BIN_K];
    uint32_t total;  // padded length of the sorted queue
    uint32_t pad[15];
};
// q_sorted / bi (K7b, when the shade queue is binned): the same entries sorted by class and padded to whole waves, so that a wave of this
// kernel too evaluates ONE material's texture graph (escaped paths and the depth limit have waves of their own and return at once)
#ifdef RSPT_TEX_WAVES   // A/B knob (tools/ab_build.sh AB_DEFS=-DRSPT_TEX_WAVES=3): the stage built for 3 / 4 waves per SIMD
#define RSPT_TEX_OCC __attribute__((amdgpu_waves_per_eu(RSPT_TEX_WAVES, RSPT_TEX_WAVES)))
#else
#define RSPT_TEX_OCC
#endif
RSPT_PLAIN_KERNEL __launch_bounds__(256) RSPT_TEX_OCC void k_texture(SceneDev sc, TexTables tt, RenderDev rd, PathBuf pb, const uint32_t* __restrict__ q_active,
                                                 const uint32_t* __restrict__ count_in, const uint32_t* __restrict__ q_sorted, const BinInfo* __restrict__ bi) {
    const uint32_t n = q_sorted ? bi->total : *count_in;
    // inlined at this call site: as a call (the compiler's choice once the stage had a second caller) the launch runs 28 % longer — 248 VGPRs and
    // 1056 B of scratch against 209 and 368; the tile-serial kernel, at its register ceiling, keeps calling it
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t p = q_sorted ? q_sorted[i] : q_active[i];
        if (p != RSPT_BIN_INVALID) [[clang::always_inline]] texture_path(sc, tt, rd, pb, p, nullptr);
    }
}
RDEV uint32_t bin_key(const SceneDev& sc, const PathBuf& pb, uint32_t max_depth, uint32_t p) {
    const uint32_t st = pb.state[p];
    const uint32_t prim = __float_as_uint(pb.hit_cont[p].x);
    if (!(st & ST_ALIVE) || prim == RSPT_MISS) return 0u;
    if (((st >> ST_BOUNCE_SHIFT) & 0xffu) >= max_depth) return 1u;
    const uint32_t mat = __float_as_uint(sc.tris[3 * (size_t)prim + 2].y);
    return mat == 0xffffffffu ? 1u : 2u + mat % (RSPT_BIN_K - 2u);
}
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_bin_count(SceneDev sc, PathBuf pb, uint32_t max_depth, const uint32_t* __restrict__ q_active,
                                                   const QueueCounts* __restrict__ cnt_in, uint8_t* __restrict__ keys, BinInfo* bi) {
    __shared__ uint32_t hist[RSPT_BIN_K];
    if (threadIdx.x < RSPT_BIN_K) hist[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t n = cnt_in->active;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t k = bin_key(sc, pb, max_depth, q_active[i]);
        keys[i] = (uint8_t)k;
        atomicAdd(&hist[k], 1u);
    }
    __syncthreads();
    if (threadIdx.x < RSPT_BIN_K && hist[threadIdx.x]) atomicAdd(&bi->count[threadIdx.x], hist[threadIdx.x]);
}
RSPT_PLAIN_KERNEL void k_bin_starts(BinInfo* bi, uint32_t* __restrict__ q_sorted) {  // one 64-thread block
    __shared__ uint32_t start[RSPT_BIN_K];
    if (threadIdx.x == 0) {
        uint32_t at = 0;
        for (int k = 0; k < RSPT_BIN_K; k++) { start[k] = at; bi->start[k] = at; at += (bi->count[k] + 63u) & ~63u; }
        bi->total = at;
    }
    __syncthreads();
    for (int k = 0; k < RSPT_BIN_K; k++) {  // the idle lanes that pad every class to whole waves
        const uint32_t c = bi->count[k], idx = c + threadIdx.x;
        if (idx < ((c + 63u) & ~63u)) q_sorted[start[k] + idx] = RSPT_BIN_INVALID;
    }
}
#define RSPT_BIN_E 8  // queue entries per thread and round of k_bin_scatter: one global atomic per class per 2048 entries (its round trip is what a round waits for)
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_bin_scatter(const uint32_t* __restrict__ q_active, const QueueCounts* __restrict__ cnt_in, const uint8_t* __restrict__ keys,
                                                     BinInfo* bi, uint32_t* __restrict__ q_sorted) {
    __shared__ uint32_t s_cnt[RSPT_BIN_E * 4][RSPT_BIN_K], s_base[RSPT_BIN_K];
    const uint32_t n = cnt_in->active;
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    const uint64_t lt = (1ull << lane) - 1ull;
    for (uint32_t base = blockIdx.x * (256u * RSPT_BIN_E); base < n; base += gridDim.x * (256u * RSPT_BIN_E)) {
        for (uint32_t t = threadIdx.x; t < RSPT_BIN_E * 4 * RSPT_BIN_K; t += 256u) (&s_cnt[0][0])[t] = 0u;
        __syncthreads();
        uint32_t key[RSPT_BIN_E], rank[RSPT_BIN_E];
#pragma unroll
        for (uint32_t j = 0; j < RSPT_BIN_E; j++) {  // slice j: entries base + 256 j + thread (coalesced); rank inside (slice, wave, class)
            const uint32_t i = base + 256u * j + threadIdx.x;
            const uint32_t k = i < n ? keys[i] : RSPT_BIN_K;
            uint32_t r = 0;
            uint64_t todo = __ballot(k < RSPT_BIN_K);
            while (todo) {  // one ballot per class present in the wave (rarely more than three)
                const uint32_t c = (uint32_t)__shfl((int)k, (int)__builtin_ctzll(todo));
                const uint64_t m = __ballot(k == c);
                if (k == c) r = (uint32_t)__popcll(m & lt);
                if (lane == (uint32_t)__builtin_ctzll(todo)) s_cnt[4u * j + wave][c] = (uint32_t)__popcll(m);
                todo &= ~m;
            }
            key[j] = k; rank[j] = r;
        }
        __syncthreads();
        if (threadIdx.x < RSPT_BIN_K) {  // exclusive prefix over (slice, wave) per class, then one atomic for the class
            uint32_t run = 0;
            for (uint32_t u = 0; u < RSPT_BIN_E * 4; u++) { const uint32_t c = s_cnt[u][threadIdx.x]; s_cnt[u][threadIdx.x] = run; run += c; }
            s_base[threadIdx.x] = run ? bi->start[threadIdx.x] + atomicAdd(&bi->cursor[threadIdx.x], run) : 0u;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t j = 0; j < RSPT_BIN_E; j++) {
            const uint32_t i = base + 256u * j + threadIdx.x;
            if (key[j] < RSPT_BIN_K) q_sorted[s_base[key[j]] + s_cnt[4u * j + wave][key[j]] + rank[j]] = q_active[i];
        }
        __syncthreads();
    }
}

// The shade kernel's body; F = the feature set (dev_bsdf.h SF_*) the instantiation is compiled for.  k_shade<F> leaves the register budget
// to the compiler, k_shade_w<F, W> asks for W waves per SIMD (the allocator spills what does not fit 512 / W registers).
// MOVE (round 6): queue entries are positions; what a path carries on is stored at its position in the next queue (PathBuf, ShadeMove)
struct ShadeStage { float4 c1[256]; float4 sh0[256]; float2 sh1[256]; };   // 10 KB of LDS per workgroup (MOVE instantiations only)
template <uint32_t F, bool MOVE>
__device__ __forceinline__ void shade_kernel(const SceneDev& sc, const LightDistDev& ld, const RenderDev& rd, const PathBuf& pb, const uint32_t* __restrict__ q_active,
                                             const QueueCounts* __restrict__ cnt_in, QueueCounts* cnt_out, uint32_t* __restrict__ q_active_next,
                                             uint32_t* __restrict__ q_closest_next, uint32_t* __restrict__ q_any_next, unsigned long long* stats,
                                             uint32_t sob_nd, uint32_t sob_bits, uint32_t qcap, const uint32_t* __restrict__ q_sorted, const BinInfo* __restrict__ bi,
                                             uint32_t* sob_tab, uint32_t (*s_wave)[5], uint32_t* s_base, ShadeStage* stage) {
    // sob_tab: Sobol' generator matrices of the dimensions this render can reach, transposed to [bit][dim]
    for (uint32_t t = threadIdx.x; ((F & SF_SOBOL) || !(F & SF_HALTON)) && (!(F & SF_HALTON) || rd.sampler_kind == RSPT_SAMPLER_SOBOL) && t < sob_nd * sob_bits; t += 256u) {
        uint32_t dd = t % sob_nd;  // read-ahead columns past the last of the 1024 dimensions are never consumed
        sob_tab[t] = rd.sobol32[(dd < 1024u ? dd : 1023u) * 52u + (t / sob_nd)];
    }
    __syncthreads();
    // the active queue has two ends: paths with a continuation ray in flight from the front, paths that only have a
    // pending next-event estimate left from the back (they take a fraction of the instructions), so a wave is either
    // all-alive or all-short; measured effect on C3 is small (+0.5 %) because most paths end in the same last iterations
    // q_sorted (K7b): the front part binned by class and padded to whole waves (RSPT_BIN_INVALID = idle lane)
    const uint32_t n_front = q_sorted ? bi->total : cnt_in->active, n = n_front + cnt_in->active_tail;
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t base = virtual_block() * 256u; base < n; base += stride) {
        uint32_t i = base + threadIdx.x;
        ShadeOut o{false, false, false, false};
        ShadeMove mv;
        if (MOVE) { mv.valid = false; mv.s_c1 = &stage->c1[threadIdx.x]; mv.s_sh0 = &stage->sh0[threadIdx.x]; mv.s_sh1 = &stage->sh1[threadIdx.x]; }
        uint32_t p = 0;
        if (i < n) {
            p = i < n_front ? (q_sorted ? q_sorted[i] : q_active[i]) : q_active[qcap - 1u - (i - n_front)];
            if (p != RSPT_BIN_INVALID) o = shade_path<false, F, MOVE>(sc, ld, rd, pb, p, stats, sob_tab, sob_nd, nullptr, MOVE ? &mv : nullptr);
        }
        // queue appends, aggregated per workgroup: a single counter word sustains only ~90 M atomics/s
        // (MI355X_MICROARCH "dequeue" row), so one atomic per queue per 256 paths instead of per wave.
        // Order: active | closest = continuation rays then MIS rays | any.
        const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
        const uint64_t lt = (1ull << lane) - 1ull;
        const bool tail = o.active && !o.cont;  // only ST_PENDING left
        const uint64_t m_act = __ballot(o.cont), m_tail = __ballot(tail), m_cont = __ballot(o.cont), m_mis = __ballot(o.mis), m_sh = __ballot(o.shadow);
        if (lane == 0) {
            s_wave[wave][4] = (uint32_t)__popcll(m_tail);
            s_wave[wave][0] = (uint32_t)__popcll(m_act);
            s_wave[wave][1] = (uint32_t)__popcll(m_cont);
            s_wave[wave][2] = (uint32_t)__popcll(m_mis);
            s_wave[wave][3] = (uint32_t)__popcll(m_sh);
        }
        __syncthreads();
        if (threadIdx.x < 4) {
            uint32_t q = threadIdx.x;  // 0 active (front), 1 closest (cont + mis), 2 any, 3 active (tail)
            uint32_t tot = 0;
            for (int w = 0; w < 4; w++) tot += q == 0 ? s_wave[w][0] : (q == 1 ? s_wave[w][1] + s_wave[w][2] : (q == 2 ? s_wave[w][3] : s_wave[w][4]));
            uint32_t* ctr = q == 0 ? &cnt_out->active : (q == 1 ? &cnt_out->closest : (q == 2 ? &cnt_out->any : &cnt_out->active_tail));
#if defined(RSPT_SHADE_EXP) && RSPT_SHADE_EXP == 1   // timing experiment (tools/ab_build.sh, WRONG pictures): what the queue-counter atomics cost the first shade launch
            s_base[q] = base;
            if (tot == 0xffffffffu) *ctr = tot;
#else
            s_base[q] = tot ? atomicAdd(ctr, tot) : 0u;
#endif
        }
        __syncthreads();
        uint32_t off_act = s_base[0], off_cont = s_base[1], off_mis = s_base[1], off_sh = s_base[2], off_tail = s_base[3];
        for (uint32_t w = 0; w < 4; w++) {
            if (w < wave) { off_act += s_wave[w][0]; off_cont += s_wave[w][1]; off_mis += s_wave[w][2]; off_sh += s_wave[w][3]; off_tail += s_wave[w][4]; }
            off_mis += s_wave[w][1];  // MIS entries follow all continuation entries of the workgroup
        }
        if (MOVE) {
            const uint32_t og = (mv.valid && !pb.orig_is_p) ? pb.orig[p] : p;   // the original slot (the first MOVE launch of a batch: position = slot)
            // the path's next position: its place in the next active queue (front: a continuation ray is in flight; back: only an estimate is pending)
            const uint32_t p2 = o.cont ? off_act + (uint32_t)__popcll(m_act & lt) : qcap - 1u - (off_tail + (uint32_t)__popcll(m_tail & lt));
            if (o.cont) {
                q_active_next[p2] = p2;
                q_closest_next[off_cont + (uint32_t)__popcll(m_cont & lt)] = p2;
                store_ray(pb.o_ray_cont + p2, mv.o, mv.d, RSPT_INF, p2);
                pb.o_beta[p2] = make_float4(mv.beta.r, mv.beta.g, mv.beta.b, 0.0f);
                pb.o_sobol_index[p2] = pb.sobol_index[p];
            }
            if (tail) q_active_next[p2] = p2;
            if (o.cont || tail) {
                pb.o_L_eta[p2] = mv.L;
                pb.o_state[p2] = mv.st;
                pb.o_orig[p2] = og;
                if (mv.st & ST_PENDING) pb.o_nee_c1[p2] = *mv.s_c1;
            } else if (mv.valid)
                pb.L_final[og] = mv.L;   // the path ends here: its radiance goes where k_film reads it
            if (o.mis) q_closest_next[off_mis + (uint32_t)__popcll(m_mis & lt)] = og | RSPT_Q_MIS;   // (the general form of an estimate stays with the original slot)
            if (o.shadow) {
                const float4 a = *mv.s_sh0; const float2 b = *mv.s_sh1;
                store_ray(pb.ray_sh + p2, f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, 1.0f - RSPT_SHADOW_EPS, p2);
                q_any_next[off_sh + (uint32_t)__popcll(m_sh & lt)] = p2;
            }
        } else {
            if (o.cont) q_active_next[off_act + (uint32_t)__popcll(m_act & lt)] = p;
            if (tail) q_active_next[qcap - 1u - (off_tail + (uint32_t)__popcll(m_tail & lt))] = p;
            if (o.cont) q_closest_next[off_cont + (uint32_t)__popcll(m_cont & lt)] = p;
            if (o.mis) q_closest_next[off_mis + (uint32_t)__popcll(m_mis & lt)] = p | RSPT_Q_MIS;
            if (o.shadow) q_any_next[off_sh + (uint32_t)__popcll(m_sh & lt)] = p;
        }
        __syncthreads();  // s_wave / s_base are reused by the next stripe
    }
}

#define RSPT_SHADE_ARGS SceneDev sc, LightDistDev ld, RenderDev rd, PathBuf pb, const uint32_t* __restrict__ q_active, const QueueCounts* __restrict__ cnt_in, QueueCounts* cnt_out, \
                        uint32_t* __restrict__ q_active_next, uint32_t* __restrict__ q_closest_next, uint32_t* __restrict__ q_any_next, unsigned long long* stats, uint32_t sob_nd, \
                        uint32_t sob_bits, uint32_t qcap, const uint32_t* __restrict__ q_sorted, const BinInfo* __restrict__ bi
#define RSPT_SHADE_CALL(MOVE, STAGE) shade_kernel<F, MOVE>(sc, ld, rd, pb, q_active, cnt_in, cnt_out, q_active_next, q_closest_next, q_any_next, stats, sob_nd, sob_bits, qcap, q_sorted, bi, sob_tab, s_wave, s_base, STAGE)
template <uint32_t F>
__global__ __launch_bounds__(256) void k_shade(RSPT_SHADE_ARGS) {
    extern __shared__ uint32_t sob_tab[];
    __shared__ uint32_t s_wave[4][5], s_base[4];
    RSPT_SHADE_CALL(false, nullptr);
}
template <uint32_t F, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void k_shade_w(RSPT_SHADE_ARGS) {
    extern __shared__ uint32_t sob_tab[];
    __shared__ uint32_t s_wave[4][5], s_base[4];
    RSPT_SHADE_CALL(false, nullptr);
}
// the MOVE forms (W = 0: the compiler's own register budget)
template <uint32_t F, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W))) void k_shade_mw(RSPT_SHADE_ARGS) {
    extern __shared__ uint32_t sob_tab[];
    __shared__ uint32_t s_wave[4][5], s_base[4];
    __shared__ ShadeStage stage;
    RSPT_SHADE_CALL(true, &stage);
}
template <uint32_t F>
__global__ __launch_bounds__(256) void k_shade_m(RSPT_SHADE_ARGS) {
    extern __shared__ uint32_t sob_tab[];
    __shared__ uint32_t s_wave[4][5], s_base[4];
    __shared__ ShadeStage stage;
    RSPT_SHADE_CALL(true, &stage);
}
// MOVE: after the last iteration of a batch, whatever is still in the queue (paths cut by RSPT_NULL_PASSES; normally nothing) hands its radiance to the film
// moved = 0: no MOVE launch has run in this batch (max_depth below RSPT_MOVE_FROM): every slot's radiance is still in L_eta by slot, copied over as a whole
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_move_flush(PathBuf pb, const uint32_t* __restrict__ q_active, const QueueCounts* __restrict__ cnt, uint32_t qcap, uint32_t moved) {
    const uint32_t n_front = cnt->active, n = n_front + cnt->active_tail;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t p = i < n_front ? q_active[i] : q_active[qcap - 1u - (i - n_front)];
        pb.L_final[moved ? pb.orig[p] : p] = pb.L_eta[p];
    }
}

// ---- AOIntegrator::li (src/integrators/ao.rs:50-96) --------------------------
// Stage 1, one lane per camera sample: frame at the hit, then the pixel sample's slice of the sampler's 2-D array
// (GlobalSampler dimensions 5, 6 of the samples get_index_for_sample(s * n + k), sobol.rs:166-179 /
// halton.rs:260-272) -> n shadow rays.  Ray k of path i sits at ray_sh[i * n + k]; its id field carries the
// term dot(wi, n) / (pdf * n) that stage 2 adds when the ray is unoccluded.
#define RSPT_AO_SKIP 0xffffffffu  // id of a ray that is not traced (pdf == 0)
template <bool ANIM>   // ANIM: the scene has moving instances (their Transform at the camera sample's time, inst_at)
__global__ __launch_bounds__(256) void k_ao_spawn(SceneDev sc, RenderDev rd, Batch bt, PathBuf pb, const uint32_t* __restrict__ pix_list,
                                                  uint32_t n_samples, uint32_t cos_sample, uint32_t* __restrict__ q_any, QueueCounts* cnt) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= bt.n) return;
    const float4 hc = pb.hit_cont[i];
    const uint32_t prim = __float_as_uint(hc.x);
    pb.state[i] = prim == RSPT_MISS ? 0u : 1u;
    if (prim == RSPT_MISS) return;
    const float4* rp = reinterpret_cast<const float4*>(pb.ray_cont + i);
    const float4 r0 = rp[0], r1 = rp[1];
    const f3 ray_d{r0.w, r1.x, r1.y};
    const TriRec tri = load_tri(sc, prim);
    TexHit h;  // the full interaction: li needs the geometric dpdu, not the shading one
    tri_fill_tex(sc, prim, tri, hc.y, hc.z, hc.w, &h);
    Hit hp;    // p_error for spawn_ray
    tri_fill(sc, prim, tri, hc.y, hc.z, hc.w, &hp);
    if (pb.hit_inst) {
        const uint32_t hi = pb.hit_inst[i];
        InstDev moved{};   // (a moving instance: its Transform at the camera sample's time)
        if (ANIM && hi) moved = inst_at(sc, hi - 1u, sc.ray_time ? sc.ray_time[i] : 0.0f);
        const InstDev& in = ANIM ? moved : sc.inst[hi ? hi - 1u : 0u];
        if (hi && !in.identity) { inst_texhit(in, &h); inst_hit(in, &hp); }
    }
    const f3 n = faceforward(h.n, -ray_d);
    const f3 sv = normalize(h.dpdu);
    const f3 tv = cross(h.n, sv);  // nrm_cross_vec3(&isect.common.n, &s)
    const uint32_t pk = pix_list[bt.pix0 + i / bt.ns];
    const int32_t px = (int32_t)(int16_t)(pk & 0xffffu), py = (int32_t)(int16_t)(pk >> 16);
    const uint64_t first = (uint64_t)(bt.s0 + i % bt.ns) * n_samples;
    const uint32_t base = atomicAdd(&cnt->any, n_samples);
    rspt_ray* out = pb.ray_sh + (size_t)i * n_samples;
    for (uint32_t k = 0; k < n_samples; k++) {
        f2 u;
        if (rd.sampler_kind == RSPT_SAMPLER_HALTON) {
            const uint64_t index = halton_index(rd, px, py, first + k);
            u = f2{halton_dim(rd, index, 5), halton_dim(rd, index, 6)};
        } else {
            const uint64_t index = sobol_interval_to_index(rd, (uint32_t)rd.log2_res, first + k, px - rd.sample_bounds[0], py - rd.sample_bounds[1]);
            u = f2{sobol_dim(rd, index, 5), sobol_dim(rd, index, 6)};
        }
        f3 wi;
        float pdf;
        if (cos_sample) {
            wi = cosine_hemisphere(u);
            pdf = fabsf(wi.z) * RSPT_INV_PI;
        } else {  // uniform_sample_hemisphere (sampling.rs:309-318)
            const float z = u.x, r = sqrtf(fmaxf(0.0f, 1.0f - z * z)), phi = 2.0f * RSPT_PI * u.y;
            wi = f3{r * rspt_cosf(phi), r * rspt_sinf(phi), z};
            pdf = 0.15915494309189533577f;
        }
        wi = f3{sv.x * wi.x + tv.x * wi.y + n.x * wi.z, sv.y * wi.x + tv.y * wi.y + n.y * wi.z, sv.z * wi.x + tv.z * wi.y + n.z * wi.z};
        uint32_t id = RSPT_AO_SKIP;
        if (pdf != 0.0f) id = __float_as_uint(dot(wi, n) / (pdf * (float)n_samples));
        store_ray(out + k, offset_ray_origin(hp.p, hp.p_err, hp.n, wi), wi, pdf != 0.0f ? RSPT_INF : 0.0f, id);
        q_any[base + k] = i * n_samples + k;
    }
}
// Stage 2: l += Spectrum::new(term) for the unoccluded rays, in array order (ao.rs:86-91)
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_ao_resolve(Batch bt, PathBuf pb, uint32_t n_samples) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= bt.n) return;
    float l = 0.0f;
    if (pb.state[i]) {
        const rspt_ray* rays = pb.ray_sh + (size_t)i * n_samples;
        const uint32_t* occ = pb.occluded + (size_t)i * n_samples;
        for (uint32_t k = 0; k < n_samples; k++) {
            const uint32_t id = rays[k].id;
            if (id != RSPT_AO_SKIP && occ[k] == 0u) l += __uint_as_float(id);
        }
    }
    pb.L_eta[i] = make_float4(l, l, l, 1.0f);
}

// ---- K8 -------------------------------------------------------------------------------------
// FilmTile::add_sample (film.rs:94-147) for all samples of one pixel in this batch, in sample
// order.  The pixel's own contributions accumulate in registers; splats into other pixels
// (wide filters; exact-zero film offsets with the box filter) go through atomics into a
// separate buffer that is folded in by k_film_resolve.
#define RSPT_FILM_CHUNK 4
#define RSPT_FILM_ROW (RSPT_FILM_CHUNK + 1)  // padded LDS row: the per-thread walk would otherwise hit the same banks
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_film(RenderDev rd, Batch bt, PathBuf pb, const uint32_t* __restrict__ pix_list, float4* __restrict__ film_own,
                                              float* __restrict__ film_splat, float* __restrict__ li_out, unsigned long long* nan_count) {
    // The block's 256 pixels own one contiguous range of path slots (pixel-major).  It is staged through LDS in chunks
    // of RSPT_FILM_CHUNK samples per pixel with unit-stride loads; each thread then walks its own pixel's samples in
    // order (a thread reading its 16 consecutive records directly touches a different cache line per lane per load).
    __shared__ float4 s_l[256 * RSPT_FILM_ROW];
    __shared__ float2 s_pf[256 * RSPT_FILM_ROW];
    const uint32_t k0 = blockIdx.x * 256u;
    uint32_t k = k0 + threadIdx.x;
    const bool live = k < bt.n_pix;
    uint32_t pk = live ? pix_list[bt.pix0 + k] : 0u;
    int32_t px = (int32_t)(int16_t)(pk & 0xffffu), py = (int32_t)(int16_t)(pk >> 16);
    const int32_t* sb = rd.sample_bounds;
    const int32_t* cp = rd.crop_px;
    const int32_t ts = (int32_t)rd.tile_size;
    // tile of this pixel (integrator.rs:115-120) and its FilmTile pixel bounds (film.rs:308-345)
    int32_t tx0 = sb[0] + ((px - sb[0]) / ts) * ts, ty0 = sb[1] + ((py - sb[1]) / ts) * ts;
    int32_t tx1 = min(tx0 + ts, sb[2]), ty1 = min(ty0 + ts, sb[3]);
    float rx = rd.filter_radius[0], ry = rd.filter_radius[1];
    int32_t bx0 = max(f2i_sat(ceilf((float)tx0 - 0.5f - rx)), cp[0]), by0 = max(f2i_sat(ceilf((float)ty0 - 0.5f - ry)), cp[1]);
    int32_t bx1 = min(f2i_sat(floorf((float)tx1 - 0.5f + rx)) + 1, cp[2]), by1 = min(f2i_sat(floorf((float)ty1 - 0.5f + ry)) + 1, cp[3]);
    const int32_t cw = cp[2] - cp[0];
    const bool own_in_crop = live && px >= cp[0] && px < cp[2] && py >= cp[1] && py < cp[3];
    const size_t own_idx = own_in_crop ? (size_t)(py - cp[1]) * cw + (size_t)(px - cp[0]) : 0;
    float4 acc = own_in_crop ? film_own[own_idx] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    float inv_rx = 1.0f / rx, inv_ry = 1.0f / ry;
    unsigned long long nans = 0;
    const uint32_t n_blk = min(256u, bt.n_pix - k0);  // pixels of this block
    for (uint32_t j = 0; j < bt.ns; j++) {
        const uint32_t jc = j % RSPT_FILM_CHUNK;
        if (jc == 0) {  // stage samples j .. j + chunk - 1 of all the block's pixels
            __syncthreads();
            const uint32_t cs = min((uint32_t)RSPT_FILM_CHUNK, bt.ns - j);
            for (uint32_t e = threadIdx.x; e < n_blk * cs; e += 256u) {
                const uint32_t pl = e / cs, sj = e % cs;  // consecutive e -> consecutive slots inside a pixel's run
                const uint32_t slot = (k0 + pl) * bt.ns + j + sj;
                s_l[pl * RSPT_FILM_ROW + sj] = pb.L_eta[slot];
                s_pf[pl * RSPT_FILM_ROW + sj] = pb.p_film[slot];
            }
            __syncthreads();
        }
        if (!live) continue;
        float4 le = s_l[threadIdx.x * RSPT_FILM_ROW + jc];
        rgb l{le.x, le.y, le.z};
        if (has_nans(l)) { l = mkrgb(0.0f); nans++; }  // integrator.rs:165-173
        if (li_out && own_in_crop) {
            float* o = li_out + (own_idx * (size_t)rd.spp + (size_t)(bt.s0 + j)) * 3;
            o[0] = l.r; o[1] = l.g; o[2] = l.b;
        }
        if (lum(l) > rd.max_sample_luminance) l = l * mkrgb(rd.max_sample_luminance / lum(l));
        float2 pf = s_pf[threadIdx.x * RSPT_FILM_ROW + jc];
        float dx = pf.x - 0.5f, dy = pf.y - 0.5f;
        int32_t x0 = max(f2i_sat(ceilf(dx - rx)), bx0), y0 = max(f2i_sat(ceilf(dy - ry)), by0);
        int32_t x1 = min(f2i_sat(floorf(dx + rx)) + 1, bx1), y1 = min(f2i_sat(floorf(dy + ry)) + 1, by1);
        for (int32_t y = y0; y < y1; y++) {
            float fy = fabsf(((float)y - dy) * inv_ry * 16.0f);
            int32_t ify = (int32_t)fminf(floorf(fy), 15.0f);
            for (int32_t x = x0; x < x1; x++) {
                float fx = fabsf(((float)x - dx) * inv_rx * 16.0f);
                int32_t ifx = (int32_t)fminf(floorf(fx), 15.0f);
                float w = rd.filter_table[ify * 16 + ifx];
                rgb c = l * mkrgb(1.0f) * mkrgb(w);  // l * sample_weight * filter_weight
                if (x == px && y == py) {
                    acc.x += c.r; acc.y += c.g; acc.z += c.b; acc.w += w;
                } else {
                    float* s = film_splat + 4 * ((size_t)(y - cp[1]) * cw + (size_t)(x - cp[0]));
                    atomicAdd(s + 0, c.r); atomicAdd(s + 1, c.g); atomicAdd(s + 2, c.b); atomicAdd(s + 3, w);
                }
            }
        }
    }
    if (own_in_crop) film_own[own_idx] = acc;
    if (nans) atomicAdd(nan_count, nans);
}

// ---- K8 for filters wider than a pixel (gaussian / mitchell / triangle / sinc scene files): a gather -------------------------------
// k_film above adds a sample to its own pixel in registers and to every other pixel of its footprint through four atomics: with a radius-2
// filter that is 60 atomics per sample, and the stage took 309 ms of a C2 step (2.4 ms under the box filter) and two thirds of a C3 step.
// Here a block owns 16 x 16 pixels of the sample-bounds grid; it stages the samples of those pixels and of a K-pixel halo (K = floor(radius + 0.5):
// the pixels whose samples can reach the block) through LDS, RSPT_FG chunk samples per pixel at a time — NaN test, luminance clamp and the
// footprint box (film.rs:94-147) are done once per sample there — and every thread sums what reaches ITS pixel: the arithmetic per (sample,
// pixel) pair is add_sample's, the order of a pixel's additions is fixed (source pixels row by row, samples in order; the reference's own
// order depends on which thread finishes a tile first), nothing is atomic and nothing lands in film_splat.
// pix_index[(y - sb1) * sbw + (x - sb0)] = position of pixel (x, y) in the pixel list the batch indexes, 0xffffffff = not in it.
#define RSPT_FG_T 16u
#define RSPT_FG_NONE 0xffffffffu
RSPT_PLAIN_KERNEL void k_pix_index(const uint32_t* __restrict__ pix_list, uint32_t n, int32_t sb0, int32_t sb1, int32_t sbw, uint32_t* __restrict__ pix_index) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t pk = pix_list[i];
    const int32_t px = (int32_t)(int16_t)(pk & 0xffffu), py = (int32_t)(int16_t)(pk >> 16);
    pix_index[(size_t)(py - sb1) * sbw + (size_t)(px - sb0)] = i;
}
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_film_gather(RenderDev rd, Batch bt, PathBuf pb, const uint32_t* __restrict__ pix_index, int32_t K, uint32_t chunk,
                                                     float4* __restrict__ film_own, float* __restrict__ li_out, unsigned long long* nan_count) {
    extern __shared__ float4 fg_smem[];
    const int32_t* sb = rd.sample_bounds;
    const int32_t* cp = rd.crop_px;
    const int32_t S = (int32_t)RSPT_FG_T + 2 * K, sbw = sb[2] - sb[0];
    const uint32_t row = chunk + 1u;   // padded sample row of a source pixel
    float4* s_a = fg_smem;                                                   // [S * S][row]: clamped L rgb, dx
    float2* s_b = reinterpret_cast<float2*>(s_a + (size_t)S * S * row);      // [S * S][row]: dy, footprint box (4 x int8 relative to the block's origin)
    uint32_t* s_pos = reinterpret_cast<uint32_t*>(s_b + (size_t)S * S * row);  // [S * S]: the source pixel's position in this batch, or NONE
    float* s_tab = reinterpret_cast<float*>(s_pos + S * S);                  // the 16 x 16 filter table
    const int32_t bx = sb[0] + (int32_t)(blockIdx.x * RSPT_FG_T), by = sb[1] + (int32_t)(blockIdx.y * RSPT_FG_T);
    int any = 0;
    for (int32_t e = (int32_t)threadIdx.x; e < S * S; e += 256) {
        const int32_t sx = bx - K + e % S, sy = by - K + e / S;
        uint32_t pos = RSPT_FG_NONE;
        if (sx >= sb[0] && sx < sb[2] && sy >= sb[1] && sy < sb[3]) pos = pix_index[(size_t)(sy - sb[1]) * sbw + (size_t)(sx - sb[0])];
        const bool in_batch = pos != RSPT_FG_NONE && pos >= bt.pix0 && pos - bt.pix0 < bt.n_pix;
        s_pos[e] = in_batch ? pos - bt.pix0 : RSPT_FG_NONE;
        any |= in_batch ? 1 : 0;
    }
    s_tab[threadIdx.x] = rd.filter_table[threadIdx.x];
    if (!__syncthreads_or(any)) return;   // none of this batch's samples can reach the block
    const int32_t tx = (int32_t)(threadIdx.x % RSPT_FG_T), ty = (int32_t)(threadIdx.x / RSPT_FG_T);
    const int32_t x = bx + tx, y = by + ty;
    const int32_t cw = cp[2] - cp[0];
    const bool in_crop = x >= cp[0] && x < cp[2] && y >= cp[1] && y < cp[3];
    const size_t idx = in_crop ? (size_t)(y - cp[1]) * cw + (size_t)(x - cp[0]) : 0;
    float4 acc = in_crop ? film_own[idx] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const float rx = rd.filter_radius[0], ry = rd.filter_radius[1];
    const float inv_rx = 1.0f / rx, inv_ry = 1.0f / ry;
    unsigned long long nans = 0;
    for (uint32_t j0 = 0; j0 < bt.ns; j0 += chunk) {
        const uint32_t cs = min(chunk, bt.ns - j0);
        __syncthreads();
        for (uint32_t e = threadIdx.x; e < (uint32_t)(S * S) * cs; e += 256u) {
            const uint32_t pl = e / cs, sj = e % cs;   // consecutive e -> consecutive slots of one pixel's run
            const uint32_t p = s_pos[pl];
            float4 oa = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            float2 ob = make_float2(0.0f, __uint_as_float(0x00000000u));   // an empty box (x1 = x0)
            if (p != RSPT_FG_NONE) {
                const uint32_t slot = p * bt.ns + j0 + sj;
                const float4 le = pb.L_eta[slot];
                const float2 pf = pb.p_film[slot];
                rgb l{le.x, le.y, le.z};
                const int32_t sx = bx - K + (int32_t)pl % S, sy = by - K + (int32_t)pl / S;   // the sample's own pixel
                const bool mine = sx >= bx && sx < bx + (int32_t)RSPT_FG_T && sy >= by && sy < by + (int32_t)RSPT_FG_T;   // counted / reported by exactly one block
                if (has_nans(l)) { l = mkrgb(0.0f); if (mine) nans++; }  // integrator.rs:165-173
                if (li_out && mine && sx >= cp[0] && sx < cp[2] && sy >= cp[1] && sy < cp[3]) {
                    float* o = li_out + (((size_t)(sy - cp[1]) * cw + (size_t)(sx - cp[0])) * (size_t)rd.spp + (size_t)(bt.s0 + j0 + sj)) * 3;
                    o[0] = l.r; o[1] = l.g; o[2] = l.b;
                }
                if (lum(l) > rd.max_sample_luminance) l = l * mkrgb(rd.max_sample_luminance / lum(l));
                const float dx = pf.x - 0.5f, dy = pf.y - 0.5f;
                // the footprint (film.rs:104-116); a FilmTile's own pixel bounds never cut it (they are the tile's sample bounds grown by the radius), the crop window does
                const int32_t x0 = max(f2i_sat(ceilf(dx - rx)), cp[0]) - bx, y0 = max(f2i_sat(ceilf(dy - ry)), cp[1]) - by;
                const int32_t x1 = min(f2i_sat(floorf(dx + rx)) + 1, cp[2]) - bx, y1 = min(f2i_sat(floorf(dy + ry)) + 1, cp[3]) - by;
                auto b8 = [](int32_t v) { return (uint32_t)(v < -128 ? -128 : (v > 127 ? 127 : v)) & 0xffu; };
                oa = make_float4(l.r, l.g, l.b, dx);
                ob = make_float2(dy, __uint_as_float(b8(x0) | (b8(x1) << 8) | (b8(y0) << 16) | (b8(y1) << 24)));
            }
            s_a[pl * row + sj] = oa;
            s_b[pl * row + sj] = ob;
        }
        __syncthreads();
        if (!in_crop) continue;
        for (int32_t dyy = -K; dyy <= K; dyy++)
            for (int32_t dxx = -K; dxx <= K; dxx++) {
                const uint32_t pl = (uint32_t)((ty + K + dyy) * S + (tx + K + dxx));
                if (s_pos[pl] == RSPT_FG_NONE) continue;
                for (uint32_t sj = 0; sj < cs; sj++) {
                    const float2 b = s_b[pl * row + sj];
                    const uint32_t box = __float_as_uint(b.y);
                    const int32_t x0 = (int32_t)(int8_t)(box & 0xffu), x1 = (int32_t)(int8_t)((box >> 8) & 0xffu);
                    const int32_t y0 = (int32_t)(int8_t)((box >> 16) & 0xffu), y1 = (int32_t)(int8_t)(box >> 24);
                    if (tx < x0 || tx >= x1 || ty < y0 || ty >= y1) continue;
                    const float4 a = s_a[pl * row + sj];
                    const float fy = fabsf(((float)y - b.x) * inv_ry * 16.0f);
                    const int32_t ify = (int32_t)fminf(floorf(fy), 15.0f);
                    const float fx = fabsf(((float)x - a.w) * inv_rx * 16.0f);
                    const int32_t ifx = (int32_t)fminf(floorf(fx), 15.0f);
                    const float w = s_tab[ify * 16 + ifx];
                    const rgb c = rgb{a.x, a.y, a.z} * mkrgb(1.0f) * mkrgb(w);  // l * sample_weight * filter_weight
                    acc.x += c.r; acc.y += c.g; acc.z += c.b; acc.w += w;
                }
            }
    }
    if (in_crop) film_own[idx] = acc;
    if (nans) atomicAdd(nan_count, nans);
}

// Film::merge_film_tile (film.rs:346-371): contrib_sum RGB -> XYZ, plus filter weight sum.
// `add` = 1 accumulates into film_out (multi-pass renders), 0 overwrites.
RSPT_PLAIN_KERNEL void k_film_resolve(const float4* __restrict__ film_own, const float4* __restrict__ film_splat, float4* __restrict__ film_out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 a = film_own[i], s = film_splat[i];
    float r = a.x + s.x, g = a.y + s.y, b = a.z + s.z;
    float4 o;
    o.x = 0.412453f * r + 0.357580f * g + 0.180423f * b;  // rgb_to_xyz spectrum.rs:1829-1835
    o.y = 0.212671f * r + 0.715160f * g + 0.072169f * b;
    o.z = 0.019334f * r + 0.119193f * g + 0.950227f * b;
    o.w = a.w + s.w;
    film_out[i] = o;
}

// ---- K9: light sampling distributions (src/core/lightdistrib.rs) ------------------------------
RDEV uint32_t rev32(uint32_t n) {  // lowdiscrepancy.rs:770-779
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
    n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
    n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
    n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
    return n;
}
RDEV float radical_inverse(int base_index, uint64_t a) {  // lowdiscrepancy.rs:1082-1135
    if (base_index == 0) {
        uint64_t r = ((uint64_t)rev32((uint32_t)a) << 32) | (uint64_t)rev32((uint32_t)(a >> 32));
        return (float)r * 0x1.0p-64f;
    }
    const uint64_t base = base_index == 1 ? 3 : (base_index == 2 ? 5 : (base_index == 3 ? 7 : 11));
    float inv_base = 1.0f / (float)base;
    uint64_t reversed = 0;
    float inv_base_n = 1.0f;
    while (a != 0) {
        uint64_t next = a / base, digit = a - next * base;
        reversed = reversed * base + digit;
        inv_base_n *= inv_base;
        a = next;
    }
    return fminf((float)reversed * inv_base_n, RSPT_ONE_MINUS_EPS);
}
// SpatialLightDistribution::compute_distribution, per (voxel, light) (lightdistrib.rs:169-260)
RDEV float ld_voxel_light_contrib(const SceneDev& sc, int32_t nvx, int32_t nvy, int32_t nvz, uint64_t v, uint32_t j) {
    int32_t ix = (int32_t)(v % nvx), iy = (int32_t)((v / nvx) % nvy), iz = (int32_t)(v / ((uint64_t)nvx * nvy));
    f3 wmin{sc.wb_min[0], sc.wb_min[1], sc.wb_min[2]}, wmax{sc.wb_max[0], sc.wb_max[1], sc.wb_max[2]};
    f3 p0{(float)ix / (float)nvx, (float)iy / (float)nvy, (float)iz / (float)nvz};
    f3 p1{(float)(ix + 1) / (float)nvx, (float)(iy + 1) / (float)nvy, (float)(iz + 1) / (float)nvz};
    f3 vmin{lerpf(p0.x, wmin.x, wmax.x), lerpf(p0.y, wmin.y, wmax.y), lerpf(p0.z, wmin.z, wmax.z)};
    f3 vmax{lerpf(p1.x, wmin.x, wmax.x), lerpf(p1.y, wmin.y, wmax.y), lerpf(p1.z, wmin.z, wmax.z)};
    const rspt_light lt = sc.lights[j];
    float contrib = 0.0f;
    for (uint64_t i = 0; i < 128; i++) {
        f3 t{radical_inverse(0, i), radical_inverse(1, i), radical_inverse(2, i)};
        f3 po{lerpf(t.x, vmin.x, vmax.x), lerpf(t.y, vmin.y, vmax.y), lerpf(t.z, vmin.z, vmax.z)};
        f2 u{radical_inverse(3, i), radical_inverse(4, i)};
        float pdf = 0.0f;
        f3 wi{0.0f, 0.0f, 0.0f};
        LightSample ls;
        rgb li = light_sample_li(sc, lt, po, u, &wi, &pdf, &ls);
        if (pdf > 0.0f) contrib += lum(li) / pdf;
    }
    return contrib;
}
RSPT_PLAIN_KERNEL void k_ld_contrib(SceneDev sc, int32_t nvx, int32_t nvy, int32_t nvz, float* __restrict__ func) {
    uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t total = (uint64_t)nvx * nvy * nvz * sc.n_lights;
    if (gid >= total) return;
    func[gid] = ld_voxel_light_contrib(sc, nvx, nvy, nvz, gid / sc.n_lights, (uint32_t)(gid % sc.n_lights));
}
// one Distribution1D per voxel (sampling.rs:24-49) after the min-contribution clamp (:262-268);
// mode 0: spatial (func holds raw contributions), 1: use func as is (uniform / power)
RDEV void ld_build_row(uint32_t nl, int mode, float* f, float* c, float* func_int) {
    if (mode == 0) {
        float sum = 0.0f;
        for (uint32_t j = 0; j < nl; j++) sum += f[j];
        float avg = sum / (float)(128u * nl);
        float min_contrib = avg > 0.0f ? 0.001f * avg : 1.0f;
        for (uint32_t j = 0; j < nl; j++) f[j] = fmaxf(f[j], min_contrib);
    }
    c[0] = 0.0f;
    for (uint32_t i = 1; i <= nl; i++) c[i] = c[i - 1] + f[i - 1] / (float)nl;
    float fi = c[nl];
    if (fi == 0.0f) for (uint32_t i = 1; i <= nl; i++) c[i] = (float)i / (float)nl;
    else for (uint32_t i = 1; i <= nl; i++) c[i] /= fi;
    *func_int = fi;
}
RSPT_PLAIN_KERNEL void k_ld_build(uint32_t n_vox, uint32_t nl, int mode, float* __restrict__ func, float* __restrict__ cdf, float* __restrict__ func_int) {
    uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n_vox) return;
    ld_build_row(nl, mode, func + (size_t)v * nl, cdf + (size_t)v * (nl + 1), func_int + v);
}
// uniform: func = 1; power: func = Light::power().y() (integrator.rs:573-584, diffuse.rs:85-93)
RSPT_PLAIN_KERNEL void k_ld_fixed(SceneDev sc, int power, float* __restrict__ func) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= sc.n_lights) return;
    if (!power) { func[j] = 1.0f; return; }
    func[j] = lum(light_power(sc, sc.lights[j]));
}

// ---- on-demand voxels of the spatial distribution (the eager table is n_vox x n_lights) ---------------------------
// Before a shade launch: every path about to look a voxel up (a surface hit below the depth limit, with a material) claims the
// voxel if nobody has; the claimed voxels get their rows built by the kernels above (same arithmetic as the eager build, so the
// values do not depend on when a voxel is built, Q18) and are published in the table.
struct LightLazy {
    uint32_t n_new;      // voxels claimed in this round
    uint32_t n_rows;     // rows handed out so far
    uint32_t max_rows;
    uint32_t overflow;   // a voxel could not get a row: the render fails with RSPT_E_NOMEM
};
RSPT_PLAIN_KERNEL __launch_bounds__(256) void k_ld_mark(SceneDev sc, LightDistDev ld, PathBuf pb, uint32_t max_depth, const uint32_t* __restrict__ q_active,
                                                 const QueueCounts* __restrict__ cnt_in, LightLazy* lz, uint32_t* __restrict__ new_list) {
    const uint32_t n = cnt_in->active;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t p = q_active[i];
        const uint32_t st = pb.state[p];
        const float4 hc = pb.hit_cont[p];
        const uint32_t prim = __float_as_uint(hc.x);
        if (!(st & ST_ALIVE) || prim == RSPT_MISS || ((st >> ST_BOUNCE_SHIFT) & 0xffu) >= max_depth) continue;
        const TriRec t = load_tri(sc, prim);
        if (t.material == 0xffffffffu) continue;
        f3 hp = t.p0 * hc.y + t.p1 * hc.z + t.p2 * hc.w;  // the hit point as tri_fill forms it
        if (pb.hit_inst) {
            const uint32_t hi = pb.hit_inst[p];
            if (hi && sc.inst[hi - 1u].anim != RSPT_MISS) {
                const InstDev at = inst_at(sc, hi - 1u, sc.ray_time ? sc.ray_time[p] : 0.0f);
                if (!at.identity) {
                    if (!sc.inst_fixed) continue;
                    f3 pe;
                    inst_point(at.m, at.m3, hp, f3{0.0f, 0.0f, 0.0f}, &hp, &pe);
                }
            } else if (hi && !sc.inst[hi - 1u].identity) {
                if (!sc.inst_fixed) continue;
                f3 pe;
                inst_point(sc.inst[hi - 1u].m, sc.inst[hi - 1u].m3, hp, f3{0.0f, 0.0f, 0.0f}, &hp, &pe);
            }
        }
        const uint32_t vox = light_voxel(sc, ld, hp);
        if (ld.table[vox] == -1 && atomicCAS(&ld.table[vox], -1, -2) == -1) {
            const uint32_t k = atomicAdd(&lz->n_new, 1u);
            if (lz->n_rows + k < lz->max_rows) new_list[k] = vox;
            else { lz->overflow = 1u; ld.table[vox] = -1; }
        }
    }
}
// contributions of every light to the claimed voxels: the body of k_ld_contrib with (row, voxel) from the list
RSPT_PLAIN_KERNEL void k_ld_contrib_list(SceneDev sc, int32_t nvx, int32_t nvy, int32_t nvz, const LightLazy* lz, const uint32_t* __restrict__ new_list, float* __restrict__ func) {
    const uint64_t n_new = lz->n_new < lz->max_rows - lz->n_rows ? lz->n_new : lz->max_rows - lz->n_rows;
    const uint64_t total = n_new * sc.n_lights;
    for (uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t j = (uint32_t)(gid % sc.n_lights);
        const uint64_t k = gid / sc.n_lights;
        const uint64_t v = new_list[k];
        func[((uint64_t)lz->n_rows + k) * sc.n_lights + j] = ld_voxel_light_contrib(sc, nvx, nvy, nvz, v, j);
    }
}
RSPT_PLAIN_KERNEL void k_ld_build_list(uint32_t nl, const LightLazy* lz, const uint32_t* __restrict__ new_list, float* __restrict__ func, float* __restrict__ cdf,
                                float* __restrict__ func_int, int32_t* __restrict__ table) {
    const uint32_t n_new = lz->n_new < lz->max_rows - lz->n_rows ? lz->n_new : lz->max_rows - lz->n_rows;
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n_new; k += gridDim.x * blockDim.x) {
        const uint32_t row = lz->n_rows + k;
        ld_build_row(nl, 0, func + (size_t)row * nl, cdf + (size_t)row * (nl + 1), func_int + row);
        table[new_list[k]] = (int32_t)row;
    }
}
RSPT_PLAIN_KERNEL void k_ld_commit(LightLazy* lz) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t n_new = lz->n_new < lz->max_rows - lz->n_rows ? lz->n_new : lz->max_rows - lz->n_rows;
    lz->n_rows += n_new;
    lz->n_new = 0u;
}

// scene upload helper: build the 48-byte triangle records from the indexed ABI arrays
RSPT_PLAIN_KERNEL void k_build_tris(const rspt_prim* __restrict__ prims, const rspt_mesh* __restrict__ meshes, const float* __restrict__ P, uint32_t n,
                             float4* __restrict__ tris, const uint32_t* __restrict__ inst_cont, const uint32_t* __restrict__ mesh_mask) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    rspt_prim pr = prims[i];
    if (pr.mesh == RSPT_MESH_INSTANCE) {  // a TransformedPrimitive: instance index and (four-box kernel) the reference to the rest of its leaf
        tris[3 * (size_t)i] = make_float4(__uint_as_float(pr.v[0]), __uint_as_float(inst_cont ? inst_cont[pr.v[0]] : 0xffffffffu), 0.0f, 0.0f);
        tris[3 * (size_t)i + 1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        tris[3 * (size_t)i + 2] = make_float4(0.0f, __uint_as_float(0xffffffffu), __uint_as_float(0xffffffffu), __uint_as_float((uint32_t)MF_INSTANCE));
        return;
    }
    rspt_mesh m = meshes[pr.mesh];
    f3 p0 = ld3(P, pr.v[0]), p1 = ld3(P, pr.v[1]), p2 = ld3(P, pr.v[2]);
    uint32_t flags = (m.has_n ? MF_HAS_N : 0u) | (m.has_s ? MF_HAS_S : 0u) | (m.has_uv ? MF_HAS_UV : 0u) | (m.flip ? MF_FLIP : 0u) |
                     ((m.alpha_tex || m.shadow_alpha_tex) ? MF_ALPHA : 0u) | (mesh_mask ? mesh_mask[pr.mesh] << MF_MASK_SHIFT : 0u);
    tris[3 * (size_t)i] = make_float4(p0.x, p0.y, p0.z, p1.x);
    tris[3 * (size_t)i + 1] = make_float4(p1.y, p1.z, p2.x, p2.y);
    tris[3 * (size_t)i + 2] = make_float4(p2.z, __uint_as_float(pr.material), __uint_as_float((uint32_t)pr.area_light), __uint_as_float(flags));
}

// stage hook rspt_libm: one of the device's libm restatements (glibc_libm.h) over an array
RSPT_PLAIN_KERNEL void k_libm(uint32_t fn, const float* __restrict__ x, const float* __restrict__ y, uint64_t n, float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fn == 7u) {   // RSPT_LIBM_MAT4_INVERSE: Matrix4x4::inverse (mat4_inverse.h) of the i-th 4x4 matrix
        float a[16], b[16];
#pragma unroll
        for (int k = 0; k < 16; k++) a[k] = x[16 * i + k];
        mat4_inverse(a, b);
#pragma unroll
        for (int k = 0; k < 16; k++) out[16 * i + k] = b[k];
        return;
    }
    const float v = x[i];
    out[i] = fn == 0 ? rspt_sinf(v) : fn == 1 ? rspt_cosf(v) : fn == 2 ? rspt_logf(v) : fn == 3 ? rspt_log2f(v) : fn == 4 ? rspt_expf(v) : fn == 5 ? rspt_acosf(v) : rspt_atan2f(v, y[i]);
}

}  // namespace rspt
