// k_trace_pw — the production traversal kernel (K2/K3): persistent waves with dynamic ray fetch,
// two-children-per-record BVH nodes, deferred leaf phase, LDS-staged stack.
//
// It visits, for every ray, exactly the node / triangle sequence of BVHAccel::intersect /
// intersect_p (bvh.rs:401-514) — near child first by dir_is_neg[axis], far child deferred, the
// shrinking t_max applied to every later box and triangle test — so (prim, t, b0, b1, b2) are
// bit-identical to the reference, ties included.  What differs is only HOW the sequence is
// produced on wave64 hardware:
//   * BVH2 "pair" records (64 B, one per interior node): both children's boxes + child refs, so a
//     traversal step costs one dependent 64-byte fetch instead of two dependent 32-byte node
//     fetches.  The far child's box is tested when its parent is processed (with the t_max of
//     that moment); it is pushed only if it passes.  Because Bounds3f::intersect_p depends on
//     ray.t_max only through the final `t_min < ray.t_max`, a box that fails early also fails
//     later; a box that passed is re-tested on pop (against its LinearBVHNode) only if a hit
//     shrank t_max after the push — tracked with one watermark (`stale_sp`), not per entry.
//   * Leaf hits are rare next to box tests, so a lane that reaches a leaf parks (it does not
//     run ahead: order is preserved) until enough lanes of the wave hold leaves; the expensive
//     watertight triangle code then runs with many lanes instead of on every iteration.
//   * Rays have wildly different traversal lengths; lanes that finish pull new rays from a
//     wave-local chunk of the queue (one global atomic per RSPT_PW_CHUNK rays) instead of idling
//     until the slowest ray of their wave is done.
//   * The stack lives entirely in LDS (column per lane, conflict-free).  A ray that would need
//     more than RSPT_PW_LDS entries is marked and re-traced by k_trace_fixup with the 64-entry
//     reference-order loop (mixing LDS and scratch in one stack made the compiler emit flat loads
//     with full vmcnt/lgkmcnt drains on every pop).
#pragma once
#include "kernels.h"

namespace rspt {

struct PairNode {       // 64 B, 64-byte aligned
    float4 q0;          // x slabs: c0.min.x, c1.min.x, c0.max.x, c1.max.x  (children interleaved so that
    float4 q1;          // y slabs: c0.min.y, c1.min.y, c0.max.y, c1.max.y   each (c0, c1) pair feeds one
    float4 q2;          // z slabs: c0.min.z, c1.min.z, c0.max.z, c1.max.z   packed-f32 instruction)
    uint32_t c0, c1;    // child refs: bit 31 = leaf; low bits = pair index (interior) / LinearBVHNode index (leaf)
    uint32_t self;      // LinearBVHNode index of this interior node
    uint32_t axis;
};
#define RSPT_REF_LEAF 0x80000000u
#define RSPT_NONE 0xffffffffu
#define RSPT_RETRACE 0xfffffffeu  // result marker: stack overflow, k_trace_fixup redoes this ray
#define RSPT_PW_BLOCK 256
#ifndef RSPT_PW_LDS
#define RSPT_PW_LDS 24       // stack entries per lane (4 B each, all in LDS): 24 KB per workgroup, 6 workgroups per CU
                             // (measured on C2 / C3: 16 -> 256 / 882, 24 -> 259 / 895, 32 -> 239 / 834 Msamples/s)
#endif
#define RSPT_PW_CHUNK 256    // rays a wave claims per global atomic
#define RSPT_PW_REFILL 16    // refill when at least this many lanes are idle ...
#define RSPT_PW_LEAF 8       // ... and run the leaf phase when at least this many lanes hold a leaf

typedef float v2f __attribute__((ext_vector_type(2)));

// Bounds3f::intersect_p (geometry.rs:2211-2269) for the two children of a pair at once, valid when
// every reciprocal direction component is finite (no 0*inf NaNs).  Per axis the reference forms
//     near = ((dir_is_neg ? max : min) - o) * inv,   far = ((dir_is_neg ? min : max) - o) * inv * (1 + 2 gamma(3))
// With t_lo = (min - o)*inv and t_hi = (max - o)*inv (the same two products), near = min(t_lo, t_hi)
// and far = max(t_lo, t_hi) * widen: multiplication by a finite inv is monotone, so the sign of inv
// decides the order of t_lo and t_hi exactly as it decides the reference's select.
// The reference's compare-and-select chain (x/y cross checks, merge, z cross checks, merge,
// t_min < ray.t_max && t_max > 0) equals max3(near) <= min3(far) && max3(near) < ray.t_max &&
// min3(far) > 0: the cross checks are the six conditions near_a <= far_b (a != b); the same-axis
// pairs near_a <= far_a can fail only when far_a < 0 (widening a negative value moves it below
// near), where the final t_max > 0 test rejects anyway.
RDEV void box_pair_hit(float4 q0, float4 q1, float4 q2, float ox, float oy, float oz, float ix, float iy, float iz, float ray_tmax, bool* h0, bool* h1) {
    const float widen = 1.0f + 2.0f * gamma_n(3);
    v2f lx = (v2f{q0.x, q0.y} - ox) * ix, hx = (v2f{q0.z, q0.w} - ox) * ix;
    v2f ly = (v2f{q1.x, q1.y} - oy) * iy, hy = (v2f{q1.z, q1.w} - oy) * iy;
    v2f lz = (v2f{q2.x, q2.y} - oz) * iz, hz = (v2f{q2.z, q2.w} - oz) * iz;
    v2f fx = v2f{fmaxf(lx.x, hx.x), fmaxf(lx.y, hx.y)} * widen;
    v2f fy = v2f{fmaxf(ly.x, hy.x), fmaxf(ly.y, hy.y)} * widen;
    v2f fz = v2f{fmaxf(lz.x, hz.x), fmaxf(lz.y, hz.y)} * widen;
    float m0 = fmaxf(fmaxf(fminf(lx.x, hx.x), fminf(ly.x, hy.x)), fminf(lz.x, hz.x)), M0 = fminf(fminf(fx.x, fy.x), fz.x);
    float m1 = fmaxf(fmaxf(fminf(lx.y, hx.y), fminf(ly.y, hy.y)), fminf(lz.y, hz.y)), M1 = fminf(fminf(fx.y, fy.y), fz.y);
    *h0 = (m0 <= M0) && (m0 < ray_tmax) && (M0 > 0.0f);
    *h1 = (m1 <= M1) && (m1 < ray_tmax) && (M1 > 0.0f);
}
// literal reference chain for rays with zero / non-finite direction components
RDEV bool box_hit6(float lx, float ly, float lz, float hx, float hy, float hz, f3 o, f3 inv, bool ng0, bool ng1, bool ng2, float ray_tmax) {
    const float widen = 1.0f + 2.0f * gamma_n(3);
    float t_min = ((ng0 ? hx : lx) - o.x) * inv.x;
    float t_max = ((ng0 ? lx : hx) - o.x) * inv.x;
    float ty_min = ((ng1 ? hy : ly) - o.y) * inv.y;
    float ty_max = ((ng1 ? ly : hy) - o.y) * inv.y;
    t_max *= widen;
    ty_max *= widen;
    if (t_min > ty_max || ty_min > t_max) return false;
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_max) t_max = ty_max;
    float tz_min = ((ng2 ? hz : lz) - o.z) * inv.z;
    float tz_max = ((ng2 ? lz : hz) - o.z) * inv.z;
    tz_max *= widen;
    if (t_min > tz_max || tz_min > t_max) return false;
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_max) t_max = tz_max;
    return (t_min < ray_tmax) && (t_max > 0.0f);
}

template <bool ANY, int OUT_MODE>
__global__ __launch_bounds__(RSPT_PW_BLOCK) void k_trace_pw(SceneDev sc, const PairNode* __restrict__ pairs, const uint32_t* __restrict__ queue,
                                                           const uint32_t* __restrict__ count_ptr, uint32_t count_imm, uint32_t* cursor,
                                                           const rspt_ray* __restrict__ rays_a, const rspt_ray* __restrict__ rays_b,
                                                           float4* __restrict__ out_a, float4* __restrict__ out_b, uint32_t* __restrict__ out_occ,
                                                           rspt_hit* __restrict__ out_hits, uint32_t* n_overflow, uint32_t* __restrict__ overflow_list,
                                                           int refill_thresh, int leaf_thresh) {
    __shared__ uint32_t stack[RSPT_PW_LDS * RSPT_PW_BLOCK];
    uint32_t* my = stack + threadIdx.x;
    const uint32_t n = count_ptr ? *count_ptr : count_imm;
    if (sc.n_nodes == 0) {  // empty scene: every ray misses
        for (uint32_t i = blockIdx.x * RSPT_PW_BLOCK + threadIdx.x; i < n; i += gridDim.x * RSPT_PW_BLOCK) {
            uint32_t e = queue ? queue[i] : i, slot = e & ~RSPT_Q_MIS;
            if (OUT_MODE == 0) {
                if (ANY) out_occ[slot] = 0u;
                else ((e & RSPT_Q_MIS) ? out_b : out_a)[slot] = make_float4(__uint_as_float(RSPT_MISS), 0.0f, 0.0f, 0.0f);
            } else {
                rspt_hit h; h.prim = RSPT_MISS; h.t = h.b0 = h.b1 = h.b2 = 0.0f;
                out_hits[i] = h;
            }
        }
        return;
    }
    const uint32_t lane = __lane_id();
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t root_ref = sc.n_nodes == 1 ? (0u | RSPT_REF_LEAF) : 0u;
    uint32_t chunk_lo = 0, chunk_hi = 0;  // wave-uniform
    bool exhausted = false;               // wave-uniform
    // per-lane ray state
    bool active = false;
    float ox = 0, oy = 0, oz = 0, ix = 0, iy = 0, iz = 0;
    RayShear rs{0, 0, 0, 0, 0, 0};
    float t_max = 0.0f;
    uint32_t negbits = 0;  // bit a = dir_is_neg[a]; bit 3 = zero / non-finite direction component
    uint32_t sp = 0, stale_sp = 0, cur = RSPT_NONE, leaf_node = RSPT_NONE;
    uint32_t best = RSPT_MISS, entry = 0, qpos = 0;
    float bt = 0.0f, bb0 = 0.0f, bb1 = 0.0f, bb2 = 0.0f;

    auto finish = [&]() {
        uint32_t slot = entry & ~RSPT_Q_MIS;
        if (OUT_MODE == 0) {
            if (ANY) out_occ[slot] = best == RSPT_RETRACE ? 2u : (best != RSPT_MISS ? 1u : 0u);
            else ((entry & RSPT_Q_MIS) ? out_b : out_a)[slot] = make_float4(__uint_as_float(best), bb0, bb1, bb2);
        } else {
            rspt_hit h;
            h.prim = best; h.t = bt; h.b0 = bb0; h.b1 = bb1; h.b2 = bb2;
            out_hits[qpos] = h;
        }
        active = false;
    };

    for (;;) {
        // ---- refill idle lanes from the wave's chunk ----
        const uint64_t idle = __ballot(!active);
        if (!exhausted && (__popcll(idle) >= refill_thresh || ~idle == 0)) {
            if (chunk_lo == chunk_hi) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(cursor, (uint32_t)RSPT_PW_CHUNK);
                base = __builtin_amdgcn_readfirstlane(base);
                chunk_lo = base < n ? base : n;
                chunk_hi = (base + RSPT_PW_CHUNK) < n ? (base + RSPT_PW_CHUNK) : n;
                if (chunk_lo == chunk_hi) exhausted = true;
            }
            if (!exhausted) {
                const uint32_t avail = chunk_hi - chunk_lo;
                const uint32_t rank = (uint32_t)__popcll(idle & lt_mask);
                if (!active && rank < avail) {
                    qpos = chunk_lo + rank;
                    entry = queue ? queue[qpos] : qpos;
                    const float4* rp = reinterpret_cast<const float4*>(((entry & RSPT_Q_MIS) ? rays_b : rays_a) + (entry & ~RSPT_Q_MIS));
                    float4 r0 = rp[0], r1 = rp[1];
                    ox = r0.x; oy = r0.y; oz = r0.z;
                    f3 d{r0.w, r1.x, r1.y};
                    t_max = r1.z;
                    ix = 1.0f / d.x; iy = 1.0f / d.y; iz = 1.0f / d.z;
                    negbits = (ix < 0.0f ? 1u : 0u) | (iy < 0.0f ? 2u : 0u) | (iz < 0.0f ? 4u : 0u);
                    // zero / denormal / NaN direction components: keep the reference's literal compare chain
                    if (!(fabsf(ix) < RSPT_INF && fabsf(iy) < RSPT_INF && fabsf(iz) < RSPT_INF)) negbits |= 8u;
                    rs = ray_shear(d);
                    best = RSPT_MISS; bt = bb0 = bb1 = bb2 = 0.0f;
                    my[0] = root_ref;   // the root enters as a stale entry: its own box is tested first
                    sp = 1; stale_sp = 1; cur = RSPT_NONE; leaf_node = RSPT_NONE;
                    active = true;
                }
                const uint32_t want = (uint32_t)__popcll(idle);
                chunk_lo += want < avail ? want : avail;
            }
        }
        if (__ballot(active) == 0) {
            if (exhausted) break;
            continue;
        }

        // ---- node phase: one traversal step for every lane that is not parked at a leaf ----
        if (active && leaf_node == RSPT_NONE) {
            uint32_t pidx = cur;
            bool stale = false;
            if (pidx == RSPT_NONE) {
                if (sp == 0) {
                    finish();
                } else {
                    sp--;
                    uint32_t ref = my[sp * RSPT_PW_BLOCK];
                    stale = sp < stale_sp;
                    stale_sp = stale ? sp : stale_sp;
                    if (ref & RSPT_REF_LEAF) {
                        uint32_t li = ref & ~RSPT_REF_LEAF;
                        bool ok = true;
                        if (stale) {
                            float4 n0 = sc.nodes[2 * (size_t)li], n1 = sc.nodes[2 * (size_t)li + 1];
                            ok = box_hit(n0, n1, f3{ox, oy, oz}, f3{ix, iy, iz}, negbits & 1u, negbits & 2u, negbits & 4u, t_max);
                        }
                        if (ok) leaf_node = li;
                    } else
                        pidx = ref;
                }
            }
            if (pidx != RSPT_NONE) {
                const float4* pp = reinterpret_cast<const float4*>(pairs + pidx);
                float4 q0 = pp[0], q1 = pp[1], q2 = pp[2], q3 = pp[3];
                bool ok = true;
                if (stale) {  // t_max shrank since this node was pushed: redo its own box test (bvh.rs:424)
                    uint32_t self = __float_as_uint(q3.z);
                    float4 n0 = sc.nodes[2 * (size_t)self], n1 = sc.nodes[2 * (size_t)self + 1];
                    ok = box_hit(n0, n1, f3{ox, oy, oz}, f3{ix, iy, iz}, negbits & 1u, negbits & 2u, negbits & 4u, t_max);
                }
                cur = RSPT_NONE;
                if (ok) {
                    bool h0, h1;
                    if (!(negbits & 8u)) {
                        box_pair_hit(q0, q1, q2, ox, oy, oz, ix, iy, iz, t_max, &h0, &h1);
                    } else {
                        h0 = box_hit6(q0.x, q1.x, q2.x, q0.z, q1.z, q2.z, f3{ox, oy, oz}, f3{ix, iy, iz}, negbits & 1u, negbits & 2u, negbits & 4u, t_max);
                        h1 = box_hit6(q0.y, q1.y, q2.y, q0.w, q1.w, q2.w, f3{ox, oy, oz}, f3{ix, iy, iz}, negbits & 1u, negbits & 2u, negbits & 4u, t_max);
                    }
                    const bool neg = ((negbits >> __float_as_uint(q3.w)) & 1u) != 0;  // dir_is_neg[axis]
                    const uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
                    const uint32_t near_ref = neg ? c1 : c0, far_ref = neg ? c0 : c1;
                    const bool near_hit = (neg && h1) || (!neg && h0), far_hit = (neg && h0) || (!neg && h1);
                    uint32_t next = near_hit ? near_ref : (far_hit ? far_ref : RSPT_NONE);  // far alone == push + pop of a fresh entry
                    if (near_hit && far_hit) {
                        if (sp < RSPT_PW_LDS) {
                            my[sp * RSPT_PW_BLOCK] = far_ref;
                            sp++;
                        } else {  // deeper than the LDS stack: hand the ray to k_trace_fixup
                            best = RSPT_RETRACE;
                            overflow_list[atomicAdd(n_overflow, 1u)] = OUT_MODE == 0 ? entry : qpos;
                            finish();
                            next = RSPT_NONE;
                        }
                    }
                    if (next != RSPT_NONE) {
                        if (next & RSPT_REF_LEAF) leaf_node = next & ~RSPT_REF_LEAF;
                        else cur = next;
                    }
                }
            }
        }

        // ---- leaf phase: watertight triangle tests for parked lanes, when enough of them wait ----
        const uint64_t parked = __ballot(active && leaf_node != RSPT_NONE);
        if (parked) {
            const uint64_t running = __ballot(active && leaf_node == RSPT_NONE);
            if (__popcll(parked) >= leaf_thresh || running == 0) {
                if (active && leaf_node != RSPT_NONE) {
                    float4 n1 = sc.nodes[2 * (size_t)leaf_node + 1];
                    uint32_t w = __float_as_uint(n1.w);
                    uint32_t n_prims = w & 0xffffu, offset = __float_as_uint(n1.z);
                    leaf_node = RSPT_NONE;
                    const f3 o{ox, oy, oz};
                    for (uint32_t i = 0; i < n_prims; i++) {
                        uint32_t pi = offset + i;
                        float4 a = sc.tris[3 * (size_t)pi], b = sc.tris[3 * (size_t)pi + 1], c = sc.tris[3 * (size_t)pi + 2];
                        float t, b0, b1, b2;
                        if (tri_test(f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c.x}, o, rs, t_max, &t, &b0, &b1, &b2)) {
                            if (ANY) { best = 0; break; }
                            t_max = t;       // primitive.rs:155
                            stale_sp = sp;   // every entry on the stack was pushed against the old t_max
                            best = pi; bt = t; bb0 = b0; bb1 = b1; bb2 = b2;
                        }
                    }
                    if (ANY && best != RSPT_MISS) finish();
                }
            }
        }
    }
}

// Second pass for the rays whose stack outgrew the persistent kernel's LDS column (their queue entries
// were appended to overflow_list): the 64-entry reference-order loop (kernels.h traverse<>).
// Returns at once when no ray overflowed (the common case).
template <bool ANY, int OUT_MODE, bool INST, bool ALPHA, bool ANIM = false>
__global__ __launch_bounds__(RSPT_TRACE_BLOCK) void k_trace_fixup(SceneDev sc, TexTables tt, const uint32_t* __restrict__ n_overflow, const uint32_t* __restrict__ overflow_list,
                                                                  const rspt_ray* __restrict__ rays_a, const rspt_ray* __restrict__ rays_b,
                                                                  float4* __restrict__ out_a, float4* __restrict__ out_b, uint32_t* __restrict__ out_occ,
                                                                  rspt_hit* __restrict__ out_hits, uint32_t* __restrict__ out_inst) {
    __shared__ uint32_t stack[RSPT_LDS_STACK * RSPT_TRACE_BLOCK];
    const uint32_t n = *n_overflow;
    for (uint32_t i = blockIdx.x * RSPT_TRACE_BLOCK + threadIdx.x; i < n; i += gridDim.x * RSPT_TRACE_BLOCK) {
        const uint32_t e = overflow_list[i];
        const uint32_t slot = OUT_MODE == 0 ? (e & ~RSPT_Q_MIS) : e;
        const bool mis = OUT_MODE == 0 && (e & RSPT_Q_MIS) != 0;
        const float4* rp = reinterpret_cast<const float4*>((mis ? rays_b : rays_a) + slot);
        float4 r0 = rp[0], r1 = rp[1];
        const float time = (ANIM && OUT_MODE == 0 && sc.ray_time) ? sc.ray_time[slot / sc.time_div] : 0.0f;
        TraceResult res = traverse<ANY, INST, ALPHA, RSPT_TRACE_BLOCK, ANIM>(sc, tt, f3{r0.x, r0.y, r0.z}, f3{r0.w, r1.x, r1.y}, r1.z, stack + threadIdx.x, time);
        if (OUT_MODE == 0) {
            if (ANY) out_occ[slot] = res.prim != RSPT_MISS ? 1u : 0u;
            else {
                (mis ? out_b : out_a)[slot] = make_float4(__uint_as_float(res.prim), (INST && res.prim == RSPT_MISS) ? res.t_end : res.b0, res.b1, res.b2);
                if (INST && !mis && out_inst) out_inst[slot] = res.inst;
            }
        } else {
            rspt_hit h;
            h.prim = res.prim; h.t = res.t; h.b0 = res.b0; h.b1 = res.b1; h.b2 = res.b2;
            out_hits[slot] = h;
        }
    }
}

}  // namespace rspt
