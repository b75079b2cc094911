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
#define RSPT_PW_BLOCK 256
#define RSPT_PW_LDS 16       // stack entries per lane in LDS (4 B each); deeper levels spill to scratch
#define RSPT_PW_CHUNK 256    // rays a wave claims per global atomic
#define RSPT_PW_REFILL 16    // refill when at least this many lanes are idle ...
#define RSPT_PW_LEAF 8       // ... and run the leaf phase when at least this many lanes hold a leaf

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

typedef float v2f __attribute__((ext_vector_type(2)));

// Bounds3f::intersect_p (geometry.rs:2211-2269) for the two children of a pair at once, valid when
// every reciprocal direction component is finite (no 0*inf NaNs).  With near_a = (near plane - o)*inv
// and far_a = (far plane - o)*inv*(1+2*gamma(3)) per axis, the reference computes
//     x/y cross checks, t_min = max(near_x, near_y), t_max = min(far_x, far_y), z cross checks,
//     t_min = max(t_min, near_z), t_max = min(t_max, far_z), result = t_min < ray.t_max && t_max > 0.
// The cross checks are exactly the six conditions near_a <= far_b (a != b).  They differ from
// max3(near) <= min3(far) only in the same-axis pairs near_a <= far_a, and those can fail only when
// far_a < 0 (widening a negative far value moves it below near), where the final t_max > 0 test
// rejects anyway.  So hit == (max3(near) <= min3(far)) && max3(near) < ray.t_max && min3(far) > 0.
RDEV void box_pair_hit(v2f nx, v2f fx, v2f ny, v2f fy, v2f nz, v2f fz, f3 o, f3 inv, float ray_tmax, bool* h0, bool* h1) {
    const float widen = 1.0f + 2.0f * gamma_n(3);
    v2f tnx = (nx - o.x) * inv.x, tny = (ny - o.y) * inv.y, tnz = (nz - o.z) * inv.z;
    v2f tfx = (fx - o.x) * inv.x, tfy = (fy - o.y) * inv.y, tfz = (fz - o.z) * inv.z;
    tfx = tfx * widen; tfy = tfy * widen; tfz = tfz * widen;
    float m0 = fmaxf(fmaxf(tnx.x, tny.x), tnz.x), M0 = fminf(fminf(tfx.x, tfy.x), tfz.x);
    float m1 = fmaxf(fmaxf(tnx.y, tny.y), tnz.y), M1 = fminf(fminf(tfx.y, tfy.y), tfz.y);
    *h0 = (m0 <= M0) && (m0 < ray_tmax) && (M0 > 0.0f);
    *h1 = (m1 <= M1) && (m1 < ray_tmax) && (M1 > 0.0f);
}

template <bool ANY, int OUT_MODE>
__global__ __launch_bounds__(RSPT_PW_BLOCK) void k_trace_pw(SceneDev sc, const PairNode* __restrict__ pairs, const uint32_t* __restrict__ queue,
                                                           const uint32_t* __restrict__ count_ptr, uint32_t count_imm, uint32_t* cursor,
                                                           const rspt_ray* __restrict__ rays_a, const rspt_ray* __restrict__ rays_b,
                                                           float4* __restrict__ out_a, float4* __restrict__ out_b, uint32_t* __restrict__ out_occ,
                                                           rspt_hit* __restrict__ out_hits, int refill_thresh, int leaf_thresh) {
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
    f3 o{0, 0, 0}, inv{0, 0, 0};
    RayShear rs{0, 0, 0, 0, 0, 0};
    float t_max = 0.0f;
    bool ng0 = false, ng1 = false, ng2 = false, degenerate = false;
    uint32_t sp = 0, stale_sp = 0, cur = RSPT_NONE, leaf_node = RSPT_NONE;
    uint32_t best = RSPT_MISS, entry = 0, qpos = 0;
    float bt = 0.0f, bb0 = 0.0f, bb1 = 0.0f, bb2 = 0.0f;
    uint32_t spill[64 - RSPT_PW_LDS];

    auto finish = [&]() {
        uint32_t slot = entry & ~RSPT_Q_MIS;
        if (OUT_MODE == 0) {
            if (ANY) out_occ[slot] = best != RSPT_MISS ? 1u : 0u;
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
        const uint64_t busy = ~idle;
        if (!exhausted && (__popcll(idle) >= refill_thresh || busy == 0)) {
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
                    o = f3{r0.x, r0.y, r0.z};
                    f3 d{r0.w, r1.x, r1.y};
                    t_max = r1.z;
                    inv = f3{1.0f / d.x, 1.0f / d.y, 1.0f / d.z};
                    ng0 = inv.x < 0.0f; ng1 = inv.y < 0.0f; ng2 = inv.z < 0.0f;
                    // zero / denormal / NaN direction components: keep the reference's literal compare chain
                    degenerate = !(fabsf(inv.x) < RSPT_INF && fabsf(inv.y) < RSPT_INF && fabsf(inv.z) < RSPT_INF);
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
                    uint32_t ref = sp < RSPT_PW_LDS ? my[sp * RSPT_PW_BLOCK] : spill[sp - RSPT_PW_LDS];
                    stale = sp < stale_sp;
                    if (stale) stale_sp = sp;
                    if (ref & RSPT_REF_LEAF) {
                        uint32_t li = ref & ~RSPT_REF_LEAF;
                        bool ok = true;
                        if (stale) {
                            float4 n0 = sc.nodes[2 * (size_t)li], n1 = sc.nodes[2 * (size_t)li + 1];
                            ok = box_hit(n0, n1, o, inv, ng0, ng1, ng2, t_max);
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
                    ok = box_hit(n0, n1, o, inv, ng0, ng1, ng2, t_max);
                }
                cur = RSPT_NONE;
                if (ok) {
                    bool h0, h1;
                    if (!degenerate) {
                        // Both children's slab tests on packed f32 pairs (v_pk_add_f32 / v_pk_mul_f32), one
                        // pair = (child 0, child 1).  For finite reciprocals the reference's compare-and-
                        // select chain equals max3/min3 (see box_pair_hit): same booleans, ~40 % of the VALU.
                        v2f nx = ng0 ? v2f{q0.z, q0.w} : v2f{q0.x, q0.y}, fx = ng0 ? v2f{q0.x, q0.y} : v2f{q0.z, q0.w};
                        v2f ny = ng1 ? v2f{q1.z, q1.w} : v2f{q1.x, q1.y}, fy = ng1 ? v2f{q1.x, q1.y} : v2f{q1.z, q1.w};
                        v2f nz = ng2 ? v2f{q2.z, q2.w} : v2f{q2.x, q2.y}, fz = ng2 ? v2f{q2.x, q2.y} : v2f{q2.z, q2.w};
                        box_pair_hit(nx, fx, ny, fy, nz, fz, o, inv, t_max, &h0, &h1);
                    } else {
                        h0 = box_hit6(q0.x, q1.x, q2.x, q0.z, q1.z, q2.z, o, inv, ng0, ng1, ng2, t_max);
                        h1 = box_hit6(q0.y, q1.y, q2.y, q0.w, q1.w, q2.w, o, inv, ng0, ng1, ng2, t_max);
                    }
                    uint32_t axis = __float_as_uint(q3.w);
                    bool neg = axis == 0 ? ng0 : (axis == 1 ? ng1 : ng2);
                    uint32_t c0 = __float_as_uint(q3.x), c1 = __float_as_uint(q3.y);
                    uint32_t near_ref = neg ? c1 : c0, far_ref = neg ? c0 : c1;
                    bool near_hit = neg ? h1 : h0, far_hit = neg ? h0 : h1;
                    uint32_t next = RSPT_NONE;
                    if (near_hit) {
                        next = near_ref;
                        if (far_hit) {
                            if (sp < RSPT_PW_LDS) my[sp * RSPT_PW_BLOCK] = far_ref;
                            else spill[sp - RSPT_PW_LDS] = far_ref;
                            sp++;
                        }
                    } else if (far_hit)
                        next = far_ref;  // == push + immediate pop of a fresh entry
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

}  // namespace rspt
