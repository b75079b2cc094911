// k_trace_w4q — the shadow-ray (any-hit) form of k_trace_w4 over 64-byte QUANTISED four-box records (round 6; the record format and the grid test are round 4's
// experiment, experiments/trace_w4_quantised_records.patch, which lost as a replacement of the closest-hit kernel and was never built for shadow rays alone).
//
// BVHAccel::intersect_p (bvh.rs:463-514) answers "does any triangle stop this ray before t_max" and never shrinks t_max, so neither the order in which boxes are
// visited nor their entry distances can change the answer — only WHICH leaves get their triangles tested can.  The reference tests a leaf's triangles iff the leaf's
// own box passes Bounds3f::intersect_p (its ancestors' boxes contain it and the test is monotone in the box: trace_w4.h), so:
//   * interior AND leaf slots are tested conservatively on an 8-bit grid in the record's own frame (origin + one power-of-two cell per axis): four boxes + four
//     refs = 64 bytes = FOUR scattered 16-byte loads per step instead of seven — the unit the plain kernel is short of (L1 lane requests);
//   * a lane that arrives at a leaf first runs the reference's exact test on the leaf's real box (leaf_boxes[first primitive], box_hit = the reference's compare
//     chain) — a superset of box passes could otherwise ADD a triangle the reference never tests (a watertight hit whose leaf box the slab test just misses), and
//     the flag would differ.  With the re-test the tested triangle set is the reference's, so the flag is: byte-identical to k_trace on every ray.
//   * "conservative" covers the arithmetic: near / far = q * (cell * inv) + (org - o) * inv with fused multiply-adds, lowered / raised per axis by
//     D = 2^-19 * (largest |coordinate| of the scene + largest |o|) * |inv| — eight times the sum of the rounding bounds of both evaluations and of the reference's
//     widening factor 1 + 2 gamma(3).  An axis whose reciprocal is not finite (or beyond 2^60) is left out of the grid test (near -inf, far +inf) and, as the
//     reference's slab test then passes only while o lies between the planes, replaced by that containment test with the same pad.
// The stack holds refs only (4 bytes: every pushed entry is due, t_max is constant): half the LDS column of k_trace_w4, spent on a longer root-side prefix of records.
// Plain scenes only (no instances, no alpha masks); everything else keeps k_trace_w4<true, ..>.
#pragma once
#include "trace_w4.h"

namespace rspt {

struct Quad4Node {      // 64 B, 64-byte aligned: the four boxes of a Wide4Node on an 8-bit grid (rspt_scene_create quantises, outward)
    float org[3];       // the frame's origin: plane = org + q * 2^(e - 127)
    uint32_t meta;      // e_x | e_y << 8 | e_z << 16 (biased exponents of the cell sizes) | empty-slot mask << 24
    uint32_t qlo[3];    // per axis: the four slots' lower planes, slot k in byte k
    uint32_t qhi_x;
    uint32_t qhi_y, qhi_z, pad0, pad1;
    uint32_t ref[4];    // as Wide4Node::ref (the axis bits are not used here)
};
static_assert(sizeof(Quad4Node) == 64, "Quad4Node is four 16-byte loads");
#ifndef RSPT_W4Q_LDS
#define RSPT_W4Q_LDS 16      // stack entries (4 B) per lane in LDS: 16 KB per workgroup
#endif
#ifndef RSPT_W4Q_TOP
#define RSPT_W4Q_TOP 216     // root-side records in LDS: 64 * 216 = 13.5 KB; with the stack columns 29.5 KB = five workgroups per CU, as k_trace_w4
#endif
#define RSPT_W4Q_SPILL (RSPT_W4_MAX_STACK - RSPT_W4Q_LDS)
#ifndef RSPT_W4Q_ORDERED
#define RSPT_W4Q_ORDERED 0   // 1: the four slots of a record in the reference's near-first order (A/B, tools/ab_build.sh AB_DEFS=-DRSPT_W4Q_ORDERED=1)
#endif

template <int OUT_MODE>
__global__ __launch_bounds__(RSPT_PW_BLOCK) void k_trace_w4q(SceneDev sc, const Quad4Node* __restrict__ recs, const uint2* __restrict__ big_leaves, uint32_t root_ref,
                                                             const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_ptr, uint32_t count_imm, uint32_t* cursor,
                                                             const rspt_ray* __restrict__ rays, uint32_t* __restrict__ out_occ, rspt_hit* __restrict__ out_hits,
                                                             uint32_t* __restrict__ spill, int refill_thresh, int leaf_thresh, uint32_t n_top, uint32_t chunk,
                                                             const float4* __restrict__ leaf_boxes) {
    constexpr int BLOCK = RSPT_PW_BLOCK;
    __shared__ uint32_t stack[RSPT_W4Q_LDS * BLOCK];
    __shared__ float4 top[4 * RSPT_W4Q_TOP];   // top[j * TOP + r] = j-th 16 bytes of record r
    uint32_t* my = stack + threadIdx.x;
    if (n_top > (uint32_t)RSPT_W4Q_TOP) n_top = (uint32_t)RSPT_W4Q_TOP;
    for (uint32_t i = threadIdx.x; i < 4u * n_top; i += BLOCK) {
        const uint32_t r = i >> 2, j = i & 3u;
        top[j * RSPT_W4Q_TOP + r] = reinterpret_cast<const float4*>(recs + r)[j];
    }
    __syncthreads();
    const size_t spill_stride = (size_t)gridDim.x * BLOCK;
    uint32_t* my_spill = spill + (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t n = count_ptr ? *count_ptr : count_imm;
    bool late_box = false;   // wave-uniform
    {   // a short queue is spread over all waves (trace_w4.h: the claim shrinks to the queue's share per wave)
        const bool adapt = !(chunk & 1u);   // (bit 0 of the launch parameter: RSPT_PW_ADAPT=0, the A/B switch; the claim itself is a multiple of 64)
        late_box = (chunk & 2u) != 0u;      // (bit 1: the leaf's exact box test after a triangle hit instead of before the triangles — below)
        chunk &= ~63u;
        const uint32_t waves = gridDim.x * (uint32_t)(BLOCK / 64);
        uint32_t per = ((n + waves - 1u) / waves + 63u) & ~63u;
        if (per < 64u) per = 64u;
        if (adapt && per < chunk) chunk = per;
    }
    const uint32_t lane = __lane_id();
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const float4 root0 = sc.nodes[0], root1 = sc.nodes[1];
    const float q_scene = fmaxf(fmaxf(fmaxf(fabsf(root0.x), fabsf(root0.y)), fmaxf(fabsf(root0.z), fabsf(root0.w))), fmaxf(fabsf(root1.x), fabsf(root1.y)));
    uint32_t chunk_lo = 0, chunk_hi = 0;  // wave-uniform
    bool exhausted = false;               // wave-uniform
    // per-lane ray state
    bool active = false;
    float ox = 0, oy = 0, oz = 0, ix = 0, iy = 0, iz = 0;
    RayShear rs{0, 0, 0, 0, 0, 0};
    float t_max = 0.0f, q_margin = 0.0f;
    uint32_t negbits = 0;  // bit a = dir_is_neg[a]; bits 4..6: axes the grid test leaves out
    uint32_t sp = 0, cur = RSPT_NONE, leaf = RSPT_NONE;
    uint32_t entry = 0, qpos = 0;

    auto finish = [&](bool occluded) {
        if (OUT_MODE == 0) out_occ[entry] = occluded ? 1u : 0u;
        else {
            rspt_hit h;
            h.prim = occluded ? 0u : RSPT_MISS; h.t = h.b0 = h.b1 = h.b2 = 0.0f;
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
                if (lane == 0) base = atomicAdd(cursor, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                chunk_lo = base < n ? base : n;
                chunk_hi = (base + chunk) < n ? (base + chunk) : n;
                if (chunk_lo == chunk_hi) exhausted = true;
            }
            if (!exhausted) {
                const uint32_t avail = chunk_hi - chunk_lo;
                const uint32_t rank = (uint32_t)__popcll(idle & lt_mask);
                if (!active && rank < avail) {
                    qpos = chunk_lo + rank;
                    entry = queue ? queue[qpos] : qpos;
                    const float4* rp = reinterpret_cast<const float4*>(rays + entry);
                    float4 r0 = rp[0], r1 = rp[1];
                    ox = r0.x; oy = r0.y; oz = r0.z;
                    f3 d{r0.w, r1.x, r1.y};
                    t_max = r1.z;
                    ix = 1.0f / d.x; iy = 1.0f / d.y; iz = 1.0f / d.z;
                    negbits = (ix < 0.0f ? 1u : 0u) | (iy < 0.0f ? 2u : 0u) | (iz < 0.0f ? 4u : 0u);
                    const float big = 0x1.0p60f;   // cell sizes stop at 2^60 (rspt_scene_create): cell * inv stays finite
                    negbits |= (fabsf(ix) < big ? 0u : 16u) | (fabsf(iy) < big ? 0u : 32u) | (fabsf(iz) < big ? 0u : 64u);
                    q_margin = 0x1.0p-19f * (q_scene + fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)));   // in position units; times |inv| of an axis = that axis's D
                    rs = ray_shear(d);
                    sp = 0; cur = RSPT_NONE; leaf = RSPT_NONE;
                    active = true;
                    // the root's own box (bvh.rs:480 on node 0), the reference's test
                    if (box_hit(root0, root1, f3{ox, oy, oz}, f3{ix, iy, iz}, negbits & 1u, negbits & 2u, negbits & 4u, t_max)) {
                        if (root_ref & RSPT_REF_LEAF) leaf = root_ref;
                        else cur = root_ref;
                    } else
                        finish(false);
                }
                const uint32_t want = (uint32_t)__popcll(idle);
                chunk_lo += want < avail ? want : avail;
            }
        }
        if (__ballot(active) == 0) {
            if (exhausted) break;
            continue;
        }

        // ---- node phase ----
#pragma unroll 1
        for (int step = 0; step < RSPT_W4_STEPS; step++)
        if (active && leaf == RSPT_NONE) {
            uint32_t ridx = cur;
            if (ridx == RSPT_NONE) {
                if (sp == 0) finish(false);   // every box the ray reaches has been looked at: nothing stops it
                else {
                    sp--;
                    uint32_t e = my[(sp < RSPT_W4Q_LDS ? sp : RSPT_W4Q_LDS - 1u) * BLOCK];
                    asm volatile("" : "+v"(e));  // pins the LDS read (never one load through a selected pointer: that becomes a flat load)
                    if (sp >= RSPT_W4Q_LDS) e = my_spill[(size_t)(sp - RSPT_W4Q_LDS) * spill_stride];
                    if (e & RSPT_REF_LEAF) leaf = e;
                    else ridx = e;
                }
            }
            if (ridx != RSPT_NONE) {
                float4 a0, a1, a2, rf;
                if (ridx < n_top) {
                    const float4* lp = top + ridx;
                    a0 = lp[0]; a1 = lp[RSPT_W4Q_TOP]; a2 = lp[2 * RSPT_W4Q_TOP]; rf = lp[3 * RSPT_W4Q_TOP];
                    asm volatile("" : "+v"(rf.w));
                } else {
                    const float4* pp = reinterpret_cast<const float4*>(recs + ridx);
                    a0 = pp[0]; a1 = pp[1]; a2 = pp[2]; rf = pp[3];
                }
                cur = RSPT_NONE;
                // the grid test (header comment): per axis near / far = q * (cell * inv) + (org - o) * inv -/+ D, the near plane being the upper one where the
                // direction is negative; axes left out contribute -inf / +inf
                const uint32_t meta = __float_as_uint(a0.w);
                const bool ux = !(negbits & 16u), uy = !(negbits & 32u), uz = !(negbits & 64u);
                const float cx = ux ? ix : 0.0f, cy = uy ? iy : 0.0f, cz = uz ? iz : 0.0f;
                const float bx = __uint_as_float((meta & 0xffu) << 23) * cx, by = __uint_as_float((meta & 0xff00u) << 15) * cy, bz = __uint_as_float((meta & 0xff0000u) << 7) * cz;
                const float ax = (a0.x - ox) * cx, ay = (a0.y - oy) * cy, az = (a0.z - oz) * cz;
                const float dx = q_margin * fabsf(cx), dy = q_margin * fabsf(cy), dz = q_margin * fabsf(cz);
                const float anx = ux ? ax - dx : -RSPT_INF, afx = ux ? ax + dx : RSPT_INF;
                const float any_ = uy ? ay - dy : -RSPT_INF, afy = uy ? ay + dy : RSPT_INF;
                const float anz = uz ? az - dz : -RSPT_INF, afz = uz ? az + dz : RSPT_INF;
                const uint32_t lox = __float_as_uint(a1.x), loy = __float_as_uint(a1.y), loz = __float_as_uint(a1.z);
                const uint32_t hix = __float_as_uint(a1.w), hiy = __float_as_uint(a2.x), hiz = __float_as_uint(a2.y);
                const uint32_t nqx = (negbits & 1u) ? hix : lox, fqx = (negbits & 1u) ? lox : hix;
                const uint32_t nqy = (negbits & 2u) ? hiy : loy, fqy = (negbits & 2u) ? loy : hiy;
                const uint32_t nqz = (negbits & 4u) ? hiz : loz, fqz = (negbits & 4u) ? loz : hiz;
                bool h0, h1, h2, h3;
#define RSPT_Q4_SLOT(K, H)                                                                                                                                \
                {                                                                                                                                     \
                    const float tn = fmaxf(fmaxf(__builtin_fmaf((float)((nqx >> (8 * K)) & 0xffu), bx, anx), __builtin_fmaf((float)((nqy >> (8 * K)) & 0xffu), by, any_)), \
                                           __builtin_fmaf((float)((nqz >> (8 * K)) & 0xffu), bz, anz));                                               \
                    const float tf = fminf(fminf(__builtin_fmaf((float)((fqx >> (8 * K)) & 0xffu), bx, afx), __builtin_fmaf((float)((fqy >> (8 * K)) & 0xffu), by, afy)), \
                                           __builtin_fmaf((float)((fqz >> (8 * K)) & 0xffu), bz, afz));                                               \
                    H = (tn <= tf) && (tn < t_max) && (tf > 0.0f) && !((meta >> (24 + K)) & 1u);                                                      \
                }
                RSPT_Q4_SLOT(0, h0) RSPT_Q4_SLOT(1, h1) RSPT_Q4_SLOT(2, h2) RSPT_Q4_SLOT(3, h3)
#undef RSPT_Q4_SLOT
                if (negbits & 0x70u) {   // (rare) an axis without a finite reciprocal: the reference's slab test passes only while o lies between the planes
                    auto between = [&](uint32_t lo_dw, uint32_t hi_dw, uint32_t ebits, float org_a, float o_a, uint32_t k) {
                        const float cell = __uint_as_float(ebits << 23);
                        const float lo = __builtin_fmaf((float)((lo_dw >> (8u * k)) & 0xffu), cell, org_a), hi = __builtin_fmaf((float)((hi_dw >> (8u * k)) & 0xffu), cell, org_a);
                        return o_a >= lo - q_margin && o_a <= hi + q_margin;
                    };
                    bool in[4] = {true, true, true, true};
#pragma unroll
                    for (uint32_t k = 0; k < 4u; k++) {
                        if (negbits & 16u) in[k] = in[k] && between(lox, hix, meta & 0xffu, a0.x, ox, k);
                        if (negbits & 32u) in[k] = in[k] && between(loy, hiy, (meta >> 8) & 0xffu, a0.y, oy, k);
                        if (negbits & 64u) in[k] = in[k] && between(loz, hiz, (meta >> 16) & 0xffu, a0.z, oz, k);
                    }
                    h0 = h0 && in[0]; h1 = h1 && in[1]; h2 = h2 && in[2]; h3 = h3 && in[3];
                }
                const uint32_t g0 = __float_as_uint(rf.x), g1 = __float_as_uint(rf.y), g2 = __float_as_uint(rf.z);
                const uint32_t f0 = g0 & ~RSPT_W4_AXIS_MASK, f1 = g1 & ~RSPT_W4_AXIS_MASK, f2 = g2 & ~RSPT_W4_AXIS_MASK, f3w = __float_as_uint(rf.w);
                uint32_t next = RSPT_NONE;
#if RSPT_W4Q_ORDERED
                {   // the reference's near-first order of the four (trace_w4.h: three sign bits): an occluder near the origin is found before the far side of the tree is walked
                    const bool sA = ((negbits >> ((g0 >> RSPT_W4_AXIS_SHIFT) & 3u)) & 1u) != 0, sB0 = ((negbits >> ((g1 >> RSPT_W4_AXIS_SHIFT) & 3u)) & 1u) != 0,
                               sB1 = ((negbits >> ((g2 >> RSPT_W4_AXIS_SHIFT) & 3u)) & 1u) != 0;
                    const uint32_t r0 = h0 ? f0 : RSPT_NONE, r1 = h1 ? f1 : RSPT_NONE, r2 = h2 ? f2 : RSPT_NONE, r3 = h3 ? f3w : RSPT_NONE;
                    const uint32_t g0n = sB0 ? r1 : r0, g0f = sB0 ? r0 : r1, g1n = sB1 ? r3 : r2, g1f = sB1 ? r2 : r3;
                    const uint32_t e0 = sA ? g1n : g0n, e1 = sA ? g1f : g0f, e2 = sA ? g0n : g1n, e3 = sA ? g0f : g1f;
                    auto put = [&](uint32_t ref) {
                        if (sp < RSPT_W4Q_LDS) {
                            my[sp * BLOCK] = ref;
                            asm volatile("");
                        } else
                            my_spill[(size_t)(sp - RSPT_W4Q_LDS) * spill_stride] = ref;
                        sp++;
                    };
                    next = e0 != RSPT_NONE ? e0 : (e1 != RSPT_NONE ? e1 : (e2 != RSPT_NONE ? e2 : e3));
                    const bool p1 = e0 != RSPT_NONE, p2 = p1 || e1 != RSPT_NONE, p3 = p2 || e2 != RSPT_NONE;
                    if (e3 != RSPT_NONE && p3) put(e3);
                    if (e2 != RSPT_NONE && p2) put(e2);
                    if (e1 != RSPT_NONE && p1) put(e1);
                }
#else
                // stored order: the first slot that passes is walked next, the others wait on the stack (occlusion does not depend on the order)
                auto take = [&](bool h, uint32_t ref) {
                    if (!h) return;
                    if (next == RSPT_NONE) { next = ref; return; }
                    if (sp < RSPT_W4Q_LDS) {
                        my[sp * BLOCK] = ref;
                        asm volatile("");  // keeps the LDS store and the global store apart
                    } else
                        my_spill[(size_t)(sp - RSPT_W4Q_LDS) * spill_stride] = ref;
                    sp++;
                };
                take(h0, f0); take(h1, f1); take(h2, f2); take(h3, f3w);
#endif
                if (next != RSPT_NONE) {
                    if (next & RSPT_REF_LEAF) leaf = next;
                    else cur = next;
                }
            }
        }

        // ---- leaf phase ----
        const uint64_t parked = __ballot(active && leaf != RSPT_NONE);
        if (parked) {
            const uint64_t running = __ballot(active && leaf == RSPT_NONE);
            if (__popcll(parked) >= leaf_thresh || running == 0) {
                if (active && leaf != RSPT_NONE) {
                    uint32_t offset = leaf & RSPT_W4_OFFSET_MASK, n_prims = ((leaf >> RSPT_W4_COUNT_SHIFT) & 15u) + 1u;
                    if (n_prims == 16u) {
                        const uint2 bl = big_leaves[offset];
                        offset = bl.x; n_prims = bl.y;
                    }
                    leaf = RSPT_NONE;
                    const f3 o{ox, oy, oz};
                    // the reference's own test of this leaf's box (bvh.rs:480): what the grid let through too generously stops here.  The reference's answer is "some
                    // triangle passes the watertight test AND its leaf's box passes the exact test" (the ancestors' boxes pass whenever the leaf's does, and t_max never
                    // changes): the two tests commute.  late_box runs the box test only for a leaf in which a triangle was hit — once per occluded ray instead of once per
                    // leaf visit (4.6 per ray on C2), one dependent 32-byte fetch less in front of the triangles; a hit in a leaf the reference never opens is dropped and
                    // the walk goes on.  Flags byte-identical either way (tools/trace_bench.py --check, the GPU suite under RSPT_ANY_Q_LATE=0 / 1).
                    if (!late_box) {
                        const float4 q0 = leaf_boxes[2 * (size_t)offset], q1 = leaf_boxes[2 * (size_t)offset + 1];
                        if (!box_hit(q0, q1, o, f3{ix, iy, iz}, negbits & 1u, negbits & 2u, negbits & 4u, t_max)) n_prims = 0u;
                    }
                    bool hit = false;
                    for (uint32_t i = 0; i < n_prims && !hit; i++) {
                        const uint32_t pi = offset + i;
                        const float4 a = sc.tris[3 * (size_t)pi], b = sc.tris[3 * (size_t)pi + 1], c = sc.tris[3 * (size_t)pi + 2];
                        float t, b0, b1, b2;
                        hit = tri_test(f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c.x}, o, rs, t_max, &t, &b0, &b1, &b2);
                    }
                    if (hit && late_box) {
                        const float4 q0 = leaf_boxes[2 * (size_t)offset], q1 = leaf_boxes[2 * (size_t)offset + 1];
                        hit = box_hit(q0, q1, o, f3{ix, iy, iz}, negbits & 1u, negbits & 2u, negbits & 4u, t_max);
                    }
                    if (hit) finish(true);
                }
            }
        }
    }
}

}  // namespace rspt

