// traverse_w4 — one lane, one ray, over the four-box records of trace_w4.h: the traversal of the per-lane kernels (k_tile_serial: the pixel samplers,
// one lane per tile; k_lane_dl: directlighting's specular trees, one lane per camera sample).
//
// Those kernels are chains of dependent loads — nothing in a lane can start before the previous fetch is back, and with a handful of lanes per
// wave there is nothing else to run meanwhile — so what counts is the NUMBER of dependent round trips per ray.  The reference-order loop
// (kernels.h traverse<>) makes one per LinearBVHNode it visits (38 per ray on the C3 stand-in, 147 on C2); a four-box record covers two levels
// of the tree per fetch and its entries carry their entry distances, so a popped subtree that a closer hit has culled costs no fetch at all:
// about a fifth of the round trips.  Same visiting order, same boxes, same arithmetic as k_trace_w4's node step (the argument of trace_w4.h's
// header: a grandchild's box implies its parent's; `t_min < t_max` is re-checked on pop), triangles tested the moment a lane reaches a leaf —
// so (prim, t, b0, b1, b2) and the final t_max are the reference's, bit for bit.  Scenes with object instances, or with alpha masks outside the in-line form (dev_scene.h AlphaMask), keep traverse<>.
#pragma once
#include "trace_w4.h"

namespace rspt {

#define RSPT_SERIAL_LDS 16   // stack entries (8 B) per lane in the block's 8 KB of LDS (the 32 four-byte levels traverse<> uses); the rest in scratch

template <bool ANY, bool ALPHA = false /* alpha masks in the in-line form (kernels.h alpha_simple) */>
RDEVN TraceResult traverse_w4(const SceneDev& sc, const TexTables& tt, f3 o, f3 d, float t_max, uint32_t* lds_stack /* this lane's column of the block's 32 x 64 words */) {
    TraceResult res;
    res.prim = RSPT_MISS; res.t = 0.0f; res.b0 = res.b1 = res.b2 = 0.0f; res.nodes = 0; res.tris = 0; res.inst = 0; res.t_end = t_max;
    if (sc.n_nodes == 0) return res;
    const Wide4Node* __restrict__ recs = reinterpret_cast<const Wide4Node*>(sc.w4);
    const float ox = o.x, oy = o.y, oz = o.z;
    const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    uint32_t negbits = (ix < 0.0f ? 1u : 0u) | (iy < 0.0f ? 2u : 0u) | (iz < 0.0f ? 4u : 0u);
    if (!(fabsf(ix) < RSPT_INF && fabsf(iy) < RSPT_INF && fabsf(iz) < RSPT_INF)) negbits |= 8u;   // zero / non-finite components: the reference's literal compare chain
    const RayShear rs = ray_shear(d);
    if (!box_hit(sc.nodes[0], sc.nodes[1], o, f3{ix, iy, iz}, negbits & 1u, negbits & 2u, negbits & 4u, t_max)) return res;   // node 0's own box (bvh.rs:424)
    uint2* col = reinterpret_cast<uint2*>(lds_stack - threadIdx.x) + threadIdx.x;   // the same LDS as 16 levels of (ref, t_min)
    uint2 spill[RSPT_W4_MAX_STACK - RSPT_SERIAL_LDS];
    uint32_t sp = 0, cur = RSPT_NONE, leaf = RSPT_NONE;
    if (sc.w4_root & RSPT_REF_LEAF) leaf = sc.w4_root;
    else cur = sc.w4_root;
    for (;;) {
        if (leaf != RSPT_NONE) {
            uint32_t offset = leaf & RSPT_W4_OFFSET_MASK, n_prims = ((leaf >> RSPT_W4_COUNT_SHIFT) & 15u) + 1u;
            if (n_prims == 16u) {
                const uint2 bl = sc.w4_big[offset];
                offset = bl.x; n_prims = bl.y;
            }
            leaf = RSPT_NONE;
            for (uint32_t i = 0; i < n_prims; i++) {
                const uint32_t pi = offset + i;
                const float4 a = sc.tris[3 * (size_t)pi], b = sc.tris[3 * (size_t)pi + 1], c = sc.tris[3 * (size_t)pi + 2];
                res.tris++;
                float t, b0, b1, b2;
                if (tri_test(f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c.x}, o, rs, t_max, &t, &b0, &b1, &b2)) {
                    if (ALPHA && (__float_as_uint(c.w) & MF_ALPHA) &&
                        !alpha_simple<ANY>(sc, tt, pi, __float_as_uint(c.w), f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c.x}, b0, b1, b2)) continue;
                    if (ANY) { res.prim = 0; return res; }
                    t_max = t;   // primitive.rs:155
                    res.prim = pi; res.t = t; res.b0 = b0; res.b1 = b1; res.b2 = b2;
                }
            }
        }
        uint32_t ridx = cur;
        if (ridx == RSPT_NONE) {   // pop: entries a closer hit has culled meanwhile are dropped without a fetch
            bool got = false;
            while (sp > 0) {
                sp--;
                const uint2 e = sp < RSPT_SERIAL_LDS ? col[sp * 64u] : spill[sp - RSPT_SERIAL_LDS];
                if (__uint_as_float(e.y) < t_max) {   // the reference's box test at this later moment (bvh.rs:424)
                    if (e.x & RSPT_REF_LEAF) leaf = e.x;
                    else ridx = e.x;
                    got = true;
                    break;
                }
            }
            if (!got) break;
            if (leaf != RSPT_NONE) continue;
        }
        const float4* pp = reinterpret_cast<const float4*>(recs + ridx);
        const float4 a0 = pp[0], a1 = pp[1], a2 = pp[2], a3 = pp[3], a4 = pp[4], a5 = pp[5], rf = pp[6];
        res.nodes++;
        cur = RSPT_NONE;
        bool h0, h1, h2, h3;
        float m0, m1, m2, m3;
        if (!(negbits & 8u)) {
            box_pair_hit_m(a0, a1, a2, ox, oy, oz, ix, iy, iz, t_max, &h0, &h1, &m0, &m1);
            box_pair_hit_m(a3, a4, a5, ox, oy, oz, ix, iy, iz, t_max, &h2, &h3, &m2, &m3);
        } else {
            const f3 inv{ix, iy, iz};
            const bool n0 = negbits & 1u, n1 = negbits & 2u, n2 = negbits & 4u;
            h0 = box_hit6_m(a0.x, a1.x, a2.x, a0.z, a1.z, a2.z, o, inv, n0, n1, n2, t_max, &m0);
            h1 = box_hit6_m(a0.y, a1.y, a2.y, a0.w, a1.w, a2.w, o, inv, n0, n1, n2, t_max, &m1);
            h2 = box_hit6_m(a3.x, a4.x, a5.x, a3.z, a4.z, a5.z, o, inv, n0, n1, n2, t_max, &m2);
            h3 = box_hit6_m(a3.y, a4.y, a5.y, a3.w, a4.w, a5.w, o, inv, n0, n1, n2, t_max, &m3);
        }
        // the visiting order of the four slots from the three axis bits, as in k_trace_w4
        const uint32_t f0 = __float_as_uint(rf.x), f1 = __float_as_uint(rf.y), f2 = __float_as_uint(rf.z), f3w = __float_as_uint(rf.w);
        const uint32_t r0 = h0 ? (f0 & ~RSPT_W4_AXIS_MASK) : RSPT_NONE, r1 = h1 ? (f1 & ~RSPT_W4_AXIS_MASK) : RSPT_NONE;
        const uint32_t r2 = h2 ? (f2 & ~RSPT_W4_AXIS_MASK) : RSPT_NONE, r3 = h3 ? f3w : RSPT_NONE;
        const bool sA = ((negbits >> ((f0 >> RSPT_W4_AXIS_SHIFT) & 3u)) & 1u) != 0;
        const bool sB0 = ((negbits >> ((f1 >> RSPT_W4_AXIS_SHIFT) & 3u)) & 1u) != 0;
        const bool sB1 = ((negbits >> ((f2 >> RSPT_W4_AXIS_SHIFT) & 3u)) & 1u) != 0;
        const uint32_t g0n = sB0 ? r1 : r0, g0f = sB0 ? r0 : r1, g1n = sB1 ? r3 : r2, g1f = sB1 ? r2 : r3;
        const float mg0n = sB0 ? m1 : m0, mg0f = sB0 ? m0 : m1, mg1n = sB1 ? m3 : m2, mg1f = sB1 ? m2 : m3;
        const uint32_t e0 = sA ? g1n : g0n, e1 = sA ? g1f : g0f, e2 = sA ? g0n : g1n, e3 = sA ? g0f : g1f;
        const float me1 = sA ? mg1f : mg0f, me2 = sA ? mg0n : mg1n, me3 = sA ? mg0f : mg1f;
        const bool v0 = e0 != RSPT_NONE, v1 = e1 != RSPT_NONE, v2 = e2 != RSPT_NONE, v3 = e3 != RSPT_NONE;
        const bool p1 = v0, p2 = v0 || v1, p3 = p2 || v2;
        const uint32_t next = v0 ? e0 : (v1 ? e1 : (v2 ? e2 : e3));
        auto push = [&](uint32_t ref, float m) {
            const uint2 e = make_uint2(ref, __float_as_uint(m));
            if (sp < RSPT_SERIAL_LDS) col[sp * 64u] = e;
            else spill[sp - RSPT_SERIAL_LDS] = e;
            sp++;
        };
        if (v3 && p3) push(e3, me3);
        if (v2 && p2) push(e2, me2);
        if (v1 && p1) push(e1, me1);
        if (next != RSPT_NONE) {
            if (next & RSPT_REF_LEAF) leaf = next;
            else cur = next;
        }
    }
    res.t_end = t_max;
    return res;
}

// the traversal a per-lane kernel runs for one ray: the four-box records where the scene has them in the plain form
// ANIM (round 6: moving instances under the pixel samplers): the reference-order loop with primitive_to_world interpolated at the ray's time (kernels.h traverse<.., ANIM>)
template <bool ANY, bool INST, bool ALPHA, bool ANIM = false>
RDEV TraceResult serial_trace(const SceneDev& sc, const TexTables& tt, f3 o, f3 d, float t_max, uint32_t* lds, float time = 0.0f) {
    if (!INST && sc.w4) return traverse_w4<ANY, ALPHA>(sc, tt, o, d, t_max, lds);   // (with alpha masks: rspt_scene_create sets w4 only when they have the in-line form)
    return traverse<ANY, INST, ALPHA, 64, ANIM>(sc, tt, o, d, t_max, lds, time);
}

}  // namespace rspt
