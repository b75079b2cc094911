// k_trace_w4 — the production traversal kernel (K2/K3): persistent waves, dynamic ray fetch,
// four-boxes-per-record BVH nodes, entry distances on the stack, deferred leaf phase.
//
// Like k_trace_pw (trace_wide.h) it produces, for every ray, exactly the triangle-test sequence
// of BVHAccel::intersect / intersect_p (bvh.rs:401-514) — near child first by dir_is_neg[axis],
// the shrinking t_max applied to every later box and triangle test — so (prim, t, b0, b1, b2) are
// bit-identical to the reference, ties included.  Two things are done differently from k_trace_pw:
//
//   * One 128-byte record per interior node A at even depth holds the boxes of A's grandchildren
//     (or of a child that is a leaf): slots 0,1 = children of A's first child, slots 2,3 = children
//     of A's second child.  The reference visits them in the order fixed by three sign bits
//     (dir_is_neg[A.axis] picks the group, dir_is_neg[child.axis] the order inside a group).  The
//     skipped intermediate box test is redundant: a grandchild's box lies inside its parent's box
//     and Bounds3f::intersect_p (geometry.rs:2211-2269) is monotone in the box (same products,
//     monotone rounding), so a ray that reaches a grandchild's box also passes the parent's.
//     Halving the number of dependent fetch + pop/push rounds is what pays: the kernel is bound by
//     the L1 request rate of its per-lane record loads (seven per step) and VALU issue, not by
//     bytes.
//   * intersect_p depends on ray.t_max only through its last comparison `t_min < ray.t_max`.  The
//     stack therefore keeps (ref, t_min) and a pop re-checks `t_min < t_max` — exactly the outcome
//     of the reference's box test at the later moment (with the smaller t_max) — without fetching
//     the node again.  Entries whose box failed the t_max-independent part are never pushed.
//
// Leaf refs carry (first primitive, count) so the leaf phase needs no LinearBVHNode fetch.
#pragma once
#include "trace_wide.h"

namespace rspt {

struct Wide4Node {      // 128 B, 128-byte aligned
    float4 b[6];        // b[0..2]: x, y, z slabs of slots 0,1 as (s0.min, s1.min, s0.max, s1.max); b[3..5]: slots 2,3
    uint32_t ref[4];    // bit 31 = leaf (RSPT_W4_* fields), else Wide4Node index; empty slots have a NaN box (ref never read).
                        // Bits 25..26 of ref[0], ref[1], ref[2] carry A.axis, the first child's axis and the second child's
                        // axis: the kernel is bound by the number of scattered load instructions per step (one more dword
                        // load per step cost 10 % of the kernel's time), so the axes ride in the load that fetches the refs.
    uint32_t pad[4];
};
// leaf ref: bit 31 | (count - 1) << 27 | first primitive; count field 15 = look the pair up in big_leaves[low bits]
#define RSPT_W4_COUNT_SHIFT 27
#define RSPT_W4_AXIS_SHIFT 25
#define RSPT_W4_AXIS_MASK 0x06000000u
#define RSPT_W4_OFFSET_MASK 0x01ffffffu
#define RSPT_W4_ENTER 0xfffffffeu   // k_trace_w4<.., ANIM>: "parked in front of an instance" in the lane's leaf word (no leaf reference carries the axis bits)
#ifndef RSPT_W4_LDS
#define RSPT_W4_LDS 12       // stack entries per lane (8 B each) kept in LDS: 24 KB per workgroup
#endif
#ifndef RSPT_W4_TOP
#define RSPT_W4_TOP 56       // records nearest the root (a breadth-first prefix, numbered first by rspt_scene_create) that every
                             // workgroup keeps in LDS: their fetches leave the L1 request path that bounds the kernel.
                             // 2048 * RSPT_W4_LDS + 112 * RSPT_W4_TOP must stay under ~31 KB or only four workgroups fit a CU
                             // (measured C2 / C3 Msamples/s: (16, 0) 366 / 1139; (12, 56) 379 / 1172; (11, 72) 382 / 1161;
                             //  (10, 85) 381 / 1153; (13, 36) 378 / 1170; four workgroups (12, 72) 375 / 1137; three (16, 85) 340 / 1065;
                             //  six waves per SIMD forced with amdgpu_waves_per_eu (80 VGPRs, 13 spilled) and (10, 56): 366 / 1141)
#endif
static_assert(2048 * RSPT_W4_LDS + 112 * RSPT_W4_TOP <= 64 * 1024, "k_trace_w4 LDS budget (64 KB per workgroup)");
#define RSPT_W4_TOP_MAX 512  // the breadth-first prefix rspt_scene_create numbers first: the largest TOPCAP any instantiation keeps in LDS
#ifndef RSPT_W4_SHAPE_DEFAULT
#define RSPT_W4_SHAPE_DEFAULT 0
#endif
#ifndef RSPT_W4_POP_TRIES
#define RSPT_W4_POP_TRIES 1  // stack entries a lane may discard (t_min >= t_max) in one iteration before it gives up the slot
#endif
#ifndef RSPT_W4_STEPS
#define RSPT_W4_STEPS 2      // node steps per outer iteration (refill / leaf-phase checks in between); measured on C2 / C3:
                             // (tries, steps) = (1,1) 334.5 / 1059, (3,1) 326.9 / 1041, (1,2) 338.5 / 1073, (3,2) 333.7 / 1049 Msamples/s
#endif
#define RSPT_W4_MAX_STACK 96 // deepest stack a ray can need: rspt_scene_create admits at most 64 LinearBVHNode levels (the reference's own
                             // fixed stack, bvh.rs:420) = 32 record levels, and a step leaves at most three entries behind per level
#define RSPT_W4_SPILL (RSPT_W4_MAX_STACK - RSPT_W4_LDS)  // further entries per lane in a global spill buffer (spill_rows <= this): with all of
                             // them no ray can overflow, so no ray is re-traced from scratch (round 1 stopped at 48 rows and paid 132 ms
                             // per C2 frame for a handful of deep rays redone one per lane by k_trace_fixup); fewer rows
                             // (RSPT_W4_SPILL_ROWS) bring k_trace_fixup back

// box_pair_hit (trace_wide.h) that also returns the entry distances
RDEV void box_pair_hit_m(float4 q0, float4 q1, float4 q2, float ox, float oy, float oz, float ix, float iy, float iz, float ray_tmax,
                         bool* h0, bool* h1, float* m0o, float* m1o) {
    const float widen = 1.0f + 2.0f * gamma_n(3);
    v2f lx = (v2f{q0.x, q0.y} - ox) * ix, hx = (v2f{q0.z, q0.w} - ox) * ix;
    v2f ly = (v2f{q1.x, q1.y} - oy) * iy, hy = (v2f{q1.z, q1.w} - oy) * iy;
    v2f lz = (v2f{q2.x, q2.y} - oz) * iz, hz = (v2f{q2.z, q2.w} - oz) * iz;
    // x -> x * widen is monotone, so min3 of the widened fars == widened min3 of the fars (one multiply instead of three)
    float m0 = fmaxf(fmaxf(fminf(lx.x, hx.x), fminf(ly.x, hy.x)), fminf(lz.x, hz.x));
    float m1 = fmaxf(fmaxf(fminf(lx.y, hx.y), fminf(ly.y, hy.y)), fminf(lz.y, hz.y));
    v2f M = v2f{fminf(fminf(fmaxf(lx.x, hx.x), fmaxf(ly.x, hy.x)), fmaxf(lz.x, hz.x)),
                fminf(fminf(fmaxf(lx.y, hx.y), fmaxf(ly.y, hy.y)), fmaxf(lz.y, hz.y))} * widen;
    *h0 = (m0 <= M.x) && (m0 < ray_tmax) && (M.x > 0.0f);
    *h1 = (m1 <= M.y) && (m1 < ray_tmax) && (M.y > 0.0f);
    *m0o = m0; *m1o = m1;
}
// literal reference chain (zero / non-finite direction components), returning the final t_min
RDEVN bool box_hit6_m(float lx, float ly, float lz, float hx, float hy, float hz, f3 o, f3 inv, bool ng0, bool ng1, bool ng2, float ray_tmax, float* mo) {
    const float widen = 1.0f + 2.0f * gamma_n(3);
    float t_min = ((ng0 ? hx : lx) - o.x) * inv.x;
    float t_max = ((ng0 ? lx : hx) - o.x) * inv.x;
    float ty_min = ((ng1 ? hy : ly) - o.y) * inv.y;
    float ty_max = ((ng1 ? ly : hy) - o.y) * inv.y;
    t_max *= widen;
    ty_max *= widen;
    *mo = 0.0f;
    if (t_min > ty_max || ty_min > t_max) return false;
    if (ty_min > t_min) t_min = ty_min;
    if (ty_max < t_max) t_max = ty_max;
    float tz_min = ((ng2 ? hz : lz) - o.z) * inv.z;
    float tz_max = ((ng2 ? lz : hz) - o.z) * inv.z;
    tz_max *= widen;
    if (t_min > tz_max || tz_min > t_max) return false;
    if (tz_min > t_min) t_min = tz_min;
    if (tz_max < t_max) t_max = tz_max;
    *mo = t_min;
    return (t_min < ray_tmax) && (t_max > 0.0f);
}

// INST (scenes with object instances): a leaf primitive may be a TransformedPrimitive (primitive.rs:216-265).  The lane
// then pushes the reference to the rest of that leaf, switches to the instance's object-space ray (Transform::transform_ray with
// m_inv), runs the object's own records on top of the same stack, and on coming back down to that stack level reloads the world
// ray from its queue record.  t_max is carried over as the reference does (r.t_max.set(ray.t_max), quirks Q10 / Q11 in kernels.h
// traverse<>).  Without INST the code is the one measured in DESIGN.md (the flag is a template parameter, not a branch).
// ALPHA (scenes with alpha-masked meshes): a candidate that passed the watertight test on such a mesh is checked by alpha_pass
// (kernels.h) before it counts — a call into the texture code, which is why this too is a template flag.
// ANIM (with INST): some instances move (AnimatedTransform primitive_to_world, primitive.rs:198-222): entering such an instance interpolates its Transform at the
// ray's time (dev_scene.h inst_at — two key decompositions blended, a 4x4 inverse: ~100 live values for a moment), so it is its own instantiation and every
// other instanced scene keeps the register budget it was measured with.
// BLOCK / TOPCAP (round 5): threads per workgroup and the number of root-side records a workgroup keeps in LDS.  The measured default is five 256-thread
// workgroups per CU with 56 records each (30 KB); ONE 1024-thread workgroup per CU has room for 512 (96 KB of stack columns + 56 KB of records = the CU's
// 160 KB, declared as dynamic LDS): a ninth of the record fetches of an incoherent ray come from the first 56 records, about a quarter from the first 512 —
// fetches that leave the L1 request path the kernel is bound by.
#ifdef RSPT_W4_WAVES   // A/B (tools/ab_build.sh AB_DEFS=-DRSPT_W4_WAVES=n): every instantiation built for n waves per SIMD (what does not fit the budget is spilled)
#define RSPT_W4_ATTR __attribute__((amdgpu_waves_per_eu(RSPT_W4_WAVES, RSPT_W4_WAVES)))
#else
// the closest-hit kernel over moving instances is built for FOUR waves per SIMD: 147 -> 128 VGPRs + 64 B of scratch around the entry phase; all-moving C5 stand-in 183.2 -> 199.4
// Msamples/s (two alternating rounds, profiles/r06_c5_anim_waves_ab.txt).  (1, 8) is the backend's own default: every other instantiation keeps the budget it was measured with.
#define RSPT_W4_ATTR __attribute__((amdgpu_waves_per_eu((ANIM && !ANY) ? 4 : 1, (ANIM && !ANY) ? 4 : 8)))
#endif
template <bool ANY, int OUT_MODE, bool INST, int ALPHA /* 0: no masks, 1: alpha_pass (any texture graph, a call), 2: alpha_simple (in line) */, bool ANIM = false,
          int BLOCK = RSPT_PW_BLOCK, int TOPCAP = RSPT_W4_TOP>
__global__ __launch_bounds__(BLOCK) RSPT_W4_ATTR void k_trace_w4(SceneDev sc, TexTables tt, const Wide4Node* __restrict__ recs, const uint2* __restrict__ big_leaves, uint32_t root_ref,
                                                           const uint32_t* __restrict__ queue, const uint32_t* __restrict__ count_ptr, uint32_t count_imm, uint32_t* cursor,
                                                           const rspt_ray* __restrict__ rays_a, const rspt_ray* __restrict__ rays_b,
                                                           float4* __restrict__ out_a, float4* __restrict__ out_b, uint32_t* __restrict__ out_occ,
                                                           rspt_hit* __restrict__ out_hits, uint32_t* n_overflow, uint32_t* __restrict__ overflow_list,
                                                           uint2* __restrict__ spill, uint32_t spill_rows, int refill_thresh, int leaf_thresh, uint32_t n_top,
                                                           uint32_t* __restrict__ out_inst, uint32_t* xcd_cursors, uint32_t chunk /* rays a wave claims per global atomic (a multiple of 64) */) {
    uint2* stack;
    float4* top;   // top[j * TOPCAP + r] = j-th 16 bytes of record r (neighbouring records in neighbouring banks)
    if constexpr (8 * RSPT_W4_LDS * BLOCK + 112 * TOPCAP <= 64 * 1024) {   // the static form (the kernels measured since round 2 keep their code)
        __shared__ uint2 stack_s[RSPT_W4_LDS * BLOCK];
        __shared__ float4 top_s[7 * (TOPCAP > 0 ? TOPCAP : 1)];
        stack = stack_s; top = top_s;
    } else {
        extern __shared__ uint4 w4_dyn_lds[];
        stack = reinterpret_cast<uint2*>(w4_dyn_lds);
        top = reinterpret_cast<float4*>(stack + RSPT_W4_LDS * BLOCK);
    }
    uint2* my = stack + threadIdx.x;
    // ANIM: entering a moving instance costs ~1 400 instructions (inst_inverse_at), ten node steps' worth, and a leaf phase runs with a fifth of the wave: lanes that
    // reach an instance are parked (leaf = RSPT_W4_ENTER) until enter_thresh of them wait — bits 8.. of the leaf_thresh parameter — or nothing else can run
    int enter_thresh = 1;
    if constexpr (ANIM) { enter_thresh = leaf_thresh >> 8; leaf_thresh &= 0xff; if (enter_thresh < 1) enter_thresh = 1; }
    if (n_top > (uint32_t)TOPCAP) n_top = (uint32_t)TOPCAP;   // (the scene numbers a longer breadth-first prefix first than the small form keeps)
    if constexpr (TOPCAP > 0) {
        for (uint32_t i = threadIdx.x; i < 7u * n_top; i += BLOCK) {
            const uint32_t r = i / 7u, j = i - 7u * r;
            top[j * TOPCAP + r] = reinterpret_cast<const float4*>(recs + r)[j];
        }
        __syncthreads();
    }
    // rows RSPT_W4_LDS .. RSPT_W4_LDS + RSPT_W4_SPILL - 1 of a lane's stack live in global memory (row-major over all threads of the grid)
    const size_t spill_stride = (size_t)gridDim.x * BLOCK;
    uint2* my_spill = spill + (size_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t n = count_ptr ? *count_ptr : count_imm;
    // A short queue (the late iterations of a batch, a shard of a frame): with 256 rays per claim a few waves hold all of it and run their rays 64 at a time, one
    // round after the other, while the rest of the chip idles — the launch then lasts four dependent walks instead of one.  The claim shrinks to the queue's share per
    // wave (wave-uniform; which wave traces which ray changes, no ray's result does).  Queues of more than chunk x waves entries keep the launch parameter.
    {
        const bool adapt = !(chunk & 1u);   // (bit 0 of the launch parameter: RSPT_PW_ADAPT=0, the A/B switch; the claim itself is a multiple of 64)
        chunk &= ~63u;
        const uint32_t waves = gridDim.x * (uint32_t)(BLOCK / 64);
        uint32_t per = ((n + waves - 1u) / waves + 63u) & ~63u;
        if (per < 64u) per = 64u;
        if (adapt && per < chunk) chunk = per;
    }
    if (sc.n_nodes == 0) {  // empty scene: every ray misses
        for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += gridDim.x * BLOCK) {
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
    const float4 root0 = sc.nodes[0], root1 = sc.nodes[1];
    uint32_t chunk_lo = 0, chunk_hi = 0;  // wave-uniform
    bool exhausted = false;               // wave-uniform
    // XCD-affine dealing (xcd_cursors != nullptr): the queue is cut into eight contiguous ranges, one per XCD; a wave draws its chunks from the
    // range of the XCD it runs on (HW_REG_XCC_ID — placement is read, not assumed) and, once that is empty, from the others' in turn.  Neighbouring
    // queue entries are neighbouring path slots = neighbouring pixels, whose rays walk the same part of the tree: with one cursor the eight XCDs
    // drain the queue interleaved in 256-ray chunks and every XCD's 4 MB L2 sees the whole tree; with eight, each L2 sees its own region.
    // Which rays a wave traces changes, nothing about any ray's result does.
    uint32_t xcc = 0, victim = 0, per_xcd = 0;   // wave-uniform
    if (xcd_cursors) {
        uint32_t id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc = id & 7u;
        per_xcd = (uint32_t)((((uint64_t)n + 8ull * chunk - 1ull) / (8ull * chunk)) * chunk);
    }
    // per-lane ray state
    bool active = false;
    float ox = 0, oy = 0, oz = 0, ix = 0, iy = 0, iz = 0;
    RayShear rs{0, 0, 0, 0, 0, 0};
    float t_max = 0.0f;
    uint32_t negbits = 0;  // bit a = dir_is_neg[a]; bit 3 = zero / non-finite direction component
    uint32_t sp = 0, cur = RSPT_NONE, leaf = RSPT_NONE;
    uint32_t best = RSPT_MISS, entry = 0, qpos = 0;
    float bt = 0.0f, bb0 = 0.0f, bb1 = 0.0f, bb2 = 0.0f;
    // INST: instance being traversed, stack level at which its traversal ends, 1 + instance of the best hit, the world ray's t_max at
    // entry, BVHAccel::intersect's `hit` flag of the top-level aggregate, "the object reported a hit"
    uint32_t inst = RSPT_NONE, sp_base = 0, best_inst = 0;
    float w_tmax = 0.0f;
    bool hitflag = false, inst_hit = false, inst_ident = false;   // inst_ident: Transform::is_identity of the Transform the instance was entered with

    auto finish = [&]() {
        uint32_t slot = entry & ~RSPT_Q_MIS;
        if (INST && !ANY && !hitflag && best != RSPT_RETRACE) { best = RSPT_MISS; best_inst = 0; bt = bb0 = bb1 = bb2 = 0.0f; }  // `hit`, not "isect was written" (Q10)
        if (INST && !ANY && OUT_MODE == 0 && !(entry & RSPT_Q_MIS) && out_inst) out_inst[slot] = best_inst;
        if (OUT_MODE == 0) {
            if (ANY) out_occ[slot] = best == RSPT_RETRACE ? 2u : (best != RSPT_MISS ? 1u : 0u);
            else if (INST) {   // (two stores, not one store through a selected pointer: with the instance state live the select sent the kernel's pointer arguments — and with them the per-lane instance state — to scratch)
                const float4 v = make_float4(__uint_as_float(best), best == RSPT_MISS ? t_max : bb0, bb1, bb2);  // a miss: .y = the ray's final t_max (Q10)
                if (entry & RSPT_Q_MIS) { out_b[slot] = v; asm volatile(""); }   // (the empty asm keeps the compiler from merging the two stores back into one)
                else out_a[slot] = v;
            } else ((entry & RSPT_Q_MIS) ? out_b : out_a)[slot] = make_float4(__uint_as_float(best), bb0, bb1, bb2);
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
                if (xcd_cursors) {
                    for (; victim < 8u; victim++) {
                        const uint32_t x = (xcc + victim) & 7u;
                        const uint64_t lo64 = (uint64_t)x * per_xcd, hi64 = lo64 + per_xcd;
                        const uint32_t lo = lo64 < n ? (uint32_t)lo64 : n, hi = hi64 < n ? (uint32_t)hi64 : n;
                        if (lo == hi) continue;
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(xcd_cursors + x, chunk);
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base < hi - lo) {
                            chunk_lo = lo + base;
                            chunk_hi = (hi - lo - base) > chunk ? chunk_lo + chunk : hi;
                            break;
                        }
                    }
                    if (victim == 8u) exhausted = true;
                } else {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(cursor, chunk);
                    base = __builtin_amdgcn_readfirstlane(base);
                    chunk_lo = base < n ? base : n;
                    chunk_hi = (base + chunk) < n ? (base + chunk) : n;
                    if (chunk_lo == chunk_hi) exhausted = true;
                }
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
                    sp = 0; cur = RSPT_NONE; leaf = RSPT_NONE;
                    if (INST) { inst = RSPT_NONE; sp_base = 0; best_inst = 0; hitflag = false; inst_hit = false; }
                    active = true;
                    // the root's own box (bvh.rs:424 on node 0)
                    if (box_hit(root0, root1, f3{ox, oy, oz}, f3{ix, iy, iz}, negbits & 1u, negbits & 2u, negbits & 4u, t_max)) {
                        if (root_ref & RSPT_REF_LEAF) leaf = root_ref;
                        else cur = root_ref;
                    } else
                        finish();
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
#pragma unroll 1
        for (int step = 0; step < RSPT_W4_STEPS; step++)
        if (active && leaf == RSPT_NONE) {
            uint32_t ridx = cur;
            if (ridx == RSPT_NONE) {
                // pop; entries that a closer hit has meanwhile culled are skipped at once (a few per iteration)
#pragma unroll 1
                for (int tries = 0; tries < RSPT_W4_POP_TRIES; tries++) {
                    if (sp == sp_base) {
                        if (INST && inst != RSPT_NONE) {  // the object's traversal is over: back to world space (primitive.rs:224-253)
                            if (inst_hit) { if (sc.inst_fixed || !inst_ident) hitflag = true; }
                            else t_max = w_tmax;
                            const float4* rp = reinterpret_cast<const float4*>(((entry & RSPT_Q_MIS) ? rays_b : rays_a) + (entry & ~RSPT_Q_MIS));
                            const float4 r0 = rp[0], r1 = rp[1];
                            ox = r0.x; oy = r0.y; oz = r0.z;
                            const f3 d{r0.w, r1.x, r1.y};
                            ix = 1.0f / d.x; iy = 1.0f / d.y; iz = 1.0f / d.z;
                            negbits = (ix < 0.0f ? 1u : 0u) | (iy < 0.0f ? 2u : 0u) | (iz < 0.0f ? 4u : 0u);
                            if (!(fabsf(ix) < RSPT_INF && fabsf(iy) < RSPT_INF && fabsf(iz) < RSPT_INF)) negbits |= 8u;
                            rs = ray_shear(d);
                            inst = RSPT_NONE; sp_base = 0;
                            break;  // the next pop finds the rest of the instance's leaf (if any), then the world-space entries
                        }
                        finish();
                        break;
                    }
                    sp--;
                    // two separate accesses (never one pointer select: that becomes a flat load with full waitcnt drains)
                    uint2 e = my[(sp < RSPT_W4_LDS ? sp : RSPT_W4_LDS - 1u) * BLOCK];
                    asm volatile("" : "+v"(e.x), "+v"(e.y));  // pins the LDS read: without it the two accesses are merged into flat loads again
                    if (sp >= RSPT_W4_LDS) e = my_spill[(size_t)(sp - RSPT_W4_LDS) * spill_stride];
                    if (__uint_as_float(e.y) < t_max) {  // the reference's box test at this later moment (bvh.rs:424)
                        if (e.x & RSPT_REF_LEAF) leaf = e.x;
                        else ridx = e.x;
                        break;
                    }
                }
            }
            if (ridx != RSPT_NONE) {
                float4 a0, a1, a2, a3, a4, a5, rf;
                if (TOPCAP > 0 && ridx < n_top) {
                    const float4* lp = top + ridx;
                    a0 = lp[0]; a1 = lp[TOPCAP]; a2 = lp[2 * TOPCAP]; a3 = lp[3 * TOPCAP];
                    a4 = lp[4 * TOPCAP]; a5 = lp[5 * TOPCAP]; rf = lp[6 * TOPCAP];
                    // keeps the two branches from being merged into one set of flat loads through a selected pointer
                    // (flat loads of LDS drain every counter and were 2.6x slower)
                    asm volatile("" : "+v"(rf.w));
                } else {
                    const float4* pp = reinterpret_cast<const float4*>(recs + ridx);
                    a0 = pp[0]; a1 = pp[1]; a2 = pp[2]; a3 = pp[3]; a4 = pp[4]; a5 = pp[5]; rf = pp[6];
                }
                cur = RSPT_NONE;
                bool h0, h1, h2, h3;
                float m0, m1, m2, m3;
                if (!(negbits & 8u)) {
                    box_pair_hit_m(a0, a1, a2, ox, oy, oz, ix, iy, iz, t_max, &h0, &h1, &m0, &m1);
                    box_pair_hit_m(a3, a4, a5, ox, oy, oz, ix, iy, iz, t_max, &h2, &h3, &m2, &m3);
                } else {
                    const f3 o{ox, oy, oz}, inv{ix, iy, iz};
                    const bool n0 = negbits & 1u, n1 = negbits & 2u, n2 = negbits & 4u;
                    h0 = box_hit6_m(a0.x, a1.x, a2.x, a0.z, a1.z, a2.z, o, inv, n0, n1, n2, t_max, &m0);
                    h1 = box_hit6_m(a0.y, a1.y, a2.y, a0.w, a1.w, a2.w, o, inv, n0, n1, n2, t_max, &m1);
                    h2 = box_hit6_m(a3.x, a4.x, a5.x, a3.z, a4.z, a5.z, o, inv, n0, n1, n2, t_max, &m2);
                    h3 = box_hit6_m(a3.y, a4.y, a5.y, a3.w, a4.w, a5.w, o, inv, n0, n1, n2, t_max, &m3);
                }
                const uint32_t f0 = __float_as_uint(rf.x), f1 = __float_as_uint(rf.y), f2 = __float_as_uint(rf.z), f3w = __float_as_uint(rf.w);
                const uint32_t r0 = h0 ? (f0 & ~RSPT_W4_AXIS_MASK) : RSPT_NONE, r1 = h1 ? (f1 & ~RSPT_W4_AXIS_MASK) : RSPT_NONE;
                const uint32_t r2 = h2 ? (f2 & ~RSPT_W4_AXIS_MASK) : RSPT_NONE, r3 = h3 ? f3w : RSPT_NONE;
                const bool sA = ((negbits >> ((f0 >> RSPT_W4_AXIS_SHIFT) & 3u)) & 1u) != 0;   // dir_is_neg[A.axis]: second child's subtree first
                const bool sB0 = ((negbits >> ((f1 >> RSPT_W4_AXIS_SHIFT) & 3u)) & 1u) != 0;  // order inside the first child
                const bool sB1 = ((negbits >> ((f2 >> RSPT_W4_AXIS_SHIFT) & 3u)) & 1u) != 0;  // order inside the second child
                const uint32_t g0n = sB0 ? r1 : r0, g0f = sB0 ? r0 : r1, g1n = sB1 ? r3 : r2, g1f = sB1 ? r2 : r3;
                const float mg0n = sB0 ? m1 : m0, mg0f = sB0 ? m0 : m1, mg1n = sB1 ? m3 : m2, mg1f = sB1 ? m2 : m3;
                // visiting order e0, e1, e2, e3
                const uint32_t e0 = sA ? g1n : g0n, e1 = sA ? g1f : g0f, e2 = sA ? g0n : g1n, e3 = sA ? g0f : g1f;
                const float me1 = sA ? mg1f : mg0f, me2 = sA ? mg0n : mg1n, me3 = sA ? mg0f : mg1f;
                const bool v0 = e0 != RSPT_NONE, v1 = e1 != RSPT_NONE, v2 = e2 != RSPT_NONE, v3 = e3 != RSPT_NONE;
                const bool p1 = v0, p2 = v0 || v1, p3 = p2 || v2;  // something earlier in the order is visited first
                uint32_t next = v0 ? e0 : (v1 ? e1 : (v2 ? e2 : e3));
                const bool push3 = v3 && p3, push2 = v2 && p2, push1 = v1 && p1;
                if (sp <= RSPT_W4_LDS - 3) {  // the common case: everything fits the LDS column
                    if (push3) { my[sp * BLOCK] = make_uint2(e3, __float_as_uint(me3)); sp++; }
                    if (push2) { my[sp * BLOCK] = make_uint2(e2, __float_as_uint(me2)); sp++; }
                    if (push1) { my[sp * BLOCK] = make_uint2(e1, __float_as_uint(me1)); sp++; }
                } else if (sp + (push3 ? 1u : 0u) + (push2 ? 1u : 0u) + (push1 ? 1u : 0u) > RSPT_W4_LDS + spill_rows) {
                    best = RSPT_RETRACE;  // deeper than LDS column + spill rows: k_trace_fixup redoes this ray
                    overflow_list[atomicAdd(n_overflow, 1u)] = OUT_MODE == 0 ? entry : qpos;
                    finish();
                    next = RSPT_NONE;
                } else {
                    auto push = [&](uint32_t ref, float m) {
                        if (sp < RSPT_W4_LDS) {
                            my[sp * BLOCK] = make_uint2(ref, __float_as_uint(m));
                            asm volatile("");  // keeps the LDS store and the global store apart (no flat store through a selected pointer)
                        } else
                            my_spill[(size_t)(sp - RSPT_W4_LDS) * spill_stride] = make_uint2(ref, __float_as_uint(m));
                        sp++;
                    };
                    if (push3) push(e3, me3);
                    if (push2) push(e2, me2);
                    if (push1) push(e1, me1);
                }
                if (next != RSPT_NONE) {
                    if (next & RSPT_REF_LEAF) leaf = next;
                    else cur = next;
                }
            }
        }

        // ---- leaf phase: watertight triangle tests for parked lanes, when enough of them wait ----
        const uint64_t parked = __ballot(active && leaf != RSPT_NONE && (!ANIM || leaf != RSPT_W4_ENTER));
        if (parked) {
            const uint64_t running = __ballot(active && leaf == RSPT_NONE);
            if (__popcll(parked) >= leaf_thresh || running == 0) {
                if (active && leaf != RSPT_NONE && (!ANIM || leaf != RSPT_W4_ENTER)) {
                    uint32_t offset = leaf & RSPT_W4_OFFSET_MASK, n_prims = ((leaf >> RSPT_W4_COUNT_SHIFT) & 15u) + 1u;
                    if (n_prims == 16u) {
                        const uint2 bl = big_leaves[offset];
                        offset = bl.x; n_prims = bl.y;
                    }
                    leaf = RSPT_NONE;
                    const f3 o{ox, oy, oz};
                    for (uint32_t i = 0; i < n_prims; i++) {
                        uint32_t pi = offset + i;
                        float4 a = sc.tris[3 * (size_t)pi], b = sc.tris[3 * (size_t)pi + 1], c = sc.tris[3 * (size_t)pi + 2];
                        if (INST && (__float_as_uint(c.w) & MF_INSTANCE)) {  // TransformedPrimitive::intersect / intersect_p
                            const uint32_t cont = __float_as_uint(a.y);
                            if (cont != RSPT_NONE) {  // the rest of this leaf waits on the stack below the object's entries; -inf: it is always due
                                if (sp >= RSPT_W4_LDS + spill_rows) {
                                    best = RSPT_RETRACE;
                                    overflow_list[atomicAdd(n_overflow, 1u)] = OUT_MODE == 0 ? entry : qpos;
                                    finish();
                                    break;
                                }
                                if (sp < RSPT_W4_LDS) {
                                    my[sp * BLOCK] = make_uint2(cont, 0xff800000u);
                                    asm volatile("");
                                } else
                                    my_spill[(size_t)(sp - RSPT_W4_LDS) * spill_stride] = make_uint2(cont, 0xff800000u);
                                sp++;
                            }
                            inst = __float_as_uint(a.x);
                            if constexpr (ANIM) leaf = RSPT_W4_ENTER;   // (the entry phase below)
                            else
#include "trace_w4_enter.h"
                            break;
                        }
                        float t, b0, b1, b2;
                        if (tri_test(f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c.x}, INST ? f3{ox, oy, oz} : o, rs, t_max, &t, &b0, &b1, &b2)) {
                            bool there = true;
                            if (ALPHA && (__float_as_uint(c.w) & MF_ALPHA)) {
                                // (in line, the test must not see through (t, b): without this pin the <closest, INST> instantiations came out of hipcc 7.2 keeping
                                //  the PREVIOUS hit's (t, b0, b1, b2) next to the new primitive index — tests/test_alpha_masks.py [instanced] catches it; the
                                //  not-inlined build of alpha_simple was right but 4 % slower on the masked soup)
                                if constexpr (ALPHA == 2) asm volatile("" : "+v"(t), "+v"(b0), "+v"(b1), "+v"(b2));
                                if constexpr (ALPHA == 2) there = alpha_simple<ANY>(sc, tt, pi, __float_as_uint(c.w), f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c.x}, b0, b1, b2);
                                else there = alpha_pass<ANY>(sc, tt, pi, f3{a.x, a.y, a.z}, f3{a.w, b.x, b.y}, f3{b.z, b.w, c.x}, b0, b1, b2);
                            }
                            if (there) {
                                if (ANY) { best = 0; break; }
                                t_max = t;       // primitive.rs:155: later pops compare their t_min with this
                                best = pi; bt = t; bb0 = b0; bb1 = b1; bb2 = b2;
                                if (INST) {
                                    if (inst != RSPT_NONE) { best_inst = inst + 1u; inst_hit = true; }
                                    else { best_inst = 0; hitflag = true; }
                                }
                            }
                        }
                    }
                    if (ANY && best != RSPT_MISS) finish();
                }
            }
        }

        // ---- entry phase (ANIM): lanes parked in front of an instance, when enough of them wait or no other lane can move ----
        if constexpr (ANIM) {
            const uint64_t waiting = __ballot(active && leaf == RSPT_W4_ENTER);
            if (waiting) {
                const uint64_t others = __ballot(active && leaf != RSPT_W4_ENTER);
                if (__popcll(waiting) >= enter_thresh || others == 0) {
                    if (active && leaf == RSPT_W4_ENTER) {
                        leaf = RSPT_NONE;
#include "trace_w4_enter.h"
                    }
                }
            }
        }
    }
}

}  // namespace rspt
