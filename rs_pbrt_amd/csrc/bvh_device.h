// Device-side BVH construction: BVHAccel::new with SplitMethod::SAH (src/accelerators/bvh.rs:96-392)
// rebuilt level by level on the GPU, producing bit for bit the LinearBVHNode array and primitive order of the
// reference's recursion (= rspt_bvh_build, bvh_build.cpp).  Nothing in the recursion depends on the order in which
// sibling subtrees are processed, so one pass over all primitives handles every node of a tree level at once:
//   bounds + centroid bounds (segmented min / max)  ->  axis, early leaves        (bvh.rs:196-229)
//   12-bucket counts + bounds                        ->  SAH costs, split or leaf   (:247-296)
//   order-preserving partition = segmented exclusive scan of the "goes left" flags  (:297-320, Iterator::partition)
// Unions of bounds are min / max (exact, order-free) and the costs are computed per node from the bucket sums with
// the reference's expression, so the floating-point results cannot differ.  Leaves write their primitives into
// ordered_prims at the slot range the recursion would have used: the RIGHT subtree is emitted first (:333-352).
// The depth-first node numbering (first child = own index + 1, :358-392) is recovered afterwards from subtree sizes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rspt.h"

namespace rspt {
namespace bvhdev {

#define BVD_NONE 0xffffffffu
#define BVD_FMAX 3.402823466e+38f

// order-preserving float <-> uint map for atomicMin / atomicMax
__device__ __forceinline__ uint32_t f2ord(float f) { uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

struct Prims {  // BVHPrimitiveInfo (bvh.rs:27-42) in the current order, SoA
    float* lo[3];
    float* hi[3];
    float* c[3];
    uint32_t* prim;
    uint32_t* node;  // build node that owns this position at the current level, BVD_NONE once it sits in a leaf
};

struct Node {  // BVHBuildNode (bvh.rs:44-69) plus the recursion's arguments
    uint32_t start, end;    // position range in the primitive order
    uint32_t base;          // first slot of this subtree in ordered_prims
    uint32_t child0, child1;
    uint32_t axis;
    uint32_t leaf;          // 1 once create_leaf ran
    uint32_t state;         // per-level scratch: 0 undecided, 1 leaf, 2 pair (n == 2), 3 SAH candidate, 4 split
    uint32_t min_bucket, swap;
    uint32_t size, index;   // subtree size, depth-first index (flattening)
    uint32_t b[6];          // bounds, ordered-uint encoded while accumulating (lo xyz, hi xyz)
    uint32_t cb[6];         // centroid bounds
};
struct Buckets {
    uint32_t count[12];
    uint32_t lo[12][3], hi[12][3];
};

RSPT_PLAIN_KERNEL void k_prim_info(const float* __restrict__ P, const uint32_t* __restrict__ tri, uint32_t n, Prims pr) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p0 = P + 3 * (size_t)tri[3 * (size_t)i];
    const float* p1 = P + 3 * (size_t)tri[3 * (size_t)i + 1];
    const float* p2 = P + 3 * (size_t)tri[3 * (size_t)i + 2];
    for (int k = 0; k < 3; k++) {  // Triangle::world_bound (triangle.rs:126-133), centroid = 0.5 lo + 0.5 hi (bvh.rs:39)
        float lo = fminf(fminf(p0[k], p1[k]), p2[k]), hi = fmaxf(fmaxf(p0[k], p1[k]), p2[k]);
        pr.lo[k][i] = lo; pr.hi[k][i] = hi;
        pr.c[k][i] = lo * 0.5f + hi * 0.5f;
    }
    pr.prim[i] = i;
    pr.node[i] = 0u;
}

RSPT_PLAIN_KERNEL void k_node_reset(Node* nodes, const uint32_t* __restrict__ level_nodes, uint32_t n_level) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_level) return;
    Node& nd = nodes[level_nodes[k]];
    for (int a = 0; a < 3; a++) { nd.b[a] = f2ord(BVD_FMAX); nd.b[3 + a] = f2ord(-BVD_FMAX); nd.cb[a] = f2ord(BVD_FMAX); nd.cb[3 + a] = f2ord(-BVD_FMAX); }
    nd.state = 0;
}

// wave-level combine when the whole wave works for one node (top levels), else one atomic per lane
__device__ __forceinline__ void seg_min(uint32_t* addr, uint32_t v, bool uniform) {
    if (uniform) {
        for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off); v = o < v ? o : v; }
        if (__lane_id() == 0) atomicMin(addr, v);
    } else
        atomicMin(addr, v);
}
__device__ __forceinline__ void seg_max(uint32_t* addr, uint32_t v, bool uniform) {
    if (uniform) {
        for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off); v = o > v ? o : v; }
        if (__lane_id() == 0) atomicMax(addr, v);
    } else
        atomicMax(addr, v);
}
__device__ __forceinline__ bool wave_uniform(uint32_t node, bool active) {
    // all 64 lanes active and on the same node
    if (__ballot(active) != ~0ull) return false;
    uint32_t first = __builtin_amdgcn_readfirstlane(node);
    return __ballot(node == first) == ~0ull;
}

RSPT_PLAIN_KERNEL void k_bounds(Prims pr, uint32_t n, Node* nodes) {  // bvh.rs:196-199, 211-216
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < n;
    const uint32_t node = in ? pr.node[i] : BVD_NONE;
    const bool active = in && node != BVD_NONE;
    // whole workgroup on one node (upper levels): reduce in LDS, 12 global atomics per workgroup
    __shared__ uint32_t s_acc[12];
    __shared__ uint32_t s_node;
    if (threadIdx.x == 0) s_node = node;
    if (threadIdx.x < 12) s_acc[threadIdx.x] = (threadIdx.x % 6) < 3 ? f2ord(BVD_FMAX) : f2ord(-BVD_FMAX);  // [b lo, b hi, cb lo, cb hi]
    __syncthreads();
    if (__syncthreads_and(active && node == s_node)) {
        for (int a = 0; a < 3; a++) {
            uint32_t v;
            v = f2ord(pr.lo[a][i]); for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off); v = o < v ? o : v; } if (__lane_id() == 0) atomicMin(&s_acc[a], v);
            v = f2ord(pr.hi[a][i]); for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off); v = o > v ? o : v; } if (__lane_id() == 0) atomicMax(&s_acc[3 + a], v);
            const uint32_t c = f2ord(pr.c[a][i]);
            v = c; for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off); v = o < v ? o : v; } if (__lane_id() == 0) atomicMin(&s_acc[6 + a], v);
            v = c; for (int off = 32; off > 0; off >>= 1) { uint32_t o = __shfl_xor(v, off); v = o > v ? o : v; } if (__lane_id() == 0) atomicMax(&s_acc[9 + a], v);
        }
        __syncthreads();
        Node& nb = nodes[s_node];
        if (threadIdx.x < 3) atomicMin(&nb.b[threadIdx.x], s_acc[threadIdx.x]);
        else if (threadIdx.x < 6) atomicMax(&nb.b[threadIdx.x], s_acc[threadIdx.x]);
        else if (threadIdx.x < 9) atomicMin(&nb.cb[threadIdx.x - 6], s_acc[threadIdx.x]);
        else if (threadIdx.x < 12) atomicMax(&nb.cb[threadIdx.x - 6], s_acc[threadIdx.x]);
        return;
    }
    const bool uni = wave_uniform(node, active);
    if (!active) return;
    Node& nd = nodes[node];
    for (int a = 0; a < 3; a++) {
        seg_min(&nd.b[a], f2ord(pr.lo[a][i]), uni);
        seg_max(&nd.b[3 + a], f2ord(pr.hi[a][i]), uni);
        const uint32_t c = f2ord(pr.c[a][i]);
        seg_min(&nd.cb[a], c, uni);
        seg_max(&nd.cb[3 + a], c, uni);
    }
}

__device__ __forceinline__ float box_area(const float lo[3], const float hi[3]) {  // Bounds3::surface_area geometry.rs:2126-2131
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    float r = dx * dy + dx * dz + dy * dz;
    return r + r;
}
__device__ __forceinline__ uint32_t bucket_of(float lo, float hi, float c) {  // bvh.rs:252-258
    float o = c - lo;
    if (hi > lo) o /= hi - lo;
    float v = 12.0f * o;
    uint32_t b = (v != v || v <= 0.0f) ? 0u : (v >= 4294967296.0f ? 0xffffffffu : (uint32_t)v);  // `as usize`, then the == 12 / assert clamp
    return b > 11u ? 11u : b;
}

// per node: axis, the early outs of recursive_build (n == 1; all centroids equal), n == 2
RSPT_PLAIN_KERNEL void k_node_axis(Node* nodes, const uint32_t* __restrict__ level_nodes, uint32_t n_level, Buckets* buckets, Prims pr, uint32_t* n_sah) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_level) return;
    Node& nd = nodes[level_nodes[k]];
    const uint32_t n = nd.end - nd.start;
    float clo[3], chi[3];
    for (int a = 0; a < 3; a++) { clo[a] = ord2f(nd.cb[a]); chi[a] = ord2f(nd.cb[3 + a]); }
    float dx = chi[0] - clo[0], dy = chi[1] - clo[1], dz = chi[2] - clo[2];
    int dim = (dx > dy && dx > dz) ? 0 : (dy > dz ? 1 : 2);  // maximum_extent geometry.rs:2132-2144
    nd.axis = (uint32_t)dim;
    nd.swap = 0;
    if (n == 1 || chi[dim] == clo[dim]) nd.state = 1;
    else if (n <= 2) {
        nd.state = 2;  // mid = start + 1; the two are exchanged when the second centroid is smaller (select_nth of 2)
        nd.swap = pr.c[dim][nd.end - 1] < pr.c[dim][nd.start] ? 1u : 0u;
    } else {
        nd.state = 3;
        const uint32_t slot = atomicAdd(n_sah, 1u);  // at most n / 3 nodes of a level have more than two primitives
        nd.min_bucket = slot;                        // (index of this node's bucket set while state == 3)
        Buckets& bk = buckets[slot];
        for (int j = 0; j < 12; j++) {
            bk.count[j] = 0;
            for (int a = 0; a < 3; a++) { bk.lo[j][a] = f2ord(BVD_FMAX); bk.hi[j][a] = f2ord(-BVD_FMAX); }
        }
    }
}

RSPT_PLAIN_KERNEL void k_buckets(Prims pr, uint32_t n, const Node* __restrict__ nodes, Buckets* buckets) {  // bvh.rs:247-265
    // A workgroup whose 256 primitives all belong to one node (the rule on the upper levels, where a handful of nodes own
    // everything) accumulates in LDS and issues 84 global atomics instead of 1792.
    __shared__ Buckets s_bk;
    __shared__ uint32_t s_node;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t node = i < n ? pr.node[i] : BVD_NONE;
    if (threadIdx.x == 0) s_node = node;
    __syncthreads();
    const bool uniform = __syncthreads_and(node == s_node) != 0 && s_node != BVD_NONE;
    if (!uniform) {
        if (node == BVD_NONE) return;
        const Node& nd = nodes[node];
        if (nd.state != 3) return;
        const int dim = (int)nd.axis;
        const uint32_t b = bucket_of(ord2f(nd.cb[dim]), ord2f(nd.cb[3 + dim]), pr.c[dim][i]);
        Buckets& bk = buckets[nd.min_bucket];
        atomicAdd(&bk.count[b], 1u);
        for (int a = 0; a < 3; a++) {
            atomicMin(&bk.lo[b][a], f2ord(pr.lo[a][i]));
            atomicMax(&bk.hi[b][a], f2ord(pr.hi[a][i]));
        }
        return;
    }
    const Node& nd = nodes[node];
    if (nd.state != 3) return;  // (block-uniform)
    for (uint32_t j = threadIdx.x; j < 12; j += blockDim.x) {
        s_bk.count[j] = 0;
        for (int a = 0; a < 3; a++) { s_bk.lo[j][a] = f2ord(BVD_FMAX); s_bk.hi[j][a] = f2ord(-BVD_FMAX); }
    }
    __syncthreads();
    const int dim = (int)nd.axis;
    const uint32_t b = bucket_of(ord2f(nd.cb[dim]), ord2f(nd.cb[3 + dim]), pr.c[dim][i]);
    atomicAdd(&s_bk.count[b], 1u);
    for (int a = 0; a < 3; a++) {
        atomicMin(&s_bk.lo[b][a], f2ord(pr.lo[a][i]));
        atomicMax(&s_bk.hi[b][a], f2ord(pr.hi[a][i]));
    }
    __syncthreads();
    Buckets& bk = buckets[nd.min_bucket];
    for (uint32_t j = threadIdx.x; j < 12; j += blockDim.x) {
        if (s_bk.count[j] == 0) continue;
        atomicAdd(&bk.count[j], s_bk.count[j]);
        for (int a = 0; a < 3; a++) { atomicMin(&bk.lo[j][a], s_bk.lo[j][a]); atomicMax(&bk.hi[j][a], s_bk.hi[j][a]); }
    }
}

// SAH decision per node (bvh.rs:266-296)
RSPT_PLAIN_KERNEL void k_split(Node* nodes, const uint32_t* __restrict__ level_nodes, uint32_t n_level, const Buckets* __restrict__ buckets, uint32_t max_prims) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_level) return;
    Node& nd = nodes[level_nodes[k]];
    if (nd.state != 3) return;
    const Buckets& bk = buckets[nd.min_bucket];
    const uint32_t n = nd.end - nd.start;
    float blo[3], bhi[3];
    for (int a = 0; a < 3; a++) { blo[a] = ord2f(nd.b[a]); bhi[a] = ord2f(nd.b[3 + a]); }
    const float total_area = box_area(blo, bhi);
    float min_cost = 0.0f;
    uint32_t min_b = 0;
    for (int i = 0; i < 11; i++) {
        float l0[3] = {BVD_FMAX, BVD_FMAX, BVD_FMAX}, h0[3] = {-BVD_FMAX, -BVD_FMAX, -BVD_FMAX};
        float l1[3] = {BVD_FMAX, BVD_FMAX, BVD_FMAX}, h1[3] = {-BVD_FMAX, -BVD_FMAX, -BVD_FMAX};
        uint32_t c0 = 0, c1 = 0;
        for (int j = 0; j < 12; j++) {
            float* l = j <= i ? l0 : l1;
            float* h = j <= i ? h0 : h1;
            for (int a = 0; a < 3; a++) { l[a] = fminf(l[a], ord2f(bk.lo[j][a])); h[a] = fmaxf(h[a], ord2f(bk.hi[j][a])); }
            if (j <= i) c0 += bk.count[j]; else c1 += bk.count[j];
        }
        const float cost = 1.0f + ((float)c0 * box_area(l0, h0) + (float)c1 * box_area(l1, h1)) / total_area;
        if (i == 0) { min_cost = cost; min_b = 0; }
        else if (cost < min_cost) { min_cost = cost; min_b = (uint32_t)i; }
    }
    if (!(n > max_prims || min_cost < (float)n)) nd.state = 1;  // leaf
    else { nd.state = 4; nd.min_bucket = min_b; }
}

// leaves emit their primitives; everyone else marks who goes to the first child
RSPT_PLAIN_KERNEL void k_flags(Prims pr, uint32_t n, Node* nodes, uint32_t* __restrict__ flag, uint32_t* __restrict__ ordered) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t f = 0;
    const uint32_t node = pr.node[i];
    if (node != BVD_NONE) {
        Node& nd = nodes[node];
        if (nd.state == 1) {  // create_leaf (bvh.rs:201-210, 219-229, 322-331)
            ordered[nd.base + (i - nd.start)] = pr.prim[i];
            pr.node[i] = BVD_NONE;
            if (i == nd.start) nd.leaf = 1;
        } else if (nd.state == 2) {
            f = ((i == nd.start) != (nd.swap != 0)) ? 1u : 0u;
        } else {
            const int dim = (int)nd.axis;
            f = bucket_of(ord2f(nd.cb[dim]), ord2f(nd.cb[3 + dim]), pr.c[dim][i]) <= nd.min_bucket ? 1u : 0u;
        }
    }
    flag[i] = f;
}

// ---- exclusive scan of n uint32 (three passes, 1024 elements per block) ----
#define BVD_SCAN_BLOCK 256
#define BVD_SCAN_ITEMS 4
RSPT_PLAIN_KERNEL void k_scan_blocks(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t* __restrict__ block_sums, uint32_t n) {
    __shared__ uint32_t s[BVD_SCAN_BLOCK];
    const uint32_t base = (blockIdx.x * BVD_SCAN_BLOCK + threadIdx.x) * BVD_SCAN_ITEMS;
    uint32_t v[BVD_SCAN_ITEMS], sum = 0;
    for (int k = 0; k < BVD_SCAN_ITEMS; k++) { v[k] = base + k < n ? in[base + k] : 0u; sum += v[k]; }
    s[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t off = 1; off < BVD_SCAN_BLOCK; off <<= 1) {  // Hillis-Steele inclusive scan of the thread sums
        uint32_t t = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
        __syncthreads();
        s[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = s[threadIdx.x] - sum;
    for (int k = 0; k < BVD_SCAN_ITEMS; k++) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == BVD_SCAN_BLOCK - 1) block_sums[blockIdx.x] = s[threadIdx.x];
}
RSPT_PLAIN_KERNEL void k_scan_sums(uint32_t* __restrict__ block_sums, uint32_t n_blocks, uint32_t* __restrict__ total) {  // one block
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    __shared__ uint32_t s[1024];
    for (uint32_t base = 0; base < n_blocks; base += 1024) {
        uint32_t i = base + threadIdx.x;
        uint32_t v = i < n_blocks ? block_sums[i] : 0u;
        s[threadIdx.x] = v;
        __syncthreads();
        for (uint32_t off = 1; off < 1024; off <<= 1) {
            uint32_t t = threadIdx.x >= off ? s[threadIdx.x - off] : 0u;
            __syncthreads();
            s[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < n_blocks) block_sums[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}
RSPT_PLAIN_KERNEL void k_scan_add(uint32_t* __restrict__ out, const uint32_t* __restrict__ block_sums, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += block_sums[i / (BVD_SCAN_BLOCK * BVD_SCAN_ITEMS)];
}

// children of the nodes that split: their ranges follow from the scan (mid = start + number of "first child" flags)
RSPT_PLAIN_KERNEL void k_children(Node* nodes, const uint32_t* __restrict__ level_nodes, uint32_t n_level, const uint32_t* __restrict__ scan,
                           const uint32_t* __restrict__ scan_total, uint32_t n, uint32_t* n_nodes, uint32_t* __restrict__ next_level, uint32_t* n_next) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_level) return;
    Node& nd = nodes[level_nodes[k]];
    if (nd.state != 2 && nd.state != 4) return;
    const uint32_t s_end = nd.end < n ? scan[nd.end] : *scan_total;
    const uint32_t mid = nd.start + (s_end - scan[nd.start]);
    const uint32_t c = atomicAdd(n_nodes, 2u);
    const uint32_t slot = atomicAdd(n_next, 2u);
    nd.child0 = c; nd.child1 = c + 1;
    Node l{}, r{};
    l.start = nd.start; l.end = mid; l.base = nd.base + (nd.end - mid);  // the right subtree is emitted first (bvh.rs:333-352)
    r.start = mid; r.end = nd.end; r.base = nd.base;
    l.child0 = l.child1 = r.child0 = r.child1 = BVD_NONE;
    nodes[c] = l; nodes[c + 1] = r;
    next_level[slot] = c; next_level[slot + 1] = c + 1;
    nd.min_bucket = mid;  // (now: the split position, for k_scatter)
}

RSPT_PLAIN_KERNEL void k_scatter(Prims src, Prims dst, uint32_t n, const Node* __restrict__ nodes, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ scan) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t node = src.node[i];
    if (node == BVD_NONE) { dst.node[i] = BVD_NONE; return; }  // (positions inside leaves are never read again)
    const Node& nd = nodes[node];
    const uint32_t mid = nd.min_bucket;
    const uint32_t left_rank = scan[i] - scan[nd.start];
    const uint32_t to = flag[i] ? nd.start + left_rank : mid + ((i - nd.start) - left_rank);
    for (int a = 0; a < 3; a++) { dst.lo[a][to] = src.lo[a][i]; dst.hi[a][to] = src.hi[a][i]; dst.c[a][to] = src.c[a][i]; }
    dst.prim[to] = src.prim[i];
    dst.node[to] = flag[i] ? nd.child0 : nd.child1;
}

// ---- flattening (bvh.rs:358-392): subtree sizes bottom-up, depth-first indices top-down, one level per launch ----
RSPT_PLAIN_KERNEL void k_sizes(Node* nodes, const uint32_t* __restrict__ level_nodes, uint32_t n_level) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_level) return;
    Node& nd = nodes[level_nodes[k]];
    nd.size = nd.leaf ? 1u : 1u + nodes[nd.child0].size + nodes[nd.child1].size;
}
RSPT_PLAIN_KERNEL void k_indices(Node* nodes, const uint32_t* __restrict__ level_nodes, uint32_t n_level, rspt_bvh_node* __restrict__ out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_level) return;
    Node& nd = nodes[level_nodes[k]];
    rspt_bvh_node ln;
    for (int a = 0; a < 3; a++) { ln.bmin[a] = ord2f(nd.b[a]); ln.bmax[a] = ord2f(nd.b[3 + a]); }
    ln.pad = 0;
    if (nd.leaf) {
        ln.offset = (int32_t)nd.base; ln.n_prims = (uint16_t)(nd.end - nd.start); ln.axis = 0;
    } else {
        nodes[nd.child0].index = nd.index + 1;
        nodes[nd.child1].index = nd.index + 1 + nodes[nd.child0].size;
        ln.offset = (int32_t)(nd.index + 1 + nodes[nd.child0].size); ln.n_prims = 0; ln.axis = (uint8_t)nd.axis;
    }
    out[nd.index] = ln;
}

}  // namespace bvhdev
}  // namespace rspt
