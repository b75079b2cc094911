// Host-side BVH construction for callers that do not bring rs_pbrt's own BVHAccel (bench.py,
// tests, stand-alone tools).  Produces exactly the LinearBVHNode array and primitive order that
// BVHAccel::new builds with SplitMethod::SAH (src/accelerators/bvh.rs:96-392):
//   * 12-bucket surface-area heuristic on the largest centroid extent (:247-292),
//   * leaf when n <= max_prims_in_node and leaf cost <= split cost (:296),
//   * ORDER-PRESERVING partition (Iterator::partition, :297-320),
//   * the RIGHT child is built first, so ordered_prims holds right-subtree leaves first (:333-352),
//   * depth-first flattening, first child at own index + 1 (:358-392).
// Unlike the reference's single-threaded recursion this builder is task-parallel: the slot range
// a subtree occupies in ordered_prims is known before it is built (right subtree first, sizes
// known), so disjoint subtrees are built concurrently and the result is identical.
// All arithmetic is f32 without FMA contraction (-ffp-contract=off).
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/rspt.h"

namespace {

struct B3 {
    float lo[3], hi[3];
};
inline B3 empty_box() { return B3{{3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f}, {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f}}; }
inline void grow(B3& b, const B3& o) {
    for (int i = 0; i < 3; i++) { b.lo[i] = fminf(b.lo[i], o.lo[i]); b.hi[i] = fmaxf(b.hi[i], o.hi[i]); }
}
inline void grow(B3& b, const float p[3]) {
    for (int i = 0; i < 3; i++) { b.lo[i] = fminf(b.lo[i], p[i]); b.hi[i] = fmaxf(b.hi[i], p[i]); }
}
inline float area(const B3& b) {  // Bounds3::surface_area geometry.rs:2126-2131
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    float r = dx * dy + dx * dz + dy * dz;
    return r + r;
}
inline int max_extent(const B3& b) {  // geometry.rs:2132-2144
    float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (dx > dy && dx > dz) return 0;
    return dy > dz ? 1 : 2;
}

struct Info {  // BVHPrimitiveInfo bvh.rs:27-42
    uint32_t prim;
    B3 box;
    float c[3];
};

struct Node {
    B3 box;
    uint32_t child0 = 0, child1 = 0;  // pool indices; 0 = none (root is never a child)
    uint32_t first = 0, count = 0;
    uint8_t axis = 0;
};

struct Task {
    uint32_t node;
    size_t start, end, base;  // info range and first slot in `ordered`
};

struct Builder {
    std::vector<Info> info;
    std::vector<Node> pool;
    std::atomic<uint32_t> n_nodes{0};
    uint32_t* ordered;
    size_t max_prims;
    // task queue
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Task> queue;
    size_t outstanding = 0;
    bool done = false;
    static constexpr size_t kParallelAbove = 1 << 14;

    uint32_t alloc_node() { return n_nodes.fetch_add(1); }

    static size_t bucket_of(const B3& cb, const float c[3], int dim) {  // bvh.rs:252-258
        float o = c[dim] - cb.lo[dim];
        if (cb.hi[dim] > cb.lo[dim]) o /= cb.hi[dim] - cb.lo[dim];
        float v = 12.0f * o;
        size_t b = (v != v || v <= 0.0f) ? 0 : (v >= 1.8446744e19f ? SIZE_MAX : (size_t)v);  // `as usize`
        if (b == 12) b = 11;
        return b;
    }

    // One node of recursive_build; returns true and fills the two child tasks when it split.
    bool split(const Task& t, std::vector<Info>& scratch, Task* right, Task* left) {
        Node& node = pool[t.node];
        const size_t start = t.start, end = t.end, n = end - start;
        B3 bounds = empty_box();
        for (size_t i = start; i < end; i++) grow(bounds, info[i].box);
        auto make_leaf = [&]() {
            for (size_t i = start; i < end; i++) ordered[t.base + (i - start)] = info[i].prim;
            node.box = bounds; node.first = (uint32_t)t.base; node.count = (uint32_t)n;
            return false;
        };
        if (n == 1) return make_leaf();
        B3 cb = empty_box();
        for (size_t i = start; i < end; i++) grow(cb, info[i].c);
        const int dim = max_extent(cb);
        size_t mid = (start + end) / 2;
        if (cb.hi[dim] == cb.lo[dim]) return make_leaf();
        if (n <= 2) {
            if (info[end - 1].c[dim] < info[start].c[dim]) std::swap(info[start], info[end - 1]);
        } else {
            size_t count[12] = {0};
            B3 bb[12];
            for (auto& b : bb) b = empty_box();
            for (size_t i = start; i < end; i++) {
                size_t b = bucket_of(cb, info[i].c, dim);
                if (b > 11) b = 11;  // the reference asserts b < 12
                count[b]++;
                grow(bb[b], info[i].box);
            }
            float cost[11];
            const float total_area = area(bounds);
            for (int i = 0; i < 11; i++) {
                B3 b0 = empty_box(), b1 = empty_box();
                size_t c0 = 0, c1 = 0;
                for (int j = 0; j <= i; j++) { grow(b0, bb[j]); c0 += count[j]; }
                for (int j = i + 1; j < 12; j++) { grow(b1, bb[j]); c1 += count[j]; }
                cost[i] = 1.0f + ((float)c0 * area(b0) + (float)c1 * area(b1)) / total_area;
            }
            float min_cost = cost[0];
            size_t min_bucket = 0;
            for (int i = 0; i < 11; i++)
                if (cost[i] < min_cost) { min_cost = cost[i]; min_bucket = (size_t)i; }
            if (!(n > max_prims || min_cost < (float)n)) return make_leaf();
            scratch.clear();
            size_t w = start;
            for (size_t i = start; i < end; i++) {
                size_t b = bucket_of(cb, info[i].c, dim);
                if (b > 11) b = 11;
                if (b <= min_bucket) info[w++] = info[i];  // w <= i: no unread element is overwritten
                else scratch.push_back(info[i]);
            }
            mid = w;
            std::memcpy(static_cast<void*>(&info[mid]), scratch.data(), scratch.size() * sizeof(Info));
        }
        node.axis = (uint8_t)dim;
        node.count = 0;
        node.child0 = alloc_node();
        node.child1 = alloc_node();
        // right subtree [mid, end) is emitted first into ordered_prims
        *right = Task{node.child1, mid, end, t.base};
        *left = Task{node.child0, start, mid, t.base + (end - mid)};
        return true;
    }

    void build_local(Task root, std::vector<Info>& scratch, std::vector<Task>& stack) {
        stack.clear();
        stack.push_back(root);
        while (!stack.empty()) {
            Task t = stack.back();
            stack.pop_back();
            Task r, l;
            if (!split(t, scratch, &r, &l)) continue;
            for (const Task& c : {r, l}) {
                if (c.end - c.start > kParallelAbove) {
                    std::lock_guard<std::mutex> g(mu);
                    queue.push_back(c);
                    outstanding++;
                    cv.notify_one();
                } else
                    stack.push_back(c);
            }
        }
    }

    void worker() {
        std::vector<Info> scratch;
        std::vector<Task> stack;
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return done || !queue.empty(); });
                if (queue.empty()) return;
                t = queue.front();
                queue.pop_front();
            }
            build_local(t, scratch, stack);
            {
                std::lock_guard<std::mutex> g(mu);
                if (--outstanding == 0) { done = true; cv.notify_all(); }
            }
        }
    }
};

thread_local char g_msg[256];

}  // namespace

extern "C" const char* rspt_bvh_last_error(void) { return g_msg; }

namespace {
// shared tail of the two entry points: b.info holds every primitive's bounds and centroid
int64_t build_from_info(Builder& b, uint64_t n_prims, rspt_bvh_node* nodes_out, uint64_t nodes_cap, int32_t n_threads) {
    b.pool.resize(2 * n_prims);
    const uint32_t root = b.alloc_node();
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 64) nt = 64;
    b.queue.push_back(Task{root, 0, (size_t)n_prims, 0});
    b.outstanding = 1;
    std::vector<std::thread> th;
    for (int i = 1; i < nt; i++) th.emplace_back([&b] { b.worker(); });
    b.worker();
    for (auto& t : th) t.join();
    const uint32_t total = b.n_nodes.load();
    if (total > nodes_cap) { snprintf(g_msg, sizeof g_msg, "nodes_cap %llu < %u nodes", (unsigned long long)nodes_cap, total); return RSPT_E_INVALID; }
    // flatten_bvh_tree (bvh.rs:358-392): pre-order, second child offset patched after the first subtree
    struct Frame { uint32_t node; uint32_t parent_slot; };
    std::vector<Frame> stack;
    stack.push_back({root, 0xffffffffu});
    uint32_t next = 0;
    while (!stack.empty()) {
        Frame f = stack.back();
        stack.pop_back();
        const Node& n = b.pool[f.node];
        const uint32_t my = next++;
        if (f.parent_slot != 0xffffffffu) nodes_out[f.parent_slot].offset = (int32_t)my;  // this is a second child
        rspt_bvh_node ln;
        std::memset(&ln, 0, sizeof ln);
        std::memcpy(ln.bmin, n.box.lo, sizeof ln.bmin);
        std::memcpy(ln.bmax, n.box.hi, sizeof ln.bmax);
        if (n.count > 0) {
            ln.offset = (int32_t)n.first;
            ln.n_prims = (uint16_t)n.count;
        } else {
            ln.axis = n.axis;
            stack.push_back({n.child1, my});          // visited after the whole first subtree
            stack.push_back({n.child0, 0xffffffffu});  // first child = my + 1
        }
        nodes_out[my] = ln;
    }
    // interior bounds = union of the children's (BVHBuildNode::init_interior bvh.rs:60-68), bottom-up
    for (int64_t i = (int64_t)total - 1; i >= 0; i--) {
        rspt_bvh_node& n = nodes_out[i];
        if (n.n_prims > 0) continue;
        const rspt_bvh_node& c0 = nodes_out[i + 1];
        const rspt_bvh_node& c1 = nodes_out[n.offset];
        for (int k = 0; k < 3; k++) { n.bmin[k] = fminf(c0.bmin[k], c1.bmin[k]); n.bmax[k] = fmaxf(c0.bmax[k], c1.bmax[k]); }
    }
    return (int64_t)total;
}
}  // namespace

// Replaces: BVHAccel::new over Triangle shapes (src/accelerators/bvh.rs:96-152;
// Triangle::world_bound src/shapes/triangle.rs:126-133).
extern "C" int64_t rspt_bvh_build(const float* P, const uint32_t* tri_idx, uint64_t n_tris, uint32_t max_prims_in_node,
                                  rspt_bvh_node* nodes_out, uint64_t nodes_cap, uint32_t* ordered_out, int32_t n_threads) {
    g_msg[0] = 0;
    if (n_tris == 0) return 0;
    if (!P || !tri_idx || !nodes_out || !ordered_out) { snprintf(g_msg, sizeof g_msg, "null argument"); return RSPT_E_INVALID; }
    if (n_tris > 0x7fffffffull) { snprintf(g_msg, sizeof g_msg, "too many triangles"); return RSPT_E_UNSUPPORTED; }
    Builder b;
    b.max_prims = max_prims_in_node < 255 ? max_prims_in_node : 255;  // bvh.rs:102
    b.ordered = ordered_out;
    b.info.resize(n_tris);
    for (uint64_t i = 0; i < n_tris; i++) {
        const uint32_t* v = tri_idx + 3 * i;
        const float* p0 = P + 3 * (size_t)v[0];
        const float* p1 = P + 3 * (size_t)v[1];
        const float* p2 = P + 3 * (size_t)v[2];
        Info& in = b.info[i];
        in.prim = (uint32_t)i;
        for (int k = 0; k < 3; k++) {
            in.box.lo[k] = fminf(fminf(p0[k], p1[k]), p2[k]);
            in.box.hi[k] = fmaxf(fmaxf(p0[k], p1[k]), p2[k]);
            in.c[k] = in.box.lo[k] * 0.5f + in.box.hi[k] * 0.5f;  // bvh.rs:39
        }
    }
    return build_from_info(b, n_tris, nodes_out, nodes_cap, n_threads);
}

// The same BVHAccel::new over primitives given by their world bounds (bounds: n x (min xyz, max xyz)): the top-level
// aggregate of a scene with object instances holds TransformedPrimitives next to triangles (primitive.rs:212-215 for their bounds).
extern "C" int64_t rspt_bvh_build_bounds(const float* bounds, uint64_t n_prims, uint32_t max_prims_in_node,
                                         rspt_bvh_node* nodes_out, uint64_t nodes_cap, uint32_t* ordered_out, int32_t n_threads) {
    g_msg[0] = 0;
    if (n_prims == 0) return 0;
    if (!bounds || !nodes_out || !ordered_out) { snprintf(g_msg, sizeof g_msg, "null argument"); return RSPT_E_INVALID; }
    if (n_prims > 0x7fffffffull) { snprintf(g_msg, sizeof g_msg, "too many primitives"); return RSPT_E_UNSUPPORTED; }
    Builder b;
    b.max_prims = max_prims_in_node < 255 ? max_prims_in_node : 255;
    b.ordered = ordered_out;
    b.info.resize(n_prims);
    for (uint64_t i = 0; i < n_prims; i++) {
        Info& in = b.info[i];
        in.prim = (uint32_t)i;
        for (int k = 0; k < 3; k++) {
            in.box.lo[k] = bounds[6 * i + k];
            in.box.hi[k] = bounds[6 * i + 3 + k];
            in.c[k] = in.box.lo[k] * 0.5f + in.box.hi[k] * 0.5f;  // bvh.rs:39
        }
    }
    return build_from_info(b, n_prims, nodes_out, nodes_cap, n_threads);
}
