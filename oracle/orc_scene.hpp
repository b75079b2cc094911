// TEST INFRASTRUCTURE — CPU oracle (see orc_math.hpp header).  Scene, BVH, triangle.
#pragma once
#include <algorithm>
#include <cassert>
#include <vector>

#include "../include/rspt.h" // POD scene structs only (interface header, no product code)
#include "orc_math.hpp"
#include "orc_animated.hpp"

namespace orc {

struct alignas(128) Counters {  // one per worker thread; padded so the hot increments never share a cache line
    uint64_t nodes_visited = 0, tris_tested = 0, rays_closest = 0, rays_any = 0, bounces = 0, samples = 0,
             nan_samples = 0, mis_rays = 0;
    void add(const Counters& o) {
        nodes_visited += o.nodes_visited; tris_tested += o.tris_tested; rays_closest += o.rays_closest;
        rays_any += o.rays_any; bounces += o.bounces; samples += o.samples; nan_samples += o.nan_samples;
        mis_rays += o.mis_rays;
    }
};

// ------------------------------------------------------------------------------------------
// BVH build: src/accelerators/bvh.rs:96-392
// ------------------------------------------------------------------------------------------
struct BVHPrimitiveInfo { // bvh.rs:27-42
    size_t primitive_number;
    Bounds3 bounds;
    V3 centroid;
};
struct BuildNode { // bvh.rs:44-69
    Bounds3 bounds;
    BuildNode* child1 = nullptr;
    BuildNode* child2 = nullptr;
    uint8_t split_axis = 0;
    size_t first_prim_offset = 0, n_primitives = 0;
};

struct BVHBuilder {
    size_t max_prims_in_node;
    std::vector<BVHPrimitiveInfo> info;
    std::vector<uint32_t> ordered; // ordered_prims (indices into the input list)
    std::vector<BuildNode*> arena;
    size_t total_nodes = 0;

    ~BVHBuilder() { for (auto* n : arena) delete n; }

    size_t bucket_of(const Bounds3& cb, V3 c, int dim) const { // bvh.rs:252-258
        size_t b = f2usize(12.0f * cb.offset(c)[dim]);
        if (b == 12) b = 11;
        return b;
    }

    BuildNode* recursive_build(size_t start, size_t end) { // bvh.rs:178-357
        BuildNode* node = new BuildNode();
        arena.push_back(node);
        total_nodes += 1;
        Bounds3 bounds;
        for (size_t i = start; i < end; i++) bounds = bunion(bounds, info[i].bounds);
        size_t n_primitives = end - start;
        auto make_leaf = [&]() {
            size_t first = ordered.size();
            for (size_t i = start; i < end; i++) ordered.push_back((uint32_t)info[i].primitive_number);
            node->first_prim_offset = first;
            node->n_primitives = n_primitives;
            node->bounds = bounds;
            return node;
        };
        if (n_primitives == 1) return make_leaf();
        Bounds3 cb;
        for (size_t i = start; i < end; i++) cb = bunion(cb, info[i].centroid);
        int dim = cb.maximum_extent();
        size_t mid = (start + end) / 2;
        if (cb.p_max[dim] == cb.p_min[dim]) return make_leaf();
        // SplitMethod::SAH | HLBVH (bvh.rs:237); Middle/EqualCounts are empty stubs (Q13)
        if (n_primitives <= 2) {
            mid = (start + end) / 2;
            if (start != end - 1 && info[end - 1].centroid[dim] < info[start].centroid[dim])
                std::swap(info[start], info[end - 1]);
        } else {
            struct Bucket { size_t count = 0; Bounds3 bounds; } buckets[12];
            for (size_t i = start; i < end; i++) {
                size_t b = bucket_of(cb, info[i].centroid, dim);
                buckets[b].count += 1;
                buckets[b].bounds = bunion(buckets[b].bounds, info[i].bounds);
            }
            Float cost[11];
            for (int i = 0; i < 11; i++) {
                Bounds3 b0, b1;
                size_t c0 = 0, c1 = 0;
                for (int j = 0; j <= i; j++) { b0 = bunion(b0, buckets[j].bounds); c0 += buckets[j].count; }
                for (int j = i + 1; j < 12; j++) { b1 = bunion(b1, buckets[j].bounds); c1 += buckets[j].count; }
                cost[i] = 1.0f + ((Float)c0 * b0.surface_area() + (Float)c1 * b1.surface_area()) / bounds.surface_area();
            }
            Float min_cost = cost[0];
            size_t min_bucket = 0;
            for (int i = 0; i < 11; i++)
                if (cost[i] < min_cost) { min_cost = cost[i]; min_bucket = i; }
            Float leaf_cost = (Float)n_primitives;
            if (n_primitives > max_prims_in_node || min_cost < leaf_cost) {
                // order-preserving partition (Iterator::partition, bvh.rs:297-320; Q13)
                std::vector<BVHPrimitiveInfo> left, right;
                for (size_t i = start; i < end; i++) {
                    if (bucket_of(cb, info[i].centroid, dim) <= min_bucket) left.push_back(info[i]);
                    else right.push_back(info[i]);
                }
                mid = start + left.size();
                std::copy(left.begin(), left.end(), info.begin() + start);
                std::copy(right.begin(), right.end(), info.begin() + mid);
            } else {
                return make_leaf();
            }
        }
        // "make sure we get result for c1 before c0" (bvh.rs:333-352)
        BuildNode* c1 = recursive_build(mid, end);
        BuildNode* c0 = recursive_build(start, mid);
        node->n_primitives = 0;
        node->bounds = bunion(c0->bounds, c1->bounds);
        node->child1 = c0;
        node->child2 = c1;
        node->split_axis = (uint8_t)dim;
        return node;
    }

    static size_t flatten(BuildNode* node, std::vector<rspt_bvh_node>& nodes, size_t* offset) { // bvh.rs:358-392
        size_t my = *offset;
        *offset += 1;
        rspt_bvh_node ln;
        std::memset(&ln, 0, sizeof ln);
        ln.bmin[0] = node->bounds.p_min.x; ln.bmin[1] = node->bounds.p_min.y; ln.bmin[2] = node->bounds.p_min.z;
        ln.bmax[0] = node->bounds.p_max.x; ln.bmax[1] = node->bounds.p_max.y; ln.bmax[2] = node->bounds.p_max.z;
        if (node->n_primitives > 0) {
            ln.offset = (int32_t)node->first_prim_offset;
            ln.n_prims = (uint16_t)node->n_primitives;
            ln.axis = 0;
            nodes[my] = ln;
        } else {
            flatten(node->child1, nodes, offset);
            ln.offset = (int32_t)flatten(node->child2, nodes, offset);
            ln.n_prims = 0;
            ln.axis = node->split_axis;
            nodes[my] = ln;
        }
        return my;
    }
};

// BVHAccel::new (bvh.rs:96-152).  bounds[i] = world_bound of input primitive i.
static inline void bvh_build(const Bounds3* prim_bounds, size_t n, size_t max_prims_in_node,
                             std::vector<rspt_bvh_node>& nodes, std::vector<uint32_t>& ordered) {
    nodes.clear(); ordered.clear();
    if (n == 0) return;
    BVHBuilder b;
    b.max_prims_in_node = std::min<size_t>(max_prims_in_node, 255);
    b.info.resize(n);
    for (size_t i = 0; i < n; i++) {
        b.info[i].primitive_number = i;
        b.info[i].bounds = prim_bounds[i];
        b.info[i].centroid = prim_bounds[i].p_min * 0.5f + prim_bounds[i].p_max * 0.5f; // bvh.rs:39
    }
    b.ordered.reserve(n);
    BuildNode* root = b.recursive_build(0, n);
    nodes.resize(b.total_nodes);
    size_t off = 0;
    BVHBuilder::flatten(root, nodes, &off);
    ordered.swap(b.ordered);
}

// ------------------------------------------------------------------------------------------
// Interaction: src/core/interaction.rs:46-55,226-246
// ------------------------------------------------------------------------------------------
struct Interaction {
    V3 p, p_error, wo, n;
    Float time = 0;
    // SurfaceInteraction
    P2 uv{0, 0};
    V3 dpdu, dpdv;
    V3 sh_n, sh_dpdu, sh_dpdv; // Shading
    V3 sh_dndu{0, 0, 0}, sh_dndv{0, 0, 0}; // shading.dndu/dndv; isect.dndu/dndv themselves stay 0 (triangle.rs:322,433-434: the
                                           // values computed for the shading block shadow the outer zeros)
    // screen-space differentials (interaction.rs:388-479)
    Float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;
    V3 dpdx{0, 0, 0}, dpdy{0, 0, 0};
    int64_t prim = -1;         // isect.primitive (-1: None)
    int64_t inst = 0;          // test bookkeeping only: 1 + instance the hit lies in
    int64_t geo_prim = -1;     // test bookkeeping only: the GeometricPrimitive that was hit, also after Q11 dropped isect.primitive
    // InteractionCommon.medium_interface (interaction.rs:54): inside / outside as 0 = None or 1 + medium index.  (0, 0) stands for
    // both `None` and `Some(MediumInterface { None, None })`: get_medium cannot tell them apart.
    uint32_t med_in = 0, med_out = 0;
    bool is_medium = false;    // a MediumInteraction (n == 0, interaction.rs:174-180)
    Float phase_g = 0;         // its HenyeyGreenstein { g }
    uint32_t get_medium(V3 w) const { return dot(w, n) > 0.0f ? med_out : med_in; } // interaction.rs:95-107
    // interaction.rs:58-94
    Ray spawn_ray(V3 d) const {
        Ray r{offset_ray_origin(p, p_error, n, d), d, INF, time};
        r.medium = get_medium(d);
        return r;
    }
    Ray spawn_ray_to(const Interaction& it) const {
        V3 origin = offset_ray_origin(p, p_error, n, it.p - p);
        V3 target = offset_ray_origin(it.p, it.p_error, it.n, origin - it.p);
        V3 d = target - origin;
        Ray r{origin, d, 1.0f - SHADOW_EPSILON, time};
        r.medium = get_medium(d);
        return r;
    }
};

struct Scene;
// Texture::evaluate of float texture `ti` at an interaction (orc_texture.hpp): TriangleMesh.alpha_mask / shadow_alpha_mask
static inline Float alpha_texture_value(const Scene& sc, uint32_t ti, const Interaction& si);

// GridDensityMedium (src/media/grid.rs:17-56) as built by ::new; the functions on it are in orc_render.hpp
struct GridMedium {
    Spec sigma_a, sigma_s;
    Float g;
    int32_t nx, ny, nz;
    const Float* density;
    Float world_to_medium[16];
    Float sigma_t, inv_max_density;
};
static inline GridMedium grid_medium_new(const Float sigma_a[3], const Float sigma_s[3], Float g, int32_t nx, int32_t ny, int32_t nz, const Float world_to_medium[16], const Float* d) { // grid.rs:30-56
    GridMedium m;
    m.sigma_a = Spec(sigma_a[0], sigma_a[1], sigma_a[2]); m.sigma_s = Spec(sigma_s[0], sigma_s[1], sigma_s[2]);
    m.g = g; m.nx = nx; m.ny = ny; m.nz = nz; m.density = d;
    for (int i = 0; i < 16; i++) m.world_to_medium[i] = world_to_medium[i];
    Float max_density = 0.0f;
    for (int64_t i = 0; i < (int64_t)nx * ny * nz; i++) max_density = std::fmax(max_density, d[i]); // f32::max
    m.sigma_t = (m.sigma_s + m.sigma_a).c[0]; // [RGBEnum::Red]
    m.inv_max_density = 1.0f / max_density;
    return m;
}

struct Scene {
    rspt_scene_desc d;
    // GridDensityMedium::new runs once per medium when the scene is made (api.rs:1016-1030); prepare_media() before the first render / hook call
    mutable std::vector<GridMedium> grids;
    void prepare_media() const {
        grids.clear();
        for (uint32_t i = 0; i < d.n_media; i++) {
            const rspt_medium& m = d.media[i];
            if (m.kind == RSPT_MEDIUM_GRID) grids.push_back(grid_medium_new(m.sigma_a, m.sigma_s, m.g, m.nx, m.ny, m.nz, m.world_to_medium, m.density));
            else grids.push_back(GridMedium{});
        }
    }
    const GridMedium& grid(uint32_t medium) const { return grids[medium - 1]; }
    Bounds3 world_bound() const { // bvh.rs:394-400
        Bounds3 b;
        if (d.n_nodes) {
            b.p_min = V3{d.nodes[0].bmin[0], d.nodes[0].bmin[1], d.nodes[0].bmin[2]};
            b.p_max = V3{d.nodes[0].bmax[0], d.nodes[0].bmax[1], d.nodes[0].bmax[2]};
        }
        return b;
    }
    V3 P(uint32_t i) const { return V3{d.P[3 * i], d.P[3 * i + 1], d.P[3 * i + 2]}; }
    V3 N(uint32_t i) const { return V3{d.N[3 * i], d.N[3 * i + 1], d.N[3 * i + 2]}; }
    V3 S(uint32_t i) const { return V3{d.S[3 * i], d.S[3 * i + 1], d.S[3 * i + 2]}; }
    P2 UV(uint32_t i) const { return P2{d.UV[2 * i], d.UV[2 * i + 1]}; }

    // ---- Triangle: src/shapes/triangle.rs ----
    void get_uvs(const rspt_prim& pr, P2 uv[3]) const { // triangle.rs:97-112
        const rspt_mesh& m = d.meshes[pr.mesh];
        if (!m.has_uv || !d.UV) { uv[0] = P2{0, 0}; uv[1] = P2{1, 0}; uv[2] = P2{1, 1}; }
        else { uv[0] = UV(pr.v[0]); uv[1] = UV(pr.v[1]); uv[2] = UV(pr.v[2]); }
    }

    // The watertight test shared by intersect (triangle.rs:134-273) and intersect_p (:450-579).
    bool tri_hit_test(const rspt_prim& pr, const Ray& ray, Float* t_out, Float b[3]) const {
        V3 p0 = P(pr.v[0]), p1 = P(pr.v[1]), p2 = P(pr.v[2]);
        V3 p0t = p0 - ray.o, p1t = p1 - ray.o, p2t = p2 - ray.o;
        int kz = max_dimension(vabs(ray.d));
        int kx = kz + 1; if (kx == 3) kx = 0;
        int ky = kx + 1; if (ky == 3) ky = 0;
        V3 dd = permute(ray.d, kx, ky, kz);
        p0t = permute(p0t, kx, ky, kz); p1t = permute(p1t, kx, ky, kz); p2t = permute(p2t, kx, ky, kz);
        Float sx = -dd.x / dd.z, sy = -dd.y / dd.z, sz = 1.0f / dd.z;
        p0t.x += sx * p0t.z; p0t.y += sy * p0t.z;
        p1t.x += sx * p1t.z; p1t.y += sy * p1t.z;
        p2t.x += sx * p2t.z; p2t.y += sy * p2t.z;
        Float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
        Float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
        Float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
        if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) { // f64 fallback, triangle.rs:189-200
            double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
            e0 = (Float)(p2typ1tx - p2txp1ty);
            double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
            e1 = (Float)(p0typ2tx - p0txp2ty);
            double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
            e2 = (Float)(p1typ0tx - p1txp0ty);
        }
        if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
        Float det = e0 + e1 + e2;
        if (det == 0.0f) return false;
        p0t.z *= sz; p1t.z *= sz; p2t.z *= sz;
        Float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
        if ((det < 0.0f && (t_scaled >= 0.0f || t_scaled < ray.t_max * det)) ||
            (det > 0.0f && (t_scaled <= 0.0f || t_scaled > ray.t_max * det)))
            return false;
        Float inv_det = 1.0f / det;
        Float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
        Float t = t_scaled * inv_det;
        Float max_zt = max_component(vabs(V3{p0t.z, p1t.z, p2t.z}));
        Float delta_z = gamma(3) * max_zt;
        Float max_xt = max_component(vabs(V3{p0t.x, p1t.x, p2t.x}));
        Float max_yt = max_component(vabs(V3{p0t.y, p1t.y, p2t.y}));
        Float delta_x = gamma(5) * (max_xt + max_zt);
        Float delta_y = gamma(5) * (max_yt + max_zt);
        Float delta_e = 2.0f * (gamma(2) * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
        Float max_e = max_component(vabs(V3{e0, e1, e2}));
        Float delta_t = 3.0f * (gamma(3) * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * std::fabs(inv_det);
        if (t <= delta_t) return false;
        *t_out = t; b[0] = b0; b[1] = b1; b[2] = b2;
        return true;
    }

    // Second half of Triangle::intersect (triangle.rs:274-448): fill the SurfaceInteraction.
    void tri_fill(const rspt_prim& pr, const Ray& ray, const Float b[3], Interaction* isect) const {
        const rspt_mesh& m = d.meshes[pr.mesh];
        V3 p0 = P(pr.v[0]), p1 = P(pr.v[1]), p2 = P(pr.v[2]);
        Float b0 = b[0], b1 = b[1], b2 = b[2];
        P2 uv[3]; get_uvs(pr, uv);
        P2 duv02{uv[0].x - uv[2].x, uv[0].y - uv[2].y}, duv12{uv[1].x - uv[2].x, uv[1].y - uv[2].y};
        V3 dp02 = p0 - p2, dp12 = p1 - p2;
        Float determinant = duv02.x * duv12.y - duv02.y * duv12.x;
        bool degenerate_uv = std::fabs(determinant) < 1e-8f;
        V3 dpdu{0, 0, 0}, dpdv{0, 0, 0};
        if (!degenerate_uv) {
            Float invdet = 1.0f / determinant;
            dpdu = (dp02 * duv12.y - dp12 * duv02.y) * invdet;
            dpdv = (dp02 * -duv12.x + dp12 * duv02.x) * invdet;
        }
        if (degenerate_uv || length_squared(cross(dpdu, dpdv)) == 0.0f)
            coordinate_system(normalize(cross(p2 - p0, p1 - p0)), &dpdu, &dpdv);
        Float xs = std::fabs(b0 * p0.x) + std::fabs(b1 * p1.x) + std::fabs(b2 * p2.x);
        Float ys = std::fabs(b0 * p0.y) + std::fabs(b1 * p1.y) + std::fabs(b2 * p2.y);
        Float zs = std::fabs(b0 * p0.z) + std::fabs(b1 * p1.z) + std::fabs(b2 * p2.z);
        V3 p_error = V3{xs, ys, zs} * gamma(7);
        V3 p_hit = p0 * b0 + p1 * b1 + p2 * b2;
        P2 uv_hit{uv[0].x * b0 + uv[1].x * b1 + uv[2].x * b2, uv[0].y * b0 + uv[1].y * b1 + uv[2].y * b2};
        // (alpha masks: textures are out of scope)
        V3 surface_normal = normalize(cross(dp02, dp12));
        if (m.flip) surface_normal = -surface_normal;
        V3 sh_n = surface_normal, sh_dpdu = dpdu, sh_dpdv = dpdv, sh_dndu{0, 0, 0}, sh_dndv{0, 0, 0};
        bool has_n = m.has_n && d.N, has_s = m.has_s && d.S;
        if (has_n || has_s) {
            V3 ns;
            if (has_n) {
                ns = N(pr.v[0]) * b0 + N(pr.v[1]) * b1 + N(pr.v[2]) * b2;
                if (length_squared(ns) > 0.0f) ns = normalize(ns); else ns = surface_normal;
            } else ns = surface_normal;
            V3 ss;
            if (has_s) {
                ss = S(pr.v[0]) * b0 + S(pr.v[1]) * b1 + S(pr.v[2]) * b2;
                if (length_squared(ss) > 0.0f) ss = normalize(ss); else ss = normalize(dpdu);
            } else ss = normalize(dpdu);
            V3 ts = cross(ss, ns);
            if (length_squared(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
            else coordinate_system(ns, &ss, &ts);
            // dndu / dndv of the shading geometry (triangle.rs:389-416): consumed by Material::bump
            if (has_n) {
                V3 dn1 = N(pr.v[0]) - N(pr.v[2]), dn2 = N(pr.v[1]) - N(pr.v[2]);
                if (!degenerate_uv) {
                    Float inv_det = 1.0f / determinant;
                    sh_dndu = (dn1 * duv12.y - dn2 * duv02.y) * inv_det;
                    sh_dndv = (dn1 * -duv12.x + dn2 * duv02.x) * inv_det;
                }
            }
            sh_n = normalize(cross(ss, ts));
            surface_normal = faceforward(surface_normal, sh_n);
            sh_dpdu = ss; sh_dpdv = ts;
        }
        *isect = Interaction{};
        isect->p = p_hit; isect->time = ray.time; isect->p_error = p_error;
        isect->wo = -ray.d; // not normalised (Q8)
        isect->n = surface_normal;
        isect->uv = uv_hit; isect->dpdu = dpdu; isect->dpdv = dpdv;
        isect->sh_n = sh_n; isect->sh_dpdu = sh_dpdu; isect->sh_dpdv = sh_dpdv;
        isect->sh_dndu = sh_dndu; isect->sh_dndv = sh_dndv;
        isect->prim = -1;
    }

    // the alpha tests of Triangle::intersect (triangle.rs:313-330: alpha_mask only) and ::intersect_p (:593-655: both masks, and the
    // degenerate-triangle rejection that only exists on that branch): false = the candidate is no hit
    bool alpha_pass(const rspt_prim& pr, const Ray& ray, const Float b[3], bool shadow) const {
        const rspt_mesh& m = d.meshes[pr.mesh];
        if (!m.alpha_tex && !(shadow && m.shadow_alpha_tex)) return true;
        V3 p0 = P(pr.v[0]), p1 = P(pr.v[1]), p2 = P(pr.v[2]);
        P2 uv[3]; get_uvs(pr, uv);
        P2 duv02{uv[0].x - uv[2].x, uv[0].y - uv[2].y}, duv12{uv[1].x - uv[2].x, uv[1].y - uv[2].y};
        V3 dp02 = p0 - p2, dp12 = p1 - p2;
        Float determinant = duv02.x * duv12.y - duv02.y * duv12.x;
        bool degenerate_uv = std::fabs(determinant) < 1e-8f;
        V3 dpdu{0, 0, 0}, dpdv{0, 0, 0};
        if (!degenerate_uv) {
            Float invdet = 1.0f / determinant;
            dpdu = (dp02 * duv12.y - dp12 * duv02.y) * invdet;
            dpdv = (dp02 * -duv12.x + dp12 * duv02.x) * invdet;
        }
        if (degenerate_uv || length_squared(cross(dpdu, dpdv)) == 0.0f) {
            V3 ng = cross(p2 - p0, p1 - p0);
            if (shadow && length_squared(ng) == 0.0f) return false; // triangle.rs:617-621 (intersect_p only)
            coordinate_system(normalize(ng), &dpdu, &dpdv);
        }
        Interaction local; // SurfaceInteraction::new(p_hit, 0, uv_hit, wo, dpdu, dpdv, 0, 0, time): no differentials
        local.p = p0 * b[0] + p1 * b[1] + p2 * b[2];
        local.uv = P2{uv[0].x * b[0] + uv[1].x * b[1] + uv[2].x * b[2], uv[0].y * b[0] + uv[1].y * b[1] + uv[2].y * b[2]};
        local.wo = -ray.d; local.dpdu = dpdu; local.dpdv = dpdv;
        if (m.alpha_tex && alpha_texture_value(*this, m.alpha_tex - 1u, local) == 0.0f) return false;
        if (shadow && m.shadow_alpha_tex && alpha_texture_value(*this, m.shadow_alpha_tex - 1u, local) == 0.0f) return false;
        return true;
    }
    // Triangle::intersect (triangle.rs:134-449)
    bool tri_intersect(const rspt_prim& pr, const Ray& ray, Float* t_hit, Interaction* isect, Float bout[3] = nullptr) const {
        Float b[3], t;
        if (!tri_hit_test(pr, ray, &t, b)) return false;
        if (!alpha_pass(pr, ray, b, false)) return false;
        tri_fill(pr, ray, b, isect);
        *t_hit = t;
        if (bout) { bout[0] = b[0]; bout[1] = b[1]; bout[2] = b[2]; }
        return true;
    }
    // Triangle::area (triangle.rs:667-675)
    Float tri_area(const rspt_prim& pr) const {
        V3 p0 = P(pr.v[0]), p1 = P(pr.v[1]), p2 = P(pr.v[2]);
        return 0.5f * length(cross(p1 - p0, p2 - p0));
    }
    // Triangle::sample (triangle.rs:676-723)
    Interaction tri_sample(const rspt_prim& pr, P2 u, Float* pdf) const {
        const rspt_mesh& m = d.meshes[pr.mesh];
        Float su0 = std::sqrt(u.x);
        Float bx = 1.0f - su0, by = u.y * su0;
        V3 p0 = P(pr.v[0]), p1 = P(pr.v[1]), p2 = P(pr.v[2]);
        Interaction it;
        it.p = p0 * bx + p1 * by + p2 * (1.0f - bx - by);
        V3 n = normalize(cross(p1 - p0, p2 - p0));
        if (m.has_n && d.N) {
            V3 ns = N(pr.v[0]) * bx + N(pr.v[1]) * by + N(pr.v[2]) * (1.0f - bx - by);
            n = faceforward(n, ns);
        } else if (m.flip) n = n * -1.0f;
        V3 pas = vabs(p0 * bx) + vabs(p1 * by) + vabs(p2 * (1.0f - bx - by));
        it.p_error = pas * gamma(6);
        Float area = 0.5f * length(cross(p1 - p0, p2 - p0));
        *pdf = 1.0f / area;
        it.n = n; it.time = 0.0f; it.wo = V3{0, 0, 0};
        return it;
    }
    // Triangle::sample_with_ref_point (triangle.rs:724-744)
    Interaction tri_sample_ref(const rspt_prim& pr, const Interaction& iref, P2 u, Float* pdf) const {
        Interaction intr = tri_sample(pr, u, pdf);
        V3 wi = intr.p - iref.p;
        if (length_squared(wi) == 0.0f) *pdf = 0.0f;
        else {
            wi = normalize(wi);
            *pdf *= distance_squared(iref.p, intr.p) / abs_dot(intr.n, -wi);
            if (std::isinf(*pdf)) *pdf = 0.0f;
        }
        return intr;
    }
    // Triangle::pdf_with_ref_point (triangle.rs:745-764)
    Float tri_pdf_ref(const rspt_prim& pr, const Interaction& iref, V3 wi) const {
        Ray ray = iref.spawn_ray(wi);
        Float t_hit = 0.0f;
        Interaction il;
        if (tri_intersect(pr, ray, &t_hit, &il)) {
            Float pdf = distance_squared(iref.p, il.p) / (abs_dot(il.n, -wi) * tri_area(pr));
            if (std::isinf(pdf)) pdf = 0.0f;
            return pdf;
        }
        return 0.0f;
    }

    // ---- Bounds3f::intersect_p: src/core/geometry.rs:2211-2269 ----
    static bool box_hit(const rspt_bvh_node& nd, const Ray& ray, V3 inv_dir, const uint8_t neg[3]) {
        const Float* lo = nd.bmin; const Float* hi = nd.bmax;
        Float t_min = ((neg[0] ? hi[0] : lo[0]) - ray.o.x) * inv_dir.x;
        Float t_max = ((neg[0] ? lo[0] : hi[0]) - ray.o.x) * inv_dir.x;
        Float ty_min = ((neg[1] ? hi[1] : lo[1]) - ray.o.y) * inv_dir.y;
        Float ty_max = ((neg[1] ? lo[1] : hi[1]) - ray.o.y) * inv_dir.y;
        t_max *= 1.0f + 2.0f * gamma(3);
        ty_max *= 1.0f + 2.0f * gamma(3);
        if (t_min > ty_max || ty_min > t_max) return false;
        if (ty_min > t_min) t_min = ty_min;
        if (ty_max < t_max) t_max = ty_max;
        Float tz_min = ((neg[2] ? hi[2] : lo[2]) - ray.o.z) * inv_dir.z;
        Float tz_max = ((neg[2] ? lo[2] : hi[2]) - ray.o.z) * inv_dir.z;
        tz_max *= 1.0f + 2.0f * gamma(3);
        if (t_min > tz_max || tz_min > t_max) return false;
        if (tz_min > t_min) t_min = tz_min;
        if (tz_max < t_max) t_max = tz_max;
        return (t_min < ray.t_max) && (t_max > 0.0f);
    }

    // isect.primitive -> GeometricPrimitive: a hit whose primitive was dropped (Q11) has no material and no area light
    const rspt_prim& hit_prim(const Interaction& isect) const {
        static const rspt_prim none = {{0, 0, 0}, 0, 0xffffffffu, -1};
        return isect.prim < 0 ? none : d.prims[isect.prim];
    }
    bool instanced() const { return d.n_instances > 0; }
    uint64_t top_nodes() const { return instanced() ? d.n_top_nodes : d.n_nodes; }

    // ---- Transform::transform_surface_interaction: src/core/transform.rs:815-860 (m = to_world, mi = from_world) ----
    static V3 transform_normal(const Float* mi, V3 n) { // :528-537: through the transposed inverse
        return V3{mi[0] * n.x + mi[4] * n.y + mi[8] * n.z, mi[1] * n.x + mi[5] * n.y + mi[9] * n.z, mi[2] * n.x + mi[6] * n.y + mi[10] * n.z};
    }
    static void transform_surface_interaction(const Float* m, const Float* mi, Interaction* si) {
        Interaction ret;
        { // transform_point_with_abs_error :709-760
            Float x = si->p.x, y = si->p.y, z = si->p.z;
            const V3 pe = si->p_error;
            ret.p = V3{m[0] * x + m[1] * y + m[2] * z + m[3], m[4] * x + m[5] * y + m[6] * z + m[7], m[8] * x + m[9] * y + m[10] * z + m[11]};
            ret.p_error.x = (gamma(3) + 1.0f) * (std::fabs(m[0]) * pe.x + std::fabs(m[1]) * pe.y + std::fabs(m[2]) * pe.z)
                            + gamma(3) * (std::fabs(m[0] * x) + std::fabs(m[1] * y) + std::fabs(m[2] * z) + std::fabs(m[3]));
            ret.p_error.y = (gamma(3) + 1.0f) * (std::fabs(m[4]) * pe.x + std::fabs(m[5]) * pe.y + std::fabs(m[6]) * pe.z)
                            + gamma(3) * (std::fabs(m[4] * x) + std::fabs(m[5] * y) + std::fabs(m[6] * z) + std::fabs(m[7]));
            ret.p_error.z = (gamma(3) + 1.0f) * (std::fabs(m[8]) * pe.x + std::fabs(m[9]) * pe.y + std::fabs(m[10]) * pe.z)
                            + gamma(3) * (std::fabs(m[8] * x) + std::fabs(m[9] * y) + std::fabs(m[10] * z) + std::fabs(m[11]));
            const Float wp = m[12] * x + m[13] * y + m[14] * z + m[15];   // :747-760 (the reference asserts wp != 0)
            if (wp != 1.0f) { const Float inv = 1.0f / wp; ret.p = V3{inv * ret.p.x, inv * ret.p.y, inv * ret.p.z}; }
        }
        ret.n = normalize(transform_normal(mi, si->n));
        ret.wo = normalize(transform_vector(m, si->wo));
        ret.time = si->time;
        ret.uv = si->uv;
        ret.dpdu = transform_vector(m, si->dpdu); ret.dpdv = transform_vector(m, si->dpdv);
        ret.sh_n = normalize(transform_normal(mi, si->sh_n));
        ret.sh_dpdu = transform_vector(m, si->sh_dpdu); ret.sh_dpdv = transform_vector(m, si->sh_dpdv);
        ret.sh_dndu = transform_normal(mi, si->sh_dndu); ret.sh_dndv = transform_normal(mi, si->sh_dndv);
        ret.dudx = si->dudx; ret.dvdx = si->dvdx; ret.dudy = si->dudy; ret.dvdy = si->dvdy;
        ret.dpdx = si->dpdx; ret.dpdy = si->dpdy;
        ret.prim = -1; // ret.primitive = None (:856), Q11
        ret.geo_prim = si->geo_prim;
        ret.sh_n = faceforward(ret.sh_n, ret.n);
        *si = ret;
    }
    static bool is_identity(const Float* m) { // transform.rs:291-308
        for (int r = 0; r < 4; r++)
            for (int c = 0; c < 4; c++)
                if (m[4 * r + c] != (r == c ? 1.0f : 0.0f)) return false;
        return true;
    }

    // ---- Primitive::intersect (primitive.rs:37-47) for entry pi of an aggregate's primitive list ----
    bool prim_intersect(uint32_t pi, const Ray& ray, Interaction* isect, Counters* c, Float* t_out, Float* b_out) const {
        const rspt_prim& pr = d.prims[pi];
        if (pr.mesh == RSPT_MESH_INSTANCE) return transformed_intersect(pr.v[0], ray, isect, c, t_out, b_out);
        Float t_hit = 0.0f; // GeometricPrimitive::intersect primitive.rs:150-186
        if (tri_intersect(pr, ray, &t_hit, isect, b_out)) {
            ray.t_max = t_hit; // primitive.rs:155
            isect->prim = pi;  // primitive.rs:42
            isect->geo_prim = pi; isect->inst = 0;
            // primitive.rs:160-170: the primitive's MediumInterface at a medium transition, else the medium the ray travels in
            const rspt_mesh& me = d.meshes[pr.mesh];
            if (me.medium_inside != me.medium_outside) { isect->med_in = me.medium_inside; isect->med_out = me.medium_outside; }
            else isect->med_in = isect->med_out = ray.medium;
            if (t_out) *t_out = t_hit;
            return true;
        }
        return false;
    }
    bool prim_intersect_p(uint32_t pi, const Ray& ray, Counters* c) const {
        const rspt_prim& pr = d.prims[pi];
        if (pr.mesh == RSPT_MESH_INSTANCE) { // TransformedPrimitive::intersect_p primitive.rs:258-265
            const rspt_instance& in = d.instances[pr.v[0]];
            const rspt_object& o = d.objects[in.object];
            Float to_world[16], from_world[16];
            instance_transform(in, ray.time, to_world, from_world);   // primitive_to_world.interpolate(r.time), then its inverse (:259-262)
            Ray r2 = transform_ray(from_world, ray);
            if (!o.n_nodes && c) c->tris_tested++; // (counter convention: every primitive test counts, also the lone primitive of an object)
            return o.n_nodes ? bvh_intersect_p((uint32_t)o.first_node, r2, c) : prim_intersect_p((uint32_t)o.first_prim, r2, c);
        }
        Float t, b[3];
        return tri_hit_test(pr, ray, &t, b) && alpha_pass(pr, ray, b, true);
    }
    // ---- TransformedPrimitive::intersect: primitive.rs:216-253 (static transform: interpolate() returns start_transform) ----
    // AnimatedTransform::interpolate (transform.rs:2081-2113) of a TransformedPrimitive's primitive_to_world at the ray's time: the start
    // Transform for a static instance (and before the interval), the end Transform after it, the product of the interpolated factors inside
    static void instance_transform(const rspt_instance& in, Float time, Float* to_world, Float* from_world) {
        if (!in.animated) { std::memcpy(to_world, in.to_world, 64); std::memcpy(from_world, in.from_world, 64); return; }
        const AnimatedTransform a(in.to_world, in.time[0], in.to_world_end, in.time[1]);
        M44 m, mi;
        a.interpolate_full(time, m44_from(in.from_world), m44_from(in.from_world_end), &m, &mi);
        std::memcpy(to_world, &m.m[0][0], 64); std::memcpy(from_world, &mi.m[0][0], 64);
    }
    bool transformed_intersect(uint32_t k, const Ray& r, Interaction* isect, Counters* c, Float* t_out, Float* b_out) const {
        const rspt_instance& in0 = d.instances[k];
        const rspt_object& o = d.objects[in0.object];
        struct { Float to_world[16], from_world[16]; } in;
        instance_transform(in0, r.time, in.to_world, in.from_world);
        Ray ray = transform_ray(in.from_world, r); // Transform::inverse(&interpolated_prim_to_world).transform_ray(r)
        ray.has_diff = false;                      // (differentials are not used below this point)
        if (!o.n_nodes && c) c->tris_tested++;
        bool hit = o.n_nodes ? bvh_intersect((uint32_t)o.first_node, ray, isect, c, t_out, b_out)
                             : prim_intersect((uint32_t)o.first_prim, ray, isect, c, t_out, b_out);
        if (!hit) return false;
        r.t_max = ray.t_max; // :224
        const bool fixed = d.instancing_mode == RSPT_INSTANCING_FIXED;
        if (!is_identity(in.to_world)) {
            const int64_t keep = isect->prim;
            transform_surface_interaction(in.to_world, in.from_world, isect);
            if (fixed) isect->prim = keep; // the fix commented out at :226-250: "we need to preserve the primitive pointer"
            isect->inst = (int64_t)k + 1;
            return true;
        }
        isect->inst = (int64_t)k + 1;
        return fixed; // Q10: an identity instance has shrunk r.t_max and reports no hit
    }

    // ---- BVHAccel::intersect: src/accelerators/bvh.rs:401-462 over the aggregate whose node 0 is `root` ----
    bool bvh_intersect(uint32_t root, const Ray& ray, Interaction* isect, Counters* c, Float* t_out, Float* b_out) const {
        bool hit = false;
        V3 inv_dir{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
        uint8_t neg[3] = {(uint8_t)(inv_dir.x < 0.0f), (uint8_t)(inv_dir.y < 0.0f), (uint8_t)(inv_dir.z < 0.0f)};
        uint32_t to_visit = 0, cur = root;
        uint32_t stack[64];
        for (;;) {
            const rspt_bvh_node& node = d.nodes[cur];
            if (c) c->nodes_visited++;
            if (box_hit(node, ray, inv_dir, neg)) {
                if (node.n_prims > 0) {
                    for (uint32_t i = 0; i < node.n_prims; i++) {
                        if (c) c->tris_tested++;
                        if (prim_intersect((uint32_t)node.offset + i, ray, isect, c, t_out, b_out)) hit = true;
                    }
                    if (to_visit == 0) break;
                    cur = stack[--to_visit];
                } else {
                    if (neg[node.axis]) { stack[to_visit++] = cur + 1; cur = (uint32_t)node.offset; }
                    else { stack[to_visit++] = (uint32_t)node.offset; cur = cur + 1; }
                }
            } else {
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            }
        }
        return hit;
    }
    // Scene::intersect (scene.rs:55-66)
    bool intersect(const Ray& ray, Interaction* isect, Counters* c, Float* t_out = nullptr, Float* b_out = nullptr) const {
        if (c) c->rays_closest++;
        if (d.n_nodes == 0) return false;
        return bvh_intersect(0, ray, isect, c, t_out, b_out);
    }
    // ---- BVHAccel::intersect_p: bvh.rs:463-514 ----
    bool bvh_intersect_p(uint32_t root, const Ray& ray, Counters* c) const {
        V3 inv_dir{1.0f / ray.d.x, 1.0f / ray.d.y, 1.0f / ray.d.z};
        uint8_t neg[3] = {(uint8_t)(inv_dir.x < 0.0f), (uint8_t)(inv_dir.y < 0.0f), (uint8_t)(inv_dir.z < 0.0f)};
        uint32_t to_visit = 0, cur = root;
        uint32_t stack[64];
        for (;;) {
            const rspt_bvh_node& node = d.nodes[cur];
            if (c) c->nodes_visited++;
            if (box_hit(node, ray, inv_dir, neg)) {
                if (node.n_prims > 0) {
                    for (uint32_t i = 0; i < node.n_prims; i++) {
                        if (c) c->tris_tested++;
                        if (prim_intersect_p((uint32_t)node.offset + i, ray, c)) return true;
                    }
                    if (to_visit == 0) break;
                    cur = stack[--to_visit];
                } else if (neg[node.axis]) { stack[to_visit++] = cur + 1; cur = (uint32_t)node.offset; }
                else { stack[to_visit++] = (uint32_t)node.offset; cur = cur + 1; }
            } else {
                if (to_visit == 0) break;
                cur = stack[--to_visit];
            }
        }
        return false;
    }
    bool intersect_p(const Ray& ray, Counters* c) const {
        if (c) c->rays_any++;
        if (d.n_nodes == 0) return false;
        return bvh_intersect_p(0, ray, c);
    }
    // brute force closest hit over all primitives in list order (test helper, not in the reference)
    bool intersect_brute(const Ray& ray, uint32_t* prim, Float* t_out) const {
        bool hit = false;
        for (uint64_t i = 0; i < d.n_prims; i++) {
            Float t, b[3];
            if (d.prims[i].mesh != RSPT_MESH_INSTANCE && tri_hit_test(d.prims[i], ray, &t, b)) { ray.t_max = t; *prim = (uint32_t)i; *t_out = t; hit = true; }
        }
        return hit;
    }
};

} // namespace orc
