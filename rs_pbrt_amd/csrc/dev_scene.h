// Device view of the flattened scene and the per-hit geometry / light / sampler routines of
// the shade stage.  Layouts are private to librspt (include/rspt.h only fixes the host ABI).
#pragma once
#include "dev_bsdf.h"
#include "material_assembly.h"

namespace rspt {

#define RSPT_MISS 0xffffffffu
// mesh flag bits packed into the 48-byte triangle record
enum : uint32_t { MF_HAS_N = 1, MF_HAS_S = 2, MF_HAS_UV = 4, MF_FLIP = 8,
                  MF_ALPHA = 16,          // the mesh has an alpha_mask or a shadow_alpha_mask (triangle.rs:39-40)
                  MF_INSTANCE = 0x100,
                  MF_MASK_SHIFT = 12 };   // bits 12 .. 31: the mesh's entry in SceneDev::alpha_masks (scenes whose masks all have the simple form)  // the record stands for a TransformedPrimitive: t0.x = instance index, t0.y = the four-box kernel's
                                          // reference to the primitives that follow it in its leaf (RSPT_NONE: it is the last one)

// A float texture used as alpha mask, in the two forms that cover what scene files use ("float imagemap" cut-outs and constants): at the alpha test
// the interaction has no ray differentials (SurfaceInteraction::new), so an ImageTexture lookup is MipMap::triangle at level 0 whatever its filter
// (mipmap.rs:233-262: width 0 -> level < 0; :268-275: minor axis 0) and only the texel's first channel is compared with 0.
struct AlphaMask {
    uint32_t kind;               // 0: no mask, 1: ConstantTexture (value), 2: ImageTexture under a UVMapping2D
    float value;
    float su, sv, du, dv;        // UVMapping2D (texture.rs:64-83)
    uint32_t width, height, wrap, channels;
    uint64_t base;               // level 0's first float in TexTables::texel_pool
};
struct AlphaEntry { AlphaMask alpha, shadow; };   // TriangleMesh.alpha_mask / shadow_alpha_mask (triangle.rs:39-40)

// One ObjectInstance (api.rs:3024-3109): TransformedPrimitive { primitive: the object's aggregate or single primitive,
// primitive_to_world } (primitive.rs:198-211), static transform
struct InstDev {
    float m[12];          // primitive_to_world.m rows 0..2
    float mi[12];         // .m_inv rows 0..2
    uint32_t root_node;   // the object's BVHAccel: index of its node 0 in nodes[]; RSPT_MISS = a single primitive, no aggregate
    uint32_t first_prim;  // the object's first primitive
    uint32_t w4_root;     // k_trace_w4's reference to the object's root record (or leaf reference of a one-leaf aggregate)
    uint32_t identity;    // Transform::is_identity(primitive_to_world) (transform.rs:291-308)
    uint32_t anim;        // a moving instance: index into SceneDev::inst_anim (m / mi / identity above are then the START key's); RSPT_MISS = static
    uint32_t pad[3];
    float m3[4], mi3[4];  // rows 3 of m and m_inv.  A CTM made of Translate / Rotate / Scale has (0 0 0 1) there in both; the inverse that Transform::new computes
                          // for a `Transform [..]` / `ConcatTransform` matrix (Matrix4x4::inverse, Gauss-Jordan) does not — a few 1e-8 — and transform_point
                          // divides by the homogeneous weight whenever it is not exactly 1 (transform.rs:490-516, :661-760): so do inst_ray / inst_point
};

struct SceneDev {
    // LinearBVHNode array as uploaded (bvh.rs:77-85): 2 x float4 per node
    //   n0 = (bmin.xyz, bmax.x)  n1 = (bmax.y, bmax.z, offset:i32, n_prims:u16 | axis:u8 << 16)
    const float4* nodes;
    // 48-byte triangle records in BVH leaf order: 3 x float4
    //   t0 = (p0.xyz, p1.x)  t1 = (p1.yz, p2.xy)  t2 = (p2.z, material:u32, area_light:i32, mesh flags:u32)
    const float4* tris;
    const rspt_prim* prims;  // vertex indices for meshes carrying N / S / UV
    const float* N;
    const float* S;
    const float* UV;
    const rspt_material* materials;
    const rspt_bxdf* bxdfs;
    const rspt_light* lights;
    const struct EnvMapDev* envmaps;    // InfiniteAreaLight maps (rspt_light.prim indexes them)
    const uint32_t* infinite_lights;    // Scene.infinite_lights: indices into lights[] (scene.rs:40-43)
    uint32_t n_nodes, n_prims, n_lights, n_infinite;
    float wb_min[3], wb_max[3];  // BVHAccel::world_bound = nodes[0].bounds (bvh.rs:394-400)
    const uint8_t* mat_flags;    // per material: RSPT_MAT_TEXTURED | RSPT_MAT_BUMP (dev_texture.h); nullptr = no textures in the scene
    const rspt_mesh* meshes;     // per-mesh flags and alpha-mask textures (alpha tests only)
    const InstDev* inst;         // object instances; nullptr = none
    uint32_t n_inst;
    uint32_t inst_fixed;         // RSPT_INSTANCING_FIXED: instanced hits keep their primitive (material)
    const rspt_medium* media;    // RenderOptions.named_media (VolPathIntegrator only; meshes[] carry the medium interfaces)
    uint32_t n_media;
    const rspt_mat::DynMaterial* dyn;  // per material, valid where mat_flags has RSPT_MAT_DYNAMIC (material_assembly.h); nullptr = none
    uint32_t n_grid_media;       // GridDensityMedium records among media[] (their density pointers are device pointers, pad = 1 / max density as float bits)
    // the four-box records of trace_w4.h for the per-lane kernels (trace_serial.h traverse_w4): Wide4Node array, the big-leaf table, the root
    // reference; w4 = nullptr where the scene has none in the plain form (object instances, alpha masks, more records than a reference holds)
    const void* w4;
    const uint2* w4_big;
    uint32_t w4_root;
    // per-primitive copies of the vertex attributes the interaction fill interpolates (round 4): tri_nuv[5 * prim ..] = the three normals (9 floats),
    // the three uvs (6), 5 of padding = 80 bytes next to each other instead of a 24-byte index record and three scattered 12-byte + three 8-byte
    // vertex reads (a dependent round trip and up to seven 64-byte sectors per hit: the shade stage is bound by the bytes it moves).  nullptr = the
    // scene has no per-vertex normals or uvs (or too many primitives for the copy: the gathers remain)
    const float4* tri_nuv;
    // alpha masks in the form k_trace_w4<.., ALPHA = 2> evaluates in line (kernels.h alpha_simple): one entry per distinct (alpha, shadowalpha) pair,
    // addressed by the triangle record's flag bits; nullptr = some mask of the scene is a texture graph only alpha_pass / tex_eval can evaluate
    const struct AlphaEntry* alpha_masks;
    const struct InstAnim* inst_anim;   // the keys of the moving instances (inst_at); nullptr = none
    const float* ray_time;              // [path slot] Ray.time of the path's rays (the camera sample's time, perspective.rs:226), set by rspt_render while a
                                        // scene with moving instances is rendered; nullptr = time 0 (rspt_trace)
    uint32_t time_div;                  // ray slots per entry of ray_time: 1 for path / volpath (a ray's slot is its path's), ao_n_samples for the AO
                                        // integrator's shadow rays (slot = sample * n + k), nodes per camera sample for directlighting's tree; never 0
};

// MipMap<Spectrum> pyramid + Distribution2D of one InfiniteAreaLight (mipmap.rs, sampling.rs:150-198)
struct EnvMapDev {
    const float* texels;        // rgb, levels concatenated
    uint32_t level_offset[16];  // in texels (not floats)
    uint32_t width, height, n_levels, nu, nv;
    const float* cond_func;     // [nv][nu]
    const float* cond_cdf;      // [nv][nu + 1]
    const float* cond_int;      // [nv]
    const float* marg_func;     // [nv]
    const float* marg_cdf;      // [nv + 1]
    float marg_int;
};

// Light sampling distributions (src/core/lightdistrib.rs): one Distribution1D per voxel for
// "spatial", a single one (n_vox = 1) for "uniform" / "power".
struct LightDistDev {
    const float* func;      // [n_vox][n_lights]
    const float* cdf;       // [n_vox][n_lights + 1]
    const float* func_int;  // [n_vox]
    int32_t nvox[3];
    int32_t spatial;
    // on-demand voxels (the reference fills its hash table the first time a voxel is looked up, lightdistrib.rs:297-384): table[voxel] =
    // row of func / cdf / func_int, or < 0 while the voxel has no distribution yet; nullptr = every voxel was built up front, row = voxel
    int32_t* table;
    // Kernels that meet their lookup points only while they run (volpath: a point in a medium is sampled inside the kernel; the pixel samplers:
    // a whole tile is one serial chain) claim a missing voxel themselves — the words k_ld_mark uses — and the host builds the claimed rows and
    // runs the step again (librspt.hip).  nullptr for the kernels whose lookup points are known before the launch (`path`: k_ld_mark).
    struct LightLazyWords* lazy;
    uint32_t* new_list;
};
struct LightLazyWords { uint32_t n_new, n_rows, max_rows, overflow; };   // = LightLazy (kernels.h)
// the voxel's row, or -1 when it has no distribution yet (after claiming it for the next build round where the kernel may do that)
RDEV int32_t light_row_try(const LightDistDev& ld, uint32_t vox) {
    if (!ld.table) return (int32_t)vox;
    const int32_t r = ld.table[vox];
    if (r >= 0) return r;
    if (ld.lazy && r == -1 && atomicCAS(&ld.table[vox], -1, -2) == -1) {
        const uint32_t k = atomicAdd(&ld.lazy->n_new, 1u);
        if (ld.lazy->n_rows + k < ld.lazy->max_rows) ld.new_list[k] = vox;
        else { ld.lazy->overflow = 1u; ld.table[vox] = -1; }
    }
    return -1;
}
RDEV uint32_t light_row(const LightDistDev& ld, uint32_t vox) {
    const int32_t r = light_row_try(ld, vox);
    return r < 0 ? 0u : (uint32_t)r;  // < 0: the row pool ran out (that render fails with RSPT_E_NOMEM) or the step is run again once the row exists; it must not fault
}

// Everything SamplerIntegrator::render reads per sample (subset of rspt_render_desc)
// A moving camera (AnimatedTransform, core/transform.rs:894-2124): what AnimatedTransform::new leaves for interpolate — the end matrix and
// the two decompositions (translation, rotation quaternion xyzw, 4x4 scale matrix) — computed by rspt_render on the host (librspt.hip
// decompose_camera) and read by every lane from device memory.
struct CamAnim {
    float end[16];
    float t[2][3];
    float r[2][4];
    float s[2][16];
    float time[2];   // start_time, end_time
};
struct RenderDev {
    const CamAnim* cam_anim;   // nullptr: camera_to_world holds for every ray (rspt_render_desc.camera_animated = 0, or equal key matrices)
    float raster_to_camera[16], camera_to_world[16];
    float lens_radius, focal_distance, shutter_open, shutter_close;
    int32_t sample_bounds[4], crop_px[4];
    int32_t resolution, log2_res;  // sobol.rs:46-48
    int64_t spp;
    uint32_t max_depth;
    float rr_threshold;
    float filter_radius[2];
    float max_sample_luminance;
    uint32_t tile_size;
    const uint32_t* sobol32;   // [1024*52]
    const uint64_t* vdc;       // [25*52]
    const uint64_t* vdc_inv;   // [26*52]
    const float* filter_table; // [256]
    // HaltonSampler (src/samplers/halton.rs:54-131)
    uint32_t sampler_kind, sample_at_pixel_center;
    int32_t base_scales[2], base_exponents[2];
    uint64_t sample_stride, mult_inverse[2];
    const uint16_t* halton_perms;  // RADICAL_INVERSE_PERMUTATIONS prefix
    const uint32_t* primes;        // PRIMES [1000]
    const uint32_t* prime_sums;    // PRIME_SUMS [1000]
};

// ---- Sobol' sampler (src/samplers/sobol.rs:110-140,190-201; lowdiscrepancy.rs:1014-1076) ----
RDEV uint64_t sobol_interval_to_index(const RenderDev& rd, uint32_t m, uint64_t frame, int32_t px, int32_t py) {
    if (m == 0) return 0;
    uint64_t index = frame << (m << 1);
    uint64_t delta = 0;
    const uint64_t* M = rd.vdc + (m - 1) * 52;
    const uint64_t* MI = rd.vdc_inv + (m - 1) * 52;
    for (uint64_t f = frame; f != 0; f &= f - 1) delta ^= M[__builtin_ctzll(f)];
    uint64_t b = ((uint64_t)((uint32_t)px << m) | (uint64_t)(int64_t)py) ^ delta;
    for (; b != 0; b &= b - 1) index ^= MI[__builtin_ctzll(b)];
    return index;
}
RDEV float sobol_dim(const RenderDev& rd, uint64_t index, uint32_t dim) {
    uint32_t v = 0;
    const uint32_t* row = rd.sobol32 + dim * 52;
    // XOR of the generator-matrix columns selected by the set bits of the index; visiting only the
    // set bits (ctz) instead of shifting through all of them gives the same value (XOR commutes)
    for (uint64_t a = index; a != 0; a &= a - 1) v ^= row[__builtin_ctzll(a)];
    return fminf((float)v * 0x1.0p-32f, RSPT_ONE_MINUS_EPS);
}
// SobolSampler::sample_dimension for dim >= 2 is sobol_dim; dims 0/1 are remapped into the pixel
RDEV float sobol_pixel_dim(const RenderDev& rd, uint64_t index, uint32_t dim, int32_t pix) {
    float s = sobol_dim(rd, index, dim);
    s = s * (float)rd.resolution + (float)rd.sample_bounds[dim];
    return clampf(s - (float)pix, 0.0f, RSPT_ONE_MINUS_EPS);
}
// The shade stage consumes at most 8 consecutive Sobol' dimensions per bounce (light choice 1,
// u_light 2, u_scattering 2, continuation 2, Russian roulette 1).  They are
// evaluated together from an LDS copy of the generator matrices laid out [bit][dimension]: one pass
// over the set bits of the index, 8 independent XOR chains, instead of 8 serial passes through
// L2-resident global memory.  Values are identical to sobol_dim().
struct SobolBlock {
    float v0, v1, v2, v3, v4, v5, v6, v7;   // eight scalars, not an array: selecting between array elements became a selected ADDRESS into a
                                            // scratch copy of the array (two scratch loads per draw in every k_shade instantiation)
    uint32_t base, dim;  // first dimension held, next dimension to hand out
    RDEV void fill(const uint32_t* __restrict__ tab, uint32_t nd, uint64_t index, uint32_t first_dim) {
        uint32_t x0 = 0, x1 = 0, x2 = 0, x3 = 0, x4 = 0, x5 = 0, x6 = 0, x7 = 0;
        for (uint64_t a = index; a != 0; a &= a - 1) {
            const uint32_t* row = tab + (uint32_t)__builtin_ctzll(a) * nd + first_dim;
            x0 ^= row[0]; x1 ^= row[1]; x2 ^= row[2]; x3 ^= row[3]; x4 ^= row[4]; x5 ^= row[5]; x6 ^= row[6]; x7 ^= row[7];
        }
        v0 = fminf((float)x0 * 0x1.0p-32f, RSPT_ONE_MINUS_EPS); v1 = fminf((float)x1 * 0x1.0p-32f, RSPT_ONE_MINUS_EPS);
        v2 = fminf((float)x2 * 0x1.0p-32f, RSPT_ONE_MINUS_EPS); v3 = fminf((float)x3 * 0x1.0p-32f, RSPT_ONE_MINUS_EPS);
        v4 = fminf((float)x4 * 0x1.0p-32f, RSPT_ONE_MINUS_EPS); v5 = fminf((float)x5 * 0x1.0p-32f, RSPT_ONE_MINUS_EPS);
        v6 = fminf((float)x6 * 0x1.0p-32f, RSPT_ONE_MINUS_EPS); v7 = fminf((float)x7 * 0x1.0p-32f, RSPT_ONE_MINUS_EPS);
        base = dim = first_dim;
    }
    RDEV float at(uint32_t k) const {
        float a0 = v0, a1 = v1, a2 = v2, a3 = v3, a4 = v4, a5 = v5, a6 = v6, a7 = v7;
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));   // values, not addresses, go into the selects
        float lo = (k & 1) ? ((k & 2) ? a3 : a1) : ((k & 2) ? a2 : a0);
        float hi = (k & 1) ? ((k & 2) ? a7 : a5) : ((k & 2) ? a6 : a4);
        return (k & 4) ? hi : lo;
    }
    RDEV float get_1d() { return at((dim++) - base); }
    RDEV f2 get_2d() {
        f2 r{at(dim - base), at(dim + 1 - base)};
        dim += 2;
        return r;
    }
};

// ---- Halton sampler (src/samplers/halton.rs:173-226; lowdiscrepancy.rs:788-797, 1082-1122) ----
// digit loops in 32-bit arithmetic whenever the index fits (it does up to ~10^5 spp): 64-bit
// integer division is a long software sequence on the GPU
template <class U>
RDEV float radical_inverse_digits(U a, uint32_t base, const uint16_t* perm, float* inv_base_n_out) {
    float inv_base = 1.0f / (float)base, inv_base_n = 1.0f;
    uint64_t reversed = 0;
    while (a != 0) {
        U next = a / (U)base;
        uint32_t digit = (uint32_t)(a - next * (U)base);
        reversed = reversed * base + (perm ? (uint64_t)perm[digit] : (uint64_t)digit);
        inv_base_n *= inv_base;
        a = next;
    }
    *inv_base_n_out = inv_base_n;
    return (float)reversed;
}
RDEV float radical_inverse_base(uint32_t base, uint64_t a) {  // radical_inverse_specialized
    float ibn;
    float rev = (a >> 32) ? radical_inverse_digits<uint64_t>(a, base, nullptr, &ibn) : radical_inverse_digits<uint32_t>((uint32_t)a, base, nullptr, &ibn);
    return fminf(rev * ibn, RSPT_ONE_MINUS_EPS);
}
RDEV float scrambled_radical_inverse(uint32_t base, uint64_t a, const uint16_t* perm) {  // scrambled_radical_inverse_specialized
    float ibn;
    float rev = (a >> 32) ? radical_inverse_digits<uint64_t>(a, base, perm, &ibn) : radical_inverse_digits<uint32_t>((uint32_t)a, base, perm, &ibn);
    float inv_base = 1.0f / (float)base;
    return fminf(ibn * (rev + inv_base * (float)perm[0] / (1.0f - inv_base)), RSPT_ONE_MINUS_EPS);
}
RDEV uint64_t inverse_radical_inverse(uint64_t base, uint64_t inverse, uint64_t n_digits) {
    uint64_t index = 0;
    for (uint64_t i = 0; i < n_digits; i++) {
        uint64_t digit = inverse % base;
        inverse /= base;
        index = index * base + digit;
    }
    return index;
}
RDEV uint64_t halton_index(const RenderDev& rd, int32_t px, int32_t py, uint64_t sample_num) {  // get_index_for_sample
    uint64_t offset = 0;
    if (rd.sample_stride > 1) {
        int32_t pm0 = px % 128, pm1 = py % 128;  // mod_t(p, K_MAX_RESOLUTION)
        if (pm0 < 0) pm0 += 128;
        if (pm1 < 0) pm1 += 128;
        offset += inverse_radical_inverse(2, (uint64_t)pm0, (uint64_t)rd.base_exponents[0]) * (rd.sample_stride / (uint64_t)rd.base_scales[0]) * rd.mult_inverse[0];
        offset += inverse_radical_inverse(3, (uint64_t)pm1, (uint64_t)rd.base_exponents[1]) * (rd.sample_stride / (uint64_t)rd.base_scales[1]) * rd.mult_inverse[1];
        offset %= rd.sample_stride;
    }
    return offset + sample_num * rd.sample_stride;
}
RDEV float halton_dim(const RenderDev& rd, uint64_t index, uint32_t dim) {  // sample_dimension
    if (rd.sample_at_pixel_center && dim < 2) return 0.5f;
    if (dim == 0) {
        uint64_t a = index >> (uint64_t)rd.base_exponents[0];
        uint64_t r = ((uint64_t)__brev((uint32_t)a) << 32) | (uint64_t)__brev((uint32_t)(a >> 32));  // reverse_bits_64
        return (float)r * 0x1.0p-64f;
    }
    if (dim == 1) return radical_inverse_base(3, index / (uint64_t)rd.base_scales[1]);
    return scrambled_radical_inverse(rd.primes[dim], index, rd.halton_perms + rd.prime_sums[dim]);
}

// the sampler cursor of the shade stage: Sobol' dimensions come from the LDS block, Halton's are
// computed on demand (both are pure functions of (index, dimension): GlobalSampler, sampler.rs)
template <bool HALTON_POSSIBLE = true, bool SOBOL_POSSIBLE = true>
struct PathSamplerT {
    SobolBlock blk;
    uint64_t index;
    uint32_t hdim;
    bool halton;
    RDEV void start(const RenderDev& rd, const uint32_t* __restrict__ tab, uint32_t nd, uint64_t idx, uint32_t first_dim) {
        halton = HALTON_POSSIBLE && (!SOBOL_POSSIBLE || rd.sampler_kind == RSPT_SAMPLER_HALTON);
        index = idx;
        hdim = first_dim;
        if (!halton) blk.fill(tab, nd, idx, first_dim);
    }
    RDEV uint32_t dim() const { return (HALTON_POSSIBLE && halton) ? hdim : blk.dim; }
    RDEV float get_1d(const RenderDev& rd) { return (HALTON_POSSIBLE && halton) ? halton_dim(rd, index, hdim++) : blk.get_1d(); }
    RDEV f2 get_2d(const RenderDev& rd) {
        if (!(HALTON_POSSIBLE && halton)) return blk.get_2d();
        float y = halton_dim(rd, index, hdim + 1), x = halton_dim(rd, index, hdim);
        hdim += 2;
        return f2{x, y};
    }
};
using PathSampler = PathSamplerT<true>;

// ---- geometry at a hit: second half of Triangle::intersect (triangle.rs:274-448) ----
struct Hit {
    f3 p, p_err, n;        // point, conservative error bound, geometric normal (oriented)
    f3 sh_n, sh_dpdu;      // shading frame inputs of Bsdf::new
    uint32_t material;
    int32_t area_light;
};

struct TriRec {
    f3 p0, p1, p2;
    uint32_t material;
    int32_t area_light;
    uint32_t flags;
};
RDEV TriRec load_tri(const SceneDev& sc, uint32_t prim) {
    float4 a = sc.tris[3 * (size_t)prim], b = sc.tris[3 * (size_t)prim + 1], c = sc.tris[3 * (size_t)prim + 2];
    TriRec t;
    t.p0 = f3{a.x, a.y, a.z};
    t.p1 = f3{a.w, b.x, b.y};
    t.p2 = f3{b.z, b.w, c.x};
    t.material = __float_as_uint(c.y);
    t.area_light = (int32_t)__float_as_uint(c.z);
    t.flags = __float_as_uint(c.w);
    return t;
}
RDEV f3 ld3(const float* a, uint32_t i) { return f3{a[3 * (size_t)i], a[3 * (size_t)i + 1], a[3 * (size_t)i + 2]}; }

// VERTEX = false: the scene has no mesh with per-vertex normals / tangents / uvs (SF_VERTEX, dev_bsdf.h)
template <bool VERTEX = true>
RDEVN void tri_fill(const SceneDev& sc, uint32_t prim, const TriRec& t, float b0, float b1, float b2, Hit* h) {
    f3 p0 = t.p0, p1 = t.p1, p2 = t.p2;
    f2 uv0{0.0f, 0.0f}, uv1{1.0f, 0.0f}, uv2{1.0f, 1.0f};  // triangle.rs:97-112
    const bool has_uv = VERTEX && (t.flags & MF_HAS_UV) && sc.UV;
    const bool has_n = VERTEX && (t.flags & MF_HAS_N) && sc.N, has_s = VERTEX && (t.flags & MF_HAS_S) && sc.S;
    uint32_t v0 = 0, v1 = 0, v2 = 0;
    f3 vn0{0.0f, 0.0f, 0.0f}, vn1 = vn0, vn2 = vn0;
    const bool packed = sc.tri_nuv != nullptr && (has_uv || has_n);   // the primitive's own copy of its normals / uvs (SceneDev::tri_nuv)
    if (packed) {
        const float4* q = sc.tri_nuv + 5 * (size_t)prim;
        if (has_n) { const float4 a = q[0], b = q[1]; const float c = q[2].x; vn0 = f3{a.x, a.y, a.z}; vn1 = f3{a.w, b.x, b.y}; vn2 = f3{b.z, b.w, c}; }
        if (has_uv) { const float4 c = q[2], d = q[3]; uv0 = f2{c.y, c.z}; uv1 = f2{c.w, d.x}; uv2 = f2{d.y, d.z}; }
    }
    if (((has_uv || has_n) && !packed) || has_s) {
        rspt_prim pr = sc.prims[prim];
        v0 = pr.v[0]; v1 = pr.v[1]; v2 = pr.v[2];
    }
    if (has_uv && !packed) {
        uv0 = f2{sc.UV[2 * (size_t)v0], sc.UV[2 * (size_t)v0 + 1]};
        uv1 = f2{sc.UV[2 * (size_t)v1], sc.UV[2 * (size_t)v1 + 1]};
        uv2 = f2{sc.UV[2 * (size_t)v2], sc.UV[2 * (size_t)v2 + 1]};
    }
    if (has_n && !packed) { vn0 = ld3(sc.N, v0); vn1 = ld3(sc.N, v1); vn2 = ld3(sc.N, v2); }
    f2 duv02{uv0.x - uv2.x, uv0.y - uv2.y}, duv12{uv1.x - uv2.x, uv1.y - uv2.y};
    f3 dp02 = p0 - p2, dp12 = p1 - p2;
    float det = duv02.x * duv12.y - duv02.y * duv12.x;
    bool degenerate = fabsf(det) < 1e-8f;
    f3 dpdu{0.0f, 0.0f, 0.0f}, dpdv{0.0f, 0.0f, 0.0f};
    if (!degenerate) {
        float invdet = 1.0f / det;
        dpdu = (dp02 * duv12.y - dp12 * duv02.y) * invdet;
        dpdv = (dp02 * -duv12.x + dp12 * duv02.x) * invdet;
    }
    if (degenerate || len2(cross(dpdu, dpdv)) == 0.0f) coordinate_system(normalize(cross(p2 - p0, p1 - p0)), &dpdu, &dpdv);
    float xs = fabsf(b0 * p0.x) + fabsf(b1 * p1.x) + fabsf(b2 * p2.x);
    float ys = fabsf(b0 * p0.y) + fabsf(b1 * p1.y) + fabsf(b2 * p2.y);
    float zs = fabsf(b0 * p0.z) + fabsf(b1 * p1.z) + fabsf(b2 * p2.z);
    h->p_err = f3{xs, ys, zs} * gamma_n(7);
    h->p = p0 * b0 + p1 * b1 + p2 * b2;
    f3 n = normalize(cross(dp02, dp12));
    if (t.flags & MF_FLIP) n = -n;
    f3 sh_n = n, sh_dpdu = dpdu;
    if (has_n || has_s) {
        f3 ns = n;
        if (has_n) {
            ns = vn0 * b0 + vn1 * b1 + vn2 * b2;
            ns = len2(ns) > 0.0f ? normalize(ns) : n;
        }
        f3 ss;
        if (has_s) {
            ss = ld3(sc.S, v0) * b0 + ld3(sc.S, v1) * b1 + ld3(sc.S, v2) * b2;
            ss = len2(ss) > 0.0f ? normalize(ss) : normalize(dpdu);
        } else
            ss = normalize(dpdu);
        f3 ts = cross(ss, ns);
        if (len2(ts) > 0.0f) { ts = normalize(ts); ss = cross(ts, ns); }
        else coordinate_system(ns, &ss, &ts);
        sh_n = normalize(cross(ss, ts));  // SurfaceInteraction::set_shading_geometry
        n = faceforward(n, sh_n);
        sh_dpdu = ss;
    }
    h->n = n; h->sh_n = sh_n; h->sh_dpdu = sh_dpdu;
    h->material = t.material; h->area_light = t.area_light;
}

// ---- TransformedPrimitive (primitive.rs:198-272) ----------------------------------------------------------------
// Transform::transform_vector / transform_normal with a 3x4 matrix (transform.rs:518-537); the normal goes through the
// transposed inverse
RDEV f3 xf_normal(const float* mi, f3 n) {
    return f3{mi[0] * n.x + mi[4] * n.y + mi[8] * n.z, mi[1] * n.x + mi[5] * n.y + mi[9] * n.z, mi[2] * n.x + mi[6] * n.y + mi[10] * n.z};
}
// Transform::inverse(primitive_to_world).transform_ray(r) (transform.rs:538-595 with transform_point_with_error :661-704):
// origin and direction through m_inv, the origin pushed along d to the edge of its error bound, t_max shortened by the same dt
RDEV void xf_ray(const float* m /* rows 0..2, row major */, const float* m3 /* row 3; nullptr = (0 0 0 1) */, f3 o, f3 d, float t_max, f3* oo, f3* od, float* ot) {
    const float x = o.x, y = o.y, z = o.z;
    f3 op{m[0] * x + m[1] * y + m[2] * z + m[3], m[4] * x + m[5] * y + m[6] * z + m[7], m[8] * x + m[9] * y + m[10] * z + m[11]};
    if (m3) {   // transform_point_with_error's homogeneous divide (transform.rs:694-707)
        const float wp = m3[0] * x + m3[1] * y + m3[2] * z + m3[3];
        if (wp != 1.0f) { const float inv = 1.0f / wp; op = f3{inv * op.x, inv * op.y, inv * op.z}; }
    }
    const f3 o_err = f3{fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3]), fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7]),
                        fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11])} * gamma_n(3);
    const f3 dd = xf_vector(m, d);
    const float l2 = dd.x * dd.x + dd.y * dd.y + dd.z * dd.z;
    if (l2 > 0.0f) {
        const f3 a = vabs(dd);
        const float dt = (a.x * o_err.x + a.y * o_err.y + a.z * o_err.z) / l2;
        op = op + dd * dt;
        t_max -= dt;
    }
    *oo = op; *od = dd; *ot = t_max;
}
RDEV void inst_ray(const InstDev& in, f3 o, f3 d, float t_max, f3* oo, f3* od, float* ot) { xf_ray(in.mi, in.mi3, o, d, t_max, oo, od, ot); }
// Transform::transform_surface_interaction (transform.rs:815-860) on the fields the path needs: p with
// transform_point_with_abs_error (:709-760), n / shading.n normalised, shading.n face-forwarded to n, shading.dpdu as a vector
RDEV void inst_point(const float* m, const float* m3 /* row 3 */, f3 p, f3 pe, f3* po, f3* peo) {
    const float x = p.x, y = p.y, z = p.z, g3 = gamma_n(3);
    *po = f3{m[0] * x + m[1] * y + m[2] * z + m[3], m[4] * x + m[5] * y + m[6] * z + m[7], m[8] * x + m[9] * y + m[10] * z + m[11]};
    const float wp = m3[0] * x + m3[1] * y + m3[2] * z + m3[3];   // transform_point_with_abs_error's homogeneous divide (transform.rs:748-760)
    if (wp != 1.0f) { const float inv = 1.0f / wp; *po = f3{inv * po->x, inv * po->y, inv * po->z}; }
    *peo = f3{(g3 + 1.0f) * (fabsf(m[0]) * pe.x + fabsf(m[1]) * pe.y + fabsf(m[2]) * pe.z) + g3 * (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3])),
              (g3 + 1.0f) * (fabsf(m[4]) * pe.x + fabsf(m[5]) * pe.y + fabsf(m[6]) * pe.z) + g3 * (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7])),
              (g3 + 1.0f) * (fabsf(m[8]) * pe.x + fabsf(m[9]) * pe.y + fabsf(m[10]) * pe.z) + g3 * (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11]))};
}
RDEV void inst_hit(const InstDev& in, Hit* h) {
    f3 p, pe;
    inst_point(in.m, in.m3, h->p, h->p_err, &p, &pe);
    h->p = p; h->p_err = pe;
    h->n = normalize(xf_normal(in.mi, h->n));
    f3 sn = normalize(xf_normal(in.mi, h->sh_n));
    h->sh_dpdu = xf_vector(in.m, h->sh_dpdu);
    h->sh_n = dot(sn, h->n) < 0.0f ? -sn : sn;  // nrm_faceforward_nrm (geometry.rs:1852-1858)
}

// Watertight ray/triangle test shared by Triangle::intersect and ::intersect_p
// (triangle.rs:134-273, 450-579).  The ray-dependent permutation/shear is precomputed once
// per ray (it does not depend on the triangle), which leaves every value bit-identical.
struct RayShear {
    int kx, ky, kz;
    float sx, sy, sz;
};
RDEV RayShear ray_shear(f3 d) {
    f3 a = vabs(d);
    RayShear s;
    s.kz = a.x > a.y ? (a.x > a.z ? 0 : 2) : (a.y > a.z ? 1 : 2);  // max_dimension geometry.rs:721-733
    s.kx = s.kz + 1; if (s.kx == 3) s.kx = 0;
    s.ky = s.kx + 1; if (s.ky == 3) s.ky = 0;
    float dx = comp(d, s.kx), dy = comp(d, s.ky), dz = comp(d, s.kz);
    s.sx = -dx / dz; s.sy = -dy / dz; s.sz = 1.0f / dz;
    return s;
}
RDEV f3 permute(f3 v, int kx, int ky, int kz) { return f3{comp(v, kx), comp(v, ky), comp(v, kz)}; }

RDEV bool tri_test(f3 p0, f3 p1, f3 p2, f3 o, const RayShear& rs, float t_max, float* t_out, float* b0o, float* b1o, float* b2o) {
    f3 p0t = permute(p0 - o, rs.kx, rs.ky, rs.kz), p1t = permute(p1 - o, rs.kx, rs.ky, rs.kz), p2t = permute(p2 - o, rs.kx, rs.ky, rs.kz);
    p0t.x += rs.sx * p0t.z; p0t.y += rs.sy * p0t.z;
    p1t.x += rs.sx * p1t.z; p1t.y += rs.sy * p1t.z;
    p2t.x += rs.sx * p2t.z; p2t.y += rs.sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {  // f64 fallback, triangle.rs:189-200
        e0 = (float)((double)p2t.y * (double)p1t.x - (double)p2t.x * (double)p1t.y);
        e1 = (float)((double)p0t.y * (double)p2t.x - (double)p0t.x * (double)p2t.y);
        e2 = (float)((double)p1t.y * (double)p0t.x - (double)p1t.x * (double)p0t.y);
    }
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
    float det = e0 + e1 + e2;
    if (det == 0.0f) return false;
    p0t.z *= rs.sz; p1t.z *= rs.sz; p2t.z *= rs.sz;
    float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if ((det < 0.0f && (t_scaled >= 0.0f || t_scaled < t_max * det)) || (det > 0.0f && (t_scaled <= 0.0f || t_scaled > t_max * det))) return false;
    float inv_det = 1.0f / det;
    float b0 = e0 * inv_det, b1 = e1 * inv_det, b2 = e2 * inv_det;
    float t = t_scaled * inv_det;
    float max_zt = max3(fabsf(p0t.z), fabsf(p1t.z), fabsf(p2t.z));
    float delta_z = gamma_n(3) * max_zt;
    float max_xt = max3(fabsf(p0t.x), fabsf(p1t.x), fabsf(p2t.x));
    float max_yt = max3(fabsf(p0t.y), fabsf(p1t.y), fabsf(p2t.y));
    float delta_x = gamma_n(5) * (max_xt + max_zt);
    float delta_y = gamma_n(5) * (max_yt + max_zt);
    float delta_e = 2.0f * (gamma_n(2) * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
    float max_e = max3(fabsf(e0), fabsf(e1), fabsf(e2));
    float delta_t = 3.0f * (gamma_n(3) * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * fabsf(inv_det);
    if (t <= delta_t) return false;
    *t_out = t; *b0o = b0; *b1o = b1; *b2o = b2;
    return true;
}

// ---- DiffuseAreaLight on one triangle (src/lights/diffuse.rs, triangle.rs:667-764) ----
RDEV rgb light_l(const rspt_light& lt, f3 n, f3 w) {  // diffuse.rs:164-170
    if (lt.two_sided || dot(n, w) > 0.0f) return ldrgb(lt.L);
    return mkrgb(0.0f);
}
RDEV float tri_area(const TriRec& t) { return 0.5f * len(cross(t.p1 - t.p0, t.p2 - t.p0)); }

struct LightSample {
    f3 p, p_err, n;
};
// Triangle::sample + sample_with_ref_point (triangle.rs:676-744)
RDEV LightSample tri_sample_ref(const SceneDev& sc, uint32_t prim, const TriRec& t, f3 ref_p, f2 u, float* pdf) {
    float su0 = sqrtf(u.x);
    float bx = 1.0f - su0, by = u.y * su0;
    float bz = 1.0f - bx - by;
    LightSample s;
    s.p = t.p0 * bx + t.p1 * by + t.p2 * bz;
    f3 n = normalize(cross(t.p1 - t.p0, t.p2 - t.p0));
    if ((t.flags & MF_HAS_N) && sc.N) {
        rspt_prim pr = sc.prims[prim];
        f3 ns = ld3(sc.N, pr.v[0]) * bx + ld3(sc.N, pr.v[1]) * by + ld3(sc.N, pr.v[2]) * bz;
        n = faceforward(n, ns);
    } else if (t.flags & MF_FLIP)
        n = n * -1.0f;
    s.n = n;
    s.p_err = (vabs(t.p0 * bx) + vabs(t.p1 * by) + vabs(t.p2 * bz)) * gamma_n(6);
    *pdf = 1.0f / tri_area(t);
    f3 wi = s.p - ref_p;
    if (len2(wi) == 0.0f) *pdf = 0.0f;
    else {
        wi = normalize(wi);
        *pdf *= dist2(ref_p, s.p) / absdot(s.n, -wi);
        if (__builtin_isinf(*pdf)) *pdf = 0.0f;
    }
    return s;
}
template <uint32_t F = 0xffffffffu>
RDEV bool light_is_delta(const rspt_light& lt) {  // light.rs:178-188
    return ((F & SF_L_POINT) && lt.kind == RSPT_LIGHT_POINT) || ((F & SF_L_SPOT) && lt.kind == RSPT_LIGHT_SPOT) || ((F & SF_L_DISTANT) && lt.kind == RSPT_LIGHT_DISTANT);
}
// ---- MipMap<Spectrum> lookups, wrap mode Repeat (mipmap.rs:206-252, 323-336) ----
RDEV rgb env_texel(const EnvMapDev& m, uint32_t level, int64_t s_, int64_t t_) {
    uint32_t w = m.width >> level, h = m.height >> level;
    w = w ? w : 1u; h = h ? h : 1u;
    uint64_t ss = (uint64_t)s_ % (uint64_t)w, tt = (uint64_t)t_ % (uint64_t)h;  // mod_t(s as usize, u_size)
    return ldrgb(m.texels + 3 * ((size_t)m.level_offset[level] + tt * w + ss));
}
RDEV rgb env_triangle(const EnvMapDev& m, uint32_t level, f2 st) {
    if (level > m.n_levels - 1) level = m.n_levels - 1;
    uint32_t w = m.width >> level, h = m.height >> level;
    w = w ? w : 1u; h = h ? h : 1u;
    float s = st.x * (float)w - 0.5f, t = st.y * (float)h - 0.5f;
    int64_t s0 = (int64_t)floorf(s), t0 = (int64_t)floorf(t);
    float ds = s - (float)s0, dt = t - (float)t0;
    rgb tmp1 = env_texel(m, level, s0 + 1, t0 + 1) * (ds * dt);
    rgb tmp2 = env_texel(m, level, s0 + 1, t0) * (ds * (1.0f - dt));
    rgb tmp3 = env_texel(m, level, s0, t0 + 1) * ((1.0f - ds) * dt);
    rgb tmp4 = env_texel(m, level, s0, t0) * ((1.0f - ds) * (1.0f - dt));
    return tmp4 + tmp3 + tmp2 + tmp1;
}
RDEV rgb env_lookup(const EnvMapDev& m, f2 st, float width) {  // lookup_pnt_flt
    float level = (float)m.n_levels - 1.0f + rspt_log2f(fmaxf(width, 1e-8f));
    if (level < 0.0f) return env_triangle(m, 0, st);
    if (level >= (float)m.n_levels - 1.0f) return env_texel(m, m.n_levels - 1, 0, 0);
    uint32_t il = (uint32_t)floorf(level);
    float delta = level - (float)il;
    rgb a = env_triangle(m, il, st), b = env_triangle(m, il + 1, st);
    return a * (1.0f - delta) + b * delta;
}
// Distribution1D::sample_continuous (sampling.rs:53-101)
RDEV float dist1d_sample_continuous(const float* func, const float* cdf, float func_int, uint32_t n, float u, float* pdf, uint32_t* off) {
    uint32_t first = 0, length = n + 1;
    while (length > 0) {
        uint32_t half = length >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; length -= half + 1; }
        else length = half;
    }
    int64_t o = (int64_t)first - 1;
    o = o < 0 ? 0 : (o > (int64_t)n - 1 ? (int64_t)n - 1 : o);
    *off = (uint32_t)o;
    float du = u - cdf[o];
    if ((cdf[o + 1] - cdf[o]) > 0.0f) du /= cdf[o + 1] - cdf[o];
    *pdf = func_int > 0.0f ? func[o] / func_int : 0.0f;
    return ((float)o + du) / (float)n;
}
RDEV f2 env_sample_continuous(const EnvMapDev& m, f2 u, float* pdf) {  // Distribution2D::sample_continuous
    float pdf0, pdf1;
    uint32_t v, dummy;
    float d1 = dist1d_sample_continuous(m.marg_func, m.marg_cdf, m.marg_int, m.nv, u.y, &pdf1, &v);
    float d0 = dist1d_sample_continuous(m.cond_func + (size_t)v * m.nu, m.cond_cdf + (size_t)v * (m.nu + 1), m.cond_int[v], m.nu, u.x, &pdf0, &dummy);
    *pdf = pdf0 * pdf1;
    return f2{d0, d1};
}
RDEV uint32_t f2u_sat(float x) { return (x != x || x <= 0.0f) ? 0u : (x >= 4294967296.0f ? 0xffffffffu : (uint32_t)x); }
RDEV float env_pdf(const EnvMapDev& m, f2 p) {  // Distribution2D::pdf
    uint32_t iu = min(f2u_sat(p.x * (float)m.nu), m.nu - 1), iv = min(f2u_sat(p.y * (float)m.nv), m.nv - 1);
    return m.cond_func[(size_t)iv * m.nu + iu] / m.marg_int;
}
RDEV f3 mat3_mul(const float* m, f3 w) { return f3{m[0] * w.x + m[1] * w.y + m[2] * w.z, m[3] * w.x + m[4] * w.y + m[5] * w.z, m[6] * w.x + m[7] * w.y + m[8] * w.z}; }
RDEV float spherical_theta(f3 v) { return rspt_acosf(clampf(v.z, -1.0f, 1.0f)); }  // geometry.rs:1584-1586
RDEV float spherical_phi(f3 v) { float p = rspt_atan2f(v.y, v.x); return p < 0.0f ? p + 2.0f * RSPT_PI : p; }
#define RSPT_INV_2_PI 0.15915494309189533577f
// InfiniteAreaLight::le / pdf_li (infinite.rs:369-392)
RDEVN rgb infinite_le(const SceneDev& sc, const rspt_light& lt, f3 ray_d) {
    f3 w = normalize(mat3_mul(lt.p + 9, ray_d));
    f2 st{spherical_phi(w) * RSPT_INV_2_PI, spherical_theta(w) * RSPT_INV_PI};
    return env_lookup(sc.envmaps[lt.prim], st, 0.0f);
}
RDEVN float infinite_pdf_li(const SceneDev& sc, const rspt_light& lt, f3 w) {
    f3 wi = mat3_mul(lt.p + 9, w);
    float theta = spherical_theta(wi), phi = spherical_phi(wi);
    float sin_theta = rspt_sinf(theta);
    if (sin_theta == 0.0f) return 0.0f;
    return env_pdf(sc.envmaps[lt.prim], f2{phi * RSPT_INV_2_PI, theta * RSPT_INV_PI}) / (2.0f * RSPT_PI * RSPT_PI * sin_theta);
}
// Bounds3f::bounding_sphere of the scene bound (geometry.rs:2160-2172), DistantLight::preprocess
RDEV float world_radius(const SceneDev& sc) {
    f3 lo{sc.wb_min[0], sc.wb_min[1], sc.wb_min[2]}, hi{sc.wb_max[0], sc.wb_max[1], sc.wb_max[2]};
    f3 c = vdiv(lo + hi, 2.0f);
    bool inside = c.x >= lo.x && c.x <= hi.x && c.y >= lo.y && c.y <= hi.y && c.z >= lo.z && c.z <= hi.z;
    return inside ? sqrtf(dist2(c, hi)) : 0.0f;
}
RDEV float spot_falloff(const rspt_light& lt, f3 w) {  // spot.rs:67-80
    const float* m = lt.p + 3;
    f3 wl = normalize(f3{m[0] * w.x + m[1] * w.y + m[2] * w.z, m[3] * w.x + m[4] * w.y + m[5] * w.z, m[6] * w.x + m[7] * w.y + m[8] * w.z});
    float c = wl.z, c_total = lt.p[12], c_start = lt.p[13];
    if (c < c_total) return 0.0f;
    if (c >= c_start) return 1.0f;
    float delta = (c - c_total) / (c_start - c_total);
    return (delta * delta) * (delta * delta);
}
// Light::sample_li: DiffuseAreaLight (diffuse.rs:64-84), PointLight (point.rs:52-68), SpotLight
// (spot.rs:81-106), DistantLight (distant.rs:41-58).  Delta lights return pdf = 1 and a light point
// with zero normal and zero error bounds (InteractionCommon::default()).
// pre: the light's triangle record where the caller already holds it (direct.h keeps the lights and their triangles in LDS), else nullptr
template <uint32_t F = 0xffffffffu>
RDEV rgb light_sample_li(const SceneDev& sc, const rspt_light& lt, f3 ref_p, f2 u, f3* wi, float* pdf, LightSample* ls, const TriRec* pre = nullptr) {
    if (!(F & (SF_L_POINT | SF_L_SPOT | SF_L_DISTANT | SF_L_INFINITE)) || lt.kind == RSPT_LIGHT_DIFFUSE_AREA) {
        TriRec t = pre ? *pre : load_tri(sc, lt.prim);
        *ls = tri_sample_ref(sc, lt.prim, t, ref_p, u, pdf);
        if (*pdf == 0.0f || len2(ls->p - ref_p) == 0.0f) { *pdf = 0.0f; return mkrgb(0.0f); }
        *wi = normalize(ls->p - ref_p);
        return light_l(lt, ls->n, -*wi);
    }
    ls->p_err = f3{0.0f, 0.0f, 0.0f};
    ls->n = f3{0.0f, 0.0f, 0.0f};
    if ((F & SF_L_INFINITE) && lt.kind == RSPT_LIGHT_INFINITE) {  // infinite.rs:298-341
        const EnvMapDev& m = sc.envmaps[lt.prim];
        float map_pdf = 0.0f;
        f2 uv = env_sample_continuous(m, u, &map_pdf);
        if (map_pdf == 0.0f) { *pdf = 0.0f; return mkrgb(0.0f); }
        float theta = uv.y * RSPT_PI, phi = uv.x * 2.0f * RSPT_PI;
        float cos_theta = rspt_cosf(theta), sin_theta = rspt_sinf(theta), sin_phi = rspt_sinf(phi), cos_phi = rspt_cosf(phi);
        *wi = mat3_mul(lt.p, f3{sin_theta * cos_phi, sin_theta * sin_phi, cos_theta});
        *pdf = map_pdf / (2.0f * RSPT_PI * RSPT_PI * sin_theta);
        if (sin_theta == 0.0f) *pdf = 0.0f;
        ls->p = ref_p + *wi * (2.0f * world_radius(sc));
        return env_lookup(m, uv, 0.0f);
    }
    *pdf = 1.0f;
    if ((F & SF_L_DISTANT) && lt.kind == RSPT_LIGHT_DISTANT) {
        f3 w{lt.p[0], lt.p[1], lt.p[2]};
        *wi = w;
        ls->p = ref_p + w * (2.0f * world_radius(sc));
        return ldrgb(lt.L);
    }
    f3 pl{lt.p[0], lt.p[1], lt.p[2]};
    *wi = normalize(pl - ref_p);
    ls->p = pl;
    float d2 = dist2(pl, ref_p);
    if (!(F & SF_L_SPOT) || lt.kind == RSPT_LIGHT_POINT) return ldrgb(lt.L) / d2;
    return ldrgb(lt.L) * spot_falloff(lt, -*wi) / d2;
}
// Light::power (diffuse.rs:85-93, point.rs:69-71, spot.rs:107-113, distant.rs:59-62)
RDEV rgb light_power(const SceneDev& sc, const rspt_light& lt) {
    if (lt.kind == RSPT_LIGHT_POINT) return ldrgb(lt.L) * (4.0f * RSPT_PI);
    if (lt.kind == RSPT_LIGHT_SPOT) return ldrgb(lt.L) * 2.0f * RSPT_PI * (1.0f - 0.5f * (lt.p[13] + lt.p[12]));
    if (lt.kind == RSPT_LIGHT_DISTANT) { float r = world_radius(sc); return ldrgb(lt.L) * RSPT_PI * r * r; }
    if (lt.kind == RSPT_LIGHT_INFINITE) { float r = world_radius(sc); return env_lookup(sc.envmaps[lt.prim], f2{0.5f, 0.5f}, 0.5f) * mkrgb(RSPT_PI * r * r); }  // infinite.rs:342-346
    TriRec t = load_tri(sc, lt.prim);
    float factor = lt.two_sided ? 2.0f : 1.0f;
    return ldrgb(lt.L) * factor * tri_area(t) * RSPT_PI;
}

// ---- Distribution1D::sample_discrete (sampling.rs:103-142) over a device-resident cdf ----
RDEV uint32_t sample_discrete(const float* func, const float* cdf, float func_int, uint32_t n, float u, float* pdf) {
    uint32_t first = 0, length = n + 1;
    while (length > 0) {
        uint32_t half = length >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; length -= half + 1; }
        else length = half;
    }
    int64_t off = (int64_t)first - 1;
    off = off < 0 ? 0 : (off > (int64_t)n - 1 ? (int64_t)n - 1 : off);
    *pdf = func_int > 0.0f ? func[off] / (func_int * (float)n) : 0.0f;
    return (uint32_t)off;
}
// SpatialLightDistribution::lookup voxel addressing (lightdistrib.rs:276-295)
RDEV uint32_t light_voxel(const SceneDev& sc, const LightDistDev& ld, f3 p) {
    if (!ld.spatial) return 0;
    float o[3] = {p.x - sc.wb_min[0], p.y - sc.wb_min[1], p.z - sc.wb_min[2]};  // Bounds3::offset geometry.rs:2160-2173
    int32_t pi[3];
    for (int i = 0; i < 3; i++) {
        if (sc.wb_max[i] > sc.wb_min[i]) o[i] /= sc.wb_max[i] - sc.wb_min[i];
        int32_t v = f2i_sat(o[i] * (float)ld.nvox[i]);
        pi[i] = v < 0 ? 0 : (v > ld.nvox[i] - 1 ? ld.nvox[i] - 1 : v);
    }
    return (uint32_t)(((int64_t)pi[2] * ld.nvox[1] + pi[1]) * ld.nvox[0] + pi[0]);
}

// ---- the camera_to_world matrix of a ray: AnimatedTransform::transform_ray / interpolate (transform.rs:2081-2124) at
// ray.time = lerp(CameraSample.time, shutter_open, shutter_close) (perspective.rs:226) ----
RDEV void mat4_mul(const float* a, const float* b, float* r) {  // mtx_mul (transform.rs:238-249): all four products of every element, in this order
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) r[4 * i + j] = a[4 * i] * b[j] + a[4 * i + 1] * b[4 + j] + a[4 * i + 2] * b[8 + j] + a[4 * i + 3] * b[12 + j];
}
// the factors AnimatedTransform::interpolate multiplies for a time inside the interval (transform.rs:2091-2105), dt in (0, 1): translate(trans).m,
// rotate.to_transform().m, the scale matrix
RDEVN void anim_factors(const CamAnim* ca, float dt, float* tr, float* rot, float* scale) {
    const f3 t0{ca->t[0][0], ca->t[0][1], ca->t[0][2]}, t1{ca->t[1][0], ca->t[1][1], ca->t[1][2]};
    const f3 trans = t0 * (1.0f - dt) + t1 * dt;
    // quat_slerp (quaternion.rs:168-180)
    const float q1[4] = {ca->r[0][0], ca->r[0][1], ca->r[0][2], ca->r[0][3]}, q2[4] = {ca->r[1][0], ca->r[1][1], ca->r[1][2], ca->r[1][3]};
    const float cos_theta = (q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2]) + q1[3] * q2[3];
    float q[4];
    if (cos_theta > 0.9995f) {
        float u[4];
#pragma unroll
        for (int i = 0; i < 4; i++) u[i] = q1[i] * (1.0f - dt) + q2[i] * dt;
        const float n = sqrtf((u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) + u[3] * u[3]), inv = 1.0f / n;  // quat_normalize: v * (1 / n), w / n
        q[0] = u[0] * inv; q[1] = u[1] * inv; q[2] = u[2] * inv; q[3] = u[3] / n;
    } else {
        const float theta = rspt_acosf(cos_theta < -1.0f ? -1.0f : (cos_theta > 1.0f ? 1.0f : cos_theta));
        const float thetap = theta * dt;
        float u[4];
#pragma unroll
        for (int i = 0; i < 4; i++) u[i] = q2[i] - q1[i] * cos_theta;
        const float n = sqrtf((u[0] * u[0] + u[1] * u[1] + u[2] * u[2]) + u[3] * u[3]), inv = 1.0f / n;
        const float qp[4] = {u[0] * inv, u[1] * inv, u[2] * inv, u[3] / n};
        const float c = rspt_cosf(thetap), sn = rspt_sinf(thetap);
#pragma unroll
        for (int i = 0; i < 4; i++) q[i] = q1[i] * c + qp[i] * sn;
    }
    // Quaternion::to_transform().m (quaternion.rs:80-109): the transpose of the matrix written there
    const float xx = q[0] * q[0], yy = q[1] * q[1], zz = q[2] * q[2], xy = q[0] * q[1], xz = q[0] * q[2], yz = q[1] * q[2];
    const float wx = q[0] * q[3], wy = q[1] * q[3], wz = q[2] * q[3];
    const float rot_[16] = {1.0f - 2.0f * (yy + zz), 2.0f * (xy - wz), 2.0f * (xz + wy), 0.0f,
                            2.0f * (xy + wz), 1.0f - 2.0f * (xx + zz), 2.0f * (yz - wx), 0.0f,
                            2.0f * (xz - wy), 2.0f * (yz + wx), 1.0f - 2.0f * (xx + yy), 0.0f,
                            0.0f, 0.0f, 0.0f, 1.0f};
#pragma unroll
    for (int i = 0; i < 16; i++) { rot[i] = rot_[i]; scale[i] = (i % 5 == 0) ? 1.0f : 0.0f; tr[i] = (i % 5 == 0) ? 1.0f : 0.0f; }
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) scale[4 * i + j] = ca->s[0][4 * i + j] * (1.0f - dt) + ca->s[1][4 * i + j] * dt;
    tr[3] = trans.x; tr[7] = trans.y; tr[11] = trans.z;
}
RDEVN void camera_to_world_at(const RenderDev& rd, float time_sample, float* m) {
#pragma unroll
    for (int i = 0; i < 16; i++) m[i] = rd.camera_to_world[i];
    const CamAnim* ca = rd.cam_anim;
    if (!ca) return;
    const float time = rd.shutter_open * (1.0f - time_sample) + rd.shutter_close * time_sample;  // pbrt.rs lerp
    if (time <= ca->time[0]) return;
    if (time >= ca->time[1]) {
#pragma unroll
        for (int i = 0; i < 16; i++) m[i] = ca->end[i];
        return;
    }
    const float dt = (time - ca->time[0]) / (ca->time[1] - ca->time[0]);
    float tr[16], rot[16], scale[16], tmp[16];
    anim_factors(ca, dt, tr, rot, scale);
    mat4_mul(tr, rot, tmp);      // Transform::translate(&trans) * rotate.to_transform()
    mat4_mul(tmp, scale, m);     //   * Transform { m: scale, .. }
}

// ---- a moving TransformedPrimitive (primitive.rs:198-265) at a ray's time ----
struct InstAnim {        // per moving instance: what AnimatedTransform::new leaves (as for the camera) + the end key's stored inverse
    CamAnim keys;        // keys.end = end_transform.m
    float mi_end[16];    // end_transform.m_inv
    uint32_t identity_end;
    uint32_t pad[3];
};
// Matrix4x4::inverse (transform.rs:128-200): Gauss-Jordan elimination, the pivot the largest remaining element (later candidates win ties) — mat4_inverse.h
}  // namespace rspt
#include "mat4_inverse.h"
namespace rspt {
// primitive_to_world.interpolate(r.time) (transform.rs:2081-2113) as an InstDev: the start Transform up to the start time (and for a static
// instance), the end Transform from the end time on, in between translate(trans) * rotate.to_transform() * Transform { scale, inverse(scale) }
// — m the product of the m's, m_inv the reverse product of the inverses (Transform * Transform, :869-877)
RDEVN InstDev inst_at(const SceneDev& sc, uint32_t index, float time) {
    InstDev in = sc.inst[index];
    if (in.anim == RSPT_MISS || !sc.inst_anim) return in;
    const InstAnim& an = sc.inst_anim[in.anim];
    if (time <= an.keys.time[0]) return in;
    if (time >= an.keys.time[1]) {
        for (int i = 0; i < 12; i++) { in.m[i] = an.keys.end[i]; in.mi[i] = an.mi_end[i]; }
        for (int i = 0; i < 4; i++) { in.m3[i] = an.keys.end[12 + i]; in.mi3[i] = an.mi_end[12 + i]; }
        in.identity = an.identity_end;
        return in;
    }
    const float dt = (time - an.keys.time[0]) / (an.keys.time[1] - an.keys.time[0]);
    float tr[16], rot[16], scale[16], tmp[16], m[16], rot_t[16], tr_inv[16], scale_inv[16], mi[16];
    anim_factors(&an.keys, dt, tr, rot, scale);
    mat4_mul(tr, rot, tmp);
    mat4_mul(tmp, scale, m);
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) { rot_t[4 * i + j] = rot[4 * j + i]; tr_inv[4 * i + j] = i == j ? 1.0f : 0.0f; }
    tr_inv[3] = -tr[3]; tr_inv[7] = -tr[7]; tr_inv[11] = -tr[11];
    mat4_inverse(scale, scale_inv);
    mat4_mul(rot_t, tr_inv, tmp);        // (T * R).m_inv = R.m_inv * T.m_inv
    mat4_mul(scale_inv, tmp, mi);        // ((T * R) * S).m_inv = S.m_inv * (T * R).m_inv
    bool ident = true;                   // Transform::is_identity (transform.rs:291-308): m against the identity, element by element
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) if (m[4 * i + j] != (i == j ? 1.0f : 0.0f)) ident = false;
    for (int i = 0; i < 12; i++) { in.m[i] = m[i]; in.mi[i] = mi[i]; }
    for (int i = 0; i < 4; i++) { in.m3[i] = m[12 + i]; in.mi3[i] = mi[12 + i]; }
    in.identity = ident ? 1u : 0u;
    return in;
}

// What a TRAVERSAL needs of primitive_to_world.interpolate(r.time): the inverse's matrix (Transform::transform_ray with m_inv, primitive.rs:218-222) and, in the
// reference's instancing mode only, Transform::is_identity of the interpolated Transform (want_ident).  Same values as inst_at's; what it leaves out is the product
// m = T * R * S, needed only for that flag: row i of m ends in ((+-0 + +-0) + +-0) + trans[i] * 1 (the fourth columns of R and S are (0 0 0 1) by construction), so a
// translation with a non-zero (or NaN) component cannot be the identity and m is formed only when all three are zero.
RDEVN void inst_inverse_at(const SceneDev& sc, const InstDev& in, float time, bool want_ident, float* mi /* rows 0..2 */, float* mi3, bool* ident) {
    const InstAnim& an = sc.inst_anim[in.anim];
    if (time <= an.keys.time[0]) {
#pragma unroll
        for (int i = 0; i < 12; i++) mi[i] = in.mi[i];
#pragma unroll
        for (int i = 0; i < 4; i++) mi3[i] = in.mi3[i];
        *ident = in.identity != 0u;
        return;
    }
    if (time >= an.keys.time[1]) {
#pragma unroll
        for (int i = 0; i < 12; i++) mi[i] = an.mi_end[i];
#pragma unroll
        for (int i = 0; i < 4; i++) mi3[i] = an.mi_end[12 + i];
        *ident = an.identity_end != 0u;
        return;
    }
    const float dt = (time - an.keys.time[0]) / (an.keys.time[1] - an.keys.time[0]);
    float tr[16], rot[16], scale[16], tmp[16], rot_t[16], tr_inv[16], scale_inv[16], full[16];
    anim_factors(&an.keys, dt, tr, rot, scale);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) { rot_t[4 * i + j] = rot[4 * j + i]; tr_inv[4 * i + j] = i == j ? 1.0f : 0.0f; }
    tr_inv[3] = -tr[3]; tr_inv[7] = -tr[7]; tr_inv[11] = -tr[11];
    mat4_inverse(scale, scale_inv);
    mat4_mul(rot_t, tr_inv, tmp);        // (T * R).m_inv = R.m_inv * T.m_inv
    mat4_mul(scale_inv, tmp, full);      // ((T * R) * S).m_inv = S.m_inv * (T * R).m_inv
#pragma unroll
    for (int i = 0; i < 12; i++) mi[i] = full[i];
#pragma unroll
    for (int i = 0; i < 4; i++) mi3[i] = full[12 + i];
    bool id = false;
    if (want_ident && !(tr[3] != 0.0f || tr[7] != 0.0f || tr[11] != 0.0f)) {
        float m[16];
        mat4_mul(tr, rot, tmp);
        mat4_mul(tmp, scale, m);
        id = true;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) if (m[4 * i + j] != (i == j ? 1.0f : 0.0f)) id = false;
    }
    *ident = id;
}

// ---- PerspectiveCamera::generate_ray_differential (perspective.rs:190-280), differentials dropped ----
// p_lens: (CameraSample.p_lens, CameraSample.time) — the time value matters to a moving camera only
RDEV void camera_ray(const RenderDev& rd, f2 p_film, f3 p_lens, f3* o_out, f3* d_out, float* tmax_out) {
    f3 p_camera = xf_point(rd.raster_to_camera, f3{p_film.x, p_film.y, 0.0f});
    f3 o{0.0f, 0.0f, 0.0f}, d = normalize(p_camera);
    float c2w[16];
    camera_to_world_at(rd, p_lens.z, c2w);
    if (rd.lens_radius > 0.0f) {
        f2 pl = concentric_disk(f2{p_lens.x, p_lens.y});
        pl = f2{pl.x * rd.lens_radius, pl.y * rd.lens_radius};
        float ft = rd.focal_distance / d.z;
        f3 p_focus = o + d * ft;
        o = f3{pl.x, pl.y, 0.0f};
        d = normalize(p_focus - o);
    }
    // Transform::transform_ray (transform.rs:538-595)
    f3 o_err;
    f3 ow = xf_point_err(c2w, o, &o_err);
    f3 dw = xf_vector(c2w, d);
    float ls = len2(dw);
    float t_max = RSPT_INF;
    if (ls > 0.0f) {
        float dt = dot(vabs(dw), o_err) / ls;
        ow = ow + dw * dt;
        t_max -= dt;
    }
    *o_out = ow; *d_out = dw; *tmax_out = t_max;
}

}  // namespace rspt
