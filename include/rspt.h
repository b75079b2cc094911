/* rspt.h — C ABI of librspt.so: the MI355X wavefront path-tracing integrator that
 * replaces rs_pbrt's per-tile render loop.
 *
 * The reference (wahn/rs_pbrt v0.9.12) has NO FFI of its own; this
 * header is the boundary a maintainer binds from Rust (see INTEGRATION.md for the
 * `extern "C"` block and the shim that flattens `Scene` into these structs).
 * Each entry point cites the reference interface it stands in for; all paths are
 * relative to the rs_pbrt source tree.
 *
 * Conventions
 *   - plain C, little endian, 4-byte IEEE floats, no torch / HIP types in signatures;
 *   - every function returns 0 on success or a negative RSPT_E_* code; the text of
 *     the last error on the calling thread is available from rspt_last_error();
 *   - no C++ exception and no panic crosses this boundary;
 *   - inputs are deep-copied at rspt_scene_create (caller may free afterwards);
 *   - calls on one scene handle are blocking and not re-entrant
 *     (mirrors: render() is entered once from pbrt_cleanup, src/core/api.rs:2366-2369).
 */
#ifndef RSPT_H
#define RSPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSPT_ABI_VERSION 21

/* error codes */
#define RSPT_OK 0
#define RSPT_E_INVALID (-1)     /* null pointer, bad size, unsupported enum value       */
#define RSPT_E_NODEVICE (-2)    /* no usable gfx950 device / HIP runtime failure at init  */
#define RSPT_E_HIP (-3)         /* a HIP call failed (message has file:line + hip string) */
#define RSPT_E_UNSUPPORTED (-4) /* scene feature outside the accelerated path (caller
                                    falls back to the CPU loop, integrator.rs:70)           */
#define RSPT_E_NOMEM (-5)
#define RSPT_E_PEER (-6)        /* film_reduce: another rank of the communicator failed its render (see film_reduce) */

typedef struct rspt_scene_s* rspt_scene_t;

/* ---- acceleration structure ------------------------------------------------------
 * One entry per LinearBVHNode, same order as BVHAccel.nodes
 * (src/accelerators/bvh.rs:77-85; flatten order :358-392).  For interior nodes
 * `offset` is the index of the second child (first child = own index + 1); for
 * leaves it is the index of the first primitive in the BVH-ordered primitive list. */
typedef struct {
    float bmin[3], bmax[3];
    int32_t offset;
    uint16_t n_prims;
    uint8_t axis, pad;
} rspt_bvh_node; /* 32 B */

/* One entry per element of BVHAccel.primitives (the *ordered* list, bvh.rs:144-149);
 * each is a GeometricPrimitive wrapping a Triangle (src/core/primitive.rs:100-105,
 * src/shapes/triangle.rs:84-96).  v[] index the global vertex arrays below. */
#define RSPT_MESH_INSTANCE 0xffffffffu /* rspt_prim.mesh of a TransformedPrimitive: v[0] = index into instances[] */
typedef struct {
    uint32_t v[3];
    uint32_t mesh;       /* index into meshes[], or RSPT_MESH_INSTANCE               */
    uint32_t material;   /* index into materials[]; 0xffffffff = no material
                            (path.rs:109-116 passes straight through)               */
    int32_t area_light;  /* index into lights[] or -1 (primitive.rs:193-195)         */
} rspt_prim; /* 24 B */

/* Per-TriangleMesh flags (triangle.rs:24-46).  Vertex data of all meshes is
 * concatenated into the global arrays; a mesh without normals/tangents/uvs simply
 * has has_* = 0 and its slots in N/S/UV are ignored. */
typedef struct {
    uint32_t has_n, has_s, has_uv;
    uint32_t flip; /* reverse_orientation ^ transform_swaps_handedness (triangle.rs:324) */
    /* TriangleMesh.alpha_mask / shadow_alpha_mask (triangle.rs:39-40; "alpha" / "shadowalpha" shape parameters, api.rs:1920-1965):
     * 0, or 1 + index of a float texture (its first channel).  A candidate hit where the texture evaluates to exactly 0 — at the hit's
     * uv / p, without ray differentials — is no hit: Triangle::intersect tests alpha_mask (triangle.rs:313-330), Triangle::intersect_p
     * both (:593-655).  Meshes that emit light may not carry masks (the pdf of a light's own triangle would need them). */
    uint32_t alpha_tex, shadow_alpha_tex;
    /* GeometricPrimitive.medium_interface (primitive.rs:103, filled from the graphics state's MediumInterface, api.rs:734-767,
     * 2858-2870): 0 = no medium, else 1 + index into media[].  Only VolPathIntegrator looks at it; a surface is a medium transition
     * when the two differ (medium.rs:340-362), otherwise rays keep the medium they travel in (primitive.rs:160-170). */
    uint32_t medium_inside, medium_outside;
} rspt_mesh; /* 32 B */

/* ---- participating media (src/core/medium.rs, src/media/{homogeneous,grid}.rs; MakeNamedMedium api.rs:953-1037) ----
 * sigma_a / sigma_s already multiplied by "scale".  The phase function is HenyeyGreenstein { g } (medium.rs:296-331).
 * HOMOGENEOUS: sigma_t = sigma_s + sigma_a is formed by the library as HomogeneousMedium::new does.
 * GRID (GridDensityMedium, "heterogeneous"): density[nz][ny][nx] over the unit cube of medium space, world_to_medium = the
 *   inverse of the medium's medium_to_world (Transform.m, row major).  sigma_t is the RED channel of sigma_a + sigma_s and the
 *   majorant 1 / max(density), as GridDensityMedium::new computes them (grid.rs:30-56).  Its tr / sample draw one or two sampler
 *   values per tracking step, so it is rendered under the pixel samplers only (each lane follows its tile's stream in program
 *   order); with Sobol' / Halton — whose 1024 / 1000 dimensions such a walk exhausts within a bounce or two, where the reference
 *   panics — rspt_render answers RSPT_E_UNSUPPORTED. */
enum { RSPT_MEDIUM_HOMOGENEOUS = 1, RSPT_MEDIUM_GRID = 2 };
typedef struct {
    uint32_t kind;
    float sigma_a[3], sigma_s[3];
    float g;
    int32_t nx, ny, nz;      /* GRID */
    uint32_t pad;
    const float* density;    /* GRID: nx * ny * nz floats, x fastest */
    float world_to_medium[16];
} rspt_medium; /* 120 B */

/* ---- materials: the parameters of Material::create, assembled into BxDF lists by the library --------------------
 * Every parameter of a reference material is a texture (TextureParams::get_spectrum_texture / get_float_texture wrap a literal
 * value in a ConstantTexture, src/core/paramset.rs:622-735), so a material record is the material's kind plus one texture
 * reference per parameter: 1 + index into textures[], 0 = parameter absent (possible only for the *_or_null parameters:
 * bumpmap; metal / uber uroughness, vroughness).  Float parameters read the first channel of their texture.  The shim passes
 * what the reference's material structs hold and nothing else; Material::compute_scattering_functions
 * (src/core/material.rs:63-113, src/materials/{matte,plastic,mirror,glass,metal,substrate,uber,translucent,mixmat}.rs) is
 * restated by the library (rs_pbrt_amd/csrc/material_assembly.h): parameters bound to constant textures are folded into the
 * lobe list once per material (clamp, `is_black` guards, roughness remapping, OrenNayar A / B), parameters bound to other
 * textures are evaluated per hit by the texture stage: Kd, Ks and the roughnesses leave the lobe list's shape alone and only
 * scale lobes (the fast path); a non-constant texture on any other parameter (sigma, index, opacity, Kr, Kt, reflect, transmit, eta, k,
 * a glass roughness, a mix amount) makes the material "dynamic" — its lobe list is built per hit on the device, under every
 * integrator and sampler.  A mix may contain mixes: the tree is flattened into its non-mix materials, each scaled by its immediate
 * parent's amount only (a MixMaterial ignores the scale it is handed itself, mixmat.rs:50).  Refused with RSPT_E_UNSUPPORTED: a
 * tree of mixes with more than 8 non-mix materials, more than 12 varying textures on a dynamic material, a material that can push
 * more than 8 BxDFs (Bsdf::add asserts there, reflection.rs:247). */
enum {
    RSPT_MAT_MATTE = 1,       /* matte.rs:43-86:       kd, sigma, bumpmap                                               */
    RSPT_MAT_PLASTIC = 2,     /* plastic.rs:57-125:    kd, ks, roughness, remap_roughness, bumpmap                      */
    RSPT_MAT_MIRROR = 3,      /* mirror.rs:34-70:      kr, bumpmap                                                      */
    RSPT_MAT_GLASS = 4,       /* glass.rs:83-211:      kr, kt, uroughness, vroughness, index, remap_roughness, bumpmap  */
    RSPT_MAT_METAL = 5,       /* metal.rs:144-205:     eta, k, roughness, [uroughness], [vroughness], remap_roughness, bumpmap */
    RSPT_MAT_SUBSTRATE = 6,   /* substrate.rs:62-114:  kd, ks, uroughness, vroughness, remap_roughness, bumpmap         */
    RSPT_MAT_UBER = 7,        /* uber.rs:114-259:      kd, ks, kr, kt, roughness, [uroughness], [vroughness], opacity, index, remap_roughness, bumpmap */
    RSPT_MAT_TRANSLUCENT = 8, /* translucent.rs:64-189: kd, ks, reflect, transmit, roughness, remap_roughness, bumpmap  */
    RSPT_MAT_MIX = 9          /* mixmat.rs:43-305:     m1, m2 (material indices), amount                                */
};
typedef struct {
    uint32_t kind;            /* RSPT_MAT_*                                                                              */
    uint32_t kd, ks, kr, kt;  /* spectrum parameters "Kd" "Ks" "Kr" "Kt"                                                */
    uint32_t reflect, transmit; /* translucent                                                                           */
    uint32_t opacity;         /* uber                                                                                    */
    uint32_t eta, k;          /* metal: spectrum "eta", "k"                                                              */
    uint32_t amount;          /* mix: spectrum "amount" (MixMaterial.scale)                                              */
    uint32_t sigma;           /* matte: float, degrees                                                                   */
    uint32_t roughness, uroughness, vroughness; /* float                                                                 */
    uint32_t index;           /* glass / uber: float "index" (uber.rs calls it eta)                                      */
    uint32_t bumpmap;         /* float, or 0                                                                             */
    uint32_t remap_roughness; /* "remaproughness" (default true)                                                         */
    uint32_t m1, m2;          /* mix: indices into materials[]; either may be a mix itself (which then ignores the scale
                                 this one hands it, mixmat.rs:50: `_scale`)                                              */
} rspt_material_desc; /* 80 B */

/* What the library assembles from a material record: the BxDF list of Bsdf.bxdfs in push order (it matters: Bsdf::sample_f
 * picks the comp-th matching lobe, reflection.rs:307-336).  Not an input of the ABI; rspt_material_lobes returns it so that a
 * test can compare the assembly with a restatement of the reference's recipes. */
enum {
    RSPT_BXDF_LAMBERT_R = 1,    /* LambertianReflection      reflection.rs:953-998   */
    RSPT_BXDF_OREN_NAYAR = 2,   /* OrenNayar                 reflection.rs:1049-1125 */
    RSPT_BXDF_SPECULAR_R = 3,   /* SpecularReflection        reflection.rs:711-752   */
    RSPT_BXDF_SPECULAR_T = 4,   /* SpecularTransmission      reflection.rs:755-838   */
    RSPT_BXDF_FRESNEL_SPEC = 5, /* FresnelSpecular           reflection.rs:841-950   */
    RSPT_BXDF_MICROFACET_R = 6, /* MicrofacetReflection (TrowbridgeReitz, visible-area
                                   sampling)                 reflection.rs:1128-1209 */
    RSPT_BXDF_LAMBERT_T = 7,    /* LambertianTransmission    reflection.rs:1001-1046 */
    RSPT_BXDF_MICROFACET_T = 8, /* MicrofacetTransmission (TrowbridgeReitz, radiance mode):
                                   r = T, eta_a, eta_b, alphas  reflection.rs:1214-1371 */
    RSPT_BXDF_FRESNEL_BLEND = 9 /* FresnelBlend: r = Rd, t = Rs, alphas
                                                             reflection.rs:1374-1478 */
};
enum {
    RSPT_FRESNEL_NOOP = 0,      /* reflection.rs:698-705 */
    RSPT_FRESNEL_DIELECTRIC = 1,/* reflection.rs:686-696; eta_a = eta_i, eta_b = eta_t */
    RSPT_FRESNEL_CONDUCTOR = 2  /* reflection.rs:672-684; eta_i = 1, c1 = eta_t, c2 = k */
};
typedef struct {
    uint32_t type;      /* RSPT_BXDF_*                                                */
    uint32_t fresnel;   /* RSPT_FRESNEL_* (SPECULAR_R, MICROFACET_R)                  */
    float r[3];         /* R (reflective lobes) / T for pure transmissive lobes       */
    float t[3];         /* T of FRESNEL_SPEC                                          */
    float eta_a, eta_b; /* dielectric indices (SPECULAR_T, FRESNEL_SPEC, dielectric Fresnel) */
    float alpha_x, alpha_y; /* TrowbridgeReitz alphas, already remapped and max(.,1e-3)
                               (microfacet.rs:233-254)                                */
    float c1[3], c2[3]; /* conductor eta_t, k                                         */
    float on_a, on_b;   /* OrenNayar A, B (reflection.rs:1057-1065)                   */
    float sc[3];        /* MixMaterial scale of this lobe (sc_opt, src/materials/mixmat.rs:43-70):
                           the reference multiplies it in front of the lobe's value       */
    uint32_t has_sc;    /* 0 = sc_opt is None                                         */
    uint32_t tex_r;     /* 0, or 1 + index of the texture bound to this lobe's colour:
                           the lobe is built with  r * texture.evaluate(si).clamp(0, inf)  (e.g.
                           matte.rs:59-62 with r = 1; uber.rs `op * kd`), evaluated per hit     */
    uint32_t tex_t;     /* the same for t (FRESNEL_SPEC T, FRESNEL_BLEND Rs).  A lobe whose resulting
                           colour(s) are black is not added, as in the reference's
                           `if !r.is_black()` guards (matte.rs:70, plastic.rs:70,84, substrate.rs:72) */
    uint32_t tex_ax;    /* 0, or 1 + index of the float texture behind alpha_x: a "roughness" / "uroughness" parameter
                           (plastic.rs:86-92, uber.rs, substrate.rs:76-85, metal.rs, translucent.rs); per hit
                           alpha_x = max(0.001, remap ? roughness_to_alpha(v) : v) (microfacet.rs:233-254)        */
    uint32_t tex_ay;    /* the same for alpha_y ("vroughness"; materials with one roughness bind it to both)     */
    uint32_t remap;     /* "remaproughness" of the textured alphas; bit 1 (RSPT_LOBE_NODIFF): the lobe came from the m2 side of a
                           MixMaterial, whose textures are evaluated at an interaction rebuilt without ray differentials
                           (mixmat.rs:58-69: SurfaceInteraction::new)                                              */
} rspt_bxdf; /* 116 B */
#define RSPT_LOBE_REMAP 1u
#define RSPT_LOBE_NODIFF 2u

typedef struct {
    float eta;          /* Bsdf.eta (reflection.rs:224)                               */
    uint32_t first_bxdf, n_bxdfs; /* slice of the lobe list; n_bxdfs <= 8 (reflection.rs:40) */
    uint32_t bump_tex;  /* 0, or 1 + index of the float texture of Material::bump
                           (src/core/material.rs:116-219), applied before the lobes are built */
} rspt_material;

/* ---- textures ---------------------------------------------------- */
/* MipMap<T> pyramid of an ImageTexture as rs_pbrt built it (src/core/mipmap.rs:56-196: power-of-two
 * levels after its Lanczos resampling, box-filtered with the texture's wrap mode; texels already
 * scaled / inverse-gamma-corrected by ImageTexture::new, src/textures/imagemap.rs:34-96).
 * channels = 3 for a spectrum texture, 1 for a float texture (convert_to_float = y()). */
typedef struct {
    uint32_t width, height;  /* level 0 resolution                                         */
    uint32_t n_levels;       /* MipMap::levels()                                           */
    uint32_t channels;       /* 1 or 3                                                     */
    const float* texels;     /* levels concatenated; level i is max(1, width >> i) x
                                max(1, height >> i), row major [t][s], `channels` floats   */
} rspt_image;
enum {
    RSPT_TEX_CONSTANT = 1,   /* ConstantTexture        src/textures/constant.rs: value                        */
    RSPT_TEX_IMAGE = 2,      /* ImageTexture           src/textures/imagemap.rs:114-149 (2-D mapping)         */
    RSPT_TEX_SCALE = 3,      /* ScaleTexture           src/textures/scale.rs: tex1 * tex2                     */
    RSPT_TEX_MIX = 4,        /* MixTexture             src/textures/mix.rs: tex1 (1 - a) + tex2 a, a = tex3 (float) */
    RSPT_TEX_CHECKERBOARD = 5, /* Checkerboard2DTexture src/textures/checkerboard.rs (no anti-aliasing there): tex1 / tex2,
                                  2-D mapping                                                                */
    RSPT_TEX_DOTS = 6,       /* DotsTexture            src/textures/dots.rs: tex1 = outside, tex2 = inside, 2-D mapping */
    RSPT_TEX_FBM = 7,        /* FBmTexture             src/textures/fbm.rs: octaves, omega (3-D mapping)      */
    RSPT_TEX_MARBLE = 8,     /* MarbleTexture          src/textures/marble.rs: octaves, omega, scale, variation */
    RSPT_TEX_WINDY = 9,      /* WindyTexture           src/textures/windy.rs                                  */
    RSPT_TEX_WRINKLED = 10   /* WrinkledTexture        src/textures/wrinkled.rs: octaves, omega               */
};
enum {
    RSPT_MAP_UV = 1,         /* UVMapping2D            texture.rs:91-121: map = su, sv, du, dv                */
    RSPT_MAP_PLANAR = 2,     /* PlanarMapping2D        texture.rs:222-257: map = vs[3], vt[3], ds, dt         */
    RSPT_MAP_SPHERICAL = 3,  /* SphericalMapping2D     texture.rs:123-170: world_to_texture                   */
    RSPT_MAP_CYLINDRICAL = 4,/* CylindricalMapping2D   texture.rs:172-220: world_to_texture                   */
    RSPT_MAP_IDENTITY3D = 5  /* IdentityMapping3D      texture.rs:259-283 (FBM / MARBLE / WINDY / WRINKLED)   */
};
enum { RSPT_WRAP_REPEAT = 0, RSPT_WRAP_BLACK = 1, RSPT_WRAP_CLAMP = 2 }; /* mipmap.rs:23-27; Black
                                looks texels up like Clamp (mipmap.rs:217-227, "TMP" branch)   */
typedef struct {
    uint32_t kind;           /* RSPT_TEX_*                                                 */
    uint32_t mapping;        /* RSPT_MAP_*                                                 */
    float map[8];
    uint32_t image;          /* index into images[] (IMAGE)                                */
    uint32_t trilinear;      /* do_trilinear (mipmap.rs:253-262); 0 = EWA                  */
    float max_aniso;         /* EWA eccentricity clamp (default 8)                         */
    uint32_t wrap;           /* RSPT_WRAP_*                                                */
    float value[3];          /* CONSTANT (float textures: value[0])                        */
    uint32_t tex1, tex2;     /* children: texture indices.  The graph below a texture bound to a material may
                                be at most three levels deep                                */
    uint32_t tex3;           /* MIX: the float `amount` texture                            */
    float world_to_texture[16]; /* row major, for SPHERICAL / CYLINDRICAL / IDENTITY3D      */
    int32_t octaves;         /* FBM, MARBLE, WRINKLED (default 8)                          */
    float omega;             /* FBM, MARBLE, WRINKLED (default 0.5)                        */
    float scale;             /* MARBLE (default 1)                                         */
    float variation;         /* MARBLE (default 0.2)                                       */
} rspt_texture; /* 160 B */

/* ---- lights (Scene.lights order, src/core/scene.rs:19-24) ------------------------- */
enum {
    RSPT_LIGHT_DIFFUSE_AREA = 1,/* DiffuseAreaLight on one triangle, src/lights/diffuse.rs:19-27;
                                    one per emissive triangle (api.rs:2810-2852)       */
    RSPT_LIGHT_POINT = 2,       /* PointLight   src/lights/point.rs:20-68:  p[0..2] = p_light, L = I      */
    RSPT_LIGHT_SPOT = 3,        /* SpotLight    src/lights/spot.rs:20-110: p[0..2] = p_light, L = I,
                                    p[3..11] = upper 3x3 of world_to_light.m (row major),
                                    p[12] = cos_total_width, p[13] = cos_falloff_start  */
    RSPT_LIGHT_DISTANT = 4,     /* DistantLight src/lights/distant.rs:25-75: p[0..2] = w_light (normalised), L = L;
                                    the world radius comes from the scene bounds (preprocess, :76-86)      */
    RSPT_LIGHT_INFINITE = 5     /* InfiniteAreaLight src/lights/infinite.rs:38-392: prim = index into envmaps[]
                                    (texels already multiplied by L), p[0..8] = upper 3x3 of light_to_world.m,
                                    p[9..17] = upper 3x3 of world_to_light.m (row major)                    */
};
typedef struct {
    uint32_t kind;
    uint32_t prim;       /* DIFFUSE_AREA: BVH-ordered primitive index of the emitting triangle */
    float L[3];          /* l_emit / I / L                                            */
    uint32_t two_sided;
    float p[24];         /* kind-specific parameters, see above                       */
} rspt_light; /* 120 B */

/* Environment map of an InfiniteAreaLight: the MipMap<Spectrum> pyramid rs_pbrt built
 * (src/core/mipmap.rs:56-196; power-of-two levels after its resampling, wrap mode Repeat) and the
 * scalar image its Distribution2D was built from (infinite.rs:120-137: lookup(st, fwidth).y() *
 * sin(theta) on a 2w x 2h grid).  The library rebuilds the conditional / marginal CDFs with
 * Distribution1D::new's arithmetic (src/core/sampling.rs:24-49,150-170). */
typedef struct {
    uint32_t width, height;  /* level 0 resolution                                         */
    uint32_t n_levels;       /* MipMap::levels(): 1 + log2(max(width, height))             */
    uint32_t pad;
    const float* texels;     /* rgb triples, levels concatenated; level i is
                                max(1, width >> i) x max(1, height >> i), row major [t][s] */
    uint32_t dist_nu, dist_nv;
    const float* dist_func;  /* [dist_nv][dist_nu]                                         */
} rspt_envmap;

/* ---- object instancing --------------------------------------------
 * ObjectBegin / ObjectEnd collect the primitives of a named object; every ObjectInstance wraps them — in a BVHAccel of their
 * own when there is more than one — in a TransformedPrimitive that sits in the scene's top-level aggregate like any other
 * primitive (api.rs:3024-3109, primitive.rs:198-272; its bounds: primitive_to_world.motion_bounds, primitive.rs:212-215).
 * Static transforms only (actually_animated == false), affine (last row 0 0 0 1).  The objects' nodes and primitives follow
 * the top-level aggregate's in nodes[] / prims[]; all indices (child offsets, leaf offsets) are absolute. */
typedef struct {
    uint64_t first_node, n_nodes; /* this object's BVHAccel.nodes; n_nodes == 0: a single primitive, no aggregate (api.rs:3046) */
    uint64_t first_prim, n_prims; /* its BVHAccel.primitives (BVH leaf order); triangles only, area_light = -1 ("Area lights
                                     not supported with object instancing", api.rs:2899)                                   */
} rspt_object;
typedef struct {
    uint32_t object;     /* index into objects[]                                       */
    float to_world[16];  /* primitive_to_world.start_transform.m, row major            */
    float from_world[16];/* ... .m_inv (the reference inverts once, Transform holds both) */
    /* A moving instance (ABI 20): TransformedPrimitive.primitive_to_world is an AnimatedTransform (primitive.rs:198-201,
     * transform.rs:894-943).  animated = its actually_animated (the two key matrices differ); then to_world / from_world are
     * start_transform at time[0], to_world_end / from_world_end are end_transform at time[1].  intersect / intersect_p
     * interpolate at the ray's time (primitive.rs:216-222, :258-262; transform.rs:2081-2113: translation and scale linearly,
     * rotation by slerp between the decomposed keys, m_inv as inverse(scale) * rotation^T * translate(-t)); the library does the
     * same per instance visit.  The top-level aggregate's bounds of such a primitive are the reference's
     * AnimatedTransform::motion_bounds (the shim passes the BVH rs_pbrt built; the library never computes them).
     * (A caller that builds the top-level tree itself gets them from rspt_motion_bounds.)
     * Served by all four integrators under every sampler (Sobol' / Halton: ABI 21 builds; the PCG-backed pixel samplers and the per-lane form of
     * directlighting: round 6), also next to alpha-masked meshes.  RSPT_E_UNSUPPORTED: next to a dynamic material under a pixel sampler. */
    uint32_t animated;
    float to_world_end[16];
    float from_world_end[16];
    float time[2];
} rspt_instance; /* 272 B */
/* What a hit inside an instance is:
 * REFERENCE  what rs_pbrt v0.9.12 does: Transform::transform_surface_interaction drops isect.primitive (transform.rs:856), so
 *            the hit has no material and no emission and PathIntegrator::li passes straight through it like a null-material
 *            surface (path.rs:109-116) while shadow rays are blocked (primitive.rs:258-265); an instance whose transform is the
 *            identity shrinks the ray's t_max and then reports no hit (primitive.rs:220-253).
 * FIXED      the behaviour of the fix commented out at primitive.rs:226-250: the transformed interaction keeps its primitive
 *            (material), identity instances report their hits. */
enum { RSPT_INSTANCING_REFERENCE = 0, RSPT_INSTANCING_FIXED = 1 };

typedef struct {
    const rspt_bvh_node* nodes; uint64_t n_nodes;
    const rspt_prim* prims;     uint64_t n_prims;     /* BVH leaf order */
    const rspt_mesh* meshes;    uint32_t n_meshes;
    const float* P;             /* xyz per vertex, world space (api.rs:1967-1971)     */
    const float* N;             /* xyz per vertex or NULL                             */
    const float* S;             /* xyz per vertex or NULL                             */
    const float* UV;            /* uv  per vertex or NULL                             */
    uint64_t n_vertices;
    const rspt_material_desc* materials; uint32_t n_materials;
    const rspt_light* lights;       uint32_t n_lights;
    const rspt_envmap* envmaps;     uint32_t n_envmaps;
    const rspt_texture* textures;   uint32_t n_textures;
    const rspt_image* images;       uint32_t n_images;
    /* instancing: with n_instances == 0 the whole of nodes[] / prims[] is the scene's aggregate */
    const rspt_object* objects;     uint32_t n_objects;
    const rspt_instance* instances; uint32_t n_instances;
    uint64_t n_top_nodes, n_top_prims; /* the top-level aggregate = nodes[0 .. n_top_nodes), prims[0 .. n_top_prims) */
    uint32_t instancing_mode;          /* RSPT_INSTANCING_* */
    uint32_t n_media;
    const rspt_medium* media;          /* RenderOptions.named_media, referenced by rspt_mesh.medium_inside / _outside */
} rspt_scene_desc;

/* Sampler tables owned by the host.
 * Sobol': generator matrices (src/core/sobolmatrices.rs:5-7, :53463, :54155); vdc rows are
 * zero-padded to 52 entries.  May be NULL when sampler_kind is not SOBOL.
 * Halton: RADICAL_INVERSE_PERMUTATIONS (src/samplers/halton.rs:19-26, built by
 * compute_radical_inverse_permutations, src/core/lowdiscrepancy.rs:2165-2187): the digit
 * permutation of the i-th prime starts at the sum of the primes before it; a prefix covering
 * the dimensions a render can reach (5 + 8 per bounce) is enough.  NULL unless HALTON. */
typedef struct {
    const uint32_t* sobol32;   /* [1024*52]  SOBOL_MATRICES_32          */
    const uint64_t* vdc;       /* [25*52]    VD_C_SOBOL_MATRICES        */
    const uint64_t* vdc_inv;   /* [26*52]    VD_C_SOBOL_MATRICES_INV    */
    const uint16_t* halton_perms;
    uint64_t n_halton_perms;   /* entries available in halton_perms     */
} rspt_sampler_tables;

enum { RSPT_SAMPLER_SOBOL = 1,            /* src/samplers/sobol.rs                   */
       RSPT_SAMPLER_HALTON = 2,           /* src/samplers/halton.rs (the reference's default, api.rs:526) */
       /* The pixel samplers: their PCG32 state runs through all pixels and samples of a 16x16 tile (reseeded per
        * tile with tile.y * n_tiles.x + tile.x, integrator.rs:113-114; start_pixel before the pixel-bounds test, Q9) and every
        * dimension past `pixel_dimensions` is drawn from it on demand — a tile is one serial chain.  librspt runs one lane per
        * tile (tile_serial.h); `path` only. */
       RSPT_SAMPLER_RANDOM = 3,           /* src/samplers/random.rs                  */
       RSPT_SAMPLER_ZEROTWO = 4,          /* src/samplers/zerotwosequence.rs ("lowdiscrepancy" / "02sequence", api.rs:1694) */
       RSPT_SAMPLER_STRATIFIED = 5,       /* src/samplers/stratified.rs: spp must equal strat_x * strat_y */
       RSPT_SAMPLER_MAXMINDIST = 6 };     /* src/samplers/maxmin.rs: spp a power of two <= 65536, maxmin_c_pixel = C_MAX_MIN_DIST[log2 spp] */
enum { RSPT_LIGHTS_UNIFORM = 0, RSPT_LIGHTS_POWER = 1, RSPT_LIGHTS_SPATIAL = 2 };
                                           /* src/core/lightdistrib.rs:393-418         */

/* Everything SamplerIntegrator::render reads from camera, film, sampler and
 * PathIntegrator (integrator.rs:70-100; path.rs:24-34; film.rs:159-173;
 * perspective.rs:22-43; sobol.rs:15-20). */
typedef struct {
    int32_t full_res[2];           /* Film.full_resolution                            */
    int32_t crop_px[4];            /* cropped_pixel_bounds x0,y0,x1,y1 (film.rs:187-196) */
    int32_t sample_bounds[4];      /* Film::get_sample_bounds (film.rs:266-292)       */
    float filter_radius[2];
    float filter_table[256];       /* film.rs:198-211                                 */
    float max_sample_luminance;    /* film.rs:250-251; +inf by default                */
    float raster_to_camera[16];    /* row-major Transform.m (perspective.rs:32)       */
    float camera_to_world[16];     /* camera_to_world.start_transform.m (the only one unless camera_animated, below) */
    float lens_radius, focal_distance;
    float shutter_open, shutter_close;
    uint32_t sampler_kind;         /* RSPT_SAMPLER_*                                  */
    int64_t spp;                   /* Sobol': already rounded up to 2^k (sobol.rs:38-45) */
    uint32_t max_depth;            /* path.rs:30                                      */
    float rr_threshold;            /* path.rs:31                                      */
    uint32_t light_strategy;       /* RSPT_LIGHTS_*  (uniform is forced for 1 light,
                                      lightdistrib.rs:397)                            */
    uint32_t tile_size;            /* 16 (integrator.rs:75)                           */
    /* multi-GPU sharding of the Morton-ordered tile list (blockqueue/mod.rs:33-36):
     * this process renders tiles whose (morton_rank / tile_chunk) % shard_count ==
     * shard_index.  shard_count = 1 renders everything. */
    uint32_t shard_index, shard_count, tile_chunk;
    uint32_t sample_at_pixel_center; /* HaltonSampler "samplepixelcenter" (halton.rs:163-172) */
    /* which `li` the shared SamplerIntegrator::render loop calls (integrator.rs:48-69).  AO:
     * AOIntegrator::li (src/integrators/ao.rs:50-96) with its one 2-D sample array of ao_n_samples entries
     * per pixel sample (request_2d_array in preprocess, ao.rs:47-49; GlobalSampler array dimensions 5, 6);
     * max_depth, rr_threshold and light_strategy are ignored. */
    uint32_t integrator;             /* RSPT_INTEGRATOR_* */
    uint32_t ao_n_samples;           /* "nsamples" (default 64)        */
    uint32_t ao_cos_sample;          /* "cossample" (default true)     */
    uint32_t film_reduce;            /* multi-GPU (X1): 1 = before returning, sum the films of all ranks onto rank 0
                                        with one ncclReduce over the communicator of rspt_comm_init (shard_count must equal its
                                        world size, shard_index its rank); the other ranks' buffers keep their partial films.
                                        The ranks first agree (one ncclAllReduce of a status word) that all of them reached the
                                        reduce: if one failed — it returns its own error — the others return RSPT_E_PEER and no
                                        film is summed; nobody waits in the collective.  Every rank of the communicator must
                                        call rspt_render with film_reduce = 1 for the same frame (a rank that does not call at
                                        all is the communicator's time-out, not this library's).
                                        0 = no collective (single GPU, or the caller reduces) */
    rspt_sampler_tables tables;
    /* DirectLightingIntegrator (src/integrators/directlighting.rs:17-70): max_depth bounds the specular recursion
     * (default 5, api.rs:322-349); rr_threshold and light_strategy are ignored (uniform_sample_one_light gets no distribution) */
    uint32_t direct_strategy;        /* RSPT_DIRECT_SAMPLE_ALL (default, "all") | RSPT_DIRECT_SAMPLE_ONE ("one")                   */
    uint32_t pixel_dimensions;       /* pixel samplers: "dimensions" (default 4) = number of precomputed 1-D and 2-D sample vectors  */
    const int32_t* n_light_samples;  /* SAMPLE_ALL: Light::get_n_samples() per light ("samples" / "nsamples", default 1), after
                                        Sampler::round_count (the identity for Sobol' and Halton); NULL = 1 each                   */
    uint32_t strat_x, strat_y;       /* StratifiedSampler "xsamples" / "ysamples" (default 4 x 4)                                    */
    uint32_t strat_jitter;           /* "jitter" (default true)                                                                      */
    uint32_t allow_slow_paths;       /* 0: configurations this library is known to run SLOWER than the host's own tile loop are answered with
                                        RSPT_E_UNSUPPORTED so that the caller keeps the faster CPU path — today: a pixel sampler (one lane per
                                        16x16 tile, a tile is one serial PCG chain) with fewer than 2048 tiles in this shard (measured against 256
                                        host threads, Msamples/s: 920 tiles 5.6 vs 8.9, 2040 tiles 11.2 vs 8.0, 8160 tiles 26.7 vs 7.3;
                                        RSPT_SERIAL_MIN_TILES overrides).
                                        1: run them anyway (tests; a caller that wants the device's bit-identical sample values regardless) */
    const uint32_t* maxmin_c_pixel;  /* MaxMinDistSampler: the 32 columns of its generator matrix (lowdiscrepancy.rs:187-760, row log2 spp)  */
    /* Checkpoint / resume and progressive refinement (not in the reference, whose render loop is one-shot): render only
     * the pixel samples [sample_begin, sample_begin + sample_count) of every pixel; sample_count = 0 means all of spp.  With the
     * Sobol' and Halton samplers a sample's values depend on (pixel, sample index) alone, so the films of disjoint ranges add up to the
     * full frame's (xyz and filter_weight_sum alike): a caller that keeps the summed film and the next sample index can stop and
     * resume, show intermediate results, or re-render a lost rank's shard elsewhere.  The pixel samplers accept the full range only. */
    uint64_t sample_begin, sample_count;
    /* A moving camera (CameraBase.camera_to_world is an AnimatedTransform, core/transform.rs:894-2124; perspective.rs:261-279
     * transforms every camera ray — and its differentials — with the matrix interpolated at the ray's time,
     * lerp(CameraSample.time, shutter_open, shutter_close)).  camera_animated = 0: camera_to_world for every ray.  1: camera_to_world is
     * start_transform.m at camera_time[0], camera_to_world_end is end_transform.m at camera_time[1] (api.rs transform_start_time /
     * transform_end_time); the library decomposes both (AnimatedTransform::decompose: translation, rotation quaternion, scale) and
     * interpolates per ray (translation and scale linearly, rotation by slerp), start / end matrix outside the interval.  Lights
     * do not move in this ABI; object instances do (rspt_instance, ABI 20). */
    uint32_t camera_animated;
    float camera_to_world_end[16];
    float camera_time[2];
} rspt_render_desc;
/* RSPT_INTEGRATOR_VOLPATH: VolPathIntegrator::li (src/integrators/volpath.rs:60-347) with max_depth, rr_threshold and
 * light_strategy as for "path" (api.rs:350-380).  Camera rays start outside every medium (make_camera passes
 * MediumInterface::default().outside, api.rs:1638-1645).  As in v0.9.12: the BSDF- / phase-sampled half of estimate_direct is
 * multiplied by a transmittance that starts at Spectrum::default() = 0 (integrator.rs:531-536, scene.rs:79-106) and so adds nothing;
 * a ray that leaves the scene ends its path even after scattering in a medium (volpath.rs:288-345). */
enum { RSPT_INTEGRATOR_PATH = 0, RSPT_INTEGRATOR_AO = 1, RSPT_INTEGRATOR_DIRECT = 2, RSPT_INTEGRATOR_VOLPATH = 3 };
enum { RSPT_DIRECT_SAMPLE_ALL = 0, RSPT_DIRECT_SAMPLE_ONE = 1 };

typedef struct { float o[3], d[3], t_max; uint32_t id; } rspt_ray;   /* 32 B */
typedef struct { uint32_t prim; float t, b0, b1, b2; } rspt_hit;     /* prim = 0xffffffff on miss */

typedef struct {
    double t_render_s;      /* first launch -> film in host memory                    */
    double t_kernels_s;     /* sum of kernel time measured with HIP events            */
    double t_trace_s;       /* part of t_kernels_s spent in the traversal kernels     */
    uint64_t samples;       /* camera samples rendered by this process                */
    uint64_t rays_closest, rays_any;
    uint64_t nodes_visited, tris_tested; /* only filled when RSPT_COUNTERS=1          */
    uint64_t nan_samples;   /* integrator.rs:165-173                                  */
    uint64_t trace_launches;
    double alg_bytes;       /* SURVEY.md §8(d) B_alg over this render (needs counters) */
    /* per-launch durations summed over the render, each launch bracketed by HIP events on the stream it runs on (the
     * shadow-ray launch of a bounce runs on a second stream beside the closest-hit launch, so the two sums overlap in
     * wall time: t_trace_s is the wall time of the pairs, these are what rocprofv3 --kernel-trace reports per kernel) */
    double t_trace_closest_s, t_trace_any_s;
    double t_shade_s;       /* k_shade (+ k_texture) launches                         */
    uint64_t launches_closest, launches_any;
    uint64_t truncated_paths; /* paths cut off after RSPT_NULL_PASSES (default 1024) passes through surfaces without a BSDF (null materials;
                                 instanced hits in RSPT_INSTANCING_REFERENCE): rs_pbrt's loop has no such limit (path.rs:109-116) and
                                 would still be running; they keep the radiance gathered up to that point */
} rspt_stats;

/* version of this header the library was built against */
int rspt_abi_version(void);

/* Select the HIP device (ordinal among visible devices) and create streams.
 * Must be called once per process before any other call. */
int rspt_init(int32_t device);
void rspt_shutdown(void);

/* Multi-GPU: one process per GPU, scene replicated, Morton tile chunks dealt by (shard_index, shard_count,
 * tile_chunk); the only data-path collective is the final sum of the per-rank films (RCCL ncclReduce over xGMI).  The
 * reference has no counterpart (its tiles go to threads, integrator.rs:101-217; the per-tile merge is film.rs:346-371).
 * Rank 0 obtains an id, ships its bytes to the other ranks by any means (the Rust shim: a file or a pipe; bench.py: a
 * torch.distributed broadcast), every rank calls rspt_comm_init (collective).  librccl.so is bound at run time. */
#define RSPT_COMM_ID_BYTES 128
int rspt_comm_unique_id(uint8_t id[RSPT_COMM_ID_BYTES]);
int rspt_comm_init(int32_t rank, int32_t world, const uint8_t id[RSPT_COMM_ID_BYTES]);
int rspt_comm_destroy(void);
/* Which librccl the calls above are bound to: the copy the process has ALREADY mapped if there is one (a host that imported torch has torch's),
 * else $RSPT_RCCL_LIB, else librccl.so / librccl.so.1 by the loader's search, else /opt/rocm/lib/librccl.so.1 — one policy for every caller.
 * Binds on first use; the returned path (dladdr of ncclGetUniqueId) stays valid for the life of the process; NULL + RSPT_E_UNSUPPORTED in
 * rspt_last_error when no librccl can be loaded. */
const char* rspt_comm_library(void);

/* Replaces: RenderOptions::make_scene's hand-over of BVHAccel + lights to
 * Scene::new (src/core/api.rs:474-485, src/core/scene.rs:27-53): uploads the
 * flattened scene to HBM and builds the device-side triangle layout. */
int rspt_scene_create(const rspt_scene_desc* desc, rspt_scene_t* out);
int rspt_scene_destroy(rspt_scene_t scene);

/* Replaces: SamplerIntegrator::render specialised to PathIntegrator::li
 * (src/core/integrator.rs:70-220, src/integrators/path.rs:59-282).
 * film_xyzw receives, for every pixel of crop_px in row-major order, exactly what
 * Film.pixels holds after all merge_film_tile calls (film.rs:38-43,346-371):
 * xyz[3] and filter_weight_sum.  Host memory, (x1-x0)*(y1-y0)*4 floats. */
int rspt_render(rspt_scene_t scene, const rspt_render_desc* desc, float* film_xyzw,
                rspt_stats* stats);

/* Same, but leaves the film in device memory (for multi-GPU reduction with RCCL
 * by the caller): film_dev is a device pointer of the same shape. */
int rspt_render_device(rspt_scene_t scene, const rspt_render_desc* desc, void* film_dev,
                       rspt_stats* stats);

/* Debug/test hook: radiance returned by PathIntegrator::li for every camera sample,
 * before film accumulation (integrator.rs:158-164), as rgb triples indexed
 * [(pixel_index * spp + sample) * 3] over crop_px.  Host memory. */
int rspt_render_samples(rspt_scene_t scene, const rspt_render_desc* desc, float* li_rgb,
                        rspt_stats* stats);

/* Stage-level hook.  Replaces: Scene::intersect / Scene::intersect_p
 * (src/core/scene.rs:55-78) -> BVHAccel::intersect / intersect_p
 * (src/accelerators/bvh.rs:401-514) for a batch of rays.  any_hit = 0: closest hit,
 * out[i] = (prim, t, b0, b1, b2).  any_hit = 1: out[i].prim = 0 if occluded,
 * 0xffffffff if not; other fields 0. */
int rspt_trace(rspt_scene_t scene, const rspt_ray* rays, uint64_t n, rspt_hit* out,
               int any_hit);

/* Stage-level hook.  Replaces: LightDistribution::lookup(p) (src/core/lightdistrib.rs:33-39) of the distribution
 * create_light_sample_distribution(strategy) builds (:393-418; "uniform" is forced for a single light): the Distribution1D the path
 * integrator samples a light from at point p — func_out[n_lights], cdf_out[n_lights + 1] (sampling.rs:24-49), and for the spatial
 * strategy the voxel grid's resolution in nvox_out[3] (else 1 1 1) and p's voxel in voxel_out[3].  Spatial voxels that are built on
 * demand are built by this call. */
int rspt_light_distribution(rspt_scene_t scene, uint32_t light_strategy, const float p[3], float* func_out, float* cdf_out,
                            int32_t nvox_out[3], int32_t voxel_out[3]);

/* Stage-level hook, host only (no device, may be called without rspt_init).  Replaces: Material::compute_scattering_functions
 * (src/core/material.rs:63-113) for material `material` of a scene description, with the integrator's allow_multiple_lobes
 * (true for path / volpath / ao, false for directlighting): out_material / out_bxdfs (room for 8) receive the
 * lobe list the library assembles for that material — what Bsdf.bxdfs holds after the call wherever the material's textures are
 * constant; tex_* fields carry 1 + the index of the texture that completes a lobe per hit.  Returns the number of lobes;
 * RSPT_MATERIAL_DYNAMIC when a parameter that decides the SHAPE of the list (sigma, index, opacity, Kr, Kt, reflect, transmit, the
 * conductor's eta / k, a glass roughness, a mix amount) is bound to a non-constant texture — the list is then built per hit on the
 * device, from the texture values there, by the same assembly function; or an RSPT_E_* code. */
#define RSPT_MATERIAL_DYNAMIC 1000
int rspt_material_lobes(const rspt_scene_desc* desc, uint32_t material, uint32_t allow_multiple_lobes, rspt_material* out_material,
                        rspt_bxdf out_bxdfs[8]);

/* Stage-level hook, host only (no device needed).  Replaces: AnimatedTransform::new up to the rotation test (core/transform.rs:912-943) for the
 * camera's two key matrices — what rspt_render computes before its first launch when rspt_render_desc.camera_animated is set.  *animated_out:
 * 0 when the keys are equal (actually_animated = false: nothing else is written), else 1 and trs_out[46] = the two translations t[2][3], the two
 * rotation quaternions r[2][4] (x, y, z, w; the second one flipped onto the shorter arc), the two scale matrices s[2][16] (row-major; with
 * the key's translation column still in it, transform.rs:2079 — only the 3x3 block is interpolated). */
int rspt_camera_decompose(const float start_m[16], float start_time, const float end_m[16], float end_time, int32_t* animated_out, float trs_out[46]);

/* The hash of the kernel / ABI sources this binary was built from (16 hex digits; csrc/Makefile SRC_HASH): rocprofv3 summaries under
 * profiles/ carry it, and a measurement quotes a profile only when the LIBRARY that is running reports the same hash. */
const char* rspt_source_hash(void);

/* Host only (no device needed).  Replaces: AnimatedTransform::motion_bounds (core/transform.rs:2147-2163) with bound_point_motion
 * (:2164-2210) and interval_find_zeros (:2281-2350) — TransformedPrimitive::world_bound (core/primitive.rs:212-215) of a moving instance,
 * the box its caller hands rspt_bvh_build_bounds for the top-level tree.  [box_min, box_max] is the instanced object's own bound, the keys
 * are primitive_to_world's start / end matrices and times.  Equal keys: the start transform's transform_bounds; no rotation between the keys
 * (quaternion dot >= 0.9995): the union of the two keys' boxes; else per corner the union of the two end points and of the point at every
 * zero of the motion derivative (root isolation as the reference does it, in f32; the derivative's coefficients come from a matrix form
 * of the reference's expanded DerivativeTerm polynomials, :944-2030, evaluated in double precision — csrc/motion_bounds.h).
 * TOLERANCE (not bit-exact): edges that come from a key position equal the reference's bit for bit; an edge pushed out by a velocity zero
 * is located from coefficients that differ from the reference's f32 sums in their last bits (within 5e-7 of the box's largest extent,
 * either way round) and is then moved OUTWARD by 1e-6 of that extent (RSPT_MOTION_PAD): the result always CONTAINS the reference's box and
 * exceeds it by at most 1.5e-6 of the extent.  A top-level BVH built from these boxes may therefore split differently from rs_pbrt's own
 * (SAH bucket edges); a caller that needs rs_pbrt's exact tree passes the tree rs_pbrt built (rust_shim/gpu.rs does).
 * *flags_out (may be NULL): bit 0 = actually_animated, bit 1 = has_rotation.  RSPT_E_UNSUPPORTED where the reference would panic (a ninth
 * zero for one point and component), RSPT_E_INVALID for non-finite input. */
int rspt_motion_bounds(const float start_m[16], float start_time, const float end_m[16], float end_time, const float box_min[3],
                       const float box_max[3], float out_min[3], float out_max[3], int32_t* flags_out);

/* Stage-level hook.  Replaces: f32::sin / cos / ln / log2 / exp / acos / atan2 as the path uses them (concentric_sample_disk
 * sampling.rs:360-382, Trowbridge-Reitz sampling microfacet.rs:475-531, spherical directions and mappings, MIP level selection, roughness
 * remapping, medium transmittance) = the host libm's sinf / cosf / logf / log2f / expf / acosf / atan2f.  The device evaluates glibc's
 * algorithms operation by operation (rs_pbrt_amd/csrc/glibc_libm.h); this entry point runs one of them over an array so that a test can
 * compare it with the host's libm bit for bit.  y is read by RSPT_LIBM_ATAN2 only (out = atan2f(x[i], y[i])), else may be NULL.
 * RSPT_LIBM_MAT4_INVERSE (round 6): not libm but the same kind of hook — Matrix4x4::inverse (transform.rs:128-200) as the device evaluates it at every visit of a
 * moving instance (rs_pbrt_amd/csrc/mat4_inverse.h: the reference's Gauss-Jordan elimination with static indices); x and out then hold n row-major 4x4 matrices
 * (16 n floats).
 * Codes 8 .. 14 (round 6, still ABI 21: an older library answers RSPT_E_INVALID): the geometry of a traversal / shading step as the kernels call it, one element =
 * 16 floats in x and 16 floats in out (unused values 0), so that a test can hold the DEVICE functions to the reference's own text (tests/golden/geom_functions.npz):
 *   RSPT_LIBM_TRIANGLE           x = p0 p1 p2 o d t_max -> out = hit t b0 b1 b2: the watertight test of Triangle::intersect / intersect_p (triangle.rs:134-273, 450-591)
 *   RSPT_LIBM_BOX                x = p_min p_max o inv_dir dir_is_neg[3] t_max -> out[0] = Bounds3f::intersect_p (geometry.rs:2211-2268) as k_trace evaluates it,
 *                                out[1], out[2] = as k_trace_w4 does (two slots of its pair form when out[3] = 1, its literal chain for non-finite reciprocals when 0)
 *   RSPT_LIBM_OFFSET_RAY_ORIGIN  x = p p_error n w -> out = pnt3_offset_ray_origin (geometry.rs:1535-1556)
 *   RSPT_LIBM_MICROFACET         x = wo wh alpha_x alpha_y -> out = TrowbridgeReitzDistribution d(wh) lambda(wo) g1(wo) g(wo, wh) pdf(wo, wh) (microfacet.rs:256-297)
 *   RSPT_LIBM_VECTORS            x = a b eta u[2] -> out = vec3_cross_vec3(a, b) | vec3_coordinate_system(a)'s v2, v3 | refract(a, b, eta)'s wt, ok | cosine_sample_hemisphere(u)
 *   RSPT_LIBM_AREA_LIGHT         x = p0 p1 p2 ref_p u[2] flags (2: reversed orientation, 4: two-sided) -> out = pdf wi radiance(of L = 1) p n p_error: DiffuseAreaLight::sample_li
 *                                over Triangle::sample / sample_with_ref_point (lights/diffuse.rs:64-84, triangle.rs:676-744) on a triangle without vertex normals
 *   RSPT_LIBM_LOBE (48 floats per element, in and out)  x = the 29 words of one rspt_bxdf, wo, wi, u[2] -> out = f(wo, wi) pdf(wo, wi) | sample_f(wo, u): value wi pdf sampled_type | get_type:
 *                                one lobe of reflection.rs:711-1478 as the shade kernels evaluate it (the sampled value of a NON-specular lobe is what lobe_f gives: Bsdf::sample_f re-sums it) */
enum { RSPT_LIBM_SIN = 0, RSPT_LIBM_COS = 1, RSPT_LIBM_LOG = 2, RSPT_LIBM_LOG2 = 3, RSPT_LIBM_EXP = 4, RSPT_LIBM_ACOS = 5, RSPT_LIBM_ATAN2 = 6, RSPT_LIBM_MAT4_INVERSE = 7,
       RSPT_LIBM_TRIANGLE = 8, RSPT_LIBM_BOX = 9, RSPT_LIBM_OFFSET_RAY_ORIGIN = 10, RSPT_LIBM_MICROFACET = 11, RSPT_LIBM_VECTORS = 12, RSPT_LIBM_AREA_LIGHT = 13, RSPT_LIBM_LOBE = 14 };
int rspt_libm(uint32_t fn, const float* x, const float* y, uint64_t n, float* out);

/* Benchmark hook: same as rspt_trace on rays already resident in device memory,
 * repeated `repeat` times; returns average kernel milliseconds per launch. */
int rspt_trace_device(rspt_scene_t scene, const void* rays_dev, uint64_t n, void* out_dev,
                      int any_hit, int repeat, double* ms_per_launch);

/* Device memory helpers so a host without HIP bindings (tests, the Rust shim) can
 * stage buffers for the *_device entry points. */
int rspt_dev_alloc(uint64_t bytes, void** out);
int rspt_dev_free(void* p);
int rspt_dev_upload(void* dst_dev, const void* src_host, uint64_t bytes);
int rspt_dev_download(void* dst_host, const void* src_dev, uint64_t bytes);

/* Diagnostics of the last rspt_trace_device call: out[0] = BVH nodes fetched, out[1] = triangles
 * tested (both only when the environment has RSPT_COUNTERS=1; also valid after rspt_render),
 * out[2] = rays of the last launch whose traversal stack outgrew the persistent kernel's LDS
 * column and were redone by the 64-entry reference-order loop. */
int rspt_last_counters(uint64_t out[3]);

/* Host-side helper for callers that do not bring rs_pbrt's own accelerator (bench, tests,
 * tools).  Replaces: BVHAccel::new with SplitMethod::SAH over Triangle shapes
 * (src/accelerators/bvh.rs:96-392; Triangle::world_bound src/shapes/triangle.rs:126-133) and
 * yields the identical node array and primitive order.  tri_idx: n_tris*3 indices into P;
 * ordered_out[k] = input index of the primitive at BVH-ordered slot k; nodes_cap >= 2*n_tris
 * is always enough.  Returns the node count (>= 0) or an RSPT_E_* code (text from
 * rspt_bvh_last_error).  CPU only, no GPU needed; n_threads <= 0 = all host cores. */
int64_t rspt_bvh_build(const float* P, const uint32_t* tri_idx, uint64_t n_tris,
                       uint32_t max_prims_in_node, rspt_bvh_node* nodes_out, uint64_t nodes_cap,
                       uint32_t* ordered_out, int32_t n_threads);
/* The same over primitives given by their world bounds (n x (min xyz, max xyz)): the aggregate of a scene with object instances
 * holds TransformedPrimitives (bounds: Transform::transform_bounds of the object's, transform.rs:596-660) next to triangles. */
int64_t rspt_bvh_build_bounds(const float* bounds, uint64_t n_prims, uint32_t max_prims_in_node, rspt_bvh_node* nodes_out,
                              uint64_t nodes_cap, uint32_t* ordered_out, int32_t n_threads);
const char* rspt_bvh_last_error(void);

/* The same build on the GPU: every node of a tree level is split in one pass over the primitives
 * (segmented bounds / bucket reductions, scan-based order-preserving partition).  Same arguments, same result bit for bit
 * as rspt_bvh_build; n_vertices sizes the upload of P.  Needs rspt_init.  Errors through rspt_last_error. */
int64_t rspt_bvh_build_gpu(const float* P, uint64_t n_vertices, const uint32_t* tri_idx, uint64_t n_tris,
                           uint32_t max_prims_in_node, rspt_bvh_node* nodes_out, uint64_t nodes_cap, uint32_t* ordered_out);

const char* rspt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RSPT_H */
