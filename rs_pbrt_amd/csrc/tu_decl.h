// Which kernel instantiations live in which translation unit.  librspt.so is built from several .hip files compiled side by side
// (one hipcc run over everything took eight minutes): librspt.hip holds the host code and the small kernels and sees the heavy
// templates below only as `extern template` declarations (RSPT_TU_X = extern, RSPT_TU_ALL); each tu_*.hip defines RSPT_TU_X empty
// and one RSPT_TU_GROUP_* and so carries the device code and the launch stubs of that group.
#pragma once
#include "tile_serial.h"
#include "lane_serial.h"
#include "trace_w4.h"
#include "trace_w4q.h"

namespace rspt {

// ---- the shade stage's feature sets (kernels.h k_shade<F>; librspt.hip g_shade_variants) ----
constexpr uint32_t SV_DIFFUSE = RSPT_SF_LOBE(RSPT_BXDF_LAMBERT_R) | SF_L_AREA | SF_SOBOL;           // matte scenes under area lights: C1, C2
constexpr uint32_t SV_PLASTIC = SV_DIFFUSE | RSPT_SF_LOBE(RSPT_BXDF_MICROFACET_R) | SF_VERTEX;      // + plastic, smooth-shaded meshes: the C3 stand-in
constexpr uint32_t SV_TEXTURED = SV_PLASTIC | RSPT_SF_LOBE(RSPT_BXDF_OREN_NAYAR) | SF_TEX;          // + textured materials: the C4 stand-in
constexpr uint32_t SV_GENERIC = SF_ALL & ~SF_DYNAMIC & ~SF_ANIM;
// the same three under the Halton sampler (the reference's default, api.rs:526): without them a Halton render took the generic instantiation
// (212 VGPRs = 2 waves: C2 403 against Sobol's 473 M samples/s, the C3 stand-in 1705 against 2018)
constexpr uint32_t SV_DIFFUSE_H = (SV_DIFFUSE & ~SF_SOBOL) | SF_HALTON, SV_PLASTIC_H = (SV_PLASTIC & ~SF_SOBOL) | SF_HALTON, SV_TEXTURED_H = (SV_TEXTURED & ~SF_SOBOL) | SF_HALTON;
constexpr uint32_t SV_DYNAMIC = SF_ALL & ~SF_ANIM;                                                  // + per-hit lobe lists; SF_ALL itself: + moving instances

#define RSPT_TU_TS(I, A, M) \
    RSPT_TU_X template __global__ void k_tile_serial<I, A, M>(SceneDev, TexTables, LightDistDev, RenderDev, PathBuf, PixDesc, const TileRec*, uint32_t, uint32_t, int32_t, int32_t, float4*, float2*, uint32_t, uint32_t*);
#define RSPT_TU_TS2(I, M) RSPT_TU_TS(I, false, M) RSPT_TU_TS(I, true, M)
#define RSPT_TU_SHADE(F) RSPT_TU_X template __global__ void k_shade<F>(RSPT_SHADE_ARGS);
#define RSPT_TU_SHADE_W(F, W) RSPT_TU_X template __global__ void k_shade_w<F, W>(RSPT_SHADE_ARGS);
#define RSPT_TU_SHADE_M(F) RSPT_TU_X template __global__ void k_shade_m<F>(RSPT_SHADE_ARGS);          /* the MOVE forms (kernels.h PathBuf::move): one per feature set, */
#define RSPT_TU_SHADE_MW(F, W) RSPT_TU_X template __global__ void k_shade_mw<F, W>(RSPT_SHADE_ARGS);  /* built the way that set's default is built */
#define RSPT_TU_W4(ANY, OM, I, A) \
    RSPT_TU_X template __global__ void k_trace_w4<ANY, OM, I, A>(SceneDev, TexTables, const Wide4Node*, const uint2*, uint32_t, const uint32_t*, const uint32_t*, uint32_t, uint32_t*, const rspt_ray*, \
                                                                  const rspt_ray*, float4*, float4*, uint32_t*, rspt_hit*, uint32_t*, uint32_t*, uint2*, uint32_t, int, int, uint32_t, uint32_t*, uint32_t*, uint32_t);
#define RSPT_TU_W4A1(ANY, OM, A) \
    RSPT_TU_X template __global__ void k_trace_w4<ANY, OM, true, A, true>(SceneDev, TexTables, const Wide4Node*, const uint2*, uint32_t, const uint32_t*, const uint32_t*, uint32_t, uint32_t*, const rspt_ray*, \
                                                                           const rspt_ray*, float4*, float4*, uint32_t*, rspt_hit*, uint32_t*, uint32_t*, uint2*, uint32_t, int, int, uint32_t, uint32_t*, uint32_t*, uint32_t);
#define RSPT_TU_W4AF(ANY, OM, A) \
    RSPT_TU_X template __global__ void k_trace_fixup<ANY, OM, true, A, true>(SceneDev, TexTables, const uint32_t*, const uint32_t*, const rspt_ray*, const rspt_ray*, float4*, float4*, uint32_t*, rspt_hit*, uint32_t*);
#define RSPT_TU_W4A(ANY, OM) RSPT_TU_W4A1(ANY, OM, 0) RSPT_TU_W4AF(ANY, OM, false)   /* moving instances */
#define RSPT_TU_W4AM(ANY, OM) RSPT_TU_W4A1(ANY, OM, 1) RSPT_TU_W4A1(ANY, OM, 2) RSPT_TU_W4AF(ANY, OM, true)   /* moving instances next to alpha-masked meshes */
#define RSPT_TU_W4B(ANY, OM, B, T) \
    RSPT_TU_X template __global__ void k_trace_w4<ANY, OM, false, 0, false, B, T>(SceneDev, TexTables, const Wide4Node*, const uint2*, uint32_t, const uint32_t*, const uint32_t*, uint32_t, uint32_t*, const rspt_ray*, \
                                                                                   const rspt_ray*, float4*, float4*, uint32_t*, rspt_hit*, uint32_t*, uint32_t*, uint2*, uint32_t, int, int, uint32_t, uint32_t*, uint32_t*, uint32_t);   /* big workgroups, big LDS top */
#define RSPT_TU_W4_4(ANY, OM) RSPT_TU_W4(ANY, OM, false, 0) RSPT_TU_W4(ANY, OM, false, 1) RSPT_TU_W4(ANY, OM, true, 0) RSPT_TU_W4(ANY, OM, true, 1)
#define RSPT_TU_W4_S(ANY, OM) RSPT_TU_W4(ANY, OM, false, 2) RSPT_TU_W4(ANY, OM, true, 2)   /* alpha masks evaluated in line (alpha_simple) */
#define RSPT_TU_REF(ANY, OM, C, I, A) \
    RSPT_TU_X template __global__ void k_trace<ANY, OM, C, I, A>(SceneDev, TexTables, const uint32_t*, const uint32_t*, uint32_t, const rspt_ray*, const rspt_ray*, float4*, float4*, uint32_t*, rspt_hit*, \
                                                                 unsigned long long*, uint32_t*);
#define RSPT_TU_REFA1(ANY, OM, A) \
    RSPT_TU_X template __global__ void k_trace<ANY, OM, false, true, A, true>(SceneDev, TexTables, const uint32_t*, const uint32_t*, uint32_t, const rspt_ray*, const rspt_ray*, float4*, float4*, uint32_t*, rspt_hit*, \
                                                                                  unsigned long long*, uint32_t*);   /* moving instances (traverse<.., ANIM>) */
#define RSPT_TU_REFA(ANY, OM) RSPT_TU_REFA1(ANY, OM, false) RSPT_TU_REFA1(ANY, OM, true)   /* ... alone and next to alpha-masked meshes */
#define RSPT_TU_REF8(ANY, OM) \
    RSPT_TU_REF(ANY, OM, false, false, false) RSPT_TU_REF(ANY, OM, false, false, true) RSPT_TU_REF(ANY, OM, false, true, false) RSPT_TU_REF(ANY, OM, false, true, true) \
    RSPT_TU_REF(ANY, OM, true, false, false) RSPT_TU_REF(ANY, OM, true, false, true) RSPT_TU_REF(ANY, OM, true, true, false) RSPT_TU_REF(ANY, OM, true, true, true)
#define RSPT_TU_FIX(ANY, OM, I, A) \
    RSPT_TU_X template __global__ void k_trace_fixup<ANY, OM, I, A>(SceneDev, TexTables, const uint32_t*, const uint32_t*, const rspt_ray*, const rspt_ray*, float4*, float4*, uint32_t*, rspt_hit*, uint32_t*);
#define RSPT_TU_FIX4(ANY, OM) RSPT_TU_FIX(ANY, OM, false, false) RSPT_TU_FIX(ANY, OM, false, true) RSPT_TU_FIX(ANY, OM, true, false) RSPT_TU_FIX(ANY, OM, true, true)

#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS0A)
RSPT_TU_TS2(false, 0)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS0B)
RSPT_TU_TS2(true, 0)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS1A)
RSPT_TU_TS2(false, 1)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS1B)
RSPT_TU_TS2(true, 1)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS2A)
RSPT_TU_TS2(false, 2)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS2B)
RSPT_TU_TS2(true, 2)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS3A)
RSPT_TU_TS2(false, 3)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS3B)
RSPT_TU_TS2(true, 3)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS4A)
RSPT_TU_TS2(false, 4)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS4B)
RSPT_TU_TS2(true, 4)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS5)
RSPT_TU_TS2(true, 5)   /* path / ao under the pixel samplers over moving instances */
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS6)
RSPT_TU_TS2(true, 6)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS7)
RSPT_TU_TS2(true, 7)   /* volpath / directlighting per tile over moving instances */
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_TS8)
RSPT_TU_TS2(true, 8)
#endif
#define RSPT_TU_LANE(I, A) RSPT_TU_X template __global__ void k_lane_dl<I, A>(SceneDev, TexTables, LightDistDev, RenderDev, Batch, PathBuf, const uint32_t*, LaneDesc);
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_LANE_A)
RSPT_TU_LANE(false, false) RSPT_TU_LANE(false, true)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_LANE_B)
RSPT_TU_LANE(true, false) RSPT_TU_LANE(true, true)
#endif
#define RSPT_TU_LANE_ANIM(A) RSPT_TU_X template __global__ void k_lane_dl<true, A, true>(SceneDev, TexTables, LightDistDev, RenderDev, Batch, PathBuf, const uint32_t*, LaneDesc);
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_LANE_C)
RSPT_TU_LANE_ANIM(false) RSPT_TU_LANE_ANIM(true)   /* the per-lane directlighting over moving instances */
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_SHADE_A)
RSPT_TU_SHADE(SV_DIFFUSE) RSPT_TU_SHADE_W(SV_DIFFUSE, 3) RSPT_TU_SHADE_W(SV_DIFFUSE, 4)
RSPT_TU_SHADE(SV_PLASTIC) RSPT_TU_SHADE_W(SV_PLASTIC, 3) RSPT_TU_SHADE_W(SV_PLASTIC, 4)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_SHADE_B)
RSPT_TU_SHADE(SV_TEXTURED) RSPT_TU_SHADE_W(SV_TEXTURED, 3) RSPT_TU_SHADE_W(SV_TEXTURED, 4)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_SHADE_C)
RSPT_TU_SHADE(SV_GENERIC) RSPT_TU_SHADE_W(SV_GENERIC, 3) RSPT_TU_SHADE_W(SV_GENERIC, 4)
RSPT_TU_SHADE(SV_DYNAMIC) RSPT_TU_SHADE(SF_ALL)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_SHADE_H)
RSPT_TU_SHADE(SV_DIFFUSE_H) RSPT_TU_SHADE_W(SV_DIFFUSE_H, 3) RSPT_TU_SHADE(SV_PLASTIC_H) RSPT_TU_SHADE_W(SV_PLASTIC_H, 3) RSPT_TU_SHADE(SV_TEXTURED_H) RSPT_TU_SHADE_W(SV_TEXTURED_H, 3)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_SHADE_MA)
RSPT_TU_SHADE_MW(SV_DIFFUSE, 3) RSPT_TU_SHADE_MW(SV_PLASTIC, 3) RSPT_TU_SHADE_M(SV_DIFFUSE) RSPT_TU_SHADE_M(SV_PLASTIC)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_SHADE_MB)
RSPT_TU_SHADE_M(SV_TEXTURED) RSPT_TU_SHADE_MW(SV_TEXTURED_H, 3)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_SHADE_MC)
RSPT_TU_SHADE_M(SV_GENERIC)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_SHADE_MH)
RSPT_TU_SHADE_MW(SV_DIFFUSE_H, 3) RSPT_TU_SHADE_MW(SV_PLASTIC_H, 3)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_W4Q)
RSPT_TU_X template __global__ void k_trace_w4q<0>(SceneDev, const Quad4Node*, const uint2*, uint32_t, const uint32_t*, const uint32_t*, uint32_t, uint32_t*, const rspt_ray*, uint32_t*, rspt_hit*, uint32_t*, int, int, uint32_t, uint32_t, const float4*);
RSPT_TU_X template __global__ void k_trace_w4q<1>(SceneDev, const Quad4Node*, const uint2*, uint32_t, const uint32_t*, const uint32_t*, uint32_t, uint32_t*, const rspt_ray*, uint32_t*, rspt_hit*, uint32_t*, int, int, uint32_t, uint32_t, const float4*);
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_W4)
RSPT_TU_W4_4(false, 0) RSPT_TU_W4_4(false, 1) RSPT_TU_W4_4(true, 0) RSPT_TU_W4_4(true, 1)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_W4S)
RSPT_TU_W4_S(false, 0) RSPT_TU_W4_S(false, 1) RSPT_TU_W4_S(true, 0) RSPT_TU_W4_S(true, 1)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_W4A)
RSPT_TU_W4A(false, 0) RSPT_TU_W4A(false, 1) RSPT_TU_W4A(true, 0) RSPT_TU_W4A(true, 1)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_W4AM)
RSPT_TU_W4AM(false, 0) RSPT_TU_W4AM(false, 1) RSPT_TU_W4AM(true, 0) RSPT_TU_W4AM(true, 1)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_W4B)
RSPT_TU_W4B(false, 0, 1024, 512) RSPT_TU_W4B(false, 1, 1024, 512) RSPT_TU_W4B(true, 0, 1024, 512) RSPT_TU_W4B(true, 1, 1024, 512)
RSPT_TU_W4B(false, 0, 512, 256) RSPT_TU_W4B(false, 1, 512, 256) RSPT_TU_W4B(true, 0, 512, 256) RSPT_TU_W4B(true, 1, 512, 256)
#endif
#if defined(RSPT_TU_ALL) || defined(RSPT_TU_GROUP_REF)
RSPT_TU_REF8(false, 0) RSPT_TU_REF8(false, 1) RSPT_TU_REF8(true, 0) RSPT_TU_REF8(true, 1)
RSPT_TU_FIX4(false, 0) RSPT_TU_FIX4(false, 1) RSPT_TU_FIX4(true, 0) RSPT_TU_FIX4(true, 1)
RSPT_TU_REFA(false, 0) RSPT_TU_REFA(false, 1) RSPT_TU_REFA(true, 0) RSPT_TU_REFA(true, 1)
#endif

}  // namespace rspt
