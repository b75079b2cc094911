// SamplerIntegrator::render's tile loop (integrator.rs:101-217) for the PCG-backed pixel samplers.
//
// With these samplers a tile is one serial chain: the tile's PCG32 state runs through start_pixel of every pixel (Q9) and through
// every dimension a path draws past the precomputed ones — how many that is depends on where the path ends — so sample k's position
// in the stream is known only when samples 0 .. k-1 of the tile are done.  The chain cannot be cut (PCG's jump-ahead needs the
// offset) and the only parallelism the reference's definition leaves is across tiles.  One lane renders one tile: camera sample,
// PathIntegrator::li by the same shade_path the wavefront kernels use (its pending-estimate state machine runs on the lane's own
// path slot), the reference-order traversal loop for its rays, then the next sample.  Lanes are spread thinly over waves
// (lanes_per_wave, host-chosen: few tiles -> one lane per wave, so that no lane waits for a diverged neighbour).
// Radiance and film position of every sample go to arrays that k_film then splats exactly as it does for the wavefront batches.
#pragma once
#include "dl_serial.h"

namespace rspt {

struct TileRec {
    int16_t x0, y0, x1, y1;   // pixel bounds of the tile
    uint32_t seed;            // tile.y * n_tiles.x + tile.x (integrator.rs:113)
    uint32_t pix0;            // first entry of this tile's pixels in the pass's pixel list
    uint32_t pad;
};
struct PixDesc {              // the sampler's parameters and this render's vectors
    uint32_t kind, spp, n_dims, nx, ny, jitter;
    const uint32_t* c_pixel;
    float* a1;
    float2* a2;
    uint64_t* rng_state;      // [2 * n_tiles]: PCG state / inc between passes over the tile's rows
    float2* arr;              // the integrator's 2-D sample arrays (pixel_sampler.h), nullptr / 0 for `path` and `volpath`
    const uint32_t* arr_sz;
    const uint32_t* arr_base;
    uint32_t n_arr, arr_total;   // arr_total: points per pixel sample over all arrays
    uint32_t ao_cos_sample;
    const int32_t* n_light_samples;   // directlighting, strategy all: per light (device copy), nullptr = 1 each
    uint32_t direct_strategy;
    float4* dl_tex;                   // directlighting over textured materials: texture-stage rows per recursion level (dl_serial.h), [level][row][tile]
    uint32_t dl_tex_rows;
    rspt_mat::Built* dl_dyn;          // ... and, with dynamic materials, the lobe record of every level, [level][tile]
};

// MODE 0: PathIntegrator::li.  1: AOIntegrator::li (ao.rs:50-96) with its sample array from the pixel sampler: closest hit, frame on the true
// geometry, arr_n hemisphere directions from get_2d_array, one any-hit traversal each, the unoccluded terms added in array order.
// 2: VolPathIntegrator::li (vol_serial.h), homogeneous and grid media.  3: DirectLightingIntegrator::li (dl_serial.h).
// 4: PathIntegrator::li over a scene with dynamic materials (shade_path<.., SF_ALL>: lobe lists built per hit, material_assembly.h).
// 5 / 6 / 7 / 8 (round 6): modes 0 / 1 / 2 / 3 over a scene with MOVING object instances (AnimatedTransform primitive_to_world, primitive.rs:198-272): the camera sample's time
// (sampler.rs:88, lerp over the shutter) rides in pb.time[slot] — where shade_path / texture_path read it — and every traversal interpolates the instances it enters.
#ifdef RSPT_TS_WAVES   // A/B: the per-tile kernels built for that many waves per SIMD (what does not fit the budget is spilled)
#define RSPT_TS_ATTR __attribute__((amdgpu_waves_per_eu(RSPT_TS_WAVES, RSPT_TS_WAVES)))
#else
#define RSPT_TS_ATTR
#endif
template <bool INST, bool ALPHA, int MODE = 0>
__global__ __launch_bounds__(64) RSPT_TS_ATTR void k_tile_serial(SceneDev sc, TexTables tt, LightDistDev ld, RenderDev rd, PathBuf pb, PixDesc pd, const TileRec* __restrict__ tiles,
                                                    uint32_t n_tiles, uint32_t lanes_per_wave, int32_t row0, int32_t row1, float4* __restrict__ samp_L,
                                                    float2* __restrict__ samp_pf, uint32_t max_iters, uint32_t* __restrict__ truncated) {
    __shared__ uint32_t stack[RSPT_LDS_STACK * 64];   // 32 levels x the block's 64 columns
    if (threadIdx.x >= lanes_per_wave) return;
    const uint32_t t = blockIdx.x * lanes_per_wave + threadIdx.x;
    if (t >= n_tiles) return;
    const TileRec tr = tiles[t];
    PixSampler px;
    px.kind = pd.kind; px.spp = pd.spp; px.n_dims = pd.n_dims; px.nx = pd.nx; px.ny = pd.ny; px.jitter = pd.jitter; px.c_pixel = pd.c_pixel;
    px.a1 = pd.a1 + t; px.a2 = pd.a2 + t; px.stride = n_tiles;
    px.cur1 = px.cur2 = px.cur_s = 0;
    px.arr = pd.arr ? pd.arr + t : nullptr; px.arr_sz = pd.arr_sz; px.arr_base = pd.arr_base; px.n_arr = pd.n_arr; px.arr_cur = 0;
    if (row0 == 0) px.rng.set_sequence((uint64_t)tr.seed);  // tile_sampler.reseed(seed) (integrator.rs:114)
    else { px.rng.state = pd.rng_state[2 * (size_t)t]; px.rng.inc = pd.rng_state[2 * (size_t)t + 1]; }
    constexpr bool ANIM = MODE >= 5 && MODE <= 8;
    constexpr int M = MODE == 5 ? 0 : (MODE == 6 ? 1 : (MODE == 7 ? 2 : (MODE == 8 ? 3 : MODE)));
    static_assert(!ANIM || INST, "moving instances are instances");
    const uint32_t slot = t;   // the lane's own path slot
    uint32_t k = tr.pix0;
    uint32_t* lds = stack + threadIdx.x;
    for (int32_t y = tr.y0 + row0; y < tr.y0 + row1 && y < tr.y1; y++)
        for (int32_t x = tr.x0; x < tr.x1; x++, k++) {
            px.start_pixel();   // before any bounds test (Q9); pixel_bounds == sample_bounds here (Q16)
            for (uint32_t s = 0; s < pd.spp; s++) {
                // Sampler::get_camera_sample (sampler.rs:85-95)
                const f2 fs = px.get_2d();
                const f2 p_film{(float)x + fs.x, (float)y + fs.y};
                const float time_s = px.get_1d();   // time: a moving camera interpolates its matrix there
                const f2 lens2 = px.get_2d();
                const f3 p_lens{lens2.x, lens2.y, time_s};
                f3 o, d;
                float t_max;
                camera_ray(rd, p_film, p_lens, &o, &d, &t_max);
                store_ray(pb.ray_cont + slot, o, d, t_max, slot);
                pb.L_eta[slot] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);
                pb.beta[slot] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
                pb.state[slot] = ST_ALIVE;
                pb.p_film[slot] = make_float2(p_film.x, p_film.y);
                const float ray_time = ANIM ? rd.shutter_open * (1.0f - time_s) + rd.shutter_close * time_s : 0.0f;   // lerp(sample.time, shutter_open, shutter_close) (perspective.rs:226)
                if (ANIM) pb.time[slot] = ray_time;
                if (M == 3) {
                    DlSerial<INST, ALPHA, PixSampler, ANIM> dl{VolSerial<INST, ALPHA, ANIM>{sc, tt, ld, rd, pb, slot, SerialSampler{&px}, lds, max_iters, false, ray_time}, &px, pd.n_light_samples, pd.direct_strategy == RSPT_DIRECT_SAMPLE_ALL,
                                                         pd.dl_tex ? pd.dl_tex + t : nullptr, n_tiles, pd.dl_tex_rows, p_film, p_lens,
                                                         pd.dl_dyn ? pd.dl_dyn + t : nullptr, n_tiles};
                    const rgb l = dl.li(o, d, t_max);
                    if (dl.base.truncated) atomicAdd(truncated, 1u);
                    const size_t out = (size_t)k * pd.spp + s;
                    samp_L[out] = make_float4(l.r, l.g, l.b, 1.0f);
                    samp_pf[out] = make_float2(p_film.x, p_film.y);
                    px.start_next_sample();
                    continue;
                }
                if (M == 2) {
                    VolSerial<INST, ALPHA, ANIM> vs{sc, tt, ld, rd, pb, slot, SerialSampler{&px}, lds, max_iters, false, ray_time};
                    const rgb l = vs.li(o, d, t_max, p_film, p_lens);
                    if (vs.truncated) atomicAdd(truncated, 1u);
                    const size_t out = (size_t)k * pd.spp + s;
                    samp_L[out] = make_float4(l.r, l.g, l.b, 1.0f);
                    samp_pf[out] = make_float2(p_film.x, p_film.y);
                    px.start_next_sample();
                    continue;
                }
                if (M == 1) {
                    const TraceResult res = serial_trace<false, INST, ALPHA, ANIM>(sc, tt, o, d, t_max, lds, ray_time);
                    float l = 0.0f;
                    if (res.prim != RSPT_MISS) {
                        const TriRec tri = load_tri(sc, res.prim);
                        TexHit h;   // the full interaction: li needs the geometric dpdu
                        tri_fill_tex(sc, res.prim, tri, res.b0, res.b1, res.b2, &h);
                        Hit hp;     // p_error for spawn_ray
                        tri_fill(sc, res.prim, tri, res.b0, res.b1, res.b2, &hp);
                        if (INST && res.inst) {
                            const bool moving = ANIM && sc.inst[res.inst - 1u].anim != RSPT_MISS;
                            InstDev moved;
                            if (moving) moved = inst_at(sc, res.inst - 1u, ray_time);
                            const InstDev& in = moving ? moved : sc.inst[res.inst - 1u];
                            if (!in.identity) { inst_texhit(in, &h); inst_hit(in, &hp); }
                        }
                        const f3 n = faceforward(h.n, -d);
                        const f3 sv = normalize(h.dpdu);
                        const f3 tv = cross(h.n, sv);  // nrm_cross_vec3(&isect.common.n, &s)
                        uint32_t first = 0, arr_n = 0;
                        const bool have = px.get_2d_array(&first, &arr_n);   // ao.rs:75: the pixel sample's slice of the array preprocess requested
                        for (uint32_t j = 0; have && j < arr_n; j++) {
                            const float2 uu = px.va(first + j);
                            const f2 u{uu.x, uu.y};
                            f3 wi;
                            float pdf;
                            if (pd.ao_cos_sample) { wi = cosine_hemisphere(u); pdf = fabsf(wi.z) * RSPT_INV_PI; }
                            else {  // uniform_sample_hemisphere (sampling.rs:309-318)
                                const float z = u.x, r = sqrtf(fmaxf(0.0f, 1.0f - z * z)), phi = 2.0f * RSPT_PI * u.y;
                                wi = f3{r * rspt_cosf(phi), r * rspt_sinf(phi), z};
                                pdf = 0.15915494309189533577f;
                            }
                            wi = f3{sv.x * wi.x + tv.x * wi.y + n.x * wi.z, sv.y * wi.x + tv.y * wi.y + n.y * wi.z, sv.z * wi.x + tv.z * wi.y + n.z * wi.z};
                            if (pdf != 0.0f) {
                                const TraceResult occ = serial_trace<true, INST, ALPHA, ANIM>(sc, tt, offset_ray_origin(hp.p, hp.p_err, hp.n, wi), wi, RSPT_INF, lds, ray_time);
                                if (occ.prim == RSPT_MISS) l += dot(wi, n) / (pdf * (float)arr_n);
                            }
                        }
                    }
                    const size_t out = (size_t)k * pd.spp + s;
                    samp_L[out] = make_float4(l, l, l, 1.0f);
                    samp_pf[out] = make_float2(p_film.x, p_film.y);
                    px.start_next_sample();
                    continue;
                }
                ShadeOut so{true, true, false, false};
                for (uint32_t it = 0; so.active; it++) {
                    if (it >= max_iters) { atomicAdd(truncated, 1u); break; }   // the reference's loop over BSDF-less surfaces has no limit (path.rs:109-116)
                    if (so.cont) {
                        const float4* rp = reinterpret_cast<const float4*>(pb.ray_cont + slot);
                        const float4 r0 = rp[0], r1 = rp[1];
                        const TraceResult res = serial_trace<false, INST, ALPHA, ANIM>(sc, tt, f3{r0.x, r0.y, r0.z}, f3{r0.w, r1.x, r1.y}, r1.z, lds, ray_time);
                        pb.hit_cont[slot] = make_float4(__uint_as_float(res.prim), res.b0, res.b1, res.b2);
                        if (INST && pb.hit_inst) pb.hit_inst[slot] = res.inst;
                    }
                    if (so.mis) {
                        const float4* rp = reinterpret_cast<const float4*>(pb.ray_mis + slot);
                        const float4 r0 = rp[0], r1 = rp[1];
                        const TraceResult res = serial_trace<false, INST, ALPHA, ANIM>(sc, tt, f3{r0.x, r0.y, r0.z}, f3{r0.w, r1.x, r1.y}, r1.z, lds, ray_time);
                        pb.hit_mis[slot] = make_float4(__uint_as_float(res.prim), res.b0, res.b1, res.b2);
                    }
                    if (so.shadow) {
                        const float4* rp = reinterpret_cast<const float4*>(pb.ray_sh + slot);
                        const float4 r0 = rp[0], r1 = rp[1];
                        const TraceResult res = serial_trace<true, INST, ALPHA, ANIM>(sc, tt, f3{r0.x, r0.y, r0.z}, f3{r0.w, r1.x, r1.y}, r1.z, lds, ray_time);
                        pb.occluded[slot] = res.prim != RSPT_MISS ? 1u : 0u;
                    }
                    if (sc.mat_flags && so.cont) texture_path(sc, tt, rd, pb, slot, &p_lens);   // the texture stage k_texture runs in front of k_shade
                    so = shade_path<true, MODE == 4 ? (SF_ALL & ~SF_ANIM) : (ANIM ? (SF_ALL & ~SF_DYNAMIC) : (SF_ALL & ~SF_DYNAMIC & ~SF_ANIM))>(sc, ld, rd, pb, slot, nullptr, nullptr, 0u, &px);
                }
                const size_t out = (size_t)k * pd.spp + s;
                samp_L[out] = pb.L_eta[slot];
                samp_pf[out] = make_float2(p_film.x, p_film.y);
                px.start_next_sample();
            }
            px.cur_s = 0;
        }
    pd.rng_state[2 * (size_t)t] = px.rng.state;
    pd.rng_state[2 * (size_t)t + 1] = px.rng.inc;
}

}  // namespace rspt
