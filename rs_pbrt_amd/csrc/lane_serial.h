// DirectLightingIntegrator::li under the Sobol' / Halton samplers with ONE LANE PER CAMERA SAMPLE (dl_serial.h's DlSerial over a sampler
// that hands out the global samplers' dimensions in program order).  rspt_render takes this form when the wavefront form (direct.h)
// cannot serve the render: textured materials (the specular bounces carry ray differentials for them, directlighting.rs:150-191),
// max_depth > 8 (the wavefront form keeps 2^max_depth node slots per camera sample), or a material with several specular lobes of one
// kind (there the lobe choice depends on a sample value, and the specular tree cannot be traced ahead of the dimension assignment).
// Independent samples, so the parallelism is the batch's sample count; what a lane pays for is divergence inside its wave.
#pragma once
#include "dl_serial.h"

namespace rspt {

// GlobalSampler read in program order (sobol.rs:180-236 / halton.rs, sampler.rs:96-140): get_1d / get_2d jump over the dimensions that
// start_pixel spent on the 2-D sample arrays ([5, 5 + 2 n_arr)); array a's element k of pixel sample s is dimensions 5 + 2a, 5 + 2a + 1
// of the sample with index get_index_for_sample(s * n + k) — computed when it is read instead of stored per pixel.
struct LaneSampler {
    const RenderDev& rd;
    uint64_t index;          // get_index_for_sample(current pixel sample)
    uint32_t dim;            // next dimension of the regular stream
    uint32_t arr_end;        // 5 + 2 n_arr
    int32_t px, py;          // current pixel
    uint32_t sample_num;     // current_pixel_sample_index
    const int32_t* nls;      // array sizes: arrays 2 (L * n_lights + j), + 1 have n_light_samples[j] points per pixel sample (nullptr: 1)
    uint32_t n_lights, n_arr, arr_cur;
    uint32_t dim_limit;      // the sampler's dimension count: the reference panics past it (sobol.rs:119-124)
    bool overflow;

    RDEV float dimv(uint64_t idx, uint32_t d) {
        if (d >= dim_limit) { overflow = true; return 0.0f; }
        return rd.sampler_kind == RSPT_SAMPLER_HALTON ? halton_dim(rd, idx, d) : sobol_dim(rd, idx, d);
    }
    RDEV float get_1d() {
        if (dim >= 5u && dim < arr_end) dim = arr_end;
        return dimv(index, dim++);
    }
    RDEV f2 get_2d() {
        if (dim + 1u >= 5u && dim < arr_end) dim = arr_end;
        const float y = dimv(index, dim + 1u), x = dimv(index, dim);
        dim += 2u;
        return f2{x, y};
    }
    RDEV uint32_t arr_size(uint32_t a) const { return nls ? (uint32_t)nls[(a >> 1) % n_lights] : 1u; }
    // get_2d_array_idxs: the handle of the next array's slice for this pixel sample (array << 16; sizes are at most 4096)
    RDEV bool get_2d_array(uint32_t* first, uint32_t* count) {
        if (arr_cur == n_arr) return false;
        *first = arr_cur << 16; *count = arr_size(arr_cur);
        arr_cur++;
        return true;
    }
    RDEV float2 va(uint32_t h) {   // get_2d_sample(array, start + k)
        const uint32_t a = h >> 16, k = h & 0xffffu;
        const uint64_t elem = (uint64_t)sample_num * arr_size(a) + k;
        const uint64_t ei = rd.sampler_kind == RSPT_SAMPLER_HALTON ? halton_index(rd, px, py, elem)
                                                                  : sobol_interval_to_index(rd, (uint32_t)rd.log2_res, elem, px - rd.sample_bounds[0], py - rd.sample_bounds[1]);
        const float x = dimv(ei, 5u + 2u * a), y = dimv(ei, 5u + 2u * a + 1u);
        return make_float2(x, y);
    }
};

struct LaneDesc {
    const int32_t* n_light_samples;   // device copy, nullptr = 1 each
    uint32_t n_arr;                   // 2 * max_depth * n_lights with strategy all, else 0
    uint32_t sample_all;
    uint32_t dim_limit;
    float4* tex;                      // [level][row][lane] texture-stage rows, nullptr = no textured material
    uint32_t tex_stride, tex_rows;
    rspt_mat::Built* dyn;             // [level][lane] lobe records of dynamic materials, nullptr = none
    uint32_t max_walk;
    uint32_t* error;                  // 2: a camera sample drew past the sampler's dimensions
    uint32_t* truncated;
};

// k_raygen has left the camera ray, the sample's index and film position in slot i
// ANIM (round 6): moving instances — the sample's ray time is what k_raygen left in pb.time[i]
template <bool INST, bool ALPHA, bool ANIM = false>
__global__ __launch_bounds__(64) void k_lane_dl(SceneDev sc, TexTables tt, LightDistDev ld, RenderDev rd, Batch bt, PathBuf pb, const uint32_t* __restrict__ pix_list, LaneDesc ln) {
    __shared__ uint32_t stack[RSPT_LDS_STACK * 64];
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= bt.n) return;
    const uint32_t pk = pix_list[bt.pix0 + i / bt.ns];
    const float4* rp = reinterpret_cast<const float4*>(pb.ray_cont + i);
    const float4 r0 = rp[0], r1 = rp[1];
    const uint64_t index = pb.sobol_index[i];
    LaneSampler smp{rd, index, 5u, 5u + 2u * ln.n_arr, (int32_t)(int16_t)(pk & 0xffffu), (int32_t)(int16_t)(pk >> 16), bt.s0 + i % bt.ns,
                    ln.n_light_samples, sc.n_lights, ln.n_arr, 0u, ln.dim_limit, false};
    const float2 pf = pb.p_film[i];
    f3 p_lens{0.0f, 0.0f, 0.0f};
    if (rd.lens_radius > 0.0f) { p_lens.x = smp.dimv(index, 3u); p_lens.y = smp.dimv(index, 4u); }
    if (rd.cam_anim) p_lens.z = smp.dimv(index, 2u);
    DlSerial<INST, ALPHA, LaneSampler, ANIM> dl{VolSerial<INST, ALPHA, ANIM>{sc, tt, ld, rd, pb, i, SerialSampler{nullptr}, stack + threadIdx.x, ln.max_walk, false, (ANIM && pb.time) ? pb.time[i] : 0.0f}, &smp, ln.n_light_samples,
                                          ln.sample_all != 0u, ln.tex ? ln.tex + i : nullptr, ln.tex_stride, ln.tex_rows, f2{pf.x, pf.y}, p_lens,
                                          ln.dyn ? ln.dyn + i : nullptr, ln.tex_stride};
    const rgb l = dl.li(f3{r0.x, r0.y, r0.z}, f3{r0.w, r1.x, r1.y}, r1.z);
    pb.L_eta[i] = make_float4(l.r, l.g, l.b, 1.0f);
    if (dl.base.truncated) atomicAdd(ln.truncated, 1u);
    if (smp.overflow) atomicMax(ln.error, 2u);
}

}  // namespace rspt
