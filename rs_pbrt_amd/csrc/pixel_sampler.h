// The PCG-backed pixel samplers on the device: RandomSampler, ZeroTwoSequenceSampler, StratifiedSampler,
// MaxMinDistSampler (src/samplers/*.rs) over one PCG32 stream per 16x16 tile (src/core/rng.rs).  State lives with the lane that
// renders the tile (tile_serial.h); the per-pixel sample vectors live in global memory, element (d, i) of lane l at
// [(d * spp + i) * stride + l].
#pragma once
#include "dev_scene.h"

namespace rspt {

struct PcgRng {  // rng.rs:15-83
    uint64_t state, inc;
    RDEV uint32_t u32() {
        const uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        const uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
        const uint32_t rot = (uint32_t)(old >> 59);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
    }
    RDEV void set_sequence(uint64_t initseq) {
        state = 0;
        inc = (initseq << 1) | 1ull;
        (void)u32();
        state += 0x853c49e6748fea9bULL;
        (void)u32();
    }
    RDEV uint32_t bounded(uint32_t b) {  // Q2: threshold = (!b + 1) & b, the lowest set bit of b
        const uint32_t threshold = (~b + 1u) & b;
        for (;;) {
            const uint32_t r = u32();
            if (r >= threshold) return r % b;
        }
    }
    RDEV float f32() { return fminf((float)u32() * 0x1.0p-32f, RSPT_ONE_MINUS_EPS); }
};

struct PixSampler {
    uint32_t kind, spp, n_dims, nx, ny, jitter;
    const uint32_t* c_pixel;
    float* a1;
    float2* a2;
    uint32_t stride;
    uint32_t cur1, cur2, cur_s;
    PcgRng rng;

    RDEV float& v1(uint32_t d, uint32_t i) const { return a1[((size_t)d * spp + i) * stride]; }
    RDEV float2& v2(uint32_t d, uint32_t i) const { return a2[((size_t)d * spp + i) * stride]; }
    // shuffle(samp, count, 1, rng) (sampling.rs:202-212) on vector d
    RDEV void shuffle1(uint32_t d, uint32_t count) {
        for (uint32_t i = 0; i < count; i++) {
            const uint32_t other = i + rng.bounded(count - i);
            const float t = v1(d, i); v1(d, i) = v1(d, other); v1(d, other) = t;
        }
    }
    RDEV void shuffle2(uint32_t d, uint32_t count) {
        for (uint32_t i = 0; i < count; i++) {
            const uint32_t other = i + rng.bounded(count - i);
            const float2 t = v2(d, i); v2(d, i) = v2(d, other); v2(d, other) = t;
        }
    }
    // van_der_corput(1, spp, ..) (lowdiscrepancy.rs:857-916): Gray-code enumeration of the scrambled radical inverse, one
    // single-element shuffle per pixel sample (each draws from the stream and swaps an element with itself), one over all
    RDEV void van_der_corput(uint32_t d) {
        uint32_t v = rng.u32();
        for (uint32_t i = 0; i < spp; i++) {
            v1(d, i) = fminf((float)v * 0x1.0p-32f, RSPT_ONE_MINUS_EPS);
            v ^= 0x80000000u >> __builtin_ctz(i + 1u);
        }
        for (uint32_t i = 0; i < spp; i++) (void)rng.bounded(1u);
        shuffle1(d, spp);
    }
    // sobol_2d(1, spp, ..) (lowdiscrepancy.rs:920-1010); column k of the second generator matrix is v_k = v_{k-1} ^ (v_{k-1} >> 1)
    RDEV void sobol_2d(uint32_t d) {
        uint32_t x = rng.u32(), y = rng.u32();
        for (uint32_t i = 0; i < spp; i++) {
            v2(d, i) = make_float2(fminf((float)x * 0x1.0p-32f, RSPT_ONE_MINUS_EPS), fminf((float)y * 0x1.0p-32f, RSPT_ONE_MINUS_EPS));
            const uint32_t tz = (uint32_t)__builtin_ctz(i + 1u);
            x ^= 0x80000000u >> tz;
            uint32_t c = 0x80000000u;
            for (uint32_t k = 0; k < tz; k++) c ^= c >> 1;
            y ^= c;
        }
        for (uint32_t i = 0; i < spp; i++) (void)rng.bounded(1u);  // Q3: shuffle(samples, 1, 1) spp times, all on element 0
        shuffle2(d, spp);
    }
    // ---- the 2-D sample ARRAYS an integrator's preprocess asked for (request_2d_array; AOIntegrator: one array of n points per pixel sample,
    // ao.rs:47-49; DirectLightingIntegrator, strategy all: two per light and recursion level, directlighting.rs:54-70).  Array a holds
    // arr_sz[a] points per pixel sample: its flat element e = sample * arr_sz[a] + j lives at arr[(arr_base[a] + e) * stride], arr_base[a] =
    // spp * (sizes before a).  All arrays are refilled by every start_pixel after the vectors, in request order, from the same stream
    // (zerotwosequence.rs:131-148, maxmin.rs:137-152, stratified.rs:137-160, random.rs:64-77) ----
    float2* arr;
    const uint32_t* arr_sz;    // [n_arr]
    const uint32_t* arr_base;  // [n_arr]
    uint32_t n_arr, arr_cur;   // arr_cur: array_2d_offset (get_2d_array hands the arrays out in order, start_next_sample rewinds)
    RDEV float2& va(uint32_t e) const { return arr[(size_t)e * stride]; }
    // get_2d_array_idxs (zerotwosequence.rs:208-218): false when the requested arrays are used up; *first = the pixel sample's slice
    RDEV bool get_2d_array(uint32_t* first, uint32_t* count) {
        if (arr_cur == n_arr) return false;
        *count = arr_sz[arr_cur];
        *first = arr_base[arr_cur] + cur_s * arr_sz[arr_cur];
        arr_cur++;
        return true;
    }
    RDEV void fill_arrays() {
        for (uint32_t a = 0; a < n_arr; a++) fill_array(arr_base[a], arr_sz[a]);
        arr_cur = 0;
    }
    RDEV void fill_array(uint32_t base, uint32_t arr_n) {
        if (!arr_n) return;
        const uint32_t total = arr_n * spp;
        if (kind == RSPT_SAMPLER_ZEROTWO || kind == RSPT_SAMPLER_MAXMINDIST) {  // sobol_2d(arr_n, spp, ..) (lowdiscrepancy.rs:919-1010)
            uint32_t x = rng.u32(), y = rng.u32();
            for (uint32_t i = 0; i < total; i++) {
                va(base + i) = make_float2(fminf((float)x * 0x1.0p-32f, RSPT_ONE_MINUS_EPS), fminf((float)y * 0x1.0p-32f, RSPT_ONE_MINUS_EPS));
                const uint32_t tz = (uint32_t)__builtin_ctz(i + 1u);
                x ^= 0x80000000u >> tz;
                uint32_t c = 0x80000000u;
                for (uint32_t k = 0; k < tz; k++) c ^= c >> 1;
                y ^= c;
            }
            for (uint32_t i = 0; i < spp; i++)   // Q3: shuffle(samples, arr_n, 1) spp times, every time on the FIRST arr_n elements
                for (uint32_t k = 0; k < arr_n; k++) {
                    const uint32_t other = k + rng.bounded(arr_n - k);
                    const float2 t = va(base + k); va(base + k) = va(base + other); va(base + other) = t;
                }
            for (uint32_t i = 0; i < spp; i++) {   // shuffle(samples, spp, arr_n): whole blocks
                const uint32_t other = i + rng.bounded(spp - i);
                for (uint32_t j = 0; j < arr_n; j++) {
                    const float2 t = va(base + arr_n * i + j); va(base + arr_n * i + j) = va(base + arr_n * other + j); va(base + arr_n * other + j) = t;
                }
            }
        } else if (kind == RSPT_SAMPLER_STRATIFIED) {  // latin_hypercube per pixel sample (stratified.rs:150-159, sampling.rs:273-306)
            const float inv_n = 1.0f / (float)arr_n;
            for (uint32_t s = 0; s < spp; s++) {
                const uint32_t b = base + s * arr_n;
                for (uint32_t i = 0; i < arr_n; i++) {
                    const float sx = ((float)i + rng.f32()) * inv_n;
                    const float sy = ((float)i + rng.f32()) * inv_n;
                    va(b + i) = make_float2(fminf(sx, RSPT_ONE_MINUS_EPS), fminf(sy, RSPT_ONE_MINUS_EPS));
                }
                for (uint32_t dim = 0; dim < 2; dim++)
                    for (uint32_t j = 0; j < arr_n; j++) {
                        const uint32_t other = j + rng.bounded(arr_n - j);
                        float2 a = va(b + j), o = va(b + other);
                        if (dim == 0) { const float t = a.x; a.x = o.x; o.x = t; } else { const float t = a.y; a.y = o.y; o.y = t; }
                        va(b + j) = a;
                        if (other != j) va(b + other) = o;
                    }
            }
        } else {  // random.rs:70-76: x first
            for (uint32_t i = 0; i < total; i++) { const float x = rng.f32(); const float y = rng.f32(); va(base + i) = make_float2(x, y); }
        }
    }
    RDEV void start_pixel() {
        start_pixel_vectors();
        fill_arrays();
        cur_s = 0;
    }
    RDEV void start_pixel_vectors() {
        if (kind == RSPT_SAMPLER_ZEROTWO) {  // zerotwosequence.rs:127-163
            for (uint32_t d = 0; d < n_dims; d++) van_der_corput(d);
            for (uint32_t d = 0; d < n_dims; d++) sobol_2d(d);
        } else if (kind == RSPT_SAMPLER_STRATIFIED) {  // stratified.rs:101-161, sampling.rs:237-271
            const uint32_t n = nx * ny;
            const float inv_n = 1.0f / (float)n;
            for (uint32_t d = 0; d < n_dims; d++) {
                for (uint32_t i = 0; i < n; i++) v1(d, i) = fminf(((float)i + (jitter ? rng.f32() : 0.5f)) * inv_n, RSPT_ONE_MINUS_EPS);
                shuffle1(d, n);
            }
            const float dx = 1.0f / (float)nx, dy = 1.0f / (float)ny;
            for (uint32_t d = 0; d < n_dims; d++) {
                uint32_t k = 0;
                for (uint32_t y = 0; y < ny; y++)
                    for (uint32_t x = 0; x < nx; x++) {
                        const float jx = jitter ? rng.f32() : 0.5f;
                        const float jy = jitter ? rng.f32() : 0.5f;
                        v2(d, k++) = make_float2(fminf(((float)x + jx) * dx, RSPT_ONE_MINUS_EPS), fminf(((float)y + jy) * dy, RSPT_ONE_MINUS_EPS));
                    }
                shuffle2(d, n);
            }
        } else if (kind == RSPT_SAMPLER_MAXMINDIST) {  // maxmin.rs:116-159
            const float inv_spp = 1.0f / (float)spp;
            if (n_dims > 0) {
                for (uint32_t i = 0; i < spp; i++) {
                    uint32_t v = 0;  // multiply_generator (lowdiscrepancy.rs:799-814)
                    for (uint32_t a = i, k = 0; a != 0; a >>= 1, k++) if (a & 1u) v ^= c_pixel[k];
                    v2(0, i) = make_float2((float)i * inv_spp, fminf((float)v * 0x1.0p-32f, RSPT_ONE_MINUS_EPS));
                }
                shuffle2(0, spp);
            }
            for (uint32_t d = 0; d < n_dims; d++) van_der_corput(d);
            for (uint32_t d = 1; d < n_dims; d++) sobol_2d(d);
        }
        cur_s = 0;
    }
    RDEV float get_1d() {
        if (kind != RSPT_SAMPLER_RANDOM && cur1 < n_dims) return v1(cur1++, cur_s);
        return rng.f32();
    }
    RDEV f2 get_2d() {
        if (kind == RSPT_SAMPLER_RANDOM) { const float x = rng.f32(); const float y = rng.f32(); return f2{x, y}; }  // random.rs:86-92: x first
        if (cur2 < n_dims) { const float2 v = v2(cur2++, cur_s); return f2{v.x, v.y}; }
        const float y = rng.f32(); const float x = rng.f32();  // Q4: y first (zerotwosequence.rs:178-181, stratified.rs:187-190, maxmin.rs:185-188)
        return f2{x, y};
    }
    RDEV void start_next_sample() { cur1 = 0; cur2 = 0; arr_cur = 0; cur_s += 1; }
};

// what shade_path draws its samples from: the global samplers' dimension stream (k_shade: PIX = false, identical code to before)
// or the tile's pixel sampler
template <bool PIX>
struct ShadeSampler;
template <bool HALTON_POSSIBLE, bool SOBOL_POSSIBLE = true>
struct ShadeSamplerG {
    PathSamplerT<HALTON_POSSIBLE, SOBOL_POSSIBLE> g;
    RDEV void bind(PixSampler*) {}
    RDEV void start(const RenderDev& rd, const uint32_t* __restrict__ tab, uint32_t nd, uint64_t idx, uint32_t first_dim) { g.start(rd, tab, nd, idx, first_dim); }
    RDEV uint32_t dim() const { return g.dim(); }
    RDEV float get_1d(const RenderDev& rd) { return g.get_1d(rd); }
    RDEV f2 get_2d(const RenderDev& rd) { return g.get_2d(rd); }
};
template <>
struct ShadeSampler<false> : ShadeSamplerG<true> {};
template <>
struct ShadeSampler<true> {
    PixSampler* px;
    RDEV void bind(PixSampler* p) { px = p; }
    RDEV void start(const RenderDev&, const uint32_t* __restrict__, uint32_t, uint64_t, uint32_t) {}
    RDEV uint32_t dim() const { return 0u; }
    RDEV float get_1d(const RenderDev&) { return px->get_1d(); }
    RDEV f2 get_2d(const RenderDev&) { return px->get_2d(); }
};
// the sample source of shade_path<PIX, F>: a pixel sampler's stream, or the global samplers with the Halton code only where F has it
template <bool PIX, uint32_t F>
struct ShadeSamplerFor { using type = ShadeSampler<true>; };
template <uint32_t F>
struct ShadeSamplerFor<false, F> { using type = ShadeSamplerG<(F & SF_HALTON) != 0, (F & SF_SOBOL) != 0 || (F & SF_HALTON) == 0>; };

}  // namespace rspt
