// TEST INFRASTRUCTURE — CPU oracle (see orc_math.hpp header).  Sampler, lights, light
// distributions, PathIntegrator::li, film, tile render loop.
#pragma once
#include <atomic>
#include <chrono>
#include <map>
#include <memory>
#include <mutex>
#include <thread>

#include "orc_material.hpp"
#include "orc_animated.hpp"

namespace orc {

// ---- Sobol' sampler: src/samplers/sobol.rs, src/core/lowdiscrepancy.rs:1014-1076 ----
struct SobolTables { const uint32_t* sobol32; const uint64_t* vdc; const uint64_t* vdc_inv; };

static inline uint64_t sobol_interval_to_index(const SobolTables& T, uint32_t m, uint64_t frame, int32_t px, int32_t py) {
    if (m == 0) return 0;
    uint32_t m2 = m << 1;
    uint64_t index = frame << m2;
    uint64_t delta = 0;
    for (int c = 0; frame > 0; frame >>= 1, c++)
        if (frame & 1) delta ^= T.vdc[(m - 1) * 52 + c];
    uint64_t b = ((uint64_t)((uint32_t)px << m) | (uint64_t)(int64_t)py) ^ delta;
    for (int c = 0; b > 0; b >>= 1, c++)
        if (b & 1) index ^= T.vdc_inv[(m - 1) * 52 + c];
    return index;
}
static inline Float sobol_sample_float(const SobolTables& T, int64_t a, int dimension, uint32_t scramble) {
    uint32_t v = scramble;
    for (size_t i = (size_t)dimension * 52; a != 0; a >>= 1, i++)
        if (a & 1) v ^= T.sobol32[i];
    return std::fmin((Float)v * 0x1.0p-32f, FLOAT_ONE_MINUS_EPSILON);
}
static inline int32_t round_up_pow2_32(int32_t v) { // pbrt.rs:188-198
    v -= 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1;
}
static inline int log2_int(uint32_t v) { return 31 - __builtin_clz(v); } // pbrt.rs:160-163

struct SobolSampler {
    SobolTables T;
    int64_t spp;
    int32_t sb[4]; // sample_bounds x0,y0,x1,y1
    int32_t resolution, log2_res;
    int64_t dimension = 0;
    uint64_t interval_sample_index = 0;
    int32_t px = 0, py = 0;
    int64_t cur_sample = 0;
    SobolSampler(const SobolTables& t, int64_t spp_, const int32_t sb_[4]) : T(t), spp(spp_) {
        for (int i = 0; i < 4; i++) sb[i] = sb_[i];
        resolution = round_up_pow2_32(std::max(sb[2] - sb[0], sb[3] - sb[1])); // sobol.rs:46-47
        log2_res = log2_int((uint32_t)resolution);
    }
    uint64_t get_index_for_sample(uint64_t n) const { // sobol.rs:110-117
        return sobol_interval_to_index(T, (uint32_t)log2_res, n, px - sb[0], py - sb[1]);
    }
    Float sample_dimension(uint64_t index, int64_t dim) const { // sobol.rs:118-140
        Float s = sobol_sample_float(T, (int64_t)index, (int)dim, 0);
        if (dim == 0 || dim == 1) {
            s = s * (Float)resolution + (Float)sb[dim];
            s = clamp_t(s - (Float)(dim == 0 ? px : py), 0.0f, FLOAT_ONE_MINUS_EPSILON);
        }
        return s;
    }
    void start_pixel(int32_t x, int32_t y) { px = x; py = y; cur_sample = 0; dimension = 0; interval_sample_index = get_index_for_sample(0); }
    // array_start_dim == array_end_dim == 5 for `path` (Appendix B): the skip tests never fire
    Float get_1d() { Float r = sample_dimension(interval_sample_index, dimension); dimension += 1; return r; }
    P2 get_2d() { // sobol.rs:190-201: y first
        Float y = sample_dimension(interval_sample_index, dimension + 1);
        Float x = sample_dimension(interval_sample_index, dimension);
        dimension += 2;
        return P2{x, y};
    }
    bool start_next_sample() { // sobol.rs:232-241
        dimension = 0;
        interval_sample_index = get_index_for_sample((uint64_t)cur_sample + 1);
        cur_sample += 1;
        return cur_sample < spp;
    }
};

// (HaltonSampler and the sampler facade follow the radical-inverse helpers below)
// ---- radical inverse: src/core/lowdiscrepancy.rs:770-787,1082-1096,1126-1135 ----
static inline uint32_t reverse_bits_32(uint32_t n) {
    n = (n << 16) | (n >> 16);
    n = ((n & 0x00ff00ffu) << 8) | ((n & 0xff00ff00u) >> 8);
    n = ((n & 0x0f0f0f0fu) << 4) | ((n & 0xf0f0f0f0u) >> 4);
    n = ((n & 0x33333333u) << 2) | ((n & 0xccccccccu) >> 2);
    n = ((n & 0x55555555u) << 1) | ((n & 0xaaaaaaaau) >> 1);
    return n;
}
static inline uint64_t reverse_bits_64(uint64_t n) {
    uint64_t n0 = reverse_bits_32((uint32_t)n), n1 = reverse_bits_32((uint32_t)(n >> 32));
    return (n0 << 32) | n1;
}
// PRIMES / PRIME_SUMS (lowdiscrepancy.rs:31-760): the first 1000 primes and their prefix sums
struct PrimeTables {
    std::vector<uint32_t> primes, sums;
    PrimeTables() {
        for (uint32_t v = 2; primes.size() < 1000; v++) {
            bool is_p = true;
            for (uint32_t d = 2; d * d <= v; d++) if (v % d == 0) { is_p = false; break; }
            if (is_p) primes.push_back(v);
        }
        uint32_t acc = 0;
        for (uint32_t p : primes) { sums.push_back(acc); acc += p; }
    }
};
static inline const PrimeTables& prime_tables() { static PrimeTables t; return t; }
// radical_inverse: lowdiscrepancy.rs:1126-2160 (base 2 via bit reversal, others radical_inverse_specialized :1082-1096)
static inline Float radical_inverse(int base_index, uint64_t a) {
    if (base_index == 0) return (Float)reverse_bits_64(a) * 0x1.0p-64f;
    uint64_t base = prime_tables().primes[base_index];
    Float inv_base = 1.0f / (Float)base;
    uint64_t reversed = 0;
    Float inv_base_n = 1.0f;
    while (a != 0) {
        uint64_t next = a / base, digit = a - next * base;
        reversed = reversed * base + digit;
        inv_base_n *= inv_base;
        a = next;
    }
    return std::fmin((Float)reversed * inv_base_n, FLOAT_ONE_MINUS_EPSILON);
}
// scrambled_radical_inverse_specialized: lowdiscrepancy.rs:1101-1122
static inline Float scrambled_radical_inverse(int base_index, uint64_t a, const uint16_t* perm) {
    uint64_t base = prime_tables().primes[base_index];
    Float inv_base = 1.0f / (Float)base;
    uint64_t reversed = 0;
    Float inv_base_n = 1.0f;
    while (a != 0) {
        uint64_t next = a / base, digit = a - next * base;
        reversed = reversed * base + perm[digit];
        inv_base_n *= inv_base;
        a = next;
    }
    return std::fmin(inv_base_n * ((Float)reversed + inv_base * (Float)perm[0] / (1.0f - inv_base)), FLOAT_ONE_MINUS_EPSILON);
}
// inverse_radical_inverse: lowdiscrepancy.rs:788-797
static inline uint64_t inverse_radical_inverse(uint64_t base, uint64_t inverse, uint64_t n_digits) {
    uint64_t index = 0;
    for (uint64_t i = 0; i < n_digits; i++) {
        uint64_t digit = inverse % base;
        inverse /= base;
        index = index * base + digit;
    }
    return index;
}
// PCG32: src/core/rng.rs:15-83 (default state and stream, no set_sequence: halton.rs:19-26)
struct Rng {
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    uint32_t uniform_uint32() {
        uint64_t old = state;
        state = old * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
        uint32_t rot = (uint32_t)(old >> 59);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    void set_sequence(uint64_t initseq) { // rng.rs:31-39
        state = 0;
        inc = (initseq << 1) | 1;
        uniform_uint32();
        state += 0x853c49e6748fea9bULL;
        uniform_uint32();
    }
    Float uniform_float() { return std::fmin((Float)uniform_uint32() * 0x1.0p-32f, FLOAT_ONE_MINUS_EPSILON); } // rng.rs:74-82
    uint32_t uniform_uint32_bounded(uint32_t b) { // Q2: threshold = lowest set bit of b, not (-b) % b
        uint32_t threshold = (~b + 1u) & b;
        for (;;) {
            uint32_t r = uniform_uint32();
            if (r >= threshold) return r % b;
        }
    }
};
// compute_radical_inverse_permutations (lowdiscrepancy.rs:2165-2187) with shuffle (sampling.rs:200-212),
// for the first n_dims primes (the RNG stream is sequential over the primes, so a prefix is exact)
static inline std::vector<uint16_t> radical_inverse_permutations(int n_dims) {
    const PrimeTables& pt = prime_tables();
    std::vector<uint16_t> perms;
    Rng rng;
    for (int i = 0; i < n_dims; i++) {
        size_t p0 = perms.size();
        uint32_t prime = pt.primes[i];
        for (uint32_t j = 0; j < prime; j++) perms.push_back((uint16_t)j);
        for (int32_t k = 0; k < (int32_t)prime; k++) {
            int32_t other = k + (int32_t)rng.uniform_uint32_bounded((uint32_t)((int32_t)prime - k));
            std::swap(perms[p0 + k], perms[p0 + other]);
        }
    }
    return perms;
}

// ---- HaltonSampler: src/samplers/halton.rs ----
static inline int64_t mod_t(int64_t a, int64_t b) { int64_t r = a - (a / b) * b; return r < 0 ? r + b : r; } // pbrt.rs:127-135
static inline void extended_gcd(uint64_t a, uint64_t b, int64_t* x, int64_t* y) { // halton.rs:39-52
    if (b == 0) { *x = 1; *y = 0; return; }
    int64_t d = (int64_t)a / (int64_t)b, xp = 0, yp = 0;
    extended_gcd(b, a % b, &xp, &yp);
    *x = yp; *y = xp - (d * yp);
}
static inline uint64_t multiplicative_inverse(int64_t a, int64_t n) { int64_t x = 0, y = 0; extended_gcd((uint64_t)a, (uint64_t)n, &x, &y); return (uint64_t)mod_t(x, n); }
struct HaltonSampler {
    static constexpr int32_t K_MAX_RESOLUTION = 128;
    int64_t spp;
    int32_t base_scales[2], base_exponents[2];
    uint64_t sample_stride, mult_inverse[2];
    bool sample_at_pixel_center;
    const uint16_t* perms; uint64_t n_perms;
    int64_t dimension = 0; uint64_t interval_sample_index = 0; int32_t px = 0, py = 0; int64_t cur_sample = 0;
    HaltonSampler(int64_t spp_, const int32_t sb[4], bool center, const uint16_t* perms_, uint64_t n_perms_)
        : spp(spp_), sample_at_pixel_center(center), perms(perms_), n_perms(n_perms_) { // halton.rs:80-131
        int32_t res[2] = {sb[2] - sb[0], sb[3] - sb[1]};
        for (int i = 0; i < 2; i++) {
            int32_t base = i == 0 ? 2 : 3, scale = 1, exp = 0;
            while (scale < std::min(res[i], K_MAX_RESOLUTION)) { scale *= base; exp += 1; }
            base_scales[i] = scale; base_exponents[i] = exp;
        }
        sample_stride = (uint64_t)base_scales[0] * (uint64_t)base_scales[1];
        mult_inverse[0] = multiplicative_inverse(base_scales[1], base_scales[0]);
        mult_inverse[1] = multiplicative_inverse(base_scales[0], base_scales[1]);
    }
    uint64_t get_index_for_sample(uint64_t sample_num) const { // halton.rs:173-214 (the per-pixel cache recomputed)
        uint64_t offset = 0;
        if (sample_stride > 1) {
            int64_t pm[2] = {mod_t(px, K_MAX_RESOLUTION), mod_t(py, K_MAX_RESOLUTION)};
            for (int i = 0; i < 2; i++) {
                uint64_t dim_offset = inverse_radical_inverse(i == 0 ? 2 : 3, (uint64_t)pm[i], (uint64_t)base_exponents[i]);
                offset += dim_offset * (sample_stride / (uint64_t)base_scales[i]) * mult_inverse[i];
            }
            offset %= sample_stride;
        }
        return offset + sample_num * sample_stride;
    }
    Float sample_dimension(uint64_t index, int64_t dim) const { // halton.rs:215-226
        if (sample_at_pixel_center && (dim == 0 || dim == 1)) return 0.5f;
        if (dim == 0) return radical_inverse(0, index >> (uint64_t)base_exponents[0]);
        if (dim == 1) return radical_inverse(1, index / (uint64_t)base_scales[1]);
        return scrambled_radical_inverse((int)dim, index, perms + prime_tables().sums[dim]);
    }
};

// Sampler facade (src/core/sampler.rs:18-203) over the two GlobalSamplers in scope
// ---- the pixel samplers: src/samplers/{random,zerotwosequence,stratified,maxmin}.rs ----
// sampling.rs:202-212
template <class T>
static inline void shuffle(T* samp, int32_t count, int32_t n_dimensions, Rng& rng) {
    for (int32_t i = 0; i < count; i++) {
        int32_t other = i + (int32_t)rng.uniform_uint32_bounded((uint32_t)(count - i));
        for (int32_t j = 0; j < n_dimensions; j++) std::swap(samp[n_dimensions * i + j], samp[n_dimensions * other + j]);
    }
}
static inline uint32_t c_van_der_corput(int i) { return 0x80000000u >> i; } // lowdiscrepancy.rs:864-897: the identity matrix, reversed
static inline uint32_t c_sobol1(int i) { // lowdiscrepancy.rs:959-992: column i of the second Sobol' generator matrix = row i of Pascal's triangle mod 2, bit-reversed
    uint32_t v = 0x80000000u;            // v_0; v_{i} = v_{i-1} ^ (v_{i-1} >> 1)
    for (int k = 0; k < i; k++) v ^= v >> 1;
    return v;
}
// lowdiscrepancy.rs:857-916
static inline void van_der_corput(int32_t n_per, int32_t n_pixel_samples, Float* samples, Rng& rng) {
    const uint32_t scramble = rng.uniform_uint32();
    const int32_t total = n_per * n_pixel_samples;
    uint32_t v = scramble; // gray_code_sample_1d :824-835
    for (int32_t i = 0; i < total; i++) {
        samples[i] = std::fmin((Float)v * 0x1.0p-32f, FLOAT_ONE_MINUS_EPSILON);
        v ^= c_van_der_corput(__builtin_ctz((uint32_t)(i + 1)));
    }
    for (int32_t i = 0; i < n_pixel_samples; i++) shuffle(samples + (size_t)i * n_per, n_per, 1, rng);
    shuffle(samples, n_pixel_samples, n_per, rng);
}
// lowdiscrepancy.rs:920-1010.  Q3: the per-sample shuffles all start at samples[0]
static inline void sobol_2d(int32_t n_per, int32_t n_pixel_samples, P2* samples, Rng& rng) {
    uint32_t v0 = rng.uniform_uint32(), v1 = rng.uniform_uint32(); // scramble.x, scramble.y
    const int32_t total = n_per * n_pixel_samples;
    for (int32_t i = 0; i < total; i++) { // gray_code_sample_2d :840-853
        samples[i].x = std::fmin((Float)v0 * 0x1.0p-32f, FLOAT_ONE_MINUS_EPSILON);
        samples[i].y = std::fmin((Float)v1 * 0x1.0p-32f, FLOAT_ONE_MINUS_EPSILON);
        const int tz = __builtin_ctz((uint32_t)(i + 1));
        v0 ^= c_van_der_corput(tz);
        v1 ^= c_sobol1(tz);
    }
    for (int32_t i = 0; i < n_pixel_samples; i++) shuffle(samples, n_per, 1, rng);
    shuffle(samples, n_pixel_samples, n_per, rng);
}
// sampling.rs:237-271
static inline void stratified_sample_1d(Float* samp, int32_t n, Rng& rng, bool jitter) {
    const Float inv_n = 1.0f / (Float)n;
    for (int32_t i = 0; i < n; i++) {
        const Float delta = jitter ? rng.uniform_float() : 0.5f;
        samp[i] = std::fmin(((Float)i + delta) * inv_n, FLOAT_ONE_MINUS_EPSILON);
    }
}
static inline void stratified_sample_2d(P2* samp, int32_t nx, int32_t ny, Rng& rng, bool jitter) {
    const Float dx = 1.0f / (Float)nx, dy = 1.0f / (Float)ny;
    size_t k = 0;
    for (int32_t y = 0; y < ny; y++)
        for (int32_t x = 0; x < nx; x++) {
            const Float jx = jitter ? rng.uniform_float() : 0.5f;
            const Float jy = jitter ? rng.uniform_float() : 0.5f;
            samp[k].x = std::fmin(((Float)x + jx) * dx, FLOAT_ONE_MINUS_EPSILON);
            samp[k].y = std::fmin(((Float)y + jy) * dy, FLOAT_ONE_MINUS_EPSILON);
            k++;
        }
}
static inline uint32_t multiply_generator(const uint32_t* c, uint32_t a) { // lowdiscrepancy.rs:799-814
    uint32_t v = 0;
    for (int i = 0; a != 0; i++, a >>= 1) if (a & 1u) v ^= c[i];
    return v;
}
// sampling.rs:273-306 (the two-dimensional instance the stratified sampler's arrays use)
static inline void latin_hypercube(P2* samples, uint32_t n_samples, Rng& rng) {
    const Float inv_n_samples = 1.0f / (Float)n_samples;
    for (uint32_t i = 0; i < n_samples; i++) {
        const Float sx = ((Float)i + rng.uniform_float()) * inv_n_samples;
        samples[i].x = std::fmin(sx, FLOAT_ONE_MINUS_EPSILON);
        const Float sy = ((Float)i + rng.uniform_float()) * inv_n_samples;
        samples[i].y = std::fmin(sy, FLOAT_ONE_MINUS_EPSILON);
    }
    for (int dim = 0; dim < 2; dim++)
        for (uint32_t j = 0; j < n_samples; j++) {
            const uint32_t other = j + rng.uniform_uint32_bounded(n_samples - j);
            if (dim == 0) std::swap(samples[j].x, samples[other].x); else std::swap(samples[j].y, samples[other].y);
        }
}

struct PixelSampler {
    int kind = 0;
    int64_t spp = 1;
    int32_t n_dims = 4, nx = 1, ny = 1;
    bool jitter = true;
    const uint32_t* c_pixel = nullptr;
    std::vector<std::vector<Float>> samples_1d;
    std::vector<std::vector<P2>> samples_2d;
    int32_t current_1d_dimension = 0, current_2d_dimension = 0;
    int64_t cur_sample = 0;
    Rng rng;
    // 2-D sample arrays an integrator's preprocess requested (ao, directlighting): n x spp points each, refilled by every start_pixel
    // AFTER the plain vectors, from the same stream (zerotwosequence.rs:131-148, maxmin.rs:137-152, stratified.rs:137-160, random.rs:64-77)
    std::vector<int32_t> samples_2d_array_sizes;
    std::vector<std::vector<P2>> sample_array_2d;
    void request_2d_array(int32_t n) { samples_2d_array_sizes.push_back(n); sample_array_2d.emplace_back((size_t)n * (size_t)spp); } // (asserts round_count(n) == n, e.g. zerotwosequence.rs:187-193)
    int32_t round_count(int32_t n) const { // zerotwosequence.rs:194, maxmin.rs:198: round_up_pow2_32; stratified.rs:200, random.rs:107: the identity
        if (kind != RSPT_SAMPLER_ZEROTWO && kind != RSPT_SAMPLER_MAXMINDIST) return n;
        int32_t v = n - 1; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; return v + 1; // pbrt.rs round_up_pow2_32
    }
    void fill_arrays() {
        const int32_t n = (int32_t)spp;
        for (size_t i = 0; i < sample_array_2d.size(); i++) {
            const int32_t count = samples_2d_array_sizes[i];
            P2* a = sample_array_2d[i].data();
            if (kind == RSPT_SAMPLER_ZEROTWO || kind == RSPT_SAMPLER_MAXMINDIST) sobol_2d(count, n, a, rng);
            else if (kind == RSPT_SAMPLER_STRATIFIED) { for (int64_t j = 0; j < spp; j++) latin_hypercube(a + (size_t)j * (size_t)count, (uint32_t)count, rng); }
            else for (size_t j = 0; j < sample_array_2d[i].size(); j++) { const Float x = rng.uniform_float(); const Float y = rng.uniform_float(); a[j] = P2{x, y}; } // random.rs:70-76: x first
        }
    }
    void init(const rspt_render_desc& rd) {
        kind = (int)rd.sampler_kind; spp = rd.spp; n_dims = (int32_t)rd.pixel_dimensions;
        nx = (int32_t)rd.strat_x; ny = (int32_t)rd.strat_y; jitter = rd.strat_jitter != 0; c_pixel = rd.maxmin_c_pixel;
        rng.state = 0; rng.inc = 0; // Rng::default() (derive(Default)): the samplers' `new` do not call Rng::new()
        if (kind != RSPT_SAMPLER_RANDOM) { // RandomSampler keeps no per-dimension vectors (random.rs:10-22)
            samples_1d.assign((size_t)n_dims, std::vector<Float>((size_t)spp));
            samples_2d.assign((size_t)n_dims, std::vector<P2>((size_t)spp));
        }
    }
    void reseed(uint64_t seed) { rng.set_sequence(seed); }
    void start_pixel() {
        const int32_t n = (int32_t)spp;
        if (kind == RSPT_SAMPLER_ZEROTWO) { // zerotwosequence.rs:127-163
            for (auto& v : samples_1d) van_der_corput(1, n, v.data(), rng);
            for (auto& v : samples_2d) sobol_2d(1, n, v.data(), rng);
        } else if (kind == RSPT_SAMPLER_STRATIFIED) { // stratified.rs:101-161
            for (auto& v : samples_1d) { stratified_sample_1d(v.data(), nx * ny, rng, jitter); shuffle(v.data(), nx * ny, 1, rng); }
            for (auto& v : samples_2d) { stratified_sample_2d(v.data(), nx, ny, rng, jitter); shuffle(v.data(), nx * ny, 1, rng); }
        } else if (kind == RSPT_SAMPLER_MAXMINDIST) { // maxmin.rs:116-159
            const Float inv_spp = 1.0f / (Float)spp;
            if (!samples_2d.empty()) {
                for (int32_t i = 0; i < n; i++)
                    samples_2d[0][(size_t)i] = P2{(Float)i * inv_spp, std::fmin((Float)(multiply_generator(c_pixel, (uint32_t)i) ^ 0u) * 0x1.0p-32f, FLOAT_ONE_MINUS_EPSILON)};
                shuffle(samples_2d[0].data(), n, 1, rng);
            }
            for (auto& v : samples_1d) van_der_corput(1, n, v.data(), rng);
            for (size_t i = 1; i < samples_2d.size(); i++) sobol_2d(1, n, samples_2d[i].data(), rng);
        }
        fill_arrays();
        cur_sample = 0; // (current_*_dimension are reset by start_next_sample only; they are 0 here: every pixel ends with one)
    }
    Float get_1d() {
        if (kind != RSPT_SAMPLER_RANDOM && current_1d_dimension < (int32_t)samples_1d.size()) return samples_1d[(size_t)current_1d_dimension++][(size_t)cur_sample];
        return rng.uniform_float();
    }
    P2 get_2d() {
        if (kind == RSPT_SAMPLER_RANDOM) { Float x = rng.uniform_float(); Float y = rng.uniform_float(); return P2{x, y}; } // random.rs:86-92: x first
        if (current_2d_dimension < (int32_t)samples_2d.size()) return samples_2d[(size_t)current_2d_dimension++][(size_t)cur_sample];
        Float y = rng.uniform_float(); Float x = rng.uniform_float(); // Q4: y first (zerotwosequence.rs:178-181, stratified.rs, maxmin.rs)
        return P2{x, y};
    }
    bool start_next_sample() {
        current_1d_dimension = 0; current_2d_dimension = 0;
        cur_sample += 1;
        return cur_sample < spp;
    }
};

struct Sampler {
    int kind;
    SobolSampler sobol;
    HaltonSampler halton;
    PixelSampler pix;
    bool is_pixel() const { return kind >= RSPT_SAMPLER_RANDOM; }
    void reseed(uint64_t seed) { if (is_pixel()) pix.reseed(seed); } // Sampler::reseed: a no-op for Sobol' / Halton
    Sampler(const rspt_render_desc& rd)
        : kind((int)rd.sampler_kind), sobol(SobolTables{rd.tables.sobol32, rd.tables.vdc, rd.tables.vdc_inv}, rd.spp, rd.sample_bounds),
          halton(rd.spp, rd.sample_bounds, rd.sample_at_pixel_center != 0, rd.tables.halton_perms, rd.tables.n_halton_perms) { if (is_pixel()) pix.init(rd); }
    bool is_halton() const { return kind == RSPT_SAMPLER_HALTON; }
    int64_t cur_sample() const { return is_pixel() ? pix.cur_sample : (is_halton() ? halton.cur_sample : sobol.cur_sample); }
    // 2-D sample arrays requested by an integrator's preprocess (request_2d_array, sobol.rs:203-209): array i lives in
    // dimensions 5 + 2 i, 5 + 2 i + 1 (array_start_dim = 5, no 1-D arrays on this path); get_1d / get_2d skip that
    // range (sobol.rs:180-201, halton.rs:283-305).  round_count is the identity for both samplers (sobol.rs:210-212).
    std::vector<int32_t> arrays_2d;
    size_t array_2d_offset = 0;
    static const int64_t ARRAY_START_DIM = 5;
    int64_t array_end_dim() const { return ARRAY_START_DIM + 2 * (int64_t)arrays_2d.size(); }
    void request_2d_array(int32_t n) { arrays_2d.push_back(n); if (is_pixel()) pix.request_2d_array(n); }
    int32_t round_count(int32_t n) const { return is_pixel() ? pix.round_count(n) : n; }
    int64_t& dimension() { return is_halton() ? halton.dimension : sobol.dimension; }
    void start_pixel(int32_t x, int32_t y) {
        array_2d_offset = 0;
        if (is_pixel()) { pix.start_pixel(); return; }
        if (!is_halton()) { sobol.start_pixel(x, y); return; }
        halton.px = x; halton.py = y; halton.cur_sample = 0; halton.dimension = 0;
        halton.interval_sample_index = halton.get_index_for_sample(0);
    }
    Float get_1d() {
        if (is_pixel()) return pix.get_1d();
        if (dimension() >= ARRAY_START_DIM && dimension() < array_end_dim()) dimension() = array_end_dim();
        if (!is_halton()) return sobol.get_1d();
        Float r = halton.sample_dimension(halton.interval_sample_index, halton.dimension); halton.dimension += 1; return r;
    }
    P2 get_2d() {
        if (is_pixel()) return pix.get_2d();
        if (dimension() + 1 >= ARRAY_START_DIM && dimension() < array_end_dim()) dimension() = array_end_dim();
        if (!is_halton()) return sobol.get_2d();
        Float y = halton.sample_dimension(halton.interval_sample_index, halton.dimension + 1);
        Float x = halton.sample_dimension(halton.interval_sample_index, halton.dimension);
        halton.dimension += 2;
        return P2{x, y};
    }
    // element j of the (single) 2-D sample array a GlobalSampler fills in start_pixel (sobol.rs:166-179,
    // halton.rs:260-272): the sample of index get_index_for_sample(j) in the array dimensions; pixel sample s
    // owns elements [s * n, (s + 1) * n) (get_2d_array, sobol.rs:214-224)
    P2 array_2d(uint64_t j, int64_t dim) const {
        if (!is_halton()) { uint64_t idx = sobol.get_index_for_sample(j); return P2{sobol.sample_dimension(idx, dim), sobol.sample_dimension(idx, dim + 1)}; }
        uint64_t idx = halton.get_index_for_sample(j);
        return P2{halton.sample_dimension(idx, dim), halton.sample_dimension(idx, dim + 1)};
    }
    // get_2d_array_idxs (sobol.rs:225-236): false when the requested arrays are used up
    bool get_2d_array(int32_t n, size_t* array_idx, uint64_t* start) {
        if (array_2d_offset == arrays_2d.size()) return false;
        // (the reference asserts samples_2d_array_sizes[offset] == n)
        *array_idx = array_2d_offset; *start = (uint64_t)cur_sample() * (uint64_t)n;
        array_2d_offset += 1;
        return true;
    }
    P2 get_2d_sample(size_t array_idx, uint64_t j) const {
        if (is_pixel()) return pix.sample_array_2d[array_idx][(size_t)j];
        return array_2d(j, ARRAY_START_DIM + 2 * (int64_t)array_idx);
    }
    bool start_next_sample() { // halton.rs:333-343
        array_2d_offset = 0;
        if (is_pixel()) return pix.start_next_sample();
        if (!is_halton()) return sobol.start_next_sample();
        halton.dimension = 0;
        halton.interval_sample_index = halton.get_index_for_sample((uint64_t)halton.cur_sample + 1);
        halton.cur_sample += 1;
        return halton.cur_sample < halton.spp;
    }
};

// ---- Distribution1D: src/core/sampling.rs:17-147 ----
struct Distribution1D {
    std::vector<Float> func, cdf;
    Float func_int = 0;
    explicit Distribution1D(const std::vector<Float>& f) : func(f) {
        size_t n = f.size();
        cdf.resize(n + 1);
        cdf[0] = 0.0f;
        for (size_t i = 1; i <= n; i++) cdf[i] = cdf[i - 1] + f[i - 1] / (Float)n;
        func_int = cdf[n];
        if (func_int == 0.0f) for (size_t i = 1; i <= n; i++) cdf[i] = (Float)i / (Float)n;
        else for (size_t i = 1; i <= n; i++) cdf[i] /= func_int;
    }
    size_t sample_discrete(Float u, Float* pdf) const { // :103-142
        size_t first = 0, len = cdf.size();
        while (len > 0) {
            size_t half = len >> 1, middle = first + half;
            if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; } else len = half;
        }
        long off = clamp_t((long)first - 1, 0L, (long)cdf.size() - 2);
        if (pdf) *pdf = func_int > 0.0f ? func[off] / (func_int * (Float)func.size()) : 0.0f;
        return (size_t)off;
    }
};

// Integrators restated ahead of their GPU counterpart: selected through orc_render_integrator only.
enum { ORC_INTEGRATOR_FROM_DESC = 0, ORC_INTEGRATOR_DIRECT = 2, ORC_INTEGRATOR_WHITTED = 3 };
enum { ORC_DIRECT_SAMPLE_ALL = 0, ORC_DIRECT_SAMPLE_ONE = 1 }; // LightStrategy (directlighting.rs:17-21)
struct RenderCtx {
    const Scene* scene;
    const rspt_render_desc* rd;
    int ext_integrator = ORC_INTEGRATOR_FROM_DESC;
    int direct_strategy = ORC_DIRECT_SAMPLE_ALL;
    std::vector<int32_t> n_light_samples; // per light: Light::get_n_samples (api.rs "samples" / "nsamples", default 1)
    // light distribution state (src/core/lightdistrib.rs)
    int strategy; // after the "1 light -> uniform" rule (:397)
    std::shared_ptr<Distribution1D> fixed; // uniform / power
    int n_voxels[3];
    // dense grid of lazily-filled per-voxel distributions; stands in for the reference's
    // lock-free hash table (lightdistrib.rs:296-383): same values, no probing (Q18)
    std::unique_ptr<std::atomic<Distribution1D*>[]> voxels;
    size_t n_vox_total = 0;
    ~RenderCtx() { for (size_t i = 0; i < n_vox_total; i++) delete voxels[i].load(); }
};

// ---- DiffuseAreaLight: src/lights/diffuse.rs ----
static inline Spec light_l(const rspt_light& lt, V3 n, V3 w) { // :164-170
    if (lt.two_sided || dot(n, w) > 0.0f) return S3(lt.L);
    return Spec(0.0f);
}
// Bounds3f::bounding_sphere (geometry.rs:2160-2172) of the scene bound, used by DistantLight::preprocess
static inline Float world_radius(const Scene& sc) {
    Bounds3 b = sc.world_bound();
    V3 center = (b.p_min + b.p_max) / 2.0f;
    bool inside = center.x >= b.p_min.x && center.x <= b.p_max.x && center.y >= b.p_min.y && center.y <= b.p_max.y &&
                  center.z >= b.p_min.z && center.z <= b.p_max.z;
    return inside ? std::sqrt(distance_squared(center, b.p_max)) : 0.0f;
}
// ---- MipMap<Spectrum>: src/core/mipmap.rs:206-252,323-336 (wrap mode Repeat) ----
static inline const float* env_level(const rspt_envmap& m, uint32_t level, uint32_t* w, uint32_t* h) {
    const float* p = m.texels;
    uint32_t lw = m.width, lh = m.height;
    for (uint32_t i = 0; i < level; i++) { p += 3 * (size_t)lw * lh; lw = std::max(1u, lw / 2); lh = std::max(1u, lh / 2); }
    *w = lw; *h = lh;
    return p;
}
static inline Spec env_texel(const rspt_envmap& m, uint32_t level, int64_t s_, int64_t t_) { // :206-232
    uint32_t w, h;
    const float* p = env_level(m, level, &w, &h);
    uint64_t ss = (uint64_t)s_ % (uint64_t)w, tt = (uint64_t)t_ % (uint64_t)h; // mod_t(s as usize, u_size)
    return S3(p + 3 * (tt * w + ss));
}
static inline Spec env_triangle(const rspt_envmap& m, uint32_t level, P2 st) { // :323-336
    if (level > m.n_levels - 1) level = m.n_levels - 1;
    uint32_t w, h;
    env_level(m, level, &w, &h);
    Float s = st.x * (Float)w - 0.5f, t = st.y * (Float)h - 0.5f;
    int64_t s0 = (int64_t)std::floor(s), t0 = (int64_t)std::floor(t);
    Float ds = s - (Float)s0, dt = t - (Float)t0;
    Spec tmp1 = env_texel(m, level, s0 + 1, t0 + 1) * (ds * dt);
    Spec tmp2 = env_texel(m, level, s0 + 1, t0) * (ds * (1.0f - dt));
    Spec tmp3 = env_texel(m, level, s0, t0 + 1) * ((1.0f - ds) * dt);
    Spec tmp4 = env_texel(m, level, s0, t0) * ((1.0f - ds) * (1.0f - dt));
    return tmp4 + tmp3 + tmp2 + tmp1;
}
static inline Spec env_lookup(const rspt_envmap& m, P2 st, Float width) { // lookup_pnt_flt :233-252
    Float level = (Float)m.n_levels - 1.0f + std::log2(std::fmax(width, 1e-8f));
    if (level < 0.0f) return env_triangle(m, 0, st);
    if (level >= (Float)m.n_levels - 1.0f) return env_texel(m, m.n_levels - 1, 0, 0);
    uint32_t il = (uint32_t)f2usize(std::floor(level));
    Float delta = level - (Float)il;
    Spec a = env_triangle(m, il, st), b = env_triangle(m, il + 1, st);
    return a * (1.0f - delta) + b * delta; // lerp(delta, a, b)
}
// Distribution2D over the light's scalar image (sampling.rs:150-198), rebuilt per call site from dist_func
struct Distribution2D {
    std::vector<Distribution1D> cond;
    std::unique_ptr<Distribution1D> marginal;
    Distribution2D(const float* f, uint32_t nu, uint32_t nv) {
        std::vector<Float> mf;
        for (uint32_t v = 0; v < nv; v++) { cond.emplace_back(std::vector<Float>(f + (size_t)v * nu, f + (size_t)(v + 1) * nu)); mf.push_back(cond.back().func_int); }
        marginal.reset(new Distribution1D(mf));
    }
    static Float sample_continuous_1d(const Distribution1D& d, Float u, Float* pdf, size_t* off) { // sampling.rs:53-101
        size_t first = 0, len = d.cdf.size();
        while (len > 0) {
            size_t half = len >> 1, middle = first + half;
            if (d.cdf[middle] <= u) { first = middle + 1; len -= half + 1; } else len = half;
        }
        long o = clamp_t((long)first - 1, 0L, (long)d.cdf.size() - 2);
        if (off) *off = (size_t)o;
        Float du = u - d.cdf[o];
        if ((d.cdf[o + 1] - d.cdf[o]) > 0.0f) du /= d.cdf[o + 1] - d.cdf[o];
        if (pdf) *pdf = d.func_int > 0.0f ? d.func[o] / d.func_int : 0.0f;
        return ((Float)o + du) / (Float)d.func.size();
    }
    P2 sample_continuous(P2 u, Float* pdf) const {
        Float pdfs[2] = {0, 0};
        size_t v = 0;
        Float d1 = sample_continuous_1d(*marginal, u.y, &pdfs[1], &v);
        Float d0 = sample_continuous_1d(cond[v], u.x, &pdfs[0], nullptr);
        *pdf = pdfs[0] * pdfs[1];
        return P2{d0, d1};
    }
    Float pdf(P2 p) const {
        size_t nu = cond[0].func.size(), nv = marginal->func.size();
        size_t iu = std::min((size_t)f2usize(p.x * (Float)nu), nu - 1), iv = std::min((size_t)f2usize(p.y * (Float)nv), nv - 1);
        return cond[iv].func[iu] / marginal->func_int;
    }
};
static inline const Distribution2D& env_distribution(const rspt_envmap& m) { // cached per envmap pointer
    static std::mutex mu;
    static std::map<const float*, std::unique_ptr<Distribution2D>> cache;
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(m.dist_func);
    if (it == cache.end()) it = cache.emplace(m.dist_func, std::unique_ptr<Distribution2D>(new Distribution2D(m.dist_func, m.dist_nu, m.dist_nv))).first;
    return *it->second;
}
static inline V3 mat3_mul(const float* m, V3 w) { return V3{m[0] * w.x + m[1] * w.y + m[2] * w.z, m[3] * w.x + m[4] * w.y + m[5] * w.z, m[6] * w.x + m[7] * w.y + m[8] * w.z}; }
// InfiniteAreaLight::le (infinite.rs:369-377)
static inline Spec infinite_le(const Scene& sc, const rspt_light& lt, V3 ray_d) {
    V3 w = normalize(mat3_mul(lt.p + 9, ray_d));
    P2 st{spherical_phi(w) * INV_2_PI, spherical_theta(w) * INV_PI};
    return env_lookup(sc.d.envmaps[lt.prim], st, 0.0f);
}
// InfiniteAreaLight::pdf_li (infinite.rs:378-392)
static inline Float infinite_pdf_li(const Scene& sc, const rspt_light& lt, V3 w) {
    V3 wi = mat3_mul(lt.p + 9, w);
    Float theta = spherical_theta(wi), phi = spherical_phi(wi);
    Float sin_theta = std::sin(theta);
    if (sin_theta == 0.0f) return 0.0f;
    return env_distribution(sc.d.envmaps[lt.prim]).pdf(P2{phi * INV_2_PI, theta * INV_PI}) / (2.0f * PI * PI * sin_theta);
}
static inline bool light_is_delta(const rspt_light& lt) { return lt.kind == RSPT_LIGHT_POINT || lt.kind == RSPT_LIGHT_SPOT || lt.kind == RSPT_LIGHT_DISTANT; } // light.rs:178-188
// SpotLight::falloff spot.rs:67-80
static inline Float spot_falloff(const rspt_light& lt, V3 w) {
    const float* m = lt.p + 3;
    V3 wl = normalize(V3{m[0] * w.x + m[1] * w.y + m[2] * w.z, m[3] * w.x + m[4] * w.y + m[5] * w.z, m[6] * w.x + m[7] * w.y + m[8] * w.z});
    Float cos_theta_ = wl.z, cos_total = lt.p[12], cos_start = lt.p[13];
    if (cos_theta_ < cos_total) return 0.0f;
    if (cos_theta_ >= cos_start) return 1.0f;
    Float delta = (cos_theta_ - cos_total) / (cos_start - cos_total);
    return (delta * delta) * (delta * delta);
}
// Light::sample_li: diffuse.rs:64-84, point.rs:52-68, spot.rs:81-106, distant.rs:41-58
static inline Spec light_sample_li(const Scene& sc, const rspt_light& lt, const Interaction& iref, P2 u, V3* wi, Float* pdf, Interaction* light_intr) {
    if (lt.kind == RSPT_LIGHT_DIFFUSE_AREA) {
        *light_intr = sc.tri_sample_ref(sc.d.prims[lt.prim], iref, u, pdf);
        if (*pdf == 0.0f || length_squared(light_intr->p - iref.p) == 0.0f) { *pdf = 0.0f; return Spec(); }
        *wi = normalize(light_intr->p - iref.p);
        return light_l(lt, light_intr->n, -*wi);
    }
    Interaction li; // InteractionCommon::default(): n = 0, p_error = 0
    li.p_error = V3{0, 0, 0}; li.n = V3{0, 0, 0}; li.wo = V3{0, 0, 0}; li.time = iref.time;
    if (lt.kind == RSPT_LIGHT_INFINITE) { // infinite.rs:298-341
        const rspt_envmap& m = sc.d.envmaps[lt.prim];
        Float map_pdf = 0.0f;
        P2 uv = env_distribution(m).sample_continuous(u, &map_pdf);
        if (map_pdf == 0.0f) { *pdf = 0.0f; return Spec(); } // (the reference leaves *pdf at the caller's 0)
        Float theta = uv.y * PI, phi = uv.x * 2.0f * PI;
        Float cos_theta = std::cos(theta), sin_theta = std::sin(theta), sin_phi = std::sin(phi), cos_phi = std::cos(phi);
        *wi = mat3_mul(lt.p, V3{sin_theta * cos_phi, sin_theta * sin_phi, cos_theta});
        *pdf = map_pdf / (2.0f * PI * PI * sin_theta);
        if (sin_theta == 0.0f) *pdf = 0.0f;
        li.p = iref.p + *wi * (2.0f * world_radius(sc));
        *light_intr = li;
        return env_lookup(m, uv, 0.0f);
    }
    *pdf = 1.0f;
    Spec out;
    if (lt.kind == RSPT_LIGHT_DISTANT) {
        V3 w{lt.p[0], lt.p[1], lt.p[2]};
        *wi = w;
        li.p = iref.p + w * (2.0f * world_radius(sc));
        out = S3(lt.L);
    } else {
        V3 pl{lt.p[0], lt.p[1], lt.p[2]};
        *wi = normalize(pl - iref.p);
        li.p = pl;
        Float d2 = distance_squared(pl, iref.p);
        if (lt.kind == RSPT_LIGHT_POINT) out = S3(lt.L) / d2;
        else out = S3(lt.L) * spot_falloff(lt, -*wi) / d2;
    }
    *light_intr = li;
    return out;
}
// Light::power: diffuse.rs:85-93, point.rs:69-71, spot.rs:107-113, distant.rs:59-62
static inline Spec light_power(const Scene& sc, const rspt_light& lt) {
    switch (lt.kind) {
    case RSPT_LIGHT_POINT: return S3(lt.L) * (4.0f * PI);
    case RSPT_LIGHT_SPOT: return S3(lt.L) * 2.0f * PI * (1.0f - 0.5f * (lt.p[13] + lt.p[12]));
    case RSPT_LIGHT_DISTANT: { Float r = world_radius(sc); return S3(lt.L) * PI * r * r; }
    case RSPT_LIGHT_INFINITE: { Float r = world_radius(sc); return env_lookup(sc.d.envmaps[lt.prim], P2{0.5f, 0.5f}, 0.5f) * Spec(PI * r * r); } // infinite.rs:342-346
    default: {
        Float factor = lt.two_sided ? 2.0f : 1.0f;
        return S3(lt.L) * factor * sc.tri_area(sc.d.prims[lt.prim]) * PI;
    }
    }
}

// ---- SpatialLightDistribution::compute_distribution: lightdistrib.rs:169-269 ----
static inline Distribution1D* spatial_compute(const RenderCtx& cx, const int pi[3]) {
    const Scene& sc = *cx.scene;
    Bounds3 wb = sc.world_bound();
    V3 p0{(Float)pi[0] / (Float)cx.n_voxels[0], (Float)pi[1] / (Float)cx.n_voxels[1], (Float)pi[2] / (Float)cx.n_voxels[2]};
    V3 p1{(Float)(pi[0] + 1) / (Float)cx.n_voxels[0], (Float)(pi[1] + 1) / (Float)cx.n_voxels[1], (Float)(pi[2] + 1) / (Float)cx.n_voxels[2]};
    Bounds3 vb; vb.p_min = wb.lerp3(p0); vb.p_max = wb.lerp3(p1);
    const size_t n_samples = 128;
    uint32_t nl = sc.d.n_lights;
    std::vector<Float> contrib(nl, 0.0f);
    for (size_t i = 0; i < n_samples; i++) {
        V3 po = vb.lerp3(V3{radical_inverse(0, i), radical_inverse(1, i), radical_inverse(2, i)});
        Interaction intr; intr.p = po; intr.time = 0; intr.p_error = V3{0, 0, 0}; intr.wo = V3{1, 0, 0}; intr.n = V3{0, 0, 0};
        P2 u{radical_inverse(3, i), radical_inverse(4, i)};
        for (uint32_t j = 0; j < nl; j++) {
            Float pdf = 0; V3 wi{0, 0, 0}; Interaction li_intr;
            Spec li = light_sample_li(sc, sc.d.lights[j], intr, u, &wi, &pdf, &li_intr);
            if (pdf > 0.0f) contrib[j] += li.y() / pdf;
        }
    }
    Float sum = 0.0f; for (Float c : contrib) sum += c; // iter().sum()
    Float avg = sum / (Float)(n_samples * contrib.size());
    Float min_contrib = avg > 0.0f ? 0.001f * avg : 1.0f;
    for (Float& c : contrib) c = std::fmax(c, min_contrib);
    return new Distribution1D(contrib);
}
// lookup: lightdistrib.rs:276-384 (hash probing replaced by an ordered map: same values, Q18)
static inline const Distribution1D* light_lookup(RenderCtx& cx, V3 p) {
    if (cx.strategy != RSPT_LIGHTS_SPATIAL) return cx.fixed.get();
    V3 off = cx.scene->world_bound().offset(p);
    int pi[3];
    for (int i = 0; i < 3; i++) pi[i] = clamp_t(f2i(off[i] * (Float)cx.n_voxels[i]), 0, cx.n_voxels[i] - 1);
    size_t key = ((size_t)pi[2] * cx.n_voxels[1] + pi[1]) * cx.n_voxels[0] + pi[0];
    Distribution1D* cur = cx.voxels[key].load(std::memory_order_acquire);
    if (cur) return cur;
    Distribution1D* dist = spatial_compute(cx, pi);
    Distribution1D* expected = nullptr;
    if (cx.voxels[key].compare_exchange_strong(expected, dist, std::memory_order_acq_rel)) return dist;
    delete dist; // another thread filled the voxel first (same value: pure function of the voxel)
    return expected;
}
static inline void light_distrib_init(RenderCtx& cx) { // create_light_sample_distribution :393-418, new :127-166
    const Scene& sc = *cx.scene;
    uint32_t nl = sc.d.n_lights;
    cx.strategy = (int)cx.rd->light_strategy;
    if (cx.strategy == RSPT_LIGHTS_UNIFORM || nl == 1) {
        cx.strategy = RSPT_LIGHTS_UNIFORM;
        cx.fixed = std::make_shared<Distribution1D>(std::vector<Float>(nl, 1.0f));
    } else if (cx.strategy == RSPT_LIGHTS_POWER) {
        std::vector<Float> pw;
        for (uint32_t i = 0; i < nl; i++) pw.push_back(light_power(sc, sc.d.lights[i]).y()); // integrator.rs:573-584
        cx.fixed = std::make_shared<Distribution1D>(pw);
    } else {
        cx.strategy = RSPT_LIGHTS_SPATIAL;
        Bounds3 b = sc.world_bound();
        V3 diag = b.diagonal();
        Float bmax = diag[b.maximum_extent()];
        for (int i = 0; i < 3; i++) cx.n_voxels[i] = std::max(1, f2i(std::round(diag[i] / bmax * 64.0f)));
        cx.n_vox_total = (size_t)cx.n_voxels[0] * cx.n_voxels[1] * cx.n_voxels[2];
        cx.voxels.reset(new std::atomic<Distribution1D*>[cx.n_vox_total]);
        for (size_t i = 0; i < cx.n_vox_total; i++) cx.voxels[i].store(nullptr);
    }
}

// ---- integrator.rs:359-570 ----
static inline Spec estimate_direct(RenderCtx& cx, const Interaction& it, const Bsdf& bsdf, P2 u_scattering, uint32_t light_num, P2 u_light, Counters* c) {
    const Scene& sc = *cx.scene;
    const rspt_light& light = sc.d.lights[light_num];
    const uint8_t flags = BSDF_ALL & ~BSDF_SPECULAR;
    Spec ld(0.0f);
    V3 wi{0, 0, 0};
    Float light_pdf = 0.0f, scattering_pdf = 0.0f;
    Interaction light_intr;
    Spec li = light_sample_li(sc, light, it, u_light, &wi, &light_pdf, &light_intr);
    if (light_pdf > 0.0f && !li.is_black()) {
        Spec f = bsdf.f(it.wo, wi, flags) * Spec(abs_dot(wi, it.sh_n));
        scattering_pdf = bsdf.pdf(it.wo, wi, flags);
        if (!f.is_black()) {
            Ray sray = it.spawn_ray_to(light_intr); // VisibilityTester::unoccluded light.rs:199-206
            if (sc.intersect_p(sray, c)) li = Spec(0.0f);
            if (!li.is_black()) {
                if (light_is_delta(light)) ld = ld + f * li / light_pdf; // integrator.rs:470-471
                else {
                    Float weight = power_heuristic(1, light_pdf, 1, scattering_pdf);
                    ld = ld + f * li * Spec(weight) / light_pdf;
                }
            }
        }
    }
    // sample BSDF with MIS, skipped for delta lights (integrator.rs:480)
    if (!light_is_delta(light)) {
        uint8_t sampled_type = 0; // Q6: stays 0 => sampled_specular always false
        Spec f = bsdf.sample_f(it.wo, &wi, u_scattering, &scattering_pdf, flags, &sampled_type);
        f = f * Spec(abs_dot(wi, it.sh_n));
        bool sampled_specular = (sampled_type & BSDF_SPECULAR) != 0;
        if (!f.is_black() && scattering_pdf > 0.0f) {
            Float weight = 1.0f;
            if (!sampled_specular) {
                light_pdf = light.kind == RSPT_LIGHT_INFINITE ? infinite_pdf_li(sc, light, wi)
                                                              : sc.tri_pdf_ref(sc.d.prims[light.prim], it, wi); // pdf_li diffuse.rs:100-103
                if (light_pdf == 0.0f) return ld;
                weight = power_heuristic(1, scattering_pdf, 1, light_pdf);
            }
            Ray ray = it.spawn_ray(wi);
            Spec li2;
            Interaction light_isect;
            if (c) c->mis_rays++;
            if (sc.intersect(ray, &light_isect, c)) {
                const rspt_prim& hp = sc.hit_prim(light_isect);
                if (light.kind == RSPT_LIGHT_DIFFUSE_AREA && hp.area_light >= 0 && (uint32_t)hp.area_light == light_num) // pointer compare :550-558
                    li2 = light_l(light, light_isect.n, -wi);
            } else {
                li2 = light.kind == RSPT_LIGHT_INFINITE ? infinite_le(sc, light, ray.d) : Spec(); // Light::le (integrator.rs:561-563)
            }
            if (!li2.is_black()) ld = ld + f * li2 * Spec(1.0f) * weight / scattering_pdf;
        }
    }
    return ld;
}

static inline Spec uniform_sample_one_light(RenderCtx& cx, const Interaction& it, const Bsdf& bsdf, Sampler& sampler, const Distribution1D& distrib, Counters* c) {
    uint32_t nl = cx.scene->d.n_lights;
    if (nl == 0) return Spec();
    Float pdf = 0.0f;
    size_t light_num = distrib.sample_discrete(sampler.get_1d(), &pdf);
    if (pdf == 0.0f) return Spec();
    P2 u_light = sampler.get_2d();
    P2 u_scattering = sampler.get_2d();
    return estimate_direct(cx, it, bsdf, u_scattering, (uint32_t)light_num, u_light, c) / pdf;
}

// ---- PathIntegrator::li: src/integrators/path.rs:59-282 ----
static inline Spec path_li(RenderCtx& cx, const Ray& r, Sampler& sampler, Counters* c) {
    const Scene& sc = *cx.scene;
    Spec l, beta(1.0f);
    Ray ray = r;
    bool specular_bounce = false;
    uint32_t bounces = 0;
    Float eta_scale = 1.0f;
    for (;;) {
        Interaction isect;
        if (sc.intersect(ray, &isect, c)) {
            const rspt_prim& hp = sc.hit_prim(isect);
            if (bounces == 0 || specular_bounce) {
                if (hp.area_light >= 0) l = l + beta * light_l(sc.d.lights[hp.area_light], isect.n, -ray.d); // interaction.rs:475-483
                else l = l + beta * Spec();
            }
            if (bounces >= cx.rd->max_depth) break;
            if (hp.material == 0xffffffffu) { ray = isect.spawn_ray(ray.d); continue; } // null bsdf :109-116
            // isect.compute_scattering_functions (interaction.rs:371-386): differentials of the camera ray
            // (bounce rays carry none), then the material
            compute_differentials(&isect, ray);
            Bsdf bsdf;
            make_bsdf(sc, isect, hp.material, true, &bsdf); // path.rs:108: allow_multiple_lobes = true
            if (c) c->bounces++;
            // lookup happens for every hit (path.rs:118); with no lights its result is never used
            const Distribution1D* distrib = sc.d.n_lights ? light_lookup(cx, isect.p) : nullptr;
            if (sc.d.n_lights && bsdf.num_components(BSDF_ALL & ~BSDF_SPECULAR) > 0) {
                Spec ld = beta * uniform_sample_one_light(cx, isect, bsdf, sampler, *distrib, c);
                l = l + ld;
            }
            V3 wo = -ray.d, wi{0, 0, 0};
            Float pdf = 0.0f;
            uint8_t sampled_type = 255;
            Spec f = bsdf.sample_f(wo, &wi, sampler.get_2d(), &pdf, BSDF_ALL, &sampled_type);
            if (f.is_black() || pdf == 0.0f) break;
            beta = beta * ((f * abs_dot(wi, isect.sh_n)) / pdf);
            specular_bounce = (sampled_type & BSDF_SPECULAR) != 0;
            if ((sampled_type & BSDF_SPECULAR) && (sampled_type & BSDF_TRANSMISSION)) {
                Float eta = bsdf.eta;
                if (dot(wo, isect.n) > 0.0f) eta_scale *= eta * eta;
                else eta_scale *= 1.0f / (eta * eta);
            }
            ray = isect.spawn_ray(wi);
            // (BSSRDF branch :191-249 out of scope)
            Spec rr_beta = beta * eta_scale;
            if (rr_beta.max_component_value() < cx.rd->rr_threshold && bounces > 3) {
                Float q = std::fmax(0.05f, 1.0f - rr_beta.max_component_value());
                if (sampler.get_1d() < q) break;
                beta = beta / (1.0f - q);
            }
        } else {
            if (bounces == 0 || specular_bounce) // path.rs:267-277: scene.infinite_lights in Scene.lights order (scene.rs:40-43)
                for (uint32_t i = 0; i < sc.d.n_lights; i++)
                    if (sc.d.lights[i].kind == RSPT_LIGHT_INFINITE) l = l + beta * infinite_le(sc, sc.d.lights[i], ray.d);
            break;
        }
        bounces += 1;
    }
    return l;
}

// ---- participating media: src/media/homogeneous.rs, src/core/medium.rs (VolPathIntegrator) ----
static inline Spec spec_exp(Spec a) { return Spec(std::exp(a.c[0]), std::exp(a.c[1]), std::exp(a.c[2])); } // Spectrum::exp spectrum.rs:1620-1622
static inline Spec medium_sigma_t(const rspt_medium& m) { // HomogeneousMedium::new homogeneous.rs:24-31: sigma_s + sigma_a
    return Spec(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]) + Spec(m.sigma_a[0], m.sigma_a[1], m.sigma_a[2]);
}
static inline Spec homogeneous_tr(const rspt_medium& m, const Ray& ray) { // HomogeneousMedium::tr homogeneous.rs:33-36
    Spec st = medium_sigma_t(m);
    Spec nst(-st.c[0], -st.c[1], -st.c[2]);
    return spec_exp(nst * std::fmin(ray.t_max * length(ray.d), std::numeric_limits<Float>::max()));
}
// Medium::sample -> HomogeneousMedium::sample (homogeneous.rs:37-91): returns the path-throughput factor; *mi is filled (and
// *sampled set) when the sampled distance ends before ray.t_max
static inline Spec homogeneous_sample(const rspt_medium& m, uint32_t medium, const Ray& ray, Float u_channel, Float u_dist, Interaction* mi, bool* sampled) {
    const Spec sigma_t = medium_sigma_t(m), sigma_s(m.sigma_s[0], m.sigma_s[1], m.sigma_s[2]);
    size_t channel = (size_t)(u_channel * 3.0f);
    if (channel > 2) channel = 2;
    const Float dist = -std::log(1.0f - u_dist) / sigma_t.c[channel];
    const Float len = length(ray.d);
    const Float t = std::fmin(dist / len, ray.t_max);
    const bool sampled_medium = t < ray.t_max;
    if (sampled_medium) { // MediumInteraction::new (interaction.rs:128-151): n = 0, p_error = 0, interface (medium, medium)
        *mi = Interaction{};
        mi->p = ray.o + ray.d * t; // Ray::position geometry.rs:2392-2394
        mi->wo = -ray.d;
        mi->time = ray.time;
        mi->med_in = mi->med_out = medium;
        mi->is_medium = true;
        mi->phase_g = m.g;
    }
    *sampled = sampled_medium;
    const Spec nst(-sigma_t.c[0], -sigma_t.c[1], -sigma_t.c[2]);
    const Spec tr = spec_exp(nst * std::fmin(t, std::numeric_limits<Float>::max()) * len);
    const Spec density = sampled_medium ? sigma_t * tr : tr;
    Float pdf = 0.0f;
    for (int i = 0; i < 3; i++) pdf += density.c[i];
    pdf *= 1.0f / 3.0f;
    if (pdf == 0.0f) pdf = 1.0f; // (assert!(tr.is_black()))
    return sampled_medium ? tr * sigma_s / pdf : tr / pdf;
}
// ---- GridDensityMedium (src/media/grid.rs).  Its tr / sample draw a data-dependent number of sampler values — also in the middle of
// estimate_direct — which only the pixel samplers' serial streams can follow for long (the dimension-indexed samplers panic past their
// last dimension, sobol.rs:119-124).  `next_1d` stands for sampler.get_1d(). ----
static inline Float grid_d(const GridMedium& m, int32_t x, int32_t y, int32_t z) { // :57-75 (pnt3i_inside_exclusive)
    if (!(x >= 0 && x < m.nx && y >= 0 && y < m.ny && z >= 0 && z < m.nz)) return 0.0f;
    return m.density[((size_t)z * m.ny + y) * m.nx + x];
}
static inline Float grid_density(const GridMedium& m, V3 p) { // :76-153
    const V3 ps{p.x * (Float)m.nx - 0.5f, p.y * (Float)m.ny - 0.5f, p.z * (Float)m.nz - 0.5f};
    const int32_t ix = f2i(std::floor(ps.x)), iy = f2i(std::floor(ps.y)), iz = f2i(std::floor(ps.z));
    const V3 d{ps.x - (Float)ix, ps.y - (Float)iy, ps.z - (Float)iz};
    const Float d00 = lerp(d.x, grid_d(m, ix, iy, iz), grid_d(m, ix + 1, iy, iz));
    const Float d10 = lerp(d.x, grid_d(m, ix, iy + 1, iz), grid_d(m, ix + 1, iy + 1, iz));
    const Float d01 = lerp(d.x, grid_d(m, ix, iy, iz + 1), grid_d(m, ix + 1, iy, iz + 1));
    const Float d11 = lerp(d.x, grid_d(m, ix, iy + 1, iz + 1), grid_d(m, ix + 1, iy + 1, iz + 1));
    const Float d0 = lerp(d.y, d00, d10), d1 = lerp(d.y, d01, d11);
    return lerp(d.z, d0, d1);
}
// Bounds3f::intersect_b (geometry.rs:2183-2210) against the unit cube the medium lives in
static inline bool unit_cube_intersect_b(const Ray& ray, Float* hitt0, Float* hitt1) {
    Float t0 = 0.0f, t1 = ray.t_max;
    const Float o[3] = {ray.o.x, ray.o.y, ray.o.z}, dd[3] = {ray.d.x, ray.d.y, ray.d.z};
    for (int i = 0; i < 3; i++) {
        const Float inv_ray_dir = 1.0f / dd[i];
        Float t_near = (0.0f - o[i]) * inv_ray_dir, t_far = (1.0f - o[i]) * inv_ray_dir;
        if (t_near > t_far) std::swap(t_near, t_far);
        t_far *= 1.0f + 2.0f * gamma(3);
        if (t_near > t0) t0 = t_near;
        if (t_far < t1) t1 = t_far;
        if (t0 > t1) return false;
    }
    *hitt0 = t0; *hitt1 = t1;
    return true;
}
// the prelude tr and sample share (:158-176, :216-235): the world ray normalised (t_max scaled), then taken to medium space
static inline Ray grid_medium_ray(const GridMedium& m, const Ray& r_world) {
    Ray in_ray{r_world.o, normalize(r_world.d), r_world.t_max * length(r_world.d), 0.0f}; // ..Default::default(): time 0, no differential, no medium
    return transform_ray(m.world_to_medium, in_ray);
}
template <class Next1D>
static inline Spec grid_tr(const GridMedium& m, const Ray& r_world, Next1D&& next_1d) { // GridDensityMedium::tr :155-208: ratio tracking
    const Ray ray = grid_medium_ray(m, r_world);
    Float t_min = 0.0f, t_max = 0.0f;
    if (!unit_cube_intersect_b(ray, &t_min, &t_max)) return Spec(1.0f);
    Float tr = 1.0f, t = t_min;
    for (;;) {
        t -= std::log(1.0f - next_1d()) * m.inv_max_density / m.sigma_t;
        if (t >= t_max) break;
        const Float density = grid_density(m, ray.o + ray.d * t);
        tr *= 1.0f - std::fmax(0.0f, density * m.inv_max_density);
        const Float rr_threshold = 0.1f; // "added after book publication"
        if (tr < rr_threshold) {
            const Float q = std::fmax(0.05f, 1.0f - tr);
            if (next_1d() < q) return Spec(0.0f);
            tr /= 1.0f - q;
        }
    }
    return Spec(tr);
}
template <class Next1D>
static inline Spec grid_sample(const GridMedium& m, uint32_t medium, const Ray& r_world, Next1D&& next_1d, Interaction* mi, bool* sampled) { // ::sample :209-270: delta tracking
    *sampled = false;
    const Ray ray = grid_medium_ray(m, r_world);
    Float t_min = 0.0f, t_max = 0.0f;
    if (!unit_cube_intersect_b(ray, &t_min, &t_max)) return Spec(1.0f);
    Float t = t_min;
    for (;;) {
        t -= std::log(1.0f - next_1d()) * m.inv_max_density / m.sigma_t;
        if (t >= t_max) break;
        if (grid_density(m, ray.o + ray.d * t) * m.inv_max_density > next_1d()) {
            *mi = Interaction{};
            mi->p = r_world.o + r_world.d * t; // r_world.position(t): the WORLD ray as given (not normalised) at the normalised ray's parameter (:243)
            mi->wo = -r_world.d;
            mi->time = r_world.time;
            mi->med_in = mi->med_out = medium;
            mi->is_medium = true;
            mi->phase_g = m.g;
            *sampled = true;
            return m.sigma_s / m.sigma_t;
        }
    }
    return Spec(1.0f);
}

// Medium::tr / Medium::sample (medium.rs:268-294): dispatch on the medium's kind
static inline Spec medium_tr(const Scene& sc, const Ray& ray, Sampler& sampler) {
    const rspt_medium& m = sc.d.media[ray.medium - 1];
    if (m.kind == RSPT_MEDIUM_GRID) return grid_tr(sc.grid(ray.medium), ray, [&]() { return sampler.get_1d(); });
    return homogeneous_tr(m, ray);
}
static inline Spec medium_sample(const Scene& sc, const Ray& ray, Sampler& sampler, Interaction* mi, bool* sampled) {
    const rspt_medium& m = sc.d.media[ray.medium - 1];
    if (m.kind == RSPT_MEDIUM_GRID) return grid_sample(sc.grid(ray.medium), ray.medium, ray, [&]() { return sampler.get_1d(); }, mi, sampled);
    const Float u_channel = sampler.get_1d(); // homogeneous.rs:43
    const Float u_dist = sampler.get_1d();    // :49
    return homogeneous_sample(m, ray.medium, ray, u_channel, u_dist, mi, sampled);
}

static inline Float phase_hg(Float cos_theta, Float g) { // medium.rs:389-392
    const Float denom = 1.0f + g * g + 2.0f * g * cos_theta;
    return INV_4_PI * (1.0f - g * g) / (denom * std::sqrt(denom));
}
static inline Float hg_sample_p(Float g, V3 wo, V3* wi, P2 u) { // HenyeyGreenstein::sample_p medium.rs:306-331
    Float cos_theta;
    if (std::fabs(g) < 1e-3f) cos_theta = 1.0f - 2.0f * u.x;
    else {
        const Float sqr_term = (1.0f - g * g) / (1.0f + g - 2.0f * g * u.x);
        cos_theta = -(1.0f + g * g - sqr_term * sqr_term) / (2.0f * g);
    }
    const Float sin_theta = std::sqrt(std::fmax(0.0f, 1.0f - cos_theta * cos_theta));
    const Float phi = 2.0f * PI * u.y;
    V3 v1, v2;
    coordinate_system(wo, &v1, &v2);
    *wi = v1 * (sin_theta * std::cos(phi)) + v2 * (sin_theta * std::sin(phi)) + wo * cos_theta; // spherical_direction_vec3 geometry.rs:1570-1579
    return phase_hg(cos_theta, g);
}
// VisibilityTester::tr (light.rs:207-239)
static inline Spec visibility_tr(const Scene& sc, const Interaction& p0, const Interaction& p1, Sampler& sampler, Counters* c) {
    Ray ray = p0.spawn_ray_to(p1);
    Spec tr(1.0f);
    for (;;) {
        Interaction isect;
        if (sc.intersect(ray, &isect, c)) {
            if (isect.prim >= 0) { // isect.primitive is Some (an instanced hit has lost it, Q11: then neither branch runs)
                if (sc.d.prims[isect.prim].material != 0xffffffffu) return Spec();
                if (ray.medium) tr = tr * medium_tr(sc, ray, sampler);
            }
        } else {
            if (ray.medium) tr = tr * medium_tr(sc, ray, sampler);
            break;
        }
        ray = isect.spawn_ray_to(p1);
    }
    return tr;
}
// Scene::intersect_tr (scene.rs:79-106)
static inline bool intersect_tr(const Scene& sc, Ray* ray, Interaction* isect, Spec* tr, Sampler& sampler, Counters* c) {
    for (;;) {
        const bool hit_surface = sc.intersect(*ray, isect, c);
        if (ray->medium) *tr = *tr * medium_tr(sc, *ray, sampler);
        if (!hit_surface) return false;
        if (isect->prim >= 0 && sc.d.prims[isect->prim].material != 0xffffffffu) return true;
        *ray = isect->spawn_ray(ray->d);
    }
}
// estimate_direct (integrator.rs:406-570) with handle_media = true, specular = false; `bsdf` is null for a medium interaction
static inline Spec estimate_direct_media(RenderCtx& cx, const Interaction& it, const Bsdf* bsdf, P2 u_scattering, uint32_t light_num, P2 u_light, Sampler& sampler, Counters* c) {
    const Scene& sc = *cx.scene;
    const rspt_light& light = sc.d.lights[light_num];
    const uint8_t flags = BSDF_ALL & ~BSDF_SPECULAR;
    Spec ld(0.0f);
    V3 wi{0, 0, 0};
    Float light_pdf = 0.0f, scattering_pdf = 0.0f;
    Interaction light_intr;
    Spec li = light_sample_li(sc, light, it, u_light, &wi, &light_pdf, &light_intr);
    if (light_pdf > 0.0f && !li.is_black()) {
        Spec f(0.0f);
        if (!it.is_medium) { // is_surface_interaction: n != 0
            f = bsdf->f(it.wo, wi, flags) * Spec(abs_dot(wi, it.sh_n));
            scattering_pdf = bsdf->pdf(it.wo, wi, flags);
        } else {
            const Float p = phase_hg(dot(it.wo, wi), it.phase_g); // HenyeyGreenstein::p medium.rs:302-305
            f = Spec(p);
            scattering_pdf = p;
        }
        if (!f.is_black()) {
            li = li * visibility_tr(sc, it, light_intr, sampler, c); // handle_media (:462-463)
            if (!li.is_black()) {
                if (light_is_delta(light)) ld = ld + f * li / light_pdf;
                else {
                    Float weight = power_heuristic(1, light_pdf, 1, scattering_pdf);
                    ld = ld + f * li * Spec(weight) / light_pdf;
                }
            }
        }
    }
    if (!light_is_delta(light)) {
        Spec f(0.0f);
        bool sampled_specular = false;
        if (!it.is_medium) {
            uint8_t sampled_type = 0;
            f = bsdf->sample_f(it.wo, &wi, u_scattering, &scattering_pdf, flags, &sampled_type);
            f = f * Spec(abs_dot(wi, it.sh_n));
            sampled_specular = (sampled_type & BSDF_SPECULAR) != 0;
        } else {
            const Float p = hg_sample_p(it.phase_g, it.wo, &wi, u_scattering);
            f = Spec(p);
            scattering_pdf = p;
        }
        if (!f.is_black() && scattering_pdf > 0.0f) {
            Float weight = 1.0f;
            if (!sampled_specular) {
                light_pdf = light.kind == RSPT_LIGHT_INFINITE ? infinite_pdf_li(sc, light, wi) : sc.tri_pdf_ref(sc.d.prims[light.prim], it, wi);
                if (light_pdf == 0.0f) return ld;
                weight = power_heuristic(1, scattering_pdf, 1, light_pdf);
            }
            Ray ray = it.spawn_ray(wi);
            bool found_surface_interaction = false;
            Spec li2;
            Interaction light_isect;
            Spec tr_spectrum; // Spectrum::default(): the transmittance intersect_tr multiplies into starts at ZERO (integrator.rs:531, scene.rs:86)
            if (c) c->mis_rays++;
            const bool hit_surface = intersect_tr(sc, &ray, &light_isect, &tr_spectrum, sampler, c);
            const Spec tr = tr_spectrum;
            if (hit_surface) {
                found_surface_interaction = true;
                const rspt_prim& hp = sc.hit_prim(light_isect);
                if (light.kind == RSPT_LIGHT_DIFFUSE_AREA && hp.area_light >= 0 && (uint32_t)hp.area_light == light_num) li2 = light_l(light, light_isect.n, -wi);
            }
            if (!found_surface_interaction) li2 = light.kind == RSPT_LIGHT_INFINITE ? infinite_le(sc, light, ray.d) : Spec();
            if (!li2.is_black()) ld = ld + f * li2 * tr * weight / scattering_pdf;
        }
    }
    return ld;
}
static inline Spec uniform_sample_one_light_media(RenderCtx& cx, const Interaction& it, const Bsdf* bsdf, Sampler& sampler, const Distribution1D& distrib, Counters* c) {
    if (cx.scene->d.n_lights == 0) return Spec();
    Float pdf = 0.0f;
    size_t light_num = distrib.sample_discrete(sampler.get_1d(), &pdf);
    if (pdf == 0.0f) return Spec();
    P2 u_light = sampler.get_2d();
    P2 u_scattering = sampler.get_2d();
    return estimate_direct_media(cx, it, bsdf, u_scattering, (uint32_t)light_num, u_light, sampler, c) / pdf;
}

// ---- VolPathIntegrator::li: src/integrators/volpath.rs:60-347 ----
static inline Spec volpath_li(RenderCtx& cx, const Ray& r, Sampler& sampler, Counters* c) {
    const Scene& sc = *cx.scene;
    Spec l, beta(1.0f);
    Ray ray = r;
    bool specular_bounce = false;
    uint32_t bounces = 0;
    Float eta_scale = 1.0f;
    // the block both branches run for a medium interaction (:101-127 / :304-330)
    auto scatter_in_medium = [&](const Interaction& mi) {
        if (sc.d.n_lights) { // (light_distribution is Some only with lights; lookup panics on an empty scene otherwise)
            const Distribution1D* distrib = light_lookup(cx, mi.p);
            l = l + beta * uniform_sample_one_light_media(cx, mi, nullptr, sampler, *distrib, c);
        }
        V3 wi{0, 0, 0};
        hg_sample_p(mi.phase_g, -ray.d, &wi, sampler.get_2d());
        ray = mi.spawn_ray(wi);
        specular_bounce = false;
    };
    for (;;) {
        Interaction isect, mi;
        bool have_mi = false;
        if (sc.intersect(ray, &isect, c)) {
            if (ray.medium) beta = beta * medium_sample(sc, ray, sampler, &mi, &have_mi);
            if (beta.is_black()) break;
            if (have_mi) {
                if (bounces >= cx.rd->max_depth) break;
                scatter_in_medium(mi);
            } else {
                const rspt_prim& hp = sc.hit_prim(isect);
                if (bounces == 0 || specular_bounce) {
                    if (hp.area_light >= 0) l = l + beta * light_l(sc.d.lights[hp.area_light], isect.n, -ray.d);
                    else l = l + beta * Spec();
                }
                if (bounces >= cx.rd->max_depth) break;
                if (hp.material == 0xffffffffu) { ray = isect.spawn_ray(ray.d); continue; } // :141-145: `continue` skips `bounces += 1`
                compute_differentials(&isect, ray);
                Bsdf bsdf;
                make_bsdf(sc, isect, hp.material, true, &bsdf); // volpath.rs:146
                if (c) c->bounces++;
                if (sc.d.n_lights) {
                    const Distribution1D* distrib = light_lookup(cx, isect.p);
                    l = l + beta * uniform_sample_one_light_media(cx, isect, &bsdf, sampler, *distrib, c); // no non-specular-lobe test here (:146-161)
                }
                V3 wo = -ray.d, wi{0, 0, 0};
                Float pdf = 0.0f;
                uint8_t sampled_type = 255;
                Spec f = bsdf.sample_f(wo, &wi, sampler.get_2d(), &pdf, BSDF_ALL, &sampled_type);
                if (f.is_black() || pdf == 0.0f) break;
                beta = beta * ((f * abs_dot(wi, isect.sh_n)) / pdf);
                specular_bounce = (sampled_type & BSDF_SPECULAR) != 0;
                if ((sampled_type & BSDF_SPECULAR) && (sampled_type & BSDF_TRANSMISSION)) {
                    Float eta = bsdf.eta;
                    if (dot(wo, isect.n) > 0.0f) eta_scale *= eta * eta;
                    else eta_scale *= 1.0f / (eta * eta);
                }
                ray = isect.spawn_ray(wi);
                // (BSSRDF branch :193-270 out of scope)
            }
            Spec rr_beta = beta * eta_scale; // :275-285: also after scattering in a medium
            if (rr_beta.max_component_value() < cx.rd->rr_threshold && bounces > 3) {
                Float q = std::fmax(0.05f, 1.0f - rr_beta.max_component_value());
                if (sampler.get_1d() < q) break;
                beta = beta / (1.0f - q);
            }
        } else {
            if (ray.medium) beta = beta * medium_sample(sc, ray, sampler, &mi, &have_mi);
            if (beta.is_black()) break;
            if (have_mi) {
                if (bounces >= cx.rd->max_depth) break;
                scatter_in_medium(mi);
            }
            if (bounces == 0 || specular_bounce) // :332-337: with the ray as it is NOW (after a scattering event: the scattered ray)
                for (uint32_t i = 0; i < sc.d.n_lights; i++)
                    if (sc.d.lights[i].kind == RSPT_LIGHT_INFINITE) l = l + beta * infinite_le(sc, sc.d.lights[i], ray.d);
            break; // :338-339: the path ends here even if it has just scattered
        }
        bounces += 1;
    }
    return l;
}

// ---- AOIntegrator::li: src/integrators/ao.rs:50-96 ----
static inline V3 uniform_sample_hemisphere(P2 u) { // sampling.rs:236-242
    Float z = u.x;
    Float r = std::sqrt(std::fmax(0.0f, 1.0f - z * z));
    Float phi = 2.0f * PI * u.y;
    return V3{r * std::cos(phi), r * std::sin(phi), z};
}
static inline Spec ao_li(RenderCtx& cx, const Ray& ray, Sampler& sampler, Counters* c) {
    const Scene& sc = *cx.scene;
    const rspt_render_desc& rd = *cx.rd;
    Spec l;
    Interaction isect;
    if (sc.intersect(ray, &isect, c)) {
        // (compute_scattering_functions only touches shading geometry and the BSDF; li reads neither)
        V3 n = faceforward(isect.n, -ray.d);
        V3 s = normalize(isect.dpdu);
        V3 t = cross(isect.n, s); // nrm_cross_vec3(&isect.common.n, &s)
        int32_t ns = (int32_t)rd.ao_n_samples;
        size_t which = 0;
        uint64_t first = 0;
        const bool have = sampler.get_2d_array(ns, &which, &first); // ao.rs:75: the pixel sample's slice of the array preprocess requested
        for (int32_t k = 0; have && k < ns; k++) {
            P2 u = sampler.get_2d_sample(which, first + (uint64_t)k); // GlobalSampler: array dimensions 5, 6 (array_start_dim = 5); PixelSampler: its own array
            V3 wi; Float pdf;
            if (rd.ao_cos_sample) { wi = cosine_sample_hemisphere(u); pdf = std::fabs(wi.z) * INV_PI; }
            else { wi = uniform_sample_hemisphere(u); pdf = INV_2_PI; }
            wi = V3{s.x * wi.x + t.x * wi.y + n.x * wi.z, s.y * wi.x + t.y * wi.y + n.y * wi.z, s.z * wi.x + t.z * wi.y + n.z * wi.z};
            if (pdf != 0.0f && !sc.intersect_p(isect.spawn_ray(wi), c)) l = l + Spec(dot(wi, n) / (pdf * (Float)ns));
        }
    }
    return l;
}

// ---- DirectLightingIntegrator (src/integrators/directlighting.rs) and WhittedIntegrator (src/integrators/whitted.rs) ----
// Both recurse through specular_reflect / specular_transmit; the sampler's dimension counter follows that depth-first
// order.  Materials must have been flattened with allow_multiple_lobes = false (compute_scattering_functions(ray, false, ..)).
static inline Spec uniform_sample_all_lights(RenderCtx& cx, const Interaction& it, const Bsdf& bsdf, Sampler& sampler, Counters* c) { // integrator.rs:300-355
    const Scene& sc = *cx.scene;
    Spec l;
    for (uint32_t j = 0; j < sc.d.n_lights && j < cx.n_light_samples.size(); j++) {
        int32_t n_samples = cx.n_light_samples[j];
        size_t ia = 0, ib = 0; uint64_t sa = 0, sb = 0;
        bool have_a = sampler.get_2d_array(n_samples, &ia, &sa);
        bool have_b = sampler.get_2d_array(n_samples, &ib, &sb);
        if (!have_a || !have_b) {
            P2 u_light = sampler.get_2d();
            P2 u_scattering = sampler.get_2d();
            l = l + estimate_direct(cx, it, bsdf, u_scattering, j, u_light, c);
        } else {
            Spec ld;
            for (int32_t k = 0; k < n_samples; k++) {
                P2 u_scattering = sampler.get_2d_sample(ib, sb + (uint64_t)k);
                P2 u_light = sampler.get_2d_sample(ia, sa + (uint64_t)k);
                ld = ld + estimate_direct(cx, it, bsdf, u_scattering, j, u_light, c);
            }
            l = l + ld / (Float)n_samples;
        }
    }
    return l;
}
static inline Spec uniform_sample_one_light_nodistrib(RenderCtx& cx, const Interaction& it, const Bsdf& bsdf, Sampler& sampler, Counters* c) { // integrator.rs:359-403 with light_distrib = None
    uint32_t nl = cx.scene->d.n_lights;
    if (nl == 0) return Spec();
    uint32_t light_num = std::min((uint32_t)(sampler.get_1d() * (Float)nl), nl - 1u);
    Float light_pdf = 1.0f / (Float)nl;
    P2 u_light = sampler.get_2d();
    P2 u_scattering = sampler.get_2d();
    return estimate_direct(cx, it, bsdf, u_scattering, light_num, u_light, c) / light_pdf;
}
static inline Spec recursive_li(RenderCtx& cx, const Ray& ray, Sampler& sampler, int depth, Counters* c);
static inline V3 differential_normal(const Interaction& si, bool x) { // dndx / dndy (directlighting.rs:164-167)
    return x ? si.sh_dndu * si.dudx + si.sh_dndv * si.dvdx : si.sh_dndu * si.dudy + si.sh_dndv * si.dvdy;
}
static inline Spec specular_reflect(RenderCtx& cx, const Ray& ray, const Interaction& isect, const Bsdf& bsdf, Sampler& sampler, int depth, Counters* c) { // directlighting.rs:133-192 == whitted.rs:127-186
    V3 wo = isect.wo, wi{0, 0, 0};
    Float pdf = 0.0f;
    V3 ns = isect.sh_n;
    uint8_t sampled_type = 0;
    Spec f = bsdf.sample_f(wo, &wi, sampler.get_2d(), &pdf, BSDF_REFLECTION | BSDF_SPECULAR, &sampled_type);
    if (!(pdf > 0.0f && !f.is_black() && abs_dot(wi, ns) != 0.0f)) return Spec();
    Ray rd = isect.spawn_ray(wi);
    if (ray.has_diff) {
        V3 dndx = differential_normal(isect, true), dndy = differential_normal(isect, false);
        V3 dwodx = -ray.rx_d - wo, dwody = -ray.ry_d - wo;
        Float ddndx = dot(dwodx, ns) + dot(wo, dndx);
        Float ddndy = dot(dwody, ns) + dot(wo, dndy);
        rd.has_diff = true;
        rd.rx_o = isect.p + isect.dpdx; rd.ry_o = isect.p + isect.dpdy;
        rd.rx_d = wi - dwodx + (dndx * dot(wo, ns) + ns * ddndx) * 2.0f;
        rd.ry_d = wi - dwody + (dndy * dot(wo, ns) + ns * ddndy) * 2.0f;
    }
    return f * recursive_li(cx, rd, sampler, depth + 1, c) * Spec(abs_dot(wi, ns) / pdf);
}
static inline Spec specular_transmit(RenderCtx& cx, const Ray& ray, const Interaction& isect, const Bsdf& bsdf, Sampler& sampler, int depth, Counters* c) { // directlighting.rs:193-258 == whitted.rs:187-253
    V3 wo = isect.wo, wi{0, 0, 0};
    Float pdf = 0.0f;
    V3 ns = isect.sh_n;
    uint8_t sampled_type = 0;
    Spec f = bsdf.sample_f(wo, &wi, sampler.get_2d(), &pdf, BSDF_TRANSMISSION | BSDF_SPECULAR, &sampled_type);
    if (!(pdf > 0.0f && !f.is_black() && abs_dot(wi, ns) != 0.0f)) return Spec();
    Ray rd = isect.spawn_ray(wi);
    if (ray.has_diff) {
        Float eta = bsdf.eta;
        V3 w = -wo;
        if (dot(wo, ns) < 0.0f) eta = 1.0f / eta;
        V3 dndx = differential_normal(isect, true), dndy = differential_normal(isect, false);
        V3 dwodx = -ray.rx_d - wo, dwody = -ray.ry_d - wo;
        Float ddndx = dot(dwodx, ns) + dot(wo, dndx);
        Float ddndy = dot(dwody, ns) + dot(wo, dndy);
        Float mu = eta * dot(w, ns) - dot(wi, ns);
        Float dmudx = (eta - (eta * eta * dot(w, ns)) / dot(wi, ns)) * ddndx;
        Float dmudy = (eta - (eta * eta * dot(w, ns)) / dot(wi, ns)) * ddndy;
        rd.has_diff = true;
        rd.rx_o = isect.p + isect.dpdx; rd.ry_o = isect.p + isect.dpdy;
        rd.rx_d = wi + dwodx * eta - (dndx * mu + ns * dmudx);
        rd.ry_d = wi + dwody * eta - (dndy * mu + ns * dmudy);
    }
    return f * recursive_li(cx, rd, sampler, depth + 1, c) * Spec(abs_dot(wi, ns) / pdf);
}
static inline Spec recursive_li(RenderCtx& cx, const Ray& ray, Sampler& sampler, int depth, Counters* c) { // directlighting.rs:71-123, whitted.rs:43-117
    const Scene& sc = *cx.scene;
    const bool whitted = cx.ext_integrator == ORC_INTEGRATOR_WHITTED;
    Spec l;
    Interaction isect;
    if (!sc.intersect(ray, &isect, c)) {
        for (uint32_t i = 0; i < sc.d.n_lights; i++) // every light's le(ray); only the infinite light's is not black
            if (sc.d.lights[i].kind == RSPT_LIGHT_INFINITE) l = l + infinite_le(sc, sc.d.lights[i], ray.d);
        return l;
    }
    const rspt_prim& hp = sc.hit_prim(isect);
    V3 n_before = isect.sh_n; // whitted.rs:58: shading.n read before compute_scattering_functions (i.e. before a bump map moves it)
    if (hp.material == 0xffffffffu) return recursive_li(cx, isect.spawn_ray(ray.d), sampler, depth, c);
    compute_differentials(&isect, ray);
    Bsdf bsdf;
    make_bsdf(sc, isect, hp.material, false, &bsdf); // directlighting.rs:86 / whitted.rs:63: allow_multiple_lobes = false
    if (c) c->bounces++;
    V3 wo = isect.wo;
    if (hp.area_light >= 0) l = l + light_l(sc.d.lights[hp.area_light], isect.n, wo); // isect.le(&wo)
    if (whitted) {
        for (uint32_t j = 0; j < sc.d.n_lights; j++) { // whitted.rs:74-101: one sample per light, no MIS
            const rspt_light& light = sc.d.lights[j];
            V3 wi{0, 0, 0};
            Float pdf = 0.0f;
            Interaction light_intr;
            Spec li = light_sample_li(sc, light, isect, sampler.get_2d(), &wi, &pdf, &light_intr);
            if (li.is_black() || pdf == 0.0f) continue;
            Spec f = bsdf.f(wo, wi, BSDF_ALL);
            if (!f.is_black() && !sc.intersect_p(isect.spawn_ray_to(light_intr), c)) l = l + f * li * Spec(abs_dot(wi, n_before) / pdf);
        }
    } else if (sc.d.n_lights) {
        if (cx.direct_strategy == ORC_DIRECT_SAMPLE_ALL) l = l + uniform_sample_all_lights(cx, isect, bsdf, sampler, c);
        else l = l + uniform_sample_one_light_nodistrib(cx, isect, bsdf, sampler, c);
    }
    if ((uint32_t)(depth + 1) < cx.rd->max_depth) {
        l = l + specular_reflect(cx, ray, isect, bsdf, sampler, depth, c);
        l = l + specular_transmit(cx, ray, isect, bsdf, sampler, depth, c);
    }
    return l;
}

// ---- PerspectiveCamera::generate_ray_differential: src/cameras/perspective.rs:190-280 ----
// CameraBase.camera_to_world (an AnimatedTransform): the key matrices of a moving camera, or the one matrix twice
static inline AnimatedTransform camera_animation(const rspt_render_desc& rd) {
    return rd.camera_animated ? AnimatedTransform(rd.camera_to_world, rd.camera_time[0], rd.camera_to_world_end, rd.camera_time[1])
                              : AnimatedTransform(rd.camera_to_world, 0.0f, rd.camera_to_world, 1.0f);
}
static inline Ray camera_ray(const rspt_render_desc& rd, const AnimatedTransform& c2w, P2 p_film, Float time_s, P2 p_lens) {
    V3 p_camera = transform_point(rd.raster_to_camera, V3{p_film.x, p_film.y, 0.0f});
    V3 dir = normalize(p_camera);
    Ray in_ray{V3{0, 0, 0}, dir, INF, lerp(time_s, rd.shutter_open, rd.shutter_close)};
    if (rd.lens_radius > 0.0f) {
        P2 pl = concentric_sample_disk(p_lens);
        pl = P2{pl.x * rd.lens_radius, pl.y * rd.lens_radius};
        Float ft = rd.focal_distance / in_ray.d.z;
        V3 p_focus = in_ray.o + in_ray.d * ft;
        in_ray.o = V3{pl.x, pl.y, 0.0f};
        in_ray.d = normalize(p_focus - in_ray.o);
    }
    // offset rays (perspective.rs:205-220, 245-271); dx_camera / dy_camera as in PerspectiveCamera::new (:82-97)
    V3 c0 = transform_point(rd.raster_to_camera, V3{0, 0, 0});
    V3 dx_camera = transform_point(rd.raster_to_camera, V3{1, 0, 0}) - c0;
    V3 dy_camera = transform_point(rd.raster_to_camera, V3{0, 1, 0}) - c0;
    in_ray.has_diff = true;
    in_ray.rx_o = in_ray.ry_o = V3{0, 0, 0};
    in_ray.rx_d = normalize(p_camera + dx_camera);
    in_ray.ry_d = normalize(p_camera + dy_camera);
    if (rd.lens_radius > 0.0f) {
        P2 pl = concentric_sample_disk(p_lens);
        pl = P2{pl.x * rd.lens_radius, pl.y * rd.lens_radius};
        V3 dx = normalize(p_camera + dx_camera);
        Float ftx = rd.focal_distance / dx.z;
        V3 pfx = V3{0, 0, 0} + dx * ftx;
        in_ray.rx_o = V3{pl.x, pl.y, 0.0f};
        in_ray.rx_d = normalize(pfx - in_ray.rx_o);
        V3 dy = normalize(p_camera + dy_camera);
        Float fty = rd.focal_distance / dy.z;
        V3 pfy = V3{0, 0, 0} + dy * fty;
        in_ray.ry_o = V3{pl.x, pl.y, 0.0f};
        in_ray.ry_d = normalize(pfy - in_ray.ry_o);
    }
    return c2w.transform_ray(in_ray); // perspective.rs:279
}
static inline Ray camera_ray(const rspt_render_desc& rd, P2 p_film, Float time_s, P2 p_lens) { return camera_ray(rd, camera_animation(rd), p_film, time_s, p_lens); }

// ---- film: src/core/film.rs:57-153,308-371 ----
struct FilmTilePixel { Spec contrib_sum; Float filter_weight_sum = 0; };
struct FilmTile {
    int32_t pb[4]; // pixel_bounds
    std::vector<FilmTilePixel> pixels;
    void add_sample(const rspt_render_desc& rd, P2 p_film, Spec l, Float sample_weight) { // :94-147
        if (l.y() > rd.max_sample_luminance) l = l * Spec(rd.max_sample_luminance / l.y());
        P2 pd{p_film.x - 0.5f, p_film.y - 0.5f};
        int32_t p0x = f2i(std::ceil(pd.x - rd.filter_radius[0])), p0y = f2i(std::ceil(pd.y - rd.filter_radius[1]));
        int32_t p1x = f2i(std::floor(pd.x + rd.filter_radius[0])) + 1, p1y = f2i(std::floor(pd.y + rd.filter_radius[1])) + 1;
        p0x = std::max(p0x, pb[0]); p0y = std::max(p0y, pb[1]);
        p1x = std::min(p1x, pb[2]); p1y = std::min(p1y, pb[3]);
        Float inv_rx = 1.0f / rd.filter_radius[0], inv_ry = 1.0f / rd.filter_radius[1];
        const Float ts = 16.0f;
        for (int32_t y = p0y; y < p1y; y++) {
            Float fy = std::fabs(((Float)y - pd.y) * inv_ry * ts);
            size_t ify = (size_t)f2usize(std::fmin(std::floor(fy), ts - 1.0f));
            for (int32_t x = p0x; x < p1x; x++) {
                Float fx = std::fabs(((Float)x - pd.x) * inv_rx * ts);
                size_t ifx = (size_t)f2usize(std::fmin(std::floor(fx), ts - 1.0f));
                Float w = rd.filter_table[ify * 16 + ifx];
                FilmTilePixel& px = pixels[(size_t)(y - pb[1]) * (size_t)(pb[2] - pb[0]) + (size_t)(x - pb[0])];
                px.contrib_sum = px.contrib_sum + l * Spec(sample_weight) * Spec(w);
                px.filter_weight_sum += w;
            }
        }
    }
};
static inline FilmTile get_film_tile(const rspt_render_desc& rd, const int32_t tb[4]) { // :308-345
    FilmTile t;
    Float pminx = (Float)tb[0] - 0.5f - rd.filter_radius[0], pminy = (Float)tb[1] - 0.5f - rd.filter_radius[1];
    Float pmaxx = (Float)tb[2] - 0.5f + rd.filter_radius[0], pmaxy = (Float)tb[3] - 0.5f + rd.filter_radius[1];
    int32_t p0x = f2i(std::ceil(pminx)), p0y = f2i(std::ceil(pminy));
    int32_t p1x = f2i(std::floor(pmaxx)) + 1, p1y = f2i(std::floor(pmaxy)) + 1;
    // bnd2_intersect_bnd2i with cropped_pixel_bounds
    t.pb[0] = std::max(p0x, rd.crop_px[0]); t.pb[1] = std::max(p0y, rd.crop_px[1]);
    t.pb[2] = std::min(p1x, rd.crop_px[2]); t.pb[3] = std::min(p1y, rd.crop_px[3]);
    int64_t area = (int64_t)std::max(0, t.pb[2] - t.pb[0]) * std::max(0, t.pb[3] - t.pb[1]);
    t.pixels.resize((size_t)area);
    return t;
}

// Film::merge_film_tile (film.rs:346-371) into film_xyzw = Film.pixels over crop_px (xyz + filter_weight_sum per pixel)
static inline void merge_film_tile(const rspt_render_desc& rd, const FilmTile& t, float* film_xyzw) {
    const int cw = rd.crop_px[2] - rd.crop_px[0];
    const int w = t.pb[2] - t.pb[0];
    for (int32_t y = t.pb[1]; y < t.pb[3]; y++)
        for (int32_t x = t.pb[0]; x < t.pb[2]; x++) {
            const FilmTilePixel& tp = t.pixels[(size_t)(y - t.pb[1]) * w + (x - t.pb[0])];
            float* mp = film_xyzw + 4 * ((size_t)(y - rd.crop_px[1]) * cw + (x - rd.crop_px[0]));
            Float xyz[3];
            rgb_to_xyz(tp.contrib_sum.c, xyz);
            for (int i = 0; i < 3; i++) mp[i] += xyz[i];
            mp[3] += tp.filter_weight_sum;
        }
}

// blockqueue/mod.rs:100-115
static inline uint32_t part1_by1(uint32_t x) {
    x &= 0x0000ffff; x = (x ^ (x << 8)) & 0x00ff00ff; x = (x ^ (x << 4)) & 0x0f0f0f0f;
    x = (x ^ (x << 2)) & 0x33333333; return (x ^ (x << 1)) & 0x55555555;
}
static inline uint32_t morton2(uint32_t x, uint32_t y) { return (part1_by1(y) << 1) + part1_by1(x); }

struct RenderOut {
    Counters counters;
    double seconds = 0;
};

// ---- SamplerIntegrator::render: src/core/integrator.rs:70-220 ----
// PathIntegrator::li from somewhere else: oracle/make_flow_fixtures.py compiles the REFERENCE'S TEXT of li (path.rs:59-282) over this oracle's leaf functions and runs it
// through this tile loop (oracle/_ref/libflowref.so sets the pointer in its own copy of this header-only code; liboracle.so never does)
inline Spec (*g_li_override)(RenderCtx&, const Ray&, Sampler&, Counters*) = nullptr;
inline Spec (*g_ao_li_override)(RenderCtx&, const Ray&, Sampler&, Counters*) = nullptr;         // AOIntegrator::li (ao.rs:53-110)
inline Spec (*g_direct_li_override)(RenderCtx&, const Ray&, Sampler&, Counters*) = nullptr;     // the same for DirectLightingIntegrator::li (directlighting.rs:71-131)
// film_xyzw: Film.pixels after all merges (xyz + filter_weight_sum per cropped pixel);
// li_rgb (optional): radiance per camera sample, [(pixel*spp+s)*3] over crop_px.
static inline void render(const Scene& scene, const rspt_render_desc& rd, int num_threads, float* film_xyzw, float* li_rgb, RenderOut* out,
                          int ext_integrator = ORC_INTEGRATOR_FROM_DESC, int direct_strategy = ORC_DIRECT_SAMPLE_ALL, const int32_t* n_light_samples = nullptr) {
    scene.prepare_media();
    RenderCtx cx;
    cx.scene = &scene; cx.rd = &rd;
    cx.ext_integrator = ext_integrator; cx.direct_strategy = direct_strategy;
    for (uint32_t i = 0; i < scene.d.n_lights; i++) cx.n_light_samples.push_back(n_light_samples ? n_light_samples[i] : 1);
    light_distrib_init(cx);
    const int32_t* sb = rd.sample_bounds;
    int32_t ext_x = sb[2] - sb[0], ext_y = sb[3] - sb[1];
    int32_t ts = (int32_t)rd.tile_size;
    int32_t ntx = (ext_x + ts - 1) / ts, nty = (ext_y + ts - 1) / ts;
    // BlockQueue::new: Morton-sorted tile list (blockqueue/mod.rs:23-52); sort_by_key is stable
    std::vector<std::pair<uint32_t, uint32_t>> blocks;
    for (int32_t i = 0; i < ntx * nty; i++) blocks.push_back({(uint32_t)(i % ntx), (uint32_t)(i / ntx)});
    std::stable_sort(blocks.begin(), blocks.end(), [](auto a, auto b) { return morton2(a.first, a.second) < morton2(b.first, b.second); });
    // multi-GPU style sharding of the tile list (not in the reference; shard_count == 1 there)
    std::vector<size_t> mine;
    uint32_t chunk = rd.tile_chunk ? rd.tile_chunk : 1, sc_ = rd.shard_count ? rd.shard_count : 1;
    for (size_t i = 0; i < blocks.size(); i++) if ((i / chunk) % sc_ == rd.shard_index) mine.push_back(i);
    std::vector<FilmTile> tiles(mine.size());
    std::atomic<size_t> next{0};
    int cw = rd.crop_px[2] - rd.crop_px[0], ch = rd.crop_px[3] - rd.crop_px[1];
    std::vector<Counters> tc((size_t)std::max(1, num_threads));
    const AnimatedTransform cam_c2w = camera_animation(rd);
    auto worker = [&](int tid) {
        Sampler sampler(rd);
        if (rd.integrator == RSPT_INTEGRATOR_AO && ext_integrator == ORC_INTEGRATOR_FROM_DESC) sampler.request_2d_array((int32_t)rd.ao_n_samples); // ao.rs:44-48
        if (ext_integrator == ORC_INTEGRATOR_DIRECT && direct_strategy == ORC_DIRECT_SAMPLE_ALL) // preprocess, directlighting.rs:54-70
            for (uint32_t i = 0; i < rd.max_depth; i++)
                for (uint32_t j = 0; j < scene.d.n_lights; j++) { sampler.request_2d_array(cx.n_light_samples[j]); sampler.request_2d_array(cx.n_light_samples[j]); }
        Counters& c = tc[tid];
        for (;;) {
            size_t k = next.fetch_add(1);
            if (k >= mine.size()) break;
            auto blk = blocks[mine[k]];
            int32_t x0 = sb[0] + (int32_t)blk.first * ts, x1 = std::min(x0 + ts, sb[2]);
            int32_t y0 = sb[1] + (int32_t)blk.second * ts, y1 = std::min(y0 + ts, sb[3]);
            int32_t tb[4] = {x0, y0, x1, y1};
            FilmTile ft = get_film_tile(rd, tb);
            sampler.reseed((uint64_t)(int64_t)(int32_t)((int32_t)blk.second * ntx + (int32_t)blk.first)); // integrator.rs:113-114: seed = tile.y * n_tiles.x + tile.x
            for (int32_t py = y0; py < y1; py++)
                for (int32_t px = x0; px < x1; px++) {
                    sampler.start_pixel(px, py);
                    // pixel_bounds == sample_bounds (Q16)
                    if (!(px >= sb[0] && px < sb[2] && py >= sb[1] && py < sb[3])) continue;
                    bool done = false;
                    while (!done) {
                        P2 f2 = sampler.get_2d();
                        P2 p_film{(Float)px + f2.x, (Float)py + f2.y}; // sampler.rs:85-95
                        Float time_s = sampler.get_1d();
                        P2 p_lens = sampler.get_2d();
                        Ray ray = camera_ray(rd, cam_c2w, p_film, time_s, p_lens);
                        ray.scale_differentials(1.0f / std::sqrt((Float)rd.spp)); // integrator.rs:140-144 (get_samples_per_pixel)
                        Float ray_weight = 1.0f;
                        if (rd.sample_count && ((uint64_t)sampler.cur_sample() < rd.sample_begin || (uint64_t)sampler.cur_sample() >= rd.sample_begin + rd.sample_count)) {
                            done = !sampler.start_next_sample(); // checkpoint / resume (not in the reference): this sample belongs to another range
                            continue;
                        }
                        Spec l = (ext_integrator == ORC_INTEGRATOR_DIRECT && g_direct_li_override) ? g_direct_li_override(cx, ray, sampler, &c)
                                 : ext_integrator != ORC_INTEGRATOR_FROM_DESC ? recursive_li(cx, ray, sampler, 0, &c)
                                 : rd.integrator == RSPT_INTEGRATOR_AO       ? (g_ao_li_override ? g_ao_li_override(cx, ray, sampler, &c) : ao_li(cx, ray, sampler, &c))
                                 : rd.integrator == RSPT_INTEGRATOR_VOLPATH  ? volpath_li(cx, ray, sampler, &c)
                                 : g_li_override                             ? g_li_override(cx, ray, sampler, &c)
                                                                             : path_li(cx, ray, sampler, &c);
                        c.samples++;
                        if (l.has_nans()) { l = Spec(0.0f); c.nan_samples++; } // integrator.rs:165-173 (Q1)
                        if (li_rgb && px >= rd.crop_px[0] && px < rd.crop_px[2] && py >= rd.crop_px[1] && py < rd.crop_px[3]) {
                            size_t pix = (size_t)(py - rd.crop_px[1]) * cw + (size_t)(px - rd.crop_px[0]);
                            float* o = li_rgb + (pix * (size_t)rd.spp + (size_t)sampler.cur_sample()) * 3;
                            o[0] = l.c[0]; o[1] = l.c[1]; o[2] = l.c[2];
                        }
                        ft.add_sample(rd, p_film, l, ray_weight);
                        done = !sampler.start_next_sample();
                    }
                }
            tiles[k] = std::move(ft);
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    if (num_threads <= 1) worker(0);
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < num_threads; i++) th.emplace_back(worker, i);
        for (auto& t : th) t.join();
    }
    // merge_film_tile (film.rs:346-371), in Morton order (the reference's channel-arrival order is
    // nondeterministic; only pixels with >=3 contributing tiles can tell the difference)
    if (film_xyzw) {
        std::memset(film_xyzw, 0, sizeof(float) * 4 * (size_t)cw * ch);
        for (const FilmTile& t : tiles) merge_film_tile(rd, t, film_xyzw);
    }
    auto t1 = std::chrono::steady_clock::now();
    if (out) {
        out->seconds = std::chrono::duration<double>(t1 - t0).count();
        for (auto& c : tc) out->counters.add(c);
    }
}

} // namespace orc
