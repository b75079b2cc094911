// Two sweeps around the random-walk model of k_trace_w4's node step (step_model.hip), asked for by VERDICT r4 #3 / #5:
// what a lane's dependent chain of 128-byte record fetches costs on gfx950 as a function of
//   (a) OCCUPANCY — 1 / 2 / 3 / 4 / 5 / 8 workgroups of 256 threads per CU (dynamic LDS padding sets the limit), 64 MB of records, uniform walk;
//   (b) the L2 HIT RATE at a fixed table size — a fraction p of the steps goes to a HOT subset that fits one XCD's 4 MB L2 (1 MB, the same for
//       every XCD, or — "affine" — a different 1 MB per XCD, picked by HW_REG_XCC_ID: what XCD-affine ray dealing would do to the tree), the rest
//       uniformly to the whole 64 MB;
//   (c) the WORKING SET — uniform walks over 1 MB ... 512 MB.
// Prints G steps/s (= G lines/s: one 128-byte line per step, K = 7 loads of 16 B, ~100 VALU per step) per point.  The production kernel's C2
// launches: 67.5 G lines/s at an L2 hit rate of 0.63 and 326 cycles of latency (BENCH_r04); C3: 32.7 G at 0.50 / 426 cycles.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void k_walk(const float4* __restrict__ recs, uint32_t mask, uint32_t hot_mask, uint32_t hot_thresh /* of 2^16 */, int affine, uint32_t steps, uint32_t* out) {
    __shared__ uint2 stack[12 * 256];
    extern __shared__ float4 pad[];   // occupancy limiter
    if (threadIdx.x == 0) pad[0] = make_float4(0, 0, 0, 0);
    uint2* my = stack + threadIdx.x;
    uint32_t xcc = 0;
    if (affine) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7u; }
    const uint32_t hot_base = xcc * (hot_mask + 1u);   // affine: XCD x keeps its own hot region; else everyone shares region 0
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
    float acc = 1.0f, ox = 0.37f, ix = 1.13f;
    uint32_t sp = 0;
    for (uint32_t s = 0; s < steps; s++) {
        const float4* p = recs + (size_t)idx * 8;
        float4 a[7];
#pragma unroll
        for (int j = 0; j < 7; j++) a[j] = p[j];
        float m0 = -1e30f, m1 = 1e30f;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            const float lo = (a[j].x - ox) * ix, hi = (a[j].z - ox) * ix, lo2 = (a[j].y - ox) * ix, hi2 = (a[j].w - ox) * ix;
            m0 = fmaxf(m0, fmaxf(fminf(lo, hi), fminf(lo2, hi2)));
            m1 = fminf(m1, fminf(fmaxf(lo, hi), fmaxf(lo2, hi2)));
        }
        ox += 1e-3f;
        acc = acc * 0.999f + (m0 < m1 ? 1e-7f : 2e-7f);
        my[(sp % 12u) * 256u] = make_uint2(idx, __float_as_uint(m0));
        sp++;
        const uint2 e = my[((sp + 5u) % 12u) * 256u];
        uint32_t h = __float_as_uint(a[6].w) ^ (e.x * 0x9E3779B9u) ^ (s * 0x85EBCA6Bu) ^ __float_as_uint(acc);
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
        const uint32_t sel = (h * 0x61C88647u) >> 16;
        idx = sel < hot_thresh ? hot_base + (h & hot_mask) : (h & mask);
    }
    if (acc == 12345.0f) out[0] = idx;
}

static double run(const float4* recs, uint32_t mask, uint32_t hot_mask, double p_hot, int affine, uint32_t steps, uint32_t* out, int n_cus, int wg_per_cu) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // 160 KB of LDS per CU: the static 24 KB + this padding makes exactly wg_per_cu workgroups fit
    const size_t pad = wg_per_cu >= 6 ? 16 : (size_t)(160 * 1024 / wg_per_cu) - 24 * 1024 - 1024;
    hipFuncSetAttribute((const void*)k_walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
    const dim3 grid(n_cus * wg_per_cu), block(256);
    const uint32_t thresh = (uint32_t)(p_hot * 65536.0);
    hipLaunchKernelGGL(k_walk, grid, block, pad, 0, recs, mask, hot_mask, thresh, affine, steps / 8, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_walk, grid, block, pad, 0, recs, mask, hot_mask, thresh, affine, steps, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (double)grid.x * 256.0 * steps / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const uint32_t steps = argc > 1 ? (uint32_t)atoi(argv[1]) : 1500;
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    const uint32_t n_max = 1u << 22;   // 512 MB of records
    std::vector<float> h((size_t)n_max * 32);
    uint64_t st = 88172645463325252ull;
    for (auto& v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (float)(st & 0xffffff) / 16777216.0f; }
    float4* recs; uint32_t* out;
    hipMalloc((void**)&recs, h.size() * 4); hipMalloc((void**)&out, 64);
    hipMemcpy(recs, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int cus = pr.multiProcessorCount;
    const uint32_t m64 = (1u << 19) - 1;
    printf("device %s, %d CUs; 128-byte records, K = 7 loads + ~100 VALU per step, %u steps per lane; G steps/s = G lines/s\n", pr.name, cus, steps);
    printf("(a) occupancy, uniform walk over 64 MB:");
    for (int w : {1, 2, 3, 4, 5, 8}) printf("  %d WG/CU %.1f", w, run(recs, m64, 0, 0.0, 0, steps, out, cus, w));
    printf("\n(c) working set, uniform walk, 5 WG/CU:");
    for (int lg = 13; lg <= 22; lg++) printf("  %u MB %.1f", (1u << lg) / 8192u, run(recs, (1u << lg) - 1, 0, 0.0, 0, steps, out, cus, 5));
    const uint32_t hot = (1u << 13) - 1;   // 1 MB
    printf("\n(b) hot fraction p of the steps in a 1 MB subset, the rest uniform over 64 MB, 5 WG/CU\n    shared hot set  :");
    for (double p : {0.0, 0.25, 0.5, 0.63, 0.75, 0.85, 0.95, 1.0}) printf("  p=%.2f %.1f", p, run(recs, m64, hot, p, 0, steps, out, cus, 5));
    printf("\n    per-XCD hot sets:");
    for (double p : {0.0, 0.25, 0.5, 0.63, 0.75, 0.85, 0.95, 1.0}) printf("  p=%.2f %.1f", p, run(recs, m64, hot, p, 1, steps, out, cus, 5));
    printf("\n(b') the same at 3 WG/CU, per-XCD hot sets:");
    for (double p : {0.0, 0.5, 0.75, 0.95}) printf("  p=%.2f %.1f", p, run(recs, m64, hot, p, 1, steps, out, cus, 3));
    printf("\n");
    return 0;
}
