// Sensitivity model of k_trace_w4's node step (experiments/w8_model/README.md): what a traversal step costs on gfx950 as a function of
//   K = scattered 16-byte loads per lane and step (k_trace_w4: 7 = one 128-byte four-box record; an 8-wide quantised record: 6)
//   V = dependent VALU work per step (k_trace_w4: ~140 instructions; an 8-wide step: ~255, see README)
// with everything else as in the production kernel: 256-thread workgroups, 5 per CU (30 KB of LDS each, <= 96 VGPRs), one ray per lane,
// every lane walking its own dependent chain through a 64 MB record array (C2's records: 0.5 M x 128 B), an LDS stack column written and
// read once per step.  Not a traversal: the next record index is a hash of what was loaded, so the chain is as dependent and as
// incoherent as a ray's.  Prints G steps / s for each (K, V).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

template <int K, int V, int STRIDE = 8>
__global__ __launch_bounds__(256) void k_step(const float4* __restrict__ recs, uint32_t mask, uint32_t steps, uint32_t* out) {
    __shared__ uint2 stack[12 * 256];
    __shared__ float4 pad[384];   // brings the block to ~30 KB: five workgroups per CU like k_trace_w4
    if (threadIdx.x == 0) pad[blockIdx.x % 384] = make_float4(0, 0, 0, 0);
    uint2* my = stack + threadIdx.x;
    uint32_t idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u & mask;
    float acc = 1.0f, ox = 0.37f, ix = 1.13f;
    uint32_t sp = 0;
    for (uint32_t s = 0; s < steps; s++) {
        const float4* p = recs + (size_t)idx * STRIDE;
        float4 a[K];
#pragma unroll
        for (int j = 0; j < K; j++) a[j] = p[j];
        // ~V VALU instructions over ALL loaded values (slab-test shaped: sub, mul, min, max on the (x, z) and (y, w) pairs of every 16-byte piece)
        float m0 = -1e30f, m1 = 1e30f;
#pragma unroll
        for (int r = 0; r < V / (14 * K) + 1; r++) {
#pragma unroll
            for (int j = 0; j < K; j++) {
                const float lo = (a[j].x - ox) * ix, hi = (a[j].z - ox) * ix, lo2 = (a[j].y - ox) * ix, hi2 = (a[j].w - ox) * ix;   // 8 instructions
                m0 = fmaxf(m0, fmaxf(fminf(lo, hi), fminf(lo2, hi2)) + (float)r);                                              // + 4 (5 with r)
                m1 = fminf(m1, fminf(fmaxf(lo, hi), fmaxf(lo2, hi2)));                                                          // + 4: ~14 per (r, j)
            }
            ox += 1e-3f;
        }
        acc = acc * 0.999f + (m0 < m1 ? 1e-7f : 2e-7f);
        my[(sp % 12u) * 256u] = make_uint2(idx, __float_as_uint(m0));
        sp++;
        const uint2 e = my[((sp + 5u) % 12u) * 256u];
        uint32_t h = __float_as_uint(a[K - 1].w) ^ (e.x * 0x9E3779B9u) ^ (s * 0x85EBCA6Bu) ^ __float_as_uint(acc);
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
        idx = h & mask;
    }
    if (acc == 12345.0f) out[0] = idx;   // keeps everything live
}

template <int K, int V, int STRIDE = 8>
double run(const float4* recs, uint32_t mask, uint32_t steps, uint32_t* out, int n_cus) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid(n_cus * 5), block(256);
    hipLaunchKernelGGL((k_step<K, V, STRIDE>), grid, block, 0, 0, recs, mask, steps / 8, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_step<K, V, STRIDE>), grid, block, 0, 0, recs, mask, steps, out);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)grid.x * 256.0 * steps / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
    const uint32_t n_rec = 1u << 19;   // 64 MB
    const uint32_t steps = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000;
    hipDeviceProp_t pr;
    hipGetDeviceProperties(&pr, 0);
    std::vector<float> h((size_t)n_rec * 32);
    uint64_t st = 88172645463325252ull;
    for (auto& v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (float)(st & 0xffffff) / 16777216.0f; }
    float4* recs; uint32_t* out;
    hipMalloc((void**)&recs, h.size() * 4); hipMalloc((void**)&out, 64);
    hipMemcpy(recs, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    printf("device %s, %d CUs; %u records of 128 B; %u steps per lane; G steps/s\n", pr.name, pr.multiProcessorCount, n_rec, steps);
#define ROW(K) printf("K=%d loads/step:  V=0 %.1f   V=100 %.1f   V=200 %.1f   V=300 %.1f   V=400 %.1f\n", K, \
        run<K, 0>(recs, n_rec - 1, steps, out, pr.multiProcessorCount), run<K, 100>(recs, n_rec - 1, steps, out, pr.multiProcessorCount), \
        run<K, 200>(recs, n_rec - 1, steps, out, pr.multiProcessorCount), run<K, 300>(recs, n_rec - 1, steps, out, pr.multiProcessorCount), \
        run<K, 400>(recs, n_rec - 1, steps, out, pr.multiProcessorCount));
    ROW(7) ROW(6) ROW(5) ROW(4) ROW(3)
    printf("K=4, records of 64 B (stride 4 x 16 B, 2^20 of them = the same 64 MB): V=0 %.1f  V=100 %.1f  V=200 %.1f;  2^19 of them (32 MB): V=100 %.1f\n", run<4, 0, 4>(recs, 2 * n_rec - 1, steps, out, pr.multiProcessorCount),
           run<4, 100, 4>(recs, 2 * n_rec - 1, steps, out, pr.multiProcessorCount), run<4, 200, 4>(recs, 2 * n_rec - 1, steps, out, pr.multiProcessorCount), run<4, 100, 4>(recs, n_rec - 1, steps, out, pr.multiProcessorCount));
    return 0;
}
