// What do rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access shapes of the shade stage?  MI355X_MICROARCH.md calibrates FETCH_SIZE on ONE
// shape (a wide coalesced streaming read reports half its bytes) and leaves the others open; profiles/r05_pmc_traffic.json doubled FETCH_SIZE for every kernel.
// Each kernel below moves a KNOWN number of useful bytes through a buffer far larger than the 256 MiB Infinity Cache; the PMC passes of run.sh put the
// counters next to those byte counts (tools/per_dispatch.py), which gives the factor per shape.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// n accesses of W bytes, access i at byte offset STRIDE * i (dense when STRIDE == W)
template <int W, int STRIDE> __global__ void rd_stride(const uint8_t* __restrict__ buf, uint64_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t* p = buf + (uint64_t)STRIDE * i;
        if (W == 4) acc ^= *reinterpret_cast<const uint32_t*>(p);
        else for (int k = 0; k < W / 16; k++) { uint4 v = reinterpret_cast<const uint4*>(p)[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345u) *sink = acc;
}
// n accesses of W bytes at records of REC bytes chosen by a hash (every lane somewhere else): the tris[prim] / slot-after-many-bounces shape
template <int W, int REC> __global__ void rd_random(const uint8_t* __restrict__ buf, uint64_t n, uint32_t n_rec_mask, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint8_t* p = buf + (uint64_t)REC * (mix((uint32_t)i) & n_rec_mask);
        if (W == 4) acc ^= *reinterpret_cast<const uint32_t*>(p);
        else for (int k = 0; k < W / 16; k++) { uint4 v = reinterpret_cast<const uint4*>(p)[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345u) *sink = acc;
}
// a queue that keeps a fraction KEEP / 8 of the slots, in slot order: the path state after a few bounces (monotone, gappy)
template <int W, int KEEP> __global__ void rd_sparse(const uint8_t* __restrict__ buf, uint64_t n, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if ((mix((uint32_t)i) & 7u) >= (uint32_t)KEEP) continue;
        const uint8_t* p = buf + (uint64_t)W * i;
        if (W == 4) acc ^= *reinterpret_cast<const uint32_t*>(p);
        else for (int k = 0; k < W / 16; k++) { uint4 v = reinterpret_cast<const uint4*>(p)[k]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345u) *sink = acc;
}
template <int W, int STRIDE> __global__ void wr_stride(uint8_t* __restrict__ buf, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint8_t* p = buf + (uint64_t)STRIDE * i;
        if (W == 4) *reinterpret_cast<uint32_t*>(p) = (uint32_t)i;
        else for (int k = 0; k < W / 16; k++) reinterpret_cast<uint4*>(p)[k] = make_uint4((uint32_t)i, k, 2, 3);
    }
}
template <int W, int KEEP> __global__ void wr_sparse(uint8_t* __restrict__ buf, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        if ((mix((uint32_t)i) & 7u) >= (uint32_t)KEEP) continue;
        uint8_t* p = buf + (uint64_t)W * i;
        if (W == 4) *reinterpret_cast<uint32_t*>(p) = (uint32_t)i;
        else for (int k = 0; k < W / 16; k++) reinterpret_cast<uint4*>(p)[k] = make_uint4((uint32_t)i, k, 2, 3);
    }
}

int main() {
    const uint64_t bytes = 4ull << 30;   // 16 x the Infinity Cache
    uint8_t* buf; uint32_t* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, bytes));
    const dim3 g(256 * 16), b(256);
    const uint64_t n = 1ull << 26;   // accesses per kernel (dense 16-byte: 1 GiB)
    // name = kernel template arguments; useful bytes = n x W (x KEEP / 8 for the sparse forms)
    hipLaunchKernelGGL((rd_stride<16, 16>), g, b, 0, 0, buf, n, sink);       // 1 GiB dense, 16 B / lane
    hipLaunchKernelGGL((rd_stride<4, 4>), g, b, 0, 0, buf, n, sink);         // 256 MiB dense, 4 B / lane
    hipLaunchKernelGGL((rd_stride<32, 32>), g, b, 0, 0, buf, n, sink);       // 2 GiB dense 32-byte records (two 16-byte loads per lane)
    hipLaunchKernelGGL((rd_stride<16, 32>), g, b, 0, 0, buf, n, sink);       // 16 of every 32 bytes
    hipLaunchKernelGGL((rd_stride<16, 64>), g, b, 0, 0, buf, n, sink);       // 16 of every 64
    hipLaunchKernelGGL((rd_stride<4, 64>), g, b, 0, 0, buf, n, sink);        // 4 of every 64
    hipLaunchKernelGGL((rd_random<16, 16>), g, b, 0, 0, buf, n, (uint32_t)(bytes / 16 - 1), sink);
    hipLaunchKernelGGL((rd_random<48, 48>), g, b, 0, 0, buf, n, (uint32_t)((1u << 26) - 1), sink);   // 48-byte records (tris): 64 M records = 3 GiB
    hipLaunchKernelGGL((rd_random<4, 4>), g, b, 0, 0, buf, n, (uint32_t)(bytes / 4 - 1), sink);
    hipLaunchKernelGGL((rd_sparse<16, 4>), g, b, 0, 0, buf, 4 * n, sink);    // half of the slots alive: 2 GiB useful of 4 GiB spanned
    hipLaunchKernelGGL((rd_sparse<16, 2>), g, b, 0, 0, buf, 4 * n, sink);    // a quarter: 1 GiB useful of 4
    hipLaunchKernelGGL((rd_sparse<4, 4>), g, b, 0, 0, buf, 16 * n, sink);    // 4-byte field, half alive: 2 GiB useful of 4
    hipLaunchKernelGGL((rd_sparse<4, 2>), g, b, 0, 0, buf, 16 * n, sink);    // a quarter: 1 GiB useful of 4
    hipLaunchKernelGGL((wr_stride<16, 16>), g, b, 0, 0, buf, n);
    hipLaunchKernelGGL((wr_stride<4, 4>), g, b, 0, 0, buf, n);
    hipLaunchKernelGGL((wr_stride<32, 32>), g, b, 0, 0, buf, n);
    hipLaunchKernelGGL((wr_stride<16, 32>), g, b, 0, 0, buf, n);
    hipLaunchKernelGGL((wr_stride<16, 64>), g, b, 0, 0, buf, n);
    hipLaunchKernelGGL((wr_stride<4, 64>), g, b, 0, 0, buf, n);
    hipLaunchKernelGGL((wr_sparse<16, 4>), g, b, 0, 0, buf, 4 * n);
    hipLaunchKernelGGL((wr_sparse<16, 2>), g, b, 0, 0, buf, 4 * n);
    hipLaunchKernelGGL((wr_sparse<4, 4>), g, b, 0, 0, buf, 16 * n);
    hipLaunchKernelGGL((wr_sparse<4, 2>), g, b, 0, 0, buf, 16 * n);
    CK(hipDeviceSynchronize());
    printf("done\n");
    return 0;
}
