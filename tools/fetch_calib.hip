// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes of the pair kernel:
// known byte counts, arrays far larger than the 256 MiB on-die cache, each byte touched once.
//   k_stream16   : 16 B per lane, consecutive lanes consecutive (state-n arrays, candidate rows)
//   k_stream8    : 8 B per lane (double arrays read one value per lane)
//   k_quad_gather: 4 lanes fetch one 64-byte record at a random position (neighbour records)
//   k_store16    : 16 B per lane streaming store
//   k_store8     : 8 B per lane streaming store
// build: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o gpurun_out/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void k_stream16(const uint4* __restrict__ a, size_t n, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = a[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_stream8(const double* __restrict__ a, size_t n, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += a[i];
    if (acc == 1.2345) *sink = 1;
}
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
// the same streams with the non-temporal hint the pair kernel puts on its streamed loads
__global__ void k_stream16_nt(const u32x4_t* __restrict__ a, size_t n, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u32x4_t v = __builtin_nontemporal_load(a + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_stream8_nt(const double* __restrict__ a, size_t n, uint32_t* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += __builtin_nontemporal_load(a + i);
    if (acc == 1.2345) *sink = 1;
}
__global__ void k_store16_nt(u32x4_t* __restrict__ a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        u32x4_t v = {(uint32_t)i, 1u, 2u, 3u};
        __builtin_nontemporal_store(v, a + i);
    }
}
__global__ void k_quad_gather(const uint4* __restrict__ rec, const uint32_t* __restrict__ perm, size_t nrec, uint32_t* sink) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t q = threadIdx.x & 3;
    uint32_t acc = 0;
    for (size_t r = t >> 2; r < nrec; r += ((size_t)gridDim.x * blockDim.x) >> 2) {
        uint4 v = rec[(size_t)perm[r] * 4 + q];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void k_store16(uint4* __restrict__ a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_uint4((uint32_t)i, 1, 2, 3);
}
__global__ void k_store8(double* __restrict__ a, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)i;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const size_t bytes = (size_t)2 << 30;  // 2 GiB per test array
    const size_t nrec = bytes / 64;
    void *a = nullptr, *b = nullptr;
    uint32_t *perm = nullptr, *sink = nullptr;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc((void**)&perm, nrec * 4));
    CK(hipMalloc((void**)&sink, 4));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 1, bytes));
    {
        // a permutation that scatters neighbours far apart: multiplicative hash of the index
        std::vector<uint32_t> h(nrec);
        const uint64_t mul = 2654435761ull;  // odd, nrec is a power of two => a bijection
        for (size_t i = 0; i < nrec; ++i) h[i] = (uint32_t)((i * mul) & (nrec - 1));
        CK(hipMemcpy(perm, h.data(), nrec * 4, hipMemcpyHostToDevice));
    }
    dim3 g(256 * 16), t(256);
    // the memsets above may leave lines on die: flush with a pass over the other array in between
    hipLaunchKernelGGL(k_stream16, g, t, 0, 0, (const uint4*)b, bytes / 16, sink);
    hipLaunchKernelGGL(k_stream16, g, t, 0, 0, (const uint4*)a, bytes / 16, sink);   // 2 GiB read
    hipLaunchKernelGGL(k_stream8, g, t, 0, 0, (const double*)b, bytes / 8, sink);    // 2 GiB read
    hipLaunchKernelGGL(k_quad_gather, g, t, 0, 0, (const uint4*)a, perm, nrec, sink); // 2 GiB + perm 128 MiB
    hipLaunchKernelGGL(k_store16, g, t, 0, 0, (uint4*)b, bytes / 16);                // 2 GiB written
    hipLaunchKernelGGL(k_store8, g, t, 0, 0, (double*)a, bytes / 8);                 // 2 GiB written
    hipLaunchKernelGGL(k_stream16_nt, g, t, 0, 0, (const u32x4_t*)b, bytes / 16, sink); // 2 GiB read
    hipLaunchKernelGGL(k_stream8_nt, g, t, 0, 0, (const double*)a, bytes / 8, sink);    // 2 GiB read
    hipLaunchKernelGGL(k_store16_nt, g, t, 0, 0, (u32x4_t*)b, bytes / 16);              // 2 GiB written
    CK(hipDeviceSynchronize());
    printf("expected KiB: stream16/stream8 read %zu; quad_gather read %zu (+%zu perm); stores %zu\n", bytes >> 10, bytes >> 10,
           (nrec * 4) >> 10, bytes >> 10);
    return 0;
}
