// Development probe: how long one wave, alone on its SIMD, takes per FP64 instruction (dependent
// chain, four independent chains, divisions), in nanoseconds of the 100 MHz wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, unsigned long long* t, int n, double seed) {
    double a = seed + threadIdx.x, b = 1.0000001, c = 1e-9;
    unsigned long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) { a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); a = fma(a, b, c); }
    unsigned long long t1 = wall_clock64();
    double p = seed, q = seed + 1, r = seed + 2, s = seed + 3;
    for (int i = 0; i < n; ++i) { p = fma(p, b, c); q = fma(q, b, c); r = fma(r, b, c); s = fma(s, b, c); }
    unsigned long long t2 = wall_clock64();
    double d = seed + 5;
    for (int i = 0; i < n; ++i) { d = 1.0 / (d + c); d = 1.0 / (d + c); d = 1.0 / (d + c); d = 1.0 / (d + c); }
    unsigned long long t3 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + p + q + r + s + d;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; }
}
int main() {
    double* out; unsigned long long* t; unsigned long long h[3];
    hipMalloc(&out, 8 * 64 * 256); hipMalloc(&t, 24);
    for (int grid : {1, 256}) for (int rep = 0; rep < 2; ++rep) {
        const int n = 20000;
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, t, n, 1.5);
        hipMemcpy(h, t, 24, hipMemcpyDeviceToHost);
        printf("grid %d: dependent fma %.2f ns each, 4 independent fma %.2f ns each, dependent division %.1f ns each\n", grid,
               h[0] * 10.0 / (4.0 * n), h[1] * 10.0 / (4.0 * n), h[2] * 10.0 / (4.0 * n));
    }
    return 0;
}
