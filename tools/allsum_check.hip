// Development check: the row-swap form of the wave-wide sum gives the same bits as the shuffle form.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "../sph_amd/csrc/sph_types.h"
#include "../sph_amd/csrc/sph_kernels.h"
__device__ __forceinline__ double allsum_shuffle(double v) {
    v += quad_perm_d<QP_XOR1>(v);
    v += quad_perm_d<QP_XOR2>(v);
    v += quad_perm_d<0x141>(v);
    v += quad_perm_d<0x140>(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__global__ void k(const double* in, double* a, double* b) {
    const double v = in[threadIdx.x];
    a[threadIdx.x] = allsum_shuffle(v);
    b[threadIdx.x] = wave_allsum(v);
}
__global__ void k10(const double* in, double* out) {
    double v[10];
    for (int q = 0; q < 10; ++q) v[q] = in[q * 64 + threadIdx.x];
    wave_allsum10(v);
    for (int q = 0; q < 10; ++q) out[q * 64 + threadIdx.x] = v[q];
}
int check10() {
    static double h[640], o[640];
    double *d, *dout;
    hipMalloc(&d, sizeof h); hipMalloc(&dout, sizeof o);
    int bad = 0;
    for (int rep = 0; rep < 50; ++rep) {
        for (int i = 0; i < 640; ++i) h[i] = (double)(rand() % 2001 - 1000); /* integers: any order of additions is exact */
        hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k10, dim3(1), dim3(64), 0, 0, d, dout);
        hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
        for (int q = 0; q < 10; ++q) {
            double want = 0;
            for (int l = 0; l < 64; ++l) want += h[q * 64 + l];
            for (int l = 0; l < 64; ++l) if (o[q * 64 + l] != want) { if (bad < 5) printf("value %d lane %d: %g, want %g\n", q, l, o[q * 64 + l], want); ++bad; }
        }
    }
    printf("ten sums at once: %d mismatches\n", bad);
    return bad;
}
__global__ void kdiv(const double* num, const double* den, unsigned long long n, unsigned long long* bad) {
    unsigned long long b = 0;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const shared_divisor d(den[i]);
        const double a = d.quotient(num[i]), w = num[i] / den[i];
        if (__double_as_longlong(a) != __double_as_longlong(w)) ++b;
    }
    if (b) atomicAdd(bad, b);
}
int check_div() {
    const unsigned long long n = 10000000ull;
    double *hn = (double*)malloc(8 * n), *hd = (double*)malloc(8 * n), *dn, *dd;
    unsigned long long *dbad, hbad = 0;
    for (unsigned long long i = 0; i < n; ++i) {
        /* numerators: zero, or 1e-250 .. 1e3 of either sign; denominators 1e-3 .. 1e12 */
        const double en = -250.0 + 253.0 * (rand() / (double)RAND_MAX), ed = -3.0 + 15.0 * (rand() / (double)RAND_MAX);
        hn[i] = (rand() % 50 == 0) ? 0.0 : ((rand() & 1) ? -1.0 : 1.0) * (1.0 + rand() / (double)RAND_MAX) * pow(10.0, en);
        hd[i] = (1.0 + rand() / (double)RAND_MAX) * pow(10.0, ed);
    }
    hipMalloc(&dn, 8 * n); hipMalloc(&dd, 8 * n); hipMalloc(&dbad, 8);
    hipMemcpy(dn, hn, 8 * n, hipMemcpyHostToDevice); hipMemcpy(dd, hd, 8 * n, hipMemcpyHostToDevice);
    hipMemset(dbad, 0, 8);
    hipLaunchKernelGGL(kdiv, dim3(1024), dim3(256), 0, 0, dn, dd, n, dbad);
    hipMemcpy(&hbad, dbad, 8, hipMemcpyDeviceToHost);
    printf("quotients by a shared reciprocal against the division: %llu of %llu differ\n", hbad, n);
    return hbad != 0;
}
int main() {
    if (check10()) return 1;
    if (check_div()) return 1;
    double h[64], ha[64], hb[64], *d, *da, *db;
    hipMalloc(&d, 512); hipMalloc(&da, 512); hipMalloc(&db, 512);
    int bad = 0;
    for (int rep = 0; rep < 200; ++rep) {
        for (int i = 0; i < 64; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5) * pow(10.0, rand() % 12 - 6);
        hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, da, db);
        hipMemcpy(ha, da, 512, hipMemcpyDeviceToHost); hipMemcpy(hb, db, 512, hipMemcpyDeviceToHost);
        for (int i = 0; i < 64; ++i) if (memcmp(&ha[i], &hb[i], 8) || memcmp(&ha[i], &ha[0], 8)) ++bad;
    }
    printf("allsum check: %d mismatches\n", bad);
    return bad != 0;
}
