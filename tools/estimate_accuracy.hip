// Development check: relative error of the hardware v_rsq_f64 / v_rcp_f64 estimates and of the
// refined forms the pair kernel uses, against correctly rounded division and square root.
// build: hipcc --offload-arch=gfx950 -O3 tools/estimate_accuracy.hip -o tools/bin/estimate_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void k(const double* x, int n, double* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    double y = __builtin_amdgcn_rsq(v);
    out[i] = y;
    double e = fma(-(0.5 * v) * y, y, 0.5);
    double y1 = fma(y, e, y);
    out[n + i] = y1;
    e = fma(-(0.5 * v) * y1, y1, 0.5);
    out[2 * n + i] = fma(y1, e, y1);
    double r = __builtin_amdgcn_rcp(v);
    out[3 * n + i] = r;
    e = fma(-v, r, 1.0);
    double r1 = fma(r, e, r);
    out[4 * n + i] = r1;
    e = fma(-v, r1, 1.0);
    out[5 * n + i] = fma(r1, e, r1);
}

int main() {
    const int n = 1 << 22;
    std::vector<double> h(n);
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        h[i] = (i & 1) ? 900.0 + 200.0 * u : std::exp(-20.0 + 40.0 * u);
    }
    double *dx, *dout;
    hipMalloc(&dx, n * 8);
    hipMalloc(&dout, (size_t)6 * n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, n, dout);
    std::vector<double> o((size_t)6 * n);
    if (hipMemcpy(o.data(), dout, (size_t)6 * n * 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    const char* name[6] = {"rsq estimate", "rsq + 1 step", "rsq + 2 steps", "rcp estimate", "rcp + 1 step", "rcp + 2 steps"};
    for (int c = 0; c < 6; ++c) {
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            const long double ex = c < 3 ? 1.0L / sqrtl((long double)h[i]) : 1.0L / (long double)h[i];
            const long double err = fabsl(((long double)o[(size_t)c * n + i] - ex) / ex);
            if (err > worst) worst = err;
        }
        printf("%-14s max rel err %.3Le (2^%.1Lf)\n", name[c], worst, log2l(worst));
    }
    return 0;
}
