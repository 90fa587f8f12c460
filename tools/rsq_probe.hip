// Scratch probe: accuracy of v_rsq_f64 / v_rcp_f64 on gfx950 (decides how many Newton steps the kernels need).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

__global__ void k(const double* x, double* rsq, double* rcp, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { rsq[i] = __builtin_amdgcn_rsq(x[i]); rcp[i] = __builtin_amdgcn_rcp(x[i]); }
}

int main() {
    const int n = 1 << 22;
    std::vector<double> h(n), a(n), b(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-60.0, 60.0);
    for (auto& v : h) v = std::exp2(u(g));
    double *dx, *da, *db;
    hipMalloc(&dx, n * 8); hipMalloc(&da, n * 8); hipMalloc(&db, n * 8);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, da, db, n);
    hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    long double e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        long double r1 = 1.0L / sqrtl((long double)h[i]), r2 = 1.0L / (long double)h[i];
        e1 = fmaxl(e1, fabsl((a[i] - r1) / r1));
        e2 = fmaxl(e2, fabsl((b[i] - r2) / r2));
    }
    printf("v_rsq_f64 max rel err %.3Le (2^%.1Lf)   v_rcp_f64 max rel err %.3Le (2^%.1Lf)\n", e1, log2l(e1), e2, log2l(e2));
    return 0;
}
