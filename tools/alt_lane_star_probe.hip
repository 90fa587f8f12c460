// Design experiment (NOT part of the product): the decomposition the north-star sketch describes -- lane = star,
// coalesced vector loads of SoA columns, per-walker wavefront shuffle reduction, LDS-staged per-block partial sums --
// against the shipped decomposition (lane = walker, wave-uniform scalar record loads, no cross-lane traffic), for the
// same arithmetic (fraction tree over 8 stars + log-product, MODEL_CONST, fixed centre, f64).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I mcmc_dynamics_amd/csrc tools/alt_lane_star_probe.hip -o tools/alt_lane_star_probe
//   ./tools/alt_lane_star_probe [n_stars] [n_walkers]
//
// Both kernels produce per-(block, walker) partial log-likelihoods that are summed on the host and compared.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "mcd_math.h"

using namespace mcd;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kStarsPerLane = 8;
constexpr int kBlock = 256;
constexpr int kStarsPerBlock = kBlock * kStarsPerLane;     // 2048

__device__ __forceinline__ double shfl_xor_f64(double x, int m) { return __shfl_xor(x, m, 64); }

// ---------------------------------------------------------------------------------------------------------------
// Variant A (north-star sketch): lane = star.  Each lane keeps 8 stars in registers (coalesced loads), loops over all
// walkers (parameters are wave-uniform -> SGPRs), reduces each walker's (sum q/n, prod n) over the 64 lanes with
// xor-shuffles, lane 0 adds into an LDS row per wave; the 4 wave rows are combined at the end.
__global__ __launch_bounds__(kBlock) void lane_star_kernel(const double* __restrict__ v, const double* __restrict__ e2,
                                                            const double* __restrict__ sn, const double* __restrict__ cs,
                                                            int64_t n, const double* __restrict__ wpar, int W,
                                                            double* __restrict__ partials) {
    extern __shared__ double lds[];                 // [4 waves][64 walkers][2]: sum q/n, sum log n
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double sv[kStarsPerLane], se[kStarsPerLane], ss[kStarsPerLane], sc[kStarsPerLane];
    const int64_t base = (int64_t)blockIdx.x * kStarsPerBlock;
#pragma unroll
    for (int j = 0; j < kStarsPerLane; ++j) {
        const int64_t i = base + (int64_t)j * kBlock + threadIdx.x;          // coalesced across the block
        const bool ok = i < n;
        sv[j] = ok ? v[i] : 0.0; se[j] = ok ? e2[i] : 1.0; ss[j] = ok ? sn[i] : 0.0; sc[j] = ok ? cs[i] : 0.0;
    }
    // blockIdx.y selects a group of 64 walkers so that the grid has as many waves as variant B
    const int w0 = blockIdx.y * 64, w1 = (w0 + 64 < W) ? w0 + 64 : W;
    for (int w = w0; w < w1; ++w) {
        const double* p = wpar + (int64_t)w * KD;                              // wave-uniform -> scalar loads
        const double vsys = p[W_VSYS], s2 = p[W_S2], vx = p[W_VX], vy = p[W_VY];
        double qq[kStarsPerLane], nn[kStarsPerLane];
#pragma unroll
        for (int j = 0; j < kStarsPerLane; ++j) {
            const double d = fma_(-vx, ss[j], fma_(vy, sc[j], sv[j] - vsys));
            const bool ok = base + (int64_t)j * kBlock + threadIdx.x < n;
            qq[j] = ok ? d * d : 0.0;
            nn[j] = ok ? se[j] + s2 : 1.0;
        }
        const Frac f = frac_join(frac_join(frac_leaf2(qq[0], nn[0], qq[1], nn[1]), frac_leaf2(qq[2], nn[2], qq[3], nn[3])),
                                 frac_join(frac_leaf2(qq[4], nn[4], qq[5], nn[5]), frac_leaf2(qq[6], nn[6], qq[7], nn[7])));
        double q = f.num * rcp_nr(f.den);
        // product of 64 lane denominators would overflow: reduce log-domain pieces (mantissa product + exponent sum)
        int ex;
        double m = __builtin_frexp(f.den, &ex);
        double e = (double)ex;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            q += shfl_xor_f64(q, off);
            m *= shfl_xor_f64(m, off);                                          // 64 mantissas in [0.5, 1): >= 2^-64
            e += shfl_xor_f64(e, off);
        }
        if (lane == 0) {
            double* row = lds + ((size_t)wave * 64 + (w - w0)) * 2;
            row[0] = q;
            row[1] = fma_(e, kLn2, log(m));
        }
    }
    __syncthreads();
    if (threadIdx.x < w1 - w0) {
        const int t = threadIdx.x;
        double q = 0.0, l = 0.0;
        for (int k = 0; k < 4; ++k) { q += lds[((size_t)k * 64 + t) * 2]; l += lds[((size_t)k * 64 + t) * 2 + 1]; }
        partials[(int64_t)blockIdx.x * W + w0 + t] = -0.5 * (q + l);            // count * log(2 pi) left out in both variants
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Variant B (shipped design, reduced to its core): lane = walker, AoS records through the scalar path.
__global__ __launch_bounds__(kBlock) void lane_walker_kernel(const double* __restrict__ recs, int64_t n, int chunk_len,
                                                              const double* __restrict__ wpar, int W, int n_wtiles,
                                                              int64_t n_tasks, double* __restrict__ partials) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int64_t task = (int64_t)blockIdx.x * 4 + wave;
    if (task >= n_tasks) return;
    const int64_t chunk = task / n_wtiles;
    const int wt = (int)(task - chunk * n_wtiles);
    const int w = wt * 64 + lane;
    const int wc = w < W ? w : W - 1;
    WalkerConsts<double> c;
    c.load(wpar + (int64_t)wc * KD);
    const int64_t begin = chunk * chunk_len;
    const int count = (int)((n - begin) < chunk_len ? (n - begin) : chunk_len);
    bool denormal;
    const double r = chunk_loglike<MODEL_CONST, false, double, double, true>(recs + begin * 4, count, c, denormal, nullptr);
    if (w < W) partials[chunk * W + w] = r + 0.5 * count * kLn2Pi;              // same convention as variant A
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1000000;
    const int W = argc > 2 ? atoi(argv[2]) : 256;
    std::mt19937_64 g(3);
    std::normal_distribution<double> nv(0.0, 10.0);
    std::uniform_real_distribution<double> ua(-3.14159, 3.14159), ue(0.3, 3.0);
    std::vector<double> v(n), e2(n), sn(n), cs(n), recs(4 * n), wpar((size_t)W * KD, 0.0);
    for (int64_t i = 0; i < n; ++i) {
        const double th = ua(g), e = ue(g);
        v[i] = nv(g); e2[i] = e * e; sn[i] = std::sin(th); cs[i] = std::cos(th);
        recs[4 * i] = v[i]; recs[4 * i + 1] = e2[i]; recs[4 * i + 2] = sn[i]; recs[4 * i + 3] = cs[i];
    }
    for (int w = 0; w < W; ++w) {
        double* p = &wpar[(size_t)w * KD];
        p[W_VSYS] = 0.1 * nv(g) * 0.1; p[W_S2] = std::pow(10.0 * (1 + 0.05 * nv(g) / 10.0), 2); p[W_VX] = 3 + 0.01 * nv(g); p[W_VY] = -4 + 0.01 * nv(g);
        p[W_CAC] = p[W_CDC] = 1.0;
    }
    double *dv, *de, *ds, *dc, *dr, *dw, *pa, *pb;
    const int64_t blocks_a = (n + kStarsPerBlock - 1) / kStarsPerBlock;
    const int n_wtiles = (W + 63) / 64;
    int chunk_len = (int)((n * n_wtiles + 12287) / 12288);
    chunk_len = (chunk_len + 31) / 32 * 32; if (chunk_len < 64) chunk_len = 64;
    const int64_t n_chunks = (n + chunk_len - 1) / chunk_len, n_tasks = n_chunks * n_wtiles;
    CHECK(hipMalloc(&dv, n * 8)); CHECK(hipMalloc(&de, n * 8)); CHECK(hipMalloc(&ds, n * 8)); CHECK(hipMalloc(&dc, n * 8));
    CHECK(hipMalloc(&dr, n * 32)); CHECK(hipMalloc(&dw, wpar.size() * 8));
    CHECK(hipMalloc(&pa, blocks_a * W * 8)); CHECK(hipMalloc(&pb, n_chunks * W * 8));
    CHECK(hipMemcpy(dv, v.data(), n * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(de, e2.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(ds, sn.data(), n * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dc, cs.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dr, recs.data(), n * 32, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dw, wpar.data(), wpar.size() * 8, hipMemcpyHostToDevice));
    const size_t lds_bytes = (size_t)4 * 64 * 2 * 8;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto run_a = [&]() { hipLaunchKernelGGL(lane_star_kernel, dim3((unsigned)blocks_a, (unsigned)n_wtiles), dim3(kBlock), lds_bytes, 0, dv, de, ds, dc, n, dw, W, pa); };
    auto run_b = [&]() { hipLaunchKernelGGL(lane_walker_kernel, dim3((unsigned)((n_tasks + 3) / 4)), dim3(kBlock), 0, 0, dr, n, chunk_len, dw, W, n_wtiles, n_tasks, pb); };
    float ms_a = 0, ms_b = 0;
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 5; ++i) run_a();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); for (int i = 0; i < 50; ++i) run_a(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms_a, e0, e1));
        for (int i = 0; i < 5; ++i) run_b();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); for (int i = 0; i < 50; ++i) run_b(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms_b, e0, e1));
    }
    CHECK(hipGetLastError());
    std::vector<double> ha(blocks_a * W), hb(n_chunks * W);
    CHECK(hipMemcpy(ha.data(), pa, ha.size() * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hb.data(), pb, hb.size() * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int w = 0; w < W; ++w) {
        long double sa = 0, sb = 0;
        for (int64_t b = 0; b < blocks_a; ++b) sa += ha[b * W + w];
        for (int64_t c = 0; c < n_chunks; ++c) sb += hb[c * W + w];
        worst = std::fmax(worst, std::fabs((double)((sa - sb) / sb)));
    }
    const double terms = (double)n * W;
    printf("N = %lld stars, W = %d walkers (%.3g terms per launch)\n", (long long)n, W, terms);
    printf("A  lane = star   (coalesced loads, shuffle reduction per walker, LDS rows): %8.1f us/launch  %.3e terms/s  (%lld blocks)\n",
           ms_a / 50 * 1e3, terms / (ms_a / 50 * 1e-3), (long long)blocks_a);
    printf("B  lane = walker (scalar record loads, no cross-lane traffic)             : %8.1f us/launch  %.3e terms/s  (%lld chunks)\n",
           ms_b / 50 * 1e3, terms / (ms_b / 50 * 1e-3), (long long)n_chunks);
    printf("max relative difference of the per-walker sums: %.2e\n", worst);
    return 0;
}
