// mcd_kde.hip -- background.SingleStars: kernel-density background log-likelihood of N test stars against M
// comparison stars (reference: background/single_stars.py:42-77), the O(N M) precompute in front of the
// fixed-background likelihood (runner.py:96-106).
//
//   lnL_i = max_j e_ij + log( sum_j exp(e_ij - max_j e_ij) / sqrt(2 pi n_i) ) - log M,
//   e_ij  = -(c_j - v_i)^2 / (2 n_i),   n_i = verr_i^2 + sigma_int^2
//
// Mapping (same as the main kernel): lane = test star, the comparison velocities are wave-uniform and arrive through
// the scalar cache; a wave evaluates 64 test stars against one slice of the comparison stars.  Per slice the wave
// makes two passes over the (scalar-cached) slice: the nearest comparison star (largest exponent, exact), then the
// kernel sum about it.  Slices are combined in a fixed order by kde_combine_kernel: no atomics, bitwise repeatable.
// Bound: f64 VALU issue (about 19 f64 instructions per (i, j) pair, the 2^(j/256) lookups on the LDS pipe); HBM traffic is N * 24 B + slices * M * 8 B.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "mcd_internal.h"
#include "mcd_math.h"

namespace mcd {
namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr int kWavesPerBlock = kBlock / kWave;

__device__ const double kExpTabDevice[kExpTabSize] = {MCD_EXP_TABLE_VALUES};

__global__ __launch_bounds__(kBlock) void kde_slice_kernel(const double* __restrict__ comp, int64_t m,
                                                            const double* __restrict__ v,
                                                            const double* __restrict__ verr, int64_t n,
                                                            double sigma_int2, int slice_len, int n_slices,
                                                            double* __restrict__ part_dmin,
                                                            double* __restrict__ part_sum) {
    __shared__ double exptab[kExpTabSize];
    static_assert(kExpTabSize % kBlock == 0, "whole table entries per thread");
#pragma unroll
    for (int i = 0; i < kExpTabSize / kBlock; ++i) exptab[i * kBlock + threadIdx.x] = kExpTabDevice[i * kBlock + threadIdx.x];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    const int64_t task = (int64_t)blockIdx.x * kWavesPerBlock + wave;
    const int64_t n_tiles = (n + kWave - 1) / kWave;
    if (task >= n_tiles * n_slices) return;
    const int slice = (int)(task % n_slices);           // neighbouring waves share a tile: their v/verr loads hit L2
    const int64_t tile = task / n_slices;
    const int64_t i = tile * kWave + lane;
    const int64_t ic = i < n ? i : n - 1;
    const int64_t j0 = (int64_t)slice * slice_len;
    const int count = (int)((m - j0) < slice_len ? (m - j0) : slice_len);
    const double* __restrict__ c = comp + j0;            // wave-uniform

    KdeLane a;
    a.init(v[ic], verr[ic], sigma_int2);

    double dmin = __builtin_inf();
    int j = 0;
    for (; j + 8 <= count; j += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a.nearest(c[j + q], dmin);
    }
    for (; j < count; ++j) a.nearest(c[j], dmin);
    a.begin_sum(dmin);
    j = 0;
    for (; j + 8 <= count; j += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) a.add(c[j + q], exptab);
    }
    for (; j < count; ++j) a.add(c[j], exptab);
    if (i < n) {
        part_dmin[(int64_t)slice * n + i] = dmin;
        part_sum[(int64_t)slice * n + i] = a.sum;
    }
}

// out_i = -dmin^2 h + log( sum_s S_s exp((dmin^2 - dmin_s^2) h) / sqrt(2 pi n_i) ) - log M
__global__ __launch_bounds__(kBlock) void kde_combine_kernel(const double* __restrict__ part_dmin,
                                                              const double* __restrict__ part_sum,
                                                              const double* __restrict__ verr, int64_t n,
                                                              double sigma_int2, int n_slices, double log_m,
                                                              double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double e = verr[i];
    const double norm = fma_(e, e, sigma_int2);
    const double h = 0.5 / norm;
    double dmin = part_dmin[i];
    for (int s = 1; s < n_slices; ++s) dmin = __builtin_fmin(dmin, part_dmin[(int64_t)s * n + i]);
    const double d2 = dmin * dmin;
    double total = 0.0;
    for (int s = 0; s < n_slices; ++s) {
        const double ds = part_dmin[(int64_t)s * n + i];
        total += part_sum[(int64_t)s * n + i] * exp(fma_(-ds, ds, d2) * h);
    }
    out[i] = -d2 * h + log(total / sqrt(2.0 * 3.14159265358979323846 * norm)) - log_m;
}

}  // namespace

int kde_slices(int64_t n, int64_t m, int* slice_len) {
    // enough waves to fill 256 CUs x 4 SIMDs a few times over; slices no shorter than 512 comparison stars
    const int64_t n_tiles = (n + kWave - 1) / kWave;
    int64_t want = n_tiles > 0 ? (8192 + n_tiles - 1) / n_tiles : 1;
    int64_t max_slices = (m + 511) / 512;
    if (want > max_slices) want = max_slices;
    if (want < 1) want = 1;
    if (want > 4096) want = 4096;
    int64_t len = (m + want - 1) / want;
    len = (len + 7) / 8 * 8;
    *slice_len = (int)len;
    return (int)((m + len - 1) / len);
}

hipError_t launch_kde(hipStream_t s, const double* comp, int64_t m, const double* v, const double* verr, int64_t n,
                      double sigma_int, int slice_len, int n_slices, double* part_dmin, double* part_sum,
                      double* out) {
    if (n <= 0) return hipSuccess;
    const int64_t n_tiles = (n + kWave - 1) / kWave;
    const int64_t n_tasks = n_tiles * n_slices;
    const int64_t grid = (n_tasks + kWavesPerBlock - 1) / kWavesPerBlock;
    hipLaunchKernelGGL(kde_slice_kernel, dim3((unsigned)grid), dim3(kBlock), 0, s, comp, m, v, verr, n,
                       sigma_int * sigma_int, slice_len, n_slices, part_dmin, part_sum);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kde_combine_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, part_dmin,
                       part_sum, verr, n, sigma_int * sigma_int, n_slices, std::log((double)m), out);
    return hipGetLastError();
}

}  // namespace mcd
