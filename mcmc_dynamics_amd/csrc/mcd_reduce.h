// mcd_reduce.h -- the fixed-order sum of one walker group's partial sums, shared by the reduction kernel
// (mcd_kernels.hip: reduce_group_kernel) and the resident chain's step kernel (mcd_stretch.hip), which adds up the partial
// sums of small launches itself: both get the same bits from the same code.  Device-only.
//
// Reference counterpart: the two np.sum calls of analysis/runner.py:269-270 (pairwise summation on one core).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

namespace mcd {

constexpr int kPartialGroup = 8;            // walkers per group of the partial-sum array (8 doubles = one 64-byte segment)

// reduce_sublane_sum: what thread (s, q) of a team of 2 SUB sublanes adds up; the caller combines the sublanes.
// ONE_ROUND: the caller guarantees c1 - c0 <= 2 SUB x U, i.e. every thread has at most U slots -- they are loaded straight
// into the accumulators (half the registers: the step kernel runs 1024 threads, 128 VGPRs each).
template <int SUB, int U, bool ONE_ROUND = false>
__device__ __forceinline__ void reduce_sublane_sum(const double2* __restrict__ col, int64_t c0, int64_t c1, int s,
                                                   double& ax, double& ay) {
    constexpr int kSublanes = 2 * SUB;
    static_assert((U & (U - 1)) == 0 && U >= 2, "the combining tree below halves U");
    double2 a[U];
    if constexpr (ONE_ROUND) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t cu = c0 + s + (int64_t)u * kSublanes;
            a[u] = cu < c1 ? col[cu * 4] : make_double2(0.0, 0.0);
        }
    } else {
#pragma unroll
    for (int u = 0; u < U; ++u) a[u] = make_double2(0.0, 0.0);
    for (int64_t c = c0 + s; c < c1; c += (int64_t)U * kSublanes) {
        double2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                                  // all loads of the round first
            const int64_t cu = c + (int64_t)u * kSublanes;
            v[u] = cu < c1 ? col[cu * 4] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u].x += v[u].x;
            a[u].y += v[u].y;
        }
    }
    }
#pragma unroll
    for (int w = U / 2; w >= 1; w >>= 1) {
#pragma unroll
        for (int u = 0; u < w; ++u) {
            a[u].x += a[u + w].x;
            a[u].y += a[u + w].y;
        }
    }
    ax = a[0].x;
    ay = a[0].y;
}

// the 16 sublanes of a wave (lanes q, q + 4, ..., q + 60) -> lanes 0 .. 3 hold the wave's sums of walkers (2q, 2q + 1)
__device__ __forceinline__ void reduce_wave_combine(double& ax, double& ay) {
#pragma unroll
    for (int off = 32; off >= 4; off >>= 1) {
        ax += __shfl_xor(ax, off, 64);
        ay += __shfl_xor(ay, off, 64);
    }
}


}  // namespace mcd
