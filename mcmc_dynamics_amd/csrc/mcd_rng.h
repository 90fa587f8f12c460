// mcd_rng.h -- counter-based random numbers of the stretch move, the same code on the device (mcd_stretch.hip:
// chain_numbers_kernel fills a seeded block's numbers in device memory: nothing crosses PCIe) and on the host (the host-driven
// block of mcd_api.hip, mcd_chain_numbers, the CPU test harness tests/emul), so that a chain is a function of (seed, step,
// half step, ensemble, walker) alone and any step of it can be replayed anywhere.
//
//   * generator: Philox4x64-10 (Salmon et al. 2011), the algorithm of NumPy's `numpy.random.Philox` bit generator --
//     tests/test_chain_rng_cpu.py checks this implementation against `numpy.random.Philox(key=..., counter=...).random_raw`
//     for random keys and counters;
//   * uniform doubles as NumPy makes them: (x >> 11) 2^-53 in [0, 1);
//   * the logarithm of the acceptance threshold is NOT a libm call: libm's last bits differ between the device (ocml) and the
//     host (glibc), and a threshold that differs by an ulp can flip an accept decision.  `det_log` is a fixed sequence of IEEE
//     operations (division, fma), bit-identical wherever it is compiled with -ffp-contract=off; it is accurate to ~2 ulp,
//     which is all an acceptance threshold needs.
//
// What the reference does instead: emcee draws from NumPy's global Mersenne twister on the host (analysis/runner.py:59 seeds
// it; emcee's StretchMove: `random.shuffle`, `random.rand`, `random.randint`).
#pragma once

#include <algorithm>
#include <cstdint>
#include <utility>
#include <vector>

#include "mcd_math.h"   // MCD_HD, fma_

namespace mcd {

struct Philox4x64 { uint64_t v[4]; };

MCD_HD void mulhilo64(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
    lo = a * b;
#if defined(__HIP_DEVICE_COMPILE__)
    hi = __umul64hi(a, b);
#else
    hi = (uint64_t)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

// ten rounds; key schedule with the Weyl constants of the reference implementation (Random123 / NumPy)
MCD_HD Philox4x64 philox4x64_10(uint64_t c0, uint64_t c1, uint64_t c2, uint64_t c3, uint64_t k0, uint64_t k1) {
    constexpr uint64_t M0 = 0xD2E7470EE14C6C93ull, M1 = 0xCA5A826395121157ull;
    constexpr uint64_t W0 = 0x9E3779B97F4A7C15ull, W1 = 0xBB67AE8584CAA73Bull;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t hi0, lo0, hi1, lo1;
        mulhilo64(M0, c0, hi0, lo0);
        mulhilo64(M1, c2, hi1, lo1);
        const uint64_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox4x64 out;
    out.v[0] = c0; out.v[1] = c1; out.v[2] = c2; out.v[3] = c3;
    return out;
}

MCD_HD double uniform53(uint64_t x) { return (double)(x >> 11) * (1.0 / 9007199254740992.0); }

// log(x) for finite x > 0 as a fixed sequence of IEEE operations: x = 2^e m, m in [sqrt(1/2), sqrt(2)); f = (m - 1) / (m + 1),
// log m = 2 f (1 + s/3 + s^2/5 + ... + s^10/21), s = f^2 <= 0.0295 (remainder < 1e-18).  x == 0 -> -inf (an acceptance
// threshold of -inf accepts everything, as log(0) does for NumPy's draws).
MCD_HD double det_log(double x) {
    if (!(x > 0.0)) return -__builtin_huge_val();
    uint64_t bits;
    __builtin_memcpy(&bits, &x, sizeof bits);
    int e = (int)((bits >> 52) & 0x7ff);
    if (e == 0) {                                             // subnormal: scale by 2^64 first
        x *= 18446744073709551616.0;
        __builtin_memcpy(&bits, &x, sizeof bits);
        e = (int)((bits >> 52) & 0x7ff) - 64;
    }
    bits = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;          // mantissa in [1, 2)
    double m;
    __builtin_memcpy(&m, &bits, sizeof m);
    e -= 1023;
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double f = (m - 1.0) / (m + 1.0), s = f * f;
    double p = 1.0 / 21.0;
    p = fma_(p, s, 1.0 / 19.0);
    p = fma_(p, s, 1.0 / 17.0);
    p = fma_(p, s, 1.0 / 15.0);
    p = fma_(p, s, 1.0 / 13.0);
    p = fma_(p, s, 1.0 / 11.0);
    p = fma_(p, s, 1.0 / 9.0);
    p = fma_(p, s, 1.0 / 7.0);
    p = fma_(p, s, 1.0 / 5.0);
    p = fma_(p, s, 1.0 / 3.0);
    p = fma_(p, s, 1.0);
    return fma_((double)e, 0.693147180559945309417232121458, (2.0 * f) * p);
}

// ---- the numbers of a chain --------------------------------------------------------------------------------------
// ONE generator call per walker and step: counter = (step, half step h, ensemble b, slot j), key = (seed, a constant that
// names this use of the generator).  Its four words:
//   0 -> stretch factor z of slot j in half step h          1 -> that slot's acceptance uniform
//   2 -> that slot's partner index                           3 -> ordering key of walker w = h W/2 + j in this step
// The split of the ensemble in a step = the walkers sorted by (ordering key, walker index), where the ordering key is the
// word's upper 44 bits: the device ranks by counting with ONE unsigned comparison per pair on (key << 20 | walker), which is
// why ensembles are limited to 2^20 walkers here.
constexpr uint64_t kChainKey1 = 0x6d63645f636861ull;          // "mcd_cha"
constexpr int kOrderKeyShift = 20;
constexpr int64_t kSeededMaxWalkers = (int64_t)1 << kOrderKeyShift;

struct ChainDraw { double z, thr; int32_t pick; uint64_t order_key; };

// a = 2 (emcee's default): z = ((a - 1) u + 1)^2 / a, thr = log(u') - (P - 1) log z, partner = floor(u'' half) -- the
// operations of sampler.py's draw(), with det_log for the logarithms.  order_key: (upper 44 bits) << 20 | walker.
MCD_HD ChainDraw chain_draw(uint64_t seed, int64_t step, int h, int64_t b, int64_t j, int64_t half, int n_dim) {
    const Philox4x64 r = philox4x64_10((uint64_t)step, (uint64_t)h, (uint64_t)b, (uint64_t)j, seed, kChainKey1);
    ChainDraw d;
    double z = 1.0 * uniform53(r.v[0]) + 1.0;
    z *= z;
    z *= 0.5;
    d.z = z;
    d.thr = det_log(uniform53(r.v[1])) - (double)(n_dim - 1) * det_log(z);
    int64_t p = (int64_t)(uniform53(r.v[2]) * (double)half);
    d.pick = (int32_t)(p < half ? p : half - 1);
    d.order_key = ((r.v[3] >> kOrderKeyShift) << kOrderKeyShift) | (uint64_t)((int64_t)h * half + j);
    return d;
}

// Host: the numbers of ONE step in the layout mcd_stretch_move takes for a block of one step: order [B][W], zz / thr / pick
// [2][B][W/2].
inline void chain_numbers_of_step(uint64_t seed, int64_t step, int64_t B, int64_t W, int n_dim, int32_t* order, double* zz,
                                  double* thr, int32_t* pick, std::vector<uint64_t>& sorter) {
    const int64_t half = W / 2;
    sorter.resize((size_t)W);
    for (int64_t b = 0; b < B; ++b) {
        for (int h = 0; h < 2; ++h)
            for (int64_t j = 0; j < half; ++j) {
                const ChainDraw cd = chain_draw(seed, step, h, b, j, half, n_dim);
                const int64_t at = ((int64_t)h * B + b) * half + j;
                zz[at] = cd.z; thr[at] = cd.thr; pick[at] = cd.pick;
                sorter[(size_t)(h * half + j)] = cd.order_key;
            }
        std::sort(sorter.begin(), sorter.end());                       // (distinct: the walker index is part of the key)
        for (int64_t x = 0; x < W; ++x) order[b * W + x] = (int32_t)(sorter[(size_t)x] & (((uint64_t)1 << kOrderKeyShift) - 1));
    }
}

}  // namespace mcd
