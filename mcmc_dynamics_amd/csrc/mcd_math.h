// mcd_math.h -- per-term arithmetic of the log-likelihood kernels.
//
// Written once as host+device inline code so that the exact expression trees the gfx950 kernels
// execute can also be compiled for the CPU by tests/emul (test infrastructure only; the product
// never runs this on the CPU).
//
// Reference formulas (skamann/mcmc-dynamics):
//   v_los  = v_sys + v_max sin(theta - theta_0)                      analysis/constant.py:107-111
//          = v_sys + v_maxx sin(theta) - v_maxy cos(theta)
//   norm   = verr^2 + sigma^2 ; exponent = -1/2 (v - v_los)^2 / norm  analysis/runner.py:261-262
//   no background : lnL = -1/2 sum log(2 pi norm) + sum exponent      analysis/runner.py:269-271
//   background    : per-star log-sum-exp mixture                      analysis/runner.py:280-286
//   GB            : per-walker Gaussian background + density prior    analysis/constant.py:326-364
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#include "mcd_exp_table.h"

#if defined(__HIPCC__)
#define MCD_HD __host__ __device__ __forceinline__
#else
#define MCD_HD inline
#endif
// Star records are read through the CONSTANT address space on the device: a wave-uniform load from it is a scalar load
// (s_load_dwordxN) by construction.  From the global address space the compiler emits scalar loads only while it can
// prove that nothing in the kernel may have written the memory; an inline asm that takes the record pointer (the
// software prefetch below) ends that proof and the record reads silently become per-lane global_load instructions.
#if defined(__HIP_DEVICE_COMPILE__)
#define MCD_CONST_AS __attribute__((address_space(4)))
#else
#define MCD_CONST_AS
#endif
namespace mcd { template <class T> using RecPtr = const T MCD_CONST_AS*; }

// An empty volatile asm keeps the compiler from turning a small wave-uniform `if` into per-lane selects (v_cndmask on
// every iteration): the block stays behind a scalar branch.
#if defined(__HIP_DEVICE_COMPILE__)
#define MCD_KEEP_BRANCH() asm volatile("")
#else
#define MCD_KEEP_BRANCH() ((void)0)
#endif

namespace mcd {

constexpr double kLn2 = 0.693147180559945309417232121458;
constexpr double kLn2Pi = 1.837877066409345483560659472811;   // log(2 pi)
constexpr double kHalfLn2Pi = 0.918938533204672741780329736406;

// Derived per-walker constants (one row of KD doubles per (parameter set, walker)), produced by the
// walker-prep kernel from the resolved parameter table.
enum WalkerSlot : int {
    W_VSYS = 0, W_S2 = 1, W_VX = 2, W_VY = 3,       // v_sys, sigma_max^2, v_maxx, v_maxy
    W_SAC = 4, W_CAC = 5, W_SDC = 6, W_CDC = 7,     // sin/cos(ra_center), sin/cos(dec_center)
    W_VB = 8, W_SB2 = 9, W_FB = 10,                 // v_back, sigma_back^2, f_back
    W_A2 = 11, W_S2A = 12, W_RP2 = 13, W_2RP = 14,  // profile models: a^2, sigma_max^2 a, r_peak^2, 2 r_peak (arcsec)
    KD = 16
};

// Likelihood variants.  Cluster part: CONST (analysis/constant.py) or PROFILE (analysis/model.py:93-180);
// background part: none, fixed per-star lnL + pmember (runner.py:272-286), per-walker Gaussian + density prior
// (constant.py:326-364, model.py:391-456), fixed per-star lnL + density prior with per-walker f_back (model.py:565-623).
enum Model : int {
    MODEL_CONST = 0, MODEL_BGFIXED = 1, MODEL_BGGAUSS = 2,
    MODEL_PROFILE = 3, MODEL_PROFILE_BGGAUSS = 4, MODEL_PROFILE_BGDENS = 5,
    MODEL_PROFILE_BGFIXED = 6          // ModelFit(background=...): profile cluster part + the pmember mixture of runner.py:272-286
};
constexpr int kNumModels = 7;
enum Background : int { BG_NONE = 0, BG_FIXED = 1, BG_GAUSS = 2, BG_FIXED_DENSITY = 3 };

MCD_HD constexpr bool is_profile(int model) { return model >= MODEL_PROFILE; }
MCD_HD constexpr int bg_kind(int model) {
    return (model == MODEL_CONST || model == MODEL_PROFILE) ? BG_NONE
           : (model == MODEL_BGFIXED || model == MODEL_PROFILE_BGFIXED) ? BG_FIXED
           : (model == MODEL_PROFILE_BGDENS) ? BG_FIXED_DENSITY : BG_GAUSS;
}

// Star record slots (doubles).
//   CONST,   fixed centre: v, e2, sin(theta), cos(theta)                     (4)
//   PROFILE, fixed centre: v, e2, dx, dy [arcsec], r^2 [arcsec^2], pad       (6)
//   free centre (both)   : v, e2, sin(ra), cos(ra), sin(dec), cos(dec)       (6)
// extras: BG_FIXED -> lnL_bg, pmember, 1 - pmember, -(lnL_bg + 1/2 log 2pi)   (4)
//         BG_GAUSS -> density, pad                                            (2)
//         BG_FIXED_DENSITY -> lnL_bg, -(lnL_bg + 1/2 log 2pi), density, pad   (4)
MCD_HD constexpr int geometry_doubles(int model, bool free_centre) {
    return (free_centre || is_profile(model)) ? 6 : 4;
}
MCD_HD constexpr int record_doubles(int model, bool free_centre) {
    return geometry_doubles(model, free_centre) +
           (bg_kind(model) == BG_NONE ? 0 : (bg_kind(model) == BG_GAUSS ? 2 : 4));
}

constexpr double kArcsecPerRad = 206264.80624709635516;   // 10800 / pi arcmin x 60: r0 of calc_xy_offset.py:11 in arcsec

template <class T>
MCD_HD T fma_(T a, T b, T c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return std::fma(a, b, c);
#endif
}
MCD_HD float fma_(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaf(a, b, c);
#else
    return std::fmaf(a, b, c);
#endif
}

MCD_HD double rsqrt_nr(double n);
MCD_HD double rcp_nr(double x);

// Free centre: tangent-plane offsets of calc_xy_offset.py:30-31 (in units of r0) from per-star products prepared at
// upload, A = cos(dec) sin(ra), B = cos(dec) cos(ra), sd = sin(dec), and the walker's sin/cos of the centre:
//   x = -cos(dec) sin(ra - ra_c)                         = B sin(ra_c) - A cos(ra_c)
//   y = sin(dec) cos(dec_c) - cos(dec) sin(dec_c) cos(ra - ra_c) = sd cos(dec_c) - sin(dec_c) (B cos(ra_c) + A sin(ra_c))
// Six operations, no per-term trigonometry.
template <class T>
MCD_HD void free_centre_xy(T A, T B, T sd, T sac, T cac, T sdc, T cdc, T& x, T& y) {
    x = fma_(B, sac, -(A * cac));
    const T t = fma_(B, cac, A * sac);
    y = fma_(sd, cdc, -(sdc * t));
}

// v - v_los for the constant-rotation models with a free centre (constant.py:106-111):
//   v_los = v_sys + v_maxx sin(theta) - v_maxy cos(theta),  sin(theta) = y / r, cos(theta) = x / r
//         = v_sys + (v_maxx y - v_maxy x) / r,
// one reciprocal square root and no separate sin/cos.  r == 0 follows numpy's arctan2(+0, -+0) = pi / 0: sin = 0, cos = -+1.
template <bool FASTMATH, class T>
MCD_HD T free_centre_residual(T A, T B, T sd, T sac, T cac, T sdc, T cdc, T vx, T vy, T v_minus_vsys) {
    T x, y;
    free_centre_xy(A, B, sd, sac, cac, sdc, cdc, x, y);
    const T r2 = fma_(x, x, y * y);
    T inv;
    if constexpr (FASTMATH && sizeof(T) == 8) {
        inv = (T)rsqrt_nr((double)r2);            // offsets are O(1e-9 .. 1) rad: r2 is a normal number (or exactly 0)
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        inv = T(1) / sqrt(r2);
#else
        inv = T(1) / std::sqrt(r2);
#endif
    }
    const T cross = fma_(vx, y, -(vy * x));
    const T general = fma_(-cross, inv, v_minus_vsys);
    const T on_centre = fma_(vy, std::signbit(x) ? T(-1) : T(1), v_minus_vsys);
    return r2 > T(0) ? general : on_centre;
}

// a * b + c with the addend c known to be wave-uniform (a star-record value held in an SGPR pair).  hipcc would
// otherwise copy c into VGPRs to use the two-address v_fmac_f64 (2 extra v_mov_b32 per use); the three-address
// VOP3 form takes the SGPR pair directly.
MCD_HD double fma_sgpr_addend(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
#else
    return std::fma(a, b, c);
#endif
}

// -(a * b) + c, same SGPR-addend form (the negation is a source modifier)
MCD_HD double fnma_sgpr_addend(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_fma_f64 %0, -%1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
    return r;
#else
    return std::fma(-a, b, c);
#endif
}

// max(x, lo) for an x that is already an arithmetic result (never a signalling NaN): the plain v_max_f64.  The
// builtin fmax after an inline-asm producer makes hipcc insert a canonicalising v_max_f64 x, x, x first.
MCD_HD double fmax_raw(double x, double lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "s"(lo));
    return r;
#else
    return std::fmax(x, lo);
#endif
}

// log(m) for the mantissa of a rescaled product, m in [1/2, 1): m' = m or 2m in [sqrt(1/2), sqrt(2)), f = (m' - 1) / (m' + 1),
// log m' = 2 f (1 + s/3 + ... + s^10/21), s = f^2 <= 0.0295 (remainder < 1e-18) -- one division and eleven fused
// multiply-adds instead of libm's double-double logarithm (~100 vector instructions, once per wave and product: 7 % of a
// wave's work at 100 stars per chunk).  The absolute error, ~1e-16, sits far below one ulp of what it is added to
// (exponent x ln 2).  Anything else (0 from an underflowed product, inf, NaN) takes libm's log and its special cases.
MCD_HD double log_unit(double m) {
#ifdef MCD_LIBM_LOG_UNIT                     // A/B build (tools/ab_bench.sh): libm's logarithm as before round 3
    const bool always_libm = true;
#else
    const bool always_libm = false;
#endif
    if (always_libm || !(m >= 0.5 && m < 1.0)) {
#if defined(__HIP_DEVICE_COMPILE__)
        return log(m);
#else
        return std::log(m);
#endif
    }
    const bool low = m < 0.70710678118654752440;
    const double mm = low ? m + m : m;
    const double f = (mm - 1.0) / (mm + 1.0), s = f * f;
    double t = 1.0 / 21.0;
    t = fma_(t, s, 1.0 / 19.0);
    t = fma_(t, s, 1.0 / 17.0);
    t = fma_(t, s, 1.0 / 15.0);
    t = fma_(t, s, 1.0 / 13.0);
    t = fma_(t, s, 1.0 / 11.0);
    t = fma_(t, s, 1.0 / 9.0);
    t = fma_(t, s, 1.0 / 7.0);
    t = fma_(t, s, 1.0 / 5.0);
    t = fma_(t, s, 1.0 / 3.0);
    t = fma_(t, s, 1.0);
    return fma_(low ? -1.0 : 0.0, 0.693147180559945309417232121458, (f + f) * t);
}

// ---------------------------------------------------------------------------------------------
// sum of logs as the log of a running product with explicit exponent tracking:
//   sum_i log(x_i) = log(prod_i m_i) + ln2 * sum_i e_i .
// One multiply per factor instead of one log per factor; the product's relative error grows by
// 2^-53 per multiply, i.e. the absolute error of the log sum is <= 1.1e-16 per term -- tighter than
// summing individually rounded logs.
struct LogProduct {
    double p;
    int64_t e;     // exponent carried over from finished groups
    int e32;       // exponent of the current group (folded into e by rescale(); |e32| stays far below 2^31)
    MCD_HD void init() { p = 1.0; e = 0; e32 = 0; }
    MCD_HD void mul(double x) { p *= x; }            // caller keeps |log2 p| < ~1000 between rescales
    MCD_HD void rescale() {
        int ex;
#if defined(__HIP_DEVICE_COMPILE__)
        p = __builtin_frexp(p, &ex);
#else
        p = std::frexp(p, &ex);
#endif
        e += (int64_t)(e32 + ex);
        e32 = 0;
    }
    // narrow-range products: the group exponent stays in the 32-bit counter (|ex| <= 1000 per group of up to 8 factors,
    // chunks hold <= 2^20 stars: mcd_chunks.h kMaxChunkLen), folded into e by value()
    MCD_HD void rescale_narrow() {
        int ex;
#if defined(__HIP_DEVICE_COMPILE__)
        p = __builtin_frexp(p, &ex);
#else
        p = std::frexp(p, &ex);
#endif
        e32 += ex;
    }
    MCD_HD void mul_any(double x) {                  // any positive finite x: split first
        int ex;
#if defined(__HIP_DEVICE_COMPILE__)
        double m = __builtin_frexp(x, &ex);
#else
        double m = std::frexp(x, &ex);
#endif
        p *= m;
        e32 += ex;
    }
    // as mul_any for x >= 0, additionally keeping the smallest BIASED exponent field seen (one v_min_i32): a field
    // below kTrackFloor (x < 2^-1000, which includes denormals and an exact 0) marks the regime where the reference's
    // log-sum-exp works on denormal numbers (see BgFixedAcc::denormal).  The exponent comes from the bit field (one
    // shift) instead of v_frexp_exp; for a denormal or zero x it is off, but those batches are re-evaluated anyway.
    static constexpr int kTrackInit = 2047, kTrackFloor = 23;      // 23 - 1023 = -1000
    MCD_HD void mul_any_track(double x, int& emin) {
        uint64_t bits;
        std::memcpy(&bits, &x, sizeof bits);
        const int bx = (int)(bits >> 52);                            // sign bit is 0
#if defined(__HIP_DEVICE_COMPILE__)
        double m = __builtin_amdgcn_frexp_mant(x);
#else
        int unused;
        double m = std::frexp(x, &unused);
#endif
        p *= m;
        e32 += bx - 1022;
        emin = bx < emin ? bx : emin;
    }
    MCD_HD double value() {
        rescale();
        return fma_((double)e, kLn2, log_unit(p));
    }
};

// ---------------------------------------------------------------------------------------------
// MODEL_CONST, fast path: G stars of one walker reduced to ONE division and ONE product factor.
//   sum_j q_j / n_j = NUM / DEN,  DEN = prod_j n_j,  built as a balanced tree of (num, den) pairs:
//   (a, m) (+) (b, n) = (a n + b m, m n).
// Valid while DEN and NUM stay in range: host guards 2^-60 <= n <= 2^60 and q < 2^120 for G = 8.
struct Frac { double num, den; };
MCD_HD Frac frac_leaf2(double q0, double n0, double q1, double n1) {
    Frac f;
    f.den = n0 * n1;
    f.num = fma_(q1, n0, q0 * n1);
    return f;
}
MCD_HD Frac frac_join(Frac a, Frac b) {
    Frac f;
    f.den = a.den * b.den;
    f.num = fma_(b.num, a.den, a.num * b.den);
    return f;
}

struct ConstAcc {          // accumulators of one walker over one chunk (MODEL_CONST)
    double q;              // sum (v - v_los)^2 / norm
    LogProduct l;          // sum log(norm)
    MCD_HD void init() { q = 0.0; l.init(); }
    MCD_HD void add8(const double* qq, const double* nn) {
        Frac f = frac_join(frac_join(frac_leaf2(qq[0], nn[0], qq[1], nn[1]), frac_leaf2(qq[2], nn[2], qq[3], nn[3])),
                           frac_join(frac_leaf2(qq[4], nn[4], qq[5], nn[5]), frac_leaf2(qq[6], nn[6], qq[7], nn[7])));
        // DEN is a normal number far from the range limits (host guard), so the IEEE division's scaling and
        // fix-up instructions are not needed: reciprocal (v_rcp_f64 + one residual step, < 1 ulp) times NUM.
        q += f.num * rcp_nr(f.den);
        l.mul(f.den);
        l.rescale();
    }
    // 16 stars: one more level of the tree, still ONE reciprocal (DEN = prod of 16 norms needs |log2 norm| <= 50)
    MCD_HD void add16(const double* qq, const double* nn) {
        Frac a = frac_join(frac_join(frac_leaf2(qq[0], nn[0], qq[1], nn[1]), frac_leaf2(qq[2], nn[2], qq[3], nn[3])),
                           frac_join(frac_leaf2(qq[4], nn[4], qq[5], nn[5]), frac_leaf2(qq[6], nn[6], qq[7], nn[7])));
        Frac b = frac_join(frac_join(frac_leaf2(qq[8], nn[8], qq[9], nn[9]), frac_leaf2(qq[10], nn[10], qq[11], nn[11])),
                           frac_join(frac_leaf2(qq[12], nn[12], qq[13], nn[13]), frac_leaf2(qq[14], nn[14], qq[15], nn[15])));
        Frac f = frac_join(a, b);
        q += f.num * rcp_nr(f.den);
        l.mul(f.den);
        l.rescale();
    }
    MCD_HD void add1(double q1, double n1) {
        q += q1 / n1;
        l.mul_any(n1);
    }
    // lnL contribution of `count` stars: -1/2 (count log 2pi + sum log n + sum q/n)
    MCD_HD double finish(int64_t count) { return -0.5 * (fma_((double)count, kLn2Pi, l.value()) + q); }
};

// MODEL_PROFILE, narrow-range variant (guard level 2, mcd_guard.h: level_verdict): the Lynden-Bell residual
//   d = dv - K c / m,   dv = v - v_sys,  c = v_maxx dy - v_maxy dx,  K = 2 r_peak,  m = r_peak^2 + r^2        (model.py:124-127)
// needs the reciprocal of m in the general fast form (v_rcp_f64 + a residual step: ~5 issue slots per term).  Here the
// division is left to the fraction tree:  d^2 / n = (dv m - K c)^2 / (m^2 n), i.e. the tree runs on
// (q', n') = ((dv m - K c)^2, m^2 n) -- two more multiplications instead of the reciprocal -- and since the tree's
// denominator is now prod m_i^2 n_i, the log term needs  sum log n_i = log prod n'_i - 2 log prod m_i: a second running
// product (one multiplication per term, one rescale and one log per 8 terms / per chunk).  Ranges (level_verdict): m <=
// 2^34 arcsec^2, 2^-30 <= n <= 2^30, |dv| <= 2^30: the 8-star denominator stays within 2^+-784, the numerator below 2^820.
struct ProfileNarrowAcc {
    double q;
    LogProduct l;          // prod m_i^2 n_i
    LogProduct lm;         // prod m_i
    MCD_HD void init() { q = 0.0; l.init(); lm.init(); }
    MCD_HD void add8(const double* qq, const double* nn, double m_prod) {
        Frac f = frac_join(frac_join(frac_leaf2(qq[0], nn[0], qq[1], nn[1]), frac_leaf2(qq[2], nn[2], qq[3], nn[3])),
                           frac_join(frac_leaf2(qq[4], nn[4], qq[5], nn[5]), frac_leaf2(qq[6], nn[6], qq[7], nn[7])));
        q += f.num * rcp_nr(f.den);
        l.mul(f.den);
        l.rescale();
        lm.mul(m_prod);
        lm.rescale();
    }
    MCD_HD void add1(double q1, double n1, double m1) {
        q += q1 / n1;
        l.mul_any(n1);
        lm.mul_any(m1);
    }
    MCD_HD double finish(int64_t count) {
        const double sum_log_n = l.value() - 2.0 * lm.value();
        return -0.5 * (fma_((double)count, kLn2Pi, sum_log_n) + q);
    }
};

// float32 counterpart (MCD_F32 / MCD_F32_ACC64): groups of 4 stars so that DEN <= 2^60 and NUM <= 2^77 stay inside the
// f32 range under the host guard 2^-15 <= n <= 2^15, q <= 2^30.  The quotient sum accumulates in A (float or double).
struct FracF { float num, den; };
MCD_HD FracF fracf_leaf2(float q0, float n0, float q1, float n1) {
    FracF f;
    f.den = n0 * n1;
    f.num = fma_(q1, n0, q0 * n1);
    return f;
}
MCD_HD FracF fracf_join(FracF a, FracF b) {
    FracF f;
    f.den = a.den * b.den;
    f.num = fma_(b.num, a.den, a.num * b.den);
    return f;
}
template <class A>
struct ConstAccF {
    A q;
    float p;
    int e;
    MCD_HD void init() { q = 0; p = 1.0f; e = 0; }
    MCD_HD void fold(float x) {
        int ex;
#if defined(__HIP_DEVICE_COMPILE__)
        p = __builtin_frexpf(p * x, &ex);
#else
        p = std::frexp(p * x, &ex);
#endif
        e += ex;
    }
    MCD_HD void add4(const float* qq, const float* nn) {
        const FracF f = fracf_join(fracf_leaf2(qq[0], nn[0], qq[1], nn[1]), fracf_leaf2(qq[2], nn[2], qq[3], nn[3]));
#if defined(__HIP_DEVICE_COMPILE__)
        q += (A)(f.num * __builtin_amdgcn_rcpf(f.den));     // v_rcp_f32: 1 ulp
#else
        q += (A)(f.num / f.den);
#endif
        fold(f.den);
    }
    MCD_HD void add1(float q1, float n1) {
        q += (A)(q1 / n1);
        fold(n1);
    }
    MCD_HD double finish(int64_t count) {
#if defined(__HIP_DEVICE_COMPILE__)
        const double lg = fma_((double)e, kLn2, log((double)p));
#else
        const double lg = fma_((double)e, kLn2, std::log((double)p));
#endif
        return -0.5 * (fma_((double)count, kLn2Pi, lg) + (double)q);
    }
};

// ---------------------------------------------------------------------------------------------
// Plain per-term forms (robust path and mixtures): one log / exp per term as written in the reference.
template <class T> MCD_HD T log_(T x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return log(x);
#else
    return std::log(x);
#endif
}
template <class T> MCD_HD T exp_(T x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return exp(x);
#else
    return std::exp(x);
#endif
}
template <class T> MCD_HD T max_(T a, T b) { return a > b ? a : b; }

// lnL_member of runner.py:280 / lnlike_cluster of constant.py:362
template <class T>
MCD_HD T gauss_lnl(T d, T n) {
    return T(-0.5) * (log_(n) + T(kLn2Pi) + d * d / n);
}

// per-star mixture of runner.py:282-284 / constant.py:320-323
template <class T>
MCD_HD T mixture_lnl(T m, T b, T p) {
    T mx = max_(m, b);
    return mx + log_(p * exp_(m - mx) + (T(1) - p) * exp_(b - mx));
}

// ---------------------------------------------------------------------------------------------
// Fast mixture paths (f64): no log, no divide per term.
//   exp(-1/2 d^2/n) / sqrt(n)  is formed from g = n^(-1/2)  (v_rsq_f64 + one third-order Newton step)
//   and one exp whose argument is <= 0 or exponent-clamped; the per-star mixture value y_i > 0 is
//   folded into a LogProduct:  sum_i log y_i = log prod_i y_i.

MCD_HD double rsqrt_nr(double n) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rsq(n);      // v_rsq_f64: measured max rel. error 2^-24.2 on gfx950 (tools/rsq_probe.hip)
#else
    double y = 1.0 / std::sqrt(n);
#endif
    // one third-order step: with e = 1 - n y^2 (|e| <= 2^-23.2),  n^-1/2 = y (1 + e/2 + 3 e^2/8 + O(e^3)),
    // remaining error 5/16 e^3 < 2^-71: full f64 after the final rounding.  5 instructions.
    const double e = fma_(-(n * y), y, 1.0);
    const double t = fma_(0.375, e, 0.5);
    return fma_(y, t * e, y);
}

MCD_HD double rcp_nr(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double y = __builtin_amdgcn_rcp(x);      // v_rcp_f64: measured max rel. error 2^-24.4 on gfx950
#else
    double y = 1.0 / x;
#endif
    // 1/x = y (1 + e + e^2 + O(e^3)),  e = 1 - x y
    const double e = fma_(-x, y, 1.0);
    return fma_(y, fma_(e, e, e), y);
}

// 2 m^(-1/2) from v_rsq_f64 and ONE Newton step in its three-instruction form,  y (3 - m y^2) = 2 y (1 + e/2),
// e = 1 - m y^2 (|e| <= 2^-23.2):  m^-1/2 = y (1 + e/2 + 3 e^2/8 + ...), so the result is low by 3/8 e^2 <= 4.1e-15
// relative (1.4e-15 on average) plus three roundings.  Used by the narrow-range mixture variants only, where a term's
// log-likelihood responds to that factor with a sensitivity <= |1 - d^2/n|: over N stars the sum moves by <= 4e-15 N,
// i.e. 1e-15 of |lnL| -- inside the rounding error of the reference's own float64 summation.  The factor 2 is free:
// callers scale the variance they pass (m = 8 n gives (2 n)^-1/2).  Two instructions fewer than rsqrt_nr.
MCD_HD double rsqrt2_newton(double m) {
#if defined(MCD_AB_NEWTON3)                       // A/B builds only (csrc/Makefile: variant): the third-order form, for timing
    return 2.0 * rsqrt_nr(m);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    const double y = __builtin_amdgcn_rsq(m);
#else
    const double y = 1.0 / std::sqrt(m);
#endif
    const double s = fma_(-m, y * y, 3.0);
    return y * s;
}

// e^u = 2^e T[j] e^r with k = rint(u N / ln 2) = N e + j, |r| <= ln 2 / 2N, T[j] = 2^(j/N) (mcd_exp_table.h,
// correctly rounded).  N = 1024 (default build): degree-3 polynomial 1 + r + c2 r^2 + c3 r^3 with the even part of its
// error levelled by c2 (tools/gen_exp_table.py), max error 9.4e-17, i.e. below half an ulp -- one FMA per term fewer than
// N = 256 with the degree-4 Taylor polynomial (remainder 3.8e-17; -DMCD_EXP_TAB_BITS=8), for one more integer
// instruction (the 8-bit index is a byte select, the 10-bit one an and + shift).
// Returns the mantissa part T[j] e^r in [1, 2) and e.  `tab` points to the table: LDS on the device (each workgroup
// copies it there; the per-lane lookup is a ds_read_b64, off the VALU), a static array on the host.
// k comes out of the low word of u * (N / ln 2) + 1.5 * 2^52 (round-to-nearest-even), so no v_rndne / v_cvt.
// Requires |u| < 1.4e6 (k inside int32; callers clamp or are bounded by the host guard, mcd_guard.h).
template <bool TWO_STEP = true>
MCD_HD double exp_tab(double u, int& e_out, const double* __restrict__ tab) {
    constexpr double kMagic = 6755399441055744.0;            // 1.5 * 2^52
    const double shifted = fma_(u, kExpTabInvStep, kMagic);
    const double kf = shifted - kMagic;
    uint64_t bits;
    std::memcpy(&bits, &shifted, sizeof bits);
    const int k = (int)(uint32_t)bits;
    double r;
    if constexpr (TWO_STEP) {
        r = fma_(-kf, kExpTabStepHi, u);
        r = fma_(-kf, kExpTabStepLo, r);
    } else {
        // one-constant reduction: ln 2 / N rounded to f64 is off by < 2^-53 of itself, so r is off by < 1.1e-16 |k| ln 2 / N,
        // i.e. a relative error of 8e-17 |u| in e^u -- for callers whose |u| is small wherever e^u matters
        r = fma_(-kf, kExpTabStepHi + kExpTabStepLo, u);
    }
    double p;
    if constexpr (kExpPolyDegree == 4) p = fma_(fma_(r, kExpPolyC4, kExpPolyC3), r, kExpPolyC2);
    else p = fma_(r, kExpPolyC3, kExpPolyC2);
    p = fma_(p, r, 1.0);
    p = fma_(p, r, 1.0);
    e_out = k >> kExpTabBits;
    return tab[k & (kExpTabSize - 1)] * p;
}

// x == +-0 tested on the bit pattern: for a wave-uniform x (SGPR pair) this stays on the scalar unit.
MCD_HD bool is_zero_bits(double x) {
    uint64_t b;
    __builtin_memcpy(&b, &x, sizeof b);
    return (b << 1) == 0;
}

MCD_HD double ldexp_(double x, int k) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_ldexp(x, k);
#else
    return std::ldexp(x, k);
#endif
}
MCD_HD double fmin_(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmin(a, b);              // v_min_f64
#else
    return std::fmin(a, b);
#endif
}
MCD_HD double fabs_(double a) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fabs(a);                 // source modifier, no instruction
#else
    return std::fabs(a);
#endif
}
MCD_HD double fmax_(double a, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmax(a, b);              // v_max_f64
#else
    return std::fmax(a, b);
#endif
}

// MODEL_BGFIXED: lnL_i = b_i + log((1 - p_i) + p_i t_i),  t_i = exp(m_i - b_i) = g exp(-1/2 q g^2 - b'_i),
// b'_i = b_i + 1/2 log 2pi  (the record carries nbp = -b'_i + log p_i).  Same value as runner.py:280-286.
// BG_FIXED_DENSITY (model.py:565-623) is the same with p -> rho_i, (1 - p) -> f_back and an extra -log(rho_i + f_back).
struct BgFixedAcc {
    LogProduct l;          // sum log y_i   (sum b_i is walker-independent: added once per parameter set by the reduce kernel)
    LogProduct lden;       // BG_FIXED_DENSITY: sum log(rho_i + f)
    int emin;              // smallest biased exponent of any mixture value y_i (see denormal())
    MCD_HD void init() { l.init(); lden.init(); emin = LogProduct::kTrackInit; }
    // True when some y_i fell below 2^-1000.  Since 1 - p >= 2^-53 unless p == 1 exactly, that only happens for a star
    // with pmember == 1 (or a walker with f_back == 0) whose cluster term is e^-693 or less: there the reference's
    // log-sum-exp (runner.py:282-284) works on DENORMAL numbers and its result carries their rounding noise
    // (1e-7 .. 1e-2 absolute).  The library then re-evaluates the batch with the plain kernels, which execute the
    // reference's expression literally, so that fast and plain results never differ by more than rounding.
    // An exact y_i == 0 is flagged too: the reference applies its prefactors inside the exponent (e^{m - M}), this path
    // outside (g e^u), so the two underflow at slightly different outliers (found by tools/fuzz_gpu.py).
    MCD_HD bool denormal() const { return emin < LogProduct::kTrackFloor; }
    template <bool NARROW = false>
    MCD_HD void add_density(double d, double n, double rho, double f, double nbp, const double* __restrict__ exptab) {
        add<false, false, NARROW>(d, n, f, nbp, exptab);        // f_back is a per-walker (VGPR) value here
        lden.mul(rho + f);
    }
    // HALVED: the caller passes 2 n instead of n and the table sqrt(2) 2^(j/256) (MCD_EXP_TABLE_SQRT2_VALUES):
    //   gh = (2 n)^(-1/2) = g / sqrt(2),  -(d gh)^2 = -1/2 d^2 g^2,  p gh (sqrt(2) T[j]) e^r = p g T[j] e^r,
    // which drops the multiplication by -1/2 (2 n = 2 verr^2 + 2 sigma^2 is formed by one FMA, like n by one add).
    // NARROW (chosen per call by the host guard, mcd_guard.h: fast_level): every y_i is known to lie in [2^-53, 2^120]
    // -- pmember < 1 everywhere, so y >= 1 - p >= 2^-53; lnL_bg >= -60 and norm >= 2^-60, so y <= 1 + 2^30 e^{60} --
    // hence eight raw factors can be multiplied between two rescales without the per-star mantissa/exponent split,
    // without the k > 1000 exponent carry and without the denormal-regime tracking; g comes from the one-step Newton
    // form (rsqrt2_newton).
    template <bool UNIFORM_OMP = true, bool HALVED = false, bool NARROW = false>
    // The prior weight p of the cluster component is folded into the exponent by the record preparation:
    // nbp = -(b + 1/2 log 2pi) + log p (floored at -2000, where e^u is an exact 0: p == 0 gives y = 1 - p = 1), so
    // y = (1 - p) + g e^{u} with u = -1/2 d^2 g^2 + nbp needs no multiplication by p.
    MCD_HD void add(double d, double n, double omp, double nbp, const double* __restrict__ exptab) {
        // NARROW + HALVED: the caller passes 8 n and the one-step Newton form returns 2 (8 n)^-1/2 = (2 n)^-1/2
        const double g = (NARROW && HALVED) ? rsqrt2_newton(n) : rsqrt_nr(n);
        const double dg = d * g;
        // u <= 1e5 by the host guard (|lnL_bg| <= 1e5); below -1100 e^u is an exact 0 in f64 (as in the reference),
        // and the clamp keeps u N / ln 2 inside the int range of exp_tab.
        // (NARROW: the guard bounds |v - v_los|^2 / norm by 2e6 and the record floors nbp at -2000, so u > -1.1e6 needs no clamp)
        const double u0 = HALVED ? fnma_sgpr_addend(dg, dg, nbp) : fma_sgpr_addend(-0.5 * dg, dg, nbp);
        const double u = NARROW ? u0 : fmax_raw(u0, -1100.0);
        int k;
        const double er = exp_tab<!NARROW>(u, k, exptab);
        if constexpr (NARROW) {
            const double y = UNIFORM_OMP ? fma_sgpr_addend(g, ldexp_(er, k), omp) : fma_(g, ldexp_(er, k), omp);
            l.mul(y);
            return;
        }
        // y = (1 - p) + g e^r 2^k.  k > 1000 (cluster likelihood e^693 times the background's) is carried in the
        // integer part of the product: there the (1 - p) term is below 2^-900 of y and drops out exactly as in f64.
        // k < -1074 underflows inside ldexp; with p == 1 exactly that gives y = 0 and lnL = -inf, which is also what
        // the reference returns there (runner.py:283: log(1 * exp(m - b) + 0) with exp underflowing).
        const int kc = k > 1000 ? 1000 : k;
        const double y = UNIFORM_OMP ? fma_sgpr_addend(g, ldexp_(er, kc), omp) : fma_(g, ldexp_(er, kc), omp);
        l.mul_any_track(y, emin);
        l.e32 += k - kc;
    }
    MCD_HD void rescale() { l.rescale(); }
    MCD_HD void rescale_density() { l.rescale(); lden.rescale(); }
    MCD_HD void rescale_narrow() { l.rescale_narrow(); }
    MCD_HD void rescale_density_narrow() { l.rescale_narrow(); lden.rescale_narrow(); }
    MCD_HD double finish() { return l.value(); }
    MCD_HD double finish_density() { return l.value() - lden.value(); }
};

// MODEL_BGGAUSS (constant.py:320-364):
//   lnL_i = log( mu_i C_i + (1 - mu_i) B_i ),  mu_i = rho_i / (rho_i + f)
//         = -1/2 log 2pi - log(rho_i + f) - 1/2 min(w, wb) + log y_i
//   y_i   = rho g + f gb e^{-delta}   (w <= wb)   or   rho g e^{-delta} + f gb   (w > wb),  delta = |wb - w| / 2
//   with g = n^-1/2, w = d^2 g^2 (cluster) and gb, wb (background).  One exp with a non-positive argument.
struct BgGaussAcc {
    double sum_min;        // sum min(w, wb)   (HALVED: sum of min(w, wb) / 2)
    LogProduct ly;         // sum log y_i      (HALVED: y_i / sqrt(2))
    LogProduct lden;       // sum log(rho_i + f)
    int emin;              // as BgFixedAcc::emin: y_i < 2^-1000 needs the undamped component to be exactly zero
    MCD_HD void init() { sum_min = 0.0; ly.init(); lden.init(); emin = LogProduct::kTrackInit; }
    MCD_HD bool denormal() const { return emin < LogProduct::kTrackFloor; }
    // HALVED: the caller passes 2 n and 2 nb: gh = g / sqrt(2), (d gh)^2 = w / 2, so that the exponent argument
    //   -|wb/2 - w/2| = -delta needs no multiplication by -1/2; y comes out divided by sqrt(2) and min(w, wb) halved,
    //   both undone by constants in finish().
    // NARROW (host guard, mcd_guard.h: fast_level): density and f_back in [2^-20, 2^20], norms in [2^-60, 2^60], so
    //   y >= the undamped term >= 2^-51 and y <= 2^52 -- eight raw factors between rescales, no mantissa/exponent split,
    //   no denormal tracking; |d|^2 <= 2e6 norm, so the exponent argument needs no clamp; one-constant range reduction.
    template <bool HALVED = false, bool NARROW = false>
    MCD_HD void add(double d, double n, double db, double nb, double rho, double f, const double* __restrict__ exptab) {
        // NARROW + HALVED: the caller passes 8 n and 8 nb (one-step Newton form, see BgFixedAcc::add)
        const double g = (NARROW && HALVED) ? rsqrt2_newton(n) : rsqrt_nr(n);
        const double gb = (NARROW && HALVED) ? rsqrt2_newton(nb) : rsqrt_nr(nb);
        const double dg = d * g, dbg = db * gb;
        const double w = dg * dg, wb = dbg * dbg;
        const double t = wb - w;
        const bool cluster_big = t >= 0.0;                 // w <= wb: the cluster exponent is the larger one
        const double u0 = HALVED ? -fabs_(t) : -0.5 * fabs_(t);
        int k;                                            // e^-1100 == 0 in f64; the clamp keeps k inside int range
        const double er = exp_tab<!NARROW>(NARROW ? u0 : fmax_(u0, -1100.0), k, exptab);
        const double e = ldexp_(er, k);                    // k <= 0: underflows to 0 inside ldexp
        const double a = rho * g, b = f * gb;
        // if the undamped component is exactly zero (f_back = 0 or density = 0) and e^{-delta} underflows, y = 0 and
        // lnL = -inf -- the same as the reference's log-sum-exp about the larger exponent (constant.py:320-323).
        const double y = cluster_big ? fma_(b, e, a) : fma_(a, e, b);
        if constexpr (NARROW) ly.mul(y);
        else ly.mul_any_track(y, emin);
        lden.mul(rho + f);
        sum_min += fmin_(w, wb);
    }
    MCD_HD void rescale() { ly.rescale(); lden.rescale(); }
    MCD_HD void rescale_narrow() { ly.rescale_narrow(); lden.rescale_narrow(); }
    template <bool HALVED = false>
    MCD_HD double finish(int64_t count) {
        // HALVED: log y = log(y / sqrt 2) + 1/2 log 2 per star, and sum_min already carries its factor 1/2
        const double per_star = HALVED ? kHalfLn2Pi - 0.5 * kLn2 : kHalfLn2Pi;
        return fma_(-(double)count, per_star, HALVED ? -sum_min : -0.5 * sum_min) + (ly.value() - lden.value());
    }
};

// ---------------------------------------------------------------------------------------------
// float32 fast mixtures (MCD_F32 / MCD_F32_ACC64): the same formulations with v_rsq_f32 / v_exp_f32 (1 ulp each, no
// Newton step, no table) and the running products in A = float or double.  Valid under the f32 conditions of
// mcd_guard.h (fast_guard): every mixture value y lies in [2^-27, 2^31], so four factors fit between two rescales.
MCD_HD float rsqf_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rsqf(x);
#else
    return 1.0f / std::sqrt(x);
#endif
}
MCD_HD float expf_(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);      // v_exp_f32; arguments below -126 / ln 2 flush to 0
#else
    return std::exp(x);
#endif
}
template <class A>
struct LogProductF {                 // sum of logs as log of a product, mantissa in A, exponent in an int
    A p;
    int e;
    MCD_HD void init() { p = 1; e = 0; }
    MCD_HD void mul(float x) { p *= (A)x; }
    MCD_HD void rescale() {
        int ex;
        if constexpr (sizeof(A) == 8) {
#if defined(__HIP_DEVICE_COMPILE__)
            p = __builtin_frexp(p, &ex);
#else
            p = std::frexp(p, &ex);
#endif
        } else {
#if defined(__HIP_DEVICE_COMPILE__)
            p = __builtin_frexpf(p, &ex);
#else
            p = std::frexp(p, &ex);
#endif
        }
        e += ex;
    }
    MCD_HD double value() {
        rescale();
        return fma_((double)e, kLn2, log_unit((double)p));
    }
};
// BG_FIXED / BG_FIXED_DENSITY:  y = w0 + g exp(nbp - 1/2 d^2 g^2),  w0 = 1 - p (record) or f_back (walker)
template <class A>
struct BgFixedAccF {
    LogProductF<A> l, lden;
    MCD_HD void init() { l.init(); lden.init(); }
    MCD_HD void add(float d, float n, float w0, float nbp) {
        const float g = rsqf_(n);
        const float dg = d * g;
        l.mul(fma_(g, expf_(fma_(-0.5f * dg, dg, nbp)), w0));
    }
    MCD_HD void add_density(float d, float n, float rho, float f, float nbp) {
        add(d, n, f, nbp);
        lden.mul(rho + f);
    }
    MCD_HD void rescale() { l.rescale(); }
    MCD_HD void rescale_density() { l.rescale(); lden.rescale(); }
    MCD_HD double finish() { return l.value(); }
    MCD_HD double finish_density() { return l.value() - lden.value(); }
    MCD_HD A value_for_anchor() const { return l.p; }
};
// BG_GAUSS:  y = rho g + f gb e^{-delta} (or mirrored), as BgGaussAcc
template <class A>
struct BgGaussAccF {
    A sum_min;
    LogProductF<A> ly, lden;
    MCD_HD void init() { sum_min = 0; ly.init(); lden.init(); }
    MCD_HD void add(float d, float n, float db, float nb, float rho, float f) {
        const float g = rsqf_(n), gb = rsqf_(nb);
        const float dg = d * g, dbg = db * gb;
        const float w = dg * dg, wb = dbg * dbg;
        const float t = wb - w;
        const float e = expf_(-0.5f * (t < 0.0f ? -t : t));
        const float a = rho * g, b = f * gb;
        ly.mul(t >= 0.0f ? fma_(b, e, a) : fma_(a, e, b));
        lden.mul(rho + f);
        sum_min += (A)(w < wb ? w : wb);
    }
    MCD_HD void rescale() { ly.rescale(); lden.rescale(); }
    MCD_HD double finish(int64_t count) {
        return fma_(-(double)count, kHalfLn2Pi, -0.5 * (double)sum_min) + (ly.value() - lden.value());
    }
    MCD_HD A value_for_anchor() const { return ly.p; }
};

// ---------------------------------------------------------------------------------------------
// background.SingleStars (single_stars.py:42-77): one test star against a slice of the comparison stars.
//   e_j = -(c_j - v)^2 h,  h = 1 / (2 (verr^2 + sigma_int^2));   slice result: nearest distance + sum_j exp(e_j - e_max)
// Two passes over the slice: the nearest comparison star gives the largest exponent exactly, every term of the
// second pass then has a non-positive exponent and the nearest star contributes exactly 1.
struct KdeLane {
    double v, h, d2min, sum;
    MCD_HD void init(double v_, double verr, double sigma_int2) {
        v = v_;
        h = 0.5 / fma_(verr, verr, sigma_int2);
    }
    MCD_HD void nearest(double c, double& dmin) const { dmin = fmin_(dmin, fabs_(c - v)); }
    MCD_HD void begin_sum(double dmin) { d2min = dmin * dmin; sum = 0.0; }
    MCD_HD void add(double c, const double* __restrict__ exptab) {
        const double d = c - v;
        // exp(u) == 0 in f64 below u = -745.2; the clamp keeps k inside int range for far outliers
        const double u = fmax_(fma_(-d, d, d2min) * h, -800.0);
        int k;
        // one-constant range reduction: relative error 8e-17 |u| in a term that is e^u <= 1 of a sum >= 1
        const double er = exp_tab<false>(u, k, exptab);
        sum += ldexp_(er, k);
    }
};

// ---------------------------------------------------------------------------------------------
// Software prefetch of the star records the NEXT loop iteration reads, through the VECTOR memory path: lane l < LINES
// loads one word of the l-th 64-byte line, which pulls the lines into L2; the scalar loads of the next iteration then
// find them there.  (A scalar-load prefetch was tried first: the scalar cache serves a wave's requests in order, so the
// iteration's own loads queued behind the prefetch misses -- 7 - 18 % slower, gpurun_out/ab_prefetch.txt.)  The loaded
// word is kept alive until retire(), so the compiler's own s_waitcnt vmcnt covers it.  Reads up to 1.5 KiB past the
// chunk: the record array is allocated with that slack.  Measured on C3 (gpurun_out/ab_prefetch*.txt): 256 walkers
// 202.7 -> 200.1 us, 128 walkers 117.2 -> 107.4 us, 64 walkers 89.5 -> 61.5 us; with 256 walkers three of a chunk's four
// waves find their records fetched by the first, with fewer walkers every wave waits for memory on its own (VALUBusy
// 85 % / 67 % without the prefetch, tools/sq_w128.sh).
#ifndef MCD_PREFETCH_DISTANCE
#define MCD_PREFETCH_DISTANCE 1          // loop iterations ahead
#endif
template <int BYTES, bool ON>
struct RecordPrefetch {
    static constexpr int kLines = (BYTES + 63) / 64 > 8 ? 8 : (BYTES + 63) / 64;
    uint32_t t;
    // ON is a template parameter of the kernel, not a launch parameter: even switched off at run time the lane index,
    // the predicate and the asm of retire() cost the tight CONST loops 6 - 16 % (C5: 190 us against 164 us compiled
    // out).  Which launches get the prefetching instantiation: mcd_api.hip, wants_prefetch().
    template <class P>
    MCD_HD void issue(P next) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MCD_NO_PREFETCH)
        if constexpr (ON) {
            const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            typedef const uint32_t __attribute__((address_space(1)))* global_word_ptr;
            t = 0;
            if (lane < (unsigned)kLines) t = *(global_word_ptr)((uint64_t)next + lane * 64u);
            return;
        }
#endif
        (void)next;
        t = 0;
    }
    MCD_HD void retire(double anchor) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MCD_NO_PREFETCH)
        if constexpr (ON) asm volatile("" :: "v"(t), "v"(anchor));
#endif
        (void)anchor;
    }
};

// ---------------------------------------------------------------------------------------------
// One chunk of stars for one walker.  On the GPU `r` is wave-uniform (lane = walker), so every record
// read below is a scalar load and the record values are SGPR operands of the vector ops.
template <class T> struct WalkerConsts {
    T vsys, s2, vx, vy, sac, cac, sdc, cdc, vb, sb2, fb, a2, s2a, rp2, rp_2;
    template <class P>
    MCD_HD void load(const P* __restrict__ p) {
        vsys = p[W_VSYS]; s2 = p[W_S2]; vx = p[W_VX]; vy = p[W_VY];
        sac = p[W_SAC]; cac = p[W_CAC]; sdc = p[W_SDC]; cdc = p[W_CDC];
        vb = p[W_VB]; sb2 = p[W_SB2]; fb = p[W_FB];
        a2 = p[W_A2]; s2a = p[W_S2A]; rp2 = p[W_RP2]; rp_2 = p[W_2RP];
    }
};

template <class T> MCD_HD T sqrt_(T x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return sqrt(x);
#else
    return std::sqrt(x);
#endif
}

// Residual d = v - v_los and variance n = verr^2 + sigma_los^2 of one star for one walker.
//   CONST   (constant.py:52-111): v_los = v_sys + v_maxx sin(theta) - v_maxy cos(theta), sigma_los = sigma_max
//   PROFILE (model.py:93-180):    v_los = v_sys + 2 r_peak (v_maxx dy - v_maxy dx) / (r_peak^2 + r^2)
//                                 sigma_los^2 = sigma_max^2 a / sqrt(a^2 + r^2)           (all lengths in arcsec)
// NEWTON1 (narrow-range variants of the profile mixtures, f64): the Plummer root from ONE Newton step in its
// three-instruction form (rsqrt2_newton: low by <= 4.1e-15 relative, see there), two instructions fewer than rsqrt_nr.
template <int MODEL, class T, bool FREE, bool FASTMATH = false, bool NEWTON1 = false>
MCD_HD void star_d_n(RecPtr<T> r, const WalkerConsts<T>& w, T& d, T& n) {
    if constexpr (!is_profile(MODEL)) {
        if (FREE) d = free_centre_residual<FASTMATH>(r[2], r[3], r[4], w.sac, w.cac, w.sdc, w.cdc, w.vx, w.vy, r[0] - w.vsys);
        else d = fma_(-w.vx, r[2], fma_(w.vy, r[3], r[0] - w.vsys));
        n = r[1] + w.s2;
    } else {
        T dx, dy, r2;
        if (FREE) {
            T x, y;
            free_centre_xy(r[2], r[3], r[4], w.sac, w.cac, w.sdc, w.cdc, x, y);
            dx = T(kArcsecPerRad) * x;
            dy = T(kArcsecPerRad) * y;
            r2 = fma_(dx, dx, dy * dy);
        } else { dx = r[2]; dy = r[3]; r2 = r[4]; }
        T t, inv;
        if constexpr (FASTMATH && NEWTON1 && sizeof(T) == 8) {
            // rsqrt2_newton returns 2 (a^2 + r^2)^-1/2: the factor goes into sigma_max^2 a / 2 (loop-invariant)
            const T cross = fma_(w.vx, dy, -(w.vy * dx));
            inv = (T)rcp_nr((double)(w.rp2 + r2));
            d = fma_(-(w.rp_2 * inv), cross, r[0] - w.vsys);
            n = fma_(T(0.5) * w.s2a, (T)rsqrt2_newton((double)(w.a2 + r2)), r[1]);
            return;
        } else if constexpr (FASTMATH && sizeof(T) == 8) {
            t = (T)rsqrt_nr((double)(w.a2 + r2));
            inv = (T)rcp_nr((double)(w.rp2 + r2));
        } else {
            t = T(1) / sqrt_(w.a2 + r2);
            inv = T(1) / (w.rp2 + r2);
        }
        const T cross = fma_(w.vx, dy, -(w.vy * dx));
        d = fma_(-(w.rp_2 * inv), cross, r[0] - w.vsys);
        n = fma_(w.s2a, t, r[1]);
    }
}

// `denormal` is set when a fast mixture path met the denormal regime described at BgFixedAcc::denormal().
// `exptab`: the 2^(j/256) table of exp_tab (read by the fast mixture paths only; may be null otherwise);
// for MODEL_BGFIXED it is the sqrt(2)-scaled table (exp_table_is_sqrt2_scaled).
MCD_HD constexpr bool exp_table_is_sqrt2_scaled(int model) { return model == MODEL_BGFIXED; }

// FAST: 0 = plain (the reference's expressions term by term), 1 = fast formulation, 2 = fast formulation with the
// narrow-range products of BgFixedAcc::add (MODEL_BGFIXED, MODEL_PROFILE_BGDENS) / BgGaussAcc::add (MODEL_BGGAUSS,
// MODEL_PROFILE_BGGAUSS); for the models without background the same as 1.
template <int MODEL, bool FREE, class T, class A, int FAST, bool PF = false>
MCD_HD double chunk_loglike(RecPtr<T> r, int count, const WalkerConsts<T>& w, bool& denormal,
                            const double* __restrict__ exptab) {
    constexpr int ND = record_doubles(MODEL, FREE);
    denormal = false;
    constexpr int XB = geometry_doubles(MODEL, FREE);      // first background slot of a record
    constexpr int BG = bg_kind(MODEL);
    double result;

    if constexpr (BG != BG_NONE && FAST && sizeof(T) == 4) {
        // float32 fast mixtures: four stars (one scalar record batch) per rescale; A = float or double
        auto run4 = [&](auto& acc, auto&& one, auto&& rescale) {
            const int n4 = count >> 2;
            for (int g = 0; g < n4; ++g, r += 4 * ND) {
                RecordPrefetch<4 * ND * 4, PF> pf;
                pf.issue(r + MCD_PREFETCH_DISTANCE * 4 * ND);
#pragma unroll
                for (int j = 0; j < 4; ++j) one(r + j * ND);
                rescale();
                pf.retire((double)acc.value_for_anchor());
            }
            for (int j = n4 * 4; j < count; ++j, r += ND) { one(r); rescale(); }
        };
        if constexpr (BG == BG_GAUSS) {
            BgGaussAccF<A> acc;
            acc.init();
            run4(acc, [&](RecPtr<T> rr) {
                T d, n;
                star_d_n<MODEL, T, FREE, true>(rr, w, d, n);
                acc.add(d, n, rr[0] - w.vb, rr[1] + w.sb2, rr[XB], w.fb);
            }, [&]() { acc.rescale(); });
            result = acc.finish(count);
        } else if constexpr (BG == BG_FIXED) {
            BgFixedAccF<A> acc;
            acc.init();
            run4(acc, [&](RecPtr<T> rr) {
                T d, n;
                star_d_n<MODEL, T, FREE, true>(rr, w, d, n);
                acc.add(d, n, rr[XB + 2], rr[XB + 3]);
            }, [&]() { acc.rescale(); });
            result = acc.finish();
        } else {
            BgFixedAccF<A> acc;
            acc.init();
            run4(acc, [&](RecPtr<T> rr) {
                T d, n;
                star_d_n<MODEL, T, FREE, true>(rr, w, d, n);
                acc.add_density(d, n, rr[XB + 2], w.fb, rr[XB + 1]);
            }, [&]() { acc.rescale_density(); });
            result = acc.finish_density();
        }
    } else if constexpr (BG == BG_NONE && FAST && sizeof(T) == 4) {
        // f32 fraction tree over 4 stars + f32 log-product.  One iteration covers 16 stars (four trees) so that four
        // 64-byte scalar record loads are in flight per wave: a 4-star iteration is only ~40 ns of VALU work, far
        // less than one load latency even with 8 waves per SIMD.
        ConstAccF<A> acc;
        acc.init();
        // MODEL_CONST with a fixed centre has registers to spare for a 16-star tree (one reciprocal per 16 stars); the
        // other instantiations keep 8-star trees (a 16-star tree there costs occupancy)
        constexpr bool TREE16 = MODEL == MODEL_CONST && !FREE;
        const int n16 = TREE16 ? count >> 4 : 0;
        for (int g = 0; g < n16; ++g, r += 16 * ND) {
            RecordPrefetch<16 * ND * 4, PF> pf;
            pf.issue(r + MCD_PREFETCH_DISTANCE * 16 * ND);
            float qq[16], nn[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float d;
                star_d_n<MODEL, float, FREE, true>(r + j * ND, w, d, nn[j]);
                qq[j] = d * d;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc.add4(qq + 4 * t, nn + 4 * t);
            pf.retire((double)acc.p);
        }
        const int done = n16 << 4;
        const int n4 = (count - done) >> 2;
        for (int g = 0; g < n4; ++g, r += 4 * ND) {
            float qq[4], nn[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float d;
                star_d_n<MODEL, float, FREE, true>(r + j * ND, w, d, nn[j]);
                qq[j] = d * d;
            }
            acc.add4(qq, nn);
        }
        for (int j = done + n4 * 4; j < count; ++j, r += ND) {
            float d, n;
            star_d_n<MODEL, float, FREE, true>(r, w, d, n);
            acc.add1(d * d, n);
        }
        result = acc.finish(count);
    } else if constexpr (BG == BG_NONE && FAST == 2 && MODEL == MODEL_PROFILE && !FREE) {
        // narrow-range profile variant (ProfileNarrowAcc): no reciprocal per term, one-step Newton root for the Plummer
        // dispersion (rsqrt2_newton returns 2 (a^2 + r^2)^-1/2: the factor goes into sigma_max^2 a / 2)
        ProfileNarrowAcc acc;
        acc.init();
        const double hs2a = 0.5 * w.s2a;
        auto one = [&](RecPtr<double> rr, double& q1, double& n1, double& m1) {
            m1 = w.rp2 + rr[4];
            const double t2 = rsqrt2_newton(w.a2 + rr[4]);
            const double n = fma_(hs2a, t2, rr[1]);
            const double cross = fma_(w.vx, rr[3], -(w.vy * rr[2]));
            const double nd = fma_(rr[0] - w.vsys, m1, -(w.rp_2 * cross));
            q1 = nd * nd;
            n1 = (m1 * m1) * n;
        };
        const int n8 = count >> 3;
        for (int g = 0; g < n8; ++g, r += 8 * ND) {
            RecordPrefetch<8 * ND * 8, PF> pf;
            pf.issue(r + MCD_PREFETCH_DISTANCE * 8 * ND);
            double qq[8], nn[8], mm[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) one(r + j * ND, qq[j], nn[j], mm[j]);
            const double m_prod = ((mm[0] * mm[1]) * (mm[2] * mm[3])) * ((mm[4] * mm[5]) * (mm[6] * mm[7]));
            acc.add8(qq, nn, m_prod);
            pf.retire(acc.q);
        }
        for (int j = count & ~7; j < count; ++j, r += ND) {
            double q1, n1, m1;
            one(r, q1, n1, m1);
            acc.add1(q1, n1, m1);
        }
        result = acc.finish(count);
    } else if constexpr (BG == BG_NONE && FAST) {
        // fraction-tree + log-product path (f64): 8 stars -> one division, one product factor
        ConstAcc acc;
        acc.init();
        // MODEL_CONST with a fixed centre has registers to spare for a 16-star tree (one reciprocal per 16 stars); the
        // other instantiations keep 8-star trees (a 16-star tree there costs occupancy)
        constexpr bool TREE16 = MODEL == MODEL_CONST && !FREE;
        const int n16 = TREE16 ? count >> 4 : 0;
        for (int g = 0; g < n16; ++g, r += 16 * ND) {
            RecordPrefetch<16 * ND * 8, PF> pf;
            pf.issue(r + MCD_PREFETCH_DISTANCE * 16 * ND);
            double qq[16], nn[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                double d;
                star_d_n<MODEL, double, FREE, true>(r + j * ND, w, d, nn[j]);
                qq[j] = d * d;
            }
            acc.add16(qq, nn);
            pf.retire(acc.q);
        }
        const int n8 = TREE16 ? (count >> 3) & 1 : count >> 3;
        for (int g = 0; g < n8; ++g, r += 8 * ND) {
            RecordPrefetch<8 * ND * 8, PF> pf;
            pf.issue(r + MCD_PREFETCH_DISTANCE * 8 * ND);
            double qq[8], nn[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                double d;
                star_d_n<MODEL, double, FREE, true>(r + j * ND, w, d, nn[j]);
                qq[j] = d * d;
            }
            acc.add8(qq, nn);
            pf.retire(acc.q);
        }
        for (int j = count & ~7; j < count; ++j, r += ND) {
            double d, n;
            star_d_n<MODEL, double, FREE, true>(r, w, d, n);
            acc.add1(d * d, n);
        }
        result = acc.finish(count);
    } else if constexpr (BG == BG_NONE) {
        // plain path: one log and one division per term (runner.py:269-270 keeps two sums as well)
        A sum_log = 0, sum_q = 0;
#pragma unroll 4
        for (int j = 0; j < count; ++j, r += ND) {
            T d, n;
            star_d_n<MODEL, T, FREE>(r, w, d, n);
            sum_log += (A)log_(n);
            sum_q += (A)(d * d / n);
        }
        result = -0.5 * ((double)count * kLn2Pi + (double)sum_log + (double)sum_q);
    } else if constexpr (BG == BG_FIXED && FAST) {
        // MODEL_BGFIXED has norm = verr^2 + sigma^2 (constant.py:52-74): the accumulator takes 2 norm (HALVED form; 8 norm
        // for the narrow-range variant's one-step Newton reciprocal root) and `exptab` is then the sqrt(2)-scaled table;
        // star_d_n's own norm is dead code there.
        constexpr bool HALVED = MODEL == MODEL_BGFIXED;
        constexpr bool NARROW = FAST == 2 && MODEL == MODEL_BGFIXED;
        constexpr double kScale = NARROW ? 8.0 : 2.0;
        const double s2x = kScale * (double)w.s2;
        BgFixedAcc acc;
        acc.init();
        auto four = [&](RecPtr<double> r4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                RecPtr<double> rr = r4 + j * ND;
                double d, n;
                star_d_n<MODEL, double, FREE, true>(rr, w, d, n);
                if constexpr (HALVED) n = fma_(kScale, rr[1], s2x);
                acc.add<true, HALVED, NARROW>(d, n, rr[XB + 2], rr[XB + 3], exptab);
            }
        };
        const int n4 = count >> 2;
        if constexpr (NARROW) {
            // eight raw factors per rescale: every second 4-star group (one scalar record-load batch each)
            for (int g = 0; g < n4; ++g, r += 4 * ND) {
                RecordPrefetch<4 * ND * 8, PF> pf;
                pf.issue(r + MCD_PREFETCH_DISTANCE * 4 * ND);
                four(r);
                pf.retire(acc.l.p);
                if (g & 1) { MCD_KEEP_BRANCH(); acc.rescale_narrow(); }    // wave-uniform: a scalar branch, not a select
            }
            if (n4 & 1) acc.rescale_narrow();
        } else {
            for (int g = 0; g < n4; ++g, r += 4 * ND) {
                RecordPrefetch<4 * ND * 8, PF> pf;
                pf.issue(r + MCD_PREFETCH_DISTANCE * 4 * ND);
                four(r);
                pf.retire(acc.l.p);
                acc.rescale();
            }
        }
        for (int j = n4 * 4; j < count; ++j, r += ND) {
            double d, n;
            star_d_n<MODEL, double, FREE, true>(r, w, d, n);
            if constexpr (HALVED) n = fma_(kScale, r[1], s2x);
            acc.add<true, HALVED, NARROW>(d, n, r[XB + 2], r[XB + 3], exptab);
            acc.rescale();
        }
        result = acc.finish();
        denormal = acc.denormal();
    } else if constexpr (BG == BG_FIXED_DENSITY && FAST) {
        constexpr bool NARROW = FAST == 2;          // f_back >= 2^-20 bounds every mixture value from below (mcd_guard.h)
        BgFixedAcc acc;
        acc.init();
        auto four = [&](RecPtr<double> r4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                RecPtr<double> rr = r4 + j * ND;
                double d, n;
                star_d_n<MODEL, double, FREE, true, NARROW>(rr, w, d, n);
                acc.add_density<NARROW>(d, n, rr[XB + 2], w.fb, rr[XB + 1], exptab);
            }
        };
        const int n4 = count >> 2;
        if constexpr (NARROW) {
            for (int g = 0; g < n4; ++g, r += 4 * ND) {
                RecordPrefetch<4 * ND * 8, PF> pf;
                pf.issue(r + MCD_PREFETCH_DISTANCE * 4 * ND);
                four(r);
                pf.retire(acc.l.p);
                if (g & 1) { MCD_KEEP_BRANCH(); acc.rescale_density_narrow(); }    // wave-uniform: a scalar branch, not a select
            }
            if (n4 & 1) acc.rescale_density_narrow();
        } else {
            for (int g = 0; g < n4; ++g, r += 4 * ND) {
                RecordPrefetch<4 * ND * 8, PF> pf;
                pf.issue(r + MCD_PREFETCH_DISTANCE * 4 * ND);
                four(r);
                pf.retire(acc.l.p);
                acc.rescale_density();
            }
        }
        for (int j = n4 * 4; j < count; ++j, r += ND) {
            double d, n;
            star_d_n<MODEL, double, FREE, true, NARROW>(r, w, d, n);
            acc.add_density<NARROW>(d, n, r[XB + 2], w.fb, r[XB + 1], exptab);
            acc.rescale_density();
        }
        result = acc.finish_density();
        denormal = acc.denormal();
    } else if constexpr (BG == BG_GAUSS && FAST) {
        // MODEL_BGGAUSS has norm = verr^2 + sigma^2 and verr^2 + sigma_back^2: doubled norms are one FMA each (HALVED form;
        // 8 norm for the narrow-range variant's one-step Newton reciprocal roots)
        constexpr bool HALVED = MODEL == MODEL_BGGAUSS;
        constexpr bool NARROW = FAST == 2;
        constexpr double kScale = NARROW ? 8.0 : 2.0;
        const double s2x = kScale * (double)w.s2, sb2x = kScale * (double)w.sb2;
        BgGaussAcc acc;
        acc.init();
        auto one = [&](RecPtr<double> rr) {
            double d, n;
            star_d_n<MODEL, double, FREE, true, NARROW>(rr, w, d, n);
            if constexpr (HALVED) acc.add<true, NARROW>(d, fma_(kScale, rr[1], s2x), rr[0] - w.vb, fma_(kScale, rr[1], sb2x), rr[XB], w.fb, exptab);
            else acc.add<false, NARROW>(d, n, rr[0] - w.vb, rr[1] + w.sb2, rr[XB], w.fb, exptab);
        };
        auto four = [&](RecPtr<double> r4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) one(r4 + j * ND);
        };
        const int n4 = count >> 2;
        if constexpr (NARROW) {
            for (int g = 0; g < n4; ++g, r += 4 * ND) {
                RecordPrefetch<4 * ND * 8, PF> pf;
                pf.issue(r + MCD_PREFETCH_DISTANCE * 4 * ND);
                four(r);
                pf.retire(acc.ly.p);
                if (g & 1) { MCD_KEEP_BRANCH(); acc.rescale_narrow(); }    // wave-uniform: a scalar branch, not a select
            }
            if (n4 & 1) acc.rescale_narrow();
        } else {
            for (int g = 0; g < n4; ++g, r += 4 * ND) {
                RecordPrefetch<4 * ND * 8, PF> pf;
                pf.issue(r + MCD_PREFETCH_DISTANCE * 4 * ND);
                four(r);
                pf.retire(acc.ly.p);
                acc.rescale();
            }
        }
        for (int j = n4 * 4; j < count; ++j, r += ND) {
            one(r);
            acc.rescale();
        }
        result = acc.finish<HALVED>(count);
        denormal = acc.denormal();
    } else {
        A sum = 0;
#pragma unroll 2
        for (int j = 0; j < count; ++j, r += ND) {
            T d, n;
            star_d_n<MODEL, T, FREE>(r, w, d, n);
            const T m = gauss_lnl(d, n);
            T b, p;
            if (BG == BG_FIXED) {
                b = r[XB];
                p = r[XB + 1];
            } else if (BG == BG_FIXED_DENSITY) {
                b = r[XB];
                const T rho = r[XB + 2];
                p = rho / (rho + w.fb);                          // model.py:588
            } else {
                const T nb = r[1] + w.sb2;                       // constant.py:333, model.py:423
                const T db = r[0] - w.vb;
                b = gauss_lnl(db, nb);                           // constant.py:334-336
                const T rho = r[XB];
                p = rho / (rho + w.fb);                          // constant.py:339, model.py:429
            }
            sum += (A)mixture_lnl(m, b, p);                      // runner.py:282-284, constant.py:320-323
        }
        result = (double)sum;
    }
    return result;
}

// Per-star log-likelihood pieces for the membership / no_sum outputs: cluster lnL, background lnL, prior m.
template <int MODEL, bool FREE, class T>
MCD_HD void star_components(RecPtr<T> r, const WalkerConsts<T>& w, T& lc, T& lb, T& m) {
    constexpr int XB = geometry_doubles(MODEL, FREE);
    constexpr int BG = bg_kind(MODEL);
    T d, n;
    star_d_n<MODEL, T, FREE>(r, w, d, n);
    lc = gauss_lnl(d, n);
    if (BG == BG_FIXED) { lb = r[XB]; m = r[XB + 1]; }
    else if (BG == BG_FIXED_DENSITY) { lb = r[XB]; const T rho = r[XB + 2]; m = rho / (rho + w.fb); }
    else if (BG == BG_GAUSS) { lb = gauss_lnl(r[0] - w.vb, r[1] + w.sb2); const T rho = r[XB]; m = rho / (rho + w.fb); }
    else { lb = T(-INFINITY); m = T(1); }
}

}  // namespace mcd
