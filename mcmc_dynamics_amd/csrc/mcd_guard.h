// mcd_guard.h -- range guard that decides, per call, whether the fast kernel formulations may be used.
// Shared by the C-ABI (mcd_api.hip), by the resident stretch-move chain (mcd_stretch.hip evaluates the same verdict on the
// device for the tables it builds there) and by the CPU test harness (tests/emul), so that the randomized tests exercise
// exactly the condition the library applies.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "mcd_math.h"

namespace mcd {

constexpr double kInfinity = std::numeric_limits<double>::infinity();

// Range statistics of one catalogue, gathered at upload.  The scalar part travels to the device by value (the resident
// stretch-move chain re-evaluates the guard there, mcd_stretch.hip).
struct StatsScalars {
    double e2_min = 0, e2_max = 0, v_abs_max = 0;   // verr^2 range, max |v|
    double rho_min = 0, rho_max = 0;                // density range (BG_GAUSS / BG_FIXED_DENSITY)
    double lnbg_min = 0, lnbg_max = 0, pm_max = 0;  // ranges of the fixed background columns (float32 fast mixtures)
    bool stats_finite = true;                       // v, verr all finite
    bool extras_ok = true;                          // background columns inside the fast-path ranges
    bool narrow_possible = false;                   // at most 1/8 of the stars are exceptions (else: general form throughout)
    double r_max_fixed = 0.0;                       // fixed centre: largest separation of a star from it [arcsec] (0: unknown /
                                                    // free centre) -- bounds r_peak^2 + r^2 for the narrow-range profile variant
};

struct CatalogStats : StatsScalars {
    // Stars that rule out the narrow-range variants for the CHUNK that holds them (narrow_exception below), ascending
    // indices; the kernel then takes the general fast form for those chunks only (LaunchShape::chunk_general).
    std::vector<int64_t> narrow_exceptions;
    // float32 accuracy domain (f32_domain below), gathered when the positions are given: reference point (the fixed centre,
    // else the catalogue's centroid) [deg], harmonic mean [rad] and maximum [arcsec] of the stars' angular separations from it
    double ref_ra = 0.0, ref_dec = 0.0, sep_harm = 0.0, r_max_arcsec = 0.0;
};

// BGFIXED: a certain member (pmember == 1: the mixture value y = (1 - p) + ... has no floor) or lnL_bg < -60 (y can
// exceed 2^120: eight raw factors no longer fit between two rescales).  BGGAUSS family: density outside [2^-20, 2^20]
// (the undamped term rho g can vanish or explode).  PROFILE_BGDENS: lnL_bg < -45 or density > 2^20.
inline bool narrow_exception(int bg, double lnbg, double pm, double rho) {
    if (bg == BG_FIXED) return !(pm < 1.0) || lnbg < -60.0;
    if (bg == BG_GAUSS) return !(rho >= 0x1p-20 && rho <= 0x1p20);
    if (bg == BG_FIXED_DENSITY) return lnbg < -45.0 || !(rho <= 0x1p20);
    return false;
}

// Gathered once at upload from the host columns (runner.py:261: norm = verr*verr + sigma*sigma).
inline CatalogStats compute_stats(int64_t n, const double* v, const double* verr, const double* lnbg,
                                  const double* pmember, const double* density, int bg, const double* ra = nullptr,
                                  const double* dec = nullptr, bool fixed_centre = false, double ra_c = 0.0,
                                  double dec_c = 0.0) {
    CatalogStats st;
    if (ra && dec && n > 0) {
        double mr = ra_c, md = dec_c;
        if (!fixed_centre) {
            mr = md = 0.0;
            for (int64_t i = 0; i < n; ++i) { mr += ra[i]; md += dec[i]; }
            mr /= (double)n; md /= (double)n;
        }
        const double kDeg = 0.017453292519943295, cd = std::cos(md * kDeg);
        double inv = 0.0, far = 0.0;
        for (int64_t i = 0; i < n; ++i) {
            const double sep = std::hypot((ra[i] - mr) * cd, dec[i] - md) * kDeg;
            inv += 1.0 / std::max(sep, 1e-9);                      // (a star on the reference point: 2e-4 arcsec)
            far = std::max(far, sep);
        }
        st.ref_ra = mr; st.ref_dec = md;
        st.sep_harm = inv > 0.0 && std::isfinite(inv) ? (double)n / inv : 0.0;
        st.r_max_arcsec = std::isfinite(far) ? far / kDeg * 3600.0 : kInfinity;
        // (small-angle separation against the exact tangent-plane radius of the records: 1 + 1e-5 within a degree; the
        // narrow-range bound below has a factor 4 to spare)
        if (fixed_centre) st.r_max_fixed = st.r_max_arcsec;
    }
    double e2_min = std::numeric_limits<double>::infinity(), e2_max = 0.0, v_abs = 0.0;
    double r_min = std::numeric_limits<double>::infinity(), r_max = 0.0;
    bool finite = true, ok = true;
    if ((bg == BG_FIXED || bg == BG_FIXED_DENSITY) && n > 0) {
        st.lnbg_min = std::numeric_limits<double>::infinity();
        st.lnbg_max = -st.lnbg_min;
        for (int64_t i = 0; i < n; ++i) {
            st.lnbg_min = std::min(st.lnbg_min, lnbg[i]);
            st.lnbg_max = std::max(st.lnbg_max, lnbg[i]);
            if (bg == BG_FIXED) st.pm_max = std::max(st.pm_max, pmember[i]);
        }
    }
    for (int64_t i = 0; i < n; ++i) {
        const double e2 = verr[i] * verr[i];
        const double av = std::fabs(v[i]);
        if (!(std::isfinite(e2) && std::isfinite(av))) { finite = false; continue; }
        e2_min = std::min(e2_min, e2);
        e2_max = std::max(e2_max, e2);
        v_abs = std::max(v_abs, av);
    }
    for (int64_t i = 0; i < n && ok; ++i) {
        if (bg == BG_FIXED || bg == BG_FIXED_DENSITY) {
            const double b = lnbg[i];
            if (!(std::isfinite(b) && b > -1.0e5 && b < 1.0e5)) ok = false;
        }
        if (bg == BG_FIXED) {
            const double pm = pmember[i];
            if (!(pm >= 0.0 && pm <= 1.0)) ok = false;
            // lnL_bg < -690: the cluster term can exceed 2^1000 times the background's and is then carried in the
            // exponent (BgFixedAcc::add), which drops (1 - p): only valid while p 2^1000 dwarfs 1.  A zero or
            // vanishing prior there also makes the reference's log-sum-exp underflow (runner.py:282-284 gives -inf for
            // p == 0): the plain kernels reproduce that literally.
            if (lnbg[i] < -690.0 && !(pm >= 0x1p-700)) ok = false;
        }
        if (bg == BG_GAUSS || bg == BG_FIXED_DENSITY) {
            const double rho = density[i];
            if (!(std::isfinite(rho) && rho >= 0.0)) ok = false;
            if (bg == BG_FIXED_DENSITY && lnbg[i] < -690.0 && !(rho >= 0x1p-700)) ok = false;   // as for pmember above
            r_min = std::min(r_min, rho);
            r_max = std::max(r_max, rho);
        }
        if (bg != BG_NONE && narrow_exception(bg, lnbg ? lnbg[i] : 0.0, pmember ? pmember[i] : 0.0, density ? density[i] : 1.0))
            st.narrow_exceptions.push_back(i);
    }
    if (n == 0) { e2_min = 0.0; r_min = 0.0; }
    st.e2_min = e2_min; st.e2_max = e2_max; st.v_abs_max = v_abs; st.stats_finite = finite;
    st.extras_ok = ok; st.rho_min = r_min; st.rho_max = r_max;
    st.narrow_possible = ok && bg != BG_NONE && (int64_t)st.narrow_exceptions.size() * 8 <= n;
    if (!st.narrow_possible) std::vector<int64_t>().swap(st.narrow_exceptions);
    return st;
}

// Fast paths (f64 only) are valid while their intermediate products stay far from over/underflow.
//   CONST   (fraction tree over 8 / 16 stars + log product): 2^-55 <= verr^2 + sigma^2 <= 2^55, |v - v_los| < 2^50
//   BGFIXED / BGGAUSS (rsqrt + one exp + log product):  2^-200 <= norm <= 2^200, |v - v_los|^2 / norm <= 1.6e9, finite columns,
//            lnlike_bg > -1e5, 0 <= pmember <= 1 (>= 2^-700 where lnlike_bg < -690);  density >= 0 (likewise), f_back >= 0, 2^-100 <= density + f_back <= 2^100
// Anything else (including NaN/inf parameters) takes the plain kernels, which evaluate the reference's
// expressions term by term.
// per-call ranges fast_guard derives from the catalogue statistics and the parameter table (for fast_level)
struct GuardRanges {
    double n_min = 0, n_max = 0;        // cluster variance verr^2 + sigma_los^2
    double nb_min = 0, nb_max = 0;      // background variance verr^2 + sigma_back^2 (BG_GAUSS)
    double d_max = 0;                   // bound on |v - v_los| (and |v - v_back|)
    double f_min = 0, f_max = 0;        // f_back (BG_GAUSS, BG_FIXED_DENSITY)
};

// Ranges of one parameter table, gathered row by row.  Only min / max of per-row quantities: the result does not depend
// on the order in which rows (or partial results, merge()) are visited, so a parallel reduction on the device and the
// serial loop on the host give the same bits.
struct ParamRanges {
    double s2_min = kInfinity, s2_max = 0.0, amp = 0.0, sb2_min = kInfinity, sb2_max = 0.0, f_min = kInfinity, f_max = 0.0;
    double len_min = kInfinity, len_max = 0.0;                 // a and r_peak of the profile models
    bool finite = true;                                        // every row finite where the guard needs it

    MCD_HD static double lesser(double a, double b) { return b < a ? b : a; }
    MCD_HD static double greater(double a, double b) { return a < b ? b : a; }

    MCD_HD void add_row(const double* p, int k, int model, bool free_centre) {
        const bool prof = is_profile(model);
        const int bg = bg_kind(model);
        const int ix = prof ? 3 : 2, iy = prof ? 4 : 3;
        const double s2 = p[1] * p[1];
        double a = fabs(p[0]) + fabs(p[ix]) + fabs(p[iy]);
        bool ok = true;
        if (prof) ok = ok && finite_value(p[2]) && finite_value(p[5]);
        double sb2 = 0.0, f = 0.0;
        if (bg == BG_GAUSS) {
            sb2 = p[k - 2] * p[k - 2];
            a = greater(a, fabs(p[k - 3]));
            ok = ok && finite_value(sb2);
        }
        if (bg == BG_GAUSS || bg == BG_FIXED_DENSITY) {
            f = p[k - 1];
            ok = ok && finite_value(f);
        }
        if (free_centre) {
            const int ic = prof ? 6 : 4;
            ok = ok && finite_value(p[ic]) && finite_value(p[ic + 1]);
        }
        ok = ok && finite_value(s2) && finite_value(a);
        if (!ok) { finite = false; return; }                   // the verdict is "plain kernels" whatever the other rows hold
        if (prof) {
            len_min = lesser(len_min, lesser(p[2], p[5]));
            len_max = greater(len_max, greater(p[2], p[5]));
        }
        if (bg == BG_GAUSS) { sb2_min = lesser(sb2_min, sb2); sb2_max = greater(sb2_max, sb2); }
        if (bg == BG_GAUSS || bg == BG_FIXED_DENSITY) { f_min = lesser(f_min, f); f_max = greater(f_max, f); }
        s2_min = lesser(s2_min, s2);
        s2_max = greater(s2_max, s2);
        amp = greater(amp, a);
    }

    MCD_HD void merge(const ParamRanges& o) {
        s2_min = lesser(s2_min, o.s2_min); s2_max = greater(s2_max, o.s2_max); amp = greater(amp, o.amp);
        sb2_min = lesser(sb2_min, o.sb2_min); sb2_max = greater(sb2_max, o.sb2_max);
        f_min = lesser(f_min, o.f_min); f_max = greater(f_max, o.f_max);
        len_min = lesser(len_min, o.len_min); len_max = greater(len_max, o.len_max);
        finite = finite && o.finite;
    }

    MCD_HD static bool finite_value(double x) { return fabs(x) <= 1.7976931348623157e308; }   // false for NaN and +-inf
};

// The verdict from the catalogue's statistics and a table's ranges.
MCD_HD bool guard_verdict(const StatsScalars& st, int model, bool f32, int64_t n_rows, const ParamRanges& pr,
                          GuardRanges* ranges) {
    if (!st.stats_finite || n_rows == 0 || !pr.finite) return false;
    const bool prof = is_profile(model);
    const int bg = bg_kind(model);
    const double s2_min = pr.s2_min, s2_max = pr.s2_max, amp = pr.amp, sb2_min = pr.sb2_min, sb2_max = pr.sb2_max;
    const double f_min = pr.f_min, f_max = pr.f_max, len_min = pr.len_min, len_max = pr.len_max;
    // |v - v_los| <= |v| + |v_sys| + |v_max| (the Lynden-Bell factor 2 r r_peak / (r^2 + r_peak^2) is <= 1)
    const double d_max = st.v_abs_max + amp;
    if (!(d_max <= 0x1p58)) return false;
    // sigma_los of the profile models decays to 0 at large r: only verr^2 bounds the variance from below
    const double n_min = st.e2_min + (prof ? 0.0 : s2_min), n_max = st.e2_max + s2_max;
    if (ranges) {
        ranges->n_min = n_min; ranges->n_max = n_max; ranges->d_max = d_max;
        ranges->nb_min = st.e2_min + sb2_min; ranges->nb_max = st.e2_max + sb2_max;
        ranges->f_min = f_min; ranges->f_max = f_max;
    }
    if (prof && !(len_min >= 0x1p-100 && len_max <= 0x1p100)) return false;   // a, r_peak > 0
    if (bg == BG_NONE) {
        if (f32)      // 4-star tree in f32: DEN <= 2^60, NUM <= 2^77
            return (n_min >= 0x1p-15) && (n_max <= 0x1p15) && (d_max <= 0x1p15) &&
                   (!prof || (len_min >= 0x1p-20 && len_max <= 0x1p20));
        // 16-star tree of MODEL_CONST: DEN = prod of 16 norms within 2^+-880, NUM <= q norm^15 <= 2^100 2^825
        return (n_min >= 0x1p-55) && (n_max <= 0x1p55) && (d_max <= 0x1p50);
    }
    if (f32) {
        // float32 fast mixtures (BgFixedAccF / BgGaussAccF): four mixture values y are multiplied in float between two
        // rescales, so every y must stay within [2^-30, 2^30]; variances and residuals inside the float range of (d g)^2.
        const double flo = 0x1p-15, fhi = 0x1p15;
        if (!(n_min >= flo && n_max <= fhi && d_max <= fhi && st.extras_ok)) return false;
        if (prof && !(len_min >= 0x1p-20 && len_max <= 0x1p20)) return false;
        const double g_max_log2 = -0.5 * log2(n_min), kLog2e = 1.4426950408889634;
        if (bg == BG_FIXED) {
            // y = (1 - p) + g e^{u},  u <= log p - lnL_bg - 1/2 log 2pi:  lower bound 1 - p, upper 1 + g_max e^{-lnL_bg,min}
            if (!(st.pm_max <= 1.0 - 0x1p-20 && st.lnbg_max <= 60.0 && st.lnbg_min >= -80.0)) return false;
            return ParamRanges::greater(0.0, -(st.lnbg_min + kHalfLn2Pi) * kLog2e + g_max_log2) + 1.0 <= 30.0;
        }
        if (bg == BG_FIXED_DENSITY) {
            // y = f + g e^{u},  u <= log rho - lnL_bg - 1/2 log 2pi
            if (!(f_min >= 0x1p-20 && f_max <= 0x1p20 && st.rho_min >= 0.0 && st.lnbg_max <= 60.0 && st.lnbg_min >= -80.0)) return false;
            const double up = log2(ParamRanges::greater(st.rho_max, 0x1p-60)) - (st.lnbg_min + kHalfLn2Pi) * kLog2e + g_max_log2;
            return ParamRanges::greater(log2(f_max), up) + 1.0 <= 30.0 && st.rho_min + f_min >= 0x1p-30 && st.rho_max + f_max <= 0x1p30;
        }
        // BG_GAUSS: y = rho g + f gb e^{-delta} (or mirrored): between min and sum of the two undamped terms
        if (!(st.e2_min + sb2_min >= flo && st.e2_max + sb2_max <= fhi)) return false;
        if (!(f_min >= 0x1p-20 && f_max <= 0x1p20 && st.rho_min >= 0x1p-20 && st.rho_max <= 0x1p20)) return false;
        const double gb_max_log2 = -0.5 * log2(st.e2_min + sb2_min);
        const double g_min_log2 = -0.5 * log2(n_max), gb_min_log2 = -0.5 * log2(st.e2_max + sb2_max);
        const double hi2 = ParamRanges::greater(log2(st.rho_max) + g_max_log2, log2(f_max) + gb_max_log2) + 1.0;
        const double lo2 = ParamRanges::lesser(log2(st.rho_min) + g_min_log2, log2(f_min) + gb_min_log2);
        return hi2 <= 30.0 && lo2 >= -30.0;
    }
    const double lo = 0x1p-200, hi = 0x1p200;
    if (!((n_min >= lo) && (n_max <= hi))) return false;
    // exponent arguments stay below 1e9 in magnitude: (|v| + |v_los|)^2 / norm_min <= 1.6e9
    if (!(d_max * d_max <= 1.6e9 * n_min)) return false;
    if (!st.extras_ok) return false;
    if (bg == BG_GAUSS) {
        if (!((st.e2_min + sb2_min >= lo) && (st.e2_max + sb2_max <= hi))) return false;
        if (!(d_max * d_max <= 1.6e9 * (st.e2_min + sb2_min))) return false;
    }
    if (bg == BG_GAUSS || bg == BG_FIXED_DENSITY)
        return f_min >= 0.0 && (st.rho_min + f_min >= 0x1p-100) && (st.rho_max + f_max <= 0x1p100);
    return true;
}

// Launch level for mcd::LaunchShape::fast: 0 plain kernels, 1 fast formulation, 2 narrow-range variant (chunks that hold
// a narrow_exception star still run the fast formulation of level 1).
// BGFIXED (BgFixedAcc::add<.., NARROW>): pmember < 1 (so every mixture value y >= 1 - p >= 2^-53), lnlike_bg >= -60 and
// norm >= 2^-60 (so y <= 1 + norm^-1/2 e^60 < 2^120): eight raw factors fit between rescales; |v - v_los|^2 <= 2e6 norm.
MCD_HD int level_verdict(const StatsScalars& st, int model, bool f32, int64_t n_rows, const ParamRanges& pr) {
    GuardRanges g;
    if (!guard_verdict(st, model, f32, n_rows, pr, &g)) return 0;
    if (f32) return 1;
    // MODEL_PROFILE with a fixed centre (ProfileNarrowAcc, mcd_math.h): m = r_peak^2 + r^2 <= 2^34 arcsec^2 (a 36 degree
    // field), lengths >= 2^-10 arcsec, variances within 2^-30 .. 2^30, residuals below 2^30: the 8-star tree on
    // ((dv m - K c)^2, m^2 n) stays within 2^+-820
    if (model == MODEL_PROFILE)
        return (st.r_max_fixed > 0.0 && pr.len_min >= 0x1p-10 && pr.len_max <= 0x1p16 &&
                st.r_max_fixed * st.r_max_fixed * 1.1 + pr.len_max * pr.len_max <= 0x1p34 && g.n_min >= 0x1p-30 &&
                g.n_max <= 0x1p30 && g.d_max <= 0x1p30) ? 2 : 1;
    const double lo = 0x1p-60, hi = 0x1p60;
    // Per-call conditions of the narrow-range variants (the per-star ones are CatalogStats::narrow_exceptions):
    // d_max^2 <= 2e6 n_min keeps the exponent argument above -1.1e6, inside the int range of exp_tab (|u| < 1.4e6) without a clamp.
    if (!st.narrow_possible) return 1;
    if (model == MODEL_BGFIXED && g.n_min >= lo && g.d_max * g.d_max <= 2.0e6 * g.n_min) return 2;
    // BgFixedAcc::add_density<NARROW>: y = f + rho g e^u >= f >= 2^-20 and <= 2^20 + 2^20 2^31 e^45 < 2^117 (eight factors per rescale)
    if (bg_kind(model) == BG_FIXED_DENSITY && g.f_min >= 0x1p-20 && g.f_max <= 0x1p20 && g.n_min >= lo &&
        g.d_max * g.d_max <= 2.0e6 * g.n_min)
        return 2;
    // BgGaussAcc::add<.., NARROW>: y >= the undamped term min(rho g, f g_b) >= 2^-20 2^-31 and y <= (rho + f) 2^31 <= 2^52
    if (bg_kind(model) == BG_GAUSS && g.f_min >= 0x1p-20 && g.f_max <= 0x1p20 && g.n_min >= lo && g.n_max <= hi &&
        g.nb_min >= lo && g.nb_max <= hi && g.d_max * g.d_max <= 2.0e6 * ParamRanges::lesser(g.n_min, g.nb_min))
        return 2;
    return 1;
}

// ---------------------------------------------------------------------------------------------------------------
// float32 accuracy domain (MCD_F32, MCD_F32_ACC64; host only -- the resident chain is float64).  The float32 kernels round
// every record field and every walker constant to 24 bits before the first operation, so what they can deliver is set by
// how strongly the per-star terms amplify those roundings, not by the formulation.  Derived from a campaign of 138 000
// random cases against the float64 kernels (tools/fuzz_f32.py, profiles/r03_fuzz_f32*.txt; error on the scale
// max(|lnL|, N, 32) -- a single float32 term has an absolute floor of a few 1e-6):
//   * RANGES: variances, residuals and mixture values inside the ranges the float32 fast formulations were built for
//     (guard_verdict with f32 = true: 2^-15 <= norm <= 2^15, |v - v_los| <= 2^15, lnL_bg in [-80, 60], pmember <=
//     1 - 2^-20, density and f_back in [2^-20, 2^20], every mixture value in [2^-30, 2^30]).  Outside them the literal
//     float32 log-sum-exp of the plain kernels is what runs, and it is off by up to 3.5e-4 for the mixture models.
//   * kappa_v = d_max / sqrt(n_min), d_max = max|v| + |v_sys| + |v_maxx| + |v_maxy| (+ |v_back|), n_min the smallest total
//     variance: the residual d = v - v_los is a difference of numbers of size d_max, its rounding error 2^-24 d_max enters
//     through (d / sqrt n)^2.  Inside the ranges the campaign finds <= 2.5e-7 (float64 sums) / 3.7e-7 (float32 sums) for
//     kappa_v <= 96 and up to 1.2e-6 beyond (systemic velocities of 1000 km/s with dispersions of 1 km/s; a walker at
//     sigma -> 0): the float64 kernels are the tool there.
//   * FREE centre: the tangent-plane offsets are differences of O(1) products, their absolute error 2^-23 rad becomes a
//     position-angle error 2^-23 / separation, which the rotation term amplifies by v_rot / sqrt n.  Summed over the
//     catalogue:  kappa_theta = (max(|v_maxx| + |v_maxy|) / sqrt(n_min)) 2^-23 / sep_harm, sep_harm the harmonic mean
//     separation [rad] of the stars from the catalogue's centroid (CatalogStats).  Errors reach 8.5e-4 for compact
//     catalogues with strong rotation; they are 0.002 kappa_theta in the median, 0.12 at the 99.9th percentile and 0.51 at
//     worst over 350 000 cases (one star close to the centroid with a residual of many sigma; 12-star catalogue of the long
//     campaign, profiles/r03_long_campaign.txt): kappa_theta <= 2e-5 keeps them below 2e-5 with a factor two to spare.
// Stated tolerances inside the domain: fixed centre 1e-6 (MCD_F32_ACC64) / 2e-5 (MCD_F32: the float32 sums of 1e6 terms
// add to the per-term error, tests/test_gpu_baseline_shapes.py); free centre 2e-5 / 1e-4.
constexpr double kF32KappaV = 96.0;
constexpr double kF32KappaTheta = 2.0e-5;

struct F32Domain {
    bool inside = false;
    double kappa_v = 0.0, kappa_theta = 0.0;
    const char* reason = "";
};

inline F32Domain f32_domain(const CatalogStats& st, int model, bool free_centre, int k, const double* params, int64_t n_rows) {
    F32Domain out;
    ParamRanges pr;
    double rot = 0.0, a_min = kInfinity, off_max = 0.0;
    const bool prof = is_profile(model);
    const int ix = prof ? 3 : 2, ic = prof ? 6 : 4;
    for (int64_t i = 0; i < n_rows; ++i) {
        const double* p = params + i * k;
        pr.add_row(p, k, model, free_centre);
        rot = ParamRanges::greater(rot, std::fabs(p[ix]) + std::fabs(p[ix + 1]));
        if (prof) a_min = ParamRanges::lesser(a_min, p[2]);
        if (free_centre)        // how far this row's centre is from the reference point of the catalogue statistics [arcsec]
            off_max = ParamRanges::greater(off_max, std::hypot((p[ic] - st.ref_ra) * std::cos(st.ref_dec * 0.017453292519943295),
                                                               p[ic + 1] - st.ref_dec) * 3600.0);
    }
    GuardRanges g;
    const bool ranges_ok = guard_verdict(st, model, true, n_rows, pr, &g);
    double n_min = g.n_min;
    if (prof && st.r_max_arcsec > 0.0 && a_min > 0.0 && a_min < kInfinity) {
        // guard_verdict bounds the variance of the profile models by verr^2 alone (the Plummer dispersion decays to 0 at
        // large r); within THIS catalogue it is at least sigma_max^2 a / sqrt(a^2 + R^2) at the outermost star
        const double R = st.r_max_arcsec + off_max;
        const double floor2 = pr.s2_min * a_min / std::sqrt(a_min * a_min + R * R);
        if (floor2 > 0.0 && floor2 < kInfinity) n_min += floor2;
    }
    if (bg_kind(model) == BG_GAUSS) n_min = ParamRanges::lesser(n_min, g.nb_min);
    out.kappa_v = n_min > 0.0 ? g.d_max / std::sqrt(n_min) : kInfinity;
    out.kappa_theta = free_centre ? (st.sep_harm > 0.0 && n_min > 0.0 ? rot / std::sqrt(n_min) * 0x1p-23 / st.sep_harm : kInfinity) : 0.0;
    if (!ranges_ok) out.reason = "variances, residuals or mixture values outside the float32 ranges (norm within 2^-15 .. 2^15, |v - v_los| <= 2^15, lnL_bg within -80 .. 60, pmember <= 1 - 2^-20, density and f_back within 2^-20 .. 2^20)";
    else if (!(out.kappa_v <= kF32KappaV)) out.reason = "(max|v| + |v_sys| + |v_maxx| + |v_maxy|) / sqrt(min(verr^2) + sigma^2) exceeds 96: the float32 rounding of the velocities is amplified beyond the stated tolerance";
    else if (free_centre && !(out.kappa_theta <= kF32KappaTheta)) out.reason = "free centre: rotation amplitude over dispersion times 2^-23 / (harmonic mean separation of the stars) exceeds 2e-5: the float32 tangent-plane offsets are too coarse for this catalogue";
    else out.inside = true;
    return out;
}

inline ParamRanges table_ranges(int model, bool free_centre, int k, const double* params, int64_t n_rows) {
    ParamRanges pr;
    for (int64_t i = 0; i < n_rows; ++i) pr.add_row(params + i * k, k, model, free_centre);
    return pr;
}

inline bool fast_guard(const CatalogStats& st, int model, bool free_centre, bool f32, int k, const double* params,
                       int64_t n_rows, GuardRanges* ranges = nullptr) {
    return guard_verdict(st, model, f32, n_rows, table_ranges(model, free_centre, k, params, n_rows), ranges);
}

inline int fast_level(const CatalogStats& st, int model, bool free_centre, bool f32, int k, const double* params,
                      int64_t n_rows) {
    return level_verdict(st, model, f32, n_rows, table_ranges(model, free_centre, k, params, n_rows));
}

}  // namespace mcd
