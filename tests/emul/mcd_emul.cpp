// TEST INFRASTRUCTURE ONLY -- compiles the kernels' per-chunk arithmetic (csrc/mcd_math.h) for the CPU
// so that the fast-path algebra (fraction tree, log-product, rsqrt/exp mixture) can be checked against
// the golden vectors and the oracle without a GPU.  Never loaded by the product package.
#include <cstdint>
#include <vector>

#include "mcd_guard.h"
#include "mcd_math.h"

using namespace mcd;

static const double kExpTabHost[kExpTabSize] = {MCD_EXP_TABLE_VALUES};
static const double kExpTabSqrt2Host[kExpTabSize] = {MCD_EXP_TABLE_SQRT2_VALUES};

template <int MODEL, bool FREE, int FAST>
static void run(int64_t n, const double* recs, const double* wpar, int64_t W, int64_t chunk_len, double* out) {
    constexpr int ND = record_doubles(MODEL, FREE);
    for (int64_t w = 0; w < W; ++w) {
        WalkerConsts<double> c;
        c.load(wpar + w * KD);
        double total = 0.0;
        bool rerun = false;
        for (int64_t s = 0; s < n; s += chunk_len) {
            const int count = (int)((n - s) < chunk_len ? (n - s) : chunk_len);
            bool denormal;
            const double* tab = exp_table_is_sqrt2_scaled(MODEL) ? kExpTabSqrt2Host : kExpTabHost;
            if constexpr (FAST == 2) {
                // the library's per-chunk choice (build_workset + loglike_kernel): a chunk holding a narrow_exception star
                // takes the general fast form
                constexpr int XB = geometry_doubles(MODEL, FREE);
                constexpr int BG = bg_kind(MODEL);
                bool general = false;
                for (int64_t i = s; i < s + count && !general; ++i) {
                    const double* r = recs + i * ND;
                    general = BG == BG_FIXED ? narrow_exception(BG, r[XB], r[XB + 1], 1.0)
                            : BG == BG_GAUSS ? narrow_exception(BG, 0.0, 0.0, r[XB])
                                             : narrow_exception(BG, r[XB], 0.0, r[XB + 2]);
                }
                total += general ? chunk_loglike<MODEL, FREE, double, double, 1>(recs + s * ND, count, c, denormal, tab)
                                 : chunk_loglike<MODEL, FREE, double, double, 2>(recs + s * ND, count, c, denormal, tab);
            } else {
                total += chunk_loglike<MODEL, FREE, double, double, FAST>(recs + s * ND, count, c, denormal, tab);
            }
            rerun = rerun || denormal;
        }
        if (FAST && rerun) {                          // what the library does: the batch is re-evaluated with the plain kernels
            total = 0.0;
            for (int64_t s = 0; s < n; s += chunk_len) {
                const int count = (int)((n - s) < chunk_len ? (n - s) : chunk_len);
                bool dummy;
                total += chunk_loglike<MODEL, FREE, double, double, 0>(recs + s * ND, count, c, dummy, kExpTabHost);
            }
            out[w] = total;
            continue;
        }
        if (FAST && (bg_kind(MODEL) == BG_FIXED || bg_kind(MODEL) == BG_FIXED_DENSITY)) {
            // walker-independent sum of lnL_bg: added by the reduce kernel on the GPU
            constexpr int XB = geometry_doubles(MODEL, FREE);
            double sb = 0.0;
            for (int64_t i = 0; i < n; ++i) sb += recs[i * ND + XB];
            total += sb;
        }
        out[w] = total;
    }
}

template <int MODEL, bool FREE>
static void per_star(int64_t n, const double* recs, const double* wrow, int mode, double* out) {
    constexpr int ND = record_doubles(MODEL, FREE);
    WalkerConsts<double> c;
    c.load(wrow);
    for (int64_t i = 0; i < n; ++i) {
        double lc, lb, m;
        star_components<MODEL, FREE, double>(recs + i * ND, c, lc, lb, m);
        if (mode == 1) { out[i] = mixture_lnl(lc, lb, m); continue; }
        const double shift = is_profile(MODEL) ? max_(lc, lb) : 0.0;
        const double ec = m * std::exp(lc - shift), eb = (1.0 - m) * std::exp(lb - shift);
        out[i] = ec / (ec + eb);
    }
}

extern "C" int emul_record_doubles(int model, int free_centre) { return record_doubles(model, free_centre != 0); }
extern "C" int emul_geometry_doubles(int model, int free_centre) { return geometry_doubles(model, free_centre != 0); }
extern "C" int emul_kd() { return KD; }

#define FOR_ALL(M) CASE(M, false, false) CASE(M, false, true) CASE(M, true, false) CASE(M, true, true)
extern "C" int emul_loglike(int model, int free_centre, int fast, int64_t n, const double* recs, const double* wpar,
                            int64_t W, int64_t chunk_len, double* out) {
#define CASE(M, F, X) if (model == M && (free_centre != 0) == F && (fast != 0) == X && fast != 2) { run<M, F, X ? 1 : 0>(n, recs, wpar, W, chunk_len, out); return 0; }
    FOR_ALL(0) FOR_ALL(1) FOR_ALL(2) FOR_ALL(3) FOR_ALL(4) FOR_ALL(5)
#undef CASE
    if (fast == 2) {                          // narrow-range variants
#define NARROW_CASE(M) if (model == M) { if (free_centre) run<M, true, 2>(n, recs, wpar, W, chunk_len, out); else run<M, false, 2>(n, recs, wpar, W, chunk_len, out); return 0; }
        NARROW_CASE(1) NARROW_CASE(2) NARROW_CASE(4) NARROW_CASE(5)
#undef NARROW_CASE
    }
    return -1;
}

extern "C" int emul_per_star(int model, int free_centre, int mode, int64_t n, const double* recs, const double* wrow,
                             double* out) {
#define CASE(M, F, X) if (X && model == M && (free_centre != 0) == F) { per_star<M, F>(n, recs, wrow, mode, out); return 0; }
    FOR_ALL(1) FOR_ALL(2) FOR_ALL(4) FOR_ALL(5)
#undef CASE
    return -1;
}

// the library's own range guard (csrc/mcd_guard.h) on raw host columns and a C-ABI-ordered parameter table
extern "C" int emul_fast_guard(int model, int free_centre, int f32, int64_t n, const double* v, const double* verr,
                               const double* lnbg, const double* pmember, const double* density, int k,
                               const double* params, int64_t n_rows) {
    const CatalogStats st = compute_stats(n, v, verr, lnbg, pmember, density, bg_kind(model));
    return fast_level(st, model, free_centre != 0, f32 != 0, k, params, n_rows);
}

// background.SingleStars: the device's per-lane slice arithmetic (KdeLane) and the slice combination of
// csrc/mcd_kde.hip: kde_combine_kernel, on the CPU
extern "C" int emul_kde(int64_t m, const double* comp, int64_t n, const double* v, const double* verr, double sigma_int,
                        int64_t slice_len, double* out) {
    if (m <= 0 || slice_len <= 0) return -1;
    const int64_t n_slices = (m + slice_len - 1) / slice_len;
    std::vector<double> dmins(n_slices), sums(n_slices);
    for (int64_t i = 0; i < n; ++i) {
        KdeLane a;
        a.init(v[i], verr[i], sigma_int * sigma_int);
        for (int64_t s = 0; s < n_slices; ++s) {
            const int64_t j0 = s * slice_len, j1 = std::min(m, j0 + slice_len);
            double dmin = INFINITY;
            for (int64_t j = j0; j < j1; ++j) a.nearest(comp[j], dmin);
            a.begin_sum(dmin);
            for (int64_t j = j0; j < j1; ++j) a.add(comp[j], kExpTabHost);
            dmins[s] = dmin;
            sums[s] = a.sum;
        }
        const double norm = verr[i] * verr[i] + sigma_int * sigma_int, h = 0.5 / norm;
        double dmin = dmins[0];
        for (int64_t s = 1; s < n_slices; ++s) dmin = std::fmin(dmin, dmins[s]);
        const double d2 = dmin * dmin;
        double total = 0.0;
        for (int64_t s = 0; s < n_slices; ++s) total += sums[s] * std::exp((d2 - dmins[s] * dmins[s]) * h);
        out[i] = -d2 * h + std::log(total / std::sqrt(2.0 * 3.14159265358979323846 * norm)) - std::log((double)m);
    }
    return 0;
}
