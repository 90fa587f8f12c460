// TEST INFRASTRUCTURE ONLY -- compiles the kernels' per-chunk arithmetic (csrc/mcd_math.h) for the CPU
// so that the fast-path algebra (fraction tree, log-product, rsqrt/exp mixture) can be checked against
// the golden vectors and the oracle without a GPU.  Never loaded by the product package.
#include <cstdint>
#include <cstring>
#include <vector>

#include "mcd_chunks.h"
#include "mcd_guard.h"
#include "mcd_math.h"
#include "mcd_rng.h"
#include "mcd_stretch.h"

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
                for (int64_t i = s; BG != BG_NONE && i < s + count && !general; ++i) {
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
    FOR_ALL(0) FOR_ALL(1) FOR_ALL(2) FOR_ALL(3) FOR_ALL(4) FOR_ALL(5) FOR_ALL(6)
#undef CASE
    if (fast == 2) {                          // narrow-range variants
#define NARROW_CASE(M) if (model == M) { if (free_centre) run<M, true, 2>(n, recs, wpar, W, chunk_len, out); else run<M, false, 2>(n, recs, wpar, W, chunk_len, out); return 0; }
        NARROW_CASE(1) NARROW_CASE(2) NARROW_CASE(4) NARROW_CASE(5) NARROW_CASE(6)
        if (model == 3 && !free_centre) { run<3, false, 2>(n, recs, wpar, W, chunk_len, out); return 0; }   // ProfileNarrowAcc
#undef NARROW_CASE
    }
    return -1;
}

extern "C" int emul_per_star(int model, int free_centre, int mode, int64_t n, const double* recs, const double* wrow,
                             double* out) {
#define CASE(M, F, X) if (X && model == M && (free_centre != 0) == F) { per_star<M, F>(n, recs, wrow, mode, out); return 0; }
    FOR_ALL(1) FOR_ALL(2) FOR_ALL(4) FOR_ALL(5) FOR_ALL(6)
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

// float32 accuracy domain (csrc/mcd_guard.h: f32_domain): verdict, the two condition numbers and the reason
extern "C" int emul_f32_domain(int model, int free_centre, int64_t n, const double* ra, const double* dec, const double* v,
                               const double* verr, const double* lnbg, const double* pmember, const double* density, int k,
                               const double* params, int64_t n_rows, double* kappa, char* reason, int reason_cap,
                               double ra_c, double dec_c) {
    const CatalogStats st = compute_stats(n, v, verr, lnbg, pmember, density, bg_kind(model), ra, dec, free_centre == 0, ra_c, dec_c);
    const F32Domain d = f32_domain(st, model, free_centre != 0, k, params, n_rows);
    kappa[0] = d.kappa_v; kappa[1] = d.kappa_theta; kappa[2] = st.sep_harm;
    if (reason && reason_cap > 0) { std::strncpy(reason, d.reason, reason_cap - 1); reason[reason_cap - 1] = 0; }
    return d.inside ? 1 : 0;
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

// ---------------------------------------------------------------------------------------------------------------
// Work decomposition (csrc/mcd_chunks.h): the chunk table the library builds for one shard, and a complete sharded
// evaluation carried out the way the library does it -- per shard: plan_chunks, one partial sum per (chunk, walker)
// with the kernel family the chunk's flag selects, the walker-independent background sum of the shard's stars, then
// the sum over shards (what the all-reduce delivers).
extern "C" int64_t emul_plan_chunks(int64_t n_psets, const int64_t* bin_offsets, int64_t star_begin, int64_t n,
                                    int64_t n_walkers, int64_t target_waves, int tail_split, int64_t n_exc,
                                    const int64_t* exc, int64_t cap, int64_t* begin, int32_t* count, int32_t* pset,
                                    uint8_t* general, int64_t* offsets, int64_t* info, int balance) {
    const std::vector<int64_t> offs(bin_offsets, bin_offsets + n_psets + 1);
    const std::vector<int64_t> ex(exc, exc + n_exc);
    const ChunkPlan plan = plan_chunks(offs, star_begin, n, n_walkers, target_waves, tail_split, ex, 0, balance);
    const int64_t nc = (int64_t)plan.chunks.size();
    if (nc > cap) return -nc;
    for (int64_t i = 0; i < nc; ++i) {
        begin[i] = plan.chunks[i].begin; count[i] = plan.chunks[i].count; pset[i] = plan.chunks[i].pset;
        general[i] = plan.general.empty() ? 0 : plan.general[i];
    }
    for (int64_t p = 0; p <= n_psets; ++p) offsets[p] = plan.offsets[p];
    info[0] = plan.max_chunks_per_pset; info[1] = plan.len; info[2] = plan.uniform_len;
    info[3] = plan.general.empty() ? 0 : 1; info[4] = main_grid(nc, n_walkers);
    info[5] = plan.uniform_extra; info[6] = plan.balanced_m;
    // the arithmetic form the main kernel computes instead of loading a descriptor must BE the table
    if (plan.uniform_len > 0)
        for (int64_t i = 0; i < nc; ++i) {
            const Chunk u = uniform_chunk(i, plan.uniform_len, plan.uniform_extra, n, nc);
            if (u.begin != plan.chunks[i].begin || u.count != plan.chunks[i].count) info[5] = -1;
        }
    return nc;
}

extern "C" void emul_shard_range(int64_t n_stars, int i, int n_shards, int64_t* out) {
    const ShardRange r = shard_range(n_stars, i, n_shards);
    out[0] = r.begin; out[1] = r.n;
}

extern "C" void emul_pset_background_sums(const double* lnbg, int64_t n_psets, const int64_t* bin_offsets,
                                          int64_t star_begin, int64_t n, double* out) {
    const std::vector<int64_t> offs(bin_offsets, bin_offsets + n_psets + 1);
    const std::vector<double> sums = pset_background_sums(lnbg, offs, star_begin, n);
    for (int64_t p = 0; p < n_psets; ++p) out[p] = sums[p];
}

template <int MODEL, bool FREE>
static void sharded(int level, int64_t n, const double* recs, int64_t n_psets, const int64_t* bin_offsets,
                    const double* lnbg, const std::vector<int64_t>& exc, int64_t W, const double* wpar, int n_shards,
                    int64_t target_waves, int tail_split, double* out, int64_t* n_general) {
    constexpr int ND = record_doubles(MODEL, FREE);
    constexpr int BG = bg_kind(MODEL);
    const std::vector<int64_t> offs(bin_offsets, bin_offsets + n_psets + 1);
    const double* tab = exp_table_is_sqrt2_scaled(MODEL) ? kExpTabSqrt2Host : kExpTabHost;
    for (int64_t i = 0; i < n_psets * W; ++i) out[i] = 0.0;
    *n_general = 0;
    for (int sidx = 0; sidx < n_shards; ++sidx) {
        const ShardRange sr = shard_range(n, sidx, n_shards);
        const ChunkPlan plan = plan_chunks(offs, sr.begin, sr.n, W, target_waves, tail_split, exc);
        const double* shard_recs = recs + sr.begin * ND;            // what this device holds
        std::vector<double> acc(n_psets * W, 0.0);
        for (size_t c = 0; c < plan.chunks.size(); ++c) {
            const Chunk& ch = plan.chunks[c];
            const bool general = !plan.general.empty() && plan.general[c];
            if (general) ++*n_general;
            for (int64_t w = 0; w < W; ++w) {
                WalkerConsts<double> wc;
                wc.load(wpar + ((int64_t)ch.pset * W + w) * KD);
                bool den;
                double r;
                if (level == 0) r = chunk_loglike<MODEL, FREE, double, double, 0>(shard_recs + ch.begin * ND, ch.count, wc, den, tab);
                else if (level == 1 || general || BG == BG_NONE) r = chunk_loglike<MODEL, FREE, double, double, 1>(shard_recs + ch.begin * ND, ch.count, wc, den, tab);
                else r = chunk_loglike<MODEL, FREE, double, double, (BG == BG_NONE ? 1 : 2)>(shard_recs + ch.begin * ND, ch.count, wc, den, tab);
                acc[(int64_t)ch.pset * W + w] += r;
            }
        }
        if (level && lnbg && (BG == BG_FIXED || BG == BG_FIXED_DENSITY)) {
            const std::vector<double> sums = pset_background_sums(lnbg, offs, sr.begin, sr.n);
            for (int64_t p = 0; p < n_psets; ++p)
                for (int64_t w = 0; w < W; ++w) acc[p * W + w] += sums[p];
        }
        for (int64_t i = 0; i < n_psets * W; ++i) out[i] += acc[i];          // the all-reduce
    }
}

extern "C" int emul_sharded_loglike(int model, int free_centre, int level, int64_t n, const double* recs,
                                    int64_t n_psets, const int64_t* bin_offsets, const double* lnbg, int64_t n_exc,
                                    const int64_t* exc, int64_t W, const double* wpar, int n_shards,
                                    int64_t target_waves, int tail_split, double* out, int64_t* n_general) {
    const std::vector<int64_t> ex(exc, exc + n_exc);
#define CASE(M, F, X) if (X && model == M && (free_centre != 0) == F) { sharded<M, F>(level, n, recs, n_psets, bin_offsets, lnbg, ex, W, wpar, n_shards, target_waves, tail_split, out, n_general); return 0; }
    FOR_ALL(0) FOR_ALL(1) FOR_ALL(2) FOR_ALL(3) FOR_ALL(4) FOR_ALL(5) FOR_ALL(6)
#undef CASE
    return -1;
}

// narrow_exception star list of a catalogue (mcd_guard.h: compute_stats), ascending global indices
extern "C" int64_t emul_narrow_exceptions(int model, int64_t n, const double* v, const double* verr, const double* lnbg,
                                          const double* pmember, const double* density, int64_t cap, int64_t* out) {
    const CatalogStats st = compute_stats(n, v, verr, lnbg, pmember, density, bg_kind(model));
    if (!st.narrow_possible) return -1;
    const int64_t m = (int64_t)st.narrow_exceptions.size();
    for (int64_t i = 0; i < m && i < cap; ++i) out[i] = st.narrow_exceptions[i];
    return m;
}

// ---------------------------------------------------------------------------------------------------------------
// The library's stretch-move block (csrc/mcd_stretch.h) with the likelihood supplied by the test as a callback.
typedef int (*emul_eval_fn)(const double* table, int64_t n, double* out);
extern "C" int emul_stretch_block(int64_t B, int64_t W, int P, int K, const int32_t* col_source, const double* col_const,
                                  const double* col_factor, const double* lo, const double* hi, int fixed_ok,
                                  int64_t n_steps, double* pos, double* lnp, const int32_t* order, const double* zz,
                                  const double* thr, const int32_t* pick, double* chain, double* lnprob_chain,
                                  int64_t* accepted, emul_eval_fn eval) {
    StretchDesc d;
    d.n_bins = B; d.n_walkers = W; d.n_dim = P; d.k = K; d.col_source = col_source; d.col_const = col_const; d.col_factor = col_factor;
    d.lo = lo; d.hi = hi; d.fixed_ok = fixed_ok;
    return stretch_block(d, n_steps, pos, lnp, order, zz, thr, pick, chain, lnprob_chain, accepted,
                         [&](const double* t, int64_t n, double* out) { return eval(t, n, out); });
}

// ---------------------------------------------------------------------------------------------------------------
// The chain's counter-based random numbers (csrc/mcd_rng.h), compiled for the host.
extern "C" void emul_philox(const uint64_t* counter, const uint64_t* key, uint64_t* out) {
    const Philox4x64 r = philox4x64_10(counter[0], counter[1], counter[2], counter[3], key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}
extern "C" void emul_det_log(int64_t n, const double* x, double* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = det_log(x[i]);
}
extern "C" void emul_chain_numbers(uint64_t seed, int64_t step0, int64_t n_steps, int64_t B, int64_t W, int n_dim,
                                   int32_t* order, double* zz, double* thr, int32_t* pick) {
    std::vector<uint64_t> sorter;
    const int64_t half = W / 2;
    for (int64_t i = 0; i < n_steps; ++i)
        chain_numbers_of_step(seed, step0 + i, B, W, n_dim, order + i * B * W, zz + i * 2 * B * half, thr + i * 2 * B * half,
                              pick + i * 2 * B * half, sorter);
}
extern "C" void emul_chain_keys(uint64_t seed, int64_t step, int64_t b, int64_t W, uint64_t* out) {
    for (int64_t w = 0; w < W; ++w) out[w] = chain_draw(seed, step, (int)(w / (W / 2)), b, w % (W / 2), W / 2, 1).order_key >> kOrderKeyShift;
}
