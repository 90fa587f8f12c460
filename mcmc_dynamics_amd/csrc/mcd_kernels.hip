// mcd_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the per-walker log-likelihood hot path.
//
// Work decomposition:
//   * lane  = walker.  A 64-lane wave evaluates 64 walkers against one chunk of stars.
//   * star records are wave-uniform: the compiler turns the record reads into s_load_dwordxN
//     (scalar cache, broadcast to all lanes for free as SGPR operands of the f64 VALU ops), so one
//     32..64-byte record load feeds 64 star-walker terms and no LDS/VGPR staging is spent on it.
//   * each wave keeps its walkers' running sums in registers; no cross-lane traffic in the hot loop.
//   * per-(walker, chunk) partials go to HBM as partials[walker / 8][chunk][walker % 8] (a wave's store is eight
//     full 64-byte segments); a second kernel reduces them with a fixed tree (wave shuffle + LDS), so results are
//     bitwise reproducible.  No float atomics anywhere.
//
// The kernel is bound by the f64 VALU issue rate, not by HBM: the catalogue (32..64 B/star) is read
// once per 64..256 walkers.  MFMA has nothing to offer here (no contraction).
#include "mcd_internal.h"
#include "mcd_math.h"
#include "mcd_prep.h"

namespace mcd {

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;               // 4 waves: one per SIMD of a CU
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kPartialGroup = 8;            // walkers per group of the partial-sum array (8 doubles = one 64-byte segment)

constexpr double kR0Arcmin = 3437.7467707849392526;   // 10800 / pi, calc_xy_offset.py:11

// ------------------------------------------------------------------------------------------------
// star prep: raw float64 columns -> packed records (one thread per star; runs once per catalogue)
template <class T>
__global__ __launch_bounds__(kBlock) void prepare_records_kernel(RawColumns raw, int64_t n, int model,
                                                                  int free_centre, double ra_c, double dec_c,
                                                                  T* __restrict__ rec) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int nd = record_doubles(model, free_centre != 0);
    T* r = rec + i * nd;
    const double verr = raw.verr[i];
    r[0] = (T)raw.v[i];
    r[1] = (T)(verr * verr);                                        // runner.py:261 (verr * verr)
    const double ra = raw.ra[i], dec = raw.dec[i];
    const int k = geometry_doubles(model, free_centre != 0);
    if (free_centre) {
        double sa, ca, sd, cd;
        sincos(ra * kDeg2Rad, &sa, &ca);
        sincos(dec * kDeg2Rad, &sd, &cd);
        r[2] = (T)(cd * sa); r[3] = (T)(cd * ca); r[4] = (T)sd; r[5] = (T)0;      // A, B, sin(dec) of free_centre_xy
    } else {
        // calc_xy_offset.py:30-31
        const double dra = (ra - ra_c) * kDeg2Rad;
        const double dec_r = dec * kDeg2Rad, dec_cr = dec_c * kDeg2Rad;
        const double dx = -kR0Arcmin * cos(dec_r) * sin(dra);
        const double dy = kR0Arcmin * (sin(dec_r) * cos(dec_cr) - cos(dec_r) * sin(dec_cr) * cos(dra));
        if (is_profile(model)) {
            // model.py:124-127, 171-180 use r, dx, dy themselves; arcsec because a and r_peak are in arcsec
            const double xs = 60.0 * dx, ys = 60.0 * dy;
            r[2] = (T)xs; r[3] = (T)ys; r[4] = (T)(xs * xs + ys * ys); r[5] = (T)0;
        } else {
            // followed by arctan2 (constant.py:107), reduced to sin/cos(theta)
            const double rr = hypot(dx, dy);
            double s, c;
            if (rr > 0.0) { s = dy / rr; c = dx / rr; }
            else { s = 0.0; c = signbit(dx) ? -1.0 : 1.0; }        // numpy arctan2(+0, -0) = pi
            r[2] = (T)s; r[3] = (T)c;
        }
    }
    const int bg = bg_kind(model);
    if (bg == BG_FIXED) {
        // fast paths: exponent offset with the prior weight folded in, log p - (b + 1/2 log 2pi); the floor keeps it finite
        // for p == 0 (e^-2000 is an exact 0 in f64, so y = 1 - p exactly as it must)
        const double b = raw.lnbg[i], pm = raw.pmember[i];
        r[k] = (T)b; r[k + 1] = (T)pm; r[k + 2] = (T)(1.0 - pm); r[k + 3] = (T)fmax(log(pm) - (b + kHalfLn2Pi), -2000.0);
    } else if (bg == BG_GAUSS) {
        r[k] = (T)raw.density[i]; r[k + 1] = (T)0;
    } else if (bg == BG_FIXED_DENSITY) {
        const double b = raw.lnbg[i];
        r[k] = (T)b; r[k + 1] = (T)fmax(log(raw.density[i]) - (b + kHalfLn2Pi), -2000.0); r[k + 2] = (T)raw.density[i];
        r[k + 3] = (T)0;
    }
}

// ------------------------------------------------------------------------------------------------
// walker prep: resolved parameter rows (reference order) -> derived constants, one thread per row (mcd_prep.h)
template <class T>
__global__ __launch_bounds__(kBlock) void prepare_walkers_kernel(const double* __restrict__ params, int64_t n_rows,
                                                                  int k, int model, int free_centre,
                                                                  T* __restrict__ wpar) {
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_rows) return;
    walker_constants<T>(params + i * k, model, free_centre != 0, wpar + i * KD);
}

__device__ const double kExpTabDevice[kExpTabSize] = {MCD_EXP_TABLE_VALUES};
__device__ const double kExpTabSqrt2Device[kExpTabSize] = {MCD_EXP_TABLE_SQRT2_VALUES};

// ------------------------------------------------------------------------------------------------
// main kernel (the per-chunk arithmetic lives in mcd_math.h: chunk_loglike)
template <int MODEL, bool FREE, class T, class A, int FAST, bool PF>
__global__ __launch_bounds__(kBlock) void loglike_kernel(const T* __restrict__ recs,
                                                          const Chunk* __restrict__ chunks,
                                                          const T* __restrict__ wpar,
                                                          double* __restrict__ partials, int64_t n_tasks,
                                                          int n_wtiles, int64_t n_walkers, int64_t n_chunks,
                                                          int uniform_len, int64_t n_records,
                                                          double* __restrict__ rerun_flag, double launch_tag,
                                                          const uint8_t* __restrict__ chunk_general) {
    constexpr int ND = record_doubles(MODEL, FREE);
    // fast mixtures: the 2^(j/1024) table of exp_tab lives in LDS (8 KiB per workgroup), four entries copied per thread
    constexpr bool kUsesExpTab = FAST && bg_kind(MODEL) != BG_NONE && sizeof(T) == 8;
    __shared__ double exptab_lds[kUsesExpTab ? kExpTabSize : 1];
    if constexpr (kUsesExpTab) {
        static_assert(kExpTabSize % kBlock == 0, "whole table entries per thread");
        const double* __restrict__ src = exp_table_is_sqrt2_scaled(MODEL) ? kExpTabSqrt2Device : kExpTabDevice;
#pragma unroll
        for (int i = 0; i < kExpTabSize / kBlock; ++i) exptab_lds[i * kBlock + threadIdx.x] = src[i * kBlock + threadIdx.x];
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & (kWave - 1);
    int64_t chunk_id;
    int wtile;
    if (n_wtiles <= kWavesPerBlock) {
        // <= 256 walkers: consecutive waves share a chunk, so every chunk is read by one workgroup (one CU, one XCD)
        const int64_t task = (int64_t)blockIdx.x * kWavesPerBlock + wave;  // wave-uniform
        if (task >= n_tasks) return;
        chunk_id = task / n_wtiles;
        wtile = (int)(task - chunk_id * n_wtiles);
    } else {
        // > 256 walkers: a chunk needs m = ceil(n_wtiles / 4) workgroups.  Workgroups are dealt round-robin over the
        // 8 XCDs, so workgroups b and b + 8 share an XCD (and its L2): within a group of 8 m workgroups, workgroup j
        // takes chunk j % 8 and walker-tile quartet j / 8 -- all m readers of a chunk sit on one XCD and the chunk is
        // fetched from HBM once.  (Placement only affects traffic, never results.)
        const int m = (n_wtiles + kWavesPerBlock - 1) / kWavesPerBlock;
        const int64_t group = blockIdx.x / (8 * m);
        const int j = (int)(blockIdx.x - group * (8 * m));
        chunk_id = group * 8 + (j & 7);
        wtile = (j >> 3) * kWavesPerBlock + wave;
        if (chunk_id >= n_chunks || wtile >= n_wtiles) return;
    }
    // Equal-length chunk tables of a single parameter set are addressed arithmetically: the first record load
    // then does not wait behind a descriptor load (three dependent memory latencies at wave start become one).
    Chunk ch;
    if (uniform_len > 0) {
        ch.begin = chunk_id * uniform_len;
        const int64_t left = n_records - ch.begin;
        ch.count = (int32_t)(left < uniform_len ? left : uniform_len);
        ch.pset = 0;
    } else {
        ch = chunks[chunk_id];                                               // scalar load
    }
    const int64_t w_raw = (int64_t)wtile * kWave + lane;
    const bool active = w_raw < n_walkers;
    const int64_t w_idx = active ? w_raw : n_walkers - 1;                   // idle lanes shadow the last walker

    const T* __restrict__ wp = wpar + ((int64_t)ch.pset * n_walkers + w_idx) * KD;
    WalkerConsts<T> w;
    w.load(wp);                                   // unused constants are dead code for a given MODEL

    // wave-uniform record pointer in the constant address space: the reads inside chunk_loglike are scalar loads
    const RecPtr<T> chunk_recs = (RecPtr<T>)(recs + ch.begin * ND);
    bool denormal;
    double result;
    if constexpr (FAST == 2) {
        // a chunk that holds a star outside the narrow-range domain (certain member, extreme background, ...) takes the
        // general fast form; the flag is wave-uniform (one scalar byte load), so this is a scalar branch
        const bool general = chunk_general != nullptr && chunk_general[chunk_id] != 0;
        if (general) result = chunk_loglike<MODEL, FREE, T, A, 1, PF>(chunk_recs, ch.count, w, denormal, exptab_lds);
        else result = chunk_loglike<MODEL, FREE, T, A, 2, PF>(chunk_recs, ch.count, w, denormal, exptab_lds);
    } else {
        result = chunk_loglike<MODEL, FREE, T, A, FAST, PF>(chunk_recs, ch.count, w, denormal, exptab_lds);
    }
    // denormal regime of the reference's log-sum-exp met: tell the host to re-evaluate this batch with the plain kernels
    // One device: the flag word gets this launch's tag.  Several ranks / devices (rerun_flag == nullptr): the partial sum
    // itself becomes NaN, which survives the reduce kernel and the all-reduce, so every rank sees it in the same walkers
    // and takes the same decision -- no flag word to zero before every launch.
    if (FAST && denormal && active) {
        if (rerun_flag) *rerun_flag = launch_tag;
        else result = __builtin_nan("");
    }
    // partials[walker group of 8][chunk][walker in group]: eight full 64-byte segments per wave store (the rows are
    // padded to whole walker tiles, so idle lanes store their shadow value into padding), and the reduce kernel streams
    // one contiguous [chunk][8] block per walker group.
    partials[((w_raw >> 3) * n_chunks + chunk_id) * kPartialGroup + (w_raw & (kPartialGroup - 1))] = result;
}

// ------------------------------------------------------------------------------------------------
// final reduction: one block per (parameter set, group of 8 walkers).  The group's partial sums are one contiguous
// [chunk][8] array of 64-byte segments.  Thread t = (sublane s = t / 4, quarter q = t % 4) reads the 16 bytes of walkers
// 2q, 2q + 1 of the chunks c0 + s, c0 + s + 2 SUB, ...: a wave's load covers sixteen whole segments (1 KiB, contiguous),
// and kReduceUnroll independent accumulator pairs keep that many loads in flight per thread -- the kernel is bound by
// memory latency (6 MB over 16..32 workgroups), and with 8-byte loads, four in flight, it took 6.5 us where this takes 3.
// The sublanes are combined by a wave shuffle tree and a fixed-order sum over the waves: the order of every addition is
// fixed by (n_chunks, SUB) alone -- bitwise repeatable.
constexpr int kReduceUnroll = 8;

template <int SUB>
__global__ __launch_bounds__(kPartialGroup * SUB) void reduce_group_kernel(const double* __restrict__ partials,
                                                                            const int64_t* __restrict__ offs,
                                                                            int64_t n_chunks, int64_t n_walkers,
                                                                            int64_t n_groups,
                                                                            const double* __restrict__ pset_const,
                                                                            double* __restrict__ out) {
    constexpr int kThreads = kPartialGroup * SUB;
    constexpr int kWaves = kThreads / kWave;
    constexpr int kSublanes = kThreads / 4;
    __shared__ double lds[kWaves > 1 ? kWaves : 1][kPartialGroup];
    const int64_t pset = blockIdx.x / n_groups, g = blockIdx.x - pset * n_groups;
    const int q = threadIdx.x & 3, s = threadIdx.x >> 2;
    const int64_t c0 = offs[pset], c1 = offs[pset + 1];
    const double2* __restrict__ col = reinterpret_cast<const double2*>(partials + g * n_chunks * kPartialGroup) + q;
    double2 a[kReduceUnroll];
#pragma unroll
    for (int u = 0; u < kReduceUnroll; ++u) a[u] = make_double2(0.0, 0.0);
    int64_t c = c0 + s;
    for (; c + (kReduceUnroll - 1) * kSublanes < c1; c += kReduceUnroll * kSublanes) {
#pragma unroll
        for (int u = 0; u < kReduceUnroll; ++u) {
            const double2 v = col[(c + u * kSublanes) * 4];
            a[u].x += v.x;
            a[u].y += v.y;
        }
    }
#pragma unroll
    for (int u = 0; u < kReduceUnroll; ++u) {
        if (c < c1) {
            const double2 v = col[c * 4];
            a[u].x += v.x;
            a[u].y += v.y;
        }
        c += kSublanes;
    }
    double ax = ((a[0].x + a[1].x) + (a[2].x + a[3].x)) + ((a[4].x + a[5].x) + (a[6].x + a[7].x));
    double ay = ((a[0].y + a[1].y) + (a[2].y + a[3].y)) + ((a[4].y + a[5].y) + (a[6].y + a[7].y));
    static_assert(kReduceUnroll == 8, "the combining tree above is written for eight accumulators");
    // the 16 sublanes of a wave: lanes q, q + 4, ..., q + 60
#pragma unroll
    for (int off = 32; off >= 4; off >>= 1) {
        ax += __shfl_xor(ax, off, kWave);
        ay += __shfl_xor(ay, off, kWave);
    }
    const int lane = threadIdx.x & (kWave - 1);
    if constexpr (kWaves > 1) {
        if (lane < 4) { lds[threadIdx.x >> 6][2 * q] = ax; lds[threadIdx.x >> 6][2 * q + 1] = ay; }
        __syncthreads();
        if (threadIdx.x < kPartialGroup) {
            double acc = lds[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) acc += lds[w][threadIdx.x];
            const int64_t w = g * kPartialGroup + threadIdx.x;
            if (w < n_walkers) out[pset * n_walkers + w] = acc + (pset_const ? pset_const[pset] : 0.0);
        }
    } else {
        // one wave: lanes 0..3 hold the sums of walkers (2q, 2q + 1)
        if (lane < 4) {
            const int64_t w = g * kPartialGroup + 2 * q;
            const double add = pset_const ? pset_const[pset] : 0.0;      // walker-independent part (sum of lnL_bg)
            if (w < n_walkers) out[pset * n_walkers + w] = ax + add;
            if (w + 1 < n_walkers) out[pset * n_walkers + w + 1] = ay + add;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per-star outputs for ONE parameter row (one thread per star):
//   mode 0: membership probability  m e^{lc} / (m e^{lc} + (1 - m) e^{lb})   (constant.py:366-374; the ModelFit
//           classes subtract max(lc, lb) first, model.py:505-510, 680-687)
//   mode 1: per-star log-likelihood of the mixture, `lnlike(no_sum=True)` of model.py:565-623
template <int MODEL, bool FREE, class T>
__global__ __launch_bounds__(kBlock) void per_star_kernel(const T* __restrict__ recs, int64_t n,
                                                           const T* __restrict__ wp, int mode,
                                                           double* __restrict__ out) {
    constexpr int ND = record_doubles(MODEL, FREE);
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    WalkerConsts<T> w;
    w.load(wp);
    T lc, lb, m;
    star_components<MODEL, FREE, T>((RecPtr<T>)(recs + i * ND), w, lc, lb, m);
    if (mode == 1) { out[i] = (double)mixture_lnl(lc, lb, m); return; }
    const T shift = is_profile(MODEL) ? max_(lc, lb) : T(0);
    const T ec = m * exp_(lc - shift), eb = (T(1) - m) * exp_(lb - shift);
    out[i] = (double)(ec / (ec + eb));
}

template <int MODEL, bool FREE, class T, class A, int FAST>
hipError_t launch_one(hipStream_t s, const void* records, const Chunk* chunks, int64_t n_chunks, const void* wpar,
                      double* partials, int64_t n_walkers, int uniform_len, int64_t n_records, double* rerun_flag,
                      double launch_tag, const uint8_t* chunk_general, int prefetch) {
    const int n_wtiles = (int)((n_walkers + kWave - 1) / kWave);
    const int64_t n_tasks = n_chunks * n_wtiles;
    const int64_t grid = main_grid(n_chunks, n_walkers);   // > 256 walkers: XCD-aware grouping, see loglike_kernel
    if (grid <= 0) return hipSuccess;
    // the prefetching instantiation exists for the fast formulations only (the plain kernels have no prefetch code)
    if (FAST != 0 && prefetch)
        hipLaunchKernelGGL((loglike_kernel<MODEL, FREE, T, A, FAST, FAST != 0>), dim3((unsigned)grid), dim3(kBlock), 0, s,
                           (const T*)records, chunks, (const T*)wpar, partials, n_tasks, n_wtiles, n_walkers, n_chunks,
                           uniform_len, n_records, rerun_flag, launch_tag, chunk_general);
    else
        hipLaunchKernelGGL((loglike_kernel<MODEL, FREE, T, A, FAST, false>), dim3((unsigned)grid), dim3(kBlock), 0, s,
                           (const T*)records, chunks, (const T*)wpar, partials, n_tasks, n_wtiles, n_walkers, n_chunks,
                           uniform_len, n_records, rerun_flag, launch_tag, chunk_general);
    return hipGetLastError();
}

template <int MODEL, bool FREE>
hipError_t launch_precision(hipStream_t s, const LaunchShape& sh, const void* records, const Chunk* chunks,
                            int64_t n_chunks, const void* wpar, double* partials, int64_t n_walkers) {
    const int uniform_len = sh.uniform_len;
    const int64_t n_records = sh.n_records;
    double* const rerun_flag = sh.rerun_flag;
    const double launch_tag = sh.launch_tag;
    const uint8_t* const chunk_general = sh.chunk_general;
    const int prefetch = sh.prefetch ? 1 : 0;
    switch (sh.precision) {
        case 0:
            if constexpr (bg_kind(MODEL) != BG_NONE) {
                if (sh.fast == 2)
                    return launch_one<MODEL, FREE, double, double, 2>(s, records, chunks, n_chunks, wpar, partials,
                                                                      n_walkers, uniform_len, n_records, rerun_flag, launch_tag, chunk_general, prefetch);
            }
            if (sh.fast)
                return launch_one<MODEL, FREE, double, double, 1>(s, records, chunks, n_chunks, wpar, partials,
                                                                  n_walkers, uniform_len, n_records, rerun_flag, launch_tag, chunk_general, prefetch);
            return launch_one<MODEL, FREE, double, double, 0>(s, records, chunks, n_chunks, wpar, partials, n_walkers, uniform_len, n_records, rerun_flag, launch_tag, chunk_general, prefetch);
        case 1:      // float32 terms and sums: fast formulations for every model when the f32 guard admits them (mcd_guard.h)
            if (sh.fast)
                return launch_one<MODEL, FREE, float, float, 1>(s, records, chunks, n_chunks, wpar, partials, n_walkers,
                                                                uniform_len, n_records, rerun_flag, launch_tag, chunk_general, prefetch);
            return launch_one<MODEL, FREE, float, float, 0>(s, records, chunks, n_chunks, wpar, partials, n_walkers, uniform_len, n_records, rerun_flag, launch_tag, chunk_general, prefetch);
        case 2:      // float32 terms, float64 accumulation
            if (sh.fast)
                return launch_one<MODEL, FREE, float, double, 1>(s, records, chunks, n_chunks, wpar, partials, n_walkers,
                                                                 uniform_len, n_records, rerun_flag, launch_tag, chunk_general, prefetch);
            return launch_one<MODEL, FREE, float, double, 0>(s, records, chunks, n_chunks, wpar, partials, n_walkers, uniform_len, n_records, rerun_flag, launch_tag, chunk_general, prefetch);
    }
    return hipErrorInvalidValue;
}

}  // namespace

int record_bytes(int model, bool free_centre, int precision) {
    return record_doubles(model, free_centre) * (precision == 0 ? 8 : 4);
}

hipError_t launch_prepare_records(hipStream_t s, const RawColumns& raw, int64_t n, int model, bool free_centre,
                                  int precision, double ra_c_deg, double dec_c_deg, void* records) {
    if (n <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    if (precision == 0)
        hipLaunchKernelGGL(prepare_records_kernel<double>, dim3(grid), dim3(kBlock), 0, s, raw, n, model,
                           (int)free_centre, ra_c_deg, dec_c_deg, (double*)records);
    else
        hipLaunchKernelGGL(prepare_records_kernel<float>, dim3(grid), dim3(kBlock), 0, s, raw, n, model,
                           (int)free_centre, ra_c_deg, dec_c_deg, (float*)records);
    return hipGetLastError();
}

hipError_t launch_prepare_walkers(hipStream_t s, const double* params, int64_t n_rows, int k, int model,
                                  bool free_centre, int precision, void* wpar) {
    if (n_rows <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((n_rows + kBlock - 1) / kBlock);
    if (precision == 0)
        hipLaunchKernelGGL(prepare_walkers_kernel<double>, dim3(grid), dim3(kBlock), 0, s, params, n_rows, k, model,
                           (int)free_centre, (double*)wpar);
    else
        hipLaunchKernelGGL(prepare_walkers_kernel<float>, dim3(grid), dim3(kBlock), 0, s, params, n_rows, k, model,
                           (int)free_centre, (float*)wpar);
    return hipGetLastError();
}

hipError_t launch_loglike(hipStream_t s, const LaunchShape& sh, const void* records, const Chunk* chunks,
                          int64_t n_chunks, const void* wpar, double* partials, int64_t n_walkers) {
#define MCD_DISPATCH(M)                                                                                          \
    case M:                                                                                                      \
        return sh.free_centre ? launch_precision<M, true>(s, sh, records, chunks, n_chunks, wpar, partials, n_walkers) \
                              : launch_precision<M, false>(s, sh, records, chunks, n_chunks, wpar, partials, n_walkers);
    switch (sh.model) {
        MCD_DISPATCH(MODEL_CONST)
        MCD_DISPATCH(MODEL_BGFIXED)
        MCD_DISPATCH(MODEL_BGGAUSS)
        MCD_DISPATCH(MODEL_PROFILE)
        MCD_DISPATCH(MODEL_PROFILE_BGGAUSS)
        MCD_DISPATCH(MODEL_PROFILE_BGDENS)
        MCD_DISPATCH(MODEL_PROFILE_BGFIXED)
    }
#undef MCD_DISPATCH
    return hipErrorInvalidValue;
}

hipError_t launch_reduce(hipStream_t s, const double* partials, const int64_t* offs, int64_t n_psets,
                         int64_t n_chunks, int64_t max_chunks_per_pset, int64_t n_walkers, const double* pset_const,
                         double* out) {
    if (n_psets * n_walkers <= 0) return hipSuccess;
    const int64_t n_groups = (n_walkers + kPartialGroup - 1) / kPartialGroup;
    const dim3 grid((unsigned)(n_psets * n_groups));
    // sublanes per walker group by the longest parameter set: 128 (1024 threads) streams ~28 chunks per thread at
    // C3's 3551 chunks; radial bins with a few chunks each take one wave per group
    if (max_chunks_per_pset > 512)
        hipLaunchKernelGGL(reduce_group_kernel<128>, grid, dim3(kPartialGroup * 128), 0, s, partials, offs, n_chunks,
                           n_walkers, n_groups, pset_const, out);
    else if (max_chunks_per_pset > 32)
        hipLaunchKernelGGL(reduce_group_kernel<32>, grid, dim3(kPartialGroup * 32), 0, s, partials, offs, n_chunks,
                           n_walkers, n_groups, pset_const, out);
    else
        hipLaunchKernelGGL(reduce_group_kernel<8>, grid, dim3(kPartialGroup * 8), 0, s, partials, offs, n_chunks,
                           n_walkers, n_groups, pset_const, out);
    return hipGetLastError();
}

namespace {
template <int MODEL>
hipError_t per_star_model(hipStream_t s, const LaunchShape& sh, const void* records, int64_t n, const void* wpar_row,
                          int mode, double* out) {
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
    if (sh.precision == 0) {
        if (sh.free_centre)
            hipLaunchKernelGGL((per_star_kernel<MODEL, true, double>), dim3(grid), dim3(kBlock), 0, s,
                               (const double*)records, n, (const double*)wpar_row, mode, out);
        else
            hipLaunchKernelGGL((per_star_kernel<MODEL, false, double>), dim3(grid), dim3(kBlock), 0, s,
                               (const double*)records, n, (const double*)wpar_row, mode, out);
    } else {
        if (sh.free_centre)
            hipLaunchKernelGGL((per_star_kernel<MODEL, true, float>), dim3(grid), dim3(kBlock), 0, s,
                               (const float*)records, n, (const float*)wpar_row, mode, out);
        else
            hipLaunchKernelGGL((per_star_kernel<MODEL, false, float>), dim3(grid), dim3(kBlock), 0, s,
                               (const float*)records, n, (const float*)wpar_row, mode, out);
    }
    return hipGetLastError();
}
}  // namespace

hipError_t launch_per_star(hipStream_t s, const LaunchShape& sh, const void* records, int64_t n,
                           const void* wpar_row, int mode, double* out) {
    if (n <= 0) return hipSuccess;
    switch (sh.model) {
        case MODEL_BGFIXED: return per_star_model<MODEL_BGFIXED>(s, sh, records, n, wpar_row, mode, out);
        case MODEL_BGGAUSS: return per_star_model<MODEL_BGGAUSS>(s, sh, records, n, wpar_row, mode, out);
        case MODEL_PROFILE_BGGAUSS: return per_star_model<MODEL_PROFILE_BGGAUSS>(s, sh, records, n, wpar_row, mode, out);
        case MODEL_PROFILE_BGDENS: return per_star_model<MODEL_PROFILE_BGDENS>(s, sh, records, n, wpar_row, mode, out);
        case MODEL_PROFILE_BGFIXED: return per_star_model<MODEL_PROFILE_BGFIXED>(s, sh, records, n, wpar_row, mode, out);
    }
    return hipErrorInvalidValue;
}

}  // namespace mcd
