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
#include "mcd_reduce.h"

namespace mcd {

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;               // 4 waves: one per SIMD of a CU
constexpr int kWavesPerBlock = kBlock / kWave;

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

#ifdef MCD_MAIN_STAMPS
// development aid (make variant NAME=stamps DEFS=-DMCD_MAIN_STAMPS, tools/main_stamps_probe.py): per workgroup of the
// last main-kernel launch, s_memrealtime (100 MHz) at entry and before the partial-sum store, and where it ran
constexpr int kStampBlocks = 1 << 16;
__device__ unsigned long long g_main_stamps[kStampBlocks][4];
#endif

__device__ const double kExpTabDevice[kExpTabSize] = {MCD_EXP_TABLE_VALUES};
__device__ const double kExpTabSqrt2Device[kExpTabSize] = {MCD_EXP_TABLE_SQRT2_VALUES};

// ------------------------------------------------------------------------------------------------
// main kernel (the per-chunk arithmetic lives in mcd_math.h: chunk_loglike)
// WAVES = 4: one partial sum per (chunk, walker).  WAVES = 8 or 16 (balanced single-round plans of small catalogues, f64
// fast kernels): the workgroup's waves are WAVES / n_wtiles consecutive chunks x n_wtiles walker tiles (n_wtiles 1, 2 or
// 4); the chunks' sums are added in chunk order through LDS and ONE partial sum per (workgroup, walker) leaves -- 256
// partial sums per walker for a launch of one workgroup per CU, few enough for the step kernel of the resident chain to
// add up itself (mcd_stretch.hip) and for a one-wave-per-group reduction.  The second launch bound (minimum waves per
// SIMD) keeps the register budget of the 4-wave kernel: left alone, the compiler spends up to 150 VGPRs on the 8-wave
// BGGAUSS kernels (106 - 127 with 4 waves) and the occupancy drops from 4 to 3 waves per SIMD.
template <int MODEL, bool FREE, class T, class A, int FAST, bool PF, int WAVES>
__global__ __launch_bounds__(WAVES * kWave, WAVES > 4 ? 4 : 1) void loglike_kernel(const T* __restrict__ recs,
                                                                 const Chunk* __restrict__ chunks,
                                                                 const T* __restrict__ wpar,
                                                                 double* __restrict__ partials, int64_t n_tasks,
                                                                 int n_wtiles, int64_t n_walkers, int64_t n_chunks,
                                                                 int uniform_len, int uniform_extra, int64_t n_records,
                                                                 double* __restrict__ rerun_flag, double launch_tag,
                                                                 const uint8_t* __restrict__ chunk_general,
                                                                 int64_t n_slots) {
    constexpr int ND = record_doubles(MODEL, FREE);
    constexpr int kThreads = WAVES * kWave;
    constexpr bool kCombine = WAVES > kWavesPerBlock;
#ifdef MCD_MAIN_STAMPS
    const unsigned long long stamp_entry = __builtin_amdgcn_s_memrealtime();      // (kept in SGPRs: no store before the loop)
#endif
    // fast mixtures: the 2^(j/1024) table of exp_tab lives in LDS (8 KiB per workgroup), four entries copied per thread
    constexpr bool kUsesExpTab = FAST && bg_kind(MODEL) != BG_NONE && sizeof(T) == 8;
    __shared__ double exptab_lds[kUsesExpTab ? kExpTabSize : 1];
    __shared__ double combine_lds[kCombine ? WAVES : 1][kCombine ? kWave : 1];
    if constexpr (kUsesExpTab) {
        static_assert(kExpTabSize % kThreads == 0, "whole table entries per thread");
        const double* __restrict__ src = exp_table_is_sqrt2_scaled(MODEL) ? kExpTabSqrt2Device : kExpTabDevice;
#pragma unroll
        for (int i = 0; i < kExpTabSize / kThreads; ++i) exptab_lds[i * kThreads + threadIdx.x] = src[i * kThreads + threadIdx.x];
        __syncthreads();
    }
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & (kWave - 1);
    int64_t chunk_id;
    int wtile;
    bool live = true;                                                        // (combining workgroups keep idle waves for the barrier)
    if (kCombine || n_wtiles <= kWavesPerBlock) {
        // <= 256 walkers: consecutive waves share a chunk, so every chunk is read by one workgroup (one CU, one XCD)
        const int64_t task = (int64_t)blockIdx.x * WAVES + wave;            // wave-uniform
        if (task >= n_tasks) {
            if constexpr (!kCombine) return;
            live = false;
        }
        chunk_id = task / n_wtiles;
        wtile = (int)(task - chunk_id * n_wtiles);
        if (!live) { chunk_id = 0; wtile = wave % n_wtiles; }
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
    // Arithmetic chunk tables of a single parameter set (equal lengths, or the balanced plans' len / len + 8:
    // mcd_chunks.h: uniform_chunk): the first record load then does not wait behind a descriptor load (three dependent
    // memory latencies at wave start become one).
    Chunk ch;
    if (uniform_len > 0) {
        const int64_t shifted = chunk_id < (int64_t)uniform_extra ? chunk_id : (int64_t)uniform_extra;
        ch.begin = chunk_id * uniform_len + 8 * shifted;
        const int64_t left = n_records - ch.begin;
        ch.count = chunk_id == n_chunks - 1 ? (int32_t)left : uniform_len + (chunk_id < (int64_t)uniform_extra ? 8 : 0);
        ch.pset = 0;
    } else {
        ch = chunks[chunk_id];                                               // scalar load
    }
    const int64_t w_raw = (int64_t)wtile * kWave + lane;
    const bool active = w_raw < n_walkers;
    const int64_t w_idx = active ? w_raw : n_walkers - 1;                   // idle lanes shadow the last walker

    double result = 0.0;
    if (live) {
        const T* __restrict__ wp = wpar + ((int64_t)ch.pset * n_walkers + w_idx) * KD;
        WalkerConsts<T> w;
        w.load(wp);                                   // unused constants are dead code for a given MODEL

        // wave-uniform record pointer in the constant address space: the reads inside chunk_loglike are scalar loads
        const RecPtr<T> chunk_recs = (RecPtr<T>)(recs + ch.begin * ND);
        bool denormal;
        if constexpr (FAST == 2) {
            // a chunk that holds a star outside the narrow-range domain (certain member, extreme background, ...) takes the
            // general fast form; the flag is wave-uniform (one scalar byte load), so this is a scalar branch
            // (the narrow-range profile variant without background has no per-star conditions: no flags, one form)
            const bool general = bg_kind(MODEL) != BG_NONE && chunk_general != nullptr && chunk_general[chunk_id] != 0;
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
    }
    // partials[walker group of 8][slot][walker in group]: eight full 64-byte segments per wave store (the rows are
    // padded to whole walker tiles, so idle lanes store their shadow value into padding), and the reduce kernel streams
    // one contiguous [slot][8] block per walker group.  slot = chunk, or the workgroup when its chunks are combined.
    if constexpr (kCombine) {
        combine_lds[wave][lane] = result;
        __syncthreads();
        if (wave < n_wtiles) {                                    // (wave == wtile for the workgroup's first chunk)
            double sum = combine_lds[wave][lane];
            for (int c = n_wtiles; c < WAVES; c += n_wtiles) sum += combine_lds[c + wave][lane];     // chunk order
            partials[((w_raw >> 3) * n_slots + blockIdx.x) * kPartialGroup + (w_raw & (kPartialGroup - 1))] = sum;
        }
    } else {
        partials[((w_raw >> 3) * n_slots + chunk_id) * kPartialGroup + (w_raw & (kPartialGroup - 1))] = result;
    }
#ifdef MCD_MAIN_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < kStampBlocks) {
        g_main_stamps[blockIdx.x][0] = stamp_entry;
        g_main_stamps[blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
        g_main_stamps[blockIdx.x][2] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                       (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4);     // XCC_ID | HW_ID
        g_main_stamps[blockIdx.x][3] = (unsigned long long)ch.count;
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// final reduction: one block per (parameter set, group of 8 walkers).  The group's partial sums are one contiguous
// [slot][8] array of 64-byte segments.  Thread t = (sublane s = t / 4, quarter q = t % 4) reads the 16 bytes of walkers
// 2q, 2q + 1 of the slots c0 + s, c0 + s + 2 SUB, ...: a wave's load covers sixteen whole segments (1 KiB, contiguous),
// and U independent accumulator pairs keep that many loads in flight per thread -- the kernel is bound by memory latency
// (the sums were written by workgroups on other XCDs: every round of loads is a trip to memory), so the shapes below are
// chosen to need ONE round: U x 2 SUB >= slots (launch_reduce).  The sublanes are combined by a wave shuffle tree and a
// fixed-order sum over the waves: the order of every addition is fixed by (slot range, SUB, U) alone -- bitwise
// repeatable, and the resident chain's step kernel (mcd_stretch.hip) gets the same bits from the same function.
//
// single0 >= 0: ONE parameter set whose slots are [0, single0) -- the range arrives as a kernel argument instead of two
// dependent loads from `offs` in front of the first load of partial sums (one memory round trip less)
template <int SUB, int U, bool ONE_ROUND>
__global__ __launch_bounds__(kPartialGroup * SUB) void reduce_group_kernel(const double* __restrict__ partials,
                                                                            const int64_t* __restrict__ offs,
                                                                            int64_t n_slots, int64_t n_walkers,
                                                                            int64_t n_groups,
                                                                            const double* __restrict__ pset_const,
                                                                            double* __restrict__ out, int64_t single0) {
    constexpr int kThreads = kPartialGroup * SUB;
    constexpr int kWaves = kThreads / kWave;
    __shared__ double lds[kWaves > 1 ? kWaves : 1][kPartialGroup];
    const int64_t pset = blockIdx.x / n_groups, g = blockIdx.x - pset * n_groups;
    const int q = threadIdx.x & 3, s = threadIdx.x >> 2;
    const int64_t c0 = single0 >= 0 ? 0 : offs[pset], c1 = single0 >= 0 ? single0 : offs[pset + 1];
    const double add = pset_const ? pset_const[pset] : 0.0;          // walker-independent part (sum of lnL_bg), in flight early
    const double2* __restrict__ col = reinterpret_cast<const double2*>(partials + g * n_slots * kPartialGroup) + q;
    double ax, ay;
    reduce_sublane_sum<SUB, U, ONE_ROUND>(col, c0, c1, s, ax, ay);
    reduce_wave_combine(ax, ay);
    const int lane = threadIdx.x & (kWave - 1);
    if constexpr (kWaves > 1) {
        if (lane < 4) { lds[threadIdx.x >> 6][2 * q] = ax; lds[threadIdx.x >> 6][2 * q + 1] = ay; }
        __syncthreads();
        if (threadIdx.x < kPartialGroup) {
            double acc = lds[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) acc += lds[w][threadIdx.x];
            const int64_t w = g * kPartialGroup + threadIdx.x;
            if (w < n_walkers) out[pset * n_walkers + w] = acc + add;
        }
    } else {
        // one wave: lanes 0..3 hold the sums of walkers (2q, 2q + 1)
        if (lane < 4) {
            const int64_t w = g * kPartialGroup + 2 * q;
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
hipError_t launch_one(hipStream_t s, const LaunchShape& sh, const void* records, const Chunk* chunks, int64_t n_chunks,
                      const void* wpar, double* partials, int64_t n_walkers) {
    const int n_wtiles = (int)((n_walkers + kWave - 1) / kWave);
    const int64_t n_tasks = n_chunks * n_wtiles;
    // the prefetching instantiation exists for the fast formulations only (the plain kernels have no prefetch code)
    constexpr bool kCanCombine = FAST != 0 && sizeof(T) == 8 && sizeof(A) == 8;
    const bool combine = kCanCombine && sh.waves > kWavesPerBlock;
    const int64_t grid = combine ? (n_tasks + sh.waves - 1) / sh.waves
                                 : main_grid(n_chunks, n_walkers);   // > 256 walkers: XCD-aware grouping, see loglike_kernel
    if (grid <= 0) return hipSuccess;
    const int64_t n_slots = combine ? grid : n_chunks;
#define MCD_LAUNCH_MAIN(PF_, WAVES_)                                                                                         \
    hipLaunchKernelGGL((loglike_kernel<MODEL, FREE, T, A, FAST, PF_, WAVES_>), dim3((unsigned)grid), dim3(WAVES_ * kWave), 0, \
                       s, (const T*)records, chunks, (const T*)wpar, partials, n_tasks, n_wtiles, n_walkers, n_chunks,       \
                       sh.uniform_len, sh.uniform_extra, sh.n_records, sh.rerun_flag, sh.launch_tag, sh.chunk_general, n_slots)
    if constexpr (kCanCombine) {
        if (combine && sh.waves == 8) {
            if (sh.prefetch) MCD_LAUNCH_MAIN(true, 8);
            else MCD_LAUNCH_MAIN(false, 8);
            return hipGetLastError();
        }
        // 16 waves: 1024 threads, at most 128 VGPRs -- the kernels of the per-walker Gaussian background need more
        if constexpr (bg_kind(MODEL) != BG_GAUSS) {
            if (combine && sh.waves == 16) {
                if (sh.prefetch) MCD_LAUNCH_MAIN(true, 16);
                else MCD_LAUNCH_MAIN(false, 16);
                return hipGetLastError();
            }
        }
        if (combine) return hipErrorInvalidValue;
    }
    if (FAST != 0 && sh.prefetch) MCD_LAUNCH_MAIN((FAST != 0), kWavesPerBlock);
    else MCD_LAUNCH_MAIN(false, kWavesPerBlock);
#undef MCD_LAUNCH_MAIN
    return hipGetLastError();
}

template <int MODEL, bool FREE>
hipError_t launch_precision(hipStream_t s, const LaunchShape& sh, const void* records, const Chunk* chunks,
                            int64_t n_chunks, const void* wpar, double* partials, int64_t n_walkers) {
    switch (sh.precision) {
        case 0:
            if constexpr (bg_kind(MODEL) != BG_NONE || (MODEL == MODEL_PROFILE && !FREE)) {
                if (sh.fast == 2)
                    return launch_one<MODEL, FREE, double, double, 2>(s, sh, records, chunks, n_chunks, wpar, partials, n_walkers);
            }
            if (sh.fast) return launch_one<MODEL, FREE, double, double, 1>(s, sh, records, chunks, n_chunks, wpar, partials, n_walkers);
            return launch_one<MODEL, FREE, double, double, 0>(s, sh, records, chunks, n_chunks, wpar, partials, n_walkers);
        case 1:      // float32 terms and sums: fast formulations for every model when the f32 guard admits them (mcd_guard.h)
            if (sh.fast) return launch_one<MODEL, FREE, float, float, 1>(s, sh, records, chunks, n_chunks, wpar, partials, n_walkers);
            return launch_one<MODEL, FREE, float, float, 0>(s, sh, records, chunks, n_chunks, wpar, partials, n_walkers);
        case 2:      // float32 terms, float64 accumulation
            if (sh.fast) return launch_one<MODEL, FREE, float, double, 1>(s, sh, records, chunks, n_chunks, wpar, partials, n_walkers);
            return launch_one<MODEL, FREE, float, double, 0>(s, sh, records, chunks, n_chunks, wpar, partials, n_walkers);
    }
    return hipErrorInvalidValue;
}

}  // namespace

#ifdef MCD_MAIN_STAMPS
extern "C" int mcd_debug_main_stamps(unsigned long long* out, long long n_blocks) {
    if (n_blocks > kStampBlocks) n_blocks = kStampBlocks;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_main_stamps), (size_t)n_blocks * 4 * sizeof(unsigned long long));
}
#endif

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

// Partial-sum slots per walker of a launch (what the reduction reads): the chunks, or the workgroups when they combine
// their chunks (LaunchShape::waves == 8)
int64_t partial_slots(const LaunchShape& sh, int64_t n_chunks, int64_t n_walkers) {
    if (sh.waves <= kWavesPerBlock || sh.fast == 0 || sh.precision != 0) return n_chunks;
    const int64_t n_wtiles = (n_walkers + kWave - 1) / kWave;
    return (n_chunks * n_wtiles + sh.waves - 1) / sh.waves;
}

hipError_t launch_reduce(hipStream_t s, const double* partials, const int64_t* offs, int64_t n_psets,
                         int64_t n_slots, int64_t max_slots_per_pset, int64_t n_walkers, const double* pset_const,
                         double* out) {
    if (n_psets * n_walkers <= 0) return hipSuccess;
    const int64_t n_groups = (n_walkers + kPartialGroup - 1) / kPartialGroup;
    const dim3 grid((unsigned)(n_psets * n_groups));
    const int64_t single0 = n_psets == 1 ? n_slots : -1;
    // Shapes by the longest parameter set, each needing one round of loads up to its limit (2 SUB sublanes x U loads):
    //   <= 256 slots: one wave per group, 16 loads per thread (radial bins with a few chunks each; the balanced plans of
    //   small catalogues; the SAME order as the step kernel's fused reduction, mcd_stretch.hip)
    //   <= 1024: 4 waves x 16;  <= 4096: 16 waves x 16 (C3: 3551 chunks);  beyond: 16 waves, several rounds
#define MCD_LAUNCH_REDUCE(SUB_, U_, ONE_)                                                                                   \
    hipLaunchKernelGGL((reduce_group_kernel<SUB_, U_, ONE_>), grid, dim3(kPartialGroup * SUB_), 0, s, partials, offs,       \
                       n_slots, n_walkers, n_groups, pset_const, out, single0)
    static_assert(kFusedReduceSlots == 2 * 8 * 16, "one round of the one-wave shape");
    if (max_slots_per_pset <= kFusedReduceSlots) MCD_LAUNCH_REDUCE(8, 16, true);
    else if (max_slots_per_pset <= 1024) MCD_LAUNCH_REDUCE(32, 16, true);
    else if (max_slots_per_pset <= 4096) MCD_LAUNCH_REDUCE(128, 16, true);
    else MCD_LAUNCH_REDUCE(128, 16, false);
#undef MCD_LAUNCH_REDUCE
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
