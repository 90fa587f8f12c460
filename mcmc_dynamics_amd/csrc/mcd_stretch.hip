// mcd_stretch.hip -- the stretch move of mcd_stretch.h with the ensemble resident on the device.
//
// The host-driven block (mcd_stretch.h) spends ~45 us per half step between two kernels: wait for the sums, accept,
// propose, range guard, upload, two launches.  With 1e6 stars x 128 walkers the kernel itself takes ~107 us, so a third
// of the device time is idle.  Here the positions, their log-probabilities and the block's random numbers live on the
// device, and ONE small kernel per half step does what the host loop does between two evaluations:
//
//     accept / reject the previous half step  ->  [end of a step: chain row]  ->  propose the next half step,
//     box prior, resolved parameter rows, derived walker constants, range guard of the new table
//
// so a whole block of steps is a chain of launches (step kernel, main kernel, reduction [, all-reduce]) that the host
// enqueues ahead and waits for once.  The arithmetic is the host loop's, operation for operation (no FMA contraction:
// -ffp-contract=off), the walker constants come from the same device function as the prep kernel's (mcd_prep.h) and the
// range guard is the same code (mcd_guard.h): the chain is bit-identical to the host-driven one.
//
// What the device cannot decide alone it only detects: a NaN sum, a re-run request of the fast mixture kernels, a table
// whose guard verdict differs from the kernel family the host enqueued, a half step with no proposal inside the prior.
// Each sets a bit in the status word; the host then discards the block and runs it through the host-driven loop from the
// same inputs (mcd_api.hip: mcd_stretch_move), which handles all of them.
#include "mcd_internal.h"   // first: <hip/hip_runtime.h> before the MCD_HD headers
#include "mcd_guard.h"
#include "mcd_math.h"
#include "mcd_prep.h"
#include "mcd_reduce.h"
#include "mcd_rng.h"

namespace mcd {

namespace {

constexpr int kStepBlock = 256;

#ifdef MCD_CHAIN_STAMPS
#define MCD_STAMP(k) do { if (threadIdx.x == 0 && d.stamps) d.stamps[d.launch_index * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MCD_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ double wave_lesser(double v) {
    for (int off = 32; off > 0; off >>= 1) v = ParamRanges::lesser(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ double wave_greater(double v) {
    for (int off = 32; off > 0; off >>= 1) v = ParamRanges::greater(v, __shfl_xor(v, off));
    return v;
}

__device__ __forceinline__ void wave_merge(ParamRanges& r) {
    r.s2_min = wave_lesser(r.s2_min); r.s2_max = wave_greater(r.s2_max); r.amp = wave_greater(r.amp);
    r.sb2_min = wave_lesser(r.sb2_min); r.sb2_max = wave_greater(r.sb2_max);
    r.f_min = wave_lesser(r.f_min); r.f_max = wave_greater(r.f_max);
    r.len_min = wave_lesser(r.len_min); r.len_max = wave_greater(r.len_max);
    r.finite = __all(r.finite ? 1 : 0) != 0;
}

// the quantities the model has (the others keep their initial values in every lane: nothing to exchange)
__device__ __forceinline__ void wave_merge_for(ParamRanges& r, int model) {
    r.s2_min = wave_lesser(r.s2_min); r.s2_max = wave_greater(r.s2_max); r.amp = wave_greater(r.amp);
    const int bg = bg_kind(model);
    if (bg == BG_GAUSS) { r.sb2_min = wave_lesser(r.sb2_min); r.sb2_max = wave_greater(r.sb2_max); }
    if (bg == BG_GAUSS || bg == BG_FIXED_DENSITY) { r.f_min = wave_lesser(r.f_min); r.f_max = wave_greater(r.f_max); }
    if (is_profile(model)) { r.len_min = wave_lesser(r.len_min); r.len_max = wave_greater(r.len_max); }
    r.finite = __all(r.finite ? 1 : 0) != 0;
}

__device__ __forceinline__ void store_ranges(const ParamRanges& r, double* o) {
    o[0] = r.s2_min; o[1] = r.s2_max; o[2] = r.amp; o[3] = r.sb2_min; o[4] = r.sb2_max; o[5] = r.f_min; o[6] = r.f_max;
    o[7] = r.len_min; o[8] = r.len_max; o[9] = r.finite ? 1.0 : 0.0;
}
__device__ __forceinline__ ParamRanges load_ranges(const double* o) {
    ParamRanges r;
    r.s2_min = o[0]; r.s2_max = o[1]; r.amp = o[2]; r.sb2_min = o[3]; r.sb2_max = o[4]; r.f_min = o[5]; r.f_max = o[6];
    r.len_min = o[7]; r.len_max = o[8]; r.finite = o[9] != 0.0;
    return r;
}

// One workgroup.  acc_step >= 0: accept / reject half step (acc_step, acc_h) from the sums `ll` (written by the reduction
// [+ all-reduce] that precedes this launch on the stream).  prop_step >= 0: propose half step (prop_step, prop_h).
__global__ __launch_bounds__(kStepBlock) void stretch_step_kernel(StretchDevice d, int64_t acc_step, int acc_h,
                                                                   int64_t prop_step, int prop_h,
                                                                   const double* __restrict__ ll, double rerun_tag) {
    const int64_t W = d.n_walkers, half = W / 2;
    const int P = d.n_dim, K = d.k;
    const int tid = threadIdx.x;
    __shared__ int s_n_ok, s_donor;
    __shared__ double s_ranges[kStepBlock / 64][10];          // ParamRanges of each wave (a type with initialisers cannot be __shared__)

    if (acc_step >= 0) {
        const int32_t* first = d.order + acc_step * W + (acc_h == 0 ? 0 : half);
        const double* t = d.thr + (acc_step * 2 + acc_h) * half;
        const int n_ok = d.meta[META_N_OK];
        // single device: the fast mixture kernels write the launch's tag behind the sums when they want the plain kernels
        if (tid == 0 && rerun_tag != 0.0 && ll[half] == rerun_tag) atomicOr(&d.meta[META_STATUS], CHAIN_RERUN);
        for (int64_t j = tid; j < half; j += kStepBlock) {
            const double new_lnp = (n_ok > 0 && d.ok[j]) ? ll[j] : -kInfinity;
            if (new_lnp != new_lnp) atomicOr(&d.meta[META_STATUS], CHAIN_NAN);   // genuine, or the multi-rank re-run signal
            const int64_t w = first[j];
            if (t[j] < new_lnp - d.lnp[w]) {                                     // accept iff thr < new_lnp - old_lnp
                const double* p = d.proposal + j * P;
                for (int c = 0; c < P; ++c) d.pos[w * P + c] = p[c];
                d.lnp[w] = new_lnp;
                d.accepted[w] += 1;
            }
        }
        __syncthreads();
        if (acc_h == 1) {
            if (d.chain) for (int64_t x = tid; x < W * P; x += kStepBlock) d.chain[acc_step * W * P + x] = d.pos[x];
            if (d.lnprob_chain) for (int64_t w = tid; w < W; w += kStepBlock) d.lnprob_chain[acc_step * W + w] = d.lnp[w];
        }
    }
    if (prop_step < 0) return;

    const int32_t* first = d.order + prop_step * W + (prop_h == 0 ? 0 : half);
    const int32_t* second = d.order + prop_step * W + (prop_h == 0 ? half : 0);
    const double* z = d.zz + (prop_step * 2 + prop_h) * half;
    const int32_t* pk = d.pick + (prop_step * 2 + prop_h) * half;
    if (tid == 0) { s_n_ok = 0; s_donor = 0x7fffffff; }
    __syncthreads();
    // proposal = partner - (partner - s) * z, inside the inclusive box prior (false for NaN)
    for (int64_t j = tid; j < half; j += kStepBlock) {
        const double* s = d.pos + (int64_t)first[j] * P;
        const double* q = d.pos + (int64_t)second[pk[j]] * P;
        double* p = d.proposal + j * P;
        bool good = d.fixed_ok != 0;
        for (int c = 0; c < P; ++c) {
            const double v = q[c] - (q[c] - s[c]) * z[j];
            p[c] = v;
            good = good && (v >= d.lo[c]) && (v <= d.hi[c]);
        }
        d.ok[j] = good;
        if (good) { atomicAdd(&s_n_ok, 1); atomicMin(&s_donor, (int)j); }
    }
    __syncthreads();
    const int n_ok = s_n_ok, donor = s_donor;
    ParamRanges mine;
    if (n_ok > 0) {
        // rows outside the prior take the first valid row's place in the launch and are masked when the sums come back
        for (int64_t j = tid; j < half; j += kStepBlock) {
            const double* p = d.proposal + (d.ok[j] ? j : (int64_t)donor) * P;
            double* row = d.table + j * K;
            for (int c = 0; c < K; ++c) {
                const int src = d.col_source[c];
                row[c] = src < 0 ? d.col_const[c] : (d.col_factor[c] == 1.0 ? p[src] : p[src] * d.col_factor[c]);
            }
            walker_constants<double>(row, d.model, d.free_centre != 0, d.wpar + j * KD);
            mine.add_row(row, K, d.model, d.free_centre != 0);
        }
    }
    wave_merge(mine);
    if ((tid & 63) == 0) store_ranges(mine, s_ranges[tid >> 6]);
    __syncthreads();
    if (tid == 0) {
        d.meta[META_N_OK] = n_ok;
        if (n_ok == 0) {
            atomicOr(&d.meta[META_STATUS], CHAIN_NO_PROPOSAL);
        } else {
            ParamRanges all = load_ranges(s_ranges[0]);
            for (int i = 1; i < kStepBlock / 64; ++i) all.merge(load_ranges(s_ranges[i]));
            int level = 0;
            if (d.allow_fast) {
                level = level_verdict(d.stats, d.model, false, half, all);
                if (d.allow_fast == 2 && level > 1) level = 1;
            }
            if (level != d.expected_level) {
                if (!(d.meta[META_STATUS] & CHAIN_LEVEL)) d.meta[META_LEVEL] = level;   // the first verdict that differed
                atomicOr(&d.meta[META_STATUS], CHAIN_LEVEL);
            }
        }
    }
}

// The same step for ensembles whose positions fit into LDS (every shipped configuration: W x P x 8 bytes <= 32 KiB,
// W <= 2 x kStepBlock, <= 12 columns): one proposal per thread, held in registers.  The general kernel above is written
// like the host loop and the compiler has to keep its loads in program order (stores to pos / lnp may alias them): ~16
// dependent trips to memory, and the memory is far away -- the sums were written by workgroups on other XCDs.  This one
// is built around what a single-workgroup kernel between two 105 us main kernels pays for (measured by cutting it short
// phase by phase, 1e6 stars x 128 walkers per half step):
//   * memory round trips: ONE round of loads brings in everything (positions, log-probabilities, the sums, the previous
//     proposals, the random numbers of both half steps); the indirections (lnp[first], second[pick], the pair's
//     positions) are LDS reads; results leave in one round of stores;
//   * instruction fetch: the code runs once per launch from a cold instruction cache and costs ~1 us per KiB of
//     straight-line code it executes (a version with every column loop unrolled to 12 spent 8 of its 12 us there, for 4
//     columns in use; rolled loops over LDS rows are worse still: every iteration waits out an LDS round trip).  Hence
//     MAXC, the unroll bound of the column loops, is a template parameter (4, 8 or 12, the smallest that holds n_dim and
//     k), and the guard's range reduction skips the quantities the model does not have.
// Same operations on the same numbers as the general kernel and the host loop, same results.
constexpr int kSmallPosBytes = 32 << 10;

extern __shared__ double s_dynamic[];     // [W * P] positions | [W] log-probabilities | [W / 2] int32: the partner half

// kFused: the kernel runs with kFusedThreads threads; all of its waves first add up the main kernel's partial sums of this
// ensemble (one wave per group of 8 walkers at a time, the reduction kernel's own code and order: mcd_reduce.h) into LDS,
// then the waves beyond the first kStepBlock threads leave and the step proceeds as below with the sums read from LDS.
constexpr int kFusedThreads = 1024;

template <int kMaxCols, bool kBinned, bool kFused>
__global__ __launch_bounds__(kFused ? kFusedThreads : kStepBlock) void stretch_step_small_kernel(
    StretchDevice d, int64_t acc_step, int acc_h, int64_t prop_step, int prop_h, const double* __restrict__ ll,
    double rerun_tag) {
    const int half = (int)(d.n_walkers / 2), W = 2 * half, P = d.n_dim, K = d.k;
    const int j = threadIdx.x;
    __shared__ double s_ll[kFused ? kStepBlock : 1];
    if constexpr (kFused) {
        if (acc_step >= 0) {
            const int64_t bb = kBinned ? blockIdx.x : 0;
            const int64_t c0 = d.slot_offsets ? d.slot_offsets[bb] : 0, c1 = d.slot_offsets ? d.slot_offsets[bb + 1] : d.n_slots;
            const double add = d.pset_const ? d.pset_const[bb] : 0.0;
            const int q = j & 3, s = (j & 63) >> 2, n_groups = (half + kPartialGroup - 1) / kPartialGroup;
            for (int g = j >> 6; g < n_groups; g += kFusedThreads / 64) {
                const double2* __restrict__ col = reinterpret_cast<const double2*>(d.partials + (int64_t)g * d.n_slots * kPartialGroup) + q;
                double ax, ay;
                reduce_sublane_sum<8, 16, true>(col, c0, c1, s, ax, ay);      // (<= kFusedReduceSlots = 8 x 2 x 16 slots)
                reduce_wave_combine(ax, ay);
                if ((j & 63) < 4) {
                    const int w = g * kPartialGroup + 2 * q;
                    if (w < half) s_ll[w] = ax + add;
                    if (w + 1 < half) s_ll[w + 1] = ay + add;
                }
            }
        }
        if (j >= kStepBlock) {
            __syncthreads();          // hand the sums over, then leave (a finished wave no longer counts at later barriers)
            return;
        }
    }
    // one workgroup per ensemble (radial bin): arrays carry the bin index in front of the walker index
    const int64_t B = kBinned ? d.n_bins : 1, b = kBinned ? blockIdx.x : 0;     // (one ensemble: no index arithmetic left)
    double* const pos_b = d.pos + b * W * P;
    double* const lnp_b = d.lnp + b * W;
    double* const proposal_b = d.proposal + b * half * P;
    uint8_t* const ok_b = d.ok + b * half;
    const double* const ll_b = ll + b * half;
    // n_ok and the guard's ranges are kept per half-step parity: workgroup 0 judges the PREVIOUS table of all ensembles
    // while the other workgroups already write the next one's
    const int64_t slot_acc = ((acc_step * 2 + acc_h) & 1) * B, slot_prop = ((prop_step * 2 + prop_h) & 1) * B;
    const bool active = j < half, do_acc = acc_step >= 0, do_prop = prop_step >= 0;
    __shared__ int s_wave_ok[kStepBlock / 64], s_wave_first[kStepBlock / 64];
    __shared__ double s_ranges[kStepBlock / 64][10];
    __shared__ double s_donor_prop[kMaxCols];
    __shared__ double s_lo[kMaxCols], s_hi[kMaxCols], s_const[kMaxCols], s_factor[kMaxCols];
    __shared__ int s_source[kMaxCols];
    __shared__ double s_rows[kStepBlock][kMaxCols + 1];        // (+1: rows of different threads start in different banks)
    double* const s_pos = s_dynamic;
    double* const s_lnp = s_dynamic + W * P;
    int* const s_second = reinterpret_cast<int*>(s_lnp + W);

    MCD_STAMP(0);
    // ---- the one round of loads: nothing here depends on this kernel's own stores ----
    // (all loads of the copy first, then the LDS stores: a plain copy loop of unknown trip count is compiled to one
    // round trip per iteration, 3.5 us for the four iterations of 256 walkers x 4 columns)
    {
        constexpr int kPairs = kSmallPosBytes / 16 / kStepBlock;          // double2 loads per thread that cover the largest ensemble
        const double2* __restrict__ src = reinterpret_cast<const double2*>(pos_b);
        const int n_pairs = W * P / 2;                                     // (W is even)
        double2 r[kPairs];
#pragma unroll
        for (int u = 0; u < kPairs; ++u) {
            const int x = j + u * kStepBlock;
            r[u] = x < n_pairs ? src[x] : make_double2(0.0, 0.0);
        }
        const double l0 = j < W ? lnp_b[j] : 0.0, l1 = j + kStepBlock < W ? lnp_b[j + kStepBlock] : 0.0;
#pragma unroll
        for (int u = 0; u < kPairs; ++u) {
            const int x = j + u * kStepBlock;
            if (x < n_pairs) { s_pos[2 * x] = r[u].x; s_pos[2 * x + 1] = r[u].y; }
        }
        if (j < W) s_lnp[j] = l0;
        if (j + kStepBlock < W) s_lnp[j + kStepBlock] = l1;
    }
    int n_ok_prev = 0, w_acc = 0;
    bool ok_prev = false;
    double ll_j = 0.0, thr_j = 0.0, flag_word = 0.0;
    double prev[kMaxCols];
    if (do_acc) {
        n_ok_prev = d.n_ok[slot_acc + b];
        if (b == 0 && j == 0 && rerun_tag != 0.0) flag_word = ll[B * half];
        if (active) {
            ok_prev = ok_b[j] != 0;
            if constexpr (!kFused) ll_j = ll_b[j];
            w_acc = d.order[(acc_step * B + b) * W + (acc_h == 0 ? 0 : half) + j];
            thr_j = d.thr[((acc_step * 2 + acc_h) * B + b) * half + j];
#pragma unroll
            for (int c = 0; c < kMaxCols; ++c) prev[c] = c < P ? proposal_b[j * P + c] : 0.0;
        }
    }
    // several ensembles: workgroup 0 judges the guard on the table of all of them (below); the first kStepBlock ensembles'
    // ranges are fetched with this round of loads, not behind the barrier
    int first_n_ok = 0;
    double first_ranges[10];
    if (kBinned && b == 0 && do_acc && j < B) {
        first_n_ok = d.n_ok[slot_acc + j];
#pragma unroll
        for (int i = 0; i < 10; ++i) first_ranges[i] = d.ranges[(slot_acc + j) * 10 + i];
    }
    int w_s = 0, pick_j = 0;
    double z_j = 0.0;
    if (do_prop && active) {
        w_s = d.order[(prop_step * B + b) * W + (prop_h == 0 ? 0 : half) + j];
        s_second[j] = d.order[(prop_step * B + b) * W + (prop_h == 0 ? half : 0) + j];
        pick_j = d.pick[((prop_step * 2 + prop_h) * B + b) * half + j];
        z_j = d.zz[((prop_step * 2 + prop_h) * B + b) * half + j];
    }
    if (do_prop && j < kMaxCols) {
        s_lo[j] = j < P ? d.lo[j] : 0.0;
        s_hi[j] = j < P ? d.hi[j] : 0.0;
        s_source[j] = j < K ? d.col_source[j] : -1;
        s_const[j] = j < K ? d.col_const[j] : 0.0;
        s_factor[j] = j < K ? d.col_factor[j] : 1.0;
    }
    __syncthreads();
    MCD_STAMP(1);
    if constexpr (kFused) { if (do_acc && active) ll_j = s_ll[j]; }
    // ---- accept / reject (walkers of one half step are distinct: no two threads touch the same row) ----
    if (do_acc) {
        if (b == 0 && j == 0 && rerun_tag != 0.0 && flag_word == rerun_tag) atomicOr(&d.meta[META_STATUS], CHAIN_RERUN);
        if (B > 1 && b == 0) {
            // several ensembles: was the kernel family of the launch whose sums arrive now the one the guard picks for
            // the table of all ensembles?  (each workgroup of the previous launch left its ensemble's ranges in memory;
            // ensembles without a valid proposal raised CHAIN_NO_PROPOSAL themselves)
            ParamRanges mine;
            if (j < B && first_n_ok > 0) mine.merge(load_ranges(first_ranges));
            for (int64_t e = j + kStepBlock; e < B; e += kStepBlock)
                if (d.n_ok[slot_acc + e] > 0) mine.merge(load_ranges(d.ranges + (slot_acc + e) * 10));
            wave_merge(mine);
            if ((j & 63) == 0) store_ranges(mine, s_ranges[j >> 6]);
            __syncthreads();
            if (j == 0) {
                ParamRanges all = load_ranges(s_ranges[0]);
                for (int i = 1; i < kStepBlock / 64; ++i) all.merge(load_ranges(s_ranges[i]));
                int level = 0;
                if (d.allow_fast) {
                    level = level_verdict(d.stats, d.model, false, B * half, all);
                    if (d.allow_fast == 2 && level > 1) level = 1;
                }
                if (level != d.expected_level && !(d.meta[META_STATUS] & CHAIN_NO_PROPOSAL)) {
                    if (!(d.meta[META_STATUS] & CHAIN_LEVEL)) d.meta[META_LEVEL] = level;
                    atomicOr(&d.meta[META_STATUS], CHAIN_LEVEL);
                }
            }
            __syncthreads();                           // (s_ranges is used again by the propose phase)
        }
        if (active) {
            const double new_lnp = (n_ok_prev > 0 && ok_prev) ? ll_j : -kInfinity;
            if (new_lnp != new_lnp) atomicOr(&d.meta[META_STATUS], CHAIN_NAN);
            if (thr_j < new_lnp - s_lnp[w_acc]) {
#pragma unroll
                for (int c = 0; c < kMaxCols; ++c)
                    if (c < P) { s_pos[w_acc * P + c] = prev[c]; pos_b[w_acc * P + c] = prev[c]; }
                s_lnp[w_acc] = new_lnp;
                lnp_b[w_acc] = new_lnp;
                atomicAdd((unsigned long long*)&d.accepted[b * W + w_acc], 1ull);
            }
        }
        __syncthreads();                              // the ensemble is final for this half step
        if (acc_h == 1) {
            if (d.chain) for (int x = j; x < W * P; x += kStepBlock) d.chain[(acc_step * B + b) * W * P + x] = s_pos[x];
            if (d.lnprob_chain) for (int w = j; w < W; w += kStepBlock) d.lnprob_chain[(acc_step * B + b) * W + w] = s_lnp[w];
        }
    }
    MCD_STAMP(2);
    if (!do_prop) return;
    // ---- propose ----
    bool good = false;
    double mine_prop[kMaxCols];
    if (active) {
        const int w_q = s_second[pick_j];
        good = d.fixed_ok != 0;
#pragma unroll
        for (int c = 0; c < kMaxCols; ++c) {
            if (c < P) {
                const double sc = s_pos[w_s * P + c], qc = s_pos[w_q * P + c];
                const double v = qc - (qc - sc) * z_j;
                mine_prop[c] = v;
                proposal_b[j * P + c] = v;
                good = good && (v >= s_lo[c]) && (v <= s_hi[c]);
            } else {
                mine_prop[c] = 0.0;
            }
        }
        ok_b[j] = good;
    }
    // how many proposals lie inside the prior, and the first of them: one ballot per wave (LDS atomics of 64 lanes on
    // one address are served lane by lane: 2.5 us for the two of them)
    {
        const unsigned long long inside = __ballot(good ? 1 : 0);
        if ((j & 63) == 0) {
            s_wave_ok[j >> 6] = __popcll(inside);
            s_wave_first[j >> 6] = inside ? j + __ffsll((long long)inside) - 1 : 0x7fffffff;
        }
    }
    MCD_STAMP(3);
    __syncthreads();
    int n_ok = 0, donor = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < kStepBlock / 64; ++i) {
        n_ok += s_wave_ok[i];
        donor = s_wave_first[i] < donor ? s_wave_first[i] : donor;
    }
    if (j == donor) {
#pragma unroll
        for (int c = 0; c < kMaxCols; ++c) s_donor_prop[c] = mine_prop[c];
    }
    __syncthreads();
    MCD_STAMP(4);
    // One ensemble with the guard deferred (d.defer_guard): the resolved rows go to this launch's place in the table log and
    // stretch_judge_kernel gives the verdicts of all launches at the end of the block -- the verdict only decides whether the
    // block is kept, so the range reduction, its barrier and the verdict (1.9 of 7 us) need not sit between two main kernels.
    const bool defer = !kBinned && d.defer_guard != 0;
    const int64_t launch = prop_step * 2 + prop_h;
    double* const table_rows = defer ? d.table_log + launch * half * K : d.table + b * half * K;
    ParamRanges mine;
    if (active && n_ok > 0) {
        // a row outside the prior takes the first valid row's place in the launch and is masked when the sums come back
        double* row = s_rows[j];
#pragma unroll
        for (int c = 0; c < kMaxCols; ++c) row[c] = good ? mine_prop[c] : s_donor_prop[c];          // (the proposal, for now)
        double resolved[kMaxCols];
#pragma unroll
        for (int c = 0; c < kMaxCols; ++c) {
            const int src = s_source[c];
            const double pv = row[src < 0 ? 0 : src];
            resolved[c] = src < 0 ? s_const[c] : (s_factor[c] == 1.0 ? pv : pv * s_factor[c]);
        }
#pragma unroll
        for (int c = 0; c < kMaxCols; ++c) {
            row[c] = resolved[c];
            if (c < K) table_rows[j * K + c] = resolved[c];
        }
        walker_constants<double>(row, d.model, d.free_centre != 0, d.wpar + (b * half + j) * KD);
        if (!defer) mine.add_row(row, K, d.model, d.free_centre != 0);
    }
    MCD_STAMP(5);
    if (defer) {
        if (j == 0) {
            d.n_ok[slot_prop + b] = n_ok;
            d.n_ok_log[launch] = n_ok;
            if (n_ok == 0) atomicOr(&d.meta[META_STATUS], CHAIN_NO_PROPOSAL);
        }
        MCD_STAMP(6);
        MCD_STAMP(7);
        return;
    }
    wave_merge_for(mine, d.model);
    if ((j & 63) == 0) store_ranges(mine, s_ranges[j >> 6]);
    __syncthreads();
    MCD_STAMP(6);
    if (j == 0) {
        d.n_ok[slot_prop + b] = n_ok;
        if (n_ok == 0) {
            // (with several ensembles the host loop would lend this one a valid row of another: it takes the block back)
            atomicOr(&d.meta[META_STATUS], CHAIN_NO_PROPOSAL);
        } else {
            ParamRanges all = load_ranges(s_ranges[0]);
            for (int i = 1; i < kStepBlock / 64; ++i) all.merge(load_ranges(s_ranges[i]));
            if (B > 1) {
                // the guard's verdict is on the table of ALL ensembles: this one's ranges go to memory and workgroup 0 of
                // the next launch judges them together (below, judge_all_ensembles)
                store_ranges(all, d.ranges + (slot_prop + b) * 10);
            } else {
                int level = 0;
                if (d.allow_fast) {
                    level = level_verdict(d.stats, d.model, false, half, all);
                    if (d.allow_fast == 2 && level > 1) level = 1;
                }
                if (level != d.expected_level) {
                    if (!(d.meta[META_STATUS] & CHAIN_LEVEL)) d.meta[META_LEVEL] = level;
                    atomicOr(&d.meta[META_STATUS], CHAIN_LEVEL);
                }
            }
        }
    }
    MCD_STAMP(7);
}

// The deferred guard of one-ensemble blocks: workgroup l judges the table of launch l (the rows the step kernel logged) with
// the step kernel's own operations -- ranges of the rows, merged (minima and maxima: any order gives the same bits), the
// verdict against the kernel family the block was enqueued with.  level_log[l] = the verdict, or -1 where the launch had no
// proposal inside the prior (no table); the host takes the first launch whose verdict differs as the next block's hint.
__global__ __launch_bounds__(kStepBlock) void stretch_judge_kernel(StretchDevice d) {
    __shared__ double s_rows[kStepBlock][13];
    __shared__ double s_ranges[kStepBlock / 64][10];
    const int64_t launch = blockIdx.x;
    const int j = threadIdx.x, half = (int)(d.n_walkers / 2), K = d.k;
    const int n_ok = d.n_ok_log[launch];                                 // (workgroup-uniform)
    if (n_ok <= 0) {
        if (j == 0) d.level_log[launch] = -1;
        return;
    }
    ParamRanges mine;
    if (j < half) {
        for (int c = 0; c < K; ++c) s_rows[j][c] = d.table_log[(launch * half + j) * K + c];
        mine.add_row(s_rows[j], K, d.model, d.free_centre != 0);
    }
    wave_merge_for(mine, d.model);
    if ((j & 63) == 0) store_ranges(mine, s_ranges[j >> 6]);
    __syncthreads();
    if (j == 0) {
        ParamRanges all = load_ranges(s_ranges[0]);
        for (int i = 1; i < kStepBlock / 64; ++i) all.merge(load_ranges(s_ranges[i]));
        int level = 0;
        if (d.allow_fast) {
            level = level_verdict(d.stats, d.model, false, half, all);
            if (d.allow_fast == 2 && level > 1) level = 1;
        }
        d.level_log[launch] = level;
        if (level != d.expected_level) atomicOr(&d.meta[META_STATUS], CHAIN_LEVEL);
    }
}

// ---- seeded blocks (mcd_stretch_move_seeded): the random numbers of steps [i0, i1) of a block, written where the host's
// upload would have put them (order [i][B][W], zz / thr / pick [i][2][B][W/2]) -- the step kernels do not know the difference.
// One workgroup per (ensemble, step): one generator call per walker (mcd_rng.h: chain_draw), the split of the ensemble by
// counting -- every thread ranks its walkers' keys against all keys in LDS, one unsigned comparison per pair (the keys are
// distinct: the walker index is their low bits).  A block of 256 steps x 55 ensembles x 512 walkers is 14 080 workgroups of
// ~10 us: ~0.1 ms next to 40 ms of likelihood kernels, on a stream of its own.
constexpr int kNumbersBlock = 256;
__global__ __launch_bounds__(kNumbersBlock) void chain_numbers_kernel(uint64_t seed, int64_t step0, int64_t i0, int64_t B,
                                                                      int64_t W, int n_dim, int32_t* __restrict__ order,
                                                                      double* __restrict__ zz, double* __restrict__ thr,
                                                                      int32_t* __restrict__ pick) {
    extern __shared__ unsigned long long s_order_keys[];                // [W]
    const int64_t b = blockIdx.x, i = i0 + blockIdx.y, half = W / 2;
    const int t = threadIdx.x;
    for (int64_t w = t; w < W; w += kNumbersBlock) {
        const int h = w >= half ? 1 : 0;
        const int64_t j = w - h * half;
        const ChainDraw cd = chain_draw(seed, step0 + i, h, b, j, half, n_dim);
        const int64_t at = ((i * 2 + h) * B + b) * half + j;
        zz[at] = cd.z;
        thr[at] = cd.thr;
        pick[at] = cd.pick;
        s_order_keys[w] = cd.order_key;
    }
    __syncthreads();
    const ulonglong2* pairs = reinterpret_cast<const ulonglong2*>(s_order_keys);       // (W is even)
    for (int64_t w = t; w < W; w += kNumbersBlock) {
        const unsigned long long mine = s_order_keys[w];
        int rank = 0;
        for (int64_t v = 0; v < half; ++v) {
            const ulonglong2 k = pairs[v];                               // every lane reads the same address: a broadcast
            rank += (k.x < mine ? 1 : 0) + (k.y < mine ? 1 : 0);
        }
        order[(i * B + b) * W + rank] = (int32_t)w;
    }
}

// status word -> a double every rank can sum (multi-rank: ranks hold different catalogue statistics, so their guard
// verdicts may differ; all of them must discard the block if one does)
__global__ void stretch_status_kernel(const int32_t* meta, double* out) { out[0] = meta[META_STATUS] != 0 ? 1.0 : 0.0; }

}  // namespace

namespace {
bool small_step(const StretchDevice& d) {
    const size_t pos_bytes = (size_t)d.n_walkers * d.n_dim * sizeof(double);
    const int cols = d.k > d.n_dim ? d.k : d.n_dim;
    return d.n_walkers <= 2 * kStepBlock && cols <= 12 && pos_bytes <= (size_t)kSmallPosBytes && !d.force_general;
}
}  // namespace

bool stretch_step_handles(const StretchDevice& d) { return d.n_bins == 1 || small_step(d); }
bool stretch_step_fuses(const StretchDevice& d) { return small_step(d); }

hipError_t launch_stretch_step(hipStream_t s, const StretchDevice& d, int64_t acc_step, int acc_h, int64_t prop_step,
                               int prop_h, const double* ll, double rerun_tag) {
    const size_t pos_bytes = (size_t)d.n_walkers * d.n_dim * sizeof(double);
    const size_t lds = pos_bytes + (size_t)d.n_walkers * sizeof(double) + (size_t)(d.n_walkers / 2) * sizeof(int32_t);
    const int cols = d.k > d.n_dim ? d.k : d.n_dim;
    const bool small = small_step(d);
    if (!small && d.n_bins != 1) return hipErrorInvalidValue;
#define MCD_LAUNCH_SMALL(C, BINNED)                                                                                            \
    do {                                                                                                                       \
        if (d.fused)                                                                                                           \
            hipLaunchKernelGGL((stretch_step_small_kernel<C, BINNED, true>), dim3((unsigned)d.n_bins), dim3(kFusedThreads),    \
                               lds, s, d, acc_step, acc_h, prop_step, prop_h, ll, rerun_tag);                                  \
        else                                                                                                                   \
            hipLaunchKernelGGL((stretch_step_small_kernel<C, BINNED, false>), dim3((unsigned)d.n_bins), dim3(kStepBlock), lds, \
                               s, d, acc_step, acc_h, prop_step, prop_h, ll, rerun_tag);                                       \
    } while (0)
    if ((d.fused || d.defer_guard) && (!small || (d.defer_guard && d.n_bins != 1))) return hipErrorInvalidValue;
    const bool binned = d.n_bins > 1;
    if (small && cols <= 4) { if (binned) MCD_LAUNCH_SMALL(4, true); else MCD_LAUNCH_SMALL(4, false); }
    else if (small && cols <= 8) { if (binned) MCD_LAUNCH_SMALL(8, true); else MCD_LAUNCH_SMALL(8, false); }
    else if (small) { if (binned) MCD_LAUNCH_SMALL(12, true); else MCD_LAUNCH_SMALL(12, false); }
#undef MCD_LAUNCH_SMALL
    else
        hipLaunchKernelGGL(stretch_step_kernel, dim3(1), dim3(kStepBlock), 0, s, d, acc_step, acc_h, prop_step, prop_h, ll, rerun_tag);
    return hipGetLastError();
}

bool chain_numbers_on_device(int64_t n_walkers) { return n_walkers * 8 <= (64 << 10) && n_walkers <= kSeededMaxWalkers; }

hipError_t launch_chain_numbers(hipStream_t s, uint64_t seed, int64_t step0, int64_t i0, int64_t i1, int64_t n_bins,
                                int64_t n_walkers, int n_dim, int32_t* order, double* zz, double* thr, int32_t* pick) {
    if (!chain_numbers_on_device(n_walkers) || n_bins < 1 || n_bins > 0x7fffffff) return hipErrorInvalidValue;
    for (int64_t at = i0; at < i1; at += 65535) {                        // (gridDim.y <= 65535)
        const int64_t n = i1 - at < 65535 ? i1 - at : 65535;
        hipLaunchKernelGGL(chain_numbers_kernel, dim3((unsigned)n_bins, (unsigned)n), dim3(kNumbersBlock),
                           (size_t)n_walkers * 8, s, seed, step0, at, n_bins, n_walkers, n_dim, order, zz, thr, pick);
    }
    return hipGetLastError();
}

hipError_t launch_stretch_judge(hipStream_t s, const StretchDevice& d, int64_t n_launches) {
    if (!d.defer_guard || !stretch_step_fuses(d) || d.n_bins != 1 || n_launches < 1 || n_launches > 0x7fffffff) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stretch_judge_kernel, dim3((unsigned)n_launches), dim3(kStepBlock), 0, s, d);
    return hipGetLastError();
}

hipError_t launch_stretch_status(hipStream_t s, const int32_t* meta, double* out) {
    hipLaunchKernelGGL(stretch_status_kernel, dim3(1), dim3(1), 0, s, meta, out);
    return hipGetLastError();
}

}  // namespace mcd
