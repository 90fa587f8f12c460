// mcd_chunks.h -- host-only planning of the work decomposition: star shards per device / rank and the chunk table of
// one shard.  No HIP types here, so that the C-ABI (mcd_api.hip) and the CPU test harness (tests/emul) share exactly
// the code that decides which stars a wave evaluates (as mcd_guard.h does for the range guard).
//
// Reference counterpart: none.  The reference evaluates all stars of one walker in one NumPy pass
// (analysis/runner.py:261-286); its only parallelism is a process pool over walkers (runner.py:398-403).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace mcd {

// One unit of work of the main kernel: a contiguous run of stars that all use parameter set `pset`.
// A wave evaluates one chunk for 64 walkers (lane = walker; star records arrive through the scalar
// cache as wave-uniform loads).
struct Chunk {
    int64_t begin;   // first star (record index on this device)
    int32_t count;   // stars in the chunk
    int32_t pset;    // parameter set (radial bin) the stars belong to
};

// Contiguous, balanced star range of shard `i` of `n_shards` (sizes differ by at most one); the same rule as
// distributed.shard_bounds on the Python side.
struct ShardRange { int64_t begin, n; };
inline ShardRange shard_range(int64_t n_stars, int i, int n_shards) {
    const int64_t b = n_stars * i / n_shards;
    return ShardRange{b, n_stars * (i + 1) / n_shards - b};
}

constexpr int64_t kMaxChunkLen = (int64_t)1 << 20;      // keeps per-chunk exponent sums far inside int32 (mcd_math.h: LogProduct)

struct ChunkPlan {
    std::vector<Chunk> chunks;          // ascending star order; a parameter set's chunks are consecutive
    std::vector<int64_t> offsets;       // [n_psets + 1] first chunk of each parameter set
    std::vector<uint8_t> general;       // per chunk: holds a narrow_exception star (empty when no chunk does)
    int64_t max_chunks_per_pset = 0;
    int64_t len = 0;                    // nominal chunk length
    int uniform_len = 0;                // > 0: the table is arithmetic (uniform_chunk below), parameter set 0 only
    int uniform_extra = 0;              // the first `uniform_extra` chunks hold uniform_len + 8 stars (balanced plans)
    int balanced_m = 0;                 // > 0: one round of equal waves, this many workgroups per CU (balanced_chunk_count)
};

// Chunk c of an arithmetic table: the first `extra` chunks hold len + 8 stars, the others len, the last one whatever is
// left (legacy equal-length tables: extra = 0 and a shorter last chunk; balanced tables: the n % 8 odd stars).  The main
// kernel computes the same three lines (mcd_kernels.hip) instead of waiting for a descriptor load.
inline Chunk uniform_chunk(int64_t c, int64_t len, int64_t extra, int64_t n, int64_t n_chunks) {
    Chunk ch;
    ch.begin = c * len + 8 * std::min(c, extra);
    ch.count = (int32_t)(c == n_chunks - 1 ? n - ch.begin : len + (c < extra ? 8 : 0));
    ch.pset = 0;
    return ch;
}

constexpr int64_t kNumCUs = 256;        // MI355X: 8 XCDs x 32 CUs, 4 SIMDs each

// One round of EQUAL waves for catalogues whose whole launch fits the resident wave slots (8 per SIMD): the number of
// chunks G such that every CU receives exactly `m` workgroups (the dispatcher deals a fresh grid evenly: measured with
// per-workgroup timestamps, tools/main_stamps_probe.py -- 1042 workgroups land as 4 or 5 per CU, and the CUs with 5 finish
// 25 % later).  Device timestamps at 1e5 stars x 256 walkers (CONST): a lone wave per SIMD needs 0.56 us per 16-star
// iteration, two waves 0.66 us each (0.33 us per iteration for the SIMD), four or five waves no less -- two waves per
// SIMD saturate the f64 pipe, more waves only add their ~100-instruction prologue and epilogue (the final log) per wave.
//   n_wtiles 1, 2, 4: a workgroup's four waves cover 4, 2, 1 chunks      -> G = 256 m (4 / n_wtiles)
//   n_wtiles 3:       workgroups straddle chunks                           -> G = ceil(1024 m / 3)
//   n_wtiles > 4:     a chunk needs mb = ceil(n_wtiles / 4) workgroups     -> G = 256 m / mb, a multiple of 8 (XCD grouping)
inline int64_t balanced_chunk_count(int64_t n_wtiles, int64_t m) {
    if (n_wtiles <= 0 || m <= 0) return 0;
    if (n_wtiles == 3) return (4 * kNumCUs * m + 2) / 3;
    if (n_wtiles <= 4) return kNumCUs * m * (4 / n_wtiles);
    const int64_t mb = (n_wtiles + 3) / 4;
    return std::max<int64_t>(8, kNumCUs * m / mb / 8 * 8);
}

// Chunk table of the shard [star_begin, star_begin + n) of a catalogue whose parameter sets (radial bins) are the
// global star ranges bin_offsets[p] .. bin_offsets[p + 1]; a bin that straddles the shard edge contributes its local
// part (the partial sums of the shards add up in the all-reduce).
//   * nominal length = the odd multiple of 32 nearest to n n_wtiles / target_waves for >= 256 walkers, to
//     1.25 n n_wtiles / 8000 below (one round of resident waves, guided tail included); at least 96
//   * guided schedule (tail_split): workgroups are dispatched in chunk order, so the end of a large parameter set is
//     cut into shorter chunks and the launch ends on short waves.  0 equal chunks; 1 = 85/10/5 % at len, len/2, len/4;
//     2 = 70/15/10/5 % down to len/8; 3, 4 = guided self-scheduling, chunk = remaining work / (G x resident waves)
//   * every chunk length is a multiple of 8 except the last chunk of a parameter set
//   * narrow_exceptions: ascending GLOBAL star indices (mcd_guard.h)
inline ChunkPlan plan_chunks(const std::vector<int64_t>& bin_offsets, int64_t star_begin, int64_t n, int64_t n_walkers,
                             int64_t target_waves, int tail_split, const std::vector<int64_t>& narrow_exceptions,
                             int64_t chunk_len = 0, int balance = 0) {
    ChunkPlan plan;
    const int64_t n_psets = (int64_t)bin_offsets.size() - 1;
    const int64_t n_wtiles = (n_walkers + 63) / 64;
    // ---- one round of equal waves: `balance` = workgroups per CU (0: the multi-round schedules below; which catalogues
    // take it is the caller's rule, mcd_api.hip: balance_auto_m) ----
    if (balance > 0 && n_psets == 1 && chunk_len == 0 && n > 0 && bin_offsets[0] <= star_begin &&
        bin_offsets[1] >= star_begin + n) {
        const int64_t m = std::min<int64_t>(balance, 8);
        const int64_t G = balanced_chunk_count(n_wtiles, m);
        const int64_t q = G > 0 ? n / G / 8 * 8 : 0;
        if (q >= 16 && q + 8 < kMaxChunkLen) {
            const int64_t extra = (n - G * q) / 8;
            plan.len = q;
            plan.uniform_len = (int)q;
            plan.uniform_extra = (int)extra;
            plan.balanced_m = (int)m;
            plan.chunks.reserve((size_t)G);
            for (int64_t c = 0; c < G; ++c) plan.chunks.push_back(uniform_chunk(c, q, extra, n, G));
            plan.offsets = {0, G};
            plan.max_chunks_per_pset = G;
        }
    }
    if (!plan.chunks.empty()) {
        // (flags below)
    } else {
    plan = ChunkPlan();
    // 256 walkers: `target_waves` full-length waves (default 10240 = 1.25 full-occupancy sets of 256 CUs x 4 SIMDs x 8
    // waves), i.e. ~1.5 rounds of resident waves once the guided tail is added; more than 256 walkers (several workgroups
    // per chunk): 1.2 x that (C5, 55 bins x 512 walkers: 158.6 us per step against 162.6).  <= 192 walkers (a workgroup's
    // four waves are spread over several chunks): ONE round -- 0.6 target_waves - 136 = 6008 waves plus the ~25 % extra
    // chunks of the guided tail; 1.1 - 1.5 rounds, where the last round runs nearly empty, cost 10 - 25 %
    // (tools/chunk_len_probe.py, tools/w128_len_sweep.py: 1e6 stars x 128 walkers 105.7 us per step at 416 stars per
    // chunk, 107.7 at 352, 110.2 at 288, 119.9 at 480; x 64 walkers 64.1 at 224, 66.6 at 160).  The default was 12288
    // until the records were prefetched (mcd_math.h: RecordPrefetch): with the memory latency hidden, longer chunks win
    // 1.5 - 3 % by writing and reducing fewer partial sums (tools/w128_sweep.py, tools/ab_option.sh).
    int64_t len;
    if (n_wtiles > 4) len = (5 * n * n_wtiles + 6 * target_waves - 1) / std::max<int64_t>(1, 6 * target_waves);
    else if (n_wtiles == 4) len = (n * n_wtiles + target_waves - 1) / std::max<int64_t>(1, target_waves);
    else len = (5 * n * n_wtiles / 4 + (target_waves * 3 / 5 - 136) - 1) / std::max<int64_t>(1, target_waves * 3 / 5 - 136);
    // Several parameter sets (radial bins): only the LAST one gets the guided tail (the launch ends there; a tail of half-
    // and quarter-length chunks at the end of each of 55 bins is 55 x the per-wave overhead for nothing), so the others
    // need 1.25 x shorter chunks for the same number of waves.  C5 (55 bins x 512 walkers): 159.5 -> 145 us per step.
    if (n_psets > 1) len = std::max<int64_t>(1, len * 4 / 5);
    if (chunk_len > 0) {
        len = std::max<int64_t>(64, (chunk_len + 31) / 32 * 32);   // explicit nominal length (option "chunk_len", tuning)
    } else {
        // nearest ODD multiple of 32 (quarter-length tail chunks stay multiples of 8), at least 96: chunk strides that are
        // multiples of 4 KiB (64 stars of 64-byte records, 128 stars of 32-byte records) put the scalar record loads of
        // all resident waves on the same cache channels -- measured 15 - 35 % slower at 128, 256, 384, 512 stars per chunk
        const int64_t k = std::max<int64_t>(1, len / 64);
        len = 32 * (2 * k + 1);
    }
    len = std::min(len, kMaxChunkLen - 32);
    plan.len = len;
    plan.offsets.assign(n_psets + 1, 0);
    for (int64_t p = 0; p < n_psets; ++p) {
        plan.offsets[p] = (int64_t)plan.chunks.size();
        const int64_t b0 = std::max(bin_offsets[p], star_begin);
        const int64_t b1 = std::min(bin_offsets[p + 1], star_begin + n);
        const int64_t total = b1 - b0;
        const int mode = (total >= 16 * len && len >= 128 && p == n_psets - 1) ? tail_split : 0;
        const int64_t resident_chunks = std::max<int64_t>(1, 8192 / n_wtiles);
        for (int64_t s = b0; s < b1;) {
            const int64_t done = s - b0, rem = b1 - s;
            int64_t step = len;
            if (mode == 1) step = done * 100 < total * 85 ? len : (done * 100 < total * 95 ? len / 2 : len / 4);
            else if (mode == 2) step = done * 100 < total * 70 ? len : (done * 100 < total * 85 ? len / 2 : (done * 100 < total * 95 ? len / 4 : len / 8));
            else if (mode == 3) step = std::min(len, std::max<int64_t>(64, rem / (2 * resident_chunks) / 8 * 8));
            else if (mode == 4) step = std::min(len, std::max<int64_t>(128, rem / resident_chunks / 8 * 8));
            step = std::max<int64_t>(8, step / 8 * 8);
            Chunk c;
            c.begin = s - star_begin;
            c.count = (int32_t)std::min(step, rem);
            c.pset = (int32_t)p;
            plan.chunks.push_back(c);
            s += step;
        }
        plan.max_chunks_per_pset = std::max<int64_t>(plan.max_chunks_per_pset, (int64_t)plan.chunks.size() - plan.offsets[p]);
    }
    plan.offsets[n_psets] = (int64_t)plan.chunks.size();
    if (n_psets == 1 && !plan.chunks.empty() && len < (int64_t)1 << 30) {
        bool uniform = true;
        for (size_t i = 0; i < plan.chunks.size() && uniform; ++i) {
            const Chunk u = uniform_chunk((int64_t)i, len, 0, n, (int64_t)plan.chunks.size());
            uniform = plan.chunks[i].begin == u.begin && plan.chunks[i].count == u.count;
        }
        if (uniform) plan.uniform_len = (int)len;
    }
    }   // (multi-round schedules)
    if (!narrow_exceptions.empty()) {
        plan.general.assign(plan.chunks.size(), 0);
        bool any = false;
        for (size_t i = 0; i < plan.chunks.size(); ++i) {
            const int64_t lo = star_begin + plan.chunks[i].begin, hi = lo + plan.chunks[i].count;
            const auto it = std::lower_bound(narrow_exceptions.begin(), narrow_exceptions.end(), lo);
            if (it != narrow_exceptions.end() && *it < hi) { plan.general[i] = 1; any = true; }
        }
        if (!any) plan.general.clear();
    }
    return plan;
}

// Walker-independent part of the fixed-background likelihoods: sum of lnL_bg over the shard's stars of each parameter
// set (Neumaier-compensated).  Added once per output by the reduce kernel when a fast mixture kernel ran.
inline std::vector<double> pset_background_sums(const double* lnlike_bg, const std::vector<int64_t>& bin_offsets,
                                                int64_t star_begin, int64_t n) {
    const int64_t n_psets = (int64_t)bin_offsets.size() - 1;
    std::vector<double> sums(n_psets, 0.0);
    for (int64_t p = 0; p < n_psets; ++p) {
        const int64_t b0 = std::max(bin_offsets[p], star_begin);
        const int64_t b1 = std::min(bin_offsets[p + 1], star_begin + n);
        double sum = 0.0, comp = 0.0;
        for (int64_t i = b0; i < b1; ++i) {
            const double x = lnlike_bg[i], t = sum + x;
            comp += (std::fabs(sum) >= std::fabs(x)) ? (sum - t) + x : (x - t) + sum;
            sum = t;
        }
        sums[p] = sum + comp;
    }
    return sums;
}

// Grid of the main kernel for a chunk table (mcd_kernels.hip: loglike_kernel): up to 4 walker tiles share a workgroup;
// beyond 256 walkers the workgroups of a chunk are dealt so that they share an XCD (groups of 8 chunks).
inline int64_t main_grid(int64_t n_chunks, int64_t n_walkers) {
    const int64_t n_wtiles = (n_walkers + 63) / 64;
    if (n_wtiles <= 4) return (n_chunks * n_wtiles + 3) / 4;
    return (n_chunks + 7) / 8 * 8 * ((n_wtiles + 3) / 4);
}

}  // namespace mcd
