// mcd_api.hip -- C-ABI of the MI355X log-likelihood library (see include/mcd.h).
//
// Host-side responsibilities: device/stream/communicator set-up, one-off upload and packing of the
// star catalogue into HBM, star sharding across devices, chunk tables, per-call launch sequence
//   params H2D -> walker prep -> main kernel -> fixed-order reduce -> [RCCL all-reduce] -> D2H,
// and HIP-event timing for the measurement harness.  No C++ exception leaves this file.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and prototypes only: librccl.so is dlopen'ed on first multi-GPU use

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mcd.h"
#include "mcd_internal.h"
#include "mcd_guard.h"
#include "mcd_math.h"
#include "mcd_rng.h"
#include "mcd_stretch.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    try { g_last_error = msg; } catch (...) { g_last_error.clear(); }       // (assigning can allocate)
    return code;
}

// Every entry point that can allocate host memory (std::vector / std::map / std::string) runs inside try / catch and
// lands here: no C++ exception crosses the C boundary, the caller gets a status code and a message instead.
int on_exception(const char* where) noexcept {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        try { g_last_error = std::string(where) + ": out of host memory"; } catch (...) { g_last_error.clear(); }
        return MCD_ERR_NOMEM;
    } catch (const std::exception& e) {
        try { g_last_error = std::string(where) + ": internal error: " + e.what(); } catch (...) { g_last_error.clear(); }
        return MCD_ERR_INVALID;
    } catch (...) {
        g_last_error.clear();
        return MCD_ERR_INVALID;
    }
}

#define MCD_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(MCD_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));                \
    } while (0)

// RCCL entry points, resolved lazily so that single-GPU processes never load or initialise RCCL.
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
Rccl g_rccl;

std::mutex g_rccl_mutex;

int load_rccl() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);          // contexts may be created from several host threads
    if (g_rccl.handle) return MCD_OK;
    // MCD_RCCL_LIBRARY: an explicit library path.  Used by the tests to substitute tests/fake_rccl (a host-staged
    // stand-in) so that one GPU can run the multi-rank / multi-device call sequences with real shards and kernels.
    void* h = nullptr;
    if (const char* forced = std::getenv("MCD_RCCL_LIBRARY")) {
        h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
        if (!h) return fail(MCD_ERR_RCCL, std::string("cannot load MCD_RCCL_LIBRARY: ") + dlerror());
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(MCD_ERR_RCCL, std::string("cannot load librccl.so: ") + dlerror());
#define MCD_SYM(field, name)                                                                            \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name));                            \
    if (!g_rccl.field) return fail(MCD_ERR_RCCL, std::string("librccl.so lacks ") + name);
    MCD_SYM(GetUniqueId, "ncclGetUniqueId")
    MCD_SYM(CommInitRank, "ncclCommInitRank")
    MCD_SYM(CommInitAll, "ncclCommInitAll")
    MCD_SYM(CommDestroy, "ncclCommDestroy")
    MCD_SYM(AllReduce, "ncclAllReduce")
    MCD_SYM(GroupStart, "ncclGroupStart")
    MCD_SYM(GroupEnd, "ncclGroupEnd")
    MCD_SYM(GetErrorString, "ncclGetErrorString")
    MCD_SYM(CommCount, "ncclCommCount")
    MCD_SYM(CommUserRank, "ncclCommUserRank")
    MCD_SYM(GetVersion, "ncclGetVersion")
#undef MCD_SYM
    g_rccl.handle = h;
    return MCD_OK;
}

#define MCD_NCCL(call)                                                                                  \
    do {                                                                                                \
        ncclResult_t r_ = (call);                                                                       \
        if (r_ != ncclSuccess)                                                                          \
            return fail(MCD_ERR_RCCL, std::string(#call) + ": " + g_rccl.GetErrorString(r_));           \
    } while (0)

struct DeviceSlot {
    int device = 0;
    hipStream_t stream = nullptr;        // kernels, copies
    hipStream_t stream2 = nullptr;       // second compute lane of pipelined evaluations (WorkSet: two lanes)
    hipStream_t comm_stream = nullptr;   // the per-step all-reduce, so that it overlaps the next step's kernels
    ncclComm_t comm = nullptr;
};

// per-(shard, walker-count) work buffers
struct WorkSet {
    int64_t n_walkers = 0;
    int64_t n_chunks = 0;
    int64_t max_chunks_per_pset = 0;
    int uniform_len = 0;               // > 0 when the chunk table is arithmetic (single set; mcd_chunks.h: uniform_chunk)
    int uniform_extra = 0;
    int waves = 4;                     // 8: balanced plan whose workgroups combine their chunks' sums (f64 fast kernels only)
    mcd::Chunk* d_chunks = nullptr;
    int64_t* d_offsets = nullptr;      // [n_psets + 1] chunk offsets
    uint8_t* d_chunk_general = nullptr;   // [n_chunks] chunks excluded from the narrow-range variant; null when there are none
    double* d_params = nullptr;        // [n_psets][W][K]
    void* d_wpar = nullptr;            // [n_psets][W][KD]
    double* d_partials = nullptr;      // [roundup64(W) / 8][n_chunks][8]
    double* d_out = nullptr;           // [n_psets][W] (+ flag word)
    double* d_out2 = nullptr;          // second result buffer: collective mode alternates between the two, so that the
                                       // all-reduce of step i (comm stream) overlaps the kernels of step i + 1
    int buf = 0;                       // buffer the last enqueue wrote (0 for blocking calls without a collective)
    // Pipelined evaluations on ONE device without a collective alternate between two LANES: lane 0 = the compute stream
    // with (d_partials, d_out), lane 1 = the second stream with (d_partials2, d_out2).  Consecutive evaluations are
    // independent (each has its parameters staged), so the reduction and the launch ramp of one overlap the main kernel
    // of the next instead of sitting between two main kernels on one stream (enqueue(): two_lanes).
    double* d_partials2 = nullptr;
    hipEvent_t ev_staged = nullptr;    // parameters staged (on the compute stream): lane 1 waits for it once per staging
    bool lane1_knows_staging = false;
    bool lane1_used = false;           // something may be in flight on the second stream
    hipEvent_t ev_reduced[2] = {nullptr, nullptr};   // reduce kernel done, buffer b ready for the all-reduce
    hipEvent_t ev_comm[2] = {nullptr, nullptr};      // all-reduce of buffer b done
    bool comm_pending[2] = {false, false};
    double* h_params = nullptr;        // pinned + mapped
    double* h_out = nullptr;           // pinned + mapped
    double* m_params = nullptr;        // device view of h_params (zero-copy path of the blocking call)
    double* m_out = nullptr;           // device view of h_out
    bool mapped = false;               // last staging used the zero-copy path: results land in h_out directly
    double launch_tag = 0.0;           // tag of the last fast-path launch (written to out[n_out] by a kernel that wants a re-run)
    int fast = 0;                      // mcd::LaunchShape::fast level of the staged batch
    bool staged = false;
};

// device arena of the resident stretch-move chain and its pinned host mirror (same layout, see stretch_block_device)
struct ChainArena {
    char* d = nullptr;
    char* h = nullptr;
    size_t bytes = 0;
};

struct Shard {
    int slot = 0;                      // index into ctx->slots
    int64_t star_begin = 0;            // global index of the first star held here
    int64_t n = 0;
    void* records = nullptr;
    double* d_pset_const = nullptr;    // BGFIXED: sum of lnlike_bg over this shard's stars of each parameter set
    std::map<int64_t, WorkSet> work;   // keyed by walker count
    hipEvent_t ev_begin = nullptr, ev_k0 = nullptr, ev_k1 = nullptr, ev_end = nullptr;
    // "timing" = 2: one (start, stop) event pair per main-kernel launch, summed by mcd_timing_collect
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ring;
    size_t ring_used = 0;
};

}  // namespace

struct mcd_ctx {
    std::vector<DeviceSlot> slots;
    int rank = 0;
    int n_ranks = 1;
    bool multi_process = false;
    bool force_collective = false;     // MCD_FORCE_RCCL=1: run the all-reduce even on a 1-rank communicator (tests)
    // collective deadline (include/mcd.h): waits on streams that carry an all-reduce poll, bounded by the deadline and
    // by the abort flag another host thread may raise
    int64_t collective_timeout_ms = 120000;
    std::atomic<int> abort_flag{0};
    std::atomic<int> failed{0};
    std::mutex note_mutex;
    std::string abort_reason;          // (guarded by note_mutex)
    std::string failure;               // first failure: stage and cause
    bool has_comm() const { return n_ranks > 1 || slots.size() > 1 || force_collective; }
};

struct mcd_catalog {
    mcd_ctx* ctx = nullptr;
    int model = 0;
    bool free_centre = false;
    int precision = 0;
    int k = 4;
    int64_t n_stars = 0;               // stars held by this process
    int64_t n_psets = 1;
    std::vector<int64_t> bin_offsets;  // [n_psets + 1], indices into this process' stars
    std::vector<Shard> shards;
    mcd::CatalogStats stats;           // range statistics for the fast-path guard (mcd_guard.h)
    // options
    bool timing = false;
    bool timing_all = false;           // keep an event pair for every launch (measurement harness)
    int allow_fast = 1;                // option "fast_path": 0 plain kernels only, 1 guard decides, 2 guard decides but never the narrow variant
    bool zero_copy = true;             // blocking call reads params / writes results through mapped pinned memory
    int64_t timing_stride = 1;         // "timing" = 2: event pair on every n-th launch only (option "timing_stride")
    int64_t timing_launches = 0;
    int64_t spin_us = 20000;           // option "spin_us": poll a stream this long before blocking in hipStreamSynchronize
    int tail_split = 1;                // guided chunk schedule (shorter chunks at the end of a launch)
    int64_t target_waves = 10240;      // see mcd_chunks.h: plan_chunks
    int64_t chunk_len = 0;             // option "chunk_len": explicit nominal chunk length (0: from target_waves)
    int prefetch = -1;                 // option "prefetch": -1 by record volume (>= 8 MiB per device), 0 off, 1 on
    int balance = -1;                  // option "balance": one round of equal waves (mcd_chunks.h): -1 when the catalogue is
                                       // small enough, 0 never, m > 0 forced with m workgroups per CU
    int two_lanes = 1;                 // option "two_lanes": pipelined evaluations of one device alternate between two streams
    int f32_domain = 1;                // option "f32_domain": 1 calls outside the float32 accuracy domain (mcd_guard.h) are refused
                                       // with MCD_ERR_INVALID, 0 they are evaluated anyway (mcd_last_f32_domain tells)
    mcd::F32Domain last_f32;           // verdict on the last staged parameter table (float32 catalogues)
    int combine = 1;                   // option "combine": balanced plans may use 8- / 16-wave workgroups that combine their
                                       // chunks' sums: 0 never, 1 the largest the plan allows, 8 / 16 at most that many waves
    // state of the last evaluation
    int64_t cur_walkers = 0;
    double last_kernel_ms = -1.0, last_device_ms = -1.0;
    bool timing_pending = false;
    int64_t last_grid = 0, last_chunks = 0;
    uint64_t launch_seq = 0;           // source of launch tags
    int64_t n_reruns = 0;              // batches re-evaluated with the plain kernels (denormal regime of the reference)
    // resident stretch-move chain (mcd_stretch.hip)
    int device_chain = 1;              // option "device_chain": 0 host-driven blocks only
    int fused_reduce = 1;              // option "fused_reduce": the step kernel adds up small launches' partial sums itself
    int defer_guard = 1;               // option "defer_guard": one-ensemble resident blocks judge their tables at the end
    bool chain_last_fused = false;
    ChainArena chain;
    int chain_hint = -1;               // kernel family the device's guard asked for when it last disagreed (-1: none)
    int64_t chain_backoff = 0;         // blocks left to run host-driven after a discarded block
    int chain_consecutive = 0;         // discarded blocks in a row (the back-off doubles with each)
    int64_t chain_device_blocks = 0, chain_host_blocks = 0, chain_discarded = 0;
    int chain_last_status = 0;         // status word of the last discarded block (mcd::ChainStatus bits)
    std::vector<hipEvent_t> chain_events;   // large blocks: parts joined by events (stretch_block_device)
    int last_prefetch = -1;            // the last main-kernel launch used the prefetching instantiation (-1: none yet)
};

namespace {

int param_count(int model, bool free_centre) {
    int k = (mcd::is_profile(model) ? 6 : 4) + (free_centre ? 2 : 0);
    const int bg = mcd::bg_kind(model);
    if (bg == mcd::BG_GAUSS) k += 3;
    if (bg == mcd::BG_FIXED_DENSITY) k += 1;
    return k;
}

void free_workset(WorkSet& w) {
    if (w.d_chunks) (void)hipFree(w.d_chunks);
    if (w.d_offsets) (void)hipFree(w.d_offsets);
    if (w.d_chunk_general) (void)hipFree(w.d_chunk_general);
    if (w.d_params) (void)hipFree(w.d_params);
    if (w.d_wpar) (void)hipFree(w.d_wpar);
    if (w.d_partials) (void)hipFree(w.d_partials);
    if (w.d_out) (void)hipFree(w.d_out);
    if (w.d_out2) (void)hipFree(w.d_out2);
    if (w.d_partials2) (void)hipFree(w.d_partials2);
    if (w.ev_staged) (void)hipEventDestroy(w.ev_staged);
    for (int b = 0; b < 2; ++b) {
        if (w.ev_reduced[b]) (void)hipEventDestroy(w.ev_reduced[b]);
        if (w.ev_comm[b]) (void)hipEventDestroy(w.ev_comm[b]);
    }
    if (w.h_params) (void)hipHostFree(w.h_params);
    if (w.h_out) (void)hipHostFree(w.h_out);
    w = WorkSet();
}

// Which catalogues get ONE round of equal waves (mcd_chunks.h: balanced plans) and with how many workgroups per CU.
// Measured on MI355X (tools/balance_sweep.py, us per pipelined step = main kernel + reduction; multi-round schedule /
// best balanced plan): CONST x 256 walkers 1e4 stars 7.9 / 6.7, 3e4 9.7 / 8.8, 1e5 19.2 / 13.3, 2e5 27.4 / 20.3, 4e5
// 40.7 / 33.3, 8e5 67.0 / 63.3, 1.25e6 94.2 / 95.2;  BGFIXED x 256: 1e4 18.1 / 9.1, 1e5 38.1 / 28.5, 4e5 94.0 / 89.7,
// 6e5 129.5 / 130.7, 1e6 202 / 213;  BGGAUSS 1e5 x 256 58.1 / 46.4;  x 128 walkers: CONST 1e5 15.4 / 9.7, BGFIXED 27.5 /
// 18.0.  Small catalogues gain because every CU gets the same number of workgroups (1042 workgroups land as 4 or 5 per
// CU, and the launch waits for the CUs with 5) and because the workgroups add up their chunks' sums themselves; beyond
// ~0.9e6 (CONST) .. 1.3e6 (mixtures) CONST-equivalent stars per 256 walkers the dynamic balancing of 1.5 rounds with a
// guided tail wins.
// "work" = stars x (walker tiles / 4) x (instructions per term / those of CONST): the thresholds are in CONST stars.
double model_cost(int model, bool free_centre) {
    static const double kCost[mcd::kNumModels] = {8.5, 24.0, 45.0, 25.0, 60.0, 40.0, 42.0};   // fast f64 loops, DESIGN 3.3
    return (kCost[model] + (free_centre ? 7.0 : 0.0)) / 8.5;
}
int balance_auto_m(const mcd_catalog* cat, int64_t n, int64_t n_walkers) {
    const int64_t n_wtiles = (n_walkers + 63) / 64;
    const double tiles = n_wtiles <= 4 ? (double)n_wtiles : 4.0 * (double)((n_wtiles + 3) / 4);
    const double work = (double)n * tiles / 4.0 * model_cost(cat->model, cat->free_centre);
    // crossover to the multi-round schedules: CONST 1e6 x 256 is 76.4 us multi-round against 78.6 balanced (8e5: 67.0 / 63.3);
    // the mixtures keep winning a little longer per unit of work (BGFIXED 4e5 stars = 1.1e6 units: 94.0 / 89.7; 6e5: tie)
    const double limit = mcd::bg_kind(cat->model) == mcd::BG_NONE ? 9.0e5 : 1.3e6;
    if (cat->n_psets != 1 || work > limit) return 0;
    return work < 6.0e4 ? 2 : (work <= 3.4e5 ? 4 : 8);
}

// Work buffers of one shard for a given walker count; the chunk table itself is planned by mcd_chunks.h: plan_chunks
// (host-only, unit-tested on the CPU).
int build_workset(mcd_catalog* cat, Shard& sh, int64_t n_walkers, WorkSet** out) {
    auto it = sh.work.find(n_walkers);
    if (it != sh.work.end()) { *out = &it->second; return MCD_OK; }
    if (sh.work.size() >= 8) {                       // bound the cache (emcee uses W and W/2)
        for (auto& kv : sh.work) free_workset(kv.second);
        sh.work.clear();
        cat->cur_walkers = 0;                        // whatever was staged is gone; stage_params sets it again on success
    }
    const DeviceSlot& slot = cat->ctx->slots[sh.slot];
    MCD_HIP(hipSetDevice(slot.device));

    // balanced single-round plan where it pays (option "balance": -1 by the rule above, 0 never, m forced); a catalogue
    // too small for m workgroups per CU (fewer than 16 stars per chunk) takes half as many, down to the multi-round table
    mcd::ChunkPlan plan;
    for (int m = cat->balance < 0 ? balance_auto_m(cat, sh.n, n_walkers) : cat->balance;; m /= 2) {
        plan = mcd::plan_chunks(cat->bin_offsets, sh.star_begin, sh.n, n_walkers, cat->target_waves, cat->tail_split,
                                cat->stats.narrow_exceptions, cat->chunk_len, m);
        if (m == 0 || plan.balanced_m > 0) break;
    }
    const std::vector<mcd::Chunk>& chunks = plan.chunks;
    const std::vector<int64_t>& offs = plan.offsets;
    const std::vector<uint8_t>& general = plan.general;

    WorkSet w;
    w.n_walkers = n_walkers;
    w.n_chunks = (int64_t)chunks.size();
    w.max_chunks_per_pset = plan.max_chunks_per_pset;
    w.uniform_len = plan.uniform_len;
    w.uniform_extra = plan.uniform_extra;
    {
        // balanced plans with an even number of workgroups per CU run as half as many 8-wave workgroups that add their
        // chunks' sums up themselves: half (to an eighth of) the partial sums per walker (mcd_kernels.hip: loglike_kernel)
        const int64_t n_wtiles = (n_walkers + 63) / 64;
        const bool shape_ok = plan.balanced_m > 0 && plan.balanced_m % 2 == 0 && (n_wtiles == 1 || n_wtiles == 2 || n_wtiles == 4) &&
                              cat->precision == MCD_F64;
        w.waves = 4;
        if (cat->combine != 0 && shape_ok) {
            // 4 workgroups per CU as one 16-wave workgroup: 256 partial sums per walker, which the resident chain's step
            // kernel adds up itself (mcd_stretch.hip) -- 3 - 6 % slower than two 8-wave workgroups, one kernel less per
            // half step; 8 per CU stay 8-wave workgroups (1e5 stars x 256 walkers: 14.5 us per step against 15.1)
            const bool can16 = mcd::bg_kind(cat->model) != mcd::BG_GAUSS;
            w.waves = 8;
            if (cat->combine == 16 && plan.balanced_m % 4 == 0 && can16) w.waves = 16;
            // (kernel traces of the C2 bench: 16-wave main kernel 10.7 us + one-wave-per-group reduction 4.6 against 12.0 + 4.0
            // with 8-wave workgroups and 512 partial sums per walker; wall-clock sweeps put the two within their noise)
            if (cat->combine == 1 && plan.balanced_m == 4 && can16) w.waves = 16;
        }
    }
    const int64_t padded_walkers = (n_walkers + 63) / 64 * 64;       // partial sums: whole walker tiles (mcd_kernels.hip)
    const int64_t n_out = cat->n_psets * n_walkers;
    const size_t term_bytes = cat->precision == MCD_F64 ? 8 : 4;
    auto allocate = [&]() -> hipError_t {
        hipError_t e;
        if (!general.empty()) {
            if ((e = hipMalloc(&w.d_chunk_general, general.size())) != hipSuccess) return e;
            if ((e = hipMemcpy(w.d_chunk_general, general.data(), general.size(), hipMemcpyHostToDevice)) != hipSuccess) return e;
        }
        if ((e = hipMalloc(&w.d_chunks, std::max<size_t>(1, chunks.size()) * sizeof(mcd::Chunk))) != hipSuccess) return e;
        if ((e = hipMalloc(&w.d_offsets, offs.size() * sizeof(int64_t))) != hipSuccess) return e;
        if ((e = hipMalloc(&w.d_params, (size_t)n_out * cat->k * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMalloc(&w.d_wpar, (size_t)n_out * mcd::KD * term_bytes)) != hipSuccess) return e;
        if ((e = hipMalloc(&w.d_partials, std::max<size_t>(1, (size_t)padded_walkers * w.n_chunks) * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMalloc(&w.d_out, (size_t)(n_out + 1) * sizeof(double))) != hipSuccess) return e;   // + re-run flag word
        if ((e = hipMemset(w.d_out, 0, (size_t)(n_out + 1) * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMalloc(&w.d_out2, (size_t)(n_out + 1) * sizeof(double))) != hipSuccess) return e;
        if ((e = hipMemset(w.d_out2, 0, (size_t)(n_out + 1) * sizeof(double))) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&w.ev_staged, hipEventDisableTiming)) != hipSuccess) return e;
        for (int b = 0; b < 2; ++b) {
            if ((e = hipEventCreateWithFlags(&w.ev_reduced[b], hipEventDisableTiming)) != hipSuccess) return e;
            if ((e = hipEventCreateWithFlags(&w.ev_comm[b], hipEventDisableTiming)) != hipSuccess) return e;
        }
        if ((e = hipHostMalloc(&w.h_params, (size_t)n_out * cat->k * sizeof(double), hipHostMallocMapped)) != hipSuccess) return e;
        if ((e = hipHostMalloc(&w.h_out, (size_t)(n_out + 1) * sizeof(double), hipHostMallocMapped)) != hipSuccess) return e;
        std::memset(w.h_out, 0, (size_t)(n_out + 1) * sizeof(double));
        if ((e = hipHostGetDevicePointer((void**)&w.m_params, w.h_params, 0)) != hipSuccess) return e;
        if ((e = hipHostGetDevicePointer((void**)&w.m_out, w.h_out, 0)) != hipSuccess) return e;
        if (!chunks.empty() &&
            (e = hipMemcpy(w.d_chunks, chunks.data(), chunks.size() * sizeof(mcd::Chunk), hipMemcpyHostToDevice)) != hipSuccess)
            return e;
        return hipMemcpy(w.d_offsets, offs.data(), offs.size() * sizeof(int64_t), hipMemcpyHostToDevice);
    };
    const hipError_t err = allocate();
    if (err != hipSuccess) {
        free_workset(w);
        return fail(MCD_ERR_HIP, std::string("work buffers for this walker count: ") + hipGetErrorString(err));
    }
    auto ins = sh.work.emplace(n_walkers, w);
    *out = &ins.first->second;
    return MCD_OK;
}

// Wait for a stream: poll it (hipStreamQuery) for up to `spin_us` microseconds before handing the thread to the blocking
// hipStreamSynchronize.  A blocking wait that lasts more than a fraction of a millisecond sleeps on an interrupt and
// wakes the host 50 - 500 us after the device is done (measured as jitter of a 4 ms timed region, tools/k20_probe.py); an
// MCMC driver has nothing else to do with its thread while an evaluation is in flight, so it polls (option "spin_us",
// default 20000; 0 = always block).
hipError_t wait_stream(hipStream_t s, int64_t spin_us) {
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) return q;
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
        }
    }
    return hipStreamSynchronize(s);
}

// The context failed (deadline, abort, an error in the middle of a block of launches other ranks are already committed
// to): remember the first cause; every later call answers MCD_ERR_RCCL at once (ctx_usable).
int ctx_fail(mcd_ctx* ctx, const std::string& what) {
    {
        std::lock_guard<std::mutex> lock(ctx->note_mutex);
        if (!ctx->failed.load()) ctx->failure = what;
        ctx->failed.store(1);
    }
    return fail(MCD_ERR_RCCL, what + " -- the context is unusable from here on: report and exit the process (no fallback "
                                     "inside it; mcd.h: collective deadline)");
}

int ctx_usable(mcd_ctx* ctx) {
    if (!ctx || !ctx->failed.load()) return MCD_OK;
    std::lock_guard<std::mutex> lock(ctx->note_mutex);
    return fail(MCD_ERR_RCCL, "this context failed earlier (" + ctx->failure + "): exit the process");
}

// Wait for a stream of a context.  Without a communicator: wait_stream.  With one, the stream may sit behind an
// all-reduce whose peers never arrive: poll (spin first, then sleep between polls), give up at the deadline or when
// another host thread raises the abort flag, and mark the context failed.  `stage` names the wait in the message.
int wait_ctx_stream(mcd_ctx* ctx, hipStream_t s, int64_t spin_us, const char* stage) {
    if (!ctx->has_comm() || ctx->collective_timeout_ms < 0) {
        const hipError_t e = wait_stream(s, spin_us);
        if (e != hipSuccess) return fail(MCD_ERR_HIP, std::string(stage) + ": " + hipGetErrorString(e));
        return MCD_OK;
    }
    const auto t0 = std::chrono::steady_clock::now();
    const int64_t limit_us = ctx->collective_timeout_ms * 1000;
    for (;;) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return MCD_OK;
        if (q != hipErrorNotReady) return fail(MCD_ERR_HIP, std::string(stage) + ": " + hipGetErrorString(q));
        if (ctx->abort_flag.load(std::memory_order_acquire)) {
            std::string why;
            { std::lock_guard<std::mutex> lock(ctx->note_mutex); why = ctx->abort_reason; }
            return ctx_fail(ctx, std::string(stage) + ": aborted by the host while waiting for a collective (" +
                                 (why.empty() ? "no reason given" : why) + ")");
        }
        const int64_t waited = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        if (limit_us > 0 && waited > limit_us) {
            char buf[256];
            snprintf(buf, sizeof buf, "%s: no completion within collective_timeout_ms = %lld (rank %d of %d): a peer never "
                                      "reached the all-reduce, or the fabric is down",
                     stage, (long long)ctx->collective_timeout_ms, ctx->rank, ctx->n_ranks);
            return ctx_fail(ctx, buf);
        }
        if (waited > spin_us) std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
}

#define MCD_WAIT(ctx, stream, spin, stage)                                                              \
    do {                                                                                                \
        const int w_rc_ = wait_ctx_stream((ctx), (stream), (spin), (stage));                            \
        if (w_rc_ != MCD_OK) return w_rc_;                                                              \
    } while (0)

// Work buffers staged for walker count W (nullptr when the cache no longer holds them, e.g. after a failed upload)
WorkSet* find_work(Shard& sh, int64_t W) {
    const auto it = sh.work.find(W);
    return it == sh.work.end() ? nullptr : &it->second;
}

bool all_staged(mcd_catalog* cat) {
    if (cat->cur_walkers <= 0) return false;
    for (Shard& sh : cat->shards) {
        const WorkSet* w = find_work(sh, cat->cur_walkers);
        if (!w || !w->staged) return false;
    }
    return true;
}

int fast_level(const mcd_catalog* cat, const double* params, int64_t n_rows) {
    if (!cat->allow_fast) return 0;
    const int level = mcd::fast_level(cat->stats, cat->model, cat->free_centre, cat->precision != MCD_F64, cat->k, params, n_rows);
    return cat->allow_fast == 2 && level > 1 ? 1 : level;
}

// Which launches use the prefetching instantiation of the main kernel (mcd_math.h: RecordPrefetch; option "prefetch").
// Measured per shape with the prefetch compiled in and out (1x MI355X, us per step, in / out):
//   mixtures   C3 bgfixed 1e6 x 256: 201 / 217    x 128: 107 / 117    x 64: 62 / 90    bggauss 1e6 x 256: 370 / 381
//   CONST      1e6 x 256: 77.0 / 76.4   x 512: 148 / 144   x 128: 46.4 / 46.9   C5 (55 bins x 512): 173 / 164
//              C5 x 256: 94 / 90        C4 1e7 x 256: 694 / 711                  C2 1e5 x 256: 17.0 / 16.1
// The mixture loops wait for their records (4 stars per iteration, a third of the instructions are one dependent chain);
// the fraction tree of the no-background models reads 16 stars per iteration and hides the latency by itself, so the
// prefetch only pays there when the catalogue is far beyond every cache (C4).
bool wants_prefetch(const mcd_catalog* cat, const Shard& sh) {
    if (cat->prefetch >= 0) return cat->prefetch != 0;
    const size_t bytes = (size_t)sh.n * (size_t)mcd::record_bytes(cat->model, cat->free_centre, cat->precision);
    if (mcd::bg_kind(cat->model) == mcd::BG_NONE) return cat->n_psets == 1 && bytes >= ((size_t)128 << 20);
    return bytes >= ((size_t)8 << 20);
}

int stage_params_impl(mcd_catalog* cat, int64_t n_walkers, int32_t k, const double* params, bool zero_copy) {
    if (!cat || !params) return fail(MCD_ERR_INVALID, "null catalogue or params");
    if (int rc = ctx_usable(cat->ctx)) return rc;
    if (n_walkers <= 0) return fail(MCD_ERR_INVALID, "n_walkers must be positive");
    if (k != cat->k) {
        char buf[128];
        snprintf(buf, sizeof buf, "parameter table has %d columns, catalogue expects %d", (int)k, cat->k);
        return fail(MCD_ERR_INVALID, buf);
    }
    const int64_t n_rows = cat->n_psets * n_walkers;
    if (cat->precision != MCD_F64) {
        // float32 catalogues: is this table inside the domain in which float32 keeps the stated tolerances?
        cat->last_f32 = mcd::f32_domain(cat->stats, cat->model, cat->free_centre, cat->k, params, n_rows);
        if (!cat->last_f32.inside && cat->f32_domain)
            return fail(MCD_ERR_INVALID, std::string("outside the float32 accuracy domain (use an MCD_F64 catalogue, or option "
                                                     "f32_domain = 0 to evaluate regardless): ") + cat->last_f32.reason);
    }
    const int fast = fast_level(cat, params, n_rows);
    for (Shard& sh : cat->shards) {
        WorkSet* w = nullptr;
        int rc = build_workset(cat, sh, n_walkers, &w);
        if (rc != MCD_OK) return rc;
        const DeviceSlot& slot = cat->ctx->slots[sh.slot];
        MCD_HIP(hipSetDevice(slot.device));
        // the pinned staging buffer (and, on lane 1, the walker constants) may still be in flight from the previous call
        MCD_WAIT(cat->ctx, slot.stream, cat->spin_us, "mcd_params_upload (previous evaluation)");
        if (w->lane1_used) {
            MCD_WAIT(cat->ctx, slot.stream2, cat->spin_us, "mcd_params_upload (previous evaluation, second lane)");
            w->lane1_used = false;
        }
        std::memcpy(w->h_params, params, (size_t)n_rows * k * sizeof(double));
        // Blocking single-device call: the walker-prep kernel reads the pinned host table over PCIe and the reduce
        // kernel writes the results straight into pinned host memory -- no copy-engine operations on the critical
        // path.  The pipelined API and multi-device contexts keep device-resident tables.
        w->mapped = zero_copy;
        const double* src = w->m_params;
        if (!zero_copy) {
            MCD_HIP(hipMemcpyAsync(w->d_params, w->h_params, (size_t)n_rows * k * sizeof(double), hipMemcpyHostToDevice,
                                   slot.stream));
            src = w->d_params;
        }
        MCD_HIP(mcd::launch_prepare_walkers(slot.stream, src, n_rows, k, cat->model, cat->free_centre,
                                            cat->precision, w->d_wpar));
        MCD_HIP(hipEventRecord(w->ev_staged, slot.stream));
        w->lane1_knows_staging = false;
        w->fast = fast;
        w->staged = true;
    }
    cat->cur_walkers = n_walkers;
    return MCD_OK;
}

// A failed upload leaves nothing staged: a later enqueue / fetch answers MCD_ERR_INVALID instead of working on buffers
// that may have been evicted.
int stage_params(mcd_catalog* cat, int64_t n_walkers, int32_t k, const double* params, bool zero_copy) {
    const int rc = stage_params_impl(cat, n_walkers, k, params, zero_copy);
    if (rc != MCD_OK && cat) cat->cur_walkers = 0;
    return rc;
}

// pipelined (mcd_loglike_enqueue): the all-reduce goes to the communication stream and overlaps the next step's kernels.
// A blocking call gains nothing from that hop: its all-reduce stays on the compute stream (after any collective still
// pending on the communication stream, so that operations on one communicator never run concurrently).
int enqueue(mcd_catalog* cat, bool pipelined) {
    if (!cat) return fail(MCD_ERR_INVALID, "null catalogue");
    if (int rc = ctx_usable(cat->ctx)) return rc;
    if (cat->cur_walkers <= 0) return fail(MCD_ERR_INVALID, "no parameters staged (call mcd_params_upload first)");
    if (!all_staged(cat)) return fail(MCD_ERR_INVALID, "no parameters staged for this walker count (the last upload failed?)");
    const int64_t W = cat->cur_walkers;
    const int64_t n_out = cat->n_psets * W;
    mcd_ctx* ctx = cat->ctx;
    for (Shard& sh : cat->shards) {
        WorkSet& w = (*find_work(sh, W));
        const DeviceSlot& slot = ctx->slots[sh.slot];
        MCD_HIP(hipSetDevice(slot.device));
        mcd::LaunchShape shape{cat->model, cat->free_centre, cat->precision, w.fast, w.uniform_len, sh.n};
        shape.uniform_extra = w.uniform_extra;
        shape.waves = w.waves;
        // Re-run signal of the fast mixture kernels.  One device: a flag word behind the outputs receives a fresh tag per
        // launch (no reset needed).  Several ranks / devices: the kernels poison the affected partial sums with NaN
        // instead, which travels through the reduce kernel and the all-reduce to every rank.
        const bool coll = ctx->n_ranks > 1 || ctx->slots.size() > 1 || ctx->force_collective;
        double* out_buf = w.mapped ? w.m_out : w.d_out;
        // two lanes: see WorkSet.  (Not with per-launch timing of the harness' plain mode, whose begin / end events
        // bracket ONE stream; the sampled per-kernel events of "timing" = 2 are recorded on the lane's stream.)
        const bool two_lanes = pipelined && cat->two_lanes && !(cat->timing && !cat->timing_all);
        hipStream_t lane_stream = slot.stream;
        double* lane_partials = w.d_partials;
        if (coll || two_lanes) {
            // alternate result buffers (and, with two lanes, streams and partial-sum buffers).  With a collective this step
            // may only overwrite its buffer once the all-reduce that last used it (two steps ago, on the communication
            // stream) has finished; the all-reduces themselves stay in order on that one stream.
            w.buf ^= 1;
            out_buf = w.buf ? w.d_out2 : w.d_out;
            if (two_lanes && w.buf) {
                if (!w.d_partials2) {
                    const int64_t padded_walkers = (W + 63) / 64 * 64;
                    MCD_HIP(hipMalloc(&w.d_partials2, std::max<size_t>(1, (size_t)padded_walkers * w.n_chunks) * sizeof(double)));
                }
                if (!w.lane1_knows_staging) {
                    MCD_HIP(hipStreamWaitEvent(slot.stream2, w.ev_staged, 0));
                    w.lane1_knows_staging = true;
                }
                lane_stream = slot.stream2;
                lane_partials = w.d_partials2;
                w.lane1_used = true;
            }
        } else {
            w.buf = 0;
        }
        shape.chunk_general = w.d_chunk_general;
        // records beyond what the caches hold between two passes: prefetch the next loop iteration's records (mcd_math.h)
        shape.prefetch = wants_prefetch(cat, sh);
        cat->last_prefetch = shape.prefetch && shape.fast != 0;
        shape.rerun_flag = coll ? nullptr : out_buf + n_out;
        w.launch_tag = coll ? 0.0 : (double)(++cat->launch_seq);
        shape.launch_tag = w.launch_tag;
        hipEvent_t k0 = sh.ev_k0, k1 = sh.ev_k1;
        // per-launch events cost a signal packet each (~3 us per pair between back-to-back kernels): a harness may sample
        const bool sampled = !cat->timing_all || (cat->timing_launches % cat->timing_stride) == 0;
        if (cat->timing_all && sampled) {
            if (sh.ring_used >= (size_t)1 << 16) sh.ring_used = 0;        // harness option left on: recycle, never grow without bound
            if (sh.ring_used == sh.ring.size()) {
                hipEvent_t a, b;
                MCD_HIP(hipEventCreate(&a));
                MCD_HIP(hipEventCreate(&b));
                sh.ring.emplace_back(a, b);
            }
            k0 = sh.ring[sh.ring_used].first;
            k1 = sh.ring[sh.ring_used].second;
            ++sh.ring_used;
        }
        if (cat->timing && !cat->timing_all) MCD_HIP(hipEventRecord(sh.ev_begin, slot.stream));
        if (cat->timing && sampled) MCD_HIP(hipEventRecord(k0, lane_stream));
        MCD_HIP(mcd::launch_loglike(lane_stream, shape, sh.records, w.d_chunks, w.n_chunks, w.d_wpar, lane_partials, W));
        if (cat->timing && sampled) MCD_HIP(hipEventRecord(k1, lane_stream));
        // the fast BGFIXED kernel leaves the walker-independent sum of lnL_bg to the reduction
        const int bgk = mcd::bg_kind(cat->model);
        const double* pset_const =
            (w.fast && (bgk == mcd::BG_FIXED || bgk == mcd::BG_FIXED_DENSITY)) ? sh.d_pset_const : nullptr;
        // With a collective this step may only overwrite its result buffer once the all-reduce that last used it has
        // finished.  Only the REDUCTION writes that buffer (the main kernel's re-run signal travels in the partial sums
        // here), so the wait sits in front of it, not in front of the main kernel: with two lanes a lane reuses the buffer
        // of its own previous step, and the all-reduce of that step would otherwise be on the lane's critical path.
        if (coll && w.comm_pending[w.buf]) {
            MCD_HIP(hipStreamWaitEvent(lane_stream, w.ev_comm[w.buf], 0));
            w.comm_pending[w.buf] = false;
        }
        {
            const int64_t n_slots = mcd::partial_slots(shape, w.n_chunks, W);
            MCD_HIP(mcd::launch_reduce(lane_stream, lane_partials, w.d_offsets, cat->n_psets, n_slots,
                                       cat->n_psets == 1 ? n_slots : w.max_chunks_per_pset, W, pset_const, out_buf));
        }
        if (coll && pipelined) MCD_HIP(hipEventRecord(w.ev_reduced[w.buf], lane_stream));
    }
    // sum the per-device / per-rank partial log-likelihoods: one all-reduce of n_out doubles
    const bool collective = ctx->n_ranks > 1 || ctx->slots.size() > 1 || ctx->force_collective;
    if (collective) {
        if (!ctx->multi_process) MCD_NCCL(g_rccl.GroupStart());
        for (Shard& sh : cat->shards) {
            WorkSet& w = (*find_work(sh, W));
            const DeviceSlot& slot = ctx->slots[sh.slot];
            MCD_HIP(hipSetDevice(slot.device));
            double* buf = w.buf ? w.d_out2 : w.d_out;
            if (pipelined) {
                MCD_HIP(hipStreamWaitEvent(slot.comm_stream, w.ev_reduced[w.buf], 0));
                MCD_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n_out, ncclDouble, ncclSum, slot.comm, slot.comm_stream));
            } else {
                if (w.comm_pending[w.buf ^ 1]) {           // the newest collective still on the communication stream
                    MCD_HIP(hipStreamWaitEvent(slot.stream, w.ev_comm[w.buf ^ 1], 0));
                    w.comm_pending[w.buf ^ 1] = false;
                }
                MCD_NCCL(g_rccl.AllReduce(buf, buf, (size_t)n_out, ncclDouble, ncclSum, slot.comm, slot.stream));
            }
        }
        if (!ctx->multi_process) MCD_NCCL(g_rccl.GroupEnd());
        if (pipelined) {
            for (Shard& sh : cat->shards) {
                WorkSet& w = (*find_work(sh, W));
                const DeviceSlot& slot = ctx->slots[sh.slot];
                MCD_HIP(hipSetDevice(slot.device));
                MCD_HIP(hipEventRecord(w.ev_comm[w.buf], slot.comm_stream));
                w.comm_pending[w.buf] = true;
            }
        }
    }
    if (cat->timing_all) ++cat->timing_launches;
    if (cat->timing) {
        if (!cat->timing_all) {
            for (Shard& sh : cat->shards) {
                const DeviceSlot& slot = ctx->slots[sh.slot];
                MCD_HIP(hipSetDevice(slot.device));
                MCD_HIP(hipEventRecord(sh.ev_end, slot.stream));
            }
        }
        cat->timing_pending = true;
    }
    {
        WorkSet& w0 = (*find_work(cat->shards[0], W));
        cat->last_chunks = w0.n_chunks;
        mcd::LaunchShape sh0{cat->model, cat->free_centre, cat->precision, w0.fast};
        sh0.waves = w0.waves;
        const int64_t slots = mcd::partial_slots(sh0, w0.n_chunks, W);
        cat->last_grid = slots != w0.n_chunks ? slots : mcd::main_grid(w0.n_chunks, W);
    }
    return MCD_OK;
}

int sync_all(mcd_catalog* cat) {
    if (!cat) return fail(MCD_ERR_INVALID, "null catalogue");
    if (int rc = ctx_usable(cat->ctx)) return rc;
    for (Shard& sh : cat->shards) {
        const DeviceSlot& slot = cat->ctx->slots[sh.slot];
        MCD_HIP(hipSetDevice(slot.device));
        MCD_WAIT(cat->ctx, slot.stream, cat->spin_us, "mcd_sync / mcd_loglike_fetch (compute stream)");
        MCD_WAIT(cat->ctx, slot.stream2, cat->spin_us, "mcd_sync / mcd_loglike_fetch (second compute lane)");
        MCD_WAIT(cat->ctx, slot.comm_stream, cat->spin_us, "mcd_sync / mcd_loglike_fetch (communication stream)");
    }
    if (cat->timing && cat->timing_pending) {
        Shard& sh = cat->shards[0];
        float k_ms = 0.f, d_ms = 0.f;
        if (cat->timing_all) {
            if (sh.ring_used > 0)          // (none yet when only unsampled launches ran since the last collect)
                MCD_HIP(hipEventElapsedTime(&k_ms, sh.ring[sh.ring_used - 1].first, sh.ring[sh.ring_used - 1].second));
        } else {
            MCD_HIP(hipEventElapsedTime(&k_ms, sh.ev_k0, sh.ev_k1));
        }
        if (!cat->timing_all) MCD_HIP(hipEventElapsedTime(&d_ms, sh.ev_begin, sh.ev_end));
        cat->last_kernel_ms = k_ms;
        cat->last_device_ms = cat->timing_all ? -1.0 : d_ms;
        cat->timing_pending = false;
    }
    return MCD_OK;
}

int fetch_once(mcd_catalog* cat, bool* rerun) {
    const int64_t W = cat->cur_walkers;
    const int64_t n_out = cat->n_psets * W;
    *rerun = false;
    const bool coll = cat->ctx->n_ranks > 1 || cat->ctx->slots.size() > 1 || cat->ctx->force_collective;
    bool any_fast = false;
    for (Shard& sh : cat->shards) {
        WorkSet& w = (*find_work(sh, W));
        any_fast = any_fast || w.fast != 0;
        const DeviceSlot& slot = cat->ctx->slots[sh.slot];
        MCD_HIP(hipSetDevice(slot.device));
        // after the all-reduce every device holds the same results: only the first shard's are copied
        if (!w.mapped && &sh == &cat->shards[0]) {
            const double* res = w.buf ? w.d_out2 : w.d_out;
            if (w.comm_pending[w.buf]) MCD_HIP(hipStreamWaitEvent(slot.stream, w.ev_comm[w.buf], 0));
            // (single device, two lanes: the newest results sit behind the work of the lane that produced them)
            hipStream_t copy_stream = (!coll && w.buf && w.lane1_used) ? slot.stream2 : slot.stream;
            MCD_HIP(hipMemcpyAsync(w.h_out, res, (size_t)(n_out + 1) * sizeof(double), hipMemcpyDeviceToHost, copy_stream));
        }
    }
    int rc = sync_all(cat);
    if (rc != MCD_OK) return rc;
    const WorkSet& w0 = (*find_work(cat->shards[0], W));
    if (coll) {
        // Every rank decides on the all-reduced values alone (identical everywhere), whatever kernel family it ran itself:
        // the re-evaluation is collective.  (A NaN that the plain kernels produce legitimately costs one extra pass.)
        for (int64_t i = 0; i < n_out && !*rerun; ++i) *rerun = w0.h_out[i] != w0.h_out[i];      // NaN-poisoned sums
    } else if (any_fast) {
        *rerun = w0.h_out[n_out] == w0.launch_tag;
    }
    return MCD_OK;
}

int fetch(mcd_catalog* cat, double* out) {
    if (!cat || !out) return fail(MCD_ERR_INVALID, "null catalogue or output");
    if (cat->cur_walkers <= 0 || !all_staged(cat)) return fail(MCD_ERR_INVALID, "nothing evaluated yet");
    const int64_t W = cat->cur_walkers;
    const int64_t n_out = cat->n_psets * W;
    bool rerun = false;
    int rc = fetch_once(cat, &rerun);
    if (rc != MCD_OK) return rc;
    if (rerun) {
        // A fast mixture kernel met the regime where the reference's log-sum-exp runs on denormal numbers (a star with
        // pmember == 1, f_back == 0 or density == 0 that is a > 37 sigma outlier of the remaining component).  Only the
        // plain kernels reproduce the reference's value there: evaluate the staged batch again with them.  In a
        // multi-rank job every rank takes the same decision (the all-reduce is collective): the affected partial sums
        // are NaN-poisoned by the kernel, so the all-reduced results carry the signal to every rank (fetch_once).
        ++cat->n_reruns;
        for (Shard& sh : cat->shards) (*find_work(sh, W)).fast = 0;
        rc = enqueue(cat, false);
        if (rc != MCD_OK) return rc;
        rc = fetch_once(cat, &rerun);
        if (rc != MCD_OK) return rc;
    }
    std::memcpy(out, (*find_work(cat->shards[0], W)).h_out, (size_t)n_out * sizeof(double));
    return MCD_OK;
}

// ------------------------------------------------------------------------------------------------
// mcd_stretch_move with the ensemble resident on the device (mcd_stretch.hip).  Runs the whole block as one chain of
// launches and waits once.  *done = false when the block has to be run host-driven: the configuration is not covered, or
// the device met something only the host loop handles (mcd::ChainStatus).  `pos`, `lnp` and `accepted` are then
// untouched; the chain rows of a large block (cut into parts) may already hold rows of the discarded attempt, which the
// host-driven re-run overwrites (and which stay if that re-run fails).
// user <-> pinned copies of tens of MB (a binned block: 40 MB of random numbers in, 70 MB of chain rows out) on four
// threads: one core moves ~10 GB/s, the copies would otherwise cost as much as the block's device time
void big_copy(void* dst, const void* src, size_t bytes) {
    constexpr size_t kSerial = (size_t)4 << 20;
    if (bytes < kSerial) { std::memcpy(dst, src, bytes); return; }
    constexpr int kMaxThreads = 16;
    static const int kThreads = [] {
        const char* e = std::getenv("MCD_COPY_THREADS");                  // tuning aid
        const int n = e ? std::atoi(e) : 4;
        return n < 1 ? 1 : (n > kMaxThreads ? kMaxThreads : n);
    }();
    std::thread workers[kMaxThreads - 1];
    const size_t part = (bytes / kThreads + 63) / 64 * 64;
    int started = 0;
    for (int t = 1; t < kThreads; ++t) {
        const size_t at = std::min(bytes, part * t), len = std::min(bytes, part * (t + 1)) - at;
        try {
            workers[t - 1] = std::thread([=] { if (len) std::memcpy((char*)dst + at, (const char*)src + at, len); });
            ++started;
        } catch (...) {                                        // no thread to be had: this share is copied here instead
            if (len) std::memcpy((char*)dst + at, (const char*)src + at, len);
        }
    }
    std::memcpy(dst, src, std::min(bytes, part));
    for (int t = 0; t < kThreads - 1; ++t)
        if (workers[t].joinable()) workers[t].join();
    (void)started;
}

// `seed` != nullptr: a seeded block (mcd_stretch_move_seeded) -- order / zz / thr / pick are null, a kernel generates the
// numbers of absolute steps step0 .. step0 + n_steps - 1 on the device (mcd_rng.h, mcd_stretch.hip: chain_numbers_kernel)
int stretch_block_device(mcd_catalog* cat, const mcd_stretch_desc* d, int64_t n_steps, double* pos, double* lnp,
                         const int32_t* order, const double* zz, const double* thr, const int32_t* pick, double* chain,
                         double* lnprob_chain, int64_t* accepted, bool* done, const uint64_t* seed = nullptr,
                         int64_t step0 = 0) {
    *done = false;
    mcd_ctx* ctx = cat->ctx;
    if (int urc = ctx_usable(ctx)) return urc;
    const int64_t B = cat->n_psets;                    // ensembles: one per parameter set (mcd_stretch_move checked n_bins)
    if (!cat->device_chain || cat->shards.size() != 1 || cat->precision != MCD_F64 || n_steps < 1 || cat->timing || !d->fixed_ok)
        return MCD_OK;
    if (B > 1 && cat->device_chain != 1) return MCD_OK;            // (the testing variants exist for one ensemble only)
    if (cat->chain_backoff > 0) { --cat->chain_backoff; return MCD_OK; }
    const int64_t W = d->n_walkers, half = W / 2;
    const int P = d->n_dim, K = d->k;
    bool small_ensemble = false;                                   // the step kernel that keeps the ensemble in LDS takes it
    {
        mcd::StretchDevice probe;
        probe.n_bins = B; probe.n_walkers = W; probe.n_dim = P; probe.k = K;
        probe.force_general = cat->device_chain == 2;
        if (!mcd::stretch_step_handles(probe)) return MCD_OK;
        small_ensemble = mcd::stretch_step_fuses(probe);
    }
    Shard& sh = cat->shards[0];
    const DeviceSlot& slot = ctx->slots[sh.slot];
    MCD_HIP(hipSetDevice(slot.device));
    WorkSet* wp = nullptr;
    int rc = build_workset(cat, sh, half, &wp);
    if (rc != MCD_OK) return rc;
    WorkSet& w = *wp;
    // nothing of an earlier call may still use the work buffers, the arena or the communicator
    MCD_WAIT(cat->ctx, slot.stream, cat->spin_us, "mcd_stretch_move (previous evaluation)");
    MCD_WAIT(cat->ctx, slot.stream2, cat->spin_us, "mcd_stretch_move (previous evaluation, second lane)");
    MCD_WAIT(cat->ctx, slot.comm_stream, cat->spin_us, "mcd_stretch_move (previous collective)");
    w.comm_pending[0] = w.comm_pending[1] = false;
    w.staged = false;                     // the walker constants are about to be overwritten on the device

    // ---- arena layout: [state, copied both ways | inputs, copied in | scratch | chain rows, copied out] ----
    // (every per-walker array carries the ensemble index in front: [B][W]..., random numbers [n_steps][..][B][..])
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 63) / 64 * 64; return at; };
    const size_t BW = (size_t)B * W, Bh = (size_t)B * half;
    const size_t o_pos = take(BW * P * 8), o_lnp = take(BW * 8), o_acc = take(BW * 8);
    const size_t o_meta = take(mcd::META_WORDS * 4), o_status = take(8);
    // deferred guard (one ensemble, the in-LDS step kernel; option "defer_guard"): every launch's table rows are kept and
    // judged together at the end of the block (mcd_stretch.hip: stretch_judge_kernel) -- unless the log would be large
    const size_t n_launches = (size_t)2 * n_steps;
    const bool defer = cat->defer_guard != 0 && B == 1 && small_ensemble && n_launches * half * K * 8 <= ((size_t)64 << 20);
    const size_t o_levels = take(defer ? n_launches * 4 : 0);
    const size_t state_end = off;
    const size_t o_src = take((size_t)K * 4), o_const = take((size_t)K * 8), o_fac = take((size_t)K * 8);
    const size_t o_lo = take((size_t)P * 8), o_hi = take((size_t)P * 8);
    const bool seeded = seed != nullptr;
    // seeded blocks: a kernel writes the numbers where the upload would have put them (ensembles too large for it: the host
    // build of the same functions fills the pinned copy, uploaded as usual)
    const bool gen_device = seeded && mcd::chain_numbers_on_device(W);
    const size_t n_in = (size_t)n_steps;
    const size_t o_order = take(n_in * BW * 4), o_zz = take(n_in * BW * 8);
    const size_t o_thr = take(n_in * BW * 8), o_pick = take(n_in * BW * 4);
    const size_t input_end = off;
    const size_t o_prop = take(Bh * P * 8), o_ok = take(Bh), o_nok = take((size_t)2 * B * 4), o_ranges = take((size_t)2 * B * 10 * 8);
    const size_t o_nok_log = take(defer ? n_launches * 4 : 0), o_table_log = take(defer ? n_launches * half * K * 8 : 0);
    const size_t o_chain = take(chain ? (size_t)n_steps * BW * P * 8 : 0);
    const size_t o_lnpc = take(lnprob_chain ? (size_t)n_steps * BW * 8 : 0);
    const size_t total = off;
    ChainArena& a = cat->chain;
    if (a.bytes < total) {
        if (a.d) (void)hipFree(a.d);
        if (a.h) (void)hipHostFree(a.h);
        a = ChainArena();
        const size_t want = total + total / 2;
        // (a block too large for a device arena or for pinned host memory is no error on ONE device: it runs host-driven,
        // which needs neither; callers keep blocks of many ensembles short, analysis/binned.py.  In a multi-rank job the
        // choice between the resident and the host-driven sequence must be the same on every rank -- their collectives
        // differ in number and size -- so there a rank that cannot allocate reports the error instead)
        const bool must_agree = ctx->n_ranks > 1 || ctx->force_collective;
        if (hipMalloc((void**)&a.d, want) != hipSuccess) {
            (void)hipGetLastError();
            a = ChainArena();
            if (must_agree) return fail(MCD_ERR_NOMEM, "mcd_stretch_move: no device memory for the resident block of this rank (use shorter blocks)");
            return MCD_OK;
        }
        if (hipHostMalloc((void**)&a.h, want, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(a.d);
            a = ChainArena();
            if (must_agree) return fail(MCD_ERR_NOMEM, "mcd_stretch_move: no pinned host memory for the resident block of this rank (use shorter blocks)");
            return MCD_OK;
        }
        a.bytes = want;
    }
    std::memcpy(a.h + o_pos, pos, BW * P * 8);
    std::memcpy(a.h + o_lnp, lnp, BW * 8);
    if (accepted) std::memcpy(a.h + o_acc, accepted, BW * 8);
    else std::memset(a.h + o_acc, 0, BW * 8);
    std::memset(a.h + o_meta, 0, mcd::META_WORDS * 4);
    std::memset(a.h + o_status, 0, 8);
    std::memcpy(a.h + o_src, d->col_source, (size_t)K * 4);
    std::memcpy(a.h + o_const, d->col_const, (size_t)K * 8);
    std::memcpy(a.h + o_fac, d->col_factor, (size_t)K * 8);
    std::memcpy(a.h + o_lo, d->lo, (size_t)P * 8);
    std::memcpy(a.h + o_hi, d->hi, (size_t)P * 8);
    // Blocks that move tens of MB (binned catalogues: the random numbers of 64 steps of 55 x 512 walkers are 40 MB, their
    // chain rows 70 MB -- a third of the block's device time in copies) are cut into parts: the host copies and the PCIe
    // transfers of one part overlap the device work of another (copies on the second stream, joined by events).
    const size_t moved = (size_t)n_steps * BW * ((gen_device ? 0 : 24) + (chain ? (size_t)P * 8 : 0) + (lnprob_chain ? 8 : 0));
    size_t part_threshold = (size_t)16 << 20;
    if (const char* e = std::getenv("MCD_CHAIN_PART_BYTES")) part_threshold = (size_t)std::strtoull(e, nullptr, 10);   // testing aid
    // (what stays exposed is the first part's numbers going in and the last part's rows coming out: 1 / n_parts of the copies.
    // C5 shape, 64-step blocks: 233.8 us per step with 4 parts, 219.6 with 8, 216.7 with 16, 213.8 with 32)
    int64_t max_parts = 16;
    if (const char* e = std::getenv("MCD_CHAIN_PARTS")) max_parts = std::max<long long>(1, std::atoll(e));   // tuning aid
    const int64_t n_parts = moved >= part_threshold ? std::min<int64_t>(max_parts, n_steps) : 1;
    auto part_begin = [&](int64_t k) { return n_steps * k / n_parts; };
    auto copy_in = [&](int64_t i0, int64_t i1) {              // the random numbers of steps i0 .. i1: user -> pinned
        if (gen_device) return;
        const size_t at = (size_t)i0 * BW, n = (size_t)(i1 - i0) * BW;
        if (seeded) {
            std::vector<uint64_t> sorter;
            for (int64_t i = i0; i < i1; ++i)
                mcd::chain_numbers_of_step(*seed, step0 + i, B, W, P, (int32_t*)(a.h + o_order) + (size_t)i * BW,
                                           (double*)(a.h + o_zz) + (size_t)i * BW, (double*)(a.h + o_thr) + (size_t)i * BW,
                                           (int32_t*)(a.h + o_pick) + (size_t)i * BW, sorter);
            return;
        }
        big_copy(a.h + o_order + at * 4, order + at, n * 4);
        big_copy(a.h + o_zz + at * 8, zz + at, n * 8);
        big_copy(a.h + o_thr + at * 8, thr + at, n * 8);
        big_copy(a.h + o_pick + at * 4, pick + at, n * 4);
    };
    if (n_parts == 1) copy_in(0, n_steps);

    // kernel family of this block: what the device's guard last asked for, else the verdict on the current positions
    int level = cat->chain_hint;
    if (level < 0) {
        std::vector<double> table(BW * K);
        for (size_t j = 0; j < BW; ++j)
            for (int c = 0; c < K; ++c) {
                const int src = d->col_source[c];
                table[j * K + c] = src < 0 ? d->col_const[c]
                                           : (d->col_factor[c] == 1.0 ? pos[j * P + src] : pos[j * P + src] * d->col_factor[c]);
            }
        level = fast_level(cat, table.data(), (int64_t)BW);
    }

    mcd::StretchDevice sd;
    sd.n_bins = B;
    sd.n_walkers = W; sd.n_dim = P; sd.k = K; sd.fixed_ok = d->fixed_ok;
    sd.model = cat->model; sd.free_centre = cat->free_centre ? 1 : 0; sd.allow_fast = cat->allow_fast;
    sd.expected_level = level;
    sd.force_general = cat->device_chain == 2;
    sd.stats = cat->stats;                                              // (slices to the scalar part)
    sd.col_source = (const int32_t*)(a.d + o_src); sd.col_const = (const double*)(a.d + o_const);
    sd.col_factor = (const double*)(a.d + o_fac); sd.lo = (const double*)(a.d + o_lo); sd.hi = (const double*)(a.d + o_hi);
    sd.pos = (double*)(a.d + o_pos); sd.lnp = (double*)(a.d + o_lnp); sd.accepted = (long long*)(a.d + o_acc);
    sd.order = (const int32_t*)(a.d + o_order); sd.zz = (const double*)(a.d + o_zz); sd.thr = (const double*)(a.d + o_thr);
    sd.pick = (const int32_t*)(a.d + o_pick);
    sd.chain = chain ? (double*)(a.d + o_chain) : nullptr;
    sd.lnprob_chain = lnprob_chain ? (double*)(a.d + o_lnpc) : nullptr;
    sd.proposal = (double*)(a.d + o_prop); sd.ok = (uint8_t*)(a.d + o_ok); sd.meta = (int32_t*)(a.d + o_meta);
    sd.n_ok = (int32_t*)(a.d + o_nok); sd.ranges = (double*)(a.d + o_ranges);
    sd.table = w.d_params; sd.wpar = (double*)w.d_wpar;
    if (defer) {
        sd.defer_guard = 1;
        sd.table_log = (double*)(a.d + o_table_log); sd.n_ok_log = (int32_t*)(a.d + o_nok_log); sd.level_log = (int32_t*)(a.d + o_levels);
    }

    const bool coll = ctx->n_ranks > 1 || ctx->force_collective;
    mcd::LaunchShape shape{cat->model, cat->free_centre, cat->precision, level, w.uniform_len, sh.n};
    shape.uniform_extra = w.uniform_extra;
    shape.waves = w.waves;
    shape.chunk_general = w.d_chunk_general;
    shape.prefetch = wants_prefetch(cat, sh);
    cat->last_prefetch = shape.prefetch && shape.fast != 0;
    double* const out_buf = w.d_out;
    shape.rerun_flag = coll ? nullptr : out_buf + Bh;
    const int bgk = mcd::bg_kind(cat->model);
    const double* pset_const = (level && (bgk == mcd::BG_FIXED || bgk == mcd::BG_FIXED_DENSITY)) ? sh.d_pset_const : nullptr;
    // launches with few partial sums per walker (balanced plans of small catalogues, radial bins of a few chunks): no
    // reduction kernel, the step kernel adds them up itself -- the same code in the same order (mcd_reduce.h), so the
    // chain keeps the bits of the host-driven loop.  Not with a collective: the all-reduce needs the sums in memory.
    const int64_t n_slots = mcd::partial_slots(shape, w.n_chunks, half);
    const int64_t max_slots = B == 1 ? n_slots : w.max_chunks_per_pset;
    const bool fused = cat->fused_reduce != 0 && !coll && max_slots <= mcd::kFusedReduceSlots && mcd::stretch_step_fuses(sd);
    sd.fused = fused ? 1 : 0;
    sd.partials = w.d_partials;
    sd.n_slots = n_slots;
    sd.slot_offsets = B == 1 ? nullptr : w.d_offsets;
    sd.pset_const = pset_const;
    cat->chain_last_fused = fused;
    double prev_tag = 0.0;
    int64_t acc_step = -1;
    int acc_h = 0;
#ifdef MCD_CHAIN_STAMPS
    static unsigned long long* d_stamps = nullptr;
    if (!d_stamps) MCD_HIP(hipMalloc((void**)&d_stamps, 8 * 8 * 4096));
    MCD_HIP(hipMemsetAsync(d_stamps, 0, 8 * 8 * 4096, slot.stream));
    sd.stamps = d_stamps;
#endif
    // launches of steps i0 .. i1; `mark` (may be null) is recorded right after the first step kernel, i.e. when the chain
    // rows of every step before i0 are final (the row of step i0 - 1 is written by that launch)
    auto enqueue_steps = [&](int64_t i0, int64_t i1, hipEvent_t mark) -> int {
        for (int64_t i = i0; i < i1; ++i)
            for (int h = 0; h < 2; ++h) {
#ifdef MCD_CHAIN_STAMPS
                sd.launch_index = i * 2 + h;
#endif
                MCD_HIP(mcd::launch_stretch_step(slot.stream, sd, acc_step, acc_h, i, h, out_buf, prev_tag));
                if (mark && i == i0 && h == 0) MCD_HIP(hipEventRecord(mark, slot.stream));
                prev_tag = coll ? 0.0 : (double)(++cat->launch_seq);
                shape.launch_tag = prev_tag;
                MCD_HIP(mcd::launch_loglike(slot.stream, shape, sh.records, w.d_chunks, w.n_chunks, w.d_wpar, w.d_partials, half));
                if (!fused)
                    MCD_HIP(mcd::launch_reduce(slot.stream, w.d_partials, w.d_offsets, B, n_slots, max_slots, half, pset_const,
                                               out_buf));
                if (coll) {
                    if (!ctx->multi_process) MCD_NCCL(g_rccl.GroupStart());
                    MCD_NCCL(g_rccl.AllReduce(out_buf, out_buf, Bh, ncclDouble, ncclSum, slot.comm, slot.stream));
                    if (!ctx->multi_process) MCD_NCCL(g_rccl.GroupEnd());
                }
                acc_step = i;
                acc_h = h;
            }
        return MCD_OK;
    };
    auto finish = [&]() -> int {                               // the last accept [+ the collective status word]
        MCD_HIP(mcd::launch_stretch_step(slot.stream, sd, acc_step, acc_h, -1, 0, out_buf, prev_tag));
        if (defer) MCD_HIP(mcd::launch_stretch_judge(slot.stream, sd, (int64_t)n_launches));
        if (coll) {
            // ranks hold different shares of the catalogue, so their guard verdicts may differ: all discard the block if one does
            double* status = (double*)(a.d + o_status);
            MCD_HIP(mcd::launch_stretch_status(slot.stream, sd.meta, status));
            if (!ctx->multi_process) MCD_NCCL(g_rccl.GroupStart());
            MCD_NCCL(g_rccl.AllReduce(status, status, 1, ncclDouble, ncclSum, slot.comm, slot.stream));
            if (!ctx->multi_process) MCD_NCCL(g_rccl.GroupEnd());
        }
        return MCD_OK;
    };
    // From the first launch on, the other ranks of a multi-rank job are committed to this block's sequence of
    // collectives: an error on this rank in the middle of it (a failed launch, a refused ncclAllReduce) cannot be undone
    // locally -- the peers would wait in an all-reduce this rank never enters.  The context is marked failed (the host
    // tells the peers: hostgroup.abort -> mcd_ctx_abort on their side; their deadline bounds the wait otherwise).
    struct MidBlockGuard {
        mcd_ctx* ctx; bool armed; bool ok;
        ~MidBlockGuard() {
            if (armed && !ok && !ctx->failed.load()) {
                const std::string cause = g_last_error;
                (void)ctx_fail(ctx, "mcd_stretch_move: this rank failed in the middle of a resident block (" + cause + ")");
            }
        }
    } mid_block{ctx, coll, false};
    bool rows_delivered = false;                               // chain rows already in the caller's arrays (parts)
    auto generate = [&](hipStream_t s, int64_t i0, int64_t i1) {   // seeded: the numbers of steps i0 .. i1, on the device
        return mcd::launch_chain_numbers(s, *seed, step0, i0, i1, B, W, P, (int32_t*)(a.d + o_order), (double*)(a.d + o_zz),
                                         (double*)(a.d + o_thr), (int32_t*)(a.d + o_pick));
    };
    if (n_parts == 1) {
        MCD_HIP(hipMemcpyAsync(a.d, a.h, gen_device ? o_order : input_end, hipMemcpyHostToDevice, slot.stream));
        if (gen_device) MCD_HIP(generate(slot.stream, 0, n_steps));
        rc = enqueue_steps(0, n_steps, nullptr);
        if (rc != MCD_OK) return rc;
        rc = finish();
        if (rc != MCD_OK) return rc;
        MCD_HIP(hipMemcpyAsync(a.h, a.d, state_end, hipMemcpyDeviceToHost, slot.stream));
        if (total > o_chain)
            MCD_HIP(hipMemcpyAsync(a.h + o_chain, a.d + o_chain, total - o_chain, hipMemcpyDeviceToHost, slot.stream));
        MCD_WAIT(cat->ctx, slot.stream, cat->spin_us, "mcd_stretch_move (resident block)");
    } else {
        // events: in[k] the numbers of part k are on the device; rows[k] the chain rows of part k are final; out[k] they
        // are in pinned memory
        while ((int64_t)cat->chain_events.size() < 3 * n_parts) {          // (events are kept for the catalogue's lifetime)
            hipEvent_t e;
            MCD_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            cat->chain_events.push_back(e);
        }
        hipEvent_t* ev_in = cat->chain_events.data();
        hipEvent_t* ev_rows = ev_in + n_parts;
        hipEvent_t* ev_out = ev_rows + n_parts;
        auto rows_to_host = [&](int64_t k) -> int {            // D2H of part k's rows on the copy stream, behind rows[k]
            const size_t at = (size_t)part_begin(k) * BW, n = (size_t)(part_begin(k + 1) - part_begin(k)) * BW;
            MCD_HIP(hipStreamWaitEvent(slot.comm_stream, ev_rows[k], 0));
            if (chain) MCD_HIP(hipMemcpyAsync(a.h + o_chain + at * P * 8, a.d + o_chain + at * P * 8, n * P * 8, hipMemcpyDeviceToHost, slot.comm_stream));
            if (lnprob_chain) MCD_HIP(hipMemcpyAsync(a.h + o_lnpc + at * 8, a.d + o_lnpc + at * 8, n * 8, hipMemcpyDeviceToHost, slot.comm_stream));
            MCD_HIP(hipEventRecord(ev_out[k], slot.comm_stream));
            return MCD_OK;
        };
        auto rows_to_caller = [&](int64_t k) -> int {          // wait for out[k], pinned -> the caller's arrays
            const size_t at = (size_t)part_begin(k) * BW, n = (size_t)(part_begin(k + 1) - part_begin(k)) * BW;
            // (the event is the last thing on the copy stream at this point: wait for the stream, with the context's deadline)
            MCD_WAIT(cat->ctx, slot.comm_stream, cat->spin_us, "mcd_stretch_move (chain rows of a part)");
            if (chain) big_copy(chain + at * P, a.h + o_chain + at * P * 8, n * P * 8);
            if (lnprob_chain) big_copy(lnprob_chain + at, a.h + o_lnpc + at * 8, n * 8);
            return MCD_OK;
        };
        MCD_HIP(hipMemcpyAsync(a.d, a.h, o_order, hipMemcpyHostToDevice, slot.stream));   // state, column map, bounds
        for (int64_t k = 0; k < n_parts; ++k) {
            const int64_t i0 = part_begin(k), i1 = part_begin(k + 1);
            const size_t at = (size_t)i0 * BW, n = (size_t)(i1 - i0) * BW;
            if (gen_device) {
                MCD_HIP(generate(slot.comm_stream, i0, i1));
                MCD_HIP(hipEventRecord(ev_in[k], slot.comm_stream));
                MCD_HIP(hipStreamWaitEvent(slot.stream, ev_in[k], 0));
            } else {
                copy_in(i0, i1);
                MCD_HIP(hipMemcpyAsync(a.d + o_order + at * 4, a.h + o_order + at * 4, n * 4, hipMemcpyHostToDevice, slot.comm_stream));
                MCD_HIP(hipMemcpyAsync(a.d + o_zz + at * 8, a.h + o_zz + at * 8, n * 8, hipMemcpyHostToDevice, slot.comm_stream));
                MCD_HIP(hipMemcpyAsync(a.d + o_thr + at * 8, a.h + o_thr + at * 8, n * 8, hipMemcpyHostToDevice, slot.comm_stream));
                MCD_HIP(hipMemcpyAsync(a.d + o_pick + at * 4, a.h + o_pick + at * 4, n * 4, hipMemcpyHostToDevice, slot.comm_stream));
                MCD_HIP(hipEventRecord(ev_in[k], slot.comm_stream));
                MCD_HIP(hipStreamWaitEvent(slot.stream, ev_in[k], 0));
            }
            rc = enqueue_steps(i0, i1, k > 0 ? ev_rows[k - 1] : nullptr);
            if (rc != MCD_OK) return rc;
            if (k > 0) {
                rc = rows_to_host(k - 1);
                if (rc != MCD_OK) return rc;
                rc = rows_to_caller(k - 1);                     // (the device is busy with part k meanwhile)
                if (rc != MCD_OK) return rc;
            }
        }
        rc = finish();
        if (rc != MCD_OK) return rc;
        MCD_HIP(hipEventRecord(ev_rows[n_parts - 1], slot.stream));
        MCD_HIP(hipMemcpyAsync(a.h, a.d, state_end, hipMemcpyDeviceToHost, slot.stream));
        rc = rows_to_host(n_parts - 1);
        if (rc != MCD_OK) return rc;
        rc = rows_to_caller(n_parts - 1);
        if (rc != MCD_OK) return rc;
        MCD_WAIT(cat->ctx, slot.stream, cat->spin_us, "mcd_stretch_move (resident block, last part)");
        MCD_WAIT(cat->ctx, slot.comm_stream, cat->spin_us, "mcd_stretch_move (resident block, chain rows)");
        rows_delivered = true;   // (a discarded block leaves garbage there: the host-driven re-run overwrites every row)
    }
#ifdef MCD_CHAIN_STAMPS
    {
        const int64_t n = std::min<int64_t>(n_steps * 2, 4096);
        std::vector<unsigned long long> st((size_t)n * 8);
        MCD_HIP(hipMemcpy(st.data(), d_stamps, st.size() * 8, hipMemcpyDeviceToHost));
        double sum[8] = {0}, period = 0;
        int64_t cnt = 0;
        for (int64_t i = 2; i + 1 < n; ++i) {
            for (int k = 1; k < 8; ++k) sum[k] += (double)(st[i * 8 + k] - st[i * 8 + k - 1]);
            period += (double)(st[(i + 1) * 8] - st[i * 8]);
            ++cnt;
        }
        if (cnt > 0) {
            fprintf(stderr, "[chain stamps] period %.2f us; phases (us):", period / cnt / 100.0);
            for (int k = 1; k < 8; ++k) fprintf(stderr, " %d:%.2f", k, sum[k] / cnt / 100.0);
            fprintf(stderr, "\n");
        }
    }
#endif

    mid_block.ok = true;
    w.fast = level;
    cat->cur_walkers = half;
    cat->last_chunks = w.n_chunks;
    {
        const int64_t slots = mcd::partial_slots(shape, w.n_chunks, half);
        cat->last_grid = slots != w.n_chunks ? slots : mcd::main_grid(w.n_chunks, half);
    }
    const int32_t* meta = (const int32_t*)(a.h + o_meta);
    const bool discard = meta[mcd::META_STATUS] != 0 || (coll && *(const double*)(a.h + o_status) != 0.0);
    if (discard) {
        ++cat->chain_discarded;
        cat->chain_last_status = meta[mcd::META_STATUS];
        cat->chain_hint = (meta[mcd::META_STATUS] & mcd::CHAIN_LEVEL) ? meta[mcd::META_LEVEL] : -1;
        if (defer && (meta[mcd::META_STATUS] & mcd::CHAIN_LEVEL)) {      // the first launch whose verdict differed
            const int32_t* levels = (const int32_t*)(a.h + o_levels);
            cat->chain_hint = -1;
            for (size_t l = 0; l < n_launches; ++l)
                if (levels[l] >= 0 && levels[l] != level) { cat->chain_hint = levels[l]; break; }
        }
        // every rank of a job counts the same discards (the verdict above is collective), so the back-off stays in step
        cat->chain_consecutive = std::min(cat->chain_consecutive + 1, 7);
        cat->chain_backoff = cat->chain_consecutive > 1 ? ((int64_t)1 << (cat->chain_consecutive - 1)) : 0;
        return MCD_OK;
    }
    cat->chain_consecutive = 0;
    cat->chain_hint = level;
    ++cat->chain_device_blocks;
    std::memcpy(pos, a.h + o_pos, BW * P * 8);
    std::memcpy(lnp, a.h + o_lnp, BW * 8);
    if (accepted) std::memcpy(accepted, a.h + o_acc, BW * 8);
    if (chain && !rows_delivered) big_copy(chain, a.h + o_chain, (size_t)n_steps * BW * P * 8);
    if (lnprob_chain && !rows_delivered) big_copy(lnprob_chain, a.h + o_lnpc, (size_t)n_steps * BW * 8);
    *done = true;
    return MCD_OK;
}

int make_slot(int device, DeviceSlot* slot) {
    MCD_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    MCD_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MCD_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    slot->device = device;
    MCD_HIP(hipStreamCreateWithFlags(&slot->stream, hipStreamNonBlocking));
    // the communication stream gets the highest priority: its one small all-reduce kernel per step should take the next
    // free CU slots while the following step's main kernel (thousands of queued workgroups) is being dispatched
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_greatest = 0;
    MCD_HIP(hipStreamCreateWithPriority(&slot->comm_stream, hipStreamNonBlocking, prio_greatest));
    MCD_HIP(hipStreamCreateWithFlags(&slot->stream2, hipStreamNonBlocking));
    return MCD_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

const char* mcd_last_error(void) { return g_last_error.c_str(); }
int mcd_abi_version(void) { return MCD_ABI_VERSION; }

int mcd_ctx_create(int n_dev, const int* dev_ids, mcd_ctx** out) {
    try {
    if (!out || n_dev <= 0) return fail(MCD_ERR_INVALID, "mcd_ctx_create: bad arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(MCD_ERR_NO_DEVICE, "no HIP device visible");
    // MCD_ALLOW_SHARED_DEVICE=1 (testing aid, with MCD_RCCL_LIBRARY): several shards may sit on the same device, each with
    // its own streams -- real RCCL refuses that, the stand-in of tests/fake_rccl does not
    const char* shared_dev = std::getenv("MCD_ALLOW_SHARED_DEVICE");
    const bool allow_shared = shared_dev && shared_dev[0] == '1' && dev_ids != nullptr;
    if (n_dev > count && !allow_shared) return fail(MCD_ERR_NO_DEVICE, "more devices requested than visible");
    if (dev_ids)
        for (int i = 0; i < n_dev; ++i)
            if (dev_ids[i] < 0 || dev_ids[i] >= count) return fail(MCD_ERR_NO_DEVICE, "device index out of range");
    std::unique_ptr<mcd_ctx, int (*)(mcd_ctx*)> ctx(new (std::nothrow) mcd_ctx(), &mcd_ctx_destroy);   // streams / communicators released on every error path
    if (!ctx) return fail(MCD_ERR_INVALID, "out of memory");
    ctx->slots.resize(n_dev);
    if (const char* t = std::getenv("MCD_COLLECTIVE_TIMEOUT_MS")) ctx->collective_timeout_ms = std::max<long long>(0, std::atoll(t));
    std::vector<int> ids(n_dev);
    for (int i = 0; i < n_dev; ++i) {
        ids[i] = dev_ids ? dev_ids[i] : i;
        int rc = make_slot(ids[i], &ctx->slots[i]);
        if (rc != MCD_OK) return rc;
    }
    // MCD_FORCE_RCCL=1: also a one-device context gets its communicator from ncclCommInitAll and all-reduces inside
    // ncclGroupStart/End, so that a single-GPU box runs the call sequence of the multi-device mode
    const char* force = std::getenv("MCD_FORCE_RCCL");
    ctx->force_collective = n_dev == 1 && force && force[0] == '1';
    if (n_dev > 1 || ctx->force_collective) {
        int rc = load_rccl();
        if (rc != MCD_OK) return rc;
        std::vector<ncclComm_t> comms(n_dev);
        MCD_NCCL(g_rccl.CommInitAll(comms.data(), n_dev, ids.data()));
        for (int i = 0; i < n_dev; ++i) ctx->slots[i].comm = comms[i];
    }
    ctx->rank = 0;
    ctx->n_ranks = 1;
    ctx->multi_process = false;
    *out = ctx.release();
    return MCD_OK;
    } catch (...) { return on_exception("mcd_ctx_create"); }
}

int mcd_get_unique_id(void* out_id) {
    try {
    if (!out_id) return fail(MCD_ERR_INVALID, "null id buffer");
    static_assert(sizeof(ncclUniqueId) <= MCD_UNIQUE_ID_BYTES, "unique id does not fit");
    int rc = load_rccl();
    if (rc != MCD_OK) return rc;
    ncclUniqueId id;
    MCD_NCCL(g_rccl.GetUniqueId(&id));
    std::memset(out_id, 0, MCD_UNIQUE_ID_BYTES);
    std::memcpy(out_id, &id, sizeof id);
    return MCD_OK;
    } catch (...) { return on_exception("mcd_get_unique_id"); }
}

int mcd_ctx_create_rank(int device, int rank, int n_ranks, const void* unique_id, mcd_ctx** out) {
    try {
    if (!out || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return fail(MCD_ERR_INVALID, "mcd_ctx_create_rank: bad arguments");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(MCD_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= count) return fail(MCD_ERR_NO_DEVICE, "device index out of range");
    std::unique_ptr<mcd_ctx, int (*)(mcd_ctx*)> ctx(new (std::nothrow) mcd_ctx(), &mcd_ctx_destroy);   // streams / communicators released on every error path
    if (!ctx) return fail(MCD_ERR_INVALID, "out of memory");
    ctx->slots.resize(1);
    if (const char* t = std::getenv("MCD_COLLECTIVE_TIMEOUT_MS")) ctx->collective_timeout_ms = std::max<long long>(0, std::atoll(t));
    int rc = make_slot(device, &ctx->slots[0]);
    if (rc != MCD_OK) return rc;
    const char* force = std::getenv("MCD_FORCE_RCCL");
    ctx->force_collective = force && force[0] == '1' && unique_id;
    if (n_ranks > 1 || ctx->force_collective) {
        if (!unique_id) return fail(MCD_ERR_INVALID, "unique_id required when n_ranks > 1");
        rc = load_rccl();
        if (rc != MCD_OK) return rc;
        ncclUniqueId id;
        std::memcpy(&id, unique_id, sizeof id);
        MCD_NCCL(g_rccl.CommInitRank(&ctx->slots[0].comm, n_ranks, id, rank));
    }
    ctx->rank = rank;
    ctx->n_ranks = n_ranks;
    ctx->multi_process = true;
    *out = ctx.release();
    return MCD_OK;
    } catch (...) { return on_exception("mcd_ctx_create_rank"); }
}

int mcd_ctx_destroy(mcd_ctx* ctx) {
    if (!ctx) return MCD_OK;
    // a failed context has streams blocked behind a collective that will never finish: destroying them or the
    // communicator would block this thread as well.  The handles are abandoned; the process is about to exit.
    if (ctx->failed.load()) { delete ctx; return MCD_OK; }
    for (DeviceSlot& s : ctx->slots) {
        (void)hipSetDevice(s.device);
        if (s.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(s.comm);
        if (s.stream) (void)hipStreamDestroy(s.stream);
        if (s.comm_stream) (void)hipStreamDestroy(s.comm_stream);
        if (s.stream2) (void)hipStreamDestroy(s.stream2);
    }
    delete ctx;
    return MCD_OK;
}

int mcd_ctx_n_devices(const mcd_ctx* ctx) { return ctx ? (int)ctx->slots.size() : 0; }

int mcd_ctx_set_option(mcd_ctx* ctx, const char* key, int64_t value) {
    try {
    if (!ctx || !key) return fail(MCD_ERR_INVALID, "mcd_ctx_set_option: null argument");
    if (!std::strcmp(key, "collective_timeout_ms")) {
        if (value < 0) return fail(MCD_ERR_INVALID, "collective_timeout_ms must be >= 0 (0: wait for ever)");
        ctx->collective_timeout_ms = value;
        return MCD_OK;
    }
    return fail(MCD_ERR_INVALID, std::string("unknown context option: ") + key);
    } catch (...) { return on_exception("mcd_ctx_set_option"); }
}

int mcd_ctx_abort(mcd_ctx* ctx, const char* reason) {
    try {
    if (!ctx) return fail(MCD_ERR_INVALID, "mcd_ctx_abort: null context");
    {
        std::lock_guard<std::mutex> lock(ctx->note_mutex);
        if (ctx->abort_reason.empty() && reason) ctx->abort_reason = reason;
    }
    ctx->abort_flag.store(1, std::memory_order_release);
    return MCD_OK;
    } catch (...) { return on_exception("mcd_ctx_abort"); }
}

int mcd_ctx_failed(const mcd_ctx* ctx) { return ctx && ctx->failed.load() ? 1 : 0; }

int mcd_ctx_comm_info(const mcd_ctx* ctx, int* comm_size, int* comm_rank, int* rccl_version) {
    try {
    if (!ctx || ctx->slots.empty()) return fail(MCD_ERR_INVALID, "mcd_ctx_comm_info: null context");
    if (comm_size) *comm_size = 0;
    if (comm_rank) *comm_rank = -1;
    if (rccl_version) *rccl_version = 0;
    const ncclComm_t comm = ctx->slots[0].comm;
    if (!comm) return MCD_OK;                       // single device, RCCL never loaded
    int n = 0, r = -1, v = 0;
    MCD_NCCL(g_rccl.CommCount(comm, &n));
    MCD_NCCL(g_rccl.CommUserRank(comm, &r));
    MCD_NCCL(g_rccl.GetVersion(&v));
    if (comm_size) *comm_size = n;
    if (comm_rank) *comm_rank = r;
    if (rccl_version) *rccl_version = v;
    return MCD_OK;
    } catch (...) { return on_exception("mcd_ctx_comm_info"); }
}

static int catalog_create_impl(mcd_ctx* ctx, const mcd_catalog_desc* d, std::unique_ptr<mcd_catalog>& cat);

int mcd_catalog_create(mcd_ctx* ctx, const mcd_catalog_desc* d, mcd_catalog** out) {
    try {
    if (!ctx || !d || !out) return fail(MCD_ERR_INVALID, "mcd_catalog_create: null argument");
    std::unique_ptr<mcd_catalog> cat;
    const int rc = catalog_create_impl(ctx, d, cat);
    if (rc != MCD_OK) {
        if (cat) {                                   // release whatever device memory was already allocated
            const std::string msg = g_last_error;
            mcd_catalog_destroy(cat.release());
            g_last_error = msg;
        }
        return rc;
    }
    *out = cat.release();
    return MCD_OK;
    } catch (...) { return on_exception("mcd_catalog_create"); }
}

static int catalog_create_impl(mcd_ctx* ctx, const mcd_catalog_desc* d, std::unique_ptr<mcd_catalog>& cat) {
    if (int urc = ctx_usable(ctx)) return urc;
    if (d->n_stars < 0) return fail(MCD_ERR_INVALID, "negative n_stars");
    if (d->model < 0 || d->model >= mcd::kNumModels) return fail(MCD_ERR_INVALID, "unknown model");
    const int bgk = mcd::bg_kind(d->model);
    if (d->centre != MCD_CENTRE_FIXED && d->centre != MCD_CENTRE_FREE) return fail(MCD_ERR_INVALID, "unknown centre mode");
    if (d->precision < MCD_F64 || d->precision > MCD_F32_ACC64) return fail(MCD_ERR_INVALID, "unknown precision");
    if (d->n_stars > 0 && (!d->ra || !d->dec || !d->v || !d->verr)) return fail(MCD_ERR_INVALID, "missing ra/dec/v/verr column");
    if (bgk == mcd::BG_FIXED && d->n_stars > 0 && (!d->lnlike_bg || !d->pmember))
        return fail(MCD_ERR_INVALID, "background model needs lnlike_bg and pmember columns");
    if (bgk == mcd::BG_GAUSS && d->n_stars > 0 && !d->density)
        return fail(MCD_ERR_INVALID, "Gaussian-background model needs the density column");
    if (bgk == mcd::BG_FIXED_DENSITY && d->n_stars > 0 && (!d->lnlike_bg || !d->density))
        return fail(MCD_ERR_INVALID, "constant-background model needs lnlike_bg and density columns");

    cat.reset(new (std::nothrow) mcd_catalog());
    if (!cat) return fail(MCD_ERR_INVALID, "out of memory");
    cat->ctx = ctx;
    cat->model = d->model;
    cat->free_centre = d->centre == MCD_CENTRE_FREE;
    cat->precision = d->precision;
    cat->k = param_count(d->model, cat->free_centre);
    cat->n_stars = d->n_stars;
    if (const char* tw = std::getenv("MCD_TARGET_WAVES")) {
        long v = std::atol(tw);
        if (v > 0) cat->target_waves = v;
    }
    if (d->n_bins > 1) {
        if (!d->bin_offsets) return fail(MCD_ERR_INVALID, "bin_offsets required when n_bins > 1");
        cat->n_psets = d->n_bins;
        cat->bin_offsets.assign(d->bin_offsets, d->bin_offsets + d->n_bins + 1);
        if (cat->bin_offsets.front() != 0 || cat->bin_offsets.back() != d->n_stars)
            return fail(MCD_ERR_INVALID, "bin_offsets must start at 0 and end at n_stars");
        for (int64_t b = 0; b < d->n_bins; ++b)
            if (cat->bin_offsets[b + 1] < cat->bin_offsets[b]) return fail(MCD_ERR_INVALID, "bin_offsets must be non-decreasing");
    } else {
        cat->n_psets = 1;
        cat->bin_offsets = {0, d->n_stars};
    }

    // range statistics for the fast-path guard
    cat->stats = mcd::compute_stats(d->n_stars, d->v, d->verr, d->lnlike_bg, d->pmember, d->density, bgk,
                                    cat->precision != MCD_F64 || mcd::is_profile(cat->model) ? d->ra : nullptr, d->dec,
                                    !cat->free_centre, d->ra_center, d->dec_center);

    // contiguous star shards, one per device of this process
    const int n_dev = (int)ctx->slots.size();
    cat->shards.resize(n_dev);
    const int rec_bytes = mcd::record_bytes(cat->model, cat->free_centre, cat->precision);
    for (int i = 0; i < n_dev; ++i) {
        Shard& sh = cat->shards[i];
        sh.slot = i;
        const mcd::ShardRange range = mcd::shard_range(d->n_stars, i, n_dev);
        sh.star_begin = range.begin;
        sh.n = range.n;
        const DeviceSlot& slot = ctx->slots[i];
        MCD_HIP(hipSetDevice(slot.device));
        MCD_HIP(hipEventCreate(&sh.ev_begin));
        MCD_HIP(hipEventCreate(&sh.ev_k0));
        MCD_HIP(hipEventCreate(&sh.ev_k1));
        MCD_HIP(hipEventCreate(&sh.ev_end));
        // slack: wide scalar loads and the software prefetch of the following loop iterations (mcd_math.h: RecordPrefetch)
        // read up to 1.5 KiB past a chunk's last record
        MCD_HIP(hipMalloc(&sh.records, (size_t)sh.n * rec_bytes + 2048));
        MCD_HIP(hipMemsetAsync(sh.records, 0, (size_t)sh.n * rec_bytes + 2048, slot.stream));
        if (bgk == mcd::BG_FIXED || bgk == mcd::BG_FIXED_DENSITY) {
            const std::vector<double> sums = mcd::pset_background_sums(d->lnlike_bg, cat->bin_offsets, sh.star_begin, sh.n);
            MCD_HIP(hipMalloc(&sh.d_pset_const, sums.size() * sizeof(double)));
            MCD_HIP(hipMemcpy(sh.d_pset_const, sums.data(), sums.size() * sizeof(double), hipMemcpyHostToDevice));
        }
        if (sh.n == 0) continue;
        // raw columns -> device scratch -> packed records (device-side trig), scratch freed afterwards
        const double* host_cols[7] = {d->ra, d->dec, d->v, d->verr, d->lnlike_bg, d->pmember, d->density};
        struct Scratch {                                    // freed on every exit path
            double* p[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
            ~Scratch() { for (double* q : p) if (q) (void)hipFree(q); }
        } dev;
        for (int c = 0; c < 7; ++c) {
            if (!host_cols[c]) continue;
            MCD_HIP(hipMalloc(&dev.p[c], (size_t)sh.n * sizeof(double)));
            MCD_HIP(hipMemcpyAsync(dev.p[c], host_cols[c] + sh.star_begin, (size_t)sh.n * sizeof(double),
                                   hipMemcpyHostToDevice, slot.stream));
        }
        mcd::RawColumns raw{dev.p[0], dev.p[1], dev.p[2], dev.p[3], dev.p[4], dev.p[5], dev.p[6]};
        MCD_HIP(mcd::launch_prepare_records(slot.stream, raw, sh.n, cat->model, cat->free_centre, cat->precision,
                                            d->ra_center, d->dec_center, sh.records));
        MCD_HIP(hipStreamSynchronize(slot.stream));
    }
    return MCD_OK;
}

int mcd_catalog_destroy(mcd_catalog* cat) {
    if (!cat) return MCD_OK;
    if (cat->ctx && cat->ctx->failed.load()) { delete cat; return MCD_OK; }      // (see mcd_ctx_destroy: nothing may be waited for)
    for (Shard& sh : cat->shards) {
        (void)hipSetDevice(cat->ctx->slots[sh.slot].device);
        (void)hipStreamSynchronize(cat->ctx->slots[sh.slot].stream);
        (void)hipStreamSynchronize(cat->ctx->slots[sh.slot].comm_stream);
        (void)hipStreamSynchronize(cat->ctx->slots[sh.slot].stream2);
        for (auto& kv : sh.work) free_workset(kv.second);
        if (sh.records) (void)hipFree(sh.records);
        if (sh.d_pset_const) (void)hipFree(sh.d_pset_const);
        if (sh.ev_begin) (void)hipEventDestroy(sh.ev_begin);
        if (sh.ev_k0) (void)hipEventDestroy(sh.ev_k0);
        if (sh.ev_k1) (void)hipEventDestroy(sh.ev_k1);
        if (sh.ev_end) (void)hipEventDestroy(sh.ev_end);
        for (auto& pr : sh.ring) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    }
    for (hipEvent_t e : cat->chain_events) (void)hipEventDestroy(e);
    if (cat->chain.d) (void)hipFree(cat->chain.d);
    if (cat->chain.h) (void)hipHostFree(cat->chain.h);
    delete cat;
    return MCD_OK;
}

int mcd_catalog_param_count(const mcd_catalog* cat) { return cat ? cat->k : MCD_ERR_INVALID; }
int64_t mcd_catalog_n_stars(const mcd_catalog* cat) { return cat ? cat->n_stars : MCD_ERR_INVALID; }
int64_t mcd_catalog_n_outputs(const mcd_catalog* cat, int64_t n_walkers) {
    return cat ? cat->n_psets * n_walkers : MCD_ERR_INVALID;
}

int mcd_params_upload(mcd_catalog* cat, int64_t n_walkers, int32_t k, const double* params) {
    try { return stage_params(cat, n_walkers, k, params, false); } catch (...) { return on_exception("mcd_params_upload"); }
}

int mcd_loglike_enqueue(mcd_catalog* cat) {
    try { return enqueue(cat, true); } catch (...) { return on_exception("mcd_loglike_enqueue"); }
}
int mcd_loglike_fetch(mcd_catalog* cat, double* out) {
    try { return fetch(cat, out); } catch (...) { return on_exception("mcd_loglike_fetch"); }
}
int mcd_sync(mcd_catalog* cat) {
    try { return sync_all(cat); } catch (...) { return on_exception("mcd_sync"); }
}

int mcd_loglike_batch(mcd_catalog* cat, int64_t n_walkers, int32_t k, const double* params, double* out) {
    try {
    if (!out) return fail(MCD_ERR_INVALID, "null output");
    const bool collective = cat && (cat->ctx->n_ranks > 1 || cat->ctx->slots.size() > 1 || cat->ctx->force_collective);
    int rc = stage_params(cat, n_walkers, k, params, !collective && cat && cat->zero_copy);
    if (rc != MCD_OK) return rc;
    rc = enqueue(cat, false);
    if (rc != MCD_OK) return rc;
    return fetch(cat, out);
    } catch (...) { return on_exception("mcd_loglike_batch"); }
}

namespace {
int per_star(mcd_catalog* cat, int32_t k, const double* params, int mode, double* out) {
    if (!cat || !params || !out) return fail(MCD_ERR_INVALID, "per-star output: null argument");
    if (mcd::bg_kind(cat->model) == mcd::BG_NONE) return fail(MCD_ERR_INVALID, "per-star outputs need a background model");
    if (cat->n_psets != 1) return fail(MCD_ERR_INVALID, "per-star outputs are defined for un-binned catalogues");
    if (k != cat->k) return fail(MCD_ERR_INVALID, "parameter row has the wrong number of columns");
    const size_t term_bytes = cat->precision == MCD_F64 ? 8 : 4;
    for (Shard& sh : cat->shards) {
        if (sh.n == 0) continue;
        const DeviceSlot& slot = cat->ctx->slots[sh.slot];
        MCD_HIP(hipSetDevice(slot.device));
        double* d_p = nullptr; void* d_w = nullptr; double* d_o = nullptr;
        MCD_HIP(hipMalloc(&d_p, k * sizeof(double)));
        MCD_HIP(hipMalloc(&d_w, mcd::KD * term_bytes));
        MCD_HIP(hipMalloc(&d_o, (size_t)sh.n * sizeof(double)));
        MCD_HIP(hipMemcpyAsync(d_p, params, k * sizeof(double), hipMemcpyHostToDevice, slot.stream));
        MCD_HIP(mcd::launch_prepare_walkers(slot.stream, d_p, 1, k, cat->model, cat->free_centre, cat->precision, d_w));
        mcd::LaunchShape shape{cat->model, cat->free_centre, cat->precision, 0};
        MCD_HIP(mcd::launch_per_star(slot.stream, shape, sh.records, sh.n, d_w, mode, d_o));
        MCD_HIP(hipMemcpyAsync(out + sh.star_begin, d_o, (size_t)sh.n * sizeof(double), hipMemcpyDeviceToHost, slot.stream));
        MCD_HIP(hipStreamSynchronize(slot.stream));
        MCD_HIP(hipFree(d_p)); MCD_HIP(hipFree(d_w)); MCD_HIP(hipFree(d_o));
    }
    return MCD_OK;
}
}  // namespace

int mcd_membership(mcd_catalog* cat, int32_t k, const double* params, double* out) {
    try {
    return per_star(cat, k, params, 0, out);
    } catch (...) { return on_exception("mcd_membership"); }
}

int mcd_loglike_per_star(mcd_catalog* cat, int32_t k, const double* params, double* out) {
    try {
    return per_star(cat, k, params, 1, out);
    } catch (...) { return on_exception("mcd_loglike_per_star"); }
}

namespace {
// device scratch of mcd_kde_background, released on every exit path
struct KdeScratch {
    double *comp = nullptr, *v = nullptr, *verr = nullptr, *dmin = nullptr, *sum = nullptr, *out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~KdeScratch() {
        for (double* p : {comp, v, verr, dmin, sum, out})
            if (p) (void)hipFree(p);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};
}  // namespace

int mcd_kde_background(mcd_ctx* ctx, int64_t n_comp, const double* comp, int64_t n, const double* v,
                       const double* verr, double sigma_int, double* out, double* kernel_ms) {
    try {
    if (kernel_ms) *kernel_ms = 0.0;
    if (!ctx || ctx->slots.empty()) return fail(MCD_ERR_INVALID, "kde background: null context");
    if (n < 0 || n_comp < 0) return fail(MCD_ERR_INVALID, "kde background: negative size");
    if (n == 0) return MCD_OK;
    if (n_comp == 0) return fail(MCD_ERR_INVALID, "kde background: no comparison stars");
    if (!comp || !v || !verr || !out) return fail(MCD_ERR_INVALID, "kde background: null argument");
    if (!(sigma_int == sigma_int)) return fail(MCD_ERR_INVALID, "kde background: sigma_int is NaN");
    const DeviceSlot& slot = ctx->slots[0];
    MCD_HIP(hipSetDevice(slot.device));
    int slice_len = 0;
    const int n_slices = mcd::kde_slices(n, n_comp, &slice_len);
    KdeScratch d;
    MCD_HIP(hipMalloc(&d.comp, (size_t)n_comp * sizeof(double)));
    MCD_HIP(hipMalloc(&d.v, (size_t)n * sizeof(double)));
    MCD_HIP(hipMalloc(&d.verr, (size_t)n * sizeof(double)));
    MCD_HIP(hipMalloc(&d.dmin, (size_t)n * n_slices * sizeof(double)));
    MCD_HIP(hipMalloc(&d.sum, (size_t)n * n_slices * sizeof(double)));
    MCD_HIP(hipMalloc(&d.out, (size_t)n * sizeof(double)));
    MCD_HIP(hipEventCreate(&d.e0));
    MCD_HIP(hipEventCreate(&d.e1));
    MCD_HIP(hipMemcpyAsync(d.comp, comp, (size_t)n_comp * sizeof(double), hipMemcpyHostToDevice, slot.stream));
    MCD_HIP(hipMemcpyAsync(d.v, v, (size_t)n * sizeof(double), hipMemcpyHostToDevice, slot.stream));
    MCD_HIP(hipMemcpyAsync(d.verr, verr, (size_t)n * sizeof(double), hipMemcpyHostToDevice, slot.stream));
    MCD_HIP(hipEventRecord(d.e0, slot.stream));
    MCD_HIP(mcd::launch_kde(slot.stream, d.comp, n_comp, d.v, d.verr, n, sigma_int, slice_len, n_slices, d.dmin, d.sum,
                            d.out));
    MCD_HIP(hipEventRecord(d.e1, slot.stream));
    MCD_HIP(hipMemcpyAsync(out, d.out, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, slot.stream));
    MCD_HIP(hipStreamSynchronize(slot.stream));
    if (kernel_ms) {
        float ms = 0.f;
        MCD_HIP(hipEventElapsedTime(&ms, d.e0, d.e1));
        *kernel_ms = ms;
    }
    return MCD_OK;
    } catch (...) { return on_exception("mcd_kde_background"); }
}

namespace {
// common part of mcd_stretch_move / mcd_stretch_move_seeded: argument checks, the resident block, else the host-driven one
int run_stretch(mcd_catalog* cat, const mcd_stretch_desc* d, int64_t n_steps, double* pos, double* lnp,
                const int32_t* order, const double* zz, const double* thr, const int32_t* pick, double* chain,
                double* lnprob_chain, int64_t* accepted, const uint64_t* seed, int64_t step0) {
    const char* who = seed ? "mcd_stretch_move_seeded" : "mcd_stretch_move";
    if (!cat || !d || !pos || !lnp) return fail(MCD_ERR_INVALID, std::string(who) + ": null argument");
    if (!seed && (!order || !zz || !thr || !pick)) return fail(MCD_ERR_INVALID, std::string(who) + ": null argument");
    if (seed && d->n_walkers > mcd::kSeededMaxWalkers)
        return fail(MCD_ERR_INVALID, std::string(who) + ": the generator's ordering keys hold the walker index in 20 bits (n_walkers <= 1048576)");
    const int64_t B = d->n_bins > 1 ? d->n_bins : 1;
    if (B != cat->n_psets)
        return fail(MCD_ERR_INVALID, std::string(who) + ": desc->n_bins must equal the catalogue's number of parameter sets (radial bins)");
    if (d->k != cat->k) return fail(MCD_ERR_INVALID, std::string(who) + ": descriptor has the wrong number of kernel columns");
    if (d->n_walkers <= 0 || (d->n_walkers & 1) || d->n_dim <= 0 || n_steps < 0 || step0 < 0)
        return fail(MCD_ERR_INVALID, std::string(who) + ": n_walkers must be positive and even, n_dim positive, steps non-negative");
    if (!d->col_source || !d->col_const || !d->col_factor || !d->lo || !d->hi) return fail(MCD_ERR_INVALID, std::string(who) + ": null descriptor array");
    for (int c = 0; c < d->k; ++c)
        if (d->col_source[c] >= d->n_dim) return fail(MCD_ERR_INVALID, std::string(who) + ": col_source outside the free parameters");
    const int64_t W = d->n_walkers, half = W / 2;
    if (!seed) {
        // (range checks as min / max reductions: branch-free, vectorised -- a binned block holds millions of indices)
        auto within = [](const int32_t* a, int64_t n, int64_t bound) {
            int32_t lo = 0, hi = 0;
            for (int64_t i = 0; i < n; ++i) { lo = a[i] < lo ? a[i] : lo; hi = a[i] > hi ? a[i] : hi; }
            return lo >= 0 && (int64_t)hi < bound;
        };
        if (!within(order, n_steps * B * W, W)) return fail(MCD_ERR_INVALID, "mcd_stretch_move: order holds an index outside 0..W-1");
        if (!within(pick, n_steps * B * W, half)) return fail(MCD_ERR_INVALID, "mcd_stretch_move: pick holds an index outside the half ensemble");
    }
    mcd::StretchDesc sd;
    sd.n_bins = B;
    sd.n_walkers = W; sd.n_dim = d->n_dim; sd.k = d->k; sd.col_source = d->col_source; sd.col_const = d->col_const;
    sd.col_factor = d->col_factor; sd.lo = d->lo; sd.hi = d->hi; sd.fixed_ok = d->fixed_ok;
    bool done = false;
    const int dev_rc = stretch_block_device(cat, d, n_steps, pos, lnp, order, zz, thr, pick, chain, lnprob_chain, accepted, &done,
                                            seed, step0);
    if (dev_rc != MCD_OK) return dev_rc;
    if (done) return MCD_OK;
    ++cat->chain_host_blocks;
    // host-driven block.  Seeded: the same numbers, from the same functions compiled for the host (mcd_rng.h) -- one step at
    // a time, so that a block of any length needs the numbers of one step only
    std::vector<int32_t> g_order, g_pick;
    std::vector<double> g_zz, g_thr, g_keys;
    std::vector<uint64_t> sorter;
    int eval_rc = MCD_OK;
    auto eval = [&](const double* table, int64_t n, double* out) {
        eval_rc = mcd_loglike_batch(cat, n, d->k, table, out);
        return eval_rc;
    };
    int rc = mcd::STRETCH_OK;
    if (!seed) {
        rc = mcd::stretch_block(sd, n_steps, pos, lnp, order, zz, thr, pick, chain, lnprob_chain, accepted, eval);
    } else {
        g_order.resize((size_t)B * W); g_pick.resize((size_t)2 * B * half); g_zz.resize((size_t)2 * B * half); g_thr.resize((size_t)2 * B * half);
        for (int64_t i = 0; i < n_steps && rc == mcd::STRETCH_OK; ++i) {
            mcd::chain_numbers_of_step(*seed, step0 + i, B, W, d->n_dim, g_order.data(), g_zz.data(), g_thr.data(), g_pick.data(), sorter);
            rc = mcd::stretch_block(sd, 1, pos, lnp, g_order.data(), g_zz.data(), g_thr.data(), g_pick.data(),
                                    chain ? chain + (size_t)i * B * W * d->n_dim : nullptr,
                                    lnprob_chain ? lnprob_chain + (size_t)i * B * W : nullptr, accepted, eval);
        }
    }
    if (rc == mcd::STRETCH_EVAL_FAILED) return eval_rc;                  // message already set by mcd_loglike_batch
    if (rc == mcd::STRETCH_NAN) return fail(MCD_ERR_NONFINITE, std::string(who) + ": the log-likelihood returned NaN");
    if (rc != mcd::STRETCH_OK) return fail(MCD_ERR_INVALID, std::string(who) + ": bad arguments");
    return MCD_OK;
}
}  // namespace

int mcd_stretch_move(mcd_catalog* cat, const mcd_stretch_desc* d, int64_t n_steps, double* pos, double* lnp,
                     const int32_t* order, const double* zz, const double* thr, const int32_t* pick, double* chain,
                     double* lnprob_chain, int64_t* accepted) {
    try {
    return run_stretch(cat, d, n_steps, pos, lnp, order, zz, thr, pick, chain, lnprob_chain, accepted, nullptr, 0);
    } catch (...) { return on_exception("mcd_stretch_move"); }
}

int mcd_stretch_move_seeded(mcd_catalog* cat, const mcd_stretch_desc* d, int64_t n_steps, double* pos, double* lnp,
                            uint64_t seed, int64_t step0, double* chain, double* lnprob_chain, int64_t* accepted) {
    try {
    return run_stretch(cat, d, n_steps, pos, lnp, nullptr, nullptr, nullptr, nullptr, chain, lnprob_chain, accepted, &seed, step0);
    } catch (...) { return on_exception("mcd_stretch_move_seeded"); }
}

int mcd_chain_numbers(uint64_t seed, int64_t step0, int64_t n_steps, int64_t n_bins, int64_t n_walkers, int32_t n_dim,
                      int32_t* order, double* zz, double* thr, int32_t* pick) {
    try {
    if (!order || !zz || !thr || !pick || n_steps < 0 || step0 < 0 || n_walkers <= 0 || (n_walkers & 1) || n_dim <= 0)
        return fail(MCD_ERR_INVALID, "mcd_chain_numbers: bad arguments");
    if (n_walkers > mcd::kSeededMaxWalkers) return fail(MCD_ERR_INVALID, "mcd_chain_numbers: n_walkers <= 1048576");
    const int64_t B = n_bins > 1 ? n_bins : 1, W = n_walkers, half = W / 2;
    std::vector<uint64_t> sorter;
    for (int64_t i = 0; i < n_steps; ++i)
        mcd::chain_numbers_of_step(seed, step0 + i, B, W, n_dim, order + (size_t)i * B * W, zz + (size_t)i * 2 * B * half,
                                   thr + (size_t)i * 2 * B * half, pick + (size_t)i * 2 * B * half, sorter);
    return MCD_OK;
    } catch (...) { return on_exception("mcd_chain_numbers"); }
}

int mcd_set_option(mcd_catalog* cat, const char* key, int64_t value) {
    try {
    if (!cat || !key) return fail(MCD_ERR_INVALID, "mcd_set_option: null argument");
    if (!std::strcmp(key, "timing")) {
        int rc = sync_all(cat);
        if (rc != MCD_OK) return rc;
        cat->timing = value != 0;
        cat->timing_all = value == 2;
        for (Shard& sh : cat->shards) sh.ring_used = 0;
        return MCD_OK;
    }
    if (!std::strcmp(key, "timing_discard")) {
        // forget the event pairs recorded so far without reading them (hipEventElapsedTime over hundreds of pairs takes
        // milliseconds, long enough for an idle GPU to leave its sustained clocks right before a measured region)
        int rc = sync_all(cat);
        if (rc != MCD_OK) return rc;
        for (Shard& sh : cat->shards) sh.ring_used = 0;
        cat->timing_launches = 0;
        return MCD_OK;
    }
    if (!std::strcmp(key, "timing_stride")) {
        if (value < 1) return fail(MCD_ERR_INVALID, "timing_stride must be >= 1");
        int rc = sync_all(cat);
        if (rc != MCD_OK) return rc;
        cat->timing_stride = value;
        cat->timing_launches = 0;
        return MCD_OK;
    }
    if (!std::strcmp(key, "timing_reserve")) {
        // create the per-launch event pairs of "timing" = 2 ahead of a measured loop (hipEventCreate costs microseconds)
        if (value < 0 || value > ((int64_t)1 << 16)) return fail(MCD_ERR_INVALID, "timing_reserve: 0 .. 65536 launches");
        for (Shard& sh : cat->shards) {
            MCD_HIP(hipSetDevice(cat->ctx->slots[sh.slot].device));
            while ((int64_t)sh.ring.size() < value) {
                hipEvent_t a, b;
                MCD_HIP(hipEventCreate(&a));
                if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return fail(MCD_ERR_HIP, "hipEventCreate"); }
                sh.ring.emplace_back(a, b);
            }
        }
        return MCD_OK;
    }
    if (!std::strcmp(key, "fast_path")) {
        if (value < 0 || value > 2) return fail(MCD_ERR_INVALID, "fast_path: 0 (plain), 1 (guarded, default) or 2 (guarded, no narrow variant)");
        cat->allow_fast = (int)value;
        return MCD_OK;
    }
    if (!std::strcmp(key, "zero_copy")) { cat->zero_copy = value != 0; return MCD_OK; }
    if (!std::strcmp(key, "device_chain")) {
        if (value < 0 || value > 2)
            return fail(MCD_ERR_INVALID, "device_chain: 0 (host-driven blocks), 1 (default) or 2 (as 1 with the general step kernel, testing aid)");
        cat->device_chain = (int)value;
        cat->chain_backoff = 0;
        cat->chain_consecutive = 0;
        cat->chain_hint = -1;
        return MCD_OK;
    }
    if (!std::strcmp(key, "fused_reduce")) { cat->fused_reduce = value != 0; return MCD_OK; }
    if (!std::strcmp(key, "defer_guard")) { cat->defer_guard = value != 0; return MCD_OK; }
    if (!std::strcmp(key, "f32_domain")) { cat->f32_domain = value != 0; return MCD_OK; }
    if (!std::strcmp(key, "two_lanes")) {
        int rc = sync_all(cat);
        if (rc != MCD_OK) return rc;
        cat->two_lanes = value != 0;
        return MCD_OK;
    }
    if (!std::strcmp(key, "prefetch")) {
        if (value < -1 || value > 1) return fail(MCD_ERR_INVALID, "prefetch: -1 (by record volume, default), 0 (off) or 1 (on)");
        cat->prefetch = (int)value;
        return MCD_OK;
    }
    if (!std::strcmp(key, "spin_us")) {
        if (value < 0) return fail(MCD_ERR_INVALID, "spin_us must be >= 0");
        cat->spin_us = value;
        return MCD_OK;
    }
    if (!std::strcmp(key, "tail_split") || !std::strcmp(key, "target_waves") || !std::strcmp(key, "chunk_len") ||
        !std::strcmp(key, "balance") || !std::strcmp(key, "combine")) {
        const bool is_split = !std::strcmp(key, "tail_split"), is_len = !std::strcmp(key, "chunk_len");
        const bool is_bal = !std::strcmp(key, "balance"), is_comb = !std::strcmp(key, "combine");
        if (!std::strcmp(key, "target_waves") && value <= 0) return fail(MCD_ERR_INVALID, "target_waves must be positive");
        if (is_len && value < 0) return fail(MCD_ERR_INVALID, "chunk_len must be >= 0");
        if (is_bal && (value < -1 || value > 8)) return fail(MCD_ERR_INVALID, "balance: -1 (auto), 0 (off) or 1 .. 8 workgroups per CU");
        if (is_comb && value != 0 && value != 1 && value != 8 && value != 16)
            return fail(MCD_ERR_INVALID, "combine: 0 (never), 1 (largest workgroup the plan allows), 8 or 16 (waves per workgroup at most)");
        int rc = sync_all(cat);
        if (rc != MCD_OK) return rc;
        if (is_split) cat->tail_split = (int)value;
        else if (is_len) cat->chunk_len = value;
        else if (is_bal) cat->balance = (int)value;
        else if (is_comb) cat->combine = (int)value;
        else cat->target_waves = value;
        for (Shard& sh : cat->shards) {            // chunk tables depend on it: rebuild lazily
            (void)hipSetDevice(cat->ctx->slots[sh.slot].device);
            for (auto& kv : sh.work) free_workset(kv.second);
            sh.work.clear();
        }
        cat->cur_walkers = 0;
        return MCD_OK;
    }
    return fail(MCD_ERR_INVALID, std::string("unknown option: ") + key);
    } catch (...) { return on_exception("mcd_set_option"); }
}

int mcd_timing_collect(mcd_catalog* cat, double* total_kernel_ms, int64_t* n_launches) {
    try {
    if (!cat) return fail(MCD_ERR_INVALID, "null catalogue");
    int rc = sync_all(cat);
    if (rc != MCD_OK) return rc;
    Shard& sh = cat->shards[0];
    double total = 0.0;
    for (size_t i = 0; i < sh.ring_used; ++i) {
        float ms = 0.f;
        MCD_HIP(hipEventElapsedTime(&ms, sh.ring[i].first, sh.ring[i].second));
        total += ms;
    }
    if (total_kernel_ms) *total_kernel_ms = total;
    if (n_launches) *n_launches = (int64_t)sh.ring_used;
    for (Shard& s2 : cat->shards) s2.ring_used = 0;
    cat->timing_launches = 0;                  // the first launch after a collect is sampled
    return MCD_OK;
    } catch (...) { return on_exception("mcd_timing_collect"); }
}

int64_t mcd_rerun_count(const mcd_catalog* cat) { return cat ? cat->n_reruns : MCD_ERR_INVALID; }

int mcd_stretch_info(const mcd_catalog* cat, int64_t* device_blocks, int64_t* host_blocks, int64_t* discarded_blocks,
                     int32_t* last_discard_status) {
    if (!cat) return fail(MCD_ERR_INVALID, "null catalogue");
    if (device_blocks) *device_blocks = cat->chain_device_blocks;
    if (host_blocks) *host_blocks = cat->chain_host_blocks;
    if (discarded_blocks) *discarded_blocks = cat->chain_discarded;
    if (last_discard_status) *last_discard_status = cat->chain_last_status;
    return MCD_OK;
}

int mcd_last_prefetch(const mcd_catalog* cat) { return cat ? cat->last_prefetch : -1; }

int mcd_last_f32_domain(const mcd_catalog* cat, double* kappa_v, double* kappa_theta) {
    if (!cat) return -1;
    if (kappa_v) *kappa_v = cat->last_f32.kappa_v;
    if (kappa_theta) *kappa_theta = cat->last_f32.kappa_theta;
    if (cat->precision == MCD_F64) return 1;
    return cat->last_f32.inside ? 1 : 0;
}

int mcd_last_fast_level(const mcd_catalog* cat) {
    if (!cat || cat->cur_walkers <= 0 || cat->shards.empty()) return -1;
    const auto it = cat->shards.front().work.find(cat->cur_walkers);
    return it == cat->shards.front().work.end() ? -1 : it->second.fast;
}

double mcd_last_kernel_ms(const mcd_catalog* cat) { return cat ? cat->last_kernel_ms : -1.0; }
double mcd_last_device_ms(const mcd_catalog* cat) { return cat ? cat->last_device_ms : -1.0; }

int mcd_last_launch_info(const mcd_catalog* cat, int64_t* n_workgroups, int32_t* walker_tile, int64_t* n_chunks,
                         int32_t* record_bytes) {
    if (!cat) return fail(MCD_ERR_INVALID, "null catalogue");
    if (n_workgroups) *n_workgroups = cat->last_grid;
    if (walker_tile) *walker_tile = 64;
    if (n_chunks) *n_chunks = cat->last_chunks;
    if (record_bytes) *record_bytes = mcd::record_bytes(cat->model, cat->free_centre, cat->precision);
    return MCD_OK;
}

}  // extern "C"
