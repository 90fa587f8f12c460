// mcd_internal.h -- declarations shared by the kernels (mcd_kernels.hip) and the C-ABI (mcd_api.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "mcd_chunks.h"   // Chunk, chunk-table planning (host-only, shared with the CPU tests)
#include "mcd_guard.h"    // StatsScalars: the range guard's catalogue statistics travel to the device by value

namespace mcd {

struct LaunchShape {
    int model;        // mcd::Model
    bool free_centre;
    int precision;    // mcd_precision
    int fast;         // 0 plain; 1 product/fraction-tree path (range guard passed); 2 narrow-range BGFIXED variant
    int uniform_len = 0;     // > 0: arithmetic chunk table of parameter set 0 (mcd_chunks.h: uniform_chunk)
    int64_t n_records = 0;
    int uniform_extra = 0;   // its first `uniform_extra` chunks hold uniform_len + 8 stars
    int waves = 4;           // waves per workgroup of the main kernel: 4, or 8 / 16 = the workgroup combines its chunks' sums
                             // (f64 fast kernels, <= 256 walkers in 1, 2 or 4 tiles; 16 not for the BG_GAUSS models;
                             // partial_slots() below)
    const uint8_t* chunk_general = nullptr;   // fast == 2: chunks that must take the general fast form (hold a star that
                                              // rules out the narrow-range variant, mcd_guard.h: narrow_exception); may be null
    bool prefetch = false;          // software prefetch of the next iteration's records (catalogues beyond the caches)
    double* rerun_flag = nullptr;   // device word the fast mixture kernels set to `launch_tag` in the denormal regime
    double launch_tag = 0.0;
};

// raw per-star columns on the device (float64), consumed once by prepare_records
struct RawColumns {
    const double* ra;
    const double* dec;
    const double* v;
    const double* verr;
    const double* lnbg;
    const double* pmember;
    const double* density;
};

hipError_t launch_prepare_records(hipStream_t s, const RawColumns& raw, int64_t n, int model, bool free_centre,
                                  int precision, double ra_c_deg, double dec_c_deg, void* records);

// params [n_rows][k] (float64, reference order) -> derived walker constants [n_rows][KD] in term precision
hipError_t launch_prepare_walkers(hipStream_t s, const double* params, int64_t n_rows, int k, int model,
                                  bool free_centre, int precision, void* wpar);

hipError_t launch_loglike(hipStream_t s, const LaunchShape& shape, const void* records, const Chunk* chunks,
                          int64_t n_chunks, const void* wpar, double* partials, int64_t n_walkers);

// out[pset][w] = sum over the slots of pset of partials[w / 8][slot][w % 8]  (fixed order) [+ pset_const[pset]];
// `partials` holds roundup64(n_walkers) x n_slots doubles; slots = chunks unless the workgroups combine (partial_slots)
hipError_t launch_reduce(hipStream_t s, const double* partials, const int64_t* pset_slot_offsets,
                         int64_t n_psets, int64_t n_slots, int64_t max_slots_per_pset, int64_t n_walkers,
                         const double* pset_const, double* out);
int64_t partial_slots(const LaunchShape& shape, int64_t n_chunks, int64_t n_walkers);
constexpr int64_t kFusedReduceSlots = 256;   // up to here the reduction is one wave per walker group with one round of loads

// per-star outputs for one parameter row: mode 0 membership probability, mode 1 mixture log-likelihood
hipError_t launch_per_star(hipStream_t s, const LaunchShape& shape, const void* records, int64_t n,
                           const void* wpar_row, int mode, double* out);

int record_bytes(int model, bool free_centre, int precision);

// ---- resident stretch-move chain (mcd_stretch.hip): all pointers are device memory of ONE device ----
enum ChainMeta : int { META_N_OK = 0, META_STATUS = 1, META_LEVEL = 2, META_WORDS = 4 };
enum ChainStatus : int {
    CHAIN_NAN = 1,           // a log-likelihood came back NaN (genuine, or the NaN-poisoned re-run signal of a multi-rank job)
    CHAIN_RERUN = 2,         // a fast mixture kernel asked for the plain kernels (single device: tag behind the sums)
    CHAIN_LEVEL = 4,         // the guard's verdict on a proposed table differs from the kernel family that was enqueued
    CHAIN_NO_PROPOSAL = 8    // a half step without a single proposal inside the prior (the host loop skips the evaluation)
};
struct StretchDevice {
    int64_t n_bins = 1;                    // B ensembles (one per parameter set of the catalogue), one workgroup each
    int64_t n_walkers = 0;                 // W (even), per ensemble
    int32_t n_dim = 0, k = 0;              // free parameters; columns of the kernel parameter table
    int32_t fixed_ok = 1;
    int32_t model = 0, free_centre = 0;
    int32_t allow_fast = 1;                // option "fast_path"
    int32_t expected_level = 0;            // LaunchShape::fast the main kernels of this block are enqueued with
#ifdef MCD_CHAIN_STAMPS
    unsigned long long* stamps = nullptr;  // development aid: 8 timestamps (100 MHz) per step-kernel launch
    long long launch_index = 0;
#endif
    int32_t force_general = 0;             // testing aid (option "device_chain" = 2): the general step kernel for any size
    // fused reduction (launches whose partial sums are few, kFusedReduceSlots): the step kernel adds up the main kernel's
    // partial sums itself instead of reading the sums a reduction kernel left in `ll` -- one kernel less per half step
    int32_t fused = 0;
    const double* partials = nullptr;      // [ceil64(W / 2) / 8][n_slots][8]
    int64_t n_slots = 0;
    const int64_t* slot_offsets = nullptr; // [n_bins + 1] slots of each ensemble (null for one ensemble: 0 .. n_slots)
    const double* pset_const = nullptr;    // walker-independent sum per ensemble (fast fixed-background kernels), or null
    StatsScalars stats;                    // this device's share of the catalogue
    const int32_t* col_source = nullptr;   // [k]   as mcd::StretchDesc (mcd_stretch.h)
    const double* col_const = nullptr;     // [k]
    const double* col_factor = nullptr;    // [k]
    const double* lo = nullptr;            // [P]
    const double* hi = nullptr;            // [P]
    double* pos = nullptr;                 // [W][P]  ensemble, updated in place
    double* lnp = nullptr;                 // [W]
    long long* accepted = nullptr;         // [W]
    const int32_t* order = nullptr;        // [n_steps][W]       the block's random numbers, as mcd_stretch_move takes them
    const double* zz = nullptr;            // [n_steps][2][W/2]
    const double* thr = nullptr;           // [n_steps][2][W/2]
    const int32_t* pick = nullptr;         // [n_steps][2][W/2]
    double* chain = nullptr;               // [n_steps][W][P] or null
    double* lnprob_chain = nullptr;        // [n_steps][W] or null
    double* proposal = nullptr;            // [W/2][P]
    uint8_t* ok = nullptr;                 // [W/2]   proposal inside the prior
    int32_t* meta = nullptr;               // [META_WORDS]
    int32_t* n_ok = nullptr;               // [2][B] proposals inside the prior, per ensemble and half-step parity
    double* ranges = nullptr;              // [2][B][10] ParamRanges of each ensemble's table (B > 1: judged by the next launch)
    double* table = nullptr;               // [W/2][k]   resolved parameter rows (the guard reads them back)
    // deferred guard (ONE ensemble, the in-LDS step kernel): the step kernel logs each launch's rows and the number of
    // proposals inside the prior; stretch_judge_kernel gives all verdicts at the end of the block (mcd_stretch.hip)
    int32_t defer_guard = 0;
    double* table_log = nullptr;           // [2 n_steps][W/2][k]
    int32_t* n_ok_log = nullptr;           // [2 n_steps]
    int32_t* level_log = nullptr;          // [2 n_steps]  verdict per launch, -1 where there was no table
    double* wpar = nullptr;                // [W/2][KD]  derived walker constants: what the main kernel reads
};

hipError_t launch_stretch_step(hipStream_t s, const StretchDevice& d, int64_t acc_step, int acc_h, int64_t prop_step,
                               int prop_h, const double* ll, double rerun_tag);
hipError_t launch_stretch_status(hipStream_t s, const int32_t* meta, double* out);
hipError_t launch_stretch_judge(hipStream_t s, const StretchDevice& d, int64_t n_launches);
// seeded blocks: the numbers of steps [i0, i1) of a block generated on the device, in the layout of the host's upload
// (mcd_rng.h); ensembles whose keys do not fit in LDS get their numbers from the host build of the same functions
bool chain_numbers_on_device(int64_t n_walkers);
hipError_t launch_chain_numbers(hipStream_t s, uint64_t seed, int64_t step0, int64_t i0, int64_t i1, int64_t n_bins,
                                int64_t n_walkers, int n_dim, int32_t* order, double* zz, double* thr, int32_t* pick);
// several ensembles need the step kernel that keeps an ensemble in LDS (<= 512 walkers, <= 12 columns, <= 32 KiB of positions)
bool stretch_step_handles(const StretchDevice& d);
// ... and only that kernel can add up the main kernel's partial sums itself (StretchDevice::fused)
bool stretch_step_fuses(const StretchDevice& d);

// background.SingleStars (mcd_kde.hip): slice plan and launch.  part_dmin / part_sum hold [n_slices][n] doubles.
int kde_slices(int64_t n, int64_t m, int* slice_len);
hipError_t launch_kde(hipStream_t s, const double* comp, int64_t m, const double* v, const double* verr, int64_t n,
                      double sigma_int, int slice_len, int n_slices, double* part_dmin, double* part_sum,
                      double* out);

}  // namespace mcd
