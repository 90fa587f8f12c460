// mcd_internal.h -- declarations shared by the kernels (mcd_kernels.hip) and the C-ABI (mcd_api.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "mcd_chunks.h"   // Chunk, chunk-table planning (host-only, shared with the CPU tests)

namespace mcd {

struct LaunchShape {
    int model;        // mcd::Model
    bool free_centre;
    int precision;    // mcd_precision
    int fast;         // 0 plain; 1 product/fraction-tree path (range guard passed); 2 narrow-range BGFIXED variant
    int uniform_len = 0;     // > 0: chunk c covers records [c * len, min((c + 1) * len, n_records)) of parameter set 0
    int64_t n_records = 0;
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

// out[pset][w] = sum over the chunks of pset of partials[w / 8][chunk][w % 8]  (fixed order) [+ pset_const[pset]];
// `partials` holds roundup64(n_walkers) x n_chunks doubles
hipError_t launch_reduce(hipStream_t s, const double* partials, const int64_t* pset_chunk_offsets,
                         int64_t n_psets, int64_t n_chunks, int64_t max_chunks_per_pset, int64_t n_walkers,
                         const double* pset_const, double* out);

// per-star outputs for one parameter row: mode 0 membership probability, mode 1 mixture log-likelihood
hipError_t launch_per_star(hipStream_t s, const LaunchShape& shape, const void* records, int64_t n,
                           const void* wpar_row, int mode, double* out);

int record_bytes(int model, bool free_centre, int precision);

// background.SingleStars (mcd_kde.hip): slice plan and launch.  part_dmin / part_sum hold [n_slices][n] doubles.
int kde_slices(int64_t n, int64_t m, int* slice_len);
hipError_t launch_kde(hipStream_t s, const double* comp, int64_t m, const double* v, const double* verr, int64_t n,
                      double sigma_int, int slice_len, int n_slices, double* part_dmin, double* part_sum,
                      double* out);

}  // namespace mcd
