/*
 * mcd.h -- C-ABI of the MI355X-native log-likelihood hot path of mcmc_dynamics.
 *
 * The reference (skamann/mcmc-dynamics, pure Python) has no FFI of its own; its boundary for this
 * path is the bound method `Runner.lnprob` handed to emcee (analysis/runner.py:288-306, :403).
 * The entry points below are what a ctypes/cffi binding inside `Runner` binds to replace the
 * NumPy/astropy body of that method:
 *
 *   mcd_catalog_create   replaces the per-instance column extraction of `Runner.__init__`
 *                        (analysis/runner.py:75-81, :96-106) and the walker-independent part of
 *                        `calc_xy_offset` + `arctan2` (utils/coordinates/calc_xy_offset.py:9-33,
 *                        analysis/constant.py:106-107): star columns are copied to HBM ONCE.
 *   mcd_loglike_batch    replaces `ConstantFit.lnlike` / `ConstantFitGB.lnlike` /
 *                        `Runner._calculate_lnlike` (analysis/constant.py:113-154, :293-364;
 *                        analysis/runner.py:240-286) for W walkers per call.
 *   mcd_membership       replaces `ConstantFitGB.calculate_membership_probabilities`
 *                        (analysis/constant.py:366-374) and the ModelFit variants (analysis/model.py:458-510, 625-687).
 *
 * Conventions
 *   - plain pointers and sizes only; all arrays are float64 unless stated; canonical units of the
 *     reference: deg (ra, dec, centres), km/s (velocities), dimensionless (pmember, density, f_back).
 *   - the caller owns every host buffer; the library copies inputs during the call and never keeps
 *     a host pointer.  `out` buffers are caller-allocated.
 *   - every function returns 0 on success or a negative mcd_status; `mcd_last_error()` returns a
 *     thread-local message.  No C++ exception crosses this boundary.
 *   - results are deterministic: fixed chunking and a fixed reduction tree (no float atomics).
 */
#ifndef MCD_H
#define MCD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCD_ABI_VERSION 1          /* bumped when an existing signature or struct layout changes; additions keep it */
#define MCD_UNIQUE_ID_BYTES 128

typedef struct mcd_ctx mcd_ctx;          /* devices + streams (+ RCCL communicator when > 1 rank) */
typedef struct mcd_catalog mcd_catalog;  /* HBM-resident star records of one Runner instance     */

typedef enum {
    MCD_OK = 0,
    MCD_ERR_INVALID = -1,   /* bad argument (null pointer, size, enum, K mismatch)  */
    MCD_ERR_HIP = -2,       /* a HIP runtime call failed (message has the call)     */
    MCD_ERR_RCCL = -3,      /* an RCCL call failed                                  */
    MCD_ERR_NO_DEVICE = -4, /* no usable gfx950 device                              */
    MCD_ERR_NONFINITE = -5, /* a NaN log-likelihood inside mcd_stretch_move         */
    MCD_ERR_NOMEM = -6      /* host memory exhausted inside the library             */
} mcd_status;

/* which per-star likelihood the catalogue is evaluated with */
typedef enum {
    MCD_MODEL_CONST = 0,          /* ConstantFit, no background  (runner.py:264-271)                 */
    MCD_MODEL_CONST_BGFIXED = 1,  /* ConstantFit + fixed per-star background lnL and pmember
                                     (runner.py:272-286; background/gaussian.py:23-28)              */
    MCD_MODEL_CONST_BGGAUSS = 2,  /* ConstantFitGB: per-walker (v_back, sigma_back, f_back) and the
                                     `density` prior (constant.py:293-364)                          */
    MCD_MODEL_PROFILE = 3,        /* ModelFit: Lynden-Bell rotation curve + Plummer dispersion profile
                                     (analysis/model.py:93-222)                                     */
    MCD_MODEL_PROFILE_BGGAUSS = 4,/* ModelFitGB: + per-walker Gaussian background, density prior
                                     (analysis/model.py:391-456)                                    */
    MCD_MODEL_PROFILE_BGDENS = 5, /* ModelFitConstantBackground: fixed per-star background lnL, density prior
                                     with per-walker f_back (analysis/model.py:565-623)             */
    MCD_MODEL_PROFILE_BGFIXED = 6 /* ModelFit(background=Gaussian / SingleStars): the profiles of MCD_MODEL_PROFILE
                                     with the fixed per-star background lnL and pmember mixture that
                                     Runner._calculate_lnlike applies to every subclass
                                     (analysis/model.py:182-222 -> analysis/runner.py:272-286)      */
} mcd_model;

typedef enum {
    MCD_CENTRE_FIXED = 0,  /* (ra_center, dec_center) fixed: sin/cos(theta_i) precomputed at upload */
    MCD_CENTRE_FREE = 1    /* centre is a walker parameter: geometry recomputed per term            */
} mcd_centre;

typedef enum {
    MCD_F64 = 0,        /* float64 terms, float64 accumulation (parity mode)      */
    MCD_F32 = 1,        /* float32 terms, float32 accumulation                     */
    MCD_F32_ACC64 = 2   /* float32 terms, float64 accumulation                     */
} mcd_precision;

/* Catalogue description.  Pointers not needed by `model` may be NULL. */
typedef struct {
    int64_t n_stars;
    const double* ra;         /* deg  */
    const double* dec;        /* deg  */
    const double* v;          /* km/s */
    const double* verr;       /* km/s */
    const double* lnlike_bg;  /* CONST_BGFIXED, PROFILE_BGFIXED, PROFILE_BGDENS: background(v, verr) per star */
    const double* pmember;    /* CONST_BGFIXED, PROFILE_BGFIXED: prior membership probability      */
    const double* density;    /* *_BGGAUSS, PROFILE_BGDENS: normalised stellar surface density     */
    int32_t model;            /* mcd_model     */
    int32_t centre;           /* mcd_centre    */
    int32_t precision;        /* mcd_precision */
    int32_t reserved;
    double ra_center;         /* deg; used when centre == MCD_CENTRE_FIXED                         */
    double dec_center;        /* deg                                                               */
    int64_t n_bins;           /* 0 or 1: one parameter set for all stars.  B > 1: radial bins
                                 (utils/files/data_reader.py:71-140); stars must be sorted by bin   */
    const int64_t* bin_offsets; /* n_bins + 1 offsets into the star arrays when n_bins > 1        */
} mcd_catalog_desc;

/* ---- context ------------------------------------------------------------------------------ */

/* Single process driving n_dev devices (dev_ids == NULL: devices 0..n_dev-1).  With n_dev > 1 the
 * catalogue is sharded over the devices and per-walker partial sums are combined with one
 * ncclAllReduce(sum, f64, count = outputs) per batched call (communicator from ncclCommInitAll). */
/* MCD_FORCE_RCCL=1 in the environment: a one-device context also creates its communicator with ncclCommInitAll and
 * runs the all-reduce inside ncclGroupStart/End (lets a single-GPU box exercise the call sequence of this mode). */
int mcd_ctx_create(int n_dev, const int* dev_ids, mcd_ctx** out);

/* One process per GPU (torchrun-style).  Rank 0 calls mcd_get_unique_id and distributes the
 * MCD_UNIQUE_ID_BYTES blob out of band; every rank then calls mcd_ctx_create_rank.  Each rank
 * uploads ITS shard of the stars; mcd_loglike_batch all-reduces so every rank gets the total.
 * n_ranks == 1 needs no id (unique_id may be NULL) and never touches RCCL -- unless the environment
 * variable MCD_FORCE_RCCL=1 is set and an id is given: then a 1-rank communicator is created and the
 * all-reduce runs on it (lets a single-GPU box exercise the RCCL call path). */
int mcd_get_unique_id(void* out_id);
int mcd_ctx_create_rank(int device, int rank, int n_ranks, const void* unique_id, mcd_ctx** out);

/* ---- collective deadline (contexts with a communicator; nothing equivalent in the reference, whose only parallelism
 * is the process pool of analysis/runner.py:398-403) -------------------------------------------------------------
 * Every wait on a stream that carries an all-reduce (mcd_loglike_batch / _fetch, mcd_sync, mcd_stretch_move) polls
 * instead of blocking: after `collective_timeout_ms` (option below; default 120000, 0 = wait for ever; the environment
 * variable MCD_COLLECTIVE_TIMEOUT_MS sets the default of new contexts) the call returns MCD_ERR_RCCL with the stage in
 * mcd_last_error(), and the context is marked FAILED: every later call on it (and on its catalogues) returns
 * MCD_ERR_RCCL at once, and the destroy functions no longer synchronise the blocked streams (they would never
 * return) -- the process is expected to report and exit non-zero.  There is no fallback inside the process.
 *
 * mcd_ctx_abort may be called from ANOTHER host thread while a call is waiting (e.g. by the host application's own
 * control channel when a peer rank reports that it failed): the waiting call returns MCD_ERR_RCCL within a
 * millisecond instead of running into the deadline.  This is how a rank that fails in the middle of a block of
 * mcd_stretch_move keeps its peers from waiting inside the collective (mcmc_dynamics_amd/hostgroup.py: abort). */
int mcd_ctx_set_option(mcd_ctx* ctx, const char* key, int64_t value);   /* "collective_timeout_ms" */
int mcd_ctx_abort(mcd_ctx* ctx, const char* reason);                    /* thread-safe; reason may be NULL */
int mcd_ctx_failed(const mcd_ctx* ctx);                                 /* 1 after a deadline / abort / mid-block error */

int mcd_ctx_destroy(mcd_ctx* ctx);
int mcd_ctx_n_devices(const mcd_ctx* ctx);
/* What RCCL itself reports for the communicator of the context's first device: ncclCommCount, ncclCommUserRank and
 * ncclGetVersion (e.g. 22707).  comm_size = 0 / comm_rank = -1 when the context has no communicator (single device:
 * RCCL is never loaded).  A measurement harness echoes these to prove the collective really spans N ranks. */
int mcd_ctx_comm_info(const mcd_ctx* ctx, int* comm_size, int* comm_rank, int* rccl_version);

/* ---- catalogue ---------------------------------------------------------------------------- */

int mcd_catalog_create(mcd_ctx* ctx, const mcd_catalog_desc* desc, mcd_catalog** out);
int mcd_catalog_destroy(mcd_catalog* cat);

/* Number of columns K of the resolved parameter table expected by mcd_loglike_batch:
 *   CONST    : v_sys, sigma_max, v_maxx, v_maxy [, ra_center, dec_center]
 *   *_BGGAUSS: ... + v_back, sigma_back, f_back
 *   PROFILE, PROFILE_BGFIXED: v_sys, sigma_max, a, v_maxx, v_maxy, r_peak [, ra_center, dec_center] (a, r_peak in arcsec)
 *   PROFILE_BGGAUSS: ... + v_back, sigma_back, f_back        PROFILE_BGDENS: ... + f_back
 * (order of config/constant.json:6-11, constant_with_background.json:6-14, model_with_background.json:6-16;
 * config/model.json interleaves the centre between v_maxx and v_maxy -- the host maps columns by name). */
int mcd_catalog_param_count(const mcd_catalog* cat);
int64_t mcd_catalog_n_stars(const mcd_catalog* cat);      /* stars held by THIS process */
int64_t mcd_catalog_n_outputs(const mcd_catalog* cat, int64_t n_walkers); /* W * max(1, n_bins) */

/* ---- evaluation --------------------------------------------------------------------------- */

/* Log-likelihood of W walkers.  params: row-major [max(1,n_bins)][W][K]; out: [max(1,n_bins)][W].
 * Synchronous: H2D of params, kernels, reduction, (all-reduce), D2H of out. */
int mcd_loglike_batch(mcd_catalog* cat, int64_t n_walkers, int32_t k, const double* params, double* out);

/* Device-resident pipeline used by throughput measurements and by callers that keep walkers on the
 * GPU: stage params once, enqueue any number of evaluations, fetch the last result. */
int mcd_params_upload(mcd_catalog* cat, int64_t n_walkers, int32_t k, const double* params);
int mcd_loglike_enqueue(mcd_catalog* cat);                 /* asynchronous on the catalogue's stream(s) */
int mcd_loglike_fetch(mcd_catalog* cat, double* out);      /* waits, copies [max(1,n_bins)][W] doubles    */
int mcd_sync(mcd_catalog* cat);

/* Posterior membership probability per star for ONE parameter row (BGGAUSS models):
 * m e^{lc} / (m e^{lc} + (1 - m) e^{lb}); out has n_stars doubles (this process' shard). */
int mcd_membership(mcd_catalog* cat, int32_t k, const double* params, double* out);
/* Per-star mixture log-likelihood for ONE parameter row, `lnlike(values, no_sum=True)` of
 * ModelFitConstantBackground (analysis/model.py:565-623); defined for every background model. */
int mcd_loglike_per_star(mcd_catalog* cat, int32_t k, const double* params, double* out);

/* Background log-likelihood of n test stars against the kernel-density estimate built from n_comp comparison
 * stars: replaces background.SingleStars.__call__ (background/single_stars.py:42-77), the O(n * n_comp) precompute
 * whose output is the `lnlike_bg` column of a MCD_MODEL_CONST_BGFIXED / _PROFILE_BGDENS catalogue (runner.py:96-106).
 *   out_i = log( (1 / n_comp) sum_j N(v_i - comp_j; verr_i^2 + sigma_int^2) ),   all velocities in km/s.
 * Runs on the first device of `ctx`; host buffers in, n doubles out; synchronous.  n = 0 is a no-op; n_comp = 0 is
 * MCD_ERR_INVALID (the reference raises on the empty maximum). `kernel_ms`, if not NULL, receives the HIP-event time
 * of the two kernels. */
int mcd_kde_background(mcd_ctx* ctx, int64_t n_comp, const double* comp, int64_t n, const double* v,
                       const double* verr, double sigma_int, double* out, double* kernel_ms);

/* ---- sampler support ---------------------------------------------------------------------- */

/* One block of affine-invariant stretch-move steps with the per-half-step host loop inside the library: proposals from
 * the complementary half of the ensemble, box prior, ONE mcd_loglike_batch of n_walkers / 2 rows, accept / reject.
 * Replaces the Python loop around `Runner.lnprob` that emcee's EnsembleSampler runs for the reference
 * (analysis/runner.py:403-419; prior: runner.py:182-217, parameter.py:684-705) when the prior is a box, as in the shipped
 * parameter files.  All random numbers are the caller's, in the layout of mcmc_dynamics_amd/sampler.py:
 *   order [n_steps][W]       permutation of 0..W-1 per step: first half = order[:W/2], second = order[W/2:]
 *   zz    [n_steps][2][W/2]  stretch factors z ~ g(z), thr [n_steps][2][W/2] = log(u) - (n_dim - 1) log(z)
 *   pick  [n_steps][2][W/2]  partner index into the complementary half
 * pos [W][n_dim] and lnp [W] are updated in place; chain [n_steps][W][n_dim], lnprob_chain [n_steps][W] (either may be
 * NULL) receive the state after every step; accepted [W] (may be NULL) is incremented.  The chain is bit-identical to the
 * one the Python loop produces from the same numbers.  Returns MCD_ERR_NONFINITE when the likelihood produced a NaN (emcee
 * raises "Probability function returned NaN").
 *
 * Binned catalogues (desc->n_bins = B = the catalogue's number of radial bins): B independent ensembles, one per bin --
 * the reference runs one MCMC per bin (bin/run_tests.py:75-124) -- advance in lockstep and share every evaluation (one
 * launch of B x W/2 rows per half step).  Every array gains a bin dimension in front of the walker dimension:
 * pos [B][W][n_dim], lnp [B][W], accepted [B][W], order [n_steps][B][W], zz / thr / pick [n_steps][2][B][W/2],
 * chain [n_steps][B][W][n_dim], lnprob_chain [n_steps][B][W]; prior bounds and column map are shared by the bins.
 *
 * Where it runs.  A float64 catalogue on one device per process (single GPU, or one rank of a multi-process job) keeps
 * the ensemble RESIDENT on the device for the block: positions, log-probabilities and the block's random numbers are
 * uploaded once, one small kernel per half step accepts / rejects the previous half step and proposes the next one
 * (prior, resolved parameter rows, walker constants, range guard), and the whole block is a chain of launches the host
 * waits for once -- no host round trip between two evaluations (option "device_chain", default 1; csrc/mcd_stretch.hip).
 * The numbers are the host-driven loop's, bit for bit.  What only the host loop can handle -- a NaN, a re-run request of
 * the fast mixture kernels, a proposal table for which the range guard picks another kernel family than was enqueued, a
 * half step with every proposal outside the prior (binned: an ensemble without a valid proposal) -- makes the library
 * discard the block and run it host-driven from the same inputs; mcd_stretch_info counts both kinds.  pos, lnp and
 * accepted are only ever written with final values; chain / lnprob_chain of a block that moves more than 16 MB are
 * filled part by part while the device works on (a discarded block's rows are overwritten by its host-driven re-run).
 * Binned catalogues run resident for ensembles of up to 512 walkers and 12 columns. */
typedef struct {
    int64_t n_walkers;          /* W, even */
    int32_t n_dim;              /* free parameters (columns of pos) */
    int32_t k;                  /* mcd_catalog_param_count(cat) */
    const int32_t* col_source;  /* [k] index of the free parameter feeding kernel column j, or -1 for a constant column */
    const double* col_const;    /* [k] value of a constant column (a fixed parameter), in the kernel's unit */
    const double* col_factor;   /* [k] unit factor applied to a free-parameter column (1.0: none) */
    const double* lo;           /* [n_dim] inclusive prior bounds; -inf / +inf where unbounded */
    const double* hi;
    int32_t fixed_ok;           /* 0: a fixed parameter violates its own bounds, every proposal is rejected (runner.py:207-214) */
    int32_t n_bins;             /* 0 or 1: one ensemble (un-binned catalogue); B > 1: B lock-stepped ensembles, one per
                                 * parameter set (radial bin) of the catalogue -- must equal its number of bins */
} mcd_stretch_desc;

int mcd_stretch_move(mcd_catalog* cat, const mcd_stretch_desc* desc, int64_t n_steps, double* pos, double* lnp,
                     const int32_t* order, const double* zz, const double* thr, const int32_t* pick, double* chain,
                     double* lnprob_chain, int64_t* accepted);
/* Blocks of mcd_stretch_move that ran resident on the device / host-driven, blocks the device discarded (they were then
 * run host-driven and count there too) and the status bits of the last discarded one (1 NaN, 2 re-run request, 4 kernel
 * family changed, 8 no proposal inside the prior).  Any pointer may be NULL. */
/* The same block with its random numbers GENERATED INSIDE the library from a counter-based generator (Philox4x64-10, the
 * algorithm of numpy.random.Philox; mcmc_dynamics_amd/csrc/mcd_rng.h): the chain is a function of (seed, step, half step,
 * ensemble, walker) alone -- no numbers cross PCIe, blocks of any length continue each other (step0 = index of the block's first
 * step), and any step can be replayed on the host: mcd_chain_numbers returns the numbers of steps step0 .. step0 + n_steps - 1 in
 * the layout mcd_stretch_move takes (order [n_steps][B][W], zz / thr / pick [n_steps][2][B][W/2]), so that
 * mcd_stretch_move(..., those arrays, ...) gives the same chain bit for bit (tests/test_gpu_device_chain.py).  The logarithms
 * of the acceptance thresholds are a fixed sequence of IEEE operations (det_log), not libm calls, for that reason.
 * emcee (analysis/runner.py:403-419) draws from NumPy's Mersenne twister on the host instead. */
int mcd_stretch_move_seeded(mcd_catalog* cat, const mcd_stretch_desc* desc, int64_t n_steps, double* pos, double* lnp,
                            uint64_t seed, int64_t step0, double* chain, double* lnprob_chain, int64_t* accepted);
int mcd_chain_numbers(uint64_t seed, int64_t step0, int64_t n_steps, int64_t n_bins, int64_t n_walkers, int32_t n_dim,
                      int32_t* order, double* zz, double* thr, int32_t* pick);

int mcd_stretch_info(const mcd_catalog* cat, int64_t* device_blocks, int64_t* host_blocks, int64_t* discarded_blocks,
                     int32_t* last_discard_status);

/* ---- introspection for the measurement harness ------------------------------------------- */

const char* mcd_last_error(void);
int mcd_abi_version(void);
/* HIP-event time of the most recent mcd_loglike_batch / enqueue+sync: main kernel only, and the
 * whole device-side sequence (prep + main + reduce), milliseconds, device 0 of this process. */
double mcd_last_kernel_ms(const mcd_catalog* cat);
double mcd_last_device_ms(const mcd_catalog* cat);
/* Tuning / measurement switches, per catalogue.  Keys:
 *   "timing"        1: record HIP events around every enqueue (default 0); 2: additionally keep one
 *                      event pair per main-kernel launch for mcd_timing_collect
 *   "timing_stride" n: with "timing" = 2, record the event pair on every n-th launch only (default 1); the events cost a
 *                      signal packet each between back-to-back kernels, so a throughput harness samples
 *   "timing_discard" (any value): drop the event pairs recorded so far without reading them -- cheap, unlike
 *                      mcd_timing_collect, whose hipEventElapsedTime calls idle the GPU for milliseconds
 *   "timing_reserve" n: create the event pairs for n launches of "timing" = 2 now, so that a measured loop does not
 *                      pay for hipEventCreate
 *   "fast_path"     0: always use the plain per-term log/divide kernels; 1 (default): the fast formulations
 *                      (fraction tree / log-product / single-exp mixtures) are used whenever the per-call range
 *                      guard allows, including the narrow-range variant of the fixed-background mixture;
 *                      2: as 1 but never the narrow-range variant (testing aid)
 *   "zero_copy"     1 (default): on a single device mcd_loglike_batch lets the kernels read the parameter table
 *                      from / write the results to pinned mapped host memory instead of issuing H2D / D2H copies
 *   "spin_us"       microseconds mcd_sync / fetch / batch poll the stream (hipStreamQuery) before they fall back to the
 *                      blocking hipStreamSynchronize, whose interrupt wake-up adds 50 - 500 us of jitter to waits longer
 *                      than a fraction of a millisecond (default 20000; 0: always block)
 *   "device_chain"  1 (default): mcd_stretch_move keeps the ensemble resident on the device where it can (see there);
 *                      0: host-driven blocks only
 *   "prefetch"      software prefetch of the star records of the next loop iteration (a second instantiation of the fast
 *                      kernels): -1 (default) by shape -- mixture models from 8 MiB of records per device up (it hides
 *                      the memory latency there: +8 % at 256 walkers to +45 % at 64 on 1e6 stars), models without
 *                      background only for un-binned catalogues from 128 MiB up (their 16-star loop hides the latency
 *                      itself; the prefetch costs binned and many-walker shapes 3 - 6 %); 0 off, 1 on.  Results do not
 *                      depend on it.
 *   "target_waves"  number of waves the chunking aims for per device (default 10240)
 *   "chunk_len"     explicit nominal chunk length in stars (rounded up to a multiple of 32; 0, the default: derived from
 *                      "target_waves"); tuning aid
 *   "tail_split"    chunk schedule: 0 equal-length chunks; 1 (default): the last ~15 % of a large parameter set is
 *                      cut into half- and quarter-length chunks so that the launch ends on short waves; 2-4:
 *                      other guided schedules kept for tuning (see build_workset in mcd_api.hip)
 *   "balance"       balanced single-round chunk plans for small catalogues (every workgroup of the launch resident at once,
 *                      chunks of equal length: no tail, no second round): -1 (default) by work -- 2, 4 or 8 workgroups per
 *                      CU below ~1e6 work units, the multi-round table beyond; 0 never; 1 .. 8 forced
 *   "combine"       balanced plans run as 8- / 16-wave workgroups that add up their chunks' sums themselves (one partial
 *                      sum per workgroup and walker instead of one per chunk): 1 (default) the largest the plan allows,
 *                      0 never, 8 / 16 at most that many waves
 *   "two_lanes"     1 (default): pipelined evaluations (mcd_loglike_enqueue back to back) alternate between two streams
 *                      with their own partial-sum and result buffers, so that one launch's tail and reduction overlap
 *                      the next launch's start; 0: one stream.  Results do not depend on it.
 *   "fused_reduce"  1 (default): resident stretch-move blocks whose launches leave <= 256 partial sums per walker run
 *                      without the reduction kernel (the step kernel adds them up, same code, same order); 0: never
 *   "defer_guard"   1 (default): resident blocks of ONE ensemble judge the range guard of all their launches after the
 *                      last step instead of between two main kernels (same verdicts, same discards); 0: in the step kernel
 *   "f32_domain"    1 (default): float32 catalogues refuse parameter tables outside the float32 accuracy domain (below)
 * Returns MCD_ERR_INVALID for an unknown key. */
int mcd_set_option(mcd_catalog* cat, const char* key, int64_t value);
/* With "timing" = 2: waits for the device, returns the summed HIP-event duration (ms) of all main-kernel
 * launches on device 0 of this process since the last collect / option change and their number. */
int mcd_timing_collect(mcd_catalog* cat, double* total_kernel_ms, int64_t* n_launches);
/* Number of batches this catalogue re-evaluated with the plain kernels because a fast mixture kernel met the regime in
 * which the reference's log-sum-exp (runner.py:282-284) itself runs on denormal numbers -- a star with pmember == 1,
 * f_back == 0 or density == 0 that lies > 37 sigma from the only remaining component.  Only the literal expression
 * reproduces the reference's value there; the re-evaluation is automatic and synchronous inside fetch / batch. */
int64_t mcd_rerun_count(const mcd_catalog* cat);
/* 1 when the most recent main-kernel launch used the instantiation that prefetches the next loop iteration's star records
 * (option "prefetch"), 0 when not, -1 before the first launch.  The harness picks the per-term instruction count of the
 * roofline by it (csrc/isa_mix.json holds both instantiations). */
int mcd_last_prefetch(const mcd_catalog* cat);
/* Kernel family the range guard chose for the batch staged last: 0 plain, 1 fast formulation, 2 narrow-range variant of
 * the mixture kernels (no per-star exponent bookkeeping; chunks holding a star outside its domain -- a certain member, an
 * extreme background likelihood, an empty component -- still run the fast formulation); -1 before any call. */
int mcd_last_fast_level(const mcd_catalog* cat);

/* float32 accuracy domain (MCD_F32, MCD_F32_ACC64).  The float32 kernels round every record field and walker constant to
 * 24 bits first; they stay within 1e-6 (MCD_F32_ACC64) / 2e-5 (MCD_F32) of the float64 kernels -- fixed centre; 2e-5 /
 * 1e-4 with a free centre -- on the scale max(|lnL|, N, 32) only while
 *     kappa_v = (max|v| + |v_sys| + |v_maxx| + |v_maxy|) / sqrt(min(verr^2) + min(sigma^2))      <= 96
 *     kappa_theta = (|v_maxx| + |v_maxy|) / sqrt(min(verr^2) + min(sigma^2)) * 2^-23 / sep_harm   <= 2e-5   (free centre;
 *                   sep_harm: harmonic mean angular separation [rad] of the stars from the catalogue's centroid)
 * and variances, residuals and mixture values lie in the float32 ranges (norm within 2^-15 .. 2^15, |v - v_los| <= 2^15,
 * lnL_bg within -80 .. 60, pmember <= 1 - 2^-20, density and f_back within 2^-20 .. 2^20); derivation in
 * mcmc_dynamics_amd/csrc/mcd_guard.h (f32_domain), evidence in profiles/r03_fuzz_f32.txt (tools/fuzz_f32.py: errors up to
 * 8.5e-4 outside on its ranges, 0.58 in the wider campaign of round 2).  A parameter table outside the domain is REFUSED: mcd_params_upload / mcd_loglike_batch return
 * MCD_ERR_INVALID with the reason (option "f32_domain" = 0: evaluate regardless).  mcd_last_f32_domain reports the verdict
 * on the last staged table (1 inside, 0 outside; always 1 for MCD_F64 catalogues) and the two condition numbers. */
int mcd_last_f32_domain(const mcd_catalog* cat, double* kappa_v, double* kappa_theta);
/* Launch geometry of the main kernel for the last call: workgroups, walker tile (walkers that
 * reuse one star record load), chunks per parameter set, bytes per star record. */
int mcd_last_launch_info(const mcd_catalog* cat, int64_t* n_workgroups, int32_t* walker_tile,
                         int64_t* n_chunks, int32_t* record_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MCD_H */
