// mcd_stretch.h -- host-only: one block of affine-invariant stretch-move steps (Goodman & Weare 2010, the default move of
// emcee's EnsembleSampler that the reference hands Runner.lnprob to, analysis/runner.py:403-419) with the per-half-step
// host work in C++: proposals from the complementary half of the ensemble, the box prior of Runner.lnprior
// (runner.py:182-217: inclusive bounds, NaN outside), the resolved kernel parameter table, accept / reject.  Between two
// kernel launches the host then spends a few microseconds instead of ~50 us of NumPy.
//
// Every random number comes from the caller, in the layout mcmc_dynamics_amd/sampler.py draws them, and the arithmetic is
// written operation for operation like the Python loop (no FMA contraction: the library is built with -ffp-contract=off),
// so that a chain produced here is bit-identical to the one the Python loop produces from the same generator state.
// No HIP types: shared by the C-ABI (mcd_api.hip) and the CPU test harness (tests/emul).
#pragma once

#include <cmath>
#include <cstdint>
#include <vector>

namespace mcd {

struct StretchDesc {
    int64_t n_bins = 1;                 // B lock-stepped ensembles, one per parameter set (radial bin) of the catalogue
    int64_t n_walkers = 0;              // W (even), per ensemble
    int32_t n_dim = 0;                  // P free parameters, in the sampler's column order
    int32_t k = 0;                      // columns of the kernel parameter table (include/mcd.h: mcd_catalog_param_count)
    const int32_t* col_source = nullptr;   // [k] index of the free parameter that feeds table column j, or -1: constant
    const double* col_const = nullptr;     // [k] value of a constant column (fixed parameter, already in kernel units)
    const double* col_factor = nullptr;    // [k] unit factor of a free-parameter column (applied only when != 1)
    const double* lo = nullptr;            // [P] inclusive lower prior bounds (-inf: none)
    const double* hi = nullptr;            // [P] inclusive upper prior bounds
    int32_t fixed_ok = 1;                  // 0: a fixed parameter violates its own bounds, every lnprob is -inf (runner.py:207-214)
};

enum StretchStatus : int { STRETCH_OK = 0, STRETCH_NAN = 1, STRETCH_EVAL_FAILED = 2, STRETCH_BAD_ARGS = 3 };

// eval(table [B * W/2][k], W/2, out [B * W/2]) -> 0 on success: the batched log-likelihood of W/2 parameter rows per
// parameter set (bin-major, the layout mcd_loglike_batch takes for a binned catalogue; B = 1: one table of W/2 rows).
// Array layouts with B ensembles: pos [B][W][P], lnp [B][W], accepted [B][W], order [n_steps][B][W],
// zz / thr / pick [n_steps][2][B][W/2], chain [n_steps][B][W][P], lnprob_chain [n_steps][B][W].  The ensembles are
// independent chains (reference: one MCMC per radial bin, bin/run_tests.py:75-124) that share every evaluation.
template <class Eval>
int stretch_block(const StretchDesc& d, int64_t n_steps, double* pos, double* lnp, const int32_t* order, const double* zz,
                  const double* thr, const int32_t* pick, double* chain, double* lnprob_chain, int64_t* accepted,
                  Eval&& eval) {
    const int64_t B = d.n_bins, W = d.n_walkers, half = W / 2;
    const int P = d.n_dim, K = d.k;
    if (B <= 0 || W <= 0 || (W & 1) || P <= 0 || K <= 0) return STRETCH_BAD_ARGS;
    const int64_t rows = B * half;
    std::vector<double> proposal((size_t)rows * P), table((size_t)rows * K), ll((size_t)rows), new_lnp((size_t)rows);
    std::vector<uint8_t> ok((size_t)rows);
    for (int64_t i = 0; i < n_steps; ++i) {
        for (int h = 0; h < 2; ++h) {
            // proposal = partner - (partner - s) * z   (sampler.py: `partners - (partners - s) * zz[:, None]`)
            int64_t n_ok = 0, donor = -1;
            for (int64_t b = 0; b < B; ++b) {
                const int32_t* ord = order + (i * B + b) * W;
                const int32_t* first = ord + (h == 0 ? 0 : half);
                const int32_t* second = ord + (h == 0 ? half : 0);
                const double* z = zz + ((i * 2 + h) * B + b) * half;
                const int32_t* pk = pick + ((i * 2 + h) * B + b) * half;
                const double* ens = pos + b * W * P;
                for (int64_t j = 0; j < half; ++j) {
                    const double* s = ens + (int64_t)first[j] * P;
                    const double* q = ens + (int64_t)second[pk[j]] * P;
                    double* p = proposal.data() + (b * half + j) * P;
                    bool good = d.fixed_ok != 0;
                    for (int c = 0; c < P; ++c) {
                        p[c] = q[c] - (q[c] - s[c]) * z[j];
                        good = good && (p[c] >= d.lo[c]) && (p[c] <= d.hi[c]);        // false for NaN as well
                    }
                    ok[b * half + j] = good;
                    if (good) { ++n_ok; if (donor < 0) donor = b * half + j; }
                }
            }
            if (n_ok > 0) {
                // rows outside the prior take a valid row's place in the launch and are masked afterwards (runner.py's
                // lnprob skips their evaluation; Runner.lnprob_batch does the same substitution)
                for (int64_t r = 0; r < rows; ++r) {
                    const double* p = proposal.data() + (ok[r] ? r : donor) * P;
                    double* row = table.data() + r * K;
                    for (int c = 0; c < K; ++c) {
                        const int src = d.col_source[c];
                        row[c] = src < 0 ? d.col_const[c] : (d.col_factor[c] == 1.0 ? p[src] : p[src] * d.col_factor[c]);
                    }
                }
                if (eval(table.data(), half, ll.data()) != 0) return STRETCH_EVAL_FAILED;
            }
            for (int64_t r = 0; r < rows; ++r) {
                new_lnp[r] = (n_ok > 0 && ok[r]) ? ll[r] : -INFINITY;
                if (new_lnp[r] != new_lnp[r]) return STRETCH_NAN;                 // "Probability function returned NaN"
            }
            // accept iff thr < new_lnp - old_lnp   (thr = log(u) - (P - 1) log(z), drawn by the caller)
            for (int64_t b = 0; b < B; ++b) {
                const int32_t* first = order + (i * B + b) * W + (h == 0 ? 0 : half);
                const double* t = thr + ((i * 2 + h) * B + b) * half;
                for (int64_t j = 0; j < half; ++j) {
                    const int64_t w = b * W + first[j], r = b * half + j;
                    if (t[j] < new_lnp[r] - lnp[w]) {
                        const double* p = proposal.data() + r * P;
                        for (int c = 0; c < P; ++c) pos[w * P + c] = p[c];
                        lnp[w] = new_lnp[r];
                        if (accepted) accepted[w] += 1;
                    }
                }
            }
        }
        if (chain) for (int64_t x = 0; x < B * W * P; ++x) chain[i * B * W * P + x] = pos[x];
        if (lnprob_chain) for (int64_t w = 0; w < B * W; ++w) lnprob_chain[i * B * W + w] = lnp[w];
    }
    return STRETCH_OK;
}

}  // namespace mcd
