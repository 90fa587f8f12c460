"""Randomised fast-path vs plain-path campaign on the device (longer version of
tests/test_gpu_kernels.py::test_fast_and_plain_kernels_agree_over_wide_ranges).

    python tools/fuzz_gpu.py [--trials 400] [--seed 1] [--seconds 240]

For every model (fixed and free centre) random catalogues / walker tables spanning many orders of magnitude are evaluated
with the guarded fast path and with the plain (reference-literal) kernels; any disagreement in the finite / -inf pattern
or beyond 1e-11 relative is printed with its seed.  Exit status 1 if any trial disagreed.

Float32 modes compare the float32 fast formulations with the plain float32 kernels: a consistency check between two
approximations, with the tolerance at the accuracy floor of the less accurate one -- which is the plain kernel (its
literal log-sum-exp in float32: seed 36136862, model 2, f32acc64, deviates 3.0e-5 from the float64 value where the fast
formulation deviates 9e-8).  Accuracy against float64 on realistic ranges is what tests/test_gpu_baseline_shapes.py and
test_precision_sweep_c5 assert."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mcmc_dynamics_amd import _native as native  # noqa: E402
from test_guard_random_cpu import CENTRE, random_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=400)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--max-stars", type=int, default=1500)
    ap.add_argument("--max-walkers", type=int, default=200, help="> 256 exercises the XCD-aware workgroup mapping")
    ap.add_argument("--schedule", action="store_true", help="also randomise target_waves / tail_split per case")
    ap.add_argument("--precision", default="f64", choices=["f64", "f32acc64", "f32"],
                    help="float32 modes compare the float32 fast formulations with the plain float32 kernels (tolerance 5e-5 / 5e-4)")
    ap.add_argument("--force-rccl", action="store_true",
                    help="1-rank communicator (MCD_FORCE_RCCL=1): the collective code path, where the re-run signal travels "
                         "as NaN-poisoned partial sums through the all-reduce")
    a = ap.parse_args()
    if a.force_rccl:
        os.environ["MCD_FORCE_RCCL"] = "1"
        ctx = native.Context(rank=0, n_ranks=1, unique_id=native.Context.unique_id(), device=0)
        del os.environ["MCD_FORCE_RCCL"]
    else:
        ctx = native.default_context()
    t0 = time.time()
    bad = total = reruns = admitted_inf = 0
    worst_case = None
    levels = [0, 0, 0]                                     # batches per kernel family (plain / fast / narrow)
    worst = 0.0
    for trial in range(a.trials):
        if time.time() - t0 > a.seconds:
            break
        for model in range(7):
            for free in (False, True):
                if free and model == 4:
                    continue
                seed = a.seed * 1000003 + trial * 97 + model * 7 + int(free)
                rng = np.random.default_rng(seed)
                n = int(rng.integers(1, a.max_stars))
                w = int(rng.integers(1, a.max_walkers))
                cat, params = random_case(rng, model, n=n, w=w)
                kw = {}
                if model in (1, 6):
                    if trial % 2:                          # no certain members: the narrow-range variant becomes eligible
                        cat["pmember"] = np.minimum(cat["pmember"], 1.0 - 2.0 ** -float(rng.integers(1, 54)))
                    kw = dict(lnlike_bg=cat["lnlike_bg"], pmember=cat["pmember"])
                elif model in (2, 4):
                    if trial % 2:                          # no empty component: the narrow-range variant becomes eligible
                        cat["density"] = np.maximum(cat["density"], 10.0 ** -rng.uniform(0, 6.0))
                        params[:, -1] = np.maximum(params[:, -1], 10.0 ** -rng.uniform(0, 6.0))
                    kw = dict(density=cat["density"])
                elif model == 5:
                    if trial % 2:
                        params[:, -1] = np.maximum(params[:, -1], 10.0 ** -rng.uniform(0, 6.0))
                    kw = dict(lnlike_bg=cat["lnlike_bg"], density=cat["density"])
                centre = CENTRE
                if free:
                    head = 6 if model >= 3 else 4          # (model 6 has no tail columns)
                    cc = np.column_stack([CENTRE[0] + rng.normal(0, 0.01, len(params)),
                                          CENTRE[1] + rng.normal(0, 0.01, len(params))])
                    params = np.hstack([params[:, :head], cc, params[:, head:]])
                    centre = None
                g = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=model, centre=centre,
                                   precision=a.precision, **kw)
                if a.precision != "f64":
                    g.set_option("f32_domain", 0)          # fast vs plain float32 consistency over wide ranges (accuracy: tools/fuzz_f32.py)
                if a.schedule:
                    g.set_option("target_waves", int(rng.integers(1, 20000)))
                    g.set_option("tail_split", int(rng.integers(0, 3)))
                fast = g.loglike(params)
                g_level = g.fast_level
                levels[g_level] += 1
                reruns += g.rerun_count
                g.set_option("fast_path", 0)
                plain = g.loglike(params)
                g.close()
                total += 1
                same = np.array_equal(np.isfinite(fast), np.isfinite(plain)) and \
                    np.array_equal(np.isnan(fast), np.isnan(plain))
                ok = np.isfinite(plain) & np.isfinite(fast)
                # every term is O(1 .. 10): a total that cancels to less than n is judged on the scale n
                err = float(np.max(np.abs(fast[ok] - plain[ok]) / np.maximum(np.abs(plain[ok]), float(n)))) if ok.any() else 0.0
                admitted_inf += int(np.isneginf(plain).sum())
                if err > worst:
                    worst, worst_case = err, (seed, model, free, n, w, g_level)
                if not same or err > {"f64": 1e-11, "f32acc64": 5e-5, "f32": 5e-4}[a.precision]:
                    bad += 1
                    print("MISMATCH seed", seed, "model", model, "free", free, "n", n, "w", w, "pattern_same", same, "err", err,
                          flush=True)
        if trial % 20 == 0:
            print("trial", trial, "cases", total, "bad", bad, "worst rel err", worst, "reruns", reruns, "-inf walkers", admitted_inf,
                  "elapsed %.0f s" % (time.time() - t0), flush=True)
    print("kernel families chosen (plain, fast, narrow):", levels)
    print("worst case (seed, model, free centre, stars, walkers, family):", worst_case)
    print("DONE cases", total, "bad", bad, "worst rel err", worst, "reruns", reruns, "-inf walkers", admitted_inf, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
