"""Randomised campaign: `mcd_stretch_move` resident on the device against the host-driven loop of the same library
(longer version of tests/test_gpu_device_chain.py).

    python tools/fuzz_chain.py [--trials 300] [--seed 1] [--seconds 200] [--force-rccl]

Random catalogues and ensembles over many orders of magnitude (tests/test_guard_random_cpu.py: random_case), all seven
models, fixed and free centre, 4 .. 300 walkers, random box priors, random plans (fixed columns, unit factors); every
third case a binned catalogue (2 .. 6 ensembles in lockstep).  The two
paths must agree in every bit of positions, log-probabilities, chain and acceptance counts -- whether the device keeps the
block or gives it back -- and must raise the same error for a NaN.  Exit status 1 on any difference."""
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


def run(cat, plan, pos, lnp, randoms, mode):
    cat.set_option("device_chain", mode)
    n = randoms[0].shape[0]
    p, l = pos.copy(), lnp.copy()
    chain, lnpc, acc = np.empty((n,) + pos.shape), np.empty((n,) + lnp.shape), np.zeros(lnp.shape, dtype=np.int64)
    try:
        cat.stretch_move(plan, p, l, *randoms, chain, lnpc, acc)
    except native.NativeError as e:
        return ("error", "NaN" in str(e))
    return p, l, chain, lnpc, acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=200.0)
    ap.add_argument("--force-rccl", action="store_true", help="1-rank communicator: sums and status word through ncclAllReduce")
    a = ap.parse_args()
    if a.force_rccl:
        os.environ["MCD_FORCE_RCCL"] = "1"
        ctx = native.Context(rank=0, n_ranks=1, unique_id=native.Context.unique_id(), device=0)
        del os.environ["MCD_FORCE_RCCL"]
    else:
        ctx = native.default_context()
    t0 = time.time()
    total = bad = kept = discarded = errors = binned = 0
    reasons = {1: 0, 2: 0, 4: 0, 8: 0}
    for trial in range(a.trials):
        if time.time() - t0 > a.seconds:
            break
        for model in range(7):
            for free in (False, True):
                if free and model == 4:
                    continue
                seed = a.seed * 1000003 + trial * 131 + model * 7 + int(free)
                rng = np.random.default_rng(seed)
                n = int(rng.integers(1, 3000))
                w = 2 * int(rng.integers(2, 150))
                cat, params = random_case(rng, model, n=n, w=w)
                tame = trial % 3 != 0                       # two thirds: ranges the fast kernels accept
                if tame:
                    cat["pmember"] = np.clip(cat["pmember"], 0.01, 0.99)
                    cat["density"] = np.clip(cat["density"], 0.01, 1.0)
                    if params.shape[1] > 4 and model in (2, 4, 5):
                        params[:, -1] = np.clip(params[:, -1], 0.05, 0.95)
                kw = {}
                if model in (1, 6):
                    kw = dict(lnlike_bg=cat["lnlike_bg"], pmember=cat["pmember"])
                elif model in (2, 4):
                    kw = dict(density=cat["density"])
                elif model == 5:
                    kw = dict(lnlike_bg=cat["lnlike_bg"], density=cat["density"])
                centre = CENTRE
                if free:
                    head = 6 if model >= 3 else 4
                    cc = np.column_stack([CENTRE[0] + rng.normal(0, 0.01, w), CENTRE[1] + rng.normal(0, 0.01, w)])
                    params = np.hstack([params[:, :head], cc, params[:, head:]])
                    centre = None
                k = params.shape[1]
                # every third case: a binned catalogue (B radial bins = B lock-stepped ensembles sharing each evaluation)
                n_bins = int(rng.integers(2, 7)) if trial % 3 == 1 and n >= 12 else 1
                if n_bins > 1:
                    edges = np.sort(rng.choice(np.arange(1, n), size=n_bins - 1, replace=False))
                    kw["bin_offsets"] = np.concatenate([[0], edges, [n]]).astype(np.int64)
                g = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=model, centre=centre, **kw)
                assert g.k == k
                # a plan: some kernel columns fixed at a walker's value, some sampled in other units
                fixed = rng.random(k) < 0.15
                fixed[1] = False
                factor = np.where(rng.random(k) < 0.2, 10.0 ** rng.integers(-2, 3, k), 1.0)
                free_cols = np.flatnonzero(~fixed)
                src = np.full(k, -1, dtype=np.int32)
                src[free_cols] = np.arange(free_cols.size)
                const = np.where(fixed, params[0], 0.0)
                pos = np.ascontiguousarray(params[:, free_cols] / factor[free_cols])
                lo, hi = np.full(free_cols.size, -np.inf), np.full(free_cols.size, np.inf)
                for c in range(free_cols.size):             # priors that cut into the ensemble now and then
                    r = rng.random()
                    span = np.ptp(pos[:, c]) + 1e-300
                    if r < 0.25:
                        lo[c] = pos[:, c].min() - span * rng.uniform(0.0, 0.5)
                    elif r < 0.5:
                        hi[c] = pos[:, c].max() + span * rng.uniform(0.0, 0.5)
                if model >= 3:                              # a, r_peak stay positive
                    for col in (2, 5):
                        if not fixed[col]:
                            lo[src[col]] = max(lo[src[col]], 1e-3 / factor[col])
                if not fixed[1]:
                    lo[src[1]] = max(lo[src[1]], 0.0)
                plan = {"col_source": src, "col_const": const, "col_factor": np.where(fixed, 1.0, factor), "lo": lo, "hi": hi,
                        "fixed_ok": True}
                steps = int(rng.integers(1, 12))
                half = w // 2
                if n_bins > 1:
                    # each ensemble a jittered copy of the walkers (inside the prior where the original is)
                    jitter = 1.0 + 1e-3 * rng.normal(size=(n_bins,) + pos.shape)
                    pos = np.ascontiguousarray(np.clip(pos[None] * jitter, np.where(np.isfinite(lo), lo, -np.inf), np.where(np.isfinite(hi), hi, np.inf)))
                    flat = pos.reshape(-1, pos.shape[-1])
                    table = np.where(src[None, :] >= 0, flat[:, np.maximum(src, 0)] * plan["col_factor"][None, :], const[None, :])
                    lnp = np.ascontiguousarray(g.loglike(np.ascontiguousarray(table).reshape(n_bins, w, k)))
                    lead = (n_bins,)
                else:
                    table = np.where(src[None, :] >= 0, pos[:, np.maximum(src, 0)] * plan["col_factor"][None, :], const[None, :])
                    lnp = g.loglike(np.ascontiguousarray(table))
                    lead = ()
                if np.isnan(lnp).any():
                    g.close()
                    continue
                order = np.argsort(rng.random((steps,) + lead + (w,)), axis=-1).astype(np.int32)
                u = rng.random((steps, 4) + lead + (half,))
                zz = np.ascontiguousarray((u[:, :2] + 1.0) ** 2 / 2.0)
                thr = np.ascontiguousarray(np.log(u[:, 2:]) - (free_cols.size - 1.0) * np.log(zz))
                pick = rng.integers(0, half, size=(steps, 2) + lead + (half,)).astype(np.int32)
                randoms = (order, zz, thr, pick)
                before = g.stretch_info()
                dev = run(g, plan, pos, lnp, randoms, 2 if trial % 5 == 4 and n_bins == 1 else 1)
                info = g.stretch_info()
                host = run(g, plan, pos, lnp, randoms, 0)
                g.close()
                total += 1
                binned += n_bins > 1
                if info["discarded_blocks"] > before["discarded_blocks"]:
                    discarded += 1
                    for bit in reasons:
                        reasons[bit] += bool(info["last_discard_status"] & bit)
                else:
                    kept += 1
                if isinstance(dev[0], str) or isinstance(host[0], str):
                    errors += 1
                    same = dev == host
                else:
                    same = all(np.array_equal(x, y, equal_nan=True) for x, y in zip(dev, host))
                if not same:
                    bad += 1
                    print("MISMATCH seed", seed, "model", model, "free", free, "n", n, "w", w, "steps", steps, info, flush=True)
        if trial % 10 == 0:
            print("trial", trial, "blocks", total, "bad", bad, "kept on device", kept, "given back", discarded, reasons,
                  "NaN errors", errors, "elapsed %.0f s" % (time.time() - t0), flush=True)
    print("DONE blocks", total, "(binned:", binned, ") bad", bad, "kept on device", kept, "given back", discarded,
          "reasons (1 NaN, 2 re-run, 4 family, 8 no proposal):", reasons, "NaN errors", errors, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
