"""Measure mcd_kde_background (background.SingleStars on the device) and check a sample against the oracle.

    python tests/probes/kde_probe.py [--stars 1000000] [--comp 10000] [--repeat 5]

Prints one JSON line: pairs/s from the HIP-event time of the two kernels, the blocking-call wall time (host buffers in
and out, PCIe included) and the NumPy restatement's rate on a bounded sample of the same test stars."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcmc_dynamics_amd import _native  # noqa: E402
from oracle import lnprob_numpy as oracle  # noqa: E402  (checker + CPU baseline only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stars", type=int, default=1_000_000)
    ap.add_argument("--comp", type=int, default=10_000)
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=2000)
    a = ap.parse_args()
    rng = np.random.default_rng(99)
    comp = rng.normal(20.0, 40.0, a.comp)
    v = np.where(rng.random(a.stars) < 0.2, rng.normal(20.0, 40.0, a.stars), rng.normal(0.0, 10.0, a.stars))
    verr = rng.lognormal(0.0, 0.5, a.stars)
    ctx = _native.default_context()
    ctx.kde_background(comp, v[:1000], verr[:1000])                      # warm-up (module load)
    kernel_ms, wall_ms = [], []
    for _ in range(a.repeat):
        t0 = time.perf_counter()
        out, ms = ctx.kde_background(comp, v, verr, 0.0, return_kernel_ms=True)
        wall_ms.append(1e3 * (time.perf_counter() - t0))
        kernel_ms.append(ms)
    ns = min(a.cpu_sample, a.stars)
    t0 = time.perf_counter()
    want = oracle.single_stars_background(comp, v[:ns], verr[:ns])
    cpu_s = time.perf_counter() - t0
    err = float(np.max(np.abs(out[:ns] - want) / np.maximum(1.0, np.abs(want))))
    pairs = float(a.stars) * a.comp
    k = float(np.median(kernel_ms))
    print(json.dumps({"what": "kde_background", "stars": a.stars, "comp": a.comp, "kernel_ms": k,
                      "pairs_per_s": pairs / (k * 1e-3), "call_wall_ms": float(np.median(wall_ms)),
                      "valu_slots_per_pair": 19, "cpu_port_pairs_per_s": ns * a.comp / cpu_s,
                      "cpu_sample_stars": ns, "max_err_vs_port": err}))


if __name__ == "__main__":
    main()
