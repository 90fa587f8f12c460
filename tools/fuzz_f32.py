"""Randomised float32 vs float64 campaign on the device: how far the float32 kernels (MCD_F32: float32
terms and sums; MCD_F32_ACC64: float32 terms, float64 sums) are from the float64 kernels, inside and outside the float32
accuracy domain of csrc/mcd_guard.h (f32_domain), for every model, fixed and free centre.

    python tools/fuzz_f32.py [--seconds 120] [--seed 1] [--csv gpurun_out/fuzz_f32_cases.csv]

Each case: a random catalogue (half of them over the wide ranges of tests/test_guard_random_cpu.py -- velocity scales 0.1 ..
3000 km/s, errors 1e-3 .. 300 km/s, gross outliers --, half over ranges of real data) and a random walker table, evaluated
with a float64 catalogue and with the two float32 catalogues, the latter with option f32_domain = 0 (evaluate anyway,
mcd_last_f32_domain() tells whether the call was inside the domain).  Error scale as in tools/fuzz_gpu.py:
|lnL32 - lnL64| / max(|lnL64|, N, 32).  Reported: worst error inside / outside the domain per precision and centre mode; that
every call outside the domain is refused (MCD_ERR_INVALID) when the option is on; the cases beyond the stated tolerances
inside the domain (exit status 1 if there is one):
    fixed centre  f32acc64 1e-6    f32 2e-5        free centre  f32acc64 2e-5    f32 1e-4
"""
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

TOL = {(False, "f32acc64"): 1e-6, (False, "f32"): 2e-5, (True, "f32acc64"): 2e-5, (True, "f32"): 1e-4}


def realistic_case(rng, model, n, w):
    """Ranges of real data: dispersion 1 .. 30 km/s, errors 0.03 .. 3 sigma, systemic velocities up to 30 sigma."""
    sig0 = 10.0 ** rng.uniform(0.0, 1.5)
    sep = np.maximum(np.abs(rng.normal(0, 10.0 ** rng.uniform(-2.0, -0.3), n)), 1e-5)       # degrees: 0.6' .. 30' scale
    th = rng.uniform(-np.pi, np.pi, n)
    vsys = rng.normal(0, 30.0 * sig0) if rng.random() < 0.5 else rng.normal(0, sig0)
    cat = {"ra": CENTRE[0] + sep * np.cos(th) / np.cos(np.radians(CENTRE[1])), "dec": CENTRE[1] + sep * np.sin(th),
           "v": vsys + rng.normal(0, sig0, n), "verr": sig0 * 10.0 ** rng.uniform(-1.5, 0.5) * rng.lognormal(0, 0.5, n)}
    out = rng.random(n) < 0.1
    cat["v"][out] = vsys + rng.normal(0, 5.0 * sig0, int(out.sum()))
    cat["density"] = np.clip(np.exp(-0.5 * (sep / np.median(sep)) ** 2), 0.02, 1.0)
    cat["pmember"] = cat["density"] / (cat["density"] + 0.25)
    cat["lnlike_bg"] = -0.5 * ((cat["v"] - vsys) / (5 * sig0)) ** 2 - np.log(5 * sig0) - 0.9
    sig = sig0 * 10.0 ** rng.uniform(-0.3, 0.3, w)
    cols = [vsys + rng.normal(0, 0.3 * sig0, w), sig]
    if model >= 3:
        cols.append(10.0 ** rng.uniform(0.5, 2.5, w))
    rot = sig0 * 10.0 ** rng.uniform(-1.5, 0.0)
    cols += [rng.normal(0, rot, w), rng.normal(0, rot, w)]
    if model >= 3:
        cols.append(10.0 ** rng.uniform(0.5, 2.5, w))
    if model in (2, 4):
        cols += [vsys + rng.normal(0, sig0, w), 5 * sig0 * 10.0 ** rng.uniform(-0.3, 0.3, w), 0.05 + 0.9 * rng.random(w)]
    if model == 5:
        cols.append(0.05 + 0.9 * rng.random(w))
    return cat, np.stack(cols, axis=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-stars", type=int, default=3000)
    ap.add_argument("--csv", default=None, help="one line per case: metrics and errors (for deriving the domain)")
    a = ap.parse_args()
    ctx = native.default_context()
    t0 = time.time()
    csv = open(a.csv, "w") if a.csv else None
    if csv:
        csv.write("seed,kind,model,free,n,w,precision,in_domain,level,err,lnl_scale,kappa_v,kappa_theta\n")
    stats = {}          # (free, precision, in_domain) -> [cases, worst, worst seed]
    refused_ok = refused_bad = accepted_ok = accepted_bad = 0
    beyond = []
    trial = 0
    while time.time() - t0 < a.seconds:
        trial += 1
        for model in range(7):
            for free in (False, True):
                if free and model == 4:
                    continue
                seed = a.seed * 1000003 + trial * 97 + model * 7 + int(free)
                rng = np.random.default_rng(seed)
                n = int(rng.integers(1, a.max_stars))
                w = int(rng.integers(1, 100))
                kind = "wide" if trial % 2 else "real"
                cat, params = (random_case if kind == "wide" else realistic_case)(rng, model, n=n, w=w)
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
                    cc = np.column_stack([CENTRE[0] + rng.normal(0, 0.003, len(params)), CENTRE[1] + rng.normal(0, 0.003, len(params))])
                    params = np.hstack([params[:, :head], cc, params[:, head:]])
                    centre = None
                g64 = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=model, centre=centre, **kw)
                want = g64.loglike(params)
                g64.close()
                for precision in ("f32acc64", "f32"):
                    g = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=model, centre=centre,
                                       precision=precision, **kw)
                    g.set_option("f32_domain", 0)
                    got = g.loglike(params)
                    inside = g.f32_in_domain
                    kv, kt = g.f32_condition
                    level = g.fast_level
                    g.set_option("f32_domain", 1)
                    try:
                        g.loglike(params)
                        refused = False
                    except native.NativeError as exc:
                        refused = "float32 accuracy domain" in str(exc)
                    g.close()
                    if inside:
                        accepted_ok += int(not refused)
                        accepted_bad += int(refused)
                    else:
                        refused_ok += int(refused)
                        refused_bad += int(not refused)
                    ok = np.isfinite(want) & np.isfinite(got)
                    pattern = np.array_equal(np.isfinite(want), np.isfinite(got))
                    scale = np.maximum(np.abs(want[ok]), max(float(n), 32.0))     # (one float32 term has an absolute floor of a few 1e-6)
                    err = float(np.max(np.abs(got[ok] - want[ok]) / scale)) if ok.any() else 0.0
                    if not pattern and inside:
                        err = max(err, 1.0)
                    key = (free, precision, bool(inside))
                    s = stats.setdefault(key, [0, 0.0, None])
                    s[0] += 1
                    if err > s[1]:
                        s[1], s[2] = err, (seed, kind, model, n, w, level)
                    if inside and err > TOL[(free, precision)]:
                        beyond.append((seed, kind, model, free, n, w, precision, level, err))
                    if csv:
                        csv.write("{0},{1},{2},{3},{4},{5},{6},{7},{8},{9:.3e},{10:.3e},{11:.3e},{12:.3e}\n".format(
                            seed, kind, model, int(free), n, w, precision, int(inside), level, err,
                            float(np.median(np.abs(want[ok]))) if ok.any() else 0.0, kv, kt))
        if trial % 50 == 0:
            print("trial", trial, "elapsed %.0f s" % (time.time() - t0), flush=True)
    print("float32 against float64, error on the scale max(|lnL|, N); tolerances inside the domain:", TOL)
    for key in sorted(stats):
        free, precision, inside = key
        c, worst, where = stats[key]
        print("  {0:5s} centre  {1:8s}  {2:7s} the domain: {3:6d} cases, worst {4:.2e}  (seed, kind, model, stars, walkers, family) {5}".format(
            "free" if free else "fixed", precision, "inside" if inside else "outside", c, worst, where))
    print("enforcement (option f32_domain = 1): outside the domain refused {0} / {1}; inside the domain evaluated {2} / {3}".format(
        refused_ok, refused_ok + refused_bad, accepted_ok, accepted_ok + accepted_bad))
    for b in beyond[:20]:
        print("BEYOND TOLERANCE inside the domain:", b)
    inside_cases = sum(v[0] for k, v in stats.items() if k[2])
    print("DONE cases", sum(v[0] for v in stats.values()), "inside the domain", inside_cases, "beyond tolerance inside", len(beyond),
          "not refused outside", refused_bad, "refused inside", accepted_bad, flush=True)
    return 1 if beyond or refused_bad or accepted_bad else 0


if __name__ == "__main__":
    sys.exit(main())
