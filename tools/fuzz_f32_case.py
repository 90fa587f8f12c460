"""Scratch: one case of tools/fuzz_f32.py again, in detail.    python tools/fuzz_f32_case.py SEED_ARG TRIAL MODEL FREE"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
from mcmc_dynamics_amd import _native as native
from test_guard_random_cpu import CENTRE, random_case
from fuzz_f32 import realistic_case

base, trial, model, free = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), bool(int(sys.argv[4]))
seed = base * 1000003 + trial * 97 + model * 7 + int(free)
rng = np.random.default_rng(seed)
n = int(rng.integers(1, 3000)); w = int(rng.integers(1, 100))
kind = "wide" if trial % 2 else "real"
cat, params = (random_case if kind == "wide" else realistic_case)(rng, model, n=n, w=w)
kw = {}
if model in (1, 6): kw = dict(lnlike_bg=cat["lnlike_bg"], pmember=cat["pmember"])
elif model in (2, 4): kw = dict(density=cat["density"])
elif model == 5: kw = dict(lnlike_bg=cat["lnlike_bg"], density=cat["density"])
centre = CENTRE
if free:
    head = 6 if model >= 3 else 4
    cc = np.column_stack([CENTRE[0] + rng.normal(0, 0.003, len(params)), CENTRE[1] + rng.normal(0, 0.003, len(params))])
    params = np.hstack([params[:, :head], cc, params[:, head:]])
    centre = None
ctx = native.default_context()
g64 = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=model, centre=centre, **kw)
want = g64.loglike(params)
print("seed", seed, kind, "n", n, "w", w, "lnL median", np.median(want))
sep = np.hypot((cat["ra"] - CENTRE[0]) * np.cos(np.radians(CENTRE[1])), cat["dec"] - CENTRE[1]) * 3600
print("sep arcsec min/med", sep.min(), np.median(sep), "verr min", cat["verr"].min(), "v range", cat["v"].min(), cat["v"].max())
for precision in ("f32acc64", "f32"):
    for fast in (1, 0):
        g = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=model, centre=centre, precision=precision, **kw)
        g.set_option("f32_domain", 0); g.set_option("fast_path", fast)
        got = g.loglike(params)
        scale = np.maximum(np.abs(want), max(float(n), 32.0))
        err = np.abs(got - want) / scale
        i = int(np.argmax(err))
        print(precision, "fast", fast, "level", g.fast_level, "in domain", g.f32_in_domain, "kappa", g.f32_condition, "worst err %.3e at walker %d abs %.3e" % (err[i], i, abs(got[i] - want[i])), "row", params[i])
        g.close()
