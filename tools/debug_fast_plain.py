"""Scratch: locate the star / walker where fast and plain kernels disagree for a random case."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from mcmc_dynamics_amd import _native
from test_guard_random_cpu import CENTRE, random_case
import emul_helper as emul
ctx = _native.default_context()
model = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.default_rng(500 + model)
def kwargs(cat, sl=slice(None)):
    if model == 1: return dict(lnlike_bg=cat["lnlike_bg"][sl], pmember=cat["pmember"][sl])
    if model in (2, 4): return dict(density=cat["density"][sl])
    if model == 5: return dict(lnlike_bg=cat["lnlike_bg"][sl], density=cat["density"][sl])
    return {}
for trial in range(25):
    cat, params = random_case(rng, model, n=500, w=70)
    g = _native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=model, centre=CENTRE, **kwargs(cat))
    fast = g.loglike(params); g.set_option("fast_path", 0); plain = g.loglike(params); g.close()
    ok = np.isfinite(plain)
    err = np.abs(fast[ok] - plain[ok]) / np.abs(plain[ok])
    if err.max() < 1e-11: continue
    w = np.flatnonzero(ok)[np.argmax(err)]
    print("trial", trial, "walker", w, "rel err", err.max(), "abs", fast[w] - plain[w], "params", params[w])
    cpu_fast = emul.loglike(cat, params[w:w+1], model, CENTRE, 1, 64)[0]; cpu_plain = emul.loglike(cat, params[w:w+1], model, CENTRE, 0, 64)[0]
    print("  CPU emul fast-plain", cpu_fast - cpu_plain, " GPU plain - CPU plain", plain[w] - cpu_plain, " GPU fast - CPU fast", fast[w] - cpu_fast)
    worst = []
    for i in range(len(cat["v"])):
        sl = slice(i, i + 1)
        g1 = _native.Catalog(ctx, cat["ra"][sl], cat["dec"][sl], cat["v"][sl], cat["verr"][sl], model=model, centre=CENTRE, **kwargs(cat, sl))
        f = g1.loglike(params[w:w+1])[0]; g1.set_option("fast_path", 0); p = g1.loglike(params[w:w+1])[0]; g1.close()
        if abs(f - p) > 1e-9 * max(1.0, abs(p)): worst.append((abs(f - p), i, f, p))
    worst.sort(reverse=True)
    for d, i, f, p in worst[:5]:
        print("  star", i, "fast", f, "plain", p, "diff", f - p, "v", cat["v"][i], "verr", cat["verr"][i],
              {k: cat[k][i] for k in ("lnlike_bg", "pmember", "density") if k in cat})
    break
