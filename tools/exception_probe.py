"""Scratch: cost of narrow-range exceptions on C3 (1e6 stars x 256 walkers, fixed background): kernel time with no
exception star, a handful (10 certain members + 10 extreme-background stars) and 1 % of the catalogue."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native, synthetic
from mcmc_dynamics_amd.background import Gaussian
ctx = _native.default_context()
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
names4 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]
c = synthetic.make_catalog(1000000, config=3, background=True)
pos = synthetic.make_walkers(256, names4, c["truth"], config=3)
lnbg0 = Gaussian(20.0, 40.0)(c["v"], c["verr"])
rng = np.random.default_rng(1)
for label, n_exc in (("no exception", 0), ("20 exception stars", 20), ("1 % exception stars", 10000), ("13 % (general form)", 130000)):
    pm, lnbg = c["pmember"].copy(), lnbg0.copy()
    idx = rng.choice(len(pm), size=n_exc, replace=False)
    pm[idx[: n_exc // 2]] = 1.0
    lnbg[idx[n_exc // 2:]] = -400.0
    g = _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST_BGFIXED, centre=centre,
                        lnlike_bg=lnbg, pmember=pm)
    g.set_option("timing", 2)
    g.upload_params(pos)
    for _ in range(1500): g.enqueue()
    g.sync(); g.timing_collect()
    for _ in range(300): g.enqueue()
    g.sync()
    k = g.timing_collect()[0] / 300 * 1e3
    fast = g.fetch(); level = g.fast_level
    g.set_option("fast_path", 0)
    plain = g.loglike(pos)
    print("%-24s level %d  kernel %.1f us  max rel diff to plain %.1e" % (label, level, k, np.max(np.abs(fast - plain) / np.abs(plain))), flush=True)
    g.close()
