"""Scratch: blocking-call latency with / without the zero-copy path at several sizes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native, synthetic
ctx = _native.default_context()
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
names4 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]
for n in (6284, 100000, 1000000):
    c = synthetic.make_catalog(n, config=2)
    g = _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST, centre=centre)
    for W in (16, 128, 256):
        pos = synthetic.make_walkers(W, names4, c["truth"], config=2)
        res = {}
        for zc in (0, 1, 0, 1):
            g.set_option("zero_copy", zc)
            ref = g.loglike(pos)
            for _ in range(50): g.loglike(pos)
            t0 = time.perf_counter()
            for _ in range(500): out = g.loglike(pos)
            res.setdefault(zc, []).append((time.perf_counter() - t0) / 500 * 1e6)
            assert np.array_equal(out, ref)
        print(f"N={n:8d} W={W:4d}  blocking call us: copies {res[0][0]:7.1f} {res[0][1]:7.1f}   zero-copy {res[1][0]:7.1f} {res[1][1]:7.1f}", flush=True)
    g.close()
