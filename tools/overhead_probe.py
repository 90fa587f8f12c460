"""Scratch: pipelined step time with and without per-launch HIP events; blocking-call latency breakdown."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native, synthetic
from mcmc_dynamics_amd.background import Gaussian
ctx = _native.default_context()
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
names4 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]
for n, model in ((1000000, "bgfixed"), (1000000, "const"), (100000, "const"), (6284, "const")):
    c = synthetic.make_catalog(n, config=3, background=True)
    for W in (256, 128):
        pos = synthetic.make_walkers(W, names4, c["truth"], config=3)
        if model == "bgfixed":
            g = _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST_BGFIXED, centre=centre,
                                lnlike_bg=Gaussian(20.0, 40.0)(c["v"], c["verr"]), pmember=c["pmember"])
        else:
            g = _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST, centre=centre)
        g.upload_params(pos)
        res = {}
        for mode in (0, 2, 0, 2):
            g.set_option("timing", mode)
            for _ in range(20): g.enqueue()
            g.sync()
            if mode == 2: g.timing_collect()
            t0 = time.perf_counter()
            for _ in range(200): g.enqueue()
            g.sync()
            dt = (time.perf_counter() - t0) / 200
            k = g.timing_collect()[0] / 200 * 1e3 if mode == 2 else float("nan")
            res.setdefault(mode, []).append((dt * 1e6, k))
        g.set_option("timing", 0)
        t0 = time.perf_counter()
        for _ in range(200): g.loglike(pos)
        sync = (time.perf_counter() - t0) / 200 * 1e6
        print(f"{model:8s} N={n:8d} W={W:4d}  pipelined us/step no-events {res[0][0][0]:8.1f} {res[0][1][0]:8.1f}   with events {res[2][0][0]:8.1f} {res[2][1][0]:8.1f} (kernel {res[2][0][1]:7.1f} {res[2][1][1]:7.1f})   blocking call {sync:8.1f} us", flush=True)
        g.close()
