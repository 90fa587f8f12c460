"""Scratch: what the per-launch HIP events of "timing" = 2 cost in the pipelined loop (C3 and C3 without background)."""
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
cats = {"const": _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST, centre=centre),
        "bgfixed": _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST_BGFIXED, centre=centre,
                                   lnlike_bg=Gaussian(20.0, 40.0)(c["v"], c["verr"]), pmember=c["pmember"])}
for name, g in cats.items():
    g.upload_params(pos)
    for _ in range(1500): g.enqueue()
    g.sync()
    for mode in (0, 2, 0, 2):
        g.set_option("timing", mode)
        if mode == 2: g.set_option("timing_reserve", 600)
        for _ in range(50): g.enqueue()
        g.sync()
        if mode == 2: g.timing_collect()
        t0 = time.perf_counter()
        for _ in range(500): g.enqueue()
        g.sync()
        dt = (time.perf_counter() - t0) / 500
        print(f"{name:8s} timing={mode}  step {dt*1e6:7.1f} us", flush=True)
