"""Scratch: effect of the guided chunk schedule and target_waves on kernel time (1e6 x 256)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native, synthetic
from mcmc_dynamics_amd.background import Gaussian
ctx = _native.default_context()
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
names4 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]
c = synthetic.make_catalog(1000000, config=3, background=True)
WW = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pos = synthetic.make_walkers(WW, names4, c["truth"], config=3)
cats = {"const": _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST, centre=centre),
        "bgfixed": _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST_BGFIXED, centre=centre,
                                   lnlike_bg=Gaussian(20.0, 40.0)(c["v"], c["verr"]), pmember=c["pmember"])}
ref = {k: g.loglike(pos) for k, g in cats.items()}
for rep in range(1):
    for name, g in cats.items():
        for tw in (6144, 8192, 12288, 16384, 24576):
            for split in (0, 1, 2):
                g.set_option("target_waves", tw); g.set_option("tail_split", split); g.set_option("timing", 2)
                g.upload_params(pos)
                for _ in range(800): g.enqueue()
                g.sync(); g.timing_collect()
                t0 = time.perf_counter()
                for _ in range(300): g.enqueue()
                g.sync()
                dt = (time.perf_counter() - t0) / 300
                k = g.timing_collect()[0] / 300 * 1e3
                ok = np.max(np.abs(g.fetch() - ref[name]) / np.abs(ref[name]))
                print(f"{name:8s} target_waves={tw:6d} tail_split={split}  kernel {k:7.1f} us  step {dt*1e6:7.1f} us  chunks {g.launch_info()['chunks']:5d}  rel diff {ok:.1e}", flush=True)
