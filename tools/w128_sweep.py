"""Scratch: pipelined step time of the C3 catalogue for 128 and 256 walkers over the chunk schedule options."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native as native, synthetic
from mcmc_dynamics_amd.background import Gaussian

n = 1000000
cat = synthetic.make_catalog(n, config=3, background=True)
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
lnbg = Gaussian(20.0, 40.0)(cat["v"], cat["verr"])
g = native.Catalog(native.default_context(), cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGFIXED,
                   centre=centre, lnlike_bg=lnbg, pmember=cat["pmember"])
pos = synthetic.make_walkers(256, ["v_sys", "sigma_max", "v_maxx", "v_maxy"], cat["truth"], config=3)
g.upload_params(pos)
for _ in range(1500):
    g.enqueue()
g.sync()
for W in (128, 256):
    for key, values in (("target_waves", (6144, 8192, 10240, 12288, 14336, 16384, 20480, 24576)), ("tail_split", (0, 1, 2, 3))):
        for v in values:
            g.set_option(key, v)
            p = pos[:W]
            best = 1e9
            for rep in range(3):
                g.upload_params(p)
                for _ in range(40):
                    g.enqueue()
                g.sync()
                t0 = time.perf_counter()
                for _ in range(300):
                    g.enqueue()
                g.sync()
                best = min(best, (time.perf_counter() - t0) / 300)
            print("W {0:3d} {1} {2:6d}: chunks {3:5d} step {4:6.1f} us".format(W, key, v, g.launch_info()["chunks"], best * 1e6), flush=True)
        g.set_option(key, 12288 if key == "target_waves" else 1)
