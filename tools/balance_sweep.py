"""Scratch: pipelined step time (main kernel + reduction) of one catalogue shape over the chunk-plan options of round 3 --
the multi-round schedule (balance 0), the balanced single round with m workgroups per CU (balance m), and 8-wave
workgroups that combine their chunks (combine 1) -- with the results checked against the multi-round schedule's.
    python tools/balance_sweep.py [stars] [walkers] [const|bgfixed|bggauss] [m,m,...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native as native, synthetic
from mcmc_dynamics_amd.background import Gaussian

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
kind = sys.argv[3] if len(sys.argv) > 3 else "const"
ms = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, -1, 1, 2, 3, 4, 6, 8]
cat = synthetic.make_catalog(n, config=3, background=True)
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
names = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]
ctx = native.default_context()
if kind == "const":
    g = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST, centre=centre)
elif kind == "bgfixed":
    lnbg = Gaussian(20.0, 40.0)(cat["v"], cat["verr"])
    g = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGFIXED, centre=centre,
                       lnlike_bg=lnbg, pmember=cat["pmember"])
else:
    names = names + ["v_back", "sigma_back", "f_back"]
    g = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGGAUSS, centre=centre,
                       density=cat["density"])
truth = dict(cat["truth"])
truth.setdefault("v_back", 20.0)
truth.setdefault("sigma_back", 40.0)
truth.setdefault("f_back", 0.25)
pos = synthetic.make_walkers(max(W, 256), names, truth, config=3)[:W]
g.set_option("balance", 0)
ref = g.loglike(pos)
g.upload_params(pos)
for _ in range(3000):
    g.enqueue()
g.sync()
for combine in (0, 8, 16):
    for m in ms:
        if combine and (m in (0, 1, 3) or m < 0 or (combine == 16 and m % 4)):
            continue
        g.set_option("combine", combine)
        g.set_option("balance", m)
        got = g.loglike(pos)
        err = float(np.max(np.abs(got - ref) / np.abs(ref)))
        best = 1e9
        for rep in range(3):
            g.upload_params(pos)
            for _ in range(100):
                g.enqueue()
            g.sync()
            t0 = time.perf_counter()
            for _ in range(500):
                g.enqueue()
            g.sync()
            best = min(best, (time.perf_counter() - t0) / 500)
        t0 = time.perf_counter()
        for _ in range(50):
            g.loglike(pos)
        blocking = (time.perf_counter() - t0) / 50
        info = g.launch_info()
        print("{0} {1} x {2} balance {3:2d} combine {4}: chunks {5:5d} workgroups {6:5d} step {7:6.2f} us  blocking {8:6.1f} us  "
              "rel.diff {9:.1e}".format(kind, n, W, m, combine, info["chunks"], info["workgroups"], best * 1e6, blocking * 1e6, err),
              flush=True)
