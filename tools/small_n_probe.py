"""Scratch: target_waves sweep for small catalogues (C2: 1e5 stars x 256 walkers, no background; and 2e4 stars)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native, synthetic
ctx = _native.default_context()
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
names4 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]
for n in (100000, 20000, 6284):
    c = synthetic.make_catalog(n, config=2)
    for WW in (256, 128):
        pos = synthetic.make_walkers(WW, names4, c["truth"], config=2)
        g = _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST, centre=centre)
        ref = g.loglike(pos)
        for tw in (1024, 2048, 3072, 4096, 6144, 8192, 12288, 16384):
            g.set_option("target_waves", tw); g.set_option("timing", 2)
            g.upload_params(pos)
            for _ in range(50): g.enqueue()
            g.sync(); g.timing_collect()
            t0 = time.perf_counter()
            for _ in range(500): g.enqueue()
            g.sync()
            dt = (time.perf_counter() - t0) / 500
            k = g.timing_collect()[0] / 500 * 1e3
            g.set_option("timing", 0)
            t0 = time.perf_counter()
            for _ in range(200): out = g.loglike(pos)
            sync = (time.perf_counter() - t0) / 200
            ok = np.max(np.abs(out - ref) / np.abs(ref))
            print("n %7d W %3d target_waves %5d  step %.1f us  kernel %.1f us  blocking call %.1f us  chunks %s  err %.1e" % (
                n, WW, tw, dt * 1e6, k, sync * 1e6, g.launch_info()["chunks"], ok), flush=True)
