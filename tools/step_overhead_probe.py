"""Scratch: fixed cost of a timed region (sync; K pipelined steps; sync), blocking-call time, and the chunk schedule
(option "target_waves") for 256 and 128 walkers.  C3 shape by default.
Result (MI355X): K = 1: 216 us, K >= 20: 205 us per step for a 203 us kernel + 6 us reduce; polling the stream with
hipStreamQuery before hipStreamSynchronize changes nothing (the runtime already spins)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native as native, synthetic
from mcmc_dynamics_amd.background import Gaussian

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
cat = synthetic.make_catalog(n, config=3, background=True)
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
lnbg = Gaussian(20.0, 40.0)(cat["v"], cat["verr"])
ctx = native.default_context()
g = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGFIXED, centre=centre,
                   lnlike_bg=lnbg, pmember=cat["pmember"])
pos = synthetic.make_walkers(256, ["v_sys", "sigma_max", "v_maxx", "v_maxy"], cat["truth"], config=3)
g.upload_params(pos)
for _ in range(2000):
    g.enqueue()
g.sync()
for K in (1, 2, 5, 20, 200):
    best = 1e9
    for rep in range(7):
        g.sync()
        t0 = time.perf_counter()
        for _ in range(K):
            g.enqueue()
        g.sync()
        best = min(best, time.perf_counter() - t0)
    print("K {0:4d}: {1:9.1f} us total, {2:7.1f} us per step".format(K, best * 1e6, best * 1e6 / K))
for W in (256, 128, 64):
    for tw in (3072, 4096, 6144, 8192, 10240, 12288, 16384, 24576):
        g.set_option("target_waves", tw)
        p = pos[:W]
        g.upload_params(p)
        for _ in range(50):
            g.enqueue()
        g.sync()
        t0 = time.perf_counter()
        for _ in range(300):
            g.enqueue()
        g.sync()
        step = (time.perf_counter() - t0) / 300
        info = g.launch_info()
        ts = []
        for _ in range(100):
            t0 = time.perf_counter(); g.loglike(p); ts.append(time.perf_counter() - t0)
        print("W {0:3d} target_waves {1:5d}: chunks {2:5d}  step {3:6.1f} us  ({4:.3e} terms/s)  blocking call median {5:6.1f} us".format(
            W, tw, info["chunks"], step * 1e6, n * W / step, np.median(ts) * 1e6), flush=True)
