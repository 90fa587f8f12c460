"""Scratch: the seeded binned block (C5 shape) -- where the time of a step goes: with / without chain rows, block lengths."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import DataReader, synthetic
from mcmc_dynamics_amd.analysis import BinnedConstantFit

n, W = 1000000, 512
cat = synthetic.make_catalog(n, config=5)
reader = DataReader({k: cat[k] for k in ("ra", "dec", "v", "verr")})
reader.make_radial_bins(synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG, nstars=1000, dlogr=0.05)
fit = BinnedConstantFit(reader)
fit.parameters["ra_center"].set(value=synthetic.CENTER_RA_DEG, fixed=True)
fit.parameters["dec_center"].set(value=synthetic.CENTER_DEC_DEG, fixed=True)
B = fit.n_bins
pos1 = synthetic.make_walkers(W, ["v_sys", "sigma_max", "v_maxx", "v_maxy"], cat["truth"], config=5)
pos = np.ascontiguousarray(np.broadcast_to(pos1, (B,) + pos1.shape))
lnp = np.ascontiguousarray(fit.lnprob_batch(pos))
print("env", {k: v for k, v in os.environ.items() if k.startswith("MCD_")}, flush=True)
for steps in (64, 256):
    for rows in ("none", "fresh", "reused"):
        p, l = pos.copy(), lnp.copy()
        acc = np.zeros((B, W), dtype=np.int64)
        chain = lnpc = None
        if rows != "none":
            chain, lnpc = np.empty((steps, B, W, 4)), np.empty((steps, B, W))
        fit._stretch_block_seeded(p, l, 9, 0, steps, chain, lnpc, acc)          # warm (arena)
        ts = []
        for rep in range(4):
            if rows == "fresh":
                chain, lnpc = np.empty((steps, B, W, 4)), np.empty((steps, B, W))
            t0 = time.perf_counter()
            fit._stretch_block_seeded(p, l, 9, steps * (rep + 1), steps, chain, lnpc, acc)
            ts.append((time.perf_counter() - t0) / steps * 1e6)
        print("steps {0:4d} rows {1:7s}: {2} us per step".format(steps, rows, " ".join("%.1f" % t for t in ts)), flush=True)
