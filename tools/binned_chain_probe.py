"""Scratch: where a step of the binned (C5-shaped) sampler goes: the draws, the library's block, one blocking evaluation."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import DataReader, synthetic
from mcmc_dynamics_amd.analysis import BinnedConstantFit
from mcmc_dynamics_amd.analysis.binned import BinnedSampler

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 512
cat = synthetic.make_catalog(n, config=5)
reader = DataReader({k: cat[k] for k in ("ra", "dec", "v", "verr")})
reader.make_radial_bins(synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG, nstars=1000, dlogr=0.05)
fit = BinnedConstantFit(reader)
fit.parameters["ra_center"].set(value=synthetic.CENTER_RA_DEG, fixed=True)
fit.parameters["dec_center"].set(value=synthetic.CENTER_DEC_DEG, fixed=True)
B = fit.n_bins
pos1 = synthetic.make_walkers(W, ["v_sys", "sigma_max", "v_maxx", "v_maxy"], cat["truth"], config=5)
pos = np.ascontiguousarray(np.broadcast_to(pos1, (B,) + pos1.shape))
for bs in (64, 128, 256):
    s = BinnedSampler(B, W, 4, fit.lnprob_batch, seed=1, block_fn=fit._stretch_block)
    s.block_steps = bs
    state = s.run_mcmc(pos, bs + 6)
    t0 = time.perf_counter(); s.run_mcmc(state[0], 512, log_prob0=state[1]); dt = time.perf_counter() - t0
    print("B {0} W {1} block_steps {5}: library blocks {2:.1f} us per step ({3:.0f} steps/s) {4}".format(B, W, dt / 512 * 1e6, 512 / dt, fit._catalog.stretch_info(), bs), flush=True)
for bs in (64, 256):
    s = BinnedSampler(B, W, 4, fit.lnprob_batch, seed=1, rng="device", seeded_block_fn=fit._stretch_block_seeded)
    s.device_block_steps = bs
    state = s.run_mcmc(pos, bs)
    t0 = time.perf_counter(); s.run_mcmc(state[0], 512, log_prob0=state[1]); dt = time.perf_counter() - t0
    print("B {0} W {1} block_steps {5}: SEEDED library blocks {2:.1f} us per step ({3:.0f} steps/s) {4}".format(B, W, dt / 512 * 1e6, 512 / dt, fit._catalog.stretch_info(), bs), flush=True)
    s.close()
if len(sys.argv) > 3 and sys.argv[3] == "seeded-only":
    sys.exit(0)
half = W // 2
tab = np.ascontiguousarray(np.broadcast_to(pos1[:half], (B, half, 4)))
g = fit._catalog
for _ in range(5): g.loglike(tab)
ts = []
for _ in range(30):
    t0 = time.perf_counter(); g.loglike(tab); ts.append(time.perf_counter() - t0)
print("blocking evaluation of (B, W/2) rows: median {0:.1f} us".format(np.median(ts) * 1e6), flush=True)
# the block call alone, numbers prepared beforehand
rng = np.random.RandomState(3)
def numbers(k):
    order = np.argsort(rng.rand(k, B, W), axis=2).astype(np.int32)
    u = rng.rand(k, 4, B, half)
    zz = np.ascontiguousarray((u[:, :2] + 1.0) ** 2 / 2.0)
    thr = np.ascontiguousarray(np.log(u[:, 2:]) - 3.0 * np.log(zz))
    return order, zz, thr, rng.randint(half, size=(k, 2, B, half)).astype(np.int32)
t0 = time.perf_counter(); r = numbers(64); print("draws of 64 steps on one thread: {0:.1f} ms".format((time.perf_counter() - t0) * 1e3), flush=True)
p, l = state[0].copy(), state[1].copy()
chain, lnpc, acc = np.empty((64, B, W, 4)), np.empty((64, B, W)), np.zeros((B, W), dtype=np.int64)
fit._stretch_block(p, l, *r, chain, lnpc, acc)
t0 = time.perf_counter(); fit._stretch_block(p, l, *r, chain, lnpc, acc); dt = time.perf_counter() - t0
print("library block of 64 steps alone: {0:.1f} us per step".format(dt / 64 * 1e6), flush=True)
# the sampler's own cost: draws on four threads one block ahead, bookkeeping -- with a library call that does nothing
s = BinnedSampler(B, W, 4, fit.lnprob_batch, seed=1, block_fn=lambda *a: None)
state = s.run_mcmc(pos, 70)
t0 = time.perf_counter(); s.run_mcmc(state[0], 512, log_prob0=state[1]); dt = time.perf_counter() - t0
print("sampler without the library call: {0:.1f} us per step".format(dt / 512 * 1e6), flush=True)
