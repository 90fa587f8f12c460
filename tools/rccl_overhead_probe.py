"""Scratch: per-step cost of the RCCL all-reduce call site on ONE rank (MCD_FORCE_RCCL=1), C4 shard size."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native, synthetic
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
names4 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]
c = synthetic.make_catalog(1250000, config=4)
pos = synthetic.make_walkers(256, names4, c["truth"], config=4)
os.environ["MCD_FORCE_RCCL"] = "1"
ctx_r = _native.Context(rank=0, n_ranks=1, unique_id=_native.Context.unique_id(), device=0)
del os.environ["MCD_FORCE_RCCL"]
ctx_p = _native.default_context()
for label, ctx in (("no collective", ctx_p), ("1-rank ncclAllReduce per step", ctx_r)):
    g = _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST, centre=centre)
    g.upload_params(pos)
    for _ in range(50): g.enqueue()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(500): g.enqueue()
    g.sync()
    dt = (time.perf_counter() - t0) / 500
    t0 = time.perf_counter()
    for _ in range(200): g.loglike(pos)
    ds = (time.perf_counter() - t0) / 200
    print(f"{label:32s} 1.25e6 stars x 256: pipelined {dt*1e6:7.1f} us/step, blocking call {ds*1e6:7.1f} us", flush=True)
    g.close()
