"""Scratch: one block of `mcd_stretch_move` on a C3-shaped catalogue, resident on the device and host-driven, wall clock
per step; under `rocprofv3 --kernel-trace` the trace of the resident block shows the kernel durations and the gaps between
them (tools/chain_trace_summary.py).
    python tools/chain_probe.py [n_stars] [n_walkers] [n_steps] [bgfixed|const]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native as native, synthetic
from mcmc_dynamics_amd.background import Gaussian

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
w = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 64
const = len(sys.argv) > 4 and sys.argv[4] == "const"
cat = synthetic.make_catalog(n, config=3, background=True)
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
lnbg = Gaussian(20.0, 40.0)(cat["v"], cat["verr"])
if const:
    g = native.Catalog(native.default_context(), cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST, centre=centre)
else:
    g = native.Catalog(native.default_context(), cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGFIXED,
                       centre=centre, lnlike_bg=lnbg, pmember=cat["pmember"])
pos = synthetic.make_walkers(w, ["v_sys", "sigma_max", "v_maxx", "v_maxy"], cat["truth"], config=3)
lnp = g.loglike(pos)
plan = {"col_source": np.arange(4, dtype=np.int32), "col_const": np.zeros(4), "col_factor": np.ones(4),
        "lo": np.array([-np.inf, 0.0, -np.inf, -np.inf]), "hi": np.full(4, np.inf), "fixed_ok": True}
rng = np.random.default_rng(1)
half = w // 2


def randoms(k):
    order = np.argsort(rng.random((k, w)), axis=1).astype(np.int32)
    u = rng.random((k, 4, half))
    zz = np.ascontiguousarray((u[:, :2] + 1.0) ** 2 / 2.0)
    thr = np.ascontiguousarray(np.log(u[:, 2:]) - 3.0 * np.log(zz))
    return order, zz, thr, rng.integers(0, half, size=(k, 2, half)).astype(np.int32)


for device, fused, defer in ((1, 1, 1), (0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1)):
    g.set_option("device_chain", device)
    g.set_option("fused_reduce", fused)
    g.set_option("defer_guard", defer)
    p, l = pos.copy(), lnp.copy()
    g.stretch_move(plan, p, l, *randoms(8))                      # warm: buffers, clocks
    best = 1e9
    for rep in range(5):
        r = randoms(steps)
        t0 = time.perf_counter()
        g.stretch_move(plan, p, l, *r)
        best = min(best, time.perf_counter() - t0)
    print("device_chain {0} fused_reduce {4} defer_guard {5}: {1:7.1f} us per step ({2:.0f} steps/s), info {3}".format(
        device, best / steps * 1e6, steps / best, g.stretch_info(), fused, defer), flush=True)
