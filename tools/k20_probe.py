"""Scratch: why does a K = 20 timed region report ~225 us per step when K = 20 back-to-back regions take 205 us?"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native as native, synthetic
from mcmc_dynamics_amd.background import Gaussian
cat = synthetic.make_catalog(1000000, config=3, background=True)
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
lnbg = Gaussian(20.0, 40.0)(cat["v"], cat["verr"])
ctx = native.default_context()
g = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGFIXED, centre=centre,
                   lnlike_bg=lnbg, pmember=cat["pmember"])
pos = synthetic.make_walkers(256, ["v_sys", "sigma_max", "v_maxx", "v_maxy"], cat["truth"], config=3)
g.upload_params(pos)

def region(K, idle_us=0.0, events=0, collect=False):
    g.set_option("timing", events)
    if events:
        g.set_option("timing_stride", 4); g.set_option("timing_reserve", 64)
    for _ in range(1700):
        g.enqueue()
    for _ in range(5):
        g.enqueue()
    g.sync()
    if collect:
        g.timing_collect()
    if idle_us:
        t = time.perf_counter()
        while (time.perf_counter() - t) * 1e6 < idle_us:
            pass
    t0 = time.perf_counter()
    for _ in range(K):
        g.enqueue()
    t_enq = time.perf_counter() - t0
    g.sync()
    dt = time.perf_counter() - t0
    k = None
    if events:
        ms, nl = g.timing_collect(); k = ms * 1e3 / max(1, nl)
    return dt * 1e6 / K, t_enq * 1e6 / K, k

for rep in range(2):
    for label, kw in (("plain", {}), ("idle 100us", dict(idle_us=100)), ("idle 1ms", dict(idle_us=1000)), ("idle 10ms", dict(idle_us=10000)),
                      ("events stride 4", dict(events=2)), ("events + collect before", dict(events=2, collect=True))):
        for K in (20, 100):
            step, enq, k = region(K, **kw)
            print("{0:26s} K {1:3d}: {2:6.1f} us/step (enqueue loop {3:4.1f} us/step){4}".format(
                label, K, step, enq, "" if k is None else "  kernel {0:.1f}".format(k)), flush=True)
