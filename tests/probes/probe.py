"""Scratch perf probe (not part of the product): times the main kernel at BASELINE shapes."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from mcmc_dynamics_amd import _native, synthetic
from oracle import lnprob_numpy as oracle

ctx = _native.default_context()
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
names4 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]

def run(label, cat, pos, n, W, iters=20, bytes_per=32):
    cat.set_option("timing", 1)
    out = cat.loglike(pos)
    cat.upload_params(pos)
    for _ in range(3):
        cat.enqueue()
    cat.sync()
    ks = []
    t0 = time.perf_counter()
    for _ in range(iters):
        cat.enqueue()
    cat.sync()
    t1 = time.perf_counter()
    cat.enqueue(); cat.sync()
    k_ms = cat.last_kernel_ms; d_ms = cat.last_device_ms
    wall = (t1 - t0) / iters
    terms = n * W
    print(f"{label:28s} N={n:8d} W={W:4d} kernel {k_ms*1e3:9.1f} us  device {d_ms*1e3:9.1f} us  pipelined wall {wall*1e6:9.1f} us"
          f"  -> {terms/ (k_ms*1e-3):.3e} terms/s (kernel)  {terms/wall:.3e} terms/s (wall)  alg {terms*bytes_per/(k_ms*1e-3)/1e12:.2f} TB/s  {cat.launch_info()}", flush=True)
    return out

for n in (100000, 1000000):
    c = synthetic.make_catalog(n, config=3, background=True)
    lnbg = oracle.gaussian_background(c["v"], c["verr"], 20.0, 40.0)
    for W in (256, 128):
        pos = synthetic.make_walkers(W, names4, c["truth"], config=3)
        for fast in (1, 0):
            g = _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST, centre=centre)
            g.set_option("fast_path", fast)
            out = run(f"const fixed fast={fast}", g, pos, n, W)
            g.close()
        if n == 100000 and W == 256:
            want = oracle.batched_constant_lnlike(c, pos[:8], *centre)
            print("  rel err vs oracle", np.max(np.abs(out[:8] - want) / np.abs(want)))
        g = _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST_BGFIXED, centre=centre,
                            lnlike_bg=lnbg, pmember=c["pmember"])
        out = run("bgfixed fixed", g, pos, n, W, bytes_per=48)
        g.close()
        if W == 256:
            names7 = names4 + ["v_back", "sigma_back", "f_back"]
            pos7 = synthetic.make_walkers(W, names7, c["truth"], config=3)
            g = _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST_BGGAUSS, centre=centre,
                                density=c["density"])
            run("bggauss fixed", g, pos7, n, W, bytes_per=40)
            g.close()
            names6 = names4 + ["ra_center", "dec_center"]
            pos6 = synthetic.make_walkers(W, names6, c["truth"], config=3)
            g = _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST, centre=None)
            run("const free fast", g, pos6, n, W)
            g.close()
# chunking sweep on the headline shape
c = synthetic.make_catalog(1000000, config=3, background=True)
pos = synthetic.make_walkers(256, names4, c["truth"], config=3)
g = _native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=_native.MODEL_CONST, centre=centre)
for tw in (2048, 4096, 8192, 16384, 32768):
    g.set_option("target_waves", tw)
    run(f"const fast target_waves={tw}", g, pos, 1000000, 256)
