"""Scratch: pipelined step time over explicit chunk lengths (option "chunk_len") and the equal / guided schedules.
    python tools/w128_len_sweep.py [walkers] [stars] [const|bgfixed] [len,len,...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native as native, synthetic
from mcmc_dynamics_amd.background import Gaussian

W = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
const = len(sys.argv) > 3 and sys.argv[3] == "const"
lengths = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else (0, 160, 192, 224, 240, 248, 256, 264, 288, 320, 352, 384, 416, 480)
cat = synthetic.make_catalog(n, config=3, background=True)
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
lnbg = Gaussian(20.0, 40.0)(cat["v"], cat["verr"])
if const:
    g = native.Catalog(native.default_context(), cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST, centre=centre)
else:
    g = native.Catalog(native.default_context(), cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGFIXED,
                       centre=centre, lnlike_bg=lnbg, pmember=cat["pmember"])
pos = synthetic.make_walkers(256, ["v_sys", "sigma_max", "v_maxx", "v_maxy"], cat["truth"], config=3)[:W]
g.upload_params(pos)
for _ in range(1500):
    g.enqueue()
g.sync()
for split in (1, 0, 2):
    g.set_option("tail_split", split)
    for length in lengths:
        g.set_option("chunk_len", length)
        best = 1e9
        for rep in range(3):
            g.upload_params(pos)
            for _ in range(40):
                g.enqueue()
            g.sync()
            t0 = time.perf_counter()
            for _ in range(300):
                g.enqueue()
            g.sync()
            best = min(best, (time.perf_counter() - t0) / 300)
        info = g.launch_info()
        print("W {0} tail_split {1} chunk_len {2:4d}: chunks {3:5d} waves {4:6d} step {5:6.1f} us".format(
            W, split, length, info["chunks"], info["chunks"] * ((W + 63) // 64), best * 1e6), flush=True)
