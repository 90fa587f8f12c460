"""Scratch: step time of the C3 kernel as a function of the nominal chunk length (option "chunk_len") and of the
guided schedule, for 256 / 128 / 64 walkers."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native as native, synthetic
from mcmc_dynamics_amd.background import Gaussian

model = sys.argv[1] if len(sys.argv) > 1 else "bgfixed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
cat = synthetic.make_catalog(n, config=3, background=True)
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
ctx = native.default_context()
if model == "bgfixed":
    lnbg = Gaussian(20.0, 40.0)(cat["v"], cat["verr"])
    g = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGFIXED, centre=centre,
                       lnlike_bg=lnbg, pmember=cat["pmember"])
else:
    g = native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST, centre=centre)
pos = synthetic.make_walkers(512, ["v_sys", "sigma_max", "v_maxx", "v_maxy"], cat["truth"], config=3)
g.upload_params(pos[:256])
for _ in range(2000):
    g.enqueue()
g.sync()
lens = [64, 96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 416, 448, 480, 512, 544, 576, 640, 672, 704, 736, 800, 864, 928, 992, 1056]
for W in (256, 128, 64, 512):
    for split in (1, 0):
        g.set_option("tail_split", split)
        row = []
        for L in lens:
            g.set_option("chunk_len", L)
            g.upload_params(pos[:W])
            for _ in range(30):
                g.enqueue()
            g.sync()
            t0 = time.perf_counter()
            for _ in range(150):
                g.enqueue()
            g.sync()
            row.append((time.perf_counter() - t0) / 150 * 1e6)
        best = int(np.argmin(row))
        print("W {0:3d} tail_split {1}: ".format(W, split) + " ".join("{0}:{1:.0f}".format(L, t) for L, t in zip(lens, row)) +
              "   best {0} ({1:.1f} us, {2:.3e} terms/s)".format(lens[best], row[best], n * W / row[best] * 1e6), flush=True)
