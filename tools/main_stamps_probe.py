"""Scratch: where the time of ONE main-kernel launch goes, from device timestamps of every workgroup (entry, just before
the partial-sum store; s_memrealtime, 100 MHz) and the XCD / CU each ran on.
    make -C mcmc_dynamics_amd/csrc variant NAME=stamps DEFS=-DMCD_MAIN_STAMPS
    MCD_LIB_PATH=$PWD/mcmc_dynamics_amd/libmcd_hip_stamps.so python tools/main_stamps_probe.py [stars] [walkers] [const|bgfixed] [chunk_len,...]"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import _native as native, synthetic
from mcmc_dynamics_amd.background import Gaussian

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 256
const = not (len(sys.argv) > 3 and sys.argv[3] == "bgfixed")
lengths = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
cat = synthetic.make_catalog(n, config=3, background=True)
centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
if const:
    g = native.Catalog(native.default_context(), cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST, centre=centre)
else:
    lnbg = Gaussian(20.0, 40.0)(cat["v"], cat["verr"])
    g = native.Catalog(native.default_context(), cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGFIXED,
                       centre=centre, lnlike_bg=lnbg, pmember=cat["pmember"])
pos = synthetic.make_walkers(max(W, 256), ["v_sys", "sigma_max", "v_maxx", "v_maxy"], cat["truth"], config=3)[:W]
lib = native.load_library()
lib.mcd_debug_main_stamps.restype = ctypes.c_int
lib.mcd_debug_main_stamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
g.upload_params(pos)
for _ in range(3000):
    g.enqueue()
g.sync()
for split in (1, 0):
    g.set_option("tail_split", split)
    for length in lengths:
        g.set_option("chunk_len", length)
        g.upload_params(pos)
        for _ in range(200):
            g.enqueue()
        g.sync()
        t0 = time.perf_counter()
        for _ in range(500):
            g.enqueue()
        g.sync()
        step = (time.perf_counter() - t0) / 500 * 1e6
        info = g.launch_info()
        nb = int(info["workgroups"])
        st = np.zeros((nb, 4), dtype=np.uint64)
        rc = lib.mcd_debug_main_stamps(st.ctypes.data_as(ctypes.c_void_p), nb)
        assert rc == 0, rc
        t_in = st[:, 0].astype(np.int64)
        t_out = st[:, 1].astype(np.int64)
        first = t_in.min()
        xcc = (st[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
        hw = st[:, 2].astype(np.int64) & 0xffffffff
        cu = (hw >> 8) & 0xf
        sh = (hw >> 12) & 0x1
        se = (hw >> 13) & 0x7
        place = xcc * 1000 + se * 100 + sh * 10 + cu
        per_cu = np.bincount(np.unique(place, return_inverse=True)[1])
        dur = (t_out - t_in) / 100.0
        print("stars {0} W {1} tail_split {2} chunk_len {3}: chunks {4} workgroups {5} step {6:.2f} us".format(
            n, W, split, length, info["chunks"], nb, step))
        print("   entry: first 0, median {0:.2f}, last {1:.2f} us after the first;  exit: first {2:.2f}, median {3:.2f}, last {4:.2f}".format(
            np.median(t_in - first) / 100.0, (t_in.max() - first) / 100.0, (t_out.min() - first) / 100.0,
            np.median(t_out - first) / 100.0, (t_out.max() - first) / 100.0))
        print("   workgroup duration: min {0:.2f} median {1:.2f} max {2:.2f} us;  distinct CUs {3}, workgroups per CU min {4} max {5};  per XCD {6}".format(
            dur.min(), np.median(dur), dur.max(), per_cu.size, per_cu.min(), per_cu.max(), np.bincount(xcc, minlength=8).tolist()))
        # by entry order: the k-th decile of entry time and the duration of the workgroups that entered then
        order = np.argsort(t_in)
        dec = np.array_split(order, 8)
        print("   by entry octile: entry us " + " ".join("{0:.2f}".format(np.median(t_in[d] - first) / 100.0) for d in dec) +
              " | duration us " + " ".join("{0:.2f}".format(np.median(dur[d])) for d in dec), flush=True)
