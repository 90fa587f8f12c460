"""Test helper: CPU build of the kernels' per-chunk arithmetic (tests/emul/mcd_emul.cpp + csrc/mcd_math.h).

Test infrastructure only.  Packs star records / walker rows exactly as the device prep kernels do
(csrc/mcd_kernels.hip: prepare_records_kernel, prepare_walkers_kernel), with NumPy trigonometry."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "emul", "mcd_emul.cpp")
INC = os.path.join(ROOT, "mcmc_dynamics_amd", "csrc")
OUT = os.path.join(ROOT, "tests", "emul", "libmcd_emul.so")

DEG = np.pi / 180.0
R0 = 10800.0 / np.pi
HALF_LN_2PI = 0.5 * np.log(2.0 * np.pi)
_lib = None


def lib():
    global _lib
    if _lib is None:
        deps = [SRC, os.path.join(INC, "mcd_math.h")]
        if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", INC, SRC,
                            "-o", OUT], check=True)
        _lib = ctypes.CDLL(OUT)
        _lib.emul_loglike.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    return _lib


def pack_records(cat, model, centre):
    n = len(cat["v"])
    cols = [cat["v"], cat["verr"] * cat["verr"]]
    if centre is None:
        cols += [np.sin(cat["ra"] * DEG), np.cos(cat["ra"] * DEG), np.sin(cat["dec"] * DEG), np.cos(cat["dec"] * DEG)]
    else:
        dra = (cat["ra"] - centre[0]) * DEG
        dr, dc = cat["dec"] * DEG, centre[1] * DEG
        dx = -R0 * np.cos(dr) * np.sin(dra)
        dy = R0 * (np.sin(dr) * np.cos(dc) - np.cos(dr) * np.sin(dc) * np.cos(dra))
        r = np.hypot(dx, dy)
        safe = np.where(r > 0, r, 1.0)
        cols += [np.where(r > 0, dy / safe, 0.0), np.where(r > 0, dx / safe, np.copysign(1.0, dx))]
    if model == 1:
        b, p = cat["lnlike_bg"], cat["pmember"]
        cols += [b, p, 1.0 - p, -(b + HALF_LN_2PI)]
    elif model == 2:
        cols += [cat["density"], np.zeros(n)]
    return np.ascontiguousarray(np.stack(cols, axis=1), dtype=np.float64)


def pack_walkers(params, model, free):
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    w = np.zeros((params.shape[0], lib().emul_kd()))
    w[:, 0] = params[:, 0]
    w[:, 1] = params[:, 1] * params[:, 1]
    w[:, 2], w[:, 3] = params[:, 2], params[:, 3]
    w[:, 5] = w[:, 7] = 1.0
    j = 4
    if free:
        w[:, 4], w[:, 5] = np.sin(params[:, 4] * DEG), np.cos(params[:, 4] * DEG)
        w[:, 6], w[:, 7] = np.sin(params[:, 5] * DEG), np.cos(params[:, 5] * DEG)
        j = 6
    if model == 2:
        w[:, 8], w[:, 9], w[:, 10] = params[:, j], params[:, j + 1] ** 2, params[:, j + 2]
    return np.ascontiguousarray(w)


def loglike(cat, params, model, centre, fast, chunk_len=248):
    rec = pack_records(cat, model, centre)
    assert rec.shape[1] == lib().emul_record_doubles(model, int(centre is None))
    wp = pack_walkers(params, model, centre is None)
    out = np.empty(wp.shape[0])
    rc = lib().emul_loglike(model, int(centre is None), int(fast), rec.shape[0], rec.ctypes.data, wp.ctypes.data,
                            wp.shape[0], chunk_len, out.ctypes.data)
    assert rc == 0
    return out
