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
        deps = [SRC, os.path.join(INC, "mcd_math.h"), os.path.join(INC, "mcd_guard.h"),
                os.path.join(INC, "mcd_exp_table.h")]
        if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", INC, SRC,
                            "-o", OUT], check=True)
        _lib = ctypes.CDLL(OUT)
        _lib.emul_loglike.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
        _lib.emul_per_star.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]
    return _lib


PROFILE_MODELS = (3, 4, 5)
BG_OF = {0: 0, 1: 1, 2: 2, 3: 0, 4: 2, 5: 3}


def pack_records(cat, model, centre):
    n = len(cat["v"])
    cols = [cat["v"], cat["verr"] * cat["verr"]]
    if centre is None:
        cd = np.cos(cat["dec"] * DEG)
        cols += [cd * np.sin(cat["ra"] * DEG), cd * np.cos(cat["ra"] * DEG), np.sin(cat["dec"] * DEG), np.zeros(n)]
    else:
        dra = (cat["ra"] - centre[0]) * DEG
        dr, dc = cat["dec"] * DEG, centre[1] * DEG
        dx = -R0 * np.cos(dr) * np.sin(dra)
        dy = R0 * (np.sin(dr) * np.cos(dc) - np.cos(dr) * np.sin(dc) * np.cos(dra))
        if model in PROFILE_MODELS:
            xs, ys = 60.0 * dx, 60.0 * dy
            cols += [xs, ys, xs * xs + ys * ys, np.zeros(n)]
        else:
            r = np.hypot(dx, dy)
            safe = np.where(r > 0, r, 1.0)
            cols += [np.where(r > 0, dy / safe, 0.0), np.where(r > 0, dx / safe, np.copysign(1.0, dx))]
    bg = BG_OF[model]
    if bg == 1:
        b, p = cat["lnlike_bg"], cat["pmember"]
        with np.errstate(divide="ignore", invalid="ignore"):
            cols += [b, p, 1.0 - p, np.fmax(np.log(p) - (b + HALF_LN_2PI), -1.0e5)]
    elif bg == 2:
        cols += [cat["density"], np.zeros(n)]
    elif bg == 3:
        b = cat["lnlike_bg"]
        with np.errstate(divide="ignore", invalid="ignore"):
            cols += [b, np.fmax(np.log(cat["density"]) - (b + HALF_LN_2PI), -1.0e5), cat["density"], np.zeros(n)]
    return np.ascontiguousarray(np.stack(cols, axis=1), dtype=np.float64)


def pack_walkers(params, model, free):
    """params in the C-ABI column order (include/mcd.h: mcd_catalog_param_count)."""
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    w = np.zeros((params.shape[0], lib().emul_kd()))
    prof = model in PROFILE_MODELS
    w[:, 0] = params[:, 0]
    w[:, 1] = params[:, 1] * params[:, 1]
    w[:, 2], w[:, 3] = (params[:, 3], params[:, 4]) if prof else (params[:, 2], params[:, 3])
    if prof:
        a, rp = params[:, 2], params[:, 5]
        w[:, 11], w[:, 12], w[:, 13], w[:, 14] = a * a, params[:, 1] ** 2 * a, rp * rp, 2.0 * rp
    w[:, 5] = w[:, 7] = 1.0
    j = 6 if prof else 4
    if free:
        w[:, 4], w[:, 5] = np.sin(params[:, j] * DEG), np.cos(params[:, j] * DEG)
        w[:, 6], w[:, 7] = np.sin(params[:, j + 1] * DEG), np.cos(params[:, j + 1] * DEG)
        j += 2
    if BG_OF[model] == 2:
        w[:, 8], w[:, 9], w[:, 10] = params[:, j], params[:, j + 1] ** 2, params[:, j + 2]
    elif BG_OF[model] == 3:
        w[:, 10] = params[:, j]
    return np.ascontiguousarray(w)


def abi_columns(names, values, model, free):
    """Re-order golden `values` (columns named `names`, the reference's own ordering) into the C-ABI order."""
    prof = model in PROFILE_MODELS
    order = ["v_sys", "sigma_max"] + (["a"] if prof else []) + ["v_maxx", "v_maxy"] + (["r_peak"] if prof else [])
    if free:
        order += ["ra_center", "dec_center"]
    order += {0: [], 1: [], 2: ["v_back", "sigma_back", "f_back"], 3: ["f_back"]}[BG_OF[model]]
    names = [str(n) for n in names]
    return np.ascontiguousarray(np.asarray(values)[:, [names.index(k) for k in order]])


def loglike(cat, params, model, centre, fast, chunk_len=248):
    rec = pack_records(cat, model, centre)
    assert rec.shape[1] == lib().emul_record_doubles(model, int(centre is None))
    wp = pack_walkers(params, model, centre is None)
    out = np.empty(wp.shape[0])
    rc = lib().emul_loglike(model, int(centre is None), int(fast), rec.shape[0], rec.ctypes.data, wp.ctypes.data,
                            wp.shape[0], chunk_len, out.ctypes.data)
    assert rc == 0
    return out


def per_star(cat, params_row, model, centre, mode):
    rec = pack_records(cat, model, centre)
    wp = pack_walkers(np.asarray(params_row)[None, :], model, centre is None)
    out = np.empty(rec.shape[0])
    rc = lib().emul_per_star(model, int(centre is None), int(mode), rec.shape[0], rec.ctypes.data, wp.ctypes.data,
                             out.ctypes.data)
    assert rc == 0
    return out


def fast_guard(cat, params, model, centre, f32=False):
    """The library's per-call range guard (csrc/mcd_guard.h): is a fast formulation admitted?"""
    return fast_level(cat, params, model, centre, f32) > 0


def fast_level(cat, params, model, centre, f32=False):
    """Launch level the library would choose (csrc/mcd_guard.h: fast_level): 0 plain, 1 fast, 2 narrow-range BGFIXED."""
    lib().emul_fast_guard.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 5 + \
        [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
    cols = [np.ascontiguousarray(cat[k], dtype=np.float64) if k in cat else None
            for k in ("v", "verr", "lnlike_bg", "pmember", "density")]
    params = np.ascontiguousarray(np.atleast_2d(params), dtype=np.float64)
    ptr = [c.ctypes.data if c is not None else None for c in cols]
    return int(lib().emul_fast_guard(model, int(centre is None), int(f32), len(cat["v"]), *ptr, params.shape[1],
                                     params.ctypes.data, params.shape[0]))
