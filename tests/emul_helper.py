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
                os.path.join(INC, "mcd_exp_table.h"), os.path.join(INC, "mcd_chunks.h"), os.path.join(INC, "mcd_stretch.h"), os.path.join(INC, "mcd_rng.h")]
        if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
            subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", INC, SRC,
                            "-o", OUT], check=True)
        _lib = ctypes.CDLL(OUT)
        _lib.emul_loglike.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
        _lib.emul_per_star.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p]
    return _lib


PROFILE_MODELS = (3, 4, 5, 6)
BG_OF = {0: 0, 1: 1, 2: 2, 3: 0, 4: 2, 5: 3, 6: 1}


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
            cols += [b, p, 1.0 - p, np.fmax(np.log(p) - (b + HALF_LN_2PI), -2000.0)]
    elif bg == 2:
        cols += [cat["density"], np.zeros(n)]
    elif bg == 3:
        b = cat["lnlike_bg"]
        with np.errstate(divide="ignore", invalid="ignore"):
            cols += [b, np.fmax(np.log(cat["density"]) - (b + HALF_LN_2PI), -2000.0), cat["density"], np.zeros(n)]
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


# ---- work decomposition (csrc/mcd_chunks.h) ---------------------------------------------------------------------
def plan_chunks(bin_offsets, star_begin, n, n_walkers, target_waves=12288, tail_split=1, exceptions=(), balance=0):
    """The chunk table the library builds for the shard [star_begin, star_begin + n): dict of arrays + scalars."""
    L = lib()
    offs = np.ascontiguousarray(bin_offsets, dtype=np.int64)
    exc = np.ascontiguousarray(exceptions, dtype=np.int64)
    cap = int(n) // 8 + 8 * len(offs) + 64
    begin, count, pset = np.empty(cap, np.int64), np.empty(cap, np.int32), np.empty(cap, np.int32)
    general, offsets, info = np.empty(cap, np.uint8), np.empty(len(offs), np.int64), np.zeros(7, np.int64)
    L.emul_plan_chunks.restype = ctypes.c_int64
    L.emul_plan_chunks.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                   ctypes.c_int64, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64] + \
        [ctypes.c_void_p] * 6 + [ctypes.c_int]
    nc = L.emul_plan_chunks(len(offs) - 1, offs.ctypes.data, int(star_begin), int(n), int(n_walkers), int(target_waves),
                            int(tail_split), len(exc), exc.ctypes.data, cap, begin.ctypes.data, count.ctypes.data,
                            pset.ctypes.data, general.ctypes.data, offsets.ctypes.data, info.ctypes.data, int(balance))
    assert nc >= 0, "chunk capacity"
    return {"begin": begin[:nc], "count": count[:nc], "pset": pset[:nc], "general": general[:nc], "offsets": offsets,
            "max_chunks_per_pset": int(info[0]), "len": int(info[1]), "uniform_len": int(info[2]),
            "has_general": bool(info[3]), "grid": int(info[4]), "uniform_extra": int(info[5]), "balanced_m": int(info[6])}


def shard_range(n_stars, i, n_shards):
    out = np.empty(2, np.int64)
    lib().emul_shard_range.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib().emul_shard_range(int(n_stars), int(i), int(n_shards), out.ctypes.data)
    return int(out[0]), int(out[1])


def pset_background_sums(lnbg, bin_offsets, star_begin, n):
    offs = np.ascontiguousarray(bin_offsets, dtype=np.int64)
    lnbg = np.ascontiguousarray(lnbg, dtype=np.float64)
    out = np.empty(len(offs) - 1)
    lib().emul_pset_background_sums.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                                                ctypes.c_int64, ctypes.c_void_p]
    lib().emul_pset_background_sums(lnbg.ctypes.data, len(offs) - 1, offs.ctypes.data, int(star_begin), int(n),
                                    out.ctypes.data)
    return out


def narrow_exceptions(cat, model):
    """Ascending star indices that rule out the narrow-range variant for their chunk; None when the catalogue has too
    many of them (the library then uses the general fast form throughout)."""
    cols = [np.ascontiguousarray(cat[k], dtype=np.float64) if k in cat else None
            for k in ("v", "verr", "lnlike_bg", "pmember", "density")]
    n = len(cat["v"])
    out = np.empty(n, np.int64)
    lib().emul_narrow_exceptions.restype = ctypes.c_int64
    lib().emul_narrow_exceptions.argtypes = [ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_void_p]
    m = lib().emul_narrow_exceptions(model, n, *[c.ctypes.data if c is not None else None for c in cols], n, out.ctypes.data)
    return None if m < 0 else out[:m].copy()


def sharded_loglike(cat, params, model, centre, level, n_shards, bin_offsets=None, target_waves=12288, tail_split=1,
                    n_walkers=None):
    """A complete evaluation the way the library carries it out over `n_shards` devices / ranks (chunk tables per shard,
    per-chunk kernel family, per-shard background sums, sum over shards).  params: (W, K) or (B, W, K) in C-ABI order.
    Returns (out, number of chunks that took the general form)."""
    rec = pack_records(cat, model, centre)
    n = rec.shape[0]
    offs = np.ascontiguousarray([0, n] if bin_offsets is None else bin_offsets, dtype=np.int64)
    n_psets = len(offs) - 1
    p = np.asarray(params, dtype=np.float64)
    if p.ndim == 2:
        p = p[None]
    assert p.shape[0] == n_psets
    W = p.shape[1]
    wp = np.ascontiguousarray(np.stack([pack_walkers(p[b], model, centre is None) for b in range(n_psets)]))
    exc = narrow_exceptions(cat, model) if level == 2 else np.empty(0, np.int64)
    if exc is None:
        level, exc = 1, np.empty(0, np.int64)
    lnbg = np.ascontiguousarray(cat["lnlike_bg"], dtype=np.float64) if "lnlike_bg" in cat and BG_OF[model] in (1, 3) else None
    out = np.empty((n_psets, W))
    n_general = ctypes.c_int64(0)
    L = lib()
    L.emul_sharded_loglike.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p,
                                       ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p]
    rc = L.emul_sharded_loglike(model, int(centre is None), int(level), n, rec.ctypes.data, n_psets, offs.ctypes.data,
                                lnbg.ctypes.data if lnbg is not None else None, len(exc), exc.ctypes.data, W,
                                wp.ctypes.data, int(n_shards), int(target_waves), int(tail_split), out.ctypes.data,
                                ctypes.byref(n_general))
    assert rc == 0
    return (out[0] if n_psets == 1 else out), n_general.value


def f32_domain(cat, params, model, centre=None):
    """csrc/mcd_guard.h: f32_domain on host columns and a C-ABI-ordered table: (inside, kappa_v, kappa_theta, sep_harm, reason).
    ``centre=None``: free centre (the table then carries the centre columns)."""
    L = lib()
    n = len(cat["v"])
    cols = {k: (np.ascontiguousarray(cat[k], dtype=np.float64) if k in cat and cat[k] is not None else None)
            for k in ("ra", "dec", "v", "verr", "lnlike_bg", "pmember", "density")}
    ptr = lambda a: a.ctypes.data if a is not None else None
    p = np.ascontiguousarray(params, dtype=np.float64)
    kappa = np.zeros(3)
    reason = ctypes.create_string_buffer(400)
    L.emul_f32_domain.restype = ctypes.c_int
    L.emul_f32_domain.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 7 + \
        [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    rc_, dc_ = (0.0, 0.0) if centre is None else (float(centre[0]), float(centre[1]))
    inside = L.emul_f32_domain(int(model), int(centre is None), n, ptr(cols["ra"]), ptr(cols["dec"]), ptr(cols["v"]),
                               ptr(cols["verr"]), ptr(cols["lnlike_bg"]), ptr(cols["pmember"]), ptr(cols["density"]),
                               p.shape[1], p.ctypes.data, p.shape[0], kappa.ctypes.data, reason, 400, rc_, dc_)
    return bool(inside), float(kappa[0]), float(kappa[1]), float(kappa[2]), reason.value.decode()


def philox(counter, key):
    """Philox4x64-10 of csrc/mcd_rng.h: counter (4 words), key (2 words) -> 4 words."""
    c = np.ascontiguousarray(counter, dtype=np.uint64)
    k = np.ascontiguousarray(key, dtype=np.uint64)
    out = np.empty(4, dtype=np.uint64)
    lib().emul_philox.argtypes = [ctypes.c_void_p] * 3
    lib().emul_philox.restype = None
    lib().emul_philox(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def det_log(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().emul_det_log.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]
    lib().emul_det_log.restype = None
    lib().emul_det_log(x.size, x.ctypes.data, out.ctypes.data)
    return out


def chain_numbers(seed, step0, n_steps, n_bins, n_walkers, n_dim):
    """order [n][B][W], zz / thr / pick [n][2][B][W/2] of steps step0 .. step0 + n - 1 (host build of csrc/mcd_rng.h)."""
    B, W, half = max(int(n_bins), 1), int(n_walkers), int(n_walkers) // 2
    order = np.empty((n_steps, B, W), dtype=np.int32)
    zz = np.empty((n_steps, 2, B, half))
    thr = np.empty((n_steps, 2, B, half))
    pick = np.empty((n_steps, 2, B, half), dtype=np.int32)
    L = lib()
    L.emul_chain_numbers.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int] + \
        [ctypes.c_void_p] * 4
    L.emul_chain_numbers.restype = None
    L.emul_chain_numbers(seed, step0, n_steps, B, W, n_dim, order.ctypes.data, zz.ctypes.data, thr.ctypes.data, pick.ctypes.data)
    return order, zz, thr, pick


def chain_keys(seed, step, b, n_walkers):
    """The 44-bit ordering keys of the walkers of ensemble b in one step."""
    out = np.empty(int(n_walkers), dtype=np.uint64)
    L = lib()
    L.emul_chain_keys.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    L.emul_chain_keys.restype = None
    L.emul_chain_keys(seed, step, b, n_walkers, out.ctypes.data)
    return out
