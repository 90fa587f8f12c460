"""ctypes binding of the C-ABI in ``include/mcd.h`` (``libmcd_hip.so``, built by ``csrc/Makefile``).

This is the only route from the Python host code to the GPU: there is no CPU fallback.  If the
library is missing or no gfx950 device is usable, the functions here raise ``NativeError``.
"""
import atexit
import ctypes
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCD_LIB_PATH: A/B builds of the same library for kernel experiments (csrc/Makefile: variant); default = the in-tree build
LIB_PATH = os.environ.get("MCD_LIB_PATH") or os.path.join(_HERE, "libmcd_hip.so")

MODEL_CONST, MODEL_CONST_BGFIXED, MODEL_CONST_BGGAUSS = 0, 1, 2
MODEL_PROFILE, MODEL_PROFILE_BGGAUSS, MODEL_PROFILE_BGDENS, MODEL_PROFILE_BGFIXED = 3, 4, 5, 6
CENTRE_FIXED, CENTRE_FREE = 0, 1
F64, F32, F32_ACC64 = 0, 1, 2
PRECISIONS = {"f64": F64, "f32": F32, "f32acc64": F32_ACC64}
UNIQUE_ID_BYTES = 128

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int64_p = ctypes.POINTER(ctypes.c_int64)

# every symbol include/mcd.h declares: (restype, argtypes)
SYMBOLS = {
    "mcd_ctx_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_void_p)]),
    "mcd_get_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "mcd_ctx_create_rank": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.POINTER(ctypes.c_void_p)]),
    "mcd_ctx_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mcd_ctx_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]),
    "mcd_ctx_abort": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p]),
    "mcd_ctx_failed": (ctypes.c_int, [ctypes.c_void_p]),
    "mcd_ctx_n_devices": (ctypes.c_int, [ctypes.c_void_p]),
    "mcd_ctx_comm_info": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                         ctypes.POINTER(ctypes.c_int)]),
    "mcd_catalog_create": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "mcd_catalog_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "mcd_catalog_param_count": (ctypes.c_int, [ctypes.c_void_p]),
    "mcd_catalog_n_stars": (ctypes.c_int64, [ctypes.c_void_p]),
    "mcd_catalog_n_outputs": (ctypes.c_int64, [ctypes.c_void_p, ctypes.c_int64]),
    "mcd_loglike_batch": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, _c_double_p, _c_double_p]),
    "mcd_params_upload": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, _c_double_p]),
    "mcd_loglike_enqueue": (ctypes.c_int, [ctypes.c_void_p]),
    "mcd_loglike_fetch": (ctypes.c_int, [ctypes.c_void_p, _c_double_p]),
    "mcd_sync": (ctypes.c_int, [ctypes.c_void_p]),
    "mcd_membership": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, _c_double_p, _c_double_p]),
    "mcd_loglike_per_star": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int32, _c_double_p, _c_double_p]),
    "mcd_kde_background": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int64, _c_double_p, ctypes.c_int64, _c_double_p,
                                          _c_double_p, ctypes.c_double, _c_double_p, _c_double_p]),
    "mcd_stretch_move": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, _c_double_p, _c_double_p,
                                        ctypes.POINTER(ctypes.c_int32), _c_double_p, _c_double_p,
                                        ctypes.POINTER(ctypes.c_int32), _c_double_p, _c_double_p, _c_int64_p]),
    "mcd_stretch_move_seeded": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, _c_double_p, _c_double_p,
                                               ctypes.c_uint64, ctypes.c_int64, _c_double_p, _c_double_p, _c_int64_p]),
    "mcd_chain_numbers": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_int32, ctypes.POINTER(ctypes.c_int32), _c_double_p, _c_double_p,
                                         ctypes.POINTER(ctypes.c_int32)]),
    "mcd_stretch_info": (ctypes.c_int, [ctypes.c_void_p, _c_int64_p, _c_int64_p, _c_int64_p, ctypes.POINTER(ctypes.c_int32)]),
    "mcd_last_error": (ctypes.c_char_p, []),
    "mcd_abi_version": (ctypes.c_int, []),
    "mcd_last_kernel_ms": (ctypes.c_double, [ctypes.c_void_p]),
    "mcd_last_device_ms": (ctypes.c_double, [ctypes.c_void_p]),
    "mcd_timing_collect": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_int64_p]),
    "mcd_rerun_count": (ctypes.c_int64, [ctypes.c_void_p]),
    "mcd_last_prefetch": (ctypes.c_int, [ctypes.c_void_p]),
    "mcd_last_fast_level": (ctypes.c_int, [ctypes.c_void_p]),
    "mcd_last_f32_domain": (ctypes.c_int, [ctypes.c_void_p, _c_double_p, _c_double_p]),
    "mcd_set_option": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64]),
    "mcd_last_launch_info": (ctypes.c_int, [ctypes.c_void_p, _c_int64_p, ctypes.POINTER(ctypes.c_int32), _c_int64_p,
                                            ctypes.POINTER(ctypes.c_int32)]),
}


class NativeError(RuntimeError):
    """Raised when the HIP library is missing, fails to load, or a call returns an error status."""


class CatalogDesc(ctypes.Structure):
    """Mirror of ``mcd_catalog_desc``."""
    _fields_ = [
        ("n_stars", ctypes.c_int64),
        ("ra", _c_double_p), ("dec", _c_double_p), ("v", _c_double_p), ("verr", _c_double_p),
        ("lnlike_bg", _c_double_p), ("pmember", _c_double_p), ("density", _c_double_p),
        ("model", ctypes.c_int32), ("centre", ctypes.c_int32), ("precision", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("ra_center", ctypes.c_double), ("dec_center", ctypes.c_double),
        ("n_bins", ctypes.c_int64), ("bin_offsets", _c_int64_p),
    ]


class StretchDesc(ctypes.Structure):
    """Mirror of ``mcd_stretch_desc``."""
    _fields_ = [
        ("n_walkers", ctypes.c_int64), ("n_dim", ctypes.c_int32), ("k", ctypes.c_int32),
        ("col_source", ctypes.POINTER(ctypes.c_int32)), ("col_const", _c_double_p), ("col_factor", _c_double_p),
        ("lo", _c_double_p), ("hi", _c_double_p), ("fixed_ok", ctypes.c_int32), ("n_bins", ctypes.c_int32),
    ]


# Environment switches the library or this binding reads (INTEGRATION.md lists them).  None is needed in production: they
# select test stand-ins or tuning values, so load_library() says so on the package logger when one is set.
TEST_SWITCHES = ("MCD_LIB_PATH", "MCD_RCCL_LIBRARY", "MCD_ALLOW_SHARED_DEVICE", "MCD_FORCE_RCCL", "MCD_TARGET_WAVES",
                 "MCD_CHAIN_PART_BYTES", "MCD_CHAIN_PARTS", "MCD_COLLECTIVE_TIMEOUT_MS")

_lib = None
_live_catalogs = weakref.WeakSet()
_live_contexts = weakref.WeakSet()


@atexit.register
def _shutdown():
    """Release device objects in dependency order (catalogues, then contexts) while the interpreter
    and the HIP runtime are both still fully alive; nothing is left for __del__ at teardown."""
    for cat in list(_live_catalogs):
        try:
            cat.close()
        except Exception:
            pass
    for ctx in list(_live_contexts):
        try:
            ctx.close()
        except Exception:
            pass


def load_library(path=None):
    """Load ``libmcd_hip.so`` and declare every entry point.  Raises NativeError if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise NativeError(
            "HIP library not found at {0}: build it with `make -C mcmc_dynamics_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback.".format(p))
    try:
        lib = ctypes.CDLL(p)
    except OSError as exc:
        raise NativeError("could not load {0}: {1}".format(p, exc))
    for name, (restype, argtypes) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise NativeError("{0} does not export {1}".format(p, name))
        fn.restype = restype
        fn.argtypes = argtypes
    if path is None:
        _lib = lib
        active = ["{0}={1}".format(k, os.environ[k]) for k in TEST_SWITCHES if os.environ.get(k)]
        if active:
            import logging
            logging.getLogger("mcmc_dynamics_amd").warning(
                "test / tuning switches active (none is needed in production, see INTEGRATION.md): %s", ", ".join(active))
    return lib


def _check(lib, rc, what):
    if rc != 0:
        msg = lib.mcd_last_error()
        raise NativeError("{0} failed (status {1}): {2}".format(what, rc, msg.decode() if msg else ""))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(_c_double_p) if a is not None else None


class Context(object):
    """HIP devices + streams (+ RCCL communicator).  One per process."""

    def __init__(self, n_devices=1, device_ids=None, rank=None, n_ranks=None, unique_id=None, device=None):
        self.lib = load_library()
        handle = ctypes.c_void_p()
        if rank is not None:
            uid = ctypes.create_string_buffer(bytes(unique_id), UNIQUE_ID_BYTES) if unique_id is not None else None
            rc = self.lib.mcd_ctx_create_rank(int(device or 0), int(rank), int(n_ranks), uid, ctypes.byref(handle))
            _check(self.lib, rc, "mcd_ctx_create_rank")
            self.rank, self.n_ranks = int(rank), int(n_ranks)
        else:
            ids = None
            if device_ids is not None:
                ids = (ctypes.c_int * len(device_ids))(*[int(d) for d in device_ids])
                n_devices = len(device_ids)
            rc = self.lib.mcd_ctx_create(int(n_devices), ids, ctypes.byref(handle))
            _check(self.lib, rc, "mcd_ctx_create")
            self.rank, self.n_ranks = 0, 1
        self.handle = handle
        self.host_group = None                    # hostgroup.HostGroup of a multi-rank job (distributed.rank_context)
        self._catalogs = weakref.WeakSet()        # catalogues living on this context: closed before it
        _live_contexts.add(self)

    @staticmethod
    def unique_id():
        lib = load_library()
        buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
        _check(lib, lib.mcd_get_unique_id(buf), "mcd_get_unique_id")
        return buf.raw

    @property
    def n_devices(self):
        return self.lib.mcd_ctx_n_devices(self.handle)

    def set_option(self, key, value):
        """Context options (include/mcd.h): ``collective_timeout_ms`` -- how long a wait on a stream that carries an
        all-reduce may last before the call returns an error and the context is marked failed (0: for ever)."""
        _check(self.lib, self.lib.mcd_ctx_set_option(self.handle, key.encode(), int(value)), "mcd_ctx_set_option")

    def abort(self, reason=""):
        """Make a call of ANOTHER thread that is waiting for a collective on this context return an error now (thread-safe;
        what ``hostgroup.HostGroup`` calls when a peer rank reports a failure)."""
        if getattr(self, "handle", None):
            self.lib.mcd_ctx_abort(self.handle, str(reason).encode()[:400])

    @property
    def failed(self):
        """True after a collective deadline, an abort, or an error in the middle of a resident block: every later call on
        this context raises; the process is expected to exit non-zero (no fallback inside it)."""
        return bool(getattr(self, "handle", None)) and bool(self.lib.mcd_ctx_failed(self.handle))

    def comm_info(self):
        """What RCCL reports for this context's communicator: {'size', 'rank', 'rccl_version'} (size 0: none)."""
        n, r, v = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _check(self.lib, self.lib.mcd_ctx_comm_info(self.handle, ctypes.byref(n), ctypes.byref(r), ctypes.byref(v)),
               "mcd_ctx_comm_info")
        return {"size": n.value, "rank": r.value, "rccl_version": v.value}

    def kde_background(self, comp, v, verr, sigma_int=0.0, return_kernel_ms=False):
        """``background.SingleStars.__call__`` (single_stars.py:42-77) for km/s arrays: (n,) log-likelihoods."""
        if not getattr(self, "handle", None):
            raise NativeError("context is closed")
        comp, v, verr = _f64(comp).ravel(), _f64(v).ravel(), _f64(verr).ravel()
        if v.shape != verr.shape:
            raise ValueError("v and verr must have the same shape")
        out = np.empty(v.size, dtype=np.float64)
        ms = ctypes.c_double(0.0)
        rc = self.lib.mcd_kde_background(self.handle, comp.size, _ptr(comp), v.size, _ptr(v), _ptr(verr),
                                         float(sigma_int), _ptr(out), ctypes.byref(ms))
        _check(self.lib, rc, "mcd_kde_background")
        return (out, ms.value) if return_kernel_ms else out

    def close(self):
        if getattr(self, "handle", None):
            for cat in list(getattr(self, "_catalogs", ())):
                cat.close()
            self.lib.mcd_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    """Process-wide single-GPU context (device 0), created on first use."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(n_devices=1)
    return _default_ctx


def chain_numbers(seed, step0, n_steps, n_bins, n_walkers, n_dim, squeeze=False):
    """``mcd_chain_numbers``: the random numbers of steps ``step0 .. step0 + n_steps - 1`` of the seeded chain (host code, no
    device involved): order (steps, B, W) int32, zz / thr / pick (steps, 2, B, W/2), in the layout ``stretch_move`` takes.
    ``squeeze``: drop the ensemble axis (n_bins <= 1, a catalogue without bins)."""
    lib = load_library()
    b, w = max(int(n_bins), 1), int(n_walkers)
    order = np.empty((n_steps, b, w), dtype=np.int32)
    zz = np.empty((n_steps, 2, b, w // 2), dtype=np.float64)
    thr = np.empty_like(zz)
    pick = np.empty((n_steps, 2, b, w // 2), dtype=np.int32)
    rc = lib.mcd_chain_numbers(int(seed) & 0xFFFFFFFFFFFFFFFF, int(step0), int(n_steps), b, w, int(n_dim),
                               order.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _ptr(zz), _ptr(thr),
                               pick.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    _check(lib, rc, "mcd_chain_numbers")
    if squeeze:
        return order[:, 0], zz[:, :, 0], thr[:, :, 0], pick[:, :, 0]
    return order, zz, thr, pick


class Catalog(object):
    """Star records resident in HBM; evaluates the log-likelihood of batches of walkers."""

    def __init__(self, ctx, ra, dec, v, verr, model=MODEL_CONST, centre=None, lnlike_bg=None, pmember=None,
                 density=None, bin_offsets=None, precision="f64"):
        self.ctx = ctx
        self.lib = ctx.lib
        cols = [_f64(ra), _f64(dec), _f64(v), _f64(verr)]
        n = cols[0].size
        if any(c.size != n for c in cols):
            raise ValueError("ra, dec, v, verr must have the same length")
        extras = [None if a is None else _f64(a) for a in (lnlike_bg, pmember, density)]
        if any(a is not None and a.size != n for a in extras):
            raise ValueError("background columns must have the same length as the catalogue")
        d = CatalogDesc()
        d.n_stars = n
        d.ra, d.dec, d.v, d.verr = (_ptr(c) for c in cols)
        d.lnlike_bg, d.pmember, d.density = (_ptr(a) for a in extras)
        d.model = int(model)
        d.precision = PRECISIONS[precision] if isinstance(precision, str) else int(precision)
        if centre is None:
            d.centre = CENTRE_FREE
        else:
            d.centre = CENTRE_FIXED
            d.ra_center, d.dec_center = float(centre[0]), float(centre[1])
        offs = None
        if bin_offsets is not None and len(bin_offsets) > 2:
            offs = np.ascontiguousarray(bin_offsets, dtype=np.int64)
            d.n_bins = offs.size - 1
            d.bin_offsets = offs.ctypes.data_as(_c_int64_p)
        else:
            d.n_bins = 0
        handle = ctypes.c_void_p()
        rc = self.lib.mcd_catalog_create(ctx.handle, ctypes.byref(d), ctypes.byref(handle))
        _check(self.lib, rc, "mcd_catalog_create")
        self.handle = handle
        self.n_stars = n
        self.n_sets = max(1, int(d.n_bins))
        self.k = self.lib.mcd_catalog_param_count(handle)
        self._walkers = 0
        _live_catalogs.add(self)
        ctx._catalogs.add(self)

    def _params(self, params):
        p = _f64(params)
        if p.ndim == 1:
            p = p[None, :]
        if self.n_sets > 1:
            if p.ndim != 3 or p.shape[0] != self.n_sets:
                raise ValueError("binned catalogue expects params of shape (n_bins, W, K)")
            w = p.shape[1]
        else:
            if p.ndim == 3 and p.shape[0] == 1:
                p = p[0]
            if p.ndim != 2:
                raise ValueError("params must have shape (W, K)")
            w = p.shape[0]
        if p.shape[-1] != self.k:
            raise ValueError("params have {0} columns, catalogue expects {1}".format(p.shape[-1], self.k))
        return np.ascontiguousarray(p), w

    def _alive(self):
        if not getattr(self, "handle", None):
            raise NativeError("catalogue is closed")

    def loglike(self, params):
        """(W, K) -> (W,)   [binned: (B, W, K) -> (B, W)]   synchronous."""
        self._alive()
        p, w = self._params(params)
        out = np.empty((self.n_sets, w) if self.n_sets > 1 else (w,), dtype=np.float64)
        rc = self.lib.mcd_loglike_batch(self.handle, w, self.k, _ptr(p), _ptr(out))
        _check(self.lib, rc, "mcd_loglike_batch")
        self._walkers = w
        return out

    def upload_params(self, params):
        p, w = self._params(params)
        _check(self.lib, self.lib.mcd_params_upload(self.handle, w, self.k, _ptr(p)), "mcd_params_upload")
        self._walkers = w

    def enqueue(self):
        _check(self.lib, self.lib.mcd_loglike_enqueue(self.handle), "mcd_loglike_enqueue")

    def sync(self):
        _check(self.lib, self.lib.mcd_sync(self.handle), "mcd_sync")

    def fetch(self):
        w = self._walkers
        out = np.empty((self.n_sets, w) if self.n_sets > 1 else (w,), dtype=np.float64)
        _check(self.lib, self.lib.mcd_loglike_fetch(self.handle, _ptr(out)), "mcd_loglike_fetch")
        return out

    def membership(self, params_row):
        p = _f64(params_row).reshape(-1)
        out = np.empty(self.n_stars, dtype=np.float64)
        _check(self.lib, self.lib.mcd_membership(self.handle, p.size, _ptr(p), _ptr(out)), "mcd_membership")
        return out

    def loglike_per_star(self, params_row):
        p = _f64(params_row).reshape(-1)
        out = np.empty(self.n_stars, dtype=np.float64)
        _check(self.lib, self.lib.mcd_loglike_per_star(self.handle, p.size, _ptr(p), _ptr(out)), "mcd_loglike_per_star")
        return out

    def _stretch_desc(self, plan, w, p, n_bins):
        cols = [np.ascontiguousarray(plan["col_source"], dtype=np.int32), _f64(plan["col_const"]), _f64(plan["col_factor"]),
                _f64(plan["lo"]), _f64(plan["hi"])]
        if cols[0].size != self.k or cols[1].size != self.k or cols[2].size != self.k or cols[3].size != p or cols[4].size != p:
            raise ValueError("stretch_move: plan does not match the catalogue / the number of free parameters")
        d = StretchDesc()
        d.n_walkers, d.n_dim, d.k, d.n_bins = w, p, self.k, n_bins
        d.col_source = cols[0].ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        d.col_const, d.col_factor, d.lo, d.hi = (_ptr(c) for c in cols[1:])
        d.fixed_ok = 1 if plan.get("fixed_ok", True) else 0
        return d, cols                                          # (cols: keeps the arrays the descriptor points to alive)

    @staticmethod
    def _stretch_outputs(n_steps, lead, w, p, chain, lnprob_chain, accepted):
        for a, shape in ((chain, (n_steps,) + lead + (w, p)), (lnprob_chain, (n_steps,) + lead + (w,))):
            if a is not None and (a.dtype != np.float64 or not a.flags.c_contiguous or a.shape != shape):
                raise ValueError("stretch_move: chain buffers must be C-contiguous float64 of shape (steps, [B,] W, P) / (steps, [B,] W)")
        if accepted is not None and (accepted.dtype != np.int64 or accepted.shape != lead + (w,) or not accepted.flags.c_contiguous):
            raise ValueError("stretch_move: accepted must be a C-contiguous int64 array of shape ([B,] W)")

    def stretch_move(self, plan, pos, lnp, order, zz, thr, pick, chain=None, lnprob_chain=None, accepted=None):
        """``mcd_stretch_move``: advance the ensemble by ``len(order)`` stretch-move steps with the half-step loop inside
        the library.  ``plan``: dict with ``col_source`` (int32 [K]), ``col_const``, ``col_factor`` (float64 [K]), ``lo``,
        ``hi`` (float64 [P]) and ``fixed_ok``.  ``pos`` (W, P) and ``lnp`` (W,) are C-contiguous float64 arrays updated
        in place; random numbers as drawn by ``sampler.EnsembleSampler``.

        Binned catalogues: ``pos`` (B, W, P), ``lnp`` (B, W), ``order`` (steps, B, W), ``zz`` / ``thr`` / ``pick``
        (steps, 2, B, W/2), ``chain`` (steps, B, W, P), ``lnprob_chain`` (steps, B, W), ``accepted`` (B, W): B independent
        ensembles in lockstep, one per radial bin, as ``analysis.binned.BinnedSampler`` draws them."""
        self._alive()
        binned = pos.ndim == 3
        lead = pos.shape[:1] if binned else ()
        n_bins = pos.shape[0] if binned else 1
        n_steps, w = order.shape[0], order.shape[-1]
        p = pos.shape[-1]
        for a, dt in ((pos, np.float64), (lnp, np.float64), (zz, np.float64), (thr, np.float64), (order, np.int32), (pick, np.int32)):
            if a.dtype != dt or not a.flags.c_contiguous:
                raise ValueError("stretch_move needs C-contiguous arrays of the documented dtypes")
        half_shape = (n_steps, 2) + lead + (w // 2,)
        if pos.shape != lead + (w, p) or lnp.shape != lead + (w,) or order.shape != (n_steps,) + lead + (w,) or \
                zz.shape != half_shape or thr.shape != half_shape or pick.shape != half_shape:
            raise ValueError("stretch_move: inconsistent array shapes")
        d, _keep = self._stretch_desc(plan, w, p, n_bins)
        self._stretch_outputs(n_steps, lead, w, p, chain, lnprob_chain, accepted)
        rc = self.lib.mcd_stretch_move(self.handle, ctypes.byref(d), n_steps, _ptr(pos), _ptr(lnp),
                                       order.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _ptr(zz), _ptr(thr),
                                       pick.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), _ptr(chain), _ptr(lnprob_chain),
                                       accepted.ctypes.data_as(_c_int64_p) if accepted is not None else None)
        _check(self.lib, rc, "mcd_stretch_move")
        self._walkers = w // 2

    def stretch_move_seeded(self, plan, pos, lnp, seed, step0, n_steps, chain=None, lnprob_chain=None, accepted=None):
        """``mcd_stretch_move_seeded``: the same block with its random numbers generated inside the library from the
        counter-based generator of csrc/mcd_rng.h -- steps ``step0 .. step0 + n_steps - 1`` of the chain that ``seed`` names.
        ``chain_numbers(seed, step0, n_steps, ...)`` returns the numbers those steps use."""
        self._alive()
        binned = pos.ndim == 3
        lead = pos.shape[:1] if binned else ()
        n_bins = pos.shape[0] if binned else 1
        w, p = pos.shape[-2], pos.shape[-1]
        for a in (pos, lnp):
            if a.dtype != np.float64 or not a.flags.c_contiguous:
                raise ValueError("stretch_move_seeded needs C-contiguous float64 arrays")
        if lnp.shape != lead + (w,) or n_steps < 0 or step0 < 0:
            raise ValueError("stretch_move_seeded: inconsistent array shapes")
        d, _keep = self._stretch_desc(plan, w, p, n_bins)
        self._stretch_outputs(n_steps, lead, w, p, chain, lnprob_chain, accepted)
        rc = self.lib.mcd_stretch_move_seeded(self.handle, ctypes.byref(d), n_steps, _ptr(pos), _ptr(lnp),
                                              int(seed) & 0xFFFFFFFFFFFFFFFF, int(step0), _ptr(chain), _ptr(lnprob_chain),
                                              accepted.ctypes.data_as(_c_int64_p) if accepted is not None else None)
        _check(self.lib, rc, "mcd_stretch_move_seeded")
        self._walkers = w // 2

    @property
    def last_prefetch(self):
        """1 / 0: the last main-kernel launch used / did not use the record-prefetching instantiation; -1 before any launch."""
        return self.lib.mcd_last_prefetch(self.handle)

    def stretch_info(self):
        """Where the blocks of ``stretch_move`` ran: {'device_blocks', 'host_blocks', 'discarded_blocks', 'last_discard_status'}
        (``mcd_stretch_info``: resident on the device / host-driven / discarded by the device and re-run host-driven)."""
        a, b, c, st = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int32()
        _check(self.lib, self.lib.mcd_stretch_info(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(st)),
               "mcd_stretch_info")
        return {"device_blocks": a.value, "host_blocks": b.value, "discarded_blocks": c.value, "last_discard_status": st.value}

    def set_option(self, key, value):
        _check(self.lib, self.lib.mcd_set_option(self.handle, key.encode(), int(value)), "mcd_set_option")

    @property
    def last_kernel_ms(self):
        return self.lib.mcd_last_kernel_ms(self.handle)

    @property
    def last_device_ms(self):
        return self.lib.mcd_last_device_ms(self.handle)

    def timing_collect(self):
        """(summed main-kernel milliseconds, number of launches) since the last collect ("timing" = 2)."""
        total, n = ctypes.c_double(), ctypes.c_int64()
        _check(self.lib, self.lib.mcd_timing_collect(self.handle, ctypes.byref(total), ctypes.byref(n)),
               "mcd_timing_collect")
        return total.value, n.value

    @property
    def rerun_count(self):
        """Batches re-evaluated with the plain kernels (denormal regime of the reference's log-sum-exp)."""
        return self.lib.mcd_rerun_count(self.handle)

    @property
    def fast_level(self):
        """Kernel family of the batch staged last: 0 plain, 1 fast, 2 narrow-range fixed-background variant."""
        return self.lib.mcd_last_fast_level(self.handle)

    @property
    def f32_in_domain(self):
        """Was the table staged last inside the float32 accuracy domain (include/mcd.h)?  Always True for float64."""
        return bool(self.lib.mcd_last_f32_domain(self.handle, None, None))

    @property
    def f32_condition(self):
        """(kappa_v, kappa_theta) of the table staged last: the two condition numbers the float32 domain bounds."""
        kv, kt = ctypes.c_double(), ctypes.c_double()
        self.lib.mcd_last_f32_domain(self.handle, ctypes.byref(kv), ctypes.byref(kt))
        return kv.value, kt.value

    def launch_info(self):
        wg, ch = ctypes.c_int64(), ctypes.c_int64()
        tile, rb = ctypes.c_int32(), ctypes.c_int32()
        _check(self.lib, self.lib.mcd_last_launch_info(self.handle, ctypes.byref(wg), ctypes.byref(tile),
                                                       ctypes.byref(ch), ctypes.byref(rb)), "mcd_last_launch_info")
        return {"workgroups": wg.value, "walker_tile": tile.value, "chunks": ch.value, "record_bytes": rb.value}

    def close(self):
        if getattr(self, "handle", None):
            self.lib.mcd_catalog_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
