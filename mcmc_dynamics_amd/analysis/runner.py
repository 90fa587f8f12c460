"""``Runner``: posterior API and MCMC driver, GPU-backed.

Mirror of the reference's ``mcmc_dynamics/analysis/runner.py`` for the hot path: the method names,
arguments and error behaviour of ``lnprior`` / ``lnlike`` / ``lnprob`` / ``fetch_parameter_values`` /
``get_initials`` / ``__call__`` are kept (runner.py:143-443) so that emcee drives the outer loop
unchanged, and a batched entry ``lnprob_batch((W, P)) -> (W,)`` is added, which is what the sampler
is handed (``vectorize=True``).  The per-star arithmetic of ``_calculate_lnlike`` (runner.py:240-286)
does not exist on the host: it runs in ``libmcd_hip.so`` on the star catalogue pinned in HBM.

Differences that follow from dropping astropy: values are plain floats in the unit recorded in each
``Parameter`` (the reference returns ``Quantity`` objects); ``lnprob`` returns a Python float.
"""
import logging
import pickle
import warnings
from collections import OrderedDict

import numpy as np

from .. import _native, units
from ..background import Gaussian, SingleStars
from ..parameter import Parameters
from ..utils.data_reader import DataReader
from ..utils.results import ResultsTable

logger = logging.getLogger(__name__)


class _BatchPlan(object):
    """Everything `lnprob_batch` needs from `self.parameters`, flattened into index arrays once per
    parameter configuration: the per-call host work is then a handful of NumPy operations on a (W, n_all)
    array instead of a Python loop over parameters (the reference spends ~0.1 ms per walker here,
    SURVEY.md section 8(a) A2/A3)."""

    def __init__(self, runner):
        pars = runner.parameters
        self.names = list(pars)
        index = {n: i for i, n in enumerate(self.names)}
        self.n_all = len(self.names)
        self.free_idx = np.array([i for i, p in enumerate(pars.values()) if not p.fixed], dtype=np.intp)
        fixed = [(i, p) for i, p in enumerate(pars.values()) if p.fixed and p._expr is None]
        self.fixed_idx = np.array([i for i, _ in fixed], dtype=np.intp)
        self.fixed_val = np.array([float(p._value) for _, p in fixed], dtype=np.float64)
        self.simple = all(p._expr is None and p._lnprior is None for p in pars.values())
        self.lo, self.hi = pars.bounds()
        self.unbounded = bool(np.all(np.isneginf(self.lo)) and np.all(np.isposinf(self.hi)))
        # kernel table columns (C-ABI order) and unit factors
        key, _ = runner._catalog_spec()
        cols = list(runner._KERNEL_HEAD)
        if key[1] is None:
            cols += [("ra_center", "deg"), ("dec_center", "deg")]
        cols += list(runner._KERNEL_TAIL)
        self.kernel_idx = np.array([index[n] for n, _ in cols], dtype=np.intp)
        fac = [units.conversion_factor(pars[n].unit, u) if (pars[n].unit and u) else 1.0 for n, u in cols]
        self.kernel_fac = None if all(f == 1.0 for f in fac) else np.array(fac, dtype=np.float64)
        self.catalog_key = key
        # Direct route for the common case (every kernel column is a FREE parameter, flat bounds): the (W, P) proposal array
        # is checked and handed to the kernel without building the (W, n_all) table first.  The fixed parameters are
        # constants: their bounds are verified here once (the reference re-checks them per call, runner.py:207-214).
        free_pos = {int(i): pos for pos, i in enumerate(self.free_idx)}
        self.direct_cols = None
        if self.simple and all(int(i) in free_pos for i in self.kernel_idx):
            cols_ = np.array([free_pos[int(i)] for i in self.kernel_idx], dtype=np.intp)
            self.direct_cols = cols_
            self.direct_identity = bool(cols_.size == self.free_idx.size and np.array_equal(cols_, np.arange(cols_.size)))
            self.fixed_ok = bool(np.all((self.fixed_val >= self.lo[self.fixed_idx]) & (self.fixed_val <= self.hi[self.fixed_idx]))) \
                if self.fixed_idx.size else True
            free_lo, free_hi = self.lo[self.free_idx], self.hi[self.free_idx]
            self.lo_cols = np.flatnonzero(~np.isneginf(free_lo))
            self.hi_cols = np.flatnonzero(~np.isposinf(free_hi))
            self.lo_vals, self.hi_vals = free_lo[self.lo_cols], free_hi[self.hi_cols]

    @staticmethod
    def signature(runner):
        return tuple((p.fixed, p._value if p.fixed else None, p.min, p.max, p._expr, p._lnprior, p.unit)
                     for p in runner.parameters.values())

    def full(self, values):
        out = np.empty((values.shape[0], self.n_all), dtype=np.float64)
        out[:, self.free_idx] = values
        if self.fixed_idx.size:
            out[:, self.fixed_idx] = self.fixed_val
        return out

    def prior_ok(self, full):
        """Boolean (W,): every parameter inside its inclusive bounds (NaN counts as outside)."""
        if self.unbounded:
            return ~np.isnan(full).any(axis=1)
        return ((full >= self.lo) & (full <= self.hi)).all(axis=1)

    def table(self, full):
        t = full[:, self.kernel_idx]
        return t if self.kernel_fac is None else t * self.kernel_fac

    def prior_ok_free(self, values):
        """`prior_ok` from the (W, P) proposals alone (direct route)."""
        if not self.fixed_ok:
            return np.zeros(values.shape[0], dtype=bool)
        ok = ~np.isnan(values).any(axis=1)
        if self.lo_cols.size:
            ok &= (values[:, self.lo_cols] >= self.lo_vals).all(axis=1)
        if self.hi_cols.size:
            ok &= (values[:, self.hi_cols] <= self.hi_vals).all(axis=1)
        return ok

    def table_direct(self, values):
        t = values if self.direct_identity else values[:, self.direct_cols]
        return t if self.kernel_fac is None else t * self.kernel_fac


class Runner(object):
    """Parent of the analysis classes.  Sub-classes name the observables and model parameters they
    need (``OBSERVABLES``, ``MODEL_PARAMETERS``) and implement ``_lnlike_batch``."""

    MODEL_PARAMETERS = []
    OBSERVABLES = {"v": "km/s", "verr": "km/s"}
    parameters_file = None

    def __init__(self, data, parameters, seed=123, background=None, context=None, precision="f64", **kwargs):
        """
        Parameters
        ----------
        data : DataReader
            The observed data.
        parameters : Parameters
            The model parameters.
        seed : int, optional
            Seed of the global NumPy random number generator (runner.py:59).
        background : Gaussian or SingleStars, optional
            Fixed background population; needs a ``pmember`` column in the data (runner.py:96-103).
        context : _native.Context, optional
            GPU context (devices / rank).  Default: one process-wide context on device 0.
        precision : {'f64', 'f32', 'f32acc64'}
            Arithmetic of the kernels; 'f64' is the parity mode.
        """
        assert not kwargs, "Unknown keyword arguments provided: {0}".format(kwargs)      # runner.py:56

        np.random.seed(seed)                                                              # runner.py:59

        assert isinstance(data, DataReader), "'data' must be instance of {0}".format(DataReader.__module__)
        self.data = data

        if "ra" in self.OBSERVABLES or "dec" in self.OBSERVABLES:
            if not data.has_coordinates:
                raise IOError("Missing WCS coordinates of observed data.")                # runner.py:70-72

        for required, unit in self.OBSERVABLES.items():
            assert required in data.data.columns, "Input data missing required column <{0}>".format(required)
            if unit is not None and data.data.unit(required) is None:
                logger.warning("Missing units for <%s> values. Assuming %s.", required, unit)
            setattr(self, required, data.column(required, unit))

        assert isinstance(parameters, Parameters), "'parameters' must be instance of {0}".format(Parameters.__module__)
        self.parameters = parameters

        missing = set(self.MODEL_PARAMETERS).difference(self.parameters)
        if missing:
            raise IOError("Missing required parameter(s): '{0}'".format(missing))          # runner.py:87-89
        unused = set(self.parameters).difference(self.MODEL_PARAMETERS)
        if unused:
            logger.warning("Superfluous parameter(s) provided: '%s'", unused)

        self.background = background
        if self.background:
            assert isinstance(background, (SingleStars, Gaussian)), \
                "'background' must be an instance of a Background class."
            if "pmember" not in self.data.data.columns:
                logger.error("Inclusion of background population requires prior probabilities for membership.")
            if isinstance(background, SingleStars):        # O(N M) kernel-density precompute: on the device
                lnbg = self.background(self.v, self.verr, context=context)
            else:
                lnbg = self.background(self.v, self.verr)
            self.lnlike_background = np.asarray(lnbg, dtype=np.float64)
            self.pmember = data.column("pmember")
        else:
            self.lnlike_background = None
            self.pmember = None

        self._context = context
        self._precision = precision
        self._catalog = None
        self._catalog_key = None

    # ------------------------------------------------------------------ bookkeeping
    @classmethod
    def default_parameters(cls):
        if cls.parameters_file is None:
            raise NotImplementedError
        return Parameters().load(cls.parameters_file)

    @property
    def n_data(self):
        return self.data.sample_size

    @property
    def fitted_parameters(self):
        return [p for p in self.parameters if not self.parameters[p].fixed]

    @property
    def n_fitted_parameters(self):
        return len(self.fitted_parameters)

    @property
    def units(self):
        return {p: self.parameters[p].unit for p in self.parameters}

    @property
    def labels(self):
        return [par.label for par in self.parameters.values() if not par.fixed]

    @property
    def context(self):
        if self._context is None:
            self._context = _native.default_context()
        return self._context

    # ------------------------------------------------------------------ single-walker API (reference names)
    def fetch_parameter_values(self, values):
        """Dictionary with one value per model parameter, fixed or not (runner.py:143-180).  As in the
        reference, the values are also written back into ``self.parameters``."""
        values = np.asarray(values, dtype=np.float64).reshape(-1)
        resolved = self.parameters.resolve_batch(values[None, :])
        current = OrderedDict((name, float(col[0])) for name, col in resolved.items())
        for name, val in current.items():
            if self.parameters[name].expr is None:
                self.parameters[name].value = val                                          # runner.py:176
        return current

    def lnprior(self, values, parameters_to_ignore=None):
        """0 when every parameter lies inside its inclusive bounds (plus optional ``lnprior``
        expressions), -inf otherwise (runner.py:182-217)."""
        values = np.asarray(values, dtype=np.float64).reshape(-1)
        lnlike = 0
        for name, value in self.fetch_parameter_values(values).items():
            lnlike += self.parameters[name].evaluate_lnprior(value)
            if not np.isfinite(lnlike):
                return -np.inf
        return lnlike

    def lnlike(self, values):
        """Log-likelihood of one parameter vector, without priors (place-holder in the reference,
        runner.py:219-238; sub-classes evaluate it on the GPU)."""
        values = np.asarray(values, dtype=np.float64).reshape(1, -1)
        self.fetch_parameter_values(values[0])
        return float(self.lnlike_batch(values)[0])

    def lnprob(self, values):
        """Log-posterior of one parameter vector (runner.py:288-306): the likelihood is not evaluated
        when the prior is not finite."""
        lp = self.lnprior(values)
        if not np.isfinite(lp):
            return -np.inf
        return self.lnlike(values) + lp

    # ------------------------------------------------------------------ batched API (new)
    def lnprior_batch(self, values):
        return self.parameters.lnprior_batch(self.parameters.resolve_batch(values))

    def lnlike_batch(self, values):
        """(W, P) free-parameter vectors -> (W,) log-likelihoods, one kernel launch."""
        resolved = self.parameters.resolve_batch(values)
        return self._lnlike_batch(resolved)

    def _plan(self):
        sig = _BatchPlan.signature(self)
        if getattr(self, "_plan_sig", None) != sig:
            self._plan_cache = _BatchPlan(self)
            self._plan_sig = sig
        return self._plan_cache

    def lnprob_batch(self, values):
        """(W, P) -> (W,) log-posteriors.  Walkers outside the prior get -inf; their rows are replaced by
        a valid row for the launch and masked afterwards (the reference skips the evaluation)."""
        values = np.atleast_2d(np.asarray(values, dtype=np.float64))
        if self._context is not None and getattr(self._context, "n_ranks", 1) > 1:
            self._check_ranks_agree(values)
        plan = self._plan()
        if plan.direct_cols is not None and values.shape[1] == plan.free_idx.size:
            ok = plan.prior_ok_free(values)
            n_ok = int(np.count_nonzero(ok))
            if n_ok == 0:
                return np.full(values.shape[0], -np.inf)
            if n_ok != ok.size:
                values = values.copy()
                values[~ok] = values[int(np.flatnonzero(ok)[0])]
            cat = self._catalog
            if cat is None or plan.catalog_key != self._catalog_key:
                cat = self._ensure_catalog()
            ll = cat.loglike(plan.table_direct(values))
            if n_ok == ok.size:
                return ll
            out = np.full(values.shape[0], -np.inf)
            out[ok] = ll[ok]
            return out
        if plan.simple and values.shape[1] == plan.free_idx.size:
            # fast host path: flat bounds, no expression priors / constraints
            full = plan.full(values)
            ok = plan.prior_ok(full)
            out = np.full(values.shape[0], -np.inf)
            n_ok = int(ok.sum())
            if n_ok == 0:
                return out
            if n_ok != ok.size:
                full[~ok] = full[int(np.flatnonzero(ok)[0])]
            cat = self._catalog
            if cat is None or plan.catalog_key != self._catalog_key:
                cat = self._ensure_catalog()
            ll = cat.loglike(plan.table(full))
            if n_ok == ok.size:
                return ll
            out[ok] = ll[ok]
            return out
        resolved = self.parameters.resolve_batch(values)
        lp = self.parameters.lnprior_batch(resolved)
        ok = np.isfinite(lp)
        out = np.full(values.shape[0], -np.inf)
        if not ok.any():
            return out
        if not ok.all():
            donor = int(np.flatnonzero(ok)[0])
            resolved = OrderedDict((k, np.where(ok, col, col[donor])) for k, col in resolved.items())
        ll = self._lnlike_batch(resolved)
        out[ok] = ll[ok] + lp[ok]
        return out

    # ------------------------------------------------------------------ several ranks (one process per GPU)
    RANK_CHECK_EVERY = 256

    def _rank_group(self):
        """Host group of a multi-rank context (``distributed.rank_context``), None for a single process."""
        ctx = self._context
        if ctx is None or getattr(ctx, "n_ranks", 1) <= 1:
            return None
        group = getattr(ctx, "host_group", None)
        if group is None:
            raise RuntimeError("this Runner sits on a multi-rank context without a host group: create the context with "
                               "mcmc_dynamics_amd.distributed.rank_context(), which keeps the group for the start-up hand-offs")
        return group

    def _check_ranks_agree(self, values):
        """The all-reduce at the end of every evaluation adds the ranks' partial sums row by row: it is only meaningful
        when every rank passes the SAME (W, P) table.  Verified on the first call and every RANK_CHECK_EVERY-th call
        (a CRC over the host group); a mismatch raises instead of producing a silently wrong chain."""
        n = getattr(self, "_n_rank_batches", 0)
        self._n_rank_batches = n + 1
        if n % self.RANK_CHECK_EVERY:
            return
        if not self._rank_group().same_everywhere(values):
            raise RuntimeError("the ranks of this job evaluate different walker tables: their partial log-likelihoods must not "
                               "be summed (seed every rank's sampler identically, e.g. through Runner.__call__)")

    # ------------------------------------------------------------------ sampler driver
    def get_initials(self, n_walkers):
        """Initial positions from each free parameter's ``initials`` recipe (runner.py:308-330)."""
        initials = np.zeros((n_walkers, self.n_fitted_parameters))
        i = 0
        for parameter in self.parameters.values():
            if parameter.fixed:
                continue
            initials[:, i] = parameter.evaluate_initials(n_walkers)
            i += 1
        return initials

    # Which sampler drives ``__call__``.  The reference hands ``Runner.lnprob`` to ``emcee.EnsembleSampler`` and lets emcee
    # drive the outer loop (runner.py:403, 416-419); so does this class whenever emcee can be imported:
    #   "auto" (default)  emcee (gets ``lnprob_batch`` with ``vectorize=True``) if importable, otherwise the built-in sampler
    #                     (``mcmc_dynamics_amd.sampler``: emcee's default move and attributes), with whole blocks of steps
    #                     inside the library where that is possible (see "resident");
    #   "emcee"           the real ``emcee.EnsembleSampler`` or ImportError;
    #   "resident"        the built-in sampler with the ensemble resident on the device (``mcd_stretch_move``: 87 - 95 % of
    #                     the kernel rate against ~60 % through one Python call per half step).  Needs box priors and the
    #                     package's own posterior methods (``resident_ok``); ValueError otherwise;
    #   "builtin"         never emcee: resident blocks where possible, else the built-in Python loop around ``lnprob_batch``.
    # One INFO line per run names the driver.
    SAMPLER = "auto"

    # methods that define the posterior: a sub-class that overrides one of them OUTSIDE this package (the reference's
    # Runner is meant to be sub-classed: its ``lnlike`` is a placeholder, runner.py:219-238) changes what emcee would
    # sample -- the library's block entry evaluates the built-in model and would silently ignore it
    _POSTERIOR_METHODS = ("lnprob", "lnprob_batch", "lnlike", "lnlike_batch", "_lnlike_batch", "lnprior", "lnprior_batch",
                          "fetch_parameter_values")

    def resident_ok(self):
        """(bool, reason): may whole blocks of steps run inside the library (``mcd_stretch_move``)?"""
        if not self.NATIVE_STRETCH:
            return False, "this class evaluates more than one un-binned catalogue per call (NATIVE_STRETCH is off)"
        for name in self._POSTERIOR_METHODS:
            fn = getattr(type(self), name, None)
            module = getattr(fn, "__module__", None) or ""
            if fn is not None and not module.startswith(__name__.rsplit(".", 2)[0] + "."):
                return False, "{0}.{1} overrides the posterior outside the package".format(type(self).__name__, name)
        if not self._plan().simple:
            return False, "the priors are not plain boxes (expression priors / constrained parameters)"
        return True, ""

    def _make_sampler(self, n_walkers, seed=None):
        if self.SAMPLER not in ("auto", "emcee", "resident", "builtin"):
            raise ValueError("Runner.SAMPLER must be 'auto', 'emcee', 'resident' or 'builtin'")
        resident, why_not = self.resident_ok()
        if self.SAMPLER == "resident" and not resident:
            raise ValueError("Runner.SAMPLER = 'resident' is not possible here: " + why_not)
        if self.SAMPLER in ("auto", "emcee"):
            try:
                import emcee
                sampler = emcee.EnsembleSampler(n_walkers, self.n_fitted_parameters, self.lnprob_batch, vectorize=True)
                if seed is not None:
                    sampler._random.seed(seed)
                logger.info("MCMC driver: emcee.EnsembleSampler around lnprob_batch (vectorize=True), as runner.py:403")
                return sampler
            except ImportError:
                if self.SAMPLER == "emcee":
                    raise
        from ..sampler import EnsembleSampler
        logger.info("MCMC driver: built-in stretch move (emcee's default move), %s",
                    "blocks of steps inside the library, ensemble resident on the device" if resident
                    else "Python loop around lnprob_batch (" + why_not + ")")
        if self.RNG not in ("host", "device"):
            raise ValueError("Runner.RNG must be 'host' or 'device'")
        return EnsembleSampler(n_walkers, self.n_fitted_parameters, self.lnprob_batch, vectorize=True, seed=seed,
                               block_fn=self._stretch_block if resident else None, rng=self.RNG,
                               seeded_block_fn=self._stretch_block_seeded if resident else None)

    # the built-in move's random numbers: "device" (default) -- the counter-based generator of csrc/mcd_rng.h, generated on
    # the device for resident blocks and by the library's host code for the Python loop (a function of (seed, step, walker):
    # no numbers cross PCIe, the chain does not depend on how it is run or cut into blocks); "host" -- NumPy's Mersenne
    # twister, drawn on the host as emcee does (the draws of 256 walkers then bound small catalogues: 26 000 steps/s at 1e5
    # stars against 30 000).  emcee itself, when it drives (SAMPLER), draws as it always does.
    RNG = "device"
    NATIVE_STRETCH = True          # sub-classes whose posterior is not ONE un-binned catalogue switch this off

    def _stretch_plan(self):
        """Arguments of ``mcd_stretch_move`` for the current parameter configuration (box priors only: ``plan.simple``):
        which free parameter feeds each kernel column, constants for fixed parameters, unit factors, bounds."""
        plan = self._plan()
        cached = getattr(self, "_stretch_cache", None)
        if cached is not None and cached[0] is plan:
            return cached[1]
        free_pos = {int(i): j for j, i in enumerate(plan.free_idx)}
        fixed_val = {int(i): float(v) for i, v in zip(plan.fixed_idx, plan.fixed_val)}
        k = len(plan.kernel_idx)
        fac = np.ones(k) if plan.kernel_fac is None else np.asarray(plan.kernel_fac, dtype=np.float64)
        src, const = np.full(k, -1, dtype=np.int32), np.zeros(k)
        for c, i in enumerate(plan.kernel_idx):
            if int(i) in free_pos:
                src[c] = free_pos[int(i)]
            else:
                const[c] = fixed_val[int(i)] * fac[c] if plan.kernel_fac is not None else fixed_val[int(i)]
        fixed_ok = bool(np.all((plan.fixed_val >= plan.lo[plan.fixed_idx]) & (plan.fixed_val <= plan.hi[plan.fixed_idx]))) \
            if plan.fixed_idx.size else True
        out = {"col_source": src, "col_const": const, "col_factor": fac, "lo": plan.lo[plan.free_idx].copy(),
               "hi": plan.hi[plan.free_idx].copy(), "fixed_ok": fixed_ok}
        self._stretch_cache = (plan, out)
        return out

    def _stretch_block_seeded(self, pos, lnp, seed, step0, n_steps, chain, lnprob_chain, accepted):
        """One block of steps with the random numbers generated inside the library (``_native.Catalog.stretch_move_seeded``)."""
        plan = self._plan()
        if not plan.simple:
            raise RuntimeError("the parameter configuration changed to one with expression priors / constraints during a run")
        if self._context is not None and getattr(self._context, "n_ranks", 1) > 1:
            self._check_ranks_agree(pos)
        cat = self._catalog
        if cat is None or plan.catalog_key != self._catalog_key:
            cat = self._ensure_catalog()
        cat.stretch_move_seeded(self._stretch_plan(), pos, lnp, seed, step0, n_steps, chain, lnprob_chain, accepted)

    def _stretch_block(self, pos, lnp, order, zz, thr, pick, chain, lnprob_chain, accepted):
        """One block of stretch-move steps inside the library (``_native.Catalog.stretch_move``)."""
        plan = self._plan()
        if not plan.simple:
            raise RuntimeError("the parameter configuration changed to one with expression priors / constraints during a run")
        if self._context is not None and getattr(self._context, "n_ranks", 1) > 1:
            self._check_ranks_agree(pos)
        cat = self._catalog
        if cat is None or plan.catalog_key != self._catalog_key:
            cat = self._ensure_catalog()
        cat.stretch_move(self._stretch_plan(), pos, lnp, order, zz, thr, pick, chain, lnprob_chain, accepted)

    def __call__(self, n_walkers=100, n_steps=500, n_burn=100, n_threads=1, n_out=None, pos=None, lnprob0=None,
                 plot=False, prefix="sampler", true_values=None, **kwargs):
        """Run the MCMC (runner.py:332-443).  Same arguments as the reference; ``n_threads`` must be 1
        because the likelihood of all walkers is one GPU launch (a process pool would fork after HIP
        initialisation and hold W copies of the catalogue)."""
        if kwargs:
            if "filename" in kwargs or "plotfilename" in kwargs:
                logger.warning("Parameters <filename> and <plotfilename> not used anymore. Use <prefix> instead.")
        if n_threads != 1:
            raise ValueError("n_threads > 1 is not supported by the GPU backend: walkers are batched on the device.")

        fig = None
        if plot:
            import matplotlib.pyplot as plt
            fig, _ = plt.subplots(self.n_fitted_parameters, 1, sharex="all", figsize=(8, 9))

        if pos is not None:
            pos = np.asarray(pos, dtype=np.float64)
            assert pos.shape == (n_walkers, self.n_fitted_parameters), "Array with starting values has invalid shape."
        else:
            pos = self.get_initials(n_walkers=n_walkers)

        # Several ranks (stars sharded, one process per GPU): rank 0's start positions and one sampler seed go to every
        # rank, so that all of them propose the same walkers at every step (the reference has one process, runner.py:403).
        group, seed = self._rank_group(), None
        if group is not None:
            pos = group.bcast_array(pos, src=0)
            seed = int(group.bcast_json(int(np.random.SeedSequence().entropy % (2 ** 32)), src=0))

        lp0 = self.lnprior_batch(pos)
        for i in range(n_walkers):
            if not np.isfinite(lp0[i]):
                raise ValueError("Invalid initial guesses for walker {0}: {1}={2}".format(
                    i, self.fitted_parameters, pos[i]))                                      # runner.py:392-395

        sampler = self._make_sampler(n_walkers, seed=seed)
        logger.info("Running MCMC chain ...")
        if n_out is not None:
            logger.info("Iter. <log like>   " + "".join(" {0:12s}".format("<" + n + ">") for n in self.fitted_parameters))

        state = None
        while sampler.iteration < n_steps:
            # runner.py:418-419.  One deliberate difference: the reference re-passes the caller's
            # `lnprob0` on every chunk although the positions have moved; it is used for the first chunk only.
            try:
                result = sampler.run_mcmc(pos, n_out if n_out is not None else n_steps, log_prob0=lnprob0,
                                          rstate0=state, progress=False)
            except BaseException as exc:
                # several ranks: the peers are inside (or about to enter) an all-reduce this rank will not join any more.
                # Tell them before the exception travels on (hostgroup.abort -> mcd_ctx_abort on their side): they leave
                # their wait with an error at once instead of at the library's collective deadline.  Every rank ends with
                # an exception; nothing is retried or re-routed inside the process.
                if group is not None:
                    group.abort("{0}: {1}".format(type(exc).__name__, exc))
                raise
            pos, lnp, state = tuple(result)[:3]
            lnprob0 = None
            if n_out is not None:
                output = " {0:4d} {1:12.5e}".format(sampler.iteration, np.mean(lnp[:]))
                output += "".join(" {0:12.5e}".format(np.mean(pos[:, i])) for i in range(self.n_fitted_parameters))
                if sampler.iteration % n_out == 0:
                    if prefix is not None:
                        self.save_current_status(sampler, prefix=prefix)
                    if plot:
                        for ax in fig.axes:
                            ax.cla()
                        self.plot_chain(sampler.chain, true_values=true_values, figure=fig,
                                        filename="{0}_chains.png".format(prefix) if prefix is not None else None)
                logger.info(output)
        return sampler

    # ------------------------------------------------------------------ checkpoints (runner.py:445-519)
    @staticmethod
    def save_chain(sampler, filename="samplerchain.pkl"):
        warnings.warn("Method Runner.save_chain() is deprecated. Use Runner.save_current_status() instead.",
                      DeprecationWarning)
        prefix = filename.split(".")[0]
        if len(prefix) > 5 and prefix[-5:] == "chain":
            prefix = prefix[:-5]
        Runner.save_current_status(sampler, prefix=prefix)

    @staticmethod
    def save_current_status(sampler, prefix="sampler"):
        """Pickle ``sampler.chain`` (W, steps, P) and ``sampler.lnprobability`` (W, steps) to
        ``{prefix}_chain.pkl`` / ``{prefix}_lnprob.pkl`` -- the reference's on-disk format."""
        with open("{0}_chain.pkl".format(prefix), "wb") as f:
            pickle.dump(np.asarray(sampler.chain), f)
        with open("{0}_lnprob.pkl".format(prefix), "wb") as f:
            pickle.dump(np.asarray(sampler.lnprobability), f)

    @staticmethod
    def read_chain(filename="samplerchain.pkl"):
        with open(filename, "rb") as f:
            return pickle.load(f)

    @staticmethod
    def read_final_chain(filename="restart.plk"):
        with open(filename, "rb") as f:
            chain = pickle.load(f)
        return chain[:, -1, :]

    # ------------------------------------------------------------------ chain statistics (runner.py:521-660)
    def convert_to_parameters(self, chain, n_burn):
        """Chain (W, steps, P) -> dict name -> flat samples, including fixed and constrained parameters."""
        chain = np.asarray(chain)
        flat = chain[:, n_burn:, :].reshape(-1, chain.shape[2])
        resolved = self.parameters.resolve_batch(flat)
        return {name: np.array(col) for name, col in resolved.items()}

    def compute_percentiles(self, chain, n_burn, pct=None):
        if pct is None:
            pct = [16, 50, 84]
        samples = np.asarray(chain)[:, n_burn:, :].reshape((-1, self.n_fitted_parameters))
        return np.percentile(samples, pct, axis=0)

    def compute_bestfit_values(self, chain, n_burn):
        """Median and upper / lower 1-sigma uncertainties per fitted parameter; the medians are also
        written into ``self.parameters`` as in the reference (runner.py:649)."""
        percentiles = self.compute_percentiles(chain, n_burn=n_burn, pct=[16, 50, 84])
        results = ResultsTable()
        i = 0
        for name, parameter in self.parameters.items():
            if parameter.fixed:
                continue
            parameter.value = percentiles[1, i]
            results.add_column(name, percentiles[1, i], percentiles[2, i] - percentiles[1, i],
                               percentiles[1, i] - percentiles[0, i], unit=parameter.unit)
            i += 1
        return results

    def plot_chain(self, chain, filename="chains.png", true_values=None, figure=None, lnprob=None, plot_median=False):
        """Trace plot of every fitted parameter (simplified form of runner.py:675-770)."""
        import matplotlib.pyplot as plt
        chain = np.asarray(chain)
        if figure is None:
            figure, _ = plt.subplots(self.n_fitted_parameters, 1, sharex="all", figsize=(8, 9))
        for i, ax in enumerate(figure.axes[:self.n_fitted_parameters]):
            ax.plot(chain[:, :, i].T, color="k", alpha=0.3, lw=0.6)
            if plot_median:
                ax.plot(np.median(chain[:, :, i], axis=0), color="C3", lw=1.5)
            if true_values is not None:
                ax.axhline(true_values[i], color="C0", lw=1.5)
            ax.set_ylabel(self.labels[i])
        if filename is not None:
            figure.savefig(filename)
        return figure

    def sample_chain(self, chain, n_burn, n_samples=1):
        """``n_samples`` random parameter sets from the post-burn-in chain, each as the dictionary
        ``fetch_parameter_values`` returns (runner.py:820-850)."""
        flat = np.reshape(np.asarray(chain)[:, n_burn:], (-1, np.shape(chain)[-1]))
        indices = np.random.randint(0, flat.shape[0], (n_samples,))
        return [self.fetch_parameter_values(row) for row in flat[indices]]

    # ------------------------------------------------------------------ device catalogue
    # Sub-classes set `_model_id` and the ordered (name, canonical unit) columns of the kernel's parameter
    # table before (`_KERNEL_HEAD`) and after (`_KERNEL_TAIL`) the optional centre columns (include/mcd.h).
    _model_id = None
    _KERNEL_HEAD = ()
    _KERNEL_TAIL = ()

    def _canonical(self, resolved, name, unit):
        """Resolved column of parameter ``name`` expressed in the kernel's canonical unit."""
        f = units.conversion_factor(self.parameters[name].unit, unit) if self.parameters[name].unit else 1.0
        return resolved[name] if f == 1.0 else resolved[name] * f

    def _centre_is_fixed(self):
        pr, pd = self.parameters["ra_center"], self.parameters["dec_center"]
        return pr.fixed and pd.fixed and pr.expr is None and pd.expr is None

    def _catalog_model(self):
        """(model id, extra per-star columns) of the device catalogue."""
        return self._model_id, {}

    def _catalog_spec(self):
        """(key, constructor kwargs) of the device catalogue for the current parameter configuration:
        a fixed centre lets the walker-independent geometry be precomputed at upload."""
        if self._centre_is_fixed():
            centre = (float(units.to_unit(self.parameters["ra_center"].value, "deg", self.parameters["ra_center"].unit)),
                      float(units.to_unit(self.parameters["dec_center"].value, "deg", self.parameters["dec_center"].unit)))
        else:
            centre = None
        model, extra = self._catalog_model()
        return (model, centre), dict(model=model, centre=centre, **extra)

    def _catalog_kwargs(self):
        return {}

    def _ensure_catalog(self):
        plan = getattr(self, "_plan_cache", None)
        if (self._catalog is not None and plan is not None and self._plan_sig == _BatchPlan.signature(self)
                and plan.catalog_key == self._catalog_key):
            return self._catalog                          # nothing about the parameters changed since the plan was built
        key, spec = self._catalog_spec()
        if self._catalog is None or key != self._catalog_key:
            if self._catalog is not None:
                self._catalog.close()
            self._catalog = _native.Catalog(self.context, self.ra, self.dec, self.v, self.verr,
                                            precision=self._precision, **self._catalog_kwargs(), **spec)
            self._catalog_key = key
        return self._catalog

    def _kernel_table(self, resolved):
        """(W, K) float64 table in the column order and canonical units the C-ABI expects."""
        cols = [self._canonical(resolved, n, u) for n, u in self._KERNEL_HEAD]
        if self._catalog_key[1] is None:
            cols += [self._canonical(resolved, "ra_center", "deg"), self._canonical(resolved, "dec_center", "deg")]
        cols += [self._canonical(resolved, n, u) for n, u in self._KERNEL_TAIL]
        return np.stack(cols, axis=1)

    def _lnlike_batch(self, resolved):
        cat = self._ensure_catalog()
        return cat.loglike(self._kernel_table(resolved))

    def _per_star(self, values, what):
        """Per-star device outputs for ONE parameter vector: 'membership' or 'lnlike'."""
        resolved = self.parameters.resolve_batch(np.asarray(values, dtype=np.float64).reshape(1, -1))
        cat = self._ensure_catalog()
        row = self._kernel_table(resolved)[0]
        return cat.membership(row) if what == "membership" else cat.loglike_per_star(row)

    def close(self):
        if self._catalog is not None:
            self._catalog.close()
            self._catalog = None
            self._catalog_key = None
