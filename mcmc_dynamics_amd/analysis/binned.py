"""Radial-binned dispersion / rotation profile: B independent ``ConstantFit`` posteriors evaluated in ONE
kernel launch per proposal batch.

The reference has no multi-bin class: its "profile" is a Python loop that builds one ``ConstantFit`` per
radial bin and runs B separate MCMCs one after the other (bin/run_tests.py:75-124, bin/run.py:146-259;
bins from ``DataReader.make_radial_bins``, utils/files/data_reader.py:71-140).  Here the stars are sorted
by bin once, the catalogue carries ``bin_offsets``, every bin owns its own walker ensemble, and the B
ensembles advance in lockstep: a stretch-move half-step proposes ``(B, W/2)`` positions and one segmented
launch returns the ``(B, W/2)`` log-likelihoods.
"""
import logging
from collections import OrderedDict

import numpy as np

from .. import _native
from .constant import ConstantFit

logger = logging.getLogger(__name__)


class BinnedSampler(object):
    """B lock-stepped affine-invariant ensembles (same move as ``sampler.EnsembleSampler``)."""

    def __init__(self, n_bins, nwalkers, ndim, log_prob_fn, a=2.0, seed=None):
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError("need an even number of walkers, at least twice the dimension")
        self.n_bins, self.nwalkers, self.ndim = int(n_bins), int(nwalkers), int(ndim)
        self.log_prob_fn = log_prob_fn
        self.a = float(a)
        self._random = np.random.RandomState(seed)
        self.iteration = 0
        self._chain = []
        self._lnprob = []
        self._accepted = np.zeros((self.n_bins, self.nwalkers))

    @property
    def chain(self):
        """(B, W, steps, P): per bin the (W, steps, P) layout of the reference's pickles."""
        return np.transpose(np.array(self._chain), (1, 2, 0, 3)) if self._chain else \
            np.empty((self.n_bins, self.nwalkers, 0, self.ndim))

    @property
    def lnprobability(self):
        return np.transpose(np.array(self._lnprob), (1, 2, 0)) if self._lnprob else \
            np.empty((self.n_bins, self.nwalkers, 0))

    @property
    def acceptance_fraction(self):
        return self._accepted / max(1, self.iteration)

    def run_mcmc(self, pos, nsteps, log_prob0=None):
        pos = np.array(pos, dtype=np.float64)
        B, W, P = self.n_bins, self.nwalkers, self.ndim
        if pos.shape != (B, W, P):
            raise ValueError("incompatible input dimensions {0}".format(pos.shape))
        lnp = np.array(self.log_prob_fn(pos)) if log_prob0 is None else np.array(log_prob0, dtype=np.float64)
        if np.any(np.isnan(lnp)):
            raise ValueError("Probability function returned NaN")
        half = W // 2
        rnd = self._random
        rows = np.arange(B)[:, None]
        for _ in range(int(nsteps)):
            order = np.argsort(rnd.rand(B, W), axis=1)
            for first, second in ((order[:, :half], order[:, half:]), (order[:, half:], order[:, :half])):
                s = pos[rows, first]
                zz = ((self.a - 1.0) * rnd.rand(B, half) + 1.0) ** 2.0 / self.a
                pick = rnd.randint(half, size=(B, half))
                partners = pos[rows, np.take_along_axis(second, pick, axis=1)]
                proposal = partners - (partners - s) * zz[:, :, None]
                new_lnp = np.asarray(self.log_prob_fn(proposal), dtype=np.float64)
                if np.any(np.isnan(new_lnp)):
                    raise ValueError("Probability function returned NaN")
                old = lnp[rows, first]
                accept = np.log(rnd.rand(B, half)) < (P - 1.0) * np.log(zz) + new_lnp - old
                bb, jj = np.nonzero(accept)
                ww = first[bb, jj]
                pos[bb, ww] = proposal[bb, jj]
                lnp[bb, ww] = new_lnp[bb, jj]
                self._accepted[bb, ww] += 1
            self._chain.append(pos.copy())
            self._lnprob.append(lnp.copy())
            self.iteration += 1
        return pos, lnp, rnd.get_state()


class BinnedConstantFit(ConstantFit):
    """``ConstantFit`` in every radial bin of ``data`` (column ``bin``) simultaneously.

    ``lnprob_batch`` maps ``(B, W, P)`` positions to ``(B, W)`` log-posteriors; ``lnlike_total`` sums the
    bins for one shared parameter set (== the un-binned ``ConstantFit`` value)."""

    def __init__(self, data, parameters=None, **kwargs):
        if "bin" not in data.data.columns:
            raise IOError("BinnedConstantFit needs a 'bin' column: call DataReader.make_radial_bins() first.")
        data_sorted, self.bin_offsets = data.sorted_by_bin()
        self.n_bins = len(self.bin_offsets) - 1
        super(BinnedConstantFit, self).__init__(data=data_sorted, parameters=parameters, **kwargs)
        if self.background is not None:
            raise IOError("BinnedConstantFit does not take a background population.")

    def _catalog_kwargs(self):
        return {"bin_offsets": self.bin_offsets}

    def _resolve(self, values):
        values = np.asarray(values, dtype=np.float64)
        if values.ndim == 2:                       # one parameter set shared by all bins
            values = np.broadcast_to(values, (self.n_bins,) + values.shape)
        if values.ndim != 3 or values.shape[0] != self.n_bins:
            raise ValueError("expected positions of shape (n_bins, W, P)")
        flat = self.parameters.resolve_batch(values.reshape(-1, values.shape[2]))
        return values.shape[1], flat

    def _launch(self, w, flat):
        cat = self._ensure_catalog()
        table = self._kernel_table(flat)
        return cat.loglike(table.reshape(self.n_bins, w, -1)) if self.n_bins > 1 else \
            cat.loglike(table.reshape(w, -1))[None, :]

    def lnlike_batch(self, values):
        w, flat = self._resolve(values)
        return self._launch(w, flat)

    def lnprob_batch(self, values):
        w, flat = self._resolve(values)
        lp = self.parameters.lnprior_batch(flat)
        ok = np.isfinite(lp)
        out = np.full(self.n_bins * w, -np.inf)
        if ok.any():
            if not ok.all():
                donor = int(np.flatnonzero(ok)[0])
                flat = OrderedDict((k, np.where(ok, col, col[donor])) for k, col in flat.items())
            ll = self._launch(w, flat).reshape(-1)
            out[ok] = ll[ok] + lp[ok]
        return out.reshape(self.n_bins, w)

    def lnlike_total(self, values):
        """(W, P) -> (W,): sum over bins with one shared parameter set."""
        return self.lnlike_batch(np.atleast_2d(values)).sum(axis=0)

    def lnlike(self, values):
        return float(self.lnlike_total(np.asarray(values, dtype=np.float64).reshape(1, -1))[0])

    def __call__(self, n_walkers=100, n_steps=100, pos=None, seed=None, **kwargs):
        """Run the B ensembles in lockstep; returns a ``BinnedSampler`` (``.chain``: (B, W, steps, P))."""
        if kwargs.get("n_threads", 1) != 1:
            raise ValueError("n_threads > 1 is not supported by the GPU backend.")
        if pos is None:
            pos = np.stack([self.get_initials(n_walkers) for _ in range(self.n_bins)])
        pos = np.asarray(pos, dtype=np.float64)
        lp = self.parameters.lnprior_batch(self.parameters.resolve_batch(pos.reshape(-1, pos.shape[-1])))
        if not np.all(np.isfinite(lp)):
            raise ValueError("Invalid initial guesses for {0} walker(s).".format(int(np.sum(~np.isfinite(lp)))))
        sampler = BinnedSampler(self.n_bins, n_walkers, self.n_fitted_parameters, self.lnprob_batch, seed=seed)
        sampler.run_mcmc(pos, n_steps)
        return sampler

    def compute_bestfit_values(self, chain, n_burn):
        """List of per-bin result tables (median / uperr / loerr), as the per-bin loop of
        bin/run_tests.py:105-113 collects them."""
        saved = {k: p.value for k, p in self.parameters.items()}
        out = [super(BinnedConstantFit, self).compute_bestfit_values(chain[b], n_burn) for b in range(self.n_bins)]
        for k, v in saved.items():
            if self.parameters[k].expr is None:
                self.parameters[k].value = v
        return out
