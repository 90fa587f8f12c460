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
    """B lock-stepped affine-invariant ensembles (same move as ``sampler.EnsembleSampler``), one per radial bin: the
    reference runs one MCMC per bin (bin/run_tests.py:75-124); here the bins share every evaluation.

    ``block_fn`` (what ``BinnedConstantFit`` passes for box priors): a callable that advances all ensembles by a whole block
    of steps from the random numbers drawn here -- ``mcd_stretch_move`` with ``n_bins = B`` runs the same half-step loop in
    C++ (csrc/mcd_stretch.h), bit-identical to the Python loop below.  The Python loop costs ~7 ms of NumPy per step at
    55 bins x 512 walkers against 0.17 ms of device time per step; the library's loop ~0.2 ms."""

    N_STREAMS = 8          # part of the definition of the random stream (2: 334, 4: 313, 8: 297 us per step at 55 x 512)

    def __init__(self, n_bins, nwalkers, ndim, log_prob_fn, a=2.0, seed=None, block_fn=None, rng="host", seeded_block_fn=None):
        """``rng``: where the move's random numbers come from.

        ``"host"``: NumPy's Mersenne twister on the host, as emcee draws (B x W numbers per step cross PCIe: the draws, not
        the device, bound the rate -- 3400 steps/s at 55 bins x 512 walkers against 6000 of device time).

        ``"device"``: the counter-based generator of csrc/mcd_rng.h (Philox4x64-10): every number is a function of (seed,
        step, half step, bin, walker) alone.  ``seeded_block_fn`` (``Runner._stretch_block_seeded``) runs whole blocks with
        the numbers generated on the device (csrc/mcd_stretch.hip: chain_numbers_kernel); without it (expression priors) the NumPy loop below takes the SAME
        numbers from ``_native.chain_numbers`` -- one chain whichever way it is run, and however it is cut into blocks."""
        if nwalkers % 2 or nwalkers < 2 * ndim:
            raise ValueError("need an even number of walkers, at least twice the dimension")
        if rng not in ("host", "device"):
            raise ValueError("rng must be 'host' or 'device'")
        if rng == "device" and float(a) != 2.0:
            raise ValueError("rng='device' implements the stretch move with a = 2 (emcee's default)")
        self.rng = rng
        self.seeded_block_fn = seeded_block_fn
        self.n_bins, self.nwalkers, self.ndim = int(n_bins), int(nwalkers), int(ndim)
        self.log_prob_fn = log_prob_fn
        self.block_fn = block_fn
        self.block_steps = 64                     # steps whose random numbers are drawn together (defines the stream)
        self.a = float(a)
        self._random = np.random.RandomState(seed)
        # rng="device": the 64-bit name of the chain (no seed: from NumPy's global generator, which the reference seeds at
        # analysis/runner.py:59 -- `np.random.seed` before the run makes it reproducible, as with emcee)
        self.seed64 = None
        if rng == "device":                       # (only then: drawing a seed moves NumPy's global generator)
            self.seed64 = int(seed) & 0xFFFFFFFFFFFFFFFF if seed is not None else \
                (int(np.random.randint(0, 2 ** 32)) << 32) | int(np.random.randint(0, 2 ** 32))
        self.device_block_steps = 256             # rng="device": steps per library call (NOT part of the stream's definition)
        # the steps of a block are drawn by N_STREAMS generators seeded from the master (see run_mcmc: draw)
        self._streams = [np.random.RandomState(int(s)) for s in self._random.randint(0, 2 ** 31 - 1, size=self.N_STREAMS)]
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=self.N_STREAMS)
        self._lookahead = ThreadPoolExecutor(max_workers=1)
        self.iteration = 0
        self._chain = np.empty((0, self.n_bins, self.nwalkers, self.ndim))
        self._lnprob = np.empty((0, self.n_bins, self.nwalkers))
        self._accepted = np.zeros((self.n_bins, self.nwalkers))

    def close(self):
        """Release the drawing threads (also done when the sampler is garbage-collected)."""
        for pool in (getattr(self, "_pool", None), getattr(self, "_lookahead", None)):
            if pool is not None:
                pool.shutdown(wait=False)
        self._pool = self._lookahead = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def chain(self):
        """(B, W, steps, P): per bin the (W, steps, P) layout of the reference's pickles."""
        return np.transpose(self._chain[:self.iteration], (1, 2, 0, 3))

    @property
    def lnprobability(self):
        return np.transpose(self._lnprob[:self.iteration], (1, 2, 0))

    @property
    def acceptance_fraction(self):
        return self._accepted / max(1, self.iteration)

    def reserve(self, total_steps):
        """Chain storage for ``total_steps`` steps in all (a run continued by further ``run_mcmc`` calls otherwise grows its
        storage geometrically and copies the rows it holds: 1.1 MB per step at 55 bins x 512 walkers)."""
        B, W, P = self.n_bins, self.nwalkers, self.ndim
        if int(total_steps) > self._chain.shape[0]:
            chain, lnprob = np.empty((int(total_steps), B, W, P)), np.empty((int(total_steps), B, W))
            chain[:self.iteration], lnprob[:self.iteration] = self._chain[:self.iteration], self._lnprob[:self.iteration]
            self._chain, self._lnprob = chain, lnprob

    def run_mcmc(self, pos, nsteps, log_prob0=None):
        pos = np.ascontiguousarray(pos, dtype=np.float64).copy()
        B, W, P = self.n_bins, self.nwalkers, self.ndim
        if pos.shape != (B, W, P):
            raise ValueError("incompatible input dimensions {0}".format(pos.shape))
        lnp = np.array(self.log_prob_fn(pos), dtype=np.float64) if log_prob0 is None else np.array(log_prob0, dtype=np.float64)
        if np.any(np.isnan(lnp)):
            raise ValueError("Probability function returned NaN")
        nsteps = int(nsteps)
        need = self.iteration + nsteps
        if need > self._chain.shape[0]:
            self.reserve(max(need, 2 * self._chain.shape[0]))
        half = W // 2
        rows = np.arange(B)[:, None]
        am1, inv_a, dm1 = self.a - 1.0, 1.0 / self.a, P - 1.0

        def draw_slice(rnd, out, lo, hi):
            # steps lo..hi of a block in a handful of vectorised draws (sampler.EnsembleSampler draws the same way): split
            # of every ensemble = argsort of uniform keys, stretch factors z ~ g(z), log acceptance thresholds (accept iff
            # thr < new_lnp - old_lnp), partner indices; written straight into the block's arrays
            n = hi - lo
            order, zz, thr, pick = out
            order[lo:hi] = np.argsort(rnd.rand(n, B, W), axis=2)
            u = rnd.rand(n, 4, B, half)
            z = zz[lo:hi]
            np.multiply(u[:, :2], am1, out=z)
            z += 1.0
            z *= z
            z *= inv_a
            np.subtract(np.log(u[:, 2:]), dm1 * np.log(z), out=thr[lo:hi])
            pick[lo:hi] = rnd.randint(half, size=(n, 2, B, half))

        if self._pool is None:                               # (closed: the pools come back on the next run)
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=self.N_STREAMS)
            self._lookahead = ThreadPoolExecutor(max_workers=1)

        def draw(n):
            # B x W numbers per step make the draws the bottleneck (1.3 ms per step at 55 x 512 on one core, against
            # 0.2 ms of device time): the block is cut into len(streams) slices of steps, every slice drawn from its own
            # generator on its own thread (NumPy releases the interpreter lock).  Which numbers a step gets depends only
            # on (seed, block_steps, the step's place in the block), never on thread timing.
            out = (np.empty((n, B, W), dtype=np.int32), np.empty((n, 2, B, half)), np.empty((n, 2, B, half)),
                   np.empty((n, 2, B, half), dtype=np.int32))
            k = len(self._streams)
            cuts = [n * i // k for i in range(k + 1)]
            jobs = [(self._streams[i], out, cuts[i], cuts[i + 1]) for i in range(k) if cuts[i + 1] > cuts[i]]
            if len(jobs) > 1:
                list(self._pool.map(lambda job: draw_slice(*job), jobs))
            else:
                draw_slice(*jobs[0])
            return out

        done = 0
        # (blocks of very many ensembles are kept below ~8 M walker-steps: 200 MB of numbers in, 340 MB of rows out)
        chunk = max(1, min(int(self.block_steps), (1 << 23) // (B * W)))
        device_rng = self.rng == "device"
        if device_rng:
            chunk = max(1, min(int(self.device_block_steps), (1 << 23) // (B * W)))
            if self.seeded_block_fn is None:
                from .. import _native as native

                def draw(n):                                 # noqa: F811 -- the same numbers the device generates
                    return native.chain_numbers(self.seed64, self.iteration, n, B, W, P)
        pending = None
        while done < nsteps:
            n = min(chunk, nsteps - done)
            if device_rng and self.seeded_block_fn is not None:
                it = self.iteration
                accepted = np.zeros((B, W), dtype=np.int64)
                self.seeded_block_fn(pos, lnp, self.seed64, it, n, self._chain[it:it + n], self._lnprob[it:it + n], accepted)
                self._accepted += accepted
                self.iteration += n
                done += n
                continue
            order_b, zz_b, thr_b, pick_b = pending.result() if pending is not None else draw(n)
            pending = None
            if self.block_fn is not None and done + n < nsteps and not device_rng:
                # the next block's numbers are drawn while the library call of this block waits for the device
                pending = self._lookahead.submit(draw, min(chunk, nsteps - done - n))
            it = self.iteration
            if self.block_fn is not None and not device_rng:
                accepted = np.zeros((B, W), dtype=np.int64)
                try:
                    self.block_fn(pos, lnp, order_b, zz_b, thr_b, pick_b, self._chain[it:it + n], self._lnprob[it:it + n], accepted)
                except BaseException:
                    if pending is not None:
                        pending.result()                   # (the generators have moved past the block that was never run)
                    raise
                self._accepted += accepted
                self.iteration += n
                done += n
                continue
            for i in range(n):
                order = order_b[i]
                for h, (first, second) in enumerate(((order[:, :half], order[:, half:]), (order[:, half:], order[:, :half]))):
                    s = pos[rows, first]
                    partners = pos[rows, np.take_along_axis(second, pick_b[i, h], axis=1)]
                    proposal = partners - (partners - s) * zz_b[i, h][:, :, None]
                    new_lnp = np.asarray(self.log_prob_fn(proposal), dtype=np.float64)
                    if np.any(np.isnan(new_lnp)):
                        raise ValueError("Probability function returned NaN")
                    accept = thr_b[i, h] < new_lnp - lnp[rows, first]
                    bb, jj = np.nonzero(accept)
                    ww = first[bb, jj]
                    pos[bb, ww] = proposal[bb, jj]
                    lnp[bb, ww] = new_lnp[bb, jj]
                    self._accepted[bb, ww] += 1
                self._chain[self.iteration] = pos
                self._lnprob[self.iteration] = lnp
                self.iteration += 1
            done += n
        return pos, lnp, self._random.get_state()


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
        plan = self._plan()
        values = np.asarray(values, dtype=np.float64)
        if plan.simple and values.ndim == 3 and values.shape[0] == self.n_bins and values.shape[2] == plan.free_idx.size:
            # flat bounds, no expression priors / constraints: the vectorised plan of Runner.lnprob_batch over all bins
            w = values.shape[1]
            full = plan.full(values.reshape(-1, values.shape[2]))
            ok = plan.prior_ok(full)
            out = np.full(self.n_bins * w, -np.inf)
            if ok.any():
                if not ok.all():
                    full[~ok] = full[int(np.flatnonzero(ok)[0])]
                cat = self._ensure_catalog()
                table = plan.table(full)
                ll = (cat.loglike(table.reshape(self.n_bins, w, -1)) if self.n_bins > 1 else
                      cat.loglike(table.reshape(w, -1))[None, :]).reshape(-1)
                out[ok] = ll[ok]
            return out.reshape(self.n_bins, w)
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

    RNG = "device"                 # where BinnedSampler's random numbers come from ("host": NumPy's generator, as emcee)

    def __call__(self, n_walkers=100, n_steps=100, pos=None, seed=None, rng=None, **kwargs):
        """Run the B ensembles in lockstep; returns a ``BinnedSampler`` (``.chain``: (B, W, steps, P)).  ``rng``: see
        ``BinnedSampler`` (default: the class attribute ``RNG``)."""
        if kwargs.get("n_threads", 1) != 1:
            raise ValueError("n_threads > 1 is not supported by the GPU backend.")
        if pos is None:
            pos = np.stack([self.get_initials(n_walkers) for _ in range(self.n_bins)])
        pos = np.asarray(pos, dtype=np.float64)
        # several ranks (stars sharded): rank 0's start positions and ONE seed go to every rank, as in Runner.__call__
        group = self._rank_group()
        if group is not None:
            pos = group.bcast_array(pos, src=0)
            seed = int(group.bcast_json(int(seed) if seed is not None else int(np.random.SeedSequence().entropy % (2 ** 63)), src=0))
        lp = self.parameters.lnprior_batch(self.parameters.resolve_batch(pos.reshape(-1, pos.shape[-1])))
        if not np.all(np.isfinite(lp)):
            raise ValueError("Invalid initial guesses for {0} walker(s).".format(int(np.sum(~np.isfinite(lp)))))
        # box priors: whole blocks of steps inside the library (mcd_stretch_move with n_bins = B, Runner._stretch_block)
        # (library blocks only for the package's own posterior and box priors: a subclass that overrides a posterior method
        # outside the package is driven by the NumPy loop around ITS lnprob_batch -- Runner.resident_ok)
        native_ok = self.resident_ok()[0] and self.n_bins > 1
        sampler = BinnedSampler(self.n_bins, n_walkers, self.n_fitted_parameters, self.lnprob_batch, seed=seed,
                                block_fn=self._stretch_block if native_ok else None, rng=rng or self.RNG,
                                seeded_block_fn=self._stretch_block_seeded if native_ok else None)
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
