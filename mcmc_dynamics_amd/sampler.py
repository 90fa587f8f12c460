"""Affine-invariant ensemble sampler (Goodman & Weare 2010 stretch move) with emcee's interface.

The reference hands ``Runner.lnprob`` to ``emcee.EnsembleSampler`` (analysis/runner.py:403) and lets
emcee drive the outer loop.  emcee is not installed in the target image, so ``Runner.__call__`` uses
the real emcee when it can be imported and this class otherwise.  It implements the part of emcee 3's
``EnsembleSampler`` that the reference touches -- ``run_mcmc``, ``chain``, ``lnprobability``,
``iteration``, ``acceptance_fraction`` -- with emcee's default move: the ensemble is split in two
random halves, each half is updated against the other (``z ~ g(z) \\propto 1/sqrt(z)`` on [1/a, a],
``a = 2``), so one step costs two batched posterior calls of W/2 proposals each.

``vectorize=True`` (the mode the GPU backend uses) passes a ``(n, ndim)`` array to ``log_prob_fn`` and
expects ``(n,)`` back; otherwise the function is mapped over the rows (optionally with ``pool.map``).

``block_fn`` (optional, what ``Runner`` passes for box priors): a callable that advances the ensemble by a whole block
of steps from the random numbers drawn here -- ``libmcd_hip.so``'s ``mcd_stretch_move`` runs the same half-step loop in
C++ (csrc/mcd_stretch.h), bit-identical to the Python loop below, with a few microseconds of host time between two
kernel launches instead of ~50.
"""
import numpy as np


_DRAW_POOL = None


def _draw_pool():
    """A few helper threads shared by all samplers of the process for the row-wise part of the draws (see ``draw``)."""
    global _DRAW_POOL
    if _DRAW_POOL is None:
        try:
            from concurrent.futures import ThreadPoolExecutor
            _DRAW_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix="mcd-draw")
        except Exception:                                    # pragma: no cover
            _DRAW_POOL = False
    return _DRAW_POOL or None


class EnsembleSampler(object):

    def __init__(self, nwalkers, ndim, log_prob_fn, pool=None, a=2.0, vectorize=False, seed=None, block_fn=None, rng="host",
                 seeded_block_fn=None):
        """``rng="device"``: the move's random numbers come from the counter-based generator of csrc/mcd_rng.h instead of
        NumPy's Mersenne twister -- a function of (seed, step, half step, walker) alone, generated on the device (csrc/mcd_stretch.hip: chain_numbers_kernel) by
        ``seeded_block_fn`` (``Runner._stretch_block_seeded``), or taken from ``_native.chain_numbers`` by the Python loop
        below when there is none: the same chain either way, and however it is cut into blocks."""
        if rng not in ("host", "device"):
            raise ValueError("rng must be 'host' or 'device'")
        if rng == "device" and float(a) != 2.0:
            raise ValueError("rng='device' implements the stretch move with a = 2 (emcee's default)")
        self.rng = rng
        self.seeded_block_fn = seeded_block_fn
        self.seed64 = None
        if rng == "device":                       # (only then: drawing a seed moves NumPy's global generator)
            self.seed64 = int(seed) & 0xFFFFFFFFFFFFFFFF if seed is not None else \
                (int(np.random.randint(0, 2 ** 32)) << 32) | int(np.random.randint(0, 2 ** 32))
        if nwalkers < 2 * ndim:
            raise ValueError("The number of walkers must be at least twice the dimension.")   # as emcee
        if nwalkers % 2:
            raise ValueError("The number of walkers must be even.")
        self.nwalkers, self.ndim = int(nwalkers), int(ndim)
        self.log_prob_fn = log_prob_fn
        self.pool = pool
        self.a = float(a)
        self.vectorize = bool(vectorize)
        self.block_fn = block_fn
        # steps whose random numbers are drawn together (and, with block_fn, run as one library call: the device then
        # idles only once per block while the results travel back and the next block goes up, ~0.1 ms).  Part of the
        # definition of the random stream: samplers that should produce the same chain need the same value.
        self.block_steps = 256
        # ... except the FIRST block of a run with several blocks: its draws cannot overlap anything (nothing runs yet), so it
        # is kept short -- the device starts after 0.8 ms of draws instead of 3 ms (1e5 stars x 256 walkers: 3 of 39 us per
        # step over a 1024-step run).  Also part of the definition of the random stream.
        self.first_block_steps = 64
        self._random = np.random.RandomState(seed)
        self.reset()

    def reset(self):
        self.iteration = 0
        self._chain = np.empty((0, self.nwalkers, self.ndim))
        self._lnprob = np.empty((0, self.nwalkers))
        self._accepted = np.zeros(self.nwalkers)
        self.n_calls = 0

    # ------------------------------------------------------------------ emcee-compatible views
    @property
    def chain(self):
        """(nwalkers, nsteps, ndim), the layout the reference pickles (runner.py:471-472)."""
        return np.swapaxes(self._chain[:self.iteration], 0, 1)

    @property
    def lnprobability(self):
        return np.swapaxes(self._lnprob[:self.iteration], 0, 1)

    @property
    def flatchain(self):
        return self._chain[:self.iteration].reshape(-1, self.ndim)

    @property
    def acceptance_fraction(self):
        return self._accepted / max(1, self.iteration)

    def get_chain(self, discard=0, flat=False):
        c = self._chain[discard:self.iteration]
        return c.reshape(-1, self.ndim) if flat else c

    def get_log_prob(self, discard=0, flat=False):
        lp = self._lnprob[discard:self.iteration]
        return lp.reshape(-1) if flat else lp

    @property
    def random_state(self):
        return self._random.get_state()

    # ------------------------------------------------------------------ posterior calls
    def compute_log_prob(self, coords):
        coords = np.asarray(coords, dtype=np.float64)
        if not np.isfinite(coords).all():
            raise ValueError("At least one parameter value was infinite or NaN")
        if self.vectorize:
            lp = np.asarray(self.log_prob_fn(coords), dtype=np.float64)
        elif self.pool is not None:
            lp = np.array([float(x) for x in self.pool.map(self.log_prob_fn, list(coords))], dtype=np.float64)
        else:
            lp = np.array([float(self.log_prob_fn(c)) for c in coords], dtype=np.float64)
        self.n_calls += 1
        if lp.shape != (coords.shape[0],):
            raise ValueError("log_prob_fn returned shape {0} for {1} positions".format(lp.shape, coords.shape[0]))
        if np.isnan(lp).any():
            raise ValueError("Probability function returned NaN")
        return lp

    # ------------------------------------------------------------------ sampling
    def run_mcmc(self, initial_state, nsteps, log_prob0=None, rstate0=None, progress=False, store=True, **kwargs):
        """Advance the ensemble by ``nsteps``.  Returns ``(pos, log_prob, random_state)``."""
        pos = np.array(initial_state, dtype=np.float64)
        if pos.shape != (self.nwalkers, self.ndim):
            raise ValueError("incompatible input dimensions {0}".format(pos.shape))
        if rstate0 is not None:
            self._random.set_state(rstate0)
        lnp = self.compute_log_prob(pos) if log_prob0 is None else np.array(log_prob0, dtype=np.float64)
        if np.shape(lnp) != (self.nwalkers,):
            raise ValueError("incompatible input dimensions for log_prob0")
        if store:
            need = self.iteration + int(nsteps)
            if need > self._chain.shape[0]:               # geometric growth: amortised O(1) per stored step
                cap = max(need, 2 * self._chain.shape[0])
                chain = np.empty((cap, self.nwalkers, self.ndim))
                lnprob = np.empty((cap, self.nwalkers))
                chain[:self.iteration] = self._chain[:self.iteration]
                lnprob[:self.iteration] = self._lnprob[:self.iteration]
                self._chain, self._lnprob = chain, lnprob
        half = self.nwalkers // 2
        rnd = self._random
        nsteps = int(nsteps)
        fast_eval = self.vectorize                     # skip the generic wrapper's per-call conversions in the hot loop
        inv_a, am1, dm1 = 1.0 / self.a, self.a - 1.0, self.ndim - 1.0
        done = 0

        def draw(block):
            # Random numbers for a block of steps in a handful of vectorised draws (the per-step host cost is what limits
            # the sampler once the posterior call takes ~0.1 ms): split of the ensemble = argsort of uniform keys,
            # stretch factors z ~ g(z) and log acceptance thresholds, partner indices.  The generator is consumed in this
            # order by ONE thread (the stream is the serial one); what follows the raw draws -- the row-wise argsort and the
            # logarithms, 80 % of the time -- is a pure function of them and is spread over a few threads by rows (NumPy
            # releases the interpreter lock there): at 1e5 stars a 256-step block of 256 walkers takes the device 9 ms and
            # one host thread 8 ms to draw.
            keys = rnd.rand(block, self.nwalkers)
            u = rnd.rand(block, 4, half)
            pick_b = rnd.randint(half, size=(block, 2, half))
            order_b = np.empty((block, self.nwalkers), dtype=np.int64)
            zz_b = np.empty((block, 2, half))
            thr_b = np.empty((block, 2, half))

            def rows(lo, hi):
                order_b[lo:hi] = np.argsort(keys[lo:hi], axis=1)
                z = am1 * u[lo:hi, :2] + 1.0
                z *= z
                z *= inv_a
                zz_b[lo:hi] = z
                thr_b[lo:hi] = np.log(u[lo:hi, 2:]) - dm1 * np.log(z)      # accept iff thr < new_lnp - old_lnp

            workers = _draw_pool()
            if workers is None or block < 32:
                rows(0, block)
            else:
                n_parts = 4
                edges = [block * k // n_parts for k in range(n_parts + 1)]
                for f in [workers.submit(rows, edges[k], edges[k + 1]) for k in range(n_parts)]:
                    f.result()
            if self.block_fn is not None:
                return (np.ascontiguousarray(order_b, dtype=np.int32), np.ascontiguousarray(zz_b), np.ascontiguousarray(thr_b),
                        np.ascontiguousarray(pick_b, dtype=np.int32))
            return order_b, zz_b, thr_b, pick_b

        # With the loop inside the library the draws of the NEXT block (~2 ms per 64 steps of 256 walkers: a seventh of the
        # block's device time) are made by a helper thread while the library call of the current block waits for the
        # device -- both release the interpreter lock.  One generator, one drawing thread at a time, blocks drawn in order:
        # the stream of random numbers is the serial one.  (If the library call raises, the generator has already moved
        # past the block that was never run.)
        pool = None
        chunk = max(1, int(self.block_steps))
        device_rng = self.rng == "device"
        if device_rng and self.seeded_block_fn is None:
            from . import _native as native

            def draw(block):                                 # noqa: F811 -- the same numbers the device generates
                order_b, zz_b, thr_b, pick_b = native.chain_numbers(self.seed64, self.iteration, block, 1, self.nwalkers,
                                                                    self.ndim, squeeze=True)
                return order_b.astype(np.int64), zz_b, thr_b, pick_b
        if self.block_fn is not None and nsteps > chunk and not device_rng:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=1)
        pending = None
        try:
            # (the same partition into blocks with and without block_fn: the two then consume the generator alike)
            first = max(1, min(chunk, int(self.first_block_steps))) if nsteps > chunk else chunk
            if device_rng:
                first = chunk                                 # (no draws to hide: equal blocks)
            while done < nsteps:
                block = min(first if done == 0 else chunk, nsteps - done)
                if device_rng and self.seeded_block_fn is not None:
                    it = self.iteration
                    accepted = np.zeros(self.nwalkers, dtype=np.int64)
                    self.seeded_block_fn(pos, lnp, self.seed64, it, block, self._chain[it:it + block] if store else None,
                                         self._lnprob[it:it + block] if store else None, accepted)
                    self._accepted += accepted
                    self.iteration += block
                    self.n_calls += 2 * block
                    done += block
                    continue
                order_b, zz_b, thr_b, pick_b = pending.result() if pending is not None else draw(block)
                pending = None
                if pool is not None and done + block < nsteps:
                    pending = pool.submit(draw, min(chunk, nsteps - done - block))
                if self.block_fn is not None and not device_rng:
                    # the same half-step loop, in the library (csrc/mcd_stretch.h): identical numbers, no Python between launches
                    it = self.iteration
                    accepted = np.zeros(self.nwalkers, dtype=np.int64)
                    self.block_fn(pos, lnp, order_b, zz_b, thr_b, pick_b,
                                  self._chain[it:it + block] if store else None, self._lnprob[it:it + block] if store else None,
                                  accepted)
                    self._accepted += accepted
                    self.iteration += block
                    self.n_calls += 2 * block
                    done += block
                    continue
                for i in range(block):
                    order = order_b[i]
                    halves = (order[:half], order[half:])
                    for h in (0, 1):
                        first, second = halves[h], halves[1 - h]
                        s = pos[first]
                        partners = pos[second[pick_b[i, h]]]
                        proposal = partners - (partners - s) * zz_b[i, h][:, None]
                        if fast_eval:
                            new_lnp = np.asarray(self.log_prob_fn(proposal), dtype=np.float64)
                            self.n_calls += 1
                            if new_lnp.shape != (half,):
                                raise ValueError("log_prob_fn returned shape {0} for {1} positions".format(new_lnp.shape, half))
                            if np.isnan(new_lnp).any():
                                raise ValueError("Probability function returned NaN")
                        else:
                            new_lnp = self.compute_log_prob(proposal)
                        accept = thr_b[i, h] < new_lnp - lnp[first]
                        idx = first[accept]
                        pos[idx] = proposal[accept]
                        lnp[idx] = new_lnp[accept]
                        self._accepted[idx] += 1
                    if store:
                        self._chain[self.iteration] = pos
                        self._lnprob[self.iteration] = lnp
                    self.iteration += 1
                done += block
        finally:
            if pool is not None:
                if pending is not None:
                    pending.result()                       # (only when the library call raised)
                pool.shutdown()
        return pos, lnp, self._random.get_state()
