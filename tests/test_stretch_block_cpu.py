"""csrc/mcd_stretch.h on the CPU: the library's stretch-move block (what mcd_stretch_move runs between kernel launches)
against the Python loop of mcmc_dynamics_amd/sampler.py from the same generator state -- bit-identical chains, prior box
handling included.  The likelihood is a Python callback here; on the GPU it is mcd_loglike_batch (tests/test_gpu_runner.py)."""
import ctypes

import numpy as np
import pytest

import emul_helper as em
from mcmc_dynamics_amd.sampler import EnsembleSampler

EVAL = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int64, ctypes.POINTER(ctypes.c_double))


def _lnlike(table):
    """Some smooth function of the kernel table rows (a correlated Gaussian with a banana): (n, K) -> (n,)."""
    t = np.asarray(table)
    return -0.5 * ((t[:, 0] - 1.0) ** 2 / 0.5 + (t[:, 1] - 0.1 * t[:, 0] ** 2) ** 2 / 2.0 + np.sum(t[:, 2:] ** 2, axis=1))


def _block_fn(src, const, fac, lo, hi, fixed_ok, calls):
    lib = em.lib()
    k = len(src)

    @EVAL
    def cb(tab, n, out):
        table = np.ctypeslib.as_array(tab, shape=(n, k))
        calls.append(table.copy())
        res = _lnlike(table)
        np.ctypeslib.as_array(out, shape=(n,))[:] = res
        return 0

    def run(pos, lnp, order, zz, thr, pick, chain, lnprob_chain, accepted):
        p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None
        rc = lib.emul_stretch_block(ctypes.c_int64(1), ctypes.c_int64(pos.shape[0]), pos.shape[1], k, p(src, ctypes.c_int32), p(const, ctypes.c_double),
                                    p(fac, ctypes.c_double), p(lo, ctypes.c_double), p(hi, ctypes.c_double), int(fixed_ok),
                                    ctypes.c_int64(order.shape[0]), p(pos, ctypes.c_double), p(lnp, ctypes.c_double),
                                    p(order, ctypes.c_int32), p(zz, ctypes.c_double), p(thr, ctypes.c_double), p(pick, ctypes.c_int32),
                                    p(chain, ctypes.c_double), p(lnprob_chain, ctypes.c_double), p(accepted, ctypes.c_int64), cb)
        assert rc == 0, rc
    run.keep = cb
    return run


@pytest.mark.parametrize("case", ["identity", "fixed column and unit factor", "tight bounds"])
def test_library_stretch_block_equals_the_python_loop(case):
    P = 4
    lo, hi = np.full(P, -np.inf), np.full(P, np.inf)
    if case == "identity":
        src, const, fac = np.arange(P, dtype=np.int32), np.zeros(P), np.ones(P)
    elif case == "fixed column and unit factor":
        # kernel table: [free 0, fixed 0.25, free 1 x 60 (arcmin -> arcsec), free 3, free 2]
        src, const, fac = np.array([0, -1, 1, 3, 2], dtype=np.int32), np.array([0, 0.25, 0, 0, 0.0]), np.array([1, 1, 60.0, 1, 1.0])
    else:
        src, const, fac = np.arange(P, dtype=np.int32), np.zeros(P), np.ones(P)
        lo, hi = np.array([0.2, -np.inf, -0.8, -np.inf]), np.array([np.inf, 1.5, 0.9, np.inf])     # many proposals rejected

    def lnprob(values):                                   # what Runner.lnprob_batch does for a box prior
        values = np.atleast_2d(values)
        ok = ~np.isnan(values).any(axis=1) & (values >= lo).all(axis=1) & (values <= hi).all(axis=1)
        out = np.full(len(values), -np.inf)
        if ok.any():
            v = values.copy()
            v[~ok] = v[int(np.flatnonzero(ok)[0])]
            table = np.where(src >= 0, v[:, np.maximum(src, 0)] * fac, const)
            out[ok] = _lnlike(table)[ok]
        return out

    rng = np.random.default_rng(1)
    start = np.array([1.0, 0.3, 0.0, 0.0]) + 0.3 * rng.normal(size=(24, P))
    start[:, 0] = np.abs(start[:, 0]) + 0.25
    start[:, 2] = np.clip(start[:, 2], -0.7, 0.8)
    start[:, 1] = np.minimum(start[:, 1], 1.4)
    ref = EnsembleSampler(24, P, lnprob, vectorize=True, seed=77)
    ref.block_steps = 64                                   # several blocks (and the look-ahead drawing thread) in 150 steps
    ref.run_mcmc(start, 150)
    calls = []
    nat = EnsembleSampler(24, P, lnprob, vectorize=True, seed=77, block_fn=_block_fn(src, const, fac, lo, hi, True, calls))
    nat.block_steps = 64
    pos, lnp, _ = nat.run_mcmc(start, 150)
    assert np.array_equal(nat.chain, ref.chain) and np.array_equal(nat.lnprobability, ref.lnprobability)
    assert np.array_equal(nat.acceptance_fraction, ref.acceptance_fraction) and nat.n_calls == ref.n_calls
    assert np.array_equal(pos, ref.chain[:, -1, :]) and nat.iteration == 150
    assert 0.1 < nat.acceptance_fraction.mean() < 0.9
    assert all(c.shape == (12, len(src)) for c in calls)
    if case == "tight bounds":
        assert np.all(nat.chain[..., 0] >= 0.2) and np.all(nat.chain[..., 2] <= 0.9)          # never accepted outside the box
        assert np.any(~np.isfinite(lnprob(np.array([[0.1, 0, 0, 0.0]]))))
    if case != "identity":
        assert all(np.all(c[:, 1] == 0.25) for c in calls) if case.startswith("fixed") else True
    # a second run continues from the state (restart semantics of Runner.__call__ with n_out)
    ref.run_mcmc(ref.chain[:, -1, :], 70, log_prob0=ref.lnprobability[:, -1])
    nat.run_mcmc(pos, 70, log_prob0=lnp)
    assert np.array_equal(nat.chain, ref.chain) and nat.chain.shape == (24, 220, P)


def test_fixed_parameter_outside_its_bounds_rejects_everything():
    src, const, fac = np.arange(4, dtype=np.int32), np.zeros(4), np.ones(4)
    lo, hi = np.full(4, -np.inf), np.full(4, np.inf)
    calls = []
    start = np.random.default_rng(2).normal(size=(16, 4))
    s = EnsembleSampler(16, 4, lambda v: np.full(len(v), -np.inf), vectorize=True, seed=3,
                        block_fn=_block_fn(src, const, fac, lo, hi, False, calls))
    pos, lnp, _ = s.run_mcmc(start, 10, log_prob0=np.zeros(16))
    assert np.array_equal(pos, start) and not calls and np.all(s.acceptance_fraction == 0)


def test_binned_block_equals_the_python_loop_of_the_binned_sampler():
    """n_bins = B: B lock-stepped ensembles that share every evaluation (reference: one MCMC per radial bin,
    bin/run_tests.py:75-124).  The library's block against the NumPy loop of analysis.binned.BinnedSampler, same draws."""
    from mcmc_dynamics_amd.analysis.binned import BinnedSampler
    lib = em.lib()
    B, W, P = 5, 16, 4
    lo, hi = np.array([0.2, -np.inf, -0.8, -np.inf]), np.array([np.inf, 1.5, 0.9, np.inf])
    src, const, fac = np.arange(P, dtype=np.int32), np.zeros(P), np.ones(P)
    shift = 0.3 * np.arange(B)[:, None]                       # every bin has its own posterior

    def lnlike_rows(table, w):                                # (B * w, K) bin-major -> (B * w,)
        t = np.asarray(table).reshape(B, w, -1)
        return (-0.5 * ((t[..., 0] - 1.0 - shift) ** 2 / 0.5 + (t[..., 1] - 0.1 * t[..., 0] ** 2) ** 2 / 2.0 +
                        np.sum(t[..., 2:] ** 2, axis=-1))).reshape(-1)

    def lnprob(values):                                       # BinnedConstantFit.lnprob_batch for a box prior
        v = np.asarray(values, dtype=np.float64)
        flat = v.reshape(-1, P).copy()
        ok = ~np.isnan(flat).any(axis=1) & (flat >= lo).all(axis=1) & (flat <= hi).all(axis=1)
        out = np.full(flat.shape[0], -np.inf)
        if ok.any():
            flat[~ok] = flat[int(np.flatnonzero(ok)[0])]
            out[ok] = lnlike_rows(flat, v.shape[1])[ok]
        return out.reshape(B, v.shape[1])

    sizes = []

    @EVAL
    def cb(tab, n, out):
        sizes.append(n)
        table = np.ctypeslib.as_array(tab, shape=(B * n, P))
        np.ctypeslib.as_array(out, shape=(B * n,))[:] = lnlike_rows(table, n)
        return 0

    def block_fn(pos, lnp, order, zz, thr, pick, chain, lnprob_chain, accepted):
        p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None
        rc = lib.emul_stretch_block(ctypes.c_int64(B), ctypes.c_int64(W), P, P, p(src, ctypes.c_int32), p(const, ctypes.c_double),
                                    p(fac, ctypes.c_double), p(lo, ctypes.c_double), p(hi, ctypes.c_double), 1,
                                    ctypes.c_int64(order.shape[0]), p(pos, ctypes.c_double), p(lnp, ctypes.c_double),
                                    p(order, ctypes.c_int32), p(zz, ctypes.c_double), p(thr, ctypes.c_double), p(pick, ctypes.c_int32),
                                    p(chain, ctypes.c_double), p(lnprob_chain, ctypes.c_double), p(accepted, ctypes.c_int64), cb)
        assert rc == 0, rc

    rng = np.random.default_rng(4)
    start = np.array([1.0, 0.3, 0.0, 0.0]) + 0.3 * rng.normal(size=(B, W, P)) + np.concatenate([shift, np.zeros((B, 3))], axis=1)[:, None, :]
    start[..., 0] = np.abs(start[..., 0]) + 0.25
    start[..., 2] = np.clip(start[..., 2], -0.7, 0.8)
    start[..., 1] = np.minimum(start[..., 1], 1.4)
    ref = BinnedSampler(B, W, P, lnprob, seed=9)
    nat = BinnedSampler(B, W, P, lnprob, seed=9, block_fn=block_fn)
    ref.block_steps = nat.block_steps = 16                    # several blocks in 40 steps
    ref.run_mcmc(start, 40)
    pos, lnp, _ = nat.run_mcmc(start, 40)
    assert np.array_equal(nat.chain, ref.chain) and np.array_equal(nat.lnprobability, ref.lnprobability)
    assert np.array_equal(nat.acceptance_fraction, ref.acceptance_fraction) and nat.chain.shape == (B, W, 40, P)
    assert set(sizes) == {W // 2} and np.array_equal(pos, ref.chain[:, :, -1, :])
    assert np.all(nat.chain[..., 0] >= 0.2) and np.all(nat.chain[..., 2] <= 0.9)
    acc = nat.acceptance_fraction
    assert 0.05 < acc.mean() < 0.95 and np.all(acc.reshape(B, -1).mean(axis=1) > 0.02)       # every bin moves
    # the bins are independent chains: bin 2's chain does not change when the other bins start elsewhere
    other = start.copy()
    other[[0, 1, 3, 4]] += 0.05
    other[..., 0] = np.maximum(other[..., 0], 0.25)
    alt = BinnedSampler(B, W, P, lnprob, seed=9, block_fn=block_fn)
    alt.block_steps = 16
    alt.run_mcmc(other, 40)
    assert np.array_equal(alt.chain[2], nat.chain[2]) and not np.array_equal(alt.chain[0], nat.chain[0])
    # a second run continues from the state
    ref.run_mcmc(ref.chain[:, :, -1, :], 7, log_prob0=ref.lnprobability[:, :, -1])
    nat.run_mcmc(pos, 7, log_prob0=lnp)
    assert np.array_equal(nat.chain, ref.chain) and nat.iteration == 47
    # the counter-based numbers of csrc/mcd_rng.h (rng="device"): the library's block loop fed block by block with the host
    # build's numbers (what mcd_stretch_move_seeded does when it runs host-driven) == the NumPy loop fed by the library's own
    # mcd_chain_numbers, whatever the block lengths
    def seeded_block_fn(pos, lnp, seed, step0, n, chain, lnprob_chain, accepted):
        order, zz, thr, pick = em.chain_numbers(seed, step0, n, B, W, P)
        block_fn(pos, lnp, order, zz, thr, pick, chain, lnprob_chain, accepted)

    dev_ref = BinnedSampler(B, W, P, lnprob, seed=31, rng="device")
    dev_nat = BinnedSampler(B, W, P, lnprob, seed=31, rng="device", seeded_block_fn=seeded_block_fn)
    dev_ref.device_block_steps, dev_nat.device_block_steps = 40, 13
    dev_ref.run_mcmc(start, 40)
    dev_nat.run_mcmc(start, 40)
    assert np.array_equal(dev_nat.chain, dev_ref.chain) and np.array_equal(dev_nat.lnprobability, dev_ref.lnprobability)
    assert np.array_equal(dev_nat.acceptance_fraction, dev_ref.acceptance_fraction)
    assert not np.array_equal(dev_nat.chain, nat.chain[:, :, :40]) and 0.05 < dev_nat.acceptance_fraction.mean() < 0.95


def test_device_rng_mode_python_loop_is_block_partition_invariant():
    """rng="device" (csrc/mcd_rng.h): the chain is a function of (seed, step, bin, walker) -- the Python loops of both
    samplers, fed by `_native.chain_numbers`, give the same chain however a run is cut into blocks or run_mcmc calls, and
    the chain samples the target (a standard normal: mean and variance within sampling error)."""
    from mcmc_dynamics_amd.analysis.binned import BinnedSampler
    from mcmc_dynamics_amd.sampler import EnsembleSampler

    def lnprob(values):
        return -0.5 * np.sum(np.asarray(values) ** 2, axis=-1)

    B, W, P = 3, 12, 2
    rng = np.random.default_rng(0)
    start = rng.normal(size=(B, W, P))
    a = BinnedSampler(B, W, P, lnprob, seed=123, rng="device")
    a.run_mcmc(start, 40)
    b = BinnedSampler(B, W, P, lnprob, seed=123, rng="device")
    b.device_block_steps = 7
    pos, lnp, _ = b.run_mcmc(start, 13)
    b.run_mcmc(pos, 27, log_prob0=lnp)
    assert np.array_equal(a.chain, b.chain) and np.array_equal(a.lnprobability, b.lnprobability)
    c = BinnedSampler(B, W, P, lnprob, seed=124, rng="device")
    c.run_mcmc(start, 40)
    assert not np.array_equal(a.chain, c.chain)
    for s in (a, b, c):
        s.close()

    e = EnsembleSampler(W, P, lnprob, vectorize=True, seed=5, rng="device")
    e.run_mcmc(start[0], 6000)
    f = EnsembleSampler(W, P, lnprob, vectorize=True, seed=5, rng="device")
    f.block_steps = 37
    pos, lnp, _ = f.run_mcmc(start[0], 250)
    f.run_mcmc(pos, 5750, log_prob0=lnp)
    assert np.array_equal(e.chain, f.chain)
    flat = e.chain[:, 200:].reshape(-1, P)
    assert np.all(np.abs(flat.mean(axis=0)) < 0.1) and np.all(np.abs(flat.var(axis=0) - 1.0) < 0.1)
    assert 0.4 < e.acceptance_fraction.mean() < 0.95
    with pytest.raises(ValueError):
        EnsembleSampler(W, P, lnprob, rng="device", a=3.0)
