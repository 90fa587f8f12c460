"""CPU: the counter-based random numbers of seeded stretch-move blocks (csrc/mcd_rng.h, host build through tests/emul).

The generator is pinned against NumPy's implementation of the same algorithm (`numpy.random.Philox`, Philox4x64-10): NumPy
increments the counter BEFORE it generates a block of four words, so `Philox(counter=c, key=k).random_raw(4)` is the block of
counter c + 1.  The reference itself draws from NumPy's Mersenne twister through emcee (analysis/runner.py:403-419): what
has to hold for the chain is the distribution of the numbers, not the stream -- checked below against the host sampler's
`draw()` conventions (mcmc_dynamics_amd/sampler.py)."""
import ctypes

import numpy as np
import pytest
from scipy import stats

import emul_helper as eh


def numpy_block(counter, key):
    """The four words NumPy's Philox produces for `counter` (a 256-bit little-endian counter of four words)."""
    c = [int(x) for x in counter]
    value = sum(x << (64 * i) for i, x in enumerate(c))
    value = (value - 1) % (1 << 256)                          # NumPy pre-increments
    before = [(value >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    bg = np.random.Philox(counter=np.array(before, dtype=np.uint64), key=np.array(key, dtype=np.uint64))
    return bg.random_raw(4)


def test_philox_matches_numpys_bit_generator():
    rng = np.random.default_rng(11)
    cases = [([0, 0, 0, 0], [0, 0]), ([1, 0, 0, 0], [0, 0]), ([2**64 - 1] * 4, [2**64 - 1] * 2),
             ([0x243f6a8885a308d3, 0x13198a2e03707344, 0xa4093822299f31d0, 0x082efa98ec4e6c89],
              [0x452821e638d01377, 0xbe5466cf34e90c6c])]
    for _ in range(200):
        cases.append((rng.integers(0, 2**64, 4, dtype=np.uint64).tolist(), rng.integers(0, 2**64, 2, dtype=np.uint64).tolist()))
    for counter, key in cases:
        got = eh.philox(counter, key)
        want = numpy_block(counter, key)
        assert np.array_equal(got, want), (counter, key, got, want)


def test_philox_known_answer_of_the_random123_distribution():
    # kat_vectors of Random123 (philox4x64 10): counter 0, key 0 and the all-ones / pi-digits cases
    got = eh.philox([0, 0, 0, 0], [0, 0])
    assert [hex(int(x)) for x in got] == ["0x16554d9eca36314c", "0xdb20fe9d672d0fdc", "0xd7e772cee186176b", "0x7e68b68aec7ba23b"]
    got = eh.philox([2**64 - 1] * 4, [2**64 - 1] * 2)
    assert [hex(int(x)) for x in got] == ["0x87b092c3013fe90b", "0x438c3c67be8d0224", "0x9cc7d7c69cd777b6", "0xa09caebf594f0ba0"]
    got = eh.philox([0x243f6a8885a308d3, 0x13198a2e03707344, 0xa4093822299f31d0, 0x082efa98ec4e6c89],
                    [0x452821e638d01377, 0xbe5466cf34e90c6c])
    assert [hex(int(x)) for x in got] == ["0xa528f45403e61d95", "0x38c72dbd566e9788", "0xa5a1610e72fd18b5", "0x57bd43b5e52b7fe6"]


def test_deterministic_logarithm_is_accurate_to_a_few_ulp():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.random(200000), 10.0 ** rng.uniform(-300, 300, 50000), [1.0, 0.5, 2.0, np.sqrt(2.0), np.nextafter(1.0, 0),
                        np.nextafter(1.0, 2), 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, 2.0 ** -53]])
    got = eh.det_log(x)
    want = np.log(x)
    ulp = np.spacing(np.maximum(np.abs(want), 1e-300))
    near_one = np.abs(x - 1.0) < 0.3                         # log x -> 0: the error is relative to x - 1 there
    assert np.max(np.abs(got - want)[~near_one] / ulp[~near_one]) <= 3.0
    assert np.max(np.abs(got - want)[near_one] / np.maximum(np.abs(want[near_one]), 1e-17)) < 1e-15
    assert eh.det_log([1.0])[0] == 0.0
    assert eh.det_log([0.0])[0] == -np.inf


def test_chain_numbers_are_a_function_of_their_coordinates():
    """Any block of steps gives the numbers of those steps: a chain cut into blocks of any length is the same chain."""
    full = eh.chain_numbers(77, 0, 12, 3, 10, 4)
    for start, n in ((0, 5), (5, 7), (11, 1)):
        part = eh.chain_numbers(77, start, n, 3, 10, 4)
        for a, b in zip(full, part):
            assert np.array_equal(a[start:start + n], b)
    other = eh.chain_numbers(78, 0, 12, 3, 10, 4)
    assert not np.array_equal(full[1], other[1])
    # the library's host entry point (no device needed) returns the same arrays
    from mcmc_dynamics_amd import _native as native
    order, zz, thr, pick = native.chain_numbers(77, 0, 12, 3, 10, 4)
    for a, b in zip(full, (order, zz, thr, pick)):
        assert np.array_equal(a, b)


def test_chain_numbers_have_the_stretch_moves_distributions():
    S, B, W, P = 400, 4, 64, 5
    order, zz, thr, pick = eh.chain_numbers(2024, 0, S, B, W, P)
    half = W // 2
    # every row of `order` is a permutation, uniform: each walker is in the first half with probability 1/2, and the position
    # of walker 0 is uniform over 0..W-1
    assert np.array_equal(np.sort(order, axis=-1), np.broadcast_to(np.arange(W), order.shape))
    where0 = np.argmax(order == 0, axis=-1).ravel()
    assert stats.chisquare(np.bincount(where0, minlength=W)).pvalue > 1e-3
    first = (np.argsort(order, axis=-1) < half)            # walker w in the first half?
    assert abs(first.mean() - 0.5) < 1e-9                   # (exactly half of every row)
    pair = first[..., 0] & first[..., 1]                    # walkers 0 and 1 together: (half / W) ((half - 1) / (W - 1))
    p = 0.5 * (half - 1) / (W - 1)
    assert abs(pair.mean() - p) < 5 * np.sqrt(p * (1 - p) / pair.size)
    # z ~ g(z) ~ 1 / sqrt(z) on [1/2, 2]: cdf (sqrt(2 z) - 1)
    z = zz.ravel()
    assert z.min() >= 0.5 and z.max() <= 2.0
    assert stats.kstest(z, lambda t: np.sqrt(2.0 * t) - 1.0).pvalue > 1e-3
    # thr = log u - (P - 1) log z with u uniform and independent of z
    u = np.exp(thr.ravel() + (P - 1) * np.log(z))
    assert stats.kstest(u, "uniform").pvalue > 1e-3
    assert abs(stats.spearmanr(u[:100000], z[:100000]).statistic) < 0.02
    # partners uniform over the other half
    assert pick.min() >= 0 and pick.max() == half - 1
    assert stats.chisquare(np.bincount(pick.ravel(), minlength=half)).pvalue > 1e-3
    # numbers of different half steps, ensembles and slots differ
    assert len(np.unique(zz)) == zz.size


def test_walker_limit_of_the_ordering_keys():
    from mcmc_dynamics_amd import _native as native
    with pytest.raises(native.NativeError, match="n_walkers"):
        native.chain_numbers(1, 0, 1, 1, (1 << 20) + 2, 3)
    with pytest.raises(native.NativeError):
        native.chain_numbers(1, 0, 1, 1, 7, 3)                 # odd


def test_ordering_keys_rank_like_a_stable_argsort():
    keys = eh.chain_keys(5, 9, 2, 50)
    order = eh.chain_numbers(5, 9, 1, 3, 50, 3)[0][0, 2]
    assert np.array_equal(order, np.argsort(keys, kind="stable"))
