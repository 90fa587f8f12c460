"""GPU: ``mcd_stretch_move`` with the ensemble resident on the device (csrc/mcd_stretch.hip) against the host-driven loop
of the same library (option "device_chain" = 0, csrc/mcd_stretch.h) and against a NumPy restatement of the sampler's
half-step loop (sampler.py / runner.py:403-419).  The bar is bit-identity: the device proposes, applies the prior, builds
the walker constants and judges the range guard with the instructions the host path uses, so the chains must not differ
in a single bit -- also when the device has to give a block back (NaN, re-run request, kernel family change, no proposal
inside the prior), which ``stretch_info`` makes visible."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu
CENTRE = (56.345, -26.675)


@pytest.fixture(scope="module")
def native():
    from mcmc_dynamics_amd import _native
    return _native


@pytest.fixture(scope="module")
def ctx(native):
    return native.default_context()


def block_randoms(rng, n_steps, w, n_dim, a=2.0):
    """The random numbers of a block, drawn the way sampler.py draws them."""
    half = w // 2
    order = np.argsort(rng.random((n_steps, w)), axis=1).astype(np.int32)
    u = rng.random((n_steps, 4, half))
    zz = ((a - 1.0) * u[:, :2] + 1.0) ** 2 / a
    thr = np.log(u[:, 2:]) - (n_dim - 1.0) * np.log(zz)
    pick = rng.integers(0, half, size=(n_steps, 2, half)).astype(np.int32)
    return order, np.ascontiguousarray(zz), np.ascontiguousarray(thr), pick


def identity_plan(k, lo=None, hi=None):
    return {"col_source": np.arange(k, dtype=np.int32), "col_const": np.zeros(k), "col_factor": np.ones(k),
            "lo": np.full(k, -np.inf) if lo is None else np.asarray(lo, dtype=float),
            "hi": np.full(k, np.inf) if hi is None else np.asarray(hi, dtype=float), "fixed_ok": True}


def numpy_block(cat, plan, pos, lnp, order, zz, thr, pick):
    """sampler.py's half-step loop with ``cat.loglike`` as the posterior (box prior from the plan)."""
    pos, lnp = pos.copy(), lnp.copy()
    n_steps, w = order.shape
    half = w // 2
    chain, lnpc, acc = np.empty((n_steps,) + pos.shape), np.empty((n_steps, w)), np.zeros(w, dtype=np.int64)
    src, const, fac = plan["col_source"], plan["col_const"], plan["col_factor"]
    for i in range(n_steps):
        halves = (order[i, :half], order[i, half:])
        for h in (0, 1):
            first, second = halves[h], halves[1 - h]
            s, partners = pos[first], pos[second[pick[i, h]]]
            proposal = partners - (partners - s) * zz[i, h][:, None]
            ok = np.all((proposal >= plan["lo"]) & (proposal <= plan["hi"]), axis=1)
            new = np.full(half, -np.inf)
            if ok.any():
                # always W/2 rows (the chunk table, hence the order of summation, depends on the row count): a row outside
                # the prior is replaced by the first valid one, as Runner.lnprob_batch and the library do
                rows = np.where(src[None, :] >= 0, proposal[:, np.maximum(src, 0)] * fac[None, :], const[None, :])
                rows[~ok] = rows[np.argmax(ok)]
                new[ok] = cat.loglike(np.ascontiguousarray(rows))[ok]
            accept = thr[i, h] < new - lnp[first]
            idx = first[accept]
            pos[idx], lnp[idx] = proposal[accept], new[accept]
            acc[idx] += 1
        chain[i], lnpc[i] = pos, lnp
    return pos, lnp, chain, lnpc, acc


def run_block(cat, plan, pos, lnp, randoms, device):
    cat.set_option("device_chain", int(device))          # 0 host-driven, 1 resident, 2 resident with the general step kernel
    n_steps, w = randoms[0].shape
    pos, lnp = pos.copy(), lnp.copy()
    chain, lnpc, acc = np.empty((n_steps,) + pos.shape), np.empty((n_steps, w)), np.zeros(w, dtype=np.int64)
    cat.stretch_move(plan, pos, lnp, *randoms, chain, lnpc, acc)
    return pos, lnp, chain, lnpc, acc


def same(a, b):
    return all(np.array_equal(x, y, equal_nan=True) for x, y in zip(a, b))


def _walkers(rng, w, model, sv, free):
    cols = [rng.normal(0, 0.2 * sv, w), sv * 10.0 ** rng.uniform(-0.3, 0.3, w)]
    if model >= 3:
        cols.append(10.0 ** rng.uniform(1.0, 2.0, w))
    cols += [rng.normal(0, 0.2 * sv, w), rng.normal(0, 0.2 * sv, w)]
    if model >= 3:
        cols.append(10.0 ** rng.uniform(1.0, 2.0, w))
    if free:
        cols += [CENTRE[0] + rng.normal(0, 0.002, w), CENTRE[1] + rng.normal(0, 0.002, w)]
    if model in (2, 4):
        cols += [rng.normal(0, 0.3 * sv, w), 3 * sv * 10.0 ** rng.uniform(-0.1, 0.1, w), rng.uniform(0.2, 0.8, w)]
    if model == 5:
        cols.append(rng.uniform(0.2, 0.8, w))
    return np.ascontiguousarray(np.stack(cols, axis=1))


def _catalogue(native, ctx, rng, n, model, free):
    from oracle import lnprob_numpy as oracle
    sv = 12.0
    sep = np.maximum(np.abs(rng.normal(0, 2.0 / 60.0, n)), 1e-4)
    th = rng.uniform(-np.pi, np.pi, n)
    ra, dec = CENTRE[0] + sep * np.cos(th) / np.cos(np.radians(CENTRE[1])), CENTRE[1] + sep * np.sin(th)
    v, verr = rng.normal(0, sv, n), rng.uniform(0.5, 4.0, n)
    v[::7] = rng.normal(0, 3 * sv, len(v[::7]))                     # a background population
    lnbg = oracle.gaussian_background(v, verr, 0.0, 3 * sv)
    kw = {}
    if model in (1, 6):
        kw = dict(lnlike_bg=lnbg, pmember=np.clip(rng.random(n), 0.02, 0.98))
    elif model in (2, 4):
        kw = dict(density=np.clip(rng.random(n), 0.05, 1.0))
    elif model == 5:
        kw = dict(lnlike_bg=lnbg, density=np.clip(rng.random(n), 0.05, 1.0))
    return native.Catalog(ctx, ra, dec, v, verr, model=model, centre=None if free else CENTRE, **kw), sv


@pytest.mark.parametrize("model,free", [(0, False), (0, True), (1, False), (2, False), (2, True), (3, False), (4, False),
                                        (5, False), (6, False)])
def test_resident_chain_is_bit_identical_to_the_host_loop(native, ctx, model, free):
    rng = np.random.default_rng(9100 + 10 * model + free)
    cat, sv = _catalogue(native, ctx, rng, 30011, model, free)
    w = 48
    pos = _walkers(rng, w, 5 if model == 6 else model, sv, free)
    if model == 6:
        pos = pos[:, :-1]                                              # PROFILE_BGFIXED: no f_back column
    assert pos.shape[1] == cat.k
    # inclusive box prior that rejects a few proposals per step: sigma_max > 0 and a window around the ensemble
    lo, hi = np.full(cat.k, -np.inf), np.full(cat.k, np.inf)
    lo[1] = 0.0
    lo[0], hi[0] = pos[:, 0].min() - 0.1 * sv, pos[:, 0].max() + 0.1 * sv
    if model in (2, 4, 5):
        lo[-1], hi[-1] = 0.0, 1.0
    if model >= 3:
        lo[2] = lo[5] = 1.0                                            # a, r_peak [arcsec] stay positive
    plan = identity_plan(cat.k, lo, hi)
    lnp = cat.loglike(pos)
    assert np.all(np.isfinite(lnp))
    randoms = block_randoms(rng, 21, w, cat.k)
    before = cat.stretch_info()
    dev = run_block(cat, plan, pos, lnp, randoms, device=True)
    mid = cat.stretch_info()
    host = run_block(cat, plan, pos, lnp, randoms, device=False)
    after = cat.stretch_info()
    assert mid["device_blocks"] == before["device_blocks"] + 1 and mid["discarded_blocks"] == before["discarded_blocks"], mid
    assert after["host_blocks"] == mid["host_blocks"] + 1 and after["device_blocks"] == mid["device_blocks"]
    assert same(dev, host)
    assert same(dev, run_block(cat, plan, pos, lnp, randoms, device=2))          # the step kernel for ensembles of any size
    assert cat.stretch_info()["device_blocks"] == after["device_blocks"] + 1
    assert same(dev, numpy_block(cat, plan, pos, lnp, *randoms))
    acc = dev[4]
    assert 0 < acc.sum() < 21 * w and np.all(np.isfinite(dev[3]))
    # the prior did reject proposals (otherwise the donor-row substitution was not exercised)
    first_half = randoms[0][0, :w // 2]
    prop = pos[randoms[0][0, w // 2:][randoms[3][0, 0]]]
    prop = prop - (prop - pos[first_half]) * randoms[1][0, 0][:, None]
    rejected = ~np.all((prop >= lo) & (prop <= hi), axis=1)
    assert rejected.any()
    # a second block continues from the first one's state (hint of the kernel family kept, arena reused)
    randoms2 = block_randoms(rng, 5, w, cat.k)
    dev2 = run_block(cat, plan, dev[0], dev[1], randoms2, device=True)
    host2 = run_block(cat, plan, host[0], host[1], randoms2, device=False)
    assert same(dev2, host2) and cat.stretch_info()["discarded_blocks"] == before["discarded_blocks"]
    cat.close()


def test_fixed_columns_and_unit_factors(native, ctx):
    """A plan as Runner builds it: one kernel column fixed, one with a unit factor, fewer free parameters than columns."""
    rng = np.random.default_rng(9200)
    cat, sv = _catalogue(native, ctx, rng, 20000, 1, False)
    w = 32
    full = _walkers(rng, w, 1, sv, False)
    pos = np.ascontiguousarray(full[:, [1, 2, 3]])                     # v_sys fixed
    pos[:, 1] *= 60.0                                                 # v_maxx sampled in other units
    plan = {"col_source": np.array([-1, 0, 1, 2], dtype=np.int32), "col_const": np.array([0.3, 0, 0, 0]),
            "col_factor": np.array([1.0, 1.0, 1.0 / 60.0, 1.0]), "lo": np.array([0.0, -np.inf, -np.inf]),
            "hi": np.full(3, np.inf), "fixed_ok": True}
    table = np.column_stack([np.full(w, 0.3), pos[:, 0], pos[:, 1] * (1.0 / 60.0), pos[:, 2]])
    lnp = cat.loglike(table)
    randoms = block_randoms(rng, 17, w, 3)
    dev = run_block(cat, plan, pos, lnp, randoms, device=True)
    host = run_block(cat, plan, pos, lnp, randoms, device=False)
    info = cat.stretch_info()
    assert same(dev, host) and info["device_blocks"] == 1 and info["discarded_blocks"] == 0
    assert same(dev, numpy_block(cat, plan, pos, lnp, *randoms))
    # a fixed parameter outside its own bounds: nothing is ever accepted, no evaluation, the device path stands aside
    plan_bad = dict(plan, fixed_ok=False)
    dev = run_block(cat, plan_bad, pos, lnp, randoms, device=True)
    assert np.array_equal(dev[0], pos) and dev[4].sum() == 0 and cat.stretch_info()["device_blocks"] == 1
    cat.close()


def test_blocks_the_device_gives_back(native, ctx):
    """Each reason for a discard, and that the block then equals the host-driven one."""
    rng = np.random.default_rng(9300)
    # (8) a half step with every proposal outside the prior
    cat, sv = _catalogue(native, ctx, rng, 5000, 0, False)
    w = 16
    pos = _walkers(rng, w, 0, sv, False)
    lnp = cat.loglike(pos)
    plan = identity_plan(4, lo=[1e6, -np.inf, -np.inf, -np.inf])
    randoms = block_randoms(rng, 3, w, 4)
    dev = run_block(cat, plan, pos, lnp, randoms, device=True)
    info = cat.stretch_info()
    assert info["discarded_blocks"] == 1 and info["last_discard_status"] & 8 and info["host_blocks"] == 1
    assert np.array_equal(dev[0], pos) and dev[4].sum() == 0
    cat.close()

    # (4) the kernel family changes inside a block: |v_sys| of the proposals crosses the narrow-range bound
    # d_max^2 <= 2e6 n_min (mcd_guard.h), so some half steps want level 2 and some level 1.  With the guard deferred to the
    # end of the block (option "defer_guard", the default: mcd_stretch.hip: stretch_judge_kernel) and judged launch by launch
    # inside the step kernel: the same blocks discarded, the same kernel family picked for the next block, the same chains.
    cat, sv = _catalogue(native, ctx, rng, 5000, 1, False)
    pos = _walkers(rng, w, 1, sv, False)
    pos[:, 1] = rng.uniform(0.05, 0.2, w)                             # n_min ~ verr_min^2 = 0.25: bound at d_max ~ 707
    pos[:, 0] = rng.uniform(300.0, 640.0, w)
    lnp = cat.loglike(pos)
    plan = identity_plan(4, lo=[-np.inf, 0.0, -np.inf, -np.inf])
    blocks = [block_randoms(rng, 8, w, 4) for _ in range(6)]
    history = {}
    for defer in (1, 0):
        cat.set_option("defer_guard", defer)
        cat.loglike(pos)                                               # (both passes start from the same kernel-family hint)
        seen, before = [], cat.stretch_info()
        for trial, randoms in enumerate(blocks):
            dev = run_block(cat, plan, pos, lnp, randoms, device=True)
            info = cat.stretch_info()
            seen.append((cat.fast_level, info["discarded_blocks"] - before["discarded_blocks"], info["last_discard_status"], dev))
            host = run_block(cat, plan, pos, lnp, randoms, device=False)
            assert same(dev, host), (defer, trial)
        history[defer] = seen
        info = cat.stretch_info()
        assert info["discarded_blocks"] > before["discarded_blocks"] and info["last_discard_status"] & 4, (defer, info)
    for a, b in zip(history[1], history[0]):
        assert a[:3] == b[:3] and same(a[3], b[3]), (a[:3], b[:3])
    cat.close()

    # (2) re-run request of the fast mixture kernels: certain members that are gross outliers (denormal regime)
    g = load_golden("constant_bg_gaussian_fixed")
    pm, v = g["pmember"].copy(), g["v"].copy()
    pm[:3] = 1.0
    v[:3] = [900.0, -1500.0, 4000.0]
    gc = (float(g["ra_center"]), float(g["dec_center"]))
    mix = native.Catalog(ctx, g["ra"], g["dec"], v, g["verr"], model=native.MODEL_CONST_BGFIXED, centre=gc,
                         lnlike_bg=g["lnlike_background"], pmember=pm)
    rows = g["values"][np.isfinite(g["lnprior"]) & (g["values"][:, 1] > 0)]
    pos = np.ascontiguousarray(rows[:16])
    lnp = mix.loglike(pos)
    randoms = block_randoms(rng, 4, 16, 4)
    plan = identity_plan(4, lo=[-np.inf, 0.0, -np.inf, -np.inf])
    reruns = mix.rerun_count
    dev = run_block(mix, plan, pos, lnp, randoms, device=True)
    info = mix.stretch_info()
    assert info["discarded_blocks"] == 1 and info["last_discard_status"] & 2 and mix.rerun_count > reruns
    assert same(dev, run_block(mix, plan, pos, lnp, randoms, device=False))
    mix.close()

    # (1) a NaN log-likelihood: the error of the host loop (emcee: "Probability function returned NaN")
    ra = np.array([CENTRE[0] + 0.01, CENTRE[0] - 0.01, CENTRE[0]])
    dec = np.array([CENTRE[1], CENTRE[1] + 0.01, CENTRE[1] - 0.01])
    nan_cat = native.Catalog(ctx, ra, dec, np.array([1.0, 2.0, 3.0]), np.array([0.0, 1.0, 1.0]), model=0, centre=CENTRE)
    pos = np.zeros((8, 4))
    pos[:, 0] = 1.0                                                    # v_sys = v of the verr = 0 star, sigma = 0: 0 / 0
    lnp = np.zeros(8)
    randoms = block_randoms(rng, 2, 8, 4)
    for device in (True, False):
        with pytest.raises(native.NativeError, match="NaN"):
            run_block(nan_cat, identity_plan(4), pos, lnp, randoms, device=device)
    assert nan_cat.stretch_info()["last_discard_status"] & 1
    nan_cat.close()


@pytest.mark.parametrize("kind", ["rank", "single_process"])
def test_resident_chain_through_the_collective_path(native, kind):
    """MCD_FORCE_RCCL=1 one-rank communicators (both context kinds): the sums of every half step and the block's status
    word travel through ncclAllReduce on the catalogue's stream, as in a multi-rank job."""
    import os
    os.environ["MCD_FORCE_RCCL"] = "1"
    try:
        if kind == "rank":
            ctx1 = native.Context(rank=0, n_ranks=1, unique_id=native.Context.unique_id(), device=0)
        else:
            ctx1 = native.Context(n_devices=1)
    finally:
        del os.environ["MCD_FORCE_RCCL"]
    assert ctx1.comm_info()["size"] == 1
    rng = np.random.default_rng(9400)
    cat, sv = _catalogue(native, ctx1, rng, 20000, 1, False)
    w = 32
    pos = _walkers(rng, w, 1, sv, False)
    lnp = cat.loglike(pos)
    plan = identity_plan(4, lo=[-np.inf, 0.0, -np.inf, -np.inf])
    randoms = block_randoms(rng, 9, w, 4)
    dev = run_block(cat, plan, pos, lnp, randoms, device=True)
    host = run_block(cat, plan, pos, lnp, randoms, device=False)
    info = cat.stretch_info()
    assert same(dev, host) and info["device_blocks"] == 1 and info["discarded_blocks"] == 0, info
    plain = _catalogue(native, native.default_context(), np.random.default_rng(9400), 20000, 1, False)[0]
    assert same(dev, run_block(plain, plan, pos, lnp, randoms, device=True))
    plain.close()
    cat.close()


@pytest.fixture
def blocks_in_parts(monkeypatch):
    """Every resident block is cut into parts (what the library does from 16 MB of numbers and chain rows per block up:
    host copies and transfers of one part overlap the device work of another)."""
    monkeypatch.setenv("MCD_CHAIN_PART_BYTES", "1")


@pytest.mark.parametrize("model,free", [(1, False), (2, True)])
def test_blocks_cut_into_parts(native, ctx, blocks_in_parts, model, free):
    rng = np.random.default_rng(9600 + model)
    cat, sv = _catalogue(native, ctx, rng, 20011, model, free)
    w = 32
    pos = _walkers(rng, w, model, sv, free)
    lo, hi = np.full(cat.k, -np.inf), np.full(cat.k, np.inf)
    lo[1] = 0.0
    if model == 2:
        lo[-1], hi[-1] = 0.0, 1.0
    plan = identity_plan(cat.k, lo, hi)
    lnp = cat.loglike(pos)
    for n_steps in (1, 3, 4, 13):                                      # fewer steps than parts, uneven parts
        randoms = block_randoms(rng, n_steps, w, cat.k)
        dev = run_block(cat, plan, pos, lnp, randoms, device=True)
        assert same(dev, run_block(cat, plan, pos, lnp, randoms, device=False)), n_steps
    info = cat.stretch_info()
    assert info["device_blocks"] == 4 and info["discarded_blocks"] == 0, info
    # chain rows not asked for: the parts still join up
    randoms = block_randoms(rng, 6, w, cat.k)
    cat.set_option("device_chain", 1)
    p1, l1 = pos.copy(), lnp.copy()
    cat.stretch_move(plan, p1, l1, *randoms)
    ref = run_block(cat, plan, pos, lnp, randoms, device=False)
    assert np.array_equal(p1, ref[0]) and np.array_equal(l1, ref[1])
    cat.close()


@pytest.mark.parametrize("collective", [False, True])
def test_binned_ensembles_resident_equal_host_driven(native, collective, blocks_in_parts):
    """n_bins = B: one workgroup of the step kernel per ensemble (radial bin), one main-kernel launch of B x W/2 rows per
    half step, the guard judged on the table of all ensembles by the next launch.  Resident against host-driven, bit for
    bit; an ensemble without a single valid proposal makes the device give the block back."""
    from mcmc_dynamics_amd import DataReader
    from mcmc_dynamics_amd.analysis import BinnedConstantFit
    g = load_golden("radial_bins")
    reader = DataReader({k: g[k] for k in ("ra", "dec", "v", "verr")})
    reader.make_radial_bins(float(g["ra_center"]), float(g["dec_center"]), nstars=200, dlogr=0.05)
    context = None
    if collective:                                                       # sums of B x W/2 rows and the status word through RCCL
        import os
        os.environ["MCD_FORCE_RCCL"] = "1"
        try:
            context = native.Context(rank=0, n_ranks=1, unique_id=native.Context.unique_id(), device=0)
        finally:
            del os.environ["MCD_FORCE_RCCL"]
        assert context.comm_info()["size"] == 1
    bf = BinnedConstantFit(reader, context=context)
    bf.parameters["ra_center"].set(value=float(g["ra_center"]), fixed=True)
    bf.parameters["dec_center"].set(value=float(g["dec_center"]), fixed=True)
    B, W, P = bf.n_bins, 40, 4
    assert B > 3
    rng = np.random.default_rng(9500)
    pos = np.array([3.0, 9.0, 1.0, -1.0]) * (1.0 + 0.2 * rng.normal(size=(B, W, P)))
    pos[..., 1] = np.abs(pos[..., 1]) + 0.5
    pos = np.ascontiguousarray(pos)
    lnp = np.ascontiguousarray(bf.lnprob_batch(pos))
    assert np.all(np.isfinite(lnp))
    cat = bf._catalog
    plan = dict(bf._stretch_plan())
    plan["lo"] = plan["lo"].copy()
    plan["hi"] = plan["hi"].copy()
    plan["hi"][0] = pos[..., 0].max() - 0.3                             # cuts into the ensembles: rows get rejected
    half = W // 2

    def numbers(n):
        order = np.argsort(rng.random((n, B, W)), axis=2).astype(np.int32)
        u = rng.random((n, 4, B, half))
        zz = np.ascontiguousarray((u[:, :2] + 1.0) ** 2 / 2.0)
        thr = np.ascontiguousarray(np.log(u[:, 2:]) - (P - 1.0) * np.log(zz))
        return order, zz, thr, rng.integers(0, half, size=(n, 2, B, half)).astype(np.int32)

    def run(mode, p0, l0, r):
        cat.set_option("device_chain", mode)
        n = r[0].shape[0]
        p, l = p0.copy(), l0.copy()
        chain, lnpc, acc = np.empty((n, B, W, P)), np.empty((n, B, W)), np.zeros((B, W), dtype=np.int64)
        cat.stretch_move(plan, p, l, *r, chain, lnpc, acc)
        return p, l, chain, lnpc, acc

    ok0 = pos[..., 0] <= plan["hi"][0]
    start, start_lnp = pos.copy(), lnp.copy()
    start[~ok0] = start[ok0][0]                                          # every start position inside the prior
    start_lnp = np.ascontiguousarray(bf.lnprob_batch(start))
    r = numbers(15)
    before = cat.stretch_info()
    dev = run(1, start, start_lnp, r)
    info = cat.stretch_info()
    assert info["device_blocks"] == before["device_blocks"] + 1 and info["discarded_blocks"] == before["discarded_blocks"], info
    host = run(0, start, start_lnp, r)
    assert same(dev, host)
    acc = dev[4]
    assert acc.sum() > 0 and np.all(acc.sum(axis=1) > 0) and np.all(np.isfinite(dev[3]))
    assert np.all(dev[2][..., 0] <= plan["hi"][0])
    r2 = numbers(4)                                                      # a second block continues from the first
    assert same(run(1, dev[0], dev[1], r2), run(0, host[0], host[1], r2))
    # one ensemble entirely outside the prior: every proposal of that bin is rejected -> the device gives the block back
    # (the host loop lends such a bin a valid row of another bin)
    far = start.copy()
    far[2, :, 0] = plan["hi"][0] + 50.0
    far_lnp = start_lnp.copy()
    far_lnp[2] = -np.inf
    r3 = numbers(3)
    before = cat.stretch_info()
    dev3 = run(1, far, far_lnp, r3)
    info = cat.stretch_info()
    assert info["discarded_blocks"] == before["discarded_blocks"] + 1 and info["last_discard_status"] & 8, info
    assert same(dev3, run(0, far, far_lnp, r3)) and np.array_equal(dev3[0][2], far[2])
    bf.close()


def test_hundreds_of_ensembles(native):
    """More ensembles than the step kernel has threads (the guard's joint verdict loops over them), few walkers each."""
    from mcmc_dynamics_amd import DataReader, synthetic
    from mcmc_dynamics_amd.analysis import BinnedConstantFit
    from mcmc_dynamics_amd.analysis.binned import BinnedSampler
    cat = synthetic.make_catalog(40000, config=5)
    reader = DataReader({k: cat[k] for k in ("ra", "dec", "v", "verr")})
    reader.make_radial_bins(synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG, nstars=50, dlogr=0.002)
    bf = BinnedConstantFit(reader)
    bf.parameters["ra_center"].set(value=synthetic.CENTER_RA_DEG, fixed=True)
    bf.parameters["dec_center"].set(value=synthetic.CENTER_DEC_DEG, fixed=True)
    B = bf.n_bins
    assert B > 300, B
    pos1 = synthetic.make_walkers(16, ["v_sys", "sigma_max", "v_maxx", "v_maxy"], cat["truth"], config=5)
    pos = np.ascontiguousarray(np.broadcast_to(pos1, (B,) + pos1.shape)) * (1.0 + 1e-3 * np.random.default_rng(3).normal(size=(B, 16, 4)))
    pos[..., 1] = np.abs(pos[..., 1])
    res = []
    for mode in (1, 0):
        s = BinnedSampler(B, 16, 4, bf.lnprob_batch, seed=8, block_fn=bf._stretch_block)
        s.block_steps = 5
        bf._ensure_catalog().set_option("device_chain", mode)
        s.run_mcmc(pos, 12)
        res.append((s.chain.copy(), s.lnprobability.copy(), s.acceptance_fraction.copy()))
        s.close()
    info = bf._catalog.stretch_info()
    assert info["device_blocks"] == 3 and info["host_blocks"] == 3 and info["discarded_blocks"] == 0, info
    assert all(np.array_equal(a, b) for a, b in zip(*res)) and np.all(np.isfinite(res[0][1]))
    bf.close()


def run_seeded(cat, plan, pos, lnp, seed, step0, n_steps, device):
    cat.set_option("device_chain", int(device))
    pos, lnp = pos.copy(), lnp.copy()
    lead = pos.shape[:-1]
    chain, lnpc, acc = np.empty((n_steps,) + pos.shape), np.empty((n_steps,) + lead), np.zeros(lead, dtype=np.int64)
    cat.stretch_move_seeded(plan, pos, lnp, seed, step0, n_steps, chain, lnpc, acc)
    return pos, lnp, chain, lnpc, acc


@pytest.mark.parametrize("model,free,w", [(0, False, 48), (0, False, 512), (2, True, 64), (5, False, 130), (0, False, 8200)])
def test_seeded_blocks_generate_their_numbers_on_the_device(native, ctx, model, free, w):
    """`mcd_stretch_move_seeded`: a kernel generates the block's random numbers on the device (csrc/mcd_rng.h,
    mcd_stretch.hip: chain_numbers_kernel) -- against (a) the host-driven block of the same call (numbers generated on the
    host by the same functions), (b) `mcd_stretch_move` fed with `mcd_chain_numbers` of the same (seed, steps), resident and
    host-driven, (c) the same steps cut into two calls: every bit equal.  512 walkers: two walkers per thread in the ranking
    of the ordering keys; 130: a ragged last wave; 8200: more ordering keys than the kernel keeps in LDS -- the host build
    fills the block's numbers, uploaded as usual."""
    rng = np.random.default_rng(9700 + 10 * model + free + w)
    cat, sv = _catalogue(native, ctx, rng, 30011 if w < 1000 else 3011, model, free)
    pos = _walkers(rng, w, model, sv, free)
    lo, hi = np.full(cat.k, -np.inf), np.full(cat.k, np.inf)
    lo[1] = 0.0
    lo[0], hi[0] = pos[:, 0].min() - 0.1 * sv, pos[:, 0].max() + 0.1 * sv
    if model in (2, 4, 5):
        lo[-1], hi[-1] = 0.0, 1.0
    if model >= 3:
        lo[2] = lo[5] = 1.0
    plan = identity_plan(cat.k, lo, hi)
    lnp = cat.loglike(pos)
    seed, step0, n = 0xC0FFEE1234567 + model, 1000, 17 if w < 1000 else 7
    before = cat.stretch_info()
    dev = run_seeded(cat, plan, pos, lnp, seed, step0, n, device=1)
    mid = cat.stretch_info()
    assert mid["device_blocks"] == before["device_blocks"] + 1 and mid["discarded_blocks"] == before["discarded_blocks"], mid
    host = run_seeded(cat, plan, pos, lnp, seed, step0, n, device=0)
    after = cat.stretch_info()
    assert after["host_blocks"] == mid["host_blocks"] + 1
    assert same(dev, host)
    numbers = native.chain_numbers(seed, step0, n, 1, w, cat.k, squeeze=True)
    numbers = tuple(np.ascontiguousarray(a) for a in numbers)
    assert same(dev, run_block(cat, plan, pos, lnp, numbers, device=True))
    assert same(dev, run_block(cat, plan, pos, lnp, numbers, device=False))
    first = run_seeded(cat, plan, pos, lnp, seed, step0, 6, device=1)
    second = run_seeded(cat, plan, first[0], first[1], seed, step0 + 6, n - 6, device=1)
    assert np.array_equal(np.concatenate([first[2], second[2]]), dev[2]) and np.array_equal(second[1], dev[1])
    assert np.array_equal(first[4] + second[4], dev[4])
    assert 0 < dev[4].sum() < n * w and np.all(np.isfinite(dev[3]))
    other = run_seeded(cat, plan, pos, lnp, seed + 1, step0, n, device=1)
    assert not np.array_equal(other[2], dev[2])
    # the step kernel for ensembles of any size (option 2) reads the same generated numbers
    before = cat.stretch_info()
    assert same(dev, run_seeded(cat, plan, pos, lnp, seed, step0, n, device=2))
    assert cat.stretch_info()["device_blocks"] == before["device_blocks"] + 1
    cat.close()


@pytest.mark.parametrize("parts", [False, True])
def test_seeded_binned_blocks_and_the_sampler_modes(native, parts, monkeypatch):
    """B ensembles with device-generated numbers: resident == host-driven == the NumPy loop of BinnedSampler fed by
    `chain_numbers` (rng="device" without a block function), also with every block cut into parts."""
    from mcmc_dynamics_amd import DataReader
    from mcmc_dynamics_amd.analysis import BinnedConstantFit
    from mcmc_dynamics_amd.analysis.binned import BinnedSampler
    if parts:
        monkeypatch.setenv("MCD_CHAIN_PART_BYTES", "1")
    g = load_golden("radial_bins")
    reader = DataReader({k: g[k] for k in ("ra", "dec", "v", "verr")})
    reader.make_radial_bins(float(g["ra_center"]), float(g["dec_center"]), nstars=200, dlogr=0.05)
    bf = BinnedConstantFit(reader)
    bf.parameters["ra_center"].set(value=float(g["ra_center"]), fixed=True)
    bf.parameters["dec_center"].set(value=float(g["dec_center"]), fixed=True)
    B, W, P = bf.n_bins, 40, 4
    rng = np.random.default_rng(9800)
    pos = np.array([3.0, 9.0, 1.0, -1.0]) * (1.0 + 0.2 * rng.normal(size=(B, W, P)))
    pos[..., 1] = np.abs(pos[..., 1]) + 0.5
    pos = np.ascontiguousarray(pos)
    res = []
    for mode, seeded_fn, block_steps in ((1, bf._stretch_block_seeded, 256), (0, bf._stretch_block_seeded, 9), (1, None, 11)):
        s = BinnedSampler(B, W, P, bf.lnprob_batch, seed=77, rng="device", seeded_block_fn=seeded_fn)
        s.device_block_steps = block_steps
        bf._ensure_catalog().set_option("device_chain", mode)
        s.run_mcmc(pos, 23)
        res.append((s.chain.copy(), s.lnprobability.copy(), s.acceptance_fraction.copy()))
        s.close()
    info = bf._catalog.stretch_info()
    assert info["device_blocks"] == 1 and info["host_blocks"] == 3 and info["discarded_blocks"] == 0, info
    assert same(res[0], res[1]) and same(res[0], res[2])
    assert np.all(np.isfinite(res[0][1])) and np.all(res[0][2] > 0)
    # the class's own entry point uses the device generator by default
    np.random.seed(4)
    s = bf(n_walkers=W, n_steps=6, pos=pos)
    assert s.rng == "device" and s.chain.shape == (B, W, 6, P)
    s.close()
    bf.close()


def test_seeded_block_edge_cases(native, ctx):
    """No steps, no outputs, only one of the outputs, accepted counts that continue, bad arguments."""
    rng = np.random.default_rng(9900)
    cat, sv = _catalogue(native, ctx, rng, 4001, 0, False)
    w = 24
    pos = _walkers(rng, w, 0, sv, False)
    lnp = cat.loglike(pos)
    plan = identity_plan(4, lo=[-np.inf, 0.0, -np.inf, -np.inf])
    full = run_seeded(cat, plan, pos, lnp, 5, 0, 9, device=1)
    # zero steps: nothing changes
    p, l = pos.copy(), lnp.copy()
    cat.stretch_move_seeded(plan, p, l, 5, 0, 0)
    assert np.array_equal(p, pos) and np.array_equal(l, lnp)
    for mode in (1, 0):
        cat.set_option("device_chain", mode)
        # no outputs at all; only the log-probabilities; accepted counts continue from what the caller passes
        p, l = pos.copy(), lnp.copy()
        cat.stretch_move_seeded(plan, p, l, 5, 0, 9)
        assert np.array_equal(p, full[0]) and np.array_equal(l, full[1])
        p, l, lnpc = pos.copy(), lnp.copy(), np.empty((9, w))
        cat.stretch_move_seeded(plan, p, l, 5, 0, 9, None, lnpc)
        assert np.array_equal(lnpc, full[3])
        p, l, acc = pos.copy(), lnp.copy(), np.full(w, 100, dtype=np.int64)
        cat.stretch_move_seeded(plan, p, l, 5, 0, 9, None, None, acc)
        assert np.array_equal(acc, full[4] + 100)
    # a seed is 64 bits; negative Python integers wrap
    a = run_seeded(cat, plan, pos, lnp, -1, 0, 3, device=1)
    b = run_seeded(cat, plan, pos, lnp, 2 ** 64 - 1, 0, 3, device=1)
    assert same(a, b) and not np.array_equal(a[2], full[2][:3])
    with pytest.raises(ValueError):
        cat.stretch_move_seeded(plan, pos.copy(), lnp.copy(), 5, -1, 3)
    with pytest.raises(ValueError):
        cat.stretch_move_seeded(plan, pos.copy(), lnp[:-1].copy(), 5, 0, 3)
    with pytest.raises(native.NativeError):
        cat.stretch_move_seeded(plan, pos[:-1].copy(), lnp[:-1].copy(), 5, 0, 3)      # odd number of walkers
    cat.close()
