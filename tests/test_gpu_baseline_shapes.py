"""GPU: every BASELINE.json configuration at its EXACT shape, in the driver-run suite (the parity cases elsewhere use
smaller catalogues):

  C2  1e5 synthetic stars x 256 walkers, rotation+dispersion, f64 -- all 256 walkers against the oracle
  C3  1e6 x 256 with the fixed-Gaussian-background mixture -- tests/test_gpu_kernels.py::test_full_size_c3_properties
  C4  1e7 stars in 8 shards on one GPU -- tests/test_gpu_kernels.py::test_c4_size_shards
  C5  1e6 stars, make_radial_bins(nstars=1000, dlogr=0.05) (utils/files/data_reader.py:71-140; bin/run_tests.py:75-124),
      512 walkers per bin, f64 / f32 terms + f64 accumulation / f32

plus the combination the XCD-grouped launch path serves: a binned catalogue with more than 256 walkers and several
parameter sets, for the mixture models as well."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

RTOL = 1e-12
NAMES4 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]


@pytest.fixture(scope="module")
def native():
    from mcmc_dynamics_amd import _native
    return _native


@pytest.fixture(scope="module")
def ctx(native):
    return native.default_context()


def _catalog(n, config, background=False, min_sep_arcmin=1e-2):
    """Synthetic catalogue of SURVEY 8(d); stars within `min_sep_arcmin` of the centre are moved (theta is ill-conditioned
    there in the reference's own formula, see tests/test_gpu_kernels.py::_synthetic)."""
    from mcmc_dynamics_amd import synthetic
    from oracle import lnprob_numpy as oracle
    c = synthetic.make_catalog(n, config=config, background=background)
    centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
    if min_sep_arcmin:
        dx, dy = oracle.calc_xy_offset(c["ra"], c["dec"], *centre)
        r = np.hypot(dx, dy)
        near = r < min_sep_arcmin
        if near.any():
            donor = int(np.argmax(r))
            c["ra"][near], c["dec"][near] = c["ra"][donor], c["dec"][donor]
    return c, centre


def _oracle_blocks(cat, pos, centre, block=32, **kw):
    """oracle.batched_constant_lnlike over blocks of walkers (bounds the (W, N) temporaries at 1e5+ stars)."""
    from oracle import lnprob_numpy as oracle
    return np.concatenate([oracle.batched_constant_lnlike(cat, pos[i:i + block], *centre, **kw)
                           for i in range(0, len(pos), block)])


def test_c2_exact_shape_against_the_oracle(native, ctx):
    """C2: 1e5 stars x 256 walkers, ConstantFit, fixed centre, float64 -- every walker against the NumPy oracle."""
    from mcmc_dynamics_amd import synthetic
    c, centre = _catalog(100000, 2)
    pos = synthetic.make_walkers(256, NAMES4, c["truth"], config=2)
    cat = native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=native.MODEL_CONST, centre=centre)
    got = cat.loglike(pos)
    assert got.shape == (256,) and cat.fast_level == 1
    want = _oracle_blocks(c, pos, centre)
    assert rel_err(got, want) < RTOL
    assert np.array_equal(cat.loglike(pos), got)                       # bitwise repeatable
    cat.set_option("fast_path", 0)
    assert rel_err(cat.loglike(pos), want) < RTOL                      # the plain kernels at the same shape
    # the walker-halves an emcee stretch move evaluates (128 proposals per call) give the same numbers
    cat.set_option("fast_path", 1)
    assert rel_err(np.concatenate([cat.loglike(pos[:128]), cat.loglike(pos[128:])]), want) < RTOL


@pytest.fixture(scope="module")
def c5(native, ctx):
    """C5 catalogue: 1e6 stars sorted into the reference's radial bins, 512 walkers."""
    from mcmc_dynamics_amd import DataReader, synthetic
    c, centre = _catalog(1000000, 5, min_sep_arcmin=0)
    reader = DataReader({k: c[k] for k in ("ra", "dec", "v", "verr")})
    reader.make_radial_bins(centre[0], centre[1], nstars=1000, dlogr=0.05)
    srt, offs = reader.sorted_by_bin()
    cols = {k: np.asarray(srt.data[k]) for k in ("ra", "dec", "v", "verr")}
    pos = synthetic.make_walkers(512, NAMES4, c["truth"], config=5)
    return cols, np.asarray(offs, dtype=np.int64), pos, centre


def test_c5_exact_shape_f64(native, ctx, c5):
    cols, offs, pos, centre = c5
    n_bins = len(offs) - 1
    assert offs[-1] == 1000000 and n_bins >= 20 and np.all(np.diff(offs) >= 1000)
    params = np.ascontiguousarray(np.broadcast_to(pos, (n_bins,) + pos.shape))
    binned = native.Catalog(ctx, cols["ra"], cols["dec"], cols["v"], cols["verr"], model=native.MODEL_CONST, centre=centre,
                            bin_offsets=offs)
    got = binned.loglike(params)
    assert got.shape == (n_bins, 512) and np.all(np.isfinite(got))
    assert np.array_equal(binned.loglike(params), got)                 # bitwise repeatable
    # the same walkers in every bin: the sum over the bins is the un-binned log-likelihood (SURVEY 8(c) known answer 4)
    flat = native.Catalog(ctx, cols["ra"], cols["dec"], cols["v"], cols["verr"], model=native.MODEL_CONST, centre=centre)
    assert rel_err(got.sum(axis=0), flat.loglike(pos)) < RTOL
    # two bins against the oracle on all 512 walkers (a middle bin and the last, largest one)
    for b in (n_bins // 2, n_bins - 1):
        sub = {k: v[offs[b]:offs[b + 1]] for k, v in cols.items()}
        assert rel_err(got[b], _oracle_blocks(sub, pos, centre, block=16 if len(sub["v"]) > 200000 else 64)) < RTOL, b
    # per-bin ensembles that differ from bin to bin: every bin still equals a stand-alone catalogue of its stars
    rng = np.random.default_rng(5)
    varied = params * (1.0 + 0.01 * rng.normal(size=(n_bins, 1, 4)))
    varied[..., 1] = np.abs(varied[..., 1])
    got_v = binned.loglike(varied)
    for b in (0, 3, n_bins - 2):
        sl = slice(offs[b], offs[b + 1])
        one = native.Catalog(ctx, cols["ra"][sl], cols["dec"][sl], cols["v"][sl], cols["verr"][sl], model=native.MODEL_CONST,
                             centre=centre)
        assert rel_err(got_v[b], one.loglike(varied[b])) < RTOL, b
        one.close()
    binned.close()
    flat.close()


def test_c5_exact_shape_seeded_chain(native, ctx, c5):
    """C5 at its full size through the sampler's hot path: 55 ensembles of 512 walkers in lockstep, the move's random
    numbers generated on the device (`mcd_stretch_move_seeded`).  Size-independent properties: the resident block equals the
    host-driven block of the same seed bit for bit; every bin's chain equals the chain of a stand-alone catalogue of that
    bin's stars run with the numbers `mcd_chain_numbers` gives for that bin (the bins are independent ensembles)."""
    cols, offs, pos, centre = c5
    B, W, P = len(offs) - 1, 512, 4
    binned = native.Catalog(ctx, cols["ra"], cols["dec"], cols["v"], cols["verr"], model=native.MODEL_CONST, centre=centre,
                            bin_offsets=offs)
    rng = np.random.default_rng(55)
    start = np.ascontiguousarray(np.broadcast_to(pos, (B, W, P)) * (1.0 + 0.01 * rng.normal(size=(B, W, P))))
    start[..., 1] = np.abs(start[..., 1])
    lnp0 = np.ascontiguousarray(binned.loglike(start))
    plan = {"col_source": np.arange(P, dtype=np.int32), "col_const": np.zeros(P), "col_factor": np.ones(P),
            "lo": np.array([-np.inf, 0.0, -np.inf, -np.inf]), "hi": np.full(P, np.inf), "fixed_ok": True}
    seed, step0, n = 2026, 40, 6
    runs = {}
    for mode in (1, 0):
        binned.set_option("device_chain", mode)
        p, l = start.copy(), lnp0.copy()
        chain, lnpc, acc = np.empty((n, B, W, P)), np.empty((n, B, W)), np.zeros((B, W), dtype=np.int64)
        binned.stretch_move_seeded(plan, p, l, seed, step0, n, chain, lnpc, acc)
        runs[mode] = (p, l, chain, lnpc, acc)
    info = binned.stretch_info()
    assert info["device_blocks"] == 1 and info["host_blocks"] == 1 and info["discarded_blocks"] == 0, info
    assert all(np.array_equal(a, b) for a, b in zip(runs[1], runs[0]))
    acc = runs[1][4]
    assert np.all(np.isfinite(runs[1][3])) and np.all(acc.sum(axis=1) > 0) and acc.sum() < n * B * W
    order, zz, thr, pick = native.chain_numbers(seed, step0, n, B, W, P)
    for b in (0, B // 2, B - 1):
        sl = slice(offs[b], offs[b + 1])
        one = native.Catalog(ctx, cols["ra"][sl], cols["dec"][sl], cols["v"][sl], cols["verr"][sl], model=native.MODEL_CONST,
                             centre=centre)
        p, l = start[b].copy(), np.ascontiguousarray(one.loglike(start[b]))
        chain1, lnpc1 = np.empty((n, W, P)), np.empty((n, W))
        one.stretch_move(plan, p, l, np.ascontiguousarray(order[:, b]), np.ascontiguousarray(zz[:, :, b]),
                         np.ascontiguousarray(thr[:, :, b]), np.ascontiguousarray(pick[:, :, b]), chain1, lnpc1)
        # (a stand-alone catalogue has its own chunk table: the sums agree to rounding, so accept decisions may differ where
        # a threshold is met within 1e-12 -- none does at this size; positions are then the same bits)
        assert rel_err(lnpc1, runs[1][3][:, b]) < RTOL and np.array_equal(chain1, runs[1][2][:, b]), b
        one.close()
    binned.close()


def test_c5_exact_shape_precision_sweep(native, ctx, c5):
    """float32 vs float64 at the full C5 shape: f32 terms + f64 accumulation <= 1e-6, pure f32 <= 2e-5 relative per
    (bin, walker) output (the tolerances of test_precision_sweep_c5; SURVEY appendix: 9.5e-9 / 1.8e-7 at 1e5 stars)."""
    cols, offs, pos, centre = c5
    n_bins = len(offs) - 1
    params = np.ascontiguousarray(np.broadcast_to(pos, (n_bins,) + pos.shape))
    out = {}
    for prec in ("f64", "f32acc64", "f32"):
        cat = native.Catalog(ctx, cols["ra"], cols["dec"], cols["v"], cols["verr"], model=native.MODEL_CONST, centre=centre,
                             bin_offsets=offs, precision=prec)
        out[prec] = cat.loglike(params)
        cat.close()
    assert rel_err(out["f32acc64"], out["f64"]) < 1e-6
    assert rel_err(out["f32"], out["f64"]) < 2e-5
    # and on the un-binned sum (1e6 terms per walker)
    assert rel_err(out["f32acc64"].sum(axis=0), out["f64"].sum(axis=0)) < 1e-6
    assert rel_err(out["f32"].sum(axis=0), out["f64"].sum(axis=0)) < 2e-5


@pytest.mark.parametrize("n_walkers", [320, 512])
@pytest.mark.parametrize("model", ["const", "bgfixed", "bggauss"])
def test_binned_catalogue_with_more_than_256_walkers(native, ctx, model, n_walkers):
    """Radial bins x W > 256: several workgroups per chunk (XCD-grouped launch) with several parameter sets, idle lanes in
    the last walker tile (W = 320), mixture models included.  Every bin against the oracle, per-bin walker ensembles."""
    from mcmc_dynamics_amd import synthetic
    from oracle import lnprob_numpy as oracle
    c, centre = _catalog(60000, 3, background=True)
    dx, dy = oracle.calc_xy_offset(c["ra"], c["dec"], *centre)
    bins = oracle.make_radial_bins(np.hypot(dx, dy), 4000, 0.1).astype(np.int64)
    order = np.argsort(bins, kind="stable")
    offs = np.concatenate([[0], np.cumsum(np.bincount(bins))]).astype(np.int64)
    n_bins = len(offs) - 1
    assert n_bins >= 5
    srt = {k: (v[order] if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    names = NAMES4 + (["v_back", "sigma_back", "f_back"] if model == "bggauss" else [])
    base = synthetic.make_walkers(n_walkers, names, c["truth"], config=3)
    rng = np.random.default_rng(n_walkers)
    params = np.stack([base * (1.0 + 0.01 * rng.normal(size=base.shape)) for _ in range(n_bins)])
    params[..., 1] = np.abs(params[..., 1])
    if model == "bggauss":
        params[..., 5] = np.abs(params[..., 5])
        params[..., 6] = np.clip(params[..., 6], 0.01, 0.99)
    lnbg = oracle.gaussian_background(srt["v"], srt["verr"], synthetic.TRUTH["v_back"], synthetic.TRUTH["sigma_back"])
    if model == "const":
        kw, mid = {}, native.MODEL_CONST
    elif model == "bgfixed":
        kw, mid = dict(lnlike_bg=lnbg, pmember=srt["pmember"]), native.MODEL_CONST_BGFIXED
    else:
        kw, mid = dict(density=srt["density"]), native.MODEL_CONST_BGGAUSS
    cat = native.Catalog(ctx, srt["ra"], srt["dec"], srt["v"], srt["verr"], model=mid, centre=centre, bin_offsets=offs, **kw)
    for fast in (1, 0):
        cat.set_option("fast_path", fast)
        got = cat.loglike(params)
        assert got.shape == (n_bins, n_walkers)
        for b in range(n_bins):
            sl = slice(offs[b], offs[b + 1])
            sub = {k: v[sl] for k, v in srt.items() if isinstance(v, np.ndarray)}
            if model == "const":
                want = oracle.batched_constant_lnlike(sub, params[b], *centre)
            elif model == "bgfixed":
                want = oracle.batched_constant_lnlike(sub, params[b], *centre, lnlike_background=lnbg[sl], pmember=sub["pmember"])
            else:
                want = oracle.batched_constant_gb_lnlike(sub, params[b], *centre)
            assert rel_err(got[b], want) < RTOL, (fast, b)
    cat.close()


def test_c3_exact_shape_precision_sweep(native, ctx):
    """The C5-style precision sweep on C3 (1e6 stars x 256 walkers, fixed-Gaussian-background mixture): the float32 fast
    mixtures (v_rsq_f32 / v_exp_f32, products in float or double) against float64.  Tolerances per walker log-likelihood:
    f32 terms + f64 accumulation <= 1e-6, pure f32 <= 2e-5 relative (as for C5)."""
    from mcmc_dynamics_amd import synthetic
    from mcmc_dynamics_amd.background import Gaussian
    c, centre = _catalog(1000000, 3, background=True)
    lnbg = Gaussian(synthetic.TRUTH["v_back"], synthetic.TRUTH["sigma_back"])(c["v"], c["verr"])
    pos = synthetic.make_walkers(256, NAMES4, c["truth"], config=3)
    out, level = {}, {}
    for prec in ("f64", "f32acc64", "f32"):
        cat = native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=native.MODEL_CONST_BGFIXED, centre=centre,
                             lnlike_bg=lnbg, pmember=c["pmember"], precision=prec)
        out[prec] = cat.loglike(pos)
        level[prec] = cat.fast_level
        if prec != "f64":
            cat.set_option("fast_path", 0)
            out[prec + " plain"] = cat.loglike(pos)
        cat.close()
    assert level == {"f64": 2, "f32acc64": 1, "f32": 1}
    assert rel_err(out["f32acc64"], out["f64"]) < 1e-6
    assert rel_err(out["f32"], out["f64"]) < 2e-5
    assert rel_err(out["f32acc64 plain"], out["f64"]) < 1e-6 and rel_err(out["f32 plain"], out["f64"]) < 2e-5


@pytest.mark.parametrize("model", ["bgfixed", "bggauss", "profile_bgdens", "profile_bggauss", "profile_bgfixed"])
@pytest.mark.parametrize("free", [False, True])
def test_float32_fast_mixtures_against_float64(native, ctx, model, free):
    """Every mixture model, fixed and free centre: float32 fast formulation (guard level 1) vs the float64 result, and the
    fall-back to the plain float32 kernels when a star lies outside the float32 ranges (a certain member)."""
    from mcmc_dynamics_amd import synthetic
    from mcmc_dynamics_amd.background import Gaussian
    c, centre = _catalog(50000, 3, background=True)
    lnbg = Gaussian(synthetic.TRUTH["v_back"], synthetic.TRUTH["sigma_back"])(c["v"], c["verr"])
    names = list(NAMES4)
    rng = np.random.default_rng(3)
    base = synthetic.make_walkers(96, NAMES4 + ["v_back", "sigma_back", "f_back"], c["truth"], config=3)
    core = base[:, :4]
    prof = model.startswith("profile")
    if prof:
        a = 30.0 * (1.0 + 0.05 * rng.normal(size=96))
        rp = 60.0 * (1.0 + 0.05 * rng.normal(size=96))
        core = np.column_stack([base[:, 0], base[:, 1], a, base[:, 2], base[:, 3], rp])
    cols = [core]
    if free:
        cols.append(np.column_stack([centre[0] + 1e-3 * rng.normal(size=96), centre[1] + 1e-3 * rng.normal(size=96)]))
    kw = {}
    if model in ("bgfixed", "profile_bgfixed"):
        kw = dict(lnlike_bg=lnbg, pmember=c["pmember"])
    elif model in ("bggauss", "profile_bggauss"):
        kw = dict(density=c["density"])
        cols.append(np.column_stack([base[:, 4], np.abs(base[:, 5]), np.clip(base[:, 6], 0.01, 0.99)]))
    else:
        kw = dict(lnlike_bg=lnbg, density=c["density"])
        cols.append(np.clip(base[:, 6], 0.01, 0.99)[:, None])
    params = np.ascontiguousarray(np.column_stack(cols))
    mid = {"bgfixed": native.MODEL_CONST_BGFIXED, "bggauss": native.MODEL_CONST_BGGAUSS,
           "profile_bgdens": native.MODEL_PROFILE_BGDENS, "profile_bggauss": native.MODEL_PROFILE_BGGAUSS,
           "profile_bgfixed": native.MODEL_PROFILE_BGFIXED}[model]
    res = {}
    for prec in ("f64", "f32acc64", "f32"):
        cat = native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=mid, centre=None if free else centre,
                             precision=prec, **kw)
        if prec != "f64" and free:
            # free centre: the tangent-plane offsets are differences of O(1) float32 products (position-angle error 2^-23 /
            # separation), and this compact catalogue (stars within 5') with its rotation is OUTSIDE the float32 accuracy
            # domain (round 3, include/mcd.h: kappa_theta): refused; evaluated regardless it is what round 2 measured
            with pytest.raises(native.NativeError, match="float32 accuracy domain.*tangent-plane"):
                cat.loglike(params)
            cat.set_option("f32_domain", 0)
        res[prec] = cat.loglike(params)
        if prec != "f64":
            assert cat.fast_level == 1, (model, prec)
            assert cat.f32_in_domain == (not free) and cat.f32_condition[0] <= 96.0
        cat.close()
    assert rel_err(res["f32acc64"], res["f64"]) < (2e-5 if free else 1e-6), model
    assert rel_err(res["f32"], res["f64"]) < (1e-4 if free else 2e-5), model
    if model == "bgfixed" and not free:
        pm = c["pmember"].copy()
        pm[7] = 1.0                                                     # a certain member: outside the float32 fast ranges
        cat = native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=mid, centre=centre, precision="f32acc64",
                             lnlike_bg=lnbg, pmember=pm)
        # ... which is outside the float32 accuracy domain (round 3, include/mcd.h): refused with the reason, unless the
        # caller switches the enforcement off -- then the plain float32 kernels evaluate it and the verdict is on record
        with pytest.raises(native.NativeError, match="float32 accuracy domain.*pmember <= 1 - 2\\^-20"):
            cat.loglike(params)
        cat.set_option("f32_domain", 0)
        got = cat.loglike(params)
        assert cat.fast_level == 0 and not cat.f32_in_domain
        ref = native.Catalog(ctx, c["ra"], c["dec"], c["v"], c["verr"], model=mid, centre=centre, lnlike_bg=lnbg, pmember=pm)
        assert rel_err(got, ref.loglike(params)) < 2e-6
