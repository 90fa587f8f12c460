"""CPU check of the kernels' arithmetic: csrc/mcd_math.h compiled for the host (tests/emul) must
reproduce the golden vectors of the reference for every model, in the plain AND the fast formulation.
This pins the algebra (fraction tree, log-product, rsqrt + single-exp mixtures) before any GPU time is
spent; the GPU tests then only have to show that the device executes the same expressions."""
import numpy as np
import pytest

import emul_helper as emul
from conftest import load_golden, rel_err

RTOL = 1e-12
CASES = [
    ("constant_fixed", 0, False), ("constant_free", 0, True),
    ("constant_bg_gaussian_fixed", 1, False), ("constant_bg_gaussian_free", 1, True),
    ("constant_gb_fixed", 2, False), ("constant_gb_free", 2, True),
    ("model_fit_fixed", 3, False), ("model_fit_free", 3, True),                 # ModelFit (analysis/model.py)
    ("model_fit_gb_fixed", 4, False), ("model_fit_gb_free", 4, True),           # ModelFitGB
    ("model_fit_cb_fixed", 5, False), ("model_fit_cb_free", 5, True),           # ModelFitConstantBackground
    ("model_fit_bg_gaussian_fixed", 6, False), ("model_fit_bg_gaussian_free", 6, True),   # ModelFit(background=Gaussian)
]


@pytest.mark.parametrize("name,model,free", CASES)
@pytest.mark.parametrize("fast", [0, 1])
@pytest.mark.parametrize("chunk_len", [64, 248, 100000])
def test_host_compiled_kernel_math_matches_reference(name, model, free, fast, chunk_len):
    g = load_golden(name)
    cat = {k: g[k] for k in ("ra", "dec", "v", "verr")}
    if model in (1, 6):
        cat["lnlike_bg"], cat["pmember"] = g["lnlike_background"], g["pmember"]
    if model in (2, 4, 5):
        cat["density"] = g["density"]
    if model == 5:
        cat["lnlike_bg"] = g["lnlike_background"]
    centre = None if free else (float(g["ra_center"]), float(g["dec_center"]))
    ok = np.isfinite(g["lnprob"])
    values = emul.abi_columns(g["names"], g["values"], model, free)
    got = emul.loglike(cat, values[ok], model, centre, fast, chunk_len)
    assert rel_err(got, g["lnprob"][ok]) < RTOL


def test_per_star_outputs():
    """membership probabilities (constant.py:366-374) and lnlike(no_sum=True) (model.py:565-623)."""
    g = load_golden("constant_gb_fixed")
    cat = {k: g[k] for k in ("ra", "dec", "v", "verr", "density")}
    centre = (float(g["ra_center"]), float(g["dec_center"]))
    row = emul.abi_columns(g["names"], g["values"], 2, False)[int(g["membership_row"])]
    assert np.max(np.abs(emul.per_star(cat, row, 2, centre, 0) - g["membership"])) < 1e-12
    for which, centre_of in (("fixed", lambda g: (float(g["ra_center"]), float(g["dec_center"]))), ("free", lambda g: None)):
        g = load_golden("model_fit_cb_" + which)
        cat = {k: g[k] for k in ("ra", "dec", "v", "verr", "density")}
        cat["lnlike_bg"] = g["lnlike_background"]
        row = emul.abi_columns(g["names"], g["values"], 5, which == "free")[int(g["no_sum_row"])]
        got = emul.per_star(cat, row, 5, centre_of(g), 1)
        assert np.max(np.abs(got - g["lnlike_no_sum"]) / np.abs(g["lnlike_no_sum"])) < 1e-12
        mem = emul.per_star(cat, row, 5, centre_of(g), 0)
        assert np.all((mem >= 0) & (mem <= 1))


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_model_fit_gb_membership_golden(which):
    """ModelFitGB.calculate_membership_probabilities (model.py:458-510): the per-star kernel arithmetic at the chain's
    median row against the reference's output."""
    g = load_golden("model_fit_gb_membership_" + which)
    cat = {k: g[k] for k in ("ra", "dec", "v", "verr", "density")}
    centre = None if which == "free" else (float(g["ra_center"]), float(g["dec_center"]))
    row = emul.abi_columns(g["names"], g["median"][None, :], 4, which == "free")[0]
    got = emul.per_star(cat, row, 4, centre, 0)
    assert np.max(np.abs(got - g["membership"])) < 1e-12


def test_fast_paths_survive_outliers_and_extreme_backgrounds():
    """Stars hundreds of sigma away (the example catalogue has -928 / +676 km/s) and a background that is
    1e-300 times less likely than the cluster must not overflow/underflow the single-exp formulation."""
    g = load_golden("constant_bg_gaussian_fixed")
    cat = {k: g[k].copy() for k in ("ra", "dec", "v", "verr", "pmember")}
    cat["v"][:4] = [-928.0, 676.0, 3000.0, -5000.0]
    cat["lnlike_bg"] = g["lnlike_background"].copy()
    cat["lnlike_bg"][4:8] = [-700.0, -1500.0, -30000.0, -5.0]
    centre = (float(g["ra_center"]), float(g["dec_center"]))
    ok = np.isfinite(g["lnprior"])
    plain = emul.loglike(cat, g["values"][ok], 1, centre, 0)
    fast = emul.loglike(cat, g["values"][ok], 1, centre, 1)
    assert np.all(np.isfinite(plain))
    assert rel_err(fast, plain) < RTOL


def test_fast_mixtures_share_the_reference_underflow_behaviour():
    """pmember == 1 with a 90-sigma outlier, or f_back == 0 with a star the cluster model rejects: the
    reference's log-sum-exp about max(m, b) underflows to log(0) = -inf (runner.py:282-284,
    constant.py:320-323).  The single-exp formulation must give -inf for the same walkers and agree on the rest."""
    g = load_golden("constant_bg_gaussian_fixed")
    cat = {k: g[k].copy() for k in ("ra", "dec", "v", "verr", "pmember")}
    cat["lnlike_bg"] = g["lnlike_background"].copy()
    centre = (float(g["ra_center"]), float(g["dec_center"]))
    ok = np.isfinite(g["lnprior"]) & (g["values"][:, 1] > 0)
    cat["pmember"][:3] = 1.0
    cat["pmember"][3] = 0.0
    cat["v"][:3] = [60.0, -45.0, 80.0]                   # certain members, moderate outliers: finite
    plain = emul.loglike(cat, g["values"][ok], 1, centre, 0)
    assert np.all(np.isfinite(plain))
    assert rel_err(emul.loglike(cat, g["values"][ok], 1, centre, 1), plain) < RTOL
    cat["v"][:3] = [900.0, -1500.0, 4000.0]              # certain members, > 38 sigma: exp underflows in the reference
    plain = emul.loglike(cat, g["values"][ok], 1, centre, 0)
    assert np.all(np.isneginf(plain))
    assert np.all(np.isneginf(emul.loglike(cat, g["values"][ok], 1, centre, 1)))

    g = load_golden("constant_gb_fixed")
    cat = {k: g[k].copy() for k in ("ra", "dec", "v", "verr", "density")}
    cat["v"][:3] = [900.0, -1500.0, 4000.0]
    vals = g["values"][np.isfinite(g["lnprior"]) & (g["values"][:, 1] > 0) & (g["values"][:, 5] > 0)].copy()
    vals[:4, 6] = 0.0                                    # f_back = 0 with extreme outliers: -inf in the reference
    vals[4:8, 5] = 3.0                                   # narrow background: the damped term underflows harmlessly
    plain = emul.loglike(cat, vals, 2, centre, 0)
    assert np.all(np.isneginf(plain[:4])) and np.isfinite(plain[4:]).sum() >= 8
    assert rel_err(emul.loglike(cat, vals, 2, centre, 1), plain) < RTOL


def test_kde_lane_math_matches_reference_golden():
    """The KDE kernel's per-lane arithmetic (csrc/mcd_math.h: KdeLane, two passes per slice, polynomial exp) and its
    slice combination, compiled for the host, against background.SingleStars of the reference (single_stars.py:42-77)
    for one slice and for ragged multi-slice splits."""
    import ctypes
    from conftest import load_golden
    g = load_golden("single_stars")
    lib = emul.lib()
    lib.emul_kde.argtypes = [ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                             ctypes.c_double, ctypes.c_int64, ctypes.c_void_p]
    comp, v, verr = (np.ascontiguousarray(g[k], dtype=np.float64) for k in ("comp", "v", "verr"))
    for tag in ("s0", "s2"):
        want = g["lnlike_" + tag]
        for slice_len in (comp.size, 64, 8, 5):
            out = np.empty(v.size)
            assert lib.emul_kde(comp.size, comp.ctypes.data, v.size, v.ctypes.data, verr.ctypes.data,
                                float(g["sigma_int_" + tag]), slice_len, out.ctypes.data) == 0
            assert np.isfinite(out).all()
            assert np.max(np.abs(out - want) / np.maximum(1.0, np.abs(want))) < 1e-13


def test_vanishing_prior_with_extreme_background_takes_the_plain_path():
    """pmember == 0 (or 1e-300) on a star whose background likelihood is e^-2000: the reference's log-sum-exp about the
    cluster exponent underflows (runner.py:282-284: -inf for p == 0); the exponent-carrying fast path would return the
    mathematically exact value instead, so the range guard must send such catalogues to the plain kernels."""
    g = load_golden("constant_bg_gaussian_fixed")
    centre = (float(g["ra_center"]), float(g["dec_center"]))
    rows = g["values"][np.isfinite(g["lnprior"])]
    cat = {k: g[k].copy() for k in ("ra", "dec", "v", "verr", "pmember")}
    cat["lnlike_bg"] = g["lnlike_background"].copy()
    cat["lnlike_bg"][10] = -2000.0
    for p, admitted in ((0.0, False), (1e-300, False), (1e-30, True), (0.5, True), (1.0, True)):
        cat["pmember"][10] = p
        assert emul.fast_guard(cat, rows, 1, centre) == admitted
        if admitted:
            assert rel_err(emul.loglike(cat, rows, 1, centre, 1), emul.loglike(cat, rows, 1, centre, 0)) < RTOL
    cat["pmember"][10] = 0.0
    cat["lnlike_bg"][10] = -600.0                       # no exponent carry above -690: any prior is fine
    assert emul.fast_guard(cat, rows, 1, centre)
    assert rel_err(emul.loglike(cat, rows, 1, centre, 1), emul.loglike(cat, rows, 1, centre, 0)) < RTOL


@pytest.mark.parametrize("seed,model", [(1275109, 2), (1283548, 2)])
def test_underflow_window_between_the_two_formulations(seed, model):
    """Found by tools/fuzz_gpu.py: a star with density == 0 whose background exponent lies ~745 below the cluster's.
    The reference keeps the prefactors inside the exponent (exp(lb - lc) is still a denormal: finite result), the
    single-exp formulation applies them outside (f g_b * exp(-delta/2) with the exp already flushed to 0).  An exact
    zero mixture value therefore has to trigger the re-evaluation with the plain kernels, like a denormal one."""
    from test_guard_random_cpu import CENTRE, random_case
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 1500))
    w = int(rng.integers(1, 200))
    cat, params = random_case(rng, model, n=n, w=w)
    assert emul.fast_guard(cat, params, model, CENTRE)
    plain = emul.loglike(cat, params, model, CENTRE, 0, chunk_len=64)
    fast = emul.loglike(cat, params, model, CENTRE, 1, chunk_len=64)
    assert np.array_equal(np.isfinite(plain), np.isfinite(fast))
    ok = np.isfinite(plain)
    assert rel_err(fast[ok], plain[ok]) < 1e-11


@pytest.mark.parametrize("which", ["fixed", "free"])
@pytest.mark.parametrize("chunk_len", [64, 248, 100000])
def test_narrow_range_bgfixed_variant_matches_reference(which, chunk_len):
    """The narrow-range variant of the fixed-background fast path (raw product of four mixture values between rescales,
    no exponent carry, no denormal tracking; chosen by mcd_guard.h: fast_level when pmember < 1 everywhere, lnL_bg >= -150
    and norm >= 2^-60) against lnprob of the reference (runner.py:272-286)."""
    g = load_golden("constant_bg_gaussian_" + which)
    free = which == "free"
    cat = {k: g[k] for k in ("ra", "dec", "v", "verr", "pmember")}
    cat["lnlike_bg"] = g["lnlike_background"]
    centre = None if free else (float(g["ra_center"]), float(g["dec_center"]))
    ok = np.isfinite(g["lnprob"]) & (g["values"][:, 1] > 0)
    values = emul.abi_columns(g["names"], g["values"], 1, free)[ok]
    assert cat["pmember"].max() < 1.0 and cat["lnlike_bg"].min() > -150.0
    assert emul.fast_level(cat, values, 1, centre) == 2
    got = emul.loglike(cat, values, 1, centre, 2, chunk_len)
    assert rel_err(got, g["lnprob"][ok]) < RTOL


def test_narrow_range_level_conditions():
    """A star that falls outside the domain of the narrow-range variant -- a certain member (pmember == 1: the mixture
    value has no floor) or a background likelihood below e^-150 (the value can exceed 2^250) -- only sends ITS CHUNK to
    the general fast form (level stays 2, results agree with the plain path); once more than 1/8 of the stars are such
    exceptions the whole launch uses the general form (level 1).  The narrow variant agrees with the wide one at the edges
    of its domain."""
    g = load_golden("constant_bg_gaussian_fixed")
    centre = (float(g["ra_center"]), float(g["dec_center"]))
    rows = g["values"][np.isfinite(g["lnprior"]) & (g["values"][:, 1] > 0)]
    base = {k: g[k].copy() for k in ("ra", "dec", "v", "verr", "pmember")}
    base["lnlike_bg"] = g["lnlike_background"].copy()
    assert emul.fast_level(base, rows, 1, centre) == 2
    n = len(base["v"])
    for field, value, outlier in (("pmember", 1.0, 400.0), ("lnlike_bg", -151.0, None), ("lnlike_bg", -5000.0, None)):
        cat = {k: v.copy() for k, v in base.items()}
        cat[field][[5, 700, n - 1]] = value
        if outlier is not None:
            cat["v"][5] = outlier                      # a certain member 40 sigma out: y ~ e^-800, far below 2^-250
        assert emul.fast_level(cat, rows, 1, centre) == 2
        plain = emul.loglike(cat, rows, 1, centre, 0, 64)
        mixed = emul.loglike(cat, rows, 1, centre, 2, 64)   # chunks of 64 stars: three of them take the general form
        assert np.array_equal(np.isfinite(plain), np.isfinite(mixed))
        ok = np.isfinite(plain)
        assert rel_err(mixed[ok], plain[ok]) < RTOL
    many = {k: v.copy() for k, v in base.items()}
    many["pmember"][: n // 8 + 1] = 1.0
    assert emul.fast_level(many, rows, 1, centre) == 1
    # (the per-call condition norm >= 2^-60 only binds for velocity scales below 1e-9 km/s: the level-1 guard
    #  |v - v_los|^2 <= 1.6e9 norm is stricter for anything larger)
    assert emul.fast_level(base, rows, 0, centre) == 1              # models without background have no narrow variant
    # edges of the domain: lnL_bg = -150 on a star the cluster model loves (largest y), pmember = 1 - 2^-53 (smallest y)
    # on 80-sigma outliers, four of them in a row so that they share one rescale group
    edge = {k: v.copy() for k, v in base.items()}
    edge["lnlike_bg"][:4] = -150.0
    edge["v"][:4] = rows[0, 0]
    edge["pmember"][8:12] = 1.0 - 2.0 ** -53
    edge["v"][8:12] = 800.0
    edge["verr"][:12] = 1e-3
    assert emul.fast_level(edge, rows, 1, centre) == 2
    narrow = emul.loglike(edge, rows, 1, centre, 2, 64)
    plain = emul.loglike(edge, rows, 1, centre, 0, 64)
    assert np.all(np.isfinite(plain)) and rel_err(narrow, plain) < RTOL
    assert rel_err(emul.loglike(edge, rows, 1, centre, 1, 64), plain) < RTOL


@pytest.mark.parametrize("name,model", [("constant_gb_fixed", 2), ("constant_gb_free", 2), ("model_fit_gb_fixed", 4),
                                        ("model_fit_gb_free", 4)])
@pytest.mark.parametrize("chunk_len", [64, 100000])
def test_narrow_range_gaussian_background_variant_matches_reference(name, model, chunk_len):
    """BgGaussAcc::add<.., NARROW> (raw products, no clamp, one-constant range reduction; density and f_back within
    [2^-20, 2^20]) against lnprob of the reference's ConstantFitGB / ModelFitGB (constant.py:293-364, model.py:391-456)."""
    g = load_golden(name)
    free = name.endswith("free")
    cat = {k: g[k] for k in ("ra", "dec", "v", "verr", "density")}
    centre = None if free else (float(g["ra_center"]), float(g["dec_center"]))
    names = [str(x) for x in g["names"]]
    f = g["values"][:, names.index("f_back")]
    ok = np.isfinite(g["lnprob"]) & (f > 1e-6) & (g["values"][:, names.index("sigma_max")] > 0)
    values = emul.abi_columns(g["names"], g["values"], model, free)[ok]
    assert ok.sum() >= 6 and emul.fast_level(cat, values, model, centre) == 2
    got = emul.loglike(cat, values, model, centre, 2, chunk_len)
    assert rel_err(got, g["lnprob"][ok]) < RTOL
    zero_f = values.copy()
    zero_f[0, -1] = 0.0                                    # f_back = 0: the undamped term can vanish -> general form
    assert emul.fast_level(cat, zero_f, model, centre) == 1
    sparse = dict(cat, density=cat["density"].copy())
    sparse["density"][[7, 333]] = 0.0                      # empty cluster component on two stars: their chunks go general
    assert emul.fast_level(sparse, values, model, centre) == 2
    plain = emul.loglike(sparse, values, model, centre, 0, 64)
    assert rel_err(emul.loglike(sparse, values, model, centre, 2, 64), plain) < RTOL


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_narrow_range_constant_background_profile_variant(which):
    """ModelFitConstantBackground (model.py:565-623) through BgFixedAcc::add_density<NARROW> (f_back >= 2^-20 bounds the
    mixture values from below, lnL_bg >= -120 from above) against lnprob of the reference."""
    g = load_golden("model_fit_cb_" + which)
    free = which == "free"
    cat = {k: g[k] for k in ("ra", "dec", "v", "verr", "density")}
    cat["lnlike_bg"] = g["lnlike_background"]
    centre = None if free else (float(g["ra_center"]), float(g["dec_center"]))
    names = [str(x) for x in g["names"]]
    ok = np.isfinite(g["lnprob"]) & (g["values"][:, names.index("f_back")] > 1e-6) & (g["values"][:, names.index("sigma_max")] > 0)
    values = emul.abi_columns(g["names"], g["values"], 5, free)[ok]
    assert ok.sum() >= 6 and emul.fast_level(cat, values, 5, centre) == 2
    assert rel_err(emul.loglike(cat, values, 5, centre, 2, 64), g["lnprob"][ok]) < RTOL
    zero_f = values.copy()
    zero_f[0, -1] = 0.0
    assert emul.fast_level(cat, zero_f, 5, centre) == 1


def test_narrow_range_profile_variant_without_reciprocal():
    """Round 3: ModelFit (model.py:93-222) through ProfileNarrowAcc -- the Lynden-Bell residual's
    division left to the fraction tree, one-step Newton root for the Plummer dispersion -- against lnprob of the reference
    (1e-12) and, in 80-bit arithmetic, no further from the exact value than the reference's own float64 evaluation."""
    from oracle import lnprob_numpy as oracle
    g = load_golden("model_fit_fixed")
    cat = {k: g[k] for k in ("ra", "dec", "v", "verr")}
    centre = (float(g["ra_center"]), float(g["dec_center"]))
    names = [str(x) for x in g["names"]]
    ok = np.isfinite(g["lnprob"]) & (g["values"][:, names.index("sigma_max")] > 0)
    values = emul.abi_columns(g["names"], g["values"], 3, False)[ok]
    assert ok.sum() >= 6
    for chunk_len in (64, 248, 100000):
        narrow = emul.loglike(cat, values, 3, centre, 2, chunk_len)
        assert rel_err(narrow, g["lnprob"][ok]) < RTOL
        assert rel_err(narrow, emul.loglike(cat, values, 3, centre, 1, chunk_len)) < 1e-13
    L = np.longdouble
    if np.finfo(L).eps < 1e-18:
        catL = {k: v.astype(L) for k, v in cat.items()}
        exact = np.array([oracle.faithful_model_lnlike(catL, *row[:6], L(centre[0]), L(centre[1])) for row in values.astype(L)])
        err_ref = np.max(np.abs((g["lnprob"][ok].astype(L) - exact) / exact)).astype(float)
        err_narrow = np.max(np.abs((emul.loglike(cat, values, 3, centre, 2, 248).astype(L) - exact) / exact)).astype(float)
        assert err_narrow < 5e-12 and err_narrow <= 2.0 * err_ref + 1e-15, (err_narrow, err_ref)
    # wide ranges: random catalogues over the guard's domain of the variant (lengths 2^-10 .. 2^16 arcsec, a degree field)
    rng = np.random.default_rng(31)
    for trial in range(30):
        n = int(rng.integers(1, 700))
        sv = 10.0 ** rng.uniform(-1, 3)
        sep = np.abs(rng.normal(0, 10.0 ** rng.uniform(-3, 0), n))
        th = rng.uniform(-np.pi, np.pi, n)
        c = {"ra": centre[0] + sep * np.cos(th) / np.cos(np.radians(centre[1])), "dec": centre[1] + sep * np.sin(th),
             "v": rng.normal(0, sv, n), "verr": sv * 10.0 ** rng.uniform(-2, 1) * rng.lognormal(0, 0.7, n)}
        w = 7
        p = np.column_stack([rng.normal(0, sv, w), sv * 10.0 ** rng.uniform(-1.5, 1, w), 10.0 ** rng.uniform(-2.5, 4.5, w),
                             rng.normal(0, sv, w), rng.normal(0, sv, w), 10.0 ** rng.uniform(-2.5, 4.5, w)])
        plain = emul.loglike(c, p, 3, centre, 0, 64)
        narrow = emul.loglike(c, p, 3, centre, 2, 64)
        assert np.max(np.abs(narrow - plain) / np.maximum(np.abs(plain), n)) < 1e-11, (trial, narrow, plain)


@pytest.mark.parametrize("name,model", [("constant_fixed", 0), ("constant_bg_gaussian_fixed", 1), ("constant_gb_fixed", 2)])
def test_fast_formulations_are_as_accurate_as_the_float64_reference(name, model):
    """Accuracy, not only agreement: against an 80-bit (numpy.longdouble) evaluation of the reference's formulas the fast
    formulations (fraction tree / log-product / table exp / Newton rsqrt) must not be further from the exact value than
    the reference's own float64 arithmetic is."""
    from oracle import lnprob_numpy as oracle
    L = np.longdouble
    if np.finfo(L).eps > 1e-18:
        pytest.skip("no extended-precision long double on this platform")
    g = load_golden(name)
    keys = ("ra", "dec", "v", "verr") + (("pmember",) if model == 1 else ()) + (("density",) if model == 2 else ())
    cat = {k: g[k] for k in keys}
    if model == 1:
        cat["lnlike_bg"] = g["lnlike_background"]
    centre = (float(g["ra_center"]), float(g["dec_center"]))
    names = [str(x) for x in g["names"]]
    ok = np.isfinite(g["lnprob"]) & (g["values"][:, names.index("sigma_max")] > 0)
    if model == 2:
        ok &= g["values"][:, names.index("f_back")] > 1e-6
    rows = g["values"][ok]
    catL = {k: v.astype(L) for k, v in cat.items()}
    exact = np.empty(len(rows), dtype=L)
    for i, row in enumerate(rows.astype(L)):
        if model == 0:
            exact[i] = oracle.faithful_constant_lnlike(catL, row[0], row[1], row[2], row[3], L(centre[0]), L(centre[1]))
        elif model == 1:
            lnbgL = oracle.gaussian_background(catL["v"], catL["verr"], L(float(g["bg_mean"])), L(float(g["bg_sigma"])))
            exact[i] = oracle.faithful_constant_lnlike(catL, row[0], row[1], row[2], row[3], L(centre[0]), L(centre[1]), lnbgL, catL["pmember"])
        else:
            exact[i] = oracle.faithful_constant_gb_lnlike(catL, row[0], row[1], row[2], row[3], L(centre[0]), L(centre[1]), row[4], row[5], row[6])
    reference64 = g["lnprob"][ok]                                    # the reference itself, float64
    level = emul.fast_level(cat, emul.abi_columns(g["names"], rows, model, False), model, centre)
    assert level >= 1
    fast = emul.loglike(cat, emul.abi_columns(g["names"], rows, model, False), model, centre, level, 248)
    err_ref = np.max(np.abs((reference64.astype(L) - exact) / exact)).astype(float)
    err_fast = np.max(np.abs((fast.astype(L) - exact) / exact)).astype(float)
    # (the common floor, up to 1e-12, is the float64 trigonometry of calc_xy_offset for stars next to the centre, which the
    #  device kernels inherit through the same float64 inputs)
    assert err_fast < 5e-12 and err_ref < 5e-12
    assert err_fast <= 2.0 * err_ref + 4e-16, (err_fast, err_ref)
