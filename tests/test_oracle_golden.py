"""The CPU oracle (oracle/lnprob_numpy.py) against golden vectors generated from the reference itself
(oracle/make_golden.py, conda python 3.9 / numpy 1.26.4 / astropy 4.3.1).  This is what pins the oracle:
the reference ships no tests or fixtures of its own.

Tolerance: 1e-13 relative -- NumPy 2.2 here vs NumPy 1.26 there differ in SIMD sin/cos/log by an ulp."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import lnprob_numpy as oracle

RTOL = 1e-13


def _cat(g, extra=()):
    return {k: g[k] for k in ("ra", "dec", "v", "verr") + tuple(extra)}


def _split(g):
    ok = np.isfinite(g["lnprior"])
    return g["values"], ok


def test_prior_pattern_matches_reference():
    """Inclusive bounds: sigma = 0, f_back = 0 and f_back = 1 are accepted; sigma < 0, f_back > 1,
    sigma_back < 0, dec_center = 91 deg are rejected (parameter.py:691)."""
    from mcmc_dynamics_amd.synthetic import BOUNDS
    for name in ("constant_fixed", "constant_free", "constant_gb_fixed", "constant_gb_free",
                 "constant_bg_gaussian_fixed", "constant_bg_gaussian_free"):
        g = load_golden(name)
        lo = np.array([BOUNDS[str(n)][0] for n in g["names"]])
        hi = np.array([BOUNDS[str(n)][1] for n in g["names"]])
        got = np.array([oracle.bounds_lnprior(row, lo, hi) for row in g["values"]])
        assert np.array_equal(got, g["lnprior"])
        assert np.array_equal(np.isfinite(g["lnprob"]), np.isfinite(g["lnprior"]))
        assert (~np.isfinite(g["lnprior"])).sum() >= 1 and np.isfinite(g["lnprior"]).sum() >= 16


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_constant_fit(which):
    g = load_golden("constant_" + which)
    cat = _cat(g)
    values, ok = _split(g)
    centre = (g["ra_center"], g["dec_center"]) if which == "fixed" else ()
    faithful = np.array([oracle.faithful_constant_lnlike(cat, *row, *centre) for row in values[ok]])
    assert rel_err(faithful, g["lnprob"][ok]) < RTOL
    batched = oracle.batched_constant_lnlike(cat, values[ok], *centre)
    assert rel_err(batched, g["lnprob"][ok]) < RTOL


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_fixed_gaussian_background(which):
    g = load_golden("constant_bg_gaussian_" + which)
    cat = _cat(g, ("pmember",))
    lnbg = oracle.gaussian_background(cat["v"], cat["verr"], float(g["bg_mean"]), float(g["bg_sigma"]))
    assert np.max(np.abs(lnbg - g["lnlike_background"])) < 1e-12          # background/gaussian.py:23-28
    values, ok = _split(g)
    centre = (g["ra_center"], g["dec_center"]) if which == "fixed" else ()
    faithful = np.array([oracle.faithful_constant_lnlike(cat, *row, *centre, lnlike_background=lnbg,
                                                         pmember=cat["pmember"]) for row in values[ok]])
    assert rel_err(faithful, g["lnprob"][ok]) < RTOL
    kw = dict(lnlike_background=lnbg, pmember=cat["pmember"])
    if which == "fixed":
        kw.update(ra_center=g["ra_center"], dec_center=g["dec_center"])
    assert rel_err(oracle.batched_constant_lnlike(cat, values[ok], **kw), g["lnprob"][ok]) < RTOL


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_walker_gaussian_background(which):
    g = load_golden("constant_gb_" + which)
    cat = _cat(g, ("density",))
    values, ok = _split(g)

    def full(row):
        if which == "fixed":
            return tuple(row[:4]) + (g["ra_center"], g["dec_center"]) + tuple(row[4:])
        return tuple(row)

    faithful = np.array([oracle.faithful_constant_gb_lnlike(cat, *full(row)) for row in values[ok]])
    assert rel_err(faithful, g["lnprob"][ok]) < RTOL
    centre = (g["ra_center"], g["dec_center"]) if which == "fixed" else ()
    assert rel_err(oracle.batched_constant_gb_lnlike(cat, values[ok], *centre), g["lnprob"][ok]) < RTOL
    row = values[int(g["membership_row"])]
    lc, lb, m = oracle.faithful_constant_gb_terms(cat, *full(row))
    assert np.max(np.abs(lc - g["lnlike_cluster"]) / np.abs(g["lnlike_cluster"])) < RTOL
    assert np.max(np.abs(lb - g["lnlike_back"]) / np.abs(g["lnlike_back"])) < RTOL
    assert np.max(np.abs(m - g["prior_m"])) < 1e-15
    assert np.max(np.abs(oracle.gb_membership_probabilities(cat, *full(row)) - g["membership"])) < 1e-12


def test_radial_bins_and_per_bin_values():
    g = load_golden("radial_bins")
    dx, dy = oracle.calc_xy_offset(g["ra"], g["dec"], g["ra_center"], g["dec_center"])
    r = np.sqrt(dx ** 2 + dy ** 2)
    assert np.max(np.abs(r - g["r_arcmin"]) / g["r_arcmin"]) < 1e-12
    assert np.array_equal(oracle.make_radial_bins(g["r_arcmin"], 200, 0.05), g["bins_n200_d005"])
    assert np.array_equal(oracle.make_radial_bins(g["r_arcmin"], 50, 0.1), g["bins_n50_d01"])
    bins = g["bins_n200_d005"]
    cat = _cat(g)
    total = np.zeros(len(g["values"]))
    for b in range(int(bins.max()) + 1):
        sub = {k: v[bins == b] for k, v in cat.items()}
        got = oracle.batched_constant_lnlike(sub, g["values"], g["ra_center"], g["dec_center"])
        assert rel_err(got, g["lnprob_per_bin"][b]) < RTOL
        total += got
    assert rel_err(total, g["lnprob_all"]) < RTOL        # sum over bins == un-binned (known-answer 4)


def test_example_catalogue_adapter():
    """example/data/test.csv is in a legacy polar layout; the adapter must hand the reference's own
    calc_xy_offset the original (r, theta) back."""
    g = load_golden("example_catalog")
    assert g["r"].shape == (6284,)
    assert np.max(np.abs(np.hypot(g["dx"], g["dy"]) - g["r"])) < 1e-9
    assert np.max(np.abs(np.arctan2(g["dy"], g["dx"]) - g["theta"])) < 1e-9
    cat = _cat(g)
    got = oracle.batched_constant_lnlike(cat, g["values"], g["ra_center"], g["dec_center"])
    assert rel_err(got, g["lnprob"]) < RTOL


def test_closed_form_known_answer():
    """v_max = 0: lnL = sum -1/2 [log(2 pi (e^2 + s^2)) + (v - v_sys)^2 / (e^2 + s^2)] on hand-computable stars."""
    cat = {"ra": np.array([10.0, 10.01, 9.99]), "dec": np.array([0.0, 0.01, -0.01]),
           "v": np.array([1.0, -2.0, 0.5]), "verr": np.array([1.0, 2.0, 0.5])}
    v_sys, sigma = 0.25, 3.0
    n = cat["verr"] ** 2 + sigma ** 2
    want = np.sum(-0.5 * (np.log(2 * np.pi * n) + (cat["v"] - v_sys) ** 2 / n))
    got = oracle.faithful_constant_lnlike(cat, v_sys, sigma, 0.0, 0.0, 10.0, 0.0)
    assert abs(got - want) < 1e-13
    # pmember == 1 mixture == no-background value (known-answer 2)
    lnbg = oracle.gaussian_background(cat["v"], cat["verr"], 20.0, 40.0)
    mix = oracle.faithful_constant_lnlike(cat, v_sys, sigma, 1.0, -0.5, 10.0, 0.0, lnbg, np.ones(3))
    plain = oracle.faithful_constant_lnlike(cat, v_sys, sigma, 1.0, -0.5, 10.0, 0.0)
    assert abs(mix - plain) < 1e-13


# ------------------------------------------------------------------------------------------ ModelFit family
def _named(g, row, which):
    d = dict(zip([str(n) for n in g["names"]], row))
    if which == "fixed":
        d["ra_center"], d["dec_center"] = float(g["ra_center"]), float(g["dec_center"])
    return d


CORE = ("v_sys", "sigma_max", "a", "v_maxx", "v_maxy", "r_peak", "ra_center", "dec_center")


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_model_fit_family(which):
    """ModelFit / ModelFitGB / ModelFitConstantBackground (analysis/model.py:93-222, 391-456, 565-623), including
    the arcmin-vs-arcsec unit reduction astropy performs inside the reference."""
    g = load_golden("model_fit_" + which)
    ok = np.isfinite(g["lnprob"])
    got = np.array([oracle.faithful_model_lnlike(_cat(g), **{k: _named(g, r, which)[k] for k in CORE}) for r in g["values"][ok]])
    assert rel_err(got, g["lnprob"][ok]) < RTOL and (~ok).sum() == 1

    g = load_golden("model_fit_gb_" + which)
    ok = np.isfinite(g["lnprob"])
    keys = CORE + ("v_back", "sigma_back", "f_back")
    got = np.array([oracle.faithful_model_gb_lnlike(_cat(g, ("density",)), **{k: _named(g, r, which)[k] for k in keys})
                    for r in g["values"][ok]])
    assert rel_err(got, g["lnprob"][ok]) < RTOL and (~ok).sum() == 2

    g = load_golden("model_fit_cb_" + which)
    ok = np.isfinite(g["lnprob"])
    lnbg = oracle.gaussian_background(g["v"], g["verr"], float(g["bg_mean"]), float(g["bg_sigma"]))
    assert np.max(np.abs(lnbg - g["lnlike_background"])) < 1e-12
    keys = CORE + ("f_back",)
    got = np.array([oracle.faithful_model_cb_lnlike(_cat(g, ("density",)), lnlike_background=lnbg,
                                                    **{k: _named(g, r, which)[k] for k in keys}) for r in g["values"][ok]])
    assert rel_err(got, g["lnprob"][ok]) < RTOL
    row = g["values"][int(g["no_sum_row"])]
    per_star = oracle.faithful_model_cb_lnlike(_cat(g, ("density",)), lnlike_background=lnbg, no_sum=True,
                                               **{k: _named(g, row, which)[k] for k in keys})
    assert np.max(np.abs(per_star - g["lnlike_no_sum"]) / np.abs(g["lnlike_no_sum"])) < RTOL


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_model_fit_with_fixed_background(which):
    """ModelFit(background=Gaussian): ModelFit.lnlike ends in Runner._calculate_lnlike (model.py:222 -> runner.py:272-286),
    so the pmember mixture applies to the profile models as well (round-2 fixture from the reference)."""
    g = load_golden("model_fit_bg_gaussian_" + which)
    ok = np.isfinite(g["lnprob"])
    lnbg = oracle.gaussian_background(g["v"], g["verr"], float(g["bg_mean"]), float(g["bg_sigma"]))
    assert np.max(np.abs(lnbg - g["lnlike_background"])) < 1e-12
    got = np.array([oracle.faithful_model_lnlike(_cat(g), lnlike_background=lnbg, prior=g["pmember"],
                                                 **{k: _named(g, r, which)[k] for k in CORE}) for r in g["values"][ok]])
    assert rel_err(got, g["lnprob"][ok]) < RTOL and (~ok).sum() == 1
    assert np.array_equal(np.isfinite(g["lnprior"]), ok)


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_model_fit_gb_membership_matches_reference(which):
    """ModelFitGB.calculate_membership_probabilities (model.py:458-510) at the chain's medians."""
    g = load_golden("model_fit_gb_membership_" + which)
    chain, n_burn = g["chain"], int(g["n_burn"])
    med = np.percentile(chain[:, n_burn:, :].reshape(-1, chain.shape[2]), 50, axis=0)
    assert np.array_equal(med, g["median"])
    p = _named(g, med, which)
    cat = _cat(g, ("density",))
    v_los = oracle.model_rotation(cat["ra"], cat["dec"], p["v_sys"], p["v_maxx"], p["v_maxy"], p["r_peak"], p["ra_center"], p["dec_center"])
    sig = oracle.model_dispersion(cat["ra"], cat["dec"], p["sigma_max"], p["a"], p["ra_center"], p["dec_center"])
    norm = cat["verr"] ** 2 + sig ** 2
    lc = -0.5 * np.log(2 * np.pi * norm) - 0.5 * (cat["v"] - v_los) ** 2 / norm
    lb = oracle.gaussian_background(cat["v"], cat["verr"], p["v_back"], p["sigma_back"])
    got = oracle.model_membership(cat, lc, lb, cat["density"] / (cat["density"] + p["f_back"]))
    assert np.max(np.abs(got - g["membership"])) < 1e-13


def test_single_stars_background_matches_reference():
    """oracle.single_stars_background against background.SingleStars of the reference (single_stars.py:42-77):
    sigma_int = 0 and 2.5 km/s, a > 1e4-sigma outlier, a test star exactly on a comparison star, and M = 1."""
    g = load_golden("single_stars")
    for tag in ("s0", "s2"):
        got = oracle.single_stars_background(g["comp"], g["v"], g["verr"], float(g["sigma_int_" + tag]))
        assert rel_err(got, g["lnlike_" + tag]) < 1e-15
    assert rel_err(oracle.single_stars_background([12.5], g["v"][:50], g["verr"][:50]), g["lnlike_m1"]) < 1e-15
    assert np.array_equal(g["lnlike_background"], g["lnlike_s0"])
    # the fixture's lnprob rows: ConstantFit(background=SingleStars) of the reference
    cat = {k: g[k] for k in ("ra", "dec", "v", "verr")}
    lnbg = oracle.single_stars_background(g["comp"], g["v"], g["verr"])
    assert np.isfinite(g["lnprior"]).sum() >= 8
    for row, want, prior in zip(g["values"], g["lnprob"], g["lnprior"]):
        if not np.isfinite(prior):
            assert want == -np.inf
            continue
        got = oracle.faithful_constant_lnlike(cat, *row, float(g["ra_center"]), float(g["dec_center"]),
                                              lnlike_background=lnbg, pmember=g["pmember"])
        assert abs(got - want) <= RTOL * abs(want)


def test_c1_plumbing_on_the_cpu_path():
    """BASELINE config C1: example/data catalogue, constant-dispersion model, centre fixed, 32 walkers x
    100 steps with the stretch-move driver on the CPU restatement of the reference's lnprob -- no GPU.  Pass = finite chain,
    acceptance fraction in (0.1, 0.9), posterior medians stable across seeds."""
    from mcmc_dynamics_amd.sampler import EnsembleSampler
    g = load_golden("example_catalog")
    cat = _cat(g)
    rc, dc = float(g["ra_center"]), float(g["dec_center"])

    def lnprob(x):                                     # flat priors of config/constant.json: sigma_max >= 0
        out = np.full(len(x), -np.inf)
        ok = x[:, 1] >= 0
        out[ok] = oracle.batched_constant_lnlike(cat, x[ok], rc, dc)
        return out

    medians = []
    for seed in (1, 2):
        rng = np.random.default_rng(seed)
        pos = np.column_stack([rng.normal(10, 2, 32), rng.lognormal(2.7, 0.3, 32), rng.normal(0, 2, 32), rng.normal(0, 2, 32)])
        s = EnsembleSampler(32, 4, lnprob, vectorize=True, seed=seed)
        s.run_mcmc(pos, 100)
        assert s.chain.shape == (32, 100, 4) and np.all(np.isfinite(s.chain)) and np.all(np.isfinite(s.lnprobability))
        assert 0.1 < s.acceptance_fraction.mean() < 0.9
        medians.append(np.median(s.get_chain(discard=60, flat=True), axis=0))
    spread = np.abs(medians[0] - medians[1])
    assert spread[1] < 0.15 * medians[0][1] and np.all(spread[[0, 2, 3]] < 3.0), (medians, spread)


def _emcee_stretch_reference(log_prob_fn, coords, nsteps, seed, a=2.0):
    """A straight NumPy transcription of emcee 3's default move (``emcee.moves.StretchMove`` on top of ``RedBlueMove``
    with ``nsplits=2, randomize_split=True``), written from emcee's published algorithm (Goodman & Weare 2010; emcee is not
    installed here): a shuffled 0/1 labelling splits the ensemble, each half S is updated against the other half C with
    ``z = ((a - 1) u + 1)^2 / a``, ``q = c[r] - (c[r] - s) z``, accepted iff ``(ndim - 1) log z + lnp(q) - lnp(s) > log u``.
    Not the package's sampler: its own generator, its own order of draws."""
    rnd = np.random.RandomState(seed)
    coords = np.array(coords, dtype=np.float64)
    nwalkers, ndim = coords.shape
    lnp = log_prob_fn(coords)
    chain = np.empty((nsteps, nwalkers, ndim))
    n_acc = np.zeros(nwalkers)
    for step in range(nsteps):
        inds = np.arange(nwalkers) % 2
        rnd.shuffle(inds)
        for split in range(2):
            s1 = inds == split
            sets = [coords[inds == j] for j in range(2)]
            s, c = sets[split], sets[1 - split]
            ns, nc = len(s), len(c)
            zz = ((a - 1.0) * rnd.rand(ns) + 1.0) ** 2.0 / a
            factors = (ndim - 1.0) * np.log(zz)
            rint = rnd.randint(nc, size=(ns,))
            q = c[rint] - (c[rint] - s) * zz[:, None]
            new_lnp = log_prob_fn(q)
            lnpdiff = factors + new_lnp - lnp[s1]
            accepted = lnpdiff > np.log(rnd.rand(ns))
            idx = np.flatnonzero(s1)[accepted]
            coords[idx] = q[accepted]
            lnp[idx] = new_lnp[accepted]
            n_acc[idx] += 1
        chain[step] = coords
    return chain, n_acc / nsteps


def test_builtin_stretch_move_samples_like_emcees_default_move():
    """the package's sampler (what drives the resident blocks when emcee is absent) against a
    transcription of emcee's StretchMove on the C1 catalogue (example/data, ConstantFit, centre fixed, 32 walkers): same
    acceptance fraction within 0.03 and posterior medians within 0.2 posterior standard deviations, over three seeds.
    (Statistical equivalence -- the two draw their random numbers in different orders, so chains are not comparable row by
    row; emcee itself is not importable here: parity with the installed package stays unpinned.)"""
    from mcmc_dynamics_amd.sampler import EnsembleSampler
    g = load_golden("example_catalog")
    cat = _cat(g)
    rc, dc = float(g["ra_center"]), float(g["dec_center"])

    def lnprob(x):
        out = np.full(len(x), -np.inf)
        ok = x[:, 1] >= 0
        if ok.any():
            out[ok] = oracle.batched_constant_lnlike(cat, x[ok], rc, dc)
        return out

    n_steps, burn = 700, 200
    acc, med, std = {"builtin": [], "emcee": [], "device_rng": []}, {"builtin": [], "emcee": [], "device_rng": []}, []
    for seed in (11, 12, 13):
        rng = np.random.default_rng(seed)
        pos = np.column_stack([rng.normal(0, 1, 32), rng.lognormal(2.2, 0.2, 32), rng.normal(0, 1, 32), rng.normal(0, 1, 32)])
        s = EnsembleSampler(32, 4, lnprob, vectorize=True, seed=seed)
        s.run_mcmc(pos, n_steps)
        flat = s.get_chain(discard=burn, flat=True)
        acc["builtin"].append(s.acceptance_fraction.mean())
        med["builtin"].append(np.median(flat, axis=0))
        std.append(flat.std(axis=0))
        # ... and the same move with the counter-based random numbers of csrc/mcd_rng.h (rng="device": what the device
        # generates for seeded blocks; here through the library's host entry point mcd_chain_numbers)
        s = EnsembleSampler(32, 4, lnprob, vectorize=True, seed=seed, rng="device")
        s.run_mcmc(pos, n_steps)
        acc["device_rng"].append(s.acceptance_fraction.mean())
        med["device_rng"].append(np.median(s.get_chain(discard=burn, flat=True), axis=0))
        chain, frac = _emcee_stretch_reference(lnprob, pos, n_steps, seed + 100)
        acc["emcee"].append(frac.mean())
        med["emcee"].append(np.median(chain[burn:].reshape(-1, 4), axis=0))
    sigma = np.mean(std, axis=0)
    assert abs(np.mean(acc["builtin"]) - np.mean(acc["emcee"])) < 0.03, acc
    assert 0.3 < np.mean(acc["builtin"]) < 0.8
    diff = np.abs(np.mean(med["builtin"], axis=0) - np.mean(med["emcee"], axis=0)) / sigma
    assert np.all(diff < 0.2), (diff, med, sigma)
    assert abs(np.mean(acc["device_rng"]) - np.mean(acc["emcee"])) < 0.03, acc
    diff = np.abs(np.mean(med["device_rng"], axis=0) - np.mean(med["emcee"], axis=0)) / sigma
    assert np.all(diff < 0.2), (diff, med, sigma)
