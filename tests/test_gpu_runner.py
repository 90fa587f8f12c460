"""GPU: the reference-shaped host API (ConstantFit / ConstantFitGB / background / binned) driving the HIP
kernels, against golden vectors of the reference.  The assertions read like calls into the reference:
``cf.lnprob(values)`` per walker, then the batched entry the sampler uses."""
import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _reader(g, extra=()):
    from mcmc_dynamics_amd import DataReader
    return DataReader({k: g[k] for k in ("ra", "dec", "v", "verr") + tuple(extra)})


def _fix(obj, g):
    obj.parameters["ra_center"].set(value=float(g["ra_center"]), fixed=True)
    obj.parameters["dec_center"].set(value=float(g["dec_center"]), fixed=True)


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_constant_fit_lnprob(which):
    from mcmc_dynamics_amd.analysis import ConstantFit
    g = load_golden("constant_" + which)
    cf = ConstantFit(_reader(g))
    if which == "fixed":
        _fix(cf, g)
    assert cf.fitted_parameters == [str(n) for n in g["names"]]
    single = np.array([cf.lnprob(row) for row in g["values"]])           # one walker per call, as emcee would
    assert rel_err(single, g["lnprob"]) < RTOL                           # includes the -inf rows
    assert rel_err(cf.lnprob_batch(g["values"]), g["lnprob"]) < RTOL
    assert isinstance(cf.lnprob(g["values"][1]), float)
    ok = np.isfinite(g["lnprior"])
    assert rel_err(cf.lnlike_batch(g["values"][ok]), g["lnprob"][ok]) < RTOL


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_constant_fit_with_gaussian_background(which):
    from mcmc_dynamics_amd import Gaussian
    from mcmc_dynamics_amd.analysis import ConstantFit
    g = load_golden("constant_bg_gaussian_" + which)
    cf = ConstantFit(_reader(g, ("pmember",)), background=Gaussian(float(g["bg_mean"]), float(g["bg_sigma"])))
    if which == "fixed":
        _fix(cf, g)
    assert np.max(np.abs(cf.lnlike_background - g["lnlike_background"])) < 1e-12
    assert rel_err(cf.lnprob_batch(g["values"]), g["lnprob"]) < RTOL
    assert rel_err(np.array([cf.lnprob(row) for row in g["values"][:6]]), g["lnprob"][:6]) < RTOL


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_constant_fit_gb(which):
    from mcmc_dynamics_amd.analysis import ConstantFitGB
    g = load_golden("constant_gb_" + which)
    gb = ConstantFitGB(_reader(g, ("density",)))
    if which == "fixed":
        _fix(gb, g)
    assert gb.fitted_parameters == [str(n) for n in g["names"]]
    assert rel_err(gb.lnprob_batch(g["values"]), g["lnprob"]) < RTOL
    mem = gb.membership_probabilities(g["values"][int(g["membership_row"])])
    assert np.max(np.abs(mem - g["membership"])) < 1e-11


def test_changing_the_centre_rebuilds_the_catalogue():
    from mcmc_dynamics_amd.analysis import ConstantFit
    g = load_golden("constant_free")
    cf = ConstantFit(_reader(g))
    free = cf.lnprob_batch(g["values"][:4])
    assert rel_err(free, g["lnprob"][:4]) < RTOL
    row = g["values"][2]
    cf.parameters["ra_center"].set(value=row[4], fixed=True)            # now a fixed-centre catalogue (theta precomputed)
    cf.parameters["dec_center"].set(value=row[5], fixed=True)
    assert rel_err([cf.lnprob(row[:4])], [g["lnprob"][2]]) < RTOL


def test_expression_constrained_parameter():
    """expr-constrained parameters are resolved on the host per walker before the kernel sees them."""
    from mcmc_dynamics_amd.analysis import ConstantFit
    g = load_golden("constant_fixed")
    cf = ConstantFit(_reader(g))
    _fix(cf, g)
    ok = np.isfinite(g["lnprior"])
    full = cf.lnprob_batch(g["values"][ok])
    cf.parameters["v_maxy"].set(expr="v_maxx * 0.5 - 1.0")
    assert cf.fitted_parameters == ["v_sys", "sigma_max", "v_maxx"]
    tied = g["values"][ok].copy()
    tied[:, 3] = tied[:, 2] * 0.5 - 1.0
    cf2 = ConstantFit(_reader(g))
    _fix(cf2, g)
    assert rel_err(cf.lnprob_batch(tied[:, :3]), cf2.lnprob_batch(tied)) < 1e-15
    assert full.shape == (ok.sum(),)


def test_binned_fit_matches_per_bin_reference_runs():
    from mcmc_dynamics_amd.analysis import BinnedConstantFit
    g = load_golden("radial_bins")
    reader = _reader(g)
    reader.make_radial_bins(float(g["ra_center"]), float(g["dec_center"]), nstars=200, dlogr=0.05)
    bf = BinnedConstantFit(reader)
    _fix(bf, g)
    assert bf.n_bins == g["lnprob_per_bin"].shape[0]
    assert rel_err(bf.lnprob_batch(g["values"]), g["lnprob_per_bin"]) < RTOL     # same walkers in every bin
    assert rel_err(bf.lnlike_total(g["values"]), g["lnprob_all"]) < RTOL
    per_bin_pos = np.stack([g["values"] * (1.0 + 0.01 * b) for b in range(bf.n_bins)])
    got = bf.lnprob_batch(per_bin_pos)
    for b in (0, bf.n_bins - 1):
        assert rel_err(got[b], bf.lnprob_batch(np.broadcast_to(per_bin_pos[b], per_bin_pos.shape))[b]) < 1e-15
    sampler = bf(n_walkers=16, n_steps=20, seed=3)
    assert sampler.chain.shape == (bf.n_bins, 16, 20, 4) and np.all(np.isfinite(sampler.lnprobability))
    best = bf.compute_bestfit_values(sampler.chain, n_burn=10)
    assert len(best) == bf.n_bins and 2.0 < best[0].loc["median"]["sigma_max"] < 30.0


def test_binned_sampler_blocks_in_the_library_equal_the_numpy_loop():
    """`mcd_stretch_move` with n_bins = B (one ensemble per radial bin, lock-stepped, every evaluation shared) against the
    NumPy loop of BinnedSampler around `BinnedConstantFit.lnprob_batch`: identical chains, bin for bin."""
    from mcmc_dynamics_amd.analysis import BinnedConstantFit
    from mcmc_dynamics_amd.analysis.binned import BinnedSampler
    g = load_golden("radial_bins")
    reader = _reader(g)
    reader.make_radial_bins(float(g["ra_center"]), float(g["dec_center"]), nstars=200, dlogr=0.05)
    bf = BinnedConstantFit(reader)
    _fix(bf, g)
    rng = np.random.default_rng(12)
    pos = np.array([3.0, 9.0, 1.0, -1.0]) * (1.0 + 0.2 * rng.normal(size=(bf.n_bins, 24, 4)))
    pos[..., 1] = np.abs(pos[..., 1]) + 0.5
    lib_s = BinnedSampler(bf.n_bins, 24, 4, bf.lnprob_batch, seed=21, block_fn=bf._stretch_block)
    ref_s = BinnedSampler(bf.n_bins, 24, 4, bf.lnprob_batch, seed=21)
    lib_s.block_steps = ref_s.block_steps = 8
    lib_s.run_mcmc(pos, 20)
    ref_s.run_mcmc(pos, 20)
    assert np.array_equal(lib_s.chain, ref_s.chain) and np.array_equal(lib_s.lnprobability, ref_s.lnprobability)
    assert np.array_equal(lib_s.acceptance_fraction, ref_s.acceptance_fraction)
    assert lib_s.chain.shape == (bf.n_bins, 24, 20, 4) and 0.1 < lib_s.acceptance_fraction.mean() < 0.9
    assert bf._catalog.stretch_info()["host_blocks"] + bf._catalog.stretch_info()["device_blocks"] == 3
    run = bf(n_walkers=16, n_steps=12, seed=3)                 # through __call__: box priors take the library's blocks
    assert run.block_fn is not None and run.chain.shape == (bf.n_bins, 16, 12, 4)
    # a descriptor whose n_bins does not match the catalogue is refused
    with pytest.raises(Exception, match="n_bins|shapes"):
        bf._catalog.stretch_move(bf._stretch_plan(), pos[0].copy(), np.zeros(24), np.zeros((1, 24), dtype=np.int32),
                                 np.ones((1, 2, 12)), np.zeros((1, 2, 12)), np.zeros((1, 2, 12), dtype=np.int32))


def test_mcmc_on_example_catalogue(tmp_path):
    """C1 plumbing: example/data catalogue, constant-dispersion model, 32 walkers x 100 steps.  Pass =
    finite chain, acceptance fraction in (0.1, 0.9), checkpoints written, posterior near the data's scale."""
    from mcmc_dynamics_amd.analysis import ConstantFit
    g = load_golden("example_catalog")
    cf = ConstantFit(_reader(g))
    _fix(cf, g)
    cf.parameters["sigma_max"].set(initials="rng.lognormal(mean=2.7, sigma=0.3, size=n)")
    cf.parameters["v_sys"].set(initials="rng.normal(loc=10, scale=2, size=n)")
    prefix = str(tmp_path / "c1")
    sampler = cf(n_walkers=32, n_steps=100, n_out=50, prefix=prefix)
    chain = np.asarray(sampler.chain)
    assert chain.shape == (32, 100, 4) and np.all(np.isfinite(chain)) and np.all(np.isfinite(sampler.lnprobability))
    acc = np.mean(sampler.acceptance_fraction)
    assert 0.1 < acc < 0.9
    assert cf.read_chain(prefix + "_chain.pkl").shape == (32, 100, 4)
    best = cf.compute_bestfit_values(chain, n_burn=50)
    assert 5.0 < best.loc["median"]["sigma_max"] < 60.0         # km/s; the example velocities scatter by tens of km/s
    restart = cf.read_final_chain(prefix + "_chain.pkl")
    sampler2 = cf(n_walkers=32, n_steps=5, pos=restart, prefix=None)
    assert np.asarray(sampler2.chain).shape == (32, 5, 4)


@pytest.mark.parametrize("case", ["bg fixed centre", "gb free centre", "one kernel column fixed"])
def test_library_stretch_move_equals_the_python_loop(case):
    """`mcd_stretch_move` (the half-step loop in C++, csrc/mcd_stretch.h, what Runner's built-in sampler uses for box
    priors) produces bit-identical chains to the Python loop of sampler.py driving `lnprob_batch` (runner.py:403-419)."""
    from mcmc_dynamics_amd import DataReader, Gaussian, synthetic
    from mcmc_dynamics_amd.analysis import ConstantFit, ConstantFitGB
    from mcmc_dynamics_amd.sampler import EnsembleSampler
    cat = synthetic.make_catalog(20000, config=3, background=True)
    centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
    names = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]
    if case == "gb free centre":
        fit = ConstantFitGB(DataReader({k: cat[k] for k in ("ra", "dec", "v", "verr", "density")}))
        fit.parameters["ra_center"].set(value=centre[0])
        fit.parameters["dec_center"].set(value=centre[1])
        names = names + ["ra_center", "dec_center", "v_back", "sigma_back", "f_back"]
    else:
        fit = ConstantFit(DataReader({k: cat[k] for k in ("ra", "dec", "v", "verr", "pmember")}), background=Gaussian(20.0, 40.0))
        fit.parameters["ra_center"].set(value=centre[0], fixed=True)
        fit.parameters["dec_center"].set(value=centre[1], fixed=True)
        if case == "one kernel column fixed":
            fit.parameters["v_sys"].set(value=0.3, fixed=True)
            names = names[1:]
    assert fit.fitted_parameters == names
    pos = synthetic.make_walkers(32, names, cat["truth"], config=3)
    pos[:, names.index("sigma_max")] = np.abs(pos[:, names.index("sigma_max")] * (1.0 + 0.5 * np.random.default_rng(1).normal(size=32)))
    for rng_mode in ("device", "host"):                     # numbers generated on the device (the default) / drawn by NumPy
        fit.RNG = rng_mode
        native = fit._make_sampler(32, seed=11)
        assert isinstance(native, EnsembleSampler) and native.block_fn is not None and native.rng == rng_mode
        native.block_steps = 64
        native.run_mcmc(pos, 70)                            # a 64-step block and a 6-step block
        python = EnsembleSampler(32, len(names), fit.lnprob_batch, vectorize=True, seed=11, rng=rng_mode)
        python.block_steps = 64 if rng_mode == "host" else 9    # (host draws: the partition defines the stream; device: any)
        python.run_mcmc(pos, 70)
        assert np.array_equal(native.chain, python.chain) and np.array_equal(native.lnprobability, python.lnprobability)
        assert np.array_equal(native.acceptance_fraction, python.acceptance_fraction) and native.n_calls == python.n_calls
        assert 0.05 < native.acceptance_fraction.mean() < 0.95 and np.all(np.isfinite(native.lnprobability))
    fit.RNG = "device"
    # through Runner.__call__ as a user would (restarts every n_out steps included)
    run = fit(n_walkers=32, n_steps=20, n_out=10, pos=pos, prefix=None)
    assert np.asarray(run.chain).shape == (32, 20, len(names))
    fit.close()


# ------------------------------------------------------------------------------------------ ModelFit family ("next" row 1)
@pytest.mark.parametrize("which", ["fixed", "free"])
def test_model_fit(which):
    from mcmc_dynamics_amd.analysis import ModelFit
    g = load_golden("model_fit_" + which)
    mf = ModelFit(_reader(g))
    if which == "fixed":
        _fix(mf, g)
    assert mf.fitted_parameters == [str(n) for n in g["names"]]       # config/model.json order, centre in the middle
    assert rel_err(mf.lnprob_batch(g["values"]), g["lnprob"]) < RTOL
    assert rel_err(np.array([mf.lnprob(row) for row in g["values"][:4]]), g["lnprob"][:4]) < RTOL
    mf.close()
    mf._precision = "f32acc64"
    ok = np.isfinite(g["lnprob"])
    try:
        got = mf.lnlike_batch(g["values"][ok])
    except Exception as exc:                               # this small golden catalogue may lie outside the float32 accuracy domain
        assert "float32 accuracy domain" in str(exc), exc
        mf._ensure_catalog().set_option("f32_domain", 0)
        got = mf.lnlike_batch(g["values"][ok])
    assert rel_err(got, g["lnprob"][ok]) < 1e-5


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_model_fit_gb(which):
    from mcmc_dynamics_amd.analysis import ModelFitGB
    g = load_golden("model_fit_gb_" + which)
    mg = ModelFitGB(_reader(g, ("density",)))
    if which == "fixed":
        _fix(mg, g)
    assert mg.fitted_parameters == [str(n) for n in g["names"]]
    assert rel_err(mg.lnprob_batch(g["values"]), g["lnprob"]) < RTOL
    mem = mg.membership_probabilities(g["values"][1])
    assert mem.shape == (len(g["v"]),) and np.all((mem >= 0) & (mem <= 1))


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_model_fit_constant_background(which):
    from mcmc_dynamics_amd import Gaussian, Parameters
    from mcmc_dynamics_amd.analysis import ModelFitConstantBackground
    g = load_golden("model_fit_cb_" + which)
    pars = ModelFitConstantBackground.default_parameters()
    del pars["v_back"]
    del pars["sigma_back"]
    mc = ModelFitConstantBackground(_reader(g, ("density",)), Gaussian(float(g["bg_mean"]), float(g["bg_sigma"])),
                                    parameters=pars)
    if which == "fixed":
        _fix(mc, g)
    assert mc.fitted_parameters == [str(n) for n in g["names"]]
    assert rel_err(mc.lnprob_batch(g["values"]), g["lnprob"]) < RTOL
    row = g["values"][int(g["no_sum_row"])]
    per_star = mc.lnlike(row, no_sum=True)                             # model.py:565-623, no_sum=True
    assert np.max(np.abs(per_star - g["lnlike_no_sum"]) / np.abs(g["lnlike_no_sum"])) < RTOL
    assert abs(per_star.sum() - mc.lnlike(row)) < 1e-9 * abs(per_star.sum())
    from oracle import lnprob_numpy as oracle
    named = dict(zip([str(n) for n in g["names"]], row))
    if which == "fixed":
        named["ra_center"], named["dec_center"] = float(g["ra_center"]), float(g["dec_center"])
    cat = {k: g[k] for k in ("ra", "dec", "v", "verr", "density")}
    v_los = oracle.model_rotation(cat["ra"], cat["dec"], named["v_sys"], named["v_maxx"], named["v_maxy"], named["r_peak"],
                                  named["ra_center"], named["dec_center"])
    sig = oracle.model_dispersion(cat["ra"], cat["dec"], named["sigma_max"], named["a"], named["ra_center"], named["dec_center"])
    norm = cat["verr"] ** 2 + sig ** 2
    lc = -0.5 * np.log(2 * np.pi * norm) - 0.5 * (cat["v"] - v_los) ** 2 / norm
    want = oracle.model_membership(cat, lc, g["lnlike_background"], cat["density"] / (cat["density"] + named["f_back"]))
    assert np.max(np.abs(mc.membership_probabilities(row) - want)) < 1e-11
    # host-side model functions agree with the reference's formulas
    assert np.max(np.abs(mc.rotation_model(named["v_sys"], named["v_maxx"], named["v_maxy"], named["ra_center"],
                                           named["dec_center"], named["r_peak"]) - v_los)) < 1e-9
    assert np.max(np.abs(mc.dispersion_model(named["sigma_max"], named["ra_center"], named["dec_center"], named["a"]) - sig)) < 1e-10


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_model_fit_with_fixed_background(which):
    """ModelFit(data, background=Gaussian(...)): the reference's ModelFit.lnlike ends in Runner._calculate_lnlike, which
    applies the pmember mixture to every subclass (model.py:222 -> runner.py:272-286).  Golden values from the reference."""
    from mcmc_dynamics_amd import Gaussian, _native
    from mcmc_dynamics_amd.analysis import ModelFit
    g = load_golden("model_fit_bg_gaussian_" + which)
    mf = ModelFit(_reader(g, ("pmember",)), background=Gaussian(float(g["bg_mean"]), float(g["bg_sigma"])))
    if which == "fixed":
        _fix(mf, g)
    assert mf.fitted_parameters == [str(n) for n in g["names"]]
    assert mf._catalog_model()[0] == _native.MODEL_PROFILE_BGFIXED
    assert np.max(np.abs(mf.lnlike_background - g["lnlike_background"])) < 1e-12
    assert rel_err(mf.lnprob_batch(g["values"]), g["lnprob"]) < RTOL                 # includes the -inf row
    assert rel_err(np.array([mf.lnprob(row) for row in g["values"][:3]]), g["lnprob"][:3]) < RTOL
    ok = np.isfinite(g["lnprob"])
    mf._catalog.set_option("fast_path", 0)                                             # the plain kernels agree as well
    assert rel_err(mf.lnlike_batch(g["values"][ok]), g["lnprob"][ok]) < RTOL
    # without the background the same class evaluates the plain profile model: a different number
    plain = ModelFit(_reader(g))
    if which == "fixed":
        _fix(plain, g)
    assert np.all(np.abs(plain.lnlike_batch(g["values"][ok]) - g["lnprob"][ok]) > 1.0)


@pytest.mark.parametrize("which", ["fixed", "free"])
def test_model_fit_gb_membership_matches_reference(which):
    """ModelFitGB.calculate_membership_probabilities(chain, n_burn) (model.py:458-510) against the reference's output."""
    from mcmc_dynamics_amd.analysis import ModelFitGB
    g = load_golden("model_fit_gb_membership_" + which)
    mg = ModelFitGB(_reader(g, ("density",)))
    if which == "fixed":
        _fix(mg, g)
    assert mg.fitted_parameters == [str(n) for n in g["names"]]
    got = mg.calculate_membership_probabilities(g["chain"], n_burn=int(g["n_burn"]))
    assert got.shape == g["membership"].shape and np.max(np.abs(got - g["membership"])) < 1e-11
    assert np.array_equal([mg.parameters[str(n)].value for n in g["names"]], g["median"])


def test_model_fit_profiles_and_full_size():
    """create_profiles post-processing, and a 1e6-star ModelFit batch: fast == plain formulation."""
    from mcmc_dynamics_amd import DataReader, synthetic
    from mcmc_dynamics_amd.analysis import ModelFit
    c = synthetic.make_catalog(1000000, config=3)
    mf = ModelFit(DataReader({k: c[k] for k in ("ra", "dec", "v", "verr")}))
    mf.parameters["ra_center"].set(value=c["truth"]["ra_center"], fixed=True)
    mf.parameters["dec_center"].set(value=c["truth"]["dec_center"], fixed=True)
    rng = np.random.default_rng(5)
    truth = [0.0, 10.0, 30.0, c["truth"]["v_maxx"], c["truth"]["v_maxy"], 60.0]
    pos = np.array(truth) * (1.0 + 0.05 * rng.normal(size=(128, 6))) + np.array([0.5, 0, 0, 0, 0, 0]) * rng.normal(size=(128, 1))
    fast = mf.lnprob_batch(pos)
    mf._ensure_catalog().set_option("fast_path", 0)
    assert rel_err(mf.lnprob_batch(pos), fast) < RTOL
    chain = np.broadcast_to(pos[:, None, :], (128, 12, 6)) * (1.0 + 0.01 * rng.normal(size=(128, 12, 6)))
    prof = mf.create_profiles(chain, n_burn=2, radii=[10.0, 60.0, 200.0])
    assert len(prof) == 3 and prof["v_rot"][1] == pytest.approx(5.0, rel=0.2) and prof["sigma"][0] < 10.6


def test_example_script_runs(tmp_path):
    """examples/run_constant_fit.py end to end (synthetic cluster -> ConstantFit + background -> MCMC -> tables)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    res = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "run_constant_fit.py"), "--stars", "5000",
                          "--walkers", "32", "--steps", "40"], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "acceptance fraction" in res.stdout and "sigma_max" in res.stdout and "theta_0" in res.stdout
