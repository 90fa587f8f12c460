"""Host logic (no GPU): Parameters / DataReader / backgrounds / Runner contract / sampler, and the
C-ABI library's load-time contract (every symbol of include/mcd.h exported; no compute calls)."""
import io
import json
import os
import pickle
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden, rel_err

from mcmc_dynamics_amd import DataReader, Gaussian, Parameter, Parameters, SingleStars, _native, synthetic
from mcmc_dynamics_amd.analysis import BinnedConstantFit, ConstantFit, ConstantFitGB, Runner
from mcmc_dynamics_amd.sampler import EnsembleSampler

# a parameter file in the reference's JSON layout (same schema as its config/constant_with_background.json)
REFERENCE_LAYOUT_JSON = json.dumps({
    "unique_symbols": {"rng_seed": 5},
    "params": [
        ["v_sys", None, "km/s", False, float("-inf"), float("inf"), "$v_{\\rm sys}$", "rng.normal(size=n)", None, None, None],
        ["sigma_max", None, "km/s", False, 0.0, float("inf"), "$\\sigma$", "rng.lognormal(size=n)", None, None, None],
        ["v_maxx", None, "km/s", False, float("-inf"), float("inf"), None, "rng.normal(size=n)", None, None, None],
        ["v_maxy", None, "km/s", False, float("-inf"), float("inf"), None, None, None, None, "2*v_maxx"],
        ["ra_center", 56.345, "deg", True, 0.0, 360.0, None, None, None, None, None],
        ["dec_center", -26.675, "deg", True, -90.0, 90.0, None, None, None, None, None],
        ["f_back", None, None, False, 0.0, 1.0, None, "rng.uniform(size=n)", "log(val + 1)", None, None],
    ]})


def small_reader(n=300, background=True):
    c = synthetic.make_catalog(n, config=2, background=background)
    return DataReader({k: c[k] for k in c if k != "truth"}), c


# ------------------------------------------------------------------------------------------ native library
def test_library_exports_every_declared_symbol(built_library):
    header = open(os.path.join(ROOT, "include", "mcd.h")).read()
    declared = set(re.findall(r"\b(mcd_[a-z_0-9]+)\s*\(", header))
    declared -= {"mcd_ctx", "mcd_catalog"}
    assert len(declared) >= 20
    lib = _native.load_library()
    for name in sorted(declared):
        assert hasattr(lib, name), "libmcd_hip.so does not export " + name
        assert name in _native.SYMBOLS, "ctypes binding lacks " + name
    assert lib.mcd_abi_version() == 1


def test_no_cpu_fallback(built_library):
    """Without a usable gfx950 device the product path must fail loudly, not compute on the host."""
    lib = _native.load_library()
    import ctypes
    n = ctypes.c_int(0)
    hip = ctypes.CDLL("libamdhip64.so")
    if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    reader, c = small_reader()
    cf = ConstantFit(reader)
    cf.parameters["ra_center"].set(value=c["truth"]["ra_center"], fixed=True)
    cf.parameters["dec_center"].set(value=c["truth"]["dec_center"], fixed=True)
    with pytest.raises(_native.NativeError):
        cf.lnprob(np.array([0.0, 10.0, 1.0, 1.0]))
    with pytest.raises(_native.NativeError):
        _native.load_library(os.path.join(ROOT, "does_not_exist.so"))
    import mcmc_dynamics_amd
    src = "".join(open(os.path.join(dp, f)).read() for dp, _, fs in os.walk(os.path.dirname(mcmc_dynamics_amd.__file__))
                  for f in fs if f.endswith(".py"))
    assert "oracle" not in src.replace("oracle/make_golden.py", ""), "the product package must not reference the oracle"


# ------------------------------------------------------------------------------------------ Parameters
def test_parameters_reference_json_layout_roundtrip():
    pars = Parameters().load(io.StringIO(REFERENCE_LAYOUT_JSON))
    assert list(pars) == ["v_sys", "sigma_max", "v_maxx", "v_maxy", "ra_center", "dec_center", "f_back"]
    assert pars.free_names() == ["v_sys", "sigma_max", "v_maxx", "f_back"]        # expr => fixed
    assert pars["sigma_max"].min == 0.0 and pars["f_back"].max == 1.0 and pars["v_sys"].unit == "km/s"
    assert pars["f_back"].value == 0.5                     # midpoint of finite bounds (parameter.py:795-796)
    again = Parameters().loads(pars.dumps())
    assert [p.__getstate__() for p in again.values()] == [p.__getstate__() for p in pars.values()]
    clone = pickle.loads(pickle.dumps(pars))
    assert list(clone) == list(pars) and clone["v_maxy"].expr == "2*v_maxx"
    draws = pars["v_sys"].evaluate_initials(7)
    assert draws.shape == (7,)
    assert np.allclose(Parameters().load(io.StringIO(REFERENCE_LAYOUT_JSON))["v_sys"].evaluate_initials(7), draws)


def test_bounds_are_inclusive_and_checked_for_fixed_parameters():
    pars = Parameters().load(io.StringIO(REFERENCE_LAYOUT_JSON))
    p = pars["sigma_max"]
    assert p.evaluate_lnprior(0.0) == 0 and p.evaluate_lnprior(-1e-300) == -np.inf
    assert pars["f_back"].evaluate_lnprior(1.0) == pytest.approx(np.log(2.0))
    assert pars["f_back"].evaluate_lnprior(1.0000001) == -np.inf
    vals = np.array([[0.0, 0.0, 1.0, 0.0], [0.0, -1.0, 1.0, 0.5], [0.0, 2.0, 1.0, 1.0], [0.0, 2.0, 1.0, 1.5]])
    res = pars.resolve_batch(vals)
    assert np.array_equal(res["v_maxy"], 2.0 * vals[:, 2])          # constraint evaluated on arrays
    lp = pars.lnprior_batch(res)
    assert np.array_equal(np.isfinite(lp), [True, False, True, False])
    assert lp[2] == pytest.approx(np.log(2.0))
    pars["dec_center"].set(value=-26.0)
    pars["dec_center"].max = -30.0                                   # a fixed value outside its bounds ...
    assert not np.isfinite(pars.lnprior_batch(pars.resolve_batch(vals))).any()   # ... rejects everything (runner.py:207-214)
    with pytest.raises(ValueError):
        Parameter("x", min=1.0, max=1.0)
    with pytest.raises(KeyError):
        Parameters().add("not a name", value=1.0)


def test_expressions_cannot_reach_the_interpreter():
    """`initials` / `lnprior` / `expr` strings come from parameter files; the reference evaluates them with asteval, which
    blocks dunder attributes, lambdas, comprehensions and imports.  The same escapes must fail here -- at parse time, before
    anything is evaluated -- while the expressions the reference's configs and scripts use keep working."""
    from mcmc_dynamics_amd.parameter import ExpressionError
    pars = ConstantFit.default_parameters()
    escape = "[c for c in ().__class__.__base__.__subclasses__() if c.__name__=='Popen']"
    for bad in (escape, "().__class__", "rng.__class__.__mro__", "(lambda: 1)()", "__import__('os').system('true')",
                "getattr(rng, 'bit_generator')", "rng._bit_generator", "norm.__init__.__globals__", "f'{n}'",
                "rng.normal(size=n).__array_interface__", "{1: 2}", "[x for x in (1, 2)]", "open('/etc/passwd')",
                # public attributes that lead out of the arithmetic (raw function pointers of the bit generator),
                # unbounded integer powers, memory bombs
                "rng.bit_generator.ctypes.next_double(12345)", "rng.bit_generator.state", "rng.bit_generator.cffi",
                "norm.dist", "uniform.logpdf.__func__", "9**9**9", "(9**9)**(9**9)", "[0.0] * 99999999999", "(1, 2) * 99999999999"):
        with pytest.raises(ExpressionError):
            pars["v_sys"].initials = bad
            pars["v_sys"].evaluate_initials(4)
        with pytest.raises(ExpressionError):
            pars["v_sys"].lnprior = bad
            pars["v_sys"].evaluate_lnprior(0.5)
    # what the reference's parameter files and scripts contain (config/*.json, bin/run.py:490, bin/run_test_5139_*.py)
    pars = ConstantFit.default_parameters()
    pars["sigma_max"].set(value=7.0)
    for good, par in (("rng.lognormal(mean=2.30, sigma=0.5, size=n)", "sigma_max"), ("rng.normal(loc=0, scale=2, size=n)", "v_maxx"),
                      ("300*rng.beta(a=2, b=5, size=n)", "v_maxy"), ("rng.uniform(-0.5, 0.5, size=n)", "v_sys"),
                      ("sigma_max*rng.lognormal(size=n)", "v_sys")):
        pars[par].initials = good
        out = pars[par].evaluate_initials(5)
        assert out.shape == (5,) and np.all(np.isfinite(out))
    pars["v_sys"].lnprior = "norm.logpdf(val, loc=1.0, scale=2.0)"
    assert pars["v_sys"].evaluate_lnprior(0.5) == pytest.approx(-1.643335713764618)
    pars["v_sys"].lnprior = "0.0 if -5 <= val <= 5 else -inf"
    assert pars["v_sys"].evaluate_lnprior(9.0) == -np.inf
    pars["v_sys"].lnprior = "-0.5 * (val - 3)**2 / 2**2 + uniform.logpdf(val, loc=-10, scale=20) - log(2**0.5)"
    assert pars["v_sys"].evaluate_lnprior(1.0) == pytest.approx(-0.5 - np.log(20.0) - 0.5 * np.log(2.0))
    pars["v_sys"].initials = "clip(rng.normal(size=n), -1, 1).real * 2**1"
    assert pars["v_sys"].evaluate_initials(6).shape == (6,)


def test_unit_handling():
    p = Parameter("a", value=30.0, unit="arcsec", min=0.0)

    class Q(object):                                       # duck-typed astropy Quantity
        def __init__(self, value, unit):
            self.value, self.unit = value, unit

    p.set(value=Q(1.0, "arcmin"))
    assert p.value == pytest.approx(60.0)
    assert p.evaluate_lnprior(Q(-1.0, "arcmin")) == -np.inf
    with pytest.raises(IOError):
        p.set(value=Q(1.0, "km/s"))


# ------------------------------------------------------------------------------------------ data + backgrounds
def test_datareader_bins_match_reference():
    g = load_golden("radial_bins")
    reader = DataReader({"ra": g["ra"], "dec": g["dec"], "v": g["v"], "verr": g["verr"]})
    r = reader.compute_distances(float(g["ra_center"]), float(g["dec_center"]))
    assert rel_err(r, g["r_arcmin"]) < 1e-12
    reader.make_radial_bins(float(g["ra_center"]), float(g["dec_center"]), nstars=200, dlogr=0.05)
    assert reader.data["bin"].dtype == np.int16
    assert np.array_equal(reader.data["bin"], g["bins_n200_d005"])
    reader.make_radial_bins(float(g["ra_center"]), float(g["dec_center"]), nstars=50, dlogr=0.1)
    assert np.array_equal(reader.data["bin"], g["bins_n50_d01"])
    sub = reader.fetch_radial_bin(3)
    assert sub.sample_size == int(np.sum(g["bins_n50_d01"] == 3)) and set(sub.data.columns) >= {"ra", "v", "bin"}
    assert reader.fetch_radial_bin(99) is None
    srt, offs = reader.sorted_by_bin()
    assert offs[0] == 0 and offs[-1] == reader.sample_size and np.all(np.diff(srt.data["bin"]) >= 0)


def test_background_models():
    g = load_golden("constant_bg_gaussian_fixed")
    bg = Gaussian(mean=float(g["bg_mean"]), sigma=float(g["bg_sigma"]))
    assert np.max(np.abs(bg(g["v"], g["verr"]) - g["lnlike_background"])) < 1e-12
    ss = SingleStars([-30.0, 10.0, 55.0, 80.0])                # evaluation is a device kernel: see test_gpu_kernels.py
    assert ss.n_stars == 4 and ss.v.dtype == np.float64
    with pytest.raises(ValueError):
        ss(g["v"][:5], g["verr"][:4])


# ------------------------------------------------------------------------------------------ Runner contract
def test_runner_constructor_contract():
    reader, c = small_reader()
    with pytest.raises(AssertionError):
        ConstantFit(reader, initials=3)                       # unknown kwargs (runner.py:56)
    with pytest.raises(AssertionError):
        ConstantFit({"v": [1.0]})                             # data must be a DataReader
    with pytest.raises(IOError):
        ConstantFit(DataReader({"v": c["v"], "verr": c["verr"]}))       # missing coordinates (runner.py:70-72)
    with pytest.raises(AssertionError):
        ConstantFitGB(DataReader({k: c[k] for k in ("ra", "dec", "v", "verr")}))   # missing density column
    pars = ConstantFit.default_parameters()
    del pars["v_maxy"]
    with pytest.raises(IOError):
        ConstantFit(reader, parameters=pars)                  # missing model parameter (runner.py:87-89)
    with pytest.raises(AssertionError):
        ConstantFit(reader, background="gaussian")            # must be a background instance
    with pytest.raises(KeyError):
        ConstantFit(DataReader({k: c[k] for k in ("ra", "dec", "v", "verr")}), background=Gaussian(20.0, 40.0))
    cf = ConstantFit(reader)
    assert cf.fitted_parameters == ["v_sys", "sigma_max", "v_maxx", "v_maxy", "ra_center", "dec_center"]
    cf.parameters["ra_center"].set(value=56.345, fixed=True)
    cf.parameters["dec_center"].set(value=-26.675, fixed=True)
    assert cf.fitted_parameters == ["v_sys", "sigma_max", "v_maxx", "v_maxy"] and cf.n_fitted_parameters == 4
    assert ConstantFitGB(reader).fitted_parameters[-3:] == ["v_back", "sigma_back", "f_back"]
    assert cf.lnprior([0.0, 0.0, 1.0, 1.0]) == 0 and cf.lnprior([0.0, -0.1, 1.0, 1.0]) == -np.inf
    assert cf.lnprob([0.0, -0.1, 1.0, 1.0]) == -np.inf       # prior failure: likelihood never evaluated (no GPU needed)
    assert cf.fetch_parameter_values([1.0, 2.0, 3.0, 4.0])["ra_center"] == 56.345
    with pytest.raises(AssertionError):
        cf.fetch_parameter_values([1.0, 2.0, 3.0])           # 'Not all parameters used.' (runner.py:178)
    init = cf.get_initials(16)
    assert init.shape == (16, 4) and np.all(init[:, 1] > 0)
    with pytest.raises(ValueError):
        cf(n_walkers=8, n_steps=2, n_threads=4)
    bad = init.copy()
    bad[3, 1] = -1.0
    with pytest.raises(ValueError, match="Invalid initial guesses for walker 3"):
        cf(n_walkers=16, n_steps=1, pos=bad)
    assert np.array_equal(cf.lnprob_batch(np.tile([0.0, -1.0, 0.0, 0.0], (4, 1))), np.full(4, -np.inf))
    v_los = cf.rotation_model(1.0, 2.0, -1.0, 56.345, -26.675)
    assert v_los.shape == (reader.sample_size,) and np.all(cf.dispersion_model(7.0) == 7.0)
    with pytest.raises(IOError):
        BinnedConstantFit(reader)                             # needs the bin column


def test_chain_statistics():
    reader, _ = small_reader()
    cf = ConstantFit(reader)
    cf.parameters["ra_center"].set(value=56.345, fixed=True)
    cf.parameters["dec_center"].set(value=-26.675, fixed=True)
    rng = np.random.default_rng(0)
    chain = rng.normal([1.0, 10.0, 3.0, 4.0], [0.1, 0.5, 0.2, 0.2], size=(20, 60, 4))
    best = cf.compute_bestfit_values(chain, n_burn=10)
    assert best.loc["median"]["sigma_max"] == pytest.approx(10.0, abs=0.1)
    assert best.loc["uperr"]["v_sys"] == pytest.approx(0.1, rel=0.3)
    tv = cf.compute_theta_vmax(chain, n_burn=10)
    assert tv.loc["median"]["v_max"] == pytest.approx(5.0, abs=0.1)
    assert tv.loc["median"]["theta_0"] == pytest.approx(np.arctan2(4.0, 3.0), abs=0.02)
    pars = cf.convert_to_parameters(chain, n_burn=10)
    assert pars["ra_center"].shape == (20 * 50,) and np.all(pars["ra_center"] == 56.345)


def test_chain_statistics_match_the_reference():
    """Runner.compute_percentiles / compute_bestfit_values / convert_to_parameters (runner.py:521-660),
    get_amplitude_and_angle (utils/coordinates/get_amplitude_and_angle.py:10-51) and ConstantFit.compute_theta_vmax
    (constant.py:156-214) against outputs of the reference for a fixed synthetic chain (tests/golden/chain_stats.npz,
    generated by oracle/make_golden.py round2): a burn-in to discard, a rotation axis scattering around +-pi (wrap), an
    axis in the first quadrant, and theta_0 given in place of v_maxx."""
    from mcmc_dynamics_amd.utils.coordinates import get_amplitude_and_angle
    g = load_golden("chain_stats")
    cf = ConstantFit(DataReader({k: g[k] for k in ("ra", "dec", "v", "verr")}))
    cf.parameters["ra_center"].set(value=float(g["ra_center"]), fixed=True)
    cf.parameters["dec_center"].set(value=float(g["dec_center"]), fixed=True)
    names = [str(n) for n in g["names"]]
    assert cf.fitted_parameters == names and list(cf.parameters) == [str(n) for n in g["all_names"]]
    chain, n_burn = g["chain"], int(g["n_burn"])
    assert np.array_equal(cf.compute_percentiles(chain, n_burn=n_burn), g["percentiles_default"])
    assert np.array_equal(cf.compute_percentiles(chain, n_burn=n_burn, pct=[2.5, 97.5]), g["percentiles_custom"])
    best = cf.compute_bestfit_values(chain, n_burn=n_burn)
    rows = np.array([[best.loc[r][n] for n in names] for r in ("median", "uperr", "loerr")])
    assert np.array_equal(rows, g["bestfit"])
    # as in the reference, the medians are written back into the parameters (runner.py:649)
    assert np.array_equal([cf.parameters[n].value for n in names], g["parameters_after_bestfit"])
    pars = cf.convert_to_parameters(chain, n_burn=n_burn)
    assert np.array_equal(np.stack([pars[str(n)] for n in g["all_names"]]), g["converted"])

    def table(res):
        return np.array([[res.loc[r][c] for c in ("v_max", "theta_0")] for r in ("median", "uperr", "loerr")])
    res, v_max, theta = get_amplitude_and_angle(pars, return_samples=True)
    np.testing.assert_allclose(table(res), g["amp_angle"], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(v_max, g["v_max_samples"], rtol=1e-14)
    np.testing.assert_allclose(theta, g["theta_samples"], rtol=1e-14, atol=1e-15)
    assert abs(abs(res.loc["median"]["theta_0"]) - np.pi) < 0.3 and np.ptp(theta) < np.pi      # the wrap was exercised
    np.testing.assert_allclose(table(cf.compute_theta_vmax(chain, n_burn=n_burn)), g["theta_vmax_method"], rtol=1e-14, atol=1e-15)
    pars_b = cf.convert_to_parameters(g["chain_b"], n_burn=0)
    res_b, v_max_b, theta_b = get_amplitude_and_angle(pars_b, return_samples=True)
    np.testing.assert_allclose(table(res_b), g["amp_angle_b"], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(v_max_b, g["v_max_samples_b"], rtol=1e-14)
    alt = {"theta_0": np.arctan2(pars_b["v_maxy"], pars_b["v_maxx"]), "v_maxy": pars_b["v_maxy"]}
    res_c, v_max_c, theta_c = get_amplitude_and_angle(alt, return_samples=True)
    np.testing.assert_allclose(table(res_c), g["amp_angle_c"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(v_max_c, g["v_max_samples_c"], rtol=1e-13)
    np.testing.assert_allclose(theta_c, g["theta_samples_c"], rtol=1e-13, atol=1e-15)


def test_checkpoint_format(tmp_path):
    """{prefix}_chain.pkl holds (W, steps, P), {prefix}_lnprob.pkl holds (W, steps) (runner.py:458-477);
    read_final_chain returns chain[:, -1, :] for restarts (runner.py:499-519)."""

    def log_prob(x):
        return -0.5 * np.sum(x ** 2, axis=1)

    s = EnsembleSampler(8, 2, log_prob, vectorize=True, seed=1)
    s.run_mcmc(np.random.default_rng(0).normal(size=(8, 2)), 5)
    prefix = str(tmp_path / "run")
    Runner.save_current_status(s, prefix=prefix)
    chain = Runner.read_chain(prefix + "_chain.pkl")
    assert chain.shape == (8, 5, 2) and Runner.read_chain(prefix + "_lnprob.pkl").shape == (8, 5)
    assert np.array_equal(Runner.read_final_chain(prefix + "_chain.pkl"), chain[:, -1, :])


def test_stretch_move_samples_a_gaussian():
    calls = []

    def log_prob(x):
        calls.append(len(x))
        return -0.5 * np.sum((x - 3.0) ** 2 / 4.0, axis=1)

    s = EnsembleSampler(32, 3, log_prob, vectorize=True, seed=11)
    pos, lnp, state = s.run_mcmc(np.random.default_rng(1).normal(3.0, 1.0, size=(32, 3)), 600)
    assert calls[0] == 32 and set(calls[1:]) == {16}          # one full call, then W/2 proposals twice per step
    assert s.chain.shape == (32, 600, 3) and s.lnprobability.shape == (32, 600) and s.iteration == 600
    flat = s.get_chain(discard=100, flat=True)
    assert np.all(np.abs(flat.mean(axis=0) - 3.0) < 0.25) and np.all(np.abs(flat.std(axis=0) - 2.0) < 0.3)
    assert 0.2 < s.acceptance_fraction.mean() < 0.9
    pos2, _, _ = s.run_mcmc(pos, 5, log_prob0=lnp, rstate0=state)
    assert s.iteration == 605 and pos2.shape == (32, 3)
    with pytest.raises(ValueError):
        EnsembleSampler(4, 3, log_prob)
    with pytest.raises(ValueError, match="NaN"):
        EnsembleSampler(8, 2, lambda x: np.full(len(x), np.nan), vectorize=True).run_mcmc(np.zeros((8, 2)), 1)


def test_batch_plan_matches_the_per_parameter_path():
    """The cached batch plan of lnprob_batch (flat bounds check + kernel table by index) against the generic
    resolve_batch / lnprior_batch / _kernel_table path, with a stand-in catalogue that echoes its table."""
    from mcmc_dynamics_amd.analysis import ModelFitGB

    class Echo(object):
        def __init__(self):
            self.tables = []

        def loglike(self, table):
            self.tables.append(np.array(table))
            return table.sum(axis=1)

    reader, c = small_reader()
    rng = np.random.default_rng(4)
    for cls, fix in ((ConstantFit, True), (ConstantFitGB, False), (ModelFitGB, True)):
        obj = cls(reader)
        if fix:
            obj.parameters["ra_center"].set(value=56.345, fixed=True)
            obj.parameters["dec_center"].set(value=-26.675, fixed=True)
        else:
            obj.parameters["ra_center"].set(value=56.345)
            obj.parameters["dec_center"].set(value=-26.675)
        if cls is ModelFitGB:
            obj.parameters["a"].set(unit="arcmin")            # a unit conversion in the kernel table (arcmin -> arcsec)
        names = obj.fitted_parameters
        vals = rng.normal(1.0, 2.0, size=(40, len(names)))
        for j, n in enumerate(names):
            if n == "ra_center":
                vals[:, j] = 56.345 + 0.01 * rng.normal(size=40)
            if n == "dec_center":
                vals[:, j] = -26.675 + 0.01 * rng.normal(size=40)
        vals[3, names.index("sigma_max")] = -0.5              # rejected rows
        vals[7, 0] = np.nan
        echo = Echo()
        obj._catalog, obj._catalog_key = echo, obj._catalog_spec()[0]
        obj._ensure_catalog = lambda echo=echo: echo
        fast = obj.lnprob_batch(vals)
        resolved = obj.parameters.resolve_batch(vals)
        lp = obj.parameters.lnprior_batch(resolved)
        ok = np.isfinite(lp)
        assert np.array_equal(np.isfinite(fast), ok) and not ok[3] and not ok[7]
        slow_table = obj._kernel_table(resolved)
        assert np.allclose(fast[ok], slow_table[ok].sum(axis=1) + lp[ok], rtol=1e-14, atol=0)
        assert echo.tables[0].shape == slow_table.shape and np.array_equal(echo.tables[0][ok], slow_table[ok])
        # changing a bound or fixing a parameter invalidates the plan
        obj.parameters["v_sys"].set(max=0.0)
        again = obj.lnprob_batch(vals)
        assert np.array_equal(np.isfinite(again), ok & (vals[:, 0] <= 0.0))


def test_lnprob_batch_direct_route_equals_the_general_one(monkeypatch):
    """`Runner.lnprob_batch` hands the (W, P) proposals straight to the kernel when every kernel column is a free
    parameter; verdicts (prior violations, NaN) and the table the kernel receives must equal those of the general route
    through the full (W, n_all) parameter table.  No GPU: the catalogue is a stub that records what it is given."""
    import mcmc_dynamics_amd.analysis.runner as runner_mod
    from mcmc_dynamics_amd.analysis import ConstantFitGB

    class StubCatalog(object):
        seen = None

        def loglike(self, p):
            self.seen = np.array(p)
            return np.arange(len(p), dtype=np.float64)

        def close(self):
            pass

    monkeypatch.setattr(runner_mod.Runner, "_ensure_catalog", lambda self: self._catalog)
    cat = synthetic.make_catalog(500, config=3, background=True)
    names7 = ["v_sys", "sigma_max", "v_maxx", "v_maxy", "v_back", "sigma_back", "f_back"]
    for cls, names, cols, kw in ((ConstantFit, names7[:4], ("ra", "dec", "v", "verr", "pmember"), dict(background=Gaussian(20.0, 40.0))),
                                 (ConstantFitGB, names7, ("ra", "dec", "v", "verr", "density"), {})):
        fit = cls(DataReader({k: cat[k] for k in cols}), **kw)
        fit.parameters["ra_center"].set(value=synthetic.CENTER_RA_DEG, fixed=True)
        fit.parameters["dec_center"].set(value=synthetic.CENTER_DEC_DEG, fixed=True)
        fit._catalog = StubCatalog()
        plan = fit._plan()
        fit._catalog_key = plan.catalog_key
        assert plan.direct_cols is not None and plan.direct_identity
        pos = synthetic.make_walkers(64, names, cat["truth"], config=3)
        rng = np.random.default_rng(3)
        for trial in range(50):
            x = pos * (1.0 + 0.5 * rng.normal(size=pos.shape))
            x[rng.integers(0, 64, 4), 1] *= -1.0                  # sigma_max < 0
            if trial % 5 == 0:
                x[3, 0] = np.nan
            got = fit.lnprob_batch(x)
            full = plan.full(x)
            ok = plan.prior_ok(full)
            assert np.array_equal(np.isfinite(got), ok)
            if ok.any():
                if not ok.all():
                    full[~ok] = full[int(np.flatnonzero(ok)[0])]
                assert np.array_equal(fit._catalog.seen, plan.table(full))
        fit.parameters["v_sys"].set(value=0.0, fixed=True)       # a fixed kernel column: back to the general route
        assert fit._plan().direct_cols is None
        fit._catalog_key = fit._plan().catalog_key
        assert fit.lnprob_batch(pos[:, 1:]).shape == (64,)


def test_sampler_selection(monkeypatch, caplog):
    """Runner.SAMPLER: "auto" hands the loop to emcee whenever it can be imported, with lnprob_batch and
    vectorize=True (runner.py:403 is what it reproduces); without emcee the built-in sampler, with blocks of steps inside
    the library for box priors; "resident" / "builtin" / "emcee" force one; one INFO line names the driver."""
    import logging
    import sys
    import types
    g = load_golden("constant_fixed")
    cf = ConstantFit(DataReader({k: g[k] for k in ("ra", "dec", "v", "verr")}))
    cf.parameters["ra_center"].set(value=float(g["ra_center"]), fixed=True)
    cf.parameters["dec_center"].set(value=float(g["dec_center"]), fixed=True)
    made = []

    class FakeEnsembleSampler(object):
        def __init__(self, nwalkers, ndim, fn, vectorize=False):
            made.append((nwalkers, ndim, vectorize, fn))
            self._random = np.random.RandomState()

    monkeypatch.setitem(sys.modules, "emcee", types.SimpleNamespace(EnsembleSampler=FakeEnsembleSampler))
    caplog.set_level(logging.INFO, logger="mcmc_dynamics_amd")
    s = cf._make_sampler(16, seed=3)
    assert isinstance(s, FakeEnsembleSampler) and made[0][:3] == (16, 4, True) and made[0][3] == cf.lnprob_batch   # emcee drives
    assert "emcee.EnsembleSampler" in caplog.text
    cf.SAMPLER = "resident"
    s = cf._make_sampler(16, seed=3)
    assert isinstance(s, EnsembleSampler) and s.block_fn is not None and len(made) == 1
    assert "resident on the device" in caplog.text
    cf.SAMPLER = "emcee"
    assert isinstance(cf._make_sampler(16, seed=3), FakeEnsembleSampler) and len(made) == 2
    cf.SAMPLER = "builtin"
    s = cf._make_sampler(16)
    assert isinstance(s, EnsembleSampler) and s.block_fn is not None and len(made) == 2
    cf.parameters["sigma_max"].set(lnprior="-0.5 * (sigma_max - 10.0)**2")                  # not a box prior any more
    assert not cf._plan().simple and not cf.resident_ok()[0]
    s = cf._make_sampler(16)
    assert isinstance(s, EnsembleSampler) and s.block_fn is None                             # Python loop around lnprob_batch
    cf.SAMPLER = "resident"
    with pytest.raises(ValueError, match="not plain boxes"):
        cf._make_sampler(16)
    cf.SAMPLER = "auto"
    assert isinstance(cf._make_sampler(16), FakeEnsembleSampler) and len(made) == 3
    monkeypatch.setitem(sys.modules, "emcee", None)                                          # import emcee -> ImportError
    s = cf._make_sampler(16)
    assert isinstance(s, EnsembleSampler) and s.block_fn is None
    cf.parameters["sigma_max"].lnprior = None                                                # box priors again
    s = cf._make_sampler(16)
    assert isinstance(s, EnsembleSampler) and s.block_fn is not None                         # no emcee, box priors: resident
    cf.SAMPLER = "emcee"
    with pytest.raises(ImportError):
        cf._make_sampler(16)
    cf.SAMPLER = "nonsense"
    with pytest.raises(ValueError):
        cf._make_sampler(16)


def test_a_subclass_that_changes_the_posterior_is_never_run_inside_the_library(monkeypatch):
    """the library's block entry evaluates the built-in model; a Runner sub-class that overrides
    lnprob_batch / lnlike_batch / lnprior ... outside the package (the reference's Runner is meant to be sub-classed) must
    get the Python loop around ITS lnprob_batch, and the chain must reflect the override."""
    import sys
    monkeypatch.setitem(sys.modules, "emcee", None)
    g = load_golden("constant_fixed")
    data = DataReader({k: g[k] for k in ("ra", "dec", "v", "verr")})

    class Tilted(ConstantFit):
        calls = 0

        def lnprob_batch(self, values):
            Tilted.calls += 1
            values = np.asarray(values, dtype=np.float64)
            return -0.5 * np.sum((values - np.array([50.0, 3.0, -20.0, 20.0])) ** 2, axis=1)      # nothing like the data's posterior

    class Penalised(ConstantFit):
        def lnprior(self, values):
            return super(Penalised, self).lnprior(values) - 1.0

    for cls, method in ((Tilted, "lnprob_batch"), (Penalised, "lnprior")):
        fit = cls(data)
        fit.parameters["ra_center"].set(value=float(g["ra_center"]), fixed=True)
        fit.parameters["dec_center"].set(value=float(g["dec_center"]), fixed=True)
        ok, why = fit.resident_ok()
        assert not ok and method in why, why
        s = fit._make_sampler(16, seed=1)
        assert isinstance(s, EnsembleSampler) and s.block_fn is None
        fit.SAMPLER = "resident"
        with pytest.raises(ValueError, match="overrides the posterior"):
            fit._make_sampler(16)
    plain = ConstantFit(data)
    plain.parameters["ra_center"].set(value=float(g["ra_center"]), fixed=True)
    plain.parameters["dec_center"].set(value=float(g["dec_center"]), fixed=True)
    assert plain.resident_ok() == (True, "")
    # the chain of the overriding class follows the override (no GPU involved: its lnprob_batch never touches the catalogue)
    fit = Tilted(data)
    fit.parameters["ra_center"].set(value=float(g["ra_center"]), fixed=True)
    fit.parameters["dec_center"].set(value=float(g["dec_center"]), fixed=True)
    rng = np.random.default_rng(5)
    pos = np.array([50.0, 3.0, -20.0, 20.0]) + rng.normal(0, 1, size=(16, 4))
    sampler = fit(n_walkers=16, n_steps=300, pos=pos, prefix=None)
    assert Tilted.calls > 300
    flat = sampler.get_chain(discard=100, flat=True)
    assert np.all(np.abs(flat.mean(axis=0) - np.array([50.0, 3.0, -20.0, 20.0])) < 0.5)


def test_a_binned_subclass_that_changes_the_posterior_gets_the_numpy_loop():
    """The same rule for `BinnedConstantFit.__call__`: library blocks (host or device numbers) only for the package's own
    posterior; an override outside the package is driven through ITS lnprob_batch, with the device generator's numbers
    (`rng="device"`: `mcd_chain_numbers`, host code) -- no catalogue is ever built here."""
    from mcmc_dynamics_amd.analysis import BinnedConstantFit
    g = load_golden("radial_bins")
    reader = DataReader({k: g[k] for k in ("ra", "dec", "v", "verr")})
    reader.make_radial_bins(float(g["ra_center"]), float(g["dec_center"]), nstars=200, dlogr=0.05)

    class Tilted(BinnedConstantFit):
        calls = 0

        def lnprob_batch(self, values):
            Tilted.calls += 1
            values = np.asarray(values, dtype=np.float64)
            return -0.5 * np.sum((values - np.array([5.0, 3.0, -2.0, 2.0])) ** 2, axis=-1)

    fit = Tilted(reader)
    fit.parameters["ra_center"].set(value=float(g["ra_center"]), fixed=True)
    fit.parameters["dec_center"].set(value=float(g["dec_center"]), fixed=True)
    assert not fit.resident_ok()[0]
    B = fit.n_bins
    pos = np.array([5.0, 3.0, -2.0, 2.0]) + np.random.default_rng(2).normal(0, 1, size=(B, 16, 4))
    pos[..., 1] = np.abs(pos[..., 1])
    s = fit(n_walkers=16, n_steps=200, pos=pos, seed=3)
    assert s.rng == "device" and s.block_fn is None and s.seeded_block_fn is None and Tilted.calls >= 400
    assert fit._catalog is None
    flat = s.chain[:, :, 80:].reshape(-1, 4)
    assert np.all(np.abs(flat.mean(axis=0) - np.array([5.0, 3.0, -2.0, 2.0])) < 0.3)
    again = fit(n_walkers=16, n_steps=200, pos=pos, seed=3)
    assert np.array_equal(again.chain, s.chain)                        # reproducible from the seed
    s.close()
    again.close()
