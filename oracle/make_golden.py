#!/opt/conda/bin/python3.9
"""Generate golden input/output vectors from the REFERENCE ITSELF (test infrastructure).

Run in the build container only (the reference does not travel to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        PYTHONPATH=/root/repo/oracle/ref_shims:/root/reference \
        /opt/conda/bin/python3.9 -W ignore /root/repo/oracle/make_golden.py

Interpreter: conda python 3.9 / numpy 1.26.4 / astropy 4.3.1 (the only interpreter in the image
with astropy).  ``oracle/ref_shims`` supplies import stand-ins for the absent, non-arithmetic
packages (see its README); ``numpy.asscalar``/``numpy.alen`` are restored because astropy 4.3.1
still touches them.  Every number written below is ``float(obj.lnprob(values))`` (or another
method) of the unmodified reference classes imported from /root/reference.

Outputs: ``tests/golden/*.npz`` -- plain float64 arrays only (inputs + expected outputs), loadable
with ``numpy.load(allow_pickle=False)``.
"""
import importlib.util
import os
import sys

import numpy

numpy.asscalar = lambda a: a.item()     # removed in numpy 1.23; astropy 4.3.1 still references it
numpy.alen = len

import numpy as np                      # noqa: E402
from astropy import units as u          # noqa: E402

from mcmc_dynamics.analysis import (ConstantFit, ConstantFitGB, ModelFit, ModelFitGB,   # noqa: E402
                                    ModelFitConstantBackground)
from mcmc_dynamics.background import Gaussian, SingleStars                 # noqa: E402
from mcmc_dynamics.utils.files import DataReader                           # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")

_spec = importlib.util.spec_from_file_location("synthetic", os.path.join(REPO, "mcmc_dynamics_amd", "synthetic.py"))
synthetic = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synthetic)

KMS = u.km / u.s


def reader(cat, extra=()):
    cols = {"ra": cat["ra"] * u.deg, "dec": cat["dec"] * u.deg, "v": cat["v"] * KMS, "verr": cat["verr"] * KMS}
    for name in extra:
        cols[name] = cat[name]
    return DataReader(cols)


def fix_center(obj, ra_c, dec_c):
    # pattern of bin/run_tests.py:92-93
    obj.parameters["ra_center"].set(value=ra_c * u.deg, fixed=True)
    obj.parameters["dec_center"].set(value=dec_c * u.deg, fixed=True)


def lnprobs(obj, values):
    return np.array([float(obj.lnprob(np.array(row))) for row in values], dtype=np.float64)


def lnpriors(obj, values):
    return np.array([float(obj.lnprior(np.array(row))) for row in values], dtype=np.float64)


def save(name, **arrays):
    clean = {}
    for k, a in arrays.items():
        a = np.asarray(a)
        if a.dtype.kind in "US":
            a = a.astype("U32")
        clean[k] = a
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **clean)
    print("wrote", name, {k: tuple(v.shape) for k, v in clean.items()})


def walkers_with_rejections(cat_truth, names, n, config):
    """Walker ball plus deliberate prior violations / boundary values in the last rows."""
    pos = synthetic.make_walkers(n, names, cat_truth, config=config)
    j = names.index("sigma_max")
    pos[-1, j] = -1.0          # sigma < 0  -> -inf
    pos[-2, j] = 0.0           # sigma == 0 -> accepted (inclusive bound)
    if "f_back" in names:
        k = names.index("f_back")
        pos[-3, k] = 1.5       # f_back > 1 -> -inf
        pos[-4, k] = 0.0       # f_back == 0 -> accepted
        pos[-5, k] = 1.0       # f_back == 1 -> accepted
        pos[-6, names.index("sigma_back")] = -2.0
    if "ra_center" in names:
        pos[-7, names.index("dec_center")] = 91.0      # out of range
    kx, ky = names.index("v_maxx"), names.index("v_maxy")
    pos[0, kx] = 0.0
    pos[0, ky] = 0.0           # v_max = 0 -> theta_0 = arctan2(0, 0) = 0
    return pos


def golden_single_stars():
    """background.SingleStars (KDE over comparison stars, single_stars.py:42-77) and a ConstantFit that uses it."""
    ra_c, dec_c = synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG
    catb = synthetic.make_catalog(1200, config=3, background=True)
    rng = np.random.default_rng(424242)
    comp = np.concatenate([rng.normal(20.0, 40.0, 340), [250.0, -310.5]])        # two isolated comparison stars
    v, verr = catb["v"].copy(), catb["verr"].copy()
    v[3], verr[3] = 900.0, 0.05            # > 1e4 sigma from every comparison star: all but the nearest kernel underflow
    v[4], verr[4] = comp[17], 0.3          # exactly on a comparison star (largest exponent = 0)
    catb["v"], catb["verr"] = v, verr
    ss = SingleStars(comp * KMS)
    out = {}
    for tag, s_int in (("s0", 0.0), ("s2", 2.5)):
        res = ss(v * KMS, verr * KMS, sigma_int=s_int * KMS)
        out["lnlike_" + tag] = np.asarray(getattr(res, "value", res), dtype=np.float64)
    one = SingleStars(np.array([12.5]) * KMS)(v[:50] * KMS, verr[:50] * KMS)      # M = 1
    cf = ConstantFit(reader(catb, extra=("pmember",)), background=ss)
    fix_center(cf, ra_c, dec_c)
    names = list(cf.fitted_parameters)
    pos = walkers_with_rejections(catb["truth"], names, 12, config=3)
    save("single_stars", comp=comp, ra=catb["ra"], dec=catb["dec"], v=v, verr=verr, pmember=catb["pmember"],
         sigma_int_s0=0.0, sigma_int_s2=2.5, lnlike_m1=np.asarray(getattr(one, "value", one), dtype=np.float64),
         lnlike_background=np.asarray(cf.lnlike_background, dtype=np.float64),
         ra_center=ra_c, dec_center=dec_c, names=names, values=pos, lnprob=lnprobs(cf, pos), lnprior=lnpriors(cf, pos),
         **out)


def _table_rows(results, names):
    """(3, len(names)) float array of a reference results table: rows median / uperr / loerr."""
    out = np.empty((3, len(names)))
    for j, name in enumerate(names):
        for i, row in enumerate(("median", "uperr", "loerr")):
            x = results.loc[row][name]
            out[i, j] = float(getattr(x, "value", x))
    return out


def golden_round2():
    """Fixtures added in round 2 (the others are left untouched):

    * ``chain_stats``: a fixed synthetic chain through the reference's chain post-processing --
      ``Runner.compute_percentiles`` / ``compute_bestfit_values`` (runner.py:566-660), ``convert_to_parameters``
      (runner.py:521-564), ``get_amplitude_and_angle`` (utils/coordinates/get_amplitude_and_angle.py:10-51, also with the
      rotation axis near +-pi so that the wrap is exercised) and ``ConstantFit.compute_theta_vmax`` (constant.py:156-214);
    * ``model_fit_gb_membership``: ``ModelFitGB.calculate_membership_probabilities`` (model.py:458-510) for that kind of chain;
    * ``model_fit_bg_gaussian_{fixed,free}``: ``ModelFit`` with a fixed ``background=Gaussian`` (the pmember mixture of
      ``Runner._calculate_lnlike``, runner.py:272-286, applied to the Lynden-Bell / Plummer profiles of model.py:93-222).
    """
    from mcmc_dynamics.utils.coordinates import get_amplitude_and_angle
    ra_c, dec_c = synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG
    cat = synthetic.make_catalog(600, config=2)

    # ---- chain statistics on a ConstantFit with one fixed parameter in the middle of the list
    cf = ConstantFit(reader(cat))
    fix_center(cf, ra_c, dec_c)
    names = list(cf.fitted_parameters)
    rng = np.random.default_rng(20261004)
    n_walkers, n_steps, n_burn = 10, 14, 3
    centre_row = np.array([0.3, 9.5, -4.0, 0.4])                     # v_maxx < 0, v_maxy ~ 0: theta_0 scatters around +-pi
    chain = centre_row + rng.normal(size=(n_walkers, n_steps, len(names))) * np.array([0.4, 0.6, 0.8, 0.9])
    chain[:, :n_burn] += 25.0                                          # burn-in that must be discarded
    pct_default = cf.compute_percentiles(chain, n_burn=n_burn)
    pct_custom = cf.compute_percentiles(chain, n_burn=n_burn, pct=[2.5, 97.5])
    best = cf.compute_bestfit_values(chain, n_burn=n_burn)
    stored_after = np.array([float(getattr(cf.parameters[n].value, "value", cf.parameters[n].value)) for n in names])
    pars = cf.convert_to_parameters(chain, n_burn=n_burn)
    conv = np.stack([np.asarray(getattr(pars[k], "value", pars[k]), dtype=np.float64) for k in cf.parameters])
    res, v_max, theta = get_amplitude_and_angle({k: np.asarray(getattr(v, "value", v), dtype=np.float64)
                                                 for k, v in pars.items()}, return_samples=True)
    res2 = cf.compute_theta_vmax(chain, n_burn=n_burn)                  # return_samples=True raises KeyError('sigma') upstream
    # a second chain whose axis sits in the first quadrant (no wrap)
    chain_b = np.array([0.0, 8.0, 2.0, 3.0]) + rng.normal(size=(n_walkers, n_steps, len(names))) * 0.3
    pars_b = cf.convert_to_parameters(chain_b, n_burn=0)
    res_b, v_max_b, theta_b = get_amplitude_and_angle({k: np.asarray(getattr(v, "value", v), dtype=np.float64)
                                                       for k, v in pars_b.items()}, return_samples=True)
    # theta_0 given instead of v_maxx (second branch of get_amplitude_and_angle.py:14-15)
    alt = {"theta_0": theta_b + 0.0, "v_maxy": np.asarray(getattr(pars_b["v_maxy"], "value", pars_b["v_maxy"]), dtype=np.float64)}
    alt["theta_0"] = np.arctan2(alt["v_maxy"], np.asarray(getattr(pars_b["v_maxx"], "value", pars_b["v_maxx"]), dtype=np.float64))
    res_c, v_max_c, theta_c = get_amplitude_and_angle(dict(alt), return_samples=True)
    save("chain_stats", ra=cat["ra"], dec=cat["dec"], v=cat["v"], verr=cat["verr"], ra_center=ra_c, dec_center=dec_c,
         names=names, all_names=list(cf.parameters), chain=chain, n_burn=n_burn,
         percentiles_default=pct_default, percentiles_custom=pct_custom, bestfit=_table_rows(best, names),
         parameters_after_bestfit=stored_after, converted=conv,
         amp_angle=_table_rows(res, ["v_max", "theta_0"]), v_max_samples=np.asarray(v_max), theta_samples=np.asarray(theta),
         theta_vmax_method=_table_rows(res2, ["v_max", "theta_0"]),
         chain_b=chain_b, amp_angle_b=_table_rows(res_b, ["v_max", "theta_0"]), v_max_samples_b=np.asarray(v_max_b),
         theta_samples_b=np.asarray(theta_b),
         amp_angle_c=_table_rows(res_c, ["v_max", "theta_0"]), v_max_samples_c=np.asarray(v_max_c),
         theta_samples_c=np.asarray(theta_c))

    # ---- ModelFitGB.calculate_membership_probabilities
    catb = synthetic.make_catalog(900, config=3, background=True)
    truth_b = dict(catb["truth"], a=30.0, r_peak=60.0)
    for free in (False, True):
        mg = ModelFitGB(reader(catb, extra=("density",)))
        if free:
            mg.parameters["ra_center"].set(value=ra_c * u.deg)
            mg.parameters["dec_center"].set(value=dec_c * u.deg)
        else:
            fix_center(mg, ra_c, dec_c)
        names_g = list(mg.fitted_parameters)
        mid = np.array([truth_b[n] for n in names_g], dtype=np.float64)
        scale = np.array([(0.05 / 60.0) if n in ("ra_center", "dec_center") else (0.5 if truth_b[n] == 0.0 else 0.05 * abs(truth_b[n]))
                          for n in names_g])
        chain_g = mid + rng.normal(size=(8, 9, len(names_g))) * scale
        chain_g[..., names_g.index("f_back")] = np.clip(chain_g[..., names_g.index("f_back")], 0.01, 0.99)
        member = mg.calculate_membership_probabilities(chain_g, n_burn=2)
        best_g = mg.compute_bestfit_values(chain_g, n_burn=2)
        save("model_fit_gb_membership" + ("_free" if free else "_fixed"),
             ra=catb["ra"], dec=catb["dec"], v=catb["v"], verr=catb["verr"], density=catb["density"],
             ra_center=ra_c, dec_center=dec_c, names=names_g, chain=chain_g, n_burn=2,
             median=_table_rows(best_g, names_g)[0],
             membership=np.asarray(getattr(member, "value", member), dtype=np.float64))

    # ---- ModelFit + fixed Gaussian background
    bg = Gaussian(mean=20.0 * KMS, sigma=40.0 * KMS)
    for free in (False, True):
        mf = ModelFit(reader(catb, extra=("pmember",)), background=bg)
        if free:
            mf.parameters["ra_center"].set(value=ra_c * u.deg)
            mf.parameters["dec_center"].set(value=dec_c * u.deg)
        else:
            fix_center(mf, ra_c, dec_c)
        names_m = list(mf.fitted_parameters)
        rng_m = np.random.default_rng(31)
        pos = np.empty((16, len(names_m)))
        for j, nme in enumerate(names_m):
            t = truth_b[nme]
            g = rng_m.normal(size=16)
            pos[:, j] = t + (0.05 / 60.0) * g if nme in ("ra_center", "dec_center") else (0.5 * g if t == 0.0 else t * (1.0 + 0.05 * g))
        pos[-1, names_m.index("r_peak")] = -5.0             # rejected by the prior
        pos[-2, names_m.index("sigma_max")] = 0.0           # inclusive bound
        save("model_fit_bg_gaussian" + ("_free" if free else "_fixed"),
             ra=catb["ra"], dec=catb["dec"], v=catb["v"], verr=catb["verr"], pmember=catb["pmember"],
             bg_mean=20.0, bg_sigma=40.0, lnlike_background=np.asarray(mf.lnlike_background, dtype=np.float64),
             ra_center=ra_c, dec_center=dec_c, names=names_m, values=pos, lnprob=lnprobs(mf, pos), lnprior=lnpriors(mf, pos))
    with open(os.path.join(OUT, "PROVENANCE.txt"), "a") as f:
        import astropy
        f.write("round 2 fixtures (chain_stats, model_fit_gb_membership_*, model_fit_bg_gaussian_*): "
                "oracle/make_golden.py round2, python {0}, numpy {1}, astropy {2}\n".format(
                    sys.version.split()[0], np.__version__, astropy.__version__))


def main():
    os.makedirs(OUT, exist_ok=True)
    if sys.argv[1:] == ["round2"]:               # the fixtures added in round 2 only
        return golden_round2()
    if sys.argv[1:] == ["single_stars"]:          # regenerate one fixture without rewriting the others
        return golden_single_stars()
    golden_single_stars()
    golden_round2()
    ra_c, dec_c = synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG

    # ---------------------------------------------------------------- ConstantFit, fixed centre
    cat = synthetic.make_catalog(1500, config=2)
    cat["ra"][7], cat["dec"][7] = ra_c, dec_c            # a star exactly on the centre (theta = arctan2(0,0) = 0)
    cf = ConstantFit(reader(cat))
    fix_center(cf, ra_c, dec_c)
    names = list(cf.fitted_parameters)
    pos = walkers_with_rejections(cat["truth"], names, 24, config=2)
    save("constant_fixed", ra=cat["ra"], dec=cat["dec"], v=cat["v"], verr=cat["verr"],
         ra_center=ra_c, dec_center=dec_c, names=names, values=pos,
         lnprob=lnprobs(cf, pos), lnprior=lnpriors(cf, pos))

    # ---------------------------------------------------------------- ConstantFit, free centre
    cf = ConstantFit(reader(cat))
    cf.parameters["ra_center"].set(value=ra_c * u.deg)
    cf.parameters["dec_center"].set(value=dec_c * u.deg)
    names = list(cf.fitted_parameters)
    pos = walkers_with_rejections(cat["truth"], names, 24, config=2)
    save("constant_free", ra=cat["ra"], dec=cat["dec"], v=cat["v"], verr=cat["verr"],
         names=names, values=pos, lnprob=lnprobs(cf, pos), lnprior=lnpriors(cf, pos))

    # ---------------------------------------------------------------- ConstantFit + fixed Gaussian background
    catb = synthetic.make_catalog(2000, config=3, background=True)
    bg = Gaussian(mean=20.0 * KMS, sigma=40.0 * KMS)
    for free in (False, True):
        cf = ConstantFit(reader(catb, extra=("pmember",)), background=bg)
        if free:
            cf.parameters["ra_center"].set(value=ra_c * u.deg)
            cf.parameters["dec_center"].set(value=dec_c * u.deg)
        else:
            fix_center(cf, ra_c, dec_c)
        names = list(cf.fitted_parameters)
        pos = walkers_with_rejections(catb["truth"], names, 24, config=3)
        save("constant_bg_gaussian" + ("_free" if free else "_fixed"),
             ra=catb["ra"], dec=catb["dec"], v=catb["v"], verr=catb["verr"], pmember=catb["pmember"],
             bg_mean=20.0, bg_sigma=40.0, lnlike_background=np.asarray(cf.lnlike_background, dtype=np.float64),
             ra_center=ra_c, dec_center=dec_c, names=names, values=pos,
             lnprob=lnprobs(cf, pos), lnprior=lnpriors(cf, pos))

    # ---------------------------------------------------------------- ConstantFitGB (per-walker Gaussian background)
    for free in (False, True):
        gb = ConstantFitGB(reader(catb, extra=("density",)))
        if free:
            gb.parameters["ra_center"].set(value=ra_c * u.deg)
            gb.parameters["dec_center"].set(value=dec_c * u.deg)
        else:
            fix_center(gb, ra_c, dec_c)
        names = list(gb.fitted_parameters)
        pos = walkers_with_rejections(catb["truth"], names, 24, config=3)
        # membership probabilities (constant.py:366-374) at the first walker, via the reference's own helper
        pd = gb.fetch_parameter_values(pos[1])
        lc, lb, m = gb._calculate_lnlike_cluster_back(dict(pd))
        lc, lb, m = (np.asarray(getattr(x, "value", x), dtype=np.float64) for x in (lc, lb, m))
        member = m * np.exp(lc) / (m * np.exp(lc) + (1. - m) * np.exp(lb))
        save("constant_gb" + ("_free" if free else "_fixed"),
             ra=catb["ra"], dec=catb["dec"], v=catb["v"], verr=catb["verr"], density=catb["density"],
             ra_center=ra_c, dec_center=dec_c, names=names, values=pos,
             lnprob=lnprobs(gb, pos), lnprior=lnpriors(gb, pos),
             membership_row=1, membership=member, lnlike_cluster=lc, lnlike_back=lb, prior_m=m)

    # ---------------------------------------------------------------- radial bins + per-bin ConstantFit (C5 workflow)
    catr = synthetic.make_catalog(3000, config=5)
    dr = reader(catr)
    dr.make_radial_bins(ra_center=ra_c * u.deg, dec_center=dec_c * u.deg, nstars=200, dlogr=0.05)
    bins = np.asarray(dr.data["bin"], dtype=np.int64)
    r = np.asarray(dr.compute_distances(ra_c * u.deg, dec_c * u.deg).value, dtype=np.float64)
    n_bins = int(bins.max()) + 1
    names4 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]
    pos = synthetic.make_walkers(8, names4, catr["truth"], config=5)
    per_bin = np.empty((n_bins, pos.shape[0]))
    for b in range(n_bins):
        cfb = ConstantFit(dr.fetch_radial_bin(b))
        fix_center(cfb, ra_c, dec_c)
        per_bin[b] = lnprobs(cfb, pos)
    cfa = ConstantFit(reader(catr))
    fix_center(cfa, ra_c, dec_c)
    # a second binning rule with the defaults of run_tests.py:75 to pin the greedy loop + tail merge
    dr2 = reader(catr)
    dr2.make_radial_bins(ra_center=ra_c * u.deg, dec_center=dec_c * u.deg, nstars=50, dlogr=0.1)
    save("radial_bins", ra=catr["ra"], dec=catr["dec"], v=catr["v"], verr=catr["verr"],
         ra_center=ra_c, dec_center=dec_c, r_arcmin=r, bins_n200_d005=bins,
         bins_n50_d01=np.asarray(dr2.data["bin"], dtype=np.int64),
         names=names4, values=pos, lnprob_per_bin=per_bin, lnprob_all=lnprobs(cfa, pos))

    # ---------------------------------------------------------------- example/data/test.csv (C1 plumbing input)
    raw = np.loadtxt("/root/reference/example/data/test.csv", delimiter=",")
    cate = synthetic.example_polar_to_catalog(raw[0], raw[1], raw[2], raw[3], ra_center_deg=180.0)
    cf = ConstantFit(reader(cate))
    fix_center(cf, 180.0, 0.0)
    names = list(cf.fitted_parameters)
    truth_e = {"v_sys": 10.0, "sigma_max": 15.0, "v_maxx": 1.0, "v_maxy": -1.0}
    pos = synthetic.make_walkers(32, names, truth_e, config=1)
    # theta as the reference's own rotation model sees it after the polar -> (ra, dec) adapter
    from mcmc_dynamics.utils.coordinates import calc_xy_offset
    dx, dy = calc_xy_offset(cate["ra"] * u.deg, cate["dec"] * u.deg, 180.0 * u.deg, 0.0 * u.deg)
    save("example_catalog", r=raw[0], theta=raw[1], v=raw[2], verr=raw[3],
         ra=cate["ra"], dec=cate["dec"], ra_center=180.0, dec_center=0.0,
         dx=np.asarray(dx.value), dy=np.asarray(dy.value),
         names=names, values=pos, lnprob=lnprobs(cf, pos))

    # ---------------------------------------------------------------- ModelFit ("next" row 1), fixed and free centre
    for free in (False, True):
        mf = ModelFit(reader(cat))
        if free:
            mf.parameters["ra_center"].set(value=ra_c * u.deg)
            mf.parameters["dec_center"].set(value=dec_c * u.deg)
        else:
            fix_center(mf, ra_c, dec_c)
        names = list(mf.fitted_parameters)
        truth_m = dict(cat["truth"], a=30.0, r_peak=60.0)
        rng = np.random.default_rng(11)
        pos = np.empty((16, len(names)))
        for j, nme in enumerate(names):
            t = truth_m[nme]
            g = rng.normal(size=16)
            if nme in ("ra_center", "dec_center"):
                pos[:, j] = t + (0.05 / 60.0) * g
            elif t == 0.0:
                pos[:, j] = 0.5 * g
            else:
                pos[:, j] = t * (1.0 + 0.05 * g)
        pos[-1, names.index("a")] = -3.0       # a < 0 -> -inf
        save("model_fit" + ("_free" if free else "_fixed"),
             ra=cat["ra"], dec=cat["dec"], v=cat["v"], verr=cat["verr"],
             ra_center=ra_c, dec_center=dec_c, names=names, values=pos, lnprob=lnprobs(mf, pos))

    # ---------------------------------------------------------------- ModelFitGB / ModelFitConstantBackground
    def model_walkers(names, truth, n, seed):
        rng = np.random.default_rng(seed)
        pos = np.empty((n, len(names)))
        for j, nme in enumerate(names):
            t = truth[nme]
            g = rng.normal(size=n)
            if nme in ("ra_center", "dec_center"):
                pos[:, j] = t + (0.05 / 60.0) * g
            elif t == 0.0:
                pos[:, j] = 0.5 * g
            else:
                pos[:, j] = t * (1.0 + 0.05 * g)
        if "f_back" in names:
            pos[:, names.index("f_back")] = np.clip(pos[:, names.index("f_back")], 0.0, 1.0)
            pos[-2, names.index("f_back")] = 0.0
            pos[-3, names.index("f_back")] = 1.2          # rejected
        pos[-1, names.index("r_peak")] = -5.0             # rejected
        return pos

    truth_b = dict(catb["truth"], a=30.0, r_peak=60.0)
    for free in (False, True):
        mg = ModelFitGB(reader(catb, extra=("density",)))
        if free:
            mg.parameters["ra_center"].set(value=ra_c * u.deg)
            mg.parameters["dec_center"].set(value=dec_c * u.deg)
        else:
            fix_center(mg, ra_c, dec_c)
        names = list(mg.fitted_parameters)
        pos = model_walkers(names, truth_b, 16, 21)
        save("model_fit_gb" + ("_free" if free else "_fixed"),
             ra=catb["ra"], dec=catb["dec"], v=catb["v"], verr=catb["verr"], density=catb["density"],
             ra_center=ra_c, dec_center=dec_c, names=names, values=pos, lnprob=lnprobs(mg, pos), lnprior=lnpriors(mg, pos))

        from mcmc_dynamics.parameter import Parameters
        pars = Parameters().load(ModelFitConstantBackground.parameters_file)
        del pars["v_back"]
        del pars["sigma_back"]
        mc = ModelFitConstantBackground(reader(catb, extra=("density",)), background=bg, parameters=pars)
        if free:
            mc.parameters["ra_center"].set(value=ra_c * u.deg)
            mc.parameters["dec_center"].set(value=dec_c * u.deg)
        else:
            fix_center(mc, ra_c, dec_c)
        names = list(mc.fitted_parameters)
        pos = model_walkers(names, truth_b, 16, 22)
        per_star = np.asarray(getattr(mc.lnlike(pos[1], no_sum=True), "value", mc.lnlike(pos[1], no_sum=True)), dtype=np.float64)
        save("model_fit_cb" + ("_free" if free else "_fixed"),
             ra=catb["ra"], dec=catb["dec"], v=catb["v"], verr=catb["verr"], density=catb["density"],
             bg_mean=20.0, bg_sigma=40.0, lnlike_background=np.asarray(mc.lnlike_background, dtype=np.float64),
             ra_center=ra_c, dec_center=dec_c, names=names, values=pos, lnprob=lnprobs(mc, pos), lnprior=lnpriors(mc, pos),
             no_sum_row=1, lnlike_no_sum=per_star)

    # ---------------------------------------------------------------- interpreter record
    import astropy
    with open(os.path.join(OUT, "PROVENANCE.txt"), "w") as f:
        f.write("generated by oracle/make_golden.py from the unmodified reference at /root/reference\n")
        f.write("python {0}\nnumpy {1}\nastropy {2}\n".format(sys.version.split()[0], np.__version__, astropy.__version__))


if __name__ == "__main__":
    main()
