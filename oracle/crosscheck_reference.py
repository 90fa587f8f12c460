#!/opt/conda/bin/python3.9
"""Randomised cross-check of the CPU oracle (oracle/lnprob_numpy.py) against the REFERENCE ITSELF (test infrastructure).

Run in the build container only, like make_golden.py (the reference does not travel to the GPU box):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        PYTHONPATH=/root/repo/oracle/ref_shims:/root/reference \
        /opt/conda/bin/python3.9 -W ignore /root/repo/oracle/crosscheck_reference.py [n_cases]

For every case a random catalogue (size, velocity scale, errors, outliers, density, prior weights) and random parameter
rows inside the priors are drawn; `float(obj.lnprob(row))` of the unmodified reference classes is compared with the
oracle's faithful functions.  Prints the largest relative deviation per class; exit status 1 above 1e-13.
"""
import importlib.util
import os
import sys

import numpy

numpy.asscalar = lambda a: a.item()
numpy.alen = len

import numpy as np                      # noqa: E402
from astropy import units as u          # noqa: E402

from mcmc_dynamics.analysis import (ConstantFit, ConstantFitGB, ModelFit, ModelFitGB,   # noqa: E402
                                    ModelFitConstantBackground)
from mcmc_dynamics.background import Gaussian, SingleStars                 # noqa: E402
from mcmc_dynamics.parameter import Parameters                             # noqa: E402
from mcmc_dynamics.utils.files import DataReader                           # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


oracle = _load("lnprob_numpy", os.path.join(REPO, "oracle", "lnprob_numpy.py"))
KMS = u.km / u.s
RA_C, DEC_C = 56.345, -26.675


def random_catalogue(rng):
    n = int(rng.integers(5, 400))
    scale_v = 10.0 ** rng.uniform(0, 2.5)
    sep = np.abs(rng.normal(0, 2.0 / 60.0, n))
    th = rng.uniform(-np.pi, np.pi, n)
    cat = {"ra": RA_C + sep * np.cos(th) / np.cos(np.radians(DEC_C)), "dec": DEC_C + sep * np.sin(th),
           "v": rng.normal(0, scale_v, n), "verr": 10.0 ** rng.uniform(-2, 1.5) * rng.lognormal(0, 0.7, n)}
    cat["v"][: min(2, n)] *= 20.0
    cat["density"] = np.clip(rng.random(n), 0.02, 1.0)
    cat["pmember"] = np.clip(rng.random(n) * 1.1 - 0.05, 0.0, 1.0)
    return cat, scale_v


def reader(cat, extra=()):
    cols = {"ra": cat["ra"] * u.deg, "dec": cat["dec"] * u.deg, "v": cat["v"] * KMS, "verr": cat["verr"] * KMS}
    for k in extra:
        cols[k] = cat[k]
    return DataReader(cols)


def fix(obj, free):
    if free:
        obj.parameters["ra_center"].set(value=RA_C * u.deg)
        obj.parameters["dec_center"].set(value=DEC_C * u.deg)
    else:
        obj.parameters["ra_center"].set(value=RA_C * u.deg, fixed=True)
        obj.parameters["dec_center"].set(value=DEC_C * u.deg, fixed=True)


def draw(rng, names, scale_v):
    row = []
    for nme in names:
        if nme in ("v_sys", "v_maxx", "v_maxy", "v_back"):
            row.append(rng.normal(0, scale_v))
        elif nme in ("sigma_max", "sigma_back"):
            row.append(scale_v * 10.0 ** rng.uniform(-1.5, 0.7))
        elif nme in ("a", "r_peak"):
            row.append(10.0 ** rng.uniform(0, 2.5))
        elif nme == "f_back":
            row.append(rng.random())
        elif nme == "ra_center":
            row.append(RA_C + rng.normal(0, 0.005))
        elif nme == "dec_center":
            row.append(DEC_C + rng.normal(0, 0.005))
        else:
            raise KeyError(nme)
    return np.array(row)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(20261004)
    worst = {}

    def note(tag, got, want):
        if np.isfinite(want):
            err = abs(got - want) / max(abs(want), 1e-300)
        else:
            err = 0.0 if got == want else np.inf
        worst[tag] = max(worst.get(tag, 0.0), err)

    for case in range(n_cases):
        cat, sv = random_catalogue(rng)
        free = bool(case % 2)
        centre = lambda p: (p.get("ra_center", RA_C), p.get("dec_center", DEC_C))    # noqa: E731

        cf = ConstantFit(reader(cat)); fix(cf, free)
        names = list(cf.fitted_parameters)
        for _ in range(3):
            row = draw(rng, names, sv); p = dict(zip(names, row))
            note("ConstantFit", oracle.faithful_constant_lnlike(cat, p["v_sys"], p["sigma_max"], p["v_maxx"], p["v_maxy"], *centre(p)),
                 float(cf.lnprob(row)))

        mean_b, sig_b = rng.normal(0, sv), 3 * sv
        cb = ConstantFit(reader(cat, ("pmember",)), background=Gaussian(mean=mean_b * KMS, sigma=sig_b * KMS)); fix(cb, free)
        lnbg = oracle.gaussian_background(cat["v"], cat["verr"], mean_b, sig_b)
        for _ in range(3):
            row = draw(rng, names, sv); p = dict(zip(names, row))
            note("ConstantFit+Gaussian", oracle.faithful_constant_lnlike(cat, p["v_sys"], p["sigma_max"], p["v_maxx"], p["v_maxy"],
                                                                          *centre(p), lnbg, cat["pmember"]), float(cb.lnprob(row)))

        comp = rng.normal(mean_b, sig_b, int(rng.integers(1, 200)))
        ss = SingleStars(comp * KMS)
        res = ss(cat["v"] * KMS, cat["verr"] * KMS)
        ref_kde = np.asarray(getattr(res, "value", res), dtype=np.float64)
        mine = oracle.single_stars_background(comp, cat["v"], cat["verr"])
        worst["SingleStars"] = max(worst.get("SingleStars", 0.0), float(np.max(np.abs(mine - ref_kde) / np.maximum(1.0, np.abs(ref_kde)))))

        gb = ConstantFitGB(reader(cat, ("density",))); fix(gb, free)
        gnames = list(gb.fitted_parameters)
        for _ in range(3):
            row = draw(rng, gnames, sv); p = dict(zip(gnames, row))
            note("ConstantFitGB", oracle.faithful_constant_gb_lnlike(cat, p["v_sys"], p["sigma_max"], p["v_maxx"], p["v_maxy"], *centre(p),
                                                                     p["v_back"], p["sigma_back"], p["f_back"]), float(gb.lnprob(row)))

        mf = ModelFit(reader(cat)); fix(mf, free)
        mnames = list(mf.fitted_parameters)
        for _ in range(3):
            row = draw(rng, mnames, sv); p = dict(zip(mnames, row))
            note("ModelFit", oracle.faithful_model_lnlike(cat, p["v_sys"], p["sigma_max"], p["a"], p["v_maxx"], p["v_maxy"], p["r_peak"],
                                                          *centre(p)), float(mf.lnprob(row)))

        mg = ModelFitGB(reader(cat, ("density",))); fix(mg, free)
        mgn = list(mg.fitted_parameters)
        for _ in range(2):
            row = draw(rng, mgn, sv); p = dict(zip(mgn, row))
            note("ModelFitGB", oracle.faithful_model_gb_lnlike(cat, p["v_sys"], p["sigma_max"], p["a"], p["v_maxx"], p["v_maxy"], p["r_peak"],
                                                               *centre(p), p["v_back"], p["sigma_back"], p["f_back"]), float(mg.lnprob(row)))

        pars = Parameters().load(ModelFitConstantBackground.parameters_file)
        del pars["v_back"]
        del pars["sigma_back"]
        mc = ModelFitConstantBackground(reader(cat, ("density",)), background=Gaussian(mean=mean_b * KMS, sigma=sig_b * KMS), parameters=pars)
        fix(mc, free)
        mcn = list(mc.fitted_parameters)
        for _ in range(2):
            row = draw(rng, mcn, sv); p = dict(zip(mcn, row))
            note("ModelFitConstantBackground",
                 oracle.faithful_model_cb_lnlike(cat, p["v_sys"], p["sigma_max"], p["a"], p["v_maxx"], p["v_maxy"], p["r_peak"], *centre(p),
                                                 p["f_back"], lnbg), float(mc.lnprob(row)))

    bad = False
    for tag in sorted(worst):
        print("{0:28s} max relative deviation {1:.2e}".format(tag, worst[tag]))
        bad = bad or not worst[tag] <= 1e-13
    print("cases", n_cases, "numpy", np.__version__)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
