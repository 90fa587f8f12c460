"""``ModelFit`` / ``ModelFitGB`` / ``ModelFitConstantBackground``: analytic radial profiles
(reference: analysis/model.py).

    v_los(r, theta) = v_sys + 2 (v_max / r_peak) x_pa / (1 + (r / r_peak)^2),  x_pa = r sin(theta - theta_0)   (Lynden-Bell 1967)
    sigma_los(r)    = sigma_max / (1 + r^2 / a^2)^(1/4)                                                         (Plummer 1911)

with ``v_max = hypot(v_maxx, v_maxy)``, ``theta_0 = arctan2(v_maxy, v_maxx)``.  Units trap of the reference:
offsets r come out of ``calc_xy_offset`` in arcmin while ``a`` and ``r_peak`` are in arcsec
(config/model.json:8,13); the kernels work in arcsec throughout.  The same Gaussian / mixture likelihood as
the constant model (``Runner._calculate_lnlike``), so the same kernel skeleton with a different
(v_los, sigma_los) per term.
"""
import logging

import numpy as np

from .. import _native, units
from ..background import SingleStars
from ..parameter import Parameter, Parameters
from ..utils.coordinates import calc_xy_offset, get_amplitude_and_angle
from ..utils.data_reader import ColumnTable
from .runner import Runner

logger = logging.getLogger(__name__)
_INF = np.inf

_ROW = {
    "v_sys": ("v_sys", "km/s", -_INF, _INF, r"$v_{\rm sys}$", "rng.normal(size=n)"),
    "sigma_max": ("sigma_max", "km/s", 0.0, _INF, r"$\sigma_{\rm max}$", "rng.lognormal(size=n)"),
    "a": ("a", "arcsec", 0.0, _INF, r"$a$", "rng.lognormal(size=n)"),
    "v_maxx": ("v_maxx", "km/s", -_INF, _INF, r"$v_{\rm max,\,x}$", "rng.normal(size=n)"),
    "v_maxy": ("v_maxy", "km/s", -_INF, _INF, r"$v_{\rm max,\,y}$", "rng.normal(size=n)"),
    "r_peak": ("r_peak", "arcsec", 0.0, _INF, r"$r_{\rm peak}$", "rng.lognormal(size=n)"),
    "ra_center": ("ra_center", "deg", 0.0, 360.0, r"$\alpha_{\rm c}$", None),
    "dec_center": ("dec_center", "deg", -90.0, 90.0, r"$\delta_{\rm c}$", None),
    "v_back": ("v_back", "km/s", -_INF, _INF, r"$v_{\rm back}$", "rng.normal(size=n)"),
    "sigma_back": ("sigma_back", "km/s", 0.0, _INF, r"$\sigma_{\rm back}$", "rng.lognormal(size=n)"),
    "f_back": ("f_back", None, 0.0, 1.0, r"$f_{\rm back}$", "rng.uniform(size=n)"),
}
# iteration order of the reference's parameter files = order of the sampler's free-parameter vector
_MODEL_ORDER = ("v_sys", "sigma_max", "a", "v_maxx", "ra_center", "dec_center", "v_maxy", "r_peak")   # config/model.json:6-13
_MODEL_BG_ORDER = ("v_sys", "sigma_max", "a", "v_maxx", "v_maxy", "r_peak", "ra_center", "dec_center",
                   "v_back", "sigma_back", "f_back")                                              # model_with_background.json:6-16


def _build(order):
    pars = Parameters()
    for name in order:
        n, unit, lo, hi, label, initials = _ROW[name]
        pars.add(Parameter(n, unit=unit, min=lo, max=hi, label=label, initials=initials))
    return pars


class ModelFit(Runner):
    MODEL_PARAMETERS = ["v_sys", "v_maxx", "v_maxy", "r_peak", "sigma_max", "a", "ra_center", "dec_center"]
    OBSERVABLES = {"v": "km/s", "verr": "km/s", "ra": "deg", "dec": "deg"}

    _default_order = _MODEL_ORDER
    _model_id = _native.MODEL_PROFILE
    _KERNEL_HEAD = (("v_sys", "km/s"), ("sigma_max", "km/s"), ("a", "arcsec"), ("v_maxx", "km/s"), ("v_maxy", "km/s"),
                    ("r_peak", "arcsec"))

    def __init__(self, data, parameters=None, **kwargs):
        self.ra = None
        self.dec = None
        if parameters is None:
            parameters = self.default_parameters()
        super(ModelFit, self).__init__(data=data, parameters=parameters, **kwargs)

    @classmethod
    def default_parameters(cls):
        if cls.parameters_file is not None:
            return Parameters().load(cls.parameters_file)
        return _build(cls._default_order)

    # ------------------------------------------------------------------ host-side model functions
    def _offsets_arcsec(self, ra_center, dec_center):
        dx, dy = calc_xy_offset(ra=self.ra, dec=self.dec, ra_center=ra_center, dec_center=dec_center)
        return 60.0 * dx, 60.0 * dy

    def dispersion_model(self, sigma_max, ra_center, dec_center, a=1, **kwargs):
        """sigma_los = sigma_max / (1 + r^2 / a^2)^0.25 at every data point (model.py:93-127); a in arcsec."""
        if kwargs:
            raise IOError('Unknown keyword argument(s) "{0}" for method {1}.dispersion_model.'.format(
                ", ".join(kwargs.keys()), self.__class__.__name__))
        dx, dy = self._offsets_arcsec(ra_center, dec_center)
        r2 = dx ** 2 + dy ** 2
        a = float(units.to_unit(a, "arcsec"))
        return sigma_max / (1. + r2 / a ** 2) ** 0.25

    def rotation_model(self, v_sys, v_maxx, v_maxy, ra_center, dec_center, r_peak=None, **kwargs):
        """v_los = v_sys + 2 (v_max / r_peak) x_pa / (1 + (r / r_peak)^2) at every data point
        (model.py:129-180); r_peak in arcsec, default median(r)."""
        if kwargs:
            raise IOError('Unknown keyword argument(s) "{0}" for method {1}.rotation_model.'.format(
                ", ".join(kwargs.keys()), self.__class__.__name__))
        dx, dy = self._offsets_arcsec(ra_center, dec_center)
        r = np.sqrt(dx ** 2 + dy ** 2)
        r_peak = np.median(r) if r_peak is None else float(units.to_unit(r_peak, "arcsec"))
        v_max = np.sqrt(v_maxx ** 2 + v_maxy ** 2)
        theta_0 = np.arctan2(v_maxy, v_maxx)
        x_pa = r * np.sin(np.arctan2(dy, dx) - theta_0)
        return v_sys + 2. * (v_max / r_peak) * x_pa / (1. + (r / r_peak) ** 2)

    def _catalog_model(self):
        # ModelFit.lnlike ends in Runner._calculate_lnlike (model.py:222 -> runner.py:240-286): with a fixed `background`
        # the pmember mixture applies to the profile models exactly as it does to ConstantFit
        if self.lnlike_background is not None:
            return _native.MODEL_PROFILE_BGFIXED, {"lnlike_bg": self.lnlike_background, "pmember": self.pmember}
        return self._model_id, {}

    def lnlike(self, values):
        """Log-likelihood of the data for one parameter vector (model.py:182-222), including the fixed background
        population when one was given (runner.py:272-286)."""
        return super(ModelFit, self).lnlike(values)

    # ------------------------------------------------------------------ post-processing
    def create_profiles(self, chains, n_burn, radii=None, filename=None):
        """Radial profiles of the rotation amplitude and the dispersion with 1 / 3 sigma bands from a chain
        (model.py:224-315).  ``radii`` in arcsec (default: 50 log-spaced points from 0.1 to ~316 arcsec)."""
        pars = self.convert_to_parameters(chains, n_burn)

        def col(name, unit):
            f = units.conversion_factor(self.parameters[name].unit, unit) if self.parameters[name].unit else 1.0
            return pars[name] * f

        v_max = np.sqrt(col("v_maxx", "km/s") ** 2 + col("v_maxy", "km/s") ** 2)
        r_peak, sigma_max, a = col("r_peak", "arcsec"), col("sigma_max", "km/s"), col("a", "arcsec")
        radii = np.logspace(-1, 2.5, 50) if radii is None else np.atleast_1d(units.to_unit(radii, "arcsec"))
        rr = radii[:, np.newaxis]
        v_rot = 2. * (v_max / r_peak) * rr / (1. + (rr / r_peak) ** 2)
        sigma = sigma_max / (1. + rr ** 2 / a ** 2) ** 0.25
        pv = np.percentile(v_rot, [50, 16, 84, 0.15, 99.85], axis=-1)
        ps = np.percentile(sigma, [50, 16, 84, 0.15, 99.85], axis=-1)
        names = ("", "_lower_1s", "_upper_1s", "_lower_3s", "_upper_3s")
        table = {"r": radii}
        table.update({"v_rot" + n: pv[i] for i, n in enumerate(names)})
        table.update({"sigma" + n: ps[i] for i, n in enumerate(names)})
        profile = ColumnTable(table, units_=dict({"r": "arcsec"}, **{k: "km/s" for k in table if k != "r"}))
        if filename is not None:
            np.savetxt(filename, np.stack([profile[k] for k in profile.columns], axis=1), delimiter=",",
                       header=",".join(profile.columns), comments="")
        return profile

    def compute_theta_vmax(self, chain, n_burn, return_samples=False):
        """Position angle and amplitude of the rotation field from a chain (model.py:317-335)."""
        pars = self.convert_to_parameters(chain=chain, n_burn=n_burn)
        results, v_max, theta = get_amplitude_and_angle(pars, return_samples=return_samples)
        if results is None:
            logger.error("Could not recover paramaters of rotation field in %s.compute_theta_vmax().",
                         self.__class__.__name__)
            return None
        results.units["v_max"] = self.units["v_maxx"]
        if return_samples:
            return results, v_max, theta, pars["sigma_max"]
        return results


class ModelFitGB(ModelFit):
    """ModelFit plus a background component that is Gaussian in radial-velocity space with per-walker
    parameters (v_back, sigma_back, f_back) and the ``density`` membership prior (model.py:338-510)."""

    MODEL_PARAMETERS = ModelFit.MODEL_PARAMETERS + ["v_back", "sigma_back", "f_back"]
    OBSERVABLES = dict(ModelFit.OBSERVABLES, **{"density": None})

    _default_order = _MODEL_BG_ORDER
    _model_id = _native.MODEL_PROFILE_BGGAUSS
    _KERNEL_TAIL = (("v_back", "km/s"), ("sigma_back", "km/s"), ("f_back", None))

    def __init__(self, data, parameters=None, **kwargs):
        self.density = None
        background = kwargs.pop("background", None)
        if background is not None:
            logger.error("Class ConstantFitGB does not support additional background components.")
        super(ModelFitGB, self).__init__(data=data, parameters=parameters, **kwargs)

    def _catalog_model(self):
        return self._model_id, {"density": self.density}

    def lnlike(self, values):
        """Log-likelihood including the Gaussian background mixture (model.py:391-456)."""
        return super(ModelFitGB, self).lnlike(values)

    def calculate_membership_probabilities(self, chain, n_burn):
        """Posterior membership probability of every star at the chain's median parameters (model.py:458-510)."""
        bestfit = self.compute_bestfit_values(chain=chain, n_burn=n_burn)
        median = np.array([bestfit.loc["median"][name] for name in self.fitted_parameters])
        return self.membership_probabilities(median)

    def membership_probabilities(self, values):
        return self._per_star(values, "membership")


class ModelFitConstantBackground(ModelFit):
    """ModelFit plus a constant background: a fixed per-star background log-likelihood (from a
    ``background`` instance) mixed in with the ``density`` prior and ONE free parameter ``f_back``
    (model.py:513-687)."""

    MODEL_PARAMETERS = ModelFit.MODEL_PARAMETERS + ["f_back", ]
    OBSERVABLES = dict(ModelFit.OBSERVABLES, **{"density": None})

    # as in the reference, the default set is the with-background file, whose v_back / sigma_back are superfluous here
    _default_order = _MODEL_BG_ORDER
    _model_id = _native.MODEL_PROFILE_BGDENS
    _KERNEL_TAIL = (("f_back", None),)

    def __init__(self, data, background, parameters=None, **kwargs):
        self.density = None
        super(ModelFitConstantBackground, self).__init__(data=data, parameters=parameters, **kwargs)
        self.background = background
        if isinstance(background, SingleStars):            # O(N M) kernel-density precompute: on this rank's device
            lnbg = self.background(self.v, self.verr, context=kwargs.get("context"))
        else:
            lnbg = self.background(self.v, self.verr)
        self.lnlike_background = np.asarray(lnbg, dtype=np.float64)

    def _catalog_model(self):
        return self._model_id, {"lnlike_bg": self.lnlike_background, "density": self.density}

    def lnlike(self, values, no_sum=False):
        """Log-likelihood (model.py:565-623); with ``no_sum`` the per-star values instead of their sum."""
        if no_sum:
            self.fetch_parameter_values(np.asarray(values, dtype=np.float64).reshape(-1))
            return self._per_star(values, "lnlike")
        return super(ModelFitConstantBackground, self).lnlike(values)

    def calculate_membership_probabilities(self, chain, n_burn):
        """A-posteriori membership probabilities at the chain's median parameters (model.py:625-687)."""
        bestfit = self.compute_bestfit_values(chain=chain, n_burn=n_burn)
        median = np.array([bestfit.loc["median"][name] for name in self.fitted_parameters])
        return self.membership_probabilities(median)

    def membership_probabilities(self, values):
        return self._per_star(values, "membership")
