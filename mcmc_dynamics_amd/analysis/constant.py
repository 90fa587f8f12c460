"""``ConstantFit`` / ``ConstantFitGB``: constant rotation amplitude + constant dispersion, optionally with
a background population (reference: analysis/constant.py).

    v_los,i   = v_sys + v_max sin(theta_i - theta_0),   v_max = hypot(v_maxx, v_maxy), theta_0 = arctan2(v_maxy, v_maxx)
    sigma_los = sigma_max

The model functions are kept as host methods for inspection and plotting; the likelihood itself is
evaluated for all walkers at once by the HIP kernels (``Runner._lnlike_batch`` below).
"""
import logging

import numpy as np

from .. import _native
from ..parameter import Parameter, Parameters
from ..utils.coordinates import calc_xy_offset, get_amplitude_and_angle
from .runner import Runner

logger = logging.getLogger(__name__)

_INF = np.inf

# Default parameter sets: name, unit, min, max, label, initials -- the contents of the reference's
# config/constant.json:6-11 and config/constant_with_background.json:12-14 (order = sampler order).
_CONSTANT_DEFAULTS = (
    ("v_sys", "km/s", -_INF, _INF, r"$v_{\rm sys}$", "rng.normal(size=n)"),
    ("sigma_max", "km/s", 0.0, _INF, r"$\sigma_{\rm max}$", "rng.lognormal(size=n)"),
    ("v_maxx", "km/s", -_INF, _INF, r"$v_{\rm max,\,x}$", "rng.normal(size=n)"),
    ("v_maxy", "km/s", -_INF, _INF, r"$v_{\rm max,\,y}$", "rng.normal(size=n)"),
    ("ra_center", "deg", 0.0, 360.0, r"$\alpha_{\rm c}$", None),
    ("dec_center", "deg", -90.0, 90.0, r"$\delta_{\rm c}$", None),
)
_BACKGROUND_DEFAULTS = (
    ("v_back", "km/s", -_INF, _INF, r"$v_{\rm back}$", "rng.normal(size=n)"),
    ("sigma_back", "km/s", 0.0, _INF, r"$\sigma_{\rm back}$", "rng.lognormal(size=n)"),
    ("f_back", None, 0.0, 1.0, r"$f_{\rm back}$", "rng.uniform(size=n)"),
)


def _build_defaults(rows):
    pars = Parameters()
    for name, unit, lo, hi, label, initials in rows:
        pars.add(Parameter(name, unit=unit, min=lo, max=hi, label=label, initials=initials))
    return pars


class ConstantFit(Runner):
    MODEL_PARAMETERS = ["v_sys", "sigma_max", "v_maxx", "v_maxy", "ra_center", "dec_center"]
    OBSERVABLES = {"v": "km/s", "verr": "km/s", "ra": "deg", "dec": "deg"}

    _default_rows = _CONSTANT_DEFAULTS
    _model_id = _native.MODEL_CONST

    def __init__(self, data, parameters=None, **kwargs):
        """
        Parameters
        ----------
        data : DataReader
            Columns ``ra, dec, v, verr`` (and ``pmember`` with a ``background``).
        parameters : Parameters, optional
            Defaults to the constant-model parameter set (free centre, as in the reference).
        kwargs
            Forwarded to ``Runner.__init__`` (``seed``, ``background``, ``context``, ``precision``).
        """
        self.ra = None
        self.dec = None
        if parameters is None:
            parameters = self.default_parameters()
        super(ConstantFit, self).__init__(data=data, parameters=parameters, **kwargs)

    @classmethod
    def default_parameters(cls):
        if cls.parameters_file is not None:
            return Parameters().load(cls.parameters_file)
        return _build_defaults(cls._default_rows)

    # ------------------------------------------------------------------ host-side model functions
    def dispersion_model(self, sigma_max, **kwargs):
        """Constant dispersion at every data point (constant.py:52-74)."""
        if kwargs:
            raise IOError('Unknown keyword argument(s) "{0}" for method {1}.dispersion_model.'.format(
                ", ".join(kwargs.keys()), self.__class__.__name__))
        return sigma_max * np.ones(self.n_data, dtype=np.float64)

    def rotation_model(self, v_sys, v_maxx, v_maxy, ra_center, dec_center, **kwargs):
        """Line-of-sight velocity of the rotation field at every data point (constant.py:76-111)."""
        if kwargs:
            raise IOError('Unknown keyword argument(s) "{0}" for method {1}.rotation_model.'.format(
                ", ".join(kwargs.keys()), self.__class__.__name__))
        dx, dy = calc_xy_offset(ra=self.ra, dec=self.dec, ra_center=ra_center, dec_center=dec_center)
        theta = np.arctan2(dy, dx)
        v_max = np.sqrt(v_maxx ** 2 + v_maxy ** 2)
        theta_0 = np.arctan2(v_maxy, v_maxx)
        return v_sys + v_max * np.sin(theta - theta_0)

    # ------------------------------------------------------------------ GPU evaluation (plumbing in Runner)
    _KERNEL_HEAD = (("v_sys", "km/s"), ("sigma_max", "km/s"), ("v_maxx", "km/s"), ("v_maxy", "km/s"))

    def _catalog_model(self):
        if self.background is not None and self._model_id == _native.MODEL_CONST:
            return _native.MODEL_CONST_BGFIXED, {"lnlike_bg": self.lnlike_background, "pmember": self.pmember}
        return self._model_id, {}

    def lnlike(self, values):
        """Log-likelihood of the data for one parameter vector (constant.py:113-154)."""
        return super(ConstantFit, self).lnlike(values)

    # ------------------------------------------------------------------ post-processing
    def compute_theta_vmax(self, chain, n_burn, return_samples=False):
        """Position angle ``theta_0`` and amplitude ``v_max`` of the rotation field from a chain
        (constant.py:156-214)."""
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


class ConstantFitGB(ConstantFit):
    """ConstantFit plus a background component that is Gaussian in radial-velocity space, with
    per-walker parameters (v_back, sigma_back, f_back) and the stellar surface density as membership
    prior, m_i = density_i / (density_i + f_back) (constant.py:250-374)."""

    MODEL_PARAMETERS = ConstantFit.MODEL_PARAMETERS + ["v_back", "sigma_back", "f_back"]
    OBSERVABLES = dict(ConstantFit.OBSERVABLES, **{"density": None})

    _default_rows = _CONSTANT_DEFAULTS + _BACKGROUND_DEFAULTS
    _model_id = _native.MODEL_CONST_BGGAUSS

    def __init__(self, data, parameters=None, **kwargs):
        self.density = None
        background = kwargs.pop("background", None)
        if background is not None:
            logger.error("Class ConstantFitGB does not support additional background components.")
        super(ConstantFitGB, self).__init__(data=data, parameters=parameters, **kwargs)

    _KERNEL_TAIL = (("v_back", "km/s"), ("sigma_back", "km/s"), ("f_back", None))

    def _catalog_model(self):
        return self._model_id, {"density": self.density}

    def lnlike(self, values):
        """Log-likelihood including the Gaussian background mixture (constant.py:293-324)."""
        return super(ConstantFitGB, self).lnlike(values)

    def calculate_membership_probabilities(self, chain, n_burn):
        """Posterior membership probability of every star at the chain's median parameters
        (constant.py:366-374)."""
        bestfit = self.compute_bestfit_values(chain=chain, n_burn=n_burn)
        median = np.array([bestfit.loc["median"][name] for name in self.fitted_parameters])
        return self.membership_probabilities(median)

    def membership_probabilities(self, values):
        """m e^{lnL_cluster} / (m e^{lnL_cluster} + (1 - m) e^{lnL_back}) for one parameter vector."""
        return self._per_star(values, "membership")
