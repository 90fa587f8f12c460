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

    # ------------------------------------------------------------------ GPU evaluation
    def _centre_is_fixed(self):
        pr, pd = self.parameters["ra_center"], self.parameters["dec_center"]
        return pr.fixed and pd.fixed and pr.expr is None and pd.expr is None

    def _catalog_spec(self):
        """(key, constructor kwargs) of the device catalogue for the current parameter configuration."""
        if self._centre_is_fixed():
            centre = (float(self.parameters["ra_center"].value), float(self.parameters["dec_center"].value))
        else:
            centre = None
        model = self._model_id
        extra = {}
        if self.background is not None and model == _native.MODEL_CONST:
            model = _native.MODEL_CONST_BGFIXED
            extra = {"lnlike_bg": self.lnlike_background, "pmember": self.pmember}
        return (model, centre), dict(model=model, centre=centre, **extra)

    def _ensure_catalog(self):
        key, spec = self._catalog_spec()
        if self._catalog is None or key != self._catalog_key:
            if self._catalog is not None:
                self._catalog.close()
            self._catalog = _native.Catalog(self.context, self.ra, self.dec, self.v, self.verr,
                                            precision=self._precision, **self._extra_columns(), **spec)
            self._catalog_key = key
        return self._catalog

    def _extra_columns(self):
        return {}

    def _kernel_table(self, resolved, free_centre):
        cols = [self._canonical(resolved, "v_sys", "km/s"), self._canonical(resolved, "sigma_max", "km/s"),
                self._canonical(resolved, "v_maxx", "km/s"), self._canonical(resolved, "v_maxy", "km/s")]
        if free_centre:
            cols += [self._canonical(resolved, "ra_center", "deg"), self._canonical(resolved, "dec_center", "deg")]
        return cols

    def _lnlike_batch(self, resolved):
        cat = self._ensure_catalog()
        table = np.stack(self._kernel_table(resolved, self._catalog_key[1] is None), axis=1)
        return cat.loglike(table)

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

    def _extra_columns(self):
        return {"density": self.density}

    def _kernel_table(self, resolved, free_centre):
        cols = super(ConstantFitGB, self)._kernel_table(resolved, free_centre)
        return cols + [self._canonical(resolved, "v_back", "km/s"), self._canonical(resolved, "sigma_back", "km/s"),
                       resolved["f_back"]]

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
        resolved = self.parameters.resolve_batch(np.asarray(values, dtype=np.float64).reshape(1, -1))
        cat = self._ensure_catalog()
        row = np.stack(self._kernel_table(resolved, self._catalog_key[1] is None), axis=1)[0]
        return cat.membership(row)
