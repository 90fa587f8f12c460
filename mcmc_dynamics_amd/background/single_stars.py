"""Background population described by M individual comparison stars (kernel-density estimate in
radial-velocity space; reference: background/single_stars.py:9-77).

    p(v_i, verr_i) = (1/M) sum_j N(v_i - v_j; verr_i^2 + sigma_int^2)

The O(N M) evaluation runs on the GPU (``mcd_kde_background``, csrc/mcd_kde.hip): at N = 1e6 test stars and
M = 1e4 comparison stars it is 1e10 Gaussian kernels, minutes of NumPy on the host and milliseconds on the device.
Its output is the per-star ``lnlike_background`` column of the fixed-background likelihood (runner.py:96-106)."""
import numpy as np

from .. import _native, units


class SingleStars(object):

    def __init__(self, v):
        self.v = np.atleast_1d(units.to_unit(v, "km/s")).astype(np.float64)
        self.n_stars = self.v.size

    def __call__(self, v, verr, sigma_int=0.0, context=None):
        """Log-likelihood of each (v, verr) under the comparison-star population (single_stars.py:42-77).

        ``context`` is the ``_native.Context`` to run on (default: the process-wide single-GPU context)."""
        v = np.atleast_1d(units.to_unit(v, "km/s"))
        verr = np.atleast_1d(units.to_unit(verr, "km/s"))
        if v.shape != verr.shape:
            raise ValueError("v and verr must have the same shape")
        sigma_int = float(units.to_unit(sigma_int, "km/s"))
        ctx = context if context is not None else _native.default_context()
        return ctx.kde_background(self.v, v, verr, sigma_int).reshape(v.shape)
