"""Fixed single-Gaussian background population in radial-velocity space (reference:
background/gaussian.py:9-28).  Evaluated ONCE per Runner instance on the host; only its per-star
output column travels to HBM."""
import numpy as np

from .. import units

_LN_2PI = np.log(2.0 * np.pi)


def gaussian_lnpdf(residual, variance):
    """ln N(residual; 0, variance) = -1/2 [ln(2 pi variance) + residual^2 / variance], elementwise."""
    variance = np.asarray(variance, dtype=np.float64)
    return -0.5 * np.log(2.0 * np.pi * variance) - 0.5 * np.square(residual) / variance


class Gaussian(object):
    """Background stars drawn from N(mean, sigma) km/s, convolved with each star's measurement error."""

    def __init__(self, mean, sigma):
        self.mean = float(units.to_unit(mean, "km/s"))
        self.sigma = float(units.to_unit(sigma, "km/s"))

    def __call__(self, v, verr):
        velocity = units.to_unit(v, "km/s")
        error = units.to_unit(verr, "km/s")
        return gaussian_lnpdf(velocity - self.mean, error * error + self.sigma * self.sigma)
