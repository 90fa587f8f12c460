"""Fixed single-Gaussian background population in radial-velocity space (reference:
background/gaussian.py:9-28).  Evaluated ONCE per Runner instance on the host; only its per-star
output column travels to HBM."""
import numpy as np

from .. import units


class Gaussian(object):

    def __init__(self, mean, sigma):
        self.mean = float(units.to_unit(mean, "km/s"))
        self.sigma = float(units.to_unit(sigma, "km/s"))

    def __call__(self, v, verr):
        v = units.to_unit(v, "km/s")
        verr = units.to_unit(verr, "km/s")
        norm = verr * verr + self.sigma * self.sigma
        exponent = -0.5 * np.power(v - self.mean, 2) / norm
        return -0.5 * np.log(2. * np.pi * norm) + exponent
