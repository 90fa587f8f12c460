"""Background population described by M individual comparison stars (kernel-density estimate in
radial-velocity space; reference: background/single_stars.py:9-77).

    p(v_i, verr_i) = (1/M) sum_j N(v_i - v_j; verr_i^2 + sigma_int^2)

One-off O(N M) precompute on the host at Runner construction; only its per-star output column is on
the hot path.  Evaluated in row blocks so that the (M, N) outer product never exceeds ~64 MiB."""
import numpy as np

from .. import units


class SingleStars(object):

    def __init__(self, v):
        self.v = np.atleast_1d(units.to_unit(v, "km/s")).astype(np.float64)
        self.n_stars = self.v.size

    def __call__(self, v, verr, sigma_int=0.0):
        v = np.atleast_1d(units.to_unit(v, "km/s"))
        verr = np.atleast_1d(units.to_unit(verr, "km/s"))
        sigma_int = float(units.to_unit(sigma_int, "km/s"))
        norm = sigma_int ** 2 + verr ** 2
        out = np.empty(v.size, dtype=np.float64)
        block = max(1, int(8_000_000 // max(1, self.n_stars)))
        for s in range(0, v.size, block):
            e = slice(s, s + block)
            # log-sum-exp over the comparison stars, single_stars.py:72-77
            exp_coeff = -(np.subtract.outer(self.v, v[e])) ** 2 / (2. * norm[e])
            exp_coeff_max = np.max(exp_coeff, axis=0)
            out[e] = exp_coeff_max + np.log(np.sum(np.exp(exp_coeff - exp_coeff_max) / (np.sqrt(2. * np.pi * norm[e])),
                                                   axis=0)) - np.log(self.n_stars)
        return out
