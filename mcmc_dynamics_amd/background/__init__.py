"""Fixed background populations in radial-velocity space.  An instance is called once per ``Runner`` with the
catalogue's ``(v, verr)`` and returns the per-star background log-likelihood column that is pinned in HBM next to the
star records (reference: ``mcmc_dynamics/background``)."""
from .gaussian import Gaussian, gaussian_lnpdf
from .single_stars import SingleStars

__all__ = ["Gaussian", "SingleStars", "gaussian_lnpdf"]
