"""MI355X-native log-likelihood hot path of mcmc_dynamics (rotation + dispersion kinematic model).

Python host code in the shape of the reference's ``mcmc_dynamics`` package, calling hand-written
gfx950 HIP kernels through the ctypes C-ABI of ``include/mcd.h``.  No CPU fallback exists.
"""
__version__ = "0.1.0"

from .parameter import Parameter, Parameters  # noqa: E402,F401
from .utils.data_reader import DataReader  # noqa: E402,F401
from .background import Gaussian, SingleStars  # noqa: E402,F401
