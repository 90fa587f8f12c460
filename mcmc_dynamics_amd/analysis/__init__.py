"""Analysis classes: the reference's ``Runner`` hierarchy with the per-star likelihood arithmetic moved to the GPU.

================================  =====================================================================
class                             reference counterpart
================================  =====================================================================
``Runner``                        ``mcmc_dynamics/analysis/runner.py`` (posterior API + MCMC driver)
``ConstantFit``, ``ConstantFitGB``  ``analysis/constant.py`` (constant rotation + dispersion, Gaussian background)
``ModelFit``, ``ModelFitGB``,     ``analysis/model.py`` (Lynden-Bell rotation curve + Plummer dispersion profile)
``ModelFitConstantBackground``
``BinnedConstantFit``             the per-radial-bin loop of ``bin/run_tests.py:75-124`` as ONE batched posterior
================================  =====================================================================
"""
from .binned import BinnedConstantFit
from .constant import ConstantFit, ConstantFitGB
from .model import ModelFit, ModelFitConstantBackground, ModelFitGB
from .runner import Runner

__all__ = ["Runner", "ConstantFit", "ConstantFitGB", "ModelFit", "ModelFitGB", "ModelFitConstantBackground",
           "BinnedConstantFit"]
