from .runner import Runner  # noqa: F401
from .constant import ConstantFit, ConstantFitGB  # noqa: F401
from .model import ModelFit, ModelFitGB, ModelFitConstantBackground  # noqa: F401
from .binned import BinnedConstantFit  # noqa: F401
