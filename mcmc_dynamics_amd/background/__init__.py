from .gaussian import Gaussian  # noqa: F401
from .single_stars import SingleStars  # noqa: F401
