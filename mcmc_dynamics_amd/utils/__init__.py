from .coordinates import calc_xy_offset, get_amplitude_and_angle  # noqa: F401
from .data_reader import DataReader  # noqa: F401
