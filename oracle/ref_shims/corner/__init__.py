"""Stand-in for corner (plots only)."""


def corner(*args, **kwargs):
    raise RuntimeError("corner stand-in: plotting is not available in the golden-vector generator")
