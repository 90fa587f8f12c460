"""Stand-in for emcee (not installed). Golden-vector generation never samples."""


class EnsembleSampler(object):
    def __init__(self, *args, **kwargs):
        raise RuntimeError("emcee stand-in: sampling is not available in the golden-vector generator")
