"""Tangent-plane geometry used on the host (binning, distances, chain post-processing).

``calc_xy_offset`` follows the reference's utils/coordinates/calc_xy_offset.py:9-33 (orthographic
projection, van de Ven et al. 2006); the per-term geometry of the likelihood itself runs on the GPU
(csrc/mcd_kernels.hip)."""
import logging

import numpy as np

from .. import units

logger = logging.getLogger(__name__)

R0_ARCMIN = 10800.0 / np.pi
DEG2RAD = np.pi / 180.0


def calc_xy_offset(ra, dec, ra_center, dec_center):
    """(x, y) offsets in arcmin of (ra, dec) from the centre; inputs in deg (bare numbers) or Quantities."""
    ra = units.to_unit(ra, "deg")
    dec = units.to_unit(dec, "deg")
    ra_center = units.to_unit(ra_center, "deg")
    dec_center = units.to_unit(dec_center, "deg")
    dra = (ra - ra_center) * DEG2RAD
    dec_r, dec_c = dec * DEG2RAD, dec_center * DEG2RAD
    dx = -R0_ARCMIN * np.cos(dec_r) * np.sin(dra)
    dy = R0_ARCMIN * (np.sin(dec_r) * np.cos(dec_c) - np.cos(dec_r) * np.sin(dec_c) * np.cos(dra))
    return dx, dy


def get_amplitude_and_angle(pars, return_samples=False):
    """Rotation amplitude ``v_max`` and axis angle ``theta_0`` statistics from chain samples
    (reference: utils/coordinates/get_amplitude_and_angle.py:10-51).

    ``pars`` maps parameter names to flat sample arrays and must allow recovering ``v_maxx``,
    ``v_maxy`` and ``theta_0`` (any two of them).  Angles are measured about the direction of the
    median (v_maxx, v_maxy) so that the +-pi wrap does not split the distribution; ``v_max`` is the
    component of each sample along that direction.  Returns (results, v_max, theta) -- the last two
    are None unless ``return_samples`` -- with ``results.loc['median']['theta_0']`` etc."""
    from .results import ResultsTable
    pars = dict(pars)
    if "theta_0" not in pars and "v_maxx" in pars and "v_maxy" in pars:
        pars["theta_0"] = np.arctan2(pars["v_maxy"], pars["v_maxx"])
    elif "v_maxx" not in pars and "theta_0" in pars and "v_maxy" in pars:
        pars["v_maxx"] = pars["v_maxy"] * np.tan(pars["theta_0"])
    elif "v_maxy" not in pars and "theta_0" in pars and "v_maxx" in pars:
        pars["v_maxy"] = pars["v_maxx"] / np.tan(pars["theta_0"])
    for name in ("theta_0", "v_maxx", "v_maxy"):
        if name not in pars:
            logger.error("Failed to recover parameter %s.", name)
            return None, None, None

    vx = np.asarray(pars["v_maxx"], dtype=np.float64)
    vy = np.asarray(pars["v_maxy"], dtype=np.float64)
    median_theta = np.arctan2(np.median(vy), np.median(vx))
    theta = np.asarray(pars["theta_0"], dtype=np.float64) - median_theta
    theta = np.where(theta < -np.pi, theta + 2 * np.pi, theta)
    theta = np.where(theta > np.pi, theta - 2 * np.pi, theta)
    v_max = vx * np.cos(-median_theta) - vy * np.sin(-median_theta)

    results = ResultsTable()
    for name, values in (("v_max", v_max), ("theta_0", theta)):
        p16, p50, p84 = np.percentile(values, [16, 50, 84])
        results.add_column(name, p50, p84 - p50, p50 - p16, unit="rad" if name == "theta_0" else None)
    results.loc["median"]["theta_0"] += median_theta
    if return_samples:
        return results, v_max, theta
    return results, None, None
