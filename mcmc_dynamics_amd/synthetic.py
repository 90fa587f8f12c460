"""Synthetic star catalogues and walker ensembles for the log-likelihood hot path.

Follows the mock-data recipe of the reference's only runnable demo
(``bin/run_tests.py:33-70``) minus astropy, with the fixed seeds and sizes that
SURVEY.md section 8(d) prescribes for the BASELINE.json configurations:

* catalogue seed = 20260401 + config number, walker seed = 7 + config number
* centre (56.345 deg, -26.675 deg)                       (run_tests.py:44)
* separation ~ TruncNormal(0, r_max; scale r_max/2), r_max = 5 x 60 arcsec (:46-49)
* position angle ~ U(-pi, pi)                              (:50)
* v_sys = 0, sigma = 10 km/s, v_max = 0.5 sigma, theta_0 ~ U(0, 2 pi)   (:24, 39-41)
* verr = 0.1 sigma LogNormal(0, 0.5)                       (:25, 66)
* v = v_los,true + N(0, sigma) + N(0, verr)                (:61-67)

Only numpy (and scipy.special.ndtri for the truncated normal) is used, so the file can
also be loaded stand-alone by ``oracle/make_golden.py`` under another interpreter.
"""
import numpy as np

CENTER_RA_DEG = 56.345
CENTER_DEC_DEG = -26.675
R0_ARCMIN = 10800.0 / np.pi          # calc_xy_offset.py:11
DEG = np.pi / 180.0

CATALOG_SEED_BASE = 20260401
WALKER_SEED_BASE = 7

TRUTH = {"v_sys": 0.0, "sigma_max": 10.0, "v_max": 5.0,
         "v_back": 20.0, "sigma_back": 40.0, "f_back": 0.25}


def _ndtri(p):
    from scipy.special import ndtri
    return ndtri(p)


def _truncated_halfnormal(rng, scale, upper, size):
    """Draw from N(0, scale) truncated to [0, upper] by inverse-CDF sampling."""
    from scipy.special import ndtr
    hi = ndtr(upper / scale)
    u = rng.random(size)
    return scale * _ndtri(0.5 + u * (hi - 0.5))


def offsets_to_radec(dx_arcmin, dy_arcmin, ra_center_deg, dec_center_deg):
    """Exact inverse of the orthographic ``calc_xy_offset`` (calc_xy_offset.py:30-31).

    dx = -r0 cos(dec) sin(ra - ra_c),  dy = r0 (sin(dec) cos(dec_c) - cos(dec) sin(dec_c) cos(ra - ra_c)).
    """
    xi = -np.asarray(dx_arcmin, dtype=np.float64) / R0_ARCMIN     # cos(dec) sin(dra)
    eta = np.asarray(dy_arcmin, dtype=np.float64) / R0_ARCMIN
    sdc, cdc = np.sin(dec_center_deg * DEG), np.cos(dec_center_deg * DEG)
    zeta = np.sqrt(np.maximum(0.0, 1.0 - xi * xi - eta * eta))    # cos(angular distance)
    sin_dec = eta * cdc + zeta * sdc
    dec = np.arcsin(np.clip(sin_dec, -1.0, 1.0))
    dra = np.arctan2(xi, zeta * cdc - eta * sdc)
    return ra_center_deg + dra / DEG, dec / DEG


def make_catalog(n_stars, config=2, seed=None, background=False, r_max_arcsec=300.0):
    """Return a dict of float64 columns: ra, dec [deg], v, verr [km/s], plus truth metadata.

    With ``background=True`` 20 % of the stars are redrawn from N(20, 40) km/s and the
    columns ``density`` and ``pmember`` are added.
    """
    if seed is None:
        seed = CATALOG_SEED_BASE + int(config)
    rng = np.random.default_rng(seed)
    n = int(n_stars)

    theta_0 = 2.0 * np.pi * rng.random()
    sigma = TRUTH["sigma_max"]
    v_max = TRUTH["v_max"]

    r_max = r_max_arcsec / 60.0                                   # arcmin
    sep = _truncated_halfnormal(rng, r_max / 2.0, r_max, n)
    theta = rng.uniform(-np.pi, np.pi, size=n)
    ra, dec = offsets_to_radec(sep * np.cos(theta), sep * np.sin(theta), CENTER_RA_DEG, CENTER_DEC_DEG)

    v = TRUTH["v_sys"] + v_max * np.sin(theta - theta_0)
    v = v + rng.normal(scale=sigma, size=n)
    verr = 0.1 * sigma * rng.lognormal(0.0, 0.5, size=n)
    v = v + rng.normal(size=n) * verr

    cat = {"ra": ra, "dec": dec, "v": v, "verr": verr}
    if background:
        is_back = rng.random(n) < 0.2
        v_back = rng.normal(TRUTH["v_back"], TRUTH["sigma_back"], size=n) + rng.normal(size=n) * verr
        cat["v"] = np.where(is_back, v_back, v)
        density = np.clip(np.exp(-sep * sep / (2.0 * (r_max / 3.0) ** 2)), 0.02, 1.0)
        cat["density"] = density
        cat["pmember"] = density / (density + TRUTH["f_back"])
    cat["truth"] = {"theta_0": float(theta_0), "v_maxx": float(v_max * np.cos(theta_0)),
                    "v_maxy": float(v_max * np.sin(theta_0)), "v_sys": TRUTH["v_sys"],
                    "sigma_max": sigma, "ra_center": CENTER_RA_DEG, "dec_center": CENTER_DEC_DEG,
                    "v_back": TRUTH["v_back"], "sigma_back": TRUTH["sigma_back"], "f_back": TRUTH["f_back"]}
    return cat


BLOCK_STARS = 125000


def make_catalog_range(n_total, lo, hi, config=4, background=False, block=BLOCK_STARS):
    """Stars [lo, hi) of an ``n_total``-star catalogue that is defined block by block (``block`` stars per block, block b
    seeded with (CATALOG_SEED_BASE + config, b)), so that every rank of a sharded run generates only its own part of
    the SAME catalogue whatever the number of ranks (strong scaling, C4).  Same recipe as ``make_catalog``; the
    rotation axis is that of block 0's generator."""
    n_total, lo, hi = int(n_total), int(lo), int(hi)
    if not 0 <= lo <= hi <= n_total:
        raise ValueError("range [{0}, {1}) outside catalogue of {2} stars".format(lo, hi, n_total))
    base = CATALOG_SEED_BASE + int(config)
    truth = make_catalog(1, config=config, seed=base, background=background)["truth"]
    parts = []
    for b in range(lo // block, (max(hi, lo + 1) - 1) // block + 1):
        b0, b1 = b * block, min(n_total, (b + 1) * block)
        if b1 <= b0:
            break
        cat = _catalog_with_axis(b1 - b0, np.random.default_rng([base, b]), truth["theta_0"], background)
        s0, s1 = max(lo, b0) - b0, min(hi, b1) - b0
        parts.append({k: v[s0:s1] for k, v in cat.items()})
    keys = parts[0].keys() if parts else ("ra", "dec", "v", "verr")
    out = {k: (np.concatenate([p[k] for p in parts]) if parts else np.empty(0)) for k in keys}
    out["truth"] = truth
    return out


def _catalog_with_axis(n, rng, theta_0, background, r_max_arcsec=300.0):
    sigma, v_max = TRUTH["sigma_max"], TRUTH["v_max"]
    r_max = r_max_arcsec / 60.0
    sep = _truncated_halfnormal(rng, r_max / 2.0, r_max, n)
    theta = rng.uniform(-np.pi, np.pi, size=n)
    ra, dec = offsets_to_radec(sep * np.cos(theta), sep * np.sin(theta), CENTER_RA_DEG, CENTER_DEC_DEG)
    v = TRUTH["v_sys"] + v_max * np.sin(theta - theta_0) + rng.normal(scale=sigma, size=n)
    verr = 0.1 * sigma * rng.lognormal(0.0, 0.5, size=n)
    v = v + rng.normal(size=n) * verr
    cat = {"ra": ra, "dec": dec, "v": v, "verr": verr}
    if background:
        is_back = rng.random(n) < 0.2
        v_back = rng.normal(TRUTH["v_back"], TRUTH["sigma_back"], size=n) + rng.normal(size=n) * verr
        cat["v"] = np.where(is_back, v_back, v)
        density = np.clip(np.exp(-sep * sep / (2.0 * (r_max / 3.0) ** 2)), 0.02, 1.0)
        cat["density"] = density
        cat["pmember"] = density / (density + TRUTH["f_back"])
    return cat


# prior bounds of config/constant.json and config/constant_with_background.json (:6-14)
BOUNDS = {"v_sys": (-np.inf, np.inf), "sigma_max": (0.0, np.inf), "v_maxx": (-np.inf, np.inf),
          "v_maxy": (-np.inf, np.inf), "ra_center": (0.0, 360.0), "dec_center": (-90.0, 90.0),
          "v_back": (-np.inf, np.inf), "sigma_back": (0.0, np.inf), "f_back": (0.0, 1.0)}


def make_walkers(n_walkers, names, truth, config=2, seed=None):
    """Gaussian ball around the truth: truth (1 + 0.05 N(0,1)); +-0.5 absolute for zero truths;
    centre coordinates get a 0.05 arcmin ball; clipped into the prior bounds."""
    if seed is None:
        seed = WALKER_SEED_BASE + int(config)
    rng = np.random.default_rng(seed)
    pos = np.empty((int(n_walkers), len(names)), dtype=np.float64)
    for j, name in enumerate(names):
        t = float(truth[name])
        g = rng.normal(size=int(n_walkers))
        if name in ("ra_center", "dec_center"):
            col = t + (0.05 / 60.0) * g
        elif t == 0.0:
            col = 0.5 * g
        else:
            col = t * (1.0 + 0.05 * g)
        lo, hi = BOUNDS[name]
        pos[:, j] = np.clip(col, lo, hi)
    return pos


def example_polar_to_catalog(r_arcmin, theta_rad, v, verr, ra_center_deg=180.0):
    """Adapter for the reference's ``example/data/test.csv`` legacy polar layout
    (rows r, theta, v, verr): place the stars about (ra_center, 0 deg) so that
    ``calc_xy_offset`` returns x = r cos(theta), y = r sin(theta)."""
    x = np.asarray(r_arcmin) * np.cos(theta_rad)
    y = np.asarray(r_arcmin) * np.sin(theta_rad)
    ra, dec = offsets_to_radec(x, y, ra_center_deg, 0.0)
    return {"ra": ra, "dec": dec, "v": np.asarray(v, dtype=np.float64), "verr": np.asarray(verr, dtype=np.float64)}
