"""CPU ORACLE (test infrastructure, NOT the product path).

Plain-NumPy restatement of the reference's per-walker log-likelihood hot path
(skamann/mcmc-dynamics, pure Python).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module; the product package
``mcmc_dynamics_amd`` never does and fails loudly when its HIP library is missing.

Parity pin: the reference ships no tests or golden values of its own, so
this restatement is pinned against outputs of the reference itself, generated in the build
container by ``oracle/make_golden.py`` (fixtures in ``tests/golden/``) -- see
``tests/test_oracle_golden.py``.

Two layers:

* ``faithful_*``: op-for-op, one walker per call, same ufunc sequence as the reference
  (including the per-call ``calc_xy_offset`` -> ``arctan2`` -> ``sin`` geometry and the two separate
  ``np.sum`` reductions).  This is the timed CPU baseline (``cpu_baseline.kind = "port"``).
* ``batched_*``: the algebraically reduced ``(W, N)`` form that the HIP kernels implement
  (``sin(theta)``, ``cos(theta)`` precomputed for a fixed centre; angle-addition geometry for a free
  centre).  It is the executable specification of the kernels.

All quantities are plain float64 in the reference's canonical units: deg (ra, dec, centres),
km/s (velocities), arcmin (offsets).
"""
import numpy as np

DEG2RAD = np.pi / 180.0            # astropy's deg->rad factor, applied inside np.sin/np.cos of a Quantity
R0_ARCMIN = 10800.0 / np.pi        # utils/coordinates/calc_xy_offset.py:11


# ----------------------------------------------------------------------------- faithful layer
def calc_xy_offset(ra, dec, ra_center, dec_center):
    """utils/coordinates/calc_xy_offset.py:9-33 (orthographic tangent-plane offsets, arcmin)."""
    dra = (ra - ra_center) * DEG2RAD
    dec_r = dec * DEG2RAD
    dec_c = dec_center * DEG2RAD
    dx = -R0_ARCMIN * np.cos(dec_r) * np.sin(dra)                                             # :30
    dy = R0_ARCMIN * (np.sin(dec_r) * np.cos(dec_c) - np.cos(dec_r) * np.sin(dec_c) * np.cos(dra))   # :31
    return dx, dy


def rotation_model(ra, dec, v_sys, v_maxx, v_maxy, ra_center, dec_center):
    """analysis/constant.py:76-111."""
    dx, dy = calc_xy_offset(ra, dec, ra_center, dec_center)        # :106
    theta = np.arctan2(dy, dx)                                     # :107
    v_max = np.sqrt(v_maxx ** 2 + v_maxy ** 2)                     # :109
    theta_0 = np.arctan2(v_maxy, v_maxx)                           # :110
    return v_sys + v_max * np.sin(theta - theta_0)                 # :111


def dispersion_model(n, sigma_max):
    """analysis/constant.py:52-74."""
    return sigma_max * np.ones(n, dtype=np.float64)


def gaussian_background(v, verr, mean, sigma):
    """background/gaussian.py:23-28."""
    norm = verr * verr + sigma * sigma
    exponent = -0.5 * np.power(v - mean, 2) / norm
    return -0.5 * np.log(2. * np.pi * norm) + exponent


def single_stars_background(comp, v, verr, sigma_int=0.0, block=20000):
    """background/single_stars.py:42-77: log of the mean of M Gaussian kernels N(v_i - comp_j; verr_i^2 + sigma_int^2),
    with the reference's log-sum-exp (largest exponent subtracted per test star).  Evaluated in column blocks so the
    (M, N) outer product stays small; each column is computed exactly as in the reference."""
    comp = np.asarray(comp, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    verr = np.asarray(verr, dtype=np.float64)
    norm_all = sigma_int ** 2 + verr ** 2
    out = np.empty(v.size, dtype=np.float64)
    block = max(1, min(block, int(4_000_000 // max(1, comp.size))))
    for s in range(0, v.size, block):
        e = slice(s, s + block)
        norm = norm_all[e]
        exp_coeff = -(np.subtract.outer(comp, v[e])) ** 2 / (2. * norm)
        exp_coeff_max = np.max(exp_coeff, axis=0)
        out[e] = exp_coeff_max + np.log(np.sum(np.exp(exp_coeff - exp_coeff_max) / (np.sqrt(2. * np.pi * norm)),
                                               axis=0)) - np.log(comp.size)
    return out


def calculate_lnlike(v, verr, v_los, sigma_los, lnlike_background=None, pmember=None):
    """analysis/runner.py:240-286."""
    norm = verr * verr + sigma_los * sigma_los                     # :261
    exponent = -0.5 * np.power(v - v_los, 2) / norm                # :262
    if lnlike_background is None:
        sum1 = -0.5 * np.sum(np.log(2. * np.pi * norm))            # :269
        sum2 = np.sum(exponent)                                    # :270
        return sum1 + sum2                                         # :271
    lnlike_member = -0.5 * np.log(2. * np.pi * norm) + exponent    # :280
    max_lnlike = np.max([lnlike_member, lnlike_background], axis=0)    # :282
    lnlike = max_lnlike + np.log(pmember * np.exp(lnlike_member - max_lnlike) + (
        1. - pmember) * np.exp(lnlike_background - max_lnlike))    # :283-284
    return lnlike.sum()                                            # :286


def faithful_constant_lnlike(cat, v_sys, sigma_max, v_maxx, v_maxy, ra_center, dec_center,
                             lnlike_background=None, pmember=None):
    """ConstantFit.lnlike, analysis/constant.py:113-154."""
    v_los = rotation_model(cat["ra"], cat["dec"], v_sys, v_maxx, v_maxy, ra_center, dec_center)
    sigma_los = dispersion_model(len(cat["v"]), sigma_max)
    return calculate_lnlike(cat["v"], cat["verr"], v_los, sigma_los, lnlike_background, pmember)


def faithful_constant_gb_terms(cat, v_sys, sigma_max, v_maxx, v_maxy, ra_center, dec_center,
                               v_back, sigma_back, f_back):
    """ConstantFitGB._calculate_lnlike_cluster_back, analysis/constant.py:326-364."""
    v, verr = cat["v"], cat["verr"]
    norm = verr * verr + sigma_back * sigma_back                   # :333
    exponent = -0.5 * np.power(v - v_back, 2) / norm               # :334
    lnlike_back = -0.5 * np.log(2. * np.pi * norm) + exponent      # :336
    m = cat["density"] / (cat["density"] + f_back)                 # :339
    v_los = rotation_model(cat["ra"], cat["dec"], v_sys, v_maxx, v_maxy, ra_center, dec_center)
    sigma_los = dispersion_model(len(v), sigma_max)
    norm = verr * verr + sigma_los * sigma_los                     # :359
    exponent = -0.5 * np.power(v - v_los, 2) / norm                # :360
    lnlike_cluster = -0.5 * np.log(2. * np.pi * norm) + exponent   # :362
    return lnlike_cluster, lnlike_back, m


def faithful_constant_gb_lnlike(cat, *params):
    """ConstantFitGB.lnlike, analysis/constant.py:293-324."""
    lnlike_cluster, lnlike_back, m = faithful_constant_gb_terms(cat, *params)
    max_lnlike = np.max([lnlike_cluster, lnlike_back], axis=0)     # :320
    lnlike = max_lnlike + np.log(
        m * np.exp(lnlike_cluster - max_lnlike) + (1. - m) * np.exp(lnlike_back - max_lnlike))   # :322-323
    return lnlike.sum()                                            # :324


def gb_membership_probabilities(cat, *params):
    """ConstantFitGB.calculate_membership_probabilities, analysis/constant.py:366-374."""
    lc, lb, m = faithful_constant_gb_terms(cat, *params)
    return m * np.exp(lc) / (m * np.exp(lc) + (1. - m) * np.exp(lb))


# ---- ModelFit family (analysis/model.py): Lynden-Bell rotation curve + Plummer dispersion profile --------------
# astropy keeps r in arcmin and a, r_peak in arcsec; the composite units are reduced when a dimensionless number is
# added (x 3600 for arcmin^2/arcsec^2) or when the velocity is added to v_sys (x 60 for arcmin/arcsec).
def model_dispersion(ra, dec, sigma_max, a, ra_center, dec_center):
    """ModelFit.dispersion_model, analysis/model.py:93-127 (a in arcsec)."""
    dx, dy = calc_xy_offset(ra, dec, ra_center, dec_center)
    r = np.sqrt(dx ** 2 + dy ** 2)
    return sigma_max / (1. + (r ** 2 / a ** 2) * 3600.0) ** 0.25                  # :127


def model_rotation(ra, dec, v_sys, v_maxx, v_maxy, r_peak, ra_center, dec_center):
    """ModelFit.rotation_model, analysis/model.py:129-180 (r_peak in arcsec)."""
    dx, dy = calc_xy_offset(ra, dec, ra_center, dec_center)
    r = np.sqrt(dx ** 2 + dy ** 2)
    v_max = np.sqrt(v_maxx ** 2 + v_maxy ** 2)
    theta_0 = np.arctan2(v_maxy, v_maxx)
    theta = np.arctan2(dy, dx)
    x_pa = r * np.sin(theta - theta_0)
    return v_sys + (2. * (v_max / r_peak) * x_pa / (1. + ((r / r_peak) ** 2) * 3600.0)) * 60.0     # :180


def faithful_model_lnlike(cat, v_sys, sigma_max, a, v_maxx, v_maxy, r_peak, ra_center, dec_center,
                          lnlike_background=None, prior=None):
    """ModelFit.lnlike (model.py:182-222); with ``lnlike_background`` and the membership prior ``prior`` also
    ModelFitGB.lnlike (:391-456) and ModelFitConstantBackground.lnlike (:565-623)."""
    v_los = model_rotation(cat["ra"], cat["dec"], v_sys, v_maxx, v_maxy, r_peak, ra_center, dec_center)
    sigma_los = model_dispersion(cat["ra"], cat["dec"], sigma_max, a, ra_center, dec_center)
    return calculate_lnlike(cat["v"], cat["verr"], v_los, sigma_los, lnlike_background, prior)


def faithful_model_gb_lnlike(cat, v_sys, sigma_max, a, v_maxx, v_maxy, r_peak, ra_center, dec_center,
                             v_back, sigma_back, f_back):
    """ModelFitGB.lnlike, analysis/model.py:391-456."""
    lnlike_back = gaussian_background(cat["v"], cat["verr"], v_back, sigma_back)      # :423-426
    m = cat["density"] / (cat["density"] + f_back)                                  # :429
    return faithful_model_lnlike(cat, v_sys, sigma_max, a, v_maxx, v_maxy, r_peak, ra_center, dec_center,
                                 lnlike_back, m)


def faithful_model_cb_lnlike(cat, v_sys, sigma_max, a, v_maxx, v_maxy, r_peak, ra_center, dec_center, f_back,
                             lnlike_background, no_sum=False):
    """ModelFitConstantBackground.lnlike, analysis/model.py:565-623 (fixed background, density prior)."""
    m = cat["density"] / (cat["density"] + f_back)                                  # :588
    if not no_sum:
        return faithful_model_lnlike(cat, v_sys, sigma_max, a, v_maxx, v_maxy, r_peak, ra_center, dec_center,
                                     lnlike_background, m)
    v_los = model_rotation(cat["ra"], cat["dec"], v_sys, v_maxx, v_maxy, r_peak, ra_center, dec_center)
    sigma_los = model_dispersion(cat["ra"], cat["dec"], sigma_max, a, ra_center, dec_center)
    norm = cat["verr"] * cat["verr"] + sigma_los * sigma_los
    lc = -0.5 * np.log(2. * np.pi * norm) - 0.5 * np.power(cat["v"] - v_los, 2) / norm
    mx = np.max([lc, lnlike_background], axis=0)
    return mx + np.log(m * np.exp(lc - mx) + (1. - m) * np.exp(lnlike_background - mx))


def model_membership(cat, lnlike_cluster, lnlike_back, m):
    """calculate_membership_probabilities of the ModelFit classes (model.py:505-510, 680-687): with the max subtracted."""
    mx = np.max([lnlike_cluster, lnlike_back], axis=0)
    return m * np.exp(lnlike_cluster - mx) / (m * np.exp(lnlike_cluster - mx) + (1. - m) * np.exp(lnlike_back - mx))


def bounds_lnprior(values, lo, hi):
    """Runner.lnprior + Parameter.evaluate_lnprior (runner.py:182-217, parameter.py:684-705):
    0 inside the INCLUSIVE bounds, -inf outside; evaluated over every parameter, fixed ones too."""
    values = np.asarray(values, dtype=np.float64)
    bad = (values < lo) | (values > hi)
    return -np.inf if np.any(bad) else 0.0


def make_radial_bins(r, nstars=50, dlogr=0.2):
    """DataReader.make_radial_bins, utils/files/data_reader.py:97-120 (bin index per star, int16)."""
    r = np.asarray(r, dtype=np.float64)
    n = r.size
    sorted_indices = np.argsort(r)
    r_sorted = r[sorted_indices]
    bin_number = -np.ones(n, dtype=np.int16)
    i = 0
    while i < (n - nstars):
        j = min(n, i + nstars)
        while (np.log10(r_sorted[j]) - np.log10(r_sorted[i])) < dlogr:
            j += 1
            if j >= n:
                break
        bin_number[i:j] = np.max(bin_number) + 1
        i = j
    if (n - i) > 0.5 * nstars or np.max(bin_number) == -1:
        bin_number[i:] = np.max(bin_number) + 1
    else:
        bin_number[i:] = np.max(bin_number)
    return bin_number[sorted_indices.argsort()]


# ----------------------------------------------------------------------------- batched layer
def star_geometry_fixed(ra, dec, ra_center, dec_center):
    """sin(theta_i), cos(theta_i) for a fixed centre.

    A star exactly on the centre has dx = -r0 cos(dec) sin(+0) = -0.0 and dy = +0.0, and numpy's
    arctan2(+0, -0) = pi (constant.py:107), so the reference evaluates it at theta = pi:
    cos(theta) = copysign(1, dx), sin(theta) = 0."""
    dx, dy = calc_xy_offset(ra, dec, ra_center, dec_center)
    r = np.hypot(dx, dy)
    safe = np.where(r > 0, r, 1.0)
    return np.where(r > 0, dy / safe, 0.0), np.where(r > 0, dx / safe, np.copysign(1.0, dx))


def batched_constant_lnlike(cat, params, ra_center=None, dec_center=None,
                            lnlike_background=None, pmember=None):
    """(W, K) parameter table -> (W,) log-likelihoods, the algebra the HIP kernels implement.

    params columns: v_sys, sigma_max, v_maxx, v_maxy [, ra_center, dec_center].
    v_los = v_sys + v_maxx sin(theta) - v_maxy cos(theta)   (== v_max sin(theta - theta_0))
    """
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    v, verr = cat["v"][None, :], cat["verr"][None, :]
    vsys, sig, vx, vy = (params[:, k][:, None] for k in range(4))
    if params.shape[1] >= 6:
        sin_t = np.empty((params.shape[0], v.shape[1]))
        cos_t = np.empty_like(sin_t)
        for w in range(params.shape[0]):
            sin_t[w], cos_t[w] = star_geometry_fixed(cat["ra"], cat["dec"], params[w, 4], params[w, 5])
    else:
        s, c = star_geometry_fixed(cat["ra"], cat["dec"], ra_center, dec_center)
        sin_t, cos_t = s[None, :], c[None, :]
    v_los = vsys + vx * sin_t - vy * cos_t
    norm = verr * verr + sig * sig
    m = -0.5 * np.log(2. * np.pi * norm) - 0.5 * (v - v_los) ** 2 / norm
    if lnlike_background is None:
        return m.sum(axis=1)
    b, p = lnlike_background[None, :], pmember[None, :]
    mx = np.maximum(m, b)
    return (mx + np.log(p * np.exp(m - mx) + (1. - p) * np.exp(b - mx))).sum(axis=1)


def batched_constant_gb_lnlike(cat, params, ra_center=None, dec_center=None):
    """ConstantFitGB in batched form. params columns:
    v_sys, sigma_max, v_maxx, v_maxy, [ra_center, dec_center,] v_back, sigma_back, f_back."""
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    free = params.shape[1] == 9
    core = params[:, :6] if free else params[:, :4]
    vb, sb, fb = (params[:, -3 + k][:, None] for k in range(3))
    v, verr, rho = cat["v"][None, :], cat["verr"][None, :], cat["density"][None, :]
    nb = verr * verr + sb * sb
    b = -0.5 * np.log(2. * np.pi * nb) - 0.5 * (v - vb) ** 2 / nb
    mu = rho / (rho + fb)
    vsys, sig, vx, vy = (core[:, k][:, None] for k in range(4))
    if free:
        sin_t = np.empty((params.shape[0], v.shape[1]))
        cos_t = np.empty_like(sin_t)
        for w in range(params.shape[0]):
            sin_t[w], cos_t[w] = star_geometry_fixed(cat["ra"], cat["dec"], core[w, 4], core[w, 5])
    else:
        s, c = star_geometry_fixed(cat["ra"], cat["dec"], ra_center, dec_center)
        sin_t, cos_t = s[None, :], c[None, :]
    v_los = vsys + vx * sin_t - vy * cos_t
    norm = verr * verr + sig * sig
    m = -0.5 * np.log(2. * np.pi * norm) - 0.5 * (v - v_los) ** 2 / norm
    mx = np.maximum(m, b)
    return (mx + np.log(mu * np.exp(m - mx) + (1. - mu) * np.exp(b - mx))).sum(axis=1)
