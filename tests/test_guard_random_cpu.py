"""Randomised check of the fast-path range guard (csrc/mcd_guard.h) together with the fast formulations
(csrc/mcd_math.h), both compiled for the host: over catalogues and walkers spanning many orders of magnitude,
whenever the guard admits the fast path its result must agree with the plain (reference-literal) path; and the guard
must refuse the inputs the fast paths cannot represent."""
import numpy as np
import pytest

import emul_helper as emul
from conftest import rel_err

CENTRE = (56.345, -26.675)
MODELS = {0: 4, 1: 4, 2: 7, 3: 6, 4: 9, 5: 7, 6: 6}    # model id -> K (fixed centre)


def random_case(rng, model, n=300, w=6):
    scale_v = 10.0 ** rng.uniform(-1, 3.5)               # velocity scale 0.1 .. 3000 km/s
    scale_e = 10.0 ** rng.uniform(-3, 2.5)               # error scale
    sep = np.abs(rng.normal(0, 2.0 / 60.0, n))
    th = rng.uniform(-np.pi, np.pi, n)
    cat = {"ra": CENTRE[0] + sep * np.cos(th) / np.cos(np.radians(CENTRE[1])), "dec": CENTRE[1] + sep * np.sin(th),
           "v": rng.normal(0, scale_v, n), "verr": scale_e * rng.lognormal(0, 1.0, n)}
    cat["v"][:3] *= 50.0                                  # a few gross outliers
    cat["density"] = np.clip(rng.random(n), 0.0, 1.0)
    cat["density"][0] = 0.0
    cat["pmember"] = np.clip(rng.random(n) * 1.2 - 0.1, 0.0, 1.0)     # includes exact 0 and 1
    cat["lnlike_bg"] = -0.5 * (cat["v"] / (4 * scale_v)) ** 2 - np.log(4 * scale_v) - rng.uniform(0, 3, n)
    sig = scale_v * 10.0 ** rng.uniform(-2, 1, w)
    cols = [rng.normal(0, scale_v, w), sig]
    if model >= 3:
        cols.append(10.0 ** rng.uniform(-1, 3, w))        # a [arcsec]
    cols += [rng.normal(0, scale_v, w), rng.normal(0, scale_v, w)]
    if model >= 3:
        cols.append(10.0 ** rng.uniform(-1, 3, w))        # r_peak
    if model in (2, 4):
        cols += [rng.normal(0, scale_v, w), 4 * scale_v * 10.0 ** rng.uniform(-1, 1, w), rng.random(w)]
    if model == 5:
        cols.append(rng.random(w))
    return cat, np.stack(cols, axis=1)


@pytest.mark.parametrize("model", sorted(MODELS))
def test_guarded_fast_path_agrees_with_plain_path(model):
    rng = np.random.default_rng(100 + model)
    admitted = refused = 0
    for trial in range(60):
        cat, params = random_case(rng, model, n=300 if trial % 2 else 500, w=6 if trial % 2 else 70)
        assert params.shape[1] == MODELS[model]
        plain = emul.loglike(cat, params, model, CENTRE, 0, chunk_len=64)
        if not emul.fast_guard(cat, params, model, CENTRE):
            refused += 1
            continue
        admitted += 1
        level = emul.fast_level(cat, params, model, CENTRE)        # 2: the narrow-range BGFIXED variant is what would run
        fast = emul.loglike(cat, params, model, CENTRE, level, chunk_len=64)
        both = np.isfinite(plain) & np.isfinite(fast)
        assert np.array_equal(np.isfinite(plain), np.isfinite(fast)), (trial, plain, fast)
        assert rel_err(fast[both], plain[both]) < 1e-11, (trial, plain, fast)
    assert admitted >= 20, (admitted, refused)            # the guard must not be so strict that it is never used


def test_guard_refuses_what_the_fast_paths_cannot_represent():
    rng = np.random.default_rng(7)
    cat, params = random_case(rng, 0)
    assert emul.fast_guard(cat, params, 0, CENTRE)
    bad = params.copy()
    bad[0, 1] = np.nan
    assert not emul.fast_guard(cat, bad, 0, CENTRE)                       # NaN parameter
    bad = params.copy()
    bad[0, 1] = 1e40
    assert not emul.fast_guard(cat, bad, 0, CENTRE)                       # sigma^2 beyond 2^60
    zero = dict(cat, verr=cat["verr"].copy())
    zero["verr"][5] = 0.0
    z = params.copy()
    z[:, 1] = 0.0
    assert not emul.fast_guard(zero, z, 0, CENTRE)                        # norm can be exactly 0
    assert emul.fast_guard(zero, params, 0, CENTRE)                       # ... but not when sigma > 0
    inf_cat = dict(cat, v=cat["v"].copy())
    inf_cat["v"][0] = np.inf
    assert not emul.fast_guard(inf_cat, params, 0, CENTRE)                # non-finite data
    cat1, p1 = random_case(rng, 1)
    assert emul.fast_guard(cat1, p1, 1, CENTRE)
    far = dict(cat1, lnlike_bg=cat1["lnlike_bg"].copy())
    far["lnlike_bg"][3] = -2e5
    assert not emul.fast_guard(far, p1, 1, CENTRE)                        # background lnL outside +-1e5
    cat2, p2 = random_case(rng, 2)
    neg = p2.copy()
    neg[0, -1] = -0.1
    assert not emul.fast_guard(cat2, neg, 2, CENTRE)                      # f_back < 0 (prior would reject it anyway)
    cat3, p3 = random_case(rng, 3)
    a0 = p3.copy()
    a0[0, 2] = 0.0
    assert not emul.fast_guard(cat3, a0, 3, CENTRE)                       # a = 0: plain path reproduces the reference's limit
    # float32 fast mixtures: refused while a certain member (pmember == 1) is in the catalogue, admitted for tame columns
    assert cat1["pmember"].max() == 1.0 and not emul.fast_guard(cat1, p1, 1, CENTRE, f32=True)
    tame = dict(cat1, pmember=np.clip(cat1["pmember"], 0.0, 0.99), v=np.clip(cat1["v"], -100, 100), verr=np.full(len(cat1["v"]), 2.0),
                lnlike_bg=np.clip(cat1["lnlike_bg"], -8.0, 0.0))
    pt = p1.copy()
    pt[:, 0], pt[:, 1], pt[:, 2], pt[:, 3] = 1.0, 9.0, 2.0, -1.0
    assert emul.fast_guard(tame, pt, 1, CENTRE, f32=True)
    assert not emul.fast_guard(dict(tame, lnlike_bg=np.full(len(cat1["v"]), -70.0)), pt, 1, CENTRE, f32=True)   # y would leave float range
    assert emul.fast_guard(cat, params, 0, CENTRE, f32=True) in (True, False)


def test_float32_accuracy_domain():
    """Round 3: the domain in which the float32 modes keep their stated tolerances (csrc/mcd_guard.h:
    f32_domain; derived from tools/fuzz_f32.py, profiles/r03_fuzz_f32.txt) -- verdicts, condition numbers and reasons on the
    host; the enforcement in the library is tests/test_gpu_baseline_shapes.py."""
    from mcmc_dynamics_amd import synthetic
    from mcmc_dynamics_amd.background import Gaussian
    c = synthetic.make_catalog(20000, config=3, background=True)
    centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
    c["lnlike_bg"] = Gaussian(20.0, 40.0)(c["v"], c["verr"])
    pos = synthetic.make_walkers(64, ["v_sys", "sigma_max", "v_maxx", "v_maxy"], c["truth"], config=3)
    inside, kv, kt, sep, why = emul.f32_domain(c, pos, 1, centre)
    assert inside and why == "" and 5 < kv < 60 and kt == 0.0                     # the C3 shape is well inside
    expect = (np.abs(c["v"]).max() + np.max(np.abs(pos[:, 0]) + np.abs(pos[:, 2]) + np.abs(pos[:, 3]))) / \
        np.sqrt((c["verr"] ** 2).min() + (pos[:, 1] ** 2).min())
    assert abs(kv - expect) < 1e-9 * expect
    # a systemic velocity of 3000 km/s with the same dispersion: the residual is a difference of large numbers
    far = dict(c, v=c["v"] + 3000.0)
    shifted = pos.copy()
    shifted[:, 0] += 3000.0
    inside, kv, _, _, why = emul.f32_domain(far, shifted, 1, centre)
    assert not inside and kv > 96 and "exceeds 96" in why
    # one walker at sigma -> 0
    tiny = pos.copy()
    tiny[3, 1] = 1e-3
    assert not emul.f32_domain(c, tiny, 1, centre)[0]
    # a certain member leaves the float32 mixture ranges
    pm = c["pmember"].copy()
    pm[5] = 1.0
    inside, _, _, _, why = emul.f32_domain(dict(c, pmember=pm), pos, 1, centre)
    assert not inside and "pmember <= 1 - 2^-20" in why
    # free centre: compact catalogue + strong rotation is outside, a wide field with slow rotation inside
    free = np.column_stack([pos, np.full(64, centre[0]), np.full(64, centre[1])])
    inside, _, kt, sep, why = emul.f32_domain(c, free, 0, None)
    assert sep > 0 and kt > 0
    assert inside == (kt <= 2e-5) and (inside or "tangent-plane" in why)
    rng = np.random.default_rng(2)
    wide = dict(c)
    sepd = np.abs(rng.normal(0, 0.5, 20000)) + 0.05                                  # degrees
    th = rng.uniform(-np.pi, np.pi, 20000)
    wide["ra"] = centre[0] + sepd * np.cos(th) / np.cos(np.radians(centre[1]))
    wide["dec"] = centre[1] + sepd * np.sin(th)
    slow = free.copy()
    slow[:, 2:4] *= 0.05
    inside, _, kt2, sep2, _ = emul.f32_domain(wide, slow, 0, None)
    assert inside and kt2 < kt and sep2 > sep
