#!/usr/bin/env python3
"""End-to-end example in the spirit of the reference's bin/run_tests.py: a synthetic cluster (rotation + dispersion,
20 % background stars), `ConstantFit` with a fixed-Gaussian background and a fixed centre, an MCMC run on the GPU and
the best-fit table.  Needs an MI355X (gfx950) and the built library (make -C mcmc_dynamics_amd/csrc).

    python examples/run_constant_fit.py [--stars 100000] [--walkers 128] [--steps 300]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mcmc_dynamics_amd import DataReader, Gaussian, synthetic          # noqa: E402
from mcmc_dynamics_amd.analysis import ConstantFit                      # noqa: E402
from mcmc_dynamics_amd.utils.coordinates import get_amplitude_and_angle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stars", type=int, default=100000)
    ap.add_argument("--walkers", type=int, default=128)
    ap.add_argument("--steps", type=int, default=300)
    a = ap.parse_args()

    cat = synthetic.make_catalog(a.stars, config=3, background=True)     # truth: sigma = 10 km/s, v_max = 5 km/s
    data = DataReader({k: cat[k] for k in ("ra", "dec", "v", "verr", "pmember")})
    fit = ConstantFit(data, background=Gaussian(synthetic.TRUTH["v_back"], synthetic.TRUTH["sigma_back"]))
    fit.parameters["ra_center"].set(value=synthetic.CENTER_RA_DEG, fixed=True)      # pattern of bin/run_tests.py:92-93
    fit.parameters["dec_center"].set(value=synthetic.CENTER_DEC_DEG, fixed=True)

    t0 = time.perf_counter()
    sampler = fit(n_walkers=a.walkers, n_steps=a.steps, n_out=None, prefix=None)
    dt = time.perf_counter() - t0
    chain = np.asarray(sampler.chain)                                    # (walkers, steps, parameters), as in the reference
    print("{0} steps x {1} walkers on {2} stars in {3:.2f} s = {4:.3g} star-walker terms/s".format(
        a.steps, a.walkers, a.stars, dt, a.stars * float(a.walkers) * a.steps / dt))
    print("acceptance fraction {0:.2f}".format(float(np.mean(sampler.acceptance_fraction))))
    best = fit.compute_bestfit_values(chain, n_burn=a.steps // 2)
    print(best)
    print("truth:", {k: cat["truth"][k] for k in ("v_sys", "sigma_max", "v_maxx", "v_maxy")})
    pars = fit.convert_to_parameters(chain, n_burn=a.steps // 2)
    print("rotation amplitude and angle:")
    print(get_amplitude_and_angle(pars)[0])
    fit.close()


if __name__ == "__main__":
    main()
