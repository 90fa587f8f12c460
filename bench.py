#!/usr/bin/env python3
"""Throughput benchmark of the per-walker log-likelihood hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: the log-likelihood of all 256 walkers over every
star of the catalogue (``mcd_loglike_enqueue``: main kernel + fixed-order reduction [+ one RCCL
all-reduce of 256 doubles when N > 1]).  Default workload = the configuration BASELINE.json's metric
is quoted on (C3 of SURVEY.md section 8): 1e6 synthetic stars per GPU x 256 walkers, rotation +
dispersion + fixed single-Gaussian background mixture, float64.  Star records and the walker table are
resident in HBM before the timed region.  For N > 1 every rank holds its own 1e6-star shard (weak
scaling) and the per-walker partial sums are all-reduced over xGMI.

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip table)
# f64 VALU issue model (the resource that actually binds these kernels):
#   slots per term of the fast-path inner loops from the gfx950 ISA (tools/isa_mix.py; bgfixed / bggauss = the narrow-range variants
#   the guard selects for the bench catalogue), one slot = one f64 wave-instruction
#   per SIMD = 2.33 ns on the fully occupied chip (tools/valu_rate_probe.hip).
VALU_SLOTS_PER_TERM = {"const": 8.56, "bgfixed": 26.65, "bggauss": 49.55, "profile": 27.91}
VALU_SLOT_NS = 2.33
COLLECTIVE_TIMEOUT_S = 180
N_SIMD = 256 * 4

WORKLOADS = {
    # name: (description, stars per GPU, walkers, model, algorithmic bytes per term, config number)
    "c2": ("C2: 1e5 synthetic stars x 256 walkers, rotation+dispersion, fixed centre", 100000, 256, "const", 32, 2),
    "c3": ("C3: 1e6 synthetic stars x 256 walkers, rotation+dispersion + fixed single-Gaussian background "
           "mixture (ConstantFit + background.Gaussian), fixed centre", 1000000, 256, "bgfixed", 48, 3),
    "c3gb": ("C3-GB: 1e6 synthetic stars x 256 walkers, ConstantFitGB (per-walker Gaussian background)", 1000000, 256,
             "bggauss", 40, 3),
    "c3const": ("1e6 synthetic stars x 256 walkers, rotation+dispersion without background", 1000000, 256, "const",
                32, 3),
    "c4": ("C4: 1e7 synthetic stars sharded over the GPUs (strong scaling), rotation+dispersion", 10000000, 256,
           "const", 32, 4),
    "m1": ("next-row ModelFit: 1e6 synthetic stars x 256 walkers, Lynden-Bell rotation curve + Plummer dispersion "
           "profile (analysis/model.py), fixed centre", 1000000, 256, "profile", 32, 3),
    "c5": ("C5: radial-binned dispersion profile (make_radial_bins nstars=1000, dlogr=0.05), 1e6 synthetic stars x 512 "
           "walkers per bin, rotation+dispersion, one segmented launch for all bins", 1000000, 512, "const", 32, 5),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--clock-ramp-seconds", type=float, default=0.4,
                    help="untimed launches before the W warm-up steps so that the GPU is at its sustained clocks")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--stars", type=int, default=None, help="stars per GPU (override)")
    ap.add_argument("--walkers", type=int, default=None)
    ap.add_argument("--precision", default="f64", choices=["f64", "f32", "f32acc64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline sample budget")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the all-host-cores pool baseline")
    ap.add_argument("--no-mcmc", action="store_true", help="skip the sampler-driven end-to-end figure (profiling runs)")
    return ap.parse_args()


def build_catalog(native, ctx, synthetic, oracle, cat, model, precision="f64", bin_offsets=None):
    centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
    if model == "const":
        return native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST, centre=centre,
                              precision=precision, bin_offsets=bin_offsets)
    if model == "bgfixed":
        # background/gaussian.py:23-28 evaluated once on the host (one-off precompute)
        from mcmc_dynamics_amd.background import Gaussian
        lnbg = Gaussian(synthetic.TRUTH["v_back"], synthetic.TRUTH["sigma_back"])(cat["v"], cat["verr"])
        return native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGFIXED,
                              centre=centre, lnlike_bg=lnbg, pmember=cat["pmember"])
    if model == "profile":
        return native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_PROFILE, centre=centre)
    return native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGGAUSS,
                          centre=centre, density=cat["density"])


def cpu_baseline(cat, pos, model, budget_s):
    """The oracle's faithful op-for-op NumPy restatement (one walker per call, as the reference's
    ``Runner.lnprob``), timed on one host core over a bounded number of walkers."""
    from oracle import lnprob_numpy as oracle
    from mcmc_dynamics_amd import synthetic
    centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
    lnbg = None
    if model == "bgfixed":
        lnbg = oracle.gaussian_background(cat["v"], cat["verr"], synthetic.TRUTH["v_back"], synthetic.TRUTH["sigma_back"])

    def one(row):
        if model == "bggauss":
            return oracle.faithful_constant_gb_lnlike(cat, row[0], row[1], row[2], row[3], centre[0], centre[1],
                                                      row[4], row[5], row[6])
        if model == "profile":          # C-ABI column order: v_sys, sigma_max, a, v_maxx, v_maxy, r_peak
            return oracle.faithful_model_lnlike(cat, row[0], row[1], row[2], row[3], row[4], row[5], centre[0], centre[1])
        return oracle.faithful_constant_lnlike(cat, row[0], row[1], row[2], row[3], centre[0], centre[1],
                                               lnlike_background=lnbg, pmember=cat.get("pmember") if lnbg is not None else None)

    one(pos[0])                                   # warm the allocator / caches
    n, t0 = 0, time.perf_counter()
    vals = []
    while n < len(pos):
        vals.append(one(pos[n]))
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    n_stars = len(cat["v"])
    return {"value": n_stars * n / dt, "unit": "terms/s", "cores": 1, "kind": "port",
            "sample": "{0} walkers x {1} stars, one walker per call, NumPy {2} (oracle/lnprob_numpy.py faithful path), "
                      "{3:.1f} s".format(n, n_stars, np.__version__, dt)}, np.array(vals)


_POOL_STATE = {}


def _pool_worker(rows):
    """One walker per call in a forked worker (mirrors the reference's `n_threads` pool over walkers,
    runner.py:398-403).  Workers inherit the catalogue from the parent and never touch the GPU."""
    one = _POOL_STATE["one"]
    t0 = time.perf_counter()
    for row in rows:
        one(row)
    return time.perf_counter() - t0


def cpu_baseline_all_cores(cat, pos, model, per_walker_s, budget_s):
    """The same NumPy port on every host core this process may use.  Must run BEFORE the HIP runtime is
    initialised (plain fork, no exec); bounded by a hard timeout so that it can never stall the bench."""
    import multiprocessing as mp
    from oracle import lnprob_numpy as oracle
    from mcmc_dynamics_amd import synthetic
    centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
    lnbg = None
    if model == "bgfixed":
        lnbg = oracle.gaussian_background(cat["v"], cat["verr"], synthetic.TRUTH["v_back"], synthetic.TRUTH["sigma_back"])

    def one(row):
        if model == "bggauss":
            return oracle.faithful_constant_gb_lnlike(cat, row[0], row[1], row[2], row[3], centre[0], centre[1],
                                                      row[4], row[5], row[6])
        return oracle.faithful_constant_lnlike(cat, row[0], row[1], row[2], row[3], centre[0], centre[1],
                                               lnlike_background=lnbg, pmember=cat.get("pmember") if lnbg is not None else None)

    _POOL_STATE["one"] = one
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))
    per_worker = max(1, min(len(pos) // cores, int(budget_s / max(per_walker_s, 1e-3))))
    jobs = [pos[i * per_worker:(i + 1) * per_worker] for i in range(cores)]
    pool = mp.get_context("fork").Pool(cores)
    try:
        pool.map_async(_pool_worker, [pos[:1]] * cores).get(timeout=60)           # warm-up outside the timed region
        t0 = time.perf_counter()
        busy = pool.map_async(_pool_worker, jobs).get(timeout=max(60.0, 6.0 * budget_s))
        wall = time.perf_counter() - t0
    finally:
        pool.terminate()
        pool.join()
    n = per_worker * cores
    return {"value": len(cat["v"]) * n / wall, "unit": "terms/s", "cores": cores, "kind": "port",
            "sample": "{0} walkers x {1} stars over {2} forked worker processes (one walker per call each), wall {3:.1f} s "
                      "(slowest worker busy {4:.1f} s)".format(n, len(cat["v"]), cores, wall, max(busy))}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and world > 1:
        raise SystemExit("WORLD_SIZE={0} does not match --gpus {1}".format(world, args.gpus))

    from mcmc_dynamics_amd import synthetic
    desc, n_stars, n_walkers, model, bytes_per_term, config = WORKLOADS[args.workload]
    strong = args.workload == "c4"
    if args.stars is not None:
        n_stars = args.stars
    if args.walkers is not None:
        n_walkers = args.walkers

    # Extra CPU figure (all host cores) -- forked workers, so it runs before anything initialises the HIP runtime.
    all_cores = None
    if (world == 1 and not args.no_cpu_baseline and not args.no_cpu_all_cores
            and args.workload in ("c2", "c3", "c3gb", "c3const")):
        try:
            cat0 = synthetic.make_catalog(n_stars, config=config, seed=synthetic.CATALOG_SEED_BASE + config,
                                          background=(model != "const"))
            names0 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"] + (["v_back", "sigma_back", "f_back"] if model == "bggauss" else [])
            pos0 = synthetic.make_walkers(n_walkers, names0, cat0["truth"], config=config)
            probe, _ = cpu_baseline(cat0, pos0[:2], model, 1.0)
            all_cores = cpu_baseline_all_cores(cat0, pos0, model, n_stars / probe["value"], min(args.cpu_seconds, 10.0))
            del cat0
        except Exception as exc:                          # never let the extra baseline break the bench line
            all_cores = {"error": repr(exc)}

    from mcmc_dynamics_amd import _native as native
    n_visible = int(os.environ.get("MCD_VISIBLE_DEVICES", "0")) or None      # testing aid: fold ranks onto fewer devices
    if n_visible:
        local_rank = local_rank % n_visible
    dist = None
    if world > 1:
        import torch.distributed as dist           # host-side rendezvous only (gloo); the data path is RCCL in the library
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        rccl_note = "ncclAllReduce(sum, f64, count = walkers) per step on the catalogue stream"
        try:
            uid = [native.Context.unique_id() if rank == 0 else None]
        except native.NativeError as exc:
            uid = [None]
            rccl_note = "unavailable: {0}".format(exc)
        dist.broadcast_object_list(uid, src=0)
        try:
            if uid[0] is None:
                raise native.NativeError("no RCCL unique id")
            ctx = native.Context(rank=rank, n_ranks=world, unique_id=uid[0], device=local_rank)
            ok = 1
        except native.NativeError as exc:
            ok = 0
            rccl_note = "unavailable: {0}".format(exc)
        import torch
        flag = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            # Degraded mode (recorded in the JSON line): every rank still evaluates its own star shard and the per-walker
            # partial sums are exchanged over the host process group instead (see the timed loop).
            if ok:
                ctx.close()
            ctx = native.Context(n_devices=1, device_ids=[local_rank])
            if not rccl_note.startswith("unavailable"):
                rccl_note = "unavailable on another rank"
    else:
        rccl_note = None
        ctx = native.Context(n_devices=1, device_ids=[local_rank])

    # ---- synthetic catalogue shard of this rank; identical walkers on every rank
    if strong:
        full = synthetic.make_catalog(n_stars, config=config, background=(model != "const"))
        lo, hi = n_stars * rank // world, n_stars * (rank + 1) // world
        cat = {k: (v[lo:hi] if isinstance(v, np.ndarray) else v) for k, v in full.items()}
        del full
        total_stars = n_stars
    else:
        cat = synthetic.make_catalog(n_stars, config=config, seed=synthetic.CATALOG_SEED_BASE + config + 1000 * rank,
                                     background=(model != "const"))
        total_stars = n_stars * world
    truth = cat["truth"]
    if dist is not None and not strong:              # walkers are drawn around rank 0's truth everywhere
        box = [truth if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        truth = box[0]
    names = ["v_sys", "sigma_max", "v_maxx", "v_maxy"] + (["v_back", "sigma_back", "f_back"] if model == "bggauss" else [])
    pos = synthetic.make_walkers(n_walkers, names, truth, config=config)
    if model == "profile":              # insert a (30 arcsec) and r_peak (60 arcsec) balls: run_tests.py:36-37
        rng_m = np.random.default_rng(synthetic.WALKER_SEED_BASE + 100 + config)
        a_col = 30.0 * (1.0 + 0.05 * rng_m.normal(size=n_walkers))
        rp_col = 60.0 * (1.0 + 0.05 * rng_m.normal(size=n_walkers))
        pos = np.column_stack([pos[:, 0], pos[:, 1], a_col, pos[:, 2], pos[:, 3], rp_col])

    bin_offsets, n_bins = None, 1
    if args.workload == "c5":
        # DataReader.make_radial_bins (utils/files/data_reader.py:71-120), stars sorted by bin; every bin gets its
        # own copy of the walker ensemble (B independent posteriors per launch)
        from mcmc_dynamics_amd import DataReader
        reader = DataReader({k: cat[k] for k in ("ra", "dec", "v", "verr")})
        reader.make_radial_bins(synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG, nstars=1000, dlogr=0.05)
        srt, bin_offsets = reader.sorted_by_bin()
        cat = dict(cat, **{k: srt.data[k] for k in ("ra", "dec", "v", "verr")})
        n_bins = len(bin_offsets) - 1
        pos = np.ascontiguousarray(np.broadcast_to(pos, (n_bins,) + pos.shape))
        bytes_per_term = bytes_per_term if args.precision == "f64" else bytes_per_term // 2

    gpu_cat = build_catalog(native, ctx, synthetic, None, cat, model, args.precision, bin_offsets)
    gpu_cat.upload_params(pos)
    abandoned = None
    if dist is not None and not rccl_note.startswith("unavailable"):
        # Watchdog for the first collective (RCCL builds its xGMI rings lazily inside it): a rank whose all-reduce has
        # not completed after COLLECTIVE_TIMEOUT_S abandons that context and every rank drops to the degraded mode, so
        # that a fabric problem yields a marked bench line instead of a hung job.
        import threading
        import torch
        done = threading.Event()

        def first_step():
            gpu_cat.enqueue()
            gpu_cat.sync()
            done.set()
        worker = threading.Thread(target=first_step, daemon=True)
        worker.start()
        worker.join(COLLECTIVE_TIMEOUT_S)
        flag = torch.tensor([1 if done.is_set() else 0], dtype=torch.int64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag[0]) == 0:
            abandoned = (gpu_cat, ctx)                   # never closed: its stream may be blocked for good
            rccl_note = "unavailable: first all-reduce did not complete within {0} s on some rank".format(COLLECTIVE_TIMEOUT_S)
            ctx = native.Context(n_devices=1, device_ids=[local_rank])
            gpu_cat = build_catalog(native, ctx, synthetic, None, cat, model, args.precision, bin_offsets)
            gpu_cat.upload_params(pos)
    gpu_cat.set_option("timing", 2)
    gpu_cat.set_option("timing_reserve", min(65536, max(args.steps, args.warmup)))   # no hipEventCreate inside the timed region
    # the event pair costs ~6 us per step between back-to-back kernels (tools/event_cost_probe.py): the kernel time is
    # sampled on every 8th launch of the timed region, not on each
    stride = 8 if args.steps >= 64 else 1
    gpu_cat.set_option("timing_stride", stride)

    def barrier():
        gpu_cat.sync()
        if dist is not None:
            dist.barrier()

    # Clock ramp (untimed, before the W warm-up steps): the GPU needs some 100 ms of load to reach its sustained clocks --
    # measured kernel time 270 us in the first 20 launches after idle, 222 us once settled.  Every rank runs the same
    # number of launches (the all-reduce is collective).
    ramp_steps = 0
    if args.clock_ramp_seconds > 0:
        gpu_cat.enqueue()
        gpu_cat.sync()
        t_r = time.perf_counter()
        for _ in range(8):
            gpu_cat.enqueue()
        gpu_cat.sync()
        per_step = max((time.perf_counter() - t_r) / 8, 1e-6)
        ramp_steps = int(min(20000, args.clock_ramp_seconds / per_step))
        if dist is not None:
            import torch
            t = torch.tensor([ramp_steps], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ramp_steps = int(t[0])
        for _ in range(ramp_steps):
            gpu_cat.enqueue()
    for _ in range(args.warmup):
        gpu_cat.enqueue()
    barrier()
    gpu_cat.timing_collect()                         # drop ramp and warm-up launches

    # Degraded mode (RCCL unavailable): the exchange is NOT skipped -- every step fetches the rank's partial sums and
    # all-reduces them over the host process group (gloo), which serialises the steps; the JSON line says so.
    host_exchange = dist is not None and rccl_note.startswith("unavailable")
    t0 = time.perf_counter()
    if host_exchange:
        import torch
        for _ in range(args.steps):
            gpu_cat.enqueue()
            part = torch.from_numpy(gpu_cat.fetch())
            dist.all_reduce(part, op=dist.ReduceOp.SUM)
    else:
        for _ in range(args.steps):
            gpu_cat.enqueue()
    gpu_cat.sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    kernel_ms_total, n_launch = gpu_cat.timing_collect()
    result = gpu_cat.fetch()
    info = gpu_cat.launch_info()
    info["kernel_family"] = {0: "plain", 1: "fast", 2: "fast, narrow-range variant", -1: "none"}[gpu_cat.fast_level]

    # blocking C-ABI call (host params in, host results out) for the PCIe/sync-inclusive rate
    gpu_cat.set_option("timing", 0)
    t1 = time.perf_counter()
    n_sync = max(5, min(50, args.steps))
    for _ in range(n_sync):
        gpu_cat.loglike(pos)
    sync_call = (time.perf_counter() - t1) / n_sync

    # sampler-driven end-to-end rate (1 GPU only): the built-in stretch move makes two blocking calls of W/2
    # proposals per step, exactly the batching emcee's default move produces (hard parts)
    mcmc = None
    if world == 1 and n_bins == 1 and not args.no_mcmc and model != "profile":
        import logging
        from mcmc_dynamics_amd import DataReader, Gaussian
        from mcmc_dynamics_amd.analysis import ConstantFit, ConstantFitGB
        logging.getLogger("mcmc_dynamics_amd").setLevel(logging.ERROR)
        cols = {k: cat[k] for k in ("ra", "dec", "v", "verr", "density", "pmember") if k in cat}
        if model == "bggauss":
            fit = ConstantFitGB(DataReader(cols), context=ctx)
        else:
            fit = ConstantFit(DataReader(cols), context=ctx,
                              background=Gaussian(synthetic.TRUTH["v_back"], synthetic.TRUTH["sigma_back"]) if model == "bgfixed" else None)
        fit.parameters["ra_center"].set(value=synthetic.CENTER_RA_DEG, fixed=True)     # pattern of bin/run_tests.py:92-93
        fit.parameters["dec_center"].set(value=synthetic.CENTER_DEC_DEG, fixed=True)
        sampler = fit._make_sampler(n_walkers)             # emcee if importable, built-in stretch move otherwise
        state = sampler.run_mcmc(pos, 3)
        t2 = time.perf_counter()
        n_mcmc = 30
        sampler.run_mcmc(tuple(state)[0], n_mcmc)
        dt = time.perf_counter() - t2
        mcmc = {"steps_per_s": n_mcmc / dt, "terms_per_s": float(len(cat["v"])) * n_walkers * n_mcmc / dt,
                "driver": type(sampler).__module__ + "." + type(sampler).__name__,
                "posterior": type(fit).__name__ + ".lnprob_batch", "calls_per_step": 2, "walkers_per_call": n_walkers // 2,
                "acceptance_fraction": float(np.mean(sampler.acceptance_fraction))}
        fit.close()

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if abandoned is not None:
            os._exit(0)                                  # skip destructors of the abandoned (possibly blocked) context
        return

    terms_per_step = float(total_stars) * n_walkers
    value = terms_per_step * args.steps / elapsed
    kernel_s = kernel_ms_total * 1e-3 / max(1, n_launch)
    local_terms = float(len(cat["v"])) * n_walkers
    achieved = local_terms * bytes_per_term / kernel_s / 1e9

    traffic = None                                    # PMC bytes per launch, measured for the default shape of a workload
    if args.stars is None and args.walkers is None and args.precision == "f64" and not strong:
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = json.load(f).get(args.workload)
        except Exception:
            pass

    out = {
        "metric": "star-walker log-L terms/sec", "value": value, "unit": "terms/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "clock_ramp_steps": ramp_steps,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": desc, "stars_per_gpu": len(cat["v"]), "stars_total": total_stars, "walkers": n_walkers,
                   "likelihood": model, "parallelism": ("stars sharded over {0} rank(s); ".format(world) +
                                   ("RCCL all-reduce of {0} doubles per step".format(n_walkers)
                                    if rccl_note and not rccl_note.startswith("unavailable") else "RCCL unavailable: per-step exchange over the host process group (gloo all-reduce of the fetched partial sums)"))
                   if world > 1 else "1 GPU"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "algorithmic_bytes_per_term": bytes_per_term, "kernel_us": kernel_s * 1e6,
                     "kernel_us_sampled_launches": int(n_launch),
                     "kernel": "mcd::loglike_kernel", "walker_tile": info["walker_tile"],
                     "compulsory_bytes_per_launch": len(cat["v"]) * info["record_bytes"],
                     "note": "streaming-model bytes (each walker's sum reads every star record once, SURVEY 8(d)); "
                             "the kernel reuses one scalar record load for 64 walkers, so frac > 1 means register "
                             "reuse and the binding resource is f64 VALU issue, not HBM"},
        "valu_f64_model": None if args.precision != "f64" else {
            "slots_per_term": VALU_SLOTS_PER_TERM[model], "slot_ns": VALU_SLOT_NS,
            "predicted_kernel_us": VALU_SLOTS_PER_TERM[model] * VALU_SLOT_NS * (local_terms / 64) / N_SIMD * 1e-3,
            "measured_over_predicted": kernel_s * 1e6 / (VALU_SLOTS_PER_TERM[model] * VALU_SLOT_NS * (local_terms / 64) / N_SIMD * 1e-3),
            "note": "the kernel is bound by f64 vector-instruction issue; predicted = instruction mix of the inner loop "
                    "priced at the measured issue rate of a fully occupied MI355X (prologue, final log and launch tail "
                    "not included)"},
        "hbm_algorithmic_GBps": value * bytes_per_term / 1e9,
        "hbm_roofline_frac": value * bytes_per_term / 1e9 / HBM_PEAK_GBPS / world,
        "sync_call_terms_per_s": local_terms * world / sync_call,
        "sync_call_us": sync_call * 1e6,
        "launch": info,
        "mcmc_end_to_end": mcmc,
        "collective": rccl_note,
    }
    out["dtype"] = {"f64": "f64", "f32": "f32", "f32acc64": "f32 terms, f64 accumulation"}[args.precision]
    if n_bins > 1:
        out["config"]["radial_bins"] = n_bins
        out["config"]["outputs_per_step"] = n_bins * n_walkers
    if world == 1 and not args.no_cpu_baseline:
        base, vals = cpu_baseline(cat, pos[0] if n_bins > 1 else pos, model, args.cpu_seconds)
        out["cpu_baseline"] = base
        got = result.sum(axis=0) if n_bins > 1 else result          # same walkers in every bin: sum over bins == un-binned
        err = np.max(np.abs(got[:len(vals)] - vals) / np.abs(vals))
        out["gpu_vs_cpu_port_max_rel_err"] = float(err)
        out["speedup_vs_cpu_1core"] = value / base["value"]
        if all_cores is not None:
            out["cpu_baseline_all_cores"] = all_cores
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if abandoned is not None:
        os._exit(0)


if __name__ == "__main__":
    main()
