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
scaling) and the per-walker partial sums are all-reduced over xGMI.  Every run (any N, default
workload) also times the north-star strong-scaling case, C4: ONE 1e7-star catalogue sharded over the N
ranks (``c4_strong`` in the JSON line), so that a sweep over N yields the strong-scaling curve.

No PyTorch anywhere: the launcher's environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*) is read
directly, rank 0's RCCL unique id, barriers and max-over-ranks travel over ``mcmc_dynamics_amd.hostgroup``
(plain TCP), so the process runs on the ROCm libraries ``libmcd_hip.so`` was built against.

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` and `cpu_baseline`.
If the collective cannot be set up or its first all-reduce hangs, rank 0 prints a line with
``"degraded": true`` and ``"value": null`` and EVERY rank exits with status 3: no number is reported
from a process that sits beside a blocked GPU context.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip table)
# f64 VALU issue peak, the roof that binds these kernels: 256 CUs x 4 SIMDs, one wave64 f64
# vector instruction per 4 cycles per SIMD, 2.4 GHz max clock (MI355X_MICROARCH.md: chip table, wave scheduling)
N_SIMD = 256 * 4
MAX_CLOCK_HZ = 2.4e9
VALU_F64_PEAK = N_SIMD * MAX_CLOCK_HZ / 4.0          # 6.144e11 wave-instructions / s
# float32 vector instructions issue at twice that rate: tools/valu_rate_probe.hip measures 1.04 ns per wave64 v_fma_f32 /
# v_mul_f32 / v_add_f32 per SIMD against 2.35 ns for v_fma_f64 in the same run (profiles/r03_valu_rate_probe.txt), i.e. 2
# cycles against 4 -- the chip table's 157.3 TFLOP/s of vector f32 against 78.6 TFLOP/s of f64; v_pk_fma_f32 is no faster
# per float.  The roof of the MCD_F32 / MCD_F32_ACC64 kernels (the f64 additions of the latter cost two such slots).
VALU_F32_PEAK = N_SIMD * MAX_CLOCK_HZ / 2.0          # 1.2288e12 wave-instructions / s
COLLECTIVE_TIMEOUT_S = 180
EXIT_COLLECTIVE = 3
ISA_MIX = os.path.join(ROOT, "mcmc_dynamics_amd", "csrc", "isa_mix.json")

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
NAMES4 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]


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
    ap.add_argument("--no-c4-strong", action="store_true", help="skip the 1e7-star strong-scaling sub-record")
    ap.add_argument("--c4-stars", type=int, default=10000000, help="total stars of the strong-scaling sub-record")
    return ap.parse_args()


def build_catalog(native, ctx, synthetic, cat, model, precision="f64", bin_offsets=None):
    g = _build_catalog(native, ctx, synthetic, cat, model, precision, bin_offsets)
    # tuning aid (tools/ab_option.sh): MCD_BENCH_OPTIONS="key=value,..." goes to mcd_set_option; echoed in the bench line
    for item in filter(None, os.environ.get("MCD_BENCH_OPTIONS", "").split(",")):
        key, value = item.split("=")
        g.set_option(key.strip(), int(value))
    return g


def _build_catalog(native, ctx, synthetic, cat, model, precision, bin_offsets):
    centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
    if model == "const":
        return native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST, centre=centre,
                              precision=precision, bin_offsets=bin_offsets)
    if model == "bgfixed":
        # background/gaussian.py:23-28 evaluated once on the host (one-off precompute)
        from mcmc_dynamics_amd.background import Gaussian
        lnbg = Gaussian(synthetic.TRUTH["v_back"], synthetic.TRUTH["sigma_back"])(cat["v"], cat["verr"])
        return native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGFIXED,
                              centre=centre, lnlike_bg=lnbg, pmember=cat["pmember"], precision=precision)
    if model == "profile":
        return native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_PROFILE, centre=centre,
                              precision=precision)
    return native.Catalog(ctx, cat["ra"], cat["dec"], cat["v"], cat["verr"], model=native.MODEL_CONST_BGGAUSS,
                          centre=centre, density=cat["density"], precision=precision)


# ---------------------------------------------------------------------------------------------- CPU baseline (oracle)
def _cpu_one(cat, model):
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
    return one


def cpu_baseline(cat, pos, model, budget_s):
    """The oracle's faithful op-for-op NumPy restatement (one walker per call, as the reference's
    ``Runner.lnprob``), timed on one host core over a bounded number of walkers."""
    one = _cpu_one(cat, model)
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
    _POOL_STATE["one"] = _cpu_one(cat, model)
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


# ---------------------------------------------------------------------------------------------- helpers
def mapped_libraries():
    """Which HIP runtime / RCCL / kernel library this process actually mapped (from /proc/self/maps)."""
    found = {}
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.split()[-1]
                base = os.path.basename(path)
                for key in ("libamdhip64", "librccl", "libfake_rccl", "libmcd_hip", "libhsa-runtime64"):
                    if base.startswith(key):
                        found[key] = path
    except OSError:
        pass
    return found


def isa_counts(model):
    """VALU wave-instructions per star-walker term of the model's hot loop, as emitted by tools/isa_mix.py at build time
    (mcmc_dynamics_amd/csrc/isa_mix.json; regenerated here when it does not belong to the sources in the tree)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import isa_mix
        want = isa_mix.source_hash()
        data = None
        if os.path.exists(ISA_MIX):
            with open(ISA_MIX) as f:
                data = json.load(f)
        if data is None or data.get("source_sha16") != want:
            res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py"), "--json", ISA_MIX],
                                 capture_output=True, timeout=600)
            if res.returncode != 0:
                return None, "tools/isa_mix.py failed: " + res.stderr.decode()[-200:]
            with open(ISA_MIX) as f:
                data = json.load(f)
        row = data["kernels"].get(model)
        if row is None:
            return None, "no entry for " + model
        return row, "mcmc_dynamics_amd/csrc/isa_mix.json (tools/isa_mix.py at build, sources sha16 {0})".format(data["source_sha16"])
    except Exception as exc:                               # the roofline block degrades to null, the bench line survives
        return None, repr(exc)


def smi_snapshot():
    try:
        res = subprocess.run(["rocm-smi", "--showuse", "--showmemuse", "--showpids", "--json"], capture_output=True, timeout=3)
        return res.stdout.decode()[-1500:]
    except Exception as exc:
        return repr(exc)


def fail_collective(rank, world, args, stage, reason, group=None):
    """The collective could not be set up or did not complete: rank 0 prints a marked line with no value; every rank
    leaves with a non-zero status WITHOUT running destructors (the GPU context may be blocked for good).  The other ranks
    linger a few seconds so that the launcher does not tear rank 0 down before its line is out."""
    sys.stderr.write("bench.py rank {0}: collective failure at {1}: {2}\n".format(rank, stage, reason))
    sys.stderr.flush()
    if rank == 0:
        line = {"metric": "star-walker log-L terms/sec", "value": None, "unit": "terms/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "degraded": True,
                "config": {"workload": WORKLOADS[args.workload][0]},
                "failure": {"stage": stage, "reason": reason, "rank_reporting": rank, "libraries": mapped_libraries(),
                            "env": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_DEBUG", "HIP_VISIBLE_DEVICES",
                                                                   "ROCR_VISIBLE_DEVICES", "MASTER_ADDR", "MASTER_PORT")}},
                "roofline": None, "cpu_baseline": None}
        print(json.dumps(line), flush=True)
        sys.stderr.write("rocm-smi at failure: {0}\n".format(smi_snapshot()))    # for root-causing; after the line is out
        sys.stderr.flush()
    else:
        time.sleep(4.0)
    os._exit(EXIT_COLLECTIVE)


def first_collective(gpu_cat, group, rank, world, args):
    """Watchdog for the first collective (RCCL builds its xGMI rings lazily inside it): the enqueue runs in a helper
    thread; if any rank's all-reduce has not completed after COLLECTIVE_TIMEOUT_S the run ends with the marked line and a
    non-zero exit status on every rank."""
    done = threading.Event()
    error = []

    def step():
        try:
            gpu_cat.enqueue()
            gpu_cat.sync()
            done.set()
        except Exception as exc:
            error.append(repr(exc))
    worker = threading.Thread(target=step, daemon=True)
    worker.start()
    worker.join(COLLECTIVE_TIMEOUT_S)
    ok = 1 if done.is_set() else 0
    try:
        all_ok = int(group.allreduce(np.array([ok], dtype=np.int64), op="min")[0])
    except Exception as exc:
        fail_collective(rank, world, args, "first all-reduce", "host group failed while agreeing on the outcome: {0!r}".format(exc))
    if not all_ok:
        why = error[0] if error else ("no completion within {0} s".format(COLLECTIVE_TIMEOUT_S) if not ok else "another rank failed")
        fail_collective(rank, world, args, "first all-reduce", why)


def timed_steps(gpu_cat, group, steps, warmup, ramp_seconds, stride):
    """W untimed warm-up steps (after an untimed clock ramp), then exactly K pipelined steps between two barriers +
    device syncs; returns the max-over-ranks wall time and this rank's sampled kernel time."""
    def barrier():
        gpu_cat.sync()
        if group is not None:
            group.barrier()

    gpu_cat.set_option("timing", 2)
    gpu_cat.set_option("timing_reserve", min(65536, max(steps, warmup) + 32))   # no hipEventCreate inside the timed region
    # first launches after idle (no ramp): what a short job sees before the clocks have settled
    gpu_cat.set_option("timing_stride", 1)
    gpu_cat.enqueue()
    gpu_cat.sync()
    gpu_cat.timing_collect()
    for _ in range(20):
        gpu_cat.enqueue()
    cold_ms, cold_n = gpu_cat.timing_collect()
    gpu_cat.set_option("timing_stride", stride)
    # Clock ramp (untimed, before the W warm-up steps): the GPU needs some 100 ms of load to reach its sustained clocks.
    # Every rank runs the same number of launches (the all-reduce is collective).  No HIP events in the ramp and the
    # warm-up: hundreds of recorded-but-unread event pairs make the FIRST launch after the next synchronisation cost
    # ~250 us of host time (the runtime retires them then) -- inside a K = 20 timed region that is 12 us per step.
    gpu_cat.set_option("timing", 0)
    ramp_steps = 0
    if ramp_seconds > 0:
        t_r = time.perf_counter()
        for _ in range(8):
            gpu_cat.enqueue()
        gpu_cat.sync()
        per_step = max((time.perf_counter() - t_r) / 8, 1e-6)
        ramp_steps = int(min(20000, ramp_seconds / per_step))
        if group is not None:
            ramp_steps = int(group.allreduce(np.array([ramp_steps], dtype=np.int64), op="max")[0])
        for _ in range(ramp_steps):
            gpu_cat.enqueue()
    # The W warm-up steps; the last one runs alone behind a synchronisation: the first launch after a sync retires the
    # runtime's backlog of completed commands (~100 us of host time after the ramp's thousands of launches), which
    # would otherwise delay the first step of the timed region.
    for _ in range(max(0, warmup - 1)):
        gpu_cat.enqueue()
    gpu_cat.sync()
    if warmup > 0:
        gpu_cat.enqueue()
    gpu_cat.set_option("timing", 2)                  # (synchronises; the event pairs were created by timing_reserve)
    barrier()
    t0 = time.perf_counter()
    if os.environ.get("MCD_BENCH_DEBUG"):
        marks = []
        for _ in range(steps):
            gpu_cat.enqueue()
            marks.append(time.perf_counter())
        sys.stderr.write("enqueue call times (us): " + " ".join("{0:.0f}".format((b - a) * 1e6) for a, b in zip([t0] + marks[:-1], marks)) + "\n")
    else:
        for _ in range(steps):
            gpu_cat.enqueue()
    t_enq = time.perf_counter() - t0
    gpu_cat.sync()
    elapsed = time.perf_counter() - t0
    if os.environ.get("MCD_BENCH_DEBUG"):
        sys.stderr.write("timed region: enqueue loop {0:.1f} us, total {1:.1f} us, {2} steps\n".format(t_enq * 1e6, elapsed * 1e6, steps))
    if group is not None:
        group.barrier()
        elapsed = float(group.allreduce(np.array([elapsed]), op="max")[0])
    kernel_ms_total, n_launch = gpu_cat.timing_collect()
    # The same kernel alone on the chip: with two lanes consecutive launches of the timed region overlap, so a
    # launch's own duration above includes time it shares with its neighbours.  A short pass on ONE lane (untimed for
    # `value`, every rank alike) gives the duration of a launch that has the SIMDs to itself -- what rocprofv3 reports for
    # a run with MCD_BENCH_OPTIONS=two_lanes=0 (profiles/: *_kernel_stats_one_lane.txt).
    one_lane_s, one_lane_n = None, 0
    try:
        gpu_cat.set_option("two_lanes", 0)
        gpu_cat.set_option("timing", 0)
        for _ in range(8):
            gpu_cat.enqueue()
        gpu_cat.set_option("timing", 2)
        for _ in range(48):
            gpu_cat.enqueue()
        gpu_cat.sync()
        ms1, n1 = gpu_cat.timing_collect()
        one_lane_s, one_lane_n = ms1 * 1e-3 / max(1, n1), int(n1)
    finally:
        gpu_cat.set_option("timing", 0)
        gpu_cat.set_option("two_lanes", 0 if "two_lanes=0" in (os.environ.get("MCD_BENCH_OPTIONS") or "") else 1)
    return {"elapsed": elapsed, "kernel_s": kernel_ms_total * 1e-3 / max(1, n_launch), "kernel_samples": int(n_launch),
            "ramp_steps": ramp_steps, "cold_kernel_us": cold_ms * 1e3 / max(1, cold_n),
            "kernel_one_lane_s": one_lane_s, "kernel_one_lane_samples": one_lane_n}


def blocking_calls(gpu_cat, pos, n_calls):
    """SURVEY 8(d) timing protocol: wall time of the blocking C-ABI call (params H2D, kernels, reduce, [all-reduce on the
    critical path], results D2H), median of >= 20 calls after 3 warm-ups."""
    gpu_cat.set_option("timing", 0)
    for _ in range(3):
        gpu_cat.loglike(pos)
    times = []
    for _ in range(n_calls):
        t = time.perf_counter()
        gpu_cat.loglike(pos)
        times.append(time.perf_counter() - t)
    return float(np.median(times)), float(np.mean(times))


def c4_strong(native, synthetic, ctx, group, rank, world, args):
    """North-star strong scaling: ONE 1e7-star catalogue (block-seeded, identical for every N) sharded over the ranks,
    rotation+dispersion, 256 walkers, one all-reduce of 256 doubles per step."""
    from mcmc_dynamics_amd import distributed
    n_total = int(args.c4_stars)
    lo, hi = distributed.shard_bounds(n_total, rank, world)
    cat = synthetic.make_catalog_range(n_total, lo, hi, config=4)
    pos = synthetic.make_walkers(256, NAMES4, cat["truth"], config=4)
    gpu_cat = build_catalog(native, ctx, synthetic, cat, "const")
    gpu_cat.upload_params(pos)
    stride = 8 if args.steps >= 64 else 4
    t = timed_steps(gpu_cat, group, args.steps, args.warmup, min(args.clock_ramp_seconds, 0.2), stride)
    result = gpu_cat.fetch()
    med, mean = blocking_calls(gpu_cat, pos, 20)
    if group is not None:
        med = float(group.allreduce(np.array([med]), op="max")[0])
        kernel_us = [float(x[0]) for x in group.allgather_array(np.array([t["kernel_s"] * 1e6]))]
        identical = bool(group.same_everywhere(result))
    else:
        kernel_us, identical = [t["kernel_s"] * 1e6], True
    terms = float(n_total) * 256
    out = {"workload": "C4: one {0:.0e}-star synthetic catalogue sharded over {1} rank(s), rotation+dispersion, 256 walkers, "
                       "fixed centre".format(n_total, world), "scaling": "strong", "stars_total": n_total,
           "stars_per_gpu": hi - lo, "walkers": 256, "likelihood": "const", "steps": args.steps, "warmup": args.warmup,
           "terms_per_s": terms * args.steps / t["elapsed"], "ms_per_step": t["elapsed"] / args.steps * 1e3,
           "kernel_us_per_rank": kernel_us, "kernel_us_sampled_launches": t["kernel_samples"],
           "allreduce": ("none (1 rank)" if world == 1 else
                         "pipelined steps (two compute lanes): ncclAllReduce(sum, f64, 256) on the communication stream behind an event, "
                         "overlapping the next step's kernels (off the critical path); blocking call: on the compute stream "
                         "(on the critical path)"),
           "blocking_call_us_median": med * 1e6, "blocking_call_terms_per_s": terms / med,
           "results_identical_on_all_ranks": identical,
           "lnlike_first_walkers": [float(x) for x in result[:2]]}
    gpu_cat.close()
    return out


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
    names = NAMES4 + (["v_back", "sigma_back", "f_back"] if model == "bggauss" else [])

    # Extra CPU figure (all host cores) -- forked workers, so it runs before anything initialises the HIP runtime.
    all_cores = None
    if (world == 1 and not args.no_cpu_baseline and not args.no_cpu_all_cores
            and args.workload in ("c2", "c3", "c3gb", "c3const")):
        try:
            cat0 = synthetic.make_catalog(n_stars, config=config, seed=synthetic.CATALOG_SEED_BASE + config,
                                          background=(model != "const"))
            pos0 = synthetic.make_walkers(n_walkers, names, cat0["truth"], config=config)
            probe, _ = cpu_baseline(cat0, pos0[:2], model, 1.0)
            all_cores = cpu_baseline_all_cores(cat0, pos0, model, n_stars / probe["value"], min(args.cpu_seconds, 10.0))
            del cat0
        except Exception as exc:                          # never let the extra baseline break the bench line
            all_cores = {"error": repr(exc)}

    from mcmc_dynamics_amd import _native as native
    n_visible = int(os.environ.get("MCD_VISIBLE_DEVICES", "0")) or None      # testing aid: fold ranks onto fewer devices
    if n_visible:
        local_rank = local_rank % n_visible
    group, comm = None, None
    if world > 1:
        from mcmc_dynamics_amd.hostgroup import HostGroup, HostGroupError
        try:
            group = HostGroup(rank, world, timeout=COLLECTIVE_TIMEOUT_S + 60)
        except HostGroupError as exc:
            fail_collective(rank, world, args, "host rendezvous", repr(exc))
        # rank 0's RCCL unique id -> every rank -> ncclCommInitRank; all ranks learn whether everybody succeeded
        uid, err = b"", ""
        if rank == 0:
            try:
                uid = native.Context.unique_id()
            except native.NativeError as exc:
                err = str(exc)
        try:
            uid = group.bcast_bytes(uid, src=0)
            ctx = None
            if len(uid) == native.UNIQUE_ID_BYTES:
                # ncclCommInitRank under the same watchdog as the first all-reduce: a bootstrap that never completes must
                # end in the marked line, not in a job that hangs until the driver's limit
                box = {}

                def init():
                    try:
                        box["ctx"] = native.Context(rank=rank, n_ranks=world, unique_id=uid, device=local_rank)
                    except native.NativeError as exc:
                        box["err"] = str(exc)
                th = threading.Thread(target=init, daemon=True)
                th.start()
                th.join(COLLECTIVE_TIMEOUT_S)
                ctx = box.get("ctx")
                err = box.get("err", "" if ctx is not None else "ncclCommInitRank did not return within {0} s".format(COLLECTIVE_TIMEOUT_S))
            ok = int(group.allreduce(np.array([1 if ctx is not None else 0], dtype=np.int64), op="min")[0])
            errs = group.bcast_json(err, src=0) if ok == 0 else ""
        except HostGroupError as exc:
            fail_collective(rank, world, args, "communicator set-up", repr(exc))
        if not ok:
            fail_collective(rank, world, args, "ncclCommInitRank", err or errs or "failed on another rank")
        comm = ctx.comm_info()
        sizes = group.allreduce(np.array([comm["size"], -comm["size"]], dtype=np.int64), op="max")
        if int(sizes[0]) != world or int(sizes[1]) != -world or comm["rank"] != rank:
            fail_collective(rank, world, args, "communicator check",
                            "ncclCommCount / ncclCommUserRank report {0}, expected size {1} rank {2}".format(comm, world, rank))
    else:
        ctx = native.Context(n_devices=1, device_ids=[local_rank])
        comm = ctx.comm_info()

    # ---- synthetic catalogue shard of this rank; identical walkers on every rank
    if strong:
        from mcmc_dynamics_amd import distributed
        lo, hi = distributed.shard_bounds(n_stars, rank, world)
        cat = synthetic.make_catalog_range(n_stars, lo, hi, config=config, background=(model != "const"))
        total_stars = n_stars
    else:
        cat = synthetic.make_catalog(n_stars, config=config, seed=synthetic.CATALOG_SEED_BASE + config + 1000 * rank,
                                     background=(model != "const"))
        total_stars = n_stars * world
    truth = cat["truth"]
    if group is not None and not strong:             # walkers are drawn around rank 0's truth everywhere
        truth = group.bcast_json(truth, src=0)
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

    gpu_cat = build_catalog(native, ctx, synthetic, cat, model, args.precision, bin_offsets)
    gpu_cat.upload_params(pos)
    if group is not None:
        first_collective(gpu_cat, group, rank, world, args)

    # HIP events around every 4th (K < 64) or 8th main-kernel launch of the timed region: an event pair costs two signal
    # packets (~6 us per step between back-to-back kernels, tools/event_cost_probe.py), so the kernel time is sampled
    stride = 8 if args.steps >= 64 else 4
    try:
        t = timed_steps(gpu_cat, group, args.steps, args.warmup, args.clock_ramp_seconds, stride)
    except Exception as exc:
        if group is None:
            raise
        fail_collective(rank, world, args, "timed region", repr(exc))
    elapsed, kernel_s = t["elapsed"], t["kernel_s"]
    result = gpu_cat.fetch()
    info = gpu_cat.launch_info()
    info["kernel_family"] = {0: "plain", 1: "fast", 2: "fast, narrow-range variant", -1: "none"}[gpu_cat.fast_level]
    info["prefetch"] = gpu_cat.last_prefetch             # 1: the instantiation that prefetches the next iteration's records

    # blocking C-ABI call (host params in, host results out): the PCIe/sync-inclusive rate, never `value`
    n_sync = max(20, min(50, args.steps))
    sync_med, sync_mean = blocking_calls(gpu_cat, pos, n_sync)
    if group is not None:
        sync_med = float(group.allreduce(np.array([sync_med]), op="max")[0])

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
        state = sampler.run_mcmc(pos, 300)                 # warm: a full block and a short one (work buffers, chain storage)
        n_mcmc = 1024                                      # four blocks of the built-in sampler (256 steps each)
        start = tuple(state)[0]

        def timed_chain():
            t2 = time.perf_counter()
            sampler.run_mcmc(start, n_mcmc)
            return time.perf_counter() - t2

        dt = timed_chain()
        mcmc = {"steps_per_s": n_mcmc / dt, "terms_per_s": float(len(cat["v"])) * n_walkers * n_mcmc / dt,
                "driver": type(sampler).__module__ + "." + type(sampler).__name__,
                "posterior": type(fit).__name__ + ".lnprob_batch", "calls_per_step": 2, "walkers_per_call": n_walkers // 2,
                "steps": n_mcmc, "acceptance_fraction": float(np.mean(sampler.acceptance_fraction))}
        fit_cat = getattr(fit, "_catalog", None)
        if getattr(sampler, "block_fn", None) is not None and fit_cat is not None:
            # where mcd_stretch_move ran its blocks (resident on the device / host-driven), and the host-driven rate of the
            # same chain beside it (option "device_chain" = 0: one blocking evaluation per half step)
            mcmc["stretch_blocks"] = fit_cat.stretch_info()
            fit_cat.set_option("device_chain", 0)
            dt_host = timed_chain()
            fit_cat.set_option("device_chain", 1)
            mcmc["host_driven_steps_per_s"] = n_mcmc / dt_host
            # the default draws the move's numbers on the device (Runner.RNG = "device", csrc/mcd_rng.h); beside it the same
            # blocks with NumPy's generator on the host (RNG = "host")
            mcmc["random_numbers"] = "rng=" + getattr(sampler, "rng", "emcee")
            from mcmc_dynamics_amd.sampler import EnsembleSampler
            host_sampler = EnsembleSampler(n_walkers, fit.n_fitted_parameters, fit.lnprob_batch, vectorize=True, seed=3,
                                           rng="host", block_fn=fit._stretch_block)
            host_state = host_sampler.run_mcmc(start, 300)
            t2 = time.perf_counter()
            host_sampler.run_mcmc(host_state[0], n_mcmc, log_prob0=host_state[1])
            mcmc["host_numbers_steps_per_s"] = n_mcmc / (time.perf_counter() - t2)
        fit.close()

    if world == 1 and n_bins > 1 and not args.no_mcmc and model == "const" and args.precision == "f64":
        # C5: one ensemble per radial bin in lockstep (the reference runs one MCMC per bin, bin/run_tests.py:75-124);
        # blocks of steps inside the library (mcd_stretch_move with n_bins = B) and, beside it, the NumPy loop
        import logging
        from mcmc_dynamics_amd import DataReader
        from mcmc_dynamics_amd.analysis import BinnedConstantFit
        from mcmc_dynamics_amd.analysis.binned import BinnedSampler
        logging.getLogger("mcmc_dynamics_amd").setLevel(logging.ERROR)
        reader = DataReader({k: cat[k] for k in ("ra", "dec", "v", "verr")})
        reader.make_radial_bins(synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG, nstars=1000, dlogr=0.05)
        fit = BinnedConstantFit(reader, context=ctx)
        fit.parameters["ra_center"].set(value=synthetic.CENTER_RA_DEG, fixed=True)
        fit.parameters["dec_center"].set(value=synthetic.CENTER_DEC_DEG, fixed=True)
        rates = {}
        # "library": the default (BinnedConstantFit.RNG = "device": numbers generated on the device, csrc/mcd_rng.h);
        # "host_numbers": the same blocks with NumPy's generator on the host (rng="host"); "numpy_loop": no library blocks
        for label, block_fn, n_mcmc, rng_mode in (("library", fit._stretch_block, 256, "device"),
                                                  ("host_numbers", fit._stretch_block, 256, "host"), ("numpy_loop", None, 8, "host")):
            sampler = BinnedSampler(fit.n_bins, n_walkers, fit.n_fitted_parameters, fit.lnprob_batch, seed=5, block_fn=block_fn,
                                    rng=rng_mode, seeded_block_fn=fit._stretch_block_seeded if block_fn is not None else None)
            sampler.reserve(520)                             # (one allocation for the warm-up and the timed run)
            state = sampler.run_mcmc(pos, 4 if block_fn is None else 256)     # (untimed: also sizes the block arena)
            t2 = time.perf_counter()
            sampler.run_mcmc(state[0], n_mcmc, log_prob0=state[1])
            rates[label] = n_mcmc / (time.perf_counter() - t2)
            if label == "library":
                acc = float(np.mean(sampler.acceptance_fraction))
            sampler.close()
        mcmc = {"steps_per_s": rates["library"], "terms_per_s": float(len(cat["v"])) * n_walkers * rates["library"],
                "driver": "mcmc_dynamics_amd.analysis.binned.BinnedSampler", "posterior": "BinnedConstantFit.lnprob_batch",
                "ensembles": fit.n_bins, "calls_per_step": 2, "walkers_per_call": n_walkers // 2, "steps": 256,
                "acceptance_fraction": acc, "random_numbers": "generated on the device (Philox4x64-10, csrc/mcd_rng.h: chain_numbers_kernel)",
                "host_numbers_steps_per_s": rates["host_numbers"], "numpy_loop_steps_per_s": rates["numpy_loop"],
                "stretch_blocks": fit._catalog.stretch_info()}
        fit.close()

    # per-rank figures of the headline workload, gathered before the catalogue goes away
    kernel_us_ranks = [kernel_s * 1e6]
    if group is not None:
        kernel_us_ranks = [float(x[0]) for x in group.allgather_array(np.array([kernel_s * 1e6]))]
    gpu_cat.close()

    strong_rec = None
    if args.workload == "c3" and not args.no_c4_strong and args.stars is None and args.walkers is None and args.precision == "f64":
        try:
            strong_rec = c4_strong(native, synthetic, ctx, group, rank, world, args)
        except Exception as exc:
            if group is not None:
                fail_collective(rank, world, args, "c4_strong", repr(exc))
            strong_rec = {"error": repr(exc)}

    if rank != 0:
        if group is not None:
            group.barrier()
            group.close()
        return

    terms_per_step = float(total_stars) * n_walkers
    value = terms_per_step * args.steps / elapsed
    local_terms = float(len(cat["v"])) * n_walkers
    streaming_gbps = local_terms * bytes_per_term / kernel_s / 1e9

    traffic, traffic_src = None, None                # PMC bytes per launch: imported from the committed rocprofv3 passes
    if args.stars is None and args.walkers is None and args.precision == "f64" and not strong:
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            traffic = pmc.get(args.workload)
            traffic_src = ("imported, not collected in this run: profiles/pmc_traffic.json = FETCH_SIZE + WRITE_SIZE per launch "
                           "of this kernel from two rocprofv3 --pmc passes of this command ({0})".format(pmc.get("_provenance", "see profiles/README.md")))
        except Exception:
            pass

    roofline = None
    if args.precision == "f64":
        isa_key = model if model in ("bgfixed", "bggauss", "profile") and info["kernel_family"].endswith("narrow-range variant") else \
            (model + "_general" if model in ("bgfixed", "bggauss", "profile") else model)
        row, src = isa_counts(isa_key)
        if row is not None:
            # the instantiation the timed launches ran: with or without the software prefetch of the records
            prefetching = info.get("prefetch") == 1
            if prefetching and "valu_per_term_prefetch" in row:
                row = dict(row, valu_per_term=row["valu_per_term_prefetch"], f64_per_term=row["f64_per_term_prefetch"])
            achieved = row["valu_per_term"] * local_terms / 64.0 / kernel_s
            roofline = {
                "bound": "valu_f64", "achieved": achieved, "peak": VALU_F64_PEAK, "unit": "VALU wave-instructions/s",
                "frac": achieved / VALU_F64_PEAK, "traffic": traffic, "traffic_source": traffic_src,
                "traffic_note": None if traffic is None else
                "FETCH_SIZE is tallied in 64-byte units whether the request was 64 or 128 bytes (MI355X_MICROARCH.md); "
                "tools/summarize_rocprof.py takes the factor (x1 / x2) that reproduces the one byte count that is known -- the "
                "compulsory read of the record array -- so the figure is CALIBRATED on the compulsory traffic and cannot "
                "by itself show over-fetch beyond a factor 2; WRITE_SIZE needs no correction",
                "kernel": "mcd::loglike_kernel", "kernel_us": kernel_s * 1e6, "kernel_us_sampled_launches": t["kernel_samples"],
                "kernel_us_first_20_launches_after_idle": t["cold_kernel_us"],
                "valu_per_term": row["valu_per_term"], "valu_f64_per_term": row["f64_per_term"],
                "prefetching_instantiation": prefetching, "valu_per_term_source": src, "terms_per_launch": local_terms,
                "peak_definition": "256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 f64 vector instruction",
                "hbm_streaming_model": {"algorithmic_bytes_per_term": bytes_per_term, "GBps": streaming_gbps,
                                        "frac_of_8TBps": streaming_gbps / HBM_PEAK_GBPS,
                                        "note": "SURVEY 8(d): each walker's sum reads every star record once; one scalar record "
                                                "load serves 64 walkers, so a figure > 1 means register reuse, not bandwidth"},
                "hbm_measured": None if traffic is None else {
                    "bytes_per_launch": traffic, "GBps": traffic / kernel_s / 1e9,
                    "frac_of_8TBps": traffic / kernel_s / 1e9 / HBM_PEAK_GBPS},
                "compulsory_bytes_per_launch": len(cat["v"]) * info["record_bytes"], "walker_tile": info["walker_tile"],
                "note": "the kernel is bound by f64 vector-instruction issue (VALUBusy 97 %, profiles/): achieved = VALU "
                        "wave-instructions of the hot loop per term x terms / 64 lanes / kernel time"}
        else:
            roofline = {"bound": "valu_f64", "achieved": None, "peak": VALU_F64_PEAK, "unit": "VALU wave-instructions/s",
                        "frac": None, "traffic": traffic, "kernel_us": kernel_s * 1e6, "error": src}
    else:
        # float32 modes (the C5 sweep): the roof is f32 vector-instruction issue, 2 cycles per wave64 instruction
        row, src = isa_counts(model + "_" + args.precision) if model in ("const", "bgfixed") else (None, "no instruction count for this model in float32")
        roofline = {"bound": "valu_f32", "achieved": None, "peak": VALU_F32_PEAK, "unit": "VALU wave-instructions/s", "frac": None,
                    "traffic": None, "kernel": "mcd::loglike_kernel", "kernel_us": kernel_s * 1e6,
                    "kernel_us_sampled_launches": t["kernel_samples"], "terms_per_launch": local_terms,
                    "peak_definition": "256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 f32 vector instruction "
                                       "(tools/valu_rate_probe.hip: 1.04 ns per v_fma_f32 against 2.35 ns per v_fma_f64, "
                                       "profiles/r03_valu_rate_probe.txt)",
                    "hbm_streaming_model": {"algorithmic_bytes_per_term": bytes_per_term, "GBps": streaming_gbps,
                                            "frac_of_8TBps": streaming_gbps / HBM_PEAK_GBPS}}
        if row is not None:
            prefetching = info.get("prefetch") == 1
            per_term = row["valu_per_term_prefetch"] if prefetching and "valu_per_term_prefetch" in row else row["valu_per_term"]
            achieved = per_term * local_terms / 64.0 / kernel_s
            roofline.update({"achieved": achieved, "frac": achieved / VALU_F32_PEAK, "valu_per_term": per_term,
                             "valu_f64_per_term": row["f64_per_term"], "prefetching_instantiation": prefetching,
                             "valu_per_term_source": src,
                             "note": "achieved = VALU wave-instructions of the hot loop per term x terms / 64 lanes / kernel time, "
                                     "every instruction counted as one f32 slot (an f64 addition of the f32acc64 sums costs two, "
                                     "v_rcp_f32 / v_frexp about three: frac is a lower bound of the pipe's occupation)"})
        else:
            roofline["error"] = src

    # Pipelined evaluations alternate between two compute streams ("two lanes"; option two_lanes, default on):
    # consecutive launches OVERLAP -- the tail and the reduction of one run beside the start of the next -- so a launch's own
    # duration (HIP events / rocprofv3) is longer than the steady-state period per launch.  `roofline.frac` stays the
    # contract's definition (per-launch duration); the period-based figure is given beside it.
    lanes = 1 if "two_lanes=0" in (os.environ.get("MCD_BENCH_OPTIONS") or "") else 2
    if roofline is not None and roofline.get("achieved") is not None:
        period_s = elapsed / args.steps
        roofline["launch_period_us"] = period_s * 1e6
        roofline["frac_by_launch_period"] = roofline["achieved"] * kernel_s / period_s / roofline["peak"]
        if t.get("kernel_one_lane_s"):
            roofline["kernel_us_one_lane"] = t["kernel_one_lane_s"] * 1e6
            roofline["kernel_us_one_lane_sampled_launches"] = t["kernel_one_lane_samples"]
            roofline["frac_one_lane"] = roofline["achieved"] * kernel_s / t["kernel_one_lane_s"] / roofline["peak"]
        roofline["lanes"] = lanes
        if lanes == 2:
            roofline["lanes_note"] = ("two lanes: consecutive launches of the timed region overlap, so kernel_us (one launch, start "
                                      "to end, shared with its neighbours) exceeds launch_period_us (wall time per launch); frac "
                                      "uses kernel_us as the contract defines it and is a lower bound; frac_one_lane: the same "
                                      "launch alone on the chip (a short extra pass on one lane); frac_by_launch_period: "
                                      "instructions per launch over the steady-state period")
    out = {
        "metric": "star-walker log-L terms/sec", "value": value, "unit": "terms/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "clock_ramp_steps": t["ramp_steps"],
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "degraded": False,
        "config": {"workload": desc, "stars_per_gpu": len(cat["v"]), "stars_total": total_stars, "walkers": n_walkers,
                   "likelihood": model,
                   "parallelism": ("stars sharded over {0} ranks (one process per GPU); one ncclAllReduce(sum, f64, {1}) per step"
                                   .format(world, n_walkers * n_bins) if world > 1 else "1 GPU")},
        "roofline": roofline,
        "pipelining": {"lanes": lanes, "api": "mcd_params_upload once, K x mcd_loglike_enqueue, one mcd_sync",
                       "note": "independent evaluations in flight on two HIP streams with their own partial-sum and result "
                               "buffers; a sampler's chain (mcmc_end_to_end) and the blocking call (value_blocking) cannot overlap "
                               "evaluations and run on one"},
        "kernel_us_per_rank": kernel_us_ranks,
        # SURVEY 8(d)'s own protocol beside `value`: the BLOCKING C-ABI call (parameters in, walker prep, kernels, reduction
        # [, all-reduce], results out, synchronisation), median of 50 calls after 3 warm-ups
        "value_blocking": local_terms * world / sync_med,
        "blocking_call_us_median": sync_med * 1e6, "blocking_call_us_mean": sync_mean * 1e6,
        "blocking_call_terms_per_s": local_terms * world / sync_med,
        "launch": info,
        "mcmc_end_to_end": mcmc,
        "collective": None if world == 1 else {
            "kind": "ncclAllReduce(sum, f64, count = {0}) per step; pipelined steps run it on the communication stream, beside the two compute lanes".format(n_walkers * n_bins),
            "comm_size": comm["size"], "comm_rank_of_rank0": comm["rank"], "rccl_version": comm["rccl_version"],
            "first_allreduce_watchdog_s": COLLECTIVE_TIMEOUT_S,
            # MCD_RCCL_LIBRARY substitutes the collective library (tests/fake_rccl on a single-GPU box): flow check only
            "library_override": os.environ.get("MCD_RCCL_LIBRARY")},
        "libraries": mapped_libraries(),
        "option_overrides": os.environ.get("MCD_BENCH_OPTIONS") or None,
        "c4_strong": strong_rec,
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
    if group is not None:
        group.barrier()
        group.close()


if __name__ == "__main__":
    main()
