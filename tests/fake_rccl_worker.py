"""Worker of tests/test_gpu_multirank_one_gpu.py (GPU): the library's multi-device and multi-rank code paths on ONE
device, with tests/fake_rccl standing in for librccl.so (RCCL refuses two ranks on one GPU).  Real shards, real kernels,
real streams; only the collective itself is host-staged.

    fake_rccl_worker.py single N         one process, N "devices" (all device 0): mcd_ctx_create(n_dev = N)
    fake_rccl_worker.py rank             one process per rank (RANK / WORLD_SIZE from the launcher): mcd_ctx_create_rank
    fake_rccl_worker.py deadline KIND    two ranks, one of which never completes a collective (KIND: hang -- the stand-in's
                                         all-reduce sits; raise -- rank 1 raises inside a block of Runner.__call__; die --
                                         rank 1's process dies): rank 0 must get the error within the collective deadline
                                         (hang) or at once (raise / die: the host group's abort channel), never wait for ever
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MCD_RCCL_LIBRARY"] = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
os.environ["MCD_ALLOW_SHARED_DEVICE"] = "1"

from mcmc_dynamics_amd import _native as native, distributed, synthetic     # noqa: E402
from mcmc_dynamics_amd.background import Gaussian                           # noqa: E402
from oracle import lnprob_numpy as oracle                                   # noqa: E402

CENTRE = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
NAMES4 = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]


def catalog(n, config, background):
    c = synthetic.make_catalog(n, config=config, background=background)
    dx, dy = oracle.calc_xy_offset(c["ra"], c["dec"], *CENTRE)
    r = np.hypot(dx, dy)
    near = r < 1e-2
    if near.any():
        donor = int(np.argmax(r))
        c["ra"][near], c["dec"][near] = c["ra"][donor], c["dec"][donor]
    return c


def cases():
    """(name, columns, model, extra kwargs, params, bin offsets or None)"""
    out = []
    c = catalog(30011, 4, False)                                            # odd size: uneven shards
    pos = synthetic.make_walkers(96, NAMES4, c["truth"], config=4)
    out.append(("const", c, native.MODEL_CONST, {}, pos, None))
    r = np.hypot(*oracle.calc_xy_offset(c["ra"], c["dec"], *CENTRE))
    order = np.argsort(r, kind="stable")
    srt = {k: (v[order] if isinstance(v, np.ndarray) else v) for k, v in c.items()}
    offs = np.array([0, 700, 5000, 5001, 14999, 15011, 22000, 30011], dtype=np.int64)       # bins straddling every kind of shard edge
    rng = np.random.default_rng(4)
    per_bin = np.stack([pos[:80] * (1.0 + 0.02 * rng.normal(size=(80, 4))) for _ in range(len(offs) - 1)])
    per_bin[..., 1] = np.abs(per_bin[..., 1])
    out.append(("const binned", srt, native.MODEL_CONST, {}, per_bin, offs))
    cb = catalog(40000, 3, True)
    lnbg = Gaussian(20.0, 40.0)(cb["v"], cb["verr"])
    pm = cb["pmember"].copy()
    pm[[11, 19999, 20000, 39990]] = 1.0                                     # certain members: narrow exceptions on both sides of the 2-shard edge
    posb = synthetic.make_walkers(160, NAMES4, cb["truth"], config=3)
    out.append(("bgfixed with exceptions", cb, native.MODEL_CONST_BGFIXED, dict(lnlike_bg=lnbg, pmember=pm), posb, None))
    # denormal regime of the reference's log-sum-exp met on ONE shard only (a certain member 60 sigma out, in the last
    # quarter of the stars): the re-run signal must reach every shard / rank through the all-reduce
    v2 = cb["v"].copy()
    v2[39990] = 900.0
    out.append(("bgfixed, re-run signal from one shard", dict(cb, v=v2), native.MODEL_CONST_BGFIXED,
                dict(lnlike_bg=Gaussian(20.0, 40.0)(v2, cb["verr"]), pmember=pm), posb, None))
    gb = synthetic.make_walkers(100, NAMES4 + ["v_back", "sigma_back", "f_back"], cb["truth"], config=3)
    gb[:, 5], gb[:, 6] = np.abs(gb[:, 5]), np.clip(gb[:, 6], 0.01, 0.99)
    out.append(("bggauss free centre", cb, native.MODEL_CONST_BGGAUSS, dict(density=cb["density"]),
                np.column_stack([gb[:, :4], np.full(100, CENTRE[0]), np.full(100, CENTRE[1]), gb[:, 4:]]), None))
    return out


def make(ctx, cols, model, kw, offs, lo=None, hi=None, free=False):
    sl = slice(lo, hi)
    extra = {k: v[sl] for k, v in kw.items()}
    return native.Catalog(ctx, cols["ra"][sl], cols["dec"][sl], cols["v"][sl], cols["verr"][sl], model=model,
                          centre=None if free else CENTRE, bin_offsets=offs, **extra)


def check(name, got, want, reruns=None):
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin), name
    err = float(np.max(np.abs(got[fin] - want[fin]) / np.abs(want[fin]))) if fin.any() else 0.0
    assert err < 1e-13, (name, err)
    return err


def single(n_dev):
    one = native.Context(n_devices=1)
    many = native.Context(device_ids=[0] * n_dev)
    assert many.n_devices == n_dev and many.comm_info() == {"size": n_dev, "rank": 0, "rccl_version": 29999}
    worst = 0.0
    for name, cols, model, kw, params, offs in cases():
        free = "free" in name
        ref = make(one, cols, model, kw, offs, free=free)
        want = ref.loglike(params)
        cat = make(many, cols, model, kw, offs, free=free)
        got = cat.loglike(params)
        worst = max(worst, check(name, got, want))
        assert np.array_equal(cat.loglike(params), got)                       # bitwise repeatable
        # pipelined use on several shards: double-buffered results, the all-reduce of step i behind the kernels of i + 1
        for depth in (1, 2, 5):
            cat.upload_params(params)
            for _ in range(depth):
                cat.enqueue()
            assert np.array_equal(cat.fetch(), got), (name, depth)
        if "re-run" in name:
            assert cat.rerun_count >= 1 and ref.rerun_count >= 1
            assert np.all(np.isneginf(want)) or np.any(np.isfinite(want))
        if model in (native.MODEL_CONST_BGFIXED, native.MODEL_CONST_BGGAUSS) and offs is None and not free:
            row = params[3]
            # per-star outputs are stitched from the shards (a certain member 90 sigma out is 0 / 0 in the reference's
            # formula, constant.py:366-374: NaN on every path)
            assert np.allclose(cat.membership(row), ref.membership(row), rtol=0, atol=1e-12, equal_nan=True)
        cat.close()
        ref.close()
    print("FAKE_RCCL_SINGLE_OK n_dev={0} worst={1:.2e}".format(n_dev, worst))


def rank_mode():
    from mcmc_dynamics_amd import DataReader
    from mcmc_dynamics_amd.analysis import ConstantFit
    ctx = distributed.rank_context(device=0)                                  # every rank on device 0
    rank, world = ctx.rank, ctx.n_ranks
    group = ctx.host_group
    assert ctx.comm_info() == {"size": world, "rank": rank, "rccl_version": 29999}
    one = native.Context(n_devices=1)
    worst = 0.0
    for name, cols, model, kw, params, offs in cases():
        free = "free" in name
        n = len(cols["v"])
        lo, hi = distributed.shard_bounds(n, rank, world)
        my_offs = None if offs is None else distributed.shard_bin_offsets(offs, rank, world)
        cat = make(ctx, cols, model, kw, my_offs, lo, hi, free=free)
        got = cat.loglike(params)
        assert group.same_everywhere(got), name                               # every rank holds the all-reduced result
        ref = make(one, cols, model, kw, offs, free=free)
        want = ref.loglike(params)
        worst = max(worst, check(name, got, want))
        for depth in (1, 3):
            cat.upload_params(params)
            for _ in range(depth):
                cat.enqueue()
            assert np.array_equal(cat.fetch(), got), (name, depth)
        if "re-run" in name:
            # the outlier sits on the LAST rank's shard only; every rank must have re-evaluated with the plain kernels
            counts = group.allgather_array(np.array([float(cat.rerun_count)]))
            assert all(c[0] >= 1 for c in counts), counts
        cat.close()
        ref.close()
    # Runner.__call__ on the rank context, real kernels on every rank's shard: identical chains everywhere, and the chain a
    # single-GPU run produces from the same start and seed (to rounding: the shard sums associate differently)
    cb = catalog(20000, 3, True)
    cols = {k: cb[k] for k in ("ra", "dec", "v", "verr", "pmember")}
    mine = distributed.shard_columns(cols, rank, world)
    fit = ConstantFit(DataReader(mine), background=Gaussian(20.0, 40.0), context=ctx)
    fit.parameters["ra_center"].set(value=CENTRE[0], fixed=True)
    fit.parameters["dec_center"].set(value=CENTRE[1], fixed=True)
    pos = synthetic.make_walkers(32, NAMES4, cb["truth"], config=3)
    sampler = fit(n_walkers=32, n_steps=12, pos=pos if rank == 0 else pos + 1.0, prefix=None)    # only rank 0's start counts
    chain = np.asarray(sampler.chain)
    assert group.same_everywhere(chain) and np.all(np.isfinite(sampler.lnprobability))
    # the blocks ran resident on every rank's device (csrc/mcd_stretch.hip: sums and the block's status word through the
    # all-reduce), and the host-driven loop gives the same chain bit for bit on this rank's shard sums
    info = fit._catalog.stretch_info()
    assert info["device_blocks"] >= 1 and info["discarded_blocks"] == 0, info
    rng = np.random.default_rng(77)                                   # (the same numbers on every rank)
    order = np.argsort(rng.random((9, 32)), axis=1).astype(np.int32)
    u = rng.random((9, 4, 16))
    zz = np.ascontiguousarray((u[:, :2] + 1.0) ** 2 / 2.0)
    thr = np.ascontiguousarray(np.log(u[:, 2:]) - 3.0 * np.log(zz))
    pick = rng.integers(0, 16, size=(9, 2, 16)).astype(np.int32)
    start, lnp0 = np.ascontiguousarray(chain[:, -1, :]), np.ascontiguousarray(np.asarray(sampler.lnprobability)[:, -1])
    blocks = []
    for mode in (1, 0):
        fit._catalog.set_option("device_chain", mode)
        p, l, c = start.copy(), lnp0.copy(), np.empty((9, 32, 4))
        fit._catalog.stretch_move(fit._stretch_plan(), p, l, order, zz, thr, pick, c, None, None)
        blocks.append((p, l, c))
    fit._catalog.set_option("device_chain", 1)
    assert all(np.array_equal(a, b) for a, b in zip(*blocks)) and group.same_everywhere(blocks[0][2])
    after = fit._catalog.stretch_info()
    assert after["device_blocks"] == info["device_blocks"] + 1 and after["host_blocks"] == info["host_blocks"] + 1, after
    full = ConstantFit(DataReader(cols), background=Gaussian(20.0, 40.0), context=one)
    full.parameters["ra_center"].set(value=CENTRE[0], fixed=True)
    full.parameters["dec_center"].set(value=CENTRE[1], fixed=True)
    lp = full.lnprob_batch(chain[:, -1, :])
    assert np.max(np.abs(lp - np.asarray(sampler.lnprobability)[:, -1]) / np.abs(lp)) < 1e-13
    fit.close()
    full.close()
    group.barrier()
    if rank == 0:
        print("FAKE_RCCL_RANKS_OK world={0} worst={1:.2e}".format(world, worst))
    ctx.close()
    group.close()


def deadline(kind):
    """VERDICT r2 item 2.  Every rank ends with exit status 3 (error seen, reported, left) or 9 (the rank that dies); a
    status 0 or a hang is the failure.  Prints DEADLINE_OK <rank> <kind> <seconds from the fault to the error>."""
    import time
    from mcmc_dynamics_amd import DataReader
    from mcmc_dynamics_amd.analysis import ConstantFit
    ctx = distributed.rank_context(device=0)
    rank, world, group = ctx.rank, ctx.n_ranks, ctx.host_group
    timeout_ms = 2500
    ctx.set_option("collective_timeout_ms", timeout_ms)
    cb = catalog(20000, 3, True)
    cols = {k: cb[k] for k in ("ra", "dec", "v", "verr", "pmember")}
    fit = ConstantFit(DataReader(distributed.shard_columns(cols, rank, world)), background=Gaussian(20.0, 40.0), context=ctx)
    fit.parameters["ra_center"].set(value=CENTRE[0], fixed=True)
    fit.parameters["dec_center"].set(value=CENTRE[1], fixed=True)
    pos = synthetic.make_walkers(32, NAMES4, cb["truth"], config=3)
    lp = fit.lnprob_batch(pos)                                                # a healthy collective first
    assert group.same_everywhere(lp) and not ctx.failed
    group.barrier()
    t0 = time.monotonic()
    err = None
    try:
        if kind == "hang":
            # from here on rank 1's all-reduces sit (FAKE_RCCL_HANG_AT_CALL was set for rank 1 before the library loaded):
            # blocking calls on both ranks run into the deadline -- rank 0 waits for a peer that never arrives
            for _ in range(6):
                fit.lnprob_batch(pos)
        else:
            calls = {"n": 0}

            def flaky_version_of(inner):
                def flaky(*args):
                    calls["n"] += 1
                    if rank == 1 and calls["n"] == 2:
                        if kind == "die":
                            sys.stdout.flush()
                            os._exit(9)
                        raise RuntimeError("injected failure inside block 2 on rank 1")
                    return inner(*args)
                return flaky
            fit._stretch_block = flaky_version_of(fit._stretch_block)                  # (Runner.RNG = "host")
            fit._stretch_block_seeded = flaky_version_of(fit._stretch_block_seeded)    # (the default: numbers from the device)
            fit.SAMPLER = "builtin"
            import mcmc_dynamics_amd.sampler as sampler_mod
            orig_init = sampler_mod.EnsembleSampler.__init__

            def short_blocks(self, *a, **k):
                orig_init(self, *a, **k)
                self.block_steps = 8
            sampler_mod.EnsembleSampler.__init__ = short_blocks
            fit(n_walkers=32, n_steps=64, pos=pos, prefix=None)
    except BaseException as exc:                                               # NativeError, RuntimeError, HostGroupError
        err = exc
    elapsed = time.monotonic() - t0
    assert err is not None, "rank {0}: no error although a peer never completed its collective".format(rank)
    text = "{0}: {1}".format(type(err).__name__, err)
    if kind == "hang":
        assert isinstance(err, native.NativeError) and "collective_timeout_ms" in text, text
        assert timeout_ms / 1000.0 * 0.8 < elapsed < timeout_ms / 1000.0 + 6.0, elapsed
    elif rank == 0:
        assert isinstance(err, native.NativeError) and "aborted by the host" in text, text
        assert ("injected failure" in text) if kind == "raise" else ("closed its control connection" in text), text
        assert elapsed < 8.0, elapsed                                          # at once, not at the deadline of a healthy job
    else:
        assert "injected failure" in text, text
    if rank == 0 or kind == "hang":
        assert ctx.failed
        try:
            fit.lnprob_batch(pos)
            raise SystemExit("a failed context accepted another call")
        except native.NativeError as again:
            assert "failed earlier" in str(again), again
    print("DEADLINE_OK {0} {1} {2:.2f} :: {3}".format(rank, kind, elapsed, text[:160]), flush=True)
    os._exit(3)                       # report and leave: destructors would wait for streams that never finish


if __name__ == "__main__":
    if sys.argv[1] == "single":
        single(int(sys.argv[2]))
    elif sys.argv[1] == "deadline":
        if sys.argv[2] == "hang" and os.environ.get("RANK") == "1":
            os.environ["FAKE_RCCL_HANG_AT_CALL"] = "2"           # call 1: the healthy evaluation
            os.environ["FAKE_RCCL_HANG_MS"] = "60000"
        deadline(sys.argv[2])
    else:
        rank_mode()
