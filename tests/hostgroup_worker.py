"""Worker of tests/test_distributed_cpu.py: one process per rank under the driver's launcher
(``python -m torch.distributed.run``), NO torch in the worker -- the host side of a multi-GPU run as bench.py and
``distributed.rank_context`` do it: HostGroup rendezvous, unique-id hand-off, sharded sums, max-over-ranks timing, and a
``Runner.__call__`` on a rank context (stub device: the oracle evaluates this rank's star shard and the host group plays
the all-reduce) whose chains must come out identical on every rank."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from mcmc_dynamics_amd import distributed, synthetic                      # noqa: E402
from mcmc_dynamics_amd.hostgroup import HostGroup                         # noqa: E402
from oracle import lnprob_numpy as oracle                                 # noqa: E402


class StubContext(object):
    """What ``distributed.rank_context`` returns, minus the GPU."""
    def __init__(self, rank, world, group):
        self.rank, self.n_ranks, self.host_group = rank, world, group


class ShardCatalog(object):
    """Stands in for the device catalogue of one rank: partial sums of the rank's stars, summed over the ranks."""
    def __init__(self, shard, centre, group):
        self.shard, self.centre, self.group, self.calls = shard, centre, group, 0

    def loglike(self, table):
        self.calls += 1
        return self.group.allreduce(oracle.batched_constant_lnlike(self.shard, np.asarray(table), *self.centre), op="sum")

    def stretch_move(self, plan, pos, lnp, order, zz, thr, pick, chain=None, lnprob_chain=None, accepted=None):
        """`_native.Catalog.stretch_move` with the library's own half-step loop (csrc/mcd_stretch.h, host build of
        tests/emul) around this stub's sharded likelihood."""
        import ctypes
        import emul_helper
        lib = emul_helper.lib()
        k = len(plan["col_source"])
        fn_t = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int64, ctypes.POINTER(ctypes.c_double))

        @fn_t
        def cb(tab, n, out):
            np.ctypeslib.as_array(out, shape=(n,))[:] = self.loglike(np.ctypeslib.as_array(tab, shape=(n, k)).copy())
            return 0
        cols = [np.ascontiguousarray(plan["col_source"], dtype=np.int32)] + \
            [np.ascontiguousarray(plan[key], dtype=np.float64) for key in ("col_const", "col_factor", "lo", "hi")]
        ptr = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None
        rc = lib.emul_stretch_block(ctypes.c_int64(1), ctypes.c_int64(pos.shape[0]), pos.shape[1], k, ptr(cols[0], ctypes.c_int32),
                                    *[ptr(c, ctypes.c_double) for c in cols[1:]], int(plan.get("fixed_ok", True)),
                                    ctypes.c_int64(order.shape[0]), ptr(pos, ctypes.c_double), ptr(lnp, ctypes.c_double),
                                    ptr(order, ctypes.c_int32), ptr(zz, ctypes.c_double), ptr(thr, ctypes.c_double),
                                    ptr(pick, ctypes.c_int32), ptr(chain, ctypes.c_double), ptr(lnprob_chain, ctypes.c_double),
                                    ptr(accepted, ctypes.c_int64), cb)
        assert rc == 0

    def stretch_move_seeded(self, plan, pos, lnp, seed, step0, n_steps, chain=None, lnprob_chain=None, accepted=None):
        """`_native.Catalog.stretch_move_seeded`: the numbers of the host build of csrc/mcd_rng.h, then the block above."""
        import emul_helper
        order, zz, thr, pick = emul_helper.chain_numbers(seed, step0, n_steps, 1, pos.shape[0], pos.shape[1])
        self.stretch_move(plan, pos, lnp, np.ascontiguousarray(order[:, 0]), np.ascontiguousarray(zz[:, :, 0]),
                          np.ascontiguousarray(thr[:, :, 0]), np.ascontiguousarray(pick[:, :, 0]), chain, lnprob_chain, accepted)

    def close(self):
        pass


def main():
    assert "torch" not in sys.modules
    rank, world, _ = distributed.env_rank()
    group = HostGroup.from_env(timeout=120)
    if rank == 0:                                   # build the host test library once, not concurrently in every rank
        import emul_helper
        emul_helper.lib()
    group.barrier()
    centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
    full = synthetic.make_catalog(4001, config=4)                    # odd size: uneven shards
    names = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]
    pos = synthetic.make_walkers(12, names, full["truth"], config=4)

    # the unique-id hand-off of distributed.rank_context / bench.py
    uid = group.bcast_bytes(bytes(range(128)) if rank == 0 else None, src=0)
    assert uid == bytes(range(128))
    assert group.bcast_json({"a": [1, 2.5, "x"]} if rank == 0 else None) == {"a": [1, 2.5, "x"]}
    arr = group.bcast_array(np.arange(6.0).reshape(2, 3) if rank == 0 else None)
    assert arr.shape == (2, 3) and arr[1, 2] == 5.0

    mine = distributed.shard_columns(full, rank, world)
    lo, hi = distributed.shard_bounds(4001, rank, world)
    sizes = group.allgather_array(np.array([float(hi - lo)]))
    assert sum(int(s[0]) for s in sizes) == 4001

    partial = oracle.batched_constant_lnlike(mine, pos, *centre)
    total = group.allreduce(partial, op="sum")                       # what ncclAllReduce does on the GPUs
    want = oracle.batched_constant_lnlike(full, pos, *centre)
    err = np.max(np.abs(total - want) / np.abs(want))
    assert err < 1e-13, err
    assert group.same_everywhere(total) and not group.same_everywhere(np.array([float(rank)]))

    # the other axis: full catalogue on every rank, walkers split across ranks, slices gathered over the host group
    many = synthetic.make_walkers(13, names, full["truth"], config=4)
    got = distributed.replicated_loglike(lambda p: oracle.batched_constant_lnlike(full, p, *centre), many, rank, world, group=group)
    assert got.shape == (13,) and np.array_equal(got, oracle.batched_constant_lnlike(full, many, *centre))

    # max-over-ranks timing reduction and barrier used by bench.py
    assert float(group.allreduce(np.array([float(rank + 1)]), op="max")[0]) == world
    assert int(group.allreduce(np.array([rank], dtype=np.int64), op="min")[0]) == 0
    group.barrier()

    # Runner on a rank context: every rank must start from the same walkers and draw the same proposals, or the
    # all-reduce would add partial sums of DIFFERENT parameter rows (silently wrong chains)
    import mcmc_dynamics_amd.analysis.runner as runner_mod
    from mcmc_dynamics_amd import DataReader
    from mcmc_dynamics_amd.analysis import ConstantFit
    runner_mod.Runner._ensure_catalog = lambda self: self._catalog
    fit = ConstantFit(DataReader({k: mine[k] for k in ("ra", "dec", "v", "verr")}), context=StubContext(rank, world, group),
                      seed=1000 + rank)                              # different global seeds on purpose
    fit.parameters["ra_center"].set(value=centre[0], fixed=True)
    fit.parameters["dec_center"].set(value=centre[1], fixed=True)
    for n, scale in (("v_sys", 1.0), ("sigma_max", None), ("v_maxx", 2.0), ("v_maxy", 2.0)):
        fit.parameters[n].set(initials="10.0 * rng.lognormal(sigma=0.05, size=n)" if scale is None else
                              "rng.normal(scale={0}, size=n)".format(scale))
    fit._catalog = ShardCatalog(mine, centre, group)
    fit._catalog_key = fit._plan().catalog_key
    sampler = fit(n_walkers=16, n_steps=6, n_out=3, prefix=None)
    chain = np.asarray(sampler.chain)
    assert chain.shape == (16, 6, 4) and np.all(np.isfinite(sampler.lnprobability))
    assert group.same_everywhere(chain) and group.same_everywhere(np.asarray(sampler.lnprobability))
    # the chain is the one a single process gets from the same start and seed on the full catalogue
    ref_lp = oracle.batched_constant_lnlike(full, chain[:, -1, :], *centre)
    assert np.max(np.abs(ref_lp - np.asarray(sampler.lnprobability)[:, -1]) / np.abs(ref_lp)) < 1e-12
    # a rank that proposes something else is caught by the checksum, not summed silently
    bad = chain[:, -1, :] + (1e-9 if rank == world - 1 else 0.0)
    fit._n_rank_batches = 0
    try:
        fit.lnprob_batch(bad)
        caught = False
    except RuntimeError as exc:
        caught = "different walker" in str(exc)
    assert caught
    group.barrier()
    if rank == 0:
        print("HOSTGROUP_OK world={0} err={1:.2e} calls={2}".format(world, err, fit._catalog.calls))
    group.close()


if __name__ == "__main__":
    main()
