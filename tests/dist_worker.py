"""Worker of tests/test_distributed_cpu.py: one process per rank, gloo backend, CPU only.

Exercises the N > 1 host path of bench.py / mcmc_dynamics_amd.distributed -- star sharding, identical
walkers on every rank, one all-reduce(sum) of the per-walker partial log-likelihoods -- with the oracle
standing in for the per-rank kernel (the RCCL call itself lives in libmcd_hip.so and needs GPUs)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from mcmc_dynamics_amd import distributed, synthetic      # noqa: E402
from oracle import lnprob_numpy as oracle                 # noqa: E402


def main():
    rank, world, _ = distributed.env_rank()
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    centre = (synthetic.CENTER_RA_DEG, synthetic.CENTER_DEC_DEG)
    full = synthetic.make_catalog(4001, config=4)                    # odd size: uneven shards
    names = ["v_sys", "sigma_max", "v_maxx", "v_maxy"]
    pos = synthetic.make_walkers(12, names, full["truth"], config=4)

    # the unique-id hand-off pattern of distributed.rank_context / bench.py
    box = [bytes(range(128)) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    assert box[0] == bytes(range(128))

    mine = distributed.shard_columns(full, rank, world)
    lo, hi = distributed.shard_bounds(4001, rank, world)
    assert len(mine["v"]) == hi - lo and mine["truth"] == full["truth"]
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([hi - lo]))
    assert sum(int(s) for s in sizes) == 4001 and max(int(s) for s in sizes) - min(int(s) for s in sizes) <= 1

    partial = torch.from_numpy(oracle.batched_constant_lnlike(mine, pos, *centre))
    dist.all_reduce(partial, op=dist.ReduceOp.SUM)                   # what ncclAllReduce does on the GPUs
    want = oracle.batched_constant_lnlike(full, pos, *centre)
    err = np.max(np.abs(partial.numpy() - want) / np.abs(want))
    assert err < 1e-13, err

    # the other axis: full catalogue on every rank, walkers split across ranks, slices gathered over the host group
    many = synthetic.make_walkers(13, names, full["truth"], config=4)          # odd count: uneven slices
    got = distributed.replicated_loglike(lambda p: oracle.batched_constant_lnlike(full, p, *centre), many, rank, world)
    ref = oracle.batched_constant_lnlike(full, many, *centre)
    assert got.shape == (13,) and np.array_equal(got, ref)

    # binned catalogue: bins that straddle the shard boundary contribute partial sums from both ranks
    dx, dy = oracle.calc_xy_offset(full["ra"], full["dec"], *centre)
    bins = oracle.make_radial_bins(np.hypot(dx, dy), 300, 0.05).astype(np.int64)
    order = np.argsort(bins, kind="stable")
    srt = {k: (v[order] if isinstance(v, np.ndarray) else v) for k, v in full.items()}
    offs = np.concatenate([[0], np.cumsum(np.bincount(bins))])
    my_offs = distributed.shard_bin_offsets(offs, rank, world)
    my_srt = distributed.shard_columns(srt, rank, world)
    assert my_offs[0] == 0 and my_offs[-1] == len(my_srt["v"])
    per_bin = np.zeros((len(offs) - 1, len(pos)))
    for b in range(len(offs) - 1):
        sub = {k: v[my_offs[b]:my_offs[b + 1]] for k, v in my_srt.items() if isinstance(v, np.ndarray)}
        if len(sub["v"]):
            per_bin[b] = oracle.batched_constant_lnlike(sub, pos, *centre)
    t = torch.from_numpy(per_bin)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    for b in range(len(offs) - 1):
        sub = {k: v[offs[b]:offs[b + 1]] for k, v in srt.items() if isinstance(v, np.ndarray)}
        ref = oracle.batched_constant_lnlike(sub, pos, *centre)
        assert np.max(np.abs(t.numpy()[b] - ref) / np.abs(ref)) < 1e-13

    # max-over-ranks timing reduction used by bench.py
    tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    assert float(tt[0]) == world
    dist.barrier()
    if rank == 0:
        print("DIST_OK world={0} err={1:.2e}".format(world, err))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
