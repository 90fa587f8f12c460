"""One-process-per-GPU operation: star sharding and communicator bootstrap.

The log-likelihood is a sum of independent per-star terms (analysis/runner.py:269-271, :286), so the
catalogue shards naturally over GPUs: rank g uploads the contiguous star range ``shard_bounds(N, g, G)``
and every batched evaluation ends in ONE RCCL all-reduce (sum, f64) of the per-walker partial
log-likelihoods (``W`` doubles; ``B x W`` for a binned catalogue) inside ``libmcd_hip.so``.  The reference
has no counterpart: its only parallelism is a process pool over walkers (runner.py:398-403).

The host-side group (any ``torch.distributed`` backend, gloo is enough) is used only to hand rank 0's
RCCL unique id to the other ranks.
"""
import os

import numpy as np

STAR_COLUMNS = ("ra", "dec", "v", "verr", "density", "pmember", "lnlike_bg")


def shard_bounds(n_stars, rank, world):
    """Contiguous, balanced star range [lo, hi) of ``rank`` (sizes differ by at most one)."""
    n_stars, rank, world = int(n_stars), int(rank), int(world)
    if not 0 <= rank < world:
        raise ValueError("rank {0} outside world of size {1}".format(rank, world))
    return n_stars * rank // world, n_stars * (rank + 1) // world


def shard_columns(columns, rank, world):
    """Slice every per-star array of ``columns`` (dict) to this rank's range; other entries pass through."""
    n = len(columns["v"])
    lo, hi = shard_bounds(n, rank, world)
    return {k: (v[lo:hi] if isinstance(v, np.ndarray) and v.shape[:1] == (n,) else v) for k, v in columns.items()}


def shard_bin_offsets(bin_offsets, rank, world):
    """Bin offsets of a sorted-by-bin catalogue, restricted to this rank's star range and re-based to it.
    Bins that straddle a shard boundary appear (partially) on both ranks; their partial sums add up."""
    offs = np.asarray(bin_offsets, dtype=np.int64)
    lo, hi = shard_bounds(offs[-1], rank, world)
    return np.clip(offs, lo, hi) - lo


def replicated_loglike(loglike_fn, params, rank=None, world=None, process_group=None):
    """The other axis ("replicas"): every rank holds the FULL catalogue and evaluates only its
    contiguous slice of the walkers; the slices are exchanged over the host process group (no device collective).  For
    catalogues too small to fill several GPUs.  ``loglike_fn((w, K)) -> (w,)``; returns the complete ``(W,)`` on every rank."""
    import torch
    import torch.distributed as dist
    if rank is None or world is None:
        rank, world, _ = env_rank()
    params = np.asarray(params, dtype=np.float64)
    lo, hi = shard_bounds(len(params), rank, world)
    mine = np.asarray(loglike_fn(params[lo:hi]), dtype=np.float64) if hi > lo else np.empty(0)
    if world == 1:
        return mine
    sizes = [shard_bounds(len(params), r, world) for r in range(world)]
    width = max(h - l for l, h in sizes)
    padded = torch.zeros(width, dtype=torch.float64)
    padded[:hi - lo] = torch.from_numpy(mine)
    parts = [torch.zeros(width, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(parts, padded, group=process_group)
    return np.concatenate([parts[r][:h - l].numpy() for r, (l, h) in enumerate(sizes)])


def env_rank():
    """(rank, world, local_rank) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def rank_context(process_group=None):
    """Create the ``_native.Context`` of this rank, distributing the RCCL unique id through an already
    initialised ``torch.distributed`` process group (gloo or nccl)."""
    from . import _native
    rank, world, local_rank = env_rank()
    if world == 1:
        return _native.Context(n_devices=1, device_ids=[local_rank])
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    box = [_native.Context.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=process_group)
    return _native.Context(rank=rank, n_ranks=world, unique_id=box[0], device=local_rank)
