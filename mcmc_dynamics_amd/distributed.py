"""One-process-per-GPU operation: star sharding and communicator bootstrap.

The log-likelihood is a sum of independent per-star terms (analysis/runner.py:269-271, :286), so the
catalogue shards naturally over GPUs: rank g uploads the contiguous star range ``shard_bounds(N, g, G)``
and every batched evaluation ends in ONE RCCL all-reduce (sum, f64) of the per-walker partial
log-likelihoods (``W`` doubles; ``B x W`` for a binned catalogue) inside ``libmcd_hip.so``.  The reference
has no counterpart: its only parallelism is a process pool over walkers (runner.py:398-403).

The host side only hands rank 0's 128-byte RCCL unique id to the other ranks (and, for ``Runner``, the start
positions and the sampler seed): that goes over ``hostgroup.HostGroup`` -- plain TCP, no PyTorch anywhere in the
package, so every process keeps the ROCm runtime and RCCL the library was built against.
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


def replicated_loglike(loglike_fn, params, rank=None, world=None, group=None):
    """The other axis ("replicas"): every rank holds the FULL catalogue and evaluates only its
    contiguous slice of the walkers; the slices are exchanged over the host group (no device collective).  For
    catalogues too small to fill several GPUs.  ``loglike_fn((w, K)) -> (w,)``; returns the complete ``(W,)`` on every
    rank.  ``group``: a ``hostgroup.HostGroup`` (created from the environment when omitted and world > 1)."""
    if rank is None or world is None:
        rank, world, _ = env_rank()
    params = np.asarray(params, dtype=np.float64)
    lo, hi = shard_bounds(len(params), rank, world)
    mine = np.asarray(loglike_fn(params[lo:hi]), dtype=np.float64) if hi > lo else np.empty(0)
    if world == 1:
        return mine
    if group is None:
        group = default_group()
    return np.concatenate(group.allgather_array(mine))


def env_rank():
    """(rank, world, local_rank) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


_default_group = None


def default_group(timeout=300.0):
    """Process-wide ``HostGroup`` built from the launcher's environment on first use."""
    global _default_group
    if _default_group is None:
        from .hostgroup import HostGroup
        _default_group = HostGroup.from_env(timeout=timeout)
    return _default_group


def rank_context(group=None, device=None):
    """Create the ``_native.Context`` of this rank: rank 0 asks RCCL for a unique id, the host group carries it to the
    other ranks, every rank joins the communicator.  The group stays attached to the context (``ctx.host_group``):
    ``Runner`` uses it to start every rank from the same walkers and the same sampler seed."""
    from . import _native
    rank, world, local_rank = env_rank()
    dev = local_rank if device is None else int(device)
    if world == 1:
        ctx = _native.Context(n_devices=1, device_ids=[dev])
        ctx.host_group = None
        return ctx
    if group is None:
        group = default_group()
    uid = group.bcast_bytes(_native.Context.unique_id() if rank == 0 else None, src=0)
    ctx = _native.Context(rank=rank, n_ranks=world, unique_id=uid, device=dev)
    ctx.host_group = group
    # a peer that fails says so over the group's abort channel: a call of this rank that is waiting inside the device
    # all-reduce then returns an error at once instead of running into the library's collective deadline
    group.on_abort(ctx.abort)
    return ctx
