"""world_size-2 (and 3) gloo runs of the sharded path on the CPU."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from mcmc_dynamics_amd import distributed


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 1000, 10 ** 7):
        for world in (1, 2, 3, 8):
            edges = [distributed.shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[r][1] == edges[r + 1][0] for r in range(world - 1))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        distributed.shard_bounds(10, 2, 2)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sum_equals_unsharded_gloo(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29511 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "DIST_OK world={0}".format(world) in res.stdout
