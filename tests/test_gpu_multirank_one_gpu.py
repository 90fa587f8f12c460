"""GPU: the multi-device (`mcd_ctx_create(n_dev > 1)`: ncclCommInitAll + ncclGroupStart/End) and multi-rank
(`mcd_ctx_create_rank`: one process per rank) code paths of the library on ONE device.  RCCL refuses two ranks on one
GPU, so a host-staged stand-in (tests/fake_rccl, selected with MCD_RCCL_LIBRARY) carries the all-reduce; everything else
is the product: real shards with `star_begin > 0`, per-shard chunk tables and background sums, bins straddling shard
edges, narrow-range exception chunks, the re-run signal crossing shards / ranks as NaN-poisoned sums, double-buffered
pipelined results, `Runner.__call__` on a rank context.  The real RCCL call sites run on a 1-rank communicator in
tests/test_gpu_kernels.py::test_rccl_call_path_on_one_rank."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
WORKER = os.path.join(ROOT, "tests", "fake_rccl_worker.py")


@pytest.fixture(scope="module")
def fake_rccl():
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "fake_rccl")], check=True, capture_output=True)


@pytest.mark.parametrize("n_dev", [2, 3])
def test_one_process_several_shards_on_one_device(fake_rccl, n_dev):
    res = subprocess.run([sys.executable, WORKER, "single", str(n_dev)], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert "FAKE_RCCL_SINGLE_OK n_dev={0}".format(n_dev) in res.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_one_process_per_rank_on_one_device(fake_rccl, world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29561 + world), WORKER, "rank"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert "FAKE_RCCL_RANKS_OK world={0}".format(world) in res.stdout
