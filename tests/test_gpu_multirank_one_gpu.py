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


@pytest.mark.parametrize("kind", ["hang", "raise", "die"])
def test_a_collective_that_never_completes_ends_in_an_error_not_a_hang(fake_rccl, kind):
    """the PRODUCT path (Runner / _native, not the bench harness) has a collective deadline.  Two ranks on
    one device over the stand-in library, started as plain processes (the launcher would tear the job down at the first
    non-zero exit): "hang" -- rank 1's all-reduce never completes, both ranks return MCD_ERR_RCCL at collective_timeout_ms;
    "raise" / "die" -- rank 1 fails inside a block of Runner.__call__ / its process dies, the host group's abort channel
    reaches rank 0, whose wait inside the resident block ends at once.  Every rank exits non-zero; nothing is re-routed."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="1",
                   MCD_RDZV_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, WORKER, "deadline", kind], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung: {0}".format(kind))
    codes = [p.returncode for p in procs]
    assert codes == ([3, 9] if kind == "die" else [3, 3]), (codes, [o[0][-1500:] + o[1][-2500:] for o in outs])
    assert "DEADLINE_OK 0 " + kind in outs[0][0], outs[0]
    if kind != "die":
        assert "DEADLINE_OK 1 " + kind in outs[1][0], outs[1]
