"""world_size-2 (and 3) runs of the sharded path on the CPU: the torch-free host group (what bench.py and
``distributed.rank_context`` use) under the driver's launcher, and gloo as a second, independent all-reduce."""
import os
import subprocess
import sys

import numpy as np
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


@pytest.mark.parametrize("world", [2, 3])
def test_host_group_under_the_drivers_launcher(world):
    """HostGroup rendezvous + collectives + a Runner on a rank context, launched exactly as the driver launches bench.py."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(29521 + world), os.path.join(ROOT, "tests", "hostgroup_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    assert "HOSTGROUP_OK world={0}".format(world) in res.stdout


def test_host_group_explicit_port_and_failure_modes(tmp_path):
    """MCD_RDZV_PORT mode (multi-node style) with plain subprocesses; a missing rank is a HostGroupError after the timeout,
    not a hang; mismatched collectives are detected."""
    import socket
    import textwrap
    from mcmc_dynamics_amd.hostgroup import HostGroup, HostGroupError
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = textwrap.dedent("""
        import sys, numpy as np
        sys.path.insert(0, {root!r})
        from mcmc_dynamics_amd.hostgroup import HostGroup
        g = HostGroup.from_env(timeout=60)
        x = g.allreduce(np.array([1.0 + g.rank, 10.0]), op="sum")
        assert x[0] == sum(1.0 + r for r in range(g.world)) and x[1] == 10.0 * g.world
        assert float(g.allreduce(3.5 * (g.rank + 1), op="max")) == 3.5 * g.world
        g.barrier()
        print("OK", g.rank)
        g.close()
    """).format(root=ROOT)
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="1", MCD_RDZV_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert sorted(o[0].strip() for o in outs) == ["OK 0", "OK 1", "OK 2"]
    # single-rank group needs no sockets at all
    g = HostGroup(0, 1)
    assert float(g.allreduce(2.0, op="sum")) == 2.0 and g.bcast_bytes(b"x") == b"x" and g.same_everywhere(np.arange(3))
    # rank 0 alone in a world of 2: rendezvous times out with an error
    env = {"MASTER_PORT": "7", "MCD_RDZV_FILE": str(tmp_path / "rdzv.json")}
    with pytest.raises(HostGroupError):
        HostGroup(0, 2, timeout=1.0, env=env)
    with pytest.raises(HostGroupError):
        HostGroup(1, 2, timeout=1.0, env=env)          # stale rendezvous file of the dead hub: no connection, error


@pytest.mark.parametrize("scenario", ["abort", "die"])
def test_host_group_abort_channel(scenario, tmp_path):
    """Round 3: a rank that fails tells the host group, and every rank's ``on_abort`` callbacks (in the
    product: ``Context.abort`` -> ``mcd_ctx_abort``, which ends a wait inside the device all-reduce) run within moments --
    also when the failing rank just dies without a word.  Afterwards every collective of the group raises."""
    import socket
    import textwrap
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    code = textwrap.dedent("""
        import os, sys, time, threading
        sys.path.insert(0, {root!r})
        from mcmc_dynamics_amd.hostgroup import HostGroup, HostGroupError
        g = HostGroup.from_env(timeout=60)
        seen = []
        flag = threading.Event()
        g.on_abort(lambda reason: (seen.append(reason), flag.set()))
        g.barrier()
        t0 = time.monotonic()
        if g.rank == 1:
            if {scenario!r} == "die":
                os._exit(9)                                   # no goodbye on the control connection
            g.abort("boom in a block")
        assert flag.wait(20.0), "no abort arrived"
        elapsed = time.monotonic() - t0
        assert elapsed < 5.0, elapsed
        want = "rank 1: boom in a block" if {scenario!r} == "abort" else "rank 1 closed its control connection"
        assert want in seen[0], seen
        try:
            g.barrier()
            raise SystemExit("collective after an abort did not raise")
        except HostGroupError as exc:
            assert "aborted" in str(exc)
        print("ABORT_SEEN", g.rank, round(elapsed, 3), flush=True)
        os._exit(3)                                           # what a failed rank does: report and leave, non-zero
    """).format(root=ROOT, scenario=scenario)
    procs = []
    for r in range(3):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="1", MCD_RDZV_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    codes = [p.returncode for p in procs]
    assert codes == ([3, 3, 3] if scenario == "abort" else [3, 9, 3]), (codes, outs)
    seen = sorted(o[0].split()[1] for o in outs if o[0].startswith("ABORT_SEEN"))
    assert seen == (["0", "1", "2"] if scenario == "abort" else ["0", "2"]), outs
