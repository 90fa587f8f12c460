"""The bench lines kept under profiles/ (printed by bench.py on the GPU box) carry every field of the driver's
contract, bench.py itself parses and exposes the contract's flags, and its multi-rank failure path ends with a marked
line and a non-zero exit status (no GPU needed)."""
import glob
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}
ROOFLINE = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
VALU_F64_PEAK = 1024 * 2.4e9 / 4            # wave64 f64 vector instructions per second, whole chip


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*.json"))))
def test_saved_bench_lines_follow_the_contract(path):
    d = json.load(open(path))
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["metric"] == "star-walker log-L terms/sec" and d["unit"] == "terms/s"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["scaling"] in ("weak", "strong") and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert ROOFLINE <= set(r)
    if os.path.basename(path).startswith("r01_"):
        # round 1 priced the kernels against the streaming-model HBM figure (frac > 1: register reuse, not bandwidth)
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    elif d["dtype"] == "f64":
        # from round 2 on: the roof that binds -- f64 vector-instruction issue -- with a fraction the judge can recompute
        assert r["bound"] == "valu_f64" and abs(r["peak"] - VALU_F64_PEAK) < 1.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.0 < r["frac"] <= 1.0
        recomputed = r["valu_per_term"] * r["terms_per_launch"] / 64.0 / (r["kernel_us"] * 1e-6)
        assert abs(recomputed - r["achieved"]) < 1e-6 * r["achieved"]
        assert r["traffic"] is None or "imported" in r["traffic_source"]
        assert d["degraded"] is False
    terms = d["config"]["stars_total"] * d["config"]["walkers"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - terms) < 1e-6 * terms
    if d["cpu_baseline"] is not None:
        assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
        assert d["value"] > 1e3 * d["cpu_baseline"]["value"]
    assert d["value"] >= 1e9                                    # north-star floor: >= 1e9 star-walker terms/s on one GPU


def test_default_bench_line_is_the_headline_configuration():
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_c3.json")))
    d = json.load(open(paths[-1]))                               # the newest round's headline line
    assert d["n_gpus"] == 1 and d["config"]["stars_per_gpu"] == 1000000 and d["config"]["walkers"] == 256
    assert d["dtype"] == "f64" and d["config"]["likelihood"] == "bgfixed"
    assert d["gpu_vs_cpu_port_max_rel_err"] < 1e-12
    if not os.path.basename(paths[-1]).startswith("r01_"):
        # every default run also carries the north-star strong-scaling record (one 1e7-star catalogue over the ranks)
        c4 = d["c4_strong"]
        assert c4["stars_total"] == 10000000 and c4["scaling"] == "strong" and c4["walkers"] == 256
        assert abs(c4["terms_per_s"] * c4["ms_per_step"] * 1e-3 - 2.56e9) < 1e-6 * 2.56e9


def test_bench_cli():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload"):
        assert flag in out.stdout


def test_bench_imports_no_torch():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in src and "torch.distributed as" not in src


def test_collective_failure_ends_with_a_marked_line_and_a_nonzero_status():
    """Two ranks under the driver's launcher on a box without GPUs: the communicator cannot be created, so rank 0 must
    print the line with "degraded": true and "value": null, and the job must not exit 0."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd="/tmp")
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    if res.returncode == 0:
        # a box with two usable GPUs: the run may simply succeed
        assert lines and json.loads(lines[-1])["degraded"] is False
        return
    assert len(lines) == 1, res.stdout[-2000:] + res.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["degraded"] is True and d["value"] is None and d["n_gpus"] == 2
    assert d["failure"]["stage"] and d["failure"]["reason"]
    assert "collective failure" in res.stderr
