"""The bench lines kept under profiles/ (printed by bench.py on the GPU box) carry every field of the driver's
contract, and bench.py itself parses and exposes the contract's flags (no GPU needed)."""
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


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_*.json"))))
def test_saved_bench_lines_follow_the_contract(path):
    d = json.load(open(path))
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["metric"] == "star-walker log-L terms/sec" and d["unit"] == "terms/s"
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["scaling"] in ("weak", "strong") and "workload" in d["config"] and "model" not in d["config"]
    assert ROOFLINE <= set(d["roofline"]) and d["roofline"]["bound"] in ("hbm", "mfma")
    r = d["roofline"]
    assert r["peak"] == 8000.0 and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    terms = d["config"]["stars_total"] * d["config"]["walkers"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - terms) < 1e-6 * terms
    if d["cpu_baseline"] is not None:
        assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"]) and d["cpu_baseline"]["kind"] == "port"
        assert d["value"] > 1e3 * d["cpu_baseline"]["value"]
    assert d["value"] >= 1e9                                    # north-star floor: >= 1e9 star-walker terms/s on one GPU


def test_default_bench_line_is_the_headline_configuration():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_c3.json")))
    assert d["n_gpus"] == 1 and d["config"]["stars_per_gpu"] == 1000000 and d["config"]["walkers"] == 256
    assert d["dtype"] == "f64" and d["config"]["likelihood"] == "bgfixed"
    assert d["roofline"]["frac"] >= 0.6                          # north-star: >= 60 % of the HBM roofline (streaming model)
    assert d["gpu_vs_cpu_port_max_rel_err"] < 1e-12


def test_bench_cli():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload"):
        assert flag in out.stdout
