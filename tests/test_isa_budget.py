"""Static check of the gfx950 code hipcc emits for the hot loops (cross-compiled, no GPU needed): the per-term VALU
instruction counts DESIGN.md section 3 quotes are a budget -- a compiler or source change that adds instructions to an
inner loop (a canonicalising v_max, a spilled register, a lost scalar load) should fail here, not in a profile later."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

BUDGET = {                                   # VALU instructions per star-walker term, tools/isa_mix.py
    "CONST fixed centre": 8.55,
    "CONST free centre": 29.1,
    "BGFIXED fixed centre": 33.8,
    "BGFIXED fixed, narrow": 23.6,
    "BGGAUSS fixed centre": 53.6,
    "BGGAUSS fixed, narrow": 45.1,
    "PROFILE fixed centre": 25.1,
    "PROFILE fixed, narrow": 22.6,       # round 3: no reciprocal per term, one-step Newton root (ProfileNarrowAcc)
    # the instantiations that prefetch the next iteration's records (mcd_math.h: RecordPrefetch): + 2 instructions per
    # iteration (lane index and predicate are hoisted, the touch and its exec mask are not)
    "CONST fixed centre, prefetch": 8.7,
    "BGFIXED fixed, narrow, prefetch": 24.1,
    "BGGAUSS fixed, narrow, prefetch": 46.1,
}


@pytest.fixture(scope="module")
def isa_table():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py")], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    rows = {}
    for line in res.stdout.splitlines():
        m = re.match(r"^(.*?)\s{2,}([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line)
        if m:
            rows[m.group(1).strip()] = float(m.group(2))
    return rows


def test_inner_loops_stay_within_their_instruction_budget(isa_table):
    for name, limit in BUDGET.items():
        assert name in isa_table, (name, sorted(isa_table))
        assert isa_table[name] <= limit, (name, isa_table[name], limit)


def test_hot_kernels_use_scalar_record_loads_and_no_scratch():
    asm_path = "/tmp/isa_mix/mcd_kernels-hip-amdgcn-amd-amdhsa-gfx950.s"
    if not os.path.exists(asm_path):
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mix.py")], capture_output=True, timeout=900, check=True)
    asm = open(asm_path).read()
    for tag, vgpr_limit, scratch_limit in (
            ("ILi0ELb0EddLi1ELb0ELi4E", 64, 0), ("ILi1ELb0EddLi2ELb0ELi4E", 64, 0), ("ILi1ELb0EddLi2ELb1ELi4E", 64, 0),
            ("ILi2ELb0EddLi2ELb1ELi4E", 128, 0),
            # the combining workgroups of the balanced plans: same budgets (16 waves = 1024 threads: 128 VGPRs).  The 8-wave
            # kernel of the per-walker Gaussian background is held to the 4-wave kernel's 128 registers by its launch
            # bound (left alone it takes 150 and loses a wave per SIMD: 53 - 58 us against 46.5 at 1e5 stars x 256 walkers)
            # and pays with three to five spilled dwords outside the loop
            ("ILi0ELb0EddLi1ELb0ELi8E", 64, 0), ("ILi0ELb0EddLi1ELb0ELi16E", 64, 0), ("ILi1ELb0EddLi2ELb0ELi16E", 64, 0),
            ("ILi2ELb0EddLi2ELb0ELi8E", 128, 32)):
        m = re.search(r"\n(_ZN3mcd12_GLOBAL__N_114loglike_kernel" + tag + r"[^\n:]*):[^\n]*\n(.*?)\n\.Lfunc_end", asm, re.S)
        assert m, tag
        body = m.group(2)
        assert "s_load_dwordx" in body                       # star records arrive through the scalar cache
        assert scratch_limit or "scratch_" not in body       # no register spills in the hot kernels
        meta = re.search(r"\.amdhsa_kernel " + re.escape(m.group(1)) + r"\n(.*?)\.end_amdhsa_kernel", asm, re.S).group(1)
        assert int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1)) <= scratch_limit
        # CONST and BGFIXED: 8 waves per SIMD; the Gaussian-background kernel trades occupancy for unrolling (4 waves)
        assert int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)) <= vgpr_limit


def test_committed_instruction_counts_belong_to_the_committed_sources():
    """mcmc_dynamics_amd/csrc/isa_mix.json (what bench.py's roofline block prices the kernels with) carries a hash of the
    kernel sources and of the extraction rules: it must be the tree's (the committed file was stale, every
    build() left the tree dirty)."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_mix
    with open(os.path.join(ROOT, "mcmc_dynamics_amd", "csrc", "isa_mix.json")) as f:
        committed = json.load(f)
    assert committed["source_sha16"] == isa_mix.source_hash(), "run: python tools/isa_mix.py --json mcmc_dynamics_amd/csrc/isa_mix.json"
    for key in ("const", "bgfixed", "bggauss", "profile", "profile_general", "const_f32", "const_f32acc64"):
        assert key in committed["kernels"] and committed["kernels"][key]["valu_per_term"] > 0
