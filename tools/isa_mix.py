#!/usr/bin/env python3
"""Instruction mix of the hot loops of mcd::loglike_kernel, from the gfx950 ISA hipcc emits.

    python tools/isa_mix.py            # compiles csrc/mcd_kernels.hip with -save-temps into /tmp and prints a table

For every fast-path instantiation the largest loop body is located (label .. backward branch), its instructions are
counted and priced with the issue costs measured by tools/valu_rate_probe.hip on MI355X: one "slot" = one f64
wave-instruction per SIMD = 2.33 ns on the fully occupied chip; v_rsq/v_rcp_f64 = 2.9 slots; other (integer,
cndmask, mov, f32) VALU instructions = 0.5 slot.  predicted kernel time for 1e6 stars x 256 walkers =
slots/term x 2.33 ns x (2.56e8 / 64 lanes) / 1024 SIMDs.
"""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mcmc_dynamics_amd", "csrc")
SLOT_NS = 2.33

KERNELS = [  # (template tag, name, stars per loop iteration)
    ("ILi0ELb0EddLi1E", "CONST fixed centre", 16), ("ILi0ELb1EddLi1E", "CONST free centre", 8),
    ("ILi1ELb0EddLi1E", "BGFIXED fixed centre", 4), ("ILi1ELb0EddLi2E", "BGFIXED fixed, narrow", 4), ("ILi2ELb0EddLi1E", "BGGAUSS fixed centre", 4),
    ("ILi2ELb0EddLi2E", "BGGAUSS fixed, narrow", 4),
    ("ILi3ELb0EddLi1E", "PROFILE fixed centre", 8), ("ILi4ELb0EddLi1E", "PROFILE_BGGAUSS fixed", 4),
    ("ILi5ELb0EddLi1E", "PROFILE_BGDENS fixed", 4), ("ILi0ELb0EffLi1E", "CONST fixed, f32", 16),
]


def main():
    out = "/tmp/isa_mix"
    os.makedirs(out, exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c",
                    os.path.join(CSRC, "mcd_kernels.hip"), "-o", os.path.join(out, "k.o"), "-save-temps=obj"],
                   check=True, capture_output=True)
    asm = open(os.path.join(out, "mcd_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    print("{0:26s} {1:>9s} {2:>8s} {3:>8s} {4:>10s} {5:>14s}".format("kernel (fast path)", "VALU/term", "f64", "other", "slots/term",
                                                                    "predicted us"))
    for tag, name, per in KERNELS:
        starts = [i for i, l in enumerate(asm) if l.startswith("_ZN3mcd12_GLOBAL__N_114loglike_kernel" + tag)]
        if not starts:
            continue
        a = starts[0]
        e = next(i for i in range(a, len(asm)) if asm[i].startswith(".Lfunc_end"))   # kernels may have several s_endpgm
        k = asm[a:e]
        labels = {l.split(":")[0]: i for i, l in enumerate(k) if re.match(r"^\.LBB\d+_\d+:", l)}
        spans = []
        for i, l in enumerate(k):
            m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                spans.append((labels[m.group(1)], i))
        # innermost loops only (no other loop nested inside); the hot loop is the largest of them
        inner = [sp for sp in spans if not any(o != sp and sp[0] <= o[0] and o[1] <= sp[1] for o in spans)]
        ranked = sorted(inner, key=lambda sp: sp[1] - sp[0], reverse=True)
        # a narrow-range instantiation (FAST = 2) carries the general 4-star loop as well (per-chunk choice): its own
        # hot loop is the second largest
        best = ranked[1] if tag.endswith("Li2E") and len(ranked) > 1 else ranked[0]
        body = [l.split()[0] for l in k[best[0]:best[1]] if l.startswith("\t") and not l.strip().startswith((";", "."))]
        c = Counter(body)
        f64 = sum(v for o, v in c.items() if o.startswith("v_") and "f64" in o)
        valu = sum(v for o, v in c.items() if o.startswith("v_"))
        slots = sum(v * (2.9 if o.startswith(("v_rsq_f64", "v_rcp_f64")) else 1.0 if "f64" in o else 0.5)
                    for o, v in c.items() if o.startswith("v_"))
        pred = slots / per * SLOT_NS * (2.56e8 / 64) / 1024 * 1e-3
        print("{0:26s} {1:9.2f} {2:8.2f} {3:8.2f} {4:10.2f} {5:14.1f}".format(name, valu / per, f64 / per, (valu - f64) / per,
                                                                           slots / per, pred))


if __name__ == "__main__":
    sys.exit(main())
