#!/usr/bin/env python3
"""Instruction mix of the hot loops of mcd::loglike_kernel, from the gfx950 ISA hipcc emits.

    python tools/isa_mix.py                 # compiles csrc/mcd_kernels.hip with -save-temps into /tmp, prints a table
    python tools/isa_mix.py --json OUT      # additionally writes the table as JSON (what bench.py's roofline block reads;
                                            # __graft_entry__.build() refreshes mcmc_dynamics_amd/csrc/isa_mix.json)

For every fast-path instantiation the hot loop nest is located (label .. backward branch) and its vector instructions are
counted per star-walker term.  The narrow-range mixture variants rescale their running product on every second 4-star
iteration (a block behind a scalar branch): their count is (2 x loop body + rescale block) / 8.

"slots" prices the mix with the issue costs measured on MI355X (tools/valu_rate_probe.hip): an f64 FMA/MUL/ADD wave-
instruction = 1 slot (4 cycles on one SIMD), v_rsq/v_rcp_f64 = 2.9 slots, other VALU instructions (integer, v_ldexp,
v_frexp, moves, f32) = 0.5 slot.
"""
import hashlib
import json
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mcmc_dynamics_amd", "csrc")
SOURCES = ("mcd_kernels.hip", "mcd_math.h", "mcd_exp_table.h", "mcd_internal.h", "mcd_chunks.h", "mcd_reduce.h")
SLOT_NS = 2.33

# (template tag, name, bench model key, stars per inner iteration, inner trips per outer iteration, selector)
#   selector(Counter of the loop body) -> bool picks the loop among the kernel's innermost loops
def _sel(rsq=None, frexp=None, rcp=None):
    def f(c):
        return ((rsq is None or c["v_rsq_f64_e32"] == rsq) and (frexp is None or c["v_frexp_mant_f64_e32"] == frexp)
                and (rcp is None or c["v_rcp_f64_e32"] == rcp))
    return f


KERNELS = [
    ("ILi0ELb0EddLi1E", "CONST fixed centre", "const", 16, 1, _sel(rsq=0, rcp=1)),
    ("ILi0ELb1EddLi1E", "CONST free centre", "const_free", 8, 1, None),
    ("ILi1ELb0EddLi1E", "BGFIXED fixed centre", "bgfixed_general", 4, 1, _sel(rsq=4, frexp=4)),
    ("ILi1ELb0EddLi2E", "BGFIXED fixed, narrow", "bgfixed", 4, 2, _sel(rsq=4, frexp=0)),
    ("ILi2ELb0EddLi1E", "BGGAUSS fixed centre", "bggauss_general", 4, 1, _sel(rsq=8, frexp=4)),
    ("ILi2ELb0EddLi2E", "BGGAUSS fixed, narrow", "bggauss", 4, 2, _sel(rsq=8, frexp=0)),
    ("ILi3ELb0EddLi1E", "PROFILE fixed centre", "profile_general", 8, 1, None),
    ("ILi3ELb0EddLi2E", "PROFILE fixed, narrow", "profile", 8, 1, _sel(rsq=8, rcp=1)),
    ("ILi4ELb0EddLi1E", "PROFILE_BGGAUSS fixed", "profile_bggauss", 4, 1, _sel(frexp=4)),
    ("ILi5ELb0EddLi1E", "PROFILE_BGDENS fixed", "profile_bgdens", 4, 1, _sel(frexp=4)),
    ("ILi0ELb0EffLi1E", "CONST fixed, f32", "const_f32", 16, 1, None),
    ("ILi0ELb0EfdLi1E", "CONST fixed, f32 terms f64 sums", "const_f32acc64", 16, 1, None),
    ("ILi1ELb0EffLi1E", "BGFIXED fixed, f32", "bgfixed_f32", 4, 1, None),
    ("ILi1ELb0EfdLi1E", "BGFIXED fixed, f32 terms f64 sums", "bgfixed_f32acc64", 4, 1, None),
]


def _ops(lines):
    return [l.split()[0] for l in lines if l.startswith("\t") and not l.strip().startswith((";", "."))]


def source_hash():
    h = hashlib.sha256()
    for name in SOURCES:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    with open(os.path.abspath(__file__), "rb") as f:       # the extraction rules are part of what the numbers mean
        h.update(f.read())
    return h.hexdigest()[:16]


def analyse(out="/tmp/isa_mix"):
    os.makedirs(out, exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c",
                    os.path.join(CSRC, "mcd_kernels.hip"), "-o", os.path.join(out, "k.o"), "-save-temps=obj"],
                   check=True, capture_output=True)
    asm = open(os.path.join(out, "mcd_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    rows = []
    # every fast kernel exists with and without the software prefetch of the next iteration's records (template
    # parameter PF, the last one of the mangled name): the main row is the instantiation without, `*_prefetch` fields and
    # a second table line give the one with
    # (the 4-wave instantiations: the combining 8- / 16-wave ones of the balanced plans run the same loops)
    variants = [(tag + "Lb0ELi4EE", name, key, per, trips, selector, False) for tag, name, key, per, trips, selector in KERNELS]
    variants += [(tag + "Lb1ELi4EE", name + ", prefetch", key, per, trips, selector, True) for tag, name, key, per, trips, selector in KERNELS]
    for tag, name, key, per, trips, selector, with_prefetch in variants:
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
        inner = [sp for sp in spans if not any(o != sp and sp[0] <= o[0] and o[1] <= sp[1] for o in spans)]
        ranked = sorted(inner, key=lambda sp: sp[1] - sp[0], reverse=True)
        best = None
        if selector is not None:
            hits = [sp for sp in ranked if selector(Counter(_ops(k[sp[0]:sp[1]])))]
            best = hits[0] if hits else None
        if best is None:
            best = ranked[0]
        body = Counter(_ops(k[best[0]:best[1] + 1]))
        extra = Counter()
        if trips > 1:
            # the rescale block of the narrow-range loops runs on every `trips`-th iteration: it sits behind the loop's
            # closing conditional branch and jumps back with an unconditional s_branch
            for j in range(best[1] + 1, min(best[1] + 16, len(k))):
                m = re.match(r"\s*s_branch (\.LBB\d+_\d+)", k[j])
                if m:
                    if m.group(1) in labels and labels[m.group(1)] <= best[1]:
                        extra = Counter(_ops(k[best[1] + 1:j]))
                    break
        terms = per * trips
        total = Counter()
        for op, v in body.items():
            total[op] += v * trips
        for op, v in extra.items():
            total[op] += v

        def is_f64(o):
            return "f64" in o
        valu = sum(v for o, v in total.items() if o.startswith("v_"))
        f64 = sum(v for o, v in total.items() if o.startswith("v_") and is_f64(o) and not o.startswith(("v_ldexp", "v_frexp")))
        trans = sum(v for o, v in total.items() if o.startswith(("v_rsq_f64", "v_rcp_f64")))
        slots = sum(v * (2.9 if o.startswith(("v_rsq_f64", "v_rcp_f64")) else
                         1.0 if (is_f64(o) and not o.startswith(("v_ldexp", "v_frexp"))) else 0.5)
                    for o, v in total.items() if o.startswith("v_"))
        rows.append({"name": name, "model": key, "prefetch": with_prefetch, "stars_per_iteration": terms,
                     "valu_per_term": valu / terms, "f64_per_term": f64 / terms, "trans_f64_per_term": trans / terms,
                     "other_per_term": (valu - f64) / terms, "slots_per_term": slots / terms,
                     "lds_per_term": sum(v for o, v in total.items() if o.startswith("ds_")) / terms,
                     "salu_smem_per_term": sum(v for o, v in total.items() if o.startswith("s_")) / terms})
    return rows


def merge_variants(rows):
    """{model: row of the instantiation without prefetch + `<field>_prefetch` for the one with}"""
    out = {r["model"]: dict(r) for r in rows if not r["prefetch"]}
    for r in rows:
        if r["prefetch"] and r["model"] in out:
            for k in ("valu_per_term", "f64_per_term", "other_per_term", "slots_per_term", "salu_smem_per_term"):
                out[r["model"]][k + "_prefetch"] = r[k]
    for r in out.values():
        del r["prefetch"]
    return out


def main():
    rows = analyse()
    print("{0:34s} {1:>9s} {2:>8s} {3:>8s} {4:>10s} {5:>14s}".format("kernel (fast path)", "VALU/term", "f64", "other",
                                                                    "slots/term", "predicted us"))
    for r in rows:
        pred = r["slots_per_term"] * SLOT_NS * (2.56e8 / 64) / 1024 * 1e-3
        print("{0:34s} {1:9.2f} {2:8.2f} {3:8.2f} {4:10.2f} {5:14.1f}".format(r["name"], r["valu_per_term"], r["f64_per_term"],
                                                                           r["other_per_term"], r["slots_per_term"], pred))
    if len(sys.argv) == 3 and sys.argv[1] == "--json":
        with open(sys.argv[2], "w") as f:
            json.dump({"source_sha16": source_hash(), "generated_by": "tools/isa_mix.py",
                       "note": "VALU wave-instructions per star-walker term in the hot loop nest of mcd::loglike_kernel "
                               "(gfx950 ISA from hipcc -save-temps); prologue, final log and tails not included",
                       "kernels": merge_variants(rows)}, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    sys.exit(main())
