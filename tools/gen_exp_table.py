"""Generate mcmc_dynamics_amd/csrc/mcd_exp_table.h: 2^(j/N) correctly rounded to f64 (60-digit Decimal arithmetic), the
split ln 2 / N = hi + lo and the polynomial used by the table-driven exp of the kernels (mcd_math.h: exp_tab), for
N = 256 (degree-4 Taylor polynomial) and N = 1024 (degree-3 polynomial with a levelled even part); the build picks one
with -DMCD_EXP_TAB_BITS=8|10 (default 10).

    python tools/gen_exp_table.py
"""
import os
import struct
from decimal import Decimal, getcontext

getcontext().prec = 60


def emit(f, N):
    ln2 = Decimal(2).ln()
    rows = [float((ln2 * j / N).exp()).hex() for j in range(N)]          # Decimal -> float rounds correctly
    rows2 = [float((ln2 * (Decimal(j) / N + Decimal("0.5"))).exp()).hex() for j in range(N)]   # sqrt(2) 2^(j/N)
    step = ln2 / N
    bits = struct.unpack("<Q", struct.pack("<d", float(step)))[0] & ~((1 << 22) - 1)   # keep 30 mantissa bits
    hi = struct.unpack("<d", struct.pack("<Q", bits))[0]
    lo = float(step - Decimal(hi))
    inv = float(Decimal(N) / ln2)
    R = ln2 / (2 * N)
    f.write("constexpr int kExpTabBits = %d;\nconstexpr int kExpTabSize = 1 << kExpTabBits;\n" % (N.bit_length() - 1))
    f.write("constexpr double kExpTabInvStep = %s;   // %d / ln 2\n" % (inv.hex(), N))
    f.write("constexpr double kExpTabStepHi = %s;    // ln 2 / %d, upper 30 mantissa bits (k * hi exact for |k| < 2^22)\n" % (hi.hex(), N))
    f.write("constexpr double kExpTabStepLo = %s;\n" % lo.hex())
    if N >= 1024:
        # e^r on |r| <= R = ln 2 / 2N as 1 + r + c2 r^2 + c3 r^3: the even part of the error, r^4/24 - (c2 - 1/2) r^2, is
        # levelled (equal ripple) by c2 = 1/2 + 2 (sqrt 2 - 1) R^2 / 24, which leaves (3 - 2 sqrt 2) R^4 / 24;
        # the odd part beyond c3 = 1/6 is r^5/120.
        c2 = float(Decimal("0.5") + 2 * (Decimal(2).sqrt() - 1) * R * R / 24)
        err = float((3 - 2 * Decimal(2).sqrt()) * R ** 4 / 24)
        f.write("// e^r = 1 + r + c2 r^2 + c3 r^3 on |r| <= ln 2 / %d, equal-ripple even part: max error %.2e\n" % (2 * N, err))
        f.write("constexpr int kExpPolyDegree = 3;\nconstexpr double kExpPolyC2 = %s;\nconstexpr double kExpPolyC3 = %s;\n"
                "constexpr double kExpPolyC4 = 0.0;\n\n" % (c2.hex(), float(Decimal(1) / 6).hex()))
    else:
        err = float(R ** 5 / 120)
        f.write("// e^r = degree-4 Taylor polynomial on |r| <= ln 2 / %d: remainder r^5/120 <= %.2e\n" % (2 * N, err))
        f.write("constexpr int kExpPolyDegree = 4;\nconstexpr double kExpPolyC2 = 0.5;\nconstexpr double kExpPolyC3 = %s;\n"
                "constexpr double kExpPolyC4 = %s;\n\n" % (float(Decimal(1) / 6).hex(), float(Decimal(1) / 24).hex()))
    f.write("#define MCD_EXP_TABLE_VALUES \\\n")
    for i in range(0, N, 4):
        f.write("    " + ", ".join(rows[i:i + 4]) + (", \\\n" if i + 4 < N else "\n"))
    f.write("\n// sqrt(2) * 2^(j/N): the table of the BGFIXED kernel, which works with g / sqrt(2) = (2 n)^(-1/2)\n")
    f.write("#define MCD_EXP_TABLE_SQRT2_VALUES \\\n")
    for i in range(0, N, 4):
        f.write("    " + ", ".join(rows2[i:i + 4]) + (", \\\n" if i + 4 < N else "\n"))


def main():
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mcmc_dynamics_amd", "csrc",
                       "mcd_exp_table.h")
    with open(out, "w") as f:
        f.write("// mcd_exp_table.h -- 2^(j/N), j = 0..N-1, correctly rounded to f64, N = 256 or 1024 (generated: tools/gen_exp_table.py).\n")
        f.write("#pragma once\n\n#ifndef MCD_EXP_TAB_BITS\n#define MCD_EXP_TAB_BITS 10\n#endif\n\nnamespace mcd {\n\n")
        f.write("#if MCD_EXP_TAB_BITS == 8\n")
        emit(f, 256)
        f.write("#elif MCD_EXP_TAB_BITS == 10\n")
        emit(f, 1024)
        f.write("#else\n#error \"MCD_EXP_TAB_BITS must be 8 or 10\"\n#endif\n")
        f.write("\n}  // namespace mcd\n")


if __name__ == "__main__":
    main()
