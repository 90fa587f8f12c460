"""Generate mcmc_dynamics_amd/csrc/mcd_exp_table.h: 2^(j/256) correctly rounded to f64 (60-digit Decimal arithmetic),
and the split ln 2 / 256 = hi + lo used by the table-driven exp of the kernels (mcd_math.h: exp_tab).

    python tools/gen_exp_table.py
"""
import os
import struct
from decimal import Decimal, getcontext

getcontext().prec = 60
N = 256


def main():
    ln2 = Decimal(2).ln()
    rows = [float((ln2 * j / N).exp()).hex() for j in range(N)]          # Decimal -> float rounds correctly
    rows2 = [float((ln2 * (Decimal(j) / N + Decimal("0.5"))).exp()).hex() for j in range(N)]   # sqrt(2) 2^(j/256)
    step = ln2 / N
    bits = struct.unpack("<Q", struct.pack("<d", float(step)))[0] & ~((1 << 20) - 1)   # keep 32 mantissa bits
    hi = struct.unpack("<d", struct.pack("<Q", bits))[0]
    lo = float(step - Decimal(hi))
    inv = float(Decimal(N) / ln2)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mcmc_dynamics_amd", "csrc",
                       "mcd_exp_table.h")
    with open(out, "w") as f:
        f.write("// mcd_exp_table.h -- 2^(j/256), j = 0..255, correctly rounded to f64 (generated: tools/gen_exp_table.py).\n")
        f.write("#pragma once\n\nnamespace mcd {\n\nconstexpr int kExpTabBits = 8;\nconstexpr int kExpTabSize = 1 << kExpTabBits;\n")
        f.write("constexpr double kExpTabInvStep = %s;   // 256 / ln 2\n" % inv.hex())
        f.write("constexpr double kExpTabStepHi = %s;    // ln 2 / 256, upper 32 mantissa bits (k * hi exact for |k| < 2^20)\n" % hi.hex())
        f.write("constexpr double kExpTabStepLo = %s;\n\n" % lo.hex())
        f.write("#define MCD_EXP_TABLE_VALUES \\\n")
        for i in range(0, N, 4):
            f.write("    " + ", ".join(rows[i:i + 4]) + (", \\\n" if i + 4 < N else "\n"))
        f.write("\n// sqrt(2) * 2^(j/256): the table of the BGFIXED kernel, which works with g / sqrt(2) = (2 n)^(-1/2)\n")
        f.write("#define MCD_EXP_TABLE_SQRT2_VALUES \\\n")
        for i in range(0, N, 4):
            f.write("    " + ", ".join(rows2[i:i + 4]) + (", \\\n" if i + 4 < N else "\n"))
        f.write("\n}  // namespace mcd\n")


if __name__ == "__main__":
    main()
