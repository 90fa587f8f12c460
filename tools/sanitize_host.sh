#!/bin/bash
# AddressSanitizer + UBSan over the host-side C++ the library shares with the CPU harness (csrc/mcd_stretch.h: the
# stretch-move loop incl. binned ensembles; mcd_chunks.h: chunk tables; mcd_guard.h: range guard; the lane arithmetic of
# mcd_math.h compiled for the host) -- CPU only (GPU sanitizers are not available on the pool).  Repo root:
#     bash tools/sanitize_host.sh
set -e
cd "$(dirname "$0")/.."
cp tests/emul/libmcd_emul.so /tmp/libmcd_emul.so.plain 2>/dev/null || true
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer \
    -I mcmc_dynamics_amd/csrc tests/emul/mcd_emul.cpp -o tests/emul/libmcd_emul.so
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_stretch_block_cpu.py tests/test_chunk_plan_cpu.py tests/test_guard_random_cpu.py tests/test_kernel_math_cpu.py tests/test_chain_rng_cpu.py -x -q
rc=$?
rm -f tests/emul/libmcd_emul.so                       # the next test run rebuilds the plain library
[ -f /tmp/libmcd_emul.so.plain ] && cp /tmp/libmcd_emul.so.plain tests/emul/libmcd_emul.so && touch tests/emul/libmcd_emul.so
exit $rc
