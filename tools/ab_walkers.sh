#!/bin/bash
# A/B of library variants at 256 / 128 / 64 walkers (pipelined step time), C3 and const shapes
#     gpurun -- 'bash tools/ab_walkers.sh default nopf'
set -o pipefail
for v in "$@"; do
    lib=mcmc_dynamics_amd/libmcd_hip_$v.so
    [ "$v" = default ] && lib=mcmc_dynamics_amd/libmcd_hip.so
    for wl in c3 c3const c3gb; do
        for W in 256 128 64; do
            MCD_LIB_PATH=$PWD/$lib python bench.py --workload $wl --walkers $W --steps 300 --warmup 30 --no-cpu-baseline --no-mcmc --no-c4-strong 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v'.ljust(8), '$wl'.ljust(8), 'W', $W, ' value %.4e  step %.1f us  kernel %.1f us' % (d['value'], d['ms_per_step']*1e3, d['roofline']['kernel_us']))"
        done
    done
done
