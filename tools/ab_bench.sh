#!/bin/bash
# A/B timing of library variants built by `make -C mcmc_dynamics_amd/csrc variant NAME=.. DEFS=..` (repo root, GPU box):
#     gpurun -- 'bash tools/ab_bench.sh "c3 c3gb" default t8 n3'
# prints value (terms/s) and the HIP-event kernel time per workload and variant; full lines under gpurun_out/ab_*.json
set -o pipefail
WORKLOADS=${1:-c3}
shift
O=gpurun_out
mkdir -p $O
for w in $WORKLOADS; do
    for v in "$@"; do
        lib=mcmc_dynamics_amd/libmcd_hip_$v.so
        [ "$v" = default ] && lib=mcmc_dynamics_amd/libmcd_hip.so
        MCD_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline --no-mcmc \
            > $O/ab_${w}_$v.json 2> $O/ab_${w}_$v.err || { tail -5 $O/ab_${w}_$v.err; exit 1; }
        python - "$w" "$v" $O/ab_${w}_$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("{0:8s} {1:10s} value {2:.4e} terms/s   step {3:.1f} us   kernel {4:.1f} us".format(
    sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"] * 1e3, d["roofline"]["kernel_us"]))
PY
    done
done
