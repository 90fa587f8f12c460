#!/bin/bash
# A/B of a catalogue option on the default library (repo root, GPU box):
#     gpurun -- 'bash tools/ab_option.sh "c2 c3" prefetch -1 0 1'
# bench.py forwards MCD_BENCH_OPTIONS="key=value,..." to mcd_set_option on every catalogue it creates.
set -o pipefail
WORKLOADS=${1:-c3}
KEY=$2
shift 2
O=gpurun_out
mkdir -p $O
for w in $WORKLOADS; do
    for v in "$@"; do
        MCD_BENCH_OPTIONS="$KEY=$v" timeout -k 10 200 python bench.py --workload $w --steps 300 --warmup 30 --no-cpu-baseline --no-mcmc --no-c4-strong \
            > $O/abo_${w}_${KEY}_$v.json 2> $O/abo_${w}_${KEY}_$v.err || { tail -5 $O/abo_${w}_${KEY}_$v.err; exit 1; }
        python - "$w" "$KEY=$v" $O/abo_${w}_${KEY}_$v.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[3]).read().strip().splitlines()[-1])
print("{0:8s} {1:14s} value {2:.4e} terms/s   step {3:.1f} us   kernel {4:.1f} us".format(
    sys.argv[1], sys.argv[2], d["value"], d["ms_per_step"] * 1e3, d["roofline"]["kernel_us"]))
PY
    done
done
