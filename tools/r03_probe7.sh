#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests -q -m gpu -x > $O/r03_gpu_all.log 2>&1 || { tail -30 $O/r03_gpu_all.log; exit 1; }
tail -1 $O/r03_gpu_all.log
for w in c3 c2 c3const c4; do
  for lanes in 1 0; do
    MCD_BENCH_OPTIONS="two_lanes=$lanes" timeout -k 10 300 python bench.py --workload $w --no-mcmc --no-cpu-baseline --no-c4-strong > $O/bench_r03l_${w}_$lanes.json 2> $O/bench_r03l_${w}_$lanes.err || { tail -5 $O/bench_r03l_${w}_$lanes.err; exit 1; }
  done
done
echo lanes done
