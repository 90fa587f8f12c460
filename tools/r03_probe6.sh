#!/bin/bash
set -o pipefail
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
rm -rf $O/prof_r03a_*
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03a_c2 -- python3 bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --no-mcmc > $O/prof_bench_r03a_c2.log 2>&1 || { tail -5 $O/prof_bench_r03a_c2.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03a_c3 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-mcmc --no-c4-strong > $O/prof_bench_r03a_c3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03a_chain_c2 -- python3 tools/chain_probe.py 100000 256 256 const > $O/prof_chain_r03a_c2.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --workload c2 > $O/bench_r03a_c2.json 2> $O/bench_r03a_c2.err || { tail -5 $O/bench_r03a_c2.err; exit 1; }
echo prof done
