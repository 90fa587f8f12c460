#!/bin/bash
# Long randomised campaigns (evidence beyond the sweep's short ones): every bit of resident vs host-driven stretch blocks in
# five modes, fast vs plain kernels, float32 accuracy domain.      gpurun --timeout 1200 -- 'bash tools/long_campaign.sh TAG'
set -o pipefail
TAG=${1:-long}
S=${2:-0}          # added to every seed: a second run covers other cases
O=gpurun_out
mkdir -p $O
L=$O/long_campaign_$TAG.log
: > $L
run() { echo "### $*" >> $L; timeout -k 10 400 "$@" >> $L 2>&1 || { tail -5 $L; exit 1; }; }
run python tools/fuzz_chain.py --trials 2000 --seconds 150 --seed $((101 + S))
run env MCD_CHAIN_PART_BYTES=1 python tools/fuzz_chain.py --trials 2000 --seconds 100 --seed $((103 + S))
run python tools/fuzz_chain.py --trials 2000 --seconds 100 --seed $((107 + S)) --force-rccl
run python tools/fuzz_chain.py --trials 2000 --seconds 150 --seed $((109 + S)) --seeded
run env MCD_CHAIN_PART_BYTES=1 python tools/fuzz_chain.py --trials 2000 --seconds 100 --seed $((113 + S)) --seeded
run python tools/fuzz_chain.py --trials 2000 --seconds 60 --seed $((301 + S)) --seeded --force-rccl
run python tools/fuzz_gpu.py --trials 100000 --seconds 150 --schedule --max-walkers 640 --max-stars 5000 --seed $((127 + S))
run python tools/fuzz_f32.py --seconds 120 --seed $((131 + S))
grep -E '^###|^DONE' $L
