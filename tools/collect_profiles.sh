#!/bin/bash
# Copy the summaries of a tools/final_sweep.sh run from gpurun_out/ into profiles/ (tracked).
#     bash tools/collect_profiles.sh TAG ROUND      e.g.  bash tools/collect_profiles.sh r02a r02
set -e
TAG=$1
R=$2
O=gpurun_out
P=profiles
last() { tail -1 "$1"; }
for w in c3 c2 c3gb c3const c4 m1 c3_k20 c5_f64 c5_f32acc64 c5_f32; do
    last $O/bench_${TAG}_$w.json | python -m json.tool > $P/${R}_bench_$w.json
done
cp $O/kde_probe_$TAG.json $P/${R}_kde_probe.json
python tools/summarize_rocprof.py stats $O/prof_${TAG}_c3 $P/${R}_c3_kernel_stats.txt > /dev/null
python tools/summarize_rocprof.py stats $O/prof_${TAG}_c2 $P/${R}_c2_kernel_stats.txt > /dev/null
cp "$(ls -t $O/prof_${TAG}_c3/*/*kernel_stats.csv | head -1)" $P/${R}_c3_kernel_stats.csv
cp "$(ls -t $O/prof_${TAG}_c2/*/*kernel_stats.csv | head -1)" $P/${R}_c2_kernel_stats.csv
for w in c3 c2; do
    if [ -d $O/prof_${TAG}_${w}_one_lane ]; then
        python tools/summarize_rocprof.py stats $O/prof_${TAG}_${w}_one_lane $P/${R}_${w}_kernel_stats_one_lane.txt > /dev/null
    fi
done
if [ -d $O/prof_${TAG}_chain_c2 ]; then
    python tools/summarize_rocprof.py stats $O/prof_${TAG}_chain_c2 $P/${R}_chain_c2_kernel_stats.txt > /dev/null
    python tools/chain_trace_summary.py $O/prof_${TAG}_chain_c2/*/*_kernel_trace.csv > $P/${R}_chain_c2_kernel_trace_summary.txt
    cp $O/chain_probe_c2_$TAG.txt $P/${R}_chain_probe_c2.txt
fi
[ -f $O/balance_sweep_$TAG.txt ] && cp $O/balance_sweep_$TAG.txt $P/${R}_balance_sweep.txt
[ -f $O/fuzz_f32_$TAG.txt ] && { echo "# tools/fuzz_f32.py --seconds 100 --seed 5 (sweep $TAG)"; grep -v '^trial' $O/fuzz_f32_$TAG.txt; } > $P/${R}_fuzz_f32_sweep.txt
[ -f $O/fuzz_gpu_$TAG.txt ] && { echo "# tools/fuzz_gpu.py --seconds 100 --schedule --max-walkers 640 --max-stars 5000 (sweep $TAG): fast vs plain kernels, float64"; grep -v '^trial' $O/fuzz_gpu_$TAG.txt; } > $P/${R}_fuzz_gpu.txt
[ -f $O/valu_rate_probe_$TAG.txt ] && cp $O/valu_rate_probe_$TAG.txt $P/${R}_valu_rate_probe_sweep.txt
[ -f $O/launch_floor_probe_$TAG.txt ] && cp $O/launch_floor_probe_$TAG.txt $P/${R}_launch_floor_probe.txt
if [ -d $O/sqf32_${TAG}_pass1 ]; then
    python tools/summarize_rocprof.py sq $P/${R}_c5_f32_sq_counters.json loglike_kernel 512000000 $O/sqf32_${TAG}_pass1 $O/sqf32_${TAG}_pass2 $O/sqf32_${TAG}_pass3 > /dev/null
fi
python tools/summarize_rocprof.py stats $O/prof_${TAG}_chain $P/${R}_chain_kernel_stats.txt > /dev/null
python tools/chain_trace_summary.py $O/prof_${TAG}_chain/*/*_kernel_trace.csv > $P/${R}_chain_kernel_trace_summary.txt
cp $O/chain_probe_$TAG.txt $P/${R}_chain_probe.txt
{ echo "# tools/fuzz_chain.py: resident vs host-driven stretch blocks, every bit compared; plain / every block cut into parts (MCD_CHAIN_PART_BYTES=1) / --force-rccl / --seeded (device-generated numbers: resident = host-driven = replay through mcd_chain_numbers) / --seeded in parts"; grep '^DONE' $O/fuzz_chain_$TAG.log; } > $P/${R}_fuzz_chain.txt
# (stage c of the sweep: the counter passes; a sweep without it keeps the committed counters -- the hot kernels decide them)
if [ -d $O/pmc_fetch_$TAG ]; then
    python tools/summarize_rocprof.py pmc c3 $O/pmc_fetch_$TAG $O/pmc_write_$TAG $P/pmc_traffic.json loglike_kernel 64000000 > /dev/null
    python tools/summarize_rocprof.py sq $P/${R}_c3_sq_counters.json loglike_kernel 256000000 $O/sq_${TAG}_pass1 $O/sq_${TAG}_pass2 $O/sq_${TAG}_pass3 > /dev/null
fi
echo "profiles/${R}_* refreshed from sweep $TAG"
