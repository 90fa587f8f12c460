#!/bin/bash
# Measurement sweep on the GPU box (repo root): GPU tests, smoke, every bench workload, rocprofv3 kernel statistics of
# C3, C2 and a resident stretch-move block (tools/chain_probe.py), the two PMC passes for roofline.traffic and the three SQ-counter passes.  Everything lands under
# gpurun_out/; tools/collect_profiles.sh then copies the summaries into profiles/ (see profiles/README.md).
#     /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/final_sweep.sh TAG'
set -o pipefail
TAG=${1:-sweep}
STAGE=${2:-all}          # a: tests + benches   b: kernel traces + campaigns   c: PMC / SQ counter passes   (a gpurun call lasts <= 20 min)
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
if [ "$STAGE" = all ] || [ "$STAGE" = a ]; then
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests_$TAG.log 2>&1 || { tail -20 $O/gpu_tests_$TAG.log; exit 1; }
tail -1 $O/gpu_tests_$TAG.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_$TAG.log 2>&1 || { tail -20 $O/smoke_$TAG.log; exit 1; }
tail -1 $O/smoke_$TAG.log
for w in c3 c2 c3gb c3const c4 m1; do
    timeout -k 10 300 python bench.py --workload $w > $O/bench_${TAG}_$w.json 2> $O/bench_${TAG}_$w.err || { tail -5 $O/bench_${TAG}_$w.err; exit 1; }
    echo "bench $w done"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_${TAG}_c3_k20.json 2> $O/bench_${TAG}_c3_k20.err || exit 1
for p in f64 f32acc64 f32; do
    timeout -k 10 300 python bench.py --workload c5 --precision $p > $O/bench_${TAG}_c5_$p.json 2> $O/bench_${TAG}_c5_$p.err || exit 1
done
timeout -k 10 200 python tests/probes/kde_probe.py > $O/kde_probe_$TAG.json 2> $O/kde_probe.err || exit 1
echo "benches done"
fi
if [ "$STAGE" = all ] || [ "$STAGE" = b ]; then
rm -rf $O/prof_${TAG}_c3 $O/prof_${TAG}_c2 $O/pmc_fetch_$TAG $O/pmc_write_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_c3 -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-mcmc --no-c4-strong > $O/prof_bench_${TAG}_c3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_c2 -- python3 bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --no-mcmc > $O/prof_bench_${TAG}_c2.log 2>&1 || exit 1
# the same two commands on ONE lane (launches do not overlap: the duration of a launch that has the chip to itself)
rm -rf $O/prof_${TAG}_c3_one_lane $O/prof_${TAG}_c2_one_lane
MCD_BENCH_OPTIONS=two_lanes=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_c3_one_lane -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-mcmc --no-c4-strong > $O/prof_bench_${TAG}_c3_one_lane.log 2>&1 || exit 1
MCD_BENCH_OPTIONS=two_lanes=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_c2_one_lane -- python3 bench.py --workload c2 --steps 200 --warmup 20 --no-cpu-baseline --no-mcmc > $O/prof_bench_${TAG}_c2_one_lane.log 2>&1 || exit 1
rm -rf $O/prof_${TAG}_chain
timeout -k 10 200 python tools/chain_probe.py > $O/chain_probe_$TAG.txt 2>&1 || { tail -5 $O/chain_probe_$TAG.txt; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_chain -- python3 tools/chain_probe.py 1000000 256 64 > $O/prof_chain_$TAG.log 2>&1 || exit 1
timeout -k 10 200 python tools/chain_probe.py 100000 256 256 const > $O/chain_probe_c2_$TAG.txt 2>&1 || { tail -5 $O/chain_probe_c2_$TAG.txt; exit 1; }
rm -rf $O/prof_${TAG}_chain_c2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${TAG}_chain_c2 -- python3 tools/chain_probe.py 100000 256 256 const > $O/prof_chain_c2_$TAG.log 2>&1 || exit 1
for shape in "100000 256 const" "100000 128 const" "100000 256 bgfixed" "1250000 256 const 0,8" "1000000 256 bgfixed 0,8"; do
    timeout -k 10 200 python tools/balance_sweep.py $shape >> $O/balance_sweep_$TAG.txt 2>&1 || { tail -5 $O/balance_sweep_$TAG.txt; exit 1; }
done
timeout -k 10 300 python tools/fuzz_f32.py --seconds 100 --seed 5 > $O/fuzz_f32_$TAG.txt 2>&1 || { tail -5 $O/fuzz_f32_$TAG.txt; exit 1; }
timeout -k 10 300 python tools/fuzz_gpu.py --seconds 100 --schedule --max-walkers 640 --max-stars 5000 > $O/fuzz_gpu_$TAG.txt 2>&1 || { tail -5 $O/fuzz_gpu_$TAG.txt; exit 1; }
timeout -k 10 300 python tools/fuzz_chain.py --seconds 60 > $O/fuzz_chain_$TAG.log 2>&1 || { tail -5 $O/fuzz_chain_$TAG.log; exit 1; }
MCD_CHAIN_PART_BYTES=1 timeout -k 10 300 python tools/fuzz_chain.py --seconds 60 --seed 7 >> $O/fuzz_chain_$TAG.log 2>&1 || { tail -5 $O/fuzz_chain_$TAG.log; exit 1; }
timeout -k 10 300 python tools/fuzz_chain.py --seconds 60 --seed 11 --force-rccl >> $O/fuzz_chain_$TAG.log 2>&1 || { tail -5 $O/fuzz_chain_$TAG.log; exit 1; }
timeout -k 10 300 python tools/fuzz_chain.py --seconds 60 --seed 13 --seeded >> $O/fuzz_chain_$TAG.log 2>&1 || { tail -5 $O/fuzz_chain_$TAG.log; exit 1; }
MCD_CHAIN_PART_BYTES=1 timeout -k 10 300 python tools/fuzz_chain.py --seconds 40 --seed 17 --seeded >> $O/fuzz_chain_$TAG.log 2>&1 || { tail -5 $O/fuzz_chain_$TAG.log; exit 1; }
echo "kernel traces done"
fi
if [ "$STAGE" = all ] || [ "$STAGE" = c ]; then
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$TAG -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mcmc --no-c4-strong > $O/pmc_fetch_$TAG.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$TAG -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mcmc --no-c4-strong > $O/pmc_write_$TAG.log 2>&1 || exit 1
echo "traffic passes done"
A="SQ_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES"
B="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"
i=0
for set in "$A" "$B" "$C"; do
    i=$((i+1))
    rm -rf $O/sq_${TAG}_pass$i
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/sq_${TAG}_pass$i -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mcmc --no-c4-strong > $O/sq_${TAG}_pass$i.log 2>&1 || { tail -5 $O/sq_${TAG}_pass$i.log; exit 1; }
done
# SQ counters of one float32 launch of the C5 sweep
i=0
for set in "$A" "$B" "$C"; do
    i=$((i+1))
    rm -rf $O/sqf32_${TAG}_pass$i
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/sqf32_${TAG}_pass$i -- python3 bench.py --workload c5 --precision f32 --steps 20 --warmup 5 --no-cpu-baseline --no-mcmc > $O/sqf32_${TAG}_pass$i.log 2>&1 || { tail -5 $O/sqf32_${TAG}_pass$i.log; exit 1; }
done
./tools/valu_rate_probe > $O/valu_rate_probe_$TAG.txt 2>&1 || true
./tools/launch_floor_probe > $O/launch_floor_probe_$TAG.txt 2>&1 || true
fi
echo "sweep $TAG stage $STAGE complete"
