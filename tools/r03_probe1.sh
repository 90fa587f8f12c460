#!/bin/bash
# round-3 probe batch 1: per-workgroup timeline of the main kernel at C2 / 128-walker / C4-shard shapes
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 200 python tools/w128_len_sweep.py 256 100000 const 0,196 > $O/r03_len_c2.txt 2>&1 || exit 1
export MCD_LIB_PATH=$PWD/mcmc_dynamics_amd/libmcd_hip_stamps.so
timeout -k 10 200 python tools/w128_len_sweep.py 256 100000 const 0,196 > $O/r03_len_c2_stamps.txt 2>&1 || exit 1
timeout -k 10 200 python tools/main_stamps_probe.py 100000 256 const 0,98,130,196,392 > $O/r03_stamps_c2.txt 2>&1 || { tail -5 $O/r03_stamps_c2.txt; exit 1; }
echo c2 done
timeout -k 10 200 python tools/main_stamps_probe.py 100000 128 const 0,196,392 > $O/r03_stamps_c2_w128.txt 2>&1 || { tail -5 $O/r03_stamps_c2_w128.txt; exit 1; }
timeout -k 10 200 python tools/main_stamps_probe.py 1250000 256 const 0 > $O/r03_stamps_c4shard.txt 2>&1 || { tail -5 $O/r03_stamps_c4shard.txt; exit 1; }
timeout -k 10 200 python tools/main_stamps_probe.py 100000 128 bgfixed 0,196 > $O/r03_stamps_c2bg_w128.txt 2>&1 || { tail -5 $O/r03_stamps_c2bg_w128.txt; exit 1; }
echo stamps done
unset MCD_LIB_PATH
timeout -k 10 200 python tools/chain_probe.py 100000 256 256 > $O/r03_chain_c2_base.txt 2>&1 || exit 1
echo chain done
