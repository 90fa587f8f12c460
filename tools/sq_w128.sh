#!/bin/bash
# SQ counters of the 128-walker launch (what a stretch-move half-step evaluates): where does the time go vs W = 256?
set -o pipefail
O=gpurun_out
export TMPDIR=/tmp
A="SQ_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES"
B="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"
for W in 128 64; do
i=0
for set in "$A" "$B" "$C"; do
    i=$((i+1))
    rm -rf $O/sq_w${W}_pass$i
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/sq_w${W}_pass$i -- python3 bench.py --walkers $W --steps 20 --warmup 5 --no-cpu-baseline --no-mcmc --no-c4-strong > $O/sq_w${W}_pass$i.log 2>&1 || { tail -5 $O/sq_w${W}_pass$i.log; exit 1; }
done
python tools/summarize_rocprof.py sq $O/sq_w${W}.json loglike_kernel $((1000000*W)) $O/sq_w${W}_pass1 $O/sq_w${W}_pass2 $O/sq_w${W}_pass3
done
