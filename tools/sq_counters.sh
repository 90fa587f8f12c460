#!/bin/bash
# SQ counter passes for the default bench (C3), one rocprofv3 --pmc run each (8 SQ slots per pass on gfx950).
#     /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/sq_counters.sh'
set -o pipefail
O=gpurun_out
export TMPDIR=/tmp
A="SQ_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES"
B="SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS"
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT"
i=0
for set in "$A" "$B" "$C"; do
    i=$((i+1))
    rm -rf $O/sq_pass$i
    timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $O/sq_pass$i -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-mcmc > $O/sq_pass$i.log 2>&1 || { tail -5 $O/sq_pass$i.log; exit 1; }
    echo "pass $i done"
done
