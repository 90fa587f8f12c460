#!/bin/bash
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu > $O/r03_gpu_kernels.log 2>&1 || { tail -30 $O/r03_gpu_kernels.log; exit 1; }
tail -2 $O/r03_gpu_kernels.log
timeout -k 10 200 python tools/balance_sweep.py 100000 256 const > $O/r03_bal_c2.txt 2>&1 || { tail -5 $O/r03_bal_c2.txt; exit 1; }
timeout -k 10 200 python tools/balance_sweep.py 100000 128 const > $O/r03_bal_c2_w128.txt 2>&1 || { tail -5 $O/r03_bal_c2_w128.txt; exit 1; }
timeout -k 10 200 python tools/balance_sweep.py 100000 128 bgfixed > $O/r03_bal_c2bg_w128.txt 2>&1 || { tail -5 $O/r03_bal_c2bg_w128.txt; exit 1; }
timeout -k 10 200 python tools/balance_sweep.py 100000 256 bgfixed > $O/r03_bal_c2bg.txt 2>&1 || { tail -5 $O/r03_bal_c2bg.txt; exit 1; }
timeout -k 10 200 python tools/balance_sweep.py 1250000 256 const 0,8 > $O/r03_bal_c4s.txt 2>&1 || { tail -5 $O/r03_bal_c4s.txt; exit 1; }
timeout -k 10 200 python tools/balance_sweep.py 1000000 256 bgfixed 0,8 > $O/r03_bal_c3.txt 2>&1 || { tail -5 $O/r03_bal_c3.txt; exit 1; }
timeout -k 10 200 python tools/balance_sweep.py 1000000 128 bgfixed 0,8 > $O/r03_bal_c3_w128.txt 2>&1 || { tail -5 $O/r03_bal_c3_w128.txt; exit 1; }
timeout -k 10 200 python tools/balance_sweep.py 300000 256 const 0,4,6,8 > $O/r03_bal_300k.txt 2>&1 || { tail -5 $O/r03_bal_300k.txt; exit 1; }
echo sweep done
