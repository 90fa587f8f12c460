#!/bin/bash
set -o pipefail
O=gpurun_out
mkdir -p $O
for n in 200000 400000 600000 800000; do
  timeout -k 10 200 python tools/balance_sweep.py $n 256 const 0,4,6,8 >> $O/r03_cross_const.txt 2>&1 || { tail -5 $O/r03_cross_const.txt; exit 1; }
  timeout -k 10 200 python tools/balance_sweep.py $n 256 bgfixed 0,4,6,8 >> $O/r03_cross_bgfixed.txt 2>&1 || { tail -5 $O/r03_cross_bgfixed.txt; exit 1; }
done
for n in 30000 10000 3000; do
  timeout -k 10 200 python tools/balance_sweep.py $n 256 const 0,1,2,4,6,8 >> $O/r03_small_const.txt 2>&1 || { tail -5 $O/r03_small_const.txt; exit 1; }
  timeout -k 10 200 python tools/balance_sweep.py $n 256 bgfixed 0,1,2,4,6,8 >> $O/r03_small_bgfixed.txt 2>&1 || { tail -5 $O/r03_small_bgfixed.txt; exit 1; }
done
timeout -k 10 200 python tools/balance_sweep.py 500000 128 bgfixed 0,4,6,8 >> $O/r03_cross_w128.txt 2>&1 || exit 1
timeout -k 10 200 python tools/balance_sweep.py 500000 128 const 0,4,6,8 >> $O/r03_cross_w128.txt 2>&1 || exit 1
timeout -k 10 200 python tools/balance_sweep.py 100000 64 const 0,2,4,6,8 >> $O/r03_cross_w64.txt 2>&1 || exit 1
timeout -k 10 200 python tools/balance_sweep.py 100000 64 bgfixed 0,2,4,6,8 >> $O/r03_cross_w64.txt 2>&1 || exit 1
timeout -k 10 200 python tools/balance_sweep.py 100000 512 const 0,2,4,6,8 >> $O/r03_cross_w512.txt 2>&1 || exit 1
timeout -k 10 200 python tools/balance_sweep.py 100000 192 const 0,2,3,6 >> $O/r03_cross_w192.txt 2>&1 || exit 1
timeout -k 10 200 python tools/balance_sweep.py 100000 256 bggauss 0,4,6,8 >> $O/r03_cross_bggauss.txt 2>&1 || exit 1
echo sweep done
