#!/bin/bash
set -o pipefail
O=gpurun_out
mkdir -p $O
rm -f $O/r03_w16_*.txt
timeout -k 10 200 python tools/balance_sweep.py 100000 256 const 0,4,8 >> $O/r03_w16_const.txt 2>&1 || { tail -5 $O/r03_w16_const.txt; exit 1; }
timeout -k 10 200 python tools/balance_sweep.py 100000 128 const 0,2,4,8 >> $O/r03_w16_const.txt 2>&1 || exit 1
timeout -k 10 200 python tools/balance_sweep.py 400000 256 const 0,4,8 >> $O/r03_w16_const.txt 2>&1 || exit 1
timeout -k 10 200 python tools/balance_sweep.py 100000 256 bgfixed 0,4,8 >> $O/r03_w16_bg.txt 2>&1 || exit 1
timeout -k 10 200 python tools/balance_sweep.py 100000 128 bgfixed 0,2,4,6,8 >> $O/r03_w16_bg.txt 2>&1 || exit 1
timeout -k 10 200 python tools/balance_sweep.py 400000 256 bgfixed 0,4,8 >> $O/r03_w16_bg.txt 2>&1 || exit 1
timeout -k 10 200 python tools/balance_sweep.py 100000 256 bggauss 0,2,4,6,8 >> $O/r03_w16_gg.txt 2>&1 || exit 1
timeout -k 10 200 python tools/balance_sweep.py 100000 128 bggauss 0,2,4,6,8 >> $O/r03_w16_gg.txt 2>&1 || exit 1
echo sweep done
