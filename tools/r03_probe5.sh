#!/bin/bash
set -o pipefail
O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_device_chain.py tests/test_gpu_kernels.py tests/test_gpu_runner.py -x -q -m gpu > $O/r03_gpu_chain.log 2>&1 || { tail -40 $O/r03_gpu_chain.log; exit 1; }
tail -2 $O/r03_gpu_chain.log
rm -f $O/r03_chain_*.txt
timeout -k 10 200 python tools/chain_probe.py 100000 256 256 const > $O/r03_chain_c2.txt 2>&1 || { tail $O/r03_chain_c2.txt; exit 1; }
timeout -k 10 200 python tools/chain_probe.py 100000 256 256 > $O/r03_chain_c2bg.txt 2>&1 || exit 1
timeout -k 10 200 python tools/chain_probe.py 1000000 256 64 > $O/r03_chain_c3.txt 2>&1 || exit 1
timeout -k 10 200 python tools/chain_probe.py 10000 256 256 const > $O/r03_chain_10k.txt 2>&1 || exit 1
timeout -k 10 200 python tools/chain_probe.py 400000 256 128 const > $O/r03_chain_400k.txt 2>&1 || exit 1
echo chain done
