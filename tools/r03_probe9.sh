#!/bin/bash
set -o pipefail
O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_device_chain.py -q -m gpu -x > $O/r03_t.log 2>&1 || { tail -20 $O/r03_t.log; exit 1; }
tail -1 $O/r03_t.log
make -C tests/fake_rccl > /dev/null 2>&1
MCD_VISIBLE_DEVICES=1 MCD_ALLOW_SHARED_DEVICE=1 MCD_RCCL_LIBRARY=$PWD/tests/fake_rccl/libfake_rccl.so timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29591 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_r03_2rank_flow.json 2> $O/bench_r03_2rank_flow.err || { tail -20 $O/bench_r03_2rank_flow.err; exit 1; }
tail -c 600 $O/bench_r03_2rank_flow.json
bash tools/r03_probe8.sh
