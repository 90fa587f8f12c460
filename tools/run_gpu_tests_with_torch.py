"""Scratch: run the GPU test-suite in a process where torch (and its bundled HIP runtime / RCCL) is loaded first --
the library situation of `bench.py --gpus N` under torch.distributed.run."""
import sys
import torch                      # noqa: F401  (loads torch/lib/libamdhip64.so, librccl.so)
import torch.distributed          # noqa: F401
import pytest
sys.exit(pytest.main(["tests", "-m", "gpu", "-q", "-x"]))
