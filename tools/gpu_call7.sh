#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/c7_tests_kernels.log
timeout 400 python tools/igemm_bench.py lin --nst 0 --rounds 4 > gpurun_out/c7_lin.log 2>&1
timeout 400 python tools/igemm_bench.py conv --variants 0 --rounds 3 > gpurun_out/c7_conv.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c7_bench.log 2>&1
tail -3 gpurun_out/c7_tests_kernels.log; grep -v amdgpu gpurun_out/c7_lin.log | cut -c1-160; grep -v amdgpu gpurun_out/c7_conv.log | cut -c1-160 | tail -25; tail -1 gpurun_out/c7_bench.log | cut -c1-600
