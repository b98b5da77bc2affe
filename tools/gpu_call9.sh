#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/c9_tests_kernels.log
timeout 400 python tools/igemm_bench.py conv --variants 0 --rounds 3 > gpurun_out/c9_conv.log 2>&1
timeout 400 python tools/igemm_bench.py vae --variants 0 --rounds 2 > gpurun_out/c9_vae.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c9_bench.log 2>&1
tail -3 gpurun_out/c9_tests_kernels.log; grep -v amdgpu gpurun_out/c9_conv.log | cut -c1-160 | tail -12; grep -v amdgpu gpurun_out/c9_vae.log | tail -12 | cut -c1-200; tail -1 gpurun_out/c9_bench.log | cut -c1-300
