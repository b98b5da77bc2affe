#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "ring" 2>&1 | tail -6 > gpurun_out/c6_tests_kernels.log
timeout 400 python tools/igemm_bench.py lin --nst 0,1,9 --rounds 4 > gpurun_out/c6_lin.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c6_bench_dma.log 2>&1
MGLD_IGEMM_RS=1 timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c6_bench_rs.log 2>&1
tail -3 gpurun_out/c6_tests_kernels.log; grep -v amdgpu gpurun_out/c6_lin.log | cut -c1-220; tail -1 gpurun_out/c6_bench_dma.log | cut -c1-400; tail -1 gpurun_out/c6_bench_rs.log | cut -c1-400
