#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -6 > gpurun_out/c8_tests_kernels.log
timeout 400 python tools/igemm_bench.py lin --nst 0 --rounds 4 > gpurun_out/c8_lin.log 2>&1
timeout 400 python tools/igemm_bench.py conv --variants 0 --rounds 3 > gpurun_out/c8_conv.log 2>&1
MGLD_CONV3Q_PF=2 timeout 400 python tools/igemm_bench.py conv --variants 0 --rounds 3 > gpurun_out/c8_conv_pf2.log 2>&1
timeout 400 python tools/igemm_bench.py vae --variants 0 --rounds 2 > gpurun_out/c8_vae.log 2>&1
MGLD_CONV3Q_PF=2 timeout 400 python tools/igemm_bench.py vae --variants 0 --rounds 2 > gpurun_out/c8_vae_pf2.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c8_bench.log 2>&1
MGLD_CONV3Q_PF=2 timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c8_bench_pf2.log 2>&1
tail -3 gpurun_out/c8_tests_kernels.log; grep -v amdgpu gpurun_out/c8_lin.log | cut -c1-160; grep -v amdgpu gpurun_out/c8_conv.log | cut -c1-160 | tail -12;  grep -v amdgpu gpurun_out/c8_conv_pf2.log | cut -c1-160 | tail -12; tail -2 gpurun_out/c8_vae.log | cut -c1-200; tail -2 gpurun_out/c8_vae_pf2.log | cut -c1-200; tail -1 gpurun_out/c8_bench.log | cut -c1-300; tail -1 gpurun_out/c8_bench_pf2.log | cut -c1-300
