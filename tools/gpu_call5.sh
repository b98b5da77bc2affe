#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "tile2d or ring or linear or geglu" 2>&1 | tail -6 > gpurun_out/c5_tests_kernels.log
timeout 600 python -m pytest tests/test_nets_gpu.py -m gpu -q -k "tile_sharded or config0" 2>&1 | tail -12 > gpurun_out/c5_tests_nets.log
timeout 600 python -m pytest tests/test_cli_gpu.py -m gpu -q -k "fixed_size" 2>&1 | tail -12 > gpurun_out/c5_tests_cli.log
timeout 600 python tools/igemm_bench.py conv --variants 0,5,8,6,9 --rounds 4 > gpurun_out/c5_conv.log 2>&1
timeout 400 python tools/igemm_bench.py vae --variants 0,8 --rounds 2 > gpurun_out/c5_vae.log 2>&1
timeout 300 python tools/igemm_bench.py lin --rounds 4 > gpurun_out/c5_lin_pf1.log 2>&1
MGLD_HIP_LIB=$GRAFT_REPO_ROOT/_variants/libmgld_pf2.so timeout 300 python tools/igemm_bench.py lin --rounds 4 > gpurun_out/c5_lin_pf2.log 2>&1
timeout 300 python tools/igemm_bench.py lin --rounds 4 > gpurun_out/c5_lin_pf1b.log 2>&1
tail -3 gpurun_out/c5_tests_kernels.log; tail -3 gpurun_out/c5_tests_nets.log; tail -3 gpurun_out/c5_tests_cli.log; cat gpurun_out/harness_old_metrics.json gpurun_out/harness_wlat_metrics.json; grep -v amdgpu gpurun_out/c5_conv.log; grep -v amdgpu gpurun_out/c5_vae.log | tail -3; tail -1 gpurun_out/c5_lin_pf1.log; tail -1 gpurun_out/c5_lin_pf2.log; tail -1 gpurun_out/c5_lin_pf1b.log
