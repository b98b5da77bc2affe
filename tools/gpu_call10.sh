#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "norm or spade" 2>&1 | tail -6 > gpurun_out/c10_tests_kernels.log
timeout 900 python -m pytest tests/test_nets_gpu.py -m gpu -q -x 2>&1 | tail -8 > gpurun_out/c10_tests_nets.log
timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c10_bench.log 2>&1
MGLD_GN_FUSED=0 timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c10_bench_nofuse.log 2>&1
tail -3 gpurun_out/c10_tests_kernels.log; tail -5 gpurun_out/c10_tests_nets.log; tail -1 gpurun_out/c10_bench.log | cut -c1-300; tail -1 gpurun_out/c10_bench_nofuse.log | cut -c1-300
