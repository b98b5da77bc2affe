#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "ring_depth or tile2d or conv3x3_patch" 2>&1 | tail -8 > gpurun_out/c2_tests_kernels.log
timeout 600 python -m pytest tests/test_nets_gpu.py -m gpu -q -k "config0 or hoisting or vae_small" 2>&1 | tail -25 > gpurun_out/c2_tests_nets.log
cp gpurun_out/parity_metrics.json gpurun_out/c2_parity_metrics.json 2>/dev/null
timeout 600 python tools/igemm_bench.py lin --nst 1,2,3 --rounds 3 > gpurun_out/c2_lin_nst.log 2>&1
timeout 300 python tools/igemm_bench.py conv --variants 0 --rounds 3 > gpurun_out/c2_conv.log 2>&1
for n in 2 3 4; do MGLD_IGEMM_NST=$n timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/c2_bench_nst$n.log 2>&1; done
tail -3 gpurun_out/c2_tests_kernels.log; tail -3 gpurun_out/c2_tests_nets.log; tail -1 gpurun_out/c2_lin_nst.log; for n in 2 3 4; do tail -1 gpurun_out/c2_bench_nst$n.log | cut -c1-200; done
