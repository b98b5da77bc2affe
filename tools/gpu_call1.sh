#!/bin/bash
# round-2 GPU call 1: parity of the new kernels + baseline numbers
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_nets_gpu.py 2>&1 | tail -15 > gpurun_out/c1_tests_kernels.log
timeout 900 python -m pytest tests/test_nets_gpu.py -m gpu -q 2>&1 | tail -25 > gpurun_out/c1_tests_nets.log
cp gpurun_out/parity_metrics.json gpurun_out/c1_parity_metrics.json 2>/dev/null
timeout 600 python tools/igemm_bench.py conv --variants 0,1,2,3,4,5,6 --rounds 3 --json gpurun_out/c1_conv.json > gpurun_out/c1_conv.log 2>&1
MGLD_CONV3Q=0 timeout 300 python tools/igemm_bench.py conv --variants 0 --rounds 3 > gpurun_out/c1_conv_old.log 2>&1
timeout 600 python tools/igemm_bench.py vae --variants 0,1,2,5 --rounds 2 > gpurun_out/c1_vae.log 2>&1
MGLD_CONV3Q=0 timeout 300 python tools/igemm_bench.py vae --variants 0 --rounds 2 > gpurun_out/c1_vae_old.log 2>&1
for o in 0 1 2; do MGLD_IGEMM_ORDER=$o timeout 300 python tools/igemm_bench.py lin --rounds 3 > gpurun_out/c1_lin_o$o.log 2>&1; done
timeout 300 python tools/igemm_bench.py lin --rounds 3 > gpurun_out/c1_lin_auto.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-shapes gpurun_out/c1_shapes.json > gpurun_out/c1_bench.log 2>&1
MGLD_CONV3Q=0 MGLD_IGEMM_ORDER=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-shapes gpurun_out/c1_shapes_old.json > gpurun_out/c1_bench_old.log 2>&1
tail -3 gpurun_out/c1_tests_kernels.log; tail -3 gpurun_out/c1_tests_nets.log; tail -2 gpurun_out/c1_conv.log; tail -1 gpurun_out/c1_conv_old.log; tail -1 gpurun_out/c1_bench.log | cut -c1-300; tail -1 gpurun_out/c1_bench_old.log | cut -c1-300
