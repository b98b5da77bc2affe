#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_cli_gpu.py -m gpu -q -k "reference" 2>&1 | tail -30 > gpurun_out/c3_tests_cli.log
timeout 600 python -m pytest tests/test_nets_gpu.py -m gpu -q -k "config0 or hoisting or single_step" 2>&1 | tail -25 > gpurun_out/c3_tests_nets.log
timeout 300 python -m pytest tests/test_sampler_kernels_gpu.py -m gpu -q 2>&1 | tail -5 > gpurun_out/c3_tests_samp.log
cp gpurun_out/parity_metrics.json gpurun_out/c3_parity_metrics.json 2>/dev/null
timeout 900 bash tools/pmc_gemm.sh g1 "lin64_320->320,geglu64,conv64_320->320,lin16_1280->1280" > gpurun_out/c3_pmc.log 2>&1
tail -4 gpurun_out/c3_tests_cli.log; tail -3 gpurun_out/c3_tests_nets.log; tail -2 gpurun_out/c3_tests_samp.log; cat gpurun_out/harness*_metrics.json; tail -5 gpurun_out/c3_pmc.log
