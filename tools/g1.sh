#!/bin/bash
# scratch GPU session 1 (round 3): new kernels' unit tests, precision probe, workload parity, a short bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "w2 or igemm" 2>&1 | tail -6 > gpurun_out/g1_kern.log
timeout 300 python -m pytest tests/test_sampler_kernels_gpu.py -q -k "adain or colorfix" 2>&1 | tail -4 > gpurun_out/g1_adain.log
timeout 900 python tools/prec_probe.py c2 4 > gpurun_out/g1_probe.log 2>&1
timeout 600 python -m pytest tests/test_workloads_gpu.py -q 2>&1 | tail -12 > gpurun_out/g1_work.log
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/g1_bench.log 2> gpurun_out/g1_bench.err
cat gpurun_out/g1_kern.log gpurun_out/g1_adain.log; grep -v Warn gpurun_out/g1_probe.log | tail -6; cat gpurun_out/g1_work.log | tail -8; cut -c1-1500 gpurun_out/g1_bench.log; tail -3 gpurun_out/g1_bench.err
