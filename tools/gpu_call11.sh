#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/c11_tests.log
timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c11_bench.log 2>&1
tail -12 gpurun_out/c11_tests.log | cut -c1-300; tail -1 gpurun_out/c11_bench.log | cut -c1-300
