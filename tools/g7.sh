#!/bin/bash
# scratch GPU session 7 (round 3): full GPU suite, up2 64x64-wave A/B in isolation, default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/g7_tests.log
timeout 300 python tools/igemm_bench.py all --rounds 3 --variants 0,2,8 --only "up_" > gpurun_out/g7_up.log 2>&1
timeout 900 python bench.py --steps 8 --warmup 3 > gpurun_out/g7_bench.log 2> gpurun_out/g7_bench.err
cat gpurun_out/g7_tests.log; grep -v amdgpu gpurun_out/g7_up.log | cut -c1-170; cut -c1-2500 gpurun_out/g7_bench.log; tail -3 gpurun_out/g7_bench.err
cat gpurun_out/parity_metrics.json | tr -d '\n' | cut -c1-3000
