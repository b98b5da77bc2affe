#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/c4_tests.log
cp gpurun_out/parity_metrics.json gpurun_out/c4_parity_metrics.json 2>/dev/null
timeout 300 python tools/igemm_bench.py lin --rounds 3 --only geglu > gpurun_out/c4_geglu.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --dump-shapes gpurun_out/c4_shapes.json > gpurun_out/c4_bench.log 2>&1
bash tools/prof_run.sh > gpurun_out/c4_prof.log 2>&1
cp gpurun_out/kernel_stats.txt gpurun_out/c4_kernel_stats.txt
tail -6 gpurun_out/c4_tests.log; grep -v amdgpu gpurun_out/c4_geglu.log | tail -4; tail -1 gpurun_out/c4_bench.log | cut -c1-250; head -25 gpurun_out/c4_kernel_stats.txt | cut -c1-160
