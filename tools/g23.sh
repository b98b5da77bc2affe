#!/bin/bash
# round 3, last session: full GPU suite (with the 50-step 1024^2 workload), profile with three segments in flight, driver-form bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/final_tests.log
timeout 600 bash tools/prof_run_inflight.sh > gpurun_out/final_prof3.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --dump-shapes gpurun_out/final_igemm_shapes.json > gpurun_out/final_bench.log 2> gpurun_out/final_bench.err
tail -1 gpurun_out/final_bench.log > gpurun_out/final_bench.json
tail -4 gpurun_out/final_tests.log; cut -c1-330 gpurun_out/final_bench.json; head -12 gpurun_out/kernel_stats_inflight3.txt | cut -c1-150
