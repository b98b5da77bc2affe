#!/bin/bash
# round 3 final numbers after the attention block order: tests, PMC traffic (source key incl. attention.hip), driver-form bench, kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/final_tests.log
timeout 900 bash tools/pmc_traffic.sh > gpurun_out/final_pmc.log 2>&1
cp gpurun_out/pmc_traffic.json profiles/r03_pmc_traffic.json 2>/dev/null
timeout 900 python bench.py --steps 20 --warmup 5 --dump-shapes gpurun_out/final_igemm_shapes.json > gpurun_out/final_bench.log 2> gpurun_out/final_bench.err
tail -1 gpurun_out/final_bench.log > gpurun_out/final_bench.json
timeout 600 python bench.py --inflight 1 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/final_bench_inflight1.json
timeout 900 bash tools/prof_run.sh > gpurun_out/final_prof.log 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
tail -4 gpurun_out/final_tests.log; cut -c1-330 gpurun_out/final_bench.json; cut -c1-300 gpurun_out/final_bench_inflight1.json; head -8 gpurun_out/kernel_stats.txt | cut -c1-150; tail -3 gpurun_out/final_pmc.log | cut -c1-300
