#!/bin/bash
# round 3: TCONV frame-interleaved rows + skinny-output K split: tests, microbench, same-box A/B of the segment
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "tconv or skinny or w2_conv3x3 or igemm_conv" 2>&1 | tail -5 > gpurun_out/g15_tests.log
timeout 300 python tools/tconv_bench.py > gpurun_out/g15_tconv.log 2>&1
MGLD_TCONV_ROWS=0 MGLD_IGEMM_SKINNY_SPLIT=0 timeout 400 python bench.py --inflight 1 --steps 8 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/g15_bench_off.json
timeout 400 python bench.py --inflight 1 --steps 8 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/g15_bench_on.json
MGLD_TCONV_ROWS=0 MGLD_IGEMM_SKINNY_SPLIT=0 timeout 400 python bench.py --inflight 1 --steps 8 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/g15_bench_off2.json
timeout 400 python bench.py --inflight 1 --steps 8 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/g15_bench_on2.json
cat gpurun_out/g15_tests.log; cat gpurun_out/g15_tconv.log | cut -c1-260
for f in off on off2 on2; do python -c "import json;d=json.load(open('gpurun_out/g15_bench_$f.json'));print('$f',d['value'],d['ms_per_step'])"; done
