#!/bin/bash
# round 3: XCD-grouped block order of the attention kernel: tests, segment A/B (one at a time and three in flight)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "flash" 2>&1 | tail -3 > gpurun_out/g19_tests.log
for rep in 1 2; do for x in 0 1; do
MGLD_ATTN_XCD=$x timeout 300 python bench.py --steps 9 --warmup 2 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/g19_x${x}_$rep.json
python -c "import json;d=json.load(open('gpurun_out/g19_x${x}_$rep.json'));print('attn xcd $x rep $rep: three in flight',d['value'],d['ms_per_step'],'one at a time',d['value_one_at_a_time'],d['one_at_a_time']['ms_per_step'])"
done; done
cat gpurun_out/g19_tests.log
