#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "attention or attn" 2>&1 | tail -4
timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c13_bench_pipe.log 2>&1
MGLD_ATTN_PIPE=0 timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c13_bench_nopipe.log 2>&1
for f in c13_bench_pipe c13_bench_nopipe; do tail -1 gpurun_out/$f.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], [ (v['kernel'], v['ms_per_segment'], v['tflops']) for v in r['by_kernel'] if 'attn' in v['kernel']])
"; done
