#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "norm or spade" 2>&1 | tail -3
timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c14_bench.log 2>&1
tail -1 gpurun_out/c14_bench.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['ms_per_step'], r['traffic'], r['hbm']['kernels'])
"
