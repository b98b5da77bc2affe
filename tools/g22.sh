#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
t0=$(date +%s.%N)
timeout 400 python bench.py --steps 9 --warmup 2 --no-roofline --no-cpu-baseline 2> gpurun_out/g22_bench.err | tail -1 > gpurun_out/g22_bench.json
t1=$(date +%s.%N)
python -c "import json;d=json.load(open('gpurun_out/g22_bench.json'));print('three in flight',d['value'],d['ms_per_step'],'one at a time',d['value_one_at_a_time'], 'wall', round($t1-$t0,1),'s')" || tail -5 gpurun_out/g22_bench.err
