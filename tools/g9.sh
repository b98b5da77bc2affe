#!/bin/bash
# scratch GPU session 9 (round 3): segments in flight 2 / 3 / 4 with the persistent-thread pool
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for k in 3 4 2 3 4; do
  timeout 600 python bench.py --inflight $k --steps 12 --warmup 2 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('inflight', d['config']['segments_in_flight'], 'value', d['value'], 'ms', d['ms_per_step'], 'lat', d['config']['segment_latency'], 'one', d.get('one_at_a_time'))" >> gpurun_out/g9_inflight.log
done
cat gpurun_out/g9_inflight.log
