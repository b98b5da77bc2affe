#!/bin/bash
# round 3: do the three segments in flight run in lock-step?  stagger the workers' entry by a fraction of a UNet pass
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for rep in 1 2; do
for s in 0 4 13 40; do
MGLD_STAGGER_MS=$s timeout 300 python bench.py --steps 12 --warmup 3 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/g16_s${s}_$rep.json
python -c "import json;d=json.load(open('gpurun_out/g16_s${s}_$rep.json'));print('stagger $s rep $rep',d['value'],d['ms_per_step'],d.get('segment_latency_ms'))"
done
done
