#!/bin/bash
# round 3, evidence for the next round: full-N projection tiles (half the staged bytes per FLOP) with three segments in flight, alternating
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for rep in 1 2; do for f in 0 1; do
MGLD_IGEMM_FULLN=$f timeout 200 python bench.py --steps 9 --warmup 2 --no-one-at-a-time --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/g24_f${f}_$rep.json
python -c "import json;d=json.load(open('gpurun_out/g24_f${f}_$rep.json'));print('fulln $f rep $rep: three in flight',d['value'],d['ms_per_step'])"
done; done
