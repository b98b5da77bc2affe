#!/bin/bash
# round 3: kernel variants that lose launch by launch, re-judged with three segments in flight (what matters there is resource-time, not latency);
# and configs[2] (RAFT inside the segment) with segments in flight
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
run() { # name, env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --steps 9 --warmup 2 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/g21_$n.json
  python -c "import json;d=json.load(open('gpurun_out/g21_$n.json'));print('$n: three in flight',d['value'],d['ms_per_step'],'one at a time',d['value_one_at_a_time'],d['one_at_a_time']['ms_per_step'])"
}
run base A=1
run w64 MGLD_CONV3Q_W64=1
run nwb3 MGLD_CONV3Q_NWB=3
run fulln MGLD_IGEMM_FULLN=1
run base2 A=1
timeout 400 python bench.py --frames 4 --guidance --raft --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>gpurun_out/g21_c3.err | tail -1 > gpurun_out/g21_c3.json
python -c "import json;d=json.load(open('gpurun_out/g21_c3.json'));print('c3:',d['value'],d['ms_per_step'],d.get('value_one_at_a_time'),d['config']['segments_in_flight'])" || tail -5 gpurun_out/g21_c3.err
