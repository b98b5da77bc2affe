#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 400 python tools/stagger_sweep.py --staggers 0,2,4,6,8,13,20,26 --segments 12 --reps 2 2>&1 | grep -v Warning > gpurun_out/g17_sweep3.log
timeout 300 python tools/stagger_sweep.py --inflight 4 --staggers 0,4,8,13 --segments 12 --reps 2 2>&1 | grep -v Warning > gpurun_out/g17_sweep4.log
timeout 300 python tools/stagger_sweep.py --inflight 2 --staggers 0,4,8,13,20 --segments 12 --reps 2 2>&1 | grep -v Warning > gpurun_out/g17_sweep2.log
tail -20 gpurun_out/g17_sweep3.log; tail -10 gpurun_out/g17_sweep4.log; tail -12 gpurun_out/g17_sweep2.log
