#!/bin/bash
# the other BASELINE configs through bench.py (one line each into gpurun_out/): configs[2] = 4 frames + RAFT flows + guidance,
# the reference's segment length (5 frames + guidance), configs[3] = 4 frames 1024^2 aggregation sampling
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python bench.py --frames 4 --guidance --raft --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/bench_c3.json
timeout 600 python bench.py --frames 5 --guidance --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/bench_t5.json
timeout 900 python bench.py --frames 4 --size 1024 --tile --guidance --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/bench_c4.json
for f in bench_c3 bench_t5 bench_c4; do cut -c1-230 gpurun_out/$f.json; done
