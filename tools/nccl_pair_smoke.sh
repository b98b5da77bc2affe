#!/bin/bash
# First thing to run on a lease with >= 2 GPUs (none was available to rounds 1-3: SCALE_r0x.json "skipped"): the RCCL paths of every
# multi-GPU mode, smallest first.  1) launcher + transport only (dummy tensors through the same DistComm calls), 2) segment-parallel
# bench (weak scaling, no data-path collective), 3) frame-sharded and 4) tile-sharded segments (halo send/recv + all-gathers over xGMI).
set -x
cd "$(dirname "$0")/.."
N=${1:-2}
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --gpus $N --backend nccl --spawn-selftest --frame-shard
python bench.py --gpus $N --backend nccl --steps 3 --warmup 1 --no-roofline --no-cpu-baseline
python bench.py --gpus $N --backend nccl --steps 3 --warmup 1 --no-roofline --no-cpu-baseline --frame-shard
python bench.py --gpus $N --backend nccl --steps 2 --warmup 1 --no-roofline --no-cpu-baseline --tile --tile-shard --size 1024 --frames 4 --guidance
