#!/bin/bash
# scratch GPU session 12 (round 3): fused q|k|v + row-major-V attention: correctness, attention microbench, in-pipeline A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention" 2>&1 | tail -4 > gpurun_out/g12_kern.log
timeout 900 python -m pytest tests/test_nets_gpu.py -q -x -k "unet or structcond or sample_small or text" 2>&1 | tail -4 >> gpurun_out/g12_kern.log
B="python bench.py --inflight 1 --steps 4 --warmup 2 --no-roofline --no-cpu-baseline"
for i in 1 2; do
  MGLD_QKV_FUSED=0 timeout 300 $B 2>/dev/null | tail -1 | cut -c1-170 | sed 's/^/qkv0 /' >> gpurun_out/g12_bench_ab.log
  MGLD_QKV_FUSED=1 timeout 300 $B 2>/dev/null | tail -1 | cut -c1-170 | sed 's/^/qkv1 /' >> gpurun_out/g12_bench_ab.log
done
cat gpurun_out/g12_kern.log gpurun_out/g12_bench_ab.log
