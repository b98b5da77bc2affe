#!/bin/bash
# scratch GPU session 4 (round 3): 160-row one-block-per-CU conv3q, up2 with 64x64 waves, in-pipeline effect of the 2x2 wave tiles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "tile2d" 2>&1 | tail -6 > gpurun_out/g4_kern.log
for nwb in 3 2; do
  MGLD_CONV3Q_NWB160=$nwb timeout 300 python tools/igemm_bench.py conv --rounds 3 --variants 0,5,8,10 --only conv64 > gpurun_out/g4_conv_nwb$nwb.log 2>&1
done
timeout 300 python tools/igemm_bench.py all --rounds 3 --variants 0,2,8 --only "up_" > gpurun_out/g4_up.log 2>&1
B="python bench.py --inflight 1 --steps 4 --warmup 2 --no-roofline --no-cpu-baseline"
for i in 1 2; do
  MGLD_W2=0 timeout 300 $B 2>/dev/null | tail -1 | cut -c1-170 | sed 's/^/v7        /' >> gpurun_out/g4_bench_ab.log
  MGLD_W2=0 MGLD_CONV3Q_V160=1 timeout 300 $B 2>/dev/null | tail -1 | cut -c1-170 | sed 's/^/v7+v160   /' >> gpurun_out/g4_bench_ab.log
  MGLD_W2=0 MGLD_CONV3Q_V160=1 MGLD_CONV3Q_UP2W64=1 timeout 300 $B 2>/dev/null | tail -1 | cut -c1-170 | sed 's/^/v7+v160+up/' >> gpurun_out/g4_bench_ab.log
done
cat gpurun_out/g4_kern.log; paste -d'\n' gpurun_out/g4_conv_nwb3.log gpurun_out/g4_conv_nwb2.log | grep -v amdgpu | cut -c1-200; grep -v amdgpu gpurun_out/g4_up.log | cut -c1-200; cat gpurun_out/g4_bench_ab.log
