#!/bin/bash
# scratch GPU session 11 (round 3): static wave-slot priority in conv3q (MGLD_CONV3Q_PRIO) A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for rep in 1 2; do
  MGLD_CONV3Q_PRIO=0 timeout 300 python tools/igemm_bench.py conv --rounds 3 > gpurun_out/g11_conv_p0_$rep.log 2>&1
  MGLD_CONV3Q_PRIO=1 timeout 300 python tools/igemm_bench.py conv --rounds 3 > gpurun_out/g11_conv_p1_$rep.log 2>&1
done
MGLD_CONV3Q_PRIO=0 timeout 300 python tools/igemm_bench.py vae --rounds 2 > gpurun_out/g11_vae_p0.log 2>&1
MGLD_CONV3Q_PRIO=1 timeout 300 python tools/igemm_bench.py vae --rounds 2 > gpurun_out/g11_vae_p1.log 2>&1
B="python bench.py --inflight 1 --steps 4 --warmup 2 --no-roofline --no-cpu-baseline"
for i in 1 2; do
  MGLD_CONV3Q_PRIO=0 timeout 300 $B 2>/dev/null | tail -1 | cut -c1-170 | sed 's/^/prio0 /' >> gpurun_out/g11_bench_ab.log
  MGLD_CONV3Q_PRIO=1 timeout 300 $B 2>/dev/null | tail -1 | cut -c1-170 | sed 's/^/prio1 /' >> gpurun_out/g11_bench_ab.log
done
paste -d'\n' gpurun_out/g11_conv_p0_1.log gpurun_out/g11_conv_p1_1.log | grep -v amdgpu | cut -c1-110
paste -d'\n' gpurun_out/g11_conv_p0_2.log gpurun_out/g11_conv_p1_2.log | grep -v amdgpu | grep weighted
paste -d'\n' gpurun_out/g11_vae_p0.log gpurun_out/g11_vae_p1.log | grep -v amdgpu | cut -c1-110
cat gpurun_out/g11_bench_ab.log
