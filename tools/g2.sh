#!/bin/bash
# scratch GPU session 2 (round 3): three-buffer conv3q ring A/B, fine-grained precision scopes, c2g workload
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv3x3 or w2" 2>&1 | tail -6 > gpurun_out/g2_kern.log
for nwb in 3 2; do
  MGLD_CONV3Q_NWB=$nwb timeout 300 python tools/igemm_bench.py conv --rounds 3 > gpurun_out/g2_conv_nwb$nwb.log 2>&1
  MGLD_CONV3Q_NWB=$nwb timeout 300 python tools/igemm_bench.py vae --rounds 2 > gpurun_out/g2_vae_nwb$nwb.log 2>&1
done
for nwb in 3 2 3 2; do
  MGLD_W2=0 MGLD_CONV3Q_NWB=$nwb timeout 300 python bench.py --inflight 1 --steps 4 --warmup 2 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/g2_bench_ab.log
done
timeout 900 python tools/prec_probe.py c2 4 "0;vae_dec_out;vae_dec_up0;vae_dec_up1;vae_dec_up2,vae_dec_up3,vae_dec_mid;vae_dec_fuse" > gpurun_out/g2_probe.log 2>&1
timeout 600 python -m pytest tests/test_workloads_gpu.py -q 2>&1 | tail -12 > gpurun_out/g2_work.log
cat gpurun_out/g2_kern.log; paste -d'\n' gpurun_out/g2_conv_nwb3.log gpurun_out/g2_conv_nwb2.log | cut -c1-130; paste -d'\n' gpurun_out/g2_vae_nwb3.log gpurun_out/g2_vae_nwb2.log | cut -c1-130
cat gpurun_out/g2_bench_ab.log; grep "^[0v]" gpurun_out/g2_probe.log; tail -5 gpurun_out/g2_work.log
