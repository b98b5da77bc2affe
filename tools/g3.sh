#!/bin/bash
# scratch GPU session 3 (round 3): 2x2 wave-tile conv3q variants, sampler options, default precision scopes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "tile2d" 2>&1 | tail -6 > gpurun_out/g3_kern.log
timeout 300 python -m pytest tests/test_nets_gpu.py -q -x -k "options or sample_small" 2>&1 | tail -6 > gpurun_out/g3_opts.log
timeout 400 python tools/igemm_bench.py conv --rounds 3 --variants 0,5,8,9 --only conv64,conv32 > gpurun_out/g3_conv.log 2>&1
timeout 400 python tools/igemm_bench.py vae --rounds 2 --variants 0,5,8,9 --only vae64,vae128,vae256,vae512 > gpurun_out/g3_vae.log 2>&1
timeout 600 python tools/prec_probe.py c2 4 "vae_dec_mid,vae_dec_up2,vae_dec_up3,vae_dec_out,unet_io" > gpurun_out/g3_probe.log 2>&1
cat gpurun_out/g3_kern.log gpurun_out/g3_opts.log; grep -v amdgpu gpurun_out/g3_conv.log | cut -c1-200; grep -v amdgpu gpurun_out/g3_vae.log | cut -c1-200; grep "^[0v]" gpurun_out/g3_probe.log
