#!/bin/bash
# scratch GPU session 5 (round 3): same-box A/B of the round-2 kernel library (_variants/libmgld_r02.so) against the current one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for rep in 1 2; do
  MGLD_HIP_LIB=$PWD/_variants/libmgld_r02.so timeout 300 python tools/igemm_bench.py conv --rounds 3 --variants 0 > gpurun_out/g5_conv_r02_$rep.log 2>&1
  timeout 300 python tools/igemm_bench.py conv --rounds 3 --variants 0,5 > gpurun_out/g5_conv_cur_$rep.log 2>&1
done
MGLD_HIP_LIB=$PWD/_variants/libmgld_r02.so timeout 300 python tools/igemm_bench.py lin --rounds 3 > gpurun_out/g5_lin_r02.log 2>&1
timeout 300 python tools/igemm_bench.py lin --rounds 3 > gpurun_out/g5_lin_cur.log 2>&1
MGLD_HIP_LIB=$PWD/_variants/libmgld_r02.so timeout 300 python tools/igemm_bench.py vae --rounds 2 > gpurun_out/g5_vae_r02.log 2>&1
timeout 300 python tools/igemm_bench.py vae --rounds 2 > gpurun_out/g5_vae_cur.log 2>&1
for f in conv_r02_1 conv_cur_1 conv_r02_2 conv_cur_2 lin_r02 lin_cur vae_r02 vae_cur; do echo "== $f"; grep -v amdgpu gpurun_out/g5_$f.log | cut -c1-150; done
