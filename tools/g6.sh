#!/bin/bash
# scratch GPU session 6 (round 3): after templating the kernels on TWO: correctness, same-box A/B vs the round-2 library, NWB 3 vs 2, bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "igemm" 2>&1 | tail -4 > gpurun_out/g6_kern.log
timeout 300 python -m pytest tests/test_sampler_kernels_gpu.py -q -x 2>&1 | tail -3 >> gpurun_out/g6_kern.log
MGLD_HIP_LIB=$PWD/_variants/libmgld_r02.so timeout 300 python tools/igemm_bench.py conv --rounds 3 --variants 0 > gpurun_out/g6_conv_r02.log 2>&1
timeout 300 python tools/igemm_bench.py conv --rounds 3 --variants 0,5 > gpurun_out/g6_conv_cur.log 2>&1
MGLD_CONV3Q_NWB=3 timeout 300 python tools/igemm_bench.py conv --rounds 3 --variants 0,5 > gpurun_out/g6_conv_nwb3.log 2>&1
MGLD_HIP_LIB=$PWD/_variants/libmgld_r02.so timeout 300 python tools/igemm_bench.py lin --rounds 3 > gpurun_out/g6_lin_r02.log 2>&1
timeout 300 python tools/igemm_bench.py lin --rounds 3 > gpurun_out/g6_lin_cur.log 2>&1
MGLD_HIP_LIB=$PWD/_variants/libmgld_r02.so timeout 300 python tools/igemm_bench.py vae --rounds 2 > gpurun_out/g6_vae_r02.log 2>&1
timeout 300 python tools/igemm_bench.py vae --rounds 2 > gpurun_out/g6_vae_cur.log 2>&1
B="python bench.py --inflight 1 --steps 4 --warmup 2 --no-roofline --no-cpu-baseline"
for i in 1 2; do
  MGLD_W2=0 timeout 300 $B 2>/dev/null | tail -1 | cut -c1-170 | sed 's/^/w2off      /' >> gpurun_out/g6_bench_ab.log
  MGLD_W2=0 MGLD_CONV3Q_UP2W64=1 timeout 300 $B 2>/dev/null | tail -1 | cut -c1-170 | sed 's/^/w2off+up64 /' >> gpurun_out/g6_bench_ab.log
  timeout 300 $B 2>/dev/null | tail -1 | cut -c1-170 | sed 's/^/w2default  /' >> gpurun_out/g6_bench_ab.log
done
cat gpurun_out/g6_kern.log; for f in conv_r02 conv_cur conv_nwb3 lin_r02 lin_cur vae_r02 vae_cur; do echo "== $f"; grep -v amdgpu gpurun_out/g6_$f.log | cut -c1-150; done; cat gpurun_out/g6_bench_ab.log
