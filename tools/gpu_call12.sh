#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
for r in 1 2; do
timeout 300 python tools/igemm_bench.py lin --nst 0 --rounds 4 > gpurun_out/c12_lin_pk_$r.log 2>&1
MGLD_HIP_LIB=$GRAFT_REPO_ROOT/_variants/libmgld_scalar.so timeout 300 python tools/igemm_bench.py lin --nst 0 --rounds 4 > gpurun_out/c12_lin_sc_$r.log 2>&1
done
timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c12_bench_pk.log 2>&1
MGLD_HIP_LIB=$GRAFT_REPO_ROOT/_variants/libmgld_scalar.so timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c12_bench_sc.log 2>&1
timeout 300 python bench.py --steps 8 --warmup 3 > gpurun_out/c12_bench_pk2.log 2>&1
timeout 600 python -m pytest tests/test_cli_gpu.py -m gpu -q -k "json_line" 2>&1 | tail -3
for f in c12_lin_pk_1 c12_lin_sc_1 c12_lin_pk_2 c12_lin_sc_2; do echo $f; grep -v amdgpu gpurun_out/$f.log | cut -c1-120 | tail -12; done
for f in c12_bench_pk c12_bench_sc c12_bench_pk2; do tail -1 gpurun_out/$f.log | cut -c1-220; done
