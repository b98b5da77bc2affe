#!/bin/bash
# scratch GPU session 10 (round 3): PMC anatomy of the dominant conv kernel and of flash attention
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 bash tools/pmc_gemm.sh conv "conv64_320->320,vae256" > gpurun_out/g10_pmc_conv.log 2>&1
timeout 600 bash tools/pmc_attn.sh > gpurun_out/g10_pmc_attn.log 2>&1
grep -v "^\s*$" gpurun_out/g10_pmc_conv.log | cut -c1-160 | head -120
tail -40 gpurun_out/g10_pmc_attn.log | cut -c1-160
