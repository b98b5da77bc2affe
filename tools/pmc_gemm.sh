#!/bin/bash
# PMC passes over a few GEMM-family problems (tools/igemm_bench.py --only ...): where do the cycles of the short-K projections and of the
# 64x64-level convolution go?   usage: tools/pmc_gemm.sh <tag> "<--only list>"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; ONLY=$2
CMD="python $R/tools/igemm_bench.py all --only $ONLY --rounds 1 --iters 3"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -o a -- $CMD > $R/gpurun_out/pmc_${TAG}_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -o b -- $CMD > $R/gpurun_out/pmc_${TAG}_b.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum --output-format csv -d $R/gpurun_out/pmc_${TAG}_c -o c -- $CMD > $R/gpurun_out/pmc_${TAG}_c.log 2>&1
cd $R
for x in a b c; do python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$x _kernel > gpurun_out/pmc_${TAG}_$x.txt 2>&1; done
find gpurun_out/pmc_${TAG}_a gpurun_out/pmc_${TAG}_b gpurun_out/pmc_${TAG}_c -name "*.csv" -delete 2>/dev/null
cat gpurun_out/pmc_${TAG}_a.txt gpurun_out/pmc_${TAG}_b.txt gpurun_out/pmc_${TAG}_c.txt | head -150
