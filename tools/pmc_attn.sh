#!/bin/bash
# PMC anatomy of the attention kernel (tools/attn_bench.py): where do the wave cycles go?   usage: tools/pmc_attn.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-attn}
CMD="python $R/tools/attn_bench.py"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -o a -- $CMD > $R/gpurun_out/pmc_${TAG}_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -o b -- $CMD > $R/gpurun_out/pmc_${TAG}_b.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_EXP_GDS SQ_INSTS_VALU_TRANS --output-format csv -d $R/gpurun_out/pmc_${TAG}_c -o c -- $CMD > $R/gpurun_out/pmc_${TAG}_c.log 2>&1
cd $R
for x in a b c; do python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$x flash_attn > gpurun_out/pmc_${TAG}_$x.txt 2>&1; done
find gpurun_out/pmc_${TAG}_a gpurun_out/pmc_${TAG}_b gpurun_out/pmc_${TAG}_c -name "*.csv" -delete 2>/dev/null
cat gpurun_out/pmc_${TAG}_a.txt gpurun_out/pmc_${TAG}_b.txt gpurun_out/pmc_${TAG}_c.txt | grep -A12 "grid=327680" | head -80
