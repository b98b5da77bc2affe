#!/bin/bash
# scratch: a further SQ pass (co-execution, latencies, fifo stalls, instruction fetch) for one variant: tools/pmc_attn_d.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-attn}
CMD="python $R/tools/attn_bench.py"
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${TAG}_d -o d -- $CMD > $R/gpurun_out/pmc_${TAG}_d.log 2>&1
rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU2 SQ_INSTS_MFMA SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_${TAG}_e -o e -- $CMD > $R/gpurun_out/pmc_${TAG}_e.log 2>&1
cd $R
for x in d e; do python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$x flash_attn > gpurun_out/pmc_${TAG}_$x.txt 2>&1; find gpurun_out/pmc_${TAG}_$x -name "*.csv" -delete 2>/dev/null; done
grep -A10 "sp_kernel.*grid=327680" gpurun_out/pmc_${TAG}_d.txt gpurun_out/pmc_${TAG}_e.txt | head -30
tail -3 gpurun_out/pmc_${TAG}_d.log
