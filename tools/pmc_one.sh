#!/bin/bash
# scratch: wave-cycle anatomy of one attention shape: tools/pmc_one.sh <tag> B H N   (variant via MGLD_ATTN_* env)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
CMD="python $R/tools/attn_one.py $@"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -o a -- $CMD > $R/gpurun_out/pmc_${TAG}_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -o b -- $CMD > $R/gpurun_out/pmc_${TAG}_b.log 2>&1
cd $R
for x in a b; do python tools/pmc_summary.py gpurun_out/pmc_${TAG}_$x flash_attn > gpurun_out/pmc_${TAG}_$x.txt 2>&1; find gpurun_out/pmc_${TAG}_$x -name "*.csv" -delete 2>/dev/null; done
cat gpurun_out/pmc_${TAG}_a.txt gpurun_out/pmc_${TAG}_b.txt
