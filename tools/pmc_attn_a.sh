#!/bin/bash
# scratch: pass A of tools/pmc_attn.sh only (wave-cycle anatomy) for one variant: tools/pmc_attn_a.sh <tag>   (variant via MGLD_ATTN_SP)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-attn}
CMD="python $R/tools/attn_bench.py"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -o a -- $CMD > $R/gpurun_out/pmc_${TAG}_a.log 2>&1
cd $R
python tools/pmc_summary.py gpurun_out/pmc_${TAG}_a flash_attn > gpurun_out/pmc_${TAG}_a.txt 2>&1
find gpurun_out/pmc_${TAG}_a -name "*.csv" -delete 2>/dev/null
grep -A9 "grid=327680" gpurun_out/pmc_${TAG}_a.txt | head -12
