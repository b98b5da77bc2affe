cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmcA -o a -- python $R/tools/attn_bench.py > $R/gpurun_out/pmcA.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcC -o c -- python $R/tools/attn_bench.py > $R/gpurun_out/pmcC.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_LEVEL_WAVES SQ_INSTS_MFMA SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/pmcD -o d -- python $R/tools/attn_bench.py > $R/gpurun_out/pmcD.log 2>&1
cd $R
for d in pmcA pmcC pmcD; do python tools/pmc_summary.py gpurun_out/$d flash_attn > gpurun_out/$d.txt 2>&1; done
find gpurun_out/pmcA gpurun_out/pmcC gpurun_out/pmcD -name "*.csv" -delete
grep -A10 "grid=40960 " gpurun_out/pmcA.txt | head -12; grep -A11 "grid=40960 " gpurun_out/pmcC.txt | head -12; grep -A10 "grid=40960 " gpurun_out/pmcD.txt | head -12
