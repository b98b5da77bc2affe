cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MGLD_BENCH_ONLY=0,12,5 MGLD_BENCH_ITERS=4
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmcA -o a -- python $R/tools/igemm_bench.py > $R/gpurun_out/pmcA.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $R/gpurun_out/pmcB -o b -- python $R/tools/igemm_bench.py > $R/gpurun_out/pmcB.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmcC -o c -- python $R/tools/igemm_bench.py > $R/gpurun_out/pmcC.log 2>&1
cd $R
for d in pmcA pmcB pmcC; do python tools/pmc_summary.py gpurun_out/$d igemm_kernel > gpurun_out/$d.txt 2>&1; done
find gpurun_out/pmcA gpurun_out/pmcB gpurun_out/pmcC -name "*.csv" -delete
cat gpurun_out/pmcA.txt gpurun_out/pmcB.txt gpurun_out/pmcC.txt
