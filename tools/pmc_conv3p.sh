# HBM counters of the 3x3 conv on three UNet shapes, im2col kernel (MGLD_CONV3P=0) vs patch kernel; summary -> gpurun_out/pmc_conv3p.txt
# FETCH_SIZE and WRITE_SIZE need SEPARATE passes (one pass with both: "Request exceeds the capabilities of the hardware to
# collect", then rocprofv3 hangs in its abort handler) — every pass runs under its own timeout.  PMC_L2=1 adds the L2 request pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MGLD_BENCH_ONLY=0,6,9 MGLD_BENCH_ITERS=4
T=${PMC_PASS_TIMEOUT:-25}
for c in 0 1; do
  export MGLD_CONV3P=$c
  timeout -k 3 $T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcF$c -o f -- python $R/tools/igemm_bench.py > $R/gpurun_out/pmcF$c.log 2>&1
  timeout -k 3 $T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcW$c -o w -- python $R/tools/igemm_bench.py > $R/gpurun_out/pmcW$c.log 2>&1
  if [ -n "$PMC_L2" ]; then
    timeout -k 3 $T rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmcL$c -o l -- python $R/tools/igemm_bench.py > $R/gpurun_out/pmcL$c.log 2>&1
  fi
done
cd $R
: > gpurun_out/pmc_conv3p.txt
for c in 0 1; do for d in pmcF$c pmcW$c pmcL$c; do [ -d gpurun_out/$d ] || continue; echo "== MGLD_CONV3P=$c $d" >> gpurun_out/pmc_conv3p.txt; python tools/pmc_summary.py gpurun_out/$d _kernel >> gpurun_out/pmc_conv3p.txt 2>&1; done; done
find gpurun_out/pmcF0 gpurun_out/pmcF1 gpurun_out/pmcW0 gpurun_out/pmcW1 gpurun_out/pmcL0 gpurun_out/pmcL1 -name "*.csv" -delete 2>/dev/null
cat gpurun_out/pmc_conv3p.txt
