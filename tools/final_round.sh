# the round's closing measurement batch on one box: full GPU suite, kernel-trace profile, PMC traffic, the default bench line, the other configs
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3
bash tools/prof_run.sh > gpurun_out/prof_run.log 2>&1
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
tail -3 gpurun_out/pmc_traffic.log
python bench.py > gpurun_out/bench_default.log 2>&1
tail -1 gpurun_out/bench_default.log | cut -c1-300
bash tools/bench_configs.sh
MGLD_STREAM_LO=0 MGLD_LN_FOLD=0 python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 2>/dev/null | tail -1 | cut -c1-200 > gpurun_out/bench_round5_path.json
python bench.py --no-cpu-baseline --no-roofline --steps 6 --warmup 2 2>/dev/null | tail -1 | cut -c1-200 > gpurun_out/bench_quick.json
cat gpurun_out/bench_round5_path.json gpurun_out/bench_quick.json
