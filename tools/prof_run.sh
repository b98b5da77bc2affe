# kernel-trace profile of the default bench workload (1 warmup + 1 timed segment); summary -> gpurun_out/kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof && mkdir -p $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o p -- python $R/bench.py --clips 2 --inflight 1 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/bench_prof.log 2>&1
cd $R
python tools/kstats.py gpurun_out/prof gpurun_out/kernel_stats.json > gpurun_out/kernel_stats.txt 2>&1
find gpurun_out/prof -name "*.csv" -size +2M -delete
tail -1 gpurun_out/bench_prof.log | cut -c1-200
head -40 gpurun_out/kernel_stats.txt
