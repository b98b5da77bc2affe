# kernel-trace profile with THREE segments in flight (6 timed segments, two per worker, no warm-up and no one-at-a-time leg: every launch of the trace ran with three segments in flight): per-kernel average durations under
# co-running, to set beside the one-at-a-time profile (tools/prof_run.sh).  Summary -> gpurun_out/kernel_stats_inflight3.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof3 && mkdir -p $R/gpurun_out/prof3
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof3 -o p -- python $R/bench.py --inflight 3 --steps 6 --warmup 0 --no-one-at-a-time --no-cpu-baseline --no-roofline > $R/gpurun_out/bench_prof3.log 2>&1
cd $R
python tools/kstats.py gpurun_out/prof3 > gpurun_out/kernel_stats_inflight3.txt 2>&1
find gpurun_out/prof3 -name "*.csv" -size +2M -delete
tail -1 gpurun_out/bench_prof3.log | cut -c1-200
head -14 gpurun_out/kernel_stats_inflight3.txt | cut -c1-160
