# HBM traffic of the dominant kernel (roofline.traffic): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in SEPARATE passes (kernel
# trace only) over one eager segment of the default bench workload.  Writes gpurun_out/pmc_traffic.json; copy to profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MGLD_SC_PRECOMPUTE=0   # rocprofv3 --pmc segfaults on the batched struct-cond passes; per-launch traffic of the kernel is unaffected
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-graph > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python - <<'PY'
import collections, csv, glob, json
def per_kernel(counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"gpurun_out/pmc_{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = r["Kernel_Name"]
            tot[key][0] += 1
            tot[key][1] += float(r["Counter_Value"])
    return tot
fe, wr = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
out = {}
for fam, match in (("igemm_kernel<128,128>", ", 128, 128, 64, 64, 2>"), ("igemm_kernel<128,64>", ", 128, 64, 64, 32, 2>"),
                   ("igemm_kernel<64,128>", ", 64, 128, 32, 64, 2>"), ("flash_attn_kernel<64>", "flash_attn_kernel<64>")):
    n = sum(v[0] for k, v in fe.items() if match in k)
    f = sum(v[1] for k, v in fe.items() if match in k)
    w = sum(v[1] for k, v in wr.items() if match in k)
    if n:
        out[fam] = {"launches": n, "fetch_size_kb_per_launch": f / n, "write_size_kb_per_launch": w / n,
                    "hbm_bytes_per_launch": (2.0 * f + w) / n * 1024.0}
dom = out.get("igemm_kernel<128,128>", {})
res = {"kernel": "igemm_kernel<128,128> (all gather modes)", **dom, "families": out,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0 --no-graph` "
               "(one 8x512^2 50-step segment, eager launches); counters in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md "
               "(gfx950 reports 1/2 of wide coalesced reads); WRITE_SIZE uncalibrated; Infinity-Cache hits are counted"}
json.dump(res, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items()}))
PY
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -delete
