# HBM traffic per launch of every GEMM-family / attention kernel (roofline.traffic): rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in
# SEPARATE passes (counter collection only, no trace domains) over one eager segment of the default bench workload.
# Writes gpurun_out/pmc_traffic.json keyed by the exact kernel instantiation name + the sha256 of the GEMM family's sources
# (bench.py: GEMM_FAMILY_SOURCES) it was taken on (bench.py uses an entry only when both match the running build);
# copy it to profiles/r06_pmc_traffic.json.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MGLD_HP_ENCODER=0      # (the fp32 / split passes of the first-stage encoder launch 2M-block grids: keep the counter hook off them)
export MGLD_SC_PRECOMPUTE=0   # rocprofv3 --pmc segfaults inside its launch hook on the batched struct-cond passes (round 1); per-launch
                              # traffic of a kernel instantiation is a property of the kernel, not of where the encoder runs
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o pmc -- python $R/bench.py --clips 2 --inflight 1 --steps 1 --warmup 0 --no-roofline --no-cpu-baseline --no-graph > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python - <<'PY'
import collections, csv, glob, hashlib, json, re
def per_kernel(counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"gpurun_out/pmc_{counter}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"((?:igemm|conv3p|conv3q|conv3r|ppgemm|pptconv|flash_attn(?:_sp2?)?|splitk_reduce|gn_\w+|layernorm)_kernel(?:<[^>]*>)?)", r["Kernel_Name"])
            if not m:
                continue
            tot[m.group(1)][0] += 1
            tot[m.group(1)][1] += float(r["Counter_Value"])
    return tot
fe, wr = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
kern = {}
for k, (n, f) in fe.items():
    w = wr.get(k, [0, 0.0])[1]
    kern[k] = {"launches": n, "fetch_size_kb_per_launch": f / n, "write_size_kb_per_launch": w / n,
               "hbm_bytes_per_launch": (2.0 * f + w) / n * 1024.0}
import sys; sys.path.insert(0, ".")
from bench import GEMM_FAMILY_SOURCES
sha = hashlib.sha256(b"".join(open("mgld_vsr_amd/csrc/" + f, "rb").read() for f in GEMM_FAMILY_SOURCES)).hexdigest()[:16]
res = {"gemm_src_sha16": sha, "kernels": kern,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `MGLD_SC_PRECOMPUTE=0 bench.py --steps 1 --warmup 0 --no-graph` "
               "(two 8x512^2 50-step segments batched as clips of one pass — the default scheduling's launch shapes — eager launches); counters in KB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE "
               "doubled as MI355X_MICROARCH.md prescribes for gfx950 (it reports 1/2 of wide coalesced reads); WRITE_SIZE uncalibrated; "
               "Infinity-Cache hits are counted (memory-side requests of the L2s)"}
json.dump(res, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in sorted(kern.items(), key=lambda kv: -kv[1]["launches"])[:12]}))
PY
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name "*.csv" -delete
