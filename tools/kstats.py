#!/usr/bin/env python
"""Per-kernel totals from a rocprofv3 --kernel-trace CSV directory: calls, total ms, avg us, share."""
import collections
import csv
import glob
import os
import sys


def main():
    files = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)
    tot = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
            t = tot[r["Kernel_Name"]]
            t[0] += 1
            t[1] += d
    allms = sum(v[1] for v in tot.values()) or 1.0
    print(f"{'calls':>8} {'total_ms':>10} {'avg_us':>10} {'pct':>7}  name   (sum {allms:.1f} ms)")
    for name, (n, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:8d} {ms:10.2f} {1e3 * ms / n:10.2f} {100 * ms / allms:7.2f}  {name[:150]}")
    if len(sys.argv) > 2:      # machine-readable copy (bench.py reads profiles/r04_kernel_stats.json for roofline.frac_rocprof_avg)
        import json
        with open(sys.argv[2], "w") as fh:
            json.dump({"what": "rocprofv3 --kernel-trace: per kernel, calls / total ms / average us over the profiled bench.py run",
                       "kernels": {name: {"calls": n, "total_ms": round(ms, 3), "avg_us": round(1e3 * ms / n, 3)} for name, (n, ms) in tot.items()}},
                      fh, indent=0)


if __name__ == "__main__":
    main()
