#!/usr/bin/env python
"""Condense rocprofv3 --pmc counter_collection CSVs: per kernel (name truncated), dispatch count and the per-dispatch mean of
every counter.  Usage: pmc_summary.py <dir-or-csv> [name-filter]"""
import collections
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if flt and flt not in name:
                continue
            key = (name[:110], r["Grid_Size"], r["LDS_Block_Size"])
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[key].add(r["Dispatch_Id"])
    for key, ctr in sorted(agg.items()):
        n = max(1, len(disp[key]))
        print(f"{key[0]}  grid={key[1]} lds={key[2]} dispatches={n}")
        for c, v in sorted(ctr.items()):
            print(f"    {c:36s} {v / n:16.1f}")


if __name__ == "__main__":
    main()
