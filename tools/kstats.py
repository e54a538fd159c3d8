#!/usr/bin/env python
"""Print a rocprofv3 *kernel_stats.csv (first argument: the file, or a directory searched for one), largest total first: name, calls,
total us, avg us.  Second argument: a substring filter, or --own for the kernels of libqt_hip.so only."""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
def own(n):          # this library's kernels live in anonymous namespaces of csrc/*.hip
    return n.startswith("void (anonymous namespace)::") or n.startswith("(anonymous namespace)::")


rows = [r for r in csv.DictReader(open(path)) if not flt or (own(r["Name"]) if flt == "--own" else flt in r["Name"])]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):          # largest total first
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:60]:60s} {int(r['Calls']):5d} {float(r['TotalDurationNs']) / 1e3:12.1f} {float(r['AverageNs']) / 1e3:10.1f}")
