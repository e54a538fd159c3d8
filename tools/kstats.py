#!/usr/bin/env python
"""Print a rocprofv3 *kernel_stats.csv (first argument: the file, or a directory searched for one): name, calls, total us, avg us."""
import csv
import glob
import os
import sys

path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True))[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(path)):
    if flt and flt not in r["Name"]:
        continue
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:60]:60s} {int(r['Calls']):5d} {float(r['TotalDurationNs']) / 1e3:12.1f} {float(r['AverageNs']) / 1e3:10.1f}")
