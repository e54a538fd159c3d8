#!/usr/bin/env python
"""Top kernels of a rocprofv3 --kernel-trace --stats run:  python tools/kstats.py <dir> [launch-divisor]."""
import csv, glob, os, sys
d, div = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
fs = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if not fs:
    sys.exit(f"no kernel_stats.csv under {d}: {os.listdir(d) if os.path.isdir(d) else 'missing'}")
rows = list(csv.DictReader(open(fs[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per pass {tot / div / 1e3:.1f} us over {len(rows)} kernels")
for r in rows[:28]:
    print(f"{r['Name'][:100]:100s} n {int(r['Calls']) / div:6.1f} avg {float(r['AverageNs']) / 1e3:8.1f} us "
          f"{float(r['TotalDurationNs']) / tot * 100:5.1f}%")
