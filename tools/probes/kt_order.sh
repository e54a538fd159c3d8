#!/bin/bash
# the kernel trace of "python $R/$SCRIPT" in start order between the last two launches of the kernel matching $ANCHOR
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/kt_ord
rocprofv3 --kernel-trace -d /tmp/kt_ord -o p --output-format csv -- python $R/$SCRIPT > /tmp/kt_ord.log 2>&1
python - <<'PY'
import csv, glob, os
rows = []
for f in glob.glob("/tmp/kt_ord/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", ""))))
rows.sort()
anchor = os.environ.get("ANCHOR", "conv_first_direct")
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
a, b = idx[-2], idx[-1]
prev_end = rows[a][0]
for s, e, n, g in rows[a:b]:
    n = n.replace("void ", "", 1).replace("(anonymous namespace)::", "")
    print(f"{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  grid {g:>9}  {n[:110]}")
    prev_end = e
print(f"span {(rows[b][0] - rows[a][0]) / 1e3:.1f} us")
PY
