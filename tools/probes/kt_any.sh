#!/bin/bash
# rocprofv3 --kernel-trace of "python $R/$SCRIPT": count x median duration per kernel name (>= MIN launches)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/kt_any
rocprofv3 --kernel-trace -d /tmp/kt_any -o p --output-format csv -- python $R/$SCRIPT > /tmp/kt_any.log 2>&1
python - <<'PY'
import csv, glob, collections, os
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/kt_any/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "", 1).replace("(anonymous namespace)::", "")
        i = n.find(">("); n = n[:i + 1] if i >= 0 else n.split("(")[0]
        dur[(n[:90], r.get("Grid_Size_X", r.get("Grid_Size", "")) + "x" + r.get("Grid_Size_Y", "1"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= int(os.environ.get("MIN", "10")):
        v.sort(); print(f"{len(v):4d} x median {v[len(v)//2]:8.1f} us  min {v[0]:8.1f}  grid {k[1]:>14}  {k[0]}")
PY
