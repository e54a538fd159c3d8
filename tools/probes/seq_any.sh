#!/bin/bash
# The launches of ONE period (training step / graph replay) of "python $R/tools/probes/$SCRIPT" in start order: the tail of the
# kernel trace is searched for a period L (MINL..MAXL) repeated PERIODS times; median duration per position, grid, LDS.
#   SCRIPT=train_resnet_prof.py STEPS=8 PERIODS=6 MAXL=1600 bash tools/probes/seq_any.sh > gpurun_out/seq.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/seqany
rocprofv3 --kernel-trace -d /tmp/seqany -o p --output-format csv -- python $R/tools/probes/${SCRIPT} > /tmp/seqany.log 2>&1
tail -3 /tmp/seqany.log
python - <<'PY'
import csv, glob, os
def short(n):
    n = n.replace("void ", "", 1).replace("(anonymous namespace)::", "")
    i = n.find(">(")
    n = n[:i + 1] if i >= 0 else n.split("(")[0]
    return n[:110]
rows = []
for f in glob.glob("/tmp/seqany/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                     int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0),
                     int(r.get("LDS_Block_Size", 0) or 0)))
rows.sort()
n = int(os.environ.get("PERIODS", "5"))
names = [r[2] for r in rows]
best = None
for L in range(int(os.environ.get("MINL", "10")), int(os.environ.get("MAXL", "1600"))):
    if len(names) >= L * n and names[-L * n:] == names[-L:] * n:
        best = L
        break
if best is None:
    print("no period found; trace length", len(names))
    raise SystemExit
L = best
tail = rows[-L * n:]
tot = 0.0
span = sorted((tail[(k + 1) * L - 1][1] - tail[k * L][0]) / 1e3 for k in range(n))[n // 2]
for i in range(L):
    d = sorted((tail[k * L + i][1] - tail[k * L + i][0]) / 1e3 for k in range(n))[n // 2]
    gap = sorted((tail[k * L + i][0] - tail[k * L + i - 1][1]) / 1e3 for k in range(n))[n // 2] if i else 0.0
    r = tail[i]
    tot += d
    print(f"{i:4d} {d:7.1f} us (gap {gap:5.1f})  grid {r[3]:8d} wg {r[4]:4d} lds {r[5]:6d}  {r[2]}")
print(f"launches / period {L}; sum of kernel medians {tot:.1f} us; median period span {span:.1f} us")
PY
