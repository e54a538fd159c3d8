#!/bin/bash
# The launches of ONE replay of the fused C4 hipGraph in start order: median duration per position over the replays, grid, LDS.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/c4s
REPLAYS=${REPLAYS:-40} rocprofv3 --kernel-trace -d /tmp/c4s -o p --output-format csv -- python $R/tools/probes/${SCRIPT:-c4_graph_kernels.py} > /tmp/c4s.log 2>&1
python - <<'PY'
import csv, glob, os
def short(n):
    n = n.replace("void ", "", 1).replace("(anonymous namespace)::", "")
    i = n.find(">(")
    n = n[:i + 1] if i >= 0 else n.split("(")[0]
    return n[:100]
rows = []
for f in glob.glob("/tmp/c4s/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                     int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0),
                     int(r.get("LDS_Block_Size", 0) or 0)))
rows.sort()
n = int(os.environ.get("REPLAYS", "40"))
# the replays are the tail of the trace: find the period as the launch count L with rows[-L*k:] repeating by name
names = [r[2] for r in rows]
best = None
for L in range(10, 200):
    if len(names) >= L * n and all(names[-L * n + i] == names[-L * (n - 1) + i] for i in range(L)) and names[-L * n:] == names[-L:] * n:
        best = L
        break
if best is None:
    print("no period found; last 120 launches:")
    for r in rows[-120:]:
        print(f"{(r[1]-r[0])/1e3:8.1f} us  grid {r[3]:8d} wg {r[4]:4d} lds {r[5]:6d}  {r[2]}")
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
    print(f"{i:3d} {d:7.1f} us (gap {gap:5.1f})  grid {r[3]:8d} wg {r[4]:4d} lds {r[5]:6d}  {r[2]}")
print(f"launches / replay {L}; sum of kernel medians {tot:.1f} us; median replay span {span:.1f} us")
PY
