#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/axg
rocprofv3 --kernel-trace -d /tmp/axg -o p --output-format csv -- python $R/tools/probes/alexnet_graph_kernels.py > /tmp/axg.log 2>&1
python - <<'PY'
import csv, glob, collections
def short(n):
    n = n.replace("void ", "", 1).replace("(anonymous namespace)::", "")
    i = n.find(">(")
    n = n[:i + 1] if i >= 0 else n.split("(")[0]
    return n[:110]
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/axg/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = [(k, len(v), sorted(v)[len(v)//2]) for k, v in dur.items() if len(v) >= 50]
tot = sum(n / 50.0 * med for _, n, med in rows)
print(f"per replay (kernels with >= 50 launches): {tot:.0f} us")
for k, n, med in sorted(rows, key=lambda r: -r[1] * r[2]):
    print(f"{n/50.0:6.2f} / replay  median {med:7.1f} us  = {n/50.0*med:7.1f} us   {k}")
PY
