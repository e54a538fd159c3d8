#!/bin/bash
# kernel-trace only (durations per kernel name) of the fused C4 forward; gpurun_out/c4_kt.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export FUSED=${FUSED:-1} FOLD=device ITERS=${ITERS:-12}
rm -rf /tmp/c4_kt
rocprofv3 --kernel-trace -d /tmp/c4_kt -o p --output-format csv -- python $R/tools/prof_c4.py > /tmp/c4_kt.log 2>&1
python - > $R/gpurun_out/c4_kt.txt <<'PY'
import csv, glob, collections
def short(n):
    n = n.replace("void ", "", 1).replace("(anonymous namespace)::", "")
    i = n.find(">(")
    n = n[:i + 1] if i >= 0 else n.split("(")[0]
    return n[:100]
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/c4_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in dur.values())
print(f"total kernel time {tot:.0f} us")
for k in sorted(dur, key=lambda k: -sum(dur[k]))[:24]:
    v = sorted(dur[k])
    print(f"{len(v):5d} x  median {v[len(v)//2]:7.1f} us  total {sum(v):8.0f} us  {k}")
PY
cat $R/gpurun_out/c4_kt.txt
