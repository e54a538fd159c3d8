#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for MODE in 0 1; do
rm -rf /tmp/qi$MODE
MODE=$MODE rocprofv3 --kernel-trace -d /tmp/qi$MODE -o p --output-format csv -- python $R/tools/probes/qi_kernels.py > /tmp/qi.log 2>&1
echo "== quant_input=$MODE"
python - $MODE <<'PY'
import csv, glob, collections, sys
dur = collections.defaultdict(list)
for f in glob.glob(f"/tmp/qi{sys.argv[1]}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "", 1).replace("(anonymous namespace)::", "")
        i = n.find(">("); n = n[:i + 1] if i >= 0 else n.split("(")[0]
        dur[n[:100]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 10:
        v.sort(); print(f"{len(v):4d} x median {v[len(v)//2]:8.1f} us  {k}")
PY
done
