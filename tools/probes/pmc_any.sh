#!/bin/bash
# counters of "python $R/$SCRIPT" for kernels matching $MATCH: kernel-trace pass + separate --pmc passes; medians per launch
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/$SCRIPT"
rm -rf /tmp/pa_*
rocprofv3 --kernel-trace -d /tmp/pa_kt -o p --output-format csv -- $CMD > /tmp/pa_kt.log 2>&1
run() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pa_$n -o p --output-format csv -- $CMD > /tmp/pa_$n.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq3 SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_INSTS_SMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
python - <<'PY'
import csv, glob, collections, os
match = os.environ.get("MATCH", "")
def short(n):
    n = n.replace("void ", "", 1).replace("(anonymous namespace)::", "")
    i = n.find(">("); n = n[:i + 1] if i >= 0 else n.split("(")[0]
    return n[:100]
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pa_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq1", "sq2", "sq3", "fetch", "write"):
    for f in glob.glob(f"/tmp/pa_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ctr[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in dur.items():
    if match not in k: continue
    v.sort()
    print(f"## {k}: launches {len(v)} median {v[len(v)//2]:.1f} us")
    for c, vals in sorted(ctr.get(k, {}).items()):
        vals.sort(); print(f"   {c:28s} {vals[len(vals)//2]:16.0f}")
PY
