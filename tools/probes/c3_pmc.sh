#!/bin/bash
# Run ON THE GPU BOX: counters of every kernel of the un-modified BinaryNet-AlexNet forward (batch 256).
#   kernel-trace pass (durations) + separate --pmc passes (no other trace domain), averaged per launch over (kernel name, grid).
#   FETCH_SIZE is doubled (gfx950 correction, MI355X_MICROARCH.md "HBM"); WRITE_SIZE as reported.
# Output: gpurun_out/c3_pmc/summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c3_pmc
mkdir -p $O
CMD="python $R/tools/probes/alexnet_eager_kernels.py"
export ITERS=${ITERS:-12}
rocprofv3 --kernel-trace --stats -d /tmp/c3_kt -o p --output-format csv -- $CMD > /tmp/c3_kt.log 2>&1
run() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/c3_$n -o p --output-format csv -- $CMD > /tmp/c3_$n.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_MFMA
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python - > $O/summary.txt <<'PY'
import csv, glob, collections
def short(n):
    n = n.replace("void ", "", 1).replace("(anonymous namespace)::", "")
    i = n.find(">(")
    n = n[:i + 1] if i >= 0 else n.split("(")[0]
    return n[:120]
def key(r):
    return (short(r["Kernel_Name"]), str(r.get("Grid_Size_X", r.get("Grid_Size", ""))))
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/c3_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[key(r)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq1", "sq2", "fetch", "write", "tcc"):
    for f in glob.glob(f"/tmp/c3_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ctr[key(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = sum(sum(v) for v in dur.values())
print(f"# un-modified AlexNet-Bin forward, batch 256; total kernel time over the run {tot:.0f} us")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    v = sorted(dur[k])
    print(f"\n## {k[0]}  grid {k[1]}\n   launches {len(v)}  median {v[len(v)//2]:.1f} us  total {sum(v):.0f} us ({100*sum(v)/tot:.1f} %)")
    for c, vals in sorted(ctr.get(k, {}).items()):
        vals = sorted(vals)
        m = vals[len(vals)//2]
        extra = f"   x2 = {2*m/1e6:9.2f} MB" if c == "FETCH_SIZE" else (f"      = {m/1e6:9.2f} MB" if c == "WRITE_SIZE" else "")
        print(f"   {c:28s} {m:16.0f}{extra}")
PY
head -c 6000 $O/summary.txt
