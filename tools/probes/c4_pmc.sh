#!/bin/bash
# Run ON THE GPU BOX: counters of the int8 conv kernels of the FUSED C4 forward (DoReFa ResNet-18 W1A4, batch 256).
#   kernel-trace pass (durations per kernel name) + separate --pmc passes (no other trace domain), averaged per launch and kernel name.
# Output: gpurun_out/c4_pmc/summary.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/c4_pmc
mkdir -p $O
CMD="python $R/tools/prof_c4.py"
export FUSED=1 FOLD=device ITERS=${ITERS:-12}
rocprofv3 --kernel-trace --stats -d /tmp/c4_kt -o p --output-format csv -- $CMD > /tmp/c4_kt.log 2>&1
run() { n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/c4_$n -o p --output-format csv -- $CMD > /tmp/c4_$n.log 2>&1
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_MFMA
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python - > $O/summary.txt <<'PY'
import csv, glob, collections, re
def short(n):
    n = n.replace("void ", "", 1).replace("(anonymous namespace)::", "")
    i = n.find(">(")
    n = n[:i + 1] if i >= 0 else n.split("(")[0]
    return n[:120]
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/c4_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq1", "sq2", "fetch", "write", "tcc"):
    for f in glob.glob(f"/tmp/c4_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            ctr[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = sum(sum(v) for v in dur.values())
print(f"# fused C4 forward, {len(next(iter(dur.values()), []))}.. launches per kernel over the run; total kernel time {tot:.0f} us")
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    v = dur[k]
    print(f"\n## {k}\n   launches {len(v)}  avg {sum(v)/len(v):.1f} us  total {sum(v):.0f} us ({100*sum(v)/tot:.1f} %)")
    for c, vals in sorted(ctr.get(k, {}).items()):
        print(f"   {c:28s} {sum(vals)/len(vals):16.0f}")
PY
grep -A 22 'mfma_gemm_kernel\|^#' $O/summary.txt | head -c 14000
