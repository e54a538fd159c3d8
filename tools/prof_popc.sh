#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 passes of the popcount (VALU) formulation of the C2 step: kernel stats, SQ issue counters,
# LDS counters, HBM-side traffic.  Output: gpurun_out/<tag>/...; condense with tools/prof_summary.py.
TAG=${1:-r2_popc}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
C2="python $R/bench.py --gemm valu --steps 20 --warmup 5 --no-cpu-baseline --alexnet-batch 0 --no-extras"
rocprofv3 --kernel-trace --stats -d "$O/kt" -o bench --output-format csv -- $C2 > "$O/kt.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d "$O/pmc_sq" -o bench --output-format csv -- $C2 > "$O/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU -d "$O/pmc_lds" -o bench --output-format csv -- $C2 > "$O/pmc_lds.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$O/pmc_fetch" -o bench --output-format csv -- $C2 > "$O/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$O/pmc_write" -o bench --output-format csv -- $C2 > "$O/pmc_write.log" 2>&1
for s in kt pmc_sq pmc_lds pmc_fetch pmc_write; do find "$O/$s" -mindepth 2 -name "*.csv" -exec mv {} "$O/$s/" \; ; done
rm -f "$O/kt/bench_kernel_trace.csv"
python - "$O" <<'PY'
import collections, csv, os, sys
O = sys.argv[1]
for sub in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write"):
    f = os.path.join(O, sub, "bench_counter_collection.csv")
    if not os.path.exists(f):
        print(sub, "missing"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "popc_gemm" in r["Kernel_Name"] or "pack_vec" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in agg.items():
        for c, v in dd.items():
            print(f"{sub:10s} {k:42s} {c:24s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
PY
head -8 "$O/kt/bench_kernel_stats.csv" | cut -c1-200
