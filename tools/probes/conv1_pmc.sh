cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name, counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pmc_$n -o p --output-format csv -- python $R/tools/probes/conv1_pmc.py bits > /tmp/pmc_$n.log 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    if "conv_first_direct" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"{k:32s} per launch {sum(v) / len(v):16.0f}   ({len(v)} launches)")
PY
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
tail -3 /tmp/pmc_sq1.log
