#!/bin/bash
# Run HERE after `gpurun -- 'bash tools/collect_profiles.sh <tag>'` (+ the four training-step traces, see below) merged its output
# into gpurun_out/: condenses the raw rocprofv3 CSVs into the summaries committed under profiles/.
#   training-step traces (on the GPU box, cwd /tmp, TMPDIR=/tmp):
#     [FUSED=1] rocprofv3 --kernel-trace --stats -d gpurun_out/tr_{rn,ax}_{mod,fused} -o t --output-format csv -- python tools/probes/train_{resnet,chain}_prof.py
TAG=${1:-r3}
R=$(cd "$(dirname "$0")/.." && pwd)
cd "$R"
python tools/prof_summary.py gpurun_out/$TAG > profiles/${TAG}_bench_rocprof.md
cp gpurun_out/$TAG/bench_line.json profiles/${TAG}_bench_line.json
cp gpurun_out/$TAG/kt/bench_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
for c in c4 c5; do
  { echo "# rocprofv3 --kernel-trace --stats of the $c line of bench.py's extra object (tools/collect_profiles.sh $TAG): name, calls, total us, avg us"
    echo "# (the command times the fused form, the un-modified module graph, the module-by-module eager graph and the reference's op sequence on the GPU:"
    echo "#  MIOpen / torch kernels belong to the last two)"
    echo; echo "## kernels of libqt_hip.so"; echo '```'; python tools/kstats.py gpurun_out/$TAG/kt_$c --own | head -45; echo '```'
    echo; echo "## all kernels"; echo '```'; python tools/kstats.py gpurun_out/$TAG/kt_$c | head -40; echo '```'; } > profiles/${TAG}_${c}_rocprof.md
done
if [ -d gpurun_out/tr_rn_mod ]; then
  { echo "# Training steps, batch 256, per-kernel-class split ($TAG)"; echo
    echo "\`rocprofv3 --kernel-trace --stats -- python tools/probes/train_resnet_prof.py\` (DoReFa ResNet-18 W1A4, 3x32x32, 8 steps; \`FUSED=1\`: bench_models.TrainFusedDorefaResNet18)"
    echo "and \`tools/probes/train_chain_prof.py\` (BinaryNet-AlexNet, 3x224x224, 6 steps; \`FUSED=1\`: TrainFusedAlexNetBin), condensed by \`tools/probes/kcat.py <dir> <steps> <top>\`:"
    echo "kernel time per step by class, then the top kernels (us per step, launches per step).  Under the profiler the fp32 stem conv of the"
    echo "ResNet (torch / MIOpen, not this path's) runs MIOpen's naive fallback kernels in its first calls; that class is listed and excluded from the totals."
    echo "Since round 4 the un-modified module graph reaches the fused training nodes by itself (lazy_train.py): 'module graph' below IS the fused chain"
    echo "(no MIOpen BatchNorm / torch pooling kernel between the layers; the ResNet's fp32 stem BatchNorm is the one MIOpen BatchNorm left)."
    for n in rn_mod:8:"ResNet-18, un-modified module graph" rn_fused:8:"ResNet-18, explicit TrainFused form" ax_mod:6:"AlexNet-Bin, un-modified module graph" ax_fused:6:"AlexNet-Bin, explicit TrainFused form"; do
      d=${n%%:*}; r=${n#*:}; st=${r%%:*}; t=${r#*:}
      [ -d gpurun_out/tr_$d ] || continue
      echo; echo "## $t"; echo '```'; python tools/probes/kcat.py gpurun_out/tr_$d $st 24; echo '```'
    done; } > profiles/${TAG}_train_steps_rocprof.md
fi
grep -n "tile 256x256, pipe=2> |" profiles/${TAG}_bench_rocprof.md | head -3
# profiles/pmc_latest.json: what bench.py quotes as roofline.traffic / rocprof_avg_us (counters cannot be read from inside the
# process): the C2 GEMM's and the pack kernel's per-launch averages of THIS collection
python - "$TAG" <<'PY'
import json, re, sys
tag = sys.argv[1]
md = open(f"profiles/{tag}_bench_rocprof.md").read()
cur = json.load(open("profiles/pmc_latest.json"))
def pmc(kernel_pat, counter):
    m = re.search(r"\| " + kernel_pat + r"[^|]*\| " + counter + r" \| \d+ \| ([0-9.]+) \|", md)
    return float(m.group(1)) if m else None
def avg(kernel_pat):
    m = re.search(r"\| " + kernel_pat + r"[^|]*\| \d+ \| [0-9.]+ \| ([0-9.]+) \|", md)
    return float(m.group(1)) if m else None
g = {"FETCH_SIZE_KiB": pmc(r"mfma_gemm_kernel<ElemFp4, tile 256x256, pipe=2>", "FETCH_SIZE"),
     "WRITE_SIZE_KiB": pmc(r"mfma_gemm_kernel<ElemFp4, tile 256x256, pipe=2>", "WRITE_SIZE"),
     "rocprof_avg_us": avg(r"mfma_gemm_kernel<ElemFp4, tile 256x256, pipe=2>")}
p = {"FETCH_SIZE_KiB": pmc(r"nib_pack_pair_kernel", "FETCH_SIZE"), "WRITE_SIZE_KiB": pmc(r"nib_pack_pair_kernel", "WRITE_SIZE"),
     "rocprof_avg_us": avg(r"nib_pack_pair_kernel")}
if all(v is not None for v in g.values()):
    cur["kernels"]["mfma_gemm_kernel"] = g
    if all(v is not None for v in p.values()):
        cur["kernels"]["nib_pack_pair_kernel"] = p
    cur["source"] = (f"profiles/{tag}_bench_rocprof.md (rocprofv3 --kernel-trace --stats of the default bench command; --kernel-trace --pmc FETCH_SIZE / "
                     f"--pmc WRITE_SIZE passes of `python bench.py --steps 40 --warmup 10 --no-cpu-baseline --alexnet-batch 0 --no-extras`, "
                     f"tools/collect_profiles.sh {tag})")
    json.dump(cur, open("profiles/pmc_latest.json", "w"))
    print("pmc_latest.json <-", tag, g)
else:
    print("pmc_latest.json unchanged: counters not found in", tag, g)
PY
