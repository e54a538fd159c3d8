"""Kernel time per step by category from a rocprofv3 kernel_stats.csv: python tools/probes/kcat.py <dir> <steps>"""
import csv, collections, glob, os, re, sys
path = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True))[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
def cat(n):
    if "naive_conv" in n or "batched_gemm_xdlops" in n: return "MIOpen naive (profiler artefact: the stem's first calls)"
    if "MIOpenBatchNorm" in n: return "MIOpen BatchNorm"
    if "igemm" in n or "kernel_groupe" in n or "Cijk" in n or "SubTensor" in n: return "MIOpen / rocBLAS (stem, classifier)"
    if "at::native" in n or "rocclr" in n: return "torch elementwise / reduce / copy / fill"
    if "mfma_gemm_kernel" in n: return "own: mfma_gemm (forward conv, grad_x)"
    if "wgrad_pm" in n or "pm_" in n: return "own: pixel-major weight gradient + packers"
    if any(k in n for k in ("pool_sum", "sqdev", "fold_kernel", "finalize", "act_bwd", "affine_codes", "bn_eval_device", "norm_sign", "bwd_")): return "own: training chain (BatchNorm / ReLU / quantiser)"
    return "own: packs, splits, scales, quantisers"
tot, calls = collections.Counter(), collections.Counter()
for r in rows:
    k = cat(r["Name"]); tot[k] += float(r["TotalDurationNs"]) / 1e3 / steps; calls[k] += int(r["Calls"]) / steps
for k, v in tot.most_common():
    print(f"{k:60s} {v:9.1f} us/step {calls[k]:7.1f} launches/step")
print("total (without the artefact)", round(sum(v for k, v in tot.items() if "artefact" not in k), 1), "us/step,",
      round(sum(v for k, v in calls.items() if "artefact" not in k), 1), "launches/step")
if len(sys.argv) > 3:
    for t, c, name in sorted(((float(r["TotalDurationNs"]) / 1e3 / steps, int(r["Calls"]) / steps, r["Name"]) for r in rows), reverse=True)[:int(sys.argv[3])]:
        short = re.sub(r"std::array<char\*, \d+ul?>", "", name)[:150]
        print(f"{t:9.1f} {c:6.1f} {short}")
