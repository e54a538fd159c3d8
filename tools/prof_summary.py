#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel-trace --stats, and separate --pmc passes) into the text
summary committed under profiles/.

    python tools/prof_summary.py gpurun_out/r1 > profiles/r1_bench_c2_rocprof.md

Expects <dir>/kt (kernel stats), and optionally <dir>/pmc_fetch, pmc_write, pmc_l2.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports HALF the bytes of a wide coalesced
read stream (MI355X_MICROARCH.md, HBM section) so the corrected column doubles it; WRITE_SIZE is
uncorrected (uncalibrated)."""
import collections
import csv
import os
import sys

d = sys.argv[1]
ours = ("mfma_gemm_kernel", "pool_affine_sign_pack_kernel", "triple_kernel", "codes_kernel", "im2col_words_kernel",
        "col_abs_mean_kernel", "sign_scale_kernel", "nib_gemm_kernel", "nib_pack_vec_kernel", "nib_pack_pair_kernel", "bits_to_nib_pad_kernel", "s2d_triple_rows_kernel", "nib_pack_scalar_kernel", "popc_gemm_kernel",
        "pack_vec_kernel", "pack_wave_kernel", "bits_to_nib_kernel", "unary_kernel", "binary_kernel",
        "check_pm1_kernel", "pair_kernel", "s2d_pair_rows_kernel", "absmax_part_kernel", "absmax_final_kernel", "pool_sum_kernel",
        "sqdev_kernel", "fold_kernel", "finalize_kernel", "norm_sign_kernel", "bwd_sum_kernel", "bwd_dx_kernel", "pool_bwd_kernel", "wgrad_pack", "wgrad_reduce_kernel", "pm_zero_kernel", "pm_bias_reduce_kernel", "pm_pack_act", "pm_pack_grad", "wgrad_pm_kernel", "wgrad_pm_full_kernel", "pm_reduce_kernel", "bn_eval_device_kernel", "absmax_ch_final_kernel", "absmax_ch_strided_kernel", "absmax_ch_rows_kernel", "fold2_kernel", "act_bwd_dx_kernel", "act_bwd_sum_kernel", "pool_bits_kernel", "affine_codes_kernel", "pool_codes_kernel", "zero_halo_kernel", "pad_pixel_plane_kernel", "s2d_triple_kernel", "popc_skinny_kernel", "popc_stream_kernel", "conv", "im2col")


import re


def short(n):
    for k in ours:
        if k in n:
            tail = ""
            m = re.search(r"GemmCfg<\(anonymous namespace\)::(\w+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\w+)(?:, (\d+))? ?>", n)
            if m:
                occ = f", {m.group(10)} wg/CU" if m.group(10) and m.group(10) != "1" else ""
                tail = (f"<{m.group(1)}, tile {int(m.group(2))*int(m.group(4))*32}x{int(m.group(3))*int(m.group(5))*32}, "
                        f"pipe={m.group(6)}{', conv' if m.group(9) in ('true', '1') else (', conv-valid' if m.group(9) == '2' else '')}{occ}>")
            elif "<" in n and k == "popc_gemm_kernel":
                tail = n[n.index("<"):n.index(">") + 1][:40]
            return k + tail
    return n.split("(")[0][-60:]


print(f"# rocprofv3 summary of `{d}`\n")
ks = os.path.join(d, "kt", "bench_kernel_stats.csv")
if os.path.exists(ks):
    print("## kernel-trace --stats (top kernels)\n")
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---|---|---|---|")
    rows = list(csv.DictReader(open(ks)))

    def row(r):
        print(f"| {short(r['Name'])} | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | "
              f"{float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.1f} |")
    for r in rows[:16]:
        row(r)
    # the headline step's own kernels and the training kernels, wherever they rank (the library's find-mode kernels of the
    # reference-op legs crowd the top of the list)
    must = ("nib_pack_pair_kernel", "wgrad_pm", "pm_pack", "pm_reduce")
    print("\n## the C2 step's kernels and the weight-gradient kernels (same run)\n")
    print("| kernel | calls | total us | avg us | % |")
    print("|---|---|---|---|---|")
    for r in rows:
        n = r["Name"]
        s_ = short(n)
        is_c2_gemm = "mfma_gemm_kernel" in n and "ElemFp4" in n and s_.endswith("pipe=2>")
        if is_c2_gemm or any(m in n for m in must):
            if not (is_c2_gemm or "nib_pack_pair" in n):
                s_ = n.replace("(anonymous namespace)::", "").replace("void ", "")
                s_ = s_[:s_.index("(")] if "(" in s_ else s_
            print(f"| {s_[:70]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.1f} |")
print("\n## PMC (separate passes, per-dispatch averages)\n")
print("| kernel | counter | dispatches | avg value | as bytes |")
print("|---|---|---|---|---|")
for sub in ("pmc_fetch", "pmc_write", "pmc_l2"):
    f = os.path.join(d, sub, "bench_counter_collection.csv")
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, dd in sorted(agg.items()):
        if not any(o in k for o in ours):
            continue
        for c, v in dd.items():
            avg = sum(v) / len(v)
            note = ""
            if c == "FETCH_SIZE":
                note = f"{avg*1024/1e6:.1f} MB raw, {2*avg*1024/1e6:.1f} MB corrected (x2, gfx950)"
            elif c == "WRITE_SIZE":
                note = f"{avg*1024/1e6:.1f} MB (uncalibrated)"
            print(f"| {k} | {c} | {len(v)} | {avg:.1f} | {note} |")
