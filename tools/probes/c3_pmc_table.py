#!/usr/bin/env python
"""profiles/r6_c3_pmc.md from gpurun_out/c3_pmc/summary.txt (tools/probes/c3_pmc.sh): the once-per-forward kernels of the un-modified
BinaryNet-AlexNet forward (batch 256) with their algorithmic bytes beside the counter bytes."""
import re, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/c3_pmc/summary.txt"
blocks, cur = [], None
for ln in open(src):
    m = re.match(r"## (.*?)  grid (\d+)", ln)
    if m:
        cur = {"name": m.group(1), "grid": int(m.group(2)), "c": {}}
        blocks.append(cur)
        continue
    if cur is None:
        continue
    m = re.match(r"\s+launches (\d+)\s+median ([\d.]+) us", ln)
    if m:
        cur["launches"], cur["us"] = int(m.group(1)), float(m.group(2))
        continue
    m = re.match(r"\s+([A-Za-z_0-9]+)\s+(\d+)", ln)
    if m:
        cur["c"][m.group(1)] = float(m.group(2))
B = 256
MB = 1e6
# algorithmic bytes of the fused inference chain: operand planes in, epilogue planes out
ALG = {
    ("conv_first_direct", 131072): ("conv1 3->192 k11 s4 (fp32 image in, threshold bits out)", B * 3 * 224 * 224 * 4 + 192 * 363 * 2 * 2, B * 55 * 55 * 192 / 8),
    ("ElemFp4, 4, 2, 3, 3", 749568): ("conv2 192->576 k5 @27 (nibble halo plane in, bits out)", B * 31 * 31 * 192 / 2 + 576 * 4800 / 2, B * 27 * 27 * 576 / 8),
    ("ElemFp4, 4, 2, 2, 3", 552960): ("conv3 576->1152 k3 @13 (nibbles in, conv4's nibble halo plane out)", B * 15 * 15 * 576 / 2 + 1152 * 5184 / 2, B * 15 * 15 * 1152 / 2),
    ("ElemFp4, 2, 4, 4, 2", 294912): ("conv4 1152->768 k3 @13 (nibbles in, conv5's nibble halo plane out)", B * 15 * 15 * 1152 / 2 + 768 * 10368 / 2, B * 15 * 15 * 768 / 2),
    ("ElemFp4, 2, 4, 4, 2", 90112): ("conv5 768->256 k3 @13 (nibbles in, bits out)", B * 15 * 15 * 768 / 2 + 256 * 6912 / 2, B * 13 * 13 * 256 / 8),
    ("ElemFp4, 2, 2, 1, 1", 65536): ("fc 9216->4096 / 4096->4096 (skinny fp4 GEMM, fp32 out; 2 per forward)", (B * 9216 / 2 + 4096 * 9216 / 2 + B * 4096 / 2 + 4096 * 4096 / 2) / 2, B * 4096 * 4),
}
CLK = 1.8e9      # the clock the chip holds under matrix load (docs/KERNEL_NOTES.md)
print("| kernel (grid) | layer | median us | MFMA duty | VALU / MFMA instr | wave cycles waiting | FETCH x2 MB | alg. read MB | WRITE MB | alg. write MB | L2 hit |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for b in blocks:
    if b.get("launches", 0) > 24 or "us" not in b:
        continue
    key = next((k for k in ALG if k[0] in b["name"] and k[1] == b["grid"]), None)
    if key is None:
        continue
    what, rd, wr = ALG[key]
    c = b["c"]
    duty = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / (b["us"] * 1e-6 * CLK) if c else float("nan")
    ratio = c.get("SQ_INSTS_VALU", 0) / max(1.0, c.get("SQ_INSTS_MFMA", 1))
    wait = c.get("SQ_WAIT_ANY", 0) / max(1.0, c.get("SQ_WAVE_CYCLES", 1))
    hit = c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0))
    name = re.sub(r"mfma_gemm_kernel<GemmCfg<", "mfma_gemm<", b["name"])[:48]
    print(f"| `{name}` ({b['grid']}) | {what} | {b['us']:.1f} | {duty:.2f} | {ratio:.1f} | {wait:.2f} | {2 * c.get('FETCH_SIZE', 0) * 1024 / MB:.1f} | {rd / MB:.1f} | "
          f"{c.get('WRITE_SIZE', 0) * 1024 / MB:.1f} | {wr / MB:.1f} | {hit:.2f} |")
