#!/usr/bin/env python
"""C2 step: can pack (HBM-bound) and the fp4 GEMM (matrix-bound) run at the same time on disjoint CUs?

VERDICT r2 item 1.  Streams with CU masks (hipExtStreamCreateWithCUMask) are created through the HIP runtime
that torch already loaded and wrapped as torch ExternalStreams, so the un-modified ops.* wrappers launch on
them.  Measures, on one MI355X:

  E0  which CUs a mask bit names (one XCD vs spread over the XCDs), from the pack rate of 32-bit masks
  E1  pack-pair time / HBM rate as a function of the number of CUs in the mask
  E2  GEMM time for each tile configuration as a function of the number of CUs in the mask
  E3  pipelined C2 steps: GEMM(i) on one masked stream || pack(i+1) on the complementary mask, double-buffered
      nibble planes, events for the two dependencies; throughput per step + bit-exactness of every 16th result
  E4  two half-chip partitions each running whole steps (pack -> GEMM) in antiphase
  E5  the same pipelines without masks

Writes gpurun_out/cu_partition.json.  Nothing here is product code.
"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pytorch_quantize_impls_amd import ops, synth  # noqa: E402

_hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
_hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32,
                                              ctypes.POINTER(ctypes.c_uint32)]
_hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int
NCU = 256


def mask_words(bits):
    words = [0] * (NCU // 32)
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    return words


def masked_stream(bits, dev):
    """torch ExternalStream over a HIP stream restricted to the CUs named by ``bits`` (None = ordinary stream)."""
    if bits is None:
        return torch.cuda.Stream(device=dev)
    words = mask_words(bits)
    arr = (ctypes.c_uint32 * len(words))(*words)
    h = ctypes.c_void_p()
    rc = _hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), len(words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(h.value, device=dev)


def first(n):
    return list(range(n))


def spread(n_per_group, group=8):
    """n_per_group bits out of every ``group`` consecutive mask bits."""
    return [b for b in range(NCU) if (b % group) < n_per_group]


def timeit(stream, fn, iters=30, warm=5):
    with torch.cuda.stream(stream):
        for _ in range(warm):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    B = K = N = 4096
    x = torch.from_numpy(synth.pm1(1, (B, K))).to(dev)
    w = torch.from_numpy(synth.uniform(2, (N, K), -1.0, 1.0)).to(dev)
    out = {}
    pack_bytes = 2 * (B * K * 4 + B * K // 2)

    def pack():
        return ops.pack_linear_operands(x, w, "binary", "mfma")

    xp, wp = pack()
    y = torch.empty((B, N), device=dev)
    want = ops.nib_gemm(xp, wp, None).clone()
    torch.cuda.synchronize()

    # ---- E0 / E1: pack rate vs mask --------------------------------------------------------------------------------
    e1 = {}
    masks = {"first32": first(32), "bits_mod8_lt1 (32)": spread(1), "first64": first(64), "bits_mod8_lt2 (64)": spread(2),
             "bits_mod32_lt4 (32)": spread(4, 32), "first96": first(96), "first128": first(128),
             "bits_mod8_lt4 (128)": spread(4), "first192": first(192), "first224": first(224), "first256": first(256),
             "no mask": None}
    for name, bits in masks.items():
        s = masked_stream(bits, dev)
        t = timeit(s, pack)
        e1[name] = {"cus": None if bits is None else len(bits), "pack_us": round(t, 2), "TBps": round(pack_bytes / t / 1e6, 3)}
        print("pack", name, e1[name], flush=True)
    out["E1_pack_vs_mask"] = e1

    # ---- E2: GEMM vs mask ------------------------------------------------------------------------------------------
    e2 = {}
    variants = {"auto(PP256)": None, "PP128": 21, "PP192": 22, "PP384x192": 24, "PP64": 23}
    for mname, bits in (("no mask", None), ("first256", first(256)), ("first248", first(248)), ("first240", first(240)),
                        ("first224", first(224)), ("first192", first(192)), ("first128", first(128)),
                        ("bits_mod8_lt7 (224)", spread(7)), ("bits_mod8_lt4 (128)", spread(4))):
        s = masked_stream(bits, dev)
        row = {}
        for vname, v in variants.items():
            row[vname] = round(timeit(s, lambda: ops.nib_gemm(xp, wp, None, out=y, variant=v), iters=20), 2)
        e2[mname] = row
        print("gemm", mname, row, flush=True)
    out["E2_gemm_vs_mask"] = e2

    # ---- E3: GEMM(i) || pack(i+1) ------------------------------------------------------------------------------------
    def pipeline(gemm_bits, pack_bits, steps=200, variant=None, check=True):
        sg, sp = masked_stream(gemm_bits, dev), masked_stream(pack_bits, dev)
        ld = ops.packed_ld_nib(K)
        bufs = [(torch.empty((B, ld), dtype=torch.int32, device=dev), torch.empty((N, ld), dtype=torch.int32, device=dev))
                for _ in range(2)]
        ys = [torch.empty((B, N), device=dev) for _ in range(2)]
        x2, w2 = x, w
        I = int
        from pytorch_quantize_impls_amd import _lib

        def pack_into(buf, stream):
            _lib.call("qt_pack_pair_nib_f32", x2.data_ptr(), I(K), buf[0].data_ptr(), I(ld), I(B), w2.data_ptr(), I(K),
                      buf[1].data_ptr(), I(ld), I(N), I(K), 0, stream.cuda_stream)

        def gemm_from(buf, yo, stream):
            args = (buf[0].data_ptr(), I(ld), buf[1].data_ptr(), I(ld), None, yo.data_ptr(), I(N), I(B), I(N), I(K),
                    stream.cuda_stream)
            if variant is None:
                _lib.call("qt_nib_gemm", *args)
            else:
                _lib.call("qt_nib_gemm_variant", I(variant), *args)

        packed_ev = [torch.cuda.Event() for _ in range(2)]
        free_ev = [torch.cuda.Event() for _ in range(2)]
        bad = 0
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)

        def run(n, timed):
            nonlocal bad
            # prologue: pack(0)
            pack_into(bufs[0], sp)
            packed_ev[0].record(sp)
            if timed:
                sg.wait_event(packed_ev[0])
                t0.record(sg)
            for i in range(n):
                b = i & 1
                nb = b ^ 1
                # pack(i+1) into the other buffer once GEMM(i-1) has released it
                if i >= 1:
                    sp.wait_event(free_ev[nb])
                pack_into(bufs[nb], sp)
                packed_ev[nb].record(sp)
                sg.wait_event(packed_ev[b])
                gemm_from(bufs[b], ys[b], sg)
                free_ev[b].record(sg)
                if check and not timed and (i % 16 == 15):
                    sg.synchronize()
                    bad += int(not torch.equal(ys[b], want))
            if timed:
                t1.record(sg)
            torch.cuda.synchronize()

        run(40, False)
        run(steps, True)
        return {"us_per_step": round(t0.elapsed_time(t1) * 1e3 / steps, 2), "mismatching_results": bad}

    e3 = {}
    for name, g, p, v in (("no masks (2 streams)", None, None, None),
                          ("gemm first224 | pack last32", first(224), list(range(224, 256)), None),
                          ("gemm first224 PP128 | pack last32", first(224), list(range(224, 256)), 21),
                          ("gemm first192 PP128 | pack last64", first(192), list(range(192, 256)), 21),
                          ("gemm first192 PP192 | pack last64", first(192), list(range(192, 256)), 22),
                          ("gemm all256 | pack last32 (shared)", first(256), list(range(224, 256)), None),
                          ("gemm all256 | pack last64 (shared)", first(256), list(range(192, 256)), None),
                          ("gemm mod8<7 | pack mod8==7", spread(7), [b for b in range(NCU) if b % 8 == 7], None),
                          ("gemm mod8<7 PP128 | pack mod8==7", spread(7), [b for b in range(NCU) if b % 8 == 7], 21)):
        try:
            e3[name] = pipeline(g, p, variant=v)
        except Exception as e:  # noqa: BLE001
            e3[name] = {"error": repr(e)}
        print("E3", name, e3[name], flush=True)
    out["E3_gemm_i_par_pack_i1"] = e3

    # ---- E4: two partitions, whole steps in antiphase ------------------------------------------------------------------
    def antiphase(bits_a, bits_b, steps=200, variant=None):
        sa, sb = masked_stream(bits_a, dev), masked_stream(bits_b, dev)
        ya, yb = torch.empty((B, N), device=dev), torch.empty((B, N), device=dev)

        def step(s, yo):
            with torch.cuda.stream(s):
                a, b_ = ops.pack_linear_operands(x, w, "binary", "mfma")
                ops.nib_gemm(a, b_, None, out=yo, variant=variant)

        for _ in range(10):
            step(sa, ya)
            step(sb, yb)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream()
        t0.record(cur)
        sa.wait_event(t0)
        sb.wait_event(t0)
        for i in range(steps // 2):
            step(sa, ya)
            step(sb, yb)
        ea, eb = torch.cuda.Event(), torch.cuda.Event()
        ea.record(sa)
        eb.record(sb)
        cur.wait_event(ea)
        cur.wait_event(eb)
        t1.record(cur)
        torch.cuda.synchronize()
        ok = torch.equal(ya, want) and torch.equal(yb, want)
        return {"us_per_step": round(t0.elapsed_time(t1) * 1e3 / (steps // 2 * 2), 2), "bit_exact": bool(ok)}

    e4 = {}
    for name, a, b_, v in (("no masks, 2 streams", None, None, None),
                           ("first128 | last128", first(128), list(range(128, 256)), None),
                           ("first128 | last128 PP128", first(128), list(range(128, 256)), 21),
                           ("mod8<4 | mod8>=4", spread(4), [b for b in range(NCU) if b % 8 >= 4], None),
                           ("mod8<4 | mod8>=4 PP128", spread(4), [b for b in range(NCU) if b % 8 >= 4], 21),
                           ("even | odd", [b for b in range(NCU) if b % 2 == 0], [b for b in range(NCU) if b % 2], None)):
        try:
            e4[name] = antiphase(a, b_, variant=v)
        except Exception as e:  # noqa: BLE001
            e4[name] = {"error": repr(e)}
        print("E4", name, e4[name], flush=True)
    out["E4_antiphase_partitions"] = e4

    # baseline: one stream, back to back
    s = torch.cuda.Stream(device=dev)

    def whole():
        a, b_ = ops.pack_linear_operands(x, w, "binary", "mfma")
        ops.nib_gemm(a, b_, None, out=y)

    out["baseline_single_stream_us"] = round(timeit(s, whole, iters=200, warm=20), 2)
    print("baseline", out["baseline_single_stream_us"])
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/cu_partition.json", "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
