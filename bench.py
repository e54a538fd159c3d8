#!/usr/bin/env python
"""bench.py — headline benchmark of the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (config.workload = "c2"): BASELINE.json configs[1] — LinearBin 4096x4096 XNOR-popcount
GEMM, batch 4096 PER GPU (weak scaling: the op is batch-sharded, no collective on the data path).
One "step" = one training-mode LinearBin forward on +-1 activations with fp32 inputs already
resident in HBM:  sign+bit-pack(x)  ->  sign+bit-pack(W)  ->  packed GEMM  ->  y (fp32), all through
the C-ABI of libqt_hip.so.  value = whole-job tera-ops/s, ops = 2*B*K*N per GPU per step.

For N > 1 the driver launches one rank per GPU with torch.distributed.run; ranks only meet in the
barriers that bracket the timed region and in the MAX reduction of the elapsed time.

Extra objects on the JSON line: "roofline" (dominant kernel, HIP-event timed inside the timed
region) and "cpu_baseline" (reference op sequence on the host cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_TOPS = 2516.6        # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz x 32 MAC / 2 lane-ops x 2 ops
MFMA_FP4_PEAK_TFLOPS = 10000.0  # dense MXFP4 (MI355X_MICROARCH.md)
EVENT_EVERY = 5                 # GEMM launches bracketed by HIP events: every 5th timed step


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="rows per GPU")
    ap.add_argument("--in-features", type=int, default=4096)
    ap.add_argument("--out-features", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=12.0)
    ap.add_argument("--gemm", default="auto", choices=["auto", "valu", "mfma"],
                    help="packed GEMM formulation (auto = fastest available for the shape)")
    ap.add_argument("--alexnet-batch", type=int, default=256, help="images per GPU (0 = skip)")
    ap.add_argument("--alexnet-iters", type=int, default=10)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (smoke tests)")
    ap.add_argument("--share-device", action="store_true",
                    help="smoke test only: every rank uses cuda:0 (exercise the N>1 code path on a 1-GPU box)")
    return ap.parse_args()


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit(f"--gpus {args.gpus} needs a torch.distributed.run launch with "
                     f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=dev)
        else:
            dist_mod.init_process_group(args.dist_backend)
        dist = dist_mod

    from pytorch_quantize_impls_amd import _lib, ops
    from pytorch_quantize_impls_amd.functions import _fused

    B, K, N = args.batch, args.in_features, args.out_features
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x5EED + rank)
    # synthetic data of the config's shape: +-1 activations (fp32), W ~ N(0, 1/K), bias = 0
    x = torch.randn((B, K), device=dev, generator=gen).sign_()
    x[x == 0] = 1
    w = torch.randn((N, K), device=dev, generator=gen) * (1.0 / K) ** 0.5
    y = torch.empty((B, N), device=dev, dtype=torch.float32)

    gemm_impl = ops.select_gemm_impl(args.gemm, B, N, K)

    ev_pairs = []

    def step(record=False):
        xp, wp = ops.pack_linear_operands(x, w, "binary", gemm_impl)   # both operands, one launch on the mfma route
        if record:
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.packed_gemm(xp, wp, None, out=y, impl=gemm_impl)
            e1.record()
            ev_pairs.append((e0, e1))
        else:
            ops.packed_gemm(xp, wp, None, out=y, impl=gemm_impl)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    # HIP events bracket the GEMM on every EVENT_EVERY-th timed step only: a recorded pair costs ~7.6 us of
    # stream time per step (tools/bench_step_overheads.py), which would otherwise be billed to `value`.
    for i in range(args.steps):
        step(record=(i % EVENT_EVERY == 0))
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    ops_per_step = 2.0 * B * K * N
    value = world * ops_per_step * args.steps / elapsed / 1e12
    gemm_ms = sum(a.elapsed_time(b) for a, b in ev_pairs) / len(ev_pairs)

    # ---- roofline of the dominant kernel (the packed GEMM) ------------------------------------
    gemm_bytes = ops.packed_gemm_algorithmic_bytes(B, N, K, gemm_impl)  # DESIGN.md "Kernels"
    if gemm_impl == "mfma":
        achieved = ops_per_step / (gemm_ms * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": "fp4 MFMA packed GEMM", "achieved": achieved,
                    "peak": MFMA_FP4_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_FP4_PEAK_TFLOPS,
                    "traffic": None, "kernel_ms": gemm_ms,
                    "kernel_ms_note": f"HIP-event bracket around the launch on every {EVENT_EVERY}th step of the timed region; "
                                      "includes the marker packets / kernel boundary (~3-5 us): rocprofv3 kernel "
                                      "duration is in profiles/",
                    "hbm_equiv": {"algorithmic_bytes": gemm_bytes,
                                  "achieved_GBs": gemm_bytes / (gemm_ms * 1e-3) / 1e9,
                                  "frac_of_8TBs": gemm_bytes / (gemm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    else:
        achieved = gemm_bytes / (gemm_ms * 1e-3) / 1e9
        valu = ops_per_step / (gemm_ms * 1e-3) / 1e12
        roofline = {"bound": "hbm", "kernel": "xnor popcount GEMM (VALU)", "achieved": achieved,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "kernel_ms": gemm_ms, "algorithmic_bytes": gemm_bytes,
                    "note": "at this shape the popcount formulation is VALU-bound, not HBM-bound "
                            "(SURVEY.md 8d); the instruction ceiling is reported beside it",
                    "valu_ceiling": {"achieved_TOPS": valu, "peak_TOPS": VALU_PEAK_TOPS,
                                     "frac": valu / VALU_PEAK_TOPS}}
    step_bytes = 4.0 * (B * K + N * K + B * N) + 4.0 * N  # fp32 in / fp32 out, SURVEY.md 8d
    step_ms = elapsed / args.steps * 1e3
    roofline["step_hbm"] = {"algorithmic_bytes": step_bytes,
                            "achieved_GBs": step_bytes / (step_ms * 1e-3) / 1e9,
                            "frac_of_8TBs": step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    roofline["traffic"], roofline["traffic_source"] = pmc_traffic(gemm_impl)

    result = {
        "metric": "XNOR-popcount GEMM TOPS (LinearBin 4096x4096 forward, batch 4096 per GPU)",
        "value": value, "unit": "TOPS", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32 bit planes (xor+popcount), int32 accumulate, fp32 in/out"
                 if gemm_impl != "mfma" else "fp4-e2m1 (+-1 exact) MFMA, fp32 accumulate, fp32 in/out",
        "data": "synthetic",
        "config": {"workload": "c2: LinearBin train-mode forward = sign+pack(x) + sign+pack(W) + packed GEMM",
                   "batch_per_gpu": B, "in_features": K, "out_features": N, "global_batch": B * world,
                   "parallelism": f"batch-shard x{world}, no collective", "gemm_impl": gemm_impl},
        "roofline": roofline,
    }

    # ---- BinaryNet-AlexNet images/s (second half of BASELINE.json's metric) ---------------------------
    if args.alexnet_batch > 0:
        result["alexnet"] = bench_alexnet(args, dev, dist, world, rank)

    # ---- parity gate + CPU baseline (rank 0, N = 1 only) --------------------------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_port
        xc, wc = x.cpu(), w.cpu()
        ref = torch_port.linear_bin_forward(xc, wc)
        result["parity_vs_cpu_port"] = bool(torch.equal(ref, y.cpu()))
        # the host has far more logical CPUs than MKL scales to: use the fastest thread count
        nthreads, _ = torch_port.best_thread_count(lambda: torch_port.linear_bin_forward(xc, wc))
        med, iters = torch_port.time_callable(lambda: torch_port.linear_bin_forward(xc, wc),
                                              budget_s=args.cpu_budget_s)
        result["cpu_baseline"] = {
            "value": ops_per_step / med / 1e12, "unit": "TOPS", "cores": torch.get_num_threads(),
            "kind": "port", "ms_per_step": med * 1e3,
            "sample": f"{iters} full-size calls of the reference op sequence "
                      f"(torch.sign + masked write + F.linear fp32, B={B} K={K} N={N}), median",
            "host": _cpu_model()}
    if rank == 0:
        result["calls"] = {k: int(v) for k, v in _lib.call_counts.items()}
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


def bench_alexnet(args, dev, dist, world, rank):
    """BinaryNet-AlexNet (models/Alexnet/Alexnet_Bin.py topology) eval-mode
    forward, 3x224x224, batch per GPU = --alexnet-batch, channels_last, weights pre-quantised and
    pre-packed by .eval() (the reference's eval protocol)."""
    import bench_models
    B = args.alexnet_batch
    torch.manual_seed(1234 + rank)
    model = bench_models.AlexNetBin()
    bench_models.randomize_bn(model)
    model = model.to(dev).to(memory_format=torch.channels_last).eval()
    x = torch.randn((B, 3, 224, 224), device=dev).contiguous(memory_format=torch.channels_last)
    def timed(fn):
        """`alexnet_iters` forwards, barrier + sync on both sides, MAX over ranks; best of 3 such runs (a
        one-off host stall of tens of ms otherwise dominates a 5-forward sample)."""
        best, last = None, None
        with torch.no_grad():
            for _ in range(3):
                last = fn(x)
            for _ in range(3):
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                t0 = time.perf_counter()
                for _ in range(args.alexnet_iters):
                    last = fn(x)
                torch.cuda.synchronize()
                e = time.perf_counter() - t0
                if dist is not None:
                    tt = torch.tensor([e], device=dev, dtype=torch.float64)
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    e = float(tt.item())
                best = e if best is None else min(best, e)
        return best, last

    el, y = timed(model)
    out = {"images_per_s": world * B * args.alexnet_iters / el, "batch_per_gpu": B,
           "ms_per_forward": el / args.alexnet_iters * 1e3, "mode": "eval (pre-packed weights), channels_last; best of 3 runs of alexnet_iters forwards",
           "macs_per_image": 4.9349e9, "finite": bool(torch.isfinite(y).all())}
    # same network fused for inference (layers.fused): every BinConv2d emits BatchNorm-threshold bits, MaxPool runs
    # on bits, FC blocks use the BN+Hardtanh+sign+pack kernel: no fp32 activation between binarised layers
    fused = bench_models.FusedAlexNetBin(model)
    elf, yf = timed(fused)
    out["fused"] = {"images_per_s": world * B * args.alexnet_iters / elf,
                    "ms_per_forward": elf / args.alexnet_iters * 1e3,
                    "same_argmax_as_unfused": bool(torch.equal(yf.argmax(1), y.argmax(1)))}
    # SURVEY 8d asks for the train-mode form as well: the quantised layers in training mode (sign + pack of W on every
    # call, STE autograd nodes), BatchNorm kept on its running statistics for determinism
    from pytorch_quantize_impls_amd.layers import BinConv2d, LinearBin
    qlayers = [m for m in model.modules() if isinstance(m, (BinConv2d, LinearBin))]
    for m in qlayers:
        m.train()
    elt, yt = timed(model)
    for m in qlayers:
        m.eval()
    out["train_mode_layers"] = {"images_per_s": world * B * args.alexnet_iters / elt,
                                "ms_per_forward": elt / args.alexnet_iters * 1e3,
                                "same_logits_as_eval": bool(torch.equal(yt, y))}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cb = min(B, 16)
        cpu_model = bench_models.AlexNetBin()
        cpu_model.load_state_dict({k: v.cpu() for k, v in model.state_dict().items()})
        cpu_model.eval()
        xc = x[:cb].cpu().contiguous()
        from oracle import torch_port
        with torch.no_grad():
            torch_port.best_thread_count(lambda: cpu_model(xc), probe_iters=1)
            med, iters = torch_port.time_callable(lambda: cpu_model(xc), budget_s=min(args.cpu_budget_s, 8.0))
        out["cpu_baseline"] = {"images_per_s": cb / med, "batch": cb, "cores": torch.get_num_threads(),
                               "kind": "port", "sample": f"{iters} forwards of the same topology on CPU tensors, median"}
    return out


def pmc_traffic(gemm_impl):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of
    this same command (profiles/pmc_latest.json; counters cannot be collected from inside the
    process).  FETCH_SIZE is doubled (gfx950 half-count of wide coalesced reads, MI355X_MICROARCH.md
    HBM section); WRITE_SIZE as reported."""
    path = os.path.join(ROOT, "profiles", "pmc_latest.json")
    try:
        with open(path) as fh:
            d = json.load(fh)
        k = d["kernels"]["mfma_gemm_kernel" if gemm_impl == "mfma" else "popc_gemm_kernel"]
        return (2.0 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024.0, d.get("source", path)
    except (OSError, KeyError, ValueError):
        return None, None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip() + f" ({os.cpu_count()} logical)"
    except OSError:
        pass
    return f"unknown ({os.cpu_count()} logical)"


if __name__ == "__main__":
    main()
