"""The N>1 path is a plain batch shard with no data-path collective: world_size-2 gloo run on CPU
checks that (a) each rank's shard of the batch produces exactly its rows of the single-process
result, (b) the only collectives are the timing barrier / MAX reduction bench.py uses."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorch_quantize_impls_amd import synth
    from pytorch_quantize_impls_amd.functions import BinaryConnect
    from pytorch_quantize_impls_amd.layers import LinearBin
    B, K, N = 64, 200, 24
    x = torch.from_numpy(synth.normal(1, (B, K)))
    w = torch.from_numpy(synth.uniform(2, (N, K), -1, 1))
    layer = LinearBin(K, N, bias=False)       # weights replicated on every rank
    layer.weight.data.copy_(w)
    shard = x[rank * (B // world):(rank + 1) * (B // world)]   # contiguous batch split
    y = layer(BinaryConnect()(shard))
    dist.barrier()
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's only reduction: max elapsed time
    assert t.item() == world
    np.save(os.path.join(out_dir, f"y{rank}.npy"), y.detach().numpy())
    dist.destroy_process_group()


def test_batch_shard_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    from pytorch_quantize_impls_amd import synth
    from oracle import oracle as O
    x = synth.normal(1, (64, 200))
    w = synth.uniform(2, (24, 200), -1, 1)
    full = O.linear_bin_forward(O.safe_sign(x), w)
    got = np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(world)], 0)
    assert np.array_equal(got, full)


def _ddp_worker(rank, world, port, out_dir):
    """Data-parallel training step: the layers are ordinary nn.Modules, so torch DDP (gradient all-reduce; RCCL on
    device, gloo here) averages the STE-masked gradients — nothing quantisation-specific crosses ranks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torch.nn.parallel import DistributedDataParallel as DDP
    import bench_models
    torch.manual_seed(0)                                   # identical replicas
    model = bench_models.BinMLP(in_features=40, hidden=32, out_features=5)
    ddp = DDP(model)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 40, generator=g)
    tgt = torch.randint(0, 5, (16,), generator=g)
    lo, hi = rank * 8, (rank + 1) * 8
    loss = torch.nn.functional.nll_loss(ddp(x[lo:hi]), tgt[lo:hi])
    loss.backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    torch.save(grads, os.path.join(out_dir, f"g{rank}.pt"))
    dist.destroy_process_group()


def test_ddp_gradients_average_over_ranks(tmp_path):
    """world_size 2 on CPU: after backward every rank holds the mean of the two per-shard gradients.  BatchNorm uses
    per-shard batch statistics (the reference has no SyncBN either), so the expectation is built the same way."""
    world = 2
    mp.spawn(_ddp_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import bench_models
    g0, g1 = (torch.load(tmp_path / f"g{r}.pt") for r in range(world))
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k                 # all-reduced: identical on both ranks
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(16, 40, generator=gen)
    tgt = torch.randint(0, 5, (16,), generator=gen)
    want = None
    for r in range(world):
        torch.manual_seed(0)
        model = bench_models.BinMLP(in_features=40, hidden=32, out_features=5)
        torch.nn.functional.nll_loss(model(x[r * 8:(r + 1) * 8]), tgt[r * 8:(r + 1) * 8]).backward()
        gr = {k: p.grad for k, p in model.named_parameters()}
        want = gr if want is None else {k: (want[k] + gr[k]) / 2 for k in gr}
    for k in want:
        assert torch.allclose(g0[k], want[k], rtol=1e-5, atol=1e-6), k


def _sync_worker(rank, world, port, out_dir):
    """The path-owned gradient all-reduce (utils/data_parallel.py): bucketed, launched from post-accumulate hooks while
    backward runs, one bucket per few parameters here so that several collectives are in flight."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench_models
    from pytorch_quantize_impls_amd.utils import GradientSynchronizer, broadcast_parameters, clamp_weights_
    torch.manual_seed(100 + rank)                          # DIFFERENT initial replicas: the broadcast makes them equal
    model = bench_models.BinMLP(in_features=40, hidden=32, out_features=5)
    broadcast_parameters(model, src=0)
    sync = GradientSynchronizer(model.parameters(), bucket_bytes=256, overlap=(os.environ.get("QT_SYNC_OVERLAP", "1") == "1"))
    assert len(sync.buckets) >= 3
    opt = torch.optim.SGD(model.parameters(), lr=0.5)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(16, 40, generator=g)
    tgt = torch.randint(0, 5, (16,), generator=g)
    lo, hi = rank * 8, (rank + 1) * 8
    for step in range(2):                                  # two steps: the hooks re-arm
        opt.zero_grad(set_to_none=(step == 1))
        torch.nn.functional.nll_loss(model(x[lo:hi]), tgt[lo:hi]).backward()
        sync.wait()
        if step == 0:
            torch.save({k: p.grad.clone() for k, p in model.named_parameters()}, os.path.join(out_dir, f"s{rank}.pt"))
        opt.step()
        clamp_weights_(model)
    torch.save({k: p.detach().clone() for k, p in model.named_parameters()}, os.path.join(out_dir, f"w{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_synchronizer_world2_without_overlap(tmp_path, monkeypatch):
    """The same protocol with every bucket launched by wait() (overlap=False, what bench.py --gpus N uses)."""
    monkeypatch.setenv("QT_SYNC_OVERLAP", "0")
    test_gradient_synchronizer_world2(tmp_path)


def test_gradient_synchronizer_world2(tmp_path):
    """world_size 2 on CPU (gloo): identical replicas after the broadcast, every rank's gradients = the mean of the per-shard
    gradients of rank 0's initial weights, weights stay identical and inside [-1, 1] after two optimiser steps + clamp."""
    world = 2
    mp.spawn(_sync_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import bench_models
    g0, g1 = (torch.load(tmp_path / f"s{r}.pt") for r in range(world))
    w0, w1 = (torch.load(tmp_path / f"w{r}.pt") for r in range(world))
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
        assert torch.equal(w0[k], w1[k]), k
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(16, 40, generator=gen)
    tgt = torch.randint(0, 5, (16,), generator=gen)
    want = None
    for r in range(world):
        torch.manual_seed(100)                             # rank 0's replica
        model = bench_models.BinMLP(in_features=40, hidden=32, out_features=5)
        torch.nn.functional.nll_loss(model(x[r * 8:(r + 1) * 8]), tgt[r * 8:(r + 1) * 8]).backward()
        gr = {k: p.grad for k, p in model.named_parameters()}
        want = gr if want is None else {k: (want[k] + gr[k]) / 2 for k in gr}
    for k in want:
        assert torch.allclose(g0[k], want[k], rtol=1e-5, atol=1e-6), k
    from pytorch_quantize_impls_amd.layers import LinearBin
    torch.manual_seed(100)
    ref_model = bench_models.BinMLP(in_features=40, hidden=32, out_features=5)
    for name, m in ref_model.named_modules():
        if isinstance(m, LinearBin):
            assert float(w0[name + ".weight"].abs().max()) <= 1.0


def test_gradient_synchronizer_single_process_is_a_noop():
    import bench_models
    from pytorch_quantize_impls_amd.utils import GradientSynchronizer, clamp_weights_
    model = bench_models.BinMLP(in_features=12, hidden=8, out_features=3)
    sync = GradientSynchronizer(model.parameters())
    x = torch.randn(4, 12)
    torch.nn.functional.nll_loss(model(x), torch.tensor([0, 1, 2, 0])).backward()
    before = {k: p.grad.clone() for k, p in model.named_parameters()}
    sync.wait()
    for k, p in model.named_parameters():
        assert torch.equal(p.grad, before[k])
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(5.0)
    clamp_weights_(model)
    from pytorch_quantize_impls_amd.layers import LinearBin
    assert all(float(m.weight.abs().max()) <= 1.0 for m in model.modules() if isinstance(m, LinearBin))


def _order_worker(rank, world, port, out_dir):
    """Ranks that differ in WHICH parameters receive gradients: rank 1 never uses the last layer's second head.
    Every rank must still issue the same all-reduces in the same order; and a second backward before wait() must raise."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorch_quantize_impls_amd.layers import LinearBin
    from pytorch_quantize_impls_amd.utils import GradientSynchronizer
    torch.manual_seed(3)
    trunk, head_a, head_b = LinearBin(20, 16), LinearBin(16, 4), LinearBin(16, 4)
    params = list(trunk.parameters()) + list(head_a.parameters()) + list(head_b.parameters())
    sync = GradientSynchronizer(params, bucket_bytes=64, overlap=True)        # one bucket per parameter
    assert len(sync.buckets) == 6
    g = torch.Generator().manual_seed(9)
    x = torch.randn(8, 20, generator=g)[rank * 4:(rank + 1) * 4]
    h = trunk(x)
    loss = head_a(h).sum() + (head_b(h).sum() if rank == 0 else 0.0)         # head_b: no gradient on rank 1
    loss.backward()
    sync.wait()
    torch.save({"trunk": trunk.weight.grad.clone(), "a": head_a.weight.grad.clone(), "b": head_b.weight.grad.clone()},
               os.path.join(out_dir, f"o{rank}.pt"))
    # second backward before wait(): an error, not a stale average
    h = trunk(x)
    head_a(h).sum().backward()
    raised = False
    try:
        head_a(trunk(x)).sum().backward()
    except RuntimeError as e:
        raised = "second backward" in str(e)
    assert raised
    dist.destroy_process_group()


def test_gradient_synchronizer_launch_order_with_unused_parameters(tmp_path):
    world = 2
    mp.spawn(_order_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    o0, o1 = (torch.load(tmp_path / f"o{r}.pt") for r in range(world))
    for k in o0:
        assert torch.equal(o0[k], o1[k]), k
    from pytorch_quantize_impls_amd.layers import LinearBin
    torch.manual_seed(3)
    trunk, head_a, head_b = LinearBin(20, 16), LinearBin(16, 4), LinearBin(16, 4)
    g = torch.Generator().manual_seed(9)
    xs = torch.randn(8, 20, generator=g)
    h0 = trunk(xs[:4])
    (head_a(h0).sum() + head_b(h0).sum()).backward()
    gb = head_b.weight.grad.clone() / 2          # rank 1 contributed zeros
    assert torch.allclose(o0["b"], gb, rtol=1e-5, atol=1e-6)


def _strong_worker(rank, world, port, out_dir):
    """bench.py --strong on C5's shape: the GLOBAL batch is fixed and cut into world contiguous shards; planes are replicated,
    nothing but the timing barrier crosses ranks."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench_models
    from pytorch_quantize_impls_amd import synth
    torch.manual_seed(11)                                   # replicated weights
    m = bench_models.TernaryVGG16(num_classes=10, image=32, fc=64)
    bench_models.randomize_bn(m, 2)
    m.eval()
    G = 8
    x = torch.from_numpy(synth.pm1(21, (G, 3, 32, 32)))    # +-1 pixels: every layer is exact integer arithmetic
    per = G // world
    with torch.no_grad():
        y = m(x[rank * per:(rank + 1) * per])
    dist.barrier()
    np.save(os.path.join(out_dir, f"v{rank}.npy"), y.numpy())
    dist.destroy_process_group()


def test_strong_scaling_c5_shards_concatenate_to_the_global_batch(tmp_path):
    world = 2
    mp.spawn(_strong_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import bench_models
    from pytorch_quantize_impls_amd import synth
    torch.manual_seed(11)
    m = bench_models.TernaryVGG16(num_classes=10, image=32, fc=64)
    bench_models.randomize_bn(m, 2)
    m.eval()
    x = torch.from_numpy(synth.pm1(21, (8, 3, 32, 32)))
    with torch.no_grad():
        full = m(x).numpy()
    got = np.concatenate([np.load(tmp_path / f"v{r}.npy") for r in range(world)], 0)
    assert np.array_equal(got, full)


def _run_bench(*flags, timeout=600):
    """`python bench.py <flags>` as a user types it — no launcher around it; returns the JSON lines of its stdout."""
    import json
    import subprocess
    import sys
    root = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *flags], cwd=root, env=env, capture_output=True,
                       text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    return [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_bench_self_launches_without_a_launcher():
    """`python bench.py --gpus 2` with WORLD_SIZE unset re-executes itself under torch.distributed.run: two
    distinct processes meet in the barrier / MAX reduction of the timed region and rank 0 alone prints ONE line."""
    lines = _run_bench("--gpus", "2", "--dist-backend", "gloo", "--launch-check")
    assert len(lines) == 1
    d = lines[0]["dist"]
    assert d["world_size_seen_by_the_process_group"] == 2 and d["ranks"] == [0, 1] and d["distinct_processes"] == 2
    assert d["max_over_ranks"] == 2.0 and lines[0]["n_gpus"] == 2
    # the line checks itself (process group size == --gpus, one device per rank) and the compact line carries the
    # C3 AND C5 per-rank times (C5's 8 x 256 split is BASELINE config 5) — assembled by the code path the measured run uses
    assert d["self_check"] == {"world_size_matches_gpus": True, "distinct_devices": True, "ok": True}
    per = lines[0]["compact_line_per_rank"]
    assert per["c3"] == [1.0, 2.0] and per["c5"] == [1.0, 2.0]
