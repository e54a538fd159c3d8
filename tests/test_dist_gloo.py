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
