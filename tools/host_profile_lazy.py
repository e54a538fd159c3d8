#!/usr/bin/env python
"""Host-side profile of the deferred (lazy.py) execution of the un-modified C4 / C3 module graphs: where the Python time
of one forward goes (the GPU work of these forwards is ~1 ms, so the host decides the eager rate)."""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_models  # noqa: E402
from pytorch_quantize_impls_amd import lazy  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    which = sys.argv[1] if len(sys.argv) > 1 else "c4"
    if which == "c4":
        m = bench_models.DorefaResNet18(w_bits=1, a_bits=4)
        x = torch.randn(256, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
    else:
        m = bench_models.AlexNetBin()
        x = torch.randn(256, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    bench_models.randomize_bn(m, 3)
    m = m.to(dev).to(memory_format=torch.channels_last).eval()
    with torch.no_grad():
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            m(x)
        t_host = (time.perf_counter() - t0) / 50
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / 50
        print(f"{which}: host enqueue {t_host * 1e3:.3f} ms / forward, with sync {t_all * 1e3:.3f} ms")
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(50):
            m(x)
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(int(os.environ.get("TOP", "28")))
        if os.environ.get("CUM"):
            pstats.Stats(pr).sort_stats("cumtime").print_stats(int(os.environ.get("TOP", "28")))
        print(dict(lazy.STATS))


if __name__ == "__main__":
    main()
