"""Loader of the EXPERIMENTAL one-launch linear forward (tools/experiments/linear_fused.hip) — not part of libqt_hip.so.

    python tools/experiments/lf.py [EXTRA_HIPCC_FLAGS...]      # builds tools/experiments/build/liblf.so for gfx950

Why it is not in the product: the kernel hands nibble chunks between co-resident workgroups through write-through
stores + per-(panel, chunk) counters.  Bit-exact on an otherwise idle device (check_linear_fused.py), but
stress_linear_fused.py shows STALE hand-offs (error word 0, ~1 stale panel-chunk per launch) whenever another stream's
kernel occupies CUs while it runs, and an agent-scope release before every signal only lowers that to ~1 % at 436 us per
call ("C2 step floor").  It is kept as the measured answer to "why not one persistent launch"."""
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from pytorch_quantize_impls_amd import ops  # noqa: E402

LIB = os.environ.get("QT_LF_LIB") or os.path.join(HERE, "build", "liblf.so")


def build(extra=()):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared",
           "-I", os.path.join(ROOT, "pytorch_quantize_impls_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
           "-DQT_LF_STANDALONE", *extra, os.path.join(HERE, "linear_fused.hip"), "-o", LIB]
    subprocess.run(cmd, check=True)


_lib = None
_WS = {}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = ctypes.CDLL(LIB)
        p, i64 = ctypes.c_void_p, ctypes.c_int64
        _lib.qt_linear_fused_workspace_bytes.restype = i64
        _lib.qt_linear_fused_workspace_bytes.argtypes = [i64, i64, i64]
        _lib.qt_linear_fused_f32.restype = ctypes.c_int
        _lib.qt_linear_fused_f32.argtypes = [p, i64, p, i64, p, p, i64, i64, i64, i64, ctypes.c_int, p, i64, p]
    return _lib


def workspace(device, M, N, K):
    key = (device.index, M, N, K)
    if key not in _WS:
        nbytes = int(lib().qt_linear_fused_workspace_bytes(M, N, K))
        if nbytes <= 0:
            raise NotImplementedError(f"shape {M} x {N} x {K} not taken")
        raw = torch.zeros(nbytes + 4096, dtype=torch.uint8, device=device)
        off = (-raw.data_ptr()) % 4096
        _WS[key] = raw[off:off + nbytes]
    return _WS[key]


def linear_fused(x, w, bias=None, kind="binary", out=None):
    (M, K), N = (int(v) for v in x.shape), int(w.shape[0])
    ws = workspace(x.device, M, N, K)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    rc = lib().qt_linear_fused_f32(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0),
                                   bias.data_ptr() if bias is not None else None, out.data_ptr(), out.stride(0), M, N, K,
                                   0 if kind == "binary" else 1, ws.data_ptr(), ws.numel(), ops._stream(x.device))
    if rc:
        raise RuntimeError(f"qt_linear_fused_f32 returned {rc}")
    return out


def linear_fused_error(device, M, N, K):
    return int(workspace(device, M, N, K)[8:12].view(torch.int32).item())


if __name__ == "__main__":
    build(sys.argv[1:])
    print(LIB)
