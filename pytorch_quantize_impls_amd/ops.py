"""Tensor-level wrappers over the C-ABI (include/qt_hip.h).

Everything here takes/returns torch tensors that live on a HIP device ("cuda" in PyTorch-ROCm);
torch is used only for device memory, streams and dtype bookkeeping.  All buffers are allocated
by torch, kernels are enqueued on ``torch.cuda.current_stream()`` and nothing synchronises.
A CPU tensor is a programming error here (``TypeError``) — there is no fallback.
"""
from __future__ import annotations

import contextlib
import threading
import functools
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib

STE_THRESHOLD = 1.001  # functions/binary_connect.py:37, terner_connect.py:33

#: upper bound on the bytes of the temporary im2col matrix (the batch is processed in chunks)
IM2COL_MAX_BYTES = 1 << 30

#: True: quantised convs run as an implicit GEMM (the LDS-DMA gathers pixel chunks from the NHWC plane,
#: nothing is materialised).  False: explicit packed-domain im2col + GEMM (kept for A/B and as a fallback).
CONV_IMPLICIT = True

# ---- route switches: process-wide DEFAULTS (the module attributes: tools and tests may set them) + per-thread overrides ------------
# ``with ops.scope(FIRST_DIRECT=False):`` changes what THIS thread's calls read and nothing else (two serving threads of one
# process can hold different routes: SURVEY 8e "one process, one stream per device"); library code reads every switch through
# ``_cfg``.  The autograd Functions of the package record the scope their forward ran under and re-open it around their backward
# (functions.common.QtFunction), like ``float_split``.
_SCOPED = ("FIRST_DIRECT", "CONV_IMPLICIT", "POPC_VARIANT", "CONV_VARIANT", "ASSUME_CODES_FIT", "PAD_PIXEL_PLANES", "FIRST_3X3", "DIRECT_BITS_128")
_scope_tls = threading.local()


def _cfg(name: str):
    ov = getattr(_scope_tls, "ov", None)
    if ov is not None and name in ov:
        return ov[name]
    return globals()[name]


def scope_overrides():
    """The overrides an enclosing ``with scope(...)`` of this thread set (a dict), or None."""
    return getattr(_scope_tls, "ov", None)


@contextlib.contextmanager
def scope(_overrides=None, **kw):
    """Thread-local overrides of the route switches (names: ``_SCOPED``); nests; ``scope(None)`` is a no-op."""
    if _overrides:
        kw = {**_overrides, **kw}
    bad = [k for k in kw if k not in _SCOPED]
    if bad:
        raise KeyError(f"not a scoped switch of ops: {bad} (known: {_SCOPED})")
    prev = getattr(_scope_tls, "ov", None)
    if kw:
        _scope_tls.ov = {**(prev or {}), **kw}
    try:
        yield
    finally:
        _scope_tls.ov = prev


@functools.lru_cache(maxsize=None)
def inv_levels(bit_width: int) -> float:
    """fl32(1 / (2^k - 1)) exactly as _quantize forms it (functions/dorefa_connect.py:24: a float32 division)."""
    n = float((1 << int(bit_width)) - 1)
    return float(torch.tensor(1.0, dtype=torch.float32) / torch.tensor(n, dtype=torch.float32))


def relu_mode(relu) -> int:
    """ReLU placement of the fused DoReFa chain: False / None -> 0 (none), True / "post" -> 1 (after BatchNorm and
    residual: quant(relu(bn(x) + r))), "pre" -> 2 (before the BatchNorm: quant(bn(relu(x))),
    models/FullNet/DorefaMNIST.py:46-48)."""
    if relu in (False, None, 0):
        return 0
    if relu in (True, 1, "post"):
        return 1
    if relu in (2, "pre"):
        return 2
    raise ValueError(f"relu must be False, True / 'post' or 'pre', got {relu!r}")


@dataclass
class CodeEpilogue:
    """Arguments of the conv code epilogue (qt_conv2d_implicit_codes): folded BatchNorm (alpha, beta), optional
    residual over the OUTPUT pixels (fp32 [M, Cout] with optional own folded BatchNorm, or CodePlanes), ReLU,
    k-bit DoReFa quantiser; ``overflow``: device int32 flag shared along the chain (None: a fresh one)."""
    alpha: torch.Tensor
    beta: torch.Tensor
    bit_width: int
    relu: bool = True
    res_f32: Optional[torch.Tensor] = None
    res_affine: Optional[tuple] = None
    res_codes: Optional["CodePlanes"] = None
    overflow: Optional[torch.Tensor] = None
    out_halo: tuple = (0, 0)        # zero border (pixels) of the produced plane: [N][Ho + 2hy][Wo + 2hx][ld]
    # "device" BatchNorm arithmetic (layers.fused.device_bn_fold): bn_stats = fp32 [mean | rs] (2 C values), alpha / beta then
    # hold the BatchNorm weight / bias; a 3-tuple res_affine = (weight, bias, stats) does the same for the residual's BatchNorm
    bn_stats: Optional[torch.Tensor] = None
    res_halo: tuple = (0, 0)        # halo of the residual code plane


@dataclass
class BnEpilogue:
    """fp32 conv output through eval-mode BatchNorm in the device's arithmetic, inside the conv's epilogue
    (qt_conv2d_implicit_halo_bn): ``weight`` / ``bias`` / ``stats`` = [mean | rs] as layers.fused.device_bn_fold supplies them."""
    weight: torch.Tensor
    bias: torch.Tensor
    stats: torch.Tensor


def integer_thresholds(bias: Optional[torch.Tensor], alpha: torch.Tensor, beta: torch.Tensor, kmax: int) -> torch.Tensor:
    """Per-channel integer thresholds T of the threshold epilogue for EXACT INTEGER accumulators (|acc| <= kmax):

        fl(fl(acc + bias) * alpha) + beta < 0   <=>   (acc < T) xor (alpha < 0)

    The left side — evaluated with exactly the fp32 roundings of the kernel's float form — is a monotone step function
    of the integer acc (rounding and multiplication by a constant are monotone), so T exists; it is found by bisection
    over [-kmax, kmax + 1], vectorised over the channels (~log2(2 kmax) fp32 evaluations, once per layer).  Constant
    predicates (alpha == 0, NaN parameters) come out as T = kmax + 1 (always) / -kmax (never).  Returns fp32 [C]."""
    alpha = alpha.detach().to(torch.float32)
    beta = beta.detach().to(torch.float32)
    b = torch.zeros_like(alpha) if bias is None else bias.detach().to(torch.float32)
    neg = alpha < 0
    nbe = -beta

    def p(k):      # the kernel's predicate, flipped for alpha < 0 so that it is "true for small k" in every channel
        return ((k.to(torch.float32) + b) * alpha < nbe) ^ neg

    lo = torch.full_like(alpha, -int(kmax), dtype=torch.int64)
    hi = torch.full_like(alpha, int(kmax) + 1, dtype=torch.int64)          # p(hi) counts as false
    for _ in range(max(1, (2 * int(kmax) + 2).bit_length())):
        mid = torch.div(lo + hi, 2, rounding_mode="floor")
        t = p(mid) & (mid <= kmax)
        lo = torch.where(t, mid + 1, lo)
        hi = torch.where(t, hi, mid)
    return hi.to(torch.float32).contiguous()        # the smallest k with p(k) false


@dataclass
class NibEpilogue:
    """Threshold-bit epilogue written as the NEXT conv's fp4 nibble pixel plane (qt_conv2d_implicit_nib): folded
    BatchNorm (alpha, beta); ``out_halo`` = zero border in pixels (the next conv's padding)."""
    alpha: torch.Tensor
    beta: torch.Tensor
    out_halo: tuple = (0, 0)
    d2s_cout: int = 0               # depth-to-space by 2: the conv's 4*d2s_cout columns are (dy, dx, channel)
    thr: Optional[torch.Tensor] = None      # integer_thresholds(...): exact one-compare form for +-1 / 0 operands


def _conv_implicit(elem: int, pixels_words: torch.Tensor, N, H, W, Cw, kh, kw, geom, wmat: torch.Tensor,
                   ldw_words: int, bias, scale: float, scale_dev, Cout: int, epi=None, in_halo=(0, 0)):
    """qt_conv2d_implicit; returns None if the shape is outside its limits (caller falls back).
    ``epi`` = (alpha, beta): threshold-bit epilogue (qt_conv2d_implicit_bits) — returns the BitPlanes of
    [(acc + bias) * alpha + beta < 0] per output pixel instead of the fp32 result."""
    (sh, sw), (ph, pw), (dh, dw) = geom
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    M = N * Ho * Wo
    hy, hx = (int(v) for v in in_halo)
    Hp, Wp = H + 2 * hy, W + 2 * hx
    if (M >= (1 << 31) or kh * kw * Cw * 4 >= (1 << 20) or Hp > 32767 or Wp > 32767 or ldw_words % 32
            or Hp * Wp * Cw * 4 >= (1 << 31)):
        return None
    if (hy or hx) and (ph > hy or pw > hx or N * Hp * Wp * Cw * 4 >= (1 << 32) or kh * kw * Cw * 4 > 32768):
        return None                      # the caller strips the halo
    dev = pixels_words.device
    I = int
    head = (int(elem), _p(pixels_words), I(N), I(H), I(W), I(Cw), I(kh), I(kw), I(sh), I(sw), I(ph), I(pw),
            I(dh), I(dw), _p(wmat), I(ldw_words), _p(bias), float(float(scale)),
            _p(_require(scale_dev, "scale_dev").reshape(1) if scale_dev is not None else None))
    if isinstance(epi, CodeEpilogue):
        alpha, beta = _check_bias(epi.alpha, Cout, dev), _check_bias(epi.beta, Cout, dev)
        if not 2 <= int(epi.bit_width) <= 8:
            raise ValueError("int8 code planes exist for 2 <= bit_width <= 8")
        rf, ra, rb, rc, ldr, ldrc, rscale, rstats = None, None, None, None, 0, 0, 0.0, None
        if epi.res_f32 is not None:
            rf = _require(epi.res_f32, "residual")
            if tuple(rf.shape) != (M, Cout) or (Cout > 1 and rf.stride(1) != 1):
                raise ValueError(f"fp32 residual must be [{M}, {Cout}] with unit channel stride, got {tuple(rf.shape)}")
            ldr = rf.stride(0) if M > 1 else max(Cout, 1)
            if epi.res_affine is not None:
                ra, rb = _check_bias(epi.res_affine[0], Cout, dev), _check_bias(epi.res_affine[1], Cout, dev)
                if len(epi.res_affine) > 2 and epi.res_affine[2] is not None:
                    rstats = _require(epi.res_affine[2], "res_bn_stats").contiguous()
                    if rstats.numel() != 2 * Cout:
                        raise ValueError("res_bn_stats must hold [mean | rs] of the residual's BatchNorm")
        elif epi.res_affine is not None:
            raise ValueError("res_affine needs res_f32")
        stats = None
        if epi.bn_stats is not None:
            stats = _require(epi.bn_stats, "bn_stats").contiguous()
            if stats.numel() != 2 * Cout:
                raise ValueError("bn_stats must hold [mean | rs]: 2 * Cout values")
        ohy, ohx = (int(v) for v in epi.out_halo)
        rhy, rhx = (int(v) for v in epi.res_halo)
        Mo = N * (Ho + 2 * ohy) * (Wo + 2 * ohx)
        if epi.res_codes is not None:
            if epi.res_codes.rows != N * (Ho + 2 * rhy) * (Wo + 2 * rhx) or epi.res_codes.K != Cout:
                raise ValueError(f"residual codes must be a [{N}x{Ho + 2 * rhy}x{Wo + 2 * rhx}, {Cout}] plane")
            rc, ldrc, rscale = epi.res_codes.codes, int(epi.res_codes.codes.shape[1]), float(epi.res_codes.inv_n)
        elif rhy or rhx:
            raise ValueError("res_halo needs res_codes")
        ldc = code_ld_bytes(Cout, 16)
        codes = torch.empty((Mo, ldc), dtype=torch.int8, device=dev)   # the launch writes every byte, halo border included
        flag = epi.overflow if epi.overflow is not None else torch.zeros((1,), dtype=torch.int32, device=dev)
        with _on(dev):
            _lib.call("qt_conv2d_implicit_codes", *head, _p(alpha), _p(beta), _p(rf), I(ldr), _p(ra), _p(rb), _p(rc),
                      I(ldrc), float(rscale), relu_mode(epi.relu),
                      int(int(epi.bit_width)), _p(codes), I(ldc), I(Cout), _p(flag), I(hy), I(hx), I(ohy),
                      I(ohx), I(rhy), I(rhx), _p(stats), _p(rstats), _stream(dev))
        inv_n = inv_levels(epi.bit_width)
        return CodePlanes(codes=codes, rows=Mo, K=Cout, inv_n=inv_n, bit_width=int(epi.bit_width), overflow=flag)
    if isinstance(epi, BnEpilogue):
        if elem != 1 or Cout % 4:
            return None
        y = torch.empty((M, Cout), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.call("qt_conv2d_implicit_halo_bn", int(elem), _p(pixels_words), I(N), I(H), I(W), I(Cw), I(hy), I(hx), I(kh), I(kw),
                      I(sh), I(sw), I(ph), I(pw), I(dh), I(dw), *head[14:], _p(_check_bias(epi.weight, Cout, dev)),
                      _p(_check_bias(epi.bias, Cout, dev)), _p(_require(epi.stats, "bn_stats")), _p(y), I(Cout), I(Cout), _stream(dev))
        return y
    if isinstance(epi, NibEpilogue):
        if hy or hx:
            raise ValueError("input halos are passed as a physically padded image for nibble planes")
        alpha, beta = _check_bias(epi.alpha, Cout, dev), _check_bias(epi.beta, Cout, dev)
        ohy, ohx = (int(v) for v in epi.out_halo)
        d2s = int(epi.d2s_cout)
        if d2s and (d2s % 32 or Cout != 4 * d2s):
            raise ValueError("depth-to-space epilogue: the conv must have 4 * d2s_cout columns, d2s_cout % 32 == 0")
        zs, Cpix = (2, d2s) if d2s else (1, Cout)
        ldn = pixel_ld_nib(Cpix)
        rows = N * (zs * Ho + 2 * ohy) * (zs * Wo + 2 * ohx)
        plane = torch.empty((rows, ldn), dtype=torch.int32, device=dev)     # the launch writes every word, border included
        with _on(dev):
            thr = _check_bias(epi.thr, Cout, dev) if (epi.thr is not None and elem < 2) else None
            _lib.call("qt_conv2d_implicit_nib", *head, _p(alpha), _p(beta), _p(thr), _p(plane), I(ldn), I(Cout), I(ohy),
                      I(ohx), I(d2s), _stream(dev))
        return NibPlanes(words=plane, rows=rows, K=Cpix)
    if hy or hx:
        if epi is not None:
            raise ValueError("halo planes exist for int8 code planes (fp32 / code-epilogue outputs) only")
        y = torch.empty((M, Cout), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.call("qt_conv2d_implicit_halo", int(elem), _p(pixels_words), I(N), I(H), I(W), I(Cw), I(hy),
                      I(hx), I(kh), I(kw), I(sh), I(sw), I(ph), I(pw), I(dh), I(dw), *head[14:], _p(y), I(Cout), I(Cout),
                      _stream(dev))
        return y
    if epi is not None:
        alpha, beta = (_require(t, nm).contiguous() for t, nm in zip(epi[:2], ("alpha", "beta")))
        thr = _check_bias(epi[2], Cout, dev) if (len(epi) > 2 and epi[2] is not None and elem < 2) else None
        if alpha.numel() != Cout or beta.numel() != Cout:
            raise ValueError(f"alpha/beta must have {Cout} entries")
        ldb = packed_ld(Cout)
        plane = torch.empty((M, ldb), dtype=torch.int32, device=dev)   # the kernel writes every word incl. the pad
        with _on(dev):
            _lib.call("qt_conv2d_implicit_bits", *head, _p(alpha), _p(beta), _p(thr), _p(plane), I(ldb), I(Cout),
                      _stream(dev))
        return BitPlanes(sign=plane, rows=M, K=Cout)
    y = torch.empty((M, Cout), dtype=torch.float32, device=dev)
    with _on(dev):
        if _cfg("CONV_VARIANT"):
            _lib.call("qt_conv2d_implicit_variant", int(_cfg("CONV_VARIANT")), *head, _p(y), I(Cout), I(Cout), _stream(dev))
        else:
            _lib.call("qt_conv2d_implicit", *head, _p(y), I(Cout), I(Cout), _stream(dev))
    return y


# Argument marshalling: _lib declares argtypes for every entry point, so plain Python ints / floats / None convert in
# ctypes' C path; building ctypes objects per argument cost more host time than the launch itself (the fused
# ResNet-18 forward is ~40 launches of 20-40 arguments: tools/host_profile_c4.py).
def _p(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None


_SAME_DEVICE = contextlib.nullcontext()


def _on(device):
    """Device guard for a launch: a no-op when ``device`` already is the current device (the usual one-process-per-GPU
    case; torch.cuda.device() costs several microseconds per launch)."""
    idx = device.index
    return _SAME_DEVICE if idx is None or idx == torch.cuda.current_device() else torch.cuda.device(device)


def _stream(device) -> int:
    """Raw hipStream_t of torch's current stream on ``device`` (what the kernels are enqueued on)."""
    idx = device.index
    return torch._C._cuda_getCurrentRawStream(idx if idx is not None else torch.cuda.current_device())


def _require(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t)}")
    if not t.is_cuda:
        raise TypeError(f"{name}: expected a tensor on a HIP device, got device {t.device}; "
                        "the HIP backend has no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t


def packed_ld(K: int) -> int:
    """Row stride (uint32 words) of a bit plane holding K bits: ceil(K/32) rounded up to 4."""
    kw = (int(K) + 31) // 32
    return max(4, (kw + 3) // 4 * 4)


@dataclass
class BitPlanes:
    """Bit-plane image of a [rows, K] matrix of +-1 (sign only) or {-1,0,+1} (mask + sign).

    ``sign``/``mask`` are int32 tensors of shape [rows, ld] (the C side treats them as uint32);
    bit j of word w is element 32*w+j; sign bit 1 <=> negative; mask bit 1 <=> non-zero;
    all bits past K are zero.
    """
    sign: torch.Tensor
    rows: int
    K: int
    mask: Optional[torch.Tensor] = None
    #: the same rows as the matrix-core GEMM's fp4 nibble operand, when the producer wrote it in the same pass (NibPlanes)
    nib: Optional[object] = None

    @property
    def ld(self) -> int:
        return int(self.sign.shape[1])

    @property
    def device(self):
        return self.sign.device

    @property
    def is_ternary(self) -> bool:
        return self.mask is not None


# ----------------------------------------------------------------------------------------------
# elementwise
# ----------------------------------------------------------------------------------------------

def _storage_dense(t: torch.Tensor) -> bool:
    """Dense in NCHW or channels-last order: an elementwise kernel can walk the storage as it lies (empty_like keeps the
    strides), so a channels-last activation or gradient is not transposed to NCHW and back around every elementwise op."""
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def _unary(name: str, x: torch.Tensor, *extra) -> torch.Tensor:
    x = _require(x, "input")
    if not _storage_dense(x):
        x = x.contiguous()
    y = torch.empty_like(x)
    if y.stride() != x.stride():           # size-1 dimensions can leave the order ambiguous
        x = x.contiguous()
        y = torch.empty_like(x)
    with _on(x.device):
        _lib.call(name, _p(x), _p(y), int(x.numel()), *extra, _stream(x.device))
    return y


def binarize(x: torch.Tensor) -> torch.Tensor:
    """safeSign: x<0 -> -1 else +1 (functions/common.py:4-7)."""
    return _unary("qt_binarize_f32", x)


def ternarize(x: torch.Tensor) -> torch.Tensor:
    """TernaryConnectDeterministic.forward (functions/terner_connect.py:24-27)."""
    return _unary("qt_ternarize_f32", x)


def dorefa_quantize(x: torch.Tensor, bit_width: int) -> torch.Tensor:
    """_quantize (functions/dorefa_connect.py:11-25)."""
    return _unary("qt_dorefa_quantize_f32", x, int(int(bit_width)))


def lin_quantize(x: torch.Tensor, fsr: int, bit_width: int, mode: int = 1) -> torch.Tensor:
    """LinQuant forward (mode 0 unsigned / 1 with_sign) or its quantised-gradient backward (mode 2)
    (functions/log_lin_connect.py:61-79)."""
    return _unary("qt_lin_quantize_f32", x, int(int(fsr)), int(int(bit_width)), int(int(mode)))


def log_quantize(x: torch.Tensor, fsr: int, bit_width: int, with_sign: bool = True) -> torch.Tensor:
    """LogQuant forward / quantised-gradient backward (functions/log_lin_connect.py:31-40)."""
    return _unary("qt_log_quantize_f32", x, int(int(fsr)), int(int(bit_width)), int(1 if with_sign else 0))


def ap2(x: torch.Tensor) -> torch.Tensor:
    """safeSign(x) * 2^round(log2|x|) (functions/binary_connect.py:157-169)."""
    return _unary("qt_ap2_f32", x)


def _binary(name: str, a: torch.Tensor, b: torch.Tensor, *extra) -> torch.Tensor:
    a, b = _require(a, "a"), _require(b, "b")
    if a.shape != b.shape:
        raise ValueError(f"shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
    if not _storage_dense(a):
        a = a.contiguous()
    if b.stride() != a.stride():           # bring the second operand into the first one's storage order
        b = b.contiguous(memory_format=torch.channels_last) if (a.dim() == 4 and not a.is_contiguous()) else b.contiguous()
        if b.stride() != a.stride():       # size-1 dimensions leave the strides ambiguous: fall back to NCHW for both
            a, b = a.contiguous(), b.contiguous()
    y = torch.empty_like(a)
    if y.stride() != a.stride():
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
    with _on(a.device):
        _lib.call(name, _p(a), _p(b), _p(y), int(a.numel()), *extra, _stream(a.device))
    return y


def binarize_stochastic(x: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """-1 + 2*[z < hardsigmoid(x)] with caller-drawn z ~ U[0,1) (binary_connect.py:57-61)."""
    return _binary("qt_binarize_stochastic_f32", x, z)


def ternarize_stochastic(x: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """s - s*[z > |x|] (terner_connect.py:54-56)."""
    return _binary("qt_ternarize_stochastic_f32", x, z)


def ste_mask(grad_out: torch.Tensor, x: torch.Tensor, thr: float = STE_THRESHOLD) -> torch.Tensor:
    """grad_out * 1[|x| <= thr] (binary_connect.py:31-38)."""
    return _binary("qt_ste_mask_f32", grad_out, x, float(thr))


def xnor_weight(w: torch.Tensor, lead_dims: int = 1):
    """XNOR-Net weight quantiser: (sign(w) * alpha, alpha) with alpha = mean(|w|) over the first
    ``lead_dims`` dimensions, keepdim (functions/xnor_connect.py:112-113, 140-141)."""
    w = _require(w, "weight").contiguous()
    R = 1
    for d in w.shape[:lead_dims]:
        R *= int(d)
    C = w.numel() // max(R, 1)
    alpha = torch.empty((C,), dtype=torch.float32, device=w.device)
    wq = torch.empty_like(w)
    with _on(w.device):
        _lib.call("qt_xnor_weight_f32", _p(w), int(C), _p(alpha), _p(wq), int(C),
                  int(R), int(C), _stream(w.device))
    return wq, alpha.view((1,) * lead_dims + tuple(w.shape[lead_dims:]))


def xnor_input_quant(x: torch.Tensor, want_image: bool = True, want_scale: bool = False):
    """Input quantiser of XNORConv2d(quant_input=True): sign(x) * mean(|x|, 1, keepdim) (functions/xnor_connect.py:142-143) for an
    NCHW or channels-last fp32 tensor, one pass (qt_xnor_input_quant_f32).  The image is the same logical [N, C, H, W] tensor in
    channels-last memory (its NHWC matrix is what the conv's operand pack and the weight-gradient routes read); ``want_scale``:
    also (or only) the per-pixel scale plane A [N, H, W].  Returns the image, or (image or None, A)."""
    x = _require(x, "input")
    if x.dim() != 4:
        raise ValueError("xnor_input_quant takes a [N, C, H, W] tensor")
    N, C, H, W = (int(v) for v in x.shape)
    y = torch.empty((N, H, W, C), dtype=torch.float32, device=x.device) if want_image else None
    a = torch.empty((N, H, W), dtype=torch.float32, device=x.device) if want_scale else None
    if x.numel():
        sn, sc, sh, sw = (int(v) for v in x.stride())
        with _on(x.device):
            _lib.call("qt_xnor_input_quant_f32", _p(x), sn, sc, sh, sw, _p(y), _p(a), N, C, H, W, _stream(x.device))
    img = y.permute(0, 3, 1, 2) if y is not None else None
    return (img, a) if want_scale else img


def conv2d_nib_taps_rows(x: torch.Tensor, a_plane: torch.Tensor, wplanes: "NibPlanes", kernel_hw, tap_rho: torch.Tensor, bias=None,
                         stride=1, padding=0, dilation=1):
    """XNORConv2d(quant_input=True) at the fp4 rate: sign(x) as an fp4 nibble pixel plane (0 stays 0) against the nibble plane of
    sign(W), alpha per tap AND the per-pixel scale ``a_plane`` [N, H, W] applied on the accumulators, one factor per output row and
    tap (qt_conv2d_implicit_taps_rows).  Returns the NHWC result [N*Ho*Wo, Cout] or None outside the kernel's limits."""
    _require(x, "input")
    N, C, H, W = (int(v) for v in x.shape)
    kh, kw = (int(v) for v in kernel_hw)
    (sh, sw), (ph, pw), (dh, dw) = _pairs(stride), _pairs(padding), _pairs(dilation)
    if kh * kw > 48 or x.numel() == 0:
        return None
    Cw = pixel_ld_nib_taps(C)
    if wplanes.K != kh * kw * Cw * 8:
        raise ValueError("weight planes do not match the activation's channel packing")
    nhwc = x.detach().permute(0, 2, 3, 1)
    if not nhwc.is_contiguous():
        nhwc = nhwc.contiguous()
    px = sign0_pack_nib(nhwc.view(N * H * W, C), ld=Cw)
    Ho, Wo = conv_out_hw(H, W, kh, kw, stride, padding, dilation)
    M, Cout = N * Ho * Wo, wplanes.rows
    if (M >= (1 << 31) or kh * kw * Cw * 4 >= (1 << 20) or H > 32767 or W > 32767 or wplanes.ld % 32
            or H * W * Cw * 4 >= (1 << 31) or Cout * wplanes.ld * 4 >= (1 << 31)):
        return None
    bias = _check_bias(bias, Cout, x.device)
    rho = _require(tap_rho, "tap_rho")
    a = _require(a_plane, "a_plane").contiguous()
    y = torch.empty((M, Cout), dtype=torch.float32, device=x.device)
    I = int
    with _on(x.device):
        _lib.call("qt_conv2d_implicit_taps_rows", _p(px.words), I(N), I(H), I(W), I(Cw), I(kh), I(kw), I(sh), I(sw), I(ph), I(pw), I(dh),
                  I(dw), _p(wplanes.words), I(wplanes.ld), _p(bias), _p(rho), _p(a), _p(y), I(Cout), I(Cout), _stream(x.device))
    return y


def shift_batch(x: torch.Tensor, running_mean, running_var, weight, bias, eps: float, want_saved: bool = True):
    """ShiftBatch.forward on the device (qt_shift_batch_f32): x [N, ...], the four statistic / affine tensors hold one
    entry per element of x[0] (broadcast over N).  Returns (y, norm_inputs or None, sqrtvar or None)."""
    x = _require(x, "input")
    N = int(x.shape[0]) if x.dim() > 0 else 1
    E = x.numel() // max(N, 1)
    vs = []
    for t, nm in ((running_mean, "running_mean"), (running_var, "running_var"), (weight, "weight"), (bias, "bias")):
        t = _require(t.detach(), nm)
        if t.numel() != E:
            raise ValueError(f"{nm} must hold one entry per element of input[0] ({E}), got {t.numel()}")
        vs.append(t.contiguous().view(-1))
    xc = x.detach().contiguous().view(N, E)
    y = torch.empty_like(xc)
    norm = torch.empty_like(xc) if want_saved else None
    sv = torch.empty((E,), dtype=torch.float32, device=x.device) if want_saved else None
    with _on(x.device):
        _lib.call("qt_shift_batch_f32", _p(xc), max(E, 1), _p(vs[0]), _p(vs[1]), _p(vs[2]), _p(vs[3]), float(eps), _p(y),
                  max(E, 1), _p(norm), max(E, 1), _p(sv), N, E, _stream(x.device))
    shape = tuple(x.shape)
    return y.view(shape), (norm.view(shape) if norm is not None else None), sv


def _xnor_act_buffers(x: torch.Tensor, dim: int):
    R, C = (int(v) for v in x.shape)
    n = R if dim == 1 else (C if dim == 0 else 1)
    mean = torch.empty((n,), dtype=torch.float32, device=x.device)
    work = torch.empty((int(_lib.load().qt_xnor_act_work_floats()),), dtype=torch.float32, device=x.device) if dim < 0 else None
    return R, C, mean, work


def xnor_act(x: torch.Tensor, dim: int):
    """XNOR activation quantiser of a 2-D tensor: (sign(x) * mean(x, dim), mean) — qt_xnor_act_f32
    (functions/xnor_connect.py:17-28; the signed mean, as upstream)."""
    x = _require(x, "input")
    if x.dim() != 2:
        raise ValueError("xnor_act: expected a 2-D tensor")
    if x.stride(1) != 1 and x.numel() > 0:
        x = x.contiguous()
    R, C, mean, work = _xnor_act_buffers(x, dim)
    y = torch.empty((R, C), dtype=torch.float32, device=x.device)
    with _on(x.device):
        _lib.call("qt_xnor_act_f32", _p(x), int(x.stride(0)) if R > 1 else max(C, 1), _p(mean), _p(work), _p(y), max(C, 1),
                  R, C, int(dim), _stream(x.device))
    return y, mean


def xnor_act_backward(grad_out: torch.Tensor, x: torch.Tensor, mean: torch.Tensor, dim: int) -> torch.Tensor:
    """sign(x) * mean(g * sign(x), dim, keepdim) + g * mean — qt_xnor_act_backward_f32 (xnor_connect.py:30-37)."""
    x = _require(x, "input")
    g = _require(grad_out, "grad_output")
    if x.stride(1) != 1 and x.numel() > 0:
        x = x.contiguous()
    if g.stride(1) != 1 and g.numel() > 0:
        g = g.contiguous()
    R, C, gmean, work = _xnor_act_buffers(x, dim)
    gin = torch.empty((R, C), dtype=torch.float32, device=x.device)
    with _on(x.device):
        _lib.call("qt_xnor_act_backward_f32", _p(g), int(g.stride(0)) if R > 1 else max(C, 1), _p(x),
                  int(x.stride(0)) if R > 1 else max(C, 1), _p(mean), _p(gmean), _p(work), _p(gin), max(C, 1), R, C, int(dim),
                  _stream(x.device))
    return gin


# ----------------------------------------------------------------------------------------------
# packing
# ----------------------------------------------------------------------------------------------

def _as_rows(x: torch.Tensor) -> torch.Tensor:
    if x.dim() < 1:
        raise ValueError("need at least 1 dimension")
    x2 = x.reshape(-1, x.shape[-1]) if x.dim() != 2 else x
    if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < x2.shape[1]):
        x2 = x2.contiguous()
    return x2


def sign_pack(x: torch.Tensor, want_f32: bool = False) -> Tuple[BitPlanes, Optional[torch.Tensor]]:
    """Sign plane of safeSign(x) packed along the last dimension.

    Returns (planes, y) where y is the +-1 fp32 image (same shape as x) when ``want_f32``.
    """
    _require(x, "input")
    x2 = _as_rows(x)
    rows, K = int(x2.shape[0]), int(x2.shape[1])
    ld = packed_ld(K)
    plane = torch.empty((rows, ld), dtype=torch.int32, device=x.device)
    y = torch.empty((rows, K), dtype=torch.float32, device=x.device) if want_f32 else None
    with _on(x.device):
        _lib.call("qt_sign_pack_f32", _p(x2), int(x2.stride(0) if rows > 1 else max(K, 1)),
                  _p(plane), int(ld), _p(y), int(K), int(rows),
                  int(K), _stream(x.device))
    if y is not None:
        y = y.view(x.shape)
    return BitPlanes(sign=plane, rows=rows, K=K), y


def ternary_pack(x: torch.Tensor) -> BitPlanes:
    """Mask + sign planes of TernaryConnectDeterministic(x), packed along the last dimension."""
    _require(x, "input")
    x2 = _as_rows(x)
    rows, K = int(x2.shape[0]), int(x2.shape[1])
    ld = packed_ld(K)
    mask = torch.empty((rows, ld), dtype=torch.int32, device=x.device)
    sign = torch.empty((rows, ld), dtype=torch.int32, device=x.device)
    with _on(x.device):
        _lib.call("qt_ternary_pack_f32", _p(x2), int(x2.stride(0) if rows > 1 else max(K, 1)),
                  _p(mask), _p(sign), int(ld), int(rows), int(K),
                  _stream(x.device))
    return BitPlanes(sign=sign, rows=rows, K=K, mask=mask)


def pool_affine_sign_pack(x: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor, pool_k: int = 1,
                          pool_s: int = 1, pre_relu: bool = False, want_nib: bool = False):
    """Fused [MaxPool2d(pool_k, pool_s)] -> eval-BatchNorm (x*alpha+beta) -> Hardtanh -> sign -> bit-pack.
    x: [N, C, H, W] fp32 (channels_last storage is used as-is, NCHW storage is transposed once) or
    [N, C].  Returns (BitPlanes with rows = N*Ho*Wo, K = C; (Ho, Wo)).  ``want_nib``: the planes also carry the fp4 nibble rows
    of the same signs (``.nib``), written by the same launch — what the next layer's matrix-core GEMM consumes."""
    _require(x, "input")
    if x.dim() == 2:
        N, C = (int(v) for v in x.shape)
        H = W = 1
        nhwc = x.contiguous()
    else:
        N, C, H, W = (int(v) for v in x.shape)
        nhwc = x.permute(0, 2, 3, 1)
        if not nhwc.is_contiguous():
            nhwc = nhwc.contiguous()
    if C % 4:
        raise ValueError("the fused epilogue needs C % 4 == 0")
    Ho, Wo = (H - pool_k) // pool_s + 1, (W - pool_k) // pool_s + 1
    ld = packed_ld(C)
    plane = torch.empty((N * Ho * Wo, ld), dtype=torch.int32, device=x.device)
    alpha = _require(alpha, "alpha").contiguous()
    beta = _require(beta, "beta").contiguous()
    I = int
    nib = None
    with _on(x.device):
        if want_nib:
            ldn = packed_ld_nib(C)
            words = torch.empty((N * Ho * Wo, ldn), dtype=torch.int32, device=x.device)
            _lib.call("qt_pool_affine_sign_pack_nib_nhwc", _p(nhwc), I(N), I(H), I(W), I(C), I(int(pool_k)), I(int(pool_s)),
                      _p(alpha), _p(beta), _p(plane), I(ld), _p(words), I(ldn), int(1 if pre_relu else 0), _stream(x.device))
            nib = NibPlanes(words=words, rows=N * Ho * Wo, K=C)
        else:
            _lib.call("qt_pool_affine_sign_pack_nhwc", _p(nhwc), I(N), I(H), I(W), I(C), I(int(pool_k)),
                      I(int(pool_s)), _p(alpha), _p(beta), _p(plane), I(ld), int(1 if pre_relu else 0),
                      _stream(x.device))
    return BitPlanes(sign=plane, rows=N * Ho * Wo, K=C, nib=nib), (Ho, Wo)


def check_pm1(x: torch.Tensor, limit: Optional[int] = None) -> torch.Tensor:
    """Device flag (int32 scalar tensor): non-zero iff some element of x (of its first ``limit``
    elements in storage order) is not exactly +-1."""
    x = _require(x, "input")
    if not (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))):
        x = x.contiguous()
    n = x.numel() if limit is None else min(int(limit), x.numel())
    flag = torch.zeros((1,), dtype=torch.int32, device=x.device)
    with _on(x.device):
        _lib.call("qt_check_pm1_f32", _p(x), int(n), _p(flag), _stream(x.device))
    return flag


def is_pm1(x: torch.Tensor) -> bool:
    """Host-side answer (synchronises): is every element exactly +-1?  A 4096-element prefix is
    looked at first so real-valued tensors (first layer of every model) are rejected after a tiny
    kernel instead of a full pass."""
    if x.numel() > 8192 and int(check_pm1(x, 4096).item()) != 0:
        return False
    return int(check_pm1(x).item()) == 0


# ----------------------------------------------------------------------------------------------
# packed GEMMs
# ----------------------------------------------------------------------------------------------

def _check_bias(bias, N, device):
    if bias is None:
        return None
    bias = _require(bias, "bias").contiguous()
    if bias.numel() != N or bias.device != device:
        raise ValueError("bias must be a length-N fp32 tensor on the same device")
    return bias


#: kernel variants handed to the library as ARGUMENTS (tests / tuning; 0 = automatic): popcount GEMM 1 = tiled, 2 = skinny, 3 = streaming (min(M, N) <= 32);
#: plain implicit conv 1 = double-buffered, 2 = ping-pong, 4 = no un-padded fast path
POPC_VARIANT = 0
CONV_VARIANT = 0


def xnor_gemm(x: BitPlanes, w: BitPlanes, bias: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None, variant: Optional[int] = None) -> torch.Tensor:
    """Y[M,N] = sum_k x[m,k]*w[n,k] (+ bias) for +-1 operands given as sign planes.  ``variant``: force one of the popcount kernels
    (1 = tiled, 2 = skinny, 3 = streaming) for this call; default: the library's own choice (POPC_VARIANT, a tools-only default)."""
    if x.K != w.K:
        raise ValueError(f"K mismatch: activations {x.K} vs weights {w.K}")
    if x.is_ternary or w.is_ternary:
        raise ValueError("xnor_gemm takes sign-only planes; use tern_gemm for ternary weights")
    M, N, K = x.rows, w.rows, x.K
    dev = x.device
    bias = _check_bias(bias, N, dev)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
    with _on(dev):
        args = (_p(x.sign), int(x.ld), _p(w.sign), int(w.ld), _p(bias), _p(out), int(out.stride(0) if M > 1 else max(N, 1)),
                int(M), int(N), int(K), _stream(dev))
        v = _cfg("POPC_VARIANT") if variant is None else int(variant)
        if v:
            _lib.call("qt_xnor_gemm_variant", int(v), *args)
        else:
            _lib.call("qt_xnor_gemm", *args)
    return out


def tern_gemm(x: BitPlanes, w: BitPlanes, bias: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Y[M,N] = sum_k x[m,k]*w[n,k] (+ bias): +-1 activations (sign plane) x ternary weights."""
    if x.K != w.K:
        raise ValueError(f"K mismatch: activations {x.K} vs weights {w.K}")
    if x.is_ternary or not w.is_ternary:
        raise ValueError("tern_gemm takes binary activations and ternary (mask+sign) weights")
    M, N, K = x.rows, w.rows, x.K
    dev = x.device
    bias = _check_bias(bias, N, dev)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
    with _on(dev):
        args = (_p(x.sign), int(x.ld), _p(w.mask), _p(w.sign), int(w.ld), _p(bias), _p(out),
                int(out.stride(0) if M > 1 else max(N, 1)), int(M), int(N), int(K), _stream(dev))
        if _cfg("POPC_VARIANT"):
            _lib.call("qt_tern_gemm_variant", int(_cfg("POPC_VARIANT")), *args)
        else:
            _lib.call("qt_tern_gemm", *args)
    return out


# ----------------------------------------------------------------------------------------------
# matrix-core formulation: nibble planes (fp4-e2m1 +-1/0) + MX-fp4 MFMA GEMM
# ----------------------------------------------------------------------------------------------

def packed_ld_nib(K: int) -> int:
    """Row stride (uint32 words) of a nibble plane: ceil(K/8) rounded up to 32 words (one 128-byte
    K stage of the MFMA kernel), so the fast pipelined kernel's contract always holds."""
    kw = (int(K) + 7) // 8
    return max(32, (kw + 31) // 32 * 32)


@dataclass
class NibPlanes:
    """fp4-e2m1 nibble image of a [rows, K] matrix with values in {-1, 0, +1}: int32 tensor
    [rows, ld], element k in nibble (k & 7) of word (k >> 3); +1 = 0x2, -1 = 0xA, 0 = 0x0; pad = 0."""
    words: torch.Tensor
    rows: int
    K: int

    @property
    def ld(self) -> int:
        return int(self.words.shape[1])

    @property
    def device(self):
        return self.words.device


def _nib_pack(entry: str, x: torch.Tensor, ld: Optional[int] = None) -> NibPlanes:
    _require(x, "input")
    x2 = _as_rows(x)
    rows, K = int(x2.shape[0]), int(x2.shape[1])
    ld = packed_ld_nib(K) if ld is None else int(ld)
    words = torch.empty((rows, ld), dtype=torch.int32, device=x.device)
    with _on(x.device):
        _lib.call(entry, _p(x2), int(x2.stride(0) if rows > 1 else max(K, 1)), _p(words),
                  int(ld), int(rows), int(K), _stream(x.device))
    return NibPlanes(words=words, rows=rows, K=K)


def neg_alpha_words(alpha: torch.Tensor) -> torch.Tensor:
    """[packed_ld(C)] int32 words, bit c = (alpha[c] < 0): the AND/OR selector of pool_bits."""
    planes, _ = sign_pack(_require(alpha, "alpha").reshape(1, -1))
    return planes.sign.reshape(-1)


def pool_bits(planes: BitPlanes, N: int, H: int, W: int, pool_k: int, pool_s: int,
              neg_alpha: torch.Tensor) -> Tuple[BitPlanes, Tuple[int, int]]:
    """MaxPool2d(pool_k, pool_s) on threshold bits (qt_pool_bits): AND over the window where alpha >= 0,
    OR where alpha < 0.  planes: NHWC pixel plane [N*H*W][ld] from the conv's threshold-bit epilogue."""
    if planes.rows != N * H * W or planes.mask is not None:
        raise ValueError("pool_bits expects the sign-only pixel plane of an [N, C, H, W] activation")
    if neg_alpha.numel() != planes.ld or neg_alpha.dtype != torch.int32:
        raise ValueError("neg_alpha must be neg_alpha_words(alpha) of the same channel count")
    if pool_k > H or pool_k > W:
        raise ValueError("pooling window larger than the image")
    Ho, Wo = (H - pool_k) // pool_s + 1, (W - pool_k) // pool_s + 1
    out = torch.empty((N * Ho * Wo, planes.ld), dtype=torch.int32, device=planes.device)
    I = int
    with _on(planes.device):
        _lib.call("qt_pool_bits", _p(planes.sign), I(N), I(H), I(W), I(planes.ld), I(pool_k), I(pool_s),
                  _p(neg_alpha), _p(out), _stream(planes.device))
    return BitPlanes(sign=out, rows=N * Ho * Wo, K=planes.K), (Ho, Wo)


def pool_bits_nib(planes: BitPlanes, N: int, H: int, W: int, pool_k: int, pool_s: int, neg_alpha: torch.Tensor,
                  out_halo=(0, 0)) -> Tuple[NibPlanes, Tuple[int, int]]:
    """pool_bits with the result written as the next conv's nibble pixel plane, optionally with a zero halo
    (qt_pool_bits_nib = qt_pool_bits + qt_bits_to_nib_pad in one pass)."""
    if planes.rows != N * H * W or planes.mask is not None:
        raise ValueError("pool_bits expects the sign-only pixel plane of an [N, C, H, W] activation")
    if neg_alpha.numel() != planes.ld or neg_alpha.dtype != torch.int32:
        raise ValueError("neg_alpha must be neg_alpha_words(alpha) of the same channel count")
    if pool_k > H or pool_k > W:
        raise ValueError("pooling window larger than the image")
    hy, hx = (int(v) for v in out_halo)
    Ho, Wo = (H - pool_k) // pool_s + 1, (W - pool_k) // pool_s + 1
    ldn = pixel_ld_nib(planes.K)
    rows = N * (Ho + 2 * hy) * (Wo + 2 * hx)
    out = torch.empty((rows, ldn), dtype=torch.int32, device=planes.device)
    I = int
    with _on(planes.device):
        _lib.call("qt_pool_bits_nib", _p(planes.sign), I(N), I(H), I(W), I(planes.ld), I(pool_k), I(pool_s),
                  _p(neg_alpha), _p(out), I(ldn), I(planes.K), I(hy), I(hx), _stream(planes.device))
    return NibPlanes(words=out, rows=rows, K=planes.K), (Ho, Wo)


def sign_pack_nib(x: torch.Tensor, ld: Optional[int] = None) -> NibPlanes:
    """Nibble plane of safeSign(x) along the last dimension (``ld``: row stride in words, % 4)."""
    return _nib_pack("qt_sign_pack_nib_f32", x, ld)


def ternary_pack_nib(x: torch.Tensor, ld: Optional[int] = None) -> NibPlanes:
    """Nibble plane of TernaryConnectDeterministic(x) along the last dimension."""
    return _nib_pack("qt_ternary_pack_nib_f32", x, ld)


def sign0_pack_nib(x: torch.Tensor, ld: Optional[int] = None) -> NibPlanes:
    """Nibble plane of torch.sign(x) (0 stays 0): the sign image of the XNOR-Net weight quantiser (xnor_connect.py:141)."""
    return _nib_pack("qt_sign0_pack_nib_f32", x, ld)


def bits_to_nib(planes: BitPlanes, ld: Optional[int] = None) -> NibPlanes:
    """Expand 1-bit planes (sign, or mask + sign) to the nibble plane the MFMA GEMM consumes."""
    ld = packed_ld_nib(planes.K) if ld is None else int(ld)
    words = torch.empty((planes.rows, ld), dtype=torch.int32, device=planes.device)
    with _on(planes.device):
        _lib.call("qt_bits_to_nib", _p(planes.sign), _p(planes.mask), int(planes.ld),
                  _p(words), int(ld), int(planes.rows), int(planes.K),
                  _stream(planes.device))
    return NibPlanes(words=words, rows=planes.rows, K=planes.K)


def bits_to_nib_pad(planes: BitPlanes, N: int, H: int, W: int, padding, ld: Optional[int] = None) -> NibPlanes:
    """NHWC pixel bit planes ([N*H*W] rows) -> nibble pixel plane of the zero-padded image
    ([N*(H+2ph)*(W+2pw)] rows; border pixels are fp4 zeros)."""
    ph, pw = _pairs(padding)
    if planes.rows != N * H * W:
        raise ValueError("planes do not hold an [N, C, H, W] activation")
    ld = packed_ld_nib(planes.K) if ld is None else int(ld)
    rows = N * (H + 2 * ph) * (W + 2 * pw)
    words = torch.empty((rows, ld), dtype=torch.int32, device=planes.device)
    I = int
    with _on(planes.device):
        _lib.call("qt_bits_to_nib_pad", _p(planes.sign), _p(planes.mask), I(planes.ld), _p(words), I(ld), I(N), I(H),
                  I(W), I(ph), I(pw), I(planes.K), _stream(planes.device))
    return NibPlanes(words=words, rows=rows, K=planes.K)


def nib_gemm(x: NibPlanes, w: NibPlanes, bias: Optional[torch.Tensor] = None,
             out: Optional[torch.Tensor] = None, variant: Optional[int] = None) -> torch.Tensor:
    """Y[M,N] = sum_k x[m,k]*w[n,k] (+ bias) on the matrix cores; bit-identical to the popcount
    GEMMs.  ``variant`` selects an explicit kernel configuration (tuning only)."""
    if x.K != w.K:
        raise ValueError(f"K mismatch: activations {x.K} vs weights {w.K}")
    M, N, K = x.rows, w.rows, x.K
    dev = x.device
    bias = _check_bias(bias, N, dev)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
    args = (_p(x.words), int(x.ld), _p(w.words), int(w.ld), _p(bias), _p(out),
            int(out.stride(0) if M > 1 else max(N, 1)), int(M), int(N),
            int(K), _stream(dev))
    with _on(dev):
        if variant is None:
            _lib.call("qt_nib_gemm", *args)
        else:
            _lib.call("qt_nib_gemm_variant", int(int(variant)), *args)
    return out


# ----------------------------------------------------------------------------------------------
# DoReFa k-bit path: int8 code planes + int8 MFMA GEMM
# ----------------------------------------------------------------------------------------------

def code_ld_bytes(K: int, granule: int = 128) -> int:
    """Row stride in BYTES of an int8 code plane: K rounded up to a whole GEMM stage (128 B), or to
    16 B for NHWC pixel planes (granule=16)."""
    return max(granule, (int(K) + granule - 1) // granule * granule)


#: opt-in: skip the per-activation int8-overflow check (one host sync per quantised tensor; 13 % of the C4 ResNet-18
#: forward, tools/bench_c4_c5.py).  Only safe when |(2^k - 1) x| <= 127 is guaranteed by the model.
ASSUME_CODES_FIT = False


@dataclass
class CodePlanes:
    """int8 code image of a [rows, K] matrix: activations q = rint((2^k-1) x) (value = inv_n * q) or
    weight codes in {-1, 0, +1}.  ``codes``: int8 tensor [rows, ld_bytes], pad bytes zero."""
    codes: torch.Tensor
    rows: int
    K: int
    inv_n: float = 1.0
    bit_width: int = 1
    overflow: Optional[torch.Tensor] = None   # device int32 flag: some |q| > 127 (then do not use)
    _usable: Optional[bool] = None             # host-resolved once (one sync per activation tensor)

    def usable(self) -> bool:
        """True iff every code fits int8 (the reference does not clamp; an out-of-range activation
        leaves the packed path).  Resolving the device flag synchronises once per tensor — unless the caller
        vouches for the range (ASSUME_CODES_FIT, e.g. activations clipped to [0, 1] as in the DoReFa paper:
        |q| <= 2^k - 1 <= 127 for k <= 7), which also makes such a forward hipGraph-capturable."""
        if _cfg("ASSUME_CODES_FIT"):
            return True
        if self._usable is None:
            self._usable = self.overflow is None or int(self.overflow.item()) == 0
        return self._usable

    @property
    def ld_words(self) -> int:
        return int(self.codes.shape[1]) // 4

    @property
    def device(self):
        return self.codes.device


def dorefa_codes(x: torch.Tensor, bit_width: int, want_f32: bool = True, ld_bytes: Optional[int] = None):
    """k-bit DoReFa activation quantiser producing int8 codes (and the fp32 image _quantize returns,
    functions/dorefa_connect.py:24-25) in one pass.  Returns (CodePlanes, y or None)."""
    _require(x, "input")
    if not 2 <= int(bit_width) <= 8:
        raise ValueError("int8 code planes exist for 2 <= bit_width <= 8")
    x2 = _as_rows(x)
    rows, K = int(x2.shape[0]), int(x2.shape[1])
    ld = code_ld_bytes(K) if ld_bytes is None else int(ld_bytes)
    codes = torch.empty((rows, ld), dtype=torch.int8, device=x.device)
    y = torch.empty((rows, K), dtype=torch.float32, device=x.device) if want_f32 else None
    flag = torch.zeros((1,), dtype=torch.int32, device=x.device)
    with _on(x.device):
        _lib.call("qt_dorefa_codes_i8", _p(x2), int(x2.stride(0) if rows > 1 else max(K, 1)),
                  _p(codes), int(ld), _p(y), int(K), int(rows),
                  int(K), int(int(bit_width)), _p(flag), _stream(x.device))
    inv_n = inv_levels(bit_width)
    if y is not None:
        y = y.view(x.shape)
    return CodePlanes(codes=codes, rows=rows, K=K, inv_n=inv_n, bit_width=int(bit_width), overflow=flag), y


def affine_dorefa_codes(x2: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor, bit_width: int, relu: bool = True,
                        res_f32: Optional[torch.Tensor] = None, res_affine=None,
                        res_codes: Optional[CodePlanes] = None, want_f32: bool = False,
                        overflow: Optional[torch.Tensor] = None, ld_bytes: Optional[int] = None,
                        bn_stats: Optional[torch.Tensor] = None, halo_nhw=None, out_halo=(0, 0)):
    """Fused eval BatchNorm (folded alpha, beta) [+ residual] [-> ReLU] -> k-bit DoReFa quantiser over a
    [rows, C] fp32 matrix (a conv output viewed as pixels x channels): returns (CodePlanes, fp32 image or None).
    ``res_f32``: fp32 [rows, C] residual, optionally with its own folded BatchNorm ``res_affine`` = (alpha, beta);
    ``res_codes``: residual held as DoReFa codes (value inv_n * code).  ``overflow``: device int32 flag to OR into
    (a fresh one is made when None).  ``bn_stats`` = [mean | rs]: the device's BatchNorm arithmetic with alpha / beta =
    weight / bias (qt_affine_dorefa_codes_i8); a 3-tuple ``res_affine`` = (weight, bias, stats) likewise for the residual.
    ``halo_nhw`` = (N, H, W) with ``out_halo`` = (hy, hx): the rows are the pixels of N images and the codes are written into a
    [N*(H+2hy)*(W+2hx), ld] plane with a zero border (qt_affine_dorefa_codes_halo_i8; no fp32 image)."""
    _require(x2, "input")
    if x2.dim() != 2 or x2.dtype != torch.float32 or (x2.shape[1] > 1 and x2.stride(1) != 1):
        raise ValueError("affine_dorefa_codes takes a [rows, C] fp32 matrix with unit channel stride")
    if not 2 <= int(bit_width) <= 8:
        raise ValueError("int8 code planes exist for 2 <= bit_width <= 8")
    rows, C = int(x2.shape[0]), int(x2.shape[1])
    dev = x2.device
    alpha, beta = _check_bias(alpha, C, dev), _check_bias(beta, C, dev)
    ra = rb = rstats = None
    ldr = 0
    if res_f32 is not None:
        if tuple(res_f32.shape) != (rows, C) or res_f32.dtype != torch.float32 or (C > 1 and res_f32.stride(1) != 1):
            raise ValueError("fp32 residual must be a [rows, C] fp32 matrix like the input")
        ldr = res_f32.stride(0) if rows > 1 else max(C, 1)
        if res_affine is not None:
            ra, rb = _check_bias(res_affine[0], C, dev), _check_bias(res_affine[1], C, dev)
            if len(res_affine) > 2 and res_affine[2] is not None:
                rstats = _require(res_affine[2], "res_bn_stats").contiguous()
                if rstats.numel() != 2 * C:
                    raise ValueError("res_bn_stats must hold [mean | rs] of the residual's BatchNorm")
    elif res_affine is not None:
        raise ValueError("res_affine needs res_f32")
    if bn_stats is not None:
        bn_stats = _require(bn_stats, "bn_stats").contiguous()
        if bn_stats.numel() != 2 * C:
            raise ValueError("bn_stats must hold [mean | rs]: 2 * C values")
    rscale, ldrc = 0.0, 0
    if res_codes is not None:
        if res_codes.rows != rows or res_codes.K != C:
            raise ValueError("residual codes must be a [rows, C] code plane like the input")
        rscale, ldrc = float(res_codes.inv_n), int(res_codes.codes.shape[1])
    ld = code_ld_bytes(C) if ld_bytes is None else int(ld_bytes)
    flag = overflow if overflow is not None else torch.zeros((1,), dtype=torch.int32, device=dev)
    I = int
    inv_n = inv_levels(bit_width)
    hy, hx = (int(v) for v in out_halo)
    if halo_nhw is not None and (hy or hx):
        N, H, W = (int(v) for v in halo_nhw)
        if N * H * W != rows or want_f32:
            raise ValueError("halo output: rows must be the N*H*W pixels of the images, and there is no fp32 image")
        prow = N * (H + 2 * hy) * (W + 2 * hx)
        codes = torch.empty((prow, ld), dtype=torch.int8, device=dev)
        with _on(dev):
            _lib.call("qt_affine_dorefa_codes_halo_i8", _p(x2), I(x2.stride(0) if rows > 1 else max(C, 1)), _p(alpha), _p(beta),
                      _p(res_f32), I(ldr), _p(ra), _p(rb), _p(res_codes.codes if res_codes is not None else None), I(ldrc),
                      float(rscale), relu_mode(relu), _p(codes), I(ld), I(N), I(H), I(W), I(C), int(int(bit_width)), _p(flag),
                      _p(bn_stats), _p(rstats), I(hy), I(hx), _stream(dev))
        return CodePlanes(codes=codes, rows=prow, K=C, inv_n=inv_n, bit_width=int(bit_width), overflow=flag), None
    codes = torch.empty((rows, ld), dtype=torch.int8, device=dev)
    y = torch.empty((rows, C), dtype=torch.float32, device=dev) if want_f32 else None
    with _on(dev):
        _lib.call("qt_affine_dorefa_codes_i8", _p(x2), I(x2.stride(0) if rows > 1 else max(C, 1)), _p(alpha), _p(beta),
                  _p(res_f32), I(ldr), _p(ra), _p(rb), _p(res_codes.codes if res_codes is not None else None), I(ldrc),
                  float(rscale), relu_mode(relu), _p(codes), I(ld), _p(y), I(C), I(rows), I(C),
                  int(int(bit_width)), _p(flag), _p(bn_stats), _p(rstats), _stream(dev))
    return CodePlanes(codes=codes, rows=rows, K=C, inv_n=inv_n, bit_width=int(bit_width), overflow=flag), y


def codes_to_f32(codes: CodePlanes, N: int, H: int, W: int, halo=(0, 0), pool_k: int = 1, flagged: bool = True) -> torch.Tensor:
    """The fp32 image fl(inv_n * code) of a code plane [N*(H+2hy)*(W+2hx), ld] as an [N, Ho, Wo, C] NHWC tensor, optionally through
    avg_pool2d(pool_k) in the same pass (qt_codes_to_f32); ``flagged``: a raised int8 range flag of the chain makes every value NaN."""
    hy, hx = (int(v) for v in halo)
    if codes.rows != N * (H + 2 * hy) * (W + 2 * hx):
        raise ValueError(f"code plane holds {codes.rows} pixels, ({N}, {H}, {W}) with halo {(hy, hx)} needs {N * (H + 2 * hy) * (W + 2 * hx)}")
    pk = int(pool_k)
    if pk < 1 or pk > H or pk > W:
        raise ValueError("pool_k must lie in [1, min(H, W)]")
    C, dev = int(codes.K), codes.codes.device
    Ho, Wo = H // pk, W // pk
    y = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=dev)
    flag = codes.overflow if (flagged and codes.overflow is not None) else None
    I = int
    with _on(dev):
        _lib.call("qt_codes_to_f32", _p(codes.codes), I(codes.codes.stride(0)), I(N), I(H), I(W), I(hy), I(hx), I(C), float(codes.inv_n),
                  _p(flag), I(pk), _p(y), I(C), _stream(dev))
    return y


def bn_eval_device(x2: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, bn_stats: torch.Tensor) -> torch.Tensor:
    """Eval-mode BatchNorm of an fp32 [rows, C] matrix in the device's arithmetic (qt_bn_eval_device_f32; layers.fused.device_bn_fold
    supplies weight, bias and bn_stats = [mean | rs] and verifies the expression against F.batch_norm)."""
    _require(x2, "input")
    rows, C = int(x2.shape[0]), int(x2.shape[1])
    if x2.dim() != 2 or (C > 1 and x2.stride(1) != 1):
        raise ValueError("bn_eval_device takes a [rows, C] fp32 matrix with unit channel stride")
    y = torch.empty((rows, C), dtype=torch.float32, device=x2.device)
    with _on(x2.device):
        _lib.call("qt_bn_eval_device_f32", _p(x2), int(x2.stride(0) if rows > 1 else max(C, 1)), _p(_require(weight, "weight")),
                  _p(_require(bias, "bias")), _p(_require(bn_stats, "bn_stats")), _p(y), int(C), int(rows), int(C), _stream(x2.device))
    return y


def pool_codes(codes: CodePlanes, N: int, H: int, W: int, pool_k: int, pool_s: int, out_halo=(0, 0)) -> CodePlanes:
    """MaxPool2d(pool_k, pool_s) on an NHWC code plane [N*H*W, ld] -> [N*(Ho+2hy)*(Wo+2hx), ld] (max of codes =
    code of the max: the quantised value is monotone in its code)."""
    if codes.rows != N * H * W:
        raise ValueError(f"code plane holds {codes.rows} pixels, ({N}, {H}, {W}) needs {N * H * W}")
    hy, hx = (int(v) for v in out_halo)
    Ho, Wo = (H - pool_k) // pool_s + 1, (W - pool_k) // pool_s + 1
    ld = int(codes.codes.shape[1])
    dev = codes.device
    out = torch.empty((N * (Ho + 2 * hy) * (Wo + 2 * hx), ld), dtype=torch.int8, device=dev)
    I = int
    with _on(dev):
        _lib.call("qt_pool_codes_i8", _p(codes.codes), I(N), I(H), I(W), I(ld), I(pool_k), I(pool_s), _p(out), I(hy),
                  I(hx), _stream(dev))
    return CodePlanes(codes=out, rows=int(out.shape[0]), K=codes.K, inv_n=codes.inv_n, bit_width=codes.bit_width,
                      overflow=codes.overflow)


def weight_codes(w2d: torch.Tensor, ternary: bool = False, ld_bytes: Optional[int] = None) -> CodePlanes:
    """int8 codes of safeSign(w) (or the ternary quantiser) for a [N, K] weight."""
    _require(w2d, "weight")
    w2 = _as_rows(w2d)
    rows, K = int(w2.shape[0]), int(w2.shape[1])
    ld = code_ld_bytes(K) if ld_bytes is None else int(ld_bytes)
    codes = torch.empty((rows, ld), dtype=torch.int8, device=w2d.device)
    with _on(w2d.device):
        _lib.call("qt_weight_codes_i8", _p(w2), int(w2.stride(0) if rows > 1 else max(K, 1)),
                  _p(codes), int(ld), int(rows), int(K),
                  int(1 if ternary else 0), _stream(w2d.device))
    return CodePlanes(codes=codes, rows=rows, K=K)


def dorefa_weight_codes(wq2d: torch.Tensor, bit_width: int, ld_bytes: Optional[int] = None) -> CodePlanes:
    """int8 codes c = rint((2^k-1) * w_q) of an already quantised k-bit DoReFa weight image w_q
    (nnQuantWeight.forward, functions/dorefa_connect.py:108-112: w_q = 2*quantize_k(...) - 1, an odd multiple of
    1/(2^k-1) up to fp32 rounding, so rint recovers the integer level exactly).  2 <= k <= 7 (|c| <= 127)."""
    if not 2 <= int(bit_width) <= 7:
        raise ValueError("k-bit weight codes fit int8 for 2 <= bit_width <= 7")
    cp, _ = dorefa_codes(wq2d, bit_width, want_f32=False, ld_bytes=ld_bytes)
    cp.overflow = None          # |c| <= 2^k - 1 <= 127 by construction
    return cp


def pack_conv_weight_dorefa_codes(wq: torch.Tensor, bit_width: int) -> CodePlanes:
    """[Cout, Cin, kh, kw] quantised k-bit DoReFa weight -> int8 code plane in the conv kernels' tap-major layout
    (see pack_conv_weight_codes)."""
    _require(wq, "weight")
    Cout, Cin, kh, kw = (int(v) for v in wq.shape)
    Cb = code_ld_bytes(Cin, 16)
    wt = wq.permute(0, 2, 3, 1).contiguous().view(Cout * kh * kw, Cin)
    taps = dorefa_weight_codes(wt, bit_width, ld_bytes=Cb)
    kbytes = kh * kw * Cb
    ld = code_ld_bytes(kbytes, 512 if kbytes >= 2048 else 128)      # whole 512-byte stages for long K (skinny conv tiles)
    codes = taps.codes.view(Cout, kbytes)
    if ld != kbytes:
        padded = torch.zeros((Cout, ld), dtype=torch.int8, device=wq.device)
        padded[:, :kbytes] = codes
        codes = padded
    return CodePlanes(codes=codes, rows=Cout, K=kbytes, bit_width=int(bit_width))


def i8_gemm(x: CodePlanes, w: CodePlanes, scale: float, bias: Optional[torch.Tensor] = None,
            out: Optional[torch.Tensor] = None, max_abs_code: int = 127,
            scale_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Y[M,N] = scale * scale_dev * (x codes . w codes^T) + bias on the int8 matrix cores (exact int32
    accumulate).  ``scale_dev``: optional fp32 device scalar factor (no host sync)."""
    if x.K != w.K:
        raise ValueError(f"K mismatch: activations {x.K} vs weights {w.K}")
    M, N, K = x.rows, w.rows, x.K
    dev = x.device
    bias = _check_bias(bias, N, dev)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.call("qt_i8_gemm", _p(x.codes), int(x.ld_words), _p(w.codes), int(w.ld_words),
                  _p(bias), float(float(scale)),
                  _p(_require(scale_dev, "scale_dev").reshape(1) if scale_dev is not None else None),
                  int(int(max_abs_code)), _p(out),
                  int(out.stride(0) if M > 1 else max(N, 1)), int(M), int(N),
                  int(K), _stream(dev))
    return out


def pack_conv_weight_codes(weight: torch.Tensor, ternary: bool = False) -> CodePlanes:
    """[Cout, Cin, kh, kw] -> int8 code plane [Cout, kh*kw*Cb] (tap-major, Cb = Cin rounded to 16 B),
    row stride padded to a whole GEMM stage."""
    _require(weight, "weight")
    Cout, Cin, kh, kw = (int(v) for v in weight.shape)
    Cb = code_ld_bytes(Cin, 16)
    kbytes = kh * kw * Cb
    ld = code_ld_bytes(kbytes, 512 if kbytes >= 2048 else 128)      # whole 512-byte stages for long K (skinny conv tiles)
    if weight.dtype == torch.float32 and weight.numel() > 0:
        codes = torch.empty((Cout, ld), dtype=torch.int8, device=weight.device)
        with _on(weight.device):                                    # one pass over the weight where it lies (any strides)
            _lib.call("qt_pack_conv_weight_codes_i8", _p(weight), *(int(v) for v in weight.stride()), Cout, Cin, kh, kw,
                      int(bool(ternary)), _p(codes), int(ld),
                      _stream(weight.device))
        return CodePlanes(codes=codes, rows=Cout, K=kbytes)
    wt = weight.permute(0, 2, 3, 1).contiguous().view(Cout * kh * kw, Cin)
    taps = weight_codes(wt, ternary, ld_bytes=Cb)
    codes = taps.codes.view(Cout, kbytes)
    if ld != kbytes:
        padded = torch.zeros((Cout, ld), dtype=torch.int8, device=weight.device)
        padded[:, :kbytes] = codes
        codes = padded
    return CodePlanes(codes=codes, rows=Cout, K=kbytes)


def conv2d_codes(pixels: CodePlanes, in_shape, wplanes: CodePlanes, kernel_hw, scale: float, bias=None,
                 stride=1, padding=0, dilation=1, scale_dev=None, max_abs_code: int = 127, epi=None,
                 in_halo=(0, 0)):
    """DoReFa conv2d on int8 code planes: NHWC pixel codes -> packed-domain im2col (zero bytes for
    padding taps = the reference's zero padding, code 0 <-> value 0) -> int8 MFMA GEMM.
    Returns the NHWC result [N*Ho*Wo, Cout]; with ``epi`` (a CodeEpilogue) the CodePlanes of the fused
    BatchNorm / residual / ReLU / quantiser chain instead (implicit-GEMM kernel only)."""
    N, C, H, W = (int(v) for v in in_shape)
    kh, kw = kernel_hw
    (sh, sw), (ph, pw), (dh, dw) = _pairs(stride), _pairs(padding), _pairs(dilation)
    Ho, Wo = conv_out_hw(H, W, kh, kw, stride, padding, dilation)
    Cw = pixels.ld_words
    hy, hx = (int(v) for v in in_halo)
    npix = N * (H + 2 * hy) * (W + 2 * hx)
    if pixels.rows != npix or int(pixels.codes.shape[0]) < npix:
        raise ValueError(f"pixel plane holds {pixels.rows} pixels, in_shape {tuple(in_shape)} (halo {(hy, hx)}) needs {npix}")
    if wplanes.K != kh * kw * Cw * 4:
        raise ValueError("weight codes do not match the activation's channel packing")
    Cout, ldA = wplanes.rows, wplanes.ld_words
    M = N * Ho * Wo
    dev = pixels.device
    bias = _check_bias(bias, Cout, dev)
    if isinstance(epi, BnEpilogue):
        # conv -> BatchNorm(eval, device arithmetic) -> fp32: one launch where the implicit kernel takes the shape, else two passes
        y = None
        if _cfg("CONV_IMPLICIT") and max_abs_code * kh * kw * Cw * 4 < (1 << 31) and not (_cfg("PAD_PIXEL_PLANES") and (ph or pw) and not (hy or hx)):
            y = _conv_implicit(1, pixels.codes, N, H, W, Cw, kh, kw, ((sh, sw), (ph, pw), (dh, dw)), wplanes.codes,
                               ldA, bias, scale, scale_dev, Cout, epi=epi, in_halo=(hy, hx))
        if y is None:
            y = bn_eval_device(conv2d_codes(pixels, in_shape, wplanes, kernel_hw, scale, bias, stride, padding, dilation, scale_dev,
                                            max_abs_code, None, in_halo), epi.weight, epi.bias, epi.stats)
        return y
    if hy or hx:
        y = None
        if _cfg("CONV_IMPLICIT") and max_abs_code * kh * kw * Cw * 4 < (1 << 31):
            y = _conv_implicit(1, pixels.codes, N, H, W, Cw, kh, kw, ((sh, sw), (ph, pw), (dh, dw)), wplanes.codes,
                               ldA, bias, scale, scale_dev, Cout, epi=epi, in_halo=(hy, hx))
        if y is not None:
            return y
        # padding larger than the halo / plane beyond the un-padded kernels' limits: drop the halo (one copy)
        inner = pixels.codes.view(N, H + 2 * hy, W + 2 * hx, -1)[:, hy:hy + H, hx:hx + W].contiguous()
        pixels = CodePlanes(codes=inner.view(N * H * W, -1), rows=N * H * W, K=pixels.K, inv_n=pixels.inv_n,
                            bit_width=pixels.bit_width, overflow=pixels.overflow)
    if _cfg("CONV_IMPLICIT") and max_abs_code * kh * kw * Cw * 4 < (1 << 31):
        if (isinstance(epi, CodeEpilogue) and (kh, kw, sh, sw, ph, pw, dh, dw) == (3, 3, 1, 1, 1, 1, 1, 1) and Cw * 4 in (64, 128)
                and Cout % 64 == 0 and bias is None and epi.res_f32 is None
                and H & (H - 1) == 0 and W & (W - 1) == 0 and W <= 128 and (N * H * W) % 128 == 0):
            # (the geometry test mirrors qt_code_conv3x3_try's own: where the direct kernel would decline — 56 x 56 / 28 x 28 maps,
            # odd batches — the pad pass below would be paid for nothing over the bounds-checked un-padded form: ADVICE r5)
            # a chain's FIRST conv (its input is the code tag of an nnDorefaQuant result: no halo yet) in the shape class of the direct
            # 3 x 3 kernel: one pass makes the zero border physical (7 us at 256 x 64 x 32 x 32) and the conv runs on the halo plane
            # like every later one (38.9 -> 22 us there; the bounds-checked implicit form is what it replaces).  Exact either way.
            padded = pad_pixel_plane(pixels.codes, N, H, W, (1, 1))
            y = _conv_implicit(1, padded, N, H, W, Cw, kh, kw, ((sh, sw), (ph, pw), (dh, dw)), wplanes.codes,
                               ldA, bias, scale, scale_dev, Cout, epi=epi, in_halo=(1, 1))
            if y is not None:
                return y
        pc_, H_, W_, pad_ = pixels.codes, H, W, (ph, pw)
        if _cfg("PAD_PIXEL_PLANES") and (ph or pw):
            pc_, H_, W_, pad_ = pad_pixel_plane(pixels.codes, N, H, W, (ph, pw)), H + 2 * ph, W + 2 * pw, (0, 0)
        y = _conv_implicit(1, pc_, N, H_, W_, Cw, kh, kw, ((sh, sw), pad_, (dh, dw)), wplanes.codes,
                           ldA, bias, scale, scale_dev, Cout, epi=epi)
        if y is not None:
            return y
    if epi is not None:
        raise ValueError("the code epilogue exists on the implicit-GEMM conv kernel only (shape outside its limits)")
    y = torch.empty((M, Cout), dtype=torch.float32, device=dev)
    rows_per_chunk = max(1, min(M, IM2COL_MAX_BYTES // (ldA * 4)))
    A = torch.empty((rows_per_chunk, ldA * 4), dtype=torch.int8, device=dev)
    I = int
    for m0 in range(0, M, rows_per_chunk):
        cnt = min(rows_per_chunk, M - m0)
        with _on(dev):
            _lib.call("qt_im2col_words", _p(pixels.codes), I(N), I(H), I(W), I(Cw), I(kh), I(kw), I(sh),
                      I(sw), I(ph), I(pw), I(dh), I(dw), _p(A), I(ldA), I(m0), I(cnt), _stream(dev))
        i8_gemm(CodePlanes(codes=A[:cnt], rows=cnt, K=wplanes.K), wplanes, scale, bias, out=y[m0:m0 + cnt],
                scale_dev=scale_dev, max_abs_code=max_abs_code)
    return y


# ----------------------------------------------------------------------------------------------
# quantised conv2d = NHWC pixel planes -> packed-domain im2col -> packed GEMM
# ----------------------------------------------------------------------------------------------

def pixel_ld_nib(C: int) -> int:
    """Words per pixel of an NHWC nibble pixel plane: ceil(C/8) rounded up to 4 (16-byte chunks)."""
    return max(4, ((int(C) + 7) // 8 + 3) // 4 * 4)


def _pairs(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def conv_out_hw(H, W, kh, kw, stride, padding, dilation):
    (sh, sw), (ph, pw), (dh, dw) = _pairs(stride), _pairs(padding), _pairs(dilation)
    return (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1, (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1


def pack_conv_weight_nib(weight: torch.Tensor, kind: str, cw: Optional[int] = None) -> NibPlanes:
    """[Cout, Cin, kh, kw] fp32 -> nibble plane [Cout, kh*kw*Cw] (tap-major, channels inside a tap,
    Cw words per tap = pixel_ld_nib(Cin), or ``cw``), row stride padded to a whole GEMM stage.  ``kind``: "binary" (safeSign),
    "ternary", or "sign" (torch.sign: the XNOR-Net weight image)."""
    _require(weight, "weight")
    Cout, Cin, kh, kw = (int(v) for v in weight.shape)
    Cw = pixel_ld_nib(Cin) if cw is None else int(cw)
    wt = weight.permute(0, 2, 3, 1).contiguous().view(Cout * kh * kw, Cin)   # plumbing (weights are small)
    taps = {"binary": sign_pack_nib, "ternary": ternary_pack_nib, "sign": sign0_pack_nib}[kind](wt, ld=Cw)
    kwords = kh * kw * Cw
    ld = max(32, (kwords + 127) // 128 * 128 if kwords * 4 >= 2048 else (kwords + 31) // 32 * 32)   # whole 512-byte stages for long K
    words = taps.words.view(Cout, kwords)
    if ld != kwords:
        padded = torch.zeros((Cout, ld), dtype=torch.int32, device=weight.device)
        padded[:, :kwords] = words
        words = padded
    return NibPlanes(words=words, rows=Cout, K=kwords * 8)


def pack_pixels_nib(x: torch.Tensor, ld: Optional[int] = None) -> NibPlanes:
    """+-1 activation [N, C, H, W] (any memory format) -> NHWC nibble pixel plane [N*H*W, Cw] (Cw = pixel_ld_nib(C) or ``ld``)."""
    _require(x, "input")
    N, C, H, W = (int(v) for v in x.shape)
    nhwc = x.permute(0, 2, 3, 1)
    if not nhwc.is_contiguous():
        nhwc = nhwc.contiguous()          # NCHW storage: one transpose copy (plumbing)
    return sign_pack_nib(nhwc.view(N * H * W, C), ld=pixel_ld_nib(C) if ld is None else int(ld))




#: True: padded convs on packed pixel planes first make the zero padding physical (qt_pad_pixel_plane: one pass over a
#: plane that is 1/8 .. 1/4 of the fp32 tensor) and then run the un-padded conv kernels (32-bit offsets, no per-tap
#: checks).  Measured neutral on the module-by-module networks (tools/ab_pad_planes.py: AlexNet-Bin 3.21 -> 3.19 ms,
#: VGG-16 6.51 -> 6.44 ms, DoReFa ResNet-18 3.59 -> 3.63 ms), so off by default; the fused inference path pads inside
#: its own pack kernel (qt_bits_to_nib_pad) where it is free.
PAD_PIXEL_PLANES = False


def pad_pixel_plane(words: torch.Tensor, N: int, H: int, W: int, padding) -> torch.Tensor:
    """[N*H*W, Cw] pixel words (any element type) -> [N*(H+2ph)*(W+2pw), Cw] with a zero border."""
    ph, pw = _pairs(padding)
    ld = int(words.shape[1]) * words.element_size() // 4          # row stride in 32-bit words
    out = torch.empty((N * (H + 2 * ph) * (W + 2 * pw), words.shape[1]), dtype=words.dtype, device=words.device)
    I = int
    with _on(words.device):
        _lib.call("qt_pad_pixel_plane", _p(words), I(N), I(H), I(W), I(ld), I(ph), I(pw), _p(out), _stream(words.device))
    return out


#: 3x3 / stride-1 / padding-1 convs of <= 128-channel nibble planes that carry a 1-pixel halo run in the direct form
#: (qt_conv3x3_direct_nib: the tile's input patch is loaded once instead of gathered per tap; tools/bench_direct_conv.py,
#: batch 256: 64->64 at 224^2 620 -> 366 us, 64->128 at 112^2 246 -> 167 us, 128->128 at 112^2 328 -> 281 us)
DIRECT_CONV3X3 = True
#: 128 -> 128 channels with a bit-plane epilogue (VGG conv2_2, pooled) on the direct kernel as well
DIRECT_BITS_128 = True


def direct_conv3x3_applicable(C: int, Cout: int, kernel_hw, stride, padding, dilation, in_halo, epi) -> bool:
    if not (DIRECT_CONV3X3 and tuple(kernel_hw) == (3, 3) and _pairs(stride) == (1, 1) and _pairs(padding) == (1, 1)
            and _pairs(dilation) == (1, 1) and tuple(in_halo) == (1, 1) and pixel_ld_nib(C) in (8, 16) and Cout <= 128):
        return False
    if isinstance(epi, NibEpilogue):
        return tuple(epi.out_halo) == (1, 1) and not epi.d2s_cout
    # bit-plane output with 128 input channels: only the whole-tile 128 -> 128 shape, which has the lean epilogue (round 6)
    return isinstance(epi, tuple) and len(epi) in (2, 3) and (pixel_ld_nib(C) == 8 or (_cfg("DIRECT_BITS_128") and Cout == 128))


def conv3x3_direct_nib(pixels, N: int, C: int, H: int, W: int, wplanes, bias, epi):
    """Direct 3x3 conv of a halo-1 pixel plane with the threshold epilogue; returns BitPlanes ([N*H*W] rows) for
    ``epi`` = (alpha, beta[, thr]) or the halo-1 NibPlanes of the next conv for a NibEpilogue.  ``pixels`` / ``wplanes``:
    NibPlanes (+-1 activations) or TriplePlanes (real-valued first layer, <= 5 channels)."""
    real = isinstance(pixels, TriplePlanes)
    pairs = real and pixels.terms == 2
    if real:
        Cw = triple_ld_bytes(C, 16, pixels.terms) // 4
        words, ldw_words, Cout = pixels.data, wplanes.ld_words, wplanes.rows
        if Cw != (4 if pairs else 8) or int(pixels.data.shape[1]) * 2 != Cw * 4 or ldw_words < 9 * Cw or wplanes.terms != pixels.terms:
            raise ValueError("direct first-layer conv expects 32-byte triple pixels (<= 5 channels) or 16-byte pair pixels (<= 4)")
        if pairs and pixels.scale is None:
            raise ValueError("fp16 pair planes carry their power-of-two scale")
        wwords = wplanes.data
    else:
        Cw = pixel_ld_nib(C)
        words, wwords, ldw_words, Cout = pixels.words, wplanes.words, wplanes.ld, wplanes.rows
        if pixels.ld != Cw:
            raise ValueError("direct conv expects the [N, H+2, W+2] halo plane of the activation")
        if wplanes.K != 9 * Cw * 8:
            raise ValueError("weight planes do not match the activation's channel packing")
    if pixels.rows != N * (H + 2) * (W + 2):
        raise ValueError("direct conv expects the [N, H+2, W+2] halo plane of the activation")
    dev = words.device
    nib_out = isinstance(epi, NibEpilogue)
    alpha, beta = (epi.alpha, epi.beta) if nib_out else epi[:2]
    alpha, beta, bias = _check_bias(alpha, Cout, dev), _check_bias(beta, Cout, dev), _check_bias(bias, Cout, dev)
    if nib_out:
        ldo = pixel_ld_nib(Cout)
        out = torch.empty((N * (H + 2) * (W + 2), ldo), dtype=torch.int32, device=dev)
    else:
        ldo = packed_ld(Cout)
        out = torch.empty((N * H * W, ldo), dtype=torch.int32, device=dev)
    with _on(dev):
        if pairs:
            _lib.call("qt_conv3x3_direct_pairs", _p(words), int(N), int(H), int(W), _p(wwords), int(ldw_words), _p(bias),
                      _p(pixels.scale[0:1]), _p(alpha), _p(beta), _p(out), int(ldo), int(Cout), 0 if nib_out else 1, _stream(dev))
        else:
            _lib.call("qt_conv3x3_direct_nib", 2 if real else 0, _p(words), int(N), int(H), int(W), int(Cw), _p(wwords),
                      int(ldw_words), _p(bias), _p(alpha), _p(beta), _p(out), int(ldo), int(Cout), 0 if nib_out else 1,
                      _stream(dev))
    if nib_out:
        return NibPlanes(words=out, rows=int(out.shape[0]), K=Cout)
    return BitPlanes(sign=out, rows=N * H * W, K=Cout)


#: the int8 (DoReFa code plane) variant of the direct kernel is bit-identical but not faster than the implicit GEMM at the
#: shapes that fit it (ResNet-18 stage 1, 64 -> 64 at 32^2, batch 256: 42.6 vs 41 us: 4.5 tiles per CU at one 8-wave
#: workgroup per CU), so it is not dispatched
DIRECT_CONV3X3_CODES = False


def direct_conv3x3_codes_applicable(C: int, Cout: int, kernel_hw, stride, padding, dilation, in_halo, epi) -> bool:
    """DoReFa code planes: 3x3 / stride 1 / padding 1, 64 input channels, <= 64 output channels, halo 1 in and out, residual
    (if any) as codes with the same halo."""
    return (DIRECT_CONV3X3_CODES and isinstance(epi, CodeEpilogue) and tuple(kernel_hw) == (3, 3) and _pairs(stride) == (1, 1)
            and _pairs(padding) == (1, 1) and _pairs(dilation) == (1, 1) and tuple(in_halo) == (1, 1)
            and tuple(epi.out_halo) == (1, 1) and code_ld_bytes(C, 16) == 64 and Cout <= 64 and epi.res_f32 is None
            and (epi.res_codes is None or tuple(epi.res_halo) == (1, 1)))


def conv3x3_direct_codes(pixels: CodePlanes, N: int, C: int, H: int, W: int, wplanes: CodePlanes, scale: float, bias,
                         scale_dev, epi: CodeEpilogue) -> CodePlanes:
    """Direct 3x3 conv of a halo-1 int8 code plane with the DoReFa code epilogue -> halo-1 code plane."""
    Cw = int(pixels.codes.shape[1]) // 4
    rows = N * (H + 2) * (W + 2)
    if pixels.rows != rows or Cw != 16 or wplanes.K != 9 * Cw * 4:
        raise ValueError("direct code conv expects the [N, H+2, W+2] halo plane of a 64-channel activation")
    Cout = wplanes.rows
    dev = pixels.device
    alpha, beta, bias = _check_bias(epi.alpha, Cout, dev), _check_bias(epi.beta, Cout, dev), _check_bias(bias, Cout, dev)
    rc, ldrc, rscale = None, 0, 0.0
    if epi.res_codes is not None:
        if epi.res_codes.rows != rows or epi.res_codes.K != Cout:
            raise ValueError("residual codes must be a halo-1 plane of the output's geometry")
        rc, ldrc, rscale = epi.res_codes.codes, int(epi.res_codes.codes.shape[1]), float(epi.res_codes.inv_n)
    ldc = code_ld_bytes(Cout, 16)
    codes = torch.empty((rows, ldc), dtype=torch.int8, device=dev)
    flag = epi.overflow if epi.overflow is not None else torch.zeros((1,), dtype=torch.int32, device=dev)
    with _on(dev):
        _lib.call("qt_conv3x3_direct_codes", _p(pixels.codes), int(N), int(H), int(W), int(Cw), _p(wplanes.codes),
                  int(wplanes.ld_words), _p(bias), float(scale),
                  _p(_require(scale_dev, "scale_dev").reshape(1) if scale_dev is not None else None), _p(alpha), _p(beta),
                  _p(rc), int(ldrc), float(rscale), relu_mode(epi.relu), int(epi.bit_width), _p(codes), int(ldc), int(Cout),
                  _p(flag), _stream(dev))
    return CodePlanes(codes=codes, rows=rows, K=Cout, inv_n=inv_levels(epi.bit_width), bit_width=int(epi.bit_width),
                      overflow=flag)


def direct_first_layer_terms(Cin: int) -> int:
    """Split of the image the direct 3x3 first-layer kernel runs with: fp16 pairs (16-byte pixels: <= 4 channels, the two lane halves
    of an MFMA take two taps) when FLOAT_SPLIT allows two terms, else the exact bf16 triples (32-byte pixels: <= 5 channels)."""
    return 2 if (split_terms(None) == 2 and int(Cin) <= 4) else 3


def direct_first_layer_applicable(Cin: int, Cout: int, kernel_hw, stride, padding, dilation) -> bool:
    """Real-valued 3x3 / stride-1 / padding-1 first layer with 32-byte bf16-triple pixels (<= 5 channels): the direct
    kernel on the physically padded triple plane (s2d_triple_pack(x, 1, 1))."""
    return (DIRECT_CONV3X3 and tuple(kernel_hw) == (3, 3) and _pairs(stride) == (1, 1) and _pairs(padding) == (1, 1)
            and _pairs(dilation) == (1, 1) and triple_ld_bytes(Cin, 16) == 32 and Cout <= 128)


def zero_halo(words: torch.Tensor, N: int, H: int, W: int, halo) -> torch.Tensor:
    """Zero (in place) the border pixels of a halo plane [N*(H+2hy)*(W+2hx), Cw] of any element type; for callers that
    fill the interior themselves (the conv / pooling entry points write their own borders)."""
    hy, hx = _pairs(halo)
    ld = int(words.shape[1]) * words.element_size() // 4
    if int(words.shape[0]) != N * (H + 2 * hy) * (W + 2 * hx):
        raise ValueError("plane does not hold an [N, H + 2hy, W + 2hx] image")
    with _on(words.device):
        _lib.call("qt_zero_halo", _p(words), int(N), int(H), int(W), ld, hy, hx, _stream(words.device))
    return words


def conv2d_nib(pixels: NibPlanes, in_shape, wplanes: NibPlanes, kernel_hw, bias=None, stride=1,
               padding=0, dilation=1, epi=None):
    """Quantised conv2d on packed operands.  pixels: NHWC nibble pixel plane of the +-1 activation
    (shape ``in_shape`` = (N, C, H, W)); wplanes: pack_conv_weight_nib(...).  Returns the NHWC result
    as a [N*Ho*Wo, Cout] fp32 matrix."""
    N, C, H, W = (int(v) for v in in_shape)
    kh, kw = kernel_hw
    (sh, sw), (ph, pw), (dh, dw) = _pairs(stride), _pairs(padding), _pairs(dilation)
    Ho, Wo = conv_out_hw(H, W, kh, kw, stride, padding, dilation)
    Cw = pixels.ld
    if pixels.rows != N * H * W or int(pixels.words.shape[0]) < N * H * W:
        raise ValueError(f"pixel plane holds {pixels.rows} pixels, in_shape {tuple(in_shape)} needs {N * H * W}")
    if wplanes.K != kh * kw * Cw * 8:
        raise ValueError("weight planes do not match the activation's channel packing")
    Cout, ldA = wplanes.rows, wplanes.ld
    M = N * Ho * Wo
    dev = pixels.device
    bias = _check_bias(bias, Cout, dev)
    if _cfg("CONV_IMPLICIT"):
        pw_, H_, W_, pad_ = pixels.words, H, W, (ph, pw)
        if _cfg("PAD_PIXEL_PLANES") and (ph or pw):
            pw_, H_, W_, pad_ = pad_pixel_plane(pixels.words, N, H, W, (ph, pw)), H + 2 * ph, W + 2 * pw, (0, 0)
        y = _conv_implicit(0, pw_, N, H_, W_, Cw, kh, kw, ((sh, sw), pad_, (dh, dw)), wplanes.words,
                           ldA, bias, 1.0, None, Cout, epi=epi)
        if y is not None:
            return y
    if epi is not None:
        raise ValueError("the threshold-bit epilogue needs the implicit-GEMM conv (shape outside its limits)")
    y = torch.empty((M, Cout), dtype=torch.float32, device=dev)
    rows_per_chunk = max(1, min(M, IM2COL_MAX_BYTES // (ldA * 4)))
    A = torch.empty((rows_per_chunk, ldA), dtype=torch.int32, device=dev)
    I = int
    for m0 in range(0, M, rows_per_chunk):
        cnt = min(rows_per_chunk, M - m0)
        with _on(dev):
            _lib.call("qt_im2col_words", _p(pixels.words), I(N), I(H), I(W), I(Cw), I(kh), I(kw), I(sh),
                      I(sw), I(ph), I(pw), I(dh), I(dw), _p(A), I(ldA), I(m0), I(cnt), _stream(dev))
        nib_gemm(NibPlanes(words=A[:cnt], rows=cnt, K=wplanes.K), wplanes, bias, out=y[m0:m0 + cnt])
    return y


# ----------------------------------------------------------------------------------------------
# per-tap scaled convs: the XNOR-Net family (csrc/conv_taps.hip)
# ----------------------------------------------------------------------------------------------

def pixel_ld_nib_taps(C: int) -> int:
    """Words per pixel of a nibble pixel plane that feeds a per-tap scaled conv: a tap must be whole 32-byte MFMA k-steps, so
    ceil(C / 8) rounded up to 8 words (64 channels)."""
    return max(8, ((int(C) + 7) // 8 + 7) // 8 * 8)


@dataclass
class TapScales:
    """Per-tap scales of an XNOR-Net conv weight (alpha = mean(|W|, [0, 1]) -> [kh * kw], functions/xnor_connect.py:140) and the
    Horner tables the kernels read (qt_xnor_tap_prep_f32): ``tables`` = [forward (taps + 1) | flipped (taps + 1)]."""
    alpha: torch.Tensor
    tables: torch.Tensor
    taps: int

    @property
    def fwd(self) -> torch.Tensor:
        return self.tables[:self.taps + 1]

    @property
    def bwd(self) -> torch.Tensor:
        return self.tables[self.taps + 1:]


def xnor_tap_prep(weight: Optional[torch.Tensor] = None, alpha: Optional[torch.Tensor] = None) -> TapScales:
    """alpha[kh * kw] of a conv weight [Cout, Cin, kh, kw] (mean of |W| over the first two dimensions) and its Horner tables, on
    the device without a host round trip.  ``alpha`` given (eval mode: the weight already holds sign(W) * alpha): tables only."""
    if alpha is not None:
        a = _require(alpha.detach(), "alpha").contiguous().view(-1)
        T = int(a.numel())
        tables = torch.empty((2 * (T + 1),), dtype=torch.float32, device=a.device)
        with _on(a.device):
            _lib.call("qt_xnor_tap_prep_f32", None, 0, T, _p(a), None, None, _p(tables), _stream(a.device))
        return TapScales(alpha=a, tables=tables, taps=T)
    w = _require(weight.detach(), "weight").contiguous()
    Cout, Cin, kh, kw = (int(v) for v in w.shape)
    R, T = Cout * Cin, kh * kw
    if T > 1024 or R == 0:
        raise ValueError("xnor_tap_prep: at most 1024 taps and a non-empty weight")
    work = torch.empty((max(1, int(_lib.load().qt_xnor_tap_prep_work_floats(R, T))),), dtype=torch.float32, device=w.device)
    a = torch.empty((T,), dtype=torch.float32, device=w.device)
    tables = torch.empty((2 * (T + 1),), dtype=torch.float32, device=w.device)
    with _on(w.device):
        _lib.call("qt_xnor_tap_prep_f32", _p(w), R, T, None, _p(work), _p(a), _p(tables), _stream(w.device))
    return TapScales(alpha=a, tables=tables, taps=T)


def _conv_taps(elem: int, pixels_words: torch.Tensor, N, H, W, Cw, kh, kw, geom, wmat: torch.Tensor, ldw_words: int, bias,
               scale: float, scale_dev, tap_rho: torch.Tensor, Cout: int, epi=None):
    """qt_conv2d_implicit_taps (fp32 result), _bits (``epi`` = (alpha, beta)) or _nib (a NibEpilogue); None when the shape is
    outside the kernel's limits."""
    (sh, sw), (ph, pw), (dh, dw) = geom
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    M = N * Ho * Wo
    if (M >= (1 << 31) or kh * kw * Cw * 4 >= (1 << 20) or H > 32767 or W > 32767 or ldw_words % 32 or Cw % 8
            or H * W * Cw * 4 >= (1 << 31) or Cout * ldw_words * 4 >= (1 << 31)):
        return None
    dev = pixels_words.device
    rho = _require(tap_rho, "tap_rho")
    if rho.numel() != kh * kw + 1 or not rho.is_contiguous():
        raise ValueError(f"tap_rho must hold kh * kw + 1 = {kh * kw + 1} contiguous entries")
    I = int
    head = (int(elem), _p(pixels_words), I(N), I(H), I(W), I(Cw), I(kh), I(kw), I(sh), I(sw), I(ph), I(pw),
            I(dh), I(dw), _p(wmat), I(ldw_words), _p(bias), float(float(scale)),
            _p(_require(scale_dev, "scale_dev").reshape(1) if scale_dev is not None else None), _p(rho))
    if isinstance(epi, NibEpilogue):
        if epi.d2s_cout:
            raise ValueError("the depth-to-space epilogue does not exist for per-tap scaled convs")
        alpha, beta = _check_bias(epi.alpha, Cout, dev), _check_bias(epi.beta, Cout, dev)
        ohy, ohx = (int(v) for v in epi.out_halo)
        ldn = pixel_ld_nib(Cout)
        rows = N * (Ho + 2 * ohy) * (Wo + 2 * ohx)
        plane = torch.empty((rows, ldn), dtype=torch.int32, device=dev)
        with _on(dev):
            _lib.call("qt_conv2d_implicit_taps_nib", *head, _p(alpha), _p(beta), _p(plane), I(ldn), I(Cout), I(ohy), I(ohx),
                      _stream(dev))
        return NibPlanes(words=plane, rows=rows, K=Cout)
    if epi is not None:
        alpha, beta = _check_bias(epi[0], Cout, dev), _check_bias(epi[1], Cout, dev)
        ldb = packed_ld(Cout)
        plane = torch.empty((M, ldb), dtype=torch.int32, device=dev)
        with _on(dev):
            _lib.call("qt_conv2d_implicit_taps_bits", *head, _p(alpha), _p(beta), _p(plane), I(ldb), I(Cout), _stream(dev))
        return BitPlanes(sign=plane, rows=M, K=Cout)
    y = torch.empty((M, Cout), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.call("qt_conv2d_implicit_taps", *head, _p(y), I(Cout), I(Cout), _stream(dev))
    return y


def conv2d_nib_taps(pixels: NibPlanes, in_shape, wplanes: NibPlanes, kernel_hw, tap_rho: torch.Tensor, bias=None, stride=1,
                    padding=0, dilation=1, epi=None):
    """XNORConv2d on packed operands (functions/xnor_connect.py:145 for +-1 activations): NHWC nibble pixel plane of the
    activation (row stride pixel_ld_nib_taps(C)) x nibble plane of sign(W) (pack_conv_weight_nib(w, "sign", cw = that stride)),
    alpha per tap through ``tap_rho`` (TapScales.fwd).  Returns the NHWC result [N*Ho*Wo, Cout] fp32 (or the epilogue's
    planes), None when the shape is outside the kernel's limits."""
    N, C, H, W = (int(v) for v in in_shape)
    kh, kw = kernel_hw
    Cw = pixels.ld
    if pixels.rows != N * H * W or int(pixels.words.shape[0]) < N * H * W:
        raise ValueError(f"pixel plane holds {pixels.rows} pixels, in_shape {tuple(in_shape)} needs {N * H * W}")
    if Cw % 8:
        raise ValueError("per-tap scaled convs take pixel planes with whole 32-byte taps (pixel_ld_nib_taps)")
    if wplanes.K != kh * kw * Cw * 8:
        raise ValueError("weight planes do not match the activation's channel packing")
    Cout = wplanes.rows
    bias = _check_bias(bias, Cout, pixels.device)
    return _conv_taps(0, pixels.words, N, H, W, Cw, kh, kw, (_pairs(stride), _pairs(padding), _pairs(dilation)), wplanes.words,
                      wplanes.ld, bias, 1.0, None, tap_rho, Cout, epi=epi)


def conv2d_real_taps(x: torch.Tensor, weight: torch.Tensor, tap_rho: torch.Tensor, bias=None, stride=1, padding=0, dilation=1,
                     weight_planes: Optional["TriplePlanes"] = None):
    """conv2d(x, sign(W) * alpha[1, 1, kh, kw]) for a REAL-valued x (XNORConv2d with quant_input=True: x = sign(.) * per-pixel
    scale, functions/xnor_connect.py:142-145): the two-term fp16 split of x against the +-1 / 0 image of sign(W) on the fp16 matrix
    cores, alpha applied per tap on the accumulators (qt_conv2d_implicit_taps, elem 3; ``tap_rho`` = TapScales.fwd) — two fp16
    passes instead of the six bf16 passes of a real x real conv.  Cin % 8 == 0 (a tap of the pair plane = whole 32-byte k-steps).
    Returns the NHWC result [N*Ho*Wo, Cout], or None when the shape is outside the kernel's limits."""
    _require(x, "input")
    Cout, Cin, kh, kw = (int(v) for v in weight.shape)
    N, C, H, W = (int(v) for v in x.shape)
    if C != Cin or Cin % 8 or x.numel() == 0:
        return None
    Cb = triple_ld_bytes(Cin, 16, 2)
    nhwc = x.detach().permute(0, 2, 3, 1)
    if not nhwc.is_contiguous():
        nhwc = nhwc.contiguous()
    px = split_bf16x3(nhwc.view(N * H * W, Cin), ld_bytes=Cb, terms=2)
    wt = weight_planes if weight_planes is not None else pack_conv_weight_bf16x3(weight.detach(), "sign", terms=2)
    bias = _check_bias(bias, Cout, x.device)
    return _conv_taps(3, px.data, N, H, W, Cb // 4, kh, kw, (_pairs(stride), _pairs(padding), _pairs(dilation)), wt.data, wt.ld_words,
                      bias, 1.0, px.scale[0:1], tap_rho, Cout)


def alpha_pairs(alpha: torch.Tensor) -> "TriplePlanes":
    """Two-term fp16 pair image of a per-feature scale row alpha[K] (LinearXNOR, xnor_connect.py:112): TriplePlanes [1, K] with
    its power-of-two scale, the table qt_bits_alpha_pairs_f16x2 reads."""
    a = _require(alpha.detach(), "alpha").contiguous().view(1, -1)
    return split_bf16x3(a, terms=2)


def bits_alpha_pairs(planes: BitPlanes, apairs: "TriplePlanes", hwc=None) -> "TriplePlanes":
    """fp16 pair plane of x[b, k] * alpha[k] for a packed +-1 activation (row bit planes): the pairs of alpha with the
    activation's signs (qt_bits_alpha_pairs_f16x2).  ``hwc`` = (C, H, W): the bit rows are a feature map flattened in (h, w, c)
    order; the pairs come out in the NCHW order the layer's weight (and ``apairs``) count in."""
    if planes.mask is not None or planes.K != apairs.K:
        raise ValueError("bits_alpha_pairs takes sign-only row planes and the pair image of a [K] scale row")
    pc, phw = (int(hwc[0]), int(hwc[1]) * int(hwc[2])) if hwc is not None else (0, 0)
    ld = triple_ld_bytes(planes.K, terms=2)
    out = torch.empty((planes.rows, ld // 2), dtype=torch.int16, device=planes.device)
    with _on(planes.device):
        _lib.call("qt_bits_alpha_pairs_f16x2", _p(planes.sign), int(planes.ld), _p(apairs.data), _p(out), int(ld),
                  int(planes.rows), int(planes.K), pc, phw, _stream(planes.device))
    return TriplePlanes(data=out, rows=planes.rows, K=planes.K, terms=2, scale=apairs.scale)


@dataclass
class AlphaDigits:
    """Fixed-point image of a per-feature scale row alpha[K] >= 0 (LinearXNOR, functions/xnor_connect.py:112):
    A[k] = rint(alpha[k] / s) < 2^21 as three 7-bit digits, table[k] = d0 | d1 << 8 | d2 << 16 with A = d0 2^14 + d1 2^7 + d2
    (int32, one entry per byte of a padded code-plane row, zero from K on); ``scale`` = the power of two s as a device fp32 [1]."""
    table: torch.Tensor
    scale: torch.Tensor
    K: int


def alpha_digits(alpha: torch.Tensor) -> Optional[AlphaDigits]:
    """The digit table of a scale row, or None when alpha has a non-finite entry (one host sync; once per weight version)."""
    a = alpha.detach()                               # (plain tensor arithmetic on K values: any device)
    if a.dtype != torch.float32:
        raise TypeError(f"alpha: expected dtype torch.float32, got {a.dtype}")
    a = a.contiguous().view(-1)
    mx = a.amax() if a.numel() else a.new_zeros(())
    _, e = torch.frexp(mx)                                   # max < 2^e
    s = torch.ldexp(torch.ones_like(mx), e - 21)             # max / s < 2^21
    s = torch.where((mx > 0) & (s > 0), s, torch.ones_like(mx))
    q = a / s
    if not bool(torch.isfinite(q).all().item()) or bool((a < 0).any().item()):
        return None
    A = torch.clamp(torch.round(q), 0, float((1 << 21) - 1)).to(torch.int32)
    table = torch.zeros((code_ld_bytes(int(a.numel())),), dtype=torch.int32, device=a.device)      # zero up to the padded row length
    table[:a.numel()] = (A >> 14) | (((A >> 7) & 127) << 8) | ((A & 127) << 16)
    return AlphaDigits(table=table, scale=s.reshape(1).to(torch.float32).contiguous(), K=int(a.numel()))


def _pick_tile_n(N: int) -> int:
    """csrc/mfma_gemm_kernel.h pick_tile_n: the tile width (256 / 192 / 128 / 64) with the fewest padded columns."""
    best, best_pad = 256, (N + 255) // 256 * 256
    for c in (192, 128, 64):
        pad = (N + c - 1) // c * c
        if pad < best_pad:
            best, best_pad = c, pad
    return best


def splitk_plan(M: int, N: int, ld_bytes: int, min_slice_bytes: int = 512, workgroups: int = 256):
    """(kslice, nslice) for qt_i8_gemm_splitk: the largest number of equal K slices (whole 64-byte stages, at least
    ``min_slice_bytes`` each, dividing the padded row) whose workgroups still fit one round of the chip."""
    tiles = ((M + 255) // 256) * ((N + _pick_tile_n(N) - 1) // _pick_tile_n(N))
    units = max(1, ld_bytes // 64)
    best = 1
    for d in range(1, units + 1):
        if units % d == 0 and tiles * d <= workgroups and (units // d) * 64 >= min_slice_bytes:
            best = d
    return (units // best) * 64, best


def bits_alpha_digits(planes: BitPlanes, digits: AlphaDigits, hwc=None, ld_bytes: Optional[int] = None) -> CodePlanes:
    """Three stacked int8 planes [3 * rows, ld] of x[b, k] * d_j[k] for a packed +-1 activation (qt_bits_alpha_digits_i8)."""
    if planes.mask is not None or planes.K != digits.K:
        raise ValueError("bits_alpha_digits takes sign-only row planes and the digit table of a [K] scale row")
    pc, phw = (int(hwc[0]), int(hwc[1]) * int(hwc[2])) if hwc is not None else (0, 0)
    ld = code_ld_bytes(planes.K) if ld_bytes is None else int(ld_bytes)
    if digits.table.numel() < ld:
        raise ValueError(f"digit table holds {digits.table.numel()} entries, the planes' rows {ld}")
    out = torch.empty((3 * planes.rows, ld), dtype=torch.int8, device=planes.device)
    with _on(planes.device):
        _lib.call("qt_bits_alpha_digits_i8", _p(planes.sign), int(planes.ld), _p(digits.table), _p(out), int(ld),
                  int(planes.rows), int(planes.K), pc, phw, _stream(planes.device))
    return CodePlanes(codes=out, rows=3 * planes.rows, K=planes.K)


#: output features up to which LinearXNOR's digit route runs as one streaming launch (qt_xnor_head_i8) instead of the split-K GEMM
XNOR_HEAD_MAX_N = 32


def xnor_digit_linear(planes: BitPlanes, digits: AlphaDigits, wcodes: CodePlanes, bias: Optional[torch.Tensor] = None,
                      hwc=None) -> torch.Tensor:
    """y = (x * alpha) . sign(W)^T + b for a packed +-1 activation x (functions/xnor_connect.py:112-115) in the integer form:
    digit planes of alpha against the int8 codes of sign(W) in one split-K int8 GEMM (exact partial sums), combined in fp64."""
    rows, K, N = planes.rows, planes.K, wcodes.rows
    if wcodes.K != K:
        raise ValueError(f"K mismatch: activations {K} vs weights {wcodes.K}")
    dev = planes.device
    bias = _check_bias(bias, N, dev)
    ld = int(wcodes.codes.shape[1])
    if 127 * ld >= (1 << 24) or (ld & 127):
        raise ValueError("digit-plane GEMM: K beyond the exact fp32 range of the partial sums, or an unpadded weight plane")
    if N <= XNOR_HEAD_MAX_N and K < (1 << 16):
        # classifier heads: the same exact integer sum in one launch from the sign bits (bit-identical to the GEMM form below)
        pc, phw = (int(hwc[0]), int(hwc[1]) * int(hwc[2])) if hwc is not None else (0, 0)
        y = torch.empty((rows, N), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.call("qt_xnor_head_i8", _p(planes.sign), int(planes.ld), _p(digits.table), _p(wcodes.codes), int(ld), _p(digits.scale),
                      _p(bias), _p(y), int(N), int(rows), int(N), int(K), pc, phw, _stream(dev))
        return y
    x3 = bits_alpha_digits(planes, digits, hwc=hwc, ld_bytes=ld)
    kslice, nslice = splitk_plan(3 * rows, N, ld)
    ldp = (N + 3) // 4 * 4
    part = torch.empty((nslice, 3 * rows, ldp), dtype=torch.float32, device=dev)
    y = torch.empty((rows, N), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.call("qt_i8_gemm_splitk", _p(x3.codes), int(ld // 4), _p(wcodes.codes), int(ld // 4), _p(part), int(ldp), int(3 * rows),
                  int(N), int(kslice), int(nslice), int(3 * rows * ldp), _stream(dev))
        _lib.call("qt_digit_reduce_f32", _p(part), int(ldp), int(3 * rows * ldp), int(nslice), _p(digits.scale), _p(bias), _p(y),
                  int(N), int(rows), int(N), _stream(dev))
    return y


# ----------------------------------------------------------------------------------------------
# real-valued activation x quantised weight: exact bf16 triples + bf16 MFMA GEMM
# ----------------------------------------------------------------------------------------------

_TRIPLE_MODES = {"binary": 1, "ternary": 2, "sign": 3, "raw": 4}

#: How a real-valued fp32 operand is split for the matrix cores:
#:   "f16x2"  two fp16 terms of x / s, s a per-tensor power of two found on the device (csrc/split_f16.hip):
#:            |x - s (hi + lo)| <= max(2^-22 |x|, 2^-39 max|x|); 2/3 of the MFMA work and operand bytes of the exact route;
#:   "bf16x3" three bf16 terms, exact (csrc/split_bf16.hip).
#: Both meet the 1e-5 normalised bar of SURVEY 8(d) for real-valued inputs ("two-term split"); routes
#: whose kernels only exist for triples (direct 3x3 first layer, swapped-conv weight gradient, XNOR alpha folding) pass
#: terms=3 explicitly.
FLOAT_SPLIT = "f16x2"


_split_tls = threading.local()


def current_float_split() -> str:
    """The split in force in THIS thread: the innermost ``with float_split(...)``, else the default FLOAT_SPLIT."""
    return getattr(_split_tls, "mode", None) or FLOAT_SPLIT


def float_split_override() -> Optional[str]:
    """The mode an enclosing ``with float_split(...)`` of this thread set, or None (what QtFunction.forward captures)."""
    return getattr(_split_tls, "mode", None)


@contextlib.contextmanager
def float_split(mode: Optional[str]):
    """Run the enclosed real-valued contractions with the split ``mode`` ("f16x2" | "bf16x3"; None = leave as is).  THREAD-LOCAL:
    two serving threads (one process, one stream per device: SURVEY 8e) can hold different modes at the same time without seeing
    each other's.  The backward of an autograd graph runs on the engine's own thread, which cannot see the forward thread's
    override — so every autograd.Function of the package records the override at forward time and re-opens it around its backward
    (functions.common.QtFunction), i.e. a graph is differentiated under the mode it was built under — a scope opened only around
    ``loss.backward()`` is NOT seen by a graph that was built outside it (set ``ops.FLOAT_SPLIT`` for that).  Library code never opens a
    scope: layers that need the exact route pass ``terms=3`` to float_linear / float_conv2d explicitly (Lin / Log layers)."""
    if mode is not None and mode not in ("f16x2", "bf16x3"):
        raise ValueError(f"FLOAT_SPLIT must be 'f16x2' or 'bf16x3', got {mode!r}")
    prev = getattr(_split_tls, "mode", None)
    if mode is not None:
        _split_tls.mode = mode
    try:
        yield
    finally:
        _split_tls.mode = prev


def split_terms(terms: Optional[int] = None) -> int:
    if terms is not None:
        return int(terms)
    mode = current_float_split()
    if mode not in ("f16x2", "bf16x3"):
        raise ValueError(f"FLOAT_SPLIT must be 'f16x2' or 'bf16x3', got {mode!r}")
    return 2 if mode == "f16x2" else 3


def triple_ld_bytes(K: int, granule: int = 128, terms: int = 3) -> int:
    """Row stride in bytes of a bf16 triple (terms = 3: 6 bytes per feature) or fp16 pair (terms = 2: 4 bytes) plane."""
    return max(granule, (2 * int(terms) * int(K) + granule - 1) // granule * granule)


@dataclass
class TriplePlanes:
    """Split image of a [rows, K] fp32 matrix for the bf16 / fp16 matrix cores: int16 tensor [rows, ld_bytes/2].
    terms = 3: bf16, element 3k+t is term t of x[k] = hi + mid + lo (activations) or the quantised weight value replicated
    (weights).  terms = 2: fp16, element 2k+t is term t of x[k] / scale[0] (activations; ``scale`` = device fp32 [s, 1/s])
    or the weight value replicated twice."""
    data: torch.Tensor
    rows: int
    K: int
    terms: int = 3
    scale: Optional[torch.Tensor] = None
    #: (device scalar m, one-element view holding scale[0] * m) when the pack was asked to fold a consumer's device scale in
    #: (``_triple_pack(mul_dev=)``): the consumer passes the view as its scale_dev instead of launching scale[0:1] * m
    scale_mul: Optional[tuple] = None

    @property
    def ld_words(self) -> int:
        return int(self.data.shape[1]) // 2

    def scale_dev_with(self, mul: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """One-element device tensor scale[0] * mul (mul None: scale[0:1]; no scale: mul)."""
        if mul is None:
            return self.scale[0:1] if self.scale is not None else None
        m = mul.detach().reshape(1)
        if self.scale is None:
            return m
        if self.scale_mul is not None and self.scale_mul[0] is mul:
            return self.scale_mul[1]
        return self.scale[0:1] * m                                     # (the split's scale is a power of two: exact)

    @property
    def device(self):
        return self.data.device

    @property
    def elem(self) -> int:
        """Element code of the conv / GEMM entry points: 2 = bf16 triples, 3 = fp16 pairs."""
        return 2 if self.terms == 3 else 3


_ABS_MEAN_WORK = {}      # (device index, stream handle) -> zero-initialised ticket / partial buffer (the kernel leaves it zero)


def abs_mean(x: torch.Tensor) -> torch.Tensor:
    """mean|x| as a 0-dim tensor: ``torch.mean(torch.abs(x))`` (functions/dorefa_connect.py:100, DoReFa's E).  Dense fp32 device
    tensors: one launch (qt_abs_mean_f32: double-precision, order-fixed fold) instead of torch's abs + mean (+ fill); every
    device-side E of the package comes from here, so training mode, the eval swap and the backward see the same bits."""
    x = x.detach()
    if not (x.is_cuda and x.dtype == torch.float32 and x.numel() > 0 and _storage_dense(x) and x.data_ptr() % 16 == 0):
        return torch.mean(torch.abs(x))
    dev = x.device
    st = _stream(dev)
    key = (dev.index, int(st) if st is not None else 0)
    work = _ABS_MEAN_WORK.get(key)
    if work is None:
        if len(_ABS_MEAN_WORK) > 64:
            _ABS_MEAN_WORK.clear()
        words = int(_lib.load().qt_abs_mean_work_words())
        with _on(dev):
            work = torch.zeros((words + 2,), dtype=torch.int32, device=dev)
        if work.data_ptr() % 8:
            work = work[1:]
        _ABS_MEAN_WORK[key] = work
    out = torch.empty((), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.call("qt_abs_mean_f32", _p(x), int(x.numel()), _p(work), _p(out), st)
    return out


def pow2_scale(x: torch.Tensor) -> torch.Tensor:
    """Device fp32 [s, 1 / s] with s = 2^k and max|x| / s in [2^14, 2^15) (qt_f16x2_scale_f32): one reduction pass over x
    (torch.aminmax), no host sync."""
    x = x.detach()
    out = torch.empty((2,), dtype=torch.float32, device=x.device)
    if _storage_dense(x) and x.data_ptr() % 16 == 0:
        # dense storage in any dimension order: max|x| at the HBM rate and the scale in one launch (torch.aminmax runs at
        # ~1.2 TB/s: 128 us for AlexNet's 154 MB input batch against 26 us)
        work = torch.empty((2048,), dtype=torch.int32, device=x.device)        # qt_f16x2_absmax_work_words()
        with _on(x.device):
            _lib.call("qt_f16x2_absmax_scale_f32", _p(x), int(x.numel()), _p(work), _p(out), _stream(x.device))
        return out
    mn, mx = torch.aminmax(x)
    with _on(x.device):
        _lib.call("qt_f16x2_scale_f32", _p(mn), _p(mx), _p(out), _stream(x.device))
    return out


def _triple_pack(x: torch.Tensor, mode: int, alpha: Optional[torch.Tensor], ld_bytes: Optional[int],
                 terms: Optional[int] = None, scale: Optional[torch.Tensor] = None, mul_dev: Optional[torch.Tensor] = None) -> TriplePlanes:
    _require(x, "input")
    x2 = _as_rows(x)
    rows, K = int(x2.shape[0]), int(x2.shape[1])
    terms = 3 if alpha is not None else split_terms(terms)        # the per-feature alpha is folded by the triple kernel only
    ld = triple_ld_bytes(K, terms=terms) if ld_bytes is None else int(ld_bytes)
    out = torch.empty((rows, ld // 2), dtype=torch.int16, device=x.device)
    if terms == 2:
        if (mode == 0 and scale is None and rows > 0 and K > 0 and x2.is_contiguous() and x2.data_ptr() % 16 == 0
                and (mul_dev is None or (mul_dev.is_cuda and mul_dev.dtype == torch.float32 and mul_dev.numel() == 1))):
            # max|x| partials + (fold, scale, split) in two launches; the consumer's other device scalar rides in scale3[2]
            ws = torch.empty((2048 + 4,), dtype=torch.int32, device=x.device)      # qt_f16x2_absmax_work_words() + the three scales
            scale3 = ws[2048:2051].view(torch.float32)
            md = mul_dev.detach() if mul_dev is not None else None
            with _on(x.device):
                _lib.call("qt_f16x2_absmax_pack_f32", _p(x2), int(rows), int(K), _p(ws), _p(md), _p(scale3), _p(out), int(ld),
                          _stream(x.device))
            return TriplePlanes(data=out, rows=rows, K=K, terms=2, scale=scale3[0:2],
                                scale_mul=(mul_dev, scale3[2:3]) if mul_dev is not None else None)
        if mode == 0 and scale is None:
            scale = pow2_scale(x2)
        with _on(x.device):
            _lib.call("qt_f16x2_pack_f32", _p(x2), int(x2.stride(0) if rows > 1 else max(K, 1)),
                      _p(scale if mode == 0 else None), _p(out), int(ld), int(rows), int(K), int(mode), _stream(x.device))
        return TriplePlanes(data=out, rows=rows, K=K, terms=2, scale=scale if mode == 0 else None)
    if alpha is not None:
        alpha = _require(alpha, "alpha").contiguous().view(-1)
        if alpha.numel() != K:
            raise ValueError("alpha must have one entry per input feature")
    with _on(x.device):
        _lib.call("qt_bf16x3_pack_f32", _p(x2), int(x2.stride(0) if rows > 1 else max(K, 1)),
                  _p(alpha), _p(out), int(ld), int(rows), int(K),
                  int(mode), _stream(x.device))
    return TriplePlanes(data=out, rows=rows, K=K)


def split_bf16x3(x: torch.Tensor, alpha: Optional[torch.Tensor] = None, ld_bytes: Optional[int] = None,
                 terms: Optional[int] = None, mul_dev: Optional[torch.Tensor] = None) -> TriplePlanes:
    """Split of an fp32 activation (optionally of x * alpha[k]) for the matrix cores: exact hi / mid / lo bf16 triples, or
    (terms = 2 / FLOAT_SPLIT = "f16x2") scaled fp16 pairs.  ``mul_dev``: a one-element device tensor the consumer multiplies its
    contraction by (TriplePlanes.scale_dev_with(mul_dev) is then free)."""
    return _triple_pack(x, 0, alpha, ld_bytes, terms, mul_dev=mul_dev)


def weight_bf16x3(w2d: torch.Tensor, kind: str, ld_bytes: Optional[int] = None, terms: Optional[int] = None) -> TriplePlanes:
    """Replicated image of the quantised weight: kind 'binary' (safeSign), 'ternary', 'sign' (torch.sign), 'raw' (the value
    itself; exact for the integer levels it is used with), as bf16 x 3 or fp16 x 2."""
    return _triple_pack(w2d, _TRIPLE_MODES[kind], None, ld_bytes, terms)


def bf16_gemm(x: TriplePlanes, w: TriplePlanes, bias: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if x.K != w.K:
        raise ValueError(f"K mismatch: activations {x.K} vs weights {w.K}")
    M, N = x.rows, w.rows
    dev = x.device
    bias = _check_bias(bias, N, dev)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
    if x.terms != w.terms:
        raise ValueError(f"operand planes disagree: {x.terms}-term activation vs {w.terms}-term weight")
    with _on(dev):
        if x.terms == 2:
            _lib.call("qt_f16_gemm", _p(x.data), int(x.ld_words), _p(w.data), int(w.ld_words), _p(bias), 1.0,
                      _p(x.scale[0:1] if x.scale is not None else None), _p(out),
                      int(out.stride(0) if M > 1 else max(N, 1)), int(M), int(N), int(2 * x.K), _stream(dev))
        else:
            _lib.call("qt_bf16_gemm", _p(x.data), int(x.ld_words), _p(w.data), int(w.ld_words),
                      _p(bias), _p(out), int(out.stride(0) if M > 1 else max(N, 1)), int(M),
                      int(N), int(3 * x.K), _stream(dev))
    return out


# ---- real x real: six-term planes ----------------------------------------------------------------------------

def sext_ld_bytes(K: int, granule: int = 128) -> int:
    """Row stride in bytes of a six-term plane holding K features (12 bytes each)."""
    return max(granule, (12 * int(K) + granule - 1) // granule * granule)


def split_bf16x6(x: torch.Tensor, role: int, ld_bytes: Optional[int] = None) -> TriplePlanes:
    """Six-term bf16 plane of an fp32 matrix: role 0 = activation order, 1 = weight order (qt_bf16x6_pack_f32).
    Returned as TriplePlanes with K = 2 * features, so that 3*K is the bf16 length of a row."""
    _require(x, "input")
    x2 = _as_rows(x)
    rows, K = int(x2.shape[0]), int(x2.shape[1])
    ld = sext_ld_bytes(K) if ld_bytes is None else int(ld_bytes)
    out = torch.empty((rows, ld // 2), dtype=torch.int16, device=x.device)
    with _on(x.device):
        _lib.call("qt_bf16x6_pack_f32", _p(x2), int(x2.stride(0) if rows > 1 else max(K, 1)), _p(out),
                  int(ld), int(rows), int(K), int(int(role)), _stream(x.device))
    return TriplePlanes(data=out, rows=rows, K=2 * K)


def real_linear(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    """y = x . weight^T (+ bias) for REAL x and REAL weight on the bf16 matrix cores (six-term planes): fp32-GEMM
    accuracy at 6 MFMA products per multiply-accumulate."""
    N = weight.shape[0]
    y = bf16_gemm(split_bf16x6(x, 0), split_bf16x6(weight.reshape(N, -1), 1), bias)
    return y.view(*x.shape[:-1], N)


def pack_conv_weight_bf16x6(weight: torch.Tensor) -> TriplePlanes:
    """[Cout, Cin, kh, kw] REAL weight -> six-term plane [Cout, kh*kw*Cb/2] (tap-major; Cb = 12*Cin bytes rounded to 16)."""
    _require(weight, "weight")
    Cout, Cin, kh, kw = (int(v) for v in weight.shape)
    Cb = sext_ld_bytes(Cin, 16)
    wt = weight.permute(0, 2, 3, 1).contiguous().view(Cout * kh * kw, Cin)
    taps = split_bf16x6(wt, 1, ld_bytes=Cb)
    kbytes = kh * kw * Cb
    ld = max(128, (kbytes + 127) // 128 * 128)
    data = taps.data.view(Cout, kbytes // 2)
    if ld != kbytes:
        padded = torch.zeros((Cout, ld // 2), dtype=torch.int16, device=weight.device)
        padded[:, :kbytes // 2] = data
        data = padded
    return TriplePlanes(data=data, rows=Cout, K=kbytes // 6)


def real_conv2d(x: torch.Tensor, weight: torch.Tensor, bias=None, stride=1, padding=0, dilation=1,
                weight_planes: Optional[TriplePlanes] = None, epi=None) -> Optional[torch.Tensor]:
    """conv2d(x, weight) for REAL x and REAL weight (groups = 1, zero padding) as an implicit GEMM over six-term
    planes on the bf16 matrix cores.  Returns the NHWC result [N*Ho*Wo, Cout], or None if the shape is outside the
    implicit kernel's limits (caller falls back to the dense library)."""
    _require(x, "input")
    N, C, H, W = (int(v) for v in x.shape)
    Cout, _, kh, kw = (int(v) for v in weight.shape)
    (sh, sw), (ph, pw), (dh, dw) = _pairs(stride), _pairs(padding), _pairs(dilation)
    Cb = sext_ld_bytes(C, 16)
    nhwc = x.permute(0, 2, 3, 1)
    if not nhwc.is_contiguous():
        nhwc = nhwc.contiguous()
    px = split_bf16x6(nhwc.view(N * H * W, C), 0, ld_bytes=Cb)
    wt = weight_planes if weight_planes is not None else pack_conv_weight_bf16x6(weight)
    bias = _check_bias(bias, Cout, x.device)
    return _conv_implicit(2, px.data, N, H, W, Cb // 4, kh, kw, ((sh, sw), (ph, pw), (dh, dw)), wt.data, wt.ld_words,
                          bias, 1.0, None, Cout, epi=epi)


def float_linear(x: torch.Tensor, weight: torch.Tensor, kind: str, bias=None, alpha=None,
                 weight_triples: Optional[TriplePlanes] = None, terms: Optional[int] = None) -> torch.Tensor:
    """y = x . Q(weight)^T (+ bias) for REAL-valued x: Q in {safeSign, ternary, torch.sign}; ``alpha``
    (per input feature) multiplies x first (XNORDense).  fp32-GEMM accuracy on the bf16 matrix cores.  ``terms``: the split
    of x (default FLOAT_SPLIT; 3 = exact, for an x whose ROWS are separate quantities of very different magnitude)."""
    N = weight.shape[0]
    xp = split_bf16x3(x, alpha, terms=terms)
    wt = weight_triples if (weight_triples is not None and weight_triples.terms == xp.terms) else \
        weight_bf16x3(weight.reshape(N, -1), kind, terms=xp.terms)
    y = bf16_gemm(xp, wt, bias)
    return y.view(*x.shape[:-1], N)


def pack_conv_weight_bf16x3(weight: torch.Tensor, kind: str, terms: Optional[int] = None, transpose_flip: bool = False) -> TriplePlanes:
    """[Cout, Cin, kh, kw] -> triple / pair plane [Cout, kh*kw*Cb/2] (tap-major; Cb = 6*Cin or 4*Cin bytes rounded to 16).
    ``transpose_flip``: the operand of grad_x instead — the plane of weight.flip(2, 3).transpose(0, 1), [Cin, kh*kw*Cb'/2].
    The fp16 pair form comes from ONE kernel that reads the weight where it lies (qt_f16x2_pack_conv_weight_f32: quantiser,
    flip / transpose, tap-major layout and row padding)."""
    _require(weight, "weight")
    terms = split_terms(terms)
    if terms == 2 and weight.dtype == torch.float32 and weight.dim() == 4 and weight.numel() > 0:
        Cout, Cin, kh, kw = (int(v) for v in weight.shape)
        rows, chans = (Cin, Cout) if transpose_flip else (Cout, Cin)
        Cb = triple_ld_bytes(chans, 16, 2)
        kbytes = kh * kw * Cb
        ld = max(128, (kbytes + 127) // 128 * 128)
        data = torch.empty((rows, ld // 2), dtype=torch.int16, device=weight.device)
        with _on(weight.device):
            _lib.call("qt_f16x2_pack_conv_weight_f32", _p(weight), *(int(v) for v in weight.stride()), Cout, Cin, kh, kw,
                      int(_TRIPLE_MODES[kind]), int(bool(transpose_flip)),
                      _p(data), int(ld), _stream(weight.device))
        return TriplePlanes(data=data, rows=rows, K=kbytes // 4, terms=2)
    if transpose_flip:
        weight = weight.detach().flip(2, 3).transpose(0, 1).contiguous()
    Cout, Cin, kh, kw = (int(v) for v in weight.shape)
    Cb = triple_ld_bytes(Cin, 16, terms)
    wt = weight.permute(0, 2, 3, 1).contiguous().view(Cout * kh * kw, Cin)
    taps = weight_bf16x3(wt, kind, ld_bytes=Cb, terms=terms)
    kbytes = kh * kw * Cb
    ld = max(128, (kbytes + 127) // 128 * 128)
    data = taps.data.view(Cout, kbytes // 2)
    if ld != kbytes:
        padded = torch.zeros((Cout, ld // 2), dtype=torch.int16, device=weight.device)
        padded[:, :kbytes // 2] = data
        data = padded
    return TriplePlanes(data=data, rows=Cout, K=kbytes // (2 * terms), terms=terms)   # K only used for consistency checks


#: fixed power-of-two scale the first layer's space-to-depth pack speculates with (None / 0: always the separate max|x| pass).
#: 2^-11 is admissible for max|x| in [2^-3, 2^4) (max|x| / s in [2^8, 2^15): below fp16's overflow with a binade to spare):
#: unit-variance and [0, 1] images; anything else is repacked on the device.
S2D_SPEC_SCALE: Optional[float] = 2.0 ** -11


def s2d_triple_pack(x: torch.Tensor, s: int, padding, terms: Optional[int] = None) -> Tuple[TriplePlanes, Tuple[int, int]]:
    """Space-to-depth gather + split (exact bf16 triples, or scaled fp16 pairs) of [N, C, H, W] (any storage) in one kernel.
    Returns (pixel planes with rows = N*Hs*Ws and K = C*s*s, (Hs, Ws))."""
    _require(x, "input")
    N, C, H, W = (int(v) for v in x.shape)
    ph, pw = _pairs(padding)
    Hs, Ws = (H + 2 * ph + s - 1) // s, (W + 2 * pw + s - 1) // s
    E = C * s * s
    terms = split_terms(terms)
    ld = triple_ld_bytes(E, 16, terms)
    out = torch.empty((N * Hs * Ws, ld // 2), dtype=torch.int16, device=x.device)
    I = int
    sN, sC, sH, sW = (int(v) for v in x.stride())
    if terms == 2:
        rowf4 = (W * C + 3) & ~3
        if (S2D_SPEC_SCALE and sC == 1 and sW == C and sH >= W * C and int(s) * rowf4 * 4 + E * 8 <= 60 * 1024
                and N * Hs < (1 << 31) and N > 0):
            # channels-last image: packed with a fixed scale while max|x| is folded on the way; the device decides whether the
            # plane stands or is rewritten with the exact-binade scale (csrc/split_f16.hip): no separate pass over the image
            scale = torch.empty((2,), dtype=torch.float32, device=x.device)
            work = torch.empty((N * Hs + 1,), dtype=torch.int32, device=x.device)        # [partial maxima | redo flag]
            with _on(x.device):
                _lib.call("qt_f16x2_s2d_pack_spec_f32", _p(x), I(sN), I(sC), I(sH), I(sW), float(S2D_SPEC_SCALE), _p(work), _p(scale),
                          _p(work[N * Hs:]), _p(out), I(ld), I(N), I(C), I(H), I(W), I(int(s)), I(ph), I(pw), _stream(x.device))
            return TriplePlanes(data=out, rows=N * Hs * Ws, K=E, terms=2, scale=scale), (Hs, Ws)
        scale = pow2_scale(x)
        with _on(x.device):
            _lib.call("qt_f16x2_s2d_pack_f32", _p(x), I(sN), I(sC), I(sH), I(sW), _p(scale), _p(out), I(ld), I(N), I(C), I(H),
                      I(W), I(int(s)), I(ph), I(pw), _stream(x.device))
        return TriplePlanes(data=out, rows=N * Hs * Ws, K=E, terms=2, scale=scale), (Hs, Ws)
    with _on(x.device):
        _lib.call("qt_bf16x3_s2d_pack_f32", _p(x), I(sN), I(sC), I(sH), I(sW), _p(out), I(ld), I(N), I(C), I(H),
                  I(W), I(int(s)), I(ph), I(pw), _stream(x.device))
    return TriplePlanes(data=out, rows=N * Hs * Ws, K=E), (Hs, Ws)


def float_conv2d(x: Optional[torch.Tensor], weight: torch.Tensor, kind: str, bias=None, stride=1, padding=0,
                 dilation=1, weight_triples: Optional[TriplePlanes] = None, pixels: Optional[TriplePlanes] = None,
                 in_shape=None, epi=None, out_scale: float = 1.0, out_scale_dev: Optional[torch.Tensor] = None,
                 terms: Optional[int] = None):
    """conv2d(x, Q(weight)) for REAL-valued x (groups = 1, zero padding): NHWC bf16 triple pixel planes ->
    implicit-GEMM conv on the bf16 matrix cores.  ``pixels``/``in_shape``: pre-built pixel planes (e.g. from
    s2d_triple_pack) instead of x.  ``out_scale`` (host float) / ``out_scale_dev`` (one-element device tensor, e.g. DoReFa's
    E = mean|W|): multiply the contraction in the kernel's epilogue, before the bias — no extra pass over the result.
    ``terms``: the split of x (default FLOAT_SPLIT; 3 = the exact bf16 route).  Returns NHWC [N*Ho*Wo, Cout]."""
    if pixels is None:
        _require(x, "input")
        N, C, H, W = (int(v) for v in x.shape)
    else:
        N, C, H, W = (int(v) for v in in_shape)
    Cout, _, kh, kw = (int(v) for v in weight.shape)
    (sh, sw), (ph, pw), (dh, dw) = _pairs(stride), _pairs(padding), _pairs(dilation)
    Ho, Wo = conv_out_hw(H, W, kh, kw, stride, padding, dilation)
    terms = pixels.terms if pixels is not None else split_terms(terms)
    Cb = triple_ld_bytes(C, 16, terms)
    if pixels is None:
        nhwc = x.permute(0, 2, 3, 1)
        if not nhwc.is_contiguous():
            nhwc = nhwc.contiguous()
        px = split_bf16x3(nhwc.view(N * H * W, C), ld_bytes=Cb, terms=terms, mul_dev=out_scale_dev)
    else:
        px = pixels
        x = pixels.data
        if pixels.rows != N * H * W or int(pixels.data.shape[0]) < N * H * W:
            raise ValueError(f"pixel plane holds {pixels.rows} pixels, in_shape {tuple(in_shape)} needs {N * H * W}")
    if weight_triples is not None and weight_triples.terms == terms:
        wt = weight_triples
    else:
        if weight.device.type == "meta":
            raise ValueError(f"cached weight planes have {weight_triples.terms} terms, the pixel planes {terms}")
        wt = pack_conv_weight_bf16x3(weight, kind, terms=terms)
    Cw, ldA = Cb // 4, wt.ld_words
    M = N * Ho * Wo
    dev = x.device
    bias = _check_bias(bias, Cout, dev)
    if out_scale_dev is not None:
        _require(out_scale_dev.detach(), "out_scale_dev")
    sdev = px.scale_dev_with(out_scale_dev)
    if _cfg("CONV_IMPLICIT"):
        y = _conv_implicit(px.elem, px.data, N, H, W, Cw, kh, kw, ((sh, sw), (ph, pw), (dh, dw)), wt.data, ldA, bias,
                           float(out_scale), sdev, Cout, epi=epi)
        if y is not None:
            return y
    if epi is not None:
        raise ValueError("the threshold-bit epilogue needs the implicit-GEMM conv (shape outside its limits)")
    y = torch.empty((M, Cout), dtype=torch.float32, device=dev)
    rows_per_chunk = max(1, min(M, IM2COL_MAX_BYTES // (ldA * 4)))
    A = torch.empty((rows_per_chunk, ldA * 2), dtype=torch.int16, device=dev)
    I = int
    kel = kh * kw * Cb // 2     # bf16 elements per im2col row actually carrying taps
    for m0 in range(0, M, rows_per_chunk):
        cnt = min(rows_per_chunk, M - m0)
        with _on(dev):
            _lib.call("qt_im2col_words", _p(px.data), I(N), I(H), I(W), I(Cw), I(kh), I(kw), I(sh), I(sw),
                      I(ph), I(pw), I(dh), I(dw), _p(A), I(ldA), I(m0), I(cnt), _stream(dev))
            if terms == 2:
                _lib.call("qt_f16_gemm", _p(A), I(ldA), _p(wt.data), I(wt.ld_words), _p(bias), float(out_scale),
                          _p(sdev), _p(y[m0:m0 + cnt]), I(Cout), I(cnt), I(Cout), I(kel), _stream(dev))
            else:
                plain = sdev is None and float(out_scale) == 1.0
                _lib.call("qt_bf16_gemm", _p(A), I(ldA), _p(wt.data), I(wt.ld_words), _p(bias if plain else None),
                          _p(y[m0:m0 + cnt]), I(Cout), I(cnt), I(Cout), I(kel), _stream(dev))
    if terms != 2 and not (sdev is None and float(out_scale) == 1.0):
        y = y * (float(out_scale) if sdev is None else sdev * float(out_scale))
        if bias is not None:
            y = y + bias
    return y


# ---- training-mode chain [MaxPool] -> BatchNorm(batch stats) -> [Hardtanh] -> sign (csrc/train_chain.hip) --------------------------

def _rows_view(x: torch.Tensor):
    """(NHWC-contiguous view of a 4-D tensor | the [N, C] matrix itself, N, H, W, C)."""
    if x.dim() == 4:
        nhwc = x.permute(0, 2, 3, 1)
        if not nhwc.is_contiguous():
            nhwc = nhwc.contiguous()
        N, H, W, C = (int(v) for v in nhwc.shape)
        return nhwc, N, H, W, C
    if x.dim() == 2:
        x2 = x.contiguous()
        return x2, int(x2.shape[0]), 1, 1, int(x2.shape[1])
    raise ValueError("the training chain takes [N, C, H, W] or [N, C] tensors")


def pool_bn_sign_train(x: torch.Tensor, gamma, beta, running_mean, running_var, eps: float, momentum: float,
                       pool_k: int = 1, pool_s: int = 1, ht=(-float("inf"), float("inf"))):
    """Forward of the training chain on a device fp32 tensor.  Returns (sign image shaped like the pooled tensor — NHWC storage
    for 4-D inputs —, saved = (p_or_x, idx, mean, invstd, geometry) for ``pool_bn_sign_train_backward``).  The running
    statistics are updated in place."""
    _require(x, "input")
    xs, N, H, W, C = _rows_view(x.detach())
    k, s = int(pool_k), int(pool_s)
    if H < k or W < k:
        raise ValueError("pooling window larger than the map")
    Ho, Wo = (H - k) // s + 1, (W - k) // s + 1
    R = N * Ho * Wo
    dev = x.device
    p = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=dev) if k > 1 else None
    idx = torch.empty((N, Ho, Wo, C), dtype=torch.int8, device=dev) if k > 1 else None
    mean = torch.empty((C,), dtype=torch.float32, device=dev)
    invstd = torch.empty((C,), dtype=torch.float32, device=dev)
    partial = torch.empty((int(_lib.load().qt_train_chain_partial_floats(R, C)),), dtype=torch.float32, device=dev)
    sgn = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=dev)
    g_ = _check_bias(gamma.detach() if gamma is not None else None, C, dev)
    b_ = _check_bias(beta.detach() if beta is not None else None, C, dev)
    with _on(dev):
        _lib.call("qt_pool_bn_sign_train_f32", _p(xs), N, H, W, C, k, s, _p(g_), _p(b_), float(eps), float(momentum),
                  float(ht[0]), float(ht[1]), _p(running_mean), _p(running_var), _p(p), _p(idx), _p(mean), _p(invstd),
                  _p(partial), _p(sgn), _stream(dev))
    out = sgn.permute(0, 3, 1, 2) if x.dim() == 4 else sgn.view(N, C)
    return out, (p if k > 1 else xs, idx, mean, invstd, (N, H, W, C, k, s))


def pool_bn_sign_train_backward(grad_out: torch.Tensor, saved, gamma, beta, ht, ste_threshold: float = STE_THRESHOLD):
    """(grad wrt the chain's input [shaped / laid out like it], dgamma, dbeta)."""
    p_or_x, idx, mean, invstd, (N, H, W, C, k, s) = saved
    g, _, Ho, Wo, _ = _rows_view(_require(grad_out, "grad_output"))
    dev = g.device
    R = N * Ho * Wo
    partial = torch.empty((int(_lib.load().qt_train_chain_partial_floats(R, C)),), dtype=torch.float32, device=dev)
    dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
    dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
    gp = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=dev)
    gx = torch.empty((N, H, W, C), dtype=torch.float32, device=dev) if k > 1 else None
    g_ = _check_bias(gamma.detach() if gamma is not None else None, C, dev)
    b_ = _check_bias(beta.detach() if beta is not None else None, C, dev)
    with _on(dev):
        _lib.call("qt_pool_bn_sign_train_backward_f32", _p(g), _p(p_or_x), _p(idx), N, H, W, C, k, s, _p(g_), _p(b_), _p(mean),
                  _p(invstd), float(ht[0]), float(ht[1]), float(ste_threshold), _p(partial), _p(dgamma), _p(dbeta), _p(gp),
                  _p(gx), _stream(dev))
    gin = gx if k > 1 else gp
    return (gin.permute(0, 3, 1, 2) if grad_out.dim() == 4 else gin.view(N, C)), dgamma, dbeta


def poison(x: Optional[torch.Tensor], flag: torch.Tensor, mask: int = -1, n: Optional[int] = None) -> torch.Tensor:
    """x (or n zeros), turned into NaN when ``flag & mask`` is set on the device (qt_poison_f32): one launch, no host sync."""
    _require(flag, "flag", torch.int32)
    if x is not None:
        x = _require(x, "input").contiguous()
        out = torch.empty_like(x)
        cnt = x.numel()
    else:
        out = torch.empty((int(n),), dtype=torch.float32, device=flag.device)
        cnt = int(n)
    with _on(flag.device):
        _lib.call("qt_poison_f32", _p(x), _p(flag), int(mask), _p(out), cnt, _stream(flag.device))
    return out


CODE_DIGIT_FLAG_BIT = 8       # bit of a chain's range flag: "a base-256 digit of a code is no longer exact in bf16" (|q| >= 2^16)


def code_digits(x: torch.Tensor, levels: float, flag: Optional[torch.Tensor] = None):
    """(hi, lo, flag) with q = rint(x * levels) = 256 hi + lo, hi = floor(q / 256) — fp32 tensors laid out like x (dense storage,
    any dimension order); ``flag`` (int32 [1], created zeroed when None) gets CODE_DIGIT_FLAG_BIT when |hi| >= 256."""
    x = _require(x.detach(), "input")
    if not _storage_dense(x):
        x = x.contiguous()
    hi, lo = torch.empty_like(x), torch.empty_like(x)
    if flag is None or flag.dtype != torch.int32:
        flag = torch.zeros((1,), dtype=torch.int32, device=x.device)
    with _on(x.device):
        _lib.call("qt_code_digits_f32", _p(x), int(x.numel()), float(levels), _p(hi), _p(lo), _p(flag), int(CODE_DIGIT_FLAG_BIT),
                  _stream(x.device))
    return hi, lo, flag


def digit_combine(g_hi: torch.Tensor, g_lo: torch.Tensor, inv: float, flag: Optional[torch.Tensor], mask: int = CODE_DIGIT_FLAG_BIT):
    """(g_hi * 256 + g_lo) * inv in fp32, NaN when ``flag & mask`` (one launch); the operands share one dense layout."""
    _require(g_hi, "g_hi")
    _require(g_lo, "g_lo")
    if g_hi.shape != g_lo.shape or g_hi.stride() != g_lo.stride() or not _storage_dense(g_hi):
        g_hi, g_lo = g_hi.contiguous(), g_lo.contiguous()
    out = torch.empty_like(g_hi)
    with _on(g_hi.device):
        _lib.call("qt_digit_combine_f32", _p(g_hi), _p(g_lo), _p(flag), int(mask), float(inv), _p(out), int(g_hi.numel()),
                  _stream(g_hi.device))
    return out


# ---- training-mode chain BatchNorm(batch stats) [+ residual] [-> ReLU] [-> nnDorefaQuant] (csrc/train_chain.hip, codes_i8.hip) -----

def bn_train_stats(x: torch.Tensor, running_mean, running_var, eps: float, momentum: float, zero_flag: Optional[torch.Tensor] = None):
    """Batch statistics of a device fp32 [N, C, H, W] / [N, C] tensor: returns (NHWC view of x, stats2 = [mean | invstd]); the
    running statistics (None: skipped) are updated in place like nn.BatchNorm2d.train() does.  ``zero_flag``: an int32 [1] tensor the
    last launch sets to 0 (the range flag of the quantiser pass that follows: saves its fill launch)."""
    _require(x, "input")
    xs, N, H, W, C = _rows_view(x.detach())
    R = N * H * W
    dev = x.device
    stats2 = torch.empty((2 * C,), dtype=torch.float32, device=dev)
    partial = torch.empty((int(_lib.load().qt_train_chain_partial_floats(R, C)),), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.call("qt_bn_train_stats_f32", _p(xs), R, C, float(eps), float(momentum), _p(running_mean), _p(running_var),
                  _p(stats2), _p(partial), _p(zero_flag), _stream(dev))
    return xs, stats2


def bn_act_train_backward(grad_out: torch.Tensor, xs: torch.Tensor, res, gamma, beta, stats2: torch.Tensor, relu: bool,
                          want_res_grad: bool):
    """Backward of quant(relu?(BatchNorm_train(x) + res)) (identity STE): (gx shaped like grad_out, dgamma, dbeta, gres or None).
    ``xs`` / ``res``: the forward's NHWC (or [N, C]) views."""
    g, N, H, W, C = _rows_view(_require(grad_out, "grad_output"))
    dev = g.device
    R = N * H * W
    partial = torch.empty((int(_lib.load().qt_train_chain_partial_floats(R, C)),), dtype=torch.float32, device=dev)
    dgamma = torch.empty((C,), dtype=torch.float32, device=dev)
    dbeta = torch.empty((C,), dtype=torch.float32, device=dev)
    gx = torch.empty_like(g)
    gres = torch.empty_like(g) if want_res_grad else None
    g_ = _check_bias(gamma.detach() if gamma is not None else None, C, dev)
    b_ = _check_bias(beta.detach() if beta is not None else None, C, dev)
    with _on(dev):
        _lib.call("qt_bn_act_train_backward_f32", _p(g), _p(xs), _p(res), R, C, _p(g_), _p(b_), _p(stats2), int(bool(relu)),
                  _p(partial), _p(dgamma), _p(dbeta), _p(gx), _p(gres), _stream(dev))
    shape = (lambda t: t.permute(0, 3, 1, 2)) if grad_out.dim() == 4 else (lambda t: t.view(N, C))
    return shape(gx), dgamma, dbeta, (shape(gres) if gres is not None else None)


# ---- backward of a quantised conv on the bf16 matrix cores ----------------------------------------------
# Both gradients of conv2d(x, Q(W)) have one +-1 / 0 operand, so the exact-split route of the forward applies:
#   grad_x = conv2d(g, flip(Q(W))^T, padding k-1-p)                 (stride 1): real g x quantised weight, the forward kernel
#   grad_W[co, ci, i, j] = sum_{n, ho, wo} g[n, co, ho, wo] x[n, ci, ho s - p + i d, wo s - p + j d]
#          = conv2d(x^T, g^T) with batch <-> channel swapped (x^T: [Cin, N, H, W], "weight" g^T: [Cout, N, Ho, Wo], stride and
#            dilation exchanged): +-1 x as the replicated bf16 operand, g as the exact hi/mid/lo triple operand, the
#            contraction over (n, ho, wo) on the matrix cores; the batch is cut into chunks that keep a row of the
#            "weight" under the kernel's 1 MiB K limit and give the launch enough tiles, partial results added in fp32.
# Reference expressions: functions/binary_connect.py:141-143 (torch.nn.grad.conv2d_input / conv2d_weight).

def zero_dilated_gradient(grad_output: torch.Tensor, input_shape, kernel_hw, stride: int, padding):
    """The gradient of a stride-s conv (square stride, un-dilated) as the operand of the equivalent STRIDE-1 transposed conv:
    g_d[.., s y, s x] = g[.., y, x], zeros elsewhere, with ``H + 2p - k - (Ho - 1) s`` rows / columns of zeros appended (the input
    rows the strided windows never reached).  Logical NCHW over NHWC memory.  None when the shapes do not belong together."""
    s = int(stride)
    kh, kw = (int(v) for v in kernel_hw)
    ph, pw = _pairs(padding)
    N, C, H, W = (int(v) for v in input_shape)
    _, Cout, Ho, Wo = (int(v) for v in grad_output.shape)
    eh, ew = H + 2 * ph - kh - (Ho - 1) * s, W + 2 * pw - kw - (Wo - 1) * s
    if not (0 <= eh < s and 0 <= ew < s):
        return None
    gd = torch.zeros((N, (Ho - 1) * s + 1 + eh, (Wo - 1) * s + 1 + ew, Cout), dtype=grad_output.dtype, device=grad_output.device)
    gd[:, 0:(Ho - 1) * s + 1:s, 0:(Wo - 1) * s + 1:s, :] = grad_output.permute(0, 2, 3, 1)
    return gd.permute(0, 3, 1, 2)


def conv2d_grad_input_q(input_shape, weight_q: torch.Tensor, grad_output: torch.Tensor, stride, padding, dilation,
                        kind: str = "sign", out_scale: float = 1.0, out_scale_dev: Optional[torch.Tensor] = None):
    """grad wrt the input of conv2d(x, Q(weight_q)): ``weight_q`` already quantised — +-1 / 0 (``kind`` "sign") or integer
    levels (``kind`` "raw": the k-bit DoReFa levels c = n * w_q, the caller scales by 1 / n) — or the latent weight with its
    quantiser (``kind`` "binary" / "ternary": applied inside the operand pack): the forward's exact-split conv
    of the gradient with the flipped, transposed weight.  Stride 1 directly; stride s > 1 (square, un-dilated — the
    3x3 / stride-2 and 1x1 / stride-2 convs of the reference's ResNets, models/Resnet/Resnet_bin.py:20-33) through the
    zero-dilated gradient: g_d[.., s y, s x] = g[.., y, x], zeros elsewhere and ``H + 2p - k - (Ho - 1) s`` rows / columns of
    zeros appended, which turns conv_transpose(g, W, stride s) into the stride-1 conv above (3/4 of its products are with the
    inserted zeros; these layers are the small ones).  ``out_scale`` / ``out_scale_dev``: see ``float_conv2d`` (1 / n of the
    levels, DoReFa's E).  None when the shape is outside the route (dilation != 1, padding > k - 1, non-square stride): the
    caller uses torch.nn.grad.conv2d_input."""
    (sh, sw), (ph, pw), (dh, dw) = _pairs(stride), _pairs(padding), _pairs(dilation)
    Cout, Cin, kh, kw = (int(v) for v in weight_q.shape)
    if (dh, dw) != (1, 1) or sh != sw or sh < 1 or ph > kh - 1 or pw > kw - 1:
        return None
    N, C, H, W = (int(v) for v in input_shape)
    g = grad_output
    if sh > 1:
        g = zero_dilated_gradient(g, input_shape, (kh, kw), sh, (ph, pw))
        if g is None:
            return None
    # the flipped, transposed weight [Cin, Cout, kh, kw] only ever exists as the conv's packed operand
    wt = pack_conv_weight_bf16x3(weight_q.detach(), kind, transpose_flip=True)
    shape_t = torch.empty((Cin, Cout, kh, kw), dtype=torch.float32, device="meta")
    y2 = float_conv2d(g, shape_t, kind, None, 1, (kh - 1 - ph, kw - 1 - pw), 1, weight_triples=wt, out_scale=out_scale,
                      out_scale_dev=out_scale_dev)
    return y2.view(N, H, W, C).permute(0, 3, 1, 2)


def conv2d_grad_input_taps(input_shape, weight: torch.Tensor, grad_output: torch.Tensor, tap_rho_flipped: torch.Tensor, stride,
                           padding, dilation):
    """grad wrt the input of an XNOR-Net conv, conv2d(x, sign(W) * alpha[1, 1, kh, kw]) (functions/xnor_connect.py:154-155):
    the two-term fp16 split of the gradient against the flipped, transposed sign(W) on the fp16 matrix cores, alpha applied per
    (flipped) tap on the accumulators (qt_conv2d_implicit_taps, elem 3; ``tap_rho_flipped`` = TapScales.bwd).  Square stride
    (s > 1 through the zero-dilated gradient), un-dilated, padding <= k - 1, Cout % 8 == 0 (a tap of the pair plane = whole 32-byte
    k-steps); None otherwise."""
    (sh, sw), (ph, pw), (dh, dw) = _pairs(stride), _pairs(padding), _pairs(dilation)
    Cout, Cin, kh, kw = (int(v) for v in weight.shape)
    if sh != sw or sh < 1 or (dh, dw) != (1, 1) or ph > kh - 1 or pw > kw - 1 or Cout % 8:
        return None
    N, C, H, W = (int(v) for v in input_shape)
    g = _require(grad_output, "grad_output")
    if sh > 1:                                     # square stride: the zero-dilated gradient (see conv2d_grad_input_q)
        g = zero_dilated_gradient(g, input_shape, (kh, kw), sh, (ph, pw))
        if g is None:
            return None
    _, _, Ho, Wo = (int(v) for v in g.shape)
    wt = pack_conv_weight_bf16x3(weight.detach(), "sign", terms=2, transpose_flip=True)        # [Cin, kh*kw*Cb/2] fp16 pairs
    Cb = triple_ld_bytes(Cout, 16, 2)
    nhwc = g.permute(0, 2, 3, 1)
    if not nhwc.is_contiguous():
        nhwc = nhwc.contiguous()
    px = split_bf16x3(nhwc.view(N * Ho * Wo, Cout), ld_bytes=Cb, terms=2)
    y2 = _conv_taps(3, px.data, N, Ho, Wo, Cb // 4, kh, kw, ((1, 1), (kh - 1 - ph, kw - 1 - pw), (1, 1)), wt.data, wt.ld_words,
                    None, 1.0, px.scale[0:1], tap_rho_flipped, Cin)
    if y2 is None:
        return None
    return y2.view(N, H, W, C).permute(0, 3, 1, 2)


@functools.lru_cache(maxsize=64)
def _inv_f32(levels: float) -> float:
    """fl(1 / levels) in fp32 arithmetic (1 for levels == 1): the factor the reduce kernels apply for a k-bit activation q / n."""
    if levels == 1.0:
        return 1.0
    return float(torch.tensor(1.0, dtype=torch.float32) / torch.tensor(float(levels), dtype=torch.float32))


_TAP_GATHER = {}


def _tap_gather_index(dev, s: int, kh: int, kw: int, p: int):
    """Device index tensors (a_h, a_w, u_h + 1, u_w + 1), each [kh, kw]: tap (i, j) of a stride-s conv inside the space-to-depth
    weight gradient [.., s, s, 3, 3] (conv2d_grad_weight_strided); built once per (device, geometry)."""
    key = (dev, s, kh, kw, p)
    idx = _TAP_GATHER.get(key)
    if idx is None:
        ua_h = [((i - p) // s, (i - p) % s) for i in range(kh)]                  # floor division: t = s u + a, 0 <= a < s
        ua_w = [((j - p) // s, (j - p) % s) for j in range(kw)]
        ah = torch.tensor([[a for _ in range(kw)] for (_, a) in ua_h], dtype=torch.long)
        aw = torch.tensor([[a for (_, a) in ua_w] for _ in range(kh)], dtype=torch.long)
        uh = torch.tensor([[u + 1 for _ in range(kw)] for (u, _) in ua_h], dtype=torch.long)
        uw = torch.tensor([[u + 1 for (u, _) in ua_w] for _ in range(kh)], dtype=torch.long)
        idx = _TAP_GATHER[key] = tuple(t.to(dev) for t in (ah, aw, uh, uw))
    return idx


def wgrad_strided_applicable(x_shape, g_shape, kernel_hw, stride, padding, dilation) -> bool:
    """Strided convs whose weight gradient runs on this backend's kernels (``conv2d_grad_weight_strided``): square stride
    s > 1, un-dilated, and either 1x1 / padding 0 or a kernel whose taps fall on offsets {-1, 0, +1} of the space-to-depth
    image (k <= s + 1 with padding 1, e.g. 3x3 / stride 2 / padding 1) with H, W multiples of s and >= 32 channels after
    the space-to-depth step."""
    (sh, sw), (ph, pw), (dh, dw) = _pairs(stride), _pairs(padding), _pairs(dilation)
    kh, kw = (int(v) for v in kernel_hw)
    if sh != sw or sh <= 1 or (dh, dw) != (1, 1) or kh != kw or ph != pw:
        return False
    N, Cin, H, W = (int(v) for v in x_shape)
    if kh == 1 and ph == 0:
        return True
    s = sh
    if H % s or W % s or ph != 1 or kh > s + 1 or kh < 2:
        return False
    if int(g_shape[2]) != H // s or int(g_shape[3]) != W // s:
        return False
    return Cin * s * s >= 32 and int(g_shape[1]) >= 32


def conv2d_grad_weight_strided(x: torch.Tensor, grad_output: torch.Tensor, kernel_hw, stride, padding,
                               x_levels: float = 1.0, bias_grad: Optional[list] = None,
                               layout_like: Optional[torch.Tensor] = None):
    """grad wrt the weight of a STRIDED conv2d(x, Q(W)) for an activation that is exact in bf16 (+-1 / 0, or a k-bit DoReFa
    image q / n with ``x_levels`` = n); un-masked (the caller applies the quantiser's STE).  See ``wgrad_strided_applicable``.

      * 1x1, padding 0: dW[co, ci] = sum_pos g[pos, co] x[s pos, ci] — the sub-sampled activation against the gradient, one
        exact-split GEMM over the positions (qt_bf16_gemm);
      * k x k, padding 1, k <= s + 1: tap i sits at t = i - 1 = s u + a of the input, i.e. at offset u in {-1, 0} (k = s + 1:
        {-1, 0}) of phase a of the space-to-depth image (H/s x W/s, Cin s^2 channels).  The stride-1 3x3 pixel-major kernel
        (csrc/wgrad_pm.hip) on that image gives dW'[co, (ci, a_h, a_w), u_h + 1, u_w + 1]; the k^2 real taps are gathered
        from it (the u = +1 row / column is computed and dropped)."""
    _require(x, "input")
    _require(grad_output, "grad_output")
    s = _pairs(stride)[0]
    kh, kw = (int(v) for v in kernel_hw)
    p = _pairs(padding)[0]
    N, Cin, H, W = (int(v) for v in x.shape)
    _, Cout, Ho, Wo = (int(v) for v in grad_output.shape)
    inv = _inv_f32(x_levels)
    if kh == 1:
        xs = x.detach()[:, :, 0:(Ho - 1) * s + 1:s, 0:(Wo - 1) * s + 1:s]
        # the sub-sampled activation (a strided VIEW: the packers take strides) against the gradient is a stride-1 1x1 weight
        # gradient: the K-major route cuts its long contraction (N Ho Wo positions, few output tiles) into K slices
        dW = conv2d_grad_weight_gemm(xs, grad_output, (1, 1), 0, x_levels=x_levels)
        if dW is not None:
            return dW
        x2 = xs.permute(1, 0, 2, 3).reshape(Cin, -1)                             # [Cin, P]  (gather copy)
        if x_levels != 1.0:
            x2 = torch.round(x2 * float(x_levels))                              # the integer codes: exact in bf16 (<= 255)
        g2 = grad_output.detach().permute(1, 0, 2, 3).reshape(Cout, -1)          # [Cout, P]
        # (three exact bf16 terms: a row of this GEMM is ONE gradient channel — the two-term form's per-tensor scale is not for it)
        dW = bf16_gemm(split_bf16x3(g2.contiguous(), terms=3), weight_bf16x3(x2.contiguous(), "raw", terms=3))
        if inv != 1.0:
            dW = dW * inv
        return dW.view(Cout, Cin, 1, 1)
    xs2d = torch.nn.functional.pixel_unshuffle(x.detach(), s)                    # [N, Cin s^2, H/s, W/s], channel = (ci, a_h, a_w)
    dWp = conv2d_grad_weight_pm(xs2d, grad_output, (3, 3), 1, weight=None, x_levels=x_levels, bias_grad=bias_grad)
    if dWp is None:
        return None
    dWp = dWp.view(Cout, Cin, s, s, 3, 3)
    ah, aw, uh, uw = _tap_gather_index(x.device, s, kh, kw, p)
    # ONE gather for the k^2 taps (it was one strided copy per tap); for a channels-last parameter the result is produced in that
    # layout (AccumulateGrad takes it as it is)
    if layout_like is not None and layout_like.dim() == 4 and layout_like.is_contiguous(memory_format=torch.channels_last) \
            and not layout_like.is_contiguous():
        return dWp.permute(0, 2, 3, 4, 5, 1)[:, ah, aw, uh, uw].permute(0, 3, 1, 2)     # [Cout, kh, kw, Cin] memory
    return dWp[:, :, ah, aw, uh, uw]


#: largest output map (Ho * Wo) for which the weight gradient takes the matrix-core route (see conv2d_grad_weight_pm1)
WEIGHT_GRAD_MAX_PIXELS = 256


def _weight_grad_chunk(HoWo: int, batch: int) -> int:
    """Images per launch of the weight-gradient conv: the largest power of two whose "weight" row (Ho*Wo pixels of
    6-byte triples, 16-byte pixel granule) stays under the implicit-GEMM kernel's K limit."""
    for nc in (32, 16, 8, 4, 2):
        if nc <= max(batch, 2) and HoWo * triple_ld_bytes(nc, 16) < (1 << 20) - 4096:
            return nc
    return 0


def conv2d_grad_weight_pm1(x_pm1: torch.Tensor, grad_output: torch.Tensor, kernel_hw, stride, padding, dilation,
                           max_launches: int = 64):
    """grad wrt the weight of conv2d(x, W) for a +-1 (or 0) activation x; None when outside the route."""
    _require(x_pm1, "input")
    _require(grad_output, "grad_output")
    (sh, sw), (ph, pw), (dh, dw) = _pairs(stride), _pairs(padding), _pairs(dilation)
    kh, kw = (int(v) for v in kernel_hw)
    N, Cin, H, W = (int(v) for v in x_pm1.shape)
    _, Cout, Ho, Wo = (int(v) for v in grad_output.shape)
    nc = _weight_grad_chunk(Ho * Wo, N)
    if nc == 0 or (N + nc - 1) // nc > max_launches or Cin * H * W >= (1 << 31):
        return None
    # measured against MIOpen's fp32 weight gradient at batch 256 (tools/bench_conv_backward.py): 13 x 13 maps with
    # 576-1152 channels 3.6 / 4.1 ms vs 4.1 / 5.5 ms; 27 x 27 x 192 -> 576 13.1 vs 11.0 ms; 112 x 112 x 64 -> 64 (batch 32)
    # 17.9 vs 0.3 ms — the swapped conv has only Cin * kh * kw output rows, so it pays on small maps with many channels only
    if Ho * Wo > WEIGHT_GRAD_MAX_PIXELS or Cin * kh * kw < 1024:
        return None
    # output of the swapped conv: ((H + 2p - s (Ho - 1) - 1) // d) + 1 >= kh positions; the surplus (when the forward conv
    # dropped trailing rows / columns) is cropped
    kh2 = (H + 2 * ph - sh * (Ho - 1) - 1) // dh + 1
    kw2 = (W + 2 * pw - sw * (Wo - 1) - 1) // dw + 1
    if kh2 < kh or kw2 < kw:
        return None
    Cb = triple_ld_bytes(nc, 16)
    kbytes = Ho * Wo * Cb
    ld = max(128, (kbytes + 127) // 128 * 128)
    dev = x_pm1.device
    acc = None
    meta_w = torch.empty((Cout, nc, Ho, Wo), device="meta")
    for n0 in range(0, N, nc):
        cnt = min(nc, N - n0)
        xs, gs = x_pm1[n0:n0 + cnt], grad_output[n0:n0 + cnt]
        if cnt < nc:      # ragged tail: zero images contribute nothing
            xs = torch.cat([xs, torch.zeros((nc - cnt,) + tuple(xs.shape[1:]), device=dev)], 0)
            gs = torch.cat([gs, torch.zeros((nc - cnt,) + tuple(gs.shape[1:]), device=dev)], 0)
        xt = xs.permute(1, 2, 3, 0).contiguous().view(Cin * H * W, nc)          # "pixels" (ci, h, w) x "channels" n
        px = weight_bf16x3(xt, "sign", ld_bytes=Cb, terms=3)                     # +-1 / 0 replicated three times
        gt = gs.permute(1, 2, 3, 0).contiguous().view(Cout * Ho * Wo, nc)
        tr = split_bf16x3(gt, ld_bytes=Cb, terms=3)                              # exact hi / mid / lo of the gradient
        data = tr.data.view(Cout, kbytes // 2)
        if ld != kbytes:
            padded = torch.zeros((Cout, ld // 2), dtype=torch.int16, device=dev)
            padded[:, :kbytes // 2] = data
            data = padded
        wt = TriplePlanes(data=data, rows=Cout, K=kbytes // 6)
        y2 = float_conv2d(None, meta_w, "sign", None, (dh, dw), (ph, pw), (sh, sw), weight_triples=wt, pixels=px,
                          in_shape=(Cin, nc, H, W))
        acc = y2 if acc is None else acc.add_(y2)
    gw = acc.view(Cin, kh2, kw2, Cout)[:, :kh, :kw, :].permute(3, 0, 1, 2)
    return gw.contiguous()


#: working-set budget (bytes) of one weight-gradient GEMM launch (operand planes + partial results); larger batches are
#: processed in chunks whose results are accumulated
WGRAD_GEMM_BYTES = 6 << 30
#: workgroups one weight-gradient launch aims for (K slices = this / (tiles x taps))
WGRAD_WORKGROUPS = 1024
WGRAD_PM_WORKGROUPS = 256


def _round_up(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def wgrad_gemm_applicable(x_shape, g_shape, kernel_hw, stride, dilation) -> bool:
    """Shapes for which conv2d_grad_weight_gemm beats the fp32 library (tools/check_wgrad_gemm.py, batch 256, MI355X):
    27 x 27 x 192 -> 576 k5 5.1 vs 11.1 ms, 13 x 13 x 576 -> 1152 2.4 vs 4.2, 13 x 13 x 1152 -> 768 3.0 vs 5.5, 28 x 28 x
    512 -> 512 4.0 vs 7.5, 14 x 14 x 512 -> 512 1.2 vs 1.9; it loses on big maps with few channels (56 x 56 x 256 -> 256 10.6 vs
    7.1, 224 x 224 x 64 -> 64 14 vs 1.3 at batch 32: every tap re-reads the gradient planes and the tiles are narrow)."""
    (sh, sw), (dh, dw) = _pairs(stride), _pairs(dilation)
    kh, kw = (int(v) for v in kernel_hw)
    return (sh == sw == 1 and dh == dw == 1 and kw <= 8 and int(g_shape[2]) * int(g_shape[3]) <= 1024
            and int(x_shape[1]) >= 128 and int(g_shape[1]) >= 128)


def conv2d_grad_weight_gemm(x_pm1: torch.Tensor, grad_output: torch.Tensor, kernel_hw, padding,
                            weight: Optional[torch.Tensor] = None, ste_threshold: float = STE_THRESHOLD,
                            x_levels: float = 1.0):
    """grad wrt the weight of a STRIDE-1, un-dilated conv2d(x, Q(W)) for an activation whose values are exact in bf16
    (+-1 / 0): batched bf16 matrix-core GEMMs over K-major operand planes (csrc/wgrad.hip).  ``weight``: apply the
    quantiser's straight-through mask 1[|W| <= ste_threshold] to the result.  ``x_levels`` = n: the activation is a k-bit
    DoReFa image q / n (n = 2^k - 1 <= 255); it enters the GEMM as its integer codes (x * n, exact in bf16) and the result
    is scaled by fl(1 / n).  Returns [Cout, Cin, kh, kw] fp32, or None when the shape is outside the route."""
    _require(x_pm1, "input")
    _require(grad_output, "grad_output")
    kh, kw = (int(v) for v in kernel_hw)
    ph, pw = _pairs(padding)
    N, Cin, H, W = (int(v) for v in x_pm1.shape)
    N2, Cout, Ho, Wo = (int(v) for v in grad_output.shape)
    if N2 != N or Ho != H + 2 * ph - kh + 1 or Wo != W + 2 * pw - kw + 1 or Ho <= 0 or Wo <= 0 or kw > 8 or N == 0:
        return None
    Wq = _round_up(W + 2 * pw, 8)
    Hp = H + 2 * ph
    M, taps = 3 * Cout, kh * kw
    ldc = _round_up(Cin, 4)
    tn = 256
    for c in (192, 128, 64):                                  # the GEMM's own tile-width rule (pick_tile_n)
        if _round_up(Cin, c) < _round_up(Cin, tn):
            tn = c
    tm = 384 if (tn == 192 and _round_up(M, 384) <= _round_up(M, 256)) else 256
    tiles = (_round_up(M, tm) // tm) * (_round_up(Cin, tn) // tn)
    nslice = max(1, min(64, -(-WGRAD_WORKGROUPS // (tiles * taps))))

    def plan(nc):
        ktot = Ho * nc * Wq
        ks = _round_up(-(-ktot // nslice), 32)
        kpad = ks * nslice
        lda = _round_up(kpad, 64)
        ldb = _round_up(max(Hp * nc * Wq, (kh - 1) * nc * Wq + kpad), 64)
        nbytes = M * lda * 2 + kw * Cin * ldb * 2 + taps * nslice * M * ldc * 4
        ok = M * lda * 2 < (1 << 31) and Cin * ldb * 2 < (1 << 31) and nbytes <= WGRAD_GEMM_BYTES
        return ok, ks, lda, ldb

    nc = N
    while nc > 1 and not plan(nc)[0]:
        nc = (nc + 1) // 2
    ok, ks, lda, ldb = plan(nc)
    if not ok:
        return None
    dev = x_pm1.device
    g = grad_output.detach()
    x = x_pm1.detach()
    out_scale = _inv_f32(x_levels)
    dW = torch.empty((Cout, Cin, kh, kw), dtype=torch.float32, device=dev)
    A = torch.empty((M, lda), dtype=torch.int16, device=dev)
    B = torch.empty((kw, Cin, ldb), dtype=torch.int16, device=dev)
    part = torch.empty((taps * nslice, M, ldc), dtype=torch.float32, device=dev)
    w = None
    if weight is not None:
        w = _require(weight.detach(), "weight").contiguous()
    I = int
    st = _stream(dev)
    with _on(dev):
        for n0 in range(0, N, nc):
            cnt = min(nc, N - n0)
            gs = g[n0:n0 + cnt]
            xs = x[n0:n0 + cnt]
            if cnt != nc:                                     # ragged last chunk: its own (smaller) position space
                ok2, ks2, lda2, ldb2 = plan(cnt)
                assert lda2 <= lda and ldb2 <= ldb      # nslice is fixed here, so the plan is monotone in the chunk size
                Au = A.view(-1)[:M * lda2].view(M, lda2)
                Bu = B.view(-1)[:kw * Cin * ldb2].view(kw, Cin, ldb2)
                k_use, lda_use, ldb_use = ks2, lda2, ldb2
            else:
                Au, Bu, k_use, lda_use, ldb_use = A, B, ks, lda, ldb
            _lib.call("qt_wgrad_pack_grad_f32", _p(gs), I(gs.stride(0)), I(gs.stride(1)), I(gs.stride(2)), I(gs.stride(3)),
                      I(cnt), I(Cout), I(Ho), I(Wo), I(Wq), _p(Au), I(lda_use), st)
            _lib.call("qt_wgrad_pack_act_f32", _p(xs), I(xs.stride(0)), I(xs.stride(1)), I(xs.stride(2)), I(xs.stride(3)),
                      I(cnt), I(Cin), I(H), I(W), I(ph), I(pw), I(Wq), I(kw), float(x_levels), _p(Bu), I(ldb_use),
                      I(Cin * ldb_use), st)
            _lib.call("qt_bf16_gemm_taps", _p(Au), I(lda_use // 2), _p(Bu), I(ldb_use // 2), _p(part), I(ldc), I(M), I(Cin),
                      I(k_use), I(kh), I(kw), I(nslice), I(Cin * ldb_use * 2), I(cnt * Wq * 2), I(M * ldc), st)
            _lib.call("qt_wgrad_reduce_f32", _p(part), I(ldc), I(M * ldc), I(taps), I(nslice), I(Cout), I(Cin), _p(w),
                      float(ste_threshold), float(out_scale), int(n0 > 0), _p(dW), st)
    return dW


def wgrad_pm_applicable(x_shape, g_shape, kernel_hw, stride, dilation) -> bool:
    """Shapes the pixel-major weight-gradient kernel (csrc/wgrad_pm.hip) is built for: stride 1, un-dilated, 3 x 3 or 5 x 5."""
    (sh, sw), (dh, dw) = _pairs(stride), _pairs(dilation)
    kh, kw = (int(v) for v in kernel_hw)
    return sh == sw == 1 and dh == dw == 1 and (kh, kw) in ((3, 3), (5, 5)) and int(x_shape[1]) >= 32 and int(g_shape[1]) >= 32


def wgrad_pm_plan(nc: int, Cout: int, Cin: int, H: int, W: int, Ho: int, kh: int, kw: int, ph: int, pw: int, slots: int):
    """Memoised front of ``_wgrad_pm_plan`` (the slice search walks up to 256 candidates in Python: ~100 us per call, twenty
    times per ResNet training step); the byte budget is part of the key because tests change it."""
    return _wgrad_pm_plan_cached(int(nc), int(Cout), int(Cin), int(H), int(W), int(Ho), int(kh), int(kw), int(ph), int(pw),
                                 int(slots), int(WGRAD_GEMM_BYTES))


@functools.lru_cache(maxsize=4096)
def _wgrad_pm_plan_cached(nc, Cout, Cin, H, W, Ho, kh, kw, ph, pw, slots, budget):
    return _wgrad_pm_plan(nc, Cout, Cin, H, W, Ho, kh, kw, ph, pw, slots)


def _wgrad_pm_plan(nc: int, Cout: int, Cin: int, H: int, W: int, Ho: int, kh: int, kw: int, ph: int, pw: int, slots: int):
    """Launch plan of the pixel-major weight gradient for ``nc`` images: (fits the byte budget?, K slices, positions in the
    gradient planes Qa = slices x slice length, rows of the activation plane Qx).  Host logic only (tests/test_wgrad_plan_cpu.py).
    K slices: the launch runs ceil(tiles * nslice / slots) rounds of ceil(ktot / nslice / 32) stages each, plus 24 stages' worth
    of prologue / partial-result write-out per round — few long slices beat many short ones; ties go to fewer slices."""
    tn = 64 if kh == 3 else 32
    Cpo, Cpi = _round_up(Cout, 64), _round_up(Cin, tn)
    Wq, Hp, taps = W + 2 * pw, H + 2 * ph, kh * kw
    tm = 128 if (kh == 3 and Cpo % 128 == 0) else 64                 # the kernel's own tile rule (qt_wgrad_pm_f32)
    tiles = (Cpo // tm) * (Cpi // tn)
    ktot = Ho * nc * Wq
    nslice = min(range(1, max(2, min(257, ktot // 256 + 1))),
                 key=lambda ns: (-(-tiles * ns // slots) * (-(-ktot // (ns * 32)) + 24), ns))
    ks = _round_up(-(-ktot // nslice), 32)
    qa = ks * nslice
    qx = max(Hp * nc * Wq, qa + (kh - 1) * nc * Wq + 48)
    nbytes = 3 * qa * Cpo * 2 + qx * Cpi * 2 + nslice * taps * Cpo * Cpi * 4
    return nbytes <= WGRAD_GEMM_BYTES, nslice, qa, qx


def _wgrad_pm_run(grad_output: torch.Tensor, geom, pack_act, weight, ste_threshold: float, out_scale: float, workgroups: int,
                  bias_grad: Optional[list] = None, terms: int = 3, layout_like: Optional[torch.Tensor] = None):
    """Pixel-major weight gradient for the position geometry ``geom`` = (N, Cin, H, W, kh, kw, ph, pw) of a stride-1 conv;
    ``pack_act(n0, cnt, Wq, Cpi, Qx, XP, stream, f16)`` writes the activation plane of images [n0, n0 + cnt) (bf16, or fp16 when
    ``f16``).  Returns [Cout, Cin, kh, kw] fp32 or None when the planes do not fit the byte budget.  ``bias_grad``: a list that
    receives the conv's bias gradient (sum of the gradient over n, y, x) when the gradient pack can produce it on the way
    (channels-last gradient, <= 2048 padded channels) — otherwise it stays empty and the caller reduces the gradient itself.
    ``terms``: 3 = the gradient as three exact bf16 planes; 2 = two fp16 planes of g[:, c] / s[c] with PER-CHANNEL powers of
    two s[c] (the reduce multiplies row c of the result by s[c]; the STE mask only zeroes entries)."""
    N, Cin, H, W, kh, kw, ph, pw = geom
    _, Cout, Ho, Wo = (int(v) for v in grad_output.shape)
    tn = 64 if kh == 3 else 32
    Cpo, Cpi = _round_up(Cout, 64), _round_up(Cin, tn)
    Wq, taps = W + 2 * pw, kh * kw
    slots = workgroups or WGRAD_PM_WORKGROUPS                        # resident workgroups: one per CU (three-stage LDS ring)

    def plan(nc):
        return wgrad_pm_plan(nc, Cout, Cin, H, W, Ho, kh, kw, ph, pw, slots)

    nc = N
    while nc > 1 and not plan(nc)[0]:
        nc = (nc + 1) // 2
    ok, nslice, qa, qx = plan(nc)
    if not ok:
        return None
    # the plan is not monotone in the chunk size (a ragged last chunk can ask for MORE slices or a longer plane than the
    # full chunks: 64 -> 64 3x3 at 32^2, 13 images: 50 slices, 12 images: 51): size every buffer for both plans
    tail = plan(N % nc) if N % nc else (True, nslice, qa, qx)
    ns_m, qa_m, qx_m = max(nslice, tail[1]), max(qa, tail[2]), max(qx, tail[3])
    dev = grad_output.device
    g = grad_output.detach()
    two = int(terms) == 2
    scale2 = None
    if two:       # per-channel powers of two: a row of dW sees one gradient channel only (csrc/split_f16.hip)
        scale2 = torch.empty((2 * Cpo,), dtype=torch.float32, device=dev)
        work = torch.empty((int(_lib.load().qt_f16x2_absmax_ch_work_words(Cout)),), dtype=torch.int32, device=dev)
        with _on(dev):
            _lib.call("qt_f16x2_absmax_scale_ch_f32", _p(g), int(g.stride(0)), int(g.stride(1)), int(g.stride(2)), int(g.stride(3)),
                      int(N), int(Cout), int(Ho), int(Wo), int(Cpo), _p(work), _p(scale2), _stream(dev))
    # the result takes the weight's own layout (a channels_last model keeps channels-last parameters): no contiguous() copy of
    # the weight for the STE mask, no re-layout when autograd accumulates the gradient
    w = None
    if weight is not None:
        w = _require(weight.detach(), "weight")
        if tuple(w.shape) != (Cout, Cin, kh, kw) or not (w.is_contiguous() or w.is_contiguous(memory_format=torch.channels_last)):
            w = w.contiguous()
    if w is not None:
        dW = torch.empty_like(w)
    elif (layout_like is not None and tuple(layout_like.shape) == (Cout, Cin, kh, kw) and layout_like.dtype == torch.float32
          and (layout_like.is_contiguous() or layout_like.is_contiguous(memory_format=torch.channels_last))):
        # un-masked gradient (DoReFa: upstream's _ignore_factor_op) in the PARAMETER's layout: autograd's AccumulateGrad takes a
        # gradient whose strides match the parameter's as it is, anything else is re-laid-out by a copy kernel per layer and step
        dW = torch.empty_like(layout_like)
    else:
        dW = torch.empty((Cout, Cin, kh, kw), dtype=torch.float32, device=dev)
    dws = tuple(int(v) for v in dW.stride())
    G3 = torch.empty(((2 if two else 3) * qa_m * Cpo,), dtype=torch.int16, device=dev)
    XP = torch.empty((qx_m * Cpi,), dtype=torch.int16, device=dev)
    part = torch.empty((ns_m * taps * Cpo * Cpi,), dtype=torch.float32, device=dev)
    I = int
    st = _stream(dev)
    want_bias = bias_grad is not None and g.stride(1) == 1 and Cpo <= 2048
    bias_part = db = None
    if want_bias:
        bias_part = torch.empty(((Ho * nc + 128) * Cpo,), dtype=torch.float32, device=dev)    # + the reduce's scratch rows
        db = torch.empty((Cout,), dtype=torch.float32, device=dev)
    with _on(dev):
        for n0 in range(0, N, nc):
            cnt = min(nc, N - n0)
            gs = g[n0:n0 + cnt]
            _, ns_u, qa_u, qx_u = tail if cnt != nc else (True, nslice, qa, qx)
            assert ns_u <= ns_m and qa_u <= qa_m and qx_u <= qx_m
            if two:
                _lib.call("qt_wgrad_pm_pack_grad_f16x2", _p(gs), I(gs.stride(0)), I(gs.stride(1)), I(gs.stride(2)), I(gs.stride(3)),
                          I(cnt), I(Cout), I(Ho), I(Wo), I(Wq), I(Cpo), I(qa_u), _p(scale2), _p(G3), _p(bias_part), st)
            elif want_bias:
                _lib.call("qt_wgrad_pm_pack_grad_bias_f32", _p(gs), I(gs.stride(0)), I(gs.stride(2)), I(gs.stride(3)),
                          I(cnt), I(Cout), I(Ho), I(Wo), I(Wq), I(Cpo), I(qa_u), _p(G3), _p(bias_part), st)
            else:
                _lib.call("qt_wgrad_pm_pack_grad_f32", _p(gs), I(gs.stride(0)), I(gs.stride(1)), I(gs.stride(2)), I(gs.stride(3)),
                          I(cnt), I(Cout), I(Ho), I(Wo), I(Wq), I(Cpo), I(qa_u), _p(G3), st)
            if want_bias:
                _lib.call("qt_wgrad_pm_bias_reduce_f32", _p(bias_part), I(Ho * cnt), I(Cpo), I(Cout), int(n0 > 0), _p(db), st)
            pack_act(n0, cnt, Wq, Cpi, qx_u, XP, st, two)
            _lib.call("qt_wgrad_pm_f16" if two else "qt_wgrad_pm_f32", _p(G3), _p(XP), _p(part), I(qa_u), I(cnt * Wq), I(ns_u),
                      I(Cpo), I(Cpi), I(kh), I(kw), st)
            _lib.call("qt_wgrad_pm_reduce_f32", _p(part), I(ns_u), I(taps), I(Cpo), I(Cpi), I(Cout), I(Cin), _p(w),
                      float(ste_threshold), float(out_scale), _p(scale2), int(n0 > 0), _p(dW), *dws, st)
    if want_bias:
        bias_grad.append(db)
    return dW


def conv2d_grad_weight_pm(x_pm1: torch.Tensor, grad_output: torch.Tensor, kernel_hw, padding,
                          weight: Optional[torch.Tensor] = None, ste_threshold: float = STE_THRESHOLD,
                          x_levels: float = 1.0, workgroups: int = 0, bias_grad: Optional[list] = None,
                          terms: Optional[int] = None, layout_like: Optional[torch.Tensor] = None):
    """Same contract as ``conv2d_grad_weight_gemm`` on the pixel-major kernel (csrc/wgrad_pm.hip): the operands stay
    [position][channel] (what channels-last tensors already are), one workgroup accumulates every tap of its tile, so
    the gradient planes are read once instead of once per tap.  Returns None outside (3, 3) / (5, 5) stride-1 convs.
    ``bias_grad``: see ``_wgrad_pm_run`` (the bias gradient as a by-product of the gradient pack).  ``terms``: how the
    real-valued gradient is split (default: FLOAT_SPLIT — two fp16 terms, or three exact bf16 terms)."""
    _require(x_pm1, "input")
    _require(grad_output, "grad_output")
    kh, kw = (int(v) for v in kernel_hw)
    ph, pw = _pairs(padding)
    N, Cin, H, W = (int(v) for v in x_pm1.shape)
    N2, Cout, Ho, Wo = (int(v) for v in grad_output.shape)
    if (N2 != N or Ho != H + 2 * ph - kh + 1 or Wo != W + 2 * pw - kw + 1 or Ho <= 0 or Wo <= 0 or N == 0
            or (kh, kw) not in ((3, 3), (5, 5))):
        return None
    x = x_pm1.detach()
    out_scale = _inv_f32(x_levels)
    I = int

    def pack_act(n0, cnt, Wq, Cpi, qx, XP, st, f16=False):
        xs = x[n0:n0 + cnt]
        _lib.call("qt_wgrad_pm_pack_act_f16" if f16 else "qt_wgrad_pm_pack_act_f32", _p(xs), I(xs.stride(0)), I(xs.stride(1)),
                  I(xs.stride(2)), I(xs.stride(3)), I(cnt), I(Cin), I(H), I(W), I(ph), I(pw), I(Wq), I(Cpi), I(qx), float(x_levels),
                  _p(XP), st)

    return _wgrad_pm_run(grad_output, (N, Cin, H, W, kh, kw, ph, pw), pack_act, weight, ste_threshold, out_scale, workgroups,
                         bias_grad, terms=split_terms(terms), layout_like=layout_like)


def wgrad_s2d_applicable(x_shape, kernel_hw, stride, dilation, any_channels: bool = False) -> bool:
    """A conv over a few REAL-VALUED input channels — strided (the first layer: AlexNet's 3 -> 192, k 11, stride 4) or stride 1
    with <= 8 channels (VGG's 3 -> 64): its weight gradient runs on the pixel-major kernel through the space-to-depth image
    (``conv2d_grad_weight_s2d``; s = 1 is just the three-term split of the image as channel groups)."""
    (sh, sw), (dh, dw) = _pairs(stride), _pairs(dilation)
    kh, kw = (int(v) for v in kernel_hw)
    k2 = -(-kh // max(sh, 1))
    if any_channels:
        # a real-valued activation with MANY channels (XNORConv2d with quant_input=True: sign(x) * per-pixel scale): stride 1 only,
        # the two / three terms of the image as channel groups of the same kernel (the plan decides whether the planes fit)
        return sh == sw == 1 and dh == dw == 1 and kh == kw and kh in (3, 5)
    return (sh == sw and dh == dw == 1 and kh == kw and kh >= sh and k2 in (3, 5)
            and (sh > 1 or int(x_shape[1]) <= 8) and 3 * int(x_shape[1]) * sh * sh <= 256)


def conv2d_grad_weight_s2d(x: torch.Tensor, grad_output: torch.Tensor, weight_shape, stride, padding,
                           weight: Optional[torch.Tensor] = None, ste_threshold: float = STE_THRESHOLD,
                           bias_grad: Optional[list] = None, terms: Optional[int] = None, any_channels: bool = False):
    """grad wrt the weight of a strided conv2d over a real-valued fp32 image with few channels (``any_channels``: a stride-1
    3 x 3 / 5 x 5 conv over a real-valued activation of any width).  The stride-s conv is the
    stride-1 conv of the space-to-depth image ([N, C s^2, H / s, W / s], kernel ceil(k / s)) — the identity the forward uses
    (``s2d_weight``) — so the gradient of the s2d weight comes from ``conv2d_grad_weight_pm`` and is folded back with
    pixel_shuffle (the taps the rounding added are dropped).  The image is real-valued: its split goes in as channel groups —
    three exact bf16 terms (3 x C s^2 channels), or, with the two-term form (``terms`` = 2 / FLOAT_SPLIT = "f16x2", the
    default), two fp16 terms of x / s (2 x C s^2 channels, per-tensor power-of-two s) against the two-plane gradient; the
    groups are summed afterwards.  Returns [Cout, C, k, k] fp32 or None."""
    _require(x, "input")
    _require(grad_output, "grad_output")
    Cout, C, kh, kw = (int(v) for v in weight_shape)
    (sh, sw), (ph, pw) = _pairs(stride), _pairs(padding)
    if not wgrad_s2d_applicable(x.shape, (kh, kw), stride, 1, any_channels):
        return None
    s = sh
    k2 = -(-kh // s)
    N, C2, H, W = (int(v) for v in x.shape)
    N2, Co2, Ho, Wo = (int(v) for v in grad_output.shape)
    if C2 != C or N2 != N or Co2 != Cout or Ho != (H + 2 * ph - kh) // s + 1 or Wo != (W + 2 * pw - kw) // s + 1:
        return None
    Hs, Ws = Ho + k2 - 1, Wo + k2 - 1
    Cs = C * s * s
    Cs8 = _round_up(Cs, 8)
    xd = x.detach()
    I = int

    nt = split_terms(terms)
    x_scale = pow2_scale(xd) if nt == 2 else None        # the image's per-tensor power of two (two fp16 terms of x / s)

    def pack_act(n0, cnt, Wq, Cpi, qx, XP, st, f16=False):
        # space-to-depth gather + split in one pass: XP[q][t * Cs8 + (c s + dy) s + dx] = term t of
        # xpad[n, c, Y s + dy - ph, X s + dx - pw] — the image's terms (three exact bf16, or two fp16 of x / s) are channel
        # groups of the activation plane
        xs = xd[n0:n0 + cnt]
        if nt == 2:
            _lib.call("qt_wgrad_pm_pack_act_s2d_f16x2", _p(xs), I(xs.stride(0)), I(xs.stride(1)), I(xs.stride(2)), I(xs.stride(3)),
                      I(cnt), I(C), I(H), I(W), I(s), I(ph), I(pw), I(Hs), I(Ws), I(Wq), I(Cs8), I(Cpi), I(qx), _p(x_scale),
                      _p(XP), st)
        else:
            _lib.call("qt_wgrad_pm_pack_act_s2d_f32", _p(xs), I(xs.stride(0)), I(xs.stride(1)), I(xs.stride(2)), I(xs.stride(3)),
                      I(cnt), I(C), I(H), I(W), I(s), I(ph), I(pw), I(Hs), I(Ws), I(Wq), I(Cs8), I(Cpi), I(qx), _p(XP), st)

    dws = _wgrad_pm_run(grad_output, (N, nt * Cs8, Hs, Ws, k2, k2, 0, 0), pack_act, None, ste_threshold, 1.0, 0, bias_grad,
                        terms=nt)
    if dws is None:
        return None
    dws = dws.view(Cout, nt, Cs8, k2, k2)[:, :, :Cs]
    if nt == 2:
        dws = (dws[:, 1] + dws[:, 0]) * x_scale[0]
    else:
        dws = (dws[:, 2] + dws[:, 1]) + dws[:, 0]
    dW = torch.nn.functional.pixel_shuffle(dws, s)[:, :, :kh, :kw].contiguous()
    if weight is not None:
        dW = ste_mask(dW, weight.detach().contiguous(), ste_threshold)
    return dW


# ---- direct first-layer conv (csrc/conv_first_direct.hip) ---------------------------------------------------------------------------

#: strided few-channel first layers (AlexNet's 3 -> 192, 11 x 11, stride 4) run on the direct kernel: the fp32 image is read once,
#: split in registers and contracted by stride addressing of an LDS patch (False: the space-to-depth pack + implicit GEMM of round 3)
FIRST_DIRECT = True


@dataclass
class FirstLayerWeights:
    """Packed weight of the direct first-layer conv: fp16 chunks [k-steps][2][Coutp][8] (int16 storage), ``lo`` / ``scale_dev`` for
    real-valued weights (w / scale = hi + lo), geometry the planes were built for."""
    hi: torch.Tensor
    lo: Optional[torch.Tensor]
    scale_dev: Optional[torch.Tensor]
    Cout: int
    Coutp: int
    Cp: int
    kernel_hw: tuple


def first_direct_cp(C: int, S: int) -> int:
    """Channels per pixel of the kernel's LDS patch: the smallest Cp >= C with S * Cp % 4 == 0 (8-byte aligned pixel runs)."""
    cp = int(C)
    while (int(S) * cp) % 4:
        cp += 1
    return cp


def first_direct_applicable(C: int, kernel_hw, stride, padding, dilation) -> bool:
    (sh, sw), (dh, dw) = _pairs(stride), _pairs(dilation)
    kh, kw = (int(v) for v in kernel_hw)
    if not (_cfg("FIRST_DIRECT") and sh == sw and sh >= 2 and (dh, dw) == (1, 1) and not isinstance(padding, str) and kh >= sh and kw >= sw):
        return False
    cp = first_direct_cp(C, sh)
    return int(C) <= 4 and cp <= 8 and kw * cp <= 256 and kh <= 64


def pack_first_layer_weight(wq: torch.Tensor, stride: int, real: bool = False) -> FirstLayerWeights:
    """[Cout, C, kh, kw] QUANTISED weight image (fp32) -> the direct kernel's fragment order.  ``real``: the values are not exact in
    fp16 (XNOR-Net's sign(W) * alpha): two fp16 terms of w / s with the tensor's power-of-two s (max|w| / s in [2^14, 2^15)), found on
    the device.  Weights are tiny: plain torch ops, cached by eval-mode layers."""
    wq = _require(wq.detach(), "weight")
    Cout, C, kh, kw = (int(v) for v in wq.shape)
    Cp = first_direct_cp(C, stride)
    g4 = (kw * Cp + 3) // 4                        # 8-byte groups (4 k) per kernel row
    ng4 = kh * g4
    nks = (ng4 + 3) // 4                           # k-steps of 16 (the last one zero-filled)
    Coutp = (Cout + 31) // 32 * 32
    F = torch.nn.functional
    scale = None
    if real:
        _, e = torch.frexp(wq.abs().amax())                      # max = m 2^e, m in [0.5, 1): max / 2^(e - 15) in [2^14, 2^15)
        scale = torch.ldexp(torch.ones((), dtype=torch.float32, device=wq.device), e - 15).reshape(1)
        wq = wq / scale

    def frag(t):                                                # [Cout, C, kh, kw] fp32 -> [nks, 2, Coutp, 8] fp16
        t = F.pad(t.permute(0, 2, 3, 1), (0, Cp - C)).reshape(Cout, kh, kw * Cp)
        t = F.pad(t, (0, g4 * 4 - kw * Cp)).reshape(Cout, ng4 * 4)
        t = F.pad(t, (0, nks * 16 - ng4 * 4, 0, Coutp - Cout)).reshape(Coutp, nks, 2, 8)
        return t.permute(1, 2, 0, 3).contiguous()
    hi = frag(wq).to(torch.float16)
    lo = None
    if real:
        lo = frag(wq).sub_(hi.float()).to(torch.float16).view(torch.int16)
    return FirstLayerWeights(hi=hi.view(torch.int16), lo=lo, scale_dev=scale, Cout=Cout, Coutp=Coutp, Cp=Cp, kernel_hw=(kh, kw))


def conv_first_direct(x: torch.Tensor, fw: FirstLayerWeights, bias=None, stride=1, padding=0, epi=None):
    """Direct first-layer conv of a real-valued [N, C, H, W] fp32 image (any storage order).  Returns the NHWC result
    [N*Ho*Wo, Cout] fp32, or with ``epi`` = (alpha, beta) the BitPlanes of the BatchNorm-threshold bits; None when the shape is
    outside the kernel's limits."""
    _require(x, "input")
    N, C, H, W = (int(v) for v in x.shape)
    kh, kw = fw.kernel_hw
    s = _pairs(stride)[0]
    ph, pw = _pairs(padding)
    Ho, Wo = (H + 2 * ph - kh) // s + 1, (W + 2 * pw - kw) // s + 1
    if Ho <= 0 or Wo <= 0 or N * Ho * Wo >= (1 << 31) or H > 32767 or W > 32767 or N == 0:
        return None
    dev = x.device
    bias = _check_bias(bias, fw.Cout, dev)
    sN, sC, sH, sW = (int(v) for v in x.stride())
    head = (_p(x), sN, sC, sH, sW, N, C, H, W, kh, kw, s, ph, pw, int(fw.Cp), _p(fw.hi), _p(fw.lo), 1.0, _p(fw.scale_dev),
            int(fw.Cout), int(fw.Coutp), _p(bias))
    if epi is not None:
        alpha, beta = _check_bias(epi[0], fw.Cout, dev), _check_bias(epi[1], fw.Cout, dev)
        ldb = packed_ld(fw.Cout)
        plane = torch.empty((N * Ho * Wo, ldb), dtype=torch.int32, device=dev)
        with _on(dev):
            _lib.call("qt_conv_first_direct_bits_f32", *head, _p(alpha), _p(beta), _p(plane), int(ldb), _stream(dev))
        return BitPlanes(sign=plane, rows=N * Ho * Wo, K=fw.Cout)
    y = torch.empty((N * Ho * Wo, fw.Cout), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.call("qt_conv_first_direct_f32", *head, _p(y), int(fw.Cout), _stream(dev))
    return y


#: stride-1 3 x 3 / padding-1 first layers with <= 4 real-valued input channels and 64 output channels (VGG-16's conv1_1) run on
#: the one-pass kernel of csrc/conv_first3x3.hip (fp32 image in, fp32 / threshold bits / the next conv's nibble halo plane out)
FIRST_3X3 = True


def first3x3_applicable(C: int, Cout: int, kernel_hw, stride, padding, dilation) -> bool:
    # (two fp16 terms per tile: only under the default two-term split — ops.float_split("bf16x3") asks for the exact route)
    return (_cfg("FIRST_3X3") and split_terms(None) == 2 and tuple(int(v) for v in kernel_hw) == (3, 3) and _pairs(stride) == (1, 1)
            and _pairs(padding) == (1, 1) and _pairs(dilation) == (1, 1) and 1 <= int(C) <= 4 and int(Cout) == 64)


def pack_first3x3_weight(wq: torch.Tensor) -> torch.Tensor:
    """[64, C <= 4, 3, 3] QUANTISED weight image (fp32, values exact in fp16: +-1 / 0) -> the MFMA row fragments of
    qt_conv3x3_first_f32 (int32 [1536]); cached by eval-mode layers."""
    wq = _require(wq.detach(), "weight")
    Cout, C, kh, kw = (int(v) for v in wq.shape)
    if (kh, kw) != (3, 3) or not 1 <= C <= 4 or Cout > 64:
        raise ValueError("pack_first3x3_weight takes a [<= 64, <= 4, 3, 3] weight")
    frag = torch.empty((3 * 2 * 64 * 4,), dtype=torch.int32, device=wq.device)
    with _on(wq.device):
        _lib.call("qt_conv3x3_first_pack_weight_f32", _p(wq), *(int(v) for v in wq.stride()), C, Cout, _p(frag), _stream(wq.device))
    return frag


def conv_first3x3(x: torch.Tensor, wfrag: torch.Tensor, Cout: int, bias=None, epi=None):
    """conv2d(x, Q(W), b, stride 1, padding 1) of a real-valued [N, C <= 4, H, W] fp32 image (any storage order) with a 3 x 3 kernel
    and 64 output channels in one pass.  ``epi`` None: the NHWC result [N*H*W, 64] fp32; (alpha, beta): BitPlanes of the
    BatchNorm-threshold bits; NibEpilogue (halo (1, 1)): the next conv's nibble halo plane.  None outside the kernel's limits."""
    _require(x, "input")
    N, C, H, W = (int(v) for v in x.shape)
    if N == 0 or N * (H + 2) >= (1 << 31) or int(Cout) != 64:
        return None
    dev = x.device
    bias = _check_bias(bias, Cout, dev)
    head = (_p(x), *(int(v) for v in x.stride()), N, C, H, W, _p(wfrag), int(Cout), _p(bias))
    if epi is None:
        y = torch.empty((N * H * W, Cout), dtype=torch.float32, device=dev)
        with _on(dev):
            _lib.call("qt_conv3x3_first_f32", *head, None, None, _p(y), int(Cout), 0, _stream(dev))
        return y
    nib = isinstance(epi, NibEpilogue)
    if nib and (tuple(epi.out_halo) != (1, 1) or epi.d2s_cout):
        return None
    alpha, beta = (epi.alpha, epi.beta) if nib else epi[:2]
    alpha, beta = _check_bias(alpha, Cout, dev), _check_bias(beta, Cout, dev)
    if nib:
        out = torch.empty((N * (H + 2) * (W + 2), 8), dtype=torch.int32, device=dev)
        with _on(dev):
            _lib.call("qt_conv3x3_first_f32", *head, _p(alpha), _p(beta), _p(out), 8, 2, _stream(dev))
        return NibPlanes(words=out, rows=int(out.shape[0]), K=Cout)
    out = torch.empty((N * H * W, 4), dtype=torch.int32, device=dev)
    with _on(dev):
        _lib.call("qt_conv3x3_first_f32", *head, _p(alpha), _p(beta), _p(out), 4, 1, _stream(dev))
    return BitPlanes(sign=out, rows=N * H * W, K=Cout)


def s2d_applicable(C: int, kh: int, kw: int, stride, dilation, padding=0) -> bool:
    """Strided first-layer style convs (few input channels) are re-expressed as stride-1 convs on the
    space-to-depth image: no padding waste in the pixel planes and ~(k/ceil(k/s)s)^2 of the K bytes.  With stride 1
    the same gather (s = 1) is simply the physically zero-padded pixel plane, which lets a padded few-channel conv
    (VGG's 3 -> 64 first layer) run on the un-padded conv kernels."""
    (sh, sw), (dh, dw), (ph, pw) = _pairs(stride), _pairs(dilation), _pairs(padding)
    if not (sh == sw and dh == dw == 1 and kh == kw and C * sh * sh <= 64 and kh >= sh):
        return False
    return sh > 1 or ph > 0 or pw > 0


def s2d_weight(wq: torch.Tensor, s: int) -> torch.Tensor:
    """[Cout, C, k, k] QUANTISED weight -> [Cout, C*s*s, k', k'] (k' = ceil(k/s)); the taps added by the
    rounding up are true zeros."""
    Cout, C, k, _ = wq.shape
    k2 = (k + s - 1) // s * s
    wp = torch.nn.functional.pad(wq, (0, k2 - k, 0, k2 - k))
    return torch.nn.functional.pixel_unshuffle(wp, s).contiguous()


def d2s_first_layer_applicable(Cin: int, Cout: int, kernel_hw, stride, padding, dilation, H: int, W: int) -> bool:
    """A 3x3 / stride 1 / padding 1 conv over a few input channels (VGG's 3 -> 64 first layer) can run in its 2x2
    output-blocked form: see ``d2s_first_layer_weight``."""
    return (tuple(kernel_hw) == (3, 3) and _pairs(stride) == (1, 1) and _pairs(padding) == (1, 1)
            and _pairs(dilation) == (1, 1) and Cin * 4 <= 64 and Cout % 32 == 0 and H % 2 == 0 and W % 2 == 0)


def d2s_first_layer_weight(wq: torch.Tensor) -> torch.Tensor:
    """[Cout, C, 3, 3] QUANTISED weight -> [4*Cout, C, 4, 4]: row (dy*2 + dx)*Cout + co holds the 3x3 kernel of channel
    co shifted by (dy, dx) inside a 4x4 window (zeros elsewhere).  The 4x4 / stride-2 / padding-1 conv with this weight
    computes, at block (by, bx), the four outputs (2by + dy, 2bx + dx) of the original 3x3 / stride-1 / padding-1 conv
    (same products; each 4x4 window of the padded image is gathered once for 2x2 output pixels instead of four 3x3
    windows), and as a stride-2 conv it takes the space-to-depth route (2x2 taps over 4C channels)."""
    parts = [torch.nn.functional.pad(wq, (dx, 1 - dx, dy, 1 - dy)) for dy in (0, 1) for dx in (0, 1)]
    return torch.cat(parts, 0).contiguous()


def s2d_input(x: torch.Tensor, s: int, padding) -> torch.Tensor:
    """[N, C, H, W] -> zero-padded, space-to-depth [N, C*s*s, ceil((H+2p)/s), ceil((W+2p)/s)] in NHWC storage."""
    ph, pw = _pairs(padding)
    N, C, H, W = x.shape
    Hp, Wp = H + 2 * ph, W + 2 * pw
    eh, ew = (-Hp) % s, (-Wp) % s
    xp = torch.nn.functional.pad(x, (pw, pw + ew, ph, ph + eh))
    return torch.nn.functional.pixel_unshuffle(xp, s).contiguous(memory_format=torch.channels_last)


# ----------------------------------------------------------------------------------------------
# formulation-agnostic front (bench.py and the layers go through this)
# ----------------------------------------------------------------------------------------------

GEMM_IMPLS = ("valu", "mfma")

#: 'auto' thresholds (tools/bench_crossover.py, MI355X).  The matrix-core route costs two launches for tagged
#: activations (bits -> nibbles, GEMM) and a workgroup walks its whole K loop alone, so it has a floor of
#: ~30 us end to end; the popcount kernels (tiled / skinny, 4x less operand traffic) start at ~20 us and grow
#: with the work.  Measured crossover: the MFMA route wins from ~8e9 ops when K is long (256x4096x4096: 33 vs
#: 38 us, 256x4096x9216: 47 vs 67 us, 2048^3: 31 vs 38 us) and from ~3e10 ops regardless (4096^3: 36 vs 148 us).
MFMA_MIN_OPS_LONG_K, MFMA_LONG_K, MFMA_MIN_OPS = 8.0e9, 2048, 3.0e10


def select_gemm_impl(requested: str, M: int, N: int, K: int) -> str:
    """'auto' -> the faster formulation for the shape (both are bit-exact)."""
    if requested == "auto":
        # a handful of rows on one side (batch <= 8: small-batch serving; <= 32 output features: classifier heads): the
        # streaming popcount kernel reads every packed word once with K along the lanes (csrc/popc_stream.hip) — 1 x 4096 x 9216
        # 3.1 us, 8 x 4096 x 9216 7.5 us, 256 x 10 x 4096 3.6 us against 11.4 / 11.9 / 10.0 us for bits -> nibbles + the
        # skinny matrix-core configuration (tools/bench_popc_stream.py); from batch 16 on the matrix cores win
        if 1 <= M <= 8 or 1 <= N <= 32:
            return "valu"
        ops_ = 2.0 * M * N * K
        big = (ops_ >= MFMA_MIN_OPS or (ops_ >= MFMA_MIN_OPS_LONG_K and K >= MFMA_LONG_K)) and K < (1 << 24)
        # skinny, long-K products (FC layers at batch <= 256): the 64x64 / 512-byte-stage matrix-core configuration beats
        # the weight-streaming popcount kernel even with the bits -> nibble expansion of the activation in front of it
        # (256x1000x4096: 5 + 4.5 us vs 19.7 us; tools/bench_gemm_variants.py variant 31)
        skinny = M <= 256 and N >= 256 and K >= MFMA_LONG_K and K % 1024 == 0 and K < (1 << 24)
        return "mfma" if (big or skinny) else "valu"
    if requested not in GEMM_IMPLS:
        raise NotImplementedError(f"packed GEMM formulation {requested!r} is not built "
                                  f"(available: {GEMM_IMPLS})")
    return requested


def pack_activations(x: torch.Tensor, impl: str = "valu"):
    """Packed image of safeSign(x) in the operand format of ``impl``."""
    if impl == "valu":
        return sign_pack(x)[0]
    if impl == "mfma":
        return sign_pack_nib(x)
    raise NotImplementedError(impl)


def pack_weights(w: torch.Tensor, kind: str = "binary", impl: str = "valu"):
    w2 = w.reshape(w.shape[0], -1)
    if impl == "valu":
        return sign_pack(w2)[0] if kind == "binary" else ternary_pack(w2)
    if impl == "mfma":
        return sign_pack_nib(w2) if kind == "binary" else ternary_pack_nib(w2)
    raise NotImplementedError(impl)


def pack_linear_operands(x: torch.Tensor, w: torch.Tensor, kind: str = "binary", impl: str = "valu"):
    """(packed safeSign(x), packed Q(w)) for one quantised linear forward.  On the matrix-core route both nibble
    planes come out of ONE launch (qt_pack_pair_nib_f32)."""
    if impl != "mfma":
        return pack_activations(x, impl), pack_weights(w, kind, impl)
    _require(x, "input")
    _require(w, "weight")
    x2, w2 = _as_rows(x), _as_rows(w.reshape(w.shape[0], -1))
    (M, K), (N, Kw) = (int(v) for v in x2.shape), (int(v) for v in w2.shape)
    if K != Kw:
        raise ValueError(f"K mismatch: activations {K} vs weights {Kw}")
    ld = packed_ld_nib(K)
    xn = torch.empty((M, ld), dtype=torch.int32, device=x.device)
    wn = torch.empty((N, ld), dtype=torch.int32, device=x.device)
    I = int
    with _on(x.device):
        _lib.call("qt_pack_pair_nib_f32", _p(x2), I(x2.stride(0) if M > 1 else max(K, 1)), _p(xn), I(ld), I(M),
                  _p(w2), I(w2.stride(0) if N > 1 else max(K, 1)), _p(wn), I(ld), I(N), I(K),
                  int(0 if kind == "binary" else 1), _stream(x.device))
    return NibPlanes(words=xn, rows=M, K=K), NibPlanes(words=wn, rows=N, K=K)


def to_impl(planes, impl: str):
    """Convert packed operands to the format ``impl`` consumes (bit planes -> nibble planes only)."""
    if impl == "valu":
        if isinstance(planes, BitPlanes):
            return planes
        raise TypeError("nibble planes cannot feed the popcount GEMM")
    if isinstance(planes, NibPlanes):
        return planes
    if getattr(planes, "nib", None) is not None:      # written by the producer's own pass (pool_affine_sign_pack(want_nib=True))
        return planes.nib
    return bits_to_nib(planes)


def packed_gemm(x, w, bias=None, out=None, impl: str = "valu") -> torch.Tensor:
    if impl == "valu":
        return tern_gemm(x, w, bias, out) if w.is_ternary else xnor_gemm(x, w, bias, out)
    if impl == "mfma":
        return nib_gemm(to_impl(x, "mfma"), to_impl(w, "mfma"), bias, out)
    raise NotImplementedError(impl)


def nib_gemm_kernel_name(M: int, N: int, K: int) -> str:
    """The kernel configuration nib_gemm's automatic dispatch takes for this shape (qt_nib_gemm_describe)."""
    import ctypes
    buf = ctypes.create_string_buffer(128)
    ld = packed_ld_nib(K)
    rc = _lib.load().qt_nib_gemm_describe(int(M), int(N), int(K), ld, ld, buf, 128)
    if rc != 0:
        raise _lib.QtStatusError(f"qt_nib_gemm_describe failed: {_lib.strerror(rc)}")
    return buf.value.decode()


def packed_gemm_algorithmic_bytes(M: int, N: int, K: int, impl: str = "valu", planes_w: int = 1,
                                  bias: bool = False) -> float:
    """Algorithmic HBM bytes of ONE packed-GEMM launch (DESIGN.md 'Kernels'): every operand read
    once, the fp32 result written once.  bit planes: 1 bit/element/plane; nibble planes: 4 bits."""
    if impl == "valu":
        return M * K / 8.0 + planes_w * N * K / 8.0 + 4.0 * M * N + (4.0 * N if bias else 0.0)
    return M * K / 2.0 + N * K / 2.0 + 4.0 * M * N + (4.0 * N if bias else 0.0)
