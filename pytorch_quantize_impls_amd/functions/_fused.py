"""Forward dispatch shared by the quantised Linear / Conv2d layers and the fused Function forms.

Reference call sites this replaces (all ``torch.nn.functional.linear / conv2d`` on a quantised
weight): layers/binary_layers.py:44,46,105,106; layers/terner_layers.py:49,51,91,92;
functions/binary_connect.py:93-98,128-131; functions/terner_connect.py:92-94.

Device tensors:
  * activation known to be +-1 (bit planes attached by the quantiser, or verified on device)
        -> bit-pack the weight, XNOR-popcount / ternary GEMM in libqt_hip.so;
  * arbitrary fp32 activation (first layer of every model)
        -> weight quantised by the HIP elementwise kernel, contraction by the dense fp32 GEMM
           library (the values are exact +-1/0, so this is the reference computation itself).
CPU tensors: the same expression in torch (host logic; never used for a device tensor).
"""
from __future__ import annotations

import contextlib
import threading
import weakref
from collections import Counter
from typing import Optional

import torch
import torch.nn.functional as F

from .. import ops, packed
from .common import QtFunction

#: When True (default) an un-tagged device activation is checked on the device for being exactly
#: +-1 before the packed path is taken (costs one read of the activation and one host sync).
#: A layer can override per instance with its ``binary_input`` attribute (True / False / None).
DETECT_BINARY_INPUT = True


#: Device tensors that went through the DENSE LIBRARY (hipBLASLt / MIOpen) instead of libqt_hip.so, by reason.  The
#: HIP path is built for fp32; other dtypes, grouped / non-zero-padding-mode convs and eval-mode weights that are not
#: a quantised image take the reference expression in torch — visibly: tests assert this counter stays empty on the
#: fp32 paths.
LIBRARY_PATHS: Counter = Counter()


# ---- route switches: module attributes = process-wide DEFAULTS; ``with _fused.scope(GEMM_IMPL="valu"):`` = per-thread overrides ----
# (same mechanism as ops.scope; the autograd Functions re-open the forward's scope around their backward: functions.common.QtFunction)
_SCOPED = ("GEMM_IMPL", "DETECT_BINARY_INPUT", "FLOAT_PATH", "PAD_PLANES", "XNOR_LINEAR_DIGITS", "BWD_CONV_MFMA", "LINEAR_GRAD_X_ONE_PACK",
           "BWD_MFMA_MIN_MACS")
_scope_tls = threading.local()


def _cfg(name: str):
    ov = getattr(_scope_tls, "ov", None)
    if ov is not None and name in ov:
        return ov[name]
    return globals()[name]


def scope_overrides():
    return getattr(_scope_tls, "ov", None)


@contextlib.contextmanager
def scope(_overrides=None, **kw):
    """Thread-local overrides of this module's route switches (names: ``_SCOPED``); nests; ``scope(None)`` is a no-op."""
    if _overrides:
        kw = {**_overrides, **kw}
    bad = [k for k in kw if k not in _SCOPED]
    if bad:
        raise KeyError(f"not a scoped switch of functions._fused: {bad} (known: {_SCOPED})")
    prev = getattr(_scope_tls, "ov", None)
    if kw:
        _scope_tls.ov = {**(prev or {}), **kw}
    try:
        yield
    finally:
        _scope_tls.ov = prev


def note_library_path(input, reason: str) -> None:
    if getattr(input, "is_cuda", False):
        LIBRARY_PATHS[reason] += 1


def _safe_sign_t(w: torch.Tensor) -> torch.Tensor:
    one = torch.ones((), dtype=w.dtype, device=w.device)
    return torch.where(w < 0, -one, one)


def quantize_weight_f32(weight: torch.Tensor, kind: str) -> torch.Tensor:
    """fp32 image of the deterministic weight quantiser of a layer family."""
    hip = weight.is_cuda and weight.dtype == torch.float32      # the kernels are fp32; other dtypes: torch expression
    if kind == "binary":  # safeSign, functions/common.py:4-7
        return ops.binarize(weight) if hip else _safe_sign_t(weight)
    if kind == "ternary":  # functions/terner_connect.py:26-27
        if hip:
            return ops.ternarize(weight)
        s = _safe_sign_t(weight)
        return (s + _safe_sign_t(weight - 0.5 * s)) / 2
    raise ValueError(f"unknown quantiser kind {kind!r}")


#: How a REAL-valued device activation meets a quantised weight: "bf16x3" = exact bf16 triple split +
#: bf16 MFMA GEMM in libqt_hip.so (fp32-GEMM accuracy); "library" = dense fp32 GEMM / conv library on
#: the HIP-quantised weight image (the reference computation itself).
FLOAT_PATH = "bf16x3"

#: Re-express strided few-channel convs (first layers) as stride-1 convs on the space-to-depth image.
USE_S2D = True

#: 'auto' | 'valu' | 'mfma' — packed-GEMM formulation used by the layers (both are bit-exact;
#: 'auto' picks by shape, see ops.select_gemm_impl).
GEMM_IMPL = "auto"
#: packed-activation convs: expand the bits into a physically zero-padded pixel plane (un-padded conv kernel)
PAD_PLANES = True


def pack_weight(weight_2d: torch.Tensor, kind: str, impl: str = "valu"):
    """Packed image (bit planes for 'valu', nibble plane for 'mfma') of the deterministic
    quantiser applied to a [N, K] device weight.  Both quantisers are idempotent, so this is also
    right for an already-quantised (eval / stochastic) weight image."""
    if kind not in ("binary", "ternary"):
        raise ValueError(f"unknown quantiser kind {kind!r}")
    return ops.pack_weights(weight_2d, kind, impl)


# ---- content-dependent routing ----------------------------------------------------------------------------------------
# Two decisions depend on the VALUES of an activation: "is this un-tagged tensor exactly +-1?" (packed route vs the
# exact bf16-triple route) and "do these DoReFa codes fit int8?" (int8 route vs fp32 route; the reference does not
# clamp).  Asking the device costs a host sync per layer per forward (13 % of the module-by-module C4 forward).
#   * A NEGATIVE answer is remembered per (consuming weight, activation shape) in every mode: the general route is
#     correct for any input, so nothing needs to be asked again (e.g. the real-valued first conv of every model).
#   * A POSITIVE answer is re-verified on every call by default (DETECT_MODE = "verify": reference-exact whatever the
#     caller feeds next, one sync per un-tagged input).  DETECT_MODE = "remember" (serving / benchmarking) trusts it and
#     folds THIS call's device flag into the bias instead: if the assumption ever stops holding, the output is NaN —
#     never a plausible wrong number — and reset_detection() re-asks.  That is the contract of the fused code-plane
#     chain (CodeActivation.float()) applied to the module-by-module graph; steady-state forwards then enqueue without
#     any host synchronisation.
# Activations that carry a tag from their quantiser (BinaryConnect, nnDorefaQuant: packed.py) never get here for the
# +-1 question.
DETECT_MODE = "verify"
_detect_tls = threading.local()


def detect_mode() -> str:
    """The detection mode in force in THIS thread: the innermost ``detect_scope(...)``, else the module default DETECT_MODE."""
    return getattr(_detect_tls, "mode", None) or DETECT_MODE


def detect_mode_override() -> Optional[str]:
    return getattr(_detect_tls, "mode", None)


@contextlib.contextmanager
def detect_scope(mode: Optional[str]):
    """``with detect_scope("remember"):`` — thread-local like ops.float_split: a capturing thread (utils.GraphedTrainStep) does not
    change what a serving thread beside it asks the device; autograd Functions re-open the forward's scope around their backward
    (functions.common.QtFunction), which runs on the engine's thread.  None = leave as is."""
    if mode is not None and mode not in ("verify", "remember"):
        raise ValueError(f"DETECT_MODE must be 'verify' or 'remember', got {mode!r}")
    prev = getattr(_detect_tls, "mode", None)
    if mode is not None:
        _detect_tls.mode = mode
    try:
        yield
    finally:
        _detect_tls.mode = prev


_VERDICTS = {}     # id(weight tensor) -> (weak reference to it, {(question, shape): bool}); tensors compare element-wise, so
#                    they cannot key a WeakKeyDictionary
_VERDICTS_LOCK = threading.RLock()   # serving threads share the store; the weak-reference callback may fire inside any of them
#: "sync": decisions that needed a host sync; "cached": decisions taken from a remembered verdict (tests assert on it)
DETECT_STATS: Counter = Counter()


def reset_detection(weight: Optional[torch.Tensor] = None) -> None:
    """Forget the remembered verdicts (of one weight, or all)."""
    with _VERDICTS_LOCK:
        if weight is None:
            _VERDICTS.clear()
        else:
            _VERDICTS.pop(id(weight), None)


def _verdict(weight, tag, resolve):
    """(answer, trusted-from-cache)"""
    mode = detect_mode()
    if mode not in ("verify", "remember"):
        raise ValueError(f"DETECT_MODE must be 'verify' or 'remember', got {mode!r}")
    if weight is None:
        DETECT_STATS["sync"] += 1
        return bool(resolve()), False
    key = id(weight)
    with _VERDICTS_LOCK:
        slot = _VERDICTS.get(key)
        if slot is None or slot[0]() is not weight:
            slot = (weakref.ref(weight, _drop_verdicts(key)), {})
            _VERDICTS[key] = slot
        d = slot[1]
        if tag in d and (d[tag] is False or mode == "remember"):
            DETECT_STATS["cached"] += 1
            return d[tag], True
    answer = bool(resolve())                   # (the device check / host sync runs outside the lock)
    with _VERDICTS_LOCK:
        d[tag] = answer
        DETECT_STATS["sync"] += 1
    return answer, False


def _drop_verdicts(key):
    def drop(_ref):
        with _VERDICTS_LOCK:
            slot = _VERDICTS.get(key)
            if slot is not None and slot[0] is _ref:          # a new tensor may have taken the id meanwhile: keep its slot
                _VERDICTS.pop(key, None)
    return drop


def verdict_is_pm1(input: torch.Tensor, weight: Optional[torch.Tensor]) -> bool:
    """Did the detection for this (weight, activation shape) — run by the forward that just finished — say +-1?  Reads the
    verdict store only (no device check): a forward that packed an un-tagged activation either verified it with a sync or
    armed the device flag that poisons its output, so the backward may contract the same +-1 image."""
    if weight is None:
        return False
    with _VERDICTS_LOCK:
        slot = _VERDICTS.get(id(weight))
        return bool(slot is not None and slot[0]() is weight and slot[1].get(("pm1", tuple(input.shape[1:]))) is True)


_LAST = threading.local()      # .pm1 = (data_ptr, shape, answer) of the detection that ran last in this thread


def last_detection_said_pm1(input: torch.Tensor) -> bool:
    """Did a detection run for exactly THIS activation since ``clear_last_detection()`` and answer +-1?  (The verdict store
    is keyed by weight and shape: a forward that skipped the detection — DETECT_BINARY_INPUT switched off, a non-fp32
    early return — must not inherit an older call's positive answer)"""
    rec = getattr(_LAST, "pm1", None)
    return rec is not None and rec == (input.data_ptr(), tuple(input.shape), True)


def clear_last_detection() -> None:
    _LAST.pm1 = None


def detect_pm1(input: torch.Tensor, weight: Optional[torch.Tensor]):
    """(treat as +-1?, device flag to fold into the bias or None) for an UN-TAGGED device activation."""
    ok, cached = _verdict(weight, ("pm1", tuple(input.shape[1:])), lambda: ops.is_pm1(input))
    _LAST.pm1 = (input.data_ptr(), tuple(input.shape), bool(ok))
    if ok and cached:
        return True, ops.check_pm1(input)       # int32[1] on the device, non-zero = some element is not +-1; no sync
    return ok, None


def codes_route(codes, weight: Optional[torch.Tensor]):
    """(use the int8 code planes?, device flag or None) for DoReFa activation codes whose range flag is on the device."""
    if ops._cfg("ASSUME_CODES_FIT") or codes.overflow is None:
        return True, None
    ok, cached = _verdict(weight, ("codes", int(codes.K)), codes.usable)
    if ok and cached:
        return True, codes.overflow
    return ok, None


def poison_bias(bias: Optional[torch.Tensor], flag: Optional[torch.Tensor], n: int, device) -> Optional[torch.Tensor]:
    """bias (+ NaN if the device flag is raised): one tiny launch instead of a host sync."""
    if flag is None:
        return bias
    if flag.dtype == torch.int32 and flag.is_cuda:
        return ops.poison(bias.detach() if bias is not None else None, flag, -1, n)
    nanv = torch.where(flag.reshape(()) != 0, float("nan"), 0.0)
    base = bias.detach() if bias is not None else torch.zeros((n,), dtype=torch.float32, device=device)
    return base + nanv


def activation_planes(input: torch.Tensor, binary_input: Optional[bool], impl: str = "valu",
                      weight: Optional[torch.Tensor] = None):
    """(packed image of a device activation in the format of ``impl`` if it is (treated as) exactly +-1 else None,
    device flag to fold into the bias or None)."""
    if input.dtype != torch.float32 or input.numel() == 0:
        return None, None
    tagged = packed.lookup(input, packed.ROWS_LAST)
    if tagged is not None:
        K = input.shape[-1]
        if tagged.K == K and tagged.rows * K == input.numel():
            return ops.to_impl(tagged, impl), None   # bit planes from the quantiser (1 bit -> 4 bits if mfma)
    if binary_input is False:
        return None, None
    flag = None
    if binary_input is None and tagged is None:
        if not _cfg("DETECT_BINARY_INPUT"):
            return None, None
        ok, flag = detect_pm1(input, weight)
        if not ok:
            return None, None
    return ops.pack_activations(input, impl), flag


def quant_linear_forward(input: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor],
                         kind: str, weight_q: Optional[torch.Tensor] = None,
                         weight_planes: Optional[ops.BitPlanes] = None,
                         binary_input: Optional[bool] = None) -> torch.Tensor:
    """y = input . Q(weight)^T + bias.

    ``weight_q``: explicit fp32 quantised weight (stochastic draw, or eval-mode pre-quantised
    weight); ``weight_planes``: cached planes of that same quantised weight.
    """
    if not input.is_cuda:
        wq = weight_q if weight_q is not None else quantize_weight_f32(weight, kind)
        return F.linear(input, wq, bias)

    K = input.shape[-1]
    N = weight.shape[0]
    M = input.numel() // max(K, 1)
    impl = ops.select_gemm_impl(_cfg("GEMM_IMPL"), M, N, K)
    if (impl == "mfma" and weight_planes is None and input.dtype == torch.float32 and input.numel() > 0
            and binary_input is not False and packed.lookup(input, packed.ROWS_LAST) is None):
        ok, flag = (True, None) if binary_input else (detect_pm1(input, weight) if _cfg("DETECT_BINARY_INPUT") else (False, None))
        if ok:
            # neither operand is packed yet (un-tagged +-1 activation, training-mode weight): one launch packs both
            wq = weight_q if weight_q is not None else weight
            xp, wp = ops.pack_linear_operands(input, wq.reshape(N, -1), kind, impl)
            return ops.packed_gemm(xp, wp, poison_bias(bias, flag, N, input.device), impl=impl).view(*input.shape[:-1], N)
        binary_input = False                      # verdict known: do not ask again below
    xp, flag = activation_planes(input, binary_input, impl, weight)
    if xp is not None:
        wp = weight_planes
        if wp is None:
            wq = weight_q if weight_q is not None else weight
            wp = pack_weight(wq.reshape(N, -1), kind, impl)
        y = ops.packed_gemm(xp, wp, poison_bias(bias, flag, N, input.device), impl=impl)
        return y.view(*input.shape[:-1], N)

    if _cfg("FLOAT_PATH") == "bf16x3" and input.dtype == torch.float32 and input.numel() > 0:
        # the quantisers are idempotent, so an explicit quantised image (eval / stochastic) goes through
        # the same weight packer
        return ops.float_linear(input, weight_q if weight_q is not None else weight, kind, bias,
                                weight_triples=weight_planes if isinstance(weight_planes, ops.TriplePlanes) else None)
    wq = weight_q if weight_q is not None else quantize_weight_f32(weight, kind)
    note_library_path(input, "non-fp32 or empty linear input")
    return F.linear(input, wq, bias)


def _pixel_planes(input: torch.Tensor, binary_input: Optional[bool], weight: Optional[torch.Tensor] = None, ld_fn=None):
    """(NHWC nibble pixel plane of a device activation that is (treated as) exactly +-1 else None, device flag to fold
    into the bias or None).  ``ld_fn``: words per pixel as a function of the channel count (default ops.pixel_ld_nib)."""
    if input.dtype != torch.float32 or input.dim() != 4 or input.numel() == 0:
        return None, None
    ld_fn = ld_fn or ops.pixel_ld_nib
    tagged = packed.lookup(input, packed.NHWC)
    if tagged is not None:
        N, C, H, W = input.shape
        if tagged.K == C and tagged.rows == N * H * W:
            return ops.bits_to_nib(tagged, ld=ld_fn(C)), None
    if binary_input is False:
        return None, None
    flag = None
    if binary_input is None and tagged is None:
        if not _cfg("DETECT_BINARY_INPUT"):
            return None, None
        ok, flag = detect_pm1(input, weight)
        if not ok:
            return None, None
    return ops.pack_pixels_nib(input, ld=ld_fn(int(input.shape[1]))), flag


def quant_conv2d_forward(input, weight, bias, stride, padding, dilation, groups, kind: str,
                         weight_q: Optional[torch.Tensor] = None, weight_planes=None,
                         binary_input: Optional[bool] = None, padding_mode: str = "zeros",
                         weight_triples_fn=None, epi=None):
    """conv2d(input, Q(weight), bias, ...).

    ``epi`` = (alpha, beta) (inference fusion, layers/fused.py): the conv does not write fp32 but the
    threshold bits [(conv + bias) * alpha + beta < 0]; returns (BitPlanes NHWC pixel plane, (N, Cout, Ho, Wo)).

    Device tensor with +-1 activations, groups == 1, zero padding: NHWC pixel planes -> packed-domain
    im2col -> matrix-core packed GEMM (libqt_hip.so); the result keeps the input's memory format
    (channels_last in -> channels_last out, like torch's own conv).  Otherwise (first layer with real
    pixels, grouped conv): weight quantised by the HIP elementwise kernel, contraction by the dense
    conv library on the exact +-1/0 weight image."""
    packable = (input.is_cuda and groups == 1 and padding_mode == "zeros" and input.dim() == 4
                and not isinstance(padding, str))
    if packable:
        px, flag = _pixel_planes(input, binary_input, weight)
        if px is not None:
            bias = poison_bias(bias, flag, int(weight.shape[0]), input.device)
            wq = weight_q if weight_q is not None else weight
            wp = weight_planes if isinstance(weight_planes, ops.NibPlanes) else ops.pack_conv_weight_nib(wq, kind)
            N, C, H, W = input.shape
            kh, kw = int(weight.shape[2]), int(weight.shape[3])
            y2 = ops.conv2d_nib(px, (N, C, H, W), wp, (kh, kw), bias, stride, padding, dilation, epi=epi)
            Ho, Wo = ops.conv_out_hw(H, W, kh, kw, stride, padding, dilation)
            if epi is not None:
                return y2, (N, int(weight.shape[0]), Ho, Wo)
            y = y2.view(N, Ho, Wo, weight.shape[0]).permute(0, 3, 1, 2)   # NCHW view, NHWC storage
            if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
                y = y.contiguous()                                         # caller works in NCHW storage
            return y
    if packable and _cfg("FLOAT_PATH") == "bf16x3" and input.dtype == torch.float32 and input.numel() > 0:
        # real-valued activation (first layer): exact bf16 triples + implicit-GEMM conv on the bf16
        # matrix cores; the quantisers are idempotent so an explicit quantised image is packed the same way
        N, C, H, W = input.shape
        kh, kw = int(weight.shape[2]), int(weight.shape[3])
        Ho, Wo = ops.conv_out_hw(H, W, kh, kw, stride, padding, dilation)
        if kind in ("binary", "ternary") and ops.first3x3_applicable(C, int(weight.shape[0]), (kh, kw), stride, padding, dilation) and \
                (epi is None or isinstance(epi, (tuple, ops.NibEpilogue))):
            # stride-1 3 x 3 first layer, 64 output channels (VGG-16's conv1_1): the one-pass kernel reads the fp32 image where it lies
            # (per-tile fp16 split in LDS) and serves all three epilogues from one accumulation — fp32 here for the module-by-module
            # execution, threshold bits / the next conv's nibble halo plane for the fused chain: same bits at ties by construction
            fw = weight_triples_fn("first3x3") if weight_triples_fn is not None else None
            if fw is None:
                wq = weight_q if weight_q is not None else quantize_weight_f32(weight, kind)
                fw = ops.pack_first3x3_weight(wq)
            nib_epi = epi if isinstance(epi, ops.NibEpilogue) else None
            e3 = epi if (epi is None or (nib_epi is not None and tuple(nib_epi.out_halo) == (1, 1) and not nib_epi.d2s_cout)) else \
                ((nib_epi.alpha, nib_epi.beta) if nib_epi is not None else epi[:2])
            y2 = ops.conv_first3x3(input, fw, int(weight.shape[0]), bias, epi=e3)
            if y2 is not None:
                if epi is not None:
                    if nib_epi is not None and isinstance(y2, ops.BitPlanes):
                        y2 = ops.bits_to_nib_pad(y2, N, Ho, Wo, nib_epi.out_halo, ld=ops.pixel_ld_nib(y2.K))
                    return y2, (N, int(weight.shape[0]), Ho, Wo)
                y = y2.view(N, Ho, Wo, weight.shape[0]).permute(0, 3, 1, 2)
                if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
                    y = y.contiguous()
                return y
        if ops.first_direct_applicable(C, (kh, kw), stride, padding, dilation):
            # (both epilogues on ONE route: the threshold bits of the fused / deferred chain must come from the very accumulators
            # the fp32 output of the module-by-module execution shows, or the two executions differ at ties.  The kernel's fp32
            # store tail makes the module-by-module conv1 ~100 us slower than the space-to-depth route at AlexNet's shape — not the
            # inference path: tools/bench_conv1.py)
            # strided few-channel first layer (AlexNet conv1): the direct kernel reads the fp32 image where it lies, splits its
            # patch in registers and contracts by stride addressing — no operand pack pass, no space-to-depth plane
            fw = weight_triples_fn("first_direct") if weight_triples_fn is not None else None
            if fw is None:
                wq = weight_q if weight_q is not None else quantize_weight_f32(weight, kind)
                fw = ops.pack_first_layer_weight(wq, ops._pairs(stride)[0])
            nib_epi = epi if isinstance(epi, ops.NibEpilogue) else None
            y2 = ops.conv_first_direct(input, fw, bias, stride, padding,
                                       epi=(nib_epi.alpha, nib_epi.beta) if nib_epi is not None else (epi[:2] if epi is not None else None))
            if y2 is not None:
                if epi is not None:
                    if nib_epi is not None:
                        y2 = ops.bits_to_nib_pad(y2, N, Ho, Wo, nib_epi.out_halo, ld=ops.pixel_ld_nib(y2.K))
                    return y2, (N, int(weight.shape[0]), Ho, Wo)
                y = y2.view(N, Ho, Wo, weight.shape[0]).permute(0, 3, 1, 2)
                if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
                    y = y.contiguous()
                return y
        if USE_S2D and ops.s2d_applicable(C, kh, kw, stride, dilation, padding):
            # strided few-channel conv (conv1) == stride-1 conv on the space-to-depth image; the gather and
            # the exact bf16 split are one kernel, the transformed weight is cached by eval-mode layers
            sd = ops._pairs(stride)[0]
            px, (Hs, Ws) = ops.s2d_triple_pack(input, sd, padding)
            cached = weight_triples_fn("s2d") if weight_triples_fn is not None else None
            if cached is not None:
                ws_shape, wtr = cached
            else:
                wq = weight_q if weight_q is not None else quantize_weight_f32(weight, kind)
                ws = ops.s2d_weight(wq.detach(), sd)
                ws_shape, wtr = tuple(ws.shape), ops.pack_conv_weight_bf16x3(ws, "sign")   # zeros stay zeros
            k2 = ws_shape[2]
            # a nibble-plane epilogue maps output rows with the kernel's own (Hs - k2 + 1, Ws - k2 + 1) geometry: when
            # the space-to-depth rounding added pixels, take threshold bits, crop, and expand afterwards
            nib_epi = epi if isinstance(epi, ops.NibEpilogue) else None
            if nib_epi is not None and (Hs - k2 + 1 != Ho or Ws - k2 + 1 != Wo):
                epi = (nib_epi.alpha, nib_epi.beta)
            y2 = ops.float_conv2d(None, torch.empty(ws_shape, device="meta"), "sign", bias, 1, 0, 1,
                                  weight_triples=wtr, pixels=px, in_shape=(N, C * sd * sd, Hs, Ws), epi=epi)
            H2, W2 = Hs - k2 + 1, Ws - k2 + 1
            if epi is not None:
                if H2 != Ho or W2 != Wo:   # drop the pixels the space-to-depth rounding added
                    sg = y2.sign.view(N, H2, W2, -1)[:, :Ho, :Wo, :].contiguous().view(N * Ho * Wo, -1)
                    y2 = ops.BitPlanes(sign=sg, rows=N * Ho * Wo, K=y2.K)
                if nib_epi is not None and isinstance(y2, ops.BitPlanes):
                    y2 = ops.bits_to_nib_pad(y2, N, Ho, Wo, nib_epi.out_halo, ld=ops.pixel_ld_nib(y2.K))
                return y2, (N, int(weight.shape[0]), Ho, Wo)
            y = y2.view(N, H2, W2, weight.shape[0])[:, :Ho, :Wo, :].permute(0, 3, 1, 2)
            if H2 != Ho or W2 != Wo:
                y = y.contiguous(memory_format=torch.channels_last)
        else:
            wt = weight_triples_fn("plain") if weight_triples_fn is not None else None
            y2 = ops.float_conv2d(input, weight_q if weight_q is not None else weight, kind, bias, stride, padding,
                                  dilation, weight_triples=wt, epi=epi)
            if epi is not None:
                return y2, (N, int(weight.shape[0]), Ho, Wo)
            y = y2.view(N, Ho, Wo, weight.shape[0]).permute(0, 3, 1, 2)
        if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous()
        return y
    if epi is not None:
        raise ValueError("the threshold-bit epilogue needs a device fp32 NCHW input, groups == 1 and zero padding")
    wq = weight_q if weight_q is not None else quantize_weight_f32(weight, kind)
    note_library_path(input, "grouped / non-zero padding mode / non-fp32 conv")
    return F.conv2d(input, wq, bias, stride, padding, dilation, groups)


def real_weight_linear(input: torch.Tensor, weight_q: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """F.linear(input, weight_q, bias) for an ALREADY QUANTISED, real-valued weight image (the deprecated functional
    forms: TernaryDense's torch.sign formula keeps w = +-0.5 as +-0.5, QuantDense's levels are odd integers / (2^k - 1),
    functions/terner_connect.py:85-94, dorefa_connect.py:124-132).  Device fp32 2-D: six-term bf16 planes on the matrix
    cores (ops.real_linear, fp32-GEMM accuracy); everything else: the torch expression."""
    if (input.is_cuda and input.dtype == torch.float32 and weight_q.dtype == torch.float32 and input.dim() == 2
            and input.numel() > 0 and weight_q.numel() > 0):
        return ops.real_linear(input.detach(), weight_q.detach(), bias.detach() if bias is not None else None)
    return F.linear(input, weight_q, bias)


def real_weight_conv2d(input: torch.Tensor, weight_q: torch.Tensor, bias, stride, padding, dilation, groups) -> torch.Tensor:
    """F.conv2d(input, weight_q, ...) for an already quantised real-valued weight image (TernaryConv2d / QuantConv2d /
    XNORConv2d functional forms): implicit-GEMM conv over six-term bf16 planes (ops.real_conv2d) for device fp32 NCHW
    inputs with groups == 1 and numeric zero padding; the result keeps the input's memory format."""
    if (input.is_cuda and input.dtype == torch.float32 and weight_q.dtype == torch.float32 and input.dim() == 4
            and input.numel() > 0 and groups == 1 and not isinstance(padding, str)):
        y2 = ops.real_conv2d(input.detach(), weight_q.detach(), bias.detach() if bias is not None else None, stride,
                             padding, dilation)
        if y2 is not None:
            N_, _, H, W = input.shape
            Ho, Wo = ops.conv_out_hw(H, W, weight_q.shape[2], weight_q.shape[3], stride, padding, dilation)
            y = y2.view(N_, Ho, Wo, weight_q.shape[0]).permute(0, 3, 1, 2)
            if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
                y = y.contiguous()
            return y
    return F.conv2d(input, weight_q, bias=bias, stride=stride, padding=padding, dilation=dilation, groups=groups)


def packed_linear(layer, act, kind: str, hwc=None) -> torch.Tensor:
    """Eval-mode LinearBin / LinearTer on a PackedActivation (row planes): planes -> packed GEMM.  ``hwc`` = (C, H, W):
    the rows are a feature map flattened in (h, w, c) order (PackedActivation.flatten_hwc) while the layer's weight
    expects the NCHW flattening — the packed weight is built from the column-permuted image (cached like the plain one)."""
    if layer.training:
        raise RuntimeError("PackedActivation inputs are an inference feature: call .eval() first")
    N = layer.weight.shape[0]
    M, K = act.planes.rows, act.planes.K
    if K != layer.weight.shape[1]:
        raise ValueError(f"packed activation has {K} features, layer expects {layer.weight.shape[1]}")
    impl = ops.select_gemm_impl(_cfg("GEMM_IMPL"), M, N, K)
    if hwc is None:
        wp = layer._eval_planes(lambda w2: pack_weight(w2, kind, impl), key=impl)
    else:
        from ..layers.fused import permute_fc_weight_hwc
        C, H, W = (int(v) for v in hwc)
        wp = layer._eval_planes(lambda w2: pack_weight(permute_fc_weight_hwc(w2, C, H, W), kind, impl),
                                key=(impl, "hwc", C, H, W))
    y = ops.packed_gemm(ops.to_impl(act.planes, impl), wp, layer.bias, impl=impl)
    return y.view(*act.shape[:-1], N)


def packed_conv2d(layer, act, kind: str, epi=None):
    """Eval-mode BinConv2d / TerConv2d on a PackedActivation (NHWC planes).  Returns a channels_last
    [N, Cout, Ho, Wo] fp32 tensor, or with ``epi`` = (alpha, beta) the threshold-bit planes and shape
    (see quant_conv2d_forward)."""
    if layer.training:
        raise RuntimeError("PackedActivation inputs are an inference feature: call .eval() first")
    if layer.groups != 1 or layer.padding_mode != "zeros":
        raise ValueError("packed conv needs groups == 1 and zero padding")
    if kind == "xnor":
        return packed_xnor_conv2d(layer, act, epi)
    N, C, H, W = act.shape
    wp = layer._eval_planes(lambda _w2: ops.pack_conv_weight_nib(layer.weight.detach(), kind), key="conv_nib")
    kh, kw = int(layer.weight.shape[2]), int(layer.weight.shape[3])
    ph, pw = ops._pairs(layer.padding)
    if act.nib is not None:
        # the producer already wrote this conv's operand: nibble pixel plane with the padding as a physical zero border
        if act.halo != (ph, pw):
            raise ValueError(f"activation carries a {act.halo} halo, this conv pads {(ph, pw)}: re-link the fused modules")
        if ops.direct_conv3x3_applicable(C, int(layer.weight.shape[0]), (kh, kw), layer.stride, layer.padding,
                                         layer.dilation, act.halo, epi):
            y2 = ops.conv3x3_direct_nib(act.nib, N, C, H, W, wp, layer.bias, epi)
            return y2, (N, int(layer.weight.shape[0]), H, W)
        y2 = ops.conv2d_nib(act.nib, (N, C, H + 2 * ph, W + 2 * pw), wp, (kh, kw), layer.bias, layer.stride, 0,
                            layer.dilation, epi=epi)
    elif _cfg("PAD_PLANES") and (ph or pw):
        # zero padding made physical while the bits are expanded: the conv runs un-padded (no per-tap checks)
        px = ops.bits_to_nib_pad(act.planes, N, H, W, (ph, pw), ld=ops.pixel_ld_nib(C))
        y2 = ops.conv2d_nib(px, (N, C, H + 2 * ph, W + 2 * pw), wp, (kh, kw), layer.bias, layer.stride, 0,
                            layer.dilation, epi=epi)
    else:
        px = ops.bits_to_nib(act.planes, ld=ops.pixel_ld_nib(C))
        y2 = ops.conv2d_nib(px, (N, C, H, W), wp, (kh, kw), layer.bias, layer.stride, layer.padding, layer.dilation,
                            epi=epi)
    Ho, Wo = ops.conv_out_hw(H, W, kh, kw, layer.stride, layer.padding, layer.dilation)
    if epi is not None:
        return y2, (N, int(layer.weight.shape[0]), Ho, Wo)
    return y2.view(N, Ho, Wo, layer.weight.shape[0]).permute(0, 3, 1, 2)


def packed_xnor_conv2d(layer, act, epi=None):
    """Eval-mode XNORConv2d on a PackedActivation: the per-tap scaled fp4 conv (ops.conv2d_nib_taps) on the activation's nibble
    plane — the producer's own (zero halo = this conv's padding, un-padded kernel) when its pixel stride is whole 32-byte taps,
    else expanded from the bit planes with the padding made physical.  Same returns as packed_conv2d."""
    N, C, H, W = act.shape
    wp, taps = layer._taps_planes()
    kh, kw = int(layer.weight.shape[2]), int(layer.weight.shape[3])
    ph, pw = ops._pairs(layer.padding)
    Cw = ops.pixel_ld_nib_taps(C)
    if act.nib is not None:
        if act.halo != (ph, pw):
            raise ValueError(f"activation carries a {act.halo} halo, this conv pads {(ph, pw)}: re-link the fused modules")
        px = act.nib
        if px.ld != Cw:       # channels not a multiple of 64: widen the pixel stride to whole 32-byte taps (zero nibbles)
            wide = torch.zeros((int(px.words.shape[0]), Cw), dtype=torch.int32, device=px.device)
            wide[:, :px.ld] = px.words
            px = ops.NibPlanes(words=wide, rows=px.rows, K=px.K)
    else:
        px = ops.bits_to_nib_pad(act.planes, N, H, W, (ph, pw), ld=Cw)
    y2 = ops.conv2d_nib_taps(px, (N, C, H + 2 * ph, W + 2 * pw), wp, (kh, kw), taps.fwd, layer.bias, layer.stride, 0,
                             layer.dilation, epi=epi)
    if y2 is None:
        raise ValueError("XNOR conv outside the per-tap kernel's limits")
    Ho, Wo = ops.conv_out_hw(H, W, kh, kw, layer.stride, layer.padding, layer.dilation)
    if epi is not None:
        return y2, (N, int(layer.weight.shape[0]), Ho, Wo)
    return y2.view(N, Ho, Wo, layer.weight.shape[0]).permute(0, 3, 1, 2)


def packed_xnor_linear(layer, act, hwc=None) -> torch.Tensor:
    """Eval-mode LinearXNOR on a PackedActivation (row planes): y = (x * alpha) . sign(W)^T + b (xnor_connect.py:112-115) with
    x * alpha = +-alpha[k] built from the sign bits as two-term fp16 pairs (qt_bits_alpha_pairs_f16x2) against the replicated
    sign(W) on the fp16 matrix cores.  ``hwc`` = (C, H, W): the rows are a feature map flattened in (h, w, c) order; the pair
    kernel reads the bits in the weight's own NCHW order, so the sum runs in the module graph's order (bit-identical logits)."""
    if layer.training:
        raise RuntimeError("PackedActivation inputs are an inference feature: call .eval() first")
    N = layer.weight.shape[0]
    K = act.planes.K
    if K != layer.weight.shape[1]:
        raise ValueError(f"packed activation has {K} features, layer expects {layer.weight.shape[1]}")

    dg = xnor_linear_digits(layer) if _cfg("XNOR_LINEAR_DIGITS") else None
    if dg is not None and 3 * act.planes.rows * int(dg[0].codes.shape[1]) >= (1 << 31):
        dg = None                                               # the stacked digit planes pass the GEMM's 32-bit operand offsets
    if dg is not None:
        # integer form: alpha as three 7-bit digits against the int8 codes of sign(W) — exact partial sums, 1 byte per weight
        y = ops.xnor_digit_linear(act.planes, dg[1], dg[0], layer.bias, hwc=hwc)
    else:
        wt, ap = xnor_linear_operands(layer)
        y = ops.bf16_gemm(ops.bits_alpha_pairs(act.planes, ap, hwc=hwc), wt, layer.bias)
    return y.view(*act.shape[:-1], N)


#: LinearXNOR on packed +-1 activations: True = the digit-plane int8 form (ops.xnor_digit_linear), False = fp16 pairs of +-alpha
XNOR_LINEAR_DIGITS = True


def xnor_linear_digits(layer):
    """(int8 codes of sign(W), digit table of alpha[K] = mean(|W|, 0)) of an eval-mode LinearXNOR weight, cached per weight
    version; None when alpha has a non-finite entry or K is beyond the exact range (the pair route then reproduces the NaN / inf
    the reference would produce)."""
    def build(w2):
        ld = ops.code_ld_bytes(int(w2.shape[1]))
        if 127 * ld >= (1 << 24) or int(w2.shape[0]) * ld >= (1 << 31):       # exact fp32 partial sums; 32-bit operand offsets
            return None
        _, alpha = ops.xnor_weight(w2.contiguous(), 1)
        dg = ops.alpha_digits(alpha.view(-1))
        if dg is None:
            return None
        return ops.weight_codes(torch.sign(w2), ternary=True), dg         # torch.sign: 0 -> 0 (xnor_connect.py:113)
    return layer._eval_planes(build, key="xnor_digits")


def xnor_linear_operands(layer=None, weight=None):
    """(fp16 pair planes of sign(W), pair image of alpha[K] = mean(|W|, 0)) of a LinearXNOR weight: cached per weight version
    for an eval-mode ``layer``, built on the spot for a training-mode ``weight``."""
    def build(w2):
        _, alpha = ops.xnor_weight(w2.contiguous(), 1)                       # alpha[1, K] (eval: of the already quantised image)
        return ops.weight_bf16x3(w2, "sign", terms=2), ops.alpha_pairs(alpha.view(-1))
    if layer is not None:
        return layer._eval_planes(build, key="xnor_pairs")
    return build(weight.detach())


#: dispatch used by the layers' forward for PackedActivation inputs: [is_linear] -> function
PACKED_FWD = {True: packed_linear, False: packed_conv2d}


def _dense(t: torch.Tensor) -> torch.Tensor:
    """``t`` itself when it is dense in NCHW or channels-last order (the conv kernels take element strides), else an NCHW
    copy.  A plain ``.contiguous()`` would transpose every channels-last gradient back to NCHW."""
    if t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)):
        return t
    return t.contiguous()


class QuantConv2dFn(QtFunction):
    """Autograd node of BinConv2d / TerConv2d in training mode: forward F.conv2d(x, Q(W), b, ...)
    (layers/binary_layers.py:105); backward = what autograd derives from F.conv2d plus the STE mask
    of the weight quantiser."""

    @staticmethod
    def forward(ctx, input, weight, bias, kind, weight_q, binary_input, conv_args):
        ctx.kind, ctx.conv_args, ctx.has_bias = kind, conv_args, bias is not None
        ctx.save_for_backward(input, weight, weight_q)
        stride, padding, dilation, groups = conv_args
        # +-1 activation known without a device check (tag of a quantiser, or the layer's binary_input hint)?
        ctx.x_is_pm1 = bool(binary_input) or (input.is_cuda and input.dtype == torch.float32 and input.dim() == 4
                                              and packed.lookup(input, packed.NHWC) is not None)
        clear_last_detection()
        out = quant_conv2d_forward(input, weight, bias, stride, padding, dilation, groups, kind,
                                   weight_q=weight_q, binary_input=binary_input)
        if not ctx.x_is_pm1 and binary_input is None and input.is_cuda and input.dim() == 4:
            # un-tagged activation THIS forward detected as +-1 (e.g. behind a MaxPool2d): same knowledge for the backward
            ctx.x_is_pm1 = last_detection_said_pm1(input)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        from .common import ste_mask
        input, weight, weight_q = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.conv_args
        grad_input = grad_weight = grad_bias = None
        go = _dense(grad_output)
        mfma = (_cfg("BWD_CONV_MFMA") and go.is_cuda and go.dtype == torch.float32 and groups == 1 and not isinstance(padding, str)
                and go.numel() * weight[0].numel() >= _cfg("BWD_MFMA_MIN_MACS"))
        if ctx.needs_input_grad[0]:
            if mfma:     # real gradient x +-1 / 0 weight: the forward's exact-split conv on flipped weights (the deterministic
                #          quantiser runs inside the operand pack; a stochastic draw was saved)
                grad_input = (ops.conv2d_grad_input_q(input.shape, weight_q, go, stride, padding, dilation) if weight_q is not None
                              else ops.conv2d_grad_input_q(input.shape, weight, go, stride, padding, dilation, kind=ctx.kind))
            if grad_input is None:
                note_library_path(go, "conv grad_input outside the matrix-core route")
                wq = weight_q if weight_q is not None else quantize_weight_f32(weight, ctx.kind)
                grad_input = torch.nn.grad.conv2d_input(input.shape, wq, go, stride=stride, padding=padding,
                                                        dilation=dilation, groups=groups)
        want_bias = ctx.has_bias and ctx.needs_input_grad[2]
        bias_by_product = []              # filled by the weight-gradient route when its gradient pack can sum the channels on the way
        if ctx.needs_input_grad[1]:
            gw = None
            if mfma and ctx.x_is_pm1:     # +-1 activation x real gradient, contraction over the pixels
                if ops.wgrad_pm_applicable(input.shape, go.shape, weight.shape[2:], stride, dilation):
                    # pixel-major kernel: every tap of a tile in one workgroup; the STE mask of the weight quantiser is the
                    # epilogue of its reduce step
                    grad_weight = ops.conv2d_grad_weight_pm(input, go, weight.shape[2:], padding, weight=weight,
                                                            bias_grad=bias_by_product if want_bias else None)
                if grad_weight is None and ops.wgrad_gemm_applicable(input.shape, go.shape, weight.shape[2:], stride, dilation):
                    # batched bf16 GEMMs over K-major planes (other kernel sizes)
                    grad_weight = ops.conv2d_grad_weight_gemm(input, grad_output, weight.shape[2:], padding, weight=weight)
                if grad_weight is None and ops.wgrad_strided_applicable(input.shape, go.shape, weight.shape[2:], stride, padding,
                                                                        dilation):
                    # strided convs (ResNet stage transitions): 1x1 -> GEMM over the sub-sampled positions, 3x3 -> the
                    # pixel-major kernel on the space-to-depth image; STE mask below
                    gw = ops.conv2d_grad_weight_strided(input, go, weight.shape[2:], stride, padding)
                if grad_weight is None and gw is None:
                    gw = ops.conv2d_grad_weight_pm1(input, go, weight.shape[2:], stride, padding, dilation)
            if (mfma and grad_weight is None and gw is None and not ctx.x_is_pm1
                    and ops.wgrad_s2d_applicable(input.shape, weight.shape[2:], stride, dilation)):
                # strided first layer over a real-valued image: space-to-depth + the pixel-major kernel
                grad_weight = ops.conv2d_grad_weight_s2d(input, go, weight.shape, stride, padding, weight=weight,
                                                         bias_grad=bias_by_product if want_bias else None)
            if grad_weight is None:
                if gw is None:
                    note_library_path(go, "conv grad_weight outside the matrix-core route")
                    gw = torch.nn.grad.conv2d_weight(input, weight.shape, go, stride=stride, padding=padding,
                                                     dilation=dilation, groups=groups)
                grad_weight = ste_mask(gw.contiguous(), weight)
        if want_bias:
            grad_bias = bias_by_product[0] if bias_by_product else go.sum((0, 2, 3))
        return grad_input, grad_weight, grad_bias, None, None, None, None


def pm1_conv_grad_weight(input, go, weight_shape, stride, padding, dilation, bias_by_product=None):
    """UN-masked grad wrt the weight of conv2d(x, .) for a +-1 / 0 activation x on this backend's weight-gradient routes
    (pixel-major kernel, K-major batched GEMMs, strided forms, swapped conv); None when no route takes the shape.  The callers
    apply their quantiser's backward to it (STE mask: QuantConv2dFn; the XNOR-Net combination: xnor_connect.py:158-159)."""
    ksz = weight_shape[2:]
    gw = None
    if ops.wgrad_pm_applicable(input.shape, go.shape, ksz, stride, dilation):
        gw = ops.conv2d_grad_weight_pm(input, go, ksz, padding, weight=None, bias_grad=bias_by_product)
    if gw is None and ops.wgrad_gemm_applicable(input.shape, go.shape, ksz, stride, dilation):
        gw = ops.conv2d_grad_weight_gemm(input, go, ksz, padding, weight=None)
    if gw is None and ops.wgrad_strided_applicable(input.shape, go.shape, ksz, stride, padding, dilation):
        gw = ops.conv2d_grad_weight_strided(input, go, ksz, stride, padding)
    if gw is None:
        gw = ops.conv2d_grad_weight_pm1(input, go, ksz, stride, padding, dilation)
    return gw


# ---- backward of the functional (fused Function) forms ------------------------------------------------------------------------------
# BinaryDense / BinaryConv2d() / TernaryDense / TernaryConv2d() / QuantDense / QuantConv2d / XNORDense / XNORConv2d write their
# backward by hand (functions/binary_connect.py:100-112,133-146, terner_connect.py:97-152, dorefa_connect.py:140-199,
# xnor_connect.py:118-168): two contractions per layer.  These helpers are the ONLY place a dense-library call (hipBLASLt GEMM,
# MIOpen backward conv) can come from, and every such call on a device tensor is counted in LIBRARY_PATHS — the functions
# themselves contain no `.mm(` / `torch.nn.grad.` (tests/test_layers_cpu.py greps for it).

def lib_mm(a: torch.Tensor, b: torch.Tensor, reason: str = "backward GEMM on the dense library") -> torch.Tensor:
    """a @ b on the dense library: host tensors (the reference expression itself), non-fp32 device tensors; counted for device
    tensors."""
    note_library_path(a, reason)
    return a.mm(b)


def lib_conv2d_input(input_shape, weight_q, grad_output, stride, padding, dilation, groups, reason="conv grad_input outside the matrix-core route"):
    note_library_path(grad_output, reason)
    return torch.nn.grad.conv2d_input(input_shape, weight_q, grad_output, stride=stride, padding=padding, dilation=dilation,
                                      groups=groups)


def lib_conv2d_weight(input, weight_shape, grad_output, stride, padding, dilation, groups, reason="conv grad_weight outside the matrix-core route"):
    note_library_path(grad_output, reason)
    return torch.nn.grad.conv2d_weight(input, weight_shape, grad_output, stride=stride, padding=padding, dilation=dilation,
                                       groups=groups)


def _hip2d(*ts) -> bool:
    return all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.numel() > 0 for t in ts)


def known_pm1(input: torch.Tensor, weight: Optional[torch.Tensor], layout) -> bool:
    """Is this device activation exactly +-1?  The tag of a quantiser (BinaryConnect) answers without a device check; otherwise
    the detection the layers use (one read of the tensor; negative verdicts are remembered per weight)."""
    if not (isinstance(input, torch.Tensor) and input.is_cuda and input.dtype == torch.float32 and input.numel() > 0):
        return False
    if packed.lookup(input, layout) is not None:
        return True
    return bool(_cfg("DETECT_BINARY_INPUT") and detect_pm1(input, weight)[0])


def real_matmul(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a [M, J] @ b [J, K] for two REAL device matrices on the bf16 matrix cores (six-term planes, fp32-GEMM accuracy); host /
    non-fp32 tensors: the library."""
    if _hip2d(a, b):
        return ops.real_linear(a.contiguous(), b.t().contiguous())
    return lib_mm(a, b)


def dense_grad_input(grad_output: torch.Tensor, weight_q: torch.Tensor, pm1: bool) -> torch.Tensor:
    """g . weight_q for the quantised image the forward multiplied with: +-1 / 0 images (``pm1``) through the exact split of g
    against the replicated image, real-valued images (DoReFa levels, sign(W) * E) through the six-term real x real route."""
    if _hip2d(grad_output, weight_q):
        g2 = grad_output.contiguous()
        if pm1:
            return ops.float_linear(g2, weight_q.t().contiguous(), "sign")
        return ops.real_linear(g2, weight_q.t().contiguous())
    return lib_mm(grad_output, weight_q)


def dense_grad_weight(grad_output: torch.Tensor, input: torch.Tensor, x_is_pm1: bool) -> torch.Tensor:
    """g^T . x: a +-1 activation is the exact operand (rows of g^T are output features: three exact bf16 terms); two real operands
    take the six-term route."""
    if _hip2d(grad_output, input):
        gt = grad_output.t().contiguous()
        if x_is_pm1:
            return ops.float_linear(gt, input.t().contiguous(), "sign", terms=3)
        return ops.real_linear(gt, input.t().contiguous())
    return lib_mm(grad_output.t(), input)


def conv_grad_input(input_shape, weight_q, grad_output, stride, padding, dilation, groups, kind="raw", out_scale=1.0,
                    out_scale_dev=None):
    """grad wrt the input of conv2d(x, weight_q) for an image that is exact in fp16 / bf16 (``kind`` "sign" / "binary" /
    "ternary": +-1 / 0; "raw": small integers or multiples of 1/2, the caller scales) on ops.conv2d_grad_input_q; anything else
    (groups, dilation, a real-valued image) on the library, counted."""
    go = _dense(grad_output)
    if (_cfg("BWD_CONV_MFMA") and kind is not None and go.is_cuda and go.dtype == torch.float32 and groups == 1
            and not isinstance(padding, str) and go.numel() > 0):
        gx = ops.conv2d_grad_input_q(input_shape, weight_q, go, stride, padding, dilation, kind=kind, out_scale=out_scale,
                                     out_scale_dev=out_scale_dev)
        if gx is not None:
            return gx
    wq = weight_q
    if kind in ("binary", "ternary"):
        wq = quantize_weight_f32(weight_q, kind)
    if out_scale_dev is not None or out_scale != 1.0:
        wq = wq * (float(out_scale) if out_scale_dev is None else out_scale_dev * float(out_scale))
    return lib_conv2d_input(input_shape, wq, grad_output, stride, padding, dilation, groups)


def conv_grad_weight(input, weight_shape, grad_output, stride, padding, dilation, groups, x_is_pm1: bool, bias_by_product=None,
                     real_any_channels: bool = False):
    """UN-masked grad wrt the weight of conv2d(x, .): +-1 activations on the weight-gradient routes, a real-valued image with few
    channels (first layers) through the space-to-depth form; anything else on the library, counted."""
    go = _dense(grad_output)
    if (_cfg("BWD_CONV_MFMA") and go.is_cuda and go.dtype == torch.float32 and groups == 1 and not isinstance(padding, str)
            and go.numel() > 0 and input.dtype == torch.float32):
        gw = None
        if x_is_pm1:
            gw = pm1_conv_grad_weight(input, go, weight_shape, stride, padding, dilation, bias_by_product)
        elif ops.wgrad_s2d_applicable(input.shape, weight_shape[2:], stride, dilation):
            gw = ops.conv2d_grad_weight_s2d(input, go, weight_shape, stride, padding, weight=None, bias_grad=bias_by_product)
        elif real_any_channels and ops.wgrad_s2d_applicable(input.shape, weight_shape[2:], stride, dilation, True):
            gw = ops.conv2d_grad_weight_s2d(input, go, weight_shape, stride, padding, weight=None, bias_grad=bias_by_product,
                                            any_channels=True)
        if gw is None and real_any_channels:
            gw = real_conv_grad_weight_taps(input, go, weight_shape, stride, padding, dilation)
        if gw is not None:
            return gw
    return lib_conv2d_weight(input, weight_shape, grad_output, stride, padding, dilation, groups)


def real_conv_grad_weight_taps(input, go, weight_shape, stride, padding, dilation):
    """grad wrt the weight of conv2d(x, .) for two REAL operands of any geometry (groups == 1), one six-term real x real GEMM per
    filter tap on the bf16 matrix cores:  dW[:, :, i, j] = G^T . X_ij  with G [n ho wo, Cout] the gradient and X_ij [n ho wo, Cin] the
    tap-shifted, strided view of the zero-padded input.  The general fallback of the real-valued layers (DoReFa at 8 < k <= 32, the
    quantised input of XNORConv2d) where the pixel-major kernel's geometry (stride 1, 3 x 3 / 5 x 5) does not apply: kh * kw launches
    of K = N Ho Wo, fp32-GEMM accuracy, no dense library.  None for host tensors / empty operands."""
    if not (go.is_cuda and input.is_cuda and go.dtype == torch.float32 and input.dtype == torch.float32 and go.numel() > 0
            and input.numel() > 0):
        return None
    Cout, Cin, kh, kw = (int(v) for v in weight_shape)
    (sh, sw), (ph, pw), (dh, dw) = ops._pairs(stride), ops._pairs(padding), ops._pairs(dilation)
    N_, _, Ho, Wo = (int(v) for v in go.shape)
    x = input.detach()
    if ph or pw:
        x = F.pad(x, (pw, pw, ph, ph))
    g2t = go.detach().permute(1, 0, 2, 3).reshape(Cout, -1).contiguous()                 # [Cout, n ho wo]
    gw = torch.empty((Cout, Cin, kh, kw), dtype=torch.float32, device=go.device)
    for i in range(kh):
        for j in range(kw):
            xs = x[:, :, i * dh:i * dh + sh * (Ho - 1) + 1:sh, j * dw:j * dw + sw * (Wo - 1) + 1:sw]
            x2t = xs.permute(1, 0, 2, 3).reshape(Cin, -1).contiguous()                   # [Cin, n ho wo]
            gw[:, :, i, j] = ops.real_linear(g2t, x2t)                                    # G^T . X_ij  = g2t . x2t^T
    return gw


# ---- XNOR-Net family (functions/xnor_connect.py:93-169, layers/xnor_layers.py) ------------------------------------------------------

def xnor_conv_fast_applicable(input, weight, dim, groups, padding) -> bool:
    """The per-tap scaled route: device fp32 NCHW input, groups == 1, numeric zero padding, and the scale reduced over exactly the
    first two weight dimensions (``dim`` = [0, 1], the default of the layer, the converter and the function: one alpha per tap)."""
    return (isinstance(input, torch.Tensor) and input.is_cuda and input.dtype == torch.float32 and input.dim() == 4
            and input.numel() > 0 and weight.dtype == torch.float32 and weight.dim() == 4 and groups == 1
            and not isinstance(padding, str) and not isinstance(dim, int) and sorted(int(d) for d in dim) == [0, 1])


def xnor_conv2d_forward(input, weight, bias, stride, padding, dilation, binary_input=None, planes=None, epi=None):
    """conv2d(x, sign(W) * alpha[1, 1, kh, kw]) for a +-1 activation on the fp4 matrix cores with per-tap scaling
    (qt_conv2d_implicit_taps).  ``planes`` = (weight nibble planes, TapScales) cached by an eval-mode layer.
    Returns (result, TapScales, pixel words per channel) with result = the NHWC fp32 matrix [N*Ho*Wo, Cout] (or the epilogue's
    planes), or None when the activation is not +-1 / the shape is outside the kernel's limits (the caller takes the real x real
    route)."""
    px, flag = _pixel_planes(input, binary_input, weight, ld_fn=ops.pixel_ld_nib_taps)
    if px is None:
        return None
    if planes is not None:
        wp, taps = planes
    else:
        taps = ops.xnor_tap_prep(weight)
        wp = ops.pack_conv_weight_nib(weight.detach(), "sign", cw=px.ld)
    N, C, H, W = (int(v) for v in input.shape)
    kh, kw = int(weight.shape[2]), int(weight.shape[3])
    b = poison_bias(bias.detach() if bias is not None else None, flag, int(weight.shape[0]), input.device)
    y2 = ops.conv2d_nib_taps(px, (N, C, H, W), wp, (kh, kw), taps.fwd, b, stride, padding, dilation, epi=epi)
    if y2 is None:
        return None
    return y2, taps


def xnor_weight_grad(gw: torch.Tensor, weight: torch.Tensor, mean: torch.Tensor, reduce_dim=0) -> torch.Tensor:
    """mean * gw + sign(W) * mean(gw * sign(W), DIM, keepdim) (functions/xnor_connect.py:126-127, 158-159; DIM = 0 upstream)."""
    sgn = torch.sign(weight)
    return mean * gw + sgn * torch.mean(gw * sgn, reduce_dim, keepdim=True)


def _dorefa_w1_scale(weight: torch.Tensor, prequantized: bool) -> torch.Tensor:
    """E of the 1-bit DoReFa weight sign(W)*E as a device scalar.  Training: E = mean|W|
    (functions/dorefa_connect.py:100).  Eval: the weight already holds sign(W)*E, so |w| == E for
    every entry and amax recovers it exactly."""
    return weight.detach().abs().amax() if prequantized else ops.abs_mean(weight)


def dorefa_w1_linear_forward(input, weight, bias, prequantized: bool, weight_codes=None, scale=None,
                             route: Optional[list] = None):
    """LinearDorefa(bit_width=1).forward on a device tensor (layers/dorefa_layers.py:41-45):
    y = x . (sign(W)*E)^T + b.  If the activation carries k-bit DoReFa codes (nnDorefaQuant output)
    the contraction runs on the int8 matrix cores: y = (E/n) * sum q*s + b; otherwise the dense GEMM
    library on the HIP-quantised weight image."""
    E = scale if scale is not None else _dorefa_w1_scale(weight, prequantized)
    K, N = input.shape[-1], weight.shape[0]
    if isinstance(input, packed.CodeActivation):
        codes = input.codes
        if len(input.shape) != 2 or codes.K != K or 127 * K >= (1 << 24):
            raise ValueError(f"CodeActivation {input.shape} does not fit LinearDorefa({K}, {N})")
        wc = weight_codes if weight_codes is not None else ops.weight_codes(weight.detach().reshape(N, -1))
        y = ops.i8_gemm(codes, wc, codes.inv_n, bias, scale_dev=E)
        y._qt_overflow = codes.overflow
        return y
    codes = packed.lookup_codes(input, packed.ROWS_LAST) if input.dtype == torch.float32 else None
    if codes is not None and codes.K == K and codes.rows * K == input.numel() and 127 * K < (1 << 24):
        ok, flag = codes_route(codes, weight)
        if ok:
            wc = weight_codes if weight_codes is not None else ops.weight_codes(weight.detach().reshape(N, -1))
            y = ops.i8_gemm(codes, wc, codes.inv_n, poison_bias(bias, flag, N, input.device), scale_dev=E)
            if route is not None:
                route.append("codes")
            return y.view(*input.shape[:-1], N)
    if _cfg("FLOAT_PATH") == "bf16x3" and input.dtype == torch.float32 and input.numel() > 0:
        # no usable int8 codes: exact bf16 split of the activation x sign(W) on the bf16 matrix cores, E and bias after
        y = ops.float_linear(input.detach(), weight.detach(), "binary") * E
        if bias is not None:
            y = y + bias.detach()
        if route is not None:
            route.append("split")
        return y
    wq = weight if prequantized else quantize_weight_f32(weight.detach(), "binary") * E
    note_library_path(input, "DoReFa activation without usable int8 codes")
    if route is not None:
        route.append("library")
    return F.linear(input, wq, bias)


def dorefa_w1_conv_forward(input, weight, bias, conv_args, prequantized: bool, weight_codes=None,
                           padding_mode: str = "zeros", scale=None, epi=None, route: Optional[list] = None):
    """DorefaConv2d(bit_width=1).forward on a device tensor (layers/dorefa_layers.py:77-82).  ``route``: a list that
    receives "codes" (int8 matrix cores on the activation's codes), "split" (exact bf16 split of a real-valued — or
    int8-overflowing — activation x sign(W), scaled by E) or "library"."""
    stride, padding, dilation, groups = conv_args
    E = scale if scale is not None else _dorefa_w1_scale(weight, prequantized)
    if isinstance(input, packed.CodeActivation):
        # fused inference: the activation exists only as codes (range checked once, at the end of the network)
        codes = input.codes
        N_, C, H, W = input.shape
        kh, kw = int(weight.shape[2]), int(weight.shape[3])
        if (groups != 1 or padding_mode != "zeros" or isinstance(padding, str) or codes.K != C
                or C != weight.shape[1] or 127 * kh * kw * codes.codes.shape[1] >= (1 << 24)):
            raise ValueError(f"CodeActivation {input.shape} does not fit this DorefaConv2d")
        wc = weight_codes if weight_codes is not None else ops.pack_conv_weight_codes(weight.detach())
        if epi is not None and epi.overflow is None:
            epi.overflow = codes.overflow
        if epi is not None and ops.direct_conv3x3_codes_applicable(C, int(weight.shape[0]), (kh, kw), stride, padding,
                                                                   dilation, input.halo, epi):
            y2 = ops.conv3x3_direct_codes(codes, N_, C, H, W, wc, codes.inv_n, bias, E, epi)
            return packed.CodeActivation(y2, (N_, weight.shape[0], H, W), halo=(1, 1))
        y2 = ops.conv2d_codes(codes, (N_, C, H, W), wc, (kh, kw), codes.inv_n, bias, stride, padding, dilation,
                              scale_dev=E, epi=epi, in_halo=input.halo)
        Ho, Wo = ops.conv_out_hw(H, W, kh, kw, stride, padding, dilation)
        if epi is not None:
            return packed.CodeActivation(y2, (N_, weight.shape[0], Ho, Wo), halo=epi.out_halo)
        y = y2.view(N_, Ho, Wo, weight.shape[0]).permute(0, 3, 1, 2)
        y._qt_overflow = codes.overflow          # the chain's range flag rides on to the next fused quantiser
        return y
    if epi is not None:
        raise TypeError("the code epilogue takes a CodeActivation input")
    codes = None
    if (input.dtype == torch.float32 and input.dim() == 4 and groups == 1 and padding_mode == "zeros"
            and not isinstance(padding, str)):
        codes = packed.lookup_codes(input, packed.NHWC)
    if codes is not None:
        N_, C, H, W = input.shape
        kh, kw = int(weight.shape[2]), int(weight.shape[3])
        Kb = kh * kw * codes.codes.shape[1]
        ok, flag = codes_route(codes, weight) if (codes.K == C and codes.rows == N_ * H * W and 127 * Kb < (1 << 24)) else (False, None)
        if ok:
            wc = weight_codes if weight_codes is not None else ops.pack_conv_weight_codes(weight.detach())
            y2 = ops.conv2d_codes(codes, (N_, C, H, W), wc, (kh, kw), codes.inv_n,
                                  poison_bias(bias, flag, int(weight.shape[0]), input.device), stride, padding,
                                  dilation, scale_dev=E)
            Ho, Wo = ops.conv_out_hw(H, W, kh, kw, stride, padding, dilation)
            if route is not None:
                route.append("codes")
            return y2.view(N_, Ho, Wo, weight.shape[0]).permute(0, 3, 1, 2)   # channels_last like the input
    if (_cfg("FLOAT_PATH") == "bf16x3" and input.dtype == torch.float32 and input.dim() == 4 and input.numel() > 0 and groups == 1
            and padding_mode == "zeros" and not isinstance(padding, str)):
        # no int8 codes (an un-quantised input, or the un-clamped quantiser left the int8 range): the activation is a real
        # number for the matrix cores — exact bf16 split x sign(W), E and the bias applied to the result
        N_, _, H, W = input.shape
        kh, kw = int(weight.shape[2]), int(weight.shape[3])
        Ho, Wo = ops.conv_out_hw(H, W, kh, kw, stride, padding, dilation)
        y2 = ops.float_conv2d(input.detach(), weight.detach(), "binary", bias.detach() if bias is not None else None, stride,
                              padding, dilation, out_scale_dev=E)
        y = y2.view(N_, Ho, Wo, weight.shape[0]).permute(0, 3, 1, 2)
        if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
            y = y.contiguous()
        if route is not None:
            route.append("split")
        return y
    wq = weight if prequantized else quantize_weight_f32(weight.detach(), "binary") * E
    note_library_path(input, "DoReFa activation without usable int8 codes")
    if route is not None:
        route.append("library")
    return F.conv2d(input, wq, bias, stride, padding, dilation, groups)


def _inv_levels(bit_width: int) -> float:
    """fl32(1 / (2^k - 1)) as _quantize forms it (functions/dorefa_connect.py:24)."""
    n = float((1 << int(bit_width)) - 1)
    return float(torch.tensor(1.0, dtype=torch.float32) / torch.tensor(n, dtype=torch.float32))


def dorefa_wk_linear_forward(input, weight_q, bias, bit_width: int, weight_codes):
    """Eval-mode LinearDorefa(bit_width = k, 2 <= k <= 7) on a device activation that carries DoReFa codes:
    with x = q / n_a and w_q = c / n_w (c = odd integer level, functions/dorefa_connect.py:108-112)
        y = (1 / (n_a n_w)) * sum q c + b
    on the int8 matrix cores (int32 accumulate).  The reference sums fl(x) * fl(w_q) in fp32, so parity is the
    float-tail tolerance, not bit equality.  Returns None when the activation has no usable codes."""
    codes = packed.lookup_codes(input, packed.ROWS_LAST) if input.dtype == torch.float32 else None
    K, N = input.shape[-1], weight_q.shape[0]
    n_w = (1 << int(bit_width)) - 1
    if codes is None or codes.K != K or codes.rows * K != input.numel() or 127 * n_w * K >= (1 << 31):
        return None
    ok, flag = codes_route(codes, weight_q)
    if not ok:
        return None
    y = ops.i8_gemm(codes, weight_codes, codes.inv_n * _inv_levels(bit_width), poison_bias(bias, flag, N, input.device),
                    max_abs_code=127 * n_w)
    return y.view(*input.shape[:-1], N)


def dorefa_wk_conv_forward(input, weight_q, bias, conv_args, bit_width: int, weight_codes, padding_mode="zeros"):
    """Eval-mode DorefaConv2d(bit_width = k, 2 <= k <= 7) on a code-carrying activation; see
    dorefa_wk_linear_forward.  Returns None when the packed path does not apply."""
    stride, padding, dilation, groups = conv_args
    if not (input.dtype == torch.float32 and input.dim() == 4 and groups == 1 and padding_mode == "zeros"
            and not isinstance(padding, str)):
        return None
    codes = packed.lookup_codes(input, packed.NHWC)
    if codes is None:
        return None
    N_, C, H, W = input.shape
    kh, kw = int(weight_q.shape[2]), int(weight_q.shape[3])
    Kb = kh * kw * codes.codes.shape[1]
    n_w = (1 << int(bit_width)) - 1
    if codes.K != C or codes.rows != N_ * H * W or 127 * n_w * Kb >= (1 << 31):
        return None
    ok, flag = codes_route(codes, weight_q)
    if not ok:
        return None
    y2 = ops.conv2d_codes(codes, (N_, C, H, W), weight_codes, (kh, kw), codes.inv_n * _inv_levels(bit_width),
                          poison_bias(bias, flag, int(weight_q.shape[0]), input.device),
                          stride, padding, dilation, max_abs_code=127 * n_w)
    Ho, Wo = ops.conv_out_hw(H, W, kh, kw, stride, padding, dilation)
    return y2.view(N_, Ho, Wo, weight_q.shape[0]).permute(0, 3, 1, 2)   # channels_last like the input


#: k-bit DoReFa weights whose integer levels c = (2^k - 1) w_q (odd, |c| <= 2^k - 1) are exact in BOTH split formats (bf16: 8
#: significant bits, fp16: 11): the level image is the exact operand of the real-activation routes up to this bit width
LEVEL_MAX_BITS = 8
#: ... and fit the int8 matrix cores (|c| <= 127) up to this one
LEVEL_INT8_BITS = 7


def dorefa_levels_conv_forward(input, weight_q, bias, conv_args, bit_width: int, level_planes=None):
    """conv2d(x, w_q, b) for a k-bit DoReFa weight image w_q = c / n_w (2 <= k <= LEVEL_MAX_BITS) and ANY fp32 activation —
    real-valued, or a k-bit image whose codes left the int8 range, or 8-bit weights whose levels do not fit int8: the split
    activation x the exact level image on the matrix cores, 1 / n_w and the bias in the epilogue
    (layers/dorefa_layers.py:77-82).  Returns None outside the route's limits (groups, padding mode, empty)."""
    stride, padding, dilation, groups = conv_args
    if not (input.is_cuda and input.dim() == 4 and input.dtype == torch.float32 and groups == 1 and not isinstance(padding, str)
            and input.numel() > 0 and 2 <= int(bit_width) <= LEVEL_MAX_BITS):
        return None
    N_, _, H, W = input.shape
    Ho, Wo = ops.conv_out_hw(H, W, int(weight_q.shape[2]), int(weight_q.shape[3]), stride, padding, dilation)
    y2 = ops.float_conv2d(input.detach(), _weight_levels(weight_q, bit_width), "raw",
                          bias.detach() if bias is not None else None, stride, padding, dilation,
                          weight_triples=level_planes, out_scale=_inv_levels(bit_width))
    y = y2.view(N_, Ho, Wo, weight_q.shape[0]).permute(0, 3, 1, 2)
    if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
        y = y.contiguous()
    return y


def dorefa_levels_linear_forward(input, weight_q, bias, bit_width: int, level_planes=None):
    """The dense twin of dorefa_levels_conv_forward (layers/dorefa_layers.py:41-45)."""
    if not (input.is_cuda and input.dtype == torch.float32 and input.numel() > 0 and 2 <= int(bit_width) <= LEVEL_MAX_BITS):
        return None
    y = ops.float_linear(input.detach(), _weight_levels(weight_q, bit_width), "raw", weight_triples=level_planes) * _inv_levels(bit_width)
    if bias is not None:
        y = y + bias.detach()
    return y


def _levels_grad_weight_linear(g2, x2, x_levels, codes_fit: bool):
    """g2^T . x2 for a k-bit image x2 = q / n: codes (or their 256-digits when they left int8) as the exact bf16 operand."""
    inv = float(torch.tensor(1.0, dtype=torch.float32) / torch.tensor(float(x_levels), dtype=torch.float32))
    q = torch.round(x2.detach() * float(x_levels))
    # three exact bf16 terms: a row of g2^T is ONE output feature's gradient (the two-term form's scale is per tensor)
    gT = ops.split_bf16x3(g2.t().contiguous(), terms=3)
    if codes_fit:
        return ops.bf16_gemm(gT, ops.weight_bf16x3(q.t().contiguous(), "raw", terms=3)) * inv
    hi = torch.floor(q * (1.0 / 256.0))
    lo = q - hi * 256.0
    bad = torch.where(hi.abs().amax() >= 256.0, float("nan"), 0.0)
    return (ops.bf16_gemm(gT, ops.weight_bf16x3(hi.t().contiguous(), "raw", terms=3)) * 256.0
            + ops.bf16_gemm(gT, ops.weight_bf16x3(lo.t().contiguous(), "raw", terms=3))) * inv + bad


class DorefaW1LinearFn(QtFunction):
    """Training-mode LinearDorefa(bit_width=1): forward above; backward as autograd derives it from
    F.linear(x, _ignore_factor_op(sign(W), E), b): grad_x = g . (sign(W) E), grad_W = g^T . x passed
    through UNscaled (functions/dorefa_connect.py:66-79) with the identity STE of the quantiser.  From 2^27 MACs on both
    GEMMs run on the bf16 matrix cores (g split exactly, sign(W) / the activation's codes as the exact operand)."""

    @staticmethod
    def forward(ctx, input, weight, bias):
        ctx.has_bias = bias is not None
        ctx.save_for_backward(input, weight)
        ctx.x_levels = _act_levels(input, packed.ROWS_LAST) if input.dim() >= 2 else None
        route = []
        y = dorefa_w1_linear_forward(input, weight, bias, prequantized=False, route=route)
        ctx.codes_fit = route == ["codes"]
        return y

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        g2 = grad_output.reshape(-1, grad_output.shape[-1])
        x2 = input.reshape(-1, input.shape[-1])
        grad_input = grad_weight = grad_bias = None
        big = (g2.is_cuda and g2.dtype == torch.float32 and g2.numel() > 0 and g2.shape[0] * g2.shape[1] * x2.shape[1] >= _cfg("BWD_MFMA_MIN_MACS"))
        if ctx.needs_input_grad[0]:
            sgn = quantize_weight_f32(weight, "binary")
            E = ops.abs_mean(weight)
            if big:
                grad_input = (ops.float_linear(g2.contiguous(), sgn.t().contiguous(), "sign") * E).view(input.shape)
            else:
                grad_input = lib_mm(g2, sgn * E).view(input.shape)
        if ctx.needs_input_grad[1]:
            if big and ctx.x_levels is not None:
                grad_weight = _levels_grad_weight_linear(g2, x2, ctx.x_levels, ctx.codes_fit)
            if grad_weight is None:
                grad_weight = real_matmul(g2.t(), x2) if big else lib_mm(g2.t(), x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = g2.sum(0)
        return grad_input, grad_weight, grad_bias


class DorefaW1Conv2dFn(QtFunction):
    """Training-mode DorefaConv2d(bit_width=1); see DorefaW1LinearFn."""

    @staticmethod
    def forward(ctx, input, weight, bias, conv_args):
        ctx.has_bias, ctx.conv_args = bias is not None, conv_args
        ctx.save_for_backward(input, weight)
        # k-bit activation with valid int8 codes (the tag nnDorefaQuant leaves): its image is codes / (2^k - 1), so the
        # weight gradient can contract the integer codes on the bf16 matrix cores
        ctx.x_levels, ctx.code_flag = None, None
        if input.is_cuda and input.dtype == torch.float32 and input.dim() == 4:
            codes = packed.lookup_codes(input, packed.NHWC)
            if codes is not None and codes.K == input.shape[1]:
                ctx.x_levels = float((1 << int(codes.bit_width)) - 1)
                ctx.code_flag = codes.overflow
        route = []
        ctx.E = _dorefa_w1_scale(weight, False) if input.is_cuda else None     # mean|W|: the backward scales grad_x by it again
        y = dorefa_w1_conv_forward(input, weight, bias, conv_args, prequantized=False, route=route, scale=ctx.E)
        ctx.codes_fit = route == ["codes"]
        return y

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.conv_args
        go = _dense(grad_output)
        grad_input = grad_weight = grad_bias = None
        mfma = (_cfg("BWD_CONV_MFMA") and go.is_cuda and go.dtype == torch.float32 and groups == 1 and not isinstance(padding, str)
                and go.numel() * weight[0].numel() >= _cfg("BWD_MFMA_MIN_MACS"))
        if ctx.needs_input_grad[0]:
            E = ctx.E if ctx.E is not None else ops.abs_mean(weight)
            if mfma:     # g * (sign(W) E) = E * (g * sign(W)): the exact-split conv on the flipped +-1 weight, scaled after
                grad_input = ops.conv2d_grad_input_q(input.shape, weight, go, stride, padding, dilation, kind="binary",
                                                     out_scale_dev=E)
            if grad_input is None:
                note_library_path(go, "conv grad_input outside the matrix-core route")
                sgn = quantize_weight_f32(weight, "binary")
                grad_input = torch.nn.grad.conv2d_input(input.shape, sgn * E, go, stride=stride, padding=padding,
                                                        dilation=dilation, groups=groups)
        if ctx.needs_input_grad[1]:
            if mfma and ctx.x_levels is not None and ctx.x_levels <= 255:
                # UNscaled and un-masked, as upstream (_ignore_factor_op, identity STE): functions/dorefa_connect.py:66-79
                grad_weight = dorefa_conv_grad_weight(input, go, weight.shape[2:], stride, padding, dilation, ctx.x_levels,
                                                      ctx.codes_fit, ctx.code_flag, layout_like=weight)
            if grad_weight is None:
                note_library_path(go, "conv grad_weight outside the matrix-core route")
                grad_weight = torch.nn.grad.conv2d_weight(input, weight.shape, go, stride=stride, padding=padding,
                                                          dilation=dilation, groups=groups)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = go.sum((0, 2, 3))
        return grad_input, grad_weight, grad_bias, None


def _weight_levels(weight_q: torch.Tensor, bit_width: int) -> torch.Tensor:
    """Integer levels c = rint((2^k - 1) * w_q) of a k-bit DoReFa weight image (odd integers, |c| <= 2^k - 1 <= 127: exact
    in bf16 and in int8), as an fp32 tensor shaped like w_q."""
    return torch.round(weight_q.detach() * float((1 << int(bit_width)) - 1))


def _act_levels(t: torch.Tensor, layout):
    """n = 2^k - 1 of a k-bit DoReFa activation that carries valid int8 codes (the tag nnDorefaQuant leaves), else None."""
    if t.is_cuda and t.dtype == torch.float32:
        codes = packed.lookup_codes(t, layout)
        if codes is not None and codes.K == t.shape[1 if layout is packed.NHWC else -1]:
            n = float((1 << int(codes.bit_width)) - 1)
            return n if n <= 255 else None
    return None


def dorefa_conv_grad_weight(input, go, ksz, stride, padding, dilation, x_levels: float, codes_fit: bool, code_flag=None,
                            layout_like=None):
    """grad wrt the (quantised) weight of a DoReFa conv whose activation is a k-bit image q / n (n = ``x_levels``): the
    integer codes q are exact in bf16 while |q| <= 256, so the contraction runs on the weight-gradient routes (pixel-major,
    K-major, strided) with the split gradient.  ``codes_fit`` False — the un-clamped quantiser left the int8 range
    (functions/dorefa_connect.py:11-25 has no clamp):
      * pixel-major routes with the two-plane gradient: their activation plane is fp16, where |q| <= 2048 is still exact — one
        pass as before; ``code_flag`` (the quantiser's device flag, bit 1 = a code beyond +-2047) poisons the result with NaN
        instead of a host sync for a case that does not occur (an activation beyond 136 at 4 bits);
      * otherwise q = 256 hi + lo with both digits exact in bf16 (ops.code_digits: one launch), two passes,
        (256 GW(hi) + GW(lo)) / n (ops.digit_combine: one launch), NaN from |q| >= 2^16 through the device flag.
    ``layout_like``: the parameter — the result takes its memory format where the route can write it (no re-layout copy when
    autograd accumulates it).  No route for the shape: None, the caller uses the library."""
    def run(xt, levels):
        gw = None
        if ops.wgrad_pm_applicable(xt.shape, go.shape, ksz, stride, dilation):
            gw = ops.conv2d_grad_weight_pm(xt, go, ksz, padding, x_levels=levels, layout_like=layout_like)
        if gw is None and ops.wgrad_gemm_applicable(xt.shape, go.shape, ksz, stride, dilation):
            gw = ops.conv2d_grad_weight_gemm(xt, go, ksz, padding, x_levels=levels)
        if gw is None and ops.wgrad_strided_applicable(xt.shape, go.shape, ksz, stride, padding, dilation):
            gw = ops.conv2d_grad_weight_strided(xt, go, ksz, stride, padding, x_levels=levels, layout_like=layout_like)
        return gw

    if codes_fit:
        return run(input, x_levels)
    if code_flag is not None and ops.split_terms() == 2:
        gw = None
        if ops.wgrad_pm_applicable(input.shape, go.shape, ksz, stride, dilation):
            gw = ops.conv2d_grad_weight_pm(input, go, ksz, padding, x_levels=x_levels, terms=2, layout_like=layout_like)
        elif int(ksz[0]) > 1 and ops.wgrad_strided_applicable(input.shape, go.shape, ksz, stride, padding, dilation):
            gw = ops.conv2d_grad_weight_strided(input, go, ksz, stride, padding, x_levels=x_levels,
                                                layout_like=layout_like)     # the pixel-major kernel too
        if gw is not None:
            if code_flag.dtype != torch.int32:
                return gw + torch.where((code_flag.reshape(()) & 2) != 0, float("nan"), 0.0)
            if ops._storage_dense(gw) and not gw.is_contiguous():
                # keep the parameter's (channels-last) layout through the poison pass: it works in storage order
                st = gw.permute(0, 2, 3, 1)
                return ops.poison(st, code_flag, 2).permute(0, 3, 1, 2)
            return ops.poison(gw, code_flag, 2)
    hi, lo, dflag = ops.code_digits(input, float(x_levels), code_flag)
    g_hi = run(hi, 1.0)
    if g_hi is None:
        return None
    g_lo = run(lo, 1.0)
    # |q| >= 2^16 (an activation beyond 4369 at 4 bits): the high digit is not exact in bf16 any more — the result is poisoned on
    # the device (NaN, like the chain's int8 flag) instead of a host sync per layer for a case that does not occur
    return ops.digit_combine(g_hi, g_lo, ops._inv_f32(float(x_levels)), dflag)


class DorefaWkConv2dFn(QtFunction):
    """Training-mode DorefaConv2d(bit_width = k), 2 <= k <= 8: conv2d(x, w_q, b) for the quantised image w_q the layer's
    weight_op produced (layers/dorefa_layers.py:77-82, functions/dorefa_connect.py:99-111); the gradient w.r.t. w_q flows on
    into weight_op's own autograd graph (tanh and its normalisation), exactly as in the reference.

    w_q = c / n_w with odd integer levels c, so every contraction has one operand that is exact in bf16 / int8:
      forward   activation with int8 codes (nnDorefaQuant tag) and k <= 7: (1 / (n_a n_w)) * sum q c on the int8 matrix cores
                (dorefa_wk_conv_forward); real-valued activation, codes beyond int8, or k = 8 (|c| <= 255 does not fit int8 but
                is exact in bf16 / fp16): (1 / n_w) * the split conv with the level image (dorefa_levels_conv_forward);
      grad_x    (1 / n_w) * the exact-split conv of the gradient with the flipped level image (any square stride);
      grad_w_q  activation codes x exact-split gradient on the pixel-major / K-major / strided weight-gradient routes."""

    @staticmethod
    def forward(ctx, input, weight_q, bias, bit_width, conv_args):
        stride, padding, dilation, groups = conv_args
        ctx.has_bias, ctx.conv_args, ctx.bit_width = bias is not None, conv_args, int(bit_width)
        ctx.save_for_backward(input, weight_q)
        ctx.x_levels = _act_levels(input, packed.NHWC) if input.dim() == 4 else None
        ctx.code_flag = packed.lookup_codes(input, packed.NHWC).overflow if ctx.x_levels is not None else None
        y = None
        ok = groups == 1 and not isinstance(padding, str) and input.dim() == 4 and input.dtype == torch.float32
        if ok and ctx.x_levels is not None and bit_width <= LEVEL_INT8_BITS:
            wc = ops.pack_conv_weight_dorefa_codes(weight_q.detach(), bit_width)
            y = dorefa_wk_conv_forward(input, weight_q, bias, conv_args, bit_width, wc)
        ctx.codes_fit = y is not None            # the int8 route ran: every |code| <= 127
        if y is None and ok and _cfg("FLOAT_PATH") == "bf16x3":
            y = dorefa_levels_conv_forward(input, weight_q, bias, conv_args, bit_width)
        if y is None:
            note_library_path(input, "k-bit DoReFa conv outside the level routes")
            y = F.conv2d(input, weight_q, bias, stride, padding, dilation, groups)
        return y

    @staticmethod
    def backward(ctx, grad_output):
        input, weight_q = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.conv_args
        go = _dense(grad_output)
        k = ctx.bit_width
        grad_input = grad_weight = grad_bias = None
        mfma = (_cfg("BWD_CONV_MFMA") and go.is_cuda and go.dtype == torch.float32 and groups == 1 and not isinstance(padding, str)
                and go.numel() * weight_q[0].numel() >= _cfg("BWD_MFMA_MIN_MACS"))
        if ctx.needs_input_grad[0]:
            if mfma:
                grad_input = ops.conv2d_grad_input_q(input.shape, _weight_levels(weight_q, k), go, stride, padding, dilation,
                                                     kind="raw", out_scale=_inv_levels(k))
            if grad_input is None:
                note_library_path(go, "conv grad_input outside the matrix-core route")
                grad_input = torch.nn.grad.conv2d_input(input.shape, weight_q, go, stride=stride, padding=padding,
                                                        dilation=dilation, groups=groups)
        if ctx.needs_input_grad[1]:
            ksz = weight_q.shape[2:]
            if mfma and ctx.x_levels is not None:
                grad_weight = dorefa_conv_grad_weight(input, go, ksz, stride, padding, dilation, ctx.x_levels, ctx.codes_fit,
                                                      ctx.code_flag)
            if grad_weight is None:
                note_library_path(go, "conv grad_weight outside the matrix-core route")
                grad_weight = torch.nn.grad.conv2d_weight(input, weight_q.shape, go, stride=stride, padding=padding,
                                                          dilation=dilation, groups=groups)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = go.sum((0, 2, 3))
        return grad_input, grad_weight, grad_bias, None, None


class DorefaWkLinearFn(QtFunction):
    """Training-mode LinearDorefa(bit_width = k), 2 <= k <= 8; see DorefaWkConv2dFn (layers/dorefa_layers.py:41-45)."""

    @staticmethod
    def forward(ctx, input, weight_q, bias, bit_width):
        ctx.has_bias, ctx.bit_width = bias is not None, int(bit_width)
        ctx.save_for_backward(input, weight_q)
        ctx.x_levels = _act_levels(input, packed.ROWS_LAST)
        y = None
        if input.dtype == torch.float32 and ctx.x_levels is not None and bit_width <= LEVEL_INT8_BITS:
            wc = ops.dorefa_weight_codes(weight_q.detach(), bit_width)
            y = dorefa_wk_linear_forward(input, weight_q, bias, bit_width, wc)
        ctx.codes_fit = y is not None
        if y is None and _cfg("FLOAT_PATH") == "bf16x3":
            y = dorefa_levels_linear_forward(input, weight_q, bias, bit_width)
        if y is None:
            note_library_path(input, "k-bit DoReFa linear outside the level routes")
            y = F.linear(input, weight_q, bias)
        return y

    @staticmethod
    def backward(ctx, grad_output):
        input, weight_q = ctx.saved_tensors
        k = ctx.bit_width
        g2 = grad_output.reshape(-1, grad_output.shape[-1])
        x2 = input.reshape(-1, input.shape[-1])
        grad_input = grad_weight = grad_bias = None
        big = (g2.is_cuda and g2.dtype == torch.float32 and g2.numel() > 0
               and g2.shape[0] * g2.shape[1] * x2.shape[1] >= _cfg("BWD_MFMA_MIN_MACS"))
        if ctx.needs_input_grad[0]:
            if big:        # real gradient x integer levels
                lv = _weight_levels(weight_q, k)
                grad_input = (ops.float_linear(g2.contiguous(), lv.t().contiguous(), "raw") * _inv_levels(k)).view(input.shape)
            else:
                grad_input = lib_mm(g2, weight_q).view(input.shape)
        if ctx.needs_input_grad[1]:
            if big and ctx.x_levels is not None:   # g^T . x with x = codes / n_a
                grad_weight = _levels_grad_weight_linear(g2, x2, ctx.x_levels, ctx.codes_fit)
            if grad_weight is None:
                grad_weight = real_matmul(g2.t(), x2) if big else lib_mm(g2.t(), x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = g2.sum(0)
        return grad_input, grad_weight, grad_bias, None


class RealLinearFn(QtFunction):
    """F.linear(x, W, b) for two REAL fp32 device operands on the matrix cores (six-term bf16 planes, fp32-GEMM accuracy), forward
    and both gradients: LinearDorefa(bit_width = 32), whose weight quantiser is the identity (functions/dorefa_connect.py:100-101,
    layers/dorefa_layers.py:41-45) — the layer is an un-quantised nn.Linear."""

    @staticmethod
    def forward(ctx, input, weight, bias):
        ctx.has_bias = bias is not None
        ctx.save_for_backward(input, weight)
        x2 = input.reshape(-1, input.shape[-1])
        y = ops.real_linear(x2.detach().contiguous(), weight.detach(), bias.detach() if bias is not None else None)
        return y.view(*input.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        g2 = grad_output.reshape(-1, grad_output.shape[-1])
        x2 = input.reshape(-1, input.shape[-1])
        grad_input = grad_weight = grad_bias = None
        if ctx.needs_input_grad[0]:
            grad_input = real_matmul(g2, weight).view(input.shape)
        if ctx.needs_input_grad[1]:
            grad_weight = real_matmul(g2.t(), x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = g2.sum(0)
        return grad_input, grad_weight, grad_bias


class RealConv2dFn(QtFunction):
    """F.conv2d(x, W, b) for two REAL fp32 device operands (groups == 1, zero padding): DorefaConv2d(bit_width = 32).  Forward and
    — for stride 1, dilation 1 — grad_x (the conv of the gradient with the flipped, transposed weight) on the six-term implicit
    GEMM; grad_W on the pixel-major kernel (stride 1, 3 x 3 / 5 x 5: the image's terms as channel groups) or one six-term GEMM per
    filter tap (real_conv_grad_weight_taps)."""

    @staticmethod
    def forward(ctx, input, weight, bias, conv_args):
        stride, padding, dilation, groups = conv_args
        ctx.has_bias, ctx.conv_args = bias is not None, conv_args
        ctx.save_for_backward(input, weight)
        return real_weight_conv2d(input, weight, bias, stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, grad_output):
        input, weight = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.conv_args
        go = _dense(grad_output)
        grad_input = grad_weight = grad_bias = None
        kh, kw = int(weight.shape[2]), int(weight.shape[3])
        (sh, sw), (ph, pw), (dh, dw) = ops._pairs(stride), ops._pairs(padding), ops._pairs(dilation)
        if ctx.needs_input_grad[0]:
            if (go.is_cuda and go.dtype == torch.float32 and go.numel() > 0 and sh == sw and (dh, dw) == (1, 1)
                    and ph <= kh - 1 and pw <= kw - 1):
                wt = weight.detach().flip(2, 3).transpose(0, 1).contiguous()
                gd = go if sh == 1 else ops.zero_dilated_gradient(go, input.shape, (kh, kw), sh, (ph, pw))   # stride s: conv_transpose
                y2 = ops.real_conv2d(gd, wt, None, 1, (kh - 1 - ph, kw - 1 - pw), 1) if gd is not None else None
                if y2 is not None:
                    N_, C, H, W = input.shape
                    grad_input = y2.view(N_, H, W, C).permute(0, 3, 1, 2)
                    if input.is_contiguous() and not input.is_contiguous(memory_format=torch.channels_last):
                        grad_input = grad_input.contiguous()
            if grad_input is None:
                grad_input = lib_conv2d_input(input.shape, weight, go, stride, padding, dilation, groups)
        if ctx.needs_input_grad[1]:
            grad_weight = conv_grad_weight(input, weight.shape, go, stride, padding, dilation, groups, x_is_pm1=False,
                                           real_any_channels=True)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = go.sum((0, 2, 3))
        return grad_input, grad_weight, grad_bias, None


def grouped_quant_conv(layer, input, kind: str, quant_op):
    """BinConv2d / TerConv2d with ``groups`` > 1 (layers/binary_layers.py:59-60,103-106 hand ``groups`` to F.conv2d): group g is
    an independent conv of its channel slice with its rows of the weight, and the element-wise quantisers (safeSign, the fixed
    ternary thresholds) commute with slicing — so every group runs on the groups == 1 routes of this backend, forward and
    backward, and the results are concatenated.  Returns None when the case is not this path's (eval mode with autograd, a
    weight off the quantiser's grid): the caller continues with its general code."""
    G = int(layer.groups)
    cin, cout = int(input.shape[1]) // G, int(layer.weight.shape[0]) // G
    args = (layer.stride, layer.padding, layer.dilation, 1)
    if not layer.training:
        if torch.is_grad_enabled() and (input.requires_grad or layer.weight.requires_grad):
            return None
        if not layer._eval_on_grid():
            return None
    wq_full = None
    if layer.training and not layer.deterministic:
        wq_full = quant_op.apply(layer.weight.detach())          # ONE stochastic draw for the whole weight, as upstream
    cl = input.is_contiguous(memory_format=torch.channels_last) and not input.is_contiguous()
    # a +-1 tag of the whole activation (BinaryConnect's sign planes) holds for every channel slice
    known_pm1 = True if (layer.binary_input or packed.lookup(input, packed.NHWC) is not None) else layer.binary_input
    flag = None
    if (known_pm1 is None and _cfg("DETECT_BINARY_INPUT") and input.is_cuda and input.dtype == torch.float32 and input.dim() == 4
            and input.numel() > 0):
        # un-tagged activation, no hint: ONE detection on the whole activation, remembered under the layer's own weight (the
        # per-group weight views are fresh tensors on every call: verdicts keyed on them die with them — a host sync per group
        # per call, and nothing for DETECT_MODE = "remember" to remember)
        ok, flag = detect_pm1(input, layer.weight)
        known_pm1 = bool(ok)
    ys = []
    for g in range(G):
        xg = input[:, g * cin:(g + 1) * cin]
        xg = xg.contiguous(memory_format=torch.channels_last) if cl else xg.contiguous()
        wg = layer.weight[g * cout:(g + 1) * cout]
        bg = layer.bias[g * cout:(g + 1) * cout] if layer.bias is not None else None
        if layer.training:
            wq = wq_full[g * cout:(g + 1) * cout] if wq_full is not None else None
            ys.append(QuantConv2dFn.apply(xg, wg, bg, kind, wq, known_pm1, args))
        else:
            ys.append(quant_conv2d_forward(xg, wg, bg, *args, kind, weight_q=wg, binary_input=known_pm1, padding_mode="zeros"))
    out = torch.cat(ys, 1)
    if flag is not None:          # a remembered "+-1" verdict: the device flag of THIS activation turns a wrong assumption into NaN
        out = out + torch.where(flag.reshape(()) != 0, float("nan"), 0.0).to(out.dtype)
    return out


#: backward GEMMs with one +-1/0 operand (grad_x = g . Q(W); grad_W = g^T . x for +-1 activations) run on the bf16
#: matrix cores from this many multiply-accumulates on: the real operand is split exactly into bf16 triples, the
#: +-1/0 operand is exact in bf16, accumulation is fp32 — fp32-GEMM accuracy at ~2.5x the fp32 library's speed
#: (4096^3: 0.51 vs 1.03 ms per GEMM incl. transposes and splits, max error 1.6e-6 vs 3.0e-6 of fp64; tools/bench_backward.py).  Below it the dense library is used (launch-bound either way).
BWD_MFMA_MIN_MACS = 0          # (round 4: every size on the own routes; 1 << 27 was the measured break-even against hipBLASLt)


#: backward convs with a +-1 / 0 operand on the bf16 matrix cores (ops.conv2d_grad_input_q / conv2d_grad_weight_pm1)
BWD_CONV_MFMA = True


#: grad_x of LinearBin / LinearTer packs Q(W)^T in one kernel from W (A/B switch for tools/)
LINEAR_GRAD_X_ONE_PACK = True


def pm1_matmul(a: torch.Tensor, b_pm1: torch.Tensor, terms=None) -> torch.Tensor:
    """a [M, J] (real) @ b_pm1 [J, K] (entries in {-1, 0, +1}) -> [M, K] fp32.  ``terms``: split of a (ops.float_linear)."""
    M, J = a.shape
    K = b_pm1.shape[1]
    if a.is_cuda and a.dtype == torch.float32 and M * J * K >= _cfg("BWD_MFMA_MIN_MACS") and a.numel() > 0 and b_pm1.numel() > 0:
        return ops.float_linear(a.contiguous(), b_pm1.t().contiguous(), "sign", terms=terms)
    return lib_mm(a, b_pm1, "backward GEMM below BWD_MFMA_MIN_MACS / non-fp32")


class QuantLinearFn(QtFunction):
    """Autograd node of LinearBin / LinearTer in training mode.

    forward : F.linear(x, Q(W), b)                         layers/binary_layers.py:44
    backward: grad_x = g . Q(W) ; grad_W = (g^T . x) * 1[|W| <= 1.001] ; grad_b = sum g
              (what autograd derives from F.linear + the STE backward of the quantiser,
              functions/binary_connect.py:31-38 / terner_connect.py:29-34)
    """

    @staticmethod
    def forward(ctx, input, weight, bias, kind, weight_q, binary_input):
        ctx.kind = kind
        ctx.has_bias = bias is not None
        # a stochastic draw must be kept; the deterministic image is recomputed in backward
        ctx.save_for_backward(input, weight, weight_q)
        # +-1 activation known without a device check (tag of a quantiser, or the layer's binary_input hint)?
        ctx.x_is_pm1 = bool(binary_input) or (input.is_cuda and input.dtype == torch.float32
                                              and packed.lookup(input, packed.ROWS_LAST) is not None)
        clear_last_detection()
        out = quant_linear_forward(input, weight, bias, kind, weight_q=weight_q, binary_input=binary_input)
        if not ctx.x_is_pm1 and binary_input is None and input.is_cuda:
            # un-tagged activation THIS forward detected as +-1 (a reshaped / flattened sign image): the backward's g^T . x then
            # runs on the matrix cores as well instead of the dense library
            ctx.x_is_pm1 = last_detection_said_pm1(input)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        from .common import ste_mask
        input, weight, weight_q = ctx.saved_tensors
        g2 = grad_output.reshape(-1, grad_output.shape[-1])
        grad_input = grad_weight = grad_bias = None
        if ctx.needs_input_grad[0]:
            Nf, Kf = int(weight.shape[0]), int(weight[0].numel())
            if (_cfg("LINEAR_GRAD_X_ONE_PACK") and weight_q is None and ctx.kind in ("binary", "ternary") and g2.is_cuda and g2.dtype == torch.float32
                    and weight.dtype == torch.float32 and ops.split_terms() == 2 and Nf % 4 == 0 and g2.numel() > 0 and weight.numel() > 0):
                # g . Q(W): the operand Q(W)^T comes from ONE kernel that reads W where it lies (quantiser + transpose + fp16 pairs:
                # qt_f16x2_pack_conv_weight_f32 on the [N, K, 1, 1] view) instead of quantise, transpose-copy and pack (three passes,
                # 280 us for AlexNet's 9216 x 4096 layer)
                wt = ops.pack_conv_weight_bf16x3(weight.detach().reshape(Nf, Kf, 1, 1), ctx.kind, terms=2, transpose_flip=True)
                shape_t = torch.empty((Kf, Nf), dtype=torch.float32, device="meta")
                grad_input = ops.float_linear(g2.contiguous(), shape_t, ctx.kind, weight_triples=wt, terms=2).view(input.shape)
            else:
                wq = weight_q if weight_q is not None else quantize_weight_f32(weight, ctx.kind)
                grad_input = pm1_matmul(g2, wq).view(input.shape)
        if ctx.needs_input_grad[1]:
            x2 = input.reshape(-1, input.shape[-1])
            # (rows of g2^T are output features: the exact three-term split, not the per-tensor-scaled two-term one)
            if ctx.x_is_pm1:
                gw = pm1_matmul(g2.t(), x2, terms=3)
            else:
                gw = real_matmul(g2.t(), x2)          # two real operands (a real-valued first layer): six-term planes
            grad_weight = ste_mask(gw, weight)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            grad_bias = g2.sum(0)
        return grad_input, grad_weight, grad_bias, None, None, None
