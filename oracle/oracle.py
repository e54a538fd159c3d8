"""numpy front-end of the CPU oracle (oracle/qt_oracle.c) — TEST INFRASTRUCTURE ONLY.

Each function restates one expression of the reference (Enderdead/Pytorch_Quantize_impls,
paths relative to its root) on numpy float32 arrays.  Elementwise / integer work is done by the
C restatement through ctypes; the thin layer-level compositions (LinearBin.forward = linear(x,
safeSign(W), b), ...) are written here.  Parity with the reference itself is pinned by
tests/golden (see tests/golden/make_golden.py) and tests/test_oracle_golden.py.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libqt_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_i64 = ctypes.c_int64


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, a second or two)."""
    src = os.path.join(_HERE, "qt_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _u(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(_u32p)


def _optf(a):
    if a is None:
        return None, None
    return _f(a)


# ---- elementwise ---------------------------------------------------------------------------

def safe_sign(x):
    """functions/common.py:4-7."""
    x, xp = _f(x)
    y = np.empty_like(x)
    lib().qo_safe_sign(xp, y.ctypes.data_as(_f32p), _i64(x.size))
    return y


def ste_mask(grad, x, thr=1.001):
    """functions/binary_connect.py:31-38 (same in terner_connect.py:29-34)."""
    g, gp = _f(grad)
    x, xp = _f(x)
    out = np.empty_like(g)
    lib().qo_ste_mask(gp, xp, out.ctypes.data_as(_f32p), _i64(g.size), ctypes.c_float(thr))
    return out


def ternarize(x):
    """functions/terner_connect.py:24-27."""
    x, xp = _f(x)
    y = np.empty_like(x)
    lib().qo_ternarize(xp, y.ctypes.data_as(_f32p), _i64(x.size))
    return y


def binarize_stochastic(x, z):
    """functions/binary_connect.py:57-61 with explicit uniforms."""
    x, xp = _f(x)
    z, zp = _f(z)
    y = np.empty_like(x)
    lib().qo_binarize_stochastic(xp, zp, y.ctypes.data_as(_f32p), _i64(x.size))
    return y


def ternarize_stochastic(x, z):
    """functions/terner_connect.py:52-56 with explicit uniforms."""
    x, xp = _f(x)
    z, zp = _f(z)
    y = np.empty_like(x)
    lib().qo_ternarize_stochastic(xp, zp, y.ctypes.data_as(_f32p), _i64(x.size))
    return y


def dorefa_quantize(x, k):
    """functions/dorefa_connect.py:11-25."""
    x, xp = _f(x)
    y = np.empty_like(x)
    lib().qo_dorefa_quantize(xp, y.ctypes.data_as(_f32p), _i64(x.size), ctypes.c_int(int(k)))
    return y


def affine_relu_dorefa_codes(x, alpha, beta, k, relu=True, res=None, res_affine=None):
    """The eval-mode chain  BatchNorm (folded: t = fl(fl(x*alpha) + beta), channel = dim 1) [+ residual, itself
    optionally through a folded BatchNorm] [-> ReLU] -> nnDorefaQuant(k)  of a DoReFa ResNet block
    (models/samples/ResNet_Dorefa.py:26,35 ; functions/dorefa_connect.py:11-25) in fp32 steps.
    Returns (integer codes rint(n*t) as float32, fp32 image = dorefa_quantize(t, k))."""
    x = np.asarray(x, dtype=np.float32)
    if relu == "pre":                       # ReLU in front of the BatchNorm (models/FullNet/DorefaMNIST.py:46-48)
        x = np.where(x < 0, np.float32(0), x).astype(np.float32)
    shp = [1] * x.ndim
    shp[1] = -1
    a, b = (np.asarray(v, dtype=np.float32).reshape(shp) for v in (alpha, beta))
    t = (x * a).astype(np.float32) + b
    if res is not None:
        u = np.asarray(res, dtype=np.float32)
        if res_affine is not None:
            ra, rb = (np.asarray(v, dtype=np.float32).reshape(shp) for v in res_affine)
            u = (u * ra).astype(np.float32) + rb
        t = (t + u).astype(np.float32)
    if relu and relu != "pre":
        t = np.where(t < 0, np.float32(0), t).astype(np.float32)
    n = np.float32((1 << int(k)) - 1)
    return np.rint((n * t).astype(np.float32)).astype(np.float32), dorefa_quantize(t, k)


def lin_quantize(x, fsr, bits, mode=1):
    """functions/log_lin_connect.py:61-79 (mode 0 unsigned, 1 with_sign, 2 quantised-gradient backward)."""
    x, xp = _f(x)
    y = np.empty_like(x)
    lib().qo_lin_quantize(xp, y.ctypes.data_as(_f32p), _i64(x.size), ctypes.c_int(int(fsr)), ctypes.c_int(int(bits)),
                          ctypes.c_int(int(mode)))
    return y


def log_quantize(x, fsr, bits, with_sign=True):
    """functions/log_lin_connect.py:31-40."""
    x, xp = _f(x)
    y = np.empty_like(x)
    lib().qo_log_quantize(xp, y.ctypes.data_as(_f32p), _i64(x.size), ctypes.c_int(int(fsr)), ctypes.c_int(int(bits)),
                          ctypes.c_int(1 if with_sign else 0))
    return y


def ap2(x):
    """functions/binary_connect.py:157-169."""
    x, xp = _f(x)
    y = np.empty_like(x)
    lib().qo_ap2(xp, y.ctypes.data_as(_f32p), _i64(x.size))
    return y


# ---- packed format ---------------------------------------------------------------------------

def packed_ld(K: int) -> int:
    kw = (int(K) + 31) // 32
    return max(4, (kw + 3) // 4 * 4)


def sign_pack(x2d):
    x, xp = _f(x2d)
    rows, K = x.shape
    ld = packed_ld(K)
    plane = np.zeros((rows, ld), dtype=np.uint32)
    lib().qo_sign_pack(xp, _i64(K), plane.ctypes.data_as(_u32p), _i64(ld), _i64(rows), _i64(K))
    return plane


def ternary_pack(x2d):
    x, xp = _f(x2d)
    rows, K = x.shape
    ld = packed_ld(K)
    mask = np.zeros((rows, ld), dtype=np.uint32)
    sign = np.zeros((rows, ld), dtype=np.uint32)
    lib().qo_ternary_pack(xp, _i64(K), mask.ctypes.data_as(_u32p), sign.ctypes.data_as(_u32p),
                          _i64(ld), _i64(rows), _i64(K))
    return mask, sign


def xnor_gemm(xs, ws, K, bias=None):
    xs, xsp = _u(xs)
    ws, wsp = _u(ws)
    M, N = xs.shape[0], ws.shape[0]
    b, bp = _optf(bias)
    y = np.empty((M, N), dtype=np.float32)
    lib().qo_xnor_gemm(xsp, _i64(xs.shape[1]), wsp, _i64(ws.shape[1]), bp,
                       y.ctypes.data_as(_f32p), _i64(N), _i64(M), _i64(N), _i64(K))
    return y


def tern_gemm(xs, wmask, wsign, K, bias=None):
    xs, xsp = _u(xs)
    wm, wmp = _u(wmask)
    wsn, wsnp = _u(wsign)
    M, N = xs.shape[0], wm.shape[0]
    b, bp = _optf(bias)
    y = np.empty((M, N), dtype=np.float32)
    lib().qo_tern_gemm(xsp, _i64(xs.shape[1]), wmp, wsnp, _i64(wm.shape[1]), bp,
                       y.ctypes.data_as(_f32p), _i64(N), _i64(M), _i64(N), _i64(K))
    return y


def packed_ld_nib(K: int) -> int:
    kw = (int(K) + 7) // 8
    return max(32, (kw + 31) // 32 * 32)


def pack_nib(x2d, ternary=False):
    x, xp = _f(x2d)
    rows, K = x.shape
    ld = packed_ld_nib(K)
    out = np.zeros((rows, ld), dtype=np.uint32)
    lib().qo_pack_nib(xp, _i64(K), out.ctypes.data_as(_u32p), _i64(ld), _i64(rows), _i64(K),
                      ctypes.c_int(1 if ternary else 0))
    return out


def bits_to_nib(sign, mask, K):
    sg, sgp = _u(sign)
    rows, ldb = sg.shape
    mp = None
    if mask is not None:
        mk, mp = _u(mask)
    ld = packed_ld_nib(K)
    out = np.zeros((rows, ld), dtype=np.uint32)
    lib().qo_bits_to_nib(sgp, mp, _i64(ldb), out.ctypes.data_as(_u32p), _i64(ld), _i64(rows), _i64(K))
    return out


def nib_gemm(xn, wn, K, bias=None):
    xn, xnp = _u(xn)
    wn, wnp = _u(wn)
    M, N = xn.shape[0], wn.shape[0]
    b, bp = _optf(bias)
    y = np.empty((M, N), dtype=np.float32)
    lib().qo_nib_gemm(xnp, _i64(xn.shape[1]), wnp, _i64(wn.shape[1]), bp, y.ctypes.data_as(_f32p),
                      _i64(N), _i64(M), _i64(N), _i64(K))
    return y


# ---- contractions (third-party in the reference: torch.nn.functional.linear / conv2d) ----------

def linear(x, w, bias=None):
    x, xp = _f(x)
    w, wp = _f(w)
    lead = x.shape[:-1]
    K = x.shape[-1]
    M = int(np.prod(lead)) if lead else 1
    N = w.shape[0]
    b, bp = _optf(bias)
    y = np.empty((M, N), dtype=np.float32)
    lib().qo_linear_f64acc(xp, wp, bp, y.ctypes.data_as(_f32p), _i64(M), _i64(N), _i64(K))
    return y.reshape(*lead, N)


def _pair(v):
    return (int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[1]))


def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    x, xp = _f(x)
    w, wp = _f(w)
    Nb, Cin, H, W = x.shape
    Cout, _, kh, kw = w.shape
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    b, bp = _optf(bias)
    y = np.empty((Nb, Cout, Ho, Wo), dtype=np.float32)
    lib().qo_conv2d_f64acc(xp, wp, bp, y.ctypes.data_as(_f32p), _i64(Nb), _i64(Cin), _i64(H),
                           _i64(W), _i64(Cout), _i64(kh), _i64(kw), _i64(sh), _i64(sw), _i64(ph),
                           _i64(pw), _i64(dh), _i64(dw), _i64(groups))
    return y


# ---- layer-level compositions ----------------------------------------------------------------

def linear_bin_forward(x, weight, bias=None, training=True):
    """LinearBin.forward (layers/binary_layers.py:42-46)."""
    return linear(x, safe_sign(weight) if training else weight, bias)


def linear_ter_forward(x, weight, bias=None, training=True):
    """LinearTer.forward (layers/terner_layers.py:47-51)."""
    return linear(x, ternarize(weight) if training else weight, bias)


def bin_conv2d_forward(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, training=True):
    """BinConv2d.forward (layers/binary_layers.py:103-106)."""
    return conv2d(x, safe_sign(weight) if training else weight, bias, stride, padding, dilation, groups)


def ter_conv2d_forward(x, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, training=True):
    """TerConv2d.forward (layers/terner_layers.py:89-92)."""
    return conv2d(x, ternarize(weight) if training else weight, bias, stride, padding, dilation, groups)


def dorefa_weight(weight, k):
    """nnQuantWeight(k).forward (functions/dorefa_connect.py:99-111).  tanh / mean / max are the
    host libm's: last-ulp differences against torch's vectorised tanh can move a value across a
    rounding boundary, so tests compare the integer CODES with a +-1 allowance on such ties."""
    w = np.asarray(weight, dtype=np.float32)
    if k == 1:
        E = np.mean(np.abs(w), dtype=np.float32)
        return safe_sign(w) * np.float32(E)
    if k == 32:
        return w.copy()
    if np.max(np.abs(w)) == 0.0:
        return np.zeros_like(w)
    t = np.tanh(w).astype(np.float32)
    t = t / (np.float32(2) * np.max(np.abs(t))) + np.float32(0.5)
    return np.float32(2) * dorefa_quantize(t, k) - np.float32(1)


def xnor_dense_weight(weight):
    """XNORDense forward weight: sign(W)*mean(|W|, DIM=0, keepdim) (functions/xnor_connect.py:112-113)."""
    w = np.asarray(weight, dtype=np.float32)
    # (float64 accumulation, rounded once: numpy's float32 reduction over an outer axis is a plain running sum — 4e-6 off at 4096
    # rows — where torch's cascade sum stays within an ulp of this)
    mean = np.mean(np.abs(w), axis=0, keepdims=True, dtype=np.float64).astype(np.float32)
    return np.sign(w).astype(np.float32) * mean


def xnor_conv_weight(weight, dim=(0, 1)):
    """XNORConv2d forward weight: sign(W)*mean(|W|, dim, keepdim) (functions/xnor_connect.py:140-141)."""
    w = np.asarray(weight, dtype=np.float32)
    mean = np.mean(np.abs(w), axis=tuple(dim), keepdims=True, dtype=np.float64).astype(np.float32)     # see xnor_dense_weight
    return np.sign(w).astype(np.float32) * mean


# ---- XNOR-Net layers: forward and the reference's hand-written backward (functions/xnor_connect.py:93-169) ---------------------
# Restated in float64 numpy (the contractions as per-tap BLAS products); used by the tests only, at sizes that finish in seconds.

def conv2d_input_f64(in_shape, wq, g, stride=1, padding=0, dilation=1):
    """torch.nn.grad.conv2d_input (groups = 1) in float64: gx[n, c, ho s - p + i d, wo s - p + j d] += g[n, o, ho, wo] wq[o, c, i, j]."""
    Nb, Cin, H, W = (int(v) for v in in_shape)
    wq = np.asarray(wq, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    Cout, _, kh, kw = wq.shape
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    _, _, Ho, Wo = g.shape
    gp = np.zeros((Nb, Cin, H + 2 * ph, W + 2 * pw), dtype=np.float64)
    g2 = g.transpose(0, 2, 3, 1).reshape(-1, Cout)                          # [n ho wo, o]
    for i in range(kh):
        for j in range(kw):
            t = (g2 @ wq[:, :, i, j]).reshape(Nb, Ho, Wo, Cin).transpose(0, 3, 1, 2)
            gp[:, :, i * dh:i * dh + sh * (Ho - 1) + 1:sh, j * dw:j * dw + sw * (Wo - 1) + 1:sw] += t
    return gp[:, :, ph:ph + H, pw:pw + W]


def conv2d_weight_f64(x, w_shape, g, stride=1, padding=0, dilation=1):
    """torch.nn.grad.conv2d_weight (groups = 1) in float64."""
    x = np.asarray(x, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    Cout, Cin, kh, kw = (int(v) for v in w_shape)
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    Nb, _, Ho, Wo = g.shape
    xp = np.pad(x, ((0, 0), (0, 0), (ph, ph), (pw, pw)))
    g2 = g.transpose(1, 0, 2, 3).reshape(Cout, -1)                          # [o, n ho wo]
    gw = np.empty((Cout, Cin, kh, kw), dtype=np.float64)
    for i in range(kh):
        for j in range(kw):
            xs = xp[:, :, i * dh:i * dh + sh * (Ho - 1) + 1:sh, j * dw:j * dw + sw * (Wo - 1) + 1:sw]
            gw[:, :, i, j] = g2 @ xs.transpose(0, 2, 3, 1).reshape(-1, Cin)
    return gw


def xnor_input_quant(x, dtype=np.float32):
    """The input quantiser of XNORConv2d(quant_input=True): sign(x) * mean(|x|, 1, keepdim) (functions/xnor_connect.py:142-143;
    torch.sign: an exact zero stays zero; the mean runs over the channel dimension of an NCHW tensor -> one scale per pixel)."""
    x = np.asarray(x, dtype=dtype)
    mean = np.mean(np.abs(x), axis=1, keepdims=True, dtype=np.float64).astype(dtype)        # see xnor_dense_weight
    return np.sign(x).astype(dtype) * mean


def xnor_conv2d_forward(x, weight, bias=None, stride=1, padding=1, dilation=1, dim=(0, 1), quant_input=False):
    """XNORConv2d forward (functions/xnor_connect.py:139-146; quant_input False as the layer hard-codes, layers/xnor_layers.py:49;
    True: the function's own switch, :142-143)."""
    if quant_input:
        x = xnor_input_quant(x)
    return conv2d(x, xnor_conv_weight(weight, dim), bias, stride, padding, dilation)


def xnor_conv2d_backward(g, x, weight, stride=1, padding=1, dilation=1, dim=(0, 1), quant_input=False):
    """XNORConv2d backward (functions/xnor_connect.py:149-168) in float64: (grad_input, grad_weight, grad_bias).  The weight
    gradient mixes ``dim`` (the saved mean) with the module-global DIM = 0 (the second term's reduction), as upstream.
    quant_input: backward sees the QUANTISED input (it is what forward saved, :144), and grad_input is the plain conv2d_input
    (straight through the input quantiser: upstream's "TODO backprob input quant", :136)."""
    if quant_input:
        x = xnor_input_quant(x, np.float64)
    w = np.asarray(weight, dtype=np.float64)
    sgn = np.sign(w)
    mean = np.mean(np.abs(w), axis=tuple(dim), keepdims=True)
    gx = conv2d_input_f64(np.shape(x), sgn * mean, g, stride, padding, dilation)
    gt = conv2d_weight_f64(x, w.shape, g, stride, padding, dilation)
    gw = mean * gt + sgn * np.mean(gt * sgn, axis=0, keepdims=True)
    gb = np.asarray(g, dtype=np.float64).sum((0, 2, 3))
    return gx, gw, gb


def xnor_dense_forward(x, weight, bias=None):
    """XNORDense forward (functions/xnor_connect.py:110-116)."""
    return linear(x, xnor_dense_weight(weight), bias)


def xnor_dense_backward(g, x, weight):
    """XNORDense backward (functions/xnor_connect.py:118-131) in float64: (grad_input, grad_weight, grad_bias)."""
    w = np.asarray(weight, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    sgn = np.sign(w)
    mean = np.mean(np.abs(w), axis=0, keepdims=True)
    gx = g @ (sgn * mean)
    gt = g.T @ np.asarray(x, dtype=np.float64)
    gw = mean * gt + sgn * np.mean(gt * sgn, axis=0, keepdims=True)
    return gx, gw, g.sum(0)


def shift_batch(x, mean, var, weight, bias, eps):
    """ShiftBatch.forward (functions/binary_connect.py:177-183): ((x - mean) * AP2(1 / sqrt(var + eps))) * AP2(weight)
    + bias in fp32 steps (the statistics broadcast over the leading dimension)."""
    x = np.asarray(x, dtype=np.float32)
    sv = np.sqrt((np.asarray(var, np.float32) + np.float32(eps)).astype(np.float32)).astype(np.float32)
    a = ap2((np.float32(1) / sv).astype(np.float32))
    norm = ((x - np.asarray(mean, np.float32)).astype(np.float32) * a).astype(np.float32)
    return ((norm * ap2(np.asarray(weight, np.float32))).astype(np.float32) + np.asarray(bias, np.float32)).astype(np.float32), norm


def xnor_act(x, dim):
    """_quantOpXnor forward (functions/xnor_connect.py:17-28): sign(x) * mean(x, dim) with np.sign == torch.sign
    (0 -> 0) and the SIGNED mean the reference computes; the mean is accumulated in double (the float tail
    tolerance is stated against it).  dim = 1: per row, dim = 0: per column, dim = -1: whole tensor."""
    x = np.asarray(x, dtype=np.float32)
    xd = x.astype(np.float64)
    if dim < 0:
        mean = np.float32(xd.mean())
        return np.sign(x) * mean, np.asarray([mean], dtype=np.float32)
    mean = xd.mean(axis=dim).astype(np.float32)
    shape = (1, -1) if dim == 0 else (-1, 1)
    return (np.sign(x) * mean.reshape(shape)).astype(np.float32), mean


def xnor_act_backward(g, x, dim):
    """_quantOpXnor backward (functions/xnor_connect.py:30-37):
    sign(x) * mean(g * sign(x), dim, keepdim) + g * mean(x, dim)."""
    x = np.asarray(x, dtype=np.float32)
    g = np.asarray(g, dtype=np.float32)
    sgn = np.sign(x).astype(np.float64)
    gd, xd = g.astype(np.float64), x.astype(np.float64)
    if dim < 0:
        return (sgn * (gd * sgn).mean() + gd * np.float32(xd.mean())).astype(np.float32)
    shape = (1, -1) if dim == 0 else (-1, 1)
    mean = xd.mean(axis=dim).astype(np.float32).reshape(shape)
    return (sgn * (gd * sgn).mean(axis=dim, keepdims=True) + gd * mean).astype(np.float32)


def functional_ternary_weight(weight):
    """Deterministic branch of the functional TernaryDense / TernaryConv2d (functions/terner_connect.py:85-90,
    :123-128): (sign(w) + sign(w - 0.5 sign(w))) / 2 with torch.sign, so w = +-0.5 gives +-0.5 and w = 0 gives 0
    (unlike TernaryConnectDeterministic, which uses safeSign)."""
    w = np.asarray(weight, dtype=np.float32)
    s = np.sign(w).astype(np.float32)
    return ((s + np.sign(w - np.float32(0.5) * s).astype(np.float32)) / np.float32(2)).astype(np.float32)


def functional_quant_weight(weight, k, conv=False):
    """QuantDense / QuantConv2d forward weight (functions/dorefa_connect.py:124-130, :170-176): k = 1: safeSign(W) *
    mean|W|; k = 32: W; else 2 quantize_k(1/2 + tanh W / (2 m)) - 1 with m = max|tanh W| (dense) or tanh(max|W|)
    (conv) — the same number mathematically, different roundings."""
    w = np.asarray(weight, dtype=np.float32)
    if k == 1:
        return safe_sign(w) * np.float32(np.mean(np.abs(w), dtype=np.float64))
    if k == 32:
        return w
    t = np.tanh(w).astype(np.float32)
    m = np.tanh(np.max(np.abs(w))).astype(np.float32) if conv else np.max(np.abs(t))
    return np.float32(2) * dorefa_quantize(np.float32(0.5) + t / (np.float32(2) * m), k) - np.float32(1)


def dorefa_w1a_linear(x_real, weight, bias, k_act=4):
    """nnDorefaQuant(k_act)(relu(x)) -> LinearDorefa(bit_width=1) (layers/dorefa_layers.py:41-45,
    functions/dorefa_connect.py:24-25,99-102), evaluated the way the int8 path factors it:
    y = (E/n) * sum q*s + b with integer q = rint(n * relu(x)), s = safeSign(W), E = mean|W| —
    in double, as the fp64 evaluation the tolerance is stated against."""
    n = float(2 ** k_act - 1)
    x = np.maximum(np.asarray(x_real, dtype=np.float32), 0)
    q = np.rint(np.float32(n) * x).astype(np.float64)
    s = safe_sign(weight).astype(np.float64)
    E = float(np.mean(np.abs(np.asarray(weight, dtype=np.float32)), dtype=np.float32))
    inv_n = float(np.float32(1) / np.float32(n))
    y = (q @ s.T) * (E * inv_n)
    if bias is not None:
        y = y + np.asarray(bias, dtype=np.float64)
    return y.astype(np.float32)


# ---- the eval-mode chain between two binarised layers ------------------------------------------

def maxpool2d(x, k, s):
    """nn.MaxPool2d(k, s) (no padding, floor mode) on NCHW, as models/Alexnet/Alexnet_Bin.py:14 uses it."""
    x = np.asarray(x, dtype=np.float32)
    Nb, C, H, W = x.shape
    Ho, Wo = (H - k) // s + 1, (W - k) // s + 1
    out = np.full((Nb, C, Ho, Wo), -np.inf, dtype=np.float32)
    for i in range(k):
        for j in range(k):
            out = np.maximum(out, x[:, :, i:i + s * Ho:s, j:j + s * Wo:s])
    return out


def pool_bn_sign_planes(y, alpha, beta, pool_k=1, pool_s=1):
    """[MaxPool2d] -> eval BatchNorm folded to t = fl(fl(y*alpha) + beta) -> Hardtanh (sign-neutral) ->
    BinaryConnectDeterministic -> NHWC sign plane (models/Alexnet/Alexnet_Bin.py:14-17 in eval mode).
    Returns (plane [N*Ho*Wo][ld] uint32, (Ho, Wo))."""
    y = np.asarray(y, dtype=np.float32)
    if pool_k > 1 or pool_s > 1:
        y = maxpool2d(y, pool_k, pool_s)
    a = np.asarray(alpha, dtype=np.float32).reshape(1, -1, 1, 1)
    b = np.asarray(beta, dtype=np.float32).reshape(1, -1, 1, 1)
    t = (y * a).astype(np.float32) + b          # two fp32 roundings
    Nb, C, Ho, Wo = t.shape
    nhwc = np.ascontiguousarray(t.astype(np.float32).transpose(0, 2, 3, 1)).reshape(Nb * Ho * Wo, C)
    return sign_pack(nhwc), (Ho, Wo)


def bin_conv_pool_bn_sign_planes(x, weight_q, bias, stride, padding, dilation, alpha, beta, pool_k=1, pool_s=1):
    """Eval-mode BinConv2d / TerConv2d (weight already quantised, layers/binary_layers.py:106) followed by
    pool_bn_sign_planes: what FusedConvPoolBnSign must reproduce bit for bit."""
    y = conv2d(x, weight_q, bias, stride, padding, dilation)
    return pool_bn_sign_planes(y, alpha, beta, pool_k, pool_s)
