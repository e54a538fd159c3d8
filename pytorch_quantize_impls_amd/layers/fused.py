"""Inference-time fusion of the chain the reference puts between two binarised layers
(SURVEY.md section 8f, n1):

    [MaxPool2d] -> BatchNorm{1,2}d (eval) -> [Hardtanh] -> BinaryConnect(deterministic)

(models/Alexnet/Alexnet_Bin.py:14-17, benchmark/BinaryNet/MLPBin.py:42-44).  In eval mode it is a
per-channel threshold on the max-pooled tensor; ``FusedPoolBnSign`` evaluates it in ONE HIP kernel
(qt_pool_affine_sign_pack_nhwc) that emits the sign bit plane directly as a ``PackedActivation`` —
none of the intermediate fp32 tensors is materialised, and the next BinConv2d / LinearBin / *Ter layer
consumes the planes without re-reading anything.  Opt-in (``fuse_sequential``): the un-fused modules keep
working exactly as in the reference.

Numerics — two folds (``fold=`` of the fused modules):

  "reference"  x*alpha + beta with alpha = weight * (1/sqrt(var+eps)), beta = bias - mean*alpha — the fold ATen's own
               CPU kernel performs — evaluated with two fp32 roundings (what oracle/ restates).  A value within an ulp of
               the threshold can land on the other side of it than in a differently-ordered evaluation (MIOpen's
               fma((x-mean)*rsqrt(var+eps), weight, bias), tools/probes/bn_eval_arith.py); everything else is identical.
  "device"     the threshold of THIS device's own ``F.batch_norm``: eval BatchNorm is a monotone function of x per
               channel, so [BatchNorm(x) < 0] == [x < theta_c] (or [x > theta_c] for a negative slope); theta_c is
               found by bisection over the fp32 bit patterns on the function torch really evaluates (same dtype, rank
               and memory format as the tensor the module graph would hand it), once per BatchNorm version.  The sign
               planes then equal the module-by-module graph's bit for bit, whatever arithmetic the library uses.
               Deferred activations (lazy.py) use this fold, so the un-modified graph gives the eager result exactly.
"""
import torch

from .. import ops, packed, lazy
from ..functions.common import QtFunction, _FunctionModule
from ..functions.binary_connect import BinaryConnectDeterministic


def fold_batchnorm(bn):
    """(alpha, beta) of eval-mode BatchNorm: y = x*alpha + beta."""
    invstd = 1.0 / torch.sqrt(bn.running_var.detach() + bn.eps)
    w = bn.weight.detach() if bn.affine else torch.ones_like(invstd)
    b = bn.bias.detach() if bn.affine else torch.zeros_like(invstd)
    alpha = invstd * w
    beta = b - bn.running_mean.detach() * alpha
    return alpha.float().contiguous(), beta.float().contiguous()


_BIG = 1.0e30     # no activation on this path comes near it (|accumulator| <= K)


def _float_key(x):
    """Monotone int64 key of fp32 values: key order == numeric order (-0.0 just below +0.0)."""
    b = x.view(torch.int32).to(torch.int64)
    return torch.where(b >= 0, b, -(b + (1 << 31)) - 1)


def _key_float(k):
    b = torch.where(k >= 0, k, (-k - 1) - (1 << 31))
    return b.to(torch.int32).view(torch.float32)


def device_sign_fold(bn, like_shape, channels_last: bool):
    """(alpha', beta') with alpha' in {+1, -1, 0} such that, for every fp32 x of channel c (|x| < 1e30),

        [ fl(x * alpha'_c + beta'_c) < 0 ]  ==  [ F.batch_norm(x)_c < 0 ]     as THIS device evaluates eval BatchNorm

    ``like_shape`` / ``channels_last``: rank, (capped) extents and memory format of the tensor the module graph would
    hand to F.batch_norm — the probe tensors have the same, so torch picks the same kernel family.  33 evaluations of
    F.batch_norm on a [<=2, C, <=2, <=2] (or [<=2, C]) tensor; callers cache the result per BatchNorm version."""
    import torch.nn.functional as F
    rm, rv = bn.running_mean.detach(), bn.running_var.detach()
    w = bn.weight.detach() if bn.affine else None
    b = bn.bias.detach() if bn.affine else None
    C, dev = int(rm.numel()), rm.device
    if len(like_shape) == 2:
        shape = (max(1, min(2, int(like_shape[0]))), C)
    else:
        shape = (max(1, min(2, int(like_shape[0]))), C, max(1, min(2, int(like_shape[2]))), max(1, min(2, int(like_shape[3]))))
    fmt = torch.channels_last if (channels_last and len(shape) == 4) else torch.contiguous_format

    def neg(v):
        if len(shape) == 2:
            y = F.batch_norm(v.unsqueeze(0).expand(shape).contiguous(), rm, rv, w, b, False, 0.0, bn.eps)[0]
        else:
            inp = v.view(1, C, 1, 1).expand(shape).contiguous(memory_format=fmt)
            y = F.batch_norm(inp, rm, rv, w, b, False, 0.0, bn.eps)[0, :, 0, 0]
        return y < 0

    with torch.no_grad():
        big = torch.full((C,), _BIG, dtype=torch.float32, device=dev)
        n_lo, n_hi = neg(-big), neg(big)
        dec = ~n_lo & n_hi                       # negative slope: negative for LARGE x
        const_neg, const_pos = n_lo & n_hi, ~n_lo & ~n_hi
        lo, hi = _float_key(-big), _float_key(big)          # g(lo) true, g(hi) false with g = neg xor dec
        for _ in range(33):
            mid = torch.div(lo + hi, 2, rounding_mode="floor")
            g = neg(_key_float(mid)) ^ dec
            lo = torch.where(g, mid, lo)
            hi = torch.where(g, hi, mid)
        one = torch.ones((C,), dtype=torch.float32, device=dev)
        # rising: neg(x) <=> x < float(hi);  falling: neg(x) <=> x > float(lo)
        alpha = torch.where(dec, -one, one)
        beta = torch.where(dec, _key_float(lo), -_key_float(hi))
        const = const_neg | const_pos
        alpha = torch.where(const, torch.zeros_like(one), alpha)
        beta = torch.where(const_neg, -one, torch.where(const_pos, one, beta))
    return alpha.contiguous(), beta.contiguous()


def device_bn_fold(bn, like_shape, channels_last: bool):
    """(weight, bias, stats) with stats = fp32 [mean | rs] such that, for every fp32 x of channel c,

        fma(fl(fl(x - mean_c) * rs_c), weight_c, bias_c)  ==  F.batch_norm(x)_c      bit for bit on THIS device

    rs is read back from the device's own eval-mode kernel (F.batch_norm(1, mean 0, running_var, weight 1, bias 0) = rs):
    the library's reciprocal square root is not the correctly rounded one (tools/probes/bn_eval_emulation.py: 100 % bit
    agreement with the read-back value, 91-94 % with torch.rsqrt).  The emulation is VERIFIED against F.batch_norm on a
    probe (512 values per channel around and far from zero, same rank / memory format as the real tensor); a mismatch —
    another library version, another kernel family — raises ValueError, which sends a deferred chain to its eager path."""
    import torch.nn.functional as F
    rm, rv = bn.running_mean.detach(), bn.running_var.detach()
    C, dev = int(rm.numel()), rm.device
    one, zero = torch.ones((C,), dtype=torch.float32, device=dev), torch.zeros((C,), dtype=torch.float32, device=dev)
    w = bn.weight.detach().float() if bn.affine else one
    b = bn.bias.detach().float() if bn.affine else zero
    if len(like_shape) != 4:
        raise ValueError("the device BatchNorm arithmetic is probed for 4-D activations")
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    n_, h_, w_ = max(1, min(2, int(like_shape[0]))), max(1, min(2, int(like_shape[2]))), max(1, min(2, int(like_shape[3])))
    with torch.no_grad():
        ones = torch.ones((n_, C, h_, w_), dtype=torch.float32, device=dev).contiguous(memory_format=fmt)
        rs = F.batch_norm(ones, zero, rv, one, zero, False, 0.0, bn.eps)[0, :, 0, 0].contiguous()
        g = torch.Generator(device=dev).manual_seed(1234)
        probe = torch.randn((8, C, 8, 8), generator=g, device=dev) * (torch.sqrt(rv + bn.eps) * 3).view(1, C, 1, 1) + rm.view(1, C, 1, 1)
        probe[:, :, ::2] *= 17.0
        probe = probe.contiguous(memory_format=fmt)
        want = F.batch_norm(probe, rm, rv, w if bn.affine else None, b if bn.affine else None, False, 0.0, bn.eps)
        # the fma formed in fp64 from the fp32 intermediate: the product of two fp32 values is exact in fp64, so rounding the
        # fp64 sum to fp32 once is the fused multiply-add (up to double rounding, which the equality below would expose)
        t32 = ((probe - rm.view(1, C, 1, 1)) * rs.view(1, C, 1, 1))
        got = (t32.double() * w.double().view(1, C, 1, 1) + b.double().view(1, C, 1, 1)).float()
        if not torch.equal(got, want):
            raise ValueError("eval-mode F.batch_norm of this device is not fma((x - mean) * rs, weight, bias): "
                             f"{int((got != want).sum())} of {want.numel()} probe values differ")
    stats = torch.cat([rm.float(), rs]).contiguous()
    return w.contiguous(), b.contiguous(), stats


def _out_channels_last(x) -> bool:
    """Memory format of the fp32 tensor a quantised conv returns for input ``x`` (functions/_fused.py: NCHW-contiguous
    tensors get NCHW storage back, everything else — channels-last tensors, packed activations — NHWC storage)."""
    return not (isinstance(x, torch.Tensor) and x.dim() == 4 and x.is_contiguous()
                and not x.is_contiguous(memory_format=torch.channels_last))


def _bn_key(bn):
    """Identity of a BatchNorm's parameters / statistics: (storage, version counter) of each tensor — load_state_dict(),
    copy_() and .to(device) change it, so a folded (alpha, beta) cached under it is never stale.  (Writes through
    ``.data`` do not bump version counters: call refold() after those.)"""
    ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    return tuple((t.data_ptr(), t._version) if t is not None else None for t in ts)


def _folded_for(owner, attr, bn, device=None, fold="reference", like=None):
    """fold_batchnorm(bn) — or, fold == "device", device_sign_fold(bn, *like) with like = (shape, channels_last) —
    cached on ``owner`` under ``attr`` for the current _bn_key (and, for the device fold, the probe signature)."""
    key = _bn_key(bn)
    if fold == "device":
        shape, cl = like
        sig = (len(shape), min(2, int(shape[0]))) + (tuple(min(2, int(v)) for v in shape[2:]) if len(shape) == 4 else ()) + (bool(cl),)
        key = key + (sig,)
    elif fold != "reference":
        raise ValueError(f"fold must be 'reference' or 'device', got {fold!r}")
    cur = getattr(owner, attr, None)
    if cur is None or cur[0] != key or (device is not None and cur[1][0].device != device):
        cur = (key, device_sign_fold(bn, *like) if fold == "device" else fold_batchnorm(bn))
        setattr(owner, attr, cur)
    return cur[1]


def _code_fold_for(owner, attr, bn, fold, like):
    """BatchNorm parameters of a DoReFa code epilogue, cached like ``_folded_for``: (alpha, beta) of the ATen-CPU fold
    ("reference") or (weight, bias, [mean | rs]) of this device's own arithmetic ("device": ``device_bn_fold``)."""
    if fold != "device":
        return _folded_for(owner, attr, bn)
    shape, cl = like
    key = _bn_key(bn) + (("codes", len(shape), bool(cl)),)
    cur = getattr(owner, attr, None)
    if cur is None or cur[0] != key:
        cur = (key, device_bn_fold(bn, shape, cl))
        setattr(owner, attr, cur)
    return cur[1]


class FusedPoolBnSign(torch.nn.Module):
    """[MaxPool2d(k, s)] + eval BatchNorm + [Hardtanh] + BinaryConnect(deterministic) -> PackedActivation."""

    def __init__(self, bn, pool=None, flatten_hwc=False, pre_relu=False, fold=None):
        super().__init__()
        self.fold = fold or DEFAULT_FOLD
        self.pre_relu = bool(pre_relu)     # ReLU between the pooling and the BatchNorm (MLPBin.py:42-44 pattern)
        if pool is not None:
            k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
            st = pool.stride if isinstance(pool.stride, int) else pool.stride[0]
            pad = pool.padding if isinstance(pool.padding, int) else pool.padding[0]
            dil = pool.dilation if isinstance(pool.dilation, int) else pool.dilation[0]
            if pad != 0 or dil != 1 or pool.ceil_mode:
                raise ValueError("only un-padded, un-dilated, floor-mode MaxPool2d can be fused")
            self.pool_k, self.pool_s = int(k), int(st)
        else:
            self.pool_k = self.pool_s = 1
        self.bn = bn
        self.flatten_hwc = flatten_hwc
        self._folded = None

    def refold(self):
        self._folded = None

    def forward(self, x):
        if self.bn.training:
            raise RuntimeError("FusedPoolBnSign is an inference module: call .eval() first")
        x = lazy.resolve(x)
        if not x.is_cuda:
            raise TypeError("FusedPoolBnSign runs on a HIP device only (use the un-fused modules on CPU)")
        like = None
        if self.fold == "device":
            if x.dim() == 4:
                N, C, H, W = x.shape
                like = ((N, C, (H - self.pool_k) // self.pool_s + 1, (W - self.pool_k) // self.pool_s + 1),
                        x.is_contiguous(memory_format=torch.channels_last))
            else:
                like = (tuple(x.shape), False)
        alpha, beta = _folded_for(self, "_folded", self.bn, x.device, self.fold, like)
        # rows of a classifier (2-D): the next quantised Linear takes them on the matrix cores from batch 33 on (ops.select_gemm_impl)
        # — its fp4 nibble operand comes out of this same launch
        planes, (Ho, Wo) = ops.pool_affine_sign_pack(x, alpha, beta, self.pool_k, self.pool_s, pre_relu=self.pre_relu,
                                                     want_nib=x.dim() == 2 and x.shape[0] > 32)
        if x.dim() == 2:
            return packed.PackedActivation(planes, (x.shape[0], x.shape[1]))
        act = packed.PackedActivation(planes, (x.shape[0], x.shape[1], Ho, Wo))
        return act.flatten_hwc() if self.flatten_hwc else act


class FusedBnDorefaQuant(torch.nn.Module):
    """eval BatchNorm [+ residual] [-> ReLU] -> nnDorefaQuant(k) over the fp32 output of a conv / linear layer, as
    ONE kernel that writes the next layer's int8 code plane (``packed.CodeActivation``): the
    ``quant(relu(bn(conv(x))))`` / ``quant(relu(bn2(conv2(.)) + shortcut))`` chains of a DoReFa ResNet
    (models/samples/ResNet_Dorefa.py:26,35) without BatchNorm / add / ReLU / quantiser passes over fp32 tensors.

    ``relu``: True / "post" (after BatchNorm and residual), "pre" (before the BatchNorm: the Linear -> ReLU ->
    BatchNorm -> quant order of models/FullNet/DorefaMNIST.py:46-48) or False.

    forward(x, residual=None, residual_bn=None):
      x           fp32 [N, C, H, W] (channels_last, as the conv kernels return it) or [N, C]
      residual    None | CodeActivation (identity shortcut; value inv_n * code) | fp32 tensor shaped like x
      residual_bn BatchNorm applied to an fp32 residual first (the 1x1-conv shortcut's own BatchNorm)
    BatchNorm is folded to t = fl(fl(x*alpha) + beta) (``fold_batchnorm``); the oracle restates exactly this chain
    (oracle.affine_relu_dorefa_codes).  The quantiser is unclamped like the reference's, so codes may leave int8:
    the kernel raises a device flag shared along the chain and ``CodeActivation.check()/float()`` raises."""

    def __init__(self, bn, bit_width: int, relu=True, out_halo=0, fold=None):
        super().__init__()
        if not 2 <= int(bit_width) <= 8:
            raise ValueError("code planes exist for 2 <= bit_width <= 8")
        self.fold = fold or DEFAULT_FOLD
        self.bn, self.bit_width, self.relu = bn, int(bit_width), ops.relu_mode(relu)
        # out_halo: zero border for the consuming conv's padding (see FusedDorefaConvBnQuant): the pass writes into the halo plane
        # and zeroes its border itself (qt_affine_dorefa_codes_halo_i8)
        self.out_halo = (int(out_halo),) * 2 if isinstance(out_halo, int) else tuple(int(v) for v in out_halo)
        self._folded = None
        self._folded_res = None

    def refold(self):
        self._folded = self._folded_res = None

    def forward(self, x, residual=None, residual_bn=None):
        if self.bn.training:
            raise RuntimeError("FusedBnDorefaQuant folds running statistics: call .eval() first")
        x, residual = lazy.resolve(x), lazy.resolve(residual)
        like = (tuple(x.shape), x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last)) if self.fold == "device" else None
        folded = _code_fold_for(self, "_folded", self.bn, self.fold, like)
        alpha, beta, stats = folded[0], folded[1], (folded[2] if len(folded) > 2 else None)
        if x.dim() == 4:
            N, C, H, W = x.shape
            x2 = x.permute(0, 2, 3, 1)
            if not x2.is_contiguous():
                x2 = x2.contiguous()
            x2 = x2.view(N * H * W, C)
        elif x.dim() == 2:
            x2 = x.contiguous()
        else:
            raise ValueError("FusedBnDorefaQuant takes [N, C, H, W] or [N, C] inputs")
        res_f32 = res_codes = res_affine = None
        flag = getattr(x, "_qt_overflow", None)
        if isinstance(residual, packed.CodeActivation):
            if residual.shape != tuple(x.shape):
                raise ValueError(f"residual {residual.shape} vs input {tuple(x.shape)}")
            res_codes = residual.without_halo().codes
            flag = flag if flag is not None else residual.codes.overflow
        elif residual is not None:
            if tuple(residual.shape) != tuple(x.shape):
                raise ValueError(f"residual {tuple(residual.shape)} vs input {tuple(x.shape)}")
            r2 = residual.permute(0, 2, 3, 1) if residual.dim() == 4 else residual
            res_f32 = (r2 if r2.is_contiguous() else r2.contiguous()).view(x2.shape)
            if residual_bn is not None:
                rlike = (tuple(residual.shape), residual.dim() == 4 and residual.is_contiguous(memory_format=torch.channels_last)) \
                    if self.fold == "device" else None
                res_affine = _code_fold_for(self, "_folded_res", residual_bn, self.fold, rlike)
        halo = x.dim() == 4 and any(self.out_halo)
        codes, _ = ops.affine_dorefa_codes(x2, alpha, beta, self.bit_width, self.relu, res_f32, res_affine, res_codes,
                                           overflow=flag,
                                           ld_bytes=ops.code_ld_bytes(x2.shape[1], 16) if x.dim() == 4 else None,
                                           bn_stats=stats, halo_nhw=(N, H, W) if halo else None,
                                           out_halo=self.out_halo if halo else (0, 0))
        return packed.CodeActivation(codes, x.shape, halo=self.out_halo if halo else (0, 0))


class FusedDorefaConvBnQuant(torch.nn.Module):
    """DorefaConv2d(bit_width=1, eval) -> BatchNorm(eval) [+ residual] [-> ReLU] -> nnDorefaQuant(k) with the whole
    tail in the conv kernel's epilogue (qt_conv2d_implicit_codes): CodeActivation in, CodeActivation out, no fp32
    activation in HBM.  Same arithmetic as ``FusedBnDorefaQuant`` applied to the conv's fp32 output (the epilogue
    forms exactly the value the conv would have stored), so the two are bit-identical.

    forward(act, residual=None, residual_bn=None, residual_conv=None): residual over the conv's OUTPUT pixels, as
    FusedBnDorefaQuant.  ``residual_conv`` = (DorefaConv2d, CodeActivation) instead of ``residual``: the shortcut branch
    conv -> ``residual_bn`` evaluated here, in ONE launch when the conv is a 1-bit DorefaConv2d and the fold is "device"
    (qt_conv2d_implicit_halo_bn: BatchNorm in the conv's epilogue), else as conv, then BatchNorm."""

    def __init__(self, conv, bn, bit_width: int, relu=True, out_halo=0, fold=None):
        super().__init__()
        if getattr(conv, "bit_width", None) != 1 or conv.groups != 1 or conv.padding_mode != "zeros":
            raise ValueError("FusedDorefaConvBnQuant takes an un-grouped, zero-padded DorefaConv2d(bit_width=1)")
        self.fold = fold or DEFAULT_FOLD
        self.conv, self.bn, self.bit_width, self.relu = conv, bn, int(bit_width), ops.relu_mode(relu)
        # out_halo = the padding of the conv(s) that consume this activation: the epilogue writes into a plane with
        # that zero border, so they run the un-padded kernels (a residual CodeActivation may carry any halo)
        self.out_halo = (int(out_halo),) * 2 if isinstance(out_halo, int) else tuple(int(v) for v in out_halo)
        self._folded = self._folded_res = None

    def refold(self):
        self._folded = self._folded_res = None

    def shortcut_in_one_launch(self, sc_conv, sc_act, residual_bn) -> bool:
        """Does the conv -> BatchNorm shortcut branch run as ONE launch (BatchNorm in the conv's epilogue)?  lazy._force_codes asks
        before it hands the un-materialised branch over: where the answer is no it materialises the branch's root itself (which
        caches the value for the branch's other consumers) instead of letting the fallback below re-enter the deferred forward."""
        if not (residual_bn is not None and self.fold == "device" and isinstance(sc_act, packed.CodeActivation)
                and getattr(sc_conv, "bit_width", None) == 1 and sc_conv.groups == 1 and sc_conv.padding_mode == "zeros"
                and not isinstance(sc_conv.padding, str) and not sc_conv.training and sc_conv.out_channels % 4 == 0):
            return False
        kh, kw = sc_conv.kernel_size
        return (sc_act.codes.K == sc_act.shape[1] == sc_conv.in_channels
                and 127 * kh * kw * sc_act.codes.codes.shape[1] < (1 << 24))

    def _shortcut(self, sc_conv, sc_act, residual_bn):
        """The conv -> BatchNorm shortcut branch: (fp32 [M, C] matrix already normalised, None) or (conv output, its BatchNorm)."""
        if self.shortcut_in_one_launch(sc_conv, sc_act, residual_bn):
            N_, C_, H_, W_ = sc_act.shape
            kh, kw = sc_conv.kernel_size
            if True:
                Ho_, Wo_ = ops.conv_out_hw(H_, W_, kh, kw, sc_conv.stride, sc_conv.padding, sc_conv.dilation)
                rw, rb, rstats = _code_fold_for(self, "_folded_res", residual_bn, "device", ((N_, sc_conv.out_channels, Ho_, Wo_), True))
                wc = sc_conv._eval_planes(lambda _w2: ops.pack_conv_weight_codes(sc_conv.weight.detach()), key="conv_i8")
                E = sc_conv._eval_planes(lambda w2: w2.abs().amax(), key="E")
                y2 = ops.conv2d_codes(sc_act.codes, sc_act.shape, wc, (kh, kw), sc_act.codes.inv_n, sc_conv.bias, sc_conv.stride,
                                      sc_conv.padding, sc_conv.dilation, scale_dev=E, epi=ops.BnEpilogue(rw, rb, rstats),
                                      in_halo=sc_act.halo)
                return y2, None
        with lazy.eager():                            # (explicit callers: the branch conv's ordinary eval path, never a deferred node)
            return sc_conv(sc_act), residual_bn

    def forward(self, act, residual=None, residual_bn=None, residual_conv=None):
        conv = self.conv
        if conv.training or self.bn.training:
            raise RuntimeError("FusedDorefaConvBnQuant is an inference form: call .eval() first")
        if not isinstance(act, packed.CodeActivation):
            raise TypeError("FusedDorefaConvBnQuant consumes a CodeActivation (FusedBnDorefaQuant output)")
        if residual_conv is not None:
            if residual is not None:
                raise ValueError("pass the shortcut either as its value (residual) or as its conv (residual_conv)")
            residual, residual_bn = self._shortcut(residual_conv[0], residual_conv[1], residual_bn)
        # the tensor the module graph hands to F.batch_norm is this conv's fp32 output: channels-last for a code-plane input
        N_, _, H_, W_ = act.shape
        Ho_, Wo_ = ops.conv_out_hw(H_, W_, conv.kernel_size[0], conv.kernel_size[1], conv.stride, conv.padding, conv.dilation)
        like = ((N_, conv.out_channels, Ho_, Wo_), True) if self.fold == "device" else None
        folded = _code_fold_for(self, "_folded", self.bn, self.fold, like)
        epi = ops.CodeEpilogue(folded[0], folded[1], self.bit_width, self.relu, out_halo=self.out_halo,
                               bn_stats=folded[2] if len(folded) > 2 else None)
        if isinstance(residual, packed.CodeActivation):
            epi.res_codes, epi.res_halo = residual.codes, residual.halo
        elif residual is not None:
            r2 = residual.permute(0, 2, 3, 1) if residual.dim() == 4 else residual
            epi.res_f32 = (r2 if r2.is_contiguous() else r2.contiguous()).view(-1, r2.shape[-1])
            if epi.res_f32.shape != (N_ * Ho_ * Wo_, conv.out_channels):
                raise ValueError(f"residual {tuple(residual.shape)} does not cover this conv's output")
            if residual_bn is not None and self.fold == "device":
                # device arithmetic: the shortcut's BatchNorm runs as its own elementwise pass in the same (verified) expression;
                # the conv epilogue then adds a plain fp32 residual and carries no second set of per-channel registers
                rlike = (tuple(residual.shape), residual.dim() == 4 and residual.is_contiguous(memory_format=torch.channels_last))
                rw, rb, rstats = _code_fold_for(self, "_folded_res", residual_bn, "device", rlike)
                r2 = residual.permute(0, 2, 3, 1) if residual.dim() == 4 else residual
                r2 = (r2 if r2.is_contiguous() else r2.contiguous()).view(-1, r2.shape[-1])
                if r2.shape[1] % 4 == 0:
                    epi.res_f32 = ops.bn_eval_device(r2, rw, rb, rstats)
                else:
                    rb_ = residual_bn
                    rn = torch.nn.functional.batch_norm(residual, rb_.running_mean, rb_.running_var, rb_.weight if rb_.affine else None,
                                                        rb_.bias if rb_.affine else None, False, 0.0, rb_.eps)
                    rn = rn.permute(0, 2, 3, 1) if rn.dim() == 4 else rn
                    epi.res_f32 = (rn if rn.is_contiguous() else rn.contiguous()).view(-1, rn.shape[-1])
            elif residual_bn is not None:
                epi.res_affine = _folded_for(self, "_folded_res", residual_bn)
        from ..functions import _fused
        wc = conv._eval_planes(lambda _w2: ops.pack_conv_weight_codes(conv.weight.detach()), key="conv_i8")
        E = conv._eval_planes(lambda w2: w2.abs().amax(), key="E")
        return _fused.dorefa_w1_conv_forward(act, conv.weight, conv.bias,
                                             (conv.stride, conv.padding, conv.dilation, conv.groups), True, wc,
                                             conv.padding_mode, scale=E, epi=epi)


class CodeMaxPool(torch.nn.Module):
    """MaxPool2d on a CodeActivation (the reference's DoReFa CNNs pool AFTER the quantiser,
    models/samples/AlexNet_Dorefa.py:38-41): the max of the int8 codes, bit-identical to pooling the fp32 image.
    ``out_halo``: zero border for the next conv's padding."""

    def __init__(self, pool, out_halo=0):
        super().__init__()
        k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
        st = pool.stride if isinstance(pool.stride, int) else pool.stride[0]
        pad = pool.padding if isinstance(pool.padding, int) else pool.padding[0]
        dil = pool.dilation if isinstance(pool.dilation, int) else pool.dilation[0]
        if pad != 0 or dil != 1 or pool.ceil_mode:
            raise ValueError("only un-padded, un-dilated, floor-mode MaxPool2d runs on code planes")
        self.pool_k, self.pool_s = int(k), int(st)
        self.out_halo = (int(out_halo),) * 2 if isinstance(out_halo, int) else tuple(int(v) for v in out_halo)

    def forward(self, act):
        if not isinstance(act, packed.CodeActivation) or len(act.shape) != 4:
            raise TypeError("CodeMaxPool consumes the (N, C, H, W) CodeActivation of a fused DoReFa block")
        act = act.without_halo()
        N, C, H, W = act.shape
        out = ops.pool_codes(act.codes, N, H, W, self.pool_k, self.pool_s, self.out_halo)
        Ho, Wo = (H - self.pool_k) // self.pool_s + 1, (W - self.pool_k) // self.pool_s + 1
        return packed.CodeActivation(out, (N, C, Ho, Wo), halo=self.out_halo)


class FusedConvPoolBnSign(torch.nn.Module):
    """BinConv2d / TerConv2d (eval) + [MaxPool2d(k, s)] + eval BatchNorm2d + [Hardtanh] + BinaryConnect(det)
    -> PackedActivation, with NO fp32 activation in between:

      * the conv runs with the threshold-bit epilogue (qt_conv2d_implicit_bits): per output pixel and
        channel the bit [(acc + bias) * alpha + beta < 0] — 1/32 of the fp32 tensor;
      * MaxPool is evaluated on those bits (qt_pool_bits): max-pooling commutes with the monotone map
        x*alpha + beta, so the pooled sign is the AND (alpha >= 0) / OR (alpha < 0) of the window's bits.

    Bit-identical to conv -> FusedPoolBnSign (and hence to the un-fused chain up to BatchNorm's own
    evaluation order, see the module docstring) for NaN-free activations.  Input: a PackedActivation, or a
    real-valued device tensor (first layer; exact bf16-triple conv)."""

    def __init__(self, conv, bn, pool=None, flatten_hwc=False, fold=None):
        super().__init__()
        from .binary_layers import BinConv2d
        from .terner_layers import TerConv2d
        from .xnor_layers import XNORConv2d
        if isinstance(conv, BinConv2d):
            self.kind = "binary"
        elif isinstance(conv, TerConv2d):
            self.kind = "ternary"
        elif isinstance(conv, XNORConv2d) and conv._qt_can_defer:
            self.kind = "xnor"           # per-tap scaled conv (real-valued accumulators: float-form threshold epilogue)
        else:
            raise ValueError("FusedConvPoolBnSign fuses BinConv2d / TerConv2d / XNORConv2d(dim=[0, 1]) only")
        if conv.groups != 1 or conv.padding_mode != "zeros" or isinstance(conv.padding, str):
            raise ValueError("only groups == 1, zero-padded convs can be fused")
        if not isinstance(bn, torch.nn.BatchNorm2d) or bn.num_features != conv.out_channels:
            raise ValueError("BatchNorm2d over the conv's output channels expected")
        self._pool = FusedPoolBnSign(bn, pool, fold=fold)           # validates the pooling geometry, owns the fold cache
        self.fold = self._pool.fold
        self.conv, self.bn = conv, bn
        self.flatten_hwc = flatten_hwc
        self._neg_alpha = None
        # (ph, pw) of the fused conv that consumes this block's output (set by fuse_sequential): the block then writes
        # that conv's operand itself — an fp4 nibble pixel plane with the padding as a zero border — from the conv
        # epilogue (no pooling) or from the pooling kernel, instead of bit planes the consumer would expand again
        self.out_nib_halo = None
        self._thr = None

    def refold(self):
        self._pool.refold()
        self._neg_alpha = None
        self._thr = None

    def forward(self, x):
        from ..functions import _fused
        if self.bn.training or self.conv.training:
            raise RuntimeError("FusedConvPoolBnSign is an inference module: call .eval() first")
        conv, fp = self.conv, self._pool
        x = lazy.resolve(x)
        dev = conv.weight.device
        prev = fp._folded
        like = None
        if self.fold == "device":
            # the tensor the module graph would hand to F.batch_norm: this conv's output, pooled
            N, _, H, W = (int(v) for v in x.shape)
            Ho, Wo = ops.conv_out_hw(H, W, conv.kernel_size[0], conv.kernel_size[1], conv.stride, conv.padding, conv.dilation)
            like = ((N, conv.out_channels, (Ho - fp.pool_k) // fp.pool_s + 1, (Wo - fp.pool_k) // fp.pool_s + 1),
                    _out_channels_last(x))
        epi = _folded_for(fp, "_folded", self.bn, dev, self.fold, like)
        if fp._folded is not prev or self._neg_alpha is None:      # BatchNorm changed (or first call): derived data too
            self._neg_alpha = ops.neg_alpha_words(epi[0])
            self._thr = None
        pooled = fp.pool_k != 1 or fp.pool_s != 1
        nib_out = self.out_nib_halo is not None and not self.flatten_hwc
        if isinstance(x, packed.PackedActivation):
            # +-1 activations x +-1 / 0 weights: the accumulator is an exact integer, so BatchNorm + sign is one integer
            # threshold per channel (found once, by bisection on the kernel's own fp32 expression)
            bkey = None if conv.bias is None else (conv.bias.data_ptr(), conv.bias._version)
            int_thr = INTEGER_THRESHOLDS and self.kind != "xnor"
            if int_thr and (self._thr is None or self._thr[0] != bkey or self._thr[1].device != dev):
                kmax = conv.in_channels * conv.kernel_size[0] * conv.kernel_size[1]
                self._thr = (bkey, ops.integer_thresholds(conv.bias, epi[0], epi[1], kmax))
            thr = self._thr[1] if int_thr else None
            epi = (epi[0], epi[1], thr)
            if nib_out and not pooled:
                epi = ops.NibEpilogue(epi[0], epi[1], self.out_nib_halo, thr=thr)
            planes, shape = _fused.packed_conv2d(conv, x, self.kind, epi=epi)
            if isinstance(planes, ops.NibPlanes):
                return packed.PackedActivation(None, shape, nib=planes, halo=self.out_nib_halo)
        else:
            if not (isinstance(x, torch.Tensor) and x.is_cuda):
                raise TypeError("FusedConvPoolBnSign runs on a HIP device only (use the un-fused modules on CPU)")
            planes = shape = None
            if self.kind == "xnor":
                planes, shape = self._xnor_real_input(x, epi, nib_out and not pooled)
                wp = None
            else:
                wp = conv._eval_planes(lambda _w2: ops.pack_conv_weight_nib(conv.weight.detach(), self.kind), key="conv_nib")
            if planes is not None:
                pass
            elif (self.kind in ("binary", "ternary") and x.dim() == 4 and x.dtype == torch.float32 and conv.binary_input is False
                    and conv.groups == 1 and conv.padding_mode == "zeros" and not isinstance(conv.padding, str)
                    and ops.first3x3_applicable(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride, conv.padding,
                                                conv.dilation)):
                # VGG-style first layer (3 -> 64, 3 x 3): one pass from the fp32 image to threshold bits / the next conv's nibble
                # plane (csrc/conv_first3x3.hip) — the same kernel, hence the same accumulators, as the module-by-module fp32 result
                N, C, H, W = (int(v) for v in x.shape)
                e2 = ops.NibEpilogue(epi[0], epi[1], (1, 1)) if (nib_out and not pooled and tuple(self.out_nib_halo) == (1, 1)) else (epi[0], epi[1])
                planes = ops.conv_first3x3(x, conv._conv_triples("first3x3"), conv.out_channels, conv.bias, epi=e2)
                shape = (N, conv.out_channels, H, W)
                if isinstance(planes, ops.NibPlanes):
                    return packed.PackedActivation(None, shape, nib=planes, halo=(1, 1))
                if planes is not None and nib_out and not pooled:
                    planes = ops.bits_to_nib_pad(planes, N, H, W, self.out_nib_halo, ld=ops.pixel_ld_nib(planes.K))
            elif (DIRECT_FIRST_LAYER and x.dim() == 4 and x.dtype == torch.float32 and conv.binary_input is False
                    and (not nib_out or pooled or tuple(self.out_nib_halo) == (1, 1))
                    and ops.direct_first_layer_applicable(conv.in_channels, conv.out_channels, conv.kernel_size, conv.stride,
                                                          conv.padding, conv.dilation)):
                # real-valued 3x3 / stride-1 / padding-1 first layer: direct kernel on the padded bf16-triple plane
                N, C, H, W = (int(v) for v in x.shape)
                terms = ops.direct_first_layer_terms(C)               # fp16 pair pixels (two taps per MFMA), or exact bf16 triples
                px, _ = ops.s2d_triple_pack(x, 1, 1, terms=terms)
                wtr = conv._conv_triples("plain", terms=terms)
                e2 = ops.NibEpilogue(epi[0], epi[1], (1, 1)) if (nib_out and not pooled) else epi
                planes = ops.conv3x3_direct_nib(px, N, C, H, W, wtr, conv.bias, e2)
                shape = (N, conv.out_channels, H, W)
                if isinstance(planes, ops.NibPlanes):
                    return packed.PackedActivation(None, shape, nib=planes, halo=(1, 1))
            elif nib_out and not pooled:
                if (D2S_FIRST_LAYER and x.dim() == 4 and x.dtype == torch.float32 and conv.binary_input is False
                        and ops.d2s_first_layer_applicable(conv.in_channels, conv.out_channels, conv.kernel_size,
                                                           conv.stride, conv.padding, conv.dilation, x.shape[2], x.shape[3])):
                    return self._first_layer_d2s(x, epi)
                epi = ops.NibEpilogue(epi[0], epi[1], self.out_nib_halo)
            if planes is None:
                planes, shape = _fused.quant_conv2d_forward(
                    x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups, self.kind,
                    weight_q=conv.weight, weight_planes=wp, binary_input=conv.binary_input,
                    padding_mode=conv.padding_mode, weight_triples_fn=conv._conv_triples, epi=epi)
        if isinstance(planes, ops.NibPlanes):
            return packed.PackedActivation(None, shape, nib=planes, halo=self.out_nib_halo)
        N, Cout, Ho, Wo = shape
        if pooled and nib_out:
            nib, (Ho, Wo) = ops.pool_bits_nib(planes, N, Ho, Wo, fp.pool_k, fp.pool_s, self._neg_alpha, self.out_nib_halo)
            return packed.PackedActivation(None, (N, Cout, Ho, Wo), nib=nib, halo=self.out_nib_halo)
        if pooled:
            planes, (Ho, Wo) = ops.pool_bits(planes, N, Ho, Wo, fp.pool_k, fp.pool_s, self._neg_alpha)
        act = packed.PackedActivation(planes, (N, Cout, Ho, Wo))
        return act.flatten_hwc() if self.flatten_hwc else act


    def _xnor_real_input(self, x, epi, nib_out: bool):
        """XNORConv2d on a device tensor inside a fused stack: a +-1 tensor (tagged / detected) takes the per-tap scaled conv, a
        real-valued one (the first layer) the real x real conv on six-term bf16 planes, both with the threshold epilogue."""
        from ..functions import _fused
        conv = self.conv
        N, C, H, W = (int(v) for v in x.shape)
        kh, kw = conv.kernel_size
        Ho, Wo = ops.conv_out_hw(H, W, kh, kw, conv.stride, conv.padding, conv.dilation)
        shape = (N, conv.out_channels, Ho, Wo)
        e2 = ops.NibEpilogue(epi[0], epi[1], self.out_nib_halo) if nib_out else (epi[0], epi[1])
        known = True if packed.lookup(x, packed.NHWC) is not None else conv.binary_input
        out = None
        if known is not False:
            out = _fused.xnor_conv2d_forward(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation,
                                             binary_input=known, planes=conv._taps_planes(), epi=e2)
        if out is not None:
            return out[0], shape
        # the op quantises the eval image AGAIN (sign(w) * mean|w| of values that are already +-alpha: the fp32 mean of n equal
        # numbers is not that number to the last bit), like upstream and like the layer's own forward: same image, same bits
        y = None
        if ops.first_direct_applicable(C, (kh, kw), conv.stride, conv.padding, conv.dilation):
            fw = conv._eval_planes(lambda _w2: ops.pack_first_layer_weight(ops.xnor_weight(conv.weight.detach(), 2)[0],
                                                                           conv.stride[0], real=True), key="first_direct_real")
            y = ops.conv_first_direct(x, fw, conv.bias, conv.stride, conv.padding, epi=(epi[0], epi[1]))
            if y is not None and nib_out:
                y = ops.bits_to_nib_pad(y, N, Ho, Wo, self.out_nib_halo, ld=ops.pixel_ld_nib(y.K))
        if y is None:
            wt = conv._eval_planes(lambda _w2: ops.pack_conv_weight_bf16x6(ops.xnor_weight(conv.weight.detach(), 2)[0]),
                                   key="conv_bf16x6")
            y = ops.real_conv2d(x, conv.weight.detach(), conv.bias, conv.stride, conv.padding, conv.dilation, weight_planes=wt, epi=e2)
        if y is None:
            raise ValueError("XNOR conv outside the implicit kernel's limits")
        return y, shape

    def _first_layer_d2s(self, x, affine):
        """Real-valued 3x3 / stride-1 / padding-1 first layer in its 2x2 output-blocked form (ops.d2s_first_layer_weight):
        space-to-depth(2) bf16-triple gather of the padded image, 2x2-tap conv with 4*Cout columns on the 256-wide tiles,
        depth-to-space in the nibble epilogue.  Same products as the direct form, a quarter of the gathered bytes
        (VGG-16 conv1 at batch 256: 550 -> 290 us)."""
        conv = self.conv
        N, C, H, W = (int(v) for v in x.shape)
        Cout = conv.out_channels

        def build(_w2):
            ws = ops.s2d_weight(ops.d2s_first_layer_weight(conv.weight.detach()), 2)        # [4*Cout, 4*C, 2, 2]
            return tuple(ws.shape), ops.pack_conv_weight_bf16x3(ws, "sign")                  # zeros stay zeros
        ws_shape, wtr = conv._eval_planes(build, key=f"conv_split{ops.split_terms()}_d2s")
        px, (Hs, Ws) = ops.s2d_triple_pack(x, 2, 1)
        alpha, beta = (t.repeat(4) for t in affine)
        bias = conv.bias.detach().repeat(4) if conv.bias is not None else None
        epi = ops.NibEpilogue(alpha, beta, self.out_nib_halo, d2s_cout=Cout)
        nib = ops.float_conv2d(None, torch.empty(ws_shape, device="meta"), "sign", bias, 1, 0, 1, weight_triples=wtr,
                               pixels=px, in_shape=(N, 4 * C, Hs, Ws), epi=epi)
        return packed.PackedActivation(None, (N, Cout, H, W), nib=nib, halo=self.out_nib_halo)


#: real-valued 3x3 / stride-1 / padding-1 first layers of fused stacks run in the 2x2 output-blocked form
D2S_FIRST_LAYER = True
#: real-valued 3x3 / stride-1 / padding-1 first layers with <= 5 channels run on the direct kernel (bf16 triple planes);
#: takes precedence over the output-blocked form
DIRECT_FIRST_LAYER = True
#: BatchNorm fold of fused modules built without an explicit ``fold=`` (module docstring).  Blocks built by lazy.py for
#: the un-modified module graph always use "device".
DEFAULT_FOLD = "reference"

#: fused conv blocks on +-1 activations use per-channel integer thresholds (ops.integer_thresholds) in the epilogue
INTEGER_THRESHOLDS = True


class PackedMaxPool(torch.nn.Module):
    """MaxPool2d(k, s) of a +-1 activation that exists only as bit planes (a pool placed AFTER the sign, as in
    VGG-style stacks: conv -> BatchNorm -> Hardtanh -> BinaryConnect -> MaxPool): max of +-1 values is -1 only if the
    whole window is -1, i.e. the AND of the negative bits — qt_pool_bits with an all-zero alpha-sign mask."""

    def __init__(self, pool):
        super().__init__()
        k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
        st = pool.stride if isinstance(pool.stride, int) else pool.stride[0]
        pad = pool.padding if isinstance(pool.padding, int) else pool.padding[0]
        dil = pool.dilation if isinstance(pool.dilation, int) else pool.dilation[0]
        if pad != 0 or dil != 1 or pool.ceil_mode:
            raise ValueError("only un-padded, un-dilated, floor-mode MaxPool2d can be fused")
        self.pool_k, self.pool_s = int(k), int(st)
        self._zero_mask = None
        self.out_nib_halo = None        # see FusedConvPoolBnSign.out_nib_halo

    def forward(self, act):
        if not isinstance(act, packed.PackedActivation) or len(act.shape) != 4:
            raise TypeError("PackedMaxPool consumes the PackedActivation of a fused conv block")
        if act.planes is None:
            raise ValueError("PackedMaxPool pools bit planes; its producer was linked to hand over a conv operand")
        N, C, H, W = act.shape
        if self._zero_mask is None or self._zero_mask.device != act.device or self._zero_mask.numel() != act.planes.ld:
            self._zero_mask = torch.zeros((act.planes.ld,), dtype=torch.int32, device=act.device)
        if self.out_nib_halo is not None:
            nib, (Ho, Wo) = ops.pool_bits_nib(act.planes, N, H, W, self.pool_k, self.pool_s, self._zero_mask,
                                              self.out_nib_halo)
            return packed.PackedActivation(None, (N, C, Ho, Wo), nib=nib, halo=self.out_nib_halo)
        planes, (Ho, Wo) = ops.pool_bits(act.planes, N, H, W, self.pool_k, self.pool_s, self._zero_mask)
        return packed.PackedActivation(planes, (N, C, Ho, Wo))


class _TrainPoolBnSignFn(QtFunction):
    """[MaxPool2d] -> BatchNorm (batch statistics) -> [Hardtanh] -> BinaryConnectDeterministic as one autograd node on this
    backend's kernels (ops.pool_bn_sign_train / _backward); see FusedTrainPoolBnSign."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, eps, momentum, k, s, lo, hi):
        out, saved = ops.pool_bn_sign_train(x, gamma, beta, running_mean, running_var, eps, momentum, k, s, (lo, hi))
        ctx.saved_chain, ctx.ht = saved, (lo, hi)
        ctx.save_for_backward(gamma, beta)
        ctx.x_nchw = x.dim() == 4 and x.is_contiguous() and not x.is_contiguous(memory_format=torch.channels_last)
        # the sign planes ride along like BinaryConnect's (the next binarised layer takes them without a detection pass)
        if out.dim() == 4:
            planes, _ = ops.sign_pack(out.permute(0, 2, 3, 1))
            out = packed.attach(out, planes, packed.NHWC)
        else:
            planes, _ = ops.sign_pack(out)
            out = packed.attach(out, planes, packed.ROWS_LAST)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        gamma, beta = ctx.saved_tensors
        gx, dgamma, dbeta = ops.pool_bn_sign_train_backward(grad_out, ctx.saved_chain, gamma, beta, ctx.ht)
        if ctx.x_nchw:
            gx = gx.contiguous()
        return (gx, dgamma if gamma is not None else None, dbeta if beta is not None else None) + (None,) * 8


class FusedTrainPoolBnSign(torch.nn.Module):
    """TRAINING-mode form of [MaxPool2d(k, s)] -> BatchNorm{1,2}d -> [Hardtanh] -> BinaryConnect(deterministic), the chain
    between two binarised layers (models/Alexnet/Alexnet_Bin.py:13-17, benchmark/BinaryNet/MLPBin.py:42-44): forward (pooling
    with its argmax, batch statistics in two passes, running-statistics update, normalise + clamp + sign) and backward (STE
    of the sign, Hardtanh mask, BatchNorm backward, max-pool gather) on this backend's kernels (csrc/train_chain.hip) instead
    of torch's max_pool2d / MIOpen's batch_norm / hardtanh and their autograd backwards.  Shares the BatchNorm module (its
    parameters and buffers ARE the model's).  CPU tensors, eval mode, momentum=None or untracked statistics: the module chain
    itself.  Opt-in: ``fuse_sequential_training``."""

    def __init__(self, bn, pool=None, hardtanh=None):
        super().__init__()
        self.bn, self.pool, self.hardtanh = bn, pool, hardtanh
        self.sign = _FunctionModule(BinaryConnectDeterministic)
        self.pool_k = self.pool_s = 1
        if pool is not None:
            k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
            st = pool.stride if isinstance(pool.stride, int) else pool.stride[0]
            pad = pool.padding if isinstance(pool.padding, int) else pool.padding[0]
            dil = pool.dilation if isinstance(pool.dilation, int) else pool.dilation[0]
            if pad != 0 or dil != 1 or pool.ceil_mode or pool.return_indices:
                raise ValueError("only un-padded, un-dilated, floor-mode MaxPool2d can be fused")
            self.pool_k, self.pool_s = int(k), int(st)

    def forward(self, x):
        x = lazy.resolve(x)
        bn = self.bn
        fast = (self.training and bn.training and x.is_cuda and x.dtype == torch.float32 and x.dim() in (2, 4)
                and bn.momentum is not None and bn.track_running_stats and x.shape[0] * (x[0, 0].numel()) > 1
                and x.shape[1] % 4 == 0
                and (self.pool is None or x.dim() == 4))
        if not fast:
            h = self.pool(x) if self.pool is not None else x
            h = bn(h)
            if self.hardtanh is not None:
                h = self.hardtanh(h)
            return self.sign(h)
        lo, hi = ((float(self.hardtanh.min_val), float(self.hardtanh.max_val)) if self.hardtanh is not None
                  else (-float("inf"), float("inf")))
        out = _TrainPoolBnSignFn.apply(x, bn.weight if bn.affine else None, bn.bias if bn.affine else None, bn.running_mean,
                                       bn.running_var, float(bn.eps), float(bn.momentum), self.pool_k, self.pool_s, lo, hi)
        bn.num_batches_tracked.add_(1)
        return out


class _TrainBnActQuantFn(QtFunction):
    """BatchNorm (batch statistics) [+ residual] [-> ReLU] [-> nnDorefaQuant(k)] as one autograd node on this backend's kernels
    (ops.bn_train_stats + ops.affine_dorefa_codes / bn_eval_device forward, ops.bn_act_train_backward); see FusedTrainBnActQuant."""

    @staticmethod
    def forward(ctx, x, res, gamma, beta, running_mean, running_var, eps, momentum, relu, bits):
        # (the quantiser's int8 range flag is zeroed by the statistics' last launch instead of a torch.zeros fill)
        flag = torch.empty((1,), dtype=torch.int32, device=x.device) if bits else None
        xs, stats2 = ops.bn_train_stats(x, running_mean, running_var, eps, momentum, zero_flag=flag)
        C = int(x.shape[1])
        x2 = xs.view(-1, C)
        dev = x.device
        g_ = gamma.detach() if gamma is not None else torch.ones((C,), dtype=torch.float32, device=dev)
        b_ = beta.detach() if beta is not None else torch.zeros((C,), dtype=torch.float32, device=dev)
        rs = None
        if res is not None:
            rs = ops._rows_view(res.detach())[0]
        four = x.dim() == 4
        if bits:
            cp, y = ops.affine_dorefa_codes(x2, g_, b_, bits, relu=bool(relu), res_f32=rs.view(-1, C) if rs is not None else None,
                                            want_f32=True, ld_bytes=ops.code_ld_bytes(C, 16) if four else None, bn_stats=stats2,
                                            overflow=flag)
        else:
            y = ops.bn_eval_device(x2, g_, b_, stats2)
        ctx.saved_chain = (xs, rs, stats2)
        ctx.relu = bool(relu)
        ctx.save_for_backward(gamma, beta)
        ctx.x_nchw = four and x.is_contiguous() and not x.is_contiguous(memory_format=torch.channels_last)
        if four:
            N, _, H, W = x.shape
            out = y.view(N, H, W, C).permute(0, 3, 1, 2)
            return packed.attach_codes(out, cp, packed.NHWC) if bits else out
        out = y.view(x.shape)
        return packed.attach_codes(out, cp, packed.ROWS_LAST) if bits else out

    @staticmethod
    def backward(ctx, grad_out):
        gamma, beta = ctx.saved_tensors
        xs, rs, stats2 = ctx.saved_chain
        want_res = rs is not None and ctx.needs_input_grad[1]
        gx, dgamma, dbeta, gres = ops.bn_act_train_backward(grad_out, xs, rs, gamma, beta, stats2, ctx.relu, want_res)
        if ctx.x_nchw:
            gx = gx.contiguous()
            gres = gres.contiguous() if gres is not None else None
        return (gx, gres, dgamma if gamma is not None else None, dbeta if beta is not None else None) + (None,) * 6


class FusedTrainBnActQuant(torch.nn.Module):
    """TRAINING-mode form of BatchNorm{1,2}d [+ residual] [-> ReLU] [-> nnDorefaQuant(k)] — what the reference puts behind every
    DorefaConv2d of its ResNets (models/Resnet/Resnet_bin.py:63-97; identity STE of the quantiser,
    functions/dorefa_connect.py:28-45) — as one autograd node: batch statistics (two passes, folded in double) and the
    running-statistics update, then ONE pass that normalises, adds the shortcut, applies ReLU and the un-clamped k-bit
    quantiser and writes both the fp32 image and the int8 codes the next DoReFa layer contracts; backward = ReLU mask,
    BatchNorm backward and the shortcut's gradient in two passes (csrc/train_chain.hip) — instead of MIOpen's BatchNorm
    forward / backward, torch's add / relu / threshold_backward and the separate quantiser pass.  ``a_bits`` 0: no quantiser
    (plain BatchNorm, e.g. the shortcut branch's — then without ReLU / residual).  Shares the BatchNorm module.  CPU tensors,
    eval mode, C % 4 != 0, momentum=None or untracked statistics: the module chain itself."""

    def __init__(self, bn, a_bits: int = 0, relu: bool = False):
        super().__init__()
        from ..functions import nnDorefaQuant
        self.bn, self.a_bits, self.relu = bn, int(a_bits), bool(relu)
        self.quant = nnDorefaQuant(self.a_bits) if self.a_bits else None
        if self.a_bits and not 2 <= self.a_bits <= 8:
            raise ValueError("the fused chain writes int8 codes: 2 <= a_bits <= 8 (or 0: no quantiser)")

    def forward(self, x, residual=None):
        x, residual = lazy.resolve(x), lazy.resolve(residual)
        bn = self.bn
        fast = (self.training and bn.training and x.is_cuda and x.dtype == torch.float32 and x.dim() in (2, 4)
                and bn.momentum is not None and bn.track_running_stats and x.shape[0] * (x[0, 0].numel()) > 1
                and x.shape[1] % 4 == 0 and (self.a_bits or (residual is None and not self.relu))
                and (residual is None or (residual.shape == x.shape and residual.dtype == torch.float32 and residual.is_cuda)))
        if not fast:
            h = bn(x)
            if residual is not None:
                h = h + residual
            if self.relu:
                h = torch.relu(h)
            return self.quant(h) if self.quant is not None else h
        out = _TrainBnActQuantFn.apply(x, residual, bn.weight if bn.affine else None, bn.bias if bn.affine else None,
                                       bn.running_mean, bn.running_var, float(bn.eps), float(bn.momentum), self.relu, self.a_bits)
        bn.num_batches_tracked.add_(1)
        return out


def fuse_sequential_training(seq: torch.nn.Sequential) -> torch.nn.Sequential:
    """New nn.Sequential (sharing every module of ``seq``) where each [MaxPool2d?, BatchNorm, Hardtanh?, BinaryConnect(det)]
    run is one FusedTrainPoolBnSign: the training step then runs no torch / MIOpen pooling, BatchNorm or Hardtanh kernel
    between binarised layers, forward or backward."""
    mods = list(seq.children())
    out, i = [], 0
    while i < len(mods):
        j, pool = i, None
        if isinstance(mods[j], torch.nn.MaxPool2d):
            pool, j = mods[j], j + 1
        if j < len(mods) and isinstance(mods[j], (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            bn, j2, ht = mods[j], j + 1, None
            if j2 < len(mods) and isinstance(mods[j2], torch.nn.Hardtanh):
                ht, j2 = mods[j2], j2 + 1
            if j2 < len(mods) and _is_det_binary_connect(mods[j2]):
                try:
                    out.append(FusedTrainPoolBnSign(bn, pool, ht))
                    i = j2 + 1
                    continue
                except ValueError:
                    pass
        out.append(mods[i])
        i += 1
    return torch.nn.Sequential(*out)


def _is_det_binary_connect(m):
    return isinstance(m, _FunctionModule) and m.core is BinaryConnectDeterministic


def fuse_sequential(seq: torch.nn.Sequential, fuse_conv: bool = False, packed_pool: bool = False, fold=None) -> torch.nn.Sequential:
    """New nn.Sequential where every [MaxPool2d?, BatchNorm, Hardtanh?, BinaryConnect(det)] run is
    replaced by one FusedPoolBnSign (sharing the original BatchNorm's parameters).  With ``fuse_conv`` a
    BinConv2d / TerConv2d directly in front of such a run joins it (FusedConvPoolBnSign: the conv emits
    threshold bits, no fp32 activation is written at all).  With ``packed_pool`` a MaxPool2d that directly follows a
    fused block (pool AFTER the sign, VGG style) becomes a PackedMaxPool on the bit planes.  ``fold``: the BatchNorm
    fold of the fused blocks ("reference" | "device", module docstring; default DEFAULT_FOLD)."""
    from .binary_layers import BinConv2d
    from .terner_layers import TerConv2d
    from .xnor_layers import XNORConv2d
    mods = list(seq.children())
    out, i = [], 0
    while i < len(mods):
        j = i
        conv = None
        if fuse_conv and isinstance(mods[j], (BinConv2d, TerConv2d, XNORConv2d)) and j + 1 < len(mods):
            conv, j = mods[j], j + 1
        pool = None
        if isinstance(mods[j], torch.nn.MaxPool2d):
            pool, j = mods[j], j + 1
        pre_relu = False
        if conv is None and j + 1 < len(mods) and isinstance(mods[j], torch.nn.ReLU) and \
                isinstance(mods[j + 1], (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            pre_relu, j = True, j + 1           # Linear -> ReLU -> BatchNorm -> BinaryConnect (MLPBin.py:42-44)
        if j < len(mods) and isinstance(mods[j], (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            bn, j2 = mods[j], j + 1
            if j2 < len(mods) and isinstance(mods[j2], torch.nn.Hardtanh) and mods[j2].min_val < 0 < mods[j2].max_val:
                j2 += 1
            if j2 < len(mods) and _is_det_binary_connect(mods[j2]):
                try:
                    out.append(FusedConvPoolBnSign(conv, bn, pool, fold=fold) if conv is not None
                               else FusedPoolBnSign(bn, pool, pre_relu=pre_relu, fold=fold))
                    i = j2 + 1
                    if packed_pool and i < len(mods) and isinstance(mods[i], torch.nn.MaxPool2d) and (not bn.affine or bn.weight.dim() == 1) \
                            and isinstance(bn, torch.nn.BatchNorm2d):
                        try:
                            out.append(PackedMaxPool(mods[i]))
                            i += 1
                        except ValueError:
                            pass
                    continue
                except ValueError:
                    if conv is not None:     # the conv cannot join: keep it, fuse the rest on the next turn
                        out.append(conv)
                        i += 1
                        continue
        out.append(mods[i])
        i += 1
    link_nib_planes(out)
    return torch.nn.Sequential(*out)


def link_nib_planes(mods) -> None:
    """Where a fused conv block directly consumes the output of another fused block (or of a PackedMaxPool), let the
    producer write the consumer's operand — the fp4 nibble pixel plane with the consumer's padding as a zero border —
    instead of bit planes (saves the consumer's qt_bits_to_nib_pad pass: 10 % of the fused VGG-16 forward)."""
    from ..functions import _fused
    for a, b in zip(mods, mods[1:]):
        if isinstance(a, (FusedConvPoolBnSign, PackedMaxPool)):
            a.out_nib_halo = None
            if (_fused._cfg("PAD_PLANES") and isinstance(b, FusedConvPoolBnSign) and not getattr(a, "flatten_hwc", False)):
                a.out_nib_halo = tuple(int(v) for v in ops._pairs(b.conv.padding))


class FusedFeatureClassifier(torch.nn.Module):
    """Inference form of the reference's CNN layout (models/Alexnet/Alexnet_Bin.py:12-54, and VGG-style stacks):

        features   = nn.Sequential(...)                       # binarised conv blocks
        x          = features(x).view(N, C*H*W)               # NCHW flattening
        classifier = nn.Sequential([BinaryConnect,] LinearBin | LinearTer, ...)

    Every [conv, pool?, BatchNorm, Hardtanh?, BinaryConnect] run becomes one FusedConvPoolBnSign (threshold bits, no
    fp32 activation), a MaxPool placed after the sign runs on the bit planes, a BinaryConnect that opens the
    classifier is pulled into the last feature block, the planes are flattened in (h, w, c) order and the first
    classifier layer gets its weight columns permuted once to that order.  Shares every other parameter with the
    modules it was built from (which must be in eval mode).  ``feat_chw`` = (C, H, W) of the feature map the
    classifier was trained on."""

    def __init__(self, features: torch.nn.Sequential, classifier: torch.nn.Sequential, feat_chw, fuse_conv: bool = True, fold=None):
        super().__init__()
        from .binary_layers import LinearBin
        from .terner_layers import LinearTer
        from .xnor_layers import LinearXNOR
        if any(m.training for m in (features, classifier)):
            raise ValueError("fuse eval-mode modules")
        f, c = list(features.children()), list(classifier.children())
        if c and _is_det_binary_connect(c[0]):
            f, c = f + [c[0]], c[1:]
        if not c or not isinstance(c[0], (LinearBin, LinearTer, LinearXNOR)):
            raise ValueError("classifier must start with [BinaryConnect(deterministic),] LinearBin / LinearTer / LinearXNOR")
        self.features = fuse_sequential(torch.nn.Sequential(*f), fuse_conv=fuse_conv, packed_pool=True, fold=fold)
        tail = list(self.features.children())[-1] if len(self.features) else None
        if not isinstance(tail, (FusedConvPoolBnSign, FusedPoolBnSign, PackedMaxPool)):
            raise ValueError("features must end with BatchNorm2d [Hardtanh] (+ the classifier's BinaryConnect), or with "
                             "BatchNorm2d [Hardtanh] BinaryConnect [MaxPool2d]: the last block has to produce sign bits")
        C, H, W = (int(v) for v in feat_chw)
        src = c[0]
        if src.in_features != C * H * W:
            raise ValueError(f"classifier expects {src.in_features} features, feat_chw gives {C * H * W}")
        if isinstance(src, LinearXNOR):
            # a real-valued sum depends on its order: the XNOR layer keeps its own weight and reads the (h, w, c)-flattened bits in
            # NCHW order instead (PackedActivation.hwc -> qt_bits_alpha_pairs_f16x2), bit-identical to the module graph
            fc1 = src
        else:
            fc1 = type(src)(src.in_features, src.out_features, bias=src.bias is not None).to(src.weight.device)
            if src.bias is not None:
                fc1.bias.data.copy_(src.bias.data)
            fc1.eval()
            fc1.weight.data.copy_(permute_fc_weight_hwc(src.weight.data, C, H, W))      # already the quantised image
        self.classifier = fuse_sequential(torch.nn.Sequential(fc1, *c[1:]), fold=fold)
        self.eval()

    def train(self, mode: bool = True):
        # an inference form: its first classifier layer is a permuted COPY of an eval-mode (already quantised) weight,
        # so there is no real-valued weight to go back to
        if mode:
            raise RuntimeError("FusedFeatureClassifier is an inference module: train the modules it was built from")
        return super().train(False)

    @property
    def last(self):
        return list(self.features.children())[-1]

    def forward(self, x):
        return self.classifier(self.features(x).flatten_hwc())


def permute_fc_weight_hwc(weight: torch.Tensor, C: int, H: int, W: int) -> torch.Tensor:
    """Columns of an FC weight that expects the NCHW flattening (c*H*W + h*W + w) re-ordered to the
    (h, w, c) order of PackedActivation.flatten_hwc()."""
    N = weight.shape[0]
    return weight.view(N, C, H, W).permute(0, 2, 3, 1).reshape(N, H * W * C).contiguous()
