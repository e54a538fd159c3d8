/*
 * qt_hip.h — C-ABI of libqt_hip.so, the MI355X (gfx950) backend for the QuantTorch
 * quantised-operator hot path:  sign / ternarize / k-bit quantize  ->  bit-pack  ->
 * XNOR-popcount GEMM (BinaryNet / XNOR-Net), two-plane ternary GEMM, packed low-bit GEMM.
 *
 * This header is the drop-in boundary.  Everything above it (the autograd.Function /
 * nn.Module mirror of QuantTorch.functions / QuantTorch.layers) is host plumbing.
 *
 * Conventions (all entry points):
 *   - plain `extern "C"`, raw DEVICE pointers + explicit sizes/strides, no torch types;
 *   - stateless and re-entrant: no allocation, no ownership transfer, no process-global state of any kind (kernel
 *     variants for tests / tuning are ARGUMENTS of the *_variant entry points); the only thing that survives a call is
 *     what it wrote into caller-provided buffers;
 *   - work is enqueued on `stream` (a hipStream_t, passed as an opaque pointer; NULL = the
 *     default stream) and the call returns without synchronising;
 *   - the caller has already made the right device current (hipSetDevice);
 *   - return value: QT_OK (0) or a negative qt_status; qt_strerror() names it.
 *   - "ld*" arguments are leading dimensions IN ELEMENTS of that buffer's type.
 *
 * Packed formats (defined by this library, see DESIGN.md "Data layout in HBM"):
 *   bit plane  : uint32 words along K, bit j of word w <-> element k = 32*w + j.
 *                sign plane bit = 1  <=>  value < 0   (so +0.0, -0.0, NaN -> bit 0 -> +1,
 *                exactly the reference's safeSign, QuantTorch/functions/common.py:4-7).
 *                mask plane bit = 1  <=>  ternary value != 0.
 *                Row stride ("ldp", in words) must be a multiple of 4 (16-byte rows);
 *                every bit/word between K and 32*ldp is ZERO in every plane.
 *   nib plane  : FP4-E2M1 nibbles (two per byte, element 2i in the low nibble), value set
 *                {+1 = 0x2, -1 = 0xA, 0 = 0x0}; 8 elements per uint32 word, element k of a row in
 *                nibble (k & 7) of word (k >> 3).  Row stride in words must be a multiple of 4
 *                (16-byte rows); every nibble between K and 8*ldp is ZERO.  This is the MFMA
 *                operand format (v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales is
 *                exact for +-1 / 0 operands and K < 2^24).
 *
 * Reference interfaces replaced (paths relative to the reference repo root):
 *   the reference has NO native code; the functions below replace the ATen calls made from
 *   QuantTorch/functions and QuantTorch/layers (all .py files) that are cited per entry point.
 */
#ifndef QT_HIP_H
#define QT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* qt_stream_t; /* hipStream_t */

typedef enum qt_status {
    QT_OK = 0,
    QT_ERR_INVALID_ARG = -1,  /* null pointer, negative size, inconsistent shape            */
    QT_ERR_ALIGNMENT = -2,    /* pointer / leading dimension violates the packed-format rule */
    QT_ERR_LAUNCH = -3,       /* hipLaunchKernel / hipGetLastError reported a failure        */
    QT_ERR_UNSUPPORTED = -4,  /* argument combination not implemented                        */
    QT_ERR_NO_DEVICE = -5     /* no gfx950 device / wrong architecture                        */
} qt_status;

/* Library identification. qt_version() = major*10000 + minor*100 + patch. */
int qt_version(void);
const char* qt_strerror(int status);
/* Name of the gfx arch the code objects were built for ("gfx950"). */
const char* qt_target_arch(void);
/* Fills name[cap] with the current device's gcnArchName, returns CU count (<0 on error). */
int qt_device_info(char* name, int cap);

/* ------------------------------------------------------------------------------------------
 * Elementwise quantisers (fp32 -> fp32), forward and STE backward.
 * ---------------------------------------------------------------------------------------- */

/* y[i] = x[i] < 0 ? -1 : +1.            safeSign, QuantTorch/functions/common.py:4-7;
 * forward of BinaryConnectDeterministic, functions/binary_connect.py:22-28. */
int qt_binarize_f32(const float* x, float* y, int64_t n, qt_stream_t stream);

/* y[i] = z[i] < (clamp(x[i],-1,1)+1)/2 ? +1 : -1 with caller-supplied uniforms z in [0,1).
 * forward of BinaryConnectStochastic, functions/binary_connect.py:52-61 (the RNG draw
 * `torch.rand_like` stays on the torch side so the stream of randoms is torch's). */
int qt_binarize_stochastic_f32(const float* x, const float* z, float* y, int64_t n,
                               qt_stream_t stream);

/* y[i] = x >= 0.5 ? +1 : (x < -0.5 ? -1 : 0)   (NaN -> +1, as the reference's double safeSign)
 * forward of TernaryConnectDeterministic, functions/terner_connect.py:24-27. */
int qt_ternarize_f32(const float* x, float* y, int64_t n, qt_stream_t stream);

/* y[i] = s - s*(z[i] > |x[i]|),  s = safeSign(x[i]).
 * forward of TernaryConnectStochastic, functions/terner_connect.py:52-56. */
int qt_ternarize_stochastic_f32(const float* x, const float* z, float* y, int64_t n,
                                qt_stream_t stream);

/* gin[i] = |x[i]| > thr ? 0 : gout[i]     (thr = 1.001f in every reference caller)
 * backward of Binary/TernaryConnect{Deterministic,Stochastic},
 * functions/binary_connect.py:31-38,64-71; functions/terner_connect.py:29-34,58-63. */
int qt_ste_mask_f32(const float* gout, const float* x, float* gin, int64_t n, float thr,
                    qt_stream_t stream);
/* out[i] = x[i] (x NULL: 0), or NaN for every i when (*flag & mask) != 0: folds a DEVICE range flag (codes beyond int8 / beyond the
 * fp16 plane, a remembered +-1 verdict that no longer holds) into a bias or a gradient without a host synchronisation — a broken
 * assumption yields NaN, never a plausible wrong number. */
int qt_poison_f32(const float* x, const int32_t* flag, int32_t mask, float* out, int64_t n, qt_stream_t stream);
/* The base-256 digits of a k-bit DoReFa image whose un-clamped codes left int8 (functions/dorefa_connect.py:11-25): q = rint(x *
 * levels), hi = floor(q / 256), lo = q - 256 hi over n values in storage order; *flag |= bit when |hi| >= 256 (a digit no longer
 * exact in bf16).  And the recombination of the two weight-gradient passes: out = (g_hi * 256 + g_lo) * inv, NaN when *flag & mask
 * (flag may be NULL).  Round 6: one launch each instead of 6 + 5 + 4 torch launches per strided shortcut conv and step. */
int qt_code_digits_f32(const float* x, int64_t n, float levels, float* hi, float* lo, int32_t* flag, int32_t bit, qt_stream_t stream);
int qt_digit_combine_f32(const float* g_hi, const float* g_lo, const int32_t* flag, int32_t mask, float inv, float* out, int64_t n,
                         qt_stream_t stream);

/* DoReFa k-bit quantiser: k==1 -> safeSign; k==32 -> copy; else
 * y = fl(fl(1/n) * rint(n*x)), n = 2^k - 1, round-half-even, NO clamp.
 * _quantize, functions/dorefa_connect.py:11-25. */
int qt_dorefa_quantize_f32(const float* x, float* y, int64_t n, int bit_width,
                           qt_stream_t stream);

/* Lin / Log fixed-point quantisers of "CNNs using Logarithmic Data Representation" (functions/log_lin_connect.py).
 * Lin (:61-67): step = 2^(fsr - bit_width); mode 0: clamp(round(x/step)*step, 0, 2^fsr); mode 1 (with_sign):
 * sign(x) * the same of |x|; mode 2: the quantised-gradient backward (:79) sign(g) * clamp(round(g/step)*step, 0,
 * 2^fsr) (negative g -> -0, as upstream); bit_width 32 = identity.  round = half to even, sign(0) = 0, NaN kept.
 * Log (:31-33): [sign(x) *] 2^clamp(round(log2|x|), fsr - 2^bit_width, fsr); x = 0 -> 0 (signed) / 2^(fsr-2^bits).
 * log2 is the device's fp32 log2f: inputs within an ulp of 2^(k+1/2) may round to the other neighbour than on
 * the host (measure-zero boundary). */
int qt_lin_quantize_f32(const float* x, float* y, int64_t n, int fsr, int bit_width, int mode, qt_stream_t stream);
int qt_log_quantize_f32(const float* x, float* y, int64_t n, int fsr, int bit_width, int with_sign,
                        qt_stream_t stream);

/* AP2 of the shift-based batch norm (functions/binary_connect.py:157-169): safeSign(x) * 2^round(log2|x|). */
int qt_ap2_f32(const float* x, float* y, int64_t n, qt_stream_t stream);

/* Shift-based batch-norm primitive, ShiftBatch.forward (functions/binary_connect.py:173-186; layers ShiftNormBatch1d / 2d,
 * layers/binary_layers.py:110-160): y = ((x - mean) * AP2(1 / sqrt(var + eps))) * AP2(weight) + bias over x[N, E] with E-entry
 * statistics / affine vectors broadcast over N (each product / sum rounded separately, as the torch expression).
 * norm (optional, [N, ldn]) = (x - mean) * AP2(1 / sqrt(var + eps)) and sqrtvar (optional, [E]) are what the reference saves
 * for its backward (:188-204). */
int qt_shift_batch_f32(const float* x, int64_t ldx, const float* running_mean, const float* running_var, const float* weight,
                       const float* bias, float eps, float* y, int64_t ldy, float* norm, int64_t ldn, float* sqrtvar, int64_t N,
                       int64_t E, qt_stream_t stream);

/* XNOR-Net weight quantiser over a row-major [R, C] view of the weight:
 * alpha[c] = mean_r |w[r,c]| ;  wq[r,c] = sign(w[r,c]) * alpha[c]  (torch.sign: 0 -> 0).  wq may be NULL
 * (alpha only; with a separate wq buffer tall matrices use it as scratch for chunked partial sums).  XNORDense: R = N, C = K
 * (functions/xnor_connect.py:112-113, global DIM = 0);
 * XNORConv2d(dim=[0,1]): R = Cout*Cin, C = kh*kw (functions/xnor_connect.py:140-141). */
int qt_xnor_weight_f32(const float* w, int64_t ldw, float* alpha, float* wq, int64_t ldq, int64_t R,
                       int64_t C, qt_stream_t stream);

/* LinearXNOR on a packed +-1 activation (layers/xnor_layers.py:8-33 in eval mode behind the fused inference chain): the operand
 * x[b, k] * alpha[k] of y = (x * alpha) . sign(W)^T (functions/xnor_connect.py:112-115) as two-term fp16 pairs, from the
 * activation's SIGN BITS and the pair image of alpha / s (alpha_pairs[k] = hi | lo << 16; qt_f16x2_pack_f32 of the [1, K] scale
 * row): out[r][k] = bit ? -pair[k] : pair[k] (exact).  out: [rows][ld_bytes], ld_bytes % 128 == 0, pad pairs zero.
 * perm_C > 0 (perm_C * perm_HW == K): the bit rows hold a feature map flattened in (h, w, c) order (what the fused conv blocks hand
 * over) while k counts in the NCHW order c * HW + hw of the module graph's reshape: feature k reads bit (k % HW) * C + k / HW, so
 * the GEMM sums in the module graph's order (bit-identical results for the two executions of a model). */
int qt_bits_alpha_pairs_f16x2(const uint32_t* bits, int64_t ldb, const uint32_t* alpha_pairs, uint32_t* out, int64_t ld_bytes,
                              int64_t rows, int64_t K, int64_t perm_C, int64_t perm_HW, qt_stream_t stream);

/* LinearXNOR on a packed +-1 activation, INTEGER form (replaces the fp16 pair operand above wherever alpha is finite): alpha in
 * fixed point, A[k] = rint(alpha[k] / s) < 2^21 as three 7-bit digits (digit_table[k] = d0 | d1 << 8 | d2 << 16, A = d0 2^14 +
 * d1 2^7 + d2), out = three stacked int8 planes [3 * rows][ld_bytes] with out[j * rows + r][k] = +-d_j[k] by the activation's
 * sign bit (ld_bytes % 32 == 0, pad zero; perm_C / perm_HW as above).  The three planes meet sign(W)'s int8 codes in ONE
 * qt_i8_gemm / qt_i8_gemm_splitk launch (exact integer partial sums) and qt_digit_reduce_f32 forms
 *     Y[b][n] = fp32(s * (2^14 S0 + 2^7 S1 + S2)) + bias[n],   S_j = sum_z partial[z * slice_stride + (j * rows + b) * ldp + n]
 * in fp64 (exact; one rounding before the bias).  functions/xnor_connect.py:112-115. */
int qt_bits_alpha_digits_i8(const uint32_t* bits, int64_t ldb, const uint32_t* digit_table, int8_t* out, int64_t ld_bytes,
                            int64_t rows, int64_t K, int64_t perm_C, int64_t perm_HW, qt_stream_t stream);
int qt_digit_reduce_f32(const float* partial, int64_t ldp, int64_t slice_stride, int64_t nslice, const float* scale_dev,
                        const float* bias, float* Y, int64_t ldy, int64_t rows, int64_t N, qt_stream_t stream);

/* The digit route for small heads (N of a few outputs: the classifier layer) in ONE launch, straight from the sign bits:
 * Y[b][n] = fp32(s * sum_k x[b, k] A[k] W[n][k]) + bias[n] with the exact integer sum (A from digit_table, W = int8 codes of
 * sign(W), [N][ldw_bytes], zero past K) — bit-identical to qt_bits_alpha_digits_i8 + qt_i8_gemm_splitk + qt_digit_reduce_f32.
 * K < 2^16. */
int qt_xnor_head_i8(const uint32_t* bits, int64_t ldb, const uint32_t* digit_table, const int8_t* wcodes, int64_t ldw_bytes,
                    const float* scale_dev, const float* bias, float* Y, int64_t ldy, int64_t rows, int64_t N, int64_t K,
                    int64_t perm_C, int64_t perm_HW, qt_stream_t stream);

/* XNOR-Net ACTIVATION quantiser on a row-major [R, C] tensor (_quantOpXnor / nnQuantXnor / QuantXnor,
 * functions/xnor_connect.py:17-66):  y = sign(x) * mean(x, dim)  with torch.sign (0 -> 0) and the SIGNED mean the
 * reference computes (:21-28).  dim = 1: mean[R] per row; dim = 0: mean[C] per column; dim = -1: mean[1] over all
 * elements (needs `work`, qt_xnor_act_work_floats() floats; NULL otherwise).  `mean` is an output (saved for backward).
 * Backward (:30-37): gin = sign(x) * mean(g * sign(x), dim, keepdim) + g * mean;  gmean = scratch of mean's size. */
int64_t qt_xnor_act_work_floats(void);
int qt_xnor_act_f32(const float* x, int64_t ldx, float* mean, float* work, float* y, int64_t ldy, int64_t R, int64_t C,
                    int dim, qt_stream_t stream);
int qt_xnor_act_backward_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* mean, float* gmean,
                             float* work, float* gin, int64_t ldi, int64_t R, int64_t C, int dim, qt_stream_t stream);

/* Diagnostic: launches of the persistent direct 3 x 3 code conv (csrc/code_conv3x3.hip) since the library was loaded.
 * qt_conv2d_implicit_codes takes that kernel for 3 x 3 / stride 1 / padding 1 layers with 64 or 128 input channels, Cout % 64 == 0
 * and power-of-two maps (models/Resnet/Resnet_bin.py:63-97 stages 1 and 2), bit-identical to its implicit-GEMM route; tests use the
 * counter to assert which route ran.  Setting the environment variable QT_NO_CODE_CONV3X3 (read per call) keeps every conv on the
 * implicit-GEMM kernel. */
int64_t qt_code_conv3x3_launch_count(void);

/* Input quantiser of XNORConv2d(quant_input = True) (functions/xnor_connect.py:142-143):
 *   y[n, c, h, w] = sign(x[n, c, h, w]) * mean_c |x[n, :, h, w]|        torch.sign (0 -> 0, NaN -> NaN), one scale per pixel.
 * x: logical [N, C, H, W] fp32 with ELEMENT strides (sn, sc, sh, sw) (contiguous NCHW or channels-last); y: the same logical
 * tensor written densely in NHWC memory order [N][H][W][C] — the layout the per-tap scaled conv's operand pack and the
 * weight-gradient routes read (backward sees this quantised tensor, :144).
 * a_plane (optional, like y): the scale plane A[N][H][W] itself — the per-(row, tap) factor of qt_conv2d_implicit_taps_rows. */
int qt_xnor_input_quant_f32(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float* y, float* a_plane, int64_t N,
                            int64_t C, int64_t H, int64_t W, qt_stream_t stream);

/* XNORConv2d(quant_input = True) at the fp4 rate (functions/xnor_connect.py:142-145):
 *   y[m, co] = sum_t alpha_t * A[pixel(m, t)] * (integer dot of sign(x) and sign(W) over tap t's channels)  (+ bias)
 * P: fp4 nibble pixel plane of torch.sign(x) (+1 = 0x2, -1 = 0xA, 0 = 0x0; qt_sign0_pack_nib_f32), UN-padded [Nimg][H][W][Cw words],
 * Cw % 8 == 0; Wmat: nibble plane of sign(W), tap-major; tap_rho: the forward table of qt_xnor_tap_prep_f32; a_plane: A[Nimg][H][W]
 * (qt_xnor_input_quant_f32).  The Horner factor of a tap boundary is formed per output row from a_plane inside the kernel
 * (ElemFp4TapsRows, csrc/mfma_gemm_kernel.h).  kh * kw <= 48 (the per-row table lives in LDS); QT_ERR_UNSUPPORTED beyond. */
int qt_conv2d_implicit_taps_rows(const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw, int64_t kh, int64_t kw, int64_t sh,
                                 int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldwp,
                                 const float* bias, const float* tap_rho, const float* a_plane, float* Y, int64_t ldy, int64_t Cout,
                                 qt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Bit-pack kernels (fp32 -> packed planes).  rows x K fp32 (row stride ldx) ->
 * rows x ldp uint32 (only the first ceil(K/32) words of a row carry data, the rest are 0).
 * ---------------------------------------------------------------------------------------- */

/* sign plane of safeSign(x): bit = (x < 0).  If y_f32 != NULL the +-1 fp32 image is written too
 * (row stride ldy) — one pass for BinaryConnectDeterministic.forward + the pack that feeds the
 * next layer (functions/binary_connect.py:22-28 followed by layers/binary_layers.py:44). */
int qt_sign_pack_f32(const float* x, int64_t ldx, uint32_t* sign_plane, int64_t ldp,
                     float* y_f32, int64_t ldy, int64_t rows, int64_t K, qt_stream_t stream);

/* ternary planes of TernaryConnectDeterministic(x): mask = (x>=0.5 || x<-0.5 || isnan),
 * sign = (x < -0.5).  functions/terner_connect.py:24-27 + layers/terner_layers.py:49. */
int qt_ternary_pack_f32(const float* x, int64_t ldx, uint32_t* mask_plane, uint32_t* sign_plane,
                        int64_t ldp, int64_t rows, int64_t K, qt_stream_t stream);

/* Device-side check used when a layer receives an un-tagged activation tensor:
 * *flag (int32, device) is OR-ed with 1 if any element of x is not exactly +1.0f or -1.0f.
 * The caller zeroes *flag beforehand. */
int qt_check_pm1_f32(const float* x, int64_t n, int32_t* flag, qt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Packed GEMMs.  Y[M,N] (fp32, row stride ldy) = dot over K of the +-1 / {-1,0,+1} values the
 * planes encode, + bias[n] (bias may be NULL).  Replaces torch.nn.functional.linear on
 * quantised operands: layers/binary_layers.py:44,46; layers/terner_layers.py:49,51;
 * functions/binary_connect.py:93-98 (BinaryDense); functions/terner_connect.py:92-94.
 * Integer part is exact (int32 accumulate), converted to fp32 once, bias added once.
 * ---------------------------------------------------------------------------------------- */

/* XNOR-popcount GEMM (VALU path: v_xor_b32 + v_bcnt_u32_b32, LDS-staged tiles).
 * Xs: M x ldxp sign plane of the activations, Ws: N x ldwp sign plane of the weights.
 * y = K - 2*popcount(x ^ w). */
int qt_xnor_gemm(const uint32_t* Xs, int64_t ldxp, const uint32_t* Ws, int64_t ldwp,
                 const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                 qt_stream_t stream);

/* The same two GEMMs with the kernel chosen by the caller (tests / tuning; no process state involved):
 * variant 0 = automatic (streaming kernel — K along the lanes, DPP wavefront reduction — when min(M, N) <= 32; skinny
 * lane-per-batch-row kernel for other small M or N; 128x128-tile kernel otherwise),
 * 1 = tiled, 2 = skinny, 3 = streaming (QT_ERR_UNSUPPORTED unless min(M, N) <= 32). */
int qt_xnor_gemm_variant(int variant, const uint32_t* Xs, int64_t ldxp, const uint32_t* Ws, int64_t ldwp,
                         const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, qt_stream_t stream);
int qt_tern_gemm_variant(int variant, const uint32_t* Xs, int64_t ldxp, const uint32_t* Wmask, const uint32_t* Wsign,
                         int64_t ldwp, const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                         qt_stream_t stream);

/* Binary activations x ternary weights (two planes): y = popc(m) - 2*popc((x ^ s) & m). */
int qt_tern_gemm(const uint32_t* Xs, int64_t ldxp, const uint32_t* Wmask, const uint32_t* Wsign,
                 int64_t ldwp, const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N,
                 int64_t K, qt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Matrix-core formulation of the same contractions (nibble planes, MX-fp4 MFMA).  Bit-identical
 * results to qt_xnor_gemm / qt_tern_gemm; used for large M*N*K where the popcount path is
 * VALU-bound (DESIGN.md "Kernels").
 * ---------------------------------------------------------------------------------------- */

/* nibble plane of safeSign(x): +1 -> 0x2, -1 -> 0xA (functions/common.py:4-7). */
int qt_sign_pack_nib_f32(const float* x, int64_t ldx, uint32_t* nib_plane, int64_t ldp,
                         int64_t rows, int64_t K, qt_stream_t stream);

/* nibble plane of TernaryConnectDeterministic(x): 0 -> 0x0 (functions/terner_connect.py:24-27). */
int qt_ternary_pack_nib_f32(const float* x, int64_t ldx, uint32_t* nib_plane, int64_t ldp,
                            int64_t rows, int64_t K, qt_stream_t stream);

/* nibble plane of torch.sign(x): 0 (and NaN) -> 0x0 — the sign image of the XNOR-Net weight quantiser, which keeps W == 0 at 0
 * (functions/xnor_connect.py:112-113, 140-141). */
int qt_sign0_pack_nib_f32(const float* x, int64_t ldx, uint32_t* nib_plane, int64_t ldp, int64_t rows, int64_t K,
                          qt_stream_t stream);

/* Both operands of one LinearBin / LinearTer forward (layers/binary_layers.py:44: F.linear(x, bin_op(W), b) on a
 * +-1 activation that is not packed yet) in ONE launch: x -> safeSign nibble plane, w -> safeSign (w_ternary = 0)
 * or ternary (1) nibble plane.  Same formats and contracts as the two single-operand entries. */
int qt_pack_pair_nib_f32(const float* x, int64_t ldx, uint32_t* x_plane, int64_t ldxp, int64_t rows_x,
                         const float* w, int64_t ldw, uint32_t* w_plane, int64_t ldwp, int64_t rows_w,
                         int64_t K, int w_ternary, qt_stream_t stream);

/* Name of the kernel configuration qt_nib_gemm's automatic dispatch launches for (M, N, K) with these row strides — written to
 * out[cap] (bench.py quotes it in roofline.kernel, so the record names the kernel that ran). */
int qt_nib_gemm_describe(int64_t M, int64_t N, int64_t K, int64_t ldxp, int64_t ldwp, char* out, int cap);

/* Y[M,N] = Xn . Wn^T (+ bias): replaces the same F.linear call sites as qt_xnor_gemm /
 * qt_tern_gemm (binary and ternary weights share this entry point: zero is a nibble value). */
int qt_nib_gemm(const uint32_t* Xn, int64_t ldxp, const uint32_t* Wn, int64_t ldwp,
                const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                qt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DoReFa k-bit path: int8 code planes + int8 MFMA (v_mfma_i32_32x32x32_i8), exact int32 accumulate.
 * Replaces F.linear / F.conv2d on DoReFa-quantised operands (layers/dorefa_layers.py:43,45,79,81)
 * when the activation is a k-bit code image (output of nnDorefaQuant, functions/dorefa_connect.py:28-45)
 * and the weight is 1-bit (sign(W) * E, functions/dorefa_connect.py:99-102).
 * code plane: int8 per element, row stride in uint32 words % 4 (16-byte rows), pad bytes zero.
 * ---------------------------------------------------------------------------------------- */

/* Activation codes: q = rint((2^k - 1) * x) as int8, optionally also the fp32 image
 * y = fl(fl(1/(2^k-1)) * q) that nnDorefaQuant returns (y_f32 may be NULL).  *overflow (int32,
 * device, caller-zeroed) is OR-ed with 1 if any |q| > 127 (the reference does NOT clamp; the caller
 * must then use the dense path). 2 <= bit_width <= 8. */
int qt_dorefa_codes_i8(const float* x, int64_t ldx, int8_t* codes, int64_t ldc_bytes, float* y_f32,
                       int64_t ldy, int64_t rows, int64_t K, int bit_width, int32_t* overflow,
                       qt_stream_t stream);

/* Inference fusion of the DoReFa activation chain between two quantised layers (k-bit form; the
 * module sequence conv -> BatchNorm2d(eval) [-> + shortcut] -> ReLU -> nnDorefaQuant(k) of
 * models/samples/ResNet_Dorefa.py:26,35 collapsed into one pass over the conv output):
 *   t = fl(fl(x*alpha[c]) + beta[c])                       eval BatchNorm folded to per-channel (alpha, beta)
 *   t += fl(fl(r*res_alpha[c]) + res_beta[c])              res_f32 != NULL (res_alpha/res_beta NULL: t += r)
 *   t += fl(res_scale * code)                              res_codes != NULL (identity shortcut held as codes)
 *   t = max(t, 0) if relu == 1 ; q = rint((2^k-1) * t)     functions/dorefa_connect.py:24-25, unclamped
 * relu == 2: ReLU BEFORE the BatchNorm instead (x = max(x, 0) first: the Linear -> ReLU -> BatchNorm -> quant order of
 * models/FullNet/DorefaMNIST.py:46-48), no ReLU after.
 * codes <- q as int8 (pad bytes of the 16-byte rows zero), y_f32 (may be NULL) <- fl(fl(1/(2^k-1)) * q).
 * *overflow is OR-ed with 1 if any |q| > 127 or NaN (code written as 0), as qt_dorefa_codes_i8.
 * bn_stats != NULL (round 3, "device" BatchNorm arithmetic): bn_stats = [mean[C] | rs[C]], alpha = the BatchNorm weight,
 * beta = its bias, and  t = fma(fl(fl(x - mean[c]) * rs[c]), weight[c], bias[c])  — bit for bit what eval-mode
 * F.batch_norm evaluates on this device when rs is read back from its kernel (tools/probes/bn_eval_emulation.py); the un-modified
 * module graph then equals its module-by-module execution exactly.  res_bn_stats: the same for the fp32 residual's BatchNorm. */
/* Eval-mode BatchNorm of an fp32 [rows][C] matrix in the device's arithmetic (bn_stats = [mean | rs], see above):
 * y = fma(fl(fl(x - mean[c]) * rs[c]), weight[c], bias[c]) — replaces F.batch_norm on the conv -> BatchNorm shortcut branch of
 * models/samples/ResNet_Dorefa.py when the block runs fused.  C % 4 == 0, 16-byte aligned rows. */
int qt_bn_eval_device_f32(const float* x, int64_t ldx, const float* weight, const float* bias, const float* bn_stats, float* y,
                          int64_t ldy, int64_t rows, int64_t C, qt_stream_t stream);

int qt_affine_dorefa_codes_i8(const float* x, int64_t ldx, const float* alpha, const float* beta,
                              const float* res_f32, int64_t ldr, const float* res_alpha, const float* res_beta,
                              const int8_t* res_codes, int64_t ldrc_bytes, float res_scale, int relu, int8_t* codes,
                              int64_t ldc_bytes, float* y_f32, int64_t ldy, int64_t rows, int64_t C, int bit_width,
                              int32_t* overflow, const float* bn_stats, const float* res_bn_stats, qt_stream_t stream);

/* qt_affine_dorefa_codes_i8 over the N*H*W pixel rows of a conv output, written INTO a halo plane
 * [N][H + 2*out_halo_h][W + 2*out_halo_w][ldc_bytes] whose border the same launch zeroes: the padding of the conv that consumes
 * the activation is then physical (replaces the pass + qt_pad_pixel_plane pair behind the fp32 stem conv of
 * models/samples/ResNet_Dorefa.py; no fp32 image output). */
int qt_affine_dorefa_codes_halo_i8(const float* x, int64_t ldx, const float* alpha, const float* beta,
                                   const float* res_f32, int64_t ldr, const float* res_alpha, const float* res_beta,
                                   const int8_t* res_codes, int64_t ldrc_bytes, float res_scale, int relu, int8_t* codes,
                                   int64_t ldc_bytes, int64_t N, int64_t H, int64_t W, int64_t C, int bit_width,
                                   int32_t* overflow, const float* bn_stats, const float* res_bn_stats, int64_t out_halo_h,
                                   int64_t out_halo_w, qt_stream_t stream);

/* The fp32 image of a code plane, y = fl(inv_n * code) — what nnDorefaQuant (functions/dorefa_connect.py:24-25) hands to the
 * first NON-quantised consumer of a fused DoReFa chain (the avg_pool2d + Linear head of models/samples/ResNet_Dorefa.py).
 * in [N][H + 2*halo_h][W + 2*halo_w][ld_bytes] -> y [N*Ho*Wo][ldy] fp32 (NHWC rows), C channels per pixel.
 * *overflow != 0 (may be NULL): the chain left int8 somewhere, every output is NaN (decided on the device, no host sync).
 * pool_k > 1: avg_pool2d(pool_k) (kernel = stride, no padding, floor mode) applied to the image in the same pass, Ho = H / pool_k:
 * window values added in (row, column) order in fp32, then divided by pool_k^2 — bit-identical to ATen's kernel on this device. */
int qt_codes_to_f32(const int8_t* codes, int64_t ld_bytes, int64_t N, int64_t H, int64_t W, int64_t halo_h, int64_t halo_w,
                    int64_t C, float inv_n, const int32_t* overflow, int64_t pool_k, float* y, int64_t ldy, qt_stream_t stream);

/* MaxPool2d(pool_k, pool_s) (no padding, floor mode) on an NHWC int8 DoReFa code plane: the reference pools after the
 * quantiser (models/samples/AlexNet_Dorefa.py:38-41) and fl(inv_n * code) is monotone in the code, so the max over
 * the codes is bit-identical to pooling the fp32 image and re-deriving the codes.
 * in [N][H][W][ld_bytes] -> out [N][Ho + 2*out_halo_h][Wo + 2*out_halo_w][ld_bytes], border pixels written as zeros.
 * ld_bytes % 16 == 0. */
int qt_pool_codes_i8(const int8_t* in_plane, int64_t N, int64_t H, int64_t W, int64_t ld_bytes, int64_t pool_k,
                     int64_t pool_s, int8_t* out_plane, int64_t out_halo_h, int64_t out_halo_w, qt_stream_t stream);

/* Weight codes: ternary == 0: safeSign(w) as +1/-1 ; ternary != 0: TernaryConnect codes {-1,0,+1}. */
int qt_weight_codes_i8(const float* w, int64_t ldw, int8_t* codes, int64_t ldc_bytes, int64_t rows,
                       int64_t K, int ternary, qt_stream_t stream);
/* The conv weight [Cout][Cin][kh][kw] (fp32, given by its element strides) as the int8 operand of qt_conv2d_implicit*(elem = 1) in one pass: codes of
 * safeSign / ternary, tap-major, Cin rounded to 16 bytes per tap, rows zero-padded to ldc_bytes (layers/dorefa_layers.py:77-82's
 * weight_op for 1-bit weights, without the permute / pad copies). */
int qt_pack_conv_weight_codes_i8(const float* w, int64_t stride_o, int64_t stride_i, int64_t stride_h, int64_t stride_w, int64_t Cout,
                                 int64_t Cin, int64_t kh, int64_t kw, int ternary, int8_t* codes, int64_t ldc_bytes, qt_stream_t stream);

/* ---- weight gradient of a stride-1 conv with +-1 / 0 activations (training; replaces torch.nn.grad.conv2d_weight behind
 * layers/binary_layers.py:105, functions/binary_connect.py:141-143) -------------------------------------------------------
 * dW[co, ci, kh, kw] = sum_{n, oy, ox} g[n, co, oy, ox] * xpad[n, ci, oy + kh, ox + kw] as ONE batched bf16 GEMM launch over
 * K-major operands in the position space q = (y * N + n) * Wq + x (Wq: row pitch, multiple of 8, >= W + 2 pw):
 *   qt_wgrad_pack_grad_f32 : g (fp32, element strides given; channels-last sources take an LDS-tiled transpose) -> A[(t * Cout + co) * lda + q], t = hi / mid / lo of the exact bf16
 *                            split; zero for x >= Wo and q >= Ho * N * Wq.  lda: elements per row, multiple of 64.
 *   qt_wgrad_pack_act_f32  : x (fp32, any strides; x * x_scale must be exact in bf16: +-1 / 0, or k-bit DoReFa activations
 *                            with x_scale = 2^k - 1, whose products are the integer codes) -> kw_count shifted copies
 *                            B[j * copy_elems + ci * ldb + q] = xpad[q + j] (0 outside the image / past the pitch);
 *                            copy_elems == Cin * ldb.
 *   qt_bf16_gemm_taps      : for tap = (r, c), r < tap_rows, c < tap_cols and K slice s < nslice (blockIdx.y):
 *                            Y[(tap * nslice + s) * y_stride + m * ldy + n] = sum_{k < K} X[m][s K + k] * W_tap[n][s K + k],
 *                            W_tap = W + c * w_copy_bytes + r * w_row_bytes (bytes).  K: multiple of 32; ldxp / ldwp in 32-bit
 *                            words, multiples of 32.  Plain bf16 GEMM semantics otherwise (qt_bf16_gemm).
 *   qt_wgrad_reduce_f32    : dW[(co * Cin + ci) * taps + tap] (+)= out_scale * sum over slices and the three row blocks of the partials,
 *                            times 1[|weight| <= ste_threshold] when weight != NULL (the quantiser's straight-through mask). */
int qt_wgrad_pack_grad_f32(const float* g, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                           int64_t Cout, int64_t Ho, int64_t Wo, int64_t Wq, uint16_t* A, int64_t lda, qt_stream_t stream);
int qt_wgrad_pack_act_f32(const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                          int64_t Cin, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t Wq, int64_t kw_count, float x_scale,
                          uint16_t* B, int64_t ldb, int64_t copy_elems, qt_stream_t stream);
int qt_bf16_gemm_taps(const uint32_t* Xh, int64_t ldxp, const uint32_t* Wh, int64_t ldwp, float* Y, int64_t ldy,
                      int64_t M, int64_t N, int64_t K, int64_t tap_rows, int64_t tap_cols, int64_t nslice,
                      int64_t w_copy_bytes, int64_t w_row_bytes, int64_t y_stride, qt_stream_t stream);
int qt_wgrad_reduce_f32(const float* partial, int64_t ldc, int64_t z_stride, int64_t taps, int64_t nslice, int64_t Cout,
                        int64_t Cin, const float* weight, float ste_threshold, float out_scale, int accumulate, float* dW,
                        qt_stream_t stream);

/* ---- the same weight gradient, PIXEL-MAJOR (csrc/wgrad_pm.hip): operands stay [position][channel], a tap is a row offset,
 * one workgroup accumulates all kh * kw taps of a 64 (co) x 64 / 32 (ci) tile and reads its MFMA fragments with the
 * transposing LDS read.  Position space q = (y * N + n) * Wq + x as above.
 *   qt_wgrad_pm_pack_grad_f32 : G3[(t * Qa + q) * Cp + co] = t-th term (hi / mid / lo) of the exact bf16 split of g; zero for
 *                               x >= Wo, q >= Ho * N * Wq, co >= Cout.  Cp % 64 == 0, Qa % 32 == 0.
 *   qt_wgrad_pm_pack_act_f32  : XP[q * Cp + ci] = bf16(xpad * x_scale) over Qx >= (H + 2 ph) * N * Wq rows (zero outside the
 *                               image, past the pitch, in the tail rows and for ci >= Cin).  Cp % 32 == 0.
 *   qt_wgrad_pm_f32           : part[((s * taps + tap) * Cpo + co) * Cpi + ci] = sum over the positions of slice s (Qa / nslice
 *                               each, a multiple of 32) of g[q][co] * XP[q + r * kh_rows + c][ci], tap = r * kw + c.  XP must hold
 *                               Qa + (kh - 1) * kh_rows + 48 rows.  (kh, kw) = (3, 3) with Cpi % 64 == 0 or (5, 5) with
 *                               Cpi % 32 == 0; QT_ERR_UNSUPPORTED otherwise.
 *   qt_wgrad_pm_reduce_f32    : dW[co, ci, i, j] (+)= out_scale * row_scale[co] * sum over slices (row_scale NULL: 1), times the
 *                               straight-through mask 1[|weight| <= ste_threshold] when weight != NULL; dW and weight are addressed
 *                               by ONE set of element strides (contiguous: (Cin taps, taps, kw, 1); or the channels-last
 *                               strides of a channels_last model's parameter). */
int qt_wgrad_pm_pack_grad_f32(const float* g, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                              int64_t Cout, int64_t Ho, int64_t Wo, int64_t Wq, int64_t Cp, int64_t Qa, uint16_t* G3,
                              qt_stream_t stream);
/* qt_wgrad_pm_pack_grad_f32 for channels-last gradients (channel stride 1, Cp <= 2048) that also leaves the per-row channel sums
 * bias_part[(y * N + n) * Cp + c] behind; qt_wgrad_pm_bias_reduce_f32 adds the rows: db[c] (+)= sum_rows bias_part[row][c], c < Cout —
 * the conv's bias gradient without another pass over g.  Fixed summation order (no atomics).  bias_part must hold (Ho * N + 128) * Cp
 * floats: the reduce uses the last 128 rows as scratch for its first pass. */
int qt_wgrad_pm_pack_grad_bias_f32(const float* g, int64_t stride_n, int64_t stride_h, int64_t stride_w, int64_t N, int64_t Cout,
                                   int64_t Ho, int64_t Wo, int64_t Wq, int64_t Cp, int64_t Qa, uint16_t* G3, float* bias_part,
                                   qt_stream_t stream);
int qt_wgrad_pm_bias_reduce_f32(float* bias_part, int64_t rows, int64_t Cp, int64_t Cout, int accumulate, float* db,
                                qt_stream_t stream);
int qt_wgrad_pm_pack_act_f32(const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                             int64_t Cin, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t Wq, int64_t Cp, int64_t Qx,
                             float x_scale, uint16_t* XP, qt_stream_t stream);
/* Strided first layer over a real-valued image: XP[q * Cp + t * Cs8 + (c * s + dy) * s + dx] = t-th term of the exact bf16 split of
 * xpad[n, c, Y s + dy - ph, X s + dx - pw], q = (Y * N + n) * Wq + X over Hs x Ws space-to-depth pixels; Cs8 = C s^2 rounded up
 * to 8, Cp >= 3 Cs8.  The weight gradient of the s2d conv then has 3 Cs8 input channels whose three groups are summed. */
int qt_wgrad_pm_pack_act_s2d_f32(const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                                 int64_t C, int64_t H, int64_t W, int64_t s, int64_t ph, int64_t pw, int64_t Hs, int64_t Ws,
                                 int64_t Wq, int64_t Cs8, int64_t Cp, int64_t Qx, uint16_t* XP, qt_stream_t stream);
int qt_wgrad_pm_f32(const uint16_t* G3, const uint16_t* XP, float* part, int64_t Qa, int64_t kh_rows, int64_t nslice,
                    int64_t Cpo, int64_t Cpi, int64_t kh, int64_t kw, qt_stream_t stream);
int qt_wgrad_pm_reduce_f32(const float* part, int64_t nslice, int64_t taps, int64_t Cpo, int64_t Cpi, int64_t Cout, int64_t Cin,
                           const float* weight, float ste_threshold, float out_scale, const float* row_scale, int accumulate,
                           float* dW, int64_t stride_o, int64_t stride_i, int64_t stride_h, int64_t stride_w, qt_stream_t stream);
/* Round 3: the same weight gradient with the gradient as TWO fp16 planes of g[.., c] / s[c] and the activation plane in fp16 —
 * 2/3 of the MFMAs and of the gradient bytes, bound in csrc/split_f16.hip.  scale2c = [s[0..Cp) | 1 / s[0..Cp)]: PER-CHANNEL
 * powers of two from qt_f16x2_absmax_scale_ch_f32 (a row of dW only sees its own gradient channel, so every row keeps the
 * bound relative to that channel's maximum — a gradient whose channels span many decades loses nothing).  bias_part may be
 * NULL (then any gradient strides are accepted).  qt_wgrad_pm_reduce_f32 takes row_scale = scale2c. */
int qt_wgrad_pm_pack_grad_f16x2(const float* g, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                                int64_t Cout, int64_t Ho, int64_t Wo, int64_t Wq, int64_t Cp, int64_t Qa, const float* scale2c,
                                uint16_t* G2, float* bias_part, qt_stream_t stream);
/* qt_wgrad_pm_pack_act_s2d_f32 with the image as two fp16 terms of x / scale2[0] (qt_f16x2_absmax_scale_f32) in two channel
 * groups, XP[q][t * Cs8 + e], Cp >= 2 * Cs8: the activation plane of qt_wgrad_pm_f16 for a strided real-valued first layer; the
 * caller adds the two channel groups of the result and multiplies by scale2[0]. */
int qt_wgrad_pm_pack_act_s2d_f16x2(const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                                   int64_t C, int64_t H, int64_t W, int64_t s, int64_t ph, int64_t pw, int64_t Hs, int64_t Ws,
                                   int64_t Wq, int64_t Cs8, int64_t Cp, int64_t Qx, const float* scale2, uint16_t* XP,
                                   qt_stream_t stream);
int qt_wgrad_pm_pack_act_f16(const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                             int64_t Cin, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t Wq, int64_t Cp, int64_t Qx,
                             float x_scale, uint16_t* XP, qt_stream_t stream);
int qt_wgrad_pm_f16(const uint16_t* G2, const uint16_t* XP, float* part, int64_t Qa, int64_t kh_rows, int64_t nslice,
                    int64_t Cpo, int64_t Cpi, int64_t kh, int64_t kw, qt_stream_t stream);

/* Y[M,N] = scale * (*scale_dev) * (Xc . Wc^T) + bias, Xc / Wc int8 code planes (ld in uint32 words).
 * scale_dev: optional DEVICE scalar (e.g. E = mean|W| computed on the device) so no host sync is
 * needed; NULL = 1.  max_abs_code bounds |x code * w code| (127 for +-1/0 weight codes, up to 127*127 for k-bit DoReFa
 * weight codes c = rint((2^k-1) w_q)); requires max_abs_code * K < 2^31 (int32 accumulator); the result is exact
 * while max_abs_code * K < 2^24, beyond that the int32 -> fp32 conversion rounds once. */
int qt_i8_gemm(const uint32_t* Xc, int64_t ldxp, const uint32_t* Wc, int64_t ldwp, const float* bias,
               float scale, const float* scale_dev, int64_t max_abs_code, float* Y, int64_t ldy,
               int64_t M, int64_t N, int64_t K, qt_stream_t stream);

/* Split-K form of qt_i8_gemm for skinny problems (few row tiles, long K): slice z (blockIdx.y) contracts bytes
 * [z * kslice, (z + 1) * kslice) of every row (kslice % 64 == 0; both planes hold nslice * kslice zero-padded bytes per row,
 * row strides % 128 bytes == 0) and writes its exact integer partial sums as fp32 — no scale, no bias — to Y + z * y_stride
 * (127 * K < 2^24).  Used by the digit-plane form of LinearXNOR (layers/xnor_layers.py:8-33). */
int qt_i8_gemm_splitk(const uint32_t* Xc, int64_t ldxp, const uint32_t* Wc, int64_t ldwp, float* Y, int64_t ldy, int64_t M, int64_t N,
                      int64_t kslice, int64_t nslice, int64_t y_stride, qt_stream_t stream);

/* Fused inference epilogue between binarised layers:
 *   [MaxPool2d(pool_k, pool_s, no padding, floor)] -> BatchNorm(eval) -> Hardtanh -> BinaryConnect -> pack
 * as ONE pass: bit = (max_window(x) * alpha[c] + beta[c]) < 0.  x: NHWC fp32 [N][H][W][C], C % 4 == 0;
 * alpha/beta: the folded eval BatchNorm (alpha = weight/sqrt(var+eps), beta = bias - mean*alpha);
 * pool_k = pool_s = 1 means no pooling; H = W = 1 covers BatchNorm1d on [N, C].
 * pre_relu != 0 inserts a ReLU between the pooling and the BatchNorm (the Linear -> ReLU -> BatchNorm1d ->
 * BinaryConnect pattern of benchmark/BinaryNet/MLPBin.py:42-44).  Output: NHWC pixel sign plane [N*Ho*Wo][ldp].  Replaces models/Alexnet/Alexnet_Bin.py:14-17 style
 * chains (MaxPool2d, BatchNorm2d, Hardtanh, BinaryConnect) in eval mode. */
int qt_pool_affine_sign_pack_nhwc(const float* x, int64_t N, int64_t H, int64_t W, int64_t C,
                                  int64_t pool_k, int64_t pool_s, const float* alpha, const float* beta,
                                  uint32_t* sign_plane, int64_t ldp, int pre_relu, qt_stream_t stream);

/* The same pass writing, beside the sign plane, the fp4 NIBBLE row plane the next layer's matrix-core GEMM consumes ([N*Ho*Wo][ldn
 * words], +1 = 0x2, -1 = 0xA, features >= C zero; ldn >= 4 * ldp, % 4 == 0): the Linear -> BatchNorm1d -> Hardtanh -> BinaryConnect ->
 * Linear chain of the classifiers (models/Alexnet/Alexnet_Bin.py:40-54) without a separate bits -> nibbles launch. */
int qt_pool_affine_sign_pack_nib_nhwc(const float* x, int64_t N, int64_t H, int64_t W, int64_t C, int64_t pool_k, int64_t pool_s,
                                      const float* alpha, const float* beta, uint32_t* sign_plane, int64_t ldp, uint32_t* nib_plane,
                                      int64_t ldn, int pre_relu, qt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Real-valued activation x quantised weight (first layer of every model, XNOR-Net layers, the general
 * case of LinearBin/LinearTer/BinConv2d/TerConv2d.forward: layers/binary_layers.py:44,105 with an
 * arbitrary fp32 input).  fp32 activations are split EXACTLY into bf16 triples (x = hi + mid + lo) and
 * contracted with +-1/0 bf16 weights (replicated x3) on the bf16 matrix cores: fp32-GEMM accuracy
 * (only the fp32 accumulation order differs) at 1/3 of the bf16 MFMA rate instead of the fp32 rate.
 * ---------------------------------------------------------------------------------------- */

/* mode 0: activation triples of x (times alpha[k] if alpha != NULL: XNORDense's per-input-feature scale,
 * functions/xnor_connect.py:112-113); mode 1/2/3: weight triples of safeSign(x) / TernaryConnect(x) /
 * torch.sign(x).  out: bf16 [rows][ld_bytes/2], element 3k+s = term s of element k; ld_bytes % 16 == 0,
 * ld_bytes >= 6*K, pad zero.  mode 4 = raw weight: bf16_rn(w) replicated three times, for weights that are exactly
 * representable in bf16 (Lin / Log fixed-point levels, layers/log_lin_layers.py). */
int qt_bf16x3_pack_f32(const float* x, int64_t ldx, const float* alpha, uint16_t* out, int64_t ld_bytes,
                       int64_t rows, int64_t K, int mode, qt_stream_t stream);

/* Six-term planes for REAL x REAL contractions (XNORConv2d: weights sign(W) * alpha[kh,kw] are not bf16 values,
 * functions/xnor_connect.py:135-146).  Element k -> bf16 slots 6k..6k+5: role 0 (activation) [xh xh xh xm xm xl],
 * role 1 (weight) [wh wm wl wh wm wh] with x = xh + xm + xl the exact bf16 split: a bf16 GEMM / implicit conv over
 * 6K of the two planes sums the six largest cross terms, i.e. x*w to ~2^-24 relative with fp32 accumulation.
 * out: [rows][ld_bytes/2] bf16, ld_bytes % 16 == 0, ld_bytes >= 12*K, pad zero. */
int qt_bf16x6_pack_f32(const float* x, int64_t ldx, uint16_t* out, int64_t ld_bytes, int64_t rows, int64_t K,
                       int role, qt_stream_t stream);

/* Space-to-depth gather + split for strided first-layer convs: out pixel (n, Y, X), element
 * e = (c*s + dy)*s + dx  <-  triple of x[n, c, s*Y+dy-ph, s*X+dx-pw] (zero outside).  x is addressed by
 * element strides (sN, sC, sH, sW): NCHW or NHWC storage.  Plane rows = N * ceil((H+2ph)/s) * ceil((W+2pw)/s).
 * A k x k / stride s / padding p conv on x == a ceil(k/s)^2, stride-1, un-padded conv on this plane. */
int qt_bf16x3_s2d_pack_f32(const float* x, int64_t sN, int64_t sC, int64_t sH, int64_t sW, uint16_t* out,
                           int64_t ld_bytes, int64_t N, int64_t C, int64_t H, int64_t W, int64_t s,
                           int64_t ph, int64_t pw, qt_stream_t stream);

/* ---- fp16 pair planes: the two-term form of the same path (round 3; csrc/split_f16.hip) -------------------------------------
 * Replaces the same F.linear / F.conv2d call sites as the bf16 triple planes (layers/binary_layers.py:44,105 with a real-valued
 * input: models/Alexnet/Alexnet_Bin.py:13, and the backward convs of functions/binary_connect.py:141-143) at 2/3 of the matrix
 * work: x / s = hi + lo (fp16, 2 x 11 significand bits) with s a per-tensor power of two chosen on the device,
 * |x - s (hi + lo)| <= max(2^-22 |x|, 2^-39 max|x|); weights (+-1 / 0 / integer levels) exact in fp16, replicated twice.
 *   qt_f16x2_scale_f32     : scale2[0] = s = 2^k with max(|*mn|, |*mx|) / s in [2^14, 2^15) (1 for an all-zero or non-finite
 *                            tensor), scale2[1] = 1 / s.  mn / mx: device scalars (torch.aminmax); no host sync.
 *   qt_f16x2_pack_f32      : mode 0: slots (2k, 2k+1) of a row = (hi, lo) of x[k] * scale2[1] (scale2 NULL: 1); modes 1..4:
 *                            safeSign / ternary / torch.sign / raw weight value as fp16, twice.  ld_bytes >= 4 K, multiple
 *                            of 16 (128 for GEMM operands), pad = 0.
 *   qt_f16x2_s2d_pack_f32  : the space-to-depth gather of qt_bf16x3_s2d_pack_f32 with pairs instead of triples.
 *   qt_f16_gemm            : Y[M,N] = scale * (*scale_dev) * Xh . Wh^T (+ bias) over K fp16 elements per row (K = 2 * features);
 *                            ld in uint32 words.  Conv: qt_conv2d_implicit* with elem = 3 and scale_dev = &scale2[0].
 *   qt_f16x2_absmax_scale_f32 : the same scale2 from the tensor itself: max|x| over n DENSE fp32 values (any order) at the HBM
 *                            rate (per-workgroup partial maxima, then a one-workgroup fold; no atomics); work = scratch of
 *                            qt_f16x2_absmax_work_words() uint32 (contents irrelevant).  x 16-byte aligned.
 *   qt_f16x2_absmax_scale_ch_f32 : PER-CHANNEL scales of an [N, C, H, W] tensor given by its strides: scale2c[c] = s[c] from
 *                            max|g[:, c]|, scale2c[Cp + c] = 1 / s[c] (1 for C <= c < Cp).  Dense channels-last tensors are
 *                            reduced at the HBM rate (row-walking workgroups, partial maxima, one fold); any other layout
 *                            works, slower.  work = qt_f16x2_absmax_ch_work_words(C) uint32 of scratch. */
int64_t qt_f16x2_absmax_work_words(void);
int64_t qt_f16x2_absmax_ch_work_words(int64_t C);
int qt_f16x2_absmax_scale_ch_f32(const float* g, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                                 int64_t C, int64_t H, int64_t W, int64_t Cp, uint32_t* work, float* scale2c, qt_stream_t stream);
int qt_f16x2_scale_f32(const float* mn, const float* mx, float* scale2, qt_stream_t stream);
int qt_f16x2_absmax_scale_f32(const float* x, int64_t n, uint32_t* work, float* scale2, qt_stream_t stream);
/* qt_f16x2_absmax_scale_f32 + qt_f16x2_pack_f32(mode 0) of a DENSE [rows, K] fp32 matrix in TWO launches instead of three (round 6,
 * the training step's launch diet — functions/binary_connect.py:104-112's backward GEMMs split their gradient operand here): the
 * pack's workgroups fold the partial maxima themselves.  scale3 = [s, 1 / s, s * (*mul_dev)] (mul_dev NULL: s): the third slot
 * is the scale_dev of the consuming qt_f16_gemm / qt_conv2d_implicit* when the contraction carries another device scalar (DoReFa's
 * E = mean|W|, functions/dorefa_connect.py:100) — s is a power of two, so the product is exact.  work as above. */
/* E = mean|x| of n dense fp32 values as a device scalar, ONE launch (DoReFa's 1-bit weight scale, functions/dorefa_connect.py:100;
 * torch: abs + mean [+ fill]).  Deterministic: partial sums are added in index order by the last workgroup to arrive.  `work`:
 * qt_abs_mean_work_words() uint32, 8-byte aligned, ZERO on first use — the kernel leaves it zero again, so one buffer serves every
 * later call that is ordered behind this one on the same stream (calls on different streams need different buffers). */
int64_t qt_abs_mean_work_words(void);
int qt_abs_mean_f32(const float* x, int64_t n, uint32_t* work, float* out, qt_stream_t stream);
/* Stride-1 3 x 3 / padding-1 FIRST LAYER on a real-valued fp32 image (<= 4 channels, any strides) with 64 output channels, one pass
 * over the image (csrc/conv_first3x3.hip; VGG-16's conv1_1 as TerConv2d / BinConv2d: layers/terner_layers.py:89-92,
 * binary_layers.py:103-106 — the reference's F.conv2d(x, Q(W), b, 1, 1)).  The 34 x 34-pixel patch of a 32 x 32 output tile is split
 * into two fp16 terms with the tile's own power-of-two scale (|x - s (hi + lo)| <= max(2^-22 |x|, 2^-39 tilemax)); weights +-1 / 0.
 *   wfrag : qt_conv3x3_first_pack_weight_f32 of the QUANTISED weight [Cout <= 64, C, 3, 3] (element strides), 6144 bytes.
 *   mode 0: out = fp32 [N*H*W, ldo] (+ bias);  mode 1: threshold bits [N*H*W, 4 words], bit c = ((conv + bias) * alpha[c] < -beta[c]);
 *   mode 2: the same predicate as the fp4 nibble plane [N*(H+2)*(W+2), 8 words] (-1 = 0xA, +1 = 0x2) with its 1-pixel zero halo —
 *   the operand of the next 3 x 3 conv.  All three come from ONE accumulation order.  Cout != 64: QT_ERR_UNSUPPORTED. */
int qt_conv3x3_first_pack_weight_f32(const float* wq, int64_t stride_o, int64_t stride_i, int64_t stride_h, int64_t stride_w, int64_t C,
                                     int64_t Cout, uint32_t* wfrag, qt_stream_t stream);
int qt_conv3x3_first_f32(const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N, int64_t C,
                         int64_t H, int64_t W, const uint32_t* wfrag, int64_t Cout, const float* bias, const float* alpha,
                         const float* beta, void* out, int64_t ldo, int mode, qt_stream_t stream);
int qt_f16x2_absmax_pack_f32(const float* x, int64_t rows, int64_t K, uint32_t* work, const float* mul_dev, float* scale3, uint16_t* out,
                             int64_t ld_bytes, qt_stream_t stream);
int qt_f16x2_pack_f32(const float* x, int64_t ldx, const float* scale2, uint16_t* out, int64_t ld_bytes, int64_t rows,
                      int64_t K, int mode, qt_stream_t stream);
int qt_f16x2_s2d_pack_f32(const float* x, int64_t sN, int64_t sC, int64_t sH, int64_t sW, const float* scale2, uint16_t* out,
                          int64_t ld_bytes, int64_t N, int64_t C, int64_t H, int64_t W, int64_t s, int64_t ph, int64_t pw,
                          qt_stream_t stream);
/* The same plane WITHOUT a separate max|x| pass over the image (channels-last images only, else QT_ERR_UNSUPPORTED): the image is
 * packed with the fixed power-of-two scale `spec_scale` while max|x| is folded on the way (work = qt_f16x2_s2d_spec_work_words(N, H,
 * s, ph) uint32 of scratch); a one-workgroup launch then decides: max|x| / spec_scale in [2^8, 2^16) (or max|x| = 0) -> scale2 =
 * [spec_scale, 1 / spec_scale], *redo = 0 and the plane stands, with |x - s (hi + lo)| <= max(2^-22 |x|, 2^-33 max|x|); otherwise
 * scale2 = the exact-binade scale of qt_f16x2_absmax_scale_f32, *redo = 1, and a third launch (which returns at once when *redo = 0)
 * rewrites the plane with it.  No host synchronisation; the consumer reads the scale from scale2 on the device.  Replaces the
 * operand preparation of the first layer's F.conv2d on real pixels (models/Alexnet/Alexnet_Bin.py:13, layers/binary_layers.py:105). */
int64_t qt_f16x2_s2d_spec_work_words(int64_t N, int64_t H, int64_t s, int64_t ph);
int qt_f16x2_s2d_pack_spec_f32(const float* x, int64_t sN, int64_t sC, int64_t sH, int64_t sW, float spec_scale, uint32_t* work,
                               float* scale2, int* redo, uint16_t* out, int64_t ld_bytes, int64_t N, int64_t C, int64_t H,
                               int64_t W, int64_t s, int64_t ph, int64_t pw, qt_stream_t stream);
/* The conv weight [Cout][Cin][kh][kw] (fp32, given by its element strides: contiguous or channels-last) as the fp16 pair-plane operand of qt_conv2d_implicit(elem = 3): quantised
 * by `mode` (1 safeSign, 2 ternary, 3 torch.sign, 4 raw) and replicated twice, rows tap-major with 16-byte tap granules, padded with
 * zeros to ld_bytes (a multiple of 128).  transpose_flip = 1: the operand of grad_x — rows = Cin, channels = Cout, taps flipped
 * (torch.nn.grad.conv2d_input's conv_transpose as a conv; functions/binary_connect.py:141). */
int qt_f16x2_pack_conv_weight_f32(const float* w, int64_t stride_o, int64_t stride_i, int64_t stride_h, int64_t stride_w, int64_t Cout,
                                  int64_t Cin, int64_t kh, int64_t kw, int mode, int transpose_flip, uint16_t* out, int64_t ld_bytes,
                                  qt_stream_t stream);
int qt_f16_gemm(const uint32_t* Xh, int64_t ldxp, const uint32_t* Wh, int64_t ldwp, const float* bias, float scale,
                const float* scale_dev, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, qt_stream_t stream);

/* ---- training-mode chain between two binarised layers (csrc/train_chain.hip, round 3) -------------------------------------
 * [MaxPool2d(k, s)] -> BatchNorm (BATCH statistics) -> [Hardtanh(lo, hi)] -> BinaryConnectDeterministic, forward and backward:
 * replaces F.max_pool2d / F.batch_norm(training=True) / F.hardtanh and their autograd backwards in
 * models/Alexnet/Alexnet_Bin.py:13-17, benchmark/BinaryNet/MLPBin.py:42-44 (torch / MIOpen kernels in the module graph), and
 * the STE of functions/binary_connect.py:31-38.  x: NHWC fp32 [N][H][W][C] (a [N, C] matrix: H = W = 1); pooled rows
 * R = N * Ho * Wo, Ho = (H - k) / s + 1 (no padding, floor mode; k = 1: no pooling, p and idx unused / NULL).
 *   forward : p[R][C] pooled values, idx[R][C] int8 argmax (first maximum in scan order), mean[C], invstd[C] (biased variance,
 *             two passes), running statistics updated in place (NULL: skipped; unbiased variance, as torch), sgn[R][C] = +-1 of
 *             clamp(((p - mean) * invstd) * gamma + beta, lo, hi) (no Hardtanh: lo = -inf, hi = +inf; gamma / beta NULL: 1 / 0).
 *             partial: scratch of qt_train_chain_partial_floats(R, C) floats.
 *   backward: g[R][C] = dL/dsgn -> dgamma[C], dbeta[C], gp[R][C] = dL/dp, and (k > 1) gx[N][H][W][C] = dL/dx by a gather over
 *             the windows that contain each pixel (no atomics).  p_or_x: p (k > 1) or x (k = 1) of the forward. */
int64_t qt_train_chain_partial_floats(int64_t R, int64_t C);
int qt_pool_bn_sign_train_f32(const float* x, int64_t N, int64_t H, int64_t W, int64_t C, int64_t k, int64_t s,
                              const float* gamma, const float* beta, float eps, float momentum, float ht_lo, float ht_hi,
                              float* running_mean, float* running_var, float* p, int8_t* idx, float* mean, float* invstd,
                              float* partial, float* sgn, qt_stream_t stream);
int qt_pool_bn_sign_train_backward_f32(const float* g, const float* p_or_x, const int8_t* idx, int64_t N, int64_t H, int64_t W,
                                       int64_t C, int64_t k, int64_t s, const float* gamma, const float* beta,
                                       const float* mean, const float* invstd, float ht_lo, float ht_hi, float ste_threshold,
                                       float* partial, float* dgamma, float* dbeta, float* gp, float* gx, qt_stream_t stream);

/* Training-mode form of the chain between two DoReFa layers (reference: models/Resnet/Resnet_bin.py:63-97 —
 * conv -> BatchNorm2d (batch statistics) [+ shortcut] -> ReLU -> nnDorefaQuant; functions/dorefa_connect.py:28-45 identity STE),
 * over channels-last fp32 pixels x[R][C], C % 4 == 0.
 *   qt_bn_train_stats_f32        : stats2 = [mean[C] | invstd[C]] (biased variance, two passes, folded in double) + the running
 *                                  statistics update (NULL: skipped).  The forward's second half is qt_affine_dorefa_codes_i8 with
 *                                  alpha / beta = BatchNorm weight / bias and bn_stats = stats2 (codes, fp32 image, residual, ReLU).
 *   qt_bn_act_train_backward_f32 : t = fma((x - mean) * invstd, gamma, beta) + res (res NULL: 0); g_t = g * 1[t > 0] when relu,
 *                                  else g; dgamma = sum g_t xhat, dbeta = sum g_t, gx = gamma invstd (g_t - dbeta / R -
 *                                  xhat dgamma / R), gres (optional) = g_t.  partial: qt_train_chain_partial_floats(R, C) floats. */
int qt_bn_train_stats_f32(const float* x, int64_t R, int64_t C, float eps, float momentum, float* running_mean,
                          float* running_var, float* stats2, float* partial, int32_t* zero_flag /* NULL, or a word set to 0: the range
                          flag of the quantiser pass that follows (no fill launch) */, qt_stream_t stream);
int qt_bn_act_train_backward_f32(const float* g, const float* x, const float* res, int64_t R, int64_t C, const float* gamma,
                                 const float* beta, const float* stats2, int relu, float* partial, float* dgamma, float* dbeta,
                                 float* gx, float* gres, qt_stream_t stream);

/* Y[M,N] = Xh . Wh^T (+ bias) over K bf16 elements per row (K = 3 * features for triple planes);
 * ld in uint32 words. */
int qt_bf16_gemm(const uint32_t* Xh, int64_t ldxp, const uint32_t* Wh, int64_t ldwp, const float* bias,
                 float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, qt_stream_t stream);

/* nibble plane from existing bit planes (sign only: mask_plane == NULL; ternary: mask + sign).
 * 1 bit -> 4 bits per element; lets the canonical 1-bit planes (what the quantisers emit and what
 * eval-mode layers cache) feed the matrix-core GEMM without re-reading the fp32 tensor. */
int qt_bits_to_nib(const uint32_t* sign_plane, const uint32_t* mask_plane, int64_t ldb,
                   uint32_t* nib_plane, int64_t ldn, int64_t rows, int64_t K, qt_stream_t stream);

/* Physical zero padding of an NHWC pixel plane of any element type (nibble, int8 code, bf16 triple: pixels are whole
 * 16-byte chunks, zero bytes = value 0): P [N][H][W][Cw words] -> Q [N][H+2ph][W+2pw][Cw], border pixels zero, Cw % 4 == 0.
 * The zero-padded conv on P (F.conv2d's padding argument) is the un-padded conv on Q. */
int qt_pad_pixel_plane(const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw, int64_t ph, int64_t pw,
                       uint32_t* Q, qt_stream_t stream);

/* Zero the border pixels (only) of a halo plane Q [N][H + 2*halo_h][W + 2*halo_w][Cw words]: for callers that fill
 * the interior of such a plane themselves (the conv / pooling entry points above write their own borders). */
int qt_zero_halo(uint32_t* Q, int64_t N, int64_t H, int64_t W, int64_t Cw, int64_t halo_h, int64_t halo_w,
                 qt_stream_t stream);

/* Same expansion into a PHYSICALLY zero-padded NHWC pixel plane: sign/mask planes hold N*H*W pixel rows,
 * nib_plane gets N*(H+2ph)*(W+2pw) rows of ldn words whose border pixels are fp4 zeros.  A conv with zero
 * padding (ph, pw) on the original image equals the un-padded conv on this plane, which the implicit-GEMM
 * kernels run without per-tap bounds checks (F.conv2d's padding argument, layers/binary_layers.py:105). */
int qt_bits_to_nib_pad(const uint32_t* sign_plane, const uint32_t* mask_plane, int64_t ldb,
                       uint32_t* nib_plane, int64_t ldn, int64_t N, int64_t H, int64_t W, int64_t ph,
                       int64_t pw, int64_t K, qt_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Quantised conv2d = packed-domain im2col + packed GEMM (replaces torch.nn.functional.conv2d on
 * quantised operands: layers/binary_layers.py:105,106; layers/terner_layers.py:91,92).
 * P: NHWC pixel plane [N][H][W][Cw] of words (bit or nibble plane of the channel vector, Cw % 4 == 0,
 * pad channels zero).  A: rows m = (n,ho,wo) in [m_begin, m_begin+m_count), each the kh*kw taps' Cw
 * words (ZERO words for taps in the padding), zero-filled up to ldA.  With nibble planes the zero
 * words are the reference's zero padding exactly.  Y = A . Wpacked^T with Wpacked[Cout][kh*kw*Cw]
 * (weights packed per (cout, tap) with the same Cw) is then the NHWC conv output.
 * ---------------------------------------------------------------------------------------- */
int qt_im2col_words(const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw, int64_t kh,
                    int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh,
                    int64_t dw, uint32_t* A, int64_t ldA, int64_t m_begin, int64_t m_count,
                    qt_stream_t stream);

/* Implicit-GEMM form of the same conv: no im2col matrix is materialised — the GEMM kernel's LDS-DMA
 * gathers 16-byte pixel chunks straight from the NHWC plane P (zero page for padding taps).
 * elem: 0 = fp4 nibble planes (scale ignored), 1 = int8 code planes (Y = scale * (*scale_dev) * acc), 3 = fp16 pair planes (same scaling),
 * 2 = bf16 triple planes.  Wmat: [Cout][ldwp] words, row = kh*kw taps x Cw words, ldwp % 32 == 0.
 * Y: NHWC [N*Ho*Wo][ldy] fp32. */
int qt_conv2d_implicit(int elem, const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw,
                       int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                       int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldwp, const float* bias,
                       float scale, const float* scale_dev, float* Y, int64_t ldy, int64_t Cout,
                       qt_stream_t stream);

/* qt_conv2d_implicit with the main loop chosen by the caller (tests / tuning; an argument, not process state):
 * variant 0 = automatic (ping-pong 384x192 tile for 192-wide column tiles, double-buffered otherwise),
 * 1 = double-buffered, 2 = ping-pong, 4 = automatic without the un-padded fast path (A/B only);
 * 3 (stamped kernel, Y garbage) exists only in -DQT_PROFILING_VARIANTS builds and is QT_ERR_UNSUPPORTED otherwise. */
int qt_conv2d_implicit_variant(int variant, int elem, const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw,
                               int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                               int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldwp, const float* bias,
                               float scale, const float* scale_dev, float* Y, int64_t ldy, int64_t Cout,
                               qt_stream_t stream);

/* Same conv with the threshold-bit epilogue (inference fusion of
 *   BinConv2d -> [MaxPool2d] -> BatchNorm2d(eval) -> Hardtanh -> BinaryConnect, models/Alexnet/Alexnet_Bin.py:13-17):
 * no fp32 output is written; per output element t = acc (+ bias), v = fl(fl(t * alpha[c]) + beta[c]) and
 * bit c%32 of word c/32 of row (n, ho, wo) of neg_plane is (v < 0).  neg_plane: [N*Ho*Wo][ldb] words,
 * ldb % 4 == 0, ceil(Cout/32) <= ldb < ceil(Cout/32) + 64; every word of every row is written (pad words as 0).
 * alpha/beta: folded eval BatchNorm, Cout floats each.  Follow with qt_pool_bits when a MaxPool sits
 * between conv and BatchNorm.
 * thr (may be NULL; elem 0 / 1 only): per-channel INTEGER thresholds T_c such that, for every integer accumulator value
 * |acc| <= K,  fl(fl(acc + bias_c)*alpha_c) + beta_c < 0  <=>  (acc < T_c) xor (alpha_c < 0)  — the left side is a
 * monotone step function of the integer acc, so T_c exists and the caller finds it by bisection once per layer.  The
 * kernel then spends one compare per output instead of add + multiply + compare ("BatchNorm + sign collapses to a
 * per-channel integer threshold on the popcount"); bias / beta are ignored, the sign of alpha is still read. */
int qt_conv2d_implicit_bits(int elem, const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw,
                            int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                            int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldwp, const float* bias,
                            float scale, const float* scale_dev, const float* alpha, const float* beta,
                            const float* thr, uint32_t* neg_plane, int64_t ldb, int64_t Cout, qt_stream_t stream);

/* qt_conv2d_implicit_bits with the sign bits written as the NEXT conv's operand instead: an fp4 nibble pixel plane
 * (+1 = 0x2, -1 = 0xA, channels >= Cout zero), nib_plane [N][Ho + 2*out_halo_h][Wo + 2*out_halo_w][ldn words],
 * ldn == ceil(Cout/32)*4.  The launch writes every word of the plane, the zero border included — the halo is the next
 * conv's zero padding.  Replaces qt_conv2d_implicit_bits +
 * qt_bits_to_nib_pad when no pooling sits between two binarised convs.
 * d2s_cout != 0 (depth-to-space by 2): Cout == 4*d2s_cout columns ordered (dy, dx, channel) are written to pixel
 * (2*ho + dy, 2*wo + dx) of a [N][2*Ho + 2*halo_h][2*Wo + 2*halo_w][ldn] plane with d2s_cout channels
 * (ldn == ceil(d2s_cout/32)*4, d2s_cout % 32 == 0): the epilogue of the 2x2 output-blocked form of a few-channel
 * stride-1 3x3 first layer (a 4x4 stride-2 conv embedding the four shifted copies of the 3x3 kernel). */
int qt_conv2d_implicit_nib(int elem, const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw,
                           int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh,
                           int64_t dw, const uint32_t* Wmat, int64_t ldw, const float* bias, float scale,
                           const float* scale_dev, const float* alpha, const float* beta, const float* thr,
                           uint32_t* nib_plane, int64_t ldn, int64_t Cout, int64_t out_halo_h, int64_t out_halo_w,
                           int64_t d2s_cout, qt_stream_t stream);

/* ---- direct first-layer conv (csrc/conv_first_direct.hip) ---------------------------------------------------------------------------
 * conv2d(x, Q(W), bias, stride S, padding (PH, PW)) for a REAL-valued fp32 image with a few channels (C <= Cp <= 8) and a large /
 * strided kernel — the first layer of the reference's CNNs (models/Alexnet/Alexnet_Bin.py:13: 3 -> 192, 11 x 11, stride 4, padding 2;
 * layers/binary_layers.py:103-106, terner_layers.py:89-92, functions/xnor_connect.py:139-146).  x: [N, C, H, W] with element strides
 * (sn, sc, sh, sw) — NCHW or channels-last storage, read where it lies.  A workgroup loads the input patch of its <= 128 output
 * pixels once, splits it into two fp16 terms of x / s with the TILE's power-of-two s (|x - s (hi + lo)| <= max(2^-22 |x|, 2^-39 tile
 * max)) and contracts K = (ky, kx, c) on the fp16 matrix cores by stride addressing of the patch: no space-to-depth plane, no
 * operand pack pass.  Weights, packed once by the caller ([k-step][2][Coutp][8 fp16]; the k axis strings the KH kernel rows together, each
 * ceil(KW * Cp / 4) * 4 elements in (kx, c) order, zero padded; k-steps = ceil(k / 16), the last one zero-filled; Coutp = Cout rounded up to 32):
 *   w_lo == NULL: +-1 / 0 (or small-integer) weights, exact in fp16 (BinConv2d / TerConv2d): w_hi alone;
 *   w_lo != NULL: real-valued weights (XNOR-Net's sign(W) * alpha) as w / (w_scale * w_scale_dev[0]) = w_hi + w_lo (three
 *                 products per element: hi whi + lo whi + hi wlo).
 * S * Cp % 4 == 0 (QT_ERR_ALIGNMENT otherwise: pad the channel count).  y: fp32 NHWC [N * Ho * Wo][ldy]. */
int qt_conv_first_direct_f32(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int64_t N, int64_t C, int64_t H,
                             int64_t W, int64_t KH, int64_t KW, int64_t S, int64_t PH, int64_t PW, int64_t Cp, const uint32_t* w_hi,
                             const uint32_t* w_lo, float w_scale, const float* w_scale_dev, int64_t Cout, int64_t Coutp,
                             const float* bias, float* y, int64_t ldy, qt_stream_t stream);

/* ... with the BatchNorm-threshold bit epilogue of qt_conv2d_implicit_bits (float form): neg_plane [N * Ho * Wo][ldb words]. */
int qt_conv_first_direct_bits_f32(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int64_t N, int64_t C, int64_t H,
                                  int64_t W, int64_t KH, int64_t KW, int64_t S, int64_t PH, int64_t PW, int64_t Cp,
                                  const uint32_t* w_hi, const uint32_t* w_lo, float w_scale, const float* w_scale_dev, int64_t Cout,
                                  int64_t Coutp, const float* bias, const float* alpha, const float* beta, uint32_t* neg_plane,
                                  int64_t ldb, qt_stream_t stream);

/* ---- per-tap scaled convs: the XNOR-Net family (functions/xnor_connect.py:135-169, layers/xnor_layers.py:36-69) ----------------
 * XNORConv2d computes conv2d(x, sign(W) * alpha) with alpha = mean(|W|, dim = [0, 1], keepdim) -> [1, 1, kh, kw]: one scale per
 * filter TAP (xnor_connect.py:140-145).  With the implicit GEMM's tap-major K order
 *     y[m, n] = sum_t alpha_t * D_t[m, n],    D_t = sum over the input channels of tap t      (exact integer for +-1 activations)
 * is evaluated in Horner form on the accumulators: at the boundary in front of tap t they are multiplied by
 * rho_t = alpha_{t-1} / alpha_t, the taps' MFMAs keep accumulating onto them, and the epilogue multiplies by alpha_{T-1}: one
 * matrix-core pass over the sign planes (the six bf16 passes of a real x real conv otherwise), 2 (T - 1 - t) extra fp32 roundings
 * on tap t's term.  This is the `qt_xnor_conv2d_taps` of SURVEY 8(b).
 *
 * qt_xnor_tap_prep_f32: alpha and the Horner tables of a conv weight viewed as row-major [R = Cout * Cin, taps = kh * kw].
 *   w != NULL, alpha_in == NULL: alpha[t] = mean_r |w[r, t]| (deterministic two-stage reduction; `work` =
 *       qt_xnor_tap_prep_work_floats(R, taps) floats), written to `alpha` (optional) — training mode, xnor_connect.py:140;
 *   alpha_in != NULL: the scales are given (eval mode: the weight already holds sign(W) * alpha, layers/xnor_layers.py:54-61).
 *   tables [2 * (taps + 1)]: [1, rho_1 .. rho_{T-1}, a'_{T-1}] for the forward tap order, then the same for the REVERSED tap order
 *   (the gradient w.r.t. the input convolves with the flipped kernel, xnor_connect.py:154-155).  a' = alpha with zeros replaced
 *   by the preceding non-zero scale (a tap with alpha == 0 has all-zero weights, D_t = 0).  taps <= 1024. */
int64_t qt_xnor_tap_prep_work_floats(int64_t R, int64_t taps);
int qt_xnor_tap_prep_f32(const float* w, int64_t R, int64_t taps, const float* alpha_in, float* work, float* alpha,
                         float* tables, qt_stream_t stream);

/* qt_conv2d_implicit with per-tap scaling: Y = (scale * scale_dev) * sum_t alpha_t D_t + bias.  elem 0: fp4 nibble planes (+-1
 * activations x sign(W), zeros for W == 0 as torch.sign gives); elem 3: fp16 pair planes (a real-valued operand, e.g. the
 * gradient, against the replicated +-1 / 0 weight).  tap_rho: one (taps + 1)-entry table of qt_xnor_tap_prep_f32.  A tap must be
 * a whole number of 32-byte MFMA k-steps: Cw % 8 == 0 (channels padded to 64 for fp4, to 8 for fp16 pairs), else
 * QT_ERR_ALIGNMENT.  Everything else as qt_conv2d_implicit. */
int qt_conv2d_implicit_taps(int elem, const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw, int64_t kh, int64_t kw,
                            int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat,
                            int64_t ldw, const float* bias, float scale, const float* scale_dev, const float* tap_rho, float* Y,
                            int64_t ldy, int64_t Cout, qt_stream_t stream);

/* ... with the threshold-bit epilogue of qt_conv2d_implicit_bits (float form: the accumulators are not integers), */
int qt_conv2d_implicit_taps_bits(int elem, const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw, int64_t kh, int64_t kw,
                                 int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat,
                                 int64_t ldw, const float* bias, float scale, const float* scale_dev, const float* tap_rho,
                                 const float* alpha, const float* beta, uint32_t* neg_plane, int64_t ldb, int64_t Cout,
                                 qt_stream_t stream);

/* ... and with the sign bits written as the next conv's nibble pixel plane (qt_conv2d_implicit_nib without depth-to-space). */
int qt_conv2d_implicit_taps_nib(int elem, const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw, int64_t kh, int64_t kw,
                                int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat,
                                int64_t ldw, const float* bias, float scale, const float* scale_dev, const float* tap_rho,
                                const float* alpha, const float* beta, uint32_t* nib_plane, int64_t ldn, int64_t Cout,
                                int64_t out_halo_h, int64_t out_halo_w, qt_stream_t stream);

/* Direct form of the 3x3 / stride 1 / padding 1 conv of a +-1 activation for few channels at large spatial size
 * (VGG / ResNet early layers), where the implicit-GEMM gather of qt_conv2d_implicit_* is bound by L2 traffic (each
 * input pixel fetched 9 times).  P: fp4 nibble pixel plane WITH a 1-pixel zero halo, [N][H+2][W+2][Cw words], Cw = 8 or
 * 16 (64 / 128 channels); Wmat [Cout][ldw words] as for qt_conv2d_implicit (tap-major K); Cout <= 128.
 * Threshold epilogue as qt_conv2d_implicit_bits (alpha, beta = folded BatchNorm, bit = fl((acc + bias)*alpha) + beta < 0):
 *   out_bits == 0: out = the NEXT conv's operand, nibble plane [N][H+2][W+2][ldo] with the same halo (ldo ==
 *                  ceil(Cout/32)*4); every word is written, the halo as zeros;
 *   out_bits != 0: out = bit plane [N*H*W][ldo] (ldo >= ceil(Cout/32)), for a MaxPool on bits.
 * elem 0: fp4 nibble planes as above; elem 2: bf16 triple planes of a REAL-valued input (first layer: <= 5 channels, 32-byte
 * pixels, Cw = 8; weights as for the bf16 implicit conv).  Bit-identical to the implicit-GEMM entry points for integer
 * accumulators (elem 0); elem 2 sums the same exact products in fp32 in tap order.  QT_ERR_UNSUPPORTED for other channel
 * counts. */
int qt_conv3x3_direct_nib(int elem, const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw,
                          const uint32_t* Wmat, int64_t ldw, const float* bias, const float* alpha, const float* beta, uint32_t* out,
                          int64_t ldo, int64_t Cout, int out_bits, qt_stream_t stream);

/* The real-valued 3 x 3 / stride 1 / padding 1 first layer (<= 4 channels) on fp16 PAIR planes (round 4): P = the halo-1 plane
 * qt_f16x2_s2d_pack_f32 / _spec_f32 write for s = 1, padding 1 (16 bytes per pixel: [hi, lo] of each channel of x / scale),
 * scale_dev = that plane's power-of-two scale, Wmat = the tap-major pair rows of qt_f16x2_pack_conv_weight_f32 (16 bytes per
 * tap, ldw words per output channel).  Same outputs as qt_conv3x3_direct_nib (threshold bits, or the next conv's halo-1 nibble
 * plane); the two lane halves of one MFMA take two taps (5 MFMAs per 32 x 32 block instead of 9).  Replaces the
 * F.conv2d(x, ter_op(W)) of layers/terner_layers.py:89-92 / binary_layers.py:103-106 for VGG-style first layers. */
int qt_conv3x3_direct_pairs(const uint32_t* P, int64_t N, int64_t H, int64_t W, const uint32_t* Wmat, int64_t ldw,
                            const float* bias, const float* scale_dev, const float* alpha, const float* beta, uint32_t* out,
                            int64_t ldo, int64_t Cout, int out_bits, qt_stream_t stream);

/* The direct 3x3 kernel for DoReFa int8 code planes with the code epilogue of qt_conv2d_implicit_codes (same arithmetic,
 * bit-identical): P, codes and (optional) res_codes are planes with a 1-pixel halo of identical geometry,
 * [N][H+2][W+2][.]; every byte of `codes` is written, the halo as zeros.  Cw = 16 words (64 input channels), Cout <= 64,
 * ldc_bytes == Cout rounded up to 16.  QT_ERR_UNSUPPORTED for other channel counts. */
int qt_conv3x3_direct_codes(const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw, const uint32_t* Wmat,
                            int64_t ldw, const float* bias, float scale, const float* scale_dev, const float* alpha,
                            const float* beta, const int8_t* res_codes, int64_t ldrc_bytes, float res_scale, int relu,
                            int bit_width, int8_t* codes, int64_t ldc_bytes, int64_t Cout, int32_t* overflow,
                            qt_stream_t stream);

/* qt_pool_bits with the pooled bits written the same way (nibble pixel plane of C channels, optional halo):
 * qt_pool_bits + qt_bits_to_nib_pad in one pass. */
int qt_pool_bits_nib(const uint32_t* in_plane, int64_t N, int64_t H, int64_t W, int64_t ld, int64_t pool_k,
                     int64_t pool_s, const uint32_t* neg_alpha, uint32_t* nib_plane, int64_t ldn, int64_t C,
                     int64_t out_halo_h, int64_t out_halo_w, qt_stream_t stream);

/* The same conv with the k-bit DoReFa chain of qt_affine_dorefa_codes_i8 applied to the accumulators (code epilogue):
 *   t = fl(fl(y*alpha[c]) + beta[c]) [+ residual as there] ; [ReLU] ; q = rint((2^k-1) * t)
 * where y is exactly the fp32 value qt_conv2d_implicit would have stored.  codes: int8 [N*Ho*Wo][ldc_bytes]
 * (ldc_bytes % 16 == 0, >= Cout rounded up to 4; pad bytes zeroed), i.e. the NHWC code plane the next DorefaConv2d
 * gathers from: no fp32 activation is written between two quantised convs.  elem must be 1 (int8 code planes).  Residual rows are output pixels
 * (res_f32 [N*Ho*Wo][ldr] / res_codes [N*Ho*Wo][ldrc_bytes]).  *overflow as qt_dorefa_codes_i8.
 * Halo planes: a pixel plane may carry a zero border of (halo_h, halo_w) pixels around every image,
 * [N][H + 2*halo_h][W + 2*halo_w][C]: the zero padding of a conv that reads it is then physical and the conv runs
 * the un-padded kernels (no per-tap bounds checks).  in_halo_*: halo of P (needs ph <= in_halo_h, pw <= in_halo_w;
 * QT_ERR_UNSUPPORTED when the plane exceeds 4 GiB); out_halo_*: halo of `codes` — the launch writes
 * the interior pixels and the zero border (every byte of the plane); res_halo_*: halo of `res_codes`.
 * bn_stats / res_bn_stats: the device BatchNorm arithmetic of qt_affine_dorefa_codes_i8 ([mean | rs], alpha / beta = weight / bias). */
int qt_conv2d_implicit_codes(int elem, const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw,
                             int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh,
                             int64_t dw, const uint32_t* Wmat, int64_t ldw, const float* bias, float scale,
                             const float* scale_dev, const float* alpha, const float* beta, const float* res_f32,
                             int64_t ldr, const float* res_alpha, const float* res_beta, const int8_t* res_codes,
                             int64_t ldrc_bytes, float res_scale, int relu, int bit_width, int8_t* codes,
                             int64_t ldc_bytes, int64_t Cout, int32_t* overflow, int64_t in_halo_h,
                             int64_t in_halo_w, int64_t out_halo_h, int64_t out_halo_w, int64_t res_halo_h,
                             int64_t res_halo_w, const float* bn_stats, const float* res_bn_stats, qt_stream_t stream);

/* qt_conv2d_implicit reading a halo plane P [N][H + 2*halo_h][W + 2*halo_w][Cw] (see qt_conv2d_implicit_codes):
 * fp32 output as qt_conv2d_implicit.  ph <= halo_h, pw <= halo_w. */
int qt_conv2d_implicit_halo(int elem, const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw,
                            int64_t halo_h, int64_t halo_w, int64_t kh, int64_t kw, int64_t sh, int64_t sw,
                            int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldw,
                            const float* bias, float scale, const float* scale_dev, float* Y, int64_t ldy,
                            int64_t Cout, qt_stream_t stream);

/* qt_conv2d_implicit_halo followed by eval-mode BatchNorm in the DEVICE's arithmetic, in the conv's epilogue:
 *   y = fma(fl(fl(v - mean[c]) * rs[c]), bn_weight[c], bn_bias[c]),  v = the value qt_conv2d_implicit_halo stores, bn_stats = [mean | rs]
 * (the expression of qt_bn_eval_device_f32).  The shortcut branch DorefaConv2d(1x1, stride 2) -> BatchNorm2d of
 * models/samples/ResNet_Dorefa.py in one launch: its fp32 result joins the main branch's code epilogue as a plain residual.
 * elem == 1 (int8 codes) only; Cout % 4 == 0, ldy % 4 == 0, 16-byte aligned Y / BatchNorm vectors. */
int qt_conv2d_implicit_halo_bn(int elem, const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw,
                               int64_t halo_h, int64_t halo_w, int64_t kh, int64_t kw, int64_t sh, int64_t sw,
                               int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldw,
                               const float* bias, float scale, const float* scale_dev, const float* bn_weight,
                               const float* bn_bias, const float* bn_stats, float* Y, int64_t ldy, int64_t Cout,
                               qt_stream_t stream);

/* MaxPool2d(pool_k, pool_s, no padding, floor mode) evaluated on threshold bits: out = AND over the
 * window where alpha >= 0, OR where alpha < 0 (max-pooling commutes with the monotone map x*alpha+beta;
 * bit-identical to pooling the fp32 tensor first, NaNs excepted).  in_plane: [N*H*W][ld] words NHWC
 * pixel plane, out_plane: [N*Ho*Wo][ld]; neg_alpha: ld words, bit c = (alpha[c] < 0). */
int qt_pool_bits(const uint32_t* in_plane, int64_t N, int64_t H, int64_t W, int64_t ld, int64_t pool_k,
                 int64_t pool_s, const uint32_t* neg_alpha, uint32_t* out_plane, qt_stream_t stream);

/* Tuning / diagnostic entry: same contract as qt_nib_gemm with an explicit kernel configuration.
 * 0 = automatic (what qt_nib_gemm does: tile width 256/192/128/64 by N; an asm-DMA kernel when row
 * strides are % 32 words and operands < 2 GiB, else the generic builtin-DMA kernel);
 * 20/21/22/23 = ping-pong kernel (64-byte stages, ring of 4, SIMD partners alternate load / compute
 * segments), tile 256x256 / 256x128 / 256x192 / 256x64, 24 = ping-pong 384x192; 6/7/8/15 = double-buffered pipelined kernel, tile 256x256 /
 * 256x128 / 256x64 / 256x192 (QT_ERR_ALIGNMENT if the contract does not hold); 5/9/10/16 = generic
 * kernel with the same tiles.  (161..166, stamped / ablated profiling kernels whose Y is not valid, exist only in
 * -DQT_PROFILING_VARIANTS builds of the library; the product build answers QT_ERR_UNSUPPORTED.) */
int qt_nib_gemm_variant(int variant, const uint32_t* Xn, int64_t ldxp, const uint32_t* Wn,
                        int64_t ldwp, const float* bias, float* Y, int64_t ldy, int64_t M,
                        int64_t N, int64_t K, qt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* QT_HIP_H */
