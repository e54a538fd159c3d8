// Per-tap scaled implicit-GEMM convs: the XNOR-Net family on the matrix cores.
//
//   XNORConv2d (functions/xnor_connect.py:135-146, layers/xnor_layers.py:36-69):  y = conv2d(x, sign(W) * alpha),
//   alpha = mean(|W|, dim = [0, 1], keepdim) -> [1, 1, kh, kw]: ONE scale per filter tap.
//
// For +-1 activations (every layer behind a BinaryConnect) the conv factorises as  y = sum_taps alpha[i, j] * D[i, j]  with D the
// exact integer contraction over the input channels of one tap.  The implicit-GEMM kernel walks K tap-major,
// so the tap sums are formed on the fp4 matrix cores exactly as for BinConv2d and combined in Horner form on the accumulators
// (mfma_gemm_kernel.h, ElemFp4Taps): one fp4 pass + one VALU multiply per accumulator register per tap, instead of the six
// bf16 passes of a real x real conv.  The same holds for the gradient w.r.t. the input (xnor_connect.py:154-155): a real-valued
// gradient (two fp16 terms) against the flipped sign(W), alpha per flipped tap (ElemF16Taps).
//
// Entry points (include/qt_hip.h): qt_conv2d_implicit_taps (fp32 result), qt_conv2d_implicit_taps_bits / _nib (inference
// fusion: BatchNorm-threshold bits, or the next conv's nibble plane), qt_xnor_tap_prep_f32 (alpha + the Horner tables of a weight).
#include "mfma_gemm_kernel.h"
#include "xnor_alpha.h"

namespace {

// ---- alpha[c] = mean_r |W[r, c]|: the shared column-sum order of xnor_alpha.h (tap_abs_partial_kernel + tap_alpha_final) ----

// tables[0 .. T]      forward  : [1, rho_1 .. rho_{T-1}, a'_{T-1}],   rho_t = a'_{t-1} / a'_t
// tables[T+1 .. 2T+1] flipped  : the same for the reversed tap order (grad_x convolves with the flipped kernel)
// a' = alpha with zeros replaced by the previous non-zero entry (a tap whose alpha is 0 has all-zero weights: D_t = 0, any factor
// is right), leading zeros by the first non-zero one (or 1).
__global__ __launch_bounds__(1024) void tap_tables_kernel(const float* __restrict__ work, int nblk, float rows, const float* __restrict__ alpha_in,
                                                          int T, float* __restrict__ alpha, float* __restrict__ tables) {
    __shared__ float a[1024];
    __shared__ float stage[TAP_FINAL_STAGE];
    const int t = threadIdx.x;
    float s = 0.0f;
    if (!alpha_in) s = tap_alpha_final(work, nblk, T, t, rows, stage);        // (block-cooperative: every thread calls it)
    if (t < T) {
        if (alpha_in) s = alpha_in[t];
        a[t] = s;
        if (alpha) alpha[t] = s;
    }
    __syncthreads();
    if (t == 0) {
        for (int dir = 0; dir < 2; ++dir) {
            float* tab = tables + dir * (T + 1);
            float first = 1.0f;
            for (int i = 0; i < T; ++i) {
                const float v = a[dir ? T - 1 - i : i];
                if (v != 0.0f) { first = v; break; }
            }
            float prev = first;
            tab[0] = 1.0f;
            for (int i = 0; i < T; ++i) {
                float v = a[dir ? T - 1 - i : i];
                if (!(v != 0.0f)) v = prev;            // zero (or NaN compares unequal: kept, poisons the result like upstream)
                if (i > 0) tab[i] = prev / v;
                prev = v;
            }
            tab[T] = prev;
        }
    }
}

template <class E>
int dispatch_taps(const uint32_t* P, const uint32_t* Wmat, int64_t ldwp, const float* bias, float scale, const float* scale_dev,
                  float* Y, int64_t ldy, int64_t M, int64_t Cout, int64_t K, int64_t kwords, bool valid, qt_stream_t stream,
                  const ConvArgs& cg, const EpiArgs& epi) {
#define QT_TAPS(...) return launch_cfg<__VA_ARGS__>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi)
    // Tile widths: the in-place multiply needs the accumulators in VALU-addressable registers next to the fragments (a wave of an
    // 8-wave workgroup owns 256 registers, arch + acc together).  256x256 (128 accumulator registers) fits its MAIN LOOP in them —
    // the ~80 spilled dwords are prologue / epilogue values — and is what Cout = 768 / 256 want (AlexNet conv4: 163 -> 137 us,
    // conv5: 60 -> 41 us against 256x192 / 256x128 tiles); the 144-register 384x192 tile of the un-scaled conv does not.
    int tn = 256;
    {
        int64_t best = (Cout + 255) / 256 * 256;
        for (int c : {192, 128, 64})
            if ((Cout + c - 1) / c * c < best) { tn = c; best = (Cout + c - 1) / c * c; }
    }
    const int64_t tiles = ((M + 255) / 256) * ((Cout + tn - 1) / tn);
    const bool long_k = kwords * 4 >= 2048 && !(ldwp & 127);
    if (valid) {
        // small M (small-batch inference, late layers): the same rules as the un-scaled conv (mfma_gemm.hip, QT_CONV)
        if (long_k && !epi.d2s_cout && (M <= 4096 || (tiles < 256 && (M / 64) * Cout * kwords * 4 <= (256ll << 20)))) {
            if (M > 4096 && ((M + 127) / 128) * ((Cout + 127) / 128) >= 200) QT_TAPS(ConvV128x128<E>);
            if (((M + 127) / 128) * ((Cout + 63) / 64) >= 200) QT_TAPS(ConvV128x64<E>);
            QT_TAPS(ConvVSkinny<E>);
        }
        if (tn == 256) QT_TAPS(ConvVPP256<E>);
        if (tn == 192 && prefer_384_rows(M, Cout)) QT_TAPS(ConvVPP192<E>);
        if (tn == 192) QT_TAPS(ConvVPP256x192<E>);
        if (tn == 128) QT_TAPS(ConvVPP128<E>);
        QT_TAPS(ConvV64<E>);
    }
    if (!epi.alpha && epi.mode == 0 && tiles < 200) {
        if (((M + 127) / 128) * ((Cout + 127) / 128) < 200 && long_k) QT_TAPS(ConvSkinny<E>);
        QT_TAPS(Conv128x128<E>);
    }
    if (tn == 256) QT_TAPS(ConvPP256<E>);
    if (tn == 192 && prefer_384_rows(M, Cout)) QT_TAPS(ConvPP192<E>);
    if (tn == 192) QT_TAPS(ConvPP256x192<E>);
    if (tn == 128) QT_TAPS(ConvPP128<E>);
    QT_TAPS(Conv64<E>);
#undef QT_TAPS
}

int conv_taps_impl(int elem, const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw, int64_t kh, int64_t kw, int64_t sh,
                   int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldwp, const float* bias,
                   float scale, const float* scale_dev, const float* tap_rho, float* Y, int64_t ldy, int64_t Cout, qt_stream_t stream,
                   EpiArgs epi) {
    if (elem != 0 && elem != 3) return QT_ERR_UNSUPPORTED;        // fp4 nibble planes / fp16 pair planes
    if (!tap_rho) return QT_ERR_INVALID_ARG;
    if (Cw <= 0 || (Cw & 7)) return QT_ERR_ALIGNMENT;             // a tap = whole 32-byte MFMA k-steps
    ConvArgs cg;
    bool valid;
    int64_t M, K, kwords;
    const int rc = conv_prepare(elem, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, Y, ldy, Cout, epi, 0, 0, cg, valid,
                                M, K, kwords);
    if (rc != QT_OK) return rc > 0 ? QT_OK : rc;
    epi.tap_rho = tap_rho;
    epi.tap_ksteps = (int)(Cw / 8);
    epi.ntaps = (int)(kh * kw);
    if (elem == 0) return dispatch_taps<ElemFp4Taps>(P, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, kwords, valid, stream, cg, epi);
    return dispatch_taps<ElemF16Taps>(P, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, kwords, valid, stream, cg, epi);
}

}  // namespace

extern "C" {

int qt_conv2d_implicit_taps(int elem, const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw, int64_t kh, int64_t kw,
                            int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat,
                            int64_t ldwp, const float* bias, float scale, const float* scale_dev, const float* tap_rho, float* Y,
                            int64_t ldy, int64_t Cout, qt_stream_t stream) {
    return conv_taps_impl(elem, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, bias, scale, scale_dev, tap_rho, Y, ldy,
                          Cout, stream, EpiArgs{});
}

int qt_conv2d_implicit_taps_bits(int elem, const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw, int64_t kh, int64_t kw,
                                 int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat,
                                 int64_t ldwp, const float* bias, float scale, const float* scale_dev, const float* tap_rho,
                                 const float* alpha, const float* beta, uint32_t* neg_plane, int64_t ldb, int64_t Cout,
                                 qt_stream_t stream) {
    if (!alpha || !beta) return QT_ERR_INVALID_ARG;
    if (ldb & 3) return QT_ERR_ALIGNMENT;
    EpiArgs epi;
    epi.alpha = alpha;
    epi.beta = beta;
    return conv_taps_impl(elem, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, bias, scale, scale_dev, tap_rho,
                          reinterpret_cast<float*>(neg_plane), ldb, Cout, stream, epi);
}

int qt_conv2d_implicit_taps_nib(int elem, const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw, int64_t kh, int64_t kw,
                                int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat,
                                int64_t ldwp, const float* bias, float scale, const float* scale_dev, const float* tap_rho,
                                const float* alpha, const float* beta, uint32_t* nib_plane, int64_t ldn, int64_t Cout,
                                int64_t out_halo_h, int64_t out_halo_w, qt_stream_t stream) {
    if (!alpha || !beta) return QT_ERR_INVALID_ARG;
    if (out_halo_h < 0 || out_halo_w < 0 || out_halo_h > 64 || out_halo_w > 64) return QT_ERR_INVALID_ARG;
    if ((ldn & 3) || !qt_aligned16(nib_plane)) return QT_ERR_ALIGNMENT;
    if (ldn != (Cout + 31) / 32 * 4) return QT_ERR_INVALID_ARG;          // every word of a pixel is written by a column block
    EpiArgs epi;
    epi.alpha = alpha;
    epi.beta = beta;
    epi.mode = 3;
    epi.ohy = (int)out_halo_h;
    epi.ohx = (int)out_halo_w;
    return conv_taps_impl(elem, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, bias, scale, scale_dev, tap_rho,
                          reinterpret_cast<float*>(nib_plane), ldn, Cout, stream, epi);
}

int qt_conv2d_implicit_taps_rows(const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw, int64_t kh, int64_t kw, int64_t sh,
                                 int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldwp,
                                 const float* bias, const float* tap_rho, const float* a_plane, float* Y, int64_t ldy, int64_t Cout,
                                 qt_stream_t stream) {
    if (!tap_rho || !a_plane) return QT_ERR_INVALID_ARG;
    if (Cw <= 0 || (Cw & 7)) return QT_ERR_ALIGNMENT;
    if (kh * kw > 48) return QT_ERR_UNSUPPORTED;                  // (kh kw + 1) x 256 floats of LDS beside the stage buffers
    ConvArgs cg;
    EpiArgs epi;
    bool valid;
    int64_t M, K, kwords;
    const int rc = conv_prepare(0, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, Y, ldy, Cout, epi, 0, 0, cg, valid, M,
                                K, kwords);
    if (rc != QT_OK) return rc > 0 ? QT_OK : rc;
    epi.tap_rho = tap_rho;
    epi.tap_ksteps = (int)(Cw / 8);
    epi.ntaps = (int)(kh * kw);
    epi.row_A = a_plane;
    epi.aH = (int)H; epi.aW = (int)W; epi.aph = (int)ph; epi.apw = (int)pw;
    // 256 x 128 tiles, double-buffered 128-byte stages, bounds-checked taps (the per-row factors need the accumulators in
    // VALU-addressable registers next to 16 factor registers: the 64-register accumulator tile)
    // (the 256 x 128 ping-pong tile measured slower here: AlexNet conv2 810 vs 752 us for the whole function, conv4 463 vs 444)
    return launch_cfg<Conv128<ElemFp4TapsRows>>(P, 0, Wmat, ldwp, bias, 1.0f, nullptr, Y, ldy, M, Cout, K, stream, cg, epi);
}

int64_t qt_xnor_tap_prep_work_floats(int64_t R, int64_t taps) {
    if (R <= 0 || taps <= 0 || taps > 1024) return 0;
    return (int64_t)tap_alpha_blocks(R, taps, nullptr) * taps;
}

int qt_xnor_tap_prep_f32(const float* w, int64_t R, int64_t taps, const float* alpha_in, float* work, float* alpha,
                         float* tables, qt_stream_t stream) {
    if (taps <= 0 || taps > 1024 || !tables) return QT_ERR_INVALID_ARG;
    int nblk = 0;
    if (!alpha_in) {
        if (!w || R <= 0 || !work) return QT_ERR_INVALID_ARG;
        int64_t rows_per_blk = 0;
        nblk = tap_alpha_blocks(R, taps, &rows_per_blk);
        hipLaunchKernelGGL(tap_abs_partial_kernel, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, w, R, (int)taps, rows_per_blk, work);
    }
    hipLaunchKernelGGL(tap_tables_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, work, nblk, (float)R,
                       alpha_in, (int)taps, alpha, tables);
    return qt_check_launch();
}

}  // extern "C"
