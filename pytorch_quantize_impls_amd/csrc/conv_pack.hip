// Packed-domain im2col for the quantised conv layers (BinConv2d / TerConv2d,
// reference layers/binary_layers.py:103-106, terner_layers.py:89-92).
//
// The activation arrives as an NHWC pixel plane P[N][H][W][Cw] of uint32 words (Cw words per pixel:
// nibble plane -> 8 channels per word, bit plane -> 32 channels per word; Cw % 4 == 0 so a pixel is a
// whole number of 16-byte chunks, pad channels are zero).  Row m = (n, ho, wo) of the im2col matrix A
// is the concatenation over the kh*kw taps of the Cw words of the tapped pixel, or ZERO words where the
// tap falls into the padding — with nibble planes that is exactly the reference's zero padding
// (fp4 0x0 = 0.0), so border pixels need no correction.  Words past kh*kw*Cw up to ldA are zero.
//
// Pure copy kernel (HBM-bound): one thread per 16-byte chunk of A, consecutive threads walk a row, so
// both the pixel reads (Cw*4 contiguous bytes) and the row writes are coalesced.
#include "qt_common.h"

namespace {

struct ConvGeom {
    int H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Ho, Wo;
};

__global__ __launch_bounds__(256) void im2col_words_kernel(const uint32_t* __restrict__ P,
                                                           uint32_t* __restrict__ A, int64_t ldA,
                                                           int64_t m_begin, int64_t m_count,
                                                           ConvGeom g) {
    const int64_t chunks_per_row = ldA / 4;
    const int64_t total = m_count * chunks_per_row;
    const int taps = g.kh * g.kw;
    const int64_t howo = (int64_t)g.Ho * g.Wo;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / chunks_per_row;
        const int q = (int)(t - r * chunks_per_row);
        const int64_t m = m_begin + r;
        const int wq = q * 4;
        const int tap = wq / g.Cw, cw = wq - tap * g.Cw;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (tap < taps) {
            const int64_t n = m / howo;
            const int rem = (int)(m - n * howo);
            const int ho = rem / g.Wo, wo = rem - ho * g.Wo;
            const int i = tap / g.kw, j = tap - i * g.kw;
            const int hi = ho * g.sh - g.ph + i * g.dh, wi = wo * g.sw - g.pw + j * g.dw;
            if (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W)
                v = *reinterpret_cast<const uint4*>(P + (((n * g.H + hi) * g.W + wi) * (int64_t)g.Cw + cw));
        }
        *reinterpret_cast<uint4*>(A + r * ldA + wq) = v;
    }
}

// Physical zero padding of an NHWC pixel plane (any element type: nibble / int8 code / bf16-triple pixels are whole
// 16-byte chunks and their zero bytes are the value 0): P[N][H][W][Cw] -> Q[N][H+2ph][W+2pw][Cw], border = 0.
// A padded conv on P is the un-padded conv on Q, which the implicit-GEMM kernels run without per-tap checks.
__global__ __launch_bounds__(256) void pad_pixel_plane_kernel(const uint32_t* __restrict__ P, uint32_t* __restrict__ Q,
                                                              int64_t N, int H, int W, int Cw, int ph, int pw) {
    const int Hp = H + 2 * ph, Wp = W + 2 * pw, cpp = Cw / 4;
    const int64_t total = N * Hp * Wp * cpp;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = t / cpp;
        const int c = (int)(t - pix * cpp);
        const int64_t n = pix / ((int64_t)Hp * Wp);
        const int rem = (int)(pix - n * Hp * Wp);
        const int y = rem / Wp - ph, x = rem % Wp - pw;
        uint4 v = make_uint4(0, 0, 0, 0);
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W)
            v = *reinterpret_cast<const uint4*>(P + (((n * H + y) * W + x) * (int64_t)Cw) + c * 4);
        *reinterpret_cast<uint4*>(Q + pix * Cw + c * 4) = v;
    }
}

// Zero only the border pixels of a halo plane Q[N][H+2hy][W+2hx][Cw] (the interior is written by a conv epilogue).
// Border pixel b of an image: the first hy*Wp (top rows), then per interior row the 2*hx side pixels, then the bottom.
__global__ __launch_bounds__(256) void zero_halo_kernel(uint32_t* __restrict__ Q, int64_t N, int H, int W, int Cw,
                                                        int hy, int hx) {
    const int Hp = H + 2 * hy, Wp = W + 2 * hx, cpp = Cw / 4;
    const int top = hy * Wp, side = 2 * hx * H, per_img = 2 * top + side;
    const int64_t total = N * per_img * cpp;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bp = t / cpp;
        const int c = (int)(t - bp * cpp);
        const int64_t n = bp / per_img;
        const int b = (int)(bp - n * per_img);
        int pix;
        if (b < top) pix = b;
        else if (b < top + side) {
            const int s = b - top, r = s / (2 * hx), k = s - r * 2 * hx;
            pix = (hy + r) * Wp + (k < hx ? k : W + k);
        } else pix = (hy + H) * Wp + (b - top - side);
        *reinterpret_cast<uint4*>(Q + (n * Hp * Wp + pix) * (int64_t)Cw + c * 4) = make_uint4(0, 0, 0, 0);
    }
}

}  // namespace

extern "C" int qt_zero_halo(uint32_t* Q, int64_t N, int64_t H, int64_t W, int64_t Cw, int64_t hy, int64_t hx,
                            qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || Cw <= 0 || hy < 0 || hx < 0) return QT_ERR_INVALID_ARG;
    if (N == 0 || (hy == 0 && hx == 0)) return QT_OK;
    if (!Q) return QT_ERR_INVALID_ARG;
    if ((Cw & 3) || !qt_aligned16(Q)) return QT_ERR_ALIGNMENT;
    if (H + 2 * hy > 32767 || W + 2 * hx > 32767 || Cw > (1 << 20)) return QT_ERR_UNSUPPORTED;
    const int64_t total = N * (2 * hy * (W + 2 * hx) + 2 * hx * H) * (Cw / 4);
    const int grid = qt_stream_grid((total + 255) / 256);
    hipLaunchKernelGGL(zero_halo_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, Q, N, (int)H, (int)W, (int)Cw,
                       (int)hy, (int)hx);
    return qt_check_launch();
}

extern "C" int qt_pad_pixel_plane(const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw, int64_t ph,
                                  int64_t pw, uint32_t* Q, qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || Cw <= 0 || ph < 0 || pw < 0) return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!P || !Q) return QT_ERR_INVALID_ARG;
    if ((Cw & 3) || !qt_aligned16(P) || !qt_aligned16(Q)) return QT_ERR_ALIGNMENT;
    if (H + 2 * ph > 32767 || W + 2 * pw > 32767 || Cw > (1 << 20)) return QT_ERR_UNSUPPORTED;
    const int64_t total = N * (H + 2 * ph) * (W + 2 * pw) * (Cw / 4);
    const int grid = qt_stream_grid((total + 255) / 256);
    hipLaunchKernelGGL(pad_pixel_plane_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, Q, N, (int)H, (int)W,
                       (int)Cw, (int)ph, (int)pw);
    return qt_check_launch();
}

extern "C" int qt_im2col_words(const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw,
                               int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw,
                               int64_t dh, int64_t dw, uint32_t* A, int64_t ldA, int64_t m_begin,
                               int64_t m_count, qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || Cw <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 ||
        dw <= 0 || ph < 0 || pw < 0 || m_begin < 0 || m_count < 0)
        return QT_ERR_INVALID_ARG;
    const int64_t Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1;
    const int64_t Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
    if (Ho <= 0 || Wo <= 0 || m_begin + m_count > N * Ho * Wo) return QT_ERR_INVALID_ARG;
    if (m_count == 0) return QT_OK;
    if (!P || !A || ldA < kh * kw * Cw) return QT_ERR_INVALID_ARG;
    if ((Cw & 3) || (ldA & 3) || !qt_aligned16(P) || !qt_aligned16(A)) return QT_ERR_ALIGNMENT;
    if (H > INT32_MAX / 4 || W > INT32_MAX / 4 || Cw * kh * kw > INT32_MAX / 8) return QT_ERR_UNSUPPORTED;
    ConvGeom g{(int)H, (int)W, (int)Cw, (int)kh, (int)kw, (int)sh, (int)sw, (int)ph, (int)pw,
               (int)dh, (int)dw, (int)Ho, (int)Wo};
    const int grid = qt_stream_grid((m_count * (ldA / 4) + 255) / 256);
    hipLaunchKernelGGL(im2col_words_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, P, A, ldA,
                       m_begin, m_count, g);
    return qt_check_launch();
}
