// Weight gradient of a stride-1 binarised / ternarised conv as matrix-core GEMMs (training path):
//
//     dW[co, ci, kh, kw] = sum_{n, oy, ox} g[n, co, oy, ox] * xpad[n, ci, oy + kh, ox + kw]
//
// (what torch.nn.grad.conv2d_weight computes behind layers/binary_layers.py:105's F.conv2d; functions/binary_connect.py:141-143).
// The contraction index is the output position, so both operands are laid out K-MAJOR over one position space
//     q = (y * N + n) * Wq + x          (Wq = padded row pitch, a multiple of 8 >= W + 2 pw)
// in which a tap (kh, kw) is a CONSTANT shift of the activation operand: q -> q + kh * N * Wq + kw.  Then
//     dW[:, :, kh, kw] = G' . Xs(kh, kw)^T
//       G'  [3 Cout rows] : the exact bf16 split hi / mid / lo of g (row t * Cout + co; zero where x >= Wo: those positions
//                           are not outputs), so every product is exact and the three row blocks are summed afterwards;
//       Xs  [Cin rows]    : the +-1 / 0 activation as bf16; the kh shift is a 16-byte-aligned pointer offset, the kw shift is
//                           not (2 bytes per position), so kw_count pre-shifted copies are written: copy j holds xT[q + j];
// and the whole gradient is ONE launch of qt_bf16_gemm_taps (mfma_gemm.hip): blockIdx.y = (tap, K slice), partial results
// [tap][slice][3 Cout][Cin] fp32, reduced (slices, the three terms) and scattered to [Cout][Cin][kh][kw] by qt_wgrad_reduce_f32,
// which also applies the straight-through mask 1[|W| <= thr] of the weight quantiser (functions/binary_connect.py:31-38).
// All three kernels here are HBM-bound elementwise passes.
#include "qt_common.h"

namespace {

__device__ __forceinline__ uint32_t wg_bf16_rn_bits(float f) {  // round-to-nearest-even, NaN kept quiet
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float wg_bf16_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// one work item = 8 consecutive positions of one output channel: three 16-byte stores (hi / mid / lo rows)
__global__ __launch_bounds__(256) void wgrad_pack_grad_kernel(const float* __restrict__ g, int64_t sn, int64_t sc, int64_t sh_,
                                                              int64_t sw, int N, int Cout, int Ho, int Wo, int Wq,
                                                              uint16_t* __restrict__ A, int64_t lda) {
    const int64_t chunks = lda >> 3;
    const int64_t total = (int64_t)Cout * chunks;
    const int64_t row_elems = (int64_t)N * Wq;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t co = t / chunks, q0 = (t - co * chunks) << 3;
        const int64_t y = q0 / row_elems, rem = q0 - y * row_elems;
        const int n = (int)(rem / Wq), x0 = (int)(rem - (int64_t)n * Wq);
        uint32_t h[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        if (y < Ho && x0 < Wo) {
            const float* src = g + (int64_t)n * sn + co * sc + y * sh_ + (int64_t)x0 * sw;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (x0 + i >= Wo) break;
                const float v = src[(int64_t)i * sw];
                const uint32_t a = wg_bf16_rn_bits(v);
                const float r1 = v - wg_bf16_to_f32(a);
                const uint32_t b = wg_bf16_rn_bits(r1);
                const float r2 = r1 - wg_bf16_to_f32(b);
                const uint32_t c = wg_bf16_rn_bits(r2);
                const int sh = (i & 1) * 16;
                h[0][i >> 1] |= a << sh;
                h[1][i >> 1] |= b << sh;
                h[2][i >> 1] |= c << sh;
            }
        }
#pragma unroll
        for (int s = 0; s < 3; ++s)
            *reinterpret_cast<uint4*>(A + ((int64_t)s * Cout + co) * lda + q0) = make_uint4(h[s][0], h[s][1], h[s][2], h[s][3]);
    }
}

// one work item = 8 consecutive positions of one input channel, all kw_count shifted copies
__global__ __launch_bounds__(256) void wgrad_pack_act_kernel(const float* __restrict__ x, int64_t sn, int64_t sc, int64_t sh_,
                                                             int64_t sw, int N, int Cin, int H, int W, int ph, int pw,
                                                             int Hp, int Wq, int kw_count, float x_scale,
                                                             uint16_t* __restrict__ B, int64_t ldb, int64_t copy_elems) {
    const int64_t chunks = ldb >> 3;
    const int64_t total = (int64_t)Cin * chunks;
    const int64_t row_elems = (int64_t)N * Wq;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ci = t / chunks, q0 = (t - ci * chunks) << 3;
        const int64_t y = q0 / row_elems, rem = q0 - y * row_elems;
        const int n = (int)(rem / Wq), x0 = (int)(rem - (int64_t)n * Wq);
        uint32_t v[16];                                       // bf16 bits of positions x0 .. x0 + 15 of this (y, n) row
        const int yy = (int)y - ph;
        const bool row_ok = y < Hp && yy >= 0 && yy < H;
        const float* src = x + (int64_t)n * sn + ci * sc + (int64_t)yy * sh_;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int xx = x0 + i - pw;
            float f = 0.0f;
            if (row_ok && i < 8 + kw_count - 1 && x0 + i < Wq && xx >= 0 && xx < W) f = src[(int64_t)xx * sw] * x_scale;
            v[i] = wg_bf16_rn_bits(f);
        }
        for (int j = 0; j < kw_count; ++j) {
            uint32_t o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint32_t lo = 0, hi = 0;
#pragma unroll
                for (int s = 0; s < 8; ++s) {                 // select v[2i + j], v[2i + 1 + j] without dynamic indexing
                    if (s == j) { lo = v[2 * i + s]; hi = v[2 * i + 1 + s]; }
                }
                o[i] = lo | (hi << 16);
            }
            *reinterpret_cast<uint4*>(B + (int64_t)j * copy_elems + ci * ldb + q0) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}


// ---- channels-last sources (stride_c == 1: what the conv kernels of this library produce): LDS-tiled transposes ---------
// One block = one (y, n) row x 64 channels x 32 positions: loads are 256-byte channel runs, stores 64-byte position runs.
__device__ __forceinline__ void wg_split3(float v, uint32_t& a, uint32_t& b, uint32_t& c) {
    a = wg_bf16_rn_bits(v);
    const float r1 = v - wg_bf16_to_f32(a);
    b = wg_bf16_rn_bits(r1);
    c = wg_bf16_rn_bits(r1 - wg_bf16_to_f32(b));
}

__global__ __launch_bounds__(256) void wgrad_pack_grad_nhwc_kernel(const float* __restrict__ g, int64_t sn, int64_t sh_, int64_t sw,
                                                                   int N, int Cout, int Ho, int Wo, int Wq,
                                                                   uint16_t* __restrict__ A, int64_t lda) {
    __shared__ float tile[32][65];
    const int xb = blockIdx.x * 32, c0 = blockIdx.y * 64;
    const int row = blockIdx.z;                       // y * N + n
    const int y = row / N, n = row - y * N;
    const int tid = threadIdx.x;
    {
        const int c = tid & 63, p0 = tid >> 6;
#pragma unroll
        for (int p = p0; p < 32; p += 4) {
            const int x = xb + p;
            float v = 0.0f;
            if (x < Wo && c0 + c < Cout) v = g[(int64_t)n * sn + (int64_t)y * sh_ + (int64_t)x * sw + c0 + c];
            tile[p][c] = v;
        }
    }
    __syncthreads();
    const int c = tid >> 2, ch = tid & 3;
    if (c0 + c >= Cout || xb + ch * 8 >= Wq) return;
    uint32_t h[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        uint32_t a, b, d;
        wg_split3(tile[ch * 8 + i][c], a, b, d);
        const int sh = (i & 1) * 16;
        h[0][i >> 1] |= a << sh;
        h[1][i >> 1] |= b << sh;
        h[2][i >> 1] |= d << sh;
    }
    const int64_t q0 = (int64_t)row * Wq + xb + ch * 8;
#pragma unroll
    for (int s = 0; s < 3; ++s)
        *reinterpret_cast<uint4*>(A + ((int64_t)s * Cout + c0 + c) * lda + q0) = make_uint4(h[s][0], h[s][1], h[s][2], h[s][3]);
}

__global__ __launch_bounds__(256) void wgrad_pack_act_nhwc_kernel(const float* __restrict__ x, int64_t sn, int64_t sh_, int64_t sw,
                                                                  int N, int Cin, int H, int W, int ph, int pw, int Wq,
                                                                  int kw_count, float x_scale, uint16_t* __restrict__ B,
                                                                  int64_t ldb, int64_t copy_elems) {
    __shared__ uint16_t tile[40][66];
    const int xb = blockIdx.x * 32, c0 = blockIdx.y * 64;
    const int row = blockIdx.z;                       // y * N + n over the PADDED rows
    const int y = row / N, n = row - y * N;
    const int yy = y - ph;
    const int tid = threadIdx.x;
    {
        const int c = tid & 63, p0 = tid >> 6;
        const bool row_ok = yy >= 0 && yy < H && c0 + c < Cin;
#pragma unroll
        for (int p = p0; p < 40; p += 4) {
            const int xq = xb + p, xx = xq - pw;
            float v = 0.0f;
            if (row_ok && xq < Wq && xx >= 0 && xx < W) v = x[(int64_t)n * sn + (int64_t)yy * sh_ + (int64_t)xx * sw + c0 + c] * x_scale;
            tile[p][c] = (uint16_t)wg_bf16_rn_bits(v);
        }
    }
    __syncthreads();
    const int c = tid >> 2, ch = tid & 3;
    if (c0 + c >= Cin || xb + ch * 8 >= Wq) return;
    const int64_t q0 = (int64_t)row * Wq + xb + ch * 8;
    for (int j = 0; j < kw_count; ++j) {
        uint32_t o[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o[i] = (uint32_t)tile[ch * 8 + 2 * i + j][c] | ((uint32_t)tile[ch * 8 + 2 * i + 1 + j][c] << 16);
        *reinterpret_cast<uint4*>(B + (int64_t)j * copy_elems + (int64_t)(c0 + c) * ldb + q0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// zero columns [from, ld) of every row (the K padding behind the last position)
__global__ __launch_bounds__(256) void wgrad_zero_tail_kernel(uint16_t* __restrict__ P, int64_t rows, int64_t ld, int64_t from) {
    const int64_t chunks = (ld - from) >> 3;
    const int64_t total = rows * chunks;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / chunks, c = t - r * chunks;
        *reinterpret_cast<uint4*>(P + r * ld + from + c * 8) = make_uint4(0, 0, 0, 0);
    }
}

// one work item = one (tap, co, ci): sums the K slices and the hi / mid / lo row blocks, applies the STE mask
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ C, int64_t ldc, int64_t z_stride, int taps,
                                                           int nslice, int Cout, int Cin, const float* __restrict__ weight,
                                                           float thr, float out_scale, int accumulate, float* __restrict__ dW) {
    const int64_t total = (int64_t)taps * Cout * Cin;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(t % Cin);
        const int64_t r = t / Cin;
        const int co = (int)(r % Cout), tap = (int)(r / Cout);
        float s = 0.0f;
        for (int sl = 0; sl < nslice; ++sl) {
            const float* base = C + ((int64_t)tap * nslice + sl) * z_stride + ci;
            // lo + mid first: the small terms are added before they meet the large one
            s += (base[((int64_t)2 * Cout + co) * ldc] + base[((int64_t)Cout + co) * ldc]) + base[(int64_t)co * ldc];
        }
        const int64_t o = ((int64_t)co * Cin + ci) * taps + tap;
        s *= out_scale;
        if (weight && !(fabsf(weight[o]) <= thr)) s = 0.0f;   // STE of the weight quantiser; NaN weights pass nothing
        dW[o] = accumulate ? dW[o] + s : s;
    }
}

}  // namespace

extern "C" {

int qt_wgrad_pack_grad_f32(const float* g, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                           int64_t Cout, int64_t Ho, int64_t Wo, int64_t Wq, uint16_t* A, int64_t lda, qt_stream_t stream) {
    if (N <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0 || !g || !A) return QT_ERR_INVALID_ARG;
    if (Wq < Wo || (Wq & 7) || (lda & 63) || lda < Ho * N * Wq || !qt_aligned16(A)) return QT_ERR_ALIGNMENT;
    if (N * Wq >= (1ll << 31) || Cout > INT32_MAX || Ho * N > 65535 * 32) return QT_ERR_UNSUPPORTED;
    const int64_t ktot = Ho * N * Wq;
    if (stride_c == 1 && Ho * N <= 65535 && (Cout + 63) / 64 <= 65535) {
        hipLaunchKernelGGL(wgrad_pack_grad_nhwc_kernel, dim3((unsigned)((Wq + 31) / 32), (unsigned)((Cout + 63) / 64), (unsigned)(Ho * N)),
                           dim3(256), 0, (hipStream_t)stream, g, stride_n, stride_h, stride_w, (int)N, (int)Cout, (int)Ho, (int)Wo,
                           (int)Wq, A, lda);
        if (lda > ktot)
            hipLaunchKernelGGL(wgrad_zero_tail_kernel, dim3(qt_stream_grid((3 * Cout * ((lda - ktot) >> 3) + 255) / 256)), dim3(256), 0,
                               (hipStream_t)stream, A, 3 * Cout, lda, ktot);
        return qt_check_launch();
    }
    const int grid = qt_stream_grid((Cout * (lda >> 3) + 255) / 256);
    hipLaunchKernelGGL(wgrad_pack_grad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, stride_n, stride_c, stride_h,
                       stride_w, (int)N, (int)Cout, (int)Ho, (int)Wo, (int)Wq, A, lda);
    return qt_check_launch();
}

int qt_wgrad_pack_act_f32(const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                          int64_t Cin, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t Wq, int64_t kw_count, float x_scale,
                          uint16_t* B, int64_t ldb, int64_t copy_elems, qt_stream_t stream) {
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || ph < 0 || pw < 0 || !x || !B) return QT_ERR_INVALID_ARG;
    if (kw_count < 1 || kw_count > 8) return QT_ERR_UNSUPPORTED;
    if (Wq < W + 2 * pw || (Wq & 7) || (ldb & 63) || (copy_elems & 7) || copy_elems != Cin * ldb || !qt_aligned16(B))
        return QT_ERR_ALIGNMENT;
    if (N * Wq >= (1ll << 31)) return QT_ERR_UNSUPPORTED;
    const int64_t Hp = H + 2 * ph, ptot = Hp * N * Wq;
    if (stride_c == 1 && Hp * N <= 65535 && (Cin + 63) / 64 <= 65535 && ldb >= ptot) {
        hipLaunchKernelGGL(wgrad_pack_act_nhwc_kernel, dim3((unsigned)((Wq + 31) / 32), (unsigned)((Cin + 63) / 64), (unsigned)(Hp * N)),
                           dim3(256), 0, (hipStream_t)stream, x, stride_n, stride_h, stride_w, (int)N, (int)Cin, (int)H, (int)W,
                           (int)ph, (int)pw, (int)Wq, (int)kw_count, x_scale, B, ldb, copy_elems);
        if (ldb > ptot)   // the kw_count copies are kw_count * Cin rows of pitch ldb (copy_elems == Cin * ldb is required below)
            hipLaunchKernelGGL(wgrad_zero_tail_kernel, dim3(qt_stream_grid((kw_count * Cin * ((ldb - ptot) >> 3) + 255) / 256)),
                               dim3(256), 0, (hipStream_t)stream, B, kw_count * Cin, ldb, ptot);
        return qt_check_launch();
    }
    const int grid = qt_stream_grid((Cin * (ldb >> 3) + 255) / 256);
    hipLaunchKernelGGL(wgrad_pack_act_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, stride_n, stride_c, stride_h,
                       stride_w, (int)N, (int)Cin, (int)H, (int)W, (int)ph, (int)pw, (int)(H + 2 * ph), (int)Wq, (int)kw_count, x_scale,
                       B, ldb, copy_elems);
    return qt_check_launch();
}

int qt_wgrad_reduce_f32(const float* partial, int64_t ldc, int64_t z_stride, int64_t taps, int64_t nslice, int64_t Cout,
                        int64_t Cin, const float* weight, float ste_threshold, float out_scale, int accumulate, float* dW,
                        qt_stream_t stream) {
    if (taps <= 0 || nslice <= 0 || Cout <= 0 || Cin <= 0 || !partial || !dW || ldc < Cin || z_stride < 3 * Cout * ldc)
        return QT_ERR_INVALID_ARG;
    if (taps * Cout * Cin >= (1ll << 40)) return QT_ERR_UNSUPPORTED;
    const int grid = qt_stream_grid((taps * Cout * Cin + 255) / 256);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, partial, ldc, z_stride, (int)taps,
                       (int)nslice, (int)Cout, (int)Cin, weight, ste_threshold, out_scale, accumulate, dW);
    return qt_check_launch();
}

}  // extern "C"
