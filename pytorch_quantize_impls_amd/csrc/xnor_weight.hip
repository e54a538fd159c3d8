// XNOR-Net weight quantiser:  W_q = sign(W) * alpha,  alpha = mean(|W|) over the leading dimension(s).
//   XNORDense : alpha = mean(|W|, dim 0, keepdim) -> [1, K]        (functions/xnor_connect.py:112-113;
//               the reference ignores its `dim` argument and uses the module-global DIM = 0)
//   XNORConv2d: alpha = mean(|W|, dim [0,1], keepdim) -> [1,1,kh,kw] (functions/xnor_connect.py:140-141)
// Both are "column means" of a row-major [R, C] view of the weight (R = N, C = K  resp.  R = Cout*Cin,
// C = kh*kw), followed by sign(w) * alpha[c] with torch.sign semantics (0 -> 0, NaN -> NaN).
// HBM-bound, weights only (small): two passes over W.
#include "qt_common.h"

namespace {

// alpha[c] = (1/R) * sum_r |W[r, c]|.  One workgroup per 64-column strip; thread (ty, tx) walks rows
// ty, ty+4, ... of column c0+tx (coalesced 256-byte row segments), partial sums meet in LDS.
// Accumulation order differs from torch's reduction; the result enters a float tail anyway.
__global__ __launch_bounds__(256) void col_abs_mean_kernel(const float* __restrict__ w, int64_t ldw,
                                                           float* __restrict__ alpha, int64_t R,
                                                           int64_t C) {
    __shared__ float part[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + tx;
    float acc = 0.0f;
    if (c < C)
        for (int64_t r = ty; r < R; r += 4) acc += fabsf(w[r * ldw + c]);
    part[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c < C) alpha[c] = (part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx]) / (float)R;
}

__device__ __forceinline__ float torch_sign(float x) {
    return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : x);  // +-0 -> +-0, NaN -> NaN (times alpha stays NaN)
}

__global__ __launch_bounds__(256) void sign_scale_kernel(const float* __restrict__ w, int64_t ldw,
                                                         const float* __restrict__ alpha,
                                                         float* __restrict__ out, int64_t ldo, int64_t R,
                                                         int64_t C) {
    const int64_t total = R * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C, c = i - r * C;
        out[r * ldo + c] = torch_sign(w[r * ldw + c]) * alpha[c];
    }
}

// LinearXNOR on a PACKED +-1 activation (eval-mode inference: the activation exists only as sign bits): the operand of the fp16
// matrix-core GEMM is x[b, k] * alpha[k] = +-alpha[k] as a two-term fp16 pair — the (hi, lo) pair of alpha[k] / s, prepared once
// per weight version, with both signs flipped where the bit says -1 (exact).  One thread = one 32-bit word of a bit-plane row ->
// 32 pairs (128 bytes).
// perm_C > 0: the bit rows are a feature map flattened in (h, w, c) order (PackedActivation.flatten_hwc: bit hw * C + c) while the
// layer — and the pair table — count features in the NCHW order c * HW + hw the module graph flattens in: output feature k reads bit
// (k % HW) * C + k / HW, so the GEMM contracts in the same order, on the same operands, as for the un-packed activation (the two
// executions of a model then agree bit for bit; a real-valued sum depends on its order).
__global__ __launch_bounds__(256) void bits_alpha_pairs_kernel(const uint32_t* __restrict__ bits, int64_t ldb,
                                                               const uint32_t* __restrict__ apair, uint32_t* __restrict__ out,
                                                               int64_t ldo_words, int64_t rows, int64_t K, int64_t perm_C, int64_t perm_HW) {
    const int64_t wpr = ldo_words / 32;            // 32-pair groups per output row (row stride is a multiple of 128 bytes)
    const int64_t total = rows * wpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / wpr, g = i - r * wpr;
        uint32_t w = 0u;
        if (perm_C > 0) {
            for (int e = 0; e < 32; ++e) {
                const int64_t k = g * 32 + e;
                if (k < K) {
                    const int64_t c = k / perm_HW, hw = k - c * perm_HW, b = hw * perm_C + c;
                    w |= ((bits[r * ldb + (b >> 5)] >> (b & 31)) & 1u) << e;
                }
            }
        } else if (g < ldb) {
            w = bits[r * ldb + g];
        }
        uint4* o = reinterpret_cast<uint4*>(out + r * ldo_words + g * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            uint32_t v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t k = g * 32 + q * 4 + e;
                const uint32_t a = k < K ? apair[k] : 0u;                       // (hi | lo << 16) of alpha[k] / s
                v[e] = ((w >> (q * 4 + e)) & 1u) ? (a ^ 0x80008000u) : a;       // -1: flip both sign bits
            }
            o[q] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }
}

}  // namespace

extern "C" int qt_bits_alpha_pairs_f16x2(const uint32_t* bits, int64_t ldb, const uint32_t* alpha_pairs, uint32_t* out,
                                         int64_t ld_bytes, int64_t rows, int64_t K, int64_t perm_C, int64_t perm_HW,
                                         qt_stream_t stream) {
    if (rows < 0 || K < 0 || perm_C < 0 || perm_HW < 0 || (perm_C > 0 && perm_C * perm_HW != K)) return QT_ERR_INVALID_ARG;
    if (rows == 0 || K == 0) return QT_OK;
    if (!bits || !alpha_pairs || !out || ldb < (K + 31) / 32) return QT_ERR_INVALID_ARG;
    if ((ld_bytes & 127) || ld_bytes < 4 * K || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    const int grid = qt_stream_grid((rows * (ld_bytes / 128) + 255) / 256);
    hipLaunchKernelGGL(bits_alpha_pairs_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, bits, ldb, alpha_pairs, out,
                       ld_bytes / 4, rows, K, perm_C, perm_HW);
    return qt_check_launch();
}

extern "C" int qt_xnor_weight_f32(const float* w, int64_t ldw, float* alpha, float* wq, int64_t ldq,
                                  int64_t R, int64_t C, qt_stream_t stream) {
    if (R <= 0 || C <= 0) return (R == 0 || C == 0) ? QT_OK : QT_ERR_INVALID_ARG;
    if (!w || !alpha || ldw < C || (wq && ldq < C)) return QT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(col_abs_mean_kernel, dim3((unsigned)((C + 63) / 64)), dim3(256), 0,
                       (hipStream_t)stream, w, ldw, alpha, R, C);
    if (wq) {
        const int grid = qt_stream_grid((R * C + 1023) / 1024);
        hipLaunchKernelGGL(sign_scale_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, ldw, alpha,
                           wq, ldq, R, C);
    }
    return qt_check_launch();
}
