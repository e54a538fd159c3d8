// XNOR-Net weight quantiser:  W_q = sign(W) * alpha,  alpha = mean(|W|) over the leading dimension(s).
//   XNORDense : alpha = mean(|W|, dim 0, keepdim) -> [1, K]        (functions/xnor_connect.py:112-113;
//               the reference ignores its `dim` argument and uses the module-global DIM = 0)
//   XNORConv2d: alpha = mean(|W|, dim [0,1], keepdim) -> [1,1,kh,kw] (functions/xnor_connect.py:140-141)
// Both are "column means" of a row-major [R, C] view of the weight (R = N, C = K  resp.  R = Cout*Cin,
// C = kh*kw), followed by sign(w) * alpha[c] with torch.sign semantics (0 -> 0, NaN -> NaN).
// HBM-bound, weights only (small): two passes over W.
#include "qt_common.h"
#include "xnor_alpha.h"

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

// The same column sums for TALL matrices (XNORConv2d: R = Cout * Cin rows, C = kh * kw columns — one 64-column strip walked
// 110 592 rows on ONE workgroup: 8.8 ms for AlexNet conv2): rows cut into chunks of 256, one workgroup per (strip, chunk) writes its
// partial sums, a second launch adds the chunks of a column in ascending order (deterministic) and divides by R.  The partials
// live in the caller's wq buffer (R * C floats, overwritten by sign_scale_kernel afterwards): no workspace argument.
__global__ __launch_bounds__(256) void col_abs_part_kernel(const float* __restrict__ w, int64_t ldw, float* __restrict__ part, int64_t R,
                                                           int64_t C, int64_t rows_per_chunk) {
    __shared__ float sm[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + tx;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = r0 + rows_per_chunk < R ? r0 + rows_per_chunk : R;
    float acc = 0.0f;
    if (c < C)
        for (int64_t r = r0 + ty; r < r1; r += 4) acc += fabsf(w[r * ldw + c]);
    sm[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c < C) part[(int64_t)blockIdx.y * C + c] = sm[0][tx] + sm[1][tx] + sm[2][tx] + sm[3][tx];
}

__global__ __launch_bounds__(256) void col_abs_final_kernel(const float* __restrict__ part, float* __restrict__ alpha, int64_t chunks,
                                                            int64_t R, int64_t C) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float acc = 0.0f;
    for (int64_t k = 0; k < chunks; ++k) acc += part[k * C + c];
    alpha[c] = acc / (float)R;
}

// closing step of xnor_alpha.h's order for FEW columns (XNORConv2d weights): the alpha bits of qt_xnor_tap_prep_f32
__global__ __launch_bounds__(1024) void tap_alpha_final_kernel(const float* __restrict__ work, int nblk, float rows, int T,
                                                               float* __restrict__ alpha) {
    __shared__ float stage[TAP_FINAL_STAGE];
    const int t = threadIdx.x;
    const float s = tap_alpha_final(work, nblk, T, t, rows, stage);           // (block-cooperative: every thread calls it)
    if (t < T) alpha[t] = s;
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

// Four sign bits of features k0 .. k0 + 3 of a row whose bits are stored in (h, w, c) order while the features count (c, h, w):
// feature k = c * HW + hw sits at bit hw * C + c.  One 32-bit division per group (K < 2^31, checked by the entry points), the
// other three positions by increment (instead of a 64-bit division per bit: AlexNet's head 10.0 -> 8.8 us, the digit planes of
// fc1 / fc2 7.5 -> 7.1 us — small, latency-bound launches: sixteen features per thread with 16-byte stores measured 8.5 us).
__device__ __forceinline__ uint32_t perm_nibble(const uint32_t* __restrict__ rowbits, int64_t k0, int64_t K, int64_t perm_C,
                                                int64_t perm_HW) {
    const unsigned HW = (unsigned)perm_HW, C = (unsigned)perm_C;
    unsigned c = (unsigned)k0 / HW, hw = (unsigned)k0 - c * HW;
    uint32_t nib = 0u;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (k0 + e < K) {
            const unsigned b = hw * C + c;
            nib |= ((rowbits[b >> 5] >> (b & 31)) & 1u) << e;
        }
        if (++hw == HW) { hw = 0; ++c; }
    }
    return nib;
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


// ---- LinearXNOR on a packed +-1 activation, integer form -------------------------------------------------------------------------
// y[b, n] = sum_k x[b, k] alpha[k] s[n, k]  (x = +-1 bits, s = sign(W) in {-1, 0, +1}, alpha[k] = mean_n |W[n, k]| >= 0).  The fp16
// pair route treats +-alpha[k] as a real operand: 4 bytes per weight (the sign replicated for both terms) and 2 fp16 products
// per MAC.  Here alpha is FIXED POINT: A[k] = rint(alpha[k] / s), s a power of two with max A < 2^21, cut into three 7-bit digits
// A = d0 2^14 + d1 2^7 + d2 (0 <= d <= 127), so that x[b, k] d_j[k] is an int8 and
//     y[b, n] = s (2^14 P0 + 2^7 P1 + P2)[b, n],    P_j = (x d_j) . s^T     three EXACT integer GEMMs on the int8 matrix cores
// sharing one weight operand of 1 byte per weight (stacked along M: rows [j * rows + b]); the partial sums are exact integers
// whatever the summation order (tile shape, K split), so every execution of a model gives the same bits.  Error: alpha rounded
// to 2^-22 of max alpha (the fp16 pair: 2^-22 of each alpha; alpha is a mean over the output features — its entries are of one
// magnitude), the combination in fp64 rounds once.
//
// One thread = four consecutive features of one row: one 16-byte read of the digit table (d0 | d1 << 8 | d2 << 16 per feature, zero
// from K up to the padded row length), four sign bits, one dword of each of the three digit planes — table reads and plane
// writes are lane-contiguous.  perm_C > 0: see bits_alpha_pairs_kernel.
__global__ __launch_bounds__(256) void bits_alpha_digits_kernel(const uint32_t* __restrict__ bits, int64_t ldb,
                                                                const uint4* __restrict__ dtab4, uint32_t* __restrict__ out,
                                                                int64_t ldo_words, int64_t rows, int64_t K, int64_t perm_C, int64_t perm_HW) {
    const int64_t total = rows * ldo_words;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ldo_words, q = i - r * ldo_words, k0 = q * 4;
        const uint4 t4 = dtab4[q];
        uint32_t nib = 0u;                            // bit e: feature k0 + e is -1
        if (perm_C > 0) {
            nib = perm_nibble(bits + r * ldb, k0, K, perm_C, perm_HW);
        } else if ((k0 >> 5) < ldb) {
            nib = (bits[r * ldb + (k0 >> 5)] >> (k0 & 31)) & 0xFu;
        }
        const uint32_t t[4] = {t4.x, t4.y, t4.z, t4.w};
        uint32_t v[3] = {0u, 0u, 0u};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool neg = (nib >> e) & 1u;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int d = (int)((t[e] >> (8 * j)) & 0xffu);
                v[j] |= (uint32_t)(uint8_t)(int8_t)(neg ? -d : d) << (8 * e);
            }
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) out[((int64_t)j * rows + r) * ldo_words + q] = v[j];
    }
}

// y[b, n] = fp32(s * (2^14 S0 + 2^7 S1 + S2)) + bias[n],  S_j = sum over the K slices of P[z][j * rows + b][n]: the slice sums are
// exact integers below 2^24 (fp32 adds are exact), the combination is exact in fp64 (< 2^38), s is a power of two: ONE rounding
// before the bias, as if the whole dot product had been formed exactly.
__global__ __launch_bounds__(256) void digit_reduce_kernel(const float* __restrict__ P, int64_t ldp, int64_t slice_stride, int nslice,
                                                           const float* __restrict__ scale, const float* __restrict__ bias,
                                                           float* __restrict__ Y, int64_t ldy, int64_t rows, int64_t N) {
    const int64_t total = rows * N;
    const double s = (double)scale[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / N, n = i - b * N;
        float S[3] = {0.0f, 0.0f, 0.0f};
        for (int z = 0; z < nslice; ++z) {
            const float* p = P + (int64_t)z * slice_stride + b * ldp + n;
#pragma unroll
            for (int j = 0; j < 3; ++j) S[j] += p[(int64_t)j * rows * ldp];
        }
        const double t = (double)S[0] * 16384.0 + (double)S[1] * 128.0 + (double)S[2];
        Y[b * ldy + n] = (float)(t * s) + (bias ? bias[n] : 0.0f);
    }
}


// Small heads (N <= 32 outputs: the classifier layer): the same exact integer sum_k x[b, k] A[k] s[n, k] straight from the sign bits —
// one wave per (row, 8 outputs), lane = 4 consecutive features per step (one dword of each weight row's int8 codes, one 16-byte read
// of the digit table), |lane partial| < 2^21 K / 64 in int32, the wave's total in fp64 (exact), then the digit route's own last
// step fp32(total * s) + bias: BIT-IDENTICAL to the split-K GEMM + qt_digit_reduce_f32, without three launches whose tiles are
// empty at N = 10 (29 -> 6 us at 256 x 10 x 4096).
__global__ __launch_bounds__(256) void xnor_head_kernel(const uint32_t* __restrict__ bits, int64_t ldb, const uint4* __restrict__ dtab4,
                                                        const uint32_t* __restrict__ wc, int64_t ldw_words, const float* __restrict__ scale,
                                                        const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy, int64_t rows,
                                                        int64_t N, int64_t K, int64_t perm_C, int64_t perm_HW) {
    // workgroup = (one row, 8 outputs): its 4 waves stride the feature groups together (a wave per row walked K in 16 dependent
    // round trips on half the CUs: 20.8 us at AlexNet's head; this form: 4 trips on all of them)
    __shared__ double part[4][8];
    const int lane = threadIdx.x & 63;
    const int64_t r = blockIdx.x;
    const int n0 = blockIdx.y * 8;
    const int64_t kq = (K + 3) / 4;                  // groups of 4 features (table / weight rows are zero past K)
    int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t* wrow[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wrow[j] = wc + (int64_t)min((int64_t)(n0 + j), N - 1) * ldw_words;
#pragma unroll 2
    for (int64_t q = threadIdx.x; q < kq; q += 256) {
        const int64_t k0 = q * 4;
        const uint4 t4 = dtab4[q];
        uint32_t nib = 0u;
        if (perm_C > 0) {
            nib = perm_nibble(bits + r * ldb, k0, K, perm_C, perm_HW);
        } else {
            nib = (bits[r * ldb + (k0 >> 5)] >> (k0 & 31)) & 0xFu;
        }
        const uint32_t t[4] = {t4.x, t4.y, t4.z, t4.w};
        int xa[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int A = (int)(((t[e] & 0xffu) << 14) | (((t[e] >> 8) & 0xffu) << 7) | ((t[e] >> 16) & 0xffu));
            xa[e] = ((nib >> e) & 1u) ? -A : A;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t w4 = wrow[j][q];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j] += xa[e] * (int)(int8_t)(w4 >> (8 * e));
        }
    }
    const double s = (double)scale[0];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double v = (double)acc[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) part[threadIdx.x >> 6][j] = v;           // integers below 2^53: any summation order gives the same double
    }
    __syncthreads();
    if (threadIdx.x < 8 && n0 + threadIdx.x < N) {
        const int j = threadIdx.x;
        const double v = (part[0][j] + part[1][j]) + (part[2][j] + part[3][j]);
        Y[r * ldy + n0 + j] = (float)(v * s) + (bias ? bias[n0 + j] : 0.0f);
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
    const int64_t strips = (C + 63) / 64;
    if (wq && wq != w && ldw == C && ldq == C && C <= 1024 && R >= 1 && tap_alpha_blocks(R, C, nullptr) <= R) {
        // few contiguous columns (an XNORConv2d weight: R = Cout * Cin, C = kh * kw): the column-sum order qt_xnor_tap_prep_f32
        // uses (xnor_alpha.h), partials through the wq buffer — the image's alpha and the TapScales' alpha are the same bits
        int64_t rows_per_blk = 0;
        const int nblk = tap_alpha_blocks(R, C, &rows_per_blk);
        hipLaunchKernelGGL(tap_abs_partial_kernel, dim3(nblk), dim3(1024), 0, (hipStream_t)stream, w, R, (int)C, rows_per_blk, wq);
        hipLaunchKernelGGL(tap_alpha_final_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, wq, nblk, (float)R, (int)C, alpha);
    } else if (wq && wq != w && ldq == C && R >= 2048 && strips * 4 <= 256 && (R + 255) / 256 <= 65535) {
        // tall and narrow: chunked partial sums through the wq buffer (see col_abs_part_kernel)
        const int64_t chunks = (R + 255) / 256;
        hipLaunchKernelGGL(col_abs_part_kernel, dim3((unsigned)strips, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, w, ldw, wq, R,
                           C, (int64_t)256);
        hipLaunchKernelGGL(col_abs_final_kernel, dim3((unsigned)((C + 255) / 256)), dim3(256), 0, (hipStream_t)stream, wq, alpha, chunks, R, C);
    } else {
        hipLaunchKernelGGL(col_abs_mean_kernel, dim3((unsigned)strips), dim3(256), 0, (hipStream_t)stream, w, ldw, alpha, R, C);
    }
    if (wq) {
        const int grid = qt_stream_grid((R * C + 1023) / 1024);
        hipLaunchKernelGGL(sign_scale_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, ldw, alpha,
                           wq, ldq, R, C);
    }
    return qt_check_launch();
}

extern "C" int qt_bits_alpha_digits_i8(const uint32_t* bits, int64_t ldb, const uint32_t* digit_table, int8_t* out, int64_t ld_bytes,
                                       int64_t rows, int64_t K, int64_t perm_C, int64_t perm_HW, qt_stream_t stream) {
    if (rows < 0 || K < 0 || perm_C < 0 || perm_HW < 0 || (perm_C > 0 && perm_C * perm_HW != K)) return QT_ERR_INVALID_ARG;
    if (rows == 0 || K == 0) return QT_OK;
    if (K >= (1ll << 31)) return QT_ERR_UNSUPPORTED;                 // perm_nibble counts features in 32 bits
    if (!bits || !digit_table || !out || ldb < (K + 31) / 32) return QT_ERR_INVALID_ARG;
    if ((ld_bytes & 31) || ld_bytes < K || !qt_aligned16(out) || !qt_aligned16(digit_table)) return QT_ERR_ALIGNMENT;
    const int grid = qt_stream_grid((rows * (ld_bytes / 4) + 255) / 256);
    hipLaunchKernelGGL(bits_alpha_digits_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, bits, ldb,
                       reinterpret_cast<const uint4*>(digit_table), reinterpret_cast<uint32_t*>(out), ld_bytes / 4, rows, K, perm_C, perm_HW);
    return qt_check_launch();
}

extern "C" int qt_digit_reduce_f32(const float* partial, int64_t ldp, int64_t slice_stride, int64_t nslice, const float* scale_dev,
                                   const float* bias, float* Y, int64_t ldy, int64_t rows, int64_t N, qt_stream_t stream) {
    if (rows < 0 || N < 0 || nslice < 1 || nslice > 65535 || ldp < N || ldy < N || slice_stride < 3 * rows * ldp) return QT_ERR_INVALID_ARG;
    if (rows == 0 || N == 0) return QT_OK;
    if (!partial || !scale_dev || !Y) return QT_ERR_INVALID_ARG;
    const int grid = qt_stream_grid((rows * N + 255) / 256);
    hipLaunchKernelGGL(digit_reduce_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, partial, ldp, slice_stride, (int)nslice,
                       scale_dev, bias, Y, ldy, rows, N);
    return qt_check_launch();
}

extern "C" int qt_xnor_head_i8(const uint32_t* bits, int64_t ldb, const uint32_t* digit_table, const int8_t* wcodes, int64_t ldw_bytes,
                               const float* scale_dev, const float* bias, float* Y, int64_t ldy, int64_t rows, int64_t N, int64_t K,
                               int64_t perm_C, int64_t perm_HW, qt_stream_t stream) {
    if (rows < 0 || N < 0 || K < 0 || perm_C < 0 || perm_HW < 0 || (perm_C > 0 && perm_C * perm_HW != K) || ldy < N) return QT_ERR_INVALID_ARG;
    if (rows == 0 || N == 0) return QT_OK;
    if (!bits || !digit_table || !wcodes || !scale_dev || !Y || ldb < (K + 31) / 32) return QT_ERR_INVALID_ARG;
    if ((ldw_bytes & 15) || ldw_bytes < (K + 3) / 4 * 4 || !qt_aligned16(wcodes) || !qt_aligned16(digit_table)) return QT_ERR_ALIGNMENT;
    if (K >= (1ll << 16) || N > 65535 * 8) return QT_ERR_UNSUPPORTED;                  // lane partial < 2^21 * K / 64 < 2^31
    hipLaunchKernelGGL(xnor_head_kernel, dim3((unsigned)rows, (unsigned)((N + 7) / 8)), dim3(256), 0, (hipStream_t)stream, bits,
                       ldb, reinterpret_cast<const uint4*>(digit_table), reinterpret_cast<const uint32_t*>(wcodes), ldw_bytes / 4, scale_dev,
                       bias, Y, ldy, rows, N, K, perm_C, perm_HW);
    return qt_check_launch();
}
