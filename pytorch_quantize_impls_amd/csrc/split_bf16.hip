// bf16 "triple planes" for the real-valued-activation path (include/qt_hip.h):
//   activations: every fp32 x is split exactly into three bf16 terms  x = hi + mid + lo
//       hi  = bf16_rn(x),  mid = bf16_rn(x - hi),  lo = bf16_rn((x - hi) - mid)
//     (each subtraction is exact in fp32; three 8-bit significands cover the 24-bit fp32 significand),
//     stored as consecutive triples: element 3k+s of a row is term s of x[k] (optionally x[k]*alpha[k],
//     the XNOR-Net per-input-feature scale, folded in before the split: fl(x*alpha) * (+-1) is exactly the
//     product fl(x * (+-alpha)) the reference forms).
//   weights: the quantised value q in {-1, 0, +1} (safeSign / ternary / torch.sign) as bf16, replicated
//     three times (3k+s -> q[k]).
// A bf16 MFMA GEMM over K3 = 3K of these planes then equals the fp32 GEMM of x with the quantised weight
// up to fp32 accumulation order.  Row stride in bytes is a multiple of 16 (128 for GEMM operands), pad = 0.
// HBM-bound elementwise kernels: 4 B in, 6 B out per element.
#include "qt_common.h"

namespace {

__device__ __forceinline__ uint32_t bf16_rn_bits(float f) {  // round-to-nearest-even, NaN kept quiet
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

// mode: 0 = activation split, 1 = safeSign weight, 2 = ternary weight, 3 = torch.sign weight (0 -> 0)
template <int MODE>
__global__ __launch_bounds__(256) void triple_kernel(const float* __restrict__ x, int64_t ldx,
                                                     const float* __restrict__ alpha,
                                                     uint16_t* __restrict__ out, int64_t ld_elems,
                                                     int64_t rows, int64_t K) {
    const int64_t pairs_per_row = ld_elems / 6;          // one work item = 2 elements = 6 bf16 = 12 B
    const int64_t tail_words = (ld_elems - pairs_per_row * 6) / 2;  // leftover 32-bit words to zero
    const int64_t total = rows * (pairs_per_row + 1);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / (pairs_per_row + 1), p = t - row * (pairs_per_row + 1);
        uint32_t* orow = reinterpret_cast<uint32_t*>(out + row * ld_elems);
        if (p == pairs_per_row) {  // zero the (< 12-byte) remainder of the row
            for (int64_t w = 0; w < tail_words; ++w) orow[pairs_per_row * 3 + w] = 0;
            continue;
        }
        uint32_t h[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t k = p * 2 + e;
            if (k >= K) continue;
            float v = x[row * ldx + k];
            if (MODE == 0) {
                if (alpha) v *= alpha[k];
                const uint32_t a = bf16_rn_bits(v);
                const float r1 = v - bf16_bits_to_f32(a);
                const uint32_t b = bf16_rn_bits(r1);
                const float r2 = r1 - bf16_bits_to_f32(b);
                h[3 * e] = a; h[3 * e + 1] = b; h[3 * e + 2] = bf16_rn_bits(r2);
            } else {
                float q;
                if (MODE == 1) q = qt_safe_sign(v);
                else if (MODE == 2) q = qt_ternarize(v);
                else q = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
                const uint32_t qb = bf16_rn_bits(q);
                h[3 * e] = h[3 * e + 1] = h[3 * e + 2] = qb;
            }
        }
        orow[p * 3 + 0] = h[0] | (h[1] << 16);
        orow[p * 3 + 1] = h[2] | (h[3] << 16);
        orow[p * 3 + 2] = h[4] | (h[5] << 16);
    }
}

}  // namespace

extern "C" int qt_bf16x3_pack_f32(const float* x, int64_t ldx, const float* alpha, uint16_t* out,
                                  int64_t ld_bytes, int64_t rows, int64_t K, int mode, qt_stream_t stream) {
    if (rows < 0 || K < 0 || ldx < K || mode < 0 || mode > 3) return QT_ERR_INVALID_ARG;
    if (rows == 0) return QT_OK;
    if (!out || (!x && K > 0)) return QT_ERR_INVALID_ARG;
    if (ld_bytes < 6 * K || (ld_bytes & 15) || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    if (ld_bytes == 0) return QT_OK;
    const int64_t ld_elems = ld_bytes / 2;
    const int64_t total = rows * (ld_elems / 6 + 1);
    const int grid = qt_stream_grid((total + 255) / 256);
#define QT_LAUNCH(M) hipLaunchKernelGGL((triple_kernel<M>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, alpha, out, ld_elems, rows, K)
    switch (mode) {
        case 0: QT_LAUNCH(0); break;
        case 1: QT_LAUNCH(1); break;
        case 2: QT_LAUNCH(2); break;
        default: QT_LAUNCH(3); break;
    }
#undef QT_LAUNCH
    return qt_check_launch();
}
