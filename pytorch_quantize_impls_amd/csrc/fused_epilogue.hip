// Fused inference epilogue between binarised layers:
//     [MaxPool2d(k, s)] -> BatchNorm(eval) -> Hardtanh -> BinaryConnectDeterministic -> bit-pack
// (the chain models/Alexnet/Alexnet_Bin.py:14-17 and benchmark/BinaryNet/MLPBin.py:42-44 put between
// two binarised layers).  In eval mode BatchNorm is the per-channel affine t = x*alpha[c] + beta[c]
// (alpha = weight/sqrt(var+eps), beta = bias - mean*alpha: the fold ATen's CPU kernel itself performs),
// Hardtanh never changes the sign, and safeSign(t) = (t < 0 ? -1 : +1): the whole chain collapses to one
// comparison per element on the max-pooled conv output.  The kernel reads the fp32 NHWC tensor ONCE and
// writes only the sign bit plane (1/32 of a fp32 tensor) — none of the four intermediate fp32 tensors
// the unfused chain materialises exists any more.
//
// Layout: x is NHWC fp32 [N][H][W][C] (C % 4 == 0), out is the NHWC pixel bit plane [N*Ho*Wo][ldp].
// One work item = 4 consecutive channels of one output pixel; 8 adjacent lanes = one 32-channel word.
// HBM-bound: algorithmic bytes = 4*N*H*W*C (read) + N*Ho*Wo*C/8 (write).
#include <algorithm>
#include "qt_common.h"

namespace {

__device__ __forceinline__ uint32_t or_reduce8(uint32_t v) {
    v |= __shfl_xor(v, 1);
    v |= __shfl_xor(v, 2);
    v |= __shfl_xor(v, 4);
    return v;
}

__device__ __forceinline__ uint32_t pas_spread8(uint32_t b) {      // the 8 bits of a byte to bit 0 of 8 nibbles
    uint32_t t = b & 0xFFu;
    t = (t | (t << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return t;
}

__global__ __launch_bounds__(256) void pool_affine_sign_pack_kernel(
    const float* __restrict__ x, const float* __restrict__ alpha, const float* __restrict__ beta,
    uint32_t* __restrict__ plane, int64_t ldp, int64_t N, int H, int W, int C, int pk, int ps, int Ho,
    int Wo, int pre_relu, uint32_t* __restrict__ nibplane, int64_t ldn) {
    const int64_t slots_per_pixel = ldp * 8;  // float4 slots per output pixel incl. pad (pad -> bit 0)
    const int64_t total = N * Ho * Wo * slots_per_pixel;  // multiple of 32 (ldp % 4 == 0)
    const int c4max = C / 4;
    const int lane8 = threadIdx.x & 7;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
         s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = s / slots_per_pixel;
        const int slot = (int)(s - pix * slots_per_pixel);
        uint32_t nib = 0;
        if (slot < c4max) {
            const int64_t n = pix / ((int64_t)Ho * Wo);
            const int rem = (int)(pix - n * Ho * Wo);
            const int ho = rem / Wo, wo = rem - ho * Wo;
            const float* base = x + ((n * H + (int64_t)ho * ps) * W + (int64_t)wo * ps) * C + slot * 4;
            float4 m = *reinterpret_cast<const float4*>(base);
            for (int i = 0; i < pk; ++i)
                for (int j = 0; j < pk; ++j) {
                    if (i == 0 && j == 0) continue;
                    const float4 v = *reinterpret_cast<const float4*>(base + ((int64_t)i * W + j) * C);
                    // torch max_pool2d propagates NaN
                    m.x = (v.x > m.x || v.x != v.x) ? v.x : m.x;
                    m.y = (v.y > m.y || v.y != v.y) ? v.y : m.y;
                    m.z = (v.z > m.z || v.z != v.z) ? v.z : m.z;
                    m.w = (v.w > m.w || v.w != v.w) ? v.w : m.w;
                }
            if (pre_relu) {   // ReLU in front of the BatchNorm (benchmark/BinaryNet/MLPBin.py:42-44); NaN kept
                m.x = m.x < 0.0f ? 0.0f : m.x; m.y = m.y < 0.0f ? 0.0f : m.y;
                m.z = m.z < 0.0f ? 0.0f : m.z; m.w = m.w < 0.0f ? 0.0f : m.w;
            }
            const float4 a = *reinterpret_cast<const float4*>(alpha + slot * 4);
            const float4 b = *reinterpret_cast<const float4*>(beta + slot * 4);
            // two roundings (mul, add), like the un-fused x*alpha + beta
            nib = qt_neg_bit(m.x * a.x + b.x) | (qt_neg_bit(m.y * a.y + b.y) << 1) |
                  (qt_neg_bit(m.z * a.z + b.z) << 2) | (qt_neg_bit(m.w * a.w + b.w) << 3);
        }
        const uint32_t word = or_reduce8(nib << (4 * lane8));
        if (lane8 == 0) {
            const int wi = slot >> 3;
            plane[pix * ldp + wi] = word;
            if (nibplane) {
                // ... and the same signs as the fp4 operand row of the NEXT layer's matrix-core GEMM (+1 = 0x2, -1 = 0xA, features
                // >= C = 0): the consumer's separate bits -> nibbles pass (one more launch per FC layer) is gone
                const int left = C - wi * 32;
                const uint32_t mw = left >= 32 ? 0xFFFFFFFFu : (left > 0 ? ((1u << left) - 1u) : 0u);
                uint4 o;
                o.x = (pas_spread8(mw) << 1) | (pas_spread8(word) << 3);
                o.y = (pas_spread8(mw >> 8) << 1) | (pas_spread8(word >> 8) << 3);
                o.z = (pas_spread8(mw >> 16) << 1) | (pas_spread8(word >> 16) << 3);
                o.w = (pas_spread8(mw >> 24) << 1) | (pas_spread8(word >> 24) << 3);
                uint32_t* row = nibplane + pix * ldn;
                *reinterpret_cast<uint4*>(row + wi * 4) = o;
                if (wi == ldp - 1)                                  // the row's tail past the bit plane's words
                    for (int64_t t = ldp * 4; t < ldn; t += 4) *reinterpret_cast<uint4*>(row + t) = make_uint4(0, 0, 0, 0);
            }
        }
    }
}

// MaxPool on threshold bits.  With bit = [x*alpha + beta < 0] per conv-output pixel (the threshold-bit
// epilogue of the matrix-core conv), max-pooling commutes with the monotone affine map:
//   alpha >= 0:  [max(x)*alpha + beta < 0] = AND over the window of the pixel bits
//   alpha <  0:  [max(x)*alpha + beta < 0] = OR  over the window of the pixel bits
// (fp32 multiply by a constant and add of a constant are monotone, so fl(fl(max x * a) + b) equals the
// max / min over the window of fl(fl(x * a) + b): bit-identical to pooling the fp32 tensor first, NaNs
// excepted.)  neg_alpha: 1 bit per channel, 1 <=> alpha < 0.  One thread = one output word.
__global__ __launch_bounds__(256) void pool_bits_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                        const uint32_t* __restrict__ neg_alpha, int64_t ld,
                                                        int64_t N, int H, int W, int pk, int ps, int Ho, int Wo) {
    const int64_t total = N * Ho * Wo * ld;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / ld;
        const int w = (int)(i - pix * ld);
        const int64_t n = pix / ((int64_t)Ho * Wo);
        const int rem = (int)(pix - n * Ho * Wo);
        const int ho = rem / Wo, wo = rem - ho * Wo;
        const uint32_t* base = in + ((n * H + (int64_t)ho * ps) * W + (int64_t)wo * ps) * ld + w;
        uint32_t all = 0xffffffffu, any = 0u;
        for (int a = 0; a < pk; ++a)
            for (int b = 0; b < pk; ++b) {
                const uint32_t v = base[((int64_t)a * W + b) * ld];
                all &= v;
                any |= v;
            }
        const uint32_t na = neg_alpha[w];
        out[i] = (all & ~na) | (any & na);
    }
}

// qt_pool_bits with the result expanded to the next conv's fp4 nibble pixel plane (+1 = 0x2, -1 = 0xA, channels >= C
// zero), optionally into a halo plane [N][Ho + 2hy][Wo + 2hx][ldn] (border written as zeros): pool_bits +
// bits_to_nib_pad in one pass.  One thread = one 32-channel group of one output pixel (4 nibble words).
__device__ __forceinline__ uint32_t fe_spread8(uint32_t b) {
    uint32_t t = b & 0xFFu;
    t = (t | (t << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return t;
}
__global__ __launch_bounds__(256) void pool_bits_nib_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                            const uint32_t* __restrict__ neg_alpha, int64_t ld,
                                                            int64_t ldn, int64_t N, int H, int W, int pk, int ps, int Ho,
                                                            int Wo, int hy, int hx, int C) {
    const int64_t groups = ldn / 4;
    const int Hop = Ho + 2 * hy, Wop = Wo + 2 * hx;
    const int64_t total = N * Hop * Wop * groups;           // every pixel of the halo plane: border pixels get zeros
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / groups;
        const int g = (int)(i - pix * groups);
        const int64_t n = pix / ((int64_t)Hop * Wop);
        const int rem = (int)(pix - n * Hop * Wop);
        const int ho = rem / Wop - hy, wo = rem % Wop - hx;
        uint32_t sw = 0, mw = 0;
        if (g < ld && (unsigned)ho < (unsigned)Ho && (unsigned)wo < (unsigned)Wo) {
            const uint32_t* base = in + ((n * H + (int64_t)ho * ps) * W + (int64_t)wo * ps) * ld + g;
            uint32_t all = 0xffffffffu, any = 0u;
            for (int a = 0; a < pk; ++a)
                for (int b = 0; b < pk; ++b) {
                    const uint32_t v = base[((int64_t)a * W + b) * ld];
                    all &= v;
                    any |= v;
                }
            const uint32_t na = neg_alpha[g];
            const int left = C - g * 32;
            mw = left >= 32 ? 0xFFFFFFFFu : (left > 0 ? ((1u << left) - 1u) : 0u);
            sw = ((all & ~na) | (any & na)) & mw;
        }
        uint4 o;
        o.x = (fe_spread8(mw) << 1) | (fe_spread8(sw) << 3);
        o.y = (fe_spread8(mw >> 8) << 1) | (fe_spread8(sw >> 8) << 3);
        o.z = (fe_spread8(mw >> 16) << 1) | (fe_spread8(sw >> 16) << 3);
        o.w = (fe_spread8(mw >> 24) << 1) | (fe_spread8(sw >> 24) << 3);
        *reinterpret_cast<uint4*>(out + pix * ldn + g * 4) = o;
    }
}

// The same with 16-byte accesses: one thread = FOUR words (128 channels) of one output pixel — a window tap is one dwordx4 load
// instead of four dword loads scattered over four lanes, the four nibble groups are four dwordx4 stores (AlexNet's two pools
// between the convs: 15 -> ~9 us each).  Needs 16-byte aligned bit rows (ld % 4 == 0, aligned base): the entry point checks.
__global__ __launch_bounds__(256) void pool_bits_nib_vec_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                                const uint32_t* __restrict__ neg_alpha, int64_t ld, int64_t ldn,
                                                                int64_t N, int H, int W, int pk, int ps, int Ho, int Wo, int hy, int hx,
                                                                int C) {
    // grid: x over (output column, word group) of one output row, y over the N * (Ho + 2 hy) output rows (walked with a stride
    // when they exceed the grid): 32-bit index arithmetic only — the 64-bit divisions of a flat index cost more than the loads
    const unsigned g4 = (unsigned)(ld / 4);
    const int Hop = Ho + 2 * hy, Wop = Wo + 2 * hx;
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (unsigned)Wop * g4) return;
    const unsigned wop = t / g4;
    const int q = (int)(t - wop * g4);
    if ((int64_t)q * 16 >= ldn) return;                            // bit-plane pad words past the nibble row
    const int wo = (int)wop - hx;
    const int64_t nrows = N * Hop;
    for (int64_t row = blockIdx.y; row < nrows; row += gridDim.y) {
        const int64_t n = row / Hop;
        const int ho = (int)(row - n * Hop) - hy;
        const int64_t pix = row * Wop + wop;
        const bool inside = (unsigned)ho < (unsigned)Ho && (unsigned)wo < (unsigned)Wo;
        uint4 all = make_uint4(~0u, ~0u, ~0u, ~0u), any = make_uint4(0, 0, 0, 0);
        if (inside) {
            const uint32_t* base = in + ((n * H + (int64_t)ho * ps) * W + (int64_t)wo * ps) * ld + q * 4;
            for (int a = 0; a < pk; ++a)
                for (int b = 0; b < pk; ++b) {
                    const uint4 v = *reinterpret_cast<const uint4*>(base + ((int64_t)a * W + b) * ld);
                    all.x &= v.x; all.y &= v.y; all.z &= v.z; all.w &= v.w;
                    any.x |= v.x; any.y |= v.y; any.z |= v.z; any.w |= v.w;
                }
        }
        const uint4 na = *reinterpret_cast<const uint4*>(neg_alpha + q * 4);
        const uint32_t al[4] = {all.x, all.y, all.z, all.w}, an[4] = {any.x, any.y, any.z, any.w}, nv[4] = {na.x, na.y, na.z, na.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int w = q * 4 + e;
            if ((int64_t)w * 4 >= ldn) break;
            uint32_t sw = 0, mw = 0;
            if (inside) {
                const int left = C - w * 32;
                mw = left >= 32 ? 0xFFFFFFFFu : (left > 0 ? ((1u << left) - 1u) : 0u);
                sw = ((al[e] & ~nv[e]) | (an[e] & nv[e])) & mw;
            }
            uint4 o;
            o.x = (fe_spread8(mw) << 1) | (fe_spread8(sw) << 3);
            o.y = (fe_spread8(mw >> 8) << 1) | (fe_spread8(sw >> 8) << 3);
            o.z = (fe_spread8(mw >> 16) << 1) | (fe_spread8(sw >> 16) << 3);
            o.w = (fe_spread8(mw >> 24) << 1) | (fe_spread8(sw >> 24) << 3);
            *reinterpret_cast<uint4*>(out + pix * ldn + w * 4) = o;
        }
    }
}

// MaxPool2d(k, s) on an int8 DoReFa code plane (the reference pools AFTER the quantiser,
// models/samples/AlexNet_Dorefa.py:38-41: x = quant(relu(bn(conv))); x = pool(x)).  value = fl(inv_n * code) is
// monotone in the code, so the max of the codes IS the code of the max: bit-identical to pooling the fp32 image.
// in [N][H][W][ld bytes] -> out [N][Ho + 2hy][Wo + 2hx][ld] (the halo border is written as zeros).
// One thread = 4 channels (one dword) of one output pixel.
__global__ __launch_bounds__(256) void pool_codes_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                         int64_t ldw, int64_t N, int H, int W, int pk, int ps, int Ho,
                                                         int Wo, int hy, int hx) {
    const int Hop = Ho + 2 * hy, Wop = Wo + 2 * hx;
    const int64_t total = N * Hop * Wop * ldw;              // every pixel of the halo plane: border pixels get zeros
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / ldw;
        const int w = (int)(i - pix * ldw);
        const int64_t n = pix / ((int64_t)Hop * Wop);
        const int rem = (int)(pix - n * Hop * Wop);
        const int ho = rem / Wop - hy, wo = rem % Wop - hx;
        uint32_t word = 0;
        if ((unsigned)ho < (unsigned)Ho && (unsigned)wo < (unsigned)Wo) {
            const uint32_t* base = in + ((n * H + (int64_t)ho * ps) * W + (int64_t)wo * ps) * ldw + w;
            int m0 = -128, m1 = -128, m2 = -128, m3 = -128;
            for (int a = 0; a < pk; ++a)
                for (int b = 0; b < pk; ++b) {
                    const uint32_t v = base[((int64_t)a * W + b) * ldw];
                    m0 = max(m0, (int)(int8_t)v);
                    m1 = max(m1, (int)(int8_t)(v >> 8));
                    m2 = max(m2, (int)(int8_t)(v >> 16));
                    m3 = max(m3, (int)(int8_t)(v >> 24));
                }
            word = (uint32_t)(uint8_t)m0 | ((uint32_t)(uint8_t)m1 << 8) | ((uint32_t)(uint8_t)m2 << 16) |
                   ((uint32_t)(uint8_t)m3 << 24);
        }
        out[pix * ldw + w] = word;
    }
}

}  // namespace

extern "C" int qt_pool_bits_nib(const uint32_t* in_plane, int64_t N, int64_t H, int64_t W, int64_t ld, int64_t pool_k,
                                int64_t pool_s, const uint32_t* neg_alpha, uint32_t* nib_plane, int64_t ldn, int64_t C,
                                int64_t out_halo_h, int64_t out_halo_w, qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || ld <= 0 || pool_k < 1 || pool_s < 1 || C <= 0 || out_halo_h < 0 || out_halo_w < 0)
        return QT_ERR_INVALID_ARG;
    if (pool_k > H || pool_k > W || ld < (C + 31) / 32 || ldn < (C + 7) / 8) return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!in_plane || !nib_plane || !neg_alpha) return QT_ERR_INVALID_ARG;
    if ((ld & 3) || (ldn & 3) || !qt_aligned16(nib_plane)) return QT_ERR_ALIGNMENT;
    if (H > 32767 || W > 32767 || out_halo_h > 64 || out_halo_w > 64) return QT_ERR_UNSUPPORTED;
    const int64_t Ho = (H - pool_k) / pool_s + 1, Wo = (W - pool_k) / pool_s + 1;
    if (qt_aligned16(in_plane) && qt_aligned16(neg_alpha) && ldn <= 4 * ld) {     // (ldn <= 4 ld: every nibble group has its bit word)
        const int64_t rows_out = N * (Ho + 2 * out_halo_h);
        const int64_t per_row = (Wo + 2 * out_halo_w) * (ld / 4);
        const unsigned bs = per_row <= 64 ? 64u : (per_row <= 128 ? 128u : 256u);          // a row of a small map fills one wave
        const dim3 gridv((unsigned)((per_row + bs - 1) / bs), (unsigned)std::min<int64_t>(rows_out, 65535));
        hipLaunchKernelGGL(pool_bits_nib_vec_kernel, gridv, dim3(bs), 0, (hipStream_t)stream, in_plane, nib_plane, neg_alpha, ld,
                           ldn, N, (int)H, (int)W, (int)pool_k, (int)pool_s, (int)Ho, (int)Wo, (int)out_halo_h, (int)out_halo_w, (int)C);
        return qt_check_launch();
    }
    const int grid = qt_stream_grid((N * (Ho + 2 * out_halo_h) * (Wo + 2 * out_halo_w) * (ldn / 4) + 255) / 256);
    hipLaunchKernelGGL(pool_bits_nib_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in_plane, nib_plane,
                       neg_alpha, ld, ldn, N, (int)H, (int)W, (int)pool_k, (int)pool_s, (int)Ho, (int)Wo,
                       (int)out_halo_h, (int)out_halo_w, (int)C);
    return qt_check_launch();
}

extern "C" int qt_pool_codes_i8(const int8_t* in_plane, int64_t N, int64_t H, int64_t W, int64_t ld_bytes,
                                int64_t pool_k, int64_t pool_s, int8_t* out_plane, int64_t out_halo_h,
                                int64_t out_halo_w, qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || ld_bytes <= 0 || pool_k < 1 || pool_s < 1 || out_halo_h < 0 || out_halo_w < 0)
        return QT_ERR_INVALID_ARG;
    if (pool_k > H || pool_k > W) return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!in_plane || !out_plane) return QT_ERR_INVALID_ARG;
    if ((ld_bytes & 15) || !qt_aligned16(in_plane) || !qt_aligned16(out_plane)) return QT_ERR_ALIGNMENT;
    if (H > 32767 || W > 32767 || out_halo_h > 64 || out_halo_w > 64) return QT_ERR_UNSUPPORTED;
    const int64_t Ho = (H - pool_k) / pool_s + 1, Wo = (W - pool_k) / pool_s + 1;  // floor mode, no padding
    const int64_t ldw = ld_bytes / 4;
    const int grid = qt_stream_grid((N * (Ho + 2 * out_halo_h) * (Wo + 2 * out_halo_w) * ldw + 255) / 256);
    hipLaunchKernelGGL(pool_codes_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint32_t*>(in_plane), reinterpret_cast<uint32_t*>(out_plane), ldw, N,
                       (int)H, (int)W, (int)pool_k, (int)pool_s, (int)Ho, (int)Wo, (int)out_halo_h, (int)out_halo_w);
    return qt_check_launch();
}

extern "C" int qt_pool_bits(const uint32_t* in_plane, int64_t N, int64_t H, int64_t W, int64_t ld,
                            int64_t pool_k, int64_t pool_s, const uint32_t* neg_alpha, uint32_t* out_plane,
                            qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || ld <= 0 || pool_k < 1 || pool_s < 1) return QT_ERR_INVALID_ARG;
    if (pool_k > H || pool_k > W) return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!in_plane || !out_plane || !neg_alpha) return QT_ERR_INVALID_ARG;
    if (ld & 3) return QT_ERR_ALIGNMENT;
    if (H > INT32_MAX / 2 || W > INT32_MAX / 2) return QT_ERR_UNSUPPORTED;
    const int64_t Ho = (H - pool_k) / pool_s + 1, Wo = (W - pool_k) / pool_s + 1;
    const int grid = qt_stream_grid((N * Ho * Wo * ld + 255) / 256);
    hipLaunchKernelGGL(pool_bits_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in_plane, out_plane,
                       neg_alpha, ld, N, (int)H, (int)W, (int)pool_k, (int)pool_s, (int)Ho, (int)Wo);
    return qt_check_launch();
}

static int pool_affine_sign_pack_impl(const float* x, int64_t N, int64_t H, int64_t W, int64_t C, int64_t pool_k, int64_t pool_s,
                                      const float* alpha, const float* beta, uint32_t* sign_plane, int64_t ldp, uint32_t* nib_plane,
                                      int64_t ldn, int pre_relu, qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || C <= 0 || pool_k < 1 || pool_s < 1) return QT_ERR_INVALID_ARG;
    if (nib_plane && (ldn < 4 * ldp || (ldn & 3) || !qt_aligned16(nib_plane))) return QT_ERR_ALIGNMENT;
    if (pool_k > H || pool_k > W) return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!x || !alpha || !beta || !sign_plane) return QT_ERR_INVALID_ARG;
    if ((C & 3) || ldp < (C + 31) / 32 || (ldp & 3)) return QT_ERR_ALIGNMENT;
    if (!qt_aligned16(x) || !qt_aligned16(alpha) || !qt_aligned16(beta) || !qt_aligned16(sign_plane))
        return QT_ERR_ALIGNMENT;
    if (H > INT32_MAX / 2 || W > INT32_MAX / 2 || C > INT32_MAX / 2) return QT_ERR_UNSUPPORTED;
    const int64_t Ho = (H - pool_k) / pool_s + 1, Wo = (W - pool_k) / pool_s + 1;  // floor mode, no padding
    const int64_t total = N * Ho * Wo * ldp * 8;
    const int grid = qt_stream_grid((total + 255) / 256);
    hipLaunchKernelGGL(pool_affine_sign_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, alpha,
                       beta, sign_plane, ldp, N, (int)H, (int)W, (int)C, (int)pool_k, (int)pool_s, (int)Ho,
                       (int)Wo, pre_relu ? 1 : 0, nib_plane, ldn);
    return qt_check_launch();
}

extern "C" int qt_pool_affine_sign_pack_nhwc(const float* x, int64_t N, int64_t H, int64_t W, int64_t C,
                                             int64_t pool_k, int64_t pool_s, const float* alpha,
                                             const float* beta, uint32_t* sign_plane, int64_t ldp,
                                             int pre_relu, qt_stream_t stream) {
    return pool_affine_sign_pack_impl(x, N, H, W, C, pool_k, pool_s, alpha, beta, sign_plane, ldp, nullptr, 0, pre_relu, stream);
}

extern "C" int qt_pool_affine_sign_pack_nib_nhwc(const float* x, int64_t N, int64_t H, int64_t W, int64_t C, int64_t pool_k,
                                                 int64_t pool_s, const float* alpha, const float* beta, uint32_t* sign_plane,
                                                 int64_t ldp, uint32_t* nib_plane, int64_t ldn, int pre_relu, qt_stream_t stream) {
    if (!nib_plane) return QT_ERR_INVALID_ARG;
    return pool_affine_sign_pack_impl(x, N, H, W, C, pool_k, pool_s, alpha, beta, sign_plane, ldp, nib_plane, ldn, pre_relu, stream);
}
