// Weight gradient of a stride-1 conv, PIXEL-MAJOR form (training path; the companion of wgrad.hip):
//
//     dW[co, ci, kh, kw] = sum_q  g[q, co] * xpad[q + kh * N * Wq + kw, ci]          q = (y * N + n) * Wq + x
//
// wgrad.hip runs this as one GEMM per tap over K-major (transposed) planes, which re-reads the gradient planes once per
// tap and needs kw pre-shifted activation copies.  Here both operands stay in the order the tensors already have —
// [position][channel], so a tap is a ROW offset — and ONE workgroup accumulates ALL kh * kw taps of its (co x ci) tile: per
// 32-position stage it loads the three exact bf16 planes of the gradient tile once and kh activation tiles of 32 + kw - 1
// rows (42 KB for 432 MFMAs on the 128 x 64 tile) — matrix-bound where the per-tap GEMM is fill-bound.  The MFMA fragments
// (8 consecutive positions of one channel per lane) are read out of the [position][channel] LDS tiles with
// ds_read_b64_tr_b16, the transposing LDS read (semantics: tools/ubench/tr_probe.hip — inside a 16-lane group lane q
// addresses 4 consecutive channels of row q / 4, lane i receives channel i of the four rows).  The three gradient terms
// (hi / mid / lo) accumulate into the same fp32 registers, so nothing has to be summed over terms afterwards.
//
// Two kernels share the layout (PmCfg), the LDS-DMA ring of three stages and the XCD-aware tile order:
//   wgrad_pm_full_kernel<128, 64, 3, 3> : one wave per 32 x 32 block, every tap in that wave; straight-line, software-pipelined
//                                         (fragments prefetched across a mid-stage hand-over).  3 x 3 convs with Cout % 128 == 0.
//   wgrad_pm_kernel<TM, TN, KH, KW>     : waves = blocks x tap groups (a wave owns the taps t = group, group + G, ...); 64 x 64
//                                         tiles for the other 3 x 3 convs, 64 x 32 for 5 x 5.
// K (the positions) is cut into slices; the partial results [slice][tap][co][ci] are reduced by pm_reduce_kernel, which applies
// the scale and the straight-through mask.  Measurements and the duty analysis: DESIGN.md section 1, profiles/r2_wgrad_pm.md.
#include "qt_common.h"

namespace {

typedef float pm_v16f __attribute__((ext_vector_type(16)));
typedef __bf16 pm_bf8 __attribute__((ext_vector_type(8)));
typedef _Float16 pm_h8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t pm_bf16_rn_bits(float f) {  // round-to-nearest-even, NaN kept quiet (as split_bf16.hip)
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

// Round 3: the gradient goes in as NP planes — 3 exact bf16 terms (v_mfma_f32_32x32x16_bf16), or 2 fp16 terms of g / s with a
// per-tensor power-of-two scale s (v_mfma_f32_32x32x16_f16; csrc/split_f16.hip has the bound): 2/3 of the MFMAs, of the gradient
// bytes and of the LDS fill.  The activation plane holds the same +-1 / 0 / small-integer values in the matching 16-bit format.
__device__ __forceinline__ uint32_t pm_f16_bits(float f) {
    const _Float16 h = (_Float16)f;
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
__device__ __forceinline__ float pm_f16_to_f32(uint32_t b) {
    const unsigned short u = (unsigned short)b;
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
}
// the NP 16-bit terms of v (NP == 2: v already divided by the scale)
template <int NP>
__device__ __forceinline__ void pm_split(float v, uint32_t (&t)[3]) {
    if constexpr (NP == 3) {
        t[0] = pm_bf16_rn_bits(v);
        const float r1 = v - __uint_as_float(t[0] << 16);
        t[1] = pm_bf16_rn_bits(r1);
        t[2] = pm_bf16_rn_bits(r1 - __uint_as_float(t[1] << 16));
    } else {
        t[0] = pm_f16_bits(v);
        t[1] = pm_f16_bits(v - pm_f16_to_f32(t[0]));
        t[2] = 0;
    }
}


// ---- packers (HBM-bound, elementwise: no transposition) --------------------------------------------------------------
// One workgroup per (row y, image n) = Wq consecutive positions; an item = 8 channels of one position (32 bytes of fp32 in,
// 16 bytes of bf16 out per plane).  Channels-last sources (channel stride 1) put the channel chunks on adjacent lanes — reads and
// writes are both contiguous; other layouts put adjacent x on adjacent lanes so that each of the eight channel reads is
// coalesced.  All index arithmetic is 32-bit.
__device__ __forceinline__ void pm_item(int item, int c8, int Wq, bool channel_fastest, int& x, int& c0) {
    if (channel_fastest) {
        x = item / c8;
        c0 = (item - x * c8) << 3;
    } else {
        const int cc = item / Wq;
        x = item - cc * Wq;
        c0 = cc << 3;
    }
}

// The rows past the packed ones (the K-slice rounding of the gradient planes, the look-ahead rows of the activation plane) are
// zeroed by PM_TAIL_BLOCKS workgroups appended to the pack launch (round 6: they used to be a launch of their own per plane).
constexpr int PM_TAIL_BLOCKS = 32;
struct PmTail {
    uint16_t* base;        // first tail element of plane 0
    long long plane;       // elements between the planes' tails
    long long n16;         // 16-byte items per tail
    int planes, rows;      // blocks [rows, rows + PM_TAIL_BLOCKS) of the grid do the zeroing
};
__device__ __forceinline__ bool pm_tail_zero(const PmTail& t) {
    if ((int)blockIdx.x < t.rows) return false;
    const long long b = (long long)blockIdx.x - t.rows;
    for (int pl = 0; pl < t.planes; ++pl) {
        uint4* p = reinterpret_cast<uint4*>(t.base + pl * t.plane);
        for (long long i = b * 256 + threadIdx.x; i < t.n16; i += (long long)PM_TAIL_BLOCKS * 256) p[i] = make_uint4(0, 0, 0, 0);
    }
    return true;
}

// G3[t][q][Cp]: exact bf16 split of g at position q = (y * N + n) * Wq + x (zero where x >= Wo and for channels >= Cout)
template <int NP>
__global__ __launch_bounds__(256) void pm_pack_grad_kernel(const float* __restrict__ g, int64_t sn, int64_t sc, int64_t sh_,
                                                           int64_t sw, int N, int Cout, int Wo, int Wq, int Cp, int64_t Qa,
                                                           uint16_t* __restrict__ G3, const float* __restrict__ scale2, PmTail tail) {
    if (pm_tail_zero(tail)) return;
    const float* inv = NP == 2 ? scale2 + Cp : nullptr;               // 1 / s[c], per channel (split_f16.hip)
    const int c8 = Cp >> 3, items = Wq * c8;
    const int y = blockIdx.x / N, n = blockIdx.x - y * N;
    const float* row = g + (int64_t)n * sn + (int64_t)y * sh_;
    uint16_t* out = G3 + (int64_t)blockIdx.x * Wq * Cp;
    const bool cf = sc == 1;
    for (int item = threadIdx.x; item < items; item += 256) {
        int x, c0;
        pm_item(item, c8, Wq, cf, x, c0);
        uint32_t h[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        if (x < Wo) {
            const float* src = row + (int64_t)x * sw + (int64_t)c0 * sc;
            float v[8];
            if (cf && c0 + 8 <= Cout && !((reinterpret_cast<uintptr_t>(src)) & 15)) {
                const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = c0 + i < Cout ? src[(int64_t)i * sc] : 0.0f;
            }
            if constexpr (NP == 2) {
                const float4 il = *reinterpret_cast<const float4*>(inv + c0), ih = *reinterpret_cast<const float4*>(inv + c0 + 4);
                v[0] *= il.x; v[1] *= il.y; v[2] *= il.z; v[3] *= il.w; v[4] *= ih.x; v[5] *= ih.y; v[6] *= ih.z; v[7] *= ih.w;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                uint32_t t[3];
                pm_split<NP>(v[i], t);
                const int s = (i & 1) * 16;
                h[0][i >> 1] |= t[0] << s;
                h[1][i >> 1] |= t[1] << s;
                h[2][i >> 1] |= t[2] << s;
            }
        }
#pragma unroll
        for (int s = 0; s < NP; ++s)
            *reinterpret_cast<uint4*>(out + (int64_t)s * Qa * Cp + (int64_t)x * Cp + c0) = make_uint4(h[s][0], h[s][1], h[s][2], h[s][3]);
    }
}

// The same pack for channels-last gradients that ALSO leaves the bias gradient behind (grad_bias = sum of g over n, y, x — a
// third pass over g otherwise): every thread keeps one fixed 8-channel chunk and walks the row's positions xl, xl + XPAR, ...,
// so its eight running sums live in registers; the XPAR partial rows are added in a fixed order through LDS and the workgroup
// writes bias_part[row][Cp].  Deterministic (no atomics): pm_bias_reduce_kernel adds the rows in a fixed order as well.
template <int NP>
__global__ __launch_bounds__(256) void pm_pack_grad_bias_kernel(const float* __restrict__ g, int64_t sn, int64_t sh_, int64_t sw, int N,
                                                                int Cout, int Wo, int Wq, int Cp, int64_t Qa, uint16_t* __restrict__ G3,
                                                                float* __restrict__ bias_part, const float* __restrict__ scale2,
                                                                PmTail tail) {
    if (pm_tail_zero(tail)) return;                          // (uniform per workgroup: taken before any barrier)
    __shared__ float red[2048];                              // [xpar][Cp] partial sums, xpar * Cp <= 256 * 8
    const int c8 = Cp >> 3, xpar = 256 / c8;
    const int y = blockIdx.x / N, n = blockIdx.x - y * N;
    const float* row = g + (int64_t)n * sn + (int64_t)y * sh_;
    uint16_t* out = G3 + (int64_t)blockIdx.x * Wq * Cp;
    const int t = threadIdx.x, chunk = t % c8, xl = t / c8, c0 = chunk << 3;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float inv[8] = {1, 1, 1, 1, 1, 1, 1, 1};                          // NP == 2: 1 / s[c] of this thread's eight channels
    if constexpr (NP == 2) {
        if (xl < xpar) {
#pragma unroll
            for (int i = 0; i < 8; ++i) inv[i] = scale2[Cp + c0 + i];
        }
    }
    if (xl < xpar) {
        for (int x = xl; x < Wq; x += xpar) {
            uint32_t h[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            if (x < Wo) {
                const float* src = row + (int64_t)x * sw + c0;
                float v[8];
                if (c0 + 8 <= Cout && !((reinterpret_cast<uintptr_t>(src)) & 15)) {
                    const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
                    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = c0 + i < Cout ? src[i] : 0.0f;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] += v[i];
                    uint32_t t3[3];
                    pm_split<NP>(NP == 2 ? v[i] * inv[i] : v[i], t3);
                    const int s = (i & 1) * 16;
                    h[0][i >> 1] |= t3[0] << s;
                    h[1][i >> 1] |= t3[1] << s;
                    h[2][i >> 1] |= t3[2] << s;
                }
            }
#pragma unroll
            for (int s = 0; s < NP; ++s)
                *reinterpret_cast<uint4*>(out + (int64_t)s * Qa * Cp + (int64_t)x * Cp + c0) = make_uint4(h[s][0], h[s][1], h[s][2], h[s][3]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) red[xl * Cp + c0 + i] = acc[i];
    }
    __syncthreads();
    for (int c = t; c < Cp; c += 256) {
        float s = red[c];
        for (int j = 1; j < xpar; ++j) s += red[j * Cp + c];
        bias_part[(int64_t)blockIdx.x * Cp + c] = s;
    }
}

// out[seg][c] (+)= sum over the rows [seg * seg_len, (seg + 1) * seg_len) of in[row][Cp], c < climit: one workgroup per (32 channels,
// segment), 8 row groups added in a fixed order.  Two passes (rows -> <= 128 segments -> 1) keep every pass wide enough for the chip.
__global__ __launch_bounds__(256) void pm_bias_reduce_kernel(const float* __restrict__ in, int rows, int seg_len, int Cp, int climit,
                                                             int accumulate, float* __restrict__ out) {
    __shared__ float red[8][32];
    const int c = blockIdx.x * 32 + (threadIdx.x & 31), rg = threadIdx.x >> 5;
    const int r0 = blockIdx.y * seg_len, r1 = min(rows, r0 + seg_len);
    float s = 0.0f;
    if (c < Cp)
        for (int r = r0 + rg; r < r1; r += 8) s += in[(int64_t)r * Cp + c];
    red[rg][threadIdx.x & 31] = s;
    __syncthreads();
    if (rg == 0 && c < climit) {
        float tot = red[0][threadIdx.x];
#pragma unroll
        for (int j = 1; j < 8; ++j) tot += red[j][threadIdx.x];
        float* o = out + (int64_t)blockIdx.y * Cp + c;
        *o = accumulate ? *o + tot : tot;
    }
}

// XP[q][Cp]: bf16(x * x_scale) at padded position q = (y * N + n) * Wq + x over the H + 2 ph padded rows (zero outside the image,
// past the pitch and in channels >= Cin)
template <bool F16>
__global__ __launch_bounds__(256) void pm_pack_act_kernel(const float* __restrict__ xin, int64_t sn, int64_t sc, int64_t sh_,
                                                          int64_t sw, int N, int Cin, int H, int W, int ph, int pw, int Wq,
                                                          int Cp, float x_scale, uint16_t* __restrict__ XP, PmTail tail) {
    if (pm_tail_zero(tail)) return;
    const int c8 = Cp >> 3, items = Wq * c8;
    const int y = blockIdx.x / N, n = blockIdx.x - y * N;
    const int yy = y - ph;
    const bool row_ok = yy >= 0 && yy < H;
    const float* row = xin + (int64_t)n * sn + (int64_t)yy * sh_;
    uint16_t* out = XP + (int64_t)blockIdx.x * Wq * Cp;
    const bool cf = sc == 1;
    for (int item = threadIdx.x; item < items; item += 256) {
        int x, c0;
        pm_item(item, c8, Wq, cf, x, c0);
        const int xx = x - pw;
        uint32_t h[4] = {0, 0, 0, 0};
        if (row_ok && xx >= 0 && xx < W) {
            const float* src = row + (int64_t)xx * sw + (int64_t)c0 * sc;
            float v[8];
            if (cf && c0 + 8 <= Cin && !((reinterpret_cast<uintptr_t>(src)) & 15)) {
                const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = c0 + i < Cin ? src[(int64_t)i * sc] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) h[i >> 1] |= (F16 ? pm_f16_bits(v[i] * x_scale) : pm_bf16_rn_bits(v[i] * x_scale)) << ((i & 1) * 16);
        }
        *reinterpret_cast<uint4*>(out + (int64_t)x * Cp + c0) = make_uint4(h[0], h[1], h[2], h[3]);
    }
}

// Strided first layer over a real-valued image (ops.conv2d_grad_weight_s2d): the stride-s conv is the stride-1 conv of the
// space-to-depth image, and the image's exact bf16 split goes in as three channel groups —
// XP[q][t * Cs8 + (c * s + dy) * s + dx] = term t (hi / mid / lo) of xpad[n, c, Y s + dy - ph, X s + dx - pw],
// q = (Y * N + n) * Wq + X; zero outside the image, for X >= Ws, and in the channels between the groups' ends and Cp.
__global__ __launch_bounds__(256) void pm_pack_act_s2d_kernel(const float* __restrict__ xin, int64_t sn, int64_t sc, int64_t sh_,
                                                              int64_t sw, int N, int C, int H, int W, int s, int ph, int pw,
                                                              int Ws, int Wq, int Cs8, int Cp, uint16_t* __restrict__ XP) {
    const int c8 = Cp >> 3, real8 = Cs8 >> 3, items = Wq * c8;
    const int Y = blockIdx.x / N, n = blockIdx.x - Y * N;
    const int Cs = C * s * s;
    uint16_t* out = XP + (int64_t)blockIdx.x * Wq * Cp;
    for (int item = threadIdx.x; item < items; item += 256) {
        const int X = item / c8, chunk = item - X * c8;
        if (chunk >= real8) {
            if (chunk >= 3 * real8) *reinterpret_cast<uint4*>(out + (int64_t)X * Cp + chunk * 8) = make_uint4(0, 0, 0, 0);
            continue;                                            // chunks [real8, 3 real8) are the mid / lo groups, written below
        }
        uint32_t h[3][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        if (X < Ws) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int cc = chunk * 8 + i;
                if (cc >= Cs) break;
                const int c = cc / (s * s), r = cc - c * s * s;
                const int dy = r / s, dx = r - dy * s;
                const int yy = Y * s + dy - ph, xx = X * s + dx - pw;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const float v = xin[(int64_t)n * sn + (int64_t)c * sc + (int64_t)yy * sh_ + (int64_t)xx * sw];
                const uint32_t a = pm_bf16_rn_bits(v);
                const float r1 = v - __uint_as_float(a << 16);
                const uint32_t b = pm_bf16_rn_bits(r1);
                const uint32_t cbits = pm_bf16_rn_bits(r1 - __uint_as_float(b << 16));
                const int sft = (i & 1) * 16;
                h[0][i >> 1] |= a << sft;
                h[1][i >> 1] |= b << sft;
                h[2][i >> 1] |= cbits << sft;
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t)
            *reinterpret_cast<uint4*>(out + (int64_t)X * Cp + t * Cs8 + chunk * 8) = make_uint4(h[t][0], h[t][1], h[t][2], h[t][3]);
    }
}

// The same gather with the image as TWO fp16 terms of x / s (s = scale2[0], the per-tensor power of two of split_f16.hip) in two
// channel groups, XP[q][t * Cs8 + e] — the activation of the two-plane kernel (2/3 of the channels, 2/3 of the gradient planes).
// Channels-last images: one workgroup per plane row (Y, n); the s input rows go through LDS with full-line loads (the generic
// kernel above gathers eight scattered floats per item: 416 us for AlexNet's 154 MB batch, this one 60).
__global__ __launch_bounds__(256) void pm_pack_act_s2d_f16_rows_kernel(const float* __restrict__ xin, int64_t sn, int64_t sh_,
                                                                       const float* __restrict__ scale2, int N, int C, int H, int W,
                                                                       int s, int ph, int pw, int Ws, int Wq, int Cs8, int Cp,
                                                                       int vec_ok, uint16_t* __restrict__ XP) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pm_s2d_smem[];
    const float inv = scale2[1];
    const int E = C * s * s, rowf = W * C, rowf4 = (rowf + 3) & ~3;
    float* rows = reinterpret_cast<float*>(pm_s2d_smem);                  // [s][rowf4]
    int* lut_off = reinterpret_cast<int*>(rows + (size_t)s * rowf4);      // [E] dy * rowf4 + (dx - pw) * C + c
    int* lut_dx = lut_off + E;                                            // [E] dx - pw
    const int tid = threadIdx.x;
    const int Y = blockIdx.x / N, n = blockIdx.x - Y * N;
    for (int e = tid; e < E; e += 256) {
        const int c = e / (s * s), r = e - c * s * s, dy = r / s, dx = r - dy * s;
        lut_off[e] = dy * rowf4 + (dx - pw) * C + c;
        lut_dx[e] = dx - pw;
    }
    for (int dy = 0; dy < s; ++dy) {
        const int hh = s * Y + dy - ph;
        const bool ok = hh >= 0 && hh < H;
        const float* src = xin + (int64_t)n * sn + (int64_t)(ok ? hh : 0) * sh_;
        float* dst = rows + dy * rowf4;
        if (vec_ok) {
            for (int i = tid * 4; i < rowf; i += 1024)
                *reinterpret_cast<float4*>(dst + i) = ok ? *reinterpret_cast<const float4*>(src + i) : make_float4(0, 0, 0, 0);
        } else {
            for (int i = tid; i < rowf; i += 256) dst[i] = ok ? src[i] : 0.0f;
        }
    }
    __syncthreads();
    const int c8 = Cp >> 3, real8 = Cs8 >> 3, items = Wq * c8;
    uint16_t* out = XP + (int64_t)blockIdx.x * Wq * Cp;
    for (int item = tid; item < items; item += 256) {
        const int X = item / c8, chunk = item - X * c8;
        uint32_t h[4] = {0, 0, 0, 0};
        if (chunk < 2 * real8 && X < Ws) {
            const int t = chunk >= real8, e0 = (chunk - t * real8) * 8;
            const int base = s * X * C, wx = s * X;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e = e0 + i;
                float v = 0.0f;
                if (e < E) {
                    const int ww = wx + lut_dx[e];
                    if ((unsigned)ww < (unsigned)W) v = rows[lut_off[e] + base] * inv;
                }
                uint32_t b = pm_f16_bits(v);
                if (t) b = pm_f16_bits(v - pm_f16_to_f32(b));
                h[i >> 1] |= b << ((i & 1) * 16);
            }
        }
        *reinterpret_cast<uint4*>(out + (int64_t)X * Cp + chunk * 8) = make_uint4(h[0], h[1], h[2], h[3]);
    }
}

// any other layout: the scattered gather
__global__ __launch_bounds__(256) void pm_pack_act_s2d_f16_kernel(const float* __restrict__ xin, int64_t sn, int64_t sc, int64_t sh_,
                                                                  int64_t sw, const float* __restrict__ scale2, int N, int C, int H,
                                                                  int W, int s, int ph, int pw, int Ws, int Wq, int Cs8, int Cp,
                                                                  uint16_t* __restrict__ XP) {
    const float inv = scale2[1];
    const int c8 = Cp >> 3, real8 = Cs8 >> 3, items = Wq * c8;
    const int Y = blockIdx.x / N, n = blockIdx.x - Y * N;
    const int Cs = C * s * s;
    uint16_t* out = XP + (int64_t)blockIdx.x * Wq * Cp;
    for (int item = threadIdx.x; item < items; item += 256) {
        const int X = item / c8, chunk = item - X * c8;
        uint32_t h[4] = {0, 0, 0, 0};
        if (chunk < 2 * real8 && X < Ws) {
            const int t = chunk >= real8, e0 = (chunk - t * real8) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int cc = e0 + i;
                if (cc >= Cs) break;
                const int c = cc / (s * s), r = cc - c * s * s;
                const int dy = r / s, dx = r - dy * s;
                const int yy = Y * s + dy - ph, xx = X * s + dx - pw;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const float v = xin[(int64_t)n * sn + (int64_t)c * sc + (int64_t)yy * sh_ + (int64_t)xx * sw] * inv;
                uint32_t b = pm_f16_bits(v);
                if (t) b = pm_f16_bits(v - pm_f16_to_f32(b));
                h[i >> 1] |= b << ((i & 1) * 16);
            }
        }
        *reinterpret_cast<uint4*>(out + (int64_t)X * Cp + chunk * 8) = make_uint4(h[0], h[1], h[2], h[3]);
    }
}

// rows past the packed ones (the K-slice rounding of the gradient planes, the look-ahead rows of the activation plane)
__global__ __launch_bounds__(256) void pm_zero_kernel(uint4* __restrict__ p, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = make_uint4(0, 0, 0, 0);
}

// ---- the kernel ------------------------------------------------------------------------------------------------------
struct PmArgs {
    const unsigned char* G3;   // [3][Qa][Cpo] bf16
    const unsigned char* XP;   // [Qx][Cpi] bf16
    float* part;               // [nslice][taps][Cpo][Cpi] fp32
    long long Qa;              // positions per gradient plane (a multiple of 32 * nslice ... see launch)
    long long kh_rows;         // N * Wq: position offset of one kernel row
    long long slice_pos;       // positions per K slice (multiple of 32)
    int Cpo, Cpi;
    int tiles_co, tiles_ci, total;   // 64-row tiles, TN-column tiles, tiles_co * tiles_ci * nslice
};

constexpr int PM_KS = 32;       // positions per stage (two MFMA k-steps of 16)

template <int TM, int TN, int KH, int KW, int NP = 3>
struct PmCfg {
    static constexpr int T = KH * KW;
    static constexpr int NH = TN / 32;                    // 32-channel sub-tiles of the activation tile
    static constexpr int MB = TM / 32;                    // 32-channel sub-tiles of the gradient tile
    static constexpr int MW = TM / 64;                    // 32 x 32 output blocks a wave owns per tap (stacked along co)
    static constexpr int WB = 2 * NH;                     // wave positions in the tile: 2 (co halves) x NH
    static constexpr int G = 8 / WB;                      // tap groups
    static constexpr int MAXT = (T + G - 1) / G;          // taps per wave
    // LDS image: 32-channel sub-tiles with a 64-byte row pitch — the four rows x 64 bytes a half-wave's transposing read
    // touches are then 256 consecutive bytes (every bank once); a 128-byte pitch would put rows r and r + 2 on the same banks
    static constexpr int XNEED = PM_KS + KW - 1;          // activation rows one stage reads per kernel row
    static constexpr int XR = (XNEED + 15) / 16 * 16;     // rows reserved (whole 16-row DMA pieces; surplus lanes are masked off)
    static constexpr int A_BYTES = NP * MB * PM_KS * 64;  // gradient tile: NP planes x MB sub-tiles x 32 rows
    static constexpr int X_BYTES = KH * NH * XR * 64;
    static constexpr int STAGE = A_BYTES + X_BYTES;
    static constexpr int NST = 3;                         // ring depth: the DMA runs two stages ahead of the MFMAs
    static constexpr int LDS = NST * STAGE;
    // one stage = LDS-DMA pieces of 1 KiB (64 lanes x 16 bytes = 16 rows of one sub-tile), dealt round-robin to the 8 waves
    static constexpr int A_PIECES = NP * MB * 2;
    static constexpr int XP_PER = XR / 16;
    static constexpr int PIECES = A_PIECES + KH * NH * XP_PER;
    static constexpr int NPW = (PIECES + 7) / 8;          // pieces per wave (the last round may be short)
};

// 64 lanes x 16 bytes, global -> LDS at the wave-uniform byte address lds_dst, no VGPR round trip.  Issued from asm so that
// the compiler does not drain vmcnt in front of later LDS reads: the kernel counts its own outstanding pieces.
__device__ __forceinline__ void pm_dma16(const unsigned char* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_dst)
                 : "memory");
}

template <int NP>
__device__ __forceinline__ pm_v16f pm_mfma(const uint4& aq, const uint4& bq, pm_v16f c) {
    if constexpr (NP == 3) {
        pm_bf8 av, bv;
        __builtin_memcpy(&av, &aq, 16);
        __builtin_memcpy(&bv, &bq, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
    } else {
        pm_h8 av, bv;
        __builtin_memcpy(&av, &aq, 16);
        __builtin_memcpy(&bv, &bq, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
    }
}

template <int N>
__device__ __forceinline__ void pm_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int TM, int TN, int KH, int KW, int NP>
__global__ __launch_bounds__(512) void wgrad_pm_kernel(PmArgs a) {
    using C = PmCfg<TM, TN, KH, KW, NP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wb = wave % C::WB, tg = wave / C::WB;
    const int mg = wb & 1, nb = wb >> 1;
    // XCD-aware order: the hardware deals consecutive workgroup ids round-robin over the 8 XCDs (each with its own L2), so id
    // -> (id % 8) * (grid / 8) + id / 8 puts a contiguous run of logical tiles on one XCD; logical order = ci tile fastest, then
    // co tile, then K slice: the workgroups sharing a gradient tile (the larger operand) and a slice's activation rows sit
    // behind the same L2 and move through the positions together
    const int per_xcd = (int)(gridDim.x >> 3);
    const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= a.total) return;
    const int ci_t = logical % a.tiles_ci, rest = logical / a.tiles_ci;
    const int co_t = rest % a.tiles_co, slice = rest / a.tiles_co;
    const int co0 = co_t * TM, ci0 = ci_t * TN;
    const long long q_begin = (long long)slice * a.slice_pos;
    const int nstages = (int)(a.slice_pos / PM_KS);
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);

    pm_v16f acc[C::MAXT][C::MW];
#pragma unroll
    for (int i = 0; i < C::MAXT; ++i)
#pragma unroll
        for (int j = 0; j < C::MW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // this wave's pieces: per-lane source address at stage 0, LDS offset inside a stage, bytes to advance per stage
    const int prow = lane >> 2, pch = (lane & 3) * 16;
    const unsigned char* psrc[C::NPW];
    unsigned pdst[C::NPW];
    bool pact[C::NPW];
    const long long a_step = (long long)PM_KS * a.Cpo * 2, x_step = (long long)PM_KS * a.Cpi * 2;
#pragma unroll
    for (int j = 0; j < C::NPW; ++j) {
        const int p = j * 8 + wave;
        if (p < C::A_PIECES) {
            const int t = p / (2 * C::MB), sub = (p >> 1) % C::MB, r16 = p & 1;
            psrc[j] = a.G3 + (((long long)t * a.Qa + q_begin + r16 * 16 + prow) * a.Cpo + co0 + sub * 32) * 2 + pch;
            pdst[j] = (unsigned)(((t * C::MB + sub) * PM_KS + r16 * 16) * 64);
            pact[j] = true;
        } else {
            const int px = (p < C::PIECES ? p : C::A_PIECES) - C::A_PIECES;
            const int sub = px / C::XP_PER, pr = px - sub * C::XP_PER;           // sub = kh * NH + h
            const int kh = sub / C::NH, h = sub - kh * C::NH;
            psrc[j] = a.XP + ((q_begin + (long long)kh * a.kh_rows + pr * 16 + prow) * a.Cpi + ci0 + h * 32) * 2 + pch;
            pdst[j] = (unsigned)(C::A_BYTES + (sub * C::XR + pr * 16) * 64);
            pact[j] = pr * 16 + prow < C::XNEED;                                 // rows past the ones a stage reads: lane masked off
        }
    }
    const int n_mine = (C::PIECES - wave + 7) / 8;                               // pieces this wave issues per stage (wave-uniform)

    auto issue_stage = [&](int s) {
        const unsigned base = lds0 + (unsigned)((s % C::NST) * C::STAGE);
#pragma unroll
        for (int j = 0; j < C::NPW; ++j) {
            const int p = j * 8 + wave;
            if (j == C::NPW - 1 && p >= C::PIECES) break;
            const unsigned char* src = psrc[j] + (long long)s * (p < C::A_PIECES ? a_step : x_step);
            if (pact[j]) pm_dma16(src, base + pdst[j]);
        }
    };
    // wait until at most `this wave's pieces of ONE stage` are outstanding (the DMA queue retires in order)
    auto wait_one_stage_in_flight = [&]() {
        if (n_mine == C::NPW) pm_wait_vm<C::NPW>();
        else pm_wait_vm<C::NPW - 1>();
    };

    // fragment addressing (ds_read_b64_tr_b16): lane l, read r of a k-step: row = 8 (l >> 5) + 4 r + ((l & 15) >> 2),
    // column (within the 32-channel sub-tile) = 16 ((l >> 4) & 1) + 4 (l & 3)
    const int frow = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int fcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const unsigned a_off = (unsigned)((mg * C::MW * PM_KS + frow) * 64 + fcol * 2);
    const unsigned x_off = (unsigned)(C::A_BYTES + (nb * C::XR + frow) * 64 + fcol * 2);

    if (nstages > 0) issue_stage(0);
    if (nstages > 1) issue_stage(1);
    for (int s = 0; s < nstages; ++s) {
        if (s + 1 < nstages) wait_one_stage_in_flight();      // stage s landed (this wave's share) ...
        else pm_wait_vm<0>();
        __builtin_amdgcn_s_barrier();                         // ... and everybody's; everybody is also done reading stage s - 1,
        if (s + 2 < nstages) issue_stage(s + 2);              // whose buffer takes stage s + 2
        const unsigned sb = lds0 + (unsigned)((s % C::NST) * C::STAGE);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint2 af[C::MW][NP][2], xf[2][2];
            auto read_x = [&](int i, uint2 (&dst)[2]) {
                const int tap = tg + i * C::G;
                const int kh = tap / KW, kw = tap - kh * KW;
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"
                                 : "=v"(dst[r])
                                 : "v"(sb + x_off + (unsigned)((kh * C::NH * C::XR + kw) * 64)), "n"((ks * 16 + r * 4) * 64)
                                 : "memory");
            };
#pragma unroll
            for (int j = 0; j < C::MW; ++j)
#pragma unroll
                for (int t = 0; t < NP; ++t)
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2"
                                     : "=v"(af[j][t][r])
                                     : "v"(sb + a_off), "n"(((t * C::MB + j) * PM_KS + ks * 16 + r * 4) * 64)
                                     : "memory");
            read_x(0, xf[0]);
#pragma unroll
            for (int i = 0; i < C::MAXT; ++i) {
                const int tap = tg + i * C::G;
                // only the last tap index can fall off the end (group tg > (T - 1) % G): everything before it is straight-line code
                if ((C::G - 1) + i * C::G < C::T || tap < C::T) {
                    const bool more = i + 1 < C::MAXT && ((C::G - 1) + (i + 1) * C::G < C::T || tap + C::G < C::T);
                    // one tap ahead: its reads fly under this tap's MFMAs.  The fragment registers are operands of the wait so
                    // that no MFMA reading them can be scheduled above it
#define PM_WAIT(n)                                                                                                     \
    _Pragma("unroll") for (int j_ = 0; j_ < C::MW; ++j_) _Pragma("unroll") for (int t_ = 0; t_ < NP; ++t_)            \
        asm volatile("" : "+v"(af[j_][t_][0]), "+v"(af[j_][t_][1]));                                                   \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(xf[i & 1][0]), "+v"(xf[i & 1][1])::"memory");                      \
    _Pragma("unroll") for (int j_ = 0; j_ < C::MW; ++j_) _Pragma("unroll") for (int t_ = 0; t_ < NP; ++t_)            \
        asm volatile("" : "+v"(af[j_][t_][0]), "+v"(af[j_][t_][1]))
                    if (more) {
                        read_x(i + 1, xf[(i + 1) & 1]);
                        PM_WAIT(2);
                    } else {
                        PM_WAIT(0);
                    }
#undef PM_WAIT
                    const uint4 bq = make_uint4(xf[i & 1][0].x, xf[i & 1][0].y, xf[i & 1][1].x, xf[i & 1][1].y);
#pragma unroll
                    for (int t = 0; t < NP; ++t)
#pragma unroll
                        for (int j = 0; j < C::MW; ++j) {
                            const uint4 aq = make_uint4(af[j][t][0].x, af[j][t][0].y, af[j][t][1].x, af[j][t][1].y);
                            acc[i][j] = pm_mfma<NP>(aq, bq, acc[i][j]);
                        }
                }
            }
        }
    }

    // partial result: part[slice][tap][co][ci]; accumulator register r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
    const int lrow4 = 4 * (lane >> 5), col = lane & 31;
#pragma unroll
    for (int i = 0; i < C::MAXT; ++i) {
        const int tap = tg + i * C::G;
        if (tap < C::T) {
#pragma unroll
            for (int j = 0; j < C::MW; ++j) {
                float* dst = a.part + (((long long)slice * C::T + tap) * a.Cpo + co0 + (mg * C::MW + j) * 32) * a.Cpi + ci0 + nb * 32 + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(long long)((r & 3) + 8 * (r >> 2) + lrow4) * a.Cpi] = acc[i][j][r];
            }
        }
    }
}

// ---- the software-pipelined form for tiles of exactly 8 blocks (128 x 64): one wave = one 32 x 32 block, ALL taps ---------------
// Every wave runs the same straight-line code: per stage 2 k-steps x T taps = 2 T "steps" of three MFMAs (the gradient's hi / mid
// / lo terms) into the tap's accumulator.  The activation fragment of step j + 1 is read under the MFMAs of step j; the gradient
// fragments of the next k-step are read two steps before they are needed — for the last k-step of a stage these come from the
// NEXT stage's buffer, which is why the stage hand-over (vmcnt(0) for this wave's share of stage s + 1, the workgroup barrier, the
// DMA issue for stage s + 2 into the buffer stage s - 1 used) sits three steps before the end of the stage instead of at its
// end: no wave ever waits on LDS latency with an idle matrix pipe except in the prologue.
template <int TM, int TN, int KH, int KW, int NP>
__global__ __launch_bounds__(512) void wgrad_pm_full_kernel(PmArgs a) {
    using C = PmCfg<TM, TN, KH, KW, NP>;
    static_assert(C::MB * C::NH == 8, "one wave per 32 x 32 block");
    constexpr int T = C::T, STEPS = 2 * T, HANDOVER = STEPS - 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mblk = wave % C::MB, nb = wave / C::MB;
    const int per_xcd = (int)(gridDim.x >> 3);                                   // XCD-aware order, as wgrad_pm_kernel
    const int logical = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (logical >= a.total) return;
    const int ci_t = logical % a.tiles_ci, rest = logical / a.tiles_ci;
    const int co_t = rest % a.tiles_co, slice = rest / a.tiles_co;
    const int co0 = co_t * TM, ci0 = ci_t * TN;
    const long long q_begin = (long long)slice * a.slice_pos;
    const int nstages = (int)(a.slice_pos / PM_KS);
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);

    pm_v16f acc[T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    // The DMA is issued by waves 0..3 only — one per SIMD: while it runs its ~50 scalar / DMA instructions after the hand-over
    // barrier, the SIMD's other wave (4..7) keeps the matrix pipe busy, and the mid-stage position of the barrier gives the
    // issuing wave most of a stage to catch up.  A piece = wave-uniform 64-bit base (SGPRs, advanced by scalar adds) + one of
    // two per-lane byte offsets (row within the piece x row pitch + 16-byte chunk): no vector ALU work per piece.
    constexpr int NPD = (C::PIECES + 3) / 4;                                     // pieces per issuing wave
    const bool dma_wave = wave < 4;
    const int prow = lane >> 2, pch = (lane & 3) * 16;
    const unsigned voff_a = (unsigned)(prow * a.Cpo * 2 + pch), voff_x = (unsigned)(prow * a.Cpi * 2 + pch);
    const bool x_tail_lane = 32 + prow < C::XNEED;                               // last piece of a sub-tile: rows 32.. of which XNEED - 32 are read
    const unsigned char* pbase[NPD];
    unsigned pdst[NPD];
    const long long a_step = (long long)PM_KS * a.Cpo * 2, x_step = (long long)PM_KS * a.Cpi * 2;
#pragma unroll
    for (int j = 0; j < NPD; ++j) {
        const int p = j * 4 + (wave & 3);
        if (p < C::A_PIECES) {
            const int t = p / (2 * C::MB), sub = (p >> 1) % C::MB, r16 = p & 1;
            pbase[j] = a.G3 + (((long long)t * a.Qa + q_begin + r16 * 16) * a.Cpo + co0 + sub * 32) * 2;
            pdst[j] = (unsigned)(((t * C::MB + sub) * PM_KS + r16 * 16) * 64);
        } else {
            const int px = (p < C::PIECES ? p : C::A_PIECES) - C::A_PIECES;
            const int sub = px / C::XP_PER, pr = px - sub * C::XP_PER;
            const int kh = sub / C::NH, h = sub - kh * C::NH;
            pbase[j] = a.XP + ((q_begin + (long long)kh * a.kh_rows + pr * 16) * a.Cpi + ci0 + h * 32) * 2;
            pdst[j] = (unsigned)(C::A_BYTES + (sub * C::XR + pr * 16) * 64);
        }
    }
    static_assert(C::XP_PER == 3 && C::XNEED > 32 && C::XNEED <= 48, "the masked tail piece is the third of its sub-tile");
    auto issue_stage = [&](int s) {
        const unsigned base = lds0 + (unsigned)((s % C::NST) * C::STAGE);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0" : "=s"(keep)::"memory");
#pragma unroll
        for (int j = 0; j < NPD; ++j) {
            const int p = j * 4 + (wave & 3);
            if (j == NPD - 1 && p >= C::PIECES) break;
            const bool is_a = p < C::A_PIECES;
            const bool tail = !is_a && (p - C::A_PIECES) % C::XP_PER == C::XP_PER - 1;
            const unsigned long long pb = (unsigned long long)pbase[j];
            const unsigned long long pbu = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(pb >> 32)) << 32) |
                                           (unsigned)__builtin_amdgcn_readfirstlane((unsigned)pb);
            const unsigned dst = __builtin_amdgcn_readfirstlane(base + pdst[j]);
            if (!tail || x_tail_lane)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"(is_a ? voff_a : voff_x), "s"(pbu), "s"(dst) : "memory");
            pbase[j] += is_a ? a_step : x_step;
        }
        asm volatile("s_mov_b32 m0, %0" ::"s"(keep) : "memory");
    };

    const int frow = 8 * (lane >> 5) + ((lane & 15) >> 2);
    const int fcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const unsigned a_off = (unsigned)((mblk * PM_KS + frow) * 64 + fcol * 2);
    const unsigned x_off = (unsigned)(C::A_BYTES + (nb * C::XR + frow) * 64 + fcol * 2);

    uint2 af[2][NP][2], xf[2][2];
#define PM_READ_A(base, ks, dst)                                                                                          \
    _Pragma("unroll") for (int t_ = 0; t_ < NP; ++t_) _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_)                  \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst[t_][r_]) : "v"((base) + a_off),                     \
                     "n"((t_ * C::MB * PM_KS + (ks) * 16 + r_ * 4) * 64) : "memory")
#define PM_READ_X(base, ks, tap, dst)                                                                                     \
    _Pragma("unroll") for (int r_ = 0; r_ < 2; ++r_)                                                                     \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst[r_]) : "v"((base) + x_off),                         \
                     "n"(((((tap) / KW) * C::NH * C::XR + (tap) % KW) + (ks) * 16 + r_ * 4) * 64) : "memory")

    if (nstages <= 0) return;
    if (dma_wave) {
        issue_stage(0);
        if (nstages > 1) issue_stage(1);
    }
    pm_wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    PM_READ_A(lds0, 0, af[0]);
    PM_READ_X(lds0, 0, 0, xf[0]);
    for (int s = 0; s < nstages; ++s) {
        const unsigned sb = lds0 + (unsigned)((s % C::NST) * C::STAGE);
        const unsigned sbn = lds0 + (unsigned)(((s + 1) % C::NST) * C::STAGE);
        const bool has_next = s + 1 < nstages;
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            const int ks = j / T, tap = j - ks * T;
            if (j == HANDOVER && has_next) {
                pm_wait_vm<0>();                                  // this wave's share of stage s + 1 (issued a whole stage ago)
                __builtin_amdgcn_s_barrier();                     // everybody's; and everybody is past stage s - 1
                if (dma_wave && s + 2 < nstages) issue_stage(s + 2);
            }
            // next step's activation fragment, then (two steps ahead of their first use) the next k-step's gradient fragments
            int newer = 0;
            if (j + 1 < STEPS) {
                PM_READ_X(sb, (j + 1) / T, (j + 1) % T, xf[(j + 1) & 1]);
                newer += 2;
            } else if (has_next) {
                PM_READ_X(sbn, 0, 0, xf[(j + 1) & 1]);
                newer += 2;
            }
            if (tap == T - 3) {
                if (ks == 0) {
                    PM_READ_A(sb, 1, af[1]);
                    newer += 2 * NP;
                } else if (has_next) {
                    PM_READ_A(sbn, 0, af[0]);
                    newer += 2 * NP;
                }
            }
            const bool a_prev = tap == T - 2;                     // the gradient reads issued one step ago may still be in flight
            // LDS returns in order: everything older than the `newer` most recent reads has landed.  `newer` is a compile-time
            // number on every path but the last stage, where the skipped prefetches make the wait stricter (never looser)
            // (the fragment registers are operands of the wait so that no MFMA reading them can be scheduled above it)
#define PM_WAIT(n)                                                                                                                \
    do {                                                                                                                          \
        if constexpr (NP == 3)                                                                                                    \
            asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                              \
                         : "+v"(xf[j & 1][0]), "+v"(xf[j & 1][1]), "+v"(af[ks][0][0]), "+v"(af[ks][0][1]), "+v"(af[ks][1][0]),    \
                           "+v"(af[ks][1][1]), "+v"(af[ks][NP - 1][0]), "+v"(af[ks][NP - 1][1])::"memory");                       \
        else                                                                                                                      \
            asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                              \
                         : "+v"(xf[j & 1][0]), "+v"(xf[j & 1][1]), "+v"(af[ks][0][0]), "+v"(af[ks][0][1]), "+v"(af[ks][1][0]),    \
                           "+v"(af[ks][1][1])::"memory");                                                                         \
    } while (0)
            // (the counts: 2 activation fragment reads per step, 2 NP gradient fragment reads per k-step)
            if (!has_next && (j + 1 >= STEPS || (tap >= T - 3 && ks == 1))) PM_WAIT(0);
            else if (newer + (a_prev ? 2 * NP : 0) == 2 + 2 * NP) {
                if constexpr (NP == 3) PM_WAIT(8); else PM_WAIT(6);
            } else PM_WAIT(2);
#undef PM_WAIT
            const uint4 bq = make_uint4(xf[j & 1][0].x, xf[j & 1][0].y, xf[j & 1][1].x, xf[j & 1][1].y);
#pragma unroll
            for (int t = 0; t < NP; ++t) {
                const uint4 aq = make_uint4(af[ks][t][0].x, af[ks][t][0].y, af[ks][t][1].x, af[ks][t][1].y);
                acc[tap] = pm_mfma<NP>(aq, bq, acc[tap]);
            }
        }
    }
#undef PM_READ_A
#undef PM_READ_X

    const int lrow4 = 4 * (lane >> 5), col = lane & 31;
#pragma unroll
    for (int tap = 0; tap < T; ++tap) {
        float* dst = a.part + (((long long)slice * T + tap) * a.Cpo + co0 + mblk * 32) * a.Cpi + ci0 + nb * 32 + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[(long long)((r & 3) + 8 * (r >> 2) + lrow4) * a.Cpi] = acc[tap][r];
    }
}

// One workgroup = one (co, 64-wide ci chunk, tap): its four waves take every fourth K slice (reads coalesced along ci, four
// independent loads in flight per lane), meet in LDS in a fixed order, then the scales and the STE mask are applied and
// dW[co][ci][tap] is written / accumulated.  (Round 6: the first form gave one THREAD a (co, ci) with all taps and let it walk the
// slices alone — 64 x 64 channels were 16 workgroups chasing 256 x 9 dependent loads each: 37 us whatever the layer, as much as the
// gradient kernel itself on the 32 x 32 maps.)  The sum of a (co, ci, tap) is the same expression for every launch geometry.
template <int T, int KW>
__global__ __launch_bounds__(256) void pm_reduce_kernel(const float* __restrict__ part, int nslice, int Cpo, int Cpi,
                                                        int Cout, int Cin, const float* __restrict__ weight, float thr,
                                                        float out_scale, const float* __restrict__ row_scale, int accumulate,
                                                        float* __restrict__ dW, int64_t so, int64_t si, int64_t sh, int64_t sw) {
    __shared__ float fold[4][64];
    const int nchunk = (Cin + 63) >> 6;
    const int64_t total = (int64_t)Cout * nchunk * T;
    const int64_t tap_elems = (int64_t)Cpo * Cpi, slice_elems = (int64_t)T * tap_elems;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    for (int64_t b = blockIdx.x; b < total; b += gridDim.x) {
        const int tap = (int)(b % T);
        const int64_t r = b / T;
        const int chunk = (int)(r % nchunk), co = (int)(r / nchunk);
        const int ci = chunk * 64 + lane;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
        if (ci < Cin) {
            const float* p = part + (int64_t)tap * tap_elems + (int64_t)co * Cpi + ci;
            int sl = g;
            for (; sl + 12 < nslice; sl += 16) {
                a0 += p[(int64_t)sl * slice_elems];
                a1 += p[(int64_t)(sl + 4) * slice_elems];
                a2 += p[(int64_t)(sl + 8) * slice_elems];
                a3 += p[(int64_t)(sl + 12) * slice_elems];
            }
            for (; sl < nslice; sl += 4) a0 += p[(int64_t)sl * slice_elems];
        }
        fold[g][lane] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (g == 0 && ci < Cin) {
            float v = ((fold[0][lane] + fold[1][lane]) + (fold[2][lane] + fold[3][lane])) * out_scale;
            if (row_scale) v *= row_scale[co];                      // the two-plane gradient's per-channel power of two
            // dW and the weight share ONE set of element strides (contiguous, or channels-last like the parameter of a channels_last model)
            const int64_t o = co * so + ci * si + (tap / KW) * sh + (tap % KW) * sw;
            if (weight && !(fabsf(weight[o]) <= thr)) v = 0.0f;
            dW[o] = accumulate ? dW[o] + v : v;
        }
        __syncthreads();
    }
}

template <int TM, int TN, int KH, int KW, int NP, bool FULL = false>
int pm_launch(const PmArgs& a, int nslice, hipStream_t stream) {
    using C = PmCfg<TM, TN, KH, KW, NP>;
    void (*kernel)(PmArgs) = wgrad_pm_kernel<TM, TN, KH, KW, NP>;
    if constexpr (FULL) kernel = wgrad_pm_full_kernel<TM, TN, KH, KW, NP>;
    static QtLdsOnce once;                       // per instantiation (TM, TN, KH, KW, NP, FULL)
    if (qt_ensure_dyn_lds(once, reinterpret_cast<const void*>(kernel), C::LDS) != QT_OK) return QT_ERR_LAUNCH;
    PmArgs b = a;
    b.tiles_co = a.Cpo / TM;
    b.tiles_ci = a.Cpi / TN;
    const long long total = (long long)b.tiles_co * b.tiles_ci * nslice;
    if (total > (1ll << 30)) return QT_ERR_UNSUPPORTED;
    b.total = (int)total;
    hipLaunchKernelGGL(kernel, dim3((unsigned)((total + 7) / 8 * 8)), dim3(512), C::LDS, stream, b);
    return qt_check_launch();
}

}  // namespace

extern "C" {

static int pm_pack_grad_impl(int np, const float* g, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                            int64_t Cout, int64_t Ho, int64_t Wo, int64_t Wq, int64_t Cp, int64_t Qa, uint16_t* G, float* bias_part,
                            const float* scale2, qt_stream_t stream) {
    if (N <= 0 || Cout <= 0 || Ho <= 0 || Wo <= 0 || !g || !G || (np == 2 && !scale2)) return QT_ERR_INVALID_ARG;
    if (Wq < Wo || Cp < Cout || (Cp & 63) || Qa < Ho * N * Wq || (Qa & 31) || !qt_aligned16(G)) return QT_ERR_ALIGNMENT;
    if (Ho * N >= (1ll << 31) || Wq * Cp >= (1ll << 28) || (bias_part && Cp > 2048)) return QT_ERR_UNSUPPORTED;
    if (bias_part && stride_c != 1) return QT_ERR_INVALID_ARG;
    const int64_t tail = (Qa - Ho * N * Wq) * Cp * 2 / 16;
    PmTail tz{G + Ho * N * Wq * Cp, (long long)(Qa * Cp), (long long)tail, np, (int)(Ho * N)};
    const dim3 grid((unsigned)(Ho * N + (tail > 0 ? PM_TAIL_BLOCKS : 0)));
    hipStream_t st = (hipStream_t)stream;
    if (bias_part) {
        if (np == 2)
            hipLaunchKernelGGL(pm_pack_grad_bias_kernel<2>, grid, dim3(256), 0, st, g, stride_n, stride_h, stride_w, (int)N, (int)Cout,
                               (int)Wo, (int)Wq, (int)Cp, Qa, G, bias_part, scale2, tz);
        else
            hipLaunchKernelGGL(pm_pack_grad_bias_kernel<3>, grid, dim3(256), 0, st, g, stride_n, stride_h, stride_w, (int)N, (int)Cout,
                               (int)Wo, (int)Wq, (int)Cp, Qa, G, bias_part, scale2, tz);
    } else {
        if (np == 2)
            hipLaunchKernelGGL(pm_pack_grad_kernel<2>, grid, dim3(256), 0, st, g, stride_n, stride_c, stride_h, stride_w, (int)N,
                               (int)Cout, (int)Wo, (int)Wq, (int)Cp, Qa, G, scale2, tz);
        else
            hipLaunchKernelGGL(pm_pack_grad_kernel<3>, grid, dim3(256), 0, st, g, stride_n, stride_c, stride_h, stride_w, (int)N,
                               (int)Cout, (int)Wo, (int)Wq, (int)Cp, Qa, G, scale2, tz);
    }
    return qt_check_launch();
}

int qt_wgrad_pm_pack_grad_f32(const float* g, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                              int64_t Cout, int64_t Ho, int64_t Wo, int64_t Wq, int64_t Cp, int64_t Qa, uint16_t* G3,
                              qt_stream_t stream) {
    return pm_pack_grad_impl(3, g, stride_n, stride_c, stride_h, stride_w, N, Cout, Ho, Wo, Wq, Cp, Qa, G3, nullptr, nullptr, stream);
}

int qt_wgrad_pm_pack_grad_bias_f32(const float* g, int64_t stride_n, int64_t stride_h, int64_t stride_w, int64_t N, int64_t Cout,
                                   int64_t Ho, int64_t Wo, int64_t Wq, int64_t Cp, int64_t Qa, uint16_t* G3, float* bias_part,
                                   qt_stream_t stream) {
    if (!bias_part) return QT_ERR_INVALID_ARG;
    return pm_pack_grad_impl(3, g, stride_n, 1, stride_h, stride_w, N, Cout, Ho, Wo, Wq, Cp, Qa, G3, bias_part, nullptr, stream);
}

int qt_wgrad_pm_pack_grad_f16x2(const float* g, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                                int64_t Cout, int64_t Ho, int64_t Wo, int64_t Wq, int64_t Cp, int64_t Qa, const float* scale2,
                                uint16_t* G2, float* bias_part, qt_stream_t stream) {
    return pm_pack_grad_impl(2, g, stride_n, stride_c, stride_h, stride_w, N, Cout, Ho, Wo, Wq, Cp, Qa, G2, bias_part, scale2, stream);
}

int qt_wgrad_pm_bias_reduce_f32(float* bias_part, int64_t rows, int64_t Cp, int64_t Cout, int accumulate, float* db,
                                qt_stream_t stream) {
    if (!bias_part || !db || rows <= 0 || Cp <= 0 || Cout <= 0 || Cout > Cp || rows >= (1ll << 31)) return QT_ERR_INVALID_ARG;
    const int64_t segs = rows >= 256 ? (rows / 128 < 128 ? rows / 128 : 128) : 1;
    const float* in = bias_part;
    int64_t in_rows = rows;
    if (segs > 1) {      // first pass into the 128 scratch rows behind the partial sums
        float* scratch = bias_part + rows * Cp;
        const int64_t seg_len = (rows + segs - 1) / segs;
        hipLaunchKernelGGL(pm_bias_reduce_kernel, dim3((unsigned)((Cp + 31) / 32), (unsigned)segs), dim3(256), 0, (hipStream_t)stream, in,
                           (int)rows, (int)seg_len, (int)Cp, (int)Cp, 0, scratch);
        in = scratch;
        in_rows = segs;
    }
    hipLaunchKernelGGL(pm_bias_reduce_kernel, dim3((unsigned)((Cp + 31) / 32), 1u), dim3(256), 0, (hipStream_t)stream, in, (int)in_rows,
                       (int)in_rows, (int)Cp, (int)Cout, accumulate, db);
    return qt_check_launch();
}

static int pm_pack_act_impl(bool f16, const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                           int64_t Cin, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t Wq, int64_t Cp, int64_t Qx, float x_scale,
                           uint16_t* XP, qt_stream_t stream) {
    if (N <= 0 || Cin <= 0 || H <= 0 || W <= 0 || ph < 0 || pw < 0 || !x || !XP) return QT_ERR_INVALID_ARG;
    const int64_t rows = (H + 2 * ph) * N;
    if (Wq < W + 2 * pw || Cp < Cin || (Cp & 31) || Qx < rows * Wq || !qt_aligned16(XP)) return QT_ERR_ALIGNMENT;
    if (rows >= (1ll << 31) || Wq * Cp >= (1ll << 28)) return QT_ERR_UNSUPPORTED;
    const int64_t tail = (Qx - rows * Wq) * Cp * 2 / 16;
    PmTail tz{XP + rows * Wq * Cp, 0ll, (long long)tail, 1, (int)rows};
    const dim3 grid((unsigned)(rows + (tail > 0 ? PM_TAIL_BLOCKS : 0)));
    if (f16)
        hipLaunchKernelGGL(pm_pack_act_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, stride_n, stride_c,
                           stride_h, stride_w, (int)N, (int)Cin, (int)H, (int)W, (int)ph, (int)pw, (int)Wq, (int)Cp, x_scale, XP, tz);
    else
        hipLaunchKernelGGL(pm_pack_act_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, stride_n, stride_c,
                           stride_h, stride_w, (int)N, (int)Cin, (int)H, (int)W, (int)ph, (int)pw, (int)Wq, (int)Cp, x_scale, XP, tz);
    return qt_check_launch();
}

int qt_wgrad_pm_pack_act_f32(const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                             int64_t Cin, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t Wq, int64_t Cp, int64_t Qx,
                             float x_scale, uint16_t* XP, qt_stream_t stream) {
    return pm_pack_act_impl(false, x, stride_n, stride_c, stride_h, stride_w, N, Cin, H, W, ph, pw, Wq, Cp, Qx, x_scale, XP, stream);
}

int qt_wgrad_pm_pack_act_f16(const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                             int64_t Cin, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t Wq, int64_t Cp, int64_t Qx,
                             float x_scale, uint16_t* XP, qt_stream_t stream) {
    return pm_pack_act_impl(true, x, stride_n, stride_c, stride_h, stride_w, N, Cin, H, W, ph, pw, Wq, Cp, Qx, x_scale, XP, stream);
}

int qt_wgrad_pm_pack_act_s2d_f32(const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                                 int64_t C, int64_t H, int64_t W, int64_t s, int64_t ph, int64_t pw, int64_t Hs, int64_t Ws,
                                 int64_t Wq, int64_t Cs8, int64_t Cp, int64_t Qx, uint16_t* XP, qt_stream_t stream) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || s <= 0 || ph < 0 || pw < 0 || Hs <= 0 || Ws <= 0 || !x || !XP) return QT_ERR_INVALID_ARG;
    const int64_t rows = Hs * N;
    if (Wq < Ws || (Cs8 & 7) || Cs8 < C * s * s || Cp < 3 * Cs8 || (Cp & 31) || Qx < rows * Wq || !qt_aligned16(XP)) return QT_ERR_ALIGNMENT;
    if (rows >= (1ll << 31) || Wq * Cp >= (1ll << 28) || H * s >= (1ll << 30) || W * s >= (1ll << 30)) return QT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(pm_pack_act_s2d_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, x, stride_n, stride_c, stride_h,
                       stride_w, (int)N, (int)C, (int)H, (int)W, (int)s, (int)ph, (int)pw, (int)Ws, (int)Wq, (int)Cs8, (int)Cp, XP);
    const int64_t tail = (Qx - rows * Wq) * Cp * 2 / 16;
    if (tail > 0)
        hipLaunchKernelGGL(pm_zero_kernel, dim3(qt_stream_grid((tail + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<uint4*>(XP + rows * Wq * Cp), tail);
    return qt_check_launch();
}

// qt_wgrad_pm_pack_act_s2d_f32 with the image as two fp16 terms of x / scale2[0] in two channel groups (Cp >= 2 Cs8): the
// activation plane of qt_wgrad_pm_f16; the caller adds the two groups of the result and multiplies by scale2[0].
int qt_wgrad_pm_pack_act_s2d_f16x2(const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N,
                                   int64_t C, int64_t H, int64_t W, int64_t s, int64_t ph, int64_t pw, int64_t Hs, int64_t Ws,
                                   int64_t Wq, int64_t Cs8, int64_t Cp, int64_t Qx, const float* scale2, uint16_t* XP,
                                   qt_stream_t stream) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || s <= 0 || ph < 0 || pw < 0 || Hs <= 0 || Ws <= 0 || !x || !XP || !scale2) return QT_ERR_INVALID_ARG;
    const int64_t rows = Hs * N;
    if (Wq < Ws || (Cs8 & 7) || Cs8 < C * s * s || Cp < 2 * Cs8 || (Cp & 31) || Qx < rows * Wq || !qt_aligned16(XP)) return QT_ERR_ALIGNMENT;
    if (rows >= (1ll << 31) || Wq * Cp >= (1ll << 28) || H * s >= (1ll << 30) || W * s >= (1ll << 30)) return QT_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int64_t E = C * s * s, rowf4 = (W * C + 3) & ~3ll;
    const int64_t lds = s * rowf4 * 4 + E * 8;
    if (stride_c == 1 && stride_w == C && lds <= 60 * 1024 && stride_h >= W * C) {
        const int vec_ok = ((W * C) % 4 == 0) && qt_aligned16(x) && (stride_h % 4 == 0) && (stride_n % 4 == 0);
        hipLaunchKernelGGL(pm_pack_act_s2d_f16_rows_kernel, dim3((unsigned)rows), dim3(256), (size_t)lds, st, x, stride_n, stride_h, scale2,
                           (int)N, (int)C, (int)H, (int)W, (int)s, (int)ph, (int)pw, (int)Ws, (int)Wq, (int)Cs8, (int)Cp, vec_ok, XP);
    } else {
        hipLaunchKernelGGL(pm_pack_act_s2d_f16_kernel, dim3((unsigned)rows), dim3(256), 0, st, x, stride_n, stride_c, stride_h, stride_w,
                           scale2, (int)N, (int)C, (int)H, (int)W, (int)s, (int)ph, (int)pw, (int)Ws, (int)Wq, (int)Cs8, (int)Cp, XP);
    }
    const int64_t tail = (Qx - rows * Wq) * Cp * 2 / 16;
    if (tail > 0)
        hipLaunchKernelGGL(pm_zero_kernel, dim3(qt_stream_grid((tail + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<uint4*>(XP + rows * Wq * Cp), tail);
    return qt_check_launch();
}

// part[nslice][kh * kw][Cpo][Cpi] = per-slice partial gradients; Qa = nslice * slice_pos, slice_pos % 32 == 0;
// XP must hold (kh - 1) * kh_rows + Qa + 48 rows (kw - 1 are read).  Supported: (kh, kw) = (3, 3) with Cpi % 64 == 0, (5, 5) with Cpi % 32 == 0.
static int pm_run(int np, const uint16_t* G, const uint16_t* XP, float* part, int64_t Qa, int64_t kh_rows, int64_t nslice, int64_t Cpo,
                  int64_t Cpi, int64_t kh, int64_t kw, qt_stream_t stream) {
    if (!G || !XP || !part || Qa <= 0 || nslice <= 0 || nslice > 65535 || kh_rows <= 0) return QT_ERR_INVALID_ARG;
    if ((Qa % (32 * nslice)) || (Cpo & 63) || !qt_aligned16(G) || !qt_aligned16(XP) || !qt_aligned16(part)) return QT_ERR_ALIGNMENT;
    PmArgs a;
    a.G3 = reinterpret_cast<const unsigned char*>(G);
    a.XP = reinterpret_cast<const unsigned char*>(XP);
    a.part = part; a.Qa = Qa; a.kh_rows = kh_rows; a.slice_pos = Qa / nslice; a.Cpo = (int)Cpo; a.Cpi = (int)Cpi;
    hipStream_t s = (hipStream_t)stream;
    const int ns = (int)nslice;
    if (kh == 3 && kw == 3 && !(Cpi & 63)) {
        if (np == 3) return (Cpo & 127) ? pm_launch<64, 64, 3, 3, 3>(a, ns, s) : pm_launch<128, 64, 3, 3, 3, true>(a, ns, s);
        return (Cpo & 127) ? pm_launch<64, 64, 3, 3, 2>(a, ns, s) : pm_launch<128, 64, 3, 3, 2, true>(a, ns, s);
    }
    if (kh == 5 && kw == 5 && !(Cpi & 31)) return np == 3 ? pm_launch<64, 32, 5, 5, 3>(a, ns, s) : pm_launch<64, 32, 5, 5, 2>(a, ns, s);
    return QT_ERR_UNSUPPORTED;
}

int qt_wgrad_pm_f32(const uint16_t* G3, const uint16_t* XP, float* part, int64_t Qa, int64_t kh_rows, int64_t nslice,
                    int64_t Cpo, int64_t Cpi, int64_t kh, int64_t kw, qt_stream_t stream) {
    return pm_run(3, G3, XP, part, Qa, kh_rows, nslice, Cpo, Cpi, kh, kw, stream);
}

// the same contraction over TWO fp16 gradient planes (qt_wgrad_pm_pack_grad_f16x2) and an fp16 activation plane
// (qt_wgrad_pm_pack_act_f16); the caller multiplies the reduced result by scale2[0]
int qt_wgrad_pm_f16(const uint16_t* G2, const uint16_t* XP, float* part, int64_t Qa, int64_t kh_rows, int64_t nslice,
                    int64_t Cpo, int64_t Cpi, int64_t kh, int64_t kw, qt_stream_t stream) {
    return pm_run(2, G2, XP, part, Qa, kh_rows, nslice, Cpo, Cpi, kh, kw, stream);
}

int qt_wgrad_pm_reduce_f32(const float* part, int64_t nslice, int64_t taps, int64_t Cpo, int64_t Cpi, int64_t Cout, int64_t Cin,
                           const float* weight, float ste_threshold, float out_scale, const float* row_scale, int accumulate,
                           float* dW, int64_t stride_o, int64_t stride_i, int64_t stride_h, int64_t stride_w, qt_stream_t stream) {
    if (!part || !dW || nslice <= 0 || taps <= 0 || Cout <= 0 || Cin <= 0 || Cpo < Cout || Cpi < Cin) return QT_ERR_INVALID_ARG;
    if (taps != 9 && taps != 25) return QT_ERR_UNSUPPORTED;       // the kernels this reduce serves: 3 x 3 and 5 x 5
    const dim3 grid(qt_stream_grid(Cout * ((Cin + 63) / 64) * taps, 256 * 8));
    if (taps == 9)
        hipLaunchKernelGGL((pm_reduce_kernel<9, 3>), grid, dim3(256), 0, (hipStream_t)stream, part, (int)nslice, (int)Cpo, (int)Cpi,
                           (int)Cout, (int)Cin, weight, ste_threshold, out_scale, row_scale, accumulate, dW, stride_o, stride_i, stride_h,
                           stride_w);
    else
        hipLaunchKernelGGL((pm_reduce_kernel<25, 5>), grid, dim3(256), 0, (hipStream_t)stream, part, (int)nslice, (int)Cpo, (int)Cpi,
                           (int)Cout, (int)Cin, weight, ste_threshold, out_scale, row_scale, accumulate, dW, stride_o, stride_i, stride_h,
                           stride_w);
    return qt_check_launch();
}

}  // extern "C"
