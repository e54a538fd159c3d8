// fp16 "pair planes": the two-term form of the real-valued-activation path (include/qt_hip.h, round 3).
//   activations: x / s = hi + lo + e with hi = fp16_rn(x / s), lo = fp16_rn(x / s - hi); s = 2^k is a PER-TENSOR scale chosen
//     on the device (qt_f16x2_scale_f32) so that max|x| / s lies in [2^14, 2^15): nothing overflows fp16 (max 65504), the
//     division is exact, x / s - hi is exact in fp32 (it has at most 13 significant bits), and lo is a normal fp16 number for
//     every |x| >= 2^-17 max|x| (below that it is rounded on the subnormal grid 2^-24).  Hence
//         |x - s (hi + lo)| <= max(2^-22 |x|, 2^-39 max|x|)                                  (2 x 11 significand bits)
//     stored as consecutive pairs: fp16 slot 2k + t of a row is term t of x[k].  (The speculative first-layer pack below may keep
//     a FIXED scale with max|x| / s anywhere in [2^8, 2^16): the floor term is then 2^-33 max|x|.)
//   weights: a value that is exact in fp16 — the quantised q in {-1, 0, +1} (safeSign / ternary / torch.sign) or an integer
//     level |q| <= 2048 ("raw": k-bit DoReFa levels) — replicated twice.
// An fp16 MFMA GEMM (v_mfma_f32_32x32x16_f16, exact products, fp32 accumulate) over 2K of these planes, times s, equals the
// fp32 GEMM of x with the quantised weight to the bound above: normalised error ~1e-7, two orders inside the 1e-5 bar of
// SURVEY 8(d) for real-valued inputs, at 2/3 of the matrix work and operand bytes of the exact three-term bf16 route
// (split_bf16.hip), which stays selectable.  HBM-bound elementwise kernels: 4 B in, 4 B out per element.
#include "qt_common.h"

namespace {

__device__ __forceinline__ uint32_t f16_bits(float f) {
    const _Float16 h = (_Float16)f;            // v_cvt_f16_f32: round to nearest even, subnormals kept
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
__device__ __forceinline__ float f16_bits_to_f32(uint32_t b) {
    const unsigned short u = (unsigned short)b;
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
}
// (hi | lo << 16) of v (already divided by the scale)
__device__ __forceinline__ uint32_t split2(float v) {
    const uint32_t hi = f16_bits(v);
    const uint32_t lo = f16_bits(v - f16_bits_to_f32(hi));
    return hi | (lo << 16);
}

__device__ __forceinline__ void write_scale(float a, float* __restrict__ scale2) {
    float s = 1.0f;
    if (a > 0.0f && a < __builtin_huge_valf()) {
        int e;
        frexpf(a, &e);                          // a = m 2^e, m in [0.5, 1): floor(log2 a) = e - 1
        int k = e - 15;                         // a / 2^k in [2^14, 2^15)
        k = k < -100 ? -100 : (k > 100 ? 100 : k);
        s = ldexpf(1.0f, k);
    }
    scale2[0] = s;
    scale2[1] = 1.0f / s;
}

__global__ void scale_kernel(const float* __restrict__ mn, const float* __restrict__ mx, float* __restrict__ scale2) {
    write_scale(fmaxf(fabsf(*mn), fabsf(*mx)), scale2);
}

// max|x| over a dense tensor at the HBM rate: every workgroup writes the maximum of its share (the bit pattern of |x|:
// non-negative floats order like unsigned integers, NaN above everything) to part[blockIdx]; a second, one-workgroup launch
// folds the partials and writes scale2.  (One launch with a device-scope atomicMax per workgroup was 85 us for 154 MB: 2048
// same-address atomics serialise at ~40 ns each across the XCDs.)
__global__ __launch_bounds__(256) void absmax_part_kernel(const float* __restrict__ x, int64_t n, unsigned* __restrict__ part) {
    unsigned m = 0;
    const int64_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
#define QT_FOLD(v) m = max(max(m, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu, \
            max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)))
    for (; i + 3 * stride < n4; i += 4 * stride) {          // four independent 16-byte loads in flight per lane
        const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
        QT_FOLD(a); QT_FOLD(b); QT_FOLD(c); QT_FOLD(d);
    }
    for (; i < n4; i += stride) {
        const float4 a = x4[i];
        QT_FOLD(a);
    }
#undef QT_FOLD
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, __float_as_uint(x[(n4 << 2) + threadIdx.x]) & 0x7fffffffu);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    __shared__ unsigned sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
}

__global__ __launch_bounds__(256) void absmax_final_kernel(const unsigned* __restrict__ part, int nparts, float* __restrict__ scale2) {
    unsigned m = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) m = max(m, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    __shared__ unsigned sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) write_scale(__uint_as_float(max(max(sh[0], sh[1]), max(sh[2], sh[3]))), scale2);
}

// Fold of the speculative pack's partial maxima: a = max|x|.  The fixed scale s0 = 1 / spec_inv is ADMISSIBLE when a / s0 lies in
// [2^8, 2^15) (nothing overflows fp16 — its largest finite value is 65504 and v_cvt_f16_f32 rounds anything >= 65520 to inf,
// so the window stops at write_scale's own target binade [2^14, 2^15) instead of 2^16; the subnormal grid 2^-24 s0 of the low term is <= 2^-32 a, i.e. the split's error is
// max(2^-22 |x|, 2^-33 max|x|)), or when a == 0: then scale2 = [s0, 1 / s0] and *redo = 0.  Otherwise scale2 is the exact-binade
// scale of write_scale and *redo = 1: the repack launch behind this one rewrites the plane with it.
__global__ __launch_bounds__(1024) void spec_final_kernel(const unsigned* __restrict__ part, int nparts, float spec_inv,
                                                          float* __restrict__ scale2, int* __restrict__ redo) {
    unsigned m = 0;
    for (int i = threadIdx.x; i < nparts; i += 1024) m = max(m, part[i]);       // AlexNet: 14592 partials, 15 per thread
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    __shared__ unsigned sh[16];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned mm = sh[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) mm = max(mm, sh[j]);
        const float a = __uint_as_float(mm);
        const float r = a * spec_inv;                                     // exact: spec_inv is a power of two
        const bool ok = a == 0.0f || (r >= 256.0f && r < 32768.0f);      // NaN / inf: not ok
        if (ok) {
            scale2[0] = 1.0f / spec_inv;
            scale2[1] = spec_inv;
        } else {
            write_scale(a, scale2);
        }
        *redo = ok ? 0 : 1;
    }
}

// ---- per-CHANNEL scales (the weight gradient: one output row of dW per gradient channel) -------------------------------------
// A weight-gradient row only sees ITS channel of the gradient, so a per-tensor scale would cost the channels far below the
// tensor's maximum their low bits (2^-39 max|g| absolute, per element); with s[c] chosen from max|g[:, c]| every row keeps
// the bound relative to its own channel, like an fp32 GEMM.  scale2c = [s[0..Cp) | 1 / s[0..Cp)] (1 for padding channels).
//
// channels-last, dense: the gradient is a [P][C] matrix.  A workgroup = rpb rows x cw channel groups (VEC channels each, cw =
// min(C / VEC, 256)); it walks rows bx * rpb + lr, + gridDim.x * rpb, ... (whole contiguous rows when cw covers the row), folds
// its rpb partial rows through LDS and writes part[bx][C].
template <int VEC>
__global__ __launch_bounds__(256) void absmax_ch_rows_kernel(const float* __restrict__ g, int64_t P, int C, int cw, int rpb,
                                                             unsigned* __restrict__ part) {
    __shared__ unsigned sm[256 * VEC];
    const int tid = threadIdx.x, lr = tid / cw, lq = tid - lr * cw;
    const int cq = C / VEC, q = blockIdx.y * cw + lq;
    unsigned m[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) m[e] = 0;
    if (lr < rpb && q < cq) {
        const int64_t step = (int64_t)gridDim.x * rpb;
        int64_t r = (int64_t)blockIdx.x * rpb + lr;
        if constexpr (VEC == 4) {
            const float4* g4 = reinterpret_cast<const float4*>(g);
#define QT_FOLD4(v) do { m[0] = max(m[0], __float_as_uint(v.x) & 0x7fffffffu); m[1] = max(m[1], __float_as_uint(v.y) & 0x7fffffffu); \
                         m[2] = max(m[2], __float_as_uint(v.z) & 0x7fffffffu); m[3] = max(m[3], __float_as_uint(v.w) & 0x7fffffffu); } while (0)
            for (; r + 3 * step < P; r += 4 * step) {
                const float4 a = g4[r * cq + q], b = g4[(r + step) * cq + q], c = g4[(r + 2 * step) * cq + q], d = g4[(r + 3 * step) * cq + q];
                QT_FOLD4(a); QT_FOLD4(b); QT_FOLD4(c); QT_FOLD4(d);
            }
            for (; r < P; r += step) {
                const float4 a = g4[r * cq + q];
                QT_FOLD4(a);
            }
#undef QT_FOLD4
        } else {
            for (; r < P; r += step) m[0] = max(m[0], __float_as_uint(g[r * C + q]) & 0x7fffffffu);
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) sm[tid * VEC + e] = m[e];
    __syncthreads();
    if (lr == 0 && q < cq) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            unsigned t = m[e];
            for (int j = 1; j < rpb; ++j) t = max(t, sm[(j * cw + lq) * VEC + e]);
            part[(int64_t)blockIdx.x * C + q * VEC + e] = t;
        }
    }
}

// any other layout: workgroup (c, j) walks images j, j + gridDim.y, ... of channel c
__global__ __launch_bounds__(256) void absmax_ch_strided_kernel(const float* __restrict__ g, int64_t sn, int64_t sc, int64_t sh,
                                                                int64_t sw, int N, int C, int H, int W, unsigned* __restrict__ part) {
    const int c = blockIdx.x;
    unsigned m = 0;
    const int hw = H * W;
    for (int n = blockIdx.y; n < N; n += gridDim.y) {
        const float* base = g + (int64_t)n * sn + (int64_t)c * sc;
        for (int i = threadIdx.x; i < hw; i += 256) {
            const int y = i / W, x = i - y * W;
            m = max(m, __float_as_uint(base[(int64_t)y * sh + (int64_t)x * sw]) & 0x7fffffffu);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    __shared__ unsigned sh4[4];
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[(int64_t)blockIdx.y * C + c] = max(max(sh4[0], sh4[1]), max(sh4[2], sh4[3]));
}

// part[nparts][C] -> scale2c[2 Cp]: 64 channels x 16 row groups per workgroup (a thread folds nparts / 16 <= 32 partials)
__global__ __launch_bounds__(1024) void absmax_ch_final_kernel(const unsigned* __restrict__ part, int nparts, int C, int Cp,
                                                               float* __restrict__ scale2c) {
    __shared__ unsigned sm[16][64];
    const int lc = threadIdx.x & 63, rg = threadIdx.x >> 6, c = blockIdx.x * 64 + lc;
    unsigned m = 0;
    if (c < C) {
#pragma unroll 8
        for (int r = rg; r < nparts; r += 16) m = max(m, part[(int64_t)r * C + c]);
    }
    sm[rg][lc] = m;
    __syncthreads();
    if (rg == 0 && c < Cp) {
        float pair[2] = {1.0f, 1.0f};
        if (c < C) {
#pragma unroll
            for (int j = 1; j < 16; ++j) m = max(m, sm[j][lc]);
            write_scale(__uint_as_float(m), pair);
        }
        scale2c[c] = pair[0];
        scale2c[Cp + c] = pair[1];
    }
}

// mode 0 = activation split (x * scale2[1]), 1 = safeSign weight, 2 = ternary weight, 3 = torch.sign weight, 4 = raw weight
template <int MODE>
__global__ __launch_bounds__(256) void pair_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ scale2,
                                                   uint32_t* __restrict__ out, int64_t ld_words, int64_t rows, int64_t K) {
    const float inv = (MODE == 0 && scale2) ? scale2[1] : 1.0f;
    const int64_t quads = ld_words / 4;                      // one work item = 4 elements = 8 fp16 = 16 B
    const int64_t total = rows * quads;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / quads, q = t - row * quads;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t k = q * 4 + e;
            if (k >= K) continue;
            const float v = x[row * ldx + k];
            if (MODE == 0) {
                w[e] = split2(v * inv);
            } else {
                float qv;
                if (MODE == 1) qv = qt_safe_sign(v);
                else if (MODE == 2) qv = qt_ternarize(v);
                else if (MODE == 3) qv = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
                else qv = v;
                const uint32_t b = f16_bits(qv);
                w[e] = b | (b << 16);
            }
        }
        reinterpret_cast<uint4*>(out + row * ld_words)[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// The activation split with the scale's fold inside (round 6: training-step launch diet).  Every workgroup folds the <= 2048
// partial maxima of absmax_part_kernel itself (8 KB from L2, the same value in every workgroup: max is order-free) instead of a
// one-workgroup launch in front of it; workgroup 0 leaves scale3 = [s, 1 / s, s * (*mul_dev or 1)] behind for the consumer — the
// third slot is the GEMM's device scale when the contraction is multiplied by another device scalar anyway (DoReFa's E = mean|W|:
// s is a power of two, so s * E is E with another exponent — exact), which used to be a torch launch of its own.
__global__ __launch_bounds__(256) void pair_fold_kernel(const float* __restrict__ x, int64_t ldx, const unsigned* __restrict__ part,
                                                        int nparts, const float* __restrict__ mul_dev, float* __restrict__ scale3,
                                                        uint32_t* __restrict__ out, int64_t ld_words, int64_t rows, int64_t K) {
    unsigned m = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) m = max(m, part[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    __shared__ unsigned sh[4];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    float pair[2];
    write_scale(__uint_as_float(max(max(sh[0], sh[1]), max(sh[2], sh[3]))), pair);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        scale3[0] = pair[0];
        scale3[1] = pair[1];
        scale3[2] = mul_dev ? pair[0] * *mul_dev : pair[0];
    }
    const float inv = pair[1];
    const int64_t quads = ld_words / 4;
    const int64_t total = rows * quads;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / quads, q = t - row * quads;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t k = q * 4 + e;
            if (k < K) w[e] = split2(x[row * ldx + k] * inv);
        }
        reinterpret_cast<uint4*>(out + row * ld_words)[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// Space-to-depth gather + split (see split_bf16.hip: s2d_triple_kernel / s2d_triple_rows_kernel for the geometry):
// pixel (n, Y, X) of the output plane holds, for e = (c*s + dy)*s + dx, the pair of x[n, c, s*Y + dy - ph, s*X + dx - pw] / scale.
__global__ __launch_bounds__(256) void s2d_pair_kernel(const float* __restrict__ x, int64_t sN, int64_t sC, int64_t sH,
                                                       int64_t sW, const float* __restrict__ scale2,
                                                       uint32_t* __restrict__ out, int64_t ld_words, int64_t N, int C, int H,
                                                       int W, int s, int ph, int pw, int Hs, int Ws) {
    const float inv = scale2 ? scale2[1] : 1.0f;
    const int E = C * s * s;
    const int64_t quads = ld_words / 4;
    const int64_t total = N * Hs * Ws * quads;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = t / quads, q = t - pix * quads;
        const int64_t n = pix / ((int64_t)Hs * Ws);
        const int rem = (int)(pix - n * Hs * Ws);
        const int Y = rem / Ws, X = rem - Y * Ws;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = (int)q * 4 + j;
            if (e >= E) continue;
            const int c = e / (s * s), r = e - c * s * s, dy = r / s, dx = r - dy * s;
            const int hh = s * Y + dy - ph, ww = s * X + dx - pw;
            float v = 0.0f;
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) v = x[n * sN + c * sC + hh * sH + ww * sW];
            w[j] = split2(v * inv);
        }
        reinterpret_cast<uint4*>(out + pix * ld_words)[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// channels-last images: one workgroup per output row (n, Y); the s input rows go through LDS with full-line loads, the
// output row leaves as coalesced 16-byte stores (4 elements x 2 terms each)
//
// Speculative form (qt_f16x2_s2d_pack_spec_f32): ``spec_inv`` != 0 packs with the FIXED scale 1 / spec_inv and folds max|x| of the
// rows it stages into part[blockIdx] on the way (the kernel is HBM-bound: the maximum is free), so that the separate max|x| pass
// over the image — 38 us of AlexNet's 0.95 ms forward — disappears whenever the fixed scale turns out to be admissible;
// ``run_if`` != NULL: the repack launch, which returns at once unless *run_if != 0.
__global__ __launch_bounds__(256) void s2d_pair_rows_kernel(const float* __restrict__ x, int64_t sN, int64_t sH,
                                                            const float* __restrict__ scale2, uint32_t* __restrict__ out,
                                                            int64_t ld_words, int C, int H, int W, int s, int ph, int pw,
                                                            int Hs, int Ws, int vec_ok, float spec_inv,
                                                            unsigned* __restrict__ part, const int* __restrict__ run_if, int nrows_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s2d_smem[];
    if (run_if && *run_if == 0) return;                                   // uniform, before any barrier
    const float inv = spec_inv != 0.0f ? spec_inv : (scale2 ? scale2[1] : 1.0f);
    unsigned amax = 0;
    const int E = C * s * s, rowf = W * C, rowf4 = (rowf + 3) & ~3;
    float* rows = reinterpret_cast<float*>(s2d_smem);                     // [s][rowf4]
    int* lut_off = reinterpret_cast<int*>(rows + (size_t)s * rowf4);      // [E] dy*rowf4 + (dx - pw)*C + c
    int* lut_dx = lut_off + E;                                            // [E] dx - pw
    const int tid = threadIdx.x;
    for (int e = tid; e < E; e += 256) {
        const int c = e / (s * s), r = e - c * s * s, dy = r / s, dx = r - dy * s;
        lut_off[e] = dy * rowf4 + (dx - pw) * C + c;
        lut_dx[e] = dx - pw;
    }
    // one output row (n, Y) per workgroup when the grid covers nrows_out; the repack launch uses a small grid (its workgroups
    // normally return above: 14592 empty workgroups cost ~8 us, 1024 cost ~2) and walks the rows
    for (int blk = blockIdx.x; blk < nrows_out; blk += gridDim.x) {
    const int n = blk / Hs, Y = blk - n * Hs;
    amax = 0;
    for (int dy = 0; dy < s; ++dy) {
        const int hh = s * Y + dy - ph;
        const bool ok = hh >= 0 && hh < H;
        const float* src = x + (int64_t)n * sN + (int64_t)(ok ? hh : 0) * sH;
        float* dst = rows + dy * rowf4;
        if (vec_ok) {
            for (int i = tid * 4; i < rowf; i += 1024) {
                const float4 v = ok ? *reinterpret_cast<const float4*>(src + i) : make_float4(0, 0, 0, 0);
                *reinterpret_cast<float4*>(dst + i) = v;
                amax = max(max(amax, __float_as_uint(v.x) & 0x7fffffffu), max(__float_as_uint(v.y) & 0x7fffffffu,
                           max(__float_as_uint(v.z) & 0x7fffffffu, __float_as_uint(v.w) & 0x7fffffffu)));
            }
        } else {
            for (int i = tid; i < rowf; i += 256) {
                const float v = ok ? src[i] : 0.0f;
                dst[i] = v;
                amax = max(amax, __float_as_uint(v) & 0x7fffffffu);
            }
        }
    }
    if (part) {                                                           // workgroup maximum of |x| (bit patterns order like uints)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) amax = max(amax, (unsigned)__shfl_xor((int)amax, o));
        __shared__ unsigned wmax[4];
        if ((tid & 63) == 0) wmax[tid >> 6] = amax;
        __syncthreads();
        if (tid == 0) part[blk] = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
    }
    __syncthreads();
    const int quads = (int)(ld_words >> 2), total = Ws * quads;
    uint4* orow = reinterpret_cast<uint4*>(out + ((int64_t)blk * Ws) * ld_words);
    for (int q = tid; q < total; q += 256) {
        const int X = q / quads, cq = q - X * quads;
        const int base = s * X * C, wx = s * X;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = cq * 4 + j;
            float v = 0.0f;
            if (e < E) {
                const int ww = wx + lut_dx[e];
                if ((unsigned)ww < (unsigned)W) v = rows[lut_off[e] + base];
            }
            w[j] = split2(v * inv);
        }
        orow[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();                                                      // the staged rows (and wmax) are re-used by the next row
    }
}


// The conv weight as the GEMM operand of the pair-plane convs, straight from its [Cout][Cin][kh][kw] storage (any strides: contiguous
// or channels-last): row r, tap (i, j),
// channel c hold q(w[...]) twice (fp16), q = the quantiser of `mode` (1 safeSign, 2 ternary, 3 torch.sign, 4 the value itself):
//   forward operand  (transpose_flip = 0): rows = Cout, channels = Cin:  w[r][c][i][j]
//   grad_x operand   (transpose_flip = 1): rows = Cin, channels = Cout: w[c][r][kh-1-i][kw-1-j]   (flipped, transposed weight)
// Row layout as ops.pack_conv_weight_bf16x3 builds it: tap-major, Cb = 4 * channels bytes rounded to 16 per tap, the row padded
// with zeros to ld_bytes.  (Was: quantise, flip, transpose copy, permute copy, pack — five launches and three copies per conv.)
__global__ __launch_bounds__(256) void conv_weight_pair_kernel(const float* __restrict__ w, int64_t so, int64_t si, int64_t sh, int64_t sw,
                                                               int Cout, int Cin, int kh, int kw, int mode, int transpose_flip,
                                                               uint32_t* __restrict__ out, int64_t ld_words, int cb_words) {
    const int rows = transpose_flip ? Cin : Cout, chans = transpose_flip ? Cout : Cin, taps = kh * kw;
    const int64_t total = (int64_t)rows * ld_words;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(t / ld_words), word = (int)(t - (int64_t)r * ld_words);
        const int tap = word / cb_words, c = word - tap * cb_words;
        uint32_t v = 0;
        if (tap < taps && c < chans) {
            const int i = tap / kw, j = tap - i * kw;
            const float x = transpose_flip ? w[c * so + r * si + (kh - 1 - i) * sh + (kw - 1 - j) * sw]
                                           : w[r * so + c * si + i * sh + j * sw];
            float q;
            if (mode == 1) q = qt_safe_sign(x);
            else if (mode == 2) q = qt_ternarize(x);
            else if (mode == 3) q = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
            else q = x;
            const uint32_t b = f16_bits(q);
            v = b | (b << 16);
        }
        out[t] = v;
    }
}

// The transposed 1 x 1 case with unit stride along Cin (a Linear weight [N, K] whose plane of Q(W)^T is wanted: the operand of
// grad_x = g . Q(W), functions/binary_connect.py:104-112): the generic kernel above would read one float per 4 K-byte row stride —
// every lane its own cache line.  Here a 64 x 64 tile goes through LDS: coalesced reads along K, coalesced pair writes along N.
__global__ __launch_bounds__(256) void weight_pair_transpose_kernel(const float* __restrict__ w, int64_t so, int N, int K, int mode,
                                                                    uint32_t* __restrict__ out, int64_t ld_words) {
    __shared__ float tile[64][65];
    const int tn = (N + 63) / 64, tk = (K + 63) / 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;                 // 64 x 4
    for (int64_t t = blockIdx.x; t < (int64_t)tn * tk; t += gridDim.x) {
        const int n0 = (int)(t % tn) * 64, k0 = (int)(t / tn) * 64;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int n = n0 + ly + 4 * j, k = k0 + lx;
            tile[ly + 4 * j][lx] = (n < N && k < K) ? w[(int64_t)n * so + k] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = k0 + ly + 4 * j, n = n0 + lx;
            if (k < K && n < N) {
                const float x = tile[lx][ly + 4 * j];
                float q;
                if (mode == 1) q = qt_safe_sign(x);
                else if (mode == 2) q = qt_ternarize(x);
                else if (mode == 3) q = x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f);
                else q = x;
                const uint32_t b = f16_bits(q);
                out[(int64_t)k * ld_words + n] = b | (b << 16);
            }
        }
        __syncthreads();
    }
}

// the words of a row beyond `chans` (row padding to the 128-byte stride) of the plane written by the kernel above
__global__ __launch_bounds__(256) void pair_row_pad_kernel(uint32_t* __restrict__ out, int64_t ld_words, int chans, int64_t rows) {
    const int pad = (int)(ld_words - chans);
    const int64_t total = rows * pad;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / pad;
        out[r * ld_words + chans + (int)(t - r * pad)] = 0u;
    }
}

}  // namespace

extern "C" int qt_f16x2_scale_f32(const float* mn, const float* mx, float* scale2, qt_stream_t stream) {
    if (!mn || !mx || !scale2) return QT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(scale_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, mn, mx, scale2);
    return qt_check_launch();
}

extern "C" int64_t qt_f16x2_absmax_work_words() { return 2048; }

extern "C" int qt_f16x2_absmax_scale_f32(const float* x, int64_t n, uint32_t* work, float* scale2, qt_stream_t stream) {
    if (n < 0 || !work || !scale2 || (n > 0 && !x)) return QT_ERR_INVALID_ARG;
    if (!qt_aligned16(x)) return QT_ERR_ALIGNMENT;
    const int grid = qt_stream_grid(((n >> 2) + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(absmax_part_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, work);
    hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, work, grid, scale2);
    return qt_check_launch();
}

extern "C" int qt_f16x2_absmax_pack_f32(const float* x, int64_t rows, int64_t K, uint32_t* work, const float* mul_dev, float* scale3,
                                        uint16_t* out, int64_t ld_bytes, qt_stream_t stream) {
    if (rows < 0 || K < 0 || !work || !scale3) return QT_ERR_INVALID_ARG;
    if (rows > 0 && K > 0 && (!x || !out)) return QT_ERR_INVALID_ARG;
    if (!qt_aligned16(x) || ld_bytes < 4 * K || (ld_bytes & 15) || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    const int64_t n = rows * K;
    const int gridp = qt_stream_grid(((n >> 2) + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(absmax_part_kernel, dim3(gridp), dim3(256), 0, (hipStream_t)stream, x, n, work);
    const int64_t ld_words = ld_bytes / 4;
    const int grid = qt_stream_grid((rows * (ld_words / 4) + 255) / 256 + 1);
    hipLaunchKernelGGL(pair_fold_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, K, work, gridp, mul_dev, scale3,
                       reinterpret_cast<uint32_t*>(out), ld_words, rows, K);
    return qt_check_launch();
}

extern "C" int64_t qt_f16x2_absmax_ch_work_words(int64_t C) { return 512 * (C > 0 ? C : 1); }

extern "C" int qt_f16x2_absmax_scale_ch_f32(const float* g, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int64_t N, int64_t C,
                                            int64_t H, int64_t W, int64_t Cp, uint32_t* work, float* scale2c, qt_stream_t stream) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C || !g || !work || !scale2c) return QT_ERR_INVALID_ARG;
    if (C > (1 << 20) || H * W >= (1ll << 31) || N >= (1ll << 31)) return QT_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    int nparts;
    const int64_t P = N * H * W;
    if (sc == 1 && sw == C && sh == W * C && sn == H * W * C) {
        const bool v4 = !(C & 3) && qt_aligned16(g);
        const int cq = (int)(v4 ? C / 4 : C);
        const int cw = cq < 256 ? cq : 256, rpb = 256 / cw;
        const int gy = (cq + cw - 1) / cw;
        int64_t gx = (P + rpb - 1) / rpb;
        const int64_t cap = gy >= 512 ? 1 : 512 / gy;
        if (gx > cap) gx = cap;
        nparts = (int)gx;
        if (v4) hipLaunchKernelGGL((absmax_ch_rows_kernel<4>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, st, g, P, (int)C, cw, rpb, work);
        else hipLaunchKernelGGL((absmax_ch_rows_kernel<1>), dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, st, g, P, (int)C, cw, rpb, work);
    } else {
        nparts = (int)(N < 16 ? N : 16);
        hipLaunchKernelGGL(absmax_ch_strided_kernel, dim3((unsigned)C, (unsigned)nparts), dim3(256), 0, st, g, sn, sc, sh, sw, (int)N,
                           (int)C, (int)H, (int)W, work);
    }
    hipLaunchKernelGGL(absmax_ch_final_kernel, dim3((unsigned)((Cp + 63) / 64)), dim3(1024), 0, st, work, nparts, (int)C, (int)Cp, scale2c);
    return qt_check_launch();
}

extern "C" int qt_f16x2_pack_f32(const float* x, int64_t ldx, const float* scale2, uint16_t* out, int64_t ld_bytes,
                                 int64_t rows, int64_t K, int mode, qt_stream_t stream) {
    if (rows < 0 || K < 0 || ldx < K || mode < 0 || mode > 4) return QT_ERR_INVALID_ARG;
    if (rows == 0) return QT_OK;
    if (!out || (!x && K > 0)) return QT_ERR_INVALID_ARG;
    if (ld_bytes < 4 * K || (ld_bytes & 15) || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    if (ld_bytes == 0) return QT_OK;
    const int64_t ld_words = ld_bytes / 4;
    const int grid = qt_stream_grid((rows * (ld_words / 4) + 255) / 256);
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
#define QT_LAUNCH(M) hipLaunchKernelGGL((pair_kernel<M>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, scale2, o, ld_words, rows, K)
    switch (mode) {
        case 0: QT_LAUNCH(0); break;
        case 1: QT_LAUNCH(1); break;
        case 2: QT_LAUNCH(2); break;
        case 3: QT_LAUNCH(3); break;
        default: QT_LAUNCH(4); break;
    }
#undef QT_LAUNCH
    return qt_check_launch();
}

extern "C" int qt_f16x2_s2d_pack_f32(const float* x, int64_t sN, int64_t sC, int64_t sH, int64_t sW, const float* scale2,
                                     uint16_t* out, int64_t ld_bytes, int64_t N, int64_t C, int64_t H, int64_t W, int64_t s,
                                     int64_t ph, int64_t pw, qt_stream_t stream) {
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || s < 1 || ph < 0 || pw < 0) return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!x || !out) return QT_ERR_INVALID_ARG;
    const int64_t E = C * s * s;
    if (ld_bytes < 4 * E || (ld_bytes & 15) || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    if (H > 32767 || W > 32767 || E > 4096) return QT_ERR_UNSUPPORTED;
    const int64_t Hs = (H + 2 * ph + s - 1) / s, Ws = (W + 2 * pw + s - 1) / s;
    const int64_t ld_words = ld_bytes / 4;
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    const int64_t rowf4 = (W * C + 3) & ~3ll;
    const int64_t lds = s * rowf4 * 4 + E * 8;
    if (sC == 1 && sW == C && lds <= 60 * 1024 && N * Hs < (1ll << 31) && sH >= W * C) {
        const int vec_ok = ((W * C) % 4 == 0) && qt_aligned16(x) && (sH % 4 == 0) && (sN % 4 == 0);
        hipLaunchKernelGGL(s2d_pair_rows_kernel, dim3((unsigned)(N * Hs)), dim3(256), (size_t)lds, (hipStream_t)stream, x, sN, sH,
                           scale2, o, ld_words, (int)C, (int)H, (int)W, (int)s, (int)ph, (int)pw, (int)Hs, (int)Ws, vec_ok, 0.0f,
                           (unsigned*)nullptr, (const int*)nullptr, (int)(N * Hs));
        return qt_check_launch();
    }
    const int64_t total = N * Hs * Ws * (ld_words / 4);
    const int grid = qt_stream_grid((total + 255) / 256);
    hipLaunchKernelGGL(s2d_pair_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, sN, sC, sH, sW, scale2, o, ld_words, N,
                       (int)C, (int)H, (int)W, (int)s, (int)ph, (int)pw, (int)Hs, (int)Ws);
    return qt_check_launch();
}

extern "C" int64_t qt_f16x2_s2d_spec_work_words(int64_t N, int64_t H, int64_t s, int64_t ph) {
    return N > 0 && s > 0 ? N * ((H + 2 * ph + s - 1) / s) : 0;
}

extern "C" int qt_f16x2_s2d_pack_spec_f32(const float* x, int64_t sN, int64_t sC, int64_t sH, int64_t sW, float spec_scale,
                                          uint32_t* work, float* scale2, int* redo, uint16_t* out, int64_t ld_bytes, int64_t N,
                                          int64_t C, int64_t H, int64_t W, int64_t s, int64_t ph, int64_t pw, qt_stream_t stream) {
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || s < 1 || ph < 0 || pw < 0) return QT_ERR_INVALID_ARG;
    if (!x || !out || !work || !scale2 || !redo) return QT_ERR_INVALID_ARG;
    int e2;
    if (!(spec_scale > 0.0f) || frexpf(spec_scale, &e2) != 0.5f) return QT_ERR_INVALID_ARG;     // a power of two
    if (N == 0) return QT_OK;
    const int64_t E = C * s * s;
    if (ld_bytes < 4 * E || (ld_bytes & 15) || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    if (H > 32767 || W > 32767 || E > 4096) return QT_ERR_UNSUPPORTED;
    const int64_t Hs = (H + 2 * ph + s - 1) / s, Ws = (W + 2 * pw + s - 1) / s;
    const int64_t ld_words = ld_bytes / 4;
    const int64_t rowf4 = (W * C + 3) & ~3ll;
    const int64_t lds = s * rowf4 * 4 + E * 8;
    // the row-staging kernel only (channels-last images): it sees every input row exactly once
    if (!(sC == 1 && sW == C && lds <= 60 * 1024 && N * Hs < (1ll << 31) && sH >= W * C)) return QT_ERR_UNSUPPORTED;
    const int vec_ok = ((W * C) % 4 == 0) && qt_aligned16(x) && (sH % 4 == 0) && (sN % 4 == 0);
    uint32_t* o = reinterpret_cast<uint32_t*>(out);
    hipStream_t st = (hipStream_t)stream;
    const float spec_inv = 1.0f / spec_scale;
    hipLaunchKernelGGL(s2d_pair_rows_kernel, dim3((unsigned)(N * Hs)), dim3(256), (size_t)lds, st, x, sN, sH, (const float*)nullptr, o,
                       ld_words, (int)C, (int)H, (int)W, (int)s, (int)ph, (int)pw, (int)Hs, (int)Ws, vec_ok, spec_inv, work,
                       (const int*)nullptr, (int)(N * Hs));
    hipLaunchKernelGGL(spec_final_kernel, dim3(1), dim3(1024), 0, st, work, (int)(N * Hs), spec_inv, scale2, redo);
    const int64_t rgrid = N * Hs < 1024 ? N * Hs : 1024;
    hipLaunchKernelGGL(s2d_pair_rows_kernel, dim3((unsigned)rgrid), dim3(256), (size_t)lds, st, x, sN, sH, (const float*)scale2, o,
                       ld_words, (int)C, (int)H, (int)W, (int)s, (int)ph, (int)pw, (int)Hs, (int)Ws, vec_ok, 0.0f,
                       (unsigned*)nullptr, (const int*)redo, (int)(N * Hs));
    return qt_check_launch();
}

extern "C" int qt_f16x2_pack_conv_weight_f32(const float* w, int64_t stride_o, int64_t stride_i, int64_t stride_h, int64_t stride_w,
                                             int64_t Cout, int64_t Cin, int64_t kh, int64_t kw, int mode, int transpose_flip,
                                             uint16_t* out, int64_t ld_bytes, qt_stream_t stream) {
    if (Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || mode < 1 || mode > 4 || !w || !out) return QT_ERR_INVALID_ARG;
    const int64_t rows = transpose_flip ? Cin : Cout, chans = transpose_flip ? Cout : Cin;
    const int64_t cb = (4 * chans + 15) / 16 * 16;             // bytes per tap (fp16 pairs, 16-byte granule)
    if (ld_bytes < kh * kw * cb || (ld_bytes & 127) || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    if (Cout * Cin * kh * kw >= (1ll << 31) || rows * ld_bytes >= (1ll << 33)) return QT_ERR_UNSUPPORTED;
    const int64_t ld_words = ld_bytes / 4;
    if (transpose_flip && kh == 1 && kw == 1 && stride_i == 1 && Cout * Cin >= (1 << 16)) {
        // (1 x 1, transposed, unit stride along Cin: the LDS-tiled transpose; the row padding, if any, by a second small launch)
        const int64_t tiles = ((Cout + 63) / 64) * ((Cin + 63) / 64);
        hipLaunchKernelGGL(weight_pair_transpose_kernel, dim3(qt_stream_grid(tiles, 256 * 8)), dim3(256), 0, (hipStream_t)stream, w, stride_o,
                           (int)Cout, (int)Cin, mode, reinterpret_cast<uint32_t*>(out), ld_words);
        if (ld_words > Cout)
            hipLaunchKernelGGL(pair_row_pad_kernel, dim3(qt_stream_grid((rows * (ld_words - Cout) + 255) / 256)), dim3(256), 0,
                               (hipStream_t)stream, reinterpret_cast<uint32_t*>(out), ld_words, (int)Cout, rows);
        return qt_check_launch();
    }
    hipLaunchKernelGGL(conv_weight_pair_kernel, dim3(qt_stream_grid((rows * ld_words + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       stride_o, stride_i, stride_h, stride_w, (int)Cout, (int)Cin, (int)kh, (int)kw, mode, transpose_flip,
                       reinterpret_cast<uint32_t*>(out), ld_words, (int)(cb / 4));
    return qt_check_launch();
}
