// int8 code planes for the DoReFa k-bit path (include/qt_hip.h).  HBM-bound elementwise kernels:
// fp32 in (4 B/element), int8 out (1 B/element) (+ optional fp32 image, 4 B/element).
//   activations: q = rint((2^k-1) * x)   (functions/dorefa_connect.py:24-25; round-half-even, NO clamp:
//                an |q| > 127 is reported through *overflow and the caller leaves the packed path)
//   weights    : safeSign(w) -> +-1  (k = 1 weights are sign(W) * E with the scalar E applied in the
//                GEMM epilogue, functions/dorefa_connect.py:99-102) or ternary codes.
#include "qt_common.h"

namespace {

// One thread = 4 consecutive elements of the padded row (ldc bytes, multiple of 16): slots past K
// write zero bytes so the pad-is-zero invariant of the plane holds.
template <bool WEIGHT>
__global__ __launch_bounds__(256) void codes_kernel(const float* __restrict__ x, int64_t ldx,
                                                    int8_t* __restrict__ codes, int64_t ldc,
                                                    float* __restrict__ yf, int64_t ldy, int64_t rows,
                                                    int64_t K, float n, float inv_n, int ternary,
                                                    int32_t* __restrict__ overflow, int vec) {
    const int64_t slots_per_row = ldc / 4;
    const int64_t total = rows * slots_per_row;
    int bad = 0;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
         s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = s / slots_per_row, slot = s - row * slots_per_row;
        const int64_t k0 = slot * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec && k0 + 3 < K) {
            const float4 t = *reinterpret_cast<const float4*>(x + row * ldx + k0);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k0 + e < K) v[e] = x[row * ldx + k0 + e];
        }
        uint32_t word = 0;
        float q4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int q = 0;
            if (k0 + e < K) {
                if (WEIGHT) {
                    q = (int)(ternary ? qt_ternarize(v[e]) : qt_safe_sign(v[e]));
                } else {
                    const float r = rintf(n * v[e]);
                    q4[e] = r;
                    // NaN / inf / out-of-range codes cannot be represented: flag them
                    // (bit 1: beyond +-2047 as well — no longer exact in an fp16 plane either, see the training weight gradient)
                    if (!(r >= -127.0f && r <= 127.0f)) { bad |= (r >= -2047.0f && r <= 2047.0f) ? 1 : 3; q = 0; } else q = (int)r;
                }
            }
            word |= (uint32_t)(uint8_t)(int8_t)q << (8 * e);
        }
        *reinterpret_cast<uint32_t*>(codes + row * ldc + k0) = word;
        if (!WEIGHT && yf) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k0 + e < K) yf[row * ldy + k0 + e] = inv_n * q4[e];   // fl(fl(1/n) * r), as _quantize
        }
    }
    if (!WEIGHT && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(overflow, __any(bad & 2) ? 3 : 1);
}

// Eval-mode BatchNorm of an fp32 [rows][C] matrix in the DEVICE's arithmetic, y = fma(fl(fl(x - mean) * rs), weight, bias) with
// bn_stats = [mean | rs]: the shortcut branch conv -> BatchNorm of a DoReFa ResNet block, whose result joins the main conv's code
// epilogue as a plain fp32 residual (the library's own inference kernel needs 38 us for the 33 MB tensor this pass moves in ~11).
__global__ __launch_bounds__(256) void bn_eval_device_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ w,
                                                             const float* __restrict__ b, const float* __restrict__ st,
                                                             float* __restrict__ y, int64_t ldy, int64_t rows, int64_t C) {
    const int64_t quads = C / 4, total = rows * quads;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / quads, c = (t - row * quads) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + c);
        const float4 m = *reinterpret_cast<const float4*>(st + c), r = *reinterpret_cast<const float4*>(st + C + c);
        const float4 ww = *reinterpret_cast<const float4*>(w + c), bb = *reinterpret_cast<const float4*>(b + c);
        float4 o;
        o.x = __fmaf_rn(__fmul_rn(__fsub_rn(v.x, m.x), r.x), ww.x, bb.x);
        o.y = __fmaf_rn(__fmul_rn(__fsub_rn(v.y, m.y), r.y), ww.y, bb.y);
        o.z = __fmaf_rn(__fmul_rn(__fsub_rn(v.z, m.z), r.z), ww.z, bb.z);
        o.w = __fmaf_rn(__fmul_rn(__fsub_rn(v.w, m.w), r.w), ww.w, bb.w);
        *reinterpret_cast<float4*>(y + row * ldy + c) = o;
    }
}

// Inference fusion of the DoReFa activation chain (k-bit form):
//   conv / linear output x -> eval BatchNorm folded to t = fl(fl(x*alpha[c]) + beta[c])
//     [+ residual: fl(fl(r*ralpha[c]) + rbeta[c]) of an fp32 tensor (a shortcut conv before ITS BatchNorm), or
//        fl(rscale * code) of an int8 code plane (identity shortcut that carries DoReFa codes)]
//     [-> ReLU] -> nnDorefaQuant(k): q = rint(n * t)   (functions/dorefa_connect.py:24-25, unclamped)
// written as the next layer's int8 code plane (and, on request, the fp32 image fl(fl(1/n) * q)).
// One thread = 4 consecutive channels; per element 4 B in (+ 4 B or 1 B residual), 1 B out: HBM-bound.
// Per-channel constants of one thread's 4 channels (loaded once per thread on the fixed-slot path).
struct AffineConsts {
    float al[4], be[4], mean[4], rs[4], ral[4], rbe[4], rmean[4], rrs[4];
};

__device__ __forceinline__ void affine_load_consts(AffineConsts& c, int64_t k0, int64_t C, const float* __restrict__ alpha,
                                                   const float* __restrict__ beta, const float* __restrict__ ralpha,
                                                   const float* __restrict__ rbeta, const float* __restrict__ bn_stats,
                                                   const float* __restrict__ rbn_stats) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bool in = k0 + e < C;
        c.al[e] = in ? alpha[k0 + e] : 0.0f;
        c.be[e] = in ? beta[k0 + e] : 0.0f;
        c.mean[e] = (in && bn_stats) ? bn_stats[k0 + e] : 0.0f;
        c.rs[e] = (in && bn_stats) ? bn_stats[C + k0 + e] : 1.0f;
        c.ral[e] = (in && ralpha) ? ralpha[k0 + e] : 1.0f;
        c.rbe[e] = (in && ralpha) ? rbeta[k0 + e] : 0.0f;
        c.rmean[e] = (in && rbn_stats) ? rbn_stats[k0 + e] : 0.0f;
        c.rrs[e] = (in && rbn_stats) ? rbn_stats[C + k0 + e] : 1.0f;
    }
}

// one output dword (4 channels of one row): the arithmetic of the header comment, every rounding spelled out
__device__ __forceinline__ uint32_t affine_codes_word(const float (&v)[4], const float (&r)[4], uint32_t rword, const AffineConsts& c,
                                                      int nvalid, bool devbn, bool has_rf, bool has_ralpha, bool rdevbn, bool has_rc,
                                                      float rscale, int relu, float n, float (&q4)[4], int& bad) {
    uint32_t word = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int q = 0;
        q4[e] = 0.0f;
        if (e < nvalid) {
            const float x0 = (relu == 2 && v[e] < 0.0f) ? 0.0f : v[e];          // ReLU in front of the BatchNorm
            // folded form: fl(fl(x * alpha) + beta).  Device form (bn_stats = [mean | rs], alpha = weight, beta = bias):
            // fma(fl(fl(x - mean) * rs), weight, bias) — the expression this device's eval-mode F.batch_norm evaluates
            float t = devbn ? __fmaf_rn(__fmul_rn(__fsub_rn(x0, c.mean[e]), c.rs[e]), c.al[e], c.be[e])
                            : __fadd_rn(__fmul_rn(x0, c.al[e]), c.be[e]);
            if (has_rf) {
                float u = r[e];
                if (has_ralpha)
                    u = rdevbn ? __fmaf_rn(__fmul_rn(__fsub_rn(u, c.rmean[e]), c.rrs[e]), c.ral[e], c.rbe[e])
                               : __fadd_rn(__fmul_rn(u, c.ral[e]), c.rbe[e]);
                t = __fadd_rn(t, u);
            }
            if (has_rc) t = __fadd_rn(t, __fmul_rn(rscale, (float)(int8_t)(rword >> (8 * e))));
            if (relu == 1) t = t < 0.0f ? 0.0f : t;             // NaN stays NaN (flagged below)
            const float q_ = rintf(__fmul_rn(n, t));
            q4[e] = q_;
            if (!(q_ >= -127.0f && q_ <= 127.0f)) { bad |= (q_ >= -2047.0f && q_ <= 2047.0f) ? 1 : 3; q = 0; } else q = (int)q_;
        }
        word |= (uint32_t)(uint8_t)(int8_t)q << (8 * e);
    }
    return word;
}

// FIXED: the launch's thread count is a multiple of the slots per row, so a thread keeps its 4 channels for its whole
// grid-stride walk: the per-channel constants are loaded once, the row index advances by a constant (no 64-bit division per
// element — 180 VALU instructions per dword and 1.2 TB/s before, profiles/r5_c4_pmc.md), and two rows are in flight per
// iteration.  Otherwise: the general walk (slot recomputed per element).
// Output rows of a halo plane: row m = (img, h, w) of the [N][H][W] pixel matrix lands on pixel (img, h + hy, w + hx) of a
// [N][H + 2hy][W + 2hx] plane whose border the launch's surplus blocks write as zeros (the consuming conv's padding, physical).
struct HaloMap {
    int hw = 0, W = 0, H = 0, hy = 0, hx = 0, sh_hw = -1, sh_w = -1, main_blocks = 0, border_blocks = 0;
    int64_t N = 0;
};
__device__ __forceinline__ int64_t halo_row(const HaloMap& hm, int64_t row) {
    if (!(hm.hy | hm.hx)) return row;
    const unsigned m = (unsigned)row;
    unsigned img, h, w;
    if (hm.sh_w >= 0) {
        img = m >> hm.sh_hw;
        const unsigned rem = m & (unsigned)(hm.hw - 1);
        h = rem >> hm.sh_w;
        w = rem & (unsigned)(hm.W - 1);
    } else {
        img = m / (unsigned)hm.hw;
        const unsigned rem = m - img * (unsigned)hm.hw;
        h = rem / (unsigned)hm.W;
        w = rem - h * (unsigned)hm.W;
    }
    return ((int64_t)img * (hm.H + 2 * hm.hy) + h + hm.hy) * (hm.W + 2 * hm.hx) + w + hm.hx;
}
// border pixels of the plane, 16-byte chunks, by the FIRST hm.border_blocks blocks of the launch (they start with the main blocks, not
// as its tail).  Only border pixels are enumerated, in 32-bit arithmetic: per image hy rows on top, hy rows below, 2 hx columns beside
// each of the H image rows.
__device__ __forceinline__ void halo_zero_border(const HaloMap& hm, int8_t* __restrict__ codes, int64_t ldc) {
    const unsigned Hp = hm.H + 2 * hm.hy, Wp = hm.W + 2 * hm.hx, cpp = (unsigned)(ldc / 16);
    const unsigned top = hm.hy * Wp, side = 2u * hm.hx * hm.H, per_img = 2u * top + side;
    const unsigned total = (unsigned)hm.N * per_img * cpp;                       // < 2^32: host check
    const unsigned nthr = (unsigned)hm.border_blocks * blockDim.x;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += nthr) {
        const unsigned b = t / cpp, c = t - b * cpp;
        const unsigned img = b / per_img, r = b - img * per_img;
        unsigned h, w;
        if (r < top) { h = r / Wp; w = r - h * Wp; }
        else if (r < top + side) {
            const unsigned q = r - top, row = q / (2u * hm.hx), k = q - row * 2u * hm.hx;
            h = hm.hy + row;
            w = k < (unsigned)hm.hx ? k : hm.W + k;
        } else { const unsigned q = r - top - side; h = hm.hy + hm.H + q / Wp; w = q % Wp; }
        *reinterpret_cast<uint4*>(codes + ((int64_t)(img * Hp + h) * Wp + w) * ldc + c * 16) = make_uint4(0, 0, 0, 0);
    }
}

template <bool FIXED>
__global__ __launch_bounds__(256) void affine_codes_kernel(
    const float* __restrict__ x, int64_t ldx, const float* __restrict__ alpha, const float* __restrict__ beta,
    const float* __restrict__ rf, int64_t ldr, const float* __restrict__ ralpha, const float* __restrict__ rbeta,
    const int8_t* __restrict__ rc, int64_t ldrc, float rscale, int relu, int8_t* __restrict__ codes, int64_t ldc,
    float* __restrict__ yf, int64_t ldy, int64_t rows, int64_t C, float n, float inv_n,
    int32_t* __restrict__ overflow, int vec, const float* __restrict__ bn_stats, const float* __restrict__ rbn_stats, HaloMap hm) {
    const int64_t slots_per_row = ldc / 4;
    int bad = 0;
    if ((int)blockIdx.x < hm.border_blocks) {        // (uniform for the block)
        halo_zero_border(hm, codes, ldc);
        return;
    }
    const unsigned block = blockIdx.x - hm.border_blocks;
    const bool devbn = bn_stats != nullptr, has_rf = rf != nullptr, has_ralpha = ralpha != nullptr, rdevbn = rbn_stats != nullptr,
               has_rc = rc != nullptr;
    auto load4 = [&](int64_t row, int64_t k0, bool full, float (&v)[4], float (&r)[4], uint32_t& rword) {
        v[0] = v[1] = v[2] = v[3] = 0.0f;
        r[0] = r[1] = r[2] = r[3] = 0.0f;
        if (vec && full) {
            const float4 t = *reinterpret_cast<const float4*>(x + row * ldx + k0);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            if (has_rf) {
                const float4 u = *reinterpret_cast<const float4*>(rf + row * ldr + k0);
                r[0] = u.x; r[1] = u.y; r[2] = u.z; r[3] = u.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k0 + e < C) {
                    v[e] = x[row * ldx + k0 + e];
                    if (has_rf) r[e] = rf[row * ldr + k0 + e];
                }
        }
        rword = has_rc ? *reinterpret_cast<const uint32_t*>(rc + row * ldrc + k0) : 0u;   // plane rows are 16-byte padded
    };
    auto store4 = [&](int64_t row, int64_t k0, uint32_t word, const float (&q4)[4]) {
        *reinterpret_cast<uint32_t*>(codes + halo_row(hm, row) * ldc + k0) = word;
        if (yf) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (k0 + e < C) yf[row * ldy + k0 + e] = inv_n * q4[e];
        }
    };
    if constexpr (FIXED) {
        const unsigned spr = (unsigned)slots_per_row, nthreads = (unsigned)hm.main_blocks * blockDim.x;
        const unsigned s0 = block * blockDim.x + threadIdx.x;
        const unsigned slot = s0 % spr;
        const int64_t drow = nthreads / spr, k0 = (int64_t)slot * 4;
        const bool full = k0 + 3 < C;
        const int nvalid = (int)(C - k0 < 0 ? 0 : (C - k0 > 4 ? 4 : C - k0));
        AffineConsts c;
        affine_load_consts(c, k0, C, alpha, beta, ralpha, rbeta, bn_stats, rbn_stats);
        int64_t row = s0 / spr;
        for (; row + drow < rows; row += 2 * drow) {          // two rows in flight
            float v0[4], r0[4], v1[4], r1[4], q0[4], q1[4];
            uint32_t w0, w1;
            load4(row, k0, full, v0, r0, w0);
            load4(row + drow, k0, full, v1, r1, w1);
            const uint32_t o0 = affine_codes_word(v0, r0, w0, c, nvalid, devbn, has_rf, has_ralpha, rdevbn, has_rc, rscale, relu, n, q0, bad);
            const uint32_t o1 = affine_codes_word(v1, r1, w1, c, nvalid, devbn, has_rf, has_ralpha, rdevbn, has_rc, rscale, relu, n, q1, bad);
            store4(row, k0, o0, q0);
            store4(row + drow, k0, o1, q1);
        }
        if (row < rows) {
            float v0[4], r0[4], q0[4];
            uint32_t w0;
            load4(row, k0, full, v0, r0, w0);
            store4(row, k0, affine_codes_word(v0, r0, w0, c, nvalid, devbn, has_rf, has_ralpha, rdevbn, has_rc, rscale, relu, n, q0, bad), q0);
        }
    } else {
        const int64_t total = rows * slots_per_row;
        for (int64_t s = (int64_t)block * blockDim.x + threadIdx.x; s < total; s += (int64_t)hm.main_blocks * blockDim.x) {
            const int64_t row = s / slots_per_row, slot = s - row * slots_per_row;
            const int64_t k0 = slot * 4;
            const int nvalid = (int)(C - k0 < 0 ? 0 : (C - k0 > 4 ? 4 : C - k0));
            AffineConsts c;
            affine_load_consts(c, k0, C, alpha, beta, ralpha, rbeta, bn_stats, rbn_stats);
            float v0[4], r0[4], q0[4];
            uint32_t w0;
            load4(row, k0, k0 + 3 < C, v0, r0, w0);
            store4(row, k0, affine_codes_word(v0, r0, w0, c, nvalid, devbn, has_rf, has_ralpha, rdevbn, has_rc, rscale, relu, n, q0, bad), q0);
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(overflow, __any(bad & 2) ? 3 : 1);
}


// Code plane -> the fp32 image nnDorefaQuant would have returned, fl(inv_n * q), as an NHWC matrix [N * Ho * Wo][ldy]; a raised int8
// range flag of the chain turns every value into NaN (there is no fp32 image to fall back to: packed.CodeActivation.float).
// pool_k > 1: followed by avg_pool2d(pool_k) (kernel = stride, no padding, floor mode): the window's values are added in (row, column)
// order in fp32 and divided by the window size — the order of ATen's avg_pool2d kernels, so the result is theirs bit for bit.
// in [N][H + 2hy][W + 2hx][ld bytes]; one thread = 4 channels of one output pixel.  (Was: five ATen kernels + the pooling.)
__global__ __launch_bounds__(256) void codes_to_f32_kernel(const int8_t* __restrict__ codes, int64_t ld, int N, int H, int W, int hy,
                                                           int hx, int C, float inv_n, const int32_t* __restrict__ overflow, int pk,
                                                           float* __restrict__ y, int64_t ldy, int vec) {
    const int Ho = H / pk, Wo = W / pk, quads = (C + 3) / 4;
    const int64_t total = (int64_t)N * Ho * Wo * quads;
    const bool nan_all = overflow != nullptr && *overflow != 0;
    const float div = (float)(pk * pk);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = t / quads;
        const int c0 = (int)(t - pix * quads) * 4;
        const int wo = (int)(pix % Wo);
        const int64_t r = pix / Wo;
        const int ho = (int)(r % Ho), n = (int)(r / Ho);
        float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = 0; i < pk; ++i)
            for (int j = 0; j < pk; ++j) {
                const int64_t src = ((int64_t)n * (H + 2 * hy) + ho * pk + i + hy) * (W + 2 * hx) + wo * pk + j + hx;
                const uint32_t word = *reinterpret_cast<const uint32_t*>(codes + src * ld + c0);   // rows are padded to 4 bytes
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = __fmul_rn((float)(int8_t)(word >> (8 * e)), inv_n);
                    a[e] = pk == 1 ? v : __fadd_rn(a[e], v);
                }
            }
        if (pk > 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = __fdiv_rn(a[e], div);
        }
        if (nan_all) a[0] = a[1] = a[2] = a[3] = __builtin_nanf("");
        float* dst = y + pix * ldy + c0;
        if (vec && c0 + 3 < C) {
            *reinterpret_cast<float4*>(dst) = make_float4(a[0], a[1], a[2], a[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + e < C) dst[e] = a[e];
        }
    }
}

// The conv weight [Cout][Cin][kh][kw] as the int8 operand of the code-plane convs in one pass: row co, tap (i, j), channel ci
// hold safeSign / ternary of w[co][ci][i][j]; taps are cb bytes apart (Cin rounded to 16), the row is zero-padded to ldc bytes.
// One thread = one 4-byte word of the output row.  (Was: permute copy, pack, zero fill, padded copy.)
__global__ __launch_bounds__(256) void conv_weight_codes_kernel(const float* __restrict__ w, int64_t so, int64_t si, int64_t sh, int64_t sw,
                                                                int Cout, int Cin, int taps, int kw, int ternary,
                                                                int8_t* __restrict__ codes, int64_t ldc, int cb) {
    const int64_t words_per_row = ldc / 4, total = (int64_t)Cout * words_per_row;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int co = (int)(t / words_per_row);
        const int byte0 = (int)(t - (int64_t)co * words_per_row) * 4;
        const int tap = byte0 / cb, c0 = byte0 - tap * cb;
        uint32_t word = 0;
        if (tap < taps) {
            const int i = tap / kw, j = tap - i * kw;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = c0 + e;
                if (ci >= Cin) break;
                const float v = w[co * so + ci * si + i * sh + j * sw];
                const int q = (int)(ternary ? qt_ternarize(v) : qt_safe_sign(v));
                word |= (uint32_t)(uint8_t)(int8_t)q << (8 * e);
            }
        }
        *reinterpret_cast<uint32_t*>(codes + (int64_t)co * ldc + byte0) = word;
    }
}

}  // namespace

extern "C" {

int qt_bn_eval_device_f32(const float* x, int64_t ldx, const float* weight, const float* bias, const float* bn_stats, float* y,
                          int64_t ldy, int64_t rows, int64_t C, qt_stream_t stream) {
    if (rows < 0 || C < 0 || ldx < C || ldy < C) return QT_ERR_INVALID_ARG;
    if (rows == 0 || C == 0) return QT_OK;
    if (!x || !y || !weight || !bias || !bn_stats) return QT_ERR_INVALID_ARG;
    if ((C & 3) || (ldx & 3) || (ldy & 3) || !qt_aligned16(x) || !qt_aligned16(y) || !qt_aligned16(weight) || !qt_aligned16(bias) ||
        !qt_aligned16(bn_stats))
        return QT_ERR_ALIGNMENT;
    const int grid = qt_stream_grid((rows * (C / 4) + 255) / 256);
    hipLaunchKernelGGL(bn_eval_device_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, weight, bias, bn_stats, y, ldy,
                       rows, C);
    return qt_check_launch();
}

static int affine_codes_impl(const float* x, int64_t ldx, const float* alpha, const float* beta,
                             const float* res_f32, int64_t ldr, const float* res_alpha, const float* res_beta,
                             const int8_t* res_codes, int64_t ldrc_bytes, float res_scale, int relu, int8_t* codes,
                             int64_t ldc_bytes, float* y_f32, int64_t ldy, int64_t rows, int64_t C, int bit_width,
                             int32_t* overflow, const float* bn_stats, const float* res_bn_stats, int64_t N, int64_t H, int64_t W,
                             int64_t hy, int64_t hx, qt_stream_t stream) {
    if (rows < 0 || C < 0 || ldx < C || bit_width < 2 || bit_width > 8 || relu < 0 || relu > 2) return QT_ERR_INVALID_ARG;
    if (rows == 0) return QT_OK;
    if (!codes || !overflow || !alpha || !beta || (!x && C > 0) || (y_f32 && ldy < C)) return QT_ERR_INVALID_ARG;
    if ((res_f32 && ldr < C) || (!res_alpha != !res_beta) || (res_alpha && !res_f32)) return QT_ERR_INVALID_ARG;
    if (res_bn_stats && !res_alpha) return QT_ERR_INVALID_ARG;
    if (ldc_bytes < C || (ldc_bytes & 15) || !qt_aligned16(codes)) return QT_ERR_ALIGNMENT;
    if (res_codes && (ldrc_bytes < ((C + 3) & ~(int64_t)3) || (ldrc_bytes & 3) || ((uintptr_t)res_codes & 3)))
        return QT_ERR_ALIGNMENT;
    if (ldc_bytes == 0) return QT_OK;
    const float n = (float)((1 << bit_width) - 1);
    const int vec = qt_aligned16(x) && (ldx % 4 == 0) && (!res_f32 || (qt_aligned16(res_f32) && ldr % 4 == 0));
    int grid = qt_stream_grid((rows * (ldc_bytes / 4) + 255) / 256);
    // fixed-slot walk: a thread count that is a multiple of the slots per row (grid rounded DOWN to a multiple of
    // spr / gcd(spr, 256) blocks — the grid-stride loop covers the rest); rows * spr < 2^32 for its 32-bit slot arithmetic
    const int64_t spr = ldc_bytes / 4;
    int64_t g = 256, t = spr;
    while (t) { const int64_t r_ = g % t; g = t; t = r_; }
    const int64_t unit = spr / g;                              // blocks per whole number of rows
    const bool fixed = unit <= grid && spr <= (1 << 20) && rows * spr < (1ll << 32);
    if (fixed) grid = (int)(grid / unit * unit);
    HaloMap hm;
    hm.main_blocks = grid;
    if (hy | hx) {
        hm.N = N; hm.H = (int)H; hm.W = (int)W; hm.hw = (int)(H * W); hm.hy = (int)hy; hm.hx = (int)hx;
        auto lg = [](int64_t v) { int s_ = 0; while ((1ll << s_) < v) ++s_; return (1ll << s_) == v ? s_ : -1; };
        hm.sh_hw = lg(H * W);
        hm.sh_w = hm.sh_hw >= 0 ? lg(W) : -1;
        hm.border_blocks = 32;                                 // border-zeroing blocks, in front
        grid += hm.border_blocks;
    }
#define QT_AFFINE(F) hipLaunchKernelGGL((affine_codes_kernel<F>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, alpha, beta, \
                       res_f32, ldr, res_alpha, res_beta, res_codes, ldrc_bytes, res_scale, relu, codes, ldc_bytes,                  \
                       y_f32, ldy, rows, C, n, 1.0f / n, overflow, vec, bn_stats, res_bn_stats, hm)
    if (fixed) QT_AFFINE(true); else QT_AFFINE(false);
#undef QT_AFFINE
    return qt_check_launch();
}

int qt_affine_dorefa_codes_i8(const float* x, int64_t ldx, const float* alpha, const float* beta,
                              const float* res_f32, int64_t ldr, const float* res_alpha, const float* res_beta,
                              const int8_t* res_codes, int64_t ldrc_bytes, float res_scale, int relu, int8_t* codes,
                              int64_t ldc_bytes, float* y_f32, int64_t ldy, int64_t rows, int64_t C, int bit_width,
                              int32_t* overflow, const float* bn_stats, const float* res_bn_stats, qt_stream_t stream) {
    return affine_codes_impl(x, ldx, alpha, beta, res_f32, ldr, res_alpha, res_beta, res_codes, ldrc_bytes, res_scale, relu, codes,
                             ldc_bytes, y_f32, ldy, rows, C, bit_width, overflow, bn_stats, res_bn_stats, 0, 0, 0, 0, 0, stream);
}

int qt_affine_dorefa_codes_halo_i8(const float* x, int64_t ldx, const float* alpha, const float* beta,
                                   const float* res_f32, int64_t ldr, const float* res_alpha, const float* res_beta,
                                   const int8_t* res_codes, int64_t ldrc_bytes, float res_scale, int relu, int8_t* codes,
                                   int64_t ldc_bytes, int64_t N, int64_t H, int64_t W, int64_t C, int bit_width,
                                   int32_t* overflow, const float* bn_stats, const float* res_bn_stats, int64_t out_halo_h,
                                   int64_t out_halo_w, qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || out_halo_h < 0 || out_halo_w < 0 || out_halo_h > 64 || out_halo_w > 64) return QT_ERR_INVALID_ARG;
    if (N * H * W >= (1ll << 31) || N * (H + 2 * out_halo_h) * (W + 2 * out_halo_w) * (ldc_bytes / 16 + 1) >= (1ll << 31)) return QT_ERR_UNSUPPORTED;
    return affine_codes_impl(x, ldx, alpha, beta, res_f32, ldr, res_alpha, res_beta, res_codes, ldrc_bytes, res_scale, relu, codes,
                             ldc_bytes, nullptr, 0, N * H * W, C, bit_width, overflow, bn_stats, res_bn_stats, N, H, W, out_halo_h,
                             out_halo_w, stream);
}

int qt_codes_to_f32(const int8_t* codes, int64_t ld_bytes, int64_t N, int64_t H, int64_t W, int64_t halo_h, int64_t halo_w,
                    int64_t C, float inv_n, const int32_t* overflow, int64_t pool_k, float* y, int64_t ldy, qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || C <= 0 || halo_h < 0 || halo_w < 0 || pool_k < 1 || pool_k > H || pool_k > W || ldy < C)
        return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!codes || !y) return QT_ERR_INVALID_ARG;
    if (ld_bytes < C || (ld_bytes & 3) || ((uintptr_t)codes & 3)) return QT_ERR_ALIGNMENT;
    const int64_t Ho = H / pool_k, Wo = W / pool_k;
    if (N * (H + 2 * halo_h) * (W + 2 * halo_w) >= (1ll << 31) || N * Ho * Wo * ((C + 3) / 4) >= (1ll << 31)) return QT_ERR_UNSUPPORTED;
    const int vec = qt_aligned16(y) && (ldy % 4 == 0);
    const int grid = qt_stream_grid((N * Ho * Wo * ((C + 3) / 4) + 255) / 256);
    hipLaunchKernelGGL(codes_to_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, codes, ld_bytes, (int)N, (int)H, (int)W,
                       (int)halo_h, (int)halo_w, (int)C, inv_n, overflow, (int)pool_k, y, ldy, vec);
    return qt_check_launch();
}

int qt_dorefa_codes_i8(const float* x, int64_t ldx, int8_t* codes, int64_t ldc_bytes, float* y_f32,
                       int64_t ldy, int64_t rows, int64_t K, int bit_width, int32_t* overflow,
                       qt_stream_t stream) {
    if (rows < 0 || K < 0 || ldx < K || bit_width < 2 || bit_width > 8) return QT_ERR_INVALID_ARG;
    if (rows == 0) return QT_OK;
    if (!codes || !overflow || (!x && K > 0) || (y_f32 && ldy < K)) return QT_ERR_INVALID_ARG;
    if (ldc_bytes < K || (ldc_bytes & 15) || !qt_aligned16(codes)) return QT_ERR_ALIGNMENT;
    if (ldc_bytes == 0) return QT_OK;
    const float n = (float)((1 << bit_width) - 1);
    const int vec = qt_aligned16(x) && (ldx % 4 == 0);
    const int grid = qt_stream_grid((rows * (ldc_bytes / 4) + 255) / 256);
    hipLaunchKernelGGL((codes_kernel<false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, codes,
                       ldc_bytes, y_f32, ldy, rows, K, n, 1.0f / n, 0, overflow, vec);
    return qt_check_launch();
}

int qt_weight_codes_i8(const float* w, int64_t ldw, int8_t* codes, int64_t ldc_bytes, int64_t rows,
                       int64_t K, int ternary, qt_stream_t stream) {
    if (rows < 0 || K < 0 || ldw < K) return QT_ERR_INVALID_ARG;
    if (rows == 0) return QT_OK;
    if (!codes || (!w && K > 0)) return QT_ERR_INVALID_ARG;
    if (ldc_bytes < K || (ldc_bytes & 15) || !qt_aligned16(codes)) return QT_ERR_ALIGNMENT;
    if (ldc_bytes == 0) return QT_OK;
    const int vec = qt_aligned16(w) && (ldw % 4 == 0);
    const int grid = qt_stream_grid((rows * (ldc_bytes / 4) + 255) / 256);
    hipLaunchKernelGGL((codes_kernel<true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, w, ldw, codes,
                       ldc_bytes, nullptr, 0, rows, K, 0.0f, 0.0f, ternary, nullptr, vec);
    return qt_check_launch();
}

int qt_pack_conv_weight_codes_i8(const float* w, int64_t stride_o, int64_t stride_i, int64_t stride_h, int64_t stride_w, int64_t Cout,
                                 int64_t Cin, int64_t kh, int64_t kw, int ternary, int8_t* codes, int64_t ldc_bytes, qt_stream_t stream) {
    if (Cout <= 0 || Cin <= 0 || kh <= 0 || kw <= 0 || !w || !codes) return QT_ERR_INVALID_ARG;
    const int64_t cb = (Cin + 15) / 16 * 16;
    if (ldc_bytes < kh * kw * cb || (ldc_bytes & 15) || !qt_aligned16(codes)) return QT_ERR_ALIGNMENT;
    if (Cout * Cin * kh * kw >= (1ll << 31)) return QT_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(conv_weight_codes_kernel, dim3(qt_stream_grid((Cout * (ldc_bytes / 4) + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, w, stride_o, stride_i, stride_h, stride_w, (int)Cout, (int)Cin, (int)(kh * kw), (int)kw, ternary, codes,
                       ldc_bytes, (int)cb);
    return qt_check_launch();
}

}  // extern "C"
