// Elementwise quantisers and STE masks (fp32 -> fp32).  All HBM-bound: 16-byte loads/stores per
// lane, grid-stride, scalar head/tail so any pointer alignment and any n is accepted.
#include "qt_common.h"

namespace {

struct OpBinarize {
    __device__ __forceinline__ float operator()(float x) const { return qt_safe_sign(x); }
};
struct OpTernarize {
    __device__ __forceinline__ float operator()(float x) const { return qt_ternarize(x); }
};
// _quantize, functions/dorefa_connect.py:24-25:  (1/(2^k-1)) * round((2^k-1) * x).
// torch.round is round-half-even -> rintf under the default rounding mode.  The reciprocal is
// computed in fp32 first (fl(1/n)), then multiplied: NOT r/n (differs in the last ulp).
struct OpDorefa {
    float n, inv_n;
    __device__ __forceinline__ float operator()(float x) const { return inv_n * rintf(n * x); }
};
struct OpCopy {
    __device__ __forceinline__ float operator()(float x) const { return x; }
};

// ---- Lin / Log fixed-point quantisers (functions/log_lin_connect.py) -----------------------------------------
// torch.sign = (0 < x) - (x < 0): sign(+-0) = +0, sign(NaN) = NaN
__device__ __forceinline__ float qt_torch_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : (x != x ? x : 0.0f)); }
__device__ __forceinline__ float qt_torch_clamp(float v, float lo, float hi) {   // torch.clamp propagates NaN
    return v != v ? v : (v < lo ? lo : (v > hi ? hi : v));   // compare chain: clamp(-0, 0, hi) stays -0, like ATen
}
struct OpLinQuant {   // log_lin_connect.py:61-67: mode 0: clamp(round(x/step)*step, 0, 2^fsr); mode 1: sign(x) * the
    float step, maxv;  // same of |x|; mode 2: sign(g) * clamp(round(g/step)*step, 0, 2^fsr) (the quantised-gradient
    int mode;          // backward, :79 — negative g clamps to 0, so it yields -0: reproduced)
    __device__ __forceinline__ float operator()(float x) const {
        const float a = mode == 1 ? fabsf(x) : x;
        const float q = qt_torch_clamp(rintf(a / step) * step, 0.0f, maxv);
        return mode == 0 ? q : qt_torch_sign(x) * q;
    }
};
struct OpLogQuant {   // log_lin_connect.py:31-33: [sign(x) *] 2^clamp(round(log2|x|), fsr - 2^bits, fsr)
    float lo, hi;
    int with_sign;
    __device__ __forceinline__ float operator()(float x) const {
        const float e = qt_torch_clamp(rintf(log2f(fabsf(x))), lo, hi);   // x = 0: -inf -> lo
        const float p = exp2f(e);                                        // integer e: exact (0 below 2^-149)
        return with_sign == 2 ? qt_safe_sign(x) * p : (with_sign ? qt_torch_sign(x) * p : p);   // 2: AP2's safeSign
    }
};

template <class Op>
__global__ __launch_bounds__(256) void unary_kernel(const float* __restrict__ x,
                                                    float* __restrict__ y, int64_t n, int64_t head,
                                                    Op op) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    // scalar head up to the first 16-byte boundary of x (y shares x's misalignment or we were
    // launched with head = n, i.e. fully scalar)
    for (int64_t i = tid; i < head; i += nthreads) y[i] = op(x[i]);
    const int64_t n4 = (n - head) / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x + head);
    float4* y4 = reinterpret_cast<float4*>(y + head);
    for (int64_t i = tid; i < n4; i += nthreads) {
        float4 v = x4[i];
        float4 r;
        r.x = op(v.x); r.y = op(v.y); r.z = op(v.z); r.w = op(v.w);
        y4[i] = r;
    }
    for (int64_t i = head + n4 * 4 + tid; i < n; i += nthreads) y[i] = op(x[i]);
}

struct Op2BinStoch {  // -1 + 2*[z < (clamp(x,-1,1)+1)/2]   binary_connect.py:57-61
    __device__ __forceinline__ float operator()(float x, float z) const {
        const float c = fminf(fmaxf(x, -1.0f), 1.0f);
        // NaN: torch.clamp propagates NaN -> p = NaN -> (z < p) false -> -1.  fmaxf/fminf drop the
        // NaN, so restore it explicitly.
        const float p = (x != x) ? x : (c + 1.0f) / 2.0f;
        return (z < p) ? 1.0f : -1.0f;
    }
};
struct Op2TerStoch {  // s - s*[z > |x|]   terner_connect.py:54-56
    __device__ __forceinline__ float operator()(float x, float z) const {
        const float s = qt_safe_sign(x);
        return s - s * ((z > fabsf(x)) ? 1.0f : 0.0f);
    }
};
struct Op2SteMask {  // g * [|x| <= thr]  (NaN input keeps g: |NaN| > thr is false)
    float thr;
    __device__ __forceinline__ float operator()(float g, float x) const {
        return (fabsf(x) > thr) ? 0.0f : g;
    }
};

template <class Op>
__global__ __launch_bounds__(256) void binary_kernel(const float* __restrict__ a,
                                                     const float* __restrict__ b,
                                                     float* __restrict__ y, int64_t n, int vec,
                                                     Op op) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = vec ? n / 4 : 0;
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (int64_t i = tid; i < n4; i += nthreads) {
        float4 u = a4[i], v = b4[i], r;
        r.x = op(u.x, v.x); r.y = op(u.y, v.y); r.z = op(u.z, v.z); r.w = op(u.w, v.w);
        y4[i] = r;
    }
    for (int64_t i = n4 * 4 + tid; i < n; i += nthreads) y[i] = op(a[i], b[i]);
}

template <class Op>
int launch_unary(const float* x, float* y, int64_t n, qt_stream_t stream, Op op) {
    if (n < 0 || (n > 0 && (!x || !y))) return QT_ERR_INVALID_ARG;
    if (n == 0) return QT_OK;
    // vector body needs x and y to share their offset from a 16-byte boundary
    const uintptr_t ax = reinterpret_cast<uintptr_t>(x) & 15u, ay = reinterpret_cast<uintptr_t>(y) & 15u;
    int64_t head = n;
    if (ax == ay && (ax & 3u) == 0) head = ax ? (int64_t)((16 - ax) / 4) : 0;
    if (head > n) head = n;
    const int grid = qt_stream_grid((n + 1023) / 1024);
    hipLaunchKernelGGL(unary_kernel<Op>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, n,
                       head, op);
    return qt_check_launch();
}

template <class Op>
int launch_binary(const float* a, const float* b, float* y, int64_t n, qt_stream_t stream, Op op) {
    if (n < 0 || (n > 0 && (!a || !b || !y))) return QT_ERR_INVALID_ARG;
    if (n == 0) return QT_OK;
    const int vec = qt_aligned16(a) && qt_aligned16(b) && qt_aligned16(y);
    const int grid = qt_stream_grid((n + 1023) / 1024);
    hipLaunchKernelGGL(binary_kernel<Op>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a, b, y,
                       n, vec, op);
    return qt_check_launch();
}

}  // namespace

namespace {
// out[i] = (x ? x[i] : 0) and NaN instead when (*flag & mask) != 0: how a device-side range flag (the un-clamped DoReFa
// quantiser left int8 / the fp16 plane; an un-tagged activation was not +-1 after all) reaches a result without a host sync
__global__ __launch_bounds__(256) void poison_kernel(const float* __restrict__ x, const int32_t* __restrict__ flag, int32_t mask,
                                                     float* __restrict__ out, int64_t n) {
    const bool bad = (*flag & mask) != 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = bad ? __builtin_nanf("") : (x ? x[i] : 0.0f);
}
// Base-256 digits of the integer codes of a k-bit DoReFa image x = q / n whose codes left int8 (functions/dorefa_connect.py:11-25 has
// no clamp): q = rint(x * levels), hi = floor(q / 256), lo = q - 256 hi, both exact in bf16 while |q| < 2^16 — beyond that *flag |= bit.
// One pass instead of six torch launches (+ five for the range test) in front of the two digit passes of the weight gradient.
__global__ __launch_bounds__(256) void code_digits_kernel(const float* __restrict__ x, int64_t n, float levels, float* __restrict__ hi,
                                                          float* __restrict__ lo, int32_t* __restrict__ flag, int32_t bit) {
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float q = rintf(x[i] * levels);
        const float h = floorf(q * (1.0f / 256.0f));
        hi[i] = h;
        lo[i] = q - h * 256.0f;
        bad |= !(fabsf(h) < 256.0f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, bit);
}

// out = fl(fl(fl(g_hi * 256) + g_lo) * inv), NaN when (*flag & mask) != 0: the two digit passes put together
__global__ __launch_bounds__(256) void digit_combine_kernel(const float* __restrict__ ghi, const float* __restrict__ glo,
                                                            const int32_t* __restrict__ flag, int32_t mask, float inv,
                                                            float* __restrict__ out, int64_t n) {
    const bool bad = flag && (*flag & mask) != 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = (ghi[i] * 256.0f + glo[i]) * inv;
        out[i] = bad ? __builtin_nanf("") : v;
    }
}
}  // namespace

extern "C" {

int qt_code_digits_f32(const float* x, int64_t n, float levels, float* hi, float* lo, int32_t* flag, int32_t bit, qt_stream_t stream) {
    if (n < 0 || !flag || (n > 0 && (!x || !hi || !lo))) return QT_ERR_INVALID_ARG;
    if (n == 0) return QT_OK;
    hipLaunchKernelGGL(code_digits_kernel, dim3(qt_stream_grid((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, levels, hi, lo,
                       flag, bit);
    return qt_check_launch();
}

int qt_digit_combine_f32(const float* ghi, const float* glo, const int32_t* flag, int32_t mask, float inv, float* out, int64_t n,
                         qt_stream_t stream) {
    if (n < 0 || (n > 0 && (!ghi || !glo || !out))) return QT_ERR_INVALID_ARG;
    if (n == 0) return QT_OK;
    hipLaunchKernelGGL(digit_combine_kernel, dim3(qt_stream_grid((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ghi, glo, flag, mask,
                       inv, out, n);
    return qt_check_launch();
}

int qt_binarize_f32(const float* x, float* y, int64_t n, qt_stream_t stream) {
    return launch_unary(x, y, n, stream, OpBinarize{});
}

int qt_ternarize_f32(const float* x, float* y, int64_t n, qt_stream_t stream) {
    return launch_unary(x, y, n, stream, OpTernarize{});
}

int qt_dorefa_quantize_f32(const float* x, float* y, int64_t n, int bit_width,
                           qt_stream_t stream) {
    if (bit_width < 1 || bit_width > 32) return QT_ERR_INVALID_ARG;
    if (bit_width == 1) return launch_unary(x, y, n, stream, OpBinarize{});
    if (bit_width == 32) return launch_unary(x, y, n, stream, OpCopy{});
    // the reference builds 2^k with torch.pow on fp32 tensors: exact for k <= 24, and for
    // 25..31 the fp32 value of 2^k - 1 rounds to 2^k; mirror that by forming it in fp32.
    const float two_k = (float)(1ull << bit_width);
    const float nf = two_k - 1.0f;
    OpDorefa op{nf, 1.0f / nf};
    return launch_unary(x, y, n, stream, op);
}

int qt_lin_quantize_f32(const float* x, float* y, int64_t n, int fsr, int bit_width, int mode, qt_stream_t stream) {
    if (bit_width < 1 || bit_width > 32 || mode < 0 || mode > 2 || fsr < -60 || fsr > 60) return QT_ERR_INVALID_ARG;
    if (bit_width == 32) return launch_unary(x, y, n, stream, OpCopy{});
    OpLinQuant op{ldexpf(1.0f, fsr - bit_width), ldexpf(1.0f, fsr), mode};
    return launch_unary(x, y, n, stream, op);
}

int qt_log_quantize_f32(const float* x, float* y, int64_t n, int fsr, int bit_width, int with_sign,
                        qt_stream_t stream) {
    if (bit_width < 1 || bit_width > 16 || fsr < -60 || fsr > 60) return QT_ERR_INVALID_ARG;
    OpLogQuant op{(float)fsr - (float)(1 << bit_width), (float)fsr, with_sign ? 1 : 0};
    return launch_unary(x, y, n, stream, op);
}

int qt_ap2_f32(const float* x, float* y, int64_t n, qt_stream_t stream) {
    OpLogQuant op{-INFINITY, INFINITY, 2};   // safeSign(x) * 2^round(log2|x|), no clamp
    return launch_unary(x, y, n, stream, op);
}

int qt_binarize_stochastic_f32(const float* x, const float* z, float* y, int64_t n,
                               qt_stream_t stream) {
    return launch_binary(x, z, y, n, stream, Op2BinStoch{});
}

int qt_ternarize_stochastic_f32(const float* x, const float* z, float* y, int64_t n,
                                qt_stream_t stream) {
    return launch_binary(x, z, y, n, stream, Op2TerStoch{});
}

int qt_ste_mask_f32(const float* gout, const float* x, float* gin, int64_t n, float thr,
                    qt_stream_t stream) {
    return launch_binary(gout, x, gin, n, stream, Op2SteMask{thr});
}

int qt_poison_f32(const float* x, const int32_t* flag, int32_t mask, float* out, int64_t n, qt_stream_t stream) {
    if (n < 0 || !flag || (n > 0 && !out)) return QT_ERR_INVALID_ARG;
    if (n == 0) return QT_OK;
    hipLaunchKernelGGL(poison_kernel, dim3(qt_stream_grid((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, flag, mask, out, n);
    return qt_check_launch();
}

}  // extern "C"
