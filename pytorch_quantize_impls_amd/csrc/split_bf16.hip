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

// mode: 0 = activation split, 1 = safeSign weight, 2 = ternary weight, 3 = torch.sign weight (0 -> 0),
//       4 = raw weight: bf16_rn(w) replicated (for weights that are exact in bf16: Lin / Log fixed-point levels)
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
                else if (MODE == 3) q = v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f);
                else q = v;
                const uint32_t qb = bf16_rn_bits(q);
                h[3 * e] = h[3 * e + 1] = h[3 * e + 2] = qb;
            }
        }
        orow[p * 3 + 0] = h[0] | (h[1] << 16);
        orow[p * 3 + 1] = h[2] | (h[3] << 16);
        orow[p * 3 + 2] = h[4] | (h[5] << 16);
    }
}

// Six-term planes for REAL x REAL products (XNOR-Net convs: weights sign(W) * alpha are not bf16 values).
// With x = xh + xm + xl and w = wh + wm + wl (exact bf16 splits), the six largest of the nine cross terms,
//     xh*wh + xh*wm + xh*wl + xm*wh + xm*wm + xl*wh,
// reproduce x*w to ~2^-24 relative (the dropped terms are <= 2^-25 |x w|), each product exact in the fp32
// accumulator.  Element k occupies bf16 slots 6k..6k+5:  role 0 (activation) [xh xh xh xm xm xl],
// role 1 (weight) [wh wm wl wh wm wh], so a plain bf16 GEMM over 6K of the two planes pairs them term by term.
template <int ROLE>
__global__ __launch_bounds__(256) void sext_kernel(const float* __restrict__ x, int64_t ldx, uint16_t* __restrict__ out,
                                                   int64_t ld_elems, int64_t rows, int64_t K) {
    const int64_t per_row = ld_elems / 6;                     // one work item = 1 element = 6 bf16 = 12 B
    const int64_t tail_words = (ld_elems - per_row * 6) / 2;
    const int64_t total = rows * (per_row + 1);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / (per_row + 1), k = t - row * (per_row + 1);
        uint32_t* orow = reinterpret_cast<uint32_t*>(out + row * ld_elems);
        if (k == per_row) {
            for (int64_t w = 0; w < tail_words; ++w) orow[per_row * 3 + w] = 0;
            continue;
        }
        uint32_t h = 0, m = 0, l = 0;
        if (k < K) {
            const float v = x[row * ldx + k];
            h = bf16_rn_bits(v);
            const float r1 = v - bf16_bits_to_f32(h);
            m = bf16_rn_bits(r1);
            l = bf16_rn_bits(r1 - bf16_bits_to_f32(m));
        }
        if (ROLE == 0) {
            orow[k * 3 + 0] = h | (h << 16);
            orow[k * 3 + 1] = h | (m << 16);
            orow[k * 3 + 2] = m | (l << 16);
        } else {
            orow[k * 3 + 0] = h | (m << 16);
            orow[k * 3 + 1] = l | (h << 16);
            orow[k * 3 + 2] = m | (h << 16);
        }
    }
}

// Space-to-depth gather + split in one pass: pixel (n, Y, X) of the output plane holds, for every
// e = (c*s + dy)*s + dx, the triple of x[n, c, s*Y + dy - ph, s*X + dx - pw] (zero outside the image).
// A strided first-layer conv (k x k, stride s, padding p) on x equals a stride-1 ceil(k/s)^2 conv on this
// plane.  x is addressed through element strides, so NCHW and NHWC storage both work without a copy.
__global__ __launch_bounds__(256) void s2d_triple_kernel(const float* __restrict__ x, int64_t sN, int64_t sC,
                                                         int64_t sH, int64_t sW, uint16_t* __restrict__ out,
                                                         int64_t ld_elems, int64_t N, int C, int H, int W,
                                                         int s, int ph, int pw, int Hs, int Ws) {
    const int E = C * s * s;                      // s2d channels
    const int64_t pairs_per_row = ld_elems / 6;  // 2 elements (6 bf16) per work item; ld_elems % 8 == 0
    const int64_t tail_words = (ld_elems - pairs_per_row * 6) / 2;
    const int64_t total = N * Hs * Ws * (pairs_per_row + 1);
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = t / (pairs_per_row + 1), p = t - pix * (pairs_per_row + 1);
        uint32_t* orow = reinterpret_cast<uint32_t*>(out + pix * ld_elems);
        if (p == pairs_per_row) {
            for (int64_t w = 0; w < tail_words; ++w) orow[pairs_per_row * 3 + w] = 0;
            continue;
        }
        const int64_t n = pix / ((int64_t)Hs * Ws);
        const int rem = (int)(pix - n * Hs * Ws);
        const int Y = rem / Ws, X = rem - Y * Ws;
        uint32_t h[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
            const int e = (int)p * 2 + e2;
            if (e >= E) continue;
            const int c = e / (s * s), r = e - c * s * s, dy = r / s, dx = r - dy * s;
            const int hh = s * Y + dy - ph, ww = s * X + dx - pw;
            float v = 0.0f;
            if (hh >= 0 && hh < H && ww >= 0 && ww < W) v = x[n * sN + c * sC + hh * sH + ww * sW];
            const uint32_t a = bf16_rn_bits(v);
            const float r1 = v - bf16_bits_to_f32(a);
            const uint32_t b = bf16_rn_bits(r1);
            const float r2 = r1 - bf16_bits_to_f32(b);
            h[3 * e2] = a; h[3 * e2 + 1] = b; h[3 * e2 + 2] = bf16_rn_bits(r2);
        }
        orow[p * 3 + 0] = h[0] | (h[1] << 16);
        orow[p * 3 + 1] = h[2] | (h[3] << 16);
        orow[p * 3 + 2] = h[4] | (h[5] << 16);
    }
}


// Same result, organised for the memory system (channels-last images: sC == 1, sW == C): one workgroup per
// output row (n, Y).  Phase 1 copies the s input rows the row needs into LDS with full-line coalesced
// loads (rows outside the image become zeros); phase 2 produces the output row 16 bytes per lane per step
// (fully coalesced dwordx4 stores), gathering its up-to-4 source values from LDS through a per-workgroup
// table e -> (LDS offset, column displacement) so no per-element division is left.  A 16-byte chunk holds
// bf16 positions p0 .. p0+7 with p0 = 8*cq; its alignment against the 3-term triples has three phases.
__device__ __forceinline__ void split3(float v, uint32_t (&t)[3]) {
    t[0] = bf16_rn_bits(v);
    const float r1 = v - bf16_bits_to_f32(t[0]);
    t[1] = bf16_rn_bits(r1);
    t[2] = bf16_rn_bits(r1 - bf16_bits_to_f32(t[1]));
}

__global__ __launch_bounds__(256) void s2d_triple_rows_kernel(const float* __restrict__ x, int64_t sN, int64_t sH,
                                                              uint16_t* __restrict__ out, int64_t ld_bytes,
                                                              int C, int H, int W, int s, int ph, int pw, int Hs,
                                                              int Ws, int vec_ok) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s2d_smem[];
    const int E = C * s * s, rowf = W * C, rowf4 = (rowf + 3) & ~3;
    float* rows = reinterpret_cast<float*>(s2d_smem);                     // [s][rowf4]
    int* lut_off = reinterpret_cast<int*>(rows + (size_t)s * rowf4);      // [E] dy*rowf4 + (dx - pw)*C + c
    int* lut_dx = lut_off + E;                                            // [E] dx - pw
    const int tid = threadIdx.x;
    const int n = blockIdx.x / Hs, Y = blockIdx.x - n * Hs;
    for (int e = tid; e < E; e += 256) {
        const int c = e / (s * s), r = e - c * s * s, dy = r / s, dx = r - dy * s;
        lut_off[e] = dy * rowf4 + (dx - pw) * C + c;
        lut_dx[e] = dx - pw;
    }
    for (int dy = 0; dy < s; ++dy) {
        const int hh = s * Y + dy - ph;
        const bool ok = hh >= 0 && hh < H;
        const float* src = x + (int64_t)n * sN + (int64_t)(ok ? hh : 0) * sH;
        float* dst = rows + dy * rowf4;
        if (vec_ok) {
            for (int i = tid * 4; i < rowf; i += 1024)
                *reinterpret_cast<float4*>(dst + i) = ok ? *reinterpret_cast<const float4*>(src + i) : make_float4(0, 0, 0, 0);
        } else {
            for (int i = tid; i < rowf; i += 256) dst[i] = ok ? src[i] : 0.0f;
        }
    }
    __syncthreads();
    const int cpp = (int)(ld_bytes >> 4), total = Ws * cpp;
    uint4* orow = reinterpret_cast<uint4*>(reinterpret_cast<unsigned char*>(out) + ((int64_t)blockIdx.x * Ws) * ld_bytes);
    for (int q = tid; q < total; q += 256) {
        const int X = q / cpp, cq = q - X * cpp;
        const int p0 = cq * 8;
        const int e0 = (int)(((unsigned)p0 * 43691u) >> 17);      // p0 / 3 (exact for p0 < 98304)
        const int ph3 = p0 - 3 * e0;                              // 0, 1, 2: first position's term index
        const int base = s * X * C, wx = s * X;
        uint32_t t[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = e0 + j;
            float v = 0.0f;
            if (e < E) {
                const int ww = wx + lut_dx[e];
                if ((unsigned)ww < (unsigned)W) v = rows[lut_off[e] + base];
            }
            split3(v, t[j]);
        }
        uint4 o;
        if (ph3 == 0) {
            o.x = t[0][0] | (t[0][1] << 16); o.y = t[0][2] | (t[1][0] << 16);
            o.z = t[1][1] | (t[1][2] << 16); o.w = t[2][0] | (t[2][1] << 16);
        } else if (ph3 == 1) {
            o.x = t[0][1] | (t[0][2] << 16); o.y = t[1][0] | (t[1][1] << 16);
            o.z = t[1][2] | (t[2][0] << 16); o.w = t[2][1] | (t[2][2] << 16);
        } else {
            o.x = t[0][2] | (t[1][0] << 16); o.y = t[1][1] | (t[1][2] << 16);
            o.z = t[2][0] | (t[2][1] << 16); o.w = t[2][2] | (t[3][0] << 16);
        }
        orow[q] = o;
    }
}

}  // namespace

extern "C" int qt_bf16x3_s2d_pack_f32(const float* x, int64_t sN, int64_t sC, int64_t sH, int64_t sW,
                                      uint16_t* out, int64_t ld_bytes, int64_t N, int64_t C, int64_t H,
                                      int64_t W, int64_t s, int64_t ph, int64_t pw, qt_stream_t stream) {
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || s < 1 || ph < 0 || pw < 0) return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!x || !out) return QT_ERR_INVALID_ARG;
    const int64_t E = C * s * s;
    if (ld_bytes < 6 * E || (ld_bytes & 15) || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    if (H > 32767 || W > 32767 || E > 4096) return QT_ERR_UNSUPPORTED;
    const int64_t Hs = (H + 2 * ph + s - 1) / s, Ws = (W + 2 * pw + s - 1) / s;
    const int64_t ld_elems = ld_bytes / 2;
    // channels-last images take the row-staged kernel (coalesced on both sides)
    const int64_t rowf4 = (W * C + 3) & ~3ll;
    const int64_t lds = s * rowf4 * 4 + E * 8;
    if (sC == 1 && sW == C && lds <= 60 * 1024 && N * Hs < (1ll << 31) && ld_bytes / 2 < 98304 && sH >= W * C) {
        const int vec_ok = ((W * C) % 4 == 0) && qt_aligned16(x) && (sH % 4 == 0) && (sN % 4 == 0);
        hipLaunchKernelGGL(s2d_triple_rows_kernel, dim3((unsigned)(N * Hs)), dim3(256), (size_t)lds,
                           (hipStream_t)stream, x, sN, sH, out, ld_bytes, (int)C, (int)H, (int)W, (int)s, (int)ph,
                           (int)pw, (int)Hs, (int)Ws, vec_ok);
        return qt_check_launch();
    }
    const int64_t total = N * Hs * Ws * (ld_elems / 6 + 1);
    const int grid = qt_stream_grid((total + 255) / 256);
    hipLaunchKernelGGL(s2d_triple_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, sN, sC, sH, sW,
                       out, ld_elems, N, (int)C, (int)H, (int)W, (int)s, (int)ph, (int)pw, (int)Hs, (int)Ws);
    return qt_check_launch();
}

extern "C" int qt_bf16x6_pack_f32(const float* x, int64_t ldx, uint16_t* out, int64_t ld_bytes, int64_t rows,
                                  int64_t K, int role, qt_stream_t stream) {
    if (rows < 0 || K < 0 || ldx < K || role < 0 || role > 1) return QT_ERR_INVALID_ARG;
    if (rows == 0) return QT_OK;
    if (!out || (!x && K > 0)) return QT_ERR_INVALID_ARG;
    if (ld_bytes < 12 * K || (ld_bytes & 15) || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    if (ld_bytes == 0) return QT_OK;
    const int64_t ld_elems = ld_bytes / 2;
    const int grid = qt_stream_grid((rows * (ld_elems / 6 + 1) + 255) / 256);
    if (role == 0)
        hipLaunchKernelGGL((sext_kernel<0>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ld_elems, rows, K);
    else
        hipLaunchKernelGGL((sext_kernel<1>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, out, ld_elems, rows, K);
    return qt_check_launch();
}

extern "C" int qt_bf16x3_pack_f32(const float* x, int64_t ldx, const float* alpha, uint16_t* out,
                                  int64_t ld_bytes, int64_t rows, int64_t K, int mode, qt_stream_t stream) {
    if (rows < 0 || K < 0 || ldx < K || mode < 0 || mode > 4) return QT_ERR_INVALID_ARG;
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
        case 3: QT_LAUNCH(3); break;
        default: QT_LAUNCH(4); break;
    }
#undef QT_LAUNCH
    return qt_check_launch();
}
