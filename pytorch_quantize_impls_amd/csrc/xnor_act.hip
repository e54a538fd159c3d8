// XNOR-Net activation quantiser on 2-D activations (functions/xnor_connect.py:17-66: _quantOpXnor / nnQuantXnor /
// QuantXnor):
//     forward   y = sign(x) * mean(x, dim)                       (torch.sign: 0 -> 0, NaN -> NaN; the SIGNED mean, as
//                                                                  upstream computes it despite its docstring, :21-28)
//     backward  gin = sign(x) * mean(g * sign(x), dim, keepdim) + g * mean       (:30-37)
// dim = 1: one mean per row; dim = 0: one per column; dim = -1: one for the whole tensor.
// HBM-bound: two passes over x (reduce, then scale) — 8 B read + 4 B written per element forward.  The reduction order
// differs from torch's, so the result sits in the float-tail class (tolerance 1e-5 normalised, tests).
#include "qt_common.h"

namespace {

__device__ __forceinline__ float torch_sign_f(float x) {
    return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : x);  // +-0 -> +-0, NaN -> NaN
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// value reduced: x (forward) or g * sign(x) (backward)
template <bool PROD>
__device__ __forceinline__ float red_val(const float* __restrict__ x, const float* __restrict__ g, int64_t ix, int64_t ig) {
    return PROD ? g[ig] * torch_sign_f(x[ix]) : x[ix];
}

// dim = 1: one wave per row, four rows per workgroup; lanes stride the row (coalesced 256-byte segments)
template <bool PROD>
__global__ __launch_bounds__(256) void row_mean_kernel(const float* __restrict__ x, int64_t ldx,
                                                       const float* __restrict__ g, int64_t ldg,
                                                       float* __restrict__ mean, int64_t R, int64_t C) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float acc = 0.0f;
    for (int64_t c = lane; c < C; c += 64) acc += red_val<PROD>(x, g, r * ldx + c, r * ldg + c);
    acc = wave_sum(acc);
    if (lane == 0) mean[r] = acc / (float)C;
}

// dim = 0: one workgroup of 16 waves per 64-column strip; wave w walks rows w, w+16, ...; partial sums meet in LDS
template <bool PROD>
__global__ __launch_bounds__(1024) void col_mean_kernel(const float* __restrict__ x, int64_t ldx,
                                                        const float* __restrict__ g, int64_t ldg,
                                                        float* __restrict__ mean, int64_t R, int64_t C) {
    __shared__ float part[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + tx;
    float acc = 0.0f;
    if (c < C)
        for (int64_t r = ty; r < R; r += 16) acc += red_val<PROD>(x, g, r * ldx + c, r * ldg + c);
    part[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && c < C) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += part[i][tx];
        mean[c] = s / (float)R;
    }
}

// dim = -1, stage 1: per-workgroup partial sums into work[blockIdx]; stage 2 (one workgroup) adds them in a fixed
// order, so the result does not depend on scheduling
template <bool PROD>
__global__ __launch_bounds__(256) void all_partial_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ g, int64_t ldg,
                                                          float* __restrict__ work, int64_t R, int64_t C) {
    __shared__ float part[4];
    const int64_t total = R * C;
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C, c = i - r * C;
        acc += red_val<PROD>(x, g, r * ldx + c, r * ldg + c);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) work[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(256) void all_final_kernel(const float* __restrict__ work, int nparts, float* __restrict__ mean,
                                                        float inv_count) {
    __shared__ float part[4];
    float acc = 0.0f;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += work[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) mean[0] = (part[0] + part[1] + part[2] + part[3]) * inv_count;
}

// y = sign(x) * mean[idx]   /   gin = sign(x) * gmean[idx] + g * mean[idx]
template <bool BWD>
__global__ __launch_bounds__(256) void xnor_scale_kernel(const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ g, int64_t ldg,
                                                         const float* __restrict__ mean, const float* __restrict__ gmean,
                                                         float* __restrict__ out, int64_t ldo, int64_t R, int64_t C, int dim) {
    const int64_t total = R * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / C, c = i - r * C;
        const int64_t m = dim == 1 ? r : (dim == 0 ? c : 0);
        const float s = torch_sign_f(x[r * ldx + c]);
        out[r * ldo + c] = BWD ? s * gmean[m] + g[r * ldg + c] * mean[m] : s * mean[m];
    }
}

constexpr int XA_PARTS = 1024;   // workgroups (= partial sums) of the dim = -1 reduction

template <bool PROD>
void launch_mean(const float* x, int64_t ldx, const float* g, int64_t ldg, float* mean, float* work, int64_t R, int64_t C,
                 int dim, hipStream_t st) {
    if (dim == 1) {
        hipLaunchKernelGGL((row_mean_kernel<PROD>), dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, x, ldx, g, ldg, mean, R, C);
    } else if (dim == 0) {
        hipLaunchKernelGGL((col_mean_kernel<PROD>), dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, st, x, ldx, g, ldg, mean, R, C);
    } else {
        const int parts = qt_stream_grid((R * C + 255) / 256, XA_PARTS);
        hipLaunchKernelGGL((all_partial_kernel<PROD>), dim3(parts), dim3(256), 0, st, x, ldx, g, ldg, work, R, C);
        hipLaunchKernelGGL(all_final_kernel, dim3(1), dim3(256), 0, st, work, parts, mean, 1.0f / ((float)R * (float)C));
    }
}

int check_args(const float* x, int64_t ldx, const float* mean, const float* work, int64_t R, int64_t C, int dim) {
    if (R < 0 || C < 0 || (dim != -1 && dim != 0 && dim != 1)) return QT_ERR_INVALID_ARG;
    if (R == 0 || C == 0) return 1;
    if (!x || !mean || ldx < C || (dim == -1 && !work)) return QT_ERR_INVALID_ARG;
    if (R * C > (1ll << 40)) return QT_ERR_UNSUPPORTED;
    return QT_OK;
}


// ---- input quantiser of XNORConv2d(quant_input=True): y = sign(x) * mean(|x|, channel dim) per pixel ---------------------------------
// (functions/xnor_connect.py:142-143; torch.sign: +-0 -> +-0, NaN -> NaN).  x: logical [N, C, H, W] with element strides
// (sn, sc, sh, sw); y: the same logical tensor written in NHWC memory order (what the per-tap scaled conv's operand pack and
// the weight-gradient routes read).  HBM-bound: one read of x (the second walk of a pixel's channels hits L1 / L2), one write.
//   channels-last input (sc == 1): G = 4 .. 64 lanes share a pixel (lanes along the channels: coalesced), butterfly sum;
//   NCHW input: one thread per pixel (lanes along w: every channel's read is coalesced), the NHWC writes are strided.
template <int G>
__global__ __launch_bounds__(256) void xnor_input_quant_cl_kernel(const float* __restrict__ x, int64_t sn, int64_t sh, int64_t sw,
                                                                  float* __restrict__ y, float* __restrict__ aplane, int64_t P, int C,
                                                                  int H, int W) {
    const int sub = threadIdx.x % G;
    const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G, ngroups = (int64_t)gridDim.x * blockDim.x / G;
    const float inv_note = (float)C;
    for (int64_t p = gid; p < P; p += ngroups) {
        const int64_t n = p / ((int64_t)H * W), r = p - n * (int64_t)H * W;
        const int h = (int)(r / W), w = (int)(r - (int64_t)h * W);
        const float* px = x + n * sn + h * sh + w * sw;
        float acc = 0.0f;
        for (int c = sub; c < C; c += G) acc += fabsf(px[c]);
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, G);
        const float a = acc / inv_note;
        if (aplane && sub == 0) aplane[p] = a;
        if (!y) continue;
        float* py = y + p * C;
        for (int c = sub; c < C; c += G) {
            const float v = px[c];
            py[c] = (v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : v)) * a;
        }
    }
}

__global__ __launch_bounds__(256) void xnor_input_quant_strided_kernel(const float* __restrict__ x, int64_t sn, int64_t sc, int64_t sh,
                                                                       int64_t sw, float* __restrict__ y, float* __restrict__ aplane,
                                                                       int64_t P, int C, int H, int W) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = p / ((int64_t)H * W), r = p - n * (int64_t)H * W;
        const int h = (int)(r / W), w = (int)(r - (int64_t)h * W);
        const float* px = x + n * sn + h * sh + w * sw;
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) acc += fabsf(px[c * sc]);
        const float a = acc / (float)C;
        if (aplane) aplane[p] = a;
        if (!y) continue;
        float* py = y + p * C;
        for (int c = 0; c < C; ++c) {
            const float v = px[c * sc];
            py[c] = (v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : v)) * a;
        }
    }
}

}  // namespace

extern "C" {

int64_t qt_xnor_act_work_floats(void) { return XA_PARTS; }

int qt_xnor_act_f32(const float* x, int64_t ldx, float* mean, float* work, float* y, int64_t ldy, int64_t R, int64_t C,
                    int dim, qt_stream_t stream) {
    const int rc = check_args(x, ldx, mean, work, R, C, dim);
    if (rc != QT_OK) return rc > 0 ? QT_OK : rc;
    if (!y || ldy < C) return QT_ERR_INVALID_ARG;
    launch_mean<false>(x, ldx, nullptr, 0, mean, work, R, C, dim, (hipStream_t)stream);
    hipLaunchKernelGGL((xnor_scale_kernel<false>), dim3(qt_stream_grid((R * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, nullptr, (int64_t)0, mean, nullptr, y, ldy, R, C, dim);
    return qt_check_launch();
}

int qt_xnor_act_backward_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, const float* mean, float* gmean,
                             float* work, float* gin, int64_t ldi, int64_t R, int64_t C, int dim, qt_stream_t stream) {
    const int rc = check_args(x, ldx, mean, work, R, C, dim);
    if (rc != QT_OK) return rc > 0 ? QT_OK : rc;
    if (!g || !gmean || !gin || ldg < C || ldi < C) return QT_ERR_INVALID_ARG;
    launch_mean<true>(x, ldx, g, ldg, gmean, work, R, C, dim, (hipStream_t)stream);
    hipLaunchKernelGGL((xnor_scale_kernel<true>), dim3(qt_stream_grid((R * C + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, ldx, g, ldg, mean, gmean, gin, ldi, R, C, dim);
    return qt_check_launch();
}

int qt_xnor_input_quant_f32(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, float* y, float* a_plane, int64_t N,
                            int64_t C, int64_t H, int64_t W, qt_stream_t stream) {
    if (N < 0 || C < 0 || H < 0 || W < 0) return QT_ERR_INVALID_ARG;
    const int64_t P = N * H * W;
    if (P == 0 || C == 0) return QT_OK;
    if (!x || (!y && !a_plane) || C > (1 << 24) || H > (1 << 24) || W > (1 << 24)) return QT_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (sc == 1) {
        const int G = C >= 48 ? 64 : C >= 24 ? 32 : C >= 12 ? 16 : C >= 6 ? 8 : 4;
        const int grid = qt_stream_grid((P * G + 255) / 256);
#define QT_XIQ(g) hipLaunchKernelGGL((xnor_input_quant_cl_kernel<g>), dim3(grid), dim3(256), 0, st, x, sn, sh, sw, y, a_plane, P, (int)C, (int)H, (int)W)
        switch (G) {
            case 64: QT_XIQ(64); break;
            case 32: QT_XIQ(32); break;
            case 16: QT_XIQ(16); break;
            case 8: QT_XIQ(8); break;
            default: QT_XIQ(4); break;
        }
#undef QT_XIQ
    } else {
        hipLaunchKernelGGL(xnor_input_quant_strided_kernel, dim3(qt_stream_grid((P + 255) / 256)), dim3(256), 0, st, x, sn, sc, sh, sw,
                           y, a_plane, P, (int)C, (int)H, (int)W);
    }
    return qt_check_launch();
}

}  // extern "C"
