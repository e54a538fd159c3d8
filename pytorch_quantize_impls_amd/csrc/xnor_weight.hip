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

}  // namespace

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
