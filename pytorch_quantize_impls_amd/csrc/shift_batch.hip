// Shift-based batch normalisation primitive (functions/binary_connect.py:173-186, ShiftBatch.forward; used by
// ShiftNormBatch1d / ShiftNormBatch2d, layers/binary_layers.py:110-160):
//     mu   = x - running_mean
//     sv   = sqrt(running_var + eps)
//     norm = mu * AP2(1 / sv)                    AP2(z) = safeSign(z) * 2^round(log2|z|)   (:157-169)
//     y    = norm * AP2(weight) + bias
// x: [N, E] row-major (E = all trailing dimensions; the statistics / affine tensors have E entries and broadcast over
// N, which is how both layers call it).  One pass: 4 B read + 4 B (y) [+ 4 B norm, kept for the weight gradient] per
// element — HBM-bound; a thread owns four consecutive e, folds the four per-feature scalars once and walks the rows.
// Every product / sum is a separate fp32 rounding as in the reference's torch expression (-ffp-contract=off); log2f is
// the device's (same caveat as qt_ap2_f32: inputs within an ulp of 2^(k+1/2) may round to the other neighbour).
#include "qt_common.h"

namespace {

__device__ __forceinline__ float ap2_dev(float z) {
    return qt_safe_sign(z) * exp2f(rintf(log2f(fabsf(z))));
}

template <bool VEC>
__global__ __launch_bounds__(256) void shift_batch_kernel(const float* __restrict__ x, int64_t ldx,
                                                          const float* __restrict__ mean, const float* __restrict__ var,
                                                          const float* __restrict__ weight, const float* __restrict__ bias,
                                                          float eps, float* __restrict__ y, int64_t ldy,
                                                          float* __restrict__ norm, int64_t ldn,
                                                          float* __restrict__ sqrtvar, int64_t N, int64_t E, int rows_per_block) {
    constexpr int W = VEC ? 4 : 1;
    const int64_t e0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * W;
    if (e0 >= E) return;
    float mu[W], a[W], w2[W], b[W];
#pragma unroll
    for (int c = 0; c < W; ++c) {
        const int64_t e = e0 + c;
        mu[c] = mean[e];
        const float sv = sqrtf(var[e] + eps);
        a[c] = ap2_dev(1.0f / sv);
        w2[c] = ap2_dev(weight[e]);
        b[c] = bias[e];
        if (sqrtvar && blockIdx.y == 0) sqrtvar[e] = sv;
    }
    const int64_t n_lo = (int64_t)blockIdx.y * rows_per_block, n_hi = min(N, n_lo + rows_per_block);
    for (int64_t n = n_lo; n < n_hi; ++n) {
        float v[W], nv[W], o[W];
        if constexpr (VEC) {
            const float4 t = *reinterpret_cast<const float4*>(x + n * ldx + e0);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
            v[0] = x[n * ldx + e0];
        }
#pragma unroll
        for (int c = 0; c < W; ++c) {
            nv[c] = (v[c] - mu[c]) * a[c];
            o[c] = nv[c] * w2[c] + b[c];
        }
        if constexpr (VEC) {
            *reinterpret_cast<float4*>(y + n * ldy + e0) = make_float4(o[0], o[1], o[2], o[3]);
            if (norm) *reinterpret_cast<float4*>(norm + n * ldn + e0) = make_float4(nv[0], nv[1], nv[2], nv[3]);
        } else {
            y[n * ldy + e0] = o[0];
            if (norm) norm[n * ldn + e0] = nv[0];
        }
    }
}

}  // namespace

extern "C" int qt_shift_batch_f32(const float* x, int64_t ldx, const float* running_mean, const float* running_var,
                                  const float* weight, const float* bias, float eps, float* y, int64_t ldy, float* norm,
                                  int64_t ldn, float* sqrtvar, int64_t N, int64_t E, qt_stream_t stream) {
    if (N < 0 || E < 0) return QT_ERR_INVALID_ARG;
    if (N == 0 || E == 0) return QT_OK;
    if (!x || !running_mean || !running_var || !weight || !bias || !y || ldx < E || ldy < E || (norm && ldn < E))
        return QT_ERR_INVALID_ARG;
    const bool vec = (E % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (!norm || ldn % 4 == 0) && qt_aligned16(x) &&
                     qt_aligned16(y) && (!norm || qt_aligned16(norm));
    const int64_t lanes = vec ? E / 4 : E;
    const int64_t gx = (lanes + 255) / 256;
    if (gx > INT32_MAX) return QT_ERR_UNSUPPORTED;
    // enough workgroups to fill the chip: split the rows when there are few feature strips
    int64_t gy = 1;
    while (gx * gy < 2048 && gy * 8 <= N && gy < 65535) gy *= 2;
    const int rows_per_block = (int)((N + gy - 1) / gy);
    const dim3 grid((unsigned)gx, (unsigned)((N + rows_per_block - 1) / rows_per_block));
    if (vec)
        hipLaunchKernelGGL((shift_batch_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, running_mean, running_var,
                           weight, bias, eps, y, ldy, norm, ldn, sqrtvar, N, E, rows_per_block);
    else
        hipLaunchKernelGGL((shift_batch_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, x, ldx, running_mean, running_var,
                           weight, bias, eps, y, ldy, norm, ldn, sqrtvar, N, E, rows_per_block);
    return qt_check_launch();
}
