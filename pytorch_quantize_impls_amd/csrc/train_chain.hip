// Training-mode form of the chain the reference puts between two binarised layers (VERDICT r2 missing #4):
//     [MaxPool2d(k, s)] -> BatchNorm (batch statistics) -> [Hardtanh] -> BinaryConnectDeterministic
// (models/Alexnet/Alexnet_Bin.py:13-17, benchmark/BinaryNet/MLPBin.py:42-44), forward AND backward, as HBM-bound passes over
// channels-last fp32 tensors — in the module graph these are torch / MIOpen kernels (max_pool2d, batch_norm, hardtanh and
// their backwards: 3.2 ms of the 23 ms AlexNet-Bin training step).
//
// x is NHWC fp32 [N][H][W][C]; "rows" are the pooled pixels r = (n, ho, wo), R = N * Ho * Wo (k = 1: the pixels themselves;
// a [N, C] matrix is N rows).  Forward
//     p[r, c]  = max over the k x k window (first maximum in scan order, NaN propagates: torch's rule), idx = its position
//     mean_c   = sum_r p / R ;  var_c = sum_r (p - mean_c)^2 / R  (two passes over p: no cancellation) ;  invstd = 1/sqrt(var+eps)
//     xhat     = (p - mean) * invstd ;  y = xhat * gamma + beta ;  h = clamp(y, lo, hi) ;  s = h < 0 ? -1 : +1
//     running_mean = (1 - m) running_mean + m mean ;  running_var = (1 - m) running_var + m var R / (R - 1)
// Backward (g = dL/ds)
//     g_h = g * 1[|h| <= 1.001]                  (STE of BinaryConnect, functions/binary_connect.py:31-38)
//     g_y = g_h * 1[lo < y < hi]                 (Hardtanh; no Hardtanh: lo = -inf, hi = +inf)
//     dgamma = sum_r g_y xhat ;  dbeta = sum_r g_y ;  g_p = gamma invstd (g_y - dbeta / R - xhat dgamma / R)
//     g_x[n, h, w, c] = sum of g_p over the windows whose idx names (h, w)      (gather: no atomics)
// Per-channel sums: a workgroup owns a slab of rows and all channels (thread <-> channel, coalesced along c), writes one partial
// per channel; a second tiny launch folds the partials in double.  Everything is elementwise or a column reduction:
// algorithmic bytes = what each pass reads + writes, all at the HBM rate.
#include "qt_common.h"

namespace {

constexpr int TC_ROWS_PER_BLOCK_MIN = 8;

__device__ __forceinline__ float tc_y(float p, float mean, float invstd, float gamma, float beta) {
    return (p - mean) * invstd * gamma + beta;          // ((p - mean) * invstd) * gamma + beta, two-rounding steps as torch's
}

// ---- forward pass 1: pooling (+ argmax) and per-block channel sums of p ------------------------------------------------------
__global__ __launch_bounds__(256) void pool_sum_kernel(const float* __restrict__ x, float* __restrict__ p, int8_t* __restrict__ idx,
                                                       float* __restrict__ part, int64_t R, int H, int W, int C, int k, int s,
                                                       int Ho, int Wo, int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc = 0.0f;
        for (int64_t r = r0; r < r1; ++r) {
            float m;
            if (k == 1) {
                m = x[r * C + c];
            } else {
                const int64_t n = r / ((int64_t)Ho * Wo);
                const int rem = (int)(r - n * Ho * Wo);
                const int ho = rem / Wo, wo = rem - ho * Wo;
                const float* base = x + ((n * H + (int64_t)ho * s) * W + (int64_t)wo * s) * C + c;
                m = base[0];
                int best = 0;
                for (int i = 0; i < k; ++i)
                    for (int j = 0; j < k; ++j) {
                        if (i == 0 && j == 0) continue;
                        const float v = base[((int64_t)i * W + j) * C];
                        if (v > m || v != v) { m = v; best = i * k + j; }
                    }
                p[r * C + c] = m;
                idx[r * C + c] = (int8_t)best;
            }
            acc += m;
        }
        part[(int64_t)blockIdx.x * C + c] = acc;
    }
}

// ---- forward pass 2: per-block channel sums of (p - mean)^2 ---------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqdev_kernel(const float* __restrict__ p, const float* __restrict__ mean,
                                                    float* __restrict__ part, int64_t R, int C, int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        const float mu = mean[c];
        float acc = 0.0f;
        for (int64_t r = r0; r < r1; ++r) {
            const float d = p[r * C + c] - mu;
            acc += d * d;
        }
        part[(int64_t)blockIdx.x * C + c] = acc;
    }
}

// fold partials [nblk][C] in double: out[c] = scale * sum  (scale = 1 / R for a mean)
__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ part, int nblk, int C, double scale,
                                                   float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)part[(int64_t)b * C + c];
    out[c] = (float)(s * scale);
}

// var partials -> invstd, running statistics
__global__ __launch_bounds__(256) void finalize_kernel(const float* __restrict__ part, int nblk, int C, double R, float eps,
                                                       float momentum, const float* __restrict__ mean, float* __restrict__ invstd,
                                                       float* __restrict__ running_mean, float* __restrict__ running_var) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)part[(int64_t)b * C + c];
    const float var = (float)(s / R);
    invstd[c] = 1.0f / sqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean[c];
    if (running_var) {
        const float unbiased = R > 1.0 ? (float)(s / (R - 1.0)) : var;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// ---- forward pass 3: normalise, clamp, sign ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void norm_sign_kernel(const float* __restrict__ p, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float lo, float hi,
                                                        float* __restrict__ sgn, int64_t total, int C) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        float y = tc_y(p[t], mean[c], invstd[c], gamma ? gamma[c] : 1.0f, beta ? beta[c] : 0.0f);
        y = y < lo ? lo : (y > hi ? hi : y);               // NaN stays NaN
        sgn[t] = qt_safe_sign(y);
    }
}

// ---- backward pass 1: per-block channel sums of g_y and g_y * xhat -----------------------------------------------------------------
__device__ __forceinline__ float tc_gy(float g, float y, float lo, float hi, float ste) {
    const float h = y < lo ? lo : (y > hi ? hi : y);
    float gy = fabsf(h) <= ste ? g : 0.0f;                // STE of the sign
    if (!(y > lo && y < hi)) gy = 0.0f;                   // Hardtanh (strict, as torch's hardtanh_backward)
    return gy;
}

__global__ __launch_bounds__(256) void bwd_sum_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float lo,
                                                      float hi, float ste, float* __restrict__ part_b, float* __restrict__ part_g,
                                                      int64_t R, int C, int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        const float mu = mean[c], is = invstd[c], ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
        float sb = 0.0f, sg = 0.0f;
        for (int64_t r = r0; r < r1; ++r) {
            const float xh = (p[r * C + c] - mu) * is;
            const float gy = tc_gy(g[r * C + c], xh * ga + be, lo, hi, ste);
            sb += gy;
            sg += gy * xh;
        }
        part_b[(int64_t)blockIdx.x * C + c] = sb;
        part_g[(int64_t)blockIdx.x * C + c] = sg;
    }
}

// ---- backward pass 2: g_p ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bwd_dx_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ dgamma, const float* __restrict__ dbeta, float lo,
                                                     float hi, float ste, float invR, float* __restrict__ gp, int64_t total, int C) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const float is = invstd[c], ga = gamma ? gamma[c] : 1.0f, be = beta ? beta[c] : 0.0f;
        const float xh = (p[t] - mean[c]) * is;
        const float gy = tc_gy(g[t], xh * ga + be, lo, hi, ste);
        gp[t] = ga * is * (gy - dbeta[c] * invR - xh * dgamma[c] * invR);
    }
}

// ---- backward pass 3: max-pool backward as a gather ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ gp, const int8_t* __restrict__ idx,
                                                       float* __restrict__ gx, int64_t N, int H, int W, int C, int k, int s, int Ho,
                                                       int Wo) {
    const int64_t total = N * H * W * C;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const int64_t pix = t / C;
        const int w = (int)(pix % W);
        const int64_t nh = pix / W;
        const int h = (int)(nh % H);
        const int64_t n = nh / H;
        // windows (ho, wo) that contain (h, w): ho in [ceil((h - k + 1) / s), floor(h / s)]
        const int ho0 = h - k + 1 > 0 ? (h - k + 1 + s - 1) / s : 0, ho1 = min(h / s, Ho - 1);
        const int wo0 = w - k + 1 > 0 ? (w - k + 1 + s - 1) / s : 0, wo1 = min(w / s, Wo - 1);
        float acc = 0.0f;
        for (int ho = ho0; ho <= ho1; ++ho)
            for (int wo = wo0; wo <= wo1; ++wo) {
                const int64_t r = ((n * Ho + ho) * Wo + wo) * C + c;
                if ((int)idx[r] == (h - ho * s) * k + (w - wo * s)) acc += gp[r];
            }
        gx[t] = acc;
    }
}

int rows_per_block_for(int64_t R, int* nblk) {
    int64_t rpb = (R + 1023) / 1024;                        // ~1024 workgroups
    if (rpb < TC_ROWS_PER_BLOCK_MIN) rpb = TC_ROWS_PER_BLOCK_MIN;
    *nblk = (int)((R + rpb - 1) / rpb);
    return (int)rpb;
}

}  // namespace

extern "C" int64_t qt_train_chain_partial_floats(int64_t R, int64_t C) {
    int nblk;
    rows_per_block_for(R > 0 ? R : 1, &nblk);
    return 2 * (int64_t)nblk * C;
}

extern "C" int qt_pool_bn_sign_train_f32(const float* x, int64_t N, int64_t H, int64_t W, int64_t C, int64_t k, int64_t s,
                                         const float* gamma, const float* beta, float eps, float momentum, float ht_lo,
                                         float ht_hi, float* running_mean, float* running_var, float* p, int8_t* idx,
                                         float* mean, float* invstd, float* partial, float* sgn, qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || C <= 0 || k < 1 || s < 1 || k > 11 || H < k || W < k) return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!x || !mean || !invstd || !partial || !sgn || (k > 1 && (!p || !idx))) return QT_ERR_INVALID_ARG;
    const int Ho = (int)((H - k) / s + 1), Wo = (int)((W - k) / s + 1);
    const int64_t R = N * Ho * Wo;
    if (R * C >= (1ll << 40) || C > (1 << 24)) return QT_ERR_UNSUPPORTED;
    int nblk;
    const int rpb = rows_per_block_for(R, &nblk);
    hipStream_t st = (hipStream_t)stream;
    const float* pp = k > 1 ? p : x;
    hipLaunchKernelGGL(pool_sum_kernel, dim3(nblk), dim3(256), 0, st, x, p, idx, partial, R, (int)H, (int)W, (int)C, (int)k, (int)s,
                       Ho, Wo, rpb);
    const int cg = (int)((C + 255) / 256);
    hipLaunchKernelGGL(fold_kernel, dim3(cg), dim3(256), 0, st, partial, nblk, (int)C, 1.0 / (double)R, mean);
    hipLaunchKernelGGL(sqdev_kernel, dim3(nblk), dim3(256), 0, st, pp, mean, partial, R, (int)C, rpb);
    hipLaunchKernelGGL(finalize_kernel, dim3(cg), dim3(256), 0, st, partial, nblk, (int)C, (double)R, eps, momentum, mean, invstd,
                       running_mean, running_var);
    const int64_t total = R * C;
    hipLaunchKernelGGL(norm_sign_kernel, dim3(qt_stream_grid((total + 255) / 256)), dim3(256), 0, st, pp, mean, invstd, gamma, beta,
                       ht_lo, ht_hi, sgn, total, (int)C);
    return qt_check_launch();
}

extern "C" int qt_pool_bn_sign_train_backward_f32(const float* g, const float* p_or_x, const int8_t* idx, int64_t N, int64_t H,
                                                  int64_t W, int64_t C, int64_t k, int64_t s, const float* gamma,
                                                  const float* beta, const float* mean, const float* invstd, float ht_lo,
                                                  float ht_hi, float ste_threshold, float* partial, float* dgamma, float* dbeta,
                                                  float* gp, float* gx, qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || C <= 0 || k < 1 || s < 1 || k > 11 || H < k || W < k) return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!g || !p_or_x || !mean || !invstd || !partial || !dgamma || !dbeta || !gp || (k > 1 && (!idx || !gx))) return QT_ERR_INVALID_ARG;
    const int Ho = (int)((H - k) / s + 1), Wo = (int)((W - k) / s + 1);
    const int64_t R = N * Ho * Wo;
    int nblk;
    const int rpb = rows_per_block_for(R, &nblk);
    hipStream_t st = (hipStream_t)stream;
    float* part_b = partial;
    float* part_g = partial + (int64_t)nblk * C;
    hipLaunchKernelGGL(bwd_sum_kernel, dim3(nblk), dim3(256), 0, st, p_or_x, g, mean, invstd, gamma, beta, ht_lo, ht_hi, ste_threshold,
                       part_b, part_g, R, (int)C, rpb);
    const int cg = (int)((C + 255) / 256);
    hipLaunchKernelGGL(fold_kernel, dim3(cg), dim3(256), 0, st, part_b, nblk, (int)C, 1.0, dbeta);
    hipLaunchKernelGGL(fold_kernel, dim3(cg), dim3(256), 0, st, part_g, nblk, (int)C, 1.0, dgamma);
    const int64_t total = R * C;
    hipLaunchKernelGGL(bwd_dx_kernel, dim3(qt_stream_grid((total + 255) / 256)), dim3(256), 0, st, p_or_x, g, mean, invstd, gamma, beta,
                       dgamma, dbeta, ht_lo, ht_hi, ste_threshold, (float)(1.0 / (double)R), gp, total, (int)C);
    if (k > 1) {
        const int64_t tx = N * H * W * C;
        hipLaunchKernelGGL(pool_bwd_kernel, dim3(qt_stream_grid((tx + 255) / 256)), dim3(256), 0, st, gp, idx, gx, N, (int)H, (int)W,
                           (int)C, (int)k, (int)s, Ho, Wo);
    }
    return qt_check_launch();
}
