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
// Per-channel sums: a workgroup owns a slab of rows and all channels (64 x 16 threads: channel quads x rows in flight, 16-byte
// accesses coalesced along c), writes one partial per channel; a second tiny launch folds the partials in double.  Everything is
// elementwise or a column reduction: algorithmic bytes = what each pass reads + writes.  C % 4 == 0.
#include "qt_common.h"

namespace {

// Thread layout of every kernel: blockDim = (64, 16).  x <-> a channel quad (float4) of a 256-channel chunk, y <-> a row
// sub-index; a workgroup owns a slab of rows, walks it sixteen rows at a time and the channels in chunks of 256.  No index
// division per element, 16-byte accesses, coalesced along the channels.  C % 4 == 0.
constexpr int TC_TX = 64, TC_TY = 16;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4_or(const float* p, float v) { return p ? ld4(p) : make_float4(v, v, v, v); }

// sum the TC_TY row-partials of a channel quad through LDS and let y == 0 write it
__device__ __forceinline__ void fold_rows_store(float4 acc, float* dst, float4 (*sh)[TC_TX], bool live) {
    sh[threadIdx.y][threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.y == 0 && live) {
        float4 a = sh[0][threadIdx.x];
#pragma unroll
        for (int y = 1; y < TC_TY; ++y) {
            const float4 b = sh[y][threadIdx.x];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        *reinterpret_cast<float4*>(dst) = a;
    }
    __syncthreads();
}

// ---- forward pass 1: pooling (+ argmax) and per-block channel sums of p ------------------------------------------------------
__global__ __launch_bounds__(1024) void pool_sum_kernel(const float* __restrict__ x, float* __restrict__ p, int8_t* __restrict__ idx,
                                                       float* __restrict__ part, int64_t R, int H, int W, int C, int k, int s,
                                                       int Ho, int Wo, int rows_per_block) {
    __shared__ float4 sh[TC_TY][TC_TX];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    for (int c0 = 0; c0 < C; c0 += 4 * TC_TX) {
        const int c = c0 + 4 * threadIdx.x;
        const bool live = c < C;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live)
            for (int64_t r = r0 + threadIdx.y; r < r1; r += TC_TY) {
                float4 m;
                if (k == 1) {
                    m = ld4(x + r * C + c);
                } else {
                    const int64_t n = r / ((int64_t)Ho * Wo);
                    const int rem = (int)(r - n * Ho * Wo);
                    const int ho = rem / Wo, wo = rem - ho * Wo;
                    const float* base = x + ((n * H + (int64_t)ho * s) * W + (int64_t)wo * s) * C + c;
                    m = ld4(base);
                    int bx = 0, by = 0, bz = 0, bw = 0;
                    for (int i = 0; i < k; ++i)
                        for (int j = 0; j < k; ++j) {
                            if (i == 0 && j == 0) continue;
                            const float4 v = ld4(base + ((int64_t)i * W + j) * C);
                            const int t = i * k + j;
                            if (v.x > m.x || v.x != v.x) { m.x = v.x; bx = t; }
                            if (v.y > m.y || v.y != v.y) { m.y = v.y; by = t; }
                            if (v.z > m.z || v.z != v.z) { m.z = v.z; bz = t; }
                            if (v.w > m.w || v.w != v.w) { m.w = v.w; bw = t; }
                        }
                    *reinterpret_cast<float4*>(p + r * C + c) = m;
                    *reinterpret_cast<uint32_t*>(idx + r * C + c) = (uint32_t)bx | ((uint32_t)by << 8) | ((uint32_t)bz << 16) | ((uint32_t)bw << 24);
                }
                acc.x += m.x; acc.y += m.y; acc.z += m.z; acc.w += m.w;
            }
        fold_rows_store(acc, part + (int64_t)blockIdx.x * C + (live ? c : 0), sh, live);
    }
}

// ---- forward pass 2: per-block channel sums of (p - mean)^2 ---------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sqdev_kernel(const float* __restrict__ p, const float* __restrict__ mean,
                                                    float* __restrict__ part, int64_t R, int C, int rows_per_block) {
    __shared__ float4 sh[TC_TY][TC_TX];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    for (int c0 = 0; c0 < C; c0 += 4 * TC_TX) {
        const int c = c0 + 4 * threadIdx.x;
        const bool live = c < C;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            const float4 mu = ld4(mean + c);
            for (int64_t r = r0 + threadIdx.y; r < r1; r += TC_TY) {
                const float4 v = ld4(p + r * C + c);
                const float dx = v.x - mu.x, dy = v.y - mu.y, dz = v.z - mu.z, dw = v.w - mu.w;
                acc.x += dx * dx; acc.y += dy * dy; acc.z += dz * dz; acc.w += dw * dw;
            }
        }
        fold_rows_store(acc, part + (int64_t)blockIdx.x * C + (live ? c : 0), sh, live);
    }
}

// fold partials [nblk][C] in double.  blockDim = (64, 16): x <-> channel, y <-> every 16th partial; the 16 row sums of a channel
// meet in LDS.  (One thread per channel walking all 2048 partials alone was 400 us per call: a chain of dependent-latency loads
// on a single workgroup.)
constexpr int TC_FY = 16;
__device__ __forceinline__ double fold_partials(const float* __restrict__ part, int nblk, int C, int c, double (*sh)[64]) {
    double s = 0.0;
    if (c < C) {
        // eight loads in flight per thread (same summation order): rolled, the loop was one L2 round trip per partial — 9.5 us for
        // 512 partials (profiles/r6_train_steps.md)
#pragma unroll 8
        for (int b = threadIdx.y; b < nblk; b += TC_FY) s += (double)part[(int64_t)b * C + c];
    }
    sh[threadIdx.y][threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.y == 0) {
        s = 0.0;
#pragma unroll
        for (int y = 0; y < TC_FY; ++y) s += sh[y][threadIdx.x];
    }
    return s;        // valid for threadIdx.y == 0
}

__global__ __launch_bounds__(1024) void fold_kernel(const float* __restrict__ part, int nblk, int C, double scale,
                                                    float* __restrict__ out) {
    __shared__ double sh[TC_FY][64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    const double s = fold_partials(part, nblk, C, c, sh);
    if (threadIdx.y == 0 && c < C) out[c] = (float)(s * scale);
}

// two folds in one launch (dbeta and dgamma of a backward): blockIdx.y picks the pair
__global__ __launch_bounds__(1024) void fold2_kernel(const float* __restrict__ part_a, const float* __restrict__ part_b, int nblk, int C,
                                                     float* __restrict__ out_a, float* __restrict__ out_b) {
    __shared__ double sh[TC_FY][64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    const double s = fold_partials(blockIdx.y ? part_b : part_a, nblk, C, c, sh);
    if (threadIdx.y == 0 && c < C) (blockIdx.y ? out_b : out_a)[c] = (float)s;
}

// var partials -> invstd, running statistics
__global__ __launch_bounds__(1024) void finalize_kernel(const float* __restrict__ part, int nblk, int C, double R, float eps,
                                                        float momentum, const float* __restrict__ mean, float* __restrict__ invstd,
                                                        float* __restrict__ running_mean, float* __restrict__ running_var,
                                                        int32_t* __restrict__ zero_flag) {
    __shared__ double sh[TC_FY][64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    // (the chain's int8 range flag starts at zero here: the quantiser pass behind this launch ORs into it — no fill launch)
    if (zero_flag && blockIdx.x == 0 && threadIdx.x == 0 && threadIdx.y == 0) *zero_flag = 0;
    const double s = fold_partials(part, nblk, C, c, sh);
    if (threadIdx.y != 0 || c >= C) return;
    const float var = (float)(s / R);
    invstd[c] = 1.0f / sqrtf(var + eps);
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mean[c];
    if (running_var) {
        const float unbiased = R > 1.0 ? (float)(s / (R - 1.0)) : var;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
    }
}

__device__ __forceinline__ float tc_y(float p, float mean, float invstd, float gamma, float beta) {
    return (p - mean) * invstd * gamma + beta;
}
__device__ __forceinline__ float tc_clamp(float y, float lo, float hi) { return y < lo ? lo : (y > hi ? hi : y); }

// ---- forward pass 3: normalise, clamp, sign ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void norm_sign_kernel(const float* __restrict__ p, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float lo, float hi,
                                                        float* __restrict__ sgn, int64_t R, int C, int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    for (int c = 4 * threadIdx.x; c < C; c += 4 * TC_TX) {
        const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4_or(gamma ? gamma + c : nullptr, 1.0f),
                     be = ld4_or(beta ? beta + c : nullptr, 0.0f);
        for (int64_t r = r0 + threadIdx.y; r < r1; r += TC_TY) {
            const float4 v = ld4(p + r * C + c);
            float4 o;
            o.x = qt_safe_sign(tc_clamp(tc_y(v.x, mu.x, is.x, ga.x, be.x), lo, hi));
            o.y = qt_safe_sign(tc_clamp(tc_y(v.y, mu.y, is.y, ga.y, be.y), lo, hi));
            o.z = qt_safe_sign(tc_clamp(tc_y(v.z, mu.z, is.z, ga.z, be.z), lo, hi));
            o.w = qt_safe_sign(tc_clamp(tc_y(v.w, mu.w, is.w, ga.w, be.w), lo, hi));
            *reinterpret_cast<float4*>(sgn + r * C + c) = o;
        }
    }
}

// ---- backward pass 1: per-block channel sums of g_y and g_y * xhat -----------------------------------------------------------------
__device__ __forceinline__ float tc_gy(float g, float y, float lo, float hi, float ste) {
    const float h = tc_clamp(y, lo, hi);
    float gy = fabsf(h) <= ste ? g : 0.0f;                // STE of the sign
    if (!(y > lo && y < hi)) gy = 0.0f;                   // Hardtanh (strict, as torch's hardtanh_backward)
    return gy;
}

__global__ __launch_bounds__(1024) void bwd_sum_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                      const float* __restrict__ mean, const float* __restrict__ invstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float lo,
                                                      float hi, float ste, float* __restrict__ part_b, float* __restrict__ part_g,
                                                      int64_t R, int C, int rows_per_block) {
    __shared__ float4 sh[TC_TY][TC_TX];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    for (int c0 = 0; c0 < C; c0 += 4 * TC_TX) {
        const int c = c0 + 4 * threadIdx.x;
        const bool live = c < C;
        float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;
        if (live) {
            const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4_or(gamma ? gamma + c : nullptr, 1.0f),
                         be = ld4_or(beta ? beta + c : nullptr, 0.0f);
            for (int64_t r = r0 + threadIdx.y; r < r1; r += TC_TY) {
                const float4 v = ld4(p + r * C + c), gv = ld4(g + r * C + c);
                const float hx = (v.x - mu.x) * is.x, hy = (v.y - mu.y) * is.y, hz = (v.z - mu.z) * is.z, hw = (v.w - mu.w) * is.w;
                const float gx_ = tc_gy(gv.x, hx * ga.x + be.x, lo, hi, ste), gy_ = tc_gy(gv.y, hy * ga.y + be.y, lo, hi, ste),
                            gz_ = tc_gy(gv.z, hz * ga.z + be.z, lo, hi, ste), gw_ = tc_gy(gv.w, hw * ga.w + be.w, lo, hi, ste);
                sb.x += gx_; sb.y += gy_; sb.z += gz_; sb.w += gw_;
                sg.x += gx_ * hx; sg.y += gy_ * hy; sg.z += gz_ * hz; sg.w += gw_ * hw;
            }
        }
        fold_rows_store(sb, part_b + (int64_t)blockIdx.x * C + (live ? c : 0), sh, live);
        fold_rows_store(sg, part_g + (int64_t)blockIdx.x * C + (live ? c : 0), sh, live);
    }
}

// ---- backward pass 2: g_p ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bwd_dx_kernel(const float* __restrict__ p, const float* __restrict__ g,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ dgamma, const float* __restrict__ dbeta, float lo,
                                                     float hi, float ste, float invR, float* __restrict__ gp, int64_t R, int C,
                                                     int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    for (int c = 4 * threadIdx.x; c < C; c += 4 * TC_TX) {
        const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4_or(gamma ? gamma + c : nullptr, 1.0f),
                     be = ld4_or(beta ? beta + c : nullptr, 0.0f), dg = ld4(dgamma + c), db = ld4(dbeta + c);
        for (int64_t r = r0 + threadIdx.y; r < r1; r += TC_TY) {
            const float4 v = ld4(p + r * C + c), gv = ld4(g + r * C + c);
            const float hx = (v.x - mu.x) * is.x, hy = (v.y - mu.y) * is.y, hz = (v.z - mu.z) * is.z, hw = (v.w - mu.w) * is.w;
            float4 o;
            o.x = ga.x * is.x * (tc_gy(gv.x, hx * ga.x + be.x, lo, hi, ste) - db.x * invR - hx * dg.x * invR);
            o.y = ga.y * is.y * (tc_gy(gv.y, hy * ga.y + be.y, lo, hi, ste) - db.y * invR - hy * dg.y * invR);
            o.z = ga.z * is.z * (tc_gy(gv.z, hz * ga.z + be.z, lo, hi, ste) - db.z * invR - hz * dg.z * invR);
            o.w = ga.w * is.w * (tc_gy(gv.w, hw * ga.w + be.w, lo, hi, ste) - db.w * invR - hw * dg.w * invR);
            *reinterpret_cast<float4*>(gp + r * C + c) = o;
        }
    }
}

// ---- backward pass 3: max-pool backward as a gather over the windows that contain each input pixel -------------------------------
__global__ __launch_bounds__(1024) void pool_bwd_kernel(const float* __restrict__ gp, const int8_t* __restrict__ idx,
                                                       float* __restrict__ gx, int64_t NHW, int H, int W, int C, int k, int s,
                                                       int Ho, int Wo, int rows_per_block) {
    const int64_t q0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t q1 = min(NHW, q0 + rows_per_block);
    for (int64_t q = q0 + threadIdx.y; q < q1; q += TC_TY) {
        const int w = (int)(q % W);
        const int64_t nh = q / W;
        const int h = (int)(nh % H);
        const int64_t n = nh / H;
        const int ho0 = h - k + 1 > 0 ? (h - k + s) / s : 0, ho1 = min(h / s, Ho - 1);
        const int wo0 = w - k + 1 > 0 ? (w - k + s) / s : 0, wo1 = min(w / s, Wo - 1);
        for (int c = 4 * threadIdx.x; c < C; c += 4 * TC_TX) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int ho = ho0; ho <= ho1; ++ho)
                for (int wo = wo0; wo <= wo1; ++wo) {
                    const int64_t r = ((n * Ho + ho) * Wo + wo) * C + c;
                    const uint32_t id = *reinterpret_cast<const uint32_t*>(idx + r);
                    const uint32_t me = (uint32_t)((h - ho * s) * k + (w - wo * s));
                    const float4 v = ld4(gp + r);
                    if ((id & 0xffu) == me) acc.x += v.x;
                    if (((id >> 8) & 0xffu) == me) acc.y += v.y;
                    if (((id >> 16) & 0xffu) == me) acc.z += v.z;
                    if ((id >> 24) == me) acc.w += v.w;
                }
            *reinterpret_cast<float4*>(gx + q * C + c) = acc;
        }
    }
}

// ---- DoReFa chain: BatchNorm (batch statistics) [+ residual] [-> ReLU] -> nnDorefaQuant (identity STE) ------------------------------
// (models/Resnet/Resnet_bin.py:63-97: every block of the reference's DoReFa ResNets.)  Forward = the statistics launches above
// + the code epilogue pass (csrc/codes_i8.hip: qt_affine_dorefa_codes_i8 with bn_stats = [mean | invstd]); backward of
//     t = fma((x - mean) * invstd, gamma, beta) [+ res] ;  out = quant(relu(t))          (the forward pass's own expression)
//     g_t = g * 1[t > 0] (ReLU; the quantiser's STE is the identity) ;  g_res = g_t ;  BatchNorm backward on g_t as above.
__device__ __forceinline__ float act_gt(float g, float v, float mu, float is, float ga, float be, float res, int relu) {
    const float t = __fadd_rn(__fmaf_rn(__fmul_rn(__fsub_rn(v, mu), is), ga, be), res);
    return (relu && !(t > 0.0f)) ? 0.0f : g;
}

__global__ __launch_bounds__(1024) void act_bwd_sum_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                          const float* __restrict__ res, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int relu, float* __restrict__ part_b,
                                                          float* __restrict__ part_g, int64_t R, int C, int rows_per_block) {
    __shared__ float4 sh[TC_TY][TC_TX];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    for (int c0 = 0; c0 < C; c0 += 4 * TC_TX) {
        const int c = c0 + 4 * threadIdx.x;
        const bool live = c < C;
        float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;
        if (live) {
            const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4_or(gamma ? gamma + c : nullptr, 1.0f),
                         be = ld4_or(beta ? beta + c : nullptr, 0.0f);
            for (int64_t r = r0 + threadIdx.y; r < r1; r += TC_TY) {
                const float4 v = ld4(x + r * C + c), gv = ld4(g + r * C + c), rv = ld4_or(res ? res + r * C + c : nullptr, 0.0f);
                const float gx_ = act_gt(gv.x, v.x, mu.x, is.x, ga.x, be.x, rv.x, relu), gy_ = act_gt(gv.y, v.y, mu.y, is.y, ga.y, be.y, rv.y, relu),
                            gz_ = act_gt(gv.z, v.z, mu.z, is.z, ga.z, be.z, rv.z, relu), gw_ = act_gt(gv.w, v.w, mu.w, is.w, ga.w, be.w, rv.w, relu);
                sb.x += gx_; sb.y += gy_; sb.z += gz_; sb.w += gw_;
                sg.x += gx_ * ((v.x - mu.x) * is.x); sg.y += gy_ * ((v.y - mu.y) * is.y);
                sg.z += gz_ * ((v.z - mu.z) * is.z); sg.w += gw_ * ((v.w - mu.w) * is.w);
            }
        }
        fold_rows_store(sb, part_b + (int64_t)blockIdx.x * C + (live ? c : 0), sh, live);
        fold_rows_store(sg, part_g + (int64_t)blockIdx.x * C + (live ? c : 0), sh, live);
    }
}

__global__ __launch_bounds__(1024) void act_bwd_dx_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         const float* __restrict__ res, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ dgamma,
                                                         const float* __restrict__ dbeta, int relu, float invR,
                                                         float* __restrict__ gx, float* __restrict__ gres, int64_t R, int C,
                                                         int rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    for (int c = 4 * threadIdx.x; c < C; c += 4 * TC_TX) {
        const float4 mu = ld4(mean + c), is = ld4(invstd + c), ga = ld4_or(gamma ? gamma + c : nullptr, 1.0f),
                     be = ld4_or(beta ? beta + c : nullptr, 0.0f), dg = ld4(dgamma + c), db = ld4(dbeta + c);
        for (int64_t r = r0 + threadIdx.y; r < r1; r += TC_TY) {
            const float4 v = ld4(x + r * C + c), gv = ld4(g + r * C + c), rv = ld4_or(res ? res + r * C + c : nullptr, 0.0f);
            float4 gt;
            gt.x = act_gt(gv.x, v.x, mu.x, is.x, ga.x, be.x, rv.x, relu); gt.y = act_gt(gv.y, v.y, mu.y, is.y, ga.y, be.y, rv.y, relu);
            gt.z = act_gt(gv.z, v.z, mu.z, is.z, ga.z, be.z, rv.z, relu); gt.w = act_gt(gv.w, v.w, mu.w, is.w, ga.w, be.w, rv.w, relu);
            if (gres) *reinterpret_cast<float4*>(gres + r * C + c) = gt;
            float4 o;
            o.x = ga.x * is.x * (gt.x - db.x * invR - (v.x - mu.x) * is.x * dg.x * invR);
            o.y = ga.y * is.y * (gt.y - db.y * invR - (v.y - mu.y) * is.y * dg.y * invR);
            o.z = ga.z * is.z * (gt.z - db.z * invR - (v.z - mu.z) * is.z * dg.z * invR);
            o.w = ga.w * is.w * (gt.w - db.w * invR - (v.w - mu.w) * is.w * dg.w * invR);
            *reinterpret_cast<float4*>(gx + r * C + c) = o;
        }
    }
}

int rows_per_block_for(int64_t R, int* nblk) {
    int64_t rpb = (R + 511) / 512;                          // ~512 workgroups of 1024 threads: few partials to fold
    if (rpb < TC_TY) rpb = TC_TY;
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
    if (C & 3) return QT_ERR_ALIGNMENT;                       // float4 channel quads
    int nblk;
    const int rpb = rows_per_block_for(R, &nblk);
    hipStream_t st = (hipStream_t)stream;
    const float* pp = k > 1 ? p : x;
    hipLaunchKernelGGL(pool_sum_kernel, dim3(nblk), dim3(TC_TX, TC_TY), 0, st, x, p, idx, partial, R, (int)H, (int)W, (int)C, (int)k, (int)s,
                       Ho, Wo, rpb);
    const int cg = (int)((C + 63) / 64);
    hipLaunchKernelGGL(fold_kernel, dim3(cg), dim3(64, TC_FY), 0, st, partial, nblk, (int)C, 1.0 / (double)R, mean);
    hipLaunchKernelGGL(sqdev_kernel, dim3(nblk), dim3(TC_TX, TC_TY), 0, st, pp, mean, partial, R, (int)C, rpb);
    hipLaunchKernelGGL(finalize_kernel, dim3(cg), dim3(64, TC_FY), 0, st, partial, nblk, (int)C, (double)R, eps, momentum, mean, invstd,
                       running_mean, running_var, (int32_t*)nullptr);
    hipLaunchKernelGGL(norm_sign_kernel, dim3(nblk), dim3(TC_TX, TC_TY), 0, st, pp, mean, invstd, gamma, beta, ht_lo, ht_hi, sgn, R,
                       (int)C, rpb);
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
    if (C & 3) return QT_ERR_ALIGNMENT;
    int nblk;
    const int rpb = rows_per_block_for(R, &nblk);
    hipStream_t st = (hipStream_t)stream;
    float* part_b = partial;
    float* part_g = partial + (int64_t)nblk * C;
    hipLaunchKernelGGL(bwd_sum_kernel, dim3(nblk), dim3(TC_TX, TC_TY), 0, st, p_or_x, g, mean, invstd, gamma, beta, ht_lo, ht_hi, ste_threshold,
                       part_b, part_g, R, (int)C, rpb);
    const int cg = (int)((C + 63) / 64);
    hipLaunchKernelGGL(fold2_kernel, dim3(cg, 2), dim3(64, TC_FY), 0, st, part_b, part_g, nblk, (int)C, dbeta, dgamma);
    hipLaunchKernelGGL(bwd_dx_kernel, dim3(nblk), dim3(TC_TX, TC_TY), 0, st, p_or_x, g, mean, invstd, gamma, beta, dgamma, dbeta, ht_lo,
                       ht_hi, ste_threshold, (float)(1.0 / (double)R), gp, R, (int)C, rpb);
    if (k > 1) {
        int nblk_x;
        const int rpb_x = rows_per_block_for(N * H * W, &nblk_x);
        hipLaunchKernelGGL(pool_bwd_kernel, dim3(nblk_x), dim3(TC_TX, TC_TY), 0, st, gp, idx, gx, N * H * W, (int)H, (int)W, (int)C,
                           (int)k, (int)s, Ho, Wo, rpb_x);
    }
    return qt_check_launch();
}

// Batch statistics of an fp32 [R][C] matrix (channels-last pixels x channels): mean, invstd = 1 / sqrt(var + eps) (biased variance,
// two passes, partials folded in double) and the running-statistics update of nn.BatchNorm2d.train().  stats2 = [mean | invstd]
// (the layout the code epilogue's bn_stats takes).
extern "C" int qt_bn_train_stats_f32(const float* x, int64_t R, int64_t C, float eps, float momentum, float* running_mean,
                                     float* running_var, float* stats2, float* partial, int32_t* zero_flag, qt_stream_t stream) {
    if (R <= 0 || C <= 0 || !x || !stats2 || !partial) return QT_ERR_INVALID_ARG;
    if (R * C >= (1ll << 40) || C > (1 << 24)) return QT_ERR_UNSUPPORTED;
    if ((C & 3) || !qt_aligned16(x)) return QT_ERR_ALIGNMENT;
    int nblk;
    const int rpb = rows_per_block_for(R, &nblk);
    hipStream_t st = (hipStream_t)stream;
    float* mean = stats2;
    float* invstd = stats2 + C;
    hipLaunchKernelGGL(pool_sum_kernel, dim3(nblk), dim3(TC_TX, TC_TY), 0, st, x, (float*)nullptr, (int8_t*)nullptr, partial, R, 1, 1,
                       (int)C, 1, 1, 1, 1, rpb);
    const int cg = (int)((C + 63) / 64);
    hipLaunchKernelGGL(fold_kernel, dim3(cg), dim3(64, TC_FY), 0, st, partial, nblk, (int)C, 1.0 / (double)R, mean);
    hipLaunchKernelGGL(sqdev_kernel, dim3(nblk), dim3(TC_TX, TC_TY), 0, st, x, mean, partial, R, (int)C, rpb);
    hipLaunchKernelGGL(finalize_kernel, dim3(cg), dim3(64, TC_FY), 0, st, partial, nblk, (int)C, (double)R, eps, momentum, mean, invstd,
                       running_mean, running_var, zero_flag);
    return qt_check_launch();
}

// Backward of out = quant(relu?(BatchNorm_train(x) [+ res])) w.r.t. x, gamma, beta and res (see act_gt above): gx[R][C], dgamma[C],
// dbeta[C] and, when gres != NULL, gres[R][C] = g * 1[t > 0] (the gradient that flows on into the shortcut branch).
extern "C" int qt_bn_act_train_backward_f32(const float* g, const float* x, const float* res, int64_t R, int64_t C, const float* gamma,
                                            const float* beta, const float* stats2, int relu, float* partial, float* dgamma,
                                            float* dbeta, float* gx, float* gres, qt_stream_t stream) {
    if (R <= 0 || C <= 0 || !g || !x || !stats2 || !partial || !dgamma || !dbeta || !gx) return QT_ERR_INVALID_ARG;
    if (C & 3) return QT_ERR_ALIGNMENT;
    int nblk;
    const int rpb = rows_per_block_for(R, &nblk);
    hipStream_t st = (hipStream_t)stream;
    const float* mean = stats2;
    const float* invstd = stats2 + C;
    float* part_b = partial;
    float* part_g = partial + (int64_t)nblk * C;
    hipLaunchKernelGGL(act_bwd_sum_kernel, dim3(nblk), dim3(TC_TX, TC_TY), 0, st, x, g, res, mean, invstd, gamma, beta, relu, part_b, part_g,
                       R, (int)C, rpb);
    const int cg = (int)((C + 63) / 64);
    hipLaunchKernelGGL(fold2_kernel, dim3(cg, 2), dim3(64, TC_FY), 0, st, part_b, part_g, nblk, (int)C, dbeta, dgamma);
    hipLaunchKernelGGL(act_bwd_dx_kernel, dim3(nblk), dim3(TC_TX, TC_TY), 0, st, x, g, res, mean, invstd, gamma, beta, dgamma, dbeta, relu,
                       (float)(1.0 / (double)R), gx, gres, R, (int)C, rpb);
    return qt_check_launch();
}
