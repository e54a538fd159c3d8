// ONE column-sum arithmetic for the per-tap alpha of an XNORConv2d weight (alpha[c] = mean_r |W[r, c]| over a row-major,
// contiguous [R = Cout * Cin, C = kh * kw <= 1024] view).  Shared by qt_xnor_tap_prep_f32 (conv_taps.hip: TapScales of the
// +-1 routes, forward and backward) and qt_xnor_weight_f32 (xnor_weight.hip: the sign(W) * alpha image of .eval() and of the
// real-input route), so every execution of a layer sees the SAME alpha bits whatever route it takes (three
// different summation orders).  Order: tap_alpha_blocks() row blocks; inside a block thread t owns column t % C and rows
// t / C, t / C + rpp, ...; the rpp partials of a column are added in ascending order, then the blocks in ascending order,
// then ONE division by R.
#pragma once
#include "qt_common.h"
#include <algorithm>

namespace {

__global__ __launch_bounds__(1024) void tap_abs_partial_kernel(const float* __restrict__ w, int64_t R, int C, int64_t rows_per_blk,
                                                               float* __restrict__ work) {
    __shared__ float sm[1024];
    const int t = threadIdx.x, rpp = 1024 / C, c = t % C, r0 = t / C;
    float acc = 0.0f;
    if (r0 < rpp) {
        const int64_t rb = (int64_t)blockIdx.x * rows_per_blk, re = min(R, rb + rows_per_blk);
        for (int64_t r = rb + r0; r < re; r += rpp) acc += fabsf(w[r * C + c]);
    }
    sm[t] = acc;
    __syncthreads();
    if (t < C) {
        float s = 0.0f;
        for (int j = 0; j < rpp; ++j) s += sm[j * C + t];
        work[(int64_t)blockIdx.x * C + t] = s;
    }
}

// the closing step of the order above: blocks in ascending order, one division.  Called by ALL threads of a 1024-thread block
// (thread t < T returns column t's alpha, the others 0): the nblk x T partials are staged through LDS in chunks of whole block
// rows — every thread's loads independent and coalesced — and column t is then added up from LDS in ascending block order, the
// order (and so the bits) of the plain loop  for b: s += work[b * T + t]  this replaces: that loop was a chain of nblk dependent
// global loads, 67 us for the 512 blocks of AlexNet's conv2 weight (tools/probes/qi_kt.sh), every forward of a training step.
constexpr int TAP_FINAL_STAGE = 8192;     // floats of LDS staging (32 KiB)
__device__ __forceinline__ float tap_alpha_final(const float* __restrict__ work, int nblk, int T, int t, float rows, float* stage) {
    float s = 0.0f;
    const int per = TAP_FINAL_STAGE / T > 0 ? TAP_FINAL_STAGE / T : 1;      // block rows per chunk (T <= 1024 <= TAP_FINAL_STAGE)
    for (int b0 = 0; b0 < nblk; b0 += per) {
        const int nb = min(per, nblk - b0), n = nb * T;
        for (int i = t; i < n; i += 1024) stage[i] = work[(int64_t)b0 * T + i];
        __syncthreads();
        if (t < T)
            for (int b = 0; b < nb; ++b) s += stage[b * T + t];
        __syncthreads();
    }
    return t < T ? s / rows : 0.0f;
}

// number of row blocks (= workspace rows of C floats) and the rows per block of the order above
inline int tap_alpha_blocks(int64_t R, int64_t taps, int64_t* rows_per_blk) {
    const int64_t rpp = 1024 / taps;
    const int nblk = (int)std::min<int64_t>(512, (R + rpp * 8 - 1) / (rpp * 8));
    if (rows_per_blk) *rows_per_blk = ((R + nblk - 1) / nblk + rpp - 1) / rpp * rpp;
    return nblk;
}

}  // namespace
