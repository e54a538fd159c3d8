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

// the closing step of the order above: blocks in ascending order, one division
__device__ __forceinline__ float tap_alpha_final(const float* __restrict__ work, int nblk, int T, int t, float rows) {
    float s = 0.0f;
    for (int b = 0; b < nblk; ++b) s += work[(int64_t)b * T + t];
    return s / rows;
}

// number of row blocks (= workspace rows of C floats) and the rows per block of the order above
inline int tap_alpha_blocks(int64_t R, int64_t taps, int64_t* rows_per_blk) {
    const int64_t rpp = 1024 / taps;
    const int nblk = (int)std::min<int64_t>(512, (R + rpp * 8 - 1) / (rpp * 8));
    if (rows_per_blk) *rows_per_blk = ((R + nblk - 1) / nblk + rpp - 1) / rpp * rpp;
    return nblk;
}

}  // namespace
