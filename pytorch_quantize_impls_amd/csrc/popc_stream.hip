// Streaming XNOR / ternary popcount GEMM for the regime where ONE operand has only a handful of rows: batch <= 32 against a
// whole weight matrix (small-batch serving: the packed weight planes are read once, N*K/8 bytes, and nothing else matters), or a
// classifier head with <= 32 output features against a batch.  The lane <-> batch-row mapping of popc_skinny.hip leaves 63 of 64
// lanes idle at batch 1 and runs a classifier head on ceil(M/64) workgroups (256 x 10 x 4096: 4 workgroups, 15-22 us of scalar-load
// latency); here the K dimension runs ALONG THE LANES instead:
//
//   lane  <-> two consecutive k-words (one 8-byte load; a wave-load covers 512 contiguous bytes of a packed row), K is walked in
//             steps of 64 lanes x 64 bits;
//   "streamed" operand (many rows): every wave owns RS consecutive rows — coalesced HBM loads of packed words, each word fetched once;
//   "few" operand (<= F rows): re-read by every wave at the same k positions (a few KB: L1 / L2 hits);
//   accumulate: RS x F per-lane popcounts (v_xor + v_bcnt accumulate), then a wavefront reduction of each by DPP (quad_perm,
//             row_half_mirror, row_mirror: 16-lane row sums on every lane) + four v_readlane / scalar adds across the rows;
//   store : lane j writes result j of the wave (one predicated store).
//
// SWAP = false: streamed = weights (N rows), few = activations (M <= F rows).   SWAP = true: streamed = activations (M rows), few =
// weights (N <= F rows).  Ternary weights carry (mask, sign): sum = popc(mask) - 2 popc((x ^ sign) & mask).
// Planes are zero-padded to their leading dimension (a multiple of 4 words), so whole 8-byte steps are in bounds and padding
// bits contribute nothing.  Replaces for these shapes: torch.nn.functional.linear(BinaryConnect(x), safeSign / ternary(W))
// (QuantTorch/layers/binary_layers.py:42-46, terner_layers.py:47-51).
#include "qt_common.h"

namespace {

__device__ __forceinline__ int bcnt_acc(uint32_t v, int acc) {
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(acc));
    return r;
}

// sum over the 64 lanes of a wave, returned wave-uniform
__device__ __forceinline__ int wave_sum_dpp(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror: lanes i <-> 7 - i of each 8
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror: lanes i <-> 15 - i of each 16
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

template <bool TERNARY, bool SWAP, int RS, int F>
__global__ __launch_bounds__(256) void popc_stream_kernel(
    const uint32_t* __restrict__ S0, const uint32_t* __restrict__ S1, int64_t lds_w,    // streamed operand: [Rs][lds_w] words
    const uint32_t* __restrict__ F0, const uint32_t* __restrict__ F1, int64_t ldf_w,    // few operand: [Rf][ldf_w] words
    const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy, int Rs, int Rf, int K) {
    static_assert(RS * F <= 64, "one result per lane");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * RS;          // wave-uniform
    if (row0 >= Rs) return;
    const int kw2 = ((K + 31) / 32 + 1) / 2;                              // 8-byte steps per row

    // the ternary operand's planes: (mask, sign); the binary one: sign only
    const uint2* srow[RS];
    const uint2* srow1[RS];
#pragma unroll
    for (int r = 0; r < RS; ++r) {
        const int64_t row = row0 + r < Rs ? row0 + r : Rs - 1;           // clamped: results of rows past the end are not stored
        srow[r] = reinterpret_cast<const uint2*>(S0 + row * lds_w);
        srow1[r] = (TERNARY && !SWAP) ? reinterpret_cast<const uint2*>(S1 + row * lds_w) : nullptr;
    }
    int acc[RS][F];
    int macc[SWAP ? F : RS];
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
        for (int f = 0; f < F; ++f) acc[r][f] = 0;
#pragma unroll
    for (int i = 0; i < (SWAP ? F : RS); ++i) macc[i] = 0;

#pragma unroll 2
    for (int k = lane; k < kw2; k += 64) {
        uint2 s[RS], s1[RS];
#pragma unroll
        for (int r = 0; r < RS; ++r) {
            s[r] = srow[r][k];
            if (TERNARY && !SWAP) s1[r] = srow1[r][k];
        }
        if (TERNARY && !SWAP) {
#pragma unroll
            for (int r = 0; r < RS; ++r) macc[r] = bcnt_acc(s[r].y, bcnt_acc(s[r].x, macc[r]));
        }
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int64_t fr = f < Rf ? f : Rf - 1;
            const uint2 x = reinterpret_cast<const uint2*>(F0 + fr * ldf_w)[k];
            uint2 x1 = make_uint2(0, 0);
            if (TERNARY && SWAP) {
                x1 = reinterpret_cast<const uint2*>(F1 + fr * ldf_w)[k];
                macc[f] = bcnt_acc(x.y, bcnt_acc(x.x, macc[f]));
            }
#pragma unroll
            for (int r = 0; r < RS; ++r) {
                if (!TERNARY) {
                    acc[r][f] = bcnt_acc(x.y ^ s[r].y, bcnt_acc(x.x ^ s[r].x, acc[r][f]));
                } else if (!SWAP) {     // streamed = (mask s, sign s1), few = activation signs x
                    acc[r][f] = bcnt_acc((x.y ^ s1[r].y) & s[r].y, bcnt_acc((x.x ^ s1[r].x) & s[r].x, acc[r][f]));
                } else {                // streamed = activation signs s, few = (mask x, sign x1)
                    acc[r][f] = bcnt_acc((s[r].y ^ x1.y) & x.y, bcnt_acc((s[r].x ^ x1.x) & x.x, acc[r][f]));
                }
            }
        }
    }

    // wavefront reductions; result j = r * F + f ends up on lane j
    int mine = 0, mbase = K;
    int mtot[SWAP ? F : RS];
    if (TERNARY) {
#pragma unroll
        for (int i = 0; i < (SWAP ? F : RS); ++i) mtot[i] = wave_sum_dpp(macc[i]);
    }
#pragma unroll
    for (int r = 0; r < RS; ++r)
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int t = wave_sum_dpp(acc[r][f]);
            if (lane == r * F + f) {
                mine = t;
                if (TERNARY) mbase = mtot[SWAP ? f : r];
            }
        }
    if (lane < RS * F) {
        const int r = lane / F, f = lane - r * F;
        const int64_t row = row0 + r;
        if (row < Rs && f < Rf) {
            const int64_t m = SWAP ? row : f, n = SWAP ? f : row;
            Y[m * ldy + n] = (float)(mbase - 2 * mine) + (bias ? bias[n] : 0.0f);
        }
    }
}

template <bool TERNARY, bool SWAP, int RS, int F>
void launch_one(const uint32_t* S0, const uint32_t* S1, int64_t lds_w, const uint32_t* F0, const uint32_t* F1, int64_t ldf_w,
                const float* bias, float* Y, int64_t ldy, int64_t Rs, int64_t Rf, int64_t K, hipStream_t st) {
    const int64_t grid = (Rs + 4 * RS - 1) / (4 * RS);
    hipLaunchKernelGGL((popc_stream_kernel<TERNARY, SWAP, RS, F>), dim3((unsigned)grid), dim3(256), 0, st, S0, S1, lds_w, F0, F1, ldf_w,
                       bias, Y, ldy, (int)Rs, (int)Rf, (int)K);
}

template <bool TERNARY, bool SWAP>
void launch_f(const uint32_t* S0, const uint32_t* S1, int64_t lds_w, const uint32_t* F0, const uint32_t* F1, int64_t ldf_w,
              const float* bias, float* Y, int64_t ldy, int64_t Rs, int64_t Rf, int64_t K, hipStream_t st) {
    // rows per wave: enough waves to cover the chip (1024 = 256 CUs x 4) before a wave takes several rows
    if (Rf <= 2) {
        if (Rs >= 4096) launch_one<TERNARY, SWAP, 4, 2>(S0, S1, lds_w, F0, F1, ldf_w, bias, Y, ldy, Rs, Rf, K, st);
        else launch_one<TERNARY, SWAP, 1, 2>(S0, S1, lds_w, F0, F1, ldf_w, bias, Y, ldy, Rs, Rf, K, st);
    } else if (Rf <= 8) {
        if (Rs >= 4096) launch_one<TERNARY, SWAP, 4, 8>(S0, S1, lds_w, F0, F1, ldf_w, bias, Y, ldy, Rs, Rf, K, st);
        else launch_one<TERNARY, SWAP, 1, 8>(S0, S1, lds_w, F0, F1, ldf_w, bias, Y, ldy, Rs, Rf, K, st);
    } else if (Rf <= 16) {     // the few operand's loads are shared by the RS rows of a wave
        if (Rs >= 4096) launch_one<TERNARY, SWAP, 4, 16>(S0, S1, lds_w, F0, F1, ldf_w, bias, Y, ldy, Rs, Rf, K, st);
        else if (Rs >= 2048) launch_one<TERNARY, SWAP, 2, 16>(S0, S1, lds_w, F0, F1, ldf_w, bias, Y, ldy, Rs, Rf, K, st);
        else launch_one<TERNARY, SWAP, 1, 16>(S0, S1, lds_w, F0, F1, ldf_w, bias, Y, ldy, Rs, Rf, K, st);
    } else {
        if (Rs >= 2048) launch_one<TERNARY, SWAP, 2, 32>(S0, S1, lds_w, F0, F1, ldf_w, bias, Y, ldy, Rs, Rf, K, st);
        else launch_one<TERNARY, SWAP, 1, 32>(S0, S1, lds_w, F0, F1, ldf_w, bias, Y, ldy, Rs, Rf, K, st);
    }
}

}  // namespace

// shared with popc_gemm.hip's launcher (arguments already validated there).  Applicable iff min(M, N) <= 32.
bool qt_popc_stream_applicable(int64_t M, int64_t N) { return (M >= 1 && N >= 1) && (M <= 32 || N <= 32); }

int qt_launch_popc_stream(bool ternary, const uint32_t* Xs, int64_t ldx, const uint32_t* W0, const uint32_t* W1, int64_t ldw,
                          const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, qt_stream_t stream) {
    if (!qt_popc_stream_applicable(M, N)) return QT_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // the operand with fewer rows is the "few" one; a tie streams the weights (activations of a batch <= 32 are the few rows)
    const bool swap = N < M && N <= 32;
    if (!swap) {     // streamed = W (ternary: mask W0, sign W1), few = X
        if (ternary) launch_f<true, false>(W0, W1, ldw, Xs, nullptr, ldx, bias, Y, ldy, N, M, K, st);
        else launch_f<false, false>(W0, nullptr, ldw, Xs, nullptr, ldx, bias, Y, ldy, N, M, K, st);
    } else {         // streamed = X, few = W
        if (ternary) launch_f<true, true>(Xs, nullptr, ldx, W0, W1, ldw, bias, Y, ldy, M, N, K, st);
        else launch_f<false, true>(Xs, nullptr, ldx, W0, nullptr, ldw, bias, Y, ldy, M, N, K, st);
    }
    return qt_check_launch();
}
