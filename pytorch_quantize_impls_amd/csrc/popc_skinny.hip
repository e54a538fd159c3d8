// Skinny XNOR / ternary popcount GEMM for the weight-streaming regime: small batch (M <= a few hundred)
// or very few output features (classifier heads).  There the contraction is bound by streaming the
// packed weight planes once (N*K/8 bytes) and by launch latency, not by VALU throughput, and the
// 128x128-tile kernel leaves most CUs idle.
//
//   lane  <-> batch row m (64 rows per workgroup pass), so the popcount accumulate needs NO cross-lane
//             reduction;
//   X     :   the 64 x Kw sign words of the pass are staged in LDS TRANSPOSED ([kw][m], row stride 65 words
//             -> conflict-free writes by kw and reads by m);
//   W     :   each wave owns NR consecutive weight rows at a time; its words are wave-uniform, so they are
//             streamed with scalar loads (s_load_dwordx4/x8) straight into SGPRs — every packed weight word
//             is fetched once per 64 batch rows and costs no VGPR;
//   inner :   per k-word: 1 ds_read_b32 + NR x (v_xor + v_bcnt-accumulate)  [ternary: bitop3 + bcnt].
//
// Grid: x = ceil(N / (4 waves * NR)), y = ceil(M / 64).  K is processed in LDS tiles of KT words.
#include "qt_common.h"

namespace {

constexpr int SK_NR = 4;      // weight rows per wave per pass
constexpr int SK_KT = 256;    // k-words per LDS tile (64 x 256 x 4 B = 65 KB incl. padding)

__device__ __forceinline__ int popc_acc_sk(uint32_t v, int acc) {
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(acc));
    return r;
}

template <bool TERNARY>
__global__ __launch_bounds__(256) void popc_skinny_kernel(
    const uint32_t* __restrict__ Xs, int64_t ldx, const uint32_t* __restrict__ W0,
    const uint32_t* __restrict__ W1, int64_t ldw, const float* __restrict__ bias,
    float* __restrict__ Y, int64_t ldy, int M, int N, int K) {
    __shared__ uint32_t xt[SK_KT * 65];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * 64;
    const int n_base = (blockIdx.x * 4 + wave) * SK_NR;   // wave-uniform
    const int kw = (K + 31) / 32;

    int acc[SK_NR];
    int macc[SK_NR];   // popcount of the mask rows (ternary): wave-uniform, kept scalar by the compiler
#pragma unroll
    for (int r = 0; r < SK_NR; ++r) { acc[r] = 0; macc[r] = 0; }

    for (int k0 = 0; k0 < kw; k0 += SK_KT) {
        const int kn = min(SK_KT, kw - k0);
        const int kn4 = (kn + 3) & ~3;   // planes are zero-padded to ld (% 4 words): whole uint4 steps are safe
        __syncthreads();  // previous tile consumed
        // stage X[m0..m0+63][k0..k0+kn4) transposed: coalesced global reads along k, LDS address k*65+m
        for (int i = tid; i < 64 * kn4; i += 256) {
            const int m = i / kn4, k = i - m * kn4;
            uint32_t v = 0;
            if (m0 + m < M) v = Xs[(int64_t)(m0 + m) * ldx + k0 + k];
            xt[k * 65 + m] = v;
        }
        __syncthreads();
        if (n_base < N) {
            // rows past N are clamped (their results are not stored) so the scalar loads stay in bounds
            const uint4* w0r[SK_NR];
            const uint4* w1r[SK_NR];
#pragma unroll
            for (int r = 0; r < SK_NR; ++r) {
                const int n = min(n_base + r, N - 1);
                w0r[r] = reinterpret_cast<const uint4*>(W0 + (int64_t)n * ldw + k0);
                w1r[r] = TERNARY ? reinterpret_cast<const uint4*>(W1 + (int64_t)n * ldw + k0) : nullptr;
            }
#pragma unroll 2
            for (int k4 = 0; k4 < kn4 / 4; ++k4) {
                // 4 k-words per step: the wave-uniform 16-byte weight loads become s_load_dwordx4
                uint4 a[SK_NR], sg[SK_NR];
#pragma unroll
                for (int r = 0; r < SK_NR; ++r) {
                    a[r] = w0r[r][k4];
                    if (TERNARY) sg[r] = w1r[r][k4];
                }
                const uint32_t x0 = xt[(k4 * 4 + 0) * 65 + lane], x1 = xt[(k4 * 4 + 1) * 65 + lane];
                const uint32_t x2 = xt[(k4 * 4 + 2) * 65 + lane], x3 = xt[(k4 * 4 + 3) * 65 + lane];
#pragma unroll
                for (int r = 0; r < SK_NR; ++r) {
                    if (TERNARY) {
                        macc[r] += __builtin_popcount(a[r].x) + __builtin_popcount(a[r].y) +
                                   __builtin_popcount(a[r].z) + __builtin_popcount(a[r].w);
                        acc[r] = popc_acc_sk((x0 ^ sg[r].x) & a[r].x, acc[r]);
                        acc[r] = popc_acc_sk((x1 ^ sg[r].y) & a[r].y, acc[r]);
                        acc[r] = popc_acc_sk((x2 ^ sg[r].z) & a[r].z, acc[r]);
                        acc[r] = popc_acc_sk((x3 ^ sg[r].w) & a[r].w, acc[r]);
                    } else {
                        acc[r] = popc_acc_sk(x0 ^ a[r].x, acc[r]);
                        acc[r] = popc_acc_sk(x1 ^ a[r].y, acc[r]);
                        acc[r] = popc_acc_sk(x2 ^ a[r].z, acc[r]);
                        acc[r] = popc_acc_sk(x3 ^ a[r].w, acc[r]);
                    }
                }
            }
        }
    }
    const int m = m0 + lane;
    if (m < M && n_base < N) {
#pragma unroll
        for (int r = 0; r < SK_NR; ++r) {
            const int n = n_base + r;
            if (n < N) {
                const int base = TERNARY ? macc[r] : K;
                Y[(int64_t)m * ldy + n] = (float)(base - 2 * acc[r]) + (bias ? bias[n] : 0.0f);
            }
        }
    }
}

}  // namespace

// shared with popc_gemm.hip's launcher
int qt_launch_popc_skinny(bool ternary, const uint32_t* Xs, int64_t ldx, const uint32_t* W0,
                          const uint32_t* W1, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                          int64_t M, int64_t N, int64_t K, qt_stream_t stream) {
    dim3 grid((unsigned)((N + 4 * SK_NR - 1) / (4 * SK_NR)), (unsigned)((M + 63) / 64));
    if (ternary)
        hipLaunchKernelGGL((popc_skinny_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, Xs, ldx, W0, W1,
                           ldw, bias, Y, ldy, (int)M, (int)N, (int)K);
    else
        hipLaunchKernelGGL((popc_skinny_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, Xs, ldx, W0, W1,
                           ldw, bias, Y, ldy, (int)M, (int)N, (int)K);
    return qt_check_launch();
}
