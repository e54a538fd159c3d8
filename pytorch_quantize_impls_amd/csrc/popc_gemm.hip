// XNOR-popcount GEMM and binary x ternary GEMM on bit planes — the VALU formulation
// (v_xor_b32 + v_bcnt_u32_b32 with accumulate: 32 MACs per two full-rate lane-ops).
//
//   Y[m,n] = K - 2 * sum_w popc(X[m,w] ^ W[n,w])                       (binary  x binary)
//   Y[m,n] = sum_w popc(Wm[n,w]) - 2 * sum_w popc((X[m,w]^Ws[n,w]) & Wm[n,w])  (binary x ternary)
//
// Pad bits are zero in every plane, so they vanish from the XOR (and from the mask) and the true
// K enters only through the constant term.
//
// Tiling (wave64, 256-thread workgroup = 4 waves):
//   * workgroup tile 128 (M) x 128 (N); thread (ty = tid>>4, tx = tid&15) owns 8 rows
//     (ty*8+i) x 8 columns (tx*4+{0..3} and 64+tx*4+{0..3}) -> two float4 stores per row,
//     16 lanes covering 256 contiguous bytes of a Y row.
//   * K is consumed in tiles of KT words staged through LDS with a row stride of KT+4 words:
//     for KT=32 (stride 36) and KT=16 (stride 20) sixteen consecutive rows start on sixteen
//     distinct multiples of 4 banks mod 64, so a 16-lane ds_read_b128 group is conflict-free.
//     W rows are stored in LDS permuted (p = j*16 + tx) so that, for a fixed register index j,
//     the 16 tx-lanes read 16 consecutive LDS rows; X rows are wave-broadcast reads.
//   * inner step: 8 ds_read_b128 (X) + 8 ds_read_b128 (W) feed 8*8*4 xor+bcnt pairs = 512 VALU
//     ops -> LDS traffic is ~3 % of issue slots; the kernel is VALU-bound by construction.
//   * <=128 VGPRs and 36 KiB LDS -> 4 workgroups (16 waves) per CU; a 4096x4096 problem is
//     1024 workgroups = exactly one resident round on 256 CUs.
#include "qt_common.h"

int qt_launch_popc_skinny(bool ternary, const uint32_t* Xs, int64_t ldx, const uint32_t* W0,
                          const uint32_t* W1, int64_t ldw, const float* bias, float* Y, int64_t ldy,
                          int64_t M, int64_t N, int64_t K, qt_stream_t stream);  // popc_skinny.hip
bool qt_popc_stream_applicable(int64_t M, int64_t N);                             // popc_stream.hip
int qt_launch_popc_stream(bool ternary, const uint32_t* Xs, int64_t ldx, const uint32_t* W0, const uint32_t* W1, int64_t ldw,
                          const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, qt_stream_t stream);

namespace {

constexpr int BM = 128, BN = 128;

template <int KT>
struct Tile {
    static constexpr int STRIDE = KT + 4;          // words
    static constexpr int U4_PER_ROW = KT / 4;      // uint4 per tile row
    static constexpr int NLD = BM * U4_PER_ROW / 256;  // uint4 loads per thread per plane
};

// acc + popcount(v) in ONE instruction.  Written as `acc += __builtin_popcount(v)` LLVM
// reassociates chains of these into v_bcnt(v, 0) + v_add3_u32 (5 VALU per 64 MACs instead of 4),
// so the accumulating form of v_bcnt_u32_b32 (D = popcount(S0) + S1) is pinned here.  Plain
// VALU -> VALU dependency: no wait states needed inside the string.
__device__ __forceinline__ int popc_acc(uint32_t v, int acc) {
    int r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(acc));
    return r;
}

__device__ __forceinline__ void popc8(int (&acc)[8][8], int j, const uint2 (&xv)[8], uint32_t w, int hi) {
    uint32_t t0, t1, t2, t3, t4, t5, t6, t7;
#define QT_X(i) (hi ? xv[i].y : xv[i].x)
    asm("v_xor_b32 %8, %16, %24\n\tv_xor_b32 %9, %17, %24\n\tv_xor_b32 %10, %18, %24\n\tv_xor_b32 %11, %19, %24\n\t"
        "v_xor_b32 %12, %20, %24\n\tv_xor_b32 %13, %21, %24\n\tv_xor_b32 %14, %22, %24\n\tv_xor_b32 %15, %23, %24\n\t"
        "v_bcnt_u32_b32 %0, %8, %0\n\tv_bcnt_u32_b32 %1, %9, %1\n\tv_bcnt_u32_b32 %2, %10, %2\n\tv_bcnt_u32_b32 %3, %11, %3\n\t"
        "v_bcnt_u32_b32 %4, %12, %4\n\tv_bcnt_u32_b32 %5, %13, %5\n\tv_bcnt_u32_b32 %6, %14, %6\n\tv_bcnt_u32_b32 %7, %15, %7"
        : "+v"(acc[0][j]), "+v"(acc[1][j]), "+v"(acc[2][j]), "+v"(acc[3][j]), "+v"(acc[4][j]), "+v"(acc[5][j]),
          "+v"(acc[6][j]), "+v"(acc[7][j]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6),
          "=&v"(t7)
        : "v"(QT_X(0)), "v"(QT_X(1)), "v"(QT_X(2)), "v"(QT_X(3)), "v"(QT_X(4)), "v"(QT_X(5)), "v"(QT_X(6)), "v"(QT_X(7)),
          "v"(w));
#undef QT_X
}

__device__ __forceinline__ int w_lds_row(int n_local) {
    // inverse of: column c_j(tx) = (j<4 ? tx*4+j : 64+tx*4+(j-4))  ->  LDS row j*16+tx
    const int half = n_local >> 6, r = n_local & 63;
    const int tx = r >> 2, j = (r & 3) + 4 * half;
    return j * 16 + tx;
}

template <int KT, bool TERNARY>
__global__ __launch_bounds__(256, 4) void popc_gemm_kernel(
    const uint32_t* __restrict__ Xs, int64_t ldx, const uint32_t* __restrict__ W0,
    const uint32_t* __restrict__ W1, int64_t ldw, const float* __restrict__ bias,
    float* __restrict__ Y, int64_t ldy, int M, int N, int K, int vec_store) {
    using T = Tile<KT>;
    constexpr int S = T::STRIDE;
    __shared__ __attribute__((aligned(16))) uint32_t lds[(TERNARY ? 3 : 2) * BM * S];
    uint32_t* Xl = lds;
    uint32_t* W0l = lds + BM * S;
    uint32_t* W1l = lds + 2 * BM * S;  // only touched when TERNARY

    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    int acc[8][8];
    int macc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        macc[i] = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = 0;
    }

    // rows are zero beyond ceil(K/32) words up to ld (format invariant)
    const int kw = (K + 31) / 32;
    const int ntiles = (kw + KT - 1) / KT;

    for (int kt = 0; kt < ntiles; ++kt) {
        // stage one K-tile: global -> registers -> (barrier) -> LDS.  Out-of-range rows / words
        // are replaced by zeros (they then vanish from the XOR / mask).
        uint4 xr[T::NLD], w0r[T::NLD], w1r[TERNARY ? T::NLD : 1];
#pragma unroll
        for (int q = 0; q < T::NLD; ++q) {
            const int id = q * 256 + tid;
            const int row = id / T::U4_PER_ROW, c4 = id % T::U4_PER_ROW;
            const int64_t kword = (int64_t)kt * KT + c4 * 4;
            uint4 vx = make_uint4(0, 0, 0, 0), v0 = vx, v1 = vx;
            if (m0 + row < M && kword < ldx)
                vx = *reinterpret_cast<const uint4*>(Xs + (int64_t)(m0 + row) * ldx + kword);
            if (n0 + row < N && kword < ldw) {
                v0 = *reinterpret_cast<const uint4*>(W0 + (int64_t)(n0 + row) * ldw + kword);
                if constexpr (TERNARY)
                    v1 = *reinterpret_cast<const uint4*>(W1 + (int64_t)(n0 + row) * ldw + kword);
            }
            xr[q] = vx;
            w0r[q] = v0;
            if constexpr (TERNARY) w1r[q] = v1;
        }
        if (kt > 0) __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int q = 0; q < T::NLD; ++q) {
            const int id = q * 256 + tid;
            const int row = id / T::U4_PER_ROW, c4 = id % T::U4_PER_ROW;
            *reinterpret_cast<uint4*>(Xl + row * S + c4 * 4) = xr[q];
            const int wrow = w_lds_row(row);
            *reinterpret_cast<uint4*>(W0l + wrow * S + c4 * 4) = w0r[q];
            if constexpr (TERNARY) *reinterpret_cast<uint4*>(W1l + wrow * S + c4 * 4) = w1r[q];
        }
        __syncthreads();

#pragma unroll 1
        for (int kk = 0; kk < KT / 2; ++kk) {
            // 2 words (64 bits of K) per step: 8 + 8 ds_read_b64 feed 8*8*2 xor+bcnt pairs.
            uint2 xv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                xv[i] = *reinterpret_cast<const uint2*>(Xl + (ty * 8 + i) * S + kk * 2);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint2 wv = *reinterpret_cast<const uint2*>(W0l + (j * 16 + tx) * S + kk * 2);
                if (TERNARY) {
                    const uint2 sv =
                        *reinterpret_cast<const uint2*>(W1l + (j * 16 + tx) * S + kk * 2);
                    macc[j] = popc_acc(wv.y, popc_acc(wv.x, macc[j]));
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        int a = acc[i][j];
                        a = popc_acc((xv[i].x ^ sv.x) & wv.x, a);
                        a = popc_acc((xv[i].y ^ sv.y) & wv.y, a);
                        acc[i][j] = a;
                    }
                } else {
                    // eight independent xors, then eight independent accumulating popcounts (one asm statement, so the
                    // scheduler cannot re-pair them): a bcnt issued right behind the xor it depends on waits out the VALU
                    // latency (SQ_WAIT_INST_ANY 46 % of wave cycles at 4 waves / SIMD, profiles/r2_popc_pmc.md)
                    popc8(acc, j, xv, wv.x, 0);
                    popc8(acc, j, xv, wv.y, 1);
                }
            }
        }
    }

    // epilogue: integer -> fp32 once, bias added once, two float4 stores per row
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int nb = n0 + h * 64 + tx * 4;
        float b4[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (nb + c < N) b4[c] = bias[nb + c];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + ty * 8 + i;
            if (m >= M) continue;
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = h * 4 + c;
                const int base = TERNARY ? macc[j] : K;
                v[c] = (float)(base - 2 * acc[i][j]) + b4[c];
            }
            float* yp = Y + (int64_t)m * ldy + nb;
            if (vec_store && nb + 3 < N) {
                *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (nb + c < N) yp[c] = v[c];
            }
        }
    }
}

// variant: 0 = automatic, 1 = 128x128-tile kernel, 2 = skinny (lane <-> batch row) kernel, 3 = streaming kernel (K along the
// lanes + DPP wavefront reduction; min(M, N) <= 32 only)
template <bool TERNARY>
int launch_popc_gemm(int variant, const uint32_t* Xs, int64_t ldx, const uint32_t* W0, const uint32_t* W1,
                     int64_t ldw, const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N,
                     int64_t K, qt_stream_t stream) {
    if (variant < 0 || variant > 3) return QT_ERR_INVALID_ARG;
    if (M < 0 || N < 0 || K < 0) return QT_ERR_INVALID_ARG;
    if (M == 0 || N == 0) return QT_OK;
    if (!Y || ldy < N) return QT_ERR_INVALID_ARG;
    if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return QT_ERR_UNSUPPORTED;
    const int64_t kw = (K + 31) / 32;
    if (K > 0 && (!Xs || !W0 || (TERNARY && !W1))) return QT_ERR_INVALID_ARG;
    if (ldx < kw || ldw < kw) return QT_ERR_INVALID_ARG;
    if ((ldx & 3) || (ldw & 3)) return QT_ERR_ALIGNMENT;
    if (K > 0 && (!qt_aligned16(Xs) || !qt_aligned16(W0) || (TERNARY && !qt_aligned16(W1))))
        return QT_ERR_ALIGNMENT;
    // one operand with a handful of rows (batch <= 32, or a classifier head): K along the lanes, every packed word read once
    if (variant == 3 && !qt_popc_stream_applicable(M, N)) return QT_ERR_UNSUPPORTED;
    if (variant == 3 || (variant == 0 && qt_popc_stream_applicable(M, N)))
        return qt_launch_popc_stream(TERNARY, Xs, ldx, W0, W1, ldw, bias, Y, ldy, M, N, K, stream);
    // weight-streaming regime: few batch rows or few output features -> the tiled kernel would leave
    // most CUs idle (tiles = ceil(M/128)*ceil(N/128) << 256) while the skinny kernel spreads N over waves
    const int64_t tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const bool skinny = variant == 2 || (variant == 0 && (M <= 64 || N <= 64 || tiles < 128) && M <= 4096);
    if (skinny && (M + 63) / 64 <= 65535)
        return qt_launch_popc_skinny(TERNARY, Xs, ldx, W0, W1, ldw, bias, Y, ldy, M, N, K, stream);
    const int64_t gy = (M + BM - 1) / BM, gx = (N + BN - 1) / BN;
    if (gy > 65535) return QT_ERR_UNSUPPORTED;  // callers split M (conv im2col rows) above this
    const int vec_store = qt_aligned16(Y) && (ldy % 4 == 0);
    dim3 grid((unsigned)gx, (unsigned)gy);
    if (TERNARY) {
        hipLaunchKernelGGL((popc_gemm_kernel<16, true>), grid, dim3(256), 0, (hipStream_t)stream, Xs,
                           ldx, W0, W1, ldw, bias, Y, ldy, (int)M, (int)N, (int)K, vec_store);
    } else {
        hipLaunchKernelGGL((popc_gemm_kernel<16, false>), grid, dim3(256), 0, (hipStream_t)stream,
                           Xs, ldx, W0, W1, ldw, bias, Y, ldy, (int)M, (int)N, (int)K, vec_store);
    }
    return qt_check_launch();
}

}  // namespace

extern "C" {

int qt_xnor_gemm_variant(int variant, const uint32_t* Xs, int64_t ldxp, const uint32_t* Ws, int64_t ldwp,
                         const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, qt_stream_t stream) {
    return launch_popc_gemm<false>(variant, Xs, ldxp, Ws, nullptr, ldwp, bias, Y, ldy, M, N, K, stream);
}

int qt_tern_gemm_variant(int variant, const uint32_t* Xs, int64_t ldxp, const uint32_t* Wmask, const uint32_t* Wsign,
                         int64_t ldwp, const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                         qt_stream_t stream) {
    return launch_popc_gemm<true>(variant, Xs, ldxp, Wmask, Wsign, ldwp, bias, Y, ldy, M, N, K, stream);
}

int qt_xnor_gemm(const uint32_t* Xs, int64_t ldxp, const uint32_t* Ws, int64_t ldwp,
                 const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                 qt_stream_t stream) {
    return launch_popc_gemm<false>(0, Xs, ldxp, Ws, nullptr, ldwp, bias, Y, ldy, M, N, K, stream);
}

int qt_tern_gemm(const uint32_t* Xs, int64_t ldxp, const uint32_t* Wmask, const uint32_t* Wsign,
                 int64_t ldwp, const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N,
                 int64_t K, qt_stream_t stream) {
    return launch_popc_gemm<true>(0, Xs, ldxp, Wmask, Wsign, ldwp, bias, Y, ldy, M, N, K, stream);
}

}  // extern "C"
