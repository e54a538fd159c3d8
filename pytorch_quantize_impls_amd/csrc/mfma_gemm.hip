// Packed low-bit GEMMs on the matrix cores.
//
//   fp4 ("nibble planes", include/qt_hip.h): +-1 / {-1,0,+1} operands as FP4-E2M1 nibbles, contracted
//       with the block-scaled MX MFMA  v_mfma_scale_f32_32x32x64_f8f6f4  (A = B = fp4, every block
//       scale 2^0).  E2M1 holds -1, 0, +1 exactly and the accumulator is fp32, so every partial sum is
//       an integer of magnitude <= K < 2^24: bit-identical to the popcount formulation and to the
//       reference's fp32 GEMM on +-1 tensors.  Measured issue rate on MI355X: 9.0 PFLOP/s vs 1.49 Pop/s
//       for the xor+bcnt pair (profiles/r1_ubench.txt) — which is why large shapes are routed here.
//   int8 ("code planes"): DoReFa k-bit activation codes q = rint((2^k-1) x) x +-1/0 weight codes with
//       v_mfma_i32_32x32x32_i8, int32 accumulate (exact), one fp32 scale in the epilogue.
//   bf16 ("triple planes"): real-valued activations split exactly into hi + mid + lo bf16 terms x +-1/0 weights
//       replicated three times, v_mfma_f32_32x32x16_bf16, fp32 accumulate (fp32-GEMM accuracy).
//
// Y[m,n] = scale * sum_k X[m,k] * W[n,k] (+ bias);  X: M x K, W: N x K, row-major byte rows.
//
// One kernel template serves all three: everything is organised in BYTES of K.  A stage is 128 (PIPE 0/1) or 64
// (PIPE 2) bytes of K per row = 4 / 2 MFMA k-steps of 32 bytes; lane l of a wave supplies, for the 32x32 MFMA,
// row (l & 31) and the 16-byte chunk (2*kk + (l >> 5)) of the stage row.
//
// Workgroup = 8 waves (2 per SIMD).  X is the MFMA "A" operand (D rows = m) and W the "B" operand (D cols = n).
// Epilogues: fp32 Y through a wave-private LDS transpose and dwordx4 full-line stores (or dword stores when Y is
// not 16-byte friendly), or — inference fusion — one THRESHOLD BIT per output ([(acc + bias) * alpha + beta < 0],
// EpiArgs) assembled with wave ballots.
//
// LDS: stage rows are 16-byte chunks, chunk c of row r stored at position c ^ f(r) (swz<>): the 16 rows a
// ds_read_b128 lane group touches (distinct mod 16) fall on 16 distinct 16-B slots of the 256-B bank row ->
// conflict-free fragment reads.
//
// Staging is LDS-DMA (global_load_lds_dwordx4: no VGPR round trip; the swizzle is applied on the per-lane SOURCE
// address because the LDS side of the instruction is lane-linear).
//   PIPE = 2 (ping-pong; every GEMM tile, conv on the 384x192 / 256x256 tiles): 64-byte stages in a ring of 4;
//       a wave's stage is a LOAD segment (all fragment reads of the stage + its DMA pieces of stage s+3) followed
//       by a COMPUTE segment (register-only MFMAs), and the two waves of a SIMD run one segment apart, so one owns
//       the matrix pipe while the other owns LDS / DMA issue.  See the comment at the loop.
//   PIPE = 1 (double-buffered; the remaining conv tiles): the DMA is issued from inline asm, so hipcc has no
//       outstanding-DMA knowledge (with the builtin it drains vmcnt(0) in front of the next ds_read); the pieces of
//       stage s+1 are interleaved with software-pipelined fragment reads and the MFMAs of stage s; one manual
//       vmcnt(0) + barrier per stage; last stage peeled so the steady-state body is one basic block.
//       Contract of PIPE 1/2: row strides are whole 128-byte stages (ld % 32 words == 0, pad zero) and each operand
//       spans < 2^31 bytes (32-bit lane offsets).
//   PIPE = 0 (generic): builtin DMA, 64-bit addresses, any ld % 4 words; chunks past the row stride read a 16-byte
//       zero word.  All fragment reads of a stage are issued before the DMA of the next.
// Conv (CONV_ = 1 / 2): the X operand is an NHWC pixel plane and a stage gathers the 16-byte chunks of the taps it
// covers (implicit GEMM, no im2col buffer): per-tap bounds checks + zero page (1), or — un-padded / physically
// padded planes — plain 32-bit offsets with the per-(stage, chunk) tap offsets tabulated once in LDS (2).
// Profiling-only variants (ABL_ != 0) exist only in -DQT_PROFILING_VARIANTS builds (qt_nib_gemm_variant 161-166,
// qt_conv2d_implicit_variant 3); the product library does not contain them.

#include <cstdio>
#include "mfma_gemm_kernel.h"

namespace {

int check_common(const void* Xn, int64_t ldxp, const void* Wn, int64_t ldwp, const float* Y, int64_t ldy,
                 int64_t M, int64_t N, int64_t K, int64_t kwords) {
    if (M < 0 || N < 0 || K < 0) return QT_ERR_INVALID_ARG;
    if (M == 0 || N == 0) return 1;  // nothing to do
    if (!Y || ldy < N) return QT_ERR_INVALID_ARG;
    if (M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return QT_ERR_UNSUPPORTED;
    if (K > 0 && (!Xn || !Wn)) return QT_ERR_INVALID_ARG;
    if (ldxp < kwords || ldwp < kwords) return QT_ERR_INVALID_ARG;
    if ((ldxp & 3) || (ldwp & 3)) return QT_ERR_ALIGNMENT;
    if (K > 0 && (!qt_aligned16(Xn) || !qt_aligned16(Wn))) return QT_ERR_ALIGNMENT;
    return QT_OK;
}

template <class E>
int dispatch_gemm(int variant, const uint32_t* Xn, int64_t ldxp, const uint32_t* Wn, int64_t ldwp,
                  const float* bias, float scale, const float* scale_dev, float* Y, int64_t ldy, int64_t M,
                  int64_t N, int64_t K, qt_stream_t stream) {
    const bool pipe_ok = !(ldxp & 31) && !(ldwp & 31) && M * ldxp * 4 < (1ll << 31) &&
                         N * ldwp * 4 < (1ll << 31);
#define QT_GO(...) return launch_cfg<__VA_ARGS__>(Xn, ldxp, Wn, ldwp, bias, scale, scale_dev, Y, ldy, M, N, K, stream)
    switch (variant) {
        case 0: {  // automatic: tile width by N and CU fill, fast path when its contract holds
            const int tn = pick_tile_n_gemm(M, N);
            // skinny (FC at batch <= 512): the K loop is a chain of latency-bound stage round trips on a quarter of
            // the CUs — 128x64 tiles and 256-byte stages (tools/bench_gemm_variants.py: 256x4096x25088 75 -> 48 us)
            // ... and 64x64 tiles with 512-byte stages up to M = 256 (256x4096x9216: 21.4 -> 13.7 us)
            if (pipe_ok && M <= 256 && !((ldxp | ldwp) & 127)) QT_GO(CfgSkinny512<E>);
            if (pipe_ok && M <= 512 && !((ldxp | ldwp) & 63)) QT_GO(CfgSkinny<E>);
            if (pipe_ok) {
                if (tn == 256) QT_GO(PP256<E>);
                if (tn == 192 && prefer_384_rows(M, N)) QT_GO(PP384x192<E>);
                if (tn == 192) QT_GO(PP192<E>);
                if (tn == 128) QT_GO(PP128<E>);
                QT_GO(Cfg64<E, 1>);
            }
            if (tn == 256) QT_GO(Cfg256<E, 0>);
            if (tn == 192) QT_GO(Cfg192<E, 0>);
            if (tn == 128) QT_GO(Cfg128<E, 0>);
            QT_GO(Cfg64<E, 0>);
        }
        case 15: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(Cfg192<E, 1>);
        case 16: QT_GO(Cfg192<E, 0>);
        case 5: QT_GO(Cfg256<E, 0>);
        case 6: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(Cfg256<E, 1>);
        case 7: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(Cfg128<E, 1>);
        case 8: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(Cfg64<E, 1>);
        case 9: QT_GO(Cfg128<E, 0>);
        case 10: QT_GO(Cfg64<E, 0>);
        case 30: if (!pipe_ok || ((ldxp | ldwp) & 63)) return QT_ERR_ALIGNMENT; QT_GO(CfgSkinny<E>);
        case 31: if (!pipe_ok || ((ldxp | ldwp) & 127)) return QT_ERR_ALIGNMENT; QT_GO(CfgSkinny512<E>);
        case 20: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(PP256<E>);
#ifdef QT_PROFILING_VARIANTS   // stamped / ablated kernels (Y is garbage): never in the product library
        case 165: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(PP256<E, 5>);
        case 166: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(PP256<E, 6>);
        case 161: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(Cfg256<E, 1, 1>);
        case 162: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(Cfg256<E, 1, 2>);
        case 163: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(Cfg256<E, 1, 3>);
        case 164: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(Cfg256<E, 1, 4>);
#endif
        case 21: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(PP128<E>);
        case 22: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(PP192<E>);
        case 23: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(PP64<E>);
        case 24: if (!pipe_ok) return QT_ERR_ALIGNMENT; QT_GO(PP384x192<E>);
        default: return QT_ERR_UNSUPPORTED;
    }
#undef QT_GO
}

// ---- fp32 -> nibble plane ------------------------------------------------------------------------
struct NibSign {  // safeSign: +1 -> 0x2, -1 -> 0xA
    __device__ __forceinline__ static uint32_t nib(float x) { return x < 0.0f ? 0xAu : 0x2u; }
};
struct NibTernary {  // TernaryConnectDeterministic: 0 -> 0x0
    __device__ __forceinline__ static uint32_t nib(float x) {
        const float t = qt_ternarize(x);
        return t == 0.0f ? 0x0u : (t < 0.0f ? 0xAu : 0x2u);
    }
};

struct NibSign0 {  // torch.sign: 0 (and NaN) -> 0x0 — the XNOR-Net weight image sign(W) (functions/xnor_connect.py:141)
    __device__ __forceinline__ static uint32_t nib(float x) { return x > 0.0f ? 0x2u : (x < 0.0f ? 0xAu : 0x0u); }
};

// One work item = one float4 slot of the padded row (ldp words = ldp*2 slots); two adjacent lanes
// form one output word.  Slots past K/4 produce zero nibbles, so the pad-is-zero invariant holds.
template <class Enc>
__global__ __launch_bounds__(256) void nib_pack_vec_kernel(const float* __restrict__ x, int64_t ldx,
                                                           uint32_t* __restrict__ out, int64_t ldp,
                                                           int64_t rows, int64_t K) {
    const int64_t slots_per_row = ldp * 2;
    const int64_t total = rows * slots_per_row;  // even
    const int64_t k4 = K / 4;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
         s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = s / slots_per_row, slot = s - row * slots_per_row;
        uint32_t h = 0;
        if (slot < k4) {
            const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + slot * 4);
            h = Enc::nib(v.x) | (Enc::nib(v.y) << 4) | (Enc::nib(v.z) << 8) | (Enc::nib(v.w) << 12);
        }
        const uint32_t other = __shfl_xor(h, 1);
        if ((threadIdx.x & 1) == 0) out[row * ldp + (slot >> 1)] = h | (other << 16);
    }
}

// Both operands of one LinearBin / LinearTer forward in ONE launch (activation: safeSign, weight: EncW): saves a
// kernel boundary and one ramp / tail of a ~12 us HBM-bound kernel.  Same work item as nib_pack_vec_kernel.
template <class EncW>
__global__ __launch_bounds__(256) void nib_pack_pair_kernel(const float* __restrict__ xa, int64_t lda,
                                                            uint32_t* __restrict__ oa, int64_t ldpa, int64_t rowsa,
                                                            const float* __restrict__ xb, int64_t ldb,
                                                            uint32_t* __restrict__ ob, int64_t ldpb, int64_t rowsb,
                                                            int64_t K) {
    const int64_t ta = rowsa * ldpa * 2, total = ta + rowsb * ldpb * 2;   // both even
    const int64_t k4 = K / 4;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
         s += (int64_t)gridDim.x * blockDim.x) {
        const bool second = s >= ta;
        const int64_t t = second ? s - ta : s;
        const int64_t spr = (second ? ldpb : ldpa) * 2;
        const int64_t row = t / spr, slot = t - row * spr;
        uint32_t h = 0;
        if (slot < k4) {
            const float4 v = *reinterpret_cast<const float4*>((second ? xb + row * ldb : xa + row * lda) + slot * 4);
            h = second ? (EncW::nib(v.x) | (EncW::nib(v.y) << 4) | (EncW::nib(v.z) << 8) | (EncW::nib(v.w) << 12))
                       : (NibSign::nib(v.x) | (NibSign::nib(v.y) << 4) | (NibSign::nib(v.z) << 8) | (NibSign::nib(v.w) << 12));
        }
        const uint32_t other = __shfl_xor(h, 1);
        if ((threadIdx.x & 1) == 0) (second ? ob + row * ldpb : oa + row * ldpa)[slot >> 1] = h | (other << 16);
    }
}

// Generic path (any K / alignment): one thread per output word, scalar loads.
template <class Enc>
__global__ __launch_bounds__(256) void nib_pack_scalar_kernel(const float* __restrict__ x, int64_t ldx,
                                                              uint32_t* __restrict__ out, int64_t ldp,
                                                              int64_t rows, int64_t K) {
    const int64_t total = rows * ldp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / ldp, w = i - row * ldp;
        uint32_t word = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t k = w * 8 + e;
            if (k < K) word |= Enc::nib(x[row * ldx + k]) << (4 * e);
        }
        out[i] = word;
    }
}

// ---- bit planes -> nibble plane (derived MFMA operand format) ---------------------------------------
// One thread = one 32-bit word of the planes -> four nibble words (16 B).  sign-only planes:
// nibble = 0x2 | s<<3.  mask+sign: nibble = m<<1 | (s&m)<<3.  Words past the bit planes' stride are 0.
__global__ __launch_bounds__(256) void bits_to_nib_kernel(const uint32_t* __restrict__ sign,
                                                          const uint32_t* __restrict__ mask,
                                                          int64_t ldb, uint32_t* __restrict__ out,
                                                          int64_t ldn, int64_t rows, int64_t K) {
    const int64_t groups_per_row = ldn / 4;  // one group = 4 nibble words = 32 elements
    const int64_t total = rows * groups_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / groups_per_row, g = i - row * groups_per_row;
        uint32_t sw = 0, mw = 0;
        if (g < ldb) {
            sw = sign[row * ldb + g];
            if (mask) {
                mw = mask[row * ldb + g];
            } else {  // binary: every element inside K is non-zero
                const int64_t rem = K - g * 32;
                mw = rem >= 32 ? 0xFFFFFFFFu : (rem > 0 ? ((1u << rem) - 1u) : 0u);
            }
        }
        sw &= mw;
        uint4 o;
        o.x = (spread8(mw) << 1) | (spread8(sw) << 3);
        o.y = (spread8(mw >> 8) << 1) | (spread8(sw >> 8) << 3);
        o.z = (spread8(mw >> 16) << 1) | (spread8(sw >> 16) << 3);
        o.w = (spread8(mw >> 24) << 1) | (spread8(sw >> 24) << 3);
        *reinterpret_cast<uint4*>(out + row * ldn + g * 4) = o;
    }
}

// Same expansion into a PHYSICALLY zero-padded NHWC pixel plane [N][H + 2ph][W + 2pw][ldn]: border pixels
// are fp4 zeros, so a padded conv becomes an un-padded one on this plane and takes the conv kernels'
// VALID mode (no per-tap bounds checks, 32-bit offsets).
__global__ __launch_bounds__(256) void bits_to_nib_pad_kernel(const uint32_t* __restrict__ sign,
                                                              const uint32_t* __restrict__ mask, int64_t ldb,
                                                              uint32_t* __restrict__ out, int64_t ldn, int64_t N,
                                                              int H, int W, int ph, int pw, int64_t K) {
    const int64_t groups_per_row = ldn / 4;
    const int Hp = H + 2 * ph, Wp = W + 2 * pw;
    const int64_t total = N * Hp * Wp * groups_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t orow = i / groups_per_row, g = i - orow * groups_per_row;
        const int64_t n = orow / ((int64_t)Hp * Wp);
        const int rem = (int)(orow - n * Hp * Wp);
        const int y = rem / Wp - ph, x = rem % Wp - pw;
        uint32_t sw = 0, mw = 0;
        if (g < ldb && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            const int64_t row = (n * H + y) * W + x;
            sw = sign[row * ldb + g];
            if (mask) {
                mw = mask[row * ldb + g];
            } else {
                const int64_t r2 = K - g * 32;
                mw = r2 >= 32 ? 0xFFFFFFFFu : (r2 > 0 ? ((1u << r2) - 1u) : 0u);
            }
        }
        sw &= mw;
        uint4 o;
        o.x = (spread8(mw) << 1) | (spread8(sw) << 3);
        o.y = (spread8(mw >> 8) << 1) | (spread8(sw >> 8) << 3);
        o.z = (spread8(mw >> 16) << 1) | (spread8(sw >> 16) << 3);
        o.w = (spread8(mw >> 24) << 1) | (spread8(sw >> 24) << 3);
        *reinterpret_cast<uint4*>(out + orow * ldn + g * 4) = o;
    }
}

template <class Enc>
int launch_nib_pack(const float* x, int64_t ldx, uint32_t* out, int64_t ldp, int64_t rows, int64_t K,
                    qt_stream_t stream) {
    if (rows < 0 || K < 0 || ldx < K) return QT_ERR_INVALID_ARG;
    if (rows == 0) return QT_OK;
    if (!out || (!x && K > 0)) return QT_ERR_INVALID_ARG;
    const int64_t kw = (K + 7) / 8;
    if (ldp < kw || (ldp & 3) != 0 || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    if (ldp == 0) return QT_OK;
    const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && qt_aligned16(x);
    if (vec) {
        const int grid = qt_stream_grid((rows * ldp * 2 + 255) / 256);
        hipLaunchKernelGGL((nib_pack_vec_kernel<Enc>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x,
                           ldx, out, ldp, rows, K);
    } else {
        const int grid = qt_stream_grid((rows * ldp + 255) / 256);
        hipLaunchKernelGGL((nib_pack_scalar_kernel<Enc>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           x, ldx, out, ldp, rows, K);
    }
    return qt_check_launch();
}

}  // namespace

// code_conv3x3.hip
int qt_code_conv3x3_try(const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw, int64_t kh, int64_t kw, int64_t sh, int64_t sw,
                        int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldwp, const float* bias, float scale,
                        const float* scale_dev, const float* alpha, const float* beta, const float* res_f32, const float* res_alpha,
                        const int8_t* res_codes, int64_t ldrc_bytes, float res_scale, int relu, int bit_width, int8_t* codes,
                        int64_t ldc_bytes, int64_t Cout, int32_t* overflow, int64_t ihy, int64_t ihx, int64_t ohy, int64_t ohx,
                        int64_t rhy, int64_t rhx, const float* bn_stats, qt_stream_t stream);

extern "C" {

int qt_sign_pack_nib_f32(const float* x, int64_t ldx, uint32_t* nib_plane, int64_t ldp, int64_t rows,
                         int64_t K, qt_stream_t stream) {
    return launch_nib_pack<NibSign>(x, ldx, nib_plane, ldp, rows, K, stream);
}

int qt_ternary_pack_nib_f32(const float* x, int64_t ldx, uint32_t* nib_plane, int64_t ldp,
                            int64_t rows, int64_t K, qt_stream_t stream) {
    return launch_nib_pack<NibTernary>(x, ldx, nib_plane, ldp, rows, K, stream);
}

int qt_sign0_pack_nib_f32(const float* x, int64_t ldx, uint32_t* nib_plane, int64_t ldp, int64_t rows, int64_t K,
                          qt_stream_t stream) {
    return launch_nib_pack<NibSign0>(x, ldx, nib_plane, ldp, rows, K, stream);
}

int qt_pack_pair_nib_f32(const float* x, int64_t ldx, uint32_t* x_plane, int64_t ldxp, int64_t rows_x,
                         const float* w, int64_t ldw, uint32_t* w_plane, int64_t ldwp, int64_t rows_w, int64_t K,
                         int w_ternary, qt_stream_t stream) {
    if (rows_x < 0 || rows_w < 0 || K < 0 || ldx < K || ldw < K) return QT_ERR_INVALID_ARG;
    const bool vec = rows_x > 0 && rows_w > 0 && K > 0 && (K % 4 == 0) && (ldx % 4 == 0) && (ldw % 4 == 0) &&
                     qt_aligned16(x) && qt_aligned16(w) && x && w && x_plane && w_plane;
    if (!vec) {   // ragged / empty operands: the two single-operand launches handle every case
        const int rc = launch_nib_pack<NibSign>(x, ldx, x_plane, ldxp, rows_x, K, stream);
        if (rc != QT_OK) return rc;
        return w_ternary ? launch_nib_pack<NibTernary>(w, ldw, w_plane, ldwp, rows_w, K, stream)
                         : launch_nib_pack<NibSign>(w, ldw, w_plane, ldwp, rows_w, K, stream);
    }
    const int64_t kw = (K + 7) / 8;
    if (ldxp < kw || ldwp < kw || (ldxp & 3) || (ldwp & 3) || !qt_aligned16(x_plane) || !qt_aligned16(w_plane))
        return QT_ERR_ALIGNMENT;
    const int grid = qt_stream_grid(((rows_x * ldxp + rows_w * ldwp) * 2 + 255) / 256);
    if (w_ternary)
        hipLaunchKernelGGL((nib_pack_pair_kernel<NibTernary>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx,
                           x_plane, ldxp, rows_x, w, ldw, w_plane, ldwp, rows_w, K);
    else
        hipLaunchKernelGGL((nib_pack_pair_kernel<NibSign>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx,
                           x_plane, ldxp, rows_x, w, ldw, w_plane, ldwp, rows_w, K);
    return qt_check_launch();
}

int qt_nib_gemm_variant(int variant, const uint32_t* Xn, int64_t ldxp, const uint32_t* Wn,
                        int64_t ldwp, const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N,
                        int64_t K, qt_stream_t stream) {
    const int rc = check_common(Xn, ldxp, Wn, ldwp, Y, ldy, M, N, K, (K + 7) / 8);
    if (rc != QT_OK) return rc > 0 ? QT_OK : rc;
    if (K >= (1 << 24)) return QT_ERR_UNSUPPORTED;  // fp32-exact bound of the accumulator
    return dispatch_gemm<ElemFp4>(variant, Xn, ldxp, Wn, ldwp, bias, 1.0f, nullptr, Y, ldy, M, N, K, stream);
}

int qt_nib_gemm_describe(int64_t M, int64_t N, int64_t K, int64_t ldxp, int64_t ldwp, char* out, int cap) {
    // the tile configuration dispatch_gemm's automatic rule (variant 0) launches for this shape: "<kernel><element, tile, pipeline>"
    if (!out || cap < 2 || M <= 0 || N <= 0 || K < 0) return QT_ERR_INVALID_ARG;
    const bool pipe_ok = !(ldxp & 31) && !(ldwp & 31) && M * ldxp * 4 < (1ll << 31) && N * ldwp * 4 < (1ll << 31);
    const int tn = pick_tile_n_gemm(M, N);
    const char* cfg;
    if (pipe_ok && M <= 256 && !((ldxp | ldwp) & 127)) cfg = "64x64, 512-byte stages, pipe=1";
    else if (pipe_ok && M <= 512 && !((ldxp | ldwp) & 63)) cfg = "128x64, 256-byte stages, pipe=1";
    else if (pipe_ok && tn == 256) cfg = "256x256, pipe=2 (ping-pong)";
    else if (pipe_ok && tn == 192) cfg = prefer_384_rows(M, N) ? "384x192, pipe=2 (ping-pong)" : "256x192, pipe=2 (ping-pong)";
    else if (pipe_ok && tn == 128) cfg = "256x128, pipe=2 (ping-pong)";
    else if (pipe_ok) cfg = "256x64, pipe=1";
    else cfg = tn == 256 ? "256x256, pipe=0" : tn == 192 ? "256x192, pipe=0" : tn == 128 ? "256x128, pipe=0" : "256x64, pipe=0";
    snprintf(out, (size_t)cap, "mfma_gemm_kernel<ElemFp4, %s>", cfg);
    return QT_OK;
}

int qt_nib_gemm(const uint32_t* Xn, int64_t ldxp, const uint32_t* Wn, int64_t ldwp, const float* bias,
                float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, qt_stream_t stream) {
    return qt_nib_gemm_variant(0, Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
}

int qt_bf16_gemm(const uint32_t* Xh, int64_t ldxp, const uint32_t* Wh, int64_t ldwp, const float* bias,
                 float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, qt_stream_t stream) {
    const int rc = check_common(Xh, ldxp, Wh, ldwp, Y, ldy, M, N, K, (K + 1) / 2);
    if (rc != QT_OK) return rc > 0 ? QT_OK : rc;
    return dispatch_gemm<ElemBf16>(0, Xh, ldxp, Wh, ldwp, bias, 1.0f, nullptr, Y, ldy, M, N, K, stream);
}

int qt_f16_gemm(const uint32_t* Xh, int64_t ldxp, const uint32_t* Wh, int64_t ldwp, const float* bias, float scale,
                const float* scale_dev, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, qt_stream_t stream) {
    const int rc = check_common(Xh, ldxp, Wh, ldwp, Y, ldy, M, N, K, (K + 1) / 2);
    if (rc != QT_OK) return rc > 0 ? QT_OK : rc;
    return dispatch_gemm<ElemF16>(0, Xh, ldxp, Wh, ldwp, bias, scale, scale_dev, Y, ldy, M, N, K, stream);
}

int qt_bf16_gemm_taps(const uint32_t* Xh, int64_t ldxp, const uint32_t* Wh, int64_t ldwp, float* Y, int64_t ldy,
                      int64_t M, int64_t N, int64_t K, int64_t tap_rows, int64_t tap_cols, int64_t nslice,
                      int64_t w_copy_bytes, int64_t w_row_bytes, int64_t y_stride, qt_stream_t stream) {
    const int rc = check_common(Xh, ldxp, Wh, ldwp, Y, ldy, M, N, K, 0);
    if (rc != QT_OK) return rc > 0 ? QT_OK : rc;
    if (tap_rows < 1 || tap_cols < 1 || nslice < 1 || tap_rows * tap_cols * nslice > 65535) return QT_ERR_INVALID_ARG;
    if (K <= 0 || (K & 31)) return QT_ERR_ALIGNMENT;                       // whole 64-byte stages per slice
    if ((w_copy_bytes | w_row_bytes) & 15) return QT_ERR_ALIGNMENT;        // every tap's W base stays 16-byte aligned
    if (y_stride < M * ldy || (ldy & 3) || !qt_aligned16(Y) || (y_stride & 3)) return QT_ERR_ALIGNMENT;
    if ((ldxp & 31) || (ldwp & 31) || M * ldxp * 4 >= (1ll << 31) || N * ldwp * 4 >= (1ll << 31)) return QT_ERR_UNSUPPORTED;
    ConvArgs cg{};
    cg.H = (int)(tap_rows * tap_cols);
    cg.z_nslice = (int)nslice;
    cg.z_kw = (int)tap_cols;
    cg.z_kslice_bytes = K * 2;
    cg.z_w_copy_bytes = w_copy_bytes;
    cg.z_w_row_bytes = w_row_bytes;
    cg.z_y_stride = y_stride;
#define QT_GOZ(...) return launch_cfg<__VA_ARGS__>(Xh, ldxp, Wh, ldwp, nullptr, 1.0f, nullptr, Y, ldy, M, N, K, stream, cg)
    const int tn = pick_tile_n(N);
    if (tn == 256) QT_GOZ(PP256<ElemBf16>);
    if (tn == 192) {
        if ((M + 383) / 384 * 384 <= (M + 255) / 256 * 256) QT_GOZ(PP384x192<ElemBf16>);
        QT_GOZ(PP192<ElemBf16>);
    }
    if (tn == 128) QT_GOZ(PP128<ElemBf16>);
    QT_GOZ(PP64<ElemBf16>);
#undef QT_GOZ
}

int qt_i8_gemm(const uint32_t* Xc, int64_t ldxp, const uint32_t* Wc, int64_t ldwp, const float* bias,
               float scale, const float* scale_dev, int64_t max_abs_code, float* Y, int64_t ldy, int64_t M,
               int64_t N, int64_t K, qt_stream_t stream) {
    const int rc = check_common(Xc, ldxp, Wc, ldwp, Y, ldy, M, N, K, (K + 3) / 4);
    if (rc != QT_OK) return rc > 0 ? QT_OK : rc;
    // max_abs_code bounds |x code * w code| (127 for +-1/0 weight codes, up to 127*127 for k-bit weight codes):
    // |sum| <= max_abs_code * K must fit the int32 accumulator; below 2^24 the int32 -> fp32 conversion in
    // the epilogue is exact as well (the +-1/0 weight case), above it rounds once (relative 6e-8)
    if (max_abs_code < 0 || max_abs_code > 127 * 127 || max_abs_code * K >= (1ll << 31)) return QT_ERR_UNSUPPORTED;
    return dispatch_gemm<ElemI8>(0, Xc, ldxp, Wc, ldwp, bias, scale, scale_dev, Y, ldy, M, N, K, stream);
}

// Split-K form of qt_i8_gemm for skinny problems (few row tiles, long K: the digit-plane GEMM of LinearXNOR at batch 256):
// blockIdx.y = K slice, slice z contracts bytes [z * kslice, (z + 1) * kslice) of every row and writes its exact integer partial
// sums (as fp32, no scale, no bias) to Y + z * y_stride — 256 x 256 tiles keep the operand bytes per MAC low, the slices fill the
// CUs the few tiles leave idle.  The planes must hold nslice * kslice bytes per row (zero padded).
int qt_i8_gemm_splitk(const uint32_t* Xc, int64_t ldxp, const uint32_t* Wc, int64_t ldwp, float* Y, int64_t ldy, int64_t M, int64_t N,
                      int64_t kslice, int64_t nslice, int64_t y_stride, qt_stream_t stream) {
    const int rc = check_common(Xc, ldxp, Wc, ldwp, Y, ldy, M, N, kslice * nslice, (kslice * nslice + 3) / 4);
    if (rc != QT_OK) return rc > 0 ? QT_OK : rc;
    if (nslice < 1 || nslice > 65535 || kslice <= 0 || (kslice & 63)) return QT_ERR_ALIGNMENT;          // whole 64-byte stages per slice
    if (127 * kslice * nslice >= (1ll << 24)) return QT_ERR_UNSUPPORTED;                                 // partial sums exact in fp32
    if (y_stride < M * ldy || (ldy & 3) || !qt_aligned16(Y) || (y_stride & 3)) return QT_ERR_ALIGNMENT;
    if ((ldxp & 31) || (ldwp & 31) || M * ldxp * 4 >= (1ll << 31) || N * ldwp * 4 >= (1ll << 31)) return QT_ERR_UNSUPPORTED;
    ConvArgs cg{};
    cg.H = 1;
    cg.z_nslice = (int)nslice;
    cg.z_kw = 1;
    cg.z_kslice_bytes = kslice;
    cg.z_y_stride = y_stride;
#define QT_GOZ(...) return launch_cfg<__VA_ARGS__>(Xc, ldxp, Wc, ldwp, nullptr, 1.0f, nullptr, Y, ldy, M, N, kslice, stream, cg)
    const int tn = pick_tile_n(N);
    if (tn == 256) QT_GOZ(PP256<ElemI8>);
    if (tn == 192) QT_GOZ(PP192<ElemI8>);
    if (tn == 128) QT_GOZ(PP128<ElemI8>);
    QT_GOZ(PP64<ElemI8>);
#undef QT_GOZ
}

// conv kernel variant (an ARGUMENT of qt_conv2d_implicit_variant; every other entry point passes 0): 0 = automatic
// (192-wide tiles: ping-pong on a 384x192 tile, whose 96x96 wave tiles keep the load segment under the compute segment:
// AlexNet conv2 302 -> 275 us; other widths: double-buffered, equal or faster there), 1 = double-buffered, 2 = ping-pong,
// 4 = automatic without the un-padded fast path; 3 = stamped 384x192 ping-pong, only in -DQT_PROFILING_VARIANTS builds

// elem: 0 = fp4 nibble planes, 1 = int8 code planes, 2 = bf16 (triple) planes, 3 = fp16 (pair) planes.  epi.alpha != nullptr:
// Y is the threshold-bit plane and ldy its row stride in words.
static int conv_implicit_impl(int elem, const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw,
                              int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh,
                              int64_t dw, const uint32_t* Wmat, int64_t ldwp, const float* bias, float scale,
                              const float* scale_dev, float* Y, int64_t ldy, int64_t Cout, qt_stream_t stream,
                              const EpiArgs& epi_in, int64_t hy = 0, int64_t hx = 0, int g_conv_force = 0) {
    // (hy, hx): halo of the INPUT plane, [N][H + 2hy][W + 2hx][Cw] with a zero border: a conv whose padding fits in
    // the halo runs as the un-padded conv on the window that starts (hy - ph, hx - pw) into the plane.
    EpiArgs epi = epi_in;
#ifdef QT_EXPERIMENT   // A/B builds only (make EXTRA=-DQT_EXPERIMENT): a variant for the entry points that take none
    if (g_conv_force == 0 && getenv("QT_CONV_FORCE_EXP")) g_conv_force = atoi(getenv("QT_CONV_FORCE_EXP"));
#endif
    ConvArgs cg;
    bool valid;
    int64_t M, K, kwords;
    {
        const int rc = conv_prepare(elem, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, Y, ldy, Cout, epi, hy, hx,
                                    cg, valid, M, K, kwords);
        if (rc != QT_OK) return rc > 0 ? QT_OK : rc;
    }
#ifdef QT_PROFILING_VARIANTS
#define QT_CONV_STAMPS_V(E) if (valid && g_conv_force == 3 && tn == 192 && !epi.alpha) \
        return launch_cfg<ConvVPP192Stamps<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi);
#define QT_CONV_STAMPS(E) if (g_conv_force == 3 && tn == 192 && !epi.alpha) \
        return launch_cfg<ConvPP192Stamps<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi);
#else
#define QT_CONV_STAMPS_V(E)
#define QT_CONV_STAMPS(E)
    if (g_conv_force == 3) return QT_ERR_UNSUPPORTED;
#endif
    // ring of 3 / 4 stage buffers on the small-map tiles (ConvV128x128D / ConvV128x64D); QT_NO_CONV_DEEP_RING=1: the double-buffered
    // configurations of round 4 (A/B runs and the bit-identity test; read per call like the direct kernel's switches)
    const bool deep_ring = !getenv("QT_NO_CONV_DEEP_RING");
    // fewer 256-row tiles than this: 128 x 128 tiles with the deep ring (QT_SMALL_GRID: A/B runs)
    const long long small_grid = getenv("QT_SMALL_GRID") ? atoll(getenv("QT_SMALL_GRID")) : 128;
    const long long small_tiles = getenv("QT_SMALL_TILES") ? atoll(getenv("QT_SMALL_TILES")) : 512;
    // weights-as-rows threshold epilogue (sign-bit form): fp4, integer thresholds, whole 32-channel blocks, bit plane or nibble plane
    // out, no depth-to-space; QT_NO_SWAPT=1: the compare form (A/B runs and the bit-identity test; read per call like the line above)
    const bool swapt = elem == 0 && epi.alpha && epi.thr && (Cout & 31) == 0 && !epi.d2s_cout && (epi.mode == 0 || epi.mode == 3) &&
                       !getenv("QT_NO_SWAPT");
#define QT_CONV(E)                                                                                              \
    do {                                                                                                        \
        const int tn = pick_tile_n(Cout);                                                                       \
        QT_CONV_STAMPS_V(E)                                                                                     \
        if (valid && g_conv_force != 4 && g_conv_force != 3) {                                                  \
            if (g_conv_force == 0 && kwords * 4 >= 2048 && !(ldwp & 127) && !epi.d2s_cout &&                  \
                (M <= 4096 || (((M + 255) / 256) * ((Cout + tn - 1) / tn) < 256 &&                              \
                               (M / 64) * Cout * kwords * 4 <= (256ll << 20)))) {                               \
                if (M > 4096 && ((M + 127) / 128) * ((Cout + 127) / 128) >= 200) {                             \
                    if (deep_ring) return launch_cfg_t<ConvV128x128D, E>(swapt, P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
                    return launch_cfg<ConvV128x128<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
                }                                                                                               \
                if (((M + 127) / 128) * ((Cout + 63) / 64) >= 200) { /* 512 ch @ 4x4: 128x64 tiles, 256-byte stages */ \
                    if (deep_ring) return launch_cfg<ConvV128x64D<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
                    return launch_cfg<ConvV128x64<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
                }                                                                                               \
                return launch_cfg<ConvVSkinny<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            }                                                                                                   \
            /* a handful of K stages: a tile is all prologue + epilogue, so 2 co-resident 256x128 workgroups per CU */ \
            /* that overlap each other's beat the 1-per-CU ping-pong tiles (output-blocked first layers: K = 320 B) */ \
            if (g_conv_force == 0 && tn == 256 && kwords * 4 <= 1024)                                           \
                return launch_cfg_t<ConvV128x2, E>(swapt, P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            /* a few big tiles on a small map (128 -> 256 stride 2 @ 16x16, K = 1152 B: 64 tiles of 256x256): 128x128 tiles */ \
            /* with the deep ring give every CU one                                                                        */ \
            if (g_conv_force == 0 && deep_ring && !epi.d2s_cout && ((M + 255) / 256) * ((Cout + tn - 1) / tn) <= small_grid && \
                ((M + 127) / 128) * ((Cout + 127) / 128) >= 200 && ((M + 127) / 128) * ((Cout + 127) / 128) <= small_tiles) \
                return launch_cfg_t<ConvV128x128D, E>(swapt, P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            if (g_conv_force != 1) {                                                                            \
                if (tn == 192 && (g_conv_force == 2 || prefer_384_rows(M, Cout)))                               \
                    return launch_cfg_t<ConvVPP192, E>(swapt, P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
                if (tn == 256) return launch_cfg_t<ConvVPP256, E>(swapt, P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
                /* 192-wide tiles whose 384-row form wastes a round (576 -> 1152 @ 13x13): 256x192 ping-pong for long K */ \
                if (tn == 192 && g_conv_force == 0 && kwords * 4 >= 2048)                                         \
                    return launch_cfg_t<ConvVPP256x192, E>(swapt, P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            }                                                                                                   \
            if (g_conv_force == 0 && tn == 64 && kwords * 4 <= 1024)                                            \
                return launch_cfg<ConvV64x2<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            if (g_conv_force == 0 && tn == 128 && kwords * 4 <= 1024)                                           \
                return launch_cfg_t<ConvV128x2, E>(swapt, P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            if (tn == 256) return launch_cfg<ConvV256<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            if (tn == 192) return launch_cfg<ConvV192<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            if (tn == 128) return launch_cfg<ConvV128<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            return launch_cfg<ConvV64<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi);                 \
        }                                                                                                       \
        QT_CONV_STAMPS(E)                                                                                       \
        if (g_conv_force == 5 && !epi.alpha && epi.mode == 0)                                                   \
            return launch_cfg<Conv128x128<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
        if (g_conv_force == 6 && !epi.alpha && epi.mode == 0 && kwords * 4 >= 2048 && !(ldwp & 127))            \
            return launch_cfg<ConvSkinny<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
        if (g_conv_force == 0 && !epi.alpha && epi.mode == 0 && ((M + 255) / 256) * ((Cout + tn - 1) / tn) < 200) {     \
            /* small maps: fewer 256-row tiles than CUs (tools/bench_conv_small_maps.py: 256 ch @ 8x8 137 -> 83 us,  */ \
            /* 512 ch @ 4x4 239 -> 102 us incl. the operand split; same accumulation order, bit-identical results)   */ \
            if (((M + 127) / 128) * ((Cout + 127) / 128) < 200 && kwords * 4 >= 2048 && !(ldwp & 127))          \
                return launch_cfg<ConvSkinny<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            return launch_cfg<Conv128x128<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
        }                                                                                                       \
        if (g_conv_force == 0 && tn == 192 && prefer_384_rows(M, Cout))                                         \
            return launch_cfg<ConvPP192<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
        /* long K (>= 2 KiB per output row), wide tiles: the ping-pong main loop beats the double-buffered one on the   */ \
        /* padded convs too (tools/bench_grad_input_variants.py: grad_x 512 ch @ 28x28 0.532 -> 0.485 ms, 768 -> 1152   */ \
        /* @ 13x13 1.34 -> 1.25); the 384-row tile only where its rounds pay (above), else 256 rows                     */ \
        if (g_conv_force == 0 && kwords * 4 >= 2048) {                                                          \
            if (tn == 256) return launch_cfg<ConvPP256<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            if (tn == 192) return launch_cfg<ConvPP256x192<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
        }                                                                                                       \
        if (g_conv_force == 2) {                                                                                \
            if (tn == 256) return launch_cfg<ConvPP256<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            if (tn == 192) return launch_cfg<ConvPP192<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            if (tn == 128) return launch_cfg<ConvPP128<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
            return launch_cfg<ConvPP64<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi);                 \
        }                                                                                                       \
        if (tn == 256) return launch_cfg<Conv256<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
        if (tn == 192) return launch_cfg<Conv192<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
        if (tn == 128) return launch_cfg<Conv128<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi); \
        return launch_cfg<Conv64<E>>(P, 0, Wmat, ldwp, bias, scale, scale_dev, Y, ldy, M, Cout, K, stream, cg, epi);                 \
    } while (0)
    if (elem == 0) QT_CONV(ElemFp4);
    if (elem == 1) QT_CONV(ElemI8);
    if (elem == 3) QT_CONV(ElemF16);
    QT_CONV(ElemBf16);
#undef QT_CONV
#undef QT_CONV_STAMPS
#undef QT_CONV_STAMPS_V
}

int qt_conv2d_implicit_variant(int variant, int elem, const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw,
                               int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh,
                               int64_t dw, const uint32_t* Wmat, int64_t ldwp, const float* bias, float scale,
                               const float* scale_dev, float* Y, int64_t ldy, int64_t Cout, qt_stream_t stream) {
    if (variant < 0 || variant > 6) return QT_ERR_INVALID_ARG;
    return conv_implicit_impl(elem, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, bias, scale,
                              scale_dev, Y, ldy, Cout, stream, EpiArgs{}, 0, 0, variant);
}

int qt_conv2d_implicit(int elem, const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw,
                       int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh,
                       int64_t dw, const uint32_t* Wmat, int64_t ldwp, const float* bias, float scale,
                       const float* scale_dev, float* Y, int64_t ldy, int64_t Cout, qt_stream_t stream) {
    return conv_implicit_impl(elem, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, bias, scale,
                              scale_dev, Y, ldy, Cout, stream, EpiArgs{});
}

int qt_conv2d_implicit_bits(int elem, const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw,
                            int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh,
                            int64_t dw, const uint32_t* Wmat, int64_t ldwp, const float* bias, float scale,
                            const float* scale_dev, const float* alpha, const float* beta, const float* thr,
                            uint32_t* neg_plane, int64_t ldb, int64_t Cout, qt_stream_t stream) {
    if (!alpha || !beta) return QT_ERR_INVALID_ARG;
    if (thr && elem >= 2) return QT_ERR_INVALID_ARG;       // integer thresholds need integer accumulators
    if (ldb & 3) return QT_ERR_ALIGNMENT;
    EpiArgs epi;
    epi.alpha = alpha;
    epi.beta = beta;
    epi.thr = thr;
    return conv_implicit_impl(elem, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, bias, scale,
                              scale_dev, reinterpret_cast<float*>(neg_plane), ldb, Cout, stream, epi);
}

int qt_conv2d_implicit_nib(int elem, const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw,
                           int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh,
                           int64_t dw, const uint32_t* Wmat, int64_t ldwp, const float* bias, float scale,
                           const float* scale_dev, const float* alpha, const float* beta, const float* thr,
                           uint32_t* nib_plane, int64_t ldn, int64_t Cout, int64_t out_halo_h, int64_t out_halo_w,
                           int64_t d2s_cout, qt_stream_t stream) {
    if (!alpha || !beta) return QT_ERR_INVALID_ARG;
    if (thr && elem >= 2) return QT_ERR_INVALID_ARG;       // integer thresholds need integer accumulators
    if (out_halo_h < 0 || out_halo_w < 0 || out_halo_h > 64 || out_halo_w > 64) return QT_ERR_INVALID_ARG;
    if ((ldn & 3) || !qt_aligned16(nib_plane)) return QT_ERR_ALIGNMENT;
    if (d2s_cout < 0 || (d2s_cout && (d2s_cout % 32 || Cout != 4 * d2s_cout))) return QT_ERR_INVALID_ARG;
    // every word of a pixel is written by a column block
    if (ldn != ((d2s_cout ? d2s_cout : Cout) + 31) / 32 * 4) return QT_ERR_INVALID_ARG;
    EpiArgs epi;
    epi.alpha = alpha;
    epi.beta = beta;
    epi.mode = 3;
    epi.thr = thr;
    epi.ohy = (int)out_halo_h;
    epi.ohx = (int)out_halo_w;
    epi.d2s_cout = (int)d2s_cout;
    return conv_implicit_impl(elem, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, bias, scale,
                              scale_dev, reinterpret_cast<float*>(nib_plane), ldn, Cout, stream, epi);
}

int qt_conv2d_implicit_codes(int elem, const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw,
                             int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh,
                             int64_t dw, const uint32_t* Wmat, int64_t ldwp, const float* bias, float scale,
                             const float* scale_dev, const float* alpha, const float* beta, const float* res_f32,
                             int64_t ldr, const float* res_alpha, const float* res_beta, const int8_t* res_codes,
                             int64_t ldrc_bytes, float res_scale, int relu, int bit_width, int8_t* codes,
                             int64_t ldc_bytes, int64_t Cout, int32_t* overflow, int64_t in_halo_h,
                             int64_t in_halo_w, int64_t out_halo_h, int64_t out_halo_w, int64_t res_halo_h,
                             int64_t res_halo_w, const float* bn_stats, const float* res_bn_stats, qt_stream_t stream) {
    if (res_bn_stats || (bn_stats && res_alpha)) return QT_ERR_UNSUPPORTED;   // device form: the residual arrives normalised
    if (!alpha || !beta || !overflow || bit_width < 2 || bit_width > 8 || relu < 0 || relu > 2) return QT_ERR_INVALID_ARG;
    if (out_halo_h < 0 || out_halo_w < 0 || res_halo_h < 0 || res_halo_w < 0 || out_halo_h > 64 || out_halo_w > 64 ||
        res_halo_h > 64 || res_halo_w > 64 || ((res_halo_h | res_halo_w) && !res_codes))
        return QT_ERR_INVALID_ARG;
    if (elem != 1) return QT_ERR_UNSUPPORTED;   // the epilogue is instantiated for the int8 (DoReFa) configs
    if ((res_f32 && ldr < Cout) || (!res_alpha != !res_beta) || (res_alpha && !res_f32)) return QT_ERR_INVALID_ARG;
    if ((ldc_bytes & 15) || !qt_aligned16(codes)) return QT_ERR_ALIGNMENT;
    if (res_codes && (ldrc_bytes < ((Cout + 3) & ~3ll) || (ldrc_bytes & 3) || (reinterpret_cast<uintptr_t>(res_codes) & 3)))
        return QT_ERR_ALIGNMENT;
    EpiArgs epi;
    epi.alpha = alpha;
    epi.beta = beta;
    epi.mode = 2;
    epi.relu = relu;
    epi.levels = (float)((1 << bit_width) - 1);
    epi.res_f32 = res_f32;
    epi.ldr = ldr;
    epi.ralpha = res_alpha;
    epi.rbeta = res_beta;
    epi.res_codes = res_codes;
    epi.ldrc = ldrc_bytes;
    epi.rscale = res_scale;
    epi.overflow = overflow;
    epi.ohy = (int)out_halo_h;
    epi.ohx = (int)out_halo_w;
    epi.rhy = (int)res_halo_h;
    epi.rhx = (int)res_halo_w;
    epi.bn_stats = bn_stats;
    {   // small-channel 3 x 3 / stride 1 layers: the persistent direct kernel (code_conv3x3.hip), bit-identical codes
        const int rc = qt_code_conv3x3_try(P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, bias, scale, scale_dev, alpha,
                                           beta, res_f32, res_alpha, res_codes, ldrc_bytes, res_scale, relu, bit_width, codes, ldc_bytes,
                                           Cout, overflow, in_halo_h, in_halo_w, out_halo_h, out_halo_w, res_halo_h, res_halo_w, bn_stats,
                                           stream);
        if (rc != QT_ERR_UNSUPPORTED) return rc;
    }
    return conv_implicit_impl(elem, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, bias, scale,
                              scale_dev, reinterpret_cast<float*>(codes), ldc_bytes, Cout, stream, epi, in_halo_h,
                              in_halo_w);
}

int qt_conv2d_implicit_halo(int elem, const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw,
                            int64_t halo_h, int64_t halo_w, int64_t kh, int64_t kw, int64_t sh, int64_t sw,
                            int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldwp,
                            const float* bias, float scale, const float* scale_dev, float* Y, int64_t ldy,
                            int64_t Cout, qt_stream_t stream) {
    return conv_implicit_impl(elem, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, bias, scale,
                              scale_dev, Y, ldy, Cout, stream, EpiArgs{}, halo_h, halo_w);
}

int qt_conv2d_implicit_halo_bn(int elem, const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw,
                               int64_t halo_h, int64_t halo_w, int64_t kh, int64_t kw, int64_t sh, int64_t sw,
                               int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldwp,
                               const float* bias, float scale, const float* scale_dev, const float* bn_weight,
                               const float* bn_bias, const float* bn_stats, float* Y, int64_t ldy, int64_t Cout,
                               qt_stream_t stream) {
    if (!bn_weight || !bn_bias || !bn_stats || ldy < Cout) return QT_ERR_INVALID_ARG;
    if (elem != 1) return QT_ERR_UNSUPPORTED;               // instantiated for the int8 (DoReFa) configurations
    if ((Cout & 3) || (ldy & 3) || !qt_aligned16(Y) || !qt_aligned16(bn_weight) || !qt_aligned16(bn_bias) || !qt_aligned16(bn_stats))
        return QT_ERR_ALIGNMENT;
    EpiArgs epi;
    epi.alpha = bn_weight;
    epi.beta = bn_bias;
    epi.bn_stats = bn_stats;
    epi.mode = 4;
    return conv_implicit_impl(elem, P, Nimg, H, W, Cw, kh, kw, sh, sw, ph, pw, dh, dw, Wmat, ldwp, bias, scale,
                              scale_dev, Y, ldy, Cout, stream, epi, halo_h, halo_w);
}

int qt_bits_to_nib(const uint32_t* sign_plane, const uint32_t* mask_plane, int64_t ldb,
                   uint32_t* nib_plane, int64_t ldn, int64_t rows, int64_t K, qt_stream_t stream) {
    if (rows < 0 || K < 0) return QT_ERR_INVALID_ARG;
    if (rows == 0) return QT_OK;
    if (!nib_plane || (K > 0 && !sign_plane)) return QT_ERR_INVALID_ARG;
    if (ldb < (K + 31) / 32 || ldn < (K + 7) / 8) return QT_ERR_INVALID_ARG;
    if ((ldb & 3) || (ldn & 3) || !qt_aligned16(nib_plane)) return QT_ERR_ALIGNMENT;
    const int grid = qt_stream_grid((rows * (ldn / 4) + 255) / 256);
    hipLaunchKernelGGL(bits_to_nib_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, sign_plane,
                       mask_plane, ldb, nib_plane, ldn, rows, K);
    return qt_check_launch();
}

int qt_bits_to_nib_pad(const uint32_t* sign_plane, const uint32_t* mask_plane, int64_t ldb, uint32_t* nib_plane,
                       int64_t ldn, int64_t N, int64_t H, int64_t W, int64_t ph, int64_t pw, int64_t K,
                       qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || ph < 0 || pw < 0 || K < 0) return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!nib_plane || (K > 0 && !sign_plane)) return QT_ERR_INVALID_ARG;
    if (ldb < (K + 31) / 32 || ldn < (K + 7) / 8) return QT_ERR_INVALID_ARG;
    if ((ldb & 3) || (ldn & 3) || !qt_aligned16(nib_plane)) return QT_ERR_ALIGNMENT;
    if (H + 2 * ph > 32767 || W + 2 * pw > 32767) return QT_ERR_UNSUPPORTED;
    const int64_t total = N * (H + 2 * ph) * (W + 2 * pw) * (ldn / 4);
    const int grid = qt_stream_grid((total + 255) / 256);
    hipLaunchKernelGGL(bits_to_nib_pad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, sign_plane, mask_plane,
                       ldb, nib_plane, ldn, N, (int)H, (int)W, (int)ph, (int)pw, K);
    return qt_check_launch();
}

}  // extern "C"
