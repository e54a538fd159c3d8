// The matrix-core GEMM / implicit-conv kernel template, its tile configurations and launcher (see mfma_gemm.hip for the
// design notes).  A header so that more than one translation unit can instantiate configurations: mfma_gemm.hip (the
// GEMM and conv entry points) and conv_taps.hip (the per-tap-scaled convs of the XNOR-Net family) compile in parallel.
#pragma once
#include <cstdlib>
#include <type_traits>
#include "qt_common.h"
#include "pp_common.h"

namespace {

// 16 zero bytes in global memory: the source of every DMA chunk that lies past a row's stride.
__device__ __attribute__((aligned(16))) const unsigned char zero16_storage[16] = {0};


// Implicit-GEMM conv: the X operand is not a matrix but the NHWC pixel plane P[N][H][W][cpp*16 bytes];
// row m = (n, ho, wo), K byte index = ((i*kw + j)*cpp + sub)*16 + byte: chunk q of a row is 16 bytes of
// pixel (ho*sh - ph + i*dh, wo*sw - pw + j*dw) or zeros when that pixel is padding / q is past the taps.
struct ConvArgs {
    int H, W, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw;
    int cpp;                       // 16-byte chunks per pixel
    int kbytes;                    // K bytes per (virtual) im2col row
    unsigned magic_cpp, magic_kw;  // floor(2^32/d)+1: q/d == __umulhi(q, magic) for q < 2^16 (d > 1)
    // log2(Ho * Wo), log2(Wo) when both are powers of two (-1 otherwise): output row m -> (image, ho, wo) by shifts and masks
    // instead of integer divisions (prologue) / 64-bit magic multiplies (halo-plane epilogues) — quarter-rate VALU work that
    // bounds the short small-N launches of the 32 x 32 .. 4 x 4 maps (profiles/r5_c4_pmc.md)
    int sh_hw = -1, sh_w = -1;
    // GEMM mode, bf16 (qt_bf16_gemm_taps) and int8 (qt_i8_gemm_splitk: one tap, K slices only): a launch of blockIdx.y = tap * z_nslice + slice problems of one shape —
    // X and W advance by z_kslice_bytes per slice along K, W additionally by the tap's offset (tap = row * z_kw + col:
    // col * z_w_copy_bytes + row * z_w_row_bytes), Y by z_y_stride floats per problem.  z_nslice == 0: a plain launch.
    int z_nslice = 0, z_kw = 1;
    long long z_kslice_bytes = 0, z_w_copy_bytes = 0, z_w_row_bytes = 0, z_y_stride = 0;
};

// Threshold-bit epilogue (inference fusion of conv -> [MaxPool] -> BatchNorm(eval) -> Hardtanh -> sign):
// when alpha != nullptr the kernel does not store fp32 Y but, per output element,
//     t = out(acc) (+ bias);  v = fl(fl(t * alpha[n]) + beta[n]);  bit = v < 0
// into the bit plane (uint32_t*)Y with ldy WORDS per row (bit n%32 of word n/32 of row m).
//
// Code epilogue (mode == 2; inference fusion of conv -> BatchNorm(eval) [-> + shortcut] -> ReLU -> nnDorefaQuant(k),
// the chain qt_affine_dorefa_codes_i8 runs on an fp32 tensor, applied to the accumulators instead):
//     t = fl(fl(out(acc) * alpha[n]) + beta[n]) [+ fl(fl(r*ralpha[n]) + rbeta[n]) | + fl(rscale * rcode)]
//     t = max(t, 0) if relu;  q = rint(levels * t)  ->  int8 code plane (int8_t*)Y with ldy BYTES per row
// (pad bytes zero; |q| > 127 or NaN -> code 0 and *overflow |= 1).
struct EpiArgs {
    const float* alpha = nullptr;
    const float* beta = nullptr;
    int mode = 0;                      // 0: threshold bits iff alpha != nullptr ; 2: int8 codes ; 3: threshold bits
                                       // expanded to the NEXT conv's fp4 nibble pixel plane (+1 = 0x2, -1 = 0xA), ldy WORDS per
                                       // pixel, optionally with an (ohy, ohx) zero halo (border zeroed by the caller)
    int relu = 0;
    float levels = 0.0f;               // 2^k - 1
    float rscale = 0.0f;
    const float* res_f32 = nullptr;    // [M][ldr] fp32 residual, optionally through its own (ralpha, rbeta)
    const float* ralpha = nullptr;
    const float* rbeta = nullptr;
    const int8_t* res_codes = nullptr; // [M][ldrc bytes] residual held as codes
    int64_t ldr = 0, ldrc = 0;
    int32_t* overflow = nullptr;
    // halo planes: the code plane (and a residual code plane) may carry a zero border of (hy, hx) pixels around
    // each image, [N][Ho + 2hy][Wo + 2hx][ld]: output row m = (img, ho, wo) lands on pixel
    // (img, ho + hy, wo + hx) so the NEXT conv's zero padding is physical and it runs the un-padded kernels.  The border
    // is zeroed by 64 extra workgroups appended to the launch (zero_halo_border); writing it from the edge pixels' lanes
    // was measured at +10 us per conv of the C4 ResNet (divergent short loops in an already VALU-heavy epilogue).
    int ohy = 0, ohx = 0, rhy = 0, rhx = 0;
    // mode 3 only: depth-to-space by 2.  The N = 4*d2s_cout output columns are (dy, dx, channel): column block nb of
    // output row (img, ho, wo) belongs to pixel (img, 2*ho + dy, 2*wo + dx) of a [2*Ho][2*Wo] image with d2s_cout
    // channels — the 2x2 output-blocked form of a few-channel stride-1 first layer (a 4x4 stride-2 conv that embeds
    // the four shifted copies of the 3x3 kernel), which gathers a quarter of the bytes of the direct form.
    int d2s_cout = 0;
    // threshold modes (0 / 3) with EXACT INTEGER accumulators (+-1 / 0 operands): thr[c] = the integer T_c with
    //   fl(fl(acc + bias_c) * alpha_c) + beta_c < 0   <=>   (acc < T_c) xor (alpha_c < 0)      for every |acc| <= K
    // (the left side is a monotone step function of the integer acc; the caller finds T_c by bisection, once per layer:
    // ops.integer_thresholds).  One compare per accumulator register instead of add + multiply + compare — the
    // "BatchNorm + sign collapses to a per-channel integer threshold on the popcount" of SURVEY 8f n1.
    const float* thr = nullptr;
    // code epilogue with the DEVICE's BatchNorm arithmetic: bn_stats = [mean | rs] (alpha / beta then hold weight / bias):
    //   t = fma(fl(fl(x - mean) * rs), weight, bias)      what eval-mode F.batch_norm evaluates on this device (an fp32 residual
    //   that has its own BatchNorm arrives already normalised: the caller applies F.batch_norm itself)
    const float* bn_stats = nullptr;
    unsigned long long magic_hw = 0, magic_w = 0;   // ceil(2^64 / (Ho*Wo)), ceil(2^64 / Wo) (0: divisor 1)
    // per-tap scaling (E::TAPS kernels only; see ElemFp4Taps): [1, rho_1 .. rho_{T-1}, alpha_{T-1}], MFMA k-steps per tap, T
    const float* tap_rho = nullptr;
    int tap_ksteps = 0, ntaps = 0;
    // ElemFp4TapsRows only: per-pixel scale plane A [Nimg][aH][aW] of the conv's (logical, un-padded) input and its padding
    const float* row_A = nullptr;
    int aH = 0, aW = 0, aph = 0, apw = 0;
};

// spread the 8 bits of a byte to bit 0 of 8 nibbles
__device__ __forceinline__ uint32_t spread8(uint32_t b) {
    uint32_t t = b & 0xFFu;
    t = (t | (t << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return t;
}

// Zero border of a halo output plane [nimg][Ho + 2hy][Wo + 2hx][cpp 16-byte chunks], done by the surplus workgroups a
// halo-plane launch appends to its grid (they run in the tail of the launch, when CUs idle anyway): a separate 5 us
// launch per conv otherwise (9 % of the fused DoReFa ResNet-18 forward).  Border pixel order: top rows, then the 2*hx side
// pixels of every interior row, then the bottom rows.
__device__ __forceinline__ void zero_halo_border(void* plane, int cpp, int64_t nimg, int H, int W, int hy, int hx,
                                                 int zb, int nzb) {
    const int Hp = H + 2 * hy, Wp = W + 2 * hx;
    const int top = hy * Wp, side = 2 * hx * H, per_img = 2 * top + side;
    const int64_t total = nimg * per_img * cpp;
    uint4* Q = reinterpret_cast<uint4*>(plane);
    for (int64_t t = (int64_t)zb * blockDim.x + threadIdx.x; t < total; t += (int64_t)nzb * blockDim.x) {
        const int64_t bp = t / cpp;
        const int c = (int)(t - bp * cpp);
        const int64_t n = bp / per_img;
        const int b = (int)(bp - n * per_img);
        int pix;
        if (b < top) pix = b;
        else if (b < top + side) {
            const int s2 = b - top, r = s2 / (2 * hx), k = s2 - r * 2 * hx;
            pix = (hy + r) * Wp + (k < hx ? k : W + k);
        } else pix = (hy + H) * Wp + (b - top - side);
        Q[(n * Hp * Wp + pix) * (int64_t)cpp + c] = make_uint4(0, 0, 0, 0);
    }
}

// ---- element types ----------------------------------------------------------------------------------
struct ElemFp4 {
    static constexpr bool TAPS = false;
    using acc_t = v16f;
    static constexpr bool CODE_EPI = false;
    __device__ __forceinline__ static int kbytes(int K) { return (K + 1) / 2; }
    __device__ __forceinline__ static acc_t mfma(const uint4& a, const uint4& b, acc_t c) {
        const v8i av = (v8i){(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
        const v8i bv = (v8i){(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
        // cbsz = blgp = 4: both operands FP4; scales 0x7f = E8M0 for 2^0 on every 32-element block
        return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    __device__ __forceinline__ static float out(float v, float scale, float bias) { return v + bias; }
};
struct ElemI8 {
    static constexpr bool TAPS = false;
    using acc_t = v16i;
    static constexpr bool CODE_EPI = true;   // the DoReFa code epilogue is instantiated for int8 conv configs only
    __device__ __forceinline__ static int kbytes(int K) { return K; }
    __device__ __forceinline__ static acc_t mfma(const uint4& a, const uint4& b, acc_t c) {
        const v4i av = (v4i){(int)a.x, (int)a.y, (int)a.z, (int)a.w};
        const v4i bv = (v4i){(int)b.x, (int)b.y, (int)b.z, (int)b.w};
        return __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
    }
    // exact int32 -> fp32 (|acc| < 2^24 is checked by the launcher), ONE scale multiply, bias once
    __device__ __forceinline__ static float out(int v, float scale, float bias) { return (float)v * scale + bias; }
};

// bf16 operands (the "real-valued activation x quantised weight" path): fp32 activations are split
// EXACTLY into three bf16 terms (x = hi + mid + lo: 3 x 8 significand bits) laid out as consecutive
// triples, weights are +-1/0 replicated three times; v_mfma_f32_32x32x16_bf16 forms exact products and
// accumulates in fp32, so the result has fp32-GEMM accuracy at the bf16 matrix rate / 3.
struct ElemBf16 {
    static constexpr bool TAPS = false;
    using acc_t = v16f;
    static constexpr bool CODE_EPI = false;
    __device__ __forceinline__ static int kbytes(int K) { return 2 * K; }
    __device__ __forceinline__ static acc_t mfma(const uint4& a, const uint4& b, acc_t c) {
        typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
        bf8 av, bv;
        __builtin_memcpy(&av, &a, 16);
        __builtin_memcpy(&bv, &b, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
    }
    __device__ __forceinline__ static float out(float v, float scale, float bias) { return v + bias; }
};

// fp16 operands (round 3, "pair planes"): the same idea with TWO terms.  An fp32 activation divided by a power-of-two
// scale s (per tensor: max|x| / s in [2^14, 2^15), so that nothing overflows fp16 and the low term stays a normal number for
// everything within 2^-17 of the maximum) is hi = fp16(x / s), lo = fp16(x / s - hi): 2 x 11 significand bits, i.e.
// |x - s (hi + lo)| <= max(2^-22 |x|, 2^-39 max|x|).  Weights are +-1 / 0 / small integers (exact in fp16) replicated twice.
// v_mfma_f32_32x32x16_f16 runs at the bf16 rate, products are exact, accumulation is fp32; the epilogue multiplies by s
// (exact).  Two thirds of the MFMA work and of the operand bytes of the three-term bf16 route, at a normalised error two
// orders inside the 1e-5 bar for real-valued inputs ("two-term split").
struct ElemF16 {
    static constexpr bool TAPS = false;
    using acc_t = v16f;
    static constexpr bool CODE_EPI = false;
    __device__ __forceinline__ static int kbytes(int K) { return 2 * K; }
    __device__ __forceinline__ static acc_t mfma(const uint4& a, const uint4& b, acc_t c) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 av, bv;
        __builtin_memcpy(&av, &a, 16);
        __builtin_memcpy(&bv, &b, 16);
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
    }
    __device__ __forceinline__ static float out(float v, float scale, float bias) { return v * scale + bias; }
};

// Per-tap scaled convs (the XNOR-Net family, functions/xnor_connect.py:135-146: conv2d(x, sign(W) * alpha[1, 1, kh, kw])): the K loop of
// the implicit GEMM is tap-major, so  y = sum_t alpha_t D_t  (D_t = the contraction over the channels of tap t — an exact integer
// for +-1 activations) is evaluated in Horner form ON the accumulators:
//     S_t = D_t + (alpha_{t-1} / alpha_t) S_{t-1},      y = alpha_{T-1} S_{T-1}
// i.e. at every tap boundary the accumulator registers are multiplied in place by rho_t = alpha_{t-1} / alpha_t (one VALU multiply
// per register per tap, no second accumulator set) and the MFMAs of tap t keep accumulating onto them; the last factor rides in
// the epilogue's `scale`.  One fp4 pass instead of the six bf16 passes of a real x real conv; 2 (T - t) extra fp32 roundings on
// tap t's term (<= 50 x 2^-24 for a 5 x 5 kernel).  EpiArgs::tap_rho = [1, rho_1 .. rho_{T-1}, alpha_{T-1}] (qt_xnor_tap_table_f32:
// taps whose alpha is 0 inherit their predecessor's), tap_ksteps = 32-byte MFMA k-steps per tap (a tap is a whole number of them).
struct ElemFp4Taps : ElemFp4 {
    static constexpr bool TAPS = true;
    __device__ __forceinline__ static float out(float v, float scale, float bias) { return v * scale + bias; }
};
struct ElemF16Taps : ElemF16 {      // real-valued operand (grad_x of an XNOR conv: split gradient x flipped sign(W), alpha per tap)
    static constexpr bool TAPS = true;
};
// XNORConv2d(quant_input = True) (functions/xnor_connect.py:142-145): the activation is sign(x) * A with A = mean_c |x| PER PIXEL, so
//     y[m, co] = sum_t alpha_t A[pixel(m, t)] D_t[m, co]        (D_t: the integer dot of sign(x) and sign(W) over tap t's channels)
// — the Horner factor of a tap boundary becomes one per accumulator ROW:  F_t[m] = rho_t * a'_{t-1}[m] / a'_t[m]  (a' = A along the
// row's taps with zeros — padding, all-zero pixels: their D_t is 0 — replaced by the previous non-zero entry), a table of
// (T + 1) x TM floats in LDS filled once per tile from the A plane (EpiArgs::row_A); entry 0 holds a'_{T-1}[m] for the epilogue.
// One fp4 pass like the +-1 conv instead of two fp16 passes over the real-valued x_q.
struct ElemFp4TapsRows : ElemFp4Taps {
    static constexpr bool ROWS = true;
};
// Threshold-epilogue convs on exact-integer fp4 accumulators with the WEIGHTS as the MFMA row operand (round 6): E::mfma swaps its
// operands and the weight fragment of MFMA row i is read from LDS row 16 ((i / 4) % 2) + 4 (i / 8) + i % 4 of its 32-row block, so
// accumulator register r of lane (l % 32, l / 32) is (position l % 32, channel 16 (l / 32) + r): a lane owns 16 consecutive channels
// of ONE output pixel.  The accumulators start at -(T_c - 1/2) (T_c = the integer threshold of EpiArgs::thr), so the threshold bit
// is the accumulator's sign bit and the epilogue is one v_alignbit per value — no compare, no cross-lane gather, no nibble spread
// (the v_cmp / v_writelane / spread8 form was 131 VALU instructions per 32 x 32 block against ~25; with one tile per workgroup and one
// workgroup per CU nothing overlaps a tile's epilogue: profiles/r6_threshold_epilogue.md).
struct ElemFp4T : ElemFp4 {
    static constexpr bool SWAPT = true;
    __device__ __forceinline__ static acc_t mfma(const uint4& a, const uint4& b, acc_t c) { return ElemFp4::mfma(b, a, c); }
};
template <class E, class = void> struct elem_swapt : std::false_type {};
template <class E> struct elem_swapt<E, std::void_t<decltype(E::SWAPT)>> : std::bool_constant<E::SWAPT> {};
template <class E, class = void> struct elem_rows : std::false_type {};
template <class E> struct elem_rows<E, std::void_t<decltype(E::ROWS)>> : std::bool_constant<E::ROWS> {};

// Workgroup = WM x WN waves (8 waves); wave tile = (TMW*32) m-rows x (TNW*32) n-rows.
//   ABL (profiling only; results wrong unless 0): 1 = no MFMA, 2 = no DMA, 3 = epilogue only,
//   4 = no LDS fragment reads, 5 = (ping-pong) per-segment cycle stamps + wall-clock phase stamps written over Y, 6 = phase stamps only.
template <class E_, int WM_, int WN_, int TMW_, int TNW_, int PIPE_, int ABL_ = 0, int SB_ = 128, int CONV_ = 0, int OCC_ = 1>
struct GemmCfg {
    using E = E_;
    // CONV_: 0 = GEMM, 1 = implicit conv (zero padding by per-tap bounds checks, 64-bit per-lane addresses),
    // 2 = implicit conv on an un-padded / physically padded plane (every tap in bounds: 32-bit offsets from one base)
    static constexpr bool CONV = CONV_ != 0, VALID = CONV_ == 2;
    static constexpr int WM = WM_, WN = WN_, TMW = TMW_, TNW = TNW_, PIPE = PIPE_, ABL = ABL_;
    static constexpr int STAGE_BYTES = SB_, KK = SB_ / 32;
    static constexpr int ROWS_PER_PIECE = 1024 / SB_;   // one DMA piece = 1 KiB of LDS
    static constexpr int CHUNKS = SB_ / 16;
    static constexpr int NWAVES = WM * WN, NTHREADS = NWAVES * 64;
    static constexpr int TM = WM * TMW * 32, TN = WN * TNW * 32;  // workgroup tile
    static constexpr int X_STAGE = TM * STAGE_BYTES, W_STAGE = TN * STAGE_BYTES;
    static constexpr int BUF = X_STAGE + W_STAGE;
    // stage buffers: double-buffered (PIPE 0 / 1), a ring of 4 (ping-pong, PIPE 2), or a ring of PIPE_ = 3 / 4 buffers walked by the
    // PIPE 1 loop with PIPE_ - 1 stages of DMA in flight (short-K tiles one per CU: a stage's L2 round trip, ~1 us under load, is 4-5x its
    // matrix time, so a double-buffered loop runs at one round trip per stage)
    static constexpr int NBUF = PIPE_ == 2 ? 4 : (PIPE_ >= 3 ? PIPE_ : 2);
    static constexpr int LDS_FIXED = NBUF * BUF + 64;   // stage buffers + the waves' SIMD ids (ping-pong)
    static constexpr int LDS_BYTES = LDS_FIXED;         // VALID conv: + the tap table (launch_cfg adds nstages * CHUNKS * 4)
    static_assert(PIPE_ != 2 || (SB_ == 64 && NWAVES == 8), "ping-pong: 64-byte stages, two waves per SIMD");
    // OCC_ workgroups per CU the register budget is sized for (small-accumulator tiles: 2 workgroups overlap each
    // other's prologue latency, which dominates convs whose K is only a few stages)
    static constexpr int WAVES_PER_SIMD = OCC_ * ((NWAVES + 3) / 4);
    static_assert(TM % (ROWS_PER_PIECE * NWAVES) == 0 && (TN % (ROWS_PER_PIECE * NWAVES) == 0 || PIPE_ >= 1),
                  "DMA pieces divide evenly over the waves (asm-DMA pipelines: W pieces may wrap)");
};

// DEVBN: the code epilogue in the device's BatchNorm arithmetic (EpiArgs::bn_stats) — a separate instantiation (int8 conv
// configurations only) so that the folded-form kernels keep their register budget: with both forms in one kernel the 16 extra
// per-channel registers pushed the 3-workgroups-per-CU tiles from 16 to 88 B of scratch per lane (fused C4 0.78 -> 1.1 ms).
template <class C, bool DEVBN = false>
__global__ __launch_bounds__(C::NTHREADS, C::WAVES_PER_SIMD) void mfma_gemm_kernel(
    const uint32_t* __restrict__ X, int64_t ldx, const uint32_t* __restrict__ W, int64_t ldw,
    const float* __restrict__ bias, float scale, const float* __restrict__ scale_dev,
    float* __restrict__ Y, int64_t ldy, int M, int N, int K, ConvArgs cg, EpiArgs epi) {
    using E = typename C::E;
    using acc_t = typename E::acc_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // [buf][X stage | W stage]
    constexpr int BUF = C::BUF, STAGE_BYTES = C::STAGE_BYTES, KK = C::KK;
    constexpr int RPP = C::ROWS_PER_PIECE, CH = C::CHUNKS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_n = wave % C::WN, wave_m = wave / C::WN;
    const int lrow = lane & 31, lhalf = lane >> 5;

    // XCD-aware tile order (1-D grid).  Workgroup b is observed to run on XCD b % 8 (speed only, never relied on
    // for correctness): XCD x gets a CONTIGUOUS range of the tile order, so tiles that run at the same time on one
    // XCD (= one L2) are neighbours and share operand panels.
    //   tile grid divisible into 4 (m) x 8 (n) super-tiles: super-tile-major order (X and W panels both shared);
    //   otherwise: n-fastest order, so the column tiles of one row tile — which read the same X rows / pixels —
    //   sit on one XCD (conv with 3-6 column tiles: X is fetched into one L2 instead of 3-6).
    // The grid is padded to a multiple of 8 workgroups; the surplus ones exit.
    int tile_m, tile_n;
    {
        const int gx = (N + C::TN - 1) / C::TN, gy = (M + C::TM - 1) / C::TM;
        const int ntiles = gx * gy, per_xcd = (ntiles + 7) >> 3;
        const int b = blockIdx.x;
        if constexpr (C::CONV) {
            if (b >= 8 * per_xcd) {                    // appended by launch_cfg for halo output planes
                const int zs = epi.d2s_cout ? 2 : 1;
                zero_halo_border(Y, (int)(epi.mode == 2 ? ldy / 16 : ldy / 4), (int64_t)M / (cg.Ho * cg.Wo), zs * cg.Ho,
                                 zs * cg.Wo, epi.ohy, epi.ohx, b - 8 * per_xcd, (int)gridDim.x - 8 * per_xcd);
                return;
            }
        }
        const int o = (b & 7) * per_xcd + (b >> 3);
        if (o >= ntiles) return;                       // uniform for the workgroup, before any barrier
        if ((gx & 7) == 0 && (gy & 3) == 0) {
            const int st = o >> 5, in_st = o & 31, sgx = gx >> 3;
            tile_m = (st / sgx) * 4 + (in_st >> 3);
            tile_n = (st % sgx) * 8 + (in_st & 7);
        } else {
            tile_m = o / gx;
            tile_n = o - tile_m * gx;
        }
    }
    const int m0 = tile_m * C::TM, n0 = tile_n * C::TN;

    const int64_t ldx_b = ldx * 4, ldw_b = ldw * 4;  // row strides in bytes
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(W);
    if constexpr (!C::CONV && (std::is_same<E, ElemBf16>::value || std::is_same<E, ElemI8>::value)) {   // (int8: split-K, one tap)
        if (cg.z_nslice > 0) {
            // dispatch order: taps fastest, so the workgroups running at the same time walk the SAME K slice of X (and
            // shifted views of the same W rows): one HBM read serves all taps through the L2s / MALL
            const int z = blockIdx.y, slice = z / cg.H, tap = z - slice * cg.H;
            const int tr = tap / cg.z_kw, tc = tap - tr * cg.z_kw;
            Xb += (int64_t)slice * cg.z_kslice_bytes;
            Wb += (int64_t)slice * cg.z_kslice_bytes + (int64_t)tc * cg.z_w_copy_bytes + (int64_t)tr * cg.z_w_row_bytes;
            Y += ((int64_t)tap * cg.z_nslice + slice) * cg.z_y_stride;
        }
    }

    acc_t acc[C::TMW][C::TNW];
    constexpr bool SWAPT = elem_swapt<E>::value;
    if constexpr (SWAPT) {
        // start values: acc < T (T an integer)  <=>  acc - (T - 1/2) < 0.  Channels past N: never stored
#pragma unroll
        for (int b = 0; b < C::TNW; ++b) {
            const int nb = n0 + (wave_n * C::TNW + b) * 32 + 16 * lhalf;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float4 t = make_float4(0, 0, 0, 0);
                if (nb + r4 * 4 < N) t = *reinterpret_cast<const float4*>(epi.thr + nb + r4 * 4);      // N % 32 == 0 (host)
                // acc < T <=> acc < ceil(T) for the integer acc; a NaN threshold compares false for every acc: start far above zero
                auto start = [](float T) { return T == T ? 0.5f - ceilf(T) : 3.0e38f; };
                const float c[4] = {start(t.x), start(t.y), start(t.z), start(t.w)};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int a = 0; a < C::TMW; ++a) acc[a][b][r4 * 4 + e] = c[e];
            }
        }
    } else {
#pragma unroll
    for (int a = 0; a < C::TMW; ++a)
#pragma unroll
        for (int b = 0; b < C::TNW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;
    }

    unsigned long long dbg_ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ABL == 5: cycle stamps of one mid-loop stage
    unsigned long long dbg_wall[5] = {0, 0, 0, 0, 0}, dbg_end = 0, dbg_loop0 = 0;
    int dbg_simd = 0;
    dbg_ts[7] = __builtin_readcyclecounter();
    if constexpr (C::ABL >= 5) dbg_wall[0] = wall_clock64();
    const int nstages = C::ABL == 3 ? 0 : (E::kbytes(K) + STAGE_BYTES - 1) / STAGE_BYTES;

    auto read_frags = [&](const unsigned char* xs, const unsigned char* ws, int kk,
                          uint4 (&xf)[C::TMW], uint4 (&wf)[C::TNW]) {
        const int c = kk * 2 + lhalf;
#pragma unroll
        for (int a = 0; a < C::TMW; ++a) {
            const int row = (wave_m * C::TMW + a) * 32 + lrow;
            xf[a] = *reinterpret_cast<const uint4*>(xs + row * STAGE_BYTES + swz<STAGE_BYTES>(row, c) * 16);
        }
#pragma unroll
        for (int b = 0; b < C::TNW; ++b) {
            // (SWAPT: MFMA row i = channel 16 ((i / 4) % 2) + 4 (i / 8) + i % 4 of the block; the 16 lanes of every ds_read_b128
            // group still hit 16 rows distinct mod 16)
            const int row = (wave_n * C::TNW + b) * 32 + (SWAPT ? 16 * ((lrow >> 2) & 1) + 4 * (lrow >> 3) + (lrow & 3) : lrow);
            wf[b] = *reinterpret_cast<const uint4*>(ws + row * STAGE_BYTES + swz<STAGE_BYTES>(row, c) * 16);
        }
    };
    auto mfma_step = [&](uint4 (&xf)[C::TMW], uint4 (&wf)[C::TNW]) {
#pragma unroll
        for (int a = 0; a < C::TMW; ++a)
#pragma unroll
            for (int b = 0; b < C::TNW; ++b) acc[a][b] = E::mfma(xf[a], wf[b], acc[a][b]);
    };

    // per-tap scaling (E::TAPS): wave-uniform countdown of MFMA k-steps to the next tap boundary; at a boundary every accumulator
    // register is multiplied by rho_t before the first MFMA of tap t (the k-steps past the last tap — stage padding, zero
    // weights — never trigger it).  The table sits in LDS behind the stage buffers / the conv tap table: a global load here would
    // put an s_waitcnt vmcnt(0) — i.e. a drain of the three stages of LDS-DMA in flight — in front of every boundary; the next
    // factor is fetched (ds_read_b32) right after a boundary is processed, a whole tap ahead of its use.
    [[maybe_unused]] int tap_left = 0x3fffffff, tap_idx = 0;
    [[maybe_unused]] float tap_mul = 1.0f;
    [[maybe_unused]] const float* rho_lds = nullptr;
    if constexpr (E::TAPS) {
        float* r = reinterpret_cast<float*>(smem + C::NBUF * BUF + 64 + (C::VALID ? nstages * C::CHUNKS * 4 : 0));
        for (int e = tid; e <= epi.ntaps; e += C::NTHREADS) r[e] = epi.tap_rho[e];
        __syncthreads();
        rho_lds = r;
        if (epi.ntaps > 1) { tap_left = epi.tap_ksteps; tap_mul = r[1]; }
    }
    [[maybe_unused]] const float* ftab = nullptr;            // ROWS: [ntaps + 1][TM] row factors (entry 0: the epilogue's)
    if constexpr (elem_rows<E>::value) {
        float* f = const_cast<float*>(rho_lds) + ((epi.ntaps + 1 + 3) & ~3);
        const int T = epi.ntaps;
        if (tid < C::TM) {
            const int m = min(m0 + tid, M - 1);
            const int hw = cg.Ho * cg.Wo;
            const int n = m / hw, rem = m - n * hw;
            const int ho = rem / cg.Wo, wo = rem - ho * cg.Wo;
            const float* Ap = epi.row_A + (int64_t)n * epi.aH * epi.aW;
            const int h0 = ho * cg.sh - epi.aph, w0 = wo * cg.sw - epi.apw;
            auto a_of = [&](int t) -> float {
                const int i = t / cg.kw, j = t - i * cg.kw;
                const int hi = h0 + i * cg.dh, wi = w0 + j * cg.dw;
                return ((unsigned)hi < (unsigned)epi.aH && (unsigned)wi < (unsigned)epi.aW) ? Ap[hi * epi.aW + wi] : 0.0f;
            };
            // pass 1: the T scales of the row's window, independent loads (row t + 1 of the table holds a_t for now)
#pragma unroll 5
            for (int t = 0; t < T; ++t) f[(t + 1) * C::TM + tid] = a_of(t);
            // pass 2 (this thread's own column of the table): zeros inherit, ratios in place — row t is written after it was read
            float first = 1.0f;
            for (int t = 0; t < T; ++t) {
                const float v = f[(t + 1) * C::TM + tid];
                if (v != 0.0f) { first = v; break; }
            }
            float prev = first;
            for (int t = 0; t < T; ++t) {
                float v = f[(t + 1) * C::TM + tid];
                if (!(v != 0.0f)) v = prev;            // zero: any factor is right (D_t = 0); NaN compares unequal and is kept
                if (t > 0) f[t * C::TM + tid] = rho_lds[t] * (prev / v);
                prev = v;
            }
            f[tid] = prev;
        }
        __syncthreads();
        ftab = f;
    }
    // one MFMA k-step of the wave tile; at a tap boundary (E::TAPS) the accumulators are scaled first, in place.  (Interleaving
    // the packed multiplies of tile i + 1 with the MFMA of tile i — sched_group_barrier — was measured equal, AlexNet conv2 338 vs
    // 335 us, and makes the compiler alternate between two accumulator register sets: 253 VGPRs instead of 208.)
    // tap_boundary() = the bookkeeping in front of ONE k-step; k_step() = boundary + MFMAs.  The ping-pong loop calls the two
    // separately: the boundary in front of a stage's FIRST k-step is processed in the wave's LOAD segment (its accumulators are
    // idle there and the matrix pipe belongs to its SIMD partner), only a boundary between the two k-steps of a stage sits in the
    // compute segment.
    auto tap_boundary = [&]() {
        if constexpr (E::TAPS) {
            if (tap_left == 0) {
                if constexpr (elem_rows<E>::value) {
                    // one factor per accumulator row: register r of the lane is row (r & 3) + 8 (r >> 2) + 4 lhalf of the block
                    const float* ft = ftab + (tap_idx + 1) * C::TM + 4 * lhalf;
#pragma unroll
                    for (int a = 0; a < C::TMW; ++a) {
                        float4 f4[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            f4[j] = *reinterpret_cast<const float4*>(ft + (wave_m * C::TMW + a) * 32 + 8 * j);
#pragma unroll
                        for (int b = 0; b < C::TNW; ++b)
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                acc[a][b][4 * j + 0] *= f4[j].x;
                                acc[a][b][4 * j + 1] *= f4[j].y;
                                acc[a][b][4 * j + 2] *= f4[j].z;
                                acc[a][b][4 * j + 3] *= f4[j].w;
                            }
                    }
                } else {
#pragma unroll
                    for (int a = 0; a < C::TMW; ++a)
#pragma unroll
                        for (int b = 0; b < C::TNW; ++b) acc[a][b] *= tap_mul;
                }
                ++tap_idx;
                tap_left = tap_idx + 1 < epi.ntaps ? epi.tap_ksteps : 0x3fffffff;
                tap_mul = rho_lds[min(tap_idx + 1, epi.ntaps - 1)];
            }
            --tap_left;
        }
    };
    auto k_step = [&](uint4 (&xf)[C::TMW], uint4 (&wf)[C::TNW]) {
        tap_boundary();
        mfma_step(xf, wf);
    };

    if constexpr (C::PIPE >= 1) {
        // ---- pipelined main loops (asm-issued DMA) ----------------------------------------------
        // DMA pieces per wave.  When the W tile's pieces do not divide over the waves (ping-pong tiles 192 /
        // 64 wide) the piece index wraps: the surplus waves re-load a piece (same bytes to the same LDS
        // address), which keeps every wave's outstanding-DMA count — and so its vmcnt immediates — equal.
        constexpr int WPIECES = C::TN / RPP;
        constexpr int XP = C::TM / RPP / C::NWAVES, WP = (WPIECES + C::NWAVES - 1) / C::NWAVES;
        constexpr int NP = XP + WP;
        const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
        const int uwave = __builtin_amdgcn_readfirstlane(wave);
        const int p = lane % CH, rsub = lane / CH;   // lane -> (LDS chunk position, row within piece)
        // loop-invariant per-lane byte offsets of this wave's pieces (row clamped to the last valid
        // row: products of out-of-range rows are never stored).  Lane i lands on LDS position p of
        // its row, so it fetches the logical chunk swz(row, p) (the swizzle is an involution).
        unsigned voffx[XP], voffw[WP];
        // implicit conv: per piece the output pixel's top-left input coordinate and its pixel index;
        // the lane's logical chunk c within a stage is the same for all its pieces.
        int ch0[C::CONV ? XP : 1], cw0[C::CONV ? XP : 1];
        long long cpix[C::CONV ? XP : 1];
        const int lchunk = swz<STAGE_BYTES>(uwave * RPP + rsub, p);   // rows of a lane's pieces differ by multiples of 64
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int row = (j * C::NWAVES + uwave) * RPP + rsub;
            if constexpr (C::CONV) {
                const int m = min(m0 + row, M - 1);
                const int hw = cg.Ho * cg.Wo;
                int n, ho, wo;
                if (cg.sh_w >= 0) {                     // wave-uniform: power-of-two maps
                    n = m >> cg.sh_hw;
                    const int rem = m & (hw - 1);
                    ho = rem >> cg.sh_w;
                    wo = rem & (cg.Wo - 1);
                } else {
                    n = m / hw;
                    const int rem = m - n * hw;
                    ho = rem / cg.Wo;
                    wo = rem - ho * cg.Wo;
                }
                ch0[j] = ho * cg.sh - cg.ph;
                cw0[j] = wo * cg.sw - cg.pw;
                // byte address of chunk 0 of the window's top-left pixel (may lie before the plane: only
                // dereferenced for in-range taps)
                cpix[j] = (long long)(uintptr_t)Xb + ((((long long)n * cg.H + ch0[j]) * cg.W + cw0[j]) * cg.cpp) * 16;
                // VALID: ph = pw = 0, the plane is < 4 GiB (host check): byte offset of the window's first chunk
                voffx[j] = C::VALID ? (unsigned)((((unsigned)n * cg.H + ch0[j]) * cg.W + cw0[j]) * cg.cpp) * 16u : 0u;
            } else {
                voffx[j] = (unsigned)(min(m0 + row, M - 1) * ldx_b) + (unsigned)(swz<STAGE_BYTES>(row, p) * 16);
            }
        }
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int row = ((j * C::NWAVES + uwave) % WPIECES) * RPP + rsub;
            voffw[j] = (unsigned)(min(n0 + row, N - 1) * ldw_b) + (unsigned)(swz<STAGE_BYTES>(row, p) * 16);
        }
        // implicit conv: tap coordinates of this lane's chunk in stage s (shared by all its X pieces): the tap's
        // row / column displacement and its byte offset from the window's top-left chunk.
        // Beside the partner's MFMA stream a VALU instruction of the loading wave costs ~12 cycles whatever its
        // type (tools/pp_stamps_conv.py; an add/select formulation that advances the state without multiplies
        // measured SLOWER than this 14-instruction from-scratch form), so the count is what matters:
        //   general mode: evaluated per stage (two magic divisions);
        //   VALID mode:   the byte offset of every (stage, chunk) is tabulated ONCE in LDS behind the stage
        //                 buffers (tap_table, filled in the prologue) and a stage costs one ds_read_b32.
        int tap_di = 0, tap_dj = 0, tap_boff = 0;
        bool tap_ok = false;
        const long long zero_addr = (long long)(uintptr_t)zero16_storage;
        const int last_boff = (((cg.kh - 1) * cg.dh * cg.W + (cg.kw - 1) * cg.dw) * cg.cpp + cg.cpp - 1) * 16;
        auto tap_eval = [&](unsigned q, int& di, int& dj, int& boff, bool& ok) {
            const unsigned tap = cg.cpp == 1 ? q : __umulhi(q, cg.magic_cpp);
            const int tap_sub = (int)(q - tap * cg.cpp);
            const unsigned ti = cg.kw == 1 ? tap : __umulhi(tap, cg.magic_kw);
            const int tj = (int)(tap - ti * cg.kw);
            di = (int)ti * cg.dh;
            dj = tj * cg.dw;
            ok = (int)ti < cg.kh;             // false for chunks past the last tap (K tail of the last stage)
            boff = ((di * cg.W + dj) * cg.cpp + tap_sub) * 16;   // < 2^31: host checks H*W*cpp*16
        };
        int* tap_table = reinterpret_cast<int*>(smem + C::NBUF * BUF + 64);   // VALID: [nstages][CH] byte offsets
        if constexpr (C::VALID) {
            for (int e = tid; e < nstages * CH; e += C::NTHREADS) {
                int di, dj, boff;
                bool ok;
                tap_eval((unsigned)e, di, dj, boff, ok);
                // no zero page in this mode: chunks past the last tap (their weights are zero) re-read the
                // window's last in-range chunk, which is finite data
                tap_table[e] = ok ? boff : last_boff;
            }
            __syncthreads();
        }
        auto conv_stage = [&](int s) {
            if constexpr (C::VALID) tap_boff = tap_table[s * CH + lchunk];
            else tap_eval((unsigned)(s * CH + lchunk), tap_di, tap_dj, tap_boff, tap_ok);
        };
        auto issue_piece = [&](int j, int s, int buf, auto lean_tag) {  // j is a compile-time constant after unrolling
            constexpr bool lean = decltype(lean_tag)::value;   // caller brackets the run with m0_save / m0_restore
            const unsigned ldsbuf = __builtin_amdgcn_readfirstlane(lds0 + buf * BUF);
            if (j < XP) {
                const unsigned poff = __builtin_amdgcn_readfirstlane(((j * C::NWAVES + uwave) * RPP) * STAGE_BYTES);
                if constexpr (C::VALID) {
                    const unsigned voff = voffx[j < XP ? j : 0] + (unsigned)tap_boff;     // one VALU add per piece
                    if constexpr (lean) glds16_lean(Xb, voff, ldsbuf, poff);
                    else glds16_asm(Xb, voff, ldsbuf + poff);
                } else if constexpr (C::CONV) {
                    const int jj = j < XP ? j : 0;
                    const unsigned hi = (unsigned)(ch0[jj] + tap_di), wi = (unsigned)(cw0[jj] + tap_dj);
                    const bool ok = tap_ok & (hi < (unsigned)cg.H) & (wi < (unsigned)cg.W);   // unsigned: < 0 wraps high
                    const long long a = cpix[jj] + tap_boff;
                    const long long src = ok ? a : zero_addr;          // two v_cndmask, no branch
                    if constexpr (lean) glds16_lean64(reinterpret_cast<const unsigned char*>(src), ldsbuf, poff);
                    else glds16_asm64(reinterpret_cast<const unsigned char*>(src), ldsbuf + poff);
                } else {
                    if constexpr (lean) glds16_lean(Xb + (int64_t)s * STAGE_BYTES, voffx[j < XP ? j : 0], ldsbuf, poff);
                    else glds16_asm(Xb + (int64_t)s * STAGE_BYTES, voffx[j < XP ? j : 0], ldsbuf + poff);
                }
            } else {
                const int jw = j - XP;
                const unsigned poff = __builtin_amdgcn_readfirstlane(
                    C::X_STAGE + (((jw * C::NWAVES + uwave) % WPIECES) * RPP) * STAGE_BYTES);
                if constexpr (lean) glds16_lean(Wb + (int64_t)s * STAGE_BYTES, voffw[jw >= 0 && jw < WP ? jw : 0], ldsbuf, poff);
                else glds16_asm(Wb + (int64_t)s * STAGE_BYTES, voffw[jw >= 0 && jw < WP ? jw : 0], ldsbuf + poff);
            }
        };
        if constexpr (C::PIPE == 2) {
            // ---- ping-pong: the two waves of a SIMD alternate roles every 64-byte stage ----------------
            // An in-order wave cannot issue MFMAs while it is issuing LDS-DMA pieces / fragment reads, and
            // two waves running the SAME interleaved stream stall at the same places (measured: stage time
            // = MFMA time + load time, tools/ubench/mfma_power.hip + the ABL variants).  Here a wave's
            // stage is a LOAD segment (all 12 fragment reads of the stage into registers + its 4 DMA pieces
            // of stage s+3) followed by a COMPUTE segment (16 register-only MFMAs), one s_barrier after
            // each, and waves 4-7 (the SIMD partners of waves 0-3) run one segment ahead: in every slot one
            // wave per SIMD owns the matrix pipe while its partner owns the LDS / DMA issue.
            //   slot 2s: B = waves 4-7 load stage s   | A = waves 0-3 compute stage s-1
            //   slot 2s+1: B compute stage s          | A load stage s
            // Ring of 4 stage buffers; stage s+3 is written into the buffer stage s-1 was read from (reads
            // finished, lgkmcnt(0), before the barrier that ends slot 2s-1).  A wave has three stages of
            // pieces in flight; at the end of load(s) it waits for its pieces of stage s+1 (vmcnt(2*NP)),
            // and the barrier publishes them before the first reader (B, slot 2s+2).
            constexpr int AHEAD = 3;
            // Role = rank of the wave among the workgroup's waves on ITS SIMD (read from HW_ID), so the two
            // co-resident waves of a SIMD always get opposite roles whatever the dispatcher's placement
            // (correctness does not depend on it: any split with equal barrier counts is valid).
            int grp;
            {
                volatile int* simd_of = reinterpret_cast<volatile int*>(smem + C::NBUF * BUF);
                const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);  // HW_ID.SIMD_ID
                if (lane == 0) simd_of[uwave] = simd;
                __syncthreads();
                int rank = 0;
                for (int w2 = 0; w2 < C::NWAVES; ++w2) rank += (w2 < uwave && simd_of[w2] == simd) ? 1 : 0;
                grp = __builtin_amdgcn_readfirstlane(rank & 1);
                if constexpr (C::ABL >= 5) dbg_simd = simd;
            }
#pragma unroll
            for (int s = 0; s < AHEAD; ++s)
                if (s < nstages) {
                    if constexpr (C::CONV) conv_stage(s);
#pragma unroll
                    for (int j = 0; j < NP; ++j) issue_piece(j, s, s, std::false_type{});
                }
            int tap_next = 0;
            if constexpr (C::VALID) tap_next = tap_table[min(AHEAD, max(nstages - 1, 0)) * CH + lchunk];
            if (nstages >= AHEAD) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if constexpr (C::ABL >= 5) { dbg_wall[1] = wall_clock64(); dbg_loop0 = __builtin_readcyclecounter(); }
            if (grp == 0 && nstages > 0) __syncthreads();   // group A trails by one slot

            auto pp_stage = [&](int s, auto issue_tag, auto last_tag) {
                constexpr bool issue = decltype(issue_tag)::value, last = decltype(last_tag)::value;
                const unsigned char* xs = smem + (s & 3) * BUF;
                const unsigned char* ws = xs + C::X_STAGE;
                uint4 xf0[C::TMW], wf0[C::TNW], xf1[C::TMW], wf1[C::TNW];
                if constexpr (C::ABL == 5 && issue) dbg_ts[0] = stamp_now();
                read_frags(xs, ws, 0, xf0, wf0);
                read_frags(xs, ws, 1, xf1, wf1);
                if constexpr (C::ABL == 5 && issue) dbg_ts[1] = stamp_now();
                if constexpr (issue) {
                    // VALID conv: the stage's tap offset was read from the LDS table at the END of the previous load
                    // segment (tap_next).  Read here, the pieces would wait for it behind the twelve fragment reads
                    // just issued (LDS returns in order): the whole fragment latency in front of the DMA issue.
                    if constexpr (C::VALID) tap_boff = tap_next;
                    else if constexpr (C::CONV) conv_stage(s + AHEAD);
                    const unsigned m0_keep = m0_save();
#pragma unroll
                    for (int j = 0; j < NP; ++j) issue_piece(j, s + AHEAD, (s + AHEAD) & 3, std::true_type{});
                    m0_restore(m0_keep);
                    if constexpr (C::VALID) tap_next = tap_table[min(s + 1 + AHEAD, nstages - 1) * CH + lchunk];
                    if constexpr (C::ABL == 5 && issue) dbg_ts[2] = stamp_now();
                    tap_boundary();
                    asm volatile("s_waitcnt vmcnt(%1) lgkmcnt(0)" : "+v"(tap_next) : "n"(2 * NP) : "memory");
                } else {
                    tap_boundary();
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                }
                if constexpr (C::ABL == 5 && issue) dbg_ts[3] = stamp_now();
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (C::ABL == 5 && issue) dbg_ts[4] = stamp_now();
                mfma_step(xf0, wf0);              // its boundary: processed in the load segment above
                k_step(xf1, wf1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (C::ABL == 5 && issue) dbg_ts[5] = stamp_now();
                if (!(last && grp == 0)) __syncthreads();   // A's last compute has no partner segment
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (C::ABL == 5 && issue) dbg_ts[6] = stamp_now();
            };
            int s = 0;
            for (; s + AHEAD < nstages; ++s) pp_stage(s, std::true_type{}, std::false_type{});
            for (; s + 1 < nstages; ++s) pp_stage(s, std::false_type{}, std::false_type{});
            if (nstages > 0) pp_stage(nstages - 1, std::false_type{}, std::true_type{});
            if constexpr (C::ABL >= 5) {   // profiling only: the (0,0) tile's waves overwrite Y row 0.. with their stamps
                dbg_end = __builtin_readcyclecounter();
                dbg_wall[2] = wall_clock64();
            }
        } else {
            // ring of NB stage buffers, AH = NB - 1 stages of DMA in flight (NB = 2: the double-buffered loop)
            constexpr int NB = C::NBUF, AH = NB - 1;
    #pragma unroll
            for (int s = 0; s < AH; ++s)
                if (s < nstages) {
                    if constexpr (C::CONV) conv_stage(s);
    #pragma unroll
                    for (int j = 0; j < NP; ++j) issue_piece(j, s, s, std::false_type{});
                }
            if (nstages >= AH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AH - 1) * NP) : "memory");   // stage 0 has landed
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
    
            auto stage_body = [&](int s, int buf, auto more_tag) {      // buf = s % NB; stage s + AH goes into the buffer stage s - 1 left
                constexpr bool more = decltype(more_tag)::value;
                const int nbuf = buf == 0 ? NB - 1 : buf - 1;
                const unsigned char* xs = smem + buf * BUF;
                const unsigned char* ws = xs + C::X_STAGE;
                uint4 xfA[C::TMW], wfA[C::TNW], xfB[C::TMW], wfB[C::TNW];
                if constexpr (C::ABL == 4) {
    #pragma unroll
                    for (int a = 0; a < C::TMW; ++a) xfA[a] = xfB[a] = make_uint4(0x22222222u, 0x2a2a2a2au, lane, s);
    #pragma unroll
                    for (int b = 0; b < C::TNW; ++b) wfA[b] = wfB[b] = make_uint4(0xa2a2a2a2u, 0x2a2a2a2au, lane, s);
                }
                if constexpr (C::ABL != 4) read_frags(xs, ws, 0, xfA, wfA);
                if constexpr (C::CONV && more) conv_stage(s + AH);
    #pragma unroll
                for (int kk = 0; kk < KK; kk += 2) {
                    // DMA pieces are spread over the k-steps; fragments of step kk+1 are requested before
                    // the MFMAs of step kk so LDS latency hides under the matrix pipe.
                    if constexpr (more && C::ABL != 2) {
    #pragma unroll
                        for (int j = kk * NP / KK; j < (kk + 1) * NP / KK; ++j) issue_piece(j, s + AH, nbuf, std::false_type{});
                    }
                    if constexpr (C::ABL != 4) read_frags(xs, ws, kk + 1, xfB, wfB);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (C::ABL != 1) k_step(xfA, wfA);
                    else asm volatile("" ::"v"(xfA[0].x), "v"(wfA[0].x), "v"(xfA[C::TMW - 1].w), "v"(wfA[C::TNW - 1].w));
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (more && C::ABL != 2) {
    #pragma unroll
                        for (int j = (kk + 1) * NP / KK; j < (kk + 2) * NP / KK; ++j) issue_piece(j, s + AH, nbuf, std::false_type{});
                    }
                    if constexpr (C::ABL != 4) {
                        if (kk + 2 < KK) read_frags(xs, ws, kk + 2, xfA, wfA);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (C::ABL != 1) k_step(xfB, wfB);
                    else asm volatile("" ::"v"(xfB[0].x), "v"(wfB[0].x), "v"(xfB[C::TMW - 1].w), "v"(wfB[C::TNW - 1].w));
                    __builtin_amdgcn_sched_barrier(0);
                }
                // this wave's pieces of stage s + 1 have landed (the younger stages stay in flight)
                if constexpr (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((AH - 1) * NP) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                                   // ... and everyone's are visible
            };
            int s = 0, buf = 0;
            for (; s + AH < nstages; ++s, buf = buf + 1 == NB ? 0 : buf + 1) stage_body(s, buf, std::true_type{});
            for (; s < nstages; ++s, buf = buf + 1 == NB ? 0 : buf + 1) stage_body(s, buf, std::false_type{});
        }
    } else {
        // ---- generic main loop (builtin DMA, 64-bit addresses, any row stride % 4 words) ------------
        const unsigned char* zero16 = zero16_storage;
        auto dma_operand = [&](const unsigned char* G, int64_t ld_b, int row0_global, int nrows_valid,
                               unsigned char* ls, int tile_rows, int s) {
            const int p = lane % CH, rsub = lane / CH;
            const int uwave = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
            for (int q = 0; q < (C::TM > C::TN ? C::TM : C::TN) / RPP / C::NWAVES; ++q) {
                const int g = q * C::NWAVES + uwave;
                if (g >= tile_rows / RPP) break;
                const int r0 = g * RPP, row = r0 + rsub;
                const int c = swz<STAGE_BYTES>(row, p);
                const int64_t kb = (int64_t)s * STAGE_BYTES + c * 16;
                const int grow = min(row0_global + row, nrows_valid - 1);
                const unsigned char* src = (kb < ld_b) ? G + (int64_t)grow * ld_b + kb : zero16;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(ls + r0 * STAGE_BYTES),
                                                 16, 0, 0);
            }
        };
        if (nstages > 0) {
            dma_operand(Xb, ldx_b, m0, M, smem, C::TM, 0);
            dma_operand(Wb, ldw_b, n0, N, smem + C::X_STAGE, C::TN, 0);
        }
        __syncthreads();  // drains the DMA (vmcnt(0)): buffer 0 is ready
        for (int s = 0; s < nstages; ++s) {
            const int buf = s & 1;
            const unsigned char* xs = smem + buf * BUF;
            const unsigned char* ws = xs + C::X_STAGE;
            // (1) every fragment of this stage into registers, (2) DMA of the next stage, (3) the
            // register-only MFMAs while it is in flight (hipcc drains vmcnt(0) in front of any ds_read
            // that follows a builtin LDS-DMA, so no LDS read may sit between (2) and the barrier).
            uint4 xf[KK][C::TMW], wf[KK][C::TNW];
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) read_frags(xs, ws, kk, xf[kk], wf[kk]);
            if (s + 1 < nstages) {
                dma_operand(Xb, ldx_b, m0, M, smem + (buf ^ 1) * BUF, C::TM, s + 1);
                dma_operand(Wb, ldw_b, n0, N, smem + (buf ^ 1) * BUF + C::X_STAGE, C::TN, s + 1);
            }
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) mfma_step(xf[kk], wf[kk]);
            __builtin_amdgcn_sched_barrier(0);  // keep the MFMAs above the barrier's vmcnt(0)
            __syncthreads();
        }
    }

    if (scale_dev) scale *= *scale_dev;   // device-resident factor (e.g. DoReFa's E = mean|W|): no host sync
    if constexpr (E::TAPS) scale *= epi.tap_rho[epi.ntaps];   // alpha of the last tap closes the Horner form
    if constexpr (elem_rows<E>::value) {                      // ... and the last tap's per-pixel scale, per accumulator row
        const float* ft = ftab + 4 * lhalf;
#pragma unroll
        for (int a = 0; a < C::TMW; ++a) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 f4 = *reinterpret_cast<const float4*>(ft + (wave_m * C::TMW + a) * 32 + 8 * j);
#pragma unroll
                for (int b = 0; b < C::TNW; ++b) {
                    acc[a][b][4 * j + 0] *= f4.x;
                    acc[a][b][4 * j + 1] *= f4.y;
                    acc[a][b][4 * j + 2] *= f4.z;
                    acc[a][b][4 * j + 3] *= f4.w;
                }
            }
        }
        __syncthreads();      // every wave has read its factors: the stage buffers' neighbourhood may become transpose patches
    }
    // ---- epilogue: D[row = m][col = n]; lane owns column n, rows m = mb + (r&3) + 8*(r>>2) + 4*lhalf ---
    // The store tail is store-ISSUE bound (1024 dword wave-stores per CU: 13 us of a 41 us kernel at
    // 4096^3, tools/pp_stamps.py), so each 32x32 tile is transposed through a wave-private 4 KiB LDS
    // patch (the stage buffers are dead: every fragment read was waited for before the last barrier and
    // no DMA is in flight) and leaves as 4 dwordx4 wave-stores of 8 full 128-byte lines each: 4x fewer
    // store instructions.  LDS ops of one wave execute in issue order, so the patch needs no barrier.
    const bool wide = ((ldy & 3) == 0) && ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0);
    if constexpr (SWAPT) {
        // ---- threshold epilogue, weights-as-rows form: lane (l % 32, l / 32) holds channels 16 (l / 32) .. + 15 of position l % 32 of
        // every 32 x 32 block; the bit of channel r is the sign bit of register r (flipped where alpha < 0).  Host: epi.alpha and
        // epi.thr set, N % 32 == 0, no depth-to-space; epi.mode == 3: the next conv's nibble plane, else the bit plane.
        uint32_t* B = reinterpret_cast<uint32_t*>(Y);
        // alpha < 0 per channel of block b, as the bit-plane word (uniform) and as this lane's two nibble words | the magnitude bits
        uint32_t negw[C::TNW], nn[C::TNW][2];
#pragma unroll
        for (int b = 0; b < C::TNW; ++b) {
            const int n = n0 + (wave_n * C::TNW + b) * 32 + lrow;
            negw[b] = (uint32_t)__ballot(n < N && epi.alpha[n] < 0.0f);
            const uint32_t my = (negw[b] >> (16 * lhalf)) & 0xFFFFu;
            nn[b][0] = (spread8(my) << 3) | 0x22222222u;
            nn[b][1] = (spread8(my >> 8) << 3) | 0x22222222u;
        }
#pragma unroll
        for (int a = 0; a < C::TMW; ++a) {
            const int m = m0 + (wave_m * C::TMW + a) * 32 + lrow;
            if (epi.mode == 3) {
                int orow = m;
                if (epi.ohy | epi.ohx) {
                    const unsigned um = (unsigned)m;
                    unsigned img, ho, wo;
                    if (cg.sh_w >= 0) {
                        img = um >> cg.sh_hw;
                        const unsigned rem = um & (unsigned)(cg.Ho * cg.Wo - 1);
                        ho = rem >> cg.sh_w;
                        wo = rem & (unsigned)(cg.Wo - 1);
                    } else {
                        img = epi.magic_hw ? (unsigned)__umul64hi((unsigned long long)um, epi.magic_hw) : um;
                        const unsigned rem = um - img * (unsigned)(cg.Ho * cg.Wo);
                        ho = epi.magic_w ? (unsigned)__umul64hi((unsigned long long)rem, epi.magic_w) : rem;
                        wo = rem - ho * (unsigned)cg.Wo;
                    }
                    orow = (int)((img * ((unsigned)cg.Ho + 2u * (unsigned)epi.ohy) + ho + (unsigned)epi.ohy) *
                                     ((unsigned)cg.Wo + 2u * (unsigned)epi.ohx) + wo + (unsigned)epi.ohx);
                }
                uint32_t* dst = B + (int64_t)orow * ldy + lhalf * 2;
#pragma unroll
                for (int b = 0; b < C::TNW; ++b) {
                    const int nb = n0 + (wave_n * C::TNW + b) * 32;
                    uint32_t lo = 0, hi = 0;
#pragma unroll
                    for (int r = 7; r >= 0; --r) lo = __builtin_amdgcn_alignbit(lo, __float_as_uint(acc[a][b][r]), 28);
#pragma unroll
                    for (int r = 15; r >= 8; --r) hi = __builtin_amdgcn_alignbit(hi, __float_as_uint(acc[a][b][r]), 28);
                    if (m < M && nb < N)
                        *reinterpret_cast<uint2*>(dst + (nb >> 3)) = make_uint2((lo & 0x88888888u) ^ nn[b][0], (hi & 0x88888888u) ^ nn[b][1]);
                }
            } else {
#pragma unroll
                for (int b = 0; b < C::TNW; ++b) {
                    const int nb = n0 + (wave_n * C::TNW + b) * 32;
                    uint32_t w16 = 0;
#pragma unroll
                    for (int r = 15; r >= 0; --r) w16 = __builtin_amdgcn_alignbit(w16, __float_as_uint(acc[a][b][r]), 31);
                    // both halves' 16 bits into every lane: v_permlane32_swap leaves the low half's value in x, the high half's in y
                    const auto sw = __builtin_amdgcn_permlane32_swap(w16, w16, false, false);
                    const uint32_t word = ((sw[1] << 16) | sw[0]) ^ negw[b];
                    const int wcol = nb >> 5;
                    if (lane < 32 && m < M && nb < N && wcol < ldy) B[(int64_t)m * ldy + wcol] = word;
                    // the row's pad words past the last column tile (ldy rounds ceil(N/32) up to 4) are zeroed here
                    if (b == C::TNW - 1 && wave_n == C::WN - 1 && n0 + C::TN >= N && lane < 32 && m < M)
                        for (int wc = (N + 31) >> 5; wc < ldy; ++wc) B[(int64_t)m * ldy + wc] = 0u;
                }
            }
        }
        return;
    }
    if constexpr (DEVBN) if (epi.mode == 4) {
        // fp32 output through eval-mode BatchNorm in the DEVICE's arithmetic (the conv -> BatchNorm shortcut branch of a DoReFa ResNet
        // block): y = fma(fl(fl(v - mean) * rs), weight, bias) on the value v the plain epilogue would have stored — the expression of
        // bn_eval_device_kernel, applied before the store instead of in a second pass over the fp32 tensor.  Host: wide stores only.
        float* T = reinterpret_cast<float*>(smem) + wave * 1024;
#pragma unroll
        for (int b = 0; b < C::TNW; ++b) {
            const int nb = n0 + (wave_n * C::TNW + b) * 32;
            const float bv = (bias && nb + lrow < N) ? bias[nb + lrow] : 0.0f;
            const int n = nb + (lane & 7) * 4;
            float4 mean = make_float4(0, 0, 0, 0), rs = mean, bw = mean, bb = mean;
            if (n < N) {                      // N % 4 == 0: the lane's four channels exist together
                mean = *reinterpret_cast<const float4*>(epi.bn_stats + n);
                rs = *reinterpret_cast<const float4*>(epi.bn_stats + N + n);
                bw = *reinterpret_cast<const float4*>(epi.alpha + n);
                bb = *reinterpret_cast<const float4*>(epi.beta + n);
            }
#pragma unroll
            for (int a = 0; a < C::TMW; ++a) {
                const int mb = m0 + (wave_m * C::TMW + a) * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    T[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * 32 + lrow] = E::out(acc[a][b][r], scale, bv);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 8 + (lane >> 3);
                    const float4 v = *reinterpret_cast<const float4*>(T + row * 32 + (lane & 7) * 4);
                    const int m = mb + row;
                    if (m < M && n < N) {
                        float4 o;
                        o.x = __builtin_fmaf((v.x - mean.x) * rs.x, bw.x, bb.x);
                        o.y = __builtin_fmaf((v.y - mean.y) * rs.y, bw.y, bb.y);
                        o.z = __builtin_fmaf((v.z - mean.z) * rs.z, bw.z, bb.z);
                        o.w = __builtin_fmaf((v.w - mean.w) * rs.w, bw.w, bb.w);
                        *reinterpret_cast<float4*>(Y + (int64_t)m * ldy + n) = o;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        return;
    }
    if constexpr (E::CODE_EPI && C::CONV) if (epi.mode == 2) {
        // int8 codes: the 32x32 tile is transposed through the wave-private LDS patch as in the fp32 store below,
        // so a lane holds 4 consecutive channels of one output row: per-channel affine, residual, ReLU, rint ->
        // one dword of codes per lane (4x less store traffic than fp32, and no fp32 activation in HBM at all).
        int8_t* Q = reinterpret_cast<int8_t*>(Y);
        float* T = reinterpret_cast<float*>(smem) + wave * 1024;
        const bool rwide = epi.res_f32 && ((epi.ldr & 3) == 0) && ((reinterpret_cast<uintptr_t>(epi.res_f32) & 15) == 0);
        int bad = 0;
        // plane rows of this lane's 4 x TMW output pixels (identity without halos)
        const bool halo = (epi.ohy | epi.ohx | epi.rhy | epi.rhx) != 0;
        int orow[C::TMW][4], rrow[C::TMW][4];
#pragma unroll
        for (int a = 0; a < C::TMW; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = m0 + (wave_m * C::TMW + a) * 32 + i * 8 + (lane >> 3);
                orow[a][i] = rrow[a][i] = m;
                if (halo && m < M) {
                    const unsigned um = (unsigned)m;
                    unsigned img, ho, wo;
                    if (cg.sh_w >= 0) {                 // wave-uniform: power-of-two maps
                        img = um >> cg.sh_hw;
                        const unsigned rem = um & (unsigned)(cg.Ho * cg.Wo - 1);
                        ho = rem >> cg.sh_w;
                        wo = rem & (unsigned)(cg.Wo - 1);
                    } else {
                        img = epi.magic_hw ? (unsigned)__umul64hi((unsigned long long)um, epi.magic_hw) : um;
                        const unsigned rem = um - img * (unsigned)(cg.Ho * cg.Wo);
                        ho = epi.magic_w ? (unsigned)__umul64hi((unsigned long long)rem, epi.magic_w) : rem;
                        wo = rem - ho * (unsigned)cg.Wo;
                    }
                    orow[a][i] = (int)((img * (unsigned)(cg.Ho + 2 * epi.ohy) + ho + epi.ohy) * (unsigned)(cg.Wo + 2 * epi.ohx) + wo + epi.ohx);
                    rrow[a][i] = (int)((img * (unsigned)(cg.Ho + 2 * epi.rhy) + ho + epi.rhy) * (unsigned)(cg.Wo + 2 * epi.rhx) + wo + epi.rhx);
                }
            }
        if constexpr (DEVBN) {
            // Straight-line form of the common case — whole column tiles (every channel of the lane's dword exists), no conv bias,
            // no fp32 residual, ReLU (if any) behind the BatchNorm: the residual kind and the ReLU are template flags of the body,
            // so the per-element chain is 10 (13 with a code residual) VALU instructions with no per-element scalar branch.  The
            // general form below spends ~2x that on its wave-uniform-but-unknown-at-compile-time conditions, and these short
            // small-N launches are VALU-issue bound in their epilogue (profiles/r5_c4_pmc.md).  Same roundings, same codes.
            // (whole ROW tiles too: a bounds branch per row group puts every group in its own basic block, and the waitcnt pass then
            //  drains the previous group's code store before each group's first use of a residual word — profiles/r5_c4_direct_conv.md)
            const bool fast = !bias && (!epi.res_f32 || (rwide && !epi.res_codes)) && epi.relu != 2 && (N & 3) == 0 && n0 + C::TN <= N && m0 + C::TM <= M;
            if (fast) {
                int badf = 0;
                const float levels = epi.levels, rscale = epi.rscale;
                auto body = [&](auto rc_tag, auto relu_tag, auto rf_tag) {
                    constexpr bool RC = decltype(rc_tag)::value, RELU = decltype(relu_tag)::value, RF = decltype(rf_tag)::value;
#pragma unroll
                    for (int b = 0; b < C::TNW; ++b) {
                        const int nb = n0 + (wave_n * C::TNW + b) * 32;
                        const int n = nb + (lane & 7) * 4;
                        float al[4], be[4], mean[4], rs[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            al[e] = epi.alpha[n + e];
                            be[e] = epi.beta[n + e];
                            mean[e] = epi.bn_stats[n + e];
                            rs[e] = epi.bn_stats[N + n + e];
                        }
                        // the residual words of this column block, all requested up front
                        uint32_t rwd[C::TMW][4];
#pragma unroll
                        for (int a = 0; a < C::TMW; ++a)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                rwd[a][i] = 0;
                                if constexpr (RC) rwd[a][i] = *reinterpret_cast<const uint32_t*>(epi.res_codes + (int64_t)rrow[a][i] * epi.ldrc + n);
                            }
#pragma unroll
                        for (int a = 0; a < C::TMW; ++a) {
                            const int mb = m0 + (wave_m * C::TMW + a) * 32;
                            // fp32 residual (the conv -> BatchNorm shortcut branch, already normalised): this row group's four float4,
                            // requested before the transpose below
                            float4 rf4[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                rf4[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                                if constexpr (RF)
                                    rf4[i] = *reinterpret_cast<const float4*>(epi.res_f32 + (int64_t)(mb + i * 8 + (lane >> 3)) * epi.ldr + n);
                            }
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                T[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * 32 + lrow] = (float)acc[a][b][r] * scale;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int row = i * 8 + (lane >> 3);
                                const float4 v4 = *reinterpret_cast<const float4*>(T + row * 32 + (lane & 7) * 4);
                                {
                                    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
                                    const float u[4] = {rf4[i].x, rf4[i].y, rf4[i].z, rf4[i].w};
                                    const uint32_t rword = rwd[a][i];
                                    uint32_t word = 0;
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        float t = __builtin_fmaf((v[e] - mean[e]) * rs[e], al[e], be[e]);
                                        if constexpr (RF) t = t + u[e];
                                        if constexpr (RC) t = t + rscale * (float)(int8_t)(rword >> (8 * e));
                                        if constexpr (RELU) t = t < 0.0f ? 0.0f : t;
                                        const float qf = rintf(levels * t);
                                        const bool ok = __builtin_fabsf(qf) <= 127.0f;       // NaN -> false
                                        const int q = ok ? (int)qf : 0;
                                        badf |= ok ? 0 : 1;
                                        word |= (uint32_t)(uint8_t)(int8_t)q << (8 * e);
                                    }
                                    *reinterpret_cast<uint32_t*>(Q + (int64_t)orow[a][i] * ldy + n) = word;
                                }
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            if (!halo && b == C::TNW - 1 && wave_n == C::WN - 1 && n0 + C::TN >= N && lane < 32 && mb + lane < M)
                                for (int c = n0 + C::TN; c < ldy; c += 4)
                                    *reinterpret_cast<uint32_t*>(Q + (int64_t)(mb + lane) * ldy + c) = 0u;
                        }
                    }
                };
                if (epi.res_codes) {
                    if (epi.relu == 1) body(std::true_type{}, std::true_type{}, std::false_type{});
                    else body(std::true_type{}, std::false_type{}, std::false_type{});
                } else if (epi.res_f32) {
                    if (epi.relu == 1) body(std::false_type{}, std::true_type{}, std::true_type{});
                    else body(std::false_type{}, std::false_type{}, std::true_type{});
                } else {
                    if (epi.relu == 1) body(std::false_type{}, std::true_type{}, std::false_type{});
                    else body(std::false_type{}, std::false_type{}, std::false_type{});
                }
                if (__any(badf) && lane == 0) atomicOr(epi.overflow, 1);
                return;
            }
        }
#pragma unroll
        for (int b = 0; b < C::TNW; ++b) {
            const int nb = n0 + (wave_n * C::TNW + b) * 32;
            const float bv = (bias && nb + lrow < N) ? bias[nb + lrow] : 0.0f;
            const int n = nb + (lane & 7) * 4;
            // per-channel epilogue constants of the lane's 4 channels, hoisted out of the row loops: folded form (alpha, beta,
            // residual alpha / beta), device form (weight, bias, mean, rs) — 16 registers either way
            float al[4], be[4], ral[4], rbe[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = n + e < N;
                al[e] = in ? epi.alpha[n + e] : 0.0f;
                be[e] = in ? epi.beta[n + e] : 0.0f;
                if constexpr (DEVBN) {      // ral = mean, rbe = rs
                    ral[e] = in ? epi.bn_stats[n + e] : 0.0f;
                    rbe[e] = in ? epi.bn_stats[N + n + e] : 1.0f;
                } else {
                    ral[e] = (in && epi.ralpha) ? epi.ralpha[n + e] : 1.0f;
                    rbe[e] = (in && epi.ralpha) ? epi.rbeta[n + e] : 0.0f;
                }
            }
#pragma unroll
            for (int a = 0; a < C::TMW; ++a) {
                const int mb = m0 + (wave_m * C::TMW + a) * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    T[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * 32 + lrow] = E::out(acc[a][b][r], scale, bv);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 8 + (lane >> 3);
                    const float4 v4 = *reinterpret_cast<const float4*>(T + row * 32 + (lane & 7) * 4);
                    const int m = mb + row;
                    if (m < M && n < ldy) {
                        const float v[4] = {v4.x, v4.y, v4.z, v4.w};
                        float u[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (epi.res_f32) {
                            const float* rp = epi.res_f32 + (int64_t)m * epi.ldr + n;
                            if (rwide && n + 3 < N) {
                                const float4 r4 = *reinterpret_cast<const float4*>(rp);
                                u[0] = r4.x; u[1] = r4.y; u[2] = r4.z; u[3] = r4.w;
                            } else {
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (n + e < N) u[e] = rp[e];
                            }
                        }
                        uint32_t rword = 0;
                        if (epi.res_codes && n < N)
                            rword = *reinterpret_cast<const uint32_t*>(epi.res_codes + (int64_t)rrow[a][i] * epi.ldrc + n);
                        uint32_t word = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            int q = 0;
                            if (n + e < N) {
                                const float x0 = (epi.relu == 2 && v[e] < 0.0f) ? 0.0f : v[e];   // ReLU before the BatchNorm
                                // folded: two roundings (-ffp-contract=off); device form: fma(fl(fl(x - mean) * rs), weight, bias)
                                float t;
                                if constexpr (DEVBN) t = __builtin_fmaf((x0 - ral[e]) * rbe[e], al[e], be[e]);
                                else t = x0 * al[e] + be[e];
                                if (epi.res_f32) t = t + ((!DEVBN && epi.ralpha) ? u[e] * ral[e] + rbe[e] : u[e]);
                                if (epi.res_codes) t = t + epi.rscale * (float)(int8_t)(rword >> (8 * e));
                                if (epi.relu == 1) t = t < 0.0f ? 0.0f : t;
                                const float qf = rintf(epi.levels * t);
                                if (!(qf >= -127.0f && qf <= 127.0f)) bad = 1; else q = (int)qf;
                            }
                            word |= (uint32_t)(uint8_t)(int8_t)q << (8 * e);
                        }
                        *reinterpret_cast<uint32_t*>(Q + (int64_t)orow[a][i] * ldy + n) = word;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // pad bytes past the last column tile (row strides rounded beyond the tile width)
                // (a halo plane has ldy == Cout rounded up to 16: no bytes past the tile columns)
                if (!halo && b == C::TNW - 1 && wave_n == C::WN - 1 && n0 + C::TN >= N && lane < 32 && mb + lane < M)
                    for (int c = n0 + C::TN; c < ldy; c += 4)
                        *reinterpret_cast<uint32_t*>(Q + (int64_t)(mb + lane) * ldy + c) = 0u;
            }
        }
        if (__any(bad) && lane == 0) atomicOr(epi.overflow, 1);
        return;
    }
    if (epi.alpha) {
        // threshold bits: a v_cmp over the wave yields, per accumulator register, the 32-channel word of
        // two output rows (lanes 0-31 -> row R, lanes 32-63 -> row R + 4); lane i keeps row i's word and
        // one 32-lane store per 32x32 tile writes them.  Channels >= N compare 0 < 0 -> bit 0.
        uint32_t* B = reinterpret_cast<uint32_t*>(Y);
#pragma unroll
        for (int b = 0; b < C::TNW; ++b) {
            const int nb = n0 + (wave_n * C::TNW + b) * 32;
            const int n = nb + lrow;
            const float bv = (bias && n < N) ? bias[n] : 0.0f;
            const float al = n < N ? epi.alpha[n] : 0.0f, nbe = n < N ? -epi.beta[n] : 0.0f;
            // integer-threshold form: channels >= N compare acc < -inf -> bit 0
            const float thr = (epi.thr && n < N) ? epi.thr[n] : -3.0e38f;
            const unsigned long long negmask = epi.thr ? __ballot(al < 0.0f) : 0ull;
            // the 32-channel sign words of the 32 rows of accumulator tile (a, b): lane i (< 32) ends up with row i's word
            auto word_of = [&](int a, auto use_thr) -> uint32_t {
                uint32_t myword = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    unsigned long long mask;
                    if constexpr (decltype(use_thr)::value) {
                        mask = __ballot((float)acc[a][b][r] < thr);     // channels with alpha < 0 are flipped below
                    } else {
                        const float t = E::out(acc[a][b][r], scale, bv);
                        // fl(fl(t*al) + be) < 0  <=>  fl(t*al) < -be: an IEEE sum of two floats has the sign of the
                        // exact sum (a non-zero exact sum is a multiple of the smallest subnormal and cannot round to
                        // zero), NaN and inf - inf compare false on both sides — one VALU op less per register
                        mask = __ballot(t * al < nbe);
                    }
                    const int R = (r & 3) + 8 * (r >> 2);
                    // v_writelane: the two halves of the (scalar) ballot straight into lanes R and R + 4 (a compare + select
                    // each before).  gfx950 does not interlock a VALU-written SGPR read by the next VALU (2 wait states)
                    // and the hazard recogniser does not look inside inline asm: the s_nop covers the v_cmp -> first
                    // write, the first write covers the second (they are chained through myword).
                    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(myword) : "s"((uint32_t)mask), "n"(R));
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(myword) : "s"((uint32_t)(mask >> 32)), "n"(R + 4));
                }
                // every row's word holds the same 32 channels: one xor flips the alpha < 0 channels of all rows
                if constexpr (decltype(use_thr)::value) myword ^= (uint32_t)negmask;
                return myword;
            };
#pragma unroll
            for (int a = 0; a < C::TMW; ++a) {
                const uint32_t myword = epi.thr ? word_of(a, std::true_type{}) : word_of(a, std::false_type{});
                const int m = m0 + (wave_m * C::TMW + a) * 32 + lane;
                const int wcol = nb >> 5;
                if constexpr (C::CONV) if (epi.mode == 3) {
                    // the sign word of (pixel m, channels nb..nb+31) as 32 fp4 nibbles = 4 words of the next conv's
                    // pixel plane (what qt_bits_to_nib_pad would produce from the bit plane in a second pass)
                    int cgrp = wcol, left = N - nb, dyx = 0;
                    bool live = wcol * 4 < ldy;
                    if (epi.d2s_cout) {
                        dyx = nb / epi.d2s_cout;
                        const int cb = nb - dyx * epi.d2s_cout;
                        cgrp = cb >> 5;
                        left = epi.d2s_cout - cb;
                        live = nb < N;
                    }
                    if (lane < 32 && m < M && live) {
                        int orow = m;
                        if (epi.ohy | epi.ohx | epi.d2s_cout) {
                            const unsigned um = (unsigned)m;
                            unsigned img, ho, wo;
                            if (cg.sh_w >= 0) {
                                img = um >> cg.sh_hw;
                                const unsigned rem = um & (unsigned)(cg.Ho * cg.Wo - 1);
                                ho = rem >> cg.sh_w;
                                wo = rem & (unsigned)(cg.Wo - 1);
                            } else {
                                img = epi.magic_hw ? (unsigned)__umul64hi((unsigned long long)um, epi.magic_hw) : um;
                                const unsigned rem = um - img * (unsigned)(cg.Ho * cg.Wo);
                                ho = epi.magic_w ? (unsigned)__umul64hi((unsigned long long)rem, epi.magic_w) : rem;
                                wo = rem - ho * (unsigned)cg.Wo;
                            }
                            const unsigned zs = epi.d2s_cout ? 2u : 1u;
                            const unsigned oh = zs * ho + (unsigned)(dyx >> 1), ow = zs * wo + (unsigned)(dyx & 1);
                            orow = (int)((img * (zs * (unsigned)cg.Ho + 2u * (unsigned)epi.ohy) + oh + (unsigned)epi.ohy) *
                                             (zs * (unsigned)cg.Wo + 2u * (unsigned)epi.ohx) + ow + (unsigned)epi.ohx);
                        }
                        uint4 o;
                        if (left >= 32) {        // wave-uniform: all 32 channels exist, the magnitude nibbles are constant
                            o.x = 0x22222222u | (spread8(myword) << 3);
                            o.y = 0x22222222u | (spread8(myword >> 8) << 3);
                            o.z = 0x22222222u | (spread8(myword >> 16) << 3);
                            o.w = 0x22222222u | (spread8(myword >> 24) << 3);
                        } else {
                            const uint32_t mw = left > 0 ? ((1u << left) - 1u) : 0u;
                            const uint32_t sw = myword & mw;
                            o.x = (spread8(mw) << 1) | (spread8(sw) << 3);
                            o.y = (spread8(mw >> 8) << 1) | (spread8(sw >> 8) << 3);
                            o.z = (spread8(mw >> 16) << 1) | (spread8(sw >> 16) << 3);
                            o.w = (spread8(mw >> 24) << 1) | (spread8(sw >> 24) << 3);
                        }
                        *reinterpret_cast<uint4*>(B + (int64_t)orow * ldy + cgrp * 4) = o;
                    }
                    continue;
                }
                if (lane < 32 && m < M && wcol < ldy) B[(int64_t)m * ldy + wcol] = myword;
                // the row's pad words past the last column tile (ldy rounds ceil(N/32) up to 4) are zeroed here, so
                // the plane needs no memset
                if (b == C::TNW - 1 && wave_n == C::WN - 1 && n0 + C::TN >= N && lane < 32 && m < M)
                    for (int wc = (n0 + C::TN) >> 5; wc < ldy; ++wc) B[(int64_t)m * ldy + wc] = 0u;
            }
        }
    } else if (wide) {
        float* T = reinterpret_cast<float*>(smem) + wave * 1024;
        const bool stream_out = (int64_t)M * N >= (int64_t)(8 << 20);     // >= 32 MiB of fp32: more than the eight L2s hold
#pragma unroll
        for (int b = 0; b < C::TNW; ++b) {
            const int nb = n0 + (wave_n * C::TNW + b) * 32;
            const float bv = (bias && nb + lrow < N) ? bias[nb + lrow] : 0.0f;
#pragma unroll
            for (int a = 0; a < C::TMW; ++a) {
                const int mb = m0 + (wave_m * C::TMW + a) * 32;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    T[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * 32 + lrow] = E::out(acc[a][b][r], scale, bv);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = i * 8 + (lane >> 3), c4 = (lane & 7) * 4;
                    const float4 v = *reinterpret_cast<const float4*>(T + row * 32 + c4);
                    const int m = mb + row, n = nb + c4;
                    if (m < M && n < N) {
                        float* dst = Y + (int64_t)m * ldy + n;
                        if (stream_out) {
                            // a result larger than the L2s is not re-read from them: write-through (sc1) stores leave no
                            // dirty lines for the end-of-kernel write-back (4096^2 fp32: 40.5 -> 38.5 us per launch; `nt`
                            // measured neutral).  s_nop: the store reads its data registers late and the hazard
                            // recogniser does not look inside inline asm.
                            typedef float epi_v4f __attribute__((ext_vector_type(4)));
                            const epi_v4f ev = {v.x, v.y, v.z, v.w};
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(ev) : "memory");
                        } else {
                            *reinterpret_cast<float4*>(dst) = v;
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < C::TNW; ++b) {
            const int n = n0 + (wave_n * C::TNW + b) * 32 + lrow;
            const float bv = (bias && n < N) ? bias[n] : 0.0f;
#pragma unroll
            for (int a = 0; a < C::TMW; ++a) {
                const int mb = m0 + (wave_m * C::TMW + a) * 32 + 4 * lhalf;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m < M && n < N) Y[(int64_t)m * ldy + n] = E::out(acc[a][b][r], scale, bv);
                }
            }
        }
    }
    if constexpr (C::ABL >= 5) {   // profiling only: waves 0 and 4 overwrite the head of their own first Y row
        dbg_wall[3] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg_wall[4] = wall_clock64();
        if (lane == 0 && wave_n == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(Y + (int64_t)(m0 + wave_m * C::TMW * 32) * ldy + n0);
            for (int i = 0; i < 8; ++i) o[i] = dbg_ts[i];
            o[8] = dbg_end;
            for (int i = 0; i < 5; ++i) o[9 + i] = dbg_wall[i];
            o[14] = (unsigned long long)dbg_simd;
            o[15] = dbg_end - dbg_loop0;
        }
    }
}

template <class C>
int launch_cfg(const uint32_t* Xn, int64_t ldxp, const uint32_t* Wn, int64_t ldwp, const float* bias,
               float scale, const float* scale_dev, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
               qt_stream_t stream, const ConvArgs& cg = ConvArgs{}, const EpiArgs& epi = EpiArgs{}) {
    const int64_t gy = (M + C::TM - 1) / C::TM, gx = (N + C::TN - 1) / C::TN;
    if (gx * gy > (1ll << 30)) return QT_ERR_UNSUPPORTED;
    unsigned grid = (unsigned)((gx * gy + 7) / 8 * 8);
    if (C::CONV && (epi.mode == 2 || epi.mode == 3) && (epi.ohy | epi.ohx)) grid += 64;   // border-zeroing workgroups
    // > 64 KiB of dynamic LDS needs the opt-in attribute (per kernel and device: qt_ensure_dyn_lds raises it once)
    // VALID conv: + the tap table, one 4-byte offset per (stage, chunk)
    const int lds_bytes = C::LDS_BYTES + (C::VALID ? ((cg.kbytes + C::STAGE_BYTES - 1) / C::STAGE_BYTES) * C::CHUNKS * 4 : 0) +
                          (C::E::TAPS ? (epi.ntaps + 1 + 3) / 4 * 16 : 0) +      // E::TAPS: + the per-tap factor table
                          (elem_rows<typename C::E>::value ? (epi.ntaps + 1) * C::TM * 4 : 0);   // ROWS: + the per-row factor table
    if (lds_bytes > 160 * 1024) return QT_ERR_UNSUPPORTED;
    if constexpr (C::E::CODE_EPI && C::CONV) {
        if ((epi.mode == 2 || epi.mode == 4) && epi.bn_stats) {
            static QtLdsOnce once_bn;
            if (qt_ensure_dyn_lds(once_bn, reinterpret_cast<const void*>(mfma_gemm_kernel<C, true>), lds_bytes) != QT_OK) return QT_ERR_LAUNCH;
            hipLaunchKernelGGL((mfma_gemm_kernel<C, true>), dim3(grid, 1), dim3(C::NTHREADS), lds_bytes, (hipStream_t)stream, Xn,
                               ldxp, Wn, ldwp, bias, scale, scale_dev, Y, ldy, (int)M, (int)N, (int)K, cg, epi);
            return qt_check_launch();
        }
    }
    static QtLdsOnce once;
    if (qt_ensure_dyn_lds(once, reinterpret_cast<const void*>(mfma_gemm_kernel<C>), lds_bytes) != QT_OK) return QT_ERR_LAUNCH;
    const unsigned nz = (!C::CONV && cg.z_nslice > 0) ? (unsigned)(cg.z_nslice * cg.H) : 1u;   // GEMM batch: cg.H = taps
    hipLaunchKernelGGL(mfma_gemm_kernel<C>, dim3(grid, nz), dim3(C::NTHREADS),
                       lds_bytes, (hipStream_t)stream, Xn, ldxp, Wn, ldwp, bias, scale, scale_dev, Y, ldy,
                       (int)M, (int)N, (int)K, cg, epi);
    return qt_check_launch();
}

// tile shapes: 256x256 (wave 128x64), 256x128 (wave 128x32), 256x64 (wave 64x32)
template <class E, int PIPE, int ABL = 0> using Cfg256 = GemmCfg<E, 2, 4, 4, 2, PIPE, ABL>;
template <class E, int PIPE> using Cfg128 = GemmCfg<E, 2, 4, 4, 1, PIPE>;
template <class E, int PIPE> using Cfg64 = GemmCfg<E, 4, 2, 2, 1, PIPE>;
template <class E, int PIPE> using Cfg192 = GemmCfg<E, 4, 2, 2, 3, PIPE>;
// skinny GEMMs (M <= 256: FC layers): 128x64 tiles (2x the workgroups) with 256-byte stages (half the latency-bound
// stage round trips of the K loop)
template <class E> using CfgSkinny = GemmCfg<E, 4, 2, 1, 1, 1, 0, 256>;
template <class E> using CfgSkinny512 = GemmCfg<E, 2, 2, 1, 1, 1, 0, 512>;   // 64x64 tiles, 512-byte stages, 4 waves   // 256x192 (wave 64x96): N = 576, 1152, ...

// ping-pong configurations (64-byte stages, ring of 4)
template <class E, int ABL = 0> using PP256 = GemmCfg<E, 2, 4, 4, 2, 2, ABL, 64, false>;
template <class E> using PP128 = GemmCfg<E, 2, 4, 4, 1, 2, 0, 64, false>;
template <class E> using PP192 = GemmCfg<E, 4, 2, 2, 3, 2, 0, 64, false>;
template <class E> using PP384x192 = GemmCfg<E, 4, 2, 3, 3, 2, 0, 64, false>;   // wave tile 96x96: 6 fragment reads per 9 MFMAs
template <class E> using PP64 = GemmCfg<E, 4, 2, 2, 1, 2, 0, 64, false>;
template <class E> using ConvPP256 = GemmCfg<E, 2, 4, 4, 2, 2, 0, 64, 1>;
template <class E> using ConvPP192 = GemmCfg<E, 4, 2, 3, 3, 2, 0, 64, 1>;   // 384x192 tile: wave tile 96x96, 6 reads per 9 MFMAs
template <class E> using ConvPP128 = GemmCfg<E, 2, 4, 4, 1, 2, 0, 64, 1>;
template <class E> using ConvPP256x192 = GemmCfg<E, 4, 2, 2, 3, 2, 0, 64, 1>;   // 256x192 ping-pong (the GEMM's PP192 as a conv)
#ifdef QT_PROFILING_VARIANTS
template <class E> using ConvPP192Stamps = GemmCfg<E, 4, 2, 3, 3, 2, 5, 64, 1>;   // profiling builds only (conv variant 3)
#endif
template <class E> using ConvPP64 = GemmCfg<E, 4, 2, 2, 1, 2, 0, 64, 1>;

// implicit-conv configurations (pipelined kernel only)
template <class E> using Conv256 = GemmCfg<E, 2, 4, 4, 2, 1, 0, 128, 1>;
template <class E> using Conv128 = GemmCfg<E, 2, 4, 4, 1, 1, 0, 128, 1>;
template <class E> using Conv64 = GemmCfg<E, 4, 2, 2, 1, 1, 0, 128, 1>;
template <class E> using Conv192 = GemmCfg<E, 4, 2, 2, 3, 1, 0, 128, 1>;
// small maps with padding (the late stages of a CIFAR ResNet: M = 16384 / 4096 output pixels): the 256-row tiles above leave
// most CUs idle — 128x128 tiles (wave 64x32) quadruple the workgroups; 64x64 with 512-byte stages below 64 big tiles
template <class E> using Conv128x128 = GemmCfg<E, 2, 4, 2, 1, 1, 0, 128, 1>;
template <class E> using ConvSkinny = GemmCfg<E, 2, 2, 1, 1, 1, 0, 512, 1>;

// ... and on un-padded / physically padded planes (CONV_ = 2)
template <class E> using ConvV256 = GemmCfg<E, 2, 4, 4, 2, 1, 0, 128, 2>;
template <class E> using ConvV128 = GemmCfg<E, 2, 4, 4, 1, 1, 0, 128, 2>;
template <class E> using ConvV64 = GemmCfg<E, 4, 2, 2, 1, 1, 0, 128, 2>;
template <class E> using ConvV192 = GemmCfg<E, 4, 2, 2, 3, 1, 0, 128, 2>;
template <class E> using ConvVPP256 = GemmCfg<E, 2, 4, 4, 2, 2, 0, 64, 2>;
template <class E> using ConvV64x2 = GemmCfg<E, 4, 2, 2, 1, 1, 0, 64, 2, 3>;    // 256x64 tile, 64-byte stages, 3 workgroups / CU
template <class E> using ConvV128x2 = GemmCfg<E, 2, 4, 4, 1, 1, 0, 64, 2, 2>;   // 256x128 tile, same
template <class E> using ConvVPP192 = GemmCfg<E, 4, 2, 3, 3, 2, 0, 64, 2>;
template <class E> using ConvVPP128 = GemmCfg<E, 2, 4, 4, 1, 2, 0, 64, 2>;      // 256x128 ping-pong (per-tap scaled convs)
template <class E> using ConvVPP256x192 = GemmCfg<E, 4, 2, 2, 3, 2, 0, 64, 2>;   // 256x192 ping-pong on un-padded / physically padded planes
// small M (small-batch inference, late layers of small images): few tiles and a long, latency-bound K loop — 64x64 tiles
// with 512-byte stages, as the skinny GEMM configuration (weight rows must be padded to whole 512-byte stages).  Taken
// for M <= 4096, and beyond that while the standard tiling leaves CUs idle (< 256 tiles) and the weight re-reads of the
// small row tiles ((M / 64) x the weight matrix through L2) stay under 256 MB
template <class E> using ConvVSkinny = GemmCfg<E, 2, 2, 1, 1, 1, 0, 512, 2>;
// ... and 128x128 tiles where those already give every CU a workgroup (M = 16384 pixels x 256 channels: 256 tiles): half the
// L2 -> LDS traffic of the 64x64 tiles, which is what bounds these layers
template <class E> using ConvV128x128 = GemmCfg<E, 2, 4, 2, 1, 1, 0, 128, 2>;
// ... and 128x64 tiles (256-byte stages, the skinny GEMM's shape) where THOSE fill the chip (M = 4096 pixels x 512 channels)
template <class E> using ConvV128x64 = GemmCfg<E, 4, 2, 1, 1, 1, 0, 256, 2>;
// ... the same two tiles with a DEEPER DMA ring (4 x 32 KiB / 3 x 48 KiB of stage buffers, 3 / 2 stages in flight): one workgroup per
// CU and 18 stages whose matrix time (~0.2 us) is a fraction of the L2 round trip — the double-buffered loop ran at ~0.8 us per stage
template <class E> using ConvV128x128D = GemmCfg<E, 2, 4, 2, 1, 4, 0, 128, 2>;
template <class E> using ConvV128x64D = GemmCfg<E, 4, 2, 1, 1, 3, 0, 256, 2>;
#ifdef QT_PROFILING_VARIANTS
template <class E> using ConvVPP192Stamps = GemmCfg<E, 4, 2, 3, 3, 2, 5, 64, 2>;   // profiling builds only
#endif

// tile width (256 / 192 / 128 / 64) that wastes the fewest padded columns; ties go to the wider tile
int pick_tile_n(int64_t N) {
    int best = 256;
    int64_t best_pad = (N + 255) / 256 * 256;
    const int cands[3] = {192, 128, 64};
    for (int c : cands) {
        const int64_t pad = (N + c - 1) / c * c;
        if (pad < best_pad) { best = c; best_pad = pad; }
    }
    return best;
}

// GEMM tile width: the padding-minimal width, narrowed while the launch would leave most of the 256 CUs
// without a tile (a workgroup walks the whole K loop alone, so a 256x4096x9216 problem on 16 wide tiles
// takes 50 us and on 64 narrow ones 24 us).
int pick_tile_n_gemm(int64_t M, int64_t N) {
    int tn = pick_tile_n(N);
    const int64_t mt = (M + 255) / 256;
    while (tn > 64 && mt * ((N + tn - 1) / tn) < 160) tn = tn == 256 ? 128 : 64;
    return tn;
}

// 192-wide column tiles come with 256 or 384 rows.  The 384-row tile does 50 % more work per workgroup at a
// better MFMA : fragment-read ratio; it wins unless it leaves CUs idle (fewer tiles than the 256 CUs) or adds a
// partial round.  Cost model: rounds of 256 concurrent workgroups x rows per tile; ties go to 384.
bool prefer_384_rows(int64_t M, int64_t N) {
    const int64_t nt = (N + 191) / 192;
    const int64_t c256 = (((M + 255) / 256) * nt + 255) / 256 * 256;
    const int64_t c384 = (((M + 383) / 384) * nt + 255) / 256 * 384;
    return c384 <= c256;
}

// Argument checks and geometry of an implicit-GEMM conv launch, shared by the entry points of mfma_gemm.hip and conv_taps.hip.
// (hy, hx): halo of the INPUT plane, [N][H + 2hy][W + 2hx][Cw] with a zero border: a conv whose padding fits in the halo runs as
// the un-padded conv on the window that starts (hy - ph, hx - pw) into the plane (P is advanced accordingly).  Returns QT_OK, an
// error code, or 1 = nothing to do.  elem: 0 = fp4 nibble planes, 1 = int8 code planes, 2 = bf16 (triple) planes, 3 = fp16 (pair) planes.
static int conv_prepare(int elem, const uint32_t*& P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw, int64_t kh, int64_t kw,
                        int64_t sh, int64_t sw, int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldwp,
                        const float* Y, int64_t ldy, int64_t Cout, EpiArgs& epi, int64_t hy, int64_t hx, ConvArgs& cg, bool& valid,
                        int64_t& M, int64_t& K, int64_t& kwords) {
    if (hy < 0 || hx < 0 || ((hy | hx) && (ph > hy || pw > hx))) return QT_ERR_INVALID_ARG;
    if (Nimg < 0 || H <= 0 || W <= 0 || Cw <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 ||
        dw <= 0 || ph < 0 || pw < 0 || Cout < 0 || elem < 0 || elem > 3)
        return QT_ERR_INVALID_ARG;
    const int64_t Ho = (H + 2 * ph - dh * (kh - 1) - 1) / sh + 1, Wo = (W + 2 * pw - dw * (kw - 1) - 1) / sw + 1;
    if (Ho <= 0 || Wo <= 0) return QT_ERR_INVALID_ARG;
    M = Nimg * Ho * Wo;
    if (M == 0 || Cout == 0) return 1;
    if (!P || !Wmat || !Y || ldy < (epi.mode == 2 ? ((Cout + 3) & ~3ll) : epi.mode == 3 ? ((epi.d2s_cout ? epi.d2s_cout : Cout) + 31) / 32 * 4 : epi.alpha ? (Cout + 31) / 32 : Cout))
        return QT_ERR_INVALID_ARG;
    kwords = kh * kw * Cw;                 // words per (virtual) im2col row
    if ((Cw & 3) || (ldwp & 31) || ldwp < kwords || !qt_aligned16(P) || !qt_aligned16(Wmat)) return QT_ERR_ALIGNMENT;
    const int64_t Hp = H + 2 * hy, Wp = W + 2 * hx;
    if (M > INT32_MAX || kwords * 4 >= (1 << 20) || Cout * ldwp * 4 >= (1ll << 31) || Hp > 32767 || Wp > 32767 ||
        Hp * Wp * Cw * 4 >= (1ll << 31))   // per-image plane bytes: 32-bit tap offsets
        return QT_ERR_UNSUPPORTED;
    if ((epi.mode == 2 || epi.mode == 3) && (epi.ohy | epi.ohx | epi.rhy | epi.rhx | epi.d2s_cout)) {
        const int64_t zs = epi.d2s_cout ? 2 : 1;
        if (Nimg * (zs * Ho + 2 * epi.ohy) * (zs * Wo + 2 * epi.ohx) > INT32_MAX || Nimg * (Ho + 2 * epi.rhy) * (Wo + 2 * epi.rhx) > INT32_MAX)
            return QT_ERR_UNSUPPORTED;
        const unsigned long long hw = (unsigned long long)(Ho * Wo), wo_ = (unsigned long long)Wo;
        epi.magic_hw = hw > 1 ? ~0ull / hw + 1 : 0;     // ceil(2^64 / d) for d > 1 (exact quotients for 32-bit numerators)
        epi.magic_w = wo_ > 1 ? ~0ull / wo_ + 1 : 0;
    }
    const int64_t kbytes = kwords * 4;
    K = elem == 0 ? kbytes * 2 : (elem == 1 ? kbytes : kbytes / 2);   // elements
    if (elem == 0 && K >= (1 << 24)) return QT_ERR_UNSUPPORTED;
    cg.H = (int)H; cg.W = (int)W; cg.Ho = (int)Ho; cg.Wo = (int)Wo; cg.kh = (int)kh; cg.kw = (int)kw;
    cg.sh = (int)sh; cg.sw = (int)sw; cg.ph = (int)ph; cg.pw = (int)pw; cg.dh = (int)dh; cg.dw = (int)dw;
    cg.cpp = (int)(Cw / 4);
    cg.kbytes = (int)(kwords * 4);
    cg.magic_cpp = cg.cpp > 1 ? (unsigned)((1ull << 32) / (unsigned)cg.cpp + 1) : 0;
    cg.magic_kw = kw > 1 ? (unsigned)((1ull << 32) / (unsigned)kw + 1) : 0;
    cg.sh_hw = cg.sh_w = -1;
    if (((Ho * Wo) & (Ho * Wo - 1)) == 0 && (Wo & (Wo - 1)) == 0) {
        cg.sh_hw = __builtin_ctzll((unsigned long long)(Ho * Wo));
        cg.sh_w = __builtin_ctzll((unsigned long long)Wo);
    }
    // un-padded conv on a plane < 4 GiB: every tap of every window is in bounds -> 32-bit offsets, no checks
    valid = ph == 0 && pw == 0 && Nimg * H * W * Cw * 4 < (1ll << 32) && kwords * 4 <= 32768;
    if (hy | hx) {
        valid = Nimg * Hp * Wp * Cw * 4 < (1ll << 32) && kwords * 4 <= 32768;
        if (!valid) return QT_ERR_UNSUPPORTED;      // the caller strips the halo and uses the bounds-checked kernels
        cg.H = (int)Hp; cg.W = (int)Wp; cg.ph = cg.pw = 0;
        P += ((hy - ph) * Wp + (hx - pw)) * Cw;
    }
    return QT_OK;
}

}  // namespace

// The tile configuration CFG for element E — or, for fp4 convs whose threshold epilogue has integer thresholds (swapt), its
// weights-as-rows instantiation (ElemFp4T, mfma_gemm_kernel.h).  Only the four configurations the fused AlexNet / VGG-16 chains
// launch exist in that form; every other one keeps the compare epilogue.
template <template <class> class CFG, class E, class... A>
int launch_cfg_t(bool swapt, A&&... a) {
    if constexpr (std::is_same<E, ElemFp4>::value) {
        if (swapt) return launch_cfg<CFG<ElemFp4T>>(a...);
    }
    return launch_cfg<CFG<E>>(a...);
}
