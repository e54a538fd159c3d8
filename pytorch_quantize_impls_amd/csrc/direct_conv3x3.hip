// Direct 3x3 / stride 1 / padding 1 conv of a +-1 (fp4 nibble) activation on a HALO plane, threshold-bit epilogue
// (include/qt_hip.h: qt_conv3x3_direct_nib).  The layer shape of VGG / ResNet stacks at large spatial size and few
// channels, where the implicit-GEMM gather of mfma_gemm.hip is bound by L2 -> LDS traffic (every input pixel is
// fetched kh*kw = 9 times: 16.5 TB/s for VGG-16 conv2 at batch 256).
//
// Layout trick: the input plane carries a 1-pixel zero halo, [N][H+2][W+2][pixel], and the M dimension is tiled over
// POSITIONS q of that padded plane (borders included).  The 3x3 window of position q is  P[q + (i-1)*Wp + (j-1)],
// so the 256 positions of a tile need three CONTIGUOUS runs of 258 pixels — 25 KB for 64 channels instead of the
// 147 KB of taps the implicit GEMM gathers.  A workgroup keeps the whole weight matrix in LDS (18 KB for 64 -> 64),
// walks position tiles persistently, loads the three runs once per tile and reads the A fragment of tap (i, j) as an
// LDS load at patch[i][pos + j].
//
// STATUS (round 1): bit-identical to the implicit-GEMM kernels (tests/test_gpu_parity.py).  tools/bench_direct_conv.py,
// batch 256, nibble-plane / bit-plane output:   64 -> 64 at 224^2   350 / 324 us  (implicit GEMM 615 / 505)
//                                               64 -> 128 at 112^2  168 / 143 us  (247 / 186)
//                                               128 -> 128 at 112^2 275 / 256 us  (321 / 255)
// What it took after the first (1.6x slower than the implicit GEMM) version: the tap loop rolled over the kernel rows —
// fully unrolled, the scheduler hoisted all 36 fragment reads (198 VGPRs, 2 waves per SIMD) —, the next tile's patch
// prefetched into registers right after the barrier that publishes the current one, and 8 waves (4 position x 2 column)
// per workgroup for 128 input channels, whose LDS footprint allows one workgroup per CU, and a contiguous range of
// tiles per workgroup (consecutive tiles share two of their three input rows in that workgroup's L2).
// Not done yet: LDS-DMA for the patch, keeping the two shared rows in LDS across consecutive tiles.
//
// fp4 MFMA operand layout as in mfma_gemm.hip: v_mfma_scale_f32_32x32x64_f8f6f4, lane l supplies 16 bytes (32
// nibbles) of row l % 32: K elements 0..31 from lanes 0..31, 32..63 from lanes 32..63; accumulator register r of lane
// l = (row (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), column l & 31).
#include <cstdlib>
#include <type_traits>

#include "qt_common.h"

namespace {

typedef int d3_v8i __attribute__((ext_vector_type(8)));
typedef float d3_v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ d3_v16f d3_mfma(const uint4& a, const uint4& b, d3_v16f c) {
    const d3_v8i av = (d3_v8i){(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
    const d3_v8i bv = (d3_v8i){(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// bf16 triple planes (real-valued first layers: x = hi + mid + lo exactly, weights +-1 / 0 replicated three times):
// v_mfma_f32_32x32x16_bf16 takes the same 16 bytes per lane (8 bf16: K 0..7 from lanes 0..31, 8..15 from lanes 32..63)
__device__ __forceinline__ d3_v16f d3_mfma_bf16(const uint4& a, const uint4& b, d3_v16f c) {
    typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
    bf8 av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c, 0, 0, 0);
}

// fp16 pair planes (real-valued first layers, two terms: x / s = hi + lo, weights +-1 / 0 replicated twice; round 4): a pixel of
// <= 4 channels is ONE 16-byte chunk, so the two lane halves of v_mfma_f32_32x32x16_f16 (K 0..7 from lanes 0..31, 8..15 from lanes
// 32..63) take two different TAPS: 5 MFMAs cover the 9 taps (+ one zero slot) where the triple form needs 9
__device__ __forceinline__ d3_v16f d3_mfma_f16(const uint4& a, const uint4& b, d3_v16f c) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
}

// int8 code planes (DoReFa activations x +-1 weight codes): v_mfma_i32_32x32x32_i8, 16 bytes = 16 K elements per lane
typedef int d3_v4i __attribute__((ext_vector_type(4)));
typedef int d3_v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ d3_v16i d3_mfma_i8(const uint4& a, const uint4& b, d3_v16i c) {
    const d3_v4i av = (d3_v4i){(int)a.x, (int)a.y, (int)a.z, (int)a.w};
    const d3_v4i bv = (d3_v4i){(int)b.x, (int)b.y, (int)b.z, (int)b.w};
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, c, 0, 0, 0);
}

__device__ __forceinline__ uint32_t d3_spread8(uint32_t b) {
    uint32_t t = b & 0xFFu;
    t = (t | (t << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return t;
}

struct D3Args {
    const unsigned char* P;   // input nibble halo plane, CPP * 16 bytes per pixel
    const unsigned char* Wm;  // [Cout][ldw bytes], K order: tap-major (i, j), then the pixel's chunks
    const float* bias;
    const float* alpha;
    const float* beta;
    uint32_t* out;
    long long total;          // N * Hp * Wp positions
    int H, W, Hp, Wp, Cout, ldw, ldo, out_bits, buf_ok;
    int epi32;                // fp4: the lean epilogue applies (whole 32-channel blocks, output plane below 4 GiB, no pad words)
    unsigned out_bytes;       // of the output plane (epi32)
    unsigned m20_wp, m20_hp;  // ceil(2^20 / Wp), ceil(2^20 / Hp): exact quotients of numbers below 512 (planes narrower / lower than 256)
    unsigned long long magic_plane, magic_wp;   // ceil(2^64 / (Hp*Wp)), ceil(2^64 / Wp): exact 32-bit quotients
    // EL == 1 (int8 code planes, DoReFa code epilogue as qt_conv2d_implicit_codes; out = int8 halo plane, ldo in BYTES)
    float scale, rscale, levels;
    const float* scale_dev;
    const unsigned char* res_codes;   // residual code plane with the same halo geometry, ldrc bytes per pixel
    int ldrc, relu;
    int* overflow;
};

constexpr int D3_TM = 256, D3_RUN = D3_TM + 2;
constexpr int d3_wrow(int cpp) { return 9 * cpp * 16 + (cpp == 1 ? 32 : 16); }   // LDS pitch of a weight row (see the kernel)

// CPP: 16-byte chunks per input pixel (Cin = 32 * CPP); a workgroup = 4 (position) x WN (column) waves, each wave owns
// 64 positions x TNW 32-column blocks (Cout <= 32 * TNW * WN); OCC: waves per SIMD the register budget is sized for
template <int CPP, int TNW, int WN, int OCC, int EL, bool LEAN = false>
__global__ __launch_bounds__(256 * WN, OCC) void direct3x3_kernel(D3Args g) {
    constexpr int NT = 256 * WN;
    constexpr bool I8 = EL == 1, BF16 = EL == 2, F16 = EL == 3, FP4 = EL == 0;
    static_assert(!F16 || CPP == 1, "fp16 pair pixels are one 16-byte chunk");
    using acc_t = typename std::conditional<I8, d3_v16i, d3_v16f>::type;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // row pitch of the weight matrix: + 16 bytes (+ 32 for one-chunk pixels, whose 10th slot is data) puts the 16 rows of every
    // ds_read_b128 lane group on 16 different 16-byte slots of the 256-byte bank row
    constexpr int WROW = d3_wrow(CPP);
    constexpr int WBYTES = TNW * WN * 32 * WROW;
    unsigned char* wl = smem;
    unsigned char* patch = smem + WBYTES;          // [3][RUN][CPP chunks], chunk c of pixel px at c ^ swz(px)
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, wave_n = tid >> 8;
    const int lrow = lane & 31, lhalf = lane >> 5;
    // conflict-free fragment reads: a ds_read_b128 is served in four groups of 16 lanes — {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and
    // the same + 32 (MI355X_MICROARCH.md, LDS) — against a bank row of 256 bytes = 16 / CPP pixels, so a group holds pairs of pixels
    // 16 / CPP (and 3 x 16 / CPP) apart: their chunks must differ.  (Was px * CPP / 8, built for contiguous groups of 8 lanes on a
    // 128-byte row: every pixel-fragment read of the 64-channel layers took 8 LDS cycles instead of 4 — SQ_LDS_BANK_CONFLICT =
    // 2 cycles per MFMA at VGG conv1_2, profiles/r6_c5_direct_conv.md.)
    auto swz = [](int px) { return (px * CPP / 16) & (CPP - 1); };

    // ---- +-1 / 0 activations (FP4): the accumulators are exact integers, so the threshold test needs no compare at all (round 6).
    // bit = (u < theta) with u an integer  <=>  u - (ceil(theta) - 1/2) < 0: the accumulators START at -(ceil(theta) - 1/2) (exact: every
    // partial sum is a half-integer below 2^23, or |theta| is so large that no rounding can change the sign) and the bit is the SIGN BIT of
    // the result.  The weights are the MFMA's ROW operand here (rows assigned to channels so that lane half h, register r = channel
    // 16 h + r of a 32-channel block): a lane holds 16 channels of ONE position and collects their sign bits with one v_alignbit per value —
    // the ballot + two v_writelane per accumulator register of the first form made this kernel VALU-bound (15.6 VALU per MFMA, 265 us of
    // VALU issue against 128 us of matrix time at VGG conv1_2: profiles/r6_c5_direct_conv.md).  Channels with a negative slope get their
    // weights negated in LDS (-u < -theta').
    float* cinit_s = reinterpret_cast<float*>(smem + WBYTES + 3 * D3_RUN * CPP * 16);          // [TNW * WN * 32]
    unsigned* flip_s = reinterpret_cast<unsigned*>(cinit_s + TNW * WN * 32);
    if constexpr (FP4) {
        for (int n = tid; n < TNW * WN * 32; n += NT) {
            float ci = 1.0e30f;                                           // channel beyond Cout: acc' > 0, bit 0
            unsigned fl = 0u;
            if (n < g.Cout) {
                const float alv = g.alpha[n], nbv = -g.beta[n], bvv = g.bias ? g.bias[n] : 0.0f;
                auto key2f = [](unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); };
                auto pred = [&](unsigned k) { const float u = key2f(k); const float v = u + bvv; return v * alv < nbv; };
                const unsigned klo = 0x007fffffu, khi = 0xff800000u;      // keys of -inf, +inf
                float th;
                if (!(alv > 0.0f) && !(alv < 0.0f)) {
                    th = (alv == 0.0f && 0.0f < nbv) ? __uint_as_float(0x7f800000u) : __uint_as_float(0xff800000u);
                } else if (alv > 0.0f) {
                    unsigned lo = klo, hi = khi;
                    while (lo < hi) {
                        const unsigned mid = lo + ((hi - lo) >> 1);
                        if (!pred(mid)) hi = mid; else lo = mid + 1;
                    }
                    th = key2f(lo);
                } else {
                    fl = 1u;
                    if (!pred(khi)) {
                        th = __uint_as_float(0xff800000u);
                    } else {
                        unsigned lo = klo, hi = khi;
                        while (lo < hi) {
                            const unsigned mid = lo + ((hi - lo) >> 1);
                            if (pred(mid)) hi = mid; else lo = mid + 1;
                        }
                        th = -key2f(lo - 1);
                    }
                }
                // bit <=> u < th (u integer): start value -(ceil(th) - 0.5); th = +inf: always (start -> -huge), -inf / NaN: never
                if (th >= 3.0e38f) ci = -1.0e30f;                       // always: acc' = acc - 1e30 < 0
                else if (!(th > -3.0e38f)) ci = 1.0e30f;              // never (th = -inf)
                else ci = -(ceilf(th) - 0.5f);
            }
            cinit_s[n] = ci;
            flip_s[n] = fl;
        }
        __syncthreads();
    }
    // weights: resident for the whole launch
    for (int e = tid; e < TNW * WN * 32 * 9 * CPP; e += NT) {
        const int row = e / (9 * CPP), c = e - row * (9 * CPP);
        int src = row;
        if constexpr (FP4) {                       // LDS row (block, i) = MFMA row i of the block = channel 16 ((i / 4) % 2) + 4 (i / 8) + i % 4
            const int i = row & 31;
            src = (row & ~31) + 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
        }
        uint4 v = make_uint4(0, 0, 0, 0);
        if (src < g.Cout) v = *reinterpret_cast<const uint4*>(g.Wm + (long long)src * g.ldw + c * 16);
        if constexpr (FP4) {
            if (flip_s[src]) { v.x ^= 0x88888888u; v.y ^= 0x88888888u; v.z ^= 0x88888888u; v.w ^= 0x88888888u; }
        }
        *reinterpret_cast<uint4*>(wl + row * WROW + c * 16) = v;
    }
    if constexpr (F16) {     // the row's 16 pad bytes are tap slot 9 (the odd tap's partner): zero weights
        for (int row = tid; row < TNW * WN * 32; row += NT) *reinterpret_cast<uint4*>(wl + row * WROW + 9 * 16) = make_uint4(0, 0, 0, 0);
    }
    // fp16 pairs: this lane half's tap of pair p = 2 p + lhalf -> byte offset of its run / column in the patch, of its slot in a weight row
    [[maybe_unused]] int tpo[5], two[5];
#pragma unroll
    for (int p_ = 0; p_ < 5; ++p_) {
        const int tap = 2 * p_ + lhalf, tc = tap > 8 ? 8 : tap;          // slot 9 reads tap 8's pixel against zero weights
        tpo[p_] = ((tc / 3) * D3_RUN + tc % 3) * 16;
        two[p_] = tap * 16;
    }
    [[maybe_unused]] const float sc16 = (F16 && g.scale_dev) ? *g.scale_dev : 1.0f;       // power-of-two scale of the pair plane
    float al[TNW], nbe[TNW], bv[TNW];
#pragma unroll
    for (int b = 0; b < TNW; ++b) {
        const int n = (wave_n * TNW + b) * 32 + lrow;
        const bool in = n < g.Cout;
        al[b] = in ? g.alpha[n] : 0.0f;
        nbe[b] = in ? -g.beta[n] : 0.0f;
        bv[b] = (in && g.bias) ? g.bias[n] : 0.0f;
    }
    // threshold epilogue as ONE compare per value: bit = fl(fl(u + bias) * alpha) < -beta is monotone in u (u = the accumulator, times
    // the pair plane's power-of-two scale), so per channel it IS a comparison of u with one fp32 threshold — found once per
    // (persistent) workgroup by bisection over the ordered fp32 values WITH the epilogue's own arithmetic (exact by construction,
    // infinities and alpha <= 0 / NaN included; the scheme of csrc/conv_first_direct.hip).  alpha < 0: the bit is u > theta', tested
    // as -u < -theta' (e_sg flips the sign of u).  Was: add, multiply, compare per accumulator register
    [[maybe_unused]] float e_th[TNW];
    [[maybe_unused]] unsigned e_sg[TNW];
    // FP4: this lane's start values (channel (wave_n TNW + b) 32 + 16 lhalf + r) are re-read from LDS per tile — 16 TNW registers
    // more pushed the kernel into scratch (470 us instead of 278 at VGG conv1_2)
    if constexpr (!I8 && !FP4) {
#pragma unroll
        for (int b = 0; b < TNW; ++b) {
            const float alv = al[b], nbv = nbe[b], bvv = bv[b];
            e_th[b] = 0.0f;
            e_sg[b] = 0u;
            auto key2f = [](unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); };
            auto pred = [&](unsigned k) { const float u = key2f(k); const float v = u + bvv; return v * alv < nbv; };
            const unsigned klo = 0x007fffffu, khi = 0xff800000u;          // keys of -inf, +inf
            if (!(alv > 0.0f) && !(alv < 0.0f)) {
                // alpha == 0: +-0 < -beta for every finite u;  alpha NaN: never
                e_th[b] = (alv == 0.0f && 0.0f < nbv) ? __uint_as_float(0x7f800000u) : __uint_as_float(0xff800000u);
            } else if (alv > 0.0f) {                                      // first key whose bit is 0 (the bit of +inf is 0)
                unsigned lo = klo, hi = khi;
                while (lo < hi) {
                    const unsigned mid = lo + ((hi - lo) >> 1);
                    if (!pred(mid)) hi = mid; else lo = mid + 1;
                }
                e_th[b] = key2f(lo);
            } else {                                                      // first key whose bit is 1 (the bit of -inf is 0); none: never
                e_sg[b] = 0x80000000u;
                if (!pred(khi)) {
                    e_th[b] = __uint_as_float(0xff800000u);
                } else {
                    unsigned lo = klo, hi = khi;
                    while (lo < hi) {
                        const unsigned mid = lo + ((hi - lo) >> 1);
                        if (pred(mid)) hi = mid; else lo = mid + 1;
                    }
                    e_th[b] = -key2f(lo - 1);
                }
            }
            if constexpr (F16) e_th[b] *= 1.0f / sc16;                    // u = acc * s, s a power of two: compare acc with theta / s (exact)
        }
    }
    const long long ntiles = (g.total + D3_TM - 1) / D3_TM;
    const unsigned plane = (unsigned)(g.Hp * g.Wp);
    // the patch of the NEXT tile travels through registers while this tile computes (global latency off the critical
    // path): NLD 16-byte loads per thread, issued right after the barrier that publishes the current patch
    constexpr int NLD = (3 * D3_RUN * CPP + NT - 1) / NT;
    uint4 pre[NLD];
    // the input plane as a BUFFER (planes below 2 GiB: g.buf_ok): a thread's NLD chunk offsets relative to the tile's first run are
    // tile-invariant registers, the tile's base is a scalar, chunks outside the plane (the runs of the first / last positions) come
    // back as zeros from the bounds check — the 64-bit clamp-and-multiply address chain of the pointer form was 246 VALU instructions per
    // tile and thread, almost half of what made this kernel VALU-bound (profiles/r6_c5_direct_conv.md)
    typedef unsigned d3_u4 __attribute__((ext_vector_type(4)));
    [[maybe_unused]] unsigned voff[NLD];
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t prs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(g.P), 0, g.buf_ok ? (int)(g.total * CPP * 16) : 0, 0x00020000);
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e = tid + k * NT;
        const int run = e / (D3_RUN * CPP), rem = e - run * (D3_RUN * CPP);
        voff[k] = e < 3 * D3_RUN * CPP ? (unsigned)((run * g.Wp) * CPP * 16 + rem * 16) : 0x80000000u;
    }
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t ors =          // the output plane (lean fp4 epilogue: 32-bit offsets)
        __builtin_amdgcn_make_buffer_rsrc(g.out, 0, (int)(LEAN ? g.out_bytes : 0u), 0x00020000);
    auto fetch = [&](long long t) {
        const long long q0n = t * D3_TM;
        if (g.buf_ok) {
            // byte offset of position q0n - Wp - 1 (the first chunk of run 0): may be "negative" — 32-bit wrap-around puts exactly the
            // chunks in front of the plane beyond its extent
            const unsigned soff = (unsigned)((q0n - g.Wp - 1) * CPP * 16);
#pragma unroll
            for (int k = 0; k < NLD; ++k) {
                const d3_u4 v = __builtin_amdgcn_raw_buffer_load_b128(prs, (int)(voff[k] + soff), 0, 0);
                pre[k] = make_uint4(v.x, v.y, v.z, v.w);
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = tid + k * NT;
            const int run = e / (D3_RUN * CPP), rem = e - run * (D3_RUN * CPP);
            const int px = rem / CPP, c = rem - px * CPP;
            long long src = q0n + (long long)(run - 1) * g.Wp - 1 + px;
            src = src < 0 ? 0 : (src >= g.total ? g.total - 1 : src);      // only border / tail positions, never stored
            pre[k] = make_uint4(0, 0, 0, 0);
            if (e < 3 * D3_RUN * CPP) pre[k] = *reinterpret_cast<const uint4*>(g.P + (src * CPP + c) * 16);
        }
    };
    // every workgroup walks a CONTIGUOUS range of position tiles: consecutive tiles share two of their three input rows,
    // which then come from this workgroup's own L2 instead of being fetched into every XCD's L2
    const long long per_wg = (ntiles + gridDim.x - 1) / gridDim.x;
    const long long t_begin = (long long)blockIdx.x * per_wg;
    const long long t_end = t_begin + per_wg < ntiles ? t_begin + per_wg : ntiles;
    if (t_begin < t_end) fetch(t_begin);
    for (long long tile = t_begin; tile < t_end; ++tile) {
        const long long q0 = tile * D3_TM;
        __syncthreads();                            // the previous tile's fragment reads (and the weight fill) are done
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int e = tid + k * NT;
            const int run = e / (D3_RUN * CPP), rem = e - run * (D3_RUN * CPP);
            const int px = rem / CPP, c = rem - px * CPP;
            if (e < 3 * D3_RUN * CPP)
                *reinterpret_cast<uint4*>(patch + ((run * D3_RUN + px) * CPP + (c ^ swz(px))) * 16) = pre[k];
        }
        __syncthreads();
        if (tile + 1 < t_end) fetch(tile + 1);
        acc_t acc[2][TNW];
        constexpr bool CIN = FP4 && LEAN && CPP == 2;       // start values as the C operand of the first tap (no copies into both blocks)
        if constexpr (!CIN) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < TNW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if constexpr (FP4) acc[a][b][r] = cinit_s[(wave_n * TNW + b) * 32 + 16 * lhalf + r];
                    else acc[a][b][r] = 0;
                }
        }
        [[maybe_unused]] auto tap_step = [&](int i, int j, auto first) {
            if constexpr (CIN) {
            const int tap = i * 3 + j;
#pragma unroll
            for (int kk = 0; kk < CPP / 2; ++kk) {
                const int c = kk * 2 + lhalf;
                uint4 xf[2], wf[TNW];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int px = wave * 64 + a * 32 + lrow + j;
                    xf[a] = *reinterpret_cast<const uint4*>(patch + ((i * D3_RUN + px) * CPP + (c ^ swz(px))) * 16);
                }
#pragma unroll
                for (int b = 0; b < TNW; ++b)
                    wf[b] = *reinterpret_cast<const uint4*>(wl + ((wave_n * TNW + b) * 32 + lrow) * WROW + (tap * CPP + c) * 16);
                if constexpr (decltype(first)::value) {
                    acc_t cin[TNW];
#pragma unroll
                    for (int b = 0; b < TNW; ++b)
#pragma unroll
                        for (int r = 0; r < 16; ++r) cin[b][r] = cinit_s[(wave_n * TNW + b) * 32 + 16 * lhalf + r];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < TNW; ++b) acc[a][b] = d3_mfma(wf[b], xf[a], cin[b]);
                } else {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < TNW; ++b) acc[a][b] = d3_mfma(wf[b], xf[a], acc[a][b]);
                }
            }
            }
        };
        if constexpr (CIN) {
            tap_step(0, 0, std::true_type{});
            tap_step(0, 1, std::false_type{});
            tap_step(0, 2, std::false_type{});
#pragma unroll 1
            for (int i = 1; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) tap_step(i, j, std::false_type{});
        } else
        if constexpr (F16) {
#pragma unroll
            for (int p_ = 0; p_ < 5; ++p_) {
                uint4 xf[2], wf[TNW];
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    xf[a] = *reinterpret_cast<const uint4*>(patch + (wave * 64 + a * 32 + lrow) * 16 + tpo[p_]);
#pragma unroll
                for (int b = 0; b < TNW; ++b)
                    wf[b] = *reinterpret_cast<const uint4*>(wl + ((wave_n * TNW + b) * 32 + lrow) * WROW + two[p_]);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < TNW; ++b) acc[a][b] = d3_mfma_f16(xf[a], wf[b], acc[a][b]);
            }
        } else
        // one kernel row at a time: a fully unrolled tap loop lets the scheduler hoist all 36 fragment reads (198 VGPRs,
        // 2 waves per SIMD); rolled over i it keeps 12 in flight
#pragma unroll 1
        for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int tap = i * 3 + j;
#pragma unroll
            for (int kk = 0; kk < CPP / 2; ++kk) {
                const int c = kk * 2 + lhalf;
                uint4 xf[2], wf[TNW];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int px = wave * 64 + a * 32 + lrow + j;
                    xf[a] = *reinterpret_cast<const uint4*>(patch + ((i * D3_RUN + px) * CPP + (c ^ swz(px))) * 16);
                }
#pragma unroll
                for (int b = 0; b < TNW; ++b)
                    wf[b] = *reinterpret_cast<const uint4*>(wl + ((wave_n * TNW + b) * 32 + lrow) * WROW + (tap * CPP + c) * 16);
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < TNW; ++b) {
                        if constexpr (I8) acc[a][b] = d3_mfma_i8(xf[a], wf[b], acc[a][b]);
                        else if constexpr (BF16) acc[a][b] = d3_mfma_bf16(xf[a], wf[b], acc[a][b]);
                        else acc[a][b] = d3_mfma(wf[b], xf[a], acc[a][b]);     // FP4: weights = rows (channels), positions = columns
                    }
            }
        }
        if constexpr (I8) {
            // ---- DoReFa code epilogue (arithmetic of mfma_gemm.hip's mode-2 epilogue): y = (float)acc * scale + bias is the
            // fp32 value the conv would store; t = fl(fl(y*alpha) + beta) [+ fl(rscale * rcode)]; ReLU; q = rint(levels*t).
            // The 32x32 tile goes through a wave-private LDS patch so that a lane holds 4 consecutive channels of a row.
            float* T = reinterpret_cast<float*>(patch + 3 * D3_RUN * CPP * 16) + (tid >> 6) * 1024;
            const float sc = g.scale_dev ? g.scale * *g.scale_dev : g.scale;
            int bad = 0;
            unsigned vmask = 0;                     // validity of this lane's 4 rows per a: bit a*4 + i
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const long long q = q0 + wave * 64 + a * 32 + i * 8 + (lane >> 3);
                    if (q < g.total) {
                        const unsigned uq = (unsigned)q;
                        const unsigned img = (unsigned)__umul64hi((unsigned long long)uq, g.magic_plane);
                        const unsigned rem = uq - img * plane;
                        const int y = (int)__umul64hi((unsigned long long)rem, g.magic_wp), x = (int)rem - y * g.Wp;
                        if (y >= 1 && y <= g.H && x >= 1 && x <= g.W) vmask |= 1u << (a * 4 + i);
                    }
                }
            unsigned char* Q = reinterpret_cast<unsigned char*>(g.out);
#pragma unroll
            for (int b = 0; b < TNW; ++b) {
                const int nb = (wave_n * TNW + b) * 32;
                const float bvv = (g.bias && nb + lrow < g.Cout) ? g.bias[nb + lrow] : 0.0f;
                const int n = nb + (lane & 7) * 4;
                float al4[4], be4[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = n + e < g.Cout;
                    al4[e] = in ? g.alpha[n + e] : 0.0f;
                    be4[e] = in ? g.beta[n + e] : 0.0f;
                }
#pragma unroll
                for (int a = 0; a < 2; ++a) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        T[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * 32 + lrow] = (float)acc[a][b][r] * sc + bvv;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = i * 8 + (lane >> 3);
                        const float4 v4 = *reinterpret_cast<const float4*>(T + row * 32 + (lane & 7) * 4);
                        const long long q = q0 + wave * 64 + a * 32 + row;
                        if (q < g.total && n < g.ldo) {
                            uint32_t word = 0;                       // border position: the output plane's halo
                            if ((vmask >> (a * 4 + i)) & 1u) {
                                const float v[4] = {v4.x, v4.y, v4.z, v4.w};
                                uint32_t rword = 0;
                                if (g.res_codes && n < g.Cout)
                                    rword = *reinterpret_cast<const uint32_t*>(g.res_codes + q * g.ldrc + n);
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    int qc = 0;
                                    if (n + e < g.Cout) {
                                        const float x0 = (g.relu == 2 && v[e] < 0.0f) ? 0.0f : v[e];
                                        float t = x0 * al4[e] + be4[e];                 // two roundings (-ffp-contract=off)
                                        if (g.res_codes) t = t + g.rscale * (float)(int8_t)(rword >> (8 * e));
                                        if (g.relu == 1) t = t < 0.0f ? 0.0f : t;
                                        const float qf = rintf(g.levels * t);
                                        if (!(qf >= -127.0f && qf <= 127.0f)) bad = 1; else qc = (int)qf;
                                    }
                                    word |= (uint32_t)(uint8_t)(int8_t)qc << (8 * e);
                                }
                            }
                            *reinterpret_cast<uint32_t*>(Q + q * g.ldo + n) = word;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            if (__any(bad) && lane == 0) atomicOr(g.overflow, 1);
        } else {
            // ---- threshold epilogue (same arithmetic as mfma_gemm.hip: bit = fl((acc + bias) * alpha) < -beta) ----
            if constexpr (FP4 && LEAN) {           // (its own instantiation: with both epilogues in one kernel the 3-per-CU tile spilled)
                // The lean form (round 6, profiles/r6_c5_direct_conv.md): the general form below spends ~10 quarter-rate
                // instructions (64-bit multiply-adds of the magic divisions and of the row addresses) per 32 positions and output
                // word — with the matrix work this short that was as much SIMD time as the MFMAs.  Here the tile's first position
                // is decomposed ONCE on the scalar unit; a lane's position is that plus an offset below 256, so its row / image
                // carries are quotients of numbers below 512 (one 24-bit multiply and a shift) or a single compare; stores go
                // through a buffer resource with 32-bit offsets.
                const unsigned uq0 = (unsigned)q0;
                const unsigned img0 = (unsigned)__umul64hi((unsigned long long)uq0, g.magic_plane);
                const unsigned rem0 = uq0 - img0 * plane;
                const int y0 = (int)__umul64hi((unsigned long long)rem0, g.magic_wp), x0 = (int)rem0 - y0 * g.Wp;
                const int mrow0 = ((int)img0 * g.H + (y0 - 1)) * g.W + (x0 - 1);      // wraps for halo positions: only valid ones use it
                const long long left_ll = g.total - q0;
                const int lim = left_ll < D3_TM ? (int)left_ll : D3_TM;
                const bool wide = g.Wp >= 256, tall = g.Hp >= 256;
    #pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int off = wave * 64 + a * 32 + lrow;
                    const int t = x0 + off;
                    // (24-bit multiplies: full rate; t, yt < 512 there, and k, ky <= 86)
                    const int k = wide ? (t >= g.Wp ? 1 : 0) : (int)(__umul24((unsigned)t, g.m20_wp) >> 20);
                    const int x = t - __mul24(k, g.Wp), yt = y0 + k;
                    const int ky = tall ? (yt >= g.Hp ? 1 : 0) : (int)(__umul24((unsigned)yt, g.m20_hp) >> 20);
                    const int y = yt - __mul24(ky, g.Hp);
                    const bool in = off < lim;
                    const bool valid = in && (unsigned)(y - 1) < (unsigned)g.H && (unsigned)(x - 1) < (unsigned)g.W;
                    if (g.out_bits) {
                        // bit plane (one column wave, or two with 64 channels each; the host sends nothing else here): the two lane halves hold bits
                        // 0-15 / 16-31 of every block's word.  One v_permlane32_swap per pair of blocks completes block 2p's word in
                        // the low half and block 2p + 1's in the high half; a second one brings a position's words together:
                        // 4 blocks -> 8 bytes per lane (words 0-1 low half, 2-3 high half), 2 blocks -> the low half stores the row
                        if constexpr ((WN == 1 && (TNW == 2 || TNW == 4)) || (WN == 2 && TNW == 2)) {
                            // row = mrow0 + off - 2 k - 2 W ky (a row carry skips 2 halo pixels, an image carry 2 halo rows): the
                            // tile's part is a scalar product, the lane's part small enough for a 24-bit multiply
                            const unsigned vo = (unsigned)mrow0 * (unsigned)(g.ldo * 4) +
                                                (unsigned)__mul24(off - 2 * k - __mul24(2 * g.W, ky), g.ldo * 4);
                            uint32_t word[TNW / 2];
    #pragma unroll
                            for (int b = 0; b < TNW; b += 2) {
                                uint32_t w0 = 0, w1 = 0;
    #pragma unroll
                                for (int r = 15; r >= 0; --r) w0 = __builtin_amdgcn_alignbit(w0, __float_as_uint(acc[a][b][r]), 31);
    #pragma unroll
                                for (int r = 15; r >= 0; --r) w1 = __builtin_amdgcn_alignbit(w1, __float_as_uint(acc[a][b + 1][r]), 31);
                                const auto sw = __builtin_amdgcn_permlane32_swap(w0, w1, false, false);
                                word[b / 2] = (sw[1] << 16) | sw[0];
                            }
                            typedef unsigned d3_u2 __attribute__((ext_vector_type(2)));
                            if constexpr (TNW == 4) {
                                const auto s2 = __builtin_amdgcn_permlane32_swap(word[0], word[1], false, false);
                                const d3_u2 v = {s2[0], s2[1]};
                                if (valid) __builtin_amdgcn_raw_buffer_store_b64(v, ors, (int)(vo + (unsigned)(lhalf * 8)), 0, 0);
                            } else {
                                const auto s2 = __builtin_amdgcn_permlane32_swap(word[0], word[0], false, false);
                                if constexpr (WN == 2) {            // 128 channels over two column waves: each stores its two words
                                    const d3_u2 v = {s2[0], s2[1]};
                                    if (valid && lhalf == 0) __builtin_amdgcn_raw_buffer_store_b64(v, ors, (int)(vo + (unsigned)(wave_n * 8)), 0, 0);
                                } else
                                if (valid && lhalf == 0) {
                                    if (g.ldo == 4) {
                                        const d3_u4 v = {s2[0], s2[1], 0u, 0u};
                                        __builtin_amdgcn_raw_buffer_store_b128(v, ors, (int)vo, 0, 0);
                                    } else {
                                        const d3_u2 v = {s2[0], s2[1]};
                                        __builtin_amdgcn_raw_buffer_store_b64(v, ors, (int)vo, 0, 0);
                                    }
                                }
                            }
                        }
                    } else {
                        // nibble halo plane: a lane owns 16 channels = 8 bytes of its position; nibble = sign << 3 | 2, zeros on the border.
                        // v_alignbit(w, acc, 28) shifts a nibble in whose top bit is the sign; the other three bits are masked at the end
                        const uint32_t msk = valid ? 0x88888888u : 0u, two = valid ? 0x22222222u : 0u;
                        const unsigned vo = uq0 * (unsigned)(g.ldo * 4) + __umul24((unsigned)off, (unsigned)(g.ldo * 4)) +
                                            (unsigned)(wave_n * TNW * 16 + lhalf * 8);
    #pragma unroll
                        for (int b = 0; b < TNW; ++b) {
                            uint32_t lo = 0, hi = 0;
    #pragma unroll
                            for (int r = 7; r >= 0; --r) lo = __builtin_amdgcn_alignbit(lo, __float_as_uint(acc[a][b][r]), 28);
    #pragma unroll
                            for (int r = 15; r >= 8; --r) hi = __builtin_amdgcn_alignbit(hi, __float_as_uint(acc[a][b][r]), 28);
                            typedef unsigned d3_u2 __attribute__((ext_vector_type(2)));
                            const d3_u2 v = {(lo & msk) | two, (hi & msk) | two};
                            if (in) __builtin_amdgcn_raw_buffer_store_b64(v, ors, (int)(vo + (unsigned)(b * 16)), 0, 0);
                        }
                    }
                }
                continue;
            }
    #pragma unroll
            for (int a = 0; a < 2; ++a) {
                const long long q = q0 + wave * 64 + a * 32 + lrow;           // lanes 0..31 own the 32 positions
                bool valid = false;
                long long mrow = 0;
                if (q < g.total) {
                    const unsigned uq = (unsigned)q;                       // total <= 2^31 (host check)
                    const unsigned img = (unsigned)__umul64hi((unsigned long long)uq, g.magic_plane);
                    const unsigned rem = uq - img * plane;
                    const int y = (int)__umul64hi((unsigned long long)rem, g.magic_wp), x = (int)rem - y * g.Wp;
                    valid = y >= 1 && y <= g.H && x >= 1 && x <= g.W;
                    mrow = ((long long)img * g.H + (y - 1)) * g.W + (x - 1);
                }
    #pragma unroll
                for (int b = 0; b < TNW; ++b) {
                    uint32_t myword = 0;
                    if constexpr (FP4) {
                        // this lane: position lrow of block a, channels 16 lhalf .. + 15 of column block b: sign bits, channel r -> bit r
                        uint32_t w16 = 0;
    #pragma unroll
                        for (int r = 15; r >= 0; --r) w16 = __builtin_amdgcn_alignbit(w16, __float_as_uint(acc[a][b][r]), 31);
                        const uint32_t other = (uint32_t)__shfl_xor((int)w16, 32);
                        myword = lhalf ? ((w16 << 16) | (other & 0xFFFFu)) : ((other << 16) | (w16 & 0xFFFFu));
                    } else {
    #pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned long long mask = __ballot(__uint_as_float(__float_as_uint(acc[a][b][r]) ^ e_sg[b]) < e_th[b]);
                        const int R = (r & 3) + 8 * (r >> 2);
                        // gfx950 does not interlock a VALU-written SGPR read by the next VALU: see mfma_gemm.hip
                        asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(myword) : "s"((uint32_t)mask), "n"(R));
                        asm("v_writelane_b32 %0, %1, %2" : "+v"(myword) : "s"((uint32_t)(mask >> 32)), "n"(R + 4));
                    }
                    }
                    const int bg = wave_n * TNW + b;                            // column block of the whole tile
                    if (lane < 32 && q < g.total) {
                        if (g.out_bits) {
                            if (valid && bg < g.ldo) {
                                g.out[mrow * g.ldo + bg] = myword;
                                if (bg == TNW * WN - 1)
                                    for (int wc = TNW * WN; wc < g.ldo; ++wc) g.out[mrow * g.ldo + wc] = 0u;
                            }
                        } else if (bg * 4 < g.ldo) {
                            const int left = g.Cout - bg * 32;
                            uint4 o = make_uint4(0, 0, 0, 0);                 // border position: the output plane's halo
                            if (valid) {
                                const uint32_t mw = left >= 32 ? 0xFFFFFFFFu : (left > 0 ? ((1u << left) - 1u) : 0u);
                                const uint32_t sw = myword & mw;
                                o.x = (d3_spread8(mw) << 1) | (d3_spread8(sw) << 3);
                                o.y = (d3_spread8(mw >> 8) << 1) | (d3_spread8(sw >> 8) << 3);
                                o.z = (d3_spread8(mw >> 16) << 1) | (d3_spread8(sw >> 16) << 3);
                                o.w = (d3_spread8(mw >> 24) << 1) | (d3_spread8(sw >> 24) << 3);
                            }
                            *reinterpret_cast<uint4*>(g.out + q * g.ldo + bg * 4) = o;
                        }
                    }
                }
            }
        }
    }
}

template <int CPP, int TNW, int WN, int OCC, int EL = 0, bool LEAN = false>
int d3_launch(const D3Args& g, int wg_per_cu, hipStream_t stream) {
    const int lds = TNW * WN * 32 * d3_wrow(CPP) + 3 * D3_RUN * CPP * 16 + (EL == 1 ? 4 * WN * 4096 : 0) +
                    (EL == 0 ? TNW * WN * 32 * 8 : 0);                    // EL 0: + the start values and flip flags per channel
    static QtLdsOnce once;
    if (qt_ensure_dyn_lds(once, reinterpret_cast<const void*>(direct3x3_kernel<CPP, TNW, WN, OCC, EL, LEAN>), lds) != QT_OK) return QT_ERR_LAUNCH;
    const long long ntiles = (g.total + D3_TM - 1) / D3_TM;
    const long long cap = 256ll * wg_per_cu;
    const unsigned grid = (unsigned)(ntiles < cap ? ntiles : cap);
    hipLaunchKernelGGL((direct3x3_kernel<CPP, TNW, WN, OCC, EL, LEAN>), dim3(grid), dim3(256 * WN), lds, stream, g);
    return qt_check_launch();
}

}  // namespace

extern "C" int qt_conv3x3_direct_nib(int elem, const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw,
                                     const uint32_t* Wmat, int64_t ldw, const float* bias, const float* alpha,
                                     const float* beta, uint32_t* out, int64_t ldo, int64_t Cout, int out_bits,
                                     qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || Cout <= 0 || ldo <= 0) return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!P || !Wmat || !alpha || !beta || !out) return QT_ERR_INVALID_ARG;
    if (elem != 0 && elem != 2) return QT_ERR_INVALID_ARG;
    if ((Cw != 8 && Cw != 16) || Cout > 128 || (elem == 2 && Cw != 8)) return QT_ERR_UNSUPPORTED;   // 32 / 64-byte pixels, one column tile
    if (ldw < 9 * Cw || (ldw & 3) || !qt_aligned16(P) || !qt_aligned16(Wmat) || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    if (out_bits ? (ldo < (Cout + 31) / 32) : (ldo != (Cout + 31) / 32 * 4)) return QT_ERR_INVALID_ARG;
    if (H + 2 > 32767 || W + 2 > 32767 || N * (H + 2) * (W + 2) > (1ll << 31)) return QT_ERR_UNSUPPORTED;
    D3Args g;
    g.P = reinterpret_cast<const unsigned char*>(P);
    g.Wm = reinterpret_cast<const unsigned char*>(Wmat);
    g.bias = bias; g.alpha = alpha; g.beta = beta; g.out = out;
    g.H = (int)H; g.W = (int)W; g.Hp = (int)H + 2; g.Wp = (int)W + 2;
    g.total = N * (int64_t)g.Hp * g.Wp;
    g.buf_ok = g.total * (int64_t)(Cw * 4) < (1ll << 31) ? 1 : 0;
    g.Cout = (int)Cout; g.ldw = (int)(ldw * 4); g.ldo = (int)ldo; g.out_bits = out_bits ? 1 : 0;
    g.magic_plane = ~0ull / (unsigned long long)(g.Hp * g.Wp) + 1;     // divisors >= 9: ceil(2^64 / d)
    g.magic_wp = ~0ull / (unsigned long long)g.Wp + 1;
    g.scale = 1.0f; g.scale_dev = nullptr; g.rscale = 0.0f; g.levels = 0.0f; g.res_codes = nullptr; g.ldrc = 0; g.relu = 0;
    g.overflow = nullptr;
    // the lean fp4 epilogue: 64 / 128 output channels (whole tiles), 32-bit byte offsets into the output plane, bit rows of 2 / 4 words;
    // QT_D3_EPI=general keeps the general form (A/B runs and the bit-identity test)
    {
        static const bool general = [] { const char* e = getenv("QT_D3_EPI"); return e && e[0] == 'g'; }();
        const long long out_bytes = out_bits ? N * H * W * ldo * 4 : g.total * ldo * 4;
        const bool whole = Cout == 64 || Cout == 128;      // every column block of the tile shapes below is a real one
        const bool bits_ok = Cout == 128 ? ldo == 4 : (Cw == 8 && (ldo == 4 || ldo == 2));
        g.epi32 = (elem == 0 && !general && whole && out_bytes < (1ll << 32) && (!out_bits || bits_ok)) ? 1 : 0;
        g.out_bytes = g.epi32 ? (unsigned)out_bytes : 0u;
        g.m20_wp = (unsigned)(((1u << 20) + g.Wp - 1) / g.Wp);
        g.m20_hp = (unsigned)(((1u << 20) + g.Hp - 1) / g.Hp);
    }
    hipStream_t s = (hipStream_t)stream;
    // LDS per workgroup: 44 / 64 KB (64 input channels), 87 / 125 KB (128)
    if (elem == 2) return Cout <= 64 ? d3_launch<2, 2, 1, 3, 2>(g, 3, s) : d3_launch<2, 4, 1, 2, 2>(g, 2, s);
    if (g.epi32) {
        if (Cw == 8) return Cout <= 64 ? d3_launch<2, 2, 1, 3, 0, true>(g, 3, s) : d3_launch<2, 4, 1, 2, 0, true>(g, 2, s);
        return Cout <= 64 ? d3_launch<4, 1, 2, 2, 0, true>(g, 1, s) : d3_launch<4, 2, 2, 2, 0, true>(g, 1, s);
    }
    if (Cw == 8) return Cout <= 64 ? d3_launch<2, 2, 1, 3>(g, 3, s) : d3_launch<2, 4, 1, 2>(g, 2, s);
    return Cout <= 64 ? d3_launch<4, 1, 2, 2>(g, 1, s) : d3_launch<4, 2, 2, 2>(g, 1, s);
}

// The real-valued first layer on fp16 PAIR planes (two terms: P = the halo-1 plane of qt_f16x2_s2d_pack*_f32(s = 1, padding 1), 16
// bytes per pixel = up to 4 channels x [hi, lo]; scale_dev = the plane's power-of-two scale; Wmat = qt_f16x2_pack_conv_weight_f32's
// tap-major rows, 16 bytes per tap): the two lane halves of one MFMA take two taps, 5 MFMAs per position block instead of 9, half
// the plane bytes of the triple form (elem = 2 above) at the two-term bound max(2^-22 |x|, 2^-39 max|x|).
extern "C" int qt_conv3x3_direct_pairs(const uint32_t* P, int64_t N, int64_t H, int64_t W, const uint32_t* Wmat, int64_t ldw,
                                       const float* bias, const float* scale_dev, const float* alpha, const float* beta,
                                       uint32_t* out, int64_t ldo, int64_t Cout, int out_bits, qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || Cout <= 0 || ldo <= 0) return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!P || !Wmat || !alpha || !beta || !out || !scale_dev) return QT_ERR_INVALID_ARG;
    if (Cout > 128) return QT_ERR_UNSUPPORTED;
    if (ldw < 9 * 4 || (ldw & 3) || !qt_aligned16(P) || !qt_aligned16(Wmat) || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    if (out_bits ? (ldo < (Cout + 31) / 32) : (ldo != (Cout + 31) / 32 * 4)) return QT_ERR_INVALID_ARG;
    if (H + 2 > 32767 || W + 2 > 32767 || N * (H + 2) * (W + 2) > (1ll << 31)) return QT_ERR_UNSUPPORTED;
    D3Args g;
    g.P = reinterpret_cast<const unsigned char*>(P);
    g.Wm = reinterpret_cast<const unsigned char*>(Wmat);
    g.bias = bias; g.alpha = alpha; g.beta = beta; g.out = out;
    g.H = (int)H; g.W = (int)W; g.Hp = (int)H + 2; g.Wp = (int)W + 2;
    g.total = N * (int64_t)g.Hp * g.Wp;
    g.buf_ok = g.total * (int64_t)16 < (1ll << 31) ? 1 : 0;
    g.Cout = (int)Cout; g.ldw = (int)(ldw * 4); g.ldo = (int)ldo; g.out_bits = out_bits ? 1 : 0;
    g.magic_plane = ~0ull / (unsigned long long)(g.Hp * g.Wp) + 1;
    g.magic_wp = ~0ull / (unsigned long long)g.Wp + 1;
    g.scale = 1.0f; g.scale_dev = scale_dev; g.rscale = 0.0f; g.levels = 0.0f; g.res_codes = nullptr; g.ldrc = 0; g.relu = 0;
    g.overflow = nullptr;
    g.epi32 = 0; g.out_bytes = 0; g.m20_wp = g.m20_hp = 0;
    hipStream_t s = (hipStream_t)stream;
    return Cout <= 64 ? d3_launch<1, 2, 1, 3, 3>(g, 3, s) : d3_launch<1, 4, 1, 2, 3>(g, 2, s);
}

// The same direct kernel for DoReFa int8 code planes with the code epilogue of qt_conv2d_implicit_codes: P and codes are
// halo-1 planes of identical geometry [N][H+2][W+2][.], res_codes (optional) likewise.  64 input channels (Cw = 16 words),
// Cout <= 64.
extern "C" int qt_conv3x3_direct_codes(const uint32_t* P, int64_t N, int64_t H, int64_t W, int64_t Cw, const uint32_t* Wmat,
                                       int64_t ldw, const float* bias, float scale, const float* scale_dev,
                                       const float* alpha, const float* beta, const int8_t* res_codes, int64_t ldrc_bytes,
                                       float res_scale, int relu, int bit_width, int8_t* codes, int64_t ldc_bytes,
                                       int64_t Cout, int32_t* overflow, qt_stream_t stream) {
    if (N < 0 || H <= 0 || W <= 0 || Cout <= 0 || bit_width < 2 || bit_width > 8 || relu < 0 || relu > 2)
        return QT_ERR_INVALID_ARG;
    if (N == 0) return QT_OK;
    if (!P || !Wmat || !alpha || !beta || !codes || !overflow) return QT_ERR_INVALID_ARG;
    if (Cw != 16 || Cout > 64) return QT_ERR_UNSUPPORTED;
    if (ldw < 9 * Cw || (ldw & 3) || !qt_aligned16(P) || !qt_aligned16(Wmat) || !qt_aligned16(codes) || (ldc_bytes & 15))
        return QT_ERR_ALIGNMENT;
    if (ldc_bytes != ((Cout + 15) & ~15ll)) return QT_ERR_INVALID_ARG;
    if (res_codes && (ldrc_bytes < ((Cout + 3) & ~3ll) || (ldrc_bytes & 3) || (reinterpret_cast<uintptr_t>(res_codes) & 3)))
        return QT_ERR_ALIGNMENT;
    if (H + 2 > 32767 || W + 2 > 32767 || N * (H + 2) * (W + 2) > (1ll << 31)) return QT_ERR_UNSUPPORTED;
    D3Args g;
    g.P = reinterpret_cast<const unsigned char*>(P);
    g.Wm = reinterpret_cast<const unsigned char*>(Wmat);
    g.bias = bias; g.alpha = alpha; g.beta = beta; g.out = reinterpret_cast<uint32_t*>(codes);
    g.H = (int)H; g.W = (int)W; g.Hp = (int)H + 2; g.Wp = (int)W + 2;
    g.total = N * (int64_t)g.Hp * g.Wp;
    g.buf_ok = g.total * (int64_t)(Cw * 4) < (1ll << 31) ? 1 : 0;
    g.Cout = (int)Cout; g.ldw = (int)(ldw * 4); g.ldo = (int)ldc_bytes; g.out_bits = 0;
    g.magic_plane = ~0ull / (unsigned long long)(g.Hp * g.Wp) + 1;
    g.magic_wp = ~0ull / (unsigned long long)g.Wp + 1;
    g.scale = scale; g.scale_dev = scale_dev; g.rscale = res_scale; g.levels = (float)((1 << bit_width) - 1);
    g.res_codes = reinterpret_cast<const unsigned char*>(res_codes); g.ldrc = (int)ldrc_bytes; g.relu = relu;
    g.overflow = overflow;
    g.epi32 = 0; g.out_bytes = 0; g.m20_wp = g.m20_hp = 0;
    return d3_launch<4, 1, 2, 2, 1>(g, 1, (hipStream_t)stream);
}
