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
    int H, W, Hp, Wp, Cout, ldw, ldo, out_bits;
    unsigned long long magic_plane, magic_wp;   // ceil(2^64 / (Hp*Wp)), ceil(2^64 / Wp): exact 32-bit quotients
};

constexpr int D3_TM = 256, D3_RUN = D3_TM + 2;

// CPP: 16-byte chunks per input pixel (Cin = 32 * CPP); a workgroup = 4 (position) x WN (column) waves, each wave owns
// 64 positions x TNW 32-column blocks (Cout <= 32 * TNW * WN); OCC: waves per SIMD the register budget is sized for
template <int CPP, int TNW, int WN, int OCC, bool BF16>
__global__ __launch_bounds__(256 * WN, OCC) void direct3x3_kernel(D3Args g) {
    constexpr int NT = 256 * WN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WROW = 9 * CPP * 16 + 16;        // + 16: consecutive rows land on different bank groups
    constexpr int WBYTES = TNW * WN * 32 * WROW;
    unsigned char* wl = smem;
    unsigned char* patch = smem + WBYTES;          // [3][RUN][CPP chunks], chunk c of pixel px at c ^ swz(px)
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, wave_n = tid >> 8;
    const int lrow = lane & 31, lhalf = lane >> 5;
    // conflict-free ds_read_b128 of 8 consecutive pixels: spread their chunk over the 128-byte bank line
    auto swz = [](int px) { return (px * CPP / 8) & (CPP - 1); };

    // weights: resident for the whole launch
    for (int e = tid; e < TNW * WN * 32 * 9 * CPP; e += NT) {
        const int row = e / (9 * CPP), c = e - row * (9 * CPP);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (row < g.Cout) v = *reinterpret_cast<const uint4*>(g.Wm + (long long)row * g.ldw + c * 16);
        *reinterpret_cast<uint4*>(wl + row * WROW + c * 16) = v;
    }
    float al[TNW], nbe[TNW], bv[TNW];
#pragma unroll
    for (int b = 0; b < TNW; ++b) {
        const int n = (wave_n * TNW + b) * 32 + lrow;
        const bool in = n < g.Cout;
        al[b] = in ? g.alpha[n] : 0.0f;
        nbe[b] = in ? -g.beta[n] : 0.0f;
        bv[b] = (in && g.bias) ? g.bias[n] : 0.0f;
    }
    const long long ntiles = (g.total + D3_TM - 1) / D3_TM;
    const unsigned plane = (unsigned)(g.Hp * g.Wp);
    // the patch of the NEXT tile travels through registers while this tile computes (global latency off the critical
    // path): NLD 16-byte loads per thread, issued right after the barrier that publishes the current patch
    constexpr int NLD = (3 * D3_RUN * CPP + NT - 1) / NT;
    uint4 pre[NLD];
    auto fetch = [&](long long t) {
        const long long q0n = t * D3_TM;
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
        d3_v16f acc[2][TNW];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < TNW; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;
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
                    for (int b = 0; b < TNW; ++b)
                        acc[a][b] = BF16 ? d3_mfma_bf16(xf[a], wf[b], acc[a][b]) : d3_mfma(xf[a], wf[b], acc[a][b]);
            }
        }
        // ---- threshold epilogue (same arithmetic as mfma_gemm.hip: bit = fl((acc + bias) * alpha) < -beta) ----
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
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t = acc[a][b][r] + bv[b];
                    const unsigned long long mask = __ballot(t * al[b] < nbe[b]);
                    const int R = (r & 3) + 8 * (r >> 2);
                    // gfx950 does not interlock a VALU-written SGPR read by the next VALU: see mfma_gemm.hip
                    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(myword) : "s"((uint32_t)mask), "n"(R));
                    asm("v_writelane_b32 %0, %1, %2" : "+v"(myword) : "s"((uint32_t)(mask >> 32)), "n"(R + 4));
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

template <int CPP, int TNW, int WN, int OCC, bool BF16 = false>
int d3_launch(const D3Args& g, int wg_per_cu, hipStream_t stream) {
    const int lds = TNW * WN * 32 * (9 * CPP * 16 + 16) + 3 * D3_RUN * CPP * 16;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(direct3x3_kernel<CPP, TNW, WN, OCC, BF16>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return QT_ERR_LAUNCH;
    const long long ntiles = (g.total + D3_TM - 1) / D3_TM;
    const long long cap = 256ll * wg_per_cu;
    const unsigned grid = (unsigned)(ntiles < cap ? ntiles : cap);
    hipLaunchKernelGGL((direct3x3_kernel<CPP, TNW, WN, OCC, BF16>), dim3(grid), dim3(256 * WN), lds, stream, g);
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
    g.Cout = (int)Cout; g.ldw = (int)(ldw * 4); g.ldo = (int)ldo; g.out_bits = out_bits ? 1 : 0;
    g.magic_plane = ~0ull / (unsigned long long)(g.Hp * g.Wp) + 1;     // divisors >= 9: ceil(2^64 / d)
    g.magic_wp = ~0ull / (unsigned long long)g.Wp + 1;
    hipStream_t s = (hipStream_t)stream;
    // LDS per workgroup: 44 / 64 KB (64 input channels), 87 / 125 KB (128)
    if (elem == 2) return Cout <= 64 ? d3_launch<2, 2, 1, 3, true>(g, 3, s) : d3_launch<2, 4, 1, 2, true>(g, 2, s);
    if (Cw == 8) return Cout <= 64 ? d3_launch<2, 2, 1, 3>(g, 3, s) : d3_launch<2, 4, 1, 2>(g, 2, s);
    return Cout <= 64 ? d3_launch<4, 1, 2, 2>(g, 1, s) : d3_launch<4, 2, 2, 2>(g, 1, s);
}
