// Direct 3 x 3 / stride 1 / padding 1 conv of the DoReFa code planes for SMALL channel counts (models/Resnet/Resnet_bin.py:63-97
// with DorefaConv2d(bit_width = 1) + nnDorefaQuant(k): stage 1 = 64 -> 64 at 32 x 32, stage 2 = 128 -> 128 at 16 x 16): the
// launches profiles/r5_c4_pmc.md shows furthest below their roofline.  The implicit-GEMM kernel (mfma_gemm_kernel.h) treats such
// a layer as ~1000 independent 256 x 64 tiles, each of which re-derives its addressing, re-loads the whole weight tile and
// fetches every input pixel nine times (once per filter tap) through the L2 — a tile is all prologue and epilogue around 34
// MFMAs.  Here instead
//   * a workgroup is PERSISTENT: it loads its 64 output channels of the weight (all nine taps: 36 / 72 KiB) into LDS ONCE and
//     walks over the row tiles of the layer;
//   * the input of a tile (128 consecutive output pixels = whole output rows) is ONE contiguous range of the halo plane —
//     (rows + 2) x (W + 2 hx) pixels — fetched once by LDS-DMA; the nine taps are nine LDS offsets into that patch (no im2col,
//     no per-tap global traffic);
//   * the patch of tile i + 1 streams into the other LDS buffer while tile i's MFMAs and its code epilogue run.
// int8 operands on v_mfma_i32_32x32x32_i8, exact int32 sums, then the code epilogue of the implicit-GEMM kernel's straight-line
// form (device BatchNorm arithmetic [+ code residual] [-> ReLU] -> rint(levels * t) -> int8 codes into the next layer's halo
// plane): the same roundings in the same order, so the codes are bit-identical to the implicit-GEMM route (integer sums are
// order-independent).  Whatever does not meet the conditions of qt_code_conv3x3_try stays on the implicit-GEMM kernel.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include "qt_common.h"
#include "pp_common.h"

namespace {

typedef int c3_v4i __attribute__((ext_vector_type(4)));
typedef int c3_v16i __attribute__((ext_vector_type(16)));
typedef float c3_v2f __attribute__((ext_vector_type(2)));

struct C3Args {
    const unsigned char* P;     // input code plane [Nimg][Hp][Wp][CB bytes], zero halo of (ihy, ihx) pixels
    const unsigned char* Wm;    // weights [Cout][ldw bytes]: tap t of output channel n at n * ldw + t * CB
    int ldw;
    int Nimg, H, W, Hp, Wp, ihy, ihx;
    int sh_w, sh_h;             // log2(W), log2(H)
    int Cout;
    int rows_per_tile;          // flattened output rows (of W pixels) per 128-pixel tile
    int tiles_m, tiles_n;
    // epilogue (EpiArgs of mfma_gemm_kernel.h, mode 2, device BatchNorm form)
    const float* weight;        // BatchNorm weight
    const float* bias;          // BatchNorm bias
    const float* bn_stats;      // [mean | rs]
    float scale;
    const float* scale_dev;
    float levels, rscale;
    int relu;
    const int8_t* res_codes;
    int ldrc, rhy, rhx;
    int8_t* Q;
    int ldq, ohy, ohx;
    int32_t* overflow;
    int no_xcd_order;           // tools: plain workgroup order (A/B)
};

// LDS chunk swizzle: 16-byte chunk c of pixel / weight row q lands on chunk c ^ x(q) of its CB-byte record, so that the 16 lanes
// a ds_read_b128 serves per pass (consecutive q, the same logical chunk) spread over all 64 banks
template <int CPP>
__device__ __forceinline__ int c3_x(int q) {
    constexpr int SH = CPP == 4 ? 2 : (CPP == 8 ? 1 : 0);
    return (q >> SH) & (CPP - 1);
}

// Zero border of the output halo plane (same order as mfma_gemm_kernel.h's zero_halo_border; every workgroup takes a share)
__device__ __forceinline__ void c3_zero_border(void* plane, int cpp, int64_t nimg, int H, int W, int hy, int hx, int zb, int nzb) {
    const int Hp = H + 2 * hy, Wp = W + 2 * hx;
    const int top = hy * Wp, side = 2 * hx * H, per_img = 2 * top + side;
    const int64_t total = nimg * per_img * cpp;
    uint4* Qz = reinterpret_cast<uint4*>(plane);
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
        Qz[(n * Hp * Wp + pix) * (int64_t)cpp + c] = make_uint4(0, 0, 0, 0);
    }
}

// CB: bytes per input pixel (= padded input channels: 64 or 128).  Workgroup: 4 waves, tile 128 pixels x 64 output channels;
// wave w owns pixels [32 w, 32 w + 32) and both 32-channel blocks.
// LDS: [W: 64 rows x 9 CB] [patch 0] [patch 1] [4 x 4 KiB transpose patches]
template <int CB, int MODE>
__global__ __launch_bounds__(MODE == 2 ? 320 : 256) void code_conv3x3_kernel(const C3Args a, int patch_bytes_max) {
    constexpr int CPP = CB / 16, TM = 128, TN = 64, WROW = 9 * CB;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Wl = smem;
    // (patch buffer b lives at smem + TN * WROW + b * patch_bytes_max: always addressed as an offset from ``smem`` itself — a
    //  pointer picked from an array by a run-time index loses its LDS address space and its reads become flat loads)
    // MODE 5: ONE patch buffer, the transpose patches alias it (dead after the main loop): 49 KiB per workgroup -> 3 per CU
    float* Tall = reinterpret_cast<float*>(smem + TN * WROW + (MODE == 5 ? 0 : 2 * patch_bytes_max));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lrow = lane & 31, lhalf = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);

    const int ntiles = a.tiles_m * a.tiles_n;
    // persistent walk: workgroup b takes column tile b % tiles_n and the row tiles b / tiles_n, + gridDim / tiles_n, ...
    // XCD-aware order (speed only): workgroup b is observed to run on XCD b % 8, so XCD x takes a CONTIGUOUS range of the walk's
    // slots — neighbouring row tiles share two of their patch rows, which then meet in one L2
    int slot = blockIdx.x;
    if ((gridDim.x & 7) == 0 && !a.no_xcd_order) slot = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int tile_n = slot % a.tiles_n;
    const int n0 = tile_n * TN;
    const int mstep = gridDim.x / a.tiles_n;          // the host launches a multiple of tiles_n workgroups
    int tm = slot / a.tiles_n;

    if (tm < a.tiles_m) {
        // ---- weights of this column tile, once: slot s = 16 bytes; row n = s / (9 CPP), tap t, physical chunk c' -------------
        {
            const unsigned char* wb = a.Wm + (int64_t)n0 * a.ldw;
            constexpr int NSLOT = TN * 9 * CPP;
            for (int s0 = 0; s0 < NSLOT; s0 += (MODE == 2 ? 320 : 256)) {
                const int s = s0 + tid;
                const int n = s / (9 * CPP), rem = s - n * 9 * CPP, t = rem / CPP, cp = rem - t * CPP;
                const int c = cp ^ c3_x<CPP>(n);
                const unsigned voff = (unsigned)(n * a.ldw + t * CB + c * 16);
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(s0 + wave * 64) * 16u);
                if (s < NSLOT) glds16_asm(wb, voff, dst);
            }
        }
        const int total_rows = a.Nimg * a.Hp;          // plane rows of Wp pixels
        const int RT = a.rows_per_tile;
        // patch of row tile t: plane rows [prow0, prow0 + nrows)
        auto patch_range = [&](int t, int& prow0, int& nrows) {
            const int fr0 = t * RT;                      // first flattened output row
            if (RT <= a.H) {
                const int img = fr0 >> a.sh_h, ho0 = fr0 & (a.H - 1);
                prow0 = img * a.Hp + ho0 + a.ihy - 1;
                nrows = RT + 2;
            } else {
                const int img = fr0 >> a.sh_h;
                prow0 = img * a.Hp;
                nrows = (RT >> a.sh_h) * a.Hp;
            }
            if (prow0 + nrows > total_rows) nrows = total_rows - prow0;
        };
        // ``nthr`` threads starting at thread ``t0`` of the workgroup share the DMA of a patch (whole waves)
        auto issue_patch = [&](int t, int buf, int t0, int nthr) {
            int prow0, nrows;
            patch_range(t, prow0, nrows);
            const int nslot = nrows * a.Wp * CPP;
            const unsigned char* pb = a.P + (int64_t)prow0 * a.Wp * CB;
            const unsigned base = lds0 + (unsigned)(TN * WROW + buf * patch_bytes_max);
            const int me = tid - t0;
            for (int s0 = 0; s0 < nslot; s0 += nthr) {
                const int s = s0 + me;
                const int q = s / CPP, cp = s - q * CPP;
                const int c = cp ^ c3_x<CPP>(q);
                const unsigned voff = (unsigned)(q * CB + c * 16);
                const unsigned dst = __builtin_amdgcn_readfirstlane(base + (unsigned)(s0 + (me & ~63)) * 16u);
                if (s < nslot) glds16_asm(pb, voff, dst);
            }
        };
        issue_patch(tm, 0, 0, MODE == 2 ? 320 : 256);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        float scale = a.scale;
        if (a.scale_dev) scale *= *a.scale_dev;
        float* T = Tall + wave * 1024;
        int badf = 0;
        int cur = 0;
        // per-channel epilogue constants of the lane's 2 x 4 channels: the column tile is fixed for the workgroup's whole walk
        float al[2][4], be[2][4], mean[2][4], rs[2][4];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + b * 32 + (lane & 7) * 4 + e;
                al[b][e] = a.weight[n];
                be[b][e] = a.bias[n];
                mean[b][e] = a.bn_stats[n];
                rs[b][e] = a.bn_stats[a.Cout + n];
            }
        const int M = a.Nimg * a.H * a.W;
        // Wave 4 is the LOADER: it issues the LDS-DMA of the next tile's patch and waits for it (its vmcnt sees loads only); the
        // four compute waves never wait for a patch — with the DMA issued by the compute waves themselves their one vmcnt(0) per
        // tile also waited for the tile's own code stores (a counter shared by loads and stores cannot be waited on partially).
        // The barrier that ends a tile joins both: next patch landed, current patch no longer read.
        for (; tm < a.tiles_m; tm += mstep) {
            const int nxt = tm + mstep;
            if constexpr (MODE == 2) {
                if (wave == 4) {
                    if (nxt < a.tiles_m) issue_patch(nxt, cur ^ 1, 256, 64);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    cur ^= 1;
                    continue;
                }
            } else if constexpr (MODE != 5) {
                if (nxt < a.tiles_m) issue_patch(nxt, cur ^ 1, 0, 256);
            }
            // plane rows of this lane's 4 output pixels (pixel m = m0 + 32 wave + 8 i + (lane >> 3)) and the residual's code words,
            // requested BEFORE the main loop: their latency hides behind the MFMAs
            const int m0 = tm * TM;
            int orow[4];
            uint32_t rw[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pp = wave * 32 + i * 8 + (lane >> 3);
                const int f2 = tm * a.rows_per_tile + (pp >> a.sh_w), wo2 = pp & (a.W - 1);
                const int im2 = f2 >> a.sh_h, ho2 = f2 & (a.H - 1);
                orow[i] = (im2 * (a.H + 2 * a.ohy) + ho2 + a.ohy) * (a.W + 2 * a.ohx) + wo2 + a.ohx;
                const int rrow = (im2 * (a.H + 2 * a.rhy) + ho2 + a.rhy) * (a.W + 2 * a.rhx) + wo2 + a.rhx;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    rw[b][i] = 0;
                    if (a.res_codes)          // wave-uniform; every pixel of a tile exists (the host admits whole tiles only)
                        rw[b][i] = *reinterpret_cast<const uint32_t*>(a.res_codes + (int64_t)rrow * a.ldrc + n0 + b * 32 + (lane & 7) * 4);
                }
            }

            // ---- main loop: 9 taps x CB / 32 k-steps ------------------------------------------------------------------------
            int prow0, nrows;
            patch_range(tm, prow0, nrows);
            // this lane's A row = output pixel p = 32 wave + lrow of the tile -> patch pixel of its window's top-left corner
            const int p = wave * 32 + lrow;
            const int fr = tm * RT + (p >> a.sh_w), wo = p & (a.W - 1);
            const int img = fr >> a.sh_h, ho = fr & (a.H - 1);
            const int qtop = (img * a.Hp + ho + a.ihy - 1 - prow0) * a.Wp + wo + a.ihx - 1;
            const unsigned char* pl = smem + (TN * WROW + cur * patch_bytes_max);
            c3_v16i acc[2];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = 0;
            // LDS addresses without per-k-step arithmetic: chunk c = 2 kb + lhalf of record q sits at ((2 kb) ^ y(q)) * 16 with
            // y = lhalf ^ x(q) — for the weights (q = the lane's fixed row) every (row, kb) address is a register and the tap is
            // an immediate offset; for the patch one y per tap
            const unsigned char* wp0[CB / 32];
            const unsigned char* wp1[CB / 32];
            {
                const int y0 = (lhalf ^ c3_x<CPP>(lrow)) << 4, y1 = (lhalf ^ c3_x<CPP>(32 + lrow)) << 4;
#pragma unroll
                for (int kb = 0; kb < CB / 32; ++kb) {
                    wp0[kb] = Wl + lrow * WROW + ((kb * 32) ^ y0);
                    wp1[kb] = Wl + (32 + lrow) * WROW + ((kb * 32) ^ y1);
                }
            }
            // software pipeline, depth 2: the three fragments of k-step s + 1 are requested before the MFMAs of k-step s issue
            // (left to itself the compiler reads, waits lgkmcnt(0) and multiplies, 18 - 36 exposed LDS latencies per tile)
            constexpr int KB = CB / 32, NSTEP = (MODE == 3 ? 0 : 9 * KB);
            const unsigned char* pa[9];
            int yq[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int q = qtop + (t / 3) * a.Wp + (t % 3);
                yq[t] = (lhalf ^ c3_x<CPP>(q)) << 4;
                pa[t] = pl + q * CB;
            }
            c3_v4i fa[2], f0[2], f1[2];
            auto load_step = [&](int st, int slot) {
                const int t = st / KB, kb = st % KB;
                fa[slot] = *reinterpret_cast<const c3_v4i*>(pa[t] + ((kb * 32) ^ yq[t]));
                f0[slot] = *reinterpret_cast<const c3_v4i*>(wp0[kb] + t * CB);
                f1[slot] = *reinterpret_cast<const c3_v4i*>(wp1[kb] + t * CB);
            };
            if constexpr (NSTEP > 0) load_step(0, 0);
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                if (st + 1 < NSTEP) load_step(st + 1, (st + 1) & 1);
                acc[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[st & 1], f0[st & 1], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[st & 1], f1[st & 1], acc[1], 0, 0, 0);
            }

            if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (MODE == 5) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the residual words
                __syncthreads();                                           // every wave is done reading the patch: it becomes T
            }
            // ---- code epilogue (the straight-line form of mfma_gemm_kernel.h, mode 2, device BatchNorm arithmetic) ---------
            auto body = [&](auto rc_tag, auto relu_tag) {
                constexpr bool RC = decltype(rc_tag)::value, RELU = decltype(relu_tag)::value;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int n = n0 + b * 32 + (lane & 7) * 4;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        T[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * 32 + lrow] = (float)acc[b][r] * scale;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = i * 8 + (lane >> 3);
                        const float4 v4 = *reinterpret_cast<const float4*>(T + row * 32 + (lane & 7) * 4);
                        {   // (no per-row bounds branch: whole tiles only — a branch per row group puts every group in its own basic
                            //  block, and the waitcnt pass then drains the PREVIOUS group's store before each group's first use
                            //  of a residual word: eight store round trips per tile)
                            const uint32_t rword = rw[b][i];
                            uint32_t word = 0;
                            // two channels per instruction on the packed fp32 pipe (v_pk_add / v_pk_mul / v_pk_fma: the same IEEE
                            // roundings as the scalar forms, -ffp-contract=off keeps them separate)
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const c3_v2f vv = h ? (c3_v2f){v4.z, v4.w} : (c3_v2f){v4.x, v4.y};
                                const c3_v2f mm = {mean[b][2 * h], mean[b][2 * h + 1]}, rr = {rs[b][2 * h], rs[b][2 * h + 1]};
                                const c3_v2f aa = {al[b][2 * h], al[b][2 * h + 1]}, bb = {be[b][2 * h], be[b][2 * h + 1]};
                                c3_v2f tt = __builtin_elementwise_fma((vv - mm) * rr, aa, bb);
                                if constexpr (RC) {
                                    const c3_v2f rc = {(float)(int8_t)(rword >> (16 * h)), (float)(int8_t)(rword >> (16 * h + 8))};
                                    tt = tt + (c3_v2f){a.rscale, a.rscale} * rc;
                                }
                                if constexpr (RELU) {
                                    tt.x = tt.x < 0.0f ? 0.0f : tt.x;
                                    tt.y = tt.y < 0.0f ? 0.0f : tt.y;
                                }
                                const c3_v2f lt = (c3_v2f){a.levels, a.levels} * tt;
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    const float qf = rintf(e ? lt.y : lt.x);
                                    const bool ok = __builtin_fabsf(qf) <= 127.0f;       // NaN -> false
                                    const int qi = ok ? (int)qf : 0;
                                    badf |= ok ? 0 : 1;
                                    word |= (uint32_t)(uint8_t)(int8_t)qi << (8 * (2 * h + e));
                                }
                            }
                            *reinterpret_cast<uint32_t*>(a.Q + (int64_t)orow[i] * a.ldq + n) = word;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            };
            if constexpr (MODE == 4) {
                if (acc[0][0] == 0x7fffffff && acc[1][5] == 12345) a.Q[0] = 1;      // keep the MFMAs alive
            } else
            if (a.res_codes) {
                if (a.relu == 1) body(std::true_type{}, std::true_type{});
                else body(std::true_type{}, std::false_type{});
            } else {
                if (a.relu == 1) body(std::false_type{}, std::true_type{});
                else body(std::false_type{}, std::false_type{});
            }
            // the barrier joins the four waves' shares of the next patch (each waited for its own above) and tells that every
            // wave is done reading the current one
            if constexpr (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr (MODE == 5) {
                __syncthreads();                                           // the transpose patches are dead: the buffer may be refilled
                if (nxt < a.tiles_m) {
                    issue_patch(nxt, 0, 0, 256);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // exposed per workgroup; two other workgroups of the CU run meanwhile
                }
                __syncthreads();
                continue;
            }
            __syncthreads();
            cur ^= 1;
        }
        if (__any(badf) && lane == 0) atomicOr(a.overflow, 1);
    }
    if (a.ohy | a.ohx)
        c3_zero_border(a.Q, a.ldq / 16, a.Nimg, a.H, a.W, a.ohy, a.ohx, (int)blockIdx.x, (int)gridDim.x);
}

std::atomic<long long> c3_launches{0};

int c3_log2(int64_t v) { return (v > 0 && (v & (v - 1)) == 0) ? __builtin_ctzll((unsigned long long)v) : -1; }

}  // namespace

// Called by qt_conv2d_implicit_codes (mfma_gemm.hip) before the implicit-GEMM dispatch: QT_ERR_UNSUPPORTED = not this kernel's case.
int qt_code_conv3x3_try(const uint32_t* P, int64_t Nimg, int64_t H, int64_t W, int64_t Cw, int64_t kh, int64_t kw, int64_t sh, int64_t sw,
                        int64_t ph, int64_t pw, int64_t dh, int64_t dw, const uint32_t* Wmat, int64_t ldwp, const float* bias, float scale,
                        const float* scale_dev, const float* alpha, const float* beta, const float* res_f32, const float* res_alpha,
                        const int8_t* res_codes, int64_t ldrc_bytes, float res_scale, int relu, int bit_width, int8_t* codes,
                        int64_t ldc_bytes, int64_t Cout, int32_t* overflow, int64_t ihy, int64_t ihx, int64_t ohy, int64_t ohx,
                        int64_t rhy, int64_t rhx, const float* bn_stats, qt_stream_t stream) {
    if (getenv("QT_NO_CODE_CONV3X3")) return QT_ERR_UNSUPPORTED;          // A/B switch for tools and tests (read per call)
    const int64_t CB = Cw * 4;
    if (kh != 3 || kw != 3 || sh != 1 || sw != 1 || ph != 1 || pw != 1 || dh != 1 || dw != 1 || ihy < 1 || ihx < 1) return QT_ERR_UNSUPPORTED;
    if ((CB != 64 && CB != 128) || Cout <= 0 || (Cout & 63) || ldc_bytes != Cout || ldwp * 4 < 9 * CB) return QT_ERR_UNSUPPORTED;
    if (bias || res_f32 || res_alpha || !bn_stats || !alpha || !beta || relu < 0 || relu > 1) return QT_ERR_UNSUPPORTED;
    const int lw = c3_log2(W), lh = c3_log2(H);
    if (lw < 0 || lh < 0 || W > 128 || Nimg <= 0) return QT_ERR_UNSUPPORTED;
    const int64_t RT = 128 / W;                              // flattened output rows per tile
    if (!(RT <= H ? (H % RT == 0) : (RT % H == 0))) return QT_ERR_UNSUPPORTED;
    const int64_t Hp = H + 2 * ihy, Wp = W + 2 * ihx;
    if (Nimg * Hp * Wp * CB >= (1ll << 31) || Cout * ldwp * 4 >= (1ll << 31)) return QT_ERR_UNSUPPORTED;
    if (Nimg * (H + 2 * ohy) * (W + 2 * ohx) * ldc_bytes >= (1ll << 40) || 127ll * 9 * CB >= (1 << 24)) return QT_ERR_UNSUPPORTED;
    if (!qt_aligned16(P) || !qt_aligned16(Wmat) || !qt_aligned16(codes) || (ldwp & 3)) return QT_ERR_UNSUPPORTED;
    if ((Nimg * H * W) % 128) return QT_ERR_UNSUPPORTED;      // whole 128-pixel tiles only (no per-row bounds checks in the epilogue)
    const int64_t rows_total = Nimg * H;
    const int64_t tiles_m = (rows_total + RT - 1) / RT, tiles_n = Cout / 64;
    const int64_t patch_rows = RT <= H ? RT + 2 : (RT / H) * Hp;
    const int64_t patch_bytes = (patch_rows * Wp * CB + 255) / 256 * 256;
    // tool-only switches (occupancy / order / loader-mode experiments) are read ONCE per process: no environment walk on the
    // launch-bound module-graph path and no race with a concurrent setenv; the one A/B switch tests flip inside a
    // process, QT_NO_CODE_CONV3X3 above, stays per call as include/qt_hip.h documents
    struct C3Env { const char* mode; bool plain_order; int per_cu; };
    static const C3Env env = [] {
        C3Env e;
        e.mode = getenv("QT_C3_MODE");
        e.plain_order = getenv("QT_C3_PLAIN_ORDER") != nullptr;
        const char* pc = getenv("QT_C3_PER_CU");
        e.per_cu = pc ? atoi(pc) : 0;
        return e;
    }();
    const char* fm0 = env.mode;
#ifdef QT_PROFILING_VARIANTS
    const bool single = fm0 && atoi(fm0) == 5;                                  // profiling builds: the single-buffer variant
#else
    const bool single = false;
    (void)fm0;
#endif
    const int64_t lds = single ? 64 * 9 * CB + std::max<int64_t>(patch_bytes, 4 * 4096) : 64 * 9 * CB + 2 * patch_bytes + 4 * 4096;
    if (lds > 160 * 1024 || tiles_m * tiles_n > (1 << 30)) return QT_ERR_UNSUPPORTED;
    C3Args a;
    a.P = reinterpret_cast<const unsigned char*>(P);
    a.Wm = reinterpret_cast<const unsigned char*>(Wmat);
    a.ldw = (int)(ldwp * 4);
    a.Nimg = (int)Nimg; a.H = (int)H; a.W = (int)W; a.Hp = (int)Hp; a.Wp = (int)Wp; a.ihy = (int)ihy; a.ihx = (int)ihx;
    a.sh_w = lw; a.sh_h = lh;
    a.Cout = (int)Cout;
    a.rows_per_tile = (int)RT;
    a.tiles_m = (int)tiles_m; a.tiles_n = (int)tiles_n;
    a.weight = alpha; a.bias = beta; a.bn_stats = bn_stats;
    a.scale = scale; a.scale_dev = scale_dev;
    a.levels = (float)((1 << bit_width) - 1); a.rscale = res_scale; a.relu = relu;
    a.res_codes = res_codes; a.ldrc = (int)ldrc_bytes; a.rhy = (int)rhy; a.rhx = (int)rhx;
    a.Q = codes; a.ldq = (int)ldc_bytes; a.ohy = (int)ohy; a.ohx = (int)ohx;
    a.overflow = overflow;
    a.no_xcd_order = env.plain_order ? 1 : 0;
    // persistent grid: as many workgroups as stay resident (LDS-bound), a multiple of the column tiles
    int per_cu = (int)std::max<int64_t>(1, std::min<int64_t>(4, (160 * 1024) / lds));
    if (env.per_cu > 0) per_cu = std::max(1, std::min(per_cu, env.per_cu));     // tools: occupancy experiments
    int64_t grid = std::min<int64_t>(tiles_m * tiles_n, 256ll * per_cu);
    grid = std::max<int64_t>(tiles_n, grid / tiles_n * tiles_n);
    hipStream_t st = (hipStream_t)stream;
    // MODE 0: the compute waves issue the next patch's DMA themselves; MODE 2: a fifth (loader) wave does.  Measured per launch
    // (tools/probes/c3_modes.sh, batch 256): 64 -> 64 @ 32 x 32: 22.8 / 29.4 us, 128 -> 128 @ 16 x 16: 23.8 / 22.4 us
    // (implicit-GEMM kernel: 29.0 / 24.2 us).  QT_C3_MODE overrides (tools only; 1 / 3 / 4 exist in profiling builds).
    const char* fm = env.mode;
    const int forced = fm ? atoi(fm) : -1;
    const int mode = forced >= 0 ? forced : (CB == 64 ? 0 : 2);
#define QT_C3(CBV, MD)                                                                                                              \
    do {                                                                                                                            \
        static QtLdsOnce once;                                                                                                      \
        if (qt_ensure_dyn_lds(once, reinterpret_cast<const void*>(code_conv3x3_kernel<CBV, MD>), (int)lds) != QT_OK)               \
            return QT_ERR_LAUNCH;                                                                                                   \
        hipLaunchKernelGGL((code_conv3x3_kernel<CBV, MD>), dim3((unsigned)grid), dim3(MD == 2 ? 320 : 256), (size_t)lds, st, a,    \
                           (int)patch_bytes);                                                                                       \
    } while (0)
#ifdef QT_PROFILING_VARIANTS
    if (mode == 1 || mode == 3 || mode == 4 || mode == 5) {     // ablations: wait at the tile end / no main loop / no epilogue (results
        // wrong for 3, 4) / ONE patch buffer with the transpose patches aliased on it, 3 workgroups per CU (26.3 vs 23.4 us: slower)
        if (CB == 64) { if (mode == 1) QT_C3(64, 1); else if (mode == 3) QT_C3(64, 3); else if (mode == 4) QT_C3(64, 4); else QT_C3(64, 5); }
        else { if (mode == 1) QT_C3(128, 1); else if (mode == 3) QT_C3(128, 3); else if (mode == 4) QT_C3(128, 4); else QT_C3(128, 5); }
        return qt_check_launch();
    }
#endif
    if (CB == 64) {
        if (mode == 2) QT_C3(64, 2); else QT_C3(64, 0);
    } else {
        if (mode == 2) QT_C3(128, 2); else QT_C3(128, 0);
    }
#undef QT_C3
    c3_launches.fetch_add(1, std::memory_order_relaxed);
    return qt_check_launch();
}

extern "C" int64_t qt_code_conv3x3_launch_count(void) { return (int64_t)c3_launches.load(std::memory_order_relaxed); }
