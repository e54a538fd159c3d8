// Direct first-layer conv: a real-valued fp32 image with a few channels against a quantised weight, strided, large kernel
// (AlexNet conv1: 3 -> 192, 11 x 11, stride 4, padding 2 — models/Alexnet/Alexnet_Bin.py:13; BinConv2d / TerConv2d / XNORConv2d,
// layers/binary_layers.py:103-106, terner_layers.py:89-92, functions/xnor_connect.py:139-146).
//
// The implicit-GEMM route pays for this layer three times: a space-to-depth pack pass that writes and re-reads a 154 MB fp16
// plane, a K loop whose every output row gathers 1.7 KB into LDS (LDS-fill bound), and weight rows replicated per split term.
// Here a workgroup owns TOY x TOX output pixels (<= 128) x up to 192 output channels:
//   * the (TOY-1) s + kh  x  (TOX-1) s + kw  pixel PATCH of the fp32 image is read from HBM once (every loaded pixel feeds
//     ~ (kh / s)(kw / s) outputs x all channels), its max|x| is folded on the way and the patch goes to LDS as two fp16 planes
//     hi = fp16(x / s), lo = fp16(x / s - hi) with the TILE's own power-of-two s (max|x| / s in [2^14, 2^15)): no global max|x|
//     pass, no speculation, |x - s (hi + lo)| <= max(2^-22 |x|, 2^-39 tilemax);
//   * K = (ky, kx, c): for a fixed ky the kw * C values an output pixel needs are CONTIGUOUS in its patch row (NHWC, c fastest), so
//     the A fragment of v_mfma_f32_32x32x16_f16 (8 consecutive k per lane) is one 16-byte run of the LDS patch at
//     ((oy s + ky) RS + ox s C + 8 cc) — stride addressing replaces the space-to-depth plane; a ky row is ceil(kw C / 8)
//     chunks (AlexNet: 33 -> 40 elements), chunks of consecutive ky rows pair up into k-steps (28 instead of 33);
//   * the weight fragments come straight from global memory (L2) in the packed order [k-step][half][channel][8 fp16] — 16 bytes per
//     lane, lane-contiguous — prefetched one k-step ahead; +-1 / 0 weights are ONE fragment shared by the hi and lo terms of the
//     image (no replicated rows), real-valued weights (XNOR-Net: sign(W) * alpha) two fragments w / sw = whi + wlo and the three
//     products hi whi + lo whi + hi wlo;
//   * 4 waves, each 2 (m) x 3 (n) accumulator tiles of 32 x 32; two workgroups per CU overlap each other's patch prologue.
// Epilogues: fp32 NHWC (+ bias), or the BatchNorm-threshold bits of the fused inference chain (qt_conv2d_implicit_bits' float form).
#include <cstdlib>
#include "qt_common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct FirstArgs {
    const float* x;
    int64_t sn, sc, sh, sw;            // element strides of x [N, C, H, W] (any storage order)
    int N, C, H, W, KH, KW, S, PH, PW, Ho, Wo;
    int Cp;                            // channels per pixel in the LDS patch (C, or padded so that S * Cp % 4 == 0)
    int CPK;                           // 16-byte chunks per ky row = ceil(KW * Cp / 8)
    int NCH, NKS;                      // chunks = KH * CPK, k-steps = ceil(NCH / 2) rounded up to 4
    int TOY, TOX, tiles_y, tiles_x;
    int PR, PCE, RS;                   // patch rows, real elements per patch row (PC * Cp), LDS row stride in elements (% 4 == 0)
    unsigned long long m64_tpi;        // ceil(2^64 / tiles per image)
    unsigned m_tx;                     // magic of tiles_x
    unsigned m_ppr, m_cp, m_tox, m_cpk; // floor(2^32 / d) + 1 for d = RS / 2, Cp, TOX, CPK: q / d == __umulhi(q, m) for q < 2^16 (d > 1)
    const uint4* whi;                  // [NKS][2][Coutp] 16-byte chunks
    const uint4* wlo;                  // real-valued weights only
    float wscale;                      // weights were divided by this power of two before the fp16 split (1 for +-1 / 0)
    const float* wscale_dev;           // ... or by this device-resident one (times wscale): no host round trip for a training-mode weight
    int Cout, Coutp;
    const float* bias;
    float* y;                          // fp32 NHWC [N, Ho, Wo, ldy]   (alpha == nullptr)
    int64_t ldy;
    const float* alpha;                // threshold epilogue: bit = fl(fl(v * alpha) + beta) < 0  ->  bits[(n, oy, ox)][ldb words]
    const float* beta;
    uint32_t* bits;
    int64_t ldb;
    int step_r, step_pc, rows_dense;    // patch loader: 256 / (RS / 2), 256 % (RS / 2); channels-last image read as float2 spans
};

__device__ __forceinline__ int divm(int q, unsigned magic, int d) { return d == 1 ? q : (int)__umulhi((unsigned)q, magic); }

__device__ __forceinline__ v16f mfma16(const uint4& a, const uint4& b, v16f c) {
    h8 av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
}

// REALW: the weight is real-valued (two fp16 terms); otherwise +-1 / 0 (one fragment).  MAXP: pairs of patch elements a thread holds.
//
// Persistent workgroups (two per CU), tiles b, b + G, b + 2G, ...: the patch of the NEXT tile is requested from HBM into
// registers before the MFMA loop of the current tile starts and is converted / written to the other LDS patch buffer after it, so a
// workgroup's HBM latency and its split arithmetic sit under MFMA time (its own and the co-resident workgroup's) instead of in
// front of it — with one tile per workgroup the two workgroups of a CU run their prologues, loops and epilogues in lockstep.
template <bool REALW, bool BITS, int MAXP>
__global__ __launch_bounds__(256, 2) void conv_first_direct_kernel(FirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, lrow = lane & 31;
    const int tpi = a.tiles_y * a.tiles_x;
    const int ntiles = a.N * tpi;
    const int plane_bytes = a.PR * a.RS * 2;
    const int buf_bytes = 2 * plane_bytes;                                   // [hi plane | lo plane]
    float* red = reinterpret_cast<float*>(smem + 2 * buf_bytes);           // per buffer: 4 partial maxima, scale, 1 / scale
    const int ppr = a.RS >> 1;                   // pairs per LDS row

    // the patch in flight: a thread owns ONE pair column pc (two consecutive elements of a patch row) and walks the rows
    // rr, rr + rpp, ... (rpp = 256 / ppr rows per pass): everything that depends on the column — bounds against the image row and the
    // real patch width, the float2 alignment case — is evaluated once per tile, a pass costs a row check, one address and one load
    float v0[MAXP], v1[MAXP];
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const int rpp = a.step_r;                                   // rows per pass
    const int rr0 = divm(tid, a.m_ppr, ppr), pc0 = tid - rr0 * ppr;

    auto tile_origin = [&](int tile, int& img, int& oy0, int& ox0) __attribute__((always_inline)) {
        img = tpi == 1 ? tile : (int)__umul64hi((unsigned long long)(unsigned)tile, a.m64_tpi);     // exact for 32-bit numerators
        const int trem = tile - img * tpi;
        const int ty = divm(trem, a.m_tx, a.tiles_x), tx = trem - ty * a.tiles_x;
        oy0 = ty * a.TOY;
        ox0 = tx * a.TOX;
    };

    // ---- patch: HBM -> registers ---------------------------------------------------------------------------------------------------
    auto issue_patch = [&](int tile) __attribute__((always_inline)) {
        int img, oy0, ox0;
        tile_origin(tile, img, oy0, ox0);
        const int iy0 = oy0 * a.S - a.PH, ix0 = ox0 * a.S - a.PW;          // patch origin in the image (may be negative: padding)
        const float* xi = a.x + (int64_t)img * a.sn;
        int rr = rr0, pc = pc0;
        asm volatile("" : "+v"(rr), "+v"(pc));      // opaque per tile: what follows is loop-invariant and would be hoisted into spills
        const int e = 2 * pc;
        const bool act = rr < rpp && e < a.PCE;
        int64_t off0 = 0, off1 = 0;                 // element offsets of the pair within an image row
        bool ok0 = false, ok1 = false, vec = false;
        if (a.rows_dense) {
            // channels-last image, no channel padding: a patch row is ONE contiguous span of the image row starting at element
            // ix0 * C (even: 8-byte aligned float2 loads); spans are clipped against the row for the zero padding
            const int g = ix0 * a.C + e, gend = a.W * a.C;
            ok0 = act && g >= 0 && g < gend;
            ok1 = act && g + 1 >= 0 && g + 1 < gend && e + 1 < a.PCE;
            vec = ok0 && ok1;
            off0 = g;
            off1 = g + 1;
        } else {
            const int px0 = divm(e, a.m_cp, a.Cp), c0 = e - px0 * a.Cp;
            int px1 = px0, c1 = c0 + 1;
            if (c1 == a.Cp) { c1 = 0; ++px1; }
            const int ixa = ix0 + px0, ixb = ix0 + px1;
            ok0 = act && c0 < a.C && (unsigned)ixa < (unsigned)a.W;
            ok1 = act && e + 1 < a.PCE && c1 < a.C && (unsigned)ixb < (unsigned)a.W;
            off0 = (int64_t)ixa * a.sw + (int64_t)c0 * a.sc;
            off1 = (int64_t)ixb * a.sw + (int64_t)c1 * a.sc;
        }
        const float* p0 = xi + off0;
        const float* p1 = xi + off1;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int r = rr + i * rpp, iy = iy0 + r;
            const bool rok = r < a.PR && (unsigned)iy < (unsigned)a.H;
            const int64_t ro = (int64_t)iy * a.sh;
            float f0 = 0.0f, f1 = 0.0f;
            if (rok && vec) {
                const float2 v = *reinterpret_cast<const float2*>(p0 + ro);
                f0 = v.x;
                f1 = v.y;
            } else if (rok) {
                if (ok0) f0 = p0[ro];
                if (ok1) f1 = p1[ro];
            }
            v0[i] = f0;
            v1[i] = f1;
        }
    };

    // ---- registers -> two fp16 planes in LDS buffer `buf` with the tile's power-of-two scale (two barriers) -----------------------
    auto finish_patch = [&](int buf) __attribute__((always_inline)) {
        unsigned mx = 0;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) mx = max(mx, max(__float_as_uint(v0[i]) & 0x7fffffffu, __float_as_uint(v1[i]) & 0x7fffffffu));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        float* rb = red + buf * 8;
        if (lane == 0) rb[wave] = __uint_as_float(mx);
        __syncthreads();           // every wave is past the MFMA loop that read this buffer's predecessor; partial maxima visible
        // s = 2^(e - 14) with e the exponent of max|x|: max|x| / s in [2^14, 2^15).  max|x| == 0 / subnormal, inf, NaN: s = 1 (an
        // inf / NaN pixel then poisons its outputs through fp16 inf / NaN, as it does in the reference's fp32 conv)
        const unsigned m = max(max(__float_as_uint(rb[0]), __float_as_uint(rb[1])), max(__float_as_uint(rb[2]), __float_as_uint(rb[3])));
        const int eb = (int)(m >> 23);
        float sc = 1.0f;
        if (eb > 0 && eb < 255) {
            int se = eb - 14;                                  // biased exponent of s
            se = se < 1 ? 1 : (se > 254 ? 254 : se);
            sc = __uint_as_float((unsigned)se << 23);
        }
        const float isx = 1.0f / sc;                           // exact (power of two within the normal range)
        if (tid == 0) rb[4] = sc;
        h2* hi2 = reinterpret_cast<h2*>(smem + buf * buf_bytes);
        h2* lo2 = reinterpret_cast<h2*>(smem + buf * buf_bytes + plane_bytes);
        int rr = rr0, pc = pc0;
        asm volatile("" : "+v"(rr), "+v"(pc));
        if (rr < rpp) {
            int p = rr * ppr + pc;
            const int dp = rpp * ppr;
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                if (rr + i * rpp < a.PR) {
                    const f2 t = (f2){v0[i], v1[i]} * isx;
                    const h2 h = __builtin_convertvector(t, h2);
                    const h2 l = __builtin_convertvector(t - __builtin_convertvector(h, f2), h2);
                    hi2[p] = h;
                    lo2[p] = l;
                }
                p += dp;
            }
        }
        __syncthreads();
    };

    // ---- per-wave constants of the MFMA loop -----------------------------------------------------------------------------------------
    const int mg = wave >> 1, ng = wave & 1;                   // this wave: m-tiles 2 mg, 2 mg + 1; n-tiles 3 ng .. 3 ng + 2
    const int npix = a.TOY * a.TOX;
    int abase[2];                                              // byte offset of the pixel's run start (ky = 0, chunk 0) in a plane
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int j = min((mg * 2 + mt) * 32 + lrow, npix - 1);
        const int oyl = divm(j, a.m_tox, a.TOX), oxl = j - oyl * a.TOX;
        abase[mt] = ((oyl * a.S) * a.RS + oxl * a.S * a.Cp) * 2;
    }
    const int nt0 = blockIdx.y * 6 + ng * 3;                   // first of this wave's three 32-channel tiles
    // (branch-free main loop: a wave whose channel tiles lie past Coutp computes on the last valid tile's fragments and drops the
    // result; the k-step count is a multiple of 4 — zero k-steps appended by the weight pack — and loads past the end re-read the
    // last k-step.  With conditionals in the loop hipcc puts s_waitcnt vmcnt(0) in front of every MFMA group: the whole L2
    // latency of the prefetch just issued, per k-step.)
    const int ntl = a.Coutp / 32 - 1;
    int64_t wbase[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) wbase[t] = (int64_t)half * a.Coutp + (int64_t)min(nt0 + t, ntl) * 32 + lrow;
    const int64_t wstep = 2 * (int64_t)a.Coutp;                 // 16-byte chunks per k-step
    auto load_w = [&](int s, uint4 (&wh)[3], uint4 (&wl)[3]) __attribute__((always_inline)) {
        const int64_t o = (int64_t)min(s, a.NKS - 1) * wstep;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            wh[t] = a.whi[o + wbase[t]];
            if constexpr (REALW) wl[t] = a.wlo[o + wbase[t]];
        }
    };

    // per-channel epilogue constants of this lane's three channels: loaded once per (persistent) workgroup
    float e_bv[3], e_al[3], e_nbe[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (nt0 + t) * 32 + lrow;
        const bool nin = n < a.Cout;
        e_bv[t] = (a.bias && nin) ? a.bias[n] : 0.0f;
        e_al[t] = (BITS && nin) ? a.alpha[n] : 0.0f;
        e_nbe[t] = (BITS && nin) ? -a.beta[n] : 0.0f;
    }
    const float wsc = a.wscale_dev ? a.wscale * *a.wscale_dev : a.wscale;
    // rotated tile loop — every phase exists ONCE in the code (the unrolled patch passes and the unrolled epilogue are large; inlined
    // twice, with both epilogues, the kernel was ~70 KB of instructions and ran out of the instruction cache: every phase, loads or
    // not, took 4-8x its instruction count):   [request patch of `load_tile`]  [MFMA loop + epilogue of `tile`]  [patch -> LDS]
    int load_tile = blockIdx.x, tile = -1, buf = 0;
    while (true) {
        const bool more = load_tile < ntiles;                  // uniform
        if (more) issue_patch(load_tile);                      // in flight across the MFMA loop below
        if (tile >= 0) {

        const unsigned char* hib = smem + buf * buf_bytes;
        const unsigned char* lob = hib + plane_bytes;
        auto load_a = [&](int s, uint4 (&ah)[2], uint4 (&al)[2]) __attribute__((always_inline)) {
            // this lane's chunk of the k-step: q = 2 s + half -> (ky, cc); chunks past the last one re-read the last chunk (zero weights)
            const int q = min(2 * s + half, a.NCH - 1);
            const int ky = divm(q, a.m_cpk, a.CPK), cc = q - ky * a.CPK;
            const int koff = (ky * a.RS + cc * 8) * 2;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const uint2 h0 = *reinterpret_cast<const uint2*>(hib + abase[mt] + koff);
                const uint2 h1 = *reinterpret_cast<const uint2*>(hib + abase[mt] + koff + 8);
                const uint2 l0 = *reinterpret_cast<const uint2*>(lob + abase[mt] + koff);
                const uint2 l1 = *reinterpret_cast<const uint2*>(lob + abase[mt] + koff + 8);
                ah[mt] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                al[mt] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        };
        v16f acc[2][3];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][t][r] = 0.0f;
        // software pipeline: the weight fragments of k-step s + WD (L2 latency ~ 2-3 k-steps of MFMA time) and the patch fragments
        // of k-step s + 1 (LDS latency) are requested before the MFMAs of k-step s issue; rings indexed by the unrolled position
        constexpr int WD = 1, RING = 2;                          // (the unrolled group of 4 k-steps must be a multiple of the ring)
        static_assert(4 % RING == 0, "ring slots are indexed by the unrolled position");
        uint4 wh[RING][3], wl[RING][3], ah[2][2], al[2][2];
#pragma unroll
        for (int u = 0; u < WD; ++u) load_w(u, wh[u], wl[u]);
        load_a(0, ah[0], al[0]);
        for (int s0 = 0; s0 < a.NKS; s0 += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int s = s0 + u;
                load_w(s + WD, wh[(u + WD) % RING], wl[(u + WD) % RING]);
                load_a(s + 1, ah[(u + 1) & 1], al[(u + 1) & 1]);
                // one term at a time over the six accumulator tiles: two MFMAs on the same accumulator are never adjacent
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt][t] = mfma16(ah[u & 1][mt], wh[u % RING][t], acc[mt][t]);
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt][t] = mfma16(al[u & 1][mt], wh[u % RING][t], acc[mt][t]);
                if constexpr (REALW) {
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt) acc[mt][t] = mfma16(ah[u & 1][mt], wl[u % RING][t], acc[mt][t]);
                }
            }
        }

        // ---- epilogue: lane owns channel n = tile * 32 + lrow, rows (r & 3) + 8 (r >> 2) + 4 half of each 32-pixel tile ------------
        // (the lane's coordinates are made opaque per tile: everything below that depends only on them — 32 pixel offsets, the
        // channel constants — is loop-invariant, and hipcc otherwise hoists it out of the tile loop into 200 spilled registers)
        int lane_t = lane, half_t = half, lrow_t = lrow;
        asm volatile("" : "+v"(lane_t), "+v"(half_t), "+v"(lrow_t));
        int img, oy0, ox0;
        tile_origin(tile, img, oy0, ox0);
        const float sx = red[buf * 8 + 4];
        const float oscale = sx * wsc;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if ((nt0 + t) * 32 >= a.Coutp) continue;
            const int n = (nt0 + t) * 32 + lrow_t;
            const bool nin = n < a.Cout;
            const float bv = e_bv[t], al_ = e_al[t], nbe = e_nbe[t];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int jb = (mg * 2 + mt) * 32;
                if constexpr (BITS) {
                    uint32_t myword = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float v = acc[mt][t][r] * oscale + bv;
                        const unsigned long long mask = __ballot(v * al_ < nbe);      // channels >= Cout: 0 < 0 -> bit 0
                        const int R = (r & 3) + 8 * (r >> 2);
                        // v_writelane: the two halves of the (scalar) ballot straight into lanes R and R + 4
                        // (gfx950 does not interlock a VALU-written SGPR read by the next VALU and the hazard recogniser does not look
                        // inside inline asm: the s_nop covers v_cmp -> first write; the writes are chained through myword)
                        asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(myword) : "s"((uint32_t)mask), "n"(R));
                        asm("v_writelane_b32 %0, %1, %2" : "+v"(myword) : "s"((uint32_t)(mask >> 32)), "n"(R + 4));
                    }
                    const int j = jb + lane_t;                                          // lanes 0 .. 31: one pixel each
                    if (lane_t < 32 && j < npix) {
                        const int oyl = divm(j, a.m_tox, a.TOX), oxl = j - oyl * a.TOX;
                        const int oy = oy0 + oyl, ox = ox0 + oxl;
                        if (oy < a.Ho && ox < a.Wo) {
                            uint32_t* row = a.bits + (((int64_t)img * a.Ho + oy) * a.Wo + ox) * a.ldb;
                            row[nt0 + t] = myword;
                            // the row's pad words (ldb rounds ceil(Cout / 32) up): written once, by the tile that holds the last channels
                            if ((nt0 + t + 1) * 32 >= a.Coutp)
                                for (int wc = nt0 + t + 1; wc < a.ldb; ++wc) row[wc] = 0u;
                        }
                    }
                } else {
                  if (nin) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int j = jb + (r & 3) + 8 * (r >> 2) + 4 * half_t;
                        if (j < npix) {
                            const int oyl = divm(j, a.m_tox, a.TOX), oxl = j - oyl * a.TOX;
                            const int oy = oy0 + oyl, ox = ox0 + oxl;
                            if (oy < a.Ho && ox < a.Wo)
                                a.y[(((int64_t)img * a.Ho + oy) * a.Wo + ox) * a.ldy + n] = acc[mt][t][r] * oscale + bv;
                        }
                    }
                  }
                }
            }
        }
        }
        if (!more) break;
        const int nb = tile >= 0 ? (buf ^ 1) : 0;
        finish_patch(nb);
        tile = load_tile;
        load_tile += gridDim.x;
        buf = nb;
    }
}

int first_direct_impl(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int64_t N, int64_t C, int64_t H, int64_t W,
                      int64_t KH, int64_t KW, int64_t S, int64_t PH, int64_t PW, int64_t Cp, const uint32_t* whi, const uint32_t* wlo,
                      float wscale, const float* wscale_dev, int64_t Cout, int64_t Coutp, const float* bias, float* y, int64_t ldy,
                      const float* alpha, const float* beta, uint32_t* bits, int64_t ldb, qt_stream_t stream) {
    if (N < 0 || C <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || S <= 0 || PH < 0 || PW < 0 || Cout < 0 || Cp < C)
        return QT_ERR_INVALID_ARG;
    const int64_t Ho = (H + 2 * PH - KH) / S + 1, Wo = (W + 2 * PW - KW) / S + 1;
    if (Ho <= 0 || Wo <= 0) return QT_ERR_INVALID_ARG;
    if (N == 0 || Cout == 0) return QT_OK;
    if (!x || !whi || (Coutp & 31) || Coutp < Cout || !qt_aligned16(whi) || (wlo && !qt_aligned16(wlo))) return QT_ERR_INVALID_ARG;
    if (alpha ? (!beta || !bits || ldb < (Cout + 31) / 32) : (!y || ldy < Cout)) return QT_ERR_INVALID_ARG;
    if ((S * Cp) & 3) return QT_ERR_ALIGNMENT;                        // a pixel's run starts on an 8-byte LDS boundary
    if (Cp > 8 || KW * Cp > 256 || KH > 64 || N * Ho * Wo > INT32_MAX || H > 32767 || W > 32767) return QT_ERR_UNSUPPORTED;
    FirstArgs a;
    a.x = x; a.sn = sn; a.sc = sc; a.sh = sh; a.sw = sw;
    a.N = (int)N; a.C = (int)C; a.H = (int)H; a.W = (int)W; a.KH = (int)KH; a.KW = (int)KW; a.S = (int)S; a.PH = (int)PH; a.PW = (int)PW;
    a.Ho = (int)Ho; a.Wo = (int)Wo; a.Cp = (int)Cp;
    a.CPK = (int)((KW * Cp + 7) / 8);
    a.NCH = (int)(KH * a.CPK);
    a.NKS = ((a.NCH + 1) / 2 + 3) / 4 * 4;           // whole groups of 4 k-steps (the weight pack appends zero k-steps)
    // output tile: <= 128 pixels, as square as the map allows, sized to waste the fewest padded pixels
    int best_ty = 1, best_tx = 1;
    double best = 1e30;
    for (int ty = 1; ty <= 128; ++ty)
        for (int tx = 1; tx * ty <= 128; ++tx) {
            if (tx * ty < 64 && tx * ty < Ho * Wo) continue;
            const int64_t pr = (int64_t)(ty - 1) * S + KH, pce = ((int64_t)(tx - 1) * S + KW) * Cp;
            const int64_t rs = (std::max<int64_t>(pce, (int64_t)(tx - 1) * S * Cp + a.CPK * 8) + 3) / 4 * 4;
            if (rs / 2 > 256 || (pr + 256 / (rs / 2) - 1) / (256 / (rs / 2)) > 18) continue;          // <= 18 row passes of the patch loader
            if (2 * (2 * pr * rs * 2) + 64 > 76 * 1024) continue;     // two patch buffers, two workgroups per CU
            const int64_t tiles = ((Ho + ty - 1) / ty) * ((Wo + tx - 1) / tx);
            // cost: MFMA work (128 rows per tile whatever it holds) + the patch it loads (halo re-reads)
            const double cost = (double)tiles * (128.0 * a.NKS * 16 + 0.25 * (double)(pr * rs));
            if (cost < best) { best = cost; best_ty = ty; best_tx = tx; }
        }
    if (best > 1e29) return QT_ERR_UNSUPPORTED;
    a.TOY = best_ty; a.TOX = best_tx;
    a.tiles_y = (int)((Ho + a.TOY - 1) / a.TOY); a.tiles_x = (int)((Wo + a.TOX - 1) / a.TOX);
    a.PR = (a.TOY - 1) * a.S + a.KH;
    a.PCE = ((a.TOX - 1) * a.S + a.KW) * a.Cp;
    a.RS = (std::max(a.PCE, (a.TOX - 1) * a.S * a.Cp + a.CPK * 8) + 3) / 4 * 4;
    auto magic = [](int d) { return d > 1 ? (unsigned)((1ull << 32) / (unsigned)d + 1) : 0u; };
    a.m_tx = magic(a.tiles_x);
    { const unsigned long long d = (unsigned long long)a.tiles_y * a.tiles_x; a.m64_tpi = d > 1 ? ~0ull / d + 1 : 0; }
    a.m_ppr = magic(a.RS / 2); a.m_cp = magic(a.Cp); a.m_tox = magic(a.TOX); a.m_cpk = magic(a.CPK);
    a.whi = reinterpret_cast<const uint4*>(whi);
    a.wlo = reinterpret_cast<const uint4*>(wlo);
    a.wscale = wscale;
    a.wscale_dev = wscale_dev;
    a.Cout = (int)Cout; a.Coutp = (int)Coutp;
    a.step_r = 256 / (a.RS / 2); a.step_pc = 0;          // rows of the patch one pass of the loader covers
    // float2 spans: dense channels-last rows, no channel padding, every patch row starts on an even element of an 8-byte aligned row
    a.rows_dense = (sc == 1 && sw == C && Cp == C && !(sh & 1) && !(sn & 1) && !((a.TOX * S * C) & 1) && !((PW * C) & 1) &&
                    (reinterpret_cast<uintptr_t>(x) & 7) == 0) ? 1 : 0;
    a.bias = bias; a.y = y; a.ldy = ldy; a.alpha = alpha; a.beta = beta; a.bits = bits; a.ldb = ldb;
    const int64_t ntiles = N * a.tiles_y * a.tiles_x;
    if (ntiles > INT32_MAX) return QT_ERR_UNSUPPORTED;
    const int lds = 2 * (2 * a.PR * a.RS * 2) + 64;                   // two patch buffers (hi + lo planes each) + the scales
    const unsigned ny = (unsigned)((Coutp + 191) / 192);
    // persistent workgroups: two per CU (256 CUs), shared between the channel blocks
    const dim3 grid((unsigned)std::min<int64_t>(ntiles, std::max<int64_t>(1, 512 / ny)), ny);
#define QT_FIRST_LAUNCH(REALW, BITS, MAXP)                                                                                          \
    do {                                                                                                                      \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_first_direct_kernel<REALW, BITS, MAXP>),                         \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return QT_ERR_LAUNCH;          \
        hipLaunchKernelGGL((conv_first_direct_kernel<REALW, BITS, MAXP>), grid, dim3(256), lds, (hipStream_t)stream, a);            \
    } while (0)
    if (wlo) { if (alpha) QT_FIRST_LAUNCH(true, true, 18); else QT_FIRST_LAUNCH(true, false, 18); }
    else { if (alpha) QT_FIRST_LAUNCH(false, true, 18); else QT_FIRST_LAUNCH(false, false, 18); }
#undef QT_FIRST_LAUNCH
    return qt_check_launch();
}

}  // namespace

extern "C" {

int qt_conv_first_direct_f32(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int64_t N, int64_t C, int64_t H,
                             int64_t W, int64_t KH, int64_t KW, int64_t S, int64_t PH, int64_t PW, int64_t Cp, const uint32_t* w_hi,
                             const uint32_t* w_lo, float w_scale, const float* w_scale_dev, int64_t Cout, int64_t Coutp,
                             const float* bias, float* y, int64_t ldy, qt_stream_t stream) {
    return first_direct_impl(x, sn, sc, sh, sw, N, C, H, W, KH, KW, S, PH, PW, Cp, w_hi, w_lo, w_scale, w_scale_dev, Cout, Coutp, bias,
                             y, ldy, nullptr, nullptr, nullptr, 0, stream);
}

int qt_conv_first_direct_bits_f32(const float* x, int64_t sn, int64_t sc, int64_t sh, int64_t sw, int64_t N, int64_t C, int64_t H,
                                  int64_t W, int64_t KH, int64_t KW, int64_t S, int64_t PH, int64_t PW, int64_t Cp,
                                  const uint32_t* w_hi, const uint32_t* w_lo, float w_scale, const float* w_scale_dev, int64_t Cout,
                                  int64_t Coutp, const float* bias, const float* alpha, const float* beta, uint32_t* neg_plane,
                                  int64_t ldb, qt_stream_t stream) {
    if (!alpha || !beta) return QT_ERR_INVALID_ARG;
    return first_direct_impl(x, sn, sc, sh, sw, N, C, H, W, KH, KW, S, PH, PW, Cp, w_hi, w_lo, w_scale, w_scale_dev, Cout, Coutp, bias,
                             nullptr, 0, alpha, beta, neg_plane, ldb, stream);
}

}  // extern "C"
