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
//     the A fragment of v_mfma_f32_32x32x16_f16 (8 consecutive k per lane) is TWO 8-byte runs (4 k each) of the LDS patch at
//     ((oy s + ky) RS + ox s C + 4 g) — stride addressing replaces the space-to-depth plane; a ky row is ceil(kw C / 4)
//     groups of 4 (AlexNet: 33 -> 36 elements; with 16-byte runs it was 40), the groups of all ky rows are strung together
//     four to a k-step (25 k-steps; 28 with 16-byte runs, 36 on the space-to-depth route);
//   * the weight fragments come straight from global memory (L2) in the packed order [k-step][half][channel][8 fp16] — 16 bytes per
//     lane, lane-contiguous — prefetched one k-step ahead; +-1 / 0 weights are ONE fragment shared by the hi and lo terms of the
//     image (no replicated rows), real-valued weights (XNOR-Net: sign(W) * alpha) two fragments w / sw = whi + wlo and the three
//     products hi whi + lo whi + hi wlo;
//   * 4 waves, each 2 (m) x 3 (n) accumulator tiles of 32 x 32; two workgroups per CU overlap each other's patch prologue.
// Epilogues: fp32 NHWC (+ bias), or the BatchNorm-threshold bits of the fused inference chain (qt_conv2d_implicit_bits' float form).
#include <cstdlib>
#include <type_traits>
#include "qt_common.h"

#ifdef QT_PROFILING_VARIANTS
// profiling builds only (tools/probes/stamps_conv1.py): shader-clock cycles wave 0 of every workgroup spends per phase, summed
__device__ unsigned long long qt_first_stamps[40];
#define QT_FS_DECL unsigned long long fs_t = clock64(), fs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define QT_FS(i) do { const unsigned long long t_ = clock64(); fs_acc[i] += t_ - fs_t; fs_t = t_; } while (0)
#define QT_FS_FLUSH do { if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&qt_first_stamps[(threadIdx.x >> 6) * 8 + i_], fs_acc[i_]); if (threadIdx.x == 0) atomicAdd(&qt_first_stamps[32], 1ull); } } while (0)
#else
#define QT_FS_DECL
#define QT_FS(i)
#define QT_FS_FLUSH
#endif

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct FirstArgs {
    const float* x;
    int64_t sn, sc, sh, sw;            // element strides of x [N, C, H, W] (any storage order)
    int N, C, H, W, KH, KW, S, PH, PW, Ho, Wo;
    int Cp;                            // channels per pixel in the LDS patch (C, or padded so that S * Cp % 4 == 0)
    int G4;                            // 8-byte groups (4 k) per ky row = ceil(KW * Cp / 4)
    int NG4, NKS;                      // groups = KH * G4, k-steps = ceil(NG4 / 4)
    int TOY, TOX, tiles_y, tiles_x;
    int PR, PCE, RS;                   // patch rows, real elements per patch row (PC * Cp), LDS row stride in elements (% 4 == 0)
    unsigned long long m64_tpi;        // ceil(2^64 / tiles per image)
    unsigned m_tx;                     // magic of tiles_x
    unsigned m_ppr, m_cp, m_tox, m_g4;  // floor(2^32 / d) + 1 for d = RS / 2, Cp, TOX, G4: q / d == __umulhi(q, m) for q < 2^16 (d > 1)
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
    QT_FS_DECL;
    const int tpi = a.tiles_y * a.tiles_x;
    const int ntiles = a.N * tpi;
    const int plane_bytes = a.PR * a.RS * 2;
    const int buf_bytes = 2 * plane_bytes;                                   // [hi plane | lo plane]
    float* red = reinterpret_cast<float*>(smem + 2 * buf_bytes);           // per buffer: 4 partial maxima, scale, 1 / scale
    // fp32 epilogue: per patch buffer, the byte offset of each of the tile's 128 pixels within its image's output plane
    // ((oy Wo + ox) ldy 4; OOB for pixels past the tile / the map) — written with the patch, read as 16-byte runs by the epilogue
    int* otab = reinterpret_cast<int*>(smem + 2 * buf_bytes + 64);
    // per (k-step, lane half): the byte offsets of the half's two 8-byte groups within a pixel's patch (ky RS + 4 g) 2 — filled once
    // per workgroup (the tile loop's first barriers order it before the first read), read one k-step ahead by the MFMA loop:
    // the scalar division chain it replaces was 51 SALU instructions per k-step
    int2* ktab = reinterpret_cast<int2*>(smem + 2 * buf_bytes + 64 + 1024);
    for (int i = tid; i < 2 * a.NKS; i += 256) {
        int o2[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int g = min(4 * (i >> 1) + 2 * (i & 1) + j, a.NG4 - 1);        // groups past the last one re-read it (zero weights)
            const int ky = divm(g, a.m_g4, a.G4);
            o2[j] = (ky * a.RS + (g - ky * a.G4) * 4) * 2;
        }
        ktab[i] = make_int2(o2[0], o2[1]);
    }
    constexpr int OOB = 0x40000000;                              // two of them add up to 2^31: still past any buffer extent
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
        if (a.rows_dense) {
            // channels-last image, no channel padding: a patch row is ONE contiguous span of the image row starting at element
            // ix0 * C + e, e even — 8-byte aligned pairs, and a pair lies inside the image row or outside it as a whole (row length
            // W * C and ix0 * C are even).  The image is a BUFFER: a pair outside the row / the image / the patch gets an offset past
            // the buffer's extent and the bounds check returns zeros — one load per pass for every lane, no branch, no mask kept
            // across the MFMA loop.  (With a conditional scalar path in the loop, ONE lane holding the odd last element of a patch
            // row made its whole wave execute both paths in every pass.)
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xi), 0, a.H * (int)a.sh * 4, 0x00020000);
            const int g = ix0 * a.C + e;
            // ... and the image ROWS too: the offset of row iy is cb + iy * rb4 as an unsigned number — rows above the image wrap
            // past 2^31, rows below it land at or past the extent (H * rb4 < 2^30), an invalid column starts from COLOOB, which
            // stays past the extent for every row of the patch.  One add per pass; only the rows past the PATCH (r >= PR) are
            // masked, and those passes are the last ones
            constexpr int COLOOB = 0x60000000;
            const int rb4 = (int)a.sh * 4;
            int off = ((act && g >= 0 && g < a.W * a.C) ? g * 4 : COLOOB) + (iy0 + rr) * rb4;
            const int step = rpp * rb4;
            const int rlim = a.PR - rr;                      // pass i is inside the patch iff i * rpp < rlim
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const u2 v = __builtin_amdgcn_raw_buffer_load_b64(xr, i * rpp < rlim ? off : COLOOB, 0, 0);
                v0[i] = __uint_as_float(v.x);
                v1[i] = __uint_as_float(v.y);
                off += step;
            }
        } else {
            // any layout: two scalar loads per pass from clamped (valid) addresses, zeroed by a select
            const int px0 = divm(e, a.m_cp, a.Cp), c0 = e - px0 * a.Cp;
            int px1 = px0, c1 = c0 + 1;
            if (c1 == a.Cp) { c1 = 0; ++px1; }
            const int ixa = ix0 + px0, ixb = ix0 + px1;
            const bool ok0 = act && c0 < a.C && (unsigned)ixa < (unsigned)a.W;
            const bool ok1 = act && e + 1 < a.PCE && c1 < a.C && (unsigned)ixb < (unsigned)a.W;
            const float* p0 = xi + (ok0 ? (int64_t)ixa * a.sw + (int64_t)c0 * a.sc : 0);
            const float* p1 = xi + (ok1 ? (int64_t)ixb * a.sw + (int64_t)c1 * a.sc : 0);
#pragma unroll
            for (int i = 0; i < MAXP; ++i) {
                const int r = rr + i * rpp, iy = iy0 + r;
                const bool rok = r < a.PR && (unsigned)iy < (unsigned)a.H;
                const int64_t ro = (int64_t)(rok ? iy : 0) * a.sh;
                const float f0 = p0[ro], f1 = p1[ro];
                v0[i] = (rok && ok0) ? f0 : 0.0f;
                v1[i] = (rok && ok1) ? f1 : 0.0f;
            }
        }
    };

    // ---- registers -> two fp16 planes in LDS buffer `buf` with the tile's power-of-two scale (two barriers) -----------------------
    auto finish_patch = [&](int buf, int tile_in) __attribute__((always_inline)) {
        if constexpr (!BITS) {
            int j = tid;
            asm volatile("" : "+v"(j));
            if (j < 128) {
                int img, oy0, ox0;
                tile_origin(tile_in, img, oy0, ox0);
                const int oyl = divm(j, a.m_tox, a.TOX), oxl = j - oyl * a.TOX;
                const int oy = oy0 + oyl, ox = ox0 + oxl;
                const bool ok = j < a.TOY * a.TOX && oy < a.Ho && ox < a.Wo;
                otab[buf * 128 + j] = ok ? (int)(((int64_t)oy * a.Wo + ox) * a.ldy * 4) : OOB;
            }
        }
        {   // the element past an odd patch row: the dense loader fetched image data there
            int pc = pc0;
            asm volatile("" : "+v"(pc));
            const bool tail = 2 * pc + 1 >= a.PCE;
#pragma unroll
            for (int i = 0; i < MAXP; ++i) v1[i] = tail ? 0.0f : v1[i];
        }
        unsigned mx = 0;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) mx = max(mx, max(__float_as_uint(v0[i]) & 0x7fffffffu, __float_as_uint(v1[i]) & 0x7fffffffu));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        float* rb = red + buf * 8;
        if (lane == 0) rb[wave] = __uint_as_float(mx);
        QT_FS(4);
        __syncthreads();           // every wave is past the MFMA loop that read this buffer's predecessor; partial maxima visible
        // s = 2^(e - 14) with e the exponent of max|x|: max|x| / s in [2^14, 2^15).  max|x| == 0 / subnormal, inf, NaN: s = 1 (an
        // inf / NaN pixel then poisons its outputs through fp16 inf / NaN, as it does in the reference's fp32 conv)
        QT_FS(5);
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
        QT_FS(6);
        __syncthreads();
        QT_FS(7);
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
    // the weight planes are BUFFERS: per-lane byte offset of the lane's chunk within a k-step (constant), k-step offset in an SGPR —
    // no vector address arithmetic in the loop
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    int wbase[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) wbase[t] = (half * a.Coutp + min(nt0 + t, ntl) * 32 + lrow) * 16;
    const int wstep = 2 * a.Coutp * 16;                        // bytes per k-step
    const int wextent = a.NKS * wstep;
    const __amdgpu_buffer_rsrc_t whr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.whi), 0, wextent, 0x00020000);
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t wlr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(REALW ? a.wlo : a.whi), 0, wextent, 0x00020000);
    auto load_w = [&](int s, uint4 (&wh)[3], uint4 (&wl)[3]) __attribute__((always_inline)) {
        const int o = min(s, a.NKS - 1) * wstep;                // uniform
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const u4v h = __builtin_amdgcn_raw_buffer_load_b128(whr, wbase[t], o, 0);
            wh[t] = make_uint4(h.x, h.y, h.z, h.w);
            if constexpr (REALW) {
                const u4v l = __builtin_amdgcn_raw_buffer_load_b128(wlr, wbase[t], o, 0);
                wl[t] = make_uint4(l.x, l.y, l.z, l.w);
            }
        }
    };

    // per-channel epilogue constants of this lane's three channels: loaded once per (persistent) workgroup
    float e_bv[3];
    int e_co[3];                                               // fp32 epilogue: byte offset of the lane's channel within a pixel, or OOB
    // threshold epilogue: bit = fl(fl(fl(u) + bias) * alpha) < -beta with u = acc * oscale, oscale a power of two (u exact).  The
    // left side is monotone in u, so per channel the test IS a comparison of u with one fp32 threshold theta — found once per
    // workgroup by bisection over the ordered fp32 values WITH the epilogue's own arithmetic (exact by construction, NaN and the
    // infinities included), rescaled per tile by the exact 1 / oscale: one compare per value instead of mul, add, mul, compare.
    //   alpha > 0: bit <=> u < theta;   alpha < 0: bit <=> u > theta' <=> -u < -theta' (e_sg flips the sign of u, e_th = -theta');
    //   alpha == 0: the left side is +-0 for every finite u — the bit is the constant (0 < -beta): theta = +-inf;  alpha NaN: never
    float e_th[3];
    unsigned e_sg[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int n = (nt0 + t) * 32 + lrow;
        const bool nin = n < a.Cout;
        e_co[t] = nin ? n * 4 : OOB;
        e_bv[t] = (a.bias && nin) ? a.bias[n] : 0.0f;
        e_th[t] = 0.0f;
        e_sg[t] = 0u;
        if constexpr (BITS) {
            const float al = nin ? a.alpha[n] : 0.0f, nbe = nin ? -a.beta[n] : 0.0f, bv = e_bv[t];
            auto key2f = [](unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); };
            auto pred = [&](unsigned k) { const float u = key2f(k); const float v = u + bv; return v * al < nbe; };
            const unsigned klo = 0x007fffffu, khi = 0xff800000u;          // keys of -inf, +inf
            if (!(al > 0.0f) && !(al < 0.0f)) {
                // alpha == 0: +-0 < -beta for every finite u;  alpha NaN: never
                e_th[t] = (al == 0.0f && 0.0f < nbe) ? __uint_as_float(0x7f800000u) : __uint_as_float(0xff800000u);
            } else if (al > 0.0f) {                                     // first key whose bit is 0 (the bit of +inf is 0)
                unsigned lo = klo, hi = khi;
                while (lo < hi) {
                    const unsigned mid = lo + ((hi - lo) >> 1);
                    if (!pred(mid)) hi = mid; else lo = mid + 1;
                }
                e_th[t] = key2f(lo);
            } else {                                                    // first key whose bit is 1 (the bit of -inf is 0); none: never
                e_sg[t] = 0x80000000u;
                if (!pred(khi)) {
                    e_th[t] = __uint_as_float(0xff800000u);            // -u < -inf: never
                } else {
                    unsigned lo = klo, hi = khi;
                    while (lo < hi) {
                        const unsigned mid = lo + ((hi - lo) >> 1);
                        if (pred(mid)) hi = mid; else lo = mid + 1;
                    }
                    e_th[t] = -key2f(lo - 1);                           // theta' = the last value whose bit is 0
                }
            }
        }
    }
    const float wsc = a.wscale_dev ? a.wscale * *a.wscale_dev : a.wscale;
    // rotated tile loop — every phase exists ONCE in the code (the unrolled patch passes and the unrolled epilogue are large; inlined
    // twice, with both epilogues, the kernel was ~70 KB of instructions and ran out of the instruction cache: every phase, loads or
    // not, took 4-8x its instruction count):   [request patch of `load_tile`]  [MFMA loop + epilogue of `tile`]  [patch -> LDS]
    int load_tile = blockIdx.x, tile = -1, buf = 0;
    while (true) {
        const bool more = load_tile < ntiles;                  // uniform
        if (more) issue_patch(load_tile);                      // in flight across the MFMA loop below
        QT_FS(0);
        if (tile >= 0) {

        // LDS byte address of this lane's pixel runs in the two planes (per tile)
        const unsigned char* ahb[2];
        const unsigned char* alb[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            ahb[mt] = smem + buf * buf_bytes + abase[mt];
            alb[mt] = ahb[mt] + plane_bytes;
        }
        // A fragments of one k-step: the lane half's two groups at the table's offsets (k2 = ktab[2 s + half])
        auto load_a = [&](const int2 k2, uint4 (&ah)[2], uint4 (&al)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const uint2 h0 = *reinterpret_cast<const uint2*>(ahb[mt] + k2.x);
                const uint2 h1 = *reinterpret_cast<const uint2*>(ahb[mt] + k2.y);
                const uint2 l0 = *reinterpret_cast<const uint2*>(alb[mt] + k2.x);
                const uint2 l1 = *reinterpret_cast<const uint2*>(alb[mt] + k2.y);
                ah[mt] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                al[mt] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        };
        const int2* ktl = ktab + half;
        auto koffs = [&](int s) __attribute__((always_inline)) { return ktl[2 * min(s, a.NKS - 1)]; };
        v16f acc[2][3];
        // software pipeline: the weight fragments of k-step s + WD (L2 latency ~ 2-3 k-steps of MFMA time), the patch fragments of
        // k-step s + 1 (LDS latency) and the offset pair of k-step s + 2 are requested before the MFMAs of k-step s issue; rings
        // indexed by the k-step's parity (a compile-time constant at every call site)
        constexpr int WD = 1, RING = 2;
        uint4 wh[RING][3], wl[RING][3], ah[2][2], al[2][2];
#pragma unroll
        for (int u = 0; u < WD; ++u) load_w(u, wh[u], wl[u]);
        load_a(koffs(0), ah[0], al[0]);
        int2 kq = koffs(1);
        QT_FS(1);
        // FIRST: the k-step that starts the tile — its MFMAs take a zero C operand instead of 96 zeroed accumulator registers
        auto kstep = [&](int s, auto uc, auto firstc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            constexpr bool FIRST = decltype(firstc)::value;
            load_w(s + WD, wh[(u + WD) % RING], wl[(u + WD) % RING]);
            load_a(kq, ah[(u + 1) & 1], al[(u + 1) & 1]);
            kq = koffs(s + 2);
            __builtin_amdgcn_sched_barrier(0);               // the requests go out BEFORE this k-step's MFMAs, not after them
            // one term at a time over the six accumulator tiles: two MFMAs on the same accumulator are never adjacent
            v16f zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt][t] = mfma16(ah[u & 1][mt], wh[u % RING][t], FIRST ? zero : acc[mt][t]);
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
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        // k-step 0, then pairs (the ring positions alternate), an even count's last k-step peeled: no zero k-steps appended
        kstep(0, P0{}, std::true_type{});
        int s0 = 1;
        for (; s0 + 1 < a.NKS; s0 += 2) {
            kstep(s0, P1{}, std::false_type{});
            kstep(s0 + 1, P0{}, std::false_type{});
        }
        if (s0 < a.NKS) kstep(s0, P1{}, std::false_type{});

        QT_FS(2);
        // ---- epilogue: lane owns channel n = tile * 32 + lrow, rows (r & 3) + 8 (r >> 2) + 4 half of each 32-pixel tile ------------
        // (the lane's coordinates are made opaque per tile: everything below that depends only on them — 32 pixel offsets, the
        // channel constants — is loop-invariant, and hipcc otherwise hoists it out of the tile loop into 200 spilled registers)
        int lane_t = lane, half_t = half, lrow_t = lrow;
        asm volatile("" : "+v"(lane_t), "+v"(half_t), "+v"(lrow_t));
        int img, oy0, ox0;
        tile_origin(tile, img, oy0, ox0);
        const float sx = red[buf * 8 + 4];
        const float oscale = sx * wsc;
        [[maybe_unused]] const float inv_os = 1.0f / oscale;
        [[maybe_unused]] __amdgpu_buffer_rsrc_t yrsrc;
        if constexpr (!BITS)
            yrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.y) + (int64_t)img * a.Ho * a.Wo * a.ldy, 0,
                                                      (int)((int64_t)a.Ho * a.Wo * a.ldy * 4), 0x00020000);
        // threshold bits: lanes 0 .. 31 store one pixel's word per (m-tile, channel tile) — the pixel's row of the bit plane is the same
        // for the three channel tiles: found once per m-tile (it was recomputed per store: 25 of the epilogue's ~95 VALU instructions per
        // accumulator tile)
        [[maybe_unused]] uint32_t* brow[2] = {nullptr, nullptr};
        if constexpr (BITS) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int j = (mg * 2 + mt) * 32 + lane_t;
                const int oyl = divm(j, a.m_tox, a.TOX), oxl = j - oyl * a.TOX;
                const int oy = oy0 + oyl, ox = ox0 + oxl;
                if (lane_t < 32 && j < npix && oy < a.Ho && ox < a.Wo) brow[mt] = a.bits + (((int64_t)img * a.Ho + oy) * a.Wo + ox) * a.ldb;
            }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if ((nt0 + t) * 32 >= a.Coutp) continue;
            const float bv = e_bv[t];
            [[maybe_unused]] const float thr = e_th[t] * inv_os;     // exact (power of two) unless it leaves the normal range
            [[maybe_unused]] const unsigned sg = e_sg[t];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int jb = (mg * 2 + mt) * 32;
                if constexpr (BITS) {
                    uint32_t myword = 0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        unsigned long long m[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            m[q] = __ballot(__uint_as_float(__float_as_uint(acc[mt][t][4 * g + q]) ^ sg) < thr);    // channels >= Cout: 0
                        // v_writelane: the two halves of each (scalar) ballot straight into lanes R and R + 4, R = q + 8 g
                        // (gfx950 does not interlock a VALU-written SGPR read by the next VALU and the hazard recogniser does not look
                        // inside inline asm: the s_nop covers the last v_cmp -> first write; the writes are chained through myword)
                        asm("s_nop 1\n\tv_writelane_b32 %0, %1, %9\n\tv_writelane_b32 %0, %2, %9+4\n\t"
                            "v_writelane_b32 %0, %3, %9+1\n\tv_writelane_b32 %0, %4, %9+5\n\t"
                            "v_writelane_b32 %0, %5, %9+2\n\tv_writelane_b32 %0, %6, %9+6\n\t"
                            "v_writelane_b32 %0, %7, %9+3\n\tv_writelane_b32 %0, %8, %9+7"
                            : "+v"(myword)
                            : "s"((uint32_t)m[0]), "s"((uint32_t)(m[0] >> 32)), "s"((uint32_t)m[1]), "s"((uint32_t)(m[1] >> 32)),
                              "s"((uint32_t)m[2]), "s"((uint32_t)(m[2] >> 32)), "s"((uint32_t)m[3]), "s"((uint32_t)(m[3] >> 32)),
                              "n"(8 * g));
                    }
                    if (uint32_t* row = brow[mt]) {                                     // lanes 0 .. 31: one pixel each
                        row[nt0 + t] = myword;
                        // the row's pad words (ldb rounds ceil(Cout / 32) up): written once, by the tile that holds the last channels
                        if ((nt0 + t + 1) * 32 >= a.Coutp)
                            for (int wc = nt0 + t + 1; wc < a.ldb; ++wc) row[wc] = 0u;
                    }
                } else {
                    // buffer stores: voffset = pixel offset (LDS table, four consecutive pixels per 16-byte read) + this lane's channel;
                    // pixels past the map / channels past Cout carry OOB and are dropped by the bounds check of the buffer
                    // (no per-value index arithmetic, no branches)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int4 o4 = *reinterpret_cast<const int4*>(otab + buf * 128 + jb + 8 * g + 4 * half_t);
                        const int oq[4] = {o4.x, o4.y, o4.z, o4.w};
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[mt][t][4 * g + q] * oscale + bv), yrsrc,
                                                                  oq[q] + e_co[t], 0, 0);
                    }
                }
            }
        }
        }
        QT_FS(3);
        if (!more) break;
        const int nb = tile >= 0 ? (buf ^ 1) : 0;
        finish_patch(nb, load_tile);
        tile = load_tile;
        load_tile += gridDim.x;
        buf = nb;
    }
    QT_FS_FLUSH;
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
    if (!alpha && Ho * Wo * ldy * 4 >= (1ll << 30)) return QT_ERR_UNSUPPORTED;      // one image's output plane is a buffer (32-bit offsets)
    FirstArgs a;
    a.x = x; a.sn = sn; a.sc = sc; a.sh = sh; a.sw = sw;
    a.N = (int)N; a.C = (int)C; a.H = (int)H; a.W = (int)W; a.KH = (int)KH; a.KW = (int)KW; a.S = (int)S; a.PH = (int)PH; a.PW = (int)PW;
    a.Ho = (int)Ho; a.Wo = (int)Wo; a.Cp = (int)Cp;
    a.G4 = (int)((KW * Cp + 3) / 4);
    a.NG4 = (int)(KH * a.G4);
    a.NKS = (a.NG4 + 3) / 4;                          // (the weight pack zero-fills the last k-step)
    // output tile: <= 128 pixels, as square as the map allows, sized to waste the fewest padded pixels
    int best_ty = 1, best_tx = 1;
    double best = 1e30;
    for (int ty = 1; ty <= 128; ++ty)
        for (int tx = 1; tx * ty <= 128; ++tx) {
            if (tx * ty < 64 && tx * ty < Ho * Wo) continue;
            const int64_t pr = (int64_t)(ty - 1) * S + KH, pce = ((int64_t)(tx - 1) * S + KW) * Cp;
            const int64_t rs = (std::max<int64_t>(pce, (int64_t)(tx - 1) * S * Cp + a.G4 * 4) + 3) / 4 * 4;
            if (rs / 2 > 256 || (pr + 256 / (rs / 2) - 1) / (256 / (rs / 2)) > 18) continue;          // <= 18 row passes of the patch loader
            if (2 * (2 * pr * rs * 2) + 64 + 1024 + a.NKS * 16 > 76 * 1024) continue;     // two patch buffers, two workgroups per CU
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
    a.RS = (std::max(a.PCE, (a.TOX - 1) * a.S * a.Cp + a.G4 * 4) + 3) / 4 * 4;
    auto magic = [](int d) { return d > 1 ? (unsigned)((1ull << 32) / (unsigned)d + 1) : 0u; };
    a.m_tx = magic(a.tiles_x);
    { const unsigned long long d = (unsigned long long)a.tiles_y * a.tiles_x; a.m64_tpi = d > 1 ? ~0ull / d + 1 : 0; }
    a.m_ppr = magic(a.RS / 2); a.m_cp = magic(a.Cp); a.m_tox = magic(a.TOX); a.m_g4 = magic(a.G4);
    a.whi = reinterpret_cast<const uint4*>(whi);
    a.wlo = reinterpret_cast<const uint4*>(wlo);
    a.wscale = wscale;
    a.wscale_dev = wscale_dev;
    a.Cout = (int)Cout; a.Coutp = (int)Coutp;
    a.step_r = 256 / (a.RS / 2); a.step_pc = 0;          // rows of the patch one pass of the loader covers
    // float2 spans: dense channels-last rows, no channel padding, every patch row starts on an even element of an 8-byte aligned row
    a.rows_dense = (sc == 1 && sw == C && Cp == C && !(sh & 1) && !(sn & 1) && !((a.TOX * S * C) & 1) && !((PW * C) & 1) && H * sh * 4 < (1ll << 30) &&
                    (reinterpret_cast<uintptr_t>(x) & 7) == 0) ? 1 : 0;
    a.bias = bias; a.y = y; a.ldy = ldy; a.alpha = alpha; a.beta = beta; a.bits = bits; a.ldb = ldb;
    const int64_t ntiles = N * a.tiles_y * a.tiles_x;
    if (ntiles > INT32_MAX) return QT_ERR_UNSUPPORTED;
    const int lds = 2 * (2 * a.PR * a.RS * 2) + 64 + 1024 + a.NKS * 16;   // two patch buffers (hi + lo planes each) + the scales + pixel offsets + k-step offsets
    const unsigned ny = (unsigned)((Coutp + 191) / 192);
    // persistent workgroups: two per CU (256 CUs), shared between the channel blocks
    const dim3 grid((unsigned)std::min<int64_t>(ntiles, std::max<int64_t>(1, 512 / ny)), ny);
#define QT_FIRST_LAUNCH(REALW, BITS, MAXP)                                                                                          \
    do {                                                                                                                      \
        static QtLdsOnce once;                                                                                                \
        if (qt_ensure_dyn_lds(once, reinterpret_cast<const void*>(conv_first_direct_kernel<REALW, BITS, MAXP>), lds) != QT_OK) \
            return QT_ERR_LAUNCH;                                                                                             \
        hipLaunchKernelGGL((conv_first_direct_kernel<REALW, BITS, MAXP>), grid, dim3(256), lds, (hipStream_t)stream, a);            \
    } while (0)
    if (wlo) { if (alpha) QT_FIRST_LAUNCH(true, true, 18); else QT_FIRST_LAUNCH(true, false, 18); }
    else { if (alpha) QT_FIRST_LAUNCH(false, true, 18); else QT_FIRST_LAUNCH(false, false, 18); }
#undef QT_FIRST_LAUNCH
    return qt_check_launch();
}

}  // namespace

extern "C" {

#ifdef QT_PROFILING_VARIANTS
int qt_first_stamps_fetch(unsigned long long* host40, int reset) {
    if (hipMemcpyFromSymbol(host40, HIP_SYMBOL(qt_first_stamps), 40 * sizeof(unsigned long long)) != hipSuccess) return QT_ERR_LAUNCH;
    if (reset) {
        unsigned long long z[40] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(qt_first_stamps), z, sizeof(z)) != hipSuccess) return QT_ERR_LAUNCH;
    }
    return QT_OK;
}
#endif

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
