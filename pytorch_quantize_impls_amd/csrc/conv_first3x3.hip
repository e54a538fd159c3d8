// Stride-1 3 x 3 / padding-1 first layer on a real-valued fp32 image with <= 4 channels and 64 output channels (VGG-16's conv1_1:
// 3 -> 64 at 224 x 224 — TerConv2d / BinConv2d, layers/terner_layers.py:89-92, binary_layers.py:103-106), in ONE pass over the image.
//
// Round 5's route for this layer was three passes: a pack pass that wrote the image as a padded fp16-pair pixel plane (91 us for
// 256 x 3 x 224 x 224), the scale's fold / repack launches, and the direct 3 x 3 kernel on that plane (231 us) — 23 % of the C5
// forward together with conv1_2, against ~90 us of HBM time (154 MB in, 411 MB of nibble plane out).  Here a workgroup owns a
// 32 x 32-pixel output tile:
//   * the 34 x 34-pixel patch of the fp32 image is read where it lies (any strides), its max|x| folded on the way, and goes to
//     LDS as two fp16 planes hi = fp16(x / s), lo = fp16(x / s - hi) with the TILE's own power-of-two s (max|x| / s in
//     [2^14, 2^15)): |x - s (hi + lo)| <= max(2^-22 |x|, 2^-39 tilemax) — no global max|x| pass, no operand plane in HBM;
//   * pixels are 4 fp16 apart (channel slot 3 is zero), so for one kernel row ky the K = (kx, c) run of an output pixel is 12
//     contiguous halves of its patch row: one v_mfma_f32_32x32x16_f16 k-step per ky (12 real K of 16) — 3 k-steps x 2 terms;
//   * the WEIGHTS are the MFMA's row operand (32 output channels per instruction, two instructions for the 64), the pixels its
//     column operand: a lane then holds 2 x 16 accumulators = 32 channels of ONE pixel, and the rows are assigned to channels so
//     that those are 32 consecutive channels — the lane packs its pixel's nibbles / bits in registers and stores 16 contiguous
//     bytes (a wave: 1 KiB of the output plane) with no cross-lane traffic;
//   * a wave walks 8 output rows of the tile; each new row needs ONE new patch row's fragments (the other two are kept from the
//     previous rows), i.e. 4 LDS reads for 12 MFMAs;
//   * +-1 / 0 weights are exact in fp16 (one fragment serves hi and lo); channels whose folded BatchNorm slope is negative get
//     their weights negated when the fragments are loaded, so that the threshold test is  u < theta  for every channel.
// Epilogues (ONE accumulation for all three, so the module-by-module fp32 result and the fused chain's signs come from the same
// bits): 0 = fp32 NHWC (+ bias); 1 = BatchNorm-threshold bits; 2 = the next conv's fp4 nibble plane with a 1-pixel zero halo.
// The threshold of a channel is found once per workgroup by bisection over the ordered fp32 values with the float epilogue's own
// arithmetic ((u + bias) * alpha < -beta: exactly the predicate of qt_conv2d_implicit_bits / conv_first_direct).
#include <cstdlib>
#include <type_traits>
#include "qt_common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

constexpr int F3_T = 32;                      // output tile edge
constexpr int F3_P = F3_T + 2;                // patch edge (pixels)
constexpr int F3_PITCH = (F3_P + 2) * 4;      // halves per patch row: 4 per pixel, two pixels of slack (the k = 12..15 run of the last column)
constexpr int F3_PLANE = F3_P * F3_PITCH;     // halves per plane
constexpr int F3_NPIX = F3_P * F3_P;          // 1156 patch pixels
constexpr int F3_PPT = (F3_NPIX + 255) / 256; // patch pixels per thread (5)

struct F3Args {
    const float* x;
    int64_t sn, sc, sh, sw;
    int N, C, H, W;
    const uint4* wfrag;                       // [3 ky][2 T][64 lanes] 16-byte A fragments of one 64-channel group
    const float* bias;
    const float* alpha;
    const float* beta;
    void* out;
    int64_t ldo;                              // mode 0: floats per pixel; 1 / 2: 32-bit words per pixel
    int mode, tiles_y, tiles_x;
};

__device__ __forceinline__ v16f f3_mfma(const uint4& a, const uint4& b, v16f c) {
    h8 av, bv;
    __builtin_memcpy(&av, &a, 16);
    __builtin_memcpy(&bv, &b, 16);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
}

// channel of accumulator register r of channel tile T in lane half h (see the header: rows are assigned so that this is 32 h + 16 T + r)
__device__ __forceinline__ int f3_channel(int h, int T, int r) { return 32 * h + 16 * T + r; }

// NC = channels a patch pixel keeps in registers: 3 (C <= 3: VGG) or 4
// OCC = workgroups per CU the register budget is sized for: 2 = the patch of the next tile in flight across the MFMA rows and two
// accumulator sets (intra-wave overlap), 4 = neither (128 registers: four waves per SIMD hide the latencies instead)
template <int MODE, int NC, int OCC>
__global__ __launch_bounds__(256, OCC) void first3x3_kernel(F3Args a) {
    __shared__ __attribute__((aligned(16))) _Float16 plane[2][F3_PLANE];     // hi | lo
    __shared__ float red[8];
    __shared__ float thr_s[64];
    __shared__ unsigned flip_s[64];
    __shared__ __attribute__((aligned(16))) float tht_s[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;

    // ---- once per workgroup: thresholds (modes 1, 2), A fragments ------------------------------------------------------------------
    if (MODE != 0) {
        if (tid < 64) {
            const float al = a.alpha[tid], nbe = -a.beta[tid], bv = a.bias ? a.bias[tid] : 0.0f;
            // ordered keys of the fp32 values: key2f is increasing in the key
            auto key2f = [](unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); };
            auto pred = [&](unsigned k) { const float u = key2f(k); const float v = u + bv; return v * al < nbe; };
            const unsigned klo = 0x007fffffu, khi = 0xff800000u;          // keys of -inf, +inf
            float th;
            unsigned fl = 0u;
            if (!(al > 0.0f) && !(al < 0.0f)) {
                // alpha == 0: the left side is +-0 for every finite u, the bit is the constant (0 < -beta); alpha NaN: never
                th = (al == 0.0f && 0.0f < nbe) ? __uint_as_float(0x7f800000u) : __uint_as_float(0xff800000u);
            } else if (al > 0.0f) {                                       // bit <=> u < theta: theta = the first value whose bit is 0
                unsigned lo = klo, hi = khi;
                while (lo < hi) {
                    const unsigned mid = lo + ((hi - lo) >> 1);
                    if (!pred(mid)) hi = mid; else lo = mid + 1;
                }
                th = key2f(lo);
            } else {                                                      // bit <=> u > theta' <=> -u < -theta': the weights are negated
                fl = 1u;
                if (!pred(khi)) {
                    th = __uint_as_float(0xff800000u);                   // never
                } else {
                    unsigned lo = klo, hi = khi;
                    while (lo < hi) {
                        const unsigned mid = lo + ((hi - lo) >> 1);
                        if (pred(mid)) hi = mid; else lo = mid + 1;
                    }
                    th = -key2f(lo - 1);                                  // theta' = the last value whose bit is 0
                }
            }
            thr_s[tid] = th;
            flip_s[tid] = fl;
        }
    }
    // zero the planes once: channel slots >= C and the slack pixels are never written again
    for (int i = tid; i < 2 * F3_PLANE / 8; i += 256) reinterpret_cast<uint4*>(&plane[0][0])[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    uint4 wA[3][2];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            uint4 f = a.wfrag[(ky * 2 + T) * 64 + lane];
            if (MODE != 0) {
                // this lane's fragment is MFMA row i = lane % 32 of tile T, i.e. channel 32 hh + 16 T + rr with hh = (i / 4) % 2,
                // rr = 4 (i / 8) + i % 4
                const int i = j, hh = (i >> 2) & 1, rr = 4 * (i >> 3) + (i & 3);
                if (flip_s[f3_channel(hh, T, rr)]) { f.x ^= 0x80008000u; f.y ^= 0x80008000u; f.z ^= 0x80008000u; f.w ^= 0x80008000u; }
            }
            wA[ky][T] = f;
        }
    [[maybe_unused]] float bia[2][16];
    if (MODE == 0) {
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int r = 0; r < 16; ++r) bia[T][r] = a.bias ? a.bias[f3_channel(h, T, r)] : 0.0f;
    }

    // smallest non-zero and largest finite |theta| over the 64 channels (the same in every lane: `fast` below is workgroup-uniform)
    [[maybe_unused]] float tsafe_lo = 3.402823466e38f, tsafe_hi = 0.0f;
    if (MODE != 0) {
        for (int c = 0; c < 64; ++c) {
            const float t = fabsf(thr_s[c]);
            if (t > 0.0f && t < tsafe_lo) tsafe_lo = t;                 // (inf counts: inf / s stays inf)
            if (t <= 3.402823466e38f && t > tsafe_hi) tsafe_hi = t;
        }
    }
    const int tpi = a.tiles_y * a.tiles_x;
    const int ntiles = a.N * tpi;
    // per-lane constants of the fragment reads: byte offset of this lane's run within a patch row
    const int boff = (j + 2 * h) * 8;
    // the patch in flight: requested for tile t + 1 before the MFMA rows of tile t start (the loads' HBM latency sits under them)
    float v[F3_PPT][NC];
    // 32-bit element offsets within an image (the launcher checked that an image spans < 2^31 elements)
    const int sc1 = (int)a.sc, sc2 = 2 * (int)a.sc;
    [[maybe_unused]] const int sc3 = 3 * (int)a.sc;
    auto issue_patch = [&](int tile_) __attribute__((always_inline)) {
        const int img_ = tile_ / tpi, trem_ = tile_ - img_ * tpi;
        const int ty_ = trem_ / a.tiles_x, tx_ = trem_ - ty_ * a.tiles_x;
        const int yb = ty_ * F3_T, xb = tx_ * F3_T;
        const float* xi = a.x + (int64_t)img_ * a.sn + ((int64_t)yb * a.sh + (int64_t)xb * a.sw);      // uniform
#pragma unroll
        for (int i = 0; i < F3_PPT; ++i) {
            const int pi = tid + 256 * i, pr_ = pi / F3_P, pp_ = pi - pr_ * F3_P;
            const int iy = yb - 1 + pr_, ix = xb - 1 + pp_;
            const bool ok = (i < F3_PPT - 1 || pi < F3_NPIX) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const float* p = xi + (ok ? (pr_ - 1) * (int)a.sh + (pp_ - 1) * (int)a.sw : 0);
            // (a pixel outside the image reads the tile's own origin pixel — inside the image — and is zeroed)
            const float f0 = p[0], f1 = a.C > 1 ? p[sc1] : 0.0f, f2 = a.C > 2 ? p[sc2] : 0.0f;
            v[i][0] = ok ? f0 : 0.0f;
            v[i][1] = ok ? f1 : 0.0f;
            v[i][2] = ok ? f2 : 0.0f;
            if constexpr (NC == 4) {
                const float f3 = a.C > 3 ? p[sc3] : 0.0f;
                v[i][3] = ok ? f3 : 0.0f;
            }
        }
    };
    if (OCC == 2 && (int)blockIdx.x < ntiles) issue_patch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (OCC != 2) issue_patch(tile);
        const int img = tile / tpi, trem = tile - img * tpi;
        const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
        const int y0 = ty * F3_T, x0 = tx * F3_T;
        // ---- patch: max|x| of the registers ----------------------------------------------------------------------------------------
        unsigned mx = 0;
#pragma unroll
        for (int i = 0; i < F3_PPT; ++i)
#pragma unroll
            for (int c = 0; c < NC; ++c) mx = max(mx, __float_as_uint(v[i][c]) & 0x7fffffffu);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        __syncthreads();                               // every wave is past the previous tile's fragment reads (and red[] reads)
        if (lane == 0) red[wave] = __uint_as_float(mx);
        __syncthreads();
        // s = 2^(e - 14), e the exponent of max|x|: max|x| / s in [2^14, 2^15); max|x| == 0 / subnormal, inf, NaN: s = 1 (an inf / NaN
        // pixel then poisons the outputs whose window holds it through fp16 inf / NaN, as it does in the reference's fp32 conv)
        const unsigned m = max(max(__float_as_uint(red[0]), __float_as_uint(red[1])), max(__float_as_uint(red[2]), __float_as_uint(red[3])));
        const int eb = (int)(m >> 23);
        float sc = 1.0f;
        if (eb > 0 && eb < 255) {
            int se = eb - 14;
            se = se < 1 ? 1 : (se > 254 ? 254 : se);
            sc = __uint_as_float((unsigned)se << 23);
        }
        const float inv = 1.0f / sc;                   // exact: a power of two within the normal range
#pragma unroll
        for (int i = 0; i < F3_PPT; ++i) {
            const int pi = tid + 256 * i;
            if (i < F3_PPT - 1 || pi < F3_NPIX) {
                const int pr = pi / F3_P, pp = pi - pr * F3_P;
                h4 hi4, lo4;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float t = (c < NC ? v[i][c < NC ? c : 0] : 0.0f) * inv;
                    const _Float16 hh = (_Float16)t;
                    hi4[c] = hh;
                    lo4[c] = (_Float16)(t - (float)hh);
                }
                *reinterpret_cast<h4*>(&plane[0][pr * F3_PITCH + pp * 4]) = hi4;
                *reinterpret_cast<h4*>(&plane[1][pr * F3_PITCH + pp * 4]) = lo4;
            }
        }

        // ---- this wave's 8 output rows -----------------------------------------------------------------------------------------------
        const unsigned char* ph = reinterpret_cast<const unsigned char*>(&plane[0][0]) + boff;
        const unsigned char* pl = reinterpret_cast<const unsigned char*>(&plane[1][0]) + boff;
        auto load_b = [&](int prow, uint4& bh, uint4& bl) __attribute__((always_inline)) {
            const int o = prow * F3_PITCH * 2;
            const uint2 h0 = *reinterpret_cast<const uint2*>(ph + o), h1 = *reinterpret_cast<const uint2*>(ph + o + 8);
            const uint2 l0 = *reinterpret_cast<const uint2*>(pl + o), l1 = *reinterpret_cast<const uint2*>(pl + o + 8);
            // lane half 1 holds kx = 2 and the weightless kx = 3 run: that run is the NEXT pixel's data — masked, or an inf / NaN
            // there would turn into NaN (0 x inf) in an output whose window does not contain it
            bh = make_uint4(h0.x, h0.y, h ? 0u : h1.x, h ? 0u : h1.y);
            bl = make_uint4(l0.x, l0.y, h ? 0u : l1.x, h ? 0u : l1.y);
        };
        // u = acc * s against theta  <=>  acc against theta / s — one multiply per CHANNEL and tile instead of one per value — as long
        // as theta / s is exact: no finite theta may overflow, no non-zero theta may fall into the subnormals (tsafe_lo / tsafe_hi: the
        // lane's smallest non-zero and largest finite |theta|).  Wave-uniform; the slow form multiplies every accumulator.
        // (the per-tile thresholds go through LDS — 64 floats, read back as broadcast ds_read_b128 in the epilogue — instead of 32
        //  more registers per lane: with the two accumulator sets the kernel is at the 256-register limit of two waves per SIMD)
        [[maybe_unused]] bool fast = false;
        if (MODE != 0) {
            fast = !(tsafe_lo * inv < 1.17549435e-38f) && tsafe_hi * inv <= 3.402823466e38f;
            if (tid < 64) tht_s[tid] = fast ? thr_s[tid] * inv : thr_s[tid];
        }
        __syncthreads();
        if (OCC == 2 && tile + (int)gridDim.x < ntiles) issue_patch(tile + gridDim.x);
        const float* tht = &tht_s[32 * h];
        const int r0 = wave * 8;
        uint4 bh[3], bl[3];
        load_b(r0, bh[0], bl[0]);
        load_b(r0 + 1, bh[1], bl[1]);
        // the MFMAs of row i + 1 are issued BEFORE the epilogue of row i (two accumulator sets): the matrix pipe works on the next
        // row while this wave's VALU packs the current one — with one set the wave alternated 12 MFMAs / ~100 VALU, each waiting
        // for the other
        auto mfma_row = [&](int i, v16f (&acc)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                v16f z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                acc[T] = z;
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int T = 0; T < 2; ++T) acc[T] = f3_mfma(wA[ky][T], bh[(i + ky) % 3], acc[T]);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int T = 0; T < 2; ++T) acc[T] = f3_mfma(wA[ky][T], bl[(i + ky) % 3], acc[T]);
        };
        v16f accs[OCC == 2 ? 2 : 1][2];
        load_b(r0 + 2, bh[2], bl[2]);
        if (OCC == 2) mfma_row(0, accs[0]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OCC == 2) {
                if (i < 7) {
                    load_b(r0 + i + 3, bh[(i + 3) % 3], bl[(i + 3) % 3]);  // (slot of patch row r0 + i: row i's MFMAs are issued)
                    mfma_row(i + 1, accs[(i + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                if (i > 0) load_b(r0 + i + 2, bh[(i + 2) % 3], bl[(i + 2) % 3]);
                mfma_row(i, accs[0]);
            }
            v16f (&acc)[2] = accs[OCC == 2 ? (i & 1) : 0];
            const int y = y0 + r0 + i, x = x0 + j;
            const bool inside = y < a.H && x < a.W;
            if (MODE == 0) {
                if (inside) {
                    float* o = reinterpret_cast<float*>(a.out) + ((int64_t)(img * a.H + y) * a.W + x) * a.ldo + 32 * h;
#pragma unroll
                    for (int T = 0; T < 2; ++T)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float4 f;
                            f.x = acc[T][4 * q + 0] * sc + bia[T][4 * q + 0];
                            f.y = acc[T][4 * q + 1] * sc + bia[T][4 * q + 1];
                            f.z = acc[T][4 * q + 2] * sc + bia[T][4 * q + 2];
                            f.w = acc[T][4 * q + 3] * sc + bia[T][4 * q + 3];
                            *reinterpret_cast<float4*>(o + 16 * T + 4 * q) = f;
                        }
                }
            } else if (MODE == 2) {
                uint32_t w[4];
                auto pack = [&](auto fc) __attribute__((always_inline)) {
                    constexpr bool FAST = decltype(fc)::value;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        uint32_t ww = 0;
#pragma unroll
                        for (int n = 7; n >= 0; --n) {
                            const int T = q >> 1, r = 8 * (q & 1) + n;
                            const float u = FAST ? acc[T][r] : acc[T][r] * sc;      // (acc * s is exact: a power of two)
                            ww = (ww << 4) | (u < tht[16 * T + r] ? 0xAu : 0x2u);         // fp4 -1 : +1
                        }
                        w[q] = ww;
                    }
                };
                if (fast) pack(std::true_type{}); else pack(std::false_type{});
                if (inside) {
                    uint32_t* o = reinterpret_cast<uint32_t*>(a.out) +
                                  ((int64_t)(img * (a.H + 2) + y + 1) * (a.W + 2) + x + 1) * a.ldo + 4 * h;
                    *reinterpret_cast<uint4*>(o) = make_uint4(w[0], w[1], w[2], w[3]);
                }
            } else {
                uint32_t ww = 0;
                auto pack = [&](auto fc) __attribute__((always_inline)) {
                    constexpr bool FAST = decltype(fc)::value;
#pragma unroll
                    for (int b = 31; b >= 0; --b) {
                        const int T = b >> 4, r = b & 15;
                        const float u = FAST ? acc[T][r] : acc[T][r] * sc;
                        ww = (ww << 1) | (u < tht[16 * T + r] ? 1u : 0u);
                    }
                };
                if (fast) pack(std::true_type{}); else pack(std::false_type{});
                const uint32_t other = (uint32_t)__shfl_xor((int)ww, 32);
                if (inside && h == 0) {
                    uint32_t* o = reinterpret_cast<uint32_t*>(a.out) + ((int64_t)(img * a.H + y) * a.W + x) * a.ldo;
                    *reinterpret_cast<uint4*>(o) = make_uint4(ww, other, 0u, 0u);
                }
            }
        }
        // ---- mode 2: the zero halo of the output plane next to this tile ------------------------------------------------------------
        if (MODE == 2) {
            const bool top = y0 == 0, bottom = y0 + F3_T >= a.H, left = x0 == 0, right = x0 + F3_T >= a.W;
            // border pixels this tile owns: rows -1 / H over the tile's columns (one column further out at the image's left / right
            // edge: the corners), columns -1 / W over the tile's rows — 2 x 34 + 2 x 32 candidates, 2 uint4 each
            for (int b = tid; b < 2 * (2 * F3_P + 2 * F3_T); b += 256) {
                const int q = b >> 1, half = b & 1;
                int by, bx;
                bool ok;
                if (q < 2 * F3_P) {
                    const bool up = q < F3_P;
                    by = up ? -1 : a.H;
                    bx = x0 - 1 + (up ? q : q - F3_P);
                    const int xend = min(x0 + F3_T, a.W);
                    ok = (up ? top : bottom) && ((bx >= x0 && bx < xend) || (bx == -1 && left) || (bx == a.W && right));
                } else {
                    const int qq = q - 2 * F3_P;
                    const bool lf = qq < F3_T;
                    bx = lf ? -1 : a.W;
                    by = y0 + (lf ? qq : qq - F3_T);
                    ok = (lf ? left : right) && by < a.H;
                }
                if (ok) {
                    uint32_t* o = reinterpret_cast<uint32_t*>(a.out) +
                                  ((int64_t)(img * (a.H + 2) + by + 1) * (a.W + 2) + bx + 1) * a.ldo + 4 * half;
                    *reinterpret_cast<uint4*>(o) = make_uint4(0, 0, 0, 0);
                }
            }
        }
    }
}

// A fragments from the quantised fp32 weight [64, C, 3, 3] (element strides): fragment (ky, T), lane l: the 8 halves k = 8 (l / 32)
// .. + 7 of MFMA row i = l % 32, k = 4 kx + c; row i of tile T is channel 32 hh + 16 T + rr (hh = (i / 4) % 2, rr = 4 (i / 8) + i % 4)
__global__ void first3x3_pack_kernel(const float* __restrict__ wq, int64_t so, int64_t si, int64_t sh, int64_t sw, int C, int Cout,
                                     uint4* __restrict__ frag) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 3 * 2 * 64) return;
    const int lane = t & 63, T = (t >> 6) & 1, ky = t >> 7;
    const int i = lane & 31, kh = lane >> 5;
    const int hh = (i >> 2) & 1, rr = 4 * (i >> 3) + (i & 3);
    const int ch = 32 * hh + 16 * T + rr;
    unsigned short hv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = 8 * kh + e, kx = k >> 2, c = k & 3;
        float f = 0.0f;
        if (kx < 3 && c < C && ch < Cout) f = wq[(int64_t)ch * so + (int64_t)c * si + (int64_t)ky * sh + (int64_t)kx * sw];
        const _Float16 q = (_Float16)f;
        __builtin_memcpy(&hv[e], &q, 2);
    }
    frag[t] = make_uint4(hv[0] | ((unsigned)hv[1] << 16), hv[2] | ((unsigned)hv[3] << 16), hv[4] | ((unsigned)hv[5] << 16),
                         hv[6] | ((unsigned)hv[7] << 16));
}

}  // namespace

extern "C" {

int qt_conv3x3_first_pack_weight_f32(const float* wq, int64_t stride_o, int64_t stride_i, int64_t stride_h, int64_t stride_w, int64_t C,
                                     int64_t Cout, uint32_t* wfrag, qt_stream_t stream) {
    if (!wq || !wfrag || C < 1 || C > 4 || Cout < 1 || Cout > 64) return QT_ERR_INVALID_ARG;
    if (!qt_aligned16(wfrag)) return QT_ERR_ALIGNMENT;
    hipLaunchKernelGGL(first3x3_pack_kernel, dim3(2), dim3(192), 0, (hipStream_t)stream, wq, stride_o, stride_i, stride_h, stride_w, (int)C,
                       (int)Cout, reinterpret_cast<uint4*>(wfrag));
    return qt_check_launch();
}

int qt_conv3x3_first_f32(const float* x, int64_t stride_n, int64_t stride_c, int64_t stride_h, int64_t stride_w, int64_t N, int64_t C,
                         int64_t H, int64_t W, const uint32_t* wfrag, int64_t Cout, const float* bias, const float* alpha,
                         const float* beta, void* out, int64_t ldo, int mode, qt_stream_t stream) {
    if (!x || !wfrag || !out || N <= 0 || C < 1 || C > 4 || H <= 0 || W <= 0 || mode < 0 || mode > 2) return QT_ERR_INVALID_ARG;
    if (Cout != 64) return QT_ERR_UNSUPPORTED;                      // one 64-channel group (VGG-16's conv1_1); others: the older routes
    if (mode != 0 && (!alpha || !beta)) return QT_ERR_INVALID_ARG;
    if (!qt_aligned16(wfrag) || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    if (mode == 0 ? (ldo < 64 || (ldo & 3)) : (mode == 1 ? ldo != 4 : ldo != 8)) return QT_ERR_ALIGNMENT;
    {
        auto mag = [](int64_t v) { return v < 0 ? -v : v; };
        if ((H + 2) * mag(stride_h) + (W + 2) * mag(stride_w) + 4 * mag(stride_c) >= (1ll << 31)) return QT_ERR_UNSUPPORTED;   // 32-bit offsets
    }
    if (N * (H + 2) >= (1ll << 31) || W + 2 >= (1ll << 30) || N * (H + 2) * (W + 2) * ldo >= (1ll << 40)) return QT_ERR_UNSUPPORTED;
    F3Args a;
    a.x = x; a.sn = stride_n; a.sc = stride_c; a.sh = stride_h; a.sw = stride_w;
    a.N = (int)N; a.C = (int)C; a.H = (int)H; a.W = (int)W;
    a.wfrag = reinterpret_cast<const uint4*>(wfrag);
    a.bias = bias; a.alpha = alpha; a.beta = beta; a.out = out; a.ldo = ldo; a.mode = mode;
    a.tiles_y = (int)((H + F3_T - 1) / F3_T);
    a.tiles_x = (int)((W + F3_T - 1) / F3_T);
    const int64_t ntiles = N * a.tiles_y * a.tiles_x;
    if (ntiles >= (1ll << 31)) return QT_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(ntiles < 512 ? ntiles : 512));      // persistent: two workgroups per CU
    hipStream_t st = (hipStream_t)stream;
    // three workgroups per CU without the register-hungry pipelining beat two with it: 174 vs 212 us inside the C5 graph, batch 256
    // (tools/probes/c5_graph_kernels.py).  QT_F3_OCC=2 (tools only, read once): the pipelined form.
    static const int occ = [] { const char* e = getenv("QT_F3_OCC"); return e ? atoi(e) : 3; }();
    const dim3 grid4((unsigned)(ntiles < 768 ? ntiles : 768));
#define QT_F3(M)                                                                                         \
    do {                                                                                                 \
        if (occ == 3) {                                                                                  \
            if (C <= 3) hipLaunchKernelGGL((first3x3_kernel<M, 3, 3>), grid4, dim3(256), 0, st, a);    \
            else hipLaunchKernelGGL((first3x3_kernel<M, 4, 3>), grid4, dim3(256), 0, st, a);           \
        } else {                                                                                         \
            if (C <= 3) hipLaunchKernelGGL((first3x3_kernel<M, 3, 2>), grid, dim3(256), 0, st, a);     \
            else hipLaunchKernelGGL((first3x3_kernel<M, 4, 2>), grid, dim3(256), 0, st, a);            \
        }                                                                                                \
    } while (0)
    if (mode == 0) QT_F3(0);
    else if (mode == 1) QT_F3(1);
    else QT_F3(2);
#undef QT_F3
    return qt_check_launch();
}

}  // extern "C"
