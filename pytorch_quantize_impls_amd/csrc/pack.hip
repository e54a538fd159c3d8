// Bit-pack kernels: fp32 rows -> uint32 bit planes (include/qt_hip.h "Packed formats").
//
// HBM-bound (reads 4 B/element, writes 1/8 B/element per plane).  Two code paths:
//   * vec  : K % 4 == 0, 16-byte aligned rows.  Each lane loads one float4 (a wave-load is 1 KiB
//            contiguous), turns it into a 4-bit nibble, and the 8 lanes that share a 32-element
//            word OR their shifted nibbles together with three DPP-class lane exchanges; lane
//            (l & 7) == 0 of each group stores the word.  Words past ceil(K/32) up to ldp are
//            written as zero by the same pass (the pad-is-zero invariant of the format).
//   * wave : any K / alignment.  Lane i of a wave loads element i of a 64-element chunk
//            (coalesced dwords) and a 64-bit wave ballot yields two words at once.
#include "qt_common.h"

namespace {

struct SignBits {
    static constexpr int NPLANES = 1;
    __device__ __forceinline__ static void bits(float x, uint32_t& p0, uint32_t& p1) {
        p0 = qt_neg_bit(x);
        p1 = 0;
    }
};
struct TernaryBits {  // plane0 = mask (t != 0), plane1 = sign (t < 0)
    static constexpr int NPLANES = 2;
    __device__ __forceinline__ static void bits(float x, uint32_t& p0, uint32_t& p1) {
        const float t = qt_ternarize(x);
        p0 = (t != 0.0f) ? 1u : 0u;
        p1 = (t < 0.0f) ? 1u : 0u;
    }
};

__device__ __forceinline__ uint32_t or_reduce8(uint32_t v) {
    v |= __shfl_xor(v, 1);
    v |= __shfl_xor(v, 2);
    v |= __shfl_xor(v, 4);
    return v;
}

// One row of packed output = ldp words = ldp*8 float4 "slots" (slots past K/4 load nothing and
// contribute zero bits).  Work item = one slot; 8 consecutive slots = 8 consecutive lanes = 1 word.
// ldp % 4 == 0 guarantees a row's slot count (8*ldp) is a multiple of 32, and 64-lane waves start
// at multiples of 64 slots in the flattened (row, slot) index space, so a word never straddles
// two waves.
template <class Enc, bool WRITE_F32>
__global__ __launch_bounds__(256) void pack_vec_kernel(const float* __restrict__ x, int64_t ldx,
                                                       uint32_t* __restrict__ p0,
                                                       uint32_t* __restrict__ p1, int64_t ldp,
                                                       float* __restrict__ yf, int64_t ldy,
                                                       int64_t rows, int64_t K) {
    const int64_t slots_per_row = ldp * 8;
    const int64_t total = rows * slots_per_row;
    const int64_t k4 = K / 4;  // K % 4 == 0 on this path
    const int lane8 = threadIdx.x & 7;
    // total % 32 == 0 and every 8-lane group starts at a multiple of 8, so a group is either
    // entirely inside the loop or entirely outside it: the cross-lane OR only ever reads lanes
    // that are active with it.
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
         s += (int64_t)gridDim.x * blockDim.x) {
        uint32_t b0 = 0, b1 = 0;
        const bool live = true;
        const int64_t row = s / slots_per_row;
        const int64_t slot = s - row * slots_per_row;
        {
            if (slot < k4) {
                const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + slot * 4);
                uint32_t a0, a1, c0, c1, d0, d1, e0, e1;
                Enc::bits(v.x, a0, a1);
                Enc::bits(v.y, c0, c1);
                Enc::bits(v.z, d0, d1);
                Enc::bits(v.w, e0, e1);
                b0 = a0 | (c0 << 1) | (d0 << 2) | (e0 << 3);
                b1 = a1 | (c1 << 1) | (d1 << 2) | (e1 << 3);
                if (WRITE_F32) {
                    float4 r;
                    r.x = qt_safe_sign(v.x); r.y = qt_safe_sign(v.y);
                    r.z = qt_safe_sign(v.z); r.w = qt_safe_sign(v.w);
                    *reinterpret_cast<float4*>(yf + row * ldy + slot * 4) = r;
                }
            }
        }
        const uint32_t w0 = or_reduce8(b0 << (4 * lane8));
        if (live && lane8 == 0) p0[row * ldp + (slot >> 3)] = w0;
        if (Enc::NPLANES == 2) {
            const uint32_t w1 = or_reduce8(b1 << (4 * lane8));
            if (live && lane8 == 0) p1[row * ldp + (slot >> 3)] = w1;
        }
    }
}

// Generic path: one wave per (row, 64-element chunk); chunks cover the padded row (ldp*32
// elements) so pad words are zeroed as well.
template <class Enc, bool WRITE_F32>
__global__ __launch_bounds__(256) void pack_wave_kernel(const float* __restrict__ x, int64_t ldx,
                                                        uint32_t* __restrict__ p0,
                                                        uint32_t* __restrict__ p1, int64_t ldp,
                                                        float* __restrict__ yf, int64_t ldy,
                                                        int64_t rows, int64_t K) {
    const int lane = threadIdx.x & 63;
    const int64_t chunks_per_row = ldp / 2;  // ldp % 4 == 0
    const int64_t total = rows * chunks_per_row;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t c = wave; c < total; c += nwaves) {
        const int64_t row = c / chunks_per_row;
        const int64_t ch = c - row * chunks_per_row;
        const int64_t k = ch * 64 + lane;
        uint32_t b0 = 0, b1 = 0;
        if (k < K) {
            const float v = x[row * ldx + k];
            Enc::bits(v, b0, b1);
            if (WRITE_F32) yf[row * ldy + k] = qt_safe_sign(v);
        }
        const unsigned long long m0 = __ballot(b0 != 0);
        if (lane < 2) p0[row * ldp + ch * 2 + lane] = (uint32_t)(m0 >> (32 * lane));
        if (Enc::NPLANES == 2) {
            const unsigned long long m1 = __ballot(b1 != 0);
            if (lane < 2) p1[row * ldp + ch * 2 + lane] = (uint32_t)(m1 >> (32 * lane));
        }
    }
}

template <class Enc, bool WRITE_F32>
int launch_pack(const float* x, int64_t ldx, uint32_t* p0, uint32_t* p1, int64_t ldp, float* yf,
                int64_t ldy, int64_t rows, int64_t K, qt_stream_t stream) {
    if (rows < 0 || K < 0 || ldx < K || ldp < 0) return QT_ERR_INVALID_ARG;
    if (rows == 0) return QT_OK;
    if (!x && K > 0) return QT_ERR_INVALID_ARG;
    if (!p0 || (Enc::NPLANES == 2 && !p1)) return QT_ERR_INVALID_ARG;
    if (WRITE_F32 && (!yf || ldy < K)) return QT_ERR_INVALID_ARG;
    const int64_t kw = (K + 31) / 32;
    if (ldp < kw || (ldp & 3) != 0) return QT_ERR_ALIGNMENT;
    if (!qt_aligned16(p0) || (Enc::NPLANES == 2 && !qt_aligned16(p1))) return QT_ERR_ALIGNMENT;
    if (ldp == 0) return QT_OK;
    const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && qt_aligned16(x) &&
                     (!WRITE_F32 || ((ldy % 4 == 0) && qt_aligned16(yf)));
    if (vec) {
        const int64_t total = rows * ldp * 8;
        const int grid = qt_stream_grid((total + 255) / 256);
        hipLaunchKernelGGL((pack_vec_kernel<Enc, WRITE_F32>), dim3(grid), dim3(256), 0,
                           (hipStream_t)stream, x, ldx, p0, p1, ldp, yf, ldy, rows, K);
    } else {
        const int64_t total_waves = rows * (ldp / 2);
        const int grid = qt_stream_grid((total_waves + 3) / 4);
        hipLaunchKernelGGL((pack_wave_kernel<Enc, WRITE_F32>), dim3(grid), dim3(256), 0,
                           (hipStream_t)stream, x, ldx, p0, p1, ldp, yf, ldy, rows, K);
    }
    return qt_check_launch();
}

__global__ __launch_bounds__(256) void check_pm1_kernel(const float* __restrict__ x, int64_t n,
                                                        int32_t* __restrict__ flag) {
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        bad |= !(v == 1.0f || v == -1.0f);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

}  // namespace

extern "C" {

int qt_sign_pack_f32(const float* x, int64_t ldx, uint32_t* sign_plane, int64_t ldp, float* y_f32,
                     int64_t ldy, int64_t rows, int64_t K, qt_stream_t stream) {
    if (y_f32)
        return launch_pack<SignBits, true>(x, ldx, sign_plane, nullptr, ldp, y_f32, ldy, rows, K,
                                           stream);
    return launch_pack<SignBits, false>(x, ldx, sign_plane, nullptr, ldp, nullptr, 0, rows, K,
                                        stream);
}

int qt_ternary_pack_f32(const float* x, int64_t ldx, uint32_t* mask_plane, uint32_t* sign_plane,
                        int64_t ldp, int64_t rows, int64_t K, qt_stream_t stream) {
    return launch_pack<TernaryBits, false>(x, ldx, mask_plane, sign_plane, ldp, nullptr, 0, rows, K,
                                           stream);
}

int qt_check_pm1_f32(const float* x, int64_t n, int32_t* flag, qt_stream_t stream) {
    if (n < 0 || !flag || (n > 0 && !x)) return QT_ERR_INVALID_ARG;
    if (n == 0) return QT_OK;
    const int grid = qt_stream_grid((n + 2047) / 2048);
    hipLaunchKernelGGL(check_pm1_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, flag);
    return qt_check_launch();
}

}  // extern "C"
