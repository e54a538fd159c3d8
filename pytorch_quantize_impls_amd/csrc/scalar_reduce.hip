// Tensor -> device scalar reductions of the weight quantisers, one launch each (round 6: the training step's launch diet).
//
//   qt_abs_mean_f32 : E = mean|W| of DoReFa's 1-bit weight sign(W) * E (functions/dorefa_connect.py:100, `torch.mean(torch.abs(x))`).
//                     torch runs it as abs + mean (+ a buffer fill for tensors beyond one reduction block): three launches per
//                     layer and step, 2 x 4 bytes per weight of extra traffic.  Here: every workgroup sums |w| over its
//                     grid-stride share (fp32 lanes, folded in double), writes its partial, and the LAST workgroup to arrive
//                     (ticket counter, release / acquire at agent scope) adds the partials IN INDEX ORDER in double and writes
//                     float(sum / n): the value does not depend on which workgroup is last.  The counter resets itself, so the
//                     caller's `work` buffer (qt_abs_mean_work_words() uint32, zero-initialised ONCE) serves every later call on
//                     the same stream.
#include "qt_common.h"

namespace {

constexpr int AM_MAX_BLOCKS = 256;

__global__ __launch_bounds__(256) void abs_mean_kernel(const float* __restrict__ x, int64_t n, uint32_t* __restrict__ work,
                                                       float* __restrict__ out) {
    double* part = reinterpret_cast<double*>(work + 4);          // work[0] = ticket counter; partials behind it (8-byte aligned)
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    const int64_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = x4[i];
        s0 += fabsf(v.x); s1 += fabsf(v.y); s2 += fabsf(v.z); s3 += fabsf(v.w);
    }
    double s = ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) s += (double)fabsf(x[(n4 << 2) + threadIdx.x]);
    // workgroup sum in a fixed order: lanes by xor-shuffle (a balanced tree), then the four waves in index order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    __shared__ double sh[4];
    __shared__ bool last;
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double b = (sh[0] + sh[1]) + (sh[2] + sh[3]);
        __hip_atomic_store(&part[blockIdx.x], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __atomic_thread_fence(__ATOMIC_RELEASE);                                      // the partial is visible before the ticket
        const unsigned t = __hip_atomic_fetch_add(&work[0], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (last) {
        // the fold of the partials: thread t adds part[t], part[t + 256] (index order), then the same fixed tree as above — the
        // value depends on the partials only, not on which workgroup happens to be the last (a serial loop of agent-scope loads in
        // one lane was 20 us for 512 partials)
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        double tot = 0.0;
        for (unsigned b = threadIdx.x; b < gridDim.x; b += 256)
            tot += __hip_atomic_load(&part[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = tot;
        __syncthreads();
        if (threadIdx.x == 0) {
            *out = (float)(((sh[0] + sh[1]) + (sh[2] + sh[3])) / (double)n);
            __hip_atomic_store(&work[0], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next call on this stream
        }
    }
}

}  // namespace

extern "C" int64_t qt_abs_mean_work_words(void) { return 4 + 2 * AM_MAX_BLOCKS; }

extern "C" int qt_abs_mean_f32(const float* x, int64_t n, uint32_t* work, float* out, qt_stream_t stream) {
    if (n <= 0 || !x || !work || !out) return QT_ERR_INVALID_ARG;
    if (!qt_aligned16(x) || (reinterpret_cast<uintptr_t>(work) & 7u)) return QT_ERR_ALIGNMENT;
    int64_t blocks = ((n >> 2) + 1023) / 1024;                   // >= 4 float4 per lane before another workgroup pays
    if (blocks < 1) blocks = 1;
    if (blocks > AM_MAX_BLOCKS) blocks = AM_MAX_BLOCKS;
    hipLaunchKernelGGL(abs_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, work, out);
    return qt_check_launch();
}
