// Issue cost of the matrix-core instructions a "real-valued operand x +-1 operand" product could be built from (gfx950):
// v_mfma_scale_f32_32x32x64_f8f6f4 with fp4 / fp6 (e2m3) / fp8 (e4m3) A operands against an fp4 B operand, the fp16 and int8
// 32x32 forms.  One workgroup per CU, 4 waves (one per SIMD), 8 independent accumulator tiles per wave, no memory traffic in the
// loop: shader-clock cycles per instruction and the rate in MACs per cycle and CU.  Planning data for DESIGN.md section 8
// ("digits on the f8f6f4 cores"), not part of the library.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_formats.hip -o mfma_formats ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int FMT_A, int FMT_B>
__device__ __forceinline__ v16f mfma_f8f6f4(v8i a, v8i b, v16f c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FMT_A, FMT_B, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// KIND: 0 = f8f6f4 (FMT_A, FMT_B), 1 = f16 32x32x16, 2 = i8 32x32x32
template <int KIND, int FMT_A, int FMT_B>
__global__ __launch_bounds__(256) void k(const v8i* __restrict__ data, float* out, int iters, unsigned long long* ts) {
    const int t = threadIdx.x;
    v8i a[2], b[4];
    for (int i = 0; i < 2; ++i) a[i] = data[i * 256 + t];
    for (int i = 0; i < 4; ++i) b[i] = data[(2 + i) * 256 + t];
    v16f c[2][4];
    v16i ci[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) { c[i][j][r] = 0.f; ci[i][j][r] = 0; }
    const unsigned long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (KIND == 0) c[i][j] = mfma_f8f6f4<FMT_A, FMT_B>(a[i], b[j], c[i][j]);
                    else if constexpr (KIND == 1) {
                        h8 av, bv;
                        __builtin_memcpy(&av, &a[i], 16);
                        __builtin_memcpy(&bv, &b[j], 16);
                        c[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c[i][j], 0, 0, 0);
                    } else {
                        v4i av = {a[i][0], a[i][1], a[i][2], a[i][3]}, bv = {b[j][0], b[j][1], b[j][2], b[j][3]};
                        ci[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, ci[i][j], 0, 0, 0);
                    }
                }
    }
    const unsigned long long c1 = clock64();
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 16; ++r) s += c[i][j][r] + (float)ci[i][j][r];
    out[blockIdx.x * 256 + t] = s;
    if (t == 0) ts[blockIdx.x] = c1 - c0;
}

template <int KIND, int FA, int FB>
void run(const char* name, int kdim, const v8i* d, float* out, unsigned long long* ts, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<KIND, FA, FB>), dim3(blocks), dim3(256), 0, 0, d, out, 50, ts);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<KIND, FA, FB>), dim3(blocks), dim3(256), 0, 0, d, out, iters, ts);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks);
    CK(hipMemcpy(h.data(), ts, blocks * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : h) avg += (double)v;
    avg /= blocks;
    const double per = avg / (iters * 32.0);                 // cycles per MFMA of one wave (= one SIMD)
    const double macs = 32.0 * 32.0 * kdim / per * 4.0;      // MACs per cycle and CU (4 SIMDs)
    printf("%-34s K=%3d  %6.2f cycles / instruction   %8.0f MACs / cycle / CU\n", name, kdim, per, macs);
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount;
    printf("device %s, %d CUs (all CUs busy: the clock is the one the chip holds under this load)\n", prop.gcnArchName, blocks);
    std::vector<uint32_t> h(6 * 256 * 8);
    uint64_t st = 0x9e3779b97f4a7c15ull;
    for (auto& w : h) { st = st * 6364136223846793005ull + 1442695040888963407ull; w = (uint32_t)(st >> 32) & 0x3b3b3b3bu; }
    v8i* d; float* out; unsigned long long* ts;
    CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&out, (size_t)blocks * 256 * 4)); CK(hipMalloc(&ts, blocks * 8));
    CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    run<0, 4, 4>("f8f6f4  A fp4      x B fp4", 64, d, out, ts, blocks);
    run<0, 2, 4>("f8f6f4  A fp6 e2m3 x B fp4", 64, d, out, ts, blocks);
    run<0, 2, 2>("f8f6f4  A fp6 e2m3 x B fp6 e2m3", 64, d, out, ts, blocks);
    run<0, 0, 4>("f8f6f4  A fp8 e4m3 x B fp4", 64, d, out, ts, blocks);
    run<0, 0, 0>("f8f6f4  A fp8 e4m3 x B fp8 e4m3", 64, d, out, ts, blocks);
    run<1, 0, 0>("f16 32x32x16", 16, d, out, ts, blocks);
    run<2, 0, 0>("i8  32x32x32", 32, d, out, ts, blocks);
    return 0;
}
