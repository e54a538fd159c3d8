// Does the fp4 matrix pipe hold its short-burst rate (profiles/r1_ubench.txt: 9.0 PFLOP/s) under the
// conditions of the packed GEMM?  Same register shape as mfma_gemm.hip's 256x256 tile (8 waves per
// workgroup, 2 per SIMD, 8 accumulator tiles per wave, 4 A + 2 B fragments), no memory traffic in the
// loop, varying (a) operand data (zeros / all +1 / random +-1), (b) run length, (c) a workgroup
// barrier every 32 MFMAs.  Reports the achieved rate and the shader clock during the kernel
// (clock64 ticks per wall_clock64 tick).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v16f mfma(uint4 a, uint4 b, v16f c) {
    v8i av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
    v8i bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

template <int MODE>   // 0 = MFMAs only, 1 = + workgroup barrier every 32 MFMAs
__global__ __launch_bounds__(512, 2) void k_fp4(const uint4* __restrict__ data, float* out, int iters,
                                                unsigned long long* ts) {
    const int t = threadIdx.x;
    uint4 a[4], b[2];
    for (int i = 0; i < 4; ++i) a[i] = data[(blockIdx.x * 6 + i) * 512 + t];
    for (int i = 0; i < 2; ++i) b[i] = data[(blockIdx.x * 6 + 4 + i) * 512 + t];
    v16f c[4][2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) c[i][j][r] = 0.f;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) c[i][j] = mfma(a[i], b[j], c[i][j]);
        }
        if (MODE == 1) __syncthreads();
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += c[i][j][r];
    out[blockIdx.x * 512 + t] = s;
    if (t == 0) { ts[blockIdx.x * 2] = c1 - c0; ts[blockIdx.x * 2 + 1] = w1 - w0; }
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, blocks = cus;
    int wall_khz = 0; CK(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0));
    printf("device %s, %d CUs, nominal clock %d MHz, wall clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate / 1000, wall_khz);
    const size_t nvec = (size_t)blocks * 6 * 512;
    std::vector<uint32_t> h(nvec * 4);
    uint4* d; float* out; unsigned long long* ts;
    CK(hipMalloc(&d, nvec * 16)); CK(hipMalloc(&out, (size_t)blocks * 512 * 4)); CK(hipMalloc(&ts, blocks * 16));
    std::vector<unsigned long long> hts(blocks * 2);
    const char* dname[3] = {"zeros", "all +1", "random +-1"};
    for (int dk = 0; dk < 3; ++dk) {
        uint64_t st = 0x9e3779b97f4a7c15ull;
        for (auto& w : h) {
            if (dk == 0) w = 0;
            else if (dk == 1) w = 0x22222222u;
            else {
                st = st * 6364136223846793005ull + 1442695040888963407ull;
                uint32_t r = (uint32_t)(st >> 32), v = 0;
                for (int n = 0; n < 8; ++n) v |= (((r >> n) & 1) ? 0xAu : 0x2u) << (4 * n);
                w = v;
            }
        }
        CK(hipMemcpy(d, h.data(), nvec * 16, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 2; ++mode)
            for (int iters : {16, 160, 1600, 16000}) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                double best = 1e30;
                for (int rep = 0; rep < 6; ++rep) {
                    CK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(k_fp4<0>, dim3(blocks), dim3(512), 0, 0, d, out, iters, ts);
                    else hipLaunchKernelGGL(k_fp4<1>, dim3(blocks), dim3(512), 0, 0, d, out, iters, ts);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) best = ms;
                }
                CK(hipMemcpy(hts.data(), ts, blocks * 16, hipMemcpyDeviceToHost));
                double cyc = 0, wall = 0;
                for (int b = 0; b < blocks; ++b) { cyc += hts[2 * b]; wall += hts[2 * b + 1]; }
                cyc /= blocks; wall /= blocks;
                const double flop = 2.0 * 32 * 32 * 64 * 32.0 * iters * 8 * blocks;
                const double in_us = wall / (wall_khz * 1e-3);
                printf("%-11s %-8s iters %6d: event %9.1f us  in-kernel %9.1f us  %7.1f TFLOP/s (in-kernel)  clock64/wall = %.3f  -> %.0f MHz if clock64 = shader clock; cycles/MFMA/SIMD %.1f\n",
                       dname[dk], mode ? "barrier" : "free", iters, best * 1e3, in_us, flop / in_us / 1e6,
                       cyc / wall, cyc / wall * wall_khz * 1e-3, cyc / (iters * 32.0 * 2));
            }
    }
    return 0;
}
