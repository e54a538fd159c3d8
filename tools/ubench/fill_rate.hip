// How fast can one CU fill its LDS from L2-resident data on gfx950?  (a) LDS-DMA (global_load_lds_dwordx4), (b) global_load_dwordx4
// into VGPRs + ds_write_b128.  Every kernel in this repo that is not matrix-bound is bound by (a) at ~22 B / cycle / CU; this
// measures whether the register path has a higher ceiling.  One workgroup per CU streams a private 64 KiB region (L2-resident
// after the first pass) into LDS over and over.
// Build: hipcc --offload-arch=gfx950 -O3 fill_rate.hip -o fill_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int REGION = 96 * 1024;      // bytes per workgroup: 32 workgroups per XCD x 96 KiB = 3 MiB of each 4 MiB L2, three times the 32 KiB L1
constexpr int CHUNK = 32 * 1024;       // bytes per iteration

// PATTERN 0: a piece = 1 KiB contiguous; 1: 8 rows x 128 B, rows 768 B apart (GEMM / conv stage rows); 2: 16 rows x 64 B, rows 768 B apart
// (the pixel-major kernel's 32-channel sub-tiles: half a cache line per row)
template <int WAVES, int PATTERN>
__global__ __launch_bounds__(WAVES * 64) void dma_kernel(const unsigned char* __restrict__ src, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = src + (size_t)blockIdx.x * REGION;
    constexpr int PIECES = CHUNK / 1024, PW = PIECES / WAVES;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
    for (int it = 0; it < iters; ++it) {
        const unsigned char* s = base + (it % 3) * CHUNK;
        const unsigned buf = lds0 + (unsigned)((it % 3) * CHUNK);
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            const int p = j * WAVES + wave;
            // patterns 1 / 2 walk the same 32 KiB as [rows][768 B]: 42 rows of 768 B; a piece takes 8 (16) consecutive rows at one
            // of the 6 (12) column positions
            const unsigned char* a = PATTERN == 0 ? s + p * 1024 + lane * 16
                                   : PATTERN == 1 ? s + ((p % 5) * 8 + (lane >> 3)) * 768 + (p / 5) * 128 + (lane & 7) * 16
                                                  : s + ((p % 2) * 16 + (lane >> 2)) * 768 + (p / 2 % 12) * 64 + (lane & 3) * 16;
            const unsigned dst = buf + p * 1024;
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(a), "s"(dst) : "memory");
        }
        // two chunks in flight
        if (PW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (PW == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *(volatile unsigned*)smem;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void vgpr_kernel(const unsigned char* __restrict__ src, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned char* base = src + (size_t)blockIdx.x * REGION;
    constexpr int PER = CHUNK / (WAVES * 64) / 16;     // dwordx4 loads per thread per chunk
    uint4 r[2][PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) r[0][j] = *reinterpret_cast<const uint4*>(base + (j * WAVES * 64 + threadIdx.x) * 16);
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const unsigned char* s = base + ((it + h + 1) % 3) * CHUNK;
#pragma unroll
            for (int j = 0; j < PER; ++j) r[h ^ 1][j] = *reinterpret_cast<const uint4*>(s + (j * WAVES * 64 + threadIdx.x) * 16);
            unsigned char* buf = smem + ((it + h) % 3) * CHUNK;
#pragma unroll
            for (int j = 0; j < PER; ++j) *reinterpret_cast<uint4*>(buf + (j * WAVES * 64 + threadIdx.x) * 16) = r[h][j];
            __builtin_amdgcn_s_barrier();
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = *(volatile unsigned*)smem + r[0][0].x;
}


// Streaming case: every workgroup walks a private region much larger than any cache (no reuse: the data comes from HBM), with
// two 32 KiB chunks of LDS-DMA in flight — the situation of the conv kernels' A operand.  PF > 0: a NINTH wave touches one dword
// per 128-byte line PF chunks ahead (plain global loads into a register nobody reads; its own vmcnt), so that the DMA finds its
// lines in L2.
template <int PF>
__global__ __launch_bounds__(576) void stream_kernel(const unsigned char* __restrict__ src, size_t region, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned char* base = src + (size_t)blockIdx.x * region;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (wave < 8) {
            const unsigned char* s = base + (size_t)it * CHUNK;
            const unsigned buf = lds0 + (unsigned)((it % 3) * CHUNK);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = j * 8 + wave;
                const unsigned char* a = s + p * 1024 + lane * 16;
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(a), "s"(buf + p * 1024) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (PF > 0 && it + PF < iters) {
            const unsigned char* s = base + (size_t)(it + PF) * CHUNK;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += *reinterpret_cast<const volatile unsigned*>(s + (j * 64 + lane) * 128);   // 256 lines = 32 KiB
        }
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) sink[blockIdx.x * 16 + wave] = *(volatile unsigned*)smem + acc;
}

template <int PF>
void run_stream(const unsigned char* src, size_t region, unsigned* sink) {
    const int cus = 256, iters = (int)(region / CHUNK);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<PF>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * CHUNK);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(stream_kernel<PF>, dim3(cus), dim3(576), 3 * CHUNK, 0, src, region, iters, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)cus * iters * CHUNK;
    printf("HBM stream, 2 x 32 KiB DMA in flight, L2 prefetch %2d chunks ahead: %7.3f ms  %6.2f TB/s = %5.1f B/ns/CU  [%s]\n", PF, ms,
           bytes / ms / 1e9, bytes / ms / 1e6 / cus, hipGetErrorString(hipGetLastError()));
}

template <typename K>
void run(const char* name, K kernel, int threads, int wgs_per_cu, const unsigned char* src, unsigned* sink) {
    const int cus = 256, iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * CHUNK);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel, dim3(cus * wgs_per_cu), dim3(threads), 3 * CHUNK, 0, src, iters, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)cus * wgs_per_cu * iters * CHUNK;
    printf("%-34s %d WG/CU x %3d thr: %7.3f ms  %6.2f TB/s  = %5.1f B/ns/CU (%.1f B/cycle/CU at 2.4 GHz)  [%s]\n", name, wgs_per_cu, threads, ms,
           bytes / ms / 1e9, bytes / ms / 1e6 / cus, bytes / ms / 1e6 / cus / 2.4, hipGetErrorString(hipGetLastError()));
}

int main() {
    unsigned char* src;
    unsigned* sink;
    hipMalloc(&src, (size_t)REGION * 512);
    hipMemset(src, 1, (size_t)REGION * 512);
    hipMalloc(&sink, 65536);
    run("LDS-DMA 1 KiB contiguous, 8 waves", dma_kernel<8, 0>, 512, 1, src, sink);
    run("LDS-DMA 1 KiB contiguous, 4 waves", dma_kernel<4, 0>, 256, 1, src, sink);
    run("LDS-DMA 8 rows x 128 B, 8 waves", dma_kernel<8, 1>, 512, 1, src, sink);
    run("LDS-DMA 16 rows x 64 B, 8 waves", dma_kernel<8, 2>, 512, 1, src, sink);
    run("global_load + ds_write, 8 waves", vgpr_kernel<8>, 512, 1, src, sink);
    run("global_load + ds_write, 4 waves", vgpr_kernel<4>, 256, 1, src, sink);
    run("global_load + ds_write, 16 waves", vgpr_kernel<16>, 1024, 1, src, sink);
    unsigned char* big;
    const size_t region = 4u << 20;
    (void)hipMalloc(&big, region * 256);
    (void)hipMemset(big, 1, region * 256);
    (void)hipDeviceSynchronize();
    run_stream<0>(big, region, sink);
    run_stream<0>(big, region, sink);
    run_stream<2>(big, region, sink);
    run_stream<4>(big, region, sink);
    run_stream<8>(big, region, sink);
    run_stream<16>(big, region, sink);
    return 0;
}
