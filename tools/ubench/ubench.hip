// Instruction-rate micro-benchmarks + MFMA fragment-layout checks for gfx950 (MI355X).
// Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench ; run on the GPU box.
// Output is committed under profiles/ as the measured basis of DESIGN.md's ceilings.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int ITERS = 4096;
constexpr int UNROLL = 16;

// ---- VALU op rates: 16 independent chains per lane, ITERS iterations --------------------------
#define RATE_KERNEL(NAME, ASMSTR)                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, uint32_t seed) {           \
        uint32_t r[UNROLL];                                                               \
        uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;                       \
        for (int i = 0; i < UNROLL; ++i) r[i] = a + i;                                    \
        for (int it = 0; it < ITERS; ++it) {                                              \
            _Pragma("unroll") for (int i = 0; i < UNROLL; ++i)                            \
                asm volatile(ASMSTR : "+v"(r[i]) : "v"(a), "v"(b));                       \
        }                                                                                 \
        uint32_t s = 0;                                                                   \
        for (int i = 0; i < UNROLL; ++i) s += r[i];                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                   \
    }

RATE_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
RATE_KERNEL(k_add, "v_add_u32 %0, %0, %1")
RATE_KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0")
RATE_KERNEL(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
RATE_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
RATE_KERNEL(k_xad, "v_xad_u32 %0, %0, %1, %2")
RATE_KERNEL(k_mul24, "v_mul_u32_u24 %0, %0, %1")
RATE_KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
RATE_KERNEL(k_dot8_i4, "v_dot8_i32_i4 %0, %1, %2, %0")
RATE_KERNEL(k_dot4_i8, "v_dot4_i32_i8 %0, %1, %2, %0")
RATE_KERNEL(k_sad_u8, "v_sad_u8 %0, %1, %2, %0")
RATE_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1")
RATE_KERNEL(k_xor_bcnt, "v_xor_b32 %0, %0, %1\n\tv_bcnt_u32_b32 %0, %0, %2")

// ---- MFMA rates: 4 independent accumulators per wave -------------------------------------------
__global__ __launch_bounds__(256) void k_mfma_i8_32(int* out) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    v16i c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < ITERS / 4; ++it) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mfma_i8_16(int* out) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, 6, (int)blockIdx.x};
    typedef int v4a __attribute__((ext_vector_type(4)));
    v4a c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < ITERS / 4; ++it) {
        c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c3, 0, 0, 0);
    }
    int s = 0;
    for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mfma_fp4_32(float* out) {
    v8i a = {0x22222222, 0x2a2a2a2a, (int)0xa2a2a2a2, 0x22aa22aa, 0, 0, 0, 0};
    v8i b = {0x2222aaaa, 0x2a2a2a2a, (int)0xa2a2a2a2, 0x22aa22aa, 0, 0, 0, 0};
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < ITERS / 4; ++it) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_mfma_fp4_16(float* out) {
    v8i a = {0x22222222, 0x2a2a2a2a, (int)0xa2a2a2a2, 0x22aa22aa, 0, 0, 0, 0};
    v8i b = {0x2222aaaa, 0x2a2a2a2a, (int)0xa2a2a2a2, 0x22aa22aa, 0, 0, 0, 0};
    v4f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < ITERS / 4; ++it) {
        c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c1, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c2 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c2, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c3 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c3, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- layout checks: one wave, operands from global memory in the ASSUMED lane layout --------------
// fp4 32x32x64: lane l holds A[row=l&31][k = 32*(l>>5) + j], j=0..31 as 32 nibbles (4 dwords),
// element j in nibble (j&7) of dword (j>>3); B likewise with col = l&31.
__global__ void k_check_fp4_32(const uint32_t* A, const uint32_t* B, float* C) {
    const int l = threadIdx.x;
    v8i a = {0}, b = {0};
    for (int d = 0; d < 4; ++d) { a[d] = A[l * 4 + d]; b[d] = B[l * 4 + d]; }
    v16f c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 16; ++r) C[r * 64 + l] = c[r];
}
// fp4 16x16x128: lane l holds A[row=l&15][k = 32*(l>>4) + j]
__global__ void k_check_fp4_16(const uint32_t* A, const uint32_t* B, float* C) {
    const int l = threadIdx.x;
    v8i a = {0}, b = {0};
    for (int d = 0; d < 4; ++d) { a[d] = A[l * 4 + d]; b[d] = B[l * 4 + d]; }
    v4f c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 4; ++r) C[r * 64 + l] = c[r];
}
// i8 32x32x32: lane l holds A[row=l&31][k = 16*(l>>5) + j], j=0..15 (4 dwords, byte j)
__global__ void k_check_i8_32(const uint32_t* A, const uint32_t* B, int* C) {
    const int l = threadIdx.x;
    v4i a, b;
    for (int d = 0; d < 4; ++d) { a[d] = A[l * 4 + d]; b[d] = B[l * 4 + d]; }
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[r * 64 + l] = c[r];
}

static float fp4_val(uint32_t nib) {
    static const float mag[8] = {0.f, .5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    float v = mag[nib & 7];
    return (nib & 8) ? -v : v;
}

template <class F>
static double time_kernel(F launch, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz\n", prop.gcnArchName, cus, prop.clockRate / 1000);
    const int blocks = cus * 8, threads = 256;   // 8 waves/SIMD... 32 waves/CU
    uint32_t* dout; CK(hipMalloc(&dout, (size_t)blocks * threads * 4));

    struct R { const char* name; void (*fn)(uint32_t*, uint32_t); int ops_per_asm; };
    R rates[] = {{"v_xor_b32", k_xor, 1}, {"v_add_u32", k_add, 1}, {"v_bcnt_u32_b32", k_bcnt, 1},
                 {"v_bitop3_b32", k_bitop3, 1}, {"v_add3_u32", k_add3, 1}, {"v_xad_u32", k_xad, 1},
                 {"v_mul_u32_u24", k_mul24, 1}, {"v_mul_lo_u32", k_mullo, 1},
                 {"v_dot8_i32_i4", k_dot8_i4, 1}, {"v_dot4_i32_i8", k_dot4_i8, 1}, {"v_sad_u8", k_sad_u8, 1},
                 {"v_lshl_or_b32", k_lshl_or, 1}, {"xor+bcnt pair", k_xor_bcnt, 2}};
    printf("\n== VALU issue rates (32 waves/CU resident, %d x %d dependent-free ops per lane) ==\n", ITERS, UNROLL);
    for (auto& r : rates) {
        double ms = time_kernel([&] { hipLaunchKernelGGL(r.fn, dim3(blocks), dim3(threads), 0, 0, dout, 7u); });
        double lane_ops = (double)blocks * threads * ITERS * UNROLL * r.ops_per_asm;
        double tlops = lane_ops / (ms * 1e-3) / 1e12;
        // cycles per wave-instruction per SIMD at the nominal 2.4 GHz clock
        double wave_instr = lane_ops / 64.0;
        double cyc = (ms * 1e-3) * 2.4e9 * cus * 4 / wave_instr;
        printf("%-16s %8.3f ms  %7.2f T lane-ops/s  %5.2f cyc/wave-instr/SIMD @2.4GHz\n", r.name, ms, tlops, cyc);
    }

    printf("\n== MFMA rates (4 waves/CU... blocks=%d x 256 thr, 4 independent accumulators) ==\n", cus * 2);
    {
        const int mb = cus * 2;
        double ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma_i8_32, dim3(mb), dim3(256), 0, 0, (int*)dout); });
        double macs = (double)mb * 4 * ITERS * 32.0 * 32 * 32;
        printf("%-28s %8.3f ms  %8.1f TOPS (2*MAC)\n", "mfma_i32_32x32x32_i8", ms, 2 * macs / (ms * 1e-3) / 1e12);
        ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma_i8_16, dim3(mb), dim3(256), 0, 0, (int*)dout); });
        macs = (double)mb * 4 * ITERS * 16.0 * 16 * 64;
        printf("%-28s %8.3f ms  %8.1f TOPS (2*MAC)\n", "mfma_i32_16x16x64_i8", ms, 2 * macs / (ms * 1e-3) / 1e12);
        ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma_fp4_32, dim3(mb), dim3(256), 0, 0, (float*)dout); });
        macs = (double)mb * 4 * ITERS * 32.0 * 32 * 64;
        printf("%-28s %8.3f ms  %8.1f TFLOPS\n", "mfma_scale_32x32x64 fp4", ms, 2 * macs / (ms * 1e-3) / 1e12);
        ms = time_kernel([&] { hipLaunchKernelGGL(k_mfma_fp4_16, dim3(mb), dim3(256), 0, 0, (float*)dout); });
        macs = (double)mb * 4 * ITERS * 16.0 * 16 * 128;
        printf("%-28s %8.3f ms  %8.1f TFLOPS\n", "mfma_scale_16x16x128 fp4", ms, 2 * macs / (ms * 1e-3) / 1e12);
    }

    // ---- layout checks ----
    printf("\n== MFMA fragment layout checks (assumed lane layouts, asymmetric random operands) ==\n");
    srand(12345);
    const uint32_t nibs[3] = {0x2, 0xA, 0x0};
    {   // fp4 32x32x64
        std::vector<float> Am(32 * 64), Bm(64 * 32);
        std::vector<uint32_t> Ap(64 * 4, 0), Bp(64 * 4, 0);
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
            uint32_t na = nibs[rand() % 3], nb = nibs[rand() % 3];
            int k = 32 * (l >> 5) + j;
            Am[(l & 31) * 64 + k] = fp4_val(na); Bm[k * 32 + (l & 31)] = fp4_val(nb);
            Ap[l * 4 + (j >> 3)] |= na << (4 * (j & 7)); Bp[l * 4 + (j >> 3)] |= nb << (4 * (j & 7));
        }
        uint32_t *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 16 * 64 * 4));
        CK(hipMemcpy(dA, Ap.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bp.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_check_fp4_32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        std::vector<float> C(16 * 64); CK(hipMemcpy(C.data(), dC, 16 * 64 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
            int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            float ref = 0; for (int k = 0; k < 64; ++k) ref += Am[row * 64 + k] * Bm[k * 32 + col];
            if (ref != C[r * 64 + l]) ++bad;
        }
        printf("fp4 32x32x64  : %s (%d mismatches of 1024)\n", bad ? "MISMATCH" : "layout OK", bad);
    }
    {   // fp4 16x16x128
        std::vector<float> Am(16 * 128), Bm(128 * 16);
        std::vector<uint32_t> Ap(64 * 4, 0), Bp(64 * 4, 0);
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) {
            uint32_t na = nibs[rand() % 3], nb = nibs[rand() % 3];
            int k = 32 * (l >> 4) + j;
            Am[(l & 15) * 128 + k] = fp4_val(na); Bm[k * 16 + (l & 15)] = fp4_val(nb);
            Ap[l * 4 + (j >> 3)] |= na << (4 * (j & 7)); Bp[l * 4 + (j >> 3)] |= nb << (4 * (j & 7));
        }
        uint32_t *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 4 * 64 * 4));
        CK(hipMemcpy(dA, Ap.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bp.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_check_fp4_16, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        std::vector<float> C(4 * 64); CK(hipMemcpy(C.data(), dC, 4 * 64 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            int col = l & 15, row = (l >> 4) * 4 + r;
            float ref = 0; for (int k = 0; k < 128; ++k) ref += Am[row * 128 + k] * Bm[k * 16 + col];
            if (ref != C[r * 64 + l]) ++bad;
        }
        printf("fp4 16x16x128 : %s (%d mismatches of 256)\n", bad ? "MISMATCH" : "layout OK", bad);
    }
    {   // i8 32x32x32
        std::vector<int> Am(32 * 32), Bm(32 * 32);
        std::vector<uint32_t> Ap(64 * 4, 0), Bp(64 * 4, 0);
        for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) {
            int va = rand() % 31 - 15, vb = rand() % 31 - 15;
            int k = 16 * (l >> 5) + j;
            Am[(l & 31) * 32 + k] = va; Bm[k * 32 + (l & 31)] = vb;
            Ap[l * 4 + (j >> 2)] |= (uint32_t)(uint8_t)va << (8 * (j & 3)); Bp[l * 4 + (j >> 2)] |= (uint32_t)(uint8_t)vb << (8 * (j & 3));
        }
        uint32_t *dA, *dB; int* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 16 * 64 * 4));
        CK(hipMemcpy(dA, Ap.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, Bp.data(), 1024, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_check_i8_32, dim3(1), dim3(64), 0, 0, dA, dB, dC);
        std::vector<int> C(16 * 64); CK(hipMemcpy(C.data(), dC, 16 * 64 * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
            int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            int ref = 0; for (int k = 0; k < 32; ++k) ref += Am[row * 32 + k] * Bm[k * 32 + col];
            if (ref != C[r * 64 + l]) ++bad;
        }
        printf("i8  32x32x32  : %s (%d mismatches of 1024)\n", bad ? "MISMATCH" : "layout OK", bad);
    }
    return 0;
}
