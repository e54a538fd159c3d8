// Numerics prototype for DESIGN.md section 8 ("real-valued operands as fp6 digits on the f8f6f4 cores"), not part of the library:
// y[i][n] = sum_k x[i][k] s[n][k] for real x and s in {+1, -1}, with x written per row as four balanced base-31 digits of a
// fixed-point value (d in -15 .. 15, stored as the fp6 e2m3 number d / 8) and s as fp4, one v_mfma_scale_f32_32x32x64_f8f6f4 per
// digit and 64 K-elements.  One wave, a 32 x 32 output tile.  Checks (a) every digit plane's accumulator against the exact
// integer sum computed on the host — bit-exact, which pins the 6-bit packing and the e2m3 encoding — and (b) the combined result
// against the fp64 product.
// Build: hipcc --offload-arch=gfx950 -O3 fp6_digits.hip -o fp6_digits ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int M = 32, N = 32, K = 1024, ND = 4;

// A: [ND][M][K / 32][6] dwords (32 six-bit values per 24-byte group), B: [N][K / 32][4] dwords (32 nibbles per group)
// SC: [M][K / 32] E8M0 bytes (as ints): the power of two of each (row, 32-element block) of A relative to its row, or nullptr
__global__ __launch_bounds__(64) void k(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, const int* __restrict__ SC,
                                        float* __restrict__ acc_out) {
    const int lane = threadIdx.x, row = lane & 31, half = lane >> 5;
    v16f acc[ND];
    for (int d = 0; d < ND; ++d)
        for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 64) {
        const int grp = k0 / 32 + half;                       // this lane's 32 K-elements
        const uint32_t* bp = B + ((size_t)row * (K / 32) + grp) * 4;
        const v8i bv = {(int)bp[0], (int)bp[1], (int)bp[2], (int)bp[3], 0, 0, 0, 0};
        const int sa = SC ? SC[(size_t)row * (K / 32) + grp] : 0x7f;     // byte 0 of the lane's scale register = its block's E8M0
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            const uint32_t* ap = A + (((size_t)d * M + row) * (K / 32) + grp) * 6;
            const v8i av = {(int)ap[0], (int)ap[1], (int)ap[2], (int)ap[3], (int)ap[4], (int)ap[5], 0, 0};
            // cbsz = 2: A is fp6 e2m3; blgp = 4: B is fp4; scales 0x7f = 2^0
            acc[d] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[d], 2, 4, 0, sa, 0, 0x7f7f7f7f);
        }
    }
    // D[i][j]: j = lane % 32, i = 8 * (r / 4) + 4 * (lane / 32) + r % 4
    for (int d = 0; d < ND; ++d)
        for (int r = 0; r < 16; ++r) {
            const int i = 8 * (r / 4) + 4 * half + (r & 3);
            acc_out[((size_t)d * M + i) * N + row] = acc[d][r];
        }
}

static uint32_t e2m3_of_digit(int d) {      // the fp6 e2m3 code of d / 8, |d| <= 15
    const uint32_t s = d < 0 ? 0x20u : 0u;
    const int a = d < 0 ? -d : d;
    return s | (a < 8 ? (uint32_t)a : (0x8u | (uint32_t)(a - 8)));   // exp field 0: subnormal m / 8; exp field 1: (1 + m / 8)
}

static int run(bool block_scales) {
    std::vector<double> x((size_t)M * K);
    std::vector<int> s((size_t)N * K);
    uint64_t st = 0x243f6a8885a308d3ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)(st >> 11) / 9007199254740992.0; };
    for (int i = 0; i < M; ++i) {
        const double rowscale = exp2((double)(i % 9) - 4.0);                       // rows of different magnitude
        for (int kk = 0; kk < K; ++kk) {
            const double u = rnd(), v = rnd();
            const double blk = block_scales ? exp2(-(double)((kk / 32 * 5 + i) % 7)) : 1.0;   // blocks of different magnitude
            x[(size_t)i * K + kk] = (double)(float)(sqrt(-2.0 * log(u + 1e-300)) * cos(6.283185307179586 * v) * rowscale * blk);
        }
    }
    for (auto& v : s) v = rnd() < 0.5 ? -1 : 1;
    const long long HALF = (31ll * 31 * 31 * 31 - 1) / 2;                           // balanced 4-digit range: |q| <= HALF
    std::vector<uint32_t> A((size_t)ND * M * (K / 32) * 6, 0), B((size_t)N * (K / 32) * 4, 0);
    std::vector<int> dig((size_t)ND * M * K);
    std::vector<double> unit(M);                                                    // x ~ q * unit * 2^(block exponent)
    std::vector<int> bexp((size_t)M * (K / 32), 0), sc((size_t)M * (K / 32), 0x7f);
    for (int i = 0; i < M; ++i) {
        double mx = 0;
        for (int kk = 0; kk < K; ++kk) mx = fmax(mx, fabs(x[(size_t)i * K + kk]));
        int e; frexp(mx, &e);                                                       // mx < 2^e
        unit[i] = ldexp(1.0, e) / (double)HALF;
        for (int g = 0; g < K / 32; ++g) {
            int be = 0;
            if (block_scales) {
                double bm = 0;
                for (int kk = g * 32; kk < g * 32 + 32; ++kk) bm = fmax(bm, fabs(x[(size_t)i * K + kk]));
                int eb; frexp(bm, &eb);
                be = bm > 0 ? eb - e : 0;                                           // <= 0
                if (be < -60) be = -60;
            }
            bexp[(size_t)i * (K / 32) + g] = be;
            sc[(size_t)i * (K / 32) + g] = 0x7f + be;
        }
        for (int kk = 0; kk < K; ++kk) {
            const double ub = unit[i] * ldexp(1.0, bexp[(size_t)i * (K / 32) + kk / 32]);
            long long q = llround(x[(size_t)i * K + kk] / ub);
            if (q > HALF) q = HALF; if (q < -HALF) q = -HALF;
            for (int d = 0; d < ND; ++d) {                                          // balanced base 31
                long long r = ((q % 31) + 31) % 31;
                if (r > 15) r -= 31;
                dig[((size_t)d * M + i) * K + kk] = (int)r;
                q = (q - r) / 31;
                const uint32_t code = e2m3_of_digit((int)r);
                const size_t g = (((size_t)d * M + i) * (K / 32) + kk / 32) * 6;
                const int bit = (kk % 32) * 6;
                A[g + bit / 32] |= code << (bit % 32);
                if (bit % 32 > 26) A[g + bit / 32 + 1] |= code >> (32 - bit % 32);
            }
        }
    }
    for (int n = 0; n < N; ++n)
        for (int kk = 0; kk < K; ++kk)
            B[((size_t)n * (K / 32) + kk / 32) * 4 + (kk % 32) / 8] |= (s[(size_t)n * K + kk] > 0 ? 0x2u : 0xAu) << (4 * (kk % 8));
    uint32_t *dA, *dB; float* dacc; int* dsc;
    CK(hipMalloc(&dsc, sc.size() * 4)); CK(hipMemcpy(dsc, sc.data(), sc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dacc, (size_t)ND * M * N * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, block_scales ? dsc : (const int*)nullptr, dacc);
    CK(hipDeviceSynchronize());
    std::vector<float> acc((size_t)ND * M * N);
    CK(hipMemcpy(acc.data(), dacc, acc.size() * 4, hipMemcpyDeviceToHost));
    long long bad = 0;
    double num = 0, den = 0, worst = 0;
    for (int i = 0; i < M; ++i)
        for (int n = 0; n < N; ++n) {
            double comb = 0, p31 = 1;
            for (int d = 0; d < ND; ++d) {
                double want = 0;                                                    // exact: multiples of 2^-60 well inside 53 bits here
                for (int kk = 0; kk < K; ++kk)
                    want += ldexp((double)(dig[((size_t)d * M + i) * K + kk] * s[(size_t)n * K + kk]), bexp[(size_t)i * (K / 32) + kk / 32]);
                const double got8 = (double)acc[((size_t)d * M + i) * N + n] * 8.0;  // the instruction summed d / 8
                if (got8 != want) { if (bad < 5) printf("digit %d D[%d][%d]: got %.6f want %.6f\n", d, i, n, got8, want); ++bad; }
                comb += got8 * p31;
                p31 *= 31.0;
            }
            const double y = comb * unit[i];
            double ref = 0;
            for (int kk = 0; kk < K; ++kk) ref += x[(size_t)i * K + kk] * s[(size_t)n * K + kk];
            num += (y - ref) * (y - ref); den += ref * ref;
            worst = fmax(worst, fabs(y - ref) / (fabs(ref) + 1e-30));
        }
    printf("%s: M %d N %d K %d, %d fp6 digits per element: digit-plane sums wrong in %lld of %d entries (must be 0: exact)\n", block_scales ? "E8M0 scale per 32-element block of A" : "per-row scale only", M, N, K, ND, bad, ND * M * N);
    printf("combined result vs fp64: normalised error %.3e (bar 1e-5)\n", sqrt(num / den));
    return bad ? 1 : 0;
}

int main() { return run(false) | run(true); }
