// Packed low-bit GEMM on the matrix cores: +-1 / {-1,0,+1} operands stored as FP4-E2M1 nibbles
// ("nibble planes", include/qt_hip.h) and contracted with the block-scaled MX MFMA
//     v_mfma_scale_f32_32x32x64_f8f6f4   (A = fp4, B = fp4, every block scale = 2^0)
// E2M1 represents -1, 0, +1 exactly and the accumulator is fp32, so every partial sum is an
// integer of magnitude <= K < 2^24: the result is bit-identical to the popcount formulation and
// to the reference's fp32 GEMM on +-1 tensors.  Measured issue rate of this instruction on
// MI355X: 9.0 PFLOP/s (profiles/ubench_r1.txt) vs 1.49 Pop/s for the xor+bcnt pair
// (v_bcnt_u32_b32 is half-rate), which is why large shapes are routed here.
//
// Y[m,n] = sum_k X[m,k] * W[n,k] (+ bias[n]);  X: M x K nibbles, W: N x K nibbles, row-major,
// row stride a multiple of 4 words (32 nibbles); nibbles past K are zero (fp4 zero = 0x0).
//
// Tiling: 512-thread workgroup (8 waves = 2 waves per SIMD) owns a 256 (m) x 256 (n) tile; wave w
// owns 128 (m) x 64 (n) = 4 x 2 MFMA tiles (128 accumulator registers).  The W tile is the MFMA
// "A" operand (D rows = n) and the X tile the "B" operand (D cols = m): with the 32x32 C/D layout
// (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) each lane then owns 4 CONSECUTIVE n
// for a fixed m, i.e. a float4 of a Y row -> 32 wide stores per wave instead of 128 scalar ones.
//
// K is consumed in stages of 256 elements (128 B per row).  A stage is 256 rows x 128 B per
// operand = 64 KiB for both, double-buffered in LDS (128 KiB of the CU's 160 KiB).  Rows are stored
// as eight 16-byte chunks, chunk c of row r at position c ^ ((r>>1)&7): the 16 rows a
// ds_read_b128 lane group touches (distinct mod 16) then fall on 16 distinct 16-B slots of the
// 256-B bank row -> conflict-free fragment reads.  The stage s+1 tiles are brought in by
// LDS-DMA (global_load_lds_dwordx4: no staging registers, swizzle applied on the source address)
// issued before the MFMAs of stage s; one barrier per stage.
#include <type_traits>
#include "qt_common.h"

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr int STAGE_BYTES = 128;  // K bytes per row per stage (256 nibbles = 4 MFMA k-steps)

// 16 zero bytes in global memory: the source of every DMA chunk that lies past a row's stride.
__device__ __attribute__((aligned(16))) const unsigned char zero16_storage[16] = {0};

__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }

// One LDS-DMA piece (64 lanes x 16 B -> 1 KiB of LDS at the wave-uniform byte address lds_dst)
// issued from inline asm so that hipcc does not know a VMEM->LDS write is outstanding: with the
// builtin it drains vmcnt(0) in front of the next ds_read (possible alias), which serialises the
// DMA against the whole fragment-read + MFMA phase.  M0 = LDS byte address; M0 is compiler-
// reserved, so it is saved/restored inside the same statement; the s_nop covers the
// SALU-write-M0 -> LDS-DMA hazard.  No VGPR destination, so the statement is register-safe; the
// DATA is ordered for readers only by our own s_waitcnt vmcnt(0) + barrier at the end of a stage.
// Address form: 64-bit SGPR base + 32-bit per-lane VGPR byte offset (saddr form) — one VGPR per
// piece, and advancing to the next K stage is a scalar add on the base.
__device__ __forceinline__ void glds16_asm(const unsigned char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

// Tile configuration.  Workgroup = WM x WN waves; wave tile = (TMW*32) m-rows x (TNW*32) n-rows.
//   GLDS = true : staging by LDS-DMA (global_load_lds_dwordx4), no staging VGPRs.
//   GLDS = false: staging through registers (global_load_dwordx4 early, ds_write_b128 late).
//   ABL: ablation switch for profiling only (0 = full kernel, 1 = no MFMA, 2 = no staging,
//        3 = epilogue only); results are wrong unless ABL == 0.
//   EPI: epilogue form. 0 = W is the MFMA A operand (lane owns 4 consecutive n: float4 stores,
//        32 rows per instruction); 1 = X is the A operand (lane = column n: dword stores, each
//        instruction writes two full 128-B lines).
//   PIPE: 1 = LDS-DMA issued through inline asm (invisible to hipcc's wait-count bookkeeping) and
//        interleaved with the per-k-step fragment reads and MFMAs; one manual vmcnt(0) per stage.
template <int WM_, int WN_, int TMW_, int TNW_, bool GLDS_, int ABL_ = 0, int EPI_ = 0, int PIPE_ = 0>
struct NibCfg {
    static constexpr int ABL = ABL_, EPI = EPI_, PIPE = PIPE_;
    static constexpr int WM = WM_, WN = WN_, TMW = TMW_, TNW = TNW_;
    static constexpr bool GLDS = GLDS_;
    static constexpr int NWAVES = WM * WN, NTHREADS = NWAVES * 64;
    static constexpr int TM = WM * TMW * 32, TN = WN * TNW * 32;  // workgroup tile
    static constexpr int X_STAGE = TM * STAGE_BYTES, W_STAGE = TN * STAGE_BYTES;
    static constexpr int LDS_BYTES = 2 * (X_STAGE + W_STAGE);     // double-buffered
    static constexpr int WAVES_PER_SIMD = (NWAVES + 3) / 4;
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS, C::WAVES_PER_SIMD) void nib_gemm_kernel(
    const uint32_t* __restrict__ X, int64_t ldx, const uint32_t* __restrict__ W, int64_t ldw,
    const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy, int M, int N, int K,
    int vec_store, unsigned long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // smem: [buf 0: X stage | W stage][buf 1: X stage | W stage]
    constexpr int BUF = C::X_STAGE + C::W_STAGE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_n = wave % C::WN, wave_m = wave / C::WN;
    // XCD-aware tile order.  Workgroup b is observed to run on XCD b % 8 (speed only, never relied
    // on for correctness).  When the tile grid divides into 4 (m) x 8 (n) super-tiles, the 32 tiles
    // that share an XCD form ONE super-tile: 4 X panels + 8 W panels per L2 instead of a whole
    // tile-row/column stripe (18 panels at 16x16 tiles), which cuts the fabric re-fetch ~1.5x.
    int tile_m = blockIdx.y, tile_n = blockIdx.x;
    {
        const int gx = gridDim.x, gy = gridDim.y;
        if ((gx & 7) == 0 && (gy & 3) == 0) {
            const int b = blockIdx.y * gx + blockIdx.x;
            const int xcd = b & 7, j = b >> 3;           // j-th workgroup of this XCD
            const int o = xcd * ((gx * gy) >> 3) + j;    // XCD x owns a contiguous range of the
            const int st = o >> 5, in_st = o & 31;       // super-tile-major order (bijective: 8 | gx*gy)
            const int sgx = gx >> 3;                     // super-tiles per row
            tile_m = (st / sgx) * 4 + (in_st >> 3);
            tile_n = (st % sgx) * 8 + (in_st & 7);
        }
    }
    const int m0 = tile_m * C::TM, n0 = tile_n * C::TN;

    const int64_t ldx_b = ldx * 4, ldw_b = ldw * 4;  // row strides in bytes
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(X);
    const unsigned char* Wb = reinterpret_cast<const unsigned char*>(W);
    const unsigned char* zero16 = zero16_storage;

    v16f acc[C::TNW][C::TMW];
#pragma unroll
    for (int a = 0; a < C::TNW; ++a)
#pragma unroll
        for (int b = 0; b < C::TMW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int kbytes = (K + 1) / 2;
    const int nstages = (kbytes + STAGE_BYTES - 1) / STAGE_BYTES;

    // ---- staging, variant 1: LDS-DMA ---------------------------------------------------------
    // One wave-instruction moves 64 x 16 B = 1 KiB = 8 consecutive tile rows.  The LDS side is
    // lane-linear (base + lane*16) by construction of the instruction, so the swizzle is applied
    // on the SOURCE: lane i lands on (row = r0 + i/8, position p = i%8) and therefore fetches the
    // logical chunk c = p ^ ((row>>1)&7) of that row — still one full 128-B line per 8 lanes.
    // Rows past M / N are clamped to the last valid row (their products are never stored);
    // chunks past the row stride read a 16-byte zero word (they must contribute 0).
    auto dma_operand = [&](const unsigned char* G, int64_t ld_b, int row0_global, int nrows_valid,
                           unsigned char* ls, int tile_rows, int s) {
        const int p = lane & 7, rsub = lane >> 3;
        const int uwave = __builtin_amdgcn_readfirstlane(wave);  // provably wave-uniform LDS base
#pragma unroll
        for (int q = 0; q < (C::TM > C::TN ? C::TM : C::TN) / 8 / C::NWAVES; ++q) {
            const int g = q * C::NWAVES + uwave;
            if (g >= tile_rows / 8) break;
            const int r0 = g * 8, row = r0 + rsub;
            const int c = p ^ ((row >> 1) & 7);
            const int64_t kb = (int64_t)s * STAGE_BYTES + c * 16;
            const int grow = min(row0_global + row, nrows_valid - 1);
            const unsigned char* src = (kb < ld_b) ? G + (int64_t)grow * ld_b + kb : zero16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ls + r0 * STAGE_BYTES),
                                             16, 0, 0);
        }
    };
    // ---- staging, variant 2: through registers ----------------------------------------------------
    constexpr int XCH = C::TM * 8 / C::NTHREADS, WCH = C::TN * 8 / C::NTHREADS;  // 16-B chunks/thread
    uint4 xr[C::GLDS ? 1 : XCH], wr[C::GLDS ? 1 : WCH];
    auto load_regs = [&](int s) {
#pragma unroll
        for (int q = 0; q < XCH; ++q) {
            const int id = q * C::NTHREADS + tid, row = id >> 3, c = id & 7;
            const int64_t kb = (int64_t)s * STAGE_BYTES + c * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m0 + row < M && kb < ldx_b)
                v = *reinterpret_cast<const uint4*>(Xb + (int64_t)(m0 + row) * ldx_b + kb);
            xr[q] = v;
        }
#pragma unroll
        for (int q = 0; q < WCH; ++q) {
            const int id = q * C::NTHREADS + tid, row = id >> 3, c = id & 7;
            const int64_t kb = (int64_t)s * STAGE_BYTES + c * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (n0 + row < N && kb < ldw_b)
                v = *reinterpret_cast<const uint4*>(Wb + (int64_t)(n0 + row) * ldw_b + kb);
            wr[q] = v;
        }
    };
    auto store_regs = [&](int buf) {
        unsigned char* xs = smem + buf * BUF;
        unsigned char* ws = xs + C::X_STAGE;
#pragma unroll
        for (int q = 0; q < XCH; ++q) {
            const int id = q * C::NTHREADS + tid, row = id >> 3, c = id & 7;
            *reinterpret_cast<uint4*>(xs + row * STAGE_BYTES + swz(row, c) * 16) = xr[q];
        }
#pragma unroll
        for (int q = 0; q < WCH; ++q) {
            const int id = q * C::NTHREADS + tid, row = id >> 3, c = id & 7;
            *reinterpret_cast<uint4*>(ws + row * STAGE_BYTES + swz(row, c) * 16) = wr[q];
        }
    };

    if (nstages > 0) {
        if constexpr (C::GLDS) {
            dma_operand(Xb, ldx_b, m0, M, smem, C::TM, 0);
            dma_operand(Wb, ldw_b, n0, N, smem + C::X_STAGE, C::TN, 0);
        } else {
            load_regs(0);
            store_regs(0);
        }
    }
    __syncthreads();  // (GLDS: drains the DMA, vmcnt(0)) buffer 0 is ready

    const int lrow = lane & 31, lhalf = lane >> 5;
    const int scale_one = 0x7f7f7f7f;  // E8M0 127 = 2^0 for every 32-element block

    // optional phase trace (tuning only): block (0,0), lane 0 of every wave stamps s_memtime
    const bool tr = trace != nullptr && tile_m == 0 && tile_n == 0 && lane == 0;
    auto stamp = [&](int s, int phase) {
        if (tr && s < 8) trace[(wave * 8 + s) * 8 + phase] = __builtin_amdgcn_s_memtime();
    };
    if constexpr (C::PIPE == 1) {
        // ---- pipelined main loop (asm-issued DMA) ---------------------------------------------
        static_assert(C::GLDS, "PIPE needs the LDS-DMA staging");
        constexpr int KKP = STAGE_BYTES / 32;
        constexpr int XP = C::TM / 8 / C::NWAVES, WP = C::TN / 8 / C::NWAVES;  // DMA pieces per wave
        constexpr int NP = XP + WP;
        // PIPE contract (checked by the launcher): row strides are whole stages (ld % 32 words == 0,
        // pad nibbles zero) and each operand spans < 2^31 bytes.
        const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
        const int uwave = __builtin_amdgcn_readfirstlane(wave);
        const int p = lane & 7, rsub = lane >> 3;
        // loop-invariant per-lane byte offsets of this wave's DMA pieces (row clamped to the last
        // valid row: products of out-of-range rows are never stored); (row>>1)&7 is the same for
        // every piece of a lane because piece rows differ by multiples of 64.
        unsigned voffx[XP], voffw[WP];
#pragma unroll
        for (int j = 0; j < XP; ++j) {
            const int row = (j * C::NWAVES + uwave) * 8 + rsub;
            voffx[j] = (unsigned)(min(m0 + row, M - 1) * ldx_b) + (unsigned)((p ^ ((row >> 1) & 7)) * 16);
        }
#pragma unroll
        for (int j = 0; j < WP; ++j) {
            const int row = (j * C::NWAVES + uwave) * 8 + rsub;
            voffw[j] = (unsigned)(min(n0 + row, N - 1) * ldw_b) + (unsigned)((p ^ ((row >> 1) & 7)) * 16);
        }
        auto issue_piece = [&](int j, int s, int buf) {
            const unsigned ldsbuf = lds0 + buf * BUF;
            if (j < XP) {
                const unsigned dst = ldsbuf + ((j * C::NWAVES + uwave) * 8) * STAGE_BYTES;
                glds16_asm(Xb + (int64_t)s * STAGE_BYTES, voffx[j < XP ? j : 0], __builtin_amdgcn_readfirstlane(dst));
            } else {
                const int jw = j - XP;
                const unsigned dst = ldsbuf + C::X_STAGE + ((jw * C::NWAVES + uwave) * 8) * STAGE_BYTES;
                glds16_asm(Wb + (int64_t)s * STAGE_BYTES, voffw[jw >= 0 && jw < WP ? jw : 0], __builtin_amdgcn_readfirstlane(dst));
            }
        };
        auto read_frags = [&](const unsigned char* xs, const unsigned char* ws, int kk,
                              uint4 (&af)[C::TNW], uint4 (&bf)[C::TMW]) {
            const int c = kk * 2 + lhalf;
#pragma unroll
            for (int a = 0; a < C::TNW; ++a) {
                const int row = (wave_n * C::TNW + a) * 32 + lrow;
                af[a] = *reinterpret_cast<const uint4*>(ws + row * STAGE_BYTES + swz(row, c) * 16);
            }
#pragma unroll
            for (int b = 0; b < C::TMW; ++b) {
                const int row = (wave_m * C::TMW + b) * 32 + lrow;
                bf[b] = *reinterpret_cast<const uint4*>(xs + row * STAGE_BYTES + swz(row, c) * 16);
            }
        };
        auto mfma_step = [&](uint4 (&af)[C::TNW], uint4 (&bf)[C::TMW]) {
#pragma unroll
            for (int a = 0; a < C::TNW; ++a)
#pragma unroll
                for (int b = 0; b < C::TMW; ++b) {
                    const v8i av = (v8i){(int)af[a].x, (int)af[a].y, (int)af[a].z, (int)af[a].w, 0, 0, 0, 0};
                    const v8i bv = (v8i){(int)bf[b].x, (int)bf[b].y, (int)bf[b].z, (int)bf[b].w, 0, 0, 0, 0};
                    acc[a][b] = (C::EPI == 0)
                        ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[a][b], 4, 4, 0, scale_one, 0, scale_one)
                        : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bv, av, acc[a][b], 4, 4, 0, scale_one, 0, scale_one);
                }
        };
        // The steady-state body is branch-free (the last stage is peeled) so that it stays ONE
        // basic block: otherwise LLVM sinks the register-only MFMAs below the reads/DMA issues and
        // the sched_barriers (which only order within a block) cannot hold the interleave.
        auto stage_body = [&](int s, auto more_tag) {
            constexpr bool more = decltype(more_tag)::value;
            const int buf = s & 1;
            const unsigned char* xs = smem + buf * BUF;
            const unsigned char* ws = xs + C::X_STAGE;
            uint4 afA[C::TNW], bfA[C::TMW], afB[C::TNW], bfB[C::TMW];
            read_frags(xs, ws, 0, afA, bfA);
#pragma unroll
            for (int kk = 0; kk < KKP; kk += 2) {
                // DMA pieces are spread over the k-steps; fragments of step kk+1 are requested
                // before the MFMAs of step kk so LDS latency hides under the matrix pipe.
                if constexpr (more) {
#pragma unroll
                    for (int j = kk * NP / KKP; j < (kk + 1) * NP / KKP; ++j) issue_piece(j, s + 1, buf ^ 1);
                }
                read_frags(xs, ws, kk + 1, afB, bfB);
                __builtin_amdgcn_sched_barrier(0);  // pin: requests for step kk+1 precede MFMAs of kk
                mfma_step(afA, bfA);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (more) {
#pragma unroll
                    for (int j = (kk + 1) * NP / KKP; j < (kk + 2) * NP / KKP; ++j) issue_piece(j, s + 1, buf ^ 1);
                }
                if (kk + 2 < KKP) read_frags(xs, ws, kk + 2, afA, bfA);
                __builtin_amdgcn_sched_barrier(0);
                mfma_step(afB, bfB);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces have landed
            __syncthreads();                                   // ... and everyone's are visible
        };
        for (int s = 0; s + 1 < nstages; ++s) stage_body(s, std::true_type{});
        if (nstages > 0) stage_body(nstages - 1, std::false_type{});
    } else {
    constexpr int KK = STAGE_BYTES / 32;  // MFMA k-steps (64 elements) per stage
    for (int s = 0; s < (C::ABL == 3 ? 0 : nstages); ++s) {
        const int buf = s & 1;
        const unsigned char* xs = smem + buf * BUF;
        const unsigned char* ws = xs + C::X_STAGE;
        if constexpr (C::GLDS) {
            // (1) pull every fragment of this stage into registers, (2) start the DMA of the next
            // stage into the other buffer, (3) run the register-only MFMAs while it is in flight.
            // hipcc drains vmcnt(0) in front of any ds_read that follows an LDS-DMA issue (it cannot
            // prove they do not alias), so no LDS read may sit between (2) and the barrier.
            uint4 af[KK][C::TNW], bf[KK][C::TMW];
            stamp(s, 0);
            if (C::ABL != 1) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const int c = kk * 2 + lhalf;
#pragma unroll
                    for (int a = 0; a < C::TNW; ++a) {
                        const int row = (wave_n * C::TNW + a) * 32 + lrow;
                        af[kk][a] = *reinterpret_cast<const uint4*>(ws + row * STAGE_BYTES + swz(row, c) * 16);
                    }
#pragma unroll
                    for (int b = 0; b < C::TMW; ++b) {
                        const int row = (wave_m * C::TMW + b) * 32 + lrow;
                        bf[kk][b] = *reinterpret_cast<const uint4*>(xs + row * STAGE_BYTES + swz(row, c) * 16);
                    }
                }
            }
            stamp(s, 1);
            if (s + 1 < nstages && C::ABL != 2) {
                dma_operand(Xb, ldx_b, m0, M, smem + (buf ^ 1) * BUF, C::TM, s + 1);
                dma_operand(Wb, ldw_b, n0, N, smem + (buf ^ 1) * BUF + C::X_STAGE, C::TN, s + 1);
            }
            stamp(s, 2);
            if (tr) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(s, 3); }
            if (C::ABL != 1) {
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                    for (int a = 0; a < C::TNW; ++a)
#pragma unroll
                        for (int b = 0; b < C::TMW; ++b) {
                            const v8i av = (v8i){(int)af[kk][a].x, (int)af[kk][a].y, (int)af[kk][a].z, (int)af[kk][a].w, 0, 0, 0, 0};
                            const v8i bv = (v8i){(int)bf[kk][b].x, (int)bf[kk][b].y, (int)bf[kk][b].z, (int)bf[kk][b].w, 0, 0, 0, 0};
                            acc[a][b] = (C::EPI == 0)
                                ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[a][b], 4, 4, 0, scale_one, 0, scale_one)
                                : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bv, av, acc[a][b], 4, 4, 0, scale_one, 0, scale_one);
                        }
            }
            // keep the MFMAs on this side of the barrier: they are register-only, so without this
            // hipcc sinks them below s_barrier and its vmcnt(0) then waits on the DMA right after
            // it was issued (no overlap at all).
            __builtin_amdgcn_sched_barrier(0);
            stamp(s, 4);
            if (tr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(s, 5); }
        } else {
            if (s + 1 < nstages && C::ABL != 2) load_regs(s + 1);
#pragma unroll
            for (int kk = 0; kk < (C::ABL == 1 ? 0 : KK); ++kk) {
                const int c = kk * 2 + lhalf;
                v8i afrag[C::TNW], bfrag[C::TMW];
#pragma unroll
                for (int a = 0; a < C::TNW; ++a) {
                    const int row = (wave_n * C::TNW + a) * 32 + lrow;
                    const uint4 v = *reinterpret_cast<const uint4*>(ws + row * STAGE_BYTES + swz(row, c) * 16);
                    afrag[a] = (v8i){(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
                }
#pragma unroll
                for (int b = 0; b < C::TMW; ++b) {
                    const int row = (wave_m * C::TMW + b) * 32 + lrow;
                    const uint4 v = *reinterpret_cast<const uint4*>(xs + row * STAGE_BYTES + swz(row, c) * 16);
                    bfrag[b] = (v8i){(int)v.x, (int)v.y, (int)v.z, (int)v.w, 0, 0, 0, 0};
                }
#pragma unroll
                for (int a = 0; a < C::TNW; ++a)
#pragma unroll
                    for (int b = 0; b < C::TMW; ++b)
                        acc[a][b] = (C::EPI == 0)
                            ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                                  afrag[a], bfrag[b], acc[a][b], 4, 4, 0, scale_one, 0, scale_one)
                            : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
                                  bfrag[b], afrag[a], acc[a][b], 4, 4, 0, scale_one, 0, scale_one);
            }
            if (s + 1 < nstages && C::ABL != 2) store_regs(buf ^ 1);
        }
        __syncthreads();  // all reads of `buf` done; next stage landed (vmcnt(0)) and visible
        stamp(s, 6);
    }
    }  // PIPE == 0

    if constexpr (C::EPI == 1) {
        // D[row = m][col = n]: lane owns column n = nb + lrow, rows m = mb + (r&3) + 8*(r>>2) + 4*lhalf
#pragma unroll
        for (int a = 0; a < C::TNW; ++a) {
            const int n = n0 + (wave_n * C::TNW + a) * 32 + lrow;
            const float bv = (bias && n < N) ? bias[n] : 0.0f;
#pragma unroll
            for (int b = 0; b < C::TMW; ++b) {
                const int mb = m0 + (wave_m * C::TMW + b) * 32 + 4 * lhalf;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m < M && n < N) Y[(int64_t)m * ldy + n] = acc[a][b][r] + bv;
                }
            }
        }
        return;
    }
    // epilogue: D[row = n][col = m]; lane holds, per (a, b, q): n = nb + 8q + 4*lhalf + {0..3}
#pragma unroll
    for (int a = 0; a < C::TNW; ++a) {
        const int nb = n0 + (wave_n * C::TNW + a) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = nb + 8 * q + 4 * lhalf;
            float b4[4] = {0.f, 0.f, 0.f, 0.f};
            if (bias) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) b4[e] = bias[n + e];
            }
#pragma unroll
            for (int b = 0; b < C::TMW; ++b) {
                const int m = m0 + (wave_m * C::TMW + b) * 32 + lrow;
                if (m >= M) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[a][b][4 * q + e] + b4[e];
                float* yp = Y + (int64_t)m * ldy + n;
                if (vec_store && n + 3 < N) {
                    *reinterpret_cast<float4*>(yp) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (n + e < N) yp[e] = v[e];
                }
            }
        }
    }
}

template <class C>
int launch_nib_gemm(const uint32_t* Xn, int64_t ldxp, const uint32_t* Wn, int64_t ldwp,
                    const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K,
                    qt_stream_t stream, unsigned long long* trace = nullptr) {
    const int64_t gy = (M + C::TM - 1) / C::TM, gx = (N + C::TN - 1) / C::TN;
    if (gy > 65535) return QT_ERR_UNSUPPORTED;
    const int vec_store = qt_aligned16(Y) && (ldy % 4 == 0);
    // > 64 KiB of dynamic LDS needs the opt-in attribute (per device; cheap, so set every call)
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(nib_gemm_kernel<C>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess)
        return QT_ERR_LAUNCH;
    hipLaunchKernelGGL(nib_gemm_kernel<C>, dim3((unsigned)gx, (unsigned)gy), dim3(C::NTHREADS),
                       C::LDS_BYTES, (hipStream_t)stream, Xn, ldxp, Wn, ldwp, bias, Y, ldy, (int)M,
                       (int)N, (int)K, vec_store, trace);
    return qt_check_launch();
}

using CfgA = NibCfg<2, 4, 4, 2, true>;    // 8 waves, 256x256, LDS-DMA
using CfgB = NibCfg<2, 2, 4, 4, false>;   // 4 waves (1/SIMD), 256x256, register staged
using CfgC = NibCfg<2, 2, 2, 2, false>;   // 4 waves, 128x128, register staged
using CfgD = NibCfg<2, 4, 4, 2, false>;   // 8 waves, 256x256, register staged (VGPR-tight)
using CfgE = NibCfg<2, 2, 2, 2, true>;    // 4 waves, 128x128, LDS-DMA
using CfgA1 = NibCfg<2, 4, 4, 2, true, 1>;  // ablations of CfgA
using CfgA2 = NibCfg<2, 4, 4, 2, true, 2>;
using CfgA3 = NibCfg<2, 4, 4, 2, true, 3>;
using CfgF = NibCfg<2, 4, 4, 2, true, 0, 1>;   // CfgA with the line-coalesced dword epilogue
using CfgF3 = NibCfg<2, 4, 4, 2, true, 3, 1>;
using CfgP = NibCfg<2, 4, 4, 2, true, 0, 1, 1>;   // CfgF + asm-issued, interleaved DMA
using CfgF1 = NibCfg<2, 4, 4, 2, true, 1, 1>;
using CfgF2 = NibCfg<2, 4, 4, 2, true, 2, 1>;

// ---- fp32 -> nibble plane ------------------------------------------------------------------------
struct NibSign {  // safeSign: +1 -> 0x2, -1 -> 0xA
    __device__ __forceinline__ static uint32_t nib(float x) { return x < 0.0f ? 0xAu : 0x2u; }
};
struct NibTernary {  // TernaryConnectDeterministic: 0 -> 0x0
    __device__ __forceinline__ static uint32_t nib(float x) {
        const float t = qt_ternarize(x);
        return t == 0.0f ? 0x0u : (t < 0.0f ? 0xAu : 0x2u);
    }
};

// One work item = one float4 slot of the padded row (ldp words = ldp*2 slots); two adjacent lanes
// form one output word.  Slots past K/4 produce zero nibbles, so the pad-is-zero invariant holds.
template <class Enc>
__global__ __launch_bounds__(256) void nib_pack_vec_kernel(const float* __restrict__ x, int64_t ldx,
                                                           uint32_t* __restrict__ out, int64_t ldp,
                                                           int64_t rows, int64_t K) {
    const int64_t slots_per_row = ldp * 2;
    const int64_t total = rows * slots_per_row;  // even
    const int64_t k4 = K / 4;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < total;
         s += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = s / slots_per_row, slot = s - row * slots_per_row;
        uint32_t h = 0;
        if (slot < k4) {
            const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + slot * 4);
            h = Enc::nib(v.x) | (Enc::nib(v.y) << 4) | (Enc::nib(v.z) << 8) | (Enc::nib(v.w) << 12);
        }
        const uint32_t other = __shfl_xor(h, 1);
        if ((threadIdx.x & 1) == 0) out[row * ldp + (slot >> 1)] = h | (other << 16);
    }
}

// Generic path (any K / alignment): one thread per output word, scalar loads.
template <class Enc>
__global__ __launch_bounds__(256) void nib_pack_scalar_kernel(const float* __restrict__ x, int64_t ldx,
                                                              uint32_t* __restrict__ out, int64_t ldp,
                                                              int64_t rows, int64_t K) {
    const int64_t total = rows * ldp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / ldp, w = i - row * ldp;
        uint32_t word = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int64_t k = w * 8 + e;
            if (k < K) word |= Enc::nib(x[row * ldx + k]) << (4 * e);
        }
        out[i] = word;
    }
}

// ---- bit planes -> nibble plane (derived MFMA operand format) ---------------------------------------
// spread the 8 bits of a byte to bit 0 of 8 nibbles
__device__ __forceinline__ uint32_t spread8(uint32_t b) {
    uint32_t t = b & 0xFFu;
    t = (t | (t << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return t;
}
// One thread = one 32-bit word of the planes -> four nibble words (16 B).  sign-only planes:
// nibble = 0x2 | s<<3.  mask+sign: nibble = m<<1 | (s&m)<<3.  Words past the bit planes' stride are 0.
__global__ __launch_bounds__(256) void bits_to_nib_kernel(const uint32_t* __restrict__ sign,
                                                          const uint32_t* __restrict__ mask,
                                                          int64_t ldb, uint32_t* __restrict__ out,
                                                          int64_t ldn, int64_t rows, int64_t K) {
    const int64_t groups_per_row = ldn / 4;  // one group = 4 nibble words = 32 elements
    const int64_t total = rows * groups_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / groups_per_row, g = i - row * groups_per_row;
        uint32_t sw = 0, mw = 0;
        if (g < ldb) {
            sw = sign[row * ldb + g];
            if (mask) {
                mw = mask[row * ldb + g];
            } else {  // binary: every element inside K is non-zero
                const int64_t rem = K - g * 32;
                mw = rem >= 32 ? 0xFFFFFFFFu : (rem > 0 ? ((1u << rem) - 1u) : 0u);
            }
        }
        sw &= mw;
        uint4 o;
        o.x = (spread8(mw) << 1) | (spread8(sw) << 3);
        o.y = (spread8(mw >> 8) << 1) | (spread8(sw >> 8) << 3);
        o.z = (spread8(mw >> 16) << 1) | (spread8(sw >> 16) << 3);
        o.w = (spread8(mw >> 24) << 1) | (spread8(sw >> 24) << 3);
        *reinterpret_cast<uint4*>(out + row * ldn + g * 4) = o;
    }
}

template <class Enc>
int launch_nib_pack(const float* x, int64_t ldx, uint32_t* out, int64_t ldp, int64_t rows, int64_t K,
                    qt_stream_t stream) {
    if (rows < 0 || K < 0 || ldx < K) return QT_ERR_INVALID_ARG;
    if (rows == 0) return QT_OK;
    if (!out || (!x && K > 0)) return QT_ERR_INVALID_ARG;
    const int64_t kw = (K + 7) / 8;
    if (ldp < kw || (ldp & 3) != 0 || !qt_aligned16(out)) return QT_ERR_ALIGNMENT;
    if (ldp == 0) return QT_OK;
    const bool vec = (K % 4 == 0) && (ldx % 4 == 0) && qt_aligned16(x);
    if (vec) {
        const int grid = qt_stream_grid((rows * ldp * 2 + 255) / 256);
        hipLaunchKernelGGL((nib_pack_vec_kernel<Enc>), dim3(grid), dim3(256), 0, (hipStream_t)stream, x,
                           ldx, out, ldp, rows, K);
    } else {
        const int grid = qt_stream_grid((rows * ldp + 255) / 256);
        hipLaunchKernelGGL((nib_pack_scalar_kernel<Enc>), dim3(grid), dim3(256), 0, (hipStream_t)stream,
                           x, ldx, out, ldp, rows, K);
    }
    return qt_check_launch();
}

}  // namespace

extern "C" {

int qt_sign_pack_nib_f32(const float* x, int64_t ldx, uint32_t* nib_plane, int64_t ldp, int64_t rows,
                         int64_t K, qt_stream_t stream) {
    return launch_nib_pack<NibSign>(x, ldx, nib_plane, ldp, rows, K, stream);
}

int qt_ternary_pack_nib_f32(const float* x, int64_t ldx, uint32_t* nib_plane, int64_t ldp,
                            int64_t rows, int64_t K, qt_stream_t stream) {
    return launch_nib_pack<NibTernary>(x, ldx, nib_plane, ldp, rows, K, stream);
}

int qt_nib_gemm_variant(int variant, const uint32_t* Xn, int64_t ldxp, const uint32_t* Wn,
                        int64_t ldwp, const float* bias, float* Y, int64_t ldy, int64_t M, int64_t N,
                        int64_t K, qt_stream_t stream) {
    if (M < 0 || N < 0 || K < 0) return QT_ERR_INVALID_ARG;
    if (M == 0 || N == 0) return QT_OK;
    if (!Y || ldy < N) return QT_ERR_INVALID_ARG;
    if (M > INT32_MAX || N > INT32_MAX || K >= (1 << 24)) return QT_ERR_UNSUPPORTED;  // fp32-exact bound
    const int64_t kw = (K + 7) / 8;
    if (K > 0 && (!Xn || !Wn)) return QT_ERR_INVALID_ARG;
    if (ldxp < kw || ldwp < kw) return QT_ERR_INVALID_ARG;
    if ((ldxp & 3) || (ldwp & 3)) return QT_ERR_ALIGNMENT;
    if (K > 0 && (!qt_aligned16(Xn) || !qt_aligned16(Wn))) return QT_ERR_ALIGNMENT;
    switch (variant) {
        case 0: return launch_nib_gemm<CfgA>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        case 1: return launch_nib_gemm<CfgB>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        case 2: return launch_nib_gemm<CfgC>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        case 3: return launch_nib_gemm<CfgD>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        case 4: return launch_nib_gemm<CfgE>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        case 5: return launch_nib_gemm<CfgF>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        case 6:
            if ((ldxp & 31) || (ldwp & 31) || M * ldxp * 4 >= (1ll << 31) || N * ldwp * 4 >= (1ll << 31))
                return QT_ERR_ALIGNMENT;
            return launch_nib_gemm<CfgP>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        case 205:  // CfgF with the phase trace: `bias` carries the trace buffer (8 waves x 8 stages x 8 u64)
            return launch_nib_gemm<CfgF>(Xn, ldxp, Wn, ldwp, nullptr, Y, ldy, M, N, K, stream,
                                         reinterpret_cast<unsigned long long*>(const_cast<float*>(bias)));
        case 106: return launch_nib_gemm<CfgF1>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        case 107: return launch_nib_gemm<CfgF2>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        case 105: return launch_nib_gemm<CfgF3>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        case 101: return launch_nib_gemm<CfgA1>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        case 102: return launch_nib_gemm<CfgA2>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        case 103: return launch_nib_gemm<CfgA3>(Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
        default: return QT_ERR_UNSUPPORTED;
    }
}

int qt_nib_gemm(const uint32_t* Xn, int64_t ldxp, const uint32_t* Wn, int64_t ldwp, const float* bias,
                float* Y, int64_t ldy, int64_t M, int64_t N, int64_t K, qt_stream_t stream) {
    // fast path: pipelined asm-DMA kernel; needs whole-stage row strides and < 2 GiB operands.
    const bool pipe_ok = !(ldxp & 31) && !(ldwp & 31) && M * ldxp * 4 < (1ll << 31) &&
                         N * ldwp * 4 < (1ll << 31);
    return qt_nib_gemm_variant(pipe_ok ? 6 : 5, Xn, ldxp, Wn, ldwp, bias, Y, ldy, M, N, K, stream);
}

int qt_bits_to_nib(const uint32_t* sign_plane, const uint32_t* mask_plane, int64_t ldb,
                   uint32_t* nib_plane, int64_t ldn, int64_t rows, int64_t K, qt_stream_t stream) {
    if (rows < 0 || K < 0) return QT_ERR_INVALID_ARG;
    if (rows == 0) return QT_OK;
    if (!nib_plane || (K > 0 && !sign_plane)) return QT_ERR_INVALID_ARG;
    if (ldb < (K + 31) / 32 || ldn < (K + 7) / 8) return QT_ERR_INVALID_ARG;
    if ((ldb & 3) || (ldn & 3) || !qt_aligned16(nib_plane)) return QT_ERR_ALIGNMENT;
    const int grid = qt_stream_grid((rows * (ldn / 4) + 255) / 256);
    hipLaunchKernelGGL(bits_to_nib_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, sign_plane,
                       mask_plane, ldb, nib_plane, ldn, rows, K);
    return qt_check_launch();
}

}  // extern "C"
