// Device helpers shared by the matrix-core kernels (mfma_gemm.hip; tools/experiments/linear_fused.hip): vector types, the LDS chunk
// swizzle of a K stage and the inline-asm LDS-DMA forms.  Everything is static / inline: each translation unit
// gets its own copy.
#pragma once
#include "qt_common.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v16f __attribute__((ext_vector_type(16)));

// K bytes per row per stage (SB): 128 (4 MFMA k-steps) or 64 (2 k-steps).

// chunk swizzle: SB = 128 -> 8 chunks/row, two rows per 256-B bank row: c ^ ((r>>1)&7);
//                SB = 64  -> 4 chunks/row, four rows per bank row:      c ^ ((r>>2)&3).
// Either way 16 rows distinct mod 16 land on 16 distinct 16-byte slots.
template <int SB>
__device__ __forceinline__ int swz(int row, int chunk) {
    // 512-byte stages (4-wave 64x64 tiles): only row bits 0..2, so that the pieces of one lane (rows 8 apart) share
    // their logical chunk (the conv tap table is looked up once per stage), and 8 consecutive rows — one 128-byte LDS
    // cycle of a ds_read_b128 — still land on 8 different chunk positions
    return SB == 512 ? (chunk ^ (row & 7)) : SB == 256 ? (chunk ^ (row & 15)) : SB == 128 ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((row >> 2) & 3));
}

// One LDS-DMA piece (64 lanes x 16 B -> 1 KiB of LDS at the wave-uniform byte address lds_dst), issued
// from inline asm.  Address form: 64-bit SGPR base + 32-bit per-lane VGPR byte offset.  M0 = LDS byte
// address; M0 is compiler-reserved, so it is saved/restored inside the same statement; the s_nop covers
// the SALU-write-M0 -> LDS-DMA hazard.  No VGPR destination, so the statement is register-safe; the DATA
// is ordered for readers only by our own s_waitcnt vmcnt(0) + barrier at the end of a stage.
__device__ __forceinline__ void glds16_asm(const unsigned char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(sbase), "s"(lds_dst)
        : "memory");
}

// Same, with a full 64-bit per-lane source address (implicit-GEMM conv: the source of a chunk is a
// pixel of the NHWC plane or the zero page, so there is no common base).
__device__ __forceinline__ void glds16_asm64(const unsigned char* src, unsigned lds_dst) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(lds_dst)
        : "memory");
}

// profiling builds only (ABL == 5): shader-cycle stamp that neither the compiler nor a branch can move
__device__ __forceinline__ unsigned long long stamp_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// Lean forms for the ping-pong load segment, where the wave's instruction count IS the segment length (an
// in-order wave issues one instruction per ~6-10 cycles beside its partner's MFMA stream): M0 is written
// directly by the SALU add and NOT restored per piece — the caller brackets the whole run of pieces with
// m0_save() / m0_restore().
__device__ __forceinline__ unsigned m0_save() {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep)::"memory");
    return keep;
}
__device__ __forceinline__ void m0_restore(unsigned keep) { asm volatile("s_mov_b32 m0, %0" ::"s"(keep) : "memory"); }
__device__ __forceinline__ void glds16_lean(const unsigned char* sbase, unsigned voff, unsigned lds_a, unsigned lds_b) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_a), "s"(lds_b) : "memory", "scc");
}
__device__ __forceinline__ void glds16_lean64(const unsigned char* src, unsigned lds_a, unsigned lds_b) {
    asm volatile("s_add_u32 m0, %1, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(src), "s"(lds_a), "s"(lds_b) : "memory", "scc");
}


}  // namespace
