// LinearBin / LinearTer training-mode forward in ONE launch: fp32 activations (+-1) and fp32 weights in, fp32 Y out
// (layers/binary_layers.py:44  F.linear(x, bin_op(W), b);  layers/terner_layers.py:49).
//
// Why.  As three kernels (sign+pack both operands -> nibble planes, then the MX-fp4 MFMA GEMM) the 4096^3 step is
// pack 24.8 us (HBM-bound, matrix pipe idle) + GEMM 37.5 us (matrix-bound, HBM idle during its 22 us main loop).
// Here the SAME 256 workgroups (one per CU, all co-resident) do both jobs, overlapped in time:
//
//   fill     : every workgroup packs its share of the first LEAD-1 K-chunks (512 columns each) of its X row panel
//              and its W row panel at full HBM rate, write-through (sc1) into the nibble workspace, and signals one
//              counter per (panel, chunk);
//   main loop: the ping-pong fp4 MFMA loop of mfma_gemm.hip (64-byte stages, ring of 4, LDS-DMA three stages ahead);
//              in its LOAD segments each wave additionally streams one fp32 row-chunk (2 KiB) per stage from HBM into
//              registers, converts the one it loaded three stages ago to nibbles (8 per lane), collects four rows in a
//              wave-private LDS patch and writes the 1 KiB of nibbles of chunk c + LEAD with one dwordx4 sc1 store;
//              before a wave DMAs the first stage of chunk c it has checked (poll issued three stages earlier) that
//              the 16 workgroups sharing its X panel and the 16 sharing its W panel have all signalled chunk c;
//   epilogue : LDS-transposed dwordx4 stores of the fp32 tile (as mfma_gemm.hip).
//
// Work split: tile (i, j) of the 16 x 16 tile grid packs rows [16 j, 16 j + 16) of X panel i and rows [16 i, 16 i + 16)
// of W panel j; waves 0-3 take the X rows, waves 4-7 the W rows, four rows each, one row per stage.
//
// K order.  The contraction is a sum over k, so any permutation of k applied to BOTH operands gives the same
// integers.  The workspace is therefore NOT the canonical nibble plane of qt_hip.h: within a 512-column chunk, lane l
// of the packing wave holds columns 4l..4l+3 and 256+4l..256+4l+3 (two coalesced dwordx4 loads) and they become the 8
// nibbles of dword l of the row's 256-byte chunk.  X and W use the same map, nothing else reads the workspace.
//
// vmcnt discipline.  Every VMEM instruction inside the loop is issued from inline asm in a fixed per-stage order
// [nibble store] [signal] [2 polls] [2 fp32 loads] [4 DMA pieces]; the single wait at the end of a load segment is
// vmcnt(ops(previous stage) + ops(this stage)), a compile-time constant per stage position, which guarantees that
// everything issued two or more stages ago has completed: the DMA of the next stage, the fp32 loads converted in the
// next stage, the store whose signal goes out in the next stage and the poll that is checked in the next stage.
//
// STATUS (round 2, MI355X, 4096^3): EXPERIMENT, not part of libqt_hip.so (built by tools/experiments/lf.py).
//  * check_linear_fused.py / stamps_fused.py: bit-identical to the two-launch route on an otherwise idle device, but
//    SLOWER: 76 us per call against 60 us for qt_pack_pair_nib_f32 + qt_nib_gemm.
//  * stress_linear_fused.py: with another stream's kernel occupying CUs (workgroups start staggered, consumers already
//    spinning when a signal lands) EVERY launch has a stale (panel, chunk) hand-off with error word 0; -DLF_FIX=1 (agent
//    release before each signal) -> 4 / 300 launches stale at 436 us per call, -DLF_FIX=2 (agent acquire after the check)
//    -> no change.  The write-through store + counted vmcnt + atomic flag form below is therefore NOT a valid hand-off
//    under uneven load, and the entry point was removed from the library.  Where the time goes (median workgroup, us since launch):
// fill done 15.0 (slowest 20.3: the K-major column-panel order reads HBM at 4.7 TB/s and unevenly, the row-streaming
// pack kernel gets 6.1) -> chunk 0 complete from all 16 partners 20.8 -> loop start 24.1 -> 12 stages with packing done
// 45.4 (1.75 us per stage) -> 20 drain stages done 62.5 (0.85 us per stage; the plain GEMM loop runs 0.57-0.68) ->
// stores drained 68.3.  Ablations on the same build: without the in-loop store / signal / poll operations the loop takes
// 24 us instead of 38 (those are ~1-2 us fabric round trips and vmcnt retires IN ORDER: a slow operation delays the
// completion count of every LDS-DMA piece issued behind it, and a piece has only two stages = 1.1 us of slack); with
// them removed AND L2-hot sources the 12 packing stages still cost 12.3 us against 6.9 for plain stages: the 25 VALU
// of the conversion sit in the load segment beside the partner wave's MFMA stream at ~12 cycles each.  Even with free
// communication the launch would take ~54 us (an un-overlapped fill of 15-20 us + hand-off + 24 us loop + 6 us store
// tail), i.e. the overlap this design can reach is worth at most ~6 us over two launches.  Kept as a tested reference
// for the hand-off protocol and as the measured answer to "why not one persistent launch" (DESIGN.md section 4).
//
// Cross-workgroup visibility (MI355X_MICROARCH.md, "inter-workgroup visibility"): payload = 16-byte sc1 stores
// (write-through), read by sc1 LDS-DMA loads; flag = device-scope atomic add issued only after the store is known
// complete (counted vmcnt), polled with sc1 loads.  Every spin is bounded; a timeout raises the error word of the
// sync area and the launch terminates with garbage instead of hanging.
#include <type_traits>
#include "qt_common.h"
#include "pp_common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef v4u __attribute__((may_alias)) v4u_alias;

constexpr int LF_TM = 256, LF_TN = 256, LF_SB = 64, LF_NBUF = 4;
constexpr int LF_XSTAGE = LF_TM * LF_SB, LF_BUF = (LF_TM + LF_TN) * LF_SB;
constexpr int LF_NWAVES = 8, LF_NTHREADS = 512;
constexpr int LF_RPP = 16, LF_CH = 4;                    // rows per DMA piece, 16-byte chunks per stage row
constexpr int LF_STG_WORDS = 512;                        // per wave: two 1 KiB nibble patches (chunk parity)
constexpr int LF_LDS = LF_NBUF * LF_BUF + 64 + LF_NWAVES * LF_STG_WORDS * 4;
constexpr int LF_KC = 512;                               // columns per chunk = 4 stages
// sync area (uint32 words): epoch, done, error, pad, then two counter sets (used alternately, the idle one is zeroed
// by the running launch): [operand X/W][panel 16][chunk 128]
constexpr int LF_SYNC_EPOCH = 0, LF_SYNC_DONE = 1, LF_SYNC_ERR = 2, LF_SYNC_SETS = 4;
constexpr int LF_MAXCH = 128, LF_W_OFF = 16 * LF_MAXCH, LF_SET_WORDS = 2 * 16 * LF_MAXCH;
constexpr int LF_SYNC_BYTES = (LF_SYNC_SETS + 2 * LF_SET_WORDS) * 4;
constexpr int LF_SYNC_RESERVED = 36864;                  // sync area rounded up; the nibble planes follow (4 KiB aligned)
constexpr unsigned LF_ARRIVALS = 64;                     // 16 workgroups x 4 waves per (panel, chunk)
constexpr int LF_SPIN_MAX = 1 << 18;

// Encoders.  A lane turns its 8 floats into 8 predicate bits with v_cmp + v_addc_co (w = 2w + bit: two VALU per
// element and no constant registers — the select/or formulation needs 2.5 plus eight shifted constants that the
// register allocator re-materialises every stage), then expands the byte to 8 nibbles with a byte permute.
// Element e (0..7: a.x..a.w, b.x..b.w) lands in bit 7-e, i.e. nibble 7-e of the lane's dword; X and W share the map.
struct FSign {     // safeSign: x < 0 -> -1 (0xA), else +1 (0x2)   functions/common.py:4-7
    static constexpr bool TERNARY = false;
};
struct FTernary {  // TernaryConnectDeterministic, functions/terner_connect.py:26-27: x >= 0.5 -> +1, x < -0.5 -> -1, NaN -> +1
    static constexpr bool TERNARY = true;
};
// w = 2w + (x < 0)
__device__ __forceinline__ void bit_neg(unsigned& w, float x) {
    asm("v_cmp_gt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(x) : "vcc");
}
// w = 2w + (x < thr)  /  w = 2w + !(x < thr), thr wave-uniform (an SGPR): the ternary launch runs the SAME instruction
// stream on the activation waves (thr = 0: safeSign) and on the weight waves (thr = -0.5 / +0.5), no branch
__device__ __forceinline__ void bit_lt(unsigned& w, float x, float thr) {
    asm("v_cmp_gt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(x), "s"(thr) : "vcc");
}
__device__ __forceinline__ void bit_nlt(unsigned& w, float x, float thr) {
    asm("v_cmp_ngt_f32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(w) : "v"(x), "s"(thr) : "vcc");
}
template <void (*F)(unsigned&, float, float)>
__device__ __forceinline__ unsigned bits8t(const v4f& a, const v4f& b, float thr) {
    unsigned w = 0;
    F(w, a.x, thr); F(w, a.y, thr); F(w, a.z, thr); F(w, a.w, thr); F(w, b.x, thr); F(w, b.y, thr); F(w, b.z, thr); F(w, b.w, thr);
    return w;
}
template <void (*F)(unsigned&, float)>
__device__ __forceinline__ unsigned bits8(const v4f& a, const v4f& b) {
    unsigned w = 0;
    F(w, a.x); F(w, a.y); F(w, a.z); F(w, a.w); F(w, b.x); F(w, b.y); F(w, b.z); F(w, b.w);
    return w;
}
// 8 predicate bits -> 8 nibbles without constants or an LDS table: bit pair j of w selects byte j of the result out of a
// 4-entry byte table held in one register (v_perm_b32): 3 VALU for the selectors + 1 permute.
__device__ __forceinline__ uint32_t pair_select(uint32_t w, uint32_t table) {
    uint32_t t = w | (w << 6);
    t = t | (t << 12);
    return __builtin_amdgcn_perm(0u, table, t & 0x03030303u);
}
constexpr uint32_t LF_TBL_SIGN = 0xAAA22A22u;     // bit = 1 -> 0xA (-1), bit = 0 -> 0x2 (+1), two elements per byte
constexpr uint32_t LF_TBL_SPREAD = 0x11100100u;   // bit -> bit 0 of its nibble

// LDS-DMA piece with sc1: the source was written by another CU's write-through stores in this launch (a plain
// load measured stale lines, tools/check_linear_fused.py; sc1 measured no slower)
__device__ __forceinline__ void dma16_sc1(const unsigned char* sbase, unsigned voff, unsigned lds_a, unsigned lds_b) {
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1"
                 :: "v"(voff), "s"(sbase), "s"(lds_a), "s"(lds_b) : "memory", "scc");
}

__device__ __forceinline__ v16f mfma_fp4(const v4u& a, const v4u& b, v16f c) {
    const v8i av = (v8i){(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
    const v8i bv = (v8i){(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// per-stage configuration (all compile-time): position in the 4-stage chunk period, the vmcnt immediate of the
// end-of-load wait, and which optional operations the stage carries
template <int P_, int NEND_, bool LOADS_, bool CONV_, bool STORE_, bool SIG_, bool CHECK_, bool POLL_, bool ISSUE_, bool LAST_>
struct StageCfg {
    static constexpr int P = P_, NEND = NEND_;
    static constexpr bool LOADS = LOADS_, CONV = CONV_, STORE = STORE_, SIG = SIG_, CHECK = CHECK_, POLL = POLL_,
                          ISSUE = ISSUE_, LAST = LAST_;
};
// steady state (phase 1): ops per stage by position: 6, 7 (signal), 9 (store + 2 polls), 6
template <int P, int NEND> using Steady = StageCfg<P, NEND, true, true, P == 2, P == 1, P == 1, P == 2, true, false>;
// drain (phase 2), period q after the last fp32 load: converts for three more stages, the last store, two more
// signals, a check per period, polls while a later chunk exists
template <int LEAD, int Q, int P>
struct Drain {
    static constexpr bool conv = 4 * Q + P < 3, store = 4 * Q + P == 2, sig = P == 1 && Q <= 1, check = P == 1,
                          poll = P == 2 && Q <= LEAD - 3;
    static constexpr int ops = 4 + (sig ? 1 : 0) + (store ? 1 : 0) + (poll ? 2 : 0);
};
template <int LEAD, int Q, int P> constexpr int drain_prev_ops() {
    if constexpr (P > 0) return Drain<LEAD, Q, P - 1>::ops;
    else if constexpr (Q == 0) return 6;                      // last steady stage (position 3)
    else return Drain<LEAD, Q - 1, 3>::ops;
}
template <int LEAD, int Q, int P>
using DrainCfg = StageCfg<P, drain_prev_ops<LEAD, Q, P>() + Drain<LEAD, Q, P>::ops, false, Drain<LEAD, Q, P>::conv,
                          Drain<LEAD, Q, P>::store, Drain<LEAD, Q, P>::sig, Drain<LEAD, Q, P>::check,
                          Drain<LEAD, Q, P>::poll, true, false>;

template <class EncW, int LEAD, int DBG = 0>
__global__ __launch_bounds__(LF_NTHREADS, 2) void linear_fused_kernel(
    const float* __restrict__ X, int64_t ldx, const float* __restrict__ W, int64_t ldw,
    const float* __restrict__ bias, float* __restrict__ Y, int64_t ldy, unsigned char* __restrict__ Xn,
    unsigned char* __restrict__ Wn, uint32_t* __restrict__ sync, int M, int N, int K) {
    static_assert(LEAD >= 3, "a chunk's signal must precede its check in program order (see the deadlock note)");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int BUF = LF_BUF, SB = LF_SB, TMW = 4, TNW = 2, WN = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_n = wave % WN, wave_m = wave / WN;
    const int lrow = lane & 31, lhalf = lane >> 5;

    // XCD-aware tile order, 16 x 16 tile grid (mfma_gemm.hip): XCD b % 8 owns a 4 x 8 super-tile
    int tile_m, tile_n;
    {
        const int gx = N >> 8, per_xcd = ((N >> 8) * (M >> 8) + 7) >> 3;
        const int o = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
        const int st = o >> 5, in_st = o & 31, sgx = gx >> 3;
        tile_m = (st / sgx) * 4 + (in_st >> 3);
        tile_n = (st % sgx) * 8 + (in_st & 7);
    }
    const int m0 = tile_m * LF_TM, n0 = tile_n * LF_TN;
    const int nch = K / LF_KC, nstages = nch * 4;
    const int rowb = K >> 1;                                  // nibble row stride in bytes

    unsigned long long dbg_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if constexpr (DBG & 8) dbg_t[0] = wall_clock64();
    const uint32_t epoch = sync[LF_SYNC_EPOCH];               // bumped by the last workgroup of the previous launch
    uint32_t* cnt = sync + LF_SYNC_SETS + (epoch & 1u) * LF_SET_WORDS;
    {
        uint32_t* other = sync + LF_SYNC_SETS + ((epoch & 1u) ^ 1u) * LF_SET_WORDS;
        for (int i = blockIdx.x * LF_NTHREADS + tid; i < LF_SET_WORDS; i += gridDim.x * LF_NTHREADS) other[i] = 0u;
    }

    // ---- packing role of this wave ------------------------------------------------------------------
    const bool isW = wave >= 4;
    const int wq = wave & 3;
    const int prow0 = isW ? (n0 + tile_m * 16 + wq * 4) : (m0 + tile_n * 16 + wq * 4);
    const unsigned char* psrc = reinterpret_cast<const unsigned char*>(isW ? W + (int64_t)prow0 * ldw : X + (int64_t)prow0 * ldx);
    const int64_t psld = (isW ? ldw : ldx) * 4;               // source row stride in bytes
    unsigned char* pdst = (isW ? Wn : Xn) + (int64_t)prow0 * rowb;
    uint32_t* mycnt = cnt + (isW ? LF_W_OFF + tile_n * LF_MAXCH : tile_m * LF_MAXCH);
    const uint32_t* pollX = cnt + tile_m * LF_MAXCH;
    const uint32_t* pollW = cnt + LF_W_OFF + tile_n * LF_MAXCH;
    // wave-private nibble patches (two of 1 KiB, by chunk parity): LDS byte addresses, all accesses from inline asm so
    // that the compiler never puts an lgkmcnt wait of its own into the loop
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) unsigned char*)smem);
    const unsigned stg0 = lds0 + (unsigned)(LF_NBUF * BUF + 64 + wave * LF_STG_WORDS * 4);
    const unsigned stg_w = stg0 + (unsigned)lane * 4u;     // this lane's dword of a row
    const unsigned stg_r = stg0 + (unsigned)lane * 16u;    // this lane's 16 bytes of the 4-row patch
    const unsigned lane16 = (unsigned)lane * 16u;

    auto job_src = [&](int c, int p) -> const unsigned char* { return psrc + (int64_t)p * psld + (int64_t)c * (LF_KC * 4); };
    const float thr_n = isW ? -0.5f : 0.0f, thr_p = isW ? 0.5f : 0.0f;   // ternary launch only (wave-uniform)
    auto conv8 = [&](const v4f& a, const v4f& b) -> uint32_t {
        if constexpr (EncW::TERNARY) {      // nibble = nonzero << 1 | negative << 3 ; x < thr_n -> -1, !(x < thr_p) -> +1 (NaN too)
            const unsigned neg = bits8t<bit_lt>(a, b, thr_n), pos = bits8t<bit_nlt>(a, b, thr_p);
            return (pair_select(neg | pos, LF_TBL_SPREAD) << 1) | (pair_select(neg, LF_TBL_SPREAD) << 3);
        }
        return pair_select(bits8<bit_neg>(a, b), LF_TBL_SIGN);
    };
    auto stage_put = [&](int cb, int p, uint32_t d) {     // row p of the patch of parity cb (scalar part of the address on the SALU)
        asm volatile("ds_write_b32 %0, %1" :: "v"(stg_w + (unsigned)(cb * 1024 + p * 256)), "v"(d) : "memory");
    };
    auto patch_read = [&](v4u& g, int cb) {     // issued right after the last stage_put of a chunk (LDS ops of a wave are ordered)
        asm volatile("ds_read_b128 %0, %1" : "=&v"(g) : "v"(stg_r + (unsigned)(cb * 1024)) : "memory");
    };
    const unsigned store_lane_off = (unsigned)((lane >> 4) * rowb + (lane & 15) * 16);
    auto patch_store = [&](v4u& g, int c) {     // the wave's 4 rows x 512 columns of chunk c: 16 lanes per row
        const unsigned voff = store_lane_off + (unsigned)(c * 256);
        // s_nop 1: a VMEM store of more than 8 bytes reads its data registers late; the hazard recogniser does not look
        // inside inline asm, so the wait states before the next VALU write of `g` are ours (observed without it: the
        // first dword of some lanes stored as 0)
        asm volatile("s_waitcnt lgkmcnt(0)\n\tglobal_store_dwordx4 %1, %0, %2 sc1\n\ts_nop 1" : "+v"(g) : "v"(voff), "s"(pdst) : "memory");
    };
    // Hand-off check, ONE asm block so that the compiler sees straight-line code (a branch between two stages splits the
    // basic block and lets the machine sinker move a stage's MFMAs below the next stage's fragment reads: 96 fragment
    // registers live, spills with vmcnt(0) drains all over the loop).  Fast path: both polled values (vx, vw) show the
    // counters of chunk c complete.  Slow path: bounded spin; a timeout sets bit 0 of `spin_err`, reported once after the
    // loop.  vmcnt is 0 after the slow path, which only makes the counted waits that follow stricter.
    unsigned spin_err = 0, spin_count = 0;
    auto check_ready = [&](unsigned vx, unsigned vw, int c) {
        unsigned it, t;
        asm volatile(
            "v_min_u32 %[vx], %[vx], %[vw]\n\t"
            "s_nop 0\n\t"
            "v_readfirstlane_b32 %[t], %[vx]\n\t"
            "s_cmp_ge_u32 %[t], %[need]\n\t"
            "s_cbranch_scc1 2f\n\t"
            "s_mov_b32 %[it], 0\n"
            "1:\n\t"
            "s_add_u32 %[cnt], %[cnt], 1\n\t"
            "s_sleep 4\n\t"
            "global_load_dword %[vx], %[off], %[px] sc1\n\t"
            "global_load_dword %[vw], %[off], %[pw] sc1\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            "v_min_u32 %[vx], %[vx], %[vw]\n\t"
            "s_nop 0\n\t"
            "v_readfirstlane_b32 %[t], %[vx]\n\t"
            "s_cmp_ge_u32 %[t], %[need]\n\t"
            "s_cbranch_scc1 2f\n\t"
            "s_add_u32 %[it], %[it], 1\n\t"
            "s_cmp_lt_u32 %[it], %[lim]\n\t"
            "s_cbranch_scc1 1b\n\t"
            "s_or_b32 %[err], %[err], 1\n"
            "2:"
#if defined(LF_FIX) && (LF_FIX & 2)
            "\n\tbuffer_inv sc1"
#endif
            : [vx] "+v"(vx), [vw] "+v"(vw), [it] "=&s"(it), [t] "=&s"(t), [err] "+s"(spin_err), [cnt] "+s"(spin_count)
            : [off] "v"(c * 4), [px] "s"(pollX), [pw] "s"(pollW), [need] "n"(LF_ARRIVALS), [lim] "s"(LF_SPIN_MAX)
            : "memory", "scc");
    };
    // one device-scope atomic from lane 0 only, EXEC narrowed inside the statement (an `if (lane == 0)` is a branch)
    auto signal = [&](int c, unsigned inc) {
        unsigned long long keep;
#if defined(LF_FIX) && (LF_FIX & 1)
        asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
#endif
        asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tglobal_atomic_add %1, %2, %3\n\ts_mov_b64 exec, %0"
                     : "=&s"(keep) : "v"(c * 4), "v"(inc), "s"(mycnt) : "memory");
    };

    // ---- fill: chunks 0 .. LEAD-2 complete, row 0 of chunk LEAD-1 into the patch -----------------------
    v4f ra[4], rb[4];                                         // fp32 row-chunks in flight, slot = stage % 4
    {
        v4f fa[2][4], fb[2][4];
        auto fload = [&](int c) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const unsigned char* s = job_src(c, p) + lane16;
                fa[c & 1][p] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(s));
                fb[c & 1][p] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(s + 1024));
            }
        };
        fload(0);
#pragma unroll
        for (int c = 0; c < LEAD - 1; ++c) {
            if (c + 1 < LEAD - 1) fload(c + 1);
            stage_put(c & 1, 0, conv8(fa[c & 1][0], fb[c & 1][0]));
            stage_put(c & 1, 1, conv8(fa[c & 1][1], fb[c & 1][1]));
            stage_put(c & 1, 2, conv8(fa[c & 1][2], fb[c & 1][2]));
            stage_put(c & 1, 3, conv8(fa[c & 1][3], fb[c & 1][3]));
            v4u g;
            patch_read(g, c & 1);
            patch_store(g, c);
        }
        const unsigned char* s = job_src(LEAD - 1, 0) + lane16;
        const v4f a0 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(s));
        const v4f b0 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(s + 1024));
        stage_put((LEAD - 1) & 1, 0, conv8(a0, b0));
    }
    if constexpr (DBG & 8) dbg_t[1] = wall_clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's nibble stores are complete (write-through)
    if constexpr (DBG & 8) dbg_t[2] = wall_clock64();
#pragma unroll
    for (int c = 0; c < LEAD - 1; ++c) signal(c, 1u);
    auto issue_job_loads = [&](v4f& a, v4f& b, int c, int p) {
        const unsigned char* sb = job_src(c, p);
        asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=&v"(a) : "v"(lane16), "s"(sb) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:1024 nt" : "=&v"(b) : "v"(lane16), "s"(sb) : "memory");
    };

    // ---- GEMM side: fragment reads, DMA pieces (mfma_gemm.hip, PP256 fp4) -----------------------------
    v16f acc[TMW][TNW];
#pragma unroll
    for (int a = 0; a < TMW; ++a)
#pragma unroll
        for (int b = 0; b < TNW; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;

    // Fragment reads from inline asm with immediate offsets (left to the compiler, the twelve per-stage addresses are
    // recomputed with VALU adds and spilled): lane address = row * 64 + swizzled chunk * 16; the accumulator-tile index
    // and the stage buffer go into the 16-bit offset field, buffers 2 and 3 through a second base 64 KiB up.
    unsigned fxa[2][2], fwa[2][2];   // [buffer pair][kk]
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const unsigned sw = (unsigned)(((kk * 2 + lhalf) ^ ((lrow >> 2) & 3)) * 16);
        fxa[0][kk] = lds0 + (unsigned)((wave_m * TMW * 32 + lrow) * SB) + sw;
        fwa[0][kk] = lds0 + (unsigned)(LF_XSTAGE + (wave_n * TNW * 32 + lrow) * SB) + sw;
        fxa[1][kk] = fxa[0][kk] + 65536u;
        fwa[1][kk] = fwa[0][kk] + 65536u;
    }
    auto mfma_step = [&](v4u (&xf)[TMW], v4u (&wf)[TNW]) {
#pragma unroll
        for (int a = 0; a < TMW; ++a)
#pragma unroll
            for (int b = 0; b < TNW; ++b) acc[a][b] = mfma_fp4(xf[a], wf[b], acc[a][b]);
    };
    constexpr int XP = 2, WP = 2, NP = 4;
    const int pch = lane % LF_CH, rsub = lane / LF_CH;
    unsigned voffx[XP], voffw[WP];
#pragma unroll
    for (int j = 0; j < XP; ++j) {
        const int row = (j * LF_NWAVES + wave) * LF_RPP + rsub;
        voffx[j] = (unsigned)((m0 + row) * rowb) + (unsigned)(swz<SB>(row, pch) * 16);
    }
#pragma unroll
    for (int j = 0; j < WP; ++j) {
        const int row = (j * LF_NWAVES + wave) * LF_RPP + rsub;
        voffw[j] = (unsigned)((n0 + row) * rowb) + (unsigned)(swz<SB>(row, pch) * 16);
    }
    auto issue_pieces = [&](int s) {              // the wave's 4 pieces of stage s into buffer s & 3
        const unsigned ldsbuf = __builtin_amdgcn_readfirstlane(lds0 + (s & 3) * BUF);
#pragma unroll
        for (int j = 0; j < XP; ++j)
            dma16_sc1(Xn + (int64_t)s * SB, voffx[j], ldsbuf,
                      __builtin_amdgcn_readfirstlane(((j * LF_NWAVES + wave) * LF_RPP) * SB));
#pragma unroll
        for (int j = 0; j < WP; ++j)
            dma16_sc1(Wn + (int64_t)s * SB, voffw[j], ldsbuf,
                      __builtin_amdgcn_readfirstlane(LF_XSTAGE + ((j * LF_NWAVES + wave) * LF_RPP) * SB));
    };

    // role of the wave on its SIMD (mfma_gemm.hip): the two co-resident waves of a SIMD get opposite roles
    int grp;
    {
        volatile int* simd_of = reinterpret_cast<volatile int*>(smem + LF_NBUF * BUF);
        const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);  // HW_ID.SIMD_ID
        if (lane == 0) simd_of[wave] = simd;
        __syncthreads();
        int rank = 0;
        for (int w2 = 0; w2 < LF_NWAVES; ++w2) rank += (w2 < wave && simd_of[w2] == simd) ? 1 : 0;
        grp = __builtin_amdgcn_readfirstlane(rank & 1);
    }

    // ---- hand-off of chunk 0 (and the first look at chunk 1) ------------------------------------------
    check_ready(0u, 0u, 0);
    if constexpr (DBG & 8) dbg_t[3] = wall_clock64();
    unsigned pvx = __hip_atomic_load(pollX + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned pvw = __hip_atomic_load(pollW + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pvx), "+v"(pvw) :: "memory");

    // rows 1..3 of chunk LEAD-1 go into the pipeline registers (converted in stages 0..2).  Issued only now: between an
    // asm load and its counted wait the destination registers must not be touched by anything, spills included
#pragma unroll
    for (int p = 1; p < 4; ++p) issue_job_loads(ra[p], rb[p], LEAD - 1, p);
    {
        const unsigned m0_keep = m0_save();
        issue_pieces(0);
        issue_pieces(1);
        issue_pieces(2);
        m0_restore(m0_keep);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
    __syncthreads();
    if (grp == 0) __syncthreads();   // group A trails by one slot
    if constexpr (DBG & 8) dbg_t[4] = wall_clock64();

    auto stage = [&](int s, auto cfg) {
        using C = decltype(cfg);
        constexpr int P = C::P;
        // (s & 3) == P by construction: every period starts at a multiple of 4
        v4u xf0[TMW], wf0[TNW], xf1[TMW], wf1[TNW];
        {
            constexpr int HI = P >> 1, OFF = (P & 1) * BUF;
#pragma unroll
            for (int a = 0; a < TMW; ++a)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(xf0[a]) : "v"(fxa[HI][0]), "n"(OFF + a * 32 * SB));
#pragma unroll
            for (int b = 0; b < TNW; ++b)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(wf0[b]) : "v"(fwa[HI][0]), "n"(OFF + b * 32 * SB));
#pragma unroll
            for (int a = 0; a < TMW; ++a)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(xf1[a]) : "v"(fxa[HI][1]), "n"(OFF + a * 32 * SB));
#pragma unroll
            for (int b = 0; b < TNW; ++b)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(wf1[b]) : "v"(fwa[HI][1]), "n"(OFF + b * 32 * SB));
        }
        v4u pg;
        if constexpr (C::CONV) {                   // the row-chunk loaded three stages ago: slot (P + 1) & 3
            constexpr int SL = (P + 1) & 3;
            const int u = s - 3 + 4 * LEAD;        // job index = 4 * chunk + row
            asm volatile("" : "+v"(ra[SL]), "+v"(rb[SL]));
            stage_put((u >> 2) & 1, SL, conv8(ra[SL], rb[SL]));
            if constexpr (C::STORE) patch_read(pg, (u >> 2) & 1);
        }
        if constexpr (C::SIG) {                    // the store issued three stages ago is complete
            const int sc = ((s - 1) >> 2) + LEAD - 2;
            signal(sc, sc >= LEAD - 1 ? 1u : 0u);   // chunks < LEAD-1 were signalled by the fill: add 0 keeps the op count
        }
        if constexpr (C::CHECK) check_ready(pvx, pvw, (s + 3) >> 2);   // before the DMA of the first stage of chunk (s + 3) / 4
        if constexpr (C::POLL) {
            const int pc = (s + 6) >> 2;
            asm volatile("global_load_dword %0, %1, %2 sc1" : "=&v"(pvx) : "v"(pc * 4), "s"(pollX) : "memory");
            asm volatile("global_load_dword %0, %1, %2 sc1" : "=&v"(pvw) : "v"(pc * 4), "s"(pollW) : "memory");
        }
        if constexpr (C::LOADS) {
            const int u = s + 4 * LEAD;
            issue_job_loads(ra[P], rb[P], u >> 2, P);
        }
        if constexpr (C::STORE) patch_store(pg, (s - 3 + 4 * LEAD) >> 2);
        if constexpr (C::ISSUE) {
            const unsigned m0_keep = m0_save();
            issue_pieces(s + 3);
            m0_restore(m0_keep);
        }
        // the one wait of the load segment; the fragments are tied to it so that no MFMA can be scheduled above it
        asm volatile("s_waitcnt vmcnt(%12) lgkmcnt(0)"
                     : "+v"(xf0[0]), "+v"(xf0[1]), "+v"(xf0[2]), "+v"(xf0[3]), "+v"(wf0[0]), "+v"(wf0[1]), "+v"(xf1[0]),
                       "+v"(xf1[1]), "+v"(xf1[2]), "+v"(xf1[3]), "+v"(wf1[0]), "+v"(wf1[1])
                     : "n"(C::ISSUE ? C::NEND : 0)
                     : "memory");
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(xf0, wf0);
        mfma_step(xf1, wf1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(C::LAST && grp == 0)) __syncthreads();   // A's last compute has no partner segment
        __builtin_amdgcn_sched_barrier(0);
    };

    // phase 1: stages 0 .. nstages - 4 LEAD - 1 carry a fp32 row-chunk each (first stage: the prologue issued
    // pieces(2) = 4 operations last, not a full stage of 6)
    const int s_pack_end = nstages - 4 * LEAD;
    stage(0, Steady<0, 4 + 6>{});
    stage(1, Steady<1, 6 + 7>{});
    stage(2, Steady<2, 7 + 9>{});
    stage(3, Steady<3, 9 + 6>{});
    int s = 4;
    for (; s < s_pack_end; s += 4) {
        stage(s, Steady<0, 6 + 6>{});
        stage(s + 1, Steady<1, 6 + 7>{});
        stage(s + 2, Steady<2, 7 + 9>{});
        stage(s + 3, Steady<3, 9 + 6>{});
    }
    if constexpr (DBG & 8) dbg_t[5] = wall_clock64();
    // phase 2: LEAD - 1 drain periods
    auto drain = [&](auto q_) {
        constexpr int Q = decltype(q_)::value;
        stage(s, DrainCfg<LEAD, Q, 0>{});
        stage(s + 1, DrainCfg<LEAD, Q, 1>{});
        stage(s + 2, DrainCfg<LEAD, Q, 2>{});
        stage(s + 3, DrainCfg<LEAD, Q, 3>{});
        s += 4;
    };
    drain(std::integral_constant<int, 0>{});
    drain(std::integral_constant<int, 1>{});
    if constexpr (LEAD >= 4) drain(std::integral_constant<int, 2>{});
    if constexpr (LEAD >= 5) drain(std::integral_constant<int, 3>{});
    if constexpr (LEAD >= 6) drain(std::integral_constant<int, 4>{});
    static_assert(LEAD <= 6, "add drain periods");
    // tail: the last four stages; only the first still issues DMA (stage nstages - 1)
    stage(s, StageCfg<0, 8, false, false, false, false, false, false, true, false>{});
    stage(s + 1, StageCfg<1, 0, false, false, false, false, false, false, false, false>{});
    stage(s + 2, StageCfg<2, 0, false, false, false, false, false, false, false, false>{});
    stage(s + 3, StageCfg<3, 0, false, false, false, false, false, false, false, true>{});

    if constexpr (DBG & 8) dbg_t[6] = wall_clock64();
    // every poll of this launch is done: the last workgroup to get here flips the counter set for the next launch
    if (spin_err && lane == 0) atomicOr(sync + LF_SYNC_ERR, 1u);
    if (tid == 0) {
        const unsigned old = __hip_atomic_fetch_add(sync + LF_SYNC_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {
            __hip_atomic_store(sync + LF_SYNC_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + LF_SYNC_EPOCH, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- epilogue (mfma_gemm.hip, wide form): 32x32 accumulator tiles through a wave-private 4 KiB LDS patch,
    // leaving as dwordx4 stores of full 128-byte lines
    float* T = reinterpret_cast<float*>(smem) + wave * 1024;
#pragma unroll
    for (int b = 0; b < TNW; ++b) {
        const int nb = n0 + (wave_n * TNW + b) * 32;
        const float bv = bias ? bias[nb + lrow] : 0.0f;
#pragma unroll
        for (int a = 0; a < TMW; ++a) {
            const int mb = m0 + (wave_m * TMW + a) * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) T[((r & 3) + 8 * (r >> 2) + 4 * lhalf) * 32 + lrow] = acc[a][b][r] + bv;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + (lane >> 3), c4 = (lane & 7) * 4;
                const float4 v = *reinterpret_cast<const float4*>(T + row * 32 + c4);
                *reinterpret_cast<float4*>(Y + (int64_t)(mb + row) * ldy + nb + c4) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    if constexpr (DBG & 8) {   // bring-up only: waves 0 and 4 overwrite the head of their first Y row with phase stamps
        dbg_t[7] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg_t[8] = wall_clock64();
        dbg_t[9] = spin_count;
        if (lane == 0 && wave_n == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(Y + (int64_t)(m0 + wave_m * TMW * 32) * ldy + n0);
            for (int i = 0; i < 10; ++i) o[i] = dbg_t[i];
        }
    }
}

constexpr int LF_LEAD = 5;

template <class EncW>
int launch_fused(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y, int64_t ldy,
                 int64_t M, int64_t N, int64_t K, unsigned char* ws, qt_stream_t stream) {
#ifdef QT_LF_STAMPS   // bring-up build only (tools/stamps_fused.py): phase stamps are written over Y
    auto kern = linear_fused_kernel<EncW, LF_LEAD, 8>;
#else
    auto kern = linear_fused_kernel<EncW, LF_LEAD, 0>;
#endif
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LF_LDS) != hipSuccess)
        return QT_ERR_LAUNCH;
    uint32_t* sync = reinterpret_cast<uint32_t*>(ws);
    unsigned char* xn = ws + LF_SYNC_RESERVED;
    unsigned char* wn = xn + M * (K / 2);
    hipLaunchKernelGGL(kern, dim3((unsigned)((M / 256) * (N / 256))), dim3(LF_NTHREADS), LF_LDS, (hipStream_t)stream, x, ldx,
                       w, ldw, bias, y, ldy, xn, wn, sync, (int)M, (int)N, (int)K);
    return qt_check_launch();
}

bool fused_shape_ok(int64_t M, int64_t N, int64_t K) {
    return M == 4096 && N == 4096 && K % LF_KC == 0 && K >= (int64_t)LF_KC * (LF_LEAD + 1) && K <= (int64_t)LF_KC * LF_MAXCH;
}

}  // namespace

extern "C" {

int64_t qt_linear_fused_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    if (!fused_shape_ok(M, N, K)) return 0;
    return LF_SYNC_RESERVED + (M + N) * (K / 2);
}

int qt_linear_fused_f32(const float* x, int64_t ldx, const float* w, int64_t ldw, const float* bias, float* y,
                        int64_t ldy, int64_t M, int64_t N, int64_t K, int w_ternary, void* workspace,
                        int64_t workspace_bytes, qt_stream_t stream) {
    if (M < 0 || N < 0 || K < 0 || !x || !w || !y || !workspace) return QT_ERR_INVALID_ARG;
    if (!fused_shape_ok(M, N, K)) return QT_ERR_UNSUPPORTED;
    if (ldx < K || ldw < K || ldy < N) return QT_ERR_INVALID_ARG;
    if (workspace_bytes < qt_linear_fused_workspace_bytes(M, N, K)) return QT_ERR_INVALID_ARG;
    if ((ldx & 3) || (ldw & 3) || (ldy & 3) || !qt_aligned16(x) || !qt_aligned16(w) || !qt_aligned16(y) ||
        (reinterpret_cast<uintptr_t>(workspace) & 4095u))
        return QT_ERR_ALIGNMENT;
    if (M * ldx * 4 >= (1ll << 40) || N * ldw * 4 >= (1ll << 40)) return QT_ERR_UNSUPPORTED;
    // all 256 workgroups must be resident at once (they hand data to each other): one per CU
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return QT_ERR_NO_DEVICE;
    if (cus < (M / 256) * (N / 256)) return QT_ERR_UNSUPPORTED;
    unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
    return w_ternary ? launch_fused<FTernary>(x, ldx, w, ldw, bias, y, ldy, M, N, K, ws, stream)
                     : launch_fused<FSign>(x, ldx, w, ldw, bias, y, ldy, M, N, K, ws, stream);
}

}  // extern "C"
