// Shared helpers for the gfx950 kernels behind include/qt_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/qt_hip.h"

#define QT_VERSION_INT 100 /* 0.1.0 */

static inline int qt_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? QT_OK : QT_ERR_LAUNCH;
}

// Kernels with more than 64 KiB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize raised once per (kernel, device).
// `once` is a function-local static of the launching template instantiation: the attribute call (a runtime lock + lookup) leaves
// the per-launch path after the first launch on a device (host overhead of the launch-bound module-graph path).
struct QtLdsOnce { std::atomic<int> have[16]; };
static inline int qt_ensure_dyn_lds(QtLdsOnce& once, const void* fn, int bytes) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return QT_ERR_LAUNCH;
    const bool tracked = dev >= 0 && dev < 16;
    if (tracked && bytes <= once.have[dev].load(std::memory_order_acquire)) return QT_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return QT_ERR_LAUNCH;
    if (tracked) once.have[dev].store(bytes, std::memory_order_release);     // only a LARGER request calls again
    return QT_OK;
}

static inline bool qt_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// MI355X: 256 CUs.  Memory-bound grids are capped at 8 resident 256-thread blocks per CU and
// grid-stride over the rest (cdna_hip_programming.md Guideline 11).
static inline int qt_stream_grid(int64_t work_items_per_block_total, int max_blocks = 256 * 8) {
    int64_t g = work_items_per_block_total;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// safeSign bit: 1 <=> x < 0.  Plain IEEE compare: -0.0 and NaN give 0, subnormals compare
// un-flushed (hipcc's default float_denorm_mode_32 keeps f32 subnormals).
__device__ __forceinline__ uint32_t qt_neg_bit(float x) { return x < 0.0f ? 1u : 0u; }

__device__ __forceinline__ float qt_safe_sign(float x) { return x < 0.0f ? -1.0f : 1.0f; }

// TernaryConnectDeterministic: (s + safeSign(x - 0.5*s)) / 2 with s = safeSign(x)
// (functions/terner_connect.py:26-27)  ==  x >= 0.5 -> +1 ; x < -0.5 -> -1 ; else 0 ; NaN -> +1.
__device__ __forceinline__ float qt_ternarize(float x) {
    const float s = qt_safe_sign(x);
    const float t = qt_safe_sign(x - 0.5f * s);
    return (s + t) * 0.5f;
}
