// Experiment driver (NOT product code): C2 steps enqueued from C so that the host is not the bottleneck, on streams with
// CU masks.  Built by tools/experiments/cu_pipeline.py against libqt_hip.so; see that file for what is measured.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <vector>
#include "qt_hip.h"

namespace {
__global__ void hwid_kernel(uint32_t* out, int spin) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_ID
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
    }
    // keep the workgroup resident for a while so that the launch spreads over every CU the queue may use
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}

hipStream_t make_stream(const uint32_t* mask, int words) {
    hipStream_t s = nullptr;
    if (!mask) { hipStreamCreateWithFlags(&s, hipStreamNonBlocking); return s; }
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) != hipSuccess) return nullptr;
    return s;
}

double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

extern "C" {

// which (XCC, SE, CU) the workgroups of a masked stream land on: out[2*i] = HW_ID, out[2*i+1] = XCC_ID of workgroup i
int exp_hwid(const uint32_t* mask, int words, uint32_t* out_dev, int nblocks) {
    hipStream_t s = make_stream(mask, words);
    if (!s) return -1;
    hipLaunchKernelGGL(hwid_kernel, dim3(nblocks), dim3(64), 0, s, out_dev, 2000);
    hipStreamSynchronize(s);
    hipStreamDestroy(s);
    return 0;
}

struct Bufs {
    const float *x, *w;
    uint32_t *xn[2], *wn[2];
    float* y[2];
    int64_t B, N, K, ld;
};

static int pack(const Bufs& b, int i, hipStream_t s) {
    return qt_pack_pair_nib_f32(b.x, b.K, b.xn[i], b.ld, b.B, b.w, b.K, b.wn[i], b.ld, b.N, b.K, 0, s);
}
static int gemm(const Bufs& b, int i, int variant, hipStream_t s) {
    return qt_nib_gemm_variant(variant, b.xn[i], b.ld, b.wn[i], b.ld, nullptr, b.y[i], b.N, b.B, b.N, b.K, s);
}

// mode 0: one stream, pack -> GEMM back to back
// mode 1: GEMM(i) on stream G || pack(i+1) on stream P (double-buffered planes, two event dependencies per step)
// mode 2: whole steps alternating between streams A and B (antiphase partitions)
// returns microseconds per step (device time from events bracketing the whole run on the streams), < 0 on error;
// host_us_per_step: how long the host needed to enqueue a step
double exp_run(int mode, const uint32_t* mask_a, const uint32_t* mask_b, int words, int variant, int steps,
               const float* x, const float* w, uint32_t* xn0, uint32_t* wn0, uint32_t* xn1, uint32_t* wn1, float* y0,
               float* y1, int64_t B, int64_t N, int64_t K, int64_t ld, double* host_us_per_step) {
    Bufs b{x, w, {xn0, xn1}, {wn0, wn1}, {y0, y1}, B, N, K, ld};
    hipStream_t sa = make_stream(mask_a, words), sb = make_stream(mask_b, words);
    if (!sa || !sb) return -1.0;
    hipEvent_t t0, t1, packed[2], freed[2], ja, jb;
    hipEventCreate(&t0);
    hipEventCreate(&t1);
    for (int i = 0; i < 2; ++i) {
        hipEventCreateWithFlags(&packed[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&freed[i], hipEventDisableTiming);
    }
    hipEventCreateWithFlags(&ja, hipEventDisableTiming);
    hipEventCreateWithFlags(&jb, hipEventDisableTiming);
    int rc = 0;
    double host = 0.0;
    for (int pass = 0; pass < 2; ++pass) {   // pass 0 = warm-up
        const int n = pass ? steps : 20;
        hipDeviceSynchronize();
        const double h0 = now_us();
        if (mode == 0) {
            hipEventRecord(t0, sa);
            for (int i = 0; i < n; ++i) { rc |= pack(b, 0, sa); rc |= gemm(b, 0, variant, sa); }
            hipEventRecord(t1, sa);
        } else if (mode == 1) {
            hipStream_t sg = sa, sp = sb;
            hipEventRecord(t0, sp);
            hipStreamWaitEvent(sg, t0, 0);
            rc |= pack(b, 0, sp);
            hipEventRecord(packed[0], sp);
            for (int i = 0; i < n; ++i) {
                const int c = i & 1, o = c ^ 1;
                if (i >= 1) hipStreamWaitEvent(sp, freed[o], 0);
                rc |= pack(b, o, sp);                 // operands of step i + 1
                hipEventRecord(packed[o], sp);
                hipStreamWaitEvent(sg, packed[c], 0);
                rc |= gemm(b, c, variant, sg);
                hipEventRecord(freed[c], sg);
            }
            hipEventRecord(jb, sp);
            hipStreamWaitEvent(sg, jb, 0);
            hipEventRecord(t1, sg);
        } else {
            hipEventRecord(t0, sa);
            hipStreamWaitEvent(sb, t0, 0);
            for (int i = 0; i < n; ++i) {
                hipStream_t s = (i & 1) ? sb : sa;
                rc |= pack(b, i & 1, s);
                rc |= gemm(b, i & 1, variant, s);
            }
            hipEventRecord(jb, sb);
            hipStreamWaitEvent(sa, jb, 0);
            hipEventRecord(t1, sa);
        }
        host = (now_us() - h0) / n;
        hipDeviceSynchronize();
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, t0, t1);
    if (host_us_per_step) *host_us_per_step = host;
    hipEventDestroy(t0);
    hipEventDestroy(t1);
    for (int i = 0; i < 2; ++i) { hipEventDestroy(packed[i]); hipEventDestroy(freed[i]); }
    hipEventDestroy(ja);
    hipEventDestroy(jb);
    hipStreamDestroy(sa);
    hipStreamDestroy(sb);
    if (rc) return -2.0;
    return (double)ms * 1e3 / steps;
}

}  // extern "C"
