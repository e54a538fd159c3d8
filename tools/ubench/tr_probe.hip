// Probe of ds_read_b64_tr_b16 (gfx950): which four 16-bit LDS elements does lane l receive when every lane supplies its own
// 8-byte-aligned address?  LDS holds u16 value = element index; three addressing patterns are printed.
//   hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    const unsigned base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) uint16_t*)lds);
    if (mode == 0) addr = base + l * 8;                                    // contiguous: lane l -> elements 4l .. 4l+3
    else if (mode == 1) addr = base + ((l & 15) / 4) * 64 + (l & 3) * 8 + (l >> 4) * 1024;   // [4 rows of 32 elems][16 cols] blocks
    else addr = base + (l & 15) * 128 + (l >> 4) * 8;                      // every lane its own row (pitch 64 elems)
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}

int main() {
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        probe<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d: %4d %4d %4d %4d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
            if ((l & 3) == 3) printf("\n");
        }
    }
    return 0;
}
