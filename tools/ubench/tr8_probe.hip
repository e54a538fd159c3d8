// Probe of ds_read_b64_tr_b8 (gfx950): which eight bytes does lane l receive when every lane supplies its own 8-byte-aligned
// address?  LDS holds u16-wide ids split as byte value = (element index) & 0xff with a pitch chosen so that ids are unique per block.
//   hipcc --offload-arch=gfx950 -O2 tr8_probe.hip -o tr8_probe && ./tr8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void probe(uint8_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    // byte at (row r, col c) of a [rows][32] image = r * 16 + c (c < 16 unique per row for r < 16)
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint8_t)(((i / 32) % 16) * 16 + (i % 32) % 16);
    __syncthreads();
    const int l = threadIdx.x;
    const unsigned base = (unsigned)(uintptr_t)((__attribute__((address_space(3))) uint8_t*)lds);
    unsigned addr;
    if (mode == 0) addr = base + (l & 15) * 8 + (l >> 4) * 1024;               // contiguous 128 bytes per 16-lane group: [4 rows][32 B]
    else addr = base + ((l & 15) >> 1) * 32 + (l & 1) * 8 + (l >> 4) * 1024;    // 8 rows (pitch 32 B) x 16 columns: 2 lanes per row
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 8; ++j) out[l * 8 + j] = (uint8_t)(v >> (8 * j));
}

int main() {
    uint8_t* d;
    (void)hipMalloc(&d, 64 * 8);
    uint8_t h[512];
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(d, mode);
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d  (value = row * 16 + col, printed as row.col)\n", mode);
        for (int l = 0; l < 32; ++l) {
            printf("  lane %2d:", l);
            for (int j = 0; j < 8; ++j) printf(" %2d.%-2d", h[l * 8 + j] / 16, h[l * 8 + j] % 16);
            printf("\n");
        }
    }
    return 0;
}
