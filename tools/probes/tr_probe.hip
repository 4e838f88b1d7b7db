// Probe (GPU box): what ds_read_b64_tr_b16 returns, lane by lane, for (A) consecutive 8-byte addresses and (B) the
// [col block][32 rows][16 cols] bf16 image csrc/tall.hip's gram kernel reads its MFMA operands from.  LDS element e
// (16-bit) holds the value e, so every returned element names the LDS element it came from.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/tr_probe tools/probes/tr_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ void probe(uint16_t* out, int mode)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    const int l = threadIdx.x;
    for (int e = l; e < 4096; e += 64) lds[e] = static_cast<uint16_t>(e);
    __syncthreads();
    const int t = l & 15, q = l >> 4;
    uint32_t addr;                       // bytes
    if (mode == 0) addr = 8u * l;
    else if (mode == 1) addr = static_cast<uint32_t>((4 * q + t / 4) * 32 + (t % 4) * 8);            // rows 4q.., 32-B rows
    else addr = static_cast<uint32_t>((16 + 4 * q + t / 4) * 32 + (t % 4) * 8);                      // second half
    addr += static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds) & 0xffffffffu);
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = static_cast<uint16_t>(v >> (16 * j));
}

int main()
{
    uint16_t* d;
    hipMalloc(&d, 64 * 4 * sizeof(uint16_t));
    for (int mode = 0; mode < 3; ++mode) {
        probe<<<1, 64>>>(d, mode);
        uint16_t h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d: lane: source LDS elements of result elements 0..3\n", mode);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4u %4u %4u %4u\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
        if (mode) {
            int bad = 0;
            for (int l = 0; l < 64; ++l)
                for (int j = 0; j < 4; ++j) {
                    const int i = l & 15, q = l >> 4, row = (mode == 2 ? 16 : 0) + 4 * q + j;
                    if (h[l * 4 + j] != row * 16 + i) ++bad;
                }
            printf("mode %d: expected lane (i, q) elem j = image[row %s4 q + j][col i]: %s (%d mismatches)\n", mode,
                   mode == 2 ? "16 + " : "", bad ? "NO" : "YES", bad);
        }
    }
    return 0;
}
