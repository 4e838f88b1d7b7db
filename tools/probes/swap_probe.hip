// Probe (GPU box): lane semantics of v_permlane16_swap / v_permlane32_swap and of a row_ror:8 DPP add, as csrc/spmm.hip's
// bf16 reduce-scatter uses them.  a[lane] = lane, b[lane] = 100 + lane.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/swap_probe tools/probes/swap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(unsigned* out)
{
    const unsigned l = threadIdx.x, a = l, b = 100 + l;
    const auto s16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    const auto s32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const int r = __builtin_amdgcn_update_dpp(0, static_cast<int>(a), 0x128, 0xf, 0xf, false);
    out[l] = s16[0]; out[64 + l] = s16[1]; out[128 + l] = s32[0]; out[192 + l] = s32[1]; out[256 + l] = static_cast<unsigned>(r);
}

int main()
{
    unsigned *d, h[320];
    hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[5] = {"permlane16_swap(a, b)[0]", "permlane16_swap(a, b)[1]", "permlane32_swap(a, b)[0]",
                            "permlane32_swap(a, b)[1]", "dpp row_ror:8 of a"};
    for (int k = 0; k < 5; ++k) {
        printf("%s:\n", names[k]);
        for (int l = 0; l < 64; ++l) printf("%4u%s", h[k * 64 + l], (l & 15) == 15 ? "\n" : "");
    }
    return 0;
}
