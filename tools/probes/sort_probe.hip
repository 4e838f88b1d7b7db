// Measurement helper (GPU box): rocPRIM onesweep configurations for the row-bucket sort of csrc/magop.hip
// (40 M u64 keys, 20 key bits starting at bit 32; keys only and (key, f32) pairs).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/sort_probe.hip -o tools/probes/sort_probe
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill(uint64_t* k, float* v, size_t m, uint32_t nrow)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t x = i * 0x9E3779B97F4A7C15ull;
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        const uint64_t row = x % nrow, col = (x >> 20) % nrow;
        k[i] = (row << 32) | (col << 1) | (i & 1);
        v[i] = (float)(i & 1023);
    }
}

template <class Config>
float run(const char* name, uint64_t* a, uint64_t* b, float* va, float* vb, size_t m, unsigned b0, unsigned b1, bool pairs)
{
    size_t bytes = 0;
    if (pairs) CK((rocprim::radix_sort_pairs<Config>(nullptr, bytes, a, b, va, vb, m, b0, b1, 0)));
    else CK((rocprim::radix_sort_keys<Config>(nullptr, bytes, a, b, m, b0, b1, 0)));
    void* tmp; CK(hipMalloc(&tmp, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(e0));
        if (pairs) CK((rocprim::radix_sort_pairs<Config>(tmp, bytes, a, b, va, vb, m, b0, b1, 0)));
        else CK((rocprim::radix_sort_keys<Config>(tmp, bytes, a, b, m, b0, b1, 0)));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    // sortedness on the selected bits + stability spot check on the host (first 1 M)
    std::vector<uint64_t> h(1 << 20);
    CK(hipMemcpy(h.data(), b, h.size() * 8, hipMemcpyDeviceToHost));
    bool ok = true;
    const uint64_t mask = ((1ull << (b1 - b0)) - 1) << b0;
    for (size_t i = 1; i < h.size(); ++i) ok = ok && ((h[i - 1] & mask) <= (h[i] & mask));
    printf("%-44s %s  %.3f ms  tmp %.1f MB  sorted=%d\n", name, pairs ? "pairs" : "keys ", best, bytes / 1e6, (int)ok);
    CK(hipFree(tmp));
    return best;
}

template <unsigned HBS, unsigned HIPT, unsigned BS, unsigned IPT, unsigned BITS, rocprim::block_radix_rank_algorithm ALG>
using OS = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                      rocprim::radix_sort_onesweep_config<rocprim::kernel_config<HBS, HIPT>, rocprim::kernel_config<BS, IPT>, BITS, ALG>>;

int main(int argc, char** argv)
{
    const size_t m = argc > 1 ? atol(argv[1]) : 40000000;
    const uint32_t nrow = argc > 2 ? atoi(argv[2]) : 1000000;
    unsigned bits = 1; while ((nrow >> bits) != 0) ++bits;
    uint64_t *a, *b; float *va, *vb;
    CK(hipMalloc(&a, m * 8)); CK(hipMalloc(&b, m * 8)); CK(hipMalloc(&va, m * 4)); CK(hipMalloc(&vb, m * 4));
    fill<<<4096, 256>>>(a, va, m, nrow);
    CK(hipDeviceSynchronize());
    const unsigned b0 = 32, b1 = 32 + bits;
    using A = rocprim::block_radix_rank_algorithm;
    for (int pairs = 0; pairs < 2; ++pairs) {
        run<rocprim::default_config>("default", a, b, va, vb, m, b0, b1, pairs);
        run<OS<1024, 12, 1024, 12, 10, A::match>>("h1024x12 s1024x12 10 bits", a, b, va, vb, m, b0, b1, pairs);
        run<OS<1024, 12, 1024, 14, 10, A::match>>("h1024x12 s1024x14 10 bits", a, b, va, vb, m, b0, b1, pairs);
        run<OS<1024, 12, 1024, 16, 10, A::match>>("h1024x12 s1024x16 10 bits", a, b, va, vb, m, b0, b1, pairs);
        run<OS<1024, 12, 1024, 20, 10, A::match>>("h1024x12 s1024x20 10 bits", a, b, va, vb, m, b0, b1, pairs);
        run<OS<1024, 32, 1024, 12, 10, A::match>>("h1024x32 s1024x12 10 bits", a, b, va, vb, m, b0, b1, pairs);
        run<OS<512, 32, 1024, 12, 10, A::match>>("h512x32 s1024x12 10 bits", a, b, va, vb, m, b0, b1, pairs);
        run<OS<256, 16, 1024, 12, 10, A::match>>("h256x16 s1024x12 10 bits", a, b, va, vb, m, b0, b1, pairs);
        run<OS<1024, 32, 1024, 16, 10, A::match>>("h1024x32 s1024x16 10 bits", a, b, va, vb, m, b0, b1, pairs);
    }
    return 0;
}
