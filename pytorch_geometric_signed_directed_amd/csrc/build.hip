// Operator build (COO -> CSR, stable key sort for coalesce) and the element-wise epilogues.
// Integer/byte work, HBM-bound; the sort is rocPRIM's device radix sort (stable), everything
// around it is hand-written.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace pygsd {
namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(int64_t n, int per_block = kBlock)
{
    int64_t g = (n + per_block - 1) / per_block;
    const int64_t cap = 256 * 32;  // 256 CUs x 32 blocks, grid-stride beyond
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return static_cast<unsigned>(g);
}

inline int bits_for(uint64_t max_value)
{
    int b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}

__global__ void make_seg_keys(const int64_t* __restrict__ seg, int64_t n, uint32_t* __restrict__ keys,
                              uint32_t* __restrict__ ids)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        keys[i] = static_cast<uint32_t>(seg[i]);
        ids[i] = static_cast<uint32_t>(i);
    }
}

__global__ void iota_u32(uint32_t* __restrict__ ids, int64_t n)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        ids[i] = static_cast<uint32_t>(i);
}

// rowptr[r] = number of sorted keys < r  (lower bound), r = 0 .. n_seg
__global__ void rowptr_from_sorted(const uint32_t* __restrict__ keys, int64_t n, int32_t n_seg,
                                   int32_t* __restrict__ rowptr)
{
    for (int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; r <= n_seg;
         r += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (keys[mid] < static_cast<uint32_t>(r)) lo = mid + 1; else hi = mid;
        }
        rowptr[r] = static_cast<int32_t>(lo);
    }
}

__global__ void gather_col(const int64_t* __restrict__ other, const int32_t* __restrict__ perm,
                           int64_t n, int32_t* __restrict__ col)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        col[i] = static_cast<int32_t>(other[perm[i]]);
}

__global__ void gather_f32(const float* __restrict__ src, const int32_t* __restrict__ perm, int64_t n,
                           float* __restrict__ out)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        out[i] = src[perm[i]];
}

__global__ void complex_relu_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                    float* __restrict__ ore, float* __restrict__ oim, int64_t n)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const float r = re[i];
        const float m = r >= 0.f ? 1.f : 0.f;  // NaN >= 0 is false, as in the reference mask
        ore[i] = m * r;
        oim[i] = m * im[i];
    }
}

__global__ void complex_relu_vec_kernel(const float4* __restrict__ re, const float4* __restrict__ im,
                                        float4* __restrict__ ore, float4* __restrict__ oim, int64_t n4)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n4;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const float4 r = re[i];
        const float4 q = im[i];
        const float mx = r.x >= 0.f ? 1.f : 0.f, my = r.y >= 0.f ? 1.f : 0.f;
        const float mz = r.z >= 0.f ? 1.f : 0.f, mw = r.w >= 0.f ? 1.f : 0.f;
        ore[i] = make_float4(mx * r.x, my * r.y, mz * r.z, mw * r.w);
        oim[i] = make_float4(mx * q.x, my * q.y, mz * q.z, mw * q.w);
    }
}

__global__ void complex_relu_bwd_kernel(const float* __restrict__ re, const float* __restrict__ gr,
                                        const float* __restrict__ gi, float* __restrict__ ogr,
                                        float* __restrict__ ogi, int64_t n)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const float m = re[i] >= 0.f ? 1.f : 0.f;
        ogr[i] = m * gr[i];
        ogi[i] = m * gi[i];
    }
}

struct CsrWs {
    size_t keys_in, keys_out, ids, temp, temp_bytes, total;
};

int csr_ws_layout(int64_t nnz, CsrWs* w)
{
    size_t temp = 0;
    uint32_t* k = nullptr;
    PYGSD_HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp, k, k, k, k, static_cast<size_t>(nnz), 0u, 32u,
                                            hipStream_t(nullptr)));
    size_t off = 0;
    const size_t arr = round_up(static_cast<size_t>(nnz) * sizeof(uint32_t), 256);
    w->keys_in = off; off += arr;
    w->keys_out = off; off += arr;
    w->ids = off; off += arr;
    w->temp = off; off += round_up(temp, 256);
    w->temp_bytes = temp;
    w->total = off + 256;
    return 0;
}

struct SortWs {
    size_t ids, temp, temp_bytes, total;
};

int sort_ws_layout(int64_t n, SortWs* w)
{
    size_t temp = 0;
    uint64_t* k = nullptr;
    uint32_t* v = nullptr;
    PYGSD_HIP_TRY(rocprim::radix_sort_pairs(nullptr, temp, k, k, v, v, static_cast<size_t>(n), 0u, 64u,
                                            hipStream_t(nullptr)));
    size_t off = 0;
    w->ids = off; off += round_up(static_cast<size_t>(n) * sizeof(uint32_t), 256);
    w->temp = off; off += round_up(temp, 256);
    w->temp_bytes = temp;
    w->total = off + 256;
    return 0;
}

inline char* align256(void* p)
{
    return reinterpret_cast<char*>(round_up(reinterpret_cast<uintptr_t>(p), 256));
}

}  // namespace
}  // namespace pygsd

using namespace pygsd;

extern "C" int pygsd_csr_from_coo_workspace(int64_t nnz, int32_t n_seg, size_t* bytes)
{
    PYGSD_REQUIRE(bytes, "pygsd_csr_from_coo_workspace: null output");
    PYGSD_REQUIRE(nnz >= 0 && nnz < (int64_t(1) << 31) && n_seg >= 0,
                  "pygsd_csr_from_coo_workspace: nnz=%lld n_seg=%d out of int32 range",
                  static_cast<long long>(nnz), n_seg);
    CsrWs w;
    if (int rc = csr_ws_layout(nnz > 0 ? nnz : 1, &w)) return rc;
    *bytes = w.total;
    return 0;
}

extern "C" int pygsd_csr_from_coo(const int64_t* seg, const int64_t* other, int64_t nnz, int32_t n_seg,
                                  int32_t* rowptr, int32_t* col, int32_t* perm, void* workspace,
                                  size_t workspace_bytes, void* stream)
{
    PYGSD_REQUIRE(nnz >= 0 && nnz < (int64_t(1) << 31) && n_seg >= 0,
                  "pygsd_csr_from_coo: nnz=%lld n_seg=%d out of int32 range",
                  static_cast<long long>(nnz), n_seg);
    PYGSD_REQUIRE(rowptr, "pygsd_csr_from_coo: null rowptr");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    if (nnz == 0) {
        PYGSD_HIP_TRY(hipMemsetAsync(rowptr, 0, sizeof(int32_t) * (static_cast<size_t>(n_seg) + 1), s));
        return 0;
    }
    PYGSD_REQUIRE(seg && other && col && perm && workspace, "pygsd_csr_from_coo: null pointer");
    CsrWs w;
    if (int rc = csr_ws_layout(nnz, &w)) return rc;
    PYGSD_REQUIRE(workspace_bytes >= w.total, "pygsd_csr_from_coo: workspace too small (%zu < %zu)",
                  workspace_bytes, w.total);
    char* base = align256(workspace);
    uint32_t* keys_in = reinterpret_cast<uint32_t*>(base + w.keys_in);
    uint32_t* keys_out = reinterpret_cast<uint32_t*>(base + w.keys_out);
    uint32_t* ids = reinterpret_cast<uint32_t*>(base + w.ids);
    void* temp = base + w.temp;
    size_t temp_bytes = w.temp_bytes;

    hipLaunchKernelGGL(make_seg_keys, dim3(grid_for(nnz)), dim3(kBlock), 0, s, seg, nnz, keys_in, ids);
    if (int rc = check_launch("make_seg_keys")) return rc;
    const unsigned bits = static_cast<unsigned>(bits_for(n_seg > 0 ? static_cast<uint64_t>(n_seg) - 1 : 0));
    PYGSD_HIP_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, ids,
                                            reinterpret_cast<uint32_t*>(perm), static_cast<size_t>(nnz),
                                            0u, bits, s));
    hipLaunchKernelGGL(rowptr_from_sorted, dim3(grid_for(static_cast<int64_t>(n_seg) + 1)), dim3(kBlock), 0,
                       s, keys_out, nnz, n_seg, rowptr);
    if (int rc = check_launch("rowptr_from_sorted")) return rc;
    hipLaunchKernelGGL(gather_col, dim3(grid_for(nnz)), dim3(kBlock), 0, s, other, perm, nnz, col);
    return check_launch("gather_col");
}

extern "C" int pygsd_gather_f32(const float* src, const int32_t* perm, int64_t n, float* out, void* stream)
{
    PYGSD_REQUIRE(n >= 0, "pygsd_gather_f32: negative size");
    if (n == 0) return 0;
    PYGSD_REQUIRE(src && perm && out, "pygsd_gather_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    hipLaunchKernelGGL(gather_f32, dim3(grid_for(n)), dim3(kBlock), 0, s, src, perm, n, out);
    return check_launch("gather_f32");
}

extern "C" int pygsd_sort_keys_u64_workspace(int64_t n, size_t* bytes)
{
    PYGSD_REQUIRE(bytes, "pygsd_sort_keys_u64_workspace: null output");
    PYGSD_REQUIRE(n >= 0 && n < (int64_t(1) << 31), "pygsd_sort_keys_u64_workspace: n out of range");
    SortWs w;
    if (int rc = sort_ws_layout(n > 0 ? n : 1, &w)) return rc;
    *bytes = w.total;
    return 0;
}

extern "C" int pygsd_sort_keys_u64(const uint64_t* keys_in, uint64_t* keys_out, int32_t* perm_out, int64_t n,
                                   int32_t key_bits, void* workspace, size_t workspace_bytes, void* stream)
{
    PYGSD_REQUIRE(n >= 0 && n < (int64_t(1) << 31), "pygsd_sort_keys_u64: n out of range");
    PYGSD_REQUIRE(key_bits >= 1 && key_bits <= 64, "pygsd_sort_keys_u64: key_bits must be in [1, 64]");
    if (n == 0) return 0;
    PYGSD_REQUIRE(keys_in && keys_out && perm_out && workspace, "pygsd_sort_keys_u64: null pointer");
    SortWs w;
    if (int rc = sort_ws_layout(n, &w)) return rc;
    PYGSD_REQUIRE(workspace_bytes >= w.total, "pygsd_sort_keys_u64: workspace too small (%zu < %zu)",
                  workspace_bytes, w.total);
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    char* base = align256(workspace);
    uint32_t* ids = reinterpret_cast<uint32_t*>(base + w.ids);
    size_t temp_bytes = w.temp_bytes;
    hipLaunchKernelGGL(iota_u32, dim3(grid_for(n)), dim3(kBlock), 0, s, ids, n);
    if (int rc = check_launch("iota_u32")) return rc;
    PYGSD_HIP_TRY(rocprim::radix_sort_pairs(base + w.temp, temp_bytes, keys_in, keys_out, ids,
                                            reinterpret_cast<uint32_t*>(perm_out), static_cast<size_t>(n), 0u,
                                            static_cast<unsigned>(key_bits), s));
    return 0;
}

extern "C" int pygsd_complex_relu_f32(const float* real, const float* imag, float* out_real, float* out_imag,
                                      int64_t n, void* stream)
{
    PYGSD_REQUIRE(n >= 0, "pygsd_complex_relu_f32: negative size");
    if (n == 0) return 0;
    PYGSD_REQUIRE(real && imag && out_real && out_imag, "pygsd_complex_relu_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    if (n % 4 == 0 && aligned16(real) && aligned16(imag) && aligned16(out_real) && aligned16(out_imag)) {
        hipLaunchKernelGGL(complex_relu_vec_kernel, dim3(grid_for(n / 4)), dim3(kBlock), 0, s,
                           reinterpret_cast<const float4*>(real), reinterpret_cast<const float4*>(imag),
                           reinterpret_cast<float4*>(out_real), reinterpret_cast<float4*>(out_imag), n / 4);
    } else {
        hipLaunchKernelGGL(complex_relu_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, real, imag, out_real,
                           out_imag, n);
    }
    return check_launch("complex_relu_kernel");
}

extern "C" int pygsd_complex_relu_bwd_f32(const float* real, const float* g_real, const float* g_imag,
                                          float* gi_real, float* gi_imag, int64_t n, void* stream)
{
    PYGSD_REQUIRE(n >= 0, "pygsd_complex_relu_bwd_f32: negative size");
    if (n == 0) return 0;
    PYGSD_REQUIRE(real && g_real && g_imag && gi_real && gi_imag, "pygsd_complex_relu_bwd_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    hipLaunchKernelGGL(complex_relu_bwd_kernel, dim3(grid_for(n)), dim3(kBlock), 0, s, real, g_real, g_imag,
                       gi_real, gi_imag, n);
    return check_launch("complex_relu_bwd_kernel");
}
