// Generic GEMM  C[M, N] (+)= A[M, K] B[K, N] (+ bias)  in fp32, or bf16 storage with fp32 accumulation (round 6: the same
// kernel with 2-byte loads / stores, so that no bf16 width leaves the library either), for ANY shapes and strides -- the catch-all behind the MFMA
// kernels of csrc/dense.hip / tall.hip / gram.hip, so that no dense product on the path has to leave the library for
// hipBLASLt / rocBLAS: odd widths (a 10-class head), reductions deeper than the LDS-resident W of tall.hip allows (the
// 2879-wide first layer of BASELINE config 1: x W and its weight gradient x^T g), transposed views (strides are arguments).
// Not a speed-of-light kernel and not meant to be: 64 x 64 output tiles, 16-deep k-steps through LDS, a 4 x 4 register
// block per thread of plain fmaf (exact fp32, a fixed summation order).  A reduction that is long against the output
// (K >> M N / 4096: weight gradients over 10^5..10^6 rows) is split over blockIdx.z into k-ranges whose partial products
// are added in range order by a second kernel -- deterministic, no atomics.
#include "common.hpp"

namespace pygsd {
namespace {

struct GemmArgs {
    const void* a;           // fp32, or bf16 (BF_IN) -- A, B and bias alike
    const void* b;
    const void* bias;
    void* c;                 // fp32, or bf16 (BF_OUT)
    const float* z;          // fp32 addend [M][N] at row stride ldz, or null (accumulate onto an fp32 C: z = c)
    float* partial;          // [splits][M][N] when splits > 1
    int64_t sa_m, sa_k, sb_k, sb_n, ldc, ldz;
    int64_t m, n, k, k_per_split;
    int32_t splits;
};

constexpr int kTile = 64, kStep = 16;

template <bool BF>
__device__ __forceinline__ float load_el(const void* base, int64_t at)
{
    if constexpr (BF) return __uint_as_float(static_cast<uint32_t>(static_cast<const uint16_t*>(base)[at]) << 16);
    else return static_cast<const float*>(base)[at];
}

// one rounding to nearest even of the fp32 sum (NaN stays NaN)
__device__ __forceinline__ uint16_t to_bf16(float v)
{
    const uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);
    return static_cast<uint16_t>((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <bool BF>
__device__ __forceinline__ void store_el(void* base, int64_t at, float v)
{
    if constexpr (BF) static_cast<uint16_t*>(base)[at] = to_bf16(v);
    else static_cast<float*>(base)[at] = v;
}

template <bool BF_IN, bool BF_OUT>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p)
{
    __shared__ __attribute__((aligned(16))) float as[kStep][kTile + 4];
    __shared__ __attribute__((aligned(16))) float bs[kStep][kTile + 4];
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kTile, n0 = static_cast<int64_t>(blockIdx.y) * kTile;
    const int64_t k_lo = static_cast<int64_t>(blockIdx.z) * p.k_per_split;
    const int64_t k_hi = (k_lo + p.k_per_split < p.k) ? k_lo + p.k_per_split : p.k;
    // which index runs along the unit stride decides which one the 4 loads of a thread walk (wave-uniform)
    const bool a_along_k = p.sa_k == 1 || p.sa_m != 1;
    const bool b_along_n = p.sb_n == 1 || p.sb_k != 1;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int64_t k0 = k_lo; k0 < k_hi; k0 += kStep) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int mi, ki;
            if (a_along_k) { mi = t >> 2; ki = (t & 3) * 4 + e; } else { ki = t >> 4; mi = (t & 15) * 4 + e; }
            const int64_t gm = m0 + mi, gk = k0 + ki;
            as[ki][mi] = (gm < p.m && gk < k_hi) ? load_el<BF_IN>(p.a, gm * p.sa_m + gk * p.sa_k) : 0.f;
            int ni, kj;
            if (b_along_n) { kj = t >> 4; ni = (t & 15) * 4 + e; } else { ni = t >> 2; kj = (t & 3) * 4 + e; }
            const int64_t gn = n0 + ni, gk2 = k0 + kj;
            bs[kj][ni] = (gn < p.n && gk2 < k_hi) ? load_el<BF_IN>(p.b, gk2 * p.sb_k + gn * p.sb_n) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < kStep; ++kk) {
            const float4 av = *reinterpret_cast<const float4*>(&as[kk][ty * 4]);
            const float4 bv = *reinterpret_cast<const float4*>(&bs[kk][tx * 4]);
            const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a4[i], b4[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t gm = m0 + ty * 4 + i;
        if (gm >= p.m) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t gn = n0 + tx * 4 + j;
            if (gn >= p.n) continue;
            if (p.splits > 1) {
                p.partial[(static_cast<int64_t>(blockIdx.z) * p.m + gm) * p.n + gn] = acc[i][j];
            } else {
                float v = acc[i][j] + (p.bias ? load_el<BF_IN>(p.bias, gn) : 0.f);
                if (p.z) v += p.z[gm * p.ldz + gn];
                store_el<BF_OUT>(p.c, gm * p.ldc + gn, v);
            }
        }
    }
}

// C = (z ? Z : 0) + bias + sum_s partial[s], s in order
template <bool BF_IN, bool BF_OUT>
__global__ __launch_bounds__(256) void gemm_finish_kernel(GemmArgs p)
{
    const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (e >= p.m * p.n) return;
    const int64_t gm = e / p.n, gn = e - gm * p.n;
    float v = 0.f;
    for (int s = 0; s < p.splits; ++s) v += p.partial[static_cast<int64_t>(s) * p.m * p.n + e];
    if (p.bias) v += load_el<BF_IN>(p.bias, gn);
    if (p.z) v += p.z[gm * p.ldz + gn];
    store_el<BF_OUT>(p.c, gm * p.ldc + gn, v);
}

int pick_splits(int64_t m, int64_t n, int64_t k)
{
    const int64_t tiles = ((m + kTile - 1) / kTile) * ((n + kTile - 1) / kTile);
    if (tiles >= 512 || k < 4096) return 1;                 // the output alone fills the chip / a short reduction
    int64_t s = (1024 + tiles - 1) / tiles;                 // ~4 blocks per CU
    const int64_t most = (k + 1023) / 1024;                 // at least 1024 k-steps' worth per split
    if (s > most) s = most;
    return static_cast<int>(s < 1 ? 1 : s);
}
}  // namespace
}  // namespace pygsd

using namespace pygsd;

extern "C" int pygsd_gemm_f32_workspace(int64_t m, int64_t n, int64_t k, size_t* bytes)
{
    PYGSD_REQUIRE(bytes, "pygsd_gemm_f32_workspace: null output");
    PYGSD_REQUIRE(m >= 0 && n >= 0 && k >= 0, "pygsd_gemm_f32_workspace: negative size");
    const int s = pick_splits(m, n, k);
    *bytes = s > 1 ? static_cast<size_t>(s) * m * n * sizeof(float) : 0;
    return 0;
}

namespace {
template <bool BF_IN, bool BF_OUT>
int launch_gemm(GemmArgs& g, hipStream_t s)
{
    const int64_t steps = (g.k + kStep - 1) / kStep;
    g.k_per_split = ((steps + g.splits - 1) / g.splits) * kStep;
    if (g.k_per_split == 0) g.k_per_split = kStep;
    ProfScope prof(PYGSD_K_DENSE, s);
    const dim3 grid(static_cast<unsigned>((g.m + kTile - 1) / kTile), static_cast<unsigned>((g.n + kTile - 1) / kTile),
                    static_cast<unsigned>(g.splits));
    hipLaunchKernelGGL((gemm_kernel<BF_IN, BF_OUT>), grid, dim3(256), 0, s, g);
    if (int rc = check_launch("gemm_kernel")) return rc;
    if (g.splits > 1) {
        hipLaunchKernelGGL((gemm_finish_kernel<BF_IN, BF_OUT>), dim3(static_cast<unsigned>((g.m * g.n + 255) / 256)), dim3(256), 0,
                           s, g);
        return check_launch("gemm_finish_kernel");
    }
    return 0;
}
}  // namespace

extern "C" int pygsd_gemm_f32(const float* a, int64_t sa_m, int64_t sa_k, const float* b, int64_t sb_k, int64_t sb_n,
                              const float* bias, float* c, int64_t ldc, int64_t m, int64_t n, int64_t k, int32_t accumulate,
                              void* workspace, size_t workspace_bytes, void* stream)
{
    PYGSD_REQUIRE(m >= 0 && n >= 0 && k >= 0, "pygsd_gemm_f32: negative size");
    if (m == 0 || n == 0) return 0;
    PYGSD_REQUIRE(c && ldc >= n, "pygsd_gemm_f32: null output or ldc < n");
    PYGSD_REQUIRE(k == 0 || (a && b), "pygsd_gemm_f32: null operand");
    PYGSD_REQUIRE((m + kTile - 1) / kTile < (1ll << 31) && (n + kTile - 1) / kTile < 65536, "pygsd_gemm_f32: output too large");
    const int splits = pick_splits(m, n, k);
    const size_t need = splits > 1 ? static_cast<size_t>(splits) * m * n * sizeof(float) : 0;
    PYGSD_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), "pygsd_gemm_f32: workspace too small "
                  "(pygsd_gemm_f32_workspace)");
    GemmArgs g{a, b, bias, c, accumulate ? c : nullptr, static_cast<float*>(workspace), sa_m, sa_k, sb_k, sb_n, ldc, ldc,
               m, n, k, 0, splits};
    return launch_gemm<false, false>(g, static_cast<hipStream_t>(stream));
}

extern "C" int pygsd_gemm_bf16(const void* a, int64_t sa_m, int64_t sa_k, const void* b, int64_t sb_k, int64_t sb_n,
                               const void* bias, void* c, int64_t ldc, int32_t c_is_f32, const float* z, int64_t ldz, int64_t m,
                               int64_t n, int64_t k, void* workspace, size_t workspace_bytes, void* stream)
{
    PYGSD_REQUIRE(m >= 0 && n >= 0 && k >= 0, "pygsd_gemm_bf16: negative size");
    if (m == 0 || n == 0) return 0;
    PYGSD_REQUIRE(c && ldc >= n, "pygsd_gemm_bf16: null output or ldc < n");
    PYGSD_REQUIRE(k == 0 || (a && b), "pygsd_gemm_bf16: null operand");
    PYGSD_REQUIRE(!z || ldz >= n, "pygsd_gemm_bf16: addend row stride < n");
    PYGSD_REQUIRE((m + kTile - 1) / kTile < (1ll << 31) && (n + kTile - 1) / kTile < 65536, "pygsd_gemm_bf16: output too large");
    const int splits = pick_splits(m, n, k);
    const size_t need = splits > 1 ? static_cast<size_t>(splits) * m * n * sizeof(float) : 0;
    PYGSD_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), "pygsd_gemm_bf16: workspace too small "
                  "(pygsd_gemm_f32_workspace)");
    GemmArgs g{a, b, bias, c, z, static_cast<float*>(workspace), sa_m, sa_k, sb_k, sb_n, ldc, ldz, m, n, k, 0, splits};
    hipStream_t s = static_cast<hipStream_t>(stream);
    return c_is_f32 ? launch_gemm<true, false>(g, s) : launch_gemm<true, true>(g, s);
}
