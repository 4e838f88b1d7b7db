// Generic fp32 GEMM  C[M, N] (+)= A[M, K] B[K, N] (+ bias)  for ANY shapes and strides -- the catch-all behind the MFMA
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
    const float* a;
    const float* b;
    const float* bias;
    float* c;
    float* partial;          // [splits][M][N] when splits > 1
    int64_t sa_m, sa_k, sb_k, sb_n, ldc;
    int64_t m, n, k, k_per_split;
    int32_t splits, accumulate;
};

constexpr int kTile = 64, kStep = 16;

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs p)
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
            as[ki][mi] = (gm < p.m && gk < k_hi) ? p.a[gm * p.sa_m + gk * p.sa_k] : 0.f;
            int ni, kj;
            if (b_along_n) { kj = t >> 4; ni = (t & 15) * 4 + e; } else { ni = t >> 2; kj = (t & 3) * 4 + e; }
            const int64_t gn = n0 + ni, gk2 = k0 + kj;
            bs[kj][ni] = (gn < p.n && gk2 < k_hi) ? p.b[gk2 * p.sb_k + gn * p.sb_n] : 0.f;
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
                float v = acc[i][j] + (p.bias ? p.bias[gn] : 0.f);
                if (p.accumulate) v += p.c[gm * p.ldc + gn];
                p.c[gm * p.ldc + gn] = v;
            }
        }
    }
}

// C = (accumulate ? C : 0) + bias + sum_s partial[s], s in order
__global__ __launch_bounds__(256) void gemm_finish_kernel(GemmArgs p)
{
    const int64_t e = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (e >= p.m * p.n) return;
    const int64_t gm = e / p.n, gn = e - gm * p.n;
    float v = 0.f;
    for (int s = 0; s < p.splits; ++s) v += p.partial[static_cast<int64_t>(s) * p.m * p.n + e];
    if (p.bias) v += p.bias[gn];
    if (p.accumulate) v += p.c[gm * p.ldc + gn];
    p.c[gm * p.ldc + gn] = v;
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
    GemmArgs g{a, b, bias, c, static_cast<float*>(workspace), sa_m, sa_k, sb_k, sb_n, ldc, m, n, k, 0, splits, accumulate};
    const int64_t steps = (k + kStep - 1) / kStep;
    g.k_per_split = ((steps + splits - 1) / splits) * kStep;
    if (g.k_per_split == 0) g.k_per_split = kStep;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_DENSE, s);
    const dim3 grid(static_cast<unsigned>((m + kTile - 1) / kTile), static_cast<unsigned>((n + kTile - 1) / kTile),
                    static_cast<unsigned>(splits));
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, s, g);
    if (int rc = check_launch("gemm_f32_kernel")) return rc;
    if (splits > 1) {
        hipLaunchKernelGGL(gemm_finish_kernel, dim3(static_cast<unsigned>((m * n + 255) / 256)), dim3(256), 0, s, g);
        return check_launch("gemm_finish_kernel");
    }
    return 0;
}
