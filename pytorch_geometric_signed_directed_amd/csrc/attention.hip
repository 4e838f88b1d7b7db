// Attention aggregate of SDGNN / SiGAT: the arithmetic the reference reaches
// through torch_geometric.nn.GATConv (nn/signed/SDGNN.py:35-41,57-64; nn/signed/SiGAT.py:59-64):
//   e_ij = leaky_relu(a_src[j] + a_dst[i]);  alpha_ij = softmax over the incoming edges of i
//   (max-shifted, denominator + 1e-16);      out_i = sum_j alpha_ij h_j.
// Same traversal as the SpMM: one wavefront owns a target row of the by-target CSR.  The softmax
// coefficients are produced once per (row, edge) into an [nnz] array in CSR order; the weighted sum then
// IS pygsd_spmm_csr_f32 with those values.  Backward: per edge d_alpha = <g_i, h_j> (an SDDMM), folded
// with the softmax and leaky-relu derivatives into ds_ij = alpha_ij (d_alpha - <g_i, out_i>) * lrelu'(s_ij).
#include "common.hpp"

namespace pygsd {
namespace {

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void gat_alpha_kernel(const int32_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col,
                                                        const float* __restrict__ a_src,
                                                        const float* __restrict__ a_dst, int32_t n_rows,
                                                        float slope, float* __restrict__ alpha)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    const float ad = a_dst[row];
    float mx = -INFINITY;
    for (int e = beg + lane; e < end; e += 64) {
        const float s = a_src[col[e]] + ad;
        mx = fmaxf(mx, s > 0.f ? s : slope * s);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int e = beg + lane; e < end; e += 64) {
        const float s = a_src[col[e]] + ad;
        sum += expf((s > 0.f ? s : slope * s) - mx);
    }
    sum = wave_sum(sum) + 1e-16f;
    for (int e = beg + lane; e < end; e += 64) {
        const float s = a_src[col[e]] + ad;
        alpha[e] = expf((s > 0.f ? s : slope * s) - mx) / sum;
    }
}

// one 16-lane team per edge (4 edges per wavefront pass); writes ds and alpha in COO order
__global__ __launch_bounds__(256) void gat_alpha_bwd_kernel(const int32_t* __restrict__ rowptr,
                                                            const int32_t* __restrict__ col,
                                                            const int32_t* __restrict__ perm,
                                                            const float* __restrict__ a_src,
                                                            const float* __restrict__ a_dst, float slope,
                                                            const float* __restrict__ alpha,
                                                            const float* __restrict__ h, int64_t ldh,
                                                            const float* __restrict__ g, int64_t ldg,
                                                            const float* __restrict__ out, int64_t ldo,
                                                            int32_t n_rows, int32_t n_feat,
                                                            float* __restrict__ ds_coo, float* __restrict__ alpha_coo)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    const float* gi = g + static_cast<int64_t>(row) * ldg;
    const float* oi = out + static_cast<int64_t>(row) * ldo;
    float rd = 0.f;
    for (int f = lane; f < n_feat; f += 64) rd = fmaf(gi[f], oi[f], rd);
    rd = wave_sum(rd);                                   // <g_i, out_i>
    const float ad = a_dst[row];
    const int t = lane & 15, team = lane >> 4;
    for (int e0 = beg; e0 < end; e0 += 4) {
        const int e = e0 + team;
        float dot = 0.f;
        int cj = 0;
        if (e < end) {
            cj = col[e];
            const float* hj = h + static_cast<int64_t>(cj) * ldh;
            for (int f = t; f < n_feat; f += 16) dot = fmaf(gi[f], hj[f], dot);
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) dot += __shfl_xor(dot, off);
        if (e < end && t == 0) {
            const float a = alpha[e];
            const float s = a_src[cj] + ad;
            const float ds = a * (dot - rd) * (s > 0.f ? 1.f : slope);
            const int p = perm[e];
            ds_coo[p] = ds;
            alpha_coo[p] = a;
        }
    }
}

// Vectorised variant (F % 4 == 0, 16-byte aligned rows, F <= 256): LPR = F/4 lanes hold one float4 each of a
// feature row, so a wave-wide load fetches 64/LPR neighbour rows, UN of them in flight per lane before the
// dot products (same gather shape as spmm_vec_kernel).  ds is written in CSR order (coalesced; the by-source
// sums go through the pattern's by-source -> by-target slot map) and its row sum -- the gradient of a_dst --
// is produced here.
template <int LPR>
__global__ __launch_bounds__(256) void gat_alpha_bwd_vec_kernel(const int32_t* __restrict__ rowptr,
                                                                const int32_t* __restrict__ col,
                                                                const float* __restrict__ a_src,
                                                                const float* __restrict__ a_dst, float slope,
                                                                const float* __restrict__ alpha,
                                                                const float* __restrict__ h, int64_t ldh,
                                                                const float* __restrict__ g, int64_t ldg,
                                                                const float* __restrict__ out, int64_t ldo,
                                                                int32_t n_rows, int32_t n_feat,
                                                                float* __restrict__ ds, float* __restrict__ da_dst)
{
    constexpr int NPW = 64 / LPR;
    constexpr int UN = 4;
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    const int sub = lane / LPR;
    const int fl = (lane % LPR) * 4;
    const bool fact = fl < n_feat;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float rd = 0.f;
    if (fact) {
        g4 = *reinterpret_cast<const float4*>(g + static_cast<int64_t>(row) * ldg + fl);
        const float4 o4 = *reinterpret_cast<const float4*>(out + static_cast<int64_t>(row) * ldo + fl);
        rd = g4.x * o4.x + g4.y * o4.y + g4.z * o4.z + g4.w * o4.w;
    }
#pragma unroll
    for (int off = 1; off < LPR; off <<= 1) rd += __shfl_xor(rd, off);          // <g_i, out_i> in every lane
    const float ad = a_dst[row];
    float acc = 0.f;
    for (int base = beg; base < end; base += 64) {
        const int cnt = (end - base) < 64 ? (end - base) : 64;
        int c = 0;
        float al = 0.f;
        if (lane < cnt) {
            c = col[base + lane];
            al = alpha[base + lane];
        }
        for (int u = 0; u < cnt; u += NPW * UN) {
            float4 hv[UN];
            int cj[UN];
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const int idx = u + k * NPW + sub;
                cj[k] = __shfl(c, idx & 63);
                hv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (fact && idx < cnt) hv[k] = *reinterpret_cast<const float4*>(h + static_cast<int64_t>(cj[k]) * ldh + fl);
            }
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const int idx = u + k * NPW + sub;
                float dot = g4.x * hv[k].x + g4.y * hv[k].y + g4.z * hv[k].z + g4.w * hv[k].w;
#pragma unroll
                for (int off = 1; off < LPR; off <<= 1) dot += __shfl_xor(dot, off);
                const float a = __shfl(al, idx & 63);
                if (idx < cnt && (lane % LPR) == 0) {
                    const float sc = a_src[cj[k]] + ad;
                    const float d = a * (dot - rd) * (sc > 0.f ? 1.f : slope);
                    ds[base + idx] = d;
                    acc += d;
                }
            }
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) da_dst[row] = acc;
}

// out[r] = sum over CSR row r of w[perm[slot]] (perm == NULL: w[slot]); 16-lane teams, fixed (not sequential)
// summation order -- for gradients, where pygsd_csr_row_sum_f32's reference scatter order is not required.
__global__ __launch_bounds__(256) void segment_sum_kernel(const int32_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ perm,
                                                          const float* __restrict__ w, int32_t n_rows,
                                                          float* __restrict__ out)
{
    const int t = threadIdx.x & 15;
    const int64_t row = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 4;
    float acc = 0.f;
    if (row < n_rows) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        for (int e = beg + t; e < end; e += 16) acc += w[perm ? perm[e] : e];
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (row < n_rows && t == 0) out[row] = acc;
}

// Generic segment softmax over per-entry logits already in CSR order (SNEAConv's tanh attention,
// nn/signed/SNEAConv.py:135-146: alpha = softmax(tanh(lin([x_j, x_i])), index)): max-shifted, denominator
// + 1e-16 like torch_geometric.utils.softmax.  One wavefront per segment, three coalesced passes.
__global__ __launch_bounds__(256) void segment_softmax_kernel(const int32_t* __restrict__ rowptr,
                                                              const float* __restrict__ logits, int32_t n_rows,
                                                              float* __restrict__ alpha)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    float mx = -INFINITY;
    for (int e = beg + lane; e < end; e += 64) mx = fmaxf(mx, logits[e]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int e = beg + lane; e < end; e += 64) sum += expf(logits[e] - mx);
    sum = wave_sum(sum) + 1e-16f;
    for (int e = beg + lane; e < end; e += 64) alpha[e] = expf(logits[e] - mx) / sum;
}

// d logits = alpha * (d alpha - sum_segment alpha * d alpha)
__global__ __launch_bounds__(256) void segment_softmax_bwd_kernel(const int32_t* __restrict__ rowptr,
                                                                  const float* __restrict__ alpha,
                                                                  const float* __restrict__ dalpha, int32_t n_rows,
                                                                  float* __restrict__ dlogits)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    float dot = 0.f;
    for (int e = beg + lane; e < end; e += 64) dot = fmaf(alpha[e], dalpha[e], dot);
    dot = wave_sum(dot);
    for (int e = beg + lane; e < end; e += 64) dlogits[e] = alpha[e] * (dalpha[e] - dot);
}

// ------------------------------------------------------------------------------------------
// SNEAConv (nn/signed/SNEAConv.py:135-146), fused.  Per slot e of target row i with source j and edge
// type p (0: positive / self loop, 1: negative):  pre = s_p[j] + d_p[i] + bias,  alpha = softmax_i(tanh(pre)).
// The layer's message is the TARGET row times alpha, so all the aggregate needs per row is the share of
// alpha that went to each type:  share_t[i] = sum_{e in row i, p_e = t} alpha_e.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float snea_logit(int e, const int32_t* col, const uint8_t* ptype, const float* s0,
                                            const float* s1, float d0, float d1, float bias, bool& neg)
{
    neg = ptype && ptype[e] != 0;
    const int j = col[e];
    return tanhf((neg ? s1[j] + d1 : s0[j] + d0) + bias);
}

__global__ __launch_bounds__(256) void snea_alpha_kernel(const int32_t* __restrict__ rowptr,
                                                         const int32_t* __restrict__ col,
                                                         const uint8_t* __restrict__ ptype,
                                                         const float* __restrict__ s0, const float* __restrict__ s1,
                                                         const float* __restrict__ d0, const float* __restrict__ d1,
                                                         const float* __restrict__ bias_p, int32_t n_rows,
                                                         float* __restrict__ alpha,
                                                         float* __restrict__ share0, float* __restrict__ share1)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    const float di0 = d0[row], di1 = ptype ? d1[row] : 0.f;
    const float bias = bias_p ? bias_p[0] : 0.f;
    bool neg;
    float mx = -INFINITY;
    for (int e = beg + lane; e < end; e += 64)
        mx = fmaxf(mx, snea_logit(e, col, ptype, s0, s1, di0, di1, bias, neg));
    mx = wave_max(mx);
    float sum = 0.f;
    for (int e = beg + lane; e < end; e += 64)
        sum += expf(snea_logit(e, col, ptype, s0, s1, di0, di1, bias, neg) - mx);
    sum = wave_sum(sum) + 1e-16f;
    float a0 = 0.f, a1 = 0.f;
    for (int e = beg + lane; e < end; e += 64) {
        const float a = expf(snea_logit(e, col, ptype, s0, s1, di0, di1, bias, neg) - mx) / sum;
        alpha[e] = a;
        if (neg) a1 += a; else a0 += a;
    }
    a0 = wave_sum(a0);
    a1 = wave_sum(a1);
    if (lane == 0) {
        share0[row] = a0;
        if (share1) share1[row] = a1;
    }
}

// Given d share_t[i]: d alpha_e = d share_{p_e}[i]; softmax and tanh backward give d pre_e, written per type
// in CSR order (dpre0 / dpre1, zero where the slot has the other type) for the by-source sums, and summed per
// row into dd0 / dd1 (the gradient of d_p[i]).
__global__ __launch_bounds__(256) void snea_alpha_bwd_kernel(const int32_t* __restrict__ rowptr,
                                                             const int32_t* __restrict__ col,
                                                             const uint8_t* __restrict__ ptype,
                                                             const float* __restrict__ s0, const float* __restrict__ s1,
                                                             const float* __restrict__ d0, const float* __restrict__ d1,
                                                             const float* __restrict__ bias_p,
                                                             const float* __restrict__ alpha,
                                                             const float* __restrict__ dshare0,
                                                             const float* __restrict__ dshare1, int32_t n_rows,
                                                             float* __restrict__ dpre0, float* __restrict__ dpre1,
                                                             float* __restrict__ dd0, float* __restrict__ dd1)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    const float di0 = d0[row], di1 = ptype ? d1[row] : 0.f;
    const float g0 = dshare0[row], g1 = ptype ? dshare1[row] : 0.f;
    const float bias = bias_p ? bias_p[0] : 0.f;
    float dot = 0.f;
    for (int e = beg + lane; e < end; e += 64) dot = fmaf(alpha[e], (ptype && ptype[e]) ? g1 : g0, dot);
    dot = wave_sum(dot);
    float r0 = 0.f, r1 = 0.f;
    for (int e = beg + lane; e < end; e += 64) {
        bool neg;
        const float t = snea_logit(e, col, ptype, s0, s1, di0, di1, bias, neg);
        const float dp = alpha[e] * ((neg ? g1 : g0) - dot) * (1.f - t * t);
        dpre0[e] = neg ? 0.f : dp;
        if (dpre1) dpre1[e] = neg ? dp : 0.f;
        if (neg) r1 += dp; else r0 += dp;
    }
    r0 = wave_sum(r0);
    r1 = wave_sum(r1);
    if (lane == 0) {
        dd0[row] = r0;
        if (dd1) dd1[row] = r1;
    }
}

}  // namespace
}  // namespace pygsd

using namespace pygsd;

extern "C" int pygsd_snea_alpha_csr_f32(const int32_t* rowptr, const int32_t* col, const uint8_t* edge_type,
                                        const float* s0, const float* s1, const float* d0, const float* d1,
                                        const float* bias, int32_t n_rows, float* alpha, float* share0,
                                        float* share1, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_snea_alpha_csr_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && s0 && d0 && share0, "pygsd_snea_alpha_csr_f32: null pointer");
    PYGSD_REQUIRE(!edge_type || (s1 && d1 && share1), "pygsd_snea_alpha_csr_f32: typed edges need s1, d1, share1");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    hipLaunchKernelGGL(snea_alpha_kernel, dim3((static_cast<unsigned>(n_rows) + 3) / 4), dim3(256), 0, s, rowptr, col,
                       edge_type, s0, s1, d0, d1, bias, n_rows, alpha, share0, share1);
    return check_launch("snea_alpha_kernel");
}

extern "C" int pygsd_snea_alpha_bwd_csr_f32(const int32_t* rowptr, const int32_t* col, const uint8_t* edge_type,
                                            const float* s0, const float* s1, const float* d0, const float* d1,
                                            const float* bias, const float* alpha, const float* dshare0,
                                            const float* dshare1, int32_t n_rows, float* dpre0, float* dpre1,
                                            float* dd0, float* dd1, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_snea_alpha_bwd_csr_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && s0 && d0 && dshare0 && dd0, "pygsd_snea_alpha_bwd_csr_f32: null pointer");
    PYGSD_REQUIRE(!edge_type || (s1 && d1 && dshare1 && dpre1 && dd1),
                  "pygsd_snea_alpha_bwd_csr_f32: typed edges need the type-1 arrays");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    hipLaunchKernelGGL(snea_alpha_bwd_kernel, dim3((static_cast<unsigned>(n_rows) + 3) / 4), dim3(256), 0, s, rowptr,
                       col, edge_type, s0, s1, d0, d1, bias, alpha, dshare0, dshare1, n_rows, dpre0, dpre1, dd0, dd1);
    return check_launch("snea_alpha_bwd_kernel");
}

extern "C" int pygsd_gat_alpha_bwd_csr_v2_f32(const int32_t* rowptr, const int32_t* col, const float* a_src,
                                              const float* a_dst, float negative_slope, const float* alpha,
                                              const float* h, int64_t ldh, const float* g, int64_t ldg,
                                              const float* out, int64_t ldo, int32_t n_rows, int32_t n_feat,
                                              float* ds_csr, float* da_dst, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0 && n_feat >= 0, "pygsd_gat_alpha_bwd_csr_v2_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && a_src && a_dst && h && g && out && da_dst, "pygsd_gat_alpha_bwd_csr_v2_f32: null pointer");
    PYGSD_REQUIRE(n_feat % 4 == 0 && n_feat <= 256 && ldh % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0 && aligned16(h) &&
                      aligned16(g) && aligned16(out),
                  "pygsd_gat_alpha_bwd_csr_v2_f32: needs n_feat %% 4 == 0, n_feat <= 256 and 16-byte aligned rows");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_SDDMM, s);
    const dim3 grid((static_cast<unsigned>(n_rows) + 3) / 4), block(256);
#define PYGSD_GAT_BWD(L)                                                                                             \
    hipLaunchKernelGGL(gat_alpha_bwd_vec_kernel<L>, grid, block, 0, s, rowptr, col, a_src, a_dst, negative_slope, alpha, \
                       h, ldh, g, ldg, out, ldo, n_rows, n_feat, ds_csr, da_dst)
    const int quads = n_feat / 4;
    if (quads <= 4) PYGSD_GAT_BWD(4);
    else if (quads <= 8) PYGSD_GAT_BWD(8);
    else if (quads <= 16) PYGSD_GAT_BWD(16);
    else if (quads <= 32) PYGSD_GAT_BWD(32);
    else PYGSD_GAT_BWD(64);
#undef PYGSD_GAT_BWD
    return check_launch("gat_alpha_bwd_vec_kernel");
}

extern "C" int pygsd_segment_sum_f32(const int32_t* rowptr, const int32_t* perm, const float* w, int32_t n_rows,
                                     float* out, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_segment_sum_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && out, "pygsd_segment_sum_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    const int64_t threads = static_cast<int64_t>(n_rows) * 16;
    hipLaunchKernelGGL(segment_sum_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, s, rowptr,
                       perm, w, n_rows, out);
    return check_launch("segment_sum_kernel");
}

extern "C" int pygsd_segment_softmax_csr_f32(const int32_t* rowptr, const float* logits, int32_t n_rows, float* alpha,
                                             void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_segment_softmax_csr_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr, "pygsd_segment_softmax_csr_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    hipLaunchKernelGGL(segment_softmax_kernel, dim3((static_cast<unsigned>(n_rows) + 3) / 4), dim3(256), 0, s, rowptr,
                       logits, n_rows, alpha);
    return check_launch("segment_softmax_kernel");
}

extern "C" int pygsd_segment_softmax_bwd_csr_f32(const int32_t* rowptr, const float* alpha, const float* dalpha,
                                                 int32_t n_rows, float* dlogits, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_segment_softmax_bwd_csr_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr, "pygsd_segment_softmax_bwd_csr_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    hipLaunchKernelGGL(segment_softmax_bwd_kernel, dim3((static_cast<unsigned>(n_rows) + 3) / 4), dim3(256), 0, s,
                       rowptr, alpha, dalpha, n_rows, dlogits);
    return check_launch("segment_softmax_bwd_kernel");
}

extern "C" int pygsd_gat_alpha_csr_f32(const int32_t* rowptr, const int32_t* col, const float* a_src,
                                       const float* a_dst, int32_t n_rows, float negative_slope, float* alpha,
                                       void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_gat_alpha_csr_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && a_src && a_dst, "pygsd_gat_alpha_csr_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_SPMM, s);
    hipLaunchKernelGGL(gat_alpha_kernel, dim3((static_cast<unsigned>(n_rows) + 3) / 4), dim3(256), 0, s, rowptr,
                       col, a_src, a_dst, n_rows, negative_slope, alpha);
    return check_launch("gat_alpha_kernel");
}

extern "C" int pygsd_gat_alpha_bwd_csr_f32(const int32_t* rowptr, const int32_t* col, const int32_t* perm,
                                           const float* a_src, const float* a_dst, float negative_slope,
                                           const float* alpha, const float* h, int64_t ldh, const float* g,
                                           int64_t ldg, const float* out, int64_t ldo, int32_t n_rows,
                                           int32_t n_feat, float* ds_coo, float* alpha_coo, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0 && n_feat >= 0, "pygsd_gat_alpha_bwd_csr_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && a_src && a_dst && h && g && out, "pygsd_gat_alpha_bwd_csr_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_SDDMM, s);
    hipLaunchKernelGGL(gat_alpha_bwd_kernel, dim3((static_cast<unsigned>(n_rows) + 3) / 4), dim3(256), 0, s, rowptr,
                       col, perm, a_src, a_dst, negative_slope, alpha, h, ldh, g, ldg, out, ldo, n_rows, n_feat,
                       ds_coo, alpha_coo);
    return check_launch("gat_alpha_bwd_kernel");
}
