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
                                                        float slope, float* __restrict__ alpha, int32_t skip)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    if (skip > 0 && end - beg > skip) return;            // hub row: segment-parallel path below
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
                                                            float* __restrict__ ds_coo, float* __restrict__ alpha_coo,
                                                            int32_t skip)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    if (skip > 0 && end - beg > skip) return;
    const float* gi = g + static_cast<int64_t>(row) * ldg;
    // ds_e = alpha_e (d alpha_e - sum_k alpha_k d alpha_k) slope'(s_e), with d alpha_e = <g_i, h_e>, in the reference's own form
    // (autograd of torch_geometric.utils.softmax): the row's sum is taken over the SAME d alpha values the entries use, so a row
    // with one entry gets exactly zero and the row sums of ds (the gradient of a_dst) cancel as the reference's do.  Until round 6
    // the sum was <g_i, out_i> -- equal in exact arithmetic, but out_i carries its own rounding: 4 x the reference's error in the
    // attention vectors' gradients on rows of 1 - 2 entries (tests/test_gpu_fuzz.py).  Pass 1 parks d alpha_e in ds_coo (every
    // slot is read back by the lane that wrote it), pass 2 finishes; `out` is no longer read by this kernel.
    (void)out; (void)ldo;
    const int t = lane & 15, team = lane >> 4;
    const float ad = a_dst[row];
    float sd = 0.f;
    for (int e0 = beg; e0 < end; e0 += 4) {
        const int e = e0 + team;
        float dot = 0.f;
        if (e < end) {
            const float* hj = h + static_cast<int64_t>(col[e]) * ldh;
            for (int f = t; f < n_feat; f += 16) dot = fmaf(gi[f], hj[f], dot);
        }
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) dot += __shfl_xor(dot, off);
        if (e < end && t == 0) {
            ds_coo[perm[e]] = dot;
            sd = fmaf(alpha[e], dot, sd);
        }
    }
    sd = wave_sum(sd);
    for (int e0 = beg; e0 < end; e0 += 4) {
        const int e = e0 + team;
        if (e < end && t == 0) {
            const float a = alpha[e];
            const float s = a_src[col[e]] + ad;
            const int p = perm[e];
            ds_coo[p] = a * (ds_coo[p] - sd) * (s > 0.f ? 1.f : slope);
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
                                                                float* __restrict__ ds, float* __restrict__ da_dst,
                                                                int32_t skip)
{
    constexpr int NPW = 64 / LPR;
    constexpr int UN = 4;
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    if (skip > 0 && end - beg > skip) return;
    const int sub = lane / LPR;
    const int fl = (lane % LPR) * 4;
    const bool fact = fl < n_feat;
    float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fact) g4 = *reinterpret_cast<const float4*>(g + static_cast<int64_t>(row) * ldg + fl);
    (void)out; (void)ldo;
    const float ad = a_dst[row];
    const bool writer = (lane % LPR) == 0;
    // pass 1: d alpha_e = <g_i, h_e> parked in ds[e], and the row's sum_k alpha_k d alpha_k over those same values (see
    // gat_alpha_bwd_kernel: the reference's form; every ds slot is read back in pass 2 by the lane that wrote it)
    float sd = 0.f;
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
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const int idx = u + k * NPW + sub;
                const int cj = __shfl(c, idx & 63);
                hv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (fact && idx < cnt) hv[k] = *reinterpret_cast<const float4*>(h + static_cast<int64_t>(cj) * ldh + fl);
            }
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const int idx = u + k * NPW + sub;
                float dot = g4.x * hv[k].x + g4.y * hv[k].y + g4.z * hv[k].z + g4.w * hv[k].w;
#pragma unroll
                for (int off = 1; off < LPR; off <<= 1) dot += __shfl_xor(dot, off);
                const float a = __shfl(al, idx & 63);
                if (idx < cnt && writer) {
                    ds[base + idx] = dot;
                    sd = fmaf(a, dot, sd);
                }
            }
        }
    }
    sd = wave_sum(sd);
    // pass 2: ds_e = alpha_e (d alpha_e - sum) slope'(s_e) and its row sum, the gradient of a_dst
    float acc = 0.f;
    for (int base = beg; base < end; base += 64) {
        const int cnt = (end - base) < 64 ? (end - base) : 64;
        int c = 0;
        float al = 0.f;
        if (lane < cnt) {
            c = col[base + lane];
            al = alpha[base + lane];
        }
        for (int u = 0; u < cnt; u += NPW) {
            const int idx = u + sub;
            const float a = __shfl(al, idx & 63);
            const int cj = __shfl(c, idx & 63);
            if (idx < cnt && writer) {
                const float sc = a_src[cj] + ad;
                const float d = a * (ds[base + idx] - sd) * (sc > 0.f ? 1.f : slope);
                ds[base + idx] = d;
                acc += d;
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
                                                          float* __restrict__ out, int32_t skip)
{
    const int t = threadIdx.x & 15;
    const int64_t row = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 4;
    float acc = 0.f;
    bool mine = row < n_rows;
    if (mine) {
        const int beg = rowptr[row], end = rowptr[row + 1];
        mine = !(skip > 0 && end - beg > skip);
        if (mine)
            for (int e = beg + t; e < end; e += 16) acc += w[perm ? perm[e] : e];
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (mine && t == 0) out[row] = acc;
}

// Generic segment softmax over per-entry logits already in CSR order (SNEAConv's tanh attention,
// nn/signed/SNEAConv.py:135-146: alpha = softmax(tanh(lin([x_j, x_i])), index)): max-shifted, denominator
// + 1e-16 like torch_geometric.utils.softmax.  One wavefront per segment, three coalesced passes.
__global__ __launch_bounds__(256) void segment_softmax_kernel(const int32_t* __restrict__ rowptr,
                                                              const float* __restrict__ logits, int32_t n_rows,
                                                              float* __restrict__ alpha, int32_t skip)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    if (skip > 0 && end - beg > skip) return;
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
                                                                  float* __restrict__ dlogits, int32_t skip)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    if (skip > 0 && end - beg > skip) return;
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
                                                         float* __restrict__ share0, float* __restrict__ share1,
                                                         int32_t skip)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    if (skip > 0 && end - beg > skip) return;
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
                                                             float* __restrict__ dd0, float* __restrict__ dd1,
                                                             int32_t skip)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * 4 + static_cast<int>(threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int beg = rowptr[row], end = rowptr[row + 1];
    if (skip > 0 && end - beg > skip) return;
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


// ------------------------------------------------------------------------------------------
// Hub rows.  One wavefront per row is the right shape for bounded degrees and a cliff on power-law graphs
// (SDGNN / SiGAT motif lists on real signed graphs, nn/signed/SDGNN.py:198-254): a row with 10^5..10^6 entries
// serialises the launch.  As in the SpMM (spmm_long_kernel), rows with MORE than PYGSD_LONG_ROW entries are listed
// by the caller, skipped by the kernels above and handled here, segment-parallel:
//   stage 1  grid (segment, hub): per 4096-entry segment the softmax statistics (max, sum exp(. - max)) -- or, for
//            the backward kernels, the partial of  sum alpha * d alpha  -- reduced inside the block in a fixed order;
//   combine  one thread per hub folds the segment partials IN SEGMENT ORDER into the row statistic;
//   stage 2  grid (segment, hub): per-entry outputs from the row statistic + per-segment partial row sums;
//   finish   one thread per hub adds those partials in segment order and writes the per-row outputs.
// No atomics, run-to-run deterministic; non-hub rows are untouched (bitwise).  The kernels are written once over
// a small policy type per operation (what a logit is, what to emit per entry, what to store per row).
// ------------------------------------------------------------------------------------------
constexpr int kHubSeg = PYGSD_LONG_ROW;       // entries per block
constexpr int kHubThreads = 256;

struct HubCtx {
    const int32_t* rowptr;
    const int32_t* rows;
    int32_t n_seg;
};

__device__ __forceinline__ float block_max4(float v, float* sm)
{
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    __syncthreads();
    return v;
}
__device__ __forceinline__ float block_sum4(float v, float* sm)
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    v = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return v;
}

__device__ __forceinline__ bool hub_span(const HubCtx& c, int& row, int& lo, int& hi)
{
    row = c.rows[blockIdx.y];
    const int beg = c.rowptr[row], end = c.rowptr[row + 1];
    lo = beg + static_cast<int>(blockIdx.x) * kHubSeg;
    hi = lo + kHubSeg < end ? lo + kHubSeg : end;
    return lo < end;
}

// stage 1, softmax statistics: part = (max logit, sum exp(logit - max)) of the segment
template <class Op>
__global__ __launch_bounds__(kHubThreads) void hub_softmax_stats_kernel(Op op, HubCtx c, float2* __restrict__ part)
{
    __shared__ float sm[4];
    int row, lo, hi;
    const int slot = static_cast<int>(blockIdx.y) * c.n_seg + static_cast<int>(blockIdx.x);
    if (!hub_span(c, row, lo, hi)) {
        if (threadIdx.x == 0) part[slot] = make_float2(-INFINITY, 0.f);
        return;
    }
    const typename Op::Row ctx = op.row(row);
    float m = -INFINITY;
    for (int e = lo + threadIdx.x; e < hi; e += kHubThreads) m = fmaxf(m, op.logit(ctx, e));
    m = block_max4(m, sm);
    float sum = 0.f;
    for (int e = lo + threadIdx.x; e < hi; e += kHubThreads) sum += expf(op.logit(ctx, e) - m);
    sum = block_sum4(sum, sm);
    if (threadIdx.x == 0) part[slot] = make_float2(m, sum);
}

__global__ void hub_softmax_combine_kernel(int32_t n_long, int32_t n_seg, const float2* __restrict__ part,
                                           float2* __restrict__ stat)
{
    const int h = static_cast<int>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (h >= n_long) return;
    float m = -INFINITY;
    for (int k = 0; k < n_seg; ++k) m = fmaxf(m, part[h * n_seg + k].x);
    float sum = 0.f;
    for (int k = 0; k < n_seg; ++k) {
        const float2 p = part[h * n_seg + k];
        if (p.y > 0.f) sum += p.y * expf(p.x - m);
    }
    stat[h] = make_float2(m, sum + 1e-16f);
}

// stage 1, backward: part.x = sum over the segment of op.prod(e) (= alpha_e * d alpha_e)
template <class Op>
__global__ __launch_bounds__(kHubThreads) void hub_dot_kernel(Op op, HubCtx c, float2* __restrict__ part)
{
    __shared__ float sm[4];
    int row, lo, hi;
    const int slot = static_cast<int>(blockIdx.y) * c.n_seg + static_cast<int>(blockIdx.x);
    if (!hub_span(c, row, lo, hi)) {
        if (threadIdx.x == 0) part[slot] = make_float2(0.f, 0.f);
        return;
    }
    const typename Op::Row ctx = op.row(row);
    float acc = 0.f;
    for (int e = lo + threadIdx.x; e < hi; e += kHubThreads) acc += op.prod(ctx, e);
    acc = block_sum4(acc, sm);
    if (threadIdx.x == 0) part[slot] = make_float2(acc, 0.f);
}

// folds two-component segment partials in segment order: stat[h] = (sum .x, sum .y)
__global__ void hub_sum_combine_kernel(int32_t n_long, int32_t n_seg, const float2* __restrict__ part,
                                       float2* __restrict__ stat)
{
    const int h = static_cast<int>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (h >= n_long) return;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < n_seg; ++k) {
        a += part[h * n_seg + k].x;
        b += part[h * n_seg + k].y;
    }
    stat[h] = make_float2(a, b);
}

// stage 2: per-entry outputs from the row statistic; sums = the segment's partial row sums
template <class Op>
__global__ __launch_bounds__(kHubThreads) void hub_apply_kernel(Op op, HubCtx c, const float2* __restrict__ stat,
                                                                float2* __restrict__ sums)
{
    __shared__ float sm[4];
    int row, lo, hi;
    const int slot = static_cast<int>(blockIdx.y) * c.n_seg + static_cast<int>(blockIdx.x);
    if (!hub_span(c, row, lo, hi)) {
        if (threadIdx.x == 0) sums[slot] = make_float2(0.f, 0.f);
        return;
    }
    const typename Op::Row ctx = op.row(row);
    const float2 st = stat[blockIdx.y];
    float r0 = 0.f, r1 = 0.f;
    for (int e = lo + threadIdx.x; e < hi; e += kHubThreads) op.emit(ctx, e, st, r0, r1);
    r0 = block_sum4(r0, sm);
    r1 = block_sum4(r1, sm);
    if (threadIdx.x == 0) sums[slot] = make_float2(r0, r1);
}

template <class Op>
__global__ void hub_finish_kernel(Op op, const int32_t* __restrict__ rows, int32_t n_long, int32_t n_seg,
                                  const float2* __restrict__ sums)
{
    const int h = static_cast<int>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (h >= n_long) return;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < n_seg; ++k) {
        a += sums[h * n_seg + k].x;
        b += sums[h * n_seg + k].y;
    }
    op.finish(rows[h], a, b);
}

// ---- policies ------------------------------------------------------------------------------
struct GatFwdOp {     // gat_alpha_kernel
    const int32_t* col;
    const float *a_src, *a_dst;
    float slope;
    float* alpha;
    struct Row { float ad; };
    __device__ Row row(int r) const { return {a_dst[r]}; }
    __device__ float logit(const Row& c, int e) const
    {
        const float s = a_src[col[e]] + c.ad;
        return s > 0.f ? s : slope * s;
    }
    __device__ void emit(const Row& c, int e, float2 st, float&, float&) const { alpha[e] = expf(logit(c, e) - st.x) / st.y; }
    __device__ void finish(int, float, float) const {}
};

struct SegFwdOp {     // segment_softmax_kernel
    const float* logits;
    float* alpha;
    struct Row {};
    __device__ Row row(int) const { return {}; }
    __device__ float logit(const Row&, int e) const { return logits[e]; }
    __device__ void emit(const Row&, int e, float2 st, float&, float&) const { alpha[e] = expf(logits[e] - st.x) / st.y; }
    __device__ void finish(int, float, float) const {}
};

struct SegBwdOp {     // segment_softmax_bwd_kernel
    const float *alpha, *dalpha;
    float* dlogits;
    struct Row {};
    __device__ Row row(int) const { return {}; }
    __device__ float prod(const Row&, int e) const { return alpha[e] * dalpha[e]; }
    __device__ void emit(const Row&, int e, float2 st, float&, float&) const { dlogits[e] = alpha[e] * (dalpha[e] - st.x); }
    __device__ void finish(int, float, float) const {}
};

struct SneaFwdOp {    // snea_alpha_kernel
    const int32_t* col;
    const uint8_t* ptype;
    const float *s0, *s1, *d0, *d1, *bias_p;
    float *alpha, *share0, *share1;
    struct Row { float di0, di1, bias; };
    __device__ Row row(int r) const { return {d0[r], ptype ? d1[r] : 0.f, bias_p ? bias_p[0] : 0.f}; }
    __device__ float logit(const Row& c, int e) const
    {
        bool neg;
        return snea_logit(e, col, ptype, s0, s1, c.di0, c.di1, c.bias, neg);
    }
    __device__ void emit(const Row& c, int e, float2 st, float& r0, float& r1) const
    {
        bool neg;
        const float a = expf(snea_logit(e, col, ptype, s0, s1, c.di0, c.di1, c.bias, neg) - st.x) / st.y;
        alpha[e] = a;
        if (neg) r1 += a; else r0 += a;
    }
    __device__ void finish(int r, float a0, float a1) const
    {
        share0[r] = a0;
        if (share1) share1[r] = a1;
    }
};

struct SneaBwdOp {    // snea_alpha_bwd_kernel
    const int32_t* col;
    const uint8_t* ptype;
    const float *s0, *s1, *d0, *d1, *bias_p;
    const float *alpha, *dshare0, *dshare1;
    float *dpre0, *dpre1, *dd0, *dd1;
    struct Row { float di0, di1, g0, g1, bias; };
    __device__ Row row(int r) const
    {
        return {d0[r], ptype ? d1[r] : 0.f, dshare0[r], ptype ? dshare1[r] : 0.f, bias_p ? bias_p[0] : 0.f};
    }
    __device__ float prod(const Row& c, int e) const { return alpha[e] * ((ptype && ptype[e]) ? c.g1 : c.g0); }
    __device__ void emit(const Row& c, int e, float2 st, float& r0, float& r1) const
    {
        bool neg;
        const float t = snea_logit(e, col, ptype, s0, s1, c.di0, c.di1, c.bias, neg);
        const float dp = alpha[e] * ((neg ? c.g1 : c.g0) - st.x) * (1.f - t * t);
        dpre0[e] = neg ? 0.f : dp;
        if (dpre1) dpre1[e] = neg ? dp : 0.f;
        if (neg) r1 += dp; else r0 += dp;
    }
    __device__ void finish(int r, float a0, float a1) const
    {
        dd0[r] = a0;
        if (dd1) dd1[r] = a1;
    }
};

struct SegSumOp {     // segment_sum_kernel: stage 1 only, the row sum IS the output
    const int32_t* perm;
    const float* w;
    float* out;
    struct Row {};
    __device__ Row row(int) const { return {}; }
    __device__ float prod(const Row&, int e) const { return w[perm ? perm[e] : e]; }
    __device__ void finish(int r, float a, float) const { out[r] = a; }
};

// GAT backward on a hub row: 16-lane teams, one entry per team pass (the per-entry <g_i, h_j> is an F-long gather).
// csr_out != 0: ds in CSR order + partial row sums (v2);  else ds / alpha scattered to COO order through perm (v1).
struct GatBwdHubArgs {
    const int32_t *col, *perm;
    const float *a_src, *a_dst;
    float slope;
    const float* alpha;
    const float* h;
    int64_t ldh;
    const float* g;
    int64_t ldg;
    const float* out;
    int64_t ldo;
    int32_t n_feat;
    float *ds, *alpha_coo;
    int32_t csr_out;
};

__global__ __launch_bounds__(kHubThreads) void hub_gat_bwd_kernel(GatBwdHubArgs a, HubCtx c, float2* __restrict__ sums)
{
    __shared__ float sm[4];
    int row, lo, hi;
    const int slot = static_cast<int>(blockIdx.y) * c.n_seg + static_cast<int>(blockIdx.x);
    if (!hub_span(c, row, lo, hi)) {
        if (threadIdx.x == 0) sums[slot] = make_float2(0.f, 0.f);
        return;
    }
    const float* gi = a.g + static_cast<int64_t>(row) * a.ldg;
    const float* oi = a.out + static_cast<int64_t>(row) * a.ldo;
    float rd = 0.f;
    for (int f = threadIdx.x; f < a.n_feat; f += kHubThreads) rd = fmaf(gi[f], oi[f], rd);
    rd = block_sum4(rd, sm);                              // <g_i, out_i>
    const float ad = a.a_dst[row];
    const int t = threadIdx.x & 15, team = threadIdx.x >> 4;
    constexpr int kTeams = kHubThreads / 16, UN = 4;      // 4 entries per team in flight: the gathers are latency-bound
    float acc = 0.f;
    for (int e0 = lo; e0 < hi; e0 += kTeams * UN) {
        float dot[UN];
        int cj[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int e = e0 + u * kTeams + team;
            dot[u] = 0.f;
            cj[u] = e < hi ? a.col[e] : 0;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (e0 + u * kTeams + team < hi) {
                const float* hj = a.h + static_cast<int64_t>(cj[u]) * a.ldh;
                for (int f = t; f < a.n_feat; f += 16) dot[u] = fmaf(gi[f], hj[f], dot[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) dot[u] += __shfl_xor(dot[u], off);
            const int e = e0 + u * kTeams + team;
            if (e < hi && t == 0) {
                const float al = a.alpha[e];
                const float sc = a.a_src[cj[u]] + ad;
                const float d = al * (dot[u] - rd) * (sc > 0.f ? 1.f : a.slope);
                if (a.csr_out) {
                    a.ds[e] = d;
                    acc += d;
                } else {
                    const int p = a.perm[e];
                    a.ds[p] = d;
                    a.alpha_coo[p] = al;
                }
            }
        }
    }
    acc = block_sum4(acc, sm);
    if (threadIdx.x == 0) sums[slot] = make_float2(acc, 0.f);
}

struct StoreRowOp {   // finish of hub_gat_bwd_kernel (v2): da_dst[row] = sum of the segment partials
    float* out;
    __device__ void finish(int r, float a, float) const { out[r] = a; }
};

// ---- host side ------------------------------------------------------------------------------
struct HubPlan {
    HubCtx ctx;
    int32_t n_long;
    float2 *part, *stat, *sums;
    dim3 grid;
};

inline size_t hub_ws_bytes(int64_t n_long, int64_t n_seg) { return sizeof(float2) * static_cast<size_t>(2 * n_long * n_seg + n_long) + 256; }

inline int hub_plan(const char* who, const int32_t* rowptr, const pygsd_long_rows* h, HubPlan* p)
{
    PYGSD_REQUIRE(h->rows && h->workspace && h->n_rows > 0 && h->max_entries > PYGSD_LONG_ROW,
                  "%s: long-row descriptor needs rows, workspace, n_rows > 0 and max_entries > PYGSD_LONG_ROW", who);
    const int32_t n_seg = (h->max_entries + kHubSeg - 1) / kHubSeg;
    PYGSD_REQUIRE(h->workspace_bytes >= static_cast<int64_t>(hub_ws_bytes(h->n_rows, n_seg)),
                  "%s: long-row workspace too small (pygsd_segment_long_rows_workspace)", who);
    char* base = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(h->workspace) + 255) & ~static_cast<uintptr_t>(255));
    p->ctx = HubCtx{rowptr, h->rows, n_seg};
    p->n_long = h->n_rows;
    p->part = reinterpret_cast<float2*>(base);
    p->sums = p->part + static_cast<size_t>(h->n_rows) * n_seg;
    p->stat = p->sums + static_cast<size_t>(h->n_rows) * n_seg;
    p->grid = dim3(static_cast<unsigned>(n_seg), static_cast<unsigned>(h->n_rows));
    return 0;
}

inline dim3 per_hub_grid(int32_t n_long) { return dim3(static_cast<unsigned>((n_long + 63) / 64)); }

template <class Op>
int hub_softmax_forward(const char* who, const Op& op, const int32_t* rowptr, const pygsd_long_rows* h, hipStream_t s)
{
    HubPlan p;
    if (int rc = hub_plan(who, rowptr, h, &p)) return rc;
    hipLaunchKernelGGL(hub_softmax_stats_kernel<Op>, p.grid, dim3(kHubThreads), 0, s, op, p.ctx, p.part);
    hipLaunchKernelGGL(hub_softmax_combine_kernel, per_hub_grid(p.n_long), dim3(64), 0, s, p.n_long, p.ctx.n_seg, p.part, p.stat);
    hipLaunchKernelGGL(hub_apply_kernel<Op>, p.grid, dim3(kHubThreads), 0, s, op, p.ctx, p.stat, p.sums);
    hipLaunchKernelGGL(hub_finish_kernel<Op>, per_hub_grid(p.n_long), dim3(64), 0, s, op, h->rows, p.n_long, p.ctx.n_seg, p.sums);
    return check_launch(who);
}

template <class Op>
int hub_softmax_backward(const char* who, const Op& op, const int32_t* rowptr, const pygsd_long_rows* h, hipStream_t s)
{
    HubPlan p;
    if (int rc = hub_plan(who, rowptr, h, &p)) return rc;
    hipLaunchKernelGGL(hub_dot_kernel<Op>, p.grid, dim3(kHubThreads), 0, s, op, p.ctx, p.part);
    hipLaunchKernelGGL(hub_sum_combine_kernel, per_hub_grid(p.n_long), dim3(64), 0, s, p.n_long, p.ctx.n_seg, p.part, p.stat);
    hipLaunchKernelGGL(hub_apply_kernel<Op>, p.grid, dim3(kHubThreads), 0, s, op, p.ctx, p.stat, p.sums);
    hipLaunchKernelGGL(hub_finish_kernel<Op>, per_hub_grid(p.n_long), dim3(64), 0, s, op, h->rows, p.n_long, p.ctx.n_seg, p.sums);
    return check_launch(who);
}

}  // namespace
}  // namespace pygsd

using namespace pygsd;

namespace {
inline int32_t skip_of(const pygsd_long_rows* h) { return h ? PYGSD_LONG_ROW : 0; }
}  // namespace

extern "C" int pygsd_segment_long_rows_workspace(int32_t n_long, int32_t max_entries, int64_t* bytes)
{
    PYGSD_REQUIRE(bytes, "pygsd_segment_long_rows_workspace: null pointer");
    PYGSD_REQUIRE(n_long >= 0 && max_entries >= 0, "pygsd_segment_long_rows_workspace: negative size");
    const int64_t n_seg = (static_cast<int64_t>(max_entries) + kHubSeg - 1) / kHubSeg;
    *bytes = static_cast<int64_t>(hub_ws_bytes(n_long, n_seg));
    return 0;
}

extern "C" int pygsd_snea_alpha_csr_f32(const int32_t* rowptr, const int32_t* col, const uint8_t* edge_type,
                                        const float* s0, const float* s1, const float* d0, const float* d1,
                                        const float* bias, int32_t n_rows, float* alpha, float* share0,
                                        float* share1, const pygsd_long_rows* long_rows, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_snea_alpha_csr_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && s0 && d0 && share0, "pygsd_snea_alpha_csr_f32: null pointer");
    PYGSD_REQUIRE(!edge_type || (s1 && d1 && share1), "pygsd_snea_alpha_csr_f32: typed edges need s1, d1, share1");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    hipLaunchKernelGGL(snea_alpha_kernel, dim3((static_cast<unsigned>(n_rows) + 3) / 4), dim3(256), 0, s, rowptr, col,
                       edge_type, s0, s1, d0, d1, bias, n_rows, alpha, share0, share1, skip_of(long_rows));
    if (int rc = check_launch("snea_alpha_kernel")) return rc;
    if (!long_rows) return 0;
    return hub_softmax_forward("pygsd_snea_alpha_csr_f32 (hub rows)",
                               SneaFwdOp{col, edge_type, s0, s1, d0, d1, bias, alpha, share0, share1}, rowptr, long_rows, s);
}

extern "C" int pygsd_snea_alpha_bwd_csr_f32(const int32_t* rowptr, const int32_t* col, const uint8_t* edge_type,
                                            const float* s0, const float* s1, const float* d0, const float* d1,
                                            const float* bias, const float* alpha, const float* dshare0,
                                            const float* dshare1, int32_t n_rows, float* dpre0, float* dpre1,
                                            float* dd0, float* dd1, const pygsd_long_rows* long_rows, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_snea_alpha_bwd_csr_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && s0 && d0 && dshare0 && dd0, "pygsd_snea_alpha_bwd_csr_f32: null pointer");
    PYGSD_REQUIRE(!edge_type || (s1 && d1 && dshare1 && dpre1 && dd1),
                  "pygsd_snea_alpha_bwd_csr_f32: typed edges need the type-1 arrays");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    hipLaunchKernelGGL(snea_alpha_bwd_kernel, dim3((static_cast<unsigned>(n_rows) + 3) / 4), dim3(256), 0, s, rowptr,
                       col, edge_type, s0, s1, d0, d1, bias, alpha, dshare0, dshare1, n_rows, dpre0, dpre1, dd0, dd1,
                       skip_of(long_rows));
    if (int rc = check_launch("snea_alpha_bwd_kernel")) return rc;
    if (!long_rows) return 0;
    return hub_softmax_backward("pygsd_snea_alpha_bwd_csr_f32 (hub rows)",
                                SneaBwdOp{col, edge_type, s0, s1, d0, d1, bias, alpha, dshare0, dshare1, dpre0, dpre1, dd0, dd1},
                                rowptr, long_rows, s);
}

namespace {
int gat_bwd_hubs(const char* who, const GatBwdHubArgs& a, const int32_t* rowptr, const pygsd_long_rows* h, float* da_dst,
                 hipStream_t s)
{
    HubPlan p;
    if (int rc = hub_plan(who, rowptr, h, &p)) return rc;
    hipLaunchKernelGGL(hub_gat_bwd_kernel, p.grid, dim3(kHubThreads), 0, s, a, p.ctx, p.sums);
    if (da_dst)
        hipLaunchKernelGGL(hub_finish_kernel<StoreRowOp>, per_hub_grid(p.n_long), dim3(64), 0, s, StoreRowOp{da_dst}, h->rows,
                           p.n_long, p.ctx.n_seg, p.sums);
    return check_launch(who);
}
}  // namespace

extern "C" int pygsd_gat_alpha_bwd_csr_v2_f32(const int32_t* rowptr, const int32_t* col, const float* a_src,
                                              const float* a_dst, float negative_slope, const float* alpha,
                                              const float* h, int64_t ldh, const float* g, int64_t ldg,
                                              const float* out, int64_t ldo, int32_t n_rows, int32_t n_feat,
                                              float* ds_csr, float* da_dst, const pygsd_long_rows* long_rows, void* stream)
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
    const int32_t skip = skip_of(long_rows);
#define PYGSD_GAT_BWD(L)                                                                                             \
    hipLaunchKernelGGL(gat_alpha_bwd_vec_kernel<L>, grid, block, 0, s, rowptr, col, a_src, a_dst, negative_slope, alpha, \
                       h, ldh, g, ldg, out, ldo, n_rows, n_feat, ds_csr, da_dst, skip)
    const int quads = n_feat / 4;
    if (quads <= 4) PYGSD_GAT_BWD(4);
    else if (quads <= 8) PYGSD_GAT_BWD(8);
    else if (quads <= 16) PYGSD_GAT_BWD(16);
    else if (quads <= 32) PYGSD_GAT_BWD(32);
    else PYGSD_GAT_BWD(64);
#undef PYGSD_GAT_BWD
    if (int rc = check_launch("gat_alpha_bwd_vec_kernel")) return rc;
    if (!long_rows) return 0;
    return gat_bwd_hubs("pygsd_gat_alpha_bwd_csr_v2_f32 (hub rows)",
                        GatBwdHubArgs{col, nullptr, a_src, a_dst, negative_slope, alpha, h, ldh, g, ldg, out, ldo, n_feat,
                                      ds_csr, nullptr, 1},
                        rowptr, long_rows, da_dst, s);
}

extern "C" int pygsd_segment_sum_f32(const int32_t* rowptr, const int32_t* perm, const float* w, int32_t n_rows,
                                     float* out, const pygsd_long_rows* long_rows, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_segment_sum_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && out, "pygsd_segment_sum_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    const int64_t threads = static_cast<int64_t>(n_rows) * 16;
    hipLaunchKernelGGL(segment_sum_kernel, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256), 0, s, rowptr,
                       perm, w, n_rows, out, skip_of(long_rows));
    if (int rc = check_launch("segment_sum_kernel")) return rc;
    if (!long_rows) return 0;
    HubPlan p;
    if (int rc = hub_plan("pygsd_segment_sum_f32 (hub rows)", rowptr, long_rows, &p)) return rc;
    const SegSumOp op{perm, w, out};
    hipLaunchKernelGGL(hub_dot_kernel<SegSumOp>, p.grid, dim3(kHubThreads), 0, s, op, p.ctx, p.sums);
    hipLaunchKernelGGL(hub_finish_kernel<SegSumOp>, per_hub_grid(p.n_long), dim3(64), 0, s, op, long_rows->rows, p.n_long,
                       p.ctx.n_seg, p.sums);
    return check_launch("pygsd_segment_sum_f32 (hub rows)");
}

extern "C" int pygsd_segment_softmax_csr_f32(const int32_t* rowptr, const float* logits, int32_t n_rows, float* alpha,
                                             const pygsd_long_rows* long_rows, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_segment_softmax_csr_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr, "pygsd_segment_softmax_csr_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    hipLaunchKernelGGL(segment_softmax_kernel, dim3((static_cast<unsigned>(n_rows) + 3) / 4), dim3(256), 0, s, rowptr,
                       logits, n_rows, alpha, skip_of(long_rows));
    if (int rc = check_launch("segment_softmax_kernel")) return rc;
    if (!long_rows) return 0;
    return hub_softmax_forward("pygsd_segment_softmax_csr_f32 (hub rows)", SegFwdOp{logits, alpha}, rowptr, long_rows, s);
}

extern "C" int pygsd_segment_softmax_bwd_csr_f32(const int32_t* rowptr, const float* alpha, const float* dalpha,
                                                 int32_t n_rows, float* dlogits, const pygsd_long_rows* long_rows,
                                                 void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_segment_softmax_bwd_csr_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr, "pygsd_segment_softmax_bwd_csr_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    hipLaunchKernelGGL(segment_softmax_bwd_kernel, dim3((static_cast<unsigned>(n_rows) + 3) / 4), dim3(256), 0, s,
                       rowptr, alpha, dalpha, n_rows, dlogits, skip_of(long_rows));
    if (int rc = check_launch("segment_softmax_bwd_kernel")) return rc;
    if (!long_rows) return 0;
    return hub_softmax_backward("pygsd_segment_softmax_bwd_csr_f32 (hub rows)", SegBwdOp{alpha, dalpha, dlogits}, rowptr,
                                long_rows, s);
}

extern "C" int pygsd_gat_alpha_csr_f32(const int32_t* rowptr, const int32_t* col, const float* a_src,
                                       const float* a_dst, int32_t n_rows, float negative_slope, float* alpha,
                                       const pygsd_long_rows* long_rows, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_gat_alpha_csr_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && a_src && a_dst, "pygsd_gat_alpha_csr_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_SPMM, s);
    hipLaunchKernelGGL(gat_alpha_kernel, dim3((static_cast<unsigned>(n_rows) + 3) / 4), dim3(256), 0, s, rowptr,
                       col, a_src, a_dst, n_rows, negative_slope, alpha, skip_of(long_rows));
    if (int rc = check_launch("gat_alpha_kernel")) return rc;
    if (!long_rows) return 0;
    return hub_softmax_forward("pygsd_gat_alpha_csr_f32 (hub rows)", GatFwdOp{col, a_src, a_dst, negative_slope, alpha}, rowptr,
                               long_rows, s);
}

extern "C" int pygsd_gat_alpha_bwd_csr_f32(const int32_t* rowptr, const int32_t* col, const int32_t* perm,
                                           const float* a_src, const float* a_dst, float negative_slope,
                                           const float* alpha, const float* h, int64_t ldh, const float* g,
                                           int64_t ldg, const float* out, int64_t ldo, int32_t n_rows,
                                           int32_t n_feat, float* ds_coo, float* alpha_coo,
                                           const pygsd_long_rows* long_rows, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0 && n_feat >= 0, "pygsd_gat_alpha_bwd_csr_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && a_src && a_dst && h && g && out, "pygsd_gat_alpha_bwd_csr_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_SDDMM, s);
    hipLaunchKernelGGL(gat_alpha_bwd_kernel, dim3((static_cast<unsigned>(n_rows) + 3) / 4), dim3(256), 0, s, rowptr,
                       col, perm, a_src, a_dst, negative_slope, alpha, h, ldh, g, ldg, out, ldo, n_rows, n_feat,
                       ds_coo, alpha_coo, skip_of(long_rows));
    if (int rc = check_launch("gat_alpha_bwd_kernel")) return rc;
    if (!long_rows) return 0;
    return gat_bwd_hubs("pygsd_gat_alpha_bwd_csr_f32 (hub rows)",
                        GatBwdHubArgs{col, perm, a_src, a_dst, negative_slope, alpha, h, ldh, g, ldg, out, ldo, n_feat,
                                      ds_coo, alpha_coo, 0},
                        rowptr, long_rows, nullptr, s);
}
