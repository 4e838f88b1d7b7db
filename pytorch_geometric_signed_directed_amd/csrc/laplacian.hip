// Operator build on the device: the (signed) magnetic Laplacian
// of utils/directed/get_magnetic_Laplacian.py:47-85 / utils/general/get_magnetic_signed_Laplacian.py:47-90,
// and the self-loop / degree normalisations of gcn_norm (nn/directed/DGCNConv.py:75) and conv_norm_rw
// (nn/general/conv_base.py:12-31).  Integer / byte work, HBM-bound: symmetrise into 64-bit (row, col)
// keys, stable radix sort (rocPRIM), mark run heads, scan, merge runs IN SORTED ORDER (the order
// coalesce's scatter-add sums duplicates in), per-row degree as a sequential row sum (the order
// scatter_add_ uses on the reference's CPU path), then the element-wise phase / normalisation.
// Deterministic: no atomics anywhere except one last-writer-wins index max for duplicate self loops.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace pygsd {
namespace {

constexpr int kBlock = 256;

inline unsigned grid_for(int64_t n)
{
    int64_t g = (n + kBlock - 1) / kBlock;
    if (g > 256 * 32) g = 256 * 32;
    return static_cast<unsigned>(g < 1 ? 1 : g);
}

inline int bits_for(uint64_t v)
{
    int b = 1;
    while (b < 64 && (v >> b) != 0) ++b;
    return b;
}

#define GRID_STRIDE(i, n)                                                                  \
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < (n); \
         i += static_cast<int64_t>(gridDim.x) * blockDim.x)

// entries p in [0, 2E): p < E -> (row, col) of edge p; p >= E -> (col, row) of edge p - E.
// self loops get the sentinel key (sorts last, never a head).
__global__ void sym_keys(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t e,
                         uint64_t n, uint64_t sentinel, uint64_t* __restrict__ keys, uint32_t* __restrict__ ids)
{
    GRID_STRIDE(p, 2 * e)
    {
        const int64_t k = p < e ? p : p - e;
        const uint64_t r = static_cast<uint64_t>(row[k]), c = static_cast<uint64_t>(col[k]);
        uint64_t key = sentinel;
        if (r != c) key = p < e ? r * n + c : c * n + r;
        keys[p] = key;
        ids[p] = static_cast<uint32_t>(p);
    }
}

__global__ void mark_heads(const uint64_t* __restrict__ keys, int64_t m, uint64_t sentinel,
                           uint32_t* __restrict__ flags)
{
    GRID_STRIDE(i, m)
    {
        const uint64_t k = keys[i];
        flags[i] = (k != sentinel && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
    }
}

__global__ void count_unique(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ excl, int64_t m,
                             int64_t* __restrict__ out)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *out = m > 0 ? static_cast<int64_t>(excl[m - 1]) + flags[m - 1] : 0;
}

// one thread per run head: walk the run in sorted order, sum w / +-w / |w|
__global__ void merge_runs(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ perm,
                           const uint32_t* __restrict__ flags, const uint32_t* __restrict__ seg, int64_t m,
                           int64_t e, uint64_t n, const float* __restrict__ w, int64_t* __restrict__ out_row,
                           int64_t* __restrict__ out_col, float* __restrict__ a_sym, float* __restrict__ theta,
                           float* __restrict__ a_abs)
{
    GRID_STRIDE(i, m)
    {
        if (!flags[i]) continue;
        const uint64_t key = keys[i];
        float s = 0.f, t = 0.f, a = 0.f;
        for (int64_t j = i; j < m && keys[j] == key; ++j) {
            const uint32_t p = perm[j];
            const bool first = static_cast<int64_t>(p) < e;
            const float we = w ? w[first ? p : p - e] : 1.f;
            s = s + we;
            t = t + (first ? we : -we);
            a = a + fabsf(we);
        }
        const uint32_t o = seg[i];
        out_row[o] = static_cast<int64_t>(key / n);
        out_col[o] = static_cast<int64_t>(key % n);
        a_sym[o] = s / 2.f;
        theta[o] = t;
        if (a_abs) a_abs[o] = a / 2.f;
    }
}

// off_ptr[r] = first sorted entry whose row is >= r (r = 0 .. n): entry i starts every row in
// (row[i-1], row[i]]; the virtual entry i = es closes the remaining rows.  Coalesced, no search.
__global__ void row_starts(const int64_t* __restrict__ out_row, int64_t es, int32_t n,
                           int32_t* __restrict__ off_ptr)
{
    GRID_STRIDE(i, es + 1)
    {
        const int64_t prev = i == 0 ? -1 : out_row[i - 1];
        const int64_t cur = i == es ? n : out_row[i];
        for (int64_t r = prev + 1; r <= cur; ++r) off_ptr[r] = static_cast<int32_t>(i);
    }
}

// Rows longer than this are NOT summed by one thread (the sequential sums below reproduce the reference's summation
// order, which costs 0.1 us per entry of a single row: a 10^6-entry hub row made gcn_norm 129 ms instead of 0.7): the
// sequential kernels skip them and long_row_sums adds them block-cooperatively -- per-thread strided partials, then a
// fixed tree: deterministic, equal to the sequential sum to fp32 rounding.  Same threshold as PYGSD_LONG_ROW.
constexpr int32_t kLongRow = PYGSD_LONG_ROW;

// deg[r] = sum over the (row-sorted) entries of row r, sequential in sorted order
__global__ void row_degree(const int32_t* __restrict__ off_ptr, const float* __restrict__ src, int32_t n,
                           int32_t use_abs, float* __restrict__ deg)
{
    GRID_STRIDE(r, n)
    {
        if (off_ptr[r + 1] - off_ptr[r] > kLongRow) continue;        // long_row_sums
        float d = 0.f;
        for (int32_t j = off_ptr[r]; j < off_ptr[r + 1]; ++j) d = d + (use_abs ? fabsf(src[j]) : src[j]);
        deg[r] = d;
    }
}

// Every block scans its slice of the row pointer for long rows and sums each one it finds with all its threads.
__global__ __launch_bounds__(256) void long_row_sums(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm,
                                                     const float* __restrict__ w, int32_t n, int32_t use_abs,
                                                     float* __restrict__ out)
{
    __shared__ float sm[4];
    const int32_t per = (n + static_cast<int32_t>(gridDim.x) - 1) / static_cast<int32_t>(gridDim.x);
    const int32_t r0 = static_cast<int32_t>(blockIdx.x) * per;
    const int32_t r1 = r0 + per < n ? r0 + per : n;
    for (int32_t r = r0; r < r1; ++r) {
        const int32_t beg = rowptr[r], end = rowptr[r + 1];           // block-uniform
        if (end - beg <= kLongRow) continue;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int32_t j = beg + static_cast<int32_t>(threadIdx.x);
        for (; j + 3 * 256 < end; j += 4 * 256) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float v = w[perm ? perm[j + u * 256] : j + u * 256];
                acc[u] += use_abs ? fabsf(v) : v;
            }
        }
        for (; j < end; j += 256) {
            const float v = w[perm ? perm[j] : j];
            acc[0] += use_abs ? fabsf(v) : v;
        }
        float d = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) out[r] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
        __syncthreads();
    }
}

// Compute layout of the magnetic operator: ONE int32 CSR over the symmetric pattern (off-diagonals + the
// diagonal, columns ascending inside a row) shared by both orientations, with the values of
// S[row, col] (by-source / backward product) and of the mirrored entry S[col, row] (by-target / forward
// product) side by side.  The mirror values are evaluated by lap_values (the operator is Hermitian:
// same |.|, conjugate phase, multiplied in the mirrored entry's own order), so there is no lookup.
// Scaling S = 2 L / lambda_max - I is folded in: v = (2 x) / lam with +inf -> 0 (masked_fill_), diag - 1.
// Replaces two radix sorts + six gathers of the generic COO -> CSR route.
__device__ __forceinline__ float scale_lam(float x, float lam)
{
    const float v = (2.0f * x) / lam;
    return v == INFINITY ? 0.f : v;
}

__global__ void assemble_csr(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                             const float* __restrict__ off_re, const float* __restrict__ off_im,
                             const float* __restrict__ mir_re, const float* __restrict__ mir_im,
                             const float* __restrict__ diag, const int32_t* __restrict__ off_ptr, int64_t es,
                             int32_t n, float lam, float diag_shift, int32_t* __restrict__ rowptr,
                             int32_t* __restrict__ ccol, float* __restrict__ vb_re, float* __restrict__ vb_im,
                             float* __restrict__ vf_re, float* __restrict__ vf_im)
{
    GRID_STRIDE(t, es + n)
    {
        if (t < es) {
            const int32_t r = static_cast<int32_t>(row[t]), c = static_cast<int32_t>(col[t]);
            const int64_t slot = t + r + (c > r ? 1 : 0);
            ccol[slot] = c;
            vb_re[slot] = scale_lam(off_re[t], lam);
            vb_im[slot] = scale_lam(off_im[t], lam);
            vf_re[slot] = scale_lam(mir_re[t], lam);
            vf_im[slot] = scale_lam(mir_im[t], lam);
        } else {
            const int32_t r = static_cast<int32_t>(t - es);
            int32_t lo = off_ptr[r], hi = off_ptr[r + 1];
            const int32_t beg = lo;
            while (lo < hi) {                                   // #off-diagonals of row r left of the diagonal
                const int32_t mid = (lo + hi) >> 1;
                if (col[mid] < r) lo = mid + 1; else hi = mid;
            }
            const int64_t slot = static_cast<int64_t>(beg) + r + (lo - beg);
            const float d = scale_lam(diag[r], lam) + diag_shift;
            ccol[slot] = r;
            vb_re[slot] = d;
            vf_re[slot] = d;
            vb_im[slot] = 0.f;
            vf_im[slot] = 0.f;
            rowptr[r] = beg + r;
            if (r == n - 1) rowptr[n] = static_cast<int32_t>(es + n);
        }
    }
}

// off-diagonal values of L and its diagonal.  sym: -(D^-1/2 A_s D^-1/2 (.) e^{i phase}), diag 1;
// else: -(A_s (.) e^{i phase}), diag deg.  phase = fp32(2 pi q) * theta, evaluated in fp32 like the
// reference's complex64 exp.
__global__ void lap_values(const int64_t* __restrict__ out_row, const int64_t* __restrict__ out_col,
                           const float* __restrict__ a_sym, const float* __restrict__ theta,
                           const float* __restrict__ deg, int64_t es, float two_pi_q, int32_t sym,
                           float* __restrict__ off_re, float* __restrict__ off_im,
                           float* __restrict__ mir_re, float* __restrict__ mir_im)
{
    GRID_STRIDE(i, es)
    {
        const float ph = two_pi_q * theta[i];
        float sn, cs;
        sincosf(ph, &sn, &cs);
        float mag = a_sym[i], mmag = a_sym[i];
        if (sym) {
            const float dr = deg[out_row[i]], dc = deg[out_col[i]];
            const float ir = dr == 0.f ? 0.f : powf(dr, -0.5f);
            const float ic = dc == 0.f ? 0.f : powf(dc, -0.5f);
            mag = ir * mag * ic;      // entry (row, col): deg^-1/2[row] * A_s * deg^-1/2[col]
            mmag = ic * mmag * ir;    // mirrored entry (col, row), multiplied in ITS row/col order
        }
        off_re[i] = -(mag * cs);
        off_im[i] = -(mag * sn);
        if (mir_re) {                 // Hermitian mirror: A_s symmetric, Theta antisymmetric
            mir_re[i] = -(mmag * cs);
            mir_im[i] = mmag * sn;    // -(mmag * sin(-ph))
        }
    }
}

// ---- add_remaining_self_loops + degree normalisations (gcn_norm / conv_norm_rw) -------------------
__global__ void loop_flags(const int64_t* __restrict__ row, const int64_t* __restrict__ col, int64_t e,
                           uint32_t* __restrict__ keep, int32_t* __restrict__ last_loop)
{
    GRID_STRIDE(i, e)
    {
        const bool off = row[i] != col[i];
        keep[i] = off ? 1u : 0u;
        if (!off) atomicMax(&last_loop[row[i]], static_cast<int32_t>(i));  // the LAST listed loop wins
    }
}

__global__ void fill_i32(int32_t* __restrict__ p, int64_t n, int32_t v)
{
    GRID_STRIDE(i, n) p[i] = v;
}

__global__ void compact_edges(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                              const float* __restrict__ w, const uint32_t* __restrict__ keep,
                              const uint32_t* __restrict__ pos, int64_t e, int64_t* __restrict__ o_row,
                              int64_t* __restrict__ o_col, float* __restrict__ o_w)
{
    GRID_STRIDE(i, e)
    {
        if (!keep[i]) continue;
        const uint32_t o = pos[i];
        o_row[o] = row[i];
        o_col[o] = col[i];
        if (o_w) o_w[o] = w ? w[i] : 1.f;
    }
}

__global__ void append_loops(const float* __restrict__ w, const int32_t* __restrict__ last_loop, int32_t n,
                             float fill, int64_t base, int64_t* __restrict__ o_row, int64_t* __restrict__ o_col,
                             float* __restrict__ o_w)
{
    GRID_STRIDE(v, n)
    {
        o_row[base + v] = v;
        o_col[base + v] = v;
        if (o_w) {
            const int32_t l = last_loop[v];
            o_w[base + v] = (l >= 0 && w) ? w[l] : (l >= 0 ? 1.f : fill);
        }
    }
}

// deg[r] = sum_{slots of CSR row r} w[perm[slot]]   (sequential: COO order inside the row; perm == NULL: w[slot])
__global__ void csr_row_sum(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm,
                            const float* __restrict__ w, int32_t n, float* __restrict__ deg)
{
    GRID_STRIDE(r, n)
    {
        if (rowptr[r + 1] - rowptr[r] > kLongRow) continue;          // long_row_sums
        float d = 0.f;
        for (int32_t j = rowptr[r]; j < rowptr[r + 1]; ++j) d = d + w[perm ? perm[j] : j];
        deg[r] = d;
    }
}

// mode 0 (gcn_norm): out = deg^-1/2[row] * w * deg^-1/2[col];  mode 1 (conv_norm_rw): out = deg^-1[row] * w
__global__ void degree_scale(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                             const float* __restrict__ w, const float* __restrict__ deg, int64_t e,
                             int32_t mode, float* __restrict__ out)
{
    GRID_STRIDE(i, e)
    {
        const float dr = deg[row[i]];
        if (mode == 0) {
            const float dc = deg[col[i]];
            float ir = powf(dr, -0.5f), ic = powf(dc, -0.5f);
            if (isinf(ir)) ir = 0.f;   // masked_fill_(== inf, 0)
            if (isinf(ic)) ic = 0.f;
            out[i] = ir * w[i] * ic;
        } else {
            float ir = powf(dr, -1.0f);
            if (isinf(ir)) ir = 0.f;
            out[i] = ir * w[i];
        }
    }
}

struct LapWs {
    size_t keys_in, keys_out, ids, perm, flags, seg, sort_tmp, scan_tmp, sort_tmp_bytes, scan_tmp_bytes, total;
};

int lap_layout(int64_t e, LapWs* w)
{
    const size_t m = static_cast<size_t>(2 * (e > 0 ? e : 1));
    size_t sort_tmp = 0, scan_tmp = 0;
    uint64_t* k = nullptr;
    uint32_t* v = nullptr;
    PYGSD_HIP_TRY(rocprim::radix_sort_pairs(nullptr, sort_tmp, k, k, v, v, m, 0u, 64u, hipStream_t(nullptr)));
    PYGSD_HIP_TRY(rocprim::exclusive_scan(nullptr, scan_tmp, v, v, 0u, m, rocprim::plus<uint32_t>(),
                                          hipStream_t(nullptr)));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += round_up(bytes, 256); return o; };
    w->keys_in = take(m * 8);
    w->keys_out = take(m * 8);
    w->ids = take(m * 4);
    w->perm = take(m * 4);
    w->flags = take(m * 4);
    w->seg = take(m * 4);
    w->sort_tmp = take(sort_tmp);
    w->scan_tmp = take(scan_tmp);
    w->sort_tmp_bytes = sort_tmp;
    w->scan_tmp_bytes = scan_tmp;
    w->total = off + 256;
    return 0;
}

inline char* align256(void* p)
{
    return reinterpret_cast<char*>(round_up(reinterpret_cast<uintptr_t>(p), 256));
}

struct LoopWs {
    size_t keep, pos, scan_tmp, scan_tmp_bytes, total;
};

int loop_layout(int64_t e, LoopWs* w)
{
    const size_t m = static_cast<size_t>(e > 0 ? e : 1);
    size_t scan_tmp = 0;
    uint32_t* v = nullptr;
    PYGSD_HIP_TRY(rocprim::exclusive_scan(nullptr, scan_tmp, v, v, 0u, m, rocprim::plus<uint32_t>(),
                                          hipStream_t(nullptr)));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += round_up(bytes, 256); return o; };
    w->keep = take(m * 4);
    w->pos = take(m * 4);
    w->scan_tmp = take(scan_tmp);
    w->scan_tmp_bytes = scan_tmp;
    w->total = off + 256;
    return 0;
}

}  // namespace
}  // namespace pygsd

using namespace pygsd;

extern "C" int pygsd_maglap_workspace(int64_t n_edges, size_t* bytes)
{
    PYGSD_REQUIRE(bytes, "pygsd_maglap_workspace: null output");
    PYGSD_REQUIRE(n_edges >= 0 && 2 * n_edges < (int64_t(1) << 31), "pygsd_maglap_workspace: 2*n_edges out of int32 range");
    LapWs w;
    if (int rc = lap_layout(n_edges, &w)) return rc;
    *bytes = w.total;
    return 0;
}

extern "C" int pygsd_maglap_sort(const int64_t* row, const int64_t* col, int64_t n_edges, int32_t n,
                                 void* workspace, size_t workspace_bytes, int64_t* d_num_unique, void* stream)
{
    PYGSD_REQUIRE(n_edges >= 0 && 2 * n_edges < (int64_t(1) << 31) && n >= 0, "pygsd_maglap_sort: size out of range");
    PYGSD_REQUIRE(d_num_unique && workspace, "pygsd_maglap_sort: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    if (n_edges == 0) {
        PYGSD_HIP_TRY(hipMemsetAsync(d_num_unique, 0, sizeof(int64_t), s));
        return 0;
    }
    PYGSD_REQUIRE(row && col, "pygsd_maglap_sort: null pointer");
    LapWs w;
    if (int rc = lap_layout(n_edges, &w)) return rc;
    PYGSD_REQUIRE(workspace_bytes >= w.total, "pygsd_maglap_sort: workspace too small (%zu < %zu)", workspace_bytes, w.total);
    char* base = align256(workspace);
    uint64_t* keys_in = reinterpret_cast<uint64_t*>(base + w.keys_in);
    uint64_t* keys_out = reinterpret_cast<uint64_t*>(base + w.keys_out);
    uint32_t* ids = reinterpret_cast<uint32_t*>(base + w.ids);
    uint32_t* perm = reinterpret_cast<uint32_t*>(base + w.perm);
    uint32_t* flags = reinterpret_cast<uint32_t*>(base + w.flags);
    uint32_t* seg = reinterpret_cast<uint32_t*>(base + w.seg);
    const int64_t m = 2 * n_edges;
    const uint64_t nn = static_cast<uint64_t>(n);
    const int kb = bits_for(nn * nn);           // valid keys < n*n <= 2^kb; sentinel = 2^kb needs kb+1 bits
    PYGSD_REQUIRE(kb < 63, "pygsd_maglap_sort: n too large for 64-bit (row, col) keys");
    const uint64_t sentinel = uint64_t(1) << kb;
    hipLaunchKernelGGL(sym_keys, dim3(grid_for(m)), dim3(kBlock), 0, s, row, col, n_edges, nn, sentinel, keys_in, ids);
    if (int rc = check_launch("sym_keys")) return rc;
    size_t tb = w.sort_tmp_bytes;
    PYGSD_HIP_TRY(rocprim::radix_sort_pairs(base + w.sort_tmp, tb, keys_in, keys_out, ids, perm,
                                            static_cast<size_t>(m), 0u, static_cast<unsigned>(kb + 1), s));
    hipLaunchKernelGGL(mark_heads, dim3(grid_for(m)), dim3(kBlock), 0, s, keys_out, m, sentinel, flags);
    if (int rc = check_launch("mark_heads")) return rc;
    tb = w.scan_tmp_bytes;
    PYGSD_HIP_TRY(rocprim::exclusive_scan(base + w.scan_tmp, tb, flags, seg, 0u, static_cast<size_t>(m),
                                          rocprim::plus<uint32_t>(), s));
    hipLaunchKernelGGL(count_unique, dim3(1), dim3(64), 0, s, flags, seg, m, d_num_unique);
    return check_launch("count_unique");
}

extern "C" int pygsd_maglap_merge(const float* w, int64_t n_edges, int32_t n, int32_t is_signed,
                                  int32_t absolute_degree, int64_t num_unique, void* workspace,
                                  size_t workspace_bytes, int64_t* out_row, int64_t* out_col, float* a_sym,
                                  float* theta, float* deg, int32_t* off_ptr, void* stream)
{
    PYGSD_REQUIRE(n_edges >= 0 && n >= 0 && num_unique >= 0, "pygsd_maglap_merge: negative size");
    PYGSD_REQUIRE(off_ptr, "pygsd_maglap_merge: null off_ptr");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    if (n > 0) {
        PYGSD_REQUIRE(deg, "pygsd_maglap_merge: null deg");
        PYGSD_HIP_TRY(hipMemsetAsync(deg, 0, sizeof(float) * static_cast<size_t>(n), s));
    }
    if (n_edges == 0 || num_unique == 0) {
        PYGSD_HIP_TRY(hipMemsetAsync(off_ptr, 0, sizeof(int32_t) * (static_cast<size_t>(n) + 1), s));
        return 0;
    }
    PYGSD_REQUIRE(workspace && out_row && out_col && a_sym && theta, "pygsd_maglap_merge: null pointer");
    LapWs l;
    if (int rc = lap_layout(n_edges, &l)) return rc;
    PYGSD_REQUIRE(workspace_bytes >= l.total, "pygsd_maglap_merge: workspace too small");
    char* base = align256(workspace);
    const int64_t m = 2 * n_edges;
    // |w| sums are only needed for the signed absolute-degree variant; reuse keys_in (dead after the sort)
    float* a_abs = (is_signed && absolute_degree) ? reinterpret_cast<float*>(base + l.keys_in) : nullptr;
    hipLaunchKernelGGL(merge_runs, dim3(grid_for(m)), dim3(kBlock), 0, s,
                       reinterpret_cast<uint64_t*>(base + l.keys_out), reinterpret_cast<uint32_t*>(base + l.perm),
                       reinterpret_cast<uint32_t*>(base + l.flags), reinterpret_cast<uint32_t*>(base + l.seg), m,
                       n_edges, static_cast<uint64_t>(n), w, out_row, out_col, a_sym, theta, a_abs);
    if (int rc = check_launch("merge_runs")) return rc;
    hipLaunchKernelGGL(row_starts, dim3(grid_for(num_unique + 1)), dim3(kBlock), 0, s, out_row, num_unique, n, off_ptr);
    if (int rc = check_launch("row_starts")) return rc;
    const float* src = a_abs ? a_abs : a_sym;
    const int use_abs = (is_signed && !absolute_degree) ? 1 : 0;
    hipLaunchKernelGGL(row_degree, dim3(grid_for(n)), dim3(kBlock), 0, s, off_ptr, src, n, use_abs, deg);
    hipLaunchKernelGGL(long_row_sums, dim3(n < 2048 ? 1 : 2048), dim3(256), 0, s, off_ptr, static_cast<const int32_t*>(nullptr),
                       src, n, use_abs, deg);
    return check_launch("row_degree");
}

extern "C" int pygsd_maglap_assemble_csr(const int64_t* out_row, const int64_t* out_col, const float* off_real,
                                         const float* off_imag, const float* mir_real, const float* mir_imag,
                                         const float* diag, const int32_t* off_ptr, int64_t num_unique,
                                         int32_t n, float lambda_max, float diag_shift, int32_t* rowptr,
                                         int32_t* col, float* vb_real, float* vb_imag, float* vf_real,
                                         float* vf_imag, void* stream)
{
    PYGSD_REQUIRE(num_unique >= 0 && n >= 0 && num_unique + n < (int64_t(1) << 31),
                  "pygsd_maglap_assemble_csr: size out of int32 range");
    PYGSD_REQUIRE(rowptr, "pygsd_maglap_assemble_csr: null rowptr");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    if (n == 0) {
        PYGSD_HIP_TRY(hipMemsetAsync(rowptr, 0, sizeof(int32_t), s));
        return 0;
    }
    PYGSD_REQUIRE(diag && off_ptr && col && vb_real && vb_imag && vf_real && vf_imag &&
                      (num_unique == 0 || (out_row && out_col && off_real && off_imag && mir_real && mir_imag)),
                  "pygsd_maglap_assemble_csr: null pointer");
    hipLaunchKernelGGL(assemble_csr, dim3(grid_for(num_unique + n)), dim3(kBlock), 0, s, out_row, out_col, off_real,
                       off_imag, mir_real, mir_imag, diag, off_ptr, num_unique, n, lambda_max, diag_shift, rowptr, col,
                       vb_real, vb_imag, vf_real, vf_imag);
    return check_launch("assemble_csr");
}

extern "C" int pygsd_maglap_values(const int64_t* out_row, const int64_t* out_col, const float* a_sym,
                                   const float* theta, const float* deg, int64_t num_unique, double q,
                                   int32_t sym, float* off_real, float* off_imag, float* mir_real,
                                   float* mir_imag, void* stream)
{
    PYGSD_REQUIRE(num_unique >= 0, "pygsd_maglap_values: negative size");
    if (num_unique == 0) return 0;
    PYGSD_REQUIRE(out_row && out_col && a_sym && theta && deg && off_real && off_imag, "pygsd_maglap_values: null pointer");
    PYGSD_REQUIRE((mir_real == nullptr) == (mir_imag == nullptr), "pygsd_maglap_values: mirror outputs must be given together");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    // torch evaluates 1j*2*pi*q as a double-precision Python complex, then casts it to complex64: ONE rounding, of the product
    // -- q arrives as a double for that reason (until ABI v17 it was a float, i.e. rounded before the multiplication: one ulp
    // of 2 pi q, 2e-3 rad on a phase argument of 15 456 -- a pair joined by that many parallel edges, tests/test_gpu_fuzz.py)
    const float two_pi_q = static_cast<float>(2.0 * 3.14159265358979323846 * q);
    hipLaunchKernelGGL(lap_values, dim3(grid_for(num_unique)), dim3(kBlock), 0, s, out_row, out_col, a_sym, theta,
                       deg, num_unique, two_pi_q, sym, off_real, off_imag, mir_real, mir_imag);
    return check_launch("lap_values");
}

extern "C" int pygsd_self_loops_workspace(int64_t n_edges, size_t* bytes)
{
    PYGSD_REQUIRE(bytes, "pygsd_self_loops_workspace: null output");
    PYGSD_REQUIRE(n_edges >= 0 && n_edges < (int64_t(1) << 31), "pygsd_self_loops_workspace: n_edges out of range");
    LoopWs w;
    if (int rc = loop_layout(n_edges, &w)) return rc;
    *bytes = w.total;
    return 0;
}

// add_remaining_self_loops: stage 1 marks/scans (writes the number of non-loop edges to *d_num_kept and
// the last listed loop of every node to last_loop[n]); stage 2 compacts and appends the n loops.
extern "C" int pygsd_self_loops_scan(const int64_t* row, const int64_t* col, int64_t n_edges, int32_t n,
                                     void* workspace, size_t workspace_bytes, int32_t* last_loop,
                                     int64_t* d_num_kept, void* stream)
{
    PYGSD_REQUIRE(n_edges >= 0 && n_edges < (int64_t(1) << 31) && n >= 0, "pygsd_self_loops_scan: size out of range");
    PYGSD_REQUIRE(d_num_kept && workspace && (n == 0 || last_loop), "pygsd_self_loops_scan: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    if (n > 0) {
        hipLaunchKernelGGL(fill_i32, dim3(grid_for(n)), dim3(kBlock), 0, s, last_loop, n, -1);
        if (int rc = check_launch("fill_i32")) return rc;
    }
    if (n_edges == 0) {
        PYGSD_HIP_TRY(hipMemsetAsync(d_num_kept, 0, sizeof(int64_t), s));
        return 0;
    }
    PYGSD_REQUIRE(row && col, "pygsd_self_loops_scan: null pointer");
    LoopWs w;
    if (int rc = loop_layout(n_edges, &w)) return rc;
    PYGSD_REQUIRE(workspace_bytes >= w.total, "pygsd_self_loops_scan: workspace too small");
    char* base = align256(workspace);
    uint32_t* keep = reinterpret_cast<uint32_t*>(base + w.keep);
    uint32_t* pos = reinterpret_cast<uint32_t*>(base + w.pos);
    hipLaunchKernelGGL(loop_flags, dim3(grid_for(n_edges)), dim3(kBlock), 0, s, row, col, n_edges, keep, last_loop);
    if (int rc = check_launch("loop_flags")) return rc;
    size_t tb = w.scan_tmp_bytes;
    PYGSD_HIP_TRY(rocprim::exclusive_scan(base + w.scan_tmp, tb, keep, pos, 0u, static_cast<size_t>(n_edges),
                                          rocprim::plus<uint32_t>(), s));
    hipLaunchKernelGGL(count_unique, dim3(1), dim3(64), 0, s, keep, pos, n_edges, d_num_kept);
    return check_launch("count_unique");
}

extern "C" int pygsd_self_loops_emit(const int64_t* row, const int64_t* col, const float* w, int64_t n_edges,
                                     int32_t n, float fill_value, int64_t num_kept, void* workspace,
                                     size_t workspace_bytes, const int32_t* last_loop, int64_t* out_row,
                                     int64_t* out_col, float* out_w, void* stream)
{
    PYGSD_REQUIRE(n_edges >= 0 && n >= 0 && num_kept >= 0 && num_kept <= n_edges, "pygsd_self_loops_emit: bad sizes");
    if (num_kept + n == 0) return 0;
    PYGSD_REQUIRE(out_row && out_col && workspace, "pygsd_self_loops_emit: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    if (n_edges > 0) {
        LoopWs l;
        if (int rc = loop_layout(n_edges, &l)) return rc;
        PYGSD_REQUIRE(workspace_bytes >= l.total, "pygsd_self_loops_emit: workspace too small");
        char* base = align256(workspace);
        hipLaunchKernelGGL(compact_edges, dim3(grid_for(n_edges)), dim3(kBlock), 0, s, row, col, w,
                           reinterpret_cast<uint32_t*>(base + l.keep), reinterpret_cast<uint32_t*>(base + l.pos),
                           n_edges, out_row, out_col, out_w);
        if (int rc = check_launch("compact_edges")) return rc;
    }
    if (n > 0) {
        PYGSD_REQUIRE(last_loop, "pygsd_self_loops_emit: null last_loop");
        hipLaunchKernelGGL(append_loops, dim3(grid_for(n)), dim3(kBlock), 0, s, w, last_loop, n, fill_value, num_kept,
                           out_row, out_col, out_w);
        if (int rc = check_launch("append_loops")) return rc;
    }
    return 0;
}

extern "C" int pygsd_csr_row_sum_f32(const int32_t* rowptr, const int32_t* perm, const float* w, int32_t n_rows,
                                     float* out, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_csr_row_sum_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && out, "pygsd_csr_row_sum_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    hipLaunchKernelGGL(csr_row_sum, dim3(grid_for(n_rows)), dim3(kBlock), 0, s, rowptr, perm, w, n_rows, out);
    hipLaunchKernelGGL(long_row_sums, dim3(n_rows < 2048 ? 1 : 2048), dim3(256), 0, s, rowptr, perm, w, n_rows, 0, out);
    return check_launch("csr_row_sum");
}

extern "C" int pygsd_degree_scale_f32(const int64_t* row, const int64_t* col, const float* w, const float* deg,
                                      int64_t n_edges, int32_t mode, float* out, void* stream)
{
    PYGSD_REQUIRE(n_edges >= 0 && (mode == 0 || mode == 1), "pygsd_degree_scale_f32: bad arguments");
    if (n_edges == 0) return 0;
    PYGSD_REQUIRE(row && col && w && deg && out, "pygsd_degree_scale_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    hipLaunchKernelGGL(degree_scale, dim3(grid_for(n_edges)), dim3(kBlock), 0, s, row, col, w, deg, n_edges, mode, out);
    return check_launch("degree_scale");
}
