// CSR SpMM / segment-reduce kernels for gfx950 (CDNA4, wave64) -- the gather -> scale -> reduce
// hot path.  HBM-bound: ~0.5 flop per byte, so no MFMA here; what matters is 16-byte-per-lane
// coalesced gathers of whole feature rows, many independent row fetches in flight per wave, and no
// atomics (one wavefront owns an output row, deterministic summation order).
//
// Mapping (vector path, F % 4 == 0, 16-byte aligned rows):
//   one wavefront  <-> one output row r
//   LPR lanes      <-> one gathered feature row (LPR * float4 = F floats; F = 64 -> 16 lanes)
//   NPW = 64 / LPR <-> neighbours fetched by ONE wave-wide global_load_dwordx4 (F = 64 -> 4 rows,
//                      1 KiB per instruction)
//   the row's (col, val) entries are loaded 64 at a time with one coalesced load each, kept in
//   registers, and handed to the lane groups through the LDS crossbar (ds_bpermute) -- the
//   per-wavefront "row tile" never touches LDS memory or HBM twice.
//   UNROLL independent gathers are issued before the first FMA (16 x 16 B per lane in flight; measured:
//   deeper unroll at 4-5 waves/SIMD beats 8 waves/SIMD with fewer loads per wave by 15 %).
//   Epilogue: butterfly over the NPW lane groups, fused alpha / mean / beta*Z, one float4 store.
#include "common.hpp"

#include <cstdlib>

namespace pygsd {
namespace {

struct SpmmArgs {
    const int32_t* rowptr;
    const int32_t* col;
    const float* va;
    const float* vb;
    const float* xa;
    const float* xb;
    float* ya;
    float* yb;
    const float* za;
    const float* zb;
    int64_t ldx, ldy, ldz;
    int32_t n_rows, n_feat;
    float alpha, beta;
    int32_t mean;
    int32_t skip_longer_than;   // > 0: rows with more entries are left to the long-row path
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void fma4(float4& acc, float s, const float4& v)
{
    acc.x = fmaf(s, v.x, acc.x);
    acc.y = fmaf(s, v.y, acc.y);
    acc.z = fmaf(s, v.z, acc.z);
    acc.w = fmaf(s, v.w, acc.w);
}

template <int LPR>
__device__ __forceinline__ void reduce_groups(float4& a)
{
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        a.x += __shfl_xor(a.x, off);
        a.y += __shfl_xor(a.y, off);
        a.z += __shfl_xor(a.z, off);
        a.w += __shfl_xor(a.w, off);
    }
}

__device__ __forceinline__ float4 finish(float4 acc, float alpha, float beta, bool mean, int deg,
                                         const float* z)
{
    if (mean) {
        const float d = static_cast<float>(deg > 1 ? deg : 1);
        acc.x /= d; acc.y /= d; acc.z /= d; acc.w /= d;
    }
    acc.x *= alpha; acc.y *= alpha; acc.z *= alpha; acc.w *= alpha;
    if (z) {
        const float4 zz = ld4(z);
        acc.x = fmaf(beta, zz.x, acc.x);
        acc.y = fmaf(beta, zz.y, acc.y);
        acc.z = fmaf(beta, zz.z, acc.z);
        acc.w = fmaf(beta, zz.w, acc.w);
    }
    return acc;
}

constexpr int kWavesPerBlock = 4;

// One gather step: UN independent 16-byte row fetches per lane (UN * NPW neighbours per wavefront) are
// issued before the first FMA.  Entry idx of the row chunk is handed to lane group `sub` through the LDS
// crossbar; lanes past the end of the chunk (or past F) are predicated off.
template <int LPR, bool DUAL, int UN>
__device__ __forceinline__ void gather_step(int u, int cnt, int sub, bool fact, int c, float wa, float wb,
                                            const float* xa, const float* xb, int64_t ldx, float4& acc_a,
                                            float4& acc_b)
{
    constexpr int NPW = 64 / LPR;
    float4 ga[UN];
    float4 gb[UN];
    float sa[UN];
    float sb[UN];
#pragma unroll
    for (int k = 0; k < UN; ++k) {
        const int idx = u + k * NPW + sub;
        const bool ok = fact && idx < cnt;
        const int cj = __shfl(c, idx & 63);
        const float ta = __shfl(wa, idx & 63);
        sa[k] = ok ? ta : 0.f;
        ga[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (DUAL) {
            const float tb = __shfl(wb, idx & 63);
            sb[k] = ok ? tb : 0.f;
            gb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (ok) {
            const int64_t off = static_cast<int64_t>(cj) * ldx;
            ga[k] = ld4(xa + off);
            if (DUAL) gb[k] = ld4(xb + off);
        }
    }
#pragma unroll
    for (int k = 0; k < UN; ++k) {
        fma4(acc_a, sa[k], ga[k]);
        if (DUAL) fma4(acc_b, sb[k], gb[k]);
    }
}

// second summation level of a long row (spmm_vec_kernel): slot (+)= the batch's partial sum, the accumulator starts again
__device__ __forceinline__ void flush_level(float4& slot, float4& acc, bool first)
{
    if (first) {
        slot = acc;
    } else {
        const float4 t = slot;
        slot = make_float4(t.x + acc.x, t.y + acc.y, t.z + acc.z, t.w + acc.w);
    }
    acc = make_float4(0.f, 0.f, 0.f, 0.f);
}

// DEEP = true : deepest step 16 (8 per operator in the dual kernel) row fetches per lane in flight, ~125
//               VGPRs, 4 waves/SIMD -- measured best on rows with >= 28 (dual) / 48 (single) neighbours
//               (the 41-neighbour benchmark rows of the dual kernel: +1.5 % over the light variant).
// DEEP = false: deepest step 4, <= 64 VGPRs, 8 waves/SIMD -- low-degree rows (signed SBM parts with 6-15
//               neighbours) issue few gathers each and need the wavefront count instead (the deep variant
//               is 1.8x SLOWER there).  The host picks by nnz / n_rows.
template <int LPR, bool DUAL, bool DEEP>
__global__ __launch_bounds__(kWavesPerBlock * 64) void spmm_vec_kernel(SpmmArgs p)
{
    constexpr int NPW = 64 / LPR;
    constexpr int UB = DEEP ? (DUAL ? 8 : 16) / (LPR >= 32 ? 2 : 1) : (DUAL ? 2 : 4);
    const int lane = threadIdx.x & 63;
    // wave-uniform row id -> rowptr is fetched with scalar loads
    const int row = __builtin_amdgcn_readfirstlane(
        static_cast<int>(blockIdx.x) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6));
    if (row >= p.n_rows) return;
    const int sub = lane / LPR;
    const int fl = static_cast<int>(blockIdx.y) * (LPR * 4) + (lane % LPR) * 4;
    const bool fact = fl < p.n_feat;
    const int beg = p.rowptr[row];
    const int end = p.rowptr[row + 1];
    if (p.skip_longer_than > 0 && end - beg > p.skip_longer_than) return;   // hub row: spmm_long_kernel

    float4 acc_a = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc_b = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xa = p.xa + fl;
    const float* xb = DUAL ? p.xb + fl : nullptr;
    // Rows of more than one 64-entry batch: a second summation level.  Each batch's partial sum is added to a per-lane slot in
    // LDS and the register accumulator starts again from zero, so a lane group's chain is 64 / NPW entries long, then one add per
    // batch -- a 4 000-entry row errs like a sum of ~60 terms, not ~1 000 (round 6: the randomised tests' outliers were such rows,
    // summed like the reference's sequential scatter).  The slot lives in LDS, not in registers: the deep dual variant sits 6
    // registers under its 4-wavefront budget.  Rows of one batch (every row of the benchmark graphs) never touch it: bitwise as before.
    __shared__ float4 lvl2[kWavesPerBlock * 64 * (DUAL ? 2 : 1)];
    float4* slot = lvl2 + (threadIdx.x * (DUAL ? 2 : 1));
    const bool multi = end - beg > 64;                  // (wave-uniform)

    for (int base = beg; base < end; base += 64) {
        const int cnt = (end - base) < 64 ? (end - base) : 64;
        int c = 0;
        float wa = 0.f, wb = 0.f;
        if (lane < cnt) {
            // the CSR streams are read exactly once: non-temporal, so they do not evict gathered X rows
            c = __builtin_nontemporal_load(p.col + base + lane);
            wa = p.va ? __builtin_nontemporal_load(p.va + base + lane) : 1.f;
            if (DUAL) wb = __builtin_nontemporal_load(p.vb + base + lane);
        }
        int u = 0;
        for (; cnt - u >= NPW * UB; u += NPW * UB)
            gather_step<LPR, DUAL, UB>(u, cnt, sub, fact, c, wa, wb, xa, xb, p.ldx, acc_a, acc_b);
        if (UB >= 8 && cnt - u >= NPW * (UB / 2)) {
            gather_step<LPR, DUAL, (UB >= 8 ? UB / 2 : 1)>(u, cnt, sub, fact, c, wa, wb, xa, xb, p.ldx, acc_a, acc_b);
            u += NPW * (UB / 2);
        }
        if (UB >= 16 && cnt - u >= NPW * (UB / 4)) {
            gather_step<LPR, DUAL, (UB >= 16 ? UB / 4 : 1)>(u, cnt, sub, fact, c, wa, wb, xa, xb, p.ldx, acc_a, acc_b);
            u += NPW * (UB / 4);
        }
        for (; u < cnt; u += NPW * 2)
            gather_step<LPR, DUAL, 2>(u, cnt, sub, fact, c, wa, wb, xa, xb, p.ldx, acc_a, acc_b);
        if (multi) {
            flush_level(slot[0], acc_a, base == beg);
            if (DUAL) flush_level(slot[DUAL ? 1 : 0], acc_b, base == beg);
        }
    }
    if (multi) {
        acc_a = slot[0];
        if (DUAL) acc_b = slot[DUAL ? 1 : 0];
    }

    reduce_groups<LPR>(acc_a);
    if (DUAL) reduce_groups<LPR>(acc_b);

    if (sub == 0 && fact) {
        const int deg = end - beg;
        const int64_t yo = static_cast<int64_t>(row) * p.ldy + fl;
        const int64_t zo = static_cast<int64_t>(row) * p.ldz + fl;
        st4(p.ya + yo, finish(acc_a, p.alpha, p.beta, p.mean != 0, deg, p.za ? p.za + zo : nullptr));
        if (DUAL)
            st4(p.yb + yo, finish(acc_b, p.alpha, p.beta, false, deg, p.zb ? p.zb + zo : nullptr));
    }
}

// ------------------------------------------------------------------------------------------
// K = 1 magnetic layer, forward, F_in = F_out = 64 (the north-star shape): the dense stage in the dual SpMM's epilogue.
//   T1_r = S_r^T x_r, T1_i = S_i^T x_i (written: the backward needs them)
//   out_real = (x_r - x_i) W_0 + (T1_r - T1_i) W_1 + b ,  out_imag = (x_r + x_i) W_0 + (T1_r + T1_i) W_1 + b
// (MagNetConv.py:189-247 for K = 1).  After the butterfly every lane group holds the row's two products; the 64 lanes
// then compute one output column each: 128 features are broadcast with v_readlane (an SGPR operand of the FMA), W sits in LDS
// ([128][64], lane j reads column j: conflict-free), 256 FMAs per lane on a kernel whose VALU is idle two thirds of the time
// (it is bound by the gathers).  Saves the separate dense pass' re-read of T1 and a launch; costs LDS (32 KB per block) and
// ~600 VALU instructions per row.  Behind PYGSD_FUSE_K1 / dense.set_fused_k1 -- measured both ways, see DESIGN.md.
// ------------------------------------------------------------------------------------------
struct SpmmK1Args {
    SpmmArgs s;
    const float* w;         // [2][64][64]
    const float* bias;      // [64] or null
    float* out_r;
    float* out_i;
    int64_t ldo;
};

template <bool DEEP>
__global__ __launch_bounds__(kWavesPerBlock * 64) void spmm2_k1_dense_kernel(SpmmK1Args q)
{
    constexpr int LPR = 16, NPW = 4;
    constexpr int UB = DEEP ? 8 : 2;
    __shared__ float wl[128 * 64];
    const SpmmArgs& p = q.s;
    for (int i = threadIdx.x; i < 128 * 64; i += kWavesPerBlock * 64) wl[i] = q.w[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(
        static_cast<int>(blockIdx.x) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6));
    if (row >= p.n_rows) return;
    const int sub = lane / LPR;
    const int fl = (lane % LPR) * 4;
    const int beg = p.rowptr[row];
    const int end = p.rowptr[row + 1];
    float4 acc_a = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc_b = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xa = p.xa + fl;
    const float* xb = p.xb + fl;
    __shared__ float4 lvl2[kWavesPerBlock * 64 * 2];
    float4* slot = lvl2 + threadIdx.x * 2;
    const bool multi = end - beg > 64;
    for (int base = beg; base < end; base += 64) {
        const int cnt = (end - base) < 64 ? (end - base) : 64;
        int c = 0;
        float wa = 0.f, wb = 0.f;
        if (lane < cnt) {
            c = __builtin_nontemporal_load(p.col + base + lane);
            wa = __builtin_nontemporal_load(p.va + base + lane);
            wb = __builtin_nontemporal_load(p.vb + base + lane);
        }
        int u = 0;
        for (; cnt - u >= NPW * UB; u += NPW * UB)
            gather_step<LPR, true, UB>(u, cnt, sub, true, c, wa, wb, xa, xb, p.ldx, acc_a, acc_b);
        if (UB >= 8 && cnt - u >= NPW * (UB / 2)) {
            gather_step<LPR, true, (UB >= 8 ? UB / 2 : 1)>(u, cnt, sub, true, c, wa, wb, xa, xb, p.ldx, acc_a, acc_b);
            u += NPW * (UB / 2);
        }
        for (; u < cnt; u += NPW * 2)
            gather_step<LPR, true, 2>(u, cnt, sub, true, c, wa, wb, xa, xb, p.ldx, acc_a, acc_b);
        if (multi) {                                    // (the same second level as spmm_vec_kernel: bitwise the composed path)
            flush_level(slot[0], acc_a, base == beg);
            flush_level(slot[1], acc_b, base == beg);
        }
    }
    if (multi) {
        acc_a = slot[0];
        acc_b = slot[1];
    }
    // the row's own features (T_0): loaded behind the gather loop, so that the loop keeps the plain kernel's register count
    // (and with it 4 wavefronts per SIMD); their latency overlaps the butterfly and the T_1 stores
    const float4 x0a = ld4(xa + static_cast<int64_t>(row) * p.ldx);
    const float4 x0b = ld4(xb + static_cast<int64_t>(row) * p.ldx);
    reduce_groups<LPR>(acc_a);
    reduce_groups<LPR>(acc_b);
    if (sub == 0) {
        const int64_t yo = static_cast<int64_t>(row) * p.ldy + fl;
        st4(p.ya + yo, acc_a);
        st4(p.yb + yo, acc_b);
    }
    // lane group 0 (lanes 0..15) holds feature quads 0..15 of all four 64-vectors
    const float d0[4] = {x0a.x - x0b.x, x0a.y - x0b.y, x0a.z - x0b.z, x0a.w - x0b.w};
    const float s0[4] = {x0a.x + x0b.x, x0a.y + x0b.y, x0a.z + x0b.z, x0a.w + x0b.w};
    const float d1[4] = {acc_a.x - acc_b.x, acc_a.y - acc_b.y, acc_a.z - acc_b.z, acc_a.w - acc_b.w};
    const float s1[4] = {acc_a.x + acc_b.x, acc_a.y + acc_b.y, acc_a.z + acc_b.z, acc_a.w + acc_b.w};
    float o_r = 0.f, o_i = 0.f;
    const float* wcol = wl + lane;                                  // column j = lane of W_0 (rows 0..63) and W_1 (64..127)
    // (partially unrolled: fully unrolled, the 128 LDS reads are hoisted in front of the FMAs and the kernel needs 150 VGPRs --
    // one wavefront per SIMD fewer than the plain dual kernel, whose occupancy is what the gathers live on)
#pragma unroll 2
    for (int qd = 0; qd < 16; ++qd) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const float w0 = wcol[(4 * qd + cc) * 64];
            const float dv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d0[cc]), qd));
            const float sv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s0[cc]), qd));
            o_r = fmaf(dv, w0, o_r);
            o_i = fmaf(sv, w0, o_i);
        }
    }
#pragma unroll 2
    for (int qd = 0; qd < 16; ++qd) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const float w1 = wcol[(64 + 4 * qd + cc) * 64];
            const float dv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d1[cc]), qd));
            const float sv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s1[cc]), qd));
            o_r = fmaf(dv, w1, o_r);
            o_i = fmaf(sv, w1, o_i);
        }
    }
    const float bj = q.bias ? q.bias[lane] : 0.f;
    const int64_t oo = static_cast<int64_t>(row) * q.ldo + lane;
    q.out_r[oo] = o_r + bj;
    q.out_i[oo] = o_i + bj;
}

// ------------------------------------------------------------------------------------------
// Hub rows (power-law tails).  One wavefront per row makes a row with 10^5..10^6 entries the critical path
// (27 us per 1000 entries: 27 ms for a 1M-entry hub against 0.14 ms for the rest of a 4M-entry operator).
// Rows longer than PYGSD_LONG_ROW entries are therefore skipped by the main kernel and handled here:
// block (r, s) reduces segment s (kLongSeg entries, 4 wavefronts x 1024) of long row r into a partial
// feature row, a second kernel adds the partials IN SEGMENT ORDER and applies the epilogue --
// deterministic, no atomics.
// ------------------------------------------------------------------------------------------
constexpr int kLongSeg = 4096;

template <int LPR, bool DUAL>
__global__ __launch_bounds__(256) void spmm_long_kernel(SpmmArgs p, const int32_t* __restrict__ long_rows,
                                                        float* __restrict__ part_a, float* __restrict__ part_b,
                                                        int n_seg)
{
    constexpr int NPW = 64 / LPR;
    constexpr int UB = (DUAL ? 8 : 16) / (LPR >= 32 ? 2 : 1);
    __shared__ float4 sm[4][2][LPR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = long_rows[blockIdx.x];
    const int seg = blockIdx.y;
    const int sub = lane / LPR;
    const int fl = static_cast<int>(blockIdx.z) * (LPR * 4) + (lane % LPR) * 4;
    const bool fact = fl < p.n_feat;
    const int row_end = p.rowptr[row + 1];
    if (p.rowptr[row] + seg * kLongSeg >= row_end) return;      // block-uniform: shorter hub than the longest
    int beg = p.rowptr[row] + seg * kLongSeg + wave * (kLongSeg / 4);
    int end = beg + kLongSeg / 4;
    if (end > row_end) end = row_end;
    float4 acc_a = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc_b = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xa = p.xa + fl;
    const float* xb = DUAL ? p.xb + fl : nullptr;
    for (int base = beg; base < end; base += 64) {
        const int cnt = (end - base) < 64 ? (end - base) : 64;
        int c = 0;
        float wa = 0.f, wb = 0.f;
        if (lane < cnt) {
            c = __builtin_nontemporal_load(p.col + base + lane);
            wa = p.va ? __builtin_nontemporal_load(p.va + base + lane) : 1.f;
            if (DUAL) wb = __builtin_nontemporal_load(p.vb + base + lane);
        }
        int u = 0;
        for (; cnt - u >= NPW * UB; u += NPW * UB)
            gather_step<LPR, DUAL, UB>(u, cnt, sub, fact, c, wa, wb, xa, xb, p.ldx, acc_a, acc_b);
        for (; u < cnt; u += NPW * 2)
            gather_step<LPR, DUAL, 2>(u, cnt, sub, fact, c, wa, wb, xa, xb, p.ldx, acc_a, acc_b);
    }
    reduce_groups<LPR>(acc_a);
    if (DUAL) reduce_groups<LPR>(acc_b);
    if (sub == 0) {
        sm[wave][0][lane] = acc_a;
        if (DUAL) sm[wave][1][lane] = acc_b;
    }
    __syncthreads();
    if (wave == 0 && sub == 0 && fact) {
        const int64_t o = (static_cast<int64_t>(blockIdx.x) * n_seg + seg) * p.n_feat + fl;
        float4 a = sm[0][0][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 t = sm[w][0][lane];
            a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
        st4(part_a + o, a);
        if (DUAL) {
            float4 b = sm[0][1][lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float4 t = sm[w][1][lane];
                b.x += t.x; b.y += t.y; b.z += t.z; b.w += t.w;
            }
            st4(part_b + o, b);
        }
    }
}

// one thread per (long row, feature quad): add the segment partials in order, apply the epilogue
template <bool DUAL>
__global__ __launch_bounds__(256) void spmm_long_finish_kernel(SpmmArgs p, const int32_t* __restrict__ long_rows,
                                                               int n_long, const float* __restrict__ part_a,
                                                               const float* __restrict__ part_b, int n_seg)
{
    const int quads = p.n_feat / 4;
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= static_cast<int64_t>(n_long) * quads) return;
    const int r = static_cast<int>(t / quads), fl = static_cast<int>(t - static_cast<int64_t>(r) * quads) * 4;
    const int row = long_rows[r];
    const int deg = p.rowptr[row + 1] - p.rowptr[row];
    const int used = (deg + kLongSeg - 1) / kLongSeg;          // segments past the row's end hold zeros
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < used && s < n_seg; ++s) {
        const int64_t o = (static_cast<int64_t>(r) * n_seg + s) * p.n_feat + fl;
        const float4 x = ld4(part_a + o);
        a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
        if (DUAL) {
            const float4 y = ld4(part_b + o);
            b.x += y.x; b.y += y.y; b.z += y.z; b.w += y.w;
        }
    }
    const int64_t yo = static_cast<int64_t>(row) * p.ldy + fl;
    const int64_t zo = static_cast<int64_t>(row) * p.ldz + fl;
    st4(p.ya + yo, finish(a, p.alpha, p.beta, p.mean != 0, deg, p.za ? p.za + zo : nullptr));
    if (DUAL) st4(p.yb + yo, finish(b, p.alpha, p.beta, false, deg, p.zb ? p.zb + zo : nullptr));
}

// ------------------------------------------------------------------------------------------
// Low-degree rows at narrow widths.  One wavefront per row wastes the machine when a row has fewer entries than
// one wave-wide load fetches (NPW = 64 / LPR neighbours: 16 at F = 16, 8 at F = 32, 4 at F = 64): the signed SBM
// parts (5-15 entries per row, SGCNConv / SIMPA) and the column blocks of the sharded grid product (10-20 entries
// per row and phase at 16 + 16 packed floats).  Here every LPR-lane group owns its OWN row -- NPW consecutive rows per
// wavefront -- and a group's accumulator IS the row's result, summed in the row's CSR (= the reference's scatter) order:
// no butterfly.
//
// Round 4: the rows of a wavefront are consecutive, so their entries are ONE contiguous range of the CSR: the
// wavefront reads it 64 entries at a time with one coalesced load per stream (as the row kernel does) and hands entry
// (pos_g + k) to group g through the LDS crossbar.  The first version had every group read its own col / val with
// 4-byte loads inside its loop: a dependent chain rowptr -> col -> gather -> col -> gather (5 memory latencies for a
// 6-entry row; SQ_WAVE_CYCLES showed wavefronts resident for ~9 us each and the kernel at 5.4 TB/s of a cache-resident
// set, profiles/r4a counters) where this one has rowptr -> entries -> all gathers.  Rows the long-row path takes are holes in the
// range and are jumped over.  The host picks this variant from entries per row (launch_spmm).
// ------------------------------------------------------------------------------------------
template <int LPR, bool DUAL>
__global__ __launch_bounds__(kWavesPerBlock * 64) void spmm_packed_kernel(SpmmArgs p)
{
    constexpr int NPW = 64 / LPR;
    constexpr int UN = 4;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int fl = (lane % LPR) * 4;
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t row0 = wave * NPW;
    if (row0 >= p.n_rows) return;                                   // wave-uniform
    const int64_t row = row0 + sub;
    const bool fact = fl < p.n_feat;
    const bool mine_row = row < p.n_rows;
    const int beg = p.rowptr[mine_row ? row : p.n_rows];
    const int end = mine_row ? p.rowptr[row + 1] : beg;
    const bool mine = mine_row && !(p.skip_longer_than > 0 && end - beg > p.skip_longer_than);   // hub row: spmm_long_kernel
    const int wave_end = __shfl(end, 63);                            // end of the wavefront's last row
    const int deg = end - beg;
    int pos = mine ? beg : 0;                                        // next entry of this group's row
    const int stop = mine ? end : 0;
    float4 acc_a = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc_b = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xa = p.xa + fl;
    const float* xb = DUAL ? p.xb + fl : nullptr;
    // (second summation level for rows of more than 64 entries, as in spmm_vec_kernel: here a lane group sums its whole row, so
    //  the slot takes the partial sum of every 64-entry chunk the row takes part in)
    __shared__ float4 lvl2[kWavesPerBlock * 64 * (DUAL ? 2 : 1)];
    float4* slot = lvl2 + (threadIdx.x * (DUAL ? 2 : 1));
    const bool multi = mine && deg > 64;
    while (true) {
        const unsigned long long work = __ballot(pos < stop);
        if (work == 0) break;
        // rows are consecutive: the first group with entries left holds the lowest one
        const int base = __shfl(pos, __ffsll(static_cast<long long>(work)) - 1);
        int c = 0;
        float wa = 0.f, wb = 0.f;
        if (base + lane < wave_end) {
            c = __builtin_nontemporal_load(p.col + base + lane);
            wa = p.va ? __builtin_nontemporal_load(p.va + base + lane) : 1.f;
            if (DUAL) wb = __builtin_nontemporal_load(p.vb + base + lane);
        }
        const int hi = stop < base + 64 ? stop : base + 64;          // this group's entries inside the chunk: [pos, hi)
        for (int e = pos; __any(e < hi); e += UN) {
            float4 ga[UN], gb[UN];
            float sa[UN], sb[UN];
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const int idx = e + k - base;
                const bool ok = fact && e + k < hi;
                const int cj = __shfl(c, idx & 63);
                const float ta = __shfl(wa, idx & 63);
                sa[k] = ok ? ta : 0.f;
                ga[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (DUAL) {
                    const float tb = __shfl(wb, idx & 63);
                    sb[k] = ok ? tb : 0.f;
                    gb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (ok) {
                    const int64_t off = static_cast<int64_t>(cj) * p.ldx;
                    ga[k] = ld4(xa + off);
                    if (DUAL) gb[k] = ld4(xb + off);
                }
            }
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                fma4(acc_a, sa[k], ga[k]);
                if (DUAL) fma4(acc_b, sb[k], gb[k]);
            }
        }
        if (pos < hi) {
            if (multi) {
                flush_level(slot[0], acc_a, pos == beg);
                if (DUAL) flush_level(slot[DUAL ? 1 : 0], acc_b, pos == beg);
            }
            pos = hi;
        }
    }
    if (multi) {
        acc_a = slot[0];
        if (DUAL) acc_b = slot[DUAL ? 1 : 0];
    }
    if (mine && fact) {
        const int64_t yo = row * p.ldy + fl;
        const int64_t zo = row * p.ldz + fl;
        st4(p.ya + yo, finish(acc_a, p.alpha, p.beta, p.mean != 0, deg, p.za ? p.za + zo : nullptr));
        if (DUAL) st4(p.yb + yo, finish(acc_b, p.alpha, p.beta, false, deg, p.zb ? p.zb + zo : nullptr));
    }
}

// The first version (kept for A/B probes, PYGSD_SPMM_PACKED_V1=1): every group loads its own row's col / val inside its loop.
template <int LPR, bool DUAL>
__global__ __launch_bounds__(kWavesPerBlock * 64) void spmm_packed_v1_kernel(SpmmArgs p)
{
    constexpr int NPW = 64 / LPR;
    constexpr int UN = 4;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPR;
    const int fl = (lane % LPR) * 4;
    const int64_t wave = static_cast<int64_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
    const int64_t row = wave * NPW + sub;
    const bool fact = fl < p.n_feat;
    int beg = 0, end = 0;
    bool mine = row < p.n_rows;
    if (mine) {
        beg = p.rowptr[row];
        end = p.rowptr[row + 1];
        if (p.skip_longer_than > 0 && end - beg > p.skip_longer_than) {   // hub row: spmm_long_kernel
            mine = false;
            end = beg;
        }
    }
    const int deg = end - beg;
    float4 acc_a = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc_b = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xa = p.xa + fl;
    const float* xb = DUAL ? p.xb + fl : nullptr;
    for (int e = beg; e < end; e += UN) {
        int cj[UN];
        float wa[UN], wb[UN];
        float4 ga[UN], gb[UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            const bool ok = e + k < end;
            cj[k] = ok ? __builtin_nontemporal_load(p.col + e + k) : 0;
            wa[k] = ok ? (p.va ? __builtin_nontemporal_load(p.va + e + k) : 1.f) : 0.f;
            wb[k] = (DUAL && ok) ? __builtin_nontemporal_load(p.vb + e + k) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            const bool ok = fact && e + k < end;
            const int64_t off = static_cast<int64_t>(cj[k]) * p.ldx;
            ga[k] = ok ? ld4(xa + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (DUAL) gb[k] = ok ? ld4(xb + off) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            fma4(acc_a, wa[k], ga[k]);
            if (DUAL) fma4(acc_b, wb[k], gb[k]);
        }
    }
    if (mine && fact) {
        const int64_t yo = row * p.ldy + fl;
        const int64_t zo = row * p.ldz + fl;
        st4(p.ya + yo, finish(acc_a, p.alpha, p.beta, p.mean != 0, deg, p.za ? p.za + zo : nullptr));
        if (DUAL) st4(p.yb + yo, finish(acc_b, p.alpha, p.beta, false, deg, p.zb ? p.zb + zo : nullptr));
    }
}

// Generic fallback (any F, any alignment): lane <-> feature, neighbours walked sequentially with
// wave-uniform (scalar) col/val loads, 256-byte coalesced row reads.
template <bool DUAL>
__global__ __launch_bounds__(kWavesPerBlock * 64) void spmm_scalar_kernel(SpmmArgs p)
{
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(
        static_cast<int>(blockIdx.x) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6));
    if (row >= p.n_rows) return;
    const int beg = p.rowptr[row];
    const int end = p.rowptr[row + 1];
    const int deg = end - beg;
    for (int f0 = 0; f0 < p.n_feat; f0 += 64) {
        const int f = f0 + lane;
        const bool ok = f < p.n_feat;
        float acc_a = 0.f, acc_b = 0.f;
#pragma unroll 4
        for (int e = beg; e < end; ++e) {
            const int64_t off = static_cast<int64_t>(p.col[e]) * p.ldx + f;
            const float wa = p.va ? p.va[e] : 1.f;
            if (ok) acc_a = fmaf(wa, p.xa[off], acc_a);
            if (DUAL) {
                const float wb = p.vb[e];
                if (ok) acc_b = fmaf(wb, p.xb[off], acc_b);
            }
        }
        if (ok) {
            if (p.mean) acc_a /= static_cast<float>(deg > 1 ? deg : 1);
            acc_a *= p.alpha;
            if (p.za) acc_a = fmaf(p.beta, p.za[static_cast<int64_t>(row) * p.ldz + f], acc_a);
            p.ya[static_cast<int64_t>(row) * p.ldy + f] = acc_a;
            if (DUAL) {
                acc_b *= p.alpha;
                if (p.zb) acc_b = fmaf(p.beta, p.zb[static_cast<int64_t>(row) * p.ldz + f], acc_b);
                p.yb[static_cast<int64_t>(row) * p.ldy + f] = acc_b;
            }
        }
    }
}

template <int LPR, bool DUAL>
void launch_vec(const SpmmArgs& a, bool deep, dim3 grid, dim3 block, hipStream_t stream)
{
    if (deep)
        hipLaunchKernelGGL((spmm_vec_kernel<LPR, DUAL, true>), grid, block, 0, stream, a);
    else
        hipLaunchKernelGGL((spmm_vec_kernel<LPR, DUAL, false>), grid, block, 0, stream, a);
}

template <int LPR, bool DUAL>
void launch_long(const SpmmArgs& a, const pygsd_long_rows& h, int n_seg, unsigned gz, hipStream_t stream)
{
    float* part_a = static_cast<float*>(h.workspace);
    float* part_b = DUAL ? part_a + static_cast<int64_t>(h.n_rows) * n_seg * a.n_feat : nullptr;
    hipLaunchKernelGGL((spmm_long_kernel<LPR, DUAL>), dim3(h.n_rows, n_seg, gz), dim3(256), 0, stream, a, h.rows,
                       part_a, part_b, n_seg);
    const int64_t threads = static_cast<int64_t>(h.n_rows) * (a.n_feat / 4);
    hipLaunchKernelGGL(spmm_long_finish_kernel<DUAL>, dim3(static_cast<unsigned>((threads + 255) / 256)), dim3(256),
                       0, stream, a, h.rows, h.n_rows, part_a, part_b, n_seg);
}

int64_t long_workspace_bytes(int32_t n_long, int32_t max_entries, int32_t n_feat, bool dual)
{
    const int64_t n_seg = (static_cast<int64_t>(max_entries) + kLongSeg - 1) / kLongSeg;
    return static_cast<int64_t>(n_long) * n_seg * n_feat * static_cast<int64_t>(sizeof(float)) * (dual ? 2 : 1);
}

// Entries per row below which the rows-per-wavefront variant is used.  Measured (tools/packed_probe.py ->
// profiles/r4_packed_probe.json; 20 M entries, Poisson row lengths, 1 M source rows; speed-up of the packed variant over one
// wavefront per row, round 4's kernel -- round 3's in brackets):
//   single, F = 16 / 32: x3.2 / x2.7 at 4 entries per row (2.8 / 2.0), x1.66 / x1.54 at 8 (1.3), x1.19 / x1.17 at 12 (0.85),
//                        x1.00 / x1.02 at 16, x0.98 / x0.97 at 24, x0.72 / x0.77 at 48
//   single, F = 64     : x1.74 at 4 (1.24), x1.37 at 6, x1.11 at 8 (0.92), x0.97 at 12
//   dual,   F = 16 / 32: x2.2 / x1.9 at 4, x1.17 / x1.08 at 8 (0.88), x0.98 at 12;   dual, F = 64: x1.19 at 4, x1.03 at 6, x0.97 at 8
// (with the coalesced entry chunks the variant no longer collapses on longer rows -- it ties with one wavefront per row up to
// ~32 entries -- but it does not win there either: a row's accumulation is sequential inside its lane group.)
inline int64_t packed_threshold(int lpr, bool dual)
{
    if (dual) return lpr <= 8 ? 10 : lpr == 16 ? 7 : 0;
    return lpr <= 8 ? 20 : lpr == 16 ? 10 : 4;
}

template <bool DUAL>
int launch_spmm(SpmmArgs a, int64_t nnz_hint, const pygsd_long_rows* hubs, hipStream_t stream)
{
    if (a.n_rows == 0 || a.n_feat == 0) return 0;
    const dim3 block(kWavesPerBlock * 64);
    const unsigned gx = (static_cast<unsigned>(a.n_rows) + kWavesPerBlock - 1) / kWavesPerBlock;
    bool vec = (a.n_feat % 4 == 0) && (a.ldx % 4 == 0) && (a.ldy % 4 == 0) && aligned16(a.xa) &&
               aligned16(a.ya);
    if (a.za) vec = vec && (a.ldz % 4 == 0) && aligned16(a.za);
    if (DUAL) {
        vec = vec && aligned16(a.xb) && aligned16(a.yb);
        if (a.zb) vec = vec && aligned16(a.zb);
    }
    ProfScope prof(DUAL ? PYGSD_K_SPMM2 : PYGSD_K_SPMM, stream);
    if (!vec) {
        hipLaunchKernelGGL(spmm_scalar_kernel<DUAL>, dim3(gx), block, 0, stream, a);
        return check_launch("spmm_scalar_kernel");
    }
    const int quads = a.n_feat / 4;
    const bool split = hubs && hubs->n_rows > 0;
    if (split) {
        PYGSD_REQUIRE(hubs->rows && hubs->workspace && hubs->max_entries > PYGSD_LONG_ROW,
                      "pygsd_spmm: long-row descriptor needs rows, workspace and max_entries > PYGSD_LONG_ROW");
        PYGSD_REQUIRE(hubs->workspace_bytes >= long_workspace_bytes(hubs->n_rows, hubs->max_entries, a.n_feat, DUAL),
                      "pygsd_spmm: long-row workspace too small (pygsd_spmm_long_rows_workspace)");
        a.skip_longer_than = PYGSD_LONG_ROW;
    }
    // measured crossover (tools/deep_light_probe.py, F=64, 40M entries): the deep variant wins by 1-4 % from
    // 28 entries per row (dual) / 48 (single) and loses up to 1.8x below; unknown -> light
    const bool deep = nnz_hint >= static_cast<int64_t>(DUAL ? 28 : 48) * a.n_rows;
    // rows-per-wavefront variant for low-degree rows at widths <= 128 (PYGSD_SPMM_PACKED = 0 / 1 forces it off / on:
    // tools/packed_probe.py); unknown degree (nnz_hint == 0) keeps one wavefront per row
    const int lpr = quads <= 4 ? 4 : quads <= 8 ? 8 : quads <= 16 ? 16 : quads <= 32 ? 32 : 64;
    bool packed = false;
    if (lpr <= 32) {
        const char* force = getenv("PYGSD_SPMM_PACKED");
        const char* upto = getenv("PYGSD_SPMM_PACKED_BELOW");       // probes: entries per row below which rows are packed
        const int64_t thr = upto ? atoll(upto) : packed_threshold(lpr, DUAL);
        if (force) packed = force[0] == '1';
        else packed = nnz_hint > 0 && nnz_hint < thr * static_cast<int64_t>(a.n_rows);
    }
    if (packed) {
        const int npw = 64 / lpr;
        const int64_t waves = (static_cast<int64_t>(a.n_rows) + npw - 1) / npw;
        const dim3 pg(static_cast<unsigned>((waves + kWavesPerBlock - 1) / kWavesPerBlock));
        const char* v1env = getenv("PYGSD_SPMM_PACKED_V1");
        const bool v1 = v1env && v1env[0] == '1';
        if (v1) {
            if (lpr == 4) hipLaunchKernelGGL((spmm_packed_v1_kernel<4, DUAL>), pg, block, 0, stream, a);
            else if (lpr == 8) hipLaunchKernelGGL((spmm_packed_v1_kernel<8, DUAL>), pg, block, 0, stream, a);
            else if (lpr == 16) hipLaunchKernelGGL((spmm_packed_v1_kernel<16, DUAL>), pg, block, 0, stream, a);
            else hipLaunchKernelGGL((spmm_packed_v1_kernel<32, DUAL>), pg, block, 0, stream, a);
        } else if (lpr == 4) hipLaunchKernelGGL((spmm_packed_kernel<4, DUAL>), pg, block, 0, stream, a);
        else if (lpr == 8) hipLaunchKernelGGL((spmm_packed_kernel<8, DUAL>), pg, block, 0, stream, a);
        else if (lpr == 16) hipLaunchKernelGGL((spmm_packed_kernel<16, DUAL>), pg, block, 0, stream, a);
        else hipLaunchKernelGGL((spmm_packed_kernel<32, DUAL>), pg, block, 0, stream, a);
    } else if (quads <= 4) {
        launch_vec<4, DUAL>(a, deep, dim3(gx), block, stream);
    } else if (quads <= 8) {
        launch_vec<8, DUAL>(a, deep, dim3(gx), block, stream);
    } else if (quads <= 16) {
        launch_vec<16, DUAL>(a, deep, dim3(gx), block, stream);
    } else if (quads <= 32) {
        launch_vec<32, DUAL>(a, deep, dim3(gx), block, stream);
    } else {
        const unsigned gy = (static_cast<unsigned>(quads) + 63) / 64;
        launch_vec<64, DUAL>(a, deep, dim3(gx, gy), block, stream);
    }
    if (split) {
        const int n_seg = (hubs->max_entries + kLongSeg - 1) / kLongSeg;
        if (quads <= 4) launch_long<4, DUAL>(a, *hubs, n_seg, 1, stream);
        else if (quads <= 8) launch_long<8, DUAL>(a, *hubs, n_seg, 1, stream);
        else if (quads <= 16) launch_long<16, DUAL>(a, *hubs, n_seg, 1, stream);
        else if (quads <= 32) launch_long<32, DUAL>(a, *hubs, n_seg, 1, stream);
        else launch_long<64, DUAL>(a, *hubs, n_seg, (static_cast<unsigned>(quads) + 63) / 64, stream);
    }
    return check_launch("spmm_vec_kernel");
}

// ------------------------------------------------------------------------------------------
// bf16-storage SpMM (BASELINE config C5: DiGCN inception blocks in bf16): X / Z / Y are bf16 in HBM
// (half the gather bytes), edge values and accumulation stay fp32.  Same mapping as the fp32
// kernel with 8 features per lane: LPR = F/8 lanes per gathered row (F = 64 -> 8 lanes x 16 B =
// 128 B), NPW = 64/LPR neighbours per wave-wide load.
// ------------------------------------------------------------------------------------------
struct SpmmBf16Args {
    const int32_t* rowptr;
    const int32_t* col;
    const float* val;
    const uint16_t* x;
    uint16_t* y;
    const uint16_t* z;
    int64_t ldx, ldy, ldz;
    int32_t n_rows, n_feat;
    float alpha, beta;
    int32_t mean;
    int32_t acc_f32;   // != 0: y / z are float arrays (ldy / ldz in floats): fp32 accumulation across launches
};

__device__ __forceinline__ float bf16_to_f32(uint32_t h) { return __uint_as_float(h << 16); }
// two fp32 -> two bf16 in one dword, round to nearest even: ONE v_cvt_pk_bf16_f32 on gfx950 (the integer formulation --
// NaN test, bias add, shift -- costs 6 VALU instructions per value, and this kernel is instruction-issue bound on
// short bf16 rows: ~330 instructions per 26-entry row at 4 cycles each = the measured 1.1 ms at 2M rows)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi)
{
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void unpack8(const uint4& q, float (&v)[8])
{
    v[0] = bf16_to_f32(q.x & 0xffffu); v[1] = bf16_to_f32(q.x >> 16);
    v[2] = bf16_to_f32(q.y & 0xffffu); v[3] = bf16_to_f32(q.y >> 16);
    v[4] = bf16_to_f32(q.z & 0xffffu); v[5] = bf16_to_f32(q.z >> 16);
    v[6] = bf16_to_f32(q.w & 0xffffu); v[7] = bf16_to_f32(q.w >> 16);
}

// Cross-group reduction of the bf16 kernel.  The butterfly of the fp32 kernel would move all 8 accumulators of a lane through
// every stage (24 exchanges + 24 adds at F = 64, where 8 lane groups share a row): this kernel is instruction-issue bound
// on short rows (SQ_INSTS_VALU: 294 VALU instructions per 26-entry row = 0.96 ms of issue at 2M rows against 1.10 ms
// measured), so the reduction is a REDUCE-SCATTER instead -- every stage halves the values a lane carries:
//   lane bit 3 (inside a 16-lane DPP row): keep one half, add the partner's copy of it (row_ror:8 on the add itself)
//   lane bit 4 / bit 5: v_permlane16_swap / v_permlane32_swap exchange the halves of two registers between the partner
//                       rows, so ONE swap + ONE add reduce two values at once
// after which every lane owns ONE output column of the row: 7 adds instead of 24, one conversion and one 2-byte store per
// lane (the 64 lanes of a wavefront still write one contiguous 128-byte row).
__device__ __forceinline__ float add_ror8(float keep, float send)
{
    // keep + (send rotated by 8 lanes inside its 16-lane row)
    return keep + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x128 /* row_ror:8 */,
                                                                        0xf, 0xf, false));
}
// (inline asm, not __builtin_amdgcn_permlane{16,32}_swap: this compiler folds the builtin's two results into one register when
// they are added -- it emitted v_add_f32 v2, v2, v2 after the swap, found in the disassembly after the C5 parity test failed.
// The s_nop pair covers the VALU-write -> permlane-swap-read and swap -> VALU-read wait states the compiler would have inserted.)
__device__ __forceinline__ float add_swap16(float a, float b)
{
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float add_swap32(float a, float b)
{
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

template <int LPR>
__global__ __launch_bounds__(kWavesPerBlock * 64) void spmm_vec_bf16_kernel(SpmmBf16Args p)
{
    constexpr int NPW = 64 / LPR;
    constexpr int UNROLL = 4;
    constexpr bool SCATTER = LPR == 8;           // F = 64 (BASELINE config C5): three halving stages -> one column per lane
    const int lane = threadIdx.x & 63;
    const int row = __builtin_amdgcn_readfirstlane(
        static_cast<int>(blockIdx.x) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6));
    if (row >= p.n_rows) return;
    const int sub = lane / LPR;
    const int fl = static_cast<int>(blockIdx.y) * (LPR * 8) + (lane % LPR) * 8;
    const bool fact = fl < p.n_feat;
    const int beg = p.rowptr[row];
    const int end = p.rowptr[row + 1];
    f32x2_t acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};      // packed fp32: one v_pk_fma_f32 per column pair
    const uint16_t* xb = p.x + fl;
    for (int base = beg; base < end; base += 64) {
        const int cnt = (end - base) < 64 ? (end - base) : 64;
        int c = 0;
        float w = 0.f;
        if (lane < cnt) {
            c = __builtin_nontemporal_load(p.col + base + lane);
            w = p.val ? __builtin_nontemporal_load(p.val + base + lane) : 1.f;
        }
        for (int u = 0; u < cnt; u += NPW * UNROLL) {
            uint4 g[UNROLL];
            float sc[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const int idx = u + k * NPW + sub;
                const bool ok = fact && idx < cnt;
                const int cj = __shfl(c, idx & 63);
                const float t = __shfl(w, idx & 63);
                sc[k] = ok ? t : 0.f;
                g[k] = make_uint4(0u, 0u, 0u, 0u);
                if (ok) g[k] = *reinterpret_cast<const uint4*>(xb + static_cast<int64_t>(cj) * p.ldx);
            }
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const f32x2_t s2 = {sc[k], sc[k]};
                const uint32_t q[4] = {g[k].x, g[k].y, g[k].z, g[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2_t v = {__uint_as_float(q[j] << 16), __uint_as_float(q[j] & 0xffff0000u)};
                    acc[j] = __builtin_elementwise_fma(s2, v, acc[j]);
                }
            }
        }
    }
    const int deg = end - beg;
    const float d = p.mean ? static_cast<float>(deg > 1 ? deg : 1) : 1.f;
    if constexpr (SCATTER) {
        // columns of this lane: fl + 0..7 = acc[0].x, acc[0].y, acc[1].x, ...; lane bits 3 / 4 / 5 = the row's lane group
        const bool b3 = (lane & 8) != 0;
        float r[4];
#pragma unroll
        for (int j = 0; j < 2; ++j) {                       // bit 3: columns {0..3} stay on b3 = 0, {4..7} on b3 = 1
            const float lo0 = acc[j].x, lo1 = acc[j].y, hi0 = acc[j + 2].x, hi1 = acc[j + 2].y;
            r[2 * j] = add_ror8(b3 ? hi0 : lo0, b3 ? lo0 : hi0);
            r[2 * j + 1] = add_ror8(b3 ? hi1 : lo1, b3 ? lo1 : hi1);
        }
        // bit 4: of the 4 columns a lane carries, {0, 1} stay on the even 16-lane rows, {2, 3} on the odd ones
        const float s0 = add_swap16(r[0], r[2]), s1 = add_swap16(r[1], r[3]);
        // bit 5: {0} stays on lanes 0..31, {1} on lanes 32..63
        float v = add_swap32(s0, s1);
        const int col = fl + ((lane >> 3) & 1) * 4 + ((lane >> 4) & 1) * 2 + (lane >> 5);
        if (fact) {
            v = (p.mean ? v / d : v) * p.alpha;
            if (p.acc_f32) {
                // partial products of a phased (sharded) product stay in fp32: the bf16 rounding happens ONCE, at the caller
                const float* zf = reinterpret_cast<const float*>(p.z);
                if (zf) v = fmaf(p.beta, zf[static_cast<int64_t>(row) * p.ldz + col], v);
                reinterpret_cast<float*>(p.y)[static_cast<int64_t>(row) * p.ldy + col] = v;
            } else {
                const float zz = p.z ? bf16_to_f32(p.z[static_cast<int64_t>(row) * p.ldz + col]) : 0.f;
                p.y[static_cast<int64_t>(row) * p.ldy + col] = static_cast<uint16_t>(pack_bf16x2(fmaf(p.beta, zz, v), 0.f));
            }
        }
        return;
    }
    float accs[8] = {acc[0].x, acc[0].y, acc[1].x, acc[1].y, acc[2].x, acc[2].y, acc[3].x, acc[3].y};
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j) accs[j] += __shfl_xor(accs[j], off);
    if (sub == 0 && fact) {
        if (p.acc_f32) {
            const float* zf = reinterpret_cast<const float*>(p.z);
            float* yf = reinterpret_cast<float*>(p.y) + static_cast<int64_t>(row) * p.ldy + fl;
            float r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = (p.mean ? accs[j] / d : accs[j]) * p.alpha;
            if (zf) {
                const float4 z0 = *reinterpret_cast<const float4*>(zf + static_cast<int64_t>(row) * p.ldz + fl);
                const float4 z1 = *reinterpret_cast<const float4*>(zf + static_cast<int64_t>(row) * p.ldz + fl + 4);
                r[0] = fmaf(p.beta, z0.x, r[0]); r[1] = fmaf(p.beta, z0.y, r[1]);
                r[2] = fmaf(p.beta, z0.z, r[2]); r[3] = fmaf(p.beta, z0.w, r[3]);
                r[4] = fmaf(p.beta, z1.x, r[4]); r[5] = fmaf(p.beta, z1.y, r[5]);
                r[6] = fmaf(p.beta, z1.z, r[6]); r[7] = fmaf(p.beta, z1.w, r[7]);
            }
            *reinterpret_cast<float4*>(yf) = make_float4(r[0], r[1], r[2], r[3]);
            *reinterpret_cast<float4*>(yf + 4) = make_float4(r[4], r[5], r[6], r[7]);
            return;
        }
        float zz[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (p.z) unpack8(*reinterpret_cast<const uint4*>(p.z + static_cast<int64_t>(row) * p.ldz + fl), zz);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float r = p.mean ? accs[j] / d : accs[j];
            o[j] = fmaf(p.beta, zz[j], r * p.alpha);
        }
        uint4 q = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]),
                             pack_bf16x2(o[6], o[7]));
        *reinterpret_cast<uint4*>(p.y + static_cast<int64_t>(row) * p.ldy + fl) = q;
    }
}

// ---- two rows per wavefront (round 6 experiment; PYGSD_BF16_PAIR=1) -----------------------------------------------------------
// F = 64 bf16: a gathered row is 8 lanes x 16 bytes, so a wavefront's 64 lanes cover 8 entries per pass and 32 per unrolled
// iteration of spmm_vec_bf16_kernel<8> -- a 26-entry row (C5b's average) leaves 6 of 32 slots idle and pays the whole prologue /
// epilogue for itself.  Here each HALF of the wavefront owns a row: 4 entries per pass, 16 per iteration, the CSR entries of
// both rows fetched by one load, one halving stage fewer in the reduce-scatter (bit 5 of the lane separates the rows), two
// columns per lane stored as one dword.  Control flow is wavefront-uniform: both halves run max(len_0, len_1) worth of passes
// with per-lane predicates.
__global__ __launch_bounds__(kWavesPerBlock * 64) void spmm_vec_bf16_pair_kernel(SpmmBf16Args p)
{
    constexpr int UNROLL = 4;
    const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
    const int pair = static_cast<int>(blockIdx.x) * kWavesPerBlock + static_cast<int>(threadIdx.x >> 6);
    const int row = 2 * pair + half;
    if (2 * pair >= p.n_rows) return;
    const bool live = row < p.n_rows;
    const int sub = hl >> 3;                                   // which of the half's 4 concurrent entries
    const int fl = (hl & 7) * 8;                               // first of this lane's 8 columns
    const int beg = live ? p.rowptr[row] : 0;
    const int end = live ? p.rowptr[row + 1] : 0;
    const int len = end - beg;
    const int other = __shfl_xor(len, 32);
    const int longest = __builtin_amdgcn_readfirstlane(len > other ? len : other);
    f32x2_t acc[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    const uint16_t* xb = p.x + fl;
    for (int off = 0; off < longest; off += 32) {
        const int cnt = (len - off) < 32 ? (len - off) : 32;               // this half's entries of the pass (may be <= 0)
        const int most = (longest - off) < 32 ? (longest - off) : 32;      // wavefront-uniform
        int c = 0;
        float w = 0.f;
        if (hl < cnt) {
            c = __builtin_nontemporal_load(p.col + beg + off + hl);
            w = p.val ? __builtin_nontemporal_load(p.val + beg + off + hl) : 1.f;
        }
        for (int u = 0; u < most; u += 4 * UNROLL) {
            uint4 g[UNROLL];
            float sc[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const int idx = u + k * 4 + sub;
                const bool ok = idx < cnt;
                const int src = (lane & 32) | (idx & 31);                   // the entry sits on a lane of this half
                const int cj = __shfl(c, src);
                const float t = __shfl(w, src);
                sc[k] = ok ? t : 0.f;
                g[k] = make_uint4(0u, 0u, 0u, 0u);
                if (ok) g[k] = *reinterpret_cast<const uint4*>(xb + static_cast<int64_t>(cj) * p.ldx);
            }
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const f32x2_t s2 = {sc[k], sc[k]};
                const uint32_t q[4] = {g[k].x, g[k].y, g[k].z, g[k].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2_t v = {__uint_as_float(q[j] << 16), __uint_as_float(q[j] & 0xffff0000u)};
                    acc[j] = __builtin_elementwise_fma(s2, v, acc[j]);
                }
            }
        }
    }
    // reduce-scatter over the half's 4 lane groups (lane bits 3 and 4): two columns per lane remain
    const bool b3 = (lane & 8) != 0;
    float r[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float lo0 = acc[j].x, lo1 = acc[j].y, hi0 = acc[j + 2].x, hi1 = acc[j + 2].y;
        r[2 * j] = add_ror8(b3 ? hi0 : lo0, b3 ? lo0 : hi0);
        r[2 * j + 1] = add_ror8(b3 ? hi1 : lo1, b3 ? lo1 : hi1);
    }
    float v0 = add_swap16(r[0], r[2]), v1 = add_swap16(r[1], r[3]);
    if (!live) return;
    const int col = fl + ((lane >> 3) & 1) * 4 + ((lane >> 4) & 1) * 2;
    const float d = p.mean ? static_cast<float>(len > 1 ? len : 1) : 1.f;
    v0 = (p.mean ? v0 / d : v0) * p.alpha;
    v1 = (p.mean ? v1 / d : v1) * p.alpha;
    if (p.acc_f32) {
        const float* zf = reinterpret_cast<const float*>(p.z);
        if (zf) {
            const float2 zz = *reinterpret_cast<const float2*>(zf + static_cast<int64_t>(row) * p.ldz + col);
            v0 = fmaf(p.beta, zz.x, v0);
            v1 = fmaf(p.beta, zz.y, v1);
        }
        *reinterpret_cast<float2*>(reinterpret_cast<float*>(p.y) + static_cast<int64_t>(row) * p.ldy + col) = make_float2(v0, v1);
    } else {
        float z0 = 0.f, z1 = 0.f;
        if (p.z) {
            const uint32_t zz = *reinterpret_cast<const uint32_t*>(p.z + static_cast<int64_t>(row) * p.ldz + col);
            z0 = __uint_as_float(zz << 16);
            z1 = __uint_as_float(zz & 0xffff0000u);
        }
        *reinterpret_cast<uint32_t*>(p.y + static_cast<int64_t>(row) * p.ldy + col) =
            pack_bf16x2(fmaf(p.beta, z0, v0), fmaf(p.beta, z1, v1));
    }
}

// 0 = one row per wavefront (default), 1 = two rows per wavefront at F = 64 (PYGSD_BF16_PAIR; measurement / A-B)
int bf16_pair_mode()
{
    static const int mode = [] {
        const char* e = getenv("PYGSD_BF16_PAIR");
        return (e && e[0] == '1') ? 1 : 0;
    }();
    return mode;
}

int launch_spmm_bf16(const SpmmBf16Args& a, hipStream_t stream)
{
    const dim3 block(kWavesPerBlock * 64);
    const unsigned gx = (static_cast<unsigned>(a.n_rows) + kWavesPerBlock - 1) / kWavesPerBlock;
    ProfScope prof(PYGSD_K_SPMM, stream);
    const int oct = a.n_feat / 8;
    if (oct <= 2) {
        hipLaunchKernelGGL(spmm_vec_bf16_kernel<2>, dim3(gx), block, 0, stream, a);
    } else if (oct <= 4) {
        hipLaunchKernelGGL(spmm_vec_bf16_kernel<4>, dim3(gx), block, 0, stream, a);
    } else if (oct == 8 && bf16_pair_mode() == 1) {
        const unsigned gp = ((static_cast<unsigned>(a.n_rows) + 1u) / 2u + kWavesPerBlock - 1) / kWavesPerBlock;
        hipLaunchKernelGGL(spmm_vec_bf16_pair_kernel, dim3(gp), block, 0, stream, a);
    } else if (oct <= 8) {
        hipLaunchKernelGGL(spmm_vec_bf16_kernel<8>, dim3(gx), block, 0, stream, a);
    } else if (oct <= 16) {
        hipLaunchKernelGGL(spmm_vec_bf16_kernel<16>, dim3(gx), block, 0, stream, a);
    } else if (oct <= 32) {
        hipLaunchKernelGGL(spmm_vec_bf16_kernel<32>, dim3(gx), block, 0, stream, a);
    } else {
        const unsigned gy = (static_cast<unsigned>(oct) + 63) / 64;
        hipLaunchKernelGGL(spmm_vec_bf16_kernel<64>, dim3(gx, gy), block, 0, stream, a);
    }
    return check_launch("spmm_vec_bf16_kernel");
}

// ------------------------------------------------------------------------------------------
// SDDMM: out[e] = <A[ia[e]], B[ib[e]]>; 16 lanes per edge, 4 edges per wavefront.
// ------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void sddmm_kernel(const int32_t* __restrict__ ia,
                                                    const int32_t* __restrict__ ib, int64_t nnz,
                                                    const float* __restrict__ A, int64_t lda,
                                                    const float* __restrict__ B, int64_t ldb,
                                                    int32_t n_feat, float* __restrict__ out)
{
    const int t = threadIdx.x & 15;
    const int64_t e = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) >> 4;
    float acc = 0.f;
    if (e < nnz) {
        const float* a = A + static_cast<int64_t>(ia[e]) * lda;
        const float* b = B + static_cast<int64_t>(ib[e]) * ldb;
        if (VEC) {
            for (int f = t * 4; f < n_feat; f += 64) {
                const float4 x = ld4(a + f);
                const float4 y = ld4(b + f);
                acc = fmaf(x.x, y.x, acc);
                acc = fmaf(x.y, y.y, acc);
                acc = fmaf(x.z, y.z, acc);
                acc = fmaf(x.w, y.w, acc);
            }
        } else {
            for (int f = t; f < n_feat; f += 16) acc = fmaf(a[f], b[f], acc);
        }
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (e < nnz && t == 0) out[e] = acc;
}

}  // namespace
}  // namespace pygsd

using namespace pygsd;

extern "C" int pygsd_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                                  const float* X, int64_t ldx, float* Y, int64_t ldy,
                                  const float* Z, int64_t ldz, int32_t n_rows, int32_t n_feat,
                                  float alpha, float beta, int32_t mean, int64_t nnz_hint,
                                  const pygsd_long_rows* long_rows, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0 && n_feat >= 0, "pygsd_spmm_csr_f32: negative size");
    if (n_rows == 0 || n_feat == 0) return 0;
    PYGSD_REQUIRE(rowptr && col && X && Y, "pygsd_spmm_csr_f32: null pointer");
    PYGSD_REQUIRE(ldx >= n_feat && ldy >= n_feat && (!Z || ldz >= n_feat || ldz == 0),
                  "pygsd_spmm_csr_f32: row stride smaller than n_feat");
    SpmmArgs a{rowptr, col, val, nullptr, X, nullptr, Y, nullptr, Z, nullptr,
               ldx, ldy, ldz, n_rows, n_feat, alpha, beta, mean, 0};
    return launch_spmm<false>(a, nnz_hint, long_rows, static_cast<hipStream_t>(stream));
}

extern "C" int pygsd_spmm_csr_bf16(const int32_t* rowptr, const int32_t* col, const float* val,
                                   const void* X, int64_t ldx, void* Y, int64_t ldy, const void* Z,
                                   int64_t ldz, int32_t n_rows, int32_t n_feat, float alpha, float beta,
                                   int32_t mean, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0 && n_feat >= 0, "pygsd_spmm_csr_bf16: negative size");
    if (n_rows == 0 || n_feat == 0) return 0;
    PYGSD_REQUIRE(rowptr && col && X && Y, "pygsd_spmm_csr_bf16: null pointer");
    PYGSD_REQUIRE(n_feat % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && (!Z || ldz % 8 == 0),
                  "pygsd_spmm_csr_bf16: n_feat and row strides must be multiples of 8 (16-byte rows)");
    PYGSD_REQUIRE(aligned16(X) && aligned16(Y) && (!Z || aligned16(Z)), "pygsd_spmm_csr_bf16: pointers must be 16-byte aligned");
    PYGSD_REQUIRE(ldx >= n_feat && ldy >= n_feat && (!Z || ldz >= n_feat || ldz == 0),
                  "pygsd_spmm_csr_bf16: row stride smaller than n_feat");
    SpmmBf16Args a{rowptr, col, val, static_cast<const uint16_t*>(X), static_cast<uint16_t*>(Y),
                   static_cast<const uint16_t*>(Z), ldx, ldy, ldz, n_rows, n_feat, alpha, beta, mean, 0};
    return launch_spmm_bf16(a, static_cast<hipStream_t>(stream));
}

extern "C" int pygsd_spmm_csr_bf16_acc_f32(const int32_t* rowptr, const int32_t* col, const float* val, const void* X,
                                           int64_t ldx, float* Y, int64_t ldy, const float* Z, int64_t ldz,
                                           int32_t n_rows, int32_t n_feat, float alpha, float beta, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0 && n_feat >= 0, "pygsd_spmm_csr_bf16_acc_f32: negative size");
    if (n_rows == 0 || n_feat == 0) return 0;
    PYGSD_REQUIRE(rowptr && col && X && Y, "pygsd_spmm_csr_bf16_acc_f32: null pointer");
    PYGSD_REQUIRE(n_feat % 8 == 0 && ldx % 8 == 0 && ldy % 4 == 0 && (!Z || ldz % 4 == 0),
                  "pygsd_spmm_csr_bf16_acc_f32: n_feat / ldx multiples of 8, fp32 row strides multiples of 4");
    PYGSD_REQUIRE(aligned16(X) && aligned16(Y) && (!Z || aligned16(Z)),
                  "pygsd_spmm_csr_bf16_acc_f32: pointers must be 16-byte aligned");
    PYGSD_REQUIRE(ldx >= n_feat && ldy >= n_feat && (!Z || ldz >= n_feat || ldz == 0),
                  "pygsd_spmm_csr_bf16_acc_f32: row stride smaller than n_feat");
    SpmmBf16Args a{rowptr, col, val, static_cast<const uint16_t*>(X), reinterpret_cast<uint16_t*>(Y),
                   reinterpret_cast<const uint16_t*>(Z), ldx, ldy, ldz, n_rows, n_feat, alpha, beta, 0, 1};
    return launch_spmm_bf16(a, static_cast<hipStream_t>(stream));
}

extern "C" int pygsd_spmm2_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val_a,
                                   const float* val_b, const float* Xa, const float* Xb, int64_t ldx,
                                   float* Ya, float* Yb, int64_t ldy, const float* Za,
                                   const float* Zb, int64_t ldz, int32_t n_rows, int32_t n_feat,
                                   float alpha, float beta, int64_t nnz_hint,
                                   const pygsd_long_rows* long_rows, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0 && n_feat >= 0, "pygsd_spmm2_csr_f32: negative size");
    if (n_rows == 0 || n_feat == 0) return 0;
    PYGSD_REQUIRE(rowptr && col && val_a && val_b && Xa && Xb && Ya && Yb,
                  "pygsd_spmm2_csr_f32: null pointer");
    PYGSD_REQUIRE((Za == nullptr) == (Zb == nullptr), "pygsd_spmm2_csr_f32: Za/Zb must both be set");
    PYGSD_REQUIRE(ldx >= n_feat && ldy >= n_feat && (!Za || ldz >= n_feat),
                  "pygsd_spmm2_csr_f32: row stride smaller than n_feat");
    SpmmArgs a{rowptr, col, val_a, val_b, Xa, Xb, Ya, Yb, Za, Zb,
               ldx, ldy, ldz, n_rows, n_feat, alpha, beta, 0, 0};
    return launch_spmm<true>(a, nnz_hint, long_rows, static_cast<hipStream_t>(stream));
}

extern "C" int pygsd_spmm_long_rows_workspace(int32_t n_long, int32_t max_entries, int32_t n_feat, int32_t dual,
                                              int64_t* bytes)
{
    PYGSD_REQUIRE(bytes, "pygsd_spmm_long_rows_workspace: null pointer");
    PYGSD_REQUIRE(n_long >= 0 && max_entries >= 0 && n_feat >= 0, "pygsd_spmm_long_rows_workspace: negative size");
    *bytes = long_workspace_bytes(n_long, max_entries, n_feat, dual != 0);
    return 0;
}

extern "C" int pygsd_sddmm_coo_f32(const int32_t* ia, const int32_t* ib, int64_t nnz, const float* A,
                                   int64_t lda, const float* B, int64_t ldb, int32_t n_feat,
                                   float* out, void* stream)
{
    PYGSD_REQUIRE(nnz >= 0 && n_feat >= 0, "pygsd_sddmm_coo_f32: negative size");
    if (nnz == 0) return 0;
    PYGSD_REQUIRE(ia && ib && A && B && out, "pygsd_sddmm_coo_f32: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t threads = nnz * 16;
    const unsigned grid = static_cast<unsigned>((threads + 255) / 256);
    const bool vec = (n_feat % 4 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) && aligned16(A) && aligned16(B);
    ProfScope prof(PYGSD_K_SDDMM, s);
    if (vec)
        hipLaunchKernelGGL(sddmm_kernel<true>, dim3(grid), dim3(256), 0, s, ia, ib, nnz, A, lda, B, ldb,
                           n_feat, out);
    else
        hipLaunchKernelGGL(sddmm_kernel<false>, dim3(grid), dim3(256), 0, s, ia, ib, nnz, A, lda, B, ldb,
                           n_feat, out);
    return check_launch("sddmm_kernel");
}

extern "C" int pygsd_spmm2_k1_dense_f32(const int32_t* rowptr, const int32_t* col, const float* val_a, const float* val_b,
                                        const float* Xa, const float* Xb, int64_t ldx, float* Ta, float* Tb, int64_t ldt,
                                        const float* W, const float* bias, float* out_r, float* out_i, int64_t ldo,
                                        int32_t n_rows, int64_t nnz_hint, void* stream)
{
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_spmm2_k1_dense_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(rowptr && col && val_a && val_b && Xa && Xb && Ta && Tb && W && out_r && out_i,
                  "pygsd_spmm2_k1_dense_f32: null pointer");
    PYGSD_REQUIRE(ldx >= 64 && ldt >= 64 && ldo >= 64 && ldx % 4 == 0 && ldt % 4 == 0,
                  "pygsd_spmm2_k1_dense_f32: 64 features, row strides >= 64 and multiples of 4");
    PYGSD_REQUIRE(aligned16(Xa) && aligned16(Xb) && aligned16(Ta) && aligned16(Tb), "pygsd_spmm2_k1_dense_f32: feature matrices "
                  "must be 16-byte aligned");
    SpmmK1Args a{SpmmArgs{rowptr, col, val_a, val_b, Xa, Xb, Ta, Tb, nullptr, nullptr, ldx, ldt, 0, n_rows, 64, 1.f, 0.f, 0, 0},
                 W, bias, out_r, out_i, ldo};
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_SPMM2, s);
    const dim3 block(kWavesPerBlock * 64);
    const unsigned gx = (static_cast<unsigned>(n_rows) + kWavesPerBlock - 1) / kWavesPerBlock;
    if (nnz_hint >= static_cast<int64_t>(28) * n_rows)
        hipLaunchKernelGGL(spmm2_k1_dense_kernel<true>, dim3(gx), block, 0, s, a);
    else
        hipLaunchKernelGGL(spmm2_k1_dense_kernel<false>, dim3(gx), block, 0, s, a);
    return check_launch("spmm2_k1_dense_kernel");
}
