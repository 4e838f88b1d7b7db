// Fused build of the scaled (signed) magnetic operator -- SURVEY.md 8(a) rows a3 / a4, the work the reference redoes on
// EVERY forward unless cached=True (MagNetConv.py:157-181 -> __norm__ :78-120 -> get_magnetic_Laplacian.py:47-85,
// MSConv.py:78-119 -> get_magnetic_signed_Laplacian.py:47-90): edge list in, compute layout out (ONE int32 CSR over the
// symmetrised pattern incl. the diagonal + the four value arrays of include/pygsd_hip.h's pygsd_maglap_assemble_csr).
//
// The generic pipeline of laplacian.hip sorts 2E (64-bit key, 32-bit id) pairs over all ~41 key bits (6 radix passes of
// 12 B per entry) and then makes five single-purpose passes through int64 COO intermediates.  Here:
//   1. edge_keys      one read of the int64 edge list: node-id range check folded in, both orientations emitted as
//                     u64 keys  row << 32 | col << 1 | dir  (self loops / bad ids -> row = n, a bucket nobody reads);
//                     weights (if any) ride along as the 32-bit payload, so nothing is gathered through a permutation later
//   2. rocPRIM radix sort on the ROW bits only (2 passes of 10 bits at 10^6 nodes, 8 B per entry without weights); stable, so a
//      row's entries stay in list order
//   3. key_row_starts row boundaries of the bucketed stream
//   4. row_merge_*    one wavefront per row: the row's (col, dir, position) keys are sorted in registers (bitonic over
//                     the lanes, DPP), duplicate runs are summed in sorted order (coalesce's order), the row's degree is
//                     added sequentially in column order (scatter_add_'s order on the reference's CPU path); every
//                     position of the stream gets ONE 16-byte record {col, A_s, Theta_arg, row} (or a "nothing here"
//                     mark behind a row's distinct entries).  Rows of 65..4096 entries take a block-wide LDS sort;
//                     longer rows are counted and the host falls back to laplacian.hip
//   5. scan of (distinct entries + 1) -> final row pointer; ONE host read of {E_s, #rows left to the fallback, bad id}
//   6. values_entries one thread per record (one 16-byte load): phase / normalisation / 2 x / lambda_max for both
//                     orientations, the diagonal placed by its neighbour entry; a block's outputs are one contiguous
//                     slot range of the final CSR, staged in LDS and written with aligned 16-byte stores
// Narrow accesses were the limiter of the first version (4-byte loads / stores over nine streams ran at 2.4 TB/s whatever
// the arithmetic or the gathers): every large stream here moves 8 or 16 bytes per lane.
// HBM-bound integer / byte work; no atomics on the data path (one counter append per row longer than a wavefront).
// Bit-compatible with the generic pipeline: same formulas in the same order (tests/test_gpu_kernels.py holds the two to
// equality).
#include <climits>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"

namespace pygsd {
namespace {

constexpr int kBlock = 256;
constexpr int kBlockRowMax = 4096;     // rows up to this many symmetrised entries are sorted by one block in LDS

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

// info[0] = E_s (written by finish_info), info[1] = rows too long for this pipeline, info[2] = 1 if an id was outside
// [0, n), info[3] = one such id
__global__ void edge_keys(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                          const float* __restrict__ w, int64_t e, int32_t n, uint64_t* __restrict__ keys,
                          float* __restrict__ wout, int64_t* __restrict__ info)
{
    const uint64_t nn = static_cast<uint64_t>(n), drop = nn << 32;
    GRID_STRIDE(k, e)
    {
        const int64_t r = row[k], c = col[k];
        const bool bad_r = static_cast<uint64_t>(r) >= nn, bad_c = static_cast<uint64_t>(c) >= nn;
        uint64_t kf = drop, kr = drop;
        if (bad_r || bad_c) {
            info[2] = 1;                                   // racing writers all store valid witnesses
            info[3] = bad_r ? r : c;
        } else if (r != c) {
            kf = (static_cast<uint64_t>(r) << 32) | (static_cast<uint64_t>(c) << 1);
            kr = (static_cast<uint64_t>(c) << 32) | (static_cast<uint64_t>(r) << 1) | 1u;
        }
        keys[k] = kf;
        keys[e + k] = kr;
        if (wout) {
            const float we = w[k];
            wout[k] = we;
            wout[e + k] = we;
        }
    }
}

// rs[r] = first entry of the row-bucketed stream whose row is >= r, r = 0 .. n + 1 (row n = dropped entries)
__global__ void key_row_starts(const uint64_t* __restrict__ keys, int64_t m, int32_t n, int32_t* __restrict__ rs)
{
    GRID_STRIDE(i, m + 1)
    {
        const int64_t prev = i == 0 ? -1 : static_cast<int64_t>(keys[i - 1] >> 32);
        const int64_t cur = i == m ? static_cast<int64_t>(n) + 1 : static_cast<int64_t>(keys[i] >> 32);
        for (int64_t r = prev + 1; r <= cur; ++r) rs[r] = static_cast<int32_t>(i);
    }
}

// ---- in-register sort of one value per lane (64 lanes), ascending -------------------------------------------------
// Bitonic network in its "flip" form: merge stage k first pairs lane i with i ^ (k - 1) (mirror inside the k-block),
// then with i ^ j for j = k/4 .. 1; the lower lane of a pair always keeps the minimum, so there are no direction
// flags.  18 of the 21 exchanges stay inside a 16-lane row and are DPP moves (no LDS crossbar round trip, no address
// VGPR, no s_waitcnt): quad_perm for ^1 ^2 ^3, row_half_mirror / row_mirror for ^7 ^15, row_ror:8 for ^8, a
// bank-masked row_shl:4 / row_shr:4 pair for ^4.  The three cross-row exchanges (^16, ^31, ^63) use ds_bpermute.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v)
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ uint32_t dpp_xor4(uint32_t v)
{
    // lanes with bit 2 clear (banks 0, 2 of a row) read lane i + 4 (row_shl:4), the others lane i - 4 (row_shr:4)
    int o = __builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x104, 0xF, 0x5, false);
    o = __builtin_amdgcn_update_dpp(o, static_cast<int>(v), 0x114, 0xF, 0xA, false);
    return static_cast<uint32_t>(o);
}

enum Exchange { X1, X2, X3, X4, X7, X8, X15, X16, X31, X63 };

template <Exchange E>
__device__ __forceinline__ uint32_t exchange32(uint32_t v, int lane)
{
    if constexpr (E == X1) return dpp_mov<0xB1>(v);             // quad_perm:[1,0,3,2]
    else if constexpr (E == X2) return dpp_mov<0x4E>(v);        // quad_perm:[2,3,0,1]
    else if constexpr (E == X3) return dpp_mov<0x1B>(v);        // quad_perm:[3,2,1,0]
    else if constexpr (E == X4) return dpp_xor4(v);
    else if constexpr (E == X7) return dpp_mov<0x141>(v);       // row_half_mirror
    else if constexpr (E == X8) return dpp_mov<0x128>(v);       // row_ror:8
    else if constexpr (E == X15) return dpp_mov<0x140>(v);      // row_mirror
    else if constexpr (E == X16) return static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, static_cast<int>(v)));
    else if constexpr (E == X31) return static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((lane ^ 31) << 2, static_cast<int>(v)));
    else return static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute((lane ^ 63) << 2, static_cast<int>(v)));
}

template <Exchange E>
__device__ __forceinline__ uint32_t exchange(uint32_t v, int lane) { return exchange32<E>(v, lane); }

template <Exchange E>
__device__ __forceinline__ uint64_t exchange(uint64_t v, int lane)
{
    const uint32_t lo = exchange32<E>(static_cast<uint32_t>(v), lane);
    const uint32_t hi = exchange32<E>(static_cast<uint32_t>(v >> 32), lane);
    return (static_cast<uint64_t>(hi) << 32) | lo;
}

// `lower`: this lane is the lower one of its pair
template <Exchange E, typename KT>
__device__ __forceinline__ KT compare_exchange(KT v, int lane, bool lower)
{
    const KT o = exchange<E>(v, lane);
    const KT mn = v < o ? v : o, mx = v < o ? o : v;
    return lower ? mn : mx;
}

template <typename KT>
__device__ __forceinline__ KT wave_bitonic(KT v, int lane)
{
    const bool b0 = (lane & 1) == 0, b1 = (lane & 2) == 0, b2 = (lane & 4) == 0, b3 = (lane & 8) == 0,
               b4 = (lane & 16) == 0, b5 = (lane & 32) == 0;
    v = compare_exchange<X1>(v, lane, b0);                        // k = 2
    v = compare_exchange<X3>(v, lane, b1);                        // k = 4
    v = compare_exchange<X1>(v, lane, b0);
    v = compare_exchange<X7>(v, lane, b2);                        // k = 8
    v = compare_exchange<X2>(v, lane, b1);
    v = compare_exchange<X1>(v, lane, b0);
    v = compare_exchange<X15>(v, lane, b3);                       // k = 16
    v = compare_exchange<X4>(v, lane, b2);
    v = compare_exchange<X2>(v, lane, b1);
    v = compare_exchange<X1>(v, lane, b0);
    v = compare_exchange<X31>(v, lane, b4);                       // k = 32
    v = compare_exchange<X8>(v, lane, b3);
    v = compare_exchange<X4>(v, lane, b2);
    v = compare_exchange<X2>(v, lane, b1);
    v = compare_exchange<X1>(v, lane, b0);
    v = compare_exchange<X63>(v, lane, b5);                       // k = 64
    v = compare_exchange<X16>(v, lane, b4);
    v = compare_exchange<X8>(v, lane, b3);
    v = compare_exchange<X4>(v, lane, b2);
    v = compare_exchange<X2>(v, lane, b1);
    v = compare_exchange<X1>(v, lane, b0);
    return v;
}

// 32-bit keys, `used` (wavefront-uniform) of them real, the rest ~0 in the upper lanes.  The lower lane of a pair keeps the minimum,
// the upper one the maximum = the median of {mine, partner's, 0 or ~0}: ONE v_med3_u32 behind the lane exchange instead of
// min + max + select (the row kernels are VALU-bound: 153 vector instructions per row before, 72 of them this network).  A merge
// whose upper half holds padding only moves nothing: skipped.
struct BitonicSel {
    uint32_t s0, s1, s2, s3, s4, s5;          // 0 on the lower lane of a pair at distance 1, 2, 4, 8, 16, 32; ~0 on the upper
};

__device__ __forceinline__ BitonicSel bitonic_sel(int lane)
{
    BitonicSel s;
    s.s0 = (lane & 1) ? ~0u : 0u;
    s.s1 = (lane & 2) ? ~0u : 0u;
    s.s2 = (lane & 4) ? ~0u : 0u;
    s.s3 = (lane & 8) ? ~0u : 0u;
    s.s4 = (lane & 16) ? ~0u : 0u;
    s.s5 = (lane & 32) ? ~0u : 0u;
    return s;
}

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t mn = a < b ? a : b, mx = a < b ? b : a;
    const uint32_t t = mn < c ? c : mn;                           // max(min(a, b), c)
    return mx < t ? mx : t;                                       // min(max(a, b), .)
}

template <Exchange E>
__device__ __forceinline__ uint32_t cx(uint32_t v, int lane, uint32_t sel)
{
    return umed3(v, exchange32<E>(v, lane), sel);
}

__device__ __forceinline__ uint32_t wave_bitonic32(uint32_t v, int lane, int used, const BitonicSel& s)
{
    v = cx<X1>(v, lane, s.s0);                                    // k = 2
    v = cx<X3>(v, lane, s.s1);                                    // k = 4
    v = cx<X1>(v, lane, s.s0);
    v = cx<X7>(v, lane, s.s2);                                    // k = 8
    v = cx<X2>(v, lane, s.s1);
    v = cx<X1>(v, lane, s.s0);
    v = cx<X15>(v, lane, s.s3);                                   // k = 16
    v = cx<X4>(v, lane, s.s2);
    v = cx<X2>(v, lane, s.s1);
    v = cx<X1>(v, lane, s.s0);
    if (used > 16) {
        v = cx<X31>(v, lane, s.s4);                               // k = 32
        v = cx<X8>(v, lane, s.s3);
        v = cx<X4>(v, lane, s.s2);
        v = cx<X2>(v, lane, s.s1);
        v = cx<X1>(v, lane, s.s0);
    }
    if (used > 32) {
        v = cx<X63>(v, lane, s.s5);                               // k = 64
        v = cx<X16>(v, lane, s.s4);
        v = cx<X8>(v, lane, s.s3);
        v = cx<X4>(v, lane, s.s2);
        v = cx<X2>(v, lane, s.s1);
        v = cx<X1>(v, lane, s.s0);
    }
    return v;
}

__device__ __forceinline__ float degree_source(float s, float a, int deg_mode)
{
    return deg_mode == 2 ? a / 2.f : (deg_mode == 1 ? fabsf(s / 2.f) : s / 2.f);
}

// One record per position of the row-bucketed stream (16 bytes, written and read as one dwordx4):
//   x = col | (1 << 31 if the row's diagonal goes right AFTER this entry)
//   y = A_s (fp32 bits), z = Theta_arg (fp32 bits)
//   w = row | (1 << 31 if the row's diagonal goes right BEFORE this entry), or kNoEntry behind a row's distinct entries
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kNoEntry = 0xFFFFFFFFu;
constexpr uint32_t kFlag = 0x80000000u;

// A wavefront orders one row of <= 64 symmetrised entries in registers.  deg_mode: 0 = row sums of A_s, 1 = of |A_s|
// (signed, absolute_degree off), 2 = of the |w| sums / 2 (signed, absolute_degree on).  `c2` / `w0` are the row's
// keys (col << 1 | dir) and weights as loaded by lane = list position (prefetched by the caller for several rows).
// `unordered` (round 5: rows whose entries arrived in NO particular order, behind bucket_place_rows_w): a run of three or more entries
// of one neighbour is reported in `*unordered` instead of trusted -- its fp32 sum depends on the order the reference adds them in;
// runs of one or two entries do not (a + b = b + a).
// SHORT_RUNS (with `unordered`): only the first two entries of a run are summed -- one lane shift instead of the ballot-driven
// loop over the longest run of the row; a longer run has been reported, the caller discards the build.
template <typename KT, bool SHORT_RUNS = false>
__device__ __forceinline__ void merge_one_row(int32_t r, int32_t beg, int32_t cnt, uint32_t c2, float w0, bool weighted,
                                              int lane, int32_t deg_mode, int32_t* __restrict__ ucnt,
                                              float* __restrict__ deg, uint4* __restrict__ ent, bool* unordered = nullptr,
                                              float* scratch = nullptr)
{
    const bool have = lane < cnt;
    KT lk = have ? static_cast<KT>((static_cast<KT>(c2) << 6) | static_cast<KT>(lane)) : static_cast<KT>(~static_cast<KT>(0));
    if constexpr (sizeof(KT) == 4)
        lk = wave_bitonic32(lk, lane, cnt, bitonic_sel(lane));    // (one v_med3_u32 per exchange; merges over padding skipped)
    else
        lk = wave_bitonic<KT>(lk, lane);
    // invalid keys sorted last: lanes < cnt hold the row in (col, dir, list position) order
    const int src = static_cast<int>(lk & 63);
    const uint32_t c2s = static_cast<uint32_t>(lk >> 6);
    const uint32_t colv = c2s >> 1;
    const bool rev = (c2s & 1u) != 0;
    const uint32_t prev = __shfl_up(colv, 1);
    const bool head = have && (lane == 0 || prev != colv);
    const uint64_t H = __ballot(head);
    const uint64_t above = lane == 63 ? 0ull : ((H >> (lane + 1)) << (lane + 1));
    const int end = above ? (__ffsll(static_cast<long long>(above)) - 1) : cnt;
    const int len = end - lane;                                   // run length, meaningful on head lanes
    const int u = __popcll(H);
    const uint64_t below = (1ull << lane) - 1ull;
    const int rank = __popcll(H & below);
    const int heads_upto = __popcll(H & (below | (1ull << lane)));
    if (unordered && __ballot(head && len >= 3)) *unordered = true;
    float s = 0.f, t = 0.f, a = 0.f, d = 0.f;
    if (weighted) {
        const float wv = __shfl(w0, src);                         // weight of the entry this lane holds after the sort
        if constexpr (SHORT_RUNS) {
            // the same additions in the same order as the loop below, for its first two turns (0 + w_0, then + w_1)
            const float w1 = __shfl_down(wv, 1);
            const int r1 = __shfl_down(static_cast<int>(rev), 1);
            s = s + wv;
            t = t + (rev ? -wv : wv);
            a = a + fabsf(wv);
            if (len >= 2) {
                s = s + w1;
                t = t + (r1 ? -w1 : w1);
                a = a + fabsf(w1);
            }
        } else {
            for (int j = 0;; ++j) {                               // runs are summed in sorted order, like coalesce
                const bool act = head && j < len;
                if (!__ballot(act)) break;
                const float wj = __shfl(wv, lane + j);
                const int rj = __shfl(static_cast<int>(rev), lane + j);
                if (act) {
                    s = s + wj;
                    t = t + (rj ? -wj : wj);
                    a = a + fabsf(wj);
                }
            }
        }
        // degree: SEQUENTIAL sum over the distinct entries in column order (scatter_add_'s order on the reference's
        // CPU path).  Compact the per-entry sources into lanes 0 .. u-1 (heads go to their rank, the other lanes
        // fill the rest: a permutation), then lane-serial adds.
        const int dest = head ? rank : u + (lane - heads_upto);
        if (scratch) {
            // Round 5: through 64 floats of wavefront-private LDS -- the compaction is the store itself, and ONE lane adds the
            // u values in order, four per ds_read_b128 (the lanes behind the distinct entries hold +0: d + 0 = d bit for bit, d is
            // never -0).  The readlane loop below costs a taken branch and an SGPR round trip per term: ~1.6 k cycles of a
            // 40-entry row.
            scratch[dest] = head ? degree_source(s, a, deg_mode) : 0.f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (lane == 0) {
                for (int k = 0; k < u; k += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(scratch + k);
                    d = d + v.x;
                    d = d + v.y;
                    d = d + v.z;
                    d = d + v.w;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                      // the next row's stores must not overtake these reads
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            const float dense = __int_as_float(
                __builtin_amdgcn_ds_permute(dest << 2, __float_as_int(degree_source(s, a, deg_mode))));
            for (int k = 0; k < u; ++k) d = d + __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dense), k));
        }
    } else {
        // all ones: run sums are small integers and the degree is a sum of at most 64 multiples of 1/2 -- every
        // order of summation gives the same fp32 value, so the row degree is a butterfly instead of 40 serial adds
        const uint64_t D = __ballot(have && rev);
        const uint64_t run = (len >= 64 ? ~0ull : ((1ull << len) - 1ull)) << lane;
        const int n1 = __popcll(D & run);
        s = static_cast<float>(len);
        t = static_cast<float>(len - 2 * n1);
        a = s;
        d = head ? degree_source(s, a, deg_mode) : 0.f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) d += __shfl_xor(d, off);
    }
    const int left = __popcll(__ballot(head && static_cast<int32_t>(colv) < r));   // distinct entries left of the diagonal
    if (have) {                                                   // one 16-byte store per lane: a permutation of the row
        uint4 rec;
        int pos;
        if (head) {
            rec.x = colv | ((left >= 1 && rank == left - 1) ? kFlag : 0u);
            rec.y = __float_as_uint(s / 2.f);
            rec.z = __float_as_uint(t);
            rec.w = static_cast<uint32_t>(r) | ((left == 0 && rank == 0) ? kFlag : 0u);
            pos = beg + rank;
        } else {
            rec = make_uint4(0u, 0u, 0u, kNoEntry);
            pos = beg + u + (lane - heads_upto);
        }
        ent[pos] = rec;
    }
    if (lane == 0) {
        ucnt[r] = u;
        deg[r] = d;                                               // (deg^-1/2: row_tables, one THREAD per row)
    }
}

// kRowsPerWave consecutive rows per wavefront: their row bounds and keys are loaded up front (independent loads in
// flight -- one row per wavefront was latency-bound: ~3 dependent HBM round trips per 40-entry row).
constexpr int kRowsPerWave = 4;

// ST: element type of the row-grouped stream -- the radix sort's 64-bit keys (low word = col << 1 | dir) or, behind the bucket split
// of round 5, 4-byte keys of rows whose entries arrived in NO order (UNORDERED: runs of >= 3 entries are reported in info[1]).
template <typename KT, typename ST = uint64_t, bool UNORDERED = false>
__global__ __launch_bounds__(256) void row_merge_wave(
    const ST* __restrict__ keys, const float* __restrict__ wsorted, const int32_t* __restrict__ rs, int32_t n,
    int64_t m, int32_t deg_mode, int32_t* __restrict__ ucnt, float* __restrict__ deg, uint4* __restrict__ ent,
    int32_t* __restrict__ long_rows, int32_t* __restrict__ n_long, int64_t* __restrict__ info = nullptr)
{
    __shared__ __attribute__((aligned(16))) float deg_scratch[4][64];          // merge_one_row's sequential degree sum
    const int lane = threadIdx.x & 63;
    const int64_t r0 = (static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * kRowsPerWave;
    if (r0 >= n) return;
    const int rows = n - r0 < kRowsPerWave ? static_cast<int>(n - r0) : kRowsPerWave;
    const int32_t bound = rs[r0 + (lane <= rows ? lane : rows)];   // lanes 0 .. rows hold the row bounds
    int32_t beg[kRowsPerWave], cnt[kRowsPerWave];
    uint32_t c2[kRowsPerWave];
    float w0[kRowsPerWave];
#pragma unroll
    for (int j = 0; j < kRowsPerWave; ++j) {
        beg[j] = __builtin_amdgcn_readlane(bound, j < rows ? j : rows);
        cnt[j] = __builtin_amdgcn_readlane(bound, j + 1 < rows ? j + 1 : rows) - beg[j];
    }
    // unconditional loads from clamped addresses (a load under a lane mask gets its s_waitcnt inside the branch and the
    // four rows' loads would serialise); lanes past a row's end read the next rows' keys and are masked afterwards
#pragma unroll
    for (int j = 0; j < kRowsPerWave; ++j) {
        int64_t pos = static_cast<int64_t>(beg[j]) + lane;
        pos = pos < m ? pos : m - 1;
        c2[j] = static_cast<uint32_t>(__builtin_nontemporal_load(keys + pos));      // col << 1 | dir
        w0[j] = wsorted ? wsorted[pos] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < kRowsPerWave; ++j) {
        if (j >= rows) break;
        const int32_t r = static_cast<int32_t>(r0) + j;
        if (cnt[j] > 64) {
            if (lane == 0) long_rows[atomicAdd(n_long, 1)] = r;
            continue;
        }
        if constexpr (UNORDERED) {
            bool unordered = false;
            merge_one_row<KT, true>(r, beg[j], cnt[j], c2[j], w0[j], wsorted != nullptr, lane, deg_mode, ucnt, deg, ent, &unordered,
                                    deg_scratch[__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6))]);
            if (unordered && lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(info + 1), 1ull);
        } else {
            merge_one_row<KT>(r, beg[j], cnt[j], c2[j], w0[j], wsorted != nullptr, lane, deg_mode, ucnt, deg, ent, nullptr,
                              deg_scratch[__builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6))]);
        }
    }
}

// Rows of 65 .. kBlockRowMax entries: one block per listed row, bitonic sort of (col, dir, position) keys in LDS.
template <typename ST = uint64_t, bool UNORDERED = false>
__global__ __launch_bounds__(256) void row_merge_block(
    const ST* __restrict__ keys, const float* __restrict__ wsorted, const int32_t* __restrict__ rs,
    int32_t deg_mode, int32_t* __restrict__ ucnt, float* __restrict__ deg, uint4* __restrict__ ent,
    const int32_t* __restrict__ long_rows, const int32_t* __restrict__ n_long, int64_t* __restrict__ info)
{
    __shared__ uint64_t sk[kBlockRowMax];
    __shared__ float sw[kBlockRowMax];
    __shared__ float sd[kBlockRowMax];
    __shared__ int32_t soff[kBlock + 1];
    __shared__ int32_t sleft;
    const int tid = threadIdx.x;
    const int nl = *n_long;
    for (int li = blockIdx.x; li < nl; li += gridDim.x) {
        const int32_t r = long_rows[li];
        const int32_t beg = rs[r], cnt = rs[r + 1] - beg;
        if (cnt > kBlockRowMax) {                                  // left to the generic pipeline (host falls back)
            if (tid == 0) {
                atomicAdd(reinterpret_cast<unsigned long long*>(info + 1), 1ull);
                ucnt[r] = 0;
                deg[r] = 0.f;
            }
            for (int i = tid; i < cnt; i += kBlock) ent[beg + i] = make_uint4(0u, 0u, 0u, kNoEntry);
            continue;
        }
        int n2 = kBlock;
        while (n2 < cnt) n2 <<= 1;
        for (int i = tid; i < n2; i += kBlock) {
            sk[i] = i < cnt ? ((static_cast<uint64_t>(static_cast<uint32_t>(keys[beg + i])) << 12) | static_cast<uint64_t>(i))
                            : ~0ull;
            if (wsorted && i < cnt) sw[i] = wsorted[beg + i];
        }
        if (tid == 0) sleft = 0;
        __syncthreads();
        for (int k = 2; k <= n2; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (n2 >> 1); t += kBlock) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int l = i | j;
                    const bool up = (i & k) == 0;
                    const uint64_t x = sk[i], y = sk[l];
                    if ((x > y) == up) {
                        sk[i] = y;
                        sk[l] = x;
                    }
                }
                __syncthreads();
            }
        }
        // sk[p] >> 13 = col, bit 12 = dir, low 12 bits = list position inside the row
        const int ch = n2 / kBlock, p0 = tid * ch;
        int heads = 0, left = 0;
        for (int p = p0; p < p0 + ch; ++p)
            if (p < cnt && (p == 0 || (sk[p - 1] >> 13) != (sk[p] >> 13))) {
                ++heads;
                if (static_cast<int64_t>(sk[p] >> 13) < r) ++left;
            }
        soff[tid] = heads;
        if (left) atomicAdd(&sleft, left);
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            for (int i = 0; i < kBlock; ++i) {
                const int h = soff[i];
                soff[i] = acc;
                acc += h;
            }
            soff[kBlock] = acc;
        }
        __syncthreads();
        const int u = soff[kBlock], nleft = sleft;
        int rank = soff[tid];
        for (int p = p0; p < p0 + ch; ++p) {
            if (!(p < cnt && (p == 0 || (sk[p - 1] >> 13) != (sk[p] >> 13)))) continue;
            const uint64_t colv = sk[p] >> 13;
            float s = 0.f, t = 0.f, a = 0.f;
            int run = 0;
            for (int q = p; q < cnt && (sk[q] >> 13) == colv; ++q) {
                const float we = wsorted ? sw[sk[q] & 4095u] : 1.f;
                s = s + we;
                t = t + (((sk[q] >> 12) & 1u) ? -we : we);
                a = a + fabsf(we);
                ++run;
            }
            // (rows that arrived in no order: the position inside the row is not the list position -- a run of three is not ours to sum)
            if (UNORDERED && run >= 3) atomicAdd(reinterpret_cast<unsigned long long*>(info + 1), 1ull);
            uint4 rec;
            rec.x = static_cast<uint32_t>(colv) | ((nleft >= 1 && rank == nleft - 1) ? kFlag : 0u);
            rec.y = __float_as_uint(s / 2.f);
            rec.z = __float_as_uint(t);
            rec.w = static_cast<uint32_t>(r) | ((nleft == 0 && rank == 0) ? kFlag : 0u);
            ent[beg + rank] = rec;
            sd[rank] = degree_source(s, a, deg_mode);
            ++rank;
        }
        for (int i = u + tid; i < cnt; i += kBlock) ent[beg + i] = make_uint4(0u, 0u, 0u, kNoEntry);
        __syncthreads();
        if (tid == 0) {
            float d = 0.f;
            for (int k = 0; k < u; ++k) d = d + sd[k];
            ucnt[r] = u;
            deg[r] = d;
        }
        __syncthreads();
    }
}

struct PlusOne {
    __host__ __device__ int32_t operator()(int32_t v) const { return v + 1; }
};

// Per-row tables of the second stage, one THREAD per row: deg^-1/2 with 0 -> 0 (get_magnetic_Laplacian.py:69-71; powf is
// ~150 instructions -- far too many to issue from a wavefront with one live lane per row) and the offset that turns a
// stream position into a CSR slot (slot = position + shift[row] + (col > row)).
__global__ void row_tables(const float* __restrict__ deg, const int32_t* __restrict__ rs, const int32_t* __restrict__ rowptr,
                           int32_t n, int32_t sym, float* __restrict__ dinv, int32_t* __restrict__ shift,
                           int64_t* __restrict__ info)
{
    GRID_STRIDE(r, n)
    {
        if (sym) {
            const float d = deg[r];
            dinv[r] = d == 0.f ? 0.f : powf(d, -0.5f);
        }
        shift[r] = rowptr[r] - rs[r];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) info[0] = static_cast<int64_t>(rowptr[n]) - n;
}

__device__ __forceinline__ uint4 load_record(const uint4* __restrict__ ent, int64_t p)
{
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ent) + p);     // touched once
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float scale_lam(float x, float lam)
{
    const float v = (2.0f * x) / lam;
    return v == INFINITY ? 0.f : v;
}

// Values of S = 2 L / lambda_max + diag_shift I in the final CSR slots, one thread per record of the stream.
//   vb_* = S[row, col] (by-source / backward product), vf_* = S[col, row] (by-target / forward product):
//   Hermitian, so the mirror is the same magnitude multiplied in the mirrored entry's own order with the
//   conjugate phase (laplacian.hip lap_values / assemble_csr; get_magnetic_Laplacian.py:66-85, MagNetConv.py:100-116).
// The slots a block produces form one contiguous range [lo, hi) of the CSR (entries of consecutive rows + the
// diagonals their neighbours place; the single slot of an entry-less row in between is a hole, filled afterwards by
// diagonal_of_empty_rows), so the five output arrays are staged in LDS and written with aligned 16-byte stores.
constexpr int kStage = 3 * kBlock;      // slots one block can stage; wider ranges (long runs of entry-less rows) store directly

__global__ __launch_bounds__(256) void values_entries(
    const uint4* __restrict__ ent, const int32_t* __restrict__ rs, const int32_t* __restrict__ shift,
    const float* __restrict__ deg, const float* __restrict__ dinv, int32_t n, float two_pi_q, int32_t sym, float lam,
    float diag_shift, int32_t* __restrict__ ccol, float* __restrict__ vb_re, float* __restrict__ vb_im,
    float* __restrict__ vf_re, float* __restrict__ vf_im)
{
    __shared__ float st[5][kStage];
    __shared__ int32_t s_lo[4], s_hi[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t limit = rs[n];                                   // positions of rows < n (behind: dropped entries)
    // unit phase arguments (unweighted graphs, +-1 signs): sincos once per thread; sincosf is odd / even in its argument
    float sn1, cs1;
    sincosf(two_pi_q, &sn1, &cs1);
    const bool lam2 = lam == 2.0f;       // the sym default: (2 x) / 2 is x exactly, no division
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    int64_t p = static_cast<int64_t>(blockIdx.x) * kBlock + tid;
    uint4 rec = p < limit ? load_record(ent, p) : make_uint4(0u, 0u, 0u, kNoEntry);
    for (int64_t base = static_cast<int64_t>(blockIdx.x) * kBlock; base < limit; base += stride) {
        const uint4 cur = rec;
        const int64_t pc = p;
        p += stride;
        rec = p < limit ? load_record(ent, p) : make_uint4(0u, 0u, 0u, kNoEntry);   // next round, in flight
        const bool ok = cur.w != kNoEntry;
        const int32_t row = static_cast<int32_t>(cur.w & ~kFlag), c = static_cast<int32_t>(cur.x & ~kFlag);
        const bool d_before = ok && (cur.w & kFlag) != 0, d_after = ok && (cur.x & kFlag) != 0;
        int64_t slot = 0;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f, dg = 0.f;
        if (ok) {
            slot = pc + shift[row] + (c > row ? 1 : 0);
            const float th = __uint_as_float(cur.z);
            float sn, cs;
            if (fabsf(th) == 1.0f) {
                cs = cs1;
                sn = th < 0.f ? -sn1 : sn1;
            } else {
                sincosf(two_pi_q * th, &sn, &cs);
            }
            float mag = __uint_as_float(cur.y), mmag = mag;
            if (sym) {
                const float ir = dinv[row], ic = dinv[c];
                mag = ir * mag * ic;            // entry (row, col): deg^-1/2[row] * A_s * deg^-1/2[col]
                mmag = ic * mmag * ir;          // mirrored entry (col, row), multiplied in ITS row/col order
            }
            v0 = -(mag * cs);
            v1 = -(mag * sn);
            v2 = -(mmag * cs);
            v3 = mmag * sn;
            if (!lam2) {
                v0 = scale_lam(v0, lam);
                v1 = scale_lam(v1, lam);
                v2 = scale_lam(v2, lam);
                v3 = scale_lam(v3, lam);
            }
            if (d_before || d_after) dg = scale_lam(sym ? 1.f : deg[row], lam) + diag_shift;
        }
        // slot range of the block
        int32_t lo = ok ? static_cast<int32_t>(slot) - (d_before ? 1 : 0) : INT_MAX;
        int32_t hi = ok ? static_cast<int32_t>(slot) + (d_after ? 2 : 1) : INT_MIN;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const int32_t ol = __shfl_xor(lo, off), oh = __shfl_xor(hi, off);
            lo = ol < lo ? ol : lo;
            hi = oh > hi ? oh : hi;
        }
        __syncthreads();                                           // previous round's staging has been drained
        if (lane == 0) {
            s_lo[wv] = lo;
            s_hi[wv] = hi;
        }
        __syncthreads();
        lo = min(min(s_lo[0], s_lo[1]), min(s_lo[2], s_lo[3]));
        hi = max(max(s_hi[0], s_hi[1]), max(s_hi[2], s_hi[3]));
        if (hi <= lo) continue;                                    // no entry in this round (block-uniform)
        const bool staged = hi - lo <= kStage;
        if (ok) {
            const int64_t dslot = d_before ? slot - 1 : slot + 1;
            if (staged) {
                const int o = static_cast<int>(slot - lo);
                st[0][o] = __int_as_float(c);
                st[1][o] = v0;
                st[2][o] = v1;
                st[3][o] = v2;
                st[4][o] = v3;
                if (d_before || d_after) {
                    const int od = static_cast<int>(dslot - lo);
                    st[0][od] = __int_as_float(row);
                    st[1][od] = dg;
                    st[2][od] = 0.f;
                    st[3][od] = dg;
                    st[4][od] = 0.f;
                }
            } else {
                ccol[slot] = c;
                vb_re[slot] = v0;
                vb_im[slot] = v1;
                vf_re[slot] = v2;
                vf_im[slot] = v3;
                if (d_before || d_after) {
                    ccol[dslot] = row;
                    vb_re[dslot] = dg;
                    vb_im[dslot] = 0.f;
                    vf_re[dslot] = dg;
                    vf_im[dslot] = 0.f;
                }
            }
        }
        if (!staged) continue;
        __syncthreads();
        // drain: [lo, hi) of each array; the 16-byte aligned middle as dwordx4, the ragged ends as dwords
        const int32_t a0 = (lo + 3) & ~3, a1 = hi & ~3;
        float* const outs[5] = {reinterpret_cast<float*>(ccol), vb_re, vb_im, vf_re, vf_im};
        if (a0 < a1) {
            const int quads = (a1 - a0) >> 2;
#pragma unroll
            for (int arr = 0; arr < 5; ++arr) {
                for (int g = tid; g < quads; g += kBlock) {
                    const int o = a0 - lo + 4 * g;
                    const f32x4 v = {st[arr][o], st[arr][o + 1], st[arr][o + 2], st[arr][o + 3]};
                    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(outs[arr] + a0) + g);
                }
            }
            const int ends = (a0 - lo) + (hi - a1);                // <= 6 ragged slots
            if (tid < 5 * 8) {
                const int arr = tid >> 3, e = tid & 7;
                if (e < ends) {
                    const int32_t sl = e < a0 - lo ? lo + e : a1 + (e - (a0 - lo));
                    outs[arr][sl] = st[arr][sl - lo];
                }
            }
        } else {                                                   // fewer than 4 aligned slots: all scalar
#pragma unroll
            for (int arr = 0; arr < 5; ++arr)
                for (int sl = lo + tid; sl < hi; sl += kBlock) outs[arr][sl] = st[arr][sl - lo];
        }
    }
}

// diagonal of the rows without any off-diagonal entry (isolated nodes): nobody else writes their single slot
// (runs AFTER values_entries: a staged slot range that spans such a row carries an unset value there)
__global__ void diagonal_of_empty_rows(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ ucnt,
                                       const float* __restrict__ deg, int32_t n, int32_t sym, float lam, float diag_shift,
                                       int32_t* __restrict__ ccol, float* __restrict__ vb_re, float* __restrict__ vb_im,
                                       float* __restrict__ vf_re, float* __restrict__ vf_im)
{
    GRID_STRIDE(r, n)
    {
        if (ucnt[r] != 0) continue;
        const int64_t slot = rowptr[r];
        const float d = scale_lam(sym ? 1.f : deg[r], lam) + diag_shift;
        ccol[slot] = static_cast<int32_t>(r);
        vb_re[slot] = d;
        vf_re[slot] = d;
        vb_im[slot] = 0.f;
        vf_im[slot] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 4: the UNWEIGHTED build (the graphs of the BASELINE magnetic configurations: DSBM edge lists without weights).
// With unit weights a node's degree is half the number of its stream entries (every listed non-loop edge adds 1/2 to A_s at both
// ends, duplicates and reciprocal pairs included): it is read off the row bounds, so deg^-1/2 of EVERY node exists before a row is
// merged, and multiplicity and phase argument of an entry are small integers.  The two stages then shrink to
//   A  unit_merge_rows   a wavefront orders a row in registers and parks the MERGED row: ONE 8-byte record per DISTINCT
//                        neighbour (col | multiplicity | Theta_arg) instead of a 16-byte record per stream position
//   -  scan of (distinct + 1) -> row pointer
//   B  unit_write_chunks a workgroup per 16 rows walks the merged rows (coalesced 8-byte loads, every lane busy), forms the values
//                        (same formulas, same order as values_entries: bit-compatible) and writes a wavefront's contiguous slot
//                        range through LDS with aligned 16-byte stores
// i.e. 330 + 330 MB of intermediate traffic instead of 640 + 640 at the north star, no shift[] table, no per-record row lookups.
// (Tried first and measured: A and B in ONE kernel with the row pointer chained through it by a decoupled look-back -- no
// intermediate at all.  Bit-identical, but the chain's waiting dominated: 5.4 ms with 16-row blocks, 1.8 ms with 256-row blocks
// against 0.88 ms for the two kernels it replaced; commit "Unweighted operator build in one kernel behind the sort".)
// Rows of 65 .. kUnitRowMax entries are rank-sorted through a small LDS scratch by their wavefront; longer rows are counted and the
// host takes the two-stage pipeline above.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kUnitRowMax = 512;

__global__ void unit_finish_info(const int32_t* __restrict__ rowptr, int32_t n, int64_t* __restrict__ info)
{
    info[0] = static_cast<int64_t>(rowptr[n]) - n;                // E_s for the host
}

// deg^-1/2 of a node with k stream entries (degree k / 2), k = 0 .. kUnitRowMax.  The write kernel gathers the 2-byte entry count
// of an entry's column and looks the factor up in an LDS copy of this table: the gathered table is 2 n bytes instead of 4 n -- it
// stays in the 4 MB L2 of an XCD next to the streams, where the fp32 table missed often enough to add 0.53 GB of fabric reads to the
// kernel's 1.3 GB (profiles/r4w_build_pmc.json).  Same expression as before, so the same bits.
__device__ __forceinline__ void unit_lut(int k, float* __restrict__ lut)
{
    const float d = static_cast<float>(k) / 2.f;                  // (the butterfly of merge_one_row adds multiples of 1/2: exact)
    lut[k] = d == 0.f ? 0.f : powf(d, -0.5f);
}

__device__ __forceinline__ uint16_t clamp16(int32_t c) { return static_cast<uint16_t>(c < 65535 ? c : 65535); }

__global__ void unit_row_tables(const int32_t* __restrict__ rs, int32_t n, float* __restrict__ deg, uint16_t* __restrict__ cnt16,
                                int32_t* __restrict__ row_u, float two_pi_q, float* __restrict__ trig, float* __restrict__ lut)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        row_u[n] = 0;                                             // the scan's (n + 1)-th input
        sincosf(two_pi_q, trig + 1, trig);
    }
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k <= kUnitRowMax; k += blockDim.x) unit_lut(k, lut);
    GRID_STRIDE(r, n)
    {
        const int32_t c = rs[r + 1] - rs[r];
        deg[r] = static_cast<float>(c) / 2.f;                     // = the sum of the row's A_s entries, multiples of 1/2
        cnt16[r] = clamp16(c);
    }
}

struct UnitArgs {
    const uint64_t* keys;          // row-bucketed stream (sorted on the row bits)
    uint64_t* scratch;             // the sort's input buffer (dead): sorted keys of the 65+ entry rows
    const int32_t* rs;             // [n + 2] row bounds of the stream
    const float* deg;
    const uint16_t* cnt16;         // [n] stream entries per row (<= kUnitRowMax where the build is valid): what deg^-1/2 is read off
    const float* lut;              // [kUnitRowMax + 1] deg^-1/2 of a row with k entries -- see unit_lut
    int32_t* rowptr;
    int32_t* ccol;
    float* vb_re;
    float* vb_im;
    float* vf_re;
    float* vf_im;
    int32_t* row_u;                // [n + 1] distinct entries per row (kernel A -> scan -> rowptr)
    int32_t* row_left;             // [n] distinct entries left of the diagonal
    int64_t* info;
    int64_t m;
    int32_t n, sym;
    float two_pi_q, lam, diag_shift;
    float* trig;                   // {cos, sin}(2 pi q), formed once by the first kernel of the build
};

// A chunk of kChunkRows consecutive rows occupies ONE contiguous slot range of the final CSR ([first slot of its first row,
// + sum of (distinct + 1))): with rows of <= 64 entries (<= kChunkRows x 65 slots) the five arrays are staged in LDS and drained with
// aligned 16-byte stores -- 4-byte stores over five streams ran at 2.4 TB/s in the first version of values_entries, whatever else
// changed.  STRIDE: words between the staged arrays; 0 = straight to global memory (rows the staging does not take).
constexpr int kChunkRows = 16;
constexpr int kChunkSlots = kChunkRows * 65 + 4;
constexpr uint64_t kNoRecord = ~0ull;

template <int STRIDE>
__device__ __forceinline__ void unit_store(const UnitArgs& p, float* st, int64_t wslot0, int64_t slot, int32_t c, float v0, float v1,
                                           float v2, float v3)
{
    if constexpr (STRIDE != 0) {
        const int o = static_cast<int>(slot - wslot0);
        st[o] = __int_as_float(c);
        st[STRIDE + o] = v0;
        st[2 * STRIDE + o] = v1;
        st[3 * STRIDE + o] = v2;
        st[4 * STRIDE + o] = v3;
    } else {
        p.ccol[slot] = c;
        p.vb_re[slot] = v0;
        p.vb_im[slot] = v1;
        p.vf_re[slot] = v2;
        p.vf_im[slot] = v3;
    }
}

// `ir`, `ic`: deg^-1/2 of the row and of the column (read by the caller; unused without the normalisation)
template <int STRIDE>
__device__ __forceinline__ void unit_values(const UnitArgs& p, float* st, int64_t wslot0, float ir, float ic, int32_t c, int len,
                                            int theta, float cs1, float sn1, int64_t slot)
{
    // (values_entries, specialised: A_s = len / 2, Theta_arg = theta, both small integers as floats; with weights of +-1 the caller
    // passes len = the sum of the run's weights = entries - 2 x negative ones)
    const float th = static_cast<float>(theta);
    float sn, cs;
    if (fabsf(th) == 1.0f) {
        cs = cs1;
        sn = th < 0.f ? -sn1 : sn1;
    } else {
        sincosf(p.two_pi_q * th, &sn, &cs);
    }
    float mag = static_cast<float>(len) / 2.f, mmag = mag;
    if (p.sym) {
        mag = ir * mag * ic;
        mmag = ic * mmag * ir;
    }
    float v0 = -(mag * cs), v1 = -(mag * sn), v2 = -(mmag * cs), v3 = mmag * sn;
    if (p.lam != 2.0f) {
        v0 = scale_lam(v0, p.lam);
        v1 = scale_lam(v1, p.lam);
        v2 = scale_lam(v2, p.lam);
        v3 = scale_lam(v3, p.lam);
    }
    unit_store<STRIDE>(p, st, wslot0, slot, c, v0, v1, v2, v3);
}

template <int STRIDE>
__device__ __forceinline__ void unit_diagonal(const UnitArgs& p, float* st, int64_t wslot0, int32_t row, int64_t slot)
{
    const float dg = scale_lam(p.sym ? 1.f : p.deg[row], p.lam) + p.diag_shift;
    unit_store<STRIDE>(p, st, wslot0, slot, row, dg, 0.f, dg, 0.f);
}

// One row of <= 64 stream entries, one (col << 1 | dir) key per lane: ordered in registers, duplicates / reciprocal pairs merged;
// ONE 8-byte record per distinct neighbour -- col | multiplicity << 32 | (Theta_arg + 64) << 40 | rank << 48 | (row mod kChunkRows) << 54
// -- parked at scratch[gbeg + rank]; the positions behind a row's distinct entries (duplicates merged away) are marked kNoRecord, so
// the write kernel can walk a chunk's positions without looking at its rows.
// (`cnt`, `r`, `gbeg` wavefront-uniform)
// SIGNED (weights of +-1; 32-bit keys only): the key is  col << 2 | dir << 1 | (w < 0) ; the record also carries the run's number
// of negative entries in bits 25 .. 31 (a column id has <= 24 bits there), and Theta_arg = sum of (+w forward, -w reversed).  Sums
// of +-1 are small integers: exact in every order, like the unit sums -- which is why an unordered LDS placement may feed this.
template <typename KT, bool SIGNED = false>
__device__ __forceinline__ void unit_merge_short(const UnitArgs& p, uint32_t c2, int cnt, int32_t r, int64_t gbeg, int lane,
                                                 const BitonicSel& sel, int& u, int& left)
{
    static_assert(!SIGNED || sizeof(KT) == 4, "the signed form is built for the bucket plan (32-bit keys)");
    constexpr int S = SIGNED ? 1 : 0;
    const bool have = lane < cnt;
    uint32_t c2s;
    if constexpr (sizeof(KT) == 4) {
        const uint32_t k = have ? ((c2 << 6) | static_cast<uint32_t>(lane)) : ~0u;
        c2s = wave_bitonic32(k, lane, cnt, sel) >> 6;
    } else {
        KT k = have ? static_cast<KT>((static_cast<KT>(c2) << 6) | static_cast<KT>(lane)) : static_cast<KT>(~static_cast<KT>(0));
        c2s = static_cast<uint32_t>(wave_bitonic<KT>(k, lane) >> 6);
    }
    const uint32_t cv = c2s >> (1 + S);
    const uint32_t prev = dpp_mov<0x138>(cv);                     // wave_shr:1 (lane 0 reads 0: it is a head anyway)
    const bool hd = have && (lane == 0 || prev != cv);
    const uint64_t H = __ballot(hd);
    const uint64_t D = __ballot(have && ((c2s >> S) & 1u) != 0);  // entries of the reversed orientation
    u = __popcll(H);
    left = __popcll(__ballot(hd && static_cast<int32_t>(cv) < r));
    // the run of this head ends at the next head (or at cnt); reversed entries inside it = prefix count there - prefix count here
    const uint64_t above = (H >> lane) >> 1;
    const int end = above ? lane + 1 + (__ffsll(static_cast<long long>(above)) - 1) : cnt;
    const int rl_here = static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(D >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(D), 0u)));
    const int rl_end = __builtin_amdgcn_ds_bpermute((end & 63) << 2, rl_here);
    const int n1 = (end < cnt ? rl_end : __popcll(D)) - rl_here;
    const int rank = static_cast<int>(__builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(H >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(H), 0u)));
    int nneg = 0, nrn = 0;                                        // negative entries of the run; negative AND reversed ones
    if constexpr (SIGNED) {
        const uint64_t N = __ballot(have && (c2s & 1u) != 0);
        const int ln_ = end - lane;
        const uint64_t run = (ln_ >= 64 ? ~0ull : ((1ull << (ln_ > 0 ? ln_ : 0)) - 1ull)) << lane;      // lanes [lane, end)
        nneg = __popcll(N & run);
        nrn = __popcll(N & D & run);
    }
    if (hd) {
        const int ln = end - lane;
        // Theta_arg = (forward +) - (forward -) - (reversed +) + (reversed -) = ln - 2 n1 - 2 nneg + 4 nrn
        p.scratch[gbeg + rank] = static_cast<uint64_t>(cv) | (static_cast<uint64_t>(nneg) << 25) | (static_cast<uint64_t>(ln) << 32) |
                                 (static_cast<uint64_t>(ln - 2 * n1 - 2 * nneg + 4 * nrn + 64) << 40) |
                                 (static_cast<uint64_t>(rank) << 48) | (static_cast<uint64_t>(r & (kChunkRows - 1)) << 54);
    } else if (have) {
        p.scratch[gbeg + u + (lane - rank)] = kNoRecord;          // (lane - rank: entries before this one that are no heads)
    }
}

// One row of 65 .. kUnitRowMax entries whose keys sit in LDS (`src`): rank sort -- rank = number of keys that sort before this one
// ((col, dir) order; identical keys are indistinguishable, so ties may fall either way); SORTED KEYS (not merged records) ->
// scratch[gbeg + rank], then the distinct / left-of-diagonal counts from the sorted run.
template <bool SIGNED = false>
__device__ __forceinline__ void unit_merge_long(const UnitArgs& p, const uint32_t* src, int cnt, int32_t r, int64_t gbeg, int lane,
                                                int& u, int& left)
{
    constexpr int S = SIGNED ? 1 : 0;                             // (signed keys: col << 2 | dir << 1 | negative)
    for (int i = lane; i < cnt; i += 64) {
        const uint32_t mine = src[i];
        int rk = 0;
        for (int t = 0; t < cnt; ++t) {
            const uint32_t o = src[t];
            rk += (o < mine || (o == mine && t < i)) ? 1 : 0;
        }
        p.scratch[gbeg + rk] = mine;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // the wavefront re-reads what its lanes wrote
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    u = 0;
    left = 0;
    for (int i0 = 0; i0 < cnt; i0 += 64) {
        const int i = i0 + lane;
        bool hd = false, lt = false;
        if (i < cnt) {
            const uint32_t cur = static_cast<uint32_t>(p.scratch[gbeg + i]) >> (1 + S);
            hd = i == 0 || (static_cast<uint32_t>(p.scratch[gbeg + i - 1]) >> (1 + S)) != cur;
            lt = hd && static_cast<int32_t>(cur) < r;
        }
        u += __popcll(__ballot(hd));
        left += __popcll(__ballot(lt));
    }
}

// Kernel A: one wavefront per kRowsPerWave rows (as row_merge_wave): order each row in registers, park the MERGED row at the head
// of the row's range of the sort's dead input buffer, count the distinct entries and those left of the diagonal.
template <typename KT>
__global__ __launch_bounds__(256) void unit_merge_rows(UnitArgs p)
{
    __shared__ uint32_t lk[4][kUnitRowMax];       // rank-sort staging of a 65+ entry row, one per wavefront
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r0 = (static_cast<int64_t>(blockIdx.x) * 4 + wv) * kRowsPerWave;
    if (r0 >= p.n) return;
    const int rows = p.n - r0 < kRowsPerWave ? static_cast<int>(p.n - r0) : kRowsPerWave;
    int32_t beg[kRowsPerWave], cnt[kRowsPerWave];
    uint32_t c2[kRowsPerWave];
    const BitonicSel sel = bitonic_sel(lane);
    const int32_t bound = p.rs[r0 + (lane <= rows ? lane : rows)];
#pragma unroll
    for (int j = 0; j < kRowsPerWave; ++j) {
        beg[j] = __builtin_amdgcn_readlane(bound, j < rows ? j : rows);
        cnt[j] = j < rows ? __builtin_amdgcn_readlane(bound, j + 1 < rows ? j + 1 : rows) - beg[j] : 0;
    }
#pragma unroll
    for (int j = 0; j < kRowsPerWave; ++j) {                      // unconditional loads from clamped addresses (row_merge_wave)
        int64_t pos = static_cast<int64_t>(beg[j]) + lane;
        pos = pos < p.m ? pos : p.m - 1;
        c2[j] = static_cast<uint32_t>(__builtin_nontemporal_load(p.keys + pos));
    }
#pragma unroll
    for (int j = 0; j < kRowsPerWave; ++j) {
        if (j >= rows) continue;
        const int32_t r = static_cast<int32_t>(r0) + j;
        int u = 0, left = 0;
        if (cnt[j] <= 64) {
            unit_merge_short<KT>(p, c2[j], cnt[j], r, beg[j], lane, sel, u, left);
        } else if (cnt[j] <= kUnitRowMax) {
            for (int i = lane; i < cnt[j]; i += 64) lk[wv][i] = static_cast<uint32_t>(p.keys[beg[j] + i]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            unit_merge_long(p, lk[wv], cnt[j], r, beg[j], lane, u, left);
        } else {
            if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(p.info + 1), 1ull);    // host: two-stage pipeline
        }
        if (lane == 0) {
            p.row_u[r] = u;
            p.row_left[r] = left;
        }
    }
}

// A row written straight to global memory by one wavefront (chunks holding a row of more than 64 entries): merged records for rows
// of <= 64 entries, the SORTED KEYS unit_merge_long left for longer ones.
template <bool SIGNED = false>
__device__ __forceinline__ void unit_write_row_direct(const UnitArgs& p, int32_t r, int32_t beg, int cnt, int u, int left, int64_t slot0,
                                                      int lane, float cs1, float sn1)
{
    constexpr int S = SIGNED ? 1 : 0;
    if (lane == 0) unit_diagonal<0>(p, nullptr, 0, r, slot0 + left);
    const float ir = p.sym ? p.lut[p.cnt16[r]] : 0.f;
    if (cnt <= 64) {
        if (lane < u) {
            const uint64_t rc = p.scratch[beg + lane];
            const int32_t c = static_cast<int32_t>(rc & (SIGNED ? 0x1FFFFFFull : 0xFFFFFFFFull));
            const int nneg = SIGNED ? static_cast<int>((rc >> 25) & 0x7Full) : 0;
            const int ln = static_cast<int>((rc >> 32) & 0xFFull), th = static_cast<int>((rc >> 40) & 0xFFull) - 64;
            unit_values<0>(p, nullptr, 0, ir, p.sym ? p.lut[p.cnt16[c]] : 0.f, c, ln - 2 * nneg, th, cs1, sn1,
                           slot0 + lane + (c > r ? 1 : 0));
        }
    } else if (cnt <= kUnitRowMax) {
        int base_rank = 0;
        for (int i0 = 0; i0 < cnt; i0 += 64) {
            const int i = i0 + lane;
            bool hd = false;
            uint32_t cur = 0;
            if (i < cnt) {
                cur = static_cast<uint32_t>(p.scratch[beg + i]) >> (1 + S);
                hd = i == 0 || (static_cast<uint32_t>(p.scratch[beg + i - 1]) >> (1 + S)) != cur;
            }
            const uint64_t H = __ballot(hd);
            if (hd) {
                int ln = 0, n1 = 0, nneg = 0, nrn = 0;
                for (int t = i; t < cnt; ++t) {                    // the run: short (multiplicity of one neighbour)
                    const uint32_t k2 = static_cast<uint32_t>(p.scratch[beg + t]);
                    if ((k2 >> (1 + S)) != cur) break;
                    ++ln;
                    const int rv = static_cast<int>((k2 >> S) & 1u), ng = SIGNED ? static_cast<int>(k2 & 1u) : 0;
                    n1 += rv;
                    nneg += ng;
                    nrn += rv & ng;
                }
                const int rk = base_rank + __popcll(H & ((1ull << lane) - 1ull));
                const int32_t c = static_cast<int32_t>(cur);
                unit_values<0>(p, nullptr, 0, ir, p.sym ? p.lut[p.cnt16[c]] : 0.f, c, ln - 2 * nneg, ln - 2 * n1 - 2 * nneg + 4 * nrn, cs1, sn1,
                               slot0 + rk + (c > r ? 1 : 0));
            }
            base_rank += __popcll(H);
        }
    }
}

// Kernel B (after the scan of distinct + 1 -> row pointer): one workgroup per chunk of kChunkRows rows walks the chunk's positions of
// the record buffer -- one coalesced 8-byte load per position, every lane busy whatever the rows' lengths (a record names its row and
// its rank in it) -- forms the values and stages columns, the four value arrays and the diagonals at their final slots in LDS; the
// chunk's contiguous slot range leaves with aligned 16-byte stores.  (Before: a wavefront per four rows, a row per pass: 131 vector
// instructions per row with 41 of 64 lanes busy -- VALU-bound at half its time; now 65.)
// Measured and dropped: persistent workgroups running  row bounds -> records -> deg^-1/2 gathers -> stores  as a pipeline over
// chunks c, c + G, c + 2 G (0.37 ms either way: on gfx9 a wait for the loads of the next round also waits for the drain's stores,
// which share their counter -- the same reason a grid-stride copy runs at 4.5 TB/s here and a block-per-piece copy at 6.2).
template <int THREADS, int SLOTS, bool SIGNED = false>
__global__ __launch_bounds__(THREADS) void unit_write_chunks(UnitArgs p)
{
    __shared__ __attribute__((aligned(16))) float stage[5 * SLOTS];
    __shared__ int32_t s_rs[kChunkRows + 1], s_rp[kChunkRows + 1], s_left[kChunkRows];
    __shared__ float s_lut[kUnitRowMax + 1];
    __shared__ int32_t s_cnt[kChunkRows];                          // the rows' entry counts: their own deg^-1/2 is s_lut[s_cnt[j]]
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int64_t r0 = static_cast<int64_t>(blockIdx.x) * kChunkRows;
    const int rows = p.n - r0 < kChunkRows ? static_cast<int>(p.n - r0) : kChunkRows;
    // the chunk's positions of the record buffer: two wavefront-uniform (scalar) loads, so the records can leave before the
    // per-row bounds below are back
    const int32_t beg = p.rs[r0], end = p.rs[r0 + rows];
    bool is_long = false;
    if (t <= rows) {
        s_rs[t] = p.rs[r0 + t];
        s_rp[t] = p.rowptr[r0 + t];                                // (rowptr[n] exists)
    }
    if (t < rows) {
        s_left[t] = p.row_left[r0 + t];
        const int32_t cn = p.sym ? p.cnt16[r0 + t] : 0;
        s_cnt[t] = cn < kUnitRowMax ? cn : kUnitRowMax;
        is_long = p.rs[r0 + t + 1] - p.rs[r0 + t] > 64;
        if (t == 0) is_long = is_long || p.rowptr[r0 + rows] - p.rowptr[r0] > SLOTS - 4;      // more slots than the staging holds
    }
    if (p.info[1] != 0) return;                                    // a row this pipeline does not take: outputs are discarded
    if (p.sym)
        for (int k = t; k <= kUnitRowMax; k += THREADS) s_lut[k] = p.lut[k];
    constexpr int PER = kChunkRows * 64 / THREADS;                     // rows of <= 64 entries: <= PER positions per thread
    uint64_t rc[PER];
    float ic[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {                               // every load of the chunk in flight before the first is used
        const int32_t i = beg + t + k * THREADS;
        rc[k] = i < end ? __builtin_nontemporal_load(p.scratch + i) : kNoRecord;
    }
    const bool any_long = __syncthreads_or(is_long) != 0;
    const float cs1 = p.trig[0], sn1 = p.trig[1];
    if (any_long) {
        for (int j = __builtin_amdgcn_readfirstlane(wv); j < rows; j += THREADS / 64)
            unit_write_row_direct<SIGNED>(p, static_cast<int32_t>(r0) + j, s_rs[j], s_rs[j + 1] - s_rs[j], s_rp[j + 1] - s_rp[j] - 1, s_left[j],
                                          s_rp[j], lane, cs1, sn1);
        return;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {                               // the 2-byte gathers leave together; the LDS look-ups follow
        const uint16_t cn = (p.sym && rc[k] != kNoRecord) ? p.cnt16[rc[k] & (SIGNED ? 0x1FFFFFFull : 0xFFFFFFFFull)] : uint16_t(0);
        ic[k] = __uint_as_float(cn);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) ic[k] = p.sym ? s_lut[__float_as_uint(ic[k]) < kUnitRowMax ? __float_as_uint(ic[k]) : kUnitRowMax] : 0.f;
    const int64_t wslot0 = s_rp[0];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (rc[k] == kNoRecord) continue;
        const int32_t c = static_cast<int32_t>(rc[k] & (SIGNED ? 0x1FFFFFFull : 0xFFFFFFFFull));
        const int nneg = SIGNED ? static_cast<int>((rc[k] >> 25) & 0x7Full) : 0;
        const int ln = static_cast<int>((rc[k] >> 32) & 0xFFull), th = static_cast<int>((rc[k] >> 40) & 0xFFull) - 64;
        const int rank = static_cast<int>((rc[k] >> 48) & 0x3Full), j = static_cast<int>((rc[k] >> 54) & (kChunkRows - 1));
        const int32_t r = static_cast<int32_t>(r0) + j;
        unit_values<SLOTS>(p, stage, wslot0, p.sym ? s_lut[s_cnt[j]] : 0.f, ic[k], c, ln - 2 * nneg, th, cs1, sn1,
                                 static_cast<int64_t>(s_rp[j]) + rank + (c > r ? 1 : 0));
    }
    if (t < rows) unit_diagonal<SLOTS>(p, stage, wslot0, static_cast<int32_t>(r0) + t, static_cast<int64_t>(s_rp[t]) + s_left[t]);
    __syncthreads();
    // drain [wslot0, hi): the 16-byte aligned middle as dwordx4, the ragged ends (<= 6 slots) as dwords
    const int64_t lo = wslot0, hi = s_rp[rows], a0 = (lo + 3) & ~int64_t(3), a1 = hi & ~int64_t(3);
    float* const outs[5] = {reinterpret_cast<float*>(p.ccol), p.vb_re, p.vb_im, p.vf_re, p.vf_im};
    if (a0 < a1) {
        const int quads = static_cast<int>((a1 - a0) >> 2);
#pragma unroll
        for (int arr = 0; arr < 5; ++arr) {
            for (int g = t; g < quads; g += THREADS) {
                const int o = static_cast<int>(a0 - lo) + 4 * g;
                const float* src = stage + arr * SLOTS + o;
                const f32x4 v = {src[0], src[1], src[2], src[3]};
                __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(outs[arr] + a0) + g);
            }
        }
        const int head_n = static_cast<int>(a0 - lo), ends = head_n + static_cast<int>(hi - a1);
        if (t < 5 * 8) {
            const int arr = t >> 3, e = t & 7;
            if (e < ends) {
                const int64_t sl = e < head_n ? lo + e : a1 + (e - head_n);
                outs[arr][sl] = stage[arr * SLOTS + static_cast<int>(sl - lo)];
            }
        }
    } else {
#pragma unroll
        for (int arr = 0; arr < 5; ++arr)
            for (int64_t sl = lo + t; sl < hi; sl += THREADS) outs[arr][sl] = stage[arr * SLOTS + static_cast<int>(sl - lo)];
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 4, second form of the unweighted build: NO global sort.  The radix sort above moves 8-byte keys twice and a histogram
// pass on top (0.70 ms of rocPRIM kernels at the north star) only to bucket the stream by row; the row-bucketed stream is then
// read back twice more (row bounds, merge).  Here the stream is split ONCE, into buckets of 2^rl consecutive rows that fit a
// workgroup's LDS, and everything finer happens inside LDS:
//   1  bucket_count        per tile of the edge list (one workgroup, 16 k edges): entries per bucket, counted in LDS ->
//      hist[bucket][tile]
//   -  exclusive scan of hist (bucket-major): every (bucket, tile) pair owns a private range of the stream -- no global atomics,
//      no look-back; the order inside a bucket is whatever the LDS atomics gave, which is immaterial: unit weights make a row's
//      merge order-free, and the rows are ordered by column below
//   2  bucket_scatter      the same tiles again: 4-byte entries  row_low << (cbits + 1) | col << 1 | dir  to their ranges,
//      staged by bucket in LDS so that a tile leaves as runs of neighbouring words
//   3  bucket_merge_rows   one workgroup per bucket: the bucket is counted by row in LDS (-> row bounds, degrees, deg^-1/2),
//      placed by row in LDS, and each row is ordered and merged by a wavefront exactly as unit_merge_rows does (same records, same
//      place: scratch[row start + rank]); unit_write_chunks follows unchanged.
// Traffic at the north star: 2 x 320 MB of edge list in, 160 MB out and in, against 320 + 320 (keys) + 320 (histogram) +
// 2 x 640 (two sort passes) + 320 (row bounds) + 320 (merge): 0.55 ms for the three kernels against 1.19 ms (profiles/
// r4w_build_kernel_stats.csv).  A bucket that does not fit (32 k entries) or a row above
// kUnitRowMax is counted in info[1]: the host takes the two-stage pipeline, as before.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kMaxBuckets = 3000;         // LDS of bucket_scatter: 32 k staged entries + the head map + two tables of this many words
constexpr int kPassThreads = 512;
constexpr int kPassBatch = 4;
constexpr int kTileEdges = 16384;         // edges per tile (= per workgroup of the two passes): 32 k entries staged in LDS
constexpr int kScatterThreads = 1024;

struct BucketPlan {
    int rl;            // log2(rows per bucket)
    int nb;            // buckets
    int g;             // tiles of the edge list (= workgroups of the two passes)
    int cbits;         // bits of a column id
    int cap;           // entries a bucket may hold (LDS of bucket_merge_rows)
    int threads;       // workgroup size of bucket_merge_rows
    int sb;            // sign bits per entry: 0 (unit weights), 1 (weights of +-1: pygsd_magop_unit_signed)
};

// false: this graph is not taken by the bucket form (ids too wide for a 4-byte entry, too many buckets, rows too dense)
inline bool bucket_plan(int64_t e, int32_t n, int sb, BucketPlan* pl)
{
    if (n <= 0 || e <= 0) return false;
    pl->sb = sb;
    const int64_t m = 2 * e;
    pl->threads = 1024;
    pl->cap = pl->threads * 32;                                  // 128 KB of LDS: one workgroup per CU
    pl->cbits = bits_for(static_cast<uint64_t>(n > 1 ? n - 1 : 1));
    int rl = 10;
    while (rl > 3 && (static_cast<int64_t>(n) >> rl) < 1024) --rl;                        // enough buckets to fill the chip
    // average bucket <= 3/4 of the LDS (24 k entries; the buckets of a graph without hubs scatter by a few hundred around it)
    while (rl > 3 && (m << rl) / n > static_cast<int64_t>(pl->cap) * 3 / 4) --rl;
    if ((m << rl) / n > static_cast<int64_t>(pl->cap) * 3 / 4) return false;
    const int64_t nb = (static_cast<int64_t>(n) + (int64_t(1) << rl) - 1) >> rl;
    // (25: the in-register sort key, col << 7 | dir << 6 | lane; one bit less for the column with a sign bit below dir)
    if (nb > kMaxBuckets || rl + pl->cbits + 1 + sb > 32 || pl->cbits + sb > 25) return false;
    const int64_t g = (e + kTileEdges - 1) / kTileEdges;
    if (g * nb + 1 > (int64_t(1) << 23)) return false;          // hist / off: <= 32 MB each
    pl->rl = rl;
    pl->nb = static_cast<int>(nb);
    pl->g = static_cast<int>(g);
    return true;
}

// Pass 1: a tile's entries per bucket, counted in LDS -> hist[bucket][tile]; node-id range check (info[2], info[3]) folded in.
// w != NULL (pygsd_magop_unit_signed): the weights of the tile's non-loop edges are validated on the way -- every one must be
// exactly +1 or -1 (-1 only with allow_neg: the degree convention that counts |w|); anything else counts in info[1], the later
// kernels of the build return at once and the host takes the two-stage pipeline.
__global__ __launch_bounds__(kPassThreads) void bucket_count(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                                                             const float* __restrict__ w, int32_t allow_neg, int64_t e,
                                                             int32_t n, BucketPlan pl, int32_t* __restrict__ hist,
                                                             int64_t* __restrict__ info, float two_pi_q, float* __restrict__ trig,
                                                             float* __restrict__ lut)
{
    __shared__ uint32_t cnt[kMaxBuckets];
    const int wg = blockIdx.x, t = threadIdx.x;
    bool odd_weight = false;
    if (wg == 0) {
        if (t == 0) sincosf(two_pi_q, trig + 1, trig);
        for (int k = t; k <= kUnitRowMax; k += kPassThreads) unit_lut(k, lut);
    }
    for (int b = t; b < pl.nb; b += kPassThreads) cnt[b] = 0u;
    __syncthreads();
    const int64_t lo = static_cast<int64_t>(wg) * kTileEdges, hi = lo + kTileEdges < e ? lo + kTileEdges : e;
    const uint64_t nn = static_cast<uint64_t>(n);
    for (int64_t k0 = lo; k0 < hi; k0 += kPassThreads * kPassBatch) {
        int64_t r[kPassBatch], c[kPassBatch];
        float wv[kPassBatch];
#pragma unroll
        for (int u = 0; u < kPassBatch; ++u) {
            const int64_t k = k0 + u * kPassThreads + t;
            const bool ok = k < hi;
            r[u] = ok ? row[k] : 0;                               // (0, 0): a self loop, dropped below
            c[u] = ok ? col[k] : 0;
            wv[u] = (w && ok) ? w[k] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < kPassBatch; ++u) {
            const bool bad_r = static_cast<uint64_t>(r[u]) >= nn, bad_c = static_cast<uint64_t>(c[u]) >= nn;
            if (bad_r || bad_c) {
                info[2] = 1;                                       // racing writers all store valid witnesses
                info[3] = bad_r ? r[u] : c[u];
            } else if (r[u] != c[u]) {
                atomicAdd(&cnt[static_cast<uint32_t>(r[u]) >> pl.rl], 1u);
                atomicAdd(&cnt[static_cast<uint32_t>(c[u]) >> pl.rl], 1u);
                // (a NaN fails the first test)
                if (!(fabsf(wv[u]) == 1.0f) || (wv[u] < 0.f && !allow_neg)) odd_weight = true;
            }
        }
    }
    if (w) {                                                       // (uniform: every thread of the grid takes the same side)
        if (__syncthreads_or(odd_weight) && t == 0) atomicAdd(reinterpret_cast<unsigned long long*>(info + 1), 1ull);
    }
    __syncthreads();
    for (int b = t; b < pl.nb; b += kPassThreads) hist[static_cast<int64_t>(b) * pl.g + wg] = static_cast<int32_t>(cnt[b]);
    if (wg == 0 && t == 0) hist[static_cast<int64_t>(pl.nb) * pl.g] = 0;            // the scan's last input: off[nb * g] = entries
}

// Pass 2: the tile again.  Its (bucket, tile) counts are re-read from the scanned table (adjacent words), scanned locally, and the
// entries are placed BY BUCKET in LDS first; the tile then leaves as runs of consecutive words per bucket (lanes write neighbouring
// addresses) -- 4-byte stores straight to the buckets ran at 75 G requests/s (0.59 ms for this pass), whatever the tiling.
// Which bucket a staged slot belongs to is read off a bit map of the buckets' first slots: {32 head bits, heads before this word}
// per 32 slots -> index among the tile's non-empty buckets -> that bucket's (global - staged) offset.  (A binary search over the
// buckets' first slots instead cost 11 dependent LDS reads and ~75 VALU instructions per 64 entries: 0.20 ms for the pass.)
// LDS: stage[2 * kTileEdges], cur[nb] (placement cursors), gdc[nb] (offsets of the non-empty buckets, compacted), hp[1024].
// SIGNED (weights of +-1, validated by bucket_count): the entry carries the sign below its direction bit,
//   row_low << (cbits + 2) | col << 2 | dir << 1 | (w < 0)  -- both orientations of an edge carry the edge's sign.
template <bool SIGNED>
__global__ __launch_bounds__(kScatterThreads) void bucket_scatter(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                                                                  const float* __restrict__ w, int64_t e,
                                                                  int32_t n, BucketPlan pl, const int32_t* __restrict__ off,
                                                                  uint32_t* __restrict__ stream, const int64_t* __restrict__ info)
{
    static_assert(2 * kTileEdges == 32 * kScatterThreads, "one 32-slot word of the head map per thread");
    constexpr int S = SIGNED ? 1 : 0;
    if (SIGNED && info[1] != 0) return;                           // a weight that is not +-1: the host takes the two-stage pipeline
    extern __shared__ __attribute__((aligned(8))) uint32_t scatter_lds[];
    uint32_t* stage = scatter_lds;
    uint2* hp = reinterpret_cast<uint2*>(stage + 2 * kTileEdges);
    uint32_t* cur = reinterpret_cast<uint32_t*>(hp + kScatterThreads);
    uint32_t* gdc = cur + pl.nb;
    __shared__ uint32_t wsum[kScatterThreads / 64], wsum2[kScatterThreads / 64], total_s;
    const int st = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    constexpr int PER = (kMaxBuckets + kScatterThreads - 1) / kScatterThreads;
    uint32_t cnts[PER], goff[PER], mine = 0;
    hp[t] = make_uint2(0u, 0u);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int b = t * PER + j;
        cnts[j] = 0;
        goff[j] = 0;
        if (b < pl.nb) {
            const int32_t* o = off + static_cast<int64_t>(b) * pl.g + st;
            goff[j] = static_cast<uint32_t>(o[0]);
            cnts[j] = static_cast<uint32_t>(o[1]) - goff[j];
        }
        mine += cnts[j] | (cnts[j] ? 0x10000u : 0u);             // entries (<= 2^15) | non-empty buckets << 16
    }
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t run = inc - mine;
    for (int w = 0; w < wv; ++w) run += wsum[w];
    uint32_t slot = run & 0xFFFFu, nz = run >> 16;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int b = t * PER + j;
        if (b < pl.nb) {
            cur[b] = slot;
            if (cnts[j]) {
                atomicOr(&hp[slot >> 5].x, 1u << (slot & 31u));
                gdc[nz++] = goff[j] - slot;
            }
        }
        slot += cnts[j];
    }
    if (t == kScatterThreads - 1) total_s = slot;                 // the tile's entries
    __syncthreads();
    {
        const uint32_t pc = __popc(hp[t].x);
        uint32_t pinc = pc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(pinc, d);
            if (lane >= d) pinc += o;
        }
        if (lane == 63) wsum2[wv] = pinc;
        __syncthreads();
        uint32_t before = pinc - pc;
        for (int w = 0; w < wv; ++w) before += wsum2[w];
        hp[t].y = before;
    }
    const int64_t lo = static_cast<int64_t>(st) * kTileEdges, hi = lo + kTileEdges < e ? lo + kTileEdges : e;
    const uint64_t nn = static_cast<uint64_t>(n);
    const uint32_t rmask = (1u << pl.rl) - 1u;
    const int sh = pl.cbits + 1 + S;
    int64_t r[kPassBatch], c[kPassBatch], rn[kPassBatch], cn[kPassBatch];
    float ww[kPassBatch], wn[kPassBatch];
#pragma unroll
    for (int u = 0; u < kPassBatch; ++u) {
        const int64_t k = lo + u * kScatterThreads + t;
        const bool ok = k < hi;
        rn[u] = ok ? row[k] : 0;
        cn[u] = ok ? col[k] : 0;
        wn[u] = (SIGNED && ok) ? w[k] : 1.f;
    }
    for (int64_t k0 = lo; k0 < hi; k0 += kScatterThreads * kPassBatch) {
#pragma unroll
        for (int u = 0; u < kPassBatch; ++u) {
            r[u] = rn[u];
            c[u] = cn[u];
            ww[u] = wn[u];
            const int64_t k = k0 + (kPassBatch + u) * kScatterThreads + t;      // the next batch is in flight during this one's placement
            const bool ok = k < hi;
            rn[u] = ok ? row[k] : 0;
            cn[u] = ok ? col[k] : 0;
            wn[u] = (SIGNED && ok) ? w[k] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < kPassBatch; ++u) {
            if (static_cast<uint64_t>(r[u]) >= nn || static_cast<uint64_t>(c[u]) >= nn || r[u] == c[u]) continue;
            const uint32_t rr = static_cast<uint32_t>(r[u]), cc = static_cast<uint32_t>(c[u]);
            const uint32_t neg = (SIGNED && ww[u] < 0.f) ? 1u : 0u;
            stage[atomicAdd(&cur[rr >> pl.rl], 1u)] = ((rr & rmask) << sh) | (cc << (1 + S)) | neg;
            stage[atomicAdd(&cur[cc >> pl.rl], 1u)] = ((cc & rmask) << sh) | (rr << (1 + S)) | (1u << S) | neg;
        }
    }
    __syncthreads();
    const uint32_t total = total_s;
    for (uint32_t q = t; q < total; q += kScatterThreads) {
        const uint2 w = hp[q >> 5];
        const uint32_t idx = __popc(w.x & ((2u << (q & 31u)) - 1u)) + w.y - 1u;
        stream[gdc[idx] + q] = stage[q];
    }
}

// One workgroup per bucket of 2^rl rows.  LDS: placed[cap] (the bucket's keys, row by row), rcnt[2^rl + 1] (row counts, then
// placement cursors), roff[2^rl + 1] (row offsets inside the bucket).
template <int THREADS, bool SIGNED = false>
__global__ __launch_bounds__(THREADS) void bucket_merge_rows(UnitArgs p, BucketPlan pl, const uint32_t* __restrict__ stream,
                                                             const int32_t* __restrict__ off)
{
    constexpr int EPT = 32, WAVES = THREADS / 64, S = SIGNED ? 1 : 0;
    if (SIGNED && p.info[1] != 0) return;                          // a weight that is not +-1 (bucket_count): nothing to build
    extern __shared__ uint32_t bucket_lds[];
    uint32_t* placed = bucket_lds;
    uint32_t* rcnt = bucket_lds + pl.cap;
    uint32_t* roff = rcnt + (1 << pl.rl) + 8;
    __shared__ uint32_t wsum[WAVES];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int nrow = 1 << pl.rl;
    const int32_t row0 = b << pl.rl;
    const int32_t b0 = off[static_cast<int64_t>(b) * pl.g], b1 = off[static_cast<int64_t>(b + 1) * pl.g];
    const int cnt_b = b1 - b0;
    if (b == pl.nb - 1 && t == 0) {
        const_cast<int32_t*>(p.rs)[p.n] = b1;
        p.row_u[p.n] = 0;                                         // the scan's (n + 1)-th input
    }
    if (cnt_b > pl.cap) {                                          // host: two-stage pipeline
        if (t == 0) atomicAdd(reinterpret_cast<unsigned long long*>(p.info + 1), 1ull);
        return;
    }
    for (int i = t; i <= nrow; i += THREADS) rcnt[i] = 0;
    uint32_t ent[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int i = t + k * THREADS;
        ent[k] = i < cnt_b ? __builtin_nontemporal_load(stream + b0 + i) : 0u;
    }
    __syncthreads();
    const int sh = pl.cbits + 1 + S;
    const uint32_t kmask = (1u << sh) - 1u;
#pragma unroll
    for (int k = 0; k < EPT; ++k)
        if (t + k * THREADS < cnt_b) atomicAdd(&rcnt[ent[k] >> sh], 1u);
    __syncthreads();
    // exclusive scan of the row counts (nrow <= 512 <= THREADS): wavefront scans + the wavefronts' sums
    uint32_t mine = t < nrow ? rcnt[t] : 0u, inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    const uint32_t excl = base + inc - mine;
    if (t < nrow) {
        roff[t] = excl;
        rcnt[t] = excl;                                            // placement cursor
        const int32_t r = row0 + t;
        if (r < p.n) {
            const_cast<int32_t*>(p.rs)[r] = b0 + static_cast<int32_t>(excl);
            const_cast<float*>(p.deg)[r] = static_cast<float>(mine) / 2.f;         // (unit_row_tables)
            const_cast<uint16_t*>(p.cnt16)[r] = clamp16(static_cast<int32_t>(mine));
        }
    }
    if (t == 0) roff[nrow] = static_cast<uint32_t>(cnt_b);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < EPT; ++k)
        if (t + k * THREADS < cnt_b) placed[atomicAdd(&rcnt[ent[k] >> sh], 1u)] = ent[k] & kmask;
    __syncthreads();
    const BitonicSel sel = bitonic_sel(lane);
    const int wvu = __builtin_amdgcn_readfirstlane(wv);          // (the compiler does not know t >> 6 is wavefront-uniform)
    for (int rr = wvu; rr < nrow; rr += WAVES) {
        const int32_t r = row0 + rr;
        if (r >= p.n) break;
        const int beg = __builtin_amdgcn_readfirstlane(static_cast<int>(roff[rr]));
        const int cnt = __builtin_amdgcn_readfirstlane(static_cast<int>(roff[rr + 1])) - beg;
        int u = 0, left = 0;
        if (cnt <= 64) {
            const uint32_t c2 = placed[beg + (lane < cnt ? lane : 0)];
            unit_merge_short<uint32_t, SIGNED>(p, c2, cnt, r, static_cast<int64_t>(b0) + beg, lane, sel, u, left);
        } else if (cnt <= kUnitRowMax) {
            unit_merge_long<SIGNED>(p, placed + beg, cnt, r, static_cast<int64_t>(b0) + beg, lane, u, left);
        } else {
            if (lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(p.info + 1), 1ull);
        }
        if (lane == 0) {
            p.row_u[r] = u;
            p.row_left[r] = left;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 5: the bucket form for REAL-VALUED weights (weighted MagNetConv graphs, MSConv with rated signs): the front of the two-stage
// pipeline -- edge_keys, the radix sort of (key, weight) pairs on the row bits, key_row_starts, row_merge_wave / _block: 1.2 of its
// 2.1 ms at the north star -- replaced by the bucket split, with the weight of an entry travelling in a second 4-byte stream at the
// same index.  What the stages behind it read (one 16-byte record per stream position, row starts, distinct counts, degrees) is
// produced in the same form, so row_tables / values_entries / diagonal_of_empty_rows run unchanged.
// Order.  The reference adds the duplicates of an entry in (direction, list position) order and the row degree sequentially in
// column order.  Buckets and rows are filled by LDS atomics -- no order at all -- so this form only takes what cannot show it:
// a neighbour's run of ONE or TWO entries (a reciprocal pair, or an edge listed twice: fp32 addition commutes); the degree sum runs
// over the column-sorted distinct entries and is order-independent by construction.  A run of three or more is counted in info[1]
// and the host repeats the build behind the radix sort (pygsd_magop_stage1_sorted), as it does for the rows these kernels do not
// take (more than 512 entries) and for a half bucket that overflows its share of the LDS.
// ------------------------------------------------------------------------------------------------------------------------
constexpr int kRoundsW = kTileEdges / (kScatterThreads * kPassBatch);            // batches of one tile per thread: 4

// bucket_scatter with a weight stream: the tile's keys are staged and drained exactly as above; every thread remembers the two
// staged slots of each of its 16 edges (15 bits each) and, once the keys have left, stages the edges' weights at the same slots and
// drains them to `wstream` through the same slot -> bucket map.
__global__ __launch_bounds__(kScatterThreads) void bucket_scatter_w(const int64_t* __restrict__ row, const int64_t* __restrict__ col,
                                                                    const float* __restrict__ w, int64_t e, int32_t n, BucketPlan pl,
                                                                    const int32_t* __restrict__ off, uint32_t* __restrict__ stream,
                                                                    float* __restrict__ wstream)
{
    static_assert(2 * kTileEdges == 32 * kScatterThreads, "one 32-slot word of the head map per thread");
    static_assert(kRoundsW * kScatterThreads * kPassBatch == kTileEdges, "a tile is a whole number of batches");
    extern __shared__ __attribute__((aligned(8))) uint32_t scatter_lds[];
    uint32_t* stage = scatter_lds;
    uint2* hp = reinterpret_cast<uint2*>(stage + 2 * kTileEdges);
    uint32_t* cur = reinterpret_cast<uint32_t*>(hp + kScatterThreads);
    uint32_t* gdc = cur + pl.nb;
    __shared__ uint32_t wsum[kScatterThreads / 64], wsum2[kScatterThreads / 64], total_s;
    const int st = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    constexpr int PER = (kMaxBuckets + kScatterThreads - 1) / kScatterThreads;
    uint32_t cnts[PER], goff[PER], mine = 0;
    hp[t] = make_uint2(0u, 0u);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int b = t * PER + j;
        cnts[j] = 0;
        goff[j] = 0;
        if (b < pl.nb) {
            const int32_t* o = off + static_cast<int64_t>(b) * pl.g + st;
            goff[j] = static_cast<uint32_t>(o[0]);
            cnts[j] = static_cast<uint32_t>(o[1]) - goff[j];
        }
        mine += cnts[j] | (cnts[j] ? 0x10000u : 0u);
    }
    uint32_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    uint32_t run = inc - mine;
    for (int q = 0; q < wv; ++q) run += wsum[q];
    uint32_t slot = run & 0xFFFFu, nz = run >> 16;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int b = t * PER + j;
        if (b < pl.nb) {
            cur[b] = slot;
            if (cnts[j]) {
                atomicOr(&hp[slot >> 5].x, 1u << (slot & 31u));
                gdc[nz++] = goff[j] - slot;
            }
        }
        slot += cnts[j];
    }
    if (t == kScatterThreads - 1) total_s = slot;
    __syncthreads();
    {
        const uint32_t pc = __popc(hp[t].x);
        uint32_t pinc = pc;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(pinc, d);
            if (lane >= d) pinc += o;
        }
        if (lane == 63) wsum2[wv] = pinc;
        __syncthreads();
        uint32_t before = pinc - pc;
        for (int q = 0; q < wv; ++q) before += wsum2[q];
        hp[t].y = before;
    }
    const int64_t lo = static_cast<int64_t>(st) * kTileEdges, hi = lo + kTileEdges < e ? lo + kTileEdges : e;
    const uint64_t nn = static_cast<uint64_t>(n);
    const uint32_t rmask = (1u << pl.rl) - 1u;
    const int sh = pl.cbits + 1;
    uint32_t slots[kRoundsW * kPassBatch];                         // staged slots of this thread's edges: forward | reversed << 16
#pragma unroll
    for (int it = 0; it < kRoundsW; ++it) {
        int64_t r[kPassBatch], c[kPassBatch];
#pragma unroll
        for (int u = 0; u < kPassBatch; ++u) {
            const int64_t k = lo + (static_cast<int64_t>(it) * kPassBatch + u) * kScatterThreads + t;
            const bool ok = k < hi;
            r[u] = ok ? row[k] : 0;                               // (0, 0): a self loop, dropped below
            c[u] = ok ? col[k] : 0;
        }
#pragma unroll
        for (int u = 0; u < kPassBatch; ++u) {
            uint32_t sl = 0xFFFFFFFFu;
            if (!(static_cast<uint64_t>(r[u]) >= nn || static_cast<uint64_t>(c[u]) >= nn || r[u] == c[u])) {
                const uint32_t rr = static_cast<uint32_t>(r[u]), cc = static_cast<uint32_t>(c[u]);
                const uint32_t sf = atomicAdd(&cur[rr >> pl.rl], 1u), sr = atomicAdd(&cur[cc >> pl.rl], 1u);
                stage[sf] = ((rr & rmask) << sh) | (cc << 1);
                stage[sr] = ((cc & rmask) << sh) | (rr << 1) | 1u;
                sl = sf | (sr << 16);
            }
            slots[it * kPassBatch + u] = sl;
        }
    }
    __syncthreads();
    const uint32_t total = total_s;
    for (uint32_t q = t; q < total; q += kScatterThreads) {
        const uint2 wq = hp[q >> 5];
        const uint32_t idx = __popc(wq.x & ((2u << (q & 31u)) - 1u)) + wq.y - 1u;
        stream[gdc[idx] + q] = stage[q];
    }
    __syncthreads();                                              // the keys have left: the staging area takes the weights
    float* stage_w = reinterpret_cast<float*>(stage);
#pragma unroll
    for (int it = 0; it < kRoundsW; ++it) {
        float wv4[kPassBatch];
#pragma unroll
        for (int u = 0; u < kPassBatch; ++u) {
            const int64_t k = lo + (static_cast<int64_t>(it) * kPassBatch + u) * kScatterThreads + t;
            wv4[u] = k < hi ? w[k] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kPassBatch; ++u) {
            const uint32_t sl = slots[it * kPassBatch + u];
            if (sl != 0xFFFFFFFFu) {
                stage_w[sl & 0xFFFFu] = wv4[u];
                stage_w[sl >> 16] = wv4[u];
            }
        }
    }
    __syncthreads();
    for (uint32_t q = t; q < total; q += kScatterThreads) {
        const uint2 wq = hp[q >> 5];
        const uint32_t idx = __popc(wq.x & ((2u << (q & 31u)) - 1u)) + wq.y - 1u;
        wstream[gdc[idx] + q] = stage_w[q];
    }
}

// One workgroup per bucket, TWO rounds over its rows (rows 0 .. 2^(rl-1) - 1, then the rest): keys and weights of a round's rows are
// placed row by row in LDS (8 bytes per entry: half a bucket per round) and leave as ONE coalesced copy per stream -- the row-grouped
// (key, weight) streams the radix sort used to produce, minus the order inside a row.  Merging is NOT done here: a weighted row is a
// chain of dependent steps (in-register sort, weight hand-off, run sums, a sequential degree sum over ~40 terms), and this kernel --
// 1024 threads around 130 KB of LDS -- runs four wavefronts per SIMD; its first version merged in place and took 0.85 ms where the
// unit merge takes 0.30.  The rows go to the sorted pipeline's row_merge_wave / row_merge_block instead: 41 VGPRs, no LDS to speak
// of, as many wavefronts per SIMD as the chip holds.  Stream positions: the bucket's range, round 0's rows first.
// LDS: pk[cap / 2], pw[cap / 2], rcnt[2^(rl-1) + 8], roff[2^(rl-1) + 8].
template <int THREADS>
__global__ __launch_bounds__(THREADS) void bucket_place_rows_w(BucketPlan pl, const uint32_t* __restrict__ stream,
                                                               const float* __restrict__ wstream, const int32_t* __restrict__ off,
                                                               int32_t n, int32_t* __restrict__ rs, int32_t* __restrict__ ucnt,
                                                               uint32_t* __restrict__ keys2, float* __restrict__ w2,
                                                               int64_t* __restrict__ info)
{
    constexpr int WAVES = THREADS / 64;
    extern __shared__ uint32_t bucket_lds[];
    const int half_cap = pl.cap / 2, hrow = (1 << pl.rl) >> 1;    // (rl >= 3: a bucket has at least 8 rows)
    uint32_t* pk = bucket_lds;
    float* pw = reinterpret_cast<float*>(bucket_lds + half_cap);
    uint32_t* rcnt = bucket_lds + 2 * half_cap;
    __shared__ uint32_t wsum[WAVES];
    __shared__ uint32_t round_total;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int32_t row0 = b << pl.rl;
    const int32_t b0 = off[static_cast<int64_t>(b) * pl.g], b1 = off[static_cast<int64_t>(b + 1) * pl.g];
    const int cnt_b = b1 - b0;
    if (b == pl.nb - 1 && t == 0) {
        rs[n] = b1;
        rs[n + 1] = b1;
        ucnt[n] = 0;                                              // the scan's (n + 1)-th input
    }
    const int sh = pl.cbits + 1;
    const uint32_t kmask = (1u << sh) - 1u;
    // A bucket (or half of one) that does not fit the LDS is reported in info[1] (host: the sorted pipeline) and handed on as ONE
    // over-long row of harmless keys: everything behind this kernel stays inside its arrays whatever the host decides later.
    if (cnt_b > pl.cap) {
        if (t == 0) atomicAdd(reinterpret_cast<unsigned long long*>(info + 1), 1ull);
        for (int i = t; i < cnt_b; i += THREADS) {
            keys2[b0 + i] = 0u;
            w2[b0 + i] = 0.f;
        }
        for (int i = t; i < (1 << pl.rl); i += THREADS)
            if (row0 + i < n) rs[row0 + i] = b0;
        return;
    }
    // the bucket's entries, all loads in flight at once (a loop of load -> LDS atomic ran one L2 round trip per entry and pass: the
    // kernel took 0.46 ms); this kernel merges nothing, so the registers are free for them
    constexpr int EPT = 32;
    uint32_t ek[EPT];
    float ew[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int i = t + k * THREADS;
        ek[k] = i < cnt_b ? __builtin_nontemporal_load(stream + b0 + i) : 0u;
        ew[k] = i < cnt_b ? __builtin_nontemporal_load(wstream + b0 + i) : 0.f;
    }
    int32_t pos0 = b0;                                             // first stream position of this round's rows
    for (int round = 0; round < 2; ++round) {
        for (int i = t; i <= hrow; i += THREADS) rcnt[i] = 0;
        __syncthreads();
        const uint32_t rlo = static_cast<uint32_t>(round * hrow);
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const uint32_t rl_ = ek[k] >> sh;
            if (t + k * THREADS < cnt_b && rl_ - rlo < static_cast<uint32_t>(hrow)) atomicAdd(&rcnt[rl_ - rlo], 1u);
        }
        __syncthreads();
        uint32_t mine = t < hrow ? rcnt[t] : 0u, inc = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        uint32_t base = 0, all = 0;
        for (int q = 0; q < WAVES; ++q) {
            if (q < wv) base += wsum[q];
            all += wsum[q];
        }
        const uint32_t excl = base + inc - mine;
        if (t == 0) round_total = all;
        const bool fits = all <= static_cast<uint32_t>(half_cap);  // (workgroup-uniform)
        if (t < hrow) {
            rcnt[t] = excl;                                        // placement cursor
            const int32_t r = row0 + static_cast<int32_t>(rlo) + t;
            if (r < n) rs[r] = pos0 + static_cast<int32_t>(fits ? excl : 0u);
        }
        __syncthreads();
        if (!fits) {
            if (t == 0) atomicAdd(reinterpret_cast<unsigned long long*>(info + 1), 1ull);
            for (uint32_t i = t; i < all; i += THREADS) {
                keys2[pos0 + i] = 0u;
                w2[pos0 + i] = 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const uint32_t rl_ = ek[k] >> sh;
                if (t + k * THREADS < cnt_b && rl_ - rlo < static_cast<uint32_t>(hrow)) {
                    const uint32_t at = atomicAdd(&rcnt[rl_ - rlo], 1u);
                    pk[at] = ek[k] & kmask;
                    pw[at] = ew[k];
                }
            }
            __syncthreads();
            for (uint32_t i = t; i < all; i += THREADS) {
                keys2[pos0 + i] = pk[i];
                w2[pos0 + i] = pw[i];
            }
        }
        __syncthreads();
        pos0 += static_cast<int32_t>(round_total);
        __syncthreads();
    }
}

// rocPRIM's gfx950 default for 64-bit keys sorts 8 bits per pass (3 passes for 20 row bits); 10 bits per pass with
// 1024-thread blocks needs 2 -- tools/probes/sort_probe.hip, 40 M keys: 0.82 -> 0.64 ms (keys), 1.08 -> 0.85 ms (pairs)
using RowSortConfig = rocprim::radix_sort_config<
    rocprim::default_config, rocprim::default_config,
    rocprim::radix_sort_onesweep_config<rocprim::kernel_config<1024, 12>, rocprim::kernel_config<1024, 14>, 10,
                                        rocprim::block_radix_rank_algorithm::match>>;

struct MagopWs {
    size_t keys_a, keys_b, w_a, w_b, rs, ucnt, dinv, shift, ent, long_rows, n_long, sort_tmp, scan_tmp, hist, off, sort_tmp_bytes,
        scan_tmp_bytes, total;
};

int magop_layout(int64_t e, int32_t n, int weighted, MagopWs* w)
{
    const size_t m = static_cast<size_t>(2 * (e > 0 ? e : 1));
    const size_t nn = static_cast<size_t>(n);
    size_t sort_tmp = 0, scan_tmp = 0;
    uint64_t* k = nullptr;
    float* v = nullptr;
    int32_t* i32 = nullptr;
    const unsigned b0 = 32u, b1 = 32u + static_cast<unsigned>(bits_for(static_cast<uint64_t>(n)));
    if (weighted)
        PYGSD_HIP_TRY(rocprim::radix_sort_pairs<RowSortConfig>(nullptr, sort_tmp, k, k, v, v, m, b0, b1, hipStream_t(nullptr)));
    else
        PYGSD_HIP_TRY(rocprim::radix_sort_keys<RowSortConfig>(nullptr, sort_tmp, k, k, m, b0, b1, hipStream_t(nullptr)));
    PYGSD_HIP_TRY(rocprim::exclusive_scan(nullptr, scan_tmp, rocprim::make_transform_iterator(i32, PlusOne()), i32, 0,
                                          nn + 1, rocprim::plus<int32_t>(), hipStream_t(nullptr)));
    BucketPlan pl;
    size_t table = 0;
    if (bucket_plan(e, n, 0, &pl)) {                              // (both forms: unit weights and, since round 5, real-valued ones)
        table = static_cast<size_t>(pl.nb) * pl.g + 1;
        size_t scan2 = 0;
        PYGSD_HIP_TRY(rocprim::exclusive_scan(nullptr, scan2, i32, i32, 0, table, rocprim::plus<int32_t>(), hipStream_t(nullptr)));
        if (scan2 > scan_tmp) scan_tmp = scan2;
    }
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += round_up(bytes, 256); return o; };
    w->keys_a = take(m * 8);        // edge_keys output; dead after the sort: the records (16 B) reuse keys_a + ent
    w->ent = take(m * 8);           //   (contiguous with keys_a: take() rounds to 256 B, m * 8 is kept a multiple below)
    w->keys_b = take(m * 8);
    w->w_a = take(weighted ? m * 4 : 0);
    w->w_b = take(weighted ? m * 4 : 0);
    w->rs = take((nn + 2) * 4);
    w->ucnt = take((nn + 1) * 4);
    w->dinv = take((nn + 1) * 4 + (kUnitRowMax + 1) * 4 + 256);     // (pygsd_magop_unit: 2-byte entry counts [n] + its deg^-1/2 table)
    w->shift = take((nn + 1) * 4);
    w->long_rows = take((nn + 1) * 4);
    w->n_long = take(256);
    w->sort_tmp = take(sort_tmp);
    w->scan_tmp = take(scan_tmp);
    w->hist = take(table * 4);
    w->off = take(table * 4);
    w->sort_tmp_bytes = sort_tmp;
    w->scan_tmp_bytes = scan_tmp;
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

extern "C" int pygsd_magop_workspace(int64_t n_edges, int32_t n, int32_t weighted, size_t* bytes)
{
    PYGSD_REQUIRE(bytes, "pygsd_magop_workspace: null output");
    PYGSD_REQUIRE(n >= 0 && n_edges >= 0 && 2 * n_edges < (int64_t(1) << 31), "pygsd_magop_workspace: size out of int32 range");
    MagopWs w;
    if (int rc = magop_layout(n_edges, n, weighted, &w)) return rc;
    *bytes = w.total;
    return 0;
}

// sorted_front: the radix sort on the row bits in front of the row merge (round 3; rows of up to 4096 entries, any duplicates);
// otherwise weighted graphs the bucket plan takes go through bucket_scatter_w / bucket_place_rows_w (round 5) and are merged by
// the same row kernels reading the row-grouped 4-byte stream
static int magop_stage1_impl(const int64_t* row, const int64_t* col, const float* w, int64_t n_edges, int32_t n,
                             int32_t is_signed, int32_t absolute_degree, int32_t sym, void* workspace,
                             size_t workspace_bytes, int32_t* rowptr, float* deg, int64_t* d_info, void* stream, bool sorted_front)
{
    PYGSD_REQUIRE(n >= 0 && n_edges >= 0 && 2 * n_edges < (int64_t(1) << 31), "pygsd_magop_stage1: size out of int32 range");
    PYGSD_REQUIRE(workspace && rowptr && d_info && (n == 0 || deg), "pygsd_magop_stage1: null pointer");
    PYGSD_REQUIRE(n_edges == 0 || (row && col), "pygsd_magop_stage1: null edge list");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    const int weighted = w != nullptr;
    MagopWs l;
    if (int rc = magop_layout(n_edges, n, weighted, &l)) return rc;
    PYGSD_REQUIRE(workspace_bytes >= l.total, "pygsd_magop_stage1: workspace too small (%zu < %zu)", workspace_bytes, l.total);
    char* base = align256(workspace);
    uint64_t* keys_a = reinterpret_cast<uint64_t*>(base + l.keys_a);
    uint64_t* keys_b = reinterpret_cast<uint64_t*>(base + l.keys_b);
    float* w_a = weighted ? reinterpret_cast<float*>(base + l.w_a) : nullptr;
    float* w_b = weighted ? reinterpret_cast<float*>(base + l.w_b) : nullptr;
    int32_t* rs = reinterpret_cast<int32_t*>(base + l.rs);
    int32_t* ucnt = reinterpret_cast<int32_t*>(base + l.ucnt);
    float* dinv = reinterpret_cast<float*>(base + l.dinv);
    int32_t* shift = reinterpret_cast<int32_t*>(base + l.shift);
    uint4* ent = reinterpret_cast<uint4*>(base + l.keys_a);       // 16 B x m over keys_a + ent
    int32_t* long_rows = reinterpret_cast<int32_t*>(base + l.long_rows);
    int32_t* n_long = reinterpret_cast<int32_t*>(base + l.n_long);
    const int64_t m = 2 * n_edges;
    PYGSD_REQUIRE(l.ent == l.keys_a + round_up(static_cast<size_t>(2 * (n_edges > 0 ? n_edges : 1)) * 8, 256),
                  "pygsd_magop_stage1: record area is not contiguous");

    PYGSD_HIP_TRY(hipMemsetAsync(d_info, 0, 4 * sizeof(int64_t), s));
    PYGSD_HIP_TRY(hipMemsetAsync(n_long, 0, sizeof(int32_t), s));
    PYGSD_HIP_TRY(hipMemsetAsync(ucnt + n, 0, sizeof(int32_t), s));
    const int deg_mode = is_signed ? (absolute_degree ? 2 : 1) : 0;
    BucketPlan pl;
    const char* wform = getenv("PYGSD_WEIGHTED_BUILD_FORM");      // "sort": the radix-sort front for every graph (measurement / tests)
    if (weighted && !sorted_front && !(wform && strcmp(wform, "sort") == 0) && n > 0 && n_edges > 0 &&
        bucket_plan(n_edges, n, 0, &pl)) {
        int32_t* hist = reinterpret_cast<int32_t*>(base + l.hist);
        int32_t* off = reinterpret_cast<int32_t*>(base + l.off);
        uint32_t* stream_k = reinterpret_cast<uint32_t*>(keys_b);
        hipLaunchKernelGGL(bucket_count, dim3(pl.g), dim3(kPassThreads), 0, s, row, col, static_cast<const float*>(nullptr), 0, n_edges, n,
                           pl, hist, d_info, 0.f, reinterpret_cast<float*>(base + l.n_long) + 4, dinv);
        if (int rc = check_launch("bucket_count")) return rc;
        size_t tb2 = l.scan_tmp_bytes;
        PYGSD_HIP_TRY(rocprim::exclusive_scan(base + l.scan_tmp, tb2, hist, off, 0, static_cast<size_t>(pl.nb) * pl.g + 1,
                                              rocprim::plus<int32_t>(), s));
        const size_t lds2 = (static_cast<size_t>(2 * kTileEdges) + 2 * kScatterThreads + 2 * static_cast<size_t>(pl.nb)) * sizeof(uint32_t);
        static const hipError_t once2 = hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_scatter_w),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        PYGSD_HIP_TRY(once2);
        hipLaunchKernelGGL(bucket_scatter_w, dim3(pl.g), dim3(kScatterThreads), lds2, s, row, col, w, n_edges, n, pl, off, stream_k, w_b);
        if (int rc = check_launch("bucket_scatter_w")) return rc;
        const size_t lds = (static_cast<size_t>(pl.cap) + 2 * (((size_t(1) << pl.rl) >> 1) + 8)) * sizeof(uint32_t);
        static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_place_rows_w<1024>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 148 * 1024);
        PYGSD_HIP_TRY(once);
        PYGSD_REQUIRE(lds <= 148 * 1024, "pygsd_magop_stage1: bucket plan needs %zu B of LDS", lds);
        // row-grouped streams: keys in the second half of the key buffer (the bucket stream takes 4 of its 8 bytes per entry), weights
        // in the sort's other weight buffer; the record area (keys_a + ent) is written by the merge kernels behind them
        uint32_t* keys2 = stream_k + m;
        hipLaunchKernelGGL(bucket_place_rows_w<1024>, dim3(pl.nb), dim3(1024), lds, s, pl, stream_k, static_cast<const float*>(w_b), off, n,
                           rs, ucnt, keys2, w_a, d_info);
        if (int rc = check_launch("bucket_place_rows_w")) return rc;
        {
            const int64_t per_block = 4 * kRowsPerWave;
            const unsigned grid = static_cast<unsigned>((static_cast<int64_t>(n) + per_block - 1) / per_block);
            hipLaunchKernelGGL((row_merge_wave<uint32_t, uint32_t, true>), dim3(grid), dim3(kBlock), 0, s, keys2, static_cast<const float*>(w_a),
                               rs, n, m, deg_mode, ucnt, deg, ent, long_rows, n_long, d_info);
            if (int rc = check_launch("row_merge_wave")) return rc;
            hipLaunchKernelGGL((row_merge_block<uint32_t, true>), dim3(n < 1024 ? 64 : 1024), dim3(kBlock), 0, s, keys2,
                               static_cast<const float*>(w_a), rs, deg_mode, ucnt, deg, ent, long_rows, n_long, d_info);
            if (int rc = check_launch("row_merge_block")) return rc;
        }
        size_t tb3 = l.scan_tmp_bytes;
        PYGSD_HIP_TRY(rocprim::exclusive_scan(base + l.scan_tmp, tb3, rocprim::make_transform_iterator(ucnt, PlusOne()), rowptr, 0,
                                              static_cast<size_t>(n) + 1, rocprim::plus<int32_t>(), s));
        hipLaunchKernelGGL(row_tables, dim3(grid_for(n)), dim3(kBlock), 0, s, deg, rs, rowptr, n, sym, dinv, shift, d_info);
        return check_launch("row_tables");
    }
    if (n_edges > 0) {
        hipLaunchKernelGGL(edge_keys, dim3(grid_for(n_edges)), dim3(kBlock), 0, s, row, col, w, n_edges, n, keys_a, w_a, d_info);
        if (int rc = check_launch("edge_keys")) return rc;
        size_t tb = l.sort_tmp_bytes;
        const unsigned b0 = 32u, b1 = 32u + static_cast<unsigned>(bits_for(static_cast<uint64_t>(n)));
        if (weighted)
            PYGSD_HIP_TRY(rocprim::radix_sort_pairs<RowSortConfig>(base + l.sort_tmp, tb, keys_a, keys_b, w_a, w_b, static_cast<size_t>(m),
                                                    b0, b1, s));
        else
            PYGSD_HIP_TRY(rocprim::radix_sort_keys<RowSortConfig>(base + l.sort_tmp, tb, keys_a, keys_b, static_cast<size_t>(m), b0, b1, s));
        hipLaunchKernelGGL(key_row_starts, dim3(grid_for(m + 1)), dim3(kBlock), 0, s, keys_b, m, n, rs);
        if (int rc = check_launch("key_row_starts")) return rc;
    } else {
        PYGSD_HIP_TRY(hipMemsetAsync(rs, 0, sizeof(int32_t) * (static_cast<size_t>(n) + 2), s));
    }
    if (n > 0) {
        const int64_t per_block = 4 * kRowsPerWave;
        const unsigned grid = static_cast<unsigned>((static_cast<int64_t>(n) + per_block - 1) / per_block);
        if (n <= (1 << 25))
            hipLaunchKernelGGL(row_merge_wave<uint32_t>, dim3(grid), dim3(kBlock), 0, s, keys_b, w_b, rs, n, m > 0 ? m : 1, deg_mode,
                               ucnt, deg, ent, long_rows, n_long);
        else
            hipLaunchKernelGGL(row_merge_wave<uint64_t>, dim3(grid), dim3(kBlock), 0, s, keys_b, w_b, rs, n, m > 0 ? m : 1, deg_mode,
                               ucnt, deg, ent, long_rows, n_long);
        if (int rc = check_launch("row_merge_wave")) return rc;
        hipLaunchKernelGGL(row_merge_block<>, dim3(n < 1024 ? 64 : 1024), dim3(kBlock), 0, s, keys_b, w_b, rs, deg_mode, ucnt,
                           deg, ent, long_rows, n_long, d_info);
        if (int rc = check_launch("row_merge_block")) return rc;
    }
    size_t tb = l.scan_tmp_bytes;
    PYGSD_HIP_TRY(rocprim::exclusive_scan(base + l.scan_tmp, tb, rocprim::make_transform_iterator(ucnt, PlusOne()), rowptr, 0,
                                          static_cast<size_t>(n) + 1, rocprim::plus<int32_t>(), s));
    hipLaunchKernelGGL(row_tables, dim3(grid_for(n > 0 ? n : 1)), dim3(kBlock), 0, s, deg, rs, rowptr, n, sym, dinv, shift,
                       d_info);
    return check_launch("row_tables");
}

extern "C" int pygsd_magop_stage1(const int64_t* row, const int64_t* col, const float* w, int64_t n_edges, int32_t n,
                                  int32_t is_signed, int32_t absolute_degree, int32_t sym, void* workspace,
                                  size_t workspace_bytes, int32_t* rowptr, float* deg, int64_t* d_info, void* stream)
{
    return magop_stage1_impl(row, col, w, n_edges, n, is_signed, absolute_degree, sym, workspace, workspace_bytes, rowptr, deg, d_info,
                             stream, false);
}

extern "C" int pygsd_magop_stage1_sorted(const int64_t* row, const int64_t* col, const float* w, int64_t n_edges, int32_t n,
                                         int32_t is_signed, int32_t absolute_degree, int32_t sym, void* workspace,
                                         size_t workspace_bytes, int32_t* rowptr, float* deg, int64_t* d_info, void* stream)
{
    return magop_stage1_impl(row, col, w, n_edges, n, is_signed, absolute_degree, sym, workspace, workspace_bytes, rowptr, deg, d_info,
                             stream, true);
}

extern "C" int pygsd_magop_stage2(int64_t n_edges, int32_t n, int32_t weighted, double q, int32_t sym, float lambda_max,
                                  float diag_shift, void* workspace, size_t workspace_bytes, const int32_t* rowptr,
                                  const float* deg, int32_t* col, float* vb_real, float* vb_imag, float* vf_real,
                                  float* vf_imag, void* stream)
{
    PYGSD_REQUIRE(n >= 0 && n_edges >= 0 && 2 * n_edges < (int64_t(1) << 31), "pygsd_magop_stage2: size out of int32 range");
    if (n == 0) return 0;
    PYGSD_REQUIRE(workspace && rowptr && deg && col && vb_real && vb_imag && vf_real && vf_imag,
                  "pygsd_magop_stage2: null pointer");
    PYGSD_REQUIRE(aligned16(col) && aligned16(vb_real) && aligned16(vb_imag) && aligned16(vf_real) && aligned16(vf_imag),
                  "pygsd_magop_stage2: output arrays must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    MagopWs l;
    if (int rc = magop_layout(n_edges, n, weighted, &l)) return rc;
    PYGSD_REQUIRE(workspace_bytes >= l.total, "pygsd_magop_stage2: workspace too small");
    char* base = align256(workspace);
    // torch evaluates 1j*2*pi*q as a double-precision Python complex, then casts it to complex64 (one rounding: q is a double)
    const float two_pi_q = static_cast<float>(2.0 * 3.14159265358979323846 * q);
    const int64_t m = 2 * n_edges;
    int32_t* rs = reinterpret_cast<int32_t*>(base + l.rs);
    if (m > 0) {
        hipLaunchKernelGGL(values_entries, dim3(grid_for(m)), dim3(kBlock), 0, s, reinterpret_cast<uint4*>(base + l.keys_a), rs,
                           reinterpret_cast<int32_t*>(base + l.shift), deg, reinterpret_cast<float*>(base + l.dinv), n, two_pi_q,
                           sym, lambda_max, diag_shift, col, vb_real, vb_imag, vf_real, vf_imag);
        if (int rc = check_launch("values_entries")) return rc;
    }
    hipLaunchKernelGGL(diagonal_of_empty_rows, dim3(grid_for(n)), dim3(kBlock), 0, s, rowptr,
                       reinterpret_cast<int32_t*>(base + l.ucnt), deg, n, sym, lambda_max, diag_shift, col, vb_real, vb_imag,
                       vf_real, vf_imag);
    return check_launch("diagonal_of_empty_rows");
}

// w == NULL: unit weights (pygsd_magop_unit).  w != NULL: weights that must all be +1 or -1 (pygsd_magop_unit_signed; -1 only with
// allow_neg) -- the bucket form only; a graph its plan does not take is reported like an over-long row (d_info[1] != 0).
static int magop_unit_impl(const int64_t* row, const int64_t* col, const float* w, int32_t allow_neg, int64_t n_edges, int32_t n,
                           int32_t sym, double q, float lambda_max, float diag_shift, void* workspace, size_t workspace_bytes,
                           int32_t* rowptr, float* deg, int32_t* ccol, float* vb_real, float* vb_imag, float* vf_real,
                           float* vf_imag, int64_t* d_info, int32_t phase, void* stream)
{
    const bool is_pm1 = w != nullptr;
    PYGSD_REQUIRE(n >= 0 && n_edges >= 0 && 2 * n_edges < (int64_t(1) << 31), "pygsd_magop_unit: size out of int32 range");
    PYGSD_REQUIRE(phase >= 0 && phase <= 2, "pygsd_magop_unit: phase must be 0 (all), 1 (up to the row pointer) or 2 (the write kernel)");
    PYGSD_REQUIRE(workspace && rowptr && d_info && (n == 0 || (deg && ccol && vb_real && vb_imag && vf_real && vf_imag)),
                  "pygsd_magop_unit: null pointer");
    PYGSD_REQUIRE(n_edges == 0 || (row && col), "pygsd_magop_unit: null edge list");
    PYGSD_REQUIRE(n == 0 || (aligned16(ccol) && aligned16(vb_real) && aligned16(vb_imag) && aligned16(vf_real) && aligned16(vf_imag)),
                  "pygsd_magop_unit: output arrays must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    MagopWs l;
    if (int rc = magop_layout(n_edges, n, 0, &l)) return rc;
    PYGSD_REQUIRE(workspace_bytes >= l.total, "pygsd_magop_unit: workspace too small (%zu < %zu)", workspace_bytes, l.total);
    char* base = align256(workspace);
    uint64_t* keys_a = reinterpret_cast<uint64_t*>(base + l.keys_a);
    uint64_t* keys_b = reinterpret_cast<uint64_t*>(base + l.keys_b);
    int32_t* rs = reinterpret_cast<int32_t*>(base + l.rs);
    const uint16_t* cnt16 = reinterpret_cast<const uint16_t*>(base + l.dinv);
    const float* lut = reinterpret_cast<const float*>(base + l.dinv + round_up(static_cast<size_t>(n) * 2 + 2, 256));
    const int64_t m = 2 * n_edges;
    if (phase != 2) PYGSD_HIP_TRY(hipMemsetAsync(d_info, 0, 4 * sizeof(int64_t), s));
    if (n == 0) {
        if (phase != 2) PYGSD_HIP_TRY(hipMemsetAsync(rowptr, 0, sizeof(int32_t), s));
        return 0;
    }
    int32_t* ucnt = reinterpret_cast<int32_t*>(base + l.ucnt);
    int32_t* left = reinterpret_cast<int32_t*>(base + l.shift);
    // torch evaluates 1j*2*pi*q as a double-precision Python complex, then casts it to complex64 (one rounding: q is a double)
    const float two_pi_q = static_cast<float>(2.0 * 3.14159265358979323846 * q);
    UnitArgs a{keys_b, keys_a, rs, deg, cnt16, lut, rowptr, ccol, vb_real, vb_imag, vf_real, vf_imag, ucnt, left, d_info,
               m > 0 ? m : 1, n, sym, two_pi_q, lambda_max, diag_shift, reinterpret_cast<float*>(base + l.n_long)};
    const int64_t per_block = 4 * kRowsPerWave;
    const unsigned grid = static_cast<unsigned>((static_cast<int64_t>(n) + per_block - 1) / per_block);
    const unsigned chunks = static_cast<unsigned>((static_cast<int64_t>(n) + kChunkRows - 1) / kChunkRows);
    if (phase == 2) {
        if (is_pm1)
            hipLaunchKernelGGL((unit_write_chunks<256, kChunkSlots, true>), dim3(chunks), dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL((unit_write_chunks<256, kChunkSlots>), dim3(chunks), dim3(256), 0, s, a);
        return check_launch("unit_write_chunks");
    }
    BucketPlan pl;
    const char* form = getenv("PYGSD_UNIT_BUILD_FORM");          // "sort": the radix-sort form (measurement / tests)
    const bool buckets = !(form && strcmp(form, "sort") == 0) && bucket_plan(n_edges, n, is_pm1 ? 1 : 0, &pl);
    if (is_pm1 && !buckets) {
        // the signed form exists for the bucket plan only: report "not taken" (every byte 1: non-zero) -- the caller's two-stage
        // pipeline builds this graph
        PYGSD_HIP_TRY(hipMemsetAsync(d_info + 1, 1, sizeof(int64_t), s));
        return 0;
    }
    if (buckets) {
        int32_t* hist = reinterpret_cast<int32_t*>(base + l.hist);
        int32_t* off = reinterpret_cast<int32_t*>(base + l.off);
        uint32_t* stream = reinterpret_cast<uint32_t*>(keys_b);
        hipLaunchKernelGGL(bucket_count, dim3(pl.g), dim3(kPassThreads), 0, s, row, col, w, allow_neg, n_edges, n, pl, hist, d_info,
                           two_pi_q, a.trig, const_cast<float*>(a.lut));
        if (int rc = check_launch("bucket_count")) return rc;
        size_t tb = l.scan_tmp_bytes;
        PYGSD_HIP_TRY(rocprim::exclusive_scan(base + l.scan_tmp, tb, hist, off, 0, static_cast<size_t>(pl.nb) * pl.g + 1,
                                              rocprim::plus<int32_t>(), s));
        const size_t lds2 = (static_cast<size_t>(2 * kTileEdges) + 2 * kScatterThreads + 2 * static_cast<size_t>(pl.nb)) * sizeof(uint32_t);
        const size_t lds = (static_cast<size_t>(pl.cap) + 2 * ((size_t(1) << pl.rl) + 8)) * sizeof(uint32_t);
        if (is_pm1) {
            static const hipError_t once2 = hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_scatter<true>),
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            PYGSD_HIP_TRY(once2);
            hipLaunchKernelGGL(bucket_scatter<true>, dim3(pl.g), dim3(kScatterThreads), lds2, s, row, col, w, n_edges, n, pl, off, stream,
                               static_cast<const int64_t*>(d_info));
            if (int rc = check_launch("bucket_scatter")) return rc;
            static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_merge_rows<1024, true>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            PYGSD_HIP_TRY(once);
            hipLaunchKernelGGL((bucket_merge_rows<1024, true>), dim3(pl.nb), dim3(1024), lds, s, a, pl, stream, off);
        } else {
            static const hipError_t once2 = hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_scatter<false>),
                                                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            PYGSD_HIP_TRY(once2);
            hipLaunchKernelGGL(bucket_scatter<false>, dim3(pl.g), dim3(kScatterThreads), lds2, s, row, col, w, n_edges, n, pl, off, stream,
                               static_cast<const int64_t*>(d_info));
            if (int rc = check_launch("bucket_scatter")) return rc;
            static const hipError_t once = hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_merge_rows<1024, false>),
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            PYGSD_HIP_TRY(once);
            hipLaunchKernelGGL((bucket_merge_rows<1024, false>), dim3(pl.nb), dim3(1024), lds, s, a, pl, stream, off);
        }
        if (int rc = check_launch("bucket_merge_rows")) return rc;
    } else {
        if (n_edges > 0) {
            hipLaunchKernelGGL(edge_keys, dim3(grid_for(n_edges)), dim3(kBlock), 0, s, row, col, static_cast<const float*>(nullptr),
                               n_edges, n, keys_a, static_cast<float*>(nullptr), d_info);
            if (int rc = check_launch("edge_keys")) return rc;
            size_t tb = l.sort_tmp_bytes;
            const unsigned b0 = 32u, b1 = 32u + static_cast<unsigned>(bits_for(static_cast<uint64_t>(n)));
            PYGSD_HIP_TRY(rocprim::radix_sort_keys<RowSortConfig>(base + l.sort_tmp, tb, keys_a, keys_b, static_cast<size_t>(m), b0, b1, s));
            hipLaunchKernelGGL(key_row_starts, dim3(grid_for(m + 1)), dim3(kBlock), 0, s, keys_b, m, n, rs);
            if (int rc = check_launch("key_row_starts")) return rc;
        } else {
            PYGSD_HIP_TRY(hipMemsetAsync(rs, 0, sizeof(int32_t) * (static_cast<size_t>(n) + 2), s));
        }
        hipLaunchKernelGGL(unit_row_tables, dim3(grid_for(n)), dim3(kBlock), 0, s, rs, n, deg, const_cast<uint16_t*>(a.cnt16), ucnt, two_pi_q,
                           a.trig, const_cast<float*>(a.lut));
        if (int rc = check_launch("unit_row_tables")) return rc;
        if (n <= (1 << 25))
            hipLaunchKernelGGL(unit_merge_rows<uint32_t>, dim3(grid), dim3(kBlock), 0, s, a);
        else
            hipLaunchKernelGGL(unit_merge_rows<uint64_t>, dim3(grid), dim3(kBlock), 0, s, a);
        if (int rc = check_launch("unit_merge_rows")) return rc;
    }
    size_t tb = l.scan_tmp_bytes;
    PYGSD_HIP_TRY(rocprim::exclusive_scan(base + l.scan_tmp, tb, rocprim::make_transform_iterator(ucnt, PlusOne()), rowptr, 0,
                                          static_cast<size_t>(n) + 1, rocprim::plus<int32_t>(), s));
    hipLaunchKernelGGL(unit_finish_info, dim3(1), dim3(1), 0, s, rowptr, n, d_info);          // E_s: d_info is final from here on
    if (int rc = check_launch("unit_finish_info")) return rc;
    if (phase == 1) return 0;
    if (is_pm1)
        hipLaunchKernelGGL((unit_write_chunks<256, kChunkSlots, true>), dim3(chunks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((unit_write_chunks<256, kChunkSlots>), dim3(chunks), dim3(256), 0, s, a);
    return check_launch("unit_write_chunks");
}

extern "C" int pygsd_magop_unit(const int64_t* row, const int64_t* col, int64_t n_edges, int32_t n, int32_t sym, double q,
                                float lambda_max, float diag_shift, void* workspace, size_t workspace_bytes, int32_t* rowptr,
                                float* deg, int32_t* ccol, float* vb_real, float* vb_imag, float* vf_real, float* vf_imag,
                                int64_t* d_info, int32_t phase, void* stream)
{
    return magop_unit_impl(row, col, nullptr, 0, n_edges, n, sym, q, lambda_max, diag_shift, workspace, workspace_bytes, rowptr, deg,
                           ccol, vb_real, vb_imag, vf_real, vf_imag, d_info, phase, stream);
}

extern "C" int pygsd_magop_unit_signed(const int64_t* row, const int64_t* col, const float* w, int64_t n_edges, int32_t n,
                                       int32_t is_signed, int32_t absolute_degree, int32_t sym, double q, float lambda_max,
                                       float diag_shift, void* workspace, size_t workspace_bytes, int32_t* rowptr, float* deg,
                                       int32_t* ccol, float* vb_real, float* vb_imag, float* vf_real, float* vf_imag,
                                       int64_t* d_info, int32_t phase, void* stream)
{
    PYGSD_REQUIRE(w || n_edges == 0, "pygsd_magop_unit_signed: null weights (pygsd_magop_unit builds unweighted graphs)");
    if (!w) {
        PYGSD_REQUIRE(d_info, "pygsd_magop_unit_signed: null pointer");
        if (phase != 2) PYGSD_HIP_TRY(hipMemsetAsync(d_info, 0, 4 * sizeof(int64_t), static_cast<hipStream_t>(stream)));
        if (phase != 2) PYGSD_HIP_TRY(hipMemsetAsync(d_info + 1, 1, sizeof(int64_t), static_cast<hipStream_t>(stream)));
        return 0;
    }
    // -1 is admitted where the degree counts |w| (signed Laplacian, absolute_degree): deg = entries / 2 then, as with unit weights.
    // Elsewhere (deg = sum of A_s or of |A_s|) only +1 keeps that identity, so a -1 sends the graph to the two-stage pipeline.
    const int32_t allow_neg = (is_signed && absolute_degree) ? 1 : 0;
    return magop_unit_impl(row, col, w, allow_neg, n_edges, n, sym, q, lambda_max, diag_shift, workspace, workspace_bytes, rowptr, deg,
                           ccol, vb_real, vb_imag, vf_real, vf_imag, d_info, phase, stream);
}
