// Tall-skinny linear maps of the non-magnetic layers on the matrix cores (bf16: v_mfma_f32_16x16x32_bf16, fp32 accumulate;
// fp32: the exact v_mfma_f32_16x16x4_f32) and the column sums their bias gradients need.
//
//   Y[n, f_out] = [X_0 | X_1 | ...] W (+ bias)          n ~ 10^5..10^7 rows, K = sum of the segment widths <= 256, f_out <= 256
//
// One pass: every operand row is read once, every output row written once, whatever the number of column segments --
// the segments are what the layers would otherwise concatenate or multiply one GEMM at a time and accumulate:
//   x W                        DiGCNConv.py:66, the Linear of DiGCN_Inception_Block.py:44, SGCNConv.py:121-126
//   [g_0 | dP_1 | dP_2] W^T    their input gradient in ONE product (two library GEMMs + an accumulation pass before)
//
// Tiling (as csrc/dense.hip): one wavefront owns 16 rows.  The MFMA is issued TRANSPOSED -- A operand = a fragment of W
// (its 16 rows are OUTPUT columns), B operand = the lane's own 16-byte row load (lane l: row l & 15, k-slots of quarter
// l >> 4), so activations never pass through LDS and the C/D layout (col = lane & 15, row = 4 (lane >> 4) + reg) hands every
// lane consecutive output columns of ONE row.  W is laid out in LDS once per block IN FRAGMENT ORDER ([k-block][tile][lane],
// 16 B per lane for bf16, 4 B for fp32): every A fragment is one conflict-free ds_read.  For bf16 the output columns of two
// neighbouring tiles are interleaved (column 32 m + 8 q + 4 (t & 1) + r for tile t = 2 m + (t & 1), quarter q, register r) so
// that a lane's 8 results form one 16-byte store.  The next tile's rows are loaded before the current tile's MFMAs.
#include "common.hpp"

namespace pygsd {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kMaxKB = 16;     // k-blocks: 32 bf16 / 16 fp32 columns each (one 16-byte load per lane)
constexpr int kMaxSeg = 4;      // input column segments
constexpr int kMaxOut = 8;      // output column segments

struct TallArgs {
    const void* x[kMaxKB];     // first element of k-block kb in row 0 of its segment
    int64_t ld[kMaxKB];        // that segment's row stride in elements
    const void* w;             // W[k][n] at w[k * ldw + n]  (w_t == 0)  or  w[n * ldw + k]  (w_t != 0)
    int64_t ldw;
    const void* bias;          // f_out elements of the storage type, or null
    void* y[kMaxKB];           // first element of output block b (32 bf16 / 16 fp32 columns) in row 0 of its segment
    int64_t ldy[kMaxKB];       // that segment's row stride in elements
    int32_t n_rows, f_out, w_t;
};

__device__ __forceinline__ float bf16_value(uint32_t h) { return __uint_as_float(h << 16); }
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two fp32 -> two bf16 in one dword, round to nearest even: one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack2(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// the value lane (l ^ 8) holds: a rotation by 8 within each row of 16 lanes (DPP row_ror:8), full VALU rate
__device__ __forceinline__ uint32_t rotate8(uint32_t v)
{
    return static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x128, 0xf, 0xf, true));
}
__device__ __forceinline__ float rotate8(float v) { return __uint_as_float(rotate8(__float_as_uint(v))); }

// the 16-byte piece lane (j, q) reads of every k-block of row `tile * 16 + j` (clamped)
template <int KB>
__device__ __forceinline__ void bf16_rows_in(const TallArgs& p, int tile, int j, int q, uint4 (&dst)[KB])
{
    int64_t row = static_cast<int64_t>(tile) * 16 + j;
    row = row < p.n_rows ? row : p.n_rows - 1;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
        dst[kb] = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p.x[kb]) + row * p.ld[kb] + 8 * q);
}

// one tile of the bf16 kernel: MFMAs over the rows in `cur`, bias, rounding, stores
template <int KB, int NT>
__device__ __forceinline__ void bf16_tile_out(const TallArgs& p, const uint4* frag, const float* bias, int tile, int lane,
                                              const uint4 (&cur)[KB])
{
    const int j = lane & 15, q = lane >> 4;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const bf16x8 b = __builtin_bit_cast(bf16x8, cur[kb]);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bf16x8 a = __builtin_bit_cast(bf16x8, frag[(kb * NT + t) * 64 + lane]);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
        }
    }
    // lane (j, q): tiles 2 m and 2 m + 1 hold columns [32 m + 8 q, 32 m + 8 q + 8) of row 16 tile + j
    uint4 o[NT / 2];
#pragma unroll
    for (int m = 0; m < NT / 2; ++m) {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + 32 * m + 8 * q);
        const float4 b1 = *reinterpret_cast<const float4*>(bias + 32 * m + 8 * q + 4);
        const f32x4 lo = acc[2 * m], hi = acc[2 * m + 1];
        o[m].x = pack2(lo[0] + b0.x, lo[1] + b0.y);
        o[m].y = pack2(lo[2] + b0.z, lo[3] + b0.w);
        o[m].z = pack2(hi[0] + b1.x, hi[1] + b1.y);
        o[m].w = pack2(hi[2] + b1.z, hi[3] + b1.w);
    }
    const int64_t last = p.n_rows - 1;
    if constexpr ((NT / 2) % 2 == 0) {
        // output blocks (32 columns = 64 bytes a row) in pairs: lanes j and j ^ 8 trade one block of each pair, so that a store
        // instruction writes 8 rows x 128 contiguous bytes wherever the pair is one segment
        const bool upper = j >= 8;
        int64_t row_a = static_cast<int64_t>(tile) * 16 + (j & 7), row_b = row_a + 8;
        row_a = row_a < last ? row_a : last;          // clamped: such a lane holds row n_rows - 1's own result (bf16_rows_in)
        row_b = row_b < last ? row_b : last;
#pragma unroll
        for (int mm = 0; mm < NT / 4; ++mm) {
            const uint4 lo = o[2 * mm], hi = o[2 * mm + 1];
            const uint4 lo_far = make_uint4(rotate8(lo.x), rotate8(lo.y), rotate8(lo.z), rotate8(lo.w));
            const uint4 hi_far = make_uint4(rotate8(hi.x), rotate8(hi.y), rotate8(hi.z), rotate8(hi.w));
            const uint4 va = upper ? hi_far : lo, vb = upper ? hi : lo_far;
            uint16_t* const lo_a = static_cast<uint16_t*>(p.y[2 * mm]) + row_a * p.ldy[2 * mm] + 8 * q;
            uint16_t* const hi_a = static_cast<uint16_t*>(p.y[2 * mm + 1]) + row_a * p.ldy[2 * mm + 1] + 8 * q;
            uint16_t* const lo_b = static_cast<uint16_t*>(p.y[2 * mm]) + row_b * p.ldy[2 * mm] + 8 * q;
            uint16_t* const hi_b = static_cast<uint16_t*>(p.y[2 * mm + 1]) + row_b * p.ldy[2 * mm + 1] + 8 * q;
            *reinterpret_cast<uint4*>(upper ? hi_a : lo_a) = va;
            *reinterpret_cast<uint4*>(upper ? hi_b : lo_b) = vb;
        }
    } else {
        int64_t row = static_cast<int64_t>(tile) * 16 + j;
        row = row < last ? row : last;
#pragma unroll
        for (int m = 0; m < NT / 2; ++m)
            *reinterpret_cast<uint4*>(static_cast<uint16_t*>(p.y[m]) + row * p.ldy[m] + 8 * q) = o[m];
    }
}

// ---- bf16 storage, fp32 accumulation -----------------------------------------------------------
template <int KB, int NT>
__global__ __launch_bounds__(256) void tall_linear_bf16_kernel(TallArgs p)
{
    static_assert(NT % 2 == 0, "output tiles come in interleaved pairs");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* frag = reinterpret_cast<uint4*>(smem);                                   // [KB][NT][64] x 8 bf16
    float* bias = reinterpret_cast<float*>(smem + static_cast<size_t>(KB) * NT * 64 * 16);   // [NT * 16]
    const int tid = threadIdx.x;
    const uint16_t* w = static_cast<const uint16_t*>(p.w);
    for (int idx = tid; idx < KB * NT * 64; idx += 256) {
        const int lane = idx & 63, t = (idx >> 6) % NT, kb = (idx >> 6) / NT;
        const int i = lane & 15, q = lane >> 4;
        const int64_t n = 32 * (t >> 1) + 8 * (i >> 2) + 4 * (t & 1) + (i & 3);
        const int64_t k0 = kb * 32 + 8 * q;
        uint32_t h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = p.w_t ? w[n * p.ldw + k0 + e] : w[(k0 + e) * p.ldw + n];
        frag[idx] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    }
    for (int c = tid; c < NT * 16; c += 256)
        bias[c] = p.bias ? bf16_value(static_cast<const uint16_t*>(p.bias)[c]) : 0.f;
    __syncthreads();

    // Memory side as the fp32 split kernel below (round 5; the reasons are spelled out there): rows past the end clamped instead of
    // masked, two row buffers that trade roles with the first pair of steps written out, whole-line stores where the output
    // blocks come in pairs.  C5b: 0.242 + 0.202 -> 2 x 0.185 ms per step for the two products (profiles/r5u_configs_bf16.json).
    const int lane = tid & 63, j = lane & 15, q = lane >> 4;
    const int n_tiles = (p.n_rows + 15) >> 4;
    const int stride = static_cast<int>(gridDim.x) * 4;
    int tile = static_cast<int>(blockIdx.x) * 4 + (tid >> 6);
    if (tile >= n_tiles) return;
    const int last_tile = n_tiles - 1;
    uint4 rows_a[KB], rows_b[KB];
    bf16_rows_in<KB>(p, tile, j, q, rows_a);
    bf16_rows_in<KB>(p, min(tile + stride, last_tile), j, q, rows_b);
#define PYGSD_TALL_STEP(ROWS)                                                          \
    bf16_tile_out<KB, NT>(p, frag, bias, tile, lane, ROWS);                            \
    bf16_rows_in<KB>(p, min(tile + 2 * stride, last_tile), j, q, ROWS);                \
    tile += stride;                                                                    \
    if (tile >= n_tiles) return;
    PYGSD_TALL_STEP(rows_a)
    PYGSD_TALL_STEP(rows_b)
    for (;;) {
        PYGSD_TALL_STEP(rows_a)
        PYGSD_TALL_STEP(rows_b)
    }
#undef PYGSD_TALL_STEP
}

// ---- fp32, exact (an fmaf chain per output) ----------------------------------------------------
template <int KB, int NT>
__global__ __launch_bounds__(256) void tall_linear_f32_kernel(TallArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* frag = reinterpret_cast<float*>(smem);                  // [KB][4][NT][64]
    float* bias = frag + static_cast<size_t>(KB) * 4 * NT * 64;    // [NT * 16]
    const int tid = threadIdx.x;
    const float* w = static_cast<const float*>(p.w);
    for (int idx = tid; idx < KB * 4 * NT * 64; idx += 256) {
        const int lane = idx & 63, t = (idx >> 6) % NT, m = ((idx >> 6) / NT) & 3, kb = (idx >> 6) / NT / 4;
        const int64_t n = 16 * t + (lane & 15);
        const int64_t k = kb * 16 + 4 * (lane >> 4) + m;
        frag[idx] = p.w_t ? w[n * p.ldw + k] : w[k * p.ldw + n];
    }
    for (int c = tid; c < NT * 16; c += 256) bias[c] = p.bias ? static_cast<const float*>(p.bias)[c] : 0.f;
    __syncthreads();

    const int lane = tid & 63, j = lane & 15, q = lane >> 4;
    const int n_tiles = (p.n_rows + 15) >> 4;
    const int stride = static_cast<int>(gridDim.x) * 4;
    int tile = static_cast<int>(blockIdx.x) * 4 + (tid >> 6);
    float4 cur[KB], nxt[KB];
    if (tile < n_tiles) {
        const int64_t row = (tile * 16 + j < p.n_rows) ? tile * 16 + j : p.n_rows - 1;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
            cur[kb] = *reinterpret_cast<const float4*>(static_cast<const float*>(p.x[kb]) + row * p.ld[kb] + 4 * q);
    }
    for (; tile < n_tiles; tile += stride) {
        const int next = tile + stride;
        if (next < n_tiles) {
            const int64_t row = (next * 16 + j < p.n_rows) ? next * 16 + j : p.n_rows - 1;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                nxt[kb] = *reinterpret_cast<const float4*>(static_cast<const float*>(p.x[kb]) + row * p.ld[kb] + 4 * q);
        }
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // The W fragments are re-read from LDS for every tile.  Left alone, the compiler hoists all KB x 4 x NT of them into
        // registers (loop-invariant): 252-348 VGPRs, ONE or two wavefronts per SIMD and 0.42-0.52 of the HBM peak
        // (profiles/r4d_configs.json).  One tile's fragments cost KB x NT KiB of LDS reads against KB x 4 x NT x 32 cycles
        // of MFMA issue on each of the CU's four SIMDs: a quarter of the LDS bandwidth.
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const float xs[4] = {cur[kb].x, cur[kb].y, cur[kb].z, cur[kb].w};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(frag[((kb * 4 + m) * NT + t) * 64 + lane], xs[m], acc[t],
                                                                  0, 0, 0);
            }
        }
        if (tile * 16 + j < p.n_rows) {
            const int64_t row = tile * 16 + j;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float4 b = *reinterpret_cast<const float4*>(bias + 16 * t + 4 * q);
                *reinterpret_cast<float4*>(static_cast<float*>(p.y[t]) + row * p.ldy[t] + 4 * q) =
                    make_float4(acc[t][0] + b.x, acc[t][1] + b.y, acc[t][2] + b.z, acc[t][3] + b.w);
            }
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) cur[kb] = nxt[kb];
    }
}

// ---- fp32 storage, products on the bf16 matrix pipe by three-way splitting (round 5) --------------
// The exact kernel above sits on the power budget, not on a pipe: its memory half alone streams 5.4 TB/s at 2.1 - 2.3 GHz, its
// matrix half alone reaches 104 - 137 TF, together they run at 1.8 - 1.9 GHz and 0.49 - 0.55 of the HBM peak
// (tools/probes/tall_probe.hip, profiles/r5m_tall_probe.txt).  v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate; the way out
// is fewer matrix cycles per product.  Every fp32 value is the sum of three bf16 values up to 2^-24 of its magnitude:
//   hi = bf16(x),  mid = bf16(x - hi),  lo = bf16(x - hi - mid)          (round to nearest even; both differences are exact)
// and x w = the nine partial products of the two triples, of which the six largest are kept -- everything down to 2^-16 |x w|:
//   hi hi,  hi mid,  mid hi,  mid mid,  hi lo,  lo hi            (dropped: mid lo, lo mid <= 2^-24 |x w| each, lo lo <= 2^-32)
// Each bf16 x bf16 product is exact in fp32 (8 + 8 mantissa bits); v_mfma_f32_16x16x32_bf16 sums 32 of them into an fp32
// accumulator per instruction.  6 of those (16 cycles each, 32 k-slots) replace 8 of the fp32 form (32 cycles each, 4 k-slots):
// 2.7x fewer matrix cycles, and the kernel becomes what its memory half is.  Measured error against a float64 product, relative to
// sum |x| |w|: 1.3 - 1.7e-7 with the six terms added straight into the running sums (tools/probes/split_probe.hip ->
// profiles/r5n_split_probe.txt), 0.6 - 0.8e-7 with the six terms of every 32-column block summed apart and added once, as
// split_tile_out does (tools/tall_forms_probe.py -> profiles/r5t_tall_forms.json), against 2.8 - 3.5e-7 for the fmaf chain on the
// same inputs -- the dropped terms are smaller than the chain's own roundings.  NOT bitwise the fmaf chain.  Operands the split
// cannot carry -- +-inf, NaN, magnitudes above the largest bf16 (3.39e38) -- send their tile to exact fp32 arithmetic (see
// any_not_finite / exact_tile_store below); PYGSD_TALL_F32=exact keeps every fp32 product on the kernel above.
//
// Memory side (the kernel is memory-bound now, so these pay: 81 -> 77 us at K = 128 / f_out = 64, 90 -> 78 at 64 / 128, 500 -> 411
// at 64 / 192 for 2M rows -- same probe):
//   * whole-line stores: lanes j and j ^ 8 trade one tile of each pair (a DPP rotation by 8 inside the row of 16 lanes), so that a
//     store instruction writes 8 rows x 128 contiguous bytes instead of 16 rows x 64;
//   * rows past the end are CLAMPED, not masked (such a lane loaded row n_rows - 1, so what it holds IS that row's result and it
//     re-writes the same value there): every load and store is issued, the compiler's `s_waitcnt vmcnt(n)` counts are exact;
//   * two row buffers A, B; a step S(X) = { tile in X: split, MFMAs, stores; load the tile two steps on into X }; the kernel =
//     load A, B; S(A); S(B); loop { S(A); S(B) }.  With one pair and a `cur = nxt` copy the compiler moved each copy up behind the
//     last use of cur[kb] and waited there for loads issued at the top of the SAME step; with masked accesses it waits for fewer
//     operations than are in flight (a possibly-unissued store is not counted), i.e. for the previous tile's stores to be
//     acknowledged; entering the loop straight from the prologue lowers the loop's counts to the prologue's.  Steady state:
//     vmcnt(NT + 2 KB) -- the other tile's stores and its replacement's loads stay in flight across the wait.
// KB = k-blocks of 32 columns, NT = output tiles of 16 (even); W lives in LDS pre-split, in fragment order: [KB][NT][3][64] x 16 B.
__device__ __forceinline__ void split8(const float (&x)[8], uint4& h, uint4& m, uint4& l)
{
    uint32_t hh[4], mm[4], ll[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = x[2 * e], b = x[2 * e + 1];
        hh[e] = pack2(a, b);
        const float ra = a - __uint_as_float(hh[e] << 16), rb = b - __uint_as_float(hh[e] & 0xffff0000u);
        mm[e] = pack2(ra, rb);
        const float sa = ra - __uint_as_float(mm[e] << 16), sb = rb - __uint_as_float(mm[e] & 0xffff0000u);
        ll[e] = pack2(sa, sb);
    }
    h = make_uint4(hh[0], hh[1], hh[2], hh[3]);
    m = make_uint4(mm[0], mm[1], mm[2], mm[3]);
    l = make_uint4(ll[0], ll[1], ll[2], ll[3]);
}

// the two 16-byte pieces lane (j, q) reads of every k-block of row `tile * 16 + j` (clamped)
template <int KB>
__device__ __forceinline__ void split_rows_in(const TallArgs& p, int tile, int j, int q, float4 (&dst)[KB][2])
{
    int64_t row = static_cast<int64_t>(tile) * 16 + j;
    row = row < p.n_rows ? row : p.n_rows - 1;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const float* src = static_cast<const float*>(p.x[kb]) + row * p.ld[kb] + 8 * q;
        dst[kb][0] = *reinterpret_cast<const float4*>(src);
        dst[kb][1] = *reinterpret_cast<const float4*>(src + 4);
    }
}

// ---- non-finite and out-of-range operands (round 6) ------------------------------------------------------------------------
// The split is exact arithmetic on FINITE values below the largest bf16.  Outside that range it is not what the reference's
// torch.matmul computes: x = +-inf gives hi = x and mid = x - hi = NaN; a finite |x| above 3.39e38 rounds its hi to inf (then
// mid = -inf, lo = NaN); and even with the lower pieces forced to zero, inf times a W piece that happens to be zero (any weight
// that fits 8 / 16 mantissa bits) is NaN where inf * w is +-inf.  No arrangement of the three pieces repairs the last case, so
// the guard is on the RESULT: a NaN piece of either operand makes every sum it enters NaN (NaN * 0 = NaN on the matrix pipe as
// anywhere else), i.e. a tile whose sums are all finite had operands the split represents exactly.  A wavefront whose tile
// holds a non-finite sum recomputes that tile from the fp32 operands themselves (IEEE products and sums: inf, -inf and NaN come
// out where an fmaf chain puts them) -- the rows and W re-read from global memory (L2 hits), executed by the tiles that need it
// only.  The test costs two packed multiply-adds per output tile.
template <int NT>
__device__ __forceinline__ bool any_not_finite(const f32x4 (&acc)[NT])
{
    f32x4 z = acc[0] * 0.f;                                    // 0 for a finite value, NaN for +-inf and NaN
#pragma unroll
    for (int t = 1; t < NT; ++t) z += acc[t] * 0.f;
    const float c = (z[0] + z[1]) + (z[2] + z[3]);
    return __builtin_amdgcn_ballot_w64(c != c) != 0;           // wavefront-uniform
}

// The tile again, from the fp32 operands, and stored: lane (j, q) owns columns 16 t + 4 q .. + 3 of row j, as in the split form,
// and walks them as fma chains over k in order -- rolled loops, four sums and a few addresses live, so that the branch costs
// the kernel no registers (an MFMA form with its NT accumulators and loads in flight took a wavefront per SIMD from most
// instances).  The sums are kept in float64 and rounded once (a sequential fp32 chain of a few hundred terms sits at the 1e-5
// bar by itself; time is no object here): IEEE products and sums either way -- inf x 0 = NaN, inf - inf = NaN, a sum beyond
// FLT_MAX rounds to inf.
template <int KB, int NT>
__device__ __forceinline__ void exact_tile_store(const TallArgs& p, const float* bias, int tile, int j, int q)
{
    int64_t row = static_cast<int64_t>(tile) * 16 + j;
    row = row < p.n_rows ? row : p.n_rows - 1;                 // clamped as the split form: such a lane re-writes the last row
    const float* w = static_cast<const float*>(p.w);
    const int64_t ws_k = p.w_t ? 1 : p.ldw, ws_n = p.w_t ? p.ldw : 1;
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
        const float* wc = w + (16 * t + 4 * q) * ws_n;
        double s0 = 0., s1 = 0., s2 = 0., s3 = 0.;
#pragma unroll 1
        for (int kb = 0; kb < KB; ++kb) {
            const float* xr = static_cast<const float*>(p.x[kb]) + row * p.ld[kb];
            const float* wk = wc + static_cast<int64_t>(kb) * 32 * ws_k;
#pragma unroll 1
            for (int e = 0; e < 32; ++e) {
                const double xv = xr[e];
                s0 = fma(xv, static_cast<double>(wk[0]), s0);
                s1 = fma(xv, static_cast<double>(wk[ws_n]), s1);
                s2 = fma(xv, static_cast<double>(wk[2 * ws_n]), s2);
                s3 = fma(xv, static_cast<double>(wk[3 * ws_n]), s3);
                wk += ws_k;
            }
        }
        const float* b = bias + 16 * t + 4 * q;
        *reinterpret_cast<float4*>(static_cast<float*>(p.y[t]) + row * p.ldy[t] + 4 * q) =
            make_float4(static_cast<float>(s0 + b[0]), static_cast<float>(s1 + b[1]), static_cast<float>(s2 + b[2]),
                        static_cast<float>(s3 + b[3]));
    }
}

template <int KB, int NT>
__device__ __forceinline__ void split_tile_out(const TallArgs& p, const uint4* frag, const float* bias, int tile, int lane,
                                               const float4 (&cur)[KB][2])
{
    // The six partial products of a 32-column block are summed in accumulators of their own (groups of G tiles) and added to the
    // running sums ONCE per block: a running sum is rounded once per 32 columns instead of six times -- measured 0.6 - 0.8e-7 of
    // sum |x| |w| against 1.3 - 1.7e-7 with the terms added straight in and 2.8 - 3.5e-7 for the fmaf chain (tools/tall_forms_probe.py,
    // profiles/r5t_tall_forms.json).
    constexpr int G = NT % 4 == 0 ? 4 : 2;
    constexpr int kWi[6] = {2, 0, 1, 1, 0, 0}, kXi[6] = {0, 2, 1, 0, 1, 0};      // (w piece, x piece) of the six terms, smallest first
    const int j = lane & 15, q = lane >> 4;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" ::: "memory");            // (keeps the W fragments in LDS: see the exact kernel)
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const float xs[8] = {cur[kb][0].x, cur[kb][0].y, cur[kb][0].z, cur[kb][0].w,
                             cur[kb][1].x, cur[kb][1].y, cur[kb][1].z, cur[kb][1].w};
        uint4 xh4, xm4, xl4;
        split8(xs, xh4, xm4, xl4);
        const bf16x8 xp[3] = {__builtin_bit_cast(bf16x8, xh4), __builtin_bit_cast(bf16x8, xm4), __builtin_bit_cast(bf16x8, xl4)};
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += G) {
            bf16x8 w[G][3];
            f32x4 part[G];
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const uint4* src = frag + ((kb * NT + t0 + u) * 3) * 64 + lane;
                w[u][0] = __builtin_bit_cast(bf16x8, src[0]);
                w[u][1] = __builtin_bit_cast(bf16x8, src[64]);
                w[u][2] = __builtin_bit_cast(bf16x8, src[128]);
                part[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int u = 0; u < G; ++u)
                    part[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][kWi[t]], xp[kXi[t]], part[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < G; ++u) acc[t0 + u] += part[u];
        }
    }
    // Decided here.  Up to 8 output tiles it is acted on BEHIND the stores, which wait for the same sums: a branch in front of them
    // kept the stores and the next tile's loads from being scheduled under the last MFMAs (8 - 14 % on the C3a products,
    // profiles/r6j_configs.json).  With 12 / 16 output tiles the store epilogue is long and holding the decision across it costs
    // more than the branch (K = 64 -> 192: 0.52 against 0.43 ms, profiles/r6_guard_placement.txt): there the tile leaves early.
    const bool redo = any_not_finite<NT>(acc);
    if constexpr (NT >= 12) {
        if (redo) {
            exact_tile_store<KB, NT>(p, bias, tile, j, q);
            return;
        }
    }
    // lane (j, q) holds columns [16 t + 4 q, +4) of row j for every tile t; after the trade lanes j < 8 hold tile 2 m of rows
    // j and j + 8, lanes j >= 8 tile 2 m + 1 of rows j - 8 and j
    const bool upper = j >= 8;
    const int64_t last = p.n_rows - 1;
    int64_t row_a = static_cast<int64_t>(tile) * 16 + (j & 7), row_b = row_a + 8;
    row_a = row_a < last ? row_a : last;
    row_b = row_b < last ? row_b : last;
#pragma unroll
    for (int m = 0; m < NT / 2; ++m) {
        const float4 b0 = *reinterpret_cast<const float4*>(bias + 32 * m + 4 * q);
        const float4 b1 = *reinterpret_cast<const float4*>(bias + 32 * m + 16 + 4 * q);
        const float lo[4] = {acc[2 * m][0] + b0.x, acc[2 * m][1] + b0.y, acc[2 * m][2] + b0.z, acc[2 * m][3] + b0.w};
        const float hi[4] = {acc[2 * m + 1][0] + b1.x, acc[2 * m + 1][1] + b1.y, acc[2 * m + 1][2] + b1.z,
                             acc[2 * m + 1][3] + b1.w};
        float va[4], vb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float lo_far = rotate8(lo[r]), hi_far = rotate8(hi[r]);     // the values of lane j ^ 8
            va[r] = upper ? hi_far : lo[r];
            vb[r] = upper ? hi[r] : lo_far;
        }
        // both tiles' addresses from the (scalar) kernel arguments, then a select: selecting the ARGUMENT per lane turns into a
        // per-lane global load of it, and the wait for that load also waits for the next tile's rows
        float* const lo_a = static_cast<float*>(p.y[2 * m]) + row_a * p.ldy[2 * m] + 4 * q;
        float* const hi_a = static_cast<float*>(p.y[2 * m + 1]) + row_a * p.ldy[2 * m + 1] + 4 * q;
        float* const lo_b = static_cast<float*>(p.y[2 * m]) + row_b * p.ldy[2 * m] + 4 * q;
        float* const hi_b = static_cast<float*>(p.y[2 * m + 1]) + row_b * p.ldy[2 * m + 1] + 4 * q;
        *reinterpret_cast<float4*>(upper ? hi_a : lo_a) = make_float4(va[0], va[1], va[2], va[3]);
        *reinterpret_cast<float4*>(upper ? hi_b : lo_b) = make_float4(vb[0], vb[1], vb[2], vb[3]);
    }
    // the tile again over the split form's rows: the same wavefront's later stores to the same addresses
    if constexpr (NT < 12) {
        if (redo) exact_tile_store<KB, NT>(p, bias, tile, j, q);
    }
}

template <int KB, int NT>
__global__ __launch_bounds__(256) void tall_linear_f32_split_kernel(TallArgs p)
{
    static_assert(NT % 2 == 0, "output tiles are stored in pairs");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* frag = reinterpret_cast<uint4*>(smem);                                          // [KB][NT][3][64] x 8 bf16
    float* bias = reinterpret_cast<float*>(smem + static_cast<size_t>(KB) * NT * 3 * 64 * 16);   // [NT * 16]
    const int tid = threadIdx.x;
    const float* w = static_cast<const float*>(p.w);
    for (int idx = tid; idx < KB * NT * 64; idx += 256) {
        const int lane = idx & 63, t = (idx >> 6) % NT, kb = (idx >> 6) / NT;
        const int64_t n = 16 * t + (lane & 15), k0 = kb * 32 + 8 * (lane >> 4);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.w_t ? w[n * p.ldw + k0 + e] : w[(k0 + e) * p.ldw + n];
        uint4 h, m, l;
        split8(v, h, m, l);
        uint4* dst = frag + ((kb * NT + t) * 3) * 64 + lane;
        dst[0] = h; dst[64] = m; dst[128] = l;
    }
    for (int c = tid; c < NT * 16; c += 256) bias[c] = p.bias ? static_cast<const float*>(p.bias)[c] : 0.f;
    __syncthreads();

    const int lane = tid & 63, j = lane & 15, q = lane >> 4;
    const int n_tiles = (p.n_rows + 15) >> 4;
    const int stride = static_cast<int>(gridDim.x) * 4;
    int tile = static_cast<int>(blockIdx.x) * 4 + (tid >> 6);
    if (tile >= n_tiles) return;
    const int last_tile = n_tiles - 1;
    float4 rows_a[KB][2], rows_b[KB][2];
    split_rows_in<KB>(p, tile, j, q, rows_a);
    split_rows_in<KB>(p, min(tile + stride, last_tile), j, q, rows_b);
#define PYGSD_TALL_STEP(ROWS)                                                          \
    split_tile_out<KB, NT>(p, frag, bias, tile, lane, ROWS);                           \
    split_rows_in<KB>(p, min(tile + 2 * stride, last_tile), j, q, ROWS);               \
    tile += stride;                                                                    \
    if (tile >= n_tiles) return;
    PYGSD_TALL_STEP(rows_a)
    PYGSD_TALL_STEP(rows_b)
    for (;;) {
        PYGSD_TALL_STEP(rows_a)
        PYGSD_TALL_STEP(rows_b)
    }
#undef PYGSD_TALL_STEP
}

template <typename Kern>
int launch_tall(Kern kern, const TallArgs& a, int kb, int nt, hipStream_t s)
{
    const size_t lds = static_cast<size_t>(kb) * nt * 1024 + static_cast<size_t>(nt) * 16 * sizeof(float);
    if (lds > 64 * 1024)
        PYGSD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          static_cast<int>(lds)));
    size_t per_cu = (160 * 1024) / lds;        // resident blocks per CU by LDS
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const int64_t n_tiles = (static_cast<int64_t>(a.n_rows) + 15) / 16;
    int64_t grid = (n_tiles + 3) / 4;
    if (grid > static_cast<int64_t>(256 * per_cu)) grid = static_cast<int64_t>(256 * per_cu);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(256), lds, s, a);
    return check_launch("tall_linear_kernel");
}

// K = 32 kb (bf16) / 16 kb (fp32); f_out = 16 nt; kb * nt <= 64 (64 KB of W fragments per block)
bool shape_ok(int dtype, int k_total, int f_out)
{
    if (dtype == 1) {
        if (k_total % 32 || f_out % 32) return false;
        const int kb = k_total / 32, nt = f_out / 16;
        const bool kb_ok = kb == 1 || kb == 2 || kb == 3 || kb == 4 || kb == 6 || kb == 8;
        const bool nt_ok = nt == 2 || nt == 4 || nt == 6 || nt == 8 || nt == 12 || nt == 16;
        return kb_ok && nt_ok && kb * nt <= 64;
    }
    if (dtype == 0) {
        if (k_total % 16 || f_out % 16) return false;
        const int kb = k_total / 16, nt = f_out / 16;
        const bool kb_ok = kb == 1 || kb == 2 || kb == 4 || kb == 6 || kb == 8 || kb == 12 || kb == 16;
        const bool nt_ok = nt == 1 || nt == 2 || nt == 4 || nt == 6 || nt == 8 || nt == 12 || nt == 16;
        return kb_ok && nt_ok && kb * nt <= 64;
    }
    return false;
}

template <int KB>
int dispatch_bf16(const TallArgs& a, int nt, hipStream_t s)
{
    switch (nt) {
        case 2: return launch_tall(tall_linear_bf16_kernel<KB, 2>, a, KB, 2, s);
        case 4: return launch_tall(tall_linear_bf16_kernel<KB, 4>, a, KB, 4, s);
        case 6: return launch_tall(tall_linear_bf16_kernel<KB, 6>, a, KB, 6, s);
        case 8: return launch_tall(tall_linear_bf16_kernel<KB, 8>, a, KB, 8, s);
        case 12: if constexpr (KB <= 4) return launch_tall(tall_linear_bf16_kernel<KB, 12>, a, KB, 12, s); break;
        case 16: if constexpr (KB <= 4) return launch_tall(tall_linear_bf16_kernel<KB, 16>, a, KB, 16, s); break;
        default: break;
    }
    return fail("pygsd_tall_linear: unsupported bf16 shape (%d k-blocks x %d tiles)", KB, nt);
}

template <int KB>
int dispatch_f32(const TallArgs& a, int nt, hipStream_t s)
{
    switch (nt) {
        case 1: return launch_tall(tall_linear_f32_kernel<KB, 1>, a, KB, 1, s);
        case 2: return launch_tall(tall_linear_f32_kernel<KB, 2>, a, KB, 2, s);
        case 4: return launch_tall(tall_linear_f32_kernel<KB, 4>, a, KB, 4, s);
        case 6: if constexpr (KB <= 8) return launch_tall(tall_linear_f32_kernel<KB, 6>, a, KB, 6, s); break;
        case 8: if constexpr (KB <= 8) return launch_tall(tall_linear_f32_kernel<KB, 8>, a, KB, 8, s); break;
        case 12: if constexpr (KB <= 4) return launch_tall(tall_linear_f32_kernel<KB, 12>, a, KB, 12, s); break;
        case 16: if constexpr (KB <= 4) return launch_tall(tall_linear_f32_kernel<KB, 16>, a, KB, 16, s); break;
        default: break;
    }
    return fail("pygsd_tall_linear: unsupported fp32 shape (%d k-blocks x %d tiles)", KB, nt);
}

// the split kernel's shapes: K and f_out multiples of 32, 3 KB of LDS per (k-block, tile) pair
bool split_shape_ok(int k_total, int f_out)
{
    if (k_total % 32 || f_out % 32) return false;
    const int kb = k_total / 32, nt = f_out / 16;
    const bool kb_ok = kb == 1 || kb == 2 || kb == 3 || kb == 4 || kb == 6 || kb == 8;
    const bool nt_ok = nt == 2 || nt == 4 || nt == 6 || nt == 8 || nt == 12 || nt == 16;
    return kb_ok && nt_ok && kb * nt <= 48;
}

bool split_allowed() { return tall_f32_form().load() == 0; }

template <typename Kern>
int launch_split(Kern kern, const TallArgs& a, int kb, int nt, hipStream_t s)
{
    const size_t lds = static_cast<size_t>(kb) * nt * 3072 + static_cast<size_t>(nt) * 16 * sizeof(float);
    if (lds > 64 * 1024)
        PYGSD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          static_cast<int>(lds)));
    size_t per_cu = (160 * 1024) / lds;
    if (per_cu > 8) per_cu = 8;
    if (per_cu < 1) per_cu = 1;
    const int64_t n_tiles = (static_cast<int64_t>(a.n_rows) + 15) / 16;
    int64_t grid = (n_tiles + 3) / 4;
    if (grid > static_cast<int64_t>(256 * per_cu)) grid = static_cast<int64_t>(256 * per_cu);
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(grid)), dim3(256), lds, s, a);
    return check_launch("tall_linear_f32_split_kernel");
}

template <int KB>
int dispatch_split(const TallArgs& a, int nt, hipStream_t s)
{
    switch (nt) {
        case 2: return launch_split(tall_linear_f32_split_kernel<KB, 2>, a, KB, 2, s);
        case 4: return launch_split(tall_linear_f32_split_kernel<KB, 4>, a, KB, 4, s);
        case 6: return launch_split(tall_linear_f32_split_kernel<KB, 6>, a, KB, 6, s);
        case 8: if constexpr (KB <= 6) return launch_split(tall_linear_f32_split_kernel<KB, 8>, a, KB, 8, s); break;
        case 12: if constexpr (KB <= 4) return launch_split(tall_linear_f32_split_kernel<KB, 12>, a, KB, 12, s); break;
        case 16: if constexpr (KB <= 3) return launch_split(tall_linear_f32_split_kernel<KB, 16>, a, KB, 16, s); break;
        default: break;
    }
    return fail("pygsd_tall_linear: unsupported split shape (%d k-blocks x %d tiles)", KB, nt);
}

// ---- column sums -------------------------------------------------------------------------------
struct ColumnSumArgs {
    const void* x;
    int64_t ldx, n_rows;
    int32_t f;
    float* partial;      // [gridDim.x][f]
};

// 16-byte loads (8 bf16 / 4 fp32 columns per thread, f / that threads per row, as many rows per pass as fit 256 threads);
// per-thread fp32 sums, combined per block through LDS -> one partial row per block; fixed order (deterministic).
template <bool BF16>
__global__ __launch_bounds__(256) void column_sums_kernel(ColumnSumArgs p)
{
    constexpr int V = BF16 ? 8 : 4;
    __shared__ float sm[256 * V];
    const int tid = threadIdx.x;
    const int tpr = p.f / V, rpp = 256 / tpr;
    const int cv = tid % tpr, rl = tid / tpr;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    if (rl < rpp) {
        for (int64_t row = static_cast<int64_t>(blockIdx.x) * rpp + rl; row < p.n_rows;
             row += static_cast<int64_t>(gridDim.x) * rpp) {
            if constexpr (BF16) {
                const uint4 u = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p.x) + row * p.ldx + cv * V);
                acc[0] += bf16_value(u.x & 0xffffu); acc[1] += bf16_value(u.x >> 16);
                acc[2] += bf16_value(u.y & 0xffffu); acc[3] += bf16_value(u.y >> 16);
                acc[4] += bf16_value(u.z & 0xffffu); acc[5] += bf16_value(u.z >> 16);
                acc[6] += bf16_value(u.w & 0xffffu); acc[7] += bf16_value(u.w >> 16);
            } else {
                const float4 u = *reinterpret_cast<const float4*>(static_cast<const float*>(p.x) + row * p.ldx + cv * V);
                acc[0] += u.x; acc[1] += u.y; acc[2] += u.z; acc[3] += u.w;
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) sm[rl * p.f + cv * V + v] = acc[v];
    }
    __syncthreads();
    for (int c = tid; c < p.f; c += 256) {
        float s = 0.f;
        for (int r = 0; r < rpp; ++r) s += sm[r * p.f + c];
        p.partial[static_cast<int64_t>(blockIdx.x) * p.f + c] = s;
    }
}

// out[c] = sum_b partial[b][c]: 16 columns per block, 16 groups of partial rows per column combined through LDS in group
// order (one 64-column block with 4 groups walked 256 dependent loads per thread: 60 us for 1024 x 64 partials, three times
// the pass that produced them -- profiles/r4b_kernel_stats_C3a.csv)
__global__ __launch_bounds__(256) void column_sums_finish_kernel(const float* __restrict__ partial, int n_partials, int f,
                                                                 float* __restrict__ out)
{
    __shared__ float sm[256];
    const int tid = threadIdx.x;
    const int c = static_cast<int>(blockIdx.x) * 16 + (tid & 15), grp = tid >> 4;
    float acc = 0.f;
    if (c < f) {
#pragma unroll 4
        for (int b = grp; b < n_partials; b += 16) acc += partial[static_cast<int64_t>(b) * f + c];
    }
    sm[tid] = acc;
    __syncthreads();
    if (grp == 0 && c < f) {
        float total = sm[tid];
#pragma unroll
        for (int g = 1; g < 16; ++g) total += sm[tid + 16 * g];
        out[c] = total;
    }
}

unsigned column_sum_blocks(int64_t n_rows, int f, int v)
{
    const int rpp = 256 / (f / v);
    int64_t b = (n_rows + rpp - 1) / rpp;
    if (b > 1024) b = 1024;                     // (512 blocks streamed at 0.52 of the peak, 1024 at 0.80: profiles/r4d vs r4b)
    return static_cast<unsigned>(b < 1 ? 1 : b);
}
}  // namespace

// 0 = the split form wherever its shapes allow (default), 1 = every fp32 product as an fmaf chain on v_mfma_f32_16x16x4_f32;
// PYGSD_TALL_F32=exact sets 1 at load, pygsd_tall_f32_form changes it at run time (measurement / bitwise tests)
// (an atomic: a thread may flip it while another is inside a launch; every entry point reads it once)
std::atomic<int>& tall_f32_form()
{
    static std::atomic<int> form{[] {
        const char* e = getenv("PYGSD_TALL_F32");
        return (e && e[0] == 'e') ? 1 : 0;
    }()};
    return form;
}
}  // namespace pygsd

using namespace pygsd;

extern "C" int pygsd_tall_f32_form(int32_t form)
{
    std::atomic<int>& cur = tall_f32_form();
    if (form == 0 || form == 1) return cur.exchange(form);
    return cur.load();
}

extern "C" int pygsd_tall_linear_supported(int32_t dtype, int32_t k_total, int32_t f_out)
{
    return (k_total > 0 && f_out > 0 && shape_ok(dtype, k_total, f_out)) ? 1 : 0;
}

extern "C" int pygsd_tall_linear(const void* const* xs, const int64_t* ldx, const int32_t* widths, int32_t n_seg,
                                 const void* w, int64_t ldw, int32_t w_transposed, const void* bias, void* const* ys,
                                 const int64_t* ldy, const int32_t* out_widths, int32_t n_out, int64_t n_rows, int32_t dtype,
                                 void* stream)
{
    PYGSD_REQUIRE(dtype == 0 || dtype == 1, "pygsd_tall_linear: dtype must be 0 (fp32) or 1 (bf16), got %d", dtype);
    PYGSD_REQUIRE(n_seg >= 1 && n_seg <= kMaxSeg && xs && ldx && widths, "pygsd_tall_linear: 1..%d input segments", kMaxSeg);
    PYGSD_REQUIRE(n_out >= 1 && n_out <= kMaxOut && ys && ldy && out_widths, "pygsd_tall_linear: 1..%d output segments",
                  kMaxOut);
    PYGSD_REQUIRE(n_rows >= 0 && n_rows < (1ll << 31) - 16, "pygsd_tall_linear: row count outside [0, 2^31)");
    const int kw = dtype == 1 ? 32 : 16;          // columns per block (inputs and outputs alike)
    const int vec = dtype == 1 ? 8 : 4;           // elements per 16 bytes
    const size_t esz = dtype == 1 ? 2 : 4;
    int k_total = 0, f_out = 0;
    for (int g = 0; g < n_seg; ++g) {
        PYGSD_REQUIRE(widths[g] > 0 && widths[g] % kw == 0, "pygsd_tall_linear: input segment %d is %d columns wide "
                      "(multiples of %d)", g, widths[g], kw);
        k_total += widths[g];
    }
    for (int g = 0; g < n_out; ++g) {
        PYGSD_REQUIRE(out_widths[g] > 0 && out_widths[g] % kw == 0, "pygsd_tall_linear: output segment %d is %d columns wide "
                      "(multiples of %d)", g, out_widths[g], kw);
        f_out += out_widths[g];
    }
    PYGSD_REQUIRE(shape_ok(dtype, k_total, f_out), "pygsd_tall_linear: unsupported shape K=%d f_out=%d (see "
                  "pygsd_tall_linear_supported)", k_total, f_out);
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(w && ldw >= (w_transposed ? k_total : f_out), "pygsd_tall_linear: W null or ldw = %lld too small",
                  static_cast<long long>(ldw));
    // fp32: the split form (bf16 matrix pipe) where its shapes allow -- every input segment whole 32-column blocks
    bool split = dtype == 0 && split_allowed() && split_shape_ok(k_total, f_out);
    for (int g = 0; split && g < n_seg; ++g) split = widths[g] % 32 == 0;
    const int kw_in = split ? 32 : kw;          // columns per input block
    TallArgs a{};
    int kb = 0;
    for (int g = 0; g < n_seg; ++g) {
        PYGSD_REQUIRE(xs[g] && aligned16(xs[g]) && ldx[g] >= widths[g] && ldx[g] % vec == 0,
                      "pygsd_tall_linear: segment %d null, not 16-byte aligned, or row stride %lld not a multiple of 16 bytes "
                      ">= its width", g, static_cast<long long>(ldx[g]));
        for (int c = 0; c < widths[g]; c += kw_in, ++kb) {
            a.x[kb] = static_cast<const unsigned char*>(xs[g]) + static_cast<size_t>(c) * esz;
            a.ld[kb] = ldx[g];
        }
    }
    int ob = 0;
    for (int g = 0; g < n_out; ++g) {
        PYGSD_REQUIRE(ys[g] && aligned16(ys[g]) && ldy[g] >= out_widths[g] && ldy[g] % vec == 0,
                      "pygsd_tall_linear: output segment %d null, not 16-byte aligned, or row stride %lld not a multiple of "
                      "16 bytes >= its width", g, static_cast<long long>(ldy[g]));
        for (int c = 0; c < out_widths[g]; c += kw, ++ob) {
            a.y[ob] = static_cast<unsigned char*>(ys[g]) + static_cast<size_t>(c) * esz;
            a.ldy[ob] = ldy[g];
        }
    }
    a.w = w; a.ldw = ldw; a.w_t = w_transposed ? 1 : 0; a.bias = bias;
    a.n_rows = static_cast<int32_t>(n_rows); a.f_out = f_out;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_DENSE, s);
    const int nt = f_out / 16;
    if (split) {
        switch (kb) {
            case 1: return dispatch_split<1>(a, nt, s);
            case 2: return dispatch_split<2>(a, nt, s);
            case 3: return dispatch_split<3>(a, nt, s);
            case 4: return dispatch_split<4>(a, nt, s);
            case 6: return dispatch_split<6>(a, nt, s);
            case 8: return dispatch_split<8>(a, nt, s);
            default: break;
        }
        return fail("pygsd_tall_linear: unsupported split shape K=%d f_out=%d", k_total, f_out);
    }
    if (dtype == 1) {
        switch (kb) {
            case 1: return dispatch_bf16<1>(a, nt, s);
            case 2: return dispatch_bf16<2>(a, nt, s);
            case 3: return dispatch_bf16<3>(a, nt, s);
            case 4: return dispatch_bf16<4>(a, nt, s);
            case 6: return dispatch_bf16<6>(a, nt, s);
            case 8: return dispatch_bf16<8>(a, nt, s);
            default: break;
        }
    } else {
        switch (kb) {
            case 1: return dispatch_f32<1>(a, nt, s);
            case 2: return dispatch_f32<2>(a, nt, s);
            case 4: return dispatch_f32<4>(a, nt, s);
            case 6: return dispatch_f32<6>(a, nt, s);
            case 8: return dispatch_f32<8>(a, nt, s);
            case 12: return dispatch_f32<12>(a, nt, s);
            case 16: return dispatch_f32<16>(a, nt, s);
            default: break;
        }
    }
    return fail("pygsd_tall_linear: unsupported shape K=%d f_out=%d", k_total, f_out);
}

extern "C" int pygsd_column_sums_workspace(int64_t n_rows, int32_t f, int32_t dtype, size_t* bytes)
{
    PYGSD_REQUIRE(bytes, "pygsd_column_sums_workspace: null output");
    PYGSD_REQUIRE(dtype == 0 || dtype == 1, "pygsd_column_sums_workspace: dtype must be 0 (fp32) or 1 (bf16)");
    const int v = dtype == 1 ? 8 : 4;
    PYGSD_REQUIRE(f > 0 && f % v == 0 && f / v <= 256, "pygsd_column_sums_workspace: f = %d must be a multiple of %d, at most %d",
                  f, v, 256 * v);
    *bytes = static_cast<size_t>(column_sum_blocks(n_rows < 0 ? 0 : n_rows, f, v)) * f * sizeof(float);
    return 0;
}

extern "C" int pygsd_column_sums(const void* x, int64_t ldx, int64_t n_rows, int32_t f, int32_t dtype, float* out,
                                 void* workspace, size_t workspace_bytes, void* stream)
{
    size_t need = 0;
    if (int rc = pygsd_column_sums_workspace(n_rows, f, dtype, &need)) return rc;
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_column_sums: negative size");
    PYGSD_REQUIRE(out && workspace && workspace_bytes >= need, "pygsd_column_sums: null output or workspace too small "
                  "(%zu < %zu)", workspace_bytes, need);
    const int v = dtype == 1 ? 8 : 4;
    PYGSD_REQUIRE(n_rows == 0 || (x && aligned16(x) && ldx >= f && ldx % v == 0),
                  "pygsd_column_sums: input null, not 16-byte aligned, or row stride not a multiple of 16 bytes >= f");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    const unsigned blocks = column_sum_blocks(n_rows, f, v);
    ColumnSumArgs a{x, ldx, n_rows, f, static_cast<float*>(workspace)};
    if (dtype == 1)
        hipLaunchKernelGGL(column_sums_kernel<true>, dim3(blocks), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL(column_sums_kernel<false>, dim3(blocks), dim3(256), 0, s, a);
    if (int rc = check_launch("column_sums_kernel")) return rc;
    hipLaunchKernelGGL(column_sums_finish_kernel, dim3((static_cast<unsigned>(f) + 15u) / 16u), dim3(256), 0, s,
                       static_cast<const float*>(workspace), static_cast<int>(blocks), f, out);
    return check_launch("column_sums_finish_kernel");
}
