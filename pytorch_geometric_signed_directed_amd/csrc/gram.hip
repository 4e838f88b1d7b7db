// Weight gradients of the tall linear maps on the matrix cores:  dW[K, F] = [X_0 | X_1 | ...]^T [G_0 | G_1 | ...]
// for tall operands (n ~ 10^5..10^7 rows, K and F a few tens to a few hundreds of columns): the reduction runs over the
// ROWS, i.e. both MFMA operands are needed column-major while they sit row-major in HBM.
//
//   x^T g   DiGCNConv.py:66 (x W), DiGCN_Inception_Block.py:44-46, SGCNConv.py:121-126 (the Linear over
//           [aggregated | own]) -- what autograd's mm backward computes for the reference; library GEMMs ran these
//           K = 10^6-deep, 64 x 128 products on a handful of CUs (split-K bmm + a reduction pass: 0.19 of the 1.22 ms
//           SGCNConv step, 0.33 of the 5.88 ms inception block)
//
// One wavefront owns a tile of 16 (fp32) / 32 (bf16) rows.  Every global load is a coalesced 16-byte row load (a
// row's columns sit on adjacent lanes); the tile goes through a wavefront-private LDS image once and comes back as
// column fragments:
//   fp32: image [16][cols + 4] (ds_write_b128 rows, conflict-free ds_read_b32 columns), v_mfma_f32_16x16x4_f32 (exact)
//   bf16: image [col block][32 rows][16 cols] (32-byte rows), read with ds_read_b64_tr_b16 -- the LDS transpose read
//         of gfx950: lane (i, q) of a 16-lane group receives column i of rows 4 q .. 4 q + 3 of the group's block --
//         v_mfma_f32_16x16x32_bf16, fp32 accumulation.  The row <-> k-slot assignment of an MFMA is free (a sum), so the
//         two reads of an operand take rows 4 q + j and 16 + 4 q + j: a 32-lane half then touches 8 consecutive 32-byte
//         rows = all 64 banks once.
// The operands are cut into column CHUNKS (host side, along segment boundaries: X 16 / 32 / 64 columns, G up to 128); block
// (chunk pair, x) accumulates the [chunk cx of X]^T [chunk cg of G] block of dW over a persistent row loop in MFMA registers,
// the four wavefronts of a block are combined through LDS in wave order, one partial per block, and a second kernel adds
// the partials in block order: no atomics, deterministic.
#include "common.hpp"

namespace pygsd {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kMaxChunks = 16;
constexpr int kNoPart = 1 << 20;      // GramChunk::c1 / c2 of a chunk without that part

// A chunk is up to three PARTS (round 6): column segments that are neighbours in dW but separate matrices in memory -- [g | g_a] of
// SGCNConv, [dx0 | dP] of the inception block (64 + 128 columns, different row strides) -- ride in ONE chunk, so that the other
// operand's rows are read once for all of them instead of once per segment (measured before: 1.38x / 1.27x the algorithmic bytes
// at C3a / C5b, profiles/r5v_configs.json).
struct GramChunk {
    const void* p;     // first element of the chunk (of its first part) in row 0
    int64_t ld;        // row stride in elements (of the first part)
    int32_t tiles;     // 16-column tiles: 1, 2, 4 (X and G), 8 or 12 (G)
    int32_t at;        // first row (X chunks) / column (G chunks) of this chunk in dW
    // parts 1 and 2 start at COLUMN c1 / c2 of the chunk (a value past the chunk's width: no such part); their first element in
    // row 0 sits d1 / d2 ELEMENTS from `p` and their row stride is ld + dl1 / ld + dl2.  (Offsets and differences, not pointers
    // and strides: a per-lane choice among struct fields is compiled into an indexed read of a private copy of the struct.)
    int64_t d1, d2, dl1, dl2;
    int32_t c1, c2;
};

struct GramArgs {
    GramChunk x[kMaxChunks];
    GramChunk g[kMaxChunks];
    float* partial;          // [gridDim.y][k_total * f_total]
    int64_t n_rows;
    int32_t k_total, f_total, n_g;
};

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- fp32 ---------------------------------------------------------------------------------------
// rows of one 16-row tile -> the wavefront's image: NT float4 loads per lane, a row's 16 NT floats on 4 NT adjacent lanes
template <int NT>
__device__ __forceinline__ void load_rows_f32(const GramChunk& c, int64_t r0, int64_t n_rows, int lane, float4 (&v)[NT])
{
    constexpr int LPR = NT * 4;            // 16-byte pieces (lanes) per row
    const float* base = static_cast<const float*>(c.p);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int piece = t * 64 + lane;                      // piece-major over the 16 x LPR pieces of the tile
        const int row = piece / LPR;
        const int col = (piece % LPR) * 4;
        int64_t part = (col >= c.c1) ? c.d1 - c.c1 : 0;       // (chunks of several parts only)
        int64_t ld = c.ld + ((col >= c.c1) ? c.dl1 : 0);
        part = (col >= c.c2) ? c.d2 - c.c2 : part;
        ld = (col >= c.c2) ? c.ld + c.dl2 : ld;
        v[t] = make_float4(0.f, 0.f, 0.f, 0.f);               // rows past the end contribute nothing
        if (r0 + row < n_rows) v[t] = *reinterpret_cast<const float4*>(base + (r0 + row) * ld + col + part);
    }
}

template <int NT>
__device__ __forceinline__ void store_rows_f32(const float4 (&v)[NT], int lane, float* img, int rs)
{
    constexpr int LPR = NT * 4;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int piece = t * 64 + lane;
        *reinterpret_cast<float4*>(img + (piece / LPR) * rs + (piece % LPR) * 4) = v[t];
    }
}

template <int NTK, int NTF>
__device__ __forceinline__ void gram_block_f32(const GramArgs& p, const GramChunk& cx, const GramChunk& cg, float* lds)
{
    constexpr int rsx = NTK * 16 + 4, rsg = NTF * 16 + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    float* imx = lds + wave * 16 * (rsx + rsg);
    float* img = imx + 16 * rsx;
    f32x4 acc[NTK][NTF];
#pragma unroll
    for (int a = 0; a < NTK; ++a)
#pragma unroll
        for (int b = 0; b < NTF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t n_tiles = (p.n_rows + 15) >> 4;
    const int64_t stride = static_cast<int64_t>(gridDim.y) * 4;
    int64_t tile = static_cast<int64_t>(blockIdx.y) * 4 + wave;
    // the NEXT tile's rows are in flight (in registers) while the current tile is multiplied out of the LDS image
    float4 rx[NTK], rg[NTF];
    if (tile < n_tiles) {
        load_rows_f32<NTK>(cx, tile << 4, p.n_rows, lane, rx);
        load_rows_f32<NTF>(cg, tile << 4, p.n_rows, lane, rg);
    }
    for (; tile < n_tiles; tile += stride) {
        store_rows_f32<NTK>(rx, lane, imx, rsx);
        store_rows_f32<NTF>(rg, lane, img, rsg);
        if (tile + stride < n_tiles) {
            load_rows_f32<NTK>(cx, (tile + stride) << 4, p.n_rows, lane, rx);
            load_rows_f32<NTF>(cg, (tile + stride) << 4, p.n_rows, lane, rg);
        }
        wave_sync();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            // MFMA k-slot q of step s <-> tile row 4 q + s; lane (i, q) supplies column i of both operands
            float xa[NTK], gb[NTF];
#pragma unroll
            for (int a = 0; a < NTK; ++a) xa[a] = imx[(4 * q + s) * rsx + a * 16 + i];
#pragma unroll
            for (int b = 0; b < NTF; ++b) gb[b] = img[(4 * q + s) * rsg + b * 16 + i];
#pragma unroll
            for (int a = 0; a < NTK; ++a)
#pragma unroll
                for (int b = 0; b < NTF; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[a], gb[b], acc[a][b], 0, 0, 0);
        }
        wave_sync();
    }
    // C/D layout: lane (i, q), register r -> dW[16 a + 4 q + r][16 b + i]; the four wavefronts add in wave order
    __syncthreads();
    constexpr int fo = NTF * 16;
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int a = 0; a < NTK; ++a)
#pragma unroll
                for (int b = 0; b < NTF; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = (a * 16 + 4 * q + r) * fo + b * 16 + i;
                        lds[e] = (turn == 0 ? 0.f : lds[e]) + acc[a][b][r];
                    }
        }
        __syncthreads();
    }
    float* part = p.partial + static_cast<int64_t>(blockIdx.y) * p.k_total * p.f_total;
    for (int e = tid; e < NTK * 16 * fo; e += 256) {
        const int row = e / fo, col = e - row * fo;
        part[static_cast<int64_t>(cx.at + row) * p.f_total + cg.at + col] = lds[e];
    }
}

template <int NTK>
__device__ __forceinline__ void gram_pick_f32(const GramArgs& p, const GramChunk& cx, const GramChunk& cg, float* lds)
{
    if (cg.tiles == 8) gram_block_f32<NTK, 8>(p, cx, cg, lds);
    else if (cg.tiles == 4) gram_block_f32<NTK, 4>(p, cx, cg, lds);
    else if (cg.tiles == 2) gram_block_f32<NTK, 2>(p, cx, cg, lds);
    else gram_block_f32<NTK, 1>(p, cx, cg, lds);
}

__global__ __launch_bounds__(256, 2) void tall_gram_f32_kernel(GramArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float lds_f32[];
    const GramChunk cx = p.x[blockIdx.x / p.n_g], cg = p.g[blockIdx.x % p.n_g];
    if (cx.tiles == 4) gram_pick_f32<4>(p, cx, cg, lds_f32);
    else if (cx.tiles == 2) gram_pick_f32<2>(p, cx, cg, lds_f32);
    else gram_pick_f32<1>(p, cx, cg, lds_f32);
}

// ---- bf16 storage, fp32 accumulation ------------------------------------------------------------
// rows of one 32-row tile -> image [tile][32 rows][16 cols]: NT 16-byte loads per lane, a row's 16 NT columns on 2 NT lanes
// DL: parts may have row strides of their own (the one-wavefront-per-SIMD instance only: the stride select costs the two-wave
// instance the registers it does not have -- 3 spills)
template <int NT, bool DL>
__device__ __forceinline__ void load_rows_bf16(const GramChunk& c, int64_t r0, int64_t n_rows, int lane, uint4 (&v)[NT])
{
    constexpr int LPR = NT * 2;            // 16-byte pieces (lanes) per row
    const uint16_t* base = static_cast<const uint16_t*>(c.p);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int piece = t * 64 + lane;                      // piece-major over the 32 x LPR pieces of the tile
        const int row = piece / LPR, col = (piece % LPR) * 8; // 8-column piece of the row
        int64_t part = (col >= c.c1) ? c.d1 - c.c1 : 0;       // (chunks of several parts only; see GramChunk)
        part = (col >= c.c2) ? c.d2 - c.c2 : part;
        int64_t ld = c.ld;
        if constexpr (DL) {
            ld = c.ld + ((col >= c.c1) ? c.dl1 : 0);
            ld = (col >= c.c2) ? c.ld + c.dl2 : ld;
        }
        v[t] = make_uint4(0u, 0u, 0u, 0u);
        if (r0 + row < n_rows) v[t] = *reinterpret_cast<const uint4*>(base + (r0 + row) * ld + col + part);
    }
}

template <int NT>
__device__ __forceinline__ void store_rows_bf16(const uint4 (&v)[NT], int lane, unsigned char* img)
{
    constexpr int LPR = NT * 2;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int piece = t * 64 + lane;
        const int row = piece / LPR, c8 = piece % LPR;
        *reinterpret_cast<uint4*>(img + (c8 >> 1) * 1024 + row * 32 + (c8 & 1) * 16) = v[t];
    }
}

// column i (= lane & 15) of rows {4 q + j} and {16 + 4 q + j}, j < 4, of N consecutive [32][16] blocks: the 8 k-slots of lane
// (i, q) for each.  All 2 N transpose reads are issued back to back; ONE s_waitcnt covers them (the empty asm statements tie
// every result register to that wait, so that no use can be scheduled in front of it).
template <int N>
__device__ __forceinline__ void column_fragments(uint32_t first_block_addr, int lane, bf16x8 (&out)[N])
{
    const int t = lane & 15, q = lane >> 4;
    const uint32_t a = first_block_addr + static_cast<uint32_t>((4 * q + (t >> 2)) * 32 + (t & 3) * 8);
    uint64_t lo[N], hi[N];
#pragma unroll
    for (int b = 0; b < N; ++b)
        asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:512"
                     : "=&v"(lo[b]), "=&v"(hi[b])
                     : "v"(a + static_cast<uint32_t>(b) * 1024u)
                     : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int b = 0; b < N; ++b) {
        asm volatile("" : "+v"(lo[b]), "+v"(hi[b]));
        const uint4 u = make_uint4(static_cast<uint32_t>(lo[b]), static_cast<uint32_t>(lo[b] >> 32), static_cast<uint32_t>(hi[b]),
                                   static_cast<uint32_t>(hi[b] >> 32));
        out[b] = __builtin_bit_cast(bf16x8, u);
    }
}

template <int NTK, int NTF, bool WIDE = false>
__device__ __forceinline__ void gram_block_bf16(const GramArgs& p, const GramChunk& cx, const GramChunk& cg, unsigned char* lds)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 15, q = lane >> 4;
    unsigned char* imx = lds + wave * (NTK + NTF) * 1024;
    unsigned char* img = imx + NTK * 1024;
    const uint32_t ax = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(imx));      // LDS offsets: the low 32 bits
    const uint32_t ag = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(img));
    f32x4 acc[NTK][NTF];
#pragma unroll
    for (int a = 0; a < NTK; ++a)
#pragma unroll
        for (int b = 0; b < NTF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t n_tiles = (p.n_rows + 31) >> 5;
    const int64_t stride = static_cast<int64_t>(gridDim.y) * 4;
    int64_t tile = static_cast<int64_t>(blockIdx.y) * 4 + wave;
    uint4 rx[NTK], rg[NTF];                       // the next tile's rows, in flight while the current one is multiplied
    if (tile < n_tiles) {
        load_rows_bf16<NTK, false>(cx, tile << 5, p.n_rows, lane, rx);
        load_rows_bf16<NTF, WIDE>(cg, tile << 5, p.n_rows, lane, rg);
    }
    for (; tile < n_tiles; tile += stride) {
        store_rows_bf16<NTK>(rx, lane, imx);
        store_rows_bf16<NTF>(rg, lane, img);
        if (tile + stride < n_tiles) {
            load_rows_bf16<NTK, false>(cx, (tile + stride) << 5, p.n_rows, lane, rx);
            load_rows_bf16<NTF, WIDE>(cg, (tile + stride) << 5, p.n_rows, lane, rg);
        }
        wave_sync();
        bf16x8 xa[NTK];
        column_fragments<NTK>(ax, lane, xa);
        constexpr int GB = NTF < 4 ? NTF : 4;          // G fragments live at a time (registers: the accumulators take 16 NTK NTF)
#pragma unroll
        for (int b0 = 0; b0 < NTF; b0 += GB) {
            bf16x8 gb[GB];
            column_fragments<GB>(ag + b0 * 1024, lane, gb);
#pragma unroll
            for (int a = 0; a < NTK; ++a)
#pragma unroll
                for (int b = 0; b < GB; ++b)
                    acc[a][b0 + b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[a], gb[b], acc[a][b0 + b], 0, 0, 0);
        }
        wave_sync();
    }
    __syncthreads();
    float* sum = reinterpret_cast<float*>(lds);
    constexpr int fo = NTF * 16;
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int a = 0; a < NTK; ++a)
#pragma unroll
                for (int b = 0; b < NTF; ++b)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = (a * 16 + 4 * q + r) * fo + b * 16 + i;
                        sum[e] = (turn == 0 ? 0.f : sum[e]) + acc[a][b][r];
                    }
        }
        __syncthreads();
    }
    float* part = p.partial + static_cast<int64_t>(blockIdx.y) * p.k_total * p.f_total;
    for (int e = tid; e < NTK * 16 * fo; e += 256) {
        const int row = e / fo, col = e - row * fo;
        part[static_cast<int64_t>(cx.at + row) * p.f_total + cg.at + col] = sum[e];
    }
}

template <int NTK, bool WIDE>
__device__ __forceinline__ void gram_pick_bf16(const GramArgs& p, const GramChunk& cx, const GramChunk& cg, unsigned char* lds)
{
    if constexpr (WIDE) {
        if (cg.tiles == 12) {
            gram_block_bf16<NTK, 12, true>(p, cx, cg, lds);
            return;
        }
    }
    if (cg.tiles == 8) gram_block_bf16<NTK, 8, WIDE>(p, cx, cg, lds);
    else if (cg.tiles == 4) gram_block_bf16<NTK, 4, WIDE>(p, cx, cg, lds);
    else if (cg.tiles == 2) gram_block_bf16<NTK, 2, WIDE>(p, cx, cg, lds);
    else gram_block_bf16<NTK, 1, WIDE>(p, cx, cg, lds);
}

__global__ __launch_bounds__(256, 2) void tall_gram_bf16_kernel(GramArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_b16[];
    const GramChunk cx = p.x[blockIdx.x / p.n_g], cg = p.g[blockIdx.x % p.n_g];
    if (cx.tiles == 4) gram_pick_bf16<4, false>(p, cx, cg, lds_b16);
    else if (cx.tiles == 2) gram_pick_bf16<2, false>(p, cx, cg, lds_b16);
    else gram_pick_bf16<1, false>(p, cx, cg, lds_b16);
}

// the same with G chunks of 12 tiles (three 64-column parts: the inception block's [dx0 | dP_1 | dP_2]): 192 accumulator registers
// against 64 columns of X -- one wavefront per SIMD, every operand row read ONCE
__global__ __launch_bounds__(256, 1) void tall_gram_bf16_wide_kernel(GramArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_b16w[];
    const GramChunk cx = p.x[blockIdx.x / p.n_g], cg = p.g[blockIdx.x % p.n_g];
    if (cx.tiles == 4) gram_pick_bf16<4, true>(p, cx, cg, lds_b16w);
    else if (cx.tiles == 2) gram_pick_bf16<2, true>(p, cx, cg, lds_b16w);
    else gram_pick_bf16<1, true>(p, cx, cg, lds_b16w);
}

// ---- fp32, round 5: no transposition at all ------------------------------------------------------------------------------
// v_mfma_f32_32x32x2_f32 takes ONE value per lane for each operand: lane l supplies A[l % 32][l / 32] and B[l / 32][l % 32].  For
// dW = X^T G the reduction index k is the ROW, so the A operand of the step over rows (r, r + 1) is X[r + l / 32][c0 + l % 32] --
// 32 consecutive floats of one row on 32 consecutive lanes -- and the B operand the same read of G: both MFMA operands ARE
// coalesced global loads (a 128-byte line per half wavefront, every byte used), and the LDS round trip of the 16x16x4 form above
// (rows in, columns out: 2 wavefront barriers and 12 ds_reads per 32 MFMAs) is gone.  A wavefront keeps the WHOLE [K, F] block
// of its block pair in registers (NBK x NBF accumulators of 16 registers, <= 12: 192 of the 512 a wavefront may hold at one
// wavefront per SIMD), so every operand element is fetched exactly once per pass: X is no longer re-read per 128-column chunk of
// G (1.26 - 1.38 x the algorithmic bytes before, profiles/r4j_configs.json).  The loads of the next D row pairs are in flight
// while the current D are multiplied (exact fp32, an fmaf chain over the rows in order).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// eight fp32 values -> their (hi, mid, lo) bf16 pieces, round to nearest even; x - hi and x - hi - mid are exact (csrc/tall.hip)
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8 (&out)[3])
{
    uint32_t hh[4], mm[4], ll[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = x[2 * e], b = x[2 * e + 1];
        const f32x2 v0 = {a, b};
        hh[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v0, bf16x2));
        const float ra = a - __uint_as_float(hh[e] << 16), rb = b - __uint_as_float(hh[e] & 0xffff0000u);
        const f32x2 v1 = {ra, rb};
        mm[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v1, bf16x2));
        const float sa = ra - __uint_as_float(mm[e] << 16), sb = rb - __uint_as_float(mm[e] & 0xffff0000u);
        const f32x2 v2 = {sa, sb};
        ll[e] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v2, bf16x2));
    }
    out[0] = __builtin_bit_cast(bf16x8, make_uint4(hh[0], hh[1], hh[2], hh[3]));
    out[1] = __builtin_bit_cast(bf16x8, make_uint4(mm[0], mm[1], mm[2], mm[3]));
    out[2] = __builtin_bit_cast(bf16x8, make_uint4(ll[0], ll[1], ll[2], ll[3]));
}


constexpr int kMaxBlocks32 = 6;          // 32-column blocks of one side handled by a wavefront

struct Gram32Args {
    const float* x[4];       // first element (row 0) of each 32-column block of this launch's X group ...
    const float* g[kMaxBlocks32];
    int64_t ldx[4];          // ... and its row stride in floats
    int64_t ldg[kMaxBlocks32];
    float* partial;          // [gridDim.x][k_total * f_total]
    int64_t n_rows;
    int32_t k_total, f_total, x_at, g_at;     // where this launch's block sits in dW
};

template <int NBK, int NBF, int D, bool SPLIT = false>
__global__ __launch_bounds__(256, 1) void tall_gram32_f32_kernel(Gram32Args p)
{
    extern __shared__ __attribute__((aligned(16))) float lds32[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, c = lane & 31;
    f32x16 acc[NBK][NBF];
#pragma unroll
    for (int a = 0; a < NBK; ++a)
#pragma unroll
        for (int b = 0; b < NBF; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    const float* xp[NBK];
    const float* gp[NBF];
#pragma unroll
    for (int a = 0; a < NBK; ++a) xp[a] = p.x[a] + c;
#pragma unroll
    for (int b = 0; b < NBF; ++b) gp[b] = p.g[b] + c;
    // batches of D row pairs, dealt round-robin over all wavefronts of the grid: neighbouring wavefronts stream neighbouring rows
    const int64_t n_batches = (p.n_rows + 2 * D - 1) / (2 * D);
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 4;
    int64_t batch = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    float xv[D][NBK], gv[D][NBF], xn[D][NBK], gn[D][NBF];
    auto fetch = [&](int64_t bt, float (&xo)[D][NBK], float (&go)[D][NBF]) {
        const int64_t row0 = bt * (2 * D) + half;
        if ((bt + 1) * (2 * D) <= p.n_rows) {                     // (wavefront-uniform) a full batch: one 64-bit product per block,
#pragma unroll                                                    // the D row pairs at uniform strides behind it
            for (int a = 0; a < NBK; ++a) {
                const float* q = xp[a] + row0 * p.ldx[a];
#pragma unroll
                for (int d = 0; d < D; ++d) xo[d][a] = q[static_cast<int64_t>(2 * d) * p.ldx[a]];
            }
#pragma unroll
            for (int b = 0; b < NBF; ++b) {
                const float* q = gp[b] + row0 * p.ldg[b];
#pragma unroll
                for (int d = 0; d < D; ++d) go[d][b] = q[static_cast<int64_t>(2 * d) * p.ldg[b]];
            }
            return;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {                              // the ragged last batch: clamped address, value masked
            const int64_t row = row0 + 2 * d;
            const bool live = row < p.n_rows;
            const int64_t rr = live ? row : 0;
#pragma unroll
            for (int a = 0; a < NBK; ++a) {
                const float v = xp[a][rr * p.ldx[a]];
                xo[d][a] = live ? v : 0.f;
            }
#pragma unroll
            for (int b = 0; b < NBF; ++b) {
                const float v = gp[b][rr * p.ldg[b]];
                go[d][b] = live ? v : 0.f;
            }
        }
    };
    if (batch < n_batches) fetch(batch, xv, gv);
    for (; batch < n_batches; batch += stride) {
        const int64_t nb = batch + stride < n_batches ? batch + stride : batch;      // (past the end: re-read this batch, L2 hits)
        fetch(nb, xn, gn);
        if constexpr (SPLIT) {
            // SPLIT (round 5, D = 8): the batch's 16 rows are the 16 k-slots of ONE v_mfma_f32_32x32x16_bf16 -- lane (half, c)'s
            // slot 8 half + e is the value it loaded for row pair e, on both operands alike -- and every product is the six largest
            // partial products of its operands' three bf16 pieces (csrc/tall.hip: closer to float64 than the fmaf chain, 2.7x fewer
            // matrix cycles).  Term-outer, accumulator-inner: consecutive MFMAs write different accumulators.
            static_assert(!SPLIT || D == 8, "one 32x32x16 step per batch");
            // (Operands the split cannot carry -- +-inf, NaN, magnitudes that round to the bf16 infinity -- leave NaN in every sum
            // they enter; gram_finish_kernel<true> recomputes those sums exactly.  Nothing in this loop tests for them.)
            bf16x8 xt[NBK][3], gt[NBF][3];
#pragma unroll
            for (int a = 0; a < NBK; ++a) {
                const float v[8] = {xv[0][a], xv[1][a], xv[2][a], xv[3][a], xv[4][a], xv[5][a], xv[6][a], xv[7][a]};
                split8(v, xt[a]);
            }
#pragma unroll
            for (int b = 0; b < NBF; ++b) {
                const float v[8] = {gv[0][b], gv[1][b], gv[2][b], gv[3][b], gv[4][b], gv[5][b], gv[6][b], gv[7][b]};
                split8(v, gt[b]);
            }
            constexpr int kXi[6] = {2, 0, 1, 1, 0, 0}, kGi[6] = {0, 2, 1, 0, 1, 0};      // (x piece, g piece), smallest term first
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int a = 0; a < NBK; ++a)
#pragma unroll
                    for (int b = 0; b < NBF; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xt[a][kXi[t]], gt[b][kGi[t]], acc[a][b], 0, 0, 0);
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d)
#pragma unroll
                for (int a = 0; a < NBK; ++a)
#pragma unroll
                    for (int b = 0; b < NBF; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[d][a], gv[d][b], acc[a][b], 0, 0, 0);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int a = 0; a < NBK; ++a) xv[d][a] = xn[d][a];
#pragma unroll
            for (int b = 0; b < NBF; ++b) gv[d][b] = gn[d][b];
        }
    }
    // C/D layout of 32x32x2: lane (half, c), register v -> dW[32 a + 8 (v / 4) + 4 half + v % 4][32 b + c]; the four wavefronts
    // of the block add in wave order through LDS, one partial per block
    constexpr int fo = NBF * 32;
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int a = 0; a < NBK; ++a)
#pragma unroll
                for (int b = 0; b < NBF; ++b)
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        const int e = (a * 32 + 8 * (v >> 2) + 4 * half + (v & 3)) * fo + b * 32 + c;
                        lds32[e] = (turn == 0 ? 0.f : lds32[e]) + acc[a][b][v];
                    }
        }
        __syncthreads();
    }
    float* part = p.partial + static_cast<int64_t>(blockIdx.x) * p.k_total * p.f_total;
    for (int e = tid; e < NBK * 32 * fo; e += 256) {
        const int row = e / fo, col = e - row * fo;
        part[static_cast<int64_t>(p.x_at + row) * p.f_total + p.g_at + col] = lds32[e];
    }
}

template <int NBK, int NBF>
int launch_gram32(const Gram32Args& a, unsigned blocks, bool split, hipStream_t s)
{
    const size_t lds = static_cast<size_t>(NBK) * 32 * NBF * 32 * sizeof(float);
    if (split) {                  // the split form: 8 row pairs = the 16 k-slots of one bf16 MFMA
        if (lds > 64 * 1024)
            PYGSD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tall_gram32_f32_kernel<NBK, NBF, 8, true>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        hipLaunchKernelGGL((tall_gram32_f32_kernel<NBK, NBF, 8, true>), dim3(blocks), dim3(256), lds, s, a);
        return check_launch("tall_gram32_f32_kernel (split)");
    }
    constexpr int D = (NBK + NBF) <= 2 ? 16 : ((NBK + NBF) <= 8 ? 8 : 4);        // row pairs per buffered batch
    if (lds > 64 * 1024)
        PYGSD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tall_gram32_f32_kernel<NBK, NBF, D>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((tall_gram32_f32_kernel<NBK, NBF, D>), dim3(blocks), dim3(256), lds, s, a);
    return check_launch("tall_gram32_f32_kernel");
}

template <int NBK>
int pick_gram32(const Gram32Args& a, int nbf, unsigned blocks, bool split, hipStream_t s)
{
    switch (nbf) {
        case 1: return launch_gram32<NBK, 1>(a, blocks, split, s);
        case 2: return launch_gram32<NBK, 2>(a, blocks, split, s);
        case 3: return launch_gram32<NBK, 3>(a, blocks, split, s);
        case 4: if constexpr (NBK <= 2) return launch_gram32<NBK, 4>(a, blocks, split, s); else break;
        case 6: if constexpr (NBK <= 2) return launch_gram32<NBK, 6>(a, blocks, split, s); else break;
        default: break;
    }
    return fail("pygsd_tall_gram: no 32x32 instance for %d x %d blocks", NBK, nbf);
}

// blocks a partial-sum workspace of `budget` bytes admits (one [k_total x f_total] fp32 partial per block)
unsigned budget_blocks(unsigned want, int64_t n_elem)
{
    const int64_t budget = int64_t(64) << 20;
    int64_t cap = budget / (n_elem * static_cast<int64_t>(sizeof(float)));
    if (cap < 1) cap = 1;
    return want > cap ? static_cast<unsigned>(cap) : want;
}

// 0 = never, 1 = where it measured faster (default), 2 = wherever the shapes admit it (PYGSD_GRAM_32X32 = 0 / unset / 1)
int gram32_mode()
{
    const char* e = getenv("PYGSD_GRAM_32X32");
    if (!e || !e[0]) return 1;
    return e[0] == '0' ? 0 : 2;
}

// a side's 32-column blocks (every segment a multiple of 32 columns wide): pointer to the block's first column + row stride
struct Blocks32 {
    const float* p[16];
    int64_t ld[16];
    int n;
};

// out[e] = sum_b partial[b][e] in block order: 64 elements per block, 4 groups of partials combined through LDS.
// RECHECK (behind the split form, round 6): the split represents finite values below the largest bf16 only -- x = +-inf gives
// hi = x and x - hi = NaN, a magnitude above 3.39e38 rounds its hi to inf, and inf times a zero PIECE of a nonzero value is NaN
// where the fp32 product is +-inf (csrc/tall.hip) -- and a NaN piece makes every sum it enters NaN.  So a sum that came out
// finite had operands the split carries exactly, and one that did not is computed again here as the reference's fp32 product
// is: an fma chain over the rows in order on the fp32 operands (summed in float64, rounded once), IEEE products and sums, so
// that inf, -inf and NaN stand where torch.mm puts them.  One compare per output element when nothing is wrong; the chain is as slow as it looks and runs for the
// elements that need it only.
template <bool RECHECK>
__global__ __launch_bounds__(256) void gram_finish_kernel(const float* __restrict__ partial, int n_partials, int64_t n_elem,
                                                          float* __restrict__ out, Blocks32 bx, Blocks32 bg, int f_total,
                                                          int64_t n_rows)
{
    __shared__ float sm[256];
    const int tid = threadIdx.x;
    const int64_t e = static_cast<int64_t>(blockIdx.x) * 64 + (tid & 63);
    const int grp = tid >> 6;
    float acc = 0.f;
    if (e < n_elem) {
#pragma unroll 8
        for (int b = grp; b < n_partials; b += 4) acc += partial[static_cast<int64_t>(b) * n_elem + e];
    }
    sm[tid] = acc;
    __syncthreads();
    if (grp == 0 && e < n_elem) {
        float total = (sm[tid] + sm[tid + 64]) + (sm[tid + 128] + sm[tid + 192]);
        if constexpr (RECHECK) {
            if (!(__builtin_fabsf(total) < __builtin_inff())) {
                const int k = static_cast<int>(e / f_total), f = static_cast<int>(e - static_cast<int64_t>(k) * f_total);
                const float* xc = bx.p[k >> 5] + (k & 31);
                const float* gc = bg.p[f >> 5] + (f & 31);
                const int64_t sx = bx.ld[k >> 5], sg = bg.ld[f >> 5];
                double sum = 0.;                              // float64, rounded once: a million-term fp32 chain is no product
                for (int64_t r = 0; r < n_rows; ++r) sum = fma(static_cast<double>(xc[r * sx]), static_cast<double>(gc[r * sg]), sum);
                total = static_cast<float>(sum);
            }
        }
        out[e] = total;
    }
}

// One [k_total x f_total] fp32 partial per block: the workspace (and the finish kernel's reads) grow with blocks x K x F, so the
// count is capped by a 64 MB budget as well -- 512 blocks of a 1024 x 2048 product would be 4 GB (ADVICE, round 4).
unsigned gram_blocks(int64_t n_rows, int rows_per_tile, int64_t n_elem)
{
    const int64_t tiles = (n_rows + rows_per_tile - 1) / rows_per_tile;
    int64_t b = (tiles + 3) / 4;
    if (b > 512) b = 512;                       // 2 blocks per CU; every block walks >= 1 tile per wavefront
    return budget_blocks(static_cast<unsigned>(b < 1 ? 1 : b), n_elem);
}

// a width (multiple of 16) as chunks of 12 / 8 / 4 / 2 / 1 tiles, none above `cap`
int cut_chunks(const void* base, int64_t ld, int width, size_t esz, int at, int cap, GramChunk* out, int have)
{
    int col = 0;
    while (col < width) {
        const int left = (width - col) / 16;
        const int tiles = (left >= 12 && cap >= 12) ? 12 : (left >= 8 && cap >= 8) ? 8 : left >= 4 ? 4 : left >= 2 ? 2 : 1;
        if (have >= kMaxChunks) return -1;
        out[have++] = GramChunk{static_cast<const unsigned char*>(base) + static_cast<size_t>(col) * esz, ld, tiles, at + col,
                                0, 0, 0, 0, kNoPart, kNoPart};
        col += tiles * 16;
    }
    return have;
}

// G chunks that are whole segments and neighbours in dW become PARTS of one chunk: up to three of them, up to `max_tiles` tiles
// together (fp32: 8, bf16: 12), in one of the tile counts the kernels are instantiated for.  `whole[c]` != 0: chunk c is all of
// its segment.  Returns the new chunk count.
int join_parts(GramChunk* g, const int* whole, int n, int max_tiles, size_t esz, bool strides_free)
{
    auto instance = [](int t) { return t == 1 || t == 2 || t == 4 || t == 8 || t == 12; };
    int out = 0;
    for (int c = 0; c < n;) {
        GramChunk cur = g[c];
        int parts = 1, tiles = cur.tiles, best_parts = 1, best_tiles = cur.tiles;
        while (whole[c] && c + parts < n && parts < 3 && whole[c + parts] && g[c + parts].at == cur.at + tiles * 16 &&
               tiles + g[c + parts].tiles <= max_tiles) {
            tiles += g[c + parts].tiles;
            ++parts;
            bool same_ld = true;
            for (int q = 1; q < parts; ++q) same_ld = same_ld && g[c + q].ld == cur.ld;
            // bf16: parts with row strides of their own only in a 12-tile chunk (the wide instance reads them)
            if (instance(tiles) && (same_ld || strides_free || tiles == 12)) {
                best_parts = parts;
                best_tiles = tiles;
            }
        }
        auto offset = [&](const GramChunk& other) {
            return (static_cast<const unsigned char*>(other.p) - static_cast<const unsigned char*>(cur.p)) /
                   static_cast<int64_t>(esz);
        };
        if (best_parts >= 2) {
            cur.c1 = cur.tiles * 16;
            cur.d1 = offset(g[c + 1]);
            cur.dl1 = g[c + 1].ld - cur.ld;
        }
        if (best_parts == 3) {
            cur.c2 = cur.c1 + g[c + 1].tiles * 16;
            cur.d2 = offset(g[c + 2]);
            cur.dl2 = g[c + 2].ld - cur.ld;
        }
        cur.tiles = best_tiles;
        g[out++] = cur;
        c += best_parts;
    }
    return out;
}
}  // namespace
}  // namespace pygsd

using namespace pygsd;

extern "C" int pygsd_tall_gram_workspace(int64_t n_rows, int32_t k_total, int32_t f_total, int32_t dtype, size_t* bytes)
{
    PYGSD_REQUIRE(bytes, "pygsd_tall_gram_workspace: null output");
    PYGSD_REQUIRE(dtype == 0 || dtype == 1, "pygsd_tall_gram_workspace: dtype must be 0 (fp32) or 1 (bf16)");
    PYGSD_REQUIRE(n_rows >= 0 && k_total > 0 && f_total > 0, "pygsd_tall_gram_workspace: sizes must be positive");
    // (the fp32 32x32 form runs <= 256 blocks: it never needs more than this)
    *bytes = static_cast<size_t>(gram_blocks(n_rows, dtype == 1 ? 32 : 16, static_cast<int64_t>(k_total) * f_total)) * k_total *
             f_total * sizeof(float);
    return 0;
}

extern "C" int pygsd_tall_gram(const void* const* xs, const int64_t* ldx, const int32_t* x_widths, int32_t n_x,
                               const void* const* gs, const int64_t* ldg, const int32_t* g_widths, int32_t n_g, int64_t n_rows,
                               int32_t dtype, float* out, void* workspace, size_t workspace_bytes, void* stream)
{
    PYGSD_REQUIRE(dtype == 0 || dtype == 1, "pygsd_tall_gram: dtype must be 0 (fp32) or 1 (bf16), got %d", dtype);
    PYGSD_REQUIRE(n_x >= 1 && n_x <= 8 && xs && ldx && x_widths && n_g >= 1 && n_g <= 8 && gs && ldg && g_widths,
                  "pygsd_tall_gram: 1..8 column segments on either side");
    PYGSD_REQUIRE(n_rows >= 0 && out, "pygsd_tall_gram: negative row count or null output");
    const size_t esz = dtype == 1 ? 2 : 4;
    const int vec = dtype == 1 ? 8 : 4;
    GramArgs a{};
    // widest G chunk: 8 tiles, bf16 12 (the one-wavefront-per-SIMD instance; PYGSD_GRAM_WIDE=0 keeps 8 -- measurement / A-B)
    const char* wide_env = getenv("PYGSD_GRAM_WIDE");
    const int g_cap = (dtype == 1 && !(wide_env && wide_env[0] == '0')) ? 12 : 8;
    int g_whole[kMaxChunks] = {};
    int nx = 0, ng = 0, k_total = 0, f_total = 0;
    for (int s = 0; s < n_x; ++s) {
        PYGSD_REQUIRE(x_widths[s] > 0 && x_widths[s] % 16 == 0, "pygsd_tall_gram: X segment %d is %d columns wide (multiples "
                      "of 16)", s, x_widths[s]);
        PYGSD_REQUIRE(n_rows == 0 || (xs[s] && aligned16(xs[s]) && ldx[s] >= x_widths[s] && ldx[s] % vec == 0),
                      "pygsd_tall_gram: X segment %d null, not 16-byte aligned, or row stride not a multiple of 16 bytes >= its "
                      "width", s);
        nx = cut_chunks(xs[s], ldx[s], x_widths[s], esz, k_total, 4, a.x, nx);
        PYGSD_REQUIRE(nx > 0, "pygsd_tall_gram: more than %d column chunks in X", kMaxChunks);
        k_total += x_widths[s];
    }
    for (int s = 0; s < n_g; ++s) {
        PYGSD_REQUIRE(g_widths[s] > 0 && g_widths[s] % 16 == 0, "pygsd_tall_gram: G segment %d is %d columns wide (multiples "
                      "of 16)", s, g_widths[s]);
        PYGSD_REQUIRE(n_rows == 0 || (gs[s] && aligned16(gs[s]) && ldg[s] >= g_widths[s] && ldg[s] % vec == 0),
                      "pygsd_tall_gram: G segment %d null, not 16-byte aligned, or row stride not a multiple of 16 bytes >= its "
                      "width", s);
        const int before = ng;
        ng = cut_chunks(gs[s], ldg[s], g_widths[s], esz, f_total, g_cap, a.g, ng);
        PYGSD_REQUIRE(ng > 0, "pygsd_tall_gram: more than %d column chunks in G", kMaxChunks);
        for (int c = before; c < ng; ++c) g_whole[c] = (ng - before == 1) ? 1 : 0;
        f_total += g_widths[s];
    }
    ng = join_parts(a.g, g_whole, ng, g_cap, esz, dtype == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int64_t n_elem = static_cast<int64_t>(k_total) * f_total;
    if (n_rows == 0) {
        PYGSD_HIP_TRY(hipMemsetAsync(out, 0, static_cast<size_t>(n_elem) * sizeof(float), s));
        return 0;
    }
    unsigned blocks = gram_blocks(n_rows, dtype == 1 ? 32 : 16, n_elem);
    PYGSD_REQUIRE(workspace && workspace_bytes >= static_cast<size_t>(blocks) * n_elem * sizeof(float),
                  "pygsd_tall_gram: workspace null or too small (pygsd_tall_gram_workspace)");
    const int mode32 = gram32_mode();
    bool all32 = dtype == 0 && mode32 != 0 && k_total / 32 <= 16 && f_total / 32 <= 16;
    for (int sgm = 0; sgm < n_x && all32; ++sgm) all32 = x_widths[sgm] % 32 == 0;
    for (int sgm = 0; sgm < n_g && all32; ++sgm) all32 = g_widths[sgm] % 32 == 0;
    if (all32 && mode32 == 1) {
        // Measured (profiles/r5d_gram_forms.json): with 12 accumulator blocks per wavefront -- 64 x 192 or 128 x 96 outputs, the
        // fp32 inception block's x^T [dx0 | dP_1 | dP_2] -- this form takes 0.53 ms where the LDS-transposing one takes 0.61; with
        // 8 blocks (SGCNConv's 64 x 128) it loses, 0.126 against 0.111 ms: too few MFMA cycles per batch to cover its loads at one
        // wavefront per SIMD.  So: only where a wavefront holds at least 10 blocks.
        const int xb = k_total / 32, gb = f_total / 32;
        const int gx = xb >= 4 ? 4 : (xb >= 2 ? 2 : 1), gmax = gx == 4 ? 3 : 6;
        all32 = gx * (gb < gmax ? gb : gmax) >= 10;
    }
    if (all32) {
        // fp32, every segment a multiple of 32 columns: the 32x32x2 form -- operands straight from coalesced loads, every element
        // fetched once per (X group, G group) pair; groups of <= 4 / 2 / 1 blocks of X against <= 3 / 6 / 6 blocks of G
        Blocks32 bx{}, bg{};
        for (int sgm = 0; sgm < n_x; ++sgm)
            for (int c0 = 0; c0 < x_widths[sgm]; c0 += 32) {
                bx.p[bx.n] = static_cast<const float*>(xs[sgm]) + c0;
                bx.ld[bx.n++] = ldx[sgm];
            }
        for (int sgm = 0; sgm < n_g; ++sgm)
            for (int c0 = 0; c0 < g_widths[sgm]; c0 += 32) {
                bg.p[bg.n] = static_cast<const float*>(gs[sgm]) + c0;
                bg.ld[bg.n++] = ldg[sgm];
            }
        const int gx = bx.n >= 4 ? 4 : (bx.n >= 2 ? 2 : 1);                     // X blocks per group
        const int gmax = gx == 4 ? 3 : 6;                                       // G blocks per group (gx * gmax <= 12)
        const int64_t batches = (n_rows + 31) / 32;                             // (16 .. 64 rows per batch, by instance)
        unsigned b32 = static_cast<unsigned>(batches / 4 < 1 ? 1 : (batches / 4 > 256 ? 256 : batches / 4));   // one block per CU
        if (b32 > blocks) b32 = blocks;                                         // (the workspace was sized for `blocks`)
        hipStream_t st = static_cast<hipStream_t>(stream);
        ProfScope prof(PYGSD_K_DENSE_BWD, st);
        const bool split = tall_f32_form().load() == 0;                         // read once: every launch of this call agrees
        for (int x0 = 0; x0 < bx.n; x0 += gx) {
            const int nk = bx.n - x0 >= gx ? gx : (bx.n - x0 >= 2 ? 2 : 1);
            for (int g0 = 0; g0 < bg.n;) {
                int nf = bg.n - g0 >= gmax ? gmax : bg.n - g0;
                if (nf == 5) nf = 4;                                             // (instances: 1, 2, 3, 4, 6 blocks of G)
                if (nk == 4 && nf > 3) nf = 3;
                Gram32Args ga{};
                for (int q = 0; q < nk; ++q) {
                    ga.x[q] = bx.p[x0 + q];
                    ga.ldx[q] = bx.ld[x0 + q];
                }
                for (int q = 0; q < nf; ++q) {
                    ga.g[q] = bg.p[g0 + q];
                    ga.ldg[q] = bg.ld[g0 + q];
                }
                ga.partial = static_cast<float*>(workspace);
                ga.n_rows = n_rows;
                ga.k_total = k_total;
                ga.f_total = f_total;
                ga.x_at = x0 * 32;
                ga.g_at = g0 * 32;
                int rc = nk == 4 ? pick_gram32<4>(ga, nf, b32, split, st)
                                 : (nk == 2 ? pick_gram32<2>(ga, nf, b32, split, st) : pick_gram32<1>(ga, nf, b32, split, st));
                if (rc) return rc;
                g0 += nf;
            }
            if (nk < gx) {                                                       // a ragged tail of the X blocks: step by what was taken
                x0 -= gx - nk;
            }
        }
        if (split)
            hipLaunchKernelGGL(gram_finish_kernel<true>, dim3(static_cast<unsigned>((n_elem + 63) / 64)), dim3(256), 0, st,
                               static_cast<const float*>(workspace), static_cast<int>(b32), n_elem, out, bx, bg, f_total, n_rows);
        else
            hipLaunchKernelGGL(gram_finish_kernel<false>, dim3(static_cast<unsigned>((n_elem + 63) / 64)), dim3(256), 0, st,
                               static_cast<const float*>(workspace), static_cast<int>(b32), n_elem, out, bx, bg, f_total, n_rows);
        return check_launch("gram_finish_kernel");
    }
    a.partial = static_cast<float*>(workspace);
    a.n_rows = n_rows;
    a.k_total = k_total;
    a.f_total = f_total;
    ProfScope prof(PYGSD_K_DENSE_BWD, s);
    // chunk pair fastest: the blocks that re-read a row range for another column chunk run side by side (the re-read is an
    // Infinity-Cache hit, not a second trip to HBM)
    a.n_g = ng;
    const dim3 grid(static_cast<unsigned>(ng * nx), blocks);
    int tk = 1, tf = 1;                           // widest chunk on either side sizes the LDS region
    for (int c = 0; c < nx; ++c) tk = a.x[c].tiles > tk ? a.x[c].tiles : tk;
    for (int c = 0; c < ng; ++c) tf = a.g[c].tiles > tf ? a.g[c].tiles : tf;
    const size_t combine = static_cast<size_t>(tk) * 16 * tf * 16 * sizeof(float);     // the block's dW chunk, wave-order sum
    if (dtype == 1) {
        size_t lds = static_cast<size_t>(4) * (tk + tf) * 1024;                        // 4 wavefronts x [tile][32][16] bf16
        if (lds < combine) lds = combine;
        if (tf == 12) {                       // a three-part chunk: the one-wavefront-per-SIMD instance holds its 192 accumulators
            if (lds > 64 * 1024)
                PYGSD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tall_gram_bf16_wide_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
            hipLaunchKernelGGL(tall_gram_bf16_wide_kernel, grid, dim3(256), lds, s, a);
        } else {
            if (lds > 64 * 1024)
                PYGSD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tall_gram_bf16_kernel),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
            hipLaunchKernelGGL(tall_gram_bf16_kernel, grid, dim3(256), lds, s, a);
        }
    } else {
        size_t lds = static_cast<size_t>(4) * 16 * ((tk * 16 + 4) + (tf * 16 + 4)) * sizeof(float);   // 4 x [16][cols + 4]
        if (lds < combine) lds = combine;
        if (lds > 64 * 1024)
            PYGSD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(tall_gram_f32_kernel),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
        hipLaunchKernelGGL(tall_gram_f32_kernel, grid, dim3(256), lds, s, a);
    }
    if (int rc = check_launch("tall_gram_kernel")) return rc;
    hipLaunchKernelGGL(gram_finish_kernel<false>, dim3(static_cast<unsigned>((n_elem + 63) / 64)), dim3(256), 0, s,
                       static_cast<const float*>(workspace), static_cast<int>(blocks), n_elem, out, Blocks32{}, Blocks32{}, f_total,
                       n_rows);
    return check_launch("gram_finish_kernel");
}
