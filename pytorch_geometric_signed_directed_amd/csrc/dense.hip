// Fused dense stage of MagNetConv / MSConv on the MFMA matrix cores -- the only GEMM-shaped work on the path.  Two arithmetic forms
// (pygsd_dense_f32_form): exact fp32 (v_mfma_f32_16x16x4_f32, bitwise an fmaf chain) for every shape, and -- the default at hidden
// widths 64 / 128 -- the bf16 pipe by three-way splitting (dense_fwd_kernel<..., SPLIT>, dense_bwd_split_kernel).
//
//   forward :  out_real = sum_k (A_k - B_k) W_k + b ,  out_imag = sum_k (A_k + B_k) W_k + b
//              (A_k / B_k = k-th Chebyshev terms of the real / imaginary chain; reference
//               nn/directed/MagNetConv.py:189-192,198-211,217-247: four matmuls per order, then
//               out_real = rr - ii, out_imag = ir + ri, += bias.  By linearity the +- is applied
//               to the GEMM inputs, so one pass reads A_k, B_k once and writes both outputs once.)
//   backward:  P = G_r + G_i , M = G_i - G_r ;  dA_k = P W_k^T , dB_k = M W_k^T ,
//              dW_k = A_k^T P + B_k^T M , db = colsum(P)
//
// Tiling: one wavefront owns 16 node rows.  The A operand of mfma_16x16x4 is ONE f32 per lane
// (lane l: row l&15, k-slot l>>4), so a lane's float4 global load of 4 consecutive features feeds
// four successive MFMAs directly (k-slots are summed, their order is free): 16 B per lane, 64 B
// contiguous per row per instruction, no LDS staging of activations.  W (or W^T) lives in LDS with a
// +4-float row pad (conflict-free ds_read_b32 of a B fragment).  Weight-gradient partials stay in
// MFMA accumulators across a persistent row loop, are combined per block through LDS and finished by
// a small deterministic tree kernel -- no atomics.
#include "common.hpp"

namespace pygsd {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxOrder = 4;   // K + 1 <= 4
constexpr int kChunk = 64;     // output-column / input-column chunk handled by one block
constexpr int kPad = 4;

__device__ __forceinline__ f32x4 mfma(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

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

struct DenseFwdArgs {
    const float* a[kMaxOrder];
    const float* b[kMaxOrder];
    const float* w;      // [k1][f_in][f_out]
    const float* bias;   // [f_out] or null
    float* out_r;
    float* out_i;
    int32_t n_rows, f_in, f_out, k1;
    pygsd_piece_layout lay;   // PIECES instances: where the rows of a[k1 - 1] / b[k1 - 1] live (common.hpp: piece_row)
    int32_t lay_shift;        // log2(lay.slot_floats / 16)
};

// grid = (row blocks, ceil(f_out / 64)); block = 256 threads = 4 wavefronts x 16 rows.
// The MFMA is issued TRANSPOSED (A operand = W^T fragment, B operand = activation fragment), so the
// C/D layout (col = lane & 15, row = 4 (lane >> 4) + reg) hands every lane 4 CONSECUTIVE output features
// of ONE node row: the epilogue is one float4 store per tile instead of four scalar stores.
// FIN > 0 fixes f_in at compile time: the feature loop unrolls and all of a Chebyshev term's 16-byte
// loads are in flight before its first MFMA.
// WAVES = 8 when the W slice leaves room for only one block per CU (f_in = 128, K = 2: 104 KB): two
// wavefronts per SIMD instead of one, so one's row loads overlap the other's MFMAs.
// PIECES (round 5, sharded layers, FIN > 0): the LAST term's operands are read through a piece layout -- straight out of the
// return exchange's receive buffer, 16-float pieces of a row at slot strides (no merge pass in front of this kernel).  The plain
// instances are untouched by it.
// SPLIT (round 5; NT = 4, FIN = 64 / 128): the products on the bf16 matrix pipe by three-way splitting, as dense_bwd_split_kernel
// and csrc/tall.hip -- D = A - B and S = A + B are formed in fp32 as before, split into three bf16 pieces each, and multiplied
// with the pre-split W fragments in LDS ([k][32-column block][tile][3][64] x 16 B) by v_mfma_f32_16x16x32_bf16; a lane's 8 k-slots
// of block kb are the features 32 kb + 16 h + 4 g + r of the two float4 pieces it already loads.  The six partial products of a
// 32-column block are summed in accumulators of their own and added to the running sums ONCE per block: the running sum is
// rounded once per 32 features instead of six times.
template <int NT, int FIN, int WAVES, bool PIECES = false, bool SPLIT = false>
__global__ __launch_bounds__(WAVES * 64) void dense_fwd_kernel(DenseFwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int n0 = static_cast<int>(blockIdx.y) * kChunk;
    constexpr int nc = NT * 16;
    constexpr int ws = nc + kPad;
    const int f_in = FIN > 0 ? FIN : p.f_in;
    const int wrows = p.k1 * f_in;
    if constexpr (SPLIT) {
        static_assert(!SPLIT || (FIN % 32 == 0 && FIN > 0 && NT % 2 == 0), "split form: whole 32-column blocks, tile pairs");
        constexpr int KB = FIN / 32;
        uint4* wfrag = reinterpret_cast<uint4*>(lds);
        for (int idx = tid; idx < p.k1 * KB * NT * 64; idx += WAVES * 64) {
            const int lane = idx & 63, nt = (idx >> 6) % NT, kb = ((idx >> 6) / NT) % KB, k = (idx >> 6) / NT / KB;
            const int i = lane & 15, g = lane >> 4;
            const float* wcol = p.w + (static_cast<int64_t>(k) * FIN + 32 * kb + 4 * g) * p.f_out + n0 + 16 * nt + i;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = wcol[static_cast<int64_t>(16 * (e >> 2) + (e & 3)) * p.f_out];
            bf16x8 t[3];
            split8(v, t);
            uint4* dst = wfrag + (((k * KB + kb) * NT + nt) * 3) * 64 + lane;
            dst[0] = __builtin_bit_cast(uint4, t[0]);
            dst[64] = __builtin_bit_cast(uint4, t[1]);
            dst[128] = __builtin_bit_cast(uint4, t[2]);
        }
    } else {
        for (int idx = tid; idx < wrows * nc; idx += WAVES * 64) {
            const int r = idx / nc, c = idx - r * nc;
            lds[r * ws + c] = p.w[static_cast<int64_t>(r) * p.f_out + n0 + c];
        }
    }
    __syncthreads();

    const int lane = tid & 63, i = lane & 15, g = lane >> 4;
    const int n_tiles = (p.n_rows + 15) >> 4;
    const int stride = static_cast<int>(gridDim.x) * WAVES;
    // ---- non-finite sums (round 6) ------------------------------------------------------------------------------------------
    // The reference forms rr = sum_k A_k W_k and ii = sum_k B_k W_k and then rr - ii, rr + ii (MagNetConv.py:217-247).  On finite
    // values (A - B) W is the same number up to rounding; on non-finite ones it is not -- W = inf gives rr = ii = +-inf and
    // rr - ii = NaN where (A - B) inf is +-inf -- and the split form carries finite values below the largest bf16 only (a NaN
    // piece makes every sum it enters NaN: csrc/tall.hip, any_not_finite).  So a tile that holds a non-finite sum, in either
    // form, is computed again the reference's way: rr and ii as fmaf chains over the terms and features in order on the fp32
    // operands, then the difference and the sum.  Rolled loops, eight sums live: the branch costs the kernel no registers.
    auto not_finite = [&](const f32x4 (&ar)[NT], const f32x4 (&ai)[NT]) {
        f32x4 z = ar[0] * 0.f;                                    // 0 for a finite sum, NaN for +-inf and NaN
        z += ai[0] * 0.f;
#pragma unroll
        for (int nt = 1; nt < NT; ++nt) {
            z += ar[nt] * 0.f;
            z += ai[nt] * 0.f;
        }
        const float c = (z[0] + z[1]) + (z[2] + z[3]);
        return __builtin_amdgcn_ballot_w64(c != c) != 0;          // wavefront-uniform
    };
    auto exact_rows = [&](int tl, f32x4 (&ar)[NT], f32x4 (&ai)[NT]) {
        const int r0 = tl << 4;
        const int lrow = (r0 + i < p.n_rows) ? r0 + i : p.n_rows - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            double sa[4] = {0., 0., 0., 0.}, sb[4] = {0., 0., 0., 0.};       // float64 sums, rounded once (csrc/tall.hip)
#pragma unroll 1
            for (int kk = 0; kk < p.k1; ++kk) {
                PieceRow pr;
                pr.base = static_cast<int64_t>(lrow) * f_in;
                pr.slot_stride = 0;
                int sh = 31, mk = 0x7fffffff;
                if constexpr (PIECES) {
                    if (kk == p.k1 - 1) {
                        pr = piece_row(p.lay, lrow);
                        sh = p.lay_shift;
                        mk = (1 << sh) - 1;
                    }
                }
                const float* wrow = p.w + static_cast<int64_t>(kk) * f_in * p.f_out + n0 + 16 * nt + 4 * g;
                const float* ak = p.a[kk];
                const float* bk = p.b[kk];
#pragma unroll 1
                for (int f = 0; f < f_in; ++f) {
                    const int64_t off = piece_offset(pr, f >> 4, sh, mk) + (f & 15);
                    const double av = ak[off], bv = bk[off];
                    const float4 w4 = ldg4(wrow + static_cast<int64_t>(f) * p.f_out);
                    const double w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sa[r] = fma(av, w[r], sa[r]);
                        sb[r] = fma(bv, w[r], sb[r]);
                    }
                }
            }
            // rr and ii rounded to fp32 as the reference holds them, then the difference and the sum in fp32
            float ra[4], rb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ra[r] = static_cast<float>(sa[r]);
                rb[r] = static_cast<float>(sb[r]);
            }
            ar[nt] = f32x4{ra[0] - rb[0], ra[1] - rb[1], ra[2] - rb[2], ra[3] - rb[3]};
            ai[nt] = f32x4{ra[0] + rb[0], ra[1] + rb[1], ra[2] + rb[2], ra[3] + rb[3]};
        }
    };
    // (wavefront-uniform, which the compiler cannot see through tid >> 6: without it the term's base pointers p.a[k] are fetched by
    // VECTOR loads, and the wait for them drains every row load in flight)
    int tile = __builtin_amdgcn_readfirstlane(static_cast<int>(blockIdx.x) * WAVES + (tid >> 6));
    if constexpr (FIN > 0) {
        // The (tile, Chebyshev term) steps of a wavefront form ONE stream, run as a two-stage pipeline: the 16-byte row loads of
        // step s + 1 are issued before the MFMAs of step s, and a tile's stores go out behind the loads of the next tile's first
        // term -- so the wait in front of the next MFMAs covers loads only (vmcnt counts in order: a wait for a load issued AFTER
        // the stores would wait for them too, which is what holds a grid-stride copy at 4.5 TB/s on this part), and a row's HBM
        // latency is hidden by this wavefront's own MFMAs, not only by the other wavefronts of its SIMD.
        constexpr int NL = FIN / 16;
        float4 ra[2][NL], rb[2][NL];
        f32x4 acc_r[NT], acc_i[NT];
        float4 bv[NT];                                             // the bias of this lane's columns, read once: a load in the
#pragma unroll                                                     // store epilogue would put a full wait in front of the stores
        for (int nt = 0; nt < NT; ++nt) bv[nt] = p.bias ? ldg4(p.bias + n0 + nt * 16 + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        auto issue = [&](int tl, int k, float4 (&qa)[NL], float4 (&qb)[NL]) {
            const int r0 = tl << 4;
            const int lrow = (r0 + i < p.n_rows) ? r0 + i : p.n_rows - 1;      // clamped: stores are masked
            if constexpr (PIECES) {
                // address arithmetic only inside the (wavefront-uniform) choice; the loads themselves are common code
                PieceRow pr;
                pr.base = static_cast<int64_t>(lrow) * FIN;
                pr.slot_stride = 0;
                int sh = 31, mk = 0x7fffffff;                         // row-major: piece q at base + 16 q
                if (k == p.k1 - 1) {
                    pr = piece_row(p.lay, lrow);
                    sh = p.lay_shift;
                    mk = (1 << sh) - 1;
                }
                const float* ap = p.a[k] + 4 * g;
                const float* bp = p.b[k] + 4 * g;
#pragma unroll
                for (int t = 0; t < NL; ++t) {
                    const int64_t off = piece_offset(pr, t, sh, mk);
                    qa[t] = ldg4(ap + off);
                    qb[t] = ldg4(bp + off);
                }
                return;
            }
            const float* ap = p.a[k] + static_cast<int64_t>(lrow) * FIN + 4 * g;
            const float* bp = p.b[k] + static_cast<int64_t>(lrow) * FIN + 4 * g;
#pragma unroll
            for (int t = 0; t < NL; ++t) {
                qa[t] = ldg4(ap + 16 * t);
                qb[t] = ldg4(bp + 16 * t);
            }
        };
        auto step = [&](int tl, int k, const float4 (&qa)[NL], const float4 (&qb)[NL]) {
            if (k == 0) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc_r[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    acc_i[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            if constexpr (SPLIT) {
                constexpr int KB = FIN / 32;
                constexpr int kWi[6] = {2, 0, 1, 1, 0, 0}, kXi[6] = {0, 2, 1, 0, 1, 0};      // (w piece, x piece), smallest term first
                const uint4* wf = reinterpret_cast<const uint4*>(lds) + static_cast<size_t>(k) * KB * NT * 3 * 64 + lane;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const float4 a0 = qa[2 * kb], a1 = qa[2 * kb + 1], b0 = qb[2 * kb], b1 = qb[2 * kb + 1];
                    const float d[8] = {a0.x - b0.x, a0.y - b0.y, a0.z - b0.z, a0.w - b0.w,
                                        a1.x - b1.x, a1.y - b1.y, a1.z - b1.z, a1.w - b1.w};
                    const float sm[8] = {a0.x + b0.x, a0.y + b0.y, a0.z + b0.z, a0.w + b0.w,
                                         a1.x + b1.x, a1.y + b1.y, a1.z + b1.z, a1.w + b1.w};
                    bf16x8 dx[3], sx[3];
                    split8(d, dx);
                    split8(sm, sx);
#pragma unroll
                    for (int np = 0; np < NT / 2; ++np) {            // tile pairs: four independent accumulators per term
                        bf16x8 w[2][3];
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc)
                                w[j][pc] = __builtin_bit_cast(bf16x8, wf[((kb * NT + 2 * np + j) * 3 + pc) * 64]);
                        f32x4 tr[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                        f32x4 ti[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                        for (int t = 0; t < 6; ++t)
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                tr[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][kWi[t]], dx[kXi[t]], tr[j], 0, 0, 0);
                                ti[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[j][kWi[t]], sx[kXi[t]], ti[j], 0, 0, 0);
                            }
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            acc_r[2 * np + j] += tr[j];
                            acc_i[2 * np + j] += ti[j];
                        }
                    }
                }
            } else {
            const float* wk = lds + (k * FIN + 4 * g) * ws + i;
#pragma unroll
            for (int t = 0; t < NL; ++t) {
                const float4 a = qa[t], b = qb[t];
                const float d[4] = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w};
                const float sm[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
                const float* wt = wk + 16 * t * ws;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float w = wt[m * ws + nt * 16];
                        acc_r[nt] = mfma(w, d[m], acc_r[nt]);   // (W^T)(D^T): rows = out features
                        acc_i[nt] = mfma(w, sm[m], acc_i[nt]);
                    }
                }
            }
            }
            if (k == p.k1 - 1) {
                const int r0 = tl << 4;
                // (decided here, acted on BEHIND the stores, which wait for the same sums: a branch in front of them is a barrier for
                // the scheduler -- the same lane then stores the tile a second time over the same addresses, in program order)
                const bool redo = not_finite(acc_r, acc_i);
                // transposed C/D: lane (i, g), reg r  ->  out[node r0 + i][feature n0 + 16 nt + 4 g + r]
                auto store_tile = [&]() {
                    if (r0 + i < p.n_rows) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            const int col = n0 + nt * 16 + 4 * g;
                            const int64_t o = static_cast<int64_t>(r0 + i) * p.f_out + col;
                            *reinterpret_cast<float4*>(p.out_r + o) = make_float4(acc_r[nt][0] + bv[nt].x, acc_r[nt][1] + bv[nt].y,
                                                                                  acc_r[nt][2] + bv[nt].z, acc_r[nt][3] + bv[nt].w);
                            *reinterpret_cast<float4*>(p.out_i + o) = make_float4(acc_i[nt][0] + bv[nt].x, acc_i[nt][1] + bv[nt].y,
                                                                                  acc_i[nt][2] + bv[nt].z, acc_i[nt][3] + bv[nt].w);
                        }
                    }
                };
                store_tile();
                if (redo) {
                    exact_rows(tl, acc_r, acc_i);
                    store_tile();
                }
            }
        };
        int k = 0;
        if (tile >= n_tiles) return;
        issue(tile, 0, ra[0], rb[0]);
        for (;;) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {                 // (unrolled: the two register sets are named, not indexed)
                // `return`, not `break`: an edge from here back to the loop header -- never taken, but the compiler cannot know --
                // makes the header wait for loads that are pending on that path only (a vmcnt(2) that also covered the stores)
                if (tile >= n_tiles) return;
                const int nk = k + 1 < p.k1 ? k + 1 : 0;
                const int ntile = nk ? tile : tile + stride;
                // no branch around the prefetch (past the last step it re-reads this step's rows, L2 hits): with one, the wait below
                // is placed for the path that skipped it and then covers the prefetched loads as well
                issue(ntile < n_tiles ? ntile : tile, ntile < n_tiles ? nk : k, ra[half ^ 1], rb[half ^ 1]);
                step(tile, k, ra[half], rb[half]);
                tile = ntile;
                k = nk;
            }
        }
        return;
    }
    for (; tile < n_tiles; tile += stride) {
        const int r0 = tile << 4;
        const int lrow = (r0 + i < p.n_rows) ? r0 + i : p.n_rows - 1;  // clamped: stores are masked
        f32x4 acc_r[NT], acc_i[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            acc_r[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc_i[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        for (int k = 0; k < p.k1; ++k) {
            const float* ap = p.a[k] + static_cast<int64_t>(lrow) * f_in + 4 * g;
            const float* bp = p.b[k] + static_cast<int64_t>(lrow) * f_in + 4 * g;
            const float* wk = lds + (k * f_in + 4 * g) * ws + i;
            for (int t = 0; t < f_in; t += 16) {
                const float4 a = ldg4(ap + t);
                const float4 b = ldg4(bp + t);
                const float d[4] = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w};
                const float s[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
                const float* wt = wk + t * ws;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float w = wt[m * ws + nt * 16];
                        acc_r[nt] = mfma(w, d[m], acc_r[nt]);   // (W^T)(D^T): rows = out features
                        acc_i[nt] = mfma(w, s[m], acc_i[nt]);
                    }
                }
            }
        }
        if (not_finite(acc_r, acc_i)) exact_rows(tile, acc_r, acc_i);
        // transposed C/D: lane (i, g), reg r  ->  out[node r0 + i][feature n0 + 16 nt + 4 g + r]
        if (r0 + i < p.n_rows) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = n0 + nt * 16 + 4 * g;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias) bv = ldg4(p.bias + col);
                const int64_t o = static_cast<int64_t>(r0 + i) * p.f_out + col;
                *reinterpret_cast<float4*>(p.out_r + o) =
                    make_float4(acc_r[nt][0] + bv.x, acc_r[nt][1] + bv.y, acc_r[nt][2] + bv.z, acc_r[nt][3] + bv.w);
                *reinterpret_cast<float4*>(p.out_i + o) =
                    make_float4(acc_i[nt][0] + bv.x, acc_i[nt][1] + bv.y, acc_i[nt][2] + bv.z, acc_i[nt][3] + bv.w);
            }
        }
    }
}

struct DenseBwdArgs {
    const float* a[kMaxOrder];
    const float* b[kMaxOrder];
    float* da[kMaxOrder];
    float* db[kMaxOrder];
    const float* gr;     // [n][f_out], row stride ldg (0: ONE row broadcast to every node -- the gradient of a sum)
    const float* gi;
    int64_t ldg;
    const float* w;      // [k1][f_in][f_out]
    float* partial;      // [n_partials][k1 * f_in * f_out + f_out]
    int32_t n_rows, f_in, f_out, k1;
    // PIECES instances (term k1 - 1 only): a / b read through lay_in, da / db stored through lay_out (every replica)
    pygsd_piece_layout lay_in, lay_out;
    int32_t in_on, out_on, in_shift, out_shift;
    // split form: tiles whose dA / dB sums came out non-finite, per wavefront of the product kernel -- [gx][k1][gz][4][kBadSlot]
    // int32 behind the partials: [0] = how many (every wavefront writes it), [1 ..] = the first kBadSlot - 1 tile numbers
    int32_t* bad;
};

constexpr int kBadSlot = 8;

// grid = (row blocks, k1, ceil(f_in / 64)); block = 256 threads.  Block (x, k, ci) produces
// dA_k[:, chunk ci], dB_k[:, chunk ci] for its rows and one partial of dW_k[chunk ci, :] (+ db when
// k == 0 and ci == 0).  NTI = f_in-chunk tiles (<= 4), NTO = f_out tiles (<= 8).
//
// XPOSE (f_out <= 64): every global load is a 16-byte row load; the column fragments phase 2 needs
// (rows in the MFMA k-slot, features across the 16 lanes) are produced by a round trip through a
// wavefront-private LDS region (ds_write_b128 of the row fragments, ds_read_b32 of the column fragments,
// both conflict-free at a +4-float row pad) instead of 64 scalar global loads per tile.
// PIECES (round 5, sharded layers; XPOSE only): the last term's A / B rows come straight out of the return exchange's receive
// buffer and its dA / dB rows go straight into the next propagate's send buffers (all replicas) -- no merge pass in front of this
// kernel, no packing pass behind it.
template <int NTI, int NTO, bool XPOSE, bool PIECES = false>
__global__ __launch_bounds__(256, (NTO <= 4 ? 2 : 1)) void dense_bwd_kernel(DenseBwdArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int k = blockIdx.y;
    const int c0 = static_cast<int>(blockIdx.z) * kChunk;   // first f_in column of this chunk
    constexpr int fc = NTI * 16;
    constexpr int fo = NTO * 16;
    constexpr int ws = fc + kPad;
    // W_k^T slice: wt[kk][j] = W[k][c0 + j][kk]
    const float* wsrc = p.w + (static_cast<int64_t>(k) * p.f_in + c0) * p.f_out;
    for (int idx = tid; idx < fc * fo; idx += 256) {
        const int j = idx / fo, kk = idx - j * fo;   // coalesced read along kk
        lds[kk * ws + j] = wsrc[static_cast<int64_t>(j) * p.f_out + kk];
    }
    __syncthreads();

    const int lane = tid & 63, i = lane & 15, g = lane >> 4;
    constexpr int rs = (fo > fc ? fo : fc) + kPad;                 // row stride of the staging region
    float* stage = lds + fo * ws + (tid >> 6) * (2 * 16 * rs);    // wavefront-private: [2][16][rs]
    const bool do_bias = (k == 0) && (blockIdx.z == 0);
    const float* ak = p.a[k];
    const float* bk = p.b[k];
    float* dak = p.da[k];
    float* dbk = p.db[k];

    f32x4 acc_w[NTI][NTO];
    float acc_bias[NTO];
#pragma unroll
    for (int ft = 0; ft < NTI; ++ft)
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) acc_w[ft][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NTO; ++nt) acc_bias[nt] = 0.f;

    const int n_tiles = (p.n_rows + 15) >> 4;
    for (int tile = static_cast<int>(blockIdx.x) * 4 + (tid >> 6); tile < n_tiles;
         tile += static_cast<int>(gridDim.x) * 4) {
        const int r0 = tile << 4;
        // ---- all of the tile's loads first (phase-1 float4 rows of G, phase-2 column fragments of
        //      G / A_k / B_k), so every fetch is in flight before the first MFMA and none of them queues
        //      behind this tile's dA / dB stores -------------------------------------------------------
        const bool lrow_live = r0 + i < p.n_rows;
        const int lrow = lrow_live ? r0 + i : p.n_rows - 1;
        const float* grp = p.gr + static_cast<int64_t>(lrow) * p.ldg + 4 * g;
        const float* gip = p.gi + static_cast<int64_t>(lrow) * p.ldg + 4 * g;
        float4 xg[NTO], yg[NTO];
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) {
            xg[nt] = ldg4(grp + nt * 16);
            yg[nt] = ldg4(gip + nt * 16);
            if (XPOSE && !lrow_live) {          // rows past the end must contribute nothing to dW
                xg[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
                yg[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float av[4][NTI], bv[4][NTI];
        if (XPOSE) {
            // A_k / B_k rows as float4, then LDS round trip into column fragments
            const float* arow = ak + static_cast<int64_t>(lrow) * p.f_in + c0 + 4 * g;
            const float* brow = bk + static_cast<int64_t>(lrow) * p.f_in + c0 + 4 * g;
            PieceRow pin;
            int ish = 31, imk = 0x7fffffff;
            const bool in_pieces = PIECES && p.in_on && k == p.k1 - 1;      // (block-uniform)
            if (in_pieces) {
                pin = piece_row(p.lay_in, lrow);
                ish = p.in_shift;
                imk = (1 << ish) - 1;
                arow = ak + 4 * g;
                brow = bk + 4 * g;
            }
#pragma unroll
            for (int ft = 0; ft < NTI; ++ft) {
                const int64_t poff = in_pieces ? piece_offset(pin, (c0 >> 4) + ft, ish, imk) : static_cast<int64_t>(ft * 16);
                float4 a4 = ldg4(arow + poff), b4 = ldg4(brow + poff);
                if (!lrow_live) {
                    a4 = make_float4(0.f, 0.f, 0.f, 0.f);
                    b4 = make_float4(0.f, 0.f, 0.f, 0.f);
                }
                *reinterpret_cast<float4*>(stage + i * rs + ft * 16 + 4 * g) = a4;
                *reinterpret_cast<float4*>(stage + 16 * rs + i * rs + ft * 16 + 4 * g) = b4;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ft = 0; ft < NTI; ++ft) {
                    av[s][ft] = stage[(4 * g + s) * rs + ft * 16 + i];
                    bv[s][ft] = stage[16 * rs + (4 * g + s) * rs + ft * 16 + i];
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // MFMA step s of phase 2 consumes rows {4 g + s : g = 0..3}; lane (i, g) supplies column i.
                // A_k / B_k come from HBM: fetched now, consumed after phase 1.
                const int row = r0 + 4 * g + s;
                const bool live = row < p.n_rows;
                const int64_t ro = static_cast<int64_t>(live ? row : 0);
#pragma unroll
                for (int ft = 0; ft < NTI; ++ft) {
                    av[s][ft] = live ? ak[ro * p.f_in + c0 + ft * 16 + i] : 0.f;
                    bv[s][ft] = live ? bk[ro * p.f_in + c0 + ft * 16 + i] : 0.f;
                }
            }
        }
        // ---- phase 1: dA_k, dB_k tile = [P | M] (16 x f_out) . W_k^T (f_out x fc) --------------------
        f32x4 acc_a[NTI], acc_b[NTI];
#pragma unroll
        for (int ft = 0; ft < NTI; ++ft) {
            acc_a[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc_b[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        {
            const float* wk = lds + (4 * g) * ws + i;
#pragma unroll
            for (int nt = 0; nt < NTO; ++nt) {
                const float4 x = xg[nt], y = yg[nt];
                const float pp[4] = {x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w};
                const float mm[4] = {y.x - x.x, y.y - x.y, y.z - x.z, y.w - x.w};
                if (XPOSE) {   // P / M row fragments -> staging region (A / B fragments were consumed above)
                    *reinterpret_cast<float4*>(stage + i * rs + nt * 16 + 4 * g) = make_float4(pp[0], pp[1], pp[2], pp[3]);
                    *reinterpret_cast<float4*>(stage + 16 * rs + i * rs + nt * 16 + 4 * g) =
                        make_float4(mm[0], mm[1], mm[2], mm[3]);
                }
                const float* wt = wk + nt * 16 * ws;
#pragma unroll
                for (int m = 0; m < 4; ++m) {
#pragma unroll
                    for (int ft = 0; ft < NTI; ++ft) {
                        const float w = wt[m * ws + ft * 16];
                        acc_a[ft] = mfma(w, pp[m], acc_a[ft]);   // transposed issue, see dense_fwd_kernel
                        acc_b[ft] = mfma(w, mm[m], acc_b[ft]);
                    }
                }
            }
        }
        // ---- phase 2: dW_k[chunk, :] += A_tile^T P + B_tile^T M  (reduction over the 16 rows) ----------
        if (XPOSE) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float pb[NTO], mb[NTO];
            if (XPOSE) {
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) {
                    pb[nt] = stage[(4 * g + s) * rs + nt * 16 + i];
                    mb[nt] = stage[16 * rs + (4 * g + s) * rs + nt * 16 + i];
                }
            } else {
                // column fragments of G for these rows: L1 / L2 hits (the same lines were read for phase 1)
                const int row = r0 + 4 * g + s;
                const bool live = row < p.n_rows;
                const int64_t ro = static_cast<int64_t>(live ? row : 0);
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) {
                    const float x = live ? p.gr[ro * p.ldg + nt * 16 + i] : 0.f;
                    const float y = live ? p.gi[ro * p.ldg + nt * 16 + i] : 0.f;
                    pb[nt] = x + y;
                    mb[nt] = y - x;
                }
            }
            // two sweeps, so that consecutive MFMAs never hit the same accumulator (40-cycle dependent
            // latency vs 32-cycle issue)
#pragma unroll
            for (int ft = 0; ft < NTI; ++ft)
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) acc_w[ft][nt] = mfma(av[s][ft], pb[nt], acc_w[ft][nt]);
#pragma unroll
            for (int ft = 0; ft < NTI; ++ft)
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) acc_w[ft][nt] = mfma(bv[s][ft], mb[nt], acc_w[ft][nt]);
            if (do_bias) {
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) acc_bias[nt] += pb[nt];
            }
        }
        // ---- stores last -----------------------------------------------------------------------------
        if (r0 + i < p.n_rows) {
            if (PIECES && p.out_on && k == p.k1 - 1) {
                // straight into the send buffers of the propagate that takes this term: one store per replica (destination
                // row block) of the column slice the piece belongs to
                const PieceRow po = piece_row(p.lay_out, r0 + i);
                const int osh = p.out_shift, omk = (1 << osh) - 1;
                const int64_t rep_stride = static_cast<int64_t>(p.lay_out.slots_per_blk) * po.slot_stride;
#pragma unroll
                for (int ft = 0; ft < NTI; ++ft) {
                    const int64_t o = piece_offset(po, (c0 >> 4) + ft, osh, omk) + 4 * g;
                    const float4 va = make_float4(acc_a[ft][0], acc_a[ft][1], acc_a[ft][2], acc_a[ft][3]);
                    const float4 vb = make_float4(acc_b[ft][0], acc_b[ft][1], acc_b[ft][2], acc_b[ft][3]);
                    for (int rep = 0; rep < p.lay_out.replicas; ++rep) {
                        *reinterpret_cast<float4*>(dak + o + rep * rep_stride) = va;
                        *reinterpret_cast<float4*>(dbk + o + rep * rep_stride) = vb;
                    }
                }
            } else {
#pragma unroll
                for (int ft = 0; ft < NTI; ++ft) {
                    const int64_t o = static_cast<int64_t>(r0 + i) * p.f_in + c0 + ft * 16 + 4 * g;
                    *reinterpret_cast<float4*>(dak + o) = make_float4(acc_a[ft][0], acc_a[ft][1], acc_a[ft][2], acc_a[ft][3]);
                    *reinterpret_cast<float4*>(dbk + o) = make_float4(acc_b[ft][0], acc_b[ft][1], acc_b[ft][2], acc_b[ft][3]);
                }
            }
        }
    }

    // ---- combine the 4 wavefronts of the block through LDS (W^T is dead now), then one partial ----
    __syncthreads();
    const int wave = tid >> 6;
    constexpr int wsz = fc * fo;   // floats of this block's dW chunk
    // element (f_in index 16 ft + 4 g + r, f_out index 16 nt + i) -> lds[(..) * fo + ..]
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int ft = 0; ft < NTI; ++ft)
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = (ft * 16 + 4 * g + r) * fo + nt * 16 + i;
                        lds[e] = (turn == 0 ? 0.f : lds[e]) + acc_w[ft][nt][r];
                    }
            if (do_bias) {
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) {
                    float v = acc_bias[nt];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    if (g == 0) lds[wsz + nt * 16 + i] = (turn == 0 ? 0.f : lds[wsz + nt * 16 + i]) + v;
                }
            }
        }
        __syncthreads();
    }
    const int64_t pstride = static_cast<int64_t>(p.k1) * p.f_in * p.f_out + p.f_out;
    float* part = p.partial + static_cast<int64_t>(blockIdx.x) * pstride;
    float* dst = part + (static_cast<int64_t>(k) * p.f_in + c0) * p.f_out;
    for (int e = tid; e < wsz; e += 256) dst[e] = lds[e];
    if (do_bias)
        for (int e = tid; e < fo; e += 256) part[pstride - p.f_out + e] = lds[wsz + e];
}

// ---- the backward stage on the bf16 matrix pipe by three-way splitting (round 5) -----------------------------------------
// dense_bwd_kernel<4, 4, true> is paced by its matrix cycles on a chip that is on its power budget (110 TF of exact-fp32 MFMA,
// 0.59 ms for 2.05 GB at the north-star shape).  As csrc/tall.hip's split kernel (the derivation and the measured error are
// there): every fp32 operand = hi + mid + lo in bf16, every product = its six largest partial products on the bf16 pipe with
// fp32 accumulation -- closer to the float64 product than an fp32 fmaf chain, 2.7x fewer matrix cycles.  Same tiling, same
// loads, same LDS transposition, same stores, same partials and reduction as the exact kernel (f_in chunk 64 = NTI 4, f_out 64 /
// 128 = NTO 4 / 8, XPOSE); what changes is the two products:
//   phase 1  dA | dB tile = [P | M] W_k^T: the reduction runs over the 64 output features = 2 blocks of 32 k-slots for
//            v_mfma_f32_16x16x32_bf16.  A lane's 8 k-slots of block kb are the columns 32 kb + 16 h + 4 g + r (h < 2, r < 4) --
//            exactly the two float4 row pieces it already holds (tiles 2 kb and 2 kb + 1); W_k^T sits in LDS pre-split, in
//            fragment order with the same slot map.
//   phase 2  dW_k += A^T P + B^T M: the reduction runs over the tile's 16 rows = ONE v_mfma_f32_16x16x16_bf16 per product
//            term, whose 4 k-slots per lane are rows 4 g + s, s < 4 -- the column fragments the exact kernel feeds to four
//            successive 16x16x4 MFMAs.
struct Triple4 { uint2 h, m, l; };       // four fp32 values as 3 x 4 bf16

__device__ __forceinline__ uint32_t bf16_pair(float lo, float hi)       // one v_cvt_pk_bf16_f32, round to nearest even
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ Triple4 split4(float x0, float x1, float x2, float x3)
{
    Triple4 t;
    const float x[4] = {x0, x1, x2, x3};
    uint32_t hh[2], mm[2], ll[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float a = x[2 * e], b = x[2 * e + 1];
        hh[e] = bf16_pair(a, b);
        const float ra = a - __uint_as_float(hh[e] << 16), rb = b - __uint_as_float(hh[e] & 0xffff0000u);      // exact
        mm[e] = bf16_pair(ra, rb);
        const float sa = ra - __uint_as_float(mm[e] << 16), sb = rb - __uint_as_float(mm[e] & 0xffff0000u);    // exact
        ll[e] = bf16_pair(sa, sb);
    }
    t.h = make_uint2(hh[0], hh[1]);
    t.m = make_uint2(mm[0], mm[1]);
    t.l = make_uint2(ll[0], ll[1]);
    return t;
}

__device__ __forceinline__ bf16x8 octet(uint2 lo, uint2 hi) { return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y)); }
__device__ __forceinline__ s16x4 quad(uint2 v) { return __builtin_bit_cast(s16x4, v); }

// c += the six largest partial products of (w_h + w_m + w_l)(x_h + x_m + x_l), smallest first
__device__ __forceinline__ f32x4 mfma6_32(const bf16x8 (&w)[3], const bf16x8 (&x)[3], f32x4 c)
{
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[2], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[2], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[1], x[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[1], c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[0], x[0], c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 mfma6_16(const Triple4& a, const Triple4& b, f32x4 c)
{
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(quad(a.l), quad(b.h), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(quad(a.h), quad(b.l), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(quad(a.m), quad(b.m), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(quad(a.m), quad(b.h), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(quad(a.h), quad(b.m), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(quad(a.h), quad(b.h), c, 0, 0, 0);
}

template <int NTO, bool PIECES>
__global__ __launch_bounds__(256, (NTO <= 4 ? 2 : 1)) void dense_bwd_split_kernel(DenseBwdArgs p)
{
    static_assert(NTO == 4 || NTO == 8, "f_out = 64 or 128");
    constexpr int NTI = 4, fc = 64, fo = NTO * 16, KBO = NTO / 2;   // KBO: 32-column blocks of the output features
    constexpr int rs = fo + kPad;                                  // row stride of the staging region (fo >= fc)
    constexpr int kFragFloats = KBO * NTI * 3 * 64 * 4;            // W_k^T fragments: [KBO][NTI][3][64] x 16 B
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int k = blockIdx.y;
    const int c0 = static_cast<int>(blockIdx.z) * kChunk;
    uint4* wfrag = reinterpret_cast<uint4*>(lds);
    for (int idx = tid; idx < KBO * NTI * 64; idx += 256) {
        const int lane = idx & 63, ft = (idx >> 6) & 3, kb = idx >> 8;
        const int i = lane & 15, g = lane >> 4;
        const float* wrow = p.w + (static_cast<int64_t>(k) * p.f_in + c0 + 16 * ft + i) * p.f_out + 32 * kb + 4 * g;
        const Triple4 lo = split4(wrow[0], wrow[1], wrow[2], wrow[3]);             // slots 0..3: column 32 kb + 4 g + r
        const Triple4 hi = split4(wrow[16], wrow[17], wrow[18], wrow[19]);         // slots 4..7: column 32 kb + 16 + 4 g + r
        uint4* dst = wfrag + ((kb * NTI + ft) * 3) * 64 + lane;
        dst[0] = make_uint4(lo.h.x, lo.h.y, hi.h.x, hi.h.y);
        dst[64] = make_uint4(lo.m.x, lo.m.y, hi.m.x, hi.m.y);
        dst[128] = make_uint4(lo.l.x, lo.l.y, hi.l.x, hi.l.y);
    }
    __syncthreads();

    const int lane = tid & 63, i = lane & 15, g = lane >> 4;
    float* stage = lds + kFragFloats + (tid >> 6) * (2 * 16 * rs);      // wavefront-private: [2][16][rs]
    const bool do_bias = (k == 0) && (blockIdx.z == 0);
    const float* ak = p.a[k];
    const float* bk = p.b[k];
    float* dak = p.da[k];
    float* dbk = p.db[k];

    f32x4 acc_w[NTI][NTO];
    float acc_bias[NTO];
#pragma unroll
    for (int ft = 0; ft < NTI; ++ft)
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) acc_w[ft][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nt = 0; nt < NTO; ++nt) acc_bias[nt] = 0.f;

    const int n_tiles = (p.n_rows + 15) >> 4;
    const int stride = static_cast<int>(gridDim.x) * 4;
    int tile = static_cast<int>(blockIdx.x) * 4 + (tid >> 6);
    // All of a tile's loads first, as the exact kernel.  (Issuing tile t + 1's loads in the middle of tile t -- behind the dA / dB
    // stores, in front of phase 2 -- was measured: 0.513 against 0.506 ms; at 256 registers the compiler sinks them back to their
    // first use.)
    float4 xg[NTO], yg[NTO], a4[NTI], b4[NTI];
    int n_bad = 0;                                                  // (wavefront-uniform: a scalar register)
    int32_t* const bad_slot = p.bad + ((((static_cast<int64_t>(blockIdx.x) * gridDim.y + blockIdx.y) * gridDim.z + blockIdx.z) * 4 +
                                        (tid >> 6)) * kBadSlot);
    const bool in_pieces = PIECES && p.in_on && k == p.k1 - 1;      // (block-uniform)
    auto rows_in = [&](int tl) {
        const int r0 = tl << 4;
        const int lrow = (r0 + i < p.n_rows) ? r0 + i : p.n_rows - 1;              // clamped; dead rows are zeroed where they are used
        const float* grp = p.gr + static_cast<int64_t>(lrow) * p.ldg + 4 * g;
        const float* gip = p.gi + static_cast<int64_t>(lrow) * p.ldg + 4 * g;
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) {
            xg[nt] = ldg4(grp + nt * 16);
            yg[nt] = ldg4(gip + nt * 16);
        }
        const float* arow = ak + static_cast<int64_t>(lrow) * p.f_in + c0 + 4 * g;
        const float* brow = bk + static_cast<int64_t>(lrow) * p.f_in + c0 + 4 * g;
        PieceRow pin;
        int ish = 31, imk = 0x7fffffff;
        if (in_pieces) {
            pin = piece_row(p.lay_in, lrow);
            ish = p.in_shift;
            imk = (1 << ish) - 1;
            arow = ak + 4 * g;
            brow = bk + 4 * g;
        }
#pragma unroll
        for (int ft = 0; ft < NTI; ++ft) {
            const int64_t poff = in_pieces ? piece_offset(pin, (c0 >> 4) + ft, ish, imk) : static_cast<int64_t>(ft * 16);
            a4[ft] = ldg4(arow + poff);
            b4[ft] = ldg4(brow + poff);
        }
    };
    for (; tile < n_tiles; tile += stride) {
        const int r0 = tile << 4;
        const bool lrow_live = r0 + i < p.n_rows;
        rows_in(tile);
        if (!lrow_live) {                       // rows past the end must contribute nothing to dW
#pragma unroll
            for (int t = 0; t < NTO; ++t) {
                xg[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                yg[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int t = 0; t < NTI; ++t) {
                a4[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                b4[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int ft = 0; ft < NTI; ++ft) {
            *reinterpret_cast<float4*>(stage + i * rs + ft * 16 + 4 * g) = a4[ft];
            *reinterpret_cast<float4*>(stage + 16 * rs + i * rs + ft * 16 + 4 * g) = b4[ft];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // column fragments of A_k / B_k: rows 4 g + s of column 16 ft + i (split further down, beside phase 1's MFMAs)
        float av[NTI][4], bv[NTI][4];
#pragma unroll
        for (int ft = 0; ft < NTI; ++ft) {
            const float* ca = stage + (4 * g) * rs + ft * 16 + i;
            const float* cb = ca + 16 * rs;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                av[ft][e] = ca[e * rs];
                bv[ft][e] = cb[e * rs];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // P / M: row fragments -> staging region (the A / B fragments were consumed above) and, split, phase 1's B operands
        Triple4 pt[NTO], mt[NTO];
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) {
            const float4 x = xg[nt], y = yg[nt];
            const float pp[4] = {x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w};
            const float mm[4] = {y.x - x.x, y.y - x.y, y.z - x.z, y.w - x.w};
            *reinterpret_cast<float4*>(stage + i * rs + nt * 16 + 4 * g) = make_float4(pp[0], pp[1], pp[2], pp[3]);
            *reinterpret_cast<float4*>(stage + 16 * rs + i * rs + nt * 16 + 4 * g) = make_float4(mm[0], mm[1], mm[2], mm[3]);
            pt[nt] = split4(pp[0], pp[1], pp[2], pp[3]);
            mt[nt] = split4(mm[0], mm[1], mm[2], mm[3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- phase 1: dA_k, dB_k tile = [P | M] (16 x f_out) . W_k^T (f_out x 64).  The six partial products are issued TERM-outer,
        //      accumulator-inner: consecutive MFMAs write different accumulators, so none waits for its predecessor's result and
        //      a vector instruction scheduled between two of them costs its own slot only.
        f32x4 acc_a[NTI], acc_b[NTI];
#pragma unroll
        for (int ft = 0; ft < NTI; ++ft) {
            acc_a[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc_b[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        constexpr int kWi[6] = {2, 0, 1, 1, 0, 0}, kXi[6] = {0, 2, 1, 0, 1, 0};      // (w piece, x piece) of the six terms, smallest first
#pragma unroll
        for (int kb = 0; kb < KBO; ++kb) {
            const bf16x8 px[3] = {octet(pt[2 * kb].h, pt[2 * kb + 1].h), octet(pt[2 * kb].m, pt[2 * kb + 1].m),
                                  octet(pt[2 * kb].l, pt[2 * kb + 1].l)};
            const bf16x8 mx[3] = {octet(mt[2 * kb].h, mt[2 * kb + 1].h), octet(mt[2 * kb].m, mt[2 * kb + 1].m),
                                  octet(mt[2 * kb].l, mt[2 * kb + 1].l)};
            // the six partial products of this 32-feature block in accumulators of their own (two f_in tiles at a time: four
            // independent chains), added to the running sums once: a running sum is rounded once per block, not six times
#pragma unroll
            for (int fp = 0; fp < NTI; fp += 2) {
                bf16x8 w[2][3];
                f32x4 part_a[2], part_b[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const uint4* src = wfrag + ((kb * NTI + fp + u) * 3) * 64 + lane;
                    w[u][0] = __builtin_bit_cast(bf16x8, src[0]);
                    w[u][1] = __builtin_bit_cast(bf16x8, src[64]);
                    w[u][2] = __builtin_bit_cast(bf16x8, src[128]);
                    part_a[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    part_b[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        part_a[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][kWi[t]], px[kXi[t]], part_a[u], 0, 0, 0);
                        part_b[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[u][kWi[t]], mx[kXi[t]], part_b[u], 0, 0, 0);
                    }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    acc_a[fp + u] += part_a[u];
                    acc_b[fp + u] += part_b[u];
                }
            }
        }
        // ---- the tile's dA / dB stores (in front of phase 2: their registers are free for it) ------------------------------
        if (r0 + i < p.n_rows) {
            if (PIECES && p.out_on && k == p.k1 - 1) {
                const PieceRow po = piece_row(p.lay_out, r0 + i);
                const int osh = p.out_shift, omk = (1 << osh) - 1;
                const int64_t rep_stride = static_cast<int64_t>(p.lay_out.slots_per_blk) * po.slot_stride;
#pragma unroll
                for (int ft = 0; ft < NTI; ++ft) {
                    const int64_t o = piece_offset(po, (c0 >> 4) + ft, osh, omk) + 4 * g;
                    const float4 va = make_float4(acc_a[ft][0], acc_a[ft][1], acc_a[ft][2], acc_a[ft][3]);
                    const float4 vb = make_float4(acc_b[ft][0], acc_b[ft][1], acc_b[ft][2], acc_b[ft][3]);
                    for (int rep = 0; rep < p.lay_out.replicas; ++rep) {
                        *reinterpret_cast<float4*>(dak + o + rep * rep_stride) = va;
                        *reinterpret_cast<float4*>(dbk + o + rep * rep_stride) = vb;
                    }
                }
            } else {
#pragma unroll
                for (int ft = 0; ft < NTI; ++ft) {
                    const int64_t o = static_cast<int64_t>(r0 + i) * p.f_in + c0 + ft * 16 + 4 * g;
                    *reinterpret_cast<float4*>(dak + o) = make_float4(acc_a[ft][0], acc_a[ft][1], acc_a[ft][2], acc_a[ft][3]);
                    *reinterpret_cast<float4*>(dbk + o) = make_float4(acc_b[ft][0], acc_b[ft][1], acc_b[ft][2], acc_b[ft][3]);
                }
            }
        }
        // ---- operands the split cannot carry (+-inf, NaN, magnitudes above the largest bf16: csrc/tall.hip, any_not_finite) leave
        //      NaN in every sum they enter.  A tile that holds a non-finite sum is NOTED (its number goes into this wavefront's
        //      slot of p.bad) and computed again from the fp32 operands by the kernel that adds the partials
        //      (reduce_dw_checked_kernel), which also looks at the weight-gradient sums of phase 2.  Behind the stores, which
        //      wait for the same sums, and WITHOUT a branch: the tile number is stored every time -- to entry 0, which the count
        //      overwrites at the end, unless the tile is bad.  (A recomputation inside this loop cost 10 registers and 5 - 10 %;
        //      a branch in front of the stores still 5 %: it keeps phase 2's operand splits from being scheduled under the last
        //      MFMAs of phase 1 -- profiles/r6_dense_forms.json.)
        {
            f32x4 z = acc_a[0] * 0.f;                             // 0 for a finite sum, NaN for +-inf and NaN
            z += acc_b[0] * 0.f;
#pragma unroll
            for (int ft = 1; ft < NTI; ++ft) {
                z += acc_a[ft] * 0.f;
                z += acc_b[ft] * 0.f;
            }
            const float c = (z[0] + z[1]) + (z[2] + z[3]);
            const int bad = __builtin_amdgcn_ballot_w64(c != c) != 0 ? 1 : 0;     // wavefront-uniform
            n_bad += bad;
            bad_slot[bad ? (n_bad < kBadSlot ? n_bad : kBadSlot - 1) : 0] = tile;   // (all lanes, one address, one value)
        }
        // ---- phase 2: dW_k[chunk, :] += A_tile^T P + B_tile^T M  (reduction over the 16 rows), term-outer likewise ---------
        Triple4 at[NTI], bt[NTI];
#pragma unroll
        for (int ft = 0; ft < NTI; ++ft) {
            at[ft] = split4(av[ft][0], av[ft][1], av[ft][2], av[ft][3]);
            bt[ft] = split4(bv[ft][0], bv[ft][1], bv[ft][2], bv[ft][3]);
        }
#pragma unroll
        for (int nt = 0; nt < NTO; ++nt) {
            // column fragments of P / M: rows 4 g + e of column 16 nt + i, read where they are used (held for all tiles at once
            // they pushed the piece-layout instance over its 256 registers: 92 bytes of scratch per lane)
            const float* cp = stage + (4 * g) * rs + nt * 16 + i;
            const float* cm = cp + 16 * rs;
            const float pb[4] = {cp[0], cp[rs], cp[2 * rs], cp[3 * rs]};
            const float mb[4] = {cm[0], cm[rs], cm[2 * rs], cm[3 * rs]};
            if (do_bias) {                      // (the exact kernel's order: one add per row slot)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc_bias[nt] += pb[e];
            }
            const Triple4 pc = split4(pb[0], pb[1], pb[2], pb[3]);
            const Triple4 mc = split4(mb[0], mb[1], mb[2], mb[3]);
            const uint2 pcs[3] = {pc.h, pc.m, pc.l}, mcs[3] = {mc.h, mc.m, mc.l};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int ft = 0; ft < NTI; ++ft) {
                    const uint2 as[3] = {at[ft].h, at[ft].m, at[ft].l};
                    acc_w[ft][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(quad(as[kWi[t]]), quad(pcs[kXi[t]]), acc_w[ft][nt], 0, 0, 0);
                }
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int ft = 0; ft < NTI; ++ft) {
                    const uint2 bs[3] = {bt[ft].h, bt[ft].m, bt[ft].l};
                    acc_w[ft][nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(quad(bs[kWi[t]]), quad(mcs[kXi[t]]), acc_w[ft][nt], 0, 0, 0);
                }
        }
    }
    bad_slot[0] = n_bad;                                           // (every wavefront, every launch: the slots are never cleared)

    // ---- combine the 4 wavefronts of the block through LDS (the W fragments are dead now), then one partial ----
    __syncthreads();
    const int wave = tid >> 6;
    constexpr int wsz = fc * fo;
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int ft = 0; ft < NTI; ++ft)
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = (ft * 16 + 4 * g + r) * fo + nt * 16 + i;
                        lds[e] = (turn == 0 ? 0.f : lds[e]) + acc_w[ft][nt][r];
                    }
            if (do_bias) {
#pragma unroll
                for (int nt = 0; nt < NTO; ++nt) {
                    float v = acc_bias[nt];
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    if (g == 0) lds[wsz + nt * 16 + i] = (turn == 0 ? 0.f : lds[wsz + nt * 16 + i]) + v;
                }
            }
        }
        __syncthreads();
    }
    const int64_t pstride = static_cast<int64_t>(p.k1) * p.f_in * p.f_out + p.f_out;
    float* part = p.partial + static_cast<int64_t>(blockIdx.x) * pstride;
    float* dst = part + (static_cast<int64_t>(k) * p.f_in + c0) * p.f_out;
    for (int e = tid; e < wsz; e += 256) dst[e] = lds[e];
    if (do_bias)
        for (int e = tid; e < fo; e += 256) part[pstride - p.f_out + e] = lds[wsz + e];
}

// out[e] = sum_p partial[p][e], fixed order (deterministic): 64 elements per block, 4 partial
// groups per element combined through LDS.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial,
                                                              int n_partials, int64_t stride,
                                                              int64_t n_elem, float* __restrict__ out)
{
    __shared__ float sm[256];
    const int tid = threadIdx.x;
    const int64_t e = static_cast<int64_t>(blockIdx.x) * 64 + (tid & 63);
    const int pg = tid >> 6;
    float acc = 0.f;
    if (e < n_elem) {
#pragma unroll 8
        for (int q = pg; q < n_partials; q += 4) acc += partial[static_cast<int64_t>(q) * stride + e];
    }
    sm[tid] = acc;
    __syncthreads();
    if (pg == 0 && e < n_elem) out[e] = (sm[tid] + sm[tid + 64]) + (sm[tid + 128] + sm[tid + 192]);
}

// dW behind dense_bwd_split_kernel: the partials added as reduce_partials_kernel adds them, and every sum that came out
// non-finite -- an operand the split cannot carry entered it (see that kernel) -- computed again as autograd's matmuls compute it:
// dW_k[c][f] = sum over the rows, in order, of A_k[r][c] P[r][f] + B_k[r][c] M[r][f] on the fp32 operands, P = G_r + G_i,
// M = G_i - G_r.  One compare per element when nothing is wrong.
__global__ __launch_bounds__(256) void reduce_dw_checked_kernel(DenseBwdArgs p, int n_partials, float* __restrict__ out)
{
    __shared__ float sm[256];
    const int tid = threadIdx.x;
    const int64_t n_elem = static_cast<int64_t>(p.k1) * p.f_in * p.f_out, stride = n_elem + p.f_out;
    const int64_t e = static_cast<int64_t>(blockIdx.x) * 64 + (tid & 63);
    const int pg = tid >> 6;
    float acc = 0.f;
    if (e < n_elem) {
#pragma unroll 8
        for (int q = pg; q < n_partials; q += 4) acc += p.partial[static_cast<int64_t>(q) * stride + e];
    }
    sm[tid] = acc;
    __syncthreads();
    if (pg == 0 && e < n_elem) {
        float total = (sm[tid] + sm[tid + 64]) + (sm[tid + 128] + sm[tid + 192]);
        if (!(__builtin_fabsf(total) < __builtin_inff())) {
            const int f = static_cast<int>(e % p.f_out), c = static_cast<int>((e / p.f_out) % p.f_in);
            const int k = static_cast<int>(e / (static_cast<int64_t>(p.f_out) * p.f_in));
            const float* ak = p.a[0];
            const float* bk = p.b[0];
#pragma unroll
            for (int q = 1; q < kMaxOrder; ++q)
                if (q == k) {
                    ak = p.a[q];
                    bk = p.b[q];
                }
            const bool in_pieces = p.in_on && k == p.k1 - 1;
            const int ish = p.in_shift, imk = (1 << ish) - 1;
            double sum = 0.;                                  // float64, rounded once: a long fp32 chain is no product
            for (int r = 0; r < p.n_rows; ++r) {
                int64_t off = static_cast<int64_t>(r) * p.f_in + c;
                if (in_pieces) off = piece_offset(piece_row(p.lay_in, r), c >> 4, ish, imk) + (c & 15);
                const float x = p.gr[static_cast<int64_t>(r) * p.ldg + f], y = p.gi[static_cast<int64_t>(r) * p.ldg + f];
                sum = fma(static_cast<double>(ak[off]), static_cast<double>(x + y), sum);
                sum = fma(static_cast<double>(bk[off]), static_cast<double>(y - x), sum);
            }
            total = static_cast<float>(sum);
        }
        out[e] = total;
    }
    // ---- the tiles the product kernel noted (non-finite dA / dB sums): 16 rows x 64 input features of term k again, from the
    //      fp32 operands -- dA_k[r][c] = sum_f P[r][f] W_k[c][f], dB_k likewise with M -- as fma chains over f in order, summed
    //      in float64 and rounded once (IEEE products and sums: inf / -inf / NaN where autograd's matmuls put them), stored
    //      where the product kernel stored them (every replica of a piece layout).  A slot that overflowed (more noted tiles than
    //      it holds) has ALL tiles of its wavefront redone.  One 4-byte read per slot when nothing is wrong.
    const int gz = (p.f_in + kChunk - 1) / kChunk;
    const int n_slots = n_partials * p.k1 * gz * 4;
    const int n_tiles = (p.n_rows + 15) >> 4;
    // (the counts of 256 slots are read at once, one per thread: walking them one after the other put 16 dependent load
    // latencies -- 20 us -- behind a 5 us reduction)
    __shared__ int counts[256];
    for (int base = static_cast<int>(blockIdx.x) * 256; base < n_slots; base += static_cast<int>(gridDim.x) * 256) {
        const int mine = base + tid;
        const int cnt = mine < n_slots ? p.bad[static_cast<int64_t>(mine) * kBadSlot] : 0;
        if (!__syncthreads_or(cnt != 0)) continue;                         // nothing noted in these 256 slots (the usual case)
        counts[tid] = cnt;
        __syncthreads();
      for (int s_in = 0; s_in < 256; ++s_in) {
        const int count = counts[s_in];
        if (count == 0) continue;                                          // (block-uniform)
        const int slot = base + s_in;
        const int32_t* sl = p.bad + static_cast<int64_t>(slot) * kBadSlot;
        const int wave = slot & 3, z = (slot >> 2) % gz, k = ((slot >> 2) / gz) % p.k1, bx = (slot >> 2) / gz / p.k1;
        const int c0 = z * kChunk;
        const bool all = count >= kBadSlot;
        const int n_list = all ? (n_tiles - (bx * 4 + wave) + n_partials * 4 - 1) / (n_partials * 4) : count;
        for (int j = 0; j < n_list; ++j) {
            const int tile = all ? bx * 4 + wave + j * n_partials * 4 : sl[1 + j];
            const int row = (tile << 4) + (tid >> 4), cq = (tid & 15) * 4;     // 16 rows x 16 column quads
            if (row >= p.n_rows || c0 + cq >= p.f_in) continue;
            const float* grow = p.gr + static_cast<int64_t>(row) * p.ldg;
            const float* girow = p.gi + static_cast<int64_t>(row) * p.ldg;
            const float* wr = p.w + (static_cast<int64_t>(k) * p.f_in + c0 + cq) * p.f_out;
            double sa[4] = {0., 0., 0., 0.}, sb[4] = {0., 0., 0., 0.};
            for (int f = 0; f < p.f_out; ++f) {
                const float x = grow[f], y = girow[f];
                const double pp = x + y, mm = y - x;                       // P and M in fp32, as autograd forms them
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double w = wr[static_cast<int64_t>(r) * p.f_out + f];
                    sa[r] = fma(pp, w, sa[r]);
                    sb[r] = fma(mm, w, sb[r]);
                }
            }
            const float4 va = make_float4(static_cast<float>(sa[0]), static_cast<float>(sa[1]), static_cast<float>(sa[2]),
                                          static_cast<float>(sa[3]));
            const float4 vb = make_float4(static_cast<float>(sb[0]), static_cast<float>(sb[1]), static_cast<float>(sb[2]),
                                          static_cast<float>(sb[3]));
            float* dak = p.da[0];
            float* dbk = p.db[0];
#pragma unroll
            for (int q = 1; q < kMaxOrder; ++q)
                if (q == k) {
                    dak = p.da[q];
                    dbk = p.db[q];
                }
            if (p.out_on && k == p.k1 - 1) {
                const PieceRow po = piece_row(p.lay_out, row);
                const int osh = p.out_shift, omk = (1 << osh) - 1;
                const int64_t rep_stride = static_cast<int64_t>(p.lay_out.slots_per_blk) * po.slot_stride;
                const int64_t o = piece_offset(po, (c0 + cq) >> 4, osh, omk) + (cq & 15);
                for (int rep = 0; rep < p.lay_out.replicas; ++rep) {
                    *reinterpret_cast<float4*>(dak + o + rep * rep_stride) = va;
                    *reinterpret_cast<float4*>(dbk + o + rep * rep_stride) = vb;
                }
            } else {
                const int64_t o = static_cast<int64_t>(row) * p.f_in + c0 + cq;
                *reinterpret_cast<float4*>(dak + o) = va;
                *reinterpret_cast<float4*>(dbk + o) = vb;
            }
        }
      }
        __syncthreads();                                                   // (counts[] is rewritten by the next pass)
    }
}

unsigned row_blocks(int n_rows, unsigned cap)
{
    unsigned g = (static_cast<unsigned>(n_rows) + 63u) / 64u;
    if (g > cap) g = cap;
    return g ? g : 1u;
}

template <typename Kern>
int set_lds(Kern kern, size_t bytes)
{
    if (bytes > 64 * 1024)
        PYGSD_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes)));
    return 0;
}

std::atomic<int>& dense_f32_form();

template <int NT, int FIN, bool PIECES = false>
int launch_fwd_fin(const DenseFwdArgs& a, unsigned gy, size_t lds_bytes, hipStream_t s)
{
    if constexpr (NT == 4 && (FIN == 64 || FIN == 128)) {
        const size_t split_bytes = static_cast<size_t>(a.k1) * (FIN / 32) * NT * 3072;       // pre-split W fragments
        if (dense_f32_form().load() == 0 && split_bytes <= 150 * 1024) {
            if (split_bytes > 80 * 1024) {      // one block per CU: give it 8 wavefronts
                if (int rc = set_lds(dense_fwd_kernel<NT, FIN, 8, PIECES, true>, split_bytes)) return rc;
                hipLaunchKernelGGL((dense_fwd_kernel<NT, FIN, 8, PIECES, true>), dim3(row_blocks(a.n_rows, 1024), gy), dim3(512),
                                   split_bytes, s, a);
                return check_launch("dense_fwd_kernel (split)");
            }
            if (int rc = set_lds(dense_fwd_kernel<NT, FIN, 4, PIECES, true>, split_bytes)) return rc;
            hipLaunchKernelGGL((dense_fwd_kernel<NT, FIN, 4, PIECES, true>), dim3(row_blocks(a.n_rows, 2048), gy), dim3(256),
                               split_bytes, s, a);
            return check_launch("dense_fwd_kernel (split)");
        }
    }
    if (lds_bytes > 80 * 1024) {        // one block per CU: give it 8 wavefronts
        if (int rc = set_lds(dense_fwd_kernel<NT, FIN, 8, PIECES>, lds_bytes)) return rc;
        hipLaunchKernelGGL((dense_fwd_kernel<NT, FIN, 8, PIECES>), dim3(row_blocks(a.n_rows, 1024), gy), dim3(512), lds_bytes, s, a);
        return check_launch("dense_fwd_kernel");
    }
    if (int rc = set_lds(dense_fwd_kernel<NT, FIN, 4, PIECES>, lds_bytes)) return rc;
    hipLaunchKernelGGL((dense_fwd_kernel<NT, FIN, 4, PIECES>), dim3(row_blocks(a.n_rows, 2048), gy), dim3(256), lds_bytes, s, a);
    return check_launch("dense_fwd_kernel");
}

template <int NT>
int launch_fwd(const DenseFwdArgs& a, unsigned gy, hipStream_t s, bool pieces = false)
{
    const size_t lds_bytes = static_cast<size_t>(a.k1) * a.f_in * (NT * 16 + kPad) * sizeof(float);
    PYGSD_REQUIRE(lds_bytes <= 160 * 1024 - 1024, "pygsd_magnetic_dense_fwd_f32: W slice needs %zu B of LDS", lds_bytes);
    if (pieces) {                         // (the entry point admits f_in = 64 / 128 only)
        if (a.f_in == 64) return launch_fwd_fin<NT, 64, true>(a, gy, lds_bytes, s);
        return launch_fwd_fin<NT, 128, true>(a, gy, lds_bytes, s);
    }
    if (a.f_in == 64) return launch_fwd_fin<NT, 64>(a, gy, lds_bytes, s);
    if (a.f_in == 128) return launch_fwd_fin<NT, 128>(a, gy, lds_bytes, s);
    return launch_fwd_fin<NT, 0>(a, gy, lds_bytes, s);
}

// 0 = the split form wherever its shapes allow (default), 1 = every product an fmaf chain on v_mfma_f32_16x16x4_f32;
// PYGSD_DENSE_F32=exact sets 1 at load, pygsd_dense_f32_form changes it at run time (measurement / bitwise tests)
std::atomic<int>& dense_f32_form()
{
    static std::atomic<int> form{[] {
        const char* e = getenv("PYGSD_DENSE_F32");
        return (e && e[0] == 'e') ? 1 : 0;
    }()};
    return form;
}

template <int NTO, bool PIECES>
int launch_bwd_split(const DenseBwdArgs& a, unsigned gx, unsigned gz, hipStream_t s)
{
    constexpr size_t rs = NTO * 16 + kPad;
    const size_t lds_bytes = (static_cast<size_t>((NTO / 2) * 4 * 3 * 64 * 4) + 4 * 2 * 16 * rs) * sizeof(float);
    if (int rc = set_lds(dense_bwd_split_kernel<NTO, PIECES>, lds_bytes)) return rc;
    hipLaunchKernelGGL((dense_bwd_split_kernel<NTO, PIECES>), dim3(gx, a.k1, gz), dim3(256), lds_bytes, s, a);
    return check_launch("dense_bwd_split_kernel");
}

// `split`: bwd_split_form() of this call (the form switch is read once per entry-point call)
template <int NTI, int NTO, bool PIECES = false>
int launch_bwd(const DenseBwdArgs& a, unsigned gx, unsigned gz, bool split, hipStream_t s)
{
    if constexpr (NTI == 4 && (NTO == 4 || NTO == 8)) {
        if (split) return launch_bwd_split<NTO, PIECES>(a, gx, gz, s);
    }
    constexpr bool kXpose = true;
    constexpr size_t rs = (NTO > NTI ? NTO : NTI) * 16 + kPad;
    size_t lds_floats = static_cast<size_t>(NTO * 16) * (NTI * 16 + kPad) + (kXpose ? 4 * 2 * 16 * rs : 0);
    const size_t red = static_cast<size_t>(NTI * 16) * (NTO * 16) + NTO * 16;
    if (red > lds_floats) lds_floats = red;
    const size_t lds_bytes = lds_floats * sizeof(float);
    if (int rc = set_lds(dense_bwd_kernel<NTI, NTO, kXpose, PIECES>, lds_bytes)) return rc;
    hipLaunchKernelGGL((dense_bwd_kernel<NTI, NTO, kXpose, PIECES>), dim3(gx, a.k1, gz), dim3(256), lds_bytes, s, a);
    return check_launch("dense_bwd_kernel");
}

// the piece-layout instances exist for the shapes the sharded layers run: f_in a multiple of 64 (NTI = 4), f_out = 64 / 128
int dispatch_bwd_pieces(const DenseBwdArgs& a, unsigned gx, unsigned gz, bool split, hipStream_t s)
{
    switch (a.f_out / 16) {
        case 4: return launch_bwd<4, 4, true>(a, gx, gz, split, s);
        case 8: return launch_bwd<4, 8, true>(a, gx, gz, split, s);
        default: return fail("pygsd_magnetic_dense_bwd_pieces_f32: f_out=%d (64 or 128 with a piece layout)", a.f_out);
    }
}

template <int NTI>
int dispatch_bwd_nto(const DenseBwdArgs& a, unsigned gx, unsigned gz, bool split, hipStream_t s)
{
    switch (a.f_out / 16) {
        case 1: return launch_bwd<NTI, 1>(a, gx, gz, split, s);
        case 2: return launch_bwd<NTI, 2>(a, gx, gz, split, s);
        case 3: return launch_bwd<NTI, 3>(a, gx, gz, split, s);
        case 4: return launch_bwd<NTI, 4>(a, gx, gz, split, s);
        case 8: return launch_bwd<NTI, 8>(a, gx, gz, split, s);
        default: return fail("pygsd_magnetic_dense_bwd_f32: unsupported f_out=%d", a.f_out);
    }
}

}  // namespace
}  // namespace pygsd

using namespace pygsd;

extern "C" int pygsd_dense_f32_form(int32_t form)
{
    std::atomic<int>& cur = dense_f32_form();
    if (form == 0 || form == 1) return cur.exchange(form);
    return cur.load();
}

extern "C" int pygsd_magnetic_dense_supported(int32_t f_in, int32_t f_out, int32_t k1)
{
    if (k1 < 1 || k1 > kMaxOrder) return 0;
    if (f_in < 16 || f_out < 16 || f_in % 16 || f_out % 16) return 0;
    const int fo16 = f_out / 16;
    if (!(fo16 == 1 || fo16 == 2 || fo16 == 3 || fo16 == 4 || fo16 == 8)) return 0;
    if (f_in % 64 != 0 && f_in > 64) return 0;   // f_in chunks of 64 (or one chunk of 16/32/48)
    if (static_cast<size_t>(k1) * f_in * (kChunk + kPad) * sizeof(float) > 150 * 1024) return 0;
    return 1;
}

extern "C" int pygsd_magnetic_dense_fwd_f32(const float* const* a, const float* const* b, int32_t k1,
                                            const float* w, const float* bias, float* out_real,
                                            float* out_imag, int32_t n_rows, int32_t f_in, int32_t f_out,
                                            void* stream)
{
    return pygsd_magnetic_dense_fwd_pieces_f32(a, b, k1, w, bias, out_real, out_imag, n_rows, f_in, f_out, nullptr, stream);
}

extern "C" int pygsd_magnetic_dense_fwd_pieces_f32(const float* const* a, const float* const* b, int32_t k1, const float* w,
                                                   const float* bias, float* out_real, float* out_imag, int32_t n_rows,
                                                   int32_t f_in, int32_t f_out, const pygsd_piece_layout* last_in, void* stream)
{
    PYGSD_REQUIRE(pygsd_magnetic_dense_supported(f_in, f_out, k1),
                  "pygsd_magnetic_dense_fwd_f32: unsupported shape f_in=%d f_out=%d k1=%d", f_in, f_out, k1);
    PYGSD_REQUIRE(n_rows >= 0, "pygsd_magnetic_dense_fwd_f32: negative size");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(a && b && w && out_real && out_imag, "pygsd_magnetic_dense_fwd_f32: null pointer");
    DenseFwdArgs args{};
    for (int k = 0; k < k1; ++k) {
        PYGSD_REQUIRE(a[k] && b[k] && aligned16(a[k]) && aligned16(b[k]),
                      "pygsd_magnetic_dense_fwd_f32: operand %d null or not 16-byte aligned", k);
        args.a[k] = a[k];
        args.b[k] = b[k];
    }
    args.w = w; args.bias = bias; args.out_r = out_real; args.out_i = out_imag;
    args.n_rows = n_rows; args.f_in = f_in; args.f_out = f_out; args.k1 = k1;
    if (last_in) {
        PYGSD_REQUIRE((f_in == 64 || f_in == 128) && f_out % kChunk == 0,
                      "pygsd_magnetic_dense_fwd_pieces_f32: a piece layout needs f_in = 64 / 128 and f_out a multiple of 64");
        if (int rc = piece_layout_check(last_in, n_rows, f_in, "pygsd_magnetic_dense_fwd_pieces_f32", &args.lay_shift)) return rc;
        args.lay = *last_in;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_DENSE, s);
    const unsigned gy = (static_cast<unsigned>(f_out) + kChunk - 1) / kChunk;
    // every y-chunk is a full 64 columns except possibly a single-chunk narrow output
    if (f_out >= kChunk) {
        PYGSD_REQUIRE(f_out % kChunk == 0 || gy == 1, "pygsd_magnetic_dense_fwd_f32: f_out=%d not chunkable", f_out);
        if (f_out % kChunk == 0) return launch_fwd<4>(args, gy, s, last_in != nullptr);
    }
    switch (f_out / 16) {
        case 1: return launch_fwd<1>(args, 1, s);
        case 2: return launch_fwd<2>(args, 1, s);
        case 3: return launch_fwd<3>(args, 1, s);
        default: return fail("pygsd_magnetic_dense_fwd_f32: f_out=%d not chunkable", f_out);
    }
}

extern "C" int pygsd_magnetic_dense_bwd_workspace(int32_t n_rows, int32_t f_in, int32_t f_out, int32_t k1,
                                                  size_t* bytes)
{
    PYGSD_REQUIRE(bytes, "pygsd_magnetic_dense_bwd_workspace: null output");
    const size_t per = static_cast<size_t>(k1) * f_in * f_out + f_out;
    const size_t gx = row_blocks(n_rows, 256), gz = (static_cast<size_t>(f_in) + kChunk - 1) / kChunk;
    // the partials, then the split form's per-wavefront slots of noted tiles (DenseBwdArgs::bad)
    *bytes = per * sizeof(float) * gx + gx * static_cast<size_t>(k1 > 0 ? k1 : 1) * gz * 4 * kBadSlot * sizeof(int32_t);
    return 0;
}

extern "C" int pygsd_magnetic_dense_bwd_f32(const float* const* a, const float* const* b, int32_t k1,
                                            const float* w, const float* g_real, const float* g_imag, int64_t ldg,
                                            float* const* da, float* const* db, float* dw, float* dbias,
                                            int32_t n_rows, int32_t f_in, int32_t f_out, void* workspace,
                                            size_t workspace_bytes, void* stream)
{
    return pygsd_magnetic_dense_bwd_pieces_f32(a, b, k1, w, g_real, g_imag, ldg, da, db, dw, dbias, n_rows, f_in, f_out, workspace,
                                               workspace_bytes, nullptr, nullptr, stream);
}

extern "C" int pygsd_magnetic_dense_bwd_pieces_f32(const float* const* a, const float* const* b, int32_t k1, const float* w,
                                                   const float* g_real, const float* g_imag, int64_t ldg, float* const* da,
                                                   float* const* db, float* dw, float* dbias, int32_t n_rows, int32_t f_in,
                                                   int32_t f_out, void* workspace, size_t workspace_bytes,
                                                   const pygsd_piece_layout* last_in, const pygsd_piece_layout* last_out,
                                                   void* stream)
{
    PYGSD_REQUIRE(pygsd_magnetic_dense_supported(f_in, f_out, k1),
                  "pygsd_magnetic_dense_bwd_f32: unsupported shape f_in=%d f_out=%d k1=%d", f_in, f_out, k1);
    PYGSD_REQUIRE(n_rows > 0, "pygsd_magnetic_dense_bwd_f32: n_rows must be positive");
    PYGSD_REQUIRE(a && b && w && g_real && g_imag && da && db && dw && dbias && workspace,
                  "pygsd_magnetic_dense_bwd_f32: null pointer");
    PYGSD_REQUIRE(aligned16(g_real) && aligned16(g_imag), "pygsd_magnetic_dense_bwd_f32: gradients not 16-byte aligned");
    PYGSD_REQUIRE(ldg == 0 || (ldg >= f_out && ldg % 4 == 0), "pygsd_magnetic_dense_bwd_f32: ldg must be 0 (one broadcast row) "
                  "or a 16-byte aligned row stride >= f_out (got %lld)", static_cast<long long>(ldg));
    size_t need = 0;
    pygsd_magnetic_dense_bwd_workspace(n_rows, f_in, f_out, k1, &need);
    PYGSD_REQUIRE(workspace_bytes >= need, "pygsd_magnetic_dense_bwd_f32: workspace too small (%zu < %zu)",
                  workspace_bytes, need);
    DenseBwdArgs args{};
    for (int k = 0; k < k1; ++k) {
        PYGSD_REQUIRE(a[k] && b[k] && da[k] && db[k], "pygsd_magnetic_dense_bwd_f32: operand %d null", k);
        args.a[k] = a[k]; args.b[k] = b[k]; args.da[k] = da[k]; args.db[k] = db[k];
    }
    args.gr = g_real; args.gi = g_imag; args.ldg = ldg; args.w = w; args.partial = static_cast<float*>(workspace);
    args.bad = reinterpret_cast<int32_t*>(static_cast<float*>(workspace) +
                                          (static_cast<size_t>(k1) * f_in * f_out + f_out) * row_blocks(n_rows, 256));
    args.n_rows = n_rows; args.f_in = f_in; args.f_out = f_out; args.k1 = k1;
    const bool pieces = last_in || last_out;
    if (pieces) {
        PYGSD_REQUIRE(f_in % kChunk == 0 && (f_out == 64 || f_out == 128),
                      "pygsd_magnetic_dense_bwd_pieces_f32: a piece layout needs f_in a multiple of 64 and f_out = 64 / 128");
        if (last_in) {
            if (int rc = piece_layout_check(last_in, n_rows, f_in, "pygsd_magnetic_dense_bwd_pieces_f32 (in)", &args.in_shift)) return rc;
            args.lay_in = *last_in;
            args.in_on = 1;
        }
        if (last_out) {
            if (int rc = piece_layout_check(last_out, n_rows, f_in, "pygsd_magnetic_dense_bwd_pieces_f32 (out)", &args.out_shift)) return rc;
            args.lay_out = *last_out;
            args.out_on = 1;
        }
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_DENSE_BWD, s);
    const unsigned gx = row_blocks(n_rows, 256);
    const unsigned gz = (static_cast<unsigned>(f_in) + kChunk - 1) / kChunk;
    // the split form's shapes (launch_bwd): f_in in chunks of 64, f_out = 64 / 128; the switch is read once per call
    const bool split = dense_f32_form().load() == 0 && f_in >= kChunk && (f_out == 64 || f_out == 128);
    int rc;
    if (pieces) rc = dispatch_bwd_pieces(args, gx, gz, split, s);
    else if (f_in >= kChunk) rc = dispatch_bwd_nto<4>(args, gx, gz, split, s);
    else if (f_in == 48) rc = dispatch_bwd_nto<3>(args, gx, 1, split, s);
    else if (f_in == 32) rc = dispatch_bwd_nto<2>(args, gx, 1, split, s);
    else rc = dispatch_bwd_nto<1>(args, gx, 1, split, s);
    if (rc) return rc;
    // dw [k1][f_in][f_out] followed by dbias [f_out] in the partial layout
    const int64_t n_w = static_cast<int64_t>(k1) * f_in * f_out;
    if (split)
        hipLaunchKernelGGL(reduce_dw_checked_kernel, dim3(static_cast<unsigned>((n_w + 63) / 64)), dim3(256), 0, s, args,
                           static_cast<int>(gx), dw);
    else
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(static_cast<unsigned>((n_w + 63) / 64)), dim3(256), 0, s,
                           args.partial, static_cast<int>(gx), n_w + f_out, n_w, dw);
    if (int rc2 = check_launch("reduce_partials_kernel")) return rc2;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(static_cast<unsigned>((f_out + 63) / 64)), dim3(256), 0, s,
                       args.partial + n_w, static_cast<int>(gx), n_w + f_out, static_cast<int64_t>(f_out), dbias);
    return check_launch("reduce_partials_kernel");
}
