// Probe (GPU box): the fp32 tall product Y = X W on the bf16 matrix pipe by three-way splitting -- every fp32 value is the exact
// sum of three bf16 values up to 2^-24 of its magnitude (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)), the product
// keeps the TERMS largest of the nine partial products (6: everything down to 2^-16 of |x w|; 8: down to 2^-24), accumulated in
// fp32 by v_mfma_f32_16x16x32_bf16 at 16x the rate of the exact v_mfma_f32_16x16x4_f32.  Prints time against the same shapes as
// tall_probe.hip and the error of 256 rows against a float64 product, next to the error of the fp32 fmaf chain (what the product
// library's kernel computes bit for bit).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/split_probe tools/probes/split_probe.hip && tools/probes/split_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

struct Args {
    const float* x;
    int64_t ldx;
    const float* w;      // [K][N]
    float* y;
    int64_t ldy;
    int n_rows;
};

// two fp32 -> two bf16 in one dword (round to nearest even): one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack2(float lo, float hi)
{
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// x[0..7] -> (hi, mid, lo) bf16 octets
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

// KB: k-blocks of 32; NT: output tiles of 16; TERMS: 6 or 8; SEP: the small terms in accumulators of their own
template <int KB, int NT, int TERMS, bool SEP>
__global__ __launch_bounds__(256) void split_kernel(Args p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* frag = reinterpret_cast<uint4*>(smem);                      // [KB][NT][3][64] x 8 bf16
    const int tid = threadIdx.x;
    for (int idx = tid; idx < KB * NT * 64; idx += 256) {
        const int lane = idx & 63, t = (idx >> 6) % NT, kb = (idx >> 6) / NT;
        const int n = 16 * t + (lane & 15), k0 = kb * 32 + 8 * (lane >> 4);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.w[(k0 + e) * (NT * 16) + n];
        uint4 h, m, l;
        split8(v, h, m, l);
        uint4* dst = frag + ((kb * NT + t) * 3) * 64 + lane;
        dst[0] = h; dst[64] = m; dst[128] = l;
    }
    __syncthreads();
    const int lane = tid & 63, j = lane & 15, q = lane >> 4;
    const int n_tiles = (p.n_rows + 15) >> 4;
    const int stride = static_cast<int>(gridDim.x) * 4;
    int tile = static_cast<int>(blockIdx.x) * 4 + (tid >> 6);
    float4 cur[KB][2], nxt[KB][2];
    auto load = [&](int t, float4 (&dst)[KB][2]) {
        int64_t row = static_cast<int64_t>(t) * 16 + j;
        row = row < p.n_rows ? row : p.n_rows - 1;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const float* src = p.x + row * p.ldx + kb * 32 + 8 * q;
            dst[kb][0] = *reinterpret_cast<const float4*>(src);
            dst[kb][1] = *reinterpret_cast<const float4*>(src + 4);
        }
    };
    if (tile < n_tiles) load(tile, cur);
    for (; tile < n_tiles; tile += stride) {
        if (tile + stride < n_tiles) load(tile + stride, nxt);
        f32x4 acc[NT], small[SEP ? NT : 1];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < (SEP ? NT : 1); ++t) small[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" ::: "memory");
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const float xs[8] = {cur[kb][0].x, cur[kb][0].y, cur[kb][0].z, cur[kb][0].w,
                                 cur[kb][1].x, cur[kb][1].y, cur[kb][1].z, cur[kb][1].w};
            uint4 xh4, xm4, xl4;
            split8(xs, xh4, xm4, xl4);
            const bf16x8 xh = __builtin_bit_cast(bf16x8, xh4), xm = __builtin_bit_cast(bf16x8, xm4),
                         xl = __builtin_bit_cast(bf16x8, xl4);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const uint4* src = frag + ((kb * NT + t) * 3) * 64 + lane;
                const bf16x8 wh = __builtin_bit_cast(bf16x8, src[0]), wm = __builtin_bit_cast(bf16x8, src[64]),
                             wl = __builtin_bit_cast(bf16x8, src[128]);
                f32x4& s = SEP ? small[t] : acc[t];
                // smallest terms first
                if (TERMS >= 8) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xm, s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xl, s, 0, 0, 0);
                }
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, s, 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[t], 0, 0, 0);
            }
        }
        if (static_cast<int64_t>(tile) * 16 + j < p.n_rows) {
            const int64_t row = static_cast<int64_t>(tile) * 16 + j;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                f32x4 v = acc[t];
                if (SEP) v += small[t];
                *reinterpret_cast<float4*>(p.y + row * p.ldy + 16 * t + 4 * q) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) { cur[kb][0] = nxt[kb][0]; cur[kb][1] = nxt[kb][1]; }
    }
}


// ---- the same product with the memory-side measures of tall_probe.hip's experiment: whole-line stores (lanes j and j ^ 8 trade one
// tile of each pair through a DPP rotation), clamped instead of masked rows (every load and store is issued: exact vmcnt waits),
// two row buffers that trade roles with the first pair of steps written out before the loop -------------------------------------
__device__ __forceinline__ float rotate8(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xf, 0xf, true));
}

template <int KB>
__device__ __forceinline__ void rows_in(const Args& p, int tile, int j, int q, float4 (&dst)[KB][2])
{
    int64_t row = static_cast<int64_t>(tile) * 16 + j;
    row = row < p.n_rows ? row : p.n_rows - 1;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const float* src = p.x + row * p.ldx + kb * 32 + 8 * q;
        dst[kb][0] = *reinterpret_cast<const float4*>(src);
        dst[kb][1] = *reinterpret_cast<const float4*>(src + 4);
    }
}

template <int KB, int NT, bool SEP, bool LINES>
__device__ __forceinline__ void tile_out(const Args& p, const uint4* frag, int tile, int lane, const float4 (&cur)[KB][2])
{
    const int j = lane & 15, q = lane >> 4;
    f32x4 acc[NT], small[SEP ? NT : 1];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < (SEP ? NT : 1); ++t) small[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("" ::: "memory");
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const float xs[8] = {cur[kb][0].x, cur[kb][0].y, cur[kb][0].z, cur[kb][0].w,
                             cur[kb][1].x, cur[kb][1].y, cur[kb][1].z, cur[kb][1].w};
        uint4 xh4, xm4, xl4;
        split8(xs, xh4, xm4, xl4);
        const bf16x8 xh = __builtin_bit_cast(bf16x8, xh4), xm = __builtin_bit_cast(bf16x8, xm4), xl = __builtin_bit_cast(bf16x8, xl4);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const uint4* src = frag + ((kb * NT + t) * 3) * 64 + lane;
            const bf16x8 wh = __builtin_bit_cast(bf16x8, src[0]), wm = __builtin_bit_cast(bf16x8, src[64]),
                         wl = __builtin_bit_cast(bf16x8, src[128]);
            f32x4& s = SEP ? small[t] : acc[t];
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, s, 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, acc[t], 0, 0, 0);
        }
    }
    if (SEP) {
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] += small[t];
    }
    const int64_t last = p.n_rows - 1;
    if (LINES) {
        const bool upper = j >= 8;
        int64_t row_a = static_cast<int64_t>(tile) * 16 + (j & 7), row_b = row_a + 8;
        row_a = row_a < last ? row_a : last;
        row_b = row_b < last ? row_b : last;
#pragma unroll
        for (int m = 0; m < NT / 2; ++m) {
            float va[4], vb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float lo = acc[2 * m][r], hi = acc[2 * m + 1][r];
                const float lo_far = rotate8(lo), hi_far = rotate8(hi);
                va[r] = upper ? hi_far : lo;
                vb[r] = upper ? hi : lo_far;
            }
            const int col = 32 * m + (upper ? 16 : 0) + 4 * q;
            *reinterpret_cast<float4*>(p.y + row_a * p.ldy + col) = make_float4(va[0], va[1], va[2], va[3]);
            *reinterpret_cast<float4*>(p.y + row_b * p.ldy + col) = make_float4(vb[0], vb[1], vb[2], vb[3]);
        }
    } else {
        int64_t row = static_cast<int64_t>(tile) * 16 + j;
        row = row < last ? row : last;
#pragma unroll
        for (int t = 0; t < NT; ++t)
            *reinterpret_cast<float4*>(p.y + row * p.ldy + 16 * t + 4 * q) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    }
}

template <int KB, int NT, bool SEP, bool LINES>
__global__ __launch_bounds__(256) void split_tuned_kernel(Args p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* frag = reinterpret_cast<uint4*>(smem);
    const int tid = threadIdx.x;
    for (int idx = tid; idx < KB * NT * 64; idx += 256) {
        const int lane = idx & 63, t = (idx >> 6) % NT, kb = (idx >> 6) / NT;
        const int n = 16 * t + (lane & 15), k0 = kb * 32 + 8 * (lane >> 4);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = p.w[(k0 + e) * (NT * 16) + n];
        uint4 h, m, l;
        split8(v, h, m, l);
        uint4* dst = frag + ((kb * NT + t) * 3) * 64 + lane;
        dst[0] = h; dst[64] = m; dst[128] = l;
    }
    __syncthreads();
    const int lane = tid & 63, j = lane & 15, q = lane >> 4;
    const int n_tiles = (p.n_rows + 15) >> 4;
    const int stride = static_cast<int>(gridDim.x) * 4;
    int tile = static_cast<int>(blockIdx.x) * 4 + (tid >> 6);
    if (tile >= n_tiles) return;
    const int last_tile = n_tiles - 1;
    float4 rows_a[KB][2], rows_b[KB][2];
    rows_in<KB>(p, tile, j, q, rows_a);
    rows_in<KB>(p, min(tile + stride, last_tile), j, q, rows_b);
#define STEP(ROWS)                                                        \
    tile_out<KB, NT, SEP, LINES>(p, frag, tile, lane, ROWS);              \
    rows_in<KB>(p, min(tile + 2 * stride, last_tile), j, q, ROWS);        \
    tile += stride;                                                       \
    if (tile >= n_tiles) return;
    STEP(rows_a)
    STEP(rows_b)
    for (;;) {
        STEP(rows_a)
        STEP(rows_b)
    }
#undef STEP
}


template <int KB, int NT, typename Kern>
void run(Kern kern, const char* what, const Args& a, const std::vector<float>& hx, const std::vector<float>& hw,
         const std::vector<double>& y64, const std::vector<double>& scale)
{
    const size_t lds = static_cast<size_t>(KB) * NT * 3 * 1024;
    CK(hipMemset(a.y, 0, size_t(256) * NT * 16 * 4));
    if (lds > 64 * 1024)
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = int((160 * 1024) / lds); if (per_cu > 8) per_cu = 8; if (per_cu < 1) per_cu = 1;
    const int64_t n_tiles = (a.n_rows + 15) / 16;
    int64_t grid = (n_tiles + 3) / 4;
    if (grid > 256 * per_cu) grid = 256 * per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f;
    const int reps = 12;
    for (int it = 0; it < reps + 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
    const int N = NT * 16, K = KB * 32, R = 256;
    std::vector<float> hy(size_t(R) * N);
    CK(hipMemcpy(hy.data(), a.y, hy.size() * 4, hipMemcpyDeviceToHost));
    double e_split = 0, e_chain = 0;
    for (int r = 0; r < R; ++r)
        for (int n = 0; n < N; ++n) {
            float chain = 0.f;
            for (int k = 0; k < K; ++k) chain = fmaf(hw[size_t(k) * N + n], hx[size_t(r) * K + k], chain);
            const double want = y64[size_t(r) * N + n], sc = scale[size_t(r) * N + n];     // sc = sum |x||w|
            e_split = fmax(e_split, fabs(hy[size_t(r) * N + n] - want) / sc);
            e_chain = fmax(e_chain, fabs(chain - want) / sc);
        }
    const double bytes = double(a.n_rows) * (K + N) * 4;
    printf("K=%3d N=%3d %-36s grid %5lld (%d/CU) regs %3d lds %3zu KB  avg %7.1f us  best %7.1f us  %5.2f TB/s   "
           "max |err| / sum|x||w|: split %.2e   fp32 fmaf chain %.2e\n", K, N, what, (long long)grid, per_cu, fa.numRegs, lds / 1024,
           sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) * 1e-12, e_split, e_chain);
}

template <int KB, int NT>
void shape(int n_rows, int wide)
{
    Args a{};
    const int K = KB * 32, N = NT * 16;
    float *x, *w, *y;
    const size_t nx = size_t(n_rows) * K, ny = size_t(n_rows) * N;
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&w, size_t(K) * N * 4));
    std::vector<float> hx(nx), hw(size_t(K) * N);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (int(s >> 8) % 200001 - 100000) * 1e-5f; };
    for (auto& v : hx) { v = rnd(); if (wide) v *= exp2f(float(int(s >> 27) - 16)); }      // wide: magnitudes over 2^-16 .. 2^15
    for (auto& v : hw) { v = rnd(); if (wide) v *= exp2f(float(int(s >> 28) - 8)); }
    CK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    a.x = x; a.ldx = K; a.w = w; a.y = y; a.ldy = N; a.n_rows = n_rows;
    const int R = 256;
    std::vector<double> y64(size_t(R) * N), scale(size_t(R) * N);
    for (int r = 0; r < R; ++r)
        for (int n = 0; n < N; ++n) {
            double acc = 0, sc = 0;
            for (int k = 0; k < K; ++k) {
                const double t = double(hw[size_t(k) * N + n]) * double(hx[size_t(r) * K + k]);
                acc += t; sc += fabs(t);
            }
            y64[size_t(r) * N + n] = acc; scale[size_t(r) * N + n] = sc > 0 ? sc : 1;
        }
    printf("-- %s inputs\n", wide ? "wide-range" : "uniform [-1, 1]");
    run<KB, NT>(split_kernel<KB, NT, 6, false>, "6 terms", a, hx, hw, y64, scale);
    run<KB, NT>(split_kernel<KB, NT, 6, true>, "6 terms, small apart", a, hx, hw, y64, scale);
    run<KB, NT>(split_tuned_kernel<KB, NT, false, false>, "6 terms, exact waits", a, hx, hw, y64, scale);
    run<KB, NT>(split_tuned_kernel<KB, NT, false, true>, "6 terms, exact waits, lines", a, hx, hw, y64, scale);
    run<KB, NT>(split_tuned_kernel<KB, NT, true, true>, "same, small apart", a, hx, hw, y64, scale);
    run<KB, NT>(split_kernel<KB, NT, 6, false>, "6 terms (again)", a, hx, hw, y64, scale);
    run<KB, NT>(split_tuned_kernel<KB, NT, false, true>, "6 terms, exact waits, lines (again)", a, hx, hw, y64, scale);
    CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(w));
}

int main()
{
    shape<4, 4>(500000 - 5, 0);     // C3a dx = [g | g_a] W^T (K = 128, f_out = 64)
    shape<4, 4>(500000 - 5, 1);
    shape<2, 8>(500000, 0);         // C3a forward (K = 64, f_out = 128)
    shape<2, 12>(2000000, 0);       // C5a forward
    shape<6, 4>(2000000, 0);        // C5a input gradient
    return 0;
}
