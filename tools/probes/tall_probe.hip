// Probe (GPU box): what bounds csrc/tall.hip's fp32 tall_linear at K = 128, f_out = 64 (C3a: 100 us for 384 MB and 8.2 GFLOP;
// the fp32 matrix pipe alone would take 52 us, HBM alone 48 us)?  The product kernel's loop, re-stated with switches:
//   MODE 0  as the product kernel (row loads, W fragments from LDS, MFMA, stores)
//   MODE 1  memory only: loads and stores, one add per loaded value instead of the MFMAs
//   MODE 2  matrix + LDS only: every wavefront re-reads rows 0..15 (cache resident) and stores to rows 0..15
//   MODE 3  as 0, W fragments held in registers (no LDS read in the loop)
//   MODE 4  as 0, 32 rows per wavefront: one fragment read feeds two MFMAs
//   MODE 5  matrix only: as 2 with W in registers
//   MODE 6  as 2, the fragment reads of step m + 1 issued before the MFMAs of step m
//   MODE 11 as 0 with the fragment reads of step m + 1 issued before the MFMAs of step m
//   mem_probe<...>: memory only -- product lane order or whole 128-byte lines per load / store instruction, 1 or 2 tiles ahead
//   LIBRARY: pygsd_tall_linear of the built library through its C entry, 128 rows checked bitwise against the host fmaf chain
//            (run with PYGSD_TALL_F32=exact for that check: the library's default fp32 form is the split one, which is not a chain)
//   [MHz]: shader cycles / 100 MHz ticks over block 0's first wavefront (the clock the kernel actually ran at)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/tall_probe tools/probes/tall_probe.hip -ldl && tools/probes/tall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <dlfcn.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Args {
    const float* x;
    int64_t ldx;
    const float* w;      // [K][N]
    float* y;
    int64_t ldy;
    int n_rows;
    long long* clk;      // [2]: shader cycles and 100 MHz ticks of block 0's first wavefront
};

template <int KB, int NT, int MODE>
__global__ __launch_bounds__(256) void probe(Args p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* frag = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int idx = tid; idx < KB * 4 * NT * 64; idx += 256) {
        const int lane = idx & 63, t = (idx >> 6) % NT, m = ((idx >> 6) / NT) & 3, kb = (idx >> 6) / NT / 4;
        const int n = 16 * t + (lane & 15), k = kb * 16 + 4 * (lane >> 4) + m;
        frag[idx] = p.w[k * (NT * 16) + n];
    }
    __syncthreads();
    constexpr int RT = MODE == 4 ? 2 : 1;          // row tiles per wavefront
    const int lane = tid & 63, j = lane & 15, q = lane >> 4;
    const int n_tiles = (p.n_rows + 16 * RT - 1) / (16 * RT);
    const int stride = static_cast<int>(gridDim.x) * 4;
    int tile = static_cast<int>(blockIdx.x) * 4 + (tid >> 6);
    float4 cur[RT][KB], nxt[RT][KB];
    auto row_of = [&](int t, int r) -> int64_t {
        if (MODE == 2 || MODE == 5 || MODE == 6) return j;
        const int64_t row = static_cast<int64_t>(t) * 16 * RT + 16 * r + j;
        return row < p.n_rows ? row : p.n_rows - 1;
    };
    if (tile < n_tiles) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
                cur[r][kb] = *reinterpret_cast<const float4*>(p.x + row_of(tile, r) * p.ldx + 16 * kb + 4 * q);
    }
    constexpr bool WREG = MODE == 3 || MODE == 5;
    float wreg[WREG ? KB * 4 * NT : 1];
    if (WREG) {
#pragma unroll
        for (int i = 0; i < KB * 4 * NT; ++i) wreg[i] = frag[i * 64 + lane];
    }
    for (; tile < n_tiles; tile += stride) {
        const int next = tile + stride;
        if (next < n_tiles) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
                    nxt[r][kb] = *reinterpret_cast<const float4*>(p.x + row_of(next, r) * p.ldx + 16 * kb + 4 * q);
        }
        f32x4 acc[RT][NT];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!WREG) asm volatile("" ::: "memory");
        if (MODE == 6 || MODE == 11) {
            float a_cur[NT], a_nxt[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) a_cur[t] = frag[t * 64 + lane];
#pragma unroll
            for (int s = 0; s < KB * 4; ++s) {
                if (s + 1 < KB * 4) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) a_nxt[t] = frag[((s + 1) * NT + t) * 64 + lane];
                }
                const float4 c = cur[0][s >> 2];
                const float xs[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[t], xs[s & 3], acc[0][t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < NT; ++t) a_cur[t] = a_nxt[t];
            }
        } else
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                if (MODE == 1) {
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        const float xs[4] = {cur[r][kb].x, cur[r][kb].y, cur[r][kb].z, cur[r][kb].w};
                        acc[r][(kb * 4 + m) % NT][m] += xs[m];
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float a = WREG ? wreg[(kb * 4 + m) * NT + t] : frag[((kb * 4 + m) * NT + t) * 64 + lane];
#pragma unroll
                        for (int r = 0; r < RT; ++r) {
                            const float xs[4] = {cur[r][kb].x, cur[r][kb].y, cur[r][kb].z, cur[r][kb].w};
                            acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xs[m], acc[r][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            const int64_t row = (MODE == 2 || MODE == 5 || MODE == 6) ? (blockIdx.x * 64 + (tid >> 6) * 16 + j) : static_cast<int64_t>(tile) * 16 * RT + 16 * r + j;
            if (row < p.n_rows) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    *reinterpret_cast<float4*>(p.y + row * p.ldy + 16 * t + 4 * q) =
                        make_float4(acc[r][t][0], acc[r][t][1], acc[r][t][2], acc[r][t][3]);
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) cur[r][kb] = nxt[r][kb];
    }
    if (blockIdx.x == 0 && tid == 0) { p.clk[0] = clock64() - c0; p.clk[1] = wall_clock64() - w0; }
}

// memory only: the loads and stores of one 16-row tile per wavefront and step, in the product kernel's lane order or in whole
// 128-byte lines (8 lanes per row); DEPTH tiles in flight ahead of the one being stored
template <int KB, int NT, bool LFULL, bool SFULL, bool STORE, int DEPTH>
__global__ __launch_bounds__(256) void mem_probe(Args p)
{
    const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, q = lane >> 4;
    const long long c0 = clock64(), w0 = wall_clock64();
    const int n_tiles = (p.n_rows + 15) / 16;
    const int stride = static_cast<int>(gridDim.x) * 4;
    int tile = static_cast<int>(blockIdx.x) * 4 + (tid >> 6);
    float4 buf[DEPTH][KB];
    auto load = [&](int t, float4* dst) {
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            int64_t row = static_cast<int64_t>(t) * 16 + (LFULL ? (lane >> 3) + 8 * (i & 1) : j);
            if (row >= p.n_rows) row = p.n_rows - 1;
            const int chunk = LFULL ? (lane & 7) + 8 * (i >> 1) : q + 4 * i;
            dst[i] = *reinterpret_cast<const float4*>(p.x + row * p.ldx + 4 * chunk);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (tile + d * stride < n_tiles) load(tile + d * stride, buf[d]);
    for (; tile < n_tiles; tile += stride) {
        float4 sum = buf[0][0];
#pragma unroll
        for (int i = 1; i < KB; ++i) { sum.x += buf[0][i].x; sum.y += buf[0][i].y; sum.z += buf[0][i].z; sum.w += buf[0][i].w; }
#pragma unroll
        for (int d = 0; d + 1 < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < KB; ++i) buf[d][i] = buf[d + 1][i];
        if (tile + DEPTH * stride < n_tiles) load(tile + DEPTH * stride, buf[DEPTH - 1]);
        if (STORE) {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const int64_t row = static_cast<int64_t>(tile) * 16 + (SFULL ? (lane >> 3) + 8 * (i & 1) : j);
                const int chunk = SFULL ? (lane & 7) + 8 * (i >> 1) : q + 4 * i;
                if (row < p.n_rows) *reinterpret_cast<float4*>(p.y + row * p.ldy + 4 * chunk) = sum;
            }
        } else if (sum.x == 12345.678f) {
            p.y[tile] = sum.y;
        }
    }
    if (blockIdx.x == 0 && tid == 0) { p.clk[0] = clock64() - c0; p.clk[1] = wall_clock64() - w0; }
}

template <int KB, int NT, bool LFULL, bool SFULL, bool STORE, int DEPTH>
void run_mem(const char* what, const Args& a, int blocks_per_cu)
{
    auto kern = mem_probe<KB, NT, LFULL, SFULL, STORE, DEPTH>;
    const int64_t n_tiles = (a.n_rows + 15) / 16;
    int64_t grid = (n_tiles + 3) / 4;
    if (grid > 256 * blocks_per_cu) grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f;
    const int reps = 12;
    for (int it = 0; it < reps + 3; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, 0, a);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
    long long clk[2]; CK(hipMemcpy(clk, a.clk, 16, hipMemcpyDeviceToHost));
    printf("[%4.0f MHz] ", clk[1] ? double(clk[0]) / clk[1] * 100.0 : 0.0);
    const double bytes = double(a.n_rows) * (KB * 16 + (STORE ? NT * 16 : 0)) * 4;
    printf("K=%3d N=%3d mem    %-34s grid %5lld (%d/CU) regs %3d  avg %7.1f us  best %7.1f us  %5.2f TB/s\n", KB * 16, NT * 16, what,
           (long long)grid, blocks_per_cu, fa.numRegs, sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) * 1e-12);
}

template <int KB, int NT, int MODE>
void run(const char* what, const Args& a, int blocks_per_cu)
{
    const size_t lds = static_cast<size_t>(KB) * NT * 1024;
    auto kern = probe<KB, NT, MODE>;
    if (lds > 64 * 1024)
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    constexpr int RT = MODE == 4 ? 2 : 1;
    const int64_t n_tiles = (a.n_rows + 16 * RT - 1) / (16 * RT);
    int64_t grid = (n_tiles + 3) / 4;
    if (grid > 256 * blocks_per_cu) grid = 256 * blocks_per_cu;
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
    long long clk[2]; CK(hipMemcpy(clk, a.clk, 16, hipMemcpyDeviceToHost));
    printf("[%4.0f MHz] ", clk[1] ? double(clk[0]) / clk[1] * 100.0 : 0.0);
    const double bytes = double(a.n_rows) * (KB * 16 + NT * 16) * 4, flops = 2.0 * a.n_rows * KB * 16 * NT * 16;
    printf("K=%3d N=%3d mode %d %-34s grid %5lld (%d/CU) regs %3d  avg %7.1f us  best %7.1f us  %5.2f TB/s  %6.1f TF\n", KB * 16,
           NT * 16, MODE, what, (long long)grid, blocks_per_cu, fa.numRegs, sum / reps * 1e3, best * 1e3,
           bytes / (sum / reps * 1e-3) * 1e-12, flops / (sum / reps * 1e-3) * 1e-12);
}

typedef int (*tall_fn)(const void* const*, const int64_t*, const int32_t*, int32_t, const void*, int64_t, int32_t, const void*,
                       void* const*, const int64_t*, const int32_t*, int32_t, int64_t, int32_t, void*);
tall_fn product_entry()
{
    static tall_fn fn = nullptr;
    if (!fn) {
        void* h = dlopen("pytorch_geometric_signed_directed_amd/csrc/libpygsd_hip.so", RTLD_NOW);
        if (!h) { printf("dlopen: %s\n", dlerror()); exit(1); }
        fn = reinterpret_cast<tall_fn>(dlsym(h, "pygsd_tall_linear"));
    }
    return fn;
}

// the product library's kernel through its C entry: timed, and 64 rows checked against a host fmaf chain (bitwise)
template <int KB, int NT>
void run_product(const Args& a, const std::vector<float>& hx, const std::vector<float>& hw)
{
    const void* xs[1] = {a.x}; const int64_t ldx[1] = {a.ldx}; const int32_t wd[1] = {KB * 16};
    void* ys[1] = {a.y}; const int64_t ldy[1] = {a.ldy}; const int32_t ow[1] = {NT * 16};
    tall_fn fn = product_entry();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f;
    const int reps = 12;
    for (int it = 0; it < reps + 3; ++it) {
        CK(hipEventRecord(e0));
        if (fn(xs, ldx, wd, 1, a.w, NT * 16, 0, nullptr, ys, ldy, ow, 1, a.n_rows, 0, nullptr)) { printf("entry failed\n"); exit(1); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 3) { sum += ms; if (ms < best) best = ms; }
    }
    std::vector<float> hy(size_t(64) * NT * 16), tail(size_t(64) * NT * 16);
    CK(hipMemcpy(hy.data(), a.y, hy.size() * 4, hipMemcpyDeviceToHost));
    const int64_t r0 = a.n_rows - 64;
    CK(hipMemcpy(tail.data(), a.y + r0 * a.ldy, tail.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int part = 0; part < 2; ++part)
        for (int r = 0; r < 64; ++r)
            for (int n = 0; n < NT * 16; ++n) {
                const int64_t row = part ? r0 + r : r;
                float acc = 0.f;
                for (int st = 0; st < KB * 4; ++st)          // the kernel's order: step (kb, m), the MFMA's four k-slots
                    for (int qq = 0; qq < 4; ++qq) {
                        const int k = (st >> 2) * 16 + 4 * qq + (st & 3);
                        acc = fmaf(hw[size_t(k) * NT * 16 + n], hx[size_t(row) * KB * 16 + k], acc);
                    }
                const float got = (part ? tail : hy)[size_t(r) * NT * 16 + n];
                if (got != acc) ++bad;
            }
    const double bytes = double(a.n_rows) * (KB * 16 + NT * 16) * 4, flops = 2.0 * a.n_rows * KB * 16 * NT * 16;
    printf("K=%3d N=%3d LIBRARY %-33s %28s avg %7.1f us  best %7.1f us  %5.2f TB/s  %6.1f TF   %d of %d checked values differ\n",
           KB * 16, NT * 16, "pygsd_tall_linear", "", sum / reps * 1e3, best * 1e3, bytes / (sum / reps * 1e-3) * 1e-12,
           flops / (sum / reps * 1e-3) * 1e-12, bad, 2 * 64 * NT * 16);
}

template <int KB, int NT>
void shape(int n_rows)
{
    Args a{};
    float *x, *w, *y;
    const size_t nx = size_t(n_rows) * KB * 16, ny = size_t(n_rows) * NT * 16;
    CK(hipMalloc(&a.clk, 16)); CK(hipMemset(a.clk, 0, 16));
    CK(hipMalloc(&x, nx * 4)); CK(hipMalloc(&y, ny * 4)); CK(hipMalloc(&w, KB * 16 * NT * 16 * 4));
    std::vector<float> h(nx);
    uint32_t s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (int(s >> 8) % 2001 - 1000) * 1e-3f; }
    CK(hipMemcpy(x, h.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h.data(), KB * 16 * NT * 16 * 4, hipMemcpyHostToDevice));
    a.x = x; a.ldx = KB * 16; a.w = w; a.y = y; a.ldy = NT * 16; a.n_rows = n_rows;
    const int by_lds = int((160 * 1024) / (KB * NT * 1024 + 256));
    const int cap = by_lds > 8 ? 8 : by_lds;
    run_product<KB, NT>(a, h, h);
    run<KB, NT, 0>("round 4 form", a, cap);
    run<KB, NT, 2>("matrix + LDS only", a, cap);
    run<KB, NT, 5>("matrix only (W in registers)", a, 2);
    run<KB, NT, 5>("matrix only, 1 block/CU", a, 1);
    run<KB, NT, 6>("matrix + LDS, reads a step ahead", a, cap);
    run<KB, NT, 11>("round 4 form + reads a step ahead", a, cap);
    run_mem<KB, NT, false, false, true, 1>("product order", a, cap);
    run_mem<KB, NT, false, false, true, 1>("product order, 8 blocks/CU", a, 8);
    run_mem<KB, NT, false, false, true, 2>("product order, 2 tiles ahead", a, cap);
    run_mem<KB, NT, false, false, false, 1>("product order, no stores", a, cap);
    run_mem<KB, NT, true, false, true, 1>("line loads", a, cap);
    run_mem<KB, NT, false, true, true, 1>("line stores", a, cap);
    run_mem<KB, NT, true, true, true, 1>("line loads + stores", a, cap);
    run_mem<KB, NT, true, true, true, 1>("line loads + stores, 8 blocks/CU", a, 8);
    run_mem<KB, NT, true, true, true, 2>("line loads + stores, 2 ahead", a, cap);
    run_mem<KB, NT, true, true, false, 1>("line loads, no stores", a, cap);
    run<KB, NT, 0>("round 4 form", a, cap);
    run_product<KB, NT>(a, h, h);
    run<KB, NT, 11>("round 4 form + reads a step ahead", a, cap);
    run_product<KB, NT>(a, h, h);
    CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(w));
}

int main()
{
    shape<8, 4>(500000 - 5);      // C3a dx = [g | g_a] W^T
    shape<4, 8>(500000);      // C3a forward
    shape<4, 12>(2000000);    // C5a forward
    shape<12, 4>(2000000);    // C5a input gradient
    return 0;
}
