// Probe (GPU box): do the fp32 matrix pipe (v_mfma_f32_16x16x4_f32) and the fp32 vector pipe (v_pk_fma_f32) of a gfx950 SIMD run
// concurrently?  Four kernels of the same structure, 4 wavefronts per SIMD: MFMA only, packed FMA only, both interleaved in every
// wavefront, and even wavefronts MFMA / odd wavefronts FMA.  Prints time and TFLOP/s of each.
//   hipcc --offload-arch=gfx950 -O3 -o pipes_probe pipes_probe.hip && ./pipes_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[8];
    f32x2 v[16];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 16; ++i) v[i] = f32x2{float(lane + i), float(lane - i)};
    const float a = 1.0f + lane * 1e-6f, b = 1.0f - lane * 1e-6f;
    const f32x2 m = {a, b}, c = {1e-7f, -1e-7f};
    const bool do_mfma = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
    const bool do_fma = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        if (do_fma) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_elementwise_fma(v[i], m, c);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, float* out, int iters, double flop_per_iter_per_wave_mfma, double flop_per_iter_per_wave_fma)
{
    const int blocks = 256 * 4;                      // 4 blocks of 4 wavefronts per CU: 4 wavefronts per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<MODE><<<blocks, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = blocks * 4.0;
    double flop = 0;
    if (MODE == 0 || MODE == 2) flop += waves * iters * flop_per_iter_per_wave_mfma;
    if (MODE == 1 || MODE == 2) flop += waves * iters * flop_per_iter_per_wave_fma;
    if (MODE == 3) flop += waves / 2 * iters * (flop_per_iter_per_wave_mfma + flop_per_iter_per_wave_fma);
    printf("%-34s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
}

int main()
{
    float* out;
    hipMalloc(&out, 256 * 4 * 256 * sizeof(float));
    const int iters = 20000;
    const double mf = 8 * 2048.0;                    // 8 MFMAs x 16x16x4x2 flop
    const double vf = 64 * 64 * 2 * 2.0;             // 64 packed FMAs x 64 lanes x 2 x 2 flop
    run<0>("mfma only", out, iters, mf, vf);
    run<1>("pk_fma only", out, iters, mf, vf);
    run<2>("both in every wavefront", out, iters, mf, vf);
    run<3>("even waves mfma / odd waves pk_fma", out, iters, mf, vf);
    return 0;
}
