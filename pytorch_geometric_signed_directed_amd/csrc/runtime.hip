// Error string + kernel-timing recorder of libpygsd_hip.so (include/pygsd_hip.h: pygsd_version,
// pygsd_last_error, pygsd_prof_*).
#include "common.hpp"

#include <cstdlib>

namespace pygsd {

std::string& last_error()
{
    static thread_local std::string err;
    return err;
}

int fail(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return 1;
}

namespace {
struct Rec {
    int id;
    hipEvent_t start, stop;
};
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;

hipEvent_t take_event()
{
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
}  // namespace

ProfScope::ProfScope(int kernel_id, hipStream_t stream) : id_(kernel_id), stream_(stream)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (!g_on) return;
    start_ = take_event();
    if (start_) (void)hipEventRecord(start_, stream_);
}

ProfScope::~ProfScope()
{
    if (!start_) return;
    std::lock_guard<std::mutex> lk(g_mu);
    hipEvent_t stop = take_event();
    if (!stop) {
        g_pool.push_back(start_);
        return;
    }
    (void)hipEventRecord(stop, stream_);
    g_recs.push_back({id_, start_, stop});
}

}  // namespace pygsd

using namespace pygsd;

namespace {
constexpr int kBlock = 256;

// Streaming copy, 16 bytes per lane, non-temporal both ways: the achievable-HBM yardstick bench.py prices
// the gather kernels against (MI355X_MICROARCH.md: "6.29 TB/s measured (float4 copy)").
typedef float vec4f __attribute__((ext_vector_type(4)));

// MODE bit 0: non-temporal loads, bit 1: non-temporal stores.  UN float4 per lane in flight before the first store.
template <int MODE, int UN>
__global__ __launch_bounds__(kBlock) void stream_copy_kernel(const vec4f* __restrict__ src,
                                                            vec4f* __restrict__ dst, int64_t n4)
{
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    int64_t i = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    for (; i + (UN - 1) * stride < n4; i += UN * stride) {
        vec4f v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u)
            v[u] = (MODE & 1) ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (MODE & 2) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

// Occupies the stream for `ticks` of the 100 MHz constant-rate counter: stands in for the wire time of an xGMI
// exchange when the sharded propagate is rehearsed on ONE GPU (tools/emulate_sharded.py).
__global__ void spin_kernel(long long ticks)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// min / max of an id list: 16-byte loads (two ids per lane), 4 in flight, ONE 64-bit atomic pair per block --
// the atomics all hit one address, so their count, not the loads, set the time (8 k wavefront atomics: 0.2 ms).
__global__ __launch_bounds__(kBlock) void id_range_kernel(const int64_t* __restrict__ ids, int64_t n,
                                                         long long* __restrict__ minmax)
{
    long long lo = 0x7fffffffffffffffLL, hi = -0x7fffffffffffffffLL - 1;
    const int64_t stride = static_cast<int64_t>(gridDim.x) * kBlock;
    const int64_t tid = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x;
    const bool vec = (reinterpret_cast<uintptr_t>(ids) & 15u) == 0;
    const int64_t n2 = vec ? n / 2 : 0;
    const longlong2* p2 = reinterpret_cast<const longlong2*>(ids);
    int64_t i = tid;
    for (; i + 3 * stride < n2; i += 4 * stride) {
        longlong2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = p2[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lo = v[u].x < lo ? v[u].x : lo;
            lo = v[u].y < lo ? v[u].y : lo;
            hi = v[u].x > hi ? v[u].x : hi;
            hi = v[u].y > hi ? v[u].y : hi;
        }
    }
    for (; i < n2; i += stride) {
        const longlong2 v = p2[i];
        lo = v.x < lo ? v.x : lo;
        lo = v.y < lo ? v.y : lo;
        hi = v.x > hi ? v.x : hi;
        hi = v.y > hi ? v.y : hi;
    }
    for (int64_t k = 2 * n2 + tid; k < n; k += stride) {     // odd tail, or everything when unaligned
        const long long v = ids[k];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const long long l2 = __shfl_xor(lo, off), h2 = __shfl_xor(hi, off);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    __shared__ long long s_lo[kBlock / 64], s_hi[kBlock / 64];
    if ((threadIdx.x & 63) == 0) {
        s_lo[threadIdx.x >> 6] = lo;
        s_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 1; w < kBlock / 64; ++w) {
            lo = s_lo[w] < lo ? s_lo[w] : lo;
            hi = s_hi[w] > hi ? s_hi[w] : hi;
        }
        atomicMin(minmax, lo);
        atomicMax(minmax + 1, hi);
    }
}
}  // namespace

extern "C" int pygsd_stream_copy_f32(const float* src, float* dst, int64_t n, void* stream)
{
    PYGSD_REQUIRE(n >= 0 && n % 4 == 0, "pygsd_stream_copy_f32: n must be a non-negative multiple of 4");
    if (n == 0) return 0;
    PYGSD_REQUIRE(src && dst, "pygsd_stream_copy_f32: null pointer");
    PYGSD_REQUIRE(aligned16(src) && aligned16(dst), "pygsd_stream_copy_f32: pointers must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    const int64_t n4 = n / 4;
    // Defaults = the fastest shape of tools/copy_probe.py on MI355X (profiles/r2_copy_probe.json): plain loads and
    // stores, ONE float4 per lane over an uncapped grid (the unrolled main loop never runs then) -- 6.16 TB/s (persistent grids of 4..64 blocks per
    // CU and non-temporal variants: 4.2..5.4 TB/s; torch's copy_: 4.55 TB/s).  PYGSD_COPY_MODE (0 plain, 1 nt loads,
    // 2 nt stores, 3 both) and PYGSD_COPY_BLOCKS_PER_CU exist for that probe only.
    const char* m = getenv("PYGSD_COPY_MODE");
    const char* b = getenv("PYGSD_COPY_BLOCKS_PER_CU");
    const int mode = m ? atoi(m) : 0;
    const int64_t per_cu = b ? atoi(b) : (int64_t(1) << 20);
    int64_t blocks = (n4 + kBlock - 1) / kBlock;
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    const dim3 grid(static_cast<unsigned>(blocks)), block(kBlock);
    const vec4f* sp = reinterpret_cast<const vec4f*>(src);
    vec4f* dp = reinterpret_cast<vec4f*>(dst);
    switch (mode & 3) {
        case 0: hipLaunchKernelGGL((stream_copy_kernel<0, 4>), grid, block, 0, s, sp, dp, n4); break;
        case 1: hipLaunchKernelGGL((stream_copy_kernel<1, 4>), grid, block, 0, s, sp, dp, n4); break;
        case 2: hipLaunchKernelGGL((stream_copy_kernel<2, 4>), grid, block, 0, s, sp, dp, n4); break;
        default: hipLaunchKernelGGL((stream_copy_kernel<3, 4>), grid, block, 0, s, sp, dp, n4); break;
    }
    return check_launch("stream_copy_kernel");
}

extern "C" int pygsd_spin_us(double microseconds, void* stream)
{
    PYGSD_REQUIRE(microseconds >= 0.0 && microseconds <= 5e6, "pygsd_spin_us: duration outside [0, 5 s]");
    int rate_khz = 0;
    int dev = 0;
    PYGSD_HIP_TRY(hipGetDevice(&dev));
    PYGSD_HIP_TRY(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev));
    if (rate_khz <= 0) rate_khz = 100000;    // gfx9: 100 MHz constant counter
    const long long ticks = static_cast<long long>(microseconds * 1e-3 * rate_khz);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), ticks);
    return check_launch("spin_kernel");
}

extern "C" int pygsd_id_range_i64(const int64_t* ids, int64_t n, int64_t* minmax, void* stream)
{
    PYGSD_REQUIRE(n >= 0, "pygsd_id_range_i64: negative size");
    PYGSD_REQUIRE(minmax, "pygsd_id_range_i64: null output");
    if (n == 0) return 0;
    PYGSD_REQUIRE(ids, "pygsd_id_range_i64: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_BUILD, s);
    const int64_t blocks = (n / 2 + kBlock - 1) / kBlock + 1;
    hipLaunchKernelGGL(id_range_kernel, dim3(static_cast<unsigned>(blocks < 1024 ? blocks : 1024)), dim3(kBlock), 0, s,
                       ids, n, reinterpret_cast<long long*>(minmax));
    return check_launch("id_range_kernel");
}

extern "C" int pygsd_version(void) { return PYGSD_ABI_VERSION; }

extern "C" const char* pygsd_last_error(void) { return last_error().c_str(); }

extern "C" int pygsd_prof_enable(int32_t on)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return 0;
}

extern "C" int pygsd_prof_reset(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.stop);
        g_pool.push_back(r.start);
        g_pool.push_back(r.stop);
    }
    g_recs.clear();
    return 0;
}

extern "C" int pygsd_prof_collect(int32_t kernel_id, int64_t* launches, double* total_ms)
{
    PYGSD_REQUIRE(launches && total_ms, "pygsd_prof_collect: null output pointer");
    PYGSD_REQUIRE(kernel_id >= 0 && kernel_id < PYGSD_K_COUNT, "pygsd_prof_collect: bad kernel id %d",
                  kernel_id);
    std::lock_guard<std::mutex> lk(g_mu);
    int64_t n = 0;
    double ms = 0.0;
    for (auto& r : g_recs) {
        if (r.id != kernel_id) continue;
        PYGSD_HIP_TRY(hipEventSynchronize(r.stop));
        float t = 0.f;
        PYGSD_HIP_TRY(hipEventElapsedTime(&t, r.start, r.stop));
        ms += t;
        ++n;
    }
    *launches = n;
    *total_ms = ms;
    return 0;
}
