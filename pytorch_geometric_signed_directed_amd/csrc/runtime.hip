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

// Packing for the sharded exchanges (parallel.PropagateEngine): G feature groups [n_rows, row_bytes] ->
// out[phase c][replica i][slice j][row t][group g][slice bytes], i.e. for every column phase the all-to-all send buffer
// whose chunk d = i * p_c + j holds column slice j of sub-range c of the local rows, groups side by side.  One thread
// per 16-byte unit of the INPUT: read once, written p_r times (the replicas an equal-split all-to-all needs).
struct PackArgs {
    const uint4* x[4];
    uint4* out;
    int64_t ld_units;        // input row stride, 16-byte units
    int32_t groups, n_rows, row_units, p_r, p_c, phases;
};

__global__ __launch_bounds__(kBlock) void pack_slices_kernel(PackArgs a)
{
    const int64_t per_group = static_cast<int64_t>(a.n_rows) * a.row_units;
    const int64_t total = per_group * a.groups;
    const int n_sub = a.n_rows / a.phases;
    const int sl = a.row_units / a.p_c;                    // units per column slice
    const int64_t chunk = static_cast<int64_t>(n_sub) * a.groups * sl;          // one (phase, replica, slice) chunk
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * kBlock + threadIdx.x; idx < total;
         idx += static_cast<int64_t>(gridDim.x) * kBlock) {
        const int g = static_cast<int>(idx / per_group);
        const int64_t rem = idx - g * per_group;
        const int row = static_cast<int>(rem / a.row_units);
        const int u = static_cast<int>(rem - static_cast<int64_t>(row) * a.row_units);
        const uint4 v = a.x[g][static_cast<int64_t>(row) * a.ld_units + u];
        const int c = row / n_sub, t = row - c * n_sub;
        const int j = u / sl, k = u - j * sl;
        const int64_t inner = (static_cast<int64_t>(t) * a.groups + g) * sl + k;
        for (int i = 0; i < a.p_r; ++i) {
            const int64_t d = static_cast<int64_t>(c) * a.p_r * a.p_c + static_cast<int64_t>(i) * a.p_c + j;
            a.out[d * chunk + inner] = v;
        }
    }
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

extern "C" int pygsd_pack_slices(const void* const* xs, int32_t groups, int32_t n_rows, int32_t row_bytes,
                                 int64_t ld_bytes, int32_t p_r, int32_t p_c, int32_t phases, void* out, void* stream)
{
    PYGSD_REQUIRE(groups >= 1 && groups <= 4 && n_rows >= 0 && row_bytes >= 0 && p_r >= 1 && p_c >= 1 && phases >= 1,
                  "pygsd_pack_slices: bad sizes (groups 1..4)");
    if (n_rows == 0 || row_bytes == 0) return 0;
    PYGSD_REQUIRE(xs && out && aligned16(out), "pygsd_pack_slices: null or unaligned output");
    PYGSD_REQUIRE(n_rows % phases == 0, "pygsd_pack_slices: %d rows do not split into %d phases", n_rows, phases);
    PYGSD_REQUIRE(row_bytes % (16 * p_c) == 0 && ld_bytes % 16 == 0 && ld_bytes >= row_bytes,
                  "pygsd_pack_slices: column slices must be multiples of 16 bytes (row %d B, %d slices)", row_bytes, p_c);
    PackArgs a{};
    for (int g = 0; g < groups; ++g) {
        PYGSD_REQUIRE(xs[g] && aligned16(xs[g]), "pygsd_pack_slices: group %d null or not 16-byte aligned", g);
        a.x[g] = static_cast<const uint4*>(xs[g]);
    }
    a.out = static_cast<uint4*>(out);
    a.ld_units = ld_bytes / 16;
    a.groups = groups; a.n_rows = n_rows; a.row_units = row_bytes / 16; a.p_r = p_r; a.p_c = p_c; a.phases = phases;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    const int64_t total = static_cast<int64_t>(n_rows) * a.row_units * groups;
    const int64_t blocks = (total + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(pack_slices_kernel, dim3(static_cast<unsigned>(blocks < (1 << 20) ? blocks : (1 << 20))), dim3(kBlock),
                       0, s, a);
    return check_launch("pack_slices_kernel");
}

namespace {
struct GatherPiecesArgs {
    const float* src[4];
    const float* z[4];
    float* out[4];
    pygsd_piece_layout lay;
    int64_t ldz, ldo;
    int32_t n_rows, width4, shift;
};

// out[g][t, :] = (z[g] ? z[g][t, :] : 0) + row t of group g read through the piece layout: the merge of a returned product of the
// sharded propagate (receive buffer of the return exchange -> [n_rows, F] rows in local order), with the adjoint's last addend
// folded in.  One float4 per thread, row-major over the output: stores are coalesced, reads are whole 64-byte pieces.
__global__ __launch_bounds__(256) void gather_pieces_kernel(GatherPiecesArgs a)
{
    const int g = blockIdx.y;
    const float* __restrict__ src = a.src[g];
    const float* __restrict__ z = a.z[g];
    float* __restrict__ out = a.out[g];
    const int mask = (1 << a.shift) - 1;
    const int64_t total = static_cast<int64_t>(a.n_rows) * a.width4;
    for (int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; idx < total; idx += static_cast<int64_t>(gridDim.x) * 256) {
        const int t = static_cast<int>(idx / a.width4);
        const int c4 = static_cast<int>(idx - static_cast<int64_t>(t) * a.width4);
        const PieceRow pr = piece_row(a.lay, t);
        const int64_t off = piece_offset(pr, c4 >> 2, a.shift, mask) + (c4 & 3) * 4;
        float4 v = *reinterpret_cast<const float4*>(src + off);
        if (z) {
            const float4 zz = *reinterpret_cast<const float4*>(z + static_cast<int64_t>(t) * a.ldz + c4 * 4);
            v = make_float4(zz.x + v.x, zz.y + v.y, zz.z + v.z, zz.w + v.w);
        }
        *reinterpret_cast<float4*>(out + static_cast<int64_t>(t) * a.ldo + c4 * 4) = v;
    }
}
}  // namespace

extern "C" int pygsd_gather_pieces_f32(const float* const* srcs, const pygsd_piece_layout* layout, const float* const* zs, int64_t ldz,
                                       float* const* outs, int64_t ldo, int32_t n_groups, int32_t n_rows, int32_t width, void* stream)
{
    PYGSD_REQUIRE(n_groups >= 1 && n_groups <= 4 && n_rows >= 0 && width > 0 && width % 16 == 0,
                  "pygsd_gather_pieces_f32: 1..4 groups, a width that is a multiple of 16 floats");
    if (n_rows == 0) return 0;
    PYGSD_REQUIRE(srcs && layout && outs && ldo >= width && ldo % 4 == 0 && (!zs || (ldz >= width && ldz % 4 == 0)),
                  "pygsd_gather_pieces_f32: null pointer or row stride");
    GatherPiecesArgs a{};
    if (int rc = piece_layout_check(layout, n_rows, width, "pygsd_gather_pieces_f32", &a.shift)) return rc;
    for (int g = 0; g < n_groups; ++g) {
        PYGSD_REQUIRE(srcs[g] && outs[g] && aligned16(srcs[g]) && aligned16(outs[g]) && (!zs || !zs[g] || aligned16(zs[g])),
                      "pygsd_gather_pieces_f32: group %d null or not 16-byte aligned", g);
        a.src[g] = srcs[g];
        a.out[g] = outs[g];
        a.z[g] = zs ? zs[g] : nullptr;
    }
    a.lay = *layout;
    a.ldz = ldz; a.ldo = ldo; a.n_rows = n_rows; a.width4 = width / 4;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    const int64_t total = static_cast<int64_t>(n_rows) * a.width4;
    const int64_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(gather_pieces_kernel, dim3(static_cast<unsigned>(blocks < (1 << 20) ? blocks : (1 << 20)), n_groups), dim3(256), 0,
                       s, a);
    return check_launch("gather_pieces_kernel");
}

namespace {
struct WeightedSumArgs {
    const vec4f* x[8];
    float w[8];
    vec4f* out;
    int64_t n4, ldo4;    // float4 units: elements in total, output row stride
    int32_t k, c4;       // operands, float4 units per row
};

// out = sum_j w[j] * x[j], every operand read once (SIMPA / DIMPA: feat = sum_h w[h] * cur_h, SIMPA.py:77-93); the output
// may be a column block of a wider matrix (SIMPA's cat([feat_p, feat_n]) written in place)
__global__ __launch_bounds__(256) void weighted_sum_kernel(WeightedSumArgs a)
{
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n4;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        vec4f acc = a.x[0][i] * a.w[0];
        for (int j = 1; j < a.k; ++j) acc += a.x[j][i] * a.w[j];       // the reference's accumulation order
        const int64_t r = i / a.c4;
        a.out[r * a.ldo4 + (i - r * a.c4)] = acc;
    }
}

constexpr int kDotBlocks = 1024;     // at most; fewer where the workspace does not hold blocks x k float64 partials

struct DotsArgs {
    const vec4f* g;
    const vec4f* x[8];
    double* partial;     // [gridDim.x][k]
    int64_t n4, ldg4;
    int32_t k, c4;
};

// partial[b][j] = this block's share of <g, x_j>, j < k: g (possibly a column block of a wider matrix) is read ONCE for all
// k products.  Fixed grid and a fixed combination order: deterministic.  Products and sums are taken in FLOAT64 (round 6): an
// fp32 x fp32 product is exact in double, so the result is the correctly rounded dot product of the fp32 operands whatever
// the order -- these are reductions over all N x F elements whose value is often a small difference of partial sums in the
// hundreds, where an fp32 tree leaves 1e-5 absolute (tests/test_gpu_fuzz.py: SIMPA's hop-weight gradients, 20 x the
// reference's error on one draw).  Cost, measured at C3b (500 k x 64, k = 3): 93 us per launch against 88 with fp32 sums by
// rocprofv3 (107 against 88 by the in-process recorder) -- the 4 (k + 1) conversions per lane and step now pace the kernel
// (issuing two strides of loads per step changed nothing).
__global__ __launch_bounds__(256) void dots_kernel(DotsArgs a)
{
    __shared__ double sm[4][8];
    double acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < a.n4;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t r = i / a.c4;
        const vec4f g = a.g[r * a.ldg4 + (i - r * a.c4)];
        const double g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < a.k) {
                const vec4f x = a.x[j][i];
                acc[j] = fma(g0, static_cast<double>(x[0]), acc[j]);
                acc[j] = fma(g1, static_cast<double>(x[1]), acc[j]);
                acc[j] = fma(g2, static_cast<double>(x[2]), acc[j]);
                acc[j] = fma(g3, static_cast<double>(x[3]), acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double v = acc[j];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6][j] = v;
    }
    __syncthreads();
    if (threadIdx.x < static_cast<unsigned>(a.k))
        a.partial[static_cast<int64_t>(blockIdx.x) * a.k + threadIdx.x] =
            (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
}

// out[j] = sum_b partial[b][j], rounded to fp32 once: one block, 32 threads per product
__global__ __launch_bounds__(256) void dots_finish_kernel(const double* __restrict__ partial, int n_partials, int k,
                                                          float* __restrict__ out)
{
    __shared__ double sm[256];
    const int j = threadIdx.x >> 5, t = threadIdx.x & 31;
    double acc = 0.0;
    if (j < k)
        for (int b = t; b < n_partials; b += 32) acc += partial[static_cast<int64_t>(b) * k + j];
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 16; off > 0; off >>= 1) {
        if (t < off) sm[threadIdx.x] += sm[threadIdx.x + off];
        __syncthreads();
    }
    if (t == 0 && j < k) out[j] = static_cast<float>(sm[threadIdx.x]);
}
}  // namespace

extern "C" int pygsd_weighted_sum_f32(const float* const* xs, const float* weights, int32_t k, int64_t n_rows, int32_t n_cols,
                                      float* out, int64_t ldo, void* stream)
{
    PYGSD_REQUIRE(k >= 1 && k <= 8 && n_rows >= 0 && n_cols >= 0 && n_cols % 4 == 0,
                  "pygsd_weighted_sum_f32: 1..8 operands of rows that are multiples of 4 elements");
    if (n_rows == 0 || n_cols == 0) return 0;
    PYGSD_REQUIRE(xs && weights && out && aligned16(out) && ldo >= n_cols && ldo % 4 == 0,
                  "pygsd_weighted_sum_f32: null or unaligned pointer, or output row stride not a multiple of 16 bytes >= n_cols");
    WeightedSumArgs a{};
    for (int j = 0; j < k; ++j) {
        PYGSD_REQUIRE(xs[j] && aligned16(xs[j]), "pygsd_weighted_sum_f32: operand %d null or not 16-byte aligned", j);
        a.x[j] = reinterpret_cast<const vec4f*>(xs[j]);
        a.w[j] = weights[j];                                           // HOST array: the weights travel by value
    }
    a.out = reinterpret_cast<vec4f*>(out);
    a.c4 = n_cols / 4;
    a.n4 = n_rows * a.c4;
    a.ldo4 = ldo / 4;
    a.k = k;
    hipStream_t s = static_cast<hipStream_t>(stream);
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    const int64_t blocks = (a.n4 + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(static_cast<unsigned>(blocks < (1 << 22) ? blocks : (1 << 22))), dim3(kBlock), 0,
                       s, a);
    return check_launch("weighted_sum_kernel");
}

extern "C" int pygsd_dots_f32(const float* g, int64_t ldg, const float* const* xs, int32_t k, int64_t n_rows, int32_t n_cols,
                              float* out, void* workspace, size_t workspace_bytes, void* stream)
{
    PYGSD_REQUIRE(k >= 1 && k <= 8 && n_rows >= 0 && n_cols >= 0 && n_cols % 4 == 0,
                  "pygsd_dots_f32: 1..8 operands of rows that are multiples of 4 elements");
    PYGSD_REQUIRE(out && workspace && workspace_bytes >= 32768 && reinterpret_cast<uintptr_t>(workspace) % 8 == 0,
                  "pygsd_dots_f32: null output, or workspace smaller than 32 KiB or not 8-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_rows == 0 || n_cols == 0) {
        PYGSD_HIP_TRY(hipMemsetAsync(out, 0, sizeof(float) * k, s));
        return 0;
    }
    PYGSD_REQUIRE(g && xs && aligned16(g) && ldg >= n_cols && ldg % 4 == 0,
                  "pygsd_dots_f32: g null, not 16-byte aligned, or its row stride not a multiple of 16 bytes >= n_cols");
    DotsArgs a{};
    for (int j = 0; j < k; ++j) {
        PYGSD_REQUIRE(xs[j] && aligned16(xs[j]), "pygsd_dots_f32: operand %d null or not 16-byte aligned", j);
        a.x[j] = reinterpret_cast<const vec4f*>(xs[j]);
    }
    a.g = reinterpret_cast<const vec4f*>(g);
    a.partial = static_cast<double*>(workspace);
    a.c4 = n_cols / 4;
    a.n4 = n_rows * a.c4;
    a.ldg4 = ldg / 4;
    a.k = k;
    ProfScope prof(PYGSD_K_ELEMENTWISE, s);
    int64_t blocks = (a.n4 + kBlock - 1) / kBlock;
    // (1024 blocks keep ~64 KB of loads in flight per CU, what the stream needs; 64 KiB of workspace hold their partials for
    //  any k, the interface's minimum of 32 KiB for k <= 4 -- beyond that the grid shrinks to what the workspace holds)
    const int64_t fit = static_cast<int64_t>(workspace_bytes / (sizeof(double) * static_cast<size_t>(k)));
    if (blocks > kDotBlocks) blocks = kDotBlocks;
    if (blocks > fit) blocks = fit;
    hipLaunchKernelGGL(dots_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kBlock), 0, s, a);
    if (int rc = check_launch("dots_kernel")) return rc;
    hipLaunchKernelGGL(dots_finish_kernel, dim3(1), dim3(256), 0, s, static_cast<const double*>(workspace),
                       static_cast<int>(blocks), k, out);
    return check_launch("dots_finish_kernel");
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

// ---- content fingerprint (round 6: memo.py's check that a memoised operator's key tensors still hold what they held) --------
// 64 bits over the bytes of a buffer: the sum over the 4-byte words w_i of splitmix64((i << 32) | w_i) -- a sum, so the blocks
// may add in any order (integer atomics: deterministic), and position-dependent, so a permutation does not cancel.  16-byte
// loads where the buffer is 16-byte aligned (160 MB of edge_index: ~0.04 ms), 4-byte loads otherwise; a 1..3-byte tail is
// folded in by one thread.  Not cryptographic: a changed buffer collides with its old self with probability 2^-64.
namespace {
__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

template <bool VEC>
__global__ __launch_bounds__(256) void fingerprint_kernel(const unsigned char* data, long long n_words, int tail_bytes,
                                                          unsigned long long* out)
{
    const long long tid = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    unsigned long long h = 0;
    if constexpr (VEC) {
        const uint4* v = reinterpret_cast<const uint4*>(data);
        const long long n_vec = n_words >> 2;
        for (long long i = tid; i < n_vec; i += stride) {
            const uint4 q = v[i];
            const unsigned long long b = static_cast<unsigned long long>(4 * i) << 32;
            h += mix64(b | q.x) + mix64((b + (1ull << 32)) | q.y) + mix64((b + (2ull << 32)) | q.z) + mix64((b + (3ull << 32)) | q.w);
        }
        const uint32_t* w = reinterpret_cast<const uint32_t*>(data);
        for (long long i = (n_vec << 2) + tid; i < n_words; i += stride) h += mix64((static_cast<unsigned long long>(i) << 32) | w[i]);
    } else {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(data);
        for (long long i = tid; i < n_words; i += stride) h += mix64((static_cast<unsigned long long>(i) << 32) | w[i]);
    }
    if (tid == 0 && tail_bytes > 0) {
        uint32_t last = 0;
        for (int b = 0; b < tail_bytes; ++b) last |= static_cast<uint32_t>(data[4 * n_words + b]) << (8 * b);
        h += mix64((static_cast<unsigned long long>(n_words) << 32) | last) + static_cast<unsigned long long>(tail_bytes);
    }
    // one atomic per BLOCK (8192 wavefronts adding to one address took 0.4 ms by themselves -- measured on SIMPA's step)
    __shared__ unsigned long long part[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) h += __shfl_xor(h, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long total = (part[0] + part[1]) + (part[2] + part[3]);
        if (total != 0) atomicAdd(out, total);
    }
}
}  // namespace

extern "C" int pygsd_fingerprint_u64(const void* data, size_t bytes, uint64_t* out, void* stream)
{
    PYGSD_REQUIRE(out, "pygsd_fingerprint_u64: null output");
    hipStream_t s = static_cast<hipStream_t>(stream);
    PYGSD_HIP_TRY(hipMemsetAsync(out, 0, sizeof(uint64_t), s));
    if (bytes == 0) return 0;
    PYGSD_REQUIRE(data && (reinterpret_cast<uintptr_t>(data) & 3u) == 0, "pygsd_fingerprint_u64: null or not 4-byte aligned buffer");
    const long long n_words = static_cast<long long>(bytes / 4);
    const int tail = static_cast<int>(bytes % 4);
    long long blocks = (n_words / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 512 ? 512 : blocks);       // two blocks per CU keep ~3 TB/s of 16-byte loads in flight
    unsigned long long* o = reinterpret_cast<unsigned long long*>(out);
    const unsigned char* d = static_cast<const unsigned char*>(data);
    if (aligned16(data))
        hipLaunchKernelGGL(fingerprint_kernel<true>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, d, n_words, tail, o);
    else
        hipLaunchKernelGGL(fingerprint_kernel<false>, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, s, d, n_words, tail, o);
    return check_launch("fingerprint_kernel");
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
