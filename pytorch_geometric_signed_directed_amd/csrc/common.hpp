// Shared host-side plumbing of libpygsd_hip.so: error string, launch checking, kernel-timing recorder.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pygsd_hip.h"

namespace pygsd {

std::string& last_error();
int fail(const char* fmt, ...);

#define PYGSD_HIP_TRY(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) return ::pygsd::fail("%s: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

#define PYGSD_REQUIRE(cond, ...)                         \
    do {                                                 \
        if (!(cond)) return ::pygsd::fail(__VA_ARGS__);  \
    } while (0)

// Brackets the launches issued inside its lifetime with two hipEvents on `stream` when the
// recorder is enabled (bench.py's roofline measurement); otherwise free.
class ProfScope {
public:
    ProfScope(int kernel_id, hipStream_t stream);
    ~ProfScope();

private:
    int id_;
    hipStream_t stream_;
    hipEvent_t start_ = nullptr;
};

inline int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- piece layouts (include/pygsd_hip.h: pygsd_piece_layout) ------------------------------------------------------------
// Where the 16-float pieces of row t live: `base` = element offset of the row inside slot (blk * slots_per_blk) of its chunk,
// `slot_stride` = elements between consecutive slots of that chunk.  Piece q of the row (columns [16 q, 16 q + 16)) sits at
//     base + (q >> shift) * slot_stride + (q & mask) * 16        shift = log2(slot_floats / 16), mask = (1 << shift) - 1
// and replica k of a stored row a further k * slots_per_blk * slot_stride on.
struct PieceRow {
    int64_t base;
    int32_t slot_stride;
};

__device__ __forceinline__ PieceRow piece_row(const pygsd_piece_layout& L, int t)
{
    int blk = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) blk += (static_cast<int64_t>(t) >= static_cast<int64_t>(k) * L.blk_rows) ? 1 : 0;   // <= 8 blocks
    const int u = t - blk * L.blk_rows;
    const int r = ((L.n_chunks > 1 && u >= L.lo[1]) ? 1 : 0) + ((L.n_chunks > 2 && u >= L.lo[2]) ? 1 : 0) +
                  ((L.n_chunks > 3 && u >= L.lo[3]) ? 1 : 0);
    const int lo_r = r == 0 ? 0 : (r == 1 ? L.lo[1] : (r == 2 ? L.lo[2] : L.lo[3]));
    const int rows_r = r == 0 ? L.rows[0] : (r == 1 ? L.rows[1] : (r == 2 ? L.rows[2] : L.rows[3]));
    const int64_t base_r = r == 0 ? L.base[0] : (r == 1 ? L.base[1] : (r == 2 ? L.base[2] : L.base[3]));
    PieceRow o;
    o.slot_stride = rows_r * L.row_stride;
    o.base = base_r + static_cast<int64_t>(blk) * L.slots_per_blk * o.slot_stride + static_cast<int64_t>(u - lo_r) * L.row_stride;
    return o;
}

__device__ __forceinline__ int64_t piece_offset(const PieceRow& pr, int q, int shift, int mask)
{
    return pr.base + static_cast<int64_t>(q >> shift) * pr.slot_stride + (q & mask) * 16;
}

// host side: argument checks shared by the entry points that take a layout; *shift = log2(slot_floats / 16)
inline int piece_layout_check(const pygsd_piece_layout* L, int32_t n_rows, int32_t width, const char* who, int* shift)
{
    PYGSD_REQUIRE(L->n_chunks >= 1 && L->n_chunks <= 4, "%s: a piece layout holds 1..4 chunks (got %d)", who, L->n_chunks);
    PYGSD_REQUIRE(L->blk_rows >= 1 && L->blk_rows <= (1 << 27) && (static_cast<int64_t>(n_rows) + L->blk_rows - 1) / L->blk_rows <= 8,
                  "%s: %d rows in blocks of %d: at most 8 blocks of at most 2^27 rows", who, n_rows, L->blk_rows);
    PYGSD_REQUIRE(L->slots_per_blk >= 1 && L->replicas >= 1 && L->replicas <= 8, "%s: bad slot / replica count", who);
    int sh = 0;
    while ((16 << sh) < L->slot_floats) ++sh;
    PYGSD_REQUIRE(L->slot_floats >= 16 && (16 << sh) == L->slot_floats && width % L->slot_floats == 0,
                  "%s: slot width %d must be 16 * 2^k floats and divide the row width %d", who, L->slot_floats, width);
    PYGSD_REQUIRE(L->row_stride >= L->slot_floats && L->row_stride % 4 == 0, "%s: row stride %d", who, L->row_stride);
    PYGSD_REQUIRE(L->lo[0] == 0 && L->lo[L->n_chunks] == L->blk_rows, "%s: chunk bounds must span [0, blk_rows)", who);
    for (int r = 0; r < L->n_chunks; ++r) {
        PYGSD_REQUIRE(L->lo[r] <= L->lo[r + 1] && L->rows[r] >= L->lo[r + 1] - L->lo[r] && L->base[r] % 4 == 0 &&
                      static_cast<int64_t>(L->rows[r]) * L->row_stride < (int64_t(1) << 31),
                      "%s: chunk %d: bounds, slot rows, base alignment or slot size out of range", who, r);
    }
    *shift = sh;
    return 0;
}

inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// 0 = fp32 tall products / weight gradients in the split form (three bf16 pieces per value, bf16 matrix pipe) wherever a shape has
// one, 1 = exact fp32 MFMA everywhere (csrc/tall.hip; pygsd_tall_f32_form, PYGSD_TALL_F32=exact)
std::atomic<int>& tall_f32_form();

}  // namespace pygsd
