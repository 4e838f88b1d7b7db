// Shared host-side plumbing of libpygsd_hip.so: error string, launch checking, kernel-timing recorder.
#pragma once
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

inline size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace pygsd
