// Error string + kernel-timing recorder of libpygsd_hip.so (include/pygsd_hip.h: pygsd_version,
// pygsd_last_error, pygsd_prof_*).
#include "common.hpp"

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
