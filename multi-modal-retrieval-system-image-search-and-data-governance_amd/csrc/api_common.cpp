// Error text, version and the optional in-library launch profiler for the C ABI (include/mmr.h).
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <vector>

#include "mmr_common.h"

namespace mmr {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ---- launch profiler: HIP events recorded on the launch stream around each kernel class.
// Off by default (one relaxed load per launch).  bench.py turns it on for the timed region so the
// roofline figures come from the very launches that were timed, on the stream they ran on.
struct ProfPair { hipEvent_t a, b; int cls; };
static bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfPair> g_pool;     // preallocated event pairs
static size_t g_used = 0;
static long long g_dropped = 0;

ProfScope::ProfScope(int cls, hipStream_t st) : slot_(-1), st_(st)
{
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_used >= g_pool.size()) { ++g_dropped; return; }
    slot_ = (long)g_used++;
    g_pool[slot_].cls = cls;
    (void)hipEventRecord(g_pool[slot_].a, st_);
}
ProfScope::~ProfScope()
{
    if (slot_ >= 0) (void)hipEventRecord(g_pool[slot_].b, st_);
}
}  // namespace mmr

using namespace mmr;

extern "C" const char *mmr_last_error(void) { return mmr::g_err; }
extern "C" int mmr_version(void) { return 1; }

extern "C" int mmr_prof_enable(int on, int max_launches)
{
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (on) {
        g_used = 0;      // a new recording window; turning it off keeps the window readable
        g_dropped = 0;
        MMR_CHECK_ARG(max_launches > 0 && max_launches <= (1 << 20), "mmr_prof_enable: max_launches %d outside (0, 2^20]", max_launches);
        while ((int)g_pool.size() < max_launches) {
            ProfPair p{};
            MMR_CHECK_HIP(hipEventCreate(&p.a));
            MMR_CHECK_HIP(hipEventCreate(&p.b));
            g_pool.push_back(p);
        }
    }
    g_prof_on = on != 0;
    return MMR_OK;
}

extern "C" int mmr_prof_read(int cls, double *total_ms, long long *launches, long long *dropped)
{
    MMR_CHECK_ARG(cls >= 0 && cls < MMR_PROF_CLASSES && total_ms && launches, "mmr_prof_read: bad argument");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double sum = 0.0;
    long long n = 0;
    for (size_t i = 0; i < g_used; ++i) {
        if (g_pool[i].cls != cls) continue;
        MMR_CHECK_HIP(hipEventSynchronize(g_pool[i].b));
        float ms = 0.f;
        MMR_CHECK_HIP(hipEventElapsedTime(&ms, g_pool[i].a, g_pool[i].b));
        sum += ms;
        ++n;
    }
    *total_ms = sum;
    *launches = n;
    if (dropped) *dropped = g_dropped;
    return MMR_OK;
}
