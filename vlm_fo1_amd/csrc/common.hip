// common.hip — error state + ABI version for libfo1hip.so
#include "common.h"

#include <string.h>

#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace fo1 {

static thread_local char g_err[512] = {0};

char* err_buf() { return g_err; }

int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

// ---- per-kernel timing ------------------------------------------------------------
struct ProfRec {
    std::string name;
    hipEvent_t e0, e1;
    double work;
};
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static size_t g_prof_mark = 0;   // records [g_prof_mark, size) have no stage tag yet
static std::vector<std::pair<hipEvent_t, hipEvent_t>> g_event_pool;
static std::mutex g_prof_mu;

bool profile_enabled() { return g_prof_on; }

void profile_events(const char* name, double work, hipEvent_t* e0, hipEvent_t* e1) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.name = name;
    r.work = work;
    if (!g_event_pool.empty()) {
        r.e0 = g_event_pool.back().first;
        r.e1 = g_event_pool.back().second;
        g_event_pool.pop_back();
    } else {
        (void)hipEventCreate(&r.e0);
        (void)hipEventCreate(&r.e1);
    }
    *e0 = r.e0;
    *e1 = r.e1;
    g_prof.push_back(r);
}

}  // namespace fo1

extern "C" {

int fo1_profile_enable(int on) {
    fo1::g_prof_on = on != 0;
    return FO1_OK;
}

// Resolves all pending records (synchronises their events), aggregates by kernel name and
// writes up to `cap` rows; returns the number of distinct kernels, or <0 on error.
int fo1_profile_read(fo1_profile_row_t* rows, int cap, int reset) {
    using namespace fo1;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::vector<fo1_profile_row_t> agg;
    for (auto& r : g_prof) {
        if (hipEventSynchronize(r.e1) != hipSuccess) return set_err(FO1_ERR_ARG, "profile: event sync failed");
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        size_t i = 0;
        for (; i < agg.size(); ++i)
            if (r.name == agg[i].name) break;
        if (i == agg.size()) {
            fo1_profile_row_t row;
            memset(&row, 0, sizeof row);
            snprintf(row.name, sizeof row.name, "%s", r.name.c_str());
            agg.push_back(row);
        }
        agg[i].calls += 1;
        agg[i].total_ms += ms;
        agg[i].total_work += r.work;
    }
    for (int i = 0; i < (int)agg.size() && i < cap; ++i) rows[i] = agg[i];
    if (reset) {
        for (auto& r : g_prof) g_event_pool.emplace_back(r.e0, r.e1);
        g_prof.clear();
        g_prof_mark = 0;
    }
    return (int)agg.size();
}

// Tags every record since the previous call (or reset) with "<tag>|"; no synchronisation.
int fo1_profile_stage(const char* tag) {
    using namespace fo1;
    if (!tag || strlen(tag) > 15) return set_err(FO1_ERR_ARG, "profile: stage tag must be 1..15 characters");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (size_t i = g_prof_mark; i < g_prof.size(); ++i) g_prof[i].name = std::string(tag) + "|" + g_prof[i].name;
    g_prof_mark = g_prof.size();
    return FO1_OK;
}

int fo1_abi_version(void) { return FO1_ABI_VERSION; }
const char* fo1_last_error(void) { return fo1::err_buf(); }
}
