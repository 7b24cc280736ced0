// Opt-in per-kernel timing with HIP events recorded on the caller's stream (used by bench.py to measure the
// dominant kernel's launch duration live, over the timed region, on the very stream the kernels run on).
// This is the ONE piece of process-global state in the library; it is off unless ls2fm_profile_enable(1) is called
// and is meant for single-threaded benchmarking.
#include <mutex>
#include <vector>

#include "render_common.h"

namespace {

const char* const kNames[LS2FM_PROF_COUNT] = {
    "prep_weights", "ray_encode_sdf", "ray_encode_rad", "shade_fwd", "shade_bwd", "wgrad", "wgrad_reduce",
    "slab_scatter_sdf", "slab_scatter_rad", "finalize", "sdf_eval", "sphere_trace"};

struct Mark { int id; hipEvent_t ev; };

std::mutex g_mu;
bool g_enabled = false;
std::vector<Mark> g_marks;
std::vector<hipEvent_t> g_pool;
double g_total_ms[LS2FM_PROF_COUNT] = {};
int64_t g_launches[LS2FM_PROF_COUNT] = {};

void resolve_locked() {
    for (size_t i = 0; i + 1 < g_marks.size(); ++i) {
        const int id = g_marks[i].id;
        if (id < 0 || id >= LS2FM_PROF_COUNT) continue;
        if (hipEventSynchronize(g_marks[i + 1].ev) != hipSuccess) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_marks[i].ev, g_marks[i + 1].ev) == hipSuccess) {
            g_total_ms[id] += ms;
            g_launches[id] += 1;
        }
    }
    for (const Mark& m : g_marks) g_pool.push_back(m.ev);
    g_marks.clear();
}

}  // namespace

// id >= 0: a kernel of that id is about to be enqueued on `stream`; id < 0: end of the current call
void ls2fm_prof_mark(int id, hipStream_t stream) {
    if (!g_enabled) return;
    std::lock_guard<std::mutex> lock(g_mu);
    hipEvent_t ev;
    if (!g_pool.empty()) { ev = g_pool.back(); g_pool.pop_back(); }
    else if (hipEventCreate(&ev) != hipSuccess) return;
    (void)hipEventRecord(ev, stream);
    g_marks.push_back({id, ev});
    if (g_marks.size() > 1u << 16) resolve_locked();          // bound the backlog (synchronises)
}

extern "C" int ls2fm_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_enabled = on != 0;
    return LS2FM_OK;
}

extern "C" int ls2fm_profile_reset(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    resolve_locked();
    for (int i = 0; i < LS2FM_PROF_COUNT; ++i) { g_total_ms[i] = 0.0; g_launches[i] = 0; }
    return LS2FM_OK;
}

extern "C" int ls2fm_profile_count(void) { return LS2FM_PROF_COUNT; }

extern "C" const char* ls2fm_profile_name(int i) { return (i >= 0 && i < LS2FM_PROF_COUNT) ? kNames[i] : ""; }

extern "C" int ls2fm_profile_get(int i, double* total_ms, int64_t* launches) {
    if (i < 0 || i >= LS2FM_PROF_COUNT || !total_ms || !launches) return LS2FM_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> lock(g_mu);
    resolve_locked();
    *total_ms = g_total_ms[i];
    *launches = g_launches[i];
    return LS2FM_OK;
}
