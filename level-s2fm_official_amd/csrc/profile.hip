// Opt-in per-kernel timing with HIP events recorded on the caller's stream (used by bench.py to measure the
// dominant kernel's launch duration live, over the timed region, on the very stream the kernels run on).
// This is the ONE piece of process-global state in the library; it is off unless ls2fm_profile_enable(1) is called
// and is meant for single-threaded benchmarking.
#include <mutex>
#include <vector>

#include "render_common.h"

namespace {

const char* const kNames[LS2FM_PROF_COUNT] = {
    "prep_weights", "ray_encode_sdf", "ray_encode_rad", "shade_fwd", "shade_bwd", "wgrad_mlp_geo", "wgrad_dec",
    "slab_accumulate", "scatter_fill", "reduce_finalize", "sdf_eval", "sphere_trace", "post_shade", "loss_head_fwd",
    "loss_head_bwd", "wgrad_mlp_sdf", "pose_grad", "ray_encode_pair"};

struct Span { int id; hipEvent_t a, b; };

std::mutex g_mu;
bool g_enabled = false;
std::vector<Span> g_spans;
std::vector<hipEvent_t> g_pool;
double g_total_ms[LS2FM_PROF_COUNT] = {};
int64_t g_launches[LS2FM_PROF_COUNT] = {};

hipEvent_t take_event() {
    hipEvent_t ev = nullptr;
    if (!g_pool.empty()) { ev = g_pool.back(); g_pool.pop_back(); }
    else if (hipEventCreate(&ev) != hipSuccess) ev = nullptr;
    return ev;
}

void resolve_locked() {
    for (const Span& sp : g_spans) {
        if (sp.a && sp.b && sp.id >= 0 && sp.id < LS2FM_PROF_COUNT && hipEventSynchronize(sp.b) == hipSuccess) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) { g_total_ms[sp.id] += ms; g_launches[sp.id] += 1; }
        }
        if (sp.a) g_pool.push_back(sp.a);
        if (sp.b) g_pool.push_back(sp.b);
    }
    g_spans.clear();
}

}  // namespace

void ls2fm_prof_begin(int id, hipStream_t stream) {
    if (!g_enabled) return;
    std::lock_guard<std::mutex> lock(g_mu);
    Span sp{id, take_event(), nullptr};
    if (sp.a) (void)hipEventRecord(sp.a, stream);
    g_spans.push_back(sp);
}

void ls2fm_prof_end(int id, hipStream_t stream) {
    if (!g_enabled) return;
    std::lock_guard<std::mutex> lock(g_mu);
    for (size_t k = g_spans.size(); k-- > 0;)
        if (g_spans[k].id == id && !g_spans[k].b) {
            g_spans[k].b = take_event();
            if (g_spans[k].b) (void)hipEventRecord(g_spans[k].b, stream);
            break;
        }
    if (g_spans.size() > (1u << 15)) resolve_locked();          // bound the backlog (synchronises)
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

bool ls2fm_prof_enabled() { return g_enabled; }

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
