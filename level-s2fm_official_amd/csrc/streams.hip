// Internal fork/join helper: a cached non-blocking side stream + events per device, so that latency-bound
// single-workgroup kernels (prep_weights, wgrad_reduce, finalize) and the MFMA weight-gradient GEMM overlap with the
// wide kernels of the same call.  Everything forked is joined back into the caller's stream before the C-ABI call
// returns, so the caller-visible contract ("asynchronous on `stream`") is unchanged.  LS2FM_SERIAL=1 disables it, and so does the opt-in per-kernel
// profiler (overlapped kernels would time each other).
#include <cstdlib>
#include <mutex>

#include "render_common.h"

namespace {
struct DeviceCtx { bool init = false; hipStream_t side = nullptr, fast = nullptr; hipEvent_t fork = nullptr, mid = nullptr, join = nullptr; };
std::mutex g_mu;
DeviceCtx g_ctx[64];
}  // namespace

bool ls2fm_side_stream(SideCtx* out) {
    static const bool serial = [] { const char* e = getenv("LS2FM_SERIAL"); return e && e[0] == '1'; }();
    if (serial || ls2fm_prof_enabled()) return false;      // per-kernel profiling: serial launches, so that the
                                                           // event-bracketed durations are those of the kernel alone
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    std::lock_guard<std::mutex> lock(g_mu);
    DeviceCtx& c = g_ctx[dev];
    if (!c.init) {
        if (hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) != hipSuccess) return false;
        {   // small latency-critical chains (the scatter's count / scan) go to a stream of the highest priority, so that their
            // few workgroups are dispatched ahead of the wide kernel they run beside
            int lo = 0, hi = 0;
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; hi = 0; }
            if (hipStreamCreateWithPriority(&c.fast, hipStreamNonBlocking, hi) != hipSuccess) return false;
        }
        if (hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&c.join, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&c.mid, hipEventDisableTiming) != hipSuccess) return false;
        c.init = true;
    }
    out->side = c.side; out->fast = c.fast; out->fork = c.fork; out->mid = c.mid; out->join = c.join;
    return true;
}
