// Internal fork/join helper: a non-blocking side stream + events PER CALLER STREAM (per device), so that latency-bound
// small kernels (the scatter's scans, finalize) and the MFMA weight-gradient chain overlap with the wide kernels of the
// same call.  Everything forked is joined back into the caller's stream before the C-ABI call returns -- on error paths
// too (ls2fm_join_on_error) -- so the caller-visible contract ("asynchronous on `stream`") is unchanged and an active stream
// capture is never left with an unjoined branch.  Calls on DIFFERENT caller streams (two host threads, autograd on two
// streams) get different side streams and events, so they cannot order against each other's forks; concurrent calls on the
// SAME stream from two threads are as undefined as any other concurrent use of one stream.
// LS2FM_SERIAL=1 disables the fork, and so does the opt-in per-kernel profiler (overlapped kernels would time each other).
#include <atomic>
#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "render_common.h"

namespace {
struct StreamCtx { hipStream_t side = nullptr; hipEvent_t fork = nullptr, mid = nullptr, join = nullptr; };
std::mutex g_mu;
std::map<std::pair<int, hipStream_t>, StreamCtx> g_ctx;
constexpr size_t kMaxContexts = 64;          // streams come and go: beyond this, calls run unforked instead of leaking
}  // namespace

bool ls2fm_side_stream(SideCtx* out, hipStream_t caller) {
    static const bool serial = [] { const char* e = getenv("LS2FM_SERIAL"); return e && e[0] == '1'; }();
    if (serial || ls2fm_prof_enabled()) return false;      // per-kernel profiling: serial launches, so that the
                                                           // event-bracketed durations are those of the kernel alone
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_mu);
    // depth of the fork tree: main -> side -> side-of-side and no further (a call running on a side-of-side stream launches
    // serially) -- deeper trees inside a stream capture take hipStreamEndCapture down on ROCm 7.2
    static const int max_depth = [] { const char* e = getenv("LS2FM_FORK_DEPTH"); return e ? atoi(e) : 2; }();
    int depth = 0;
    for (hipStream_t c = caller; depth <= max_depth;) {
        bool is_side = false;
        for (const auto& kv : g_ctx)
            if (kv.first.first == dev && kv.second.side == c) { c = kv.first.second; is_side = true; break; }
        if (!is_side) break;
        ++depth;
    }
    if (depth >= max_depth) return false;
    auto it = g_ctx.find({dev, caller});
    if (it == g_ctx.end()) {
        if (g_ctx.size() >= kMaxContexts) return false;
        StreamCtx c;
        // LS2FM_SIDE_PRIORITY (experiment): -1 = the side chains on a high-priority queue, 1 = low
        static const int prio = [] { const char* e = getenv("LS2FM_SIDE_PRIORITY"); return e ? atoi(e) : 0; }();
        if (prio != 0) {
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);          // lo = least, hi = greatest priority (numerically smaller)
            if (hipStreamCreateWithPriority(&c.side, hipStreamNonBlocking, prio < 0 ? hi : lo) != hipSuccess) return false;
        } else if (hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&c.fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c.join, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c.mid, hipEventDisableTiming) != hipSuccess)
            return false;
        it = g_ctx.emplace(std::make_pair(dev, caller), c).first;
    }
    out->side = it->second.side; out->fork = it->second.fork; out->mid = it->second.mid; out->join = it->second.join;
    return true;
}

namespace {
std::atomic<int*> g_err_word{nullptr};
}
int* ls2fm_async_error_word(hipStream_t stream) {
    int* w = g_err_word.load(std::memory_order_acquire);
    if (w) return w;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;   // (no allocation inside a capture)
    std::lock_guard<std::mutex> lock(g_mu);
    w = g_err_word.load(std::memory_order_acquire);
    if (w) return w;
    void* p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    *static_cast<volatile int*>(p) = 0;
    g_err_word.store(static_cast<int*>(p), std::memory_order_release);
    return static_cast<int*>(p);
}
extern "C" int ls2fm_async_error(int clear) {
    int* w = g_err_word.load(std::memory_order_acquire);
    if (!w) return 0;
    const int v = *static_cast<volatile int*>(w);
    if (clear && v) *static_cast<volatile int*>(w) = 0;
    return v;
}
int ls2fm_device_cus() {
    static const int n_cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    return n_cus;
}

// an error after a fork: whatever was enqueued on the side stream is still joined into the caller's stream
int ls2fm_join_on_error(bool forked, const SideCtx& sc, hipStream_t caller, int status) {
    if (forked) {
        (void)hipEventRecord(sc.join, sc.side);
        (void)hipStreamWaitEvent(caller, sc.join, 0);
    }
    return status;
}
