// Fused Adam step over a list of fp32 tensors (the stage loops update two 12 M-entry hash
// tables and 26 small tensors with torch.optim.Adam: Initialization.py:149-179, BA.py:117-182).  One launch for the whole
// list, one pass over memory: 16 B read + 12 B written per element (p, g, m, v -> p, m, v), HBM-bound.
// Arithmetic follows torch.optim.Adam (amsgrad = False, maximize = False), operation by operation:
//   g' = g + wd p ; m = lerp(m, g', 1 - b1) ; v = v b2 + (1 - b2) g' g' ;
//   p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "ls2fm_device.h"

namespace {

constexpr int kAdamMaxTensors = 32;
constexpr int kAdamThreads = 256;
constexpr int kAdamPerThread = 8;          // two float4 per thread
constexpr int kAdamTile = kAdamThreads * kAdamPerThread;

struct AdamJobs {
    float* p[kAdamMaxTensors];
    const float* g[kAdamMaxTensors];
    float* m[kAdamMaxTensors];
    float* v[kAdamMaxTensors];
    int64_t n[kAdamMaxTensors];
    // optional second destination of the updated values: an ENTRY-INTERLEAVED copy of two hash tables ([entry][t0 f0 f1 | t1 f0 f1],
    // what the render's gather pass reads: one 16-byte gather per corner serves both grids).  mirror[j] = the copy's base + 2 *
    // slot (this tensor's half of every 16-byte entry) or null; element k of the tensor goes to mirror[(k >> 1) * 4 + (k & 1)]
    float* mirror[kAdamMaxTensors];
    // per tensor: the learning-rate dependent scalars (parameter groups with different rates share a launch) -- from the host
    // (step_size, bc2_sqrt) or, when sched[j] is set, from that device-resident schedule
    float step_size[kAdamMaxTensors], bc2_sqrt[kAdamMaxTensors];
    const void* sched[kAdamMaxTensors];
    int first_block[kAdamMaxTensors + 1];
    int count;
};

struct AdamHyper { float step_size, bc2_sqrt, w1, beta2, one_minus_beta2, eps, weight_decay; };

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamHyper& h) {
    if (h.weight_decay != 0.f) g = fmaf(h.weight_decay, p, g);
    m = h.w1 < 0.5f ? m + h.w1 * (g - m) : g - (g - m) * (1.0f - h.w1);       // torch lerp
    v = v * h.beta2 + h.one_minus_beta2 * (g * g);
    const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
    p = p - h.step_size * (m / denom);
}

// Device-resident schedule of a capturable optimizer: a hipGraph replay cannot take a new learning rate or step count from the
// host, so both live in memory and a one-thread kernel in front of the update advances them
//   step += 1 ; step_size = lr / (1 - b1^step) ; bc2_sqrt = sqrt(1 - b2^step) ; lr *= gamma   (torch's ExponentialLR: one
//   multiplication per scheduler step, in double, as Python evaluates it)
struct AdamSched { double step, lr, gamma; float step_size, bc2_sqrt; };

struct AdamSchedList { AdamSched* st[16]; int count; };

// one thread per distinct schedule of the call (an optimizer's parameter groups: one launch for all of them)
__global__ void adam_advance_kernel(AdamSchedList list, double beta1, double beta2) {
    if ((int)threadIdx.x >= list.count) return;
    AdamSched* st = list.st[threadIdx.x];
    st->step += 1.0;
    const double bc1 = 1.0 - pow(beta1, st->step), bc2 = 1.0 - pow(beta2, st->step);
    st->step_size = (float)(st->lr / bc1);
    st->bc2_sqrt = (float)sqrt(bc2);
    st->lr *= st->gamma;
}

// groups without a gradient this step: the rate decays, the Adam step count stays
__global__ void adam_decay_kernel(AdamSchedList list) {
    if ((int)threadIdx.x < list.count) list.st[threadIdx.x]->lr *= list.st[threadIdx.x]->gamma;
}

__global__ void __launch_bounds__(kAdamThreads)
adam_kernel(AdamJobs jobs, AdamHyper h) {
    int j = 0;
    while (j + 1 < jobs.count && (int)blockIdx.x >= jobs.first_block[j + 1]) ++j;
    {
        const AdamSched* sched = reinterpret_cast<const AdamSched*>(jobs.sched[j]);
        h.step_size = sched ? sched->step_size : jobs.step_size[j];
        h.bc2_sqrt = sched ? sched->bc2_sqrt : jobs.bc2_sqrt[j];
    }
    const int64_t base = (int64_t)(blockIdx.x - jobs.first_block[j]) * kAdamTile;
    const int64_t n = jobs.n[j];
    float* __restrict__ p = jobs.p[j];
    const float* __restrict__ g = jobs.g[j];
    float* __restrict__ m = jobs.m[j];
    float* __restrict__ v = jobs.v[j];
    float* __restrict__ mir = jobs.mirror[j];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
#pragma unroll
    for (int q = 0; q < kAdamPerThread / 4; ++q) {
        const int64_t i = base + ((int64_t)q * kAdamThreads + threadIdx.x) * 4;
        if (i >= n) continue;
        if (vec && i + 4 <= n) {
            float4 pp = *reinterpret_cast<float4*>(p + i), mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            adam_one(pp.x, gg.x, mm.x, vv.x, h); adam_one(pp.y, gg.y, mm.y, vv.y, h);
            adam_one(pp.z, gg.z, mm.z, vv.z, h); adam_one(pp.w, gg.w, mm.w, vv.w, h);
            *reinterpret_cast<float4*>(p + i) = pp; *reinterpret_cast<float4*>(m + i) = mm; *reinterpret_cast<float4*>(v + i) = vv;
            if (mir) {                          // i is a multiple of 4: two whole entries
                *reinterpret_cast<float2*>(mir + (i >> 1) * 4) = make_float2(pp.x, pp.y);
                *reinterpret_cast<float2*>(mir + (i >> 1) * 4 + 4) = make_float2(pp.z, pp.w);
            }
        } else {
            for (int64_t k = i; k < n && k < i + 4; ++k) {
                float pp = p[k], mm = m[k], vv = v[k];
                adam_one(pp, g[k], mm, vv, h);
                p[k] = pp; m[k] = mm; v[k] = vv;
                if (mir) mir[(k >> 1) * 4 + (k & 1)] = pp;
            }
        }
    }
}

// The two hash tables of a dual-field pair in ONE job: both updates plus the entry-interleaved copy the render's gather pass
// reads ([entry][a f0 f1 | b f0 f1]).  A thread owns two entries of each table (16 bytes each way) and writes the copy as two
// whole 16-byte entries: full cache lines, where two separate jobs would each write 8 bytes of every 16.
struct AdamPair {
    float *pa, *ma, *va, *pb, *mb, *vb, *mirror;
    const float *ga, *gb;
    int64_t n;                                       // floats per table (even)
    float step_a, bc2_a, step_b, bc2_b;
    const void *sched_a, *sched_b;
};

__global__ void __launch_bounds__(kAdamThreads)
adam_pair_kernel(AdamPair q, AdamHyper h) {
    AdamHyper ha = h, hb = h;
    {
        const AdamSched* sa = reinterpret_cast<const AdamSched*>(q.sched_a);
        const AdamSched* sb = reinterpret_cast<const AdamSched*>(q.sched_b);
        ha.step_size = sa ? sa->step_size : q.step_a; ha.bc2_sqrt = sa ? sa->bc2_sqrt : q.bc2_a;
        hb.step_size = sb ? sb->step_size : q.step_b; hb.bc2_sqrt = sb ? sb->bc2_sqrt : q.bc2_b;
    }
    const int64_t i = ((int64_t)blockIdx.x * kAdamThreads + threadIdx.x) * 4;
    if (i >= q.n) return;
    if (i + 4 <= q.n) {
        float4 pa = *reinterpret_cast<float4*>(q.pa + i), ma = *reinterpret_cast<float4*>(q.ma + i), va = *reinterpret_cast<float4*>(q.va + i);
        float4 pb = *reinterpret_cast<float4*>(q.pb + i), mb = *reinterpret_cast<float4*>(q.mb + i), vb = *reinterpret_cast<float4*>(q.vb + i);
        const float4 ga = *reinterpret_cast<const float4*>(q.ga + i), gb = *reinterpret_cast<const float4*>(q.gb + i);
        adam_one(pa.x, ga.x, ma.x, va.x, ha); adam_one(pa.y, ga.y, ma.y, va.y, ha);
        adam_one(pa.z, ga.z, ma.z, va.z, ha); adam_one(pa.w, ga.w, ma.w, va.w, ha);
        adam_one(pb.x, gb.x, mb.x, vb.x, hb); adam_one(pb.y, gb.y, mb.y, vb.y, hb);
        adam_one(pb.z, gb.z, mb.z, vb.z, hb); adam_one(pb.w, gb.w, mb.w, vb.w, hb);
        *reinterpret_cast<float4*>(q.pa + i) = pa; *reinterpret_cast<float4*>(q.ma + i) = ma; *reinterpret_cast<float4*>(q.va + i) = va;
        *reinterpret_cast<float4*>(q.pb + i) = pb; *reinterpret_cast<float4*>(q.mb + i) = mb; *reinterpret_cast<float4*>(q.vb + i) = vb;
        float4* dst = reinterpret_cast<float4*>(q.mirror + 2 * i);       // entries i / 2 and i / 2 + 1
        dst[0] = make_float4(pa.x, pa.y, pb.x, pb.y);
        dst[1] = make_float4(pa.z, pa.w, pb.z, pb.w);
    } else {
        for (int64_t k = i; k < q.n; ++k) {
            float pa = q.pa[k], ma = q.ma[k], va = q.va[k], pb = q.pb[k], mb = q.mb[k], vb = q.vb[k];
            adam_one(pa, q.ga[k], ma, va, ha);
            adam_one(pb, q.gb[k], mb, vb, hb);
            q.pa[k] = pa; q.ma[k] = ma; q.va[k] = va; q.pb[k] = pb; q.mb[k] = mb; q.vb[k] = vb;
            q.mirror[(k >> 1) * 4 + (k & 1)] = pa;
            q.mirror[(k >> 1) * 4 + 2 + (k & 1)] = pb;
        }
    }
}

}  // namespace

// The general form behind every entry point: per-tensor learning rates (host scalars lrs[t], or device-resident schedules
// sched_states[t] -- each distinct schedule is advanced once), optional mirrors.  Two tensors whose mirrors are the two halves
// of one interleaved copy (base and base + 2, same length) are updated by ONE paired job.
extern "C" int ls2fm_adam_step_multi(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                                     float* const* exp_avg_sq, const int64_t* numel, float* const* mirrors, const float* lrs,
                                     void* const* sched_states, float beta1, float beta2, float eps, float weight_decay,
                                     int64_t step, void* stream_) {
    LS2FM_CHECK_ARG(n_tensors >= 0 && (n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq && numel && (lrs || sched_states))));
    LS2FM_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f);
    static_assert(sizeof(AdamSched) == 32, "sched_state is 32 bytes {double step, lr, gamma; float[2]}");
    hipStream_t stream = (hipStream_t)stream_;
    AdamHyper h;
    h.step_size = 0.f;
    h.bc2_sqrt = 1.f;
    h.w1 = (float)(1.0 - (double)beta1);
    h.beta2 = beta2;
    h.one_minus_beta2 = (float)(1.0 - (double)beta2);
    h.eps = eps;
    h.weight_decay = weight_decay;
    // every distinct device schedule advances once
    AdamSchedList adv;
    adv.count = 0;
    for (int32_t t = 0; t < n_tensors; ++t) {
        void* st = sched_states ? sched_states[t] : nullptr;
        if (!st) { LS2FM_CHECK_ARG(lrs && step >= 1); continue; }
        bool seen = false;
        for (int q = 0; q < adv.count; ++q) seen = seen || (adv.st[q] == (AdamSched*)st);
        if (seen) continue;
        if (adv.count == 16) { adam_advance_kernel<<<1, 16, 0, stream>>>(adv, (double)beta1, (double)beta2); adv.count = 0; }
        adv.st[adv.count++] = (AdamSched*)st;
    }
    if (adv.count) adam_advance_kernel<<<1, 16, 0, stream>>>(adv, (double)beta1, (double)beta2);
    const double bc1 = step >= 1 ? 1.0 - pow((double)beta1, (double)step) : 1.0, bc2 = step >= 1 ? 1.0 - pow((double)beta2, (double)step) : 1.0;
    auto host_step = [&](int32_t t) { return (sched_states && sched_states[t]) ? 0.f : (float)((double)lrs[t] / bc1); };
    const float host_bc2 = (float)sqrt(bc2);
    bool done[4096];
    LS2FM_CHECK_ARG(n_tensors <= 4096);
    for (int32_t t = 0; t < n_tensors; ++t) done[t] = numel[t] <= 0;
    // paired tables
    if (mirrors)
        for (int32_t t = 0; t < n_tensors; ++t) {
            if (done[t] || !mirrors[t]) continue;
            LS2FM_CHECK_ARG((reinterpret_cast<uintptr_t>(mirrors[t]) & 7u) == 0 && numel[t] % 2 == 0);
            for (int32_t u = 0; u < n_tensors; ++u) {
                if (u == t || done[u] || !mirrors[u] || numel[u] != numel[t] || mirrors[u] != mirrors[t] + 2) continue;
                const uintptr_t all = reinterpret_cast<uintptr_t>(params[t]) | reinterpret_cast<uintptr_t>(params[u]) |
                                      reinterpret_cast<uintptr_t>(grads[t]) | reinterpret_cast<uintptr_t>(grads[u]) |
                                      reinterpret_cast<uintptr_t>(exp_avg[t]) | reinterpret_cast<uintptr_t>(exp_avg[u]) |
                                      reinterpret_cast<uintptr_t>(exp_avg_sq[t]) | reinterpret_cast<uintptr_t>(exp_avg_sq[u]) |
                                      reinterpret_cast<uintptr_t>(mirrors[t]);
                if (all & 15u) continue;              // unaligned: the two single jobs below
                AdamPair q;
                q.pa = params[t]; q.ga = grads[t]; q.ma = exp_avg[t]; q.va = exp_avg_sq[t];
                q.pb = params[u]; q.gb = grads[u]; q.mb = exp_avg[u]; q.vb = exp_avg_sq[u];
                q.mirror = mirrors[t]; q.n = numel[t];
                q.step_a = host_step(t); q.step_b = host_step(u); q.bc2_a = q.bc2_b = host_bc2;
                q.sched_a = sched_states ? sched_states[t] : nullptr; q.sched_b = sched_states ? sched_states[u] : nullptr;
                const int64_t threads = (numel[t] + 3) / 4;
                adam_pair_kernel<<<(unsigned)((threads + kAdamThreads - 1) / kAdamThreads), kAdamThreads, 0, stream>>>(q, h);
                done[t] = done[u] = true;
                break;
            }
        }
    for (int32_t t = 0; t < n_tensors;) {
        AdamJobs jobs;
        jobs.count = 0;
        int blocks = 0;
        for (; t < n_tensors && jobs.count < kAdamMaxTensors; ++t) {
            if (done[t]) continue;
            LS2FM_CHECK_ARG(params[t] && grads[t] && exp_avg[t] && exp_avg_sq[t]);
            const int j = jobs.count++;
            jobs.p[j] = params[t]; jobs.g[j] = grads[t]; jobs.m[j] = exp_avg[t]; jobs.v[j] = exp_avg_sq[t]; jobs.n[j] = numel[t];
            jobs.mirror[j] = mirrors ? mirrors[t] : nullptr;
            jobs.step_size[j] = host_step(t); jobs.bc2_sqrt[j] = host_bc2;
            jobs.sched[j] = sched_states ? sched_states[t] : nullptr;
            jobs.first_block[j] = blocks;
            blocks += (int)((numel[t] + kAdamTile - 1) / kAdamTile);
        }
        jobs.first_block[jobs.count] = blocks;
        if (blocks) adam_kernel<<<blocks, kAdamThreads, 0, stream>>>(jobs, h);
    }
    return ls2fm_launch_status();
}

extern "C" int ls2fm_adam_sched_decay(int32_t n, void* const* sched_states, void* stream_) {
    LS2FM_CHECK_ARG(n >= 0 && (n == 0 || sched_states));
    hipStream_t stream = (hipStream_t)stream_;
    AdamSchedList list;
    list.count = 0;
    for (int32_t t = 0; t < n; ++t) {
        LS2FM_CHECK_ARG(sched_states[t]);
        bool seen = false;
        for (int q = 0; q < list.count; ++q) seen = seen || (list.st[q] == (AdamSched*)sched_states[t]);
        if (seen) continue;
        if (list.count == 16) { adam_decay_kernel<<<1, 16, 0, stream>>>(list); list.count = 0; }
        list.st[list.count++] = (AdamSched*)sched_states[t];
    }
    if (list.count) adam_decay_kernel<<<1, 16, 0, stream>>>(list);
    return ls2fm_launch_status();
}

namespace {
struct PtrList { void* v[4096]; float lr[4096]; };
}

extern "C" int ls2fm_adam_step(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, void* stream) {
    LS2FM_CHECK_ARG(n_tensors >= 0 && n_tensors <= 4096 && step >= 1);
    static thread_local PtrList pl;
    for (int32_t t = 0; t < n_tensors; ++t) pl.lr[t] = lr;
    return ls2fm_adam_step_multi(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, nullptr, pl.lr, nullptr, beta1, beta2, eps,
                                 weight_decay, step, stream);
}

extern "C" int ls2fm_adam_step_scheduled(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                                         float* const* exp_avg_sq, const int64_t* numel, void* sched_state, float beta1,
                                         float beta2, float eps, float weight_decay, void* stream) {
    LS2FM_CHECK_ARG(n_tensors >= 0 && n_tensors <= 4096 && sched_state);
    static thread_local PtrList pl;
    for (int32_t t = 0; t < n_tensors; ++t) pl.v[t] = sched_state;
    return ls2fm_adam_step_multi(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, nullptr, nullptr, pl.v, beta1, beta2, eps,
                                 weight_decay, 0, stream);
}

extern "C" int ls2fm_adam_step_mirrored(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                                        float* const* exp_avg_sq, const int64_t* numel, float* const* mirrors, void* sched_state,
                                        float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                        void* stream) {
    LS2FM_CHECK_ARG(n_tensors >= 0 && n_tensors <= 4096 && (sched_state || step >= 1));
    static thread_local PtrList pl;
    for (int32_t t = 0; t < n_tensors; ++t) { pl.v[t] = sched_state; pl.lr[t] = lr; }
    return ls2fm_adam_step_multi(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, mirrors, pl.lr, sched_state ? pl.v : nullptr,
                                 beta1, beta2, eps, weight_decay, sched_state ? 0 : step, stream);
}
