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

__global__ void adam_advance_kernel(AdamSched* st, double beta1, double beta2) {
    st->step += 1.0;
    const double bc1 = 1.0 - pow(beta1, st->step), bc2 = 1.0 - pow(beta2, st->step);
    st->step_size = (float)(st->lr / bc1);
    st->bc2_sqrt = (float)sqrt(bc2);
    st->lr *= st->gamma;
}

__global__ void __launch_bounds__(kAdamThreads)
adam_kernel(AdamJobs jobs, AdamHyper h, const AdamSched* __restrict__ sched) {
    if (sched) { h.step_size = sched->step_size; h.bc2_sqrt = sched->bc2_sqrt; }
    int j = 0;
    while (j + 1 < jobs.count && (int)blockIdx.x >= jobs.first_block[j + 1]) ++j;
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

}  // namespace

static int adam_launch(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* numel, AdamHyper h, const AdamSched* sched, hipStream_t stream,
                       float* const* mirrors = nullptr);

extern "C" int ls2fm_adam_step(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2, float eps,
                               float weight_decay, int64_t step, void* stream) {
    LS2FM_CHECK_ARG(n_tensors >= 0 && (n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq && numel)) && step >= 1);
    LS2FM_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f);
    // the scalars as torch forms them (Python doubles, then rounded to fp32 where they meet the tensors)
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    AdamHyper h;
    h.step_size = (float)((double)lr / bc1);
    h.bc2_sqrt = (float)sqrt(bc2);
    h.w1 = (float)(1.0 - (double)beta1);
    h.beta2 = beta2;
    h.one_minus_beta2 = (float)(1.0 - (double)beta2);
    h.eps = eps;
    h.weight_decay = weight_decay;
    return adam_launch(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, h, nullptr, (hipStream_t)stream);
}

extern "C" int ls2fm_adam_step_scheduled(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                                         float* const* exp_avg_sq, const int64_t* numel, void* sched_state, float beta1,
                                         float beta2, float eps, float weight_decay, void* stream) {
    LS2FM_CHECK_ARG(n_tensors >= 0 && (n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq && numel)) && sched_state);
    LS2FM_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f);
    static_assert(sizeof(AdamSched) == 32, "ls2fm_adam_step_scheduled: sched_state is 32 bytes {double step, lr, gamma; float[2]}");
    adam_advance_kernel<<<1, 1, 0, (hipStream_t)stream>>>((AdamSched*)sched_state, (double)beta1, (double)beta2);
    AdamHyper h;
    h.step_size = 0.f;
    h.bc2_sqrt = 1.f;
    h.w1 = (float)(1.0 - (double)beta1);
    h.beta2 = beta2;
    h.one_minus_beta2 = (float)(1.0 - (double)beta2);
    h.eps = eps;
    h.weight_decay = weight_decay;
    return adam_launch(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, h, (const AdamSched*)sched_state, (hipStream_t)stream);
}

// The two entry points above with a mirror list: mirrors[t] (HOST array of device pointers, entries may be null) = where tensor
// t's updated values are ALSO written, entry-interleaved (see AdamJobs::mirror); sched_state null = the unscheduled form (lr,
// step from the arguments), else the scheduled form (lr, step ignored).
extern "C" int ls2fm_adam_step_mirrored(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                                        float* const* exp_avg_sq, const int64_t* numel, float* const* mirrors, void* sched_state,
                                        float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                                        void* stream) {
    LS2FM_CHECK_ARG(n_tensors >= 0 && (n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq && numel)));
    LS2FM_CHECK_ARG(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && (sched_state || step >= 1));
    if (mirrors)
        for (int32_t t = 0; t < n_tensors; ++t)
            LS2FM_CHECK_ARG(!mirrors[t] || ((reinterpret_cast<uintptr_t>(mirrors[t]) & 7u) == 0 && numel[t] % 2 == 0));
    AdamHyper h;
    h.step_size = 0.f;
    h.bc2_sqrt = 1.f;
    if (sched_state) {
        adam_advance_kernel<<<1, 1, 0, (hipStream_t)stream>>>((AdamSched*)sched_state, (double)beta1, (double)beta2);
    } else {
        const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
        h.step_size = (float)((double)lr / bc1);
        h.bc2_sqrt = (float)sqrt(bc2);
    }
    h.w1 = (float)(1.0 - (double)beta1);
    h.beta2 = beta2;
    h.one_minus_beta2 = (float)(1.0 - (double)beta2);
    h.eps = eps;
    h.weight_decay = weight_decay;
    return adam_launch(n_tensors, params, grads, exp_avg, exp_avg_sq, numel, h, (const AdamSched*)sched_state, (hipStream_t)stream, mirrors);
}

static int adam_launch(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* numel, AdamHyper h, const AdamSched* sched, hipStream_t stream,
                       float* const* mirrors) {
    for (int32_t t = 0; t < n_tensors;) {
        AdamJobs jobs;
        jobs.count = 0;
        int blocks = 0;
        for (; t < n_tensors && jobs.count < kAdamMaxTensors; ++t) {
            if (numel[t] <= 0) continue;
            LS2FM_CHECK_ARG(params[t] && grads[t] && exp_avg[t] && exp_avg_sq[t]);
            const int j = jobs.count++;
            jobs.p[j] = params[t]; jobs.g[j] = grads[t]; jobs.m[j] = exp_avg[t]; jobs.v[j] = exp_avg_sq[t]; jobs.n[j] = numel[t];
            jobs.mirror[j] = mirrors ? mirrors[t] : nullptr;
            jobs.first_block[j] = blocks;
            blocks += (int)((numel[t] + kAdamTile - 1) / kAdamTile);
        }
        jobs.first_block[jobs.count] = blocks;
        if (blocks) adam_kernel<<<blocks, kAdamThreads, 0, stream>>>(jobs, h, sched);
    }
    return ls2fm_launch_status();
}
