// The tail of SDF.sphere_tracing for a whole ray batch in one launch each way (models/SDF.py:203-214 and the mask lines of
// CameraSet.render, pipelines/Camera.py:515-516): from the graph-enabled SDF values of the track points
//      d_pred    = min-like clamp:  d = near + sum_{k < max(K,1)} sdf[r][k];  d > far ? far : d        (SDF.py:205-206)
//      sdf_last  = sdf[r][max(K,1) - 1]                                                                 (SDF.py:211)
//      finish    = |sdf_last| < (bound extent) / 10 / Res                                               (SDF.py:213-214)
//      mask_bg   = 0.05 < mean(rgb_gt[r]) < 0.95 ;  mask_dc = finish & mask_bg                          (Camera.py:515-516)
// with K read from the DEVICE (the trip count sphere_trace left there), so that a captured stage step never returns to the
// host.  As torch ops this was ~25 launch-bound elementwise kernels (130 us of a 1.2 ms step).  The backward spreads the
// upstream of d_pred (and of sdf_last) over the K live track points of every ray that was not clamped.
#include "render_common.h"

namespace {

__global__ void __launch_bounds__(256)
trace_depth_fwd_kernel(const float* __restrict__ sdf_tracks, const int32_t* __restrict__ trips, const float* __restrict__ near,
                       const float* __restrict__ far, int64_t n_rays, int k_max, float finish_thr,
                       const float* __restrict__ rgb_gt, float bg_lo, float bg_hi, float* __restrict__ d_pred,
                       float* __restrict__ sdf_last, uint8_t* __restrict__ finish, uint8_t* __restrict__ mask_bg,
                       uint8_t* __restrict__ mask_dc, uint8_t* __restrict__ gate) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    int k_eff = trips[0] < 1 ? 1 : trips[0];
    if (k_eff > k_max) k_eff = k_max;
    const float* row = sdf_tracks + r * k_max;
    float acc = 0.f, last = 0.f;
    for (int k = 0; k < k_eff; ++k) { last = row[k]; acc += last; }
    const float d = acc + near[r];
    const bool clamped = d > far[r];
    d_pred[r] = clamped ? far[r] : d;
    sdf_last[r] = last;
    const bool fin = fabsf(last) < finish_thr;
    finish[r] = fin ? 1 : 0;
    gate[r] = clamped ? 0 : 1;
    bool bg = true;
    if (rgb_gt) {
        const float gray = (rgb_gt[3 * r] + rgb_gt[3 * r + 1] + rgb_gt[3 * r + 2]) / 3.0f;
        bg = gray < bg_hi && gray > bg_lo;
    }
    if (mask_bg) mask_bg[r] = bg ? 1 : 0;
    if (mask_dc) mask_dc[r] = (fin && bg) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
trace_depth_bwd_kernel(const float* __restrict__ d_dpred, const float* __restrict__ d_last, const int32_t* __restrict__ trips,
                       const uint8_t* __restrict__ gate, int64_t n_rays, int k_max, float* __restrict__ d_sdf_tracks) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays * k_max) return;
    const int64_t r = i / k_max;
    const int k = (int)(i - r * k_max);
    int k_eff = trips[0] < 1 ? 1 : trips[0];
    if (k_eff > k_max) k_eff = k_max;
    float g = 0.f;
    if (k < k_eff && gate[r] && d_dpred) g = d_dpred[r];
    if (k == k_eff - 1 && d_last) g += d_last[r];
    d_sdf_tracks[i] = g;
}

}  // namespace

extern "C" int ls2fm_trace_depth_fwd(const float* sdf_tracks, const int32_t* trips, const float* near, const float* far,
                                     int64_t n_rays, int32_t k_max, float finish_threshold, const float* rgb_gt, float bg_lo,
                                     float bg_hi, float* d_pred, float* sdf_last, uint8_t* finish, uint8_t* mask_bg,
                                     uint8_t* mask_dc, uint8_t* gate, void* stream) {
    LS2FM_CHECK_ARG(sdf_tracks && trips && near && far && d_pred && sdf_last && finish && gate && n_rays >= 0 && k_max >= 1);
    if (n_rays == 0) return LS2FM_OK;
    trace_depth_fwd_kernel<<<(unsigned)((n_rays + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        sdf_tracks, trips, near, far, n_rays, k_max, finish_threshold, rgb_gt, bg_lo, bg_hi, d_pred, sdf_last, finish, mask_bg,
        mask_dc, gate);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_trace_depth_bwd(const float* d_dpred, const float* d_sdf_last, const int32_t* trips, const uint8_t* gate,
                                     int64_t n_rays, int32_t k_max, float* d_sdf_tracks, void* stream) {
    LS2FM_CHECK_ARG(trips && gate && d_sdf_tracks && n_rays >= 0 && k_max >= 1);
    if (n_rays == 0) return LS2FM_OK;
    const int64_t n = n_rays * k_max;
    trace_depth_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_dpred, d_sdf_last, trips, gate, n_rays,
                                                                                        k_max, d_sdf_tracks);
    return ls2fm_launch_status();
}
