// The re-projection term of a bundle-adjustment iteration (pipelines/BA.py:126-147, 199-202): tracked 3-D points, seen through the
// LIVE poses, against their key points -- per observation
//      x_c = R_v x + t_v ;  u = K x_c ;  uv = u_xy / (u_z + 1e-6) ;  err = || uv - key point ||
//      counted when |sdf(x)| < bound and uv is not infinite ;  robust = 2 log(1 + err^2 / 4)
//      reproj = 0.5 mean(robust) + 0.5 mean(err) over the counted observations (0 when there is none)
// and its gradient w.r.t. the points and the poses.  As torch ops (gather of one pose per observation, two batched matrix
// products, divisions, norm, log, three selects, four reductions -- and autograd's mirror of all of them) this was ~100 of the
// ~450 launch-bound kernels of a captured BA iteration.  Observations are SORTED BY VIEW (view v owns [view_start[v],
// view_start[v + 1])): one workgroup per view, fixed-order reductions -- the pose gradients are deterministic (no atomics).
#include "render_common.h"

namespace {

constexpr int kRpThreads = 256;

struct Camera3 { float k[9]; };

struct Obs {                       // forward quantities of one observation
    float xc[3], u[3], e[2], err, inv;
    bool on;
};

__device__ __forceinline__ Obs project(const float* __restrict__ x, const float* __restrict__ pose, const Camera3& K,
                                       const float* __restrict__ uv_obs, const float* __restrict__ sdf, float bound, int64_t i) {
    Obs o;
#pragma unroll
    for (int a = 0; a < 3; ++a)
        o.xc[a] = fmaf(pose[4 * a + 2], x[3 * i + 2], fmaf(pose[4 * a + 1], x[3 * i + 1], pose[4 * a] * x[3 * i])) + pose[4 * a + 3];
#pragma unroll
    for (int a = 0; a < 3; ++a) o.u[a] = fmaf(K.k[3 * a + 2], o.xc[2], fmaf(K.k[3 * a + 1], o.xc[1], K.k[3 * a] * o.xc[0]));
    o.inv = 1.0f / (o.u[2] + 1e-6f);
    const float uv0 = o.u[0] * o.inv, uv1 = o.u[1] * o.inv;
    o.e[0] = uv0 - uv_obs[2 * i];
    o.e[1] = uv1 - uv_obs[2 * i + 1];
    o.err = sqrtf(fmaf(o.e[1], o.e[1], o.e[0] * o.e[0]));
    o.on = (sdf == nullptr || fabsf(sdf[i]) < bound) && !isinf(uv0) && !isinf(uv1);
    return o;
}

// one workgroup per view: partial[v] = {sum robust, sum err, count} over its counted observations, in fp64, fixed order
__global__ void __launch_bounds__(kRpThreads)
reproject_fwd_kernel(const float* __restrict__ x, const float* __restrict__ poses, const int32_t* __restrict__ view_start, Camera3 K,
                     const float* __restrict__ uv_obs, const float* __restrict__ sdf, float bound, float* __restrict__ err_out,
                     uint8_t* __restrict__ on_out, double* __restrict__ partial) {
    const int v = blockIdx.x, tid = threadIdx.x;
    __shared__ float s_pose[12];
    __shared__ double s_red[kRpThreads][3];
    if (tid < 12) s_pose[tid] = poses[12 * v + tid];
    __syncthreads();
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t i = view_start[v] + tid; i < view_start[v + 1]; i += kRpThreads) {
        const Obs o = project(x, s_pose, K, uv_obs, sdf, bound, i);
        const float err = o.on ? o.err : 0.f;
        err_out[i] = err;
        on_out[i] = o.on ? 1 : 0;
        if (o.on) {
            acc[0] += (double)(2.0f * logf(1.0f + err * err / 4.0f));
            acc[1] += (double)err;
            acc[2] += 1.0;
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) s_red[tid][q] = acc[q];
    __syncthreads();
    if (tid < 3) {
        double t = 0.0;
        for (int k = 0; k < kRpThreads; ++k) t += s_red[k][tid];
        partial[3 * v + tid] = t;
    }
}

// sums = {sum robust, sum err, count, reproj}
__global__ void reproject_finish_kernel(const double* __restrict__ partial, int n_views, double* __restrict__ sums) {
    double t[3] = {0.0, 0.0, 0.0};
    for (int v = 0; v < n_views; ++v)
        for (int q = 0; q < 3; ++q) t[q] += partial[3 * v + q];
    sums[0] = t[0]; sums[1] = t[1]; sums[2] = t[2];
    sums[3] = t[2] > 0.0 ? (double)(0.5f * (float)t[0] / (float)t[2] + 0.5f * (float)t[1] / (float)t[2]) : 0.0;
}

// one workgroup per view: d_points of its observations, d_pose[v] as a fixed-order sum over them
__global__ void __launch_bounds__(kRpThreads)
reproject_bwd_kernel(const float* __restrict__ x, const float* __restrict__ poses, const int32_t* __restrict__ view_start, Camera3 K,
                     const float* __restrict__ uv_obs, const float* __restrict__ sdf, float bound, const double* __restrict__ sums,
                     const float* __restrict__ d_reproj, float* __restrict__ d_x, float* __restrict__ d_poses) {
    const int v = blockIdx.x, tid = threadIdx.x;
    __shared__ float s_pose[12];
    __shared__ float s_red[kRpThreads][12];
    if (tid < 12) s_pose[tid] = poses[12 * v + tid];
    __syncthreads();
    const float count = (float)sums[2];
    const float g = count > 0.f ? d_reproj[0] * 0.5f / count : 0.f;       // d reproj / d (sum robust) = d / d (sum err)
    float acc[12];
#pragma unroll
    for (int q = 0; q < 12; ++q) acc[q] = 0.f;
    for (int64_t i = view_start[v] + tid; i < view_start[v + 1]; i += kRpThreads) {
        const Obs o = project(x, s_pose, K, uv_obs, sdf, bound, i);
        float dx[3] = {0.f, 0.f, 0.f};
        if (o.on && o.err > 0.f) {
            // d err: 1 from the plain mean, err / (1 + err^2 / 4) from the robust one
            const float d_err = g * (1.0f + o.err / (1.0f + o.err * o.err / 4.0f));
            const float de0 = d_err * o.e[0] / o.err, de1 = d_err * o.e[1] / o.err;
            const float du[3] = {de0 * o.inv, de1 * o.inv, -(de0 * o.u[0] + de1 * o.u[1]) * o.inv * o.inv};
            float dxc[3];
#pragma unroll
            for (int b = 0; b < 3; ++b) dxc[b] = fmaf(K.k[6 + b], du[2], fmaf(K.k[3 + b], du[1], K.k[b] * du[0]));      // K^T du
#pragma unroll
            for (int b = 0; b < 3; ++b) dx[b] = fmaf(s_pose[8 + b], dxc[2], fmaf(s_pose[4 + b], dxc[1], s_pose[b] * dxc[0]));   // R^T
#pragma unroll
            for (int a = 0; a < 3; ++a) {
#pragma unroll
                for (int b = 0; b < 3; ++b) acc[4 * a + b] = fmaf(dxc[a], x[3 * i + b], acc[4 * a + b]);
                acc[4 * a + 3] += dxc[a];
            }
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) d_x[3 * i + b] = dx[b];
    }
#pragma unroll
    for (int q = 0; q < 12; ++q) s_red[tid][q] = acc[q];
    __syncthreads();
    if (tid < 12) {
        float t = 0.f;
        for (int k = 0; k < kRpThreads; ++k) t += s_red[k][tid];
        d_poses[12 * v + tid] = t;
    }
}

bool load_intrinsic(const float* host_k, Camera3* out) {
    if (!host_k) return false;
    for (int q = 0; q < 9; ++q) out->k[q] = host_k[q];
    return true;
}

}  // namespace

extern "C" int64_t ls2fm_reproject_workspace_bytes(int32_t n_views) { return (int64_t)sizeof(double) * 3 * (n_views > 0 ? n_views : 1); }

extern "C" int ls2fm_reproject_fwd(const float* points, const float* poses, const int32_t* view_start, int32_t n_views,
                                   const float* intrinsic_host, const float* obs_uv, const float* sdf, float sdf_bound, int64_t n,
                                   float* err, uint8_t* on, double* sums, void* workspace, void* stream) {
    LS2FM_CHECK_ARG(n >= 0 && n_views >= 1 && poses && view_start && sums);
    Camera3 K;
    LS2FM_CHECK_ARG(load_intrinsic(intrinsic_host, &K));
    if (!workspace) return LS2FM_ERR_WORKSPACE;
    LS2FM_CHECK_ARG(n == 0 || (points && obs_uv && err && on));
    hipStream_t s = (hipStream_t)stream;
    reproject_fwd_kernel<<<(unsigned)n_views, kRpThreads, 0, s>>>(points, poses, view_start, K, obs_uv, sdf, sdf_bound, err, on,
                                                                  (double*)workspace);
    reproject_finish_kernel<<<1, 1, 0, s>>>((const double*)workspace, n_views, sums);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_reproject_bwd(const float* points, const float* poses, const int32_t* view_start, int32_t n_views,
                                   const float* intrinsic_host, const float* obs_uv, const float* sdf, float sdf_bound, int64_t n,
                                   const double* sums, const float* d_reproj, float* d_points, float* d_poses, void* stream) {
    LS2FM_CHECK_ARG(n >= 0 && n_views >= 1 && poses && view_start && sums && d_reproj && d_poses);
    Camera3 K;
    LS2FM_CHECK_ARG(load_intrinsic(intrinsic_host, &K));
    LS2FM_CHECK_ARG(n == 0 || (points && obs_uv && d_points));
    reproject_bwd_kernel<<<(unsigned)n_views, kRpThreads, 0, (hipStream_t)stream>>>(points, poses, view_start, K, obs_uv, sdf, sdf_bound,
                                                                                 sums, d_reproj, d_points, d_poses);
    return ls2fm_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// The tracing-consistency term of a stage-loop iteration (pipelines/Camera.py:139-143 `get_pts3D` + the callers' loss lines,
// BA.py:155-161 / Registration): the traced key points' surface points against their tracked 3-D points,
//      surface_i = center_i + ray_i d_i ;  w_i = live_i / sum(live) ;  tracing = sum_i |target_i - surface_i| w_i ;
//      sdf_surf = sum_i |sdf_last_i| w_i
// and its gradient w.r.t. the traced depths d and the last SDF values.  As torch ops (addcmul, sum, div, sub, norm, dot, abs, dot
// and autograd's mirror) ~18 of a captured iteration's nodes, each >= 4.6 us on the device's timeline whatever its size; ONE
// workgroup each way, fp64 fixed-order sums (deterministic).
namespace {

constexpr int kTtThreads = 1024;

__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int q = 0; q < kTtThreads / 64; ++q) t += red[q];
    return t;
}

__global__ void __launch_bounds__(kTtThreads)
tracing_term_fwd_kernel(const float* __restrict__ center, const float* __restrict__ ray, const float* __restrict__ d,
                        const float* __restrict__ target, const float* __restrict__ live, const float* __restrict__ sdf_last, int64_t n,
                        float* __restrict__ out) {
    __shared__ double red[kTtThreads / 64];
    const int tid = threadIdx.x;
    double s_live = 0.0, s_tr = 0.0, s_sd = 0.0;
    for (int64_t i = tid; i < n; i += kTtThreads) {
        const float lv = live[i];
        float q = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float e = target[3 * i + a] - fmaf(ray[3 * i + a], d[i], center[3 * i + a]);
            q = fmaf(e, e, q);
        }
        s_live += (double)lv;
        s_tr += (double)(sqrtf(q) * lv);
        if (sdf_last) s_sd += (double)(fabsf(sdf_last[i]) * lv);
    }
    const double tl = block_sum(s_live, red, tid), tt = block_sum(s_tr, red, tid), ts = block_sum(s_sd, red, tid);
    if (tid == 0) {
        out[0] = (float)(tt / tl);
        out[1] = (float)(ts / tl);
        out[2] = (float)tl;
    }
}

__global__ void __launch_bounds__(256)
tracing_term_bwd_kernel(const float* __restrict__ center, const float* __restrict__ ray, const float* __restrict__ d,
                        const float* __restrict__ target, const float* __restrict__ live, const float* __restrict__ sdf_last, int64_t n,
                        const float* __restrict__ out, const float* __restrict__ g_tl, const float* __restrict__ g_sd, float* __restrict__ d_d,
                        float* __restrict__ d_sdf) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float g0 = g_tl ? g_tl[0] : 0.f, g1 = g_sd ? g_sd[0] : 0.f;          // (a term nobody differentiates: no upstream)
    const float w = live[i] / out[2];
    float e[3], q = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        e[a] = target[3 * i + a] - fmaf(ray[3 * i + a], d[i], center[3 * i + a]);
        q = fmaf(e[a], e[a], q);
    }
    const float len = sqrtf(q);
    // d |e| / d d = -(e . ray) / |e|   (0 at e = 0, as torch's norm backward)
    const float dot = fmaf(e[2], ray[3 * i + 2], fmaf(e[1], ray[3 * i + 1], e[0] * ray[3 * i]));
    d_d[i] = len > 0.f ? -(g0 * w) * dot / len : 0.f;
    if (d_sdf) {
        const float s = sdf_last[i];
        d_sdf[i] = g1 * w * (s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f));
    }
}

}  // namespace

extern "C" int ls2fm_tracing_term_fwd(const float* center, const float* ray, const float* d, const float* target, const float* live,
                                      const float* sdf_last, int64_t n, float* out, void* stream) {
    LS2FM_CHECK_ARG(n >= 1 && center && ray && d && target && live && out);
    tracing_term_fwd_kernel<<<1, kTtThreads, 0, (hipStream_t)stream>>>(center, ray, d, target, live, sdf_last, n, out);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_tracing_term_bwd(const float* center, const float* ray, const float* d, const float* target, const float* live,
                                      const float* sdf_last, int64_t n, const float* out, const float* g_tl, const float* g_sd, float* d_d,
                                      float* d_sdf, void* stream) {
    LS2FM_CHECK_ARG(n >= 1 && center && ray && d && target && live && out && d_d && ((sdf_last != nullptr) == (d_sdf != nullptr)));
    tracing_term_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(center, ray, d, target, live, sdf_last, n, out, g_tl,
                                                                                        g_sd, d_d, d_sdf);
    return ls2fm_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// The loss lines of a bundle-adjustment iteration outside the render (pipelines/BA.py:160-170 of the reference):
//   sdf_surf = mean |sdfs| ;  w_reproj = reproj > thresh ? w_hi : w_lo (the adaptive weight, from the DETACHED error) ;
//   extra = w_reproj reproj + w_surf sdf_surf + w_add add
// -- nine elementwise / reduction torch kernels and seven of autograd's, one launch each way (a captured iteration pays >= 4.6 us
// per graph node, whatever its size).
namespace {
__global__ void __launch_bounds__(kTtThreads)
ba_terms_fwd_kernel(const float* __restrict__ reproj, const float* __restrict__ sdfs, int64_t n, const float* __restrict__ add, float thresh,
                    float w_lo, float w_hi, float w_surf, float w_add, float* __restrict__ out_surf, float* __restrict__ out_w,
                    float* __restrict__ out_extra) {
    __shared__ double red[kTtThreads / 64];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int64_t i = tid; i < n; i += kTtThreads) s += (double)fabsf(sdfs[i]);
    const double tot = block_sum(s, red, tid);
    if (tid == 0) {
        const float surf = (float)(tot / (double)n);
        const float r = reproj[0];
        const float w = r > thresh ? w_hi : w_lo;
        float e = fmaf(w_surf, surf, w * r);
        if (add) e = fmaf(w_add, add[0], e);
        out_surf[0] = surf; out_w[0] = w; out_extra[0] = e;
    }
}

__global__ void __launch_bounds__(256)
ba_terms_bwd_kernel(const float* __restrict__ sdfs, int64_t n, const float* __restrict__ w_reproj, const float* __restrict__ g, float w_surf,
                    float w_add, float* __restrict__ d_reproj, float* __restrict__ d_sdfs, float* __restrict__ d_add) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const float gg = g[0];
    if (i == 0) {
        d_reproj[0] = gg * w_reproj[0];
        if (d_add) d_add[0] = gg * w_add;
    }
    if (i < n) {
        const float s = sdfs[i];
        d_sdfs[i] = (gg * w_surf / (float)n) * (s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f));       // sign(0) = 0, as torch's abs backward
    }
}
}  // namespace

namespace {
__global__ void weighted_pair_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float wa, float wb, float* __restrict__ out) {
    out[0] = fmaf(wb, b[0], wa * a[0]);
}
__global__ void weighted_pair_bwd_kernel(const float* __restrict__ g, float wa, float wb, float* __restrict__ d2) {
    d2[0] = g[0] * wa;
    d2[1] = g[0] * wb;
}
}  // namespace

// wa a + wb b of two device scalars, and its gradient as ONE two-float buffer (the loops' weighted sums of two loss terms: two
// multiplications and an addition, two more multiplications in the backward -- graph nodes of a captured iteration)
extern "C" int ls2fm_weighted_pair_fwd(const float* a, const float* b, float wa, float wb, float* out, void* stream) {
    LS2FM_CHECK_ARG(a && b && out);
    weighted_pair_fwd_kernel<<<1, 1, 0, (hipStream_t)stream>>>(a, b, wa, wb, out);
    return ls2fm_launch_status();
}
extern "C" int ls2fm_weighted_pair_bwd(const float* g, float wa, float wb, float* d2, void* stream) {
    LS2FM_CHECK_ARG(g && d2);
    weighted_pair_bwd_kernel<<<1, 1, 0, (hipStream_t)stream>>>(g, wa, wb, d2);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_ba_terms_fwd(const float* reproj, const float* sdfs, int64_t n, const float* add, float thresh, float w_lo, float w_hi,
                                  float w_surf, float w_add, float* out_surf, float* out_w, float* out_extra, void* stream) {
    LS2FM_CHECK_ARG(n >= 1 && reproj && sdfs && out_surf && out_w && out_extra);
    ba_terms_fwd_kernel<<<1, kTtThreads, 0, (hipStream_t)stream>>>(reproj, sdfs, n, add, thresh, w_lo, w_hi, w_surf, w_add, out_surf, out_w,
                                                                  out_extra);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_ba_terms_bwd(const float* sdfs, int64_t n, const float* w_reproj, const float* g, float w_surf, float w_add,
                                  float* d_reproj, float* d_sdfs, float* d_add, void* stream) {
    LS2FM_CHECK_ARG(n >= 1 && sdfs && w_reproj && g && d_reproj && d_sdfs);
    ba_terms_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(sdfs, n, w_reproj, g, w_surf, w_add, d_reproj, d_sdfs, d_add);
    return ls2fm_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// SDF.get_surface_pts' last line (models/SDF.py:104-110 of the reference):  out = p - n / |n|.detach() * sdf ,  length = |n|
// -- four elementwise torch kernels and ~10 of autograd's, one launch each way.  The norm is DETACHED inside the quotient: the
// gradient reaches n through the product only, and through `length`.
namespace {

__global__ void __launch_bounds__(256)
surface_pts_fwd_kernel(const float* __restrict__ p, const float* __restrict__ nrm, const float* __restrict__ sdf, int64_t n,
                       float* __restrict__ out, float* __restrict__ length) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float a = nrm[3 * i], b = nrm[3 * i + 1], c = nrm[3 * i + 2];
    const float len = sqrtf(a * a + b * b + c * c);
    const float s = sdf[i];
    out[3 * i] = p[3 * i] - a / len * s;
    out[3 * i + 1] = p[3 * i + 1] - b / len * s;
    out[3 * i + 2] = p[3 * i + 2] - c / len * s;
    length[i] = len;
}

__global__ void __launch_bounds__(256)
surface_pts_bwd_kernel(const float* __restrict__ nrm, const float* __restrict__ sdf, const float* __restrict__ length, int64_t n,
                       const float* __restrict__ g_out, const float* __restrict__ g_len, float* __restrict__ d_n, float* __restrict__ d_sdf) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float len = length[i], s = sdf[i], gl = g_len ? g_len[i] : 0.f;
    float ds = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float na = nrm[3 * i + a], go = g_out ? g_out[3 * i + a] : 0.f;
        const float u = na / len;
        ds = fmaf(-go, u, ds);
        // out_a = p_a - n_a / len_detached * sdf ;  length = |n| (d |n| / d n = n / |n|; 0 at n = 0 as torch's norm backward)
        d_n[3 * i + a] = -go * s / len + (len > 0.f ? gl * u : 0.f);
    }
    d_sdf[i] = ds;
}

}  // namespace

extern "C" int ls2fm_surface_pts_fwd(const float* p, const float* normals, const float* sdf, int64_t n, float* out, float* length,
                                     void* stream) {
    LS2FM_CHECK_ARG(n >= 0);
    if (n == 0) return LS2FM_OK;
    LS2FM_CHECK_ARG(p && normals && sdf && out && length);
    surface_pts_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(p, normals, sdf, n, out, length);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_surface_pts_bwd(const float* normals, const float* sdf, const float* length, int64_t n, const float* g_out,
                                     const float* g_length, float* d_normals, float* d_sdf, void* stream) {
    LS2FM_CHECK_ARG(n >= 0);
    if (n == 0) return LS2FM_OK;
    LS2FM_CHECK_ARG(normals && sdf && length && d_normals && d_sdf);
    surface_pts_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(normals, sdf, length, n, g_out, g_length,
                                                                                       d_normals, d_sdf);
    return ls2fm_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------
// The explicit-match terms of the two-view initialisation (pipelines/Initialization.py:154-160, 252-255 on Camera.py:136,
// 168-178): the key points of view v are traced onto the surface (d, sdf_last from SDF.sphere_tracing), the surface points
//      pts = center + ray d
// are projected into the OTHER view (fixed poses) and compared with the matched key points there,
//      reproj_error = mean || uv(pts) - key point || ,   sdf_surf = mean |sdf_last|      over the segments' points together.
// As torch ops (two batched matrix products, divisions, norm, concatenations, means per view, and autograd's mirror) ~70 of a
// captured iteration's graph nodes; ONE workgroup each way, fp64 fixed-order sums.  Up to kMtSeg segments (views) of n points.
namespace {

constexpr int kMtSeg = 4;
struct MatchSegs { const float* d[kMtSeg]; const float* s[kMtSeg]; float* dd[kMtSeg]; float* ds[kMtSeg]; };

struct MatchPt { float p[3], u[3], e[2], err, inv; };

__device__ __forceinline__ MatchPt match_project(const float* __restrict__ center, const float* __restrict__ ray, float d,
                                                 const float* __restrict__ pose, const Camera3& K, const float* __restrict__ uv_obs,
                                                 int64_t i) {
    MatchPt o;
#pragma unroll
    for (int a = 0; a < 3; ++a) o.p[a] = fmaf(ray[3 * i + a], d, center[3 * i + a]);
    float xc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
        xc[a] = fmaf(pose[4 * a + 2], o.p[2], fmaf(pose[4 * a + 1], o.p[1], pose[4 * a] * o.p[0])) + pose[4 * a + 3];
#pragma unroll
    for (int a = 0; a < 3; ++a) o.u[a] = fmaf(K.k[3 * a + 2], xc[2], fmaf(K.k[3 * a + 1], xc[1], K.k[3 * a] * xc[0]));
    o.inv = 1.0f / (o.u[2] + 1e-6f);
    o.e[0] = o.u[0] * o.inv - uv_obs[2 * i];
    o.e[1] = o.u[1] * o.inv - uv_obs[2 * i + 1];
    o.err = sqrtf(fmaf(o.e[1], o.e[1], o.e[0] * o.e[0]));
    return o;
}

__global__ void __launch_bounds__(kTtThreads)
match_term_fwd_kernel(const float* __restrict__ center, const float* __restrict__ ray, const float* __restrict__ uv_obs,
                      const float* __restrict__ poses, Camera3 K, int n_seg, int64_t n, MatchSegs sg, float* __restrict__ surface,
                      float* __restrict__ out) {
    __shared__ double red[kTtThreads / 64];
    __shared__ float s_pose[kMtSeg][12];
    const int tid = threadIdx.x;
    if (tid < 12 * n_seg) s_pose[tid / 12][tid % 12] = poses[tid];
    __syncthreads();
    double s_err = 0.0, s_sdf = 0.0;
    for (int seg = 0; seg < n_seg; ++seg)
        for (int64_t j = tid; j < n; j += kTtThreads) {
            const int64_t i = (int64_t)seg * n + j;
            const MatchPt o = match_project(center, ray, sg.d[seg][j], s_pose[seg], K, uv_obs, i);
#pragma unroll
            for (int a = 0; a < 3; ++a) surface[3 * i + a] = o.p[a];
            s_err += (double)o.err;
            s_sdf += (double)fabsf(sg.s[seg][j]);
        }
    const double te = block_sum(s_err, red, tid), ts = block_sum(s_sdf, red, tid);
    if (tid == 0) {
        const double cnt = (double)n_seg * (double)n;
        out[0] = (float)(te / cnt);
        out[1] = (float)(ts / cnt);
    }
}

__global__ void __launch_bounds__(256)
match_term_bwd_kernel(const float* __restrict__ center, const float* __restrict__ ray, const float* __restrict__ uv_obs,
                      const float* __restrict__ poses, Camera3 K, int n_seg, int64_t n, MatchSegs sg, const float* __restrict__ g) {
    const int seg = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int64_t i = (int64_t)seg * n + j;
    const float* __restrict__ pose = poses + 12 * seg;
    const float inv_cnt = 1.0f / ((float)n_seg * (float)n);
    const MatchPt o = match_project(center, ray, sg.d[seg][j], pose, K, uv_obs, i);
    float dd = 0.f;
    if (o.err > 0.f) {                                   // norm backward at 0: zero (torch)
        const float ge = g[0] * inv_cnt;
        const float de0 = ge * o.e[0] / o.err, de1 = ge * o.e[1] / o.err;
        const float du[3] = {de0 * o.inv, de1 * o.inv, -(de0 * o.u[0] + de1 * o.u[1]) * o.inv * o.inv};
        float dxc[3], dp[3];
#pragma unroll
        for (int b = 0; b < 3; ++b) dxc[b] = fmaf(K.k[6 + b], du[2], fmaf(K.k[3 + b], du[1], K.k[b] * du[0]));        // K^T du
#pragma unroll
        for (int b = 0; b < 3; ++b) dp[b] = fmaf(pose[8 + b], dxc[2], fmaf(pose[4 + b], dxc[1], pose[b] * dxc[0]));    // R^T
        dd = fmaf(dp[2], ray[3 * i + 2], fmaf(dp[1], ray[3 * i + 1], dp[0] * ray[3 * i]));
    }
    sg.dd[seg][j] = dd;
    const float s = sg.s[seg][j];
    sg.ds[seg][j] = g[1] * inv_cnt * (s > 0.f ? 1.f : (s < 0.f ? -1.f : 0.f));
}

bool load_segs(int32_t n_seg, const float* const* d, const float* const* s, float* const* dd, float* const* ds, MatchSegs* out) {
    if (n_seg < 1 || n_seg > kMtSeg || !d || !s) return false;
    for (int q = 0; q < kMtSeg; ++q) {
        out->d[q] = q < n_seg ? d[q] : nullptr;
        out->s[q] = q < n_seg ? s[q] : nullptr;
        out->dd[q] = (q < n_seg && dd) ? dd[q] : nullptr;
        out->ds[q] = (q < n_seg && ds) ? ds[q] : nullptr;
        if (q < n_seg && (!out->d[q] || !out->s[q] || (dd && !out->dd[q]) || (ds && !out->ds[q]))) return false;
    }
    return true;
}

}  // namespace

extern "C" int ls2fm_match_term_fwd(const float* center, const float* ray, const float* uv_obs, const float* poses,
                                    const float* intrinsic_host, int32_t n_seg, int64_t n, const float* const* d,
                                    const float* const* sdf_last, float* surface, float* out, void* stream) {
    Camera3 K;
    MatchSegs sg;
    LS2FM_CHECK_ARG(n >= 1 && center && ray && uv_obs && poses && surface && out && load_intrinsic(intrinsic_host, &K));
    LS2FM_CHECK_ARG(load_segs(n_seg, d, sdf_last, nullptr, nullptr, &sg));
    match_term_fwd_kernel<<<1, kTtThreads, 0, (hipStream_t)stream>>>(center, ray, uv_obs, poses, K, n_seg, n, sg, surface, out);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_match_term_bwd(const float* center, const float* ray, const float* uv_obs, const float* poses,
                                    const float* intrinsic_host, int32_t n_seg, int64_t n, const float* const* d,
                                    const float* const* sdf_last, const float* g, float* const* d_d, float* const* d_sdf,
                                    void* stream) {
    Camera3 K;
    MatchSegs sg;
    LS2FM_CHECK_ARG(n >= 1 && center && ray && uv_obs && poses && g && d_d && d_sdf && load_intrinsic(intrinsic_host, &K));
    LS2FM_CHECK_ARG(load_segs(n_seg, d, sdf_last, d_d, d_sdf, &sg));
    match_term_bwd_kernel<<<dim3((unsigned)((n + 255) / 256), (unsigned)n_seg), 256, 0, (hipStream_t)stream>>>(center, ray, uv_obs, poses, K,
                                                                                                            n_seg, n, sg, g);
    return ls2fm_launch_status();
}
