// Pixels -> camera centers and ray directions through V pinhole cameras, optionally with the se(3) -> SE(3) exponential in
// front: the no-gradient "pose algebra" preamble of a stage-loop iteration (pipelines/Camera.py:129-133, 457-463 with
// utils/camera.py:63-147, 230-252 of the reference) as ONE launch.
//
// As torch ops that preamble is ~95 launch-bound kernels per BA iteration (the Taylor series of the exponential, skew / W @ W /
// V @ u as 3 x 3 hipBLASLt products, K^-1 and [R^T | -R^T t] products, concatenations, index selects): about a millisecond of
// the iteration's 3 (profiles/r04_ba_loop_timeline.txt), a quarter of it before the first kernel of the path proper.
//
// Arithmetic, in the reference's order (every product sum is an fma chain over k = 0, 1, .. with the first product rounded on
// its own -- what a single f32 MFMA block computes for these 3- and 4-term sums; elementwise expressions are separate
// multiplications and additions: -ffp-contract=off):
//     theta = sqrt(w0^2 + w1^2 + w2^2) ;  A, B, C = 11-term series  sum_i (-1)^i theta^(2i) / d_i  (left to right)
//     W = skew(w) ; W2 = W W ;  R = (I + A W) + B W2 ;  V = (I + B W) + C W2 ;  t = V u                 (se3_to_SE3)
//     in_cam = [x, y, 1] K^-T ;  c2w = [R^T | -(R^T t)] ;  center = c2w[:, 3] ;  ray = ([in_cam, 1] c2w^T) - center
#include "ls2fm_device.h"

namespace {

constexpr int kCrThreads = 256;
constexpr int kTerms = 11;

struct Kinv { float k[9]; };

__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
    return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
}

// The three 11-term series A, B, C (and their term-by-term derivatives) of one view, by ONE WAVE: lane 11 s + i forms term i of
// series s -- 33 powf calls side by side instead of one after the other in a single lane (15 us per call of the exponential, in
// front of every ray pick of a BA iteration) -- and every lane then sums the terms in the reference's order, left to right.
// Denominators by the reference's running products (utils/camera.py:100-147): A (2i+1)!, B (2i+2)!, C (2i+3)!.  All 64 lanes of
// the wave must be here.
template <bool WANT_D>
__device__ __forceinline__ void se3_series(float theta, float (&abc)[3], float (&dabc)[3]) {
    const int lane = threadIdx.x & 63;
    const int s = lane / kTerms, i = lane % kTerms;                 // lanes >= 33: s = 3.., never read
    double denom = s == 0 ? 1.0 : (s == 1 ? 2.0 : 6.0);
    for (int q = 1; q <= i; ++q)
        denom *= s == 0 ? (double)((2 * q) * (2 * q + 1)) : (s == 1 ? (double)((2 * q + 1) * (2 * q + 2)) : (double)((2 * q + 2) * (2 * q + 3)));
    const float sign = (i & 1) ? -1.0f : 1.0f;
    const float term = (sign * powf(theta, (float)(2 * i))) / (float)denom;
    float dterm = 0.f;
    if (WANT_D && i > 0) dterm = (sign * ((float)(2 * i) * powf(theta, (float)(2 * i - 1)))) / (float)denom;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float total = __shfl(term, q * kTerms, 64), dtotal = 0.f;
#pragma unroll
        for (int k = 1; k < kTerms; ++k) {
            total = total + __shfl(term, q * kTerms + k, 64);
            if (WANT_D) dtotal += __shfl(dterm, q * kTerms + k, 64);
        }
        abc[q] = total;
        dabc[q] = dtotal;
    }
}

// pose [3][4] of one view from its se(3) parameters (w, u): a whole wave calls this, every lane ends with the same pose
__device__ void se3_exp(const float* __restrict__ wu, float pose[12]) {
    const float w0 = wu[0], w1 = wu[1], w2 = wu[2];
    const float theta = sqrtf(w0 * w0 + w1 * w1 + w2 * w2);
    float abc[3], unused[3];
    se3_series<false>(theta, abc, unused);
    const float A = abc[0], B = abc[1], C = abc[2];
    const float W[3][3] = {{0.f, -w2, w1}, {w2, 0.f, -w0}, {-w1, w0, 0.f}};
    float W2[3][3], R[3][3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) W2[i][j] = dot3(W[i][0], W[0][j], W[i][1], W[1][j], W[i][2], W[2][j]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float eye = i == j ? 1.0f : 0.0f;
            R[i][j] = (eye + A * W[i][j]) + B * W2[i][j];
            V[i][j] = (eye + B * W[i][j]) + C * W2[i][j];
        }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        pose[4 * i + 0] = R[i][0]; pose[4 * i + 1] = R[i][1]; pose[4 * i + 2] = R[i][2];
        pose[4 * i + 3] = dot3(V[i][0], wu[3], V[i][1], wu[4], V[i][2], wu[5]);
    }
}

// d L / d (w, u) from d L / d pose [3][4]: the chain rule through  t = V u,  R = I + A W + B W2,  V = I + B W + C W2,  W2 = W W,
// W = skew(w),  theta = |w|  and the series' term-by-term derivatives  X'(theta) = sum_i (-1)^i 2i theta^(2i-1) / d_i  -- what
// autograd forms from the truncated series (the i = 0 terms are constants; |w| at 0 has the zero subgradient, as in torch).
// A whole wave calls this (se3_series); every lane computes the same d_wu, the caller lets one lane store it.
__device__ void se3_exp_bwd(const float* __restrict__ wu, const float* __restrict__ g, float* __restrict__ d_wu) {
    const float w[3] = {wu[0], wu[1], wu[2]}, u[3] = {wu[3], wu[4], wu[5]};
    const float theta = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    float abc[3], dabc[3];
    se3_series<true>(theta, abc, dabc);
    const float A = abc[0], B = abc[1], C = abc[2];
    const float W[3][3] = {{0.f, -w[2], w[1]}, {w[2], 0.f, -w[0]}, {-w[1], w[0], 0.f}};
    float W2[3][3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) W2[i][j] = dot3(W[i][0], W[0][j], W[i][1], W[1][j], W[i][2], W[2][j]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = ((i == j ? 1.0f : 0.0f) + B * W[i][j]) + C * W2[i][j];
    float gR[3][3], gV[3][3], gT[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        gT[i] = g[4 * i + 3];
#pragma unroll
        for (int j = 0; j < 3; ++j) { gR[i][j] = g[4 * i + j]; gV[i][j] = gT[i] * u[j]; }
    }
    float gA = 0.f, gB = 0.f, gC = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            gA = fmaf(gR[i][j], W[i][j], gA);
            gB = fmaf(gR[i][j], W2[i][j], fmaf(gV[i][j], W[i][j], gB));
            gC = fmaf(gV[i][j], W2[i][j], gC);
        }
    float gW[3][3], gW2[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) { gW[i][j] = fmaf(B, gV[i][j], A * gR[i][j]); gW2[i][j] = fmaf(C, gV[i][j], B * gR[i][j]); }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            // W2 = W W :  G_W += G_W2 W^T + W^T G_W2
            const float a = dot3(gW2[i][0], W[j][0], gW2[i][1], W[j][1], gW2[i][2], W[j][2]);
            const float b = dot3(W[0][i], gW2[0][j], W[1][i], gW2[1][j], W[2][i], gW2[2][j]);
            gW[i][j] += a + b;
        }
    const float g_theta = fmaf(gC, dabc[2], fmaf(gB, dabc[1], gA * dabc[0]));
    const float inv = theta > 0.f ? g_theta / theta : 0.f;
    d_wu[0] = fmaf(inv, w[0], gW[2][1] - gW[1][2]);
    d_wu[1] = fmaf(inv, w[1], gW[0][2] - gW[2][0]);
    d_wu[2] = fmaf(inv, w[2], gW[1][0] - gW[0][1]);
#pragma unroll
    for (int j = 0; j < 3; ++j) d_wu[3 + j] = dot3(V[0][j], gT[0], V[1][j], gT[1], V[2][j], gT[2]);
}

// one wave (= one 64-thread workgroup) per view
__global__ void __launch_bounds__(64) se3_exp_fwd_kernel(const float* __restrict__ se3, int n, float* __restrict__ poses) {
    const int v = blockIdx.x;
    float pose[12];
    se3_exp(se3 + 6 * v, pose);
    if (threadIdx.x == 0)
        for (int q = 0; q < 12; ++q) poses[12 * v + q] = pose[q];
}

__global__ void __launch_bounds__(64)
se3_exp_bwd_kernel(const float* __restrict__ se3, const float* __restrict__ d_poses, int n, float* __restrict__ d_se3) {
    const int v = blockIdx.x;
    float d[6];
    se3_exp_bwd(se3 + 6 * v, d_poses + 12 * v, d);
    if (threadIdx.x == 0)
        for (int q = 0; q < 6; ++q) d_se3[6 * v + q] = d[q];
}

// grid: (blocks over the pixels, views).  view_sel != null: ONE view, chosen on the device (blockIdx.y == 0 only), whose
// pixels are xy[view] when xy_per_view
__global__ void __launch_bounds__(kCrThreads)
camera_rays_kernel(const float* __restrict__ poses, const float* __restrict__ se3, Kinv K, const float* __restrict__ xy,
                   const int64_t* __restrict__ pix, int width, int xy_per_view, const int64_t* __restrict__ view_sel, int64_t n,
                   float* __restrict__ centers, float* __restrict__ rays, float* __restrict__ poses_out) {
    __shared__ float s_pose[12];
    const int out_v = (int)blockIdx.y;
    const int v = view_sel ? (int)view_sel[0] : out_v;
    if (threadIdx.x < 64) {                       // the first wave: the exponential is a wave-wide job (se3_series)
        float pose[12];
        if (se3) se3_exp(se3 + 6 * v, pose);
        else
            for (int q = 0; q < 12; ++q) pose[q] = poses[12 * v + q];
        if (threadIdx.x == 0) {
            for (int q = 0; q < 12; ++q) s_pose[q] = pose[q];
            if (poses_out && blockIdx.x == 0)
                for (int q = 0; q < 12; ++q) poses_out[12 * out_v + q] = pose[q];
        }
    }
    __syncthreads();
    // camera-to-world [R^T | -(R^T t)]  (invert_pose: the product first, then the negation)
    float c2w[3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        c2w[a][0] = s_pose[0 + a]; c2w[a][1] = s_pose[4 + a]; c2w[a][2] = s_pose[8 + a];
        c2w[a][3] = -dot3(s_pose[0 + a], s_pose[3], s_pose[4 + a], s_pose[7], s_pose[8 + a], s_pose[11]);
    }
    const int64_t i = (int64_t)blockIdx.x * kCrThreads + threadIdx.x;
    if (i >= n) return;
    float x, y;
    if (pix) {                                   // pixel centres of the image grid (mesh_grid: x + 0.5, y + 0.5)
        const int64_t p = pix[i];
        x = (float)(p % width) + 0.5f;
        y = (float)(p / width) + 0.5f;
    } else {
        const float* q = xy + 2 * ((xy_per_view ? (int64_t)v * n : 0) + i);
        x = q[0]; y = q[1];
    }
    float cam[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) cam[a] = dot3(x, K.k[3 * a], y, K.k[3 * a + 1], 1.0f, K.k[3 * a + 2]);
    float* c = centers + 3 * ((int64_t)out_v * n + i);
    float* r = rays + 3 * ((int64_t)out_v * n + i);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // to_hom(0) @ c2w^T = 0 + 0 + 0 + 1 c2w[a][3] ; to_hom(in_cam) @ c2w^T is the four-term chain ending in + 1 c2w[a][3]
        const float ctr = fmaf(1.0f, c2w[a][3], fmaf(0.0f, c2w[a][2], fmaf(0.0f, c2w[a][1], 0.0f * c2w[a][0])));
        const float world = fmaf(1.0f, c2w[a][3], fmaf(cam[2], c2w[a][2], fmaf(cam[1], c2w[a][1], cam[0] * c2w[a][0])));
        c[a] = ctr;
        r[a] = world - ctr;
    }
}

}  // namespace

extern "C" int ls2fm_se3_exp_fwd(const float* se3, int32_t n, float* poses, void* stream) {
    LS2FM_CHECK_ARG(n >= 0 && (n == 0 || (se3 && poses)));
    if (n == 0) return LS2FM_OK;
    se3_exp_fwd_kernel<<<n, 64, 0, (hipStream_t)stream>>>(se3, n, poses);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_se3_exp_bwd(const float* se3, const float* d_poses, int32_t n, float* d_se3, void* stream) {
    LS2FM_CHECK_ARG(n >= 0 && (n == 0 || (se3 && d_poses && d_se3)));
    if (n == 0) return LS2FM_OK;
    se3_exp_bwd_kernel<<<n, 64, 0, (hipStream_t)stream>>>(se3, d_poses, n, d_se3);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_camera_rays(const float* poses, const float* se3, const float* kinv_host, const float* xy, const int64_t* pix,
                                 int32_t width, int32_t xy_per_view, const int64_t* view_sel, int32_t n_views, int64_t n,
                                 float* centers, float* rays, float* poses_out, void* stream) {
    LS2FM_CHECK_ARG((poses != nullptr) != (se3 != nullptr));
    LS2FM_CHECK_ARG(kinv_host && ((xy != nullptr) != (pix != nullptr)) && centers && rays && n_views >= 1 && n >= 0);
    LS2FM_CHECK_ARG(!pix || width >= 1);
    if (n == 0) return LS2FM_OK;
    Kinv K;
    for (int q = 0; q < 9; ++q) K.k[q] = kinv_host[q];
    const dim3 grid((unsigned)((n + kCrThreads - 1) / kCrThreads), view_sel ? 1u : (unsigned)n_views);
    camera_rays_kernel<<<grid, kCrThreads, 0, (hipStream_t)stream>>>(poses, se3, K, xy, pix, width, xy_per_view, view_sel, n, centers, rays,
                                                                    poses_out);
    return ls2fm_launch_status();
}
