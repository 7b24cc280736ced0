// Loss head over the renderer's outputs: the scalar terms the reference's stages form right after Renderer.forward
// (rgb L1: pipelines/Camera.py:535; eikonal L1 on ||normals||: Initialization.py:257-258, BA.py:193-194; smooth-L1 depth
// consistency between the sphere-traced and the rendered depth over mask_finish: Camera.py:522-523; masked MSE for PSNR:
// Camera.py:533) and their 10^w weighted sum (BA.py:206-218).  In eager PyTorch this is ~35 launch-bound kernels between
// the fused forward and the fused backward; here it is one kernel each way.
//
// Deterministic: every workgroup writes fp64 partial sums, the last one to finish (ticket) adds them in index order.
#include "bin_items.h"

namespace {

constexpr int kLossThreads = 256;
constexpr int kLossMaxBlocks = 512;
constexpr int kSums = 8;      // S|rgb-gt|, n_rgb, S| |n|-1 |, n_eik, S sl1, n_dc, S (rgb-gt)^2 [mask_mse], n_mse

struct LossIn {
    const float* rgb; const float* rgb_gt;             // [R,3]
    const float* normals;                              // [R,N,3]
    const float* depth; const float* depth_ref;        // [R]  (depth_ref may be null: no DC term)
    const uint8_t* mask_eik; const uint8_t* mask_dc; const uint8_t* mask_mse;   // [R] or null (= all rays)
    int64_t n_rays; int32_t n_samples;
};

__device__ __forceinline__ float smooth_l1(float d) { const float a = fabsf(d); return a < 1.0f ? 0.5f * d * d : a - 0.5f; }
__device__ __forceinline__ float smooth_l1_grad(float d) { return fabsf(d) < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f); }
__device__ __forceinline__ float sign_of(float d) { return d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.f); }

__device__ __forceinline__ void finish_terms(const double* s, const float* w, float* terms, double* sums_out) {
    // means exactly as torch forms them: sum / count (0/0 = NaN for an empty eikonal mask, like l1_loss of an empty tensor;
    // the DC term is 0 for an empty mask: Camera.py:521, :531-532)
    const float rgb = (float)(s[0] / s[1]);
    const float eik = (float)(s[2] / s[3]);
    const float dc = s[5] > 0.0 ? (float)(s[4] / s[5]) : 0.f;
    const float mse = (float)(s[6] / s[7]);
    terms[0] = rgb; terms[1] = eik; terms[2] = dc; terms[3] = mse;
    terms[4] = fmaf(w[2], dc, fmaf(w[1], eik, w[0] * rgb));
    terms[5] = terms[4];
    terms[6] = -10.0f * log10f(mse);              // PSNR (Camera.py:534): ready with the terms, no torch op between forward and backward
    terms[7] = 0.f;
    for (int k = 0; k < kSums; ++k) sums_out[k] = s[k];
}

__global__ void __launch_bounds__(kLossThreads)
loss_head_fwd_kernel(LossIn in, const float* __restrict__ weights, double* __restrict__ partial, unsigned* __restrict__ ticket,
                     double* __restrict__ sums_out, float* __restrict__ terms) {
    __shared__ double red[kLossThreads / 64][kSums];
    __shared__ bool last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n_points = in.n_rays * in.n_samples;
    const int64_t stride = (int64_t)gridDim.x * kLossThreads;
    double s[kSums];
#pragma unroll
    for (int k = 0; k < kSums; ++k) s[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * kLossThreads + tid; i < n_points; i += stride) {
        const int64_t r = i / in.n_samples;
        if (in.mask_eik && !in.mask_eik[r]) continue;
        const float nx = in.normals[3 * i], ny = in.normals[3 * i + 1], nz = in.normals[3 * i + 2];
        const float len = sqrtf(nx * nx + ny * ny + nz * nz);
        s[2] += (double)fabsf(len - 1.0f);
        s[3] += 1.0;
    }
    for (int64_t r = (int64_t)blockIdx.x * kLossThreads + tid; r < in.n_rays; r += stride) {
        float d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) d[c] = in.rgb[3 * r + c] - in.rgb_gt[3 * r + c];
        s[0] += (double)fabsf(d[0]) + (double)fabsf(d[1]) + (double)fabsf(d[2]);
        s[1] += 3.0;
        if (!in.mask_mse || in.mask_mse[r]) {
            s[6] += (double)(d[0] * d[0]) + (double)(d[1] * d[1]) + (double)(d[2] * d[2]);
            s[7] += 3.0;
        }
        if (in.depth_ref && (!in.mask_dc || in.mask_dc[r])) {
            s[4] += (double)smooth_l1(in.depth_ref[r] - in.depth[r]);
            s[5] += 1.0;
        }
    }
#pragma unroll
    for (int k = 0; k < kSums; ++k) {
        double v = s[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid < kSums) {
        double v = 0.0;
        for (int q = 0; q < kLossThreads / 64; ++q) v += red[q][tid];
        partial[(int64_t)blockIdx.x * kSums + tid] = v;
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    // fixed-order parallel sum of the per-workgroup partials: thread t adds blocks t, t+32, ... of sum (t / 32 ... ) --
    // 32 lanes per sum, then a fixed xor tree: same association every run
    {
        const int k = tid >> 5, j = tid & 31;                // 8 sums x 32 lanes = 256 threads
        double v = 0.0;
        for (unsigned b = j; b < gridDim.x; b += 32) v += __builtin_nontemporal_load(&partial[(int64_t)b * kSums + k]);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (j == 0) red[0][k] = v;
    }
    __syncthreads();
    if (tid == 0) {
        finish_terms(red[0], weights, terms, sums_out);
        *ticket = 0u;                                   // re-armed for the next call on this workspace
    }
}

// upstream: g[5] = d/d(terms) (may be null = zeros); the weighted total's gradient is folded into the three
// differentiable terms
__global__ void __launch_bounds__(kLossThreads)
loss_head_bwd_kernel(LossIn in, const float* __restrict__ weights, const double* __restrict__ sums, const float* __restrict__ g,
                     const float* __restrict__ g_total, float* __restrict__ d_rgb, float* __restrict__ d_normals, float* __restrict__ d_depth,
                     float* __restrict__ d_depth_ref) {
    const int64_t n_points = in.n_rays * in.n_samples;
    const int64_t i = (int64_t)blockIdx.x * kLossThreads + threadIdx.x;
    float gt[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (g)
        for (int k = 0; k < 5; ++k) gt[k] = g[k];
    const float g_all = gt[4] + (g_total ? g_total[0] : 0.f);
    const float g_rgb = fmaf(weights[0], g_all, gt[0]) / (float)sums[1];
    const float g_eik = fmaf(weights[1], g_all, gt[1]) / (float)sums[3];
    const float g_dc = sums[5] > 0.0 ? fmaf(weights[2], g_all, gt[2]) / (float)sums[5] : 0.f;
    const float g_mse = gt[3] / (float)sums[7];
    if (i < n_points) {
        const int64_t r = i / in.n_samples;
        float o[3] = {0.f, 0.f, 0.f};
        if (!in.mask_eik || in.mask_eik[r]) {
            const float nx = in.normals[3 * i], ny = in.normals[3 * i + 1], nz = in.normals[3 * i + 2];
            const float len = sqrtf(nx * nx + ny * ny + nz * nz);
            const float k = len > 0.f ? g_eik * sign_of(len - 1.0f) / len : 0.f;     // d||n|| at 0 := 0 (torch)
            o[0] = k * nx; o[1] = k * ny; o[2] = k * nz;
        }
        d_normals[3 * i] = o[0]; d_normals[3 * i + 1] = o[1]; d_normals[3 * i + 2] = o[2];
    }
    if (i < in.n_rays) {
        const bool m = !in.mask_mse || in.mask_mse[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = in.rgb[3 * i + c] - in.rgb_gt[3 * i + c];
            float v = g_rgb * sign_of(d);
            if (m) v = fmaf(g_mse * 2.0f, d, v);
            d_rgb[3 * i + c] = v;
        }
        float gd = 0.f;
        if (in.depth_ref && (!in.mask_dc || in.mask_dc[i])) gd = g_dc * smooth_l1_grad(in.depth_ref[i] - in.depth[i]);
        d_depth[i] = -gd;
        if (d_depth_ref) d_depth_ref[i] = gd;
    }
}

// The launch after shade_fwd: workgroup 0 finishes the fused loss head (ls2fm_render_opts.loss) -- shade_fwd left four partial
// sums per ray (lpart [4][r_pad]); they are added in fixed order in fp64, the counts come from the masks -- and the other
// workgroups scan the item counts of the backward's scatter (bin_items.h).  (Inside the shade_fwd launch those 128 short
// jobs cost 27 us: its 1024 ray workgroups fill the chip's 1024 slots exactly once, anything extra starts a second round.)
constexpr int kPostThreads = 512;
__global__ void __launch_bounds__(kPostThreads)
post_shade_kernel(ls2fm_loss_spec loss, const float* __restrict__ lpart, int64_t n_rays, int64_t r_pad, int n_samples, ScanJob scan) {
    __shared__ int arena[kScanArenaInts];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x > 0) {
        scan_job_run(scan, (int)blockIdx.x - 1, tid, kPostThreads, arena);
        return;
    }
    if (loss.rgb_gt == nullptr) return;
    double (*red)[kSums] = reinterpret_cast<double (*)[kSums]>(arena);         // [kPostThreads / 64][kSums]
    static_assert(sizeof(double) * (kPostThreads / 64) * kSums <= sizeof(int) * kScanArenaInts, "reduction buffer fits");
    double s[kSums];
#pragma unroll
    for (int k = 0; k < kSums; ++k) s[k] = 0.0;
    for (int64_t r = tid; r < n_rays; r += kPostThreads) {
        s[0] += (double)lpart[0 * r_pad + r];
        s[1] += 3.0;
        if (ls2fm_in_eik(loss, r)) { s[2] += (double)lpart[1 * r_pad + r]; s[3] += (double)n_samples; }
        if (loss.depth_ref != nullptr && (loss.mask_dc == nullptr || loss.mask_dc[r])) {      // lpart row 2: the ray's depth
            s[4] += (double)ls2fm_smooth_l1(loss.depth_ref[r] - lpart[2 * r_pad + r]);
            s[5] += 1.0;
        }
        if (ls2fm_in_mse(loss, r)) { s[6] += (double)lpart[3 * r_pad + r]; s[7] += 3.0; }
    }
#pragma unroll
    for (int k = 0; k < kSums; ++k) {
        double v = s[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid < kSums) {
        double v = 0.0;
        for (int q = 0; q < kPostThreads / 64; ++q) v += red[q][tid];
        if ((tid & 1) && loss.count_scale > 1u) v *= (double)loss.count_scale;       // counts of a uniform data-parallel run
        red[0][tid] = v;
    }
    __syncthreads();
    if (tid == 0) finish_terms(red[0], loss.weights, loss.terms, loss.sums);
}

__global__ void terms_from_sums_kernel(const double* __restrict__ sums, const float* __restrict__ weights, float* __restrict__ terms) {
    double s[kSums], keep[kSums];
    for (int k = 0; k < kSums; ++k) s[k] = sums[k];
    finish_terms(s, weights, terms, keep);
}

}  // namespace

// loss: null = no fused loss head; scan_grid: null = no backward follows (nothing to scan)
int ls2fm_launch_post_shade(const ls2fm_loss_spec* loss, const float* ray_part, int64_t n_rays, int n_samples,
                            const ls2fm_grid_desc* scan_grid, int64_t n_points, float* bins_ws, hipStream_t stream) {
    if (!loss && !scan_grid) return LS2FM_OK;
    ls2fm_loss_spec ls{};
    if (loss) ls = *loss;
    ScanJob scan{};
    int blocks = 1;
    if (scan_grid) {
        scan.bm = make_bin_meta(bins_ws, n_points);
        scan.n_levels = scan_grid->n_levels;
        blocks += scan_grid->n_levels * kScanJobsPerLevel;
    }
    post_shade_kernel<<<blocks, kPostThreads, 0, stream>>>(ls, ray_part, n_rays, (n_rays + 63) / 64 * 64, n_samples, scan);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_loss_terms_from_sums(const double* sums, const float* weights, float* terms, void* stream) {
    LS2FM_CHECK_ARG(sums && weights && terms);
    terms_from_sums_kernel<<<1, 1, 0, (hipStream_t)stream>>>(sums, weights, terms);
    return ls2fm_launch_status();
}

extern "C" int64_t ls2fm_loss_head_workspace_bytes(void) {
    return (int64_t)sizeof(double) * kSums * kLossMaxBlocks + 64;
}

static LossIn make_loss_in(const float* rgb, const float* rgb_gt, const float* normals, const float* depth,
                           const float* depth_ref, const uint8_t* mask_eik, const uint8_t* mask_dc, const uint8_t* mask_mse,
                           int64_t n_rays, int32_t n_samples) {
    return LossIn{rgb, rgb_gt, normals, depth, depth_ref, mask_eik, mask_dc, mask_mse, n_rays, n_samples};
}

// workspace: [ticket (64 B)] [partials fp64 x 8 x blocks]; the ticket must be zero on first use (ls2fm_loss_head_fwd
// re-arms it) -- the caller zero-fills the workspace once.  sums: fp64[8], per call (kept for the backward).
extern "C" int ls2fm_loss_head_fwd(const float* rgb, const float* rgb_gt, const float* normals, const float* depth,
                                   const float* depth_ref, const uint8_t* mask_eik, const uint8_t* mask_dc,
                                   const uint8_t* mask_mse, int64_t n_rays, int32_t n_samples, const float* weights,
                                   float* terms, double* sums, void* workspace, void* stream) {
    LS2FM_CHECK_ARG(rgb && rgb_gt && normals && depth && weights && terms && sums && n_rays > 0 && n_samples > 0);
    if (!workspace) return LS2FM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned* ticket = reinterpret_cast<unsigned*>(workspace);
    double* partial = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 64);
    const int64_t n_points = n_rays * n_samples;
    int64_t blocks = (n_points + kLossThreads * 8 - 1) / (kLossThreads * 8);      // 8 points per thread: measured optimum (4: 18.8 us, 8: 16.2, 16: 17.6)
    blocks = blocks < 1 ? 1 : (blocks > kLossMaxBlocks ? kLossMaxBlocks : blocks);
    ls2fm_prof_begin(LS2FM_PROF_LOSS_FWD, s);
    loss_head_fwd_kernel<<<(unsigned)blocks, kLossThreads, 0, s>>>(
        make_loss_in(rgb, rgb_gt, normals, depth, depth_ref, mask_eik, mask_dc, mask_mse, n_rays, n_samples), weights, partial,
        ticket, sums, terms);
    ls2fm_prof_end(LS2FM_PROF_LOSS_FWD, s);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_loss_head_bwd(const float* rgb, const float* rgb_gt, const float* normals, const float* depth,
                                   const float* depth_ref, const uint8_t* mask_eik, const uint8_t* mask_dc,
                                   const uint8_t* mask_mse, int64_t n_rays, int32_t n_samples, const float* weights,
                                   const float* d_terms, const float* d_total, float* d_rgb, float* d_normals, float* d_depth,
                                   float* d_depth_ref, const double* sums, void* stream) {
    LS2FM_CHECK_ARG(rgb && rgb_gt && normals && depth && weights && (d_terms || d_total) && d_rgb && d_normals && d_depth && sums &&
                    n_rays > 0 && n_samples > 0);
    hipStream_t s = (hipStream_t)stream;
    const int64_t n_points = n_rays * n_samples;
    ls2fm_prof_begin(LS2FM_PROF_LOSS_BWD, s);
    loss_head_bwd_kernel<<<(unsigned)((n_points + kLossThreads - 1) / kLossThreads), kLossThreads, 0, s>>>(
        make_loss_in(rgb, rgb_gt, normals, depth, depth_ref, mask_eik, mask_dc, mask_mse, n_rays, n_samples), weights, sums,
        d_terms, d_total, d_rgb, d_normals, d_depth, d_depth_ref);
    ls2fm_prof_end(LS2FM_PROF_LOSS_BWD, s);
    return ls2fm_launch_status();
}
