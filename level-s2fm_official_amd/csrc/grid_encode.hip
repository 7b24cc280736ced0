// Multiresolution hash-grid encoding for gfx950: forward (+ Jacobian), backward (table scatter + dx)
// and double backward.  One thread per (point, level) -- 8 independent 8-byte gathers per thread, tiny
// register footprint, so a CU keeps its full 32 waves in flight to hide L2/MALL/HBM latency; blockIdx.y
// is the level so the blocks of one level run together and its table slice (<= 4 MiB for a 2^19-entry
// level: one XCD L2) stays cache resident.
//
// Replaces tinycudann's kernel_grid / kernel_grid_backward / kernel_grid_backward_input /
// kernel_grid_backward_input_backward_*  as used through models/base.py:17,37.
#include "ls2fm_device.h"

namespace {

constexpr int kBlock = 256;

struct Strides {          // output addressing: element (point i, channel c) lives at i*point + c*channel
    int64_t point;
    int64_t channel;
};

__device__ __forceinline__ float2 load_entry(const float* __restrict__ table, uint32_t entry) {
    return *reinterpret_cast<const float2*>(table + 2ull * entry);
}

template <bool WITH_JAC>
__global__ void __launch_bounds__(kBlock)
grid_encode_fwd_kernel(LevelSet lv, const float* __restrict__ x, const float* __restrict__ table, int64_t n,
                       float* __restrict__ y, Strides ys, float* __restrict__ jac, Strides js) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= n) return;
    const float xp[3] = {x[i * 3 + 0], x[i * 3 + 1], x[i * 3 + 2]};
    Cell c;
    locate(xp, lv.scale[l], lv.res[l], lv.size[l], lv.offset[l], lv.hashed[l], c);
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = load_entry(table, c.idx[k]);
    float y0 = 0.f, y1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float wt = corner_weight(c.w, k);
        y0 = fmaf(wt, v[k].x, y0);
        y1 = fmaf(wt, v[k].y, y1);
    }
    y[i * ys.point + (2 * l + 0) * ys.channel] = y0;
    y[i * ys.point + (2 * l + 1) * ys.channel] = y1;
    if (WITH_JAC) {
#pragma unroll
        for (int gd = 0; gd < 3; ++gd) {
            float g0 = 0.f, g1 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float dw = corner_dweight(c.w, k, gd);
                g0 = fmaf(dw, v[k].x, g0);
                g1 = fmaf(dw, v[k].y, g1);
            }
            jac[i * js.point + ((2 * l + 0) * 3 + gd) * js.channel] = lv.scale[l] * g0;
            jac[i * js.point + ((2 * l + 1) * 3 + gd) * js.channel] = lv.scale[l] * g1;
        }
    }
}

// dtable += w_corner * dy ; dx += sum_f dy_f * d y_f / d x
template <bool WANT_TABLE, bool WANT_DX>
__global__ void __launch_bounds__(kBlock)
grid_encode_bwd_kernel(LevelSet lv, const float* __restrict__ x, const float* __restrict__ table,
                       const float* __restrict__ dy, Strides ds, int64_t n, float* __restrict__ dtable,
                       float* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= n) return;
    const float xp[3] = {x[i * 3 + 0], x[i * 3 + 1], x[i * 3 + 2]};
    Cell c;
    locate(xp, lv.scale[l], lv.res[l], lv.size[l], lv.offset[l], lv.hashed[l], c);
    const float d0 = dy[i * ds.point + (2 * l + 0) * ds.channel];
    const float d1 = dy[i * ds.point + (2 * l + 1) * ds.channel];
    if (WANT_TABLE) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float wt = corner_weight(c.w, k);
            float* e = dtable + 2ull * c.idx[k];
            atomicAdd(e + 0, wt * d0);
            atomicAdd(e + 1, wt * d1);
        }
    }
    if (WANT_DX) {
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = load_entry(table, c.idx[k]);
#pragma unroll
        for (int gd = 0; gd < 3; ++gd) {
            float g = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) g = fmaf(corner_dweight(c.w, k, gd), fmaf(d0, v[k].x, d1 * v[k].y), g);
            atomicAdd(dx + i * 3 + gd, lv.scale[l] * g);
        }
    }
}

// Backward of (dy, table, x) -> dx given ddx = dL/d(dx):
//   d_dy_f   = sum_d J_fd ddx_d
//   dtable_c += dy_f * sum_d scale * dW_c/dw_d * ddx_d
//   dx2_e    = sum_f dy_f * sum_{d != e} scale^2 * (sum_c d2W_c/dw_d dw_e v_cf) * ddx_d
__global__ void __launch_bounds__(kBlock)
grid_encode_bwd_bwd_kernel(LevelSet lv, const float* __restrict__ x, const float* __restrict__ table,
                           const float* __restrict__ dy, const float* __restrict__ ddx, int64_t n,
                           float* __restrict__ d_dy, float* __restrict__ dtable, float* __restrict__ dx2) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= n) return;
    const int ch = 2 * lv.n_levels;
    const float xp[3] = {x[i * 3 + 0], x[i * 3 + 1], x[i * 3 + 2]};
    const float gx[3] = {ddx[i * 3 + 0], ddx[i * 3 + 1], ddx[i * 3 + 2]};
    Cell c;
    locate(xp, lv.scale[l], lv.res[l], lv.size[l], lv.offset[l], lv.hashed[l], c);
    const float s = lv.scale[l];
    const float d0 = dy[i * ch + 2 * l + 0];
    const float d1 = dy[i * ch + 2 * l + 1];
    float2 v[8];
    if (d_dy || dx2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = load_entry(table, c.idx[k]);
    }
    float o0 = 0.f, o1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float dirw = 0.f;       // sum_d dW_k/dw_d * ddx_d
#pragma unroll
        for (int gd = 0; gd < 3; ++gd) dirw = fmaf(corner_dweight(c.w, k, gd), gx[gd], dirw);
        dirw *= s;
        if (d_dy) {
            o0 = fmaf(dirw, v[k].x, o0);
            o1 = fmaf(dirw, v[k].y, o1);
        }
        if (dtable) {
            float* e = dtable + 2ull * c.idx[k];
            atomicAdd(e + 0, dirw * d0);
            atomicAdd(e + 1, dirw * d1);
        }
    }
    if (d_dy) {
        d_dy[i * ch + 2 * l + 0] = o0;
        d_dy[i * ch + 2 * l + 1] = o1;
    }
    if (dx2) {
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            float acc = 0.f;
#pragma unroll
            for (int gd = 0; gd < 3; ++gd) {
                if (gd == e) continue;
                float m = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    m = fmaf(corner_d2weight(c.w, k, gd, e), fmaf(d0, v[k].x, d1 * v[k].y), m);
                acc = fmaf(m, gx[gd], acc);
            }
            atomicAdd(dx2 + i * 3 + e, s * s * acc);
        }
    }
}

__global__ void __launch_bounds__(kBlock)
grid_indices_kernel(LevelSet lv, const float* __restrict__ x, int64_t n, uint32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int l = blockIdx.y;
    if (i >= n) return;
    const float xp[3] = {x[i * 3 + 0], x[i * 3 + 1], x[i * 3 + 2]};
    Cell c;
    locate(xp, lv.scale[l], lv.res[l], lv.size[l], 0u, lv.hashed[l], c);
#pragma unroll
    for (int k = 0; k < 8; ++k) out[(i * lv.n_levels + l) * 8 + k] = c.idx[k];
}

inline dim3 launch_grid(int64_t n, int levels) { return dim3((unsigned)((n + kBlock - 1) / kBlock), (unsigned)levels); }

}  // namespace

extern "C" int ls2fm_grid_encode_fwd(const ls2fm_grid_desc* grid, const float* x, const float* table, int64_t n,
                                     float* y, float* dy_dx, void* stream) {
    LS2FM_CHECK_ARG(grid_desc_ok(grid) && n >= 0 && (n == 0 || (x && table && y)));
    if (n == 0) return LS2FM_OK;
    const LevelSet lv = make_level_set(grid);
    const int ch = 2 * grid->n_levels;
    const Strides ys{ch, 1}, js{(int64_t)ch * 3, 1};
    hipStream_t s = (hipStream_t)stream;
    if (dy_dx)
        grid_encode_fwd_kernel<true><<<launch_grid(n, grid->n_levels), kBlock, 0, s>>>(lv, x, table, n, y, ys, dy_dx, js);
    else
        grid_encode_fwd_kernel<false><<<launch_grid(n, grid->n_levels), kBlock, 0, s>>>(lv, x, table, n, y, ys, nullptr, js);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_grid_encode_bwd(const ls2fm_grid_desc* grid, const float* x, const float* table,
                                     const float* dy, int64_t n, float* dtable, float* dx, void* stream) {
    LS2FM_CHECK_ARG(grid_desc_ok(grid) && n >= 0 && (n == 0 || (x && table && dy)));
    if (n == 0 || (!dtable && !dx)) return LS2FM_OK;
    const LevelSet lv = make_level_set(grid);
    const Strides ds{2 * grid->n_levels, 1};
    hipStream_t s = (hipStream_t)stream;
    if (dx && hipMemsetAsync(dx, 0, sizeof(float) * 3 * n, s) != hipSuccess) return LS2FM_ERR_LAUNCH;
    const dim3 g = launch_grid(n, grid->n_levels);
    if (dtable && dx)
        grid_encode_bwd_kernel<true, true><<<g, kBlock, 0, s>>>(lv, x, table, dy, ds, n, dtable, dx);
    else if (dtable)
        grid_encode_bwd_kernel<true, false><<<g, kBlock, 0, s>>>(lv, x, table, dy, ds, n, dtable, nullptr);
    else
        grid_encode_bwd_kernel<false, true><<<g, kBlock, 0, s>>>(lv, x, table, dy, ds, n, nullptr, dx);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_grid_encode_bwd_bwd(const ls2fm_grid_desc* grid, const float* x, const float* table,
                                         const float* dy, const float* ddx, int64_t n, float* d_dy, float* dtable,
                                         float* dx2, void* stream) {
    LS2FM_CHECK_ARG(grid_desc_ok(grid) && n >= 0 && (n == 0 || (x && table && dy && ddx)));
    if (n == 0 || (!d_dy && !dtable && !dx2)) return LS2FM_OK;
    const LevelSet lv = make_level_set(grid);
    hipStream_t s = (hipStream_t)stream;
    if (dx2 && hipMemsetAsync(dx2, 0, sizeof(float) * 3 * n, s) != hipSuccess) return LS2FM_ERR_LAUNCH;
    grid_encode_bwd_bwd_kernel<<<launch_grid(n, grid->n_levels), kBlock, 0, s>>>(lv, x, table, dy, ddx, n, d_dy,
                                                                               dtable, dx2);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_grid_indices(const ls2fm_grid_desc* grid, const float* x, int64_t n, uint32_t* out,
                                  void* stream) {
    LS2FM_CHECK_ARG(grid_desc_ok(grid) && n >= 0 && (n == 0 || (x && out)));
    if (n == 0) return LS2FM_OK;
    grid_indices_kernel<<<launch_grid(n, grid->n_levels), kBlock, 0, (hipStream_t)stream>>>(make_level_set(grid), x, n,
                                                                                          out);
    return ls2fm_launch_status();
}
