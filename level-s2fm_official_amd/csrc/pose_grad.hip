// Gradients of the rendered outputs w.r.t. the camera rays (d L / d center, d L / d ray) -- only enqueued when the caller
// asks for them (pose refinement: pipelines/BA.py:153-154; rays also carry requires_grad in Initializer.run).
//
//   d L / d p  =  (1 / rescale) (W0'^T DA)_p  [both fields, from shade_bwd]           position inputs of the MLPs
//              +  inv_ext . sum_levels sum_f  de_f  d e_f / d x                         encodings, first order (both grids)
//              +  inv_ext . sum_levels sum_f  rr_f  (d^2 e_f / d x d x) gns             mixed partials: the normal path (A.4)
//              +  Wc[:, 0:3]^T dz                                                       decoder's direct position input
//   d L / d center = sum_n d L / d p_n          (near / far come out of the AABB op outside autograd: constants, SURVEY C-10)
//   d L / d ray    = sum_n t_n d L / d p_n  +  view-embedding columns  +  (d L / d |ray|) ray / |ray|
//
// workgroup = ray, thread = sample; the trilinear first and second derivatives need the table values again: 8 gathers per
// level and grid, like the forward (this kernel is gather-bound and runs only in pose-gradient mode).
#include "render_common.h"

namespace {

template <int MAXT>
__global__ void __launch_bounds__(MAXT)
pose_grad_kernel(FieldC fc, LevelSet lv1, LevelSet lv2, int dual, WsLayout w, const Packed* __restrict__ pk,
                 const float* __restrict__ table1, const float* __restrict__ table2, const float* __restrict__ center,
                 const float* __restrict__ ray, const float* __restrict__ ws, float* __restrict__ d_center,
                 float* __restrict__ d_ray) {
    __shared__ float s_red[MAXT / 64][6];
    const int N = fc.n_samples;
    const int64_t r = blockIdx.x;
    // (derived, not blockDim.x: the compiler read that from the dispatch packet -- host memory, shade_bwd.hip)
    const int n = threadIdx.x, lane = n & 63, wave = n >> 6, n_waves = ((N + 63) / 64 * 64) >> 6;
    const bool live = n < N;
    const int nn = live ? n : N - 1;
    const int64_t i = r * N + nn;
    const int64_t P = w.p_pad;
    const RayGeom gm = load_ray(fc, center, ray, r);
    const float t = sample_depth(gm, nn, N);
    float p[3], x[3];
    sample_position(fc, gm, t, p, x);

    float gp[3] = {0.f, 0.f, 0.f};
    if (live) {
        const float4 pa = reinterpret_cast<const float4*>(ws + w.rpt)[2 * i];
        const float4 pb = reinterpret_cast<const float4*>(ws + w.rpt)[2 * i + 1];
        const float gns[3] = {pa.w, pb.x, pb.y};
        float acc[3] = {0.f, 0.f, 0.f};              // d L / d x (grid-normalised coordinates)
#pragma unroll 1
        for (int g = 0; g < (dual ? 2 : 1); ++g) {
            const LevelSet& lv = g ? lv2 : lv1;
            const float* __restrict__ table = g ? table2 : table1;
#pragma unroll 1
            for (int l = 0; l < lv.n_levels; ++l) {
                Cell c;
                locate(x, lv.scale[l], lv.res[l], lv.size[l], lv.offset[l], lv.hashed[l], c);
                float2 tv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) tv[k] = reinterpret_cast<const float2*>(table)[c.idx[k]];
                float de[2], rr[2] = {0.f, 0.f};
                if (g == 0) {
                    const float4 rec = reinterpret_cast<const float4*>(ws + w.rec1)[(int64_t)l * P + i];
                    de[0] = rec.x; de[1] = rec.y; rr[0] = rec.z; rr[1] = rec.w;
                } else {
                    const float2 rec = reinterpret_cast<const float2*>(ws + w.rec2)[(int64_t)l * P + i];
                    de[0] = rec.x; de[1] = rec.y;
                }
                const float sc = lv.scale[l];
                // first derivatives  J_f[a] = scale sum_k T_k dW_k/dw_a ; mixed seconds  H_f[ab] = scale^2 sum_k T_k d2W_k
                float j0[3] = {0.f, 0.f, 0.f}, j1[3] = {0.f, 0.f, 0.f}, h0[3] = {0.f, 0.f, 0.f}, h1[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const float dw = corner_dweight(c.w, k, a);
                        j0[a] = fmaf(tv[k].x, dw, j0[a]);
                        j1[a] = fmaf(tv[k].y, dw, j1[a]);
                    }
                    if (g == 0) {
                        const float d01 = corner_d2weight(c.w, k, 0, 1), d02 = corner_d2weight(c.w, k, 0, 2),
                                    d12 = corner_d2weight(c.w, k, 1, 2);
                        h0[0] = fmaf(tv[k].x, d01, h0[0]); h0[1] = fmaf(tv[k].x, d02, h0[1]); h0[2] = fmaf(tv[k].x, d12, h0[2]);
                        h1[0] = fmaf(tv[k].y, d01, h1[0]); h1[1] = fmaf(tv[k].y, d02, h1[1]); h1[2] = fmaf(tv[k].y, d12, h1[2]);
                    }
                }
#pragma unroll
                for (int a = 0; a < 3; ++a) acc[a] = fmaf(sc, de[0] * j0[a] + de[1] * j1[a], acc[a]);
                if (g == 0) {
                    // (H gns)_a with H symmetric, zero diagonal: index 0 = (0,1), 1 = (0,2), 2 = (1,2)
                    const float s2 = sc * sc;
                    const float hg0[3] = {h0[0] * gns[1] + h0[1] * gns[2], h0[0] * gns[0] + h0[2] * gns[2], h0[1] * gns[0] + h0[2] * gns[1]};
                    const float hg1[3] = {h1[0] * gns[1] + h1[1] * gns[2], h1[0] * gns[0] + h1[2] * gns[2], h1[1] * gns[0] + h1[2] * gns[1]};
#pragma unroll
                    for (int a = 0; a < 3; ++a) acc[a] = fmaf(s2, rr[0] * hg0[a] + rr[1] * hg1[a], acc[a]);
                }
            }
        }
        float dz[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) dz[k] = ws[w.dz + k * P + i];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = ws[w.dexyz + a * P + i];
            if (dual) v += ws[w.dexyz + (3 + a) * P + i];
            v = v / fc.rescale;
            v = fmaf(acc[a], fc.inv_ext[a], v);
#pragma unroll
            for (int k = 0; k < 3; ++k) v = fmaf(pk->wc[k][a], dz[k], v);
            gp[a] = v;
        }
    }
    // ---- per-ray sums
    float red[6] = {gp[0], gp[1], gp[2], t * gp[0], t * gp[1], t * gp[2]};
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const float s = wave_sum(red[q]);
        if (lane == 0) s_red[wave][q] = s;
    }
    __syncthreads();
    if (n < 3) {
        const int a = n;
        float sc = 0.f, sr = 0.f;
        for (int q = 0; q < n_waves; ++q) { sc += s_red[q][a]; sr += s_red[q][3 + a]; }
        // view embedding [d, sin(f d), cos(f d)] (models/base.py:143-151) through the collapsed decoder columns 6..32
        const float da = a == 0 ? gm.d[0] : (a == 1 ? gm.d[1] : gm.d[2]);      // (selects: a dynamically indexed ray record becomes an LDS alloca indexed by the flat work-item id, read from the dispatch packet in HOST memory: shade_bwd.hip)
        for (int k = 0; k < 3; ++k) {
            const float gz = ws[w.dzr + k * w.r_pad + r];
            float dv = pk->wc[k][6 + a];
            for (int q = 0; q < 8; ++q) {
                const float f = (float)(1 << (q >> 1));
                const float arg = da * f;
                dv = fmaf((q & 1) ? -sinf(arg) * f : cosf(arg) * f, pk->wc[k][6 + 3 + 3 * q + a], dv);
            }
            sr = fmaf(gz, dv, sr);
        }
        // interval lengths: delta = (t_next - t) |ray|
        const float len = sqrtf(gm.d[0] * gm.d[0] + gm.d[1] * gm.d[1] + gm.d[2] * gm.d[2]);
        sr = fmaf(ws[w.dlen + r], da / len, sr);
        d_center[r * 3 + a] = sc;
        d_ray[r * 3 + a] = sr;
    }
}

}  // namespace

int ls2fm_launch_pose_grad(const FieldC& fc, const ls2fm_grid_desc* sdf_grid, const ls2fm_grid_desc* rad_grid, int dual,
                           const WsLayout& w, const Packed* pk, const ls2fm_params* params, const float* center,
                           const float* ray, int64_t n_rays, const float* ws, float* d_center, float* d_ray, hipStream_t s) {
    const int threads = (fc.n_samples + 63) / 64 * 64;
    const LevelSet lv1 = make_level_set(sdf_grid), lv2 = make_level_set(dual ? rad_grid : sdf_grid);
    if (threads <= 256)
        pose_grad_kernel<256><<<(unsigned)n_rays, threads, 0, s>>>(fc, lv1, lv2, dual, w, pk, params->sdf_table, params->rad_table,
                                                                    center, ray, ws, d_center, d_ray);
    else
        pose_grad_kernel<512><<<(unsigned)n_rays, threads, 0, s>>>(fc, lv1, lv2, dual, w, pk, params->sdf_table, params->rad_table,
                                                                    center, ray, ws, d_center, d_ray);
    return LS2FM_OK;
}
