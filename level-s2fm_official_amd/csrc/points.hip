// Point queries of the SDF field WITH a graph: the backward of  p -> (sdf, 17 MLP outputs, analytic normal d sdf / d p)
// for M free points (M ~ 1e3 .. 1e5), to the hash table, the SDF MLP (in the reference's weight_v / weight_g / bias
// parametrisation) and the points themselves -- including the double-backward terms of the normal.
//
// Replaces what loss.backward() traverses for SDF.infer_sdf with parameters requiring grad, SDF.gradient (create_graph=True:
// callers put ||gradient|| inside losses) and SDF.get_surface_pts (models/SDF.py:55-114; callers pipelines/BA.py:123-125
// "sfm" mode, pipelines/Registration.py:202,259-261, 500-iteration loops): tcnn's backward and double-backward kernels, the
// torch Linear / Softplus / weight_norm backward nodes and autograd's double backward of them -- ~60 launch-bound kernels per
// call in the composed form.
//
// The forward is ls2fm_sdf_eval (sdf_eval.hip).  The backward keeps nothing from it: it re-encodes the points and then runs
// the render backward's own machinery on them (a point is a "ray" of one sample):
//   gather pass      ray_encode_kernel over free points: E, J channels, SDF-MLP weight prep, the scatter's item counts
//   points_bwd       thread per point: hidden layer, Softplus derivatives, DA = S1.T + S2.w1_0.Q, GJ = S1.w1_0, the encoding
//                    rows of W0^T DA / W0^T GJ (the scatter's records), the upstream vectors v / gf for the weight gradients
//   post             scans of the item counts
//   wgrad_mlp + reduce + finalize (side stream)    weight gradients on the matrix cores, weight-norm backward
//   scatter_fill + slab_accumulate                 table gradient, exact fixed-point sums, written in full
//   points_dx        (only when d p is wanted) first derivatives of the encodings contracted with DE and the mixed second
//                    partials contracted with RR and the normal's upstream (8 gathers per level again)
#include <cstdlib>

#include "bin_items.h"

int ls2fm_launch_points_encode(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                               const float* pts, const WsLayout& w, float* ws, hipStream_t s);
int ls2fm_launch_post_shade(const ls2fm_loss_spec* loss, const float* ray_part, int64_t n_rays, int n_samples,
                            const ls2fm_grid_desc* scan_grid, int64_t n_points, float* bins_ws, hipStream_t stream);
int ls2fm_launch_scatter_fill(const ls2fm_grid_desc* grid, const FieldC& fc, const float* center, const float* ray, float* bins_ws,
                              int64_t n_points, int64_t p_pad, const float* rec1, const float* rec2, const float* rpt,
                              const float* ray_bound, int64_t n_rays, int dual, hipStream_t stream, int level_lo = 0, int level_hi = -1,
                              int n_explicit = 0, const void* side_jobs = nullptr);
int ls2fm_launch_slab_accumulate(const ls2fm_grid_desc* grid, float* bins_ws, int64_t n_points, float* dtable1, float* dtable2,
                                 hipStream_t stream, int level_lo = 0, int level_hi = -1, int add_into = 0, int n_explicit = 0,
                                 int n_samples = 1);
int ls2fm_launch_finalize_sdf(const ls2fm_params* params, const ls2fm_param_grads* grads, int in_dim, const Packed* pk,
                              const float* wg, hipStream_t stream, int add = 0);
bool ls2fm_bins_levels_fit(const ls2fm_grid_desc* grid, int dual);

namespace {

// thread per point.  Weights as wave-uniform scalar loads from the packed records (Packed::sdf: [j][0..34] W0 row in the
// reference's column order (p, e), [35] b0, [36..52] W1[o][j]).
__global__ void __launch_bounds__(256)
points_bwd_kernel(FieldC fc, LevelScales lsc, int n_levels, WsLayout w, const Packed* __restrict__ pk, const float* __restrict__ pts,
                  const float* __restrict__ d_sdf, const float* __restrict__ d_feat, const float* __restrict__ d_normal,
                  float* __restrict__ ws, int want_dx, ZeroJob zero) {
    // leading workgroups: the zero fills the rest of the call needs (as in shade_bwd: no memset nodes in front of the chain)
    if ((int)blockIdx.x < zero.blocks) {
        zero_job_run(zero, (int)blockIdx.x, (int)threadIdx.x, 256);
        return;
    }
    const int64_t i = ((int64_t)blockIdx.x - zero.blocks) * 256 + threadIdx.x;
    if (i >= w.p) return;
    const int64_t P = w.p_pad;
    float p[3], x[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        p[a] = pts[i * 3 + a];
        x[a] = (p[a] - fc.bmin[a]) / (fc.bmax[a] - fc.bmin[a]);
    }
    // upstream of the 17 MLP outputs and of the normal
    float gf[kOut];
#pragma unroll
    for (int o = 0; o < kOut; ++o) gf[o] = d_feat ? d_feat[i * kOut + o] : 0.f;
    if (d_sdf) gf[0] = fmaf(fc.kappa, d_sdf[i], gf[0]);               // sdf = kappa f0
    float gnk[3], gns[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        gnk[a] = d_normal ? fc.kappa * d_normal[i * 3 + a] : 0.f;
        gns[a] = gnk[a] * fc.inv_ext[a];
    }
    // u = [p / rescale ; e(x)],  v = d u / d p contracted with kappa g_n   (reference column order)
    float u[kInMax], v[kInMax];
#pragma unroll
    for (int a = 0; a < 3; ++a) { u[a] = p[a] / fc.rescale; v[a] = gnk[a] / fc.rescale; }
#pragma unroll
    for (int c = 0; c < 2 * LS2FM_MAX_LEVELS; ++c) {
        float e = 0.f, acc = 0.f;
        if (c < 2 * n_levels) {
            e = ws[w.e1 + (int64_t)c * P + i];
#pragma unroll
            for (int a = 0; a < 3; ++a) acc = fmaf(ws[w.j1 + ((int64_t)c * P + i) * 3 + a], gns[a], acc);         // [channel][point][3]
        }
        u[3 + c] = e;
        v[3 + c] = acc;
    }
    float de[kInMax], rr[kInMax];
#pragma unroll
    for (int k = 0; k < kInMax; ++k) { de[k] = 0.f; rr[k] = 0.f; }
    const float* __restrict__ rec = pk->sdf;
#pragma unroll 1
    for (int j = 0; j < kHidden; ++j) {
        const float* __restrict__ wr = rec + j * kRecStride;
        float a0 = wr[kRecB0], a1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int k = 0; k + 1 < kInMax; k += 2) {
            a0 = fmaf(wr[k], u[k], a0);
            a1 = fmaf(wr[k + 1], u[k + 1], a1);
            q0 = fmaf(wr[k], v[k], q0);
            q1 = fmaf(wr[k + 1], v[k + 1], q1);
        }
        a0 = fmaf(wr[kInMax - 1], u[kInMax - 1], a0);
        q0 = fmaf(wr[kInMax - 1], v[kInMax - 1], q0);
        float h, s1, s2;
        softplus100(a0 + a1, h, s1, s2);
        const float q = q0 + q1;
        float t = 0.f;
#pragma unroll
        for (int o = 0; o < kOut; ++o) t = fmaf(wr[kRecW1 + o], gf[o], t);
        const float w10 = wr[kRecW1];
        const float da = fmaf(s1, t, s2 * w10 * q);
        const float gj = s1 * w10;
#pragma unroll
        for (int k = 0; k < kInMax; ++k) {
            de[k] = fmaf(wr[k], da, de[k]);
            rr[k] = fmaf(wr[k], gj, rr[k]);
        }
    }
    // ---- hand-over to the weight-gradient and scatter kernels of the render backward
#pragma unroll
    for (int k = 0; k < kInMax; ++k) ws[w.v + (int64_t)k * P + i] = v[k];
#pragma unroll
    for (int o = 0; o < kOut; ++o) ws[w.gf + (int64_t)o * P + i] = gf[o];
#pragma unroll
    for (int a = 0; a < 3; ++a) ws[w.p3 + (int64_t)a * P + i] = p[a];
    float4* rpt = reinterpret_cast<float4*>(ws + w.rpt + i * 8);
    rpt[0] = make_float4(x[0], x[1], x[2], gns[0]);
    rpt[1] = make_float4(gns[1], gns[2], 0.f, 0.f);
    const float g1 = fabsf(gns[0]) + fabsf(gns[1]) + fabsf(gns[2]);
#pragma unroll
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) {
        float b = 0.f;
        if (l < n_levels) {
            const float d0 = de[3 + 2 * l], d1 = de[4 + 2 * l], r0 = rr[3 + 2 * l], r1 = rr[4 + 2 * l];
            *reinterpret_cast<float4*>(ws + w.rec1 + ((int64_t)l * P + i) * 4) = make_float4(d0, d1, r0, r1);
            b = fmaxf(fabsf(d0), fabsf(d1)) + lsc.s[l] * g1 * fmaxf(fabsf(r0), fabsf(r1));
        }
        ws[w.smax + (int64_t)l * w.r_pad + i] = b;            // a point is its own "ray" (n_samples = 1)
    }
    if (want_dx) {
#pragma unroll
        for (int a = 0; a < 3; ++a) ws[w.dexyz + (int64_t)a * P + i] = de[a];
    }
}

// ---- the same, 16 lanes per point (up to 16 384 points -- alone it wins up to ~32 768 (44 against 54 us), but beside the render's
// backward (the stage step's traced-depth node: 21 504 points) the thread-per-point form measured better, 0.870 against 0.884 ms: the stage loops' point queries -- thread per point the 64-unit loop above
// is a 90 us latency chain in 84 workgroups).  Lane jl of a group reads the two channels of level jl, evaluates hidden units
// jl + 16 q, and in ONE pass over j = 0 .. 63 (unit j's DA and GJ broadcast by DPP) advances de[k], rr[k] for k = jl, 16 + jl,
// 32 + jl: every sum in the order of the thread-per-point kernel -- BIT-IDENTICAL records and rows.
constexpr int kPbW0 = 36;           // [j][k = 0 .. 34, b0]
constexpr int kPbW1 = kOut;         // [j][o]

template <int J>
__device__ __forceinline__ void points_chain(const float* __restrict__ w0, float da, float gj, int jl, float (&de)[3], float (&rr)[3]) {
    if constexpr (J < 16) {
        const float dav = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(da), 0x150 + J, 0xF, 0xF, false));
        const float gjv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(gj), 0x150 + J, 0xF, 0xF, false));
        const float* __restrict__ w0j = w0 + J * kPbW0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {                   // k = 32 + jl is an input only for jl < 3 (the others are never read)
            const float wk = w0j[16 * t + jl];
            de[t] = fmaf(wk, dav, de[t]);
            rr[t] = fmaf(wk, gjv, rr[t]);
        }
        points_chain<J + 1>(w0, da, gj, jl, de, rr);
    }
}

__global__ void __launch_bounds__(256)
points_bwd_wide_kernel(FieldC fc, LevelScales lsc, int n_levels, WsLayout w, const Packed* __restrict__ pk, const float* __restrict__ pts,
                       const float* __restrict__ d_sdf, const float* __restrict__ d_feat, const float* __restrict__ d_normal,
                       float* __restrict__ ws, int want_dx, ZeroJob zero) {
    if ((int)blockIdx.x < zero.blocks) {
        zero_job_run(zero, (int)blockIdx.x, (int)threadIdx.x, 256);
        return;
    }
    __shared__ float s_w0[kHidden * kPbW0 + 16];          // + 16: the k = 32 + jl reads of the last row stay inside
    __shared__ float s_w1[kHidden * kPbW1];
    __shared__ float s_o[16][96];                         // de[0 .. 47] | rr[0 .. 47] of the group's point
    for (int q = threadIdx.x; q < kHidden * kPbW0; q += 256) s_w0[q] = pk->sdf[(q / kPbW0) * kRecStride + q % kPbW0];
    for (int q = threadIdx.x; q < kHidden * kPbW1; q += 256) s_w1[q] = pk->sdf[(q / kPbW1) * kRecStride + kRecW1 + q % kPbW1];
    if (threadIdx.x < 16) s_w0[kHidden * kPbW0 + threadIdx.x] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, jl = lane & 15, gbase = lane & 48, grp = threadIdx.x >> 4;
    const int64_t i_raw = ((int64_t)blockIdx.x - zero.blocks) * 16 + grp;
    const bool live = i_raw < w.p;
    const int64_t i = live ? i_raw : w.p - 1;
    const int64_t P = w.p_pad;
    float p[3], x[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        p[a] = pts[i * 3 + a];
        x[a] = (p[a] - fc.bmin[a]) / (fc.bmax[a] - fc.bmin[a]);
    }
    float gf[kOut];
#pragma unroll
    for (int o = 0; o < kOut; ++o) gf[o] = d_feat ? d_feat[i * kOut + o] : 0.f;
    if (d_sdf) gf[0] = fmaf(fc.kappa, d_sdf[i], gf[0]);               // sdf = kappa f0
    float gnk[3], gns[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        gnk[a] = d_normal ? fc.kappa * d_normal[i * 3 + a] : 0.f;
        gns[a] = gnk[a] * fc.inv_ext[a];
    }
    // this lane's two channels (level jl): e and v = J . gns
    float e_me[2] = {0.f, 0.f}, v_me[2] = {0.f, 0.f};
    if (jl < n_levels) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int64_t c = 2 * jl + f;
            e_me[f] = ws[w.e1 + c * P + i];
            float acc = 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) acc = fmaf(ws[w.j1 + (c * P + i) * 3 + a], gns[a], acc);         // [channel][point][3]
            v_me[f] = acc;
        }
    }
    float u[kInMax], v[kInMax];
#pragma unroll
    for (int a = 0; a < 3; ++a) { u[a] = p[a] / fc.rescale; v[a] = gnk[a] / fc.rescale; }
#pragma unroll
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) {
        u[3 + 2 * l] = __shfl(e_me[0], gbase + l, 64);
        u[4 + 2 * l] = __shfl(e_me[1], gbase + l, 64);
        v[3 + 2 * l] = __shfl(v_me[0], gbase + l, 64);
        v[4 + 2 * l] = __shfl(v_me[1], gbase + l, 64);
    }
    float de[3] = {0.f, 0.f, 0.f}, rr[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {                   // (a run-time loop: see sdf_eval_wide_full_kernel)
        const int j = jl + 16 * q;
        const float* __restrict__ wr = s_w0 + j * kPbW0;
        float a0 = wr[kRecB0], a1 = 0.f, q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int k = 0; k + 1 < kInMax; k += 2) {
            a0 = fmaf(wr[k], u[k], a0);
            a1 = fmaf(wr[k + 1], u[k + 1], a1);
            q0 = fmaf(wr[k], v[k], q0);
            q1 = fmaf(wr[k + 1], v[k + 1], q1);
        }
        a0 = fmaf(wr[kInMax - 1], u[kInMax - 1], a0);
        q0 = fmaf(wr[kInMax - 1], v[kInMax - 1], q0);
        float h, s1, s2;
        softplus100(a0 + a1, h, s1, s2);
        const float qq = q0 + q1;
        const float* __restrict__ w1r = s_w1 + j * kPbW1;
        float t = 0.f;
#pragma unroll
        for (int o = 0; o < kOut; ++o) t = fmaf(w1r[o], gf[o], t);
        const float w10 = w1r[0];
        const float da = fmaf(s1, t, s2 * w10 * qq);
        const float gj = s1 * w10;
        points_chain<0>(s_w0 + 16 * q * kPbW0, da, gj, jl, de, rr);
    }
    // ---- hand-over to the weight-gradient and scatter kernels of the render backward
    if (live) {
        if (jl < 3) {
            const float va = jl == 0 ? v[0] : (jl == 1 ? v[1] : v[2]);
            const float pa = jl == 0 ? p[0] : (jl == 1 ? p[1] : p[2]);
            ws[w.v + (int64_t)jl * P + i] = va;
            ws[w.p3 + (int64_t)jl * P + i] = pa;
        }
        ws[w.v + (int64_t)(3 + 2 * jl) * P + i] = v_me[0];
        ws[w.v + (int64_t)(4 + 2 * jl) * P + i] = v_me[1];
        {
            float g_me = gf[0];                     // row jl of the 17 (a select chain: no run-time index into the register array)
#pragma unroll
            for (int o = 1; o < 16; ++o) g_me = jl == o ? gf[o] : g_me;
            ws[w.gf + (int64_t)jl * P + i] = g_me;
            if (jl == 0) ws[w.gf + (int64_t)16 * P + i] = gf[16];
        }
        if (jl == 0) {
            float4* rpt = reinterpret_cast<float4*>(ws + w.rpt + i * 8);
            rpt[0] = make_float4(x[0], x[1], x[2], gns[0]);
            rpt[1] = make_float4(gns[1], gns[2], 0.f, 0.f);
        }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) { s_o[grp][16 * t + jl] = de[t]; s_o[grp][48 + 16 * t + jl] = rr[t]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (live) {
        const float g1 = fabsf(gns[0]) + fabsf(gns[1]) + fabsf(gns[2]);
        float b = 0.f;                              // lane = level
        if (jl < n_levels) {
            const float d0 = s_o[grp][3 + 2 * jl], d1 = s_o[grp][4 + 2 * jl], r0 = s_o[grp][48 + 3 + 2 * jl], r1 = s_o[grp][48 + 4 + 2 * jl];
            *reinterpret_cast<float4*>(ws + w.rec1 + ((int64_t)jl * P + i) * 4) = make_float4(d0, d1, r0, r1);
            b = fmaxf(fabsf(d0), fabsf(d1)) + lsc.s[jl] * g1 * fmaxf(fabsf(r0), fabsf(r1));
        }
        ws[w.smax + (int64_t)jl * w.r_pad + i] = b;            // a point is its own "ray" (n_samples = 1)
        if (want_dx && jl < 3) ws[w.dexyz + (int64_t)jl * P + i] = s_o[grp][jl];
    }
}

// d L / d p  =  (1 / rescale) (W0^T DA)_p  +  inv_ext . sum_l sum_f [ de_f  d e_f / d x  +  rr_f (d^2 e_f / d x d x) gns ]
// (pose_grad.hip's per-sample expression, SDF grid only)
__global__ void __launch_bounds__(256)
points_dx_kernel(FieldC fc, LevelSet lv, WsLayout w, const float* __restrict__ table, const float* __restrict__ ws,
                 float* __restrict__ d_p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= w.p) return;
    const int64_t P = w.p_pad;
    const float4 pa = reinterpret_cast<const float4*>(ws + w.rpt)[2 * i];
    const float4 pb = reinterpret_cast<const float4*>(ws + w.rpt)[2 * i + 1];
    const float x[3] = {pa.x, pa.y, pa.z};
    const float gns[3] = {pa.w, pb.x, pb.y};
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
    for (int l = 0; l < lv.n_levels; ++l) {
        Cell c;
        locate(x, lv.scale[l], lv.res[l], lv.size[l], lv.offset[l], lv.hashed[l], c);
        float2 tv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) tv[k] = reinterpret_cast<const float2*>(table)[c.idx[k]];
        const float4 rec = reinterpret_cast<const float4*>(ws + w.rec1)[(int64_t)l * P + i];
        const float de[2] = {rec.x, rec.y}, rr[2] = {rec.z, rec.w};
        const float sc = lv.scale[l];
        float j0[3] = {0.f, 0.f, 0.f}, j1[3] = {0.f, 0.f, 0.f}, h0[3] = {0.f, 0.f, 0.f}, h1[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float dw = corner_dweight(c.w, k, a);
                j0[a] = fmaf(tv[k].x, dw, j0[a]);
                j1[a] = fmaf(tv[k].y, dw, j1[a]);
            }
            const float d01 = corner_d2weight(c.w, k, 0, 1), d02 = corner_d2weight(c.w, k, 0, 2), d12 = corner_d2weight(c.w, k, 1, 2);
            h0[0] = fmaf(tv[k].x, d01, h0[0]); h0[1] = fmaf(tv[k].x, d02, h0[1]); h0[2] = fmaf(tv[k].x, d12, h0[2]);
            h1[0] = fmaf(tv[k].y, d01, h1[0]); h1[1] = fmaf(tv[k].y, d02, h1[1]); h1[2] = fmaf(tv[k].y, d12, h1[2]);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] = fmaf(sc, de[0] * j0[a] + de[1] * j1[a], acc[a]);
        const float s2 = sc * sc;
        const float hg0[3] = {h0[0] * gns[1] + h0[1] * gns[2], h0[0] * gns[0] + h0[2] * gns[2], h0[1] * gns[0] + h0[2] * gns[1]};
        const float hg1[3] = {h1[0] * gns[1] + h1[1] * gns[2], h1[0] * gns[0] + h1[2] * gns[2], h1[1] * gns[0] + h1[2] * gns[1]};
#pragma unroll
        for (int a = 0; a < 3; ++a) acc[a] = fmaf(s2, rr[0] * hg0[a] + rr[1] * hg1[a], acc[a]);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) d_p[i * 3 + a] = fmaf(acc[a], fc.inv_ext[a], ws[w.dexyz + (int64_t)a * P + i] / fc.rescale);
}

// the same with 16 lanes per point (lane = level: ONE round of gathers instead of sixteen dependent ones); the sum over the levels
// runs through the lanes in level order -- bit-identical
__global__ void __launch_bounds__(256)
points_dx_wide_kernel(FieldC fc, LevelSet lv, WsLayout w, const float* __restrict__ table, const float* __restrict__ ws,
                      float* __restrict__ d_p) {
    const int lane = threadIdx.x & 63, jl = lane & 15, gbase = lane & 48;
    const int64_t i_raw = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = i_raw < w.p;
    const int64_t i = live ? i_raw : w.p - 1;
    const int64_t P = w.p_pad;
    const float4 pa = reinterpret_cast<const float4*>(ws + w.rpt)[2 * i];
    const float4 pb = reinterpret_cast<const float4*>(ws + w.rpt)[2 * i + 1];
    const float x[3] = {pa.x, pa.y, pa.z};
    const float gns[3] = {pa.w, pb.x, pb.y};
    float xl[3] = {0.f, 0.f, 0.f}, yl[3] = {0.f, 0.f, 0.f}, sc = 0.f;        // this level's two terms
    if (jl < lv.n_levels) {
        Cell c;
        sc = lv.scale[jl];
        locate(x, sc, lv.res[jl], lv.size[jl], lv.offset[jl], lv.hashed[jl], c);
        float2 tv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) tv[k] = reinterpret_cast<const float2*>(table)[c.idx[k]];
        const float4 rec = reinterpret_cast<const float4*>(ws + w.rec1)[(int64_t)jl * P + i];
        const float de[2] = {rec.x, rec.y}, rr[2] = {rec.z, rec.w};
        float j0[3] = {0.f, 0.f, 0.f}, j1[3] = {0.f, 0.f, 0.f}, h0[3] = {0.f, 0.f, 0.f}, h1[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float dw = corner_dweight(c.w, k, a);
                j0[a] = fmaf(tv[k].x, dw, j0[a]);
                j1[a] = fmaf(tv[k].y, dw, j1[a]);
            }
            const float d01 = corner_d2weight(c.w, k, 0, 1), d02 = corner_d2weight(c.w, k, 0, 2), d12 = corner_d2weight(c.w, k, 1, 2);
            h0[0] = fmaf(tv[k].x, d01, h0[0]); h0[1] = fmaf(tv[k].x, d02, h0[1]); h0[2] = fmaf(tv[k].x, d12, h0[2]);
            h1[0] = fmaf(tv[k].y, d01, h1[0]); h1[1] = fmaf(tv[k].y, d02, h1[1]); h1[2] = fmaf(tv[k].y, d12, h1[2]);
        }
        const float hg0[3] = {h0[0] * gns[1] + h0[1] * gns[2], h0[0] * gns[0] + h0[2] * gns[2], h0[1] * gns[0] + h0[2] * gns[1]};
        const float hg1[3] = {h1[0] * gns[1] + h1[1] * gns[2], h1[0] * gns[0] + h1[2] * gns[2], h1[1] * gns[0] + h1[2] * gns[1]};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            xl[a] = de[0] * j0[a] + de[1] * j1[a];
            yl[a] = rr[0] * hg0[a] + rr[1] * hg1[a];
        }
    }
    const float s2 = sc * sc;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int l = 0; l < lv.n_levels; ++l) {              // stage l: lane l's level continues the chain
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float t = fmaf(s2, yl[a], fmaf(sc, xl[a], acc[a]));
            acc[a] = __shfl(t, gbase + l, 64);
        }
    }
    if (live && jl < 3) {
        const float acc_a = jl == 0 ? acc[0] : (jl == 1 ? acc[1] : acc[2]);
        const float inv_a = jl == 0 ? fc.inv_ext[0] : (jl == 1 ? fc.inv_ext[1] : fc.inv_ext[2]);
        d_p[i * 3 + jl] = fmaf(acc_a, inv_a, ws[w.dexyz + (int64_t)jl * P + i] / fc.rescale);
    }
}

ls2fm_field_desc one_sample_field(const ls2fm_field_desc* field) {
    ls2fm_field_desc f = *field;
    f.n_samples = 1;
    f.dual_field = 0;
    return f;
}

}  // namespace

extern "C" int64_t ls2fm_sdf_points_workspace_bytes(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, int64_t n_points) {
    if (!field || !grid_desc_ok(grid) || n_points < 0) return LS2FM_ERR_INVALID_ARGUMENT;
    if (n_points > LS2FM_MAX_RENDER_POINTS) return LS2FM_ERR_UNSUPPORTED;
    const WsLayout w = make_ws_layout(n_points, 1, grid->n_levels, grid->n_levels, 0);
    return w.total * (int64_t)sizeof(float);
}

// ---- the stages of ls2fm_sdf_points_bwd, also driven by ls2fm_render_bwd for the backward of a traced depth (render_bwd.hip):
// front = gather pass over the points (+ weight prep), points_bwd, the scans of the item counts; scatter = the table gradient.
// zero_table: the table whose atomically flushed range the front's leading workgroups zero, or null (a caller that ADDS this
// call's sums into a table another producer owns).
int ls2fm_points_bwd_front(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params, const float* p,
                           int64_t n, const float* d_sdf, const float* d_feat, const float* d_normal, float* zero_table, bool want_dp,
                           void* workspace, hipStream_t s, WsLayout* w_out, hipEvent_t rows_ready) {
    const ls2fm_field_desc f1 = one_sample_field(field);
    const FieldC fc = make_field_c(&f1);
    const int L = grid->n_levels;
    const WsLayout w = make_ws_layout(n, 1, L, L, 0);
    float* ws = (float*)workspace;
    const Packed* pk = (const Packed*)(ws + w.packed);
    LevelScales lsc;
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) lsc.s[l] = l < L ? grid->scale[l] : 0.f;
    // zero fills (leading workgroups of points_bwd): reduced weight-gradient accumulators; the atomically flushed (point-split
    // coarse) levels of the table gradient
    ZeroJob zero{};
    zero.blocks = 16;
    zero.a = reinterpret_cast<float4*>(ws + w.wg);
    zero.na = (w.dbeta - w.wg) / 4;
    if (zero_table) {
        int64_t first = 0, count = 0;
        ls2fm_scatter_zero_range(grid, w.p, false, &first, &count);
        if (count > 0) { zero.b = reinterpret_cast<float4*>(zero_table + 2 * first); zero.nb = count / 2; }
    }
    ls2fm_prof_begin(LS2FM_PROF_ENCODE_SDF, s);
    int st = ls2fm_launch_points_encode(&f1, grid, params, p, w, ws, s);
    ls2fm_prof_end(LS2FM_PROF_ENCODE_SDF, s);
    if (st != LS2FM_OK) return st;
    ls2fm_prof_begin(LS2FM_PROF_SHADE_BWD, s);
    const char* force = getenv("LS2FM_POINTS_KERNEL");         // tests: 1 = thread per point, 2 = 16 lanes per point, whatever n
    const int forced = force ? atoi(force) : 0;
    if (forced == 2 || (forced != 1 && n <= 16384))          // latency-bound: 16 lanes per point (bit-identical)
        points_bwd_wide_kernel<<<(unsigned)((n + 15) / 16 + zero.blocks), 256, 0, s>>>(fc, lsc, L, w, pk, p, d_sdf, d_feat, d_normal, ws,
                                                                                       want_dp, zero);
    else
        points_bwd_kernel<<<(unsigned)((n + 255) / 256 + zero.blocks), 256, 0, s>>>(fc, lsc, L, w, pk, p, d_sdf, d_feat, d_normal, ws,
                                                                                    want_dp, zero);
    ls2fm_prof_end(LS2FM_PROF_SHADE_BWD, s);
    // the per-point rows a weight-gradient kernel contracts are final here (the scans below only serve the table scatter)
    if (rows_ready && hipEventRecord(rows_ready, s) != hipSuccess) return LS2FM_ERR_LAUNCH;
    ls2fm_prof_begin(LS2FM_PROF_BIN, s);
    st = ls2fm_launch_post_shade(nullptr, nullptr, n, 1, grid, w.p, ws + w.bins, s);
    ls2fm_prof_end(LS2FM_PROF_BIN, s);
    if (w_out) *w_out = w;
    return st;
}

// phase: 1 = sort the payloads (scatter_fill), 2 = sum them into `table` (slab_accumulate), 3 = both
int ls2fm_points_bwd_scatter(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, int64_t n, void* workspace, float* table,
                             int add_into, hipStream_t s, int phase) {
    const ls2fm_field_desc f1 = one_sample_field(field);
    const FieldC fc = make_field_c(&f1);
    const WsLayout w = make_ws_layout(n, 1, grid->n_levels, grid->n_levels, 0);
    float* ws = (float*)workspace;
    int st = LS2FM_OK;
    if (phase & 1) {
        ls2fm_prof_begin(LS2FM_PROF_SCATTER_RAD, s);
        st = ls2fm_launch_scatter_fill(grid, fc, nullptr, nullptr, ws + w.bins, w.p, w.p_pad, ws + w.rec1, nullptr, ws + w.rpt, ws + w.smax,
                                       n, 0, s);
        ls2fm_prof_end(LS2FM_PROF_SCATTER_RAD, s);
        if (st != LS2FM_OK) return st;
    }
    if (phase & 2) {
        ls2fm_prof_begin(LS2FM_PROF_SCATTER_SDF, s);
        st = ls2fm_launch_slab_accumulate(grid, ws + w.bins, w.p, table, nullptr, s, 0, -1, add_into);
        ls2fm_prof_end(LS2FM_PROF_SCATTER_SDF, s);
    }
    return st;
}

// add: the parameter gradients are ADDED to what `grads` holds (a previous producer of the same backward pass wrote them):
// the table through slab_accumulate's add mode (nothing zeroed, slabs without items untouched), the MLP tensors through the
// weight-norm backward's
static int points_bwd_impl(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                           const float* p, int64_t n, const float* d_sdf, const float* d_feat, const float* d_normal,
                           const ls2fm_param_grads* grads, float* d_p, void* workspace, void* stream, int add) {
    LS2FM_CHECK_ARG(field && grid_desc_ok(grid) && params && grads && n >= 0);
    LS2FM_CHECK_ARG(params->sdf_table && params->sdf_mlp[0].weight_v && params->sdf_mlp[1].weight_v);
    LS2FM_CHECK_ARG(d_sdf || d_feat || d_normal);
    if (field->bg_sdf) return LS2FM_ERR_UNSUPPORTED;            // min(sdf, bg_rad - |p|): general (composed) form only
    if (n > LS2FM_MAX_RENDER_POINTS || !ls2fm_bins_levels_fit(grid, 0)) return LS2FM_ERR_UNSUPPORTED;
    if (n == 0) return LS2FM_OK;
    LS2FM_CHECK_ARG(p && grads->sdf_table && grads->sdf_mlp[0].weight_v && grads->sdf_mlp[1].weight_v);
    LS2FM_CHECK_ARG((reinterpret_cast<uintptr_t>(grads->sdf_table) & 15u) == 0);
    if (!workspace) return LS2FM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    WsLayout w;
    int st = ls2fm_points_bwd_front(field, grid, params, p, n, d_sdf, d_feat, d_normal, add ? nullptr : grads->sdf_table, d_p != nullptr,
                                    workspace, s, &w, nullptr);
    if (st != LS2FM_OK) return st;
    const ls2fm_field_desc f1 = one_sample_field(field);
    const FieldC fc = make_field_c(&f1);
    const int L = grid->n_levels;
    float* ws = (float*)workspace;
    const Packed* pk = (const Packed*)(ws + w.packed);

    // fork: weight gradients (matrix cores) + weight-norm backward beside the table scatter
    SideCtx sc;
    const bool forked = ls2fm_side_stream(&sc, s) && hipEventRecord(sc.fork, s) == hipSuccess &&
                        hipStreamWaitEvent(sc.side, sc.fork, 0) == hipSuccess;
    hipStream_t gs = forked ? sc.side : s;
    ls2fm_launch_wgrad_mlp(fc, 0, 2 * L, 0, w, pk, nullptr, nullptr, n, ws, gs, /*sdf_only=*/true);
    ls2fm_prof_begin(LS2FM_PROF_FINALIZE, gs);
    st = ls2fm_launch_finalize_sdf(params, grads, 3 + 2 * L, pk, ws + w.wg, gs, add);
    ls2fm_prof_end(LS2FM_PROF_FINALIZE, gs);
    if (st == LS2FM_OK && forked && hipEventRecord(sc.join, sc.side) != hipSuccess) st = LS2FM_ERR_LAUNCH;
    if (st != LS2FM_OK) return ls2fm_join_on_error(forked, sc, s, st);

    st = ls2fm_points_bwd_scatter(field, grid, n, workspace, grads->sdf_table, add, s, 3);
    if (st != LS2FM_OK) return ls2fm_join_on_error(forked, sc, s, st);
    if (d_p) {
        ls2fm_prof_begin(LS2FM_PROF_POSE, s);
        const char* force = getenv("LS2FM_POINTS_KERNEL");         // tests: 1 = thread per point, 2 = 16 lanes per point
        const int forced = force ? atoi(force) : 0;
        if (forced == 2 || (forced != 1 && n <= 16384))
            points_dx_wide_kernel<<<(unsigned)((n + 15) / 16), 256, 0, s>>>(fc, make_level_set(grid), w, params->sdf_table, ws, d_p);
        else
            points_dx_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(fc, make_level_set(grid), w, params->sdf_table, ws, d_p);
        ls2fm_prof_end(LS2FM_PROF_POSE, s);
    }
    if (forked && hipStreamWaitEvent(s, sc.join, 0) != hipSuccess) return LS2FM_ERR_LAUNCH;       // join
    return ls2fm_launch_status();
}

extern "C" int ls2fm_sdf_points_bwd(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                                    const float* p, int64_t n, const float* d_sdf, const float* d_feat, const float* d_normal,
                                    const ls2fm_param_grads* grads, float* d_p, void* workspace, void* stream) {
    return points_bwd_impl(field, grid, params, p, n, d_sdf, d_feat, d_normal, grads, d_p, workspace, stream, 0);
}

extern "C" int ls2fm_sdf_points_bwd_add(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                                        const float* p, int64_t n, const float* d_sdf, const float* d_feat, const float* d_normal,
                                        const ls2fm_param_grads* grads, float* d_p, void* workspace, void* stream) {
    return points_bwd_impl(field, grid, params, p, n, d_sdf, d_feat, d_normal, grads, d_p, workspace, stream, 1);
}
