// Fused volumetric-rendering backward for gfx950, including the analytic double backward of the normal
// path.  Replaces what loss.backward() traverses for Renderer.forward:
// composite / sigma / radiance / Geometry MLPs / SDF.gradient's create_graph graph / tcnn backward kernels.
//
//   shade_bwd     (shade_bwd.hip) composite / sigma / decoder backward per sample, then the MLP backward chains as
//                 transposed f32 MFMA GEMMs; writes the scatter payload records and the small per-sample upstream vectors
//   bin build     (bin_scatter.hip) per-slab item lists, on the side stream under shade_bwd
//   wgrad_mlp     (wgrad_mlp.hip) weight gradients of both Geometry MLPs and of the decoder columns on the matrix cores,
//                 re-deriving the hidden-layer operands instead of reading them back from HBM; partials + fixed-order sum
//   slab scatter  (bin_scatter.hip) LDS-owned slabs of the table gradient, no table-wide global atomics; for the SDF
//                 grid the first-order (trilinear) and double-backward (derivative-weight) terms in one add
//   finalize      un-collapse the radiance chain, weight-norm backward, d beta
#include <cstdlib>

#include "render_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------- finalize
// weight-norm backward of a whole layer:  W = (g/||v||) v  ->  dg = <dW,v>/||v|| ; dv = (g/||v||) dW - g <dW,v>/||v||^3 v.
// Called by all 256 threads; 16 lanes cooperate on a row (coalesced accesses, 16-wide shuffle reduction).
__device__ void weight_norm_bwd_rows(const float* v, const float* g, const float* dw, int dw_ld, int rows, int n_in,
                                     float* dv, float* dg, int tid) {
    const int sub = tid & 15;
    for (int row0 = 0; row0 < rows; row0 += 16) {
        const int row = row0 + (tid >> 4);
        const bool on = row < rows;
        float ss = 0.f, dot = 0.f;
        if (on)
            for (int k = sub; k < n_in; k += 16) {
                const float x = v[row * n_in + k];
                ss = fmaf(x, x, ss);
                dot = fmaf(dw[row * dw_ld + k], x, dot);
            }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) { ss += __shfl_xor(ss, m, 16); dot += __shfl_xor(dot, m, 16); }
        if (on) {
            const float nrm = sqrtf(ss);
            const float s = g[row] / nrm;
            const float c = g[row] * dot / (nrm * nrm * nrm);
            for (int k = sub; k < n_in; k += 16) dv[row * n_in + k] = s * dw[row * dw_ld + k] - c * v[row * n_in + k];
            if (sub == 0) dg[row] = dot / nrm;
        }
    }
}

__global__ void __launch_bounds__(256)
finalize_kernel(ls2fm_params P, ls2fm_param_grads G, int in_dim, int in_dim2, int rad_in, int dual,
                const Packed* __restrict__ pk, const float* __restrict__ wg, const float* __restrict__ dbeta, int64_t n_rays) {
    __shared__ float s_dwc[3][68];
    __shared__ float s_dbc[4];
    __shared__ float s_dt1[3][64];
    __shared__ float s_row[64][68];       // effective-weight gradients of the layer being processed
    const int tid = threadIdx.x;

    // one workgroup per layer (7 independent tasks; a single workgroup doing all of them was a 58 us latency chain):
    // 0/1 SDF MLP layers, 2/3 second field's layers, 4..6 radiance layers (each re-derives the small shared terms)
    const int task = blockIdx.x;
    if (task < 4) {
        const int which = task >> 1;
        if (which && !dual) return;
        const float* dW0 = wg + (which ? WgLayout::dG0 : WgLayout::dW0);
        const float* dW1 = wg + (which ? WgLayout::dG1 : WgLayout::dW1);
        const ls2fm_linear* lin = which ? P.geo_mlp : P.sdf_mlp;
        const ls2fm_linear_grad* gl = which ? G.geo_mlp : G.sdf_mlp;
        const int ind = which ? in_dim2 : in_dim;
        if ((task & 1) == 0) {
            weight_norm_bwd_rows(lin[0].weight_v, lin[0].weight_g, dW0, 36, kHidden, ind, gl[0].weight_v, gl[0].weight_g, tid);
            if (tid < kHidden) gl[0].bias[tid] = dW0[tid * 36 + 35];
            return;
        }
        for (int idx = tid; idx < kOut * kHidden; idx += 256) {
            const int o = idx / kHidden, j = idx % kHidden;
            s_row[o][j] = dW1[o * 65 + j] + ((which == 0 && o == 0) ? wg[WgLayout::dW1r0 + j] : 0.f);
        }
        __syncthreads();
        weight_norm_bwd_rows(lin[1].weight_v, lin[1].weight_g, &s_row[0][0], 68, kOut, kHidden, gl[1].weight_v,
                             gl[1].weight_g, tid);
        if (tid < kOut) gl[1].bias[tid] = dW1[tid * 65 + 64];
        return;
    }

    // ---- radiance decoder: expand dWc (3 x rad_in) and back through Wc = R2 R1 R0, bc = T1 b0 + R2 b1 + b2
    for (int idx = tid; idx < 3 * 68; idx += 256) {
        const int c = idx / 68, k = idx % 68;
        const float* row = wg + WgLayout::dWc + c * 39;
        float val = 0.f;
        if (k < 6) val = row[k];
        else if (k < 33) val = wg[WgLayout::dWv + c * 27 + (k - 6)];
        else if (k < 49) val = row[6 + (k - 33)];
        else if (k < 65) val = dual ? row[22 + (k - 49)] : 0.f;
        s_dwc[c][k] = val;
    }
    if (tid < 3) s_dbc[tid] = wg[WgLayout::dWc + tid * 39 + 38];
    __syncthreads();
    for (int idx = tid; idx < 3 * 64; idx += 256) {       // dT1 = dWc R0^T + dbc b0^T
        const int c = idx / 64, j = idx % 64;
        float acc = s_dbc[c] * P.rad_mlp[0].bias[j];
        for (int k = 0; k < rad_in; ++k) acc = fmaf(s_dwc[c][k], pk->r0[j][k], acc);
        s_dt1[c][j] = acc;
    }
    if (task == 4) {
        for (int idx = tid; idx < 64 * 68; idx += 256) {      // dR0 = T1^T dWc
            const int j = idx / 68, k = idx % 68;
            float acc = 0.f;
            for (int c = 0; c < 3; ++c) acc = fmaf(pk->t1[c][j], s_dwc[c][k], acc);
            s_row[j][k] = acc;
        }
        __syncthreads();
        weight_norm_bwd_rows(P.rad_mlp[0].weight_v, P.rad_mlp[0].weight_g, &s_row[0][0], 68, 64, rad_in,
                             G.rad_mlp[0].weight_v, G.rad_mlp[0].weight_g, tid);
        if (tid < 64) {
            float acc = 0.f;
            for (int c = 0; c < 3; ++c) acc = fmaf(pk->t1[c][tid], s_dbc[c], acc);
            G.rad_mlp[0].bias[tid] = acc;
        }
        // beta = exp(beta_param * speed):  d/d beta_param = dL/dbeta * beta * speed ; dL/dbeta = fixed-order sum of the
        // per-ray partials of shade_bwd (fp64)
        {
            __shared__ double s_db[256];
            double acc = 0.0;
            for (int64_t r = tid; r < n_rays; r += 256) acc += reinterpret_cast<const double*>(dbeta)[r];
            s_db[tid] = acc;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if (tid < o) s_db[tid] += s_db[tid + o];
                __syncthreads();
            }
            if (tid == 0) G.beta[0] = (float)(s_db[0] * (double)pk->beta * (double)P.beta_speed);
        }
        return;
    }
    __syncthreads();          // s_dt1 complete
    if (task == 5) {
        for (int idx = tid; idx < 64 * 64; idx += 256) {      // dR1 = R2^T dT1
            const int m = idx / 64, j = idx % 64;
            float acc = 0.f;
            for (int c = 0; c < 3; ++c) acc = fmaf(pk->r2[c][m], s_dt1[c][j], acc);
            s_row[m][j] = acc;
        }
        __syncthreads();
        weight_norm_bwd_rows(P.rad_mlp[1].weight_v, P.rad_mlp[1].weight_g, &s_row[0][0], 68, 64, 64,
                             G.rad_mlp[1].weight_v, G.rad_mlp[1].weight_g, tid);
        if (tid < 64) {
            float acc = 0.f;
            for (int c = 0; c < 3; ++c) acc = fmaf(pk->r2[c][tid], s_dbc[c], acc);
            G.rad_mlp[1].bias[tid] = acc;
        }
        return;
    }
    for (int idx = tid; idx < 3 * 64; idx += 256) {       // dR2 = dT1 R1^T + dbc b1^T
        const int c = idx / 64, m = idx % 64;
        float acc = s_dbc[c] * P.rad_mlp[1].bias[m];
        for (int j = 0; j < 64; ++j) acc = fmaf(s_dt1[c][j], pk->r1[m][j], acc);
        s_row[c][m] = acc;
    }
    __syncthreads();
    weight_norm_bwd_rows(P.rad_mlp[2].weight_v, P.rad_mlp[2].weight_g, &s_row[0][0], 68, 3, 64,
                         G.rad_mlp[2].weight_v, G.rad_mlp[2].weight_g, tid);
    if (tid < 3) G.rad_mlp[2].bias[tid] = s_dbc[tid];
}

}  // namespace

// weight-norm backward of the SDF MLP alone (point queries, points.hip): tasks 0 and 1 of finalize_kernel
int ls2fm_launch_finalize_sdf(const ls2fm_params* params, const ls2fm_param_grads* grads, int in_dim, const Packed* pk,
                              const float* wg, hipStream_t stream) {
    finalize_kernel<<<2, 256, 0, stream>>>(*params, *grads, in_dim, 0, 0, 0, pk, wg, nullptr, 0);
    return ls2fm_launch_status();
}

// ------------------------------------------------------------------------------------------- C ABI
int ls2fm_launch_scatter_fill(const ls2fm_grid_desc* grid, const FieldC& fc, const float* center, const float* ray, float* bins_ws,
                              int64_t n_points, int64_t p_pad, const float* rec1, const float* rec2, const float* rpt,
                              const float* ray_bound, int64_t n_rays, int dual, hipStream_t stream, int level_lo = 0, int level_hi = -1);
int ls2fm_launch_slab_accumulate(const ls2fm_grid_desc* grid, float* bins_ws, int64_t n_points, float* dtable1, float* dtable2,
                                 hipStream_t stream, int level_lo = 0, int level_hi = -1);

extern "C" int ls2fm_render_bwd(const ls2fm_field_desc* field, const ls2fm_grid_desc* sdf_grid,
                                const ls2fm_grid_desc* rad_grid, const ls2fm_params* params, const float* center,
                                const float* ray, int64_t n_rays, const float* d_rgb, const float* d_sdfs_volume,
                                const float* d_normals, const float* d_depth_mlp, const float* d_normal_mlp,
                                const ls2fm_param_grads* grads, float* d_center, float* d_ray, void* workspace,
                                const ls2fm_render_opts* opts, void* stream) {
    LS2FM_CHECK_ARG(field && grid_desc_ok(sdf_grid) && params && grads && n_rays >= 0);
    LS2FM_CHECK_ARG(!field->dual_field || grid_desc_ok(rad_grid));
    if (field->bg_sdf) return LS2FM_ERR_UNSUPPORTED;
    if (field->dual_field && !same_grid_geometry(sdf_grid, rad_grid)) return LS2FM_ERR_UNSUPPORTED;
    if (field->n_samples < 1 || field->n_samples > 512) return LS2FM_ERR_UNSUPPORTED;
    LS2FM_CHECK_ARG((d_center == nullptr) == (d_ray == nullptr));     // pose gradients: both or neither
    const int want_pose = d_center != nullptr;
    const ls2fm_loss_spec* loss = opts ? opts->loss : nullptr;
    LS2FM_CHECK_ARG(!loss || (loss->rgb_gt && loss->weights && loss->sums && (loss->d_terms || loss->d_total)));
    if (n_rays == 0) return LS2FM_OK;
    LS2FM_CHECK_ARG(center && ray && grads->sdf_table && grads->beta && (!field->dual_field || grads->rad_table));
    if (!workspace) return LS2FM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int dual = field->dual_field ? 1 : 0;
    const int L1 = sdf_grid->n_levels, L2 = dual ? rad_grid->n_levels : 0;
    const WsLayout w = make_ws_layout(n_rays, field->n_samples, L1, dual ? L2 : L1, dual);
    float* ws = (float*)workspace;
    const Packed* pk = (const Packed*)(ws + w.packed);
    const FieldC fc = make_field_c(field);
    const int rad_in = 3 + 3 + kView + LS2FM_FEAT * (dual ? 2 : 1);
    const int64_t P = w.p_pad;

    LevelScales lsc;
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) lsc.s[l] = l < L1 ? sdf_grid->scale[l] : 0.f;

    SideCtx sc;
    bool forked = false;
    LS2FM_CHECK_ARG(((reinterpret_cast<uintptr_t>(grads->sdf_table) | (dual ? reinterpret_cast<uintptr_t>(grads->rad_table) : 0)) & 15u) == 0);

    Upstream up{d_rgb, d_sdfs_volume, d_normals, d_depth_mlp, d_normal_mlp, LossUp{}};
    if (loss)
        up.loss = LossUp{loss->rgb_gt, loss->depth_ref, loss->mask_eik, loss->mask_dc, loss->mask_mse, loss->weights, loss->sums,
                         loss->d_terms, loss->d_total, loss->d_depth_ref};
    ls2fm_prof_begin(LS2FM_PROF_SHADE_BWD, s);
    // (leading workgroups of this launch zero the weight-gradient accumulators and the atomically flushed table ranges)
    ls2fm_launch_shade_bwd(fc, lsc, dual, 2 * L1, 2 * L2, w, pk, center, ray, n_rays, ws, up, want_pose, sdf_grid, grads->sdf_table,
                           dual ? grads->rad_table : nullptr, s);
    ls2fm_prof_end(LS2FM_PROF_SHADE_BWD, s);
    if (want_pose) {
        ls2fm_prof_begin(LS2FM_PROF_POSE, s);
        ls2fm_launch_pose_grad(fc, sdf_grid, rad_grid, dual, w, pk, params, center, ray, n_rays, ws, d_center, d_ray, s);
        ls2fm_prof_end(LS2FM_PROF_POSE, s);
    }

    // the one fork of the backward: weight-gradient GEMMs -> reduce -> finalize run on the side stream, beside the table scatters
    forked = ls2fm_side_stream(&sc, s) && hipEventRecord(sc.fork, s) == hipSuccess && hipStreamWaitEvent(sc.side, sc.fork, 0) == hipSuccess;
    hipStream_t gs = forked ? sc.side : s;
    ls2fm_launch_wgrad_mlp(fc, dual, 2 * L1, 2 * L2, w, pk, center, ray, n_rays, ws, gs);
    ls2fm_prof_begin(LS2FM_PROF_FINALIZE, gs);
    finalize_kernel<<<7, 256, 0, gs>>>(*params, *grads, 3 + 2 * L1, 3 + 2 * L2, rad_in, dual, pk, ws + w.wg,
                                       ws + w.dbeta, n_rays);
    ls2fm_prof_end(LS2FM_PROF_FINALIZE, gs);
    if (forked && hipEventRecord(sc.join, sc.side) != hipSuccess) return ls2fm_join_on_error(forked, sc, s, LS2FM_ERR_LAUNCH);

    // hash-table gradients: payloads sorted by slab (scatter_fill), then one streaming pass per LDS-owned slab
    // (slab_accumulate; bin_scatter.hip); tables overwritten in full; dual field: both grids share geometry, hence items.
    // opts->n_level_groups > 1 (multi-GPU runs): the levels are processed in that many consecutive groups and an event is
    // recorded after each group's accumulate, so that the caller can start all-reducing a group's slices of the gradient
    // tables while the later groups are still being scattered.
    {
        int groups = opts ? opts->n_level_groups : 1;
        groups = groups < 1 ? 1 : (groups > LS2FM_MAX_LEVEL_GROUPS ? LS2FM_MAX_LEVEL_GROUPS : groups);
        if (groups > L1) groups = L1;
        for (int gi = 0; gi < groups; ++gi) {
            const int lo = L1 * gi / groups, hi = L1 * (gi + 1) / groups;
            ls2fm_prof_begin(LS2FM_PROF_SCATTER_RAD, s);
            int st = ls2fm_launch_scatter_fill(sdf_grid, fc, center, ray, ws + w.bins, w.p, P, ws + w.rec1, dual ? ws + w.rec2 : nullptr,
                                               ws + w.rpt, ws + w.smax, n_rays, dual, s, lo, hi);
            ls2fm_prof_end(LS2FM_PROF_SCATTER_RAD, s);
            if (st != LS2FM_OK) return ls2fm_join_on_error(forked, sc, s, st);
            ls2fm_prof_begin(LS2FM_PROF_SCATTER_SDF, s);
            st = ls2fm_launch_slab_accumulate(sdf_grid, ws + w.bins, w.p, grads->sdf_table, dual ? grads->rad_table : nullptr, s, lo, hi);
            ls2fm_prof_end(LS2FM_PROF_SCATTER_SDF, s);
            if (st != LS2FM_OK) return ls2fm_join_on_error(forked, sc, s, st);
            if (opts && opts->n_level_groups > 1 && opts->group_events[gi] &&
                hipEventRecord((hipEvent_t)opts->group_events[gi], s) != hipSuccess)
                return ls2fm_join_on_error(forked, sc, s, LS2FM_ERR_LAUNCH);
        }
    }
    if (forked && hipStreamWaitEvent(s, sc.join, 0) != hipSuccess) return LS2FM_ERR_LAUNCH;       // join
    return ls2fm_launch_status();
}
