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

#include "side_jobs.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------- finalize (wgrad_tail.h)
__global__ void __launch_bounds__(256)
finalize_kernel(FinalizeArgs fa) {
    finalize_task(fa, (int)blockIdx.x);
}

// ONE launch for the sum of the weight-gradient partials and the finalize tasks that consume it: workgroups [0, n_red) reduce
// a row each (write-through stores, drained) and take a ticket; the kFinalizeTasks workgroups after them wait for the ticket to
// reach n_red, then run their task on L2-bypassing loads of the reduced block.  All of them fit the chip at once many times over (and beside slab_accumulate's one 128 KB workgroup per
// CU: 21 KB of LDS), the reducers are dispatched first and wait for nothing, so the waiters cannot starve them; the wait is
// bounded anyway.  The ticket lives in the slack of the reduced-gradient block, which shade_bwd's leading workgroups zero.
__global__ void __launch_bounds__(256)
wgrad_tail_kernel(WgradParts wp, float* __restrict__ wg, FinalizeArgs fa, int n_red, int* __restrict__ ticket, int* __restrict__ err) {
    const int bid = (int)blockIdx.x;
    if (bid < n_red) {
        reduce_partials_row(wp, wg, bid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's write-through stores have landed
        __syncthreads();                                         // (the row is written by wave 0)
        if (threadIdx.x == 0) __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __shared__ int s_starved;
    if (threadIdx.x == 0) {
        int spins = 0;
        while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_red && ++spins < (1 << 24))
            __builtin_amdgcn_s_sleep(16);
        s_starved = __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_red;
    }
    __syncthreads();
    const int task = bid - n_red;
    finalize_task(fa, task);                                     // reads the reduced block with L2-bypassing loads (wg_load)
    // The wait is bounded (HIP promises no dispatch order inside a launch): if the reducers were starved past the bound, this
    // task consumed incomplete sums -- poison its outputs so that the step fails loudly (NaN loss / gradient) instead of quietly
    if (s_starved && threadIdx.x == 0 && err) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // (ls2fm_async_error)
    if (s_starved && threadIdx.x == 0 && (task < 2 || task >= 4 || fa.dual)) {
        const float nan = __builtin_nanf("");
        float* b = task < 2 ? fa.G.sdf_mlp[task].bias : (task < 4 ? fa.G.geo_mlp[task - 2].bias : fa.G.rad_mlp[task - 4].bias);
        b[0] = nan;
        if (task == 4) fa.G.beta[0] = nan;
    }
}

// The upstream of the traced depth (loss head inside the render: the depth-consistency term) per ray -- exactly the lines of
// shade_bwd's first part that form `d_depth_ref`, as a kernel of its own, so that the tracing's own backward (ls2fm_depth_backward: a
// chain of five small launches over the track points) can start BESIDE shade_bwd instead of behind it.
__global__ void __launch_bounds__(256)
depth_upstream_kernel(LossUp lo, const float* __restrict__ rout_depth, int64_t n_rays) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rays) return;
    float gt[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (lo.d_terms) {
#pragma unroll
        for (int k = 0; k < 5; ++k) gt[k] = lo.d_terms[k];
    }
    const float g_all = gt[4] + (lo.d_total ? lo.d_total[0] : 0.f);
    const float gl_dc = lo.sums[5] > 0.0 ? fmaf(lo.weights[2], g_all, gt[2]) / (float)lo.sums[5] : 0.f;
    float gd = 0.f;
    if (lo.depth_ref != nullptr && (lo.mask_dc == nullptr || lo.mask_dc[r] != 0))
        gd = gl_dc * ls2fm_smooth_l1_grad(lo.depth_ref[r] - rout_depth[r]);
    lo.d_depth_ref[r] = gd;
}

}  // namespace

// weight-norm backward of the SDF MLP alone (point queries, points.hip): tasks 0 and 1 of finalize_kernel
int ls2fm_launch_finalize_sdf(const ls2fm_params* params, const ls2fm_param_grads* grads, int in_dim, const Packed* pk,
                              const float* wg, hipStream_t stream, int add) {
    finalize_kernel<<<2, 256, 0, stream>>>(FinalizeArgs{*params, *grads, in_dim, 0, 0, 0, pk, wg, nullptr, 0, add});
    return ls2fm_launch_status();
}

// ------------------------------------------------------------------------------------------- C ABI
int ls2fm_launch_scatter_fill(const ls2fm_grid_desc* grid, const FieldC& fc, const float* center, const float* ray, float* bins_ws,
                              int64_t n_points, int64_t p_pad, const float* rec1, const float* rec2, const float* rpt,
                              const float* ray_bound, int64_t n_rays, int dual, hipStream_t stream, int level_lo = 0, int level_hi = -1,
                              int n_explicit = 0, const void* side_jobs = nullptr);
int ls2fm_launch_slab_accumulate(const ls2fm_grid_desc* grid, float* bins_ws, int64_t n_points, float* dtable1, float* dtable2,
                                 hipStream_t stream, int level_lo = 0, int level_hi = -1, int add_into = 0, int n_explicit = 0,
                                 int n_samples = 1);

bool ls2fm_bins_levels_fit(const ls2fm_grid_desc* grid, int dual);

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
    // (the forward refuses such a batch, too; said again here because shade_bwd's 32-bit byte offsets depend on it)
    if (n_rays * (int64_t)field->n_samples > LS2FM_MAX_RENDER_POINTS) return LS2FM_ERR_UNSUPPORTED;
    LS2FM_CHECK_ARG((d_center == nullptr) == (d_ray == nullptr));     // pose gradients: both or neither
    const int want_pose = d_center != nullptr;
    const ls2fm_loss_spec* loss = opts ? opts->loss : nullptr;
    LS2FM_CHECK_ARG(!loss || (loss->rgb_gt && loss->weights && loss->sums && (loss->d_terms || loss->d_total)));
    if (n_rays == 0) return LS2FM_OK;
    LS2FM_CHECK_ARG(center && ray && grads->sdf_table && grads->beta && (!field->dual_field || grads->rad_table));
    if (!workspace) return LS2FM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    // an earlier backward's in-launch hand-off gave up (its gradients are poisoned): reported here, sticky until cleared
    int* const err_word = ls2fm_async_error_word(s);
    if (err_word && *static_cast<volatile int*>(err_word) != 0) return LS2FM_ERR_STARVED;
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
    bool forked = false, joined = false;
    LS2FM_CHECK_ARG(((reinterpret_cast<uintptr_t>(grads->sdf_table) | (dual ? reinterpret_cast<uintptr_t>(grads->rad_table) : 0)) & 15u) == 0);

    Upstream up{d_rgb, d_sdfs_volume, d_normals, d_depth_mlp, d_normal_mlp, LossUp{}};
    if (loss)
        up.loss = LossUp{loss->rgb_gt, loss->depth_ref, loss->mask_eik, loss->mask_dc, loss->mask_mse, loss->weights, loss->sums,
                         loss->d_terms, loss->d_total, loss->d_depth_ref, loss->flags};
    // (rays of fewer than 64 samples -- BASELINE configs[0]: 256 rays x 32 -- leave half of a wave's lanes empty in the fused form:
    // C1 0.171 fused against 0.154 ms with the separate launches)
    const int fused_wgrad = ls2fm_fused_wgrad(field->n_samples) ? 1 : 0;
    // the tracing's own backward (ls2fm_depth_backward): a second internal branch.  Its stages are those of ls2fm_sdf_points_bwd
    // over the track points, MERGED into this call's own chains: the per-point rows it leaves (front: gather pass, points_bwd,
    // scans) are contracted by this call's SDF weight-gradient kernel as extra tiles (no second wgrad_mlp / reduction / weight-norm
    // backward -- whose 63 KB workgroups waited for a CU beside this call's), and its table gradient is ADDED into this call's table
    // behind the main accumulate (no second table, no final sum kernel).
    // Round 5, measured and NOT taken (LS2FM_DEPTH_EARLY=1 enables it; results identical): the branch forked IN FRONT of shade_bwd --
    // the depth's upstream from depth_upstream_kernel (the same arithmetic as shade_bwd's first part, which then leaves d_depth_ref
    // alone) -- so that its five small launches run beside shade_bwd instead of beside the scatter, where they are stretched 3-5x and
    // end the stage's backward 90 us behind the main accumulate.  shade_bwd is ONE round of workgroups that fills every slot of the
    // chip: whatever runs beside it starts a second round -- stage step 0.808 - 0.820 -> 0.856 - 0.864 ms, loops +0.06 - 0.1 ms.
    const ls2fm_depth_backward* db = (opts && loss && loss->d_depth_ref) ? opts->depth_bwd : nullptr;
    static const int depth_early_env = [] { const char* e = getenv("LS2FM_DEPTH_EARLY"); return e ? atoi(e) : 0; }();
    const bool depth_early = db != nullptr && depth_early_env != 0;
    SideCtx sc1, sc2;
    bool forked2 = false;
    Ls2fmWgradExtra extra{};
    int64_t n_track = 0;
    hipStream_t ds = s;
    auto launch_depth_branch = [&]() -> int {
        LS2FM_CHECK_ARG(db->points && db->trips && db->gate && db->d_sdf && db->workspace && db->k_max >= 1);
        if (opts->n_level_groups > 1) return LS2FM_ERR_UNSUPPORTED;      // a group's slices would not be final at its event
        n_track = n_rays * (int64_t)db->k_max;
        if (n_track > LS2FM_MAX_RENDER_POINTS || !ls2fm_bins_levels_fit(sdf_grid, 0)) return LS2FM_ERR_UNSUPPORTED;
        // (a stream of its own: the side stream of THIS call's side stream)
        forked2 = ls2fm_side_stream(&sc1, s) && ls2fm_side_stream(&sc2, sc1.side) && hipEventRecord(sc2.fork, s) == hipSuccess &&
                  hipStreamWaitEvent(sc2.side, sc2.fork, 0) == hipSuccess;
        ds = forked2 ? sc2.side : s;
        int st = ls2fm_trace_depth_bwd(loss->d_depth_ref, nullptr, db->trips, db->gate, n_rays, db->k_max, db->d_sdf, ds);
        if (st == LS2FM_OK)
            st = ls2fm_points_bwd_front(field, sdf_grid, params, db->points, n_track, db->d_sdf, nullptr, nullptr, nullptr, false,
                                        db->workspace, ds, &extra.w, forked2 ? sc2.mid : nullptr);
        extra.ws = (const float*)db->workspace;
        if (forked2) extra.ready = sc2.mid;
        if (st == LS2FM_OK) st = ls2fm_points_bwd_scatter(field, sdf_grid, n_track, db->workspace, nullptr, 0, ds, /*phase=*/1);
        if (st != LS2FM_OK) {
            if (forked2) { (void)hipEventRecord(sc2.join, sc2.side); (void)hipStreamWaitEvent(s, sc2.join, 0); }
            return st;
        }
        return LS2FM_OK;
    };
    if (depth_early) {
        depth_upstream_kernel<<<(unsigned)((n_rays + 255) / 256), 256, 0, s>>>(up.loss, ws + w.rout + 3 * w.r_pad, n_rays);
        if (opts->depth_grad_ready && hipEventRecord((hipEvent_t)opts->depth_grad_ready, s) != hipSuccess) return LS2FM_ERR_LAUNCH;
        const int st = launch_depth_branch();
        if (st != LS2FM_OK) return st;
        up.loss.d_depth_ref = nullptr;                   // (shade_bwd leaves it alone: the branch is reading it)
    }
#ifdef LS2FM_DEBUG_PTRS
    {   // (diagnosis, round 6: which of shade_bwd's pointers is not device memory?)
        static int once = 0;
        if (!once++) {
            auto show = [](const char* name, const void* p) {
                hipPointerAttribute_t a{};
                if (!p) { fprintf(stderr, "[ptr] %-12s null\n", name); return; }
                const hipError_t e = hipPointerGetAttributes(&a, p);
                if (e != hipSuccess) (void)hipGetLastError();
                fprintf(stderr, "[ptr] %-12s %p  err %d  type %d  device %d  managed %d\n", name, p, (int)e, (int)a.type, a.device, (int)a.isManaged);
            };
            show("center", center); show("ray", ray); show("ws", ws); show("pk", pk);
            show("rgb_gt", up.loss.rgb_gt); show("depth_ref", up.loss.depth_ref); show("weights", up.loss.weights); show("sums", up.loss.sums);
            show("d_terms", up.loss.d_terms); show("d_total", up.loss.d_total); show("d_depth_ref", up.loss.d_depth_ref);
            show("sdf_table", grads->sdf_table); show("beta", grads->beta);
        }
    }
#endif
    ls2fm_prof_begin(LS2FM_PROF_SHADE_BWD, s);
    // (leading workgroups of this launch zero the weight-gradient accumulators and the atomically flushed table ranges)
    // the MLPs' weight gradients are contracted inside shade_bwd (LS2FM_FUSED_WGRAD=0: round 4's separate wgrad_mlp launches,
    // for A/B measurements)
    // (one-sample rays share the workspace layout of free points, which has no per-ray partials: wgrad_mlp.hip serves them)
    ls2fm_launch_shade_bwd(fc, lsc, dual, 2 * L1, 2 * L2, w, pk, center, ray, n_rays, ws, up, want_pose, sdf_grid, grads->sdf_table,
                           dual ? grads->rad_table : nullptr, s, fused_wgrad);
    ls2fm_prof_end(LS2FM_PROF_SHADE_BWD, s);
    if (!depth_early) {
        if (opts && opts->depth_grad_ready && hipEventRecord((hipEvent_t)opts->depth_grad_ready, s) != hipSuccess) return LS2FM_ERR_LAUNCH;
        if (db) { const int st = launch_depth_branch(); if (st != LS2FM_OK) return st; }
    }
    auto fail = [&](bool f1, const SideCtx& c1, int st) {       // error after the forks: every branch is joined back into `s`
        if (forked2) { (void)hipEventRecord(sc2.join, sc2.side); (void)hipStreamWaitEvent(s, sc2.join, 0); }
        return ls2fm_join_on_error(f1, c1, s, st);
    };
    if (want_pose) {
        ls2fm_prof_begin(LS2FM_PROF_POSE, s);
        ls2fm_launch_pose_grad(fc, sdf_grid, rad_grid, dual, w, pk, params, center, ray, n_rays, ws, d_center, d_ray, s);
        ls2fm_prof_end(LS2FM_PROF_POSE, s);
    }

    // the one fork of the backward: weight-gradient GEMMs -> reduce -> finalize run on the side stream, beside the table scatters
    static const int probe_no_side = [] { const char* e = getenv("LS2FM_PROBE_NO_SIDE"); return e ? atoi(e) : 0; }();     // timing probe: wrong MLP gradients
    // Round 5: NO side stream in the render's backward when the MLPs' weight gradients were contracted by shade_bwd and no traced
    // depth rides along: what is left of the chain (decoder columns, level-1 sums, reduction rows, finalize tasks) runs as leading
    // workgroups of the scatter_fill launch (side_jobs.h) -- one chain on one queue.  LS2FM_SIDE_IN_FILL=0: the side stream of
    // rounds 2-4 (A/B).
    static const int side_in_fill_env = [] { const char* e = getenv("LS2FM_SIDE_IN_FILL"); return e ? atoi(e) : 1; }();
    // (only where the fill is long enough to hide the jobs' ~45 us chain of hand-offs: >= 32 k sample points; and while the level-1
    // sums -- 39 KB of per-ray partials per ray -- stay small beside the fill's own traffic: C5, 4096 rays x 256: 2.002 -> 1.984 ms;
    // C3, 8192 rays x 128: 3.074 -> 3.100)
    // (and only on a device with an order of magnitude more workgroup slots than the launch has WAITING workgroups -- the reduction
    // rows and finalize tasks, ~180: side_jobs.h -- so that the producers they poll can never be kept off the chip by them)
    const int n_waiters = kRegsSdf + (dual ? kRegsGeo : 0) + kRegsDec + kFinalizeTasks;
    static const int64_t side_max_rays = [] { const char* e = getenv("LS2FM_SIDE_IN_FILL_MAX_RAYS"); return e ? atoll(e) : 4096ll; }();
    const bool side_in_fill = side_in_fill_env && fused_wgrad && !db && !probe_no_side && w.p >= 32768 && n_rays <= side_max_rays &&
                              ls2fm_device_cus() >= n_waiters;      // (>= 4 workgroups of this launch fit a CU: 4x the slots)
    SideJobs sj{};
    if (side_in_fill) {
        const WgPartLayout pl = make_wg_part_layout(dual, n_rays, field->n_samples);
        float* l1_sdf = ws + w.mpart + pl.l1_sdf;
        float* l1_geo = ws + w.mpart + pl.l1_geo;
        sj.dec_blocks = kSideDecBlocks;
        sj.l1_jobs = kL1Seg * (kRegsSdf + (dual ? kRegsGeo : 0));
        sj.n_red = kRegsSdf + (dual ? kRegsGeo : 0) + kRegsDec;
        sj.w = w; sj.dual = dual; sj.n_rays = n_rays; sj.ws = ws;
        sj.part_dec = ws + w.mpart + pl.dec;
        sj.l1 = L1Job{ws + w.mpart + pl.slot_sdf, ws + w.mpart + pl.slot_geo, l1_sdf, l1_geo, pl.n_slots, dual};
        sj.parts = WgradParts{l1_sdf, l1_geo, sj.part_dec, kL1Seg, kL1Seg, sj.dec_blocks, dual};
        sj.fa = FinalizeArgs{*params, *grads, 3 + 2 * L1, 3 + 2 * L2, rad_in, dual, pk, ws + w.wg, ws + w.dbeta, n_rays, 0};
        static_assert(kSideDecBlocks + kL1Seg * (kRegsSdf + kRegsGeo) + kRegsSdf + kRegsGeo + kRegsDec <= kSideFlagInts, "one flag per side job");
        sj.flags = reinterpret_cast<int*>(ws + w.wg + kWgFlagsAt);                  // (zeroed by shade_bwd's leading workgroups)
        static const int side_probe = [] { const char* e = getenv("LS2FM_SIDE_PROBE"); return e ? atoi(e) : 0; }();
        sj.probe = side_probe;
        sj.err = err_word;
    }
    forked = !probe_no_side && !side_in_fill && ls2fm_side_stream(&sc, s) && hipEventRecord(sc.fork, s) == hipSuccess && hipStreamWaitEvent(sc.side, sc.fork, 0) == hipSuccess;
    hipStream_t gs = forked ? sc.side : s;
    // The side chain is enqueued IN FRONT of the scatter.  (Behind it -- LS2FM_SIDE_FIRST=0, tried in round 5 now that the chain is
    // short: the scatter then keeps shade_bwd's hardware queue in a hipGraph replay -- measured 0.526 against 0.509 ms per step.)
    static const int side_first = [] { const char* e = getenv("LS2FM_SIDE_FIRST"); return e ? atoi(e) : 1; }();
    auto launch_side = [&]() -> int {
    if (probe_no_side || side_in_fill) return LS2FM_OK;
    Ls2fmWgradParts parts{};
    if (ls2fm_launch_wgrad_mlp(fc, dual, 2 * L1, 2 * L2, w, pk, center, ray, n_rays, ws, gs, false, &parts, db ? &extra : nullptr,
                               fused_wgrad) != LS2FM_OK)
        return fail(forked, sc, LS2FM_ERR_LAUNCH);
    ls2fm_prof_begin(LS2FM_PROF_FINALIZE, gs);
    {   // sum of the partials + finalize tasks, one launch (the ticket word: slack of the reduced-gradient block, zeroed above)
        static_assert(WgLayout::total % 64 != 0 && (WgLayout::total + 63) / 64 * 64 - WgLayout::total >= 1, "ticket word in the block's slack");
        const int n_red = kRegsSdf + (dual ? kRegsGeo : 0) + kRegsDec;
        const FinalizeArgs fa{*params, *grads, 3 + 2 * L1, 3 + 2 * L2, rad_in, dual, pk, ws + w.wg, ws + w.dbeta, n_rays, 0};
        wgrad_tail_kernel<<<n_red + kFinalizeTasks, 256, 0, gs>>>(parts, ws + w.wg, fa, n_red,
                                                                 reinterpret_cast<int*>(ws + w.wg + WgLayout::total), err_word);
    }
    ls2fm_prof_end(LS2FM_PROF_FINALIZE, gs);
    if (forked && hipEventRecord(sc.join, sc.side) != hipSuccess) return fail(forked, sc, LS2FM_ERR_LAUNCH);
    return LS2FM_OK;
    };
    if (side_first) { const int st = launch_side(); if (st != LS2FM_OK) return st; }

    // hash-table gradients: payloads sorted by slab (scatter_fill), then one streaming pass per LDS-owned slab
    // (slab_accumulate; bin_scatter.hip); tables overwritten in full; dual field: both grids share geometry, hence items.
    // opts->n_level_groups > 1 (multi-GPU runs): the levels are processed in that many consecutive groups and an event is
    // recorded after each group's accumulate, so that the caller can start all-reducing a group's slices of the gradient
    // tables while the later groups are still being scattered.
    {
        const int n_explicit = ls2fm_explicit_levels(sdf_grid, dual, field->n_samples);      // as the forward's counting pass
        int groups = opts ? opts->n_level_groups : 1;
        static const int groups_env = [] { const char* e = getenv("LS2FM_LEVEL_GROUPS"); return e ? atoi(e) : 0; }();
        if (groups <= 1 && groups_env > 1 && !db) groups = groups_env;       // (measurement switch: no events, same results)
        groups = groups < 1 ? 1 : (groups > LS2FM_MAX_LEVEL_GROUPS ? LS2FM_MAX_LEVEL_GROUPS : groups);
        if (groups > L1) groups = L1;
        for (int gi = 0; gi < groups; ++gi) {
            const int lo = L1 * gi / groups, hi = L1 * (gi + 1) / groups;
            ls2fm_prof_begin(LS2FM_PROF_SCATTER_RAD, s);
            int st = ls2fm_launch_scatter_fill(sdf_grid, fc, center, ray, ws + w.bins, w.p, P, ws + w.rec1, dual ? ws + w.rec2 : nullptr,
                                               ws + w.rpt, ws + w.smax, n_rays, dual, s, lo, hi, n_explicit,
                                               (side_in_fill && gi == 0) ? &sj : nullptr);
            ls2fm_prof_end(LS2FM_PROF_SCATTER_RAD, s);
            if (st != LS2FM_OK) return fail(forked, sc, st);
            // LS2FM_JOIN_EARLY=1: the side chain is joined in FRONT of the (last) accumulate launch instead of at the end of the
            // call: the call then ends on one queue
            static const int join_early = [] { const char* e = getenv("LS2FM_JOIN_EARLY"); return e ? atoi(e) : 0; }();
            if (join_early && side_first && forked && !joined && gi == groups - 1) {
                if (hipStreamWaitEvent(s, sc.join, 0) != hipSuccess) return fail(forked, sc, LS2FM_ERR_LAUNCH);
                joined = true;
            }
            ls2fm_prof_begin(LS2FM_PROF_SCATTER_SDF, s);
            st = ls2fm_launch_slab_accumulate(sdf_grid, ws + w.bins, w.p, grads->sdf_table, dual ? grads->rad_table : nullptr, s, lo, hi,
                                              0, n_explicit, field->n_samples);
            ls2fm_prof_end(LS2FM_PROF_SCATTER_SDF, s);
            if (st != LS2FM_OK) return fail(forked, sc, st);
            if (opts && opts->n_level_groups > 1 && opts->group_events[gi] &&
                hipEventRecord((hipEvent_t)opts->group_events[gi], s) != hipSuccess)
                return fail(forked, sc, LS2FM_ERR_LAUNCH);
        }
    }
    if (!side_first) { const int st = launch_side(); if (st != LS2FM_OK) return st; }
    if (db) {
        // the tracing's table gradient, added into the table the scatter above has just written: behind it (an event when the
        // branch has a stream of its own)
        if (forked2 && (hipEventRecord(sc1.mid, s) != hipSuccess || hipStreamWaitEvent(ds, sc1.mid, 0) != hipSuccess))
            return fail(forked, sc, LS2FM_ERR_LAUNCH);
        const int st = ls2fm_points_bwd_scatter(field, sdf_grid, n_track, db->workspace, grads->sdf_table, 1, ds, /*phase=*/2);
        if (st != LS2FM_OK) return fail(forked, sc, st);
        if (forked2 && (hipEventRecord(sc2.join, sc2.side) != hipSuccess || hipStreamWaitEvent(s, sc2.join, 0) != hipSuccess))
            return ls2fm_join_on_error(forked, sc, s, LS2FM_ERR_LAUNCH);
    }
    if (forked && !joined && hipStreamWaitEvent(s, sc.join, 0) != hipSuccess) return LS2FM_ERR_LAUNCH;       // join
    return ls2fm_launch_status();
}
