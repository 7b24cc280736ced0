/*
 * ls2fm.h -- C ABI of libls2fm_hip.so: the MI355X (gfx950) implementation of Level-S2fM's
 * SDF ray-marching / volumetric-rendering hot path.
 *
 * Every entry point replaces something the reference reaches through a native extension or
 * through PyTorch device code on that path; the replaced interface is cited per function
 * (file:line relative to the reference tree).  The reference-side bindings a maintainer
 * would add are shown in INTEGRATION.md.
 *
 * Conventions (all functions)
 *   - plain C types only: raw DEVICE pointers, sizes, small by-value descriptor structs that
 *     live in HOST memory; no torch types
 *   - all tensors are contiguous, row-major; float = IEEE fp32, indices are uint32/int32/int64
 *   - the caller owns every buffer (inputs, outputs, gradient accumulators, workspace); the
 *     library never allocates device memory and keeps no global state
 *   - asynchronous: work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream); buffers must stay alive until the stream reaches the end of the call
 *   - gradient outputs documented as "accumulate" are added to with atomics and must be
 *     zeroed (or hold a running sum) by the caller; all others are overwritten
 *   - return value: 0 on success, a negative ls2fm_status otherwise (nothing is enqueued on
 *     LS2FM_ERR_INVALID_ARGUMENT / LS2FM_ERR_UNSUPPORTED); thread-safe for distinct streams
 */
#ifndef LS2FM_H
#define LS2FM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LS2FM_ABI_VERSION 9
#define LS2FM_MAX_LEVELS 16
#define LS2FM_HIDDEN 64        /* SDF.arch.layers = [null, 64, 16]  (options/LevelS2fM.yaml:14) */
#define LS2FM_FEAT 16
#define LS2FM_VIEW_ENC 27      /* Fourier view embedding, 4 octaves (models/base.py:143-151) */

typedef enum ls2fm_status {
    LS2FM_OK = 0,
    LS2FM_ERR_INVALID_ARGUMENT = -1,
    LS2FM_ERR_UNSUPPORTED = -2,
    LS2FM_ERR_LAUNCH = -3,
    LS2FM_ERR_WORKSPACE = -4,
    LS2FM_ERR_STARVED = -5     /* an EARLIER asynchronous call reported a starved in-launch hand-off (ls2fm_async_error) */
} ls2fm_status;

/* Geometry of one multiresolution hash grid (tcnn Grid/Hash/Linear, n_features_per_level = 2).
 * Built on the host (level-s2fm_official_amd/ls2fm/hashgrid.py) exactly as tcnn's constructor
 * does (grid_scale / grid_resolution / offset table); replaces the `encoding_config` dict the
 * reference hands to tinycudann.Encoding (models/base.py:17, :130-139). */
typedef struct ls2fm_grid_desc {
    int32_t n_levels;                         /* <= LS2FM_MAX_LEVELS */
    int32_t n_features;                       /* must be 2 */
    float scale[LS2FM_MAX_LEVELS];            /* grid_scale(level) */
    uint32_t resolution[LS2FM_MAX_LEVELS];    /* ceil(scale) + 1 */
    uint32_t size[LS2FM_MAX_LEVELS];          /* entries in the level */
    uint32_t offset[LS2FM_MAX_LEVELS + 1];    /* first entry of the level (in entries, not floats) */
    uint32_t hashed[LS2FM_MAX_LEVELS];        /* 1: coherent prime hash, 0: dense stride walk */
} ls2fm_grid_desc;

/* Scene / field constants the path reads from `opt` (SURVEY.md section 5 "config / flags"). */
typedef struct ls2fm_field_desc {
    float bound_min[3];      /* opt.data.bound_min */
    float bound_max[3];      /* opt.data.bound_max */
    float rescale;           /* opt.SDF.VolSDF.rescale        (models/base.py:38-40) */
    float scale_mlp;         /* opt.SDF.NN_Init.scale_mlp: sdf = +-f0 / scale_mlp (true division) */
    int32_t inside;          /* opt.data.inside: sign of the sdf (models/SDF.py:66-71) */
    int32_t bg_sdf;          /* opt.data.inside && opt.data.bg_sdf: sdf = min(sdf, bg_rad - |p|) */
    float bg_rad;
    float bgcolor[3];        /* models/Renderer.py:25-31 */
    int32_t n_samples;       /* opt.SDF.VolSDF.sample_intvs */
    int32_t dual_field;      /* opt.Ablate_config.dual_field */
} ls2fm_field_desc;

/* One weight-normed linear layer as the reference stores it (legacy torch weight_norm, dim 0):
 * W[o][i] = g[o] * v[o][i] / ||v[o]||,  models/base.py:200, :241; state_dict keys App. E. */
typedef struct ls2fm_linear {
    const float* weight_v;   /* [out][in] */
    const float* weight_g;   /* [out]     */
    const float* bias;       /* [out]     */
} ls2fm_linear;

typedef struct ls2fm_linear_grad {      /* same shapes; overwritten */
    float* weight_v;
    float* weight_g;
    float* bias;
} ls2fm_linear_grad;

/* All learnable state of the path (device pointers into the nn.Parameters). */
typedef struct ls2fm_params {
    const float* sdf_table;             /* SDF.embed_fn.embedder_obj.params        [n_params] */
    ls2fm_linear sdf_mlp[2];            /* SDF.SDF_MLP.mlp.{0,1}: (3+2L)->64->17   */
    const float* beta;                  /* SDF.beta [1] (log-space parameter, SDF.py:28-32) */
    float beta_speed;                   /* opt.SDF.VolSDF.beta_speed */
    const float* rad_table;             /* RadF.embed_fn.embedder_obj.params (dual field) or NULL */
    ls2fm_linear geo_mlp[2];            /* RadF.Geo_enc.mlp.{0,1} (dual field) */
    ls2fm_linear rad_mlp[3];            /* RadF.Rad_dec.mlp_radiance.{0,1,2}: (49|65)->64->64->3 */
    const float* dual_table;            /* optional (dual field): both tables entry-interleaved, [n_entries][4] =
                                           {sdf f0, sdf f1, rad f0, rad f1}, as ls2fm_interleave_tables writes it and in
                                           sync with sdf_table / rad_table; NULL: gather from the two tables */
} ls2fm_params;

typedef struct ls2fm_param_grads {      /* mirrors ls2fm_params */
    float* sdf_table;                   /* overwritten in full (LDS-slab scatter, no atomics) */
    ls2fm_linear_grad sdf_mlp[2];
    float* beta;                        /* [1] overwritten */
    float* rad_table;                   /* overwritten in full */
    ls2fm_linear_grad geo_mlp[2];
    ls2fm_linear_grad rad_mlp[3];
} ls2fm_param_grads;

int ls2fm_abi_version(void);
const char* ls2fm_status_string(int status);
/* Sticky error word of the ASYNCHRONOUS part of earlier calls (0 = none).  The render backward hands partial sums between
 * workgroups of ONE launch (weight-gradient reduction rows / finalize tasks behind their producers); consumers poll with a bound, and
 * one that gives up (producers starved of execution slots: a device cut down by a CU mask, a foreign kernel that never ends) poisons
 * its outputs with NaN AND sets this word (1) in host-visible memory -- no synchronisation is needed to read it, and every later
 * ls2fm_render_bwd returns LS2FM_ERR_STARVED until it is cleared.  clear != 0: reset after reading.  No reference counterpart
 * (autograd's kernels have no in-launch hand-offs). */
int ls2fm_async_error(int clear);

/* ---------------------------------------------------------------------------------------------
 * Ray / AABB slab test.
 * Replaces: vren.ray_aabb_intersect as bound by utils/custom_functions.py:28-31
 *           (called from models/Renderer.py:178-179 and models/SDF.py:120-121).
 * rays_o, rays_d [n_rays,3] (directions NOT normalised); center, half_size [n_voxels,3].
 * Outputs: hits_cnt int32[n_rays]; hits_t float[n_rays,max_hits,2] (near clamped to >= 0; -1,-1 in
 * unused slots); hits_voxel_idx int64[n_rays,max_hits] (-1 in unused slots); slots sorted by near t.
 */
int ls2fm_ray_aabb_intersect(const float* rays_o, const float* rays_d, const float* center,
                             const float* half_size, int64_t n_rays, int32_t n_voxels, int32_t max_hits,
                             int32_t* hits_cnt, float* hits_t, int64_t* hits_voxel_idx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Multiresolution hash-grid encoding (fp32), forward / backward / double backward.
 * Replaces: tinycudann.Encoding(Grid, Hash, Linear) -- constructor models/base.py:17, forward
 *           models/base.py:37; its backward is reached through loss.backward() and its double
 *           backward through SDF.gradient's create_graph=True (models/SDF.py:102-114).
 * x [n,3] is the NORMALISED position the reference feeds ((p-bmin)/(bmax-bmin), base.py:35); no
 * clamping.  y [n, L*2] is level-major ([l*2+f]).  dy_dx [n, L*2, 3] optional (NULL to skip).
 */
int ls2fm_grid_encode_fwd(const ls2fm_grid_desc* grid, const float* x, const float* table, int64_t n,
                          float* y, float* dy_dx, void* stream);

/* dtable (accumulate, NULL to skip) += scatter of dy;  dx [n,3] (overwritten, NULL to skip). */
int ls2fm_grid_encode_bwd(const ls2fm_grid_desc* grid, const float* x, const float* table, const float* dy,
                          int64_t n, float* dtable, float* dx, void* stream);

/* Backward of the map (dy, table, x) -> dx of ls2fm_grid_encode_bwd, given ddx = dL/d(dx) [n,3]:
 *   d_dy [n,L*2] (overwritten), dtable (accumulate), dx2 [n,3] (overwritten; mixed second partials).
 * Any output may be NULL. */
int ls2fm_grid_encode_bwd_bwd(const ls2fm_grid_desc* grid, const float* x, const float* table,
                              const float* dy, const float* ddx, int64_t n, float* d_dy, float* dtable,
                              float* dx2, void* stream);

/* Debug / parity: the 8 level-local corner indices per (point, level): out uint32[n, L, 8]. */
int ls2fm_grid_indices(const ls2fm_grid_desc* grid, const float* x, int64_t n, uint32_t* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused SDF evaluation without autograd: hash encode -> 35->64 softplus(100) -> 64->17, sign /
 * scale_mlp, optional analytic normal d sdf / d p.
 * Replaces: SDF.infer_sdf (models/SDF.py:55-78) in its no-grad uses (sphere-tracing inner loop
 *           SDF.py:185-196, mesh sweeps utils/util.py:426-428) and SDF.gradient (SDF.py:102-114)
 *           when no graph is needed.
 * p [n,3] world positions.  sdf [n]; feat [n,17] or NULL; normal [n,3] or NULL.
 * workspace: ls2fm_sdf_eval_workspace_bytes() bytes.
 */
int64_t ls2fm_sdf_eval_workspace_bytes(void);
int ls2fm_sdf_eval(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                   const float* p, int64_t n, float* sdf, float* feat, float* normal, void* workspace,
                   void* stream);

/* No-grad SDF sweep over an n_side^3 lattice generated on the device: sdf[i] = infer_sdf(point(first + i)), i < count.
 * Replaces: the batchify loop of extract_mesh (utils/util.py:411-424: 16 k-point chunks, each a host->device copy,
 * infer_sdf, device->host copy) and its host-side numpy lattice (util.py:399-409).
 * With idx = first + i:  ref_indexing != 0: column c of the point = fl32(f_c * step[c] + origin[c]) in fp64 with
 * f = (fmod(idx/N/N, N), fmod(idx/N, N), idx mod N) under TRUE division -- exactly the reference's arithmetic (its first
 * two index columns are fractional); ref_indexing == 0: f = integer lattice indices (idx div N^2, idx div N mod N,
 * idx mod N).  workspace: ls2fm_sdf_eval_workspace_bytes().
 */
int ls2fm_sdf_volume(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                     int64_t n_side, int64_t first, int64_t count, int32_t ref_indexing, const double* step,
                     const double* origin, float* sdf, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Point queries WITH a graph: backward of  p -> (sdf, the 17 MLP outputs, the analytic normal d sdf / d p)  for n free points.
 * Replaces: what loss.backward() traverses for SDF.infer_sdf with parameters requiring grad (models/SDF.py:55-78),
 * SDF.gradient (SDF.py:102-114: autograd.grad(create_graph=True) -- callers take norms of it inside losses:
 * pipelines/Registration.py:202,259-261, BA.py:123-125) and SDF.get_surface_pts (SDF.py:95-100): tcnn's backward and
 * double-backward kernels plus the torch Linear / Softplus / weight_norm backward nodes and their double backward.
 * The forward is ls2fm_sdf_eval (same values; nothing is kept from it).  Upstreams (any may be NULL = zeros, not all):
 * d_sdf [n], d_feat [n,17] (raw MLP outputs; column 0 is f0, sdf = +-f0 / scale_mlp), d_normal [n,3].
 * grads: sdf_table (overwritten in full), sdf_mlp[0..1] (weight_v / weight_g / bias, overwritten); other members ignored.
 * d_p [n,3] (overwritten) or NULL.  No background-sphere min (LS2FM_ERR_UNSUPPORTED: general composed form).
 * workspace: ls2fm_sdf_points_workspace_bytes(...) bytes, scratch for this call only.
 */
int64_t ls2fm_sdf_points_workspace_bytes(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, int64_t n_points);
int ls2fm_sdf_points_bwd(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                         const float* p, int64_t n, const float* d_sdf, const float* d_feat, const float* d_normal,
                         const ls2fm_param_grads* grads, float* d_p, void* workspace, void* stream);
/* The same, as a SECOND gradient producer of a backward pass: the parameter gradients are ADDED to what `grads` already holds
 * (written earlier on the same stream, e.g. by ls2fm_render_bwd) -- the table through the scatter's add mode (nothing is zeroed,
 * slabs without items are not touched), the MLP tensors through the weight-norm backward's.  d_p is overwritten as above.  What
 * autograd otherwise does with two dense 50 MB table gradients and seven small tensors per extra node: a sum kernel each. */
int ls2fm_sdf_points_bwd_add(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                             const float* p, int64_t n, const float* d_sdf, const float* d_feat, const float* d_normal,
                             const ls2fm_param_grads* grads, float* d_p, void* workspace, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused volumetric rendering, forward.
 * Replaces: Renderer.forward (models/Renderer.py:51-116) and everything it calls: ray/AABB near-far
 * (Renderer.py:178), uniform mid-point sampling (Renderer.py:118-127), p = c + d t (utils/camera.py:
 * 262-266), SDF.infer_sdf + SDF.gradient (SDF.py:55-78, 102-114), RadF.Geometry_feat / infer_embed_v /
 * infer_app (RadF.py:66-86), SDF.sdf_to_sigma (SDF.py:84-87), Renderer.composite (Renderer.py:33-49)
 * and the background / depth / normal epilogue (Renderer.py:89-107).
 * center, ray [n_rays,3].  Outputs: rgb [n_rays,3], sdfs_volume [n_rays,N], normals [n_rays,N,3],
 * depth_mlp [n_rays], normal_mlp [n_rays,3].
 * workspace: ls2fm_render_workspace_bytes(...) bytes; its contents are consumed by ls2fm_render_bwd
 * for the same inputs, so it must be kept untouched between the two calls (ls2fm_render_bwd may run more than once on it).
 * The hash-table gradient outputs of ls2fm_render_bwd must be 16-byte aligned (float4 stores).
 */
#define LS2FM_MAX_RENDER_POINTS (1 << 23)   /* n_rays * n_samples per call (32-bit element / byte offsets and item counts
                                               inside); more: LS2FM_ERR_UNSUPPORTED from the size query, ls2fm_render_fwd
                                               and ls2fm_render_bwd -- split the rays over several calls */
int64_t ls2fm_render_workspace_bytes(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, int64_t n_rays);
/* Dual field, optional: write both hash tables (same geometry, [n_entries][2] each) entry-interleaved into
 * dual_table [n_entries][4] for ls2fm_params.dual_table.  The forward then gathers ONE 16-byte entry per corner for both
 * encodings instead of two 8-byte ones (the gathers are request-rate bound).  The caller owns the
 * buffer and refreshes it whenever either table changed (the host side keys it on the parameters' version counters). */
int ls2fm_interleave_tables(const float* sdf_table, const float* rad_table, int64_t n_entries, float* dual_table,
                            void* stream);
/* Optional extras of one render call (HOST struct; NULL = a plain render that a backward may follow).
 *   inference_only  forward: non-zero = no ls2fm_render_bwd will follow on this workspace.  Otherwise the forward's gather
 *                   pass also counts the items of the backward's table-gradient scatter (the corner cells are known there
 *                   anyway) and scans them beside shade_fwd, so the backward starts its scatter without a counting pass.
 *   n_level_groups, group_events   see the struct.
 *   loss            the loss head of the stage loops evaluated INSIDE the render (SURVEY.md section 8f row 1 "fuse as an
 *                   epilogue of the render kernel"; replaces pipelines/Camera.py:515-537 + BA.py:206-218 exactly as
 *                   ls2fm_loss_head_fwd/bwd below do, same tensors, same `terms` / `sums` layout): the forward's last kernel
 *                   forms the per-ray partial sums while rgb / normals / depth are still in registers and a one-workgroup
 *                   reduction writes `terms` and `sums`; the backward forms d rgb / d normals / d depth per sample from
 *                   `sums` (the counts), `weights` and the scalar upstreams `d_terms` / `d_total` in its first kernel -- no
 *                   [n_rays, N, 3] gradient tensor, no loss kernels between forward and backward.  Explicit upstreams
 *                   (d_rgb ... of ls2fm_render_bwd) are added on top.  A sharded run may all-reduce `sums` between the two
 *                   calls (and refresh `terms` with ls2fm_loss_terms_from_sums) for world-size-invariant means.
 */
typedef struct ls2fm_loss_spec {
    const float* rgb_gt;          /* [n_rays,3] */
    const float* depth_ref;       /* [n_rays] sphere-traced depth, or NULL = no depth-consistency term */
    const uint8_t* mask_eik;      /* [n_rays] or NULL = every ray: rays whose samples enter the eikonal mean */
    const uint8_t* mask_dc;       /* [n_rays] or NULL: mask_finish */
    const uint8_t* mask_mse;      /* [n_rays] or NULL: mask_bg */
    const float* weights;         /* DEVICE float[3] = 10^w of (rgb, eikonal, DC) */
    float* terms;                 /* DEVICE float[8]: forward output (layout of ls2fm_loss_head_fwd) */
    double* sums;                 /* DEVICE double[8]: forward output, backward input */
    const float* d_terms;         /* backward: DEVICE float[5] upstream of terms[0..4], or NULL = zeros */
    const float* d_total;         /* backward: DEVICE float[1] additional upstream of the weighted total, or NULL */
    float* d_depth_ref;           /* backward output: [n_rays] gradient w.r.t. depth_ref (overwritten), or NULL */
    uint32_t flags;               /* LS2FM_LOSS_*_FROM_GT */
    uint32_t count_scale;         /* ABI 9 (in the struct's former padding: size and offsets unchanged).  > 1: every COUNT of `sums` is
                                     multiplied by it in the forward's reduction -- a data-parallel run whose ranks hold the same number
                                     of rays and no masks divides by the GLOBAL counts without a collective or a kernel between forward and
                                     backward (ls2fm.dist "uniform"); 0 / 1: the local counts */
} ls2fm_loss_spec;
/* mask_eik / mask_mse := CameraSet.render's mask_bg, 0.05 < mean(rgb_gt[r]) < 0.95 (Camera.py:515), evaluated in the kernels
 * from rgb_gt; the pointer of the same name is ignored.  With these the loss head's only inputs that come out of a sphere
 * tracing are depth_ref and mask_dc (see ls2fm_render_opts.loss_inputs_ready). */
#define LS2FM_LOSS_EIK_FROM_GT 1u
#define LS2FM_LOSS_MSE_FROM_GT 2u

/* The backward of the sphere tracing that produced a render's depth_ref, run BY ls2fm_render_bwd and merged into its own
 * chains.  On an internal stream of its own, forked behind the call's first kernel (which writes d_depth_ref): the masked
 * upstream of the track points (ls2fm_trace_depth_bwd) and the front stages of ls2fm_sdf_points_bwd over them (gather pass,
 * per-point backward, scans, payload sort).  Then the per-point rows are contracted by the render's own SDF weight-gradient
 * kernel as extra tiles, and the points' table gradient is added into the render's sdf_table gradient behind its scatter: the
 * render's `grads` come out as the SUM of both producers -- no second set of gradient tensors, no sum kernel.  Replaces what
 * autograd does for Camera.py:506-523's d_consistent term through SDF.sphere_tracing's differentiable tail (SDF.py:201-214): a
 * second backward chain and seven accumulation kernels.  Not combined with n_level_groups > 1 (LS2FM_ERR_UNSUPPORTED). */
typedef struct ls2fm_depth_backward {
    const float* points;          /* [n_rays * k_max, 3] the tracing's track points (ls2fm_sphere_trace) */
    const int32_t* trips;         /* DEVICE int32[1] */
    const uint8_t* gate;          /* [n_rays] from ls2fm_trace_depth_fwd */
    int32_t k_max;
    float* d_sdf;                 /* scratch [n_rays * k_max] */
    void* workspace;              /* ls2fm_sdf_points_workspace_bytes(field, grid, n_rays * k_max) bytes */
} ls2fm_depth_backward;

#define LS2FM_MAX_LEVEL_GROUPS 4
typedef struct ls2fm_render_opts {
    int32_t inference_only;
    const ls2fm_loss_spec* loss;  /* NULL: no fused loss head */
    /* backward, multi-GPU runs: scatter the table gradients in n_level_groups (2..4) groups of consecutive levels -- group g
     * holds levels [L g / n, L (g + 1) / n) -- and record group_events[g] (hipEvent_t, may be NULL) on `stream` once group g's
     * slices of the gradient tables are final, so that their all-reduce overlaps with the scatter of the later groups.
     * 0 / 1: one pass, no events. */
    int32_t n_level_groups;
    void* group_events[LS2FM_MAX_LEVEL_GROUPS];
    /* forward with a loss head: hipEvent_t (or NULL) after which depth_ref and mask_dc -- the outputs of the sphere tracing of
     * the same rays -- are final (mask_eik / mask_mse, when given as pointers, must be final in stream order at the call).  The
     * forward waits for it on `stream` only in front of the kernel that reads them, the loss reduction BEHIND its gather pass
     * and its shading kernel (which leaves the ray's depth for the reduction to form the depth-consistency term): a caller that
     * traces on another stream overlaps the whole tracing chain (latency bound, few workgroups) with both. */
    void* loss_inputs_ready;
    /* backward with a loss head: hipEvent_t (or NULL) RECORDED on `stream` as soon as the backward's first kernel is enqueued --
     * from then on d_depth_ref (the gradient w.r.t. the traced depth) is final in stream order.  A caller whose depth came from a
     * differentiable sphere tracing starts that tracing's own backward on another stream behind this event, i.e. beside the
     * table scatter and the weight-gradient chain of this call instead of behind them. */
    void* depth_grad_ready;
    /* backward with a loss head whose d_depth_ref is set: also run the tracing's backward (see ls2fm_depth_backward), or NULL */
    const ls2fm_depth_backward* depth_bwd;
} ls2fm_render_opts;

int ls2fm_render_fwd(const ls2fm_field_desc* field, const ls2fm_grid_desc* sdf_grid,
                     const ls2fm_grid_desc* rad_grid, const ls2fm_params* params, const float* center,
                     const float* ray, int64_t n_rays, float* rgb, float* sdfs_volume, float* normals,
                     float* depth_mlp, float* normal_mlp, void* workspace, const ls2fm_render_opts* opts, void* stream);

/* Fused backward of ls2fm_render_fwd, including the analytic double backward of the normal path
 * (normals feed the radiance decoder, normal_mlp and the eikonal loss).
 * Upstream gradients (any may be NULL = zero): d_rgb [n_rays,3], d_sdfs_volume [n_rays,N],
 * d_normals [n_rays,N,3], d_depth_mlp [n_rays], d_normal_mlp [n_rays,3].
 * Parameter gradients go to `grads` in the reference's own parametrisation (weight_v/weight_g/bias,
 * beta).  d_center / d_ray [n_rays,3] (overwritten) may be NULL; when requested they carry the full
 * pose gradient (near/far are constants, exactly as in the reference: SURVEY.md 8a row a1).
 */
int ls2fm_render_bwd(const ls2fm_field_desc* field, const ls2fm_grid_desc* sdf_grid,
                     const ls2fm_grid_desc* rad_grid, const ls2fm_params* params, const float* center,
                     const float* ray, int64_t n_rays, const float* d_rgb, const float* d_sdfs_volume,
                     const float* d_normals, const float* d_depth_mlp, const float* d_normal_mlp,
                     const ls2fm_param_grads* grads, float* d_center, float* d_ray, void* workspace,
                     const ls2fm_render_opts* opts, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Bidirectional sphere tracing, the no-grad root-find loop.
 * Replaces: the `while True` of SDF.sphere_tracing (models/SDF.py:149-200) including the AABB
 * near/far (SDF.py:120-122).  Each ray is advanced independently for up to iters_max trips and
 * records, per trip k, its pre-update start point (the reference's pts_track) and its far-end
 * distance; `trips` receives the GLOBAL trip count K of the reference's loop (min(iters_max, the
 * trip at which no start ray is unfinished)), which the caller uses to truncate the track.
 * Outputs: near, far [n_rays]; track [n_rays, iters_max+1, 3]; t_end [n_rays, iters_max+1];
 * trips int32[1]; track_sdf [n_rays, iters_max+1] or NULL: the field's value at every track point
 * -- what the re-evaluation of the track (SDF.py:203) returns, bit for bit (the loop has it for the
 * points it evaluated; a finished start end that still moves with its stale step is evaluated for
 * this output only).  The differentiable use of the track (SDF.py:203-214) is the caller's, through
 * the autograd-capable ops above.
 */
int ls2fm_sphere_trace(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                       const float* ray0, const float* ray_dir, int64_t n_rays, float sdf_threshold,
                       int32_t iters_max, float* near, float* far, float* track, float* t_end, float* track_sdf,
                       int32_t* trips, void* workspace, void* stream);

/* The two halves of ls2fm_sphere_trace for a caller that overlaps the tracing with other work: ls2fm_sdf_prepare packs the SDF
 * MLP's weights into `workspace` (weight-norm -> effective weights; one latency-bound workgroup, ~15 us) and zeroes *zero_word
 * (the tracing's `trips`; may be NULL); ls2fm_sphere_trace_prepared is ls2fm_sphere_trace without those two steps -- on any
 * stream ordered behind the prepare.  ls2fm.stage prepares on the step's stream and traces on another one beside the render's
 * gather pass (a one-workgroup prepare launched BESIDE that pass is starved: 15 -> 90 us, measured). */
int ls2fm_sdf_prepare(const ls2fm_grid_desc* grid, const ls2fm_params* params, void* workspace, int32_t* zero_word, void* stream);
int ls2fm_sphere_trace_prepared(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                                const float* ray0, const float* ray_dir, int64_t n_rays, float sdf_threshold,
                                int32_t iters_max, float* near, float* far, float* track, float* t_end, float* track_sdf,
                                int32_t* trips, void* workspace, void* stream);

/* ls2fm_sdf_eval (sdf only) with the packed weights a preceding ls2fm_sdf_eval / ls2fm_sphere_trace call left in `workspace`
 * (same stream, same parameters): no weight preparation launch.  Used between a tracing call and the evaluation of its track. */
int ls2fm_sdf_eval_prepared(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                            const float* p, int64_t n, float* sdf, const void* workspace, void* stream);

/* The tail of SDF.sphere_tracing for a ray batch (models/SDF.py:203-214) and the mask lines of CameraSet.render
 * (pipelines/Camera.py:515-516), with the trip count K read from the device (what ls2fm_sphere_trace left in `trips`): a
 * captured stage step never returns to the host.  sdf_tracks [n_rays][k_max] = the graph-enabled SDF values of the track points
 * (ls2fm_sdf_eval on `track`), k_max = iters_max.
 *   d_pred [n_rays]   = near + sum_{k < max(K,1)} sdf_tracks[r][k], replaced by far where it exceeds it
 *   sdf_last [n_rays] = sdf_tracks[r][max(K,1) - 1];   finish [n_rays] (u8) = |sdf_last| < finish_threshold
 *   rgb_gt [n_rays][3] (optional): mask_bg (u8) = bg_lo < mean(rgb_gt[r]) < bg_hi;  mask_dc (u8) = finish & mask_bg
 *   gate [n_rays] (u8): 1 where d_pred was not clamped (the backward's pass-through mask)
 * _bwd: d_sdf_tracks [n_rays][k_max] = (k < max(K,1) && gate[r]) * d_dpred[r] + (k == max(K,1) - 1) * d_sdf_last[r]; either
 * upstream may be NULL. */
int ls2fm_trace_depth_fwd(const float* sdf_tracks, const int32_t* trips, const float* near, const float* far, int64_t n_rays,
                          int32_t k_max, float finish_threshold, const float* rgb_gt, float bg_lo, float bg_hi, float* d_pred,
                          float* sdf_last, uint8_t* finish, uint8_t* mask_bg, uint8_t* mask_dc, uint8_t* gate, void* stream);
int ls2fm_trace_depth_bwd(const float* d_dpred, const float* d_sdf_last, const int32_t* trips, const uint8_t* gate,
                          int64_t n_rays, int32_t k_max, float* d_sdf_tracks, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The re-projection term of a bundle-adjustment iteration and its gradient.
 * Replaces: the per-observation block of BA.run_ba (pipelines/BA.py:126-147) -- world2cam / cam2img of the surface-projected
 * tracked points through the live poses, the pixel error against their key points, the on-surface / finite mask, and the
 * robust mean of BA.compute_loss (BA.py:199-202) -- ~100 launch-bound PyTorch kernels per iteration with their autograd.
 *   points [n,3] (world); poses [n_views,3,4] world-to-camera (DEVICE); observations sorted by view: view v owns
 *   [view_start[v], view_start[v+1]) (DEVICE int32 [n_views+1]); intrinsic: HOST float[9], row-major K; obs_uv [n,2];
 *   sdf [n] or NULL: an observation counts when |sdf| < sdf_bound and its projection is not infinite.
 *   fwd -> err [n] (0 where not counted), on uint8 [n], sums DEVICE double[4] = {sum robust, sum err, count, reproj} with
 *   robust = 2 log(1 + err^2 / 4), reproj = 0.5 mean(robust) + 0.5 mean(err) over the counted ones (0 if none).
 *   bwd: d_reproj DEVICE float[1] -> d_points [n,3], d_poses [n_views,3,4] (overwritten; fixed-order sums, no atomics).
 *   workspace: ls2fm_reproject_workspace_bytes(n_views).
 */
int64_t ls2fm_reproject_workspace_bytes(int32_t n_views);
int ls2fm_reproject_fwd(const float* points, const float* poses, const int32_t* view_start, int32_t n_views,
                        const float* intrinsic, const float* obs_uv, const float* sdf, float sdf_bound, int64_t n,
                        float* err, uint8_t* on, double* sums, void* workspace, void* stream);
int ls2fm_reproject_bwd(const float* points, const float* poses, const int32_t* view_start, int32_t n_views,
                        const float* intrinsic, const float* obs_uv, const float* sdf, float sdf_bound, int64_t n,
                        const double* sums, const float* d_reproj, float* d_points, float* d_poses, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Loss head over the renderer's outputs: the scalar terms the reference's stages form
 * right after Renderer.forward, in one kernel each way instead of ~35 launch-bound PyTorch kernels.
 * Replaces: rgb L1 `l1_loss(rgb, rgbs_gt)` (pipelines/Camera.py:535); eikonal `l1_loss(norm(normals[mask]), 1)`
 * (Initialization.py:257-258, BA.py:193-194); depth consistency `smooth_l1_loss(d_points[mask_finish],
 * depth_mlp[mask_finish])`, 0 when the mask is empty (Camera.py:520-532); masked MSE behind PSNR (Camera.py:533);
 * the 10^w weighted sum of summarize_loss (BA.py:206-218).
 *   rgb, rgb_gt [n_rays,3]; normals [n_rays,n_samples,3]; depth [n_rays] (depth_mlp); depth_ref [n_rays] or NULL
 *   (sphere-traced d_points; NULL = no DC term); mask_* uint8 [n_rays] or NULL (= every ray): mask_eik selects the rays
 *   whose samples enter the eikonal mean, mask_dc = mask_finish, mask_mse = mask_bg.
 *   weights float[3] (DEVICE) = 10^w of (rgb, eikonal, DC).
 *   terms float[8] (DEVICE, overwritten) = {rgb L1 mean, eikonal mean, DC mean, masked MSE, weighted total, total again,
 *   PSNR = -10 log10(masked MSE) (Camera.py:534), 0}.
 *   sums double[8] (DEVICE, overwritten) = {S|rgb-gt|, n, S| |n|-1 |, n, S smooth_l1, n, S (rgb-gt)^2, n}: kept by the
 *   caller for the backward, and what a sharded run all-reduces for global normalisation.  Deterministic (fixed-order
 *   fp64 partials).  The workspace (ls2fm_loss_head_workspace_bytes) is zero-filled ONCE by the caller and reusable.
 * ls2fm_loss_head_bwd: d_terms float[5] = upstream gradient of `terms`, d_total float[1] = an additional upstream gradient
 * of the weighted total alone (either may be NULL = zeros, not both); writes (overwrites) d_rgb [n_rays,3],
 * d_normals [n_rays,n_samples,3], d_depth [n_rays] and, if non-NULL, d_depth_ref [n_rays].  Needs the `sums` of the
 * matching forward.
 */
int64_t ls2fm_loss_head_workspace_bytes(void);
int ls2fm_loss_head_fwd(const float* rgb, const float* rgb_gt, const float* normals, const float* depth,
                        const float* depth_ref, const uint8_t* mask_eik, const uint8_t* mask_dc, const uint8_t* mask_mse,
                        int64_t n_rays, int32_t n_samples, const float* weights, float* terms, double* sums, void* workspace,
                        void* stream);
int ls2fm_loss_head_bwd(const float* rgb, const float* rgb_gt, const float* normals, const float* depth,
                        const float* depth_ref, const uint8_t* mask_eik, const uint8_t* mask_dc, const uint8_t* mask_mse,
                        int64_t n_rays, int32_t n_samples, const float* weights, const float* d_terms, const float* d_total,
                        float* d_rgb, float* d_normals, float* d_depth, float* d_depth_ref, const double* sums, void* stream);
/* terms float[8] (DEVICE, overwritten) from sums double[8] (DEVICE) -- e.g. after a sharded run all-reduced the sums. */
int ls2fm_loss_terms_from_sums(const double* sums, const float* weights, float* terms, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused Adam step over a list of fp32 tensors, one launch, one pass over memory.
 * Replaces: torch.optim.Adam.step() of the stage loops (Initialization.py:149-179, BA.py:117-182; amsgrad = False,
 * maximize = False), operation by operation:  g' = g + wd p;  m = lerp(m, g', 1 - beta1);
 * v = v beta2 + (1 - beta2) g' g';  p -= (lr / (1 - beta1^step)) m / (sqrt(v) / sqrt(1 - beta2^step) + eps).
 * params / grads / exp_avg / exp_avg_sq: HOST arrays of n_tensors device pointers; numel: HOST array; step >= 1 is the
 * count AFTER this update (torch increments before use).  Updates params, exp_avg, exp_avg_sq in place.
 */
int ls2fm_adam_step(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int64_t step, void* stream);

/* The same update with the schedule RESIDENT ON THE DEVICE, for optimisation steps captured into a hipGraph (a replay cannot
 * take a new learning rate or step count from the host).  sched_state: DEVICE, 32 bytes {double step, lr, gamma; float
 * step_size, bc2_sqrt} -- the caller initialises step = 0 (updates done so far), lr = the rate of the NEXT update and gamma =
 * ExponentialLR's factor (utils of the stage loops: BA.py:87-88, gamma = (lr_end / lr)^(1 / max_iter)); every call first
 * advances it (step += 1, bias corrections from the new step, lr *= gamma afterwards: optimizer.step() then scheduler.step())
 * in a one-thread kernel, then runs the update with those values.  Replaces torch.optim.Adam.step() + ExponentialLR.step().
 */
int ls2fm_adam_step_scheduled(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const int64_t* numel, void* sched_state, float beta1, float beta2,
                              float eps, float weight_decay, void* stream);

/* Either update above with a MIRROR list: mirrors[t] (HOST array of n_tensors device pointers, entries may be NULL; NULL array =
 * none) is a second destination of tensor t's updated values in the entry-interleaved layout of ls2fm_interleave_tables /
 * ls2fm_params.dual_table -- element k goes to mirrors[t][(k >> 1) * 4 + (k & 1)], i.e. pass dual_table for the SDF table and
 * dual_table + 2 for the second field's table (8-byte aligned, even numel).  The render's gather pass reads that copy (one
 * 16-byte gather per corner serves both grids); writing it here, in the optimizer's one pass over the tables, keeps it current
 * without a rebuild per step.  sched_state NULL: the unscheduled form (lr and step from the arguments); otherwise the scheduled
 * form (lr and step ignored).  No reference counterpart: the reference's tcnn grids are separate.
 */
int ls2fm_adam_step_mirrored(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                             float* const* exp_avg_sq, const int64_t* numel, float* const* mirrors, void* sched_state, float lr,
                             float beta1, float beta2, float eps, float weight_decay, int64_t step, void* stream);

/* The general form (the three above are special cases of it): every parameter GROUP of an optimizer in one call.  Per tensor t a
 * learning rate lrs[t] (HOST float array; with `step`, the unscheduled form) or a device-resident schedule sched_states[t] (HOST
 * array of DEVICE pointers, entries or the whole array may be NULL; every distinct schedule is advanced once per call);
 * mirrors as above or NULL.  Two tensors whose mirrors are the two halves of one interleaved copy (M and M + 2, equal length,
 * 16-byte aligned) -- the SDF table and the second field's table -- are updated by ONE job that also writes the copy as whole
 * 16-byte entries (full cache lines; two separate jobs would each write 8 bytes of every 16).  Replaces torch.optim.Adam.step()
 * over `[{sdf_func.parameters(), lr_sdf}, {color_func.parameters(), lr_color}]` (BA.py:79-83) + ExponentialLR.step().
 */
int ls2fm_adam_step_multi(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                          float* const* exp_avg_sq, const int64_t* numel, float* const* mirrors, const float* lrs,
                          void* const* sched_states, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                          void* stream);

/* ExponentialLR.step() for parameter groups that took NO Adam step this iteration (every tensor's grad is None): the device
 * schedules' learning rates decay, `lr *= gamma`, their Adam step counts do not move -- torch's scheduler decays every group at
 * every step whether or not its tensors had gradients.  sched_states: HOST array of n DEVICE pointers (32-byte states as above).
 */
int ls2fm_adam_sched_decay(int32_t n, void* const* sched_states, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Camera rays: the pixels' camera centers and (unnormalised) ray directions through n_views pinhole cameras, one launch --
 * Camera.get_pts3D's ray construction and CameraSet.render's ray pick (pipelines/Camera.py:129-133, 457-463 on
 * utils/camera.py:230-252: get_center_and_ray = img2cam + cam2world) with, optionally, the se(3) -> SE(3) exponential map of
 * the live pose parameters in front (utils/camera.py:63-147: 11-term Taylor series, as BA.py:150-151 evaluates it every
 * iteration).  No gradients (the reference detaches the render poses, BA.py:150-151); same operation order as the torch code.
 *   poses [n_views,3,4] world-to-camera (device)  XOR  se3 [n_views,6] (w, u) (device);   kinv_host: the 3 x 3 K^-1 as 9 HOST floats
 *   pixels: xy [n,2] (or [n_views,n,2] with xy_per_view) device float coordinates  XOR  pix [n] int64 pixel indices of a
 *   width-`width` image (centres x + 0.5, y + 0.5: the same pixels in every view)
 *   view_sel: NULL, or a DEVICE int64[1]: only that view is processed (outputs have ONE view) -- a captured step follows the value
 *   centers, rays [n_views or 1, n, 3] (overwritten); poses_out [n_views or 1,3,4] or NULL: the poses used (se3 given: the exponentials)
 */
/* The exponential map alone, with its gradient: poses [n,3,4] = exp(se3 [n,6]) and d_se3 [n,6] from d_poses [n,3,4] (the chain
 * rule through the truncated series, term by term, as autograd differentiates utils/camera.py:63-147) -- the live poses of a
 * bundle-adjustment iteration that the re-projection term reads (BA.py:126-147), one launch each way instead of ~90. */
int ls2fm_se3_exp_fwd(const float* se3, int32_t n, float* poses, void* stream);
int ls2fm_se3_exp_bwd(const float* se3, const float* d_poses, int32_t n, float* d_se3, void* stream);
int ls2fm_camera_rays(const float* poses, const float* se3, const float* kinv_host, const float* xy, const int64_t* pix,
                      int32_t width, int32_t xy_per_view, const int64_t* view_sel, int32_t n_views, int64_t n, float* centers,
                      float* rays, float* poses_out, void* stream);

/* The tracing-consistency term of a stage-loop iteration -- the traced key points' surface points against their tracked 3-D
 * points (pipelines/Camera.py:108-143 get_pts3D and the callers' loss lines, e.g. BA.py:155-161) -- as one launch each way:
 *   surface_i = center_i + ray_i d_i ;  w_i = live_i / sum(live) ;  out[0] = sum_i |target_i - surface_i| w_i ;
 *   out[1] = sum_i |sdf_last_i| w_i (0 when sdf_last is NULL) ;  out[2] = sum(live)            (fp64 fixed-order sums)
 * center, ray, target [n,3], d, live, sdf_last [n] (device); out [3].  bwd: g_tl, g_sd: one float each, the upstreams of out[0]
 * and out[1] (device; NULL = that term has no upstream: zero) -> d_d [n], d_sdf [n] (NULL with sdf_last NULL); d |e| / d d at
 * e = 0 is 0, sign(0) = 0, as torch. */
int ls2fm_tracing_term_fwd(const float* center, const float* ray, const float* d, const float* target, const float* live,
                           const float* sdf_last, int64_t n, float* out, void* stream);
int ls2fm_tracing_term_bwd(const float* center, const float* ray, const float* d, const float* target, const float* live,
                           const float* sdf_last, int64_t n, const float* out, const float* g_tl, const float* g_sd, float* d_d,
                           float* d_sdf, void* stream);

/* The loss lines of a bundle-adjustment iteration outside the render (pipelines/BA.py:160-170: `sdf_surf`, the adaptive weight
 * 10^1 of the re-projection error above 10 px, the weighted sum) as one launch each way:
 *   out_surf = mean |sdfs| ;  out_w = reproj > thresh ? w_hi : w_lo  (from the error's VALUE: no gradient through the choice) ;
 *   out_extra = out_w reproj + w_surf out_surf + w_add add        (add: one more device scalar, e.g. the tracing loss; may be NULL)
 * reproj, add, the three outputs: one float each (device); sdfs [n].  bwd: g [1] upstream of out_extra ->
 * d_reproj [1] = g out_w, d_sdfs [n] = g w_surf sign(sdfs) / n (sign(0) = 0, as torch), d_add [1] = g w_add (NULL with add NULL). */
int ls2fm_ba_terms_fwd(const float* reproj, const float* sdfs, int64_t n, const float* add, float thresh, float w_lo, float w_hi,
                       float w_surf, float w_add, float* out_surf, float* out_w, float* out_extra, void* stream);
int ls2fm_ba_terms_bwd(const float* sdfs, int64_t n, const float* w_reproj, const float* g, float w_surf, float w_add,
                       float* d_reproj, float* d_sdfs, float* d_add, void* stream);

/* out[0] = wa a[0] + wb b[0] of two device scalars (the loops' weighted sums of two loss terms, e.g. Initialization.py:252-255);
 * bwd: d2[0] = g[0] wa, d2[1] = g[0] wb (one two-float buffer).  One launch each way instead of five elementwise kernels. */
int ls2fm_weighted_pair_fwd(const float* a, const float* b, float wa, float wb, float* out, void* stream);
int ls2fm_weighted_pair_bwd(const float* g, float wa, float wb, float* d2, void* stream);

/* SDF.get_surface_pts' projection line (models/SDF.py:104-110):  out = p - normals / |normals|.detach() * sdf ,  length = |normals|
 * (p, normals, out [n,3]; sdf, length [n]) and its gradient w.r.t. normals and sdf (d p = g_out, the caller's); g_out / g_length
 * may be NULL (no upstream for that output).  One launch each way instead of ~14 elementwise kernels. */
int ls2fm_surface_pts_fwd(const float* p, const float* normals, const float* sdf, int64_t n, float* out, float* length, void* stream);
int ls2fm_surface_pts_bwd(const float* normals, const float* sdf, const float* length, int64_t n, const float* g_out,
                          const float* g_length, float* d_normals, float* d_sdf, void* stream);

/* The explicit-match terms of the two-view initialisation (pipelines/Initialization.py:154-160, 252-255; Camera.py:136, 168-178):
 * n_seg <= 4 segments of n traced key points each (segment = source view); pts = center + ray d is projected through
 * poses[seg] (the OTHER view's world-to-camera [3,4], device, no gradient) and K and compared with uv_obs:
 *   out[0] = mean || uv(pts) - uv_obs ||   out[1] = mean |sdf_last|   over all n_seg * n points;  surface [n_seg * n, 3] = pts.
 * center, ray [n_seg * n, 3], uv_obs [n_seg * n, 2] (device); d, sdf_last, d_d, d_sdf: HOST arrays of n_seg DEVICE pointers ([n] each);
 * g [2] (device): upstream of out[0], out[1].  One launch each way instead of ~70 torch kernels. */
int ls2fm_match_term_fwd(const float* center, const float* ray, const float* uv_obs, const float* poses, const float* intrinsic_host,
                         int32_t n_seg, int64_t n, const float* const* d, const float* const* sdf_last, float* surface, float* out,
                         void* stream);
int ls2fm_match_term_bwd(const float* center, const float* ray, const float* uv_obs, const float* poses, const float* intrinsic_host,
                         int32_t n_seg, int64_t n, const float* const* d, const float* const* sdf_last, const float* g,
                         float* const* d_d, float* const* d_sdf, void* stream);

/* ---------------------------------------------------------------------------------------------
 * How the table-gradient scatter (inside ls2fm_render_bwd / ls2fm_sdf_points_bwd) finishes the few coarse levels whose slabs
 * are split over several workgroups.  Process-wide; takes effect for the calls enqueued afterwards.
 *   1 (default; env LS2FM_SCATTER_MODE)  every part's 64-bit fixed-point partial sums are combined by one more small launch:
 *      integer sums, so every table-gradient entry is the exactly rounded sum of its fp32 contributions whatever the order
 *      the hardware produced them in, and a backward repeats itself bit for bit
 *   0  the parts flush with float atomics into a zeroed range: one launch fewer, sums of rounded partials in arrival order
 *      (what rounds 1-3 shipped; kept for measurements)
 * tcnn's kernel_grid_backward is atomicAdd on fp32/half2 throughout (non-deterministic); the reference has no switch.
 */
int ls2fm_set_scatter_mode(int mode);
int ls2fm_get_scatter_mode(void);

/* ---------------------------------------------------------------------------------------------
 * Opt-in per-kernel timing (benchmarking aid; the library's only process-global state, off by default).
 * While enabled, every internal kernel launch of the calls above is bracketed by HIP events recorded on the
 * call's own stream; ls2fm_profile_get() synchronises those events and returns, per internal kernel, the
 * accumulated device time in milliseconds and the number of launches.  While enabled, the calls launch their kernels
 * serially on `stream` (the internal side-stream overlap is off) so that each span is the kernel's own duration.
 * Single-threaded use.
 */
int ls2fm_profile_enable(int on);
int ls2fm_profile_reset(void);
int ls2fm_profile_count(void);
const char* ls2fm_profile_name(int index);
int ls2fm_profile_get(int index, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* LS2FM_H */
