// Shared definitions of the fused render pipeline (render_fwd.hip, render_bwd.hip, sdf_eval.hip).
//
// Pipeline (one C-ABI call each way, several kernels inside):
//   forward : prep_weights -> ray_encode (grid 1, +Jacobian) [-> ray_encode (grid 2)] -> shade_fwd
//   backward: shade_bwd -> wgrad (MFMA GEMM over points) -> wgrad_reduce -> table scatter(s) -> finalize
// Per-sample intermediates live in HBM as structure-of-arrays [channel][P_pad] so that every
// "lane = sample" access is a fully coalesced 256-byte wave transaction.
#pragma once

#include "ls2fm_device.h"

// opt-in per-kernel timing (profile.hip): a mark before every launch, a mark with id -1 at the end of a call
enum ls2fm_prof_id {
    LS2FM_PROF_PREP = 0, LS2FM_PROF_ENCODE_SDF, LS2FM_PROF_ENCODE_RAD, LS2FM_PROF_SHADE_FWD, LS2FM_PROF_SHADE_BWD,
    LS2FM_PROF_WGRAD_GEO, LS2FM_PROF_WGRAD_TAIL, LS2FM_PROF_SCATTER_SDF, LS2FM_PROF_SCATTER_RAD, LS2FM_PROF_FINALIZE,
    LS2FM_PROF_SDF_EVAL, LS2FM_PROF_SPHERE_TRACE, LS2FM_PROF_BIN, LS2FM_PROF_LOSS_FWD,
    LS2FM_PROF_LOSS_BWD, LS2FM_PROF_WGRAD_MLP, LS2FM_PROF_POSE, LS2FM_PROF_ENCODE_PAIR, LS2FM_PROF_COUNT
};
bool ls2fm_prof_enabled();
void ls2fm_prof_begin(int id, hipStream_t stream);      // bracket one kernel launch on the stream it is enqueued on
void ls2fm_prof_end(int id, hipStream_t stream);

// internal fork/join (streams.hip)
struct SideCtx { hipStream_t side; hipEvent_t fork, mid, join; };
bool ls2fm_side_stream(SideCtx* out, hipStream_t caller);       // side stream + events of this caller stream
int ls2fm_join_on_error(bool forked, const SideCtx& sc, hipStream_t caller, int status);
// host-visible sticky error word of the asynchronous parts (ls2fm_async_error): device-writable pinned host memory, allocated at
// the first call outside a stream capture; null until then (kernels skip a null word)
int* ls2fm_async_error_word(hipStream_t stream);
int ls2fm_device_cus();

constexpr int kHidden = LS2FM_HIDDEN;     // 64
constexpr int kOut = LS2FM_FEAT + 1;      // 17: sdf + 16 features
constexpr int kInMax = 3 + 2 * LS2FM_MAX_LEVELS;   // 35
constexpr int kView = LS2FM_VIEW_ENC;     // 27
constexpr int kRadIn = 3 + 3 + kView + 2 * LS2FM_FEAT;   // 65 (49 without the second field)

// One record per hidden unit j (256 B, scalar-load friendly):
//   [0..34] W0[j][k] (zero beyond the live input width)   [35] b0[j]   [36..52] W1[o][j], o = 0..16
constexpr int kRecStride = 64;
constexpr int kRecB0 = 35;
constexpr int kRecW1 = 36;
constexpr int kMlpFloats = kHidden * kRecStride + 32;      // + b1[17] (padded)

// MFMA-operand-ordered copies of the geometry MLP weights for v_mfma_f32_16x16x4_f32 (lane = 16 g + jl supplies
// A[i = jl][k = g]).  Internal input order k': 0..31 encoding channels, 32..34 p / rescale, 35 constant 1 (bias column).
struct MfmaField {
    float w0a[4][9][64];       // layer 0 as A operand:            W0'[16m + jl][k' = 4t + g]          [m][t][lane]
    float w1a[4][4][64];       // layer 1 rows 1..16 as A operand: W1[1 + jl][16m + 4g + r]            [m][r][lane]
};
struct MfmaW {                 // sdf .. w10 is one contiguous block (staged into LDS by the shade kernels), then geo
    MfmaField sdf;
    float w0ta[3][4][4][64];   // SDF layer 0 transposed:          W0'[16m + 4g + r][k' = 16mk + jl]   [mk][m][r][lane], 0 for k' >= 35
    float w10[4][4][64];       // SDF W1[0][16m + 4g + r]                                              [m][r][lane]
    MfmaField geo;
    float b1a[2][4][64];       // b1[1 + 4g + r]                                                       [field][r][lane]
    float b10[4];              // b1[0] per field
    float w0tx_geo[4][4][64];  // second field's W0'[16m + 4g + r][k' = 32 + jl] (p / rescale rows; pose gradients) [m][r][lane]
};
// backward: per-field contiguous blocks (staged into LDS by shade_bwd)
struct MfmaBwdSdf {
    float w0a[4][9][64];       // as MfmaField::w0a
    float w1ta[4][5][64];      // layer 1 transposed as A operand: W1[o = 4t + g][16m + jl], 0 for o > 16    [m][t][lane]
    float w0ta[2][4][4][64];   // as MfmaW::w0ta, encoding rows only (mk < 2)
    float w10[4][4][64];
};
struct MfmaBwdGeo {
    float w0a[4][9][64];
    float w1ta[4][4][64];      // W1[o = 1 + 4t + g][16m + jl]                                              [m][t][lane]
    float w0ta[2][4][4][64];   // W0'[16m + 4g + r][k' = 16mk + jl]
};
constexpr int kMfmaBwdSdfFloats = sizeof(MfmaBwdSdf) / 4;   // 6656
constexpr int kMfmaBwdGeoFloats = sizeof(MfmaBwdGeo) / 4;   // 5376
constexpr int kMfmaFieldFloats = 4 * 9 * 64 + 4 * 4 * 64;                     // 3328
constexpr int kMfmaSdfFloats = kMfmaFieldFloats + 3 * 4 * 4 * 64 + 4 * 4 * 64;  // 7424

struct Packed {
    MfmaW mw;
    MfmaBwdSdf bs;
    MfmaBwdGeo bg;
    float sdf[kMlpFloats];
    float geo[kMlpFloats];
    float wc[3][68];        // collapsed radiance decoder: cols [0,3) p  [3,6) n  [6,33) view  [33,49) f  [49,65) f2
    float bc[4];
    float t1[3][64];        // R2 * R1 (effective weights)
    float r0[64][68];       // effective R0, rows padded
    float r1[64][64];
    float r2[3][64];
    float beta, alpha, pad0, pad1;
};

// scene constants in kernel-argument form
struct FieldC {
    float bmin[3], bmax[3];
    float box_c[3], box_h[3];     // (bmax+bmin)/2, (bmax-bmin)/2 in fp32, as the reference computes them
    float inv_ext[3];             // 1 / (bmax - bmin)
    float rescale, scale_mlp, kappa;   // kappa = +-1/scale_mlp : d sdf / d f0
    int inside;
    float bg[3];
    int n_samples;
};

static inline FieldC make_field_c(const ls2fm_field_desc* f) {
    FieldC c;
    for (int d = 0; d < 3; ++d) {
        c.bmin[d] = f->bound_min[d];
        c.bmax[d] = f->bound_max[d];
        c.box_c[d] = (f->bound_max[d] + f->bound_min[d]) / 2.0f;
        c.box_h[d] = (f->bound_max[d] - f->bound_min[d]) / 2.0f;
        c.inv_ext[d] = 1.0f / (f->bound_max[d] - f->bound_min[d]);
        c.bg[d] = f->bgcolor[d];
    }
    c.rescale = f->rescale;
    c.scale_mlp = f->scale_mlp;
    c.inside = f->inside;
    c.kappa = (f->inside ? 1.0f : -1.0f) / f->scale_mlp;
    c.n_samples = f->n_samples;
    return c;
}

// Sample n of ray r: AABB near/far (constants w.r.t. the pose), mid-point depth, world position and
// grid-normalised position -- each step the same IEEE operation the reference performs
// (Renderer.py:118-127, camera.py:262-266, base.py:35), so the hash cell of every sample is the oracle's.
struct RayGeom {
    float o[3], d[3];
    float t_near, t_far;
};

__device__ __forceinline__ RayGeom load_ray(const FieldC& fc, const float* __restrict__ center,
                                            const float* __restrict__ ray, int64_t r) {
    RayGeom g;
#pragma unroll
    for (int a = 0; a < 3; ++a) { g.o[a] = center[r * 3 + a]; g.d[a] = ray[r * 3 + a]; }
    bool hit;
    ray_box(g.o, g.d, fc.box_c, fc.box_h, g.t_near, g.t_far, hit);
    return g;
}

__device__ __forceinline__ float sample_depth(const RayGeom& g, int n, int n_samples) {
    return ((float)n + 0.5f) / (float)n_samples * (g.t_far - g.t_near) + g.t_near;
}

__device__ __forceinline__ void sample_position(const FieldC& fc, const RayGeom& g, float t, float p[3], float x[3]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        p[a] = g.o[a] + g.d[a] * t;
        x[a] = (p[a] - fc.bmin[a]) / (fc.bmax[a] - fc.bmin[a]);
    }
}

// VolSDF density sigma = alpha * Psi_beta(sdf) and its derivatives (SDF.py:84-87)
__device__ __forceinline__ float sigma_of(float s, float alpha, float beta) {
    const float e = 0.5f * expf(-fabsf(s) / beta);
    return alpha * (s >= 0.0f ? e : 1.0f - e);
}

// Geometry MLP forward with the weights streamed from the packed records (wave-uniform addresses ->
// scalar loads; every FMA is a VALU op with one SGPR operand).  u[35] -> f[17] and, optionally,
// r = W0^T (s' * w1_0)  (the vector whose contraction with d u / d p gives the analytic normal).
template <bool WANT_R>
__device__ __forceinline__ void geometry_forward(const float* __restrict__ rec, const float (&u)[kInMax],
                                                 float (&f)[kOut], float (&r)[kInMax]) {
#pragma unroll
    for (int o = 0; o < kOut; ++o) f[o] = rec[kHidden * kRecStride + o];
    if (WANT_R) {
#pragma unroll
        for (int k = 0; k < kInMax; ++k) r[k] = 0.0f;
    }
#pragma unroll 1
    for (int j = 0; j < kHidden; ++j) {
        const float* __restrict__ w = rec + j * kRecStride;
        float a0 = w[kRecB0], a1 = 0.0f;
#pragma unroll
        for (int k = 0; k + 1 < kInMax; k += 2) {
            a0 = fmaf(w[k], u[k], a0);
            a1 = fmaf(w[k + 1], u[k + 1], a1);
        }
        a0 = fmaf(w[kInMax - 1], u[kInMax - 1], a0);
        float h, s1, s2;
        softplus100(a0 + a1, h, s1, s2);
#pragma unroll
        for (int o = 0; o < kOut; ++o) f[o] = fmaf(w[kRecW1 + o], h, f[o]);
        if (WANT_R) {
            const float g = s1 * w[kRecW1];
#pragma unroll
            for (int k = 0; k < kInMax; ++k) r[k] = fmaf(w[k], g, r[k]);
        }
    }
}

// Fourier view embedding component c (0..26) of direction d: [d, sin(1 d), cos(1 d), sin(2 d), ...]
__device__ __forceinline__ float view_component(const float d[3], int c) {
    // (selects, not d[a] with a run-time a: a dynamically indexed ray record is put into scratch memory as a whole)
    const int q = c < 3 ? 0 : (c - 3) / 3, a = c < 3 ? c : (c - 3) % 3;       // q: 0 sin f0, 1 cos f0, 2 sin f1, ...
    const float da = a == 0 ? d[0] : (a == 1 ? d[1] : d[2]);
    if (c < 3) return da;
    const float f = (float)(1 << (q >> 1));
    const float x = da * f;
    return (q & 1) ? cosf(x) : sinf(x);
}

// The second field's grid must have the geometry of the SDF grid (it does: both are built from the same
// encoding config, models/RadF.py:35-39): the scatter's slab-test keys are computed once and shared.
static inline bool same_grid_geometry(const ls2fm_grid_desc* a, const ls2fm_grid_desc* b) {
    if (a->n_levels != b->n_levels) return false;
    for (int l = 0; l < a->n_levels; ++l)
        if (a->scale[l] != b->scale[l] || a->resolution[l] != b->resolution[l] || a->size[l] != b->size[l] ||
            a->hashed[l] != b->hashed[l])
            return false;
    return true;
}

int64_t ls2fm_bins_workspace_floats(int n_levels, int64_t n_points);
// leading levels of a dual-field render whose table-gradient items are explicit and run-merged (bin_items.h); 0: none
int ls2fm_explicit_levels(const ls2fm_grid_desc* grid, int dual, int n_samples);

// Workspace carve-up (offsets in floats).  Channels are SoA with stride p_pad.
struct WsLayout {
    int64_t p, p_pad, r_pad;
    int l1, l2, dual;
    // forward -> backward
    int64_t packed, e1, j1, e2, sdfv, nrm, rgbs, fe, fe2, rout, lpart;
    // backward scratch
    int64_t rec1, rec2, rpt, bins, v, p3, gf, dz, gf2, dzr, renc, dexyz, dlen, mpart, wg, dbeta, smax;
    int64_t total;
};

#ifndef LS2FM_SLAB_SHIFT
#define LS2FM_SLAB_SHIFT 13
#endif
constexpr int kSlabShift = LS2FM_SLAB_SHIFT;          // table-gradient scatter: 8192-entry slabs of one grid, or 4096-entry slabs of both
static inline int ls2fm_slab_shift(int dual) { return dual ? kSlabShift - 1 : kSlabShift; }
constexpr int kWgradMlpBlocks = 256;   // persistent workgroups of wgrad_mlp: ONE per CU -- alone the kernels are slower than with two
                                      // (63 + 43 vs 52 + 34 us), but they run beside scatter_fill / slab_accumulate and leave them
                                      // more of the CU: step 0.656 -> 0.638 ms (graph), 128 or 192 workgroups are slower again
// registers (= [R][64] rows) of one weight-gradient partial: dW0 tiles, dW1 tiles, dW1 row 0, db1
constexpr int kRegsSdf = 48 + 16 + 16 + 5;
constexpr int kRegsGeo = 48 + 16 + 4;
constexpr int kRegsDec = 20;                     // 5 output tiles of the decoder columns
// Round 5: the render's backward contracts the MLPs' weight gradients inside shade_bwd (shade_bwd.hip, 3.): one partial per
// RAY (= workgroup), summed in two fixed-order levels -- kL1Seg segment sums (wgrad_l1 blocks of the wgrad_dec launch), then
// wgrad_tail.
constexpr int kL1Seg = 16;
// partial buffers inside WsLayout::mpart (floats)
struct WgPartLayout {
    int64_t l1_sdf;      // [kL1Seg + kWgradMlpBlocks][kRegsSdf][64]: level-1 sums, then wgrad_mlp's own partials (point queries;
                         //                                             the traced depth's extra tiles of a stage step)
    int64_t l1_geo;      // [kL1Seg][kRegsGeo][64]   (point queries / round-4 form: [kWgradMlpBlocks] -- sized for the larger)
    int64_t dec;         // [kWgradMlpBlocks][kRegsDec][64]
    int64_t slot_sdf;    // [n_slots][kRegsSdf][64]
    int64_t slot_geo;    // [n_slots][kRegsGeo][64]
    int64_t total;
    int n_slots;
};
// the render's backward contracts the MLPs' weight gradients inside shade_bwd (one partial per ray) for rays of >= 64 samples unless
// LS2FM_FUSED_WGRAD=0 (round 4's separate wgrad_mlp launches, for A/B runs): ONE predicate for the kernels' choice and the workspace
static inline bool ls2fm_fused_wgrad(int n_samples) {
    static const int env = [] { const char* e = getenv("LS2FM_FUSED_WGRAD"); return e ? atoi(e) : 1; }();
    return env != 0 && n_samples >= 64;
}
static inline WgPartLayout make_wg_part_layout(int dual, int64_t n_rays, int n_samples) {
    WgPartLayout L;
    int64_t o = 0;
    // per-ray slots (39 KB per ray, dual field) only where shade_bwd fills them (round-5 advisor: short rays, free points and A/B runs
    // with the switch off carried them unused -- 2.5 GB at 64 k rays)
    L.n_slots = ls2fm_fused_wgrad(n_samples) ? (int)n_rays : 0;
    L.l1_sdf = o; o += (int64_t)(kL1Seg + kWgradMlpBlocks) * kRegsSdf * 64;
    L.l1_geo = o; o += dual ? (int64_t)kWgradMlpBlocks * kRegsGeo * 64 : 0;
    L.dec = o; o += (int64_t)kWgradMlpBlocks * kRegsDec * 64;
    L.slot_sdf = o; o += (int64_t)L.n_slots * kRegsSdf * 64;
    L.slot_geo = o; o += dual ? (int64_t)L.n_slots * kRegsGeo * 64 : 0;
    L.total = o;
    return L;
}

// reduced raw weight gradients (floats)
struct WgLayout {
    static constexpr int dW0 = 0;                     // [64][36]: p/rescale(3) e(32) 1
    static constexpr int dW1 = dW0 + 64 * 36;         // [17][65]: h(64) 1
    static constexpr int dW1r0 = dW1 + 17 * 65;       // [64]    : extra row-0 term (double backward)
    static constexpr int dG0 = dW1r0 + 64;            // [64][36]
    static constexpr int dG1 = dG0 + 64 * 36;         // [17][65]
    static constexpr int dWc = dG1 + 17 * 65;         // [3][39] : p(3) n(3) f(16) f2(16) 1
    static constexpr int dWv = dWc + 3 * 39;          // [3][27] : view embedding columns
    static constexpr int total = dWv + 3 * 27;
};

// behind the reduced block (rounded up to 64 floats, its slack holds wgrad_tail's ticket): one flag word per job of the in-fill
// side chain (side_jobs.h) -- zeroed with the block by shade_bwd's leading workgroups
constexpr int kSideFlagInts = 4096;
constexpr int kWgFlagsAt = (WgLayout::total + 63) / 64 * 64;
constexpr int kWgBlockFloats = kWgFlagsAt + kSideFlagInts;

// (rows of the per-sample SoA arrays are p_pad floats apart: at the benchmark's 131 072 samples a power of two, the 32 rows a wave of
// the shading kernels reads lie exactly 512 KB apart.  De-tuning the stride by 64 / 192 / 1088 floats was measured in round 6: no
// effect on any kernel -- the memory system hashes its channels.)
static inline WsLayout make_ws_layout(int64_t n_rays, int n_samples, int l1, int l2, int dual) {
    WsLayout w;
    w.p = n_rays * n_samples;
    w.p_pad = (w.p + 63) / 64 * 64;
    w.r_pad = (n_rays + 63) / 64 * 64;
    w.l1 = l1; w.l2 = l2; w.dual = dual;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += (n + 63) / 64 * 64; return at; };
    const int64_t P = w.p_pad;
    w.packed = take((sizeof(Packed) + 3) / 4);
    w.e1 = take(2 * l1 * P);
    w.j1 = take(6 * l1 * P);
    w.e2 = take(dual ? 2 * l2 * P : 0);
    w.sdfv = take(P);
    w.nrm = take(3 * P);
    w.rgbs = take(3 * P);
    w.fe = take(16 * P);
    w.fe2 = take(dual ? 16 * P : 0);
    w.rout = take(4 * w.r_pad);      // per-ray outputs kept for a fused loss head's backward: rgb (3), depth_mlp  [4][r_pad]
    w.lpart = take(4 * w.r_pad);     // fused loss head: per-ray partial sums  S|rgb-gt|, S| |n|-1 |, smooth_l1, S (rgb-gt)^2
    // scatter payload: per point {x y z | gn0 gn1 gn2 | - -} (32 B, 4 MB per 131072 points: L2-resident across all the
    // levels' slab workgroups) + per (level, point) {de0 de1 rr0 rr1} (SDF grid, 16 B) / {de0 de1} (second grid, 8 B)
    w.rpt = take(8 * P);
    w.rec1 = take(4 * (int64_t)l1 * P);
    w.rec2 = take(dual ? 2 * (int64_t)l2 * P : 0);
    // per-sample upstream vectors shade_bwd hands to the weight-gradient kernels (SoA)
    w.v = take(35 * P);
    w.p3 = take(3 * P);
    w.gf = take(17 * P);
    w.dz = take(3 * P);
    w.gf2 = take(dual ? 17 * P : 0);
    w.dexyz = take(6 * P);       // pose gradients: d L / d (p / rescale) of the SDF field (3) and of the second field (3)
    w.dlen = take(w.r_pad);      // pose gradients: d L / d |ray| through the interval lengths of the composite
    w.dzr = take(3 * w.r_pad);
    w.renc = take(27 * w.r_pad);
    w.mpart = take(make_wg_part_layout(dual, n_rays, n_samples).total);
    w.wg = take(kWgBlockFloats);     // reduced gradients + ticket words + the side jobs' flags (side_jobs.h): zeroed as one range
    w.dbeta = take(2 * w.r_pad);     // one double per ray: d L / d beta partials (summed in fixed order by finalize)
    // bin meta (counts first) directly after wg / dbeta: one memset zeroes all three (render_bwd.hip)
    w.bins = take(ls2fm_bins_workspace_floats(l1, w.p));   // per-slab item lists of the scatter (bin_scatter.hip)
    w.smax = take(32 * w.r_pad); // [32][r_pad]: per-ray bound of one scatter contribution per level ([0,16) SDF grid, [16,32) second grid)
    w.total = o;
    return w;
}

struct LevelScales { float s[LS2FM_MAX_LEVELS]; };

// fused loss head, backward side (ls2fm_loss_spec + the per-ray outputs the forward kept): the upstream of rgb / normals /
// depth is formed per sample from the counts in `sums`, the weights and the scalar upstreams
struct LossUp {
    const float* rgb_gt;           // null: no fused loss head
    const float* depth_ref;
    const uint8_t* mask_eik; const uint8_t* mask_dc; const uint8_t* mask_mse;
    const float* weights; const double* sums; const float* d_terms; const float* d_total;
    float* d_depth_ref;            // [R] output or null
    uint32_t flags;
};

struct Upstream {                  // dL/d(outputs of render_fwd); any pointer may be null (= zeros)
    const float* d_rgb;            // [R,3]
    const float* d_sdfs;           // [R,N]
    const float* d_normals;        // [R,N,3]
    const float* d_depth;          // [R]
    const float* d_nm;             // [R,3]
    LossUp loss;
};

// CameraSet.render's mask_bg of a ray (Camera.py:515), as trace_depth_fwd_kernel forms it
__device__ __forceinline__ bool ls2fm_bg_from_gt(const float* __restrict__ rgb_gt, int64_t r) {
    const float gray = (rgb_gt[3 * r] + rgb_gt[3 * r + 1] + rgb_gt[3 * r + 2]) / 3.0f;
    return gray < 0.95f && gray > 0.05f;
}
template <typename Spec>
__device__ __forceinline__ bool ls2fm_in_eik(const Spec& lo, int64_t r) {
    return (lo.flags & LS2FM_LOSS_EIK_FROM_GT) ? ls2fm_bg_from_gt(lo.rgb_gt, r) : (lo.mask_eik == nullptr || lo.mask_eik[r] != 0);
}
template <typename Spec>
__device__ __forceinline__ bool ls2fm_in_mse(const Spec& lo, int64_t r) {
    return (lo.flags & LS2FM_LOSS_MSE_FROM_GT) ? ls2fm_bg_from_gt(lo.rgb_gt, r) : (lo.mask_mse == nullptr || lo.mask_mse[r] != 0);
}
__device__ __forceinline__ float ls2fm_smooth_l1(float d) { const float a = fabsf(d); return a < 1.0f ? 0.5f * d * d : a - 0.5f; }
__device__ __forceinline__ float ls2fm_smooth_l1_grad(float d) { return fabsf(d) < 1.0f ? d : (d > 0.f ? 1.0f : -1.0f); }
__device__ __forceinline__ float ls2fm_sign(float d) { return d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.f); }

// bin_scatter.hip: the entry range [first, first + count) of the gradient table(s) that slab_accumulate flushes with atomics
void ls2fm_scatter_zero_range(const ls2fm_grid_desc* grid, int64_t n_points, bool dual, int64_t* first, int64_t* count);

// shade_bwd.hip
// fused_wgrad: the MLPs' weight gradients are contracted inside the kernel (one partial per workgroup slot in WgPartLayout's
// slot_sdf / slot_geo); 0: the v / gf / gf2 rows are stored for wgrad_mlp.hip (round-4 form, kept for A/B measurements)
int ls2fm_launch_shade_bwd(const FieldC& fc, const LevelScales& lsc, int dual, int ch1, int ch2, const WsLayout& w,
                           const Packed* pk, const float* center, const float* ray, int64_t n_rays, float* ws,
                           const Upstream& up, int want_pose, const ls2fm_grid_desc* zero_grid, float* dtable1, float* dtable2,
                           hipStream_t s, int fused_wgrad);

// pose_grad.hip
int ls2fm_launch_pose_grad(const FieldC& fc, const ls2fm_grid_desc* sdf_grid, const ls2fm_grid_desc* rad_grid, int dual,
                           const WsLayout& w, const Packed* pk, const ls2fm_params* params, const float* center,
                           const float* ray, int64_t n_rays, const float* ws, float* d_center, float* d_ray, hipStream_t s);

// wgrad_mlp.hip.  `defer`: leave the sum of the partials to the caller (render_bwd.hip runs it in the finalize launch) and
// describe them here
struct Ls2fmWgradParts { const float* sdf; const float* geo; const float* dec; int nb_sdf, nb_geo, nb_dec, dual; };
// `extra`: a second set of per-sample rows (another workspace of the same layout family: the point-query backward of a traced
// depth, points.hip) whose samples the SDF MLP's kernel contracts in the same launch, behind `ready` (event or null)
struct Ls2fmWgradExtra { WsLayout w; const float* ws; void* ready; };
// fused_wgrad: shade_bwd has left per-slot partials of both MLPs (above) -- no wgrad_mlp launch for the render's own samples
// (only `extra`'s tiles, if any); the decoder launch sums the slots into kL1Seg segment sums in its trailing workgroups
int ls2fm_launch_wgrad_mlp(const FieldC& fc, int dual, int ch1, int ch2, const WsLayout& w, const Packed* pk, const float* center,
                           const float* ray, int64_t n_rays, float* ws, hipStream_t s, bool sdf_only = false,
                           Ls2fmWgradParts* defer = nullptr, const Ls2fmWgradExtra* extra = nullptr, int fused_wgrad = 0);

// points.hip: the stages of ls2fm_sdf_points_bwd (see there)
int ls2fm_points_bwd_front(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params, const float* p,
                           int64_t n, const float* d_sdf, const float* d_feat, const float* d_normal, float* zero_table, bool want_dp,
                           void* workspace, hipStream_t s, WsLayout* w_out, hipEvent_t rows_ready);
int ls2fm_points_bwd_scatter(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, int64_t n, void* workspace, float* table,
                             int add_into, hipStream_t s, int phase);

// shade_fwd.hip
int ls2fm_launch_shade_fwd(const FieldC& fc, int dual, int ch1, int ch2, const Packed* pk, const float* center, const float* ray,
                           int64_t n_rays, const WsLayout& w, float* ws, float* rgb, float* sdfs_volume, float* normals,
                           float* depth_mlp, float* normal_mlp, const ls2fm_loss_spec* loss, hipStream_t s);
