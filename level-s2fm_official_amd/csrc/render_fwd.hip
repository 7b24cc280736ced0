// Fused volumetric-rendering forward for gfx950 (replaces Renderer.forward, models/Renderer.py:51-116).
//
//   prep_weights : weight-norm -> effective weights, packed per hidden unit for scalar loads; the affine
//                  radiance decoder chain (no activation ever fires in the reference: base.py:255-258) is
//                  collapsed to one 3 x 65 map  Wc = R2 R1 R0
//   ray_encode   : thread per (sample, level): AABB near/far + mid-point sample + hash-grid gather with
//                  Jacobian, written as SoA channels
//   shade_fwd    : block per ray, lane per sample: SDF MLP + analytic normal + (second field) + collapsed
//                  radiance + VolSDF sigma, then the front-to-back composite as a wave/block scan
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "bin_items.h"

namespace {

// ------------------------------------------------------------------------------------------- prep
struct LayerRef { const float* v; const float* g; const float* b; int out, in; };

__device__ __forceinline__ LayerRef layer_ref(const ls2fm_params& P, int li, int in_dim, int rad_in) {
    switch (li) {
        case 0: return {P.sdf_mlp[0].weight_v, P.sdf_mlp[0].weight_g, P.sdf_mlp[0].bias, kHidden, in_dim};
        case 1: return {P.sdf_mlp[1].weight_v, P.sdf_mlp[1].weight_g, P.sdf_mlp[1].bias, kOut, kHidden};
        case 2: return {P.geo_mlp[0].weight_v, P.geo_mlp[0].weight_g, P.geo_mlp[0].bias, kHidden, in_dim};
        case 3: return {P.geo_mlp[1].weight_v, P.geo_mlp[1].weight_g, P.geo_mlp[1].bias, kOut, kHidden};
        case 4: return {P.rad_mlp[0].weight_v, P.rad_mlp[0].weight_g, P.rad_mlp[0].bias, kHidden, rad_in};
        case 5: return {P.rad_mlp[1].weight_v, P.rad_mlp[1].weight_g, P.rad_mlp[1].bias, kHidden, kHidden};
        default: return {P.rad_mlp[2].weight_v, P.rad_mlp[2].weight_g, P.rad_mlp[2].bias, 3, kHidden};
    }
}

// first weight-norm row of layer li in the 293-row list (sdf0 sdf1 geo0 geo1 rad0 rad1 rad2)
__device__ __forceinline__ int row_base(int li) {
    switch (li) {
        case 0: return 0;
        case 1: return 64;
        case 2: return 81;
        case 3: return 145;
        case 4: return 162;
        case 5: return 226;
        case 6: return 290;
        default: return 293;
    }
}

// one workgroup of `nt` >= 256 threads (a multiple of 64) per task; callable from the stand-alone kernel below and from
// the first workgroups of the gather pass (ray_encode_kernel)
__device__ void prep_weights_task(const ls2fm_params& P, int in_dim, int in_dim2, int rad_in, int dual, int with_rad,
                                  Packed* __restrict__ out, const int task, const int tid, const int nt) {
    // with_rad == 0: the no-graph SDF evaluation / sphere tracing -- scalar records of the SDF MLP only, no MFMA-ordered copies
    if (!with_rad) {
        // SDF-only users (sdf_eval, sphere tracing: `task` 0 alone, ahead of EVERY such call): one pass, no LDS hand-over -- the 16
        // lanes of a row keep its weights in registers across the norm (same summation order as below: same scale, same bits),
        // scale them and store the packed record entries themselves.  Rows 0..63 = layer 0, 64..80 = layer 1 (stored transposed).
        float* __restrict__ dst = out->sdf;
        for (int idx = tid; idx < kHidden * kRecStride; idx += nt) {          // everything that is not a scaled weight
            const int j = idx / kRecStride, k = idx % kRecStride;
            if (k == kRecB0) dst[idx] = P.sdf_mlp[0].bias[j];
            else if ((k >= in_dim && k < kRecB0) || k >= kRecW1 + kOut) dst[idx] = 0.f;
        }
        for (int o = tid; o < 32; o += nt) dst[kHidden * kRecStride + o] = o < kOut ? P.sdf_mlp[1].bias[o] : 0.f;
        for (int row0 = 0; row0 < kHidden + kOut; row0 += nt / 16) {
            const int row = row0 + (tid >> 4), sub = tid & 15;
            const bool on = row < kHidden + kOut, second = row >= kHidden;
            const int o = second ? row - kHidden : row, in = second ? kHidden : in_dim;
            const float* __restrict__ v = second ? P.sdf_mlp[1].weight_v : P.sdf_mlp[0].weight_v;
            float x[4], ss = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = sub + 16 * q;
                x[q] = (on && k < in) ? v[o * in + k] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) ss = fmaf(x[q], x[q], ss);            // k >= in adds +0: the sum of the loop below
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) ss += __shfl_xor(ss, m, 16);
            if (on) {
                const float sc = (second ? P.sdf_mlp[1].weight_g : P.sdf_mlp[0].weight_g)[o] / sqrtf(ss);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = sub + 16 * q;
                    if (k < in) dst[second ? k * kRecStride + kRecW1 + o : o * kRecStride + k] = x[q] * sc;
                }
            }
        }
        if (tid == 3) {
            const float beta = expf(P.beta[0] * P.beta_speed);
            out->beta = beta;
            out->alpha = 1.0f / beta;
        }
        return;
    }
    __shared__ float row_scale[296];
    // one workgroup per independent task (a single workgroup doing everything was a 72 us latency chain that gated
    // shade_fwd in single-field runs): 0 = SDF MLP, 1 = second field's MLP, 2 = radiance chain (+ beta)
    if (task == 1 && !dual) return;
    const int row_lo = task == 0 ? row_base(0) : (task == 1 ? row_base(2) : row_base(4));
    const int row_hi = task == 0 ? row_base(2) : (task == 1 ? row_base(4) : row_base(7));
    // 1. weight-norm row scales  s = g / ||v||   (torch._weight_norm(v, g, 0) = v * (g / norm))
    //    (16 lanes per row: coalesced reads + a 16-wide shuffle reduction instead of a serial latency chain)
    for (int row0 = row_lo; row0 < row_hi; row0 += nt / 16) {
        const int row = row0 + (tid >> 4), sub = tid & 15;
        int li = 0;
        while (li < 6 && row >= row_base(li + 1)) ++li;
        const bool on = row < row_hi && (dual || (li != 2 && li != 3)) && (with_rad || li < 4);
        const LayerRef L = layer_ref(P, li, li == 2 ? in_dim2 : in_dim, rad_in);
        const int o = row - row_base(li);
        float ss = 0.f;
        if (on)
            for (int k = sub; k < L.in; k += 16) { const float x = L.v[o * L.in + k]; ss = fmaf(x, x, ss); }
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) ss += __shfl_xor(ss, m, 16);
        if (sub == 0 && row < row_hi) row_scale[row] = on ? L.g[o] / sqrtf(ss) : 0.f;
    }
    __syncthreads();
    // 2. packed geometry MLPs
    for (int which = 0; which < (dual ? 2 : 1); ++which) {
        if (which != task) continue;
        float* dst = which ? out->geo : out->sdf;
        const int ind = which ? in_dim2 : in_dim;
        const LayerRef L0 = layer_ref(P, which ? 2 : 0, ind, rad_in), L1 = layer_ref(P, which ? 3 : 1, ind, rad_in);
        const float* s0 = row_scale + row_base(which ? 2 : 0);
        const float* s1 = row_scale + row_base(which ? 3 : 1);
        // the scaled layers once into LDS: the eight packing loops below then read LDS instead of making a global round trip each
        // (for a point query's small gather launch this chain IS the launch: 25 us for 4096 points)
        __shared__ float s_l0[kHidden * 36];          // [j][k < ind: v s0 | 35: b0]
        __shared__ float s_l1[kOut * kHidden];        // [o][j]: v s1
        for (int idx = tid; idx < kHidden * 36; idx += nt) {
            const int j = idx / 36, k = idx % 36;
            s_l0[idx] = k < ind ? L0.v[j * ind + k] * s0[j] : (k == 35 ? L0.b[j] : 0.f);
        }
        for (int idx = tid; idx < kOut * kHidden; idx += nt) s_l1[idx] = L1.v[idx] * s1[idx / kHidden];
        __syncthreads();
        for (int idx = tid; idx < kHidden * kRecStride; idx += nt) {
            const int j = idx / kRecStride, k = idx % kRecStride;
            float val = 0.f;
            if (k < ind) val = s_l0[j * 36 + k];
            else if (k == kRecB0) val = s_l0[j * 36 + 35];
            else if (k >= kRecW1 && k < kRecW1 + kOut) val = s_l1[(k - kRecW1) * kHidden + j];
            dst[idx] = val;
        }
        for (int o = tid; o < 32; o += nt) dst[kHidden * kRecStride + o] = o < kOut ? L1.b[o] : 0.f;
        if (!with_rad) continue;
        // MFMA-operand-ordered copies (shade kernels): see MfmaW
        auto w0p = [&](int j, int kp) -> float {
            if (kp < 32) return 3 + kp < ind ? s_l0[j * 36 + 3 + kp] : 0.f;
            if (kp < 35) return s_l0[j * 36 + (kp - 32)];
            return kp == 35 ? s_l0[j * 36 + 35] : 0.f;
        };
        MfmaW& mw = out->mw;
        for (int idx = tid; idx < 4 * 9 * 64; idx += nt) {
            const int m = idx / (9 * 64), t = (idx / 64) % 9, ln = idx & 63;
            (which ? mw.geo : mw.sdf).w0a[m][t][ln] = w0p(16 * m + (ln & 15), 4 * t + (ln >> 4));
        }
        for (int idx = tid; idx < 4 * 4 * 64; idx += nt) {
            const int m = idx / 256, r = (idx / 64) & 3, ln = idx & 63;
            const int hid = 16 * m + 4 * (ln >> 4) + r;
            (which ? mw.geo : mw.sdf).w1a[m][r][ln] = s_l1[(1 + (ln & 15)) * kHidden + hid];
            if (which == 1) mw.w0tx_geo[m][r][ln] = (ln & 15) < 3 ? w0p(hid, 32 + (ln & 15)) : 0.f;
            if (which == 0) {
                mw.w10[m][r][ln] = s_l1[hid];
                for (int mk = 0; mk < 3; ++mk) {
                    const int kp = 16 * mk + (ln & 15);
                    mw.w0ta[mk][m][r][ln] = kp < 35 ? w0p(hid, kp) : 0.f;
                }
            }
        }
        if (tid < 256) {
            const int r = tid >> 6, ln = tid & 63;
            mw.b1a[which][r][ln] = L1.b[1 + 4 * (ln >> 4) + r];
        }
        // backward copies
        for (int idx = tid; idx < 4 * 9 * 64; idx += nt) {
            const int m = idx / (9 * 64), t = (idx / 64) % 9, ln = idx & 63;
            const float v = w0p(16 * m + (ln & 15), 4 * t + (ln >> 4));
            if (which) out->bg.w0a[m][t][ln] = v; else out->bs.w0a[m][t][ln] = v;
        }
        for (int idx = tid; idx < 4 * 5 * 64; idx += nt) {
            const int m = idx / (5 * 64), t = (idx / 64) % 5, ln = idx & 63;
            const int hid = 16 * m + (ln & 15);
            if (which) {
                if (t < 4) out->bg.w1ta[m][t][ln] = s_l1[(1 + 4 * t + (ln >> 4)) * kHidden + hid];
            } else {
                const int o = 4 * t + (ln >> 4);
                out->bs.w1ta[m][t][ln] = o < kOut ? s_l1[o * kHidden + hid] : 0.f;
            }
        }
        for (int idx = tid; idx < 2 * 4 * 4 * 64; idx += nt) {
            const int mk = idx / 1024, m = (idx / 256) & 3, r = (idx / 64) & 3, ln = idx & 63;
            const int hid = 16 * m + 4 * (ln >> 4) + r;
            const float v = w0p(hid, 16 * mk + (ln & 15));
            if (which) out->bg.w0ta[mk][m][r][ln] = v;
            else {
                out->bs.w0ta[mk][m][r][ln] = v;
                if (mk == 0) out->bs.w10[m][r][ln] = s_l1[hid];
            }
        }
        if (tid == 0) mw.b10[which] = L1.b[0];
    }
    if (task == 0 && tid == 3) {
        const float beta = expf(P.beta[0] * P.beta_speed);
        out->beta = beta;
        out->alpha = 1.0f / beta;
    }
    if (task != 2 || !with_rad) return;          // SDF-only users (sdf_eval, sphere_trace) launch task 0 alone
    // 3. effective radiance layers
    {
        const LayerRef R0 = layer_ref(P, 4, in_dim, rad_in), R1 = layer_ref(P, 5, in_dim, rad_in),
                       R2 = layer_ref(P, 6, in_dim, rad_in);
        for (int idx = tid; idx < 64 * 68; idx += nt) {
            const int j = idx / 68, k = idx % 68;
            out->r0[j][k] = k < rad_in ? R0.v[j * rad_in + k] * row_scale[row_base(4) + j] : 0.f;
        }
        for (int idx = tid; idx < 64 * 64; idx += nt)
            out->r1[idx / 64][idx % 64] = R1.v[idx] * row_scale[row_base(5) + idx / 64];
        for (int idx = tid; idx < 3 * 64; idx += nt)
            out->r2[idx / 64][idx % 64] = R2.v[idx] * row_scale[row_base(6) + idx / 64];
    }
    __syncthreads();
    // 4. T1 = R2 R1
    for (int idx = tid; idx < 3 * 64; idx += nt) {
        const int c = idx / 64, j = idx % 64;
        float acc = 0.f;
        for (int m = 0; m < 64; ++m) acc = fmaf(out->r2[c][m], out->r1[m][j], acc);
        out->t1[c][j] = acc;
    }
    __syncthreads();
    // 5. Wc = T1 R0 ; bc = T1 b0 + R2 b1 + b2
    for (int idx = tid; idx < 3 * 68; idx += nt) {
        const int c = idx / 68, k = idx % 68;
        float acc = 0.f;
        if (k < rad_in)
            for (int j = 0; j < 64; ++j) acc = fmaf(out->t1[c][j], out->r0[j][k], acc);
        out->wc[c][k] = acc;
    }
    if (tid < 3) {
        const int c = tid;
        float acc = P.rad_mlp[2].bias[c];
        for (int m = 0; m < 64; ++m) acc = fmaf(out->r2[c][m], P.rad_mlp[1].bias[m], acc);
        for (int j = 0; j < 64; ++j) acc = fmaf(out->t1[c][j], P.rad_mlp[0].bias[j], acc);
        out->bc[c] = acc;
    }
    if (tid == 3) out->bc[3] = 0.f;
}

constexpr int kPrepThreads = 1024;      // a chain of dependent round trips (norms -> scales -> packed rows): the wider the workgroup, the
                                        // fewer trips per thread (256 threads: 13 us for the SDF MLP alone, ahead of every tracing call)
__global__ void __launch_bounds__(kPrepThreads)
prep_weights_kernel(ls2fm_params P, int in_dim, int in_dim2, int rad_in, int dual, int with_rad, Packed* __restrict__ out,
                    int32_t* __restrict__ zero_word) {
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;        // e.g. a tracing call's trip counter: no launch of its own
    prep_weights_task(P, in_dim, in_dim2, rad_in, dual, with_rad, out, (int)blockIdx.x, (int)threadIdx.x, kPrepThreads);
}

// ------------------------------------------------------------------------------------------- ray_encode
// One launch gathers every level of the SDF grid (value + Jacobian) and, dual field, of the second grid (value).
// Block -> (grid, level, point chunk) mapping is XCD-aware: workgroup b is observed to run on XCD b % 8, and each XCD has a
// private 4 MiB L2 -- exactly one 2^19-entry level of one table.  All workgroups of one XCD therefore work on the same
// level at a time, so a level's table slice is fetched into ONE L2 once instead of into all eight.  The walking order
// [grid 1: level 0 chunks .., level 1 chunks, .. ; grid 2: ..] is cut into 8 contiguous pieces of equal COST: dense levels
// are much cheaper per chunk than hashed ones (their gathers hit the L1: measured 0.28 of a fine level, tools/enc_ticks.py),
// so a fixed "levels x, x + 8 per XCD" split leaves the XCDs that own the dense levels idle at the end (152 -> 125 us for
// both grids).  At most two XCDs share a level.  This is a speed choice only; any placement gives the same result.
struct XcdPlan { int start[9]; };

#ifdef LS2FM_STAMPS
__device__ unsigned long long g_enc_ticks[2 * LS2FM_MAX_LEVELS + 24];    // [0,32) summed workgroup durations per pass-level
                                                                          // (100 MHz); [32,40) last end per XCD; [40,48) first start
#endif

// What the gather pass does besides gathering, in the same launch (a launch boundary on the main chain costs 2-10 us, a
// cross-stream join ~10 us, and a side stream's kernels take CUs from the main chain's):
//   * its first 8 workgroups are reserved for the weight prep (3 tasks; the latency-bound 33 us chain hides under the gather)
//   * when a backward will follow, every workgroup also COUNTS the items its 512 sample points will contribute to each slab
//     of the table-gradient scatter on its level (bin_items.h): the corner cells are known here anyway, and the count /
//     scan chain of the backward (16 us alone, 85 us beside shade_bwd, then 35 us of waiting before scatter_fill) is gone
struct EncodeExtras {
    ls2fm_params params;      // prep
    int in_dim, in_dim2, rad_in, dual;
    Packed* packed;           // null: no prep in this launch
    int* tile_counts;         // [L][n_tiles][kBins] or null: no counting
    int* scan_ticket;
    int n_tiles, sshift;
    int n_explicit;           // dual field: leading levels whose scatter items are explicit and run-merged (bin_items.h)
    int n_prep_tasks;         // 3: both MLPs + the radiance chain (render); 1: the SDF MLP only (point queries)
    const float* pts;         // [n,3] free points instead of ray samples (point queries), or null
    int probe;                // LS2FM_ENC_PROBE (measurements, round 6): 1 = walking order NOT pinned to XCDs (every XCD walks every
                              // level over its eighth of the chunks: what a gather fused into the per-ray shading kernel would see;
                              // same results); 2 = timing probe, WRONG results: a hashed level's corner loads restricted to one HALF
                              // of its slice by entry index, lower half during the first half of an XCD's chunks of the level, upper
                              // half during the second (one of the two passes an entry-range split between XCDs would make);
                              // 3 = timing probe: every corner reads entry 0 of its level (no table traffic); 4 = timing probe: no stores
};

constexpr int kEncThreads = 512;             // sample points of one level per workgroup = kEncRows tiles of the scatter's counting sort
constexpr int kEncRows = kEncThreads / kFillTile;
static_assert(kEncThreads % kFillTile == 0, "a gather workgroup covers whole tiles");
constexpr int kEncReserved = 8;              // leading workgroups (a multiple of 8: block -> XCD mapping stays b % 8)

// position of sample i, its cell on the level, the 8 corner entries (absolute) and fractions
struct LevelOne { float scale; uint32_t res, size, offset, hashed; };      // one level's constants (scalar registers)

__device__ __forceinline__ void locate_sample(const FieldC& fc, const float* __restrict__ center, const float* __restrict__ ray,
                                              const float* __restrict__ pts, int64_t i, const LevelOne& lv, uint32_t g[3], Cell& c) {
    float p[3], x[3];
    if (pts) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            p[a] = pts[i * 3 + a];
            x[a] = (p[a] - fc.bmin[a]) / (fc.bmax[a] - fc.bmin[a]);
        }
    } else {
        const int64_t r = i / fc.n_samples;
        const int n = (int)(i - r * fc.n_samples);
        const RayGeom gm = load_ray(fc, center, ray, r);
        sample_position(fc, gm, sample_depth(gm, n, fc.n_samples), p, x);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) pos_fract(x[d], lv.scale, g[d], c.w[d]);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        c.idx[k] = lv.offset + corner_index(g[0] + (k & 1), g[1] + ((k >> 1) & 1), g[2] + ((k >> 2) & 1), lv.res, lv.size,
                                            lv.hashed);
}

// INTERLEAVED: dual field with the entry-interleaved table copy (ls2fm_params.dual_table): one 16-byte gather per corner
// serves both grids -- the gathers are bound by the L2 -> L1 line rate, not by bytes.
// (the measurement probes are compiled in only with -DLS2FM_ENC_PROBES: inside the shipped kernel their branches cost 14 registers --
// 70 -> 84, five instead of seven waves per SIMD)
// Issue priority (round 6, profiles/r06_notes.md section 9).  LS2FM_ENC_PRIO_{PAIR,ONE} for the interleaved two-grid walk / the one-grid
// walk: 1 = a wave runs at raised priority until its eight gathers are issued (young waves get their loads out before older ones
// interpolate), 2 = the reverse: raised BEHIND the loads (a wave that has its data interpolates, stores and frees its slot first).
// Two grids (16-byte gathers, 70 registers): 98 -> 111 / 105 us, left alone.  One grid (8-byte gathers, the reference's default
// field): 77 -> 75.6 / 71.9 us at C2's shape -- 2 is its default.
#ifndef LS2FM_ENC_PRIO_PAIR
#define LS2FM_ENC_PRIO_PAIR 0
#endif
#ifndef LS2FM_ENC_PRIO_ONE
#define LS2FM_ENC_PRIO_ONE 2
#endif
#define ENC_PRIO_MODE (INTERLEAVED ? LS2FM_ENC_PRIO_PAIR : LS2FM_ENC_PRIO_ONE)
#define ENC_PRIO_HEAD(x) do { if (ENC_PRIO_MODE == 1) __builtin_amdgcn_s_setprio(x); } while (0)
#define ENC_PRIO_TAIL(x) do { if (ENC_PRIO_MODE == 2) __builtin_amdgcn_s_setprio(x); } while (0)
#ifdef LS2FM_ENC_PROBES
#define ENC_PROBE(ex) ((ex).probe)
#else
#define ENC_PROBE(ex) 0
#endif
#ifndef LS2FM_ENC_MINW
#define LS2FM_ENC_MINW 1
#endif
#ifndef LS2FM_ENC_OFF32
#define LS2FM_ENC_OFF32 0
#endif
#ifndef LS2FM_ENC_OFF32_LD
#define LS2FM_ENC_OFF32_LD LS2FM_ENC_OFF32
#endif
#ifndef LS2FM_ENC_OFF32_ST
#define LS2FM_ENC_OFF32_ST LS2FM_ENC_OFF32
#endif
template <bool INTERLEAVED>
__global__ void __launch_bounds__(kEncThreads, LS2FM_ENC_MINW)
ray_encode_kernel(LevelSet lv1, LevelSet lv2, FieldC fc, const float* __restrict__ center, const float* __restrict__ ray,
                  const float* __restrict__ table1, const float* __restrict__ table2, int64_t n_points, int64_t p_pad,
                  int n_chunks, XcdPlan plan, float* __restrict__ enc1, float* __restrict__ enc2, float* __restrict__ jac,
                  EncodeExtras ex) {
    __shared__ int hist[kEncRows][kBins];
    const int tid = threadIdx.x;
    if (blockIdx.x < kEncReserved) {
        if (ex.packed && (int)blockIdx.x < ex.n_prep_tasks)
            prep_weights_task(ex.params, ex.in_dim, ex.in_dim2, ex.rad_in, ex.dual, 1, ex.packed, (int)blockIdx.x, tid, kEncThreads);
        if (blockIdx.x == 3 && tid == 0 && ex.scan_ticket) *ex.scan_ticket = 0;      // armed for the scans in the shade_fwd launch
        return;
    }
    ENC_PRIO_HEAD(2);
    const int bx = (int)blockIdx.x - kEncReserved;
    const int xcd = bx & 7, j = bx >> 3;
    int unit = plan.start[xcd] + j;
    if (ENC_PROBE(ex) == 1) {                                 // (measurement) chunk-major: XCD x walks chunks x, x + 8, .. of EVERY pass-level
        const int n_pl = plan.start[8] / n_chunks;
        const int ck = (j / n_pl) * 8 + xcd;
        if (ck >= n_chunks) return;
        unit = (j % n_pl) * n_chunks + ck;
    } else if (unit >= plan.start[xcd + 1]) return;
    const int pl = unit / n_chunks;                      // pass-level; two grids: (level 0, grid 1), (level 0, grid 2), (level 1, ..
    const int chunk = unit % n_chunks;
    const bool two = !INTERLEAVED && table2 != nullptr;
    const bool second = two && (pl & 1);
    const int l = two ? pl >> 1 : pl;
    // the level's constants straight from the kernel-argument segment (a reference selecting between the two structs makes
    // the compiler copy both to scratch: 652 bytes per lane and an 8x slower kernel)
    LevelOne lv;
    lv.scale = second ? lv2.scale[l] : lv1.scale[l];
    lv.res = second ? lv2.res[l] : lv1.res[l];
    lv.size = second ? lv2.size[l] : lv1.size[l];
    lv.offset = second ? lv2.offset[l] : lv1.offset[l];
    lv.hashed = second ? lv2.hashed[l] : lv1.hashed[l];
    const bool counting = ex.tile_counts != nullptr && !second;          // workgroup-uniform
    if (counting) {
        for (int b = tid; b < kEncRows * kBins; b += kEncThreads) (&hist[0][0])[b] = 0;
        __syncthreads();
    }
    const int64_t i = (int64_t)chunk * kEncThreads + tid;
#ifdef LS2FM_STAMPS
    const long long t_begin = wall_clock64();
#endif
    if (i < n_points) {
        uint32_t g[3];
        Cell c;
        locate_sample(fc, center, ray, ex.pts, i, lv, g, c);
        // row `row` of a [channel][p_pad] (k = 1) / [channel][p_pad][3] (k = 3) block, this thread's point: uniform base + 32-bit byte
        // offset (LS2FM_ENC_OFF32: `global_store v_off, v, s[base]` instead of a 64-bit address per store; 32 rows x 2^23 points x 12 B < 2^32)
        const uint32_t pp32 = (uint32_t)p_pad, i32 = (uint32_t)i;
        auto enc_at = [=](float* __restrict__ base, int row, uint32_t k) -> float* {
            return LS2FM_ENC_OFF32_ST ? reinterpret_cast<float*>(reinterpret_cast<char*>(base) + ((uint32_t)row * pp32 + i32) * (4u * k))
                                   : base + ((int64_t)row * p_pad + i) * k;
        };
        if (INTERLEAVED) {
            const float4* __restrict__ table = reinterpret_cast<const float4*>(table1);
            float4 v[8];
            if (ENC_PROBE(ex) == 3) {                         // (timing probe: wrong results) no table traffic: every corner reads the level's entry 0
#pragma unroll
                for (int k = 0; k < 8; ++k) c.idx[k] = lv.offset;
            }
            if (ENC_PROBE(ex) == 2 && lv.hashed) {            // (timing probe: wrong results) only the corners in one half of the slice
                const uint32_t want = 2 * chunk >= n_chunks ? 1u : 0u, hb = lv.size >> 1;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if ((((c.idx[k] - lv.offset) >= hb) ? 1u : 0u) != want) c.idx[k] = lv.offset + want * hb;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                v[k] = LS2FM_ENC_OFF32_LD ? *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(table1) + c.idx[k] * 16u) : table[c.idx[k]];
            ENC_PRIO_HEAD(0); ENC_PRIO_TAIL(2);
            float y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float wt = corner_weight(c.w, k);
                y0 = fmaf(wt, v[k].x, y0);
                y1 = fmaf(wt, v[k].y, y1);
                y2 = fmaf(wt, v[k].z, y2);
                y3 = fmaf(wt, v[k].w, y3);
            }
            const bool st = ENC_PROBE(ex) != 4 || y0 == 1234.5f;          // (probe 4, timing only: no value / Jacobian stores)
            if (st) {
            __builtin_nontemporal_store(y0, enc_at(enc1, 2 * l + 0, 1u));
            __builtin_nontemporal_store(y1, enc_at(enc1, 2 * l + 1, 1u));
            __builtin_nontemporal_store(y2, enc_at(enc2, 2 * l + 0, 1u));
            __builtin_nontemporal_store(y3, enc_at(enc2, 2 * l + 1, 1u));
            }
#pragma unroll
            for (int gd = 0; gd < 3; ++gd) {
                float g0 = 0.f, g1 = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float dw = corner_dweight(c.w, k, gd);
                    g0 = fmaf(dw, v[k].x, g0);
                    g1 = fmaf(dw, v[k].y, g1);
                }
                if (st || g0 == 1234.5f) {
                __builtin_nontemporal_store(lv.scale * g0, enc_at(jac, 2 * l + 0, 3u) + gd);      // [channel][point][3]
                __builtin_nontemporal_store(lv.scale * g1, enc_at(jac, 2 * l + 1, 3u) + gd);
                }
            }
        } else {
            const float* __restrict__ table = second ? table2 : table1;
            float2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                v[k] = LS2FM_ENC_OFF32_LD ? *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(table) + c.idx[k] * 8u)
                                       : *reinterpret_cast<const float2*>(table + 2ull * c.idx[k]);
            ENC_PRIO_HEAD(0); ENC_PRIO_TAIL(2);
            float y0 = 0.f, y1 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float wt = corner_weight(c.w, k);
                y0 = fmaf(wt, v[k].x, y0);
                y1 = fmaf(wt, v[k].y, y1);
            }
            float* __restrict__ enc = second ? enc2 : enc1;
            __builtin_nontemporal_store(y0, enc_at(enc, 2 * l + 0, 1u));
            __builtin_nontemporal_store(y1, enc_at(enc, 2 * l + 1, 1u));
            if (!second) {
#pragma unroll
                for (int gd = 0; gd < 3; ++gd) {
                    float g0 = 0.f, g1 = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float dw = corner_dweight(c.w, k, gd);
                        g0 = fmaf(dw, v[k].x, g0);
                        g1 = fmaf(dw, v[k].y, g1);
                    }
                    __builtin_nontemporal_store(lv.scale * g0, enc_at(jac, 2 * l + 0, 3u) + gd);  // [channel][point][3]
                    __builtin_nontemporal_store(lv.scale * g1, enc_at(jac, 2 * l + 1, 3u) + gd);
                }
            }
        }
        if (counting) {        // the classification of scatter_fill (for_each_item_merged, bin_items.h) from the corner entries
                               // already at hand: pair c = (by, bz) -> x-corners k = 2 by + 4 bz and k + 1.  Every lane in
                               // here is live and so is its predecessor (live lanes are a prefix of the wave): the run flags
                               // are those the fill pass derives from the same cells
            const bool factored = ex.dual != 0 && l >= ex.n_explicit;        // the factored 32-byte dual item (fine levels)
            const RunFlags rf = wave_runs(g, true, tid & 63, factored ? kMergeMinDual : kMergeMinSingle);
            if (rf.head) {
                const bool halves = factored && rf.merged;                    // a merged run of a factored pair: two half items
#pragma unroll
                for (int cp = 0; cp < 4; ++cp) {
                    const uint32_t i0 = c.idx[2 * cp] - lv.offset, i1 = c.idx[2 * cp + 1] - lv.offset;
                    const uint32_t s0 = i0 >> ex.sshift, s1 = i1 >> ex.sshift;
                    atomicAdd(&hist[tid / kFillTile][s0], 1);
                    if (s1 != s0 || halves) atomicAdd(&hist[tid / kFillTile][s1], 1);       // split pair: two half items
                }
            }
        }
    }
    if (counting) {
        __syncthreads();
        // tiles kEncRows * chunk .. + kEncRows - 1 of this level (rows beyond the last tile: never read)
        for (int b = tid; b < kEncRows * kBins; b += kEncThreads) {
            const int tile = kEncRows * chunk + b / kBins;
            if (tile < ex.n_tiles) ex.tile_counts[((int64_t)l * ex.n_tiles + tile) * kBins + b % kBins] = hist[b / kBins][b % kBins];
        }
    }
#ifdef LS2FM_STAMPS
    __syncthreads();
    if (threadIdx.x == 0) {
        const long long t_end = wall_clock64();
        atomicAdd(&g_enc_ticks[pl], (unsigned long long)(t_end - t_begin));
        atomicMax(&g_enc_ticks[32 + xcd], (unsigned long long)t_end);
        atomicMin(&g_enc_ticks[40 + xcd], (unsigned long long)t_begin);
    }
#endif
}

__global__ void __launch_bounds__(256)
interleave_tables_kernel(const float2* __restrict__ a, const float2* __restrict__ b, float4* __restrict__ out, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 512 + 2 * threadIdx.x;      // two entries per thread: 16-byte loads
    if (i + 1 < n) {
        const float4 va = *reinterpret_cast<const float4*>(a + i), vb = *reinterpret_cast<const float4*>(b + i);
        out[i] = make_float4(va.x, va.y, vb.x, vb.y);
        out[i + 1] = make_float4(va.z, va.w, vb.z, vb.w);
    } else if (i < n) {
        out[i] = make_float4(a[i].x, a[i].y, b[i].x, b[i].y);
    }
}

}  // namespace

#ifdef LS2FM_STAMPS
extern "C" int ls2fm_debug_enc_reset(void) {
    unsigned long long init[2 * LS2FM_MAX_LEVELS + 24];
    for (int i = 0; i < 2 * LS2FM_MAX_LEVELS + 24; ++i) init[i] = i >= 40 && i < 48 ? ~0ull : 0ull;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_enc_ticks), init, sizeof(init)) == hipSuccess ? 0 : -1;
}
extern "C" int ls2fm_debug_enc_ticks(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_enc_ticks), sizeof(unsigned long long) * (2 * LS2FM_MAX_LEVELS + 24)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int ls2fm_interleave_tables(const float* sdf_table, const float* rad_table, int64_t n_entries, float* dual_table,
                                       void* stream) {
    LS2FM_CHECK_ARG(n_entries >= 0 && (n_entries == 0 || (sdf_table && rad_table && dual_table)));
    LS2FM_CHECK_ARG(((reinterpret_cast<uintptr_t>(sdf_table) | reinterpret_cast<uintptr_t>(rad_table) |
                      reinterpret_cast<uintptr_t>(dual_table)) & 15u) == 0);
    if (n_entries == 0) return LS2FM_OK;
    interleave_tables_kernel<<<(unsigned)((n_entries + 511) / 512), 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const float2*>(sdf_table), reinterpret_cast<const float2*>(rad_table),
        reinterpret_cast<float4*>(dual_table), n_entries);
    return ls2fm_launch_status();
}

// cost-balanced cut of the walking order [grid 1 levels .., grid 2 levels ..] x chunks into 8 XCD pieces
static XcdPlan make_xcd_plan(const ls2fm_grid_desc* g1, int l1, const ls2fm_grid_desc* g2, int l2, int n_samples, int n_chunks,
                             bool counting, int* most, bool interleaved = false) {
    double cost[2 * LS2FM_MAX_LEVELS], total = 0.0;
    // Relative cost of a pass-level = summed workgroup durations INSIDE a training step (tools/enc_xcd_instep.py: the launch
    // follows a backward, the L2s no longer hold the table slices; a fine hashed level of the second grid = 1): dense levels
    // 0.23-0.33; hashed levels grow with the number of distinct cells a wave's consecutive samples touch (scale / n_samples):
    // 0.71, 0.93, then 1.0.  The FIRST grid's pass-levels also write the Jacobian channels and, when a backward follows, count
    // the scatter's items: hashed levels +10 %, dense levels +50 % (their increments collide in the LDS histogram).  Two grids
    // (same geometry) alternate level by level in the walking order, so every XCD's piece holds both kinds (with "all of grid
    // 1, then all of grid 2" the XCDs owning grid 1 finished 20 us late).
    // ONE pass over the entry-interleaved table (16-byte gathers, a level slice is 8 MB: twice an XCD's L2): the coarse levels
    // are relatively cheaper than in the two-table pass -- measured 0.20-0.30 (dense), 0.61, 0.90, then 1.0 -- with the
    // two-table constants the XCD that owns them finished 21 us before the last one (86 vs 107 us).
    static double cm2[6] = {0.27, 1.5, 1.1, 0.45, 0.45, 1.0};     // two tables:  dense, first_dense, first_hashed, slope, intercept, top
    static double cmi[6] = {0.31, 1.0, 1.0, 1.0, -0.156, 1.0};    // interleaved
    static const bool cm_env = [] {
        const char* e = getenv("LS2FM_CM");                      // experiments: "dense,first_dense,first_hashed,slope,intercept"
        if (e) sscanf(e, "%lf,%lf,%lf,%lf,%lf,%lf", &cm2[0], &cm2[1], &cm2[2], &cm2[3], &cm2[4], &cm2[5]);
        const char* e2 = getenv("LS2FM_CMI");
        if (e2) sscanf(e2, "%lf,%lf,%lf,%lf,%lf,%lf", &cmi[0], &cmi[1], &cmi[2], &cmi[3], &cmi[4], &cmi[5]);
        return e != nullptr;
    }();
    (void)cm_env;
    const double* cm = interleaved ? cmi : cm2;
    const int n_pl = l1 + l2;
    for (int pl = 0; pl < n_pl; ++pl) {
        const bool second = l2 > 0 && (pl & 1);
        const ls2fm_grid_desc* gd = second ? g2 : g1;
        const int l = l2 > 0 ? pl >> 1 : pl;
        const double rho = (double)gd->scale[l] / (double)n_samples;
        const double h = cm[4] + cm[3] * log2(1.0 + rho);
        cost[pl] = gd->hashed[l] ? (h > 1.0 ? 1.0 : (h < 0.1 ? 0.1 : h)) : cm[0];
        if (!second) cost[pl] *= gd->hashed[l] ? (counting ? cm[2] : 1.05) : (counting ? cm[1] : 1.1);
        if (l >= (l2 > 0 ? l2 : l1) - 2) cost[pl] *= cm[5];        // the two finest levels
        total += cost[pl];
    }
    XcdPlan plan;
    plan.start[0] = 0;
    *most = 0;
    for (int x = 1; x <= 8; ++x) {
        const double target = total * x / 8.0;       // cumulative cost at the end of XCD x - 1's piece
        double acc = 0.0;
        int unit = n_pl * n_chunks;
        for (int pl = 0; pl < n_pl; ++pl) {
            if (acc + cost[pl] >= target - 1e-9) {
                unit = pl * n_chunks + (int)((target - acc) / cost[pl] * n_chunks + 0.5);
                break;
            }
            acc += cost[pl];
        }
        if (x == 8 || unit > n_pl * n_chunks) unit = n_pl * n_chunks;
        if (unit < plan.start[x - 1]) unit = plan.start[x - 1];
        plan.start[x] = unit;
        *most = *most > unit - plan.start[x - 1] ? *most : unit - plan.start[x - 1];
    }
    return plan;
}

// weight-norm + packing of the SDF MLP only (shared with sdf_eval.hip)
int ls2fm_launch_prep_sdf(const ls2fm_params* params, int n_levels, Packed* out, hipStream_t stream, int32_t* zero_word) {
    ls2fm_prof_begin(LS2FM_PROF_PREP, stream);
    prep_weights_kernel<<<1, kPrepThreads, 0, stream>>>(*params, 3 + 2 * n_levels, 0, 0, 0, 0, out, zero_word);      // task 0 only
    ls2fm_prof_end(LS2FM_PROF_PREP, stream);
    return ls2fm_launch_status();
}

// ------------------------------------------------------------------------------------------- C ABI
static bool render_config_ok(const ls2fm_field_desc* field, const ls2fm_grid_desc* g1, const ls2fm_grid_desc* g2) {
    if (!field || !grid_desc_ok(g1)) return false;
    if (field->dual_field && !grid_desc_ok(g2)) return false;
    if (field->n_samples < 1 || field->n_samples > 1024) return false;
    return true;
}

extern "C" int64_t ls2fm_render_workspace_bytes(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid,
                                                int64_t n_rays) {
    if (!field || !grid_desc_ok(grid) || n_rays < 0) return LS2FM_ERR_INVALID_ARGUMENT;
    if (n_rays * (int64_t)field->n_samples > LS2FM_MAX_RENDER_POINTS) return LS2FM_ERR_UNSUPPORTED;
    // the second grid (dual field) uses the same encoding config as the first (models/RadF.py:35-39)
    const WsLayout w = make_ws_layout(n_rays, field->n_samples, grid->n_levels, grid->n_levels, field->dual_field);
    return w.total * (int64_t)sizeof(float);
}

bool ls2fm_bins_levels_fit(const ls2fm_grid_desc* grid, int dual);

// Gather pass over FREE POINTS (the point-query backward, points.hip): value + Jacobian channels of the SDF grid, the SDF MLP's
// weight prep in the leading workgroups and the scatter's item counts -- the same kernel as the render's
int ls2fm_launch_points_encode(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                               const float* pts, const WsLayout& w, float* ws, hipStream_t s) {
    const FieldC fc = make_field_c(field);
    const int L = grid->n_levels;
    const int n_chunks = (int)((w.p + kEncThreads - 1) / kEncThreads);
    int most = 0;
    const XcdPlan plan = make_xcd_plan(grid, L, nullptr, 0, 1, n_chunks, true, &most);
    EncodeExtras ex;
    ex.params = *params;
    ex.in_dim = 3 + 2 * L; ex.in_dim2 = 0; ex.rad_in = 0; ex.dual = 0;
    ex.packed = (Packed*)(ws + w.packed);
    const BinMeta bm = make_bin_meta(ws + w.bins, w.p);
    ex.tile_counts = bm.tile;
    ex.scan_ticket = scan_ticket(bm);
    ex.n_tiles = bm.n_tiles;
    ex.sshift = ls2fm_slab_shift(0);
    ex.n_explicit = 0;
    ex.n_prep_tasks = 1;
    ex.pts = pts;
    ex.probe = 0;
    ray_encode_kernel<false><<<(unsigned)(kEncReserved + 8 * most), kEncThreads, 0, s>>>(
        make_level_set(grid), make_level_set(grid), fc, nullptr, nullptr, params->sdf_table, nullptr, w.p, w.p_pad, n_chunks, plan,
        ws + w.e1, nullptr, ws + w.j1, ex);
    return ls2fm_launch_status();
}
int ls2fm_launch_post_shade(const ls2fm_loss_spec* loss, const float* ray_part, int64_t n_rays, int n_samples,
                            const ls2fm_grid_desc* scan_grid, int64_t n_points, float* bins_ws, hipStream_t stream);

extern "C" int ls2fm_render_fwd(const ls2fm_field_desc* field, const ls2fm_grid_desc* sdf_grid,
                                const ls2fm_grid_desc* rad_grid, const ls2fm_params* params, const float* center,
                                const float* ray, int64_t n_rays, float* rgb, float* sdfs_volume, float* normals,
                                float* depth_mlp, float* normal_mlp, void* workspace, const ls2fm_render_opts* opts,
                                void* stream) {
    LS2FM_CHECK_ARG(render_config_ok(field, sdf_grid, rad_grid) && params && n_rays >= 0);
    if (field->bg_sdf) return LS2FM_ERR_UNSUPPORTED;       // min(sdf, bg_rad-|p|): general (composed) form only
    if (field->dual_field && !same_grid_geometry(sdf_grid, rad_grid)) return LS2FM_ERR_UNSUPPORTED;
    if (n_rays * (int64_t)field->n_samples > LS2FM_MAX_RENDER_POINTS) return LS2FM_ERR_UNSUPPORTED;
    const ls2fm_loss_spec* loss = opts ? opts->loss : nullptr;
    LS2FM_CHECK_ARG(!loss || (loss->rgb_gt && loss->weights && loss->terms && loss->sums));
    if (n_rays == 0) return LS2FM_OK;
    LS2FM_CHECK_ARG(center && ray && rgb && sdfs_volume && normals && depth_mlp && normal_mlp);
    if (!workspace) return LS2FM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int dual = field->dual_field ? 1 : 0;
    const bool prepare_bwd = !(opts && opts->inference_only);
    if (prepare_bwd && !ls2fm_bins_levels_fit(sdf_grid, dual)) return LS2FM_ERR_UNSUPPORTED;
    const int L1 = sdf_grid->n_levels, L2 = dual ? rad_grid->n_levels : 0;
    const WsLayout w = make_ws_layout(n_rays, field->n_samples, L1, dual ? L2 : L1, dual);
    float* ws = (float*)workspace;
    Packed* pk = (Packed*)(ws + w.packed);
    const FieldC fc = make_field_c(field);

    // ---- gather pass (+ weight prep in its first workgroups, + the scatter's item counts when a backward follows)
    const int n_chunks = (int)((w.p + kEncThreads - 1) / kEncThreads);
    const bool interleaved = dual && params->dual_table;
    const bool pair = dual && !interleaved;      // both grids, one launch
    const int enc_span = dual ? LS2FM_PROF_ENCODE_PAIR : LS2FM_PROF_ENCODE_SDF;       // both grids in this launch (two tables or the interleaved copy)
    int most = 0;
    const XcdPlan plan = make_xcd_plan(sdf_grid, L1, pair ? rad_grid : nullptr, pair ? L2 : 0, field->n_samples, n_chunks, prepare_bwd, &most,
                                       interleaved);
    EncodeExtras ex;
    ex.params = *params;
    ex.in_dim = 3 + 2 * L1; ex.in_dim2 = 3 + 2 * L2; ex.rad_in = 3 + 3 + kView + LS2FM_FEAT * (dual ? 2 : 1); ex.dual = dual;
    ex.packed = pk;
    ex.tile_counts = nullptr; ex.scan_ticket = nullptr; ex.n_tiles = 0; ex.sshift = ls2fm_slab_shift(dual);
    ex.n_explicit = ls2fm_explicit_levels(sdf_grid, dual, field->n_samples);
    ex.n_prep_tasks = 3; ex.pts = nullptr;
    static const int enc_probe = [] { const char* e = getenv("LS2FM_ENC_PROBE"); return e ? atoi(e) : 0; }();
#ifdef LS2FM_ENC_PROBES
    ex.probe = enc_probe;
#else
    ex.probe = 0; (void)enc_probe;
#endif
    if (ex.probe == 1) most = (n_chunks + 7) / 8 * (L1 + (pair ? L2 : 0));
    if (prepare_bwd) {
        const BinMeta bm = make_bin_meta(ws + w.bins, w.p);
        ex.tile_counts = bm.tile;
        ex.scan_ticket = scan_ticket(bm);
        ex.n_tiles = bm.n_tiles;
    }
    const unsigned enc_blocks = (unsigned)(kEncReserved + 8 * most);
    ls2fm_prof_begin(enc_span, s);
    if (interleaved)
        ray_encode_kernel<true><<<enc_blocks, kEncThreads, 0, s>>>(
            make_level_set(sdf_grid), make_level_set(sdf_grid), fc, center, ray, params->dual_table, nullptr, w.p, w.p_pad, n_chunks,
            plan, ws + w.e1, ws + w.e2, ws + w.j1, ex);
    else
        ray_encode_kernel<false><<<enc_blocks, kEncThreads, 0, s>>>(
            make_level_set(sdf_grid), make_level_set(pair ? rad_grid : sdf_grid), fc, center, ray, params->sdf_table,
            pair ? params->rad_table : nullptr, w.p, w.p_pad, n_chunks, plan, ws + w.e1, pair ? ws + w.e2 : nullptr, ws + w.j1, ex);
    ls2fm_prof_end(enc_span, s);

    // ---- shading, then ONE small launch: the fused loss head's reduction and the scans of the item counts side by side (no
    // stream fork anywhere in the forward: the main chain stays on one queue)
    ls2fm_prof_begin(LS2FM_PROF_SHADE_FWD, s);
    ls2fm_launch_shade_fwd(fc, dual, 2 * L1, 2 * L2, pk, center, ray, n_rays, w, ws, rgb, sdfs_volume, normals, depth_mlp,
                           normal_mlp, loss, s);
    ls2fm_prof_end(LS2FM_PROF_SHADE_FWD, s);
    // the traced depth and its mask may come from another stream: waited for here, behind the gather pass AND the shading
    if (loss && opts->loss_inputs_ready && hipStreamWaitEvent(s, (hipEvent_t)opts->loss_inputs_ready, 0) != hipSuccess)
        return LS2FM_ERR_LAUNCH;
    ls2fm_prof_begin(LS2FM_PROF_BIN, s);
    const int st = ls2fm_launch_post_shade(loss, ws + w.lpart, n_rays, field->n_samples, prepare_bwd ? sdf_grid : nullptr, w.p,
                                           ws + w.bins, s);
    ls2fm_prof_end(LS2FM_PROF_BIN, s);
    if (st != LS2FM_OK) return st;
    return ls2fm_launch_status();
}
