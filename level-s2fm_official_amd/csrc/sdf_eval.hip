// Fused no-autograd SDF evaluation and the sphere-tracing root-find loop for gfx950.
//
//   sdf_eval      thread per point: 16 x 8 hash-grid gathers + 35->64 softplus(100) -> 64->17 MLP (weights as
//                 wave-uniform scalar loads), sign / scale_mlp, optional background-sphere min, optional
//                 analytic normal (second gather pass contracting the hash Jacobian with W0^T (s' * w1_0))
//                 Replaces SDF.infer_sdf (models/SDF.py:55-78) where no graph is needed.
//   sphere_trace  two lanes per ray (start end / far end), the reference's `while True` (models/SDF.py:149-200)
//                 run to completion per ray without host round trips; the global trip count K of the reference is
//                 recovered as max over rays of the first trip at which the ray's start end is finished.
#include <cstdlib>

#include "render_common.h"

int ls2fm_launch_prep_sdf(const ls2fm_params* params, int n_levels, Packed* out, hipStream_t stream, int32_t* zero_word = nullptr);

namespace {

// hash-encode a world point into u[35] = [p / rescale, e(x)]
__device__ __forceinline__ void encode_point(const LevelSet& lv, const FieldC& fc, const float* __restrict__ table,
                                             const float p[3], float (&u)[kInMax]) {
    float x[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        x[a] = (p[a] - fc.bmin[a]) / (fc.bmax[a] - fc.bmin[a]);
        u[a] = p[a] / fc.rescale;
    }
#pragma unroll
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) {
        float y0 = 0.f, y1 = 0.f;
        if (l < lv.n_levels) {
            Cell c;
            locate(x, lv.scale[l], lv.res[l], lv.size[l], lv.offset[l], lv.hashed[l], c);
            float2 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float2*>(table + 2ull * c.idx[k]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float wt = corner_weight(c.w, k);
                y0 = fmaf(wt, v[k].x, y0);
                y1 = fmaf(wt, v[k].y, y1);
            }
        }
        u[3 + 2 * l] = y0;
        u[4 + 2 * l] = y1;
    }
}

__device__ __forceinline__ float signed_sdf(const FieldC& fc, int bg_sdf, float bg_rad, float f0, const float p[3],
                                            bool* bg_selected) {
    float sdf = fc.inside ? f0 / fc.scale_mlp : -f0 / fc.scale_mlp;
    *bg_selected = false;
    if (bg_sdf) {
        const float other = bg_rad - sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        if (other < sdf) { sdf = other; *bg_selected = true; }
    }
    return sdf;
}

// Lattice of a volume sweep, generated on the device (ls2fm_sdf_volume).  ref_indexing reproduces the host arithmetic
// of the reference's extract_mesh (utils/util.py:399-409), fp64 then rounded to fp32 like its `.float()`: with
// idx = x N^2 + y N + z the reference divides with numpy's TRUE division, so the three index columns are
// fmod(idx/N/N, N), fmod(idx/N, N), idx % N -- fractional in the first two (a sheared lattice), kept as is.
struct Lattice {
    int64_t n_side, first;
    double step[3], origin[3];
    int ref_indexing;
};

__device__ __forceinline__ void lattice_point(const Lattice& lat, int64_t i, float p[3]) {
    const int64_t idx = lat.first + i;
    const double n = (double)lat.n_side;
    double f[3];
    if (lat.ref_indexing) {
        const double q = (double)idx / n;
        f[0] = fmod(q / n, n);
        f[1] = fmod(q, n);
        f[2] = (double)(idx % lat.n_side);
    } else {
        f[0] = (double)(idx / (lat.n_side * lat.n_side));
        f[1] = (double)((idx / lat.n_side) % lat.n_side);
        f[2] = (double)(idx % lat.n_side);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) p[a] = (float)(f[a] * lat.step[a] + lat.origin[a]);
}

template <bool WANT_NORMAL>
__global__ void __launch_bounds__(256)
sdf_eval_kernel(LevelSet lv, FieldC fc, int bg_sdf, float bg_rad, const Packed* __restrict__ pk,
                const float* __restrict__ table, const float* __restrict__ pts, Lattice lat, int64_t n,
                float* __restrict__ sdf_out, float* __restrict__ feat_out, float* __restrict__ normal_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float p[3];
    if (pts) {
        p[0] = pts[i * 3]; p[1] = pts[i * 3 + 1]; p[2] = pts[i * 3 + 2];
    } else {
        lattice_point(lat, i, p);
    }
    float u[kInMax], f[kOut], rr[kInMax];
    encode_point(lv, fc, table, p, u);
    geometry_forward<WANT_NORMAL>(pk->sdf, u, f, rr);
    bool bg;
    sdf_out[i] = signed_sdf(fc, bg_sdf, bg_rad, f[0], p, &bg);
    if (feat_out) {
#pragma unroll
        for (int o = 0; o < kOut; ++o) feat_out[i * kOut + o] = f[o];
    }
    if (WANT_NORMAL) {
        float acc[3] = {0.f, 0.f, 0.f};
        float x[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) x[a] = (p[a] - fc.bmin[a]) / (fc.bmax[a] - fc.bmin[a]);
#pragma unroll
        for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) {
            if (l < lv.n_levels) {
                Cell c;
                locate(x, lv.scale[l], lv.res[l], lv.size[l], lv.offset[l], lv.hashed[l], c);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float2 v = *reinterpret_cast<const float2*>(table + 2ull * c.idx[k]);
                    const float m = fmaf(v.x, rr[3 + 2 * l], v.y * rr[4 + 2 * l]) * lv.scale[l];
#pragma unroll
                    for (int a = 0; a < 3; ++a) acc[a] = fmaf(corner_dweight(c.w, k, a), m, acc[a]);
                }
            }
        }
        const float len = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float nrm = fc.kappa * (rr[a] / fc.rescale + acc[a] * fc.inv_ext[a]);
            normal_out[i * 3 + a] = bg ? -p[a] / len : nrm;
        }
    }
}

// two lanes per ray: side 0 = start end (from near), side 1 = far end
__global__ void __launch_bounds__(256)
sphere_trace_kernel(LevelSet lv, FieldC fc, int bg_sdf, float bg_rad, const Packed* __restrict__ pk,
                    const float* __restrict__ table, const float* __restrict__ ray0, const float* __restrict__ ray_dir,
                    int64_t n_rays, float thr, int iters_max, float* __restrict__ near_out, float* __restrict__ far_out,
                    float* __restrict__ track, float* __restrict__ t_end, float* __restrict__ track_sdf, int* __restrict__ trips) {
    const int64_t tidg = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = tidg >> 1;
    const int side = (int)(tidg & 1);
    const bool live = r < n_rays;
    const int64_t rr_ = live ? r : n_rays - 1;
    const RayGeom g = load_ray(fc, ray0, ray_dir, rr_);
    const float far = g.t_far;
    float t_me = side ? g.t_far : g.t_near;
    float p[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) p[a] = g.o[a] + t_me * g.d[a];
    float u[kInMax], f[kOut], dummy[kInMax];
    bool bg;
    encode_point(lv, fc, table, p, u);
    geometry_forward<false>(pk->sdf, u, f, dummy);
    float sdf_me = signed_sdf(fc, bg_sdf, bg_rad, f[0], p, &bg);
    float raw_me = sdf_me;               // the field's value at the current point (sdf_me is zeroed / goes stale: (1), (6))
    const bool want_raw = track_sdf != nullptr && side == 0;
    if (live && side == 0) { near_out[r] = g.t_near; far_out[r] = g.t_far; }
    if (live && side == 1) t_end[r * (iters_max + 1)] = t_me;
    bool unf = false;
    int kfin = -1;
    for (int k = 0;; ++k) {
        if (fabsf(sdf_me) <= thr) sdf_me = 0.f;                          // (1) converged values are zeroed
        const bool m = fabsf(sdf_me) > thr;
        unf = k == 0 ? m : (unf && m);                                    // (2)
        const int unf_start = __shfl((int)unf, (threadIdx.x & 63) & ~1, 64);
        if (!unf_start && kfin < 0) kfin = k;                             // (3) this ray's start end is done at trip k
        if (k == iters_max) break;
        if (live && side == 0) {                                          // (5) pre-update start point -> track
#pragma unroll
            for (int a = 0; a < 3; ++a) track[(r * (iters_max + 1) + k) * 3 + a] = p[a];
            if (track_sdf) track_sdf[r * (iters_max + 1) + k] = raw_me;
        }
        const float t_before = t_me;
        t_me = t_me + sdf_me;                                             // (4) both ends step with '+', clamp to far
        if (t_me > far) t_me = far;
#pragma unroll
        for (int a = 0; a < 3; ++a) p[a] = g.o[a] + t_me * g.d[a];
        // (6) refresh only where unfinished; for the track's values also where a finished start end still moved (crossed
        // ends keep stepping with their stale value: the reference evaluates those points afterwards, SDF.py:203)
        if (unf || (want_raw && t_me != t_before)) {
            encode_point(lv, fc, table, p, u);
            geometry_forward<false>(pk->sdf, u, f, dummy);
            raw_me = signed_sdf(fc, bg_sdf, bg_rad, f[0], p, &bg);
            if (unf) sdf_me = raw_me;
        }
        const float t_other = __shfl_xor(t_me, 1, 64);
        const float t_s = side ? t_other : t_me, t_e = side ? t_me : t_other;
        unf = unf && (t_s < t_e);                                         // (7) crossed ends drop out
        if (live && side == 1) t_end[r * (iters_max + 1) + k + 1] = t_me;
    }
    if (live && side == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) track[(r * (iters_max + 1) + iters_max) * 3 + a] = p[a];
        if (track_sdf) track_sdf[r * (iters_max + 1) + iters_max] = raw_me;
    }
    // global trip count = max over rays: one atomic per wave (same-address global atomics serialise: ~6 ns each)
    int kmax = (live && side == 0) ? (kfin < 0 ? iters_max : kfin) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o, 64));
    if ((threadIdx.x & 63) == 0 && kmax > 0) atomicMax(trips, kmax);
}

// ---- sphere tracing, one 16-lane group per ray END (four ends = two rays per wave).  The thread-per-end kernel above is
// bound by the latency of its serial chain -- 11 SDF evaluations, each 16 levels of dependent gathers and 3.3 k scalar-operand
// FMAs in ONE lane: 477 us for 8192 rays with one wave per CU.  Here lane jl of a group gathers level jl, the 35 inputs are
// exchanged by shuffles, the lane evaluates hidden units jl, jl + 16, jl + 32, jl + 48 from an LDS copy of W0 and the sdf row is summed
// in the thread-per-end kernel's order (bit-identical results): the same loop (steps 1-7 below are those of sphere_trace_kernel) with a ~10x shorter step.
constexpr int kW0Stride = 36;       // W0[j][0..34], b0[j] : 16-byte aligned rows, 2-way bank conflicts at most

// f += w[J] * (lane J of this lane's 16-lane row).h for J = 0 .. 15, one fmaf chain; the broadcast is a DPP row_share operand
// (full-rate VALU), not an LDS permute
template <int J>
__device__ __forceinline__ float row_chain(const float* __restrict__ w, float h, float f) {
    if constexpr (J < 16) {
        const float hv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(h), 0x150 + J, 0xF, 0xF, false));
        return row_chain<J + 1>(w, h, fmaf(w[J], hv, f));
    } else {
        return f;
    }
}

__device__ __forceinline__ float group_sdf(const LevelSet& lv, const FieldC& fc, int bg_sdf, float bg_rad,
                                           const float* __restrict__ table, const float* __restrict__ s_w0,
                                           const float* __restrict__ s_w1, float b1_0, const float p[3], int jl, int gbase) {
    float x[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) x[a] = (p[a] - fc.bmin[a]) / (fc.bmax[a] - fc.bmin[a]);
    float y0 = 0.f, y1 = 0.f;
    if (jl < lv.n_levels) {
        Cell c;
        locate(x, lv.scale[jl], lv.res[jl], lv.size[jl], lv.offset[jl], lv.hashed[jl], c);
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float2*>(table + 2ull * c.idx[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float wt = corner_weight(c.w, k);
            y0 = fmaf(wt, v[k].x, y0);
            y1 = fmaf(wt, v[k].y, y1);
        }
    }
    float u[kInMax];
#pragma unroll
    for (int a = 0; a < 3; ++a) u[a] = p[a] / fc.rescale;
#pragma unroll
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) {
        u[3 + 2 * l] = __shfl(y0, gbase + l, 64);
        u[4 + 2 * l] = __shfl(y1, gbase + l, 64);
    }
    float h[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float* __restrict__ w = s_w0 + (jl + 16 * q) * kW0Stride;
        float a0 = w[kRecB0], a1 = 0.0f;           // same two-chain order as geometry_forward
#pragma unroll
        for (int k = 0; k + 1 < kInMax; k += 2) {
            a0 = fmaf(w[k], u[k], a0);
            a1 = fmaf(w[k + 1], u[k + 1], a1);
        }
        a0 = fmaf(w[kInMax - 1], u[kInMax - 1], a0);
        h[q] = softplus100_value(a0 + a1);
    }
    // the sdf row in geometry_forward's order (j = 0 .. 63, one fmaf chain): every lane of the group forms the same sum, and
    // the result is BIT-IDENTICAL to the thread-per-point kernels (a point evaluates the same in a small and in a large call)
    float f0 = b1_0;
#pragma unroll
    for (int q = 0; q < 4; ++q) f0 = row_chain<0>(s_w1 + 16 * q, h[q], f0);
    bool bg;
    return signed_sdf(fc, bg_sdf, bg_rad, f0, p, &bg);
}

// sdf only, 16 lanes per point: the latency-bound small calls of infer_sdf(mode="ret_sdf") (thread-per-point: ~90 us however
// few the points)
__global__ void __launch_bounds__(256)
sdf_eval_wide_kernel(LevelSet lv, FieldC fc, int bg_sdf, float bg_rad, const Packed* __restrict__ pk,
                     const float* __restrict__ table, const float* __restrict__ pts, int64_t n, float* __restrict__ sdf_out) {
    __shared__ float s_w0[kHidden * kW0Stride];
    __shared__ float s_w1[kHidden];
    for (int q = threadIdx.x; q < kHidden * kW0Stride; q += 256) s_w0[q] = pk->sdf[(q / kW0Stride) * kRecStride + q % kW0Stride];
    for (int q = threadIdx.x; q < kHidden; q += 256) s_w1[q] = pk->sdf[q * kRecStride + kRecW1];
    const float b1_0 = pk->sdf[kHidden * kRecStride];
    __syncthreads();
    const int lane = threadIdx.x & 63, jl = lane & 15, gbase = lane & 48;
    const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int64_t ii = i < n ? i : n - 1;
    const float p[3] = {pts[ii * 3], pts[ii * 3 + 1], pts[ii * 3 + 2]};
    const float v = group_sdf(lv, fc, bg_sdf, bg_rad, table, s_w0, s_w1, b1_0, p, jl, gbase);
    if (i < n && jl == 0) sdf_out[i] = v;
}

// ---- sdf + the 16 features + the analytic normal, 16 lanes per point (up to 32 768 points): the point queries of the stage loops (SDF.gradient /
// get_surface_pts on a few thousand key points: the thread-per-point kernel is one 80 us latency chain however few the points).
// Lane jl of a group gathers level jl and keeps its eight corner values; hidden units jl, jl + 16, .. as in group_sdf; then ONE
// pass over j = 0 .. 63 with unit j's h and s' w1_0 broadcast by DPP, in which the lane advances the sdf row (every lane),
// feature row 1 + jl and r[jl], r[16 + jl], r[32 + jl] -- each the same fmaf chain over j as geometry_forward's; the normal's
// sum over (level, corner) runs through the lanes in level order.  Every sum in the thread-per-point order: BIT-IDENTICAL outputs.
constexpr int kW1Stride = kOut;         // 17: odd, the 16 lanes' feature weights of one unit are conflict-free

template <bool WANT_R, int J>
__device__ __forceinline__ void full_chain(const float* __restrict__ w0, const float* __restrict__ w1, float h, float g, int jl,
                                           float& f0, float& fm, float (&r)[3]) {
    if constexpr (J < 16) {
        const float hv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(h), 0x150 + J, 0xF, 0xF, false));
        const float* __restrict__ w1j = w1 + J * kW1Stride;
        f0 = fmaf(w1j[0], hv, f0);
        fm = fmaf(w1j[1 + jl], hv, fm);
        if (WANT_R) {
            const float gv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(g), 0x150 + J, 0xF, 0xF, false));
            const float* __restrict__ w0j = w0 + J * kW0Stride;
            r[0] = fmaf(w0j[jl], gv, r[0]);
            r[1] = fmaf(w0j[16 + jl], gv, r[1]);
            r[2] = fmaf(w0j[32 + jl], gv, r[2]);            // k = 32 + jl: an input only for jl < 3 (the others are never read)
        }
        full_chain<WANT_R, J + 1>(w0, w1, h, g, jl, f0, fm, r);
    }
}

template <bool WANT_NORMAL>
__global__ void __launch_bounds__(256)
sdf_eval_wide_full_kernel(LevelSet lv, FieldC fc, int bg_sdf, float bg_rad, const Packed* __restrict__ pk,
                          const float* __restrict__ table, const float* __restrict__ pts, int64_t n, float* __restrict__ sdf_out,
                          float* __restrict__ feat_out, float* __restrict__ normal_out) {
    __shared__ float s_w0[kHidden * kW0Stride + 16];          // + 16: the k = 32 + jl reads of the last row stay inside
    __shared__ float s_w1[kHidden * kW1Stride];
    __shared__ float s_r[16][48];                             // r[0 .. 34] of the group's point
    for (int q = threadIdx.x; q < kHidden * kW0Stride; q += 256) s_w0[q] = pk->sdf[(q / kW0Stride) * kRecStride + q % kW0Stride];
    for (int q = threadIdx.x; q < kHidden * kW1Stride; q += 256) s_w1[q] = pk->sdf[(q / kW1Stride) * kRecStride + kRecW1 + q % kW1Stride];
    if (threadIdx.x < 16) s_w0[kHidden * kW0Stride + threadIdx.x] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, jl = lane & 15, gbase = lane & 48, grp = threadIdx.x >> 4;
    const int64_t i = (int64_t)blockIdx.x * 16 + grp;
    const int64_t ii = i < n ? i : n - 1;
    const float p[3] = {pts[ii * 3], pts[ii * 3 + 1], pts[ii * 3 + 2]};
    float x[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) x[a] = (p[a] - fc.bmin[a]) / (fc.bmax[a] - fc.bmin[a]);
    // ---- this lane's level: corner values (kept for the normal) and the trilinear value
    const bool has_level = jl < lv.n_levels;
    float2 v[8];
    Cell c;
    float y0 = 0.f, y1 = 0.f, scale_l = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = make_float2(0.f, 0.f);
    c.w[0] = c.w[1] = c.w[2] = 0.f;
    if (has_level) {
        scale_l = lv.scale[jl];
        locate(x, scale_l, lv.res[jl], lv.size[jl], lv.offset[jl], lv.hashed[jl], c);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float2*>(table + 2ull * c.idx[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float wt = corner_weight(c.w, k);
            y0 = fmaf(wt, v[k].x, y0);
            y1 = fmaf(wt, v[k].y, y1);
        }
    }
    float u[kInMax];
#pragma unroll
    for (int a = 0; a < 3; ++a) u[a] = p[a] / fc.rescale;
#pragma unroll
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) {
        u[3 + 2 * l] = __shfl(y0, gbase + l, 64);
        u[4 + 2 * l] = __shfl(y1, gbase + l, 64);
    }
    // ---- hidden units jl + 16 q, each block followed by its 16 steps of the rows over j = 0 .. 63 (a run-time loop over q: fully
    // unrolled, the scheduler hoists all 464 LDS weight reads to the top -- 426 registers, or a kilobyte of scratch under a cap)
    float f0 = pk->sdf[kHidden * kRecStride], fm = pk->sdf[kHidden * kRecStride + 1 + jl], r[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const float* __restrict__ w = s_w0 + (jl + 16 * q) * kW0Stride;
        float a0 = w[kRecB0], a1 = 0.0f;           // same two-chain order as geometry_forward
#pragma unroll
        for (int k = 0; k + 1 < kInMax; k += 2) {
            a0 = fmaf(w[k], u[k], a0);
            a1 = fmaf(w[k + 1], u[k + 1], a1);
        }
        a0 = fmaf(w[kInMax - 1], u[kInMax - 1], a0);
        float h, s1, s2;
        softplus100(a0 + a1, h, s1, s2);
        const float g = s1 * s_w1[(jl + 16 * q) * kW1Stride];
        full_chain<WANT_NORMAL, 0>(s_w0 + 16 * q * kW0Stride, s_w1 + 16 * q * kW1Stride, h, g, jl, f0, fm, r);
    }
    bool bg;
    const float sdf = signed_sdf(fc, bg_sdf, bg_rad, f0, p, &bg);
    const bool live = i < n;
    if (live && jl == 0) sdf_out[i] = sdf;
    if (live && feat_out) {
        if (jl == 0) feat_out[i * kOut] = f0;
        feat_out[i * kOut + 1 + jl] = fm;
    }
    if (WANT_NORMAL) {
        s_r[grp][jl] = r[0];
        s_r[grp][16 + jl] = r[1];
        s_r[grp][32 + jl] = r[2];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const float rr0 = s_r[grp][3 + 2 * jl], rr1 = s_r[grp][4 + 2 * jl];
        // this level's eight terms: m_k = (v.x rr0 + v.y rr1) scale, weights d w_k / d x_a
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = fmaf(v[k].x, rr0, v[k].y * rr1) * scale_l;
        float acc[3] = {0.f, 0.f, 0.f};
        for (int l = 0; l < lv.n_levels; ++l) {              // stage l: lane l's level continues the chain
            float t[3] = {acc[0], acc[1], acc[2]};
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int a = 0; a < 3; ++a) t[a] = fmaf(corner_dweight(c.w, k, a), m[k], t[a]);
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[a] = __shfl(t[a], gbase + l, 64);
        }
        if (live && jl < 3) {
            const float len = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
            const float pa = jl == 0 ? p[0] : (jl == 1 ? p[1] : p[2]);
            const float acc_a = jl == 0 ? acc[0] : (jl == 1 ? acc[1] : acc[2]);
            const float inv_a = jl == 0 ? fc.inv_ext[0] : (jl == 1 ? fc.inv_ext[1] : fc.inv_ext[2]);
            const float nrm = fc.kappa * (s_r[grp][jl] / fc.rescale + acc_a * inv_a);
            normal_out[i * 3 + jl] = bg ? -pa / len : nrm;
        }
    }
}

__global__ void __launch_bounds__(256)
sphere_trace_wide_kernel(LevelSet lv, FieldC fc, int bg_sdf, float bg_rad, const Packed* __restrict__ pk,
                         const float* __restrict__ table, const float* __restrict__ ray0, const float* __restrict__ ray_dir,
                         int64_t n_rays, float thr, int iters_max, float* __restrict__ near_out, float* __restrict__ far_out,
                         float* __restrict__ track, float* __restrict__ t_end, float* __restrict__ track_sdf, int* __restrict__ trips) {
    __shared__ float s_w0[kHidden * kW0Stride];
    __shared__ float s_w1[kHidden];
    for (int q = threadIdx.x; q < kHidden * kW0Stride; q += 256) s_w0[q] = pk->sdf[(q / kW0Stride) * kRecStride + q % kW0Stride];
    for (int q = threadIdx.x; q < kHidden; q += 256) s_w1[q] = pk->sdf[q * kRecStride + kRecW1];
    const float b1_0 = pk->sdf[kHidden * kRecStride];
    __syncthreads();
    const int lane = threadIdx.x & 63, jl = lane & 15, gq = lane >> 4, gbase = lane & 48;
    const int side = gq & 1;
    const int64_t r = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (gq >> 1);
    const bool live = r < n_rays;
    const int64_t rr_ = live ? r : n_rays - 1;
    const RayGeom g = load_ray(fc, ray0, ray_dir, rr_);
    const float far = g.t_far;
    float t_me = side ? g.t_far : g.t_near;
    float p[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) p[a] = g.o[a] + t_me * g.d[a];
    float sdf_me = group_sdf(lv, fc, bg_sdf, bg_rad, table, s_w0, s_w1, b1_0, p, jl, gbase);
    float raw_me = sdf_me;               // the field's value at the current point (see sphere_trace_kernel)
    const bool want_raw = track_sdf != nullptr && side == 0;
    const bool writer = live && jl == 0;
    if (writer && side == 0) { near_out[r] = g.t_near; far_out[r] = g.t_far; }
    if (writer && side == 1) t_end[r * (iters_max + 1)] = t_me;
    bool unf = false;
    int kfin = -1;
    for (int k = 0;; ++k) {
        if (fabsf(sdf_me) <= thr) sdf_me = 0.f;                          // (1) converged values are zeroed
        const bool m = fabsf(sdf_me) > thr;
        unf = k == 0 ? m : (unf && m);                                    // (2)
        const int unf_start = __shfl((int)unf, lane & ~16, 64);           // the start end's group of this ray
        if (!unf_start && kfin < 0) kfin = k;                             // (3) this ray's start end is done at trip k
        if (k == iters_max) break;
        if (writer && side == 0) {                                        // (5) pre-update start point -> track
#pragma unroll
            for (int a = 0; a < 3; ++a) track[(r * (iters_max + 1) + k) * 3 + a] = p[a];
            if (track_sdf) track_sdf[r * (iters_max + 1) + k] = raw_me;
        }
        const float t_before = t_me;
        t_me = t_me + sdf_me;                                             // (4) both ends step with '+', clamp to far
        if (t_me > far) t_me = far;
#pragma unroll
        for (int a = 0; a < 3; ++a) p[a] = g.o[a] + t_me * g.d[a];
        if (unf || (want_raw && t_me != t_before)) {                      // (6) group-uniform
            // compiler barrier: without it the 144 weight values a lane reads from LDS per evaluation are hoisted out of the
            // trip loop into registers (256 VGPRs + 44 AGPRs, one wave per SIMD) -- and a kernel running BESIDE this one
            // (ls2fm.stage: the render's gather pass) loses more than half of its waves on every SIMD this kernel sits on
            __asm__ volatile("" ::: "memory");
            raw_me = group_sdf(lv, fc, bg_sdf, bg_rad, table, s_w0, s_w1, b1_0, p, jl, gbase);
            if (unf) sdf_me = raw_me;
        }
        const float t_other = __shfl_xor(t_me, 16, 64);
        const float t_s = side ? t_other : t_me, t_e = side ? t_me : t_other;
        unf = unf && (t_s < t_e);                                         // (7) crossed ends drop out
        if (writer && side == 1) t_end[r * (iters_max + 1) + k + 1] = t_me;
    }
    if (writer && side == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) track[(r * (iters_max + 1) + iters_max) * 3 + a] = p[a];
        if (track_sdf) track_sdf[r * (iters_max + 1) + iters_max] = raw_me;
    }
    int kmax = (writer && side == 0) ? (kfin < 0 ? iters_max : kfin) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kmax = max(kmax, __shfl_xor(kmax, o, 64));
    if (lane == 0 && kmax > 0) atomicMax(trips, kmax);
}

// ---- sphere tracing, one WAVE per ray end (a 128-thread workgroup = one ray).  At the stage loops' ray counts (1024 rays) the
// 16-lane kernel above leaves the chip almost idle (one wave on half the SIMDs) and spends its step on 144 LDS weight reads and
// 140 FMAs per lane.  Here lane j owns hidden unit j with its W0 row in registers for the whole trace, lanes 0-15 gather the 16
// levels, the 32 encoding values are broadcast as scalars (v_readlane) and the sdf row is the same j = 0 .. 63 fmaf chain fed by
// v_readlane -- every sum in the order of geometry_forward, so the result is BIT-IDENTICAL to the other two tracing kernels.
// The two ends of a ray exchange their parameter through LDS once per trip (step 7).
__device__ __forceinline__ float lane_bcast(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

template <int J>
__device__ __forceinline__ float wave_chain(float w1_me, float h, float f) {
    if constexpr (J < kHidden) {
        const float hv = lane_bcast(h, J);
        const float wv = lane_bcast(w1_me, J);
        return wave_chain<J + 1>(w1_me, h, fmaf(wv, hv, f));
    } else {
        return f;
    }
}

__device__ __forceinline__ float wave_sdf(const LevelSet& lv, const FieldC& fc, int bg_sdf, float bg_rad,
                                          const float* __restrict__ table, const float (&w)[kInMax + 1], float w1_me, float b1_0,
                                          const float p[3], int lane) {
    float x[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) x[a] = (p[a] - fc.bmin[a]) / (fc.bmax[a] - fc.bmin[a]);
    float y0 = 0.f, y1 = 0.f;
    if (lane < lv.n_levels) {
        Cell c;
        locate(x, lv.scale[lane], lv.res[lane], lv.size[lane], lv.offset[lane], lv.hashed[lane], c);
        float2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float2*>(table + 2ull * c.idx[k]);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float wt = corner_weight(c.w, k);
            y0 = fmaf(wt, v[k].x, y0);
            y1 = fmaf(wt, v[k].y, y1);
        }
    }
    float u[kInMax];
#pragma unroll
    for (int a = 0; a < 3; ++a) u[a] = p[a] / fc.rescale;
#pragma unroll
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) {
        u[3 + 2 * l] = lane_bcast(y0, l);
        u[4 + 2 * l] = lane_bcast(y1, l);
    }
    float a0 = w[kRecB0], a1 = 0.0f;               // same two-chain order as geometry_forward
#pragma unroll
    for (int k = 0; k + 1 < kInMax; k += 2) {
        a0 = fmaf(w[k], u[k], a0);
        a1 = fmaf(w[k + 1], u[k + 1], a1);
    }
    a0 = fmaf(w[kInMax - 1], u[kInMax - 1], a0);
    const float h = softplus100_value(a0 + a1);
    const float f0 = wave_chain<0>(w1_me, h, b1_0);
    bool bg;
    return signed_sdf(fc, bg_sdf, bg_rad, f0, p, &bg);
}

__global__ void __launch_bounds__(128)
sphere_trace_wave_kernel(LevelSet lv, FieldC fc, int bg_sdf, float bg_rad, const Packed* __restrict__ pk,
                         const float* __restrict__ table, const float* __restrict__ ray0, const float* __restrict__ ray_dir,
                         int64_t n_rays, float thr, int iters_max, float* __restrict__ near_out, float* __restrict__ far_out,
                         float* __restrict__ track, float* __restrict__ t_end, float* __restrict__ track_sdf, int* __restrict__ trips) {
    __shared__ float s_t[2][2];
    const int lane = threadIdx.x & 63, side = threadIdx.x >> 6;
    const int64_t r = blockIdx.x;                  // the grid is n_rays workgroups
    float w[kInMax + 1];
#pragma unroll
    for (int k = 0; k <= kInMax; ++k) w[k] = pk->sdf[lane * kRecStride + k];
    const float w1_me = pk->sdf[lane * kRecStride + kRecW1];
    const float b1_0 = pk->sdf[kHidden * kRecStride];
    const RayGeom g = load_ray(fc, ray0, ray_dir, r);
    const float far = g.t_far;
    float t_me = side ? g.t_far : g.t_near;
    float p[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) p[a] = g.o[a] + t_me * g.d[a];
    float sdf_me = wave_sdf(lv, fc, bg_sdf, bg_rad, table, w, w1_me, b1_0, p, lane);
    float raw_me = sdf_me;               // the field's value at the current point (see sphere_trace_kernel)
    const bool want_raw = track_sdf != nullptr && side == 0;
    const bool writer = lane == 0;
    if (writer && side == 0) { near_out[r] = g.t_near; far_out[r] = g.t_far; }
    if (writer && side == 1) t_end[r * (iters_max + 1)] = t_me;
    bool unf = false;
    int kfin = -1;
    for (int k = 0;; ++k) {
        if (fabsf(sdf_me) <= thr) sdf_me = 0.f;                          // (1) converged values are zeroed
        const bool m = fabsf(sdf_me) > thr;
        unf = k == 0 ? m : (unf && m);                                    // (2)
        if (side == 0 && !unf && kfin < 0) kfin = k;                      // (3) this ray's start end is done at trip k
        if (k == iters_max) break;
        if (writer && side == 0) {                                        // (5) pre-update start point -> track
#pragma unroll
            for (int a = 0; a < 3; ++a) track[(r * (iters_max + 1) + k) * 3 + a] = p[a];
            if (track_sdf) track_sdf[r * (iters_max + 1) + k] = raw_me;
        }
        const float t_before = t_me;
        t_me = t_me + sdf_me;                                             // (4) both ends step with '+', clamp to far
        if (t_me > far) t_me = far;
#pragma unroll
        for (int a = 0; a < 3; ++a) p[a] = g.o[a] + t_me * g.d[a];
        if (unf || (want_raw && t_me != t_before)) {                      // (6) wave-uniform
            raw_me = wave_sdf(lv, fc, bg_sdf, bg_rad, table, w, w1_me, b1_0, p, lane);
            if (unf) sdf_me = raw_me;
        }
        if (writer) s_t[k & 1][side] = t_me;                              // the other end's parameter: one barrier per trip
        __syncthreads();                                                  // (slot k & 1 is rewritten two barriers later)
        const float t_other = s_t[k & 1][side ^ 1];
        const float t_s = side ? t_other : t_me, t_e = side ? t_me : t_other;
        unf = unf && (t_s < t_e);                                         // (7) crossed ends drop out
        if (writer && side == 1) t_end[r * (iters_max + 1) + k + 1] = t_me;
    }
    if (writer && side == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) track[(r * (iters_max + 1) + iters_max) * 3 + a] = p[a];
        if (track_sdf) track_sdf[r * (iters_max + 1) + iters_max] = raw_me;
        const int kmax = kfin < 0 ? iters_max : kfin;
        if (kmax > 0) atomicMax(trips, kmax);
    }
}

}  // namespace

static bool field_ok(const ls2fm_field_desc* f, const ls2fm_grid_desc* g, const ls2fm_params* p) {
    return f && grid_desc_ok(g) && p && p->sdf_table && p->beta && p->sdf_mlp[0].weight_v && p->sdf_mlp[1].weight_v;
}

extern "C" int64_t ls2fm_sdf_eval_workspace_bytes(void) { return (int64_t)((sizeof(Packed) + 255) / 256 * 256); }

extern "C" int ls2fm_sdf_eval(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                              const float* p, int64_t n, float* sdf, float* feat, float* normal, void* workspace,
                              void* stream) {
    LS2FM_CHECK_ARG(field_ok(field, grid, params) && n >= 0);
    if (n == 0) return LS2FM_OK;
    LS2FM_CHECK_ARG(p && sdf);
    if (!workspace) return LS2FM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    Packed* pk = (Packed*)workspace;
    int st = ls2fm_launch_prep_sdf(params, grid->n_levels, pk, s);
    if (st != LS2FM_OK) return st;
    const FieldC fc = make_field_c(field);
    const LevelSet lv = make_level_set(grid);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    const char* force = getenv("LS2FM_POINTS_KERNEL");           // tests: 1 = thread per point, 2 = 16 lanes per point, whatever n
    const int forced = force ? atoi(force) : 0;
    const bool wide_full = forced == 2 || (forced != 1 && n <= 32768);      // measured alone: 47 against 67 us at 32 768 points, 81 against 92 at 65 536
    ls2fm_prof_begin(LS2FM_PROF_SDF_EVAL, s);
    if (!normal && !feat && n <= 65536)       // sdf only, few points: latency-bound -> 16 lanes per point (bit-identical)
        sdf_eval_wide_kernel<<<(unsigned)((n + 15) / 16), 256, 0, s>>>(lv, fc, field->bg_sdf, field->bg_rad, pk, params->sdf_table,
                                                                      p, n, sdf);
    else if (wide_full && normal)             // sdf + features + normal of up to 32 768 points (the loops' point queries)
        sdf_eval_wide_full_kernel<true><<<(unsigned)((n + 15) / 16), 256, 0, s>>>(lv, fc, field->bg_sdf, field->bg_rad, pk,
                                                                                  params->sdf_table, p, n, sdf, feat, normal);
    else if (wide_full && feat)
        sdf_eval_wide_full_kernel<false><<<(unsigned)((n + 15) / 16), 256, 0, s>>>(lv, fc, field->bg_sdf, field->bg_rad, pk,
                                                                                   params->sdf_table, p, n, sdf, feat, nullptr);
    else if (normal)
        sdf_eval_kernel<true><<<blocks, 256, 0, s>>>(lv, fc, field->bg_sdf, field->bg_rad, pk, params->sdf_table, p, Lattice{},
                                                     n, sdf, feat, normal);
    else
        sdf_eval_kernel<false><<<blocks, 256, 0, s>>>(lv, fc, field->bg_sdf, field->bg_rad, pk, params->sdf_table, p, Lattice{},
                                                      n, sdf, feat, nullptr);
    ls2fm_prof_end(LS2FM_PROF_SDF_EVAL, s);
    return ls2fm_launch_status();
}

// sdf only, with the packed weights a preceding ls2fm_sdf_eval / ls2fm_sphere_trace left in `workspace` (same stream, same
// parameters): no weight prep -- a 14 us latency chain -- between a tracing call and the evaluation of its track
extern "C" int ls2fm_sdf_eval_prepared(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                                       const float* p, int64_t n, float* sdf, const void* workspace, void* stream) {
    LS2FM_CHECK_ARG(field_ok(field, grid, params) && n >= 0);
    if (n == 0) return LS2FM_OK;
    LS2FM_CHECK_ARG(p && sdf);
    if (!workspace) return LS2FM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const Packed* pk = (const Packed*)workspace;
    const FieldC fc = make_field_c(field);
    const LevelSet lv = make_level_set(grid);
    ls2fm_prof_begin(LS2FM_PROF_SDF_EVAL, s);
    if (n <= 65536)
        sdf_eval_wide_kernel<<<(unsigned)((n + 15) / 16), 256, 0, s>>>(lv, fc, field->bg_sdf, field->bg_rad, pk, params->sdf_table,
                                                                      p, n, sdf);
    else
        sdf_eval_kernel<false><<<(unsigned)((n + 255) / 256), 256, 0, s>>>(lv, fc, field->bg_sdf, field->bg_rad, pk, params->sdf_table, p,
                                                                          Lattice{}, n, sdf, nullptr, nullptr);
    ls2fm_prof_end(LS2FM_PROF_SDF_EVAL, s);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_sdf_volume(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                                int64_t n_side, int64_t first, int64_t count, int32_t ref_indexing, const double* step,
                                const double* origin, float* sdf, void* workspace, void* stream) {
    LS2FM_CHECK_ARG(field_ok(field, grid, params) && n_side >= 1 && n_side <= 2097151 && first >= 0 && count >= 0);
    LS2FM_CHECK_ARG(first + count <= n_side * n_side * n_side && step && origin);
    if (count == 0) return LS2FM_OK;
    LS2FM_CHECK_ARG(sdf);
    if (!workspace) return LS2FM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    Packed* pk = (Packed*)workspace;
    int st = ls2fm_launch_prep_sdf(params, grid->n_levels, pk, s);
    if (st != LS2FM_OK) return st;
    Lattice lat;
    lat.n_side = n_side; lat.first = first; lat.ref_indexing = ref_indexing ? 1 : 0;
    for (int a = 0; a < 3; ++a) { lat.step[a] = step[a]; lat.origin[a] = origin[a]; }
    const int64_t per_launch = (int64_t)1 << 30;           // grid.x stays below 2^31 / 256
    ls2fm_prof_begin(LS2FM_PROF_SDF_EVAL, s);
    for (int64_t at = 0; at < count; at += per_launch) {
        const int64_t n = count - at < per_launch ? count - at : per_launch;
        lat.first = first + at;
        sdf_eval_kernel<false><<<(unsigned)((n + 255) / 256), 256, 0, s>>>(make_level_set(grid), make_field_c(field), field->bg_sdf,
                                                                           field->bg_rad, pk, params->sdf_table, nullptr, lat, n,
                                                                           sdf + at, nullptr, nullptr);
    }
    ls2fm_prof_end(LS2FM_PROF_SDF_EVAL, s);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_sdf_prepare(const ls2fm_grid_desc* grid, const ls2fm_params* params, void* workspace, int32_t* zero_word,
                                 void* stream) {
    LS2FM_CHECK_ARG(grid_desc_ok(grid) && params && params->sdf_mlp[0].weight_v && params->sdf_mlp[1].weight_v);
    if (!workspace) return LS2FM_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    return ls2fm_launch_prep_sdf(params, grid->n_levels, (Packed*)workspace, s, zero_word);     // the word is zeroed by the same launch
}

static int sphere_trace_impl(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                             const float* ray0, const float* ray_dir, int64_t n_rays, float sdf_threshold, int32_t iters_max,
                             float* near, float* far, float* track, float* t_end, float* track_sdf, int32_t* trips, void* workspace,
                             void* stream, bool prepared);

extern "C" int ls2fm_sphere_trace(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                                  const float* ray0, const float* ray_dir, int64_t n_rays, float sdf_threshold,
                                  int32_t iters_max, float* near, float* far, float* track, float* t_end, float* track_sdf,
                                  int32_t* trips, void* workspace, void* stream) {
    return sphere_trace_impl(field, grid, params, ray0, ray_dir, n_rays, sdf_threshold, iters_max, near, far, track, t_end, track_sdf,
                             trips, workspace, stream, false);
}

extern "C" int ls2fm_sphere_trace_prepared(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                                           const float* ray0, const float* ray_dir, int64_t n_rays, float sdf_threshold,
                                           int32_t iters_max, float* near, float* far, float* track, float* t_end,
                                           float* track_sdf, int32_t* trips, void* workspace, void* stream) {
    return sphere_trace_impl(field, grid, params, ray0, ray_dir, n_rays, sdf_threshold, iters_max, near, far, track, t_end, track_sdf,
                             trips, workspace, stream, true);
}

static int sphere_trace_impl(const ls2fm_field_desc* field, const ls2fm_grid_desc* grid, const ls2fm_params* params,
                             const float* ray0, const float* ray_dir, int64_t n_rays, float sdf_threshold, int32_t iters_max,
                             float* near, float* far, float* track, float* t_end, float* track_sdf, int32_t* trips, void* workspace,
                             void* stream, bool prepared) {
    LS2FM_CHECK_ARG(field_ok(field, grid, params) && n_rays >= 0 && iters_max >= 0);
    LS2FM_CHECK_ARG(trips);
    hipStream_t s = (hipStream_t)stream;
    if (!prepared && hipMemsetAsync(trips, 0, sizeof(int32_t), s) != hipSuccess) return LS2FM_ERR_LAUNCH;
    if (n_rays == 0) return LS2FM_OK;
    LS2FM_CHECK_ARG(ray0 && ray_dir && near && far && track && t_end);
    if (!workspace) return LS2FM_ERR_WORKSPACE;
    Packed* pk = (Packed*)workspace;
    int st = prepared ? LS2FM_OK : ls2fm_launch_prep_sdf(params, grid->n_levels, pk, s);
    if (st != LS2FM_OK) return st;
    // latency-bound at stage-loop sizes (wide: 16 lanes per ray end; 8192 rays 0.26 ms per call against 0.52, 1024 rays 0.13
    // against 0.53), throughput-bound at tens of thousands of rays (one lane per ray end: no redundant lanes).  Both kernels
    // give bit-identical results.
    static const int force = [] { const char* e = getenv("LS2FM_TRACE_KERNEL"); return e ? atoi(e) : 0; }();      // 1 narrow, 2 wide, 3 wave
    const bool narrow = force == 1 || (force == 0 && n_rays > 20000);
    const bool wave = force == 3 || (force == 0 && n_rays <= 2048);
    ls2fm_prof_begin(LS2FM_PROF_SPHERE_TRACE, s);
    if (wave)           // one ray per 128-thread workgroup
        sphere_trace_wave_kernel<<<(unsigned)n_rays, 128, 0, s>>>(
            make_level_set(grid), make_field_c(field), field->bg_sdf, field->bg_rad, pk, params->sdf_table, ray0, ray_dir, n_rays,
            sdf_threshold, iters_max, near, far, track, t_end, track_sdf, trips);
    else if (narrow)
        sphere_trace_kernel<<<(unsigned)((2 * n_rays + 255) / 256), 256, 0, s>>>(
            make_level_set(grid), make_field_c(field), field->bg_sdf, field->bg_rad, pk, params->sdf_table, ray0, ray_dir, n_rays,
            sdf_threshold, iters_max, near, far, track, t_end, track_sdf, trips);
    else            // 8 rays per 256-thread workgroup
        sphere_trace_wide_kernel<<<(unsigned)((n_rays + 7) / 8), 256, 0, s>>>(
            make_level_set(grid), make_field_c(field), field->bg_sdf, field->bg_rad, pk, params->sdf_table, ray0, ray_dir, n_rays,
            sdf_threshold, iters_max, near, far, track, t_end, track_sdf, trips);
    ls2fm_prof_end(LS2FM_PROF_SPHERE_TRACE, s);
    return ls2fm_launch_status();
}
