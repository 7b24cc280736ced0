// The tail of the backward's weight-gradient chain, shared by wgrad_mlp.hip (which produces the partials) and render_bwd.hip
// (which launches the tail): the fixed-order sum of the per-workgroup partials and the `finalize` tasks (un-collapse the
// radiance decoder, weight-norm backward, d beta).
#pragma once

#include "render_common.h"

namespace {

constexpr int kWmThreads = 256;
constexpr int kWmWaves = kWmThreads / 64;
constexpr int kFinalizeTasks = 7;

// The reduced-gradient block crosses workgroups INSIDE one launch (wgrad_tail_kernel, render_bwd.hip): rows are written with
// write-through stores and read with L2-bypassing loads (agent-scope relaxed atomics), drained before the writer's ticket --
// the hand-off of the scan jobs (bin_items.h).  An agent-scope release / acquire pair instead writes back and invalidates a
// whole L2 per workgroup: measured 35.6 us for the launch and +12 us on the step (it runs beside slab_accumulate).
__device__ __forceinline__ void wg_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16 bytes write-through (a dword write-through store is one fabric write per LANE, ~6x the time per byte: fine for the 7 k
// floats of the reduced block, not for the 330 k floats of the in-launch partials of side_jobs.h)
__device__ __forceinline__ void wg_store4(float* p, const float4 v) {
    typedef float wg_f32x4 __attribute__((ext_vector_type(4)));
    const wg_f32x4 r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ float wg_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// several of them IN FLIGHT together: the compiler keeps atomic loads in program order and waits for each on the spot (a
// finalize task's nine loads per thread were nine exposed round trips to memory, ~13 us of the tail kernel's 35): the same
// L2-bypassing load instruction, issued back to back, one wait
__device__ __forceinline__ void wg_load3(const float* p0, const float* p1, const float* p2, float& v0, float& v1, float& v2) {
    asm volatile("global_load_dword %0, %3, off sc1\n\tglobal_load_dword %1, %4, off sc1\n\tglobal_load_dword %2, %5, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(p0), "v"(p1), "v"(p2) : "memory");
}

__device__ __forceinline__ void wg_load4(const float* p0, const float* p1, const float* p2, const float* p3, float& v0, float& v1, float& v2, float& v3) {
    asm volatile("global_load_dword %0, %4, off sc1\n\tglobal_load_dword %1, %5, off sc1\n\tglobal_load_dword %2, %6, off sc1\n\t"
                 "global_load_dword %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
}

// LDS of the tail's jobs, carved from an arena the calling kernel lends (floats): the standalone kernels declare one of their own
// (wrappers below); scatter_fill lends its staging area to the jobs that ride in its launch (side_jobs.h)
constexpr int kTailArenaFloats = 3 * 68 + 4 + 3 * 64 + 64 * 68 + 2 * 256;      // finalize_task: s_dwc, s_dbc, s_dt1, s_row, s_db (double[256])

// sum of the per-workgroup partials (fixed order -> deterministic), scattered into the reduced-gradient buffer
// (WgLayout).  One launch for all three producers: block ranges [0,85) SDF MLP, [85,153) second MLP, then 20 decoder.
typedef Ls2fmWgradParts WgradParts;

// one workgroup (256 threads) per output row `k` (block ranges above).  WT: the partials were written IN THIS LAUNCH by other
// workgroups (write-through stores, drained, then a ticket the caller has waited for): read with L2-bypassing loads, four in flight
template <bool WT>
__device__ __forceinline__ void reduce_partials_row_a(const WgradParts& wp, float* __restrict__ wg, int k, float* __restrict__ arena) {
    const float* __restrict__ part_sdf = wp.sdf;
    const float* __restrict__ part_geo = wp.geo;
    const float* __restrict__ part_dec = wp.dec;
    const int nb_dec = wp.nb_dec, dual = wp.dual;
    float (*s_sum)[64] = reinterpret_cast<float (*)[64]>(arena);          // [kWmWaves][64]
    int kind = 0;
    if (k >= kRegsSdf) { k -= kRegsSdf; kind = 1; if (!dual || k >= kRegsGeo) { k -= dual ? kRegsGeo : 0; kind = 2; } }
    const float* part = kind == 0 ? part_sdf : (kind == 1 ? part_geo : part_dec);
    const int R = kind == 0 ? kRegsSdf : (kind == 1 ? kRegsGeo : kRegsDec);
    const int nb = kind == 2 ? nb_dec : (kind == 1 ? wp.nb_geo : wp.nb_sdf);
    const int tid = threadIdx.x, lane = tid & 63, grp = tid >> 6, jl = lane & 15, g = lane >> 4;
    float s0 = 0.f, s1 = 0.f;
    int b = grp;
    if constexpr (WT) {
        // (same operand order as the plain form: s0 takes b, b + 2 W, ..., s1 takes b + W, b + 3 W, ...)
        for (; b < nb; b += 4 * kWmWaves) {
            const int b1 = b + kWmWaves, b2 = b + 2 * kWmWaves, b3 = b + 3 * kWmWaves;
            const float* q = part + (int64_t)k * 64 + lane;
            float v0, v1, v2, v3;
            wg_load4(q + (int64_t)b * R * 64, q + (int64_t)(b1 < nb ? b1 : b) * R * 64, q + (int64_t)(b2 < nb ? b2 : b) * R * 64,
                     q + (int64_t)(b3 < nb ? b3 : b) * R * 64, v0, v1, v2, v3);
            if (b1 < nb) { s0 += v0; s1 += v1; } else { s0 += v0; }
            if (b3 < nb) { s0 += v2; s1 += v3; } else if (b2 < nb) { s0 += v2; }
        }
    } else {
        for (; b + kWmWaves < nb; b += 2 * kWmWaves) {
            s0 += part[((int64_t)b * R + k) * 64 + lane];
            s1 += part[((int64_t)(b + kWmWaves) * R + k) * 64 + lane];
        }
        if (b < nb) s0 += part[((int64_t)b * R + k) * 64 + lane];
    }
    s_sum[grp][lane] = s0 + s1;
    __syncthreads();
    if (grp != 0) return;
    const float v = (s_sum[0][lane] + s_sum[1][lane]) + (s_sum[2][lane] + s_sum[3][lane]);
    if (kind == 2) {                              // decoder: rows = dz component 4g + q (g = 0, q < 3)
        const int t = k / 4, c3 = 4 * g + (k & 3);
        if (c3 >= 3) return;
        float* dWc = wg + WgLayout::dWc + c3 * 39;
        float* dWv = wg + WgLayout::dWv + c3 * 27;
        if (t == 0) wg_store(&dWc[6 + jl], v);
        else if (t == 1) wg_store(&dWc[22 + jl], v);
        else if (t == 2) { if (jl < 6) wg_store(&dWc[jl], v); else if (jl == 6) wg_store(&dWc[38], v); }
        else if (t == 3) wg_store(&dWv[jl], v);
        else if (16 + jl < kView) wg_store(&dWv[16 + jl], v);
        return;
    }
    const bool GEO = kind == 1;
    float* dW0 = wg + (GEO ? WgLayout::dG0 : WgLayout::dW0);
    float* dW1 = wg + (GEO ? WgLayout::dG1 : WgLayout::dW1);
    if (k < 48) {                                 // dW0'[16m + 4g + q][k' = 16mk + jl] -> the reference's column order
        const int m = k / 12, mk = (k / 4) % 3, q = k & 3;
        const int kp = 16 * mk + jl;
        if (kp < 36) wg_store(&dW0[(16 * m + 4 * g + q) * 36 + (kp < 32 ? 3 + kp : (kp < 35 ? kp - 32 : 35))], v);
    } else if (k < 64) {                          // dW1[1 + 4g + q][16m + jl]
        const int m = (k - 48) / 4, q = k & 3;
        wg_store(&dW1[(1 + 4 * g + q) * 65 + 16 * m + jl], v);
    } else if (!GEO && k < 80) {                  // dW1[0][16m + 4g + q] (both row-0 terms; already summed over jl)
        const int m = (k - 64) / 4, q = k & 3;
        if (jl == 0) wg_store(&dW1[16 * m + 4 * g + q], v);
    } else {                                      // db1
        const int t = k - (GEO ? 64 : 80);
        const int o = GEO ? 1 + 4 * t + g : 4 * t + g;
        if (jl == 0 && o < kOut) wg_store(&dW1[o * 65 + 64], v);
    }
}
[[maybe_unused]] __device__ void reduce_partials_row(const WgradParts& wp, float* __restrict__ wg, int k) {
    __shared__ float s_arena[kWmWaves * 64];
    reduce_partials_row_a<false>(wp, wg, k, s_arena);
}


// ------------------------------------------------------------------------------------------- finalize
// weight-norm backward of a whole layer:  W = (g/||v||) v  ->  dg = <dW,v>/||v|| ; dv = (g/||v||) dW - g <dW,v>/||v||^3 v.
// Called by all 256 threads; 16 lanes cooperate on a row (coalesced accesses, 16-wide shuffle reduction).
[[maybe_unused]] __device__ void weight_norm_bwd_rows(const float* v, const float* g, const float* dw, int dw_ld, int rows, int n_in,
                                     float* dv, float* dg, int tid, bool add = false) {
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
            for (int k = sub; k < n_in; k += 16) {
                const float val = s * dw[row * dw_ld + k] - c * v[row * n_in + k];
                dv[row * n_in + k] = add ? dv[row * n_in + k] + val : val;
            }
            if (sub == 0) dg[row] = add ? dg[row] + dot / nrm : dot / nrm;
        }
    }
}

struct FinalizeArgs {
    ls2fm_params P; ls2fm_param_grads G; int in_dim, in_dim2, rad_in, dual;
    const Packed* pk; const float* wg; const float* dbeta; int64_t n_rays;
    int add;        // tasks 0 / 1 (SDF MLP layers) ADD their results to G's tensors: a second producer of the same backward pass
};

// one workgroup (256 threads) per task
__device__ __forceinline__ void finalize_task_a(const FinalizeArgs& fa, const int task, float* __restrict__ arena) {
    const ls2fm_params& P = fa.P;
    const ls2fm_param_grads& G = fa.G;
    const int in_dim = fa.in_dim, in_dim2 = fa.in_dim2, rad_in = fa.rad_in, dual = fa.dual;
    const Packed* __restrict__ pk = fa.pk;
    const float* __restrict__ wg = fa.wg;
    const float* __restrict__ dbeta = fa.dbeta;
    const int64_t n_rays = fa.n_rays;
    float (*s_dwc)[68] = reinterpret_cast<float (*)[68]>(arena);                          // [3][68]
    float* s_dbc = arena + 3 * 68;                                                        // [4]
    float (*s_dt1)[64] = reinterpret_cast<float (*)[64]>(arena + 3 * 68 + 4);             // [3][64]
    float (*s_row)[68] = reinterpret_cast<float (*)[68]>(arena + 3 * 68 + 4 + 3 * 64);    // [64][68]: effective-weight gradients of the layer being processed
    double* s_db = reinterpret_cast<double*>(arena + 3 * 68 + 4 + 3 * 64 + 64 * 68);      // [256]  (offset 4752 floats: 8-byte aligned)
    static_assert((3 * 68 + 4 + 3 * 64 + 64 * 68) % 2 == 0, "s_db is 8-byte aligned inside a 16-byte aligned arena");
    const int tid = threadIdx.x;

    // one workgroup per layer (7 independent tasks; a single workgroup doing all of them was a 58 us latency chain):
    // 0/1 SDF MLP layers, 2/3 second field's layers, 4..6 radiance layers (each re-derives the small shared terms)
    if (task < 4) {
        const int which = task >> 1;
        if (which && !dual) return;
        const float* dW0 = wg + (which ? WgLayout::dG0 : WgLayout::dW0);
        const float* dW1 = wg + (which ? WgLayout::dG1 : WgLayout::dW1);
        const ls2fm_linear* lin = which ? P.geo_mlp : P.sdf_mlp;
        const ls2fm_linear_grad* gl = which ? G.geo_mlp : G.sdf_mlp;
        const int ind = which ? in_dim2 : in_dim;
        if ((task & 1) == 0) {
            static_assert(kHidden * 36 == 9 * 256, "three batches of three loads per thread");
#pragma unroll
            for (int q = 0; q < 9; q += 3) {
                const int i0 = tid + 256 * q, i1 = i0 + 256, i2 = i0 + 512;
                float v0, v1, v2;
                wg_load3(&dW0[i0], &dW0[i1], &dW0[i2], v0, v1, v2);
                s_row[i0 / 36][i0 % 36] = v0; s_row[i1 / 36][i1 % 36] = v1; s_row[i2 / 36][i2 % 36] = v2;
            }
            __syncthreads();
            const bool add = fa.add != 0 && which == 0;
            weight_norm_bwd_rows(lin[0].weight_v, lin[0].weight_g, &s_row[0][0], 68, kHidden, ind, gl[0].weight_v, gl[0].weight_g, tid, add);
            if (tid < kHidden) gl[0].bias[tid] = add ? gl[0].bias[tid] + s_row[tid][35] : s_row[tid][35];
            return;
        }
        {   // 17 x 64 values: four per thread (+ the last row's 64), the row-0 extra term with them -- two batches
            static_assert(kOut == 17 && kHidden == 64, "layout of the batches below");
            const int i0 = tid, i1 = tid + 256, i2 = tid + 512, i3 = tid + 768;
            const float* last = tid < 64 ? &dW1[16 * 65 + tid] : &dW1[0];
            const float* extra = tid < 64 ? &wg[WgLayout::dW1r0 + tid] : &dW1[0];
            float v0, v1, v2, v3, v4, v5;
            wg_load3(&dW1[(i0 / 64) * 65 + i0 % 64], &dW1[(i1 / 64) * 65 + i1 % 64], &dW1[(i2 / 64) * 65 + i2 % 64], v0, v1, v2);
            wg_load3(&dW1[(i3 / 64) * 65 + i3 % 64], last, extra, v3, v4, v5);
            s_row[i0 / 64][i0 % 64] = v0 + ((which == 0 && tid < 64) ? v5 : 0.f);        // (i0 / 64 == 0 <=> tid < 64)
            s_row[i1 / 64][i1 % 64] = v1; s_row[i2 / 64][i2 % 64] = v2; s_row[i3 / 64][i3 % 64] = v3;
            if (tid < 64) s_row[16][tid] = v4;
        }
        __syncthreads();
        const bool add = fa.add != 0 && which == 0;
        weight_norm_bwd_rows(lin[1].weight_v, lin[1].weight_g, &s_row[0][0], 68, kOut, kHidden, gl[1].weight_v,
                             gl[1].weight_g, tid, add);
        if (tid < kOut) gl[1].bias[tid] = add ? gl[1].bias[tid] + wg_load(&dW1[tid * 65 + 64]) : wg_load(&dW1[tid * 65 + 64]);
        return;
    }

    // ---- radiance decoder: expand dWc (3 x rad_in) and back through Wc = R2 R1 R0, bc = T1 b0 + R2 b1 + b2
    for (int idx = tid; idx < 3 * 68; idx += 256) {
        const int c = idx / 68, k = idx % 68;
        const float* row = wg + WgLayout::dWc + c * 39;
        float val = 0.f;
        if (k < 6) val = wg_load(&row[k]);
        else if (k < 33) val = wg_load(&wg[WgLayout::dWv + c * 27 + (k - 6)]);
        else if (k < 49) val = wg_load(&row[6 + (k - 33)]);
        else if (k < 65) val = dual ? wg_load(&row[22 + (k - 49)]) : 0.f;
        s_dwc[c][k] = val;
    }
    if (tid < 3) s_dbc[tid] = wg_load(&wg[WgLayout::dWc + tid * 39 + 38]);
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
[[maybe_unused]] __device__ void finalize_task(const FinalizeArgs& fa, const int task) {
    __shared__ __attribute__((aligned(16))) float s_arena[kTailArenaFloats];
    finalize_task_a(fa, task, s_arena);
}


}  // namespace
