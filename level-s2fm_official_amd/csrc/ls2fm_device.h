// Shared device-side helpers for the gfx950 kernels of libls2fm_hip.so.
// Built with -ffp-contract=off: every fused multiply-add in this library is an explicit fmaf(), so
// the sampling / normalisation arithmetic that feeds the hash indices rounds exactly like the
// reference's separate PyTorch ops (hash indices must be bit-exact).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ls2fm.h"

#define LS2FM_PRIME_Y 2654435761u
#define LS2FM_PRIME_Z 805459861u

#define LS2FM_CHECK_ARG(cond) \
    do { if (!(cond)) return LS2FM_ERR_INVALID_ARGUMENT; } while (0)

static inline int ls2fm_launch_status_at(const char* file, int line) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return LS2FM_OK;
    fprintf(stderr, "libls2fm_hip: %s (%s:%d)\n", hipGetErrorString(e), file, line);   // fail loudly
    return LS2FM_ERR_LAUNCH;
}
#define ls2fm_launch_status() ls2fm_launch_status_at(__FILE__, __LINE__)

// Per-level constants, passed by value in the kernel argument segment (scalar registers).
struct LevelSet {
    int32_t n_levels;
    float scale[LS2FM_MAX_LEVELS];
    uint32_t res[LS2FM_MAX_LEVELS];
    uint32_t size[LS2FM_MAX_LEVELS];
    uint32_t offset[LS2FM_MAX_LEVELS];
    uint32_t hashed[LS2FM_MAX_LEVELS];
};

static inline LevelSet make_level_set(const ls2fm_grid_desc* g) {
    LevelSet s;
    s.n_levels = g->n_levels;
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) {
        const bool live = l < g->n_levels;
        s.scale[l] = live ? g->scale[l] : 0.f;
        s.res[l] = live ? g->resolution[l] : 1u;
        s.size[l] = live ? g->size[l] : 1u;
        s.offset[l] = live ? g->offset[l] : 0u;
        s.hashed[l] = live ? g->hashed[l] : 0u;
    }
    return s;
}

static inline bool grid_desc_ok(const ls2fm_grid_desc* g) {
    if (!g || g->n_levels < 1 || g->n_levels > LS2FM_MAX_LEVELS || g->n_features != 2) return false;
    for (int l = 0; l < g->n_levels; ++l)
        if (g->size[l] == 0 || g->resolution[l] == 0) return false;
    return true;
}

// tcnn grid_index(): dense stride walk or coherent prime hash, then modulo the level size.
// `hashed` is precomputed on the host with the very stride walk of tcnn (ls2fm/hashgrid.py), so the
// two branches below are exactly its two outcomes.
__device__ __forceinline__ uint32_t corner_index(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res,
                                                 uint32_t size, uint32_t hashed) {
    uint32_t idx;
    if (hashed) {
        idx = cx ^ (cy * LS2FM_PRIME_Y) ^ (cz * LS2FM_PRIME_Z);
        // hashed levels are capped at 2^log2_hashmap_size: a power of two
        return ((size & (size - 1u)) == 0u) ? (idx & (size - 1u)) : (idx % size);
    }
    idx = cx + cy * res + cz * res * res;
    return idx % size;
}

// tcnn pos_fract() for linear interpolation: pos = fmaf(scale, x, 0.5); cell = (uint32)(int)floor(pos);
// w = pos - floor(pos).
__device__ __forceinline__ void pos_fract(float x, float scale, uint32_t& cell, float& w) {
    const float pos = fmaf(scale, x, 0.5f);
    const float fl = floorf(pos);
    cell = (uint32_t)(int32_t)fl;
    w = pos - fl;
}

// Cell of a point at one level: corner entry indices (absolute, in entries) and the 3 fractions.
struct Cell {
    uint32_t idx[8];
    float w[3];
};

__device__ __forceinline__ void locate(const float x[3], float scale, uint32_t res, uint32_t size,
                                       uint32_t offset, uint32_t hashed, Cell& c) {
    uint32_t g[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) pos_fract(x[d], scale, g[d], c.w[d]);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        c.idx[k] = offset + corner_index(g[0] + (k & 1), g[1] + ((k >> 1) & 1), g[2] + ((k >> 2) & 1), res, size,
                                         hashed);
}

// weight of corner k: product over axes of (bit ? w : 1-w), multiplied in axis order like tcnn
__device__ __forceinline__ float corner_weight(const float w[3], int k) {
    float wt = 1.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) wt *= ((k >> d) & 1) ? w[d] : 1.0f - w[d];
    return wt;
}

// d(corner weight)/d w[gd]  (without the level scale)
__device__ __forceinline__ float corner_dweight(const float w[3], int k, int gd) {
    float wt = ((k >> gd) & 1) ? 1.0f : -1.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d)
        if (d != gd) wt *= ((k >> d) & 1) ? w[d] : 1.0f - w[d];
    return wt;
}

// d2(corner weight)/d w[a] d w[b], a != b
__device__ __forceinline__ float corner_d2weight(const float w[3], int k, int a, int b) {
    const int o = 3 - a - b;
    float wt = (((k >> a) & 1) ? 1.0f : -1.0f) * (((k >> b) & 1) ? 1.0f : -1.0f);
    wt *= ((k >> o) & 1) ? w[o] : 1.0f - w[o];
    return wt;
}

// torch.nn.Softplus(beta=100, threshold=20) and its first two derivatives (models/base.py:203;
// derivative branches as in torch's softplus_backward: identity branch when 100 a > 20).
// Softplus(beta = 100, threshold = 20) (models/base.py:206) with its first two derivatives, on the hardware
// transcendental units (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp each) in the overflow-free form
//   e = exp(-|z|) in (0, 1],  softplus(z) = max(z, 0) + log1p(e),  sigmoid(z) = z >= 0 ? 1/(1+e) : e/(1+e),
//   sigmoid'(z) = e / (1+e)^2            (z = 100 a)
// ~14 VALU ops instead of ~70 for expf + log1pf + two IEEE divisions; absolute error of h <= 1e-9, relative error of the
// derivatives <= 3e-7 (the parity bar of the path is 1e-4 relative).
__device__ __forceinline__ float log1p_unit(float e) {          // log(1 + e) for e in [0, 1]
    // both forms computed, then selected: as a branch (what `c ? f() : g()` compiles to here) every softplus is a chain of
    // small basic blocks, and nothing -- no MFMA of the next block, no load -- is scheduled across them
    const float small = e * fmaf(e, fmaf(e, 0.333333333f, -0.5f), 1.0f);
    const float large = __builtin_amdgcn_logf(1.0f + e) * 0.693147181f;
    return e < 0.02f ? small : large;
}

__device__ __forceinline__ void softplus100(float a, float& h, float& d1, float& d2) {
    const float z = a * 100.0f;
    const float e = __builtin_amdgcn_exp2f(fabsf(z) * -1.44269504f);
    const float rd = __builtin_amdgcn_rcpf(1.0f + e);
    const bool lin = z > 20.0f;                                  // torch's threshold: identity above it
    const float soft = (fmaxf(z, 0.0f) + log1p_unit(e)) * 0.01f;  // a value, then a select: no branch (see log1p_unit)
    h = lin ? a : soft;
    const float er = e * rd;
    d1 = lin ? 1.0f : (z >= 0.0f ? rd : er);
    d2 = lin ? 0.0f : 100.0f * er * rd;
}

__device__ __forceinline__ float softplus100_value(float a) {
    const float z = a * 100.0f;
    const float e = __builtin_amdgcn_exp2f(fabsf(z) * -1.44269504f);
    const float soft = (fmaxf(z, 0.0f) + log1p_unit(e)) * 0.01f;
    return z > 20.0f ? a : soft;
}

// slab test of one ray against one box (ngp_pl semantics).  fminf/fmaxf ignore NaNs.
__device__ __forceinline__ void ray_box(const float o[3], const float d[3], const float c[3], const float h[3],
                                        float& t_near, float& t_far, bool& hit) {
    float t1 = -INFINITY, t2 = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float inv = 1.0f / d[a];
        const float lo = (c[a] - h[a] - o[a]) * inv;
        const float hi = (c[a] + h[a] - o[a]) * inv;
        const float mn = fminf(lo, hi), mx = fmaxf(lo, hi);
        t1 = a == 0 ? mn : fmaxf(t1, mn);
        t2 = a == 0 ? mx : fminf(t2, mx);
    }
    if (t1 > t2) { t1 = -1.0f; t2 = -1.0f; }
    hit = t2 > 0.0f;
    t_near = hit ? fmaxf(t1, 0.0f) : -1.0f;
    t_far = hit ? t2 : -1.0f;
}

// wave64 helpers -------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ float wave_scan_incl(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive suffix sum across the 64 lanes: sum of v over the lanes above this one
__device__ __forceinline__ float wave_suffix_excl(float v, int lane) {
    float s = __shfl_down(v, 1, 64);
    if (lane == 63) s = 0.f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_down(s, o, 64);
        if (lane + o < 64) s += t;
    }
    return s;
}
