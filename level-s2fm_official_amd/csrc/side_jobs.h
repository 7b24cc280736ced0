// Jobs of the backward's weight-gradient tail that can ride in ANOTHER kernel's launch (round 5).
//
// With the MLPs' weight gradients contracted inside shade_bwd, what is left of the side chain of rounds 2-4 is small: the decoder
// columns (dz x [p n f f2 1], 21 MB of operands), the level-1 sums of shade_bwd's per-ray partials (40 MB), the fixed-order reduction
// and the seven finalize tasks -- 13 + 30 us as two launches on a side stream.  Having a side stream at all costs the step 26 - 28 us
// (timing probe LS2FM_PROBE_NO_SIDE: 0.473 -> 0.445 ms replayed, 0.468 -> 0.443 eager): a cross-queue edge in front of scatter_fill,
// a fork / join pair of barrier packets, a longer replay-to-replay gap (two queues to drain) and contention with the fill.  Here the
// jobs are device functions over an LDS arena the host kernel lends, and scatter_fill runs them in LEADING rows of its grid
// (bin_scatter.hip): the backward is then ONE chain on ONE queue.  Hand-offs between the jobs inside that launch: write-through
// stores, drained, then a ticket; readers poll the ticket (one lane, bounded) and read with L2-bypassing loads -- the protocol of
// wgrad_tail_kernel (render_bwd.hip) and of the scan jobs (bin_items.h).
#pragma once

#include "wgrad_tail.h"

namespace {

typedef float sj_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ sj_f32x4 sj_mfma4(float a, float b, sj_f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------------------------------------------------------
// Radiance-decoder columns: dWc (3 x 39) = sum over samples of dz [p, n, f, f2, 1]^T and dWv (3 x 27) = sum over rays of
// (per-ray sums of dz) renc^T.  Every operand already lies in HBM as [row][sample]: the MFMA operands (row jl, 4
// consecutive samples) are plain 16-byte loads, no LDS.

// four consecutive samples s .. s + 3 of one row: the load is UNCONDITIONAL (the address clamped into the row's padded extent) and
// the masking is a separate step applied where the values are USED, one iteration later -- a load under a branch, or one whose
// lanes are patched right behind it, makes the compiler wait for it on the spot (this kernel was four exposed memory round
// trips per tile: 21 us for 21 MB)
__device__ __forceinline__ float4 load4_raw(const float* __restrict__ row, int64_t s, int64_t s_last) {
    return *reinterpret_cast<const float4*>(row + (s < s_last ? s : s_last));
}
__device__ __forceinline__ float4 mask4(const float4 v, int64_t s, int64_t n, bool on) {
    return make_float4((on && s < n) ? v.x : 0.f, (on && s + 1 < n) ? v.y : 0.f, (on && s + 2 < n) ? v.z : 0.f, (on && s + 3 < n) ? v.w : 0.f);
}

// Level 1 of the fixed-order sum of shade_bwd's per-ray weight-gradient partials (shade_bwd.hip, 3.): job (row k, segment sg)
// adds row k of the rays [sg * per, (sg + 1) * per) -- wave v takes the rays v, v + 4, ... of the segment, eight loads in
// flight, then the four waves' sums are added in order -- and leaves row k of segment sum sg.  The order of the additions is a
// function of (n_rays, kL1Seg) alone: deterministic.
// LS2FM_SIDE_PART_NT (round 6): the per-ray partials (written once by shade_bwd with non-temporal stores, read once here) as
// non-temporal loads
#ifndef LS2FM_SIDE_PART_NT
#define LS2FM_SIDE_PART_NT 0
#endif
struct L1Job { const float* slot_sdf; const float* slot_geo; float* l1_sdf; float* l1_geo; int n_slots, dual; };

template <bool WT>
__device__ __forceinline__ void wgrad_l1_job_a(const L1Job& jb, int job, float* __restrict__ arena) {
    float (*s_l1)[64] = reinterpret_cast<float (*)[64]>(arena);            // [kWmWaves][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rows = kRegsSdf + (jb.dual ? kRegsGeo : 0);
    int k = job % rows;
    const int sg = job / rows;
    const bool geo = k >= kRegsSdf;
    if (geo) k -= kRegsSdf;
    const int R = geo ? kRegsGeo : kRegsSdf;
    const float* __restrict__ src = (geo ? jb.slot_geo : jb.slot_sdf) + (int64_t)k * 64 + lane;
    const int per = (jb.n_slots + kL1Seg - 1) / kL1Seg;
    const int lo = sg * per, hi = min(lo + per, jb.n_slots);
    float acc = 0.f;
    int b = lo + wave;
    for (; b + 7 * kWmWaves < hi; b += 8 * kWmWaves) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = LS2FM_SIDE_PART_NT ? __builtin_nontemporal_load(&src[(int64_t)(b + u * kWmWaves) * (R * 64)]) : src[(int64_t)(b + u * kWmWaves) * (R * 64)];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; b < hi; b += kWmWaves) acc += LS2FM_SIDE_PART_NT ? __builtin_nontemporal_load(&src[(int64_t)b * (R * 64)]) : src[(int64_t)b * (R * 64)];
    s_l1[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        float* dst = (geo ? jb.l1_geo : jb.l1_sdf) + ((int64_t)sg * R + k) * 64;
        if (WT) {                       // sixteen lanes, 16 bytes each, write-through (same sums, same order)
            if (lane < 16) {
                float4 v;
                float* q = &v.x;
#pragma unroll
                for (int c = 0; c < 4; ++c) q[c] = (s_l1[0][4 * lane + c] + s_l1[1][4 * lane + c]) + (s_l1[2][4 * lane + c] + s_l1[3][4 * lane + c]);
                wg_store4(dst + 4 * lane, v);
            }
        } else dst[lane] = (s_l1[0][lane] + s_l1[1][lane]) + (s_l1[2][lane] + s_l1[3][lane]);
    }
}

// block `bid` of `dec_blocks`; BATCH tiles' loads in flight per trip (4: 100 + 24 registers; 2 for a host kernel with a tighter budget)
template <bool WT, int BATCH = 4>
__device__ __forceinline__ void wgrad_dec_block_a(const WsLayout& w, int dual, int64_t n_rays, const float* __restrict__ ws, float* __restrict__ part,
                                                  int dec_blocks, int bid, float* __restrict__ red) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jl = lane & 15, g = lane >> 4;
    const int64_t P = w.p_pad;
    sj_f32x4 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) acc[t] = sj_f32x4{0.f, 0.f, 0.f, 0.f};
    const int n_tiles_p = (int)((w.p + 15) / 16), n_tiles_r = (int)((n_rays + 15) / 16);
    // software-pipelined: the NEXT tile's four operand loads are in flight during this tile's twelve MFMAs (a wave has ~8
    // tiles: with the loads issued and awaited tile by tile the kernel was eight exposed memory round trips, 21 us for 21 MB)
    const int step = dec_blocks * kWmWaves;
    const int64_t s_last = P - 4;                  // rows are p_pad long (a multiple of 64)
    const float* __restrict__ row_a = ws + w.dz + (jl < 3 ? jl : 0) * P;
    const float* __restrict__ row_b0 = ws + w.fe + jl * P;
    const float* __restrict__ row_b1 = ws + (dual ? w.fe2 : w.fe) + jl * P;
    const float* __restrict__ row_b2 = jl < 3 ? ws + w.p3 + jl * P : ws + w.nrm + (jl < 6 ? jl - 3 : 0) * P;
    {
        // kDecBatch tiles per trip, all their loads issued before the first MFMA (a wave has ~8 tiles at the benchmark: two memory
        // round trips instead of thirty-two).  (A rotating "next tile" prefetch does not survive the compiler here: it merges the
        // loop-carried loads back into the iteration that uses them, or waits for vmcnt(0) at the loop head.)
        constexpr int kDecBatch = BATCH;
#pragma unroll 1
        for (int tile = bid * kWmWaves + wave; tile < n_tiles_p; tile += kDecBatch * step) {
            float4 ra[kDecBatch], rb0[kDecBatch], rb1[kDecBatch], rb2[kDecBatch];
#pragma unroll
            for (int u = 0; u < kDecBatch; ++u) {
                const int64_t s = (int64_t)(tile + u * step) * 16 + 4 * g;
                ra[u] = load4_raw(row_a, s, s_last); rb0[u] = load4_raw(row_b0, s, s_last);
                rb1[u] = load4_raw(row_b1, s, s_last); rb2[u] = load4_raw(row_b2, s, s_last);
            }
            __builtin_amdgcn_sched_barrier(0);       // (the machine scheduler otherwise sinks each tile's loads to its MFMAs)
#pragma unroll
            for (int u = 0; u < kDecBatch; ++u) {
                const int64_t s = (int64_t)(tile + u * step) * 16 + 4 * g;
                const bool on = tile + u * step < n_tiles_p;
                const float4 a = mask4(ra[u], s, w.p, on && jl < 3), b0 = mask4(rb0[u], s, w.p, on), b1 = mask4(rb1[u], s, w.p, on && dual != 0);
                float4 b2 = mask4(rb2[u], s, w.p, on && jl < 6);
                if (on && jl == 6) b2 = make_float4(1.f, 1.f, 1.f, 1.f);       // bias column (a is zero beyond the last sample)
                acc[0] = sj_mfma4(a.x, b0.x, acc[0]); acc[1] = sj_mfma4(a.x, b1.x, acc[1]); acc[2] = sj_mfma4(a.x, b2.x, acc[2]);
                acc[0] = sj_mfma4(a.y, b0.y, acc[0]); acc[1] = sj_mfma4(a.y, b1.y, acc[1]); acc[2] = sj_mfma4(a.y, b2.y, acc[2]);
                acc[0] = sj_mfma4(a.z, b0.z, acc[0]); acc[1] = sj_mfma4(a.z, b1.z, acc[1]); acc[2] = sj_mfma4(a.z, b2.z, acc[2]);
                acc[0] = sj_mfma4(a.w, b0.w, acc[0]); acc[1] = sj_mfma4(a.w, b1.w, acc[1]); acc[2] = sj_mfma4(a.w, b2.w, acc[2]);
            }
        }
    }
    {
        const int64_t r_last = w.r_pad - 4;
#pragma unroll 1
        for (int tile = bid * kWmWaves + wave; tile < n_tiles_r; tile += dec_blocks * kWmWaves) {
            const int64_t s = (int64_t)tile * 16 + 4 * g;
            const float4 ra = load4_raw(ws + w.dzr + (jl < 3 ? jl : 0) * w.r_pad, s, r_last), rb0 = load4_raw(ws + w.renc + jl * w.r_pad, s, r_last),
                         rb1 = load4_raw(ws + w.renc + (16 + jl < kView ? 16 + jl : 0) * w.r_pad, s, r_last);
            const float4 a = mask4(ra, s, n_rays, jl < 3), b0 = mask4(rb0, s, n_rays, true), b1 = mask4(rb1, s, n_rays, 16 + jl < kView);
            acc[3] = sj_mfma4(a.x, b0.x, acc[3]); acc[4] = sj_mfma4(a.x, b1.x, acc[4]); acc[3] = sj_mfma4(a.y, b0.y, acc[3]); acc[4] = sj_mfma4(a.y, b1.y, acc[4]);
            acc[3] = sj_mfma4(a.z, b0.z, acc[3]); acc[4] = sj_mfma4(a.z, b1.z, acc[4]); acc[3] = sj_mfma4(a.w, b0.w, acc[3]); acc[4] = sj_mfma4(a.w, b1.w, acc[4]);
        }
    }
    for (int wv = 0; wv < kWmWaves; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (wv == 0) red[(t * 4 + q) * 64 + lane] = acc[t][q];
                    else red[(t * 4 + q) * 64 + lane] += acc[t][q];
                }
        }
        __syncthreads();
    }
    float* dst = part + (int64_t)bid * (kRegsDec * 64);
    if (WT) {
        for (int q = tid; q < kRegsDec * 16; q += kWmThreads) wg_store4(dst + 4 * q, *reinterpret_cast<const float4*>(red + 4 * q));
    } else {
        for (int q = tid; q < kRegsDec * 64; q += kWmThreads) dst[q] = red[q];
    }
}


// ---- the jobs as leading workgroups of the scatter_fill launch
// order of the job ids: decoder blocks | level-1 sums | reduction rows | finalize tasks  (dispatched in that order, observed; the
// protocol does not depend on it: waiters poll, bounded)
struct SideJobs {
    int rows;                 // leading rows of the grid (rows * gridDim.x >= n_jobs); 0: no side jobs in this launch
    int dec_blocks, l1_jobs, n_red;
    WsLayout w;
    int dual;
    int64_t n_rays;
    float* ws;                // the render's workspace
    float* part_dec;
    L1Job l1;
    WgradParts parts;
    FinalizeArgs fa;
    int* flags;               // one word per job, zeroed by shade_bwd's zero job: [dec blocks | level-1 jobs | reduction rows]
    int probe;                // timing probe (wrong results): 1 no finalize tasks, 2 no reduction rows either, 4 no decoder tiles
    int* err;                 // host-visible sticky error word (ls2fm_async_error) or null
};

// decoder blocks of the in-fill form: a block's four waves walk (tiles / blocks / 4) tiles two at a time, one memory round trip per
// trip, and a decoder reduction row reads one partial per block with L2-bypassing loads, four in flight
#ifndef LS2FM_SIDE_DEC_BLOCKS
#define LS2FM_SIDE_DEC_BLOCKS 128
#endif
constexpr int kSideDecBlocks = LS2FM_SIDE_DEC_BLOCKS;

// Hand-offs are FLAGS, one word per producing job, not counters: 2 576 arrivals on one counter are 2 576 serialised memory-side
// atomics (~12 ns each) on ONE channel, behind which the fill's own stream on that channel queued -- scatter_fill 85 -> 113 us with
// nothing even waiting for the counter.  A producer drains its write-through stores and stores its flag (write-through); a consumer's
// first wave polls exactly the flags it depends on (L2-bypassing loads, bounded).
__device__ __forceinline__ void side_arrive(int* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's write-through stores have landed
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// flags base[i * stride], i < n, all set?  -> false when starved past the bound
__device__ __forceinline__ bool side_wait(const int* base, int n, int stride, int* s_flag) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        bool done = false;
        for (int spins = 0; spins < (1 << 20); ++spins) {
            bool mine = true;
            for (int i = lane; i < n; i += 64) mine = mine && __hip_atomic_load(base + (int64_t)i * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            done = __ballot(!mine) == 0ull;
            if (done) break;
            __builtin_amdgcn_s_sleep(16);
        }
        if (lane == 0) *s_flag = done ? 1 : 0;
    }
    __syncthreads();
    return *s_flag != 0;
}

// one 256-thread workgroup per job; `arena`: >= kTailArenaFloats floats of LDS, 16-byte aligned; s_flag: one int of LDS
__device__ __forceinline__ void side_job_run(const SideJobs& sj, int job, float* __restrict__ arena, int* s_flag) {
    int* const f_dec = sj.flags;
    int* const f_l1 = sj.flags + sj.dec_blocks;
    int* const f_red = f_l1 + sj.l1_jobs;
    if (job < sj.dec_blocks) {
        if (!(sj.probe & 4)) wgrad_dec_block_a<true, 2>(sj.w, sj.dual, sj.n_rays, sj.ws, sj.part_dec, sj.dec_blocks, job, arena);
        side_arrive(f_dec + job);
        return;
    }
    job -= sj.dec_blocks;
    if (job < sj.l1_jobs) {
        wgrad_l1_job_a<true>(sj.l1, job, arena);
        side_arrive(f_l1 + job);
        return;
    }
    job -= sj.l1_jobs;
    if (job < sj.n_red) {
        if (sj.probe & 2) return;
        // row `job` of the reduction: rows [0, rows_mlp) sum the kL1Seg segment sums of THEIR row (level-1 job ids sg * rows_mlp + row),
        // the decoder's rows every decoder block
        const int rows_mlp = kRegsSdf + (sj.dual ? kRegsGeo : 0);
        const bool ok = job < rows_mlp ? side_wait(f_l1 + job, kL1Seg, rows_mlp, s_flag) : side_wait(f_dec, sj.dec_blocks, 1, s_flag);
        reduce_partials_row_a<true>(sj.parts, sj.ws + sj.w.wg, job, arena);
        if (ok) side_arrive(f_red + job);        // (a starved producer: the flag stays down and the finalize tasks poison their outputs)
        return;
    }
    job -= sj.n_red;
    if (job < kFinalizeTasks) {
        if (sj.probe & 3) return;
        const bool ok = side_wait(f_red, sj.n_red, 1, s_flag);
        finalize_task_a(sj.fa, job, arena);
        // Bounded waits.  HIP promises no dispatch order inside a launch, so the protocol must not need one: the only workgroups of
        // this launch that ever wait are the n_red + kFinalizeTasks (~180) consumers; every other workgroup -- the producers and the
        // fill's own ~8 k -- runs to completion whatever is resident beside it.  Starvation therefore needs the device to offer no
        // more than ~180 workgroup slots to this launch (the launcher requires an order of magnitude more: render_bwd.hip), e.g. under
        // a CU mask or beside a foreign kernel that never ends.  Past the bound this task consumed incomplete sums: it poisons its
        // outputs (NaN gradient) AND sets the host-visible sticky error word, which the next ls2fm_render_bwd reports
        // (LS2FM_ERR_STARVED) -- loudly, not as one odd element.
        if (!ok && threadIdx.x == 0 && sj.err) __hip_atomic_store(sj.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!ok && threadIdx.x == 0 && (job < 2 || job >= 4 || sj.fa.dual)) {
            const float nan = __builtin_nanf("");
            float* b = job < 2 ? sj.fa.G.sdf_mlp[job].bias : (job < 4 ? sj.fa.G.geo_mlp[job - 2].bias : sj.fa.G.rad_mlp[job - 4].bias);
            b[0] = nan;
            if (job == 4) sj.fa.G.beta[0] = nan;
        }
    }
}

}  // namespace
