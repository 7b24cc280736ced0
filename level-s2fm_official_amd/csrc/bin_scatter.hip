// Hash-table gradient scatter for gfx950: a counting sort of per-corner-pair payloads by table slab, then LDS-owned slabs
// that STREAM their payload lists and accumulate in exact 64-bit fixed point.
// No table-wide global atomics, no floating-point atomics, no gathers in the accumulate phase.
//
// Measurements on MI355X that shaped this design (profiles/r01_*, tools/atomic_bench.hip, tools/acc_stamps.py):
//   * global fp32 atomics: 17-21 G/s at every scope (memory side; the 8 XCD L2s are not coherent): a tcnn-style scatter
//     was 80 % of the step
//   * LDS ds_add_f32 ~0.26 lane/clk/CU, but LDS ds_add_u64 ~3160 G/s chip-wide (30x): accumulate in 64-bit fixed point
//   * a slab workgroup that walks a list of (point, pair) ids and GATHERS each point's records is bound by the request
//     rate of the L2 -> L1 path (one cache line per lane, ~205 G requests/s: ~3 requests per item, 131 + 114 us for the
//     two grids) -- so the records are turned into per-item payloads where they are still coalesced (by sample point)
//     and delivered to the slabs sorted
//
//   count          (in the forward's gather pass, render_fwd.hip: the corner cells are known there) per (512-point tile,
//                  level): which slab does each of the four (y,z) corner pairs fall into (the two x-corners of a pair
//                  share a slab except when cx+1 carries across the slab bit or a dense row ends -- then the pair is
//                  split into two half items; bin_items.h)
//   scan           (leading workgroups of the shade_fwd launch, bin_items.h) per-(tile, slab) counts -> absolute offsets;
//                  no global atomics in any pass (1 M of them per pass cost ~50 us), and the item order is deterministic
//   scatter_fill   same classification, now with shade_bwd's records (read once, coalesced): builds the 32-byte payload
//                  {local idx0 | idx1, wx, A0 A1 B0 B1 (SDF grid), C0 C1 (second grid)} of every item, sorts the workgroup's
//                  items by slab in LDS and writes each (workgroup, slab) run with full cache lines
//   slab_accumulate  workgroup = (level, slab[, part]) OWNS the slab of the table gradient(s) in LDS as 64-bit
//                  FIXED-POINT integers -- 8192 entries x 2 features of one grid, or (dual field: same geometry, same
//                  items) 4096 entries x (2 + 2) features of both -- and streams its payload list: value(bx) =
//                  px(bx) A + sgn(bx) B  (first-order trilinear weight and the double-backward derivative weight, A.4),
//                  second grid px(bx) C.  The quantum is a per-level power of two from a bound on one contribution
//                  (per-ray maxima from shade_bwd) with head-room for the worst-case hit count, so float->fixed and the
//                  integer sums are exact: the table gradient is the exactly rounded sum of its fp32 contributions,
//                  order-independent.  Every entry belongs to one slab: the table is written once with plain coalesced
//                  stores (no zero fill); only point-split coarse levels are flushed with a few float atomics.
#include <cstdlib>

#include "bin_items.h"

namespace {

// ------------------------------------------------------------------------------------------------ fill
// rpt[point] = {x y z gn0 | gn1 gn2 - -}; rec1[level][point] = {de0 de1 rr0 rr1}; rec2[level][point] = {de0 de1}
template <bool DUAL>
__global__ void __launch_bounds__(kFillThreads)
scatter_fill_kernel(LevelSet lv, FieldC fc, const float* __restrict__ center, const float* __restrict__ ray, int64_t n_points,
                    int64_t p_pad, int sshift, const float* __restrict__ rpt, const float* __restrict__ rec1,
                    const float* __restrict__ rec2, const float* __restrict__ ray_bound, int64_t n_rays, int64_t r_pad,
                    BinMeta bm, int level_base) {
    __shared__ int hist[kBins];          // items of this workgroup per slab, then running rank
    __shared__ int lds_off[kBins];       // first LDS slot of the slab's run
    __shared__ int base[kBins];          // first global item index of the slab's run
    __shared__ int run_len[kBins];       // items of this workgroup in the slab's run
    __shared__ Item s_items[kFillCap];
    __shared__ uint32_t s_gidx[kFillCap];
    __shared__ int s_total;
    const int tid = threadIdx.x, lane = tid & 63, l = level_base + (int)blockIdx.y;
    for (int b = tid; b < kBins; b += kFillThreads) hist[b] = 0;
    __syncthreads();
    const LevelC L = make_level_c(lv, l, sshift);
    const int64_t i = (int64_t)blockIdx.x * kFillThreads + tid;
    const bool live = i < n_points;
    uint32_t g[3] = {0u, 0u, 0u};
    float w[3] = {0.f, 0.f, 0.f}, d0 = 0.f, d1 = 0.f, r0 = 0.f, r1 = 0.f, e0 = 0.f, e1 = 0.f, qd[3] = {0.f, 0.f, 0.f};
    if (live) {
        const float4 a = reinterpret_cast<const float4*>(rpt)[2 * i];
        const float4 c = reinterpret_cast<const float4*>(rpt)[2 * i + 1];
        const float4 b = reinterpret_cast<const float4*>(rec1)[(int64_t)l * p_pad + i];
        const float x[3] = {a.x, a.y, a.z};                 // the grid-normalised position the forward classified (same bits)
#pragma unroll
        for (int q = 0; q < 3; ++q) pos_fract(x[q], L.scale, g[q], w[q]);
        d0 = b.x; d1 = b.y; r0 = b.z; r1 = b.w;
        qd[0] = L.scale * a.w; qd[1] = L.scale * c.x; qd[2] = L.scale * c.y;
        if (DUAL) {
            const float2 e = reinterpret_cast<const float2*>(rec2)[(int64_t)l * p_pad + i];
            e0 = e.x; e1 = e.y;
        }
        for_each_item(L, g, [&](int slab, unsigned, uint32_t, uint32_t) { atomicAdd(&hist[slab], 1); });
    }
    __syncthreads();
    // runs: LDS offsets (exclusive prefix over the slabs, wave 0) and one global reservation per slab
    if (tid < 64) {
        int run = 0;
#pragma unroll
        for (int c = 0; c < (kBins + 63) / 64; ++c) {
            const int b = 64 * c + lane;
            const int cnt = b < kBins ? hist[b] : 0;
            const int incl = wave_scan_incl_i32(cnt, lane);
            if (b < kBins) lds_off[b] = run + incl - cnt;
            run += __shfl(incl, 63, 64);
        }
        if (lane == 0) s_total = run;
    }
    __syncthreads();
    for (int b = tid; b < kBins; b += kFillThreads) {
        const int n = hist[b];
        base[b] = bm.start[l * kBins + b] + bm.tile[((int64_t)l * bm.n_tiles + blockIdx.x) * kBins + b];
        run_len[b] = n;
    }
    __syncthreads();
    for (int b = tid; b < kBins; b += kFillThreads) hist[b] = 0;
    __syncthreads();
    if (live) {
        for_each_item(L, g, [&](int slab, unsigned c, uint32_t i0, uint32_t i1) {
            const int by = (int)(c & 1u), bz = (int)(c >> 1);
            const float py = by ? w[1] : 1.0f - w[1], pz = bz ? w[2] : 1.0f - w[2];
            const float pyz = py * pz;
            const float qyz = (by ? qd[1] : -qd[1]) * pz + py * (bz ? qd[2] : -qd[2]);
            Item it;
            it.ij = i0 | (i1 << 16);
            it.wx = w[0];
            it.a0 = fmaf(qyz, r0, pyz * d0);
            it.a1 = fmaf(qyz, r1, pyz * d1);
            it.b0 = qd[0] * pyz * r0;
            it.b1 = qd[0] * pyz * r1;
            it.c0 = pyz * e0;
            it.c1 = pyz * e1;
            int rank = atomicAdd(&hist[slab], 1);
            // long runs (coarse levels: consecutive samples of a ray fall into the same cell) are stored permuted, so that
            // the 64 lanes of an accumulate wave, which read consecutive items, do not all hit the same entry (same-address
            // LDS atomics serialise): position = rank * K mod n with K prime > n
            const int n_run = run_len[slab];
            if (n_run > 64) rank = (int)(((long long)rank * 1000003ll) % n_run);
            const int slot = lds_off[slab] + rank;
            const uint32_t gi = (uint32_t)(base[slab] + rank);
            if (slot < kFillCap) { s_items[slot] = it; s_gidx[slot] = gi; }
            else bm.items[gi] = it;                         // more split pairs than the staging area holds: direct write
        });
    }
    __syncthreads();
    // runs of one slab are contiguous in LDS and in memory: consecutive threads write consecutive 32-byte items
    const int staged = s_total < kFillCap ? s_total : kFillCap;
    for (int q = tid; q < staged; q += kFillThreads) bm.items[s_gidx[q]] = s_items[q];
    // the level's first workgroup also reduces the per-ray bounds of a single contribution (written by shade_bwd) to the
    // level's bound: the accumulate workgroups read two floats instead of n_rays each
    if (blockIdx.x == 0) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(hist);
        for (int which = 0; which < (DUAL ? 2 : 1); ++which) {
            float b = 0.f;
            for (int64_t r = tid; r < n_rays; r += kFillThreads) b = fmaxf(b, ray_bound[(int64_t)(16 * which + l) * r_pad + r]);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
            if (lane == 0) red[tid >> 6] = b;
            __syncthreads();
            if (tid == 0) {
                float m = 0.f;
                for (int q = 0; q < kFillThreads / 64; ++q) m = fmaxf(m, red[q]);
                bm.level_bound[16 * which + l] = m;
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ accumulate
struct SlabPlan {
    int first[LS2FM_MAX_LEVELS + 1];     // first workgroup of every level
    int parts[LS2FM_MAX_LEVELS];         // item-range parts per slab of the level
    int headroom_bits;                   // log2 of the worst-case number of contributions to one entry
};

__device__ __forceinline__ void add_fixed(u64* slot, float v, float to_fixed) {
    // v * to_fixed is exact (power-of-two scale); |.| < 2^62 / worst-case hits by construction of the quantum
    atomicAdd(slot, (u64)__float2ll_rn(v * to_fixed));        // two's complement: integer sums are exact
}

// Two features in ONE 64-bit add: 32-bit two's complement halves, value = hi * 2^32 + lo.  The low half's borrows and carries
// run into the high half; decoding undoes them (lo = sign-extended low word, hi = (total - lo) >> 32), which is exact as long
// as each half's SUM fits 32 bits -- guaranteed by the quantum (packed_quantum_of).  Halves the LDS atomics of the sole-owner
// levels (12 of 16 at the benchmark); the price is the resolution of a contribution: 2^-(30 - hits bits) of the level's bound
// (>= 17 bits for the <= 8 k items of a slab) instead of exact -- still order-independent and deterministic.
__device__ __forceinline__ void add_fixed2(u64* slot, float v0, float v1, float to_fixed) {
    const int lo = __float2int_rn(v0 * to_fixed), hi = __float2int_rn(v1 * to_fixed);
    atomicAdd(slot, ((u64)(uint32_t)hi << 32) + (u64)(long long)lo);
}
__device__ __forceinline__ void unpack_fixed2(u64 total, float to_float, float& v0, float& v1) {
    const int lo = (int)(uint32_t)(total & 0xFFFFFFFFull);
    const int hi = (int)(((long long)total - (long long)lo) >> 32);
    v0 = (float)lo * to_float;
    v1 = (float)hi * to_float;
}
// hits: an upper bound of the contributions to one entry (the items this workgroup streams)
__device__ __forceinline__ void packed_quantum_of(float bound, int hits, float& to_fixed, float& to_float) {
    int e_bound = 0;
    if (bound > 0.f) (void)frexpf(bound, &e_bound);             // bound < 2^e_bound
    const int hb = 32 - __clz(hits > 1 ? hits : 1);            // hits < 2^hb
    int shift = 30 - hb - e_bound;                              // |sum| <= hits * bound < 2^(hb + e_bound) -> * 2^shift < 2^30
    shift = shift > 126 ? 126 : (shift < -126 ? -126 : shift);
    to_fixed = ldexpf(1.0f, shift);
    to_float = ldexpf(1.0f, -shift);
}

__device__ __forceinline__ void quantum_of(float bound, int headroom_bits, float& to_fixed, double& to_float) {
    // contributions are bounded by 2^e (e from the level's bound), sums by 2^(e + headroom): value * 2^shift fits in 62 bits
    int e_bound = 0;
    if (bound > 0.f) (void)frexpf(bound, &e_bound);             // bound < 2^e_bound
    int shift = 62 - headroom_bits - e_bound;
    shift = shift > 126 ? 126 : (shift < -126 ? -126 : shift);
    to_fixed = ldexpf(1.0f, shift);
    to_float = ldexp(1.0, -shift);
}

// -DLS2FM_STAMPS: per-workgroup phase time stamps (100 MHz), read back by tools/acc_stamps.py
#ifdef LS2FM_STAMPS
__device__ long long g_acc_stamps[8 * 4096];
#define ACC_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_acc_stamps[8 * blockIdx.x + (k)] = wall_clock64(); } while (0)
#else
#define ACC_STAMP(k) do {} while (0)
#endif

// -DLS2FM_PACKED_ACC=true: sole-owner slabs with short lists keep each feature PAIR in one 64-bit accumulator (add_fixed2).
// Measured (C2): slab_accumulate alone 117.5 -> 95.7 us, the step unchanged (0.547 = 0.547 ms, A/B on one box: the
// weight-gradient chain ends the backward, not the scatter); contributions below 2^-17 of the level's bound are lost (54 of
// 265 884 non-zero entries of the full-size golden become zero, which Adam does not forgive: it turns ANY non-zero gradient
// into a step of the learning rate), and lists of 16 k items or more (8192-ray batches) cannot use it.  Off.
#ifndef LS2FM_PACKED_ACC
#define LS2FM_PACKED_ACC false
#endif

template <bool DUAL, bool ADD_INTO>
__global__ void __launch_bounds__(kAccThreads)
slab_accumulate_kernel(LevelSet lv, SlabPlan plan, BinMeta bm, int sshift, float* __restrict__ dtable1,
                       float* __restrict__ dtable2, int block_base) {
    constexpr bool add_into = ADD_INTO;
    constexpr int F = DUAL ? 4 : 2;
    __shared__ u64 acc[kAccSlots];
    const int tid = threadIdx.x;
    ACC_STAMP(0);
    const int bid = block_base + (int)blockIdx.x;        // a launch may cover a range of levels only (ls2fm_render_opts level groups)
    int l = 0;
    while (bid >= plan.first[l + 1]) ++l;
    const int parts = plan.parts[l];
    const uint32_t wg = bid - plan.first[l];
    const uint32_t slab = wg / parts;
    const int part = (int)(wg % parts);
    const uint32_t size = lv.size[l];
    const uint32_t lo = slab << sshift;
    const uint32_t hi = lo + (1u << sshift) < size ? lo + (1u << sshift) : size;

    // this workgroup's share of the slab's payload list; the first trip's loads are issued before the LDS is zeroed
    const int n_items = bm.count[l * kBins + slab];
    const Item* __restrict__ list = bm.items + bm.start[l * kBins + slab];
    const int j_lo = (int)((int64_t)n_items * part / parts), j_hi = (int)((int64_t)n_items * (part + 1) / parts);
    if (add_into && j_hi <= j_lo) return;                // nothing to add to the tables' current values
    uint4 q0 = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u), q1 = make_uint4(0u, 0u, 0u, 0u);
    if (j_lo + tid < j_hi) {
        q0 = reinterpret_cast<const uint4*>(list + j_lo + tid)[0];
        q1 = reinterpret_cast<const uint4*>(list + j_lo + tid)[1];
    }
    // sole owner of its entries and a short list: the feature pairs share a 64-bit accumulator (add_fixed2)
    constexpr bool packing = LS2FM_PACKED_ACC;
    const bool packed = packing && parts == 1 && j_hi - j_lo < (1 << 14);
    const int n_slots = (packed ? F / 2 : F) * (int)(hi - lo);
    for (int e = tid; e < n_slots; e += kAccThreads) acc[e] = 0ull;
    // bound of a single contribution on this level (reduced over the rays by scatter_fill); second grid: rows 16..31
    float to_fixed1, to_fixed2;
    double to_float1, to_float2;
    float pk_float1 = 0.f, pk_float2 = 0.f;
    if (packed) {
        packed_quantum_of(bm.level_bound[l], j_hi - j_lo, to_fixed1, pk_float1);
        packed_quantum_of(DUAL ? bm.level_bound[16 + l] : 0.f, j_hi - j_lo, to_fixed2, pk_float2);
        to_float1 = to_float2 = 0.0;
    } else {
        quantum_of(bm.level_bound[l], plan.headroom_bits, to_fixed1, to_float1);
        quantum_of(DUAL ? bm.level_bound[16 + l] : 0.f, plan.headroom_bits, to_fixed2, to_float2);
    }
    __syncthreads();
    ACC_STAMP(1);

    // streamed, 32 bytes per lane, fully coalesced
    for (int j = j_lo + tid; j < j_hi; j += kAccThreads) {
        if (j != j_lo + tid) {
            q0 = reinterpret_cast<const uint4*>(list + j)[0];
            q1 = reinterpret_cast<const uint4*>(list + j)[1];
        }
        const uint32_t i0 = q0.x & 0xFFFFu, i1 = q0.x >> 16;
        const float wx = __uint_as_float(q0.y);
        const float a0 = __uint_as_float(q0.z), a1 = __uint_as_float(q0.w);
        const float b0 = __uint_as_float(q1.x), b1 = __uint_as_float(q1.y);
        const float c0 = __uint_as_float(q1.z), c1 = __uint_as_float(q1.w);
        const float px0 = 1.0f - wx;
        if (packed) {
            if (i0 != 0xFFFFu) {
                u64* slot = acc + (F / 2) * i0;
                add_fixed2(slot, fmaf(px0, a0, -b0), fmaf(px0, a1, -b1), to_fixed1);
                if (DUAL) add_fixed2(slot + 1, px0 * c0, px0 * c1, to_fixed2);
            }
            if (i1 != 0xFFFFu) {
                u64* slot = acc + (F / 2) * i1;
                add_fixed2(slot, fmaf(wx, a0, b0), fmaf(wx, a1, b1), to_fixed1);
                if (DUAL) add_fixed2(slot + 1, wx * c0, wx * c1, to_fixed2);
            }
            continue;
        }
        if (i0 != 0xFFFFu) {                             // x-corner 0: px = 1 - wx, derivative sign -
            u64* slot = acc + F * i0;
            add_fixed(slot + 0, fmaf(px0, a0, -b0), to_fixed1);
            add_fixed(slot + 1, fmaf(px0, a1, -b1), to_fixed1);
            if (DUAL) { add_fixed(slot + 2, px0 * c0, to_fixed2); add_fixed(slot + 3, px0 * c1, to_fixed2); }
        }
        if (i1 != 0xFFFFu) {                             // x-corner 1: px = wx, derivative sign +
            u64* slot = acc + F * i1;
            add_fixed(slot + 0, fmaf(wx, a0, b0), to_fixed1);
            add_fixed(slot + 1, fmaf(wx, a1, b1), to_fixed1);
            if (DUAL) { add_fixed(slot + 2, wx * c0, to_fixed2); add_fixed(slot + 3, wx * c1, to_fixed2); }
        }
    }
    __syncthreads();
    ACC_STAMP(2);
    // ---- flush: fixed point -> fp32 (one rounding of the exact sum); slot e = F * entry + feature
    float* dst1 = dtable1 + 2ull * (lv.offset[l] + lo);
    float* dst2 = DUAL ? dtable2 + 2ull * (lv.offset[l] + lo) : nullptr;
    if (packed) {                                        // slot e = (F / 2) * entry + grid: one float2 per slot (parts == 1)
        for (int e = tid; e < n_slots; e += kAccThreads) {
            const int entry = DUAL ? e >> 1 : e;
            const bool second = DUAL && (e & 1);
            float2 v;
            unpack_fixed2(acc[e], second ? pk_float2 : pk_float1, v.x, v.y);
            float2* dst = reinterpret_cast<float2*>((second ? dst2 : dst1) + 2 * entry);
            if (add_into) {
                if (acc[e] != 0ull) { float2 o = *dst; o.x += v.x; o.y += v.y; *dst = o; }
            } else *dst = v;
        }
        return;
    }
    for (int e = tid; e < n_slots; e += kAccThreads) {
        const int entry = e / F, f = e % F;
        const bool second = DUAL && f >= 2;
        float* dst = (second ? dst2 : dst1) + 2 * entry + (f & 1);
        const float v = (float)((double)(long long)acc[e] * (second ? to_float2 : to_float1));
        if (add_into) {                                           // a second producer of the same table: += (ordered behind the first)
            if (acc[e] != 0ull) { if (parts == 1) *dst += v; else atomicAdd(dst, v); }
        } else if (parts == 1) *dst = v;                          // sole owner of the entry
        else if (acc[e] != 0ull) atomicAdd(dst, v);               // point-split coarse level, zeroed by the host
    }
#ifdef LS2FM_STAMPS
    __syncthreads();
    ACC_STAMP(3);
    if (tid == 0 && blockIdx.x < 4096) { g_acc_stamps[8 * blockIdx.x + 4] = l; g_acc_stamps[8 * blockIdx.x + 5] = j_hi - j_lo; }
#endif
}

bool levels_fit(const ls2fm_grid_desc* grid, int sshift) {
    for (int l = 0; l < grid->n_levels; ++l)
        if (level_slabs(grid->size[l], sshift) > kSlabBins) return false;
    return true;
}

}  // namespace

// floats of workspace the scatter needs: meta + worst case 8 items (4 pairs, each split) of 32 bytes per (point, level)
int64_t ls2fm_bins_workspace_floats(int n_levels, int64_t n_points) {
    static_assert(sizeof(Item) == 32, "item layout");
    return meta_ints(n_points) + 8 * 8 * (int64_t)n_levels * n_points + 64;
}

size_t ls2fm_bin_counts_bytes() { return 0; }      // nothing to zero: every count is written, not accumulated

bool ls2fm_bins_levels_fit(const ls2fm_grid_desc* grid, int dual) { return levels_fit(grid, ls2fm_slab_shift(dual)); }

// payloads from shade_bwd's records, sorted by slab
int ls2fm_launch_scatter_fill(const ls2fm_grid_desc* grid, const FieldC& fc, const float* center, const float* ray, float* bins_ws,
                              int64_t n_points, int64_t p_pad, const float* rec1, const float* rec2, const float* rpt,
                              const float* ray_bound, int64_t n_rays, int dual, hipStream_t stream, int level_lo, int level_hi) {
    const int64_t r_pad = (n_rays + 63) / 64 * 64;
    const int sshift = ls2fm_slab_shift(dual);
    const BinMeta bm = make_bin_meta(bins_ws, n_points);
    const LevelSet lv = make_level_set(grid);
    if (level_hi < 0) level_hi = grid->n_levels;
    if (level_hi <= level_lo) return LS2FM_OK;
    const dim3 g((unsigned)bm.n_tiles, (unsigned)(level_hi - level_lo));
    if (dual) scatter_fill_kernel<true><<<g, kFillThreads, 0, stream>>>(lv, fc, center, ray, n_points, p_pad, sshift, rpt, rec1, rec2, ray_bound, n_rays, r_pad, bm, level_lo);
    else      scatter_fill_kernel<false><<<g, kFillThreads, 0, stream>>>(lv, fc, center, ray, n_points, p_pad, sshift, rpt, rec1, nullptr, ray_bound, n_rays, r_pad, bm, level_lo);
    return ls2fm_launch_status();
}

namespace {
struct HostPlan { SlabPlan plan; int total, zero_lo, zero_hi; };

HostPlan make_plan(const ls2fm_grid_desc* grid, int64_t n_points, int sshift) {
    HostPlan h{};
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) h.plan.parts[l] = 1;
    h.plan.headroom_bits = 4;                // 8 corners per point (+1)
    while ((1ll << (h.plan.headroom_bits - 4)) < n_points) ++h.plan.headroom_bits;
    h.zero_lo = h.zero_hi = -1;
    const int64_t target = 16384;            // items a workgroup should process (a hashed-level slab sees ~4P/slabs)
    for (int l = 0; l < LS2FM_MAX_LEVELS + 1; ++l) {
        h.plan.first[l] = h.total;
        if (l >= grid->n_levels) continue;
        const int slabs = level_slabs(grid->size[l], sshift);
        int parts = 1;
        if (!grid->hashed[l] || slabs < 16) {
            // dense / tiny level: 4 pair items per point spread over few slabs -> split the slabs' lists
            const int64_t per_block = 4 * n_points / slabs;
            parts = (int)((per_block + target - 1) / target);
            if (parts > kMaxParts) parts = kMaxParts;
            if (parts < 1) parts = 1;
        }
        h.plan.parts[l] = parts;
        if (parts > 1) {      // atomically flushed level: zeroed first (one range covering all such levels)
            if (h.zero_lo < 0) h.zero_lo = l;
            h.zero_hi = l;
        }
        h.total += slabs * parts;
    }
    return h;
}
}  // namespace

// the point-split coarse levels are flushed with float atomics: their range of the gradient table(s) is zeroed first (by
// leading workgroups of the shade_bwd launch)
void ls2fm_scatter_zero_range(const ls2fm_grid_desc* grid, int64_t n_points, bool dual, int64_t* first, int64_t* count) {
    const HostPlan h = make_plan(grid, n_points, ls2fm_slab_shift(dual ? 1 : 0));
    *first = 0; *count = 0;
    if (h.zero_lo < 0) return;               // levels in between that have a sole owner are overwritten afterwards anyway
    *first = grid->offset[h.zero_lo];
    *count = (int64_t)grid->offset[h.zero_hi] + grid->size[h.zero_hi] - *first;
}

#ifdef LS2FM_STAMPS
extern "C" int ls2fm_debug_acc_stamps(long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_acc_stamps), sizeof(long long) * 8 * 4096) == hipSuccess ? 0 : -1;
}
#endif

// dtable1 (and dtable2: dual field, both grids in one pass) are OVERWRITTEN over the whole grid; ls2fm_launch_scatter_zero must
// have run on them before.  add_into: the sums are ADDED to the tables' current values instead (entries without items untouched,
// nothing zeroed beforehand): a second gradient producer, ordered behind the first one by the caller.
int ls2fm_launch_slab_accumulate(const ls2fm_grid_desc* grid, float* bins_ws, int64_t n_points, float* dtable1, float* dtable2,
                                 hipStream_t stream, int level_lo, int level_hi, int add_into) {
    const bool dual = dtable2 != nullptr;
    const int sshift = ls2fm_slab_shift(dual ? 1 : 0);
    const BinMeta bm = make_bin_meta(bins_ws, n_points);
    const LevelSet lv = make_level_set(grid);
    const HostPlan h = make_plan(grid, n_points, sshift);
    if (level_hi < 0) level_hi = grid->n_levels;
    const int base = h.plan.first[level_lo], blocks = h.plan.first[level_hi] - base;
    if (blocks <= 0) return LS2FM_OK;
    if (dual)
        (add_into ? slab_accumulate_kernel<true, true> : slab_accumulate_kernel<true, false>)<<<blocks, kAccThreads, 0, stream>>>(
            lv, h.plan, bm, sshift, dtable1, dtable2, base);
    else
        (add_into ? slab_accumulate_kernel<false, true> : slab_accumulate_kernel<false, false>)<<<blocks, kAccThreads, 0, stream>>>(
            lv, h.plan, bm, sshift, dtable1, nullptr, base);
    return ls2fm_launch_status();
}
