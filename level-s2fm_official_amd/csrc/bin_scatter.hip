// Hash-table gradient scatter for gfx950: binning pre-pass + LDS-owned slabs with exact fixed-point accumulation.
// No table-wide global atomics, no floating-point atomics.
//
// Measurements on MI355X that shaped this design (profiles/r01_*, tools/atomic_bench.hip):
//   * global fp32 atomics: 17-21 G/s at every scope (memory side; the 8 XCD L2s are not coherent): a tcnn-style scatter
//     was 80 % of the step
//   * LDS ds_add_f32 ~0.26 lane/clk/CU, but LDS ds_add_u64 ~3160 G/s chip-wide (30x): accumulate in 64-bit fixed point
//   * letting every slab workgroup scan all sample points cost ~143 M wave instructions per launch (PMC): bin first
//
//   bin_count / bin_scan / bin_fill   (once per backward, shared by both grids -- they have the same geometry)
//       thread per 8 (point, level) keys; the 4-byte key (written by ray_encode) holds the slab ids of the point's four
//       (y,z) corner pairs on a hashed level, or its first corner index on a dense level.  Every (point, pair) becomes a
//       4-byte item in the list of the slab it falls into: LDS histogram + one global reservation per (workgroup, slab).
//   slab_accumulate  (per grid)
//       workgroup = (level, slab[, part]) OWNS 8192 entries x 2 features of the table gradient in LDS as 64-bit
//       FIXED-POINT integers and walks only its own item list; an item = a 16-byte per-level record + the point's L2-resident 32-byte record, then 4 unmasked
//       ds_add_u64 (2 x-corners x 2 features; first-order trilinear weight and, for the SDF grid, the double-backward
//       derivative weight in the same add).  The quantum is a per-level power of two from a bound on one contribution
//       (per-ray maxima from shade_bwd) with head-room for the worst-case hit count, so float->fixed and the integer
//       sums are exact: the table gradient is the exactly rounded sum of its fp32 contributions, order-independent and
//       bit-reproducible.  Every entry belongs to one slab: the table is written once with plain coalesced stores (no
//       zero fill); only point-split coarse levels are flushed with a few float atomics.
#include <cstdlib>

#include "render_common.h"

namespace {

constexpr int kSlabEntries = 1 << kSlabShift;    // 8192 entries = 128 KiB of int64 pairs
constexpr int kBins = 72;                        // per level: 64 slab bins, bin 64 = "check against every slab", padding
constexpr int kGenericBin = 64;
constexpr int kBinThreads = 256;
constexpr int kBinPerThread = 8;
constexpr int kBinTile = kBinThreads * kBinPerThread;     // 2048 points per workgroup
constexpr int kAccThreads = 1024;
constexpr int kMaxParts = 16;

typedef unsigned long long u64;

// items: (point << 3) | code ; code 0..3 = (y,z) corner pair whose two x-corners lie in the slab, 4 = check all corners
struct BinMeta {           // device arrays inside the workspace
    int* count;            // [L][kBins]
    int* start;            // [L][kBins]  absolute offsets into items
    int* cursor;           // [L][kBins]
    uint32_t* items;
};

struct LevelGeom { uint32_t size, res, hashed; };

__device__ __forceinline__ bool fast_level(uint32_t size, uint32_t hashed) {
    const uint32_t slabs = size >> kSlabShift;
    return hashed && (size & (size - 1u)) == 0u && slabs >= 1u && slabs <= 64u;
}

// classification of one key: up to 4 (bin, code) targets
template <typename F>
__device__ __forceinline__ void for_each_target(uint32_t k, uint32_t size, uint32_t res, uint32_t hashed, F&& f) {
    if (size <= (uint32_t)kSlabEntries) { f(0, 4u); return; }                  // single slab: every point, all corners
    if (k == 0xFFFFFFFFu || (hashed && !fast_level(size, hashed))) { f(kGenericBin, 4u); return; }
    if (hashed) {
#pragma unroll
        for (unsigned c = 0; c < 4; ++c) f((int)((k >> (6 * c)) & 63u), c);
        return;
    }
    const uint32_t span = 1u + res + res * res;                                  // dense: slabs its 8 corners can touch
    const uint32_t s0 = k >> kSlabShift, s1 = (k + span) >> kSlabShift;
    for (uint32_t s = s0; s <= s1 && s < 64u; ++s) f((int)s, 4u);
}

template <bool FILL>
__global__ void __launch_bounds__(kBinThreads)
bin_pass_kernel(LevelSet lv, const uint32_t* __restrict__ keys, int64_t n_points, int64_t p_pad, BinMeta bm) {
    __shared__ int hist[kBins];
    __shared__ int base[kBins];
    const int tid = threadIdx.x, l = blockIdx.y;
    const uint32_t size = lv.size[l], res = lv.res[l], hashed = lv.hashed[l];
    if (tid < kBins) hist[tid] = 0;
    __syncthreads();
    uint32_t key[kBinPerThread];
    const int64_t tile = (int64_t)blockIdx.x * kBinTile;
#pragma unroll
    for (int q = 0; q < kBinPerThread; ++q) {
        const int64_t i = tile + q * kBinThreads + tid;
        key[q] = i < n_points ? keys[(int64_t)l * p_pad + i] : 0u;
    }
#pragma unroll
    for (int q = 0; q < kBinPerThread; ++q) {
        const int64_t i = tile + q * kBinThreads + tid;
        if (i < n_points) for_each_target(key[q], size, res, hashed, [&](int bin, unsigned) { atomicAdd(&hist[bin], 1); });
    }
    __syncthreads();
    if (!FILL) {
        if (tid < kBins && hist[tid]) atomicAdd(&bm.count[l * kBins + tid], hist[tid]);
        return;
    }
    if (tid < kBins) {
        const int n = hist[tid];
        base[tid] = n ? bm.start[l * kBins + tid] + atomicAdd(&bm.cursor[l * kBins + tid], n) : 0;
        hist[tid] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kBinPerThread; ++q) {
        const int64_t i = tile + q * kBinThreads + tid;
        if (i < n_points)
            for_each_target(key[q], size, res, hashed, [&](int bin, unsigned code) {
                const int rank = atomicAdd(&hist[bin], 1);
                bm.items[base[bin] + rank] = ((uint32_t)i << 3) | code;
            });
    }
}

// exclusive prefix of the per-(level, bin) counts into absolute item offsets; zeroes the fill cursors.
// one wave per level (72 bins = lanes + 8 spill lanes), level totals combined through LDS
__device__ __forceinline__ int wave_scan_incl_int(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

__global__ void __launch_bounds__(64 * LS2FM_MAX_LEVELS)
bin_scan_kernel(int n_levels, BinMeta bm) {
    __shared__ int level_total[LS2FM_MAX_LEVELS];
    const int lane = threadIdx.x & 63, l = threadIdx.x >> 6;
    const bool on = l < n_levels;
    const int c0 = on ? bm.count[l * kBins + lane] : 0;
    const int c1 = (on && lane < kBins - 64) ? bm.count[l * kBins + 64 + lane] : 0;
    const int i0 = wave_scan_incl_int(c0, lane);
    const int t0 = __shfl(i0, 63, 64);
    const int i1 = wave_scan_incl_int(c1, lane);
    const int t1 = __shfl(i1, 63, 64);
    if (lane == 0) level_total[l] = t0 + t1;
    __syncthreads();
    if (!on) return;
    int before = 0;
    for (int q = 0; q < l; ++q) before += level_total[q];
    bm.start[l * kBins + lane] = before + i0 - c0;
    bm.cursor[l * kBins + lane] = 0;
    if (lane < kBins - 64) {
        bm.start[l * kBins + 64 + lane] = before + t0 + i1 - c1;
        bm.cursor[l * kBins + 64 + lane] = 0;
    }
}

struct SlabPlan {
    int first[LS2FM_MAX_LEVELS + 1];     // first work item of every level
    int parts[LS2FM_MAX_LEVELS];         // item-range parts per slab of the level
    int headroom_bits;                   // log2 of the worst-case number of contributions to one entry
};

__device__ __forceinline__ uint32_t wrap_index(uint32_t idx, uint32_t size) {
    if (idx >= size) {                   // in-range points: at most one wrap (size >= res^3)
        idx -= size;
        if (idx >= size) idx %= size;    // only for positions far outside the unit cube
    }
    return idx;
}

struct LevelC {                          // block-uniform level constants
    uint32_t size, res, hashed, mask, lo, hi;
    bool pow2;
    float scale;
    float to_fixed;                      // 1 / quantum (a power of two)
};

__device__ __forceinline__ uint32_t level_index(const LevelC& L, uint32_t cx, uint32_t cy, uint32_t cz) {
    if (L.hashed) {
        const uint32_t h = cx ^ (cy * LS2FM_PRIME_Y) ^ (cz * LS2FM_PRIME_Z);
        return L.pow2 ? (h & L.mask) : (h % L.size);
    }
    return wrap_index(cx + cy * L.res + cz * L.res * L.res, L.size);
}

struct Payload { float x[3], d0, d1, r0, r1, qd[3]; };

__device__ __forceinline__ void add_fixed(u64* slot, float v, float to_fixed) {
    // v * to_fixed is exact (power-of-two scale); |.| < 2^62 / worst-case hits by construction of the quantum
    atomicAdd(slot, (u64)__float2ll_rn(v * to_fixed));        // two's complement: integer sums are exact
}

// trilinear weight  W = px py pz ; directional derivative weight  D = qx py pz + px qy pz + px py qz
// with p_a(b) = b ? w_a : 1 - w_a and q_a(b) = (b ? +1 : -1) * scale * gn_a
template <bool SECOND_ORDER, bool CHECK>
__device__ __forceinline__ void add_pair(const LevelC& L, u64* acc, const Payload& pl, const uint32_t g[3], const float w[3],
                                         int by, int bz) {
    const float py = by ? w[1] : 1.0f - w[1], pz = bz ? w[2] : 1.0f - w[2];
    const float pyz = py * pz;
    float qyz = 0.f;
    if (SECOND_ORDER) qyz = (by ? pl.qd[1] : -pl.qd[1]) * pz + py * (bz ? pl.qd[2] : -pl.qd[2]);
#pragma unroll
    for (int bx = 0; bx < 2; ++bx) {
        const uint32_t idx = level_index(L, g[0] + bx, g[1] + by, g[2] + bz);
        if (!CHECK || (idx >= L.lo && idx < L.hi)) {
            const float px = bx ? w[0] : 1.0f - w[0];
            const float wt = px * pyz;
            float v0 = wt * pl.d0, v1 = wt * pl.d1;
            if (SECOND_ORDER) {
                const float dirw = fmaf(bx ? pl.qd[0] : -pl.qd[0], pyz, px * qyz);
                v0 = fmaf(dirw, pl.r0, v0);
                v1 = fmaf(dirw, pl.r1, v1);
            }
            add_fixed(&acc[2 * (idx - L.lo) + 0], v0, L.to_fixed);
            add_fixed(&acc[2 * (idx - L.lo) + 1], v1, L.to_fixed);
        }
    }
}

// payload of one (level, point): per-point record rpt[point] = {x y z gn0 | gn1 gn2 - -} and the per-level record
// rec[level][point] = {de0 de1 rr0 rr1} (SDF grid) / {de0 de1} (second grid)
template <bool SECOND_ORDER>
__device__ __forceinline__ Payload load_payload(const float* __restrict__ rec_l, const float* __restrict__ rpt, int64_t i,
                                                float scale) {
    Payload pl;
    const float4 a = reinterpret_cast<const float4*>(rpt)[2 * i];
    pl.x[0] = a.x; pl.x[1] = a.y; pl.x[2] = a.z;
    pl.qd[0] = pl.qd[1] = pl.qd[2] = 0.f;
    pl.r0 = pl.r1 = 0.f;
    if (SECOND_ORDER) {
        const float4 c = reinterpret_cast<const float4*>(rpt)[2 * i + 1];
        const float4 b = reinterpret_cast<const float4*>(rec_l)[i];
        pl.d0 = b.x; pl.d1 = b.y; pl.r0 = b.z; pl.r1 = b.w;
        pl.qd[0] = scale * a.w; pl.qd[1] = scale * c.x; pl.qd[2] = scale * c.y;
    } else {
        const float2 b = reinterpret_cast<const float2*>(rec_l)[i];
        pl.d0 = b.x; pl.d1 = b.y;
    }
    return pl;
}

template <bool SECOND_ORDER>
__device__ __forceinline__ void process_item(const LevelC& L, u64* acc, const Payload& pl, unsigned code) {
    uint32_t g[3];
    float w[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) pos_fract(pl.x[a], L.scale, g[a], w[a]);
    if (code < 4u) {
        add_pair<SECOND_ORDER, false>(L, acc, pl, g, w, (int)(code & 1u), (int)(code >> 1));
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) add_pair<SECOND_ORDER, true>(L, acc, pl, g, w, c & 1, c >> 1);
    }
}

template <bool SECOND_ORDER>
__global__ void __launch_bounds__(kAccThreads)
slab_accumulate_kernel(LevelSet lv, SlabPlan plan, BinMeta bm, int64_t p_pad, const float* __restrict__ rec,
                       const float* __restrict__ rpt, const float* __restrict__ ray_bound, int64_t n_rays, int64_t r_pad, float* __restrict__ dtable) {
    constexpr int REC = SECOND_ORDER ? 4 : 2;
    __shared__ u64 acc[2 * kSlabEntries];
    __shared__ float s_bound[kAccThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int l = 0;
    while ((int)blockIdx.x >= plan.first[l + 1]) ++l;
    const int parts = plan.parts[l];
    const uint32_t item = blockIdx.x - plan.first[l];
    const uint32_t slab = item / parts;
    const int part = (int)(item % parts);

    // bound of a single contribution on this level = max over rays (written per ray by shade_bwd)
    {
        float b = 0.f;
        for (int64_t r = tid; r < n_rays; r += kAccThreads) b = fmaxf(b, ray_bound[(int64_t)l * r_pad + r]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
        if (lane == 0) s_bound[wave] = b;
    }
    for (int e = tid; e < 2 * kSlabEntries; e += kAccThreads) acc[e] = 0ull;
    __syncthreads();
    float bound = 0.f;
#pragma unroll
    for (int q = 0; q < kAccThreads / 64; ++q) bound = fmaxf(bound, s_bound[q]);

    LevelC L;
    L.size = lv.size[l]; L.res = lv.res[l]; L.hashed = lv.hashed[l]; L.scale = lv.scale[l];
    L.mask = L.size - 1u;
    L.pow2 = (L.size & L.mask) == 0u;
    L.lo = slab << kSlabShift;
    L.hi = L.lo + kSlabEntries < L.size ? L.lo + kSlabEntries : L.size;
    // fixed-point quantum: contributions are bounded by 2^e (e from the level's bound), sums by 2^(e + headroom)
    int e_bound = 0;
    if (bound > 0.f) (void)frexpf(bound, &e_bound);             // bound < 2^e_bound
    int shift = 62 - plan.headroom_bits - e_bound;              // value * 2^shift fits in 62 bits after all hits
    shift = shift > 126 ? 126 : (shift < -126 ? -126 : shift);
    L.to_fixed = ldexpf(1.0f, shift);
    const double to_float = ldexp(1.0, -shift);
    const float* __restrict__ rec_l = rec + (int64_t)l * p_pad * REC;

    // this workgroup's share of its slab's item list, then of the level's "check every slab" list
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
        const int bin = which == 0 ? (int)slab : kGenericBin;
        const int n = bm.count[l * kBins + bin];
        const int64_t st = bm.start[l * kBins + bin];
        const int lo = (int)((int64_t)n * part / parts), hi = (int)((int64_t)n * (part + 1) / parts);
        // two items per thread per trip: the second record's gather is in flight while the first is accumulated
        for (int j = lo + tid; j < hi; j += 2 * kAccThreads) {
            const int j2 = j + kAccThreads;
            const uint32_t it0 = bm.items[st + j];
            const uint32_t it1 = j2 < hi ? bm.items[st + j2] : 0u;
            const Payload p0 = load_payload<SECOND_ORDER>(rec_l, rpt, it0 >> 3, L.scale);
            Payload p1 = p0;
            if (j2 < hi) p1 = load_payload<SECOND_ORDER>(rec_l, rpt, it1 >> 3, L.scale);
            process_item<SECOND_ORDER>(L, acc, p0, which == 0 ? (it0 & 7u) : 4u);
            if (j2 < hi) process_item<SECOND_ORDER>(L, acc, p1, which == 0 ? (it1 & 7u) : 4u);
        }
    }
    __syncthreads();
    // ---- flush: fixed point -> fp32 (one rounding of the exact sum)
    float* dst = dtable + 2ull * (lv.offset[l] + L.lo);
    const int n_out = 2 * (int)(L.hi - L.lo);
    if (parts == 1) {
        for (int e = tid; e < n_out; e += kAccThreads) dst[e] = (float)((double)(long long)acc[e] * to_float);   // sole owner
    } else {
        for (int e = tid; e < n_out; e += kAccThreads)                        // small coarse level, zeroed by the host
            if (acc[e] != 0ull) atomicAdd(dst + e, (float)((double)(long long)acc[e] * to_float));
    }
}

BinMeta make_bin_meta(float* bins_ws) {
    BinMeta bm;
    int* meta = reinterpret_cast<int*>(bins_ws);
    bm.count = meta;
    bm.start = meta + LS2FM_MAX_LEVELS * kBins;
    bm.cursor = meta + 2 * LS2FM_MAX_LEVELS * kBins;
    bm.items = reinterpret_cast<uint32_t*>(meta + 3 * LS2FM_MAX_LEVELS * kBins + 64);
    return bm;
}

}  // namespace

// floats of workspace the bins need: meta + worst case 4 items per (point, level) (+1 for the generic list)
int64_t ls2fm_bins_workspace_floats(int n_levels, int64_t n_points) {
    return 3 * LS2FM_MAX_LEVELS * kBins + 64 + 5 * (int64_t)n_levels * n_points + 64;
}

size_t ls2fm_bin_counts_bytes() { return sizeof(int) * LS2FM_MAX_LEVELS * kBins; }

// count -> scan -> fill of the per-slab item lists (geometry only: shared by both grids)
int ls2fm_launch_bin_build(const ls2fm_grid_desc* grid, const uint32_t* keys, int64_t n_points, int64_t p_pad, float* bins_ws,
                           hipStream_t stream) {
    const BinMeta bm = make_bin_meta(bins_ws);              // counts: zeroed by the caller (ls2fm_bin_counts_bytes)
    const LevelSet lv = make_level_set(grid);
    const dim3 g((unsigned)((n_points + kBinTile - 1) / kBinTile), (unsigned)grid->n_levels);
    bin_pass_kernel<false><<<g, kBinThreads, 0, stream>>>(lv, keys, n_points, p_pad, bm);
    bin_scan_kernel<<<1, 64 * LS2FM_MAX_LEVELS, 0, stream>>>(grid->n_levels, bm);
    bin_pass_kernel<true><<<g, kBinThreads, 0, stream>>>(lv, keys, n_points, p_pad, bm);
    return ls2fm_launch_status();
}

// dtable is OVERWRITTEN over the whole grid.  rec: per-(level, point) payload records (see load_payload);
// ray_bound: [level][r_pad] per-ray bounds of a single contribution.
int ls2fm_launch_slab_accumulate(const ls2fm_grid_desc* grid, float* bins_ws, int64_t n_points, int64_t p_pad, const float* rec,
                                 const float* rpt, bool second_order, const float* ray_bound, int64_t n_rays, float* dtable,
                                 hipStream_t stream) {
    const BinMeta bm = make_bin_meta(bins_ws);
    SlabPlan plan{};
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) plan.parts[l] = 1;
    plan.headroom_bits = 4;                  // 8 corners per point (+1)
    while ((1ll << (plan.headroom_bits - 4)) < n_points) ++plan.headroom_bits;
    const int64_t r_pad = (n_rays + 63) / 64 * 64;
    int total = 0, zero_lo = -1, zero_hi = -1;
    const int64_t target = 16384;            // items a workgroup should process (a hashed-level slab sees ~4P/64)
    for (int l = 0; l < LS2FM_MAX_LEVELS + 1; ++l) {
        plan.first[l] = total;
        if (l >= grid->n_levels) continue;
        const int slabs = (int)((grid->size[l] + kSlabEntries - 1) / kSlabEntries);
        int parts = 1;
        if (!grid->hashed[l] || slabs < 16) {
            // dense / tiny level: each slab's list holds ~ P / slabs points, each touching up to 8 corners
            const int64_t per_block = 2 * n_points / slabs;
            parts = (int)((per_block + target - 1) / target);
            if (parts > kMaxParts) parts = kMaxParts;
            if (parts < 1) parts = 1;
        }
        plan.parts[l] = parts;
        if (parts > 1) {      // atomically flushed level: zeroed first (one memset over the range of such levels)
            if (zero_lo < 0) zero_lo = l;
            zero_hi = l;
        }
        total += slabs * parts;
    }
    if (zero_lo >= 0) {       // levels in between that have a sole owner are overwritten afterwards anyway
        const size_t first = grid->offset[zero_lo], last = (size_t)grid->offset[zero_hi] + grid->size[zero_hi];
        if (hipMemsetAsync(dtable + 2ull * first, 0, sizeof(float) * 2ull * (last - first), stream) != hipSuccess)
            return LS2FM_ERR_LAUNCH;
    }
    const LevelSet lv = make_level_set(grid);
    if (second_order)
        slab_accumulate_kernel<true><<<total, kAccThreads, 0, stream>>>(lv, plan, bm, p_pad, rec, rpt, ray_bound, n_rays, r_pad, dtable);
    else
        slab_accumulate_kernel<false><<<total, kAccThreads, 0, stream>>>(lv, plan, bm, p_pad, rec, rpt, ray_bound, n_rays, r_pad, dtable);
    return ls2fm_launch_status();
}
