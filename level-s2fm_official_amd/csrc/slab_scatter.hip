// Hash-table gradient scatter WITHOUT table-wide global atomics and WITHOUT floating-point atomics (gfx950).
//
// Measurements on MI355X that shaped this kernel (profiles/r01_*, tools/atomic_bench.hip):
//   * global fp32 atomics: 17-21 G/s at every scope (they execute at the memory side; the 8 XCD L2s are not coherent)
//     -> the classic tcnn-style scatter was 80 % of the whole step
//   * LDS ds_add_f32: ~0.26 lane/clk/CU (~100 G pair-updates/s chip-wide), cost proportional to active lanes
//   * LDS ds_add_u64: ~3160 G/s, ds_add_u32: ~2250 G/s  (22-31x the float rate)
//   * a block-lockstep test/process loop pays a full ~2.5-3.4 us memory round trip per phase: __syncthreads()
//     drains every prefetch (vmcnt(0)), so the loop below has NO workgroup barrier: each wave is autonomous
// So each workgroup OWNS one slab (8192 entries x 2 features) of one level's gradient table in LDS as 64-bit
// FIXED-POINT accumulators.  The quantum is a power of two derived per level from a bound on a single contribution
// (per-ray maxima from shade_bwd) with head-room for the worst-case number of hits, so conversion float -> fixed
// and the integer sums are exact: the result is the exactly rounded sum of the fp32 contributions, independent of
// order (bit-reproducible), which no float-atomic scatter can offer.
//
// Work item = (level, slab, point-part).  Every table entry belongs to exactly one slab, so the table is overwritten
// in full (no zero-fill by the caller).  Levels whose slabs would see far more points than a hashed-level slab (the
// coarse dense levels) are additionally split over `parts` workgroups by point chunks; only those few small levels
// are flushed with global float atomics into a zeroed region.
//
// Each wave walks its own 512-point sub-chunks (keys of the next one prefetched, nothing drains them):
//   test    : one 4-byte key per (point, level), written by ray_encode: the slab ids of the four (y,z) corner pairs
//             (hashed levels) or the first corner index (dense levels).  Every MATCHING (point, pair) becomes an entry
//             in the wave's private LDS queue (positions from a wave prefix sum: no atomics).
//   process : a (point, pair) entry adds its 2 x-corners x 2 features with 4 unmasked ds_add_u64.  Entries are walked
//             in a decorrelated order (neighbouring samples of a ray share cells on coarse levels).  The payload is one
//             32-byte (SDF grid: first-order + double-backward terms) or 8-byte record per (point, level).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "render_common.h"

namespace {

constexpr int kSlabEntries = 1 << kSlabShift;    // 8192 entries = 128 KiB of int64 pairs
constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kPerLane = 8;                      // points per lane per sub-chunk
constexpr int kSub = 64 * kPerLane;              // 512 points per wave sub-chunk
constexpr int kChunk = kSub * kWaves;            // 8192 points per workgroup round
constexpr int kQueue = 768;                      // entries per wave; overflow (pathological inputs) is processed inline
constexpr int kMaxParts = 16;
constexpr unsigned kGeneric = 0x8000u;           // entry flag: evaluate all 8 corners with slab checks

typedef unsigned long long u64;

struct SlabPlan {
    int first[LS2FM_MAX_LEVELS + 1];     // first work item of every level
    int parts[LS2FM_MAX_LEVELS];         // point-parts per slab of the level
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

template <bool SECOND_ORDER>
struct Payload {
    float d0, d1, r0, r1, qd[3];
    __device__ __forceinline__ void load(const float* __restrict__ rec_l, int64_t i, float scale) {
        if (SECOND_ORDER) {
            const float4 ra = *reinterpret_cast<const float4*>(rec_l + i * 8);
            const float4 rb = *reinterpret_cast<const float4*>(rec_l + i * 8 + 4);
            d0 = ra.x; d1 = ra.y; r0 = ra.z; r1 = ra.w;
            qd[0] = scale * rb.x; qd[1] = scale * rb.y; qd[2] = scale * rb.z;
        } else {
            const float2 ra = *reinterpret_cast<const float2*>(rec_l + i * 2);
            d0 = ra.x; d1 = ra.y; r0 = r1 = 0.f; qd[0] = qd[1] = qd[2] = 0.f;
        }
    }
};

__device__ __forceinline__ void add_fixed(u64* slot, float v, float to_fixed) {
    // v * to_fixed is exact (power-of-two scale); |.| < 2^62 / worst-case hits by construction of the quantum
    atomicAdd(slot, (u64)__float2ll_rn(v * to_fixed));        // two's complement: integer sums are exact
}

// trilinear weight  W = px py pz ; directional derivative weight  D = qx py pz + px qy pz + px py qz
// with p_a(b) = b ? w_a : 1 - w_a and q_a(b) = (b ? +1 : -1) * scale * gn_a
template <bool SECOND_ORDER, bool CHECK>
__device__ __forceinline__ void add_pair(const LevelC& L, u64* acc, const Payload<SECOND_ORDER>& pl, const uint32_t g[3],
                                         const float w[3], int by, int bz) {
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

template <bool SECOND_ORDER>
__device__ __forceinline__ void process_entry(const LevelC& L, u64* acc, const float4* __restrict__ x4,
                                              const float* __restrict__ rec_l, int64_t base, unsigned entry) {
    const int64_t i = base + (entry & (kSub - 1));
    const float4 x = x4[i];
    Payload<SECOND_ORDER> pl;
    pl.load(rec_l, i, L.scale);
    uint32_t g[3];
    float w[3];
    pos_fract(x.x, L.scale, g[0], w[0]);
    pos_fract(x.y, L.scale, g[1], w[1]);
    pos_fract(x.z, L.scale, g[2], w[2]);
    if (entry & kGeneric) {
#pragma unroll
        for (int c = 0; c < 4; ++c) add_pair<SECOND_ORDER, true>(L, acc, pl, g, w, c & 1, c >> 1);
    } else {
        const int combo = (int)(entry >> 13) & 3;           // matching (y,z) corner pair: both x-corners are in the slab
        add_pair<SECOND_ORDER, false>(L, acc, pl, g, w, combo & 1, combo >> 1);
    }
}

template <bool SECOND_ORDER>
__global__ void __launch_bounds__(kThreads)
slab_scatter_kernel(LevelSet lv, SlabPlan plan, const float4* __restrict__ x4, const uint32_t* __restrict__ keys,
                    int64_t n_points, int64_t p_pad, const float* __restrict__ rec, const float* __restrict__ ray_bound,
                    int64_t n_rays, int64_t r_pad, float* __restrict__ dtable, int dbg, u64* __restrict__ dbg_out) {
    constexpr int REC = SECOND_ORDER ? 8 : 2;
    __shared__ u64 acc[2 * kSlabEntries];
    __shared__ unsigned short queue[kWaves][kQueue];
    __shared__ float s_bound[kWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int l = 0;
    while ((int)blockIdx.x >= plan.first[l + 1]) ++l;
    const int parts = plan.parts[l];
    const uint32_t item = blockIdx.x - plan.first[l];
    const uint32_t slab = item / parts;
    const int part = (int)(item % parts);
    const u64 t_begin = wall_clock64();

    // bound of a single contribution on this level = max over rays (written per ray by shade_bwd)
    {
        float b = 0.f;
        for (int64_t r = tid; r < n_rays; r += kThreads) b = fmaxf(b, ray_bound[(int64_t)l * r_pad + r]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
        if (lane == 0) s_bound[wave] = b;
    }
    for (int e = tid; e < 2 * kSlabEntries; e += kThreads) acc[e] = 0ull;
    __syncthreads();
    float bound = 0.f;
#pragma unroll
    for (int q = 0; q < kWaves; ++q) bound = fmaxf(bound, s_bound[q]);

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
    const bool fast_hash = L.hashed && L.pow2 && (L.size >> kSlabShift) >= 1u && (L.size >> kSlabShift) <= 64u;
    const bool single_slab = L.size <= (uint32_t)kSlabEntries;
    const uint32_t span = 1u + L.res + L.res * L.res;          // last corner - first corner on a dense level
    const float* __restrict__ rec_l = rec + (int64_t)l * p_pad * REC;
    const uint32_t* __restrict__ key_l = keys + (int64_t)l * p_pad;
    unsigned short* my_queue = queue[wave];

    // ---- wave-autonomous loop (no workgroup barrier inside)
    const int64_t stride = (int64_t)parts * kChunk;
    uint32_t key[kPerLane];
    auto load_keys = [&](int64_t base) {
#pragma unroll
        for (int q = 0; q < kPerLane; ++q) {
            const int64_t i = base + q * 64 + lane;
            key[q] = (!single_slab && i < n_points) ? key_l[i] : 0xFFFFFFFFu;
        }
    };
    int n_entries = 0;
    int64_t base = (int64_t)part * kChunk + (int64_t)wave * kSub;
    if (base < n_points && dbg != 3) load_keys(base);
    for (; base < n_points && dbg != 3; base += stride) {
        // test: per point a 5-bit code: bits 0..3 = matching (y,z) pairs, bit 4 = generic entry
        unsigned code[kPerLane];
        int mine = 0;
#pragma unroll
        for (int q = 0; q < kPerLane; ++q) {
            const int64_t i = base + q * 64 + lane;
            unsigned c = 0;
            if (i < n_points) {
                const uint32_t k = key[q];
                if (k == 0xFFFFFFFFu || (L.hashed && !fast_hash)) {
                    c = 16u;                                     // single slab / x-pair straddles slabs / odd level
                } else if (fast_hash) {
                    c = ((k & 63u) == slab ? 1u : 0u) | (((k >> 6) & 63u) == slab ? 2u : 0u) |
                        (((k >> 12) & 63u) == slab ? 4u : 0u) | (((k >> 18) & 63u) == slab ? 8u : 0u);
                } else {
                    c = (k < L.hi && k + span >= L.lo) ? 16u : 0u;   // dense: conservative range overlap
                }
            }
            code[q] = c;
            mine += __popc(c);
        }
        if (base + stride < n_points) load_keys(base + stride);     // prefetch; stays in flight through the process phase
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        const int total = __shfl(incl, 63, 64);
        if (total == 0) continue;                                    // wave-uniform
        int pos = incl - mine;
#pragma unroll
        for (int q = 0; q < kPerLane; ++q) {
            const unsigned local = (unsigned)(q * 64 + lane);
            unsigned c = code[q];
            while (c) {
                const int b = __ffs((int)c) - 1;
                c &= c - 1u;
                const unsigned entry = b == 4 ? (local | kGeneric) : (local | ((unsigned)b << 13));
                if (pos < kQueue) my_queue[pos] = (unsigned short)entry;
                else if (dbg != 1) process_entry<SECOND_ORDER>(L, acc, x4, rec_l, base, entry);   // queue overflow: inline
                ++pos;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // process: this wave's own entries, decorrelated order
        int nq = total < kQueue ? total : kQueue;
        n_entries += nq;
        if (dbg == 1) nq = 0;
        int nq_pad = 1;
        while (nq_pad < nq) nq_pad <<= 1;
        for (int e0 = lane; e0 < nq_pad; e0 += 64) {
            const int e = (int)(((unsigned)e0 * 37u) & (unsigned)(nq_pad - 1));
            if (e < nq) process_entry<SECOND_ORDER>(L, acc, x4, rec_l, base, my_queue[e]);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // ---- flush: fixed point -> fp32 (one rounding of the exact sum)
    float* dst = dtable + 2ull * (lv.offset[l] + L.lo);
    const int n_out = 2 * (int)(L.hi - L.lo);
    if (parts == 1) {
        for (int e = tid; e < n_out; e += kThreads) dst[e] = (float)((double)(long long)acc[e] * to_float);   // sole owner
    } else {
        for (int e = tid; e < n_out; e += kThreads)                           // small coarse level, zeroed by the host
            if (acc[e] != 0ull) atomicAdd(dst + e, (float)((double)(long long)acc[e] * to_float));
    }
    if (dbg_out && lane == 0) {
        u64* o = dbg_out + 8ull * (blockIdx.x * (u64)kWaves + wave);
        o[0] = l; o[1] = slab; o[2] = part; o[3] = t_begin; o[4] = wall_clock64(); o[5] = wave; o[6] = 0; o[7] = n_entries;
    }
}

}  // namespace

// dtable is OVERWRITTEN over the whole grid.  x4: float4 (x, y, z, -) per point; keys: uint32 [level][point];
// rec: [level][point][8 | 2]; ray_bound: [level][r_pad] per-ray bounds of a single contribution.
int ls2fm_launch_slab_scatter(const ls2fm_grid_desc* grid, const float* x4, const uint32_t* keys, int64_t n_points,
                              int64_t p_pad, const float* rec, bool second_order, const float* ray_bound, int64_t n_rays,
                              float* dtable, hipStream_t stream) {
    SlabPlan plan{};
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) plan.parts[l] = 1;
    plan.headroom_bits = 4;                  // 8 corners per point (+1)
    while ((1ll << (plan.headroom_bits - 4)) < n_points) ++plan.headroom_bits;
    const int64_t r_pad = (n_rays + 63) / 64 * 64;
    int total = 0;
    const int64_t target = 16384;            // entries a workgroup should process (what a hashed-level slab sees: ~P/16)
    for (int l = 0; l < LS2FM_MAX_LEVELS + 1; ++l) {
        plan.first[l] = total;
        if (l >= grid->n_levels) continue;
        const int slabs = (int)((grid->size[l] + kSlabEntries - 1) / kSlabEntries);
        int parts = 1;
        if (!grid->hashed[l] || slabs < 16) {
            // dense / tiny level: each of its slabs sees ~ P / slabs points (x4 corner pairs)
            const int64_t per_block = 4 * n_points / slabs;
            parts = (int)((per_block + target - 1) / target);
            const int max_parts = (int)((n_points + kChunk - 1) / kChunk);
            if (parts > kMaxParts) parts = kMaxParts;
            if (parts > max_parts) parts = max_parts;
            if (parts < 1) parts = 1;
        }
        plan.parts[l] = parts;
        if (parts > 1) {      // atomically flushed level: zero it first
            if (hipMemsetAsync(dtable + 2ull * grid->offset[l], 0, sizeof(float) * 2ull * grid->size[l], stream) != hipSuccess)
                return LS2FM_ERR_LAUNCH;
        }
        total += slabs * parts;
    }
    const LevelSet lv = make_level_set(grid);
    static const int dbg = getenv("LS2FM_SCATTER_DBG") ? atoi(getenv("LS2FM_SCATTER_DBG")) : 0;   // ablation / timeline
    u64* dbg_out = nullptr;
    static int dumps_left = 2;
    if (dbg == 5 && dumps_left > 0 && hipMalloc(&dbg_out, sizeof(u64) * 8 * total * kWaves) != hipSuccess) dbg_out = nullptr;
    if (second_order)
        slab_scatter_kernel<true><<<total, kThreads, 0, stream>>>(lv, plan, (const float4*)x4, keys, n_points, p_pad, rec,
                                                                  ray_bound, n_rays, r_pad, dtable, dbg, dbg_out);
    else
        slab_scatter_kernel<false><<<total, kThreads, 0, stream>>>(lv, plan, (const float4*)x4, keys, n_points, p_pad, rec,
                                                                   ray_bound, n_rays, r_pad, dtable, dbg, dbg_out);
    if (dbg_out) {           // debug: per-wave timeline (wall_clock64 ticks at 100 MHz)
        --dumps_left;
        const int n = total * kWaves;
        std::vector<u64> h(8 * n);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), dbg_out, sizeof(u64) * 8 * n, hipMemcpyDeviceToHost);
        (void)hipFree(dbg_out);
        u64 t0 = ~0ull;
        for (int b = 0; b < n; ++b) t0 = h[8 * b + 3] < t0 ? h[8 * b + 3] : t0;
        for (int b = 0; b < n; b += kWaves)
            fprintf(stderr, "SLABDBG order2=%d blk=%d level=%llu slab=%llu part=%llu start=%llu end=%llu entries=%llu\n",
                    (int)second_order, b / kWaves, h[8 * b], h[8 * b + 1], h[8 * b + 2], h[8 * b + 3] - t0, h[8 * b + 4] - t0,
                    h[8 * b + 7]);
    }
    return ls2fm_launch_status();
}
