// Item classification of the table-gradient scatter, shared by the pass that COUNTS the items of every (tile, level, slab)
// -- folded into the forward's gather pass (render_fwd.hip), which computes the corner indices anyway -- and the pass that
// FILLS the sorted payload lists (bin_scatter.hip).  Both must classify identically: same functions, same arithmetic.
#pragma once

#include "render_common.h"

namespace {

constexpr int kAccSlots = 2 << kSlabShift;       // 16384 u64 accumulators = 128 KiB of LDS per workgroup
constexpr int kSlabBins = 128 << (13 - kSlabShift);                   // slabs per level (2^19 entries / 4096); bigger levels get wider slabs
constexpr int kBins = kSlabBins;
#ifndef LS2FM_FILL_TILE
#define LS2FM_FILL_TILE 256
#endif
constexpr int kCountThreads = LS2FM_FILL_TILE;
constexpr int kFillThreads = LS2FM_FILL_TILE;    // one sample point per thread
constexpr int kFillCap = kFillThreads * 17 / 4;       // items of a workgroup in the common case (4 per point + split pairs)
#ifndef LS2FM_FILL_WIN
#define LS2FM_FILL_WIN (LS2FM_FILL_TILE * 17 / 8)
#endif
constexpr int kFillWin = LS2FM_FILL_WIN;              // items staged in LDS per window pass of scatter_fill (two passes in the common case)
#ifndef LS2FM_ACC_THREADS
#define LS2FM_ACC_THREADS 1024
#endif
constexpr int kAccThreads = LS2FM_ACC_THREADS;
#ifndef LS2FM_ACC_BATCH
#define LS2FM_ACC_BATCH 4
#endif
constexpr int kAccBatch = LS2FM_ACC_BATCH;      // items per lane in flight in slab_accumulate (a hashed slab of the benchmark: 4096 items = one batch)
constexpr int kMaxParts = 16;

typedef unsigned long long u64;

// -DLS2FM_ITEM_PROBE24 (round 6, TIMING PROBE, wrong second-grid gradients): the dual item WITHOUT its second-grid words -- 24 bytes --
// with LS2FM_EXPLICIT_LEVELS=0: what the fill / accumulate pair would take if an item were a quarter smaller (profiles/r06_notes.md section 2)
#ifdef LS2FM_ITEM_PROBE24
struct __attribute__((aligned(8))) Item {
    uint32_t ij;
    float wx;
    float a0, a1, b0, b1;
};
#define LS2FM_ITEM_C0(it) ((it).b0)
#define LS2FM_ITEM_C1(it) ((it).b1)
#else
struct __attribute__((aligned(16))) Item {      // 32 bytes: BOTH grids of a dual-field pair, factored over the two x-corners
    uint32_t ij;           // local entry index of x-corner 0 (low 16 bits) and 1 (high 16 bits); 0xFFFF = not in this slab
    float wx;              // x weight: px(0) = 1 - wx, px(1) = wx
    float a0, a1, b0, b1;  // SDF grid, per feature: A = pyz de + qyz rr ; B = qd_x pyz rr
    float c0, c1;          // second grid: C = pyz de2
};
#define LS2FM_ITEM_C0(it) ((it).c0)
#define LS2FM_ITEM_C1(it) ((it).c1)
#endif

// Single field (the reference's default, options/LevelS2fM.yaml:9 `dual_field: false`, and every point query): the two
// x-corners' contributions are formed where the records are read and travel EXPLICITLY -- 20 bytes instead of the 24 a
// factored {ij, wx, A, B} would take (and of the 32 of the dual item with its second-grid half dead), one add per feature in
// the accumulate pass, and a whole run of merged samples (below) still fits one item.
struct __attribute__((aligned(4))) ItemS {      // 20 bytes
    uint32_t ij;           // as above
    float v00, v01;        // x-corner 0, features 0 / 1:  px0 A - B
    float v10, v11;        // x-corner 1:                  wx A + B
};
template <bool DUAL> struct ItemOf { typedef ItemS type; };
template <> struct ItemOf<true> { typedef Item type; };

// Round 5 -- EXPLICIT dual items on the coarse levels.  On a level whose cells are at least two sample spacings wide (ETH3D at 128
// samples: levels 0 .. 3, 7 .. 2 consecutive samples of a ray per cell) the dual field uses the single field's recipe: the x-corners'
// values of BOTH grids travel explicitly and consecutive samples in one cell are merged into one item per corner pair (run_sums).
// An explicit dual pair is nine words: the 32-byte Item re-read as {ij, v00 v01 v10 v11 (SDF grid), w00 w01 w10 (second grid)} plus
// w11 in a parallel float array indexed like the items (BinMeta::extra) -- one item size, one index arithmetic for every level.
// The explicit levels are a PREFIX of the levels (resolutions grow), decided on the host from the grid and the samples per ray:
constexpr int kMaxExplicitLevels = 6;
__host__ __device__ __forceinline__ Item explicit_item(uint32_t ij, const float* v) {        // v[0..6] -> the item, v[7] -> extra
    Item it;
    it.ij = ij; it.wx = v[0]; it.a0 = v[1]; it.a1 = v[2]; it.b0 = v[3]; it.b1 = v[4];
#ifndef LS2FM_ITEM_PROBE24
    it.c0 = v[5]; it.c1 = v[6];
#endif
    return it;
}

struct BinMeta {           // device arrays inside the workspace
    int* count;            // [L][kBins]  items per (level, slab)
    int* start;            // [L][kBins]  absolute offsets into items
    int* part_ticket;      // [L][kBins]  parts of a point-split slab that have left their partials (zeroed by scatter_fill): the
                           //             last one to arrive combines the slab (bin_scatter.hip)
    int* tile;             // [L][n_tiles][kBins]  per fill-workgroup counts, turned into absolute offsets by the scan: no
                           //                      global atomics anywhere (1 M of them cost ~50 us per pass), and the item
                           //                      order is deterministic
    Item* items;           // (single field: ItemS records in the same storage)
    float* extra;          // ninth word of the explicit dual items (levels < n_explicit): extra[absolute item index]; those levels
                           // come first in the item order, so the array covers kMaxExplicitLevels levels only
    u64* part_acc;         // kAccSlots u64 per (split slab, part): fixed-point partials, summed by slab_combine_kernel
    int part_blocks;       // capacity of part_acc in blocks of kAccSlots
    float* level_bound;    // [32] max over rays of the per-ray contribution bounds (rows 0..15 SDF grid, 16..31 second grid)
    int n_tiles;
};

struct LevelC {            // level constants
    uint32_t size, res, hashed, mask;
    bool pow2;
    float scale;
    int sshift;            // log2 of the slab size in entries on this level
};

__device__ __forceinline__ uint32_t wrap_index(uint32_t idx, uint32_t size) {
    if (idx >= size) {                   // in-range points: at most one wrap (size >= res^3)
        idx -= size;
        if (idx >= size) idx %= size;    // only for positions far outside the unit cube
    }
    return idx;
}

__device__ __forceinline__ uint32_t level_index(const LevelC& L, uint32_t cx, uint32_t cy, uint32_t cz) {
    if (L.hashed) {
        const uint32_t h = cx ^ (cy * LS2FM_PRIME_Y) ^ (cz * LS2FM_PRIME_Z);
        return L.pow2 ? (h & L.mask) : (h % L.size);
    }
    return wrap_index(cx + cy * L.res + cz * L.res * L.res, L.size);
}

// slabs of a level at the base slab size (8192 entries single grid / 4096 dual); a level with more than kSlabBins of them
// (log2_hashmap_size > 19 single / > 19 dual) is refused by the launcher
__host__ __device__ __forceinline__ int level_slabs(uint32_t size, int sshift) { return (int)((size + (1u << sshift) - 1u) >> sshift); }

__device__ __forceinline__ LevelC make_level_c(const LevelSet& lv, int l, int sshift) {
    LevelC L;
    L.size = lv.size[l]; L.res = lv.res[l]; L.hashed = lv.hashed[l]; L.scale = lv.scale[l];
    L.mask = L.size - 1u;
    L.pow2 = (L.size & L.mask) == 0u;
    L.sshift = sshift;
    return L;
}

// The items of one (point, level): for each (y,z) corner pair c = by + 2 bz the two x-corner entries idx0, idx1; one item
// when both lie in the same slab, else two half items.  f(slab, c, local idx0 or 0xFFFF, local idx1 or 0xFFFF).
// Used identically by the count and the fill pass.
template <typename F>
__device__ __forceinline__ void for_each_item(const LevelC& L, const uint32_t g[3], F&& f) {
    const uint32_t lmask = (1u << L.sshift) - 1u;
#pragma unroll
    for (unsigned c = 0; c < 4; ++c) {
        const uint32_t cy = g[1] + (c & 1u), cz = g[2] + (c >> 1);
        const uint32_t i0 = level_index(L, g[0], cy, cz), i1 = level_index(L, g[0] + 1u, cy, cz);
        const uint32_t s0 = i0 >> L.sshift, s1 = i1 >> L.sshift;
        if (s0 == s1) {
            f((int)s0, c, i0 & lmask, i1 & lmask);
        } else {
            f((int)s0, c, i0 & lmask, 0xFFFFu);
            f((int)s1, c, 0xFFFFu, i1 & lmask);
        }
    }
}


// ---- run merging.  Consecutive sample points that fall into the SAME cell of a level touch the same eight entries (coarse
// levels: 7 .. 1.4 consecutive samples of a ray per cell on ETH3D's levels 0 .. 4): their contributions are summed in registers
// by a segmented wave reduction in scatter_fill and travel as the items of the run's FIRST lane only -- fewer 20 / 32-byte
// round trips, fewer 64-bit LDS atomics, and far fewer same-address collisions in the coarse slabs.  Which lanes continue a
// run is a function of the cells alone, so the counting pass (render_fwd.hip) and the fill pass classify identically; a wave
// takes the merging path only when at least kMergeMin of its lanes continue a run (a wave-uniform decision both passes make
// from the same ballot), so the fine levels pay three shuffles and a ballot.
// Single field: a run is one explicit item per corner pair.  Dual field: the factored item cannot carry a sum over different
// x weights, so a merged run is two half items per pair (wx = 0 / 1, A = the summed corner values, B = 0).
#ifndef LS2FM_MERGE_MIN
#define LS2FM_MERGE_MIN 8
#endif
#ifndef LS2FM_MERGE_MIN_DUAL
#define LS2FM_MERGE_MIN_DUAL 65
#endif
// lanes of a wave that must continue a run for the wave to merge (65: never).  Measured at C2 (profiles/r04_ab_scatter*.txt):
// single field -- explicit items, a run is ONE item per pair -- merging takes scatter_fill 65.6 -> 53.2 us, slab_accumulate
// 62.5 -> 57.8 and the gather pass's counting 83 -> 80 (fewer colliding histogram increments), the step 0.372 -> 0.358 ms
// (ScanNet 4096 x 256: 2.48 -> 2.17 ms); dual field -- two half items per pair -- it costs scatter_fill what it saves
// slab_accumulate (91 -> 102, 123 -> 113 us; step 0.545 vs 0.551 ms): off.
constexpr int kMergeMinSingle = LS2FM_MERGE_MIN, kMergeMinDual = LS2FM_MERGE_MIN_DUAL;
struct RunFlags { unsigned long long cont; bool head, merged; };      // cont: lanes that continue their predecessor's run

__device__ __forceinline__ RunFlags wave_runs(const uint32_t g[3], bool live, int lane, int merge_min) {
    if (merge_min > 64) { RunFlags r; r.cont = 0ull; r.head = live; r.merged = false; return r; }
    const uint32_t p0 = __shfl_up(g[0], 1, 64), p1 = __shfl_up(g[1], 1, 64), p2 = __shfl_up(g[2], 1, 64);
    const int prev_live = __shfl_up((int)live, 1, 64);
    const bool same = lane > 0 && live && prev_live != 0 && p0 == g[0] && p1 == g[1] && p2 == g[2];
    unsigned long long m = __ballot(same);
    if ((int)__popcll(m) < merge_min) m = 0ull;
    RunFlags r;
    r.cont = m;
    r.head = live && ((m >> lane) & 1ull) == 0ull;
    r.merged = r.head && lane < 63 && ((m >> (lane + 1)) & 1ull) != 0ull;
    return r;
}

// v[lane] <- sum of v over the lanes of the run that STARTS at `lane` (meaningful on the run's first lane); fixed order
template <int NV>
__device__ __forceinline__ void run_sums(float (&v)[NV], unsigned long long cont, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned long long need = ((1ull << o) - 1ull) << 1;            // lanes lane+1 .. lane+o must all continue
        const bool ok = lane + o < 64 && ((cont >> lane) & need) == need;
        if (__ballot(ok) == 0ull) break;                                      // no run reaches that far: wave-uniform exit
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const float t = __shfl_down(v[q], o, 64);
            if (ok) v[q] += t;
        }
    }
}

// The items of one (point, level) with run merging: nothing for a lane that continues a run; for a run's first lane the
// classification of for_each_item, except that a merged run of a DUAL pair is always two half items.
// halves: the factored dual item (a merged run is two half items per pair); false: explicit items (single field, explicit dual levels)
template <typename F>
__device__ __forceinline__ void for_each_item_merged(const LevelC& L, const uint32_t g[3], const RunFlags& rf, const bool halves, F&& f) {
    if (!rf.head) return;
    const uint32_t lmask = (1u << L.sshift) - 1u;
#pragma unroll
    for (unsigned c = 0; c < 4; ++c) {
        const uint32_t cy = g[1] + (c & 1u), cz = g[2] + (c >> 1);
        const uint32_t i0 = level_index(L, g[0], cy, cz), i1 = level_index(L, g[0] + 1u, cy, cz);
        const uint32_t s0 = i0 >> L.sshift, s1 = i1 >> L.sshift;
        if (s0 == s1 && !(halves && rf.merged)) {
            f((int)s0, c, i0 & lmask, i1 & lmask);
        } else {
            f((int)s0, c, i0 & lmask, 0xFFFFu);
            f((int)s1, c, 0xFFFFu, i1 & lmask);
        }
    }
}

constexpr int kFillTile = kFillThreads;          // sample points per count / fill workgroup (must agree)
static_assert(kCountThreads == kFillThreads, "count and fill classify the same tiles");

inline int64_t meta_ints(int64_t n_points) {
    const int64_t n_tiles = (n_points + kFillTile - 1) / kFillTile;
    return (3 * LS2FM_MAX_LEVELS * kBins + 64 + LS2FM_MAX_LEVELS * n_tiles * kBins + 63) / 64 * 64;
}

// Scratch of the point-split slabs (dense / tiny levels whose few slabs receive every point's items: a slab's list is cut into
// `parts` workgroups): one block of kAccSlots 64-bit accumulators per (slab, part).  A level is only split while a part keeps
// > 16 k items, which bounds the blocks by ~n_points / 2048 per level; the plan (bin_scatter.hip) lowers `parts` to fit.
inline int part_blocks_capacity(int64_t n_points) {
    const int64_t want = n_points / 128 + 16;
    return (int)(want < 2048 ? want : 2048);
}
inline int64_t part_acc_floats(int64_t n_points) { return (int64_t)part_blocks_capacity(n_points) * kAccSlots * 2; }
inline int64_t extra_floats(int64_t n_points) { return (8 * (int64_t)kMaxExplicitLevels * n_points + 63) / 64 * 64; }

inline BinMeta make_bin_meta(float* bins_ws, int64_t n_points) {
    BinMeta bm;
    int* meta = reinterpret_cast<int*>(bins_ws);
    bm.n_tiles = (int)((n_points + kFillTile - 1) / kFillTile);
    bm.count = meta;
    bm.start = meta + LS2FM_MAX_LEVELS * kBins;
    bm.level_bound = reinterpret_cast<float*>(meta + 2 * LS2FM_MAX_LEVELS * kBins);
    bm.part_ticket = meta + 2 * LS2FM_MAX_LEVELS * kBins + 64;
    bm.tile = meta + 3 * LS2FM_MAX_LEVELS * kBins + 64;
    bm.part_acc = reinterpret_cast<u64*>(meta + meta_ints(n_points));   // 256-byte aligned
    bm.part_blocks = part_blocks_capacity(n_points);
    bm.extra = reinterpret_cast<float*>(meta + meta_ints(n_points) + part_acc_floats(n_points));
    bm.items = reinterpret_cast<Item*>(meta + meta_ints(n_points) + part_acc_floats(n_points) + extra_floats(n_points));
    return bm;
}

// ---- scans of the per-(tile, slab) counts, run by EXTRA WORKGROUPS of the shade_fwd launch (no launch, no stream fork: a
// cross-queue edge on the main chain costs ~10 us in a graph replay).  Job = (level, group of 16 slabs): per-slab totals
// and, inside a slab, the exclusive prefix of the tile counts; the last job to finish (ticket) turns the (level, slab)
// totals into absolute starts.  The fill adds the two.  The ticket word is zeroed by the gather pass's launch.
constexpr int kScanGroup = 16;
constexpr int kScanJobsPerLevel = kBins / kScanGroup;
struct ScanJob { BinMeta bm; int n_levels; };          // bm.tile == nullptr: no job

inline int* scan_ticket(const BinMeta& bm) { return reinterpret_cast<int*>(bm.level_bound) + 48; }
__device__ __forceinline__ int* scan_ticket_dev(const BinMeta& bm) { return reinterpret_cast<int*>(bm.level_bound) + 48; }
// unit counter of the persistent slab_accumulate launch (bin_scatter.hip): zeroed by the scatter_fill launch in front of it
__host__ __device__ __forceinline__ int* acc_claim(const BinMeta& bm) { return reinterpret_cast<int*>(bm.level_bound) + 50; }

__device__ __forceinline__ int wave_scan_incl_i32(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// nt = workgroup size: a multiple of 64, 64 <= nt <= 512.  The job's [n_tiles][16] block of counts goes through LDS in
// pieces of 512 tiles (the benchmark: one piece; coalesced, independent loads in flight together: a thread walking its column in global memory pays one
// dependent ~1 us access per tile, 20+ us per job -- and these jobs share a launch with shade_fwd).
constexpr int kScanTiles = 512;
constexpr int kScanArenaInts = kScanTiles * kScanGroup + (512 / kScanGroup) * kScanGroup + LS2FM_MAX_LEVELS + 1;
// `arena`: kScanArenaInts ints of the caller's LDS (the host kernel lends one of its own arrays: its LDS budget decides how
// many of its workgroups fit a CU)
[[maybe_unused]] __device__ void scan_job_run(const ScanJob& sj, const int job, const int tid, const int nt, int* arena) {
    int* stage = arena;
    int (*chunk_sum)[kScanGroup] = reinterpret_cast<int (*)[kScanGroup]>(arena + kScanTiles * kScanGroup);
    int* level_total = arena + kScanTiles * kScanGroup + (512 / kScanGroup) * kScanGroup;
    int& s_last = level_total[LS2FM_MAX_LEVELS];
    const BinMeta& bm = sj.bm;
    const int l = job / kScanJobsPerLevel;
    const int sub = tid % kScanGroup, ch = tid / kScanGroup, n_ch = nt / kScanGroup;
    const int b0 = (job % kScanJobsPerLevel) * kScanGroup;
    int* blk = bm.tile + (int64_t)l * bm.n_tiles * kBins + b0;       // element (t, s) at blk[t * kBins + s]
    int carry = 0;                                                  // column `sub`: items of the tiles before this piece
    for (int base = 0; base < bm.n_tiles; base += kScanTiles) {
        const int nt_here = bm.n_tiles - base < kScanTiles ? bm.n_tiles - base : kScanTiles;
        for (int e = tid; e < nt_here * kScanGroup; e += nt)
            stage[e] = blk[(int64_t)(base + e / kScanGroup) * kBins + e % kScanGroup];
        __syncthreads();
        const int per = (nt_here + n_ch - 1) / n_ch;
        const int t0 = ch * per, t1 = t0 + per < nt_here ? t0 + per : nt_here;
        int sum = 0;
        for (int t = t0; t < t1; ++t) sum += stage[t * kScanGroup + sub];
        chunk_sum[ch][sub] = sum;
        __syncthreads();
        int run = carry, total = carry;
        for (int q = 0; q < n_ch; ++q) {
            const int v = chunk_sum[q][sub];
            if (q < ch) run += v;
            total += v;
        }
        for (int t = t0; t < t1; ++t) {
            const int c = stage[t * kScanGroup + sub];
            stage[t * kScanGroup + sub] = run;
            run += c;
        }
        carry = total;
        __syncthreads();
        for (int e = tid; e < nt_here * kScanGroup; e += nt)
            blk[(int64_t)(base + e / kScanGroup) * kBins + e % kScanGroup] = stage[e];
        __syncthreads();
    }
    // ---- hand-off of the per-slab totals to the last job: write-through (sc1) stores, drained, then the ticket; the reader
    // uses sc1 loads.  (Two __threadfence() per job -- L2 write-back + invalidate, several us each with 4 jobs per CU --
    // made this launch 27 us; only these 16 words per job cross workgroups, everything else is read by a later kernel.)
    if (ch == 0) __hip_atomic_store(&bm.count[l * kBins + b0 + sub], carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(scan_ticket_dev(bm), 1) == sj.n_levels * kScanJobsPerLevel - 1;
    __syncthreads();
    if (!s_last) return;
    const int lane = tid & 63, wave = tid >> 6, n_waves = nt >> 6;
    constexpr int kChunks = (kBins + 63) / 64;
    constexpr int kHeld = 4;                     // levels a wave keeps in registers between the two passes (16 levels / >= 4 waves)
    int held[kHeld][kChunks];
#pragma unroll
    for (int h = 0; h < kHeld; ++h) {
        const int lv = wave + h * n_waves;
        int tot = 0;
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            const int bb = 64 * c + lane;
            held[h][c] = (lv < sj.n_levels && bb < kBins)
                             ? __hip_atomic_load(&bm.count[lv * kBins + bb], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            tot += held[h][c];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
        if (lane == 0 && lv < LS2FM_MAX_LEVELS) level_total[lv] = tot;
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < kHeld; ++h) {
        const int lv = wave + h * n_waves;
        if (lv >= sj.n_levels) continue;
        int before = 0;
        for (int q = 0; q < lv; ++q) before += level_total[q];
        int run2 = 0;
#pragma unroll
        for (int c = 0; c < kChunks; ++c) {
            const int bb = 64 * c + lane;
            const int cnt = held[h][c];
            const int incl = wave_scan_incl_i32(cnt, lane);
            if (bb < kBins) bm.start[lv * kBins + bb] = before + run2 + incl - cnt;
            run2 += __shfl(incl, 63, 64);
        }
    }
    // (more levels than kHeld * n_waves: not with 16 levels and the 4..8 waves of the kernels that run these jobs)
    if (tid == 0) *scan_ticket_dev(bm) = 0;
}

// ---- zero fills the backward needs, run by EXTRA WORKGROUPS of the shade_bwd launch: the weight-gradient accumulators and
// the point-split coarse levels of the gradient table(s)
struct ZeroJob { float4* a; int64_t na; float4* b; int64_t nb; float4* c; int64_t nc; int blocks; };
__device__ __forceinline__ void zero_job_run(const ZeroJob& z, const int job, const int tid, const int nt) {
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t stride = (int64_t)z.blocks * nt;
    for (int64_t i = (int64_t)job * nt + tid; i < z.na; i += stride) z.a[i] = zero;
    for (int64_t i = (int64_t)job * nt + tid; i < z.nb; i += stride) z.b[i] = zero;
    for (int64_t i = (int64_t)job * nt + tid; i < z.nc; i += stride) z.c[i] = zero;
}

}  // namespace
