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
#include <atomic>
#include <cstdlib>

#include "bin_items.h"
#include "side_jobs.h"

namespace {

// ------------------------------------------------------------------------------------------------ fill
// rpt[point] = {x y z gn0 | gn1 gn2 - -}; rec1[level][point] = {de0 de1 rr0 rr1}; rec2[level][point] = {de0 de1}
// Round 5: the sorted items leave through a WINDOW of the workgroup's sorted order (kFillWin slots of LDS, as many passes as
// the workgroup's item count needs: two at the benchmark) instead of one staging area for all of them: 22 KB instead of 41 KB
// of LDS -- 7 workgroups per CU instead of 3 for a kernel that is a chain of round trips and barriers, not a throughput --
// and no overflow path (any item count is served).  Every item's slot is fixed once (rank = LDS atomic) and kept in registers;
// a window pass re-derives the item's indices (eight hashes) and stages the items whose slot falls into the window.  The
// run offsets (two dependent global loads per slab) are requested at the very top, ahead of the records.
#ifndef LS2FM_FILL_NT
#define LS2FM_FILL_NT 0
#endif
// how the accumulate pass stores the gradient tables: 1 = non-temporal (default: 100 MB that nothing reads before the optimizer --
// kept out of the L2s they leave the table slices and item lists in place: slab_accumulate 84 -> 76 us and the NEXT step's gather
// pass 104 -> 98 us at C2), 0 = plain, 2 = write-through (sc1: no gain).  The fill's item stores stay plain: as nt / sc1 stores of
// 16-byte halves they cost 190 instead of 86 us.
#ifndef LS2FM_ACC_NT_STORE
#define LS2FM_ACC_NT_STORE 1
#endif
#ifndef LS2FM_FILL_PROBE
#define LS2FM_FILL_PROBE 0
#endif
// Round 6: how the staged items leave the LDS.  0 = one ITEM per lane (rounds 1-5): a 32-byte item is two 16-byte stores whose lanes
// are 32 bytes apart -- every store instruction touches 16 lines and fills half of each; a 20-byte item five dword stores 20 bytes
// apart (10 lines, a fifth of each).  1 = FLAT: the staging area is walked in store-sized chunks, consecutive lanes write consecutive
// 16-byte (dual) / 4-byte (single field) pieces of the sorted order, which is contiguous in memory inside a (workgroup, slab) run:
// the same number of store instructions, each covering whole lines except at run boundaries.
#ifndef LS2FM_FILL_FLAT
#define LS2FM_FILL_FLAT 1
#endif
// LS2FM_FILL_REC_NT (round 6): the scatter records (read exactly once, here) as non-temporal loads
#ifndef LS2FM_FILL_REC_NT
#define LS2FM_FILL_REC_NT 0
#endif
#ifndef LS2FM_FILL_MINW
#define LS2FM_FILL_MINW 5
#endif
// Dual field, the coarse levels (l < n_explicit; bin_items.h): EXPLICIT items -- the x-corners' values of both grids, consecutive
// samples in one cell merged -- formed pair by pair: eight live values and one segmented reduction at a time (all 32 at once cost
// 124 registers against 88, i.e. one workgroup per CU less -- as a second launch for those levels it took back what the merge
// saved: fill 86 -> 97 us).
template <bool DUAL>
__global__ void __launch_bounds__(kFillThreads, LS2FM_FILL_MINW)
scatter_fill_kernel(LevelSet lv, FieldC fc, const float* __restrict__ center, const float* __restrict__ ray, int64_t n_points,
                    int64_t p_pad, int sshift, const float* __restrict__ rpt, const float* __restrict__ rec1,
                    const float* __restrict__ rec2, const float* __restrict__ ray_bound, int64_t n_rays, int64_t r_pad,
                    BinMeta bm, int level_base, int reverse, int n_explicit, SideJobs sj) {
    typedef typename ItemOf<DUAL>::type ItemT;
    __shared__ int hist[kBins];          // items of this workgroup per slab, then running rank
    __shared__ int lds_off[kBins];       // first slot (in the workgroup's sorted order) of the slab's run
    __shared__ int base[kBins];          // first global item index of the slab's run
    __shared__ int run_len[kBins];       // items of this workgroup in the slab's run
    constexpr int kWin = DUAL ? kFillWin : kFillCap;      // (single field, 20-byte items: one pass -- two measured 52 -> 57 us)
    // staging area: items | their global indices | (dual) the explicit items' ninth word -- one raw block, which the side jobs of
    // this launch (side_jobs.h) use as their arena
    constexpr int kItemBytes = kWin * (int)sizeof(ItemT);
    constexpr int kStageBytes = kItemBytes + 4 * kWin + (DUAL ? 4 * kWin : 0);
    constexpr int kRawBytes = kStageBytes > 4 * kTailArenaFloats ? kStageBytes : 4 * kTailArenaFloats;
    static_assert(kItemBytes % 16 == 0, "the index array behind the items is aligned");
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[kRawBytes];
    ItemT* const s_items = reinterpret_cast<ItemT*>(s_raw);
    uint32_t* const s_gidx = reinterpret_cast<uint32_t*>(s_raw + kItemBytes);
    float* const s_extra = reinterpret_cast<float*>(s_raw + kItemBytes + 4 * kWin);           // ninth word of an explicit dual item
    __shared__ int s_total;
    if ((int)blockIdx.y < sj.rows) {     // leading rows: the weight-gradient tail's jobs (no side stream in the backward)
        const int job = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x;
        if (job < sj.dec_blocks + sj.l1_jobs + sj.n_red + kFinalizeTasks) side_job_run(sj, job, reinterpret_cast<float*>(s_raw), &s_total);
        return;
    }
    // issue-priority experiment (round 6, profiles/r06_notes.md section 9): LS2FM_FILL_PRIO = 1: raised until the hash pass is done (a young
    // workgroup's record loads go out first), 2: raised behind it (a workgroup that has its items staged leaves first).  Default 0.
#ifndef LS2FM_FILL_PRIO
#define LS2FM_FILL_PRIO 0
#endif
    if (LS2FM_FILL_PRIO == 1) __builtin_amdgcn_s_setprio(2);
    ItemT* __restrict__ g_items = reinterpret_cast<ItemT*>(bm.items);
    // levels are walked last to first: the accumulate launch reads the lists first to last, i.e. most recently written first
    const int y_lv = (int)blockIdx.y - sj.rows, n_lv = (int)gridDim.y - sj.rows;
    const int tid = threadIdx.x, lane = tid & 63, l = level_base + (reverse ? n_lv - 1 - y_lv : y_lv);
    // Round 6: which TILE a workgroup fills.  Workgroup x of a row runs on XCD x % 8 (observed; the rows are multiples of 8 wide
    // whenever this mapping is used), and the runs of neighbouring tiles are neighbours in a slab's list: a run is ~8 items = 256 B
    // at a 32-byte granularity, so most runs end inside a cache line whose other part the NEXT tile's workgroup writes.  With
    // tile = x the two halves of such a line are written through two different XCD L2s (two partial-line write-backs); with the
    // XCDs walking contiguous tile ranges -- tile = (x % 8) (n_tiles / 8) + x / 8 -- they meet in one L2 a dispatch round apart.
    // Any mapping gives the same items in the same places.  LS2FM_FILL_XCD_TILES=0: tile = x.
#ifndef LS2FM_FILL_XCD_TILES
#define LS2FM_FILL_XCD_TILES 1
#endif
    const int n_tiles_x = (int)gridDim.x;
    const int tile_x = (LS2FM_FILL_XCD_TILES && (n_tiles_x & 7) == 0) ? ((int)blockIdx.x & 7) * (n_tiles_x >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    if (tile_x == 0 && y_lv == 0) {              // armed for the accumulate launch behind this one: unit counter, slab tickets
        if (tid == 0) *acc_claim(bm) = 0;
        for (int q = tid; q < LS2FM_MAX_LEVELS * kBins; q += kFillThreads) bm.part_ticket[q] = 0;
    }
    static_assert(kBins <= kFillThreads, "one thread per slab for the run offsets");
    // Round 6: the workgroup's item count per slab is NOT counted again here (rounds 1-5: a pass over the pairs -- eight hashes and
    // 4-8 LDS atomics per thread -- and two barriers): the gather pass counted exactly these items (same classification,
    // bin_items.h) and the scans left, per (tile, slab), the number of items of the EARLIER tiles: the difference of two
    // neighbouring rows is this tile's count (the last tile: against the slab's total).  Timing probes of this round (no item
    // stores at all: 70 of 90 us) showed the kernel bound by its own instruction count, not by its stores.
    // LS2FM_FILL_COUNT_{DUAL,SINGLE}: 1 (default) = counted here, in the ONE hash pass, with LDS atomics; 0 = taken from the scan
    // results as described above (no atomics, one barrier fewer -- but the offsets' loads then have a consumer at the head of the
    // workgroup's chain instead of at its staging).  Measured at C2 (profiles/r06_notes.md section 3, profiles/r06_raw/c16_*), rounds 1-5's separate counting
    // pass | from the scan rows | counted in the hash pass:  dual field 87.0 | 84.3 | 81.2 us;  single field 64.0 | 66.8 | 63.7 us.
#ifndef LS2FM_FILL_COUNT_DUAL
#define LS2FM_FILL_COUNT_DUAL 1
#endif
#ifndef LS2FM_FILL_COUNT_SINGLE
#define LS2FM_FILL_COUNT_SINGLE 1
#endif
    constexpr bool kCountHere = DUAL ? (LS2FM_FILL_COUNT_DUAL != 0) : (LS2FM_FILL_COUNT_SINGLE != 0);
    int pre_base = 0, n_mine = 0;
    if (tid < kBins) {
        const int* row = bm.tile + ((int64_t)l * bm.n_tiles + tile_x) * kBins + tid;
        const int before = row[0];
        pre_base = bm.start[l * kBins + tid] + before;
        if (!kCountHere) n_mine = (tile_x + 1 < bm.n_tiles ? row[kBins] : bm.count[l * kBins + tid]) - before;
    }
    if (kCountHere) {
        for (int b = tid; b < kBins; b += kFillThreads) hist[b] = 0;
        __syncthreads();
    }
    const LevelC L = make_level_c(lv, l, sshift);
    const int64_t i = (int64_t)tile_x * kFillThreads + tid;
    const bool live = i < n_points;
    uint32_t g[3] = {0u, 0u, 0u};
    float w[3] = {0.f, 0.f, 0.f}, d0 = 0.f, d1 = 0.f, r0 = 0.f, r1 = 0.f, e0 = 0.f, e1 = 0.f, qd[3] = {0.f, 0.f, 0.f};
    if (live) {
#if (LS2FM_FILL_PROBE & 2)
        const float fi = (float)(i % 977) * 1.0e-3f;      // timing probe: no record loads
        const float4 a = make_float4(fi, 0.5f * fi + 0.1f, 0.3f, 0.1f), c = make_float4(0.2f, 0.3f, 0.f, 0.f), b = make_float4(fi, fi, 0.1f, 0.2f);
#else
        const float4 a = reinterpret_cast<const float4*>(rpt)[2 * i];
        const float4 c = reinterpret_cast<const float4*>(rpt)[2 * i + 1];
#if LS2FM_FILL_REC_NT
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 bv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rec1) + ((int64_t)l * p_pad + i));
        const float4 b = make_float4(bv[0], bv[1], bv[2], bv[3]);
#else
        const float4 b = reinterpret_cast<const float4*>(rec1)[(int64_t)l * p_pad + i];
#endif
#endif
        const float x[3] = {a.x, a.y, a.z};                 // the grid-normalised position the forward classified (same bits)
#pragma unroll
        for (int q = 0; q < 3; ++q) pos_fract(x[q], L.scale, g[q], w[q]);
        d0 = b.x; d1 = b.y; r0 = b.z; r1 = b.w;
        qd[0] = L.scale * a.w; qd[1] = L.scale * c.x; qd[2] = L.scale * c.y;
        if (DUAL) {
#if LS2FM_FILL_REC_NT
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x2 e = __builtin_nontemporal_load(reinterpret_cast<const f32x2*>(rec2) + ((int64_t)l * p_pad + i));
            e0 = e[0]; e1 = e[1];
#else
            const float2 e = reinterpret_cast<const float2*>(rec2)[(int64_t)l * p_pad + i];
            e0 = e.x; e1 = e.y;
#endif
        }
    }
    // runs of consecutive points in one cell (bin_items.h): the same flags the counting pass derived
    const bool expl = DUAL && l < n_explicit;               // (workgroup-uniform) explicit, run-merged dual items on this level
    const bool halves = DUAL && !expl;                      // factored items: a merged run would be two half items per pair
    const RunFlags rf = wave_runs(g, live, lane, halves ? kMergeMinDual : kMergeMinSingle);
    // ---- ONE pass over the pairs' hashes (rounds 1-5: one per counting / ranking / window pass).  Kept per pair c: ij (both
    // x-corners' local indices, each relative to its own slab), the slab of the pair's first item in byte c of sb0 (bit 7: the pair
    // is split into two half items), the second half item's slab in byte c of sb1.  Nothing here (nor in the factors below) needs
    // the run offsets: the work runs while their loads from the top are in flight.
    // (named scalars: as arrays indexed by the lambda's pair number they went to scratch memory)
    uint32_t ij0 = 0u, ij1 = 0u, ij2 = 0u, ij3 = 0u, sb0 = 0u, sb1 = 0u;
    auto or_ij = [&](unsigned c, uint32_t v) { if (c == 0) ij0 |= v; else if (c == 1) ij1 |= v; else if (c == 2) ij2 |= v; else ij3 |= v; };
    auto get_ij = [&](unsigned c) { return c == 0 ? ij0 : (c == 1 ? ij1 : (c == 2 ? ij2 : ij3)); };
    static_assert(kBins <= 128, "a slab fits seven bits");
    for_each_item_merged(L, g, rf, halves, [&](int slab, unsigned c, uint32_t i0, uint32_t i1) {
        // (selects, not branches: as `if / else` on two variables the compiler made them a two-element array in scratch memory)
        const bool first = i0 != 0xFFFFu;
        or_ij(c, (first ? i0 : 0u) | (i1 != 0xFFFFu ? i1 << 16 : 0u));
        const uint32_t field = ((uint32_t)slab | ((first && i1 == 0xFFFFu) ? 0x80u : 0u)) << (8 * c);
        sb0 |= first ? field : 0u;
        sb1 |= first ? 0u : field;
        if (kCountHere) atomicAdd(&hist[slab], 1);
    });
    // the items of pair c from the kept fields: f(slab, second half item?, ij word of the item)
    auto pair_items = [&](unsigned c, auto&& f) {
        const uint32_t ijc = get_ij(c), s0 = (sb0 >> (8 * c)) & 0xFFu;
        if ((s0 & 0x80u) == 0u) f((int)s0, false, ijc);
        else {
            f((int)(s0 & 0x7Fu), false, (ijc & 0xFFFFu) | 0xFFFF0000u);
            f((int)((sb1 >> (8 * c)) & 0x7Fu), true, (ijc & 0xFFFF0000u) | 0xFFFFu);
        }
    };
    // the two x-corners' values of corner pair c = by + 2 bz, both grids:
    //   corner 0: px0 A - B [, px0 C]      corner 1: wx A + B [, wx C]        (exactly what slab_accumulate forms from a factored item)
    auto pair_factors = [&](unsigned c, float (&a)[2], float (&b)[2], float (&cc)[2]) {
        const int by = c & 1, bz = c >> 1;
        const float py = by ? w[1] : 1.0f - w[1], pz = bz ? w[2] : 1.0f - w[2];
        const float pyz = py * pz;
        const float qyz = (by ? qd[1] : -qd[1]) * pz + py * (bz ? qd[2] : -qd[2]);
        a[0] = fmaf(qyz, r0, pyz * d0);
        a[1] = fmaf(qyz, r1, pyz * d1);
        b[0] = qd[0] * pyz * r0;
        b[1] = qd[0] * pyz * r1;
        cc[0] = pyz * e0;
        cc[1] = pyz * e1;
    };
    float val[DUAL ? 1 : 16];                                // single field: the pairs' explicit corner values
    float fa[DUAL ? 4 : 1][2], fb[DUAL ? 4 : 1][2], fcc[DUAL ? 4 : 1][2];       // dual field, factored items
    if constexpr (!DUAL) {
        // (formed here, ahead of the barriers; behind them -- as in rounds 1-5 -- measured the same: 66.5 vs 66.6 us)
        const float px0 = 1.0f - w[0];
#pragma unroll
        for (unsigned c = 0; c < 4; ++c) {
            float a[2], b[2], cc[2];
            pair_factors(c, a, b, cc);
            val[4 * c + 0] = fmaf(px0, a[0], -b[0]);
            val[4 * c + 1] = fmaf(px0, a[1], -b[1]);
            val[4 * c + 2] = fmaf(w[0], a[0], b[0]);
            val[4 * c + 3] = fmaf(w[0], a[1], b[1]);
        }
        if (rf.cont != 0ull) run_sums<16>(val, rf.cont, lane);          // wave-uniform: this wave has runs to merge
    } else if (!expl) {
#pragma unroll
        for (unsigned c = 0; c < 4; ++c) pair_factors(c, fa[c], fb[c], fcc[c]);
    }
    if (LS2FM_FILL_PRIO == 1) __builtin_amdgcn_s_setprio(0);
    if (LS2FM_FILL_PRIO == 2) __builtin_amdgcn_s_setprio(2);
    // ---- the run lengths (counted above, or the first consumer of the loads issued at the top)
    if (kCountHere) __syncthreads();
    if (tid < kBins) {                   // (slab tid: this thread is its only reader and writer here)
        run_len[tid] = kCountHere ? hist[tid] : n_mine;
        base[tid] = pre_base;
        hist[tid] = 0;                   // running rank
    }
    __syncthreads();
    // runs: offsets in the sorted order (exclusive prefix over the slabs, wave 0)
    if (tid < 64) {
        int run = 0;
#pragma unroll
        for (int c = 0; c < (kBins + 63) / 64; ++c) {
            const int b = 64 * c + lane;
            const int cnt = b < kBins ? run_len[b] : 0;
            const int incl = wave_scan_incl_i32(cnt, lane);
            if (b < kBins) lds_off[b] = run + incl - cnt;
            run += __shfl(incl, 63, 64);
        }
        if (lane == 0) s_total = run;
    }
    __syncthreads();
    // an item's rank inside its (workgroup, slab) run.  Long runs (coarse levels: consecutive samples of a ray fall into the same
    // cell) are stored permuted, so that the 64 lanes of an accumulate wave, which read consecutive items, do not all hit the same
    // entry (same-address LDS atomics serialise): position = rank * K mod n with K prime > n
    auto take_rank = [&](int slab) {
        int rank = atomicAdd(&hist[slab], 1);
        const int n_run = run_len[slab];
        if (n_run > 64) rank = (int)(((long long)rank * 1000003ll) % n_run);
        return rank;
    };
    if constexpr (!DUAL) {
        // ---- single field: 20-byte explicit items, ONE staging area for all of them (26 KB): ranked and staged in one pass
        // over the pairs (the windowed form below enumerates them once more: 52 -> 57 us)
        if (rf.head) {
#pragma unroll
            for (unsigned c = 0; c < 4; ++c)
                pair_items(c, [&](int slab, bool, uint32_t ij) {
                    ItemT it;
                    it.ij = ij;
                    it.v00 = val[4 * c + 0]; it.v01 = val[4 * c + 1];
                    it.v10 = val[4 * c + 2]; it.v11 = val[4 * c + 3];
                    const int rank = take_rank(slab);
                    const int slot = lds_off[slab] + rank;
                    const uint32_t gi = (uint32_t)(base[slab] + rank);
                    if (slot < kWin) { s_items[slot] = it; s_gidx[slot] = gi; }
                    else g_items[gi] = it;                           // more split pairs than the staging area holds: direct write
                });
        }
        __syncthreads();
        const int staged = s_total < kWin ? s_total : kWin;
#if LS2FM_FILL_FLAT
        {
            static_assert(sizeof(ItemT) == 20, "five words per item");
            const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(s_items);
            uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(g_items);
            for (int c = tid; c < 5 * staged; c += kFillThreads) {
                const int q = (int)(((uint32_t)c * 52429u) >> 18);            // c / 5 for c < 2^16 (5 * kFillCap = 5440)
                dst[5u * s_gidx[q] + (uint32_t)(c - 5 * q)] = src[c];
            }
        }
#else
        for (int q = tid; q < staged; q += kFillThreads) g_items[s_gidx[q]] = s_items[q];
#endif
    } else {
    // every item's slot in the workgroup's sorted order
    // (named scalars, 16 bits per slot: as arrays indexed by the lambda's pair number they went to scratch memory)
    uint32_t sl0 = 0u, sl1 = 0u, sl2 = 0u, sl3 = 0u;
    auto put_slot = [&](unsigned c, bool second, int slot) {
        const uint32_t v = second ? (uint32_t)slot << 16 : (uint32_t)slot;
        if (c == 0) sl0 |= v; else if (c == 1) sl1 |= v; else if (c == 2) sl2 |= v; else sl3 |= v;
    };
    auto get_slot = [&](unsigned c, bool second) {
        const uint32_t v = c == 0 ? sl0 : (c == 1 ? sl1 : (c == 2 ? sl2 : sl3));
        return (int)(second ? v >> 16 : v & 0xFFFFu);
    };
    static_assert(kFillThreads * 8 <= 0xFFFF, "slots fit 16 bits");
    if (rf.head) {
#pragma unroll
        for (unsigned c = 0; c < 4; ++c)
            pair_items(c, [&](int slab, bool second, uint32_t) { put_slot(c, second, lds_off[slab] + take_rank(slab)); });
    }
    const int total = s_total;
    auto write_out = [&](int win) {
        // runs of one slab are contiguous in the sorted order and in memory: consecutive threads write consecutive items
        const int staged = total - win < kWin ? total - win : kWin;
        if (LS2FM_FILL_PRIO == 3) __builtin_amdgcn_s_setprio(2);           // (3: raised for the write-out only)
#if (LS2FM_FILL_PROBE & 1)
        if (staged < 0)                    // timing probe: no item stores
#endif
#if LS2FM_FILL_FLAT
#ifdef LS2FM_ITEM_PROBE24
        for (int c = tid; c < 3 * staged; c += kFillThreads) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(2)));          // (8-byte chunks of a 24-byte item)
            const int q = (int)(((uint32_t)c * 43691u) >> 17), hf = c - 3 * q;
#else
        for (int c = tid; c < 2 * staged; c += kFillThreads) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const int q = c >> 1, hf = c & 1;
#endif
#if (LS2FM_FILL_PROBE & 4)
            const uint32_t gi = s_gidx[q] & 0x7FFFu;                                   // timing probe: every store inside a 1 MB window (cache-resident)
#elif (LS2FM_FILL_PROBE & 8)
            const uint32_t gi = (uint32_t)(((int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x) * (kFillCap + 64) + win + q);     // timing probe: the workgroup's items CONTIGUOUS (streaming stores, no scatter)
#else
            const uint32_t gi = s_gidx[q];
#endif
            const u32x4 v = reinterpret_cast<const u32x4*>(s_items)[c];
            u32x4* dst = reinterpret_cast<u32x4*>(&g_items[gi]) + hf;
#if LS2FM_FILL_NT
            __builtin_nontemporal_store(v, dst);
#else
            *dst = v;
#endif
            if (DUAL && expl && hf == 0) bm.extra[gi] = s_extra[q];
        }
#else
        for (int q = tid; q < staged; q += kFillThreads) {
            const uint32_t gi = s_gidx[q];
#if LS2FM_FILL_NT
            if constexpr (DUAL) {
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4* src = reinterpret_cast<const u32x4*>(&s_items[q]);
                u32x4* dst = reinterpret_cast<u32x4*>(&g_items[gi]);
#if LS2FM_FILL_NT == 2
                const u32x4 h0 = src[0], h1 = src[1];
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1" :: "v"(dst), "v"(h0), "v"(h1) : "memory");
#elif LS2FM_FILL_NT == 3
                const u32x4 h0 = src[0], h1 = src[1];
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1" :: "v"(dst), "v"(h0), "v"(h1) : "memory");
#else
                __builtin_nontemporal_store(src[0], dst);
                __builtin_nontemporal_store(src[1], dst + 1);
#endif
            } else
#endif
            g_items[gi] = s_items[q];
            if (DUAL && expl) bm.extra[gi] = s_extra[q];
        }
#endif
        if (LS2FM_FILL_PRIO == 3) __builtin_amdgcn_s_setprio(0);
    };
    if (!expl) {
        // ---- factored dual items (factors formed above, indices and slabs kept from the one hash pass)
        for (int win = 0; win < total; win += kWin) {
            if (win > 0) __syncthreads();                     // the previous window has been written out
            if (rf.head) {
#pragma unroll
                for (unsigned c = 0; c < 4; ++c)
                    pair_items(c, [&](int slab, bool second, uint32_t ij) {
                        const int rel = get_slot(c, second) - win;
                        if (rel < 0 || rel >= kWin) return;
                        ItemT it;
                        it.ij = ij;
                        it.wx = w[0];
                        it.a0 = fa[c][0]; it.a1 = fa[c][1];
                        it.b0 = fb[c][0]; it.b1 = fb[c][1];
#ifndef LS2FM_ITEM_PROBE24
                        it.c0 = fcc[c][0]; it.c1 = fcc[c][1];
#endif
                        s_items[rel] = it;
                        s_gidx[rel] = (uint32_t)(base[slab] + (rel + win - lds_off[slab]));
                    });
            }
            __syncthreads();
            write_out(win);
        }
    } else {
        // ---- explicit dual items, pair by pair: the pair's eight corner values, their sums over the run, its item(s)
        const float px0 = 1.0f - w[0];
        for (int win = 0; win < total; win += kWin) {         // (one window unless nothing merged)
            if (win > 0) __syncthreads();
#pragma unroll 1                                          // (unrolled, the scheduler interleaves the four pairs: 32 live values again)
            for (unsigned c = 0; c < 4; ++c) {
                float a[2], b[2], cc[2], v8[8];
                pair_factors(c, a, b, cc);
                v8[0] = fmaf(px0, a[0], -b[0]); v8[1] = fmaf(px0, a[1], -b[1]);
                v8[2] = fmaf(w[0], a[0], b[0]); v8[3] = fmaf(w[0], a[1], b[1]);
                v8[4] = px0 * cc[0]; v8[5] = px0 * cc[1];
                v8[6] = w[0] * cc[0]; v8[7] = w[0] * cc[1];
                if (rf.cont != 0ull) run_sums<8>(v8, rf.cont, lane);              // wave-uniform
                if (rf.head)
                    pair_items(c, [&](int slab, bool second, uint32_t ij) {
                        const int rel = get_slot(c, second) - win;
                        if (rel < 0 || rel >= kWin) return;
                        s_items[rel] = explicit_item(ij, v8);
                        s_extra[rel] = v8[7];
                        s_gidx[rel] = (uint32_t)(base[slab] + (rel + win - lds_off[slab]));
                    });
            }
            __syncthreads();
            write_out(win);
        }
    }
    }       // (dual field)
    // the level's first workgroup also reduces the per-ray bounds of a single contribution (written by shade_bwd) to the
    // level's bound: the accumulate workgroups read two floats instead of n_rays each
    if (tile_x == 0) {
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
    int scratch[LS2FM_MAX_LEVELS];       // first block of the level's (slab, part) partials in BinMeta::part_acc (parts > 1)
    int headroom_bits;                   // log2 of the worst-case number of contributions to one entry
    int n_explicit;                      // dual field: leading levels with explicit, run-merged items (bin_items.h)
};

// LS2FM_ACC_CVT: how a contribution becomes a 64-bit fixed-point integer.  0: __float2ll_rn(v * to_fixed) -- there is no f32 -> i64
// instruction: the compiler's expansion is ~12 VALU instructions, eight times per item, half of this kernel's SIMD cycles.  1: through
// the double's mantissa -- fma((double) v, to_fixed, 1.5 * 2^52) rounds v * to_fixed (exact in double) to the nearest integer, ties to
// even, exactly as __float2ll_rn does, and leaves it in the low 52 bits of the result (two's complement, |.| < 2^51); what is above
// them is the constant 0x4338 in the top 16 bits, whose low dword is zero: ONE 32-bit subtraction on the high dword.  Three
// instructions (v_cvt_f64_f32, v_fma_f64, v_add_u32) and the same integers bit for bit, PROVIDED |v * to_fixed| < 2^51: a single
// item is at most 64 contributions (a run merged inside a wave, bin_items.h: wave_runs), each < 2^(62 - headroom) -- make_plan
// keeps headroom_bits >= 18 (it is 4 + log2(points): only batches under 16 384 points see the floor, as a 2^-44-of-the-bound
// quantum instead of a finer one).
// Round 6, same session, alternating libraries (profiles/r06_raw/c55_ab_cvt.txt): 1 380 -> 920 VALU instructions in the kernel; slab_accumulate
// 69.1 / 69.3 / 68.8 -> 66.8 / 66.6 / 66.6 us (single field 46.0 -> 45.2, C3 399 -> 397, C5 243 -> 241); gradient digests identical.
#ifndef LS2FM_ACC_CVT
#define LS2FM_ACC_CVT 1
#endif
constexpr int kMinHeadroomBits = LS2FM_ACC_CVT ? 18 : 4;
__device__ __forceinline__ void add_fixed(u64* slot, float v, float to_fixed) {
    // v * to_fixed is exact (power-of-two scale); |.| < 2^62 / worst-case hits by construction of the quantum
#if LS2FM_ACC_CVT
    const double d = fma((double)v, (double)to_fixed, 6755399441055744.0);
    atomicAdd(slot, (u64)__double_as_longlong(d) - 0x4338000000000000ull);
#else
    atomicAdd(slot, (u64)__float2ll_rn(v * to_fixed));        // two's complement: integer sums are exact
#endif
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

// -DLS2FM_STAMPS: per-unit phase time stamps (100 MHz), read back by tools/acc_stamps_p.py
#ifdef LS2FM_STAMPS
__device__ long long g_acc_stamps[8 * 4096];
#endif

// (Rounds 1-4 ran one workgroup per (level, slab[, part]); round 5's persistent form below replaced it -- `git log -S slab_accumulate_kernel`.
// A variant that packed each feature pair into one 64-bit accumulator -- half the LDS atomics -- dropped contributions below 2^-17
// of a level's bound, which Adam does not forgive; measured and removed in round 3 / 4, `git log -S LS2FM_PACKED_ACC`.)
//
// Point-split slabs (dense / tiny levels: every point's items fall into a handful of slabs, whose lists are cut into `parts`
// units) have two flush forms, chosen per launch (ls2fm_set_scatter_mode):
//   combine = 1  every part leaves its 64-bit fixed-point partials in scratch; slab_combine_kernel, the next launch on the stream,
//                sums them -- integer sums: exact and order-independent -- and writes every entry once with a plain store.  The
//                table gradient is then the exactly rounded sum of its contributions on EVERY level, the backward repeats itself
//                bit for bit, nothing is zeroed beforehand and no floating-point atomic is left in the path.
//   combine = 0  float atomics into a table range zeroed by the backward's zero job (rounds 1-3): one launch fewer, sums of
//                rounded partials in arrival order.

// ---- the point-split slabs' partials -> table entries (combine form): entry `entry` of slab `slab` on level l.  Integer sums:
// exact, order-independent.
template <bool DUAL, bool ADD_INTO>
__device__ __forceinline__ void combine_entry(const LevelSet& lv, const SlabPlan& plan, const BinMeta& bm, int sshift, float* __restrict__ dtable1,
                                              float* __restrict__ dtable2, int l, uint32_t slab, int entry, float bound1, float bound2) {
    constexpr int F = DUAL ? 4 : 2;
    const uint32_t lo = slab << sshift;
    const uint32_t hi = lo + (1u << sshift) < lv.size[l] ? lo + (1u << sshift) : lv.size[l];
    if (lo + (uint32_t)entry >= hi) return;
    const int parts = plan.parts[l];
    float to_fixed, to_fixed2;
    double to_float1, to_float2;
    quantum_of(bound1, plan.headroom_bits, to_fixed, to_float1);
    quantum_of(bound2, plan.headroom_bits, to_fixed2, to_float2);
    // partials are feature-major like the accumulators: [part][f * E + entry].  Loads of kCombineBatch parts are issued together
    // (a part beyond the last re-reads the last one and is masked): one memory round trip per batch, not per part
    const int E = 1 << sshift;
    const u64* __restrict__ src = bm.part_acc + (size_t)(plan.scratch[l] + (int)slab * parts) * kAccSlots + entry;
    u64 tot[F];
#pragma unroll
    for (int f = 0; f < F; ++f) tot[f] = 0ull;
    constexpr int kCombineBatch = 4;
    for (int q0 = 0; q0 < parts; q0 += kCombineBatch) {
        u64 v[kCombineBatch][F];
#pragma unroll
        for (int u = 0; u < kCombineBatch; ++u) {
            const int q = q0 + u < parts ? q0 + u : parts - 1;
#pragma unroll
            for (int f = 0; f < F; ++f) v[u][f] = src[(size_t)q * kAccSlots + f * E];
        }
#pragma unroll
        for (int u = 0; u < kCombineBatch; ++u)
            if (q0 + u < parts) {
#pragma unroll
                for (int f = 0; f < F; ++f) tot[f] += v[u][f];
            }
    }
    float2 v1 = make_float2((float)((double)(long long)tot[0] * to_float1), (float)((double)(long long)tot[1] * to_float1));
    float2* d1 = reinterpret_cast<float2*>(dtable1 + 2ull * (lv.offset[l] + lo + entry));
    if (ADD_INTO) {
        if ((tot[0] | tot[1]) != 0ull) { const float2 o = *d1; v1.x += o.x; v1.y += o.y; *d1 = v1; }
    } else *d1 = v1;
    if constexpr (DUAL) {
        float2 v2 = make_float2((float)((double)(long long)tot[2] * to_float2), (float)((double)(long long)tot[3] * to_float2));
        float2* d2 = reinterpret_cast<float2*>(dtable2 + 2ull * (lv.offset[l] + lo + entry));
        if (ADD_INTO) {
            if ((tot[2] | tot[3]) != 0ull) { const float2 o = *d2; v2.x += o.x; v2.y += o.y; *d2 = v2; }
        } else *d2 = v2;
    }
}

// ------------------------------------------------------------------------------------------------ accumulate, persistent
// Round 5.  Per-slab stamps of the kernel above (profiles/r04_acc_stamps.txt): 4.8 us prologue (dispatch of a 1024-thread /
// 128 KB workgroup, two dependent scalar loads, the first item loads' round trip, 128 KB of LDS to zero) + 4.2 us of LDS atomics
// + 1.4 us flush: more than half of a hashed slab's workgroup is latency nothing overlaps, because a 128 KB workgroup is alone on
// its CU.  Here ONE workgroup per CU stays resident and walks slabs ("units", claimed from a counter: the units differ 4x in
// length):
//   * the item loads form one continuous stream of batches across unit boundaries -- while a unit's last batch is being added,
//     the NEXT unit's first batch is already in flight (its meta was read one unit ahead, the unit after that is being claimed);
//   * the flush clears the accumulators it reads: the LDS is zeroed once per workgroup, not once per slab;
//   * barriers wait for the LDS only (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() would also drain the prefetched loads.
// The sums are the same exactly rounded integers in any order: results are bit-identical to the kernel above.
struct AccUnit {
    int l, part, parts, wg, n_items, j_lo, j_hi, uid;  // l < 0: no unit
    int start;                                         // the list's first item (absolute index: the explicit items' ninth word)
    bool expl;                                         // explicit dual items on this level
    uint32_t slab;
    float bound1, bound2;                              // the level's bounds of a single contribution (read a unit ahead)
    const void* list;
};

#ifndef LS2FM_ACC_NT
#define LS2FM_ACC_NT 0
#endif
// an item of a slab's list (read exactly once, by one workgroup): optionally as non-temporal loads
__device__ __forceinline__ Item load_item(const Item* p) {
#if LS2FM_ACC_NT
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    union { u32x4 q[2]; Item it; } u;
    u.q[0] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    u.q[1] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p) + 1);
    return u.it;
#else
    return *p;
#endif
}
__device__ __forceinline__ ItemS load_item(const ItemS* p) {
#if LS2FM_ACC_NT
    union { uint32_t w[5]; ItemS it; } u;
#pragma unroll
    for (int q = 0; q < 5; ++q) u.w[q] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p) + q);
    return u.it;
#else
    return *p;
#endif
}

// one entry of a gradient table, written once by its sole owner and not read again in this launch: plain, non-temporal (1) or
// write-through (2) stores -- the table is the optimizer's input much later, keeping its lines in this XCD's L2 only evicts others
__device__ __forceinline__ void store_entry(float* d, const float2 v) {
#if LS2FM_ACC_NT_STORE == 2
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 r = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(d), "v"(r) : "memory");
#elif LS2FM_ACC_NT_STORE == 1
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 r = {v.x, v.y};
    __builtin_nontemporal_store(r, reinterpret_cast<f32x2*>(d));
#else
    *reinterpret_cast<float2*>(d) = v;
#endif
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// the same barrier, also completing the scalar loads of acc_unit_fetch<.., ASYNC = true> (they count in lgkmcnt): the values
// pass through the statement, so that nothing consuming them can be scheduled in front of it
__device__ __forceinline__ void acc_unit_arrive(int& count, int& start, float& b1, float& b2) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+s"(count), "+s"(start), "+s"(b1), "+s"(b2) :: "memory");
}

// A unit's descriptor in two steps: acc_unit_fetch issues the four loads (item count and list start of the slab, the level's
// bounds), acc_unit_finish consumes them -- the persistent kernel puts a slab's flush between the two.
struct AccUnitRaw { int l, wg, parts, slab, count, start, uid; float bound1, bound2; };

template <bool DUAL, bool ASYNC>
__device__ __forceinline__ AccUnitRaw acc_unit_fetch(const SlabPlan& plan, const BinMeta& bm, int block_base, int unit, int n_units) {
    AccUnitRaw r;
    r.l = -1; r.wg = 0; r.parts = 1; r.slab = 0; r.count = 0; r.start = 0; r.uid = 0; r.bound1 = r.bound2 = 0.f;
    if (unit >= n_units) return r;
    const int bid = block_base + unit;
    r.uid = bid;
    // level of the unit: plan.first is non-decreasing (levels beyond the grid's start at the total) -- counted, not searched (a
    // search is up to 16 DEPENDENT scalar loads from the kernel arguments, ~1.5 us per unit)
    int l = 0;
#pragma unroll
    for (int q = 1; q <= LS2FM_MAX_LEVELS; ++q) l += bid >= plan.first[q] ? 1 : 0;
    r.l = l;
    r.parts = plan.parts[l];
    r.wg = bid - plan.first[l];
    r.slab = r.parts == 1 ? r.wg : __builtin_amdgcn_readfirstlane(r.wg / r.parts);      // (the division runs on the vector ALU)
    const int* pc = bm.count + l * kBins + r.slab;
    const int* ps = bm.start + l * kBins + r.slab;
    const float* pb = bm.level_bound + l;
    if (ASYNC) {
        // SCALAR loads issued by hand and not waited for here: acc_unit_arrive (s_waitcnt lgkmcnt(0), the barrier behind the flush)
        // completes them.  As compiler-issued loads they were vector loads (the kernel writes global memory: no scalar-load
        // proof), either turned into scalar registers on the spot -- a round trip in FRONT of the flush -- or, kept in vector
        // registers, waited for behind the flush's stores, which share vmcnt with them.  The values were written by earlier
        // launches: the scalar cache cannot hold stale copies.
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(r.count) : "s"(pc) : "memory");
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(r.start) : "s"(ps) : "memory");
        asm volatile("s_load_dword %0, %1, 0x0" : "=s"(r.bound1) : "s"(pb) : "memory");
        if (DUAL) asm volatile("s_load_dword %0, %1, 0x40" : "=s"(r.bound2) : "s"(pb) : "memory");
    } else {
        r.count = *pc;
        r.start = *ps;
        r.bound1 = pb[0];
        r.bound2 = DUAL ? pb[16] : 0.f;
    }
    return r;
}

template <typename ItemT>
__device__ __forceinline__ AccUnit acc_unit_finish(const AccUnitRaw& r, const BinMeta& bm, int n_explicit) {
    AccUnit u;
    u.l = r.l; u.parts = r.parts; u.wg = r.wg; u.slab = (uint32_t)r.slab; u.part = r.wg % r.parts; u.uid = r.uid;
    u.n_items = r.count; u.bound1 = r.bound1; u.bound2 = r.bound2;
    u.start = r.start;
    u.expl = r.l >= 0 && r.l < n_explicit;
    u.list = reinterpret_cast<const ItemT*>(bm.items) + r.start;
    if (u.parts == 1) { u.j_lo = 0; u.j_hi = u.n_items; }
    else {            // floor(n part / parts) without a 64-bit division: n = q parts + r
        const uint32_t q = (uint32_t)u.n_items / (uint32_t)u.parts, rem = (uint32_t)u.n_items % (uint32_t)u.parts;
        u.j_lo = (int)(q * (uint32_t)u.part + rem * (uint32_t)u.part / (uint32_t)u.parts);
        u.j_hi = (int)(q * (uint32_t)(u.part + 1) + rem * (uint32_t)(u.part + 1) / (uint32_t)u.parts);
    }
    return u;
}

#ifndef LS2FM_ACC_PROBE
#define LS2FM_ACC_PROBE 0
#endif
#ifdef LS2FM_STAMPS
#define PACC_STAMP(uid, k) do { if (threadIdx.x == 0 && (uid) < 4096) g_acc_stamps[8 * (uid) + (k)] = wall_clock64(); } while (0)
#else
#define PACC_STAMP(uid, k) do {} while (0)
#endif

template <bool DUAL, bool ADD_INTO>
__global__ void __launch_bounds__(kAccThreads)
slab_accumulate_persistent_kernel(LevelSet lv, SlabPlan plan, BinMeta bm, int sshift, float* __restrict__ dtable1,
                                  float* __restrict__ dtable2, int block_base, int n_units, int combine, int* __restrict__ claim, int hold) {
    typedef typename ItemOf<DUAL>::type ItemT;
    constexpr int F = DUAL ? 4 : 2;
    constexpr int BT = kAccBatch * kAccThreads;
    __shared__ __attribute__((aligned(16))) u64 acc[kAccSlots];
    __shared__ int s_claim, s_claim2;
    const int tid = threadIdx.x;
    const int G = (int)gridDim.x;
    const int E = 1 << sshift;
    const int n_exp = DUAL ? plan.n_explicit : 0;
    // Every workgroup starts with two units in hand (the second one's first batch is requested behind the first one's last) and
    // claims a third from a counter (zeroed by scatter_fill's first workgroup) while it works on the first, and so on: up to
    // three units per workgroup when the pool runs dry -- the kernel ended 12 us after its AVERAGE workgroup (per-unit stamps:
    // end times 48.8 / 54.8 / 67.3 us min / mean / max; list scheduling of the same unit times: 57.8), and its last workgroup
    // was always a part of the level-0 slab (50 us with the combine) + the two hashed units it had been sitting on.  So:
    //   * unit ids run level by level, coarse first; the first `hold` are LONG (parts of point-split slabs).  Their workgroups
    //     get the LAST ids as their second unit -- work that is due at the end anyway -- and claim nothing while they work on
    //     the long one (behind it: once, if the pool is not yet down to its zone -- see the flush);
    //   * the others take G + b - hold, then claim -- until the unit they hold as `next` is one of the pool's last G - hold
    //     (the zone): every claiming workgroup takes at most one unit of the zone, as its last.  (Every id is still taken: if
    //     one of the zone were not, the counter would have stopped below the pool's end, so every claiming workgroup would have
    //     stopped claiming, so each would hold a zone id -- G - hold distinct ones: all of them.)
    const int b = (int)blockIdx.x;
    const int n_hold = hold > 0 ? hold : 0;
    const int second = b < n_hold ? n_units - 1 - b : G + b - n_hold;
    const int dyn_base = 2 * G - n_hold, dyn_end = n_units - n_hold;
    const int zone_lo = hold >= 0 && n_units >= 3 * G ? dyn_end - (G - n_hold) : 0x7fffffff;
    AccUnit cur = acc_unit_finish<ItemT>(acc_unit_fetch<DUAL, false>(plan, bm, block_base, b, n_units), bm, n_exp);
    AccUnit nxt = acc_unit_finish<ItemT>(acc_unit_fetch<DUAL, false>(plan, bm, block_base, second, n_units), bm, n_exp);
    ItemT buf[kAccBatch];
    float bufx[DUAL ? kAccBatch : 1];         // ninth word of explicit dual items
    {
        const ItemT* __restrict__ list = reinterpret_cast<const ItemT*>(cur.list);
        const int j_last = cur.j_hi > cur.j_lo ? cur.j_hi - 1 : cur.j_lo;
#pragma unroll
        for (int u = 0; u < kAccBatch; ++u) {
            const int j = cur.j_lo + tid + u * kAccThreads;
            buf[u] = load_item(list + (j < cur.j_hi ? j : j_last));
            if (DUAL) bufx[u] = cur.expl ? bm.extra[cur.start + (j < cur.j_hi ? j : j_last)] : 0.f;
        }
    }
    {
        const uint4 z = make_uint4(0u, 0u, 0u, 0u);
        for (int e = tid; e < kAccSlots / 2; e += kAccThreads) reinterpret_cast<uint4*>(acc)[e] = z;
    }
    lds_barrier();
    bool first = true;
    while (cur.l >= 0) {
        // the unit after the next: claimed in this unit's first batch, read behind its streaming phase -- unless the NEXT unit is
        // already one of the zone (a tail id, or one of the pool's last G - hold): no claim behind it
        int claimed = n_units;               // (not claimed: an id beyond the last unit)
        const bool claim_more = nxt.l >= 0 && nxt.uid - block_base < zone_lo;
        const int l = cur.l;
        PACC_STAMP(cur.uid, 0);
        float to_fixed1, to_fixed2;
        double to_float1, to_float2;
        quantum_of(cur.bound1, plan.headroom_bits, to_fixed1, to_float1);
        quantum_of(cur.bound2, plan.headroom_bits, to_fixed2, to_float2);
        auto add_item = [&](const ItemT& it, float x9) {
#if (LS2FM_FILL_PROBE & 12)
            const uint32_t i0 = it.ij & 0xFFFu, i1 = (it.ij >> 16) & 0xFFFu;       // (the fill probes leave stale / foreign items in the lists: stay inside the slab)
#else
            const uint32_t i0 = it.ij & 0xFFFFu, i1 = it.ij >> 16;
#endif
            if constexpr (DUAL) {
                if (cur.expl) {                // explicit item (bin_items.h: explicit_item): the corners' values as they are
                    if (i0 != 0xFFFFu) {
                        u64* slot = acc + i0;
                        add_fixed(slot, it.wx, to_fixed1);
                        add_fixed(slot + E, it.a0, to_fixed1);
                        add_fixed(slot + 2 * E, it.b1, to_fixed2);
                        add_fixed(slot + 3 * E, LS2FM_ITEM_C0(it), to_fixed2);
                    }
                    if (i1 != 0xFFFFu) {
                        u64* slot = acc + i1;
                        add_fixed(slot, it.a1, to_fixed1);
                        add_fixed(slot + E, it.b0, to_fixed1);
                        add_fixed(slot + 2 * E, LS2FM_ITEM_C1(it), to_fixed2);
                        add_fixed(slot + 3 * E, x9, to_fixed2);
                    }
                    return;
                }
                const float px0 = 1.0f - it.wx;
                if (i0 != 0xFFFFu) {
                    u64* slot = acc + i0;
                    add_fixed(slot, fmaf(px0, it.a0, -it.b0), to_fixed1);
                    add_fixed(slot + E, fmaf(px0, it.a1, -it.b1), to_fixed1);
                    add_fixed(slot + 2 * E, px0 * LS2FM_ITEM_C0(it), to_fixed2);
                    add_fixed(slot + 3 * E, px0 * LS2FM_ITEM_C1(it), to_fixed2);
                }
                if (i1 != 0xFFFFu) {
                    u64* slot = acc + i1;
                    add_fixed(slot, fmaf(it.wx, it.a0, it.b0), to_fixed1);
                    add_fixed(slot + E, fmaf(it.wx, it.a1, it.b1), to_fixed1);
                    add_fixed(slot + 2 * E, it.wx * LS2FM_ITEM_C0(it), to_fixed2);
                    add_fixed(slot + 3 * E, it.wx * LS2FM_ITEM_C1(it), to_fixed2);
                }
            } else {
                if (i0 != 0xFFFFu) {
                    add_fixed(acc + i0, it.v00, to_fixed1);
                    add_fixed(acc + E + i0, it.v01, to_fixed1);
                }
                if (i1 != 0xFFFFu) {
                    add_fixed(acc + i1, it.v10, to_fixed1);
                    add_fixed(acc + E + i1, it.v11, to_fixed1);
                }
            }
        };
        // ---- streaming: batches of this unit; behind the last one the first batch of the next unit is requested
        int b0 = cur.j_lo;
        do {
            ItemT now[kAccBatch];
            float nowx[DUAL ? kAccBatch : 1];
#pragma unroll
            for (int u = 0; u < kAccBatch; ++u) { now[u] = buf[u]; if (DUAL) nowx[u] = bufx[u]; }
            const bool more = b0 + BT < cur.j_hi;                       // (uniform)
            const ItemT* __restrict__ list_n = reinterpret_cast<const ItemT*>(more ? cur.list : nxt.list);
            const int lo_n = more ? b0 + BT : nxt.j_lo, hi_n = more ? cur.j_hi : nxt.j_hi;
            const int last_n = hi_n > lo_n ? hi_n - 1 : lo_n;
            const bool expl_n = DUAL && (more ? cur.expl : nxt.expl);    // (uniform)
            const float* __restrict__ extra_n = bm.extra + (more ? cur.start : nxt.start);
            // (issue-priority experiment, round 6: LS2FM_ACC_PRIO = 1 raises a wave while it issues the next batch's loads.  Default 0.)
#ifndef LS2FM_ACC_PRIO
#define LS2FM_ACC_PRIO 0
#endif
            if (LS2FM_ACC_PRIO == 1) __builtin_amdgcn_s_setprio(2);
#pragma unroll
            for (int u = 0; u < kAccBatch; ++u) {                        // unconditional loads, masked where they are used
                const int j = lo_n + tid + u * kAccThreads;
                buf[u] = load_item(list_n + (j < hi_n ? j : last_n));
                if (DUAL) bufx[u] = expl_n ? extra_n[j < hi_n ? j : last_n] : 0.f;
            }
            if (LS2FM_ACC_PRIO == 1) __builtin_amdgcn_s_setprio(0);
            // The claim: lane 0 of wave 0, issued WITHOUT waiting for the returned value (as a builtin under `if (tid == 0)` the
            // compiler waits for it at the end of the branch: one memory round trip per unit in front of wave 0's adds), BEHIND
            // the batch loads just issued: every vmcnt the compiler computes for loads older than the atomic is merely one too
            // strict, and loads younger than it complete after it (in-order return) -- its bookkeeping stays valid.
            if (tid < 64 && b0 == cur.j_lo && claim_more) {
                unsigned long long saved;
                asm volatile("s_mov_b64 %1, exec\n\t"
                             "s_mov_b64 exec, 1\n\t"
                             "global_atomic_add %0, %2, %3, %4 sc0\n\t"
                             "s_mov_b64 exec, %1"
                             : "=&v"(claimed), "=&s"(saved) : "v"(0), "v"(1), "s"(claim) : "memory");
            }
#if (LS2FM_ACC_PROBE & 1)
            if (now[0].ij == 0x12345678u)        // timing probe: (almost) no LDS atomics
#endif
            if (LS2FM_ACC_PRIO == 2) __builtin_amdgcn_s_setprio(2);            // (2: raised for the adds)
#pragma unroll
            for (int u = 0; u < kAccBatch; ++u)
                if (b0 + tid + u * kAccThreads < cur.j_hi) add_item(now[u], DUAL ? nowx[u] : 0.f);
            if (LS2FM_ACC_PRIO == 2) __builtin_amdgcn_s_setprio(0);
            b0 += BT;
        } while (b0 < cur.j_hi);
        PACC_STAMP(cur.uid, 1);
        if (tid < 64) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(claimed) :: "memory");
            if (tid == 0) s_claim = claimed;
        }
        lds_barrier();
        PACC_STAMP(cur.uid, 2);
        AccUnitRaw nraw = acc_unit_fetch<DUAL, true>(plan, bm, block_base, dyn_base + __builtin_amdgcn_readfirstlane(s_claim), dyn_end);
        // ---- flush + clear
        const uint32_t size = lv.size[l];
        const uint32_t lo = cur.slab << sshift;
        const uint32_t hi = lo + (1u << sshift) < size ? lo + (1u << sshift) : size;
        if (cur.parts > 1 && combine) {
            // A part of a point-split slab: its fixed-point partials go to scratch; the LAST part of the slab to get here (a ticket
            // per slab, zeroed by scatter_fill) sums all of them -- integers: exact, order-independent -- and writes the entries.
            // No combine launch behind this kernel (11 us at the end of the backward's chain), and no workgroup ever waits for
            // another: nothing to deadlock on.  Hand-off: plain stores, a full barrier (every wave's stores have left), one lane's
            // agent-scope release + drained counter, then the ticket; the last arriver acquires at agent scope before it reads.
            // (ADD_INTO: an empty slab has nothing to add -- its parts skip all of this.)
            if (!(ADD_INTO && cur.n_items == 0)) {
                uint4* mine = reinterpret_cast<uint4*>(bm.part_acc + (size_t)(plan.scratch[l] + cur.wg) * kAccSlots);
                const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                for (int e = tid; e < kAccSlots / 2; e += kAccThreads) {
                    mine[e] = reinterpret_cast<const uint4*>(acc)[e];
                    reinterpret_cast<uint4*>(acc)[e] = z;
                }
                __syncthreads();
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const int arrived = __hip_atomic_fetch_add(&bm.part_ticket[l * kBins + (int)cur.slab], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int last = arrived == cur.parts - 1;
                    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    s_claim2 = last;
                }
                __syncthreads();
                if (s_claim2) {
                    for (int entry = tid; entry < E; entry += kAccThreads)
                        combine_entry<DUAL, ADD_INTO>(lv, plan, bm, sshift, dtable1, dtable2, l, cur.slab, entry, cur.bound1, cur.bound2);
                }
            }
        } else if (!(ADD_INTO && cur.n_items == 0)) {
            float* dst1 = dtable1 + 2ull * (lv.offset[l] + lo);
            float* dst2 = DUAL ? dtable2 + 2ull * (lv.offset[l] + lo) : nullptr;
            for (int entry = tid; entry < (int)(hi - lo); entry += kAccThreads) {
                u64 tot[F];
#pragma unroll
                for (int f = 0; f < F; ++f) { tot[f] = acc[f * E + entry]; acc[f * E + entry] = 0ull; }
                float2 v1 = make_float2((float)((double)(long long)tot[0] * to_float1), (float)((double)(long long)tot[1] * to_float1));
                float* d1 = dst1 + 2 * entry;
                if (cur.parts > 1) {
                    if (tot[0] != 0ull) atomicAdd(d1, v1.x);
                    if (tot[1] != 0ull) atomicAdd(d1 + 1, v1.y);
                } else if (ADD_INTO) {
                    if ((tot[0] | tot[1]) != 0ull) { const float2 o = *reinterpret_cast<float2*>(d1); v1.x += o.x; v1.y += o.y; *reinterpret_cast<float2*>(d1) = v1; }
                } else store_entry(d1, v1);
                if constexpr (DUAL) {
                    float2 v2 = make_float2((float)((double)(long long)tot[2] * to_float2), (float)((double)(long long)tot[3] * to_float2));
                    float* d2 = dst2 + 2 * entry;
                    if (cur.parts > 1) {
                        if (tot[2] != 0ull) atomicAdd(d2, v2.x);
                        if (tot[3] != 0ull) atomicAdd(d2 + 1, v2.y);
                    } else if (ADD_INTO) {
                        if ((tot[2] | tot[3]) != 0ull) { const float2 o = *reinterpret_cast<float2*>(d2); v2.x += o.x; v2.y += o.y; *reinterpret_cast<float2*>(d2) = v2; }
                    } else store_entry(d2, v2);
                }
            }
        }
        if (first && b < n_hold) {
            // a long unit's workgroup joins the claiming ones behind it -- one exposed round trip, once -- unless the pool is
            // already down to its zone (the level-0 slab's combining part at C2: it ends with the tail unit it holds).  Without
            // this the `hold` workgroups idle once their two units are done: 4096 rays, 200 of 256 workgroups, accumulate 253 -> 510 us
            lds_barrier();                   // (every thread has read s_claim for this unit's fetch)
            if (tid == 0) {
                const int seen = __hip_atomic_load(claim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_claim = dyn_base + seen < zone_lo ? atomicAdd(claim, 1) : n_units;
            }
            __syncthreads();
            nraw = acc_unit_fetch<DUAL, true>(plan, bm, block_base, dyn_base + __builtin_amdgcn_readfirstlane(s_claim), dyn_end);
        }
        first = false;
        PACC_STAMP(cur.uid, 3);
#ifdef LS2FM_STAMPS
        if (tid == 0 && cur.uid < 4096) { g_acc_stamps[8 * cur.uid + 4] = l; g_acc_stamps[8 * cur.uid + 5] = cur.j_hi - cur.j_lo; g_acc_stamps[8 * cur.uid + 7] = blockIdx.x; }
#endif
        acc_unit_arrive(nraw.count, nraw.start, nraw.bound1, nraw.bound2);
        PACC_STAMP(cur.uid, 6);
        cur = nxt;
        nxt = acc_unit_finish<ItemT>(nraw, bm, n_exp);
    }
}

bool levels_fit(const ls2fm_grid_desc* grid, int sshift) {
    for (int l = 0; l < grid->n_levels; ++l)
        if (level_slabs(grid->size[l], sshift) > kSlabBins) return false;
    return true;
}

}  // namespace

// floats of workspace the scatter needs: meta + worst case 8 items (4 pairs, each split) of 32 bytes per (point, level) + the
// explicit dual items' ninth word
int64_t ls2fm_bins_workspace_floats(int n_levels, int64_t n_points) {
#ifndef LS2FM_ITEM_PROBE24
    static_assert(sizeof(Item) == 32 && sizeof(ItemS) == 20, "item layouts");
#endif
    return meta_ints(n_points) + part_acc_floats(n_points) + extra_floats(n_points) + 8 * 8 * (int64_t)n_levels * n_points + 64;
}

// Which leading levels of a dual-field render use explicit, run-merged items (bin_items.h): cells at least one sample spacing
// wide for a ray that crosses the box along an axis -- resolution <= samples per ray (ETH3D at 128 samples: levels 0 .. 4, 7 ..
// 1.4 samples per cell; measured at C2: 0 / 3 / 4 / 5 explicit levels -> step 0.519 / 0.506 / 0.504 / 0.502 ms); a prefix of the
// levels; LS2FM_EXPLICIT_LEVELS overrides (0: the factored item everywhere, rounds 1-4).
int ls2fm_explicit_levels(const ls2fm_grid_desc* grid, int dual, int n_samples) {
    static const int forced = [] { const char* e = getenv("LS2FM_EXPLICIT_LEVELS"); return e ? atoi(e) : -1; }();
    if (!dual || !grid) return 0;
    int n = 0;
    if (forced >= 0) n = forced;
    else
        while (n < grid->n_levels && n < kMaxExplicitLevels && (long long)grid->resolution[n] <= n_samples) ++n;
    if (n > kMaxExplicitLevels) n = kMaxExplicitLevels;
    if (n > grid->n_levels) n = grid->n_levels;
    return n;
}

size_t ls2fm_bin_counts_bytes() { return 0; }      // nothing to zero: every count is written, not accumulated

bool ls2fm_bins_levels_fit(const ls2fm_grid_desc* grid, int dual) { return levels_fit(grid, ls2fm_slab_shift(dual)); }

// payloads from shade_bwd's records, sorted by slab
int ls2fm_launch_scatter_fill(const ls2fm_grid_desc* grid, const FieldC& fc, const float* center, const float* ray, float* bins_ws,
                              int64_t n_points, int64_t p_pad, const float* rec1, const float* rec2, const float* rpt,
                              const float* ray_bound, int64_t n_rays, int dual, hipStream_t stream, int level_lo, int level_hi,
                              int n_explicit, const void* side_jobs) {
    const int64_t r_pad = (n_rays + 63) / 64 * 64;
    const int sshift = ls2fm_slab_shift(dual);
    const BinMeta bm = make_bin_meta(bins_ws, n_points);
    const LevelSet lv = make_level_set(grid);
    if (level_hi < 0) level_hi = grid->n_levels;
    if (level_hi <= level_lo) return LS2FM_OK;
    // (most recently written lists are read first: slab_accumulate 101 -> 98 us at C2; LS2FM_FILL_REVERSE=0 for the A/B)
    static const int reverse = [] { const char* e = getenv("LS2FM_FILL_REVERSE"); return e ? atoi(e) : 1; }();
    SideJobs sj{};
    if (side_jobs) {
        sj = *static_cast<const SideJobs*>(side_jobs);
        const int n_jobs = sj.dec_blocks + sj.l1_jobs + sj.n_red + kFinalizeTasks;
        sj.rows = (n_jobs + bm.n_tiles - 1) / bm.n_tiles;
    }
    if (kFillThreads != kWmThreads && sj.rows > 0) return LS2FM_ERR_UNSUPPORTED;      // the side jobs are written for 256-thread workgroups
    const dim3 g((unsigned)bm.n_tiles, (unsigned)(level_hi - level_lo + sj.rows));
    if (dual) scatter_fill_kernel<true><<<g, kFillThreads, 0, stream>>>(lv, fc, center, ray, n_points, p_pad, sshift, rpt, rec1, rec2, ray_bound, n_rays, r_pad, bm, level_lo, reverse, n_explicit, sj);
    else      scatter_fill_kernel<false><<<g, kFillThreads, 0, stream>>>(lv, fc, center, ray, n_points, p_pad, sshift, rpt, rec1, nullptr, ray_bound, n_rays, r_pad, bm, level_lo, reverse, 0, sj);
    return ls2fm_launch_status();
}

namespace {
struct HostPlan { SlabPlan plan; int total, zero_lo, zero_hi; };

// 1: the point-split levels are combined in fixed point by the last part of a slab to finish (deterministic, exactly rounded; default)
// 0: float atomics into a zeroed range (rounds 1-3)
std::atomic<int> g_scatter_mode{[] { const char* e = getenv("LS2FM_SCATTER_MODE"); return e ? atoi(e) : 1; }()};

// n_explicit / n_samples: on the dual field's explicit levels consecutive samples of a ray in one cell travel as one item
// (bin_items.h) -- ~0.9 n_samples / resolution of them (measured at C2: 7.0 / 4.4 / 3.0 / 2.0 on levels 0 .. 3): fewer items
// per slab, fewer parts to cut its list into
HostPlan make_plan(const ls2fm_grid_desc* grid, int64_t n_points, int sshift, int n_explicit = 0, int n_samples = 1) {
    const int capacity = part_blocks_capacity(n_points);
    for (int max_parts = kMaxParts;; --max_parts) {
        HostPlan h{};
        for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) h.plan.parts[l] = 1;
        h.plan.headroom_bits = 4;                // 8 corners per point (+1)
        while ((1ll << (h.plan.headroom_bits - 4)) < n_points) ++h.plan.headroom_bits;
        if (h.plan.headroom_bits < kMinHeadroomBits) h.plan.headroom_bits = kMinHeadroomBits;        // (add_fixed, LS2FM_ACC_CVT)
        h.plan.n_explicit = n_explicit < LS2FM_MAX_LEVELS ? n_explicit : 0;        // (the kernels' flag is a dual-field one)
        h.zero_lo = h.zero_hi = -1;
        const int64_t target = 16384;            // items a workgroup should process (a hashed-level slab sees ~4P/slabs)
        int scratch = 0;
        for (int l = 0; l < LS2FM_MAX_LEVELS + 1; ++l) {
            h.plan.first[l] = h.total;
            if (l >= grid->n_levels) continue;
            const int slabs = level_slabs(grid->size[l], sshift);
            int parts = 1;
            if (!grid->hashed[l] || slabs < 16) {
                // dense / tiny level: 4 pair items per point spread over few slabs -> split the slabs' lists
                int64_t per_block = 4 * n_points / slabs;
                if (l < n_explicit) {
                    const double run = 0.9 * (double)n_samples / (double)grid->resolution[l];
                    if (run > 1.0) per_block = (int64_t)((double)per_block / run);
                }
                parts = (int)((per_block + target - 1) / target);
                if (parts > max_parts) parts = max_parts;
                if (parts < 1) parts = 1;
            }
            h.plan.parts[l] = parts;
            h.plan.scratch[l] = scratch;
            if (parts > 1) {
                scratch += slabs * parts;        // one block of partials per (slab, part)
                if (h.zero_lo < 0) h.zero_lo = l;
                h.zero_hi = l;
            }
            h.total += slabs * parts;
        }
        if (scratch <= capacity || max_parts == 1) return h;     // (never more blocks than n_points / 2048 per level: fits)
    }
}
}  // namespace

extern "C" int ls2fm_set_scatter_mode(int mode) {
    if (mode != 0 && mode != 1) return LS2FM_ERR_INVALID_ARGUMENT;
    g_scatter_mode.store(mode);
    return LS2FM_OK;
}
extern "C" int ls2fm_get_scatter_mode(void) { return g_scatter_mode.load(); }

// float-atomic flush of the point-split coarse levels (mode 0): their range of the gradient table(s) is zeroed first (by
// leading workgroups of the shade_bwd launch); mode 1: every entry is written once with a plain store, nothing to zero
void ls2fm_scatter_zero_range(const ls2fm_grid_desc* grid, int64_t n_points, bool dual, int64_t* first, int64_t* count) {
    const HostPlan h = make_plan(grid, n_points, ls2fm_slab_shift(dual ? 1 : 0));
    *first = 0; *count = 0;
    if (g_scatter_mode.load() != 0 || h.zero_lo < 0) return;   // levels in between that have a sole owner are overwritten afterwards anyway
    *first = grid->offset[h.zero_lo];
    *count = (int64_t)grid->offset[h.zero_hi] + grid->size[h.zero_hi] - *first;
}

#ifdef LS2FM_STAMPS
extern "C" int ls2fm_debug_acc_stamps(long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_acc_stamps), sizeof(long long) * 8 * 4096) == hipSuccess ? 0 : -1;
}
#endif

// dtable1 (and dtable2: dual field, both grids in one pass) are OVERWRITTEN over the whole grid (mode 0: the zero job of
// ls2fm_scatter_zero_range must have run on them before).  add_into: the sums are ADDED to the tables' current values instead
// (entries without items untouched, nothing zeroed beforehand): a second gradient producer, ordered behind the first one by the caller.
int ls2fm_launch_slab_accumulate(const ls2fm_grid_desc* grid, float* bins_ws, int64_t n_points, float* dtable1, float* dtable2,
                                 hipStream_t stream, int level_lo, int level_hi, int add_into, int n_explicit, int n_samples) {
    const bool dual = dtable2 != nullptr;
    const int sshift = ls2fm_slab_shift(dual ? 1 : 0);
    const BinMeta bm = make_bin_meta(bins_ws, n_points);
    const LevelSet lv = make_level_set(grid);
    // (single field: explicit, run-merged items on every level)
    const HostPlan h = make_plan(grid, n_points, sshift, dual ? n_explicit : LS2FM_MAX_LEVELS, n_samples);
    const int combine = g_scatter_mode.load() != 0 ? 1 : 0;
    if (level_hi < 0) level_hi = grid->n_levels;
    const int base = h.plan.first[level_lo], blocks = h.plan.first[level_hi] - base;
    if (blocks <= 0) return LS2FM_OK;
    // one resident workgroup per CU walks the units (claimed from a counter the scatter_fill launch in front has zeroed)
    static const int n_cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
        return n;
    }();
    {
        const int g = blocks < n_cus ? blocks : n_cus;
        int* claim = acc_claim(bm);
        // leading units that are parts of point-split slabs (the kernel's `hold`; -1: no tail units, no zone)
        int hold = 0;
        for (int l = level_lo; l < level_hi && h.plan.parts[l] > 1; ++l) hold += h.plan.first[l + 1] - h.plan.first[l];
        if (hold > g || blocks < 3 * g) hold = 0;
        // LS2FM_ACC_HOLD=0 (read once per process): second units G + b for every workgroup, claims to the pool's end -- the schedule of
        // the round's first half, for A/B runs; the table gradients are the same bits either way (exact sums)
        static const int hold_env = [] { const char* e = getenv("LS2FM_ACC_HOLD"); return e ? atoi(e) : 1; }();
        if (!hold_env) hold = -1;
        if (dual)
            (add_into ? slab_accumulate_persistent_kernel<true, true> : slab_accumulate_persistent_kernel<true, false>)<<<g, kAccThreads, 0, stream>>>(
                lv, h.plan, bm, sshift, dtable1, dtable2, base, blocks, combine, claim, hold);
        else
            (add_into ? slab_accumulate_persistent_kernel<false, true> : slab_accumulate_persistent_kernel<false, false>)<<<g, kAccThreads, 0, stream>>>(
                lv, h.plan, bm, sshift, dtable1, nullptr, base, blocks, combine, claim, hold);
    }
    return ls2fm_launch_status();
}
