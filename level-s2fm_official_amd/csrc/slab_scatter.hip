// Hash-table gradient scatter WITHOUT table-wide global atomics (gfx950).
//
// Measured on MI355X (profiles/r01_*): device-scope fp32 atomics run at ~17 G/s -- they execute at the memory side
// because the 8 XCD L2s are not coherent -- which made the classic tcnn-style scatter 80 % of the whole step.
// Instead each workgroup OWNS one 128 KiB slab (16384 entries x 2 features) of one level's gradient table in LDS,
// scans the sample points, and accumulates the corners that land in its slab with LDS atomics (ds_add_f32); the
// slab is then written out with plain coalesced stores.  Every table entry belongs to exactly one slab, so the
// table is overwritten in full (no zero-fill by the caller).
//
// Work item = (level, slab, point-part).  Levels with fewer slabs than a balanced load allows (the coarse dense
// levels: every point lands in their 1..4 slabs) are additionally split over `parts` workgroups by point chunks;
// only those few small levels are flushed with global atomics into a zeroed region.
//
// Round per 8192 points:
//   test    : hashed level: entry = (cx ^ cy*P1 ^ cz*P2) & (2^k - 1) with cx < 2^14, so the slab id (entry bits >= 14)
//             depends on the (y,z) cell only -> 4 candidate slabs per point, ~35 instructions.
//             dense level: range overlap of [first corner, last corner] with the slab.
//             survivors are appended to an LDS queue with one wave-aggregated LDS atomic per wave
//   process : queue entries are evaluated in full (8 corners; first-order trilinear weights and, for the SDF grid,
//             the double-backward derivative weights) by densely packed lanes, in a decorrelated order; the
//             per-(point, level) payload is one 32-byte (SDF grid) or 8-byte (second grid) record
#include "render_common.h"

namespace {

constexpr int kSlabShift = 14;
constexpr int kSlabEntries = 1 << kSlabShift;
constexpr int kChunk = 8192;
constexpr int kThreads = 1024;
constexpr int kMaxParts = 16;

struct SlabPlan {
    int first[LS2FM_MAX_LEVELS + 1];     // first work item of every level
    int parts[LS2FM_MAX_LEVELS];         // point-parts per slab of the level
};

__device__ __forceinline__ uint32_t wrap_index(uint32_t idx, uint32_t size) {
    if (idx >= size) {                   // in-range points: at most one wrap (size >= res^3)
        idx -= size;
        if (idx >= size) idx %= size;    // only for positions far outside the unit cube
    }
    return idx;
}

__device__ __forceinline__ uint32_t level_index(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res, uint32_t size,
                                                uint32_t mask, bool hashed_pow2, uint32_t hashed) {
    if (hashed_pow2) return (cx ^ (cy * LS2FM_PRIME_Y) ^ (cz * LS2FM_PRIME_Z)) & mask;
    if (hashed) return (cx ^ (cy * LS2FM_PRIME_Y) ^ (cz * LS2FM_PRIME_Z)) % size;
    return wrap_index(cx + cy * res + cz * res * res, size);
}

// REC = floats per (point, level) record: 8 (de0 de1 rr0 rr1 gn0 gn1 gn2 -) or 2 (de0 de1)
template <bool SECOND_ORDER>
__global__ void __launch_bounds__(kThreads)
slab_scatter_kernel(LevelSet lv, SlabPlan plan, const float4* __restrict__ x4, int64_t n_points, int64_t p_pad,
                    const float* __restrict__ rec, float* __restrict__ dtable) {
    constexpr int REC = SECOND_ORDER ? 8 : 2;
    __shared__ float acc[2 * kSlabEntries];
    __shared__ unsigned short queue[kChunk];
    __shared__ int q_count;
    const int tid = threadIdx.x, lane = tid & 63;
    int l = 0;
    while ((int)blockIdx.x >= plan.first[l + 1]) ++l;
    const int parts = plan.parts[l];
    const uint32_t item = blockIdx.x - plan.first[l];
    const uint32_t slab = item / parts;
    const int part = (int)(item % parts);
    const uint32_t size = lv.size[l], res = lv.res[l], hashed = lv.hashed[l];
    const float scale = lv.scale[l];
    const uint32_t lo = slab << kSlabShift;
    const uint32_t hi = lo + kSlabEntries < size ? lo + kSlabEntries : size;
    const uint32_t mask = size - 1u;
    const bool pow2 = (size & mask) == 0u;
    const bool fast_hash = hashed && pow2 && size > (uint32_t)kSlabEntries;
    const bool single_slab = size <= (uint32_t)kSlabEntries;
    const uint32_t span = 1u + res + res * res;             // last corner - first corner on a dense level
    const float* __restrict__ rec_l = rec + (int64_t)l * p_pad * REC;

    for (int e = tid; e < 2 * kSlabEntries; e += kThreads) acc[e] = 0.f;

    for (int64_t c0 = (int64_t)part * kChunk; c0 < n_points; c0 += (int64_t)parts * kChunk) {
        if (tid == 0) q_count = 0;
        __syncthreads();
        // ---- test phase
#pragma unroll 2
        for (int q = 0; q < kChunk / kThreads; ++q) {
            const int local = q * kThreads + tid;
            const int64_t i = c0 + local;
            bool push = false;
            if (i < n_points) {
                if (single_slab) {
                    push = true;
                } else {
                    const float4 x = x4[i];
                    uint32_t cx, cy, cz;
                    float wdummy;
                    pos_fract(x.x, scale, cx, wdummy);
                    pos_fract(x.y, scale, cy, wdummy);
                    pos_fract(x.z, scale, cz, wdummy);
                    if (fast_hash && (cx + 1u) < (1u << kSlabShift)) {
                        const uint32_t y0 = cy * LS2FM_PRIME_Y, y1 = (cy + 1u) * LS2FM_PRIME_Y;
                        const uint32_t z0 = cz * LS2FM_PRIME_Z, z1 = (cz + 1u) * LS2FM_PRIME_Z;
                        push = (((y0 ^ z0) & mask) >> kSlabShift) == slab || (((y1 ^ z0) & mask) >> kSlabShift) == slab ||
                               (((y0 ^ z1) & mask) >> kSlabShift) == slab || (((y1 ^ z1) & mask) >> kSlabShift) == slab;
                    } else if (!hashed && cx < res && cy < res && cz < res) {
                        const uint32_t first = cx + cy * res + cz * res * res;      // no wrap inside the cube:
                        push = first < hi && first + span >= lo;                   // conservative range overlap
                        if (first + span >= size) push = true;                     // wraps -> let process decide
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const uint32_t idx = level_index(cx + (k & 1), cy + ((k >> 1) & 1), cz + ((k >> 2) & 1), res,
                                                             size, mask, hashed && pow2, hashed);
                            push = push || (idx >= lo && idx < hi);
                        }
                    }
                }
            }
            const unsigned long long m = __ballot(push);
            if (m) {                                            // wave-uniform
                const int leader = __ffsll((long long)m) - 1;
                int base = 0;
                if (lane == leader) base = atomicAdd(&q_count, __popcll(m));
                base = __shfl(base, leader, 64);
                if (push) queue[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)local;
            }
        }
        __syncthreads();
        // ---- process phase.  Queue order == sample order, and neighbouring samples of a ray share their cell on
        // the coarse levels: processed in order, the 64 lanes of a wave would add to the SAME LDS word (64-way
        // serialised ds_add).  Walk the queue through a multiplicative permutation (bijective on a power of two).
        const int nq = q_count;
        int nq_pad = 1;
        while (nq_pad < nq) nq_pad <<= 1;
        for (int e0 = tid; e0 < nq_pad; e0 += kThreads) {
            const int e = (int)(((unsigned)e0 * 2053u) & (unsigned)(nq_pad - 1));
            if (e >= nq) continue;
            const int64_t i = c0 + queue[e];
            const float4 x = x4[i];
            uint32_t g[3];
            float w[3];
            pos_fract(x.x, scale, g[0], w[0]);
            pos_fract(x.y, scale, g[1], w[1]);
            pos_fract(x.z, scale, g[2], w[2]);
            float d0, d1, r0 = 0.f, r1 = 0.f, qd[3] = {0.f, 0.f, 0.f};
            if (SECOND_ORDER) {
                const float4 ra = *reinterpret_cast<const float4*>(rec_l + i * 8);
                const float4 rb = *reinterpret_cast<const float4*>(rec_l + i * 8 + 4);
                d0 = ra.x; d1 = ra.y; r0 = ra.z; r1 = ra.w;
                qd[0] = scale * rb.x; qd[1] = scale * rb.y; qd[2] = scale * rb.z;
            } else {
                const float2 ra = *reinterpret_cast<const float2*>(rec_l + i * 2);
                d0 = ra.x; d1 = ra.y;
            }
            // trilinear weight  W = px py pz ; directional derivative weight  D = qx py pz + px qy pz + px py qz
            // with p_a(b) = b ? w_a : 1 - w_a and q_a(b) = (b ? +1 : -1) * scale * gn_a
#pragma unroll
            for (int bz = 0; bz < 2; ++bz)
#pragma unroll
                for (int by = 0; by < 2; ++by) {
                    const float py = by ? w[1] : 1.0f - w[1], pz = bz ? w[2] : 1.0f - w[2];
                    const float pyz = py * pz;
                    float qyz = 0.f;
                    if (SECOND_ORDER) qyz = (by ? qd[1] : -qd[1]) * pz + py * (bz ? qd[2] : -qd[2]);
#pragma unroll
                    for (int bx = 0; bx < 2; ++bx) {
                        const uint32_t idx = level_index(g[0] + bx, g[1] + by, g[2] + bz, res, size, mask, hashed && pow2,
                                                         hashed);
                        if (idx >= lo && idx < hi) {
                            const float px = bx ? w[0] : 1.0f - w[0];
                            const float wt = px * pyz;
                            float v0 = wt * d0, v1 = wt * d1;
                            if (SECOND_ORDER) {
                                const float dirw = fmaf(bx ? qd[0] : -qd[0], pyz, px * qyz);
                                v0 = fmaf(dirw, r0, v0);
                                v1 = fmaf(dirw, r1, v1);
                            }
                            atomicAdd(&acc[2 * (idx - lo) + 0], v0);
                            atomicAdd(&acc[2 * (idx - lo) + 1], v1);
                        }
                    }
                }
        }
        __syncthreads();
    }
    // ---- flush
    float* dst = dtable + 2ull * (lv.offset[l] + lo);
    const int n_out = 2 * (int)(hi - lo);
    if (parts == 1) {
        for (int e = tid; e < n_out; e += kThreads) dst[e] = acc[e];          // sole owner of the slab
    } else {
        for (int e = tid; e < n_out; e += kThreads)                           // small coarse level, zeroed by the host
            if (acc[e] != 0.f) atomicAdd(dst + e, acc[e]);
    }
}

}  // namespace

// dtable is OVERWRITTEN over the whole grid.  x4: float4 (x, y, z, -) per point; rec: [level][point][8 | 2].
int ls2fm_launch_slab_scatter(const ls2fm_grid_desc* grid, const float* x4, int64_t n_points, int64_t p_pad,
                              const float* rec, bool second_order, float* dtable, hipStream_t stream) {
    SlabPlan plan{};
    for (int l = 0; l < LS2FM_MAX_LEVELS; ++l) plan.parts[l] = 1;
    int total = 0;
    const int64_t target = 16384;            // survivors a workgroup should process (what a hashed-level slab sees)
    for (int l = 0; l < LS2FM_MAX_LEVELS + 1; ++l) {
        plan.first[l] = total;
        if (l >= grid->n_levels) continue;
        const int slabs = (int)((grid->size[l] + kSlabEntries - 1) / kSlabEntries);
        int parts = 1;
        if (!grid->hashed[l] || slabs < 8) {
            // dense / tiny level: each of its slabs sees ~ P / slabs survivors (hashed levels: ~12 % of P)
            const int64_t per_block = n_points / slabs;
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
    if (second_order)
        slab_scatter_kernel<true><<<total, kThreads, 0, stream>>>(lv, plan, (const float4*)x4, n_points, p_pad, rec, dtable);
    else
        slab_scatter_kernel<false><<<total, kThreads, 0, stream>>>(lv, plan, (const float4*)x4, n_points, p_pad, rec, dtable);
    return ls2fm_launch_status();
}
