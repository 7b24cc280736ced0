// shade_bwd: backward of the per-ray compositing and of the per-sample fields, including the analytic double backward
// of the normal path, on the matrix cores.
//
//   1. one thread per sample: composite backward (prefix + suffix scans), sigma / beta, collapsed radiance decoder ->
//      per-sample upstream vectors dz (3), g_n (3), g_sdf, handed to the MFMA lanes through LDS
//   2. wave = 64 samples as 16-column tiles, two tiles per pass; everything TRANSPOSED [feature x sample] exactly as in
//      shade_fwd.hip, so accumulator tiles feed the next GEMM without staging:
//        A  = W0' U,  Q = W0' V                   V = d(inputs)/dp contracted with the normal's upstream (A.4)
//        T  = W1^T GF                              GF = upstream of the 17 MLP outputs
//        DA = S1 . T + S2 . w1_0 . Q,  GJ = S1 . w1_0         (elementwise on accumulator registers)
//        DE = W0'^T DA,  RR = W0'^T GJ            (encoding rows only: the payload of the table-gradient scatter)
//        second field: A2, T2 = W1g^T GF2, DA2 = S1 . T2, DE2 = W0g'^T DA2
//   3. (WG) the WEIGHT GRADIENTS of both Geometry MLPs, contracted where their operands already live: DA, GJ, H leave the
//      accumulator registers through a 3.8 KB wave-private LDS tile (the contraction index of dW is the SAMPLE, which sits
//      on the wrong side of the accumulator layout: written [feature][sample], read back as one ds_read_b128 per lane),
//      U, V, GF take the same trip once per tile, and
//        dW0' += DA U^T + GJ V^T  (64 x 36)   dW1[1..16] += GF H^T  (16 x 64)   dW1[0] += gf0 H + S1 . Q   db1 += sum GF
//      accumulate in 85 (68) registers per wave over the wave's 64 samples; the workgroup's waves are summed through LDS
//      and leave ONE partial per workgroup slot (fixed-order sums downstream: wgrad_l1 + wgrad_tail, deterministic).
//      No separate weight-gradient kernel re-derives the hidden layer (round 4: wgrad_mlp x 2, 95 us alone and 150 us beside
//      the table scatter), and the v / gf / gf2 rows (36 MB) are never stored.  One 16-column tile at a time then (NC = 1):
//      the accumulators take the registers the second tile's operands had.
//      Without WG (point queries keep wgrad_mlp.hip) the small per-sample upstream vectors (v, gf, gf2, dz, p) are stored.
// MFMA-ordered weights are staged into LDS once per workgroup.  fp32 MFMA: exact fp32 products / sums.
#include "bin_items.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Issue priority (round 6, profiles/r06_notes.md section 9).  The two waves of a SIMD belong to two rays that started together: left
// alone they reach their MFMA clusters and their softplus / LDS parts at the same time and queue for the same pipe.  With the
// priority raised around the MFMA clusters of the SDF field's hidden blocks and dropped for the VALU part between them, the wave
// that gets to its MFMAs first keeps the matrix pipe and the other one's VALU work fills the gaps -- the pair falls out of step:
// shade_bwd 111.4 -> 106.0 us at C2 (single field 72.8 -> 70.6, C3 832 -> 821), bit-identical.  LS2FM_BWD_PRIO: 0 = no s_setprio,
// 1 = raised for the whole hidden-block loop (107 us at C2, but +1 % at C3), 2 = around the MFMA clusters (the default).  Measured
// and not kept: a tile's operand loads at the highest priority (no change), the VALU position columns lowered (108), a constant
// priority by the parity of the wave's slot (111: no gain), the same toggling in the second field's loop (no change), and the same
// scheme in shade_fwd (four waves per SIMD there: 57 - 60 against 57.6 us); __builtin_amdgcn_iglp_opt(0 / 1) in the hidden-block loop (106).
// LS2FM_BWD_PART_NT (round 6): the per-ray weight-gradient partials (39 KB per ray) leave with non-temporal stores: shade_bwd 105.6 -> 103.6 us,
// the fill (whose leading rows read them) 79.9 -> 81.9, the step 0.4098 -> 0.4078 ms in three alternating pairs (profiles/r06_raw/c73_ab_part_nt.txt)
#ifndef LS2FM_BWD_PART_NT
#define LS2FM_BWD_PART_NT 1
#endif
#ifndef LS2FM_BWD_PRIO
#define LS2FM_BWD_PRIO 2
#endif
#define BWD_PRIO_LOOP(x) do { if (LS2FM_BWD_PRIO == 1) __builtin_amdgcn_s_setprio(x); } while (0)
#define BWD_PRIO_MFMA(x) do { if (LS2FM_BWD_PRIO == 2) __builtin_amdgcn_s_setprio(x); } while (0)
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

#ifndef LS2FM_PROBE
#define LS2FM_PROBE 0
#endif
constexpr bool kProbeNoContract = (LS2FM_PROBE & 1) != 0, kProbeNoTrans = (LS2FM_PROBE & 2) != 0, kProbeNC1 = (LS2FM_PROBE & 4) != 0;
constexpr bool kProbeNoPos = (LS2FM_PROBE & 8) != 0, kProbeStage1 = (LS2FM_PROBE & 32) != 0;
constexpr int kTLd = 20;       // floats per row of the wave-private transpose tile: 16 samples + 4 pad (16-B aligned rows)
constexpr int kTRows = 48;     // da | gj | h of one hidden block (U / V: 36 rows, GF: 20)
constexpr int kSwWg = kMfmaBwdSdfFloats - 4 * 4 * 64 + 64;       // WG: W1[0] as 64 floats instead of its operand-ordered 4 KB
// Round 6: the fused-weight-gradient form with MORE tiles in flight per wave and FEWER waves per SIMD -- the
// unified register file gives a wave 512 registers (256 + 256 accumulation registers) at one wave per SIMD.  LS2FM_BWD_NC_WG = 16-sample
// tiles a wave works on side by side (1, 2 or 4: every weight operand read from LDS then feeds that many independent MFMA chains),
// LS2FM_BWD_WAVES_WG = the occupancy the register budget is cut for.  The tiles are contracted into the weight-gradient accumulators
// in tile order whatever NC is: results are bit-identical across the variants.
#ifndef LS2FM_BWD_NC_WG
#define LS2FM_BWD_NC_WG 1
#endif
#ifndef LS2FM_BWD_WAVES_WG
#define LS2FM_BWD_WAVES_WG 2
#endif
template <bool WG> constexpr int bwd_nc() { return WG ? LS2FM_BWD_NC_WG : (kProbeNC1 ? 1 : 2); }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// the scatter's records: written once here, read once by scatter_fill (any XCD) -- optionally non-temporal (LS2FM_REC_NT)
#ifndef LS2FM_REC_NT
#define LS2FM_REC_NT 1
#endif
__device__ __forceinline__ void st4(float* p, const float4 v) {
#if LS2FM_REC_NT
    const f32x4 r = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(p));
#else
    *reinterpret_cast<float4*>(p) = v;
#endif
}
__device__ __forceinline__ void st2(float* p, const float2 v) {
#if LS2FM_REC_NT
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 r = {v.x, v.y};
    __builtin_nontemporal_store(r, reinterpret_cast<f32x2*>(p));
#else
    *reinterpret_cast<float2*>(p) = v;
#endif
}
// LS2FM_BWD_OFF32 (round 6, as in shade_fwd.hip): per-sample rows are addressed as uniform base + 32-bit BYTE offset -- the offset is
// formed in 32-bit arithmetic, so the access is `global_load v, v_off, s[base]` (one register, one add) instead of a 64-bit address per
// lane (an element index scaled in 64 bits defeats that form).  LS2FM_MAX_RENDER_POINTS = 2^23 keeps 32 rows x p_pad x 12 bytes and 16
// levels x p_pad x 16 bytes under 2^32.
#ifndef LS2FM_BWD_EJ_NT
#define LS2FM_BWD_EJ_NT 0
#endif
#ifndef LS2FM_BWD_OFF32
#define LS2FM_BWD_OFF32 1
#endif
template <typename T> __device__ __forceinline__ const T* at_bytes(const float* base, uint32_t bytes) {
    return reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + bytes);
}
template <typename T> __device__ __forceinline__ T* at_bytes(float* base, uint32_t bytes) {
    return reinterpret_cast<T*>(reinterpret_cast<char*>(base) + bytes);
}
// element `idx` of a float row block / the `idx`-th group of K floats
__device__ __forceinline__ const float* elem(const float* base, uint32_t idx, uint32_t k = 1u) {
    return LS2FM_BWD_OFF32 ? at_bytes<float>(base, idx * (4u * k)) : base + (int64_t)idx * k;
}
__device__ __forceinline__ float* elem(float* base, uint32_t idx, uint32_t k = 1u) {
    return LS2FM_BWD_OFF32 ? at_bytes<float>(base, idx * (4u * k)) : base + (int64_t)idx * k;
}
// lane J of this lane's 16-lane row (DPP row_share: folds into the consuming VALU instruction)
template <int J> __device__ __forceinline__ float row_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + J, 0xF, 0xF, false));
}
// sum over the 16 lanes of a row, in every lane (DPP row rotations: no trip through the LDS crossbar as __shfl_xor takes)
template <int N> __device__ __forceinline__ float row_ror(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float row_sum16(float v) {
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    return v;
}
// the four position / bias columns of dW0' (k' = 32 .. 35) on the VALU, in the TRANSPOSED layout the contraction's operands have
// (lane: hidden unit jl, samples 4g .. 4g + 3): as MFMA tiles they were eight (four) products per hidden block for 4 live columns
// of 16.  u2 / v2: lanes jl < 4 of a row hold U / V rows 32 + jl at the row's four samples.
// s += x * (lane A of this lane's row).u  as ONE instruction (v_fmac_f32 with a DPP row_share source).  Written as asm on
// purpose: as a builtin broadcast + fmaf the compiler hoists the 28 broadcasts of a tile out of the hidden-block loop and keeps
// them in 28 registers -- which the kernel does not have (scratch spills inside the MFMA loop: 105 -> 183 us)
template <int A> __device__ __forceinline__ void fmac_row(float& s, float u, float x) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(s) : "v"(u), "v"(x), "n"(A));
}
template <int A, bool WITH_V>
__device__ __forceinline__ float pos_col(float s, const float4& a_da, const float4& a_gj, const float4& u2, const float4& v2) {
    fmac_row<A>(s, u2.x, a_da.x); fmac_row<A>(s, u2.y, a_da.y); fmac_row<A>(s, u2.z, a_da.z); fmac_row<A>(s, u2.w, a_da.w);
    if (WITH_V) { fmac_row<A>(s, v2.x, a_gj.x); fmac_row<A>(s, v2.y, a_gj.y); fmac_row<A>(s, v2.z, a_gj.z); fmac_row<A>(s, v2.w, a_gj.w); }
    return s;
}

// fp64 wave scans for the d beta path (below)
__device__ __forceinline__ double wave_scan_incl_f64(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ double wave_suffix_excl_f64(double v, int lane) {
    double s = __shfl_down(v, 1, 64);
    if (lane == 63) s = 0.0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_down(s, o, 64);
        if (lane + o < 64) s += t;
    }
    return s;
}

// (WG) end of a pass: this wave's accumulators in the partial's row order (wgrad_tail.h: reduce_partials_row), with the row sums
// over the 16 sample lanes for dW1[0] / db1; the workgroup's waves are summed through `buf` (s_w: the pass is done with its
// weights) in a fixed order, and wave 0 stores the ray's partial
template <bool GEO>
__device__ __forceinline__ void wg_flush(const f32x4 (&acc0)[4][2], const f32x4 (&acc1)[4], const float (&pacc)[4][4],
                                         float w1p, float gs16, float g0s, float* __restrict__ buf,
                                         float* __restrict__ dst, int wave, int n_waves, int lane) {
    constexpr int R = GEO ? kRegsGeo : kRegsSdf;
    const int jl = lane & 15, g = lane >> 4;
    float regs[R];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int mk = 0; mk < 2; ++mk)
#pragma unroll
            for (int q = 0; q < 4; ++q) regs[(m * 3 + mk) * 4 + q] = acc0[m][mk][q];
        // position / bias columns: pacc[m][a] = this lane's sample group's part of dW0'[16m + jl][32 + a]; summed over the four
        // groups, then moved into the tile layout of the partial (lane (g, jl < 4), row q: dW0'[16m + 4g + q][32 + jl])
        float tot[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float v = pacc[m][a];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            tot[a] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int src = 16 * g + 4 * g + q;
            const float t0 = __shfl(tot[0], src, 64), t1 = __shfl(tot[1], src, 64), t2 = __shfl(tot[2], src, 64), t3 = __shfl(tot[3], src, 64);
            regs[(m * 3 + 2) * 4 + q] = jl == 0 ? t0 : (jl == 1 ? t1 : (jl == 2 ? t2 : t3));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) regs[48 + m * 4 + q] = acc1[m][q];
        if (!GEO) {          // dW1[0][16m + 4g + q]: the row sums are packed one per lane, jl = 4m + q
#pragma unroll
            for (int q = 0; q < 4; ++q) regs[64 + m * 4 + q] = __shfl(w1p, 16 * g + 4 * m + q, 64);
        }
    }
    {   // db1: gs16 = this lane's sample group's part of sum_s GF[o = 1 + jl][s]; g0s = this lane's samples' gf[0] (every group
        // holds a copy: group 0's is taken)
        float v = gs16;
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        float v0 = g0s;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) v0 += __shfl_xor(v0, o, 64);
        v0 = __shfl(v0, 0, 64);
#pragma unroll
        for (int t = 0; t < (GEO ? 4 : 5); ++t) {
            const int o = GEO ? 1 + 4 * t + g : 4 * t + g;                  // the row this lane's group reports (wgrad_tail.h)
            const float x = __shfl(v, 16 * g + ((o - 1) & 15), 64);
            regs[(GEO ? 64 : 80) + t] = o == 0 ? v0 : x;
        }
    }
    for (int wv = n_waves - 1; wv >= 1; --wv) {
        __syncthreads();                     // (first trip: every wave is done with the pass's weights)
        if (wave == wv) {
#pragma unroll
            for (int q = 0; q < R; ++q) buf[q * 64 + lane] = regs[q];
        }
        __syncthreads();
        if (wave == wv - 1) {
#pragma unroll
            for (int q = 0; q < R; ++q) regs[q] += buf[q * 64 + lane];
        }
    }
    if (wave == 0) {
#pragma unroll
#if LS2FM_BWD_PART_NT
        for (int q = 0; q < R; ++q) __builtin_nontemporal_store(regs[q], LS2FM_BWD_OFF32 ? at_bytes<float>(dst, (uint32_t)lane * 4u + (uint32_t)q * 256u) : dst + lane + q * 64);
#else
        for (int q = 0; q < R; ++q) (dst + lane)[q * 64] = regs[q];
#endif
    }
}

// POSE: center / ray require gradients (BA "sfm_refine" with one camera, BA.py:153-154) -- the only case that needs the rows of
// W0'^T DA that belong to the MLP's position inputs: 16 more MFMAs per hidden block and 8 more registers.
// WG: weight gradients contracted here (header, 3.), one partial per ray.  MAXT = 128: rays of up to 128 samples, 4 workgroups
// of 2 waves per CU.  (A workgroup walking several rays and keeping one partial was tried first: the loop-invariant kernel
// arguments it keeps in scalar registers across the loop spill into vector registers -- 92 .. 136 B of scratch per lane.)
// (Rounds 5-6: this kernel had two run-time modes on MI355X -- 121-126 and 132-137 us at C2, 85 / 97 us single field -- set by the NUMA
// node the host process initialised on.  Cause, found in round 6 (profiles/r06_notes.md section 6): its waves read the DISPATCH PACKET,
// which lives in host memory -- see n_threads / dz_g below.  Without those reads: 115 us (74.5 single field) from either socket.)
// FULL: all 32 encoding channels of both grids are live (16-level grids: every shipped preset) -- known at compile time, the sixteen
// per-channel predicates of a tile's operand loads, their exec-mask branches and the zero fills behind them are gone (round 6:
// shade_bwd 115.0 -> 111.8 us, single field 74.8 -> 72.2; bit-identical).  Instantiated for the fused-weight-gradient form only.
template <bool DUAL, int MAXT, bool POSE, bool WG, bool FULL = false>
__global__ void __launch_bounds__(MAXT, WG ? LS2FM_BWD_WAVES_WG : 2)
shade_bwd_kernel(FieldC fc, LevelScales lsc, int ch1_arg, int ch2_arg, WsLayout w, const Packed* __restrict__ pk,
                 const float* __restrict__ center, const float* __restrict__ ray, const float* __restrict__ fws,
                 Upstream up, float* __restrict__ out, ZeroJob zero, int64_t n_rays, float* __restrict__ slot_sdf,
                 float* __restrict__ slot_geo) {
    constexpr bool want_pose = POSE;
    constexpr int NC = bwd_nc<WG>();                            // 16-sample column tiles in flight per wave (a wave owns four)
    constexpr int kUnrollM = WG ? 4 : 1;         // the accumulators of a hidden block are registers: its loop is unrolled
    // leading workgroups: the zero fills the rest of the backward needs (weight-gradient accumulators, point-split coarse
    // levels of the gradient tables) -- no memset / kernel launches and no cross-stream edge in front of the scatter
    // The workgroup size is DERIVED (the launcher's formula), never read as blockDim.x.  Round 6: in the fused-weight-gradient
    // variants the compiler fetched blockDim.x from the DISPATCH PACKET (`.amdhsa_user_sgpr_dispatch_ptr 1`; the other variants take
    // it from the hidden kernel arguments) -- the AQL queue lives in HOST memory: ~290 uncached 32-byte reads over PCIe per launch
    // (TCC_EA0_RDREQ_IO_32B; every other kernel of the step: 0), one on every workgroup's critical path at the head of its MFMA part.
    // That is what made the kernel 123 us from the GPU's NUMA node and 135 us from the other socket (profiles/r06_notes.md section 6).
    const int ch1 = FULL ? 32 : ch1_arg, ch2 = FULL ? 32 : ch2_arg;
    const int n_threads = (fc.n_samples + 63) / 64 * 64;
    if ((int)blockIdx.x < zero.blocks) {
        zero_job_run(zero, (int)blockIdx.x, (int)threadIdx.x, n_threads);
        return;
    }
    __shared__ float s_part[MAXT / 64][8];
    __shared__ double s_db[MAXT / 64];
    __shared__ double s_pd[MAXT / 64][2];        // fp64 twin of the composite scans (d beta): tau totals, sum of U w
    __shared__ int s_bound[32];                  // per-level max of a single scatter contribution (bits of a float >= 0)
    __shared__ float s_y[MAXT][8];               // per sample: dz(3), g_n(3), g_sdf
    __shared__ float s_wc[3][68];                // collapsed decoder (Packed::wc)
    __shared__ float s_w[WG ? kSwWg : kMfmaBwdSdfFloats];     // MFMA-ordered weights of the field being processed (26 KB; WG:
                                                 // 22 KB, W1[0] compact); WG: also the buffer of the cross-wave sum of the
                                                 // weight-gradient registers
    __shared__ __attribute__((aligned(16))) float s_t[WG ? MAXT / 64 : 1][WG ? NC * kTRows * kTLd : 4];   // wave-private transpose tiles
    static_assert(kRegsSdf * 64 <= kSwWg && kRegsGeo * 64 <= kSwWg && kMfmaBwdGeoFloats <= kSwWg, "register sums / second field's weights fit in s_w");
    const int N = fc.n_samples;
    const int n = threadIdx.x, lane = n & 63, wave = n >> 6, n_waves = n_threads >> 6;
    const int64_t r = (int64_t)blockIdx.x - zero.blocks;
    const int jl = lane & 15, g = lane >> 4;
    const int64_t P = w.p_pad;
    const uint32_t P32 = (uint32_t)w.p_pad;
    const float* __restrict__ f_e1 = fws + w.e1;
    const float* __restrict__ f_j1 = fws + w.j1;
    const float* __restrict__ f_e2 = fws + w.e2;
    float* __restrict__ o_v = out + w.v;
    float* __restrict__ o_gf = out + w.gf;
    float* __restrict__ o_gf2 = out + w.gf2;
    const RayGeom gm = load_ray(fc, center, ray, r);
    {   // stage the SDF field's operand-ordered weights (consumed after the barriers of part 1)
        const float4* src = reinterpret_cast<const float4*>(&pk->bs);
        float4* dst = reinterpret_cast<float4*>(s_w);
        constexpr int n_stage = WG ? kMfmaBwdSdfFloats - 4 * 4 * 64 : kMfmaBwdSdfFloats;       // (w10 is the last member)
        for (int q = n; q < n_stage / 4; q += n_threads) dst[q] = src[q];
        if (WG && n < 64) s_w[n_stage + n] = pk->bs.w10[n >> 4][n & 3][16 * ((n >> 2) & 3)];      // W1[0][hidden unit n]
    }
    if (n < 32) s_bound[n] = 0;
    // the decoder's feature columns, read per lane (o = 4t + g) in every tile: from LDS (one base register + immediate offsets;
    // as global loads the compiler kept fifteen 64-bit addresses alive across the tile loop -- and spilled them)
    for (int q = n; q < 3 * 68; q += n_threads) (&s_wc[0][0])[q] = (&pk->wc[0][0])[q];

    // =========================================================================== 1. one thread per sample
    {
        const bool live = n < N;
        const int nn = live ? n : N - 1;
        const int64_t i = r * N + nn;
        const float t = sample_depth(gm, nn, N);
        const float t_next = sample_depth(gm, nn + 1, N);
        const float t_last = sample_depth(gm, N - 1, N);
        const float ray_len = sqrtf(gm.d[0] * gm.d[0] + gm.d[1] * gm.d[1] + gm.d[2] * gm.d[2]);
        float g_rgb[3], g_nm[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            g_rgb[c] = up.d_rgb ? up.d_rgb[r * 3 + c] : 0.f;
            g_nm[c] = up.d_nm ? up.d_nm[r * 3 + c] : 0.f;
        }
        float g_dep = up.d_depth ? up.d_depth[r] : 0.f;
        // fused loss head (ls2fm_loss_spec): the upstream of rgb / depth / normals is formed here from the counts of the
        // forward's reduction, the weights and the scalar upstreams -- exactly loss_head_bwd_kernel's arithmetic
        float gl_eik = 0.f;
        if (up.loss.rgb_gt != nullptr) {
            const LossUp& lo = up.loss;
            float gt[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            if (lo.d_terms) {
#pragma unroll
                for (int k = 0; k < 5; ++k) gt[k] = lo.d_terms[k];
            }
            const float g_all = gt[4] + (lo.d_total ? lo.d_total[0] : 0.f);
            const float gl_rgb = fmaf(lo.weights[0], g_all, gt[0]) / (float)lo.sums[1];
            const float gl_dc = lo.sums[5] > 0.0 ? fmaf(lo.weights[2], g_all, gt[2]) / (float)lo.sums[5] : 0.f;
            const float gl_mse = gt[3] / (float)lo.sums[7];
            if (ls2fm_in_eik(lo, r)) gl_eik = fmaf(lo.weights[1], g_all, gt[1]) / (float)lo.sums[3];
            const float* __restrict__ rout = fws + w.rout;
            const bool in_mse = ls2fm_in_mse(lo, r);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = rout[c * w.r_pad + r] - lo.rgb_gt[r * 3 + c];
                float v = gl_rgb * ls2fm_sign(d);
                if (in_mse) v = fmaf(gl_mse * 2.0f, d, v);
                g_rgb[c] += v;
            }
            float gd = 0.f;
            if (lo.depth_ref != nullptr && (lo.mask_dc == nullptr || lo.mask_dc[r] != 0))
                gd = gl_dc * ls2fm_smooth_l1_grad(lo.depth_ref[r] - rout[3 * w.r_pad + r]);
            g_dep -= gd;
            if (n == 0 && lo.d_depth_ref != nullptr) lo.d_depth_ref[r] = gd;
        }
        // forward per-sample values saved by shade_fwd
        const float sdf = fws[w.sdfv + i];
        float nrm[3], col[3], n_last[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            nrm[a] = fws[w.nrm + a * P + i];
            col[a] = fws[w.rgbs + a * P + i];
            n_last[a] = fws[w.nrm + a * P + r * N + N - 1];
        }
        const float alpha = pk->alpha, beta = pk->beta;
        const float lap = 0.5f * expf(-fabsf(sdf) / beta);
        const float sigma = alpha * (sdf >= 0.f ? lap : 1.0f - lap);
        // composite forward quantities (same scans as shade_fwd)
        const bool interval = n < N - 1;
        const float delta = (t_next - t) * ray_len;
        const float tau = interval ? sigma * delta : 0.f;
        const float incl = wave_scan_incl(tau, lane);
        if (lane == 63) s_part[wave][0] = incl;
        // d L / d beta is ONE scalar, and the composite backward forms each summand's d L / d tau_k = U_k T_k e^{-tau_k} -
        // sum_{i>k} U_i w_i as a difference of nearly equal quantities: in fp32 (the reference's autograd included) the result
        // carries errors of 1e-4 .. 1e-3 even when the final sum is well conditioned.  Its whole chain -- sigma, the two
        // scans, transmittance, d tau, d sigma / d beta -- is therefore evaluated a second time in fp64 from the same fp32
        // per-sample values (sdf, colour, normal, depth): ~300 fp64 operations per sample, invisible next to the MFMA part,
        // and d beta comes out as the exactly summed value of the fp32 computation (oracle.fields.beta_gradient_exact_sum).
        const double beta_d = (double)beta, alpha_d = (double)alpha, sdf_d = (double)sdf;
        const double lap_d = 0.5 * exp(-fabs(sdf_d) / beta_d);
        const double sigma_d = alpha_d * (sdf_d >= 0.0 ? lap_d : 1.0 - lap_d);
        const double delta_d = ((double)t_next - (double)t) *
                               sqrt((double)gm.d[0] * (double)gm.d[0] + (double)gm.d[1] * (double)gm.d[1] + (double)gm.d[2] * (double)gm.d[2]);
        const double tau_d = interval ? sigma_d * delta_d : 0.0;
        const double incl_d = wave_scan_incl_f64(tau_d, lane);
        if (lane == 63) s_pd[wave][0] = incl_d;
        __syncthreads();
        float before = incl - tau;
        for (int q = 0; q < wave; ++q) before += s_part[q][0];
        const float trans = expf(-before), ex = expf(-tau);
        const float wgt = interval ? trans * (1.0f - ex) : 0.f;
        double before_d = incl_d - tau_d;
        for (int q = 0; q < wave; ++q) before_d += s_pd[q][0];
        const double trans_d = exp(-before_d), ex_d = exp(-tau_d);
        const double wgt_d = interval ? trans_d * (1.0 - ex_d) : 0.0;
        // composite backward:  L = sum_i w_i (V_i - B) + B ;  dL/dtau_k = U_k T_k e^{-tau_k} - sum_{i>k} U_i w_i
        float b_term = g_dep * t_last, v_term = g_dep * t;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            b_term = fmaf(g_rgb[c], fc.bg[c], b_term);
            b_term = fmaf(g_nm[c], n_last[c], b_term);
            v_term = fmaf(g_rgb[c], col[c], v_term);
            v_term = fmaf(g_nm[c], nrm[c], v_term);
        }
        const float u_w = interval ? (v_term - b_term) * wgt : 0.f;
        // sum_{i>k} U_i w_i as a true suffix sum (not total - prefix: that difference loses the small late-ray values to
        // the rounding of the ray total, coherently over the samples of a ray, which d beta's cancelling sum then exposes)
        const float sfx_w = wave_suffix_excl(u_w, lane);
        const float wsum_uw = wave_sum(u_w);
        const float wsum_w = wave_sum(wgt);
        if (lane == 0) { s_part[wave][1] = wsum_uw; s_part[wave][2] = wsum_w; }
        double vb_d = (double)g_dep * ((double)t - (double)t_last);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            vb_d += (double)g_rgb[c] * ((double)col[c] - (double)fc.bg[c]);
            vb_d += (double)g_nm[c] * ((double)nrm[c] - (double)n_last[c]);
        }
        const double u_w_d = interval ? vb_d * wgt_d : 0.0;
        const double sfx_d = wave_suffix_excl_f64(u_w_d, lane);
        {
            double tot = u_w_d;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o, 64);
            if (lane == 0) s_pd[wave][1] = tot;
        }
        __syncthreads();
        float opacity = 0.f, suffix = sfx_w;
        double suffix_d = sfx_d;
        for (int q = 0; q < n_waves; ++q) {
            opacity += s_part[q][2];
            if (q > wave) { suffix += s_part[q][1]; suffix_d += s_pd[q][1]; }
        }
        const float d_tau = interval ? (v_term - b_term) * trans * ex - suffix : 0.f;
        const float g_sigma = d_tau * delta;
        const double g_sigma_d = interval ? (vb_d * trans_d * ex_d - suffix_d) * delta_d : 0.0;
        const float rest = 1.0f - opacity;
        {   // d L / d |ray| : delta = (t_next - t) |ray| (Renderer.py:36-38); only consumed by the pose gradients
            const float s = wave_sum(d_tau * sigma * (t_next - t));
            if (lane == 0) s_part[wave][3] = s;
        }
        // per-sample upstream of colour, normal, sdf
        float gc[3], gn[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            gc[a] = wgt * g_rgb[a];
            float v = wgt * g_nm[a];
            if (n == N - 1) v += rest * g_nm[a];
            if (live && up.d_normals) v += up.d_normals[i * 3 + a];
            gn[a] = live ? v : 0.f;
        }
        if (gl_eik != 0.f && live) {       // eikonal term: d | |n| - 1 | / d n   (d |n| at 0 := 0, as torch)
            const float len = sqrtf(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2]);
            const float k = len > 0.f ? gl_eik * ls2fm_sign(len - 1.0f) / len : 0.f;
#pragma unroll
            for (int a = 0; a < 3; ++a) gn[a] = fmaf(k, nrm[a], gn[a]);
        }
        float g_sdf = (live && up.d_sdfs) ? up.d_sdfs[i] : 0.f;
        {   // sigma backward (+ d beta)
            const float abs_s = fabsf(sdf);
            const float dsig_ds = sdf != 0.f ? -alpha * lap / beta : 0.f;
            g_sdf = fmaf(g_sigma, dsig_ds, g_sdf);
            const float inv_b2 = 1.0f / (beta * beta);
            const float dsig_db = sdf >= 0.f ? lap * (abs_s * inv_b2 / beta - inv_b2)
                                             : -(1.0f - lap) * inv_b2 - lap * abs_s * inv_b2 / beta;
            (void)dsig_db;
            // d sigma / d beta and the sum over the samples in fp64 (see above)
            const double abs_d = fabs(sdf_d), inv_b2_d = 1.0 / (beta_d * beta_d);
            const double dsig_db_d = sdf_d >= 0.0 ? lap_d * (abs_d * inv_b2_d / beta_d - inv_b2_d)
                                                  : -(1.0 - lap_d) * inv_b2_d - lap_d * abs_d * inv_b2_d / beta_d;
            double db = g_sigma_d * dsig_db_d;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) db += __shfl_xor(db, o, 64);
            if (lane == 0) s_db[wave] = db;
        }
        // collapsed radiance decoder backward
        float dz[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) dz[c] = gc[c] * col[c] * (1.0f - col[c]);
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) gn[a] = fmaf(pk->wc[c][3 + a], dz[c], gn[a]);
        // per-ray sums of dz (view-embedding columns of the decoder) and the d beta total
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s = wave_sum(dz[c]);
            if (lane == 0) s_part[wave][4 + c] = s;
        }
        float* y = s_y[n];
        y[0] = dz[0]; y[1] = dz[1]; y[2] = dz[2];
        y[3] = gn[0]; y[4] = gn[1]; y[5] = gn[2];
        y[6] = g_sdf;
        __syncthreads();          // s_part, s_db, s_y, s_w, s_bound ready
        if (n < 3) {
            float s = 0.f;
            for (int q = 0; q < n_waves; ++q) s += s_part[q][4 + n];
            out[w.dzr + n * w.r_pad + r] = s;
        }
        if (n == 3) {
            double s = 0.0;
            for (int q = 0; q < n_waves; ++q) s += s_db[q];
            reinterpret_cast<double*>(out + w.dbeta)[r] = s;         // per-ray partial: no atomic, nothing to zero
        }
        if (n < kView) out[w.renc + n * w.r_pad + r] = view_component(gm.d, n);
        if (n == 4 && want_pose) {
            float s = 0.f;
            for (int q = 0; q < n_waves; ++q) s += s_part[q][3];
            out[w.dlen + r] = s;
        }
    }

    // =========================================================================== 2. MFMA lanes
    if (kProbeStage1) return;
    const float* __restrict__ s_w0a = s_w;                              // [m][t][lane]
    const float* __restrict__ s_w1ta = s_w + 4 * 9 * 64;                // [m][5][lane]
    const float* __restrict__ s_w0ta = s_w1ta + 4 * 5 * 64;             // [mk][m][r][lane]
    const float* __restrict__ s_w10 = s_w0ta + 2 * 4 * 4 * 64;          // [m][r][lane]; WG: [hidden unit]
    // pass 0: the SDF field for both halves of the wave's samples; pass 1 (dual): the second field for both halves -- each
    // field's weights are staged into LDS once per workgroup
#pragma unroll 1
    for (int pass = 0; pass < (DUAL ? 2 : 1); ++pass) {
    if (pass == 1) {
        __syncthreads();          // every wave is done with the SDF weights
        const float4* src = reinterpret_cast<const float4*>(&pk->bg);
        float4* dst = reinterpret_cast<float4*>(s_w);
        for (int q = n; q < kMfmaBwdGeoFloats / 4; q += n_threads) dst[q] = src[q];
        __syncthreads();          // second field's weights staged
    }
    // this lane's levels are the same in every half and column (l = 8 mk + 2 g + hv): keep the running maxima of the
    // contribution bounds in registers and publish them once per pass (a per-item LDS atomicMax: 16 lanes per address)
    float bnd[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    // (WG) this wave's weight-gradient accumulators of the field of this pass, over its 64 samples
    // (the row sums dW1[0] and db1 are packed: one value per lane, see wg_flush)
    f32x4 acc0[WG ? 4 : 1][WG ? 2 : 1], acc1[WG ? 4 : 1];
    float pacc[WG ? 4 : 1][4], w1p = 0.f, gs16 = 0.f, g0s = 0.f;
    if (WG) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc1[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mk = 0; mk < 2; ++mk) acc0[m][mk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 4; ++a) pacc[m][a] = 0.f;
        }
    }
    float* __restrict__ xt = s_t[WG ? wave : 0];
#pragma unroll 1
    for (int half = 0; half < 4 / NC; ++half) {
        uint32_t is[NC];                  // point index (32-bit offsets: uniform base + VGPR offset addressing)
        bool live_c[NC];
        float pw[NC][3], xg[NC][3], dz[NC][3], gns[NC][3], gnk[NC][3], gsdf[NC];
        // this lane GROUP's component (g < 3) of dz and of kappa g_n, picked where the values still sit in LDS.  (Picked later from the
        // register arrays -- `g == 0 ? dz[cc][0] : ...` -- the compiler folded the selects into ONE dynamically addressed load of a
        // private array, promoted that array to LDS indexed by the flat work-item id, and read the workgroup's sizes for it from the
        // dispatch packet in HOST memory: the kernel's NUMA dependence, n_threads above.)
        float dz_g[NC], gnk_g[NC];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const int sl = 64 * wave + 16 * (NC * half + cc) + jl;       // sample slot in the workgroup
            live_c[cc] = sl < N;
            const int nn = live_c[cc] ? sl : N - 1;
            is[cc] = (uint32_t)(r * N + nn);
            sample_position(fc, gm, sample_depth(gm, nn, N), pw[cc], xg[cc]);
            const float* y = s_y[sl];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                dz[cc][a] = y[a];
                gnk[cc][a] = fc.kappa * y[3 + a];
                gns[cc][a] = gnk[cc][a] * fc.inv_ext[a];
            }
            gsdf[cc] = y[6];
            dz_g[cc] = y[g < 3 ? g : 0];
            gnk_g[cc] = fc.kappa * y[3 + (g < 3 ? g : 0)];
        }
        float ub[9][NC];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const float pg = g == 0 ? pw[cc][0] : (g == 1 ? pw[cc][1] : pw[cc][2]);
            ub[8][cc] = g < 3 ? pg / fc.rescale : 1.0f;
        }
        f32x4 dex[NC];
        if (pass == 0) {
        // ---- B operands: u, v (rows k' = 4t + g) and the MLP-output upstream gf (rows o = 4t + g)
        float vb[9][NC], gfb[5][NC];
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) {
                const int ch = 4 * t + g;
                const bool on = ch < ch1;
                ub[t][cc] = on ? (LS2FM_BWD_EJ_NT ? __builtin_nontemporal_load(elem(f_e1, (uint32_t)ch * P32 + is[cc])) : *elem(f_e1, (uint32_t)ch * P32 + is[cc])) : 0.f;
                // J rows are [channel][point][3]: one 12-byte load per (channel, sample)
                // (a 4-byte-aligned 3-vector, NOT a struct of float[3]: the conditionally assigned struct stayed an alloca, which the
                // compiler promoted to LDS indexed by the FLAT work-item id -- and for that it read the workgroup's y / z sizes from the
                // dispatch packet, i.e. from host memory, on every workgroup's critical path: round 6, n_threads above)
                typedef float f32x3 __attribute__((ext_vector_type(3)));
                typedef f32x3 f32x3_a4 __attribute__((aligned(4)));
                f32x3 jv = {0.f, 0.f, 0.f};
                if (on) jv = LS2FM_BWD_EJ_NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x3_a4*>(elem(f_j1, (uint32_t)ch * P32 + is[cc], 3u)))
                                             : *reinterpret_cast<const f32x3_a4*>(elem(f_j1, (uint32_t)ch * P32 + is[cc], 3u));
                float acc = fmaf(jv.x, gns[cc][0], 0.f);
                acc = fmaf(jv.y, gns[cc][1], acc);
                acc = fmaf(jv.z, gns[cc][2], acc);
                vb[t][cc] = acc;
            }
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const float kg = gnk_g[cc];
            vb[8][cc] = g < 3 ? kg / fc.rescale : 0.f;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const int o = 4 * t + g;                                  // gf[0] = kappa g_sdf ; gf[1 + m] = Wc[:, 33 + m]^T dz
                float v = 0.f;
                if (o == 0) v = fc.kappa * gsdf[cc];
                else if (o < kOut) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) v = fmaf(s_wc[k][32 + o], dz[cc][k], v);
                }
                gfb[t][cc] = v;
            }
        }
        // per-sample operands of the weight-gradient GEMMs that exist only in this layout
        float g1[NC];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) g1[cc] = fabsf(gns[cc][0]) + fabsf(gns[cc][1]) + fabsf(gns[cc][2]);
#pragma unroll
        for (int cc = 0; cc < NC; ++cc)
            if (live_c[cc]) {
                if (!WG) {
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int kp = 4 * t + g;                             // k' -> the reference's input column
                        if (kp < 35) *elem(o_v, (uint32_t)(kp < 32 ? 3 + kp : kp - 32) * P32 + is[cc]) = vb[t][cc];
                    }
#pragma unroll
                    for (int t = 0; t < 5; ++t)
                        if (4 * t + g < kOut) *elem(o_gf, (uint32_t)(4 * t + g) * P32 + is[cc]) = gfb[t][cc];
                }
                if (g < 3) {
                    const float pg = g == 0 ? pw[cc][0] : (g == 1 ? pw[cc][1] : pw[cc][2]);
                    *elem(out + w.p3, (uint32_t)g * P32 + is[cc]) = pg;
                    *elem(out + w.dz, (uint32_t)g * P32 + is[cc]) = dz_g[cc];
                }
            }

        // per-point part of the scatter payload (position + the normal's upstream), shared by every level and both grids
#pragma unroll
        for (int cc = 0; cc < NC; ++cc)
            if (live_c[cc] && g == 0) {
                float* dst = elem(out + w.rpt, is[cc], 8u);
                st4(dst, make_float4(xg[cc][0], xg[cc][1], xg[cc][2], gns[cc][0]));
                st4(dst + 4, make_float4(gns[cc][1], gns[cc][2], 0.f, 0.f));
            }

        // (WG) U, V, GF of this tile as operands of the contraction over the samples: row jl (+ 16 mk), samples 4g .. 4g + 3.
        // A wave's DS operations execute in order: a tile's rows are rewritten right behind the reads of the previous rows.
        float4 b_u[NC][3], b_v[NC][3], a_gf[NC];
        if (WG && kProbeNoTrans) {
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) b_u[cc][0] = b_u[cc][1] = b_u[cc][2] = b_v[cc][0] = b_v[cc][1] = b_v[cc][2] = a_gf[cc] = make_float4(1.f, 2.f, 3.f, 4.f);
        }
        if (WG && !kProbeNoTrans) {
            const int row2 = jl < 4 ? 32 + jl : 35;                  // rows 32..35 exist: lanes jl < 4 are the ones pos_col reads
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) {                        // (tile cc's own block of the wave's LDS area)
                float* __restrict__ xc = xt + cc * (kTRows * kTLd);
#pragma unroll
                for (int t = 0; t < 9; ++t) xc[(4 * t + g) * kTLd + jl] = ub[t][cc];
                __builtin_amdgcn_wave_barrier();
                b_u[cc][0] = ld4(xc + jl * kTLd + 4 * g); b_u[cc][1] = ld4(xc + (16 + jl) * kTLd + 4 * g); b_u[cc][2] = ld4(xc + row2 * kTLd + 4 * g);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 9; ++t) xc[(4 * t + g) * kTLd + jl] = vb[t][cc];
                __builtin_amdgcn_wave_barrier();
                b_v[cc][0] = ld4(xc + jl * kTLd + 4 * g); b_v[cc][1] = ld4(xc + (16 + jl) * kTLd + 4 * g); b_v[cc][2] = ld4(xc + row2 * kTLd + 4 * g);
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int t = 0; t < 5; ++t) xc[(4 * t + g) * kTLd + jl] = gfb[t][cc];
                __builtin_amdgcn_wave_barrier();
                a_gf[cc] = ld4(xc + (1 + jl) * kTLd + 4 * g);
                __builtin_amdgcn_wave_barrier();
                gs16 += (a_gf[cc].x + a_gf[cc].y) + (a_gf[cc].z + a_gf[cc].w);      // db1[1 + jl], this group's four samples
                g0s += fc.kappa * gsdf[cc];                                          // db1[0]
            }
        }
        float gf0[NC];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) gf0[cc] = WG ? fc.kappa * gsdf[cc] : 0.f;
        // ---- SDF field
        f32x4 de[2][NC], rr[2][NC];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) dex[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mk = 0; mk < 2; ++mk)
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) { de[mk][cc] = f32x4{0.f, 0.f, 0.f, 0.f}; rr[mk][cc] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        BWD_PRIO_LOOP(1);
#pragma unroll kUnrollM
        for (int m = 0; m < 4; ++m) {
            BWD_PRIO_MFMA(1);
            f32x4 aa[NC], qq[NC], tt[NC];
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) { aa[cc] = f32x4{0.f, 0.f, 0.f, 0.f}; qq[cc] = aa[cc]; tt[cc] = aa[cc]; }
            // (three independent accumulator chains side by side: a dependent f32 MFMA issues after 40 cycles, an independent
            // one after 32)
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float a = s_w0a[(m * 9 + t) * 64 + lane];
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    aa[cc] = mfma4(a, ub[t][cc], aa[cc]);
                    qq[cc] = mfma4(a, vb[t][cc], qq[cc]);
                }
                if (t < 5) {
                    const float a1 = s_w1ta[(m * 5 + t) * 64 + lane];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) tt[cc] = mfma4(a1, gfb[t][cc], tt[cc]);
                }
            }
            BWD_PRIO_MFMA(0);
            float da[NC][4], gj[NC][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float w10 = WG ? s_w10[16 * m + 4 * g + q] : s_w10[(m * 4 + q) * 64 + lane];
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    float h, s1, s2;
                    softplus100(aa[cc][q], h, s1, s2);
                    da[cc][q] = fmaf(s1, tt[cc][q], s2 * w10 * qq[cc][q]);
                    gj[cc][q] = s1 * w10;
                    if (WG && !kProbeNoTrans) {       // this hidden block's DA | GJ | H, [hidden unit 4g + q][sample jl]
                        float* __restrict__ xc = xt + cc * (kTRows * kTLd);
                        xc[(4 * g + q) * kTLd + jl] = da[cc][q];
                        xc[(16 + 4 * g + q) * kTLd + jl] = gj[cc][q];
                        xc[(32 + 4 * g + q) * kTLd + jl] = h;
                        // dW1[0][16m + 4g + q] += sum over the tile's samples of gf0 H + S1 . Q: a row sum, packed one per lane
                        // (sixteen per-lane accumulators instead: spills; the summands through the LDS tile as a fourth block: +7 us)
                        const float rs = row_sum16(fmaf(gf0[cc], h, s1 * qq[cc][q]));
                        w1p += jl == 4 * m + q ? rs : 0.f;
                    }
                }
            }
            BWD_PRIO_MFMA(1);
#pragma unroll
            for (int mk = 0; mk < 2; ++mk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float at = s_w0ta[((mk * 4 + m) * 4 + q) * 64 + lane];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        de[mk][cc] = mfma4(at, da[cc][q], de[mk][cc]);
                        rr[mk][cc] = mfma4(at, gj[cc][q], rr[mk][cc]);
                    }
                }
            if (want_pose) {          // rows 32..34 of W0'^T DA: d L / d (p / rescale) through the MLP's position inputs
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float at = pk->mw.w0ta[2][m][q][lane];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) dex[cc] = mfma4(at, da[cc][q], dex[cc]);
                }
            }
            if (WG && !kProbeNoContract) {
                // contraction over the tile's 16 samples: the lane supplies row jl, samples 4g .. 4g + 3 of every operand (the
                // tile's trip through LDS was in flight during the DE / RR products above)
                __builtin_amdgcn_wave_barrier();
                float4 a_da[NC], a_gj[NC], b_h[NC];
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    const float* __restrict__ xc = xt + cc * (kTRows * kTLd);
                    a_da[cc] = ld4(xc + jl * kTLd + 4 * g); a_gj[cc] = ld4(xc + (16 + jl) * kTLd + 4 * g); b_h[cc] = ld4(xc + (32 + jl) * kTLd + 4 * g);
                }
                __builtin_amdgcn_wave_barrier();
                f32x4 c0 = acc0[m][0], c1 = acc0[m][1], c3 = acc1[m];
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {        // tile by tile, in tile order: the sums are those of NC = 1
                    c0 = mfma4(a_da[cc].x, b_u[cc][0].x, c0); c1 = mfma4(a_da[cc].x, b_u[cc][1].x, c1); c3 = mfma4(a_gf[cc].x, b_h[cc].x, c3);
                    c0 = mfma4(a_da[cc].y, b_u[cc][0].y, c0); c1 = mfma4(a_da[cc].y, b_u[cc][1].y, c1); c3 = mfma4(a_gf[cc].y, b_h[cc].y, c3);
                    c0 = mfma4(a_da[cc].z, b_u[cc][0].z, c0); c1 = mfma4(a_da[cc].z, b_u[cc][1].z, c1); c3 = mfma4(a_gf[cc].z, b_h[cc].z, c3);
                    c0 = mfma4(a_da[cc].w, b_u[cc][0].w, c0); c1 = mfma4(a_da[cc].w, b_u[cc][1].w, c1); c3 = mfma4(a_gf[cc].w, b_h[cc].w, c3);
                    c0 = mfma4(a_gj[cc].x, b_v[cc][0].x, c0); c1 = mfma4(a_gj[cc].x, b_v[cc][1].x, c1);
                    c0 = mfma4(a_gj[cc].y, b_v[cc][0].y, c0); c1 = mfma4(a_gj[cc].y, b_v[cc][1].y, c1);
                    c0 = mfma4(a_gj[cc].z, b_v[cc][0].z, c0); c1 = mfma4(a_gj[cc].z, b_v[cc][1].z, c1);
                    c0 = mfma4(a_gj[cc].w, b_v[cc][0].w, c0); c1 = mfma4(a_gj[cc].w, b_v[cc][1].w, c1);
                }
                acc0[m][0] = c0; acc0[m][1] = c1; acc1[m] = c3;
                if (!kProbeNoPos) {
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                pacc[m][0] = pos_col<0, true>(pacc[m][0], a_da[cc], a_gj[cc], b_u[cc][2], b_v[cc][2]);
                pacc[m][1] = pos_col<1, true>(pacc[m][1], a_da[cc], a_gj[cc], b_u[cc][2], b_v[cc][2]);
                pacc[m][2] = pos_col<2, true>(pacc[m][2], a_da[cc], a_gj[cc], b_u[cc][2], b_v[cc][2]);
                pacc[m][3] = pos_col<3, false>(pacc[m][3], a_da[cc], a_gj[cc], b_u[cc][2], b_v[cc][2]);       // (V has no bias row)
                }
                }
                __builtin_amdgcn_sched_barrier(0);       // (the unrolled hidden blocks are not interleaved: their temporaries would add up)
            }
        }
        BWD_PRIO_LOOP(0); BWD_PRIO_MFMA(0);
        if (want_pose && g == 0) {
#pragma unroll
            for (int cc = 0; cc < NC; ++cc)
                if (live_c[cc]) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) *elem(out + w.dexyz, (uint32_t)a * P32 + is[cc]) = dex[cc][a];
                }
        }
        // scatter payload of the SDF grid: 16 bytes per (level, point); this lane owns rows 16 mk + 4 g + {0..3}
        // = levels 8 mk + 2 g and 8 mk + 2 g + 1.  Per-level bound of a single contribution |w de + D rr| <=
        // |de| + scale |g_n|_1 |rr| fixes the fixed-point quantum of the slab accumulators.
#pragma unroll
        for (int cc = 0; cc < NC; ++cc)
#pragma unroll
            for (int mk = 0; mk < 2; ++mk)
#pragma unroll
                for (int hv = 0; hv < 2; ++hv) {
                    const int l = 8 * mk + 2 * g + hv;
                    if (2 * l < ch1 && live_c[cc]) {
                        const float d0 = de[mk][cc][2 * hv], d1 = de[mk][cc][2 * hv + 1];
                        const float r0 = rr[mk][cc][2 * hv], r1 = rr[mk][cc][2 * hv + 1];
                        st4(elem(out + w.rec1, (uint32_t)l * P32 + is[cc], 4u), make_float4(d0, d1, r0, r1));
                        const float b = fmaxf(fabsf(d0), fabsf(d1)) + lsc.s[l] * g1[cc] * fmaxf(fabsf(r0), fabsf(r1));
                        bnd[mk][hv] = fmaxf(bnd[mk][hv], b);
                    }
                }

        }   // pass 0
        // ---- second field: plain first-order backward of its Geometry MLP
        if (DUAL && pass == 1) {
                const float* __restrict__ g_w1ta = s_w + 4 * 9 * 64;        // [m][4][lane]
            const float* __restrict__ g_w0ta = g_w1ta + 4 * 4 * 64;     // [mk][m][r][lane]
            float gf2b[4][NC];
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) ub[t][cc] = (4 * t + g) < ch2 ? (LS2FM_BWD_EJ_NT ? __builtin_nontemporal_load(elem(f_e2, (uint32_t)(4 * t + g) * P32 + is[cc])) : *elem(f_e2, (uint32_t)(4 * t + g) * P32 + is[cc])) : 0.f;
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float v = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) v = fmaf(s_wc[k][49 + 4 * t + g], dz[cc][k], v);
                    gf2b[t][cc] = v;
                    if (!WG && live_c[cc]) *elem(o_gf2, (uint32_t)(1 + 4 * t + g) * P32 + is[cc]) = v;
                }
                if (!WG && live_c[cc] && g == 0) o_gf2[is[cc]] = 0.f;
            }
            float4 b_u[NC][3], a_gf[NC];
            if (WG && kProbeNoTrans) {
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) b_u[cc][0] = b_u[cc][1] = b_u[cc][2] = a_gf[cc] = make_float4(1.f, 2.f, 3.f, 4.f);
            }
            if (WG && !kProbeNoTrans) {
                const int row2 = jl < 4 ? 32 + jl : 35;
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) {
                    float* __restrict__ xc = xt + cc * (kTRows * kTLd);
#pragma unroll
                    for (int t = 0; t < 9; ++t) xc[(4 * t + g) * kTLd + jl] = ub[t][cc];
                    __builtin_amdgcn_wave_barrier();
                    b_u[cc][0] = ld4(xc + jl * kTLd + 4 * g); b_u[cc][1] = ld4(xc + (16 + jl) * kTLd + 4 * g); b_u[cc][2] = ld4(xc + row2 * kTLd + 4 * g);
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int t = 0; t < 4; ++t) xc[(4 * t + g) * kTLd + jl] = gf2b[t][cc];
                    __builtin_amdgcn_wave_barrier();
                    a_gf[cc] = ld4(xc + jl * kTLd + 4 * g);
                    __builtin_amdgcn_wave_barrier();
                    gs16 += (a_gf[cc].x + a_gf[cc].y) + (a_gf[cc].z + a_gf[cc].w);
                }
            }
            f32x4 de2[2][NC];
#pragma unroll
            for (int cc = 0; cc < NC; ++cc) dex[cc] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mk = 0; mk < 2; ++mk)
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) de2[mk][cc] = f32x4{0.f, 0.f, 0.f, 0.f};
            BWD_PRIO_LOOP(1); BWD_PRIO_MFMA(1);
#pragma unroll kUnrollM
            for (int m = 0; m < 4; ++m) {
                f32x4 aa[NC], tt[NC];
#pragma unroll
                for (int cc = 0; cc < NC; ++cc) { aa[cc] = f32x4{0.f, 0.f, 0.f, 0.f}; tt[cc] = aa[cc]; }
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float a = s_w0a[(m * 9 + t) * 64 + lane];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) aa[cc] = mfma4(a, ub[t][cc], aa[cc]);
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float a = g_w1ta[(m * 4 + t) * 64 + lane];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) tt[cc] = mfma4(a, gf2b[t][cc], tt[cc]);
                }
                float da[NC][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
    #pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        float h, s1, s2;
                        softplus100(aa[cc][q], h, s1, s2);
                        da[cc][q] = s1 * tt[cc][q];
                        if (WG && !kProbeNoTrans) {
                            float* __restrict__ xc = xt + cc * (kTRows * kTLd);
                            xc[(4 * g + q) * kTLd + jl] = da[cc][q];
                            xc[(32 + 4 * g + q) * kTLd + jl] = h;
                        }
                    }
                }
#pragma unroll
                for (int mk = 0; mk < 2; ++mk)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float at = g_w0ta[((mk * 4 + m) * 4 + q) * 64 + lane];
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc) de2[mk][cc] = mfma4(at, da[cc][q], de2[mk][cc]);
                    }
                if (want_pose) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float at = pk->mw.w0tx_geo[m][q][lane];
#pragma unroll
                        for (int cc = 0; cc < NC; ++cc) dex[cc] = mfma4(at, da[cc][q], dex[cc]);
                    }
                }
                if (WG && !kProbeNoContract) {
                    __builtin_amdgcn_wave_barrier();
                    float4 a_da[NC], b_h[NC];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        const float* __restrict__ xc = xt + cc * (kTRows * kTLd);
                        a_da[cc] = ld4(xc + jl * kTLd + 4 * g); b_h[cc] = ld4(xc + (32 + jl) * kTLd + 4 * g);
                    }
                    __builtin_amdgcn_wave_barrier();
                    f32x4 c0 = acc0[m][0], c1 = acc0[m][1], c3 = acc1[m];
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                        c0 = mfma4(a_da[cc].x, b_u[cc][0].x, c0); c1 = mfma4(a_da[cc].x, b_u[cc][1].x, c1); c3 = mfma4(a_gf[cc].x, b_h[cc].x, c3);
                        c0 = mfma4(a_da[cc].y, b_u[cc][0].y, c0); c1 = mfma4(a_da[cc].y, b_u[cc][1].y, c1); c3 = mfma4(a_gf[cc].y, b_h[cc].y, c3);
                        c0 = mfma4(a_da[cc].z, b_u[cc][0].z, c0); c1 = mfma4(a_da[cc].z, b_u[cc][1].z, c1); c3 = mfma4(a_gf[cc].z, b_h[cc].z, c3);
                        c0 = mfma4(a_da[cc].w, b_u[cc][0].w, c0); c1 = mfma4(a_da[cc].w, b_u[cc][1].w, c1); c3 = mfma4(a_gf[cc].w, b_h[cc].w, c3);
                    }
                    acc0[m][0] = c0; acc0[m][1] = c1; acc1[m] = c3;
                    if (!kProbeNoPos) {
#pragma unroll
                    for (int cc = 0; cc < NC; ++cc) {
                    pacc[m][0] = pos_col<0, false>(pacc[m][0], a_da[cc], a_da[cc], b_u[cc][2], b_u[cc][2]);
                    pacc[m][1] = pos_col<1, false>(pacc[m][1], a_da[cc], a_da[cc], b_u[cc][2], b_u[cc][2]);
                    pacc[m][2] = pos_col<2, false>(pacc[m][2], a_da[cc], a_da[cc], b_u[cc][2], b_u[cc][2]);
                    pacc[m][3] = pos_col<3, false>(pacc[m][3], a_da[cc], a_da[cc], b_u[cc][2], b_u[cc][2]);
                    }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            BWD_PRIO_LOOP(0); BWD_PRIO_MFMA(0);
            if (want_pose && g == 0) {
#pragma unroll
                for (int cc = 0; cc < NC; ++cc)
                    if (live_c[cc]) {
#pragma unroll
                        for (int a = 0; a < 3; ++a) *elem(out + w.dexyz, (uint32_t)(3 + a) * P32 + is[cc]) = dex[cc][a];
                    }
            }
#pragma unroll
            for (int cc = 0; cc < NC; ++cc)
#pragma unroll
                for (int mk = 0; mk < 2; ++mk)
#pragma unroll
                    for (int hv = 0; hv < 2; ++hv) {
                        const int l = 8 * mk + 2 * g + hv;
                        if (2 * l < ch2 && live_c[cc]) {
                            const float d0 = de2[mk][cc][2 * hv], d1 = de2[mk][cc][2 * hv + 1];
                            st2(elem(out + w.rec2, (uint32_t)l * P32 + is[cc], 2u), make_float2(d0, d1));
                            bnd[mk][hv] = fmaxf(bnd[mk][hv], fmaxf(fabsf(d0), fabsf(d1)));
                        }
                    }
        }
    }
#pragma unroll
    for (int mk = 0; mk < 2; ++mk)
#pragma unroll
        for (int hv = 0; hv < 2; ++hv) {
            const int l = 8 * mk + 2 * g + hv;
            if (2 * l < (pass == 0 ? ch1 : ch2)) atomicMax(&s_bound[16 * pass + l], __float_as_int(bnd[mk][hv]));
        }
    if constexpr (WG) {
        float* __restrict__ dst = (pass == 0 ? slot_sdf : slot_geo) + r * ((pass == 0 ? kRegsSdf : kRegsGeo) * 64);      // the ray's slot (uniform); wg_flush adds the lane
        if (pass == 0) wg_flush<false>(acc0, acc1, pacc, w1p, gs16, g0s, s_w, dst, wave, n_waves, lane);
        else wg_flush<true>(acc0, acc1, pacc, w1p, gs16, g0s, s_w, dst, wave, n_waves, lane);
    }
    }
    // per-ray bounds of the scatter contributions (max over the ray's samples), [level][ray]
    __syncthreads();
    if (n < 32) out[w.smax + n * w.r_pad + r] = __int_as_float(s_bound[n]);
}

}  // namespace

int ls2fm_launch_shade_bwd(const FieldC& fc, const LevelScales& lsc, int dual, int ch1, int ch2, const WsLayout& w,
                           const Packed* pk, const float* center, const float* ray, int64_t n_rays, float* ws,
                           const Upstream& up, int want_pose, const ls2fm_grid_desc* zero_grid, float* dtable1, float* dtable2,
                           hipStream_t s, int fused_wgrad) {
    const int threads = (fc.n_samples + 63) / 64 * 64;
    ZeroJob zero{};
    zero.blocks = 64;
    zero.a = reinterpret_cast<float4*>(ws + w.wg);                       // w.wg .. w.dbeta: multiples of 64 floats
    zero.na = (w.dbeta - w.wg) / 4;
    int64_t first = 0, count = 0;
    ls2fm_scatter_zero_range(zero_grid, w.p, dtable2 != nullptr, &first, &count);      // entries (level offsets: multiples of 8)
    if (count > 0) {                          // (float-atomic flush of the point-split levels only, ls2fm_set_scatter_mode(0))
        zero.b = reinterpret_cast<float4*>(dtable1 + 2 * first);
        zero.nb = count / 2;
        if (dtable2) { zero.c = reinterpret_cast<float4*>(dtable2 + 2 * first); zero.nc = count / 2; }
    }
    const WgPartLayout pl = make_wg_part_layout(dual, n_rays, fc.n_samples);
    float* slot_sdf = ws + w.mpart + pl.slot_sdf;
    float* slot_geo = ws + w.mpart + pl.slot_geo;
    const unsigned grid = (unsigned)(n_rays + zero.blocks);
    const bool full = ch1 == 32 && (!dual || ch2 == 32);
#define LS2FM_SHADE_BWD3(DUAL, MAXT, POSE, WG, FULL)                                                                                \
    shade_bwd_kernel<DUAL, MAXT, POSE, WG, FULL><<<grid, threads, 0, s>>>(fc, lsc, ch1, ch2, w, pk, center, ray, ws, up, ws, zero, \
                                                                         n_rays, slot_sdf, slot_geo)
#define LS2FM_SHADE_BWD2(DUAL, MAXT, POSE, WG)                                                                                \
    do { if (WG && full) LS2FM_SHADE_BWD3(DUAL, MAXT, POSE, WG, WG); else LS2FM_SHADE_BWD3(DUAL, MAXT, POSE, WG, false); } while (0)
#define LS2FM_SHADE_BWD(DUAL, MAXT)                                                                                          \
    do {                                                                                                                    \
        if (fused_wgrad) { if (want_pose) LS2FM_SHADE_BWD2(DUAL, MAXT, true, true); else LS2FM_SHADE_BWD2(DUAL, MAXT, false, true); } \
        else { if (want_pose) LS2FM_SHADE_BWD2(DUAL, MAXT, true, false); else LS2FM_SHADE_BWD2(DUAL, MAXT, false, false); } \
    } while (0)
    if (dual) { if (threads <= 128) LS2FM_SHADE_BWD(true, 128); else if (threads <= 256) LS2FM_SHADE_BWD(true, 256); else LS2FM_SHADE_BWD(true, 512); }
    else      { if (threads <= 128) LS2FM_SHADE_BWD(false, 128); else if (threads <= 256) LS2FM_SHADE_BWD(false, 256); else LS2FM_SHADE_BWD(false, 512); }
#undef LS2FM_SHADE_BWD
#undef LS2FM_SHADE_BWD2
#undef LS2FM_SHADE_BWD3
    return LS2FM_OK;
}
