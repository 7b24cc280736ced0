// Weight gradients of the two Geometry MLPs (SDF field incl. the double-backward terms, second field) on the matrix
// cores, WITHOUT materialising the per-sample GEMM operands in HBM.
//
// shade_bwd leaves only the small per-sample upstream vectors (gf: 17 rows, v: 35 rows, gf2: 16 rows).  This kernel
// re-derives the hidden-layer quantities of a 16-sample tile with the same transposed MFMA chain as shade_bwd
//      A = W0' U,  Q = W0' V,  T = W1^T GF   ->   DA = S1.T + S2.w1_0.Q,  GJ = S1.w1_0,  H
// (~40 % of the kernel's MFMAs, against 173 MB of fp32 operand stores + 278 MB of loads for the old SoA hand-over) and
// contracts them over the samples:
//      dW0'  += DA U^T + GJ V^T     (64 x 36)      dW1[1..16] += GF H^T   (16 x 64)
//      dW1[0] += sum_s gf0 H + S1.Q (row sums, VALU)               db1 += sum_s GF
// The contraction index of these products is the SAMPLE, which sits on the wrong side of the accumulator layout, so each
// operand tile takes one trip through a wave-private LDS tile: written [feature][sample] from the accumulator / B
// layout, read back as one ds_read_b128 per lane (4 consecutive samples = the 4 k-slots of 4 MFMAs).
// Persistent waves keep all 85 accumulator registers for the whole kernel; workgroups reduce through LDS and write one
// partial each, summed by wgrad_mlp_reduce in a fixed order (deterministic).
#include "side_jobs.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

constexpr int kLd = 20;                          // floats per LDS tile row: 16 samples + 4 pad (16-B aligned rows)
constexpr int kRowsU = 36, kRowsGf = 20;
constexpr int kXRows = 2 * kRowsU + kRowsGf + 3 * 16;            // u, v, gf, da, gj, h

// EXTRA: tiles beyond n_tiles come from a second set of rows (wb, wsb).  A template parameter, not a run-time branch: the
// plain instantiation keeps the register allocation (234 / 144) the step's co-scheduling with the scatter was measured with --
// with the selects compiled in for every launch the benchmark step took 0.562 instead of 0.541 ms (A/B on one box).
template <bool GEO, bool EXTRA>
__global__ void __launch_bounds__(kWmThreads, 2)
wgrad_mlp_kernel(FieldC fc, int ch, WsLayout w, const Packed* __restrict__ pk, const float* __restrict__ center,
                 const float* __restrict__ ray, const float* __restrict__ ws, float* __restrict__ part, int n_tiles,
                 WsLayout wb, const float* __restrict__ wsb, int n_tiles_b) {
    constexpr int NT = GEO ? 4 : 5;
    constexpr int R = GEO ? kRegsGeo : kRegsSdf;
    __shared__ float s_w[4 * 9 * 64 + 4 * 5 * 64 + 4 * 4 * 64];
    __shared__ __attribute__((aligned(16))) float s_x[kWmWaves][kXRows * kLd];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jl = lane & 15, g = lane >> 4;
    // tiles [0, n_tiles) come from (w, ws), tiles [n_tiles, n_all) from the second set (wb, wsb) -- wave-uniform selects
    const int n_all = EXTRA ? n_tiles + n_tiles_b : n_tiles;
    {
        const float* src = GEO ? &pk->bg.w0a[0][0][0] : &pk->bs.w0a[0][0][0];      // w0a then w1ta are contiguous
        constexpr int n01 = 4 * 9 * 64 + 4 * NT * 64;
        for (int q = tid; q < n01 / 4; q += kWmThreads) reinterpret_cast<float4*>(s_w)[q] = reinterpret_cast<const float4*>(src)[q];
        if (!GEO)
            for (int q = tid; q < 4 * 4 * 64 / 4; q += kWmThreads)
                reinterpret_cast<float4*>(s_w + 4 * 9 * 64 + 4 * 5 * 64)[q] = reinterpret_cast<const float4*>(&pk->bs.w10[0][0][0])[q];
    }
    __syncthreads();
    const float* __restrict__ s_w0a = s_w;
    const float* __restrict__ s_w1ta = s_w + 4 * 9 * 64;
    const float* __restrict__ s_w10 = s_w + 4 * 9 * 64 + 4 * 5 * 64;
    float* __restrict__ xu = s_x[wave];
    float* __restrict__ xv = xu + kRowsU * kLd;
    float* __restrict__ xgf = xv + kRowsU * kLd;
    float* __restrict__ xda = xgf + kRowsGf * kLd;
    float* __restrict__ xgj = xda + 16 * kLd;
    float* __restrict__ xh = xgj + 16 * kLd;

    f32x4 acc0[4][3], acc1[4];
    float w1r0[4][4], gsum[5];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        acc1[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mk = 0; mk < 3; ++mk) acc0[m][mk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) w1r0[m][q] = 0.f;
    }
#pragma unroll
    for (int t = 0; t < 5; ++t) gsum[t] = 0.f;

    // B-layout operands of this lane's sample of one tile: rows k' = 4t + g / o = 4t + g.  The NEXT tile's operands are
    // loaded while the current tile is being contracted: a wave is alone on its SIMD here (one workgroup per CU beside the
    // scatter kernels), so nothing else hides the ~2 us of a dependent global load
    auto load_tile = [&](int tile, float (&ub)[9], float (&vb)[9], float (&gfb)[5], float& gf0) {
        const bool second = EXTRA && tile >= n_tiles;
        // (field-by-field scalar selects: a selected struct reference would be copied to scratch)
        const float* __restrict__ base = second ? wsb : ws;
        const uint32_t P32 = (uint32_t)(second ? wb.p_pad : w.p_pad);
        const float* __restrict__ f_e = base + (second ? (GEO ? wb.e2 : wb.e1) : (GEO ? w.e2 : w.e1));
        const float* __restrict__ f_v = base + (second ? wb.v : w.v);
        const float* __restrict__ f_gf = base + (second ? (GEO ? wb.gf2 : wb.gf) : (GEO ? w.gf2 : w.gf));
        const float* __restrict__ f_p3 = base + (second ? wb.p3 : w.p3);
        const uint32_t i = (uint32_t)(second ? tile - n_tiles : tile) * 16u + (uint32_t)jl;
        const bool live = tile < n_all && (int64_t)i < (second ? wb.p : w.p);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int c = 4 * t + g;
            const bool on = live && c < ch;
            ub[t] = on ? f_e[(uint32_t)c * P32 + i] : 0.f;
            vb[t] = (!GEO && on) ? f_v[(uint32_t)(3 + c) * P32 + i] : 0.f;
        }
        {
            // world position rows: written per sample by shade_bwd / the point-query backward (no ray arithmetic here: the
            // kernel serves ray samples and free points alike)
            const float pg = (live && g < 3) ? f_p3[(uint32_t)g * P32 + i] : 0.f;
            ub[8] = live ? (g < 3 ? pg / fc.rescale : 1.0f) : 0.f;
            vb[8] = (!GEO && live && g < 3) ? f_v[(uint32_t)g * P32 + i] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const int o = GEO ? 1 + 4 * t + g : 4 * t + g;
            gfb[t] = (live && t < NT && o < kOut) ? f_gf[(uint32_t)o * P32 + i] : 0.f;
        }
        gf0 = (!GEO && live) ? f_gf[i] : 0.f;
    };
    const int tile_step = gridDim.x * kWmWaves;
    float ub_n[9], vb_n[9], gfb_n[5], gf0_n;
    load_tile(blockIdx.x * kWmWaves + wave, ub_n, vb_n, gfb_n, gf0_n);
#pragma unroll 1
    for (int tile = blockIdx.x * kWmWaves + wave; tile < n_all; tile += tile_step) {
        float ub[9], vb[9], gfb[5];
#pragma unroll
        for (int t = 0; t < 9; ++t) { ub[t] = ub_n[t]; vb[t] = vb_n[t]; }
#pragma unroll
        for (int t = 0; t < 5; ++t) { gfb[t] = gfb_n[t]; gsum[t] += gfb[t]; }
        const float gf0 = gf0_n;
        load_tile(tile + tile_step, ub_n, vb_n, gfb_n, gf0_n);         // in flight during this tile's MFMA chain
        // transposed copies: [feature row][sample jl]
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            xu[(4 * t + g) * kLd + jl] = ub[t];
            if (!GEO) xv[(4 * t + g) * kLd + jl] = vb[t];
        }
#pragma unroll
        for (int t = 0; t < 5; ++t) xgf[(4 * t + g) * kLd + jl] = gfb[t];

#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 aa = f32x4{0.f, 0.f, 0.f, 0.f}, qq = aa, tt = aa;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float a = s_w0a[(m * 9 + t) * 64 + lane];
                aa = mfma4(a, ub[t], aa);
                if (!GEO) qq = mfma4(a, vb[t], qq);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) tt = mfma4(s_w1ta[(m * NT + t) * 64 + lane], gfb[t], tt);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float h, s1, s2;
                softplus100(aa[q], h, s1, s2);
                float da;
                if (GEO) {
                    da = s1 * tt[q];
                } else {
                    const float w10 = s_w10[(m * 4 + q) * 64 + lane];
                    da = fmaf(s1, tt[q], s2 * w10 * qq[q]);
                    xgj[(4 * g + q) * kLd + jl] = s1 * w10;
                    w1r0[m][q] += fmaf(gf0, h, s1 * qq[q]);
                }
                xda[(4 * g + q) * kLd + jl] = da;
                xh[(4 * g + q) * kLd + jl] = h;
            }
            __builtin_amdgcn_wave_barrier();
            // contraction over the 16 samples: lane supplies row jl, samples 4g..4g+3 of every operand
            const float4 a_da = *reinterpret_cast<const float4*>(xda + jl * kLd + 4 * g);
            const float4 a_gj = GEO ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(xgj + jl * kLd + 4 * g);
            const float4 b_h = *reinterpret_cast<const float4*>(xh + jl * kLd + 4 * g);
            const float4 a_gf = *reinterpret_cast<const float4*>(xgf + ((GEO ? 0 : 1) + jl) * kLd + 4 * g);
#pragma unroll
            for (int mk = 0; mk < 3; ++mk) {
                const bool row_on = 16 * mk + jl < kRowsU;
                float4 b_u = make_float4(0.f, 0.f, 0.f, 0.f), b_v = b_u;
                if (row_on) {
                    b_u = *reinterpret_cast<const float4*>(xu + (16 * mk + jl) * kLd + 4 * g);
                    if (!GEO) b_v = *reinterpret_cast<const float4*>(xv + (16 * mk + jl) * kLd + 4 * g);
                }
                f32x4 c = acc0[m][mk];
                c = mfma4(a_da.x, b_u.x, c); c = mfma4(a_da.y, b_u.y, c); c = mfma4(a_da.z, b_u.z, c); c = mfma4(a_da.w, b_u.w, c);
                if (!GEO) {
                    c = mfma4(a_gj.x, b_v.x, c); c = mfma4(a_gj.y, b_v.y, c); c = mfma4(a_gj.z, b_v.z, c); c = mfma4(a_gj.w, b_v.w, c);
                }
                acc0[m][mk] = c;
            }
            {
                f32x4 c = acc1[m];
                c = mfma4(a_gf.x, b_h.x, c); c = mfma4(a_gf.y, b_h.y, c); c = mfma4(a_gf.z, b_h.z, c); c = mfma4(a_gf.w, b_h.w, c);
                acc1[m] = c;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }

    // ---- row sums over the 16 sample lanes, then workgroup reduction through LDS and one partial per workgroup
    float regs[R];
    {
        int k = 0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int mk = 0; mk < 3; ++mk)
#pragma unroll
                for (int q = 0; q < 4; ++q) regs[k++] = acc0[m][mk][q];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q) regs[k++] = acc1[m][q];
        if (!GEO) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = w1r0[m][q];
#pragma unroll
                    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                    regs[k++] = v;
                }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v = gsum[t];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            regs[k++] = v;
        }
    }
    __syncthreads();                              // every wave is done with its LDS tiles
    float* red = &s_x[0][0];                      // [R][64]
    static_assert(kRegsSdf * 64 <= kWmWaves * kXRows * kLd, "reduction buffer fits in the tile storage");
    for (int wv = 0; wv < kWmWaves; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int k = 0; k < R; ++k) {
                if (wv == 0) red[k * 64 + lane] = regs[k];
                else red[k * 64 + lane] += regs[k];
            }
        }
        __syncthreads();
    }
    float* dst = part + (int64_t)blockIdx.x * (R * 64);
    for (int q = tid; q < R * 64; q += kWmThreads) dst[q] = red[q];
}

__global__ void __launch_bounds__(kWmThreads)
wgrad_dec_kernel(WsLayout w, int dual, int64_t n_rays, const float* __restrict__ ws, float* __restrict__ part, int dec_blocks, L1Job l1) {
    __shared__ __attribute__((aligned(16))) float s_arena[kRegsDec * 64];
    if ((int)blockIdx.x >= dec_blocks) {          // trailing workgroups: level-1 sums of shade_bwd's partials
        wgrad_l1_job_a<false>(l1, (int)blockIdx.x - dec_blocks, s_arena);
        return;
    }
    wgrad_dec_block_a<false>(w, dual, n_rays, ws, part, dec_blocks, (int)blockIdx.x, s_arena);
}

__global__ void __launch_bounds__(kWmThreads)
wgrad_reduce_all_kernel(WgradParts wp, float* __restrict__ wg) {
    reduce_partials_row(wp, wg, (int)blockIdx.x);
}

}  // namespace

// enqueue every weight-gradient kernel of the backward (+ the reduction of their partials) on `s`
int ls2fm_launch_wgrad_mlp(const FieldC& fc, int dual, int ch1, int ch2, const WsLayout& w, const Packed* pk, const float* center,
                           const float* ray, int64_t n_rays, float* ws, hipStream_t s, bool sdf_only, Ls2fmWgradParts* defer,
                           const Ls2fmWgradExtra* extra, int fused_wgrad) {
    const WgPartLayout pl = make_wg_part_layout(dual, n_rays, sdf_only ? 1 : fc.n_samples);
    // fused: the render's own samples were contracted by shade_bwd; only `extra`'s tiles (if any) are left for wgrad_mlp
    const int n_tiles = fused_wgrad ? 0 : (int)((w.p + 15) / 16);
    const int n_tiles_b = extra ? (int)((extra->w.p + 15) / 16) : 0;
    int blocks = (n_tiles + n_tiles_b + kWmWaves - 1) / kWmWaves;
    if (blocks > kWgradMlpBlocks) blocks = kWgradMlpBlocks;
    int dec_blocks = (int)(((w.p + 15) / 16 + kWmWaves - 1) / kWmWaves);
    if (dec_blocks > kWgradMlpBlocks) dec_blocks = kWgradMlpBlocks;
    float* l1_sdf = ws + w.mpart + pl.l1_sdf;
    float* l1_geo = ws + w.mpart + pl.l1_geo;
    // wgrad_mlp's own partials: behind the level-1 segment sums (SDF MLP) / in the second MLP's block (never both forms)
    float* part1 = l1_sdf + (int64_t)kL1Seg * 64 * kRegsSdf;
    float* part2 = l1_geo;
    float* part3 = ws + w.mpart + pl.dec;
    // second MLP first: the order only matters through how the two kernels share the CUs with scatter_fill / slab_accumulate
    // on the other queue -- measured (A/B on one box, graph replay of the benchmark step) 0.579 vs 0.583 ms
    if (dual && !fused_wgrad) {
        ls2fm_prof_begin(LS2FM_PROF_WGRAD_GEO, s);
        wgrad_mlp_kernel<true, false><<<blocks, kWmThreads, 0, s>>>(fc, ch2, w, pk, center, ray, ws, part2, n_tiles, w, ws, 0);
        ls2fm_prof_end(LS2FM_PROF_WGRAD_GEO, s);
    }
    // the second set's rows are written on another stream: its samples join this launch behind their event
    if (extra && extra->ready && hipStreamWaitEvent(s, (hipEvent_t)extra->ready, 0) != hipSuccess) return LS2FM_ERR_LAUNCH;
    if (blocks > 0) {
        ls2fm_prof_begin(LS2FM_PROF_WGRAD_MLP, s);
        if (extra)
            wgrad_mlp_kernel<false, true><<<blocks, kWmThreads, 0, s>>>(fc, ch1, w, pk, center, ray, ws, part1, n_tiles, extra->w, extra->ws, n_tiles_b);
        else
            wgrad_mlp_kernel<false, false><<<blocks, kWmThreads, 0, s>>>(fc, ch1, w, pk, center, ray, ws, part1, n_tiles, w, ws, 0);
        ls2fm_prof_end(LS2FM_PROF_WGRAD_MLP, s);
    }
    ls2fm_prof_begin(LS2FM_PROF_WGRAD_TAIL, s);
    if (sdf_only) {          // point queries: no decoder columns, the SDF MLP's partials only (blocks [0, kRegsSdf) of the reduction)
        wgrad_reduce_all_kernel<<<kRegsSdf, kWmThreads, 0, s>>>(WgradParts{part1, part2, part3, blocks, 0, 0, 0}, ws + w.wg);
    } else {
        L1Job l1{ws + w.mpart + pl.slot_sdf, ws + w.mpart + pl.slot_geo, l1_sdf, l1_geo, pl.n_slots, dual};
        const int l1_jobs = fused_wgrad ? kL1Seg * (kRegsSdf + (dual ? kRegsGeo : 0)) : 0;
        wgrad_dec_kernel<<<dec_blocks + l1_jobs, kWmThreads, 0, s>>>(w, dual, n_rays, ws, part3, dec_blocks, l1);
        // fused: the SDF MLP's partials are the kL1Seg segment sums followed by wgrad_mlp's `blocks` partials of the extra tiles
        const WgradParts wp = fused_wgrad ? WgradParts{l1_sdf, l1_geo, part3, kL1Seg + blocks, kL1Seg, dec_blocks, dual}
                                          : WgradParts{part1, part2, part3, blocks, blocks, dec_blocks, dual};
        if (defer) *defer = wp;
        else wgrad_reduce_all_kernel<<<kRegsSdf + (dual ? kRegsGeo : 0) + kRegsDec, kWmThreads, 0, s>>>(wp, ws + w.wg);
    }
    ls2fm_prof_end(LS2FM_PROF_WGRAD_TAIL, s);
    return LS2FM_OK;
}
