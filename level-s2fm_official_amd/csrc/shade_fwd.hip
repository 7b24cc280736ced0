// shade_fwd: per-sample field evaluation + per-ray compositing of Renderer.forward on the matrix cores.
//
//   workgroup = one ray (its N samples are the N dimension of every GEMM), wave = CT 16-column tiles of consecutive samples
//   (CT = 2: 32 samples per wave and 4 waves/SIMD for rays of up to 256 samples; CT = 4 beyond).
//   Everything is kept TRANSPOSED, [feature x sample], so that the accumulator tile of one v_mfma_f32_16x16x4_f32
//   (lane 16 g + jl holds rows 4 g + r, column jl) is directly the B operand of the next GEMM over that feature
//   dimension (k-slot g <-> row 4 g + r): the whole chain
//        A1 = W0' U          (64 x 36) x (36 x S)     U = [32 encoding channels ; p / rescale ; 1]  (bias folded in)
//        H  = softplus_100(A1), S1 = softplus'
//        F  = W1[1..16] H    (16 x 64) x (64 x S)     sdf row W1[0] H: 16 FMAs per lane + 2 cross-lane adds
//        R  = W0'^T (S1 . w1_0)   (48 x 64) x (64 x S)  -> analytic normal n = kappa (dU/dp)^T R      
//        [second field: A1, H, F only]
//   runs out of registers with no LDS staging; the MFMA-ordered weight copies come from prep_weights (L2-resident, one
//   coalesced 256-B load per operand).  The per-sample scalars (sdf, normal, colour) are then transposed through LDS to
//   one-thread-per-sample order for the transmittance scan (Renderer.py:33-49).
// fp32 MFMA = exact fp32 products and sums (no reduced-precision path); only the summation order differs from the
// scalar-FMA oracle.
#include "render_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// issue-priority experiment (round 6, profiles/r06_notes.md section 9; shade_bwd.hip gains 5 % from it): LS2FM_FWD_PRIO = 1 raises a
// wave's priority inside the hidden-block loop, 2 only around its MFMA clusters.  Four waves per SIMD here: 59.9 - 60.3 / 57.0 - 57.6
// against 57.3 - 57.9 us -- default 0, no s_setprio.
#ifndef LS2FM_FWD_PRIO
#define LS2FM_FWD_PRIO 0
#endif
#define FWD_PRIO_LOOP(x) do { if (LS2FM_FWD_PRIO == 1) __builtin_amdgcn_s_setprio(x); } while (0)
#define FWD_PRIO_MFMA(x) do { if (LS2FM_FWD_PRIO == 2) __builtin_amdgcn_s_setprio(x); } while (0)
// LS2FM_FWD_NT (round 6): the per-sample outputs (sdf, normal, colour, both feature blocks: what shade_bwd reads back ~100 us later, and the
// caller's [R, N] tensors) leave with non-temporal stores: shade_fwd 57.7 -> 56.4 us at C2, 378 -> 372 at 8192 rays, shade_bwd unchanged
// (profiles/r06_raw/c71_ab_fwd_nt.txt).
#ifndef LS2FM_FWD_NT
#define LS2FM_FWD_NT 1
#endif
#ifndef LS2FM_FWD_FULL
#define LS2FM_FWD_FULL 1
#endif
#ifndef LS2FM_FWD_FULL_J
#define LS2FM_FWD_FULL_J 0
#endif
#ifndef LS2FM_FWD_REPOS
#define LS2FM_FWD_REPOS 1
#endif
#ifndef LS2FM_FWD_OFF32
#define LS2FM_FWD_OFF32 1
#endif
#ifndef LS2FM_FWD_SCHED_BAR
#define LS2FM_FWD_SCHED_BAR 0
#endif
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// sum over the four 16-lane groups (g = 0..3) of a wave: every lane ends with the total
__device__ __forceinline__ float sum_over_groups(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// CT = 16-column tiles (of consecutive samples) per wave: 4 -> a wave covers 64 samples with ~250 registers (2 waves/SIMD);
// 2 -> 32 samples per wave, half the per-lane state, twice the waves (4 per SIMD) to hide the chain's latencies behind.
// FULL (round 6, as in shade_bwd.hip): all 32 encoding channels of both grids live AND the ray's samples fill every wave's tiles (N a
// multiple of 16 CT: every lane's samples live, aligned for the vector form) -- known at compile time: no per-channel predicates, no
// scalar-load fallback paths, the operand loads are straight-line code.
template <bool DUAL, int MAXS, int CT, bool FULL = false>
__global__ void __launch_bounds__(MAXS * 4 / CT, CT == 2 ? 4 : 2)
shade_fwd_kernel(FieldC fc, int ch1_arg, int ch2_arg, const Packed* __restrict__ pk, const float* __restrict__ center,
                 const float* __restrict__ ray, int64_t p_pad, const float* __restrict__ E1,
                 const float* __restrict__ J1, const float* __restrict__ E2, float* __restrict__ rgb_out,
                 float* __restrict__ sdfs_out, float* __restrict__ normals_out, float* __restrict__ depth_out,
                 float* __restrict__ nm_out, float* __restrict__ SDFV, float* __restrict__ NRM,
                 float* __restrict__ RGBS, float* __restrict__ FE, float* __restrict__ FE2, float* __restrict__ ROUT,
                 float* __restrict__ LPART, int64_t r_pad, ls2fm_loss_spec loss) {
    __shared__ float s_part[MAXS / (16 * CT)][10];   // per wave: tau total, then w-sums of rgb(3) depth n(3) opacity
    __shared__ float s_view[3];
    __shared__ float s_x[MAXS][8];              // per sample: sdf, normal(3), colour(3)
    __shared__ float s_w[kMfmaSdfFloats];       // MFMA-ordered weights of the field being evaluated (29 KB)
    const int N = fc.n_samples;
    const int ch1 = FULL ? 32 : ch1_arg, ch2 = FULL ? 32 : ch2_arg;
    const int64_t r = blockIdx.x;
    const int n = threadIdx.x, lane = n & 63, wave = n >> 6, n_waves = blockDim.x >> 6;
    const int jl = lane & 15, g = lane >> 4;
    const RayGeom gm = load_ray(fc, center, ray, r);
    const MfmaW& mw = pk->mw;
    {   // stage the SDF field's operand-ordered weights: one coalesced pass, then 256-B ds_reads per operand
        const float4* src = reinterpret_cast<const float4*>(&mw.sdf);
        float4* dst = reinterpret_cast<float4*>(s_w);
        for (int q = n; q < kMfmaSdfFloats / 4; q += blockDim.x) dst[q] = src[q];
    }
    const float* __restrict__ s_w0a = s_w;                                   // [m][t][lane]
    const float* __restrict__ s_w1a = s_w + 4 * 9 * 64;                      // [m][r][lane]
    const float* __restrict__ s_w0ta = s_w + kMfmaFieldFloats;               // [mk][m][r][lane]
    const float* __restrict__ s_w10 = s_w0ta + 3 * 4 * 4 * 64;               // [m][r][lane]

    // view-embedding part of the radiance decoder, once per ray (wave 0)
    if (wave == 0) {
        const float e = lane < kView ? view_component(gm.d, lane) : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float s = wave_sum(lane < kView ? pk->wc[c][6 + lane] * e : 0.f);
            if (lane == 0) s_view[c] = s + pk->bc[c];
        }
    }

    // ---- this lane's four samples (column jl of the wave's four tiles)
    int64_t is[CT];
    float pw[CT][3];
    bool live_c[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const int ns = 16 * CT * wave + CT * jl + c;    // a lane's columns are CONSECUTIVE samples: 16- (8-) byte loads
        live_c[c] = FULL || ns < N;
        const int nn = live_c[c] ? ns : N - 1;
        is[c] = r * N + nn;
        float x[3];
        sample_position(fc, gm, sample_depth(gm, nn, N), pw[c], x);
    }

    // all of the lane's samples live and aligned (p_pad is a multiple of 64): one float4 / float2 per channel row
    struct Cols { float v[CT]; };
    const bool vec = FULL || (live_c[CT - 1] && ((is[0] & (CT - 1)) == 0));
    const int64_t i0 = is[0], i1 = is[1], i2 = is[CT - 2], i3 = is[CT - 1];
    // LS2FM_FWD_OFF32 (round 6): the vector forms address uniform base + 32-bit byte offset (global_load ... v_off, s[base]) instead of a
    // 64-bit address per lane: one register and one add per load instead of two and a 64-bit add chain (LS2FM_MAX_RENDER_POINTS = 2^23
    // keeps 32 rows x p_pad x 12 bytes under 2^32)
    const unsigned pp = (unsigned)p_pad, u0 = (unsigned)i0;
    auto loadv = [=](const float* __restrict__ base, int rowidx) -> Cols {        // by value: by-reference captures cost spilled registers
        Cols o;
        const float* __restrict__ row = base + (int64_t)rowidx * p_pad;
        const char* __restrict__ cb = reinterpret_cast<const char*>(base);
        const unsigned off = ((unsigned)rowidx * pp + u0) * 4u;
        if (CT == 4) {
            if (vec) {
                const float4 t = LS2FM_FWD_OFF32 ? *reinterpret_cast<const float4*>(cb + off) : *reinterpret_cast<const float4*>(row + i0);
                o.v[0] = t.x; o.v[1] = t.y; o.v[CT - 2] = t.z; o.v[CT - 1] = t.w;
            }
            else { o.v[0] = row[i0]; o.v[1] = row[i1]; o.v[CT - 2] = row[i2]; o.v[CT - 1] = row[i3]; }
        } else {
            if (vec) {
                const float2 t = LS2FM_FWD_OFF32 ? *reinterpret_cast<const float2*>(cb + off) : *reinterpret_cast<const float2*>(row + i0);
                o.v[0] = t.x; o.v[1] = t.y;
            }
            else { o.v[0] = row[i0]; o.v[1] = row[i1]; }
        }
        return o;
    };
    // ---- B operands of layer 0: ub[t][c] = U[k' = 4t + g][sample c]
    float ub[9][CT];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        Cols e{};
        if ((4 * t + g) < ch1) e = loadv(E1, 4 * t + g);
#pragma unroll
        for (int c = 0; c < CT; ++c) ub[t][c] = e.v[c];
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) ub[8][c] = g < 3 ? (g == 0 ? pw[c][0] : (g == 1 ? pw[c][1] : pw[c][2])) / fc.rescale : 1.0f;

    __syncthreads();          // weights staged

    // ---- SDF field
    f32x4 racc[3][CT], facc[CT];
    float f0p[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        f0p[c] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) facc[c][q] = mw.b1a[0][q][lane];
#pragma unroll
        for (int mk = 0; mk < 3; ++mk) racc[mk][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    FWD_PRIO_LOOP(1);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        FWD_PRIO_MFMA(1);
        f32x4 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float a = s_w0a[(m * 9 + t) * 64 + lane];
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = mfma4(a, ub[t][c], acc[c]);
        }
        FWD_PRIO_MFMA(0);
        float ga[CT][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float w10 = s_w10[(m * 4 + q) * 64 + lane];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                float h, s1, s2;
                softplus100(acc[c][q], h, s1, s2);
                acc[c][q] = h;
                ga[c][q] = s1 * w10;
                f0p[c] = fmaf(w10, h, f0p[c]);
            }
        }
        FWD_PRIO_MFMA(1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a1 = s_w1a[(m * 4 + q) * 64 + lane];
#pragma unroll
            for (int c = 0; c < CT; ++c) facc[c] = mfma4(a1, acc[c][q], facc[c]);
        }
#pragma unroll
        for (int mk = 0; mk < 3; ++mk)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float at = s_w0ta[((mk * 4 + m) * 4 + q) * 64 + lane];
#pragma unroll
                for (int c = 0; c < CT; ++c) racc[mk][c] = mfma4(at, ga[c][q], racc[mk][c]);
            }
    }

    FWD_PRIO_LOOP(0); FWD_PRIO_MFMA(0);
#if LS2FM_FWD_SCHED_BAR
    __builtin_amdgcn_sched_barrier(0);       // the Jacobian loads (48 registers) stay behind the hidden-block loop
#endif
    // sdf and the analytic normal  n = kappa (R_p / rescale + inv_ext . J^T R_enc)
    float sdf[CT], nrm[CT][3];
    float part[CT][3];
#pragma unroll
    for (int c = 0; c < CT; ++c) part[c][0] = part[c][1] = part[c][2] = 0.f;
#pragma unroll
    for (int mk = 0; mk < 2; ++mk)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = 16 * mk + 4 * g + q;
            if (ch < (LS2FM_FWD_FULL_J ? ch1 : ch1_arg)) {
                // J rows are [channel][point][3]: the lane's CT consecutive samples are 3 CT consecutive floats
                const float* __restrict__ jrow = J1 + (int64_t)ch * p_pad * 3;
                const char* __restrict__ jb = reinterpret_cast<const char*>(J1) + ((unsigned)ch * (pp * 3u) + u0 * 3u) * 4u;
                float jv[CT][3];
                if (vec) {
                    if (CT == 4) {
                        const float4* q4 = LS2FM_FWD_OFF32 ? reinterpret_cast<const float4*>(jb) : reinterpret_cast<const float4*>(jrow + i0 * 3);
                        const float4 x0 = q4[0], x1 = q4[1], x2 = q4[2];
                        const float f[12] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w};
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int a = 0; a < 3; ++a) jv[c][a] = f[3 * c + a];
                    } else {
                        const float2* q2 = LS2FM_FWD_OFF32 ? reinterpret_cast<const float2*>(jb) : reinterpret_cast<const float2*>(jrow + i0 * 3);
                        const float2 x0 = q2[0], x1 = q2[1], x2 = q2[2];
                        const float f[6] = {x0.x, x0.y, x1.x, x1.y, x2.x, x2.y};
#pragma unroll
                        for (int c = 0; c < CT; ++c)
#pragma unroll
                            for (int a = 0; a < 3; ++a) jv[c][a] = f[(3 * c + a) % 6];
                    }
                } else {
                    const int64_t ix[4] = {i0, i1, i2, i3};
#pragma unroll
                    for (int c = 0; c < CT; ++c)
#pragma unroll
                        for (int a = 0; a < 3; ++a) jv[c][a] = jrow[ix[CT == 4 ? c : (c == 0 ? 0 : 1)] * 3 + a];
                }
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int a = 0; a < 3; ++a) part[c][a] = fmaf(jv[c][a], racc[mk][c][q], part[c][a]);
            }
        }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const float f0 = sum_over_groups(f0p[c]) + mw.b10[0];
        sdf[c] = fc.inside ? f0 / fc.scale_mlp : -f0 / fc.scale_mlp;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = part[c][a] * fc.inv_ext[a];
            if (g == 0) v += racc[2][c][a] / fc.rescale;          // rows 32..34 = the p / rescale inputs (group 0)
            nrm[c][a] = fc.kappa * sum_over_groups(v);
        }
    }
    const bool l0 = live_c[0], l1 = live_c[1], l2 = live_c[CT - 2], l3 = live_c[CT - 1];
    auto storev = [=](float* __restrict__ base, int rowidx, const f32x4 (&acc)[CT], int q) {
        float* __restrict__ row = base + (int64_t)rowidx * p_pad;
        char* __restrict__ cb = reinterpret_cast<char*>(base);
        const unsigned off = ((unsigned)rowidx * pp + u0) * 4u;
        if (CT == 4) {
            if (vec) {
                if (LS2FM_FWD_OFF32) *reinterpret_cast<float4*>(cb + off) = make_float4(acc[0][q], acc[1][q], acc[CT - 2][q], acc[CT - 1][q]);
                else *reinterpret_cast<float4*>(row + i0) = make_float4(acc[0][q], acc[1][q], acc[CT - 2][q], acc[CT - 1][q]);
            } else {
                if (l0) row[i0] = acc[0][q];
                if (l1) row[i1] = acc[1][q];
                if (l2) row[i2] = acc[CT - 2][q];
                if (l3) row[i3] = acc[CT - 1][q];
            }
        } else {
            if (vec) {
#if LS2FM_FWD_NT
                { typedef float f32x2 __attribute__((ext_vector_type(2))); f32x2 v = {acc[0][q], acc[1][q]};
                  __builtin_nontemporal_store(v, LS2FM_FWD_OFF32 ? reinterpret_cast<f32x2*>(cb + off) : reinterpret_cast<f32x2*>(row + i0)); }
#else
                if (LS2FM_FWD_OFF32) *reinterpret_cast<float2*>(cb + off) = make_float2(acc[0][q], acc[1][q]);
                else *reinterpret_cast<float2*>(row + i0) = make_float2(acc[0][q], acc[1][q]);
#endif
            } else {
                if (l0) row[i0] = acc[0][q];
                if (l1) row[i1] = acc[1][q];
            }
        }
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) storev(FE, 4 * g + q, facc, q);

    // ---- second field (Geometry_feat of RadF): features only
    f32x4 facc2[CT];
    if (DUAL) {
        __syncthreads();      // every wave is done with the SDF weights
        {
            const float4* src = reinterpret_cast<const float4*>(&mw.geo);
            float4* dst = reinterpret_cast<float4*>(s_w);
            for (int q = n; q < kMfmaFieldFloats / 4; q += blockDim.x) dst[q] = src[q];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            Cols e{};
            if ((4 * t + g) < ch2) e = loadv(E2, 4 * t + g);
#pragma unroll
            for (int c = 0; c < CT; ++c) ub[t][c] = e.v[c];
        }
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q) facc2[c][q] = mw.b1a[1][q][lane];
        __syncthreads();      // second field's weights staged
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f32x4 acc[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float a = s_w0a[(m * 9 + t) * 64 + lane];
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = mfma4(a, ub[t][c], acc[c]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float a1 = s_w1a[(m * 4 + q) * 64 + lane];
#pragma unroll
                for (int c = 0; c < CT; ++c) facc2[c] = mfma4(a1, softplus100_value(acc[c][q]), facc2[c]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) storev(FE2, 4 * g + q, facc2, q);
    }
    __syncthreads();          // s_view ready

    // ---- collapsed radiance decoder: z = Wc [p, n, view, f, f2] + bc ; this lane's feature slice, then the group sum
    float wf[3][4], wf2[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wf[k][q] = pk->wc[k][33 + 4 * g + q];
            wf2[k][q] = DUAL ? pk->wc[k][49 + 4 * g + q] : 0.f;
        }
#if LS2FM_FWD_REPOS
    // the sample positions are formed again for the decoder instead of being held across both MLP chains (six registers under a cap the
    // kernel spills at; same operations, same bits -- the sample number goes through an empty asm so that the two evaluations stay apart)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        int ns = 16 * CT * wave + CT * jl + c;
        asm volatile("" : "+v"(ns));
        float x[3];
        sample_position(fc, gm, sample_depth(gm, (FULL || ns < N) ? ns : N - 1, N), pw[c], x);
    }
#endif
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        float col[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float zp = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                zp = fmaf(wf[k][q], facc[c][q], zp);
                if (DUAL) zp = fmaf(wf2[k][q], facc2[c][q], zp);
            }
            float z = s_view[k] + sum_over_groups(zp);
#pragma unroll
            for (int a = 0; a < 3; ++a) z = fmaf(pk->wc[k][a], pw[c][a], z);
#pragma unroll
            for (int a = 0; a < 3; ++a) z = fmaf(pk->wc[k][3 + a], nrm[c][a], z);
            col[k] = 1.0f / (1.0f + expf(-z));
        }
        if (g == 0) {
            float* dst = s_x[16 * CT * wave + CT * jl + c];
            dst[0] = sdf[c];
#pragma unroll
            for (int a = 0; a < 3; ++a) { dst[1 + a] = nrm[c][a]; dst[4 + a] = col[a]; }
        }
    }
    __syncthreads();

    // ---- one thread per sample from here on
    const bool live = n < N;
    const int64_t i = r * N + (live ? n : N - 1);
    const float t = sample_depth(gm, live ? n : N - 1, N);
    const float t_next = sample_depth(gm, (live ? n : N - 1) + 1, N);
    const int nx = live ? n : N - 1;             // (CT < 4: more threads than samples; rows beyond the last tile are never written)
    const float sdf_n = s_x[nx][0];
    const float nrm_n[3] = {s_x[nx][1], s_x[nx][2], s_x[nx][3]};
    const float col_n[3] = {s_x[nx][4], s_x[nx][5], s_x[nx][6]};
    const float sigma = sigma_of(sdf_n, pk->alpha, pk->beta);
    if (live) {
#if LS2FM_FWD_NT
        __builtin_nontemporal_store(sdf_n, sdfs_out + i);
        __builtin_nontemporal_store(sdf_n, SDFV + i);
#else
        sdfs_out[i] = sdf_n;
        SDFV[i] = sdf_n;
#endif
#pragma unroll
        for (int a = 0; a < 3; ++a) {
#if LS2FM_FWD_NT
            __builtin_nontemporal_store(nrm_n[a], normals_out + i * 3 + a);
            __builtin_nontemporal_store(nrm_n[a], NRM + a * p_pad + i);
            __builtin_nontemporal_store(col_n[a], RGBS + a * p_pad + i);
#else
            normals_out[i * 3 + a] = nrm_n[a];
            NRM[a * p_pad + i] = nrm_n[a];
            RGBS[a * p_pad + i] = col_n[a];
#endif
        }
    }

    // composite (Renderer.py:33-49): N-1 intervals, exclusive prefix of sigma*delta
    const float ray_len = sqrtf(gm.d[0] * gm.d[0] + gm.d[1] * gm.d[1] + gm.d[2] * gm.d[2]);
    const bool interval = n < N - 1;
    const float tau = interval ? sigma * ((t_next - t) * ray_len) : 0.f;
    const float incl = wave_scan_incl(tau, lane);
    if (lane == 63) s_part[wave][0] = incl;
    __syncthreads();
    float before = incl - tau;
    for (int w = 0; w < wave; ++w) before += s_part[w][0];
    const float wgt = interval ? expf(-before) * (1.0f - expf(-tau)) : 0.f;
    float sums[8] = {wgt * col_n[0], wgt * col_n[1], wgt * col_n[2], wgt * t, wgt * nrm_n[0], wgt * nrm_n[1], wgt * nrm_n[2], wgt};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float s = wave_sum(sums[q]);
        if (lane == 0) s_part[wave][1 + q] = s;
    }
    // fused loss head: this ray's eikonal partial  sum_n | |n_n| - 1 |  (Initialization.py:257-258, BA.py:193-194)
    const bool want_eik = loss.rgb_gt != nullptr && ls2fm_in_eik(loss, r);
    if (loss.rgb_gt != nullptr) {
        float e = 0.f;
        if (want_eik && live) e = fabsf(sqrtf(nrm_n[0] * nrm_n[0] + nrm_n[1] * nrm_n[1] + nrm_n[2] * nrm_n[2]) - 1.0f);
        e = wave_sum(e);
        if (lane == 0) s_part[wave][9] = e;
    }
    __syncthreads();
    if (n == N - 1) {         // the last sample's thread owns t_last / n_last and writes the ray outputs
        float tot[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            tot[q] = 0.f;
            for (int w = 0; w < n_waves; ++w) tot[q] += s_part[w][1 + q];
        }
        const float rest = 1.0f - tot[7];
        float rgb_r[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { rgb_r[c] = tot[c] + rest * fc.bg[c]; rgb_out[r * 3 + c] = rgb_r[c]; }
        const float depth_r = tot[3] + rest * t;
        depth_out[r] = depth_r;
#pragma unroll
        for (int a = 0; a < 3; ++a) nm_out[r * 3 + a] = tot[4 + a] + rest * nrm_n[a];
        if (loss.rgb_gt != nullptr) {
            // per-ray partial sums of the loss head (pipelines/Camera.py:520-535): reduced in fixed order by loss_reduce; the
            // ray's rgb / depth are kept for the backward, which forms the upstream of the outputs itself
            float l1 = 0.f, sq = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float d = rgb_r[c] - loss.rgb_gt[r * 3 + c];
                l1 += fabsf(d);
                sq = fmaf(d, d, sq);
                ROUT[c * r_pad + r] = rgb_r[c];
            }
            ROUT[3 * r_pad + r] = depth_r;
            float eik = 0.f;
            for (int w = 0; w < n_waves; ++w) eik += s_part[w][9];
            LPART[0 * r_pad + r] = l1;
            LPART[1 * r_pad + r] = eik;
            LPART[2 * r_pad + r] = depth_r;        // the depth-consistency term is formed by the reduction: depth_ref may still be
                                                   // on its way here (ls2fm_render_opts.loss_inputs_ready)
            LPART[3 * r_pad + r] = ls2fm_in_mse(loss, r) ? sq : 0.f;
        }
    }
}

}  // namespace

int ls2fm_launch_shade_fwd(const FieldC& fc, int dual, int ch1, int ch2, const Packed* pk, const float* center, const float* ray,
                           int64_t n_rays, const WsLayout& w, float* ws, float* rgb, float* sdfs_volume, float* normals,
                           float* depth_mlp, float* normal_mlp, const ls2fm_loss_spec* loss, hipStream_t s) {
    // up to 256 samples per ray: two tiles per wave (4 waves/SIMD; 71.6 -> 66.1 us at the benchmark); longer rays keep four
    // (a 1024-thread workgroup of the two-tile form does not fit the register budget of 4 waves/SIMD without spilling into LDS)
    const bool small = fc.n_samples <= 256;
    const int per_wave = small ? 32 : 64;
    const int threads = (fc.n_samples + per_wave - 1) / per_wave * 64;
    ls2fm_loss_spec ls{};            // rgb_gt == null: plain render
    if (loss) ls = *loss;
    const bool full = LS2FM_FWD_FULL && small && ch1 == 32 && (!dual || ch2 == 32) && fc.n_samples % 32 == 0;
#define LS2FM_SHADE_FWD(DUAL, MAXS, CT, FULL)                                                                      \
    shade_fwd_kernel<DUAL, MAXS, CT, FULL><<<(unsigned)n_rays, threads, 0, s>>>(                                    \
        fc, ch1, ch2, pk, center, ray, w.p_pad, ws + w.e1, ws + w.j1, DUAL ? ws + w.e2 : nullptr, rgb, sdfs_volume, \
        normals, depth_mlp, normal_mlp, ws + w.sdfv, ws + w.nrm, ws + w.rgbs, ws + w.fe, DUAL ? ws + w.fe2 : nullptr,        \
        ws + w.rout, ws + w.lpart, w.r_pad, ls)
    if (dual) { if (full) LS2FM_SHADE_FWD(true, 256, 2, true); else if (small) LS2FM_SHADE_FWD(true, 256, 2, false); else LS2FM_SHADE_FWD(true, 512, 4, false); }
    else      { if (full) LS2FM_SHADE_FWD(false, 256, 2, true); else if (small) LS2FM_SHADE_FWD(false, 256, 2, false); else LS2FM_SHADE_FWD(false, 512, 4, false); }
#undef LS2FM_SHADE_FWD
    return LS2FM_OK;
}
