// Micro-benchmark behind round 3's table-gradient scatter: PUSH (32-byte payload items, round 2) against PULL (8-byte index
// items + one 64-byte record gather per item) for the slab-owned fixed-point accumulate, on the item lists of a real C2 batch
// (ETH3D grid L16/F2/T19, 1024 rays x 128 samples, dual field).
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -ffp-contract=off -w tools/pull_bench.hip -o tools/pull_bench
//   run:   tools/pull_bench [n_rays=1024] [only_level=-1]
// Template axes: MODE (0 push32, 1 pull: record {d0 d1 r0 r1 | e0 e1 gx gy | gz x y z | pad}, item {pt<<2|pair, ij}),
// SSHIFT (log2 entries per slab: 12 = 128 KiB of u64 x 4 accumulators, one workgroup per CU; 11 = 64 KiB, two per CU),
// THREADS, ABITS (64: exact u64 fixed point; 32: u32 fixed point, half the LDS), PHASE (0 full, 1 no LDS atomics, 2 no streaming:
// zero + flush only).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
constexpr int kL = 16;
constexpr uint32_t PY = 2654435761u, PZ = 805459861u;

struct Levels { float scale[kL]; uint32_t res[kL], size[kL], offset[kL], hashed[kL]; };
struct __attribute__((aligned(16))) PushItem { uint32_t ij; float wx, a0, a1, b0, b1, c0, c1; };
struct __attribute__((aligned(8))) PullItem { uint32_t pp, ij; };
struct Work { int l, slab, start, n, split; };

__host__ __device__ inline void pos_fract(float x, float scale, uint32_t& cell, float& w) {
    const float pos = fmaf(scale, x, 0.5f);
    const float fl = floorf(pos);
    cell = (uint32_t)(int32_t)fl;
    w = pos - fl;
}

template <int ABITS> struct AccT { typedef u64 type; };
template <> struct AccT<32> { typedef uint32_t type; };

template <int ABITS>
__device__ __forceinline__ void add_fixed(typename AccT<ABITS>::type* slot, float v, float to_fixed) {
    if (ABITS == 64) atomicAdd(reinterpret_cast<u64*>(slot), (u64)__float2ll_rn(v * to_fixed));
    else atomicAdd(reinterpret_cast<uint32_t*>(slot), (uint32_t)__float2int_rn(v * to_fixed));
}

template <int MODE, int SSHIFT, int THREADS, int ABITS, int PHASE>
__global__ void __launch_bounds__(THREADS)
acc_kernel(Levels lv, const void* __restrict__ items, const float* __restrict__ rec, int64_t P, float* __restrict__ dt1,
           float* __restrict__ dt2, const Work* __restrict__ work) {
    typedef typename AccT<ABITS>::type acc_t;
    __shared__ acc_t acc[4 << SSHIFT];
    const int tid = threadIdx.x;
    const Work wk = work[blockIdx.x];
    const int l = wk.l, slab = wk.slab;
    const uint32_t size = lv.size[l];
    const uint32_t lo = (uint32_t)slab << SSHIFT;
    const uint32_t hi = lo + (1u << SSHIFT) < size ? lo + (1u << SSHIFT) : size;
    const int n_items = wk.n;
    const int64_t base = wk.start;
    for (int e = tid; e < 4 * (int)(hi - lo); e += THREADS) acc[e] = 0;
    const float to_fixed = ldexpf(1.0f, ABITS == 64 ? 30 : 12);
    const double to_float = ldexp(1.0, ABITS == 64 ? -30 : -12);
    const float scale = lv.scale[l];
    __syncthreads();
    float sink = 0.f;
    if (PHASE != 2) {
        if (MODE == 0) {
            const PushItem* __restrict__ list = reinterpret_cast<const PushItem*>(items) + base;
            for (int j = tid; j < n_items; j += THREADS) {
                const uint4 q0 = reinterpret_cast<const uint4*>(list + j)[0], q1 = reinterpret_cast<const uint4*>(list + j)[1];
                const uint32_t i0 = q0.x & 0xFFFFu, i1 = q0.x >> 16;
                const float wx = __uint_as_float(q0.y), a0 = __uint_as_float(q0.z), a1 = __uint_as_float(q0.w);
                const float b0 = __uint_as_float(q1.x), b1 = __uint_as_float(q1.y), c0 = __uint_as_float(q1.z), c1 = __uint_as_float(q1.w);
                const float px0 = 1.0f - wx;
                if (PHASE == 1) { sink += px0 * a0 + a1 + b0 + b1 + c0 + c1 + (float)(i0 + i1); continue; }
                if (i0 != 0xFFFFu) {
                    acc_t* s = acc + 4 * i0;
                    add_fixed<ABITS>(s + 0, fmaf(px0, a0, -b0), to_fixed); add_fixed<ABITS>(s + 1, fmaf(px0, a1, -b1), to_fixed);
                    add_fixed<ABITS>(s + 2, px0 * c0, to_fixed); add_fixed<ABITS>(s + 3, px0 * c1, to_fixed);
                }
                if (i1 != 0xFFFFu) {
                    acc_t* s = acc + 4 * i1;
                    add_fixed<ABITS>(s + 0, fmaf(wx, a0, b0), to_fixed); add_fixed<ABITS>(s + 1, fmaf(wx, a1, b1), to_fixed);
                    add_fixed<ABITS>(s + 2, wx * c0, to_fixed); add_fixed<ABITS>(s + 3, wx * c1, to_fixed);
                }
            }
        } else {
            const PullItem* __restrict__ list = reinterpret_cast<const PullItem*>(items) + base;
            const float* __restrict__ rl = rec + (int64_t)l * P * 16;
            for (int j = tid; j < n_items; j += THREADS) {
                const PullItem it = list[j];
                const uint32_t pt = it.pp >> 2;
                const float4* r = reinterpret_cast<const float4*>(rl + (int64_t)pt * 16);
                const float4 ra = r[0], rb = r[1], rc = r[2];
                const uint32_t pair = it.pp & 3u;
                const uint32_t i0 = it.ij & 0xFFFFu, i1 = it.ij >> 16;
                const float d0 = ra.x, d1 = ra.y, r0 = ra.z, r1 = ra.w, e0 = rb.x, e1 = rb.y;
                const float qd[3] = {scale * rb.z, scale * rb.w, scale * rc.x};
                const float x[3] = {rc.y, rc.z, rc.w};
                uint32_t g;
                float w[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) pos_fract(x[a], scale, g, w[a]);
                const int by = (int)(pair & 1u), bz = (int)(pair >> 1);
                const float py = by ? w[1] : 1.0f - w[1], pz = bz ? w[2] : 1.0f - w[2];
                const float pyz = py * pz;
                const float qyz = (by ? qd[1] : -qd[1]) * pz + py * (bz ? qd[2] : -qd[2]);
                const float a0 = fmaf(qyz, r0, pyz * d0), a1 = fmaf(qyz, r1, pyz * d1);
                const float b0 = qd[0] * pyz * r0, b1 = qd[0] * pyz * r1;
                const float c0 = pyz * e0, c1 = pyz * e1;
                const float wx = w[0], px0 = 1.0f - wx;
                if (PHASE == 1) { sink += px0 * a0 + a1 + b0 + b1 + c0 + c1 + (float)(i0 + i1); continue; }
                if (i0 != 0xFFFFu) {
                    acc_t* s = acc + 4 * i0;
                    add_fixed<ABITS>(s + 0, fmaf(px0, a0, -b0), to_fixed); add_fixed<ABITS>(s + 1, fmaf(px0, a1, -b1), to_fixed);
                    add_fixed<ABITS>(s + 2, px0 * c0, to_fixed); add_fixed<ABITS>(s + 3, px0 * c1, to_fixed);
                }
                if (i1 != 0xFFFFu) {
                    acc_t* s = acc + 4 * i1;
                    add_fixed<ABITS>(s + 0, fmaf(wx, a0, b0), to_fixed); add_fixed<ABITS>(s + 1, fmaf(wx, a1, b1), to_fixed);
                    add_fixed<ABITS>(s + 2, wx * c0, to_fixed); add_fixed<ABITS>(s + 3, wx * c1, to_fixed);
                }
            }
        }
    }
    if (PHASE == 1 && sink == 123.456f) acc[tid] = 1;
    __syncthreads();
    float* d1 = dt1 + 2ull * (lv.offset[l] + lo);
    float* d2 = dt2 + 2ull * (lv.offset[l] + lo);
    for (int e = tid; e < 4 * (int)(hi - lo); e += THREADS) {
        const int entry = e >> 2, f = e & 3;
        float* dst = (f >= 2 ? d2 : d1) + 2 * entry + (f & 1);
        const float v = ABITS == 64 ? (float)((double)(long long)acc[e] * to_float) : (float)((double)(int)acc[e] * to_float);
        if (wk.split) { if (acc[e] != 0) atomicAdd(dst, v); } else *dst = v;
    }
}

__global__ void touch_kernel(uint32_t* p, int64_t n, uint32_t zero) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = p[i] + zero;
}

template <typename F, typename G> float time_us(F f, G pre, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    pre(); f(); hipDeviceSynchronize();
    float tot = 0.f;
    for (int r = 0; r < reps; ++r) {
        pre();
        hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); tot += ms;
    }
    return tot * 1000.f / reps;
}

struct Data {
    Levels lv; int64_t P; uint32_t n_entries;
    std::vector<float> xs, rec16;
    float *d_rec16, *d_t1, *d_t2;
};

template <int SSHIFT>
void run_config(Data& D, int only_level) {
    const int bins = (1 << 19) >> SSHIFT;
    const Levels& lv = D.lv;
    const int64_t P = D.P;
    std::vector<std::vector<PullItem>> lists((size_t)kL * bins);
    for (int l = 0; l < kL; ++l)
        for (int64_t i = 0; i < P; ++i) {
            uint32_t g[3]; float w;
            for (int a = 0; a < 3; ++a) pos_fract(D.xs[i * 4 + a], lv.scale[l], g[a], w);
            for (uint32_t c = 0; c < 4; ++c) {
                const uint32_t cy = g[1] + (c & 1), cz = g[2] + (c >> 1);
                uint32_t i0, i1;
                if (lv.hashed[l]) { i0 = (g[0] ^ cy * PY ^ cz * PZ) & (lv.size[l] - 1u); i1 = ((g[0] + 1u) ^ cy * PY ^ cz * PZ) & (lv.size[l] - 1u); }
                else { i0 = (g[0] + cy * lv.res[l] + cz * lv.res[l] * lv.res[l]) % lv.size[l]; i1 = (g[0] + 1u + cy * lv.res[l] + cz * lv.res[l] * lv.res[l]) % lv.size[l]; }
                const uint32_t s0 = i0 >> SSHIFT, s1 = i1 >> SSHIFT, m = (1u << SSHIFT) - 1u, pp = ((uint32_t)i << 2) | c;
                if (s0 == s1) lists[(size_t)l * bins + s0].push_back({pp, (i0 & m) | ((i1 & m) << 16)});
                else { lists[(size_t)l * bins + s0].push_back({pp, (i0 & m) | 0xFFFF0000u}); lists[(size_t)l * bins + s1].push_back({pp, 0xFFFFu | ((i1 & m) << 16)}); }
            }
        }
    std::vector<int> start((size_t)kL * bins);
    int64_t total = 0;
    for (size_t k = 0; k < lists.size(); ++k) { start[k] = (int)total; total += (int64_t)lists[k].size(); }
    std::vector<PullItem> pull(total);
    std::vector<PushItem> push(total);
    for (size_t k = 0; k < lists.size(); ++k) {
        const int l = (int)(k / bins);
        for (size_t j = 0; j < lists[k].size(); ++j) {
            const PullItem it = lists[k][j];
            pull[start[k] + j] = it;
            const uint32_t pt = it.pp >> 2, pair = it.pp & 3u;
            const float* v = &D.rec16[((size_t)l * P + pt) * 16];
            uint32_t g; float w[3];
            for (int a = 0; a < 3; ++a) pos_fract(v[9 + a], lv.scale[l], g, w[a]);
            const float qd[3] = {lv.scale[l] * v[6], lv.scale[l] * v[7], lv.scale[l] * v[8]};
            const int by = pair & 1, bz = pair >> 1;
            const float py = by ? w[1] : 1.0f - w[1], pz = bz ? w[2] : 1.0f - w[2], pyz = py * pz;
            const float qyz = (by ? qd[1] : -qd[1]) * pz + py * (bz ? qd[2] : -qd[2]);
            PushItem pi;
            pi.ij = it.ij; pi.wx = w[0];
            pi.a0 = fmaf(qyz, v[2], pyz * v[0]); pi.a1 = fmaf(qyz, v[3], pyz * v[1]);
            pi.b0 = qd[0] * pyz * v[2]; pi.b1 = qd[0] * pyz * v[3];
            pi.c0 = pyz * v[4]; pi.c1 = pyz * v[5];
            push[start[k] + j] = pi;
        }
    }
    // work list, level-major; lists longer than `target` split into parts (atomic flush into zeroed tables)
    const int target = 4 << SSHIFT;
    std::vector<Work> work;
    for (int l = 0; l < kL; ++l) {
        if (only_level >= 0 && l != only_level) continue;
        for (int sl = 0; sl < bins; ++sl) {
            if (((uint32_t)sl << SSHIFT) >= lv.size[l]) continue;
            const int n = (int)lists[(size_t)l * bins + sl].size();
            int parts = (n + target - 1) / target; if (parts < 1) parts = 1; if (parts > 64) parts = 64;
            for (int p = 0; p < parts; ++p) {
                const int a = (int)((int64_t)n * p / parts), b = (int)((int64_t)n * (p + 1) / parts);
                work.push_back(Work{l, sl, start[(size_t)l * bins + sl] + a, b - a, parts > 1});
            }
        }
    }
    Work* d_work; PullItem* d_pull; PushItem* d_push;
    CK(hipMalloc(&d_work, work.size() * sizeof(Work))); CK(hipMemcpy(d_work, work.data(), work.size() * sizeof(Work), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_pull, total * 8)); CK(hipMemcpy(d_pull, pull.data(), total * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_push, total * 32)); CK(hipMemcpy(d_push, push.data(), total * 32, hipMemcpyHostToDevice));
    const int blocks = (int)work.size(), reps = 20;
    printf("slab 2^%d entries (%d KiB u64 / %d KiB u32), %d work items, %lld items (%.3f per point-level)%s\n", SSHIFT, (32 << SSHIFT) >> 10,
           (16 << SSHIFT) >> 10, blocks, (long long)total, (double)total / (P * kL), only_level >= 0 ? " [one level]" : "");
    std::vector<float> ref((size_t)D.n_entries * 2), got((size_t)D.n_entries * 2);
    bool have_ref = false;
    auto check = [&](const char* name) {
        CK(hipMemcpy(got.data(), D.d_t1, got.size() * 4, hipMemcpyDeviceToHost));
        if (!have_ref) { ref = got; have_ref = true; return; }
        double worst = 0, mx = 0;
        for (size_t k = 0; k < got.size(); ++k) { worst = fmax(worst, fabs((double)got[k] - ref[k])); mx = fmax(mx, fabs((double)ref[k])); }
        printf("      %-24s max |diff| vs first = %.3g (max |ref| %.3g)\n", name, worst, mx);
    };
    auto zero = [&] { hipMemsetAsync(D.d_t1, 0, (size_t)D.n_entries * 8, 0); hipMemsetAsync(D.d_t2, 0, (size_t)D.n_entries * 8, 0); };
    auto pre_push = [&] { zero(); touch_kernel<<<1024, 256>>>((uint32_t*)d_push, total * 8, 0u); };
    auto pre_pull = [&] { zero(); touch_kernel<<<1024, 256>>>((uint32_t*)D.d_rec16, (int64_t)D.rec16.size(), 0u); };
#define TIME(MODE, THREADS, ABITS, PHASE, label)                                                                                         \
    {                                                                                                                                    \
        auto f = [&] { acc_kernel<MODE, SSHIFT, THREADS, ABITS, PHASE><<<blocks, THREADS>>>(lv, MODE ? (const void*)d_pull : (const void*)d_push, D.d_rec16, P, D.d_t1, D.d_t2, d_work); }; \
        const float us = MODE ? time_us(f, pre_pull, reps) : time_us(f, pre_push, reps);                                                 \
        printf("  %-46s %7.1f us\n", label, us);                                                                                        \
        if (PHASE == 0 && ABITS == 64) check(label);                                                                                     \
    }
    TIME(0, 1024, 64, 0, "push32  1024 thr u64");
    TIME(0, 1024, 64, 1, "push32  1024 thr u64  no atomics");
    TIME(0, 1024, 64, 2, "push32  1024 thr u64  zero + flush only");
    TIME(0, 512, 64, 0, "push32   512 thr u64");
    TIME(0, 1024, 32, 0, "push32  1024 thr u32");
    TIME(0, 512, 32, 0, "push32   512 thr u32");
    TIME(1, 1024, 64, 0, "pull64B 1024 thr u64");
    TIME(1, 1024, 64, 1, "pull64B 1024 thr u64  no atomics");
    TIME(1, 512, 64, 0, "pull64B  512 thr u64");
    TIME(1, 1024, 32, 0, "pull64B 1024 thr u32");
    TIME(1, 512, 32, 0, "pull64B  512 thr u32");
#undef TIME
    CK(hipFree(d_work)); CK(hipFree(d_pull)); CK(hipFree(d_push));
}

int main(int argc, char** argv) {
    const int n_rays = argc > 1 ? atoi(argv[1]) : 1024, N = 128;
    const int only_level = argc > 2 ? atoi(argv[2]) : -1;
    Data D;
    D.P = (int64_t)n_rays * N;
    const int64_t P = D.P;
    Levels& lv = D.lv;
    const double b = exp(log(2048.0 * 5.0 / 16.0) / 15.0);
    uint32_t off = 0;
    for (int l = 0; l < kL; ++l) {
        const float sc = (float)(exp2(l * log2(b)) * 16.0 - 1.0);
        const uint32_t res = (uint32_t)ceilf(sc) + 1u;
        const uint64_t dense = (uint64_t)res * res * res;
        uint32_t size = dense > (1u << 19) ? (1u << 19) : (uint32_t)((dense + 7) / 8 * 8);
        lv.scale[l] = sc; lv.res[l] = res; lv.size[l] = size; lv.offset[l] = off; lv.hashed[l] = dense > size;
        off += size;
    }
    D.n_entries = off;
    std::mt19937 rng(0);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(-1.f, 1.f);
    D.xs.resize(P * 4);
    for (int r = 0; r < n_rays; ++r) {
        const float o[3] = {0.f, 0.f, -12.5f};
        const float d[3] = {0.15f * nd(rng), 0.15f * nd(rng), 1.0f + 0.15f * nd(rng)};
        float t1 = -INFINITY, t2 = INFINITY;
        for (int a = 0; a < 3; ++a) {
            const float inv = 1.0f / d[a], lo = (-5.0f - o[a]) * inv, hi = (5.0f - o[a]) * inv;
            t1 = fmaxf(t1, fminf(lo, hi)); t2 = fminf(t2, fmaxf(lo, hi));
        }
        if (t1 > t2 || t2 <= 0) { t1 = t2 = -1.0f; } else t1 = fmaxf(t1, 0.f);
        for (int n = 0; n < N; ++n) {
            const float t = ((float)n + 0.5f) / (float)N * (t2 - t1) + t1;
            for (int a = 0; a < 3; ++a) D.xs[((int64_t)r * N + n) * 4 + a] = ((o[a] + d[a] * t) + 5.0f) / 10.0f;
        }
    }
    D.rec16.resize((size_t)kL * P * 16);
    for (int64_t i = 0; i < P; ++i) {
        const float gx = ud(rng), gy = ud(rng), gz = ud(rng);
        for (int l = 0; l < kL; ++l) {
            const float v[12] = {ud(rng), ud(rng), ud(rng), ud(rng), ud(rng), ud(rng), gx, gy, gz, D.xs[i * 4], D.xs[i * 4 + 1], D.xs[i * 4 + 2]};
            for (int k = 0; k < 12; ++k) D.rec16[((size_t)l * P + i) * 16 + k] = v[k];
        }
    }
    CK(hipMalloc(&D.d_rec16, D.rec16.size() * 4)); CK(hipMemcpy(D.d_rec16, D.rec16.data(), D.rec16.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&D.d_t1, (size_t)D.n_entries * 8)); CK(hipMalloc(&D.d_t2, (size_t)D.n_entries * 8));
    printf("points %lld\n", (long long)P);
    run_config<12>(D, only_level);
    run_config<11>(D, only_level);
    return 0;
}
