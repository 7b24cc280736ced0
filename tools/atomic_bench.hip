// micro-benchmark of scatter-add primitives on gfx950 (build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int MODE>   // 0 lds f32 atomic, 1 lds u32 atomic, 2 lds u64 atomic, 3 lds plain RMW f32x2 (racy, throughput only), 4 lds f32 atomic 25% lanes
__global__ void __launch_bounds__(1024) lds_kernel(float* out, int iters) {
    __shared__ float acc[32768];
    for (int e = threadIdx.x; e < 32768; e += 1024) acc[e] = 0.f;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
    for (int it = 0; it < iters; ++it) {
        const uint32_t idx = rnd(s) & 16383u;
        if (MODE == 0) { atomicAdd(&acc[2 * idx], 1.0f); atomicAdd(&acc[2 * idx + 1], 2.0f); }
        if (MODE == 1) { atomicAdd((unsigned*)&acc[2 * idx], 1u); atomicAdd((unsigned*)&acc[2 * idx + 1], 2u); }
        if (MODE == 2) { atomicAdd((unsigned long long*)&acc[2 * idx], 0x0000000200000001ull); }
        if (MODE == 3) { float2 v = *(float2*)&acc[2 * idx]; v.x += 1.f; v.y += 2.f; *(float2*)&acc[2 * idx] = v; }
        if (MODE == 4) { if ((threadIdx.x & 3) == 0) { atomicAdd(&acc[2 * idx], 1.0f); atomicAdd(&acc[2 * idx + 1], 2.0f); } }
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc[5];
}

template <int SCOPE>  // 0 agent, 1 workgroup, 2 wavefront
__global__ void __launch_bounds__(256) glb_kernel(float* table, uint32_t mask, int iters) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    for (int it = 0; it < iters; ++it) {
        const uint32_t idx = rnd(s) & mask;
        if (SCOPE == 0) __hip_atomic_fetch_add(table + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (SCOPE == 1) __hip_atomic_fetch_add(table + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (SCOPE == 2) __hip_atomic_fetch_add(table + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

template <typename F> float time_ms(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}

int main() {
    float* out; CK(hipMalloc(&out, 4096 * 4));
    const int iters = 512, blocks = 256;
    const double n_lds = (double)blocks * 1024 * iters;       // updates of one (f0,f1) pair
    const char* names[5] = {"lds atomic f32 x2", "lds atomic u32 x2", "lds atomic u64 x1", "lds plain RMW f32x2", "lds atomic f32 x2 (25% lanes)"};
    float ms;
    ms = time_ms([&] { lds_kernel<0><<<blocks, 1024>>>(out, iters); }); printf("%-32s %8.3f ms  %7.1f G pair-updates/s\n", names[0], ms, n_lds / ms / 1e6);
    ms = time_ms([&] { lds_kernel<1><<<blocks, 1024>>>(out, iters); }); printf("%-32s %8.3f ms  %7.1f G pair-updates/s\n", names[1], ms, n_lds / ms / 1e6);
    ms = time_ms([&] { lds_kernel<2><<<blocks, 1024>>>(out, iters); }); printf("%-32s %8.3f ms  %7.1f G pair-updates/s\n", names[2], ms, n_lds / ms / 1e6);
    ms = time_ms([&] { lds_kernel<3><<<blocks, 1024>>>(out, iters); }); printf("%-32s %8.3f ms  %7.1f G pair-updates/s\n", names[3], ms, n_lds / ms / 1e6);
    ms = time_ms([&] { lds_kernel<4><<<blocks, 1024>>>(out, iters); }); printf("%-32s %8.3f ms  %7.1f G pair-updates/s\n", names[4], ms, n_lds / 4 / ms / 1e6);
    float* table; CK(hipMalloc(&table, (size_t)(1u << 24) * 4)); CK(hipMemset(table, 0, (size_t)(1u << 24) * 4));
    const int gb = 4096, git = 64; const double n_g = (double)gb * 256 * git;
    for (uint32_t bits : {20u, 24u}) {
        const uint32_t mask = (1u << bits) - 1u;
        ms = time_ms([&] { glb_kernel<0><<<gb, 256>>>(table, mask, git); }); printf("global atomic f32 agent scope     (2^%u floats) %8.3f ms %7.1f G/s\n", bits, ms, n_g / ms / 1e6);
        ms = time_ms([&] { glb_kernel<1><<<gb, 256>>>(table, mask, git); }); printf("global atomic f32 workgroup scope (2^%u floats) %8.3f ms %7.1f G/s\n", bits, ms, n_g / ms / 1e6);
        ms = time_ms([&] { glb_kernel<2><<<gb, 256>>>(table, mask, git); }); printf("global atomic f32 wavefront scope (2^%u floats) %8.3f ms %7.1f G/s\n", bits, ms, n_g / ms / 1e6);
    }
    return 0;
}
