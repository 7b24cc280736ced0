// Does the time of a kernel with a LARGE straight-line body depend on the host process's NUMA node (profiles/r06_notes.md section 6)?
// big_kernel: ~N_BLK x 64 independent-ish v_fma instructions (N_BLK x 512 B of code), one wave per CU-slot, launched repeatedly;
// small_kernel: the same arithmetic as a loop (a few hundred bytes of code).  hipcc --offload-arch=gfx950 -O3 icache_probe.hip -o icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define F8(a) a a a a a a a a
#define BLK asm volatile(F8(F8("v_fma_f32 %0, %0, %1, %2\n\t")) : "+v"(x) : "v"(y), "v"(z));
#define B8 BLK BLK BLK BLK BLK BLK BLK BLK
__global__ void big_kernel(float* out, float y, float z) {
    float x = threadIdx.x;
    B8 B8 B8 B8 B8 B8 B8 B8 B8 B8 B8 B8 B8 B8 B8 B8       // 128 blocks x 64 instructions x 8 bytes = 64 KB of code
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
__global__ void small_kernel(float* out, float y, float z) {
    float x = threadIdx.x;
    for (int i = 0; i < 128; ++i) { BLK }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
int main() {
    float* out; hipMalloc(&out, 1024 * 128 * sizeof(float));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int which = 0; which < 2; ++which) {
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 20; ++i) { if (which) small_kernel<<<1024, 128>>>(out, 0.5f, 0.25f); else big_kernel<<<1024, 128>>>(out, 0.5f, 0.25f); }
            hipEventRecord(a);
            for (int i = 0; i < 200; ++i) { if (which) small_kernel<<<1024, 128>>>(out, 0.5f, 0.25f); else big_kernel<<<1024, 128>>>(out, 0.5f, 0.25f); }
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%s  %.2f us per launch\n", which ? "small (loop)      " : "big (64 KB of code)", ms * 1000.f / 200.f);
        }
    }
    return 0;
}
