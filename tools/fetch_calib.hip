// Calibration of rocprofv3's FETCH_SIZE (and TCC hit / miss) on gfx950 for the access patterns of this library -- the guide's
// "FETCH_SIZE reports half" correction is stated for wide coalesced streaming reads; the gather pass reads 8- / 16-byte pieces of
// scattered 128-byte lines (profiles/r06_pmc_traffic.json: `uncalibrated there`).  Each kernel touches a KNOWN set of lines exactly once:
//   stream16   n threads read consecutive 16-byte pieces of a 1 GiB buffer (coalesced)         useful = lines = 1 GiB
//   gather16   thread i reads 16 bytes of line (i * odd) mod 2^23 (every 128-byte line once)    useful 128 MiB, 8 Mi distinct lines
//   gather8    the same with 8 bytes                                                            useful  64 MiB, 8 Mi distinct lines
//   gather16x2 two 16-byte pieces of each line, 64 bytes apart (both 64-byte halves touched)    useful 256 MiB
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- tools/fetch_calib ;  python tools/pmc_summary.py out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr uint64_t kBytes = 1ull << 30;
constexpr uint32_t kLines = (uint32_t)(kBytes / 128);

__global__ void stream16(const float4* __restrict__ p, float* __restrict__ out, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = p[i];
    if (v.x == 1234.5f) out[0] = v.y + v.z + v.w;
}
template <int BYTES, int PIECES>
__global__ void gather(const char* __restrict__ p, float* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= kLines) return;
    const uint32_t line = (i * 2654435761u) & (kLines - 1u);          // odd multiplier: a bijection of the lines
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < PIECES; ++q) {
        const char* a = p + (uint64_t)line * 128u + 64u * q + 16u;
        if (BYTES == 16) { const float4 v = *reinterpret_cast<const float4*>(a); acc += v.x + v.y + v.z + v.w; }
        else { const float2 v = *reinterpret_cast<const float2*>(a); acc += v.x + v.y; }
    }
    if (acc == 1234.5f) out[0] = acc;
}

int main() {
    char* buf; float* out;
    if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, kBytes);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        const uint64_t n16 = kBytes / 16;
        stream16<<<(unsigned)(n16 / 256), 256>>>(reinterpret_cast<const float4*>(buf), out, n16);
        gather<16, 1><<<kLines / 256, 256>>>(buf, out);
        gather<8, 1><<<kLines / 256, 256>>>(buf, out);
        gather<16, 2><<<kLines / 256, 256>>>(buf, out);
        (void)hipDeviceSynchronize();
    }
    printf("done: buffer %llu bytes, %u lines\n", (unsigned long long)kBytes, kLines);
    return 0;
}
