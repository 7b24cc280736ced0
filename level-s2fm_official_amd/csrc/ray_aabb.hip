// Ray / axis-aligned-box slab test (replaces vren.ray_aabb_intersect, utils/custom_functions.py:28-31).
// One thread per ray walks the voxel list and keeps the `max_hits` nearest hits sorted by near t in its
// own output row (the reference only ever passes ONE box and max_hits = 1: Renderer.py:178, SDF.py:120).
#include "ls2fm_device.h"

namespace {

__global__ void __launch_bounds__(256)
ray_aabb_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, const float* __restrict__ center,
                const float* __restrict__ half_size, int64_t n_rays, int n_voxels, int max_hits,
                int32_t* __restrict__ hits_cnt, float* __restrict__ hits_t, int64_t* __restrict__ hits_idx) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float o[3] = {rays_o[r * 3], rays_o[r * 3 + 1], rays_o[r * 3 + 2]};
    const float d[3] = {rays_d[r * 3], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
    float* t_row = hits_t + r * max_hits * 2;
    int64_t* i_row = hits_idx + r * max_hits;
    for (int k = 0; k < max_hits; ++k) { t_row[2 * k] = -1.0f; t_row[2 * k + 1] = -1.0f; i_row[k] = -1; }
    int cnt = 0;
    for (int v = 0; v < n_voxels; ++v) {
        const float c[3] = {center[v * 3], center[v * 3 + 1], center[v * 3 + 2]};
        const float h[3] = {half_size[v * 3], half_size[v * 3 + 1], half_size[v * 3 + 2]};
        float tn, tf; bool hit;
        ray_box(o, d, c, h, tn, tf, hit);
        if (!hit) continue;
        // insertion into the sorted prefix (kept entries: min(cnt, max_hits))
        int kept = cnt < max_hits ? cnt : max_hits;
        int pos = kept;
        while (pos > 0 && t_row[2 * (pos - 1)] > tn) --pos;
        if (pos < max_hits) {
            for (int k = (kept < max_hits ? kept : max_hits - 1); k > pos; --k) {
                t_row[2 * k] = t_row[2 * (k - 1)]; t_row[2 * k + 1] = t_row[2 * (k - 1) + 1]; i_row[k] = i_row[k - 1];
            }
            t_row[2 * pos] = tn; t_row[2 * pos + 1] = tf; i_row[pos] = v;
        }
        ++cnt;
    }
    hits_cnt[r] = cnt;
}

}  // namespace

extern "C" int ls2fm_ray_aabb_intersect(const float* rays_o, const float* rays_d, const float* center,
                                        const float* half_size, int64_t n_rays, int32_t n_voxels, int32_t max_hits,
                                        int32_t* hits_cnt, float* hits_t, int64_t* hits_voxel_idx, void* stream) {
    LS2FM_CHECK_ARG(n_rays >= 0 && n_voxels >= 1 && max_hits >= 1);
    if (n_rays == 0) return LS2FM_OK;
    LS2FM_CHECK_ARG(rays_o && rays_d && center && half_size && hits_cnt && hits_t && hits_voxel_idx);
    const unsigned blocks = (unsigned)((n_rays + 255) / 256);
    ray_aabb_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(rays_o, rays_d, center, half_size, n_rays, n_voxels,
                                                            max_hits, hits_cnt, hits_t, hits_voxel_idx);
    return ls2fm_launch_status();
}

extern "C" int ls2fm_abi_version(void) { return LS2FM_ABI_VERSION; }

extern "C" const char* ls2fm_status_string(int status) {
    switch (status) {
        case LS2FM_OK: return "ok";
        case LS2FM_ERR_INVALID_ARGUMENT: return "invalid argument";
        case LS2FM_ERR_UNSUPPORTED: return "unsupported configuration";
        case LS2FM_ERR_LAUNCH: return "HIP launch / runtime error";
        case LS2FM_ERR_WORKSPACE: return "workspace missing or too small";
        case LS2FM_ERR_STARVED: return "an earlier call's in-launch hand-off was starved (ls2fm_async_error); its gradients are poisoned";
        default: return "unknown status";
    }
}
