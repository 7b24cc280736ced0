/* Oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): plain-C restatement of the tiny-cuda-nn 1.7
 * Grid/Hash/Linear lookup, independent of oracle/hashgrid.py, used as the hash-INDEX authority
 * (it calls the real fmaf/floorf/exp2f/log2f).  PARITY UNPINNED by the reference: tcnn's source is
 * not under /root/reference (call sites models/base.py:17,37); restated from the published algorithm
 * (tcnn grid.h: grid_scale, grid_resolution, grid_index, kernel_grid).
 *
 * Build: make -C oracle   ->  oracle/_build/libls2fm_oracle.so  (loaded with ctypes by tests only)
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define PRIME_Y 2654435761u
#define PRIME_Z 805459861u

/* tcnn grid_scale / grid_resolution + GridEncodingTemplated offset table (3-D positions). */
int ls2fm_ref_level_table(int n_levels, int base_resolution, float per_level_scale, int log2_hashmap_size,
                          float* scale, uint32_t* res, uint32_t* size, uint32_t* offset, uint8_t* hashed)
{
    /* correctly rounded log2 / exp2 (see oracle/hashgrid.py: libm implementations differ in the last bit) */
    const float log2b = (float)log2((double)per_level_scale);
    const uint32_t max_params = 0xFFFFFFFFu / 2u;
    uint32_t off = 0;
    for (int l = 0; l < n_levels; ++l) {
        const float e = (float)exp2((double)((float)l * log2b));
        const float s = e * (float)base_resolution - 1.0f;
        const uint32_t r = (uint32_t)ceilf(s) + 1u;
        uint32_t n = (powf((float)r, 3.0f) > (float)max_params) ? max_params : r * r * r;
        n = (n + 7u) / 8u * 8u;
        const uint32_t cap = 1u << log2_hashmap_size;
        if (n > cap) n = cap;
        uint32_t stride = 1;
        for (int d = 0; d < 3 && stride <= n; ++d) stride *= r;
        scale[l] = s; res[l] = r; size[l] = n; offset[l] = off; hashed[l] = (uint8_t)(n < stride);
        off += n;
    }
    offset[n_levels] = off;
    return 0;
}

static inline uint32_t corner_index(const uint32_t c[3], uint32_t res, uint32_t size)
{
    uint32_t stride = 1, index = 0;
    for (int d = 0; d < 3 && stride <= size; ++d) { index += c[d] * stride; stride *= res; }
    if (size < stride) index = c[0] ^ (c[1] * PRIME_Y) ^ (c[2] * PRIME_Z);
    return index % size;
}

/* indices[M][L][8] (level-local entry index), weights[M][L][8] (optional) */
void ls2fm_ref_grid_indices(const float* x, int64_t n_points, int n_levels, const float* scale,
                            const uint32_t* res, const uint32_t* size, uint32_t* indices, float* weights)
{
    for (int64_t i = 0; i < n_points; ++i)
        for (int l = 0; l < n_levels; ++l) {
            uint32_t cell[3]; float w[3];
            for (int d = 0; d < 3; ++d) {
                const float pos = fmaf(scale[l], x[i * 3 + d], 0.5f);
                const float fl = floorf(pos);
                cell[d] = (uint32_t)(int32_t)fl;
                w[d] = pos - fl;
            }
            for (int corner = 0; corner < 8; ++corner) {
                uint32_t c[3]; float wt = 1.0f;
                for (int d = 0; d < 3; ++d) {
                    if (corner & (1 << d)) { wt *= w[d]; c[d] = cell[d] + 1u; }
                    else { wt *= 1.0f - w[d]; c[d] = cell[d]; }
                }
                const size_t o = ((size_t)i * n_levels + l) * 8 + corner;
                indices[o] = corner_index(c, res[l], size[l]);
                if (weights) weights[o] = wt;
            }
        }
}

/* out[M][L*F]; dy_dx[M][L*F][3] optional (d out / d x, includes the level scale) */
void ls2fm_ref_grid_encode(const float* x, int64_t n_points, const float* params, int n_levels, int n_feat,
                           const float* scale, const uint32_t* res, const uint32_t* size,
                           const uint32_t* offset, float* out, float* dy_dx)
{
    for (int64_t i = 0; i < n_points; ++i)
        for (int l = 0; l < n_levels; ++l) {
            uint32_t cell[3]; float w[3];
            for (int d = 0; d < 3; ++d) {
                const float pos = fmaf(scale[l], x[i * 3 + d], 0.5f);
                const float fl = floorf(pos);
                cell[d] = (uint32_t)(int32_t)fl;
                w[d] = pos - fl;
            }
            const float* level = params + (size_t)offset[l] * n_feat;
            for (int f = 0; f < n_feat; ++f) {
                float acc = 0.0f, g[3] = {0.0f, 0.0f, 0.0f};
                for (int corner = 0; corner < 8; ++corner) {
                    uint32_t c[3]; float wt = 1.0f;
                    for (int d = 0; d < 3; ++d) {
                        if (corner & (1 << d)) { wt *= w[d]; c[d] = cell[d] + 1u; }
                        else { wt *= 1.0f - w[d]; c[d] = cell[d]; }
                    }
                    const float v = level[(size_t)corner_index(c, res[l], size[l]) * n_feat + f];
                    acc += wt * v;
                    for (int gd = 0; gd < 3; ++gd) {           /* d wt / d w[gd] */
                        float dw = (corner & (1 << gd)) ? 1.0f : -1.0f;
                        for (int d = 0; d < 3; ++d)
                            if (d != gd) dw *= (corner & (1 << d)) ? w[d] : 1.0f - w[d];
                        g[gd] += dw * v;
                    }
                }
                out[(size_t)i * n_levels * n_feat + l * n_feat + f] = acc;
                if (dy_dx)
                    for (int gd = 0; gd < 3; ++gd)
                        dy_dx[((size_t)i * n_levels * n_feat + l * n_feat + f) * 3 + gd] = scale[l] * g[gd];
            }
        }
}
