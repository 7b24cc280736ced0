// Entry points whose kernels have not landed yet report LS2FM_ERR_UNSUPPORTED (callers then use the
// general autograd-composed form).  This file shrinks to nothing as the fused kernels are added.
#include "ls2fm_device.h"

extern "C" int64_t ls2fm_sdf_eval_workspace_bytes(void) { return 0; }
extern "C" int ls2fm_sdf_eval(const ls2fm_field_desc*, const ls2fm_grid_desc*, const ls2fm_params*, const float*,
                              int64_t, float*, float*, float*, void*, void*) { return LS2FM_ERR_UNSUPPORTED; }
extern "C" int ls2fm_sphere_trace(const ls2fm_field_desc*, const ls2fm_grid_desc*, const ls2fm_params*, const float*,
                                  const float*, int64_t, float, int32_t, float*, float*, float*, float*, int32_t*,
                                  void*, void*) { return LS2FM_ERR_UNSUPPORTED; }
