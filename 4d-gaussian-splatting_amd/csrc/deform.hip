// placeholder until the deformation kernels land
#include "fdgs_common.h"
extern "C" int fdgs_deform_fwd(const fdgs_deform_fwd_args*, void*) { return FDGS_EINVAL; }
extern "C" size_t fdgs_deform_bwd_workspace_bytes(int32_t) { return 0; }
extern "C" int fdgs_deform_bwd(const fdgs_deform_bwd_args*, void*) { return FDGS_EINVAL; }
