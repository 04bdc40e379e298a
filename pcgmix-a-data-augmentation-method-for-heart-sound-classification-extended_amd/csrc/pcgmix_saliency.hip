// placeholder translation unit — replaced by the saliency kernels
#include <hip/hip_runtime.h>
#include "pcgmix_kernels.h"
extern "C" int pcgmix_saliency_post_f32(const float*, const int32_t*, float*, int, float, int, int,
                                        int, pcgmix_stream_t) { return hipErrorNotSupported; }
extern "C" int pcgmix_salopt_disp_f32(const float*, const int32_t*, const int32_t*, float, int,
                                      int32_t*, int, int, pcgmix_stream_t) { return hipErrorNotSupported; }
