// placeholder translation unit — replaced by the log-mel kernel
#include <hip/hip_runtime.h>
#include "pcgmix_kernels.h"
extern "C" int pcgmix_logmel_f32(const float*, const int32_t*, float*, int32_t*, int, int, int, int,
                                 int, float, float, float, float, float, int, pcgmix_stream_t) {
  return hipErrorNotSupported;
}
