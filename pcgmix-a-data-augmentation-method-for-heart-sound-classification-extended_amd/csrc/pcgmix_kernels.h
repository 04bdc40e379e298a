// Internal header shared by the kernel translation units of libpcgmix_hip.so.
#ifndef PCGMIX_KERNELS_H
#define PCGMIX_KERNELS_H
#include "pcgmix_hip.h"   // public C ABI (include/)
#endif
