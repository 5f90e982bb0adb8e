// conv_mfma_f16_k3o2.hip -- the 3x3 instantiations of the fp16 convolution kernel built for two wavefronts per SIMD
// (128x64 register tiles, single pixel tile, no register prefetch: the shallow contractions of stages 1-2), a
// translation unit of their own so that the kernel family compiles in parallel (see conv_mfma_f16_kernel.h)
#define DS_F16_KERNEL_TU
#include "conv_mfma_f16_kernel.h"

void ds_f16_launch_k3o2(const PlanH &pl, void *stream) { launch_occ2_h<3, 32>(pl, stream); }
