// conv_mfma_f16_k5o2.hip -- the 5x5 instantiations of the fp16 convolution kernel built for two wavefronts per SIMD
// (128x64 register tiles, single pixel tile, no register prefetch: the shallow contractions of stages 1-2), a
// translation unit of their own so that the kernel family compiles in parallel (see conv_mfma_f16_kernel.h)
#define DS_F16_KERNEL_TU
#include "conv_mfma_f16_kernel.h"

void ds_f16_launch_k5o2(const PlanH &pl, void *stream) { if (pl.ck == 16) launch_occ2_h<5, 16>(pl, stream); else launch_occ2_h<5, 32>(pl, stream); }
