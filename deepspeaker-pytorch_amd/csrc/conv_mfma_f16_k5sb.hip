// conv_mfma_f16_k5sb.hip -- the 5x5 instantiations of the fp16 convolution kernel (single pixel tile),
// a translation unit of their own so that the kernel family compiles in parallel (see conv_mfma_f16_kernel.h)
#define DS_F16_KERNEL_TU
#include "conv_mfma_f16_kernel.h"

void ds_f16_launch_k5sb(const PlanH &pl, void *stream) { launch_h<5, false>(pl, stream); }
