// conv_mfma_f16_k5c16.hip -- the 5x5 instantiations of the fp16 convolution kernel with 16-channel chunks
// (double-buffered pixel tile), a translation unit of their own so that the kernel family compiles in parallel
// (see conv_mfma_f16_kernel.h)
#define DS_F16_KERNEL_TU
#include "conv_mfma_f16_kernel.h"

void ds_f16_launch_k5c16(const PlanH &pl, void *stream) { launch_h<5, true, 16>(pl, stream); }
