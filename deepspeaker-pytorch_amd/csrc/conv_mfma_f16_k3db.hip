// conv_mfma_f16_k3db.hip -- the 3x3 instantiations of the fp16 convolution kernel (double-buffered pixel tile),
// a translation unit of their own so that the kernel family compiles in parallel (see conv_mfma_f16_kernel.h)
#define DS_F16_KERNEL_TU
#include "conv_mfma_f16_kernel.h"

void ds_f16_launch_k3db(const PlanH &pl, void *stream) { launch_h<3, true>(pl, stream); }
