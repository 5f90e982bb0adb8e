// conv_mfma_f16_pk3.hip -- the 3x3 instantiations of the PERSISTENT fp16 convolution kernel
// (conv_mfma_f16_pkernel.h), a translation unit of their own so that the kernel family compiles in parallel
#define DS_F16_PKERNEL_TU
#include "conv_mfma_f16_pkernel.h"

void ds_f16_launch_pk3(const PlanH &pl, void *stream) { launch_p<3>(pl, stream); }
