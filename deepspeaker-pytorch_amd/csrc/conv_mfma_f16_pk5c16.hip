// conv_mfma_f16_pk5c16.hip -- the 5x5 instantiations with 16-channel chunks of the PERSISTENT fp16 convolution kernel
// (conv_mfma_f16_pkernel.h), a translation unit of their own so that the kernel family compiles in parallel
#define DS_F16_PKERNEL_TU
#include "conv_mfma_f16_pkernel.h"

void ds_f16_launch_pk5c16(const PlanH &pl, void *stream) { launch_p<5, 16>(pl, stream); }
