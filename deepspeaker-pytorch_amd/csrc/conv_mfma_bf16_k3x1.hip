// conv_mfma_bf16_k3x1.hip -- instantiations of conv_mfma_bf16_kernel for 3x3 taps, plain bf16
// arithmetic (one translation unit per combination so that they compile in parallel).
#define DS_BF16_KERNEL_TU
#include "conv_mfma_bf16_kernel.h"

void ds_bf16_launch_k3x1(const PlanB &pl, void *stream) { launch_b<3, false>(pl, stream); }
