// conv_mfma_bf16_k5x1.hip -- instantiations of conv_mfma_bf16_kernel for 5x5 taps, plain bf16
// arithmetic (one translation unit per combination so that they compile in parallel).
#define DS_BF16_KERNEL_TU
#include "conv_mfma_bf16_kernel.h"

void ds_bf16_launch_k5x1(const PlanB &pl, void *stream) { launch_b<5, false>(pl, stream); }
