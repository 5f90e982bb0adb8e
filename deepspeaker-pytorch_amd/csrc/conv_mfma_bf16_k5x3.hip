// conv_mfma_bf16_k5x3.hip -- instantiations of conv_mfma_bf16_kernel for 5x5 taps, split-operand bf16x3
// arithmetic (one translation unit per combination so that they compile in parallel).
#define DS_BF16_KERNEL_TU
#include "conv_mfma_bf16_kernel.h"

void ds_bf16_launch_k5x3(const PlanB &pl, void *stream) { launch_b<5, true>(pl, stream); }
