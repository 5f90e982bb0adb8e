// conv_mfma_bf16_k3x3g.hip -- instantiations of conv_mfma_bf16_kernel for 3x3 taps, split-operand bf16x3
// arithmetic, with the BatchNorm-backward epilogue (BNB): the stride-1 data gradient of a 3x3 layer fused with the
// first half of the backward of the BatchNorm + clipped-ReLU layer below it (ds_conv_dgrad_bnbwd_bf16).
#define DS_BF16_KERNEL_TU
#include "conv_mfma_bf16_kernel.h"

void ds_bf16_launch_k3x3g(const PlanB &pl, void *stream) { launch_b<3, true, true>(pl, stream); }
