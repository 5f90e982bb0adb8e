// wgrad_reduce.h -- fixed-order fold of the filter-gradient partial sums (shared by the f32 and bf16 kernels)
#pragma once
#include <ds_device.h>

namespace {

// dW_oihw[co][perm(ci)][tap] = scale * sum_s partial[s][tap][co][ci];  fcF > 0 applies the fc feature
// permutation ci = f*C + c  ->  c*F + f  (reference model.py:208 flatten order); scale = 1 except where the operands
// carried a loss scale (the fp16 training step: 1 / S)
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *partial, float *gw, int S, int T, int Cout,
                                                           int Cin, int fcF, float scale) {
    const long long n = (long long)T * Cout * Cin;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += partial[(size_t)k * n + i];
        const int ci = (int)(i % Cin);
        const long long r = i / Cin;
        const int co = (int)(r % Cout), tap = (int)(r / Cout);
        int cio = ci;
        if (fcF > 0) {
            const int C = Cin / fcF;
            cio = (ci % C) * fcF + ci / C;
        }
        gw[((size_t)co * Cin + cio) * T + tap] = scale == 1.0f ? s : s * scale;
    }
}

}  // namespace
