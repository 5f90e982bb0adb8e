// wgrad_reduce.h -- fixed-order fold of the filter-gradient partial sums (shared by the f32 and bf16 kernels)
#pragma once
#include <ds_device.h>

namespace {

// dW_oihw[co][perm(ci)][tap] = scale * sum_s partial[s][tap][co][ci];  fcF > 0 applies the fc feature
// permutation ci = f*C + c  ->  c*F + f  (reference model.py:208 flatten order); scale = 1 except where the operands
// carried a loss scale (the fp16 training step: 1 / S).
// L = 2^lg lanes share one output: lane r sums the splits r, r + L, ... and the L sums are folded by a fixed xor tree
// (deterministic).  The 64-channel layers have 36 864 outputs and 256 splits: one thread per output was 144 workgroups
// each walking 256 strided loads in sequence -- 340 us for 38 MB.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *partial, float *gw, int S, int T, int Cout,
                                                           int Cin, int fcF, float scale, int lg) {
    const long long n = (long long)T * Cout * Cin;
    const int L = 1 << lg, r = (int)(threadIdx.x & (L - 1));
    const long long per_block = 256 >> lg;
    for (long long i0 = (long long)blockIdx.x * per_block; i0 < n; i0 += (long long)gridDim.x * per_block) {
        const long long i = i0 + (threadIdx.x >> lg);
        float s = 0.f;
        if (i < n)
            for (int k = r; k < S; k += L) s += partial[(size_t)k * n + i];
        for (int m = 1; m < L; m <<= 1) s += ds_shfl_xor(s, m);
        if (i < n && r == 0) {
            const int ci = (int)(i % Cin);
            const long long q = i / Cin;
            const int co = (int)(q % Cout), tap = (int)(q / Cout);
            int cio = ci;
            if (fcF > 0) {
                const int C = Cin / fcF;
                cio = (ci % C) * fcF + ci / C;
            }
            gw[((size_t)co * Cin + cio) * T + tap] = scale == 1.0f ? s : s * scale;
        }
    }
}

// lanes per output (log2) and grid for n outputs summed over S splits
static inline void wgrad_reduce_shape(long long n, int S, int &lg, int &grid) {
    lg = 0;
    while (lg < 4 && (1 << (lg + 1)) <= S && (n << lg) < (1ll << 21)) ++lg;      // until ~2 M threads or 16 lanes
    const long long per_block = 256 >> lg;
    long long g = (n + per_block - 1) / per_block;
    grid = (int)(g > 8192 ? 8192 : g);
}

}  // namespace
