// wgrad_reduce.h -- fixed-order fold of the filter-gradient partial sums (shared by the f32 and bf16 kernels)
#pragma once
#include <ds_device.h>

namespace {

// dW_oihw[co][perm(ci)][tap] = scale * sum_s partial[s][tap][co][ci];  fcF > 0 applies the fc feature
// permutation ci = f*C + c  ->  c*F + f  (reference model.py:208 flatten order); scale = 1 except where the operands
// carried a loss scale (the fp16 training step: 1 / S).
// A workgroup owns 64 consecutive outputs; its W = 2^lg waves (1, 2 or 4) each sum every W-th split -- every load
// instruction of a wave is one coalesced 256-byte row -- and wave 0 adds the W sums in wave order (deterministic).  The
// 64-channel layers have 36 864 outputs and 256 splits: one thread per output was 144 workgroups each walking 256
// strided loads in sequence, 340 us for 38 MB.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *partial, float *gw, int S, int T, int Cout,
                                                           int Cin, int fcF, float scale, int lg) {
    float *red = ds_dynamic_lds();                          // [4][64]
    const long long n = (long long)T * Cout * Cin;
    const int W = 1 << lg;                                  // waves per 64 outputs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = wave >> lg, r = wave & (W - 1);         // which 64-output group of this workgroup, which split lane
    const int groups = 4 >> lg;
    for (long long g0 = (long long)blockIdx.x * groups; g0 * 64 < n; g0 += (long long)gridDim.x * groups) {
        const long long i = (g0 + sub) * 64 + lane;
        float s = 0.f;
        if (i < n)
            for (int k = r; k < S; k += W) s += partial[(size_t)k * n + i];
        __syncthreads();                                    // (the previous round's sums have been read)
        red[wave * 64 + lane] = s;
        __syncthreads();
        if (r == 0 && i < n) {
            float t = s;
            for (int m = 1; m < W; ++m) t += red[(wave + m) * 64 + lane];
            const int ci = (int)(i % Cin);
            const long long q = i / Cin;
            const int co = (int)(q % Cout), tap = (int)(q / Cout);
            int cio = ci;
            if (fcF > 0) {
                const int C = Cin / fcF;
                cio = (ci % C) * fcF + ci / C;
            }
            gw[((size_t)co * Cin + cio) * T + tap] = scale == 1.0f ? t : t * scale;
        }
    }
}

// waves per 64 outputs (log2) and grid for n outputs summed over S splits
static inline void wgrad_reduce_shape(long long n, int S, int &lg, int &grid) {
    lg = 0;
    while (lg < 2 && (1 << (lg + 1)) <= S && (n << lg) < (1ll << 20)) ++lg;      // small layers: more waves per output
    const long long per_block = 256 >> lg;
    long long g = (n + per_block - 1) / per_block;
    grid = (int)(g > 8192 ? 8192 : g);
}

}  // namespace
