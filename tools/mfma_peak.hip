// mfma_peak.hip -- what the matrix cores of THIS chip deliver when nothing else limits them: every SIMD issues
// independent v_mfma_f32_32x32x16_{f16,bf16} (and v_mfma_f32_32x32x2_f32) back to back from registers for about a
// millisecond.  The figure is the power- / clock-limited ceiling the convolution kernels are up against (the nominal
// 2.5 PFLOP/s assumes 2.4 GHz sustained).  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND, bool RANDOM>
__global__ void __launch_bounds__(256) mfma_loop(float *out, int iters) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    const float seed = (float)(threadIdx.x & 7) * 0.125f;
    // four operand pairs per lane; RANDOM: unrelated pseudo-random bit patterns (consecutive MFMAs toggle the whole
    // multiplier array, as real activations and filters do), else smooth values of one magnitude
    f16x8 ha[4], hb[4];
    bf16x8 ba[4], bb[4];
    unsigned lcg = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 8; ++i) {
            lcg = lcg * 1664525u + 1013904223u;
            const float ra = RANDOM ? ((lcg >> 8) & 0xFFFF) * (1.0f / 4096.0f) - 8.0f : seed + i * 0.01f;
            lcg = lcg * 1664525u + 1013904223u;
            const float rb = RANDOM ? ((lcg >> 8) & 0xFFFF) * (1.0f / 65536.0f) - 0.5f : 1.0f - seed;
            ha[a][i] = (_Float16)ra;
            hb[a][i] = (_Float16)rb;
            ba[a][i] = (__bf16)ra;
            bb[a][i] = (__bf16)rb;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (KIND == 0) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha[a], hb[a], acc[a], 0, 0, 0);
            else if (KIND == 1) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba[a], bb[a], acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)ha[a][0], (float)hb[a][0], acc[a], 0, 0, 0);
        }
    }
    float s = 0.0f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[0] = s;              // keeps the loop alive without a store on the timed path
}

template <int KIND, bool RANDOM>
static void run(const char *name, double flop_per_mfma) {
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * 2, iters = 40000;      // two 4-wave workgroups per CU: two waves per SIMD
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_loop<KIND, RANDOM>), dim3(grid), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.0f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 /*waves*/ * iters * 4 /*mfma*/ * flop_per_mfma;
        printf("%-32s run %d: %7.3f ms  %8.1f TFLOP/s\n", name, rep, ms, flops / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    run<0, false>("32x32x16_f16, smooth operands", 2.0 * 32 * 32 * 16);
    run<0, true>("32x32x16_f16, random operands", 2.0 * 32 * 32 * 16);
    run<1, false>("32x32x16_bf16, smooth operands", 2.0 * 32 * 32 * 16);
    run<1, true>("32x32x16_bf16, random operands", 2.0 * 32 * 32 * 16);
    run<2, false>("32x32x2_f32", 2.0 * 32 * 32 * 2);
    return 0;
}
