// mfma_lds_ratio.hip -- how much of the register-only MFMA rate (tools/mfma_peak.hip) survives when one operand comes
// from LDS, as in the convolution kernels: every PER MFMAs a wave issues one conflict-free ds_read_b128 (a pixel
// fragment: 1 KiB per wave) that feeds the following PER MFMAs.  PER = 2 is the ratio of conv_mfma_f16_kernel (a
// pixel fragment meets NSUB = 2 filter fragments), PER = 4 / 8 what wider register tiles would give.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_lds_ratio tools/mfma_lds_ratio.hip && /tmp/mfma_lds_ratio
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PER, int WAVES_PER_SIMD>
__global__ void __launch_bounds__(256) loop(float *out, int iters) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    unsigned lcg = 0x9E3779B9u * (tid + 1) + blockIdx.x;
    for (int i = tid; i < 32768 / 2; i += 256) {                    // 32 KiB of pseudo-random fp16
        lcg = lcg * 1664525u + 1013904223u;
        ((_Float16 *)lds)[i] = (_Float16)(((lcg >> 8) & 0xFFFF) * (1.0f / 65536.0f) - 0.5f);
    }
    f16x8 w[4];
    for (int a = 0; a < 4; ++a)
        for (int i = 0; i < 8; ++i) {
            lcg = lcg * 1664525u + 1013904223u;
            w[a][i] = (_Float16)(((lcg >> 8) & 0xFFFF) * (1.0f / 4096.0f) - 8.0f);
        }
    __syncthreads();
    f32x16 acc[PER];
    for (int a = 0; a < PER; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    // 80-byte records, 16 consecutive records per 16-lane service group: conflict-free (as the kernels' pixel tiles)
    const char *base = lds + (lane & 31) * 80 + (lane >> 5) * 16;
    f16x8 f0 = *(const f16x8 *)base, f1;
    for (int it = 0; it < iters; it += 2) {                          // (two copies: register arrays want static indices)
        f1 = *(const f16x8 *)(base + ((it + 1) & 7) * 2560);         // the next fragment travels during these MFMAs
#pragma unroll
        for (int a = 0; a < PER; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[a & 3], f0, acc[a], 0, 0, 0);
        f0 = *(const f16x8 *)(base + ((it + 2) & 7) * 2560);
#pragma unroll
        for (int a = 0; a < PER; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[a & 3], f1, acc[a], 0, 0, 0);
    }
    float s = 0.0f;
    for (int a = 0; a < PER; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[0] = s;
}

template <int PER, int WPS>
static void run() {
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * WPS, iters = 160000 / PER;              // (even)
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((loop<PER, WPS>), dim3(grid), dim3(256), 32768, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.0f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)grid * 4 * iters * PER * 2.0 * 32 * 32 * 16;
        printf("%d MFMAs per ds_read_b128, %d wave(s) per SIMD   run %d: %7.3f ms  %8.1f TFLOP/s\n", PER, WPS, rep, ms, flops / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    run<2, 2>();
    run<4, 2>();
    run<8, 2>();
    run<2, 1>();
    run<4, 1>();
    return 0;
}
