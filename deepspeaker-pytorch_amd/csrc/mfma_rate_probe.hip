// mfma_rate_probe.hip -- MEASUREMENT ONLY: what the matrix cores of THIS chip deliver when nothing else limits them.
// Every SIMD issues independent v_mfma_f32_32x32x16_f16 (or _bf16) back to back from registers, operands with unrelated
// pseudo-random bit patterns (consecutive MFMAs toggle the whole multiplier array, as real activations and filters do).
// The rate is the power- / clock-limited ceiling the convolution kernels are up against: the nominal 2.5 PFLOP/s assumes
// 2.4 GHz sustained, and the chip clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back").  bench.py times a few
// launches of it with HIP events and reports the figure as roofline.mfma_register_only_tflops, measured in the same
// process, on the same box, minutes apart from the kernels it is compared with.  Not on any product path.
#include <ds_device.h>
#include "ds_common.h"

namespace {

template <bool BF16>
__global__ void __launch_bounds__(256) mfma_rate_kernel(float *sink, int iters) {
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    f16x8 ha[4], hb[4];
    bf16x8 ba[4], bb[4];
    unsigned lcg = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            lcg = lcg * 1664525u + 1013904223u;
            const float ra = ((lcg >> 8) & 0xFFFF) * (1.0f / 4096.0f) - 8.0f;
            lcg = lcg * 1664525u + 1013904223u;
            const float rb = ((lcg >> 8) & 0xFFFF) * (1.0f / 65536.0f) - 0.5f;
            ha[a][i] = (_Float16)ra;
            hb[a][i] = (_Float16)rb;
            ba[a][i] = (__bf16)ra;
            bb[a][i] = (__bf16)rb;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (BF16) acc[a] = ds_mfma_32x32x16_bf16(ba[a], bb[a], acc[a]);
            else acc[a] = ds_mfma_32x32x16_f16(ha[a], hb[a], acc[a]);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) sink[0] = s;              // keeps the loop alive without a store on the timed path
}

// The same loop with operands taken from REAL tensors: the filter fragments (A) from a packed fp16 filter bank, the pixel
// fragments (B) from an fp16 activation tensor of the forward (post clipped-ReLU: about half zeros, small values).  Random
// bits are the worst case for the multiplier array's switching power; what the chip sustains on the operand values the
// convolutions actually see is the ceiling they are up against (round-5 review: "attainable" must be a number).
__global__ void __launch_bounds__(256) mfma_rate_data_kernel(const _Float16 *a_src, long long n_a8, const _Float16 *b_src,
                                                             long long n_b8, float *sink, int iters) {
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    f16x8 ha[4], hb[4];
    // fragment a of this lane: 8 consecutive halfs (one k-step's worth of one row / one pixel's 8 channels), at an index
    // that differs per lane, wave, workgroup and fragment
    unsigned long long h = 0x9E3779B97F4A7C15ull * (unsigned long long)(blockIdx.x * 256 + threadIdx.x + 1);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        h = h * 6364136223846793005ull + 1442695040888963407ull;
        ha[a] = *(const f16x8 *)(a_src + 8 * (long long)((h >> 11) % (unsigned long long)n_a8));
        h = h * 6364136223846793005ull + 1442695040888963407ull;
        hb[a] = *(const f16x8 *)(b_src + 8 * (long long)((h >> 11) % (unsigned long long)n_b8));
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = ds_mfma_32x32x16_f16(ha[a], hb[a], acc[a]);
    }
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) sink[0] = s;
}

}  // namespace

// One launch: 2 four-wave workgroups per CU (two waves per SIMD), `iters` x 4 MFMAs per wave.  Returns (through
// *flop_out) the floating-point operations the launch performs, so that the caller divides by its own event time.
extern "C" int ds_mfma_rate_probe(int bf16, int iters, float *sink, double *flop_out, void *stream) {
    DS_REQUIRE(sink && flop_out, DS_ERR_NULL);
    DS_REQUIRE(iters > 0, DS_ERR_BAD_SHAPE);
    const int grid = 2 * ds_cu_count();
    if (bf16) DS_LAUNCH(mfma_rate_kernel<true>, grid, 256, 0, stream, sink, iters);
    else DS_LAUNCH(mfma_rate_kernel<false>, grid, 256, 0, stream, sink, iters);
    *flop_out = (double)grid * 4.0 * (double)iters * 4.0 * (2.0 * 32 * 32 * 16);
    return ds_last_launch_error();
}

// The same launch with fp16 operands read from real tensors: `a_f16` (n_a halfs, e.g. a packed filter bank) supplies the
// A fragments, `b_f16` (n_b halfs, e.g. an activation tensor of the forward) the B fragments; both 16-byte aligned.
extern "C" int ds_mfma_rate_probe_data(const void *a_f16, long long n_a, const void *b_f16, long long n_b, int iters,
                                       float *sink, double *flop_out, void *stream) {
    DS_REQUIRE(a_f16 && b_f16 && sink && flop_out, DS_ERR_NULL);
    DS_REQUIRE(iters > 0 && n_a >= 8 && n_b >= 8, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(a_f16) && DS_ALIGNED16(b_f16), DS_ERR_ALIGNMENT);
    const int grid = 2 * ds_cu_count();
    DS_LAUNCH(mfma_rate_data_kernel, grid, 256, 0, stream, (const _Float16 *)a_f16, n_a / 8, (const _Float16 *)b_f16,
              n_b / 8, sink, iters);
    *flop_out = (double)grid * 4.0 * (double)iters * 4.0 * (2.0 * 32 * 32 * 16);
    return ds_last_launch_error();
}
