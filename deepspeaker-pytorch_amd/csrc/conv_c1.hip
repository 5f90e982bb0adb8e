// conv_c1.hip -- the network's first convolution: 5x5, stride 2, pad 2, ONE input channel
// -> 64 channels (reference model.py:93, called :187), fused with the BatchNorm affine and
// clipped ReLU that follow it (model.py:188-189).
//
// K = 25 is far too shallow for the matrix cores and the layer is bound by writing its
// [B,T/2,32,64] output (AI ~ 12 FLOP/B in fp32), so this is a VALU kernel built around the
// store: 16 lanes own one output pixel (4 channels each -> one coalesced 256-byte row of
// the channels-last output), the 100 filter taps a lane needs live in registers for the
// whole block, and the input halo tile is staged once in LDS and read by broadcast.
#include <ds_device.h>
#include "ds_common.h"

namespace {

constexpr int C1_RT = 8;          // output rows per workgroup
constexpr int C1_COUT = 64;

struct Conv1K {
    const float *x, *w, *scale, *shift;
    float *y, *stats;
    int H, W, Ho, Wo;
    int tiles_per_img;
    int cols_in;
    int flags;
};

__global__ void __launch_bounds__(256) conv5x5s2_c1_kernel(const Conv1K p) {
    float *lds = ds_dynamic_lds();
    const int tid = threadIdx.x;
    const int cg = tid & 15, slot = tid >> 4;
    const int b = blockIdx.x / p.tiles_per_img;
    const int r0 = (blockIdx.x - b * p.tiles_per_img) * C1_RT;
    constexpr int ROWS_IN = 2 * (C1_RT - 1) + 5;

    // stage the zero-padded input tile
    const int n_in = ROWS_IN * p.cols_in;
    const float *xb = p.x + (size_t)b * p.H * p.W;
    for (int i = tid; i < n_in; i += 256) {
        const int rr = i / p.cols_in, cc = i - rr * p.cols_in;
        const int h = 2 * r0 - 2 + rr, w = cc - 2;
        lds[i] = (h >= 0 && h < p.H && w >= 0 && w < p.W) ? xb[(size_t)h * p.W + w] : 0.0f;
    }
    float *red = lds + n_in;          // [16 slots][64][2] statistics scratch

    // this lane's 4 channels of all 25 taps
    f32x4 wt[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) wt[t] = *(const f32x4 *)(p.w + t * C1_COUT + cg * 4);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (p.flags & DS_EPI_AFFINE) {
        sc = *(const f32x4 *)(p.scale + cg * 4);
        sh = *(const f32x4 *)(p.shift + cg * 4);
    }
    __syncthreads();

    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    const int n_pix = C1_RT * p.Wo;
    for (int pix = slot; pix < n_pix; pix += 16) {
        const int r = pix / p.Wo, c = pix - r * p.Wo;
        if (r0 + r >= p.Ho) break;
        const float *in = lds + (2 * r) * p.cols_in + 2 * c;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 5; ++kh)
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
                const float v = in[kh * p.cols_in + kw];
                acc += v * wt[kh * 5 + kw];
            }
        s1 += acc;
        s2 += acc * acc;
        f32x4 o = acc;
        if (p.flags & DS_EPI_AFFINE) o = o * sc + sh;
        if (p.flags & DS_EPI_CLIP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fminf(fmaxf(o[j], 0.0f), 20.0f);
        }
        *(f32x4 *)(p.y + (((size_t)b * p.Ho + r0 + r) * p.Wo + c) * C1_COUT + cg * 4) = o;
    }
    if (p.flags & DS_EPI_STATS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[(slot * C1_COUT + cg * 4 + j) * 2 + 0] = s1[j];
            red[(slot * C1_COUT + cg * 4 + j) * 2 + 1] = s2[j];
        }
        __syncthreads();
        if (tid < C1_COUT) {
            float a1 = 0.f, a2 = 0.f;
            for (int s = 0; s < 16; ++s) {
                a1 += red[(s * C1_COUT + tid) * 2 + 0];
                a2 += red[(s * C1_COUT + tid) * 2 + 1];
            }
            float *dst = p.stats + ((size_t)blockIdx.x * C1_COUT + tid) * 2;
            dst[0] = a1;
            dst[1] = a2;
        }
    }
}

}  // namespace

extern "C" int ds_conv5x5s2_c1_stats_rows(int B, int H) {
    DS_REQUIRE(B > 0 && H > 0, DS_ERR_BAD_SHAPE);
    return B * ds_ceil_div((H - 1) / 2 + 1, C1_RT);
}

extern "C" int ds_conv5x5s2_c1_fwd_f32(const float *x, const float *w_packed, const float *scale,
                                       const float *shift, float *y, float *stats_partial, int B,
                                       int H, int W, int Cout, int flags, void *stream) {
    DS_REQUIRE(x && w_packed && y, DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_AFFINE) || (scale && shift), DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_STATS) || stats_partial, DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_RESIDUAL), DS_ERR_UNSUPPORTED);
    DS_REQUIRE(B > 0 && H > 0 && W > 0 && W <= 256, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(Cout == C1_COUT, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(DS_ALIGNED16(w_packed) && DS_ALIGNED16(y), DS_ERR_ALIGNMENT);
    Conv1K k;
    k.x = x; k.w = w_packed; k.scale = scale; k.shift = shift; k.y = y; k.stats = stats_partial;
    k.H = H; k.W = W;
    k.Ho = (H - 1) / 2 + 1;
    k.Wo = (W - 1) / 2 + 1;
    DS_REQUIRE((long long)B * k.Ho * k.Wo * C1_COUT < (1ll << 31), DS_ERR_BAD_SHAPE);
    k.tiles_per_img = ds_ceil_div(k.Ho, C1_RT);
    k.cols_in = 2 * (k.Wo - 1) + 5;
    k.flags = flags;
    const size_t lds = ((size_t)(2 * (C1_RT - 1) + 5) * k.cols_in + 16 * C1_COUT * 2) * 4;
    DS_LAUNCH(conv5x5s2_c1_kernel, B * k.tiles_per_img, 256, lds, stream, k);
    return ds_last_launch_error();
}
