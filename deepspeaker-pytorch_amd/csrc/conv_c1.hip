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

constexpr int C1_RT = 16;         // output rows per workgroup
constexpr int C1_COUT = 64;
constexpr int C1_VSLOTS = 3;      // 16-byte staging pieces per thread (35 rows x 64 columns = 560 pieces)

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

// ---- the same layer on the bf16 matrix cores with split operands (bf16x3) ----------------------------------
// As a GEMM the layer is [64 channels x 25 taps] x [25 taps x pixels]: K = 25 is padded to two k-steps of 16.
// The filters are the A operand (each lane builds its hi / lo fragments once, from the same packed f32 bank
// as the VALU kernel); a lane's B fragment is 8 taps of ONE output pixel, gathered from the f32 input tile
// in LDS by 8 scalar reads and split into hi / lo in registers.  Accumulators hold the transposed product
// (lane = pixel), so the epilogue is the convolution kernel's: turn each 32-pixel sub-tile around through a
// wave-private LDS buffer and store whole 256-byte pixel rows with raw buffer stores.  ~5x fewer issued
// instructions per pixel than the VALU kernel (which is issue-bound at 1.6x the time of its HBM traffic).
// OUT16 / STATS (the output type and the BatchNorm partial sums) are compile-time: the store loop is straight-line
// code; the affine is data (identity when off) and the clip a select.
template <bool OUT16, bool STATS>
__global__ void __launch_bounds__(256) conv5x5s2_c1_bf16_kernel(const Conv1K p, unsigned y_bytes) {
    float *lds = ds_dynamic_lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int b = blockIdx.x / p.tiles_per_img;
    const int r0 = (blockIdx.x - b * p.tiles_per_img) * C1_RT;
    constexpr int ROWS_IN = 2 * (C1_RT - 1) + 5;
    constexpr int TP = C1_COUT + 4;                         // transposition-buffer row pitch (floats)
    const int n_in = ROWS_IN * p.cols_in;
    float *tb = lds + ((n_in + 3) & ~3) + wave * (32 * TP);  // [32][TP] per wave
    float *red = lds + ((n_in + 3) & ~3) + 4 * 32 * TP;      // [4 waves][64][2]

    // filter fragments: lane (channel l31 of sub-tile ns, taps 16*ks + 8*lhi .. +7)
    bf16x8 w_hi[2][2], w_lo[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int ns = 0; ns < 2; ++ns)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int t = 16 * ks + 8 * lhi + i;
                const float v = t < 25 ? p.w[t * C1_COUT + ns * 32 + l31] : 0.0f;
                const __bf16 h = (__bf16)v;
                w_hi[ks][ns][i] = h;
                w_lo[ks][ns][i] = (__bf16)(v - (float)h);
            }
    // tap offsets of this lane's 16 taps inside the input tile (taps >= 25 are padding: the pixel's own first tap,
    // multiplied by a zero filter entry)
    int toff[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int t = 16 * ks + 8 * lhi + i;
            toff[ks][i] = t < 25 ? (t / 5) * p.cols_in + (t % 5) : 0;      // padding taps: any finite word (x 0)
        }

    // stage the zero-padded input tile.  The tile's image rows are whole rows of x, i.e. ONE contiguous span:
    // every thread requests its 16-byte pieces up front (a few independent loads instead of a chain of ~10
    // dependent word loads), zero-fills the tile meanwhile, then drops the pieces into the padded rows.
    const float *xb = p.x + (size_t)b * p.H * p.W;
    if ((p.W & 3) == 0 && ROWS_IN * p.W <= 4 * 256 * C1_VSLOTS) {
        const int h_first = 2 * r0 - 2;
        const int h_lo = h_first < 0 ? 0 : h_first;
        const int h_hi = (h_first + ROWS_IN < p.H) ? h_first + ROWS_IN : p.H;          // image rows [h_lo, h_hi)
        const int n4 = (h_hi - h_lo) * (p.W >> 2);
        const f32x4 *src = (const f32x4 *)(xb + (size_t)h_lo * p.W);
        f32x4 piece[C1_VSLOTS];
#pragma unroll
        for (int it = 0; it < C1_VSLOTS; ++it) {
            const int i = tid + it * 256;
            piece[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (i < n4) piece[it] = src[i];
        }
        for (int i = tid; i < n_in; i += 256) lds[i] = 0.0f;
        __syncthreads();
        const int w4 = p.W >> 2;
        const float rcp_w4 = 1.0f / (float)w4;
#pragma unroll
        for (int it = 0; it < C1_VSLOTS; ++it) {
            const int i = tid + it * 256;
            if (i < n4) {
                const int rr = ds_div_small(i, w4, rcp_w4), q = i - rr * w4;
                float *dst = lds + (h_lo - h_first + rr) * p.cols_in + 2 + 4 * q;
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[j] = piece[it][j];
            }
        }
    } else {
        const float rcp_ci = 1.0f / (float)p.cols_in;
        for (int i = tid; i < n_in; i += 256) {
            const int rr = ds_div_small(i, p.cols_in, rcp_ci), cc = i - rr * p.cols_in;
            const int h = 2 * r0 - 2 + rr, w = cc - 2;
            lds[i] = (h >= 0 && h < p.H && w >= 0 && w < p.W) ? xb[(size_t)h * p.W + w] : 0.0f;
        }
    }
    __syncthreads();

    const int n_pix = C1_RT * p.Wo, n_sub = (n_pix + 31) >> 5;
    const float rcp_wo = 1.0f / (float)p.Wo;
    // lanes per pixel row, rows per store instruction: a lane owns 4 channels (16 bytes of f32) or 8 (16 bytes of fp16)
    constexpr int LPP = OUT16 ? 8 : 16, PPI = 64 / LPP, NRI = 32 / PPI, CPL = C1_COUT / LPP;
    const int my_c = (lane % LPP) * CPL, my_p = lane / LPP;
    f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f}, sc4b = sc4, sh4b = sh4;
    if (p.flags & DS_EPI_AFFINE) {
        sc4 = *(const f32x4 *)(p.scale + my_c);
        sh4 = *(const f32x4 *)(p.shift + my_c);
        if (OUT16) {
            sc4b = *(const f32x4 *)(p.scale + my_c + 4);
            sh4b = *(const f32x4 *)(p.shift + my_c + 4);
        }
    }
    const bool clip = p.flags & DS_EPI_CLIP;
    const ds_buffer ybuf = ds_make_buffer(p.y, y_bytes);
    const int pix0 = (b * p.Ho + r0) * p.Wo;               // first output pixel of the tile
    const int rows_left = p.Ho - r0;
    const int pix_lim = (rows_left < C1_RT ? rows_left : C1_RT) * p.Wo;
    float ps1[4] = {0.f, 0.f, 0.f, 0.f}, ps2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int sub = wave; sub < n_sub; sub += 4) {
        // ---- B fragments: 16 taps of this lane's pixel ----
        const int m = sub * 32 + l31;
        const int r = ds_div_small(m, p.Wo, rcp_wo), c = m - r * p.Wo;
        const float *in = lds + ((m < n_pix) ? (2 * r) * p.cols_in + 2 * c : 0);
        ds_u32x4 xh[2], xl[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 8; i += 2) {            // two taps per packed conversion
                unsigned hi2, lo2;
                ds_split_bf16x2(in[toff[ks][i]], in[toff[ks][i + 1]], hi2, lo2);
                xh[ks][i >> 1] = hi2;
                xl[ks][i >> 1] = lo2;
            }
        f32x16 acc[2];
#pragma unroll
        for (int ns = 0; ns < 2; ++ns)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[ns][q] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int ns = 0; ns < 2; ++ns) acc[ns] = ds_mfma_32x32x16_bf16(w_hi[ks][ns], __builtin_bit_cast(bf16x8, xl[ks]), acc[ns]);
#pragma unroll
            for (int ns = 0; ns < 2; ++ns) acc[ns] = ds_mfma_32x32x16_bf16(w_lo[ks][ns], __builtin_bit_cast(bf16x8, xh[ks]), acc[ns]);
#pragma unroll
            for (int ns = 0; ns < 2; ++ns) acc[ns] = ds_mfma_32x32x16_bf16(w_hi[ks][ns], __builtin_bit_cast(bf16x8, xh[ks]), acc[ns]);
        }
        // ---- epilogue of the sub-tile: lane = pixel, register quad g = channels 8g + 4*lhi .. +3 ----
#pragma unroll
        for (int ns = 0; ns < 2; ++ns)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[ns][4 * g + j];
                *(f32x4 *)(tb + l31 * TP + ns * 32 + 8 * g + 4 * lhi) = v;
            }
        ds_wave_sync();
#pragma unroll
        for (int k = 0; k < NRI; ++k) {
            // the tile's pixels are consecutive rows of one image: pixel pm of the tile is pixel pix0 + pm of y
            const int pm = sub * 32 + k * PPI + my_p;
            const bool live = pm < pix_lim;
            const unsigned eoff = (unsigned)((pix0 + pm) * C1_COUT + my_c);
            const unsigned voff = live ? eoff * 4u : DS_BUFFER_OOB;
            f32x4 v = *(const f32x4 *)(tb + (k * PPI + my_p) * TP + my_c);
            if (OUT16) {
                // fp16 activations for the fp16 convolution path: packed f32 affine, packed conversion, packed fp16 clip
                // (0 and 20 are fp16 numbers and rounding is monotonic: clip(round(t)) == round(clip(t)))
                const f32x4 vb = *(const f32x4 *)(tb + (k * PPI + my_p) * TP + my_c + 4);
                const f32x4 ta = v * sc4 + sh4, tb4 = vb * sc4b + sh4b;
                f16x8 h;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h[j] = (_Float16)ta[j];
                    h[4 + j] = (_Float16)tb4[j];
                }
                const f16x8 lo = {0, 0, 0, 0, 0, 0, 0, 0}, hi = {20, 20, 20, 20, 20, 20, 20, 20};
                const f16x8 hc = __builtin_elementwise_min(__builtin_elementwise_max(h, lo), hi);
                h = clip ? hc : h;
                ds_buffer_store_out_f32x4(ybuf, live ? eoff * 2u : DS_BUFFER_OOB, __builtin_bit_cast(f32x4, h));
                continue;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = v[j];
                if (STATS) {
                    ps1[j] += live ? t : 0.0f;
                    ps2[j] += live ? t * t : 0.0f;
                }
                t = t * sc4[j] + sh4[j];
                v[j] = clip ? fminf(fmaxf(t, 0.0f), 20.0f) : t;       // a select, not a branch (NaN passes when off)
            }
            ds_buffer_store_out_f32x4(ybuf, voff, v);
        }
        ds_wave_sync();
    }
    if (STATS && !OUT16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ps1[j] += ds_shfl_xor(ps1[j], 16);
            ps2[j] += ds_shfl_xor(ps2[j], 16);
            ps1[j] += ds_shfl_xor(ps1[j], 32);
            ps2[j] += ds_shfl_xor(ps2[j], 32);
            if (my_p == 0) {
                red[(wave * C1_COUT + my_c + j) * 2 + 0] = ps1[j];
                red[(wave * C1_COUT + my_c + j) * 2 + 1] = ps2[j];
            }
        }
        __syncthreads();
        if (tid < C1_COUT) {
            float a1 = 0.f, a2 = 0.f;
            for (int s = 0; s < 4; ++s) {
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

// the same contract on the bf16 matrix cores with split operands (f32-class accuracy); same packed filter
// bank (ds_pack_conv1_weight_f32), same statistics rows
extern "C" int ds_conv5x5s2_c1_fwd_bf16(const float *x, const float *w_packed, const float *scale,
                                        const float *shift, void *y_out, float *stats_partial, int B,
                                        int H, int W, int Cout, int flags, void *stream) {
    float *y = (float *)y_out;                 // fp16 storage with DS_EPI_OUT_F16
    DS_REQUIRE(x && w_packed && y, DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_AFFINE) || (scale && shift), DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_STATS) || stats_partial, DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_RESIDUAL), DS_ERR_UNSUPPORTED);
    DS_REQUIRE(B > 0 && H > 0 && W > 0 && W <= 256, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(Cout == C1_COUT, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(DS_ALIGNED16(w_packed) && DS_ALIGNED16(y) && (!scale || DS_ALIGNED16(scale)) &&
                   (!shift || DS_ALIGNED16(shift)), DS_ERR_ALIGNMENT);
    Conv1K k;
    k.x = x; k.w = w_packed; k.scale = scale; k.shift = shift; k.y = y; k.stats = stats_partial;
    k.H = H; k.W = W;
    k.Ho = (H - 1) / 2 + 1;
    k.Wo = (W - 1) / 2 + 1;
    DS_REQUIRE((long long)B * k.Ho * k.Wo * C1_COUT < (1ll << 30), DS_ERR_BAD_SHAPE);     // 32-bit byte offsets
    k.tiles_per_img = ds_ceil_div(k.Ho, C1_RT);
    k.cols_in = 2 * (k.Wo - 1) + 5;
    k.flags = flags;
    const size_t n_in = (size_t)(2 * (C1_RT - 1) + 5) * k.cols_in;
    const size_t lds = (((n_in + 3) & ~(size_t)3) + 4 * 32 * (C1_COUT + 4) + 4 * C1_COUT * 2) * 4;
    const unsigned y_bytes = (unsigned)((long long)B * k.Ho * k.Wo * C1_COUT * ((flags & DS_EPI_OUT_F16) ? 2 : 4));
    const int grid = B * k.tiles_per_img;
    const bool h16 = flags & DS_EPI_OUT_F16, st = flags & DS_EPI_STATS;
    DS_REQUIRE(!(h16 && st), DS_ERR_UNSUPPORTED);          // statistics come with the f32 (training) output
    if (h16) DS_LAUNCH((conv5x5s2_c1_bf16_kernel<true, false>), grid, 256, lds, stream, k, y_bytes);
    else if (st) DS_LAUNCH((conv5x5s2_c1_bf16_kernel<false, true>), grid, 256, lds, stream, k, y_bytes);
    else DS_LAUNCH((conv5x5s2_c1_bf16_kernel<false, false>), grid, 256, lds, stream, k, y_bytes);
    return ds_last_launch_error();
}
