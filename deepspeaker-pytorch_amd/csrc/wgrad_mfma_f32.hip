// wgrad_mfma_f32.hip -- filter gradients of the convolution stack and of the fc layer.
//
// dW[co][ci][kh][kw] = sum over (b, h, w) of dY[b,h,w,co] * X[b, s*h+kh-p, s*w+kw-p, ci]
// (autograd of nn.Conv2d / nn.Linear under loss.backward(), reference train_triplet.py:223,290;
// layers model.py:47-50, 93-106, 164).  As a GEMM: M = Cout, N = Cin, K = every output pixel of the
// batch -- a short, very deep contraction -- so the pixel axis is split across workgroups:
//   workgroup = (tap group, 64/128 output channels, 64/32 input channels, pixel split)
//   per pixel tile: dY rows and the X halo tile are staged in LDS once and reused by every tap of
//   the group; one v_mfma_f32_32x32x2_f32 contracts TWO pixels (the two lane halves) for a
//   32(co) x 32(ci) block of one tap; a wave keeps one accumulator per tap.
// Partial sums go to a [split][tap][Cout][Cin] buffer and a second kernel folds the splits in a
// fixed order into the reference's OIHW layout (deterministic, no atomics).
#include <ds_device.h>
#include "ds_common.h"
#include "wgrad_reduce.h"

namespace {

struct WgradK {
    const float *x, *gz;
    float *partial;
    int H, W, Cin, Ho, Wo, Cout;
    int KS, IS, pad;
    int RT, NI, segs_per_img, n_segs, n_tiles;
    int rows_in, cols_in, seg_pix;
    int P;                       // output-pixel slots per tile (even, >= NI*RT*Wo)
    int S, n_co_tiles, n_ci_tiles;
};

// TG taps per workgroup: 9 (all of a 3x3), 5 (one kernel row of a 5x5; blockIdx selects the row) or 1.
// WIDE_CO = false: 64 co x 64 ci per workgroup (waves 2 x 2);  true: 128 co x 32 ci (waves 4 x 1).
template <int TG, bool WIDE_CO>
__global__ void __launch_bounds__(256) wgrad_mfma_f32_kernel(const WgradK p) {
    constexpr int COT = WIDE_CO ? 128 : 64;
    constexpr int CIT = WIDE_CO ? 32 : 64;
    float *lds = ds_dynamic_lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int co_sub = WIDE_CO ? wave : (wave & 1);
    const int ci_sub = WIDE_CO ? 0 : (wave >> 1);

    int bid = blockIdx.x;
    const int sp = bid % p.S;
    bid /= p.S;
    const int cit = bid % p.n_ci_tiles;
    bid /= p.n_ci_tiles;
    const int cot = bid % p.n_co_tiles;
    const int tg = bid / p.n_co_tiles;                  // tap group (kernel row for 5x5)

    const int tile_in_pix = p.NI * p.seg_pix;
    float *gzt = lds;                                   // [P][COT]
    float *xt = gzt + p.P * COT;                        // [tile_in_pix][CIT]
    int *pixtab = (int *)(xt + tile_in_pix * CIT);      // [P] float offset of each pixel's (0,0)-tap input

    f32x16 acc[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int pix_per_seg = p.RT * p.Wo;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int GV = COT / 4, XV = CIT / 4;           // float4 per staged pixel
    constexpr int GSL = 8, XSL = 12;                    // staging slots per thread (host plan keeps within)

    // ---- tile-invariant staging descriptors: the (segment, row, column) of every slot is decomposed ONCE
    //      (integer divisions), packed into one register per slot; per tile only the segment's image / first
    //      row change, and those come from a small LDS table written by NI threads ----
    int *segtab = pixtab + p.P;                         // [2][NI][2] = {image, first output row}, double-buffered
    int g_desc[GSL], x_desc[XSL];                       // seg << 24 | row << 12 | col   (-1: unused slot)
    const int n_g = p.P * GV, n_x = tile_in_pix * XV;
#pragma unroll
    for (int it = 0; it < GSL; ++it) {
        const int i = tid + it * 256;
        g_desc[it] = -1;
        if (i < n_g) {
            const int pp = i / GV;
            const int seg = pp / pix_per_seg, rem = pp - seg * pix_per_seg;
            const int r = rem / p.Wo, c = rem - r * p.Wo;
            g_desc[it] = (seg < p.NI) ? ((seg << 24) | (r << 12) | c) : (0xFF << 24);
        }
    }
#pragma unroll
    for (int it = 0; it < XSL; ++it) {
        const int i = tid + it * 256;
        x_desc[it] = -1;
        if (i < n_x) {
            const int pix = i / XV;
            const int seg = pix / p.seg_pix, pr = pix - seg * p.seg_pix;
            const int rr = pr / p.cols_in, cc = pr - rr * p.cols_in;
            x_desc[it] = (seg << 24) | (rr << 12) | cc;
        }
    }
    // pixel table: float offset of each output pixel's (0,0)-tap input inside the tile (tile-invariant)
    __syncthreads();
    for (int pp = tid; pp < p.P; pp += 256) {
        const int seg = pp / pix_per_seg, rem = pp - seg * pix_per_seg;
        const int r = rem / p.Wo, c = rem - r * p.Wo;
        pixtab[pp] = (seg < p.NI) ? (seg * p.seg_pix + (p.IS * r) * p.cols_in + p.IS * c) * CIT : 0;
    }

    // The accumulators alone take 144 registers, so only one workgroup fits per CU: the next tile's global
    // loads are therefore issued into registers BEFORE this tile's matrix work and written to LDS after it
    // (software pipeline over tiles; the per-tile segment table is double-buffered in LDS).
    f32x4 gv[GSL], xv[XSL];
    bool gok[GSL], xok[XSL];
    auto fill_segtab = [&](int tile, int buf) {
        if (tid < p.NI) {
            const int gseg = tile * p.NI + tid;
            int b = -1, r0 = 0;
            if (gseg < p.n_segs) {
                b = gseg / p.segs_per_img;
                r0 = (gseg - b * p.segs_per_img) * p.RT;
            }
            segtab[(buf * p.NI + tid) * 2] = b;
            segtab[(buf * p.NI + tid) * 2 + 1] = r0;
        }
    };
    auto issue_loads = [&](int buf) {
        const int *st = segtab + buf * p.NI * 2;
#pragma unroll
        for (int it = 0; it < GSL; ++it) {
            const int d = g_desc[it];
            gok[it] = false;
            size_t off = 0;
            if (d != -1) {
                const int seg = (d >> 24) & 0xFF, r = (d >> 12) & 0xFFF, c = d & 0xFFF;
                if (seg != 0xFF) {
                    const int b = st[seg * 2], row = st[seg * 2 + 1] + r;
                    if (b >= 0 && row < p.Ho) {
                        gok[it] = true;
                        off = ((size_t)(b * p.Ho + row) * p.Wo + c) * p.Cout + cot * COT + ((tid + it * 256) % GV) * 4;
                    }
                }
                gv[it] = *(const f32x4 *)(p.gz + off);
            }
        }
#pragma unroll
        for (int it = 0; it < XSL; ++it) {
            const int d = x_desc[it];
            xok[it] = false;
            size_t off = 0;
            if (d != -1) {
                const int seg = (d >> 24) & 0xFF, rr = (d >> 12) & 0xFFF, cc = d & 0xFFF;
                const int b = st[seg * 2];
                const int h = p.IS * st[seg * 2 + 1] - p.pad + rr, w = cc - p.pad;
                if (b >= 0 && h >= 0 && h < p.H && w >= 0 && w < p.W) {
                    xok[it] = true;
                    off = ((size_t)(b * p.H + h) * p.W + w) * p.Cin + cit * CIT + ((tid + it * 256) % XV) * 4;
                }
                xv[it] = *(const f32x4 *)(p.x + off);
            }
        }
    };
    fill_segtab(sp, 0);
    __syncthreads();
    if (sp < p.n_tiles) issue_loads(0);
    int buf = 0;
    for (int tile = sp; tile < p.n_tiles; tile += p.S, buf ^= 1) {
        __syncthreads();                                // previous tile's fragment reads are done
#pragma unroll
        for (int it = 0; it < GSL; ++it)
            if (g_desc[it] != -1) *(f32x4 *)(gzt + (size_t)(tid + it * 256) * 4) = gok[it] ? gv[it] : zero4;
#pragma unroll
        for (int it = 0; it < XSL; ++it)
            if (x_desc[it] != -1) *(f32x4 *)(xt + (size_t)(tid + it * 256) * 4) = xok[it] ? xv[it] : zero4;
        fill_segtab(tile + p.S, buf ^ 1);
        __syncthreads();
        if (tile + p.S < p.n_tiles) issue_loads(buf ^ 1);   // in flight during this tile's matrix work
        // ---- contract: two pixels per MFMA (lane halves), one accumulator per tap ----
        const float *ga = gzt + co_sub * 32 + l31;
        const float *xb = xt + ci_sub * 32 + l31;
        for (int s = 0; s < p.P; s += 2) {
            const int pp = s + lhi;
            const float a = ga[pp * COT];
            const float *xp = xb + pixtab[pp];
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const int kh = (TG == 9) ? t / 3 : tg, kw = (TG == 9) ? t % 3 : t;
                const float b = xp[(kh * p.cols_in + kw) * CIT];
                acc[t] = ds_mfma_32x32x2_f32(a, b, acc[t]);
            }
        }
    }

    // ---- partial[sp][tap][co][ci] ----
    const int co0 = cot * COT + co_sub * 32, ci0 = cit * CIT + ci_sub * 32;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        const int tap = (TG == 9) ? t : (TG == 5 ? tg * 5 + t : 0);
        float *dst = p.partial + (((size_t)sp * p.KS * p.KS + tap) * p.Cout) * p.Cin;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            dst[(size_t)co * p.Cin + ci0 + l31] = acc[t][r];
        }
    }
}

// ---- conv1 (Cin = 1): dW[co][kh][kw] = sum_pix dY[pix][co] * x[pix @ tap]; VALU kernel mirroring
//      conv5x5s2_c1_kernel: 16 lanes per pixel (4 channels each), 100 accumulators per lane ----
struct Wgrad1K {
    const float *x;
    const void *gz;              // f32, or fp16 (GZ16: the fp16 training step's loss-scaled gradient)
    float *partial;
    int H, W, Ho, Wo, tiles_per_img, n_tiles, cols_in;
};
constexpr int W1_RT = 4;

template <bool GZ16>
__global__ void __launch_bounds__(256) wgrad_c1_kernel(const Wgrad1K p) {
    float *lds = ds_dynamic_lds();
    const int tid = threadIdx.x, cg = tid & 15, slot = tid >> 4;
    constexpr int ROWS_IN = 2 * (W1_RT - 1) + 5;
    const int n_in = ROWS_IN * p.cols_in;
    float *red = lds + n_in;                             // [4 waves][25][64]
    f32x4 acc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        const int b = tile / p.tiles_per_img;
        const int r0 = (tile - b * p.tiles_per_img) * W1_RT;
        __syncthreads();
        const float *xb = p.x + (size_t)b * p.H * p.W;
        for (int i = tid; i < n_in; i += 256) {
            const int rr = i / p.cols_in, cc = i - rr * p.cols_in;
            const int h = 2 * r0 - 2 + rr, w = cc - 2;
            lds[i] = (h >= 0 && h < p.H && w >= 0 && w < p.W) ? xb[(size_t)h * p.W + w] : 0.0f;
        }
        __syncthreads();
        const int n_pix = W1_RT * p.Wo;
        for (int pix = slot; pix < n_pix; pix += 16) {
            const int r = pix / p.Wo, c = pix - r * p.Wo;
            if (r0 + r >= p.Ho) break;
            const size_t go = (((size_t)b * p.Ho + r0 + r) * p.Wo + c) * 64 + cg * 4;
            f32x4 g;
            if constexpr (GZ16) g = __builtin_convertvector(*(const f16x4 *)((const _Float16 *)p.gz + go), f32x4);
            else g = *(const f32x4 *)((const float *)p.gz + go);
            const float *in = lds + (2 * r) * p.cols_in + 2 * c;
#pragma unroll
            for (int kh = 0; kh < 5; ++kh)
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) acc[kh * 5 + kw] += in[kh * p.cols_in + kw] * g;
        }
    }
    // fold the 4 pixel slots that share a wave (lanes differing in bits 4,5), then the 4 waves
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int t = 0; t < 25; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[t][j];
            v += ds_shfl_xor(v, 16);
            v += ds_shfl_xor(v, 32);
            if (lane < 16) red[(wave * 25 + t) * 64 + cg * 4 + j] = v;
        }
    __syncthreads();
    for (int i = tid; i < 25 * 64; i += 256) {
        const float v = red[i] + red[1600 + i] + red[3200 + i] + red[4800 + i];
        p.partial[(size_t)blockIdx.x * 1600 + i] = v;        // [blk][tap][co]
    }
}

// gw[co][0][tap] = sum_blk partial[blk][tap][co]: 8 lanes share one output (fixed interleave), then a fixed
// xor-tree over the 8 lanes -- deterministic, and 50 workgroups instead of 7 serial ones
__global__ void __launch_bounds__(256) wgrad_c1_reduce_kernel(const float *partial, float *gw, int n_blk, float scale) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t >> 3, r = t & 7;
    float s = 0.f;
    if (i < 1600)
        for (int k = r; k < n_blk; k += 8) s += partial[(size_t)k * 1600 + i];
    s += ds_shfl_xor(s, 1);
    s += ds_shfl_xor(s, 2);
    s += ds_shfl_xor(s, 4);
    if (i < 1600 && r == 0) {
        const int tap = i / 64, co = i - tap * 64;
        gw[co * 25 + tap] = scale == 1.0f ? s : s * scale;
    }
}

struct WgradPlan {
    WgradK k;
    int tg, n_tg, wide;
    int grid;
    size_t lds_bytes;
    long long partial_floats;
};

static int plan_wgrad(WgradPlan &pl, const ds_conv_shape *s) {
    DS_REQUIRE(s != nullptr, DS_ERR_NULL);
    DS_REQUIRE(s->B > 0 && s->H > 0 && s->W > 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(s->KS == 1 || s->KS == 3 || s->KS == 5, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->stride == 1 || s->stride == 2, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->Cin % 64 == 0 && s->Cout % 64 == 0, DS_ERR_BAD_SHAPE);
    WgradK &k = pl.k;
    const int pad = s->KS / 2;
    k.H = s->H; k.W = s->W; k.Cin = s->Cin; k.Cout = s->Cout;
    k.Ho = (s->H + 2 * pad - s->KS) / s->stride + 1;
    k.Wo = (s->W + 2 * pad - s->KS) / s->stride + 1;
    DS_REQUIRE(k.Ho > 0 && k.Wo > 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE((long long)s->B * s->H * s->W * s->Cin < (1ll << 31), DS_ERR_BAD_SHAPE);
    DS_REQUIRE((long long)s->B * k.Ho * k.Wo * s->Cout < (1ll << 31), DS_ERR_BAD_SHAPE);
    k.KS = s->KS; k.IS = s->stride; k.pad = pad;
    pl.tg = s->KS == 3 ? 9 : (s->KS == 5 ? 5 : 1);
    pl.n_tg = s->KS == 5 ? 5 : 1;
    // the 5x5 stride-2 halo tile is 4x larger per output pixel: trade input channels for output ones
    pl.wide = (s->KS == 5 && s->Cout % 128 == 0) ? 1 : 0;
    const int COT = pl.wide ? 128 : 64, CIT = pl.wide ? 32 : 64;
    // segment height: as many rows as keep <= 64 output pixels and <= 40 KiB of halo tile
    int best_rt = 0, best_ni = 1;
    for (int rt = 1; rt <= k.Ho; ++rt) {
        if (rt * k.Wo > 64) break;
        const int rows_in = s->stride * (rt - 1) + s->KS, cols_in = s->stride * (k.Wo - 1) + s->KS;
        if ((long long)rows_in * cols_in * CIT * 4 > 40 * 1024) break;
        best_rt = rt;
    }
    if (best_rt == 0) {                                   // very wide rows: one row per tile
        best_rt = 1;
        DS_REQUIRE(k.Wo <= 128, DS_ERR_UNSUPPORTED);
    }
    k.RT = best_rt;
    k.segs_per_img = ds_ceil_div(k.Ho, best_rt);
    k.n_segs = s->B * k.segs_per_img;
    k.rows_in = s->stride * (best_rt - 1) + s->KS;
    k.cols_in = s->stride * (k.Wo - 1) + s->KS;
    k.seg_pix = k.rows_in * k.cols_in;
    while ((best_ni + 1) * best_rt * k.Wo <= 64 && (long long)(best_ni + 1) * k.seg_pix * CIT * 4 <= 40 * 1024 &&
           best_ni + 1 <= k.n_segs)
        ++best_ni;
    k.NI = best_ni;
    k.P = (best_ni * best_rt * k.Wo + 1) & ~1;
    k.n_tiles = ds_ceil_div(k.n_segs, best_ni);
    k.n_co_tiles = s->Cout / COT;
    k.n_ci_tiles = s->Cin / CIT;
    const int base_blocks = pl.n_tg * k.n_co_tiles * k.n_ci_tiles;
    int S = ds_ceil_div(1024, base_blocks);               // aim at ~4 workgroups per CU
    if (S > k.n_tiles) S = k.n_tiles;
    if (S < 1) S = 1;
    k.S = S;
    pl.grid = base_blocks * S;
    pl.lds_bytes = ((size_t)k.P * COT + (size_t)k.NI * k.seg_pix * CIT + k.P + 4 * k.NI) * 4;
    DS_REQUIRE(k.P * (COT / 4) <= 8 * 256 && k.NI * k.seg_pix * (CIT / 4) <= 12 * 256 && k.NI <= 255 &&
                   k.rows_in < 4096 && k.cols_in < 4096, DS_ERR_UNSUPPORTED);
    pl.partial_floats = (long long)S * s->KS * s->KS * s->Cout * s->Cin;
    return DS_OK;
}

}  // namespace

extern "C" long long ds_conv_wgrad_workspace_floats(const ds_conv_shape *s) {
    if (s && s->Cin == 1) {
        if (s->B <= 0 || s->H <= 0 || s->W <= 0) return DS_ERR_BAD_SHAPE;
        const long long tiles = (long long)s->B * ds_ceil_div((s->H - 1) / 2 + 1, W1_RT);
        return (tiles < 1024 ? tiles : 1024) * 1600;
    }
    WgradPlan pl;
    int rc = plan_wgrad(pl, s);
    return rc == DS_OK ? pl.partial_floats : rc;
}

static int wgrad_c1(const ds_conv_shape *s, const float *x, const void *gy, bool gy_f16, float *workspace, float *gw_oihw,
                    float out_scale, void *stream) {
    DS_REQUIRE(s->KS == 5 && s->stride == 2 && s->Cout == 64, DS_ERR_UNSUPPORTED);
    Wgrad1K k;
    k.x = x; k.gz = gy; k.partial = workspace;
    k.H = s->H; k.W = s->W;
    k.Ho = (s->H - 1) / 2 + 1; k.Wo = (s->W - 1) / 2 + 1;
    k.tiles_per_img = ds_ceil_div(k.Ho, W1_RT);
    k.n_tiles = s->B * k.tiles_per_img;
    k.cols_in = 2 * (k.Wo - 1) + 5;
    const int grid = k.n_tiles < 1024 ? k.n_tiles : 1024;
    const size_t lds = ((size_t)(2 * (W1_RT - 1) + 5) * k.cols_in + 4 * 25 * 64) * 4;
    if (gy_f16) DS_LAUNCH(wgrad_c1_kernel<true>, grid, 256, lds, stream, k);
    else DS_LAUNCH(wgrad_c1_kernel<false>, grid, 256, lds, stream, k);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(wgrad_c1_reduce_kernel, 50, 256, 0, stream, (const float *)workspace, gw_oihw, grid, out_scale);
    return ds_last_launch_error();
}

// conv1's filter gradient (Cin = 1: f32 network input x [B,H,W]) from an fp16 output gradient in loss-scaled units;
// workspace: ds_conv_wgrad_workspace_floats(s) floats
extern "C" int ds_conv_wgrad_c1_f16(const ds_conv_shape *s, const float *x, const void *gy_f16, float *workspace,
                                    float *gw_oihw, float out_scale, void *stream) {
    DS_REQUIRE(s && x && gy_f16 && workspace && gw_oihw, DS_ERR_NULL);
    DS_REQUIRE(s->Cin == 1, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(DS_ALIGNED16(x) && DS_ALIGNED16(gy_f16), DS_ERR_ALIGNMENT);
    return wgrad_c1(s, x, gy_f16, true, workspace, gw_oihw, out_scale, stream);
}

// fc_F > 0: `s` describes the fc layer as a 1x1 convolution over [1,B,1,K] and the gradient is written
// in the reference's [N, C*F] order (C = Cin / fc_F).
extern "C" int ds_conv_wgrad_f32(const ds_conv_shape *s, const float *x, const float *gy, float *workspace,
                                 float *gw_oihw, int fc_F, void *stream) {
    DS_REQUIRE(s && x && gy && workspace && gw_oihw, DS_ERR_NULL);
    DS_REQUIRE(DS_ALIGNED16(x) && DS_ALIGNED16(gy), DS_ERR_ALIGNMENT);
    if (s->Cin == 1) return wgrad_c1(s, x, gy, false, workspace, gw_oihw, 1.0f, stream);      // conv1
    WgradPlan pl;
    int rc = plan_wgrad(pl, s);
    if (rc != DS_OK) return rc;
    pl.k.x = x; pl.k.gz = gy; pl.k.partial = workspace;
    if (pl.tg == 9) DS_LAUNCH((wgrad_mfma_f32_kernel<9, false>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    else if (pl.tg == 5 && pl.wide) DS_LAUNCH((wgrad_mfma_f32_kernel<5, true>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    else if (pl.tg == 5) DS_LAUNCH((wgrad_mfma_f32_kernel<5, false>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    else DS_LAUNCH((wgrad_mfma_f32_kernel<1, false>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    rc = ds_last_launch_error();
    if (rc) return rc;
    const long long n = (long long)s->KS * s->KS * s->Cout * s->Cin;
    int lg, rgrid;
    wgrad_reduce_shape(n, pl.k.S, lg, rgrid);
    DS_LAUNCH(wgrad_reduce_kernel, rgrid, 256, 1024, stream, (const float *)workspace, gw_oihw,
              pl.k.S, s->KS * s->KS, s->Cout, s->Cin, fc_F, 1.0f, lg);
    return ds_last_launch_error();
}
