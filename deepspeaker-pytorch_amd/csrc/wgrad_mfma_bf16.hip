// wgrad_mfma_bf16.hip -- filter gradients of the 3x3 / 5x5 convolution layers on the bf16 matrix cores
// with split operands (bf16x3: x = hi + lo, product = hi*hi + hi*lo + lo*hi, f32 accumulate).
//
// dW[co][ci][kh][kw] = sum over (b, h, w) of dY[b,h,w,co] * X[b, s*h+kh-p, s*w+kw-p, ci]
// (autograd of nn.Conv2d under loss.backward(), reference train_triplet.py:223; layers model.py:47-50,
// 98-106).  As a GEMM: M = Cout, N = Cin, K = every output pixel of the batch.  Same decomposition as the
// f32 kernel (wgrad_mfma_f32.hip): workgroup = (tap group, 64 co, 64 ci, pixel split); per pixel tile the
// dY rows and the X halo tile are staged in LDS once and reused by every tap of the group; partial sums go
// to [split][tap][Cout][Cin] and wgrad_reduce_kernel folds them in a fixed order (deterministic).
//
// What is different: v_mfma_f32_32x32x16_bf16 contracts 16 pixels per instruction and wants, per lane, 8
// CONSECUTIVE pixels of ONE channel -- the transposed view of the channels-last activations.  The tiles
// stay pixel-major in LDS ([pixel][hi 64 ch | lo 64 ch] bf16 records, written with coalesced 8-byte
// stores) and the operands are fetched with ds_read_b64_tr_b16: within a 16-lane group lane i supplies the
// 8-byte piece (row i>>2, column quad i&3) of a 4-pixel x 16-channel block and receives column i, i.e. 4
// pixels of its own channel.  Two such reads make one bf16x8 fragment; because every lane supplies its own
// pixel address, any tap offset or stride works without alignment constraints.
#include <ds_device.h>
#include "ds_common.h"
#include "wgrad_reduce.h"

namespace {

constexpr int WB_C = 64;                     // channels per tile on both sides
constexpr int WB_REC = 4 * WB_C + 64;        // bytes per pixel record: hi (128) | lo (128) | pad -> 80 dwords = 16 mod 64

struct WgradKB {
    const float *x, *gz;
    float *partial;
    int H, W, Cin, Ho, Wo, Cout;
    int KS, IS, pad;
    int RT, NI, segs_per_img, n_segs, n_tiles;
    int rows_in, cols_in, seg_pix;
    int P;                       // output-pixel slots per tile (multiple of 16, >= NI*RT*Wo)
    int S, n_co_tiles, n_ci_tiles;
};

// one bf16x8 MFMA operand: pixels q0 .. q0+7 of this lane's channel; rec0 / rec1 are the byte addresses
// this lane supplies for the two 4-pixel blocks (its piece: pixel q0 + 4r + ((lane&15)>>2), quad lane&3)
__device__ __forceinline__ bf16x8 frag_tr(const char *rec0, const char *rec1) {
    const bf16x4 a = ds_read_tr16_b64(rec0);
    const bf16x4 b = ds_read_tr16_b64(rec1);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// TG taps per workgroup: 9 (all of a 3x3) or 5 (one kernel row of a 5x5; blockIdx selects the row).
// Four waves as 2 (co) x 2 (ci), each owning a 32 x 32 block of every tap of the group.
template <int TG>
__global__ void __launch_bounds__(256) wgrad_mfma_bf16_kernel(const WgradKB p) {
    char *lds = (char *)ds_dynamic_lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int co_sub = wave & 1, ci_sub = wave >> 1;

    int bid = blockIdx.x;
    const int sp = bid % p.S;
    bid /= p.S;
    const int cit = bid % p.n_ci_tiles;
    bid /= p.n_ci_tiles;
    const int cot = bid % p.n_co_tiles;
    const int tg = bid / p.n_co_tiles;                  // tap group (kernel row for 5x5)

    const int tile_in_pix = p.NI * p.seg_pix;
    char *gzt = lds;                                    // [P] records
    char *xt = gzt + (size_t)p.P * WB_REC;              // [tile_in_pix] records
    int *pixtab = (int *)(xt + (size_t)tile_in_pix * WB_REC);   // [P] byte offset of each pixel's (0,0)-tap input record
    int *segtab = pixtab + p.P;                         // [2][NI][4] per-tile segment origins, double-buffered

    f32x16 acc[TG];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int pix_per_seg = p.RT * p.Wo;
    constexpr int QV = WB_C / 4;                        // float4 per staged pixel
    constexpr int GSL = 4, XSL = 12;                    // staging slots per thread (the host plan keeps within)

    // ---- tile-invariant staging descriptors: per slot the float offset RELATIVE to the segment's origin and
    //      (segment << 16 | row); per tile only four numbers per segment change (segtab) ----
    int g_rel[GSL], g_sr[GSL], x_rel[XSL], x_sr[XSL];   // *_sr = -1: unused slot, -2: always-zero slot
    const int n_g = p.P * QV, n_x = tile_in_pix * QV;
#pragma unroll
    for (int it = 0; it < GSL; ++it) {
        const int i = tid + it * 256;
        g_sr[it] = -1;
        g_rel[it] = 0;
        if (i < n_g) {
            const int pp = i / QV, q = i - pp * QV;
            const int seg = pp / pix_per_seg, rem = pp - seg * pix_per_seg;
            const int r = rem / p.Wo, c = rem - r * p.Wo;
            g_sr[it] = (seg < p.NI) ? ((seg << 16) | r) : -2;
            g_rel[it] = (r * p.Wo + c) * p.Cout + cot * WB_C + q * 4;
        }
    }
#pragma unroll
    for (int it = 0; it < XSL; ++it) {
        const int i = tid + it * 256;
        x_sr[it] = -1;
        x_rel[it] = 0;
        if (i < n_x) {
            const int pix = i / QV, q = i - pix * QV;
            const int seg = pix / p.seg_pix, pr = pix - seg * p.seg_pix;
            const int rr = pr / p.cols_in, cc = pr - rr * p.cols_in;
            const int hrel = (TG == 5) ? p.IS * rr + tg : rr;          // image row = IS*r0 - pad + hrel
            const int w = cc - p.pad;
            x_sr[it] = (w >= 0 && w < p.W) ? ((seg << 16) | hrel) : -2;
            x_rel[it] = (hrel * p.W + cc) * p.Cin + cit * WB_C + q * 4;
        }
    }
    for (int pp = tid; pp < p.P; pp += 256) {
        const int seg = pp / pix_per_seg, rem = pp - seg * pix_per_seg;
        const int r = rem / p.Wo, c = rem - r * p.Wo;
        // TG == 5: the tile holds, per output row, only the ONE input row this kernel row touches
        pixtab[pp] = (seg < p.NI) ? (seg * p.seg_pix + ((TG == 5) ? r : p.IS * r) * p.cols_in + p.IS * c) * WB_REC : 0;
    }

    // software pipeline over tiles: the next tile's global loads are issued into registers before this
    // tile's matrix work and split / written to LDS after it
    f32x4 gv[GSL], xv[XSL];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto fill_segtab = [&](int tile, int buf) {          // {dY origin, output rows left, X origin, first image row}
        if (tid < p.NI) {
            const int gseg = tile * p.NI + tid;
            int gbase = 0, rows_left = 0, xbase = 0, h0 = -(1 << 20);
            if (gseg < p.n_segs) {
                const int b = gseg / p.segs_per_img;
                const int r0 = (gseg - b * p.segs_per_img) * p.RT;
                gbase = (b * p.Ho + r0) * p.Wo * p.Cout;
                rows_left = p.Ho - r0;
                h0 = p.IS * r0 - p.pad;
                xbase = ((b * p.H + h0) * p.W - p.pad) * p.Cin;
            }
            int *e = segtab + (buf * p.NI + tid) * 4;
            e[0] = gbase; e[1] = rows_left; e[2] = xbase; e[3] = h0;
        }
    };
    auto issue_loads = [&](int buf) {
        const int *st = segtab + buf * p.NI * 4;
#pragma unroll
        for (int it = 0; it < GSL; ++it) {
            gv[it] = zero4;
            if (g_sr[it] >= 0) {
                const int *e = st + (g_sr[it] >> 16) * 4;
                if ((g_sr[it] & 0xFFFF) < e[1]) gv[it] = *(const f32x4 *)(p.gz + (e[0] + g_rel[it]));
            }
        }
#pragma unroll
        for (int it = 0; it < XSL; ++it) {
            xv[it] = zero4;
            if (x_sr[it] >= 0) {
                const int *e = st + (x_sr[it] >> 16) * 4;
                const int h = e[3] + (x_sr[it] & 0xFFFF);
                if (h >= 0 && h < p.H) xv[it] = *(const f32x4 *)(p.x + (e[2] + x_rel[it]));
            }
        }
    };
    auto put_split = [&](char *rec, int q, const f32x4 v) {     // 4 channels -> hi / lo halves of the record
        bf16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = (__bf16)v[j];
            l[j] = (__bf16)(v[j] - (float)h[j]);
        }
        *(bf16x4 *)(rec + q * 8) = h;
        *(bf16x4 *)(rec + 2 * WB_C + q * 8) = l;
    };

    // this lane's piece of every transposing read: pixel (lane&15)>>2 of the 4-pixel block, channel quad
    // lane&3 of the 16-channel block (lane>>4)&1 of the wave's 32 channels
    const int piece_pix = (lane & 15) >> 2;
    const int a_col = (co_sub * 32 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;     // byte offset inside hi
    const int b_col = (ci_sub * 32 + ((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;

    fill_segtab(sp, 0);
    __syncthreads();
    if (sp < p.n_tiles) issue_loads(0);
    int buf = 0;
    for (int tile = sp; tile < p.n_tiles; tile += p.S, buf ^= 1) {
        __syncthreads();                                // previous tile's fragment reads are done
#pragma unroll
        for (int it = 0; it < GSL; ++it)
            if (g_sr[it] != -1) {
                const int i = tid + it * 256;
                put_split(gzt + (size_t)(i / QV) * WB_REC, i % QV, gv[it]);
            }
#pragma unroll
        for (int it = 0; it < XSL; ++it)
            if (x_sr[it] != -1) {
                const int i = tid + it * 256;
                put_split(xt + (size_t)(i / QV) * WB_REC, i % QV, xv[it]);
            }
        fill_segtab(tile + p.S, buf ^ 1);
        __syncthreads();
        if (tile + p.S < p.n_tiles) issue_loads(buf ^ 1);   // in flight during this tile's matrix work
        // ---- contract: 16 pixels per MFMA, one accumulator per tap ----
        for (int s = 0; s < p.P; s += 16) {
            const int pp0 = s + 8 * lhi + piece_pix, pp1 = pp0 + 4;
            const char *g0 = gzt + (size_t)pp0 * WB_REC + a_col, *g1 = gzt + (size_t)pp1 * WB_REC + a_col;
            const bf16x8 a_hi = frag_tr(g0, g1);
            const bf16x8 a_lo = frag_tr(g0 + 2 * WB_C, g1 + 2 * WB_C);
            const char *x0 = xt + pixtab[pp0] + b_col, *x1 = xt + pixtab[pp1] + b_col;
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const int kh = (TG == 9) ? t / 3 : tg, kw = (TG == 9) ? t % 3 : t;
                const int toff = ((TG == 5) ? kw : kh * p.cols_in + kw) * WB_REC;
                const bf16x8 b_hi = frag_tr(x0 + toff, x1 + toff);
                const bf16x8 b_lo = frag_tr(x0 + toff + 2 * WB_C, x1 + toff + 2 * WB_C);
                acc[t] = ds_mfma_32x32x16_bf16(a_lo, b_hi, acc[t]);
                acc[t] = ds_mfma_32x32x16_bf16(a_hi, b_lo, acc[t]);
                acc[t] = ds_mfma_32x32x16_bf16(a_hi, b_hi, acc[t]);
            }
        }
    }

    // ---- partial[sp][tap][co][ci] ----
    const int co0 = cot * WB_C + co_sub * 32, ci0 = cit * WB_C + ci_sub * 32;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        const int tap = (TG == 9) ? t : tg * 5 + t;
        float *dst = p.partial + (((size_t)sp * p.KS * p.KS + tap) * p.Cout) * p.Cin;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            dst[(size_t)co * p.Cin + ci0 + l31] = acc[t][r];
        }
    }
}

struct WgradPlanB {
    WgradKB k;
    int tg, n_tg;
    int grid;
    size_t lds_bytes;
    long long partial_floats;
};

static int plan_wgrad_b(WgradPlanB &pl, const ds_conv_shape *s) {
    DS_REQUIRE(s != nullptr, DS_ERR_NULL);
    DS_REQUIRE(s->B > 0 && s->H > 0 && s->W > 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(s->KS == 3 || s->KS == 5, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->stride == 1 || s->stride == 2, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->Cin % WB_C == 0 && s->Cout % WB_C == 0, DS_ERR_BAD_SHAPE);
    WgradKB &k = pl.k;
    const int pad = s->KS / 2;
    k.H = s->H; k.W = s->W; k.Cin = s->Cin; k.Cout = s->Cout;
    k.Ho = (s->H + 2 * pad - s->KS) / s->stride + 1;
    k.Wo = (s->W + 2 * pad - s->KS) / s->stride + 1;
    DS_REQUIRE(k.Ho > 0 && k.Wo > 0 && k.Wo <= 64, DS_ERR_BAD_SHAPE);
    DS_REQUIRE((long long)s->B * s->H * s->W * s->Cin < (1ll << 31), DS_ERR_BAD_SHAPE);
    DS_REQUIRE((long long)s->B * k.Ho * k.Wo * s->Cout < (1ll << 31), DS_ERR_BAD_SHAPE);
    k.KS = s->KS; k.IS = s->stride; k.pad = pad;
    pl.tg = s->KS == 3 ? 9 : 5;
    pl.n_tg = s->KS == 5 ? 5 : 1;
    // segment height / segments per tile: <= 64 output pixels (4 staging slots) and <= 192 halo pixels
    // (12 slots) per tile -- two workgroups of <= 80 KiB per CU
    const int max_in_pix = 190;                           // 12 staging slots, (64 + 190) records <= 80 KiB
    int best_rt = 0, best_ni = 1;
    for (int rt = 1; rt <= k.Ho; ++rt) {
        if (rt * k.Wo > 64) break;
        const int rows_in = s->KS == 5 ? rt : s->stride * (rt - 1) + s->KS, cols_in = s->stride * (k.Wo - 1) + s->KS;
        if (rows_in * cols_in > max_in_pix) break;
        best_rt = rt;
    }
    DS_REQUIRE(best_rt > 0, DS_ERR_UNSUPPORTED);
    k.RT = best_rt;
    k.segs_per_img = ds_ceil_div(k.Ho, best_rt);
    k.n_segs = s->B * k.segs_per_img;
    k.rows_in = s->KS == 5 ? best_rt : s->stride * (best_rt - 1) + s->KS;   // 5x5: one input row per output row and kernel row
    k.cols_in = s->stride * (k.Wo - 1) + s->KS;
    k.seg_pix = k.rows_in * k.cols_in;
    while ((best_ni + 1) * best_rt * k.Wo <= 64 && (best_ni + 1) * k.seg_pix <= max_in_pix && best_ni + 1 <= k.n_segs)
        ++best_ni;
    k.NI = best_ni;
    k.P = (best_ni * best_rt * k.Wo + 15) & ~15;
    k.n_tiles = ds_ceil_div(k.n_segs, best_ni);
    k.n_co_tiles = s->Cout / WB_C;
    k.n_ci_tiles = s->Cin / WB_C;
    const int base_blocks = pl.n_tg * k.n_co_tiles * k.n_ci_tiles;
    int S = ds_ceil_div(512, base_blocks);                // two workgroups per CU
    if (S > k.n_tiles) S = k.n_tiles;
    if (S < 1) S = 1;
    k.S = S;
    pl.grid = base_blocks * S;
    pl.lds_bytes = ((size_t)k.P + (size_t)k.NI * k.seg_pix) * WB_REC + ((size_t)k.P + 8 * k.NI) * 4;
    DS_REQUIRE(k.P * (WB_C / 4) <= 4 * 256 && k.NI <= 255 && s->stride * k.rows_in + s->KS < 4096 &&
                   pl.lds_bytes <= 80 * 1024, DS_ERR_UNSUPPORTED);
    pl.partial_floats = (long long)S * s->KS * s->KS * s->Cout * s->Cin;
    return DS_OK;
}

}  // namespace

extern "C" long long ds_conv_wgrad_bf16_workspace_floats(const ds_conv_shape *s) {
    WgradPlanB pl;
    int rc = plan_wgrad_b(pl, s);
    return rc == DS_OK ? pl.partial_floats : rc;
}

extern "C" int ds_conv_wgrad_bf16(const ds_conv_shape *s, const float *x, const float *gy, float *workspace,
                                  float *gw_oihw, void *stream) {
    DS_REQUIRE(s && x && gy && workspace && gw_oihw, DS_ERR_NULL);
    DS_REQUIRE(DS_ALIGNED16(x) && DS_ALIGNED16(gy), DS_ERR_ALIGNMENT);
    WgradPlanB pl;
    int rc = plan_wgrad_b(pl, s);
    if (rc != DS_OK) return rc;
    pl.k.x = x; pl.k.gz = gy; pl.k.partial = workspace;
    if (pl.tg == 9) DS_LAUNCH_BIG_LDS((wgrad_mfma_bf16_kernel<9>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    else DS_LAUNCH_BIG_LDS((wgrad_mfma_bf16_kernel<5>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    rc = ds_last_launch_error();
    if (rc) return rc;
    const long long n = (long long)s->KS * s->KS * s->Cout * s->Cin;
    long long g = (n + 255) / 256;
    DS_LAUNCH(wgrad_reduce_kernel, (int)(g > 4096 ? 4096 : g), 256, 0, stream, (const float *)workspace, gw_oihw,
              pl.k.S, s->KS * s->KS, s->Cout, s->Cin, 0);
    return ds_last_launch_error();
}
