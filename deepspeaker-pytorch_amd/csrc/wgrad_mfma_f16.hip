// wgrad_mfma_f16.hip -- filter gradients of the 3x3 / 5x5 convolution layers for the OPT-IN fp16 training step
// (train_f16.hip): fp16 activations and fp16 (loss-scaled) output gradients in HBM, ONE v_mfma_f32_32x32x16_f16 per
// product, f32 accumulation, f32 result un-scaled in the fixed-order fold.
//
// dW[co][ci][kh][kw] = sum over (b, h, w) of dY[b,h,w,co] * X[b, s*h+kh-p, s*w+kw-p, ci]
// (autograd of nn.Conv2d under loss.backward(), reference train_triplet.py:223; layers model.py:47-50, 98-106).
// Same decomposition as the split-operand bf16 kernel (wgrad_mfma_bf16.hip: workgroup = (64 co, 64 ci, pixel split),
// one accumulator per tap, operands fetched with ds_read_b64_tr_b16 from pixel-major LDS records, 5x5 as two launches
// over kernel-row groups).  What fp16 tensors change: staging is a 16-byte copy (8 channels, no conversion, no hi / lo
// halves), a pixel record is 128 B of data + 64 B of pad (conflict-free for the transposing reads) instead of 320 B, so
// tiles are twice as many pixels (256 for a 3x3) at the same LDS footprint, and a tap is one MFMA instead of three.
#include <ds_device.h>
#include "ds_common.h"
#include "wgrad_reduce.h"

namespace {

constexpr int WB_C = 64;                     // channels per tile on both sides
constexpr int WB_REC = 2 * WB_C + 64;        // bytes per pixel record: 64 halfs | pad -> 48 dwords: four consecutive records (and their
                                             // second 16-channel block, 8 dwords on) start in eight different 8-dword bank groups
// staging slots per thread (16-byte items of 8 channels: 32 pixels per slot): a 3x3 tile is up to 256 output pixels and 446
// halo pixels, a 5x5 kernel-row group up to 128 and 382
constexpr int WH_GSL3 = 8, WH_XSL3 = 14, WH_GSL5 = 4, WH_XSL5 = 12;

struct WgradKH {
    const _Float16 *x, *gz;
    float *partial;
    int H, W, Cin, Ho, Wo, Cout;
    int KS, IS, pad;
    int RT, NI, segs_per_img, n_segs, n_tiles;
    int rows_in, cols_in, seg_pix;
    int P;                       // output-pixel slots per tile (multiple of 16, >= NI*RT*Wo)
    int S, n_co_tiles, n_ci_tiles;
    unsigned x_bytes, gz_bytes;  // extents of x / gz (32-bit buffer offsets)
    int k0;                      // first kernel row of the group (5x5)
};

// one f16x8 MFMA operand: pixels q0 .. q0+7 of this lane's channel; rec0 / rec1 are the byte addresses
// this lane supplies for the two 4-pixel blocks (its piece: pixel q0 + 4r + ((lane&15)>>2), quad lane&3)
__device__ __forceinline__ f16x8 frag_tr(const char *rec0, const char *rec1) {
    const f16x4 a = __builtin_bit_cast(f16x4, ds_read_tr16_b64(rec0));
    const f16x4 b = __builtin_bit_cast(f16x4, ds_read_tr16_b64(rec1));
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

// TG taps per workgroup, KW taps per kernel row: <9, 3> = all of a 3x3; a 5x5 runs as two launches over kernel-row groups
// (<15, 5>: rows k0, k0 + stride, k0 + 2 stride; <10, 5>: the other two) -- see wgrad_mfma_bf16.hip.  Four waves as
// 2 (co) x 2 (ci), each owning a 32 x 32 block of every tap of the group (one wave per SIMD).
// GSL / XSL: 16-byte staging slots per thread for the dY rows and the X halo tile.
template <int TG, int KW, int GSL, int XSL>
__global__ void __launch_bounds__(256) DS_ONE_WAVE_PER_SIMD wgrad_mfma_f16_kernel(const WgradKH p) {
    constexpr bool GROUP = KW == 5;                     // kernel-row group of a 5x5 (see above)
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
    constexpr int QV = WB_C / 8;                        // 16-byte items (8 halfs) per staged pixel

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
            g_rel[it] = (r * p.Wo + c) * p.Cout + cot * WB_C + q * 8;
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
            const int hrel = GROUP ? p.IS * rr + p.k0 : rr;            // image row = IS*r0 - pad + hrel
            const int w = cc - p.pad;
            x_sr[it] = (w >= 0 && w < p.W) ? ((seg << 16) | hrel) : -2;
            x_rel[it] = (hrel * p.W + cc) * p.Cin + cit * WB_C + q * 8;
        }
    }
    for (int pp = tid; pp < p.P; pp += 256) {
        const int seg = pp / pix_per_seg, rem = pp - seg * pix_per_seg;
        const int r = rem / p.Wo, c = rem - r * p.Wo;
        // GROUP: tile row j is image row IS*(r0 + j) - pad + k0, output row r's first tap sits in tile row r
        pixtab[pp] = (seg < p.NI) ? (seg * p.seg_pix + (GROUP ? r : p.IS * r) * p.cols_in + p.IS * c) * WB_REC : 0;
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
    // Branch-free: a slot's segment entry is read from LDS, its validity folded into the offset (out-of-range
    // offsets of a raw buffer load return 0) -- sixteen independent loads per thread instead of sixteen
    // read -> compare -> branch -> load chains.
    const ds_buffer gbuf = ds_make_buffer(p.gz, p.gz_bytes), xbuf = ds_make_buffer(p.x, p.x_bytes);
    auto issue_loads = [&](int buf) {
        const int *st = segtab + buf * p.NI * 4;
        int g_org[GSL], g_rows[GSL], x_org[XSL], x_h0[XSL];
#pragma unroll
        for (int it = 0; it < GSL; ++it) {                 // every slot's segment entry, requested unconditionally
            const int *e = st + (g_sr[it] >= 0 ? (g_sr[it] >> 16) : 0) * 4;
            g_org[it] = e[0];
            g_rows[it] = e[1];
        }
#pragma unroll
        for (int it = 0; it < XSL; ++it) {
            const int *e = st + (x_sr[it] >= 0 ? (x_sr[it] >> 16) : 0) * 4;
            x_org[it] = e[2];
            x_h0[it] = e[3];
        }
#pragma unroll
        for (int it = 0; it < GSL; ++it) {                 // (the reads above must not sink into per-slot branches)
            DS_OPAQUE_VGPR(g_org[it]);
            DS_OPAQUE_VGPR(g_rows[it]);
        }
#pragma unroll
        for (int it = 0; it < XSL; ++it) {
            DS_OPAQUE_VGPR(x_org[it]);
            DS_OPAQUE_VGPR(x_h0[it]);
        }
#pragma unroll
        for (int it = 0; it < GSL; ++it) {
            const bool ok = (g_sr[it] >= 0) & ((g_sr[it] & 0xFFFF) < g_rows[it]);
            gv[it] = ds_buffer_load_f32x4(gbuf, ok ? (unsigned)(g_org[it] + g_rel[it]) * 2u : DS_BUFFER_OOB);
        }
#pragma unroll
        for (int it = 0; it < XSL; ++it) {
            const int h = x_h0[it] + (x_sr[it] & 0xFFFF);
            const bool ok = (x_sr[it] >= 0) & (h >= 0) & (h < p.H);
            xv[it] = ds_buffer_load_f32x4(xbuf, ok ? (unsigned)(x_org[it] + x_rel[it]) * 2u : DS_BUFFER_OOB);
        }
    };
    auto put = [&](char *rec, int q, const f32x4 v) { *(f32x4 *)(rec + q * 16) = v; };      // 8 channels, as they are

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
                put(gzt + (size_t)(i / QV) * WB_REC, i % QV, gv[it]);
            }
#pragma unroll
        for (int it = 0; it < XSL; ++it)
            if (x_sr[it] != -1) {
                const int i = tid + it * 256;
                put(xt + (size_t)(i / QV) * WB_REC, i % QV, xv[it]);
            }
        fill_segtab(tile + p.S, buf ^ 1);
        __syncthreads();
        if (tile + p.S < p.n_tiles) issue_loads(buf ^ 1);   // in flight during this tile's matrix work
        // ---- contract: 16 pixels per MFMA, one accumulator per tap.  Nothing but this wave hides its own LDS latency:
        //      fragments travel AH taps ahead of their MFMAs through NS register slots (TG is a multiple of NS, so the
        //      slot pattern repeats every step); the next step's dY fragments are requested with its first tap ----
        constexpr int NS = (TG % 3 == 0) ? 3 : 2, AH = NS - 1;
        static_assert(TG % NS == 0, "slot pattern must repeat per step");
        auto tap_off = [&](int t) { return ((t / KW) * p.cols_in + (t % KW)) * WB_REC; };
        auto step_ptrs = [&](int s, const char *&g0, const char *&g1, const char *&x0, const char *&x1) {
            const int pp0 = s + 8 * lhi + piece_pix, pp1 = pp0 + 4;
            g0 = gzt + (size_t)pp0 * WB_REC + a_col;
            g1 = gzt + (size_t)pp1 * WB_REC + a_col;
            x0 = xt + pixtab[pp0] + b_col;
            x1 = xt + pixtab[pp1] + b_col;
        };
        const char *g0, *g1, *x0, *x1;
        step_ptrs(0, g0, g1, x0, x1);
        f16x8 a = frag_tr(g0, g1);
        f16x8 bf[NS];
#pragma unroll
        for (int t = 0; t < AH; ++t) bf[t] = frag_tr(x0 + tap_off(t), x1 + tap_off(t));
        for (int s = 0; s < p.P; s += 16) {
            const char *ng0, *ng1, *nx0, *nx1;
            step_ptrs(s + 16 < p.P ? s + 16 : s, ng0, ng1, nx0, nx1);      // last step: harmless re-reads of this one
            f16x8 na = a;
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const int ahead = t + AH, slot = ahead % NS;
                if (ahead < TG) {
                    bf[slot] = frag_tr(x0 + tap_off(ahead), x1 + tap_off(ahead));
                } else {
                    if (ahead == TG) na = frag_tr(ng0, ng1);
                    bf[slot] = frag_tr(nx0 + tap_off(ahead - TG), nx1 + tap_off(ahead - TG));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[t] = ds_mfma_32x32x16_f16(a, bf[t % NS], acc[t]);
                __builtin_amdgcn_sched_barrier(0);
            }
            a = na;
            x0 = nx0;
            x1 = nx1;
        }
    }

    // ---- partial[sp][tap][co][ci] ----
    const int co0 = cot * WB_C + co_sub * 32, ci0 = cit * WB_C + ci_sub * 32;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        const int tap = GROUP ? (p.k0 + p.IS * (t / KW)) * p.KS + (t % KW) : t;
        float *dst = p.partial + (((size_t)sp * p.KS * p.KS + tap) * p.Cout) * p.Cin;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            dst[(size_t)co * p.Cin + ci0 + l31] = acc[t][r];
        }
    }
}

struct WgradPlanH {
    WgradKH k;                   // 5x5: the geometry of the three-row group
    int grid;
    size_t lds_bytes;
    long long partial_floats;
};

static int plan_wgrad_h(WgradPlanH &pl, const ds_conv_shape *s) {
    DS_REQUIRE(s != nullptr, DS_ERR_NULL);
    DS_REQUIRE(s->B > 0 && s->H > 0 && s->W > 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(s->KS == 3 || s->KS == 5, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->stride == 1 || s->stride == 2, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->Cin % WB_C == 0 && s->Cout % WB_C == 0, DS_ERR_BAD_SHAPE);
    WgradKH &k = pl.k;
    const int pad = s->KS / 2;
    k.H = s->H; k.W = s->W; k.Cin = s->Cin; k.Cout = s->Cout;
    k.Ho = (s->H + 2 * pad - s->KS) / s->stride + 1;
    k.Wo = (s->W + 2 * pad - s->KS) / s->stride + 1;
    DS_REQUIRE(k.Ho > 0 && k.Wo > 0 && k.Wo <= 64, DS_ERR_BAD_SHAPE);
    DS_REQUIRE((long long)s->B * s->H * s->W * s->Cin < (1ll << 30), DS_ERR_BAD_SHAPE);      // 32-bit byte offsets
    DS_REQUIRE((long long)s->B * k.Ho * k.Wo * s->Cout < (1ll << 30), DS_ERR_BAD_SHAPE);
    k.x_bytes = (unsigned)((long long)s->B * s->H * s->W * s->Cin * 2);
    k.gz_bytes = (unsigned)((long long)s->B * k.Ho * k.Wo * s->Cout * 2);
    k.KS = s->KS; k.IS = s->stride; k.pad = pad;
    k.k0 = 0;
    const int group_rows = 3;                             // kernel rows of the (larger) 5x5 group
    // segment height / segments per tile: the kernel's staging slots bound the tile (32 pixels per slot)
    int max_out_pix = 0, max_in_pix = 0, best_rt = 0, best_ni = 1;
    // rows per segment: the most pixels per tile among the heights that waste the fewest rows in an image's last segment
    auto search = [&](int mo, int mi) {
        max_out_pix = mo; max_in_pix = mi;
        best_rt = 0; best_ni = 1;
        double best_fill = -1.0;
        for (int rt = 1; rt <= k.Ho; ++rt) {
            if (rt * k.Wo > max_out_pix) break;
            const int rows_in = s->KS == 5 ? rt + group_rows - 1 : s->stride * (rt - 1) + s->KS;
            const int cols_in = s->stride * (k.Wo - 1) + s->KS;
            if (rows_in * cols_in > max_in_pix) break;
            const int segs = ds_ceil_div(k.Ho, rt);
            const int padded = (rt * k.Wo + 15) & ~15;
            const double fill = (double)k.Ho * k.Wo / ((double)segs * padded) + 1e-6 * rt;
            if (s->KS == 5 || fill > best_fill) { best_fill = fill; best_rt = rt; }
        }
        if (best_rt == 0) return 0;
        const int segs_per_img = ds_ceil_div(k.Ho, best_rt);
        const int rows_in = s->KS == 5 ? best_rt + group_rows - 1 : s->stride * (best_rt - 1) + s->KS;
        const int seg_pix = rows_in * (s->stride * (k.Wo - 1) + s->KS);
        while ((best_ni + 1) * best_rt * k.Wo <= max_out_pix && (best_ni + 1) * seg_pix <= max_in_pix &&
               best_ni + 1 <= s->B * segs_per_img)
            ++best_ni;
        return best_ni * best_rt * k.Wo;                  // output pixels per tile
    };
    if (s->KS == 3) search(WH_GSL3 * 32, WH_XSL3 * 32 - 2);
    else search(WH_GSL5 * 32, WH_XSL5 * 32 - 2);
    DS_REQUIRE(best_rt > 0, DS_ERR_UNSUPPORTED);
    k.RT = best_rt;
    k.segs_per_img = ds_ceil_div(k.Ho, best_rt);
    k.n_segs = s->B * k.segs_per_img;
    k.rows_in = s->KS == 5 ? best_rt + group_rows - 1 : s->stride * (best_rt - 1) + s->KS;   // 5x5: the rows of one residue
    k.cols_in = s->stride * (k.Wo - 1) + s->KS;
    k.seg_pix = k.rows_in * k.cols_in;
    k.NI = best_ni;
    k.P = (best_ni * best_rt * k.Wo + 15) & ~15;
    k.n_tiles = ds_ceil_div(k.n_segs, best_ni);
    k.n_co_tiles = s->Cout / WB_C;
    k.n_ci_tiles = s->Cin / WB_C;
    const int base_blocks = k.n_co_tiles * k.n_ci_tiles;
    int S = ds_ceil_div(ds_cu_count(), base_blocks);      // one workgroup per CU, every one with the same share of the tiles
    if (S > k.n_tiles) S = k.n_tiles;
    if (S < 1) S = 1;
    k.S = S;
    pl.grid = base_blocks * S;
    pl.lds_bytes = ((size_t)k.P + (size_t)k.NI * k.seg_pix) * WB_REC + ((size_t)k.P + 8 * k.NI) * 4;
    DS_REQUIRE(k.P <= max_out_pix && k.NI * k.seg_pix <= max_in_pix + 2 && k.NI <= 255 &&
                   s->stride * k.rows_in + s->KS < 4096 && pl.lds_bytes <= 150 * 1024, DS_ERR_UNSUPPORTED);
    pl.partial_floats = (long long)S * s->KS * s->KS * s->Cout * s->Cin;
    return DS_OK;
}

}  // namespace

extern "C" long long ds_conv_wgrad_f16_workspace_floats(const ds_conv_shape *s) {
    WgradPlanH pl;
    int rc = plan_wgrad_h(pl, s);
    return rc == DS_OK ? pl.partial_floats : rc;
}

// gw_oihw = out_scale * sum over pixels of gy (x) x: x [B,H,W,Cin] fp16 activations, gy [B,Ho,Wo,Cout] fp16 output
// gradients in loss-scaled units (out_scale = 1 / S), workspace ds_conv_wgrad_f16_workspace_floats(s) floats.
extern "C" int ds_conv_wgrad_f16(const ds_conv_shape *s, const void *x_f16, const void *gy_f16, float *workspace,
                                 float *gw_oihw, float out_scale, void *stream) {
    DS_REQUIRE(s && x_f16 && gy_f16 && workspace && gw_oihw, DS_ERR_NULL);
    DS_REQUIRE(DS_ALIGNED16(x_f16) && DS_ALIGNED16(gy_f16), DS_ERR_ALIGNMENT);
    WgradPlanH pl;
    int rc = plan_wgrad_h(pl, s);
    if (rc != DS_OK) return rc;
    pl.k.x = (const _Float16 *)x_f16; pl.k.gz = (const _Float16 *)gy_f16; pl.k.partial = workspace;
    if (s->KS == 3) {
        DS_LAUNCH_BIG_LDS((wgrad_mfma_f16_kernel<9, 3, WH_GSL3, WH_XSL3>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    } else {
        // kernel rows 0, s, 2s (15 taps), then the remaining two (10 taps): same tiles, same splits, disjoint taps
        DS_LAUNCH_BIG_LDS((wgrad_mfma_f16_kernel<15, 5, WH_GSL5, WH_XSL5>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
        rc = ds_last_launch_error();
        if (rc) return rc;
        WgradKH k2 = pl.k;
        k2.k0 = s->stride == 2 ? 1 : 3;
        k2.rows_in = pl.k.RT + 1;
        k2.seg_pix = k2.rows_in * k2.cols_in;
        DS_LAUNCH_BIG_LDS((wgrad_mfma_f16_kernel<10, 5, WH_GSL5, WH_XSL5>), pl.grid, 256, pl.lds_bytes, stream, k2);
    }
    rc = ds_last_launch_error();
    if (rc) return rc;
    const long long n = (long long)s->KS * s->KS * s->Cout * s->Cin;
    int lg, rgrid;
    wgrad_reduce_shape(n, pl.k.S, lg, rgrid);
    DS_LAUNCH(wgrad_reduce_kernel, rgrid, 256, 1024, stream, (const float *)workspace, gw_oihw,
              pl.k.S, s->KS * s->KS, s->Cout, s->Cin, 0, out_scale, lg);
    return ds_last_launch_error();
}
