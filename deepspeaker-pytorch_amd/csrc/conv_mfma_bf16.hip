// conv_mfma_bf16.hip -- implicit-GEMM convolution on the gfx950 bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, f32 accumulate), the throughput variants of conv_mfma_f32.hip for the
// forward convolutions of reference model.py:69,73,192,197,202 (+ the same fused BatchNorm-affine /
// residual / clipped-ReLU epilogue, model.py:70-80,188-205).
//
//   X3 = false ("bf16"):   one MFMA per k-step, operands rounded to bf16.  ~8 bits of mantissa: the
//                          embedding drifts ~6e-3 from the reference -- a speed mode.
//   X3 = true  ("bf16x3"): every f32 operand is split x = hi + lo (two bf16) and the product is taken
//                          as hi*hi + hi*lo + lo*hi on the matrix cores -- f32-class accuracy (~1e-5
//                          relative per term) at up to 1/3 of the bf16 peak, i.e. > 5x the f32-MFMA rate.
//
// Activations stay f32 channels-last in HBM; they are converted (and split) while being staged into
// LDS, so the kernel is a drop-in for ds_conv_fwd_f32.  Tiling / segment / halo logic is the same as the
// f32 kernel's (see that file); differences: 16 input channels per chunk (= one MFMA k-step per tap),
// LDS pixel records of 16 bf16 (+16 B pad, conflict-free ds_read_b128), no register prefetch of the
// activation tile (two workgroups per CU overlap staging with the other's matrix work).
#include <ds_device.h>
#include <algorithm>
#include "ds_common.h"

#include "conv_mfma_bf16_kernel.h"

namespace {

// OIHW f32 -> [Cin/16][tap][Cout][16] bf16 hi (+ lo = bf16(w - hi))
// dgrad != 0: the transposed, spatially flipped bank of the stride-1 data gradient (N = Cin, K = Cout)
__device__ __forceinline__ void pack_conv_weight_bf16_body(const float *w, __bf16 *hi, __bf16 *lo, int Cout, int Cin, int KS,
                                                           int dgrad, int block, int n_blocks) {
    const int T = KS * KS;
    const long long n = (long long)Cout * Cin * T;
    const int N = dgrad ? Cin : Cout;
    for (long long i = (long long)block * 256 + threadIdx.x; i < n; i += (long long)n_blocks * 256) {
        const int kk = (int)(i & 15);
        long long r = i >> 4;
        const int nn = (int)(r % N);
        r /= N;
        const int t = (int)(r % T);
        const int kc = (int)(r / T);
        const int k = kc * 16 + kk;
        const int tt = dgrad ? (T - 1 - t) : t;
        const int kh = tt / KS, kw = tt - kh * KS;
        const int co = dgrad ? k : nn, ci = dgrad ? nn : k;
        const float v = w[(((size_t)co * Cin + ci) * KS + kh) * KS + kw];
        const __bf16 h = (__bf16)v;
        hi[i] = h;
        if (lo) lo[i] = (__bf16)(v - (float)h);
    }
}

__global__ void __launch_bounds__(256) pack_conv_weight_bf16_kernel(const float *w, __bf16 *hi, __bf16 *lo, int Cout,
                                                                    int Cin, int KS, int dgrad) {
    pack_conv_weight_bf16_body(w, hi, lo, Cout, Cin, KS, dgrad, (int)blockIdx.x, (int)gridDim.x);
}

// all filters of a weight version in one launch: job j owns the workgroups [first[j], first[j + 1])
struct PackBatchB {
    const float *w[DS_PACK_BATCH_MAX];
    __bf16 *hi[DS_PACK_BATCH_MAX], *lo[DS_PACK_BATCH_MAX];
    int Cout[DS_PACK_BATCH_MAX], Cin[DS_PACK_BATCH_MAX], KS[DS_PACK_BATCH_MAX], dgrad[DS_PACK_BATCH_MAX];
    int first[DS_PACK_BATCH_MAX + 1];
    int n;
};

__global__ void __launch_bounds__(256) pack_conv_weight_bf16_batch_kernel(const PackBatchB J) {
    int j = 0;
    while (j + 1 < J.n && (int)blockIdx.x >= J.first[j + 1]) ++j;
    pack_conv_weight_bf16_body(J.w[j], J.hi[j], J.lo[j], J.Cout[j], J.Cin[j], J.KS[j], J.dgrad[j], (int)blockIdx.x - J.first[j],
                               J.first[j + 1] - J.first[j]);
}

// LDS cycles (1 = conflict-free) of one ds_read_b128 fragment read for a candidate row pitch: simulates
// the two 16-lane service groups of lanes 0..31 (the upper half-wave behaves identically; each group
// holds 16 consecutive pixels, see `lpix` in the kernel) over every 32-row sub-tile of the M tile.  Bank slot of a record = (record * PSB / 16) mod 16.
static double frag_read_cost(int MT, int NI, int RT, int Wc, int IS, int rows_in, int pitch) {
    const int pix_per_seg = RT * Wc, seg_pix = rows_in * pitch;
    double total = 0.0;
    int n = 0;
    for (int m0 = 0; m0 + 32 <= MT; m0 += 32) {
        for (int g = 0; g < 2; ++g) {
            int cnt[16] = {0};
            int worst = 0;
            for (int j = 0; j < 16; ++j) {
                const int m = m0 + 16 * g + j;       // the lane -> pixel permutation makes a group's pixels consecutive
                const int seg = m / pix_per_seg, rem = m % pix_per_seg;
                const int r = rem / Wc, c = rem % Wc;
                const int rec = (seg < NI) ? seg * seg_pix + (IS * r) * pitch + c : 0;
                const int slot = (rec * (PSB / 16)) & 15;
                if (++cnt[slot] > worst) worst = cnt[slot];
            }
            total += worst;
            ++n;
        }
    }
    return n ? total / n : 1.0;
}

// ---- host-side plan (same objective as the f32 planner; limits: 64 KiB LDS, 2 workgroups per CU) ----
struct TileCfgB { int MT, NTILE, WM, wg_per_cu, NTHR; };
constexpr int kNumCfgB = 9;
constexpr TileCfgB kCfgB[kNumCfgB] = {
    {128, 64, 2, 3, 256},      // <KS,2,1,2,2>
    {160, 128, 1, 2, 256},     // <KS,5,1,1,4>
    {256, 64, 2, 2, 256},      // <KS,4,1,2,2>
    {160, 128, 1, 2, 128},     // <KS,5,2,1,2>: two waves, 160x64 register tile each, one wave per SIMD; up to 80 KiB LDS
    // one workgroup per CU, four waves with a 160x64 register tile each and the whole 160 KiB of LDS: the
    // stride-2 layers, whose input tile is 4x the output tile, keep full M tiles this way
    {160, 256, 1, 1, 256},     // <KS,5,2,1,4>
    {320, 128, 2, 1, 256},     // <KS,5,2,2,2>
    {320, 64, 2, 2, 128},      // <KS,5,2,2,1>: the 2-wave shape for 64-channel layers
    {128, 128, 1, 2, 128},     // <KS,4,2,1,2>: 128x64 register tiles where 160-row tiles quantise badly
    {128, 256, 1, 1, 256},     // <KS,4,2,1,4>: the same with the whole LDS (three 10x4 maps of the last 5x5 layer)
};
constexpr size_t kLdsCapB[kNumCfgB] = {64 * 1024, 64 * 1024, 64 * 1024, 80 * 1024, 160 * 1024 - 64, 160 * 1024 - 64,
                                       80 * 1024, 80 * 1024, 160 * 1024 - 64};

// bytes of the epilogue's per-wave transposition buffers (they alias the pixel tile)
static size_t epi_bytes(const TileCfgB &cf) {
    const int waves = cf.NTHR / 64, nsub = cf.NTILE / (waves / cf.WM) / 32;
    return (size_t)waves * 32 * (nsub * 32 + 4) * 4;
}


static int g_forced_cfg_b = -1;      // tuning hook (tools/bf16_cfg_ab.py): plan with this tile configuration only

// `s` describes the tile grid: input [B,H,W,Cin], KS x KS taps, stride s->stride, output grid Ho x Wo computed
// with `pad`.  The forward convolution writes that grid densely; the stride-2 data gradient runs four such
// grids (parity classes) whose outputs interleave in y (out_* arguments).
static int plan_bf16(PlanB &pl, const ds_conv_shape *s, bool x3, int out_stride = 1, int out_h0 = 0, int out_w0 = 0,
                     int out_H = 0, int out_W = 0, int grid_H = 0, int grid_W = 0) {
    DS_REQUIRE(s != nullptr, DS_ERR_NULL);
    DS_REQUIRE(s->B > 0 && s->H > 0 && s->W > 0 && s->Cin > 0 && s->Cout > 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(s->KS == 3 || s->KS == 5, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->stride == 1 || s->stride == 2, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->Cin % CKB == 0 && s->Cout % 64 == 0, DS_ERR_BAD_SHAPE);
    const int pad = s->KS / 2;
    const int Ho = grid_H > 0 ? grid_H : (s->H + 2 * pad - s->KS) / s->stride + 1;
    const int Wo = grid_W > 0 ? grid_W : (s->W + 2 * pad - s->KS) / s->stride + 1;
    const int yH = out_H > 0 ? out_H : Ho, yW = out_W > 0 ? out_W : Wo;
    DS_REQUIRE(Ho > 0 && Wo > 0 && Wo <= 128, DS_ERR_BAD_SHAPE);
    DS_REQUIRE((long long)s->B * s->H * s->W * s->Cin < (1ll << 31), DS_ERR_BAD_SHAPE);
    DS_REQUIRE((long long)s->B * Ho < (1ll << 24), DS_ERR_BAD_SHAPE);                  // reciprocal index arithmetic
    DS_REQUIRE((long long)s->B * yH * yW * s->Cout < (1ll << 30), DS_ERR_BAD_SHAPE);   // 32-bit byte offsets
    const int IS = s->stride;
    const bool big_tile_ok = x3;                 // the 2-wave shape is tuned for (and only built for) bf16x3
    double best = -1.0;
    int bc = -1, brt = 0, bni = 0;
    for (int c = 0; c < kNumCfgB; ++c) {
        const TileCfgB &cf = kCfgB[c];
        if (s->Cout % cf.NTILE) continue;
        if (c >= 3 && !big_tile_ok) continue;
        if (g_forced_cfg_b >= 0 && c != g_forced_cfg_b) continue;
        const size_t lds_cap = kLdsCapB[c];
        const long long item_cap = c >= 3 ? 32 * cf.NTHR : 16 * cf.NTHR;
        for (int rt = 1; rt <= Ho; ++rt) {
            if ((long long)rt * Wo > cf.MT) break;
            const int segs_per_img = ds_ceil_div(Ho, rt);
            const long long n_segs = (long long)s->B * segs_per_img;
            int ni = cf.MT / (rt * Wo);
            if (ni > n_segs) ni = (int)n_segs;
            const int rows_in = IS * (rt - 1) + s->KS, cols_in = IS * (Wo - 1) + s->KS;
            // feasibility with the widest row pitch the conflict search may pick (cols_in + 4); the pitch
            // itself is chosen once, for the winning geometry (this function runs on every launch)
            auto lds_of = [&](int n) {
                const size_t tp = (size_t)n * rows_in * (cols_in + 4);
                return std::max(tp * PSB * (x3 ? 2 : 1), epi_bytes(cf)) + (size_t)cf.MT * 4 + (size_t)n * 8 + (size_t)cf.WM * cf.NTILE * 8;
            };
            // staged items: 4-channel quarters of the in-image pixels only
            auto items_of = [&](int n) { return (long long)n * std::min(rows_in, s->H) * s->W * (CKB / 4); };
            while (ni > 1 && (lds_of(ni) > lds_cap || items_of(ni) > item_cap)) --ni;
            if (lds_of(ni) > lds_cap || items_of(ni) > item_cap) continue;
            const long long n_mt = ds_ceil_div_ll(n_segs, ni);
            double eff = (double)s->B * Ho * Wo / ((double)n_mt * cf.MT);
            const long long blocks = n_mt * (s->Cout / cf.NTILE), slots = 256ll * cf.wg_per_cu;
            if (blocks <= slots) eff *= (double)blocks / (double)(ds_ceil_div_ll(blocks, 256) * 256);
            else eff *= (double)blocks / (double)(ds_ceil_div_ll(blocks, slots) * slots);
            // a launch with fewer waves than the chip has SIMDs leaves matrix cores idle: 256 two-wave workgroups (the
            // 10x4 layers of one 256-utterance member) ran 156 us where 256 four-wave ones ran 128 (tools/conv_bf16_ab.py)
            const long long waves = blocks * (cf.NTHR / 64);
            if (waves < 1024) eff *= (double)waves / 1024.0;
            // ties: 160x64 register tiles first; for 64 output channels the four-wave 256x64 tile beats the two-wave
            // 320x64 one (same tool: 484 against 525 us at 768 utterances, 173 against 185 at 256)
            int pref[kNumCfgB] = {0, 2, 7, 8, 5, 6, 1, 4, 3};
#ifdef DS_BF16_PREF_TOP
            pref[DS_BF16_PREF_TOP] = 9;                               // A/B builds (tools/conv_bf16_ab.py): another tie-break
#endif
            eff += 1e-9 * rt + 1e-6 * pref[c];
            if (eff > best) { best = eff; bc = c; brt = rt; bni = ni; }
        }
    }
    if (bc < 0) return DS_ERR_UNSUPPORTED;
    const TileCfgB &cf = kCfgB[bc];
    ConvKB &k = pl.k;
    k.H = s->H; k.W = s->W; k.Cin = s->Cin;
    k.Hr = Ho; k.Wc = Wo; k.Ho = yH; k.Wo = yW; k.Cout = s->Cout;
    k.OS = out_stride; k.OH0 = out_h0; k.OW0 = out_w0;
    k.IS = IS; k.dh_min = -pad; k.dw_min = -pad;
    k.RT = brt; k.NI = bni;
    k.segs_per_img = ds_ceil_div(Ho, brt);
    k.n_segs = s->B * k.segs_per_img;
    k.rows_in = IS * (brt - 1) + s->KS;
    k.cols_in = IS * (Wo - 1) + s->KS;
    k.half = (k.cols_in + 1) / 2;
    // row pitch: the cheapest fragment read among cols_in .. cols_in+4 records per row
    int best_pitch = k.cols_in;
    double best_cost = 1e30;
    for (int pt = k.cols_in; pt <= k.cols_in + 4; ++pt) {
        const double c = frag_read_cost(cf.MT, bni, brt, Wo, IS, k.rows_in, pt);
        if (c < best_cost - 1e-9) { best_cost = c; best_pitch = pt; }
    }
    k.pitch = best_pitch;
    k.seg_pix = k.rows_in * k.pitch;
    k.n_ntiles = s->Cout / cf.NTILE;
    k.y_bytes = (unsigned)((long long)s->B * yH * yW * s->Cout * 4);
    pl.cfg = bc;
    pl.n_mtiles = ds_ceil_div(k.n_segs, bni);
    pl.grid = pl.n_mtiles * k.n_ntiles;
    const size_t tp = (size_t)k.NI * k.seg_pix;
    pl.lds_bytes = std::max(tp * PSB * (x3 ? 2 : 1), epi_bytes(cf)) + (size_t)cf.MT * 4 + (size_t)k.NI * 8 + (size_t)cf.WM * cf.NTILE * 8 + 16;
    pl.nit = ds_ceil_div(k.NI * std::min(k.rows_in, s->H) * s->W * (CKB / 4), cf.NTHR);
    return DS_OK;
}

// kernel size x arithmetic -> the translation unit that holds those instantiations
template <int KS, bool X3>
static void launch_b(const PlanB &pl, void *stream) {
    if (KS == 3) { if (X3) ds_bf16_launch_k3x3(pl, stream); else ds_bf16_launch_k3x1(pl, stream); }
    else         { if (X3) ds_bf16_launch_k5x3(pl, stream); else ds_bf16_launch_k5x1(pl, stream); }
}

}  // namespace

static int pack_bf16(const float *w_oihw, void *w_hi, void *w_lo, int Cout, int Cin, int KS, int dgrad, void *stream) {
    DS_REQUIRE(w_oihw && w_hi, DS_ERR_NULL);
    DS_REQUIRE(Cout > 0 && Cin > 0 && (KS == 3 || KS == 5) && ((dgrad ? Cout : Cin) % CKB) == 0, DS_ERR_BAD_SHAPE);
    const long long n = (long long)Cout * Cin * KS * KS;
    long long g = (n + 255) / 256;
    DS_LAUNCH(pack_conv_weight_bf16_kernel, (int)(g > 4096 ? 4096 : g), 256, 0, stream, w_oihw, (__bf16 *)w_hi,
              (__bf16 *)w_lo, Cout, Cin, KS, dgrad);
    return ds_last_launch_error();
}

extern "C" int ds_pack_conv_weights_bf16_batch(const ds_pack_job *jobs, int n_jobs, void *stream) {
    DS_REQUIRE(jobs != nullptr, DS_ERR_NULL);
    DS_REQUIRE(n_jobs > 0 && n_jobs <= DS_PACK_BATCH_MAX, DS_ERR_BAD_SHAPE);
    PackBatchB J;
    J.n = n_jobs;
    int blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const ds_pack_job &b = jobs[j];
        DS_REQUIRE(b.w_oihw && b.out, DS_ERR_NULL);
        DS_REQUIRE((b.mode == 0 || b.mode == 1) && b.Cout > 0 && b.Cin > 0, DS_ERR_BAD_SHAPE);
        DS_REQUIRE(b.KS == 3 || b.KS == 5, DS_ERR_UNSUPPORTED);
        DS_REQUIRE((b.mode ? b.Cout : b.Cin) % 16 == 0, DS_ERR_BAD_SHAPE);
        J.w[j] = b.w_oihw; J.hi[j] = (__bf16 *)b.out; J.lo[j] = (__bf16 *)b.out2;
        J.Cout[j] = b.Cout; J.Cin[j] = b.Cin; J.KS[j] = b.KS; J.dgrad[j] = b.mode;
        const long long n = (long long)b.Cout * b.Cin * b.KS * b.KS;
        const long long g = (n + 255) / 256;
        J.first[j] = blocks;
        blocks += (int)(g > 512 ? 512 : g);
    }
    J.first[n_jobs] = blocks;
    DS_LAUNCH(pack_conv_weight_bf16_batch_kernel, blocks, 256, 0, stream, J);
    return ds_last_launch_error();
}

extern "C" int ds_pack_conv_weight_bf16(const float *w_oihw, void *w_hi, void *w_lo, int Cout, int Cin, int KS,
                                        void *stream) {
    return pack_bf16(w_oihw, w_hi, w_lo, Cout, Cin, KS, 0, stream);
}

extern "C" int ds_pack_conv_weight_dgrad_bf16(const float *w_oihw, void *w_hi, void *w_lo, int Cout, int Cin, int KS,
                                              void *stream) {
    return pack_bf16(w_oihw, w_hi, w_lo, Cout, Cin, KS, 1, stream);
}

extern "C" int ds_conv_bf16_stats_rows(const ds_conv_shape *s, int x3) {
    PlanB pl;
    int rc = plan_bf16(pl, s, x3 != 0);
    return rc == DS_OK ? pl.n_mtiles : rc;
}

// tuning hook: cfg in [0, 9) = plan every bf16 convolution with that tile configuration; anything else = the planner's choice
extern "C" void ds_conv_bf16_set_forced_cfg(int cfg) { g_forced_cfg_b = (cfg >= 0 && cfg < kNumCfgB) ? cfg : -1; }

extern "C" int ds_conv_bf16_plan_describe(const ds_conv_shape *s, int x3, int *out8) {
    DS_REQUIRE(out8 != nullptr, DS_ERR_NULL);
    PlanB pl;
    int rc = plan_bf16(pl, s, x3 != 0);
    if (rc != DS_OK) return rc;
    const TileCfgB &cf = kCfgB[pl.cfg];
    out8[0] = cf.MT; out8[1] = cf.NTILE; out8[2] = pl.k.RT; out8[3] = pl.k.NI;
    out8[4] = pl.grid; out8[5] = (int)pl.lds_bytes; out8[6] = cf.NTHR; out8[7] = pl.k.pitch;
    return DS_OK;
}

extern "C" int ds_conv_fwd_bf16(const ds_conv_shape *s, const float *x, const void *w_hi, const void *w_lo,
                                const float *scale, const float *shift, const float *residual, float *y,
                                float *stats_partial, int flags, void *stream) {
    DS_REQUIRE(x && w_hi && y, DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_AFFINE) || (scale && shift), DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_RESIDUAL) || residual, DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_STATS) || stats_partial, DS_ERR_NULL);
    DS_REQUIRE(DS_ALIGNED16(x) && DS_ALIGNED16(w_hi) && DS_ALIGNED16(y) && DS_ALIGNED16(w_lo), DS_ERR_ALIGNMENT);
    const bool x3 = w_lo != nullptr;
    PlanB pl;
    int rc = plan_bf16(pl, s, x3);
    if (rc != DS_OK) return rc;
    pl.k.x = x; pl.k.w_hi = (const __bf16 *)w_hi; pl.k.w_lo = (const __bf16 *)w_lo; pl.k.y = y;
    pl.k.scale = scale; pl.k.shift = shift; pl.k.res = residual; pl.k.stats = stats_partial;
    pl.k.flags = flags;
    pl.k.bn_z = pl.k.bn_mean = pl.k.bn_invstd = pl.k.bn_msc = pl.k.bn_msh = nullptr;
    pl.k.bn_mtiles = 1; pl.k.bn_rows_member = 0; pl.k.bn_row0 = 0;
    if (s->KS == 3) { if (x3) launch_b<3, true>(pl, stream); else launch_b<3, false>(pl, stream); }
    else            { if (x3) launch_b<5, true>(pl, stream); else launch_b<5, false>(pl, stream); }
    return ds_last_launch_error();
}

// banks of the 5x5 stride-2 data gradient: parity class (ph, pw) of dX only receives taps kh = ph (mod 2),
// kw = pw (mod 2), at input offsets d = (p + 2 - k) / 2 in {1, 0, -1} -- a 3x3 stride-1 convolution over dY
// per class; classes with two taps per axis carry zero filters for the third (window row/col 0, d = -1).
// Layout: [class ph*2+pw][Cout/16][tap (d_h+1)*3 + (d_w+1)][Cin][16] bf16 hi (+ lo).
__global__ void __launch_bounds__(256) pack_conv_dgrad_s2_bf16_kernel(const float *w, __bf16 *hi, __bf16 *lo, int Cout,
                                                                      int Cin) {
    const long long per_class = (long long)9 * Cout * Cin;
    const long long n = 4 * per_class;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int cls = (int)(i / per_class);
        long long r = i - cls * per_class;
        const int kk = (int)(r & 15);
        r >>= 4;
        const int ci = (int)(r % Cin);
        r /= Cin;
        const int t = (int)(r % 9);
        const int kc = (int)(r / 9);
        const int co = kc * 16 + kk;                    // contraction channel = forward output channel
        const int ph = cls >> 1, pw = cls & 1;
        const int dh = t / 3 - 1, dw = t % 3 - 1;       // input offset of this window position
        const int kh = ph + 2 - 2 * dh, kw = pw + 2 - 2 * dw;
        float v = 0.0f;
        if (kh >= 0 && kh < 5 && kw >= 0 && kw < 5) v = w[(((size_t)co * Cin + ci) * 5 + kh) * 5 + kw];
        const __bf16 h = (__bf16)v;
        hi[i] = h;
        lo[i] = (__bf16)(v - (float)h);
    }
}

extern "C" int ds_pack_conv_weight_dgrad_s2_bf16(const float *w_oihw, void *w_hi, void *w_lo, int Cout, int Cin,
                                                 void *stream) {
    DS_REQUIRE(w_oihw && w_hi && w_lo, DS_ERR_NULL);
    DS_REQUIRE(Cout > 0 && Cin > 0 && (Cout % CKB) == 0, DS_ERR_BAD_SHAPE);
    const long long n = (long long)36 * Cout * Cin;
    long long g = (n + 255) / 256;
    DS_LAUNCH(pack_conv_dgrad_s2_bf16_kernel, (int)(g > 4096 ? 4096 : g), 256, 0, stream, w_oihw, (__bf16 *)w_hi,
              (__bf16 *)w_lo, Cout, Cin);
    return ds_last_launch_error();
}

// data gradient on the bf16 matrix cores.  3x3 stride 1: the forward kernel on dY with the flipped /
// transposed bank (ds_pack_conv_weight_dgrad_bf16).  5x5 stride 2: four parity-class launches over the dY
// grid with the banks of ds_pack_conv_weight_dgrad_s2_bf16, outputs interleaved in dX.
extern "C" int ds_conv_dgrad_bf16(const ds_conv_shape *s, const float *gy, const void *w_hi, const void *w_lo,
                                  float *gx, void *stream) {
    DS_REQUIRE(s, DS_ERR_NULL);
    if (s->stride == 1) {
        DS_REQUIRE(s->KS == 3, DS_ERR_UNSUPPORTED);
        ds_conv_shape t = *s;
        t.Cin = s->Cout;
        t.Cout = s->Cin;
        return ds_conv_fwd_bf16(&t, gy, w_hi, w_lo, nullptr, nullptr, nullptr, gx, nullptr, 0, stream);
    }
    DS_REQUIRE(s->KS == 5 && s->stride == 2 && gy && w_hi && w_lo && gx, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(DS_ALIGNED16(gy) && DS_ALIGNED16(w_hi) && DS_ALIGNED16(w_lo) && DS_ALIGNED16(gx), DS_ERR_ALIGNMENT);
    const int Ho = (s->H - 1) / 2 + 1, Wo = (s->W - 1) / 2 + 1;          // dY grid
    const size_t bank = (size_t)9 * s->Cout * s->Cin;
    for (int cls = 0; cls < 4; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
        const int Hr = (s->H - ph + 1) / 2, Wc = (s->W - pw + 1) / 2;    // dX rows / columns of this parity
        if (Hr <= 0 || Wc <= 0) continue;
        ds_conv_shape t = {s->B, Ho, Wo, s->Cout, s->Cin, 3, 1};
        PlanB pl;
        int rc = plan_bf16(pl, &t, true, 2, ph, pw, s->H, s->W, Hr, Wc);
        if (rc != DS_OK) return rc;
        pl.k.x = gy;
        pl.k.w_hi = (const __bf16 *)w_hi + cls * bank;
        pl.k.w_lo = (const __bf16 *)w_lo + cls * bank;
        pl.k.y = gx;
        pl.k.scale = pl.k.shift = pl.k.res = nullptr;
        pl.k.stats = nullptr;
        pl.k.flags = 0;
        pl.k.bn_z = pl.k.bn_mean = pl.k.bn_invstd = pl.k.bn_msc = pl.k.bn_msh = nullptr;
        pl.k.bn_mtiles = 1; pl.k.bn_rows_member = 0; pl.k.bn_row0 = 0;
        launch_b<3, true>(pl, stream);
        rc = ds_last_launch_error();
        if (rc) return rc;
    }
    return DS_OK;
}


// The 3x3 stride-1 data gradient FUSED with the first half of the BatchNorm backward of the layer it feeds
// (autograd of clip(bn(conv(.))) under loss.backward(), reference train_triplet.py:223 over model.py:69-75,188-203):
//   gy = (dgrad(gz_up) [+ g2]) * [0 < z * mask_scale + mask_shift < 20],  partial[tile] = { sum gy, sum gy * xhat }
// `s` is the FORWARD shape of the convolution whose data gradient this is (as ds_conv_dgrad_bf16); z / gy / g2 are
// [B,H,W,Cin]; the batch consists of G members with their own statistics (tables [G][Cin]); the M tiles of the launch
// must not straddle members (ds_conv_dgrad_bnbwd_bf16_rows reports the partial rows per member, or an error).
static int plan_bnbwd(PlanB &pl, const ds_conv_shape *s, int G) {
    DS_REQUIRE(s, DS_ERR_NULL);
    DS_REQUIRE(s->KS == 3 && s->stride == 1 && G > 0 && s->B % G == 0, DS_ERR_UNSUPPORTED);
    ds_conv_shape t = *s;
    t.Cin = s->Cout;
    t.Cout = s->Cin;
    int rc = plan_bf16(pl, &t, true);
    if (rc != DS_OK) return rc;
    const long long segs_per_member = (long long)(s->B / G) * pl.k.segs_per_img;
    DS_REQUIRE(segs_per_member % pl.k.NI == 0, DS_ERR_UNSUPPORTED);
    pl.k.bn_mtiles = (int)(segs_per_member / pl.k.NI);
    pl.k.bn_rows_member = pl.k.bn_mtiles;
    pl.k.bn_row0 = 0;
    return DS_OK;
}

extern "C" int ds_conv_dgrad_bnbwd_bf16_rows(const ds_conv_shape *s, int G) {
    PlanB pl;
    int rc = plan_bnbwd(pl, s, G);
    return rc == DS_OK ? pl.k.bn_mtiles : rc;
}

extern "C" int ds_conv_dgrad_bnbwd_bf16(const ds_conv_shape *s, const float *gz_up, const void *w_hi, const void *w_lo,
                                        const float *g2, const float *z, const float *mean, const float *invstd,
                                        const float *mask_scale, const float *mask_shift, int G, float *gy,
                                        float *partial, void *stream) {
    DS_REQUIRE(s && gz_up && w_hi && w_lo && z && mean && invstd && mask_scale && mask_shift && gy && partial, DS_ERR_NULL);
    DS_REQUIRE(DS_ALIGNED16(gz_up) && DS_ALIGNED16(w_hi) && DS_ALIGNED16(w_lo) && DS_ALIGNED16(z) && DS_ALIGNED16(gy) &&
                   DS_ALIGNED16(mean) && DS_ALIGNED16(invstd) && DS_ALIGNED16(mask_scale) && DS_ALIGNED16(mask_shift) &&
                   (!g2 || DS_ALIGNED16(g2)), DS_ERR_ALIGNMENT);
    PlanB pl;
    int rc = plan_bnbwd(pl, s, G);
    if (rc != DS_OK) return rc;
    pl.k.x = gz_up; pl.k.w_hi = (const __bf16 *)w_hi; pl.k.w_lo = (const __bf16 *)w_lo; pl.k.y = gy;
    pl.k.scale = pl.k.shift = nullptr;
    pl.k.res = g2;
    pl.k.stats = partial;
    pl.k.flags = DS_EPI_STATS | (g2 ? DS_EPI_RESIDUAL : 0);
    pl.k.bn_z = z; pl.k.bn_mean = mean; pl.k.bn_invstd = invstd; pl.k.bn_msc = mask_scale; pl.k.bn_msh = mask_shift;
    ds_bf16_launch_k3x3g(pl, stream);
    return ds_last_launch_error();
}

// The same fusion for the 5x5 stride-2 data gradient, which feeds the activation of a BasicBlock's OUTPUT,
// out = clip(bn2(conv2(y)) + r) (model.py:76-80): that clip's mask cannot be re-derived from z alone, so it is read from
// the stored activation `act` ([B,H,W,Cin], the 5x5 layer's input); no second gradient is added.  Four parity-class
// launches (ds_conv_dgrad_bf16) write interleaved pixels of gy and consecutive blocks of partial rows.
static int plan_s2_class(PlanB &pl, const ds_conv_shape *s, int cls, int G, int &rows_member) {
    const int ph = cls >> 1, pw = cls & 1;
    const int Ho = (s->H - 1) / 2 + 1, Wo = (s->W - 1) / 2 + 1;          // dY grid
    const int Hr = (s->H - ph + 1) / 2, Wc = (s->W - pw + 1) / 2;        // dX rows / columns of this parity
    rows_member = 0;
    if (Hr <= 0 || Wc <= 0) return DS_OK;
    ds_conv_shape t = {s->B, Ho, Wo, s->Cout, s->Cin, 3, 1};
    int rc = plan_bf16(pl, &t, true, 2, ph, pw, s->H, s->W, Hr, Wc);
    if (rc != DS_OK) return rc;
    const long long segs_per_member = (long long)(s->B / G) * pl.k.segs_per_img;
    DS_REQUIRE(segs_per_member % pl.k.NI == 0, DS_ERR_UNSUPPORTED);
    rows_member = (int)(segs_per_member / pl.k.NI);
    pl.k.bn_mtiles = rows_member;
    return DS_OK;
}

extern "C" int ds_conv_dgrad_s2_bnbwd_bf16_rows(const ds_conv_shape *s, int G) {
    DS_REQUIRE(s, DS_ERR_NULL);
    DS_REQUIRE(s->KS == 5 && s->stride == 2 && G > 0 && s->B % G == 0, DS_ERR_UNSUPPORTED);
    int total = 0;
    for (int cls = 0; cls < 4; ++cls) {
        PlanB pl;
        int rows = 0;
        int rc = plan_s2_class(pl, s, cls, G, rows);
        if (rc != DS_OK) return rc;
        total += rows;
    }
    return total;
}

extern "C" int ds_conv_dgrad_s2_bnbwd_bf16(const ds_conv_shape *s, const float *gz_up, const void *w_hi, const void *w_lo,
                                           const float *act, const float *z, const float *mean, const float *invstd,
                                           int G, float *gy, float *partial, void *stream) {
    DS_REQUIRE(s && gz_up && w_hi && w_lo && act && z && mean && invstd && gy && partial, DS_ERR_NULL);
    DS_REQUIRE(s->KS == 5 && s->stride == 2 && G > 0 && s->B % G == 0, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(DS_ALIGNED16(gz_up) && DS_ALIGNED16(w_hi) && DS_ALIGNED16(w_lo) && DS_ALIGNED16(act) && DS_ALIGNED16(z) &&
                   DS_ALIGNED16(gy) && DS_ALIGNED16(mean) && DS_ALIGNED16(invstd), DS_ERR_ALIGNMENT);
    const int total = ds_conv_dgrad_s2_bnbwd_bf16_rows(s, G);
    if (total <= 0) return total < 0 ? total : DS_ERR_UNSUPPORTED;
    const size_t bank = (size_t)9 * s->Cout * s->Cin;
    int row0 = 0;
    for (int cls = 0; cls < 4; ++cls) {
        PlanB pl;
        int rows = 0;
        int rc = plan_s2_class(pl, s, cls, G, rows);
        if (rc != DS_OK) return rc;
        if (rows == 0) continue;
        pl.k.x = gz_up;
        pl.k.w_hi = (const __bf16 *)w_hi + cls * bank;
        pl.k.w_lo = (const __bf16 *)w_lo + cls * bank;
        pl.k.y = gy;
        pl.k.scale = pl.k.shift = nullptr;
        pl.k.res = act;                                  // the mask's source (not added: bn_msc == nullptr)
        pl.k.stats = partial;
        pl.k.flags = DS_EPI_STATS | DS_EPI_RESIDUAL;
        pl.k.bn_z = z; pl.k.bn_mean = mean; pl.k.bn_invstd = invstd; pl.k.bn_msc = pl.k.bn_msh = nullptr;
        pl.k.bn_rows_member = total;
        pl.k.bn_row0 = row0;
        ds_bf16_launch_k3x3g(pl, stream);
        rc = ds_last_launch_error();
        if (rc) return rc;
        row0 += rows;
    }
    return DS_OK;
}
