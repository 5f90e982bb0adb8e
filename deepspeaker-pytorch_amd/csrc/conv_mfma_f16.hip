// conv_mfma_f16.hip -- planner and C ABI of the fp16 implicit-GEMM convolution (kernel: conv_mfma_f16_kernel.h):
// the eval-forward throughput path for reference model.py:69,73,192,197,202 with the fused
// BatchNorm-affine / residual / clipped-ReLU epilogue (model.py:70-80,188-205).  Activations are fp16
// channels-last in HBM, products run on v_mfma_f32_32x32x16_f16 with f32 accumulation.
#include <ds_device.h>
#include <algorithm>
#include "ds_common.h"

#include "conv_mfma_f16_pkernel.h"

namespace {

// OIHW f32 -> [K/16][tap][N][16] fp16 (round to nearest even), the bank of a convolution contracting K input channels
// into N output channels over `taps` taps:
//   mode 0  forward:            K = Cin,  N = Cout,     taps = KS x KS            bank[k][t][n] = w[n][k][t]
//   mode 1  data gradient:      K = Cout, N = Cin,      taps flipped              bank[k][t][n] = w[k][n][T-1-t]
//   mode 2  data gradient of the 5x5 STRIDE-2 layers as ONE 3x3 stride-1 convolution over the output-gradient grid with
//           4 Cin output channels, one block of Cin per parity class (a, b) of the input pixel (2i + a, 2j + b):
//           dX[2i+a, 2j+b, ci] = sum_{r,c,co} dY[i-1+r, j-1+c, co] * w[co][ci][a + 2(2-r)][b + 2(2-c)]   (taps with a
//           kernel index of 5 do not exist: zero).  K = Cout, N = 4 Cin (n = (2a + b) Cin + ci), taps = 3 x 3.
__device__ __forceinline__ void pack_conv_weight_f16_body(const float *w, _Float16 *out, int Cout, int Cin, int KS, int mode,
                                                          int block, int n_blocks) {
    const int TK = mode == 2 ? 3 : KS, T = TK * TK;
    const int N = mode == 0 ? Cout : (mode == 1 ? Cin : 4 * Cin), K = mode == 0 ? Cin : Cout;
    const long long n = (long long)N * K * T;
    for (long long i = (long long)block * 256 + threadIdx.x; i < n; i += (long long)n_blocks * 256) {
        const int kk = (int)(i & 15);
        long long r = i >> 4;
        const int nn = (int)(r % N);
        r /= N;
        const int t = (int)(r % T);
        const int kc = (int)(r / T);
        const int k = kc * 16 + kk;
        float v;
        if (mode == 0) {
            const int kh = t / KS, kw = t - kh * KS;
            v = w[(((size_t)nn * Cin + k) * KS + kh) * KS + kw];
        } else if (mode == 1) {
            const int tt = T - 1 - t;
            const int kh = tt / KS, kw = tt - kh * KS;
            v = w[(((size_t)k * Cin + nn) * KS + kh) * KS + kw];
        } else {
            const int cls = nn / Cin, ci = nn - cls * Cin;
            const int a = cls >> 1, b = cls & 1;
            const int kh = a + 2 * (2 - t / 3), kw = b + 2 * (2 - t % 3);
            v = (kh < 5 && kw < 5) ? w[(((size_t)k * Cin + ci) * 5 + kh) * 5 + kw] : 0.0f;
        }
        out[i] = (_Float16)v;
    }
}

__global__ void __launch_bounds__(256) pack_conv_weight_f16_kernel(const float *w, _Float16 *out, int Cout, int Cin,
                                                                   int KS, int mode) {
    pack_conv_weight_f16_body(w, out, Cout, Cin, KS, mode, (int)blockIdx.x, (int)gridDim.x);
}

// all filters of a weight version in one launch: job j owns the workgroups [first[j], first[j + 1])
struct PackBatchH {
    const float *w[DS_PACK_BATCH_MAX];
    _Float16 *out[DS_PACK_BATCH_MAX];
    int Cout[DS_PACK_BATCH_MAX], Cin[DS_PACK_BATCH_MAX], KS[DS_PACK_BATCH_MAX], mode[DS_PACK_BATCH_MAX];
    int first[DS_PACK_BATCH_MAX + 1];
    int n;
};

__global__ void __launch_bounds__(256) pack_conv_weight_f16_batch_kernel(const PackBatchH J) {
    int j = 0;
    while (j + 1 < J.n && (int)blockIdx.x >= J.first[j + 1]) ++j;
    pack_conv_weight_f16_body(J.w[j], J.out[j], J.Cout[j], J.Cin[j], J.KS[j], J.mode[j], (int)blockIdx.x - J.first[j],
                              J.first[j + 1] - J.first[j]);
}

__global__ void __launch_bounds__(256) cast_f32_to_f16_kernel(const float *x, _Float16 *y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = (_Float16)x[i];
}

__global__ void __launch_bounds__(256) cast_f16_to_f32_kernel(const _Float16 *x, float *y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = (float)x[i];
}

// split-K second pass: y = epilogue(sum_s partial[s]) -- fixed summation order; 8 channels per thread
__global__ void __launch_bounds__(256) conv_f16_splitk_reduce_kernel(const float *partial, int S, unsigned stride,
                                                                     const float *scale, const float *shift,
                                                                     const _Float16 *res, void *y, long long n8, int Cout,
                                                                     int flags) {
    const float clip_lo = (flags & DS_EPI_CLIP) ? 0.0f : -__builtin_inff();
    const float clip_hi = (flags & DS_EPI_CLIP) ? 20.0f : __builtin_inff();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const long long e = i * 8;
        const int c = (int)(e % Cout);
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < S; ++s) {
            a0 += *(const f32x4 *)(partial + (size_t)s * stride + e);
            a1 += *(const f32x4 *)(partial + (size_t)s * stride + e + 4);
        }
        f16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
        if (flags & DS_EPI_RESIDUAL) r = *(const f16x8 *)(res + e);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float t = j < 4 ? a0[j] : a1[j - 4];
            if (flags & DS_EPI_AFFINE) t = t * scale[c + j] + shift[c + j];
            t += (float)r[j];
            o[j] = fminf(fmaxf(t, clip_lo), clip_hi);
        }
        if (flags & DS_EPI_OUT_F32) {
            *(f32x4 *)((float *)y + e) = f32x4{o[0], o[1], o[2], o[3]};
            *(f32x4 *)((float *)y + e + 4) = f32x4{o[4], o[5], o[6], o[7]};
        } else {
            f16x8 h;
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = (_Float16)o[j];
            *(f16x8 *)((_Float16 *)y + e) = h;
        }
    }
}

// LDS cycles (1 = conflict-free) of one ds_read_b128 fragment read for candidate row / segment strides: the two
// 16-lane service groups of lanes 0..31 each hold 16 consecutive pixels of the M tile (`lpix` in the kernel).
// Bank slot of a record = (byte offset / 16) mod 16; lanes reading the same record do not conflict.
static double frag_read_cost(int MT, int NI, int RT, int Wc, int IS, int row_bytes, int seg_bytes, int PSH) {
    const int pix_per_seg = RT * Wc;
    double total = 0.0;
    int n = 0;
    for (int m0 = 0; m0 + 32 <= MT; m0 += 32) {
        for (int g = 0; g < 2; ++g) {
            int cnt[16] = {0}, offs[16];
            int worst = 0;
            for (int j = 0; j < 16; ++j) {
                int m = m0 + 16 * g + j;
                if (m >= NI * pix_per_seg) m &= ~15;    // past the last segment: the group's first pixel (kernel: a_off)
                const int seg = m / pix_per_seg, rem = m % pix_per_seg;
                const int r = rem / Wc, c = rem % Wc;
                const int off = offs[j] = (seg < NI) ? seg * seg_bytes + (IS * r) * row_bytes + c * PSH : 0;
                bool seen = false;                      // lanes reading the SAME address are served together (the
                for (int i = 0; i < j; ++i) seen |= offs[i] == off;     // pixels past the tile's last segment: record 0)
                if (seen) continue;
                const int slot = (off / 16) & 15;
                if (++cnt[slot] > worst) worst = cnt[slot];
            }
            total += worst;
            ++n;
        }
    }
    return n ? total / n : 1.0;
}

// A tile row is `pitch` records plus `row_pad` 16-byte units, a segment rows_in rows plus `seg_pad` units: when the 16
// pixels of a service group span several rows (maps narrower than 16 columns) or images (whole-image segments), whole
// records of padding cannot always separate their bank slots (a 20x8 map wants its row stride = 8 units mod 16, a 10x4
// map 4 or 12: frag_read_cost), a few 16-byte units can.  The cheapest conflict-free layout within the LDS the plan was
// sized for (rows of up to cols_in + 4 records, 256 bytes of slack per segment).
// Measured (tools/ab_layout.py, same process, alternating): the padded, model-conflict-free layout changes no layer by more
// than its run-to-run spread (20x8: 126.9 vs 127.2 us; 10x4: 163.9 vs 161.6) -- the fragment reads of those layers are not
// what their MFMA streams wait for.  Off by default; kept as a tuning hook.
static bool g_layout_padding = false;
static int g_forced_cfg = -1;               // tuning hook (tools/f16_cfg_ab.py): plan with this tile configuration only
static void choose_strides(ConvKH &k, int MT, int PSH) {
    const int row_cap = (k.cols_in + 4) * PSH;
    const int pad_units = g_layout_padding ? 16 : 1;
    double best_cost = 1e30;
    long long best_bytes = 0;
    for (int pt = k.cols_in; pt <= k.cols_in + 4; ++pt)
        for (int rp = 0; rp < pad_units; ++rp) {
            const int row_bytes = pt * PSH + 16 * rp;
            if (row_bytes > row_cap) break;
            for (int sp = 0; sp < (k.NI > 1 ? pad_units : 1); ++sp) {
                const int seg_bytes = k.rows_in * row_bytes + 16 * sp;
                const double c = frag_read_cost(MT, k.NI, k.RT, k.Wo, k.IS, row_bytes, seg_bytes, PSH);
                const long long bytes = (long long)k.NI * seg_bytes;
                if (c < best_cost - 1e-9 || (c < best_cost + 1e-9 && bytes < best_bytes)) {
                    best_cost = c;
                    best_bytes = bytes;
                    k.pitch = pt;
                    k.row_bytes = row_bytes;
                    k.seg_bytes = seg_bytes;
                }
            }
        }
    k.seg_pix = k.rows_in * k.pitch;
}

struct TileCfgH { int MT, NTILE, WM, NTHR; };
constexpr int kNumCfgH = 7;
constexpr TileCfgH kCfgH[kNumCfgH] = {
    {160, 128, 1, 128},     // <KS,5,2,1,2>: two waves, 160x64 register tile each
    {160, 256, 1, 256},     // <KS,5,2,1,4>
    {320, 128, 2, 256},     // <KS,5,2,2,2>
    {320, 64, 2, 128},      // <KS,5,2,2,1>: the 2-wave shape for 64-channel layers
    {128, 128, 1, 128},     // <KS,4,2,1,2>: 128x64 register tiles where 160-row tiles quantise badly
    {128, 256, 1, 256},     // <KS,4,2,1,4>
    {640, 64, 4, 256},      // <KS,5,2,4,1>: four waves on a 64-channel layer
};
constexpr size_t kLdsTotal = 160 * 1024;     // per CU

static size_t epi_bytes(const TileCfgH &cf) {
    const int waves = cf.NTHR / 64, nsub = cf.NTILE / (waves / cf.WM) / 32;
    return (size_t)2 * waves * 32 * (nsub * 32 + 4) * 4;
}

// One wave per SIMD is the design point (the register tile takes most of the 512 VGPRs): a CU holds
// 256 / NTHR workgroups, each with an equal share of the LDS.
static int plan_f16(PlanH &pl, const ds_conv_shape *s, bool allow_db = true, bool force_c16 = false) {
    DS_REQUIRE(s != nullptr, DS_ERR_NULL);
    DS_REQUIRE(s->B > 0 && s->H > 0 && s->W > 0 && s->Cin > 0 && s->Cout > 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(s->KS == 3 || s->KS == 5, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->stride == 1 || s->stride == 2, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->Cin % 32 == 0 && s->Cout % 64 == 0, DS_ERR_BAD_SHAPE);
    const int pad = s->KS / 2;
    const int Ho = (s->H + 2 * pad - s->KS) / s->stride + 1;
    const int Wo = (s->W + 2 * pad - s->KS) / s->stride + 1;
    DS_REQUIRE(Ho > 0 && Wo > 0 && Wo <= 128, DS_ERR_BAD_SHAPE);
    DS_REQUIRE((long long)s->B * s->H * s->W * s->Cin < (1ll << 31), DS_ERR_BAD_SHAPE);
    DS_REQUIRE((long long)s->B * Ho < (1ll << 24), DS_ERR_BAD_SHAPE);                  // reciprocal index arithmetic
    DS_REQUIRE((long long)s->B * Ho * Wo * s->Cout < (1ll << 30), DS_ERR_BAD_SHAPE);   // 32-bit byte offsets
    const int IS = s->stride;
    double best = -1.0;
    int bc = -1, brt = 0, bni = 0, bdb = 0, bck = 32;
    for (int c = 0; c < kNumCfgH; ++c) {
        const TileCfgH &cf = kCfgH[c];
        if (s->Cout % cf.NTILE) continue;
        if (g_forced_cfg >= 0 && c != g_forced_cfg) continue;
        const int wg_per_cu = 256 / cf.NTHR;
        const size_t lds_cap = kLdsTotal / wg_per_cu - 64;
        // (two tiles of 32 channels) > (two tiles of 16 channels: 5x5 stride-2 layers, whose input tile is 4x the
        // output tile) > (one tile of 32 channels: two barriers and exposed LDS writes per chunk)
        for (int mode = allow_db ? 0 : 2; mode < 3; ++mode) {
            const int db = mode < 2, ck = mode == 1 ? 16 : 32;
            if (mode == 1 && s->KS != 5) continue;
            if (force_c16 && s->KS == 5 && mode != 1) continue;
            const int PSH = ds_f16_record_bytes(ck);
            const long long item_cap = (db ? 16 : 32) * cf.NTHR;
            for (int rt = 1; rt <= Ho; ++rt) {
                if ((long long)rt * Wo > cf.MT) break;
                const int segs_per_img = ds_ceil_div(Ho, rt);
                const long long n_segs = (long long)s->B * segs_per_img;
                int ni = cf.MT / (rt * Wo);
                if (ni > n_segs) ni = (int)n_segs;
                const int rows_in = IS * (rt - 1) + s->KS, cols_in = IS * (Wo - 1) + s->KS;
                auto lds_of = [&](int n) {
                    const size_t tp = (size_t)n * rows_in * (cols_in + 4);
                    return std::max((tp * PSH + (size_t)n * 256) * (db ? 2 : 1), epi_bytes(cf)) + (size_t)cf.MT * 4 + (size_t)n * 8;
                };
                auto items_of = [&](int n) { return (long long)n * std::min(rows_in, s->H) * s->W * (ck / 8); };
                while (ni > 1 && (lds_of(ni) > lds_cap || items_of(ni) > item_cap)) --ni;
                if (lds_of(ni) > lds_cap || items_of(ni) > item_cap) continue;
                const long long n_mt = ds_ceil_div_ll(n_segs, ni);
                double eff = (double)s->B * Ho * Wo / ((double)n_mt * cf.MT);
                const long long blocks = n_mt * (s->Cout / cf.NTILE), slots = 256ll * wg_per_cu;
                if (blocks <= slots) eff *= (double)blocks / (double)(ds_ceil_div_ll(blocks, 256) * 256);
                else eff *= (double)blocks / (double)(ds_ceil_div_ll(blocks, slots) * slots);
                if (mode == 1) eff *= 0.97;
                if (mode == 2) eff *= (s->KS == 3 ? 0.90 : 0.85);
                // ties: the four-wave 160 x 256 tile ahead of the two-wave 160 x 128 one (round 5, tools/f16_cfg_ab.py, every
                // configuration forced in turn at the bench size: 135 against 142 us on the 256-channel 3x3 layers; the
                // other layers' choices were already the fastest)
                // (same sweep: on the 128-channel 3x3 layers of the training step the four-wave 320 x 128 tile takes 146 us per
                // layer, the two-wave 320 x 64 one 157: the wider n tile first, whatever the M tile)
                // 128-pixel tiles: the two-wave plan first for a 3x3 (widen_persistent turns it into the NSUB = 4 tile where
                // Cout allows: 137.4 against 138.5 us), the four-wave one for a 5x5 (190 against 270 us)
                static const int pref3[kNumCfgH] = {3, 6, 5, 1, 4, 2, 0}, pref5[kNumCfgH] = {3, 6, 5, 1, 2, 4, 0};
                eff += 1e-9 * rt + 1e-6 * (s->KS == 3 ? pref3 : pref5)[c];
                if (eff > best) { best = eff; bc = c; brt = rt; bni = ni; bdb = db; bck = ck; }
            }
        }
    }
    if (bc < 0) return DS_ERR_UNSUPPORTED;
    const TileCfgH &cf = kCfgH[bc];
    ConvKH &k = pl.k;
    k.H = s->H; k.W = s->W; k.Cin = s->Cin;
    k.Ho = Ho; k.Wo = Wo; k.Cout = s->Cout;
    k.IS = IS; k.dh_min = -pad; k.dw_min = -pad;
    k.RT = brt; k.NI = bni;
    k.segs_per_img = ds_ceil_div(Ho, brt);
    k.n_segs = s->B * k.segs_per_img;
    k.rows_in = IS * (brt - 1) + s->KS;
    k.cols_in = IS * (Wo - 1) + s->KS;
    k.half = (k.cols_in + 1) / 2;
    choose_strides(k, cf.MT, ds_f16_record_bytes(bck));
    k.n_ntiles = s->Cout / cf.NTILE;
    pl.cfg = bc;
    pl.db = bdb;
    pl.ck = bck;
    pl.n_mtiles = ds_ceil_div(k.n_segs, bni);
    pl.grid = pl.n_mtiles * k.n_ntiles;
    const size_t tile_bytes = (size_t)k.NI * k.seg_bytes;
    pl.lds_bytes = std::max(tile_bytes * (bdb ? 2 : 1), epi_bytes(cf)) + (size_t)cf.MT * 4 + (size_t)k.NI * 8 + 16;
    pl.nit = ds_ceil_div(k.NI * std::min(k.rows_in, s->H) * s->W * (bck / 8), cf.NTHR);
    return DS_OK;
}

// The persistent kernel (conv_mfma_f16_pkernel.h) takes the common geometry: double-buffered tile, an M tile that is
// one block of rows of one image or a run of whole images, staging items within the register budget (no split-K: the
// caller checks).  On success the plan's grid / nit / lin are those of the persistent launch.
static bool plan_persistent(PlanH &pl, const ds_conv_shape *s) {
    const ConvKH &k = pl.k;
    if (!pl.db || !(k.NI == 1 || k.segs_per_img == 1)) return false;
    const TileCfgH &cf = kCfgH[pl.cfg];
    int e_rows = k.rows_in;
    if (k.NI > 1) {
        const int lo = k.dh_min < 0 ? -k.dh_min : 0, hi = std::min(k.H - k.dh_min, k.rows_in);
        e_rows = hi > lo ? hi - lo : 1;
    }
    const int nit_p = ds_ceil_div(k.NI * e_rows * s->W * (pl.ck / 8), cf.NTHR);
    // more than 8 items per thread: only as a row block whose width divides the pixels a pass of the workgroup covers
    // (the kernel then derives every item's offsets from the first one's: LIN in conv_mfma_f16_pkernel.h)
    const int pix_per_pass = cf.NTHR / (pl.ck / 8);
    const int lin = k.NI == 1 && pix_per_pass % s->W == 0;
    if (nit_p > 16 || (nit_p > 8 && !lin)) return false;
    pl.lin = lin;
    pl.nit = nit_p;
    const int resident = (256 / cf.NTHR) * ds_cu_count();        // one wave per SIMD
    if (pl.grid > resident) pl.grid = resident;
    return true;
}

// cfg 7 exists in the persistent kernel only: the 128x128 two-wave plan (cfg 4) of a 3x3 layer with 32-channel chunks
// and Cout % 256 == 0 widened to 128 x 256 -- each wave a 128 x 128 register tile (NSUB = 4), so that every pixel
// fragment read from LDS feeds four MFMAs instead of two (tools/mfma_lds_ratio.hip: the MFMA + LDS ceiling of that ratio
// is 1.58 - 1.65 PFLOP/s against 1.35 - 1.49).  Same M tiling, same staging, same per-pixel accumulation order: results
// are bit-identical to cfg 4.  `k.tiles` / `pl.grid` are recomputed for the halved number of N tiles.
constexpr int kCfgWide = 7;
static bool widen_persistent(PlanH &pl, const ds_conv_shape *s) {
    if (pl.cfg != 4 || s->KS != 3 || pl.ck != 32 || s->Cout % 256 != 0 || !pl.db) return false;
    const size_t tile_bytes = (size_t)pl.k.NI * pl.k.seg_bytes;
    const size_t epi = (size_t)2 * 2 * 32 * (2 * 32 + 4) * 4;           // two waves, two 32 x 68-float buffers each
    const size_t lds = std::max(tile_bytes * 2, epi) + (size_t)128 * 4 + (size_t)pl.k.NI * 8 + 16;
    if (lds > kLdsTotal / 2 - 64) return false;
    pl.cfg = kCfgWide;
    pl.k.n_ntiles = s->Cout / 256;
    pl.lds_bytes = lds;
    pl.grid = pl.n_mtiles * pl.k.n_ntiles;
    const int resident = 2 * ds_cu_count();
    pl.k.tiles = pl.grid;
    if (pl.grid > resident) pl.grid = resident;
    return true;
}

}  // namespace

static int pack_f16(const float *w_oihw, void *w_f16, int Cout, int Cin, int KS, int mode, void *stream) {
    DS_REQUIRE(w_oihw && w_f16, DS_ERR_NULL);
    DS_REQUIRE(Cout > 0 && Cin > 0 && (KS == 3 || KS == 5), DS_ERR_BAD_SHAPE);
    DS_REQUIRE(((mode == 0 ? Cin : Cout) % 16) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(mode != 2 || KS == 5, DS_ERR_UNSUPPORTED);
    const long long n = (long long)Cout * Cin * (mode == 2 ? 36 : KS * KS);
    long long g = (n + 255) / 256;
    DS_LAUNCH(pack_conv_weight_f16_kernel, (int)(g > 4096 ? 4096 : g), 256, 0, stream, w_oihw, (_Float16 *)w_f16, Cout,
              Cin, KS, mode);
    return ds_last_launch_error();
}

extern "C" int ds_pack_conv_weights_f16_batch(const ds_pack_job *jobs, int n_jobs, void *stream) {
    DS_REQUIRE(jobs != nullptr, DS_ERR_NULL);
    DS_REQUIRE(n_jobs > 0 && n_jobs <= DS_PACK_BATCH_MAX, DS_ERR_BAD_SHAPE);
    PackBatchH J;
    J.n = n_jobs;
    int blocks = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const ds_pack_job &b = jobs[j];
        DS_REQUIRE(b.w_oihw && b.out, DS_ERR_NULL);
        DS_REQUIRE(b.mode >= 0 && b.mode <= 2 && b.Cout > 0 && b.Cin > 0, DS_ERR_BAD_SHAPE);
        DS_REQUIRE(b.KS == 3 || b.KS == 5, DS_ERR_UNSUPPORTED);
        DS_REQUIRE((b.mode == 0 ? b.Cin : b.Cout) % 16 == 0, DS_ERR_BAD_SHAPE);
        DS_REQUIRE(b.mode != 2 || b.KS == 5, DS_ERR_UNSUPPORTED);
        J.w[j] = b.w_oihw; J.out[j] = (_Float16 *)b.out;
        J.Cout[j] = b.Cout; J.Cin[j] = b.Cin; J.KS[j] = b.KS; J.mode[j] = b.mode;
        const long long n = (long long)b.Cout * b.Cin * (b.mode == 2 ? 36 : b.KS * b.KS);
        const long long g = (n + 255) / 256;
        J.first[j] = blocks;
        blocks += (int)(g > 512 ? 512 : g);
    }
    J.first[n_jobs] = blocks;
    DS_LAUNCH(pack_conv_weight_f16_batch_kernel, blocks, 256, 0, stream, J);
    return ds_last_launch_error();
}

extern "C" int ds_pack_conv_weight_f16(const float *w_oihw, void *w_f16, int Cout, int Cin, int KS, void *stream) {
    return pack_f16(w_oihw, w_f16, Cout, Cin, KS, 0, stream);
}

// The data-gradient banks of the fp16 training step (train_f16.hip): ds_conv_fwd_f16 over dL/d(conv output) with these
// filters IS the data gradient.  stride 1: [Cout/16][tap][Cin][16], taps flipped (Cout * Cin * KS * KS halfs; the
// convolution then has Cin' = Cout, Cout' = Cin).  stride 2 (KS = 5 only): [Cout/16][9][4 Cin][16] (36 * Cout * Cin
// halfs) -- a 3x3 stride-1 convolution with Cin' = Cout, Cout' = 4 Cin whose output [B][Ho][Wo][2][2][Cin] holds the four
// parity classes of dX (ds_bn_bwd_group_f16 reads that layout directly).
extern "C" int ds_pack_conv_weight_dgrad_f16(const float *w_oihw, void *w_f16, int Cout, int Cin, int KS, int stride,
                                             void *stream) {
    DS_REQUIRE(stride == 1 || stride == 2, DS_ERR_UNSUPPORTED);
    return pack_f16(w_oihw, w_f16, Cout, Cin, KS, stride == 2 ? 2 : 1, stream);
}

extern "C" int ds_cast_f32_to_f16(const float *x, void *y_f16, long long n, void *stream) {
    DS_REQUIRE(x && y_f16, DS_ERR_NULL);
    DS_REQUIRE(n > 0, DS_ERR_BAD_SHAPE);
    long long g = (n + 255) / 256;
    DS_LAUNCH(cast_f32_to_f16_kernel, (int)(g > 8192 ? 8192 : g), 256, 0, stream, x, (_Float16 *)y_f16, n);
    return ds_last_launch_error();
}

extern "C" int ds_cast_f16_to_f32(const void *x_f16, float *y, long long n, void *stream) {
    DS_REQUIRE(x_f16 && y, DS_ERR_NULL);
    DS_REQUIRE(n > 0, DS_ERR_BAD_SHAPE);
    long long g = (n + 255) / 256;
    DS_LAUNCH(cast_f16_to_f32_kernel, (int)(g > 8192 ? 8192 : g), 256, 0, stream, (const _Float16 *)x_f16, y, n);
    return ds_last_launch_error();
}

extern "C" int ds_conv_f16_plan_describe_hinted(const ds_conv_shape *s, int flags, int *out8);
extern "C" int ds_conv_f16_plan_describe(const ds_conv_shape *s, int *out8) {
    return ds_conv_f16_plan_describe_hinted(s, 0, out8);
}

// the plan ds_conv_fwd_f16 would use under the DS_CONV_HINT_* / DS_CONV_IN_PLANES16 bits of `flags`
extern "C" int ds_conv_f16_plan_describe_hinted(const ds_conv_shape *s, int flags, int *out8) {
    DS_REQUIRE(out8 != nullptr, DS_ERR_NULL);
    PlanH pl;
    int rc = plan_f16(pl, s, !(flags & DS_CONV_HINT_SINGLE_BUFFER),
                      (flags & (DS_CONV_IN_PLANES16 | DS_CONV_HINT_CHUNK16)) != 0);
    if (rc != DS_OK) return rc;
    if (s->KS == 5 && pl.ck == 32 && !(flags & (DS_CONV_HINT_SINGLE_BUFFER | DS_CONV_HINT_NO_PERSIST))) {   // as ds_conv_fwd_f16
        PlanH p32 = pl, p16;
        if (plan_persistent(p32, s) && p32.nit > 8 && plan_f16(p16, s, true, true) == DS_OK && p16.ck == 16) {
            PlanH q = p16;
            if (plan_persistent(q, s) && q.nit <= 8) pl = p16;
        }
    }
    const TileCfgH &cf = kCfgH[pl.cfg];
    out8[0] = cf.MT; out8[1] = cf.NTILE; out8[2] = pl.k.RT; out8[3] = pl.k.NI;
    const int tiles = pl.grid;                                   // out8[4]: tiles (= workgroups of the one-tile kernel)
    const bool pers = !(flags & DS_CONV_HINT_NO_PERSIST) && plan_persistent(pl, s);
    if (pers && !(flags & DS_CONV_HINT_NO_WIDE) && widen_persistent(pl, s)) out8[1] = 256;   // cfg 7: 128 x 256, NSUB = 4
    out8[4] = tiles; out8[5] = (int)pl.lds_bytes; out8[6] = cf.NTHR;
    // out8[7]: 10000 if the persistent kernel takes this plan (large launches; small ones may still be split-K)
    //          + 1000 if double-buffered + 100 for 16-channel chunks + staging items per thread
    out8[7] = (pers ? 10000 : 0) + pl.db * 1000 + (pl.ck == 16 ? 100 : 0) + pl.nit;
    return DS_OK;
}

// tuning hook (tools/ab_layout.py): 0 = tile rows / segments of whole records only; 1 (default) = padded strides
extern "C" void ds_conv_f16_set_layout_padding(int on) { g_layout_padding = on != 0; }
// tuning hook (tools/f16_cfg_ab.py): cfg in [0, 7) = plan every fp16 convolution with that tile configuration (layers it
// does not fit return DS_ERR_UNSUPPORTED); anything else = the planner's own choice.  Results do not depend on it.
extern "C" void ds_conv_f16_set_forced_cfg(int cfg) { g_forced_cfg = (cfg >= 0 && cfg < kNumCfgH) ? cfg : -1; }

// the LDS layout of the pixel tile in that plan: out4 = { records per tile row, bytes per tile row, bytes per segment,
// 1000 x LDS cycles of a fragment read (1000 = conflict-free) }
extern "C" int ds_conv_f16_plan_lds_layout(const ds_conv_shape *s, int flags, int *out4) {
    DS_REQUIRE(out4 != nullptr, DS_ERR_NULL);
    PlanH pl;
    int rc = plan_f16(pl, s, !(flags & DS_CONV_HINT_SINGLE_BUFFER),
                      (flags & (DS_CONV_IN_PLANES16 | DS_CONV_HINT_CHUNK16)) != 0);
    if (rc != DS_OK) return rc;
    if (s->KS == 5 && pl.ck == 32 && !(flags & (DS_CONV_HINT_SINGLE_BUFFER | DS_CONV_HINT_NO_PERSIST))) {   // as ds_conv_fwd_f16
        PlanH p32 = pl, p16;
        if (plan_persistent(p32, s) && p32.nit > 8 && plan_f16(p16, s, true, true) == DS_OK && p16.ck == 16) {
            PlanH q = p16;
            if (plan_persistent(q, s) && q.nit <= 8) pl = p16;
        }
    }
    const ConvKH &k = pl.k;
    out4[0] = k.pitch; out4[1] = k.row_bytes; out4[2] = k.seg_bytes;
    out4[3] = (int)(1000.0 * frag_read_cost(kCfgH[pl.cfg].MT, k.NI, k.RT, k.Wo, k.IS, k.row_bytes, k.seg_bytes,
                                            ds_f16_record_bytes(pl.ck)) + 0.5);
    return DS_OK;
}

#ifdef DS_F16_PROBE
static long long *g_f16_probe = nullptr;
extern "C" void ds_f16_set_probe(long long *buf) { g_f16_probe = buf; }
#endif

// How many ways a small launch splits its contraction: enough workgroups for two per CU, at least one chunk each
static int splitk_ways(const PlanH &pl, const ds_conv_shape *s) {
    const int n_chunks = s->Cin / pl.ck;
    int ways = 1;
    if (pl.grid > 48) return 1;                  // (measured: from ~64 workgroups on, the partial-sum traffic costs more than it saves)
    while (ways * 2 <= 8 && ways * 2 <= n_chunks && (long long)pl.grid * ways * 2 <= 256) ways *= 2;
    return ways;
}

extern "C" long long ds_conv_f16_splitk_workspace_bytes(const ds_conv_shape *s) {
    PlanH pl;
    int rc = plan_f16(pl, s);
    if (rc != DS_OK) return rc;
    const int ways = splitk_ways(pl, s);
    return ways > 1 ? (long long)ways * s->B * pl.k.Ho * pl.k.Wo * s->Cout * 4 : 0;
}

static int conv_fwd_f16(const ds_conv_shape *s, const void *x_f16, const void *w_f16, const float *scale,
                        const float *shift, const void *residual_f16, void *y, int flags, void *workspace,
                        long long ws_bytes, void *stream) {
    DS_REQUIRE(x_f16 && w_f16 && y, DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_AFFINE) || (scale && shift), DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_RESIDUAL) || residual_f16, DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_STATS), DS_ERR_UNSUPPORTED);            // eval path only
    DS_REQUIRE(DS_ALIGNED16(x_f16) && DS_ALIGNED16(w_f16) && DS_ALIGNED16(y) && DS_ALIGNED16(residual_f16) &&
                   DS_ALIGNED16(scale) && DS_ALIGNED16(shift) && DS_ALIGNED16(workspace), DS_ERR_ALIGNMENT);
    PlanH pl;
    const bool in_planes = (flags & DS_CONV_IN_PLANES16) != 0, out_planes = (flags & DS_EPI_OUT_PLANES16) != 0;
    DS_REQUIRE(!in_planes || s->KS == 5, DS_ERR_UNSUPPORTED);          // plane-major input needs 16-channel chunks
    DS_REQUIRE(!out_planes || !(flags & DS_EPI_OUT_F32), DS_ERR_UNSUPPORTED);
    int rc = plan_f16(pl, s, !(flags & DS_CONV_HINT_SINGLE_BUFFER), in_planes || (flags & DS_CONV_HINT_CHUNK16) != 0);
    if (rc != DS_OK) return rc;
    if (s->KS == 5 && pl.ck == 32 && !(flags & (DS_CONV_HINT_SINGLE_BUFFER | DS_CONV_HINT_NO_PERSIST))) {
        // A 5x5 layer whose 32-channel chunks need more than 8 staging items per thread runs the persistent kernel with
        // 16 items in flight -- past what the register file holds next to the 160 accumulators (a few are spilled and
        // reloaded between the MFMAs).  The same layer in 16-channel chunks needs half the items: measured 187 vs
        // 200 us on the 128 -> 256 layer of the bench (tools/f16_layer_ab.py).  Same arithmetic, same results.
        PlanH p32 = pl, p16;
        if (plan_persistent(p32, s) && p32.nit > 8 && plan_f16(p16, s, true, true) == DS_OK && p16.ck == 16) {
            PlanH q = p16;
            if (plan_persistent(q, s) && q.nit <= 8) pl = p16;
        }
    }
    DS_REQUIRE(!in_planes || pl.ck == 16, DS_ERR_UNSUPPORTED);
    ConvKH &k = pl.k;
    k.x_pix_stride = in_planes ? 16 : s->Cin;
    k.x_chunk_stride = in_planes ? s->B * s->H * s->W * 16 : pl.ck;
    k.y_plane_stride = out_planes ? (unsigned)((long long)s->B * pl.k.Ho * pl.k.Wo * 16) : 0u;
    k.x = (const _Float16 *)x_f16; k.w = (const _Float16 *)w_f16; k.y = y;
    k.scale = scale; k.shift = shift; k.res = (const _Float16 *)residual_f16;
    k.flags = flags;
    const long long n_out = (long long)s->B * k.Ho * k.Wo * s->Cout;
    k.y_bytes = (unsigned)(n_out * ((flags & DS_EPI_OUT_F32) ? 4 : 2));
    k.res_bytes = (unsigned)(n_out * 2);
    const int n_chunks = s->Cin / pl.ck;
    int ways = (workspace && !out_planes) ? splitk_ways(pl, s) : 1;
    if (ways > 1 && ws_bytes < (long long)ways * n_out * 4) ways = 1;
    k.tiles = pl.grid;
    k.chunks_per_split = ds_ceil_div(n_chunks, ways);
    k.n_splits = ds_ceil_div(n_chunks, k.chunks_per_split);
    k.partial = (float *)workspace;
    k.partial_elems = (unsigned)n_out;
    pl.grid *= k.n_splits;
#ifdef DS_F16_PROBE
    k.probe = g_f16_probe;
#endif
    const bool persistent = k.n_splits == 1 && !(flags & DS_CONV_HINT_NO_PERSIST) && plan_persistent(pl, s);
    if (persistent) {
        if (!(flags & DS_CONV_HINT_NO_WIDE)) widen_persistent(pl, s);
        k.sched = ds_sched_slot(stream);
        DS_REQUIRE(k.sched != nullptr, DS_ERR_NO_WORKSPACE);     // ds_sched_set_workspace is due
        // one tile queue per XCD where the tiles of a queue (t = 8 j + q) then all belong to one n tile
        k.sched_queues = (pl.grid % 8 == 0 && 8 % k.n_ntiles == 0 && !(flags & DS_CONV_HINT_ONE_QUEUE)) ? 8 : 1;
        k.sched_lds = (int)pl.lds_bytes - 16;       // the plan's LDS size ends with tables the persistent kernel does not use
        if (s->KS == 3) ds_f16_launch_pk3(pl, stream);
        else if (pl.ck == 16) ds_f16_launch_pk5c16(pl, stream);
        else ds_f16_launch_pk5(pl, stream);
    } else if (s->KS == 3) { if (pl.db) ds_f16_launch_k3db(pl, stream); else ds_f16_launch_k3sb(pl, stream); }
    else if (pl.ck == 16) ds_f16_launch_k5c16(pl, stream);
    else            { if (pl.db) ds_f16_launch_k5db(pl, stream); else ds_f16_launch_k5sb(pl, stream); }
    rc = ds_last_launch_error();
    if (rc || k.n_splits == 1) return rc;
    const long long n8 = n_out / 8;
    long long g = (n8 + 255) / 256;
    DS_LAUNCH(conv_f16_splitk_reduce_kernel, (int)(g > 4096 ? 4096 : g), 256, 0, stream, (const float *)workspace,
              k.n_splits, k.partial_elems, scale, shift, (const _Float16 *)residual_f16, y, n8, s->Cout, flags);
    return ds_last_launch_error();
}

extern "C" int ds_conv_fwd_f16(const ds_conv_shape *s, const void *x_f16, const void *w_f16, const float *scale,
                               const float *shift, const void *residual_f16, void *y, int flags, void *stream) {
    return conv_fwd_f16(s, x_f16, w_f16, scale, shift, residual_f16, y, flags, nullptr, 0, stream);
}

// The same convolution for SMALL launches (serving latency): when the tile grid cannot fill the GPU, the contraction
// is split over up to 8 workgroups per tile (raw f32 partial sums in `workspace`, ds_conv_f16_splitk_workspace_bytes)
// and a second kernel folds them in fixed order and applies the epilogue.  Large launches take the one-pass path.
// Results differ from the one-pass path by the f32 summation order only.
extern "C" int ds_conv_fwd_f16_splitk(const ds_conv_shape *s, const void *x_f16, const void *w_f16, const float *scale,
                                      const float *shift, const void *residual_f16, void *y, int flags, void *workspace,
                                      long long ws_bytes, void *stream) {
    return conv_fwd_f16(s, x_f16, w_f16, scale, shift, residual_f16, y, flags, workspace, ws_bytes, stream);
}
