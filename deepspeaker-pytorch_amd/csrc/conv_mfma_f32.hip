// conv_mfma_f32.hip -- implicit-GEMM convolution on the gfx950 f32 matrix cores.
//
// Replaces the cuDNN convolutions the reference launches through nn.Conv2d
// (reference model.py:47-50 3x3 s1 p1; :98,102,106 5x5 s2 p2; and, as a 1x1 case,
// the fc GEMM of model.py:209), fused with the BatchNorm affine, the residual add
// and the clipped ReLU that follow them (model.py:70-80, 188-205).
//
// GEMM view (channels-last activations):  M = output pixels, N = Cout,
// K = taps x Cin.  A workgroup owns an M-tile made of NI "segments" (RT output rows
// x full width of one image) and an N-tile of output channels.  Per 8-channel chunk
// of Cin the input halo tile of every segment is staged ONCE into LDS (zero-filled
// outside the image) and reused by all KS*KS taps; a tap is just a constant LDS
// offset.  Weights stream from L2 straight into registers in the exact fragment
// order (one fully coalesced 1 KiB load per 32x8 slice).  The arithmetic is
// v_mfma_f32_32x32x2_f32 -- exact f32, the parity path.
//
// One kernel serves forward 3x3/5x5/1x1 (and, through the tap table and the
// input/output strides, the data-gradient convolutions).
#include <ds_device.h>
#include "ds_common.h"

namespace {

constexpr int CK = DS_CONV_CK;      // input channels per staged chunk
constexpr int LDS_PS = 12;          // floats per staged pixel: 8 channels + 4 pad.  48 B keeps
                                    // ds_read_b128 aligned and walks all 64 banks over 16 pixels
constexpr int MAX_STAGE_IT = 8;     // upper bound of per-thread float4 staging slots (NIT)
constexpr int MAX_TAPS = 25;

struct ConvK {
    const float *x, *w;
    float *y;
    const float *scale, *shift, *res;
    float *stats;
    int H, W, Cin;          // input tensor (per image)
    int Hr, Wc;             // virtual output grid per image (rows, cols)
    int Ho, Wo, Cout;       // output tensor (per image)
    int IS, OS, OH0, OW0;   // input stride; output stride / origin (data-gradient classes)
    int NT;                 // taps
    int dh_min, dw_min;     // input-tile origin relative to (IS*r0, 0)
    int rows_in, cols_in, seg_pix;
    int RT, NI, segs_per_img, n_segs;
    int n_ntiles;
    int flags;
    unsigned y_bytes;       // bytes of y (and of the residual): buffer descriptors of the epilogue
    int tap_off[MAX_TAPS + 3];     // LDS pixel offset of each tap inside a segment (table-driven path)
};

// Tap schedule.  KS > 0: the KS*KS taps of a forward convolution are unrolled at compile time and
// the B (filter) fragments run through a register ring RING taps deep, refilled one tap after use,
// so every filter load is issued >= RING-1 taps (one whole 3x3 chunk / one 5x5 filter row) before
// its MFMAs and is always OLDER than the activation staging loads it would otherwise queue behind
// (vmcnt retires in order).  KS == 0: table-driven taps (1x1 and the data-gradient classes), double
// buffered.
template <int KS, int MSUB>
struct TapSchedule {
    static constexpr int NT = KS * KS;
    // small wave tiles (few MFMAs per tap) prefetch a whole 3x3 chunk of filters, big ones 3 taps
    static constexpr int RING = (KS == 3) ? (MSUB <= 2 ? 9 : 3) : (KS == 5 ? 5 : 1);
    // ... and also double-buffer the pixel fragments; big wave tiles have the MFMA depth (and the
    // register budget only) for single buffering
    static constexpr bool APREF = MSUB <= 2;
};

template <int KS, int MSUB, int NSUB, int WM, int WN, int NIT>
__global__ void __launch_bounds__(WM * WN * 64) conv_mfma_f32_kernel(const ConvK p) {
    constexpr int NTHR = WM * WN * 64;
    constexpr int MT = MSUB * WM * 32;
    constexpr int NTILE = NSUB * WN * 32;

    float *lds = ds_dynamic_lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tile_n = blockIdx.x % p.n_ntiles;
    const int tile_m = blockIdx.x / p.n_ntiles;
    const int seg0 = tile_m * p.NI;
    const int pix_per_seg = p.RT * p.Wc;
    const int tile_pix = p.NI * p.seg_pix;
    int *out_off = (int *)(lds + tile_pix * LDS_PS);       // [MT] output element offsets, -1 = masked
    float *red = (float *)(out_off + MT);                  // [WM][NTILE][2] statistics scratch

    // ---- row -> output offset table (one division chain per row, not per accumulator) ----
    for (int m = tid; m < MT; m += NTHR) {
        const int seg = m / pix_per_seg;
        const int rem = m - seg * pix_per_seg;
        const int r = rem / p.Wc, c = rem - r * p.Wc;
        const int gseg = seg0 + seg;
        int off = -1;
        if (seg < p.NI && gseg < p.n_segs) {
            const int b = gseg / p.segs_per_img;
            const int rr = (gseg - b * p.segs_per_img) * p.RT + r;
            if (rr < p.Hr) off = ((b * p.Ho + p.OH0 + p.OS * rr) * p.Wo + p.OW0 + p.OS * c) * p.Cout;
        }
        out_off[m] = off;
    }

    // ---- per-lane A-fragment base: this lane's pixel of each 32-row sub-tile ----
    int a_off[MSUB];
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
        const int m = (wm * MSUB + ms) * 32 + l31;
        const int seg = m / pix_per_seg;
        const int rem = m - seg * pix_per_seg;
        const int r = rem / p.Wc, c = rem - r * p.Wc;
        const int pix = (seg < p.NI) ? seg * p.seg_pix + (p.IS * r) * p.cols_in + p.IS * c : 0;
        a_off[ms] = pix * LDS_PS + 4 * lhi;
    }

    // ---- staging descriptors: which global float4 lands in which LDS slot (chunk-invariant).
    //      NIT (compile time) float4 per thread; slots outside the tile / image load from offset 0,
    //      are zeroed by a select and land in a dump slot, so the sequence is branch-free. ----
    int g_off[NIT], l_off[NIT];
    bool g_ok[NIT];
    const int n_items = tile_pix * (CK / 4);
    const int dump_off = tile_pix * LDS_PS + MT + 2 * WM * NTILE;      // 16-byte slot past all tables
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NTHR;
        g_off[it] = 0;
        g_ok[it] = false;
        l_off[it] = dump_off;
        if (idx < n_items) {
            const int pix = idx >> 1, q = idx & 1;
            const int seg = pix / p.seg_pix;
            const int pr = pix - seg * p.seg_pix;
            const int rr = pr / p.cols_in, cc = pr - rr * p.cols_in;
            const int gseg = seg0 + seg;
            l_off[it] = pix * LDS_PS + q * 4;
            if (gseg < p.n_segs) {
                const int b = gseg / p.segs_per_img;
                const int r0 = (gseg - b * p.segs_per_img) * p.RT;
                const int h = p.IS * r0 + p.dh_min + rr, w = p.dw_min + cc;
                if (h >= 0 && h < p.H && w >= 0 && w < p.W) {
                    g_off[it] = ((b * p.H + h) * p.W + w) * p.Cin + q * 4;
                    g_ok[it] = true;
                }
            }
        }
    }

    f32x16 acc[MSUB][NSUB];
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.0f;

    const int n_chunks = p.Cin / CK;
    const int n_base = tile_n * NTILE + wn * NSUB * 32;
    const float *wlane = p.w + (size_t)(n_base + l31) * CK + 4 * lhi;
    const size_t w_tap_stride = (size_t)p.Cout * CK;
    const int NT = (KS > 0) ? KS * KS : p.NT;
    const int last_tap = n_chunks * NT - 1;

    f32x4 st[NIT];
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int it = 0; it < NIT; ++it) st[it] = *(const f32x4 *)(p.x + g_off[it]);

    constexpr int RING = TapSchedule<KS, MSUB>::RING;
    constexpr bool APREF = TapSchedule<KS, MSUB>::APREF;
    f32x4 bq[RING][NSUB];
#pragma unroll
    for (int d = 0; d < RING; ++d)
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
            const int g = d < last_tap ? d : last_tap;
            bq[d][ns] = *(const f32x4 *)(wlane + (size_t)g * w_tap_stride + (size_t)ns * 32 * CK);
        }

    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        __syncthreads();                       // previous chunk's fragment reads are done
#pragma unroll
        for (int it = 0; it < NIT; ++it) *(f32x4 *)(lds + l_off[it]) = g_ok[it] ? st[it] : zero4;
        __syncthreads();
        {   // next chunk's pixels fly while this one computes (the last chunk re-reads its own)
            const int cn = chunk + 1 < n_chunks ? chunk + 1 : chunk;
#pragma unroll
            for (int it = 0; it < NIT; ++it) st[it] = *(const f32x4 *)(p.x + g_off[it] + cn * CK);
        }
        if constexpr (KS > 0) {
            const int g0 = chunk * NT;
            f32x4 a[2][MSUB];
#pragma unroll
            for (int ms = 0; ms < MSUB; ++ms) a[0][ms] = *(const f32x4 *)(lds + a_off[ms]);
#pragma unroll
            for (int t = 0; t < KS * KS; ++t) {
                const int slot = t % RING;
                const int cur = APREF ? (t & 1) : 0;
                // (1) refill the ring slot the PREVIOUS tap consumed with the tap RING-1 ahead of this
                // one, (2) fetch pixel fragments from LDS (the next tap's when double-buffered), then
                // (3) run this tap's MFMAs.  The scheduling fences keep the loads in front of the
                // matrix work they overlap with.
                if (t > 0) {
                    int gn = g0 + t - 1 + RING;
                    gn = gn < last_tap ? gn : last_tap;
#pragma unroll
                    for (int ns = 0; ns < NSUB; ++ns)
                        bq[(t - 1) % RING][ns] =
                            *(const f32x4 *)(wlane + (size_t)gn * w_tap_stride + (size_t)ns * 32 * CK);
                }
                if constexpr (APREF) {
                    if (t + 1 < KS * KS) {
                        const int toff = (((t + 1) / KS) * p.cols_in + ((t + 1) % KS)) * LDS_PS;
#pragma unroll
                        for (int ms = 0; ms < MSUB; ++ms)
                            a[(t + 1) & 1][ms] = *(const f32x4 *)(lds + a_off[ms] + toff);
                    }
                } else if (t > 0) {
                    const int toff = ((t / KS) * p.cols_in + (t % KS)) * LDS_PS;
#pragma unroll
                    for (int ms = 0; ms < MSUB; ++ms) a[0][ms] = *(const f32x4 *)(lds + a_off[ms] + toff);
                }
                __builtin_amdgcn_sched_barrier(0);
                // k-step j multiplies channels {j, 4+j} of the chunk: lanes 0-31 carry channels
                // 0..3, lanes 32-63 channels 4..7, for the A and the B fragment alike.
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
                        for (int ns = 0; ns < NSUB; ++ns)
                            acc[ms][ns] = ds_mfma_32x32x2_f32(a[cur][ms][j], bq[slot][ns][j], acc[ms][ns]);
                __builtin_amdgcn_sched_barrier(0);
            }
            {   // the last tap's slot, refilled for the next chunk
                int gn = g0 + KS * KS - 1 + RING;
                gn = gn < last_tap ? gn : last_tap;
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns)
                    bq[(KS * KS - 1) % RING][ns] =
                        *(const f32x4 *)(wlane + (size_t)gn * w_tap_stride + (size_t)ns * 32 * CK);
            }
        } else {
            f32x4 bnext[NSUB];
            for (int t = 0; t < NT; ++t) {
                int gn = chunk * NT + t + 1;
                gn = gn < last_tap ? gn : last_tap;
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns)
                    bnext[ns] = *(const f32x4 *)(wlane + (size_t)gn * w_tap_stride + (size_t)ns * 32 * CK);
                const int toff = p.tap_off[t] * LDS_PS;
                f32x4 a[MSUB];
#pragma unroll
                for (int ms = 0; ms < MSUB; ++ms) a[ms] = *(const f32x4 *)(lds + a_off[ms] + toff);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
                        for (int ns = 0; ns < NSUB; ++ns)
                            acc[ms][ns] = ds_mfma_32x32x2_f32(a[ms][j], bq[0][ns][j], acc[ms][ns]);
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns) bq[0][ns] = bnext[ns];
            }
        }
    }

    // ---- epilogue: BatchNorm affine / residual / clipped ReLU, optional raw statistics ----
    const int flags = p.flags;
    float sc[NSUB], sh[NSUB], s1[NSUB], s2[NSUB];
    int col[NSUB];
#pragma unroll
    for (int ns = 0; ns < NSUB; ++ns) {
        col[ns] = n_base + ns * 32 + l31;
        sc[ns] = (flags & DS_EPI_AFFINE) ? p.scale[col[ns]] : 1.0f;
        sh[ns] = (flags & DS_EPI_AFFINE) ? p.shift[col[ns]] : 0.0f;
        s1[ns] = 0.0f;
        s2[ns] = 0.0f;
    }
    // Ragged-tile rows get an out-of-range buffer offset (store dropped, load returns zero) and a layer
    // without residual reads "out of range" too: no branch around any memory instruction, so the waits on
    // the residual rows never include the stores issued in between (see conv_mfma_bf16_kernel.h).
    const ds_buffer ybuf = ds_make_buffer(p.y, p.y_bytes);
    const ds_buffer rbuf = ds_make_buffer((flags & DS_EPI_RESIDUAL) ? (const void *)p.res : (const void *)p.y,
                                          (flags & DS_EPI_RESIDUAL) ? p.y_bytes : 0u);
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
        unsigned voff[16][NSUB];
        float resv[16][NSUB];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int off = out_off[(wm * MSUB + ms) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) voff[r][ns] = off >= 0 ? (unsigned)(off + col[ns]) * 4u : DS_BUFFER_OOB;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) resv[r][ns] = ds_buffer_load_f32(rbuf, voff[r][ns]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) {
                float v = acc[ms][ns][r];
                const bool live = voff[r][ns] != DS_BUFFER_OOB;
                s1[ns] += live ? v : 0.0f;
                s2[ns] += live ? v * v : 0.0f;
                v = v * sc[ns] + sh[ns];
                v += resv[r][ns];
                if (flags & DS_EPI_CLIP) v = fminf(fmaxf(v, 0.0f), 20.0f);
                ds_buffer_store_f32(ybuf, voff[r][ns], v);
            }
        }
    }
    if (flags & DS_EPI_STATS) {
        // column sums: fold the two lane halves, then the WM row-waves in a fixed order
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
            s1[ns] += ds_shfl_xor(s1[ns], 32);
            s2[ns] += ds_shfl_xor(s2[ns], 32);
            if (lhi == 0) {
                const int c = wn * NSUB * 32 + ns * 32 + l31;
                red[(wm * NTILE + c) * 2 + 0] = s1[ns];
                red[(wm * NTILE + c) * 2 + 1] = s2[ns];
            }
        }
        __syncthreads();
        for (int c = tid; c < NTILE; c += NTHR) {
            float a1 = 0.0f, a2 = 0.0f;
            for (int k = 0; k < WM; ++k) {
                a1 += red[(k * NTILE + c) * 2 + 0];
                a2 += red[(k * NTILE + c) * 2 + 1];
            }
            float *dst = p.stats + ((size_t)tile_m * p.Cout + tile_n * NTILE + c) * 2;
            dst[0] = a1;
            dst[1] = a2;
        }
    }
}

// ------------------------------------------------------------------------------------------
// host-side tiling plan
// ------------------------------------------------------------------------------------------
struct TileCfg { int MT, NTILE, NTHR, WM, wg_per_cu; };
constexpr int kNumCfg = 3;
constexpr int kNumCU = 256;
constexpr TileCfg kCfg[kNumCfg] = {
    // wg_per_cu: resident workgroups per CU allowed by the register budget (120 / 256 / 208 regs)
    {128, 64, 256, 2, 4},     // conv_mfma_f32_kernel<KS,2,1,2,2,*>
    {160, 128, 256, 1, 2},    // conv_mfma_f32_kernel<KS,5,1,1,4,*>
    {256, 64, 256, 2, 2},     // conv_mfma_f32_kernel<KS,4,1,2,2,*>
};

struct ConvPlan {
    int cfg;
    int nit;            // float4 staging slots per thread: 2, 4 or 8
    int grid;
    size_t lds_bytes;
    int n_mtiles;
    ConvK k;
};

// Choose (cfg, RT, NI): maximise (fraction of MFMA rows that are real pixels) x (occupancy of the
// last round of workgroups), then tile size and segment height.  Full-width segments only.
static int plan_tiles(ConvPlan &pl, int B, int Hr, int Wc, int IS, int ext_h, int ext_w, int Cout,
                      bool stats) {
    DS_REQUIRE((long long)B * pl.k.Ho * pl.k.Wo * pl.k.Cout < (1ll << 30), DS_ERR_BAD_SHAPE);   // 32-bit byte offsets
    pl.k.y_bytes = (unsigned)((long long)B * pl.k.Ho * pl.k.Wo * pl.k.Cout * 4);
    double best_eff = -1.0;
    int best_cfg = -1, best_rt = 0, best_ni = 0;
    for (int c = 0; c < kNumCfg; ++c) {
        const TileCfg &cf = kCfg[c];
        if (Cout % cf.NTILE) continue;
        for (int rt = 1; rt <= Hr; ++rt) {
            if ((long long)rt * Wc > cf.MT) break;
            const int segs_per_img = ds_ceil_div(Hr, rt);
            const long long n_segs = (long long)B * segs_per_img;
            int ni = cf.MT / (rt * Wc);
            if (ni > n_segs) ni = (int)n_segs;
            const int rows_in = IS * (rt - 1) + ext_h, cols_in = IS * (Wc - 1) + ext_w;
            while (ni > 1 && (long long)ni * rows_in * cols_in * (CK / 4) > (long long)MAX_STAGE_IT * cf.NTHR) --ni;
            if ((long long)ni * rows_in * cols_in * (CK / 4) > (long long)MAX_STAGE_IT * cf.NTHR) continue;
            const long long n_mt = ds_ceil_div_ll(n_segs, ni);
            double eff = (double)B * Hr * Wc / ((double)n_mt * cf.MT);
            // wave quantisation: the grid runs in rounds of (CUs x resident workgroups); a half-empty
            // last round idles matrix cores just like masked rows do
            // (measured: 768 workgroups at 2/CU run at 0.84 of the rate of 256 or 512).  A grid that
            // fits in one round is spread evenly over the CUs by the dispatcher.
            const long long blocks = n_mt * (Cout / cf.NTILE), slots = (long long)kNumCU * cf.wg_per_cu;
            if (blocks <= slots) eff *= (double)blocks / (double)(ds_ceil_div_ll(blocks, kNumCU) * kNumCU);
            else eff *= (double)blocks / (double)(ds_ceil_div_ll(blocks, slots) * slots);
            // ties: bigger tiles (more MFMAs per staged byte and per barrier), taller segments
            eff += 1e-9 * rt + 1e-6 * (c == 1 ? 2 : (c == 2 ? 1 : 0));
            if (eff > best_eff) { best_eff = eff; best_cfg = c; best_rt = rt; best_ni = ni; }
        }
    }
    if (best_cfg < 0) return DS_ERR_UNSUPPORTED;
    const TileCfg &cf = kCfg[best_cfg];
    ConvK &k = pl.k;
    k.RT = best_rt;
    k.NI = best_ni;
    k.segs_per_img = ds_ceil_div(Hr, best_rt);
    k.n_segs = B * k.segs_per_img;
    k.rows_in = IS * (best_rt - 1) + ext_h;
    k.cols_in = IS * (Wc - 1) + ext_w;
    k.seg_pix = k.rows_in * k.cols_in;
    k.n_ntiles = Cout / cf.NTILE;
    pl.cfg = best_cfg;
    pl.n_mtiles = ds_ceil_div(k.n_segs, best_ni);
    pl.grid = pl.n_mtiles * k.n_ntiles;
    const int items = k.NI * k.seg_pix * (CK / 4);
    const int its = ds_ceil_div(items, cf.NTHR);
    pl.nit = its <= 2 ? 2 : (its <= 4 ? 4 : 8);
    // pixel tile + row table + statistics scratch + the 16-byte dump slot
    pl.lds_bytes = (size_t)k.NI * k.seg_pix * LDS_PS * 4 + (size_t)cf.MT * 4 + (size_t)cf.WM * cf.NTILE * 2 * 4 + 16;
    (void)stats;
    return DS_OK;
}

static int plan_forward(ConvPlan &pl, const ds_conv_shape *s, bool stats) {
    DS_REQUIRE(s != nullptr, DS_ERR_NULL);
    DS_REQUIRE(s->B > 0 && s->H > 0 && s->W > 0 && s->Cin > 0 && s->Cout > 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(s->KS == 1 || s->KS == 3 || s->KS == 5, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->stride == 1 || s->stride == 2, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->Cin % CK == 0 && s->Cout % 64 == 0, DS_ERR_BAD_SHAPE);
    const int pad = s->KS / 2;
    const int Ho = (s->H + 2 * pad - s->KS) / s->stride + 1;
    const int Wo = (s->W + 2 * pad - s->KS) / s->stride + 1;
    DS_REQUIRE(Ho > 0 && Wo > 0 && Wo <= 128, DS_ERR_BAD_SHAPE);
    DS_REQUIRE((long long)s->B * s->H * s->W * s->Cin < (1ll << 31), DS_ERR_BAD_SHAPE);
    DS_REQUIRE((long long)s->B * Ho * Wo * s->Cout < (1ll << 31), DS_ERR_BAD_SHAPE);
    ConvK &k = pl.k;
    k.H = s->H; k.W = s->W; k.Cin = s->Cin;
    k.Hr = Ho; k.Wc = Wo; k.Ho = Ho; k.Wo = Wo; k.Cout = s->Cout;
    k.IS = s->stride; k.OS = 1; k.OH0 = 0; k.OW0 = 0;
    k.NT = s->KS * s->KS;
    k.dh_min = -pad; k.dw_min = -pad;
    int rc = plan_tiles(pl, s->B, Ho, Wo, s->stride, s->KS, s->KS, s->Cout, stats);
    if (rc != DS_OK) return rc;
    for (int kh = 0; kh < s->KS; ++kh)
        for (int kw = 0; kw < s->KS; ++kw) k.tap_off[kh * s->KS + kw] = kh * k.cols_in + kw;
    return DS_OK;
}

template <int KS, int MSUB, int NSUB, int WM, int WN>
static void launch_nit(const ConvPlan &pl, void *stream) {
    if (pl.nit == 2)
        DS_LAUNCH((conv_mfma_f32_kernel<KS, MSUB, NSUB, WM, WN, 2>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    else if (pl.nit == 4)
        DS_LAUNCH((conv_mfma_f32_kernel<KS, MSUB, NSUB, WM, WN, 4>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    else
        DS_LAUNCH((conv_mfma_f32_kernel<KS, MSUB, NSUB, WM, WN, 8>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
}

template <int KS>
static void launch_ks(const ConvPlan &pl, void *stream) {
    if (pl.cfg == 0) launch_nit<KS, 2, 1, 2, 2>(pl, stream);
    else if (pl.cfg == 1) launch_nit<KS, 5, 1, 1, 4>(pl, stream);
    else launch_nit<KS, 4, 1, 2, 2>(pl, stream);
}

// ks_unrolled = 3 or 5 selects the compile-time tap schedule; anything else the table-driven one
static int launch(const ConvPlan &pl, int ks_unrolled, void *stream) {
    if (ks_unrolled == 3) launch_ks<3>(pl, stream);
    else if (ks_unrolled == 5) launch_ks<5>(pl, stream);
    else launch_ks<0>(pl, stream);
    return ds_last_launch_error();
}

}  // namespace

extern "C" int ds_conv_out_dims(const ds_conv_shape *s, int *Ho, int *Wo) {
    DS_REQUIRE(s && Ho && Wo, DS_ERR_NULL);
    DS_REQUIRE(s->KS >= 1 && (s->stride == 1 || s->stride == 2), DS_ERR_UNSUPPORTED);
    const int pad = s->KS / 2;
    *Ho = (s->H + 2 * pad - s->KS) / s->stride + 1;
    *Wo = (s->W + 2 * pad - s->KS) / s->stride + 1;
    return (*Ho > 0 && *Wo > 0) ? DS_OK : DS_ERR_BAD_SHAPE;
}

extern "C" int ds_conv_stats_rows(const ds_conv_shape *s) {
    ConvPlan pl;
    int rc = plan_forward(pl, s, true);
    return rc == DS_OK ? pl.n_mtiles : rc;
}

// ------------------------------------------------------------------------------------------
// data gradient (autograd of nn.Conv2d, reference model.py:69,73,192,197,202 under
// loss.backward(), train_triplet.py:223,290)
//   stride 1: a plain convolution of dY with the flipped, transposed filter bank.
//   stride 2: dX[2r+ph, 2c+pw] only receives the taps kh = ph (mod 2), kw = pw (mod 2), so the
//   transposed convolution splits into four dense stride-1 convolutions over the dY grid (3x3,
//   3x2, 2x3 and 2x2 taps) whose outputs interleave -- no zero-stuffing, no wasted MFMAs.
// ------------------------------------------------------------------------------------------
namespace {
// taps of parity class p in ascending kernel index; returns count, fills k[] and d[] (input offset)
int s2_class_taps(int p, int *k, int *d) {
    int n = 0;
    for (int kk = p; kk < 5; kk += 2) { k[n] = kk; d[n] = (p + 2 - kk) / 2; ++n; }
    return n;
}
}  // namespace

extern "C" int ds_conv_dgrad_f32(const ds_conv_shape *s, const float *gy, const float *w_dgrad_packed, float *gx,
                                 void *stream) {
    DS_REQUIRE(s && gy && w_dgrad_packed && gx, DS_ERR_NULL);
    DS_REQUIRE(DS_ALIGNED16(gy) && DS_ALIGNED16(w_dgrad_packed) && DS_ALIGNED16(gx), DS_ERR_ALIGNMENT);
    int Ho, Wo;
    int rc = ds_conv_out_dims(s, &Ho, &Wo);
    if (rc != DS_OK) return rc;
    if (s->stride == 1) {
        ds_conv_shape t = *s;
        t.Cin = s->Cout;
        t.Cout = s->Cin;
        ConvPlan pl;
        rc = plan_forward(pl, &t, false);
        if (rc != DS_OK) return rc;
        pl.k.x = gy; pl.k.w = w_dgrad_packed; pl.k.y = gx;
        pl.k.scale = pl.k.shift = pl.k.res = nullptr; pl.k.stats = nullptr;
        pl.k.flags = 0;
        return launch(pl, t.KS, stream);
    }
    DS_REQUIRE(s->KS == 5 && s->stride == 2, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(s->Cout % CK == 0 && s->Cin % 64 == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE((long long)s->B * s->H * s->W * s->Cin < (1ll << 31), DS_ERR_BAD_SHAPE);
    size_t w_off = 0;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            int kh[3], dh[3], kw[3], dw[3];
            const int nh = s2_class_taps(ph, kh, dh), nw = s2_class_taps(pw, kw, dw);
            const int nt = nh * nw;
            const int Hr = (s->H - ph + 1) / 2, Wc = (s->W - pw + 1) / 2;
            const size_t w_this = w_off;
            w_off += (size_t)nt * s->Cout * s->Cin;
            if (Hr <= 0 || Wc <= 0) continue;
            ConvPlan pl;
            ConvK &k = pl.k;
            k.H = Ho; k.W = Wo; k.Cin = s->Cout;
            k.Hr = Hr; k.Wc = Wc; k.Ho = s->H; k.Wo = s->W; k.Cout = s->Cin;
            k.IS = 1; k.OS = 2; k.OH0 = ph; k.OW0 = pw;
            k.NT = nt;
            const int dh_min = dh[nh - 1], dw_min = dw[nw - 1];       // offsets descend with the kernel index
            k.dh_min = dh_min; k.dw_min = dw_min;
            rc = plan_tiles(pl, s->B, Hr, Wc, 1, nh, nw, s->Cin, false);
            if (rc != DS_OK) return rc;
            for (int i = 0; i < nh; ++i)
                for (int j = 0; j < nw; ++j) k.tap_off[i * nw + j] = (dh[i] - dh_min) * k.cols_in + (dw[j] - dw_min);
            k.x = gy; k.w = w_dgrad_packed + w_this; k.y = gx;
            k.scale = k.shift = k.res = nullptr; k.stats = nullptr;
            k.flags = 0;
            rc = launch(pl, 0, stream);
            if (rc != DS_OK) return rc;
        }
    return DS_OK;
}

extern "C" int ds_conv_plan_describe(const ds_conv_shape *s, int *out8) {
    DS_REQUIRE(out8, DS_ERR_NULL);
    ConvPlan pl;
    int rc = plan_forward(pl, s, false);
    if (rc != DS_OK) return rc;
    out8[0] = kCfg[pl.cfg].MT; out8[1] = kCfg[pl.cfg].NTILE; out8[2] = pl.k.RT; out8[3] = pl.k.NI;
    out8[4] = pl.grid; out8[5] = (int)pl.lds_bytes; out8[6] = pl.nit; out8[7] = pl.n_mtiles;
    return DS_OK;
}

extern "C" int ds_conv_fwd_f32(const ds_conv_shape *s, const float *x, const float *w_packed,
                               const float *scale, const float *shift, const float *residual,
                               float *y, float *stats_partial, int flags, void *stream) {
    DS_REQUIRE(x && w_packed && y, DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_AFFINE) || (scale && shift), DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_RESIDUAL) || residual, DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_STATS) || stats_partial, DS_ERR_NULL);
    DS_REQUIRE(DS_ALIGNED16(x) && DS_ALIGNED16(w_packed) && DS_ALIGNED16(y), DS_ERR_ALIGNMENT);
    ConvPlan pl;
    int rc = plan_forward(pl, s, (flags & DS_EPI_STATS) != 0);
    if (rc != DS_OK) return rc;
    pl.k.x = x; pl.k.w = w_packed; pl.k.y = y;
    pl.k.scale = scale; pl.k.shift = shift; pl.k.res = residual; pl.k.stats = stats_partial;
    pl.k.flags = flags;
    return launch(pl, s->KS, stream);
}
