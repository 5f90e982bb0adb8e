// conv_mfma_bf16_kernel.h -- the bf16 / bf16x3 implicit-GEMM convolution kernel template and its launch
// dispatch.  Included by the per-(kernel size, arithmetic) translation units conv_mfma_bf16_k*.hip, which
// exist only so that the ~50 instantiations compile in parallel; the planner and the C ABI live in
// conv_mfma_bf16.hip.  Design notes: DESIGN_LOG.md section 3.3.
#pragma once
#include <ds_device.h>
#include "ds_common.h"


constexpr int CKB = 16;             // input channels per chunk = K of one bf16 MFMA
constexpr int PSB = 48;             // bytes per staged pixel record: 16 bf16 + 16 B pad


struct ConvKB {
    const float *x;
    const __bf16 *w_hi, *w_lo;      // packed [Cin/16][tap][Cout][16]
    float *y;
    const float *scale, *shift, *res;
    float *stats;
    int H, W, Cin;
    int Hr, Wc, Ho, Wo, Cout;
    int IS;
    int dh_min, dw_min;
    int rows_in, cols_in, seg_pix;
    int pitch, half;                // LDS records per tile row; first odd-column slot (stride-2 de-interleave)
    int RT, NI, segs_per_img, n_segs;
    int n_ntiles;
    int flags;
    unsigned y_bytes;               // size of y (and of the residual) in bytes, for the buffer descriptors
    int OS, OH0, OW0;               // output pixel (r, c) of the tile grid lands at (OS*r + OH0, OS*c + OW0) of y
    // BNB epilogue (ds_conv_dgrad_bnbwd_bf16): this launch is a data gradient whose output is dL/d(activation) of a
    // BatchNorm + clipped-ReLU layer; z = that layer's convolution output (same shape as y), tables [G][Cout] per
    // member of the batch, bn_mtiles = M tiles per member
    const float *bn_z, *bn_mean, *bn_invstd, *bn_msc, *bn_msh;
    int bn_mtiles;
    // bn_msc == nullptr: the mask comes from the layer's stored activation, passed as `res` (it is then NOT added):
    // the layers whose activation had a residual added before the clip.  The partial rows of member m start at row
    // m * bn_rows_member + bn_row0 (several launches -- the parity classes of a stride-2 data gradient -- share a table).
    int bn_rows_member, bn_row0;
};

struct PlanB {
    int cfg, grid, n_mtiles, nit;
    size_t lds_bytes;
    ConvKB k;
};

// one entry point per translation unit (kernel size x arithmetic)
void ds_bf16_launch_k3x3(const PlanB &pl, void *stream);
void ds_bf16_launch_k5x3(const PlanB &pl, void *stream);
void ds_bf16_launch_k3x1(const PlanB &pl, void *stream);
void ds_bf16_launch_k5x1(const PlanB &pl, void *stream);
void ds_bf16_launch_k3x3g(const PlanB &pl, void *stream);      // 3x3 bf16x3 with the BatchNorm-backward epilogue

#ifdef DS_BF16_KERNEL_TU
namespace {

// NIT: float4 staging slots per thread (compile time, so all loads of a chunk are issued together);
// PREF: the next chunk's pixels are loaded into registers BEFORE this chunk's matrix work and converted
// / written to LDS after it (small tiles); otherwise they are loaded right after the barrier (big
// tiles, where 16 slots would not fit next to the accumulators).
// BNB: the epilogue is the first half of a BatchNorm backward (see ConvKB): y = (acc [+ res]) * [0 < z*msc+msh < 20]
// and the per-tile partial sums are { sum y, sum y * xhat }, xhat = (z - mean) * invstd, instead of { sum, sum sq }.
template <int KS, int MSUB, int NSUB, int WM, int WN, bool X3, int NIT, bool PREF, bool BNB = false>
__global__ void __launch_bounds__(WM * WN * 64) conv_mfma_bf16_kernel(const ConvKB p) {
    constexpr int NTHR = WM * WN * 64;
    constexpr int MT = MSUB * WM * 32;
    constexpr int NTILE = NSUB * WN * 32;
    constexpr int NT = KS * KS;
    constexpr int RING = (KS == 3) ? (MSUB <= 2 ? 9 : 3) : (KS == 5 ? 5 : 1);
    // 160x64 register tiles run one wave per SIMD: nothing else hides LDS latency there, so the next
    // tap's pixel fragments are fetched before (X3: in between) this tap's matrix work
    constexpr bool APREF = (MSUB * NSUB >= 8);
    constexpr bool ILV = X3 && APREF;

    char *lds = (char *)ds_dynamic_lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tile_n = blockIdx.x % p.n_ntiles;
    const int tile_m = blockIdx.x / p.n_ntiles;
    const int seg0 = tile_m * p.NI;
    const int pix_per_seg = p.RT * p.Wc;
    const int tile_pix = p.NI * p.seg_pix;
    // the pixel-tile region doubles as the epilogue's transposition buffers (32 x (NSUB*32+4) floats per wave)
    constexpr int EPI_BYTES = WM * WN * 32 * (NSUB * 32 + 4) * 4;
    const int tile_bytes = (X3 ? 2 : 1) * tile_pix * PSB;
    const int stage_bytes = tile_bytes > EPI_BYTES ? tile_bytes : EPI_BYTES;
    char *lds_hi = lds;                                        // [tile_pix][PSB]
    char *lds_lo = lds + (X3 ? tile_pix * PSB : 0);
    int *out_off = (int *)(lds + stage_bytes);                 // [MT]
    int *seg_lo = out_off + MT;                                // [NI] first in-image row of each segment's tile
    int *seg_cnt = seg_lo + p.NI;                              // [NI] number of in-image rows
    float *red = (float *)(seg_cnt + p.NI);                    // [WM][NTILE][2]

    // the first filter slices are requested before anything else: their latency hides behind the tables
    const int n_chunks = p.Cin / CKB;
    const int n_base = tile_n * NTILE + wn * NSUB * 32;
    const size_t lane_w = ((size_t)(n_base + l31) * CKB + 8 * lhi);      // in bf16 elements
    const size_t w_tap_stride = (size_t)p.Cout * CKB;
    const int last_tap = n_chunks * NT - 1;

    bf16x8 bq_hi[RING][NSUB], bq_lo[X3 ? RING : 1][NSUB];
#pragma unroll
    for (int d = 0; d < RING; ++d)
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) {
            const int g = d < last_tap ? d : last_tap;
            const size_t o = lane_w + (size_t)g * w_tap_stride + (size_t)ns * 32 * CKB;
            bq_hi[d][ns] = *(const bf16x8 *)(p.w_hi + o);
            if constexpr (X3) bq_lo[d][ns] = *(const bf16x8 *)(p.w_lo + o);
        }

    const float rcp_pps = 1.0f / (float)pix_per_seg, rcp_wc = 1.0f / (float)p.Wc, rcp_w = 1.0f / (float)p.W,
                rcp_spi = 1.0f / (float)p.segs_per_img;
    for (int seg = tid; seg < p.NI; seg += NTHR) {
        const int gseg = seg0 + seg;
        int lo = 0, cnt = 0;
        if (gseg < p.n_segs) {
            const int b = ds_div_small(gseg, p.segs_per_img, rcp_spi);
            const int h0 = p.IS * (gseg - b * p.segs_per_img) * p.RT + p.dh_min;      // image row of tile row 0
            lo = h0 < 0 ? -h0 : 0;
            const int hi = p.H - h0 < p.rows_in ? p.H - h0 : p.rows_in;
            cnt = hi > lo ? hi - lo : 0;
        }
        seg_lo[seg] = lo;
        seg_cnt[seg] = cnt;
    }

    f32x16 acc[MSUB][NSUB];
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ms][ns][r] = 0.0f;

    // ---- staging descriptors (chunk-invariant) of this thread's items: 4 channels of one in-image pixel each.
    // Item idx -> (quarter q, column c, valid row) in segment order; slots past the last item load x[0..3]
    // and drop it into the unused pad bytes of pixel record 0, so the chunk loop has no branches.
    int g_off[NIT], l_off[NIT];
    __syncthreads();                            // seg_lo / seg_cnt are complete
    {
        // a thread's items ascend by NTHR/4 pixels: (row, column) advance incrementally and the segment
        // walk never restarts
        const int q = tid & 3;
        const int dvr = ds_div_small(NTHR / 4, p.W, rcp_w), dc = NTHR / 4 - dvr * p.W;
        int vr = ds_div_small(tid >> 2, p.W, rcp_w);
        int c = (tid >> 2) - vr * p.W;
        int seg = -1, row0 = 0, cnt = 0, lo = 0, img_row = 0;
        auto next_seg = [&]() {
            row0 += cnt;
            ++seg;
            cnt = 0;
            if (seg < p.NI) {
                cnt = seg_cnt[seg];
                lo = seg_lo[seg];
                const int gseg = seg0 + seg;
                const int b = ds_div_small(gseg, p.segs_per_img, rcp_spi);
                img_row = b * p.H + p.IS * (gseg - b * p.segs_per_img) * p.RT + p.dh_min;   // of tile row 0
            }
        };
        next_seg();
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            while (seg < p.NI && vr >= row0 + cnt) next_seg();
            g_off[it] = 0;
            l_off[it] = 32;
            if (seg < p.NI) {
                const int rr = lo + vr - row0;
                // stride-2 layers keep even tile columns in slots [0, half) and odd ones in [half, cols_in),
                // so that the 32 lanes of a fragment read (stride-2 columns) touch CONSECUTIVE records
                const int cc = c - p.dw_min;
                const int pc = (p.IS == 2) ? ((cc & 1) ? p.half + (cc >> 1) : (cc >> 1)) : cc;
                g_off[it] = ((img_row + rr) * p.W + c) * p.Cin + q * 4;
                l_off[it] = (seg * p.seg_pix + rr * p.pitch + pc) * PSB + q * 8;
            }
            c += dc;
            vr += dvr;
            if (c >= p.W) {
                c -= p.W;
                ++vr;
            }
        }
    }
    f32x4 st[NIT];
    if constexpr (PREF) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) st[it] = *(const f32x4 *)(p.x + g_off[it]);
    }
    // ---- everything below overlaps the first chunk's loads ----
    // Only in-image pixels are ever staged: the zero halo (and the row padding) is written once, here.
    for (int i = tid; i < tile_bytes / 16; i += NTHR) *(f32x4 *)(lds + 16 * i) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int m = tid; m < MT; m += NTHR) {
        const int seg = ds_div_small(m, pix_per_seg, rcp_pps);
        const int rem = m - seg * pix_per_seg;
        const int r = ds_div_small(rem, p.Wc, rcp_wc), c = rem - r * p.Wc;
        const int gseg = seg0 + seg;
        int off = -1;
        if (seg < p.NI && gseg < p.n_segs) {
            const int b = ds_div_small(gseg, p.segs_per_img, rcp_spi);
            const int rr = (gseg - b * p.segs_per_img) * p.RT + r;
            if (rr < p.Hr) off = ((b * p.Ho + p.OS * rr + p.OH0) * p.Wo + p.OS * c + p.OW0) * p.Cout;
        }
        out_off[m] = off;
    }
    // Which pixel of its 32-pixel sub-tile a lane owns is free (the epilogue un-permutes): it is chosen so
    // that the two 16-lane SERVICE GROUPS of a ds_read_b128 -- lanes {0-3,12-15,20-27} and {4-11,16-19,
    // 28-31} -- each read 16 CONSECUTIVE pixels, i.e. consecutive 48-byte records that walk all 64 banks.
    const int lpix = (l31 < 4 || l31 >= 28) ? l31
                   : (l31 < 12) ? l31 + 12 : (l31 < 16) ? l31 - 8 : (l31 < 20) ? l31 + 8 : l31 - 12;
    int a_off[MSUB];                                           // byte offset of this lane's fragment
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
        const int m = (wm * MSUB + ms) * 32 + lpix;
        const int seg = ds_div_small(m, pix_per_seg, rcp_pps);
        const int rem = m - seg * pix_per_seg;
        const int r = ds_div_small(rem, p.Wc, rcp_wc), c = rem - r * p.Wc;
        const int pix = (seg < p.NI) ? seg * p.seg_pix + (p.IS * r) * p.pitch + c : 0;
        a_off[ms] = pix * PSB + 16 * lhi;
    }

    // CVI: the next chunk's pixels are split into hi/lo IN PLACE (4 floats -> 4+4 bf16, the same 16 bytes)
    // between the MFMAs of this chunk's later taps; between two chunks only the LDS writes remain.
    constexpr bool CVI = ILV && PREF;
    constexpr int CV_FIRST = 3;                 // taps left for the prefetch to land before conversion starts
    auto split_item = [&](int it) {
        const f32x4 v = st[it];
        bf16x8 pk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const __bf16 h = (__bf16)v[j];
            pk[j] = h;
            pk[4 + j] = (__bf16)(v[j] - (float)h);
        }
        st[it] = __builtin_bit_cast(f32x4, pk);
    };
    if constexpr (CVI) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) split_item(it);
    }

    for (int chunk = 0; chunk < n_chunks; ++chunk) {
        __syncthreads();                       // previous chunk's fragment reads are done
        if constexpr (!PREF) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) st[it] = *(const f32x4 *)(p.x + g_off[it] + chunk * CKB);
        }
        if constexpr (CVI) {                   // records were split during the previous chunk's taps
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const bf16x8 pk = __builtin_bit_cast(bf16x8, st[it]);
                *(bf16x4 *)(lds_hi + l_off[it]) = __builtin_shufflevector(pk, pk, 0, 1, 2, 3);
                *(bf16x4 *)(lds_lo + l_off[it]) = __builtin_shufflevector(pk, pk, 4, 5, 6, 7);
            }
        } else {
            // ---- convert: f32 pixels -> bf16 hi (+ lo) records ----
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const f32x4 v = st[it];
                bf16x4 h;
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = (__bf16)v[j];
                *(bf16x4 *)(lds_hi + l_off[it]) = h;
                if constexpr (X3) {
                    bf16x4 l;
#pragma unroll
                    for (int j = 0; j < 4; ++j) l[j] = (__bf16)(v[j] - (float)h[j]);
                    *(bf16x4 *)(lds_lo + l_off[it]) = l;
                }
            }
        }
        __syncthreads();
        if constexpr (PREF) {                  // next chunk's pixels fly while this one computes
            const int cn = chunk + 1 < n_chunks ? chunk + 1 : chunk;
#pragma unroll
            for (int it = 0; it < NIT; ++it) st[it] = *(const f32x4 *)(p.x + g_off[it] + cn * CKB);
        }
        const int g0 = chunk * NT;
        bf16x8 a_hi[APREF ? 2 : 1][MSUB], a_lo[APREF ? 2 : 1][X3 ? MSUB : 1];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if constexpr (ILV) {
                // One wave per SIMD: nothing but this wave's own instruction order hides the LDS / L2
                // latency, so the next tap's fragment reads and the ring refills are dealt out between
                // PAIRS of MFMAs (each MFMA holds the matrix pipe for 8 issue slots) instead of in a
                // block in front of them.
                constexpr int NM = 3 * MSUB * NSUB, NA = 2 * MSUB, NL = NA + 2 * NSUB;
                const int slot = t % RING, cur = t & 1;
                auto tap_off = [&](int tt) {
                    const int kw = tt % KS;
                    return ((tt / KS) * p.pitch + (p.IS == 2 ? (kw & 1) * p.half + (kw >> 1) : kw)) * PSB;
                };
                if (t == 0) {
                    const int toff = tap_off(0);
#pragma unroll
                    for (int ms = 0; ms < MSUB; ++ms) {
                        a_lo[0][ms] = *(const bf16x8 *)(lds_lo + a_off[ms] + toff);
                        a_hi[0][ms] = *(const bf16x8 *)(lds_hi + a_off[ms] + toff);
                    }
                }
                const bool more = t + 1 < NT;
                const char *nlo = lds_lo + tap_off(more ? t + 1 : t);
                const char *nhi = lds_hi + tap_off(more ? t + 1 : t);
                int gn = g0 + t - 1 + RING;                 // refills the slot the previous tap consumed
                gn = gn < last_tap ? gn : last_tap;
                const int rslot = (t + RING - 1) % RING;
                const __bf16 *rhi = p.w_hi + lane_w + (size_t)gn * w_tap_stride;
                const __bf16 *rlo = p.w_lo + lane_w + (size_t)gn * w_tap_stride;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NM; ++q) {
                    const int term = q / (MSUB * NSUB), ms = (q % (MSUB * NSUB)) / NSUB, ns = q % NSUB;
                    if (term == 0) acc[ms][ns] = ds_mfma_32x32x16_bf16(bq_hi[slot][ns], a_lo[cur][ms], acc[ms][ns]);
                    else if (term == 1) acc[ms][ns] = ds_mfma_32x32x16_bf16(bq_lo[slot][ns], a_hi[cur][ms], acc[ms][ns]);
                    else acc[ms][ns] = ds_mfma_32x32x16_bf16(bq_hi[slot][ns], a_hi[cur][ms], acc[ms][ns]);
                    if constexpr (CVI) {
                        if (!(q & 1) && t >= CV_FIRST) {          // one pixel item split per even slot
                            constexpr int SPAN = NT - CV_FIRST;
                            const int it = (NIT * (t - CV_FIRST)) / SPAN + (q >> 1);
                            if (it < (NIT * (t - CV_FIRST + 1)) / SPAN) {
                                split_item(it);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                    if (q & 1) {
                        const int l = q >> 1;
                        if (l < NA) {
                            if (more) {
                                const int lm = l % MSUB;
                                if (l < MSUB) a_lo[cur ^ 1][lm] = *(const bf16x8 *)(nlo + a_off[lm]);
                                else a_hi[cur ^ 1][lm] = *(const bf16x8 *)(nhi + a_off[lm]);
                            }
                        } else if (l < NL) {
                            const int ln = (l - NA) >> 1;
                            if (t > 0) {
                                if ((l - NA) & 1) bq_lo[rslot][ln] = *(const bf16x8 *)(rlo + (size_t)ln * 32 * CKB);
                                else bq_hi[rslot][ln] = *(const bf16x8 *)(rhi + (size_t)ln * 32 * CKB);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            const int slot = t % RING;
            if (t > 0) {                       // refill the slot the previous tap consumed
                int gn = g0 + t - 1 + RING;
                gn = gn < last_tap ? gn : last_tap;
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns) {
                    const size_t o = lane_w + (size_t)gn * w_tap_stride + (size_t)ns * 32 * CKB;
                    bq_hi[(t - 1) % RING][ns] = *(const bf16x8 *)(p.w_hi + o);
                    if constexpr (X3) bq_lo[(t - 1) % RING][ns] = *(const bf16x8 *)(p.w_lo + o);
                }
            }
            auto tap_off = [&](int tt) {
                const int kw = tt % KS;
                return ((tt / KS) * p.pitch + (p.IS == 2 ? (kw & 1) * p.half + (kw >> 1) : kw)) * PSB;
            };
            const int cur = APREF ? (t & 1) : 0;
            if (!APREF || t == 0) {
                const int toff = tap_off(t);
#pragma unroll
                for (int ms = 0; ms < MSUB; ++ms) {
                    a_hi[cur][ms] = *(const bf16x8 *)(lds_hi + a_off[ms] + toff);
                    if constexpr (X3) a_lo[cur][ms] = *(const bf16x8 *)(lds_lo + a_off[ms] + toff);
                }
            }
            if constexpr (APREF) {
                if (t + 1 < NT) {
                    const int toff = tap_off(t + 1);
#pragma unroll
                    for (int ms = 0; ms < MSUB; ++ms) {
                        a_hi[cur ^ 1][ms] = *(const bf16x8 *)(lds_hi + a_off[ms] + toff);
                        if constexpr (X3) a_lo[cur ^ 1][ms] = *(const bf16x8 *)(lds_lo + a_off[ms] + toff);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // term-major order: consecutive MFMAs always hit DIFFERENT accumulators (a dependent
            // accumulate chain stalls the matrix pipe); small cross terms first, the leading term last
            if constexpr (X3) {
#pragma unroll
                for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
                    for (int ns = 0; ns < NSUB; ++ns)
                        acc[ms][ns] = ds_mfma_32x32x16_bf16(bq_hi[slot][ns], a_lo[cur][ms], acc[ms][ns]);
#pragma unroll
                for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
                    for (int ns = 0; ns < NSUB; ++ns)
                        acc[ms][ns] = ds_mfma_32x32x16_bf16(bq_lo[slot][ns], a_hi[cur][ms], acc[ms][ns]);
            }
#pragma unroll
            for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns)
                    acc[ms][ns] = ds_mfma_32x32x16_bf16(bq_hi[slot][ns], a_hi[cur][ms], acc[ms][ns]);
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            int gn = g0 + NT - 1 + RING;
            gn = gn < last_tap ? gn : last_tap;
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) {
                const size_t o = lane_w + (size_t)gn * w_tap_stride + (size_t)ns * 32 * CKB;
                bq_hi[(NT - 1) % RING][ns] = *(const bf16x8 *)(p.w_hi + o);
                if constexpr (X3) bq_lo[(NT - 1) % RING][ns] = *(const bf16x8 *)(p.w_lo + o);
            }
        }
    }

    // ---- epilogue ----
    // The filters were the A operand of every MFMA, so the accumulators hold the TRANSPOSED product: a lane
    // owns one output pixel (l31 of the 32-pixel sub-tile) and, per register quad g, four consecutive output
    // channels 8g + 4*lhi .. +3.  Each 32-pixel sub-tile is turned around through a wave-private LDS buffer
    // (the pixel tile's space, free now) so that residual loads and stores move whole pixel rows: the
    // NSUB*32 channels of a pixel are contiguous across NSUB*8 lanes, 16 bytes per lane.
    constexpr int TP = NSUB * 32 + 4;           // buffer row pitch in floats (conflict-free 16-byte writes)
    constexpr int LPP = NSUB * 8;               // lanes per pixel row
    constexpr int PPI = 64 / LPP;               // pixel rows per instruction
    constexpr int NRI = 32 / PPI;               // instructions per sub-tile
    const int flags = p.flags;
    __syncthreads();                            // every wave is done reading the pixel tile
    float *tb = (float *)lds + wave * (32 * TP);
    const int my_c = (lane % LPP) * 4, my_p = lane / LPP;
    const int col = n_base + my_c;
    f32x4 sc4 = {1.0f, 1.0f, 1.0f, 1.0f}, sh4 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (flags & DS_EPI_AFFINE) {
        sc4 = *(const f32x4 *)(p.scale + col);
        sh4 = *(const f32x4 *)(p.shift + col);
    }
    // Rows of a ragged tile get an out-of-range buffer offset (the store is dropped, the load returns
    // zeros) and a layer without residual reads "out of range" too: no branch around any memory
    // instruction, so the waits on the residual rows never include the stores issued in between.
    const ds_buffer ybuf = ds_make_buffer(p.y, p.y_bytes);
    const ds_buffer rbuf = ds_make_buffer((flags & DS_EPI_RESIDUAL) ? (const void *)p.res : (const void *)p.y,
                                          (flags & DS_EPI_RESIDUAL) ? p.y_bytes : 0u);
    float ps1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, ps2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    unsigned voff[2][NRI];
    f32x4 resv[2][NRI];
    f32x4 zv[BNB ? 2 : 1][BNB ? NRI : 1];
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4 mu4 = zero4, is4 = zero4, msc4 = zero4, msh4 = zero4;
    const ds_buffer zbuf = ds_make_buffer(BNB ? (const void *)p.bn_z : (const void *)p.y, BNB ? p.y_bytes : 0u);
    if constexpr (BNB) {
        const size_t mo = (size_t)(tile_m / p.bn_mtiles) * p.Cout + col;      // this tile's member of the batch
        mu4 = *(const f32x4 *)(p.bn_mean + mo);
        is4 = *(const f32x4 *)(p.bn_invstd + mo);
        if (p.bn_msc) {
            msc4 = *(const f32x4 *)(p.bn_msc + mo);
            msh4 = *(const f32x4 *)(p.bn_msh + mo);
        }
    }
    const bool mask_from_res = BNB && p.bn_msc == nullptr;
    auto fetch_rows = [&](int ms, int buf) {    // byte offsets and residual rows of sub-tile ms
#pragma unroll
        for (int k = 0; k < NRI; ++k) {
            const int off = out_off[(wm * MSUB + ms) * 32 + k * PPI + my_p];
            voff[buf][k] = off >= 0 ? (unsigned)(off + col) * 4u : DS_BUFFER_OOB;
        }
#pragma unroll
        for (int k = 0; k < NRI; ++k) resv[buf][k] = ds_buffer_load_f32x4(rbuf, voff[buf][k]);
        if constexpr (BNB) {
#pragma unroll
            for (int k = 0; k < NRI; ++k) zv[buf][k] = ds_buffer_load_f32x4(zbuf, voff[buf][k]);
        }
    };
    fetch_rows(0, 0);
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
        const int cb = ms & 1;
        if (ms + 1 < MSUB) fetch_rows(ms + 1, cb ^ 1);
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[ms][ns][4 * g + j];
                *(f32x4 *)(tb + lpix * TP + ns * 32 + 8 * g + 4 * lhi) = v;
            }
        ds_wave_sync();
#pragma unroll
        for (int k = 0; k < NRI; ++k) {
            f32x4 v = *(const f32x4 *)(tb + (k * PPI + my_p) * TP + my_c);
            const bool live = voff[cb][k] != DS_BUFFER_OOB;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = v[j];
                if constexpr (BNB) {
                    // the layer's clipped-ReLU mask from its own pre-activation (the fma bn_apply_kernel evaluates)
                    const float zz = zv[cb][k][j];
                    const float rr = resv[cb][k][j];
                    const float am = mask_from_res ? rr : ds_bn_affine(zz, msc4[j], msh4[j]);
                    t += mask_from_res ? 0.0f : rr;
                    t = (am > 0.0f && am < 20.0f) ? t : 0.0f;
                    ps1[j] += live ? t : 0.0f;
                    ps2[j] += live ? t * ((zz - mu4[j]) * is4[j]) : 0.0f;
                    v[j] = t;
                    continue;
                }
                if (flags & DS_EPI_STATS) {
                    ps1[j] += live ? t : 0.0f;
                    ps2[j] += live ? t * t : 0.0f;
                }
                t = t * sc4[j] + sh4[j];
                t += resv[cb][k][j];
                if (flags & DS_EPI_CLIP) t = fminf(fmaxf(t, 0.0f), 20.0f);
                v[j] = t;
            }
            ds_buffer_store_f32x4(ybuf, voff[cb][k], v);
        }
        ds_wave_sync();                         // the buffer is rewritten by the next sub-tile
    }
    if (flags & DS_EPI_STATS) {
        // per-channel sums over this wave's pixels: fold the PPI lane groups that share a channel quad
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int mk = LPP; mk < 64; mk <<= 1) {
                ps1[j] += ds_shfl_xor(ps1[j], mk);
                ps2[j] += ds_shfl_xor(ps2[j], mk);
            }
            if (my_p == 0) {
                const int c = wn * NSUB * 32 + my_c + j;
                red[(wm * NTILE + c) * 2 + 0] = ps1[j];
                red[(wm * NTILE + c) * 2 + 1] = ps2[j];
            }
        }
        __syncthreads();
        for (int c = tid; c < NTILE; c += NTHR) {
            float a1 = 0.0f, a2 = 0.0f;
            for (int k = 0; k < WM; ++k) {
                a1 += red[(k * NTILE + c) * 2 + 0];
                a2 += red[(k * NTILE + c) * 2 + 1];
            }
            size_t row = tile_m;
            if constexpr (BNB) row = (size_t)(tile_m / p.bn_mtiles) * p.bn_rows_member + p.bn_row0 + tile_m % p.bn_mtiles;
            float *dst = p.stats + (row * p.Cout + tile_n * NTILE + c) * 2;
            dst[0] = a1;
            dst[1] = a2;
        }
    }
}

template <int KS, int MSUB, int NSUB, int WM, int WN, bool X3, bool BNB = false>
static void launch_nit_b(const PlanB &pl, void *stream) {
    if (pl.nit <= 4)
        DS_LAUNCH((conv_mfma_bf16_kernel<KS, MSUB, NSUB, WM, WN, X3, 4, true, BNB>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    else if (pl.nit <= 8)      // register-prefetch the next chunk where the accumulators leave room (MSUB <= 4)
        DS_LAUNCH((conv_mfma_bf16_kernel<KS, MSUB, NSUB, WM, WN, X3, 8, (MSUB <= 4), BNB>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    else
        DS_LAUNCH((conv_mfma_bf16_kernel<KS, MSUB, NSUB, WM, WN, X3, 16, false, BNB>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
}

template <int KS, int MSUB, int WM, int WN, bool BNB = false>
static void launch_big_b(const PlanB &pl, void *stream) {
    constexpr int NTHR = WM * WN * 64;
    if (pl.nit <= 8)
        DS_LAUNCH_BIG_LDS((conv_mfma_bf16_kernel<KS, MSUB, 2, WM, WN, true, 8, true, BNB>), pl.grid, NTHR, pl.lds_bytes, stream, pl.k);
    else if (pl.nit <= 16)
        DS_LAUNCH_BIG_LDS((conv_mfma_bf16_kernel<KS, MSUB, 2, WM, WN, true, 16, true, BNB>), pl.grid, NTHR, pl.lds_bytes, stream, pl.k);
    else
        DS_LAUNCH_BIG_LDS((conv_mfma_bf16_kernel<KS, MSUB, 2, WM, WN, true, 32, false, BNB>), pl.grid, NTHR, pl.lds_bytes, stream, pl.k);
}

template <int KS, bool X3, bool BNB = false>
static void launch_b(const PlanB &pl, void *stream) {
    if (pl.cfg == 0) launch_nit_b<KS, 2, 1, 2, 2, X3, BNB>(pl, stream);
    else if (pl.cfg == 1) launch_nit_b<KS, 5, 1, 1, 4, X3, BNB>(pl, stream);
    else if (pl.cfg == 2) launch_nit_b<KS, 4, 1, 2, 2, X3, BNB>(pl, stream);
    else if constexpr (X3) {                    // 160x64 register tiles, opt-in LDS sizes
        if (pl.cfg == 3) launch_big_b<KS, 5, 1, 2, BNB>(pl, stream);
        else if (pl.cfg == 4) launch_big_b<KS, 5, 1, 4, BNB>(pl, stream);
        else if (pl.cfg == 5) launch_big_b<KS, 5, 2, 2, BNB>(pl, stream);
        else if (pl.cfg == 6) launch_big_b<KS, 5, 2, 1, BNB>(pl, stream);
        else if (pl.cfg == 7) launch_big_b<KS, 4, 1, 2, BNB>(pl, stream);
        else launch_big_b<KS, 4, 1, 4, BNB>(pl, stream);
    }
}

}  // namespace
#endif
