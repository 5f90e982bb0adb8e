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

#ifndef DS_BF16_ILV
#define DS_BF16_ILV 1
#endif

namespace {

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
};

// NIT: float4 staging slots per thread (compile time, so all loads of a chunk are issued together);
// PREF: the next chunk's pixels are loaded into registers BEFORE this chunk's matrix work and converted
// / written to LDS after it (small tiles); otherwise they are loaded right after the barrier (big
// tiles, where 16 slots would not fit next to the accumulators).
template <int KS, int MSUB, int NSUB, int WM, int WN, bool X3, int NIT, bool PREF>
__global__ void __launch_bounds__(WM * WN * 64) conv_mfma_bf16_kernel(const ConvKB p) {
    constexpr int NTHR = WM * WN * 64;
    constexpr int MT = MSUB * WM * 32;
    constexpr int NTILE = NSUB * WN * 32;
    constexpr int NT = KS * KS;
    constexpr int RING = (KS == 3) ? (MSUB <= 2 ? 9 : 3) : (KS == 5 ? 5 : 1);
    // 160x64 register tiles run one wave per SIMD: nothing else hides LDS latency there, so the next
    // tap's pixel fragments are fetched before (X3: in between) this tap's matrix work
    constexpr bool APREF = (MSUB * NSUB >= 8);
    constexpr bool ILV = DS_BF16_ILV && X3 && APREF;

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
    auto fetch_rows = [&](int ms, int buf) {    // byte offsets and residual rows of sub-tile ms
#pragma unroll
        for (int k = 0; k < NRI; ++k) {
            const int off = out_off[(wm * MSUB + ms) * 32 + k * PPI + my_p];
            voff[buf][k] = off >= 0 ? (unsigned)(off + col) * 4u : DS_BUFFER_OOB;
        }
#pragma unroll
        for (int k = 0; k < NRI; ++k) resv[buf][k] = ds_buffer_load_f32x4(rbuf, voff[buf][k]);
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
            float *dst = p.stats + ((size_t)tile_m * p.Cout + tile_n * NTILE + c) * 2;
            dst[0] = a1;
            dst[1] = a2;
        }
    }
}

// OIHW f32 -> [Cin/16][tap][Cout][16] bf16 hi (+ lo = bf16(w - hi))
// dgrad != 0: the transposed, spatially flipped bank of the stride-1 data gradient (N = Cin, K = Cout)
__global__ void __launch_bounds__(256) pack_conv_weight_bf16_kernel(const float *w, __bf16 *hi, __bf16 *lo, int Cout,
                                                                    int Cin, int KS, int dgrad) {
    const int T = KS * KS;
    const long long n = (long long)Cout * Cin * T;
    const int N = dgrad ? Cin : Cout;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
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

struct PlanB {
    int cfg, grid, n_mtiles, nit;
    size_t lds_bytes;
    ConvKB k;
};

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
            static const int pref[kNumCfgB] = {0, 2, 1, 8, 5, 6, 7, 4, 3};
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

template <int KS, int MSUB, int NSUB, int WM, int WN, bool X3>
static void launch_nit_b(const PlanB &pl, void *stream) {
    if (pl.nit <= 4)
        DS_LAUNCH((conv_mfma_bf16_kernel<KS, MSUB, NSUB, WM, WN, X3, 4, true>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    else if (pl.nit <= 8)      // register-prefetch the next chunk where the accumulators leave room (MSUB <= 4)
        DS_LAUNCH((conv_mfma_bf16_kernel<KS, MSUB, NSUB, WM, WN, X3, 8, (MSUB <= 4)>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
    else
        DS_LAUNCH((conv_mfma_bf16_kernel<KS, MSUB, NSUB, WM, WN, X3, 16, false>), pl.grid, 256, pl.lds_bytes, stream, pl.k);
}

template <int KS, int MSUB, int WM, int WN>
static void launch_big_b(const PlanB &pl, void *stream) {
    constexpr int NTHR = WM * WN * 64;
    if (pl.nit <= 8)
        DS_LAUNCH_BIG_LDS((conv_mfma_bf16_kernel<KS, MSUB, 2, WM, WN, true, 8, true>), pl.grid, NTHR, pl.lds_bytes, stream, pl.k);
    else if (pl.nit <= 16)
        DS_LAUNCH_BIG_LDS((conv_mfma_bf16_kernel<KS, MSUB, 2, WM, WN, true, 16, true>), pl.grid, NTHR, pl.lds_bytes, stream, pl.k);
    else
        DS_LAUNCH_BIG_LDS((conv_mfma_bf16_kernel<KS, MSUB, 2, WM, WN, true, 32, false>), pl.grid, NTHR, pl.lds_bytes, stream, pl.k);
}

template <int KS, bool X3>
static void launch_b(const PlanB &pl, void *stream) {
    if (pl.cfg == 0) launch_nit_b<KS, 2, 1, 2, 2, X3>(pl, stream);
    else if (pl.cfg == 1) launch_nit_b<KS, 5, 1, 1, 4, X3>(pl, stream);
    else if (pl.cfg == 2) launch_nit_b<KS, 4, 1, 2, 2, X3>(pl, stream);
    else if constexpr (X3) {                    // 160x64 register tiles, opt-in LDS sizes
        if (pl.cfg == 3) launch_big_b<KS, 5, 1, 2>(pl, stream);
        else if (pl.cfg == 4) launch_big_b<KS, 5, 1, 4>(pl, stream);
        else if (pl.cfg == 5) launch_big_b<KS, 5, 2, 2>(pl, stream);
        else if (pl.cfg == 6) launch_big_b<KS, 5, 2, 1>(pl, stream);
        else if (pl.cfg == 7) launch_big_b<KS, 4, 1, 2>(pl, stream);
        else launch_big_b<KS, 4, 1, 4>(pl, stream);
    }
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
        launch_b<3, true>(pl, stream);
        rc = ds_last_launch_error();
        if (rc) return rc;
    }
    return DS_OK;
}

