// conv_mfma_f16_kernel.h -- implicit-GEMM convolution on the gfx950 fp16 matrix cores
// (v_mfma_f32_32x32x16_f16, f32 accumulate) with fp16 activations in HBM: the throughput kernel of the eval
// forward (reference model.py:69,73,192,197,202 + the fused BatchNorm-affine / residual / clipped-ReLU
// epilogue, model.py:70-80,188-205).  One MFMA per product; measured 3.7e-4 from the reference on the
// embedding (contract: 1e-3).  Included by conv_mfma_f16_k*.hip (one translation unit per kernel size x
// buffering mode so the instantiations compile in parallel); planner and C ABI: conv_mfma_f16.hip.
//
// Differences from the split-operand bf16 kernel (conv_mfma_bf16_kernel.h), all following from having a third
// of the matrix work per byte moved:
//   * activations are fp16 channels-last in HBM: staging is a 16-byte copy (no conversion), HBM traffic halves;
//   * 32 input channels per chunk (two MFMA k-steps per tap) -> half as many chunk boundaries;
//   * the pixel tile is double-buffered in LDS where it fits (DB): the next chunk's pixels are written into
//     the other buffer between the MFMAs of this chunk's last taps, leaving ONE barrier per chunk;
//   * fragment reads, filter-ring refills and the staging traffic are dealt out one per MFMA.
#pragma once
#include <ds_device.h>
#include <type_traits>
#include "ds_common.h"

// CK input channels per chunk = CK/16 k-steps of the 32x32x16 MFMA per tap (32, or 16 where two 32-channel tiles
// do not fit the LDS); a staged pixel record is CK halfs + 16 B pad = 80 / 48 bytes: 16 consecutive records walk
// all 64 banks with one ds_read_b128 each.
constexpr int ds_f16_record_bytes(int ck) { return ck * 2 + 16; }

// filter ring depth in (k-step, tap) units for the 3x3 / 5x5 kernels with 32-channel chunks (must divide 18 / 50)
#ifndef DS_F16_RING_K3
#define DS_F16_RING_K3 6
#endif
#ifndef DS_F16_RING_K5
#define DS_F16_RING_K5 5
#endif

struct ConvKH {
    const _Float16 *x;              // [B, H, W, Cin] fp16 channels-last
    const _Float16 *w;              // packed [Cin/16][tap][Cout][16]
    void *y;                        // [B, Ho, Wo, Cout] fp16 (f32 with DS_EPI_OUT_F32)
    const float *scale, *shift;
    const _Float16 *res;            // residual, fp16, laid out like y
    int H, W, Cin;
    int Ho, Wo, Cout;
    int IS;
    int dh_min, dw_min;
    int rows_in, cols_in, seg_pix;
    int pitch, half;                // LDS records per tile row; first odd-column slot (stride-2 de-interleave)
    int row_bytes, seg_bytes;       // LDS bytes from one tile row / one segment to the next: pitch (rows_in x pitch) records
                                    // plus the padding that spreads the 16 pixels of a fragment read's service group
                                    // over all banks when they span several rows / images (planner: frag_read_cost)
    int RT, NI, segs_per_img, n_segs;
    int n_ntiles;
    int flags;
    unsigned y_bytes, res_bytes;    // sizes for the buffer descriptors
    // split-K (small launches: serving latency): workgroup blockIdx = split * tiles + tile contracts the input-channel
    // chunks [split * chunks_per_split, ...) and stores its raw f32 accumulators into partial + split * partial_elems
    // Layouts.  Channels-last: pixel stride Cin, the next chunk CK elements further.  Channel-plane-major
    // ([C/16][pixels][16], DS_CONV_IN_PLANES16 / DS_EPI_OUT_PLANES16): pixel stride 16, the next (16-channel) chunk one
    // plane further -- a chunk then reads WHOLE 128-byte lines instead of a quarter of every pixel record.
    int x_pix_stride, x_chunk_stride;
    unsigned y_plane_stride;        // 0: y is channels-last; else elements per 16-channel plane of y
    int tiles, n_splits, chunks_per_split;
    float *partial;
    unsigned partial_elems;
    unsigned *sched;                // persistent kernel: tile-scheduling slot (ds_device.h)
    int sched_queues;               // ... 8: one tile queue per XCD (workgroup b draws tiles 8 j + b % 8); 1: one queue
    int sched_lds;                  // ... and the byte offset of the LDS word tile indices are passed through
#ifdef DS_F16_PROBE                 // tools/f16_phase_probe.py builds: s_memtime stamps of the phases of each workgroup
    long long *probe;
#endif
};

#ifdef DS_F16_PROBE
#define DS_F16_STAMP(i) do { if (p.probe && threadIdx.x == 0) p.probe[(size_t)blockIdx.x * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define DS_F16_STAMP(i) ((void)0)
#endif

struct PlanH {
    int cfg, grid, n_mtiles, nit, db, ck;
    int lin = 0;                    // persistent kernel: staging offsets derived from the first item's (row blocks)
    size_t lds_bytes;
    ConvKH k;
};

// one entry point per translation unit (kernel size x buffering)
void ds_f16_launch_k3db(const PlanH &pl, void *stream);
void ds_f16_launch_k3sb(const PlanH &pl, void *stream);
void ds_f16_launch_k5db(const PlanH &pl, void *stream);
void ds_f16_launch_k5sb(const PlanH &pl, void *stream);
void ds_f16_launch_k5c16(const PlanH &pl, void *stream);    // 16-channel chunks, double-buffered

#ifdef DS_F16_KERNEL_TU
namespace {

// NIT: 16-byte staging items per thread and chunk (compile time: all loads of a chunk are in flight together).
// DB:  two pixel-tile buffers in LDS; requires the register prefetch (NIT <= 16).
template <int KS, int MSUB, int NSUB, int WM, int WN, int NIT, bool DB, int CKH = 32>
__global__ void __launch_bounds__(WM * WN * 64) DS_ONE_WAVE_PER_SIMD conv_mfma_f16_kernel(const ConvKH p) {
    constexpr int NTHR = WM * WN * 64;
    constexpr int MT = MSUB * WM * 32;
    constexpr int NTILE = NSUB * WN * 32;
    constexpr int NT = KS * KS;
    constexpr int KPT = CKH / 16;                   // k-steps per tap
    constexpr int IPP = CKH / 8;                    // 16-byte staging items per pixel
    constexpr int PSH = ds_f16_record_bytes(CKH);
    constexpr int NU = KPT * NT;                    // (k-step, tap) units per chunk
    constexpr int RU = (KS == 3) ? (KPT == 2 ? DS_F16_RING_K3 : 9) : (KPT == 2 ? DS_F16_RING_K5 : 5);   // filter ring, in units; NU % RU == 0
    constexpr int NMF = MSUB * NSUB;                // MFMAs per unit
    constexpr bool PREF = DB || NIT <= 16;          // next chunk's pixels ride in registers through the taps
    static_assert(NU % RU == 0, "ring slots must be chunk-invariant");
    static_assert(!DB || NIT <= 16, "double buffering needs the register prefetch");

    char *lds = (char *)ds_dynamic_lds();
    DS_F16_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int split = (int)blockIdx.x / p.tiles, tile = (int)blockIdx.x - split * p.tiles;
    const int tile_n = tile % p.n_ntiles;
    const int tile_m = tile / p.n_ntiles;
    const int seg0 = tile_m * p.NI;
    const int pix_per_seg = p.RT * p.Wo;
    // the pixel-tile region doubles as the epilogue's transposition buffers (two of 32 x (NSUB*32+4) floats per wave)
    constexpr int EPI_BYTES = 2 * WM * WN * 32 * (NSUB * 32 + 4) * 4;
    const int tile_bytes = p.NI * p.seg_bytes;
    const int tiles_bytes = (DB ? 2 : 1) * tile_bytes;
    const int stage_bytes = tiles_bytes > EPI_BYTES ? tiles_bytes : EPI_BYTES;
    int *out_off = (int *)(lds + stage_bytes);                 // [MT]
    int *seg_lo = out_off + MT;                                // [NI] first in-image row of each segment's tile
    int *seg_cnt = seg_lo + p.NI;                              // [NI] number of in-image rows

    // the first filter fragments are requested before anything else: their latency hides behind the tables
    const int c0 = split * p.chunks_per_split;                  // this workgroup's chunk range [c0, c0 + n_chunks)
    const int n_chunks = (p.Cin / CKH - c0) < p.chunks_per_split ? (p.Cin / CKH - c0) : p.chunks_per_split;
    const int n_base = tile_n * NTILE + wn * NSUB * 32;
    const size_t lane_w = ((size_t)(n_base + l31) * 16 + 8 * lhi);      // in halfs
    const size_t w_kc_stride = (size_t)NT * p.Cout * 16;                // one 16-channel slab: [tap][Cout][16]
    const size_t w_tap_stride = (size_t)p.Cout * 16;
    // unit u of a chunk = (k-step u / NT, tap u % NT): filter slab KPT*chunk + u / NT, tap u % NT.  K-step-major, so
    // that a pixel's products are accumulated in the same order with 16- and 32-channel chunks (results do not
    // depend on which the planner picks for a batch size).
    auto w_unit = [&](int chunk, int u) {
        return p.w + lane_w + (size_t)(KPT * chunk + (u / NT)) * w_kc_stride + (size_t)(u % NT) * w_tap_stride;
    };

    f16x8 bq[RU][NSUB];
#pragma unroll
    for (int d = 0; d < RU; ++d)
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) bq[d][ns] = *(const f16x8 *)(w_unit(c0, d) + (size_t)ns * 32 * 16);

    const float rcp_pps = 1.0f / (float)pix_per_seg, rcp_wc = 1.0f / (float)p.Wo, rcp_w = 1.0f / (float)p.W,
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

    // ---- staging descriptors (chunk-invariant) of this thread's items: 8 channels (16 B) of one in-image
    // pixel each.  Item idx -> (quarter q, column c, valid row) in segment order; slots past the last item
    // load x[0..7] and drop it into the unused pad bytes of pixel record 0: the chunk loop has no branches.
    int g_off[NIT], l_off[NIT];
    DS_F16_STAMP(5);
    __syncthreads();                            // seg_lo / seg_cnt are complete
    {
        const int q = tid % IPP;
        const int dvr = ds_div_small(NTHR / IPP, p.W, rcp_w), dc = NTHR / IPP - dvr * p.W;
        int vr = ds_div_small(tid / IPP, p.W, rcp_w);
        int c = (tid / IPP) - vr * p.W;
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
        // Tiles whose segments all have the same in-image row window -- a single segment, or one whole image per
        // segment (every bench layer) -- locate an item with one reciprocal division; only mixed windows walk the table.
        const bool uniform = p.NI == 1 || p.segs_per_img == 1;
        if (uniform) {
            const int h0 = p.IS * (seg0 - ds_div_small(seg0, p.segs_per_img, rcp_spi) * p.segs_per_img) * p.RT + p.dh_min;
            const int lo_u = h0 < 0 ? -h0 : 0;
            const int hi_u = p.H - h0 < p.rows_in ? p.H - h0 : p.rows_in;
            const int cnt_u = hi_u > lo_u ? hi_u - lo_u : 1;
            const float rcp_cnt = 1.0f / (float)cnt_u;
            const int b0 = ds_div_small(seg0, p.segs_per_img, rcp_spi);
            const int img_row0 = b0 * p.H + h0;                                      // tile row 0 of segment 0
            const int live_segs = p.n_segs - seg0 < p.NI ? p.n_segs - seg0 : p.NI;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                int sg;
                if (p.NI == 1) sg = vr >= cnt_u ? 1 : 0;       // (uniform branch: a single segment needs no division)
                else sg = ds_div_small(vr, cnt_u, rcp_cnt);
                const int rr = lo_u + vr - sg * cnt_u;
                const int cc = c - p.dw_min;
                const int pc = (p.IS == 2) ? ((cc & 1) ? p.half + (cc >> 1) : (cc >> 1)) : cc;
                const bool ok = sg < live_segs;
                g_off[it] = ok ? ((img_row0 + sg * p.H + rr) * p.W + c) * p.x_pix_stride + q * 8 : 0;
                l_off[it] = ok ? sg * p.seg_bytes + rr * p.row_bytes + pc * PSH + q * 16 : CKH * 2;
                c += dc;
                vr += dvr;
                if (c >= p.W) {
                    c -= p.W;
                    ++vr;
                }
            }
        } else {
            next_seg();
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                while (seg < p.NI && vr >= row0 + cnt) next_seg();
                g_off[it] = 0;
                l_off[it] = CKH * 2;                // the pad bytes of pixel record 0
                if (seg < p.NI) {
                    const int rr = lo + vr - row0;
                    // stride-2 layers keep even tile columns in slots [0, half) and odd ones in [half, cols_in),
                    // so that the 32 lanes of a fragment read (stride-2 columns) touch CONSECUTIVE records
                    const int cc = c - p.dw_min;
                    const int pc = (p.IS == 2) ? ((cc & 1) ? p.half + (cc >> 1) : (cc >> 1)) : cc;
                    g_off[it] = ((img_row + rr) * p.W + c) * p.x_pix_stride + q * 8;
                    l_off[it] = seg * p.seg_bytes + rr * p.row_bytes + pc * PSH + q * 16;
                }
                c += dc;
                vr += dvr;
                if (c >= p.W) {
                    c -= p.W;
                    ++vr;
                }
            }
        }
    }
    DS_F16_STAMP(6);
    f32x4 st[NIT];                              // 8 halfs each, moved as 16 opaque bytes
    if constexpr (PREF) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) st[it] = *(const f32x4 *)(p.x + g_off[it] + (size_t)c0 * p.x_chunk_stride);
    }
    // ---- everything below overlaps the first chunk's loads ----
    // Only in-image pixels are ever staged: the zero halo (and the row padding) is written once, here.
    for (int i = tid; i < tiles_bytes / 16; i += NTHR) *(f32x4 *)(lds + 16 * i) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // Output pixel of tile row m.  A tile that is one block of rows of one image, or a run of whole images (every
    // bench layer), covers CONSECUTIVE pixels of y: pixel lin_base + m for m < lin_valid, no table.
    const bool linear = p.NI == 1 || p.RT == p.Ho;
    int lin_base = 0, lin_valid = 0;
    if (linear) {
        const int b = ds_div_small(seg0, p.segs_per_img, rcp_spi);
        const int r0 = (seg0 - b * p.segs_per_img) * p.RT;
        lin_base = (b * p.Ho + r0) * p.Wo;
        const int live = p.n_segs - seg0 < p.NI ? p.n_segs - seg0 : p.NI;
        lin_valid = p.NI == 1 ? (p.Ho - r0 < p.RT ? p.Ho - r0 : p.RT) * p.Wo : live * pix_per_seg;
    } else {
        for (int m = tid; m < MT; m += NTHR) {
            const int seg = ds_div_small(m, pix_per_seg, rcp_pps);
            const int rem = m - seg * pix_per_seg;
            const int r = ds_div_small(rem, p.Wo, rcp_wc), c = rem - r * p.Wo;
            const int gseg = seg0 + seg;
            int off = -1;
            if (seg < p.NI && gseg < p.n_segs) {
                const int b = ds_div_small(gseg, p.segs_per_img, rcp_spi);
                const int rr = (gseg - b * p.segs_per_img) * p.RT + r;
                if (rr < p.Ho) off = (b * p.Ho + rr) * p.Wo + c;              // output pixel index
            }
            out_off[m] = off;
        }
    }
    DS_F16_STAMP(7);
    // Which pixel of its 32-pixel sub-tile a lane owns is free (the epilogue un-permutes): it is chosen so
    // that the two 16-lane SERVICE GROUPS of a ds_read_b128 -- lanes {0-3,12-15,20-27} and {4-11,16-19,
    // 28-31} -- each read 16 CONSECUTIVE pixels, i.e. consecutive 80-byte records that walk all 64 banks.
    const int lpix = (l31 < 4 || l31 >= 28) ? l31
                   : (l31 < 12) ? l31 + 12 : (l31 < 16) ? l31 - 8 : (l31 < 20) ? l31 + 8 : l31 - 12;
    int a_off[MSUB];                                           // byte offset of this lane's fragment
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
        // (a pixel past the tile's last segment reads what the first pixel of its 16-pixel service group reads -- the
        // same address is served in the same LDS cycle -- or record 0 if that one is past the end as well)
        int m = (wm * MSUB + ms) * 32 + lpix;
        if (m >= p.NI * pix_per_seg) m &= ~15;
        const int seg = ds_div_small(m, pix_per_seg, rcp_pps);
        const int rem = m - seg * pix_per_seg;
        const int r = ds_div_small(rem, p.Wo, rcp_wc), c = rem - r * p.Wo;
        a_off[ms] = ((seg < p.NI) ? seg * p.seg_bytes + (p.IS * r) * p.row_bytes + c * PSH : 0) + 16 * lhi;
    }
    auto tap_off = [&](int tt) {
        const int kw = tt % KS;
        return (tt / KS) * p.row_bytes + (p.IS == 2 ? (kw & 1) * p.half + (kw >> 1) : kw) * PSH;
    };

    // ---- one chunk of 32 input channels: NU units of NMF MFMAs, one side operation after each MFMA ----
    //   odd slots:  the NEXT unit's pixel fragments (LDS -> registers, double-buffered per unit)
    //   even slots: the filter-ring refills (L2 -> registers, RU-1 units ahead), then the staging traffic:
    //               loads of the next chunk's pixels in the first units, their LDS writes in the last ones (DB)
    constexpr int SPU = (NMF + 1) / 2 - NSUB;              // staging slots per unit
    static_assert(SPU >= 1 && NSUB >= 2, "tile too small for the interleaved schedule (one odd slot per fragment read)");
    constexpr int UL = (NIT + SPU - 1) / SPU;              // units that issue loads / that issue LDS writes
    static_assert(!PREF || 2 * UL <= NU, "not enough units for the staging traffic");
    auto run_chunk = [&](auto last_tag, int chunk, const char *buf, char *obuf) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
#ifndef DS_F16_AFRAG_DEPTH
#define DS_F16_AFRAG_DEPTH 1                    // pixel fragments are read this many units ahead of their MFMAs (2: measured 2 % slower)
#endif
        constexpr int AD = DS_F16_AFRAG_DEPTH, AB = AD + 1;
        f16x8 a[AB][MSUB];
#pragma unroll
        for (int ms = 0; ms < MSUB; ++ms) DS_OPAQUE_VGPR(a_off[ms]);     // keep the NU x MSUB fragment addresses out of registers
#pragma unroll
        for (int d = 0; d < AD; ++d)
#pragma unroll
            for (int ms = 0; ms < MSUB; ++ms)
                a[d][ms] = *(const f16x8 *)(buf + a_off[ms] + tap_off(d % NT) + 32 * (d / NT));
        const _Float16 *xn = p.x + (size_t)(chunk + 1) * p.x_chunk_stride;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int cur = u % AB, nxt = (u + AD) % AB, slot = u % RU;
            const bool more = u + AD < NU;
            const char *nfrag = buf + tap_off((u + AD) % NT) + 32 * ((u + AD) / NT);
            // the slot the previous unit consumed is refilled with the unit RU - 1 ahead of this one
            const int ur = u - 1 + RU;                          // may run into the next chunk
            const bool refill = !(LAST && ur >= NU);            // (unit 0 of chunk 0 reloads what the prologue loaded)
            const _Float16 *rw = w_unit(ur >= NU ? chunk + 1 : chunk, ur >= NU ? ur - NU : ur);
            const int rslot = (u + RU - 1) % RU;
#pragma unroll
            for (int q = 0; q < NMF; ++q) {
                const int ms = q / NSUB, ns = q % NSUB;
                acc[ms][ns] = ds_mfma_32x32x16_f16(bq[slot][ns], a[cur][ms], acc[ms][ns]);
                if (q & 1) {
                    const int lm = q >> 1;
                    if (lm < MSUB && more) a[nxt][lm] = *(const f16x8 *)(nfrag + a_off[lm]);
                } else {
                    const int e = q >> 1;
                    if (e < NSUB) {
                        if (refill) bq[rslot][e] = *(const f16x8 *)(rw + (size_t)e * 32 * 16);
                    } else if constexpr (PREF && !LAST) {
                        const int s = e - NSUB;
                        if (u < UL) {                           // next chunk's pixels -> registers
                            const int it = u * SPU + s;
                            if (it < NIT) st[it] = *(const f32x4 *)(xn + g_off[it]);
                        } else if (DB && u >= NU - UL) {        // ... -> the other LDS buffer
                            const int it = (u - (NU - UL)) * SPU + s;
                            if (it < NIT) *(f32x4 *)(obuf + l_off[it]) = st[it];
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    DS_F16_STAMP(1);
    if constexpr (DB) {
        __syncthreads();                        // the zero fill is complete
#pragma unroll
        for (int it = 0; it < NIT; ++it) *(f32x4 *)(lds + l_off[it]) = st[it];
        __syncthreads();
        for (int i = 0; i + 1 < n_chunks; ++i) {
            char *b0 = lds + (i & 1) * tile_bytes, *b1 = lds + ((i & 1) ^ 1) * tile_bytes;
            run_chunk(std::false_type{}, c0 + i, b0, b1);
            ds_lds_barrier();                   // LDS only: the filter-ring loads in flight stay in flight
        }
        run_chunk(std::true_type{}, c0 + n_chunks - 1, lds + ((n_chunks - 1) & 1) * tile_bytes, lds);
    } else {
        auto stage_chunk = [&](int chunk) __attribute__((always_inline)) {
            ds_lds_barrier();                   // previous chunk's fragment reads (chunk 0: the zero fill) are done
            if constexpr (!PREF) {
#pragma unroll
                for (int it = 0; it < NIT; ++it) st[it] = *(const f32x4 *)(p.x + g_off[it] + (size_t)chunk * p.x_chunk_stride);
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) *(f32x4 *)(lds + l_off[it]) = st[it];
            ds_lds_barrier();
        };
        for (int i = 0; i + 1 < n_chunks; ++i) {
            stage_chunk(c0 + i);
            run_chunk(std::false_type{}, c0 + i, lds, lds);
        }
        stage_chunk(c0 + n_chunks - 1);
        run_chunk(std::true_type{}, c0 + n_chunks - 1, lds, lds);
    }

    // ---- epilogue ----
    // The filters were the A operand of every MFMA, so the accumulators hold the TRANSPOSED product: a lane
    // owns one output pixel (lpix of the 32-pixel sub-tile) and, per register quad g, four consecutive output
    // channels 8g + 4*lhi .. +3.  Each 32-pixel sub-tile is turned around through a wave-private LDS buffer
    // (the pixel tile's space, free now) so that residual loads and stores move whole pixel rows: the
    // NSUB*32 channels of a pixel are contiguous across NSUB*4 lanes, 8 channels = 16 bytes of fp16 per lane.
    constexpr int TP = NSUB * 32 + 4;           // buffer row pitch in floats (conflict-free 16-byte writes)
    constexpr int LPP = NSUB * 4;               // lanes per pixel row
    constexpr int PPI = 64 / LPP;               // pixel rows per instruction
    constexpr int NRI = 32 / PPI;               // instructions per sub-tile
    // a split-K workgroup stores raw f32 accumulators; the reduction kernel applies the epilogue
    const int flags = p.n_splits > 1 ? DS_EPI_OUT_F32 : p.flags;
    DS_F16_STAMP(2);
    __syncthreads();                            // every wave is done reading the pixel tile
    DS_F16_STAMP(3);
    float *tb = (float *)lds + wave * (2 * 32 * TP);
    const int my_c = (lane % LPP) * 8, my_p = lane / LPP;
    const int col = n_base + my_c;
    f32x4 sc[2] = {{1.0f, 1.0f, 1.0f, 1.0f}, {1.0f, 1.0f, 1.0f, 1.0f}};
    f32x4 sh[2] = {{0.0f, 0.0f, 0.0f, 0.0f}, {0.0f, 0.0f, 0.0f, 0.0f}};
    if (flags & DS_EPI_AFFINE) {
        sc[0] = *(const f32x4 *)(p.scale + col);
        sc[1] = *(const f32x4 *)(p.scale + col + 4);
        sh[0] = *(const f32x4 *)(p.shift + col);
        sh[1] = *(const f32x4 *)(p.shift + col + 4);
    }
    // Rows of a ragged tile get an out-of-range buffer offset (the store is dropped, the load returns
    // zeros) and a layer without residual reads "out of range" too: no branch around any memory instruction.
    const bool out32 = (flags & DS_EPI_OUT_F32) != 0;
    const float clip_lo = (flags & DS_EPI_CLIP) ? 0.0f : -__builtin_inff();
    const float clip_hi = (flags & DS_EPI_CLIP) ? 20.0f : __builtin_inff();
    const ds_buffer ybuf = p.n_splits > 1 ? ds_make_buffer(p.partial + (size_t)split * p.partial_elems, p.partial_elems * 4u)
                                          : ds_make_buffer(p.y, p.y_bytes);
    const ds_buffer rbuf = ds_make_buffer((flags & DS_EPI_RESIDUAL) ? (const void *)p.res : (const void *)p.y,
                                          (flags & DS_EPI_RESIDUAL) ? p.res_bytes : 0u);
    // Every row offset and every residual row of the wave's MSUB sub-tiles is requested BEFORE the first store:
    // gfx9 retires loads and stores through one in-order counter, so a residual load issued between two stores
    // could only be waited for together with the store ahead of it (a full write acknowledgement per sub-tile).
    unsigned voff[MSUB][NRI];                   // element offset of (pixel row, first channel) or OOB
    f32x4 resv[MSUB][NRI];
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms)
#pragma unroll
        for (int k = 0; k < NRI; ++k) {
            const int m = (wm * MSUB + ms) * 32 + k * PPI + my_p;
            int off;
            if (linear) off = m < lin_valid ? lin_base + m : -1;
            else off = out_off[m];                                 // output pixel index, or -1
            // y: channels-last, or 16-channel planes (the lane's 8 channels lie inside one plane); the residual is
            // always channels-last
            const unsigned cl = (unsigned)(off * p.Cout + col);
            voff[ms][k] = off < 0 ? DS_BUFFER_OOB
                          : p.y_plane_stride ? (unsigned)(col >> 4) * p.y_plane_stride + (unsigned)off * 16u + (unsigned)(col & 15) : cl;
            resv[ms][k] = ds_buffer_load_f32x4(rbuf, off >= 0 ? cl * 2u : DS_BUFFER_OOB);
        }
    auto put_tile = [&](int ms) {               // accumulators of sub-tile ms -> this wave's buffer ms & 1
        float *dst = tb + (ms & 1) * (32 * TP);
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[ms][ns][4 * g + j];
                *(f32x4 *)(dst + lpix * TP + ns * 32 + 8 * g + 4 * lhi) = v;
            }
    };
    put_tile(0);
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
        const int cb = ms & 1;
        ds_wave_sync();                         // sub-tile ms is in its buffer (LDS runs a wave's operations in order)
        const float *src = tb + cb * (32 * TP);
        f32x4 tv[NRI][2];
#pragma unroll
        for (int k = 0; k < NRI; ++k)
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) tv[k][hq] = *(const f32x4 *)(src + (k * PPI + my_p) * TP + my_c + 4 * hq);
        if (ms + 1 < MSUB) put_tile(ms + 1);    // the next sub-tile's turn-around travels while this one is finished
#pragma unroll
        for (int k = 0; k < NRI; ++k) {
            const f16x8 r8 = __builtin_bit_cast(f16x8, resv[ms][k]);
            f32x4 o[2];
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float t = tv[k][hq][j] * sc[hq][j] + sh[hq][j];
                    t += (float)r8[4 * hq + j];
                    o[hq][j] = fminf(fmaxf(t, clip_lo), clip_hi);   // (-inf, +inf) without DS_EPI_CLIP: one v_med3
                }
            }
            const unsigned vo = voff[ms][k];
            if (out32) {
                const unsigned b = vo != DS_BUFFER_OOB ? vo * 4u : DS_BUFFER_OOB;
                ds_buffer_store_out_f32x4(ybuf, b, o[0]);
                ds_buffer_store_out_f32x4(ybuf, b != DS_BUFFER_OOB ? b + 16u : DS_BUFFER_OOB, o[1]);
            } else {
                f16x8 h;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h[j] = (_Float16)o[0][j];
                    h[4 + j] = (_Float16)o[1][j];
                }
                ds_buffer_store_out_f32x4(ybuf, vo != DS_BUFFER_OOB ? vo * 2u : DS_BUFFER_OOB, __builtin_bit_cast(f32x4, h));
            }
        }
    }
    DS_F16_STAMP(4);
}

template <int KS, int MSUB, int NSUB, int WM, int WN, bool DB, int CK>
static void launch_nit_h(const PlanH &pl, void *stream) {
    constexpr int NTHR = WM * WN * 64;
    if (pl.nit <= 8)
        DS_LAUNCH_BIG_LDS((conv_mfma_f16_kernel<KS, MSUB, NSUB, WM, WN, 8, DB, CK>), pl.grid, NTHR, pl.lds_bytes, stream, pl.k);
    else if (pl.nit <= 16)
        DS_LAUNCH_BIG_LDS((conv_mfma_f16_kernel<KS, MSUB, NSUB, WM, WN, 16, DB, CK>), pl.grid, NTHR, pl.lds_bytes, stream, pl.k);
    else if constexpr (!DB)
        DS_LAUNCH_BIG_LDS((conv_mfma_f16_kernel<KS, MSUB, NSUB, WM, WN, 32, false, CK>), pl.grid, NTHR, pl.lds_bytes, stream, pl.k);
}

template <int KS, bool DB, int CK = 32>
static void launch_h(const PlanH &pl, void *stream) {
    if (pl.cfg == 0) launch_nit_h<KS, 5, 2, 1, 2, DB, CK>(pl, stream);          // 160x128, two waves
    else if (pl.cfg == 1) launch_nit_h<KS, 5, 2, 1, 4, DB, CK>(pl, stream);     // 160x256, four waves
    else if (pl.cfg == 2) launch_nit_h<KS, 5, 2, 2, 2, DB, CK>(pl, stream);     // 320x128
    else if (pl.cfg == 3) launch_nit_h<KS, 5, 2, 2, 1, DB, CK>(pl, stream);     // 320x64, two waves
    else if (pl.cfg == 4) launch_nit_h<KS, 4, 2, 1, 2, DB, CK>(pl, stream);     // 128x128, two waves
    else if (pl.cfg == 5) launch_nit_h<KS, 4, 2, 1, 4, DB, CK>(pl, stream);     // 128x256
    else launch_nit_h<KS, 5, 2, 4, 1, DB, CK>(pl, stream);                      // 640x64, four waves
}

}  // namespace
#endif
