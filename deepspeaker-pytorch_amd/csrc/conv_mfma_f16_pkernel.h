// conv_mfma_f16_pkernel.h -- PERSISTENT form of the fp16 implicit-GEMM convolution (conv_mfma_f16_kernel.h) for the
// geometry every full-size layer of the eval forward has: a double-buffered pixel tile whose M rows are either one
// block of rows of one image (NI == 1) or a run of whole images (segs_per_img == 1), no split-K.  Same products, same
// accumulation order, same epilogue arithmetic: bit-identical to conv_mfma_f16_kernel (asserted by the tests); the
// planner (conv_mfma_f16.hip) falls back to that kernel for everything else.
//
// What persistence buys (reference model.py:69,73,192,197,202 run through ds_conv_fwd_f16; measured on the
// one-tile-per-workgroup kernel with tools/f16_phase_probe.py: 7-21 k clocks of prologue and 8-10 k of epilogue next
// to MFMA streams of 16-124 k):
//   * the grid is what the chip holds at once; a workgroup walks tiles.  The staging / fragment descriptors (a dozen
//     reciprocal divisions per thread) are tile-invariant: what changes per tile is a base address and the in-image
//     window, both folded into a per-tile BUFFER DESCRIPTOR -- rows of the tile that lie outside the image (or belong
//     to an image past the end of the batch) are out of the buffer's range, read as zeros and are staged as such;
//   * only the halo columns (and, for whole-image segments, the halo rows) are zeroed per tile, not the whole tile;
//   * the filter ring runs through from a tile's last chunk into the next tile's first; the next tile's first input
//     chunk is requested at the head of the epilogue, the residual rows in the last units of the last chunk;
//   * accumulators start from a literal zero; the epilogue picks the output type once per tile and uses packed f32
//     arithmetic.
#pragma once
#include "conv_mfma_f16_kernel.h"

void ds_f16_launch_pk3(const PlanH &pl, void *stream);
void ds_f16_launch_pk5(const PlanH &pl, void *stream);
void ds_f16_launch_pk5c16(const PlanH &pl, void *stream);

#ifdef DS_F16_PKERNEL_TU
namespace {

#ifdef DS_F16_PROBE     // tools/pkernel_phase_probe.py: stamp i of a workgroup's tile number tile_no (tiles 0..3 are recorded)
#define DS_PK_STAMP(i) do { if (p.probe && threadIdx.x == 0 && tile_no < 4) p.probe[((size_t)blockIdx.x * 4 + tile_no) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define DS_PK_STAMP(i) ((void)0)
#endif

template <int KS, int MSUB, int NSUB, int WM, int WN, int NIT, int CKH, bool LIN>
__global__ void __launch_bounds__(WM * WN * 64) DS_ONE_WAVE_PER_SIMD conv_mfma_f16_pkernel(const ConvKH p) {
    constexpr int NTHR = WM * WN * 64;
    constexpr int NTILE = NSUB * WN * 32;
    constexpr int NT = KS * KS;
    constexpr int KPT = CKH / 16;                   // k-steps per tap
    constexpr int IPP = CKH / 8;                    // 16-byte staging items per pixel
    constexpr int PSH = ds_f16_record_bytes(CKH);
    constexpr int PPR = PSH / 16;                   // 16-byte pieces per record
    constexpr int NU = KPT * NT;                    // (k-step, tap) units per chunk
    // filter ring, in units.  NSUB = 4 (a 128-channel-wide register tile: every pixel fragment read from LDS feeds FOUR
    // MFMAs instead of two, tools/mfma_lds_ratio.hip) has 16 MFMAs = 512 clocks per unit and four fragments per ring
    // slot: 3 units ahead cover the L2 latency at half the registers of the 6-deep ring of the 64-wide tiles
    constexpr int RU = NSUB == 4 ? (KS == 3 ? 3 : 5)
                                 : (KS == 3) ? (KPT == 2 ? DS_F16_RING_K3 : 9) : (KPT == 2 ? DS_F16_RING_K5 : 5);
    constexpr int NMF = MSUB * NSUB;                // MFMAs per unit
    constexpr int SPU = (NMF + 1) / 2 - NSUB;       // staging slots per unit
    constexpr int UL = (NIT + SPU - 1) / SPU;       // units that issue loads / that issue LDS writes
    // The epilogue works on 64 channels at a time whatever NSUB is (ENS = 2 sub-tiles of 32 channels: the turn-around
    // buffers, the lane -> (pixel, 8 channels) assignment and the register use of a step are those of the 64-wide tiles):
    // a wave's tile is MSUB x NH steps, step e = (ms, nh) = (e / NH, e % NH), channels n_base + 64 nh ...
    constexpr int ENS = 2, NH = NSUB / ENS, NE = MSUB * NH;
    constexpr int TP = ENS * 32 + 4, LPP = ENS * 4, PPI = 64 / LPP, NRI = 32 / PPI;
    // residual rows: those of the first RPRE steps are requested in the last units of the tile's last chunk, the
    // others two steps ahead inside the epilogue (all NE * NRI of them held through the stream's tail would cost
    // up to 80 registers)
    constexpr int RPRE = 2;
    constexpr int NRES = RPRE * NRI, ULR = (NRES + SPU - 1) / SPU;
    // The 64-wide tiles run the filter ring through from a tile's last chunk into the next tile's first (RU - 1 units
    // of fragments stay live across the epilogue).  With 128-wide tiles those are 32 registers the epilogue does not
    // have (the compiler spilled them to scratch right behind their loads: an s_waitcnt vmcnt(0) per fragment inside
    // the stream); the ring is refilled at the top of each tile instead -- a tile of that shape is 2304 MFMAs long.
    constexpr bool RING_THROUGH = NSUB < 4;
    static_assert(NU % RU == 0, "ring slots must be chunk-invariant");
    static_assert(NSUB % ENS == 0, "the epilogue walks the tile 64 channels at a time");
    static_assert(SPU >= 1 && NSUB >= 2 && NIT <= 16, "tile too small for the interleaved schedule");
    static_assert(2 * UL <= NU && ULR + 2 <= NU, "not enough units for the staging traffic");
    constexpr unsigned OOB = 0x80000000u;           // stays out of any buffer's range here

    char *lds = (char *)ds_dynamic_lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int pix_per_seg = p.RT * p.Wo;
    const int tile_bytes = p.NI * p.seg_bytes;
    const bool rowblock = p.NI == 1;                // one block of rows of one image; else: NI whole images
    const int pad = -p.dw_min;
    // rows of a segment's tile that are enumerated for staging: all of them for a row block (the per-tile window
    // decides which are in the image), the in-image ones of a whole-image segment (the same for every tile)
    int e_lo = 0, e_rows = p.rows_in;
    if (!rowblock) {
        e_lo = p.dh_min < 0 ? -p.dh_min : 0;
        const int hi = p.H - p.dh_min < p.rows_in ? p.H - p.dh_min : p.rows_in;
        e_rows = hi > e_lo ? hi - e_lo : 1;
    }
    const int x_row_bytes = p.W * p.x_pix_stride * 2;
    const size_t x_chunk_bytes = (size_t)p.x_chunk_stride * 2;
    const int n_chunks = p.Cin / CKH;

    // ---- tile-invariant descriptors ----
    const float rcp_w = 1.0f / (float)p.W, rcp_er = 1.0f / (float)e_rows, rcp_pps = 1.0f / (float)pix_per_seg,
                rcp_wc = 1.0f / (float)p.Wo;
    // Item it of a thread is pixel pix0 + it * (NTHR / IPP) of the enumeration (8 channels of it).  In general its
    // global offset g_rel (bytes from the tile's row 0 of its first segment's image) and its LDS offset l_off are
    // per-item tables; for a ROW BLOCK whose width divides NTHR / IPP both advance by a constant per item (LIN): no
    // tables -- up to 32 registers that would otherwise be spilled and reloaded in the middle of the MFMA stream.  The
    // NIT = 16 instantiations exist only in that form (the planner sends everything else to the
    // one-tile-per-workgroup kernel).
    static_assert(LIN || NIT <= 8, "16 items per thread only with derived offsets");
    constexpr int TBL = LIN ? 1 : NIT;
    unsigned g_tab[TBL];
    int l_tab[TBL];
    unsigned g_step = 0;
    int l_step = 0;
    {
        const int q = tid % IPP, pix0 = tid / IPP;
#pragma unroll
        for (int it = 0; it < TBL; ++it) {
            const int pix = pix0 + it * (NTHR / IPP);
            const int vr = ds_div_small(pix, p.W, rcp_w), c = pix - vr * p.W;
            const int sg = ds_div_small(vr, e_rows, rcp_er), rr = e_lo + vr - sg * e_rows;
            const int cc = c - p.dw_min;
            // stride-2 layers keep even tile columns in slots [0, half) and odd ones in [half, cols_in), so that the
            // 32 lanes of a fragment read (stride-2 columns) touch CONSECUTIVE records
            const int pc = (p.IS == 2) ? ((cc & 1) ? p.half + (cc >> 1) : (cc >> 1)) : cc;
            const bool ok = sg < p.NI;
            g_tab[it] = ok ? (unsigned)((((sg * p.H + rr) * p.W + c) * p.x_pix_stride + q * 8) * 2) : OOB;
            l_tab[it] = ok ? sg * p.seg_bytes + rr * p.row_bytes + pc * PSH + q * 16 : CKH * 2;    // spare: pad of record 0
        }
        if (LIN) {      // rows advance by (NTHR / IPP) / W per item, the column stays
            const int rows_per_item = ds_div_small(NTHR / IPP, p.W, rcp_w);
            g_step = (unsigned)(rows_per_item * p.W * p.x_pix_stride * 2);
            l_step = rows_per_item * p.row_bytes;
        }
    }
    const int last_row_l = (p.rows_in - 1) * p.row_bytes + (p.pitch * PSH - 1);      // LIN: last byte of the tile's rows
    auto g_rel = [&](int it) -> unsigned { return LIN ? g_tab[0] + (unsigned)it * g_step : g_tab[LIN ? 0 : it]; };
    // (LIN: an item past the tile's last row is a spare slot: its load is out of range by construction -- beyond the
    // window's end -- and its zeros go to the pad bytes of record 0)
    auto l_off = [&](int it) -> int {
        if (!LIN) return l_tab[LIN ? 0 : it];
        const int o = l_tab[0] + it * l_step;
        return o <= last_row_l ? o : CKH * 2;
    };
    // Which pixel of its 32-pixel sub-tile a lane owns is free (the epilogue un-permutes): chosen so that the two
    // 16-lane SERVICE GROUPS of a ds_read_b128 each read 16 CONSECUTIVE pixels (see conv_mfma_f16_kernel.h)
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
    // halo records of one segment's tile: whole rows outside the enumerated ones, and the columns left / right of the
    // image in the enumerated rows
    const int h_cols = p.cols_in - p.W;
    const int h_rows_recs = (p.rows_in - e_rows) * p.cols_in;
    const int h_recs = h_rows_recs + e_rows * h_cols;          // per segment
    const int h_pieces = 2 * p.NI * h_recs * PPR;              // both buffers
    const float rcp_ppr = 1.0f / (float)PPR, rcp_hr = 1.0f / (float)(h_recs > 0 ? h_recs : 1),
                rcp_ci = 1.0f / (float)p.cols_in, rcp_hc = 1.0f / (float)(h_cols > 0 ? h_cols : 1);

    const float rcp_nn = 1.0f / (float)p.n_ntiles, rcp_spi = 1.0f / (float)p.segs_per_img;
    const int my_c = (lane % LPP) * 8, my_p = lane / LPP;
    const int flags = p.flags;
    const bool out32 = (flags & DS_EPI_OUT_F32) != 0;
    const float clip_lo = (flags & DS_EPI_CLIP) ? 0.0f : -__builtin_inff();
    const float clip_hi = (flags & DS_EPI_CLIP) ? 20.0f : __builtin_inff();
    const ds_buffer ybuf = ds_make_buffer(p.y, p.y_bytes);
    const ds_buffer rbuf = ds_make_buffer((flags & DS_EPI_RESIDUAL) ? (const void *)p.res : (const void *)p.y,
                                          (flags & DS_EPI_RESIDUAL) ? p.res_bytes : 0u);
    float *tb = (float *)lds + wave * (2 * 32 * TP);
    const size_t w_kc_stride = (size_t)NT * p.Cout * 16;                // one 16-channel slab: [tap][Cout][16]
    const size_t w_tap_stride = (size_t)p.Cout * 16;

    // tile t -> (n tile, first segment); the staged rows of the tile as a buffer of their own: from the first in-image
    // row of the tile's window to the end of its last one (x_lo: the byte offset of that first row from the tile's row 0)
    struct TileAt { int tile_n, seg0, lin_base, lin_valid; };
    auto tile_at = [&](int t, TileAt &ta, const char *&xb_base, unsigned &xb_bytes, unsigned &x_lo) {
        const int tile_m = ds_div_small(t, p.n_ntiles, rcp_nn);
        ta.tile_n = t - tile_m * p.n_ntiles;
        ta.seg0 = tile_m * p.NI;
        const int b0 = ds_div_small(ta.seg0, p.segs_per_img, rcp_spi);
        const int sblk = ta.seg0 - b0 * p.segs_per_img;
        const int h0 = p.IS * sblk * p.RT + p.dh_min;                  // image row of tile row 0
        const int live = rowblock ? 1 : (p.n_segs - ta.seg0 < p.NI ? p.n_segs - ta.seg0 : p.NI);
        const int lo_row = h0 < 0 ? -h0 : 0;
        int hi_row = live * p.H - h0;                                  // one past the last in-image row, tile-relative
        const int ext = (p.NI - 1) * p.H + p.rows_in;                  // rows the enumeration can reach
        hi_row = hi_row < ext ? hi_row : ext;
        x_lo = (unsigned)(lo_row * x_row_bytes);
        xb_base = (const char *)p.x + (size_t)(b0 * p.H + h0 + lo_row) * x_row_bytes;
        xb_bytes = (unsigned)((hi_row > lo_row ? hi_row - lo_row : 0) * x_row_bytes);
        ta.lin_base = (b0 * p.Ho + sblk * p.RT) * p.Wo;
        ta.lin_valid = rowblock ? (p.Ho - sblk * p.RT < p.RT ? p.Ho - sblk * p.RT : p.RT) * p.Wo : live * pix_per_seg;
    };

    // the first tile of a workgroup is its index; every further one is drawn from a counter, one tile ahead of its use
    // (t_next is known at the top of tile t_cur: the filter ring runs on into its fragments).
    // XCD-AWARE QUEUES (p.sched_queues == 8; grids that are a multiple of 8 whose n-tile count divides 8): workgroup b is
    // dispatched to XCD b % 8 (observed placement; it only decides which L2 the filters are found in) and draws from
    // queue b % 8, whose tiles are t = 8 j + (b % 8) -- all of ONE n tile (t % n_ntiles is the same for every tile of
    // the queue), so an XCD's L2 keeps holding the filters of the n tiles its statically assigned first tiles had.  With
    // ONE queue over all tiles a drawn tile is any n tile: on the 512-output-channel layers (filter banks of 4.7 and
    // 6.5 MB against a 4 MB L2 per XCD) every XCD then walks the whole bank and the ring's loads miss the L2 --
    // measured 178 us against 139 us for the one-tile-per-workgroup kernel on the 512-channel 3x3 layer, and a bimodal
    // 185 / 235 us on the 256 -> 512 5x5 one (tools/f16_layer_ab.py).
    const int nq = p.sched_queues, q = (int)blockIdx.x % nq;
    int j_cur = (int)blockIdx.x / nq;
    const int j_static = (int)gridDim.x / nq, t_end = p.tiles;
    unsigned *const q_next = p.sched + q;
    int *const sched_word = (int *)(lds + p.sched_lds);
    if (tid == 0) *sched_word = j_static + (int)ds_atomic_inc(q_next);
    __syncthreads();
    int t_cur = j_cur * nq + q;
    int t_next = ds_uniform(*sched_word) * nq + q;
    TileAt ta = {0, 0, 0, 0};
    const char *xb_base = (const char *)p.x;
    unsigned xb_bytes = 0, x_lo = 0;
    f32x4 st[NIT];
    if (t_cur < t_end) {
        tile_at(t_cur, ta, xb_base, xb_bytes, x_lo);
        const ds_buffer xb = ds_make_buffer(xb_base, xb_bytes);
#pragma unroll
        for (int it = 0; it < NIT; ++it) st[it] = ds_buffer_load_f32x4(xb, g_rel(it) - x_lo);
    }
    size_t lane_w = ((size_t)(ta.tile_n * NTILE + wn * NSUB * 32 + l31) * 16 + 8 * lhi);       // in halfs
    // unit u of a chunk = (k-step u / NT, tap u % NT): filter slab KPT*chunk + u / NT, tap u % NT (k-step-major)
    auto w_unit = [&](size_t lw, int chunk, int u) {
        return p.w + lw + (size_t)(KPT * chunk + (u / NT)) * w_kc_stride + (size_t)(u % NT) * w_tap_stride;
    };
    f16x8 bq[RU][NSUB];
    if constexpr (RING_THROUGH) {
#pragma unroll
        for (int d = 0; d < RU; ++d)
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns) bq[d][ns] = *(const f16x8 *)(w_unit(lane_w, 0, d) + (size_t)ns * 32 * 16);
    }

    f32x16 acc[MSUB][NSUB];         // never cleared: the first unit of a tile accumulates into a literal zero
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    int tile_no = 0;
    (void)tile_no;
    for (; t_cur < t_end; ++tile_no) {
        DS_PK_STAMP(0);
        int t_drawn = 0;                        // the tile after next, drawn now, published before the epilogue's barrier
        if (tid == 0) t_drawn = (j_static + (int)ds_atomic_inc(q_next)) * nq + q;
        // every filter-fragment address below is tile-invariant; hoisted out of this loop they would be dozens of live
        // 64-bit values (spilled, and reloaded from scratch between the MFMAs): keep them derived where they are used
        DS_OPAQUE_VGPR(lane_w);
        if constexpr (LIN) {                    // ... and so would the derived staging offsets be
            DS_OPAQUE_VGPR(g_tab[0]);
            DS_OPAQUE_VGPR(l_tab[0]);
        }
        const int n_base = ta.tile_n * NTILE + wn * NSUB * 32;
        const int col = n_base + my_c;
        // the next tile (its filter fragments are what the ring runs on into)
        TileAt tn = ta;
        const char *nb_base = xb_base;
        unsigned nb_bytes = 0, n_lo = 0;
        const bool has_next = t_next < t_end;
        if (has_next) tile_at(t_next, tn, nb_base, nb_bytes, n_lo);
        size_t lane_wn = ((size_t)(tn.tile_n * NTILE + wn * NSUB * 32 + l31) * 16 + 8 * lhi);
        DS_OPAQUE_VGPR(lane_wn);

        if constexpr (!RING_THROUGH) {          // the ring's first RU units, per tile
#pragma unroll
            for (int d = 0; d < RU; ++d)
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns) bq[d][ns] = *(const f16x8 *)(w_unit(lane_w, 0, d) + (size_t)ns * 32 * 16);
        }
        // ---- (1) halo of both buffers, the first chunk's pixels -> buffer 0 ----
        ds_lds_barrier();                       // the previous tile's epilogue has finished with the LDS
        for (int i = tid; i < h_pieces; i += NTHR) {
            const int r = ds_div_small(i, PPR, rcp_ppr), piece = i - r * PPR;
            const int bs = ds_div_small(r, h_recs, rcp_hr), h = r - bs * h_recs;        // bs: buffer * NI + segment
            const int buf = bs >= p.NI ? 1 : 0, sg = bs - buf * p.NI;
            int row, cc;
            if (h < h_rows_recs) {                                                      // a row outside the enumerated ones
                const int rk = ds_div_small(h, p.cols_in, rcp_ci);
                cc = h - rk * p.cols_in;
                row = rk < e_lo ? rk : rk + e_rows;
            } else {                                                                    // a column beside the image
                const int hh = h - h_rows_recs;
                const int ri = ds_div_small(hh, h_cols, rcp_hc), k = hh - ri * h_cols;
                row = e_lo + ri;
                cc = k < pad ? k : p.W + k;
            }
            const int pc = (p.IS == 2) ? ((cc & 1) ? p.half + (cc >> 1) : (cc >> 1)) : cc;
            *(f32x4 *)(lds + buf * tile_bytes + sg * p.seg_bytes + row * p.row_bytes + pc * PSH + 16 * piece) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) *(f32x4 *)(lds + l_off(it)) = st[it];
        ds_lds_barrier();
        DS_PK_STAMP(1);

        // ---- (2) the chunks: NU units of NMF MFMAs each, one side operation after each MFMA ----
        //   odd slots:  the NEXT unit's pixel fragments (LDS -> registers, double-buffered per unit)
        //   even slots: the filter-ring refills (L2 -> registers, RU-1 units ahead), then the staging traffic: loads of
        //               the next chunk's pixels in the first units, their LDS writes in the last ones; in the LAST
        //               chunk the residual rows of the epilogue instead
        f32x4 resv[NE][NRI];
        auto res_load = [&](int re, int rk, int col_) {     // step re = (sub-tile re / NH, channel half re % NH)
            const int m = (wm * MSUB + re / NH) * 32 + rk * PPI + my_p;
            return ds_buffer_load_f32x4(rbuf, m < ta.lin_valid ? (unsigned)((ta.lin_base + m) * p.Cout + col_ + 64 * (re % NH)) * 2u
                                                               : DS_BUFFER_OOB);
        };
        auto run_chunk = [&](auto first_tag, auto last_tag, int chunk, const char *buf, char *obuf) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
            f16x8 a[2][MSUB];
#pragma unroll
            for (int ms = 0; ms < MSUB; ++ms) {
                DS_OPAQUE_VGPR(a_off[ms]);     // keep the NU x MSUB fragment addresses out of registers
                a[0][ms] = *(const f16x8 *)(buf + a_off[ms] + tap_off(0));
            }
            const ds_buffer xn = ds_make_buffer(xb_base + (size_t)(chunk + 1) * x_chunk_bytes, xb_bytes);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int cur = u & 1, slot = u % RU;
                const bool more = u + 1 < NU;
                const char *nfrag = buf + tap_off((u + 1) % NT) + 32 * ((u + 1) / NT);
                // the slot the previous unit consumed is refilled with the unit RU - 1 ahead of this one: of this chunk,
                // of the next chunk, or -- past the tile's last chunk -- of the NEXT TILE's first chunk
                const int ur = u - 1 + RU;
                const _Float16 *rw = (LAST && ur >= NU) ? w_unit(lane_wn, 0, ur - NU)
                                                        : w_unit(lane_w, ur >= NU ? chunk + 1 : chunk, ur >= NU ? ur - NU : ur);
                const int rslot = (u + RU - 1) % RU;
#pragma unroll
                for (int q = 0; q < NMF; ++q) {
                    const int ms = q / NSUB, ns = q % NSUB;
                    acc[ms][ns] = ds_mfma_32x32x16_f16(bq[slot][ns], a[cur][ms], (FIRST && u == 0) ? zero16 : acc[ms][ns]);
                    if (q & 1) {
                        const int lm = q >> 1;
                        if (lm < MSUB && more) a[cur ^ 1][lm] = *(const f16x8 *)(nfrag + a_off[lm]);
                    } else {
                        const int e = q >> 1;
                        if (e < NSUB) {
                            if (RING_THROUGH || !(LAST && ur >= NU)) bq[rslot][e] = *(const f16x8 *)(rw + (size_t)e * 32 * 16);
                        } else if constexpr (!LAST) {
                            const int s = e - NSUB;
                            if (u < UL) {                           // next chunk's pixels -> registers
                                const int it = u * SPU + s;
                                if (it < NIT) st[it] = ds_buffer_load_f32x4(xn, g_rel(it) - x_lo);
                            } else if (u >= NU - UL) {              // ... -> the other LDS buffer
                                const int it = (u - (NU - UL)) * SPU + s;
                                if (it < NIT) *(f32x4 *)(obuf + l_off(it)) = st[it];
                            }
                        } else {                                    // residual rows -> registers, units NU-2-ULR ..
                            const int ri = (u - (NU - 2 - ULR)) * SPU + (e - NSUB);
                            if (u >= NU - 2 - ULR && ri < NRES) {
                                resv[ri / NRI][ri % NRI] = res_load(ri / NRI, ri % NRI, col);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (n_chunks == 1) {
            run_chunk(std::true_type{}, std::true_type{}, 0, lds, lds);
        } else {
            run_chunk(std::true_type{}, std::false_type{}, 0, lds, lds + tile_bytes);
            ds_lds_barrier();                   // LDS only: the filter-ring loads in flight stay in flight
            for (int i = 1; i + 1 < n_chunks; ++i) {
                char *b0 = lds + (i & 1) * tile_bytes, *b1 = lds + ((i & 1) ^ 1) * tile_bytes;
                run_chunk(std::false_type{}, std::false_type{}, i, b0, b1);
                ds_lds_barrier();
            }
            run_chunk(std::false_type{}, std::true_type{}, n_chunks - 1, lds + ((n_chunks - 1) & 1) * tile_bytes, lds);
        }

        // ---- (3) epilogue (see conv_mfma_f16_kernel.h: transposed accumulators turned around through wave-private LDS
        // buffers, whole pixel rows stored).  At its head the NEXT tile's first input chunk is requested. ----
        DS_PK_STAMP(2);
        if (tid == 0) *sched_word = t_drawn;
        ds_lds_barrier();                       // every wave is done reading the pixel tile
        DS_PK_STAMP(3);
        const int t_after = ds_uniform(*sched_word);
        // (16 items in flight through the epilogue do not fit the register file next to it: those kernels request the
        // next chunk at the epilogue's end instead -- it then has the tile-top barrier and the halo zeroing to arrive in)
        constexpr bool EARLY_PREFETCH = NIT <= 8;
        auto prefetch_next = [&]() __attribute__((always_inline)) {
            const ds_buffer xb = ds_make_buffer(nb_base, has_next ? nb_bytes : 0u);
#pragma unroll
            for (int it = 0; it < NIT; ++it) st[it] = ds_buffer_load_f32x4(xb, g_rel(it) - n_lo);
        };
        // The BatchNorm scale / shift rows are requested BEFORE the next tile's pixels: loads retire through one in-order
        // counter, so the first fma of the epilogue -- which needs these rows -- would otherwise wait for the whole HBM
        // prefetch issued ahead of them (round 6: `s_waitcnt vmcnt(2)` behind 8 prefetch + 4 table loads in the ISA;
        // a tile's epilogue stood still for one HBM round trip).
#ifdef DS_EPI_TABLES_LATE
        if constexpr (EARLY_PREFETCH) prefetch_next();
        __builtin_amdgcn_sched_barrier(0);
#endif
        f32x4 sc[NH][2], sh[NH][2];
#pragma unroll
        for (int nh = 0; nh < NH; ++nh)
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                sc[nh][hq] = f32x4{1.0f, 1.0f, 1.0f, 1.0f};
                sh[nh][hq] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                if (flags & DS_EPI_AFFINE) {
                    sc[nh][hq] = *(const f32x4 *)(p.scale + col + 64 * nh + 4 * hq);
                    sh[nh][hq] = *(const f32x4 *)(p.shift + col + 64 * nh + 4 * hq);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
#ifndef DS_EPI_TABLES_LATE          // (A/B builds, tools/f16_ab.py: the round-5 order -- prefetch first)
        if constexpr (EARLY_PREFETCH) prefetch_next();
#endif
        auto put_tile = [&](int e) {                // accumulators of step e -> this wave's buffer e & 1
            float *dst = tb + (e & 1) * (32 * TP);
            const int ms = e / NH, nh = e % NH;
#pragma unroll
            for (int ns = 0; ns < ENS; ++ns)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[ms][ENS * nh + ns][4 * g + j];
                    *(f32x4 *)(dst + lpix * TP + ns * 32 + 8 * g + 4 * lhi) = v;
                }
        };
        // element offset of (pixel, first channel) = pixel * y_mul + y_add: channels-last, or 16-channel planes
        const unsigned y_mul = p.y_plane_stride ? 16u : (unsigned)p.Cout;
        auto write_out = [&](auto f32_tag) __attribute__((always_inline)) {
            constexpr bool OUT32 = decltype(f32_tag)::value;
            put_tile(0);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int ms = e / NH, nh = e % NH;
                const int cb = e & 1;
                const int ecol = col + 64 * nh;
                const unsigned y_add = p.y_plane_stride ? (unsigned)(ecol >> 4) * p.y_plane_stride + (unsigned)(ecol & 15)
                                                        : (unsigned)ecol;
                ds_wave_sync();                     // step e is in its buffer (LDS runs a wave's operations in order)
                const float *src = tb + cb * (32 * TP);
                f32x4 tv[NRI][2];
#pragma unroll
                for (int k = 0; k < NRI; ++k)
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq) tv[k][hq] = *(const f32x4 *)(src + (k * PPI + my_p) * TP + my_c + 4 * hq);
                if (e + 1 < NE) put_tile(e + 1);    // the next step's turn-around travels while this one is finished
                if (e + RPRE < NE) {
#pragma unroll
                    for (int k = 0; k < NRI; ++k) resv[e + RPRE][k] = res_load(e + RPRE, k, col);
                }
#pragma unroll
                for (int k = 0; k < NRI; ++k) {
                    const f16x8 r8 = __builtin_bit_cast(f16x8, resv[e][k]);
                    const int m = (wm * MSUB + ms) * 32 + k * PPI + my_p;
                    f32x4 o[2];
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                        for (int j2 = 0; j2 < 2; ++j2) {
                            // two channels per packed instruction: fma (BatchNorm affine), add (residual); clip per channel
                            const ds_f32x2 v = {tv[k][hq][2 * j2], tv[k][hq][2 * j2 + 1]};
                            const ds_f32x2 s2 = {sc[nh][hq][2 * j2], sc[nh][hq][2 * j2 + 1]},
                                           h2 = {sh[nh][hq][2 * j2], sh[nh][hq][2 * j2 + 1]};
                            const ds_f16x2 rh = {r8[4 * hq + 2 * j2], r8[4 * hq + 2 * j2 + 1]};
                            ds_f32x2 t = v * s2 + h2;
                            t = t + __builtin_convertvector(rh, ds_f32x2);
                            o[hq][2 * j2] = fminf(fmaxf(t[0], clip_lo), clip_hi);
                            o[hq][2 * j2 + 1] = fminf(fmaxf(t[1], clip_lo), clip_hi);
                        }
                    const unsigned vo = m < ta.lin_valid ? (unsigned)(ta.lin_base + m) * y_mul + y_add : DS_BUFFER_OOB;
                    if constexpr (OUT32) {
                        const unsigned bo = vo != DS_BUFFER_OOB ? vo * 4u : DS_BUFFER_OOB;
                        ds_buffer_store_out_f32x4(ybuf, bo, o[0]);
                        ds_buffer_store_out_f32x4(ybuf, bo != DS_BUFFER_OOB ? bo + 16u : DS_BUFFER_OOB, o[1]);
                    } else {
                        ds_u32x4 hb;
#pragma unroll
                        for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                            for (int j2 = 0; j2 < 2; ++j2)
                                hb[2 * hq + j2] = __builtin_bit_cast(unsigned, __builtin_convertvector(
                                                                                   ds_f32x2{o[hq][2 * j2], o[hq][2 * j2 + 1]}, ds_f16x2));
                        ds_buffer_store_out_f32x4(ybuf, vo != DS_BUFFER_OOB ? vo * 2u : DS_BUFFER_OOB, __builtin_bit_cast(f32x4, hb));
                    }
                }
            }
        };
        if (out32) write_out(std::true_type{});
        else write_out(std::false_type{});
        if constexpr (!EARLY_PREFETCH) prefetch_next();
        DS_PK_STAMP(4);
        ta = tn;
        xb_base = nb_base;
        xb_bytes = nb_bytes;
        x_lo = n_lo;
        lane_w = lane_wn;
        t_cur = t_next;
        t_next = t_after;
    }
    // the last workgroup to leave hands the slot back zeroed (no counter is touched after a workgroup's own `done`)
    if (tid == 0 && ds_atomic_inc(p.sched + DS_SCHED_DONE) == gridDim.x - 1) {
        for (int i = 0; i <= DS_SCHED_DONE; ++i) p.sched[i] = 0u;
    }
}

template <int KS, int MSUB, int NSUB, int WM, int WN, int CK>
static void launch_nit_p(const PlanH &pl, void *stream) {
    constexpr int NTHR = WM * WN * 64;
    if (!pl.lin)
        DS_LAUNCH_BIG_LDS((conv_mfma_f16_pkernel<KS, MSUB, NSUB, WM, WN, 8, CK, false>), pl.grid, NTHR, pl.lds_bytes, stream, pl.k);
    else if (pl.nit <= 8)
        DS_LAUNCH_BIG_LDS((conv_mfma_f16_pkernel<KS, MSUB, NSUB, WM, WN, 8, CK, true>), pl.grid, NTHR, pl.lds_bytes, stream, pl.k);
    else
        DS_LAUNCH_BIG_LDS((conv_mfma_f16_pkernel<KS, MSUB, NSUB, WM, WN, 16, CK, true>), pl.grid, NTHR, pl.lds_bytes, stream, pl.k);
}

template <int KS, int CK = 32>
static void launch_p(const PlanH &pl, void *stream) {
    if (pl.cfg == 0) launch_nit_p<KS, 5, 2, 1, 2, CK>(pl, stream);          // 160x128, two waves
    else if (pl.cfg == 1) launch_nit_p<KS, 5, 2, 1, 4, CK>(pl, stream);     // 160x256, four waves
    else if (pl.cfg == 2) launch_nit_p<KS, 5, 2, 2, 2, CK>(pl, stream);     // 320x128
    else if (pl.cfg == 3) launch_nit_p<KS, 5, 2, 2, 1, CK>(pl, stream);     // 320x64, two waves
    else if (pl.cfg == 4) launch_nit_p<KS, 4, 2, 1, 2, CK>(pl, stream);     // 128x128, two waves
    else if (pl.cfg == 5) launch_nit_p<KS, 4, 2, 1, 4, CK>(pl, stream);     // 128x256
    else if (pl.cfg == 6) launch_nit_p<KS, 5, 2, 4, 1, CK>(pl, stream);     // 640x64, four waves
    else {
        // cfg 7 (persistent kernel only; the planner upgrades a cfg-4 plan of a layer with Cout % 256 == 0): 128x256,
        // two waves, each a 128 x 128 register tile (NSUB = 4: 256 accumulator registers, every LDS fragment read feeds
        // four MFMAs) -- the 512-channel 3x3 layers on 10x4 maps
        if constexpr (KS == 3 && CK == 32) launch_nit_p<KS, 4, 4, 1, 2, CK>(pl, stream);
    }
}

}  // namespace
#endif
