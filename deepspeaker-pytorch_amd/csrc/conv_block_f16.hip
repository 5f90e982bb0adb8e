// conv_block_f16.hip -- one whole BasicBlock of the eval forward in ONE kernel on the fp16 matrix cores:
//
//     y = clip( bn2( conv3x3( clip( bn1( conv3x3(x) ) ) ) ) + x )                 (reference model.py:66-82)
//
// for the two shallow stages (64 channels on 80x32 maps, 128 channels on 40x16 maps at T = 160).  There the contraction
// of one 3x3 layer is only K = 576 / 1152 deep and a workgroup of conv_mfma_f16_kernel spends more clocks in its
// prologue and epilogue than in its MFMA stream (DESIGN_LOG.md 3.1).  Fusing the block's two convolutions pays one prologue
// and one HBM epilogue for two MFMA streams: the intermediate activation (rows r0-1 .. r0+R of the image, the halo
// rows recomputed) goes from the accumulators straight into LDS in the pixel-record layout the second convolution
// reads its fragments from -- it never exists in HBM.  Arithmetic, rounding points (the intermediate is rounded to fp16
// exactly as the unfused path stores it) and accumulation order are those of two ds_conv_fwd_f16 calls: results are
// bit-identical (asserted by the tests).
#include <ds_device.h>
#include <algorithm>
#include <type_traits>
#include "ds_common.h"

namespace {

constexpr int BK_R = 8;                 // output rows per workgroup; the first convolution computes BK_R + 2
constexpr int BK_CK = 32;               // input channels per chunk (two MFMA k-steps per tap)
constexpr int BK_PS = 80;               // bytes per staged input record: 32 halfs + 16 B pad

struct BlockK {
    const _Float16 *x;                  // [B,H,W,C] fp16 channels-last: block input and residual
    const _Float16 *wa, *wb;            // packed [C/16][9][C][16] (ds_pack_conv_weight_f16)
    const float *sa, *ha, *sb, *hb;     // folded BatchNorm of the two layers (scale, shift)
    void *y;                            // [B,H,W,C] fp16 (f32 with DS_EPI_OUT_F32; plane-major with DS_EPI_OUT_PLANES16)
    int B, H, W, C;
    int tiles_per_img, n_tiles;
    int flags;
    unsigned y_bytes, x_bytes, y_plane_stride;
    const int *lens;                    // MASKED: rows of each image that belong to its utterance (zero-padded batches)
    int xcd_slots;                      // > 0: workgroups per XCD (grid = 8 * xcd_slots): image b is served by XCD b % 8
    unsigned *sched;                    // tile-scheduling slot (ds_device.h): next[queue], done
    int sched_lds;                      // byte offset of the LDS word tile indices are passed through
#ifdef DS_F16_PROBE                     // tools/block_phase_probe.py builds: s_memtime stamps of the phases of each tile
    long long *probe;
#endif
};

#ifdef DS_F16_PROBE
// stamp i of the workgroup's tile number `tile_no` (only tiles 0..3 are recorded): probe[(wg * 4 + tile_no) * 8 + i]
#define DS_BLK_STAMP(i) do { if (p.probe && threadIdx.x == 0 && tile_no < 4) p.probe[((size_t)blockIdx.x * 4 + tile_no) * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define DS_BLK_STAMP(i) ((void)0)
#endif

#ifndef DS_BLOCK_RING
#define DS_BLOCK_RING 6
#endif

// PERSISTENT workgroups: the grid is what the chip holds at once (two 2-wave workgroups per CU) and a workgroup walks
// tiles (image, block of BK_R output rows) in a loop.  What that buys, per tile (tools/f16_phase_probe.py on the
// one-tile-per-workgroup form of these kernels: 11 k clocks of prologue next to a 16 k MFMA stream at stage 1):
//   * the staging / fragment / output descriptors -- a dozen reciprocal divisions per thread -- are tile-INVARIANT
//     (every tile has the same shape; only a base offset and the in-image row window change): computed once;
//   * the next tile's first input chunk is loaded (HBM -> registers) in the idle staging slots of this tile's second
//     convolution, and the filter ring runs through from one MFMA stream into the next (first layer -> second layer ->
//     next tile's first layer): no stream starts by waiting for memory;
//   * the residual rows (the block's own input, just staged: L2-hot) are requested before the second convolution's
//     stream instead of at the head of the epilogue;
//   * only what must be zero is zeroed: the two halo columns of each tile.  Rows outside the image are STAGED as zeros
//     (out-of-range buffer loads), so every tile stages the same ROWS_IN rows.
// Arithmetic and accumulation order are unchanged: results stay bit-identical to two ds_conv_fwd_f16 calls.
//
// WM x WN waves, each a 160x64 register tile for the first convolution (10 rows x W pixels per WM) and 128x64 for the
// second (8 rows): the map is W = 16 * WM pixels wide (MT_A = 160 * WM = 10 * W).  NIT: 16-byte staging items per thread
// and chunk (12 rows x W pixels x 4 quarters over the workgroup's threads, exactly).
// MASKED (variable-length batches): rows past an image's own extent are zero in the intermediate and in the output, as
// ds_mask_rows makes them after each of the two layers -- every kept row equals the utterance's own forward.
//
// Register budget (one wave per SIMD: 512 registers, of which the 160 accumulators live in the AGPR half and everything
// else must fit the 256 architectural ones or be shuttled): the persistent loop keeps NO per-tile table alive -- staging
// offsets are the invariant g_rel minus a per-tile scalar inside a per-tile buffer descriptor, output offsets are
// recomputed in the epilogue, the residual rows are requested in the LAST units of the second convolution and the next
// tile's first chunk at the head of the epilogue (not one MFMA stream earlier).
template <int WM, int WN, int NIT, bool MASKED = false>
__global__ void __launch_bounds__(WM * WN * 64) DS_ONE_WAVE_PER_SIMD conv_block3x3_f16_kernel(const BlockK p) {
    constexpr int NTHR = WM * WN * 64;
    constexpr int W = 16 * WM, WSH = WM == 2 ? 5 : 4;          // map width (a power of two) and its log2
    constexpr int MSA = 5, MSB = 4, NSUB = 2;
    constexpr int NT = 9, NU = 2 * NT, RU = DS_BLOCK_RING;      // filter ring, in units (NU % RU == 0)
    constexpr int NMFA = MSA * NSUB, NMFB = MSB * NSUB;
    constexpr int SPU = (NMFA + 1) / 2 - NSUB;                  // staging slots per unit, first convolution
    constexpr int UL = (NIT + SPU - 1) / SPU;
    constexpr int SPUB = NMFB / 2 - NSUB;                       // spare slots per unit, second convolution
    constexpr int ROWS_A = BK_R + 2, ROWS_IN = BK_R + 4;
    constexpr int TP = NSUB * 32 + 4, LPP = NSUB * 4, PPI = 64 / LPP, NRI = 32 / PPI;
    constexpr int NRES = MSB * NRI, ULR = (NRES + SPUB - 1) / SPUB;     // residual loads / the units that carry them
    static_assert(WM == 1 || WM == 2, "map width 16 or 32");
    static_assert(2 * UL <= NU && ULR + 2 <= NU && NU % RU == 0, "not enough units for the staging traffic");
    static_assert(NIT * NTHR == ROWS_IN * W * 4, "staging items must tile the input rows exactly");
    constexpr unsigned OOB = 0x80000000u;       // stays out of range of any buffer here after adding a chunk offset
    constexpr int pitch = W + 2;                                     // records per tile row (both tiles)
    constexpr int tileA_bytes = ROWS_IN * pitch * BK_PS;

    char *lds = (char *)ds_dynamic_lds();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    constexpr int C = WM == 2 ? 64 : 128;                            // the two supported stages (checked by the host)
    constexpr int RSB = C * 2 + 16;                                  // bytes per intermediate record
    constexpr int n_chunks = C / BK_CK;
    const int n_base = wn * NSUB * 32;
    size_t lane_w = ((size_t)(n_base + l31) * 16 + 8 * lhi);
    const size_t w_kc_stride = (size_t)NT * C * 16, w_tap_stride = (size_t)C * 16;
    auto w_unit = [&](const _Float16 *w, int chunk, int u) {         // k-step-major units, as conv_mfma_f16_kernel
        return w + lane_w + (size_t)(2 * chunk + (u / NT)) * w_kc_stride + (size_t)(u % NT) * w_tap_stride;
    };

    // ---- which tiles this workgroup walks ----
    // plain: tile = blockIdx.x + k * gridDim.x.  XCD-aware (grid = 8 * xcd_slots, batches of >= 8 images): workgroup w
    // is dispatched to XCD w % 8 (observed, not contractual -- it only decides which L2 the halo rows are found in);
    // that XCD serves images b = 8 j + (w % 8), so the row blocks of an image share one L2.
    const int xcd = (int)blockIdx.x & 7, slot0 = (int)blockIdx.x >> 3;
    const int my_imgs = p.xcd_slots > 0 ? (p.B - xcd + 7) / 8 : 0;
    // the first tile of a workgroup is its (per-queue) index; every further one is drawn from the queue's counter, one
    // tile ahead of its use (t_next is known at the top of tile t_cur)
    int t_cur = p.xcd_slots > 0 ? slot0 : (int)blockIdx.x;
    const int t_static = p.xcd_slots > 0 ? p.xcd_slots : (int)gridDim.x;     // tiles handed out by index
    const int t_end = p.xcd_slots > 0 ? my_imgs * p.tiles_per_img : p.n_tiles;
    unsigned *const q_next = p.sched + (p.xcd_slots > 0 ? xcd : 0);
    int *const sched_word = (int *)(lds + p.sched_lds);
    if (tid == 0) *sched_word = t_static + (int)ds_atomic_inc(q_next);
    __syncthreads();
    int t_next = ds_uniform(*sched_word);
    const float rcp_tpi = 1.0f / (float)p.tiles_per_img;
    auto tile_of = [&](int t, int &b, int &r0) {
        const int im = ds_div_small(t, p.tiles_per_img, rcp_tpi);
        r0 = (t - im * p.tiles_per_img) * BK_R;
        b = p.xcd_slots > 0 ? im * 8 + xcd : im;
    };

    // ---- tile-invariant descriptors ----
    // staging item it of this thread: 8 channels (16 B) of pixel (tile row vr, column c), tile rows 0 .. ROWS_IN-1 =
    // image rows r0-2 .. r0+R+1.  g_rel: byte offset from the tile's row 0; l_off: where it goes in the LDS tile.
    unsigned g_rel[NIT];
    int l_off[NIT];
    {
        const int q = tid & 3, pix0 = tid >> 2;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int pix = pix0 + it * (NTHR / 4);
            const int vr = pix >> WSH, c = pix & (W - 1);
            g_rel[it] = (unsigned)((pix * C + q * 8) * 2);
            l_off[it] = (vr * pitch + c + 1) * BK_PS + q * 16;
        }
    }
    // lane -> pixel permutation inside a 32-pixel sub-tile (conflict-free ds_read_b128 service groups, see
    // conv_mfma_f16_kernel.h)
    const int lpix = (l31 < 4 || l31 >= 28) ? l31
                   : (l31 < 12) ? l31 + 12 : (l31 < 16) ? l31 - 8 : (l31 < 20) ? l31 + 8 : l31 - 12;
    int a_off[MSA];                         // first convolution: top-left record of the pixel's 3x3 window
#pragma unroll
    for (int ms = 0; ms < MSA; ++ms) {
        const int m = (wm * MSA + ms) * 32 + lpix;
        a_off[ms] = ((m >> WSH) * pitch + (m & (W - 1))) * BK_PS + 16 * lhi;
    }
    auto tap_off = [&](int tt, int rs) { return ((tt / 3) * pitch + (tt % 3)) * rs; };
    const ds_buffer ybuf = ds_make_buffer(p.y, p.y_bytes);
    const ds_buffer rbuf = ds_make_buffer(p.x, p.x_bytes);          // residual rows
    const int x_row_bytes = W * C * 2;
    const int my_c = (lane % LPP) * 8, my_p = lane / LPP;
    const int col = n_base + my_c;
    const bool out32 = (p.flags & DS_EPI_OUT_F32) != 0;
    float *tb = (float *)lds + wave * (2 * 32 * TP);

    // The staged rows of a tile as a buffer of their own: it starts at the first in-image row of the tile's window and
    // ends with the last one, so a row outside the image is out of range -- it reads as zeros and is written to LDS as
    // such -- whichever side it is on: voffset = g_rel - lo wraps to > 2 GB for the rows above the window.
    auto stage_window = [&](int b, int r0, ds_buffer &xb, unsigned &lo) {
        const int h0 = r0 - 2;
        const int first = h0 < 0 ? -h0 : 0;
        const int last = p.H - h0 < ROWS_IN ? p.H - h0 : ROWS_IN;            // one past the last in-image tile row
        lo = (unsigned)(first * x_row_bytes);
        xb = ds_make_buffer((const char *)p.x + (size_t)(b * p.H + h0 + first) * x_row_bytes,
                            (unsigned)((last > first ? last - first : 0) * x_row_bytes));
    };

    f16x8 bq[RU][NSUB];
#pragma unroll
    for (int d = 0; d < RU; ++d)
#pragma unroll
        for (int ns = 0; ns < NSUB; ++ns) bq[d][ns] = *(const f16x8 *)(w_unit(p.wa, 0, d) + (size_t)ns * 32 * 16);

    f32x16 acc[MSA][NSUB];          // never cleared: the first unit of each layer accumulates into a literal zero
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    f32x4 st[NIT];
    int b = 0, r0 = 0;
    ds_buffer xbuf = rbuf;
    unsigned x_lo = 0;
    if (t_cur < t_end) {
        tile_of(t_cur, b, r0);
        stage_window(b, r0, xbuf, x_lo);
#pragma unroll
        for (int it = 0; it < NIT; ++it) st[it] = ds_buffer_load_f32x4(xbuf, g_rel[it] - x_lo);
    }

    int tile_no = 0;
    (void)tile_no;
    for (; t_cur < t_end; ++tile_no) {
        DS_BLK_STAMP(0);
        int t_drawn = 0;                        // the tile after next, drawn now, published before the epilogue's barrier
        if (tid == 0) t_drawn = t_static + (int)ds_atomic_inc(q_next);
        // every filter-fragment address below is tile-invariant; hoisted out of this loop they would be ~70 live 64-bit
        // values (spilled, and reloaded from scratch between the MFMAs): keep them derived where they are used
        DS_OPAQUE_VGPR(lane_w);
        int h_valid = p.H;                                           // rows of this image that carry data
        if (MASKED) {
            const int len = p.lens[b];
            h_valid = len < p.H ? len : p.H;
        }
        // ---- (1) halo columns of both input buffers, the first chunk's pixels -> buffer 0 ----
        ds_lds_barrier();                       // the previous tile's epilogue has finished with the LDS (its stores to
                                                // HBM stay in flight: only LDS traffic is waited for)
        for (int i = tid; i < 2 * ROWS_IN * 2 * (BK_PS / 16); i += NTHR) {
            const int piece = i % (BK_PS / 16), rec = i / (BK_PS / 16);          // rec: (buffer, row, left | right)
            const int side = rec & 1, row = (rec >> 1) % ROWS_IN, buf = (rec >> 1) / ROWS_IN;
            *(f32x4 *)(lds + buf * tileA_bytes + (row * pitch + side * (W + 1)) * BK_PS + 16 * piece) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) *(f32x4 *)(lds + l_off[it]) = st[it];
        ds_lds_barrier();
        DS_BLK_STAMP(1);

        // ---- (2) first convolution: the MFMA stream of conv_mfma_f16_kernel (double-buffered tile, one side operation
        // per MFMA); its last chunk's ring refills already fetch the second layer's first filter fragments ----
        auto run_chunk = [&](auto first_tag, auto last_tag, int chunk, const char *buf, char *obuf) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_tag)::value, LAST = decltype(last_tag)::value;
            f16x8 a[2][MSA];
#pragma unroll
            for (int ms = 0; ms < MSA; ++ms) {
                DS_OPAQUE_VGPR(a_off[ms]);
                a[0][ms] = *(const f16x8 *)(buf + a_off[ms] + tap_off(0, BK_PS));
            }
            const unsigned xn = (unsigned)((chunk + 1) * BK_CK * 2) - x_lo;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int cur = u & 1, slot = u % RU;
                const bool more = u + 1 < NU;
                const char *nfrag = buf + tap_off((u + 1) % NT, BK_PS) + 32 * ((u + 1) / NT);
                const int ur = u - 1 + RU;
                const _Float16 *rw = (LAST && ur >= NU) ? w_unit(p.wb, 0, ur - NU)
                                                        : w_unit(p.wa, ur >= NU ? chunk + 1 : chunk, ur >= NU ? ur - NU : ur);
                const int rslot = (u + RU - 1) % RU;
#pragma unroll
                for (int q = 0; q < NMFA; ++q) {
                    const int ms = q / NSUB, ns = q % NSUB;
                    acc[ms][ns] = ds_mfma_32x32x16_f16(bq[slot][ns], a[cur][ms], (FIRST && u == 0) ? zero16 : acc[ms][ns]);
                    if (q & 1) {
                        const int lm = q >> 1;
#ifndef DS_ABL_NO_AFRAG
                        if (lm < MSA && more) a[cur ^ 1][lm] = *(const f16x8 *)(nfrag + a_off[lm]);
#endif
                    } else {
                        const int e = q >> 1;
                        if (e < NSUB) {
#ifndef DS_ABL_NO_REFILL
                            bq[rslot][e] = *(const f16x8 *)(rw + (size_t)e * 32 * 16);
#endif
                        } else if constexpr (!LAST) {
#ifndef DS_ABL_NO_STAGE
                            const int s = e - NSUB;
                            if (u < UL) {
                                const int it = u * SPU + s;
                                if (it < NIT) st[it] = ds_buffer_load_f32x4(xbuf, g_rel[it] + xn);
                            } else if (u >= NU - UL) {
                                const int it = (u - (NU - UL)) * SPU + s;
                                if (it < NIT) *(f32x4 *)(obuf + l_off[it]) = st[it];
                            }
#endif
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        run_chunk(std::true_type{}, std::false_type{}, 0, lds, lds + tileA_bytes);
        ds_lds_barrier();
        for (int i = 1; i + 1 < n_chunks; ++i) {
            char *b0 = lds + (i & 1) * tileA_bytes, *b1 = lds + ((i & 1) ^ 1) * tileA_bytes;
            run_chunk(std::false_type{}, std::false_type{}, i, b0, b1);
            ds_lds_barrier();
        }
        run_chunk(std::false_type{}, std::true_type{}, n_chunks - 1, lds + ((n_chunks - 1) & 1) * tileA_bytes, lds);
        DS_BLK_STAMP(2);

        // ---- (3) hand-over: bn1 + clip, rounded to fp16, into the intermediate tile [ROWS_A][pitch] of C-channel
        // records.  Columns 0 and W + 1 and the rows outside the image are the second convolution's zero padding. ----
        {
            f32x4 sca[NSUB][4], sha[NSUB][4];
#pragma unroll
            for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = n_base + ns * 32 + 8 * g + 4 * lhi;
                    sca[ns][g] = *(const f32x4 *)(p.sa + c0);
                    sha[ns][g] = *(const f32x4 *)(p.ha + c0);
                }
            ds_lds_barrier();                       // every wave is done reading the input tiles
            const int ppr = RSB / 16;               // 16-byte pieces per record
            for (int i = tid; i < ROWS_A * 2 * ppr; i += NTHR) {
                const int piece = i % ppr, rec = i / ppr;
                *(f32x4 *)(lds + ((rec >> 1) * pitch + (rec & 1) * (W + 1)) * RSB + 16 * piece) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int ms = 0; ms < MSA; ++ms) {
                const int m = (wm * MSA + ms) * 32 + lpix;
                const int i = m >> WSH, c = m & (W - 1);
                const int row = r0 - 1 + i;                              // image row of this intermediate pixel
                const unsigned keep = (row >= 0 && row < h_valid) ? 0xFFFFFFFFu : 0u;   // outside the image: zero padding
                char *rec = lds + (i * pitch + c + 1) * RSB;
#pragma unroll
                for (int ns = 0; ns < NSUB; ++ns)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // two channels per v_pk_fma_f32 (the same fma per channel as the unfused epilogue), clip in f32,
                        // one packed conversion per pair
                        ds_u32x2 hb;
#pragma unroll
                        for (int j2 = 0; j2 < 2; ++j2) {
                            const ds_f32x2 v = {acc[ms][ns][4 * g + 2 * j2], acc[ms][ns][4 * g + 2 * j2 + 1]};
                            const ds_f32x2 sc2 = {sca[ns][g][2 * j2], sca[ns][g][2 * j2 + 1]};
                            const ds_f32x2 sh2 = {sha[ns][g][2 * j2], sha[ns][g][2 * j2 + 1]};
                            ds_f32x2 t = v * sc2 + sh2;          // (contracted to one fma per channel, like the scalar form)
                            t[0] = fminf(fmaxf(t[0], 0.0f), 20.0f);
                            t[1] = fminf(fmaxf(t[1], 0.0f), 20.0f);
                            hb[j2] = __builtin_bit_cast(unsigned, __builtin_convertvector(t, ds_f16x2)) & keep;
                        }
                        *(ds_u32x2 *)(rec + (n_base + ns * 32 + 8 * g + 4 * lhi) * 2) = hb;
                    }
            }
            ds_lds_barrier();                       // the intermediate tile is complete
        }
        DS_BLK_STAMP(3);

        // ---- (4) second convolution: fragments straight from the intermediate tile, all chunks resident.  The ring
        // runs on into the next tile's first layer; the LAST chunk's spare slots request the residual rows of the
        // epilogue (the block's own input, staged a moment ago: L2-hot). ----
        const int lin_base = (b * p.H + r0) * W;
        const int lin_valid = (p.H - r0 < BK_R ? p.H - r0 : BK_R) * W;
        f32x4 resv[MSB][NRI];
        {
            const int total = n_chunks * NU;        // units of the whole contraction, filter ring running through
            int b_off[MSB];
#pragma unroll
            for (int ms = 0; ms < MSB; ++ms) {
                const int m = (wm * MSB + ms) * 32 + lpix;
                b_off[ms] = ((m >> WSH) * pitch + (m & (W - 1))) * RSB + 16 * lhi;
            }
            f16x8 a[2][MSB];
#pragma unroll
            for (int ms = 0; ms < MSB; ++ms) {
                DS_OPAQUE_VGPR(b_off[ms]);
                a[0][ms] = *(const f16x8 *)(lds + b_off[ms] + tap_off(0, RSB));
            }
            __builtin_amdgcn_sched_barrier(0);
            // LASTB is a compile-time property: a load behind a run-time condition would be branched around and waited
            // for on the spot
            auto run_b = [&](auto first_tag, auto last_tag, int chunk) __attribute__((always_inline)) {
                constexpr bool FIRSTB = decltype(first_tag)::value, LASTB = decltype(last_tag)::value;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int cur = u & 1, slot = u % RU;
                    const int gu = chunk * NU + u;                       // (NU is even and a multiple of RU: slots line up)
                    const bool more = gu + 1 < total;
                    const int nu = (u + 1) % NU, nchunk = (u + 1 < NU) ? chunk : chunk + 1;
                    // (past the last unit the "next" fragment is read again from this chunk: valid address, unused value)
                    const char *nfrag = lds + tap_off(nu % NT, RSB) + 32 * (nu / NT) + (more ? nchunk : chunk) * 64;
                    const int ur = u - 1 + RU;
                    const int rchunk = ur >= NU ? chunk + 1 : chunk;
                    // past the end of this layer the ring runs on into the next tile's first layer (chunk 0 of wa)
                    const _Float16 *rw = rchunk < n_chunks ? w_unit(p.wb, rchunk, ur >= NU ? ur - NU : ur)
                                                           : w_unit(p.wa, 0, ur >= NU ? ur - NU : ur);
                    const int rslot = (u + RU - 1) % RU;
#pragma unroll
                    for (int q = 0; q < NMFB; ++q) {
                        const int ms = q / NSUB, ns = q % NSUB;
                        acc[ms][ns] = ds_mfma_32x32x16_f16(bq[slot][ns], a[cur][ms], (FIRSTB && u == 0) ? zero16 : acc[ms][ns]);
                        if (q & 1) {
                            const int lm = q >> 1;
#ifndef DS_ABL_NO_AFRAG
                            if (lm < MSB) a[cur ^ 1][lm] = *(const f16x8 *)(nfrag + b_off[lm]);
#endif
                        } else {
                            const int e = q >> 1;
                            if (e < NSUB) {
#ifndef DS_ABL_NO_REFILL
                                bq[rslot][e] = *(const f16x8 *)(rw + (size_t)e * 32 * 16);
#endif
                            } else if constexpr (LASTB) {                // residual rows -> registers, units NU-2-ULR ..
                                const int ri = (u - (NU - 2 - ULR)) * SPUB + (e - NSUB);
                                if (u >= NU - 2 - ULR && ri < NRES) {
                                    const int rms = ri / NRI, rk = ri % NRI;
                                    const int m = (wm * MSB + rms) * 32 + rk * PPI + my_p;
#ifdef DS_ABL_NO_RES
                                    resv[rms][rk] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
                                    resv[rms][rk] = ds_buffer_load_f32x4(rbuf, m < lin_valid ? (unsigned)((lin_base + m) * C + col) * 2u
                                                                                            : DS_BUFFER_OOB);
#endif
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            };
            run_b(std::true_type{}, std::false_type{}, 0);
            for (int chunk = 1; chunk + 1 < n_chunks; ++chunk) run_b(std::false_type{}, std::false_type{}, chunk);
            run_b(std::false_type{}, std::true_type{}, n_chunks - 1);
        }

        DS_BLK_STAMP(4);
        // ---- (5) epilogue of the block: bn2 + residual (the block's own input) + clip, as conv_mfma_f16_kernel.  At
        // its head the NEXT tile's first input chunk is requested: it arrives while this tile is written out. ----
        if (tid == 0) *sched_word = t_drawn;
        ds_lds_barrier();                           // every wave is done reading the intermediate tile
        DS_BLK_STAMP(5);
        const int t_after = ds_uniform(*sched_word);
        // bn2's scale / shift rows are requested BEFORE the next tile's pixels: loads retire through one in-order counter,
        // and the epilogue's first fma would otherwise wait for the whole HBM prefetch issued ahead of them (round 6:
        // `s_waitcnt vmcnt(0)` behind 12 prefetch + 4 table loads in the ISA of the round-5 kernel)
#ifndef DS_EPI_TABLES_LATE          // (A/B builds, tools/f16_ab.py: the round-5 order -- prefetch first)
        const f32x4 sc[2] = {*(const f32x4 *)(p.sb + col), *(const f32x4 *)(p.sb + col + 4)};
        const f32x4 sh[2] = {*(const f32x4 *)(p.hb + col), *(const f32x4 *)(p.hb + col + 4)};
        __builtin_amdgcn_sched_barrier(0);
#endif
        int nb = b, nr0 = r0;
        {
            const bool has_next = t_next < t_end;
            if (has_next) {
                tile_of(t_next, nb, nr0);
                stage_window(nb, nr0, xbuf, x_lo);
            } else {
                xbuf = ds_make_buffer(p.x, 0u);     // nothing in range: the loads return zeros
                x_lo = 0;
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) st[it] = ds_buffer_load_f32x4(xbuf, g_rel[it] - x_lo);
        }
        const int lin_kept = (h_valid - r0) * W;                          // MASKED: pixels of the tile below the extent
#ifdef DS_EPI_TABLES_LATE
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 sc[2] = {*(const f32x4 *)(p.sb + col), *(const f32x4 *)(p.sb + col + 4)};
        const f32x4 sh[2] = {*(const f32x4 *)(p.hb + col), *(const f32x4 *)(p.hb + col + 4)};
#endif
        auto put_tile = [&](int ms) {
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
        // element offset of (pixel, first channel) = pixel * y_mul + y_add: channels-last, or 16-channel planes
        const unsigned y_mul = p.y_plane_stride ? 16u : (unsigned)C;
        const unsigned y_add = p.y_plane_stride ? (unsigned)(col >> 4) * p.y_plane_stride + (unsigned)(col & 15) : (unsigned)col;
        // one pass over the sub-tiles per output type (the choice is made once per tile, not once per store)
        auto write_out = [&](auto f32_tag) __attribute__((always_inline)) {
            constexpr bool OUT32 = decltype(f32_tag)::value;
            put_tile(0);
#pragma unroll
            for (int ms = 0; ms < MSB; ++ms) {
                const int cb = ms & 1;
                ds_wave_sync();
                const float *src = tb + cb * (32 * TP);
                f32x4 tv[NRI][2];
#pragma unroll
                for (int k = 0; k < NRI; ++k)
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq) tv[k][hq] = *(const f32x4 *)(src + (k * PPI + my_p) * TP + my_c + 4 * hq);
                if (ms + 1 < MSB) put_tile(ms + 1);
#pragma unroll
                for (int k = 0; k < NRI; ++k) {
                    const f16x8 r8 = __builtin_bit_cast(f16x8, resv[ms][k]);
                    const int m = (wm * MSB + ms) * 32 + k * PPI + my_p;
                    const float keep = (!MASKED || m < lin_kept) ? 20.0f : 0.0f;     // MASKED rows past the extent: clip to [0, 0]
                    f32x4 o[2];
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq)
#pragma unroll
                        for (int j2 = 0; j2 < 2; ++j2) {
                            // two channels per packed instruction: fma (bn2), add (residual), then the clip per channel
                            const ds_f32x2 v = {tv[k][hq][2 * j2], tv[k][hq][2 * j2 + 1]};
                            const ds_f32x2 s2 = {sc[hq][2 * j2], sc[hq][2 * j2 + 1]}, h2 = {sh[hq][2 * j2], sh[hq][2 * j2 + 1]};
                            const ds_f16x2 rh = {r8[4 * hq + 2 * j2], r8[4 * hq + 2 * j2 + 1]};
                            ds_f32x2 t = v * s2 + h2;
                            t = t + __builtin_convertvector(rh, ds_f32x2);
                            o[hq][2 * j2] = fminf(fmaxf(t[0], 0.0f), keep);
                            o[hq][2 * j2 + 1] = fminf(fmaxf(t[1], 0.0f), keep);
                        }
                    const unsigned vo = m < lin_valid ? (unsigned)(lin_base + m) * y_mul + y_add : DS_BUFFER_OOB;
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
#ifdef DS_ABL_NO_STORE
                        if (hb[0] == 0x12345678u)               // (keeps the values live)
#endif
                        ds_buffer_store_out_f32x4(ybuf, vo != DS_BUFFER_OOB ? vo * 2u : DS_BUFFER_OOB, __builtin_bit_cast(f32x4, hb));
                    }
                }
            }
        };
        if (out32) write_out(std::true_type{});
        else write_out(std::false_type{});
        DS_BLK_STAMP(6);
        b = nb;
        r0 = nr0;
        t_cur = t_next;
        t_next = t_after;
    }
    // the last workgroup to leave hands the slot back zeroed (no counter is touched after a workgroup's own `done`)
    if (tid == 0 && ds_atomic_inc(p.sched + DS_SCHED_DONE) == gridDim.x - 1) {
        for (int i = 0; i <= DS_SCHED_DONE; ++i) p.sched[i] = 0u;
    }
}

static size_t block_lds_bytes(int W, int C, int waves) {
    const size_t tileA = (size_t)(BK_R + 4) * (W + 2) * BK_PS;
    const size_t interm = (size_t)(BK_R + 2) * (W + 2) * (C * 2 + 16);
    const size_t epi = (size_t)2 * waves * 32 * (2 * 32 + 4) * 4;
    return std::max(std::max(2 * tileA, interm), epi) + 64;
}

}  // namespace

#ifdef DS_F16_PROBE
static long long *g_blk_probe = nullptr;
extern "C" void ds_block_set_probe(long long *buf) { g_blk_probe = buf; }
#endif


// 1 if ds_conv_block_f16 handles this block geometry (the shallow stages: W = 32 with 64 channels, W = 16 with 128)
extern "C" int ds_conv_block_f16_supported(int B, int H, int W, int C) {
    if (B <= 0 || H <= 0) return 0;
    if ((long long)B * H * W * C >= (1ll << 30)) return 0;
    return (W == 32 && C == 64) || (W == 16 && C == 128);
}

// y = clip(bn2(conv3x3(clip(bn1(conv3x3(x))))) + x) for one BasicBlock in eval mode (reference model.py:66-82 with
// module.eval()); x, y fp16 channels-last [B,H,W,C]; wa / wb from ds_pack_conv_weight_f16; flags: DS_EPI_OUT_F32,
// DS_EPI_OUT_PLANES16.  Bit-identical to two ds_conv_fwd_f16 calls.
static int conv_block_f16(const void *x_f16, const void *wa_f16, const void *wb_f16, const float *scale_a,
                          const float *shift_a, const float *scale_b, const float *shift_b, void *y, const int *lens,
                          int B, int H, int W, int C, int flags, void *stream) {
    DS_REQUIRE(x_f16 && wa_f16 && wb_f16 && scale_a && shift_a && scale_b && shift_b && y, DS_ERR_NULL);
    DS_REQUIRE(ds_conv_block_f16_supported(B, H, W, C), DS_ERR_UNSUPPORTED);
    DS_REQUIRE(DS_ALIGNED16(x_f16) && DS_ALIGNED16(wa_f16) && DS_ALIGNED16(wb_f16) && DS_ALIGNED16(y) &&
                   DS_ALIGNED16(scale_a) && DS_ALIGNED16(shift_a) && DS_ALIGNED16(scale_b) && DS_ALIGNED16(shift_b),
               DS_ERR_ALIGNMENT);
    DS_REQUIRE(!((flags & DS_EPI_OUT_F32) && (flags & DS_EPI_OUT_PLANES16)), DS_ERR_UNSUPPORTED);
    BlockK k;
    k.x = (const _Float16 *)x_f16; k.wa = (const _Float16 *)wa_f16; k.wb = (const _Float16 *)wb_f16;
    k.sa = scale_a; k.ha = shift_a; k.sb = scale_b; k.hb = shift_b; k.y = y;
    k.B = B; k.H = H; k.W = W; k.C = C;
    k.tiles_per_img = ds_ceil_div(H, BK_R);
    k.n_tiles = B * k.tiles_per_img;
    k.flags = flags;
    const long long n = (long long)B * H * W * C;
    k.x_bytes = (unsigned)(n * 2);
    k.y_bytes = (unsigned)(n * ((flags & DS_EPI_OUT_F32) ? 4 : 2));
    k.y_plane_stride = (flags & DS_EPI_OUT_PLANES16) ? (unsigned)((long long)B * H * W * 16) : 0u;
    k.lens = lens;
#ifdef DS_F16_PROBE
    k.probe = g_blk_probe;
#endif
    // persistent workgroups: as many as the chip holds at once (two 2-wave workgroups per CU: one wave per SIMD, 65 KB
    // of LDS each); batches of >= 8 images that fill them are dealt to the XCDs by image (see the kernel)
    const int resident = 2 * ds_cu_count();
    int grid = k.n_tiles < resident ? k.n_tiles : resident;
    k.xcd_slots = 0;
    if (B >= 8 && grid == resident && resident % 8 == 0) k.xcd_slots = resident / 8;
    k.sched = ds_sched_slot(stream);
    DS_REQUIRE(k.sched != nullptr, DS_ERR_NO_WORKSPACE);     // ds_sched_set_workspace is due
    k.sched_lds = (int)block_lds_bytes(W, C, 2) - 16;
    if (C == 64) {          // W = 32: 2 x 1 waves, 12 x 32 x 4 items over 128 threads
        if (lens) DS_LAUNCH_BIG_LDS((conv_block3x3_f16_kernel<2, 1, 12, true>), grid, 128, block_lds_bytes(W, C, 2), stream, k);
        else DS_LAUNCH_BIG_LDS((conv_block3x3_f16_kernel<2, 1, 12>), grid, 128, block_lds_bytes(W, C, 2), stream, k);
    } else {                // W = 16, C = 128: 1 x 2 waves
        if (lens) DS_LAUNCH_BIG_LDS((conv_block3x3_f16_kernel<1, 2, 6, true>), grid, 128, block_lds_bytes(W, C, 2), stream, k);
        else DS_LAUNCH_BIG_LDS((conv_block3x3_f16_kernel<1, 2, 6>), grid, 128, block_lds_bytes(W, C, 2), stream, k);
    }
    return ds_last_launch_error();
}

extern "C" int ds_conv_block_f16(const void *x_f16, const void *wa_f16, const void *wb_f16, const float *scale_a,
                                 const float *shift_a, const float *scale_b, const float *shift_b, void *y, int B, int H,
                                 int W, int C, int flags, void *stream) {
    return conv_block_f16(x_f16, wa_f16, wb_f16, scale_a, shift_a, scale_b, shift_b, y, nullptr, B, H, W, C, flags, stream);
}

// The same block over a zero-padded batch of utterances of different lengths: `lens` (device, int32 [B]) = the rows of
// each image that belong to its utterance; rows past them are zero in the intermediate and in the output, exactly as
// ds_mask_rows leaves them after each layer of the unfused sequence.
extern "C" int ds_conv_block_f16_masked(const void *x_f16, const void *wa_f16, const void *wb_f16, const float *scale_a,
                                        const float *shift_a, const float *scale_b, const float *shift_b, void *y,
                                        const int *lens, int B, int H, int W, int C, int flags, void *stream) {
    DS_REQUIRE(lens != nullptr, DS_ERR_NULL);
    return conv_block_f16(x_f16, wa_f16, wb_f16, scale_a, shift_a, scale_b, shift_b, y, lens, B, H, W, C, flags, stream);
}
