/*
 * deepspeaker_hip.h -- C ABI of libdeepspeaker_hip.so (MI355X / gfx950).
 *
 * The reference (qqueing/DeepSpeaker-pytorch) is pure Python: it has no FFI or
 * plugin interface.  Its hot path is the Python surface of model.py as used by
 * train_triplet.py; every tensor op that surface executes
 * through ATen/cuDNN is replaced here by one entry point.  Each declaration
 * cites the reference line(s) whose arithmetic it takes over.
 *
 * Conventions (all entry points):
 *   - plain device pointers and sizes, no torch types; `stream` is a hipStream_t
 *     passed as void* (the caller's current stream).  Work is only enqueued: no
 *     entry point synchronises the device or a stream (ds_event_elapsed_ms, a
 *     read-out helper, waits for its `stop` event), so calls are re-entrant and
 *     hipGraph-capturable.
 *   - the caller owns ALL device memory: tensor buffers (inputs, outputs, scratch)
 *     and the tile-scheduling workspace of the persistent fp16 kernels
 *     (ds_sched_workspace_bytes / ds_sched_set_workspace below).  The library never
 *     allocates, frees, memsets or synchronises.  State the LIBRARY keeps -- all of
 *     it host-side, each item behind its own lock or thread-local:
 *       (1) the table of which 64-byte slots of the caller's scheduler workspaces
 *           have been handed out: a ring of 8 per (device, stream) for eager
 *           launches, one slot for good per launch captured into a graph
 *           (csrc/bn_pack.hip ds_sched_slot); a launch that finds no free slot
 *           returns DS_ERR_NO_WORKSPACE;
 *       (2) the armed event pair of ds_launch_timing_arm: thread-local, consumed by
 *           the calling thread's next MFMA launch (csrc/ds_device.h);
 *       (3) process-wide tuning hooks, ds_conv_f16_set_layout_padding and
 *           ds_conv_{f16,bf16}_set_forced_cfg (off by default; A/B hooks that change speed,
 *           never results);
 *       (4) the cached compute-unit count per device (read once).
 *     Nothing else is kept between calls.
 *   - return value: 0 = DS_OK, negative = argument error (below), positive = the
 *     hipError_t of a failed launch.  Nothing throws, aborts or prints.
 *   - activations are channels-last fp32:  [B, T', F', C]  (the reference's NCHW
 *     tensors [B, C, T', F'] permuted; for the network input C = 1, so the
 *     reference's [B,1,T,64] buffer is consumed bit-for-bit as [B,T,64,1]).
 *     ds_nchw_to_nhwc_f32 / ds_nhwc_to_nchw_f32 convert for per-op use.
 *   - convolution weights are consumed in a packed layout produced once per
 *     weight version by ds_pack_* from the reference's OIHW / [out,in] tensors.
 *   - pointers must be 16-byte aligned (every torch allocation is).
 */
#ifndef DEEPSPEAKER_HIP_H
#define DEEPSPEAKER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS_OK               0
#define DS_ERR_BAD_SHAPE   (-1)
#define DS_ERR_ALIGNMENT   (-2)
#define DS_ERR_NULL        (-3)
#define DS_ERR_UNSUPPORTED (-4)
#define DS_ERR_NO_WORKSPACE (-5)    /* a persistent launch found no free tile-scheduling slot: ds_sched_set_workspace */

/* epilogue flags of the convolution entry points */
#define DS_EPI_AFFINE   1   /* y = acc * scale[c] + shift[c]   (BatchNorm with fixed statistics) */
#define DS_EPI_RESIDUAL 2   /* y += residual                    (model.py:79)                      */
#define DS_EPI_CLIP     4   /* y = min(max(y, 0), 20)           (model.py:36-44)                   */
#define DS_EPI_STATS    8   /* also emit per-tile column sums {sum, sum of squares} of the RAW
                               accumulator (train-mode BatchNorm statistics)                       */

#define DS_EPI_OUT_F32  16  /* fp16 convolution: store the result as f32 (the layer feeding the f32 tail)     */
#define DS_EPI_OUT_F16  32  /* conv1 (ds_conv5x5s2_c1_fwd_bf16): store the result as fp16                    */

#define DS_EPI_OUT_PLANES16  256 /* fp16 convolution: store y channel-plane-major, [Cout/16][B*Ho*Wo][16] fp16          */
#define DS_CONV_IN_PLANES16  512 /* fp16 5x5 convolution: x is channel-plane-major [Cin/16][B*H*W][16]: every
                                    16-channel chunk then reads whole 128-byte lines (a 64-channel channels-last
                                    record is one line, of which a chunk would use a quarter)                       */
#define DS_CONV_HINT_SINGLE_BUFFER 64  /* fp16 convolution: plan with one LDS pixel tile (tuning / test hint)  */
#define DS_CONV_HINT_CHUNK16      128  /* fp16 5x5 convolution: plan with 16-channel chunks (tuning / test hint)  */
#define DS_CONV_HINT_NO_WIDE     2048  /* fp16 convolution: keep 64-channel-wide register tiles where the persistent kernel
                                        * would take 128-wide ones (tuning / test hint: bit-identical)                */
#define DS_CONV_HINT_ONE_QUEUE   4096  /* fp16 convolution, persistent kernel: one tile queue for the whole grid instead
                                        * of one per XCD (tuning / test hint: bit-identical)                         */
#define DS_CONV_HINT_NO_PERSIST  1024  /* fp16 convolution: one tile per workgroup even where the persistent kernel
                                        * applies (tuning / test hint: the two kernels are bit-identical)          */

#define DS_CONV_CK      8   /* input-channel chunk of the packed weight layout */

int ds_version(void);            /* 100: round 1; 200: round 2 (fp16 path, grouped BatchNorm backward, ds_bn_bwd_partial_rows takes C);
                                    300 / 301: round 3 (persistent fp16 kernels, split grouped BatchNorm backward; + ds_conv_dgrad_bnbwd_bf16);
                                    400: round 4 (fp16 training step: ds_*_f16 train entry points, ds_wgrad_f16, probes of the
                                    near-tie refinement, launch-bound timing); 500: round 5 */
/* launch timing without marker packets: the next MFMA convolution / filter-gradient launch of the calling thread
 * records its own execution into the armed pair (hipExtLaunchKernelGGL); ds_launch_timing_end() disarms and returns
 * how many such launches happened since arming (1 = the timed call was a single kernel) */
int ds_event_create(void **out_event);
int ds_event_destroy(void *event);
int ds_event_elapsed_ms(void *start, void *stop, float *ms);     /* waits for `stop` */
int ds_launch_timing_arm(void *start, void *stop);
int ds_launch_timing_end(void);
const char *ds_error_string(int code);

/* ---- tile-scheduling workspace of the persistent kernels (ds_conv_fwd_f16, ds_conv_block_f16) ----
 * Those kernels draw tiles from device-side counters ("slots", 64 bytes each, left zeroed by the kernel that used them).
 * The memory is the caller's: hand over ZEROED device memory of ds_sched_workspace_bytes() on the current device before
 * the first such launch there (again whenever a launch returns DS_ERR_NO_WORKSPACE: every launch captured into a graph
 * keeps one slot for good) and keep it allocated.  ds_sched_free_slots(): slots of the current device not handed out yet. */
size_t ds_sched_workspace_bytes(void);
int ds_sched_set_workspace(void *zeroed_device_memory, size_t bytes);
long long ds_sched_free_slots(void);
/* measurement only: one launch of independent fp16 (bf16 != 0: bf16) 32x32x16 MFMAs issued back to back from registers on
 * every SIMD, random operand bits -- the matrix cores' power- / clock-limited rate on THIS chip.  *flop_out receives the
 * floating-point operations of the launch; the caller times it with its own events (bench.py:
 * roofline.mfma_register_only_tflops).  Not on any product path. */
int ds_mfma_rate_probe(int bf16, int iters, float *sink, double *flop_out, void *stream);
/* the same launch with fp16 operands taken from REAL tensors: A fragments (8 consecutive halfs each) from `a_f16` (n_a
 * halfs, e.g. a packed filter bank), B fragments from `b_f16` (n_b halfs, e.g. an fp16 activation tensor); 16-byte
 * aligned.  What the matrix cores sustain on the operand values the convolutions see (bench.py:
 * roofline.mfma_register_only_real_operands_tflops). */
int ds_mfma_rate_probe_data(const void *a_f16, long long n_a, const void *b_f16, long long n_b, int iters, float *sink,
                            double *flop_out, void *stream);

/* ---- layout ---------------------------------------------------------------------------------- */
int ds_nchw_to_nhwc_f32(const float *x, float *y, int B, int C, int H, int W, void *stream);
int ds_nhwc_to_nchw_f32(const float *x, float *y, int B, int C, int H, int W, void *stream);

/* ---- weight packing (one-off per weight version; not on the per-batch path) ------------------- */
/* Many filters in ONE launch (a training step re-packs every filter after every optimizer step: 22 launches of ~5 us
 * each in the fp16 mode, 19 in the bf16x3 mode, became one each).  A job = one ds_pack_conv_weight_{f16,bf16} call:
 * mode 0 forward bank, 1 stride-1 data-gradient bank (transposed, flipped), 2 (f16 only) the 5x5 stride-2 layer's
 * parity-class data-gradient bank.  out2: the bf16 "lo" bank (bf16 family; may be NULL), unused by the f16 family.
 * At most DS_PACK_BATCH_MAX jobs per call. */
#define DS_PACK_BATCH_MAX 32
typedef struct ds_pack_job {
    const float *w_oihw;
    void *out, *out2;
    int Cout, Cin, KS, mode;
} ds_pack_job;
int ds_pack_conv_weights_f16_batch(const ds_pack_job *jobs, int n_jobs, void *stream);
int ds_pack_conv_weights_bf16_batch(const ds_pack_job *jobs, int n_jobs, void *stream);
/* OIHW [Cout,Cin,KS,KS] -> [Cin/8][KS*KS][Cout][8].  dgrad != 0 packs the transposed, spatially
 * flipped filter bank used by ds_conv_dgrad_f32 (Cout and Cin swap roles).
 * Replaces: nothing in the reference (cuDNN picks its own filter layout, model.py:47-50,93-106). */
int ds_pack_conv_weight_f32(const float *w_oihw, float *w_packed, int Cout, int Cin, int KS,
                            int dgrad, void *stream);
/* conv1 weight [64,1,5,5] -> [25][64]  (model.py:93) */
int ds_pack_conv1_weight_f32(const float *w_oihw, float *w_packed, int Cout, void *stream);
/* fc weight [N, C*F] indexed c*F+f (model.py:164,208) -> 1x1-conv packing over k' = f*C + c,
 * the order in which ds_avgpool_time_f32 emits the pooled features. */
int ds_pack_fc_weight_f32(const float *w, float *w_packed, int N, int C, int F, void *stream);
/* the transposed bank for the fc data gradient (gpooled = gf . W), same k' order */
int ds_pack_fc_weight_dgrad_f32(const float *w, float *w_packed, int N, int C, int F, void *stream);

/* ---- BatchNorm ---------------------------------------------------------------------------------- */
/* eval mode: scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale
 * (model.py:70,74,188,193,198,203 with module.eval()). */
int ds_bn_fold_f32(const float *gamma, const float *beta, const float *running_mean,
                   const float *running_var, float eps, float *scale, float *shift, int C,
                   void *stream);
/* train mode: reduce the per-tile partial sums written under DS_EPI_STATS, produce batch mean /
 * invstd, the affine (scale, shift) that normalises with them, and update the running statistics
 * in place (momentum, unbiased variance) exactly like nn.BatchNorm2d.train() -- SURVEY 8(a) a2. */
int ds_bn_stats_finalize_f32(const float *partial, int n_partial, long long count,
                             const float *gamma, const float *beta, float eps, float momentum,
                             float *running_mean, float *running_var, float *batch_mean,
                             float *batch_invstd, float *scale, float *shift, int C, void *stream);
/* y = [clip]( x * scale[c] + shift[c] [+ residual] ) over n_pix pixels of C channels. */
int ds_bn_apply_f32(const float *x, const float *scale, const float *shift, const float *residual,
                    float *y, long long n_pix, int C, int flags, void *stream);

/* ---- convolutions ------------------------------------------------------------------------------- */
typedef struct ds_conv_shape {
    int B, H, W, Cin;   /* input  [B,H,W,Cin]                        */
    int Cout, KS;       /* square KS x KS filter, padding KS/2       */
    int stride;         /* 1 or 2                                    */
} ds_conv_shape;

int ds_conv_out_dims(const ds_conv_shape *s, int *Ho, int *Wo);
/* number of [Cout][2] partial-statistics rows DS_EPI_STATS writes for this shape */
int ds_conv_stats_rows(const ds_conv_shape *s);

/* tiling the planner picks for a shape: out8 = {M tile, N tile, rows per segment, segments per tile,
 * workgroups, LDS bytes, staging slots per thread, M tiles} (introspection for tests / DESIGN.md) */
int ds_conv_plan_describe(const ds_conv_shape *s, int *out8);
int ds_conv5x5s2_c1_stats_rows(int B, int H);
/* conv1: 5x5 stride 2 pad 2, Cin = 1 -> 64 channels (model.py:93,187) fused with the following
 * BatchNorm affine + clipped ReLU (model.py:188-189).  x is the network input [B,H,W]. */
int ds_conv5x5s2_c1_fwd_f32(const float *x, const float *w_packed, const float *scale,
                            const float *shift, float *y, float *stats_partial, int B, int H,
                            int W, int Cout, int flags, void *stream);
/* the same layer on the bf16 matrix cores with split operands (bf16x3); same packed bank and statistics rows.
 * y is f32 [B,Ho,Wo,64], or fp16 with DS_EPI_OUT_F16 (the input of the fp16 convolution path) */
int ds_conv5x5s2_c1_fwd_bf16(const float *x, const float *w_packed, const float *scale,
                             const float *shift, void *y, float *stats_partial, int B, int H,
                             int W, int Cout, int flags, void *stream);
/* generic implicit-GEMM convolution on the f32 matrix cores: 3x3 s1 p1 (model.py:47-50,69,73),
 * 5x5 s2 p2 (model.py:98,102,106 / :192,197,202) and 1x1 (the fc GEMM, model.py:209), with the
 * BatchNorm / residual / clipped-ReLU epilogue selected by `flags`. */
int ds_conv_fwd_f32(const ds_conv_shape *s, const float *x, const float *w_packed,
                    const float *scale, const float *shift, const float *residual, float *y,
                    float *stats_partial, int flags, void *stream);

/* ---- bf16 matrix-core variants of the forward convolutions (v_mfma_f32_32x32x16_bf16, f32 accumulate;
 *      same epilogue flags, f32 channels-last activations in and out -> drop-in for ds_conv_fwd_f32).
 *      w_lo == NULL: plain bf16 operands (speed mode, ~6e-3 embedding drift on this network);
 *      w_lo != NULL: "bf16x3" -- both operands split hi+lo, product = hi*hi + hi*lo + lo*hi: f32-class
 *      accuracy at up to 1/3 of the bf16 peak.  KS in {3, 5}; Cin % 16 == 0. ----------------------- */
int ds_pack_conv_weight_bf16(const float *w_oihw, void *w_hi, void *w_lo, int Cout, int Cin, int KS,
                             void *stream);             /* -> [Cin/16][KS*KS][Cout][16] bf16 (x2) */
int ds_pack_conv_weight_dgrad_bf16(const float *w_oihw, void *w_hi, void *w_lo, int Cout, int Cin,
                                   int KS, void *stream);   /* flipped / transposed bank, stride 1 */
/* the four parity-class banks of the 5x5 stride-2 data gradient: 36 * Cout * Cin bf16 each (hi, lo) */
int ds_pack_conv_weight_dgrad_s2_bf16(const float *w_oihw, void *w_hi, void *w_lo, int Cout, int Cin,
                                      void *stream);
/* data gradient: 3x3 stride 1 (banks of ds_pack_conv_weight_dgrad_bf16) or 5x5 stride 2 (banks of
 * ds_pack_conv_weight_dgrad_s2_bf16; `s` is the FORWARD layer shape in both cases) */
int ds_conv_dgrad_bf16(const ds_conv_shape *s, const float *gy, const void *w_hi, const void *w_lo,
                       float *gx, void *stream);
int ds_conv_bf16_stats_rows(const ds_conv_shape *s, int x3);
/* tiling the bf16 planner picks: out8 = {M tile, N tile, rows per segment, segments per tile, workgroups,
 * LDS bytes, threads per workgroup, LDS row pitch in pixel records} */
int ds_conv_bf16_plan_describe(const ds_conv_shape *s, int x3, int *out8);
void ds_conv_bf16_set_forced_cfg(int cfg);  /* tuning hook, as ds_conv_f16_set_forced_cfg */
int ds_conv_fwd_bf16(const ds_conv_shape *s, const float *x, const void *w_hi, const void *w_lo,
                     const float *scale, const float *shift, const float *residual, float *y,
                     float *stats_partial, int flags, void *stream);

/* ---- fp16 matrix cores: the eval-forward throughput path ------------------------------------------
 * Same layers and epilogue as ds_conv_fwd_f32 (model.py:69-80,192-205 in module.eval()), computed with
 * v_mfma_f32_32x32x16_f16 (fp16 operands, f32 accumulate) on fp16 channels-last activations
 * [B,H,W,C].  One MFMA per product: the embedding lands 3.7e-4 from the reference (contract 1e-3).
 * x, residual, y are fp16 buffers (y is f32 with DS_EPI_OUT_F32); DS_EPI_STATS is not supported. */
/* OIHW f32 -> [Cin/16][KS*KS][Cout][16] fp16 (round to nearest even) */
int ds_pack_conv_weight_f16(const float *w_oihw, void *w_f16, int Cout, int Cin, int KS, void *stream);
int ds_conv_fwd_f16(const ds_conv_shape *s, const void *x_f16, const void *w_f16, const float *scale,
                    const float *shift, const void *residual_f16, void *y, int flags, void *stream);
/* The same for SMALL launches (serving latency, reference model.py:185-218 on one or a few utterances): when the tile
 * grid cannot fill the GPU the contraction is split over up to 8 workgroups per tile (raw f32 partial sums in
 * `workspace`) and a second kernel folds them in fixed order and applies the epilogue.  Differs from
 * ds_conv_fwd_f16 by the f32 summation order only; large launches take the one-pass path. */
long long ds_conv_f16_splitk_workspace_bytes(const ds_conv_shape *s);
int ds_conv_fwd_f16_splitk(const ds_conv_shape *s, const void *x_f16, const void *w_f16, const float *scale,
                           const float *shift, const void *residual_f16, void *y, int flags, void *workspace,
                           long long ws_bytes, void *stream);
/* out8 = {M tile, N tile, rows per segment, segments per tile, workgroups, LDS bytes, threads per workgroup,
 * 1000 * double-buffered + 100 * (16-channel chunks) + staging items per thread} */
int ds_conv_f16_plan_describe(const ds_conv_shape *s, int *out8);
/* the same under the DS_CONV_HINT_* / DS_CONV_IN_PLANES16 bits of `flags` (what ds_conv_fwd_f16 would launch) */
int ds_conv_f16_plan_describe_hinted(const ds_conv_shape *s, int flags, int *out8);
/* the pixel tile's LDS layout in that plan: out4 = {records per tile row, bytes per tile row, bytes per segment,
 * 1000 x LDS cycles of one fragment read (1000 = free of bank conflicts)} */
int ds_conv_f16_plan_lds_layout(const ds_conv_shape *s, int flags, int *out4);
/* tuning hook: 1 = pad tile rows / segments by 16-byte units until a fragment read is free of bank conflicts in the
 * planner's model; default 0 (whole records only: measured equal, tools/ab_layout.py) */
void ds_conv_f16_set_layout_padding(int on);
void ds_conv_f16_set_forced_cfg(int cfg);   /* tuning hook: plan with tile configuration `cfg` only (-1: the planner's choice) */
/* One whole BasicBlock in eval mode as ONE kernel (reference model.py:66-82):
 *     y = clip(bn2(conv3x3(clip(bn1(conv3x3(x))))) + x)
 * for the shallow stages (W = 32 with 64 channels, W = 16 with 128: ds_conv_block_f16_supported), where a workgroup
 * of the single-layer kernel spends more time in its prologue and epilogue than contracting.  The intermediate
 * activation lives in LDS only.  Bit-identical to two ds_conv_fwd_f16 calls.  flags: DS_EPI_OUT_F32, DS_EPI_OUT_PLANES16. */
int ds_conv_block_f16_supported(int B, int H, int W, int C);
int ds_conv_block_f16(const void *x_f16, const void *wa_f16, const void *wb_f16, const float *scale_a,
                      const float *shift_a, const float *scale_b, const float *shift_b, void *y, int B, int H, int W,
                      int C, int flags, void *stream);
/* the same block over a zero-padded batch of utterances of different lengths (BASELINE configs[4]): `lens` (device,
 * int32 [B]) = the rows of each image that belong to its utterance; rows past them come out zero after both layers, as
 * ds_mask_rows leaves them in the unfused sequence */
int ds_conv_block_f16_masked(const void *x_f16, const void *wa_f16, const void *wb_f16, const float *scale_a,
                             const float *shift_a, const float *scale_b, const float *shift_b, void *y, const int *lens,
                             int B, int H, int W, int C, int flags, void *stream);
int ds_cast_f32_to_f16(const float *x, void *y_f16, long long n, void *stream);
int ds_cast_f16_to_f32(const void *x_f16, float *y, long long n, void *stream);

/* ---- tail: temporal average pool, L2 normalisation --------------------------------------------- */
/* x [B,Hr,Wc,C] -> pooled [B, Wc*C] (index f*C + c), mean over Hr  (model.py:111,207-208) */
int ds_avgpool_time_f32(const float *x, float *pooled, int B, int Hr, int Wc, int C, void *stream);
/* e = alpha * f / sqrt(sum f^2 + eps) per row  (model.py:172-183, 210-213) */
int ds_l2norm_scale_f32(const float *f, float *e, int B, int D, float alpha, float eps,
                        void *stream);

/* out2[0] = max |a - b|, out2[1] = max |b| over n floats (one workgroup, fixed fold order): the embedding-error
 * measure of the tests (max |d| / max |ref|) on the device -- what the fp16 precision guard of
 * DeepSpeakerModel.forward (model.py:185-218 in eval mode) compares its fp16 and f32-class sample rows with */
int ds_max_abs_diff_f32(const float *a, const float *b, long long n, float *out2, void *stream);

/* fused projection + normalisation: f = pooled . W^T + b  (model.py:209), e = alpha f / |f|
 * (model.py:210-213).  Split-K MFMA GEMM + a deterministic reduce; `workspace` holds
 * ds_fc_workspace_floats(B,K,N) floats; `bias` and `e` may be NULL (plain GEMM, used by the backward
 * pass for gpooled = gf . W). */
long long ds_fc_workspace_floats(int B, int K, int N);
int ds_fc_l2norm_fwd_f32(const float *pooled, const float *w_packed, const float *bias,
                         float *workspace, float *f, float *e, int B, int K, int N, float alpha,
                         float eps, void *stream);

/* ---- loss side ----------------------------------------------------------------------------------- */
/* d[i] = sqrt(sum_k (x1[i,k]-x2[i,k])^2 + 1e-4/D)   (PairwiseDistance, model.py:13-18, p = 2) */
int ds_pairwise_distance_f32(const float *x1, const float *x2, float *d, int N, int D,
                             void *stream);
/* the same for any norm p > 0: pow(sum_k |x1 - x2|^p + 1e-4 / D, 1 / p) (reference model.py:13-18 with self.norm = p;
 * the reference's own call sites pass 2), and its gradient */
int ds_pairwise_distance_p_f32(const float *x1, const float *x2, float *d, int N, int D, float p, void *stream);
int ds_pairwise_distance_p_bwd_f32(const float *x1, const float *x2, const float *d, const float *gd, float *g1,
                                   float *g2, int N, int D, float p, void *stream);
/* TripletMarginLoss.forward (model.py:27-33): writes d_p[N], d_n[N] and loss[1] = mean hinge. */
int ds_triplet_margin_fwd_f32(const float *a, const float *p, const float *n, float margin,
                              float *d_p, float *d_n, float *loss, int N, int D, void *stream);
/* the reference's triplet "mining" (train_triplet.py:251-262): idx = ascending i with
 * d_n[i] - d_p[i] < margin; count[0] = number selected; mean_diff[0] = mean(d_n - d_p). */
int ds_triplet_filter_f32(const float *d_p, const float *d_n, float margin, long long *idx,
                          int *count, float *mean_diff, int N, void *stream);
/* The loss side of one triplet step in two launches (distances, then one scan): loss (model.py:27-33), the
 * ordered filter + mean(d_n - d_p) (train_triplet.py:253-262) and, when amb_cap > 0, the ordered list of near
 * ties |d_n - d_p - margin| < band (first amb_cap of them; unused slots hold 0; amb_count = true count).  The near
 * ties are what a reduced-precision forward may decide differently from the reference: the caller re-embeds them
 * at f32-class precision, patches their distances with ds_refine_distances_f32 (e_ref rows: anchors | positives |
 * negatives, cap each) and re-runs the scan with ds_triplet_scan_f32. */
int ds_triplet_tail_f32(const float *a, const float *p, const float *n, float margin, float band, float *d_p,
                        float *d_n, float *loss, long long *idx, int *count, float *mean_diff, long long *amb_idx,
                        int *amb_count, int amb_cap, int N, int D, void *stream);
int ds_triplet_scan_f32(const float *d_p, const float *d_n, float margin, float *loss, long long *idx, int *count,
                        float *mean_diff, int N, void *stream);
int ds_refine_distances_f32(const float *e_ref, const long long *amb_idx, const int *amb_count, int cap, float *d_p,
                            float *d_n, int D, void *stream);
/* Round 4: the same two calls with PROBES -- how the band of the near-tie refinement checks itself.  The refinement
 * re-embeds all `cap` slots whether near ties fill them or not; ds_triplet_tail_probe_f32 puts the triplets
 * (probe_base + k) mod N, k = 0, 1, ... into the slots the near ties leave unused (probe_base < 0: none, == ds_triplet_tail_f32),
 * and ds_refine_distances_probe_f32 patches ALL cap slots and reports err[0] = max over the slots of
 * |(d_n - d_p) at f32-class precision - (d_n - d_p) before| (the fp16 forward's error on the filter's decision variable,
 * train_triplet.py:251-253) and err[1] = the number of slots sampled (as float); "before" is read from d_p_before /
 * d_n_before, the unpatched distances the scan chose the slots on (buffers other than d_p / d_n).  With the path's own
 * embeddings emb_a / emb_p / emb_n [N][D] (all or none) also err[2] = max |e_ref - emb| and err[3] = max |e_ref| over the
 * 3 * cap sampled rows: the path's distance to the 1e-3 embedding contract, watched on the same sample (err holds 4 floats).
 * One workgroup, no atomics. */
int ds_triplet_tail_probe_f32(const float *a, const float *p, const float *n, float margin, float band, float *d_p,
                              float *d_n, float *loss, long long *idx, int *count, float *mean_diff, long long *amb_idx,
                              int *amb_count, int amb_cap, int probe_base, int N, int D, void *stream);
int ds_refine_distances_probe_f32(const float *e_ref, const long long *amb_idx, const int *amb_count, int cap, float *d_p,
                                  float *d_n, const float *d_p_before, const float *d_n_before, const float *emb_a,
                                  const float *emb_p, const float *emb_n, int D, float *err, void *stream);
/* Round 6: the same in one launch instead of five (two clones, the patch, two read-back copies): d_p / d_n [N] are WRITTEN
 * (= d_p_before / d_n_before with the cap slots patched) and err5 holds five floats, err5[4] = the near-tie count. */
int ds_refine_distances_fused_f32(const float *e_ref, const long long *amb_idx, const int *amb_count, int cap, float *d_p,
                                  float *d_n, const float *d_p_before, const float *d_n_before, const float *emb_a,
                                  const float *emb_p, const float *emb_n, int N, int D, float *err5, void *stream);


/* ---- backward of the convolution stack (torch autograd of nn.Conv2d / nn.BatchNorm2d under
 *      loss.backward(), train_triplet.py:223,290) -------------------------------- */
/* filter bank of the 5x5 stride-2 data gradient: four parity classes, see ds_conv_dgrad_f32 */
int ds_pack_conv_dgrad_s2_f32(const float *w_oihw, float *w_packed, int Cout, int Cin, void *stream);
/* gx[B,H,W,Cin] = dL/dx given gy[B,Ho,Wo,Cout] = dL/dy of the convolution described by `s` (the
 * FORWARD shape).  w_dgrad_packed: ds_pack_conv_weight_f32(..., dgrad=1) for stride 1,
 * ds_pack_conv_dgrad_s2_f32 for the 5x5 stride-2 layers. */
int ds_conv_dgrad_f32(const ds_conv_shape *s, const float *gy, const float *w_dgrad_packed,
                      float *gx, void *stream);
/* gw_oihw[Cout,Cin,KS,KS] = dL/dW given the layer input x[B,H,W,Cin] and gy[B,Ho,Wo,Cout].  Pixel-split
 * MFMA GEMM + deterministic reduce; `workspace`: ds_conv_wgrad_workspace_floats(s) floats.  Cin = 1
 * selects the conv1 kernel.  fc_F > 0: `s` is the fc layer as a 1x1 convolution over [1,B,1,K] and
 * the result is written in the reference's [N, C*F] feature order (model.py:164,208). */
long long ds_conv_wgrad_workspace_floats(const ds_conv_shape *s);
int ds_conv_wgrad_f32(const ds_conv_shape *s, const float *x, const float *gy, float *workspace,
                      float *gw_oihw, int fc_F, void *stream);
/* the same filter gradient on the bf16 matrix cores with split operands (bf16x3, f32-class accuracy):
 * 3x3 stride 1 and 5x5 stride 2 layers with Cin, Cout multiples of 64.  `workspace`:
 * ds_conv_wgrad_bf16_workspace_floats(s) floats.  Deterministic (fixed-order reduction). */
long long ds_conv_wgrad_bf16_workspace_floats(const ds_conv_shape *s);
int ds_conv_wgrad_bf16(const ds_conv_shape *s, const float *x, const float *gy, float *workspace,
                       float *gw_oihw, void *stream);
/* BatchNorm (train mode) backward in one call: gy = (g1 [+ g2]) masked by the clipped-ReLU of `act`
 * (NULL: unmasked); reductions sum gy, sum gy*xhat; ggamma, gbeta; gz = dL/d(conv output).
 * partial: ds_bn_bwd_partial_rows(n_pix, C) * C * 2 floats; coef: 3*C floats; gy is written (it is the
 * masked gradient the residual branch re-uses). */
int ds_bn_bwd_partial_rows(long long n_pix, int C);
int ds_bn_bwd_f32(const float *g1, const float *g2, const float *act, const float *z,
                  const float *mean, const float *invstd, const float *gamma, float *gy,
                  float *partial, float *coef, float *ggamma, float *gbeta, float *gz,
                  long long n_pix, int C, void *stream);
/* the same for a batch made of G members with their own batch statistics (the three forwards of a triplet step run as
 * one batch, train_triplet.py:215), in four launches: tensors [G * n_pix, C]; mean, invstd [G][C]; partial
 * G * ds_bn_bwd_partial_rows(n_pix, C) * C * 2 floats; coef [G][3C]; member_sums [2][G][C] scratch; ggamma / gbeta [C] =
 * the members' gradients added in member order (what three backward passes accumulate into .grad) */
int ds_bn_bwd_group_f32(const float *g1, const float *g2, const float *act, const float *z, const float *mean,
                        const float *invstd, const float *gamma, float *gy, float *partial, float *coef,
                        float *member_sums, float *ggamma, float *gbeta, float *gz, long long n_pix, int C, int G,
                        void *stream);
/* ds_bn_bwd_group_f32 split where data-parallel training exchanges the sums (new capability, the
 * reference is single-GPU: train_triplet.py:97): local reductions of all G members -> sums [G][2C+1] float64
 * ({sum gy, sum gy*xhat} per channel, then the member's pixel count) ... the caller all-reduces `sums` over RCCL ...
 * -> coefficients, dgamma / dbeta (members added in order) and gz of all members.  2 + 3 launches per layer. */
int ds_bn_bwd_group_reduce_f32(const float *g1, const float *g2, const float *act, const float *z,
                               const float *mean, const float *invstd, float *gy, float *partial, double *sums,
                               long long n_pix, int C, int G, void *stream);
int ds_bn_bwd_group_apply_f32(const double *sums, const float *gy, const float *z, const float *mean,
                              const float *invstd, const float *gamma, float *coef, float *member_sums,
                              float *ggamma, float *gbeta, float *gz, long long n_pix, int C, int G, void *stream);
/* The 3x3 stride-1 data gradient FUSED with the first half of the BatchNorm backward of the layer it feeds (autograd of
 * y = clip(bn1(conv1(r))) / r = clip(bn(conv(x))) under loss.backward(), train_triplet.py:223 over model.py:69-75,187-189):
 *   gy = (dgrad(gz_up) [+ g2]) * [0 < z * mask_scale + mask_shift < 20],  partial[row] = { sum gy, sum gy * xhat }
 * i.e. ds_conv_dgrad_bf16 followed by the reduce kernel of ds_bn_bwd_group_f32, without the gradient's round trip
 * through HBM and without reading the activation (the clipped-ReLU mask is re-derived from the layer's own
 * pre-activation z with the very fma of ds_bn_apply_f32).  `s`: the FORWARD shape of the convolution whose data gradient
 * this is; z / g2 / gy [B,H,W,Cin]; the batch = G members of B/G utterances with their own statistics, tables
 * mean / invstd / mask_scale / mask_shift [G][Cin].  ds_conv_dgrad_bnbwd_bf16_rows: partial rows PER MEMBER (partial =
 * G * rows * Cin * 2 floats), or DS_ERR_UNSUPPORTED when a tile of the launch would straddle two members (fall back to
 * ds_conv_dgrad_bf16 + ds_bn_bwd_group_f32).  ds_bn_bwd_group_finish_f32 is the second half (coefficients, dgamma /
 * dbeta added in member order, gz); data-parallel training puts ds_partial_sum_f64_group -> all-reduce ->
 * ds_bn_bwd_group_apply_f32 in its place. */
int ds_conv_dgrad_bnbwd_bf16_rows(const ds_conv_shape *s, int G);
int ds_conv_dgrad_bnbwd_bf16(const ds_conv_shape *s, const float *gz_up, const void *w_hi, const void *w_lo,
                             const float *g2, const float *z, const float *mean, const float *invstd,
                             const float *mask_scale, const float *mask_shift, int G, float *gy, float *partial,
                             void *stream);
/* the 5x5 stride-2 data gradient with the same fusion; it feeds a BasicBlock's OUTPUT out = clip(bn2(conv2(y)) + r)
 * (model.py:76-80), whose mask needs the stored activation `act` (= the 5x5 layer's input, [B,H,W,Cin]); nothing is
 * added.  Four parity-class launches write interleaved pixels of gy and consecutive blocks of the members' partial rows
 * (ds_conv_dgrad_s2_bnbwd_bf16_rows: rows per member over all four). */
int ds_conv_dgrad_s2_bnbwd_bf16_rows(const ds_conv_shape *s, int G);
int ds_conv_dgrad_s2_bnbwd_bf16(const ds_conv_shape *s, const float *gz_up, const void *w_hi, const void *w_lo,
                                const float *act, const float *z, const float *mean, const float *invstd, int G,
                                float *gy, float *partial, void *stream);
int ds_bn_bwd_group_finish_f32(const float *partial, int n_partial, const float *gy, const float *z, const float *mean,
                               const float *invstd, const float *gamma, float *coef, float *member_sums, float *ggamma,
                               float *gbeta, float *gz, long long n_pix, int C, int G, void *stream);
/* forward counterpart: per-tile partial statistics of G members (n_partial rows of [C][2] each, consecutive)
 * -> sums [G][2C+1] float64 (count in the last slot of each row) in one launch */
int ds_partial_sum_f64_group(const float *partial, int n_partial, double *sums, long long count, int C, int G,
                             void *stream);
int ds_colsum_f32(const float *x, float *out, int R, int C, void *stream);

/* ---- split forms for data-parallel training (one process per GPU): the caller all-reduces the
 *      [C][2] float64 sums over RCCL between the two halves, so that N ranks normalise with the
 *      statistics of the GLOBAL batch exactly like one process would.  count == 0:
 *      the (all-reduced) pixel count is read from sums[2*C] on the device -- no host sync. ------- */
int ds_partial_sum_f64(const float *partial, int n_partial, double *sums, int C, void *stream);
int ds_bn_stats_from_sums_f32(const double *sums, long long count, const float *gamma,
                              const float *beta, float eps, float momentum, float *running_mean,
                              float *running_var, float *batch_mean, float *batch_invstd,
                              float *scale, float *shift, int C, void *stream);
int ds_bn_bwd_reduce_f32(const float *g1, const float *g2, const float *act, const float *z,
                         const float *mean, const float *invstd, float *gy, float *partial,
                         long long n_pix, int C, void *stream);
int ds_bn_bwd_apply_f32(const double *sums, long long count, const float *gy, const float *z,
                        const float *mean, const float *invstd, const float *gamma, float *coef,
                        float *ggamma, float *gbeta, float *gz, long long n_pix, int C, void *stream);

/* row gather and its adjoint (selection of mined candidates and the gradient back to them) */
int ds_gather_rows_f32(const float *src, const long long *idx, float *dst, int N, int D, void *stream);
/* dst [3][N][D] = the rows idx[0..N) of three sources in one launch (the utterances of the near-tie triplets; a negative
 * index gathers zeros); D % 4 == 0, pointers 16-byte aligned */
int ds_gather_rows3_f32(const float *src_a, const float *src_p, const float *src_n, const long long *idx, float *dst,
                        int N, int D, void *stream);
int ds_scatter_add_rows_f32(const float *g, const long long *idx, float *dst, int N, int M, int D,
                            int accumulate, void *stream);

/* ---- cross-GPU semi-hard negative search over an all-gathered candidate set (north_star; no
 *      reference counterpart).  out_index[i] in [0,M) or -1; out_dist may be NULL. ---- */
long long ds_mine_workspace_floats(int N, int M);
int ds_mine_semihard_f32(const float *anchor, const float *d_p, const long long *anchor_label,
                         const float *cand, const long long *cand_label, float *workspace,
                         long long *out_index, float *out_dist, int N, int M, int D, void *stream);

/* ---- backward of the loss side and the tail (torch autograd of the lines cited above; the
 *      reference obtains them from loss.backward(), train_triplet.py:223,290) ------------------- */
int ds_pairwise_distance_bwd_f32(const float *x1, const float *x2, const float *d, const float *gd,
                                 float *g1, float *g2, int N, int D, void *stream);
/* grad_loss points at ONE device float (dL/dloss); clamp(min=0) passes gradient at exactly 0. */
int ds_triplet_margin_bwd_f32(const float *a, const float *p, const float *n, const float *d_p,
                              const float *d_n, float margin, const float *grad_loss, float *ga,
                              float *gp, float *gn, int N, int D, void *stream);
int ds_l2norm_scale_bwd_f32(const float *f, const float *ge, float *gf, int B, int D, float alpha,
                            float eps, void *stream);
/* backward of [clip(0,20) -> mean over time]: gx = (0 < out < 20) ? gpooled / Hr : 0 */
int ds_avgpool_time_bwd_f32(const float *gpooled, const float *out, float *gx, int B, int Hr, int Wc,
                            int C, void *stream);


/* Softmax pre-training head in one call (model.py:220-223 + train_triplet.py:281-285): logits = x W^T + b on the
 * f32 matrix cores with row max / log-sum-exp / per-row loss taken in the split-K reduction's epilogue, then the
 * mean.  N = n_cls padded to a multiple of 128 (zero filter rows); logits is [M, N]. */
int ds_fc_ce_fwd_f32(const float *x, const float *w_packed, const float *bias, float *workspace, float *logits,
                     const long long *labels, float *row_loss, float *lse, float *loss, int M, int K, int N,
                     int n_cls, void *stream);

/* ---- softmax cross-entropy of the classifier logits (nn.CrossEntropyLoss, mean reduction;
 *      train_triplet.py:281-287).  logits [M, ld] with n_cls valid columns; lse[M] and row_loss[M] are
 *      outputs of the forward that the backward re-uses; dlogits [M, ld_out], columns >= n_cls zeroed. -- */
int ds_cross_entropy_fwd_f32(const float *logits, const long long *labels, float *row_loss, float *lse,
                             float *loss, int M, int n_cls, int ld, void *stream);
int ds_cross_entropy_bwd_f32(const float *logits, const long long *labels, const float *lse,
                             const float *grad_loss, float *dlogits, int M, int n_cls, int ld, int ld_out,
                             void *stream);

/* ---- variable-length batches (BASELINE configs[4]; the temporal mean pool of model.py:207 accepts any T) --------
 * Utterances of different lengths share one zero-padded batch [B,1,Tmax,64].  ds_mask_rows re-zeroes, after every
 * layer, the rows past each utterance's own extent (lens[b] rows of the [H][row_bytes] slab of image b are kept):
 * those zeros then act exactly like the utterance's own zero padding, so every kept row is bit-identical to the
 * forward of the utterance alone.  ds_avgpool_time_masked_f32 averages over the utterance's own rows. */
int ds_mask_rows(void *x, const int *lens, int B, int H, long long row_bytes, void *stream);
int ds_avgpool_time_masked_f32(const float *x, const int *lens, float *pooled, int B, int Hr, int Wc, int C,
                               void *stream);
/* out[i] = mean of x[offsets[i] .. offsets[i+1]): enrolment sets of different sizes (train_triplet.py:350 takes the
 * mean of a trial's distances) */
int ds_segment_mean_f32(const float *x, const long long *offsets, float *out, int n_seg, void *stream);

/* ---- verification scoring on the device --------------------------------------
 * ds_group_mean_f32: score of a trial = mean over G crop-pair distances (train_triplet.py:347-350).
 * ds_roc_sweep_f32: the threshold sweep of eval_metrics.py:5-50 (predict = dist < thr; tp/fp per
 * threshold thr0 + i*dthr) plus a summary {best-accuracy threshold index (first argmax), tpr, fpr,
 * accuracy there, EER, EER threshold}; the reference computes no EER. -------------------- */
int ds_group_mean_f32(const float *x, float *out, int n_groups, int G, void *stream);
int ds_roc_sweep_f32(const float *dist, const int *issame, int N, float thr0, float dthr, int n_thr,
                     int n_same, int n_diff, int *tp, int *fp, float *summary6, void *stream);

/* ---- small-batch tail (serving latency, round 4): temporal mean + 2048 -> 512 projection in one launch for
 *      B <= DS_TAIL_SMALL_MAX_B utterances, then the norm (model.py:207-213).  w_rows = the fc filter as [N][K'] rows in
 *      the pooled vector's f*C + c order (ds_pack_fc_weight_rows_f32).  Agrees with ds_avgpool_time_f32 +
 *      ds_fc_l2norm_fwd_f32 to f32 rounding (another summation order), 8 us instead of 30 at B = 1. ---- */
#define DS_TAIL_SMALL_MAX_B 4
int ds_pack_fc_weight_rows_f32(const float *w, float *w_rows, int N, int C, int F, void *stream);
int ds_tail_small_f32(const float *a, const float *w_rows, const float *bias, float *f, float *e, int B, int Hr, int K,
                      int N, float alpha, float eps, void *stream);

/* ---- OPT-IN fp16 training step (round 4; DeepSpeakerModel(train_precision="f16")): the triplet-regime step of
 *      train_triplet.py:215-224 with every activation, pre-activation and gradient tensor fp16 in HBM, the convolutions
 *      (forward AND data gradients) on ds_conv_fwd_f16 -- one fp16 MFMA per product, f32 accumulate -- and f32
 *      statistics / parameter gradients.  Gradient tensors hold S * g for a constant loss scale S (a power of two);
 *      what leaves in f32 (dgamma, dbeta, filter gradients) is un-scaled by `inv_scale` / `out_scale` = 1 / S.
 *      Tolerances (tests/test_gpu_train_f16.py, measured at 768 rows in brackets): loss 1e-3 (3.5e-4), train-mode
 *      embeddings 2e-3 (1.15e-3), gradients 8e-3 of the masked oracle (4.0e-3 worst, 1.4e-3 median; the reference's own fp32
 *      run is 4e-3 from its fp64 run on unmasked gradients). ---- */
/* data-gradient filter banks for ds_conv_fwd_f16.  stride 1: Cout*Cin*KS*KS halfs, [Cout/16][tap][Cin][16] with taps
 * flipped -- run with shape {B, Ho, Wo, Cin' = Cout, Cout' = Cin, KS, 1} over dL/d(conv output).  stride 2 (KS = 5):
 * 36*Cout*Cin halfs, [Cout/16][9][4 Cin][16] -- the four parity classes of dX as the output-channel blocks of ONE 3x3
 * stride-1 convolution {B, Ho, Wo, Cout, 4 Cin, 3, 1}; its output [B][Ho][Wo][2][2][Cin] is read in place by
 * ds_bn_bwd_group_f16(g1_parity = 1). */
int ds_pack_conv_weight_dgrad_f16(const float *w_oihw, void *w_f16, int Cout, int Cin, int KS, int stride, void *stream);
/* train-mode BatchNorm statistics of G members (rows [m*n_pix, (m+1)*n_pix) of z) -> [G][C] tables; running statistics
 * updated member after member (the reference's three forward calls); partial: G * ds_bn_f16_partial_rows(n_pix, C) * C * 2
 * floats of scratch.  Replaces nn.BatchNorm2d.train() forward bookkeeping (model.py:59,62,94,99,103,107). */
int ds_bn_f16_partial_rows(long long n_pix, int C);
int ds_bn_stats_group_f16(const void *z_f16, float *partial, long long n_pix, const float *gamma, const float *beta,
                          float eps, float momentum, float *running_mean, float *running_var, float *mean_t,
                          float *invstd_t, float *scale_t, float *shift_t, int C, int G, void *stream);
/* y = clip(z * scale[m] + shift[m] (+ residual)); flags: DS_EPI_RESIDUAL | DS_EPI_CLIP | DS_EPI_OUT_F32 (y f32: the last
 * stage, whose consumer is the f32 pooling / projection tail).  model.py:70-71,74-80,188-189 in train mode. */
int ds_bn_apply_group_f16(const void *z_f16, const float *scale_t, const float *shift_t, const void *res_f16, void *y,
                          long long n_pix, int C, int G, int flags, void *stream);
/* backward of BatchNorm(train) + clipped ReLU for G members: gy = (g1 [+ g2]) * mask, gz = dL/d(conv output); ggamma /
 * gbeta [C] summed over the members and un-scaled.  mask = 0 < act < 20 (act fp16, or f32 with act_is_f32), or -- act NULL,
 * mask_scale_t / mask_shift_t = the forward's [G][C] scale / shift tables -- 0 < fp16(clip(z * scale + shift)) < 20 taken
 * from the layer's own pre-activation (no third tensor read), or none (all NULL: g1 is already masked).  gy may be NULL
 * with the z-derived mask (no g2, no parity layout): the masked gradient is then never stored.
 * partial as above; coef [G][3 C] scratch.  H, W: the layer's map (used when g1_parity). */
int ds_bn_bwd_group_f16(const void *g1, int g1_parity, const void *g2, const void *act, int act_is_f32,
                        const float *mask_scale_t, const float *mask_shift_t, const void *z, const float *mean_t,
                        const float *invstd_t, const float *gamma, void *gy, float *partial, float *coef, float *ggamma,
                        float *gbeta, void *gz, long long n_pix, int H, int W, int C, int G, float inv_scale, void *stream);
/* data-parallel split forms (one process per GPU): partial sums only / the reduction only, then -- after
 * ds_partial_sum_f64_group(partial, ds_bn_f16_partial_rows(n_pix, C), sums, n_pix, C, G) and an all-reduce of the
 * [G][2C+1] float64 sums over RCCL -- ds_bn_stats_from_sums_f32 per member (forward) / ds_bn_bwd_group_apply_f16 (backward:
 * coefficients, dgamma / dbeta, gz from the GLOBAL sums; regen: gy was not stored, gy_or_g1 is g1 and the mask tables
 * are given).  An N-rank step then equals the single-process step on the global batch. */
int ds_bn_stats_partial_f16(const void *z_f16, float *partial, long long n_pix, int C, int G, void *stream);
int ds_bn_stats_from_sums_group_f32(const double *sums, const float *gamma, const float *beta, float eps, float momentum,
                                    float *running_mean, float *running_var, float *mean_t, float *invstd_t,
                                    float *scale_t, float *shift_t, int C, int G, void *stream);   /* all G members, one launch */
int ds_bn_bwd_group_reduce_f16(const void *g1, int g1_parity, const void *g2, const void *act, int act_is_f32,
                               const float *mask_scale_t, const float *mask_shift_t, const void *z, const float *mean_t,
                               const float *invstd_t, void *gy, float *partial, long long n_pix, int H, int W, int C, int G,
                               void *stream);
int ds_bn_bwd_group_apply_f16(const double *sums, const void *gy_or_g1, int regen, const float *mask_scale_t,
                              const float *mask_shift_t, const void *z, const float *mean_t, const float *invstd_t,
                              const float *gamma, float *coef, float *ggamma, float *gbeta, void *gz, long long n_pix, int C,
                              int G, float inv_scale, void *stream);
int ds_scale_cast_f32_to_f16(const float *x, void *y_f16, long long n, float scale, void *stream);
/* filter gradients from fp16 activations x and fp16 loss-scaled output gradients gy (cuDNN wgrad under
 * loss.backward(), train_triplet.py:223): 3x3 / 5x5, stride 1 / 2, Cin and Cout multiples of 64. */
long long ds_conv_wgrad_f16_workspace_floats(const ds_conv_shape *s);
int ds_conv_wgrad_f16(const ds_conv_shape *s, const void *x_f16, const void *gy_f16, float *workspace, float *gw_oihw,
                      float out_scale, void *stream);
/* conv1 (Cin = 1): f32 network input, fp16 output gradient; workspace ds_conv_wgrad_workspace_floats(s) floats */
int ds_conv_wgrad_c1_f16(const ds_conv_shape *s, const float *x, const void *gy_f16, float *workspace, float *gw_oihw,
                         float out_scale, void *stream);

/* ---- batch assembly on the device: out[b, t, :] = features[row_start[b] + t, :]
 *      for t < T, zero past row_end[b]; features = the corpus' [frames, F] fbank matrices concatenated
 *      and resident in HBM.  Replaces the host-side np.load + crop + transpose + H2D of
 *      audio_processing.py:38-74,185 / DeepSpeakerDataset_dynamic.py:82-103 per batch. ---------------- */
int ds_assemble_crops_f32(const float *features, const long long *row_start, const long long *row_end,
                          float *out, int B, int T, int F, void *stream);

/* ---- fused multi-tensor optimizer steps: one launch updates every parameter
 *      tensor; replaces torch.optim.{Adagrad,SGD,Adam}.step() of train_triplet.py:369-383,224,291 with
 *      the same arithmetic.  `params`, `grads`, `state1`, `state2` are DEVICE arrays of device pointers
 *      (one per tensor), `numel` a device int64 array; workgroup b updates elements
 *      [chunk_index[b]*ds_optim_chunk_elems(), +ds_optim_chunk_elems()) of tensor chunk_tensor[b]. ---- */
/* Steps that are captured into a HIP graph and replayed: the step count lives on the device (`step_count`, int32 [1],
 * bumped by ds_optim_step_inc once per optimizer step unless the skip flag is set) and Adagrad's decayed learning rate /
 * Adam's bias corrections are computed from it inside the kernel, in double precision as the host computes them. */
int ds_optim_step_inc(int *step_count, const int *skip_flag, void *stream);
int ds_adagrad_step_dev_f32(const void *params, const void *grads, const void *state1, const void *state2,
                            const long long *numel, const int *chunk_tensor, const int *chunk_index, int n_chunks,
                            double lr, double lr_decay, float weight_decay, float eps, const int *step_count,
                            const int *skip_flag, void *stream);
int ds_sgd_step_dev_f32(const void *params, const void *grads, const void *state1, const void *state2,
                        const long long *numel, const int *chunk_tensor, const int *chunk_index, int n_chunks, float lr,
                        float momentum, float dampening, float weight_decay, const int *step_count, const int *skip_flag,
                        void *stream);      /* the momentum buffers start at the step that counts 1 */
int ds_adam_step_dev_f32(const void *params, const void *grads, const void *state1, const void *state2,
                         const long long *numel, const int *chunk_tensor, const int *chunk_index, int n_chunks, float lr,
                         double beta1, double beta2, float eps, float weight_decay, const int *step_count,
                         const int *skip_flag, void *stream);
/* `bytes` (<= 2048, a multiple of 4) of a HOST table -> device memory as kernel arguments: no staging buffer, legal inside a
 * stream capture (a replayed graph writes the same values again).  The fused optimizers' pointer tables travel this way. */
int ds_fill_bytes(void *dst, const void *host_src, int bytes, void *stream);
int ds_optim_chunk_elems(void);
int ds_adagrad_step_f32(const void *params, const void *grads, const void *state1, const void *state2,
                        const long long *numel, const int *chunk_tensor, const int *chunk_index,
                        int n_chunks, float clr, float weight_decay, float eps, const int *skip_flag,
                        void *stream);
int ds_sgd_step_f32(const void *params, const void *grads, const void *state1, const void *state2,
                    const long long *numel, const int *chunk_tensor, const int *chunk_index, int n_chunks,
                    float lr, float momentum, float dampening, float weight_decay, int first_step,
                    const int *skip_flag, void *stream);
int ds_adam_step_f32(const void *params, const void *grads, const void *state1, const void *state2,
                     const long long *numel, const int *chunk_tensor, const int *chunk_index, int n_chunks,
                     float lr, float beta1, float beta2, float eps, float weight_decay,
                     float bias_correction1, float bias_correction2_sqrt, const int *skip_flag, void *stream);
/* `skip_flag` (nullable, device): non-zero = the launch leaves parameters and state untouched -- the overflow flag of a
 * loss-scaled fp16 training step, read on the device (no host round trip; torch.amp.GradScaler's found_inf).
 * ds_nonfinite_flag_f32: *flag = 1 if any of x[0..n) is inf or NaN (never cleared here: the tensors of one step
 * accumulate into one flag the caller zeroed). */
int ds_nonfinite_flag_f32(const float *x, long long n, int *flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPSPEAKER_HIP_H */
