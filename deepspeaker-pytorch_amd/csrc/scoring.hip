// scoring.hip -- verification scoring on the device.
//   * test-time score of a trial = mean over the crop pairs of the pairwise distance
//     (reference train_triplet.py:347-350: dists.reshape(current_sample, test_input_per_file).mean(axis=1))
//   * threshold sweep of eval_metrics.py:5-50 (thresholds 0..30 step 0.01; predict_issame =
//     dist < threshold; tp/fp/tn/fn; accuracy; best-accuracy threshold = first argmax)
//   * equal error rate (not computed by the reference at all -- SURVEY F7 -- added here)
#include <ds_device.h>
#include "ds_common.h"

namespace {

// out[i] = mean_j x[i*G + j]
__global__ void __launch_bounds__(256) group_mean_kernel(const float *x, float *out, int n_groups, int G) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_groups) {
        float s = 0.f;
        for (int j = 0; j < G; ++j) s += x[(size_t)i * G + j];
        out[i] = s / (float)G;
    }
}

// out[i] = mean of x[off[i] .. off[i+1]) -- enrolment sets of different sizes (off is a prefix-sum table)
__global__ void __launch_bounds__(256) segment_mean_kernel(const float *x, const long long *off, float *out, int n_seg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_seg) {
        const long long a = off[i], b = off[i + 1];
        float s = 0.f;
        for (long long j = a; j < b; ++j) s += x[j];
        out[i] = b > a ? s / (float)(b - a) : 0.f;
    }
}

// ---- variable-length batches (BASELINE configs[4]) ------------------------------------------------------------
// Utterances of different lengths share one zero-padded batch.  A convolution of the padded batch equals the
// convolution of each utterance alone IF the rows past an utterance's own extent are zero at every layer -- they
// then act exactly like that utterance's zero padding (same products, same accumulation order: bit-identical).
// BatchNorm's shift and the clip make them non-zero again after every layer, so they are re-zeroed here: image b
// keeps rows [0, lens[b]) of its [H][row_bytes] slab.  Work is proportional to the padding only.
__global__ void __launch_bounds__(256) mask_rows_kernel(char *x, const int *lens, int H, long long row_bytes) {
    const int b = blockIdx.x >> 3, part = blockIdx.x & 7;       // eight workgroups per image
    const int len = lens[b] < H ? (lens[b] > 0 ? lens[b] : 0) : H;
    const long long n16 = ((long long)(H - len) * row_bytes) >> 4;
    f32x4 *dst = (f32x4 *)(x + ((long long)b * H + len) * row_bytes);
    for (long long i = (long long)part * 256 + threadIdx.x; i < n16; i += 8 * 256) dst[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// temporal mean over the utterance's own rows: pooled[b] = sum_{h < lens[b]} x[b,h] / lens[b]  (model.py:207 applied
// to the unpadded utterance).  Rows past lens[b] are never read.
__global__ void __launch_bounds__(256) avgpool_time_masked_kernel(const float *x, const int *lens, float *pooled, int B,
                                                                  int Hr, int row_elems) {
    const int vec_per_row = row_elems >> 2;
    const long long n = (long long)B * vec_per_row;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / vec_per_row), v = (int)(i - (long long)b * vec_per_row);
        const int len = lens[b] < Hr ? (lens[b] > 1 ? lens[b] : 1) : Hr;
        const f32x4 *src = (const f32x4 *)(x + (size_t)b * Hr * row_elems) + v;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int h = 0; h < len; ++h) s += src[(size_t)h * vec_per_row];
        const float hr = (float)len;
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = s[j] / hr;
        ((f32x4 *)(pooled + (size_t)b * row_elems))[v] = s;
    }
}

// one thread per threshold; distances / labels are streamed through LDS 1024 at a time
__global__ void __launch_bounds__(256) roc_sweep_kernel(const float *dist, const int *issame, int N, float t0, float dt,
                                                        int n_thr, int *tp, int *fp) {
    float *sd = ds_dynamic_lds();            // [1024] distances
    int *sl = (int *)(sd + 1024);            // [1024] labels
    const int ti = blockIdx.x * 256 + threadIdx.x;
    const float thr = t0 + dt * (float)ti;
    int ctp = 0, cfp = 0;
    for (int i0 = 0; i0 < N; i0 += 1024) {
        __syncthreads();
        for (int k = threadIdx.x; k < 1024; k += 256) {
            const int i = i0 + k;
            sd[k] = i < N ? dist[i] : 3.0e38f;
            sl[k] = i < N ? issame[i] : 0;
        }
        __syncthreads();
        const int n = (N - i0) < 1024 ? (N - i0) : 1024;
        for (int k = 0; k < n; ++k) {
            const int pred = sd[k] < thr;    // np.less(dist, threshold), eval_metrics.py:41
            ctp += pred & (sl[k] != 0);
            cfp += pred & (sl[k] == 0);
        }
    }
    if (ti < n_thr) {
        tp[ti] = ctp;
        fp[ti] = cfp;
    }
}

// summary[0..5] = {best threshold index (first argmax of accuracy), tpr, fpr, accuracy at it, EER, EER threshold}
__global__ void __launch_bounds__(256) roc_summary_kernel(const int *tp, const int *fp, int n_thr, int n_same,
                                                          int n_diff, int N, float t0, float dt, float *summary) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int best = 0;
    float best_acc = -1.f;
    float eer = 1.0f, eer_thr = t0;
    bool have_eer = false;
    float prev_fpr = 0.f, prev_fnr = 1.f;
    for (int i = 0; i < n_thr; ++i) {
        const int tn = n_diff - fp[i];
        const float acc = (float)(tp[i] + tn) / (float)N;            // eval_metrics.py:49
        if (acc > best_acc) { best_acc = acc; best = i; }            // np.argmax: first maximum
        const float fpr = n_diff ? (float)fp[i] / (float)n_diff : 0.f;
        const float fnr = n_same ? 1.0f - (float)tp[i] / (float)n_same : 0.f;
        if (!have_eer && fpr >= fnr) {                               // first crossing of FPR and FNR
            if (i == 0) { eer = 0.5f * (fpr + fnr); eer_thr = t0; }
            else {
                const float d0 = prev_fnr - prev_fpr, d1 = fpr - fnr;       // both >= 0
                const float w = (d0 + d1) > 0.f ? d0 / (d0 + d1) : 0.f;
                eer = prev_fpr + w * (fpr - prev_fpr);
                eer_thr = t0 + dt * ((float)(i - 1) + w);
            }
            have_eer = true;
        }
        prev_fpr = fpr;
        prev_fnr = fnr;
    }
    summary[0] = (float)best;
    summary[1] = n_same ? (float)tp[best] / (float)n_same : 0.f;    // eval_metrics.py:47
    summary[2] = n_diff ? (float)fp[best] / (float)n_diff : 0.f;    // eval_metrics.py:48
    summary[3] = best_acc;
    summary[4] = eer;
    summary[5] = eer_thr;
}

}  // namespace

extern "C" int ds_group_mean_f32(const float *x, float *out, int n_groups, int G, void *stream) {
    DS_REQUIRE(x && out, DS_ERR_NULL);
    DS_REQUIRE(n_groups > 0 && G > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(group_mean_kernel, ds_ceil_div(n_groups, 256), 256, 0, stream, x, out, n_groups, G);
    return ds_last_launch_error();
}

extern "C" int ds_roc_sweep_f32(const float *dist, const int *issame, int N, float thr0, float dthr, int n_thr,
                                int n_same, int n_diff, int *tp, int *fp, float *summary6, void *stream) {
    DS_REQUIRE(dist && issame && tp && fp && summary6, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && n_thr > 0 && n_same >= 0 && n_diff >= 0 && n_same + n_diff == N, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(roc_sweep_kernel, ds_ceil_div(n_thr, 256), 256, 2 * 1024 * 4, stream, dist, issame, N, thr0, dthr, n_thr,
              tp, fp);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(roc_summary_kernel, 1, 256, 0, stream, (const int *)tp, (const int *)fp, n_thr, n_same, n_diff, N, thr0,
              dthr, summary6);
    return ds_last_launch_error();
}

// ---- batch assembly on the device -------------------------------------------
// The reference builds every batch on the host: np.load of three .npy feature files per triplet,
// a random fixed-length crop (audio_processing.py:58-74) and a transpose (:185), single-threaded
// (train_triplet.py:118 num_workers=0), then H2D; its filter step even round-trips the inputs through
// NumPy again (train_triplet.py:265-271).  With the corpus features resident in HBM (VoxCeleb1 fbanks
// are ~35 GB fp32 -- 288 GB holds them) a batch is one gather: out[b, t, :] = feat[start[b] + t, :].
namespace {
__global__ void __launch_bounds__(256) assemble_crops_kernel(const float *feat, const long long *row_start,
                                                             const long long *row_end, float *out, int T, int F) {
    const int b = blockIdx.x;
    const long long r0 = row_start[b], r1 = row_end[b];      // first frame of the crop, end of its utterance
    const int vec = F >> 2;
    for (int i = threadIdx.x; i < T * vec; i += 256) {
        const int t = i / vec, v = i - t * vec;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (r0 + t < r1) val = ((const f32x4 *)(feat + (size_t)(r0 + t) * F))[v];
        ((f32x4 *)(out + ((size_t)b * T + t) * F))[v] = val;       // zero padding past the utterance end
    }
}
}  // namespace

extern "C" int ds_assemble_crops_f32(const float *features, const long long *row_start, const long long *row_end,
                                     float *out, int B, int T, int F, void *stream) {
    DS_REQUIRE(features && row_start && row_end && out, DS_ERR_NULL);
    DS_REQUIRE(B > 0 && T > 0 && F > 0 && (F % 4) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(features) && DS_ALIGNED16(out), DS_ERR_ALIGNMENT);
    DS_LAUNCH(assemble_crops_kernel, B, 256, 0, stream, features, row_start, row_end, out, T, F);
    return ds_last_launch_error();
}

extern "C" int ds_segment_mean_f32(const float *x, const long long *offsets, float *out, int n_seg, void *stream) {
    DS_REQUIRE(x && offsets && out, DS_ERR_NULL);
    DS_REQUIRE(n_seg > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(segment_mean_kernel, ds_ceil_div(n_seg, 256), 256, 0, stream, x, offsets, out, n_seg);
    return ds_last_launch_error();
}

extern "C" int ds_mask_rows(void *x, const int *lens, int B, int H, long long row_bytes, void *stream) {
    DS_REQUIRE(x && lens, DS_ERR_NULL);
    DS_REQUIRE(B > 0 && H > 0 && row_bytes > 0 && (row_bytes % 16) == 0 && B < (1 << 27), DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(x), DS_ERR_ALIGNMENT);
    DS_LAUNCH(mask_rows_kernel, 8 * B, 256, 0, stream, (char *)x, lens, H, row_bytes);
    return ds_last_launch_error();
}

extern "C" int ds_avgpool_time_masked_f32(const float *x, const int *lens, float *pooled, int B, int Hr, int Wc, int C,
                                          void *stream) {
    DS_REQUIRE(x && lens && pooled, DS_ERR_NULL);
    DS_REQUIRE(B > 0 && Hr > 0 && Wc > 0 && C > 0 && (C % 4) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(x) && DS_ALIGNED16(pooled), DS_ERR_ALIGNMENT);
    const long long n = (long long)B * (Wc * C / 4);
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    DS_LAUNCH(avgpool_time_masked_kernel, grid, 256, 0, stream, x, lens, pooled, B, Hr, Wc * C);
    return ds_last_launch_error();
}
