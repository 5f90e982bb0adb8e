// tail_loss.hip -- the HBM-bound row kernels around the convolution stack:
//   * temporal average pool (reference model.py:111,207-208)
//   * L2 normalisation x alpha (model.py:172-183,210-213)
//   * PairwiseDistance / TripletMarginLoss forward (model.py:13-18, 27-33)
//   * the triplet filter of the training loop (train_triplet.py:251-262)
// One 64-lane wavefront owns one embedding row; row reductions are wave shuffles.
#include <ds_device.h>
#include "ds_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += ds_shfl_xor(v, m);
    return v;
}

// ---- temporal average pool: [B,Hr,Wc,C] -> [B, Wc*C] -------------------------------------------
__global__ void __launch_bounds__(256) avgpool_time_kernel(const float *x, float *pooled, int B, int Hr,
                                                           int row_elems /* Wc*C */) {
    const int vec_per_row = row_elems >> 2;
    const long long n = (long long)B * vec_per_row;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / vec_per_row), v = (int)(i - (long long)b * vec_per_row);
        const f32x4 *src = (const f32x4 *)(x + (size_t)b * Hr * row_elems) + v;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int h = 0; h < Hr; ++h) s += src[(size_t)h * vec_per_row];
        const float hr = (float)Hr;
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = s[j] / hr;
        ((f32x4 *)(pooled + (size_t)b * row_elems))[v] = s;
    }
}

// ---- e = alpha * f / sqrt(sum f^2 + eps) ---------------------------------------------------------
__global__ void __launch_bounds__(256) l2norm_scale_kernel(const float *f, float *e, int B, int D, float alpha,
                                                           float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = row < B;
    const float *src = f + (size_t)(live ? row : 0) * D;
    float ss = 0.f;
    for (int k = lane; k < D; k += 64) {
        const float v = src[k];
        ss += v * v;
    }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss + eps);
    if (live) {
        float *dst = e + (size_t)row * D;
        for (int k = lane; k < D; k += 64) dst[k] = (src[k] / nrm) * alpha;
    }
}

// ---- max |a - b| and max |b| over two equally shaped tensors (the precision guard's error measure) -----------
// one workgroup, fixed fold order: out[0] = max |a[i] - b[i]|, out[1] = max |b[i]|
__global__ void __launch_bounds__(1024) max_abs_diff_kernel(const float *a, const float *b, long long n, float *out) {
    float (*part)[16] = (float (*)[16])ds_dynamic_lds();          // [2][16]
    float md = 0.f, mb = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) {
        const float vb = b[i];
        // a non-finite difference (a NaN / inf row of either path) must not vanish in the fold: fmaxf drops NaN, so it is
        // turned into +inf here -- the guard then reads an infinite error and escalates
        const float d = fabsf(a[i] - vb);
        md = (d != d) ? __builtin_inff() : fmaxf(md, d);
        mb = fmaxf(mb, fabsf(vb));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        md = fmaxf(md, ds_shfl_xor(md, m));
        mb = fmaxf(mb, ds_shfl_xor(mb, m));
    }
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = md;
        part[1][threadIdx.x >> 6] = mb;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            md = fmaxf(md, part[0][w]);
            mb = fmaxf(mb, part[1][w]);
        }
        out[0] = md;
        out[1] = mb;
    }
}

// ---- pairwise distance rows ------------------------------------------------------------------------
__device__ __forceinline__ float row_sqdist(const float *a, const float *b, int D, int lane) {
    float s = 0.f;
    for (int k = lane; k < D; k += 64) {
        const float d = fabsf(a[k] - b[k]);
        s += d * d;
    }
    return wave_sum(s);
}

__global__ void __launch_bounds__(256) pairwise_distance_kernel(const float *x1, const float *x2, float *d, int N,
                                                                int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = row < N ? row : 0;
    const float s = row_sqdist(x1 + (size_t)r * D, x2 + (size_t)r * D, D, lane);
    if (row < N && lane == 0) d[row] = sqrtf(s + eps);
}

// any norm p > 0 (reference model.py:16-18: pow(pow(|x1 - x2|, p).sum(1) + eps, 1 / p)); the reference itself only uses 2
__global__ void __launch_bounds__(256) pairwise_distance_p_kernel(const float *x1, const float *x2, float *d, int N,
                                                                  int D, float eps, float p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = row < N ? row : 0;
    const float *a = x1 + (size_t)r * D, *b = x2 + (size_t)r * D;
    float s = 0.f;
    for (int k = lane; k < D; k += 64) s += powf(fabsf(a[k] - b[k]), p);
    s = wave_sum(s);
    if (row < N && lane == 0) d[row] = powf(s + eps, 1.0f / p);
}

// d/dx1 of the above = d^(1 - p) * |x1 - x2|^(p - 1) * sign(x1 - x2) (0 where x1 == x2, as torch.abs' gradient); g2 = -g1
__global__ void __launch_bounds__(256) pairwise_distance_p_bwd_kernel(const float *x1, const float *x2, const float *d,
                                                                      const float *gd, float *g1, float *g2, int N,
                                                                      int D, float p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < N) {
        const float c = gd[row] * powf(d[row], 1.0f - p);
        const size_t o = (size_t)row * D;
        for (int k = lane; k < D; k += 64) {
            const float df = x1[o + k] - x2[o + k];
            const float ad = fabsf(df);
            const float g = df == 0.0f ? 0.0f : c * powf(ad, p - 1.0f) * (df > 0.0f ? 1.0f : -1.0f);
            g1[o + k] = g;
            g2[o + k] = -g;
        }
    }
}

__global__ void __launch_bounds__(256) triplet_dist_kernel(const float *a, const float *p, const float *n, float *d_p,
                                                           float *d_n, int N, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = row < N ? row : 0;
    const float sp = row_sqdist(a + (size_t)r * D, p + (size_t)r * D, D, lane);
    const float sn = row_sqdist(a + (size_t)r * D, n + (size_t)r * D, D, lane);
    if (row < N && lane == 0) {
        d_p[row] = sqrtf(sp + eps);
        d_n[row] = sqrtf(sn + eps);
    }
}

// block-wide sum of one float per thread, fixed order (deterministic); result valid in thread 0
__device__ __forceinline__ float block_sum_256(float v, float *scratch) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = scratch[0] + scratch[1] + scratch[2] + scratch[3];
    __syncthreads();
    return r;
}

// loss = mean_i max(0, margin + d_p[i] - d_n[i]); single workgroup, fixed summation order
__global__ void __launch_bounds__(256) hinge_mean_kernel(const float *d_p, const float *d_n, float margin, float *loss,
                                                         int N) {
    float *scratch = ds_dynamic_lds();
    float acc = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) acc += fmaxf(margin + d_p[i] - d_n[i], 0.0f);
    const float tot = block_sum_256(acc, scratch);
    if (threadIdx.x == 0) loss[0] = tot / (float)N;
}

// ordered compaction of {i : d_n[i] - d_p[i] < margin}; single workgroup
__global__ void __launch_bounds__(256) triplet_filter_kernel(const float *d_p, const float *d_n, float margin,
                                                             long long *idx, int *count, float *mean_diff, int N) {
    float *scratch = ds_dynamic_lds();
    int *iscratch = (int *)(scratch + 8);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int base = 0;
    float dsum = 0.f;
    for (int i0 = 0; i0 < N; i0 += 256) {
        const int i = i0 + threadIdx.x;
        float diff = 0.f;
        int sel = 0;
        if (i < N) {
            diff = d_n[i] - d_p[i];
            sel = diff < margin;
        }
        dsum += diff;
        const unsigned long long m = ds_ballot(sel);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) iscratch[wave] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += iscratch[w];
        const int tot = iscratch[0] + iscratch[1] + iscratch[2] + iscratch[3];
        if (sel) idx[base + woff + before] = i;
        base += tot;
        __syncthreads();
    }
    const float tot = block_sum_256(dsum, scratch);
    if (threadIdx.x == 0) {
        count[0] = base;
        mean_diff[0] = tot / (float)N;
    }
}

// ---- the whole scalar side of the triplet step in ONE single-workgroup pass over d_p / d_n ------------------
// (model.py:27-33 and train_triplet.py:253-262): loss = mean hinge, the ordered filter {i : d_n - d_p < margin},
// mean(d_n - d_p), and -- new -- the ordered list of NEAR TIES |d_n - d_p - margin| < band (at most amb_cap
// entries; amb_count is the true count).  Slots the near ties leave unused hold index 0 (a valid gather index), or,
// with probe_base >= 0, PROBE triplets (probe_base + k) mod N, k = 0, 1, ...: the refinement re-embeds every slot
// whether it is used or not, so the unused ones sample the error of the fp16 decision variable for free
// (refine_distances_kernel reports it; mining.RefinePolicy sizes the band from it).
// Fixed summation order, ordered block scans: deterministic.
__global__ void __launch_bounds__(256) triplet_scan_kernel(const float *d_p, const float *d_n, float margin, float band,
                                                           float *loss, long long *idx, int *count, float *mean_diff,
                                                           long long *amb_idx, int *amb_count, int amb_cap, int N,
                                                           int probe_base) {
    float *scratch = ds_dynamic_lds();
    int *iscratch = (int *)(scratch + 8);       // [2][4] wave totals
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (amb_idx)
        for (int i = threadIdx.x; i < amb_cap; i += 256) amb_idx[i] = 0;
    int base = 0, abase = 0;
    float dsum = 0.f, hsum = 0.f;
    for (int i0 = 0; i0 < N; i0 += 256) {
        const int i = i0 + threadIdx.x;
        float diff = 0.f, hinge = 0.f;
        int sel = 0, amb = 0;
        if (i < N) {
            const float dp = d_p[i], dn = d_n[i];
            diff = dn - dp;
            sel = diff < margin;                                   // train_triplet.py:253
            hinge = fmaxf(margin + dp - dn, 0.0f);                 // model.py:30-31
            amb = fabsf(diff - margin) < band;
        }
        dsum += diff;
        hsum += hinge;
        const unsigned long long m = ds_ballot(sel), ma = ds_ballot(amb);
        const unsigned long long below = (1ull << lane) - 1ull;
        const int before = __popcll(m & below), abefore = __popcll(ma & below);
        if (lane == 0) {
            iscratch[wave] = __popcll(m);
            iscratch[4 + wave] = __popcll(ma);
        }
        __syncthreads();
        int woff = 0, awoff = 0;
        for (int w = 0; w < wave; ++w) {
            woff += iscratch[w];
            awoff += iscratch[4 + w];
        }
        const int tot = iscratch[0] + iscratch[1] + iscratch[2] + iscratch[3];
        const int atot = iscratch[4] + iscratch[5] + iscratch[6] + iscratch[7];
        if (sel) idx[base + woff + before] = i;
        if (amb && amb_idx && abase + awoff + abefore < amb_cap) amb_idx[abase + awoff + abefore] = i;
        base += tot;
        abase += atot;
        __syncthreads();
    }
    const float dtot = block_sum_256(dsum, scratch);
    const float htot = block_sum_256(hsum, scratch);
    if (threadIdx.x == 0) {
        count[0] = base;
        mean_diff[0] = dtot / (float)N;
        loss[0] = htot / (float)N;
        if (amb_count) amb_count[0] = abase;
    }
    if (amb_idx && probe_base >= 0)             // (abase is uniform; every near-tie slot was written above)
        for (int s = abase + (int)threadIdx.x; s < amb_cap; s += 256) amb_idx[s] = (probe_base + (s - abase)) % N;
}

// Near-tie refinement: slot s < min(amb_count, cap) holds triplet i = amb_idx[s], whose three utterances were
// re-embedded at f32-class precision into e_ref rows (s, cap + s, 2 cap + s); its distances are replaced.
// PROBES: with err != nullptr (ONE workgroup then) EVERY slot is live (the scan filled the unused ones with probe
// triplets) and err[0] = the largest |(d_n - d_p)_f32-class - (d_n - d_p)_before| over all slots -- the error the fp16
// forward made on the filter's decision variable, sampled on near ties and probes alike; err[1] = the slot count;
// and, given the fp16 path's own embeddings emb_a / emb_p / emb_n [N][D], err[2] = max |e_ref - emb| and err[3] =
// max |e_ref| over the 3 * cap sampled rows (their ratio is the tests' embedding-error measure, max |d| / max |ref|, on
// the sample: how the path watches its own distance to the 1e-3 contract).
__global__ void __launch_bounds__(256) refine_distances_kernel(const float *e_ref, const long long *amb_idx,
                                                               const int *amb_count, int cap, float *d_p, float *d_n,
                                                               int D, float eps, float *err, const float *d_p0,
                                                               const float *d_n0, const float *emb_a, const float *emb_p,
                                                               const float *emb_n, int n_copy) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int live_n = err ? cap : (amb_count[0] < cap ? amb_count[0] : cap);
    if (n_copy > 0) {           // (one workgroup) d_p / d_n start as copies of the unpatched distances: no clone launches
        for (int i = threadIdx.x; i < n_copy; i += 256) {
            d_p[i] = d_p0[i];
            d_n[i] = d_n0[i];
        }
        __syncthreads();
    }
    float worst = 0.f, ediff = 0.f, emax = 0.f;
    for (int s = blockIdx.x * 4 + wave; s < cap; s += gridDim.x * 4) {
        const float *a = e_ref + (size_t)s * D, *p = e_ref + (size_t)(cap + s) * D, *n = e_ref + (size_t)(2 * cap + s) * D;
        const float sp = row_sqdist(a, p, D, lane);
        const float sn = row_sqdist(a, n, D, lane);
        if (err && emb_a) {                     // (every slot is live here)
            const long long i = amb_idx[s];
            const float *r3[3] = {a, p, n}, *m3[3] = {emb_a + (size_t)i * D, emb_p + (size_t)i * D, emb_n + (size_t)i * D};
#pragma unroll
            for (int t = 0; t < 3; ++t)
                for (int k = lane; k < D; k += 64) {
                    const float rv = r3[t][k];
                    ediff = fmaxf(ediff, fabsf(rv - m3[t][k]));
                    emax = fmaxf(emax, fabsf(rv));
                }
        }
        if (s < live_n && lane == 0) {
            const long long i = amb_idx[s];
            const float dp = sqrtf(sp + eps), dn = sqrtf(sn + eps);
            // "before" is read from the untouched originals: a probe may name the same triplet as a near tie (or the
            // list may repeat an index), and another wave may already have patched d_p[i] / d_n[i]
            if (err) worst = fmaxf(worst, fabsf((dn - dp) - (d_n0[i] - d_p0[i])));
            d_p[i] = dp;
            d_n[i] = dn;
        }
    }
    if (err) {                                  // gridDim.x == 1
        float *scratch = ds_dynamic_lds();      // [3][4]
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {     // (lanes hold partial maxima of the embedding comparison)
            ediff = fmaxf(ediff, ds_shfl_xor(ediff, m));
            emax = fmaxf(emax, ds_shfl_xor(emax, m));
        }
        if (lane == 0) {
            scratch[wave] = worst;
            scratch[4 + wave] = ediff;
            scratch[8 + wave] = emax;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            err[0] = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
            err[1] = (float)cap;
            err[2] = fmaxf(fmaxf(scratch[4], scratch[5]), fmaxf(scratch[6], scratch[7]));
            err[3] = fmaxf(fmaxf(scratch[8], scratch[9]), fmaxf(scratch[10], scratch[11]));
            if (n_copy > 0) err[4] = (float)amb_count[0];       // the near-tie count rides in the same read-back
        }
    }
}

// ---- backward of the loss side (autograd of model.py:13-18, 27-33) -------------------------------
// d = sqrt(sum (x1-x2)^2 + eps)  ->  dx1 = gd * (x1-x2)/d, dx2 = -dx1
__global__ void __launch_bounds__(256) pairwise_distance_bwd_kernel(const float *x1, const float *x2, const float *d,
                                                                    const float *gd, float *g1, float *g2, int N,
                                                                    int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < N) {
        const float c = gd[row] / d[row];
        const size_t o = (size_t)row * D;
        for (int k = lane; k < D; k += 64) {
            const float g = c * (x1[o + k] - x2[o + k]);
            g1[o + k] = g;
            g2[o + k] = -g;
        }
    }
}

// loss = mean_i max(0, margin + d_p - d_n); clamp(min=0) passes gradient where its input >= 0
__global__ void __launch_bounds__(256) triplet_margin_bwd_kernel(const float *a, const float *p, const float *n,
                                                                 const float *d_p, const float *d_n, float margin,
                                                                 const float *gloss, float *ga, float *gp, float *gn,
                                                                 int N, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row < N) {
        const float act = (margin + d_p[row] - d_n[row]) >= 0.0f ? gloss[0] / (float)N : 0.0f;
        const float cp = act / d_p[row], cn = act / d_n[row];
        const size_t o = (size_t)row * D;
        for (int k = lane; k < D; k += 64) {
            const float av = a[o + k];
            const float tp = cp * (av - p[o + k]);
            const float tn = cn * (av - n[o + k]);
            ga[o + k] = tp - tn;
            gp[o + k] = -tp;
            gn[o + k] = tn;
        }
    }
}

// f -> e = alpha f / |f|:  gf = alpha * (ge / nrm - f * <ge,f> / nrm^3)
__global__ void __launch_bounds__(256) l2norm_scale_bwd_kernel(const float *f, const float *ge, float *gf, int B, int D,
                                                               float alpha, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t o = (size_t)(row < B ? row : 0) * D;
    float ss = 0.f, dot = 0.f;
    for (int k = lane; k < D; k += 64) {
        const float v = f[o + k];
        ss += v * v;
        dot += v * ge[o + k];
    }
    ss = wave_sum(ss);
    dot = wave_sum(dot);
    const float nrm = sqrtf(ss + eps);
    const float inv = 1.0f / nrm, c = dot / (nrm * nrm * nrm);
    if (row < B)
        for (int k = lane; k < D; k += 64) gf[o + k] = alpha * (ge[o + k] * inv - f[o + k] * c);
}

// pooled-gradient broadcast: gx[b,h,:,:] = gpooled[b,:] / Hr for every h, gated by the clip mask of the
// stage output `out` (0 < out < 20), i.e. the backward of clip -> mean over time in one pass
__global__ void __launch_bounds__(256) avgpool_time_bwd_kernel(const float *gpooled, const float *out, float *gx, int B,
                                                               int Hr, int row_elems) {
    const int vec_per_row = row_elems >> 2;
    const long long n = (long long)B * Hr * vec_per_row;
    const float hr = (float)Hr;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int v = (int)(i % vec_per_row);
        const int b = (int)(i / ((long long)Hr * vec_per_row));
        f32x4 g = ((const f32x4 *)(gpooled + (size_t)b * row_elems))[v];
        const f32x4 o = ((const f32x4 *)out)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = (o[j] > 0.0f && o[j] < 20.0f) ? g[j] / hr : 0.0f;
        ((f32x4 *)gx)[i] = g;
    }
}

// ---- cross-GPU semi-hard negative search (NEW capability named by BASELINE.json north_star; the
//      reference only filters pre-sampled triplets).  For anchor i: among candidates j with
//      cand_label[j] != anchor_label[i], the closest one that is farther than d_p[i]; if none is
//      semi-hard, the closest overall; ties -> lowest j; -1 when every candidate shares the label.
//      A workgroup owns 8 anchors (rows in LDS) x one tile of 256 candidates, one candidate per thread:
//      32-dimension slabs of the candidate tile are staged through LDS (coalesced 128-byte rows in,
//      conflict-free padded columns out), so every candidate element is read from L2 once per 8 anchors
//      and the anchor values are LDS broadcasts.  Per-tile winners go to a workspace; a second kernel
//      folds the tiles in ascending order.  Fixed scan / reduction order => deterministic.
//      (Round 4: a squarer tile -- 16 anchors x 64 candidates per workgroup, a sixth of the L2 traffic -- was built and
//      measured: 40 vs 41 us at 768 candidates, 135 vs 93 us at 6144.  The kernel is bound by its per-slab barriers,
//      not by L2: the wide candidate tile amortises them better.  Not kept.)
constexpr int MINE_A_MAX = 8;      // anchors per workgroup: 8, or 4 / 2 when 8 would leave the chip under-filled
constexpr int MINE_K = 32;         // dimensions per staged slab
constexpr int MINE_C = 256;        // candidates per workgroup
constexpr int MINE_P = MINE_K + 4;  // candidate-tile row pitch in floats (16-byte aligned, conflict-free)

template <int MINE_A>
__global__ void __launch_bounds__(256) mine_semihard_kernel(const float *anchor, const float *d_p,
                                                            const long long *anchor_label, const float *cand,
                                                            const long long *cand_label, float *partial,
                                                            int N, int M, int D, float eps, int n_agroups) {
    float *lds = ds_dynamic_lds();
    float *arow = lds;                                     // [D][MINE_A]: the 8 anchors' values of one dimension are adjacent
    float *ctile = arow + MINE_A * D;                      // [256][MINE_P]
    // the reduction scratch aliases the candidate tile (dead by then): 2 x [2][MINE_A][256] words
    float *red_d = ctile;
    int *red_j = (int *)(ctile + 2 * MINE_A * 256);
    const int tid = threadIdx.x;
    const int ag = blockIdx.x % n_agroups, ct = blockIdx.x / n_agroups;
    const int a0 = ag * MINE_A, j0 = ct * MINE_C;
    for (int i = tid; i < MINE_A * D; i += 256) {          // coalesced rows in, dimension-major out
        const int a = i / D, k = i - a * D;
        arow[k * MINE_A + a] = (a0 + a < N) ? anchor[(size_t)(a0 + a) * D + k] : 0.0f;
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[MINE_A / 2];                                // anchors (2p, 2p+1): packed f32 math, two anchors per lane op
#pragma unroll
    for (int a = 0; a < MINE_A / 2; ++a) acc2[a] = f32x2{0.0f, 0.0f};
    // software pipeline over 32-dimension slabs: the next slab's rows are in registers while this one is used
    constexpr int SLOTS = MINE_C * (MINE_K / 4) / 256;     // 8 float4 per thread and slab
    f32x4 pre[SLOTS];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int it = 0; it < SLOTS; ++it) {
            const int i = tid + it * 256;
            const int row = i / (MINE_K / 4), q = i - row * (MINE_K / 4);
            pre[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j0 + row < M && k0 + q * 4 < D) pre[it] = *(const f32x4 *)(cand + (size_t)(j0 + row) * D + k0 + q * 4);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < D; k0 += MINE_K) {
        __syncthreads();                                   // previous slab consumed (and the anchor rows are written)
#pragma unroll
        for (int it = 0; it < SLOTS; ++it) {
            const int i = tid + it * 256;
            const int row = i / (MINE_K / 4), q = i - row * (MINE_K / 4);
            *(f32x4 *)(ctile + row * MINE_P + q * 4) = pre[it];
        }
        __syncthreads();
        if (k0 + MINE_K < D) fetch(k0 + MINE_K);
        const int kmax = (D - k0) < MINE_K ? (D - k0) : MINE_K;
        // four dimensions per step: one 16-byte read of this thread's candidate (row pitch 36 floats: the
        // 16-lane groups of a ds_read_b128 cover all 64 banks) and, per dimension, two broadcast reads of the
        // 8 anchors; the sum over dimensions stays strictly sequential per (anchor, candidate)
        for (int k = 0; k < kmax; k += 4) {
            const f32x4 c = *(const f32x4 *)(ctile + tid * MINE_P + k);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x2 cc = {c[u], c[u]};
                const float *ar = arow + (k0 + k + u) * MINE_A;
                if constexpr (MINE_A >= 4) {
                    const f32x4 a03 = *(const f32x4 *)ar;
                    const f32x2 d0 = f32x2{a03[0], a03[1]} - cc, d1 = f32x2{a03[2], a03[3]} - cc;
                    acc2[0] = d0 * d0 + acc2[0];
                    acc2[1] = d1 * d1 + acc2[1];
                } else {
                    const f32x2 d0 = *(const f32x2 *)ar - cc;
                    acc2[0] = d0 * d0 + acc2[0];
                }
                if constexpr (MINE_A == 8) {
                    const f32x4 a47 = *(const f32x4 *)(ar + 4);
                    const f32x2 d2 = f32x2{a47[0], a47[1]} - cc, d3 = f32x2{a47[2], a47[3]} - cc;
                    acc2[2] = d2 * d2 + acc2[2];
                    acc2[3] = d3 * d3 + acc2[3];
                }
            }
        }
    }
    float accd[MINE_A];
#pragma unroll
    for (int a = 0; a < MINE_A; ++a) accd[a] = acc2[a >> 1][a & 1];
    __syncthreads();
    const int j = j0 + tid;
    const long long lc = j < M ? cand_label[j] : 0;
#pragma unroll
    for (int a = 0; a < MINE_A; ++a) {
        const int ai = a0 + a < N ? a0 + a : N - 1;
        const float d = sqrtf(accd[a] + eps);
        const bool other = j < M && lc != anchor_label[ai];
        const bool semi = other && d > d_p[ai];
        red_d[(0 * MINE_A + a) * 256 + tid] = semi ? d : 3.0e38f;  red_j[(0 * MINE_A + a) * 256 + tid] = semi ? j : -1;
        red_d[(1 * MINE_A + a) * 256 + tid] = other ? d : 3.0e38f; red_j[(1 * MINE_A + a) * 256 + tid] = other ? j : -1;
    }
    __syncthreads();
    if (tid < 2 * MINE_A) {                                // one thread per (criterion, anchor) folds the tile
        const int a = tid % MINE_A, crit = tid / MINE_A;
        if (a0 + a < N) {
            float bd = 3.0e38f;
            int bj = -1;
            for (int t = 0; t < 256; ++t) {                // ascending j: strict < keeps the lowest index on ties
                const float dd = red_d[(crit * MINE_A + a) * 256 + t];
                const int jj = red_j[(crit * MINE_A + a) * 256 + t];
                if (jj >= 0 && dd < bd) { bd = dd; bj = jj; }
            }
            float *dst = partial + (((size_t)ct * N) + a0 + a) * 4 + crit * 2;
            dst[0] = bd;
            dst[1] = __int_as_float(bj);
        }
    }
}

// fold the candidate tiles in ascending order; semi-hard winner if any, else the closest other-speaker one
__global__ void __launch_bounds__(256) mine_merge_kernel(const float *partial, long long *out, float *out_d, int N,
                                                         int n_ctiles) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float bs = 3.0e38f, ba = 3.0e38f;
    int js = -1, ja = -1;
    for (int ct = 0; ct < n_ctiles; ++ct) {
        const float *src = partial + ((size_t)ct * N + i) * 4;
        const int s_j = __float_as_int(src[1]), a_j = __float_as_int(src[3]);
        if (s_j >= 0 && src[0] < bs) { bs = src[0]; js = s_j; }
        if (a_j >= 0 && src[2] < ba) { ba = src[2]; ja = a_j; }
    }
    out[i] = js >= 0 ? js : ja;
    if (out_d) out_d[i] = js >= 0 ? bs : (ja >= 0 ? ba : 0.0f);
}

// dst[i,:] = src[idx[i],:]  (idx < 0 -> zeros)
__global__ void __launch_bounds__(256) gather_rows_kernel(const float *src, const long long *idx, float *dst, int N,
                                                          int D) {
    // long rows (whole utterances: 10240 floats) are split over several workgroups: blockIdx = row * parts + part
    const int parts = gridDim.x / N;
    const int i = blockIdx.x / parts, part = blockIdx.x - i * parts;
    const long long j = idx[i];
    for (int k = part * 256 + threadIdx.x; k < D; k += parts * 256)
        dst[(size_t)i * D + k] = j >= 0 ? src[(size_t)j * D + k] : 0.0f;
}

// the same rows of three sources in one launch (the anchors / positives / negatives of the near-tie triplets):
// dst[k][i,:] = src_k[idx[i],:]; blockIdx = (k * N + i) * parts + part
__global__ void __launch_bounds__(256) gather_rows3_kernel(const float *s0, const float *s1, const float *s2,
                                                           const long long *idx, float *dst, int N, int D) {
    const int parts = gridDim.x / (3 * N);
    const int r = blockIdx.x / parts, part = blockIdx.x - r * parts;
    const int k = r / N, i = r - k * N;
    const float *src = k == 0 ? s0 : k == 1 ? s1 : s2;
    const long long j = idx[i];
    for (int e = (part * 256 + threadIdx.x) * 4; e < D; e += parts * 1024) {
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (j >= 0) v = *(const f32x4 *)(src + (size_t)j * D + e);
        *(f32x4 *)(dst + (size_t)r * D + e) = v;
    }
}

// dst[j,:] (+)= sum over {i : idx[i] == j} of g[i,:], i ascending (deterministic; one workgroup per dst row)
__global__ void __launch_bounds__(256) scatter_add_rows_kernel(const float *g, const long long *idx, float *dst, int N,
                                                               int D, int accumulate) {
    const int j = blockIdx.x;
    for (int k = threadIdx.x; k < D; k += 256) {
        float s = accumulate ? dst[(size_t)j * D + k] : 0.0f;
        for (int i = 0; i < N; ++i)
            if (idx[i] == j) s += g[(size_t)i * D + k];
        dst[(size_t)j * D + k] = s;
    }
}

}  // namespace

extern "C" int ds_gather_rows_f32(const float *src, const long long *idx, float *dst, int N, int D, void *stream) {
    DS_REQUIRE(src && idx && dst, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && D > 0, DS_ERR_BAD_SHAPE);
    const int parts = D >= 4096 ? 8 : 1;
    DS_LAUNCH(gather_rows_kernel, N * parts, 256, 0, stream, src, idx, dst, N, D);
    return ds_last_launch_error();
}

// dst [3][N][D] = rows idx[0..N) of src_a, src_p, src_n (a negative index gathers zeros); D % 4 == 0, 16-byte aligned
extern "C" int ds_gather_rows3_f32(const float *src_a, const float *src_p, const float *src_n, const long long *idx,
                                   float *dst, int N, int D, void *stream) {
    DS_REQUIRE(src_a && src_p && src_n && idx && dst, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && D > 0 && D % 4 == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(src_a) && DS_ALIGNED16(src_p) && DS_ALIGNED16(src_n) && DS_ALIGNED16(dst), DS_ERR_ALIGNMENT);
    const int parts = D >= 4096 ? 8 : 1;
    DS_LAUNCH(gather_rows3_kernel, 3 * N * parts, 256, 0, stream, src_a, src_p, src_n, idx, dst, N, D);
    return ds_last_launch_error();
}

extern "C" int ds_scatter_add_rows_f32(const float *g, const long long *idx, float *dst, int N, int M, int D,
                                       int accumulate, void *stream) {
    DS_REQUIRE(g && idx && dst, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && M > 0 && D > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(scatter_add_rows_kernel, M, 256, 0, stream, g, idx, dst, N, D, accumulate);
    return ds_last_launch_error();
}

extern "C" long long ds_mine_workspace_floats(int N, int M) {
    if (N <= 0 || M <= 0) return DS_ERR_BAD_SHAPE;
    return (long long)ds_ceil_div(M, MINE_C) * N * 4;
}

extern "C" int ds_mine_semihard_f32(const float *anchor, const float *d_p, const long long *anchor_label,
                                    const float *cand, const long long *cand_label, float *workspace,
                                    long long *out_index, float *out_dist, int N, int M, int D, void *stream) {
    DS_REQUIRE(anchor && d_p && anchor_label && cand && cand_label && workspace && out_index, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && M > 0 && D > 0 && D <= 8192, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(D % 4 == 0 && DS_ALIGNED16(cand) && DS_ALIGNED16(anchor), DS_ERR_ALIGNMENT);
    const float eps = (float)(1e-4 / (double)D);
    const int n_ctiles = ds_ceil_div(M, MINE_C);
    // 8 anchors per workgroup re-use every staged candidate slab 8 times; 4 or 2 anchors per workgroup give more
    // workgroups (same sums, same order)
    // ... but only when that leaves the chip nearly empty (fewer workgroups than a QUARTER of the CUs).  Round 6: the search
    // runs on a side stream next to the next step's persistent convolutions, where what it costs is CU-time, not its own
    // latency -- 256 anchors x 768 candidates as 96 workgroups of 8 anchors take 41 us (4 k CU-us), as 384 workgroups of 2
    // anchors 35 us (13 k CU-us): the bench step ran 2.00 ms against 2.05 - 2.10 (`gpurun_out/r06_run11`).
    int A = MINE_A_MAX;
    while (A > 2 && ds_ceil_div(N, A) * n_ctiles < ds_cu_count() / 4) A /= 2;
    auto lds_of = [&](int a_) {
        const size_t tile_words = 256 * MINE_P > 4 * a_ * 256 ? 256 * MINE_P : 4 * a_ * 256;
        return ((size_t)a_ * D + tile_words) * 4;
    };
    while (A > 2 && lds_of(A) > 64 * 1024) A /= 2;         // long rows: fewer anchors per workgroup fit next to the tile
    const size_t lds = lds_of(A);
    DS_REQUIRE(lds <= 64 * 1024, DS_ERR_BAD_SHAPE);
    const int n_agroups = ds_ceil_div(N, A);
    if (A == 8)
        DS_LAUNCH(mine_semihard_kernel<8>, n_agroups * n_ctiles, 256, lds, stream, anchor, d_p, anchor_label, cand, cand_label,
                  workspace, N, M, D, eps, n_agroups);
    else if (A == 4)
        DS_LAUNCH(mine_semihard_kernel<4>, n_agroups * n_ctiles, 256, lds, stream, anchor, d_p, anchor_label, cand, cand_label,
                  workspace, N, M, D, eps, n_agroups);
    else
        DS_LAUNCH(mine_semihard_kernel<2>, n_agroups * n_ctiles, 256, lds, stream, anchor, d_p, anchor_label, cand, cand_label,
                  workspace, N, M, D, eps, n_agroups);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(mine_merge_kernel, ds_ceil_div(N, 256), 256, 0, stream, (const float *)workspace, out_index, out_dist, N,
              n_ctiles);
    return ds_last_launch_error();
}

extern "C" int ds_pairwise_distance_bwd_f32(const float *x1, const float *x2, const float *d, const float *gd,
                                            float *g1, float *g2, int N, int D, void *stream) {
    DS_REQUIRE(x1 && x2 && d && gd && g1 && g2, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && D > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(pairwise_distance_bwd_kernel, ds_ceil_div(N, 4), 256, 0, stream, x1, x2, d, gd, g1, g2, N, D);
    return ds_last_launch_error();
}

extern "C" int ds_triplet_margin_bwd_f32(const float *a, const float *p, const float *n, const float *d_p,
                                         const float *d_n, float margin, const float *grad_loss, float *ga,
                                         float *gp, float *gn, int N, int D, void *stream) {
    DS_REQUIRE(a && p && n && d_p && d_n && grad_loss && ga && gp && gn, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && D > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(triplet_margin_bwd_kernel, ds_ceil_div(N, 4), 256, 0, stream, a, p, n, d_p, d_n, margin, grad_loss, ga,
              gp, gn, N, D);
    return ds_last_launch_error();
}

extern "C" int ds_l2norm_scale_bwd_f32(const float *f, const float *ge, float *gf, int B, int D, float alpha,
                                       float eps, void *stream) {
    DS_REQUIRE(f && ge && gf, DS_ERR_NULL);
    DS_REQUIRE(B > 0 && D > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(l2norm_scale_bwd_kernel, ds_ceil_div(B, 4), 256, 0, stream, f, ge, gf, B, D, alpha, eps);
    return ds_last_launch_error();
}

extern "C" int ds_avgpool_time_bwd_f32(const float *gpooled, const float *out, float *gx, int B, int Hr, int Wc,
                                       int C, void *stream) {
    DS_REQUIRE(gpooled && out && gx, DS_ERR_NULL);
    DS_REQUIRE(B > 0 && Hr > 0 && Wc > 0 && C > 0 && (C % 4) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(gpooled) && DS_ALIGNED16(out) && DS_ALIGNED16(gx), DS_ERR_ALIGNMENT);
    const long long n = (long long)B * Hr * (Wc * C / 4);
    long long g = (n + 255) / 256;
    DS_LAUNCH(avgpool_time_bwd_kernel, (int)(g > 4096 ? 4096 : g), 256, 0, stream, gpooled, out, gx, B, Hr, Wc * C);
    return ds_last_launch_error();
}

extern "C" int ds_avgpool_time_f32(const float *x, float *pooled, int B, int Hr, int Wc, int C, void *stream) {
    DS_REQUIRE(x && pooled, DS_ERR_NULL);
    DS_REQUIRE(B > 0 && Hr > 0 && Wc > 0 && C > 0 && (C % 4) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(x) && DS_ALIGNED16(pooled), DS_ERR_ALIGNMENT);
    const long long n = (long long)B * (Wc * C / 4);
    int grid = (int)((n + 255) / 256);
    if (grid > 2048) grid = 2048;
    DS_LAUNCH(avgpool_time_kernel, grid, 256, 0, stream, x, pooled, B, Hr, Wc * C);
    return ds_last_launch_error();
}

extern "C" int ds_l2norm_scale_f32(const float *f, float *e, int B, int D, float alpha, float eps, void *stream) {
    DS_REQUIRE(f && e, DS_ERR_NULL);
    DS_REQUIRE(B > 0 && D > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(l2norm_scale_kernel, ds_ceil_div(B, 4), 256, 0, stream, f, e, B, D, alpha, eps);
    return ds_last_launch_error();
}

extern "C" int ds_max_abs_diff_f32(const float *a, const float *b, long long n, float *out2, void *stream) {
    DS_REQUIRE(a && b && out2, DS_ERR_NULL);
    DS_REQUIRE(n > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(max_abs_diff_kernel, 1, 1024, 2 * 16 * sizeof(float), stream, a, b, n, out2);
    return ds_last_launch_error();
}

extern "C" int ds_pairwise_distance_f32(const float *x1, const float *x2, float *d, int N, int D, void *stream) {
    DS_REQUIRE(x1 && x2 && d, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && D > 0, DS_ERR_BAD_SHAPE);
    const float eps = (float)(1e-4 / (double)D);     // model.py:15
    DS_LAUNCH(pairwise_distance_kernel, ds_ceil_div(N, 4), 256, 0, stream, x1, x2, d, N, D, eps);
    return ds_last_launch_error();
}

extern "C" int ds_pairwise_distance_p_f32(const float *x1, const float *x2, float *d, int N, int D, float p,
                                          void *stream) {
    DS_REQUIRE(x1 && x2 && d, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && D > 0 && p > 0.0f, DS_ERR_BAD_SHAPE);
    const float eps = (float)(1e-4 / (double)D);     // model.py:15
    DS_LAUNCH(pairwise_distance_p_kernel, ds_ceil_div(N, 4), 256, 0, stream, x1, x2, d, N, D, eps, p);
    return ds_last_launch_error();
}

extern "C" int ds_pairwise_distance_p_bwd_f32(const float *x1, const float *x2, const float *d, const float *gd,
                                              float *g1, float *g2, int N, int D, float p, void *stream) {
    DS_REQUIRE(x1 && x2 && d && gd && g1 && g2, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && D > 0 && p > 0.0f, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(pairwise_distance_p_bwd_kernel, ds_ceil_div(N, 4), 256, 0, stream, x1, x2, d, gd, g1, g2, N, D, p);
    return ds_last_launch_error();
}

extern "C" int ds_triplet_margin_fwd_f32(const float *a, const float *p, const float *n, float margin, float *d_p,
                                         float *d_n, float *loss, int N, int D, void *stream) {
    DS_REQUIRE(a && p && n && d_p && d_n && loss, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && D > 0, DS_ERR_BAD_SHAPE);
    const float eps = (float)(1e-4 / (double)D);
    DS_LAUNCH(triplet_dist_kernel, ds_ceil_div(N, 4), 256, 0, stream, a, p, n, d_p, d_n, N, D, eps);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(hinge_mean_kernel, 1, 256, 64, stream, (const float *)d_p, (const float *)d_n, margin, loss, N);
    return ds_last_launch_error();
}

extern "C" int ds_triplet_filter_f32(const float *d_p, const float *d_n, float margin, long long *idx, int *count,
                                     float *mean_diff, int N, void *stream) {
    DS_REQUIRE(d_p && d_n && idx && count && mean_diff, DS_ERR_NULL);
    DS_REQUIRE(N > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(triplet_filter_kernel, 1, 256, 64, stream, d_p, d_n, margin, idx, count, mean_diff, N);
    return ds_last_launch_error();
}

// The triplet step's loss side in two launches: distances (one wave per row), then ONE scan over the 2N scalars
// for the loss, the filter, the mean difference and the near-tie list.  Shared by TripletMarginLoss.forward and
// select_triplets, so the distances are computed once per step.
extern "C" int ds_triplet_tail_probe_f32(const float *a, const float *p, const float *n, float margin, float band,
                                         float *d_p, float *d_n, float *loss, long long *idx, int *count, float *mean_diff,
                                         long long *amb_idx, int *amb_count, int amb_cap, int probe_base, int N, int D,
                                         void *stream) {
    DS_REQUIRE(a && p && n && d_p && d_n && loss && idx && count && mean_diff, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && D > 0 && amb_cap >= 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(amb_cap == 0 || (amb_idx && amb_count), DS_ERR_NULL);
    const float eps = (float)(1e-4 / (double)D);
    DS_LAUNCH(triplet_dist_kernel, ds_ceil_div(N, 4), 256, 0, stream, a, p, n, d_p, d_n, N, D, eps);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(triplet_scan_kernel, 1, 256, 64, stream, (const float *)d_p, (const float *)d_n, margin,
              amb_cap > 0 ? band : -1.0f, loss, idx, count, mean_diff, amb_cap > 0 ? amb_idx : (long long *)nullptr,
              amb_cap > 0 ? amb_count : (int *)nullptr, amb_cap, N, amb_cap > 0 ? probe_base : -1);
    return ds_last_launch_error();
}

extern "C" int ds_triplet_tail_f32(const float *a, const float *p, const float *n, float margin, float band,
                                   float *d_p, float *d_n, float *loss, long long *idx, int *count, float *mean_diff,
                                   long long *amb_idx, int *amb_count, int amb_cap, int N, int D, void *stream) {
    return ds_triplet_tail_probe_f32(a, p, n, margin, band, d_p, d_n, loss, idx, count, mean_diff, amb_idx, amb_count,
                                     amb_cap, -1, N, D, stream);
}

// the scan alone, over distances that already exist (after ds_refine_distances_f32 patched the near ties)
extern "C" int ds_triplet_scan_f32(const float *d_p, const float *d_n, float margin, float *loss, long long *idx,
                                   int *count, float *mean_diff, int N, void *stream) {
    DS_REQUIRE(d_p && d_n && loss && idx && count && mean_diff, DS_ERR_NULL);
    DS_REQUIRE(N > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(triplet_scan_kernel, 1, 256, 64, stream, d_p, d_n, margin, -1.0f, loss, idx, count, mean_diff,
              (long long *)nullptr, (int *)nullptr, 0, N, -1);
    return ds_last_launch_error();
}

static int refine_distances(const float *e_ref, const long long *amb_idx, const int *amb_count, int cap, float *d_p,
                            float *d_n, int D, float *err, const float *d_p0, const float *d_n0, const float *emb_a,
                            const float *emb_p, const float *emb_n, void *stream, int n_copy = 0) {
    DS_REQUIRE(e_ref && amb_idx && amb_count && d_p && d_n, DS_ERR_NULL);
    DS_REQUIRE(cap > 0 && D > 0, DS_ERR_BAD_SHAPE);
    const float eps = (float)(1e-4 / (double)D);
    const int blocks = err ? 1 : ds_ceil_div(cap, 4);       // the error read-out folds inside one workgroup: no atomics
    DS_LAUNCH(refine_distances_kernel, blocks, 256, 64, stream, e_ref, amb_idx, amb_count, cap, d_p, d_n, D, eps, err, d_p0,
              d_n0, emb_a, emb_p, emb_n, n_copy);
    return ds_last_launch_error();
}

extern "C" int ds_refine_distances_f32(const float *e_ref, const long long *amb_idx, const int *amb_count, int cap,
                                       float *d_p, float *d_n, int D, void *stream) {
    return refine_distances(e_ref, amb_idx, amb_count, cap, d_p, d_n, D, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                            stream);
}

// ... over ALL cap slots (near ties and the probe triplets ds_triplet_tail_probe_f32 put into the unused ones), also
// reporting err[0] = max |change of d_n - d_p|, err[1] = the number of slots sampled and -- emb_a / emb_p / emb_n given
// (the path's own embeddings [N][D]; all three or none) -- err[2] = max |e_ref - emb|, err[3] = max |e_ref| over the sampled
// rows (all float; err holds 4).  d_p / d_n are patched; d_p_before / d_n_before are the unpatched distances the slots were
// chosen on (other buffers than d_p / d_n).
extern "C" int ds_refine_distances_probe_f32(const float *e_ref, const long long *amb_idx, const int *amb_count, int cap,
                                             float *d_p, float *d_n, const float *d_p_before, const float *d_n_before,
                                             const float *emb_a, const float *emb_p, const float *emb_n, int D, float *err,
                                             void *stream) {
    DS_REQUIRE(err && d_p_before && d_n_before, DS_ERR_NULL);
    DS_REQUIRE(d_p_before != d_p && d_n_before != d_n, DS_ERR_UNSUPPORTED);
    DS_REQUIRE((emb_a != nullptr) == (emb_p != nullptr) && (emb_a != nullptr) == (emb_n != nullptr), DS_ERR_NULL);
    return refine_distances(e_ref, amb_idx, amb_count, cap, d_p, d_n, D, err, d_p_before, d_n_before, emb_a, emb_p, emb_n, stream);
}

// The same in ONE launch for what used to be five on the refinement's side stream (two clones, the patch, two read-back
// copies; round 6: beside the persistent convolutions every side-stream launch waits for the drain of a main-stream
// launch): d_p / d_n (N each) are WRITTEN -- copies of d_p_before / d_n_before with the cap slots patched -- and err holds
// FIVE floats, err[4] = the near-tie count (amb_count[0]), so that one asynchronous copy brings everything back.
extern "C" int ds_refine_distances_fused_f32(const float *e_ref, const long long *amb_idx, const int *amb_count, int cap,
                                             float *d_p, float *d_n, const float *d_p_before, const float *d_n_before,
                                             const float *emb_a, const float *emb_p, const float *emb_n, int N, int D,
                                             float *err5, void *stream) {
    DS_REQUIRE(err5 && d_p_before && d_n_before && amb_count, DS_ERR_NULL);
    DS_REQUIRE(d_p_before != d_p && d_n_before != d_n, DS_ERR_UNSUPPORTED);
    DS_REQUIRE(N > 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE((emb_a != nullptr) == (emb_p != nullptr) && (emb_a != nullptr) == (emb_n != nullptr), DS_ERR_NULL);
    return refine_distances(e_ref, amb_idx, amb_count, cap, d_p, d_n, D, err5, d_p_before, d_n_before, emb_a, emb_p, emb_n, stream,
                            N);
}

// ---- softmax cross-entropy over the classifier logits (reference train_triplet.py:281-287:
//      nn.CrossEntropyLoss() on cat[cls_a, cls_p, cls_n], mean reduction) -- one wave per row ----
namespace {

__device__ __forceinline__ float ce_wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, ds_shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float ce_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += ds_shfl_xor(v, m);
    return v;
}

// row_loss[i] = logsumexp(logits[i,:]) - logits[i, label[i]];  lse[i] kept for the backward pass
__global__ void __launch_bounds__(256) ce_rows_kernel(const float *logits, const long long *labels, float *row_loss,
                                                      float *lse, int M, int n_cls, int ld) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = row < M ? row : M - 1;
    const float *x = logits + (size_t)r * ld;
    float mx = -3.0e38f;
    for (int k = lane; k < n_cls; k += 64) mx = fmaxf(mx, x[k]);
    mx = ce_wave_max(mx);
    float s = 0.f;
    for (int k = lane; k < n_cls; k += 64) s += expf(x[k] - mx);
    s = ce_wave_sum(s);
    const float l = mx + logf(s);
    if (row < M && lane == 0) {
        lse[row] = l;
        row_loss[row] = l - x[labels[row]];
    }
}

__global__ void __launch_bounds__(256) mean_kernel(const float *x, float *out, int N) {
    float *scratch = ds_dynamic_lds();
    float acc = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) acc += x[i];
    acc = ce_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (scratch[0] + scratch[1] + scratch[2] + scratch[3]) / (float)N;
}

// dlogits[i,k] = gloss/M * (softmax(logits[i,:])[k] - [k == label[i]]);  columns >= n_cls (padding) = 0
__global__ void __launch_bounds__(256) ce_bwd_kernel(const float *logits, const long long *labels, const float *lse,
                                                     const float *gloss, float *dlogits, int M, int n_cls, int ld,
                                                     int ld_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float g = gloss[0] / (float)M;
    const float *x = logits + (size_t)row * ld;
    const float l = lse[row];
    const int lab = (int)labels[row];
    float *d = dlogits + (size_t)row * ld_out;
    for (int k = lane; k < ld_out; k += 64) {
        float v = 0.f;
        if (k < n_cls) v = g * (expf(x[k] - l) - (k == lab ? 1.0f : 0.0f));
        d[k] = v;
    }
}

}  // namespace

extern "C" int ds_cross_entropy_fwd_f32(const float *logits, const long long *labels, float *row_loss, float *lse,
                                        float *loss, int M, int n_cls, int ld, void *stream) {
    DS_REQUIRE(logits && labels && row_loss && lse && loss, DS_ERR_NULL);
    DS_REQUIRE(M > 0 && n_cls > 0 && ld >= n_cls, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(ce_rows_kernel, ds_ceil_div(M, 4), 256, 0, stream, logits, labels, row_loss, lse, M, n_cls, ld);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(mean_kernel, 1, 256, 64, stream, (const float *)row_loss, loss, M);
    return ds_last_launch_error();
}

extern "C" int ds_cross_entropy_bwd_f32(const float *logits, const long long *labels, const float *lse,
                                        const float *grad_loss, float *dlogits, int M, int n_cls, int ld, int ld_out,
                                        void *stream) {
    DS_REQUIRE(logits && labels && lse && grad_loss && dlogits, DS_ERR_NULL);
    DS_REQUIRE(M > 0 && n_cls > 0 && ld >= n_cls && ld_out >= n_cls, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(ce_bwd_kernel, ds_ceil_div(M, 4), 256, 0, stream, logits, labels, lse, grad_loss, dlogits, M, n_cls, ld,
              ld_out);
    return ds_last_launch_error();
}
