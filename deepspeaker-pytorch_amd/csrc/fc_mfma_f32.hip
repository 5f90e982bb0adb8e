// fc_mfma_f32.hip -- the 2048 -> 512 projection (reference model.py:164,209) and the L2
// normalisation x alpha that follows it (model.py:172-183, 210-213).
//
// f = pooled[B,K] . W^T + b is a GEMM with M = B (a few hundred rows), far too small to fill
// 256 CUs as output tiles alone, so it is split along K: every workgroup multiplies a 32-row x
// 128-column tile over one K-slice with v_mfma_f32_32x32x2_f32, both operands streamed straight
// from L2 (no reuse across waves worth an LDS round trip), and writes its partial tile.  The second
// kernel folds the K-slices in a fixed order (deterministic, no atomics), adds the bias, stores f
// (the backward pass needs it) and writes the normalised embedding in the same pass.
#include <ds_device.h>
#include "ds_common.h"

namespace {

constexpr int CK = DS_CONV_CK;

__device__ __forceinline__ float fc_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += ds_shfl_xor(v, m);
    return v;
}

// grid = m_tiles * n_tiles * S;  block = 256 (4 waves, one 32x32 output tile each)
__global__ void __launch_bounds__(256) fc_splitk_kernel(const float *x, const float *w, float *partial, int B, int K,
                                                        int N, int S, int n_tiles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    int bid = blockIdx.x;
    const int sp = bid % S;
    bid /= S;
    const int nt = bid % n_tiles, mt = bid / n_tiles;
    const int m0 = mt * 32, n0 = nt * 128 + wave * 32;
    const int chunks = K / CK, per = chunks / S;
    const int c0 = sp * per;
    const int row = (m0 + l31 < B) ? m0 + l31 : B - 1;
    const float *xa = x + (size_t)row * K + 4 * lhi;
    const float *wb = w + ((size_t)n0 + l31) * CK + 4 * lhi;
    const size_t wstride = (size_t)N * CK;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int c = c0; c < c0 + per; c += 4) {          // per is a multiple of 4
        f32x4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *(const f32x4 *)(xa + (size_t)(c + u) * CK);
            b[u] = *(const f32x4 *)(wb + (size_t)(c + u) * wstride);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = ds_mfma_32x32x2_f32(a[u][j], b[u][j], acc);
    }
    float *dst = partial + (size_t)sp * B * N;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (m < B) dst[(size_t)m * N + n0 + l31] = acc[r];
    }
}

// one wave per row: f = sum_s partial[s] + bias;  e = alpha * f / sqrt(sum f^2 + eps).
// N <= 512 (the embedding): a lane's N / 64 columns stay in registers between the fold and the normalisation and all of the
// row's partials are requested before the first add (round 6: the row was folded column block by column block, 8 dependent
// rounds of loads, and then READ BACK from the f it had just written -- 15.8 us for 12.6 MB; same sums, same order).
template <int NK>       // columns per lane: N <= 64 * NK
__global__ void __launch_bounds__(256) fc_reduce_l2norm_kernel(const float *partial, const float *bias, float *f,
                                                               float *e, int B, int N, int S, float alpha,
                                                               float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = row < B ? row : B - 1;
    float pv[NK][8];                            // S <= 8 (fc_splits)
#pragma unroll
    for (int q = 0; q < NK; ++q) {
        const int k = lane + 64 * q;
#pragma unroll
        for (int s = 0; s < 8; ++s) pv[q][s] = (s < S && k < N) ? partial[((size_t)s * B + r) * N + k] : 0.f;
    }
    float v[NK];
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < NK; ++q) {
        const int k = lane + 64 * q;
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) t += pv[q][s];         // fixed order: s = 0 .. S-1 (+ zeros)
        if (bias && k < N) t += bias[k];
        v[q] = t;
        if (k < N) {
            if (row < B) f[(size_t)r * N + k] = t;
            ss += t * t;                                    // (column blocks in ascending order, as before)
        }
    }
    ss = fc_wave_sum(ss);
    const float nrm = sqrtf(ss + eps);
    if (row < B && e != nullptr) {
#pragma unroll
        for (int q = 0; q < NK; ++q) {
            const int k = lane + 64 * q;
            if (k < N) e[(size_t)r * N + k] = (v[q] / nrm) * alpha;
        }
    }
}

// any N: column blocks one after the other
__global__ void __launch_bounds__(256) fc_reduce_l2norm_wide_kernel(const float *partial, const float *bias, float *f,
                                                                    float *e, int B, int N, int S, float alpha,
                                                                    float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = row < B ? row : B - 1;
    float ss = 0.f;
    for (int k = lane; k < N; k += 64) {
        float pv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) pv[s] = s < S ? partial[((size_t)s * B + r) * N + k] : 0.f;
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) v += pv[s];
        if (bias) v += bias[k];
        if (row < B) f[(size_t)r * N + k] = v;
        ss += v * v;
    }
    ss = fc_wave_sum(ss);
    const float nrm = sqrtf(ss + eps);
    if (row < B && e != nullptr)
        for (int k = lane; k < N; k += 64) e[(size_t)r * N + k] = (f[(size_t)r * N + k] / nrm) * alpha;
}

static void launch_fc_reduce_l2norm(const float *ws, const float *bias, float *f, float *e, int B, int N, int S, float alpha,
                                    float eps, void *stream) {
    const int grid = ds_ceil_div(B, 4);
    if (N <= 128) DS_LAUNCH(fc_reduce_l2norm_kernel<2>, grid, 256, 0, stream, ws, bias, f, e, B, N, S, alpha, eps);
    else if (N <= 256) DS_LAUNCH(fc_reduce_l2norm_kernel<4>, grid, 256, 0, stream, ws, bias, f, e, B, N, S, alpha, eps);
    else if (N <= 512) DS_LAUNCH(fc_reduce_l2norm_kernel<8>, grid, 256, 0, stream, ws, bias, f, e, B, N, S, alpha, eps);
    else DS_LAUNCH(fc_reduce_l2norm_wide_kernel, grid, 256, 0, stream, ws, bias, f, e, B, N, S, alpha, eps);
}

// The softmax head's epilogue (model.py:220-223 + train_triplet.py:281-285): one wave per row reduces the split-K
// partials (+ bias) into the logits AND takes the row maximum, the log-sum-exp over the n_cls real classes and
// the row's loss = lse - logit[label] in the same pass -- the logits are not read back by a second kernel.
__global__ void __launch_bounds__(256) fc_reduce_ce_kernel(const float *partial, const float *bias, float *logits,
                                                           const long long *labels, float *row_loss, float *lse,
                                                           int B, int N, int n_cls, int S) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = row < B ? row : B - 1;
    float mx = -3.0e38f;
    for (int k = lane; k < N; k += 64) {
        float pv[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) pv[s] = s < S ? partial[((size_t)s * B + r) * N + k] : 0.f;
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) v += pv[s];
        if (bias) v += bias[k];
        if (row < B) logits[(size_t)r * N + k] = v;
        if (k < n_cls) mx = fmaxf(mx, v);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, ds_shfl_xor(mx, m));
    float se = 0.f;
    if (row < B)                                        // a lane re-reads only the columns it wrote itself
        for (int k = lane; k < n_cls; k += 64) se += expf(logits[(size_t)r * N + k] - mx);
    se = fc_wave_sum(se);
    const float l = mx + logf(se);
    if (row < B && lane == 0) {
        lse[row] = l;
        row_loss[row] = l - logits[(size_t)r * N + labels[row]];
    }
}

__global__ void __launch_bounds__(256) fc_mean_kernel(const float *x, float *out, int n) {
    float *scratch = ds_dynamic_lds();
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) acc += x[i];
    acc = fc_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (scratch[0] + scratch[1] + scratch[2] + scratch[3]) / (float)n;
}

// ---- small-batch tail (serving latency: B <= 4 utterances): temporal mean + projection in ONE launch --------------
// At B = 1 the three-launch tail (pool 5 us, split-K GEMM 15 us on 32 workgroups each walking a 128 KB weight slice in
// sequence, fold + norm 10 us) is pure latency.  Here a workgroup first pools ITS utterance into LDS (80 KB of the
// last stage's output -> 2048 means, 20 independent 16-byte loads per thread), then each of its 4 waves takes ONE output
// feature: a 2048-long dot product against that feature's weight row (row-major copy of the fc filter in the pooled
// vector's f*C + c order, ds_pack_fc_weight_rows_f32: eight coalesced 1 KiB loads per wave), folded by a fixed xor tree.
// B x N / 4 workgroups; the norm follows in ds_l2norm_scale_f32's kernel.  Summation order differs from the split-K
// GEMM's: results agree to f32 rounding, not bitwise.
__global__ void __launch_bounds__(256) pool_fc_small_kernel(const float *a, const float *w_rows, const float *bias, float *f,
                                                            int Hr, int K, int N) {
    float *pooled = ds_dynamic_lds();                       // [K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ngroups = N >> 2;
    const int b = blockIdx.x / ngroups, ng = blockIdx.x - b * ngroups;      // B x N/4 workgroups: (utterance, 4 features)
    const int kv = K >> 2;                                  // float4 columns per row
    for (int v = tid; v < kv; v += 256) {
        const f32x4 *src = (const f32x4 *)(a + (size_t)b * Hr * K) + v;
        f32x4 s4 = {0.f, 0.f, 0.f, 0.f};
        for (int h = 0; h < Hr; ++h) s4 += src[(size_t)h * kv];
#pragma unroll
        for (int j = 0; j < 4; ++j) s4[j] = s4[j] / (float)Hr;          // the division avgpool_time_kernel does
        ((f32x4 *)pooled)[v] = s4;
    }
    __syncthreads();
    const int n = ng * 4 + wave;
    const f32x4 *wr = (const f32x4 *)(w_rows + (size_t)n * K);
    const f32x4 *pv = (const f32x4 *)pooled;
    float acc = 0.f;
    for (int v = lane; v < kv; v += 64) {
        const f32x4 x4 = pv[v], w4 = wr[v];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_fmaf(x4[j], w4[j], acc);
    }
    acc = fc_wave_sum(acc);
    if (lane == 0) f[(size_t)b * N + n] = acc + (bias ? bias[n] : 0.0f);
}

// fc weight [N][C*F] (reference order c*F + f) -> [N][K'] with k' = f*C + c: each output feature's filter as ONE
// contiguous row in the order of the pooled channels-last vector
__global__ void __launch_bounds__(256) pack_fc_weight_rows_kernel(const float *w, float *out, int N, int C, int F) {
    const long long n = (long long)N * C * F;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int kp = (int)(i % (C * F)), nn = (int)(i / (C * F));
        const int f_ = kp / C, c = kp - f_ * C;
        out[i] = w[(size_t)nn * C * F + (size_t)c * F + f_];
    }
}

static int fc_splits(int K) {
    const int chunks = K / CK;
    for (int s = 8; s > 1; s >>= 1)
        if (chunks % (4 * s) == 0) return s;
    return 1;
}

}  // namespace

extern "C" long long ds_fc_workspace_floats(int B, int K, int N) {
    if (B <= 0 || K <= 0 || N <= 0 || K % (4 * CK) != 0 || N % 128 != 0) return DS_ERR_BAD_SHAPE;
    return (long long)fc_splits(K) * B * N;
}

extern "C" int ds_fc_l2norm_fwd_f32(const float *pooled, const float *w_packed, const float *bias, float *workspace,
                                    float *f, float *e, int B, int K, int N, float alpha, float eps,
                                    void *stream) {
    DS_REQUIRE(pooled && w_packed && workspace && f, DS_ERR_NULL);
    DS_REQUIRE(B > 0 && K > 0 && N > 0 && K % (4 * CK) == 0 && N % 128 == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(pooled) && DS_ALIGNED16(w_packed), DS_ERR_ALIGNMENT);
    const int S = fc_splits(K), n_tiles = N / 128, m_tiles = ds_ceil_div(B, 32);
    DS_LAUNCH(fc_splitk_kernel, m_tiles * n_tiles * S, 256, 0, stream, pooled, w_packed, workspace, B, K, N, S,
              n_tiles);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    launch_fc_reduce_l2norm((const float *)workspace, bias, f, e, B, N, S, alpha, eps, stream);
    return ds_last_launch_error();
}

// classifier GEMM + cross-entropy in one call: logits [M, N] (N = n_cls padded to 128; pad columns carry zero
// weights), per-row loss and log-sum-exp (kept for the backward pass), loss = mean row loss
extern "C" int ds_fc_ce_fwd_f32(const float *x, const float *w_packed, const float *bias, float *workspace,
                                float *logits, const long long *labels, float *row_loss, float *lse, float *loss,
                                int M, int K, int N, int n_cls, void *stream) {
    DS_REQUIRE(x && w_packed && workspace && logits && labels && row_loss && lse && loss, DS_ERR_NULL);
    DS_REQUIRE(M > 0 && K > 0 && N > 0 && K % (4 * CK) == 0 && N % 128 == 0 && n_cls > 0 && n_cls <= N, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(x) && DS_ALIGNED16(w_packed), DS_ERR_ALIGNMENT);
    const int S = fc_splits(K), n_tiles = N / 128, m_tiles = ds_ceil_div(M, 32);
    DS_LAUNCH(fc_splitk_kernel, m_tiles * n_tiles * S, 256, 0, stream, x, w_packed, workspace, M, K, N, S, n_tiles);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(fc_reduce_ce_kernel, ds_ceil_div(M, 4), 256, 0, stream, (const float *)workspace, bias, logits, labels,
              row_loss, lse, M, N, n_cls, S);
    rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(fc_mean_kernel, 1, 256, 64, stream, (const float *)row_loss, loss, M);
    return ds_last_launch_error();
}

extern "C" int ds_pack_fc_weight_rows_f32(const float *w, float *w_rows, int N, int C, int F, void *stream) {
    DS_REQUIRE(w && w_rows, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && C > 0 && F > 0 && ((C * F) % 4) == 0, DS_ERR_BAD_SHAPE);
    const long long n = (long long)N * C * F;
    long long g = (n + 255) / 256;
    DS_LAUNCH(pack_fc_weight_rows_kernel, (int)(g > 4096 ? 4096 : g), 256, 0, stream, w, w_rows, N, C, F);
    return ds_last_launch_error();
}

// Small-batch tail (B <= DS_TAIL_SMALL_MAX_B): f = mean_t(a) . W^T + bias from the last stage's f32 output a [B, Hr, K]
// (K = Wc * C channels-last), W as ds_pack_fc_weight_rows_f32 laid it out; e = alpha * f / sqrt(sum f^2 + eps).
// Two launches (pool + projection; norm).  model.py:207-213.
extern "C" int ds_tail_small_f32(const float *a, const float *w_rows, const float *bias, float *f, float *e, int B, int Hr,
                                 int K, int N, float alpha, float eps, void *stream) {
    DS_REQUIRE(a && w_rows && f && e, DS_ERR_NULL);
    DS_REQUIRE(B > 0 && B <= DS_TAIL_SMALL_MAX_B && Hr > 0 && K > 0 && (K % 4) == 0 && N > 0 && (N % 4) == 0 &&
                   (size_t)K * 4 <= 64 * 1024, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(a) && DS_ALIGNED16(w_rows), DS_ERR_ALIGNMENT);
    DS_LAUNCH(pool_fc_small_kernel, (N / 4) * B, 256, (size_t)K * 4, stream, a, w_rows, bias, f, Hr, K, N);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    return ds_l2norm_scale_f32(f, e, B, N, alpha, eps, stream);
}
