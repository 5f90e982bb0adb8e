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

// one wave per row: f = sum_s partial[s] + bias;  e = alpha * f / sqrt(sum f^2 + eps)
__global__ void __launch_bounds__(256) fc_reduce_l2norm_kernel(const float *partial, const float *bias, float *f,
                                                               float *e, int B, int N, int S, float alpha,
                                                               float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int r = row < B ? row : B - 1;
    float ss = 0.f;
    for (int k = lane; k < N; k += 64) {
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += partial[((size_t)s * B + r) * N + k];
        if (bias) v += bias[k];
        if (row < B) f[(size_t)r * N + k] = v;
        ss += v * v;
    }
    ss = fc_wave_sum(ss);
    const float nrm = sqrtf(ss + eps);
    if (row < B && e != nullptr)
        for (int k = lane; k < N; k += 64) e[(size_t)r * N + k] = (f[(size_t)r * N + k] / nrm) * alpha;
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
    DS_LAUNCH(fc_reduce_l2norm_kernel, ds_ceil_div(B, 4), 256, 0, stream, (const float *)workspace, bias, f, e, B, N,
              S, alpha, eps);
    return ds_last_launch_error();
}
