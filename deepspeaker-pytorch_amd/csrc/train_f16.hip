// train_f16.hip -- the element-wise side of the OPT-IN fp16 training step (DeepSpeakerModel(train_precision="f16")):
// train-mode BatchNorm forward and backward (reference model.py:59,62,70,74,94,... under model.train(); autograd as run by
// loss.backward(), train_triplet.py:223) over fp16 tensors in HBM, f32 arithmetic inside.
//
// Why a separate family: the f32-class step (bf16x3) keeps f32 activations and gradients in HBM -- 6 ms of its 18 ms are
// HBM-bound BatchNorm / clip passes over them.  Here every activation, pre-activation and gradient tensor is fp16 (half
// the bytes per pass), the convolutions are the eval path's fp16 matrix-core kernels (one MFMA per product), and the
// statistics are taken by a pass of their own over the fp16 pre-activation (2 bytes per element -- cheaper than the f32
// path's epilogue-fused sums were to give up).  Gradient tensors hold S * g (a constant loss scale, a power of two) so that
// the small gradients of the early layers stay inside fp16's normal range; dgamma / dbeta and the filter gradients are
// un-scaled where they leave in f32.
//
// Batch = G members (anchor / positive / negative forwards of a triplet step, train_triplet.py:215) with their own batch
// statistics: member m owns pixels [m * n_pix, (m + 1) * n_pix) and row m of every [G][C] table.
#include <ds_device.h>
#include "ds_common.h"
#include <stdlib.h>

namespace {

typedef _Float16 h16;
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

// 8 channels (16 bytes of fp16) per thread and access: half the memory instructions of 4-channel accesses for the same
// bytes in flight (measured on the first build, 4 channels per access: 2.6 - 3.2 TB/s on the stage-1 tensors)
__device__ __forceinline__ f32x8 ld8(const h16 *p, size_t i8) { return __builtin_convertvector(((const h16x8 *)p)[i8], f32x8); }
__device__ __forceinline__ void st8(h16 *p, size_t i8, f32x8 v) { ((h16x8 *)p)[i8] = __builtin_convertvector(v, h16x8); }
__device__ __forceinline__ f32x8 ld8f(const float *p, size_t i8) {
    const f32x4 a = ((const f32x4 *)p)[2 * i8], b = ((const f32x4 *)p)[2 * i8 + 1];
    return f32x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}
__device__ __forceinline__ void st8f(float *p, size_t i8, f32x8 v) {
    ((f32x4 *)p)[2 * i8] = f32x4{v[0], v[1], v[2], v[3]};
    ((f32x4 *)p)[2 * i8 + 1] = f32x4{v[4], v[5], v[6], v[7]};
}

constexpr int TF_FOLD_R = 128, TF_FOLD_C = 2;   // row lanes x channels per workgroup of the partial-sum folds below
constexpr int TF_MAX_ROWS = 768;                // partial rows per member (x G members: enough workgroups, short folds)

// per-block partial sums of two per-channel quantities -> partial[blockIdx.x][c][2]; red: [slots][C][2] floats
__device__ __forceinline__ void block_fold(float *red, const f32x8 &s1, const f32x8 &s2, int slot, int slots, int cg, int C,
                                           float *partial) {
    if (slot < slots) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            red[((slot * C) + cg * 8 + j) * 2 + 0] = s1[j];
            red[((slot * C) + cg * 8 + j) * 2 + 1] = s2[j];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a1 = 0.f, a2 = 0.f;
        for (int s = 0; s < slots; ++s) {
            a1 += red[(s * C + c) * 2 + 0];
            a2 += red[(s * C + c) * 2 + 1];
        }
        partial[((size_t)blockIdx.x * C + c) * 2 + 0] = a1;
        partial[((size_t)blockIdx.x * C + c) * 2 + 1] = a2;
    }
}

// partial[member][blk][c] = { sum z, sum z^2 } over the block's pixels (f32 sums of fp16 values)
__global__ void __launch_bounds__(256) bn_stats_f16_kernel(const h16 *z, float *partial, long long n_pix, int C,
                                                           int pix_per_block, int blocks_per_member) {
    const int member = blockIdx.x / blocks_per_member, mblock = blockIdx.x - member * blocks_per_member;
    z += (size_t)member * n_pix * C;
    float *red = ds_dynamic_lds();                         // [slots][C][2]
    const int cvec = C >> 3;
    const int slots = 256 / cvec;
    const int cg = threadIdx.x % cvec, slot = threadIdx.x / cvec;
    const long long p0 = (long long)mblock * pix_per_block;
    long long p1 = p0 + pix_per_block;
    if (p1 > n_pix) p1 = n_pix;
    f32x8 s1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2 = s1;
    if (slot < slots)
#pragma unroll 2
        for (long long p = p0 + slot; p < p1; p += slots) {
            const f32x8 v = ld8(z, (size_t)p * cvec + cg);
            s1 += v;
            s2 += v * v;
        }
    block_fold(red, s1, s2, slot, slots, cg, C, partial);
}

// fold of one member's partial rows for TF_FOLD_C channels per workgroup, double precision, fixed order: a thread adds
// its rows (rl, rl + TF_FOLD_R, ...), the 32 row lanes of a wave are folded by an xor tree of shuffles (lane = 2 rl + cl:
// offsets 2 .. 32), the four waves' results meet in LDS.  (Round 6: the last step used to be ONE thread adding 128 LDS
// values per channel and member in sequence -- 16 - 35 us for a launch that moves a few hundred KB, 24 of them on the
// critical path of every fp16 training step.)
__device__ __forceinline__ bool fold_rows(const float *partial, int n_partial, int C, double *red, int c0, int &c, double &t1,
                                          double &t2) {
    static_assert(TF_FOLD_C == 2 && TF_FOLD_R == 128, "lane = 2 * row lane + channel; four waves");
    const int cl = threadIdx.x % TF_FOLD_C, rl = threadIdx.x / TF_FOLD_C;
    c = c0 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int r = rl; r < n_partial; r += TF_FOLD_R) {
            const float *src = partial + ((size_t)r * C + c) * 2;
            s1 += (double)src[0];
            s2 += (double)src[1];
        }
#pragma unroll
    for (int m = 2; m <= 32; m <<= 1) {
        s1 += ds_shfl_xor_f64(s1, m);
        s2 += ds_shfl_xor_f64(s2, m);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                                        // (the previous member's fold has been read)
    if (lane < TF_FOLD_C) {
        red[(wave * TF_FOLD_C + cl) * 2 + 0] = s1;
        red[(wave * TF_FOLD_C + cl) * 2 + 1] = s2;
    }
    __syncthreads();
    if (rl != 0 || c >= C) return false;
    t1 = t2 = 0.0;
    for (int k = 0; k < 4; ++k) {
        t1 += red[(k * TF_FOLD_C + cl) * 2 + 0];
        t2 += red[(k * TF_FOLD_C + cl) * 2 + 1];
    }
    return true;
}

// All G members of one BatchNorm layer in one launch: batch mean / invstd / (scale, shift) per member into [G][C]
// tables, and the running statistics updated member after member IN CALL ORDER (the reference's model(data_a),
// model(data_p), model(data_n) are three nn.BatchNorm2d.train() calls: three momentum updates, model.py:188).
__global__ void __launch_bounds__(256) bn_stats_finalize_group_kernel(const float *partial, int n_partial, double count,
                                                                      const float *gamma, const float *beta, float eps,
                                                                      float momentum, float *running_mean,
                                                                      float *running_var, float *mean_t, float *invstd_t,
                                                                      float *scale_t, float *shift_t, int C, int G) {
    double *red = (double *)ds_dynamic_lds();              // [TF_FOLD_R][TF_FOLD_C][2]
    for (int m = 0; m < G; ++m) {
        int c;
        double t1, t2;
        if (fold_rows(partial + (size_t)m * n_partial * C * 2, n_partial, C, red, (int)blockIdx.x * TF_FOLD_C, c, t1, t2)) {
            const double mean = t1 / count;
            double var = t2 / count - mean * mean;         // biased (normalisation) variance
            if (var < 0.0) var = 0.0;
            const double invstd = 1.0 / sqrt(var + (double)eps);
            const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
            if (running_mean) {
                running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
                running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
            }
            const size_t o = (size_t)m * C + c;
            mean_t[o] = (float)mean;
            invstd_t[o] = (float)invstd;
            const double s = (double)gamma[c] * invstd;
            scale_t[o] = (float)s;
            shift_t[o] = (float)((double)beta[c] - mean * s);
        }
    }
}

// Element-wise passes walk [G members][n_vec_member] 8-channel vectors.  256 threads per workgroup and a channel-group
// count that divides 256 (C / 8 = 8 .. 64): with every workgroup starting at a multiple of 256 vectors inside a member, a
// thread keeps ONE channel group for the whole pass -- its table rows are loaded once per member, and the loop carries
// no division (the first build computed member = i / n and c = i % (C/8) in 64-bit arithmetic per vector: the passes ran
// at 2.4 - 2.9 TB/s with the wave-cycle counters showing waits on instruction issue, not on memory).
// y = clip(z * scale[m] + shift[m] (+ residual)), fp16 in; fp16 or f32 out
template <bool OUT32>
__global__ void __launch_bounds__(256) bn_apply_f16_kernel(const h16 *z, const float *scale_t, const float *shift_t,
                                                           const h16 *res, void *y, long long n_vec_member, int G, int C,
                                                           int flags) {
    const int cvec = C >> 3;
    const int c8 = threadIdx.x & (cvec - 1);                // (cvec is a power of two dividing 256: checked by the host)
    const bool with_res = (flags & DS_EPI_RESIDUAL) != 0;
    const float lo = (flags & DS_EPI_CLIP) ? 0.0f : -__builtin_inff(), hi = (flags & DS_EPI_CLIP) ? 20.0f : __builtin_inff();
    for (int member = 0; member < G; ++member) {
        const f32x8 sc = ld8f(scale_t + (size_t)member * C, c8), sh = ld8f(shift_t + (size_t)member * C, c8);
        const size_t mbase = (size_t)member * (size_t)n_vec_member;
#pragma unroll 2
        for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n_vec_member; v += (long long)gridDim.x * 256) {
            const size_t i = mbase + (size_t)v;
            f32x8 t = ld8(z, i);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = ds_bn_affine(t[j], sc[j], sh[j]);
            if (with_res) t += ld8(res, i);
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = fminf(fmaxf(t[j], lo), hi);
            if constexpr (OUT32) st8f((float *)y, i, t);
            else st8((h16 *)y, i, t);
        }
    }
}

// The masked upstream gradient of one 8-channel group: g = (g1 [+ g2]) * mask.  The mask is the clipped ReLU's, taken
//   MODE 0: from nothing (g1 is already masked),
//   MODE 1: from the stored activation `act` (fp16, or f32 with ACT32): 0 < act < 20 -- the layers whose activation had a
//           residual added before the clip,
//   MODE 2: from the layer's own pre-activation: a = fp16(clip(z * msc + msh)) recomputed exactly as bn_apply_f16_kernel
//           stored it, 0 < a < 20 -- no third tensor is read.
// PARITY: g1 is the output of the 5x5 stride-2 data gradient run as ONE 3x3 convolution with 4 C output channels
// (ds_pack_conv_weight_dgrad_f16, stride 2): [B][Ho2][Wo2][2][2][C] -- pixel (h, w) of this layer's [H][W] map is parity
// class (h & 1, w & 1) of cell (h >> 1, w >> 1).
struct BwdGeom { long long n_pix; int img_pix; int W, Ho2, Wo2, cvec; float rcp_img, rcp_w; };
template <int MODE, bool PARITY, bool ACT32>
__device__ __forceinline__ f32x8 masked_grad(const h16 *g1, const h16 *g2, const void *act, const f32x8 &zv, const f32x8 &msc,
                                             const f32x8 &msh, size_t i, long long gp, int cg, const BwdGeom &q) {
    f32x8 g;
    if constexpr (PARITY) {                     // (pixel indices stay below 2^24: reciprocal division, ds_div_small)
        const int b = ds_div_small((int)gp, q.img_pix, q.rcp_img);
        const int rem = (int)gp - b * q.img_pix;
        const int h = ds_div_small(rem, q.W, q.rcp_w), w = rem - h * q.W;
        const size_t cell = ((size_t)b * q.Ho2 + (h >> 1)) * q.Wo2 + (w >> 1);
        g = ld8(g1, (cell * 4 + (size_t)((h & 1) * 2 + (w & 1))) * q.cvec + cg);
    } else {
        g = ld8(g1, i);
    }
    if (g2) g += ld8(g2, i);
    if constexpr (MODE == 1) {
        f32x8 a;
        if constexpr (ACT32) a = ld8f((const float *)act, i);
        else a = ld8((const h16 *)act, i);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = (a[j] > 0.0f && a[j] < 20.0f) ? g[j] : 0.0f;
    } else if constexpr (MODE == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float a = (float)(h16)fminf(fmaxf(ds_bn_affine(zv[j], msc[j], msh[j]), 0.0f), 20.0f);
            g[j] = (a > 0.0f && a < 20.0f) ? g[j] : 0.0f;
        }
    }
    return __builtin_convertvector(__builtin_convertvector(g, h16x8), f32x8);      // as an fp16 tensor would hold it
}

// Backward, first half: partial[member][blk][c] = { sum gy, sum gy * xhat }, xhat = (z - mean) * invstd, gy as above; gy is
// also written out unless gy == nullptr (the second half then recomputes it: bn_bwd_apply_f16_kernel with REGEN).
template <int MODE, bool PARITY, bool ACT32>
__global__ void __launch_bounds__(256) bn_bwd_reduce_f16_kernel(const h16 *g1, const h16 *g2, const void *act, const h16 *z,
                                                                const float *mean, const float *invstd, const float *msc_t,
                                                                const float *msh_t, h16 *gy, float *partial, long long n_pix,
                                                                int C, int pix_per_block, int blocks_per_member, int H,
                                                                int W) {
    const int member = blockIdx.x / blocks_per_member, mblock = blockIdx.x - member * blocks_per_member;
    const size_t moff8 = ((size_t)member * n_pix * C) >> 3;
    float *red = ds_dynamic_lds();                         // [slots][C][2]
    const int cvec = C >> 3;
    const int slots = 256 / cvec;
    const int cg = threadIdx.x % cvec, slot = threadIdx.x / cvec;
    const long long p0 = (long long)mblock * pix_per_block;
    long long p1 = p0 + pix_per_block;
    if (p1 > n_pix) p1 = n_pix;
    const BwdGeom q = {n_pix, H * W, W, (H + 1) >> 1, (W + 1) >> 1, cvec, 1.0f / (float)(H * W), 1.0f / (float)W};
    f32x8 s1 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2 = s1;
    if (slot < slots) {
        const f32x8 mu = ld8f(mean + (size_t)member * C, cg), is = ld8f(invstd + (size_t)member * C, cg);
        f32x8 msc = s1, msh = s1;
        if constexpr (MODE == 2) {
            msc = ld8f(msc_t + (size_t)member * C, cg);
            msh = ld8f(msh_t + (size_t)member * C, cg);
        }
#pragma unroll 2
        for (long long p = p0 + slot; p < p1; p += slots) {
            const size_t i = moff8 + (size_t)p * cvec + cg;                 // 8-channel index in the [G * n_pix][C] tensors
            const f32x8 zv = ld8(z, i);
            const f32x8 g = masked_grad<MODE, PARITY, ACT32>(g1, g2, act, zv, msc, msh, i, (long long)member * n_pix + p, cg, q);
            if (gy) st8(gy, i, g);
            s1 += g;
            s2 += g * ((zv - mu) * is);
        }
    }
    block_fold(red, s1, s2, slot, slots, cg, C, partial);
}

// Backward, fold: per member coef = { gamma * invstd, sum gy / N, sum gy * xhat / N } (in the gradient tensors' scaled
// units), and dgamma / dbeta = the members' sums added in member order, UN-scaled (inv_scale = 1 / S).
__global__ void __launch_bounds__(256) bn_bwd_finalize_f16_kernel(const float *partial, int n_partial, double count,
                                                                  const float *gamma, const float *invstd_t, float *coef,
                                                                  float *ggamma, float *gbeta, int C, int G,
                                                                  float inv_scale) {
    double *red = (double *)ds_dynamic_lds();
    double gg = 0.0, gb = 0.0;
    int c = 0;
    bool mine = false;
    for (int m = 0; m < G; ++m) {
        double t1, t2;
        if (fold_rows(partial + (size_t)m * n_partial * C * 2, n_partial, C, red, (int)blockIdx.x * TF_FOLD_C, c, t1, t2)) {
            mine = true;
            float *cf = coef + (size_t)m * 3 * C;
            cf[c] = gamma[c] * invstd_t[(size_t)m * C + c];
            cf[C + c] = (float)(t1 / count);
            cf[2 * C + c] = (float)(t2 / count);
            gb += (double)(float)t1;                        // (the f32 path adds the members' f32 sums)
            gg += (double)(float)t2;
        }
    }
    if (mine) {
        ggamma[c] = (float)(gg * (double)inv_scale);
        gbeta[c] = (float)(gb * (double)inv_scale);
    }
}

// data-parallel forward: the [G][2C+1] float64 sums (per member C pairs { sum z, sum z^2 } and the pixel count) have been
// all-reduced over the ranks; tables per member, running statistics member after member (call order) -- one launch
__global__ void __launch_bounds__(256) bn_stats_from_sums_group_kernel(const double *sums, const float *gamma, const float *beta,
                                                                       float eps, float momentum, float *running_mean,
                                                                       float *running_var, float *mean_t, float *invstd_t,
                                                                       float *scale_t, float *shift_t, int C, int G) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    for (int m = 0; m < G; ++m) {
        const double *sm = sums + (size_t)m * (2 * C + 1);
        const double count = sm[2 * C];
        const double mean = sm[c * 2] / count;
        double var = sm[c * 2 + 1] / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
        if (running_mean) {
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
        const size_t o = (size_t)m * C + c;
        mean_t[o] = (float)mean;
        invstd_t[o] = (float)invstd;
        const double sc = (double)gamma[c] * invstd;
        scale_t[o] = (float)sc;
        shift_t[o] = (float)((double)beta[c] - mean * sc);
    }
}

// data-parallel fold: the [G][2C+1] float64 sums (per member C pairs { sum gy, sum gy * xhat } and the pixel count) have
// been all-reduced over the ranks; coefficients per member, dgamma / dbeta summed over the members and un-scaled
__global__ void __launch_bounds__(256) bn_bwd_from_sums_f16_kernel(const double *sums, const float *gamma, const float *invstd_t,
                                                                   float *coef, float *ggamma, float *gbeta, int C, int G,
                                                                   float inv_scale) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double gg = 0.0, gb = 0.0;
    for (int m = 0; m < G; ++m) {
        const double *sm = sums + (size_t)m * (2 * C + 1);
        const double count = sm[2 * C];
        float *cf = coef + (size_t)m * 3 * C;
        cf[c] = gamma[c] * invstd_t[(size_t)m * C + c];
        cf[C + c] = (float)(sm[c * 2] / count);
        cf[2 * C + c] = (float)(sm[c * 2 + 1] / count);
        gb += (double)(float)sm[c * 2];
        gg += (double)(float)sm[c * 2 + 1];
    }
    ggamma[c] = (float)(gg * (double)inv_scale);
    gbeta[c] = (float)(gb * (double)inv_scale);
}

// Backward, second half: gz = gamma * invstd * (gy - mean(gy) - xhat * mean(gy * xhat)), fp16 in / out.
// REGEN: gy was not stored: it is recomputed from g1 (not parity-laid-out, no g2) with the MODE-2 mask.
template <bool REGEN>
__global__ void __launch_bounds__(256) bn_bwd_apply_f16_kernel(const h16 *gy, const h16 *z, const float *mean_t,
                                                               const float *invstd_t, const float *coef, const float *msc_t,
                                                               const float *msh_t, h16 *gz, long long n_vec_member, int G,
                                                               int C) {
    const int cvec = C >> 3;
    const int c8 = threadIdx.x & (cvec - 1);
    const BwdGeom q = {0, 1, 1, 1, 1, cvec, 1.0f, 1.0f};
    const f32x8 zero8 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int member = 0; member < G; ++member) {
        const float *cf = coef + (size_t)member * 3 * C;
        const f32x8 mu = ld8f(mean_t + (size_t)member * C, c8), is = ld8f(invstd_t + (size_t)member * C, c8);
        const f32x8 k1 = ld8f(cf, c8), k2 = ld8f(cf + C, c8), k3 = ld8f(cf + 2 * C, c8);
        f32x8 msc = zero8, msh = zero8;
        if constexpr (REGEN) {
            msc = ld8f(msc_t + (size_t)member * C, c8);
            msh = ld8f(msh_t + (size_t)member * C, c8);
        }
        const size_t mbase = (size_t)member * (size_t)n_vec_member;
#pragma unroll 2
        for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n_vec_member; v += (long long)gridDim.x * 256) {
            const size_t i = mbase + (size_t)v;
            const f32x8 zv = ld8(z, i);
            f32x8 g;
            if constexpr (REGEN) g = masked_grad<2, false, false>(gy, nullptr, nullptr, zv, msc, msh, i, 0, c8, q);
            else g = ld8(gy, i);
            st8(gz, i, k1 * (g - k2 - ((zv - mu) * is) * k3));
        }
    }
}

__global__ void __launch_bounds__(256) scale_cast_f16_kernel(const float *x, h16 *y, long long n_vec, float scale) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (long long)gridDim.x * 256)
        st8(y, (size_t)i, ld8f(x, (size_t)i) * scale);
}

static int tf_grid(long long n) {
    static int cap = 0;
    if (cap == 0) {                             // DS_TF_GRID: experiment knob (tools/bn16_probe.py)
        const char *e = getenv("DS_TF_GRID");
        cap = e ? atoi(e) : 8192;
        if (cap < 1) cap = 8192;
    }
    long long g = (n + 255) / 256;
    return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

static bool tf_shape_ok(long long n_pix, int C, int G) {
    const int cvec = C / 8;
    return n_pix > 0 && G > 0 && G <= 64 && C >= 8 && (C % 8) == 0 && C <= 1024 && 256 % cvec == 0 && (cvec & (cvec - 1)) == 0;
}

// partial rows (= workgroups) per member of the reductions: ~8 pixel steps per thread, at most TF_MAX_ROWS
static int tf_rows(long long n_pix, int C) {
    long long ppb = 16384 / C;
    if (ppb < 8) ppb = 8;
    long long blocks = (n_pix + ppb - 1) / ppb;
    if (blocks > TF_MAX_ROWS) blocks = TF_MAX_ROWS;
    return (int)blocks;
}

}  // namespace

// partial rows per member that ds_bn_stats_group_f16 / ds_bn_bwd_group_f16 write (sizes their `partial` scratch)
extern "C" int ds_bn_f16_partial_rows(long long n_pix, int C) {
    if (n_pix <= 0 || C <= 0) return DS_ERR_BAD_SHAPE;
    return tf_rows(n_pix, C);
}

// Train-mode BatchNorm statistics of G members over an fp16 pre-activation z [G * n_pix][C]: partial sums (rows =
// ds_bn_f16_partial_rows(n_pix, C) per member), then tables [G][C] of mean / invstd / scale / shift and the running
// statistics updated member after member (call order).  `partial`: G * rows * C * 2 floats of scratch.
extern "C" int ds_bn_stats_group_f16(const void *z_f16, float *partial, long long n_pix, const float *gamma,
                                     const float *beta, float eps, float momentum, float *running_mean,
                                     float *running_var, float *mean_t, float *invstd_t, float *scale_t, float *shift_t,
                                     int C, int G, void *stream) {
    DS_REQUIRE(z_f16 && partial && gamma && beta && mean_t && invstd_t && scale_t && shift_t, DS_ERR_NULL);
    DS_REQUIRE(tf_shape_ok(n_pix, C, G), DS_ERR_BAD_SHAPE);
    DS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), DS_ERR_NULL);
    DS_REQUIRE(DS_ALIGNED16(z_f16), DS_ERR_ALIGNMENT);
    const int blocks = tf_rows(n_pix, C);
    const int ppb = (int)((n_pix + blocks - 1) / blocks);
    const int slots = 256 / (C / 8);
    DS_LAUNCH(bn_stats_f16_kernel, blocks * G, 256, (size_t)slots * C * 2 * 4, stream, (const h16 *)z_f16, partial, n_pix, C,
              ppb, blocks);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(bn_stats_finalize_group_kernel, ds_ceil_div(C, TF_FOLD_C), 256, TF_FOLD_R * TF_FOLD_C * 2 * sizeof(double), stream,
              (const float *)partial, blocks, (double)n_pix, gamma, beta, eps, momentum, running_mean, running_var, mean_t,
              invstd_t, scale_t, shift_t, C, G);
    return ds_last_launch_error();
}

// Data-parallel split of the above (one process per GPU): only the partial sums { sum z, sum z^2 } per member.  The caller
// folds them to float64 (ds_partial_sum_f64_group, n_partial = ds_bn_f16_partial_rows), all-reduces the [G][2C+1] sums over
// RCCL and finishes each member with ds_bn_stats_from_sums_f32: every rank normalises with the GLOBAL batch's statistics.
extern "C" int ds_bn_stats_partial_f16(const void *z_f16, float *partial, long long n_pix, int C, int G, void *stream) {
    DS_REQUIRE(z_f16 && partial, DS_ERR_NULL);
    DS_REQUIRE(tf_shape_ok(n_pix, C, G), DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(z_f16), DS_ERR_ALIGNMENT);
    const int blocks = tf_rows(n_pix, C);
    const int ppb = (int)((n_pix + blocks - 1) / blocks);
    const int slots = 256 / (C / 8);
    DS_LAUNCH(bn_stats_f16_kernel, blocks * G, 256, (size_t)slots * C * 2 * 4, stream, (const h16 *)z_f16, partial, n_pix, C,
              ppb, blocks);
    return ds_last_launch_error();
}

// ... and the second half: [G][C] tables (and the running statistics, members in call order) from the all-reduced sums
extern "C" int ds_bn_stats_from_sums_group_f32(const double *sums, const float *gamma, const float *beta, float eps,
                                               float momentum, float *running_mean, float *running_var, float *mean_t,
                                               float *invstd_t, float *scale_t, float *shift_t, int C, int G, void *stream) {
    DS_REQUIRE(sums && gamma && beta && mean_t && invstd_t && scale_t && shift_t, DS_ERR_NULL);
    DS_REQUIRE(C > 0 && G > 0 && G <= 64, DS_ERR_BAD_SHAPE);
    DS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), DS_ERR_NULL);
    DS_LAUNCH(bn_stats_from_sums_group_kernel, ds_ceil_div(C, 256), 256, 0, stream, sums, gamma, beta, eps, momentum, running_mean,
              running_var, mean_t, invstd_t, scale_t, shift_t, C, G);
    return ds_last_launch_error();
}

// y = clip(z * scale[m] + shift[m] (+ residual)) for all G members in one launch; y is fp16, or f32 with DS_EPI_OUT_F32
// (the last stage hands f32 to the pooling / projection tail).  flags: DS_EPI_RESIDUAL | DS_EPI_CLIP | DS_EPI_OUT_F32.
extern "C" int ds_bn_apply_group_f16(const void *z_f16, const float *scale_t, const float *shift_t, const void *res_f16,
                                     void *y, long long n_pix, int C, int G, int flags, void *stream) {
    DS_REQUIRE(z_f16 && scale_t && shift_t && y, DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_RESIDUAL) || res_f16, DS_ERR_NULL);
    DS_REQUIRE(tf_shape_ok(n_pix, C, G), DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(z_f16) && DS_ALIGNED16(y) && DS_ALIGNED16(res_f16) && DS_ALIGNED16(scale_t) && DS_ALIGNED16(shift_t),
               DS_ERR_ALIGNMENT);
    const long long nvm = n_pix * (C / 8);
    if (flags & DS_EPI_OUT_F32)
        DS_LAUNCH(bn_apply_f16_kernel<true>, tf_grid(nvm), 256, 0, stream, (const h16 *)z_f16, scale_t, shift_t,
                  (const h16 *)res_f16, y, nvm, G, C, flags);
    else
        DS_LAUNCH(bn_apply_f16_kernel<false>, tf_grid(nvm), 256, 0, stream, (const h16 *)z_f16, scale_t, shift_t,
                  (const h16 *)res_f16, y, nvm, G, C, flags);
    return ds_last_launch_error();
}

// BatchNorm + clipped-ReLU backward of G members over fp16 tensors (three launches): g1 [+ g2] masked -> gy; sums; gz =
// dL/d(conv output).  The clip mask comes from `act` (0 < act < 20; fp16, or f32 with act_is_f32), or -- act == nullptr and
// mask_scale_t / mask_shift_t given ([G][C]: the forward's scale / shift tables) -- from the layer's own pre-activation z,
// or from nothing (both nullptr: g1 is already masked).  gy == nullptr (allowed with the z-derived mask, no g2, no
// parity layout): the masked gradient is never stored, the second launch recomputes it from g1.  Gradient tensors are in
// loss-scaled units (S * g); ggamma / gbeta [C] leave un-scaled (inv_scale = 1 / S).
// g1_parity: g1 is the [B][ceil(H/2)][ceil(W/2)][4 C] output of the stride-2 data gradient run as one 3x3 convolution
// (H, W = this layer's map; otherwise ignored).  partial: G * ds_bn_f16_partial_rows(n_pix, C) * C * 2 floats; coef: [G][3 C].
static int bwd_reduce_f16(const void *g1, int g1_parity, const void *g2, const void *act, int act_is_f32,
                          const float *mask_scale_t, const float *mask_shift_t, const void *z, const float *mean_t,
                          const float *invstd_t, void *gy, float *partial, long long n_pix, int H, int W, int C, int G,
                          void *stream) {
    DS_REQUIRE(g1 && z && mean_t && invstd_t && partial, DS_ERR_NULL);
    DS_REQUIRE(tf_shape_ok(n_pix, C, G), DS_ERR_BAD_SHAPE);
    DS_REQUIRE(!g1_parity || (H > 0 && W > 0 && (n_pix * G) % ((long long)H * W) == 0 && n_pix * G < (1ll << 24)),
               DS_ERR_BAD_SHAPE);
    DS_REQUIRE((mask_scale_t == nullptr) == (mask_shift_t == nullptr), DS_ERR_NULL);
    DS_REQUIRE(!(act && mask_scale_t), DS_ERR_UNSUPPORTED);
    const bool maskz = mask_scale_t != nullptr;
    DS_REQUIRE(gy || (maskz && !g2 && !g1_parity), DS_ERR_UNSUPPORTED);
    DS_REQUIRE(DS_ALIGNED16(g1) && DS_ALIGNED16(g2) && DS_ALIGNED16(act) && DS_ALIGNED16(z) && DS_ALIGNED16(gy) &&
                   DS_ALIGNED16(mean_t) && DS_ALIGNED16(invstd_t) && DS_ALIGNED16(mask_scale_t) && DS_ALIGNED16(mask_shift_t),
               DS_ERR_ALIGNMENT);
    const int blocks = tf_rows(n_pix, C);
    const int ppb = (int)((n_pix + blocks - 1) / blocks);
    const int slots = 256 / (C / 8);
    const size_t lds = (size_t)slots * C * 2 * 4;
#define TF_REDUCE(M, P, A)                                                                                                 \
    DS_LAUNCH((bn_bwd_reduce_f16_kernel<M, P, A>), blocks * G, 256, lds, stream, (const h16 *)g1, (const h16 *)g2, act,     \
              (const h16 *)z, mean_t, invstd_t, mask_scale_t, mask_shift_t, (h16 *)gy, partial, n_pix, C, ppb, blocks, H, W)
    if (maskz) { if (g1_parity) TF_REDUCE(2, true, false); else TF_REDUCE(2, false, false); }
    else if (!act) { if (g1_parity) TF_REDUCE(0, true, false); else TF_REDUCE(0, false, false); }
    else if (g1_parity) { if (act_is_f32) TF_REDUCE(1, true, true); else TF_REDUCE(1, true, false); }
    else { if (act_is_f32) TF_REDUCE(1, false, true); else TF_REDUCE(1, false, false); }
#undef TF_REDUCE
    return ds_last_launch_error();
}

static int bwd_apply_f16(const void *gy_or_g1, bool regen, const void *z, const float *mean_t, const float *invstd_t,
                         const float *coef, const float *mask_scale_t, const float *mask_shift_t, void *gz, long long n_pix,
                         int C, int G, void *stream) {
    DS_REQUIRE(gy_or_g1 && z && gz && coef, DS_ERR_NULL);
    DS_REQUIRE(DS_ALIGNED16(gy_or_g1) && DS_ALIGNED16(gz) && DS_ALIGNED16(coef), DS_ERR_ALIGNMENT);
    const long long nvm = n_pix * (C / 8);
    if (!regen)
        DS_LAUNCH(bn_bwd_apply_f16_kernel<false>, tf_grid(nvm), 256, 0, stream, (const h16 *)gy_or_g1, (const h16 *)z, mean_t,
                  invstd_t, coef, mask_scale_t, mask_shift_t, (h16 *)gz, nvm, G, C);
    else
        DS_LAUNCH(bn_bwd_apply_f16_kernel<true>, tf_grid(nvm), 256, 0, stream, (const h16 *)gy_or_g1, (const h16 *)z, mean_t,
                  invstd_t, coef, mask_scale_t, mask_shift_t, (h16 *)gz, nvm, G, C);
    return ds_last_launch_error();
}

extern "C" int ds_bn_bwd_group_f16(const void *g1, int g1_parity, const void *g2, const void *act, int act_is_f32,
                                   const float *mask_scale_t, const float *mask_shift_t, const void *z, const float *mean_t,
                                   const float *invstd_t, const float *gamma, void *gy, float *partial, float *coef,
                                   float *ggamma, float *gbeta, void *gz, long long n_pix, int H, int W, int C, int G,
                                   float inv_scale, void *stream) {
    DS_REQUIRE(gamma && coef && ggamma && gbeta && gz, DS_ERR_NULL);
    int rc = bwd_reduce_f16(g1, g1_parity, g2, act, act_is_f32, mask_scale_t, mask_shift_t, z, mean_t, invstd_t, gy, partial,
                            n_pix, H, W, C, G, stream);
    if (rc) return rc;
    DS_LAUNCH(bn_bwd_finalize_f16_kernel, ds_ceil_div(C, TF_FOLD_C), 256, TF_FOLD_R * TF_FOLD_C * 2 * sizeof(double), stream,
              (const float *)partial, tf_rows(n_pix, C), (double)n_pix, gamma, invstd_t, coef, ggamma, gbeta, C, G, inv_scale);
    rc = ds_last_launch_error();
    if (rc) return rc;
    return bwd_apply_f16(gy ? gy : g1, gy == nullptr, z, mean_t, invstd_t, coef, mask_scale_t, mask_shift_t, gz, n_pix, C, G,
                         stream);
}

// Data-parallel split of ds_bn_bwd_group_f16: the reduction alone (partial sums + gy), then -- after the caller has folded
// the partials to float64 (ds_partial_sum_f64_group), and all-reduced the [G][2C+1] sums over RCCL -- coefficients, dgamma /
// dbeta and gz from the GLOBAL sums.  Same arguments, same mask choices.
extern "C" int ds_bn_bwd_group_reduce_f16(const void *g1, int g1_parity, const void *g2, const void *act, int act_is_f32,
                                          const float *mask_scale_t, const float *mask_shift_t, const void *z,
                                          const float *mean_t, const float *invstd_t, void *gy, float *partial,
                                          long long n_pix, int H, int W, int C, int G, void *stream) {
    return bwd_reduce_f16(g1, g1_parity, g2, act, act_is_f32, mask_scale_t, mask_shift_t, z, mean_t, invstd_t, gy, partial, n_pix,
                          H, W, C, G, stream);
}

// gy_or_g1: the stored masked gradient, or (gy was not stored: regen != 0) g1 itself with the z-derived mask tables
extern "C" int ds_bn_bwd_group_apply_f16(const double *sums, const void *gy_or_g1, int regen, const float *mask_scale_t,
                                         const float *mask_shift_t, const void *z, const float *mean_t, const float *invstd_t,
                                         const float *gamma, float *coef, float *ggamma, float *gbeta, void *gz,
                                         long long n_pix, int C, int G, float inv_scale, void *stream) {
    DS_REQUIRE(sums && gamma && coef && ggamma && gbeta && mean_t && invstd_t, DS_ERR_NULL);
    DS_REQUIRE(tf_shape_ok(n_pix, C, G), DS_ERR_BAD_SHAPE);
    DS_REQUIRE(!regen || (mask_scale_t && mask_shift_t), DS_ERR_NULL);
    DS_LAUNCH(bn_bwd_from_sums_f16_kernel, ds_ceil_div(C, 256), 256, 0, stream, sums, gamma, invstd_t, coef, ggamma, gbeta, C, G,
              inv_scale);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    return bwd_apply_f16(gy_or_g1, regen != 0, z, mean_t, invstd_t, coef, mask_scale_t, mask_shift_t, gz, n_pix, C, G, stream);
}

// y_f16 = fp16(x * scale): how an f32 gradient enters the fp16 backward pass (scale = the loss scale S)
extern "C" int ds_scale_cast_f32_to_f16(const float *x, void *y_f16, long long n, float scale, void *stream) {
    DS_REQUIRE(x && y_f16, DS_ERR_NULL);
    DS_REQUIRE(n > 0 && (n % 8) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(x) && DS_ALIGNED16(y_f16), DS_ERR_ALIGNMENT);
    DS_LAUNCH(scale_cast_f16_kernel, tf_grid(n / 8), 256, 0, stream, x, (h16 *)y_f16, n / 8, scale);
    return ds_last_launch_error();
}
