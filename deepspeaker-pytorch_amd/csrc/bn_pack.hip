// bn_pack.hip -- BatchNorm bookkeeping kernels and one-off layout/weight packing.
//   * eval-mode fold and train-mode statistics finalisation of nn.BatchNorm2d
//     (reference model.py:59,62,94,99,103,107; semantics SURVEY 8(a) a2)
//   * elementwise normalise (+residual, +clipped ReLU) for the train-mode path
//   * OIHW -> packed filter layouts, NCHW <-> channels-last conversion
#include <ds_device.h>
#include "ds_common.h"

namespace {

__global__ void __launch_bounds__(256) bn_fold_kernel(const float *gamma, const float *beta, const float *mean,
                                                      const float *var, float eps, float *scale, float *shift,
                                                      int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) {
        const float inv = 1.0f / sqrtf(var[c] + eps);
        const float s = gamma[c] * inv;
        scale[c] = s;
        shift[c] = beta[c] - mean[c] * s;
    }
}

// One workgroup per 32 channels; 8 row-lanes stride over the partial rows in double precision.
__global__ void __launch_bounds__(256) bn_stats_finalize_kernel(const float *partial, int n_partial, double count,
                                                                const float *gamma, const float *beta, float eps,
                                                                float momentum, float *running_mean,
                                                                float *running_var, float *batch_mean,
                                                                float *batch_invstd, float *scale, float *shift,
                                                                int C) {
    double *red = (double *)ds_dynamic_lds();              // [8][32][2]
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        for (int r = rl; r < n_partial; r += 8) {
            const float *src = partial + ((size_t)r * C + c) * 2;
            s1 += (double)src[0];
            s2 += (double)src[1];
        }
    }
    red[(rl * 32 + cl) * 2 + 0] = s1;
    red[(rl * 32 + cl) * 2 + 1] = s2;
    __syncthreads();
    if (rl == 0 && c < C) {
        double t1 = 0.0, t2 = 0.0;
        for (int k = 0; k < 8; ++k) {
            t1 += red[(k * 32 + cl) * 2 + 0];
            t2 += red[(k * 32 + cl) * 2 + 1];
        }
        const double mean = t1 / count;
        double var = t2 / count - mean * mean;             // biased (normalisation) variance
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
        if (running_mean) {
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
        if (batch_mean) batch_mean[c] = (float)mean;
        if (batch_invstd) batch_invstd[c] = (float)invstd;
        const double s = (double)gamma[c] * invstd;
        scale[c] = (float)s;
        shift[c] = (float)((double)beta[c] - mean * s);
    }
}

__global__ void __launch_bounds__(256) bn_apply_kernel(const float *x, const float *scale, const float *shift,
                                                       const float *res, float *y, long long n_vec, int C,
                                                       int flags) {
    const int cvec = C >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (long long)gridDim.x * 256) {
        const int c4 = (int)(i % cvec);
        f32x4 v = ((const f32x4 *)x)[i];
        const f32x4 sc = ((const f32x4 *)scale)[c4], sh = ((const f32x4 *)shift)[c4];
        v = v * sc + sh;
        if (flags & DS_EPI_RESIDUAL) v += ((const f32x4 *)res)[i];
        if (flags & DS_EPI_CLIP) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fminf(fmaxf(v[j], 0.0f), 20.0f);
        }
        ((f32x4 *)y)[i] = v;
    }
}

// OIHW -> [Cin/8][KS*KS][Cout][8]; dgrad: roles of Cout/Cin swapped, taps flipped
__global__ void __launch_bounds__(256) pack_conv_weight_kernel(const float *w, float *out, int Cout, int Cin, int KS,
                                                               int dgrad) {
    const int T = KS * KS;
    const long long n = (long long)Cout * Cin * T;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        // i indexes the packed tensor [K/8][T][N][8] where (N,K) = (Cout,Cin) or (Cin,Cout) for dgrad
        const int N = dgrad ? Cin : Cout, K = dgrad ? Cout : Cin;
        const int kk = (int)(i & 7);
        long long r = i >> 3;
        const int nn = (int)(r % N);
        r /= N;
        const int t = (int)(r % T);
        const int kc = (int)(r / T);
        const int k = kc * 8 + kk;
        const int tt = dgrad ? (T - 1 - t) : t;
        const int kh = tt / KS, kw = tt - kh * KS;
        const int co = dgrad ? k : nn, ci = dgrad ? nn : k;
        (void)K;
        out[i] = w[(((size_t)co * Cin + ci) * KS + kh) * KS + kw];
    }
}

__global__ void __launch_bounds__(256) pack_conv1_weight_kernel(const float *w, float *out, int Cout) {
    const int i = blockIdx.x * 256 + threadIdx.x;          // out[t][co] = w[co][0][t]
    if (i < 25 * Cout) {
        const int t = i / Cout, co = i - t * Cout;
        out[i] = w[co * 25 + t];
    }
}

// fc weight [N][C*F] (c*F+f) -> [K'/8][1][N][8] with k' = f*C + c
__global__ void __launch_bounds__(256) pack_fc_weight_kernel(const float *w, float *out, int N, int C, int F) {
    const long long n = (long long)N * C * F;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int kk = (int)(i & 7);
        long long r = i >> 3;
        const int nn = (int)(r % N);
        const int kc = (int)(r / N);
        const int kp = kc * 8 + kk;
        const int f = kp / C, c = kp - f * C;
        out[i] = w[(size_t)nn * C * F + (size_t)c * F + f];
    }
}

__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float *x, float *y, int B, int C, int HW,
                                                           int to_nhwc) {
    const long long n = (long long)B * C * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        // i indexes the DESTINATION so writes are coalesced
        if (to_nhwc) {
            const int c = (int)(i % C);
            const long long r = i / C;
            const int s = (int)(r % HW);
            const int b = (int)(r / HW);
            y[i] = x[((size_t)b * C + c) * HW + s];
        } else {
            const int s = (int)(i % HW);
            const long long r = i / HW;
            const int c = (int)(r % C);
            const int b = (int)(r / C);
            y[i] = x[((size_t)b * HW + s) * C + c];
        }
    }
}

static int grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g > 4096 ? 4096 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int ds_bn_fold_f32(const float *gamma, const float *beta, const float *running_mean,
                              const float *running_var, float eps, float *scale, float *shift, int C, void *stream) {
    DS_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, DS_ERR_NULL);
    DS_REQUIRE(C > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(bn_fold_kernel, ds_ceil_div(C, 256), 256, 0, stream, gamma, beta, running_mean, running_var, eps, scale,
              shift, C);
    return ds_last_launch_error();
}

extern "C" int ds_bn_stats_finalize_f32(const float *partial, int n_partial, long long count, const float *gamma,
                                        const float *beta, float eps, float momentum, float *running_mean,
                                        float *running_var, float *batch_mean, float *batch_invstd, float *scale,
                                        float *shift, int C, void *stream) {
    DS_REQUIRE(partial && gamma && beta && scale && shift, DS_ERR_NULL);
    DS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), DS_ERR_NULL);
    DS_REQUIRE(C > 0 && n_partial > 0 && count > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(bn_stats_finalize_kernel, ds_ceil_div(C, 32), 256, 8 * 32 * 2 * sizeof(double), stream, partial,
              n_partial, (double)count, gamma, beta, eps, momentum, running_mean, running_var, batch_mean,
              batch_invstd, scale, shift, C);
    return ds_last_launch_error();
}

extern "C" int ds_bn_apply_f32(const float *x, const float *scale, const float *shift, const float *residual,
                               float *y, long long n_pix, int C, int flags, void *stream) {
    DS_REQUIRE(x && scale && shift && y, DS_ERR_NULL);
    DS_REQUIRE(!(flags & DS_EPI_RESIDUAL) || residual, DS_ERR_NULL);
    DS_REQUIRE(n_pix > 0 && C > 0 && (C % 4) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(x) && DS_ALIGNED16(y) && DS_ALIGNED16(scale) && DS_ALIGNED16(shift), DS_ERR_ALIGNMENT);
    const long long n_vec = n_pix * (C / 4);
    int grid = grid_for(n_vec);
    DS_LAUNCH(bn_apply_kernel, grid, 256, 0, stream, x, scale, shift, residual, y, n_vec, C, flags);
    return ds_last_launch_error();
}

extern "C" int ds_pack_conv_weight_f32(const float *w_oihw, float *w_packed, int Cout, int Cin, int KS, int dgrad,
                                       void *stream) {
    DS_REQUIRE(w_oihw && w_packed, DS_ERR_NULL);
    DS_REQUIRE(Cout > 0 && Cin > 0 && (KS == 1 || KS == 3 || KS == 5), DS_ERR_BAD_SHAPE);
    DS_REQUIRE(((dgrad ? Cout : Cin) % 8) == 0, DS_ERR_BAD_SHAPE);
    const long long n = (long long)Cout * Cin * KS * KS;
    DS_LAUNCH(pack_conv_weight_kernel, grid_for(n), 256, 0, stream, w_oihw, w_packed, Cout, Cin, KS, dgrad);
    return ds_last_launch_error();
}

extern "C" int ds_pack_conv1_weight_f32(const float *w_oihw, float *w_packed, int Cout, void *stream) {
    DS_REQUIRE(w_oihw && w_packed, DS_ERR_NULL);
    DS_REQUIRE(Cout > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(pack_conv1_weight_kernel, ds_ceil_div(25 * Cout, 256), 256, 0, stream, w_oihw, w_packed, Cout);
    return ds_last_launch_error();
}

extern "C" int ds_pack_fc_weight_f32(const float *w, float *w_packed, int N, int C, int F, void *stream) {
    DS_REQUIRE(w && w_packed, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && C > 0 && F > 0 && ((C * F) % 8) == 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(pack_fc_weight_kernel, grid_for((long long)N * C * F), 256, 0, stream, w, w_packed, N, C, F);
    return ds_last_launch_error();
}

extern "C" int ds_nchw_to_nhwc_f32(const float *x, float *y, int B, int C, int H, int W, void *stream) {
    DS_REQUIRE(x && y, DS_ERR_NULL);
    DS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(nchw_to_nhwc_kernel, grid_for((long long)B * C * H * W), 256, 0, stream, x, y, B, C, H * W, 1);
    return ds_last_launch_error();
}

extern "C" int ds_nhwc_to_nchw_f32(const float *x, float *y, int B, int C, int H, int W, void *stream) {
    DS_REQUIRE(x && y, DS_ERR_NULL);
    DS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(nchw_to_nhwc_kernel, grid_for((long long)B * C * H * W), 256, 0, stream, x, y, B, C, H * W, 0);
    return ds_last_launch_error();
}

extern "C" int ds_version(void) { return 100; }

extern "C" const char *ds_error_string(int code) {
    switch (code) {
        case DS_OK: return "ok";
        case DS_ERR_BAD_SHAPE: return "bad shape";
        case DS_ERR_ALIGNMENT: return "pointer not 16-byte aligned";
        case DS_ERR_NULL: return "null pointer";
        case DS_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return code > 0 ? "HIP runtime error (hipError_t)" : "unknown error";
    }
}
