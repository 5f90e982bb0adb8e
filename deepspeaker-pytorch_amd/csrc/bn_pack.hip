// bn_pack.hip -- BatchNorm bookkeeping kernels and one-off layout/weight packing.
//   * eval-mode fold and train-mode statistics finalisation of nn.BatchNorm2d
//     (reference model.py:59,62,94,99,103,107; semantics SURVEY 8(a) a2)
//   * elementwise normalise (+residual, +clipped ReLU) for the train-mode path
//   * OIHW -> packed filter layouts, NCHW <-> channels-last conversion
#include <ds_device.h>
#include <unistd.h>
#include "ds_common.h"
#include <map>
#include <mutex>
#include <utility>
#include <vector>
#include <stdlib.h>

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

// Fold of [n_partial][C][2] partial sums in double precision, fixed order: one workgroup per FOLD_C channels,
// FOLD_R row-lanes stride over the partial rows (stage-1 layers have thousands of rows and only 64 channels --
// a channel-per-thread layout left the chip idle; 8 channels x 32 lanes still meant 8 workgroups walking 64 rows
// each: 27 us, now 8), then lane 0 folds the lane sums.
// Round 6: the lane sums are folded by an xor tree of shuffles inside each wave (lane = 2 * row lane + channel: offsets
// 2 .. 32) and the four waves' results by one thread -- the last step used to be ONE thread adding 128 LDS values per
// channel in sequence (14 - 18 us per launch, 48 launches per training step).
constexpr int FOLD_C = 2, FOLD_R = 128;
__device__ __forceinline__ bool fold_partials(const float *partial, int n_partial, int C, double *red, int &c,
                                              double &t1, double &t2, int cgroup = -1) {
    const int cl = threadIdx.x % FOLD_C, rl = threadIdx.x / FOLD_C;
    c = (cgroup < 0 ? (int)blockIdx.x : cgroup) * FOLD_C + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        for (int r = rl; r < n_partial; r += FOLD_R) {
            const float *src = partial + ((size_t)r * C + c) * 2;
            s1 += (double)src[0];
            s2 += (double)src[1];
        }
    }
#pragma unroll
    for (int m = 2; m <= 32; m <<= 1) {
        s1 += ds_shfl_xor_f64(s1, m);
        s2 += ds_shfl_xor_f64(s2, m);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < FOLD_C) {
        red[(wave * FOLD_C + cl) * 2 + 0] = s1;
        red[(wave * FOLD_C + cl) * 2 + 1] = s2;
    }
    __syncthreads();
    if (rl != 0 || c >= C) return false;
    t1 = 0.0;
    t2 = 0.0;
    for (int k = 0; k < 4; ++k) {
        t1 += red[(k * FOLD_C + cl) * 2 + 0];
        t2 += red[(k * FOLD_C + cl) * 2 + 1];
    }
    return true;
}

__global__ void __launch_bounds__(256) bn_stats_finalize_kernel(const float *partial, int n_partial, double count,
                                                                const float *gamma, const float *beta, float eps,
                                                                float momentum, float *running_mean,
                                                                float *running_var, float *batch_mean,
                                                                float *batch_invstd, float *scale, float *shift,
                                                                int C) {
    double *red = (double *)ds_dynamic_lds();              // [FOLD_R][FOLD_C][2]
    int c;
    double t1, t2;
    if (fold_partials(partial, n_partial, C, red, c, t1, t2)) {
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
    // 256 threads, grid stride a multiple of 256: when C / 4 is a power of two dividing 256 (every layer of the network) a
    // thread keeps ONE channel group for the whole pass -- table rows loaded once, no 64-bit modulo per vector (round 4:
    // the per-vector `i % cvec` in 64-bit arithmetic was a third of the pass's instructions)
    const bool fixed = (256 % cvec) == 0 && (cvec & (cvec - 1)) == 0;
    const int c4f = threadIdx.x & (cvec - 1);
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = sc;
    if (fixed) {
        sc = ((const f32x4 *)scale)[c4f];
        sh = ((const f32x4 *)shift)[c4f];
    }
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (long long)gridDim.x * 256) {
        if (!fixed) {
            const int c4 = (int)(i % cvec);
            sc = ((const f32x4 *)scale)[c4];
            sh = ((const f32x4 *)shift)[c4];
        }
        f32x4 v = ((const f32x4 *)x)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ds_bn_affine(v[j], sc[j], sh[j]);
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

// 5x5 stride-2 data-gradient packing: four parity classes (ph,pw) in order (0,0),(0,1),(1,0),(1,1),
// class taps = {kh = ph mod 2} x {kw = pw mod 2} in ascending kernel index; each class block is
// [Cout/8][taps][Cin][8] (K = Cout, N = Cin) -- see ds_conv_dgrad_f32.
__global__ void __launch_bounds__(256) pack_conv_dgrad_s2_kernel(const float *w, float *out, int Cout, int Cin) {
    const long long n = (long long)Cout * Cin * 25;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        long long r = i;
        int ph = 0, pw = 0, nh = 3, nw = 3;
        for (int c = 0; c < 4; ++c) {
            ph = c >> 1; pw = c & 1;
            nh = ph ? 2 : 3; nw = pw ? 2 : 3;
            const long long sz = (long long)nh * nw * Cout * Cin;
            if (r < sz) break;
            r -= sz;
        }
        const int kk = (int)(r & 7);
        r >>= 3;
        const int ci = (int)(r % Cin);
        r /= Cin;
        const int t = (int)(r % (nh * nw));
        const int kc = (int)(r / (nh * nw));
        const int co = kc * 8 + kk;
        const int kh = ph + 2 * (t / nw), kw = pw + 2 * (t % nw);
        out[i] = w[(((size_t)co * Cin + ci) * 5 + kh) * 5 + kw];
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

// fc data-gradient packing: gpooled[b][k'] = sum_n gf[b][n] * W[n][c*F+f]  ->  [N/8][1][K'][8]
__global__ void __launch_bounds__(256) pack_fc_weight_dgrad_kernel(const float *w, float *out, int N, int C, int F) {
    const long long n = (long long)N * C * F;
    const int K = C * F;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int kk = (int)(i & 7);
        long long r = i >> 3;
        const int kp = (int)(r % K);
        const int nc = (int)(r / K);
        const int nn = nc * 8 + kk;
        const int f = kp / C, c = kp - f * C;
        out[i] = w[(size_t)nn * K + (size_t)c * F + f];
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
    DS_LAUNCH(bn_stats_finalize_kernel, ds_ceil_div(C, FOLD_C), 256, FOLD_R * FOLD_C * 2 * sizeof(double), stream, partial,
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

extern "C" int ds_pack_conv_dgrad_s2_f32(const float *w_oihw, float *w_packed, int Cout, int Cin, void *stream) {
    DS_REQUIRE(w_oihw && w_packed, DS_ERR_NULL);
    DS_REQUIRE(Cout > 0 && Cin > 0 && (Cout % 8) == 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(pack_conv_dgrad_s2_kernel, grid_for((long long)Cout * Cin * 25), 256, 0, stream, w_oihw, w_packed, Cout,
              Cin);
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

extern "C" int ds_pack_fc_weight_dgrad_f32(const float *w, float *w_packed, int N, int C, int F, void *stream) {
    DS_REQUIRE(w && w_packed, DS_ERR_NULL);
    DS_REQUIRE(N > 0 && C > 0 && F > 0 && (N % 8) == 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(pack_fc_weight_dgrad_kernel, grid_for((long long)N * C * F), 256, 0, stream, w, w_packed, N, C, F);
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

// Scheduling slots of the persistent kernels (ds_device.h): tile counters that must be private to whatever can be in
// flight at the same time.
//   * Eager launches take the next of DS_SCHED_RING slots of their own (device, stream): launches of one stream run in
//     order, so the only launches that can ever share a slot are ones the stream itself serialises -- whatever other
//     streams, graphs or processes' worth of persistent launches are enqueued in between (a process-wide round-robin,
//     as before round 4, handed the slot of a kernel still queued on stream A to the 65th launch enqueued on stream B).
//   * A launch that is being CAPTURED into a graph keeps a slot of its own for good: the node carries the pointer and
//     can replay on any stream next to anything.
// THE MEMORY IS THE CALLER'S (round 6, SURVEY 8(b): "the library never hipMallocs ... no synchronise"): slots are carved
// from zeroed device buffers the caller hands over with ds_sched_set_workspace (the Python wrapper: one torch.zeros of
// ds_sched_workspace_bytes() per device, allocated when a model is moved to the device or on the first launch there);
// a kernel leaves its slot zeroed.  Without a workspace -- or with every slot of it taken by captured launches -- a
// persistent launch returns DS_ERR_NO_WORKSPACE and the caller hands over another buffer.  Host side is serialised by a
// mutex; the only state the library keeps is the table of what has been carved.
namespace {
struct SchedPool {
    std::mutex mu;
    struct Chunk { unsigned *base; size_t slots, used; };
    std::map<int, std::vector<Chunk>> chunks;                               // device -> the caller's buffers
    std::map<std::pair<int, void *>, std::pair<unsigned *, unsigned>> rings;  // (device, stream) -> (ring base, next)

    unsigned *carve(int dev, size_t n_slots) {
        for (auto &c : chunks[dev])
            if (c.used + n_slots <= c.slots) {
                unsigned *r = c.base + c.used * DS_SCHED_WORDS;
                c.used += n_slots;
                return r;
            }
        return nullptr;
    }
    size_t free_slots(int dev) {
        size_t n = 0;
        for (auto &c : chunks[dev]) n += c.slots - c.used;
        return n;
    }
};
SchedPool &sched_pool() { static SchedPool p; return p; }
}  // namespace

unsigned *ds_sched_slot(void *stream) {
    SchedPool &P = sched_pool();
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    bool capturing = false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess) capturing = cs == hipStreamCaptureStatusActive;
    else (void)hipGetLastError();
    std::lock_guard<std::mutex> lock(P.mu);
    if (capturing) return P.carve(dev, 1);
    auto key = std::make_pair(dev, stream);
    auto it = P.rings.find(key);
    if (it == P.rings.end()) {
        unsigned *base = P.carve(dev, DS_SCHED_RING);
        if (!base) return nullptr;
        it = P.rings.emplace(key, std::make_pair(base, 0u)).first;
    }
    const unsigned k = it->second.second++ % DS_SCHED_RING;
    return it->second.first + (size_t)k * DS_SCHED_WORDS;
}

// bytes of one scheduler workspace: 1024 slots of 64 bytes (a ring of 8 per stream that launches persistent kernels,
// one per persistent launch captured into a graph)
extern "C" size_t ds_sched_workspace_bytes(void) { return (size_t)1024 * DS_SCHED_WORDS * sizeof(unsigned); }

// Hands `bytes` of ZEROED device memory on the CURRENT device to the persistent kernels' tile scheduler.  The buffer
// must stay allocated for as long as the library may launch (the wrapper keeps the tensor alive for the life of the
// process); it may be called again to add a buffer when DS_ERR_NO_WORKSPACE says the previous ones are used up.
extern "C" int ds_sched_set_workspace(void *zeroed_device_memory, size_t bytes) {
    DS_REQUIRE(zeroed_device_memory != nullptr, DS_ERR_NULL);
    DS_REQUIRE(DS_ALIGNED16(zeroed_device_memory), DS_ERR_ALIGNMENT);
    const size_t slots = bytes / (DS_SCHED_WORDS * sizeof(unsigned));
    DS_REQUIRE(slots >= 2 * DS_SCHED_RING, DS_ERR_BAD_SHAPE);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return DS_ERR_UNSUPPORTED; }
    SchedPool &P = sched_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    P.chunks[dev].push_back({(unsigned *)zeroed_device_memory, slots, 0});
    return DS_OK;
}

// slots of the current device's workspaces that have not been handed out yet (0: ds_sched_set_workspace is due)
extern "C" long long ds_sched_free_slots(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return -1; }
    SchedPool &P = sched_pool();
    std::lock_guard<std::mutex> lock(P.mu);
    return (long long)P.free_slots(dev);
}

extern "C" int ds_version(void) { return 600; }   // 600: round 6 (caller-owned scheduler workspace, ds_mfma_rate_probe_data); 500: round 5; 400: round 4 (fp16 training step, refinement probes, launch-bound timing); 30x: round-3 ABI (300: split grouped BatchNorm backward for data parallelism, grouped f64 sums; 301: + ds_conv_dgrad_bnbwd_bf16, ds_bn_bwd_group_finish_f32)

// ---- launch timing (see DS_LAUNCH_BIG_LDS in ds_device.h) ----
extern "C" int ds_event_create(void **out_event) {
    DS_REQUIRE(out_event != nullptr, DS_ERR_NULL);
    hipEvent_t e = nullptr;
    const hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return (int)rc;
    *out_event = (void *)e;
    return DS_OK;
}

extern "C" int ds_event_destroy(void *event) {
    DS_REQUIRE(event != nullptr, DS_ERR_NULL);
    return (int)hipEventDestroy((hipEvent_t)event);
}

// milliseconds between two events (waits for `stop` first, at most 2 s)
extern "C" int ds_event_elapsed_ms(void *start, void *stop, float *ms) {
    DS_REQUIRE(start && stop && ms, DS_ERR_NULL);
    // bounded wait (2 s): an event that was armed but never bound to a launch must not hang the caller
    hipError_t rc = hipEventQuery((hipEvent_t)stop);
    for (int i = 0; rc == hipErrorNotReady && i < 20000; ++i) {
        usleep(100);
        rc = hipEventQuery((hipEvent_t)stop);
    }
    if (rc != hipSuccess) {
        (void)hipGetLastError();
        return (int)rc;
    }
    rc = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    if (rc != hipSuccess) (void)hipGetLastError();
    return (int)rc;
}

// The NEXT big-LDS kernel launch of this thread (the MFMA convolution / filter-gradient kernels) records its own
// execution into (start, stop).  ds_launch_timing_end() disarms and returns the number of such launches since arming
// (the caller expects 1: a call that launched several kernels timed only its first).
extern "C" int ds_launch_timing_arm(void *start, void *stop) {
    DS_REQUIRE(start && stop, DS_ERR_NULL);
    ds_timing_arm_state = {(hipEvent_t)start, (hipEvent_t)stop, 1, 0};
    return DS_OK;
}

extern "C" int ds_launch_timing_end(void) {
    const int n = ds_timing_arm_state.launches;
    ds_timing_arm_state = {nullptr, nullptr, 0, 0};
    return n;
}

extern "C" const char *ds_error_string(int code) {
    switch (code) {
        case DS_OK: return "ok";
        case DS_ERR_BAD_SHAPE: return "bad shape";
        case DS_ERR_ALIGNMENT: return "pointer not 16-byte aligned";
        case DS_ERR_NULL: return "null pointer";
        case DS_ERR_UNSUPPORTED: return "unsupported configuration";
        case DS_ERR_NO_WORKSPACE: return "no free tile-scheduling slot on this device: hand over zeroed device memory with ds_sched_set_workspace";
        default: return code > 0 ? "HIP runtime error (hipError_t)" : "unknown error";
    }
}

// =================================================================================================
// backward kernels of BatchNorm (train mode) -- autograd of reference model.py:70,74,188,... as
// executed by loss.backward() (train_triplet.py:223,290); formulas: SURVEY 8(a) a13
// =================================================================================================
namespace {

// gy = (g1 [+ g2]) * [0 < act < 20]   (the clipped-ReLU mask; act == nullptr: no mask)
// partial[blk][c] = { sum gy, sum gy * xhat },  xhat = (z - mean) * invstd.   gy is also written out.
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(const float *g1, const float *g2, const float *act,
                                                            const float *z, const float *mean, const float *invstd,
                                                            float *gy, float *partial, long long n_pix, int C,
                                                            int pix_per_block, int blocks_per_member) {
    // a batch of G members with their own statistics (the three forwards of a triplet step run as one batch): member
    // m = blockIdx.x / blocks_per_member owns pixels [m * n_pix, (m + 1) * n_pix), row m of mean / invstd and
    // blocks_per_member partial rows
    const int member = blockIdx.x / blocks_per_member, mblock = blockIdx.x - member * blocks_per_member;
    {
        const size_t off = (size_t)member * n_pix * C;
        g1 += off;
        if (g2) g2 += off;
        if (act) act += off;
        z += off;
        gy += off;
        mean += (size_t)member * C;
        invstd += (size_t)member * C;
    }
    float *red = ds_dynamic_lds();                         // [slots][C][2]
    const int cvec = C >> 2;
    const int slots = 256 / cvec;
    const int cg = threadIdx.x % cvec, slot = threadIdx.x / cvec;
    const long long p0 = (long long)mblock * pix_per_block;
    long long p1 = p0 + pix_per_block;
    if (p1 > n_pix) p1 = n_pix;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
    if (slot < slots) {
        const f32x4 mu = ((const f32x4 *)mean)[cg], is = ((const f32x4 *)invstd)[cg];
        for (long long p = p0 + slot; p < p1; p += slots) {
            const size_t i = (size_t)p * cvec + cg;
            f32x4 g = ((const f32x4 *)g1)[i];
            if (g2) g += ((const f32x4 *)g2)[i];
            if (act) {
                const f32x4 a = ((const f32x4 *)act)[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) g[j] = (a[j] > 0.0f && a[j] < 20.0f) ? g[j] : 0.0f;
            }
            ((f32x4 *)gy)[i] = g;
            const f32x4 xh = (((const f32x4 *)z)[i] - mu) * is;
            s1 += g;
            s2 += g * xh;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[((slot * C) + cg * 4 + j) * 2 + 0] = s1[j];
            red[((slot * C) + cg * 4 + j) * 2 + 1] = s2[j];
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

// fold the partials (double precision, fixed order): ggamma = sum gy*xhat, gbeta = sum gy,
// coef = { gamma*invstd, sum gy / N, sum gy*xhat / N }
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const float *partial, int n_partial, double count,
                                                              const float *gamma, const float *invstd,
                                                              float *ggamma, float *gbeta, float *coef, int C) {
    double *red = (double *)ds_dynamic_lds();              // [FOLD_R][FOLD_C][2]
    int c;
    double t1, t2;
    if (fold_partials(partial, n_partial, C, red, c, t1, t2)) {
        gbeta[c] = (float)t1;
        ggamma[c] = (float)t2;
        coef[c] = gamma[c] * invstd[c];
        coef[C + c] = (float)(t1 / count);
        coef[2 * C + c] = (float)(t2 / count);
    }
}

// gz = gamma*invstd * (gy - mean(gy) - xhat * mean(gy*xhat))
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float *gy, const float *z, const float *mean,
                                                           const float *invstd, const float *coef, float *gz,
                                                           long long n_vec, int C) {
    const int cvec = C >> 2;                    // a power of two dividing 256 (checked by the host): fixed channel group
    const int c4 = threadIdx.x & (cvec - 1);
    const f32x4 mu = ((const f32x4 *)mean)[c4], is = ((const f32x4 *)invstd)[c4];
    const f32x4 k1 = ((const f32x4 *)coef)[c4], k2 = ((const f32x4 *)(coef + C))[c4],
                k3 = ((const f32x4 *)(coef + 2 * C))[c4];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (long long)gridDim.x * 256) {
        const f32x4 xh = (((const f32x4 *)z)[i] - mu) * is;
        ((f32x4 *)gz)[i] = k1 * (((const f32x4 *)gy)[i] - k2 - xh * k3);
    }
}

// The same for a batch of G members in one launch: workgroup (channel group, member); per member its own partial rows,
// invstd row, coefficient block and dgamma / dbeta rows (summed over the members by bn_member_sum_kernel)
__global__ void __launch_bounds__(256) bn_bwd_finalize_group_kernel(const float *partial, int n_partial, double count,
                                                                    const float *gamma, const float *invstd,
                                                                    float *ggamma_m, float *gbeta_m, float *coef, int C,
                                                                    int n_cgroups) {
    double *red = (double *)ds_dynamic_lds();              // [FOLD_R][FOLD_C][2]
    const int member = blockIdx.x / n_cgroups, cgroup = blockIdx.x - member * n_cgroups;
    partial += (size_t)member * n_partial * C * 2;
    invstd += (size_t)member * C;
    coef += (size_t)member * 3 * C;
    int c;
    double t1, t2;
    if (fold_partials(partial, n_partial, C, red, c, t1, t2, cgroup)) {
        gbeta_m[(size_t)member * C + c] = (float)t1;
        ggamma_m[(size_t)member * C + c] = (float)t2;
        coef[c] = gamma[c] * invstd[c];
        coef[C + c] = (float)(t1 / count);
        coef[2 * C + c] = (float)(t2 / count);
    }
}

// dgamma / dbeta of the layer = the members' contributions added in member order (what accumulating the reference's
// three backward passes into .grad does)
__global__ void __launch_bounds__(256) bn_member_sum_kernel(const float *ggamma_m, const float *gbeta_m, float *ggamma,
                                                            float *gbeta, int G, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < C) {
        float a = 0.f, b = 0.f;
        for (int m = 0; m < G; ++m) {
            a += ggamma_m[(size_t)m * C + c];
            b += gbeta_m[(size_t)m * C + c];
        }
        ggamma[c] = a;
        gbeta[c] = b;
    }
}

__global__ void __launch_bounds__(256) bn_bwd_apply_group_kernel(const float *gy, const float *z, const float *mean,
                                                                 const float *invstd, const float *coef, float *gz,
                                                                 long long n_vec_member, int G, int C) {
    const int cvec = C >> 2;                    // a power of two dividing 256 (checked by the host): fixed channel group
    const int c4 = threadIdx.x & (cvec - 1);
    for (int member = 0; member < G; ++member) {            // (no per-vector division: members are walked one by one)
        const float *mu_p = mean + (size_t)member * C, *is_p = invstd + (size_t)member * C, *cf = coef + (size_t)member * 3 * C;
        const f32x4 mu = ((const f32x4 *)mu_p)[c4], is = ((const f32x4 *)is_p)[c4];
        const f32x4 k1 = ((const f32x4 *)cf)[c4], k2 = ((const f32x4 *)(cf + C))[c4], k3 = ((const f32x4 *)(cf + 2 * C))[c4];
        const size_t mbase = (size_t)member * (size_t)n_vec_member;
        for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < n_vec_member; v += (long long)gridDim.x * 256) {
            const size_t i = mbase + (size_t)v;
            const f32x4 xh = (((const f32x4 *)z)[i] - mu) * is;
            ((f32x4 *)gz)[i] = k1 * (((const f32x4 *)gy)[i] - k2 - xh * k3);
        }
    }
}

// out[c] = sum_r x[r][c]   (bias gradient of the fc layer).  A workgroup owns 32 columns; its 8 row lanes stride over
// the rows and are folded in lane order (fixed order => deterministic).  (One thread per column walking all rows left
// two workgroups busy for 170 us at the head of every backward pass.)
__global__ void __launch_bounds__(256) colsum_kernel(const float *x, float *out, int R, int C) {
    float *red = ds_dynamic_lds();                         // [8][32]
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < C)
        for (int r = rl; r < R; r += 8) s += x[(size_t)r * C + c];
    red[rl * 32 + cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        float t = 0.f;
        for (int k = 0; k < 8; ++k) t += red[k * 32 + cl];
        out[c] = t;
    }
}

}  // namespace

// ---- split forms for data-parallel training: local sums -> (all-reduce by the caller) -> finalize ----
namespace {

// sums[c][2] (double) = sum over the partial rows, folded like the single-process finalize kernels fold them
// (fold_partials: FOLD_C channels x FOLD_R row lanes per workgroup).  One workgroup per 32 channels with 8 row lanes
// left 6..48 workgroups walking up to 2048 rows each: 66 us per BatchNorm layer of the data-parallel step.
__global__ void __launch_bounds__(256) partial_sum_f64_kernel(const float *partial, int n_partial, double *sums, int C) {
    double *red = (double *)ds_dynamic_lds();              // [FOLD_R][FOLD_C][2]
    int c;
    double t1, t2;
    if (fold_partials(partial, n_partial, C, red, c, t1, t2)) {
        sums[c * 2 + 0] = t1;
        sums[c * 2 + 1] = t2;
    }
}

__global__ void __launch_bounds__(256) bn_stats_from_sums_kernel(const double *sums, double count, const float *gamma,
                                                                 const float *beta, float eps, float momentum,
                                                                 float *running_mean, float *running_var,
                                                                 float *batch_mean, float *batch_invstd, float *scale,
                                                                 float *shift, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (count <= 0.0) count = sums[2 * C];               // pixel count travelled with the all-reduce
    const double mean = sums[c * 2] / count;
    double var = sums[c * 2 + 1] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    if (running_mean) {
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
    if (batch_mean) batch_mean[c] = (float)mean;
    if (batch_invstd) batch_invstd[c] = (float)invstd;
    const double sc = (double)gamma[c] * invstd;
    scale[c] = (float)sc;
    shift[c] = (float)((double)beta[c] - mean * sc);
}

__global__ void __launch_bounds__(256) bn_bwd_from_sums_kernel(const double *sums, double count, const float *gamma,
                                                               const float *invstd, float *ggamma, float *gbeta,
                                                               float *coef, int C) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    if (count <= 0.0) count = sums[2 * C];
    gbeta[c] = (float)sums[c * 2];
    ggamma[c] = (float)sums[c * 2 + 1];
    coef[c] = gamma[c] * invstd[c];
    coef[C + c] = (float)(sums[c * 2] / count);
    coef[2 * C + c] = (float)(sums[c * 2 + 1] / count);
}

// The grouped forms (a batch of G members with their own statistics, data-parallel): workgroup = (member, channel
// group).  sums is [G][2C+1] doubles: per member C pairs, then the member's pixel count (which travels with the
// all-reduce).
__global__ void __launch_bounds__(256) partial_sum_f64_group_kernel(const float *partial, int n_partial, double *sums,
                                                                    double count, int C, int n_cgroups) {
    double *red = (double *)ds_dynamic_lds();              // [FOLD_R][FOLD_C][2]
    const int member = blockIdx.x / n_cgroups, cgroup = blockIdx.x - member * n_cgroups;
    partial += (size_t)member * n_partial * C * 2;
    sums += (size_t)member * (2 * C + 1);
    if (cgroup == 0 && threadIdx.x == 0) sums[2 * C] = count;
    int c;
    double t1, t2;
    if (fold_partials(partial, n_partial, C, red, c, t1, t2, cgroup)) {
        sums[c * 2 + 0] = t1;
        sums[c * 2 + 1] = t2;
    }
}

__global__ void __launch_bounds__(256) bn_bwd_from_sums_group_kernel(const double *sums, const float *gamma,
                                                                     const float *invstd, float *ggamma_m,
                                                                     float *gbeta_m, float *coef, int C,
                                                                     int n_cgroups) {
    const int member = blockIdx.x / n_cgroups, cgroup = blockIdx.x - member * n_cgroups;
    const int c = cgroup * 256 + threadIdx.x;
    if (c >= C) return;
    sums += (size_t)member * (2 * C + 1);
    invstd += (size_t)member * C;
    coef += (size_t)member * 3 * C;
    const double count = sums[2 * C];
    gbeta_m[(size_t)member * C + c] = (float)sums[c * 2];
    ggamma_m[(size_t)member * C + c] = (float)sums[c * 2 + 1];
    coef[c] = gamma[c] * invstd[c];
    coef[C + c] = (float)(sums[c * 2] / count);
    coef[2 * C + c] = (float)(sums[c * 2 + 1] / count);
}

}  // namespace

extern "C" int ds_partial_sum_f64(const float *partial, int n_partial, double *sums, int C, void *stream) {
    DS_REQUIRE(partial && sums, DS_ERR_NULL);
    DS_REQUIRE(n_partial > 0 && C > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(partial_sum_f64_kernel, ds_ceil_div(C, FOLD_C), 256, FOLD_R * FOLD_C * 2 * sizeof(double), stream, partial, n_partial,
              sums, C);
    return ds_last_launch_error();
}

extern "C" int ds_bn_stats_from_sums_f32(const double *sums, long long count, const float *gamma, const float *beta,
                                         float eps, float momentum, float *running_mean, float *running_var,
                                         float *batch_mean, float *batch_invstd, float *scale, float *shift, int C,
                                         void *stream) {
    DS_REQUIRE(sums && gamma && beta && scale && shift, DS_ERR_NULL);
    DS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), DS_ERR_NULL);
    DS_REQUIRE(C > 0 && count >= 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(bn_stats_from_sums_kernel, ds_ceil_div(C, 256), 256, 0, stream, sums, (double)count, gamma, beta, eps,
              momentum, running_mean, running_var, batch_mean, batch_invstd, scale, shift, C);
    return ds_last_launch_error();
}

extern "C" int ds_bn_bwd_reduce_f32(const float *g1, const float *g2, const float *act, const float *z,
                                    const float *mean, const float *invstd, float *gy, float *partial,
                                    long long n_pix, int C, void *stream) {
    DS_REQUIRE(g1 && z && mean && invstd && gy && partial, DS_ERR_NULL);
    DS_REQUIRE(n_pix > 0 && C >= 4 && (C % 4) == 0 && C <= 1024 && 256 % (C / 4) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(g1) && DS_ALIGNED16(z) && DS_ALIGNED16(gy) && DS_ALIGNED16(mean) && DS_ALIGNED16(invstd),
               DS_ERR_ALIGNMENT);
    const int blocks = ds_bn_bwd_partial_rows(n_pix, C);
    const int ppb = (int)((n_pix + blocks - 1) / blocks);
    const int slots = 256 / (C / 4);
    DS_LAUNCH(bn_bwd_reduce_kernel, blocks, 256, (size_t)slots * C * 2 * 4, stream, g1, g2, act, z, mean, invstd,
              gy, partial, n_pix, C, ppb, blocks);
    return ds_last_launch_error();
}

extern "C" int ds_bn_bwd_apply_f32(const double *sums, long long count, const float *gy, const float *z,
                                   const float *mean, const float *invstd, const float *gamma, float *coef,
                                   float *ggamma, float *gbeta, float *gz, long long n_pix, int C, void *stream) {
    DS_REQUIRE(sums && gy && z && mean && invstd && gamma && coef && ggamma && gbeta && gz, DS_ERR_NULL);
    DS_REQUIRE(n_pix > 0 && count >= 0 && C >= 4 && (C % 4) == 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(bn_bwd_from_sums_kernel, ds_ceil_div(C, 256), 256, 0, stream, sums, (double)count, gamma, invstd, ggamma,
              gbeta, coef, C);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    const long long n_vec = n_pix * (C / 4);
    DS_LAUNCH(bn_bwd_apply_kernel, grid_for(n_vec), 256, 0, stream, gy, z, mean, invstd, (const float *)coef, gz, n_vec,
              C);
    return ds_last_launch_error();
}

// Workgroups (= partial rows) of the reduction: a workgroup walks its pixels 1024 / C at a time, so the pixels per
// workgroup shrink with the channel count (about 8 steps per thread) -- the 10x4 stage of a 256-utterance member
// has only 10 k pixels, and 256 of them per workgroup left 40 workgroups on 256 CUs -- bounded by 2048 rows.
extern "C" int ds_bn_bwd_partial_rows(long long n_pix, int C) {
    if (n_pix <= 0 || C <= 0) return DS_ERR_BAD_SHAPE;
    long long ppb = 8192 / C;
    if (ppb < 8) ppb = 8;
    long long blocks = (n_pix + ppb - 1) / ppb;
    if (blocks > 2048) blocks = 2048;
    return (int)blocks;
}

extern "C" int ds_bn_bwd_f32(const float *g1, const float *g2, const float *act, const float *z, const float *mean,
                             const float *invstd, const float *gamma, float *gy, float *partial, float *coef,
                             float *ggamma, float *gbeta, float *gz, long long n_pix, int C, void *stream) {
    DS_REQUIRE(g1 && z && mean && invstd && gamma && gy && partial && coef && ggamma && gbeta && gz, DS_ERR_NULL);
    DS_REQUIRE(n_pix > 0 && C >= 4 && (C % 4) == 0 && C <= 1024 && 256 % (C / 4) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(g1) && DS_ALIGNED16(z) && DS_ALIGNED16(gy) && DS_ALIGNED16(gz) && DS_ALIGNED16(mean) &&
                   DS_ALIGNED16(invstd) && DS_ALIGNED16(coef), DS_ERR_ALIGNMENT);
    const int blocks = ds_bn_bwd_partial_rows(n_pix, C);
    const int ppb = (int)((n_pix + blocks - 1) / blocks);
    const int slots = 256 / (C / 4);
    DS_LAUNCH(bn_bwd_reduce_kernel, blocks, 256, (size_t)slots * C * 2 * 4, stream, g1, g2, act, z, mean, invstd, gy,
              partial, n_pix, C, ppb, blocks);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(bn_bwd_finalize_kernel, ds_ceil_div(C, FOLD_C), 256, FOLD_R * FOLD_C * 2 * sizeof(double), stream,
              (const float *)partial, blocks, (double)n_pix, gamma, invstd, ggamma, gbeta, coef, C);
    rc = ds_last_launch_error();
    if (rc) return rc;
    const long long n_vec = n_pix * (C / 4);
    DS_LAUNCH(bn_bwd_apply_kernel, grid_for(n_vec), 256, 0, stream, (const float *)gy, z, mean, invstd,
              (const float *)coef, gz, n_vec, C);
    return ds_last_launch_error();
}

// ds_bn_bwd_f32 for a batch made of G members with their own batch statistics (Engine.forward_train_group: the three
// forwards of a triplet step as one batch), in four launches instead of 3 G + 2: g1 / g2 / act / z / gy / gz are
// [G * n_pix, C]; mean, invstd [G][C]; partial G * ds_bn_bwd_partial_rows(n_pix, C) * C * 2 floats; coef [G][3C];
// member_sums [2][G][C] scratch; ggamma / gbeta [C] = the members' dgamma / dbeta added in member order.
extern "C" int ds_bn_bwd_group_f32(const float *g1, const float *g2, const float *act, const float *z, const float *mean,
                                   const float *invstd, const float *gamma, float *gy, float *partial, float *coef,
                                   float *member_sums, float *ggamma, float *gbeta, float *gz, long long n_pix, int C,
                                   int G, void *stream) {
    DS_REQUIRE(g1 && z && mean && invstd && gamma && gy && partial && coef && member_sums && ggamma && gbeta && gz,
               DS_ERR_NULL);
    DS_REQUIRE(n_pix > 0 && G > 0 && G <= 64 && C >= 4 && (C % 4) == 0 && C <= 1024 && 256 % (C / 4) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(g1) && DS_ALIGNED16(z) && DS_ALIGNED16(gy) && DS_ALIGNED16(gz) && DS_ALIGNED16(mean) &&
                   DS_ALIGNED16(invstd) && DS_ALIGNED16(coef) && (!g2 || DS_ALIGNED16(g2)) && (!act || DS_ALIGNED16(act)),
               DS_ERR_ALIGNMENT);
    const int blocks = ds_bn_bwd_partial_rows(n_pix, C);
    const int ppb = (int)((n_pix + blocks - 1) / blocks);
    const int slots = 256 / (C / 4);
    DS_LAUNCH(bn_bwd_reduce_kernel, blocks * G, 256, (size_t)slots * C * 2 * 4, stream, g1, g2, act, z, mean, invstd, gy,
              partial, n_pix, C, ppb, blocks);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    const int n_cgroups = ds_ceil_div(C, FOLD_C);
    float *gg_m = member_sums, *gb_m = member_sums + (size_t)G * C;
    DS_LAUNCH(bn_bwd_finalize_group_kernel, n_cgroups * G, 256, FOLD_R * FOLD_C * 2 * sizeof(double), stream,
              (const float *)partial, blocks, (double)n_pix, gamma, invstd, gg_m, gb_m, coef, C, n_cgroups);
    rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(bn_member_sum_kernel, ds_ceil_div(C, 256), 256, 0, stream, (const float *)gg_m, (const float *)gb_m, ggamma,
              gbeta, G, C);
    rc = ds_last_launch_error();
    if (rc) return rc;
    const long long n_vec_member = n_pix * (C / 4);
    DS_LAUNCH(bn_bwd_apply_group_kernel, grid_for(n_vec_member * G), 256, 0, stream, (const float *)gy, z, mean, invstd,
              (const float *)coef, gz, n_vec_member, G, C);
    return ds_last_launch_error();
}

// ds_bn_bwd_group_f32 split at the point where data-parallel training exchanges the sums: the local
// reductions of all G members -> sums [G][2C+1] float64 (C pairs {sum gy, sum gy*xhat} and the member's pixel count)
// ... all-reduce by the caller ... -> coefficients, dgamma / dbeta and gz of all members.  Two + three launches.
extern "C" int ds_bn_bwd_group_reduce_f32(const float *g1, const float *g2, const float *act, const float *z,
                                          const float *mean, const float *invstd, float *gy, float *partial,
                                          double *sums, long long n_pix, int C, int G, void *stream) {
    DS_REQUIRE(g1 && z && mean && invstd && gy && partial && sums, DS_ERR_NULL);
    DS_REQUIRE(n_pix > 0 && G > 0 && G <= 64 && C >= 4 && (C % 4) == 0 && C <= 1024 && 256 % (C / 4) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(g1) && DS_ALIGNED16(z) && DS_ALIGNED16(gy) && DS_ALIGNED16(mean) && DS_ALIGNED16(invstd) &&
                   (!g2 || DS_ALIGNED16(g2)) && (!act || DS_ALIGNED16(act)), DS_ERR_ALIGNMENT);
    const int blocks = ds_bn_bwd_partial_rows(n_pix, C);
    const int ppb = (int)((n_pix + blocks - 1) / blocks);
    const int slots = 256 / (C / 4);
    DS_LAUNCH(bn_bwd_reduce_kernel, blocks * G, 256, (size_t)slots * C * 2 * 4, stream, g1, g2, act, z, mean, invstd, gy,
              partial, n_pix, C, ppb, blocks);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(partial_sum_f64_group_kernel, ds_ceil_div(C, FOLD_C) * G, 256, FOLD_R * FOLD_C * 2 * sizeof(double), stream,
              (const float *)partial, blocks, sums, (double)n_pix, C, ds_ceil_div(C, FOLD_C));
    return ds_last_launch_error();
}

extern "C" int ds_bn_bwd_group_apply_f32(const double *sums, const float *gy, const float *z, const float *mean,
                                         const float *invstd, const float *gamma, float *coef, float *member_sums,
                                         float *ggamma, float *gbeta, float *gz, long long n_pix, int C, int G,
                                         void *stream) {
    DS_REQUIRE(sums && gy && z && mean && invstd && gamma && coef && member_sums && ggamma && gbeta && gz, DS_ERR_NULL);
    DS_REQUIRE(n_pix > 0 && G > 0 && G <= 64 && C >= 4 && (C % 4) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(gy) && DS_ALIGNED16(z) && DS_ALIGNED16(gz) && DS_ALIGNED16(mean) && DS_ALIGNED16(invstd) &&
                   DS_ALIGNED16(coef), DS_ERR_ALIGNMENT);
    float *gg_m = member_sums, *gb_m = member_sums + (size_t)G * C;
    DS_LAUNCH(bn_bwd_from_sums_group_kernel, ds_ceil_div(C, 256) * G, 256, 0, stream, sums, gamma, invstd, gg_m, gb_m,
              coef, C, ds_ceil_div(C, 256));
    int rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(bn_member_sum_kernel, ds_ceil_div(C, 256), 256, 0, stream, (const float *)gg_m, (const float *)gb_m, ggamma,
              gbeta, G, C);
    rc = ds_last_launch_error();
    if (rc) return rc;
    const long long n_vec_member = n_pix * (C / 4);
    DS_LAUNCH(bn_bwd_apply_group_kernel, grid_for(n_vec_member * G), 256, 0, stream, gy, z, mean, invstd,
              (const float *)coef, gz, n_vec_member, G, C);
    return ds_last_launch_error();
}

// The second half of ds_bn_bwd_group_f32 alone, for partial sums that were produced elsewhere (the data-gradient kernel
// whose epilogue is the reduction: ds_conv_dgrad_bnbwd_bf16): n_partial rows of [C][2] per member, consecutive.
extern "C" int ds_bn_bwd_group_finish_f32(const float *partial, int n_partial, const float *gy, const float *z,
                                          const float *mean, const float *invstd, const float *gamma, float *coef,
                                          float *member_sums, float *ggamma, float *gbeta, float *gz, long long n_pix,
                                          int C, int G, void *stream) {
    DS_REQUIRE(partial && gy && z && mean && invstd && gamma && coef && member_sums && ggamma && gbeta && gz, DS_ERR_NULL);
    DS_REQUIRE(n_partial > 0 && n_pix > 0 && G > 0 && G <= 64 && C >= 4 && (C % 4) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(DS_ALIGNED16(gy) && DS_ALIGNED16(z) && DS_ALIGNED16(gz) && DS_ALIGNED16(mean) && DS_ALIGNED16(invstd) &&
                   DS_ALIGNED16(coef), DS_ERR_ALIGNMENT);
    const int n_cgroups = ds_ceil_div(C, FOLD_C);
    float *gg_m = member_sums, *gb_m = member_sums + (size_t)G * C;
    DS_LAUNCH(bn_bwd_finalize_group_kernel, n_cgroups * G, 256, FOLD_R * FOLD_C * 2 * sizeof(double), stream, partial,
              n_partial, (double)n_pix, gamma, invstd, gg_m, gb_m, coef, C, n_cgroups);
    int rc = ds_last_launch_error();
    if (rc) return rc;
    DS_LAUNCH(bn_member_sum_kernel, ds_ceil_div(C, 256), 256, 0, stream, (const float *)gg_m, (const float *)gb_m, ggamma,
              gbeta, G, C);
    rc = ds_last_launch_error();
    if (rc) return rc;
    const long long n_vec_member = n_pix * (C / 4);
    DS_LAUNCH(bn_bwd_apply_group_kernel, grid_for(n_vec_member * G), 256, 0, stream, gy, z, mean, invstd,
              (const float *)coef, gz, n_vec_member, G, C);
    return ds_last_launch_error();
}

// The forward counterpart: per-tile partial statistics of G members (each n_partial rows of [C][2]) -> sums [G][2C+1]
// float64 in ONE launch (what the per-BatchNorm-layer all-reduce of data-parallel training carries).
extern "C" int ds_partial_sum_f64_group(const float *partial, int n_partial, double *sums, long long count, int C, int G,
                                        void *stream) {
    DS_REQUIRE(partial && sums, DS_ERR_NULL);
    DS_REQUIRE(n_partial > 0 && C > 0 && G > 0 && G <= 64 && count > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(partial_sum_f64_group_kernel, ds_ceil_div(C, FOLD_C) * G, 256, FOLD_R * FOLD_C * 2 * sizeof(double), stream, partial,
              n_partial, sums, (double)count, C, ds_ceil_div(C, FOLD_C));
    return ds_last_launch_error();
}

extern "C" int ds_colsum_f32(const float *x, float *out, int R, int C, void *stream) {
    DS_REQUIRE(x && out, DS_ERR_NULL);
    DS_REQUIRE(R > 0 && C > 0, DS_ERR_BAD_SHAPE);
    DS_LAUNCH(colsum_kernel, ds_ceil_div(C, 32), 256, 8 * 32 * sizeof(float), stream, x, out, R, C);
    return ds_last_launch_error();
}
