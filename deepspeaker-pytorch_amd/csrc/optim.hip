// optim.hip -- fused multi-tensor optimizer steps.
//
// The reference updates its 40 parameter tensors with torch.optim.{Adagrad,SGD,Adam}
// (train_triplet.py:369-383, stepped at :224,291): ~5 elementwise launches per tensor per step.
// Here ONE launch updates every tensor: the host builds (once) a chunk table -- which tensor and
// which 16384-element slice each workgroup owns -- and the kernel streams param / grad / state
// through registers once (HBM-bound: 16 B read + 8 B written per element for Adagrad).
// Arithmetic follows the single-tensor torch implementations exactly (same operation order).
#include <ds_device.h>
#include <string.h>
#include "ds_common.h"

namespace {

constexpr int OPT_CHUNK = 16384;

struct OptTables {
    float *const *params;
    const float *const *grads;
    float *const *s1;        // Adagrad: sum;  SGD: momentum buffer;  Adam: exp_avg
    float *const *s2;        // Adam: exp_avg_sq
    const long long *numel;
    const int *chunk_tensor;
    const int *chunk_index;  // chunk number inside its tensor
    const int *skip;         // nullable: a device flag; non-zero = leave everything untouched (gradient overflow of a
                             // loss-scaled step, ds_nonfinite_flag_f32) -- decided on the device, no host round trip
};

// torch.optim.Adagrad (single-tensor path): grad += wd*p; sum += grad*grad; p -= clr * grad / (sqrt(sum) + eps).
// `step` (nullable): the step count lives ON THE DEVICE (a captured step replays with the count of that replay, not of the
// capture: ds_optim_step_inc bumps it once per optimizer step); clr = lr / (1 + (step - 1) lr_decay) is then taken from it
// in double precision, the way the host computes the `clr` it passes otherwise.
__global__ void __launch_bounds__(256) adagrad_kernel(const OptTables t, float clr, float wd, float eps, const int *step,
                                                      double lr, double lr_decay) {
    if (t.skip != nullptr && *t.skip != 0) return;
    if (step != nullptr) clr = (float)(lr / (1.0 + ((double)*step - 1.0) * lr_decay));
    const int ti = t.chunk_tensor[blockIdx.x];
    const long long n = t.numel[ti];
    const long long i0 = (long long)t.chunk_index[blockIdx.x] * OPT_CHUNK;
    float *p = t.params[ti];
    const float *g = t.grads[ti];
    float *s = t.s1[ti];
    for (int k = threadIdx.x; k < OPT_CHUNK; k += 256) {
        const long long i = i0 + k;
        if (i >= n) break;
        float gr = g[i];
        const float pv = p[i];
        if (wd != 0.0f) gr = gr + wd * pv;
        const float sm = s[i] + gr * gr;
        s[i] = sm;
        p[i] = pv - clr * (gr / (sqrtf(sm) + eps));
    }
}

// torch.optim.SGD: grad += wd*p; buf = first ? grad : momentum*buf + (1-dampening)*grad; p -= lr*buf
__global__ void __launch_bounds__(256) sgd_kernel(const OptTables t, float lr, float momentum, float dampening, float wd,
                                                  int first, const int *step) {
    if (t.skip != nullptr && *t.skip != 0) return;
    if (step != nullptr) first = *step == 1;   // device-side step count: the momentum buffers start with the first COUNTED step
    const int ti = t.chunk_tensor[blockIdx.x];
    const long long n = t.numel[ti];
    const long long i0 = (long long)t.chunk_index[blockIdx.x] * OPT_CHUNK;
    float *p = t.params[ti];
    const float *g = t.grads[ti];
    float *b = momentum != 0.0f ? t.s1[ti] : nullptr;      // plain SGD has no state table at all
    for (int k = threadIdx.x; k < OPT_CHUNK; k += 256) {
        const long long i = i0 + k;
        if (i >= n) break;
        float gr = g[i];
        const float pv = p[i];
        if (wd != 0.0f) gr = gr + wd * pv;
        if (momentum != 0.0f) {
            const float bv = first ? gr : momentum * b[i] + (1.0f - dampening) * gr;
            b[i] = bv;
            gr = bv;
        }
        p[i] = pv - lr * gr;
    }
}

// torch.optim.Adam (no amsgrad): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)
__global__ void __launch_bounds__(256) adam_kernel(const OptTables t, float lr, float b1, float b2, float eps, float wd,
                                                   float bc1, float bc2_sqrt, const int *step, double b1d, double b2d) {
    if (t.skip != nullptr && *t.skip != 0) return;
    if (step != nullptr) {                      // bias corrections from the device-side step count (see adagrad_kernel)
        bc1 = (float)(1.0 - pow(b1d, (double)*step));
        bc2_sqrt = (float)sqrt(1.0 - pow(b2d, (double)*step));
    }
    const int ti = t.chunk_tensor[blockIdx.x];
    const long long n = t.numel[ti];
    const long long i0 = (long long)t.chunk_index[blockIdx.x] * OPT_CHUNK;
    float *p = t.params[ti];
    const float *g = t.grads[ti];
    float *m = t.s1[ti], *v = t.s2[ti];
    const float step_size = lr / bc1;
    for (int k = threadIdx.x; k < OPT_CHUNK; k += 256) {
        const long long i = i0 + k;
        if (i >= n) break;
        float gr = g[i];
        const float pv = p[i];
        if (wd != 0.0f) gr = gr + wd * pv;
        const float mv = m[i] + (gr - m[i]) * (1.0f - b1);          // lerp_, as torch does
        const float vv = b2 * v[i] + (1.0f - b2) * gr * gr;
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = pv - step_size * (mv / denom);
    }
}

// flag = 1 if any of x[0..n) is inf or NaN (every finder stores the same value: no atomics, no ordering needed); the flag
// is NOT cleared here -- several tensors of one step accumulate into it
__global__ void __launch_bounds__(256) nonfinite_flag_kernel(const float *x, long long n, int *flag) {
    const long long stride = (long long)gridDim.x * 256 * 4;
    bool bad = false;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            const f32x4 v = *(const f32x4 *)(x + i);
            // (v - v) is 0 for finite values and NaN for inf / NaN
            bad |= !((v[0] - v[0]) + (v[1] - v[1]) + (v[2] - v[2]) + (v[3] - v[3]) == 0.0f);
        } else {
            for (long long j = i; j < n; ++j) bad |= !(x[j] - x[j] == 0.0f);
        }
    }
    if (bad) *flag = 1;
}

// Small host tables -> device memory WITHOUT a host buffer the copy engine must read later: the values travel as kernel
// arguments (by value, up to 2 KiB per launch), so the call is legal inside a stream capture and a replayed graph
// re-writes the same values.  (A pinned staging buffer + asynchronous copy -- what the fused optimizers used for their
// pointer tables until round 6 -- records an event torch's host allocator later queries: "operation not permitted on an
// event last recorded in a capturing stream".)
__global__ void __launch_bounds__(64) step_inc_kernel(int *step, const int *skip) {
    if (threadIdx.x == 0 && !(skip != nullptr && *skip != 0)) *step += 1;
}

struct FillArg { unsigned v[512]; };
__global__ void __launch_bounds__(256) fill_words_kernel(const FillArg a, unsigned *dst, int n_words) {
    for (int i = threadIdx.x; i < n_words; i += 256) dst[i] = a.v[i];
}

}  // namespace

extern "C" int ds_fill_bytes(void *dst, const void *host_src, int bytes, void *stream) {
    DS_REQUIRE(dst && host_src, DS_ERR_NULL);
    DS_REQUIRE(bytes > 0 && bytes <= (int)sizeof(FillArg) && (bytes & 3) == 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE((((size_t)dst) & 3) == 0, DS_ERR_ALIGNMENT);
    FillArg a;
    memcpy(a.v, host_src, (size_t)bytes);
    DS_LAUNCH(fill_words_kernel, 1, 256, 0, stream, a, (unsigned *)dst, bytes / 4);
    return ds_last_launch_error();
}

extern "C" int ds_nonfinite_flag_f32(const float *x, long long n, int *flag, void *stream) {
    DS_REQUIRE(x && flag, DS_ERR_NULL);
    DS_REQUIRE(n > 0, DS_ERR_BAD_SHAPE);
    DS_REQUIRE(((size_t)x & 15) == 0, DS_ERR_ALIGNMENT);
    const long long want = (n + 4095) / 4096;
    DS_LAUNCH(nonfinite_flag_kernel, (int)(want < 1024 ? want : 1024), 256, 0, stream, x, n, flag);
    return ds_last_launch_error();
}

extern "C" int ds_optim_chunk_elems(void) { return OPT_CHUNK; }

#define DS_OPT_ARGS                                                                                              \
    const void *params, const void *grads, const void *state1, const void *state2, const long long *numel,     \
        const int *chunk_tensor, const int *chunk_index, int n_chunks

static inline int opt_tables(OptTables &t, DS_OPT_ARGS) {
    DS_REQUIRE(params && grads && numel && chunk_tensor && chunk_index, DS_ERR_NULL);
    DS_REQUIRE(n_chunks > 0, DS_ERR_BAD_SHAPE);
    t.params = (float *const *)params;
    t.grads = (const float *const *)grads;
    t.s1 = (float *const *)state1;
    t.s2 = (float *const *)state2;
    t.numel = numel;
    t.chunk_tensor = chunk_tensor;
    t.chunk_index = chunk_index;
    t.skip = nullptr;
    return DS_OK;
}

extern "C" int ds_adagrad_step_f32(DS_OPT_ARGS, float clr, float weight_decay, float eps, const int *skip_flag,
                                   void *stream) {
    OptTables t;
    int rc = opt_tables(t, params, grads, state1, state2, numel, chunk_tensor, chunk_index, n_chunks);
    if (rc) return rc;
    DS_REQUIRE(state1, DS_ERR_NULL);
    t.skip = skip_flag;
    DS_LAUNCH(adagrad_kernel, n_chunks, 256, 0, stream, t, clr, weight_decay, eps, (const int *)nullptr, 0.0, 0.0);
    return ds_last_launch_error();
}

// the same step with the step count read on the device (`step_count`, int32 [1], see ds_optim_step_inc): for steps that
// are captured into a HIP graph and replayed
extern "C" int ds_adagrad_step_dev_f32(DS_OPT_ARGS, double lr, double lr_decay, float weight_decay, float eps,
                                       const int *step_count, const int *skip_flag, void *stream) {
    OptTables t;
    int rc = opt_tables(t, params, grads, state1, state2, numel, chunk_tensor, chunk_index, n_chunks);
    if (rc) return rc;
    DS_REQUIRE(state1 && step_count, DS_ERR_NULL);
    t.skip = skip_flag;
    DS_LAUNCH(adagrad_kernel, n_chunks, 256, 0, stream, t, 0.0f, weight_decay, eps, step_count, lr, lr_decay);
    return ds_last_launch_error();
}

// *step_count += 1 unless *skip_flag is set (torch.amp.GradScaler does not count a skipped step either)
extern "C" int ds_optim_step_inc(int *step_count, const int *skip_flag, void *stream) {
    DS_REQUIRE(step_count, DS_ERR_NULL);
    DS_LAUNCH(step_inc_kernel, 1, 64, 0, stream, step_count, skip_flag);
    return ds_last_launch_error();
}

extern "C" int ds_sgd_step_f32(DS_OPT_ARGS, float lr, float momentum, float dampening, float weight_decay,
                               int first_step, const int *skip_flag, void *stream) {
    OptTables t;
    int rc = opt_tables(t, params, grads, state1, state2, numel, chunk_tensor, chunk_index, n_chunks);
    if (rc) return rc;
    DS_REQUIRE(momentum == 0.0f || state1, DS_ERR_NULL);
    t.skip = skip_flag;
    DS_LAUNCH(sgd_kernel, n_chunks, 256, 0, stream, t, lr, momentum, dampening, weight_decay, first_step, (const int *)nullptr);
    return ds_last_launch_error();
}

extern "C" int ds_sgd_step_dev_f32(DS_OPT_ARGS, float lr, float momentum, float dampening, float weight_decay,
                                   const int *step_count, const int *skip_flag, void *stream) {
    OptTables t;
    int rc = opt_tables(t, params, grads, state1, state2, numel, chunk_tensor, chunk_index, n_chunks);
    if (rc) return rc;
    DS_REQUIRE((momentum == 0.0f || state1) && step_count, DS_ERR_NULL);
    t.skip = skip_flag;
    DS_LAUNCH(sgd_kernel, n_chunks, 256, 0, stream, t, lr, momentum, dampening, weight_decay, 0, step_count);
    return ds_last_launch_error();
}

extern "C" int ds_adam_step_f32(DS_OPT_ARGS, float lr, float beta1, float beta2, float eps, float weight_decay,
                                float bias_correction1, float bias_correction2_sqrt, const int *skip_flag, void *stream) {
    OptTables t;
    int rc = opt_tables(t, params, grads, state1, state2, numel, chunk_tensor, chunk_index, n_chunks);
    if (rc) return rc;
    DS_REQUIRE(state1 && state2, DS_ERR_NULL);
    t.skip = skip_flag;
    DS_LAUNCH(adam_kernel, n_chunks, 256, 0, stream, t, lr, beta1, beta2, eps, weight_decay, bias_correction1,
              bias_correction2_sqrt, (const int *)nullptr, 0.0, 0.0);
    return ds_last_launch_error();
}

extern "C" int ds_adam_step_dev_f32(DS_OPT_ARGS, float lr, double beta1, double beta2, float eps, float weight_decay,
                                    const int *step_count, const int *skip_flag, void *stream) {
    OptTables t;
    int rc = opt_tables(t, params, grads, state1, state2, numel, chunk_tensor, chunk_index, n_chunks);
    if (rc) return rc;
    DS_REQUIRE(state1 && state2 && step_count, DS_ERR_NULL);
    t.skip = skip_flag;
    DS_LAUNCH(adam_kernel, n_chunks, 256, 0, stream, t, lr, (float)beta1, (float)beta2, eps, weight_decay, 1.0f, 1.0f, step_count,
              beta1, beta2);
    return ds_last_launch_error();
}
