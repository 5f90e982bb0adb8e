// TEST INFRASTRUCTURE -- host stand-in for csrc/ds_device.h (same API, emulated semantics).
// The MFMA model follows the documented gfx950 fragment layout of v_mfma_f32_32x32x2_f32:
//   lane l supplies A[i = l&31][k = l>>5], B[k = l>>5][j = l&31];
//   D[row = (reg&3) + 8*(reg>>2) + 4*(l>>5)][col = l&31], k accumulated in order with fmaf.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline f32x16 ds_mfma_32x32x2_f32(float a, float b, f32x16 c) {
    float A[64], B[64];
    emu::wave_exchange(a, A);
    emu::wave_exchange(b, B);
    const int lane = threadIdx.x & 63;
    const int col = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        float acc = c[r];
        acc = fmaf(A[row], B[col], acc);                 // k = 0
        acc = fmaf(A[row + 32], B[col + 32], acc);       // k = 1
        c[r] = acc;
    }
    return c;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_32x32x16_bf16 model: lane l = i + 32 g holds k-slots (g, 0..7) of row/column i; products of
// bf16 values are exact in f32, the 16-term sum is accumulated in double and rounded once into the f32
// accumulator (the hardware's internal summation order is not architecturally specified).
static inline f32x16 ds_mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    float A[8][64], B[8][64];
    for (int j = 0; j < 8; ++j) {
        emu::wave_exchange((float)a[j], A[j]);
        emu::wave_exchange((float)b[j], B[j]);
    }
    const int lane = threadIdx.x & 63;
    const int col = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        double s = 0.0;
        for (int g = 0; g < 2; ++g)
            for (int j = 0; j < 8; ++j) s += (double)A[j][row + 32 * g] * (double)B[j][col + 32 * g];
        c[r] = (float)((double)c[r] + s);
    }
    return c;
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// two f32 values -> their bf16 "hi" parts and the bf16 "lo" parts of the remainders (split operands: x = hi + lo up
// to 2^-17 |x|), each pair packed into one dword with ONE packed conversion (v_cvt_pk_bf16_f32); same bits as the
// scalar (__bf16)v / (__bf16)(v - (float)hi) sequence
typedef float ds_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 ds_f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 ds_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ds_split_bf16x2(float a, float b, unsigned &hi, unsigned &lo) {
    const ds_bf16x2 h = __builtin_convertvector(ds_f32x2{a, b}, ds_bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    const float ha = __builtin_bit_cast(float, hi << 16), hb = __builtin_bit_cast(float, hi & 0xFFFF0000u);
    const ds_bf16x2 l = __builtin_convertvector(ds_f32x2{a - ha, b - hb}, ds_bf16x2);
    lo = __builtin_bit_cast(unsigned, l);
}
// v_mfma_f32_32x32x16_f16: same fragment layout; fp16 products are exact in f32, summed as above
static inline f32x16 ds_mfma_32x32x16_f16(f16x8 a, f16x8 b, f32x16 c) {
    float A[8][64], B[8][64];
    for (int j = 0; j < 8; ++j) {
        emu::wave_exchange((float)a[j], A[j]);
        emu::wave_exchange((float)b[j], B[j]);
    }
    const int lane = threadIdx.x & 63;
    const int col = lane & 31, hi = lane >> 5;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
        double s = 0.0;
        for (int g = 0; g < 2; ++g)
            for (int j = 0; j < 8; ++j) s += (double)A[j][row + 32 * g] * (double)B[j][col + 32 * g];
        c[r] = (float)((double)c[r] + s);
    }
    return c;
}

static inline float ds_shfl_xor(float v, int mask) {
    float all[64];
    emu::wave_exchange(v, all);
    return all[(threadIdx.x & 63) ^ mask];
}
static inline double ds_shfl_xor_f64(double v, int mask) {         // the two dwords travel separately, bit for bit
    unsigned w[2];
    memcpy(w, &v, 8);
    for (int h = 0; h < 2; ++h) {
        float f, all[64];
        memcpy(&f, &w[h], 4);
        emu::wave_exchange(f, all);
        memcpy(&w[h], &all[(threadIdx.x & 63) ^ mask], 4);
    }
    double r;
    memcpy(&r, w, 8);
    return r;
}
static inline float ds_shfl_down(float v, int d) {
    float all[64];
    emu::wave_exchange(v, all);
    const int l = (threadIdx.x & 63) + d;
    return l < 64 ? all[l] : v;
}
static inline int ds_shfl_xor_i(int v, int mask) {
    float f, all[64];
    memcpy(&f, &v, 4);
    emu::wave_exchange(f, all);
    int r;
    memcpy(&r, &all[(threadIdx.x & 63) ^ mask], 4);
    return r;
}
// transposing LDS read: lane i of a 16-lane group supplies piece (row i>>2, quad i&3), receives column i
static inline bf16x4 ds_read_tr16_b64(const char *lds_piece) {
    float mine[2], all0[64], all1[64];
    memcpy(mine, lds_piece, 8);
    emu::wave_exchange(mine[0], all0);
    emu::wave_exchange(mine[1], all1);
    const int lane = threadIdx.x & 63, g0 = lane & ~15, i = lane & 15;
    unsigned short col[4];
    for (int r = 0; r < 4; ++r) {
        const int src = g0 + 4 * r + (i >> 2);          // the lane that supplied row r, quad i>>2
        unsigned short piece[4];
        memcpy(piece, &all0[src], 4);
        memcpy(piece + 2, &all1[src], 4);
        col[r] = piece[i & 3];
    }
    bf16x4 out;
    memcpy(&out, col, 8);
    return out;
}
static inline int ds_div_small(int n, int d, float rcp) {
    (void)rcp;
    return n / d;
}
struct ds_buffer { char *base; unsigned bytes; };
constexpr unsigned DS_BUFFER_OOB = 0xFFFFFFF0u;
static inline ds_buffer ds_make_buffer(const void *base, unsigned bytes) { return ds_buffer{(char *)base, bytes}; }
static inline f32x4 ds_buffer_load_f32x4(ds_buffer b, unsigned byte_off) {
    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
    if ((unsigned long long)byte_off + 16 <= b.bytes) memcpy(&v, b.base + byte_off, 16);
    return v;
}
static inline void ds_buffer_store_f32x4(ds_buffer b, unsigned byte_off, f32x4 v) {
    if ((unsigned long long)byte_off + 16 <= b.bytes) memcpy(b.base + byte_off, &v, 16);
}
static inline void ds_buffer_store_out_f32x4(ds_buffer b, unsigned byte_off, f32x4 v) { ds_buffer_store_f32x4(b, byte_off, v); }
typedef unsigned int ds_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int ds_u32x2 __attribute__((ext_vector_type(2)));
static inline void ds_buffer_store_b64(ds_buffer b, unsigned byte_off, ds_u32x2 v) {
    if ((unsigned long long)byte_off + 8 <= b.bytes) memcpy(b.base + byte_off, &v, 8);
}
static inline ds_u32x2 ds_buffer_load_b64(ds_buffer b, unsigned byte_off) {
    ds_u32x2 v = {0u, 0u};
    if ((unsigned long long)byte_off + 8 <= b.bytes) memcpy(&v, b.base + byte_off, 8);
    return v;
}
static inline float ds_buffer_load_f32(ds_buffer b, unsigned byte_off) {
    float v = 0.0f;
    if ((unsigned long long)byte_off + 4 <= b.bytes) memcpy(&v, b.base + byte_off, 4);
    return v;
}
static inline void ds_buffer_store_f32(ds_buffer b, unsigned byte_off, float v) {
    if ((unsigned long long)byte_off + 4 <= b.bytes) memcpy(b.base + byte_off, &v, 4);
}
static inline void ds_lds_barrier() { __syncthreads(); }
static inline void ds_wave_sync() {
    float all[64];
    emu::wave_exchange(0.0f, all);
}
static inline unsigned long long ds_ballot(int pred) {
    float all[64];
    emu::wave_exchange(pred ? 1.0f : 0.0f, all);
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (all[l] != 0.0f) m |= 1ull << l;
    return m;
}

#define DS_OPAQUE_VGPR(x) ((void)0)
#define DS_ONE_WAVE_PER_SIMD

static inline float *ds_dynamic_lds() { return emu::dynamic_lds(); }

#define DS_LAUNCH(kernel, grid, block, lds_bytes, stream, ...) \
    emu::launch((int)(grid), (int)(block), (size_t)(lds_bytes), [=]() { kernel(__VA_ARGS__); })

#define DS_LAUNCH_BIG_LDS(kernel, grid, block, lds_bytes, stream, ...) \
    emu::launch((int)(grid), (int)(block), (size_t)(lds_bytes), [=]() { kernel(__VA_ARGS__); }, true)

// "compute units" of the emulated device: small, so that persistent kernels walk several tiles per workgroup in the
// tests (DS_EMUL_CUS overrides)
static inline int ds_cu_count() {
    const char *e = getenv("DS_EMUL_CUS");
    const int n = e ? atoi(e) : 2;
    return n > 0 ? n : 2;
}

// dynamic tile scheduling of the persistent kernels (see csrc/ds_device.h): blocks run on several OS threads here
constexpr int DS_SCHED_RING = 8, DS_SCHED_WORDS = 16, DS_SCHED_DONE = 8;
static inline unsigned ds_atomic_inc(unsigned *p) { return __atomic_fetch_add(p, 1u, __ATOMIC_RELAXED); }
unsigned *ds_sched_slot(void *stream);          // bn_pack.hip
static inline int ds_uniform(int v) { return v; }
static inline float ds_bn_affine(float z, float scale, float shift) { return __builtin_fmaf(z, scale, shift); }

static inline int ds_last_launch_error() { return 0; }

// launch timing does not exist on the host: the armed pair is kept (the entry points compile unchanged) and no launch
// ever counts itself
#include <unistd.h>
struct ds_timing_arm_t { hipEvent_t start, stop; int armed, launches; };
inline thread_local ds_timing_arm_t ds_timing_arm_state = {nullptr, nullptr, 0, 0};
