// ds_device.h -- the thin layer between the kernels and the gfx950 toolchain:
// MFMA / cross-lane intrinsics, the dynamic-LDS base and the launch macro.
// Kernels include it as <ds_device.h>; tests/emul/ provides a host stand-in of the
// same name so the unmodified kernel sources can be exercised lane-by-lane on a CPU.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f32_32x32x2_f32: D = A(32x2) * B(2x32) + C, exact f32 (fma chain over k).
// lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
__device__ __forceinline__ f32x16 ds_mfma_32x32x2_f32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// v_mfma_f32_32x32x16_bf16: D = A(32x16) * B(16x32) + C, bf16 inputs, f32 accumulate.  Lane l supplies
// 8 k-values of row (A) / column (B) l&31, in k-slot group l>>5; the kernels give slot (g, j) the same
// channel 8g+j on both operands, which is all the contraction needs.  C/D layout as above.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 ds_mfma_32x32x16_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// v_mfma_f32_32x32x16_f16: the same shape and fragment layout with fp16 inputs
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
__device__ __forceinline__ f32x16 ds_mfma_32x32x16_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// cross-lane exchange inside one 64-lane wavefront
__device__ __forceinline__ float ds_shfl_xor(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ float ds_shfl_down(float v, int d) { return __shfl_down(v, d, 64); }
__device__ __forceinline__ double ds_shfl_xor_f64(double v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ int ds_shfl_xor_i(int v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ unsigned long long ds_ballot(int pred) { return __ballot(pred); }
// orders this wavefront's LDS traffic: writes issued before it are visible to every lane's reads after it
// (LDS executes one wavefront's instructions in order, so no workgroup barrier is needed for wave-private data)
__device__ __forceinline__ void ds_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier that orders LDS traffic ONLY: __syncthreads() also drains the vector-memory counter, i.e. it waits
// for every global load in flight -- inside a software-pipelined loop that is the youngest filter-ring refill, a full
// L2 round trip per barrier.  Here only this wave's LDS operations are waited for before the barrier.
__device__ __forceinline__ void ds_lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ds_read_b64_tr_b16 (gfx950 transposing LDS read).  Within each group of 16 lanes, lane i supplies the
// address of an 8-byte piece -- row i>>2, column quad i&3 of a [4 rows][16 columns] block of 16-bit
// elements -- and receives COLUMN i of that block (rows 0..3).  Verified on MI355X with row strides of
// 32 / 48 / 64 bytes; addresses must be 8-byte aligned.
typedef short ds_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x4 ds_read_tr16_b64(const char *lds_piece) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                          (__attribute__((address_space(3))) ds_s16x4 *)lds_piece));
}

// n / d for 0 <= n < 2^24 with a reciprocal computed once per divisor (rcp = 1.0f / d): a handful of
// instructions instead of the ~40 of an integer division (index tables are built per workgroup)
__device__ __forceinline__ int ds_div_small(int n, int d, float rcp) {
    int q = (int)((float)n * rcp);
    q -= (q * d > n) ? 1 : 0;
    q += ((q + 1) * d <= n) ? 1 : 0;
    return q;
}

// Raw buffer access (stride 0, byte offsets): an offset outside [0, bytes) reads zeros / drops the store in
// hardware, which keeps ragged-tile epilogues free of branches -- and of the conservative s_waitcnt the
// compiler must place around conditionally executed memory instructions.
typedef __amdgpu_buffer_rsrc_t ds_buffer;
typedef unsigned int ds_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned DS_BUFFER_OOB = 0xFFFFFFF0u;
__device__ __forceinline__ ds_buffer ds_make_buffer(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 ds_buffer_load_f32x4(ds_buffer b, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b, (int)byte_off, 0, 0));
}
__device__ __forceinline__ void ds_buffer_store_f32x4(ds_buffer b, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ds_u32x4, v), b, (int)byte_off, 0, 0);
}
// The ACTIVATION stores of the convolution epilogues (a layer's output: written once, read by the next launch).  The
// cache-policy operand of the store is a build-time knob (tools/f16_ab.py variants: 0 = default write-back, 2 = nt,
// 17 = sc0 sc1 write-through).  Measured (round 5, same process, profiles/r05_run7_store_policy_ab.txt): no difference --
// ten forwards back to back take 1753 / 1747 / 1753 us each, which is also the sum of their launches: the ~10 us
// "gaps" rocprofv3 shows between consecutive convolution launches are the profiler's serialisation, not the stream.
#ifndef DS_EPI_STORE_AUX
#define DS_EPI_STORE_AUX 0
#endif
__device__ __forceinline__ void ds_buffer_store_out_f32x4(ds_buffer b, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ds_u32x4, v), b, (int)byte_off, 0, DS_EPI_STORE_AUX);
}
typedef unsigned int ds_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ds_buffer_store_b64(ds_buffer b, unsigned byte_off, ds_u32x2 v) {
    __builtin_amdgcn_raw_buffer_store_b64(v, b, (int)byte_off, 0, 0);
}
__device__ __forceinline__ ds_u32x2 ds_buffer_load_b64(ds_buffer b, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b64(b, (int)byte_off, 0, 0);
}
__device__ __forceinline__ float ds_buffer_load_f32(ds_buffer b, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b, (int)byte_off, 0, 0));
}
__device__ __forceinline__ void ds_buffer_store_f32(ds_buffer b, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), b, (int)byte_off, 0, 0);
}

// Hides how a (per-lane) value was computed: derived addresses are then recomputed where they are used instead of
// being hoisted out of the enclosing loop into dozens of live registers.
#define DS_OPAQUE_VGPR(x) asm volatile("" : "+v"(x))

// Kernels built around a register tile that takes most of the 512-entry file run one wavefront per SIMD by design;
// saying so lets the register allocator use the whole file instead of aiming at a higher occupancy.
#define DS_ONE_WAVE_PER_SIMD __attribute__((amdgpu_waves_per_eu(1, 1)))

// 16-byte aligned base of the dynamic LDS allocation (no static __shared__ objects are
// declared anywhere, so the base is the start of the workgroup's LDS segment)
__device__ __forceinline__ float *ds_dynamic_lds() {
    extern __shared__ __attribute__((aligned(16))) float ds_lds_base[];
    return ds_lds_base;
}

#define DS_LAUNCH(kernel, grid, block, lds_bytes, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (lds_bytes), (hipStream_t)(stream), __VA_ARGS__)

// Launch with more than the default 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU): the per-kernel
// opt-in is set once per process (immutable function attribute, not launch state).
// ... and, when a pair of timing events is ARMED on this thread (ds_launch_timing_arm: the bench's live roofline), the
// launch carries them itself: hipExtLaunchKernelGGL binds start / stop to the dispatch's own completion signal, so the
// events read the kernel's execution time and no marker packet is queued before or after it (a hipEventRecord pair
// around each launch costs ~10 us of queue bubbles per launch -- 4 % of the eval step).  `launches` counts the
// big-LDS launches since arming: the caller checks that the call it timed was exactly one of them.
struct ds_timing_arm_t { hipEvent_t start, stop; int armed, launches; };
inline thread_local ds_timing_arm_t ds_timing_arm_state = {nullptr, nullptr, 0, 0};

#define DS_LAUNCH_BIG_LDS(kernel, grid, block, lds_bytes, stream, ...)                                          \
    do {                                                                                                         \
        static const hipError_t ds_attr_rc_ = hipFuncSetAttribute(                                               \
            (const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                       \
        (void)ds_attr_rc_;                                                                                       \
        ds_timing_arm_t &ds_arm_ = ds_timing_arm_state;                                                          \
        ++ds_arm_.launches;                                                                                      \
        if (ds_arm_.armed) {                                                                                     \
            ds_arm_.armed = 0;                                                                                   \
            hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), (lds_bytes), (hipStream_t)(stream),           \
                                  ds_arm_.start, ds_arm_.stop, 0, __VA_ARGS__);                                  \
        } else {                                                                                                 \
            hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (lds_bytes), (hipStream_t)(stream), __VA_ARGS__); \
        }                                                                                                        \
    } while (0)

// Compute units of the current device (persistent kernels size their grid by it); queried once per process.
static inline int ds_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
            n = v;
        else
            n = 256;                            // MI355X
    }
    return n;
}

// ---- dynamic tile scheduling of the persistent kernels ----
// A persistent workgroup takes its first tile from its block index and every further one from a device-side counter:
// a workgroup that starts late -- its CU was busy with another stream's kernel -- simply takes fewer tiles (a static
// stride over the tiles makes the kernel as slow as its unluckiest workgroup: measured 302 -> 617 us on one launch next
// to a side stream).  A slot is 16 words: next[0..7] (one queue per XCD, or only [0]), done at [8].  The last workgroup
// to finish zeroes the slot again, so a slot needs no memset between launches.  Slots are private to what can run
// concurrently: a ring per (device, stream) for eager launches -- a stream orders its own launches -- and a slot of its
// own, for good, for every launch captured into a graph (ds_sched_slot, bn_pack.hip).  They are the ONE kind of device
// memory the library owns: 64-byte slots carved from 64 KiB chunks allocated and zeroed on first use.
constexpr int DS_SCHED_RING = 8, DS_SCHED_WORDS = 16, DS_SCHED_DONE = 8;
__device__ __forceinline__ unsigned ds_atomic_inc(unsigned *p) {
    return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
unsigned *ds_sched_slot(void *stream);          // bn_pack.hip
// a value every lane holds identically, as a scalar (tile indices read back from LDS)
__device__ __forceinline__ int ds_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// BatchNorm's normalise-and-shift of one element, z * scale + shift as ONE fused multiply-add: the train-mode forward
// (bn_apply_kernel) and the kernels that re-derive its clipped-ReLU mask from z in the backward pass (the BNB epilogue
// of conv_mfma_bf16_kernel) must round identically.
__device__ __forceinline__ float ds_bn_affine(float z, float scale, float shift) { return __builtin_fmaf(z, scale, shift); }

static inline int ds_last_launch_error() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
