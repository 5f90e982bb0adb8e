// TEST INFRASTRUCTURE -- host stand-in for <hip/hip_runtime.h>.
//
// Lets the unmodified kernel sources under deepspeaker-pytorch_amd/csrc/ be compiled with the
// host clang++ and executed lane-by-lane on a CPU: every GPU thread becomes a cooperative
// fiber, a workgroup is a set of fibers scheduled round-robin, wavefront collectives and
// __syncthreads() are rendezvous points (emu_runtime.cpp).  Used only by tests/ to debug
// indexing / fragment-layout logic without a GPU; it is never loaded by the product package.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <functional>

#define DS_EMULATED 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)

namespace emu {
struct Dim3 { unsigned x, y, z; };
struct Idx {
    static Dim3 &tid();
    static Dim3 &bid();
    static Dim3 &bdim();
    static Dim3 &gdim();
};
void syncthreads();
float *dynamic_lds();
void launch(int grid, int block, size_t lds_bytes, const std::function<void()> &body, bool big_lds = false);
// wavefront collectives (64 lanes)
void wave_exchange(float mine, float *all64);          // all64[l] = lane l's `mine`
}  // namespace emu

#define threadIdx (emu::Idx::tid())
#define blockIdx (emu::Idx::bid())
#define blockDim (emu::Idx::bdim())
#define gridDim (emu::Idx::gdim())

static inline void __syncthreads() { emu::syncthreads(); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
typedef void *hipStream_t;
// ---- the handful of HIP runtime calls the C-ABI translation units make besides launching: on the host there is one
// "device", nothing is ever being captured, and timing events do not exist (the product sources carry no emulator
// branches: round 6)
typedef int hipError_t;
typedef void *hipEvent_t;
enum : int { hipSuccess = 0, hipErrorNotReady = 600, hipErrorNotSupported = 801 };
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
static inline hipError_t hipGetDevice(int *dev) { *dev = 0; return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *) { return hipErrorNotSupported; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipErrorNotSupported; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipErrorNotSupported; }
static inline hipError_t hipEventElapsedTime(float *, hipEvent_t, hipEvent_t) { return hipErrorNotSupported; }
// scheduling hints are no-ops on the host
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
