// Shared helpers for the gfx950 kernels of libvision_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vision_amd.h"

typedef __bf16 bf16_t;

namespace vamd {

int fail(int code, const char* fmt, ...);

#define VAMD_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return ::vamd::fail(VAMD_EINVAL, __VA_ARGS__); \
    } while (0)

#define VAMD_LAUNCH_OK(name)                                                          \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess)                                                        \
            return ::vamd::fail((int)e__, "%s: launch failed: %s", name, hipGetErrorString(e__)); \
    } while (0)

// dtype dispatch: BODY sees the element type as `T`.
#define VAMD_DISPATCH(dtype, name, ...)                                   \
    do {                                                                  \
        if ((dtype) == VAMD_F32) {                                        \
            using T = float;                                              \
            __VA_ARGS__                                                   \
        } else if ((dtype) == VAMD_BF16) {                                \
            using T = bf16_t;                                             \
            __VA_ARGS__                                                   \
        } else {                                                          \
            return ::vamd::fail(VAMD_EINVAL, "%s: bad dtype %d", name, (int)(dtype)); \
        }                                                                 \
    } while (0)

// Launch timing (vamd_launch_timing): while switched on, a TimedLaunch on the stack of a launcher
// records one HIP event right before and one right after the kernel launch it brackets.
int timed_launch_begin(int kind, double flops, double executed, hipStream_t st);     // slot index or -1 when off
void timed_launch_end(int slot, hipStream_t st);
struct TimedLaunch {
    int slot;
    hipStream_t st;
    // flops: the operator's algorithmic count; executed: what the launch really multiplies (they differ for
    // the combined-filter forms of the upsampling convolution)
    TimedLaunch(int kind, double flops, double executed, hipStream_t s)
        : slot(timed_launch_begin(kind, flops, executed, s)), st(s) {}
    ~TimedLaunch() { if (slot >= 0) timed_launch_end(slot, st); }
};
enum { TIMED_NT = 0, TIMED_TN = 1 };

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }

static inline int grid_for(int64_t work_items, int block, int max_blocks = 4096) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

__device__ __forceinline__ float to_f(float x) { return x; }
__device__ __forceinline__ float to_f(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f(float x);
template <> __device__ __forceinline__ float from_f<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float x) { return (bf16_t)x; }

// 16-byte vector of T (4 floats or 8 bf16).
template <typename T> struct alignas(16) Vec16 {
    static constexpr int N = 16 / sizeof(T);
    T v[N];
};

__device__ __forceinline__ float leaky(float z, float slope) { return z > 0.f ? z : z * slope; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

}  // namespace vamd
