// Shared device/host helpers for libeeseg (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/eeseg.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define EESEG_WAVE 64
#define EESEG_OOB 0x80000000u   // buffer voffset that is always out of range (num_records < 2^31)

// ---- error reporting (thread-local string) ----------------------
void eeseg_set_error(const char* fmt, ...);

#define EESEG_CHECK(cond, code, ...)            \
    do {                                        \
        if (!(cond)) {                          \
            eeseg_set_error(__VA_ARGS__);       \
            return (code);                      \
        }                                       \
    } while (0)

#define EESEG_HIP(call)                                                          \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            eeseg_set_error("%s failed: %s", #call, hipGetErrorString(e_));      \
            return EESEG_ERR_HIP;                                                \
        }                                                                        \
    } while (0)

#define EESEG_LAUNCH_CHECK()                                                     \
    do {                                                                         \
        hipError_t e_ = hipGetLastError();                                       \
        if (e_ != hipSuccess) {                                                  \
            eeseg_set_error("kernel launch failed: %s", hipGetErrorString(e_));  \
            return EESEG_ERR_HIP;                                                \
        }                                                                        \
    } while (0)

static inline int eeseg_dtype_size(int dt) { return dt == EESEG_BF16 ? 2 : (dt == EESEG_F32 ? 4 : 0); }

// ---- device helpers -------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// XCD-aware block remap (bijective for any grid size): consecutive logical ids
// land on one XCD so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
