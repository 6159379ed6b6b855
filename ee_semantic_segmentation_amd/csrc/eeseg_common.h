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

// ---- in-launch barrier among a GROUP of co-resident blocks (one episode per group and launch) ------------------
// For kernels whose whole grid is resident (the host sizes the grid to <= one block per CU it may count on); a group =
// the blocks that exchange data (all of them = a grid barrier; the row blocks of one channel group = the BatchNorm
// backward below: fewer arrivals per counter and no wait for unrelated blocks).
// Fence-free form of cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms" (first row of the sc1
// table): CONTRACT - every byte another block reads after the barrier was stored by eeseg_st_sc1 (write-through) and is
// loaded by eeseg_ld_sc1 (bypasses this CU's L1); every storing wave drains (s_waitcnt vmcnt(0)), the workgroup meets,
// ONE lane adds the block's arrival (agent-scope atomic) and polls the counter with sc1 loads, the workgroup meets again.
// No release / acquire fence: a release writes back the whole XCD L2 and, with the acquire, cost 13-17 us per barrier
// here against 3-6 us for this form (stamps, round 4).
// state (EESEG_BARRIER_WORDS unsigned words, 128-byte aligned, zeroed ONCE by the owner): group g's arrival counter is
// word 32*g, its departure counter word 32*(128+g) (lines of their own, g < 128), word 32*256 the sticky give-up word.
// The last block of a group to leave zeroes the group's two counters, so no memset node precedes the next launch.
// The spin is bounded: a grid that is not fully resident ends with the give-up word set and wrong results, not a hang.
#define EESEG_BARRIER_GROUPS 128
#define EESEG_BARRIER_WORDS (32 * 2 * EESEG_BARRIER_GROUPS + 32)
typedef __attribute__((address_space(1))) unsigned eeseg_gu32;
typedef __attribute__((address_space(1))) float eeseg_gf32;
__device__ __forceinline__ void eeseg_st_sc1(float* p, float v) {
    __hip_atomic_store((eeseg_gf32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float eeseg_ld_sc1(const float* p) {
    return __hip_atomic_load((eeseg_gf32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool eeseg_group_barrier(unsigned* state, unsigned group, unsigned members) {
    __shared__ int ok_s;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every wave: its sc1 stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        eeseg_gu32* arrive = (eeseg_gu32*)state + group * 32u;
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = 1;
        for (unsigned spins = 0; __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < members;) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) {                         // seconds: the grid was not co-resident
                __hip_atomic_store((eeseg_gu32*)state + 32u * 2u * EESEG_BARRIER_GROUPS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    const bool ok = ok_s != 0;
    if (threadIdx.x == 0) {
        eeseg_gu32* depart = (eeseg_gu32*)state + (EESEG_BARRIER_GROUPS + group) * 32u;
        const unsigned old = __hip_atomic_fetch_add(depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == members - 1) {                               // everybody has left the spin: reset for the next launch
            __hip_atomic_store((eeseg_gu32*)state + group * 32u, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    return ok;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
