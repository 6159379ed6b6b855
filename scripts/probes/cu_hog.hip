// A stand-in for a collective's persistent workgroups: `blocks` workgroups that hold their CU slots for `ms` milliseconds
// (RCCL kernels spin on flags from their peers; one workgroup per channel).  Only for scripts/contention_probe.py.
#include <hip/hip_runtime.h>

__global__ void hog_kernel(long long ticks) {
    extern __shared__ char hog_lds[];              // dynamic LDS only reserves space: > 24 KiB keeps a 256-tile conv block off the CU
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

extern "C" int hog_launch(int blocks, int threads, int lds_bytes, double ms, void* stream) {
    int khz = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
    const long long ticks = (long long)(ms * (double)khz);
    hipLaunchKernelGGL(hog_kernel, dim3(blocks), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream, ticks);
    return (int)hipGetLastError();
}
