#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;
// probe: LDS-DMA with some lanes out of range -> what lands in LDS?
__global__ void k(const int* src, int nbytes, int* out) {
    __shared__ __attribute__((aligned(16))) int lds[64 * 4 * 2];
    for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) lds[i] = -7;       // poison
    __syncthreads();
    auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    unsigned voff = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16;     // odd lanes out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)voff, 0, 0, 0);
    // second DMA with a scalar offset, into the second KiB
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 256), 16, (int)(threadIdx.x * 16), 1024, 0, 0);
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) out[i] = lds[i];
}
int main() {
    int *src, *out; hipMalloc(&src, 4096); hipMalloc(&out, 2048);
    int h[1024]; for (int i = 0; i < 1024; ++i) h[i] = i + 1;
    hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, 4096, out);
    int r[512]; hipMemcpy(r, out, 2048, hipMemcpyDeviceToHost);
    printf("lane0 chunk: %d %d %d %d | lane1 (OOB) chunk: %d %d %d %d | lane2: %d | lane3(OOB): %d\n", r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[12]);
    printf("second DMA (soffset 1024): lane0 %d lane1 %d lane63 %d\n", r[256], r[260], r[256 + 63 * 4]);
    int bad = 0; for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) { int want = (l & 1) ? 0 : l * 4 + e + 1; if (r[l * 4 + e] != want) ++bad; }
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 4; ++e) if (r[256 + l * 4 + e] != 256 + l * 4 + e + 1) ++bad;
    printf("mismatches vs (valid=data, OOB=0): %d\n", bad);
    return 0;
}
