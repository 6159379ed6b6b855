// Multi-class Lovasz loss on RAW scores for one exit (branchy_seg_losses.py:154 ->
// lovaszsoftmax.py:172-200, per_image=False, classes='present'), fully on device:
//   1. prep : key[c][p] = |1[y_p=c] - s_pc| (valid pixels) or -1 (void: sorts last),
//             val = pixel index | sign bit;  class pixel counts G[c], #valid pixels
//   2. sort : rocPRIM segmented radix sort, descending, one segment per class
//             (the sort is the only non-hand-written device code in libeeseg; a
//             hand-written LDS radix sort is the planned replacement)
//   3. scan : per class, blocked inclusive scan of the sorted foreground flags ->
//             Jaccard gradient J_k - J_{k-1} (lovasz_grad, lovaszsoftmax.py:19-31),
//             loss_c = sum e_k * grad_k, and d(loss)/d(score) scattered back through
//             the permutation.  Loss = mean over classes present in the labels.
// No host synchronisation: counts, #present classes and the upstream gradient scalar
// are read from device memory.
#include <cstring>
#include <string.h>

#include "eeseg_common.h"

#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace {

constexpr int SB = 2048;          // elements per scan block (256 threads x 8)

struct LvHeader {                 // lives at the start of the workspace
    int n_valid;
    int n_present;
    int pad[2];
};

__global__ __launch_bounds__(256) void lv_prep(const float* __restrict__ scores, const int64_t* __restrict__ target,
                                               int N, int C, int HW, long long ignore, float* keys, unsigned* vals,
                                               int* G, LvHeader* hdr) {
    __shared__ int sG[64];
    __shared__ int sValid;
    for (int i = threadIdx.x; i < 64; i += blockDim.x) sG[i] = 0;
    if (threadIdx.x == 0) sValid = 0;
    __syncthreads();
    const long long P = (long long)N * HW;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const long long t = target[p];
        const bool valid = (t != ignore) && t >= 0 && t < C;
        const int n = (int)(p / HW);
        const int hw = (int)(p - (long long)n * HW);
        for (int c = 0; c < C; ++c) {
            float key = -1.f;
            unsigned v = (unsigned)p;
            if (valid) {
                const float s = scores[((size_t)n * C + c) * HW + hw];
                const float d = (t == c ? 1.f : 0.f) - s;
                key = fabsf(d);
                if (d < 0.f) v |= 0x80000000u;
            }
            keys[(size_t)c * P + p] = key;
            vals[(size_t)c * P + p] = v;
        }
        if (valid) {
            atomicAdd(&sG[(int)t], 1);
            atomicAdd(&sValid, 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x)
        if (sG[i]) atomicAdd(&G[i], sG[i]);
    if (threadIdx.x == 0 && sValid) atomicAdd(&hdr->n_valid, sValid);
}

__global__ void lv_offsets(unsigned* offs, int C, long long P) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= C) offs[i] = (unsigned)((long long)i * P);
}

// per (class, block): number of foreground elements among the first n_valid sorted entries
__global__ __launch_bounds__(256) void lv_block_counts(const unsigned* __restrict__ vals, const int64_t* __restrict__ target,
                                                       long long P, int nblk, const LvHeader* hdr, int* bsum) {
    __shared__ int s;
    if (threadIdx.x == 0) s = 0;
    __syncthreads();
    const int c = blockIdx.y, b = blockIdx.x;
    const int nv = hdr->n_valid;
    int cnt = 0;
    for (int i = threadIdx.x; i < SB; i += 256) {
        const long long k = (long long)b * SB + i;
        if (k < nv) cnt += (target[vals[(size_t)c * P + k] & 0x7FFFFFFFu] == c) ? 1 : 0;
    }
    cnt = (int)wave_sum((float)cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s, cnt);
    __syncthreads();
    if (threadIdx.x == 0) bsum[(size_t)c * nblk + b] = s;
}

// exclusive scan of the block counts per class (in place) + number of present classes
__global__ void lv_scan_blocks(int* bsum, int nblk, int C, const int* G, LvHeader* hdr) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        int run = 0;
        for (int b = 0; b < nblk; ++b) {
            const int v = bsum[(size_t)c * nblk + b];
            bsum[(size_t)c * nblk + b] = run;
            run += v;
        }
        if (G[c] > 0) atomicAdd(&hdr->n_present, 1);
    }
}

__device__ __forceinline__ float jaccard(float G, float F, float kp1) {   // 1 - (G-F)/(G + k+1 - F)
    return 1.f - (G - F) / (G + kp1 - F);
}

__global__ __launch_bounds__(256) void lv_final(const float* __restrict__ keys, const unsigned* __restrict__ vals,
                                                const int64_t* __restrict__ target, long long P, int nblk, int C,
                                                int HW, const int* __restrict__ G, const int* __restrict__ bsum,
                                                const LvHeader* hdr, double* class_loss, float* dscores, float gscale,
                                                const float* gscale_dev) {
    __shared__ int swave[4];
    __shared__ double sloss[4];
    const int c = blockIdx.y, b = blockIdx.x;
    const int nv = hdr->n_valid;
    const int g = G[c];
    if (g == 0 || (long long)b * SB >= nv) return;       // class absent / block past the valid range
    const float Gf = (float)g;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // each thread owns 8 consecutive elements
    const long long k0 = (long long)b * SB + threadIdx.x * 8;
    unsigned v[8];
    float e[8];
    int fg[8], local = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long long k = k0 + i;
        if (k < nv) {
            v[i] = vals[(size_t)c * P + k];
            e[i] = keys[(size_t)c * P + k];
            fg[i] = (target[v[i] & 0x7FFFFFFFu] == c) ? 1 : 0;
        } else {
            v[i] = 0; e[i] = 0.f; fg[i] = 0;
        }
        local += fg[i];
    }
    // exclusive prefix of `local` over the block
    int incl = local;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
    }
    if (lane == 63) swave[wave] = incl;
    __syncthreads();
    int base = bsum[(size_t)c * nblk + b];
    for (int w = 0; w < wave; ++w) base += swave[w];
    int F = base + incl - local;                           // foreground count before my first element
    const float scale = (dscores != nullptr)
                            ? gscale * (gscale_dev ? gscale_dev[0] : 1.f) / (float)max(hdr->n_present, 1) : 0.f;
    double part = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const long long k = k0 + i;
        if (k >= nv) break;
        const float jprev = (k == 0) ? 0.f : jaccard(Gf, (float)F, (float)k);        // J_{k-1}: k elements, F fg
        F += fg[i];
        const float jk = jaccard(Gf, (float)F, (float)(k + 1));
        const float gr = jk - jprev;
        part += (double)e[i] * (double)gr;
        if (dscores != nullptr) {
            const unsigned px = v[i] & 0x7FFFFFFFu;
            const int n = (int)(px / (unsigned)HW);
            const int hw = (int)(px - (unsigned)n * (unsigned)HW);
            // e = |fg - s|  ->  de/ds = -sign(fg - s); sign(0) = 0 like torch.abs
            float sg = 0.f;
            if (e[i] > 0.f) sg = (v[i] & 0x80000000u) ? 1.f : -1.f;
            dscores[((size_t)n * C + c) * HW + hw] = sg * gr * scale;
        }
    }
    part = wave_sum_d(part);
    if (lane == 0) sloss[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&class_loss[c], sloss[0] + sloss[1] + sloss[2] + sloss[3]);
}

__global__ void lv_loss(const double* class_loss, const int* G, int C, const LvHeader* hdr, float* loss_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        int n = 0;
        for (int c = 0; c < C; ++c)
            if (G[c] > 0) { s += class_loss[c]; ++n; }
        loss_out[0] = n ? (float)(s / n) : 0.f;           // only void pixels: 0 (lovaszsoftmax.py:181-183)
    }
}

struct Layout {
    size_t hdr, G, closs, offs, bsum, keys_in, keys_out, vals_in, vals_out, temp, total;
};
inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }
Layout layout(long long P, int C, size_t temp_bytes) {
    const int nblk = (int)((P + SB - 1) / SB);
    Layout L;
    size_t o = 0;
    L.hdr = o; o += 256;
    L.G = o; o += align256((size_t)C * 4);
    L.closs = o; o += align256((size_t)C * 8);
    L.offs = o; o += align256((size_t)(C + 1) * 4);
    L.bsum = o; o += align256((size_t)C * nblk * 4);
    const size_t arr = align256((size_t)C * P * 4);
    L.keys_in = o; o += arr;
    L.keys_out = o; o += arr;
    L.vals_in = o; o += arr;
    L.vals_out = o; o += arr;
    L.temp = o; o += align256(temp_bytes);
    L.total = o;
    return L;
}

size_t sort_temp_bytes(long long P, int C) {
    size_t bytes = 0;
    (void)rocprim::segmented_radix_sort_pairs_desc((void*)nullptr, bytes, (const float*)nullptr, (float*)nullptr,
                                             (const unsigned*)nullptr, (unsigned*)nullptr, (unsigned)(C * P), (unsigned)C,
                                             (const unsigned*)nullptr, (const unsigned*)nullptr, 0, 32,
                                             (hipStream_t)0, false);
    return bytes;
}

}  // namespace

extern "C" int64_t eeseg_lovasz_workspace(int64_t P, int C) {
    if (P <= 0 || C <= 0 || (long long)P * C >= (1ll << 31)) return -1;
    return (int64_t)layout(P, C, sort_temp_bytes(P, C)).total;
}

extern "C" int eeseg_lovasz(const float* scores, const int64_t* target, int N, int C, int HW, int64_t ignore_index,
                            float* loss_out, float* dscores, float gscale, const float* gscale_dev, void* workspace,
                            int64_t workspace_bytes, void* stream) {
    EESEG_CHECK(scores && target && loss_out && workspace, EESEG_ERR_ARG, "lovasz: null pointer");
    EESEG_CHECK(N > 0 && C > 0 && C <= 64 && HW > 0, EESEG_ERR_ARG, "lovasz: bad shape (C <= 64)");
    const long long P = (long long)N * HW;
    EESEG_CHECK(P * C < (1ll << 31), EESEG_ERR_TOO_LARGE, "lovasz: N*HW*C must be < 2^31");
    size_t temp = sort_temp_bytes(P, C);
    const Layout L = layout(P, C, temp);
    EESEG_CHECK(workspace_bytes >= (int64_t)L.total, EESEG_ERR_ARG, "lovasz: workspace too small (%zu needed)", L.total);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)workspace;
    LvHeader* hdr = (LvHeader*)(w + L.hdr);
    int* G = (int*)(w + L.G);
    double* closs = (double*)(w + L.closs);
    unsigned* offs = (unsigned*)(w + L.offs);
    int* bsum = (int*)(w + L.bsum);
    float* keys_in = (float*)(w + L.keys_in);
    float* keys_out = (float*)(w + L.keys_out);
    unsigned* vals_in = (unsigned*)(w + L.vals_in);
    unsigned* vals_out = (unsigned*)(w + L.vals_out);
    const int nblk = (int)((P + SB - 1) / SB);

    EESEG_HIP(hipMemsetAsync(w, 0, L.offs, st));                       // header, G, class losses
    if (dscores) EESEG_HIP(hipMemsetAsync(dscores, 0, (size_t)P * C * sizeof(float), st));
    long long pb = (P + 255) / 256;
    if (pb > 4096) pb = 4096;
    hipLaunchKernelGGL(lv_prep, dim3((unsigned)pb), dim3(256), 0, st, scores, target, N, C, HW, (long long)ignore_index,
                       keys_in, vals_in, G, hdr);
    hipLaunchKernelGGL(lv_offsets, dim3(1), dim3(128), 0, st, offs, C, P);
    EESEG_LAUNCH_CHECK();
    hipError_t e = rocprim::segmented_radix_sort_pairs_desc((void*)(w + L.temp), temp, keys_in, keys_out, vals_in, vals_out,
                                                            (unsigned)(C * P), (unsigned)C, offs, offs + 1, 0, 32, st,
                                                            false);
    EESEG_CHECK(e == hipSuccess, EESEG_ERR_HIP, "lovasz: segmented sort failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(lv_block_counts, dim3(nblk, C), dim3(256), 0, st, vals_out, target, P, nblk, hdr, bsum);
    hipLaunchKernelGGL(lv_scan_blocks, dim3(1), dim3(64), 0, st, bsum, nblk, C, G, hdr);
    hipLaunchKernelGGL(lv_final, dim3(nblk, C), dim3(256), 0, st, keys_out, vals_out, target, P, nblk, C, HW, G, bsum, hdr,
                       closs, dscores, gscale, gscale_dev);
    hipLaunchKernelGGL(lv_loss, dim3(1), dim3(64), 0, st, closs, G, C, hdr, loss_out);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}
