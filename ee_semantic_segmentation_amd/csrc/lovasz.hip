// Multi-class Lovasz loss on RAW scores for one exit (branchy_seg_losses.py:154 ->
// lovaszsoftmax.py:172-200; classes = 'present' | 'all' | list as a class mask + present_only flag; per_image=True is
// the host calling this once per image), fully on device:
//   1. prep : key[c][p] = |1[y_p=c] - s_pc| (valid pixels) or -1 (void: sorts last),
//             val = pixel index | foreground bit 30 | sign bit 31;  class pixel counts G[c], #valid pixels
//   2. sort : hand-written segmented LSD radix sort (4 passes of 8 bits, descending, one
//             segment per class): per-tile LDS histograms -> per-class scan of the
//             (digit, tile) counts -> stable scatter: wave-ballot digit matching, tile sorted
//             locally in LDS, written out as contiguous runs
//   3. scan : per class, blocked inclusive scan of the sorted foreground flags ->
//             Jaccard gradient J_k - J_{k-1} (lovasz_grad, lovaszsoftmax.py:19-31),
//             loss_c = sum e_k * grad_k, and d(loss)/d(score) scattered back through
//             the permutation.  Loss = mean over classes present in the labels.
// No host synchronisation: counts, #present classes and the upstream gradient scalar
// are read from device memory.
#include "eeseg_common.h"

namespace {

constexpr int SB = 2048;          // elements per scan block (256 threads x 8)

struct LvClassIds { short id[64]; };   // label class ranked as class c (identity unless the caller ranks a subset, eeseg_lovasz class_ids)

struct LvHeader {                 // lives at the start of the workspace
    int n_valid;
    int n_present;
    int pad[2];
};

// n_label >= C: labels in [0, n_label) are valid pixels; only labels < C are the foreground of a ranked class (a rank of a
// class-sharded data-parallel Lovasz ranks ITS classes, remapped to 0 .. C-1, over pixels of every class)
__global__ __launch_bounds__(256) void lv_prep(const float* __restrict__ scores, const int64_t* __restrict__ target,
                                               int N, int C, int HW, long long ignore, int n_label, LvClassIds ids, float* keys,
                                               unsigned* vals, int* G, LvHeader* hdr) {
    __shared__ int sG[64];
    __shared__ int sValid;
    for (int i = threadIdx.x; i < 64; i += blockDim.x) sG[i] = 0;
    if (threadIdx.x == 0) sValid = 0;
    __syncthreads();
    const long long P = (long long)N * HW;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const long long t = target[p];
        const bool valid = (t != ignore) && t >= 0 && t < n_label;
        const int n = (int)(p / HW);
        const int hw = (int)(p - (long long)n * HW);
        int mine = -1;                                      // the ranked class this pixel is foreground of, if any
        for (int c = 0; c < C; ++c) {
            float key = -1.f;
            unsigned v = (unsigned)p;                       // p < 2^30 (P*C < 2^31, C >= 2): bits 30/31 are free
            const bool fg = valid && t == (long long)ids.id[c];
            if (valid) {
                const float s = scores[((size_t)n * C + c) * HW + hw];
                const float d = (fg ? 1.f : 0.f) - s;
                key = fabsf(d);
                if (d < 0.f) v |= 0x80000000u;              // sign of (fg - s)
                if (fg) v |= 0x40000000u;                   // foreground flag travels with the element
            }
            if (fg) mine = c;
            keys[(size_t)c * P + p] = key;
            vals[(size_t)c * P + p] = v;
        }
        if (valid) {
            if (mine >= 0) atomicAdd(&sG[mine], 1);
            atomicAdd(&sValid, 1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x)
        if (sG[i]) atomicAdd(&G[i], sG[i]);
    if (threadIdx.x == 0 && sValid) atomicAdd(&hdr->n_valid, sValid);
}

// ------------------------------------------------------------------ radix sort ----
// Keys are floats >= 0 or -1; descending float order == ascending order of
// ukey = ~(sign-flipped bits).  One pass sorts by 8 bits, least significant first.
constexpr int RT = 8192;            // elements per sort tile (4 waves x 32 rows x 64)
constexpr int RS_LDS = (2 * RT + 4 * 256 + 256 + 256 + 4) * 4;   // rs_scatter dynamic LDS bytes

__device__ __forceinline__ unsigned sort_key(float f) {
    unsigned u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // ascending unsigned == ascending float
    return ~u;                                          // descending float
}

// hist[c][digit][tile] = number of elements of tile with that digit
__global__ __launch_bounds__(256) void rs_hist(const float* __restrict__ keys, long long P, int ntile, int shift,
                                               unsigned* __restrict__ hist) {
    __shared__ unsigned sh[256];
    const int c = blockIdx.y, t = blockIdx.x;
    sh[threadIdx.x] = 0u;
    __syncthreads();
    const long long base = (long long)t * RT;
    for (int i = threadIdx.x; i < RT; i += 256) {
        const long long k = base + i;
        if (k < P) atomicAdd(&sh[(sort_key(keys[(size_t)c * P + k]) >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[((size_t)c * 256 + threadIdx.x) * ntile + t] = sh[threadIdx.x];
}

// exclusive scan of n consecutive entries per blockIdx.x, in place; optionally stores the total
__global__ __launch_bounds__(1024) void rs_scan(unsigned* hist, int n, unsigned* totals = nullptr) {
    __shared__ unsigned swave[16];
    __shared__ unsigned carry;
    unsigned* h = hist + (size_t)blockIdx.x * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    for (int b = 0; b < n; b += 1024) {
        const int i = b + threadIdx.x;
        const unsigned v = (i < n) ? h[i] : 0u;
        unsigned incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        if (lane == 63) swave[wave] = incl;
        __syncthreads();
        unsigned off = carry;
        for (int w = 0; w < wave; ++w) off += swave[w];
        if (i < n) h[i] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = off + incl;
        __syncthreads();
    }
    if (totals && threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// Stable scatter of one tile (RT elements) of one class.  Wave w owns the contiguous quarter [w*RT/4, (w+1)*RT/4) of the
// tile, 64 consecutive elements per row:
//   A. per wave, row by row: lanes holding the same digit find each other with 8 ballots; rank inside the wave =
//      (per-wave digit counter so far) + (same-digit lanes below me); the row's first lane of each digit bumps the counter
//      (LDS operations of one wave execute in order: no barrier inside the loop);
//   B. per digit: exclusive prefix of the wave counters, tile-local start of the digit (block scan), global start from
//      the scanned histogram;
//   C. every element goes to its tile-local sorted position in LDS;
//   D. the tile is written out in sorted order: a digit's elements are consecutive in LDS AND in global memory, so the
//      stores are contiguous runs (RT/256 = 32 elements = 128 B on average) instead of one 4-byte transaction each.
__global__ __launch_bounds__(256) void rs_scatter(const float* __restrict__ kin, const unsigned* __restrict__ vin,
                                                  float* __restrict__ kout, unsigned* __restrict__ vout, long long P,
                                                  int ntile, int shift, const unsigned* __restrict__ hist,
                                                  const unsigned* __restrict__ dtot) {
    constexpr int ROWS = RT / 256;              // rows of 64 elements per wave
    __shared__ unsigned smem_rs[RS_LDS / 4];
    unsigned* sk = smem_rs;                     // [RT] staged keys (as bits)
    unsigned* sv = smem_rs + RT;                // [RT] staged values
    unsigned* cntw = smem_rs + 2 * RT;          // [4][256] per-wave digit counters -> exclusive wave prefixes
    unsigned* lstart = cntw + 4 * 256;          // [256] tile-local start of each digit
    unsigned* gbase = lstart + 256;             // [256] global start of each digit for this tile
    unsigned* swsum = gbase + 256;              // [4] block-scan scratch
    const int c = blockIdx.y, t = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 256; i += 256) cntw[i] = 0u;
    __syncthreads();
    const long long base = (long long)t * RT + (long long)wave * (RT / 4);
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    unsigned key[ROWS], val[ROWS];
    unsigned short rank[ROWS];
    unsigned* myc = cntw + wave * 256;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const long long k = base + r * 64 + lane;
        const bool ok = k < P;
        unsigned dg = 0u;
        key[r] = 0u; val[r] = 0u;
        if (ok) {
            key[r] = __float_as_uint(kin[(size_t)c * P + k]);
            val[r] = vin[(size_t)c * P + k];
            dg = (sort_key(__uint_as_float(key[r])) >> shift) & 255u;
        }
        unsigned long long same = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((dg >> b) & 1u);
            same &= ((dg >> b) & 1u) ? m : ~m;
        }
        const unsigned in_row = (unsigned)__popcll(same & below);
        const unsigned prior = ok ? myc[dg] : 0u;
        rank[r] = (unsigned short)(prior + in_row);
        if (ok && in_row == 0) myc[dg] = prior + (unsigned)__popcll(same);
    }
    __syncthreads();
    {   // B: thread d = digit
        const int d = threadIdx.x;
        const unsigned c0 = cntw[d], c1 = cntw[256 + d], c2 = cntw[512 + d], c3 = cntw[768 + d];
        cntw[d] = 0u; cntw[256 + d] = c0; cntw[512 + d] = c0 + c1; cntw[768 + d] = c0 + c1 + c2;
        const unsigned tot = c0 + c1 + c2 + c3;
        unsigned incl = tot;                    // exclusive scan of tot over the 256 digits
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned u = __shfl_up(incl, o);
            if (lane >= o) incl += u;
        }
        if (lane == 63) swsum[wave] = incl;
        __syncthreads();
        unsigned off = 0u;
        for (int w = 0; w < wave; ++w) off += swsum[w];
        lstart[d] = off + incl - tot;
        gbase[d] = dtot[c * 256 + d] + hist[((size_t)c * 256 + d) * ntile + t];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {            // C
        const long long k = base + r * 64 + lane;
        if (k < P) {
            const unsigned dg = (sort_key(__uint_as_float(key[r])) >> shift) & 255u;
            const unsigned lp = lstart[dg] + myc[dg] + rank[r];
            sk[lp] = key[r];
            sv[lp] = val[r];
        }
    }
    __syncthreads();
    const long long tile_first = (long long)t * RT;
    const int n_here = (int)((P - tile_first < RT) ? (P - tile_first) : RT);
    for (int i = threadIdx.x; i < n_here; i += 256) {          // D
        const unsigned kb = sk[i];
        const unsigned dg = (sort_key(__uint_as_float(kb)) >> shift) & 255u;
        const size_t o = (size_t)c * P + gbase[dg] + ((unsigned)i - lstart[dg]);
        kout[o] = __uint_as_float(kb);
        vout[o] = sv[i];
    }
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
        if (k < nv) cnt += (int)((vals[(size_t)c * P + k] >> 30) & 1u);
    }
    cnt = (int)wave_sum((float)cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&s, cnt);
    __syncthreads();
    if (threadIdx.x == 0) bsum[(size_t)c * nblk + b] = s;
}

// class c takes part in the mean: selected by the caller's mask and, for classes='present', present in the labels
// (lovaszsoftmax.py:185-188: 'all' and explicit lists keep absent classes - an absent class contributes max |score|)
__device__ __forceinline__ bool lv_counted(int c, int g, unsigned long long class_mask, int present_only) {
    return ((class_mask >> c) & 1ull) && (!present_only || g > 0);
}

// number of classes in the mean (the block counts are scanned by rs_scan, one block per class)
__global__ void lv_count_present(int C, const int* G, LvHeader* hdr, unsigned long long class_mask, int present_only) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C && lv_counted(c, G[c], class_mask, present_only)) atomicAdd(&hdr->n_present, 1);
}

__device__ __forceinline__ float jaccard(float G, float F, float kp1) {   // 1 - (G-F)/(G + k+1 - F)
    return 1.f - (G - F) / (G + kp1 - F);
}

__global__ __launch_bounds__(256) void lv_final(const float* __restrict__ keys, const unsigned* __restrict__ vals,
                                                const int64_t* __restrict__ target, long long P, int nblk, int C,
                                                int HW, const int* __restrict__ G, const int* __restrict__ bsum,
                                                const LvHeader* hdr, double* class_loss, float* dscores, float gscale,
                                                const float* gscale_dev, unsigned long long class_mask, int present_only,
                                                const int* norm_dev) {
    __shared__ int swave[4];
    __shared__ double sloss[4];
    const int c = blockIdx.y, b = blockIdx.x;
    const int nv = hdr->n_valid;
    const int g = G[c];
    if (!lv_counted(c, g, class_mask, present_only) || (long long)b * SB >= nv) return;   // class not in the mean / block past the valid range
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
            fg[i] = (int)((v[i] >> 30) & 1u);
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
                            ? gscale * (gscale_dev ? gscale_dev[0] : 1.f) / (float)max(norm_dev ? norm_dev[0] : hdr->n_present, 1) : 0.f;
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
            const unsigned px = v[i] & 0x3FFFFFFFu;
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

// norm_dev (optional): the number of classes the mean runs over, from the caller (class-sharded data parallelism: this call
// ranks a SUBSET of the classes and returns its share sum_c loss_c / norm - the shares of the ranks add up to the loss)
__global__ void lv_loss(const double* class_loss, const int* G, int C, const LvHeader* hdr, float* loss_out,
                        unsigned long long class_mask, int present_only, const int* norm_dev) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        int n = 0;
        for (int c = 0; c < C; ++c)
            if (lv_counted(c, G[c], class_mask, present_only)) { s += class_loss[c]; ++n; }
        if (norm_dev) n = n ? norm_dev[0] : 0;
        loss_out[0] = (n && hdr->n_valid > 0) ? (float)(s / n) : 0.f;   // only void pixels: 0 (lovaszsoftmax.py:181-183)
    }
}

// counts[c] = number of pixels labelled c (c < C; `ignore` and out-of-range labels skipped): which classes are present
__global__ __launch_bounds__(256) void label_hist_kernel(const int64_t* __restrict__ target, long long P, int C, long long ignore,
                                                         int* counts) {
    __shared__ int sh[64];
    for (int i = threadIdx.x; i < 64; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const long long t = target[p];
        if (t != ignore && t >= 0 && t < C) atomicAdd(&sh[(int)t], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += blockDim.x)
        if (sh[i]) atomicAdd(&counts[i], sh[i]);
}

struct Layout {
    size_t hdr, G, closs, bsum, hist, dtot, keys_in, keys_out, vals_in, vals_out, total;
};
inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }
Layout layout(long long P, int C) {
    const int nblk = (int)((P + SB - 1) / SB);
    const int ntile = (int)((P + RT - 1) / RT);
    Layout L;
    size_t o = 0;
    L.hdr = o; o += 256;
    L.G = o; o += align256((size_t)C * 4);
    L.closs = o; o += align256((size_t)C * 8);
    L.bsum = o; o += align256((size_t)C * nblk * 4);
    L.hist = o; o += align256((size_t)C * 256 * ntile * 4);
    L.dtot = o; o += align256((size_t)C * 256 * 4);
    const size_t arr = align256((size_t)C * P * 4);
    L.keys_in = o; o += arr;
    L.keys_out = o; o += arr;
    L.vals_in = o; o += arr;
    L.vals_out = o; o += arr;
    L.total = o;
    return L;
}

}  // namespace

extern "C" int64_t eeseg_lovasz_workspace(int64_t P, int C) {
    if (P <= 0 || C <= 0 || (long long)P * C >= (1ll << 31)) return -1;
    return (int64_t)layout(P, C).total;
}

extern "C" int eeseg_label_hist(const int64_t* target, int64_t n, int C, int64_t ignore_index, int32_t* counts, void* stream) {
    EESEG_CHECK(target && counts && n > 0 && C > 0 && C <= 64, EESEG_ERR_ARG, "label_hist: bad argument (C <= 64)");
    hipStream_t st = (hipStream_t)stream;
    EESEG_HIP(hipMemsetAsync(counts, 0, (size_t)C * sizeof(int32_t), st));
    long long pb = (n + 255) / 256;
    if (pb > 2048) pb = 2048;
    hipLaunchKernelGGL(label_hist_kernel, dim3((unsigned)pb), dim3(256), 0, st, target, (long long)n, C, (long long)ignore_index, counts);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_lovasz(const float* scores, const int64_t* target, int N, int C, int HW, int64_t ignore_index,
                            float* loss_out, float* dscores, float gscale, const float* gscale_dev, uint64_t class_mask,
                            int present_only, int n_label_classes, const int32_t* class_ids, const int32_t* norm_classes_dev,
                            void* workspace, int64_t workspace_bytes, void* stream) {
    EESEG_CHECK(scores && target && loss_out && workspace, EESEG_ERR_ARG, "lovasz: null pointer");
    if (n_label_classes <= 0) n_label_classes = C;
    EESEG_CHECK(n_label_classes >= C && n_label_classes <= 32767, EESEG_ERR_ARG, "lovasz: n_label_classes must be in [C, 32767]");
    EESEG_CHECK(C <= 64, EESEG_ERR_ARG, "lovasz: C <= 64");
    LvClassIds ids;
    for (int c = 0; c < 64; ++c) ids.id[c] = (short)c;
    if (class_ids) {
        for (int c = 0; c < C; ++c) {
            EESEG_CHECK(class_ids[c] >= 0 && class_ids[c] < n_label_classes, EESEG_ERR_ARG, "lovasz: class_ids[%d] = %d outside [0, %d)",
                        c, class_ids[c], n_label_classes);
            ids.id[c] = (short)class_ids[c];
        }
    }
    EESEG_CHECK(present_only == 0 || present_only == 1, EESEG_ERR_ARG, "lovasz: present_only must be 0 or 1");
    EESEG_CHECK(N > 0 && C > 0 && C <= 64 && HW > 0, EESEG_ERR_ARG, "lovasz: bad shape (C <= 64)");
    const long long P = (long long)N * HW;
    EESEG_CHECK(P * C < (1ll << 31), EESEG_ERR_TOO_LARGE, "lovasz: N*HW*C must be < 2^31");
    const Layout L = layout(P, C);
    EESEG_CHECK(workspace_bytes >= (int64_t)L.total, EESEG_ERR_ARG, "lovasz: workspace too small (%zu needed)", L.total);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)workspace;
    LvHeader* hdr = (LvHeader*)(w + L.hdr);
    int* G = (int*)(w + L.G);
    double* closs = (double*)(w + L.closs);
    unsigned* hist = (unsigned*)(w + L.hist);
    unsigned* dtot = (unsigned*)(w + L.dtot);
    int* bsum = (int*)(w + L.bsum);
    float* keys_in = (float*)(w + L.keys_in);
    float* keys_out = (float*)(w + L.keys_out);
    unsigned* vals_in = (unsigned*)(w + L.vals_in);
    unsigned* vals_out = (unsigned*)(w + L.vals_out);
    const int nblk = (int)((P + SB - 1) / SB);

    EESEG_HIP(hipMemsetAsync(w, 0, L.bsum, st));                       // header, G, class losses
    if (dscores) EESEG_HIP(hipMemsetAsync(dscores, 0, (size_t)P * C * sizeof(float), st));
    long long pb = (P + 255) / 256;
    if (pb > 4096) pb = 4096;
    hipLaunchKernelGGL(lv_prep, dim3((unsigned)pb), dim3(256), 0, st, scores, target, N, C, HW, (long long)ignore_index,
                       n_label_classes, ids, keys_in, vals_in, G, hdr);
    EESEG_LAUNCH_CHECK();
    // 4 x 8-bit LSD passes, ping-pong between the two buffer pairs
    const int ntile = (int)((P + RT - 1) / RT);
    float* ka = keys_in; float* kb = keys_out;
    unsigned* va = vals_in; unsigned* vb = vals_out;
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(rs_hist, dim3(ntile, C), dim3(256), 0, st, (const float*)ka, P, ntile, pass * 8, hist);
        // two-level scan: (class, digit) blocks scan their tile counts and emit the digit totals; one block per class
        // scans the 256 totals; the scatter adds the two
        hipLaunchKernelGGL(rs_scan, dim3(C * 256), dim3(1024), 0, st, hist, ntile, dtot);
        hipLaunchKernelGGL(rs_scan, dim3(C), dim3(1024), 0, st, dtot, 256, (unsigned*)nullptr);
        hipLaunchKernelGGL(rs_scatter, dim3(ntile, C), dim3(256), 0, st, (const float*)ka, (const unsigned*)va, kb, vb, P,
                           ntile, pass * 8, (const unsigned*)hist, (const unsigned*)dtot);
        float* tk = ka; ka = kb; kb = tk;
        unsigned* tv = va; va = vb; vb = tv;
    }
    EESEG_LAUNCH_CHECK();
    keys_out = ka;                 // after an even number of passes the sorted data is back in the first pair
    vals_out = va;
    hipLaunchKernelGGL(lv_block_counts, dim3(nblk, C), dim3(256), 0, st, vals_out, target, P, nblk, hdr, bsum);
    hipLaunchKernelGGL(rs_scan, dim3(C), dim3(1024), 0, st, (unsigned*)bsum, nblk, (unsigned*)nullptr);
    const unsigned long long cm = (unsigned long long)class_mask;
    hipLaunchKernelGGL(lv_count_present, dim3(1), dim3(64), 0, st, C, G, hdr, cm, present_only);
    hipLaunchKernelGGL(lv_final, dim3(nblk, C), dim3(256), 0, st, keys_out, vals_out, target, P, nblk, C, HW, G, bsum, hdr,
                       closs, dscores, gscale, gscale_dev, cm, present_only, norm_classes_dev);
    hipLaunchKernelGGL(lv_loss, dim3(1), dim3(64), 0, st, closs, G, C, hdr, loss_out, cm, present_only, norm_classes_dev);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}
