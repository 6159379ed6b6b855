// HBM-bound kernels around the conv GEMMs: BatchNorm (finalize / apply / backward),
// column reductions (wave + LDS, deterministic two-stage), pooling, packing.
// Everything moves 16 bytes per lane; activations are NHWC rows [rows][C] with an
// explicit row stride (ld, in elements) so channel slices of a wider tensor (the
// ASPP concat buffer) are read/written in place.
#include "eeseg_common.h"

int g_bn_reverse = 0;               // EESEG_OPT_BN_REVERSE: bit 0 bn_apply, bit 1 bn_bwd_apply sweep rows from the end

int g_bn_nt = 0;                     // EESEG_OPT_BN_NT: bit 0 bn_apply, bit 1 bn_bwd_apply read their dead-after-this-pass inputs nontemporally
int g_colreduce_blocks = 512;        // EESEG_OPT_COLREDUCE_BLOCKS: blocks a column reduction aims at in all (0 = up to 1024 row blocks per
                                     // column block: 8192 blocks and 16 MB of partial sums for a 2048-channel tensor; 512: +0.5..1.4 % end to end)
int g_bn_rows = 2;                  // EESEG_OPT_BN_ROWS: rows of loads in flight per thread in bn_apply (1, 2, 4)
int g_bn_bwd_rows = 1;              // EESEG_OPT_BN_BWD_ROWS: the same for bn_bwd_apply / scale_act_bwd (three streams per row already: one row in
                                    // flight measured 10-14 % faster than two at 32 x 65 x 65, 3-10 % at 4-16 x 65 x 65; scripts/bn_knob_sweep.py)

namespace {

template <typename T> struct Vec {
    static constexpr int N = 16 / (int)sizeof(T);
    union { i32x4 q; T e[16 / sizeof(T)]; };
};

template <typename T> __device__ __forceinline__ Vec<T> ld16(const T* p) {
    Vec<T> v; v.q = *reinterpret_cast<const i32x4*>(p); return v;
}
// streamed-once loads: a nontemporal hint keeps a tensor that is dead after this pass from displacing the one the next
// kernel is about to read (EESEG_OPT_BN_NT)
template <typename T> __device__ __forceinline__ Vec<T> ld16s(const T* p, bool nt) {
    Vec<T> v;
    if (nt) v.q = __builtin_nontemporal_load(reinterpret_cast<const i32x4*>(p));
    else v.q = *reinterpret_cast<const i32x4*>(p);
    return v;
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec<T>& v) {
    *reinterpret_cast<i32x4*>(p) = v.q;
}

inline int ew_grid(long long work_items, int block = 256) {
    long long b = (work_items + block - 1) / block;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (int)b;
}

// =========================================================================
// pack / cast / im2col
// =========================================================================
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ src, T* wf, T* wb, int Cout, int Cout_pad, int Cin,
                                   int taps, int src_krsc) {
    const long long total = (long long)Cout_pad * taps * Cin;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Cin);
        const long long r = i / Cin;
        const int tap = (int)(r % taps);
        const int co = (int)(r / taps);
        float v = 0.f;
        if (co < Cout) v = src_krsc ? src[((long long)co * taps + tap) * Cin + ci]
                                    : src[((long long)co * Cin + ci) * taps + tap];
        if (wf) wf[i] = from_f32<T>(v);
        if (wb) wb[((long long)ci * taps + tap) * Cout_pad + co] = from_f32<T>(v);
    }
}

// one launch packs every conv weight of the network: desc[t] = {src, w_fwd, w_bwd, Cout, Cout_pad, Cin, taps, krsc}
struct PackDesc { const float* src; void* wf; void* wb; int Cout, Cout_pad, Cin, taps, krsc, pad_; };
// KRSC sources go through a 32x32 LDS-tiled transpose so both outputs are written in
// contiguous runs (the naive scattered 2-byte stores of w_bwd cost 8x write amplification).
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(const PackDesc* __restrict__ desc) {
    __shared__ float tile[32][33];
    const PackDesc d = desc[blockIdx.y];
    T* wf = (T*)d.wf;
    T* wb = (T*)d.wb;
    if (!d.krsc) {       // torch-default KCRS source: plain element-wise path
        const long long total = (long long)d.Cout_pad * d.taps * d.Cin;
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += gridDim.x * 256ll) {
            const int ci = (int)(i % d.Cin);
            const long long r = i / d.Cin;
            const int tap = (int)(r % d.taps);
            const int co = (int)(r / d.taps);
            const float v = (co < d.Cout) ? d.src[((long long)co * d.Cin + ci) * d.taps + tap] : 0.f;
            if (wf) wf[i] = from_f32<T>(v);
            if (wb) wb[((long long)ci * d.taps + tap) * d.Cout_pad + co] = from_f32<T>(v);
        }
        return;
    }
    const int tco = (d.Cout_pad + 31) / 32, tci = (d.Cin + 31) / 32;
    const int ntiles = tco * tci * d.taps;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tap = t % d.taps;
        const int tc = (t / d.taps) % tci;
        const int to = t / (d.taps * tci);
        const int ci = tc * 32 + tx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = to * 32 + ty + 8 * k;
            float v = 0.f;
            if (co < d.Cout && ci < d.Cin) v = d.src[((long long)co * d.taps + tap) * d.Cin + ci];
            tile[ty + 8 * k][tx] = v;
            if (wf && co < d.Cout_pad && ci < d.Cin) wf[((long long)co * d.taps + tap) * d.Cin + ci] = from_f32<T>(v);
        }
        __syncthreads();
        if (wb) {
            const int co = to * 32 + tx;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c2 = tc * 32 + ty + 8 * k;
                if (co < d.Cout_pad && c2 < d.Cin)
                    wb[((long long)c2 * d.taps + tap) * d.Cout_pad + co] = from_f32<T>(tile[tx][ty + 8 * k]);
            }
        }
        __syncthreads();
    }
}

template <typename T>
__global__ void pack_matrix_kernel(const float* __restrict__ src, int rows, int cols, int lds, T* dst, int rows_pad,
                                   int cols_pad) {
    const long long total = (long long)rows_pad * cols_pad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cols_pad);
        const int r = (int)(i / cols_pad);
        dst[i] = from_f32<T>((r < rows && c < cols) ? src[(long long)r * lds + c] : 0.f);
    }
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ x, TO* y, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = from_f32<TO>(to_f32(x[i]));
}

template <typename T>
__global__ void im2col_nchw_kernel(const float* __restrict__ x, T* col, int N, int C, int H, int W, int R, int S,
                                   int stride, int pad, int Ho, int Wo, int Kpad) {
    // one thread per (pixel, 8 consecutive k) -> 16/32-byte contiguous stores
    const int KG = Kpad / 8;
    const long long total = (long long)N * Ho * Wo * KG;
    const int K = R * S * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int kg = (int)(i % KG);
        const long long m = i / KG;
        const int wo = (int)(m % Wo);
        const long long t = m / Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        // (tap, ci) of the first k once, then incrementally: no per-element divisions; one vector store per thread
        const int k0 = kg * 8;
        int tap = k0 / C, ci = k0 - tap * C;
        int r = tap / S, s = tap - r * S;
        const int hb = ho * stride - pad, wb = wo * stride - pad;
        const float* xn = x + (long long)n * C * H * W;
        union { T e[8]; i32x4 q4[sizeof(T) == 2 ? 1 : 2]; } out;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = 0.f;
            if (k0 + e < K) {
                const int hi = hb + r, wi = wb + s;
                if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) v = xn[((long long)ci * H + hi) * W + wi];
            }
            out.e[e] = from_f32<T>(v);
            if (++ci == C) { ci = 0; if (++s == S) { s = 0; ++r; } }
        }
        i32x4* dst = reinterpret_cast<i32x4*>(col + m * Kpad + k0);
        dst[0] = out.q4[0];
        if constexpr (sizeof(T) == 4) dst[1] = out.q4[1];
    }
}

// The same through LDS: one block = IM2COL_TPX consecutive output pixels of one output row.  The C x R x span input
// patch they see is loaded once, coalesced along w (the thread-per-output-chunk form issues one 4-byte load per lane and
// k, every lane on a line of its own), then each thread assembles 16-byte output chunks from LDS.
constexpr int IM2COL_TPX = 64;
template <typename T>
__global__ __launch_bounds__(256) void im2col_nchw_tile_kernel(const float* __restrict__ x, T* col, int N, int C, int H,
                                                               int W, int R, int S, int stride, int pad, int Ho, int Wo,
                                                               int Kpad, int wo_tiles) {
    extern __shared__ float patch[];                          // [C][R][span]
    const int span = (IM2COL_TPX - 1) * stride + S;
    int b = blockIdx.x;
    const int wt = b % wo_tiles; b /= wo_tiles;
    const int ho = b % Ho;
    const int n = b / Ho;
    const int wo0 = wt * IM2COL_TPX;
    const int hb = ho * stride - pad, wb = wo0 * stride - pad;
    const float* xn = x + (long long)n * C * H * W;
    const int rows = C * R;
    for (int i = threadIdx.x; i < rows * span; i += 256) {
        const int row = i / span, j = i - row * span;
        const int ci = row / R, r = row - ci * R;
        const int hi = hb + r, wi = wb + j;
        float v = 0.f;
        if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) v = xn[((long long)ci * H + hi) * W + wi];
        patch[i] = v;
    }
    __syncthreads();
    const int KG = Kpad / 8, K = R * S * C;
    const int npx = min(IM2COL_TPX, Wo - wo0);
    for (int i = threadIdx.x; i < npx * KG; i += 256) {
        const int px = i / KG, kg = i - px * KG;
        const int k0 = kg * 8;
        int tap = k0 / C, ci = k0 - tap * C;
        int r = tap / S, sx = tap - r * S;
        union { T e[8]; i32x4 q4[sizeof(T) == 2 ? 1 : 2]; } out;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = (k0 + e < K) ? patch[(ci * R + r) * span + px * stride + sx] : 0.f;
            out.e[e] = from_f32<T>(v);
            if (++ci == C) { ci = 0; if (++sx == S) { sx = 0; ++r; } }
        }
        const long long m = ((long long)n * Ho + ho) * Wo + wo0 + px;
        i32x4* dst = reinterpret_cast<i32x4*>(col + m * Kpad + k0);
        dst[0] = out.q4[0];
        if constexpr (sizeof(T) == 4) dst[1] = out.q4[1];
    }
}

// =========================================================================
// column reductions: [rows][C] -> K sums per channel, two deterministic stages
// block = (32 chunk-columns, 8 row lanes); grid = (col blocks, row blocks, batch)
// =========================================================================
struct StatsOp {   // sum x, sum x^2
    static constexpr int K = 2;
    template <typename T> struct Args { const T* x; int ldx; };
};

// block = 256 threads arranged as TX chunk-columns x (256/TX) row lanes, TX = min(32, C/EPC)
// (power of two), so narrow tensors (C = 64) still use every thread.
template <typename T, int K, typename F>
__device__ __forceinline__ void colreduce_body(F&& elem, long long row_begin, long long row_end, int C, int TX,
                                               float* out /* [K][C] for this (rowblock,batch) */) {
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int KE = K * EPC;
    __shared__ float sred[256][KE + 1];
    const int tid = threadIdx.x;
    const int tx = tid % TX, ty = tid / TX, TY = 256 / TX;
    const int chunk = blockIdx.x * TX + tx;
    const int c0 = chunk * EPC;
    float acc[K][EPC];
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[k][e] = 0.f;
    if (c0 < C) {
#ifndef EESEG_COLRED_UNROLL
#define EESEG_COLRED_UNROLL 4     // rows whose loads are in flight per thread; BN backward reduce at 32 x 65 x 65 x 1024: 1: 152 us, 2: 117, 4: 109, 8: 117
#endif
#pragma unroll EESEG_COLRED_UNROLL
        for (long long r = row_begin + ty; r < row_end; r += TY) elem(r, c0, acc);
    }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < EPC; ++e) sred[tid][k * EPC + e] = acc[k][e];
    __syncthreads();
    for (int v = tid; v < TX * KE; v += 256) {
        const int col = v / KE, ke = v - col * KE;
        float s = 0.f;
        for (int yy = 0; yy < TY; ++yy) s += sred[yy * TX + col][ke];
        const int k = ke / EPC, e = ke - k * EPC;
        const int cg = (blockIdx.x * TX + col) * EPC + e;
        if (cg < C) out[(long long)k * C + cg] = s;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void stats_stage1(const T* x, int ldx, long long rows, long long rows_per_block,
                                                    int C, int TX, float* partials) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const long long rb = (long long)blockIdx.y * rows_per_block;
    const long long re = rb + rows_per_block < rows ? rb + rows_per_block : rows;
    colreduce_body<T, 2>(
        [&](long long r, int c0, float (*acc)[EPC]) {
            Vec<T> v = ld16(x + r * ldx + c0);
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float f = to_f32(v.e[e]);
                acc[0][e] += f;
                acc[1][e] += f * f;
            }
        },
        rb, re, C, TX, partials + (long long)blockIdx.y * 2 * C);
}

template <typename T>
__global__ __launch_bounds__(256) void colsum_stage1(const T* x, int ldx, long long rows, long long rows_per_block,
                                                     int C, int TX, float* partials) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const long long rb = (long long)blockIdx.y * rows_per_block;
    const long long re = rb + rows_per_block < rows ? rb + rows_per_block : rows;
    colreduce_body<T, 1>(
        [&](long long r, int c0, float (*acc)[EPC]) {
            Vec<T> v = ld16(x + r * ldx + c0);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[0][e] += to_f32(v.e[e]);
        },
        rb, re, C, TX, partials + (long long)blockIdx.y * C);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_stage1(const T* dy, int lddy, const T* y, int ldy, const T* x, int ldx,
                                                     const float* mean_invstd, const float* scale_shift,
                                                     long long rows, long long rows_per_block, int C, int relu, int TX,
                                                     float* partials) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const long long rb = (long long)blockIdx.y * rows_per_block;
    const long long re = rb + rows_per_block < rows ? rb + rows_per_block : rows;
    const int c0t = (blockIdx.x * TX + (threadIdx.x % TX)) * EPC;
    // relu: 0 none, 1 mask from y > 0, 2 mask recomputed as x*scale+shift > 0 (no y read), 3 byte mask written by
    // bn_apply (y = mask, one byte per 16-byte chunk, ldy = bytes per row)
    const unsigned char* bmask = reinterpret_cast<const unsigned char*>(y);
    float mu[EPC], is[EPC], msc[EPC], msh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const bool ok = c0t + e < C;
        mu[e] = ok ? mean_invstd[c0t + e] : 0.f;
        is[e] = ok ? mean_invstd[C + c0t + e] : 0.f;
        msc[e] = (ok && relu == 2) ? scale_shift[c0t + e] : 0.f;
        msh[e] = (ok && relu == 2) ? scale_shift[C + c0t + e] : 0.f;
    }
    colreduce_body<T, 2>(
        [&](long long r, int c0, float (*acc)[EPC]) {
            Vec<T> g = ld16(dy + r * lddy + c0);
            Vec<T> xv = ld16(x + r * ldx + c0);
            Vec<T> yv;
            unsigned mb = 0u;
            if (relu == 1) yv = ld16(y + r * ldy + c0);
            if (relu == 3) mb = bmask[r * ldy + c0 / EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float gf = to_f32(g.e[e]);
                const float xf = to_f32(xv.e[e]);
                if (relu == 1 && !(to_f32(yv.e[e]) > 0.f)) gf = 0.f;
                if (relu == 3 && !((mb >> e) & 1u)) gf = 0.f;
                if (relu == 2 && !(xf * msc[e] + msh[e] > 0.f)) gf = 0.f;   // sign survives the bf16 rounding
                const float xh = (xf - mu[e]) * is[e];
                acc[0][e] += gf;
                acc[1][e] += gf * xh;
            }
        },
        rb, re, C, TX, partials + (long long)blockIdx.y * 2 * C);
}

// partials[tiles][KC] -> out[seg][KC]: block = 64 columns x 16 row lanes, grid.y = segments
// of the tile range (a second call folds the segments).  Fixed order -> deterministic.
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partials, int tiles, int KC,
                                                               int tiles_per_seg, float* out, float* out2 = nullptr) {
    __shared__ double sred[16][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int t0 = blockIdx.y * tiles_per_seg;
    const int t1 = min(tiles, t0 + tiles_per_seg);
    double s = 0.0;
    if (col < KC) {
        int t = t0 + g;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (; t + 48 < t1; t += 64) {      // 4 independent loads in flight
            a0 += partials[(long long)t * KC + col];
            a1 += partials[(long long)(t + 16) * KC + col];
            a2 += partials[(long long)(t + 32) * KC + col];
            a3 += partials[(long long)(t + 48) * KC + col];
        }
        for (; t < t1; t += 16) a0 += partials[(long long)t * KC + col];
        s = (double)a0 + (double)a1 + (double)a2 + (double)a3;
    }
    sred[g][lane] = s;
    __syncthreads();
    if (g == 0 && col < KC) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sred[i][lane];
        out[(long long)blockIdx.y * KC + col] = (float)t;
        if (out2) out2[(long long)blockIdx.y * KC + col] = (float)t;      // second copy (SyncBN: one goes into the collective)
    }
}

// fused: reduce the conv-epilogue partials of 16 channels (sum and sum of squares; 64 tile lanes x 16 channel lanes)
// and finalize them in the same block (local BatchNorm: no all-reduce between the two steps)
__global__ __launch_bounds__(1024) void bn_reduce_finalize_kernel(const float* __restrict__ partials, int tiles, double count,
                                                                  const float* gamma, const float* beta, float eps,
                                                                  float momentum, float* running_mean, float* running_var,
                                                                  float* mean_invstd, float* scale_shift, int C) {
    __shared__ double sred[2][16][16];
    const int cl = threadIdx.x & 15, tl = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 16 + cl;
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
        float a0 = 0.f, b0 = 0.f;
        int t = tl;
        for (; t + 192 < tiles; t += 256) {              // 8 loads in flight; the additions keep the tile order
            float u[4], v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                u[k] = partials[((long long)(t + 64 * k) * 2 + 0) * C + c];
                v[k] = partials[((long long)(t + 64 * k) * 2 + 1) * C + c];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) { a0 += u[k]; b0 += v[k]; }
        }
        for (; t < tiles; t += 64) {
            a0 += partials[((long long)t * 2 + 0) * C + c];
            b0 += partials[((long long)t * 2 + 1) * C + c];
        }
        s1 = (double)a0;
        s2 = (double)b0;
    }
    s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
    s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
    if ((threadIdx.x & 63) < 16) { sred[0][wave][cl] = s1; sred[1][wave][cl] = s2; }
    __syncthreads();
    if (threadIdx.x < 16 && c < C) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { t1 += sred[0][i][cl]; t2 += sred[1][i][cl]; }
        // same arithmetic as bn_finalize_kernel on the fp32-rounded sums
        const double mean = (double)(float)t1 / count;
        double var = (double)(float)t2 / count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)eps);
        const float ga = gamma ? gamma[c] : 1.f, be = beta ? beta[c] : 0.f;
        if (mean_invstd) {
            mean_invstd[c] = (float)mean;
            mean_invstd[C + c] = (float)invstd;
        }
        scale_shift[c] = (float)(ga * invstd);
        scale_shift[C + c] = (float)(be - mean * ga * invstd);
        if (running_mean) {
            const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
    }
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums, double count, const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean,
                                   float* running_var, float* mean_invstd, float* scale_shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double mean = (double)sums[c] / count;
    double var = (double)sums[C + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    if (mean_invstd) {
        mean_invstd[c] = (float)mean;
        mean_invstd[C + c] = (float)invstd;
    }
    scale_shift[c] = (float)(g * invstd);
    scale_shift[C + c] = (float)(b - mean * g * invstd);
    if (running_mean) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
}

__global__ void bn_eval_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               float* scale_shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.f / sqrtf(rv[c] + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale_shift[c] = g * invstd;
    scale_shift[C + c] = b - rm[c] * g * invstd;
}

// =========================================================================
// elementwise passes over [rows][C] with row strides
// =========================================================================
// Column-fixed mapping: the host picks gridDim so that (gridDim.x*256) % (C/EPC) == 0, so a
// thread owns ONE 16-byte channel chunk for its whole grid-stride loop and keeps the
// per-channel coefficients in registers (no per-element div/mod or coefficient loads).
// Optional finalize INSIDE the apply pass (fin.sums != NULL): every thread derives the coefficients of its own 16-byte channel
// chunk from the (all-reduced) sums - the arithmetic of bn_finalize_kernel, bit for bit - and the first thread of every chunk
// also writes mean / invstd / scale / shift and updates the running statistics.  Saves the bn_finalize launch between the
// SyncBN collective and the apply pass (125 launches per step on the critical path of a data-parallel rank).
struct BnFin {
    const float* sums; double count; const float* gamma; const float* beta; float eps, momentum;
    float* running_mean; float* running_var; float* mean_invstd; float* scale_shift;
};

template <typename TI, typename TO, int U, bool FIN>
__global__ __launch_bounds__(256) void bn_apply_kernel(const TI* x, int ldx, const float* __restrict__ ss,
                                                       const TI* res, int ldres, TO* y, int ldy, long long rows,
                                                       int C, int relu, unsigned char* mask, int rev, BnFin fin, int nt) {
    constexpr int EPC = 16 / (int)sizeof(TI);
    const int cpr = C / EPC;
    const long long gid = blockIdx.x * 256ll + threadIdx.x;
    const long long T = gridDim.x * 256ll;
    const int c0 = (int)(gid % cpr) * EPC;
    const long long rstep = T / cpr;
    float sc[EPC], sh[EPC];
    if constexpr (FIN) {
        // One thread per channel chunk of the BLOCK does the (double precision) arithmetic and shares the coefficients
        // through LDS: the host makes 256 % cpr == 0 or cpr % 256 == 0 (column-fixed grid), so the block's threads cover
        // min(256, cpr) distinct chunks, chunk(tid) = chunk(tid % cpr).  Redundant fp64 divisions / square roots in all
        // 256 threads cost more than the launch this fusion removes.
        __shared__ float s_sc[256][EPC], s_sh[256][EPC];
        const int nchunk = cpr < 256 ? cpr : 256;
        const bool owner = gid < cpr;                        // exactly one thread per channel chunk of the GRID writes the outputs
        if ((int)threadIdx.x < nchunk) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const int c = c0 + e;
                const double mean = (double)fin.sums[c] / fin.count;
                double var = (double)fin.sums[C + c] / fin.count - mean * mean;
                if (var < 0.0) var = 0.0;
                const double invstd = 1.0 / sqrt(var + (double)fin.eps);
                const float g = fin.gamma ? fin.gamma[c] : 1.f, b = fin.beta ? fin.beta[c] : 0.f;
                const float sce = (float)(g * invstd), she = (float)(b - mean * g * invstd);
                s_sc[threadIdx.x][e] = sce;
                s_sh[threadIdx.x][e] = she;
                if (owner) {
                    if (fin.mean_invstd) {
                        fin.mean_invstd[c] = (float)mean;
                        fin.mean_invstd[C + c] = (float)invstd;
                    }
                    fin.scale_shift[c] = sce;
                    fin.scale_shift[C + c] = she;
                    if (fin.running_mean) {
                        const double unbiased = fin.count > 1.0 ? var * fin.count / (fin.count - 1.0) : var;
                        fin.running_mean[c] = (float)((1.0 - fin.momentum) * fin.running_mean[c] + fin.momentum * mean);
                        fin.running_var[c] = (float)((1.0 - fin.momentum) * fin.running_var[c] + fin.momentum * unbiased);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < EPC; ++e) { sc[e] = s_sc[threadIdx.x % nchunk][e]; sh[e] = s_sh[threadIdx.x % nchunk][e]; }
    } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) { sc[e] = ss[c0 + e]; sh[e] = ss[C + c0 + e]; }
    }
    for (long long r_ = gid / cpr; r_ < rows; r_ += U * rstep) {      // U rows of loads in flight per thread
        long long rr[U];
        bool on[U];
        Vec<TI> v2[U], rv2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long q = r_ + u * rstep;
            on[u] = q < rows;
            rr[u] = rev ? rows - 1 - q : q;      // rev: sweep from the end (the rows the producer wrote last)
            if (on[u]) {
                v2[u] = ld16s(x + rr[u] * ldx + c0, nt & 1);
                if (res) rv2[u] = ld16s(res + rr[u] * ldres + c0, nt & 1);
            }
        }
        if (U > 1) __builtin_amdgcn_sched_barrier(0);      // every row's loads issued before the first is consumed
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!on[u]) continue;
            const long long r = rr[u];
            float o[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float f = to_f32(v2[u].e[e]) * sc[e] + sh[e];
                if (res) f += to_f32(rv2[u].e[e]);
                if (relu) f = fmaxf(f, 0.f);
                o[e] = f;
            }
            TO* dst = y + r * ldy + c0;
            if constexpr (sizeof(TO) == sizeof(TI)) {
                Vec<TO> w;
#pragma unroll
                for (int e = 0; e < EPC; ++e) w.e[e] = from_f32<TO>(o[e]);
                st16(dst, w);
                if (mask) {        // one byte per 16-byte chunk: bit e = (stored y[e] > 0), the ReLU mask of the backward
                    unsigned m = 0u;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) m |= (to_f32(w.e[e]) > 0.f ? 1u : 0u) << e;
                    mask[r * cpr + c0 / EPC] = (unsigned char)m;
                }
            } else {   // bf16 in (8 elems) -> f32 out: two 16-byte stores
                Vec<TO> w0, w1;
#pragma unroll
                for (int e = 0; e < 4; ++e) { w0.e[e] = from_f32<TO>(o[e]); w1.e[e] = from_f32<TO>(o[4 + e]); }
                st16(dst, w0);
                st16(dst + 4, w1);
            }
        }
    }
}

// MODE 0: train BN backward (needs x, mean/invstd, gamma, sums); MODE 1: dx = g*scale
template <typename T, int MODE, int U>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* dy, int lddy, const T* y, int ldy, const T* x,
                                                           int ldx, const float* __restrict__ mean_invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ sums, float inv_count, T* dx,
                                                           int lddx, T* dres, int lddres, long long rows, int C,
                                                           int relu, const float* __restrict__ scale_shift, int rev, int nt) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int cpr = C / EPC;
    const long long gid = blockIdx.x * 256ll + threadIdx.x;
    const long long T_ = gridDim.x * 256ll;
    const int c0 = (int)(gid % cpr) * EPC;
    const long long rstep = T_ / cpr;
    // dx = ka*g + kb*(x - mean) + kc   (MODE 0: ka = gamma*invstd, kb = -ka*invstd*s1/n, kc = -ka*s0/n)
    float ka[EPC], kb[EPC], kc[EPC], km[EPC], msc[EPC], msh[EPC];
    const unsigned char* bmask = reinterpret_cast<const unsigned char*>(y);     // relu == 3: byte mask, ldy bytes per row
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int c = c0 + e;
        msc[e] = (relu == 2) ? scale_shift[c] : 0.f;
        msh[e] = (relu == 2) ? scale_shift[C + c] : 0.f;
        if (MODE == 0) {
            const float mu = mean_invstd[c], is = mean_invstd[C + c];
            const float ga = gamma ? gamma[c] : 1.f;
            ka[e] = ga * is;
            kb[e] = -ka[e] * is * sums[C + c] * inv_count;
            kc[e] = -ka[e] * sums[c] * inv_count;
            km[e] = mu;
        } else {
            ka[e] = gamma[c]; kb[e] = 0.f; kc[e] = 0.f; km[e] = 0.f;      // gamma := scale
        }
    }
    // U rows per iteration: all their loads are issued before any is consumed (memory-level parallelism at
    // 2 blocks per CU)
    for (long long r_ = gid / cpr; r_ < rows; r_ += U * rstep) {
        long long rr[U];
        bool on[U];
        Vec<T> g[U], yv[U], xv[U];
        unsigned mb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            mb[u] = 0u;
            const long long q = r_ + u * rstep;
            on[u] = q < rows;
            rr[u] = rev ? rows - 1 - q : q;      // rev: against the direction of the reduction pass before
            if (on[u]) {
                g[u] = ld16s(dy + rr[u] * lddy + c0, nt & 2);
                if (relu == 1) yv[u] = ld16(y + rr[u] * ldy + c0);
                if (relu == 3) mb[u] = bmask[rr[u] * ldy + c0 / EPC];
                if (MODE == 0) xv[u] = ld16s(x + rr[u] * ldx + c0, nt & 2);
            }
        }
        if (U > 1) __builtin_amdgcn_sched_barrier(0);      // every row's loads issued before the first is consumed
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!on[u]) continue;
            const long long r = rr[u];
            Vec<T> od, og;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float gf = to_f32(g[u].e[e]);
                if (relu == 1 && !(to_f32(yv[u].e[e]) > 0.f)) gf = 0.f;
                if (relu == 3 && !((mb[u] >> e) & 1u)) gf = 0.f;
                if (MODE == 0 && relu == 2 && !(to_f32(xv[u].e[e]) * msc[e] + msh[e] > 0.f)) gf = 0.f;
                og.e[e] = from_f32<T>(gf);
                float d = ka[e] * gf;
                if (MODE == 0) d += kb[e] * (to_f32(xv[u].e[e]) - km[e]) + kc[e];
                od.e[e] = from_f32<T>(d);
            }
            st16(dx + r * lddx + c0, od);
            if (dres) st16(dres + r * lddres + c0, og);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Train-mode BatchNorm backward in ONE launch for tensors that fit the register files of the chip (the per-GPU shards
// of a data-parallel run: 4-8 images of 65 x 65): reduce -> grid barrier -> apply, with the operands of the first pass
// kept in registers across the barrier, so dy and the conv output are read ONCE (2 streamed reads + 1 write instead of
// 4 reads + 1 write and three launches).  Grid = one 512-thread block per CU, all resident (eeseg_group_barrier among the
// row blocks of a channel group: statistics are per channel, so channel groups never wait for each other).
// Block (cg, rb) owns CHB = 8 chunks of channels (128 B of a row) and a contiguous range of `rpb` rows; thread =
// (chunk tx, row lane rl): rows rb*rpb + rl + 64k.  The first UR rows of a thread stay in registers, later ones (larger
// tensors) are read again after the barrier (L2 / Infinity Cache hits at these sizes).
// Summation order (fixed, run to run identical): thread over k ascending -> the 64 row lanes of a block in NSEG = 4 (bf16) /
// 8 (fp32) ascending segments, the segments ascending -> the row blocks of a channel group likewise.
struct BnCoopP {
    const void* dy; const void* y; const void* x; void* dx; void* dres;
    int lddy, ldy, ldx, lddx, lddres;
    const float* mean_invstd; const float* gamma; const float* scale_shift;
    float* sums; float* sums_copy; float* partials; unsigned* state;
    long long rows, rpb;
    int C, relu, ncg, nrb;
    float inv_count;
};

// ---------------------------------------------------------------------------------------------
// Train-mode BatchNorm FORWARD behind a conv in ONE launch for small tensors (per-GPU shards): every block reduces the
// conv-epilogue partial sums of ITS 64 / 32 channels itself (tiles x 2 x CHB floats out of L2: 45-90 KB per block at 4-8
// images), finalizes them and applies scale / shift (+ residual, ReLU, byte mask) to its rows - bn_reduce_finalize +
// bn_apply without the launch in between and without a barrier (the reduction is redundant per row block, not shared).
// Block / thread mapping as bn_bwd_coop_kernel.  The sums are taken in EXACTLY the order of bn_reduce_finalize_kernel
// (64 tile lanes of fp32 partial sums, then doubles: ((d0+d1)+(d2+d3)) per group of four lanes, the 16 groups in order),
// the coefficients by its arithmetic, the apply pass by bn_apply_kernel's expression: bit-identical results.
struct BnFwdP {
    const void* x; const void* res; void* y; unsigned char* mask;
    int ldx, ldres, ldy;
    const float* partials; int tiles;
    double count; const float* gamma; const float* beta; float eps, momentum;
    float* running_mean; float* running_var; float* mean_invstd; float* scale_shift;
    long long rows, rpb;
    int C, relu, ncg, nrb;
};

template <typename T>
__global__ __launch_bounds__(512) void bn_fwd_fused_kernel(BnFwdP p) {
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int CHB = 8 * EPC;                  // channels per block
    constexpr int NV = 2 * CHB;                   // (sum, sum of squares) x channels
    constexpr int NQ = 512 / NV;                  // thread groups over the 64 tile lanes (4 bf16 / 8 fp32)
    constexpr int LPQ = 64 / NQ;                  // tile lanes per thread (16 / 8): whole groups of four
    __shared__ double dsum[16][NV];
    __shared__ float s_sc[CHB], s_sh[CHB];
    const int tid = threadIdx.x;
    const int cgp = blockIdx.x % p.ncg, rb = blockIdx.x / p.ncg;
    const int C = p.C;
    {
        const int val = tid % NV, q = tid / NV;
        const int k = val / CHB, ch = cgp * CHB + (val - k * CHB);
        const float* src = p.partials + (long long)k * C + ch;
        const long long ts = 2ll * C;             // floats between consecutive tiles
        constexpr int NT = 5;                     // partial rows per tile lane, at most (tiles <= 320)
        float v[LPQ][NT];
#pragma unroll
        for (int l = 0; l < LPQ; ++l)             // ALL loads of the thread in flight at once (80 bf16 / 40 fp32): this phase is
#pragma unroll                                    // L2 latency, a dependent chain per tile lane made it 10 us
            for (int i = 0; i < NT; ++i) {
                const int t = q * LPQ + l + 64 * i;
                v[l][i] = t < p.tiles ? src[t * ts] : 0.f;           // + 0.f leaves the sum's bits alone
            }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < LPQ / 4; ++g) {       // one group of four tile lanes = one wave of bn_reduce_finalize_kernel
            double d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < NT; ++i) a += v[g * 4 + u][i];
                d[u] = (double)a;
            }
            dsum[q * (LPQ / 4) + g][val] = (d[0] + d[1]) + (d[2] + d[3]);
        }
    }
    __syncthreads();
    if (tid < CHB) {
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { t1 += dsum[i][tid]; t2 += dsum[i][CHB + tid]; }
        const int c = cgp * CHB + tid;
        const double mean = (double)(float)t1 / p.count;
        double var = (double)(float)t2 / p.count - mean * mean;
        if (var < 0.0) var = 0.0;
        const double invstd = 1.0 / sqrt(var + (double)p.eps);
        const float ga = p.gamma ? p.gamma[c] : 1.f, be = p.beta ? p.beta[c] : 0.f;
        const float sce = (float)(ga * invstd), she = (float)(be - mean * ga * invstd);
        s_sc[tid] = sce;
        s_sh[tid] = she;
        if (rb == 0) {                            // one block per channel group writes the per-channel outputs
            if (p.mean_invstd) {
                p.mean_invstd[c] = (float)mean;
                p.mean_invstd[C + c] = (float)invstd;
            }
            p.scale_shift[c] = sce;
            p.scale_shift[C + c] = she;
            if (p.running_mean) {
                const double unbiased = p.count > 1.0 ? var * p.count / (p.count - 1.0) : var;
                p.running_mean[c] = (float)((1.0 - p.momentum) * p.running_mean[c] + p.momentum * mean);
                p.running_var[c] = (float)((1.0 - p.momentum) * p.running_var[c] + p.momentum * unbiased);
            }
        }
    }
    __syncthreads();
    const int tx = tid & 7, rl = tid >> 3;
    const int c0 = cgp * CHB + tx * EPC;
    float sc[EPC], sh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sc[e] = s_sc[tx * EPC + e]; sh[e] = s_sh[tx * EPC + e]; }
    const T* x = reinterpret_cast<const T*>(p.x);
    const T* res = reinterpret_cast<const T*>(p.res);
    T* y = reinterpret_cast<T*>(p.y);
    const int cpr = C / EPC;
    const long long r_begin = (long long)rb * p.rpb;
    const long long r_end = r_begin + p.rpb < p.rows ? r_begin + p.rpb : p.rows;
    constexpr int U = 4;                          // rows of loads in flight per thread
    for (long long rq = r_begin + rl; rq < r_end; rq += U * 64) {
        Vec<T> v[U], rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = rq + 64ll * u;
            if (r < r_end) {
                v[u] = ld16(x + r * p.ldx + c0);
                if (res) rv[u] = ld16(res + r * p.ldres + c0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long r = rq + 64ll * u;
            if (r >= r_end) continue;
            Vec<T> w;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float f = to_f32(v[u].e[e]) * sc[e] + sh[e];
                if (res) f += to_f32(rv[u].e[e]);
                if (p.relu) f = fmaxf(f, 0.f);
                w.e[e] = from_f32<T>(f);
            }
            st16(y + r * p.ldy + c0, w);
            if (p.mask) {
                unsigned m = 0u;
#pragma unroll
                for (int e = 0; e < EPC; ++e) m |= (to_f32(w.e[e]) > 0.f ? 1u : 0u) << e;
                p.mask[r * cpr + c0 / EPC] = (unsigned char)m;
            }
        }
    }
}

#ifdef EESEG_COOP_STAMPS      // diagnostic build: wall-clock stamps (100 MHz) of blocks 0 and gridDim-1 behind the partials, nothing reads them
#define COOP_STAMP(i) if (threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1)) \
    reinterpret_cast<unsigned long long*>(p.partials + 256 * 2 * 64)[(blockIdx.x ? 8 : 0) + (i)] = __builtin_amdgcn_s_memrealtime()
#else
#define COOP_STAMP(i)
#endif

template <typename T, int UR>
__global__ __launch_bounds__(512) void bn_bwd_coop_kernel(BnCoopP p) {
    COOP_STAMP(0);
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int CHB = 8 * EPC;                  // channels per block
    constexpr int NSEG = 512 / (2 * CHB);         // segments of the cross-block sum
    __shared__ float sred[512][2 * EPC + 1];
    __shared__ float sseg[NSEG][2 * CHB];
    __shared__ float stot[2 * CHB];
    const int tid = threadIdx.x, tx = tid & 7, rl = tid >> 3;
    const int cgp = blockIdx.x % p.ncg, rb = blockIdx.x / p.ncg;
    const int c0 = cgp * CHB + tx * EPC;
    const int C = p.C, relu = p.relu;
    const T* dy = reinterpret_cast<const T*>(p.dy);
    const T* xin = reinterpret_cast<const T*>(p.x);
    const T* yin = reinterpret_cast<const T*>(p.y);
    const unsigned char* bmask = reinterpret_cast<const unsigned char*>(p.y);
    const long long r_begin = (long long)rb * p.rpb;
    const long long r_end = r_begin + p.rpb < p.rows ? r_begin + p.rpb : p.rows;

    float mu[EPC], is[EPC], msc[EPC], msh[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        mu[e] = p.mean_invstd[c0 + e];
        is[e] = p.mean_invstd[C + c0 + e];
        msc[e] = relu == 2 ? p.scale_shift[c0 + e] : 0.f;
        msh[e] = relu == 2 ? p.scale_shift[C + c0 + e] : 0.f;
    }
    float acc0[EPC], acc1[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }

    // masked gradient of one row chunk (the ReLU mask applied, rounded back to T: exact) and the sums it feeds
    auto row_terms = [&](long long r, Vec<T>& g, const Vec<T>& xv) {
        Vec<T> yv;
        unsigned mb = 0u;
        if (relu == 1) yv = ld16(yin + r * p.ldy + c0);
        if (relu == 3) mb = bmask[r * p.ldy + c0 / EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float gf = to_f32(g.e[e]);
            const float xf = to_f32(xv.e[e]);
            if (relu == 1 && !(to_f32(yv.e[e]) > 0.f)) gf = 0.f;
            if (relu == 3 && !((mb >> e) & 1u)) gf = 0.f;
            if (relu == 2 && !(xf * msc[e] + msh[e] > 0.f)) gf = 0.f;
            g.e[e] = from_f32<T>(gf);
            acc0[e] += gf;
            acc1[e] += gf * ((xf - mu[e]) * is[e]);
        }
    };

    // ---- pass 1: every load of the cached rows in flight before the first is consumed ----
    Vec<T> gm[UR], xc[UR];
    const long long r0 = r_begin + rl;
    {
        // per-thread row-0 pointers + a block-uniform step per cached row: no per-row 64-bit address lives in registers
        const T* dy_t = dy + r0 * p.lddy + c0;
        const T* x_t = xin + r0 * p.ldx + c0;
        const long long sdy = 64ll * p.lddy, sx = 64ll * p.ldx;
#pragma unroll
        for (int k = 0; k < UR; ++k) {
            if (r0 + 64ll * k < r_end) {
                gm[k] = ld16(dy_t + k * sdy);
                xc[k] = ld16(x_t + k * sx);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    COOP_STAMP(1);
#pragma unroll
    for (int k = 0; k < UR; ++k) {
        const long long r = r0 + 64ll * k;
        if (r < r_end) row_terms(r, gm[k], xc[k]);
    }
    COOP_STAMP(2);
    for (long long rq = r0 + 64ll * UR; rq < r_end; rq += 2 * 64) {       // rows beyond the register cache, 2 rows of loads in flight
        Vec<T> g[2], xv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long r = rq + 64ll * u;
            if (r < r_end) { g[u] = ld16(dy + r * p.lddy + c0); xv[u] = ld16(xin + r * p.ldx + c0); }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long r = rq + 64ll * u;
            if (r < r_end) row_terms(r, g[u], xv[u]);
        }
    }
    // ---- block sums -> partials[block][2][CHB]: every thread folds 64 / NSEG row lanes (all LDS reads issued before
    //      the adds), then the NSEG segments in order ----
#pragma unroll
    for (int e = 0; e < EPC; ++e) { sred[tid][e] = acc0[e]; sred[tid][EPC + e] = acc1[e]; }
    __syncthreads();
    {
        constexpr int LPS = 64 / NSEG;               // row lanes per segment
        const int val = tid % (2 * CHB), seg = tid / (2 * CHB);
        const int k = val / CHB, col = val - k * CHB;
        const int sx = col / EPC, e = col - sx * EPC;
        float v[LPS];
#pragma unroll
        for (int l = 0; l < LPS; ++l) v[l] = sred[(seg * LPS + l) * 8 + sx][k * EPC + e];
        float s = 0.f;
#pragma unroll
        for (int l = 0; l < LPS; ++l) s += v[l];
        sseg[seg][val] = s;
    }
    __syncthreads();
    if (tid < 2 * CHB) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < NSEG; ++g) s += sseg[g][tid];
        eeseg_st_sc1(p.partials + (long long)blockIdx.x * 2 * CHB + tid, s);      // read by other blocks: write-through
    }
    COOP_STAMP(3);
    eeseg_group_barrier(p.state, (unsigned)cgp, (unsigned)p.nrb);      // among the row blocks of this channel group only
    COOP_STAMP(4);
    // ---- totals of this block's channel group: NSEG segments of row blocks (8 loads in flight per thread, added in
    //      order), then the segments in order ----
    {
        const int val = tid % (2 * CHB), seg = tid / (2 * CHB);
        const int b0 = (int)((long long)seg * p.nrb / NSEG), b1 = (int)((long long)(seg + 1) * p.nrb / NSEG);
        const float* src = p.partials + (long long)cgp * 2 * CHB + val;
        const long long bs = (long long)p.ncg * 2 * CHB;
        float s = 0.f;
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = eeseg_ld_sc1(src + (b + i) * bs);
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
        }
        for (; b < b1; ++b) s += eeseg_ld_sc1(src + b * bs);
        sseg[seg][val] = s;
    }
    __syncthreads();
    if (tid < 2 * CHB) {
        float s = 0.f;
#pragma unroll
        for (int g = 0; g < NSEG; ++g) s += sseg[g][tid];
        stot[tid] = s;
        if (rb == 0) {
            const int k = tid / CHB, col = tid - k * CHB;
            p.sums[(long long)k * C + cgp * CHB + col] = s;
            if (p.sums_copy) p.sums_copy[(long long)k * C + cgp * CHB + col] = s;
        }
    }
    __syncthreads();
    // ---- pass 2: dx = ka*g + kb*(x - mean) + kc (the arithmetic of bn_bwd_apply_kernel), dres = g ----
    float ka[EPC], kb[EPC], kc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const float ga = p.gamma ? p.gamma[c0 + e] : 1.f;
        ka[e] = ga * is[e];
        kb[e] = -ka[e] * is[e] * stot[CHB + tx * EPC + e] * p.inv_count;
        kc[e] = -ka[e] * stot[tx * EPC + e] * p.inv_count;
    }
    COOP_STAMP(5);
    T* dx = reinterpret_cast<T*>(p.dx);
    T* dres = reinterpret_cast<T*>(p.dres);
    auto row_out = [&](long long r, const Vec<T>& g, const Vec<T>& xv) {
        Vec<T> od;
#pragma unroll
        for (int e = 0; e < EPC; ++e)
            od.e[e] = from_f32<T>(ka[e] * to_f32(g.e[e]) + kb[e] * (to_f32(xv.e[e]) - mu[e]) + kc[e]);
        st16(dx + r * p.lddx + c0, od);
        if (dres) st16(dres + r * p.lddres + c0, g);
    };
#pragma unroll
    for (int k = 0; k < UR; ++k) {
        const long long r = r0 + 64ll * k;
        if (r < r_end) row_out(r, gm[k], xc[k]);
    }
    for (long long rq = r0 + 64ll * UR; rq < r_end; rq += 2 * 64) {       // re-read (L2 / Infinity Cache), 2 rows in flight
        Vec<T> g2[2], xv2[2], yv2[2];
        unsigned mb2[2] = {0u, 0u};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long r = rq + 64ll * u;
            if (r < r_end) {
                g2[u] = ld16(dy + r * p.lddy + c0);
                xv2[u] = ld16(xin + r * p.ldx + c0);
                if (relu == 1) yv2[u] = ld16(yin + r * p.ldy + c0);
                if (relu == 3) mb2[u] = bmask[r * p.ldy + c0 / EPC];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long long r = rq + 64ll * u;
            if (r >= r_end) continue;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                float gf = to_f32(g2[u].e[e]);
                if (relu == 1 && !(to_f32(yv2[u].e[e]) > 0.f)) gf = 0.f;
                if (relu == 3 && !((mb2[u] >> e) & 1u)) gf = 0.f;
                // the mask coefficients are read again here (L1) instead of living in 16 registers across the barrier
                if (relu == 2 && !(to_f32(xv2[u].e[e]) * p.scale_shift[c0 + e] + p.scale_shift[C + c0 + e] > 0.f)) gf = 0.f;
                g2[u].e[e] = from_f32<T>(gf);
            }
            row_out(r, g2[u], xv2[u]);
        }
    }
    COOP_STAMP(6);
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* x, T* y, int N, int H, int W, int C, int Ho, int Wo) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int cpr = C / EPC;
    const long long total = (long long)N * Ho * Wo * cpr;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr);
        long long t = i / cpr;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float m[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) m[e] = -INFINITY;
        for (int r = 0; r < 3; ++r) {
            const int hi = ho * 2 - 1 + r;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int s = 0; s < 3; ++s) {
                const int wi = wo * 2 - 1 + s;
                if ((unsigned)wi >= (unsigned)W) continue;
                Vec<T> v = ld16(x + (((long long)n * H + hi) * W + wi) * C + ch * EPC);
#pragma unroll
                for (int e = 0; e < EPC; ++e) m[e] = fmaxf(m[e], to_f32(v.e[e]));
            }
        }
        Vec<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.e[e] = from_f32<T>(m[e]);
        st16(y + (((long long)n * Ho + ho) * Wo + wo) * C + ch * EPC, o);
    }
}

// dx[h,w] = sum over windows containing (h,w) whose FIRST maximum (scan order
// r then s, strict >, as torch's CPU kernel) is (h,w).  No atomics, no index tensor.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* x, const T* dy, T* dx, int N, int H, int W, int C,
                                                          int Ho, int Wo) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int cpr = C / EPC;
    const long long total = (long long)N * H * W * cpr;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr);
        long long t = i / cpr;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        // windows: ho with 2ho-1 <= h <= 2ho+1
        const int ho_lo = (h) / 2, ho_hi = (h + 1) / 2;
        const int wo_lo = (w) / 2, wo_hi = (w + 1) / 2;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            if (ho >= Ho) continue;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                if (wo >= Wo) continue;
                float best[EPC];
                int bidx[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) { best[e] = -INFINITY; bidx[e] = -1; }
                for (int r = 0; r < 3; ++r) {
                    const int hi = ho * 2 - 1 + r;
                    if ((unsigned)hi >= (unsigned)H) continue;
                    for (int s = 0; s < 3; ++s) {
                        const int wi = wo * 2 - 1 + s;
                        if ((unsigned)wi >= (unsigned)W) continue;
                        Vec<T> v = ld16(x + (((long long)n * H + hi) * W + wi) * C + ch * EPC);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) {
                            const float f = to_f32(v.e[e]);
                            if (f > best[e] || bidx[e] < 0) { best[e] = f; bidx[e] = hi * W + wi; }
                        }
                    }
                }
                Vec<T> g = ld16(dy + (((long long)n * Ho + ho) * Wo + wo) * C + ch * EPC);
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if (bidx[e] == h * W + w) acc[e] += to_f32(g.e[e]);
            }
        }
        Vec<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.e[e] = from_f32<T>(acc[e]);
        st16(dx + (((long long)n * H + h) * W + w) * C + ch * EPC, o);
    }
}

// The same with the forward output y = maxpool(x) at hand: an input element can only receive a window's
// gradient if it EQUALS that window's maximum, so the 9-tap rescan per window (36 loads per element) shrinks to
// one load of y per window plus, for the elements that do equal it, a scan of the EARLIER window positions for a tie.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_y_kernel(const T* x, const T* y, const T* dy, T* dx, int N, int H, int W,
                                                            int C, int Ho, int Wo) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int cpr = C / EPC;
    const long long total = (long long)N * H * W * cpr;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr);
        long long t = i / cpr;
        const int w = (int)(t % W); t /= W;
        const int h = (int)(t % H);
        const int n = (int)(t / H);
        const Vec<T> v = ld16(x + (((long long)n * H + h) * W + w) * C + ch * EPC);
        float acc[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
        const int ho_lo = h / 2, ho_hi = (h + 1) / 2;
        const int wo_lo = w / 2, wo_hi = (w + 1) / 2;
        for (int ho = ho_lo; ho <= ho_hi; ++ho) {
            if (ho >= Ho) continue;
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                if (wo >= Wo) continue;
                const Vec<T> ym = ld16(y + (((long long)n * Ho + ho) * Wo + wo) * C + ch * EPC);
                unsigned eq = 0u;
#pragma unroll
                for (int e = 0; e < EPC; ++e) eq |= (to_f32(v.e[e]) == to_f32(ym.e[e])) ? (1u << e) : 0u;
                if (!eq) continue;
                // first-maximum rule: drop the lanes for which an earlier window position holds the same value
                const int r_me = h - (ho * 2 - 1), s_me = w - (wo * 2 - 1);
                for (int r = 0; r <= r_me && eq; ++r) {
                    const int hi = ho * 2 - 1 + r;
                    if ((unsigned)hi >= (unsigned)H) continue;
                    const int s_end = (r == r_me) ? s_me : 3;
                    for (int sx = 0; sx < s_end && eq; ++sx) {
                        const int wi = wo * 2 - 1 + sx;
                        if ((unsigned)wi >= (unsigned)W) continue;
                        const Vec<T> u = ld16(x + (((long long)n * H + hi) * W + wi) * C + ch * EPC);
#pragma unroll
                        for (int e = 0; e < EPC; ++e)
                            if (to_f32(u.e[e]) == to_f32(ym.e[e])) eq &= ~(1u << e);
                    }
                }
                if (!eq) continue;
                const Vec<T> g = ld16(dy + (((long long)n * Ho + ho) * Wo + wo) * C + ch * EPC);
#pragma unroll
                for (int e = 0; e < EPC; ++e)
                    if ((eq >> e) & 1u) acc[e] += to_f32(g.e[e]);
            }
        }
        Vec<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) o.e[e] = from_f32<T>(acc[e]);
        st16(dx + (((long long)n * H + h) * W + w) * C + ch * EPC, o);
    }
}

// per-image column sums: x[n][hw][ldx] -> y[n][C] * scale   (GAP forward, sum_hw)
template <typename T>
__global__ __launch_bounds__(256) void sum_hw_kernel(const T* x, int ldx, T* y, int HW, int C, float scale) {
    constexpr int EPC = 16 / (int)sizeof(T);
    __shared__ float sred[8][32][EPC + 1];
    const int tx = threadIdx.x, ty = threadIdx.y, n = blockIdx.y;
    const int c0 = (blockIdx.x * 32 + tx) * EPC;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    if (c0 < C) {
        for (int r = ty; r < HW; r += 8) {
            Vec<T> v = ld16(x + ((long long)n * HW + r) * ldx + c0);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += to_f32(v.e[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) sred[ty][tx][e] = acc[e];
    __syncthreads();
    if (ty == 0 && c0 < C) {
        Vec<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += sred[q][tx][e];
            o.e[e] = from_f32<T>(s * scale);
        }
        st16(y + (long long)n * C + c0, o);
    }
}

// two-stage variant for large HW: grid = (col blocks, N, splits) -> fp32 partials [split][N][C]
template <typename T>
__global__ __launch_bounds__(256) void sum_hw_part_kernel(const T* x, int ldx, float* part, int N, int HW, int C,
                                                          int rows_per_split) {
    constexpr int EPC = 16 / (int)sizeof(T);
    __shared__ float sred[8][32][EPC + 1];
    const int tx = threadIdx.x, ty = threadIdx.y, n = blockIdx.y, sp = blockIdx.z;
    const int c0 = (blockIdx.x * 32 + tx) * EPC;
    const int r0 = sp * rows_per_split, r1 = min(HW, r0 + rows_per_split);
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
    if (c0 < C) {
        for (int r = r0 + ty; r < r1; r += 8) {
            Vec<T> v = ld16(x + ((long long)n * HW + r) * ldx + c0);
#pragma unroll
            for (int e = 0; e < EPC; ++e) acc[e] += to_f32(v.e[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) sred[ty][tx][e] = acc[e];
    __syncthreads();
    if (ty == 0 && c0 < C) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) s += sred[q][tx][e];
            part[((long long)sp * N + n) * C + c0 + e] = s;
        }
    }
}

template <typename T>
__global__ void sum_hw_fin_kernel(const float* __restrict__ part, T* y, int NC, int splits, float scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NC) return;
    float s = 0.f;
    for (int sp = 0; sp < splits; ++sp) s += part[(long long)sp * NC + i];
    y[i] = from_f32<T>(s * scale);
}

// y[n][hw][ldy slice] (+)= x[n][c] * scale
template <typename T>
__global__ __launch_bounds__(256) void broadcast_hw_kernel(const T* x, T* y, int ldy, int N, int HW, int C,
                                                           float scale, int accumulate) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const int cpr = C / EPC;
    const long long total = (long long)N * HW * cpr;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % cpr);
        const long long r = i / cpr;
        const int n = (int)(r / HW);
        Vec<T> v = ld16(x + (long long)n * C + ch * EPC);
        T* dst = y + r * ldy + ch * EPC;
        Vec<T> o;
        if (accumulate) {
            Vec<T> old = ld16(dst);
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.e[e] = from_f32<T>(to_f32(old.e[e]) + to_f32(v.e[e]) * scale);
        } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e) o.e[e] = from_f32<T>(to_f32(v.e[e]) * scale);
        }
        st16(dst, o);
    }
}

__device__ __forceinline__ uint32_t mix32(uint64_t z) {   // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}

template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* x, T* y, long long n, float p, uint64_t seed0,
                                                      const long long* __restrict__ step_dev, long long idx0) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const long long chunks = n / EPC;
    const float keep_scale = 1.f / (1.f - p);
    const uint32_t thr = (uint32_t)(p * 16777216.f);
    const uint64_t seed = seed0 + (step_dev ? (uint64_t)step_dev[0] * 0xA24BAED4963EE407ull : 0ull);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < chunks;
         i += (long long)gridDim.x * blockDim.x) {
        Vec<T> v = ld16(x + i * EPC);
        Vec<T> o;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const uint32_t h = mix32(seed ^ (uint64_t)(idx0 + i * EPC + e) * 0xD6E8FEB86659FD93ull) >> 8;
            o.e[e] = from_f32<T>(h >= thr ? to_f32(v.e[e]) * keep_scale : 0.f);
        }
        st16(y + i * EPC, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_inplace_kernel(T* y, const T* x, long long n) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const long long chunks = n / EPC;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < chunks;
         i += (long long)gridDim.x * blockDim.x) {
        Vec<T> a = ld16(y + i * EPC), b = ld16(x + i * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e) a.e[e] = from_f32<T>(to_f32(a.e[e]) + to_f32(b.e[e]));
        st16(y + i * EPC, a);
    }
}

// grid for the column-fixed kernels: total threads must be a multiple of chunks-per-row
int g_colfixed_cap = 512;           // eeseg_set_ew_grid_cap(): 2 blocks/CU; each thread then amortises its per-channel
                                    // coefficient prologue over many rows (end to end: 4096 -> 1024 +6 %, 1024 -> 512 +1.3 % at
                                    // B=16 and +4 % at B=4; the row loops keep two rows of loads in flight per thread)
inline int colfixed_grid(long long rows, int cpr) {
    long long items = rows * cpr;
    long long b = (items + 255) / 256;
    if (b > g_colfixed_cap) b = g_colfixed_cap;
    if (b < 1) b = 1;
    int a = cpr, c = 256;                       // m = cpr / gcd(cpr, 256)
    while (c) { int t = a % c; a = c; c = t; }
    const int m = cpr / a;
    b = (b + m - 1) / m * m;
    return (int)b;
}

// ---- host-side helpers ----------------------------------------------------
struct RowSplit { int blocks; long long rows_per_block; };
// colblocks > 0: aim at g_colreduce_blocks blocks in all (row blocks x column blocks) - a wide tensor otherwise gets
// 8 x 1024 blocks of 8 rows per thread and 16 MB of partial sums; colblocks = 0: the upper bound (workspace query)
inline RowSplit row_split(long long rows, int colblocks = 0) {
    long long max_rb = 1024;
    if (colblocks > 0 && g_colreduce_blocks > 0) {
        max_rb = g_colreduce_blocks / colblocks;
        if (max_rb < 32) max_rb = 32;
        if (max_rb > 1024) max_rb = 1024;
    }
    long long rpb = (rows + max_rb - 1) / max_rb;
    if (rpb < 64) rpb = 64;
    RowSplit s;
    s.rows_per_block = rpb;
    s.blocks = (int)((rows + rpb - 1) / rpb);
    if (s.blocks < 1) s.blocks = 1;
    return s;
}

inline int check_rows(const char* name, const void* p, int ld, int C, int dtype) {
    const int epc = 16 / eeseg_dtype_size(dtype);
    EESEG_CHECK(((uintptr_t)p & 15) == 0, EESEG_ERR_ARG, "%s: pointer not 16-byte aligned", name);
    EESEG_CHECK(C % epc == 0 && ld % epc == 0 && ld >= C, EESEG_ERR_ARG, "%s: C=%d ld=%d must be multiples of %d", name,
                C, ld, epc);
    return EESEG_OK;
}
#define CHECK_ROWS(name, p, ld, C, dt)                          \
    do {                                                        \
        int rc_ = check_rows(name, p, ld, C, dt);               \
        if (rc_) return rc_;                                    \
    } while (0)

}  // namespace

// ==========================================================================
// C ABI
// ==========================================================================
extern "C" int eeseg_pack_weight(const float* src, void* w_fwd, void* w_bwd, int Cout, int Cout_pad, int Cin, int R,
                                 int S, int src_krsc, int dtype, void* stream) {
    EESEG_CHECK(src && (w_fwd || w_bwd), EESEG_ERR_ARG, "pack_weight: null pointer");
    EESEG_CHECK(Cout_pad >= Cout && Cout > 0 && Cin > 0 && R * S > 0, EESEG_ERR_ARG, "pack_weight: bad shape");
    const long long total = (long long)Cout_pad * R * S * Cin;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EESEG_BF16)
        hipLaunchKernelGGL((pack_weight_kernel<bf16_t>), dim3(ew_grid(total)), dim3(256), 0, st, src, (bf16_t*)w_fwd,
                           (bf16_t*)w_bwd, Cout, Cout_pad, Cin, R * S, src_krsc);
    else if (dtype == EESEG_F32)
        hipLaunchKernelGGL((pack_weight_kernel<float>), dim3(ew_grid(total)), dim3(256), 0, st, src, (float*)w_fwd,
                           (float*)w_bwd, Cout, Cout_pad, Cin, R * S, src_krsc);
    else
        EESEG_CHECK(false, EESEG_ERR_ARG, "pack_weight: bad dtype");
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_set_ew_grid_cap(int blocks) {
    EESEG_CHECK(blocks >= 64 && blocks <= 65535, EESEG_ERR_ARG, "set_ew_grid_cap: out of range");
    g_colfixed_cap = blocks;
    return EESEG_OK;
}

extern "C" int eeseg_pack_weight_multi(const void* desc_table, int n, int dtype, void* stream) {
    EESEG_CHECK(desc_table && n > 0 && n <= 65535, EESEG_ERR_ARG, "pack_weight_multi: bad argument");
    static_assert(sizeof(PackDesc) == 48, "PackDesc layout is part of the ABI (48 bytes)");
    if (dtype == EESEG_BF16)
        hipLaunchKernelGGL((pack_weight_multi_kernel<bf16_t>), dim3(96, n), dim3(256), 0, (hipStream_t)stream,
                           (const PackDesc*)desc_table);
    else if (dtype == EESEG_F32)
        hipLaunchKernelGGL((pack_weight_multi_kernel<float>), dim3(96, n), dim3(256), 0, (hipStream_t)stream,
                           (const PackDesc*)desc_table);
    else
        EESEG_CHECK(false, EESEG_ERR_ARG, "pack_weight_multi: bad dtype");
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_pack_matrix(const float* src, int rows, int cols, int ld_src, void* dst, int rows_pad,
                                 int cols_pad, int dtype, void* stream) {
    EESEG_CHECK(src && dst && rows_pad >= rows && cols_pad >= cols && ld_src >= cols, EESEG_ERR_ARG,
                "pack_matrix: bad argument");
    const long long total = (long long)rows_pad * cols_pad;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EESEG_BF16)
        hipLaunchKernelGGL((pack_matrix_kernel<bf16_t>), dim3(ew_grid(total)), dim3(256), 0, st, src, rows, cols,
                           ld_src, (bf16_t*)dst, rows_pad, cols_pad);
    else
        hipLaunchKernelGGL((pack_matrix_kernel<float>), dim3(ew_grid(total)), dim3(256), 0, st, src, rows, cols,
                           ld_src, (float*)dst, rows_pad, cols_pad);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_im2col_nchw(const float* x, void* col, int N, int C, int H, int W, int R, int S, int stride,
                                 int pad, int Ho, int Wo, int Kpad, int dtype, void* stream) {
    EESEG_CHECK(x && col, EESEG_ERR_ARG, "im2col: null pointer");
    EESEG_CHECK(Kpad % 8 == 0 && Kpad >= R * S * C, EESEG_ERR_ARG, "im2col: Kpad=%d must be a multiple of 8 >= %d", Kpad,
                R * S * C);
    EESEG_CHECK(Ho == (H + 2 * pad - R) / stride + 1 && Wo == (W + 2 * pad - S) / stride + 1, EESEG_ERR_ARG,
                "im2col: inconsistent output size");
    const long long total = (long long)N * Ho * Wo * (Kpad / 8);
    hipStream_t st = (hipStream_t)stream;
    const long long lds = (long long)C * R * ((IM2COL_TPX - 1) * stride + S) * (long long)sizeof(float);
    const int wo_tiles = (Wo + IM2COL_TPX - 1) / IM2COL_TPX;
    const long long blocks = (long long)N * Ho * wo_tiles;
    if (lds <= 48 * 1024 && blocks < (1ll << 31)) {        // the stem: 3 x 7 x 133 floats = 11 KiB
        if (dtype == EESEG_BF16)
            hipLaunchKernelGGL((im2col_nchw_tile_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), (size_t)lds, st, x,
                               (bf16_t*)col, N, C, H, W, R, S, stride, pad, Ho, Wo, Kpad, wo_tiles);
        else
            hipLaunchKernelGGL((im2col_nchw_tile_kernel<float>), dim3((unsigned)blocks), dim3(256), (size_t)lds, st, x,
                               (float*)col, N, C, H, W, R, S, stride, pad, Ho, Wo, Kpad, wo_tiles);
        EESEG_LAUNCH_CHECK();
        return EESEG_OK;
    }
    if (dtype == EESEG_BF16)
        hipLaunchKernelGGL((im2col_nchw_kernel<bf16_t>), dim3(ew_grid(total)), dim3(256), 0, st, x, (bf16_t*)col, N, C,
                           H, W, R, S, stride, pad, Ho, Wo, Kpad);
    else
        hipLaunchKernelGGL((im2col_nchw_kernel<float>), dim3(ew_grid(total)), dim3(256), 0, st, x, (float*)col, N, C, H,
                           W, R, S, stride, pad, Ho, Wo, Kpad);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_cast(const void* x, int in_dtype, void* y, int out_dtype, int64_t n, void* stream) {
    EESEG_CHECK(x && y && n >= 0, EESEG_ERR_ARG, "cast: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int g = ew_grid(n);
    if (in_dtype == EESEG_F32 && out_dtype == EESEG_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)x, (bf16_t*)y, n);
    else if (in_dtype == EESEG_BF16 && out_dtype == EESEG_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (float*)y, n);
    else if (in_dtype == EESEG_F32 && out_dtype == EESEG_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)x, (float*)y, n);
    else if (in_dtype == EESEG_BF16 && out_dtype == EESEG_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n);
    else
        EESEG_CHECK(false, EESEG_ERR_ARG, "cast: bad dtype");
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int64_t eeseg_colreduce_workspace(int64_t rows, int C) {
    return ((int64_t)row_split(rows).blocks + 32) * 2 * C * (int64_t)sizeof(float);
}

static int launch_reduce_partials(const float* partials, int tiles, int KC, float* out, float* scratch,
                                  hipStream_t st, float* out2 = nullptr) {
    const int colblocks = (KC + 63) / 64;
    int segs = 1;
    if (tiles >= 2048 && scratch) {
        segs = tiles / 128;
        if (segs > 32) segs = 32;
    }
    if (segs > 1) {
        const int tps = (tiles + segs - 1) / segs;
        segs = (tiles + tps - 1) / tps;
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(colblocks, segs), dim3(1024), 0, st, partials, tiles, KC, tps,
                           scratch);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(colblocks, 1), dim3(1024), 0, st, (const float*)scratch, segs,
                           KC, segs, out, out2);
    } else {
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(colblocks, 1), dim3(1024), 0, st, partials, tiles, KC, tiles,
                           out, out2);
    }
    return EESEG_OK;
}

extern "C" int eeseg_bn_reduce_partials(const float* partials, int tiles, int KC, float* sums, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
    EESEG_CHECK(partials && sums && tiles > 0 && KC > 0, EESEG_ERR_ARG, "bn_reduce_partials: bad argument");
    float* scratch = (workspace && workspace_bytes >= (int64_t)32 * KC * (int64_t)sizeof(float)) ? (float*)workspace
                                                                                                   : nullptr;
    launch_reduce_partials(partials, tiles, KC, sums, scratch, (hipStream_t)stream);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_bn_finalize(const float* sums, double count, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, float* mean_invstd,
                                 float* scale_shift, int C, void* stream) {
    EESEG_CHECK(sums && scale_shift && C > 0 && count > 0, EESEG_ERR_ARG, "bn_finalize: bad argument");
    EESEG_CHECK((running_mean == nullptr) == (running_var == nullptr), EESEG_ERR_ARG,
                "bn_finalize: running_mean/var must both be given or both be NULL");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, sums, count, gamma,
                       beta, eps, momentum, running_mean, running_var, mean_invstd, scale_shift, C);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_bn_reduce_finalize(const float* partials, int tiles, double count, const float* gamma,
                                        const float* beta, float eps, float momentum, float* running_mean,
                                        float* running_var, float* mean_invstd, float* scale_shift, int C, void* stream) {
    EESEG_CHECK(partials && scale_shift && tiles > 0 && C > 0 && count > 0, EESEG_ERR_ARG, "bn_reduce_finalize: bad argument");
    EESEG_CHECK((running_mean == nullptr) == (running_var == nullptr), EESEG_ERR_ARG,
                "bn_reduce_finalize: running_mean/var must both be given or both be NULL");
    hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, partials, tiles,
                       count, gamma, beta, eps, momentum, running_mean, running_var, mean_invstd, scale_shift, C);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean,
                                         const float* running_var, float eps, float* scale_shift, int C,
                                         void* stream) {
    EESEG_CHECK(running_mean && running_var && scale_shift && C > 0, EESEG_ERR_ARG, "bn_eval: bad argument");
    hipLaunchKernelGGL(bn_eval_kernel, dim3((C + 127) / 128), dim3(128), 0, (hipStream_t)stream, gamma, beta,
                       running_mean, running_var, eps, scale_shift, C);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

static int bn_apply_impl(const void* x, int ldx, const float* scale_shift, const void* residual, int ldres, void* y,
                         int ldy, int64_t rows, int C, int relu, int in_dtype, int out_dtype, unsigned char* mask,
                         void* stream, BnFin fin = BnFin{nullptr, 1.0, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr}) {
    EESEG_CHECK(x && y && (scale_shift || fin.sums) && rows > 0, EESEG_ERR_ARG, "bn_apply: bad argument");
    CHECK_ROWS("bn_apply x", x, ldx, C, in_dtype);
    if (residual) CHECK_ROWS("bn_apply residual", residual, ldres, C, in_dtype);
    EESEG_CHECK(((uintptr_t)y & 15) == 0 && ldy >= C && ldy % 4 == 0, EESEG_ERR_ARG, "bn_apply: bad y/ldy");
    EESEG_CHECK(!mask || in_dtype == out_dtype, EESEG_ERR_ARG, "bn_apply: the ReLU mask needs equal in/out dtypes");
    hipStream_t st = (hipStream_t)stream;
    const int epc = 16 / eeseg_dtype_size(in_dtype);
    const int g = colfixed_grid(rows, C / epc);
    if (in_dtype == EESEG_BF16 && out_dtype == EESEG_BF16) {
        EESEG_CHECK(ldy % 8 == 0, EESEG_ERR_ARG, "bn_apply: ldy must be a multiple of 8");
        if (g_bn_rows == 4) { if (fin.sums) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, bf16_t, 4, true>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx,
                           scale_shift, (const bf16_t*)residual, ldres, (bf16_t*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); else hipLaunchKernelGGL((bn_apply_kernel<bf16_t, bf16_t, 4, false>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx,
                           scale_shift, (const bf16_t*)residual, ldres, (bf16_t*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); }
        else if (g_bn_rows == 2) { if (fin.sums) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, bf16_t, 2, true>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx,
                           scale_shift, (const bf16_t*)residual, ldres, (bf16_t*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); else hipLaunchKernelGGL((bn_apply_kernel<bf16_t, bf16_t, 2, false>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx,
                           scale_shift, (const bf16_t*)residual, ldres, (bf16_t*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); }
        else { if (fin.sums) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, bf16_t, 1, true>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx,
                           scale_shift, (const bf16_t*)residual, ldres, (bf16_t*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); else hipLaunchKernelGGL((bn_apply_kernel<bf16_t, bf16_t, 1, false>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx,
                           scale_shift, (const bf16_t*)residual, ldres, (bf16_t*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); }
    } else if (in_dtype == EESEG_BF16 && out_dtype == EESEG_F32) {
        if (g_bn_rows == 4) { if (fin.sums) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, float, 4, true>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx,
                           scale_shift, (const bf16_t*)residual, ldres, (float*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); else hipLaunchKernelGGL((bn_apply_kernel<bf16_t, float, 4, false>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx,
                           scale_shift, (const bf16_t*)residual, ldres, (float*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); }
        else if (g_bn_rows == 2) { if (fin.sums) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, float, 2, true>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx,
                           scale_shift, (const bf16_t*)residual, ldres, (float*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); else hipLaunchKernelGGL((bn_apply_kernel<bf16_t, float, 2, false>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx,
                           scale_shift, (const bf16_t*)residual, ldres, (float*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); }
        else { if (fin.sums) hipLaunchKernelGGL((bn_apply_kernel<bf16_t, float, 1, true>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx,
                           scale_shift, (const bf16_t*)residual, ldres, (float*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); else hipLaunchKernelGGL((bn_apply_kernel<bf16_t, float, 1, false>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, ldx,
                           scale_shift, (const bf16_t*)residual, ldres, (float*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); }
    } else if (in_dtype == EESEG_F32 && out_dtype == EESEG_F32) {
        if (g_bn_rows == 4) { if (fin.sums) hipLaunchKernelGGL((bn_apply_kernel<float, float, 4, true>), dim3(g), dim3(256), 0, st, (const float*)x, ldx,
                           scale_shift, (const float*)residual, ldres, (float*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); else hipLaunchKernelGGL((bn_apply_kernel<float, float, 4, false>), dim3(g), dim3(256), 0, st, (const float*)x, ldx,
                           scale_shift, (const float*)residual, ldres, (float*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); }
        else if (g_bn_rows == 2) { if (fin.sums) hipLaunchKernelGGL((bn_apply_kernel<float, float, 2, true>), dim3(g), dim3(256), 0, st, (const float*)x, ldx,
                           scale_shift, (const float*)residual, ldres, (float*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); else hipLaunchKernelGGL((bn_apply_kernel<float, float, 2, false>), dim3(g), dim3(256), 0, st, (const float*)x, ldx,
                           scale_shift, (const float*)residual, ldres, (float*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); }
        else { if (fin.sums) hipLaunchKernelGGL((bn_apply_kernel<float, float, 1, true>), dim3(g), dim3(256), 0, st, (const float*)x, ldx,
                           scale_shift, (const float*)residual, ldres, (float*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); else hipLaunchKernelGGL((bn_apply_kernel<float, float, 1, false>), dim3(g), dim3(256), 0, st, (const float*)x, ldx,
                           scale_shift, (const float*)residual, ldres, (float*)y, ldy, (long long)rows, C, relu, mask, g_bn_reverse & 1, fin, g_bn_nt); }
    } else {
        EESEG_CHECK(false, EESEG_ERR_ARG, "bn_apply: unsupported dtype pair %d -> %d", in_dtype, out_dtype);
    }
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_bn_apply(const void* x, int ldx, const float* scale_shift, const void* residual, int ldres,
                              void* y, int ldy, int64_t rows, int C, int relu, int in_dtype, int out_dtype,
                              void* stream) {
    return bn_apply_impl(x, ldx, scale_shift, residual, ldres, y, ldy, rows, C, relu, in_dtype, out_dtype, nullptr, stream);
}

extern "C" int eeseg_bn_apply_relu_mask(const void* x, int ldx, const float* scale_shift, const void* residual, int ldres,
                                        void* y, int ldy, void* relu_mask, int64_t rows, int C, int dtype, void* stream) {
    EESEG_CHECK(relu_mask, EESEG_ERR_ARG, "bn_apply_relu_mask: null mask");
    return bn_apply_impl(x, ldx, scale_shift, residual, ldres, y, ldy, rows, C, 1, dtype, dtype, (unsigned char*)relu_mask,
                         stream);
}

extern "C" int eeseg_bn_finalize_apply(const void* x, int ldx, const float* sums, double count, const float* gamma,
                                       const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                       float* mean_invstd, float* scale_shift, const void* residual, int ldres, void* y, int ldy,
                                       void* relu_mask, int64_t rows, int C, int relu, int dtype, void* stream) {
    EESEG_CHECK(sums && scale_shift && count > 0, EESEG_ERR_ARG, "bn_finalize_apply: bad argument");
    EESEG_CHECK((running_mean == nullptr) == (running_var == nullptr), EESEG_ERR_ARG,
                "bn_finalize_apply: running_mean/var must both be given or both be NULL");
    EESEG_CHECK(!relu_mask || relu, EESEG_ERR_ARG, "bn_finalize_apply: the ReLU mask needs relu");
    return bn_apply_impl(x, ldx, nullptr, residual, ldres, y, ldy, rows, C, relu, dtype, dtype, (unsigned char*)relu_mask, stream,
                         BnFin{sums, count, gamma, beta, eps, momentum, running_mean, running_var, mean_invstd, scale_shift});
}

// launches stage1 via `launch1(grid, rows_per_block, TX, partial_ptr)` then stage 2 when needed
template <typename L>
static int two_stage(L&& launch1, int64_t rows, int C, int K, int epc, float* out, void* workspace,
                     int64_t workspace_bytes, hipStream_t st, float* out2 = nullptr) {
    const int cpr = C / epc;
    int TX = 32;
    while (TX > cpr) TX >>= 1;
    if (TX < 1) TX = 1;
    const int colblocks = (cpr + TX - 1) / TX;
    const RowSplit rs = row_split(rows, colblocks);
    float* partials = out;
    float* scratch = nullptr;
    if (rs.blocks > 1) {
        const int64_t need = ((int64_t)rs.blocks + 32) * K * C * (int64_t)sizeof(float);
        EESEG_CHECK(workspace && workspace_bytes >= need, EESEG_ERR_ARG,
                    "column reduce: workspace too small (%lld bytes needed)", (long long)need);
        partials = (float*)workspace;
        scratch = partials + (int64_t)rs.blocks * K * C;
    }
    launch1(dim3(colblocks, rs.blocks), rs.rows_per_block, TX, partials);
    EESEG_LAUNCH_CHECK();
    if (rs.blocks > 1) {
        launch_reduce_partials(partials, rs.blocks, K * C, out, scratch, st, out2);
        EESEG_LAUNCH_CHECK();
    } else if (out2) {      // single-stage reduction (tiny tensor): the copy costs one more small launch
        launch_reduce_partials(out, 1, K * C, out2, nullptr, st);
        EESEG_LAUNCH_CHECK();
    }
    return EESEG_OK;
}

extern "C" int eeseg_channel_stats(const void* x, int ldx, int64_t rows, int C, float* sums, int dtype,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
    EESEG_CHECK(x && sums && rows > 0, EESEG_ERR_ARG, "channel_stats: bad argument");
    CHECK_ROWS("channel_stats", x, ldx, C, dtype);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EESEG_BF16)
        return two_stage([&](dim3 g, long long rpb, int TX, float* part) {
            hipLaunchKernelGGL((stats_stage1<bf16_t>), g, dim3(256), 0, st, (const bf16_t*)x, ldx, (long long)rows, rpb,
                               C, TX, part);
        }, rows, C, 2, 8, sums, workspace, workspace_bytes, st);
    return two_stage([&](dim3 g, long long rpb, int TX, float* part) {
        hipLaunchKernelGGL((stats_stage1<float>), g, dim3(256), 0, st, (const float*)x, ldx, (long long)rows, rpb, C, TX, part);
    }, rows, C, 2, 4, sums, workspace, workspace_bytes, st);
}

extern "C" int eeseg_colsum(const void* x, int ldx, int64_t rows, int C, float* out, int dtype, void* workspace,
                            int64_t workspace_bytes, void* stream) {
    EESEG_CHECK(x && out && rows > 0, EESEG_ERR_ARG, "colsum: bad argument");
    CHECK_ROWS("colsum", x, ldx, C, dtype);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EESEG_BF16)
        return two_stage([&](dim3 g, long long rpb, int TX, float* part) {
            hipLaunchKernelGGL((colsum_stage1<bf16_t>), g, dim3(256), 0, st, (const bf16_t*)x, ldx, (long long)rows,
                               rpb, C, TX, part);
        }, rows, C, 1, 8, out, workspace, workspace_bytes, st);
    return two_stage([&](dim3 g, long long rpb, int TX, float* part) {
        hipLaunchKernelGGL((colsum_stage1<float>), g, dim3(256), 0, st, (const float*)x, ldx, (long long)rows, rpb, C, TX, part);
    }, rows, C, 1, 4, out, workspace, workspace_bytes, st);
}

extern "C" int eeseg_bn_bwd_reduce(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                                   const float* mean_invstd, const float* scale_shift, int64_t rows, int C, int relu,
                                   float* sums, float* sums_copy, int dtype, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
    EESEG_CHECK(dy && x && mean_invstd && sums && rows > 0 && ((relu != 1 && relu != 3) || y) && (relu != 2 || scale_shift) &&
                    relu >= 0 && relu <= 3, EESEG_ERR_ARG, "bn_bwd_reduce: bad argument");
    EESEG_CHECK(relu != 3 || ldy >= C / (16 / eeseg_dtype_size(dtype)), EESEG_ERR_ARG, "bn_bwd_reduce: mask row too short");
    CHECK_ROWS("bn_bwd_reduce dy", dy, lddy, C, dtype);
    CHECK_ROWS("bn_bwd_reduce x", x, ldx, C, dtype);
    if (relu == 1) CHECK_ROWS("bn_bwd_reduce y", y, ldy, C, dtype);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EESEG_BF16)
        return two_stage([&](dim3 g, long long rpb, int TX, float* part) {
            hipLaunchKernelGGL((bn_bwd_stage1<bf16_t>), g, dim3(256), 0, st, (const bf16_t*)dy, lddy, (const bf16_t*)y,
                               ldy, (const bf16_t*)x, ldx, mean_invstd, scale_shift, (long long)rows, rpb, C, relu, TX, part);
        }, rows, C, 2, 8, sums, workspace, workspace_bytes, st, sums_copy);
    return two_stage([&](dim3 g, long long rpb, int TX, float* part) {
        hipLaunchKernelGGL((bn_bwd_stage1<float>), g, dim3(256), 0, st, (const float*)dy, lddy, (const float*)y, ldy,
                           (const float*)x, ldx, mean_invstd, scale_shift, (long long)rows, rpb, C, relu, TX, part);
    }, rows, C, 2, 4, sums, workspace, workspace_bytes, st, sums_copy);
}

extern "C" int eeseg_bn_bwd_apply(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                                  const float* mean_invstd, const float* gamma, const float* sums, double count,
                                  void* dx, int lddx, void* dres, int lddres, int64_t rows, int C, int relu,
                                  const float* scale_shift, int dtype, void* stream) {
    EESEG_CHECK(dy && x && mean_invstd && sums && dx && rows > 0 && count > 0 && ((relu != 1 && relu != 3) || y) &&
                    (relu != 2 || scale_shift) && relu >= 0 && relu <= 3, EESEG_ERR_ARG, "bn_bwd_apply: bad argument");
    EESEG_CHECK(relu != 3 || ldy >= C / (16 / eeseg_dtype_size(dtype)), EESEG_ERR_ARG, "bn_bwd_apply: mask row too short");
    CHECK_ROWS("bn_bwd_apply dy", dy, lddy, C, dtype);
    CHECK_ROWS("bn_bwd_apply x", x, ldx, C, dtype);
    CHECK_ROWS("bn_bwd_apply dx", dx, lddx, C, dtype);
    if (relu == 1) CHECK_ROWS("bn_bwd_apply y", y, ldy, C, dtype);
    if (dres) CHECK_ROWS("bn_bwd_apply dres", dres, lddres, C, dtype);
    hipStream_t st = (hipStream_t)stream;
    const int epc = 16 / eeseg_dtype_size(dtype);
    const int g = colfixed_grid(rows, C / epc);
    const float inv = (float)(1.0 / count);
    if (dtype == EESEG_BF16) {
        if (g_bn_bwd_rows == 4) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 0, 4>), dim3(g), dim3(256), 0, st, (const bf16_t*)dy, lddy,
                           (const bf16_t*)y, ldy, (const bf16_t*)x, ldx, mean_invstd, gamma, sums, inv, (bf16_t*)dx,
                           lddx, (bf16_t*)dres, lddres, (long long)rows, C, relu, scale_shift, (g_bn_reverse >> 1) & 1, g_bn_nt);
        else if (g_bn_bwd_rows == 2) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 0, 2>), dim3(g), dim3(256), 0, st, (const bf16_t*)dy, lddy,
                           (const bf16_t*)y, ldy, (const bf16_t*)x, ldx, mean_invstd, gamma, sums, inv, (bf16_t*)dx,
                           lddx, (bf16_t*)dres, lddres, (long long)rows, C, relu, scale_shift, (g_bn_reverse >> 1) & 1, g_bn_nt);
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 0, 1>), dim3(g), dim3(256), 0, st, (const bf16_t*)dy, lddy,
                           (const bf16_t*)y, ldy, (const bf16_t*)x, ldx, mean_invstd, gamma, sums, inv, (bf16_t*)dx,
                           lddx, (bf16_t*)dres, lddres, (long long)rows, C, relu, scale_shift, (g_bn_reverse >> 1) & 1, g_bn_nt);
    }
    else {
        if (g_bn_bwd_rows == 4) hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 0, 4>), dim3(g), dim3(256), 0, st, (const float*)dy, lddy,
                           (const float*)y, ldy, (const float*)x, ldx, mean_invstd, gamma, sums, inv, (float*)dx, lddx,
                           (float*)dres, lddres, (long long)rows, C, relu, scale_shift, (g_bn_reverse >> 1) & 1, g_bn_nt);
        else if (g_bn_bwd_rows == 2) hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 0, 2>), dim3(g), dim3(256), 0, st, (const float*)dy, lddy,
                           (const float*)y, ldy, (const float*)x, ldx, mean_invstd, gamma, sums, inv, (float*)dx, lddx,
                           (float*)dres, lddres, (long long)rows, C, relu, scale_shift, (g_bn_reverse >> 1) & 1, g_bn_nt);
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 0, 1>), dim3(g), dim3(256), 0, st, (const float*)dy, lddy,
                           (const float*)y, ldy, (const float*)x, ldx, mean_invstd, gamma, sums, inv, (float*)dx, lddx,
                           (float*)dres, lddres, (long long)rows, C, relu, scale_shift, (g_bn_reverse >> 1) & 1, g_bn_nt);
    }
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

// ---- one-launch BatchNorm backward (bn_bwd_coop_kernel) ----
int g_bn_coop_max_rpt = 18;          // rows per thread beyond which the fused form is not offered: the register cache holds 12, and with
                                     // most rows re-read (33 per thread at 32 x 65 x 65 x 256) the fused form measured slower than the two launches

struct CoopPlan { int ncg, nrb, ur; long long rpb; };
static bool bn_coop_plan(int64_t rows, int C, int dtype, CoopPlan* pl) {
    const int epc = 16 / eeseg_dtype_size(dtype), chb = 8 * epc;
    const int cus = eeseg_get_option(EESEG_OPT_CONV_CUS);          // CUs a launch may count on (lower while collectives hold some)
    if (rows <= 0 || C <= 0 || C % chb != 0 || C / chb > cus || C / chb > EESEG_BARRIER_GROUPS) return false;
    pl->ncg = C / chb;
    pl->nrb = cus / pl->ncg;
    if ((int64_t)pl->nrb * 64 > rows) pl->nrb = (int)((rows + 63) / 64);
    if (pl->nrb < 1) pl->nrb = 1;
    pl->rpb = (rows + pl->nrb - 1) / pl->nrb;
    pl->nrb = (int)((rows + pl->rpb - 1) / pl->rpb);
    const long long rpt = (pl->rpb + 63) / 64;
    if (rpt > g_bn_coop_max_rpt) return false;
    pl->ur = rpt <= 6 ? 6 : (rpt <= 9 ? 9 : 12);
    return true;
}

extern "C" int eeseg_bn_bwd_coop_ok(int64_t rows, int C, int dtype) {
    if (eeseg_dtype_size(dtype) == 0) return 0;
    CoopPlan pl;
    return bn_coop_plan(rows, C, dtype, &pl) ? 1 : 0;
}

extern "C" int64_t eeseg_bn_bwd_coop_workspace(void) { return 256ll * 2 * 64 * (int64_t)sizeof(float) + 256; }    // + 256 B of diagnostic stamps (EESEG_COOP_STAMPS builds)

extern "C" int eeseg_bn_bwd_coop(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                                 const float* mean_invstd, const float* gamma, const float* scale_shift, double count,
                                 float* sums, float* sums_copy, void* dx, int lddx, void* dres, int lddres, int64_t rows,
                                 int C, int relu, int dtype, void* workspace, int64_t workspace_bytes,
                                 void* barrier_state, void* stream) {
    EESEG_CHECK(dy && x && mean_invstd && sums && dx && rows > 0 && count > 0 && ((relu != 1 && relu != 3) || y) &&
                    (relu != 2 || scale_shift) && relu >= 0 && relu <= 3, EESEG_ERR_ARG, "bn_bwd_coop: bad argument");
    EESEG_CHECK(eeseg_dtype_size(dtype) != 0, EESEG_ERR_ARG, "bn_bwd_coop: bad dtype");
    EESEG_CHECK(relu != 3 || ldy >= C / (16 / eeseg_dtype_size(dtype)), EESEG_ERR_ARG, "bn_bwd_coop: mask row too short");
    EESEG_CHECK(dx != dy && dres != dy && dx != x, EESEG_ERR_ARG, "bn_bwd_coop: outputs must not alias the inputs");
    CHECK_ROWS("bn_bwd_coop dy", dy, lddy, C, dtype);
    CHECK_ROWS("bn_bwd_coop x", x, ldx, C, dtype);
    CHECK_ROWS("bn_bwd_coop dx", dx, lddx, C, dtype);
    if (relu == 1) CHECK_ROWS("bn_bwd_coop y", y, ldy, C, dtype);
    if (dres) CHECK_ROWS("bn_bwd_coop dres", dres, lddres, C, dtype);
    CoopPlan pl;
    EESEG_CHECK(bn_coop_plan(rows, C, dtype, &pl), EESEG_ERR_ARG,
                "bn_bwd_coop: rows=%lld C=%d does not fit the one-launch form (ask eeseg_bn_bwd_coop_ok first)", (long long)rows, C);
    EESEG_CHECK(workspace && workspace_bytes >= eeseg_bn_bwd_coop_workspace() && barrier_state &&
                    ((uintptr_t)barrier_state & 127) == 0, EESEG_ERR_ARG, "bn_bwd_coop: workspace / barrier state (128-byte aligned) missing");
    BnCoopP p;
    p.dy = dy; p.y = y; p.x = x; p.dx = dx; p.dres = dres;
    p.lddy = lddy; p.ldy = ldy; p.ldx = ldx; p.lddx = lddx; p.lddres = lddres;
    p.mean_invstd = mean_invstd; p.gamma = gamma; p.scale_shift = scale_shift;
    p.sums = sums; p.sums_copy = sums_copy; p.partials = (float*)workspace; p.state = (unsigned*)barrier_state;
    p.rows = rows; p.rpb = pl.rpb; p.C = C; p.relu = relu; p.ncg = pl.ncg; p.nrb = pl.nrb;
    p.inv_count = (float)(1.0 / count);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(pl.ncg * pl.nrb);
#define EESEG_COOP_LAUNCH(T_) \
    { if (pl.ur == 6) hipLaunchKernelGGL((bn_bwd_coop_kernel<T_, 6>), grid, dim3(512), 0, st, p); \
      else if (pl.ur == 9) hipLaunchKernelGGL((bn_bwd_coop_kernel<T_, 9>), grid, dim3(512), 0, st, p); \
      else hipLaunchKernelGGL((bn_bwd_coop_kernel<T_, 12>), grid, dim3(512), 0, st, p); }
    if (dtype == EESEG_BF16) EESEG_COOP_LAUNCH(bf16_t)
    else EESEG_COOP_LAUNCH(float)
#undef EESEG_COOP_LAUNCH
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

// ---- one-launch BatchNorm forward behind a conv (bn_fwd_fused_kernel) ----
int g_bn_fwd_fused_max_tiles = 320;  // conv partial-sum rows beyond which every block reducing them itself costs more than a launch

extern "C" int eeseg_bn_fwd_fused_ok(int64_t rows, int C, int tiles, int dtype) {
    if (eeseg_dtype_size(dtype) == 0 || tiles <= 0 || tiles > g_bn_fwd_fused_max_tiles) return 0;
    const int chb = 8 * (16 / eeseg_dtype_size(dtype));
    const int cus = eeseg_get_option(EESEG_OPT_CONV_CUS);
    return (rows > 0 && C > 0 && C % chb == 0 && C / chb <= cus) ? 1 : 0;
}

extern "C" int eeseg_bn_fwd_fused(const void* x, int ldx, const float* partials, int tiles, double count, const float* gamma,
                                  const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                  float* mean_invstd, float* scale_shift, const void* residual, int ldres, void* y, int ldy,
                                  void* relu_mask, int64_t rows, int C, int relu, int dtype, void* stream) {
    EESEG_CHECK(x && y && partials && scale_shift && count > 0, EESEG_ERR_ARG, "bn_fwd_fused: bad argument");
    EESEG_CHECK((running_mean == nullptr) == (running_var == nullptr), EESEG_ERR_ARG,
                "bn_fwd_fused: running_mean/var must both be given or both be NULL");
    EESEG_CHECK(!relu_mask || relu, EESEG_ERR_ARG, "bn_fwd_fused: the ReLU mask needs relu");
    EESEG_CHECK(eeseg_bn_fwd_fused_ok(rows, C, tiles, dtype), EESEG_ERR_ARG,
                "bn_fwd_fused: rows=%lld C=%d tiles=%d does not fit the one-launch form (ask eeseg_bn_fwd_fused_ok first)",
                (long long)rows, C, tiles);
    CHECK_ROWS("bn_fwd_fused x", x, ldx, C, dtype);
    CHECK_ROWS("bn_fwd_fused y", y, ldy, C, dtype);
    if (residual) CHECK_ROWS("bn_fwd_fused residual", residual, ldres, C, dtype);
    const int chb = 8 * (16 / eeseg_dtype_size(dtype));
    const int cus = eeseg_get_option(EESEG_OPT_CONV_CUS);
    BnFwdP p;
    p.x = x; p.res = residual; p.y = y; p.mask = (unsigned char*)relu_mask;
    p.ldx = ldx; p.ldres = ldres; p.ldy = ldy;
    p.partials = partials; p.tiles = tiles;
    p.count = count; p.gamma = gamma; p.beta = beta; p.eps = eps; p.momentum = momentum;
    p.running_mean = running_mean; p.running_var = running_var; p.mean_invstd = mean_invstd; p.scale_shift = scale_shift;
    p.rows = rows; p.C = C; p.relu = relu;
    p.ncg = C / chb;
    p.nrb = 2 * cus / p.ncg;                       // two 512-thread blocks per CU (no barrier: residency is not required)
    if ((int64_t)p.nrb * 64 > rows) p.nrb = (int)((rows + 63) / 64);
    if (p.nrb < 1) p.nrb = 1;
    p.rpb = (rows + p.nrb - 1) / p.nrb;
    p.nrb = (int)((rows + p.rpb - 1) / p.rpb);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EESEG_BF16) hipLaunchKernelGGL((bn_fwd_fused_kernel<bf16_t>), dim3(p.ncg * p.nrb), dim3(512), 0, st, p);
    else hipLaunchKernelGGL((bn_fwd_fused_kernel<float>), dim3(p.ncg * p.nrb), dim3(512), 0, st, p);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_scale_act_bwd(const void* dy, int lddy, const void* y, int ldy, const float* scale, void* dx,
                                   int lddx, void* dres, int lddres, int64_t rows, int C, int relu, int dtype,
                                   void* stream) {
    EESEG_CHECK(dy && scale && dx && rows > 0 && (!relu || y), EESEG_ERR_ARG, "scale_act_bwd: bad argument");
    CHECK_ROWS("scale_act_bwd dy", dy, lddy, C, dtype);
    CHECK_ROWS("scale_act_bwd dx", dx, lddx, C, dtype);
    if (relu) CHECK_ROWS("scale_act_bwd y", y, ldy, C, dtype);
    if (dres) CHECK_ROWS("scale_act_bwd dres", dres, lddres, C, dtype);
    hipStream_t st = (hipStream_t)stream;
    const int epc = 16 / eeseg_dtype_size(dtype);
    const int g = colfixed_grid(rows, C / epc);
    if (dtype == EESEG_BF16) {
        if (g_bn_bwd_rows == 4) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 1, 4>), dim3(g), dim3(256), 0, st, (const bf16_t*)dy, lddy,
                           (const bf16_t*)y, ldy, (const bf16_t*)nullptr, 0, (const float*)nullptr, scale,
                           (const float*)nullptr, 0.f, (bf16_t*)dx, lddx, (bf16_t*)dres, lddres, (long long)rows, C, relu ? 1 : 0,
                           (const float*)nullptr, 0, 0);
        else if (g_bn_bwd_rows == 2) hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 1, 2>), dim3(g), dim3(256), 0, st, (const bf16_t*)dy, lddy,
                           (const bf16_t*)y, ldy, (const bf16_t*)nullptr, 0, (const float*)nullptr, scale,
                           (const float*)nullptr, 0.f, (bf16_t*)dx, lddx, (bf16_t*)dres, lddres, (long long)rows, C, relu ? 1 : 0,
                           (const float*)nullptr, 0, 0);
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<bf16_t, 1, 1>), dim3(g), dim3(256), 0, st, (const bf16_t*)dy, lddy,
                           (const bf16_t*)y, ldy, (const bf16_t*)nullptr, 0, (const float*)nullptr, scale,
                           (const float*)nullptr, 0.f, (bf16_t*)dx, lddx, (bf16_t*)dres, lddres, (long long)rows, C, relu ? 1 : 0,
                           (const float*)nullptr, 0, 0);
    }
    else {
        if (g_bn_bwd_rows == 4) hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 1, 4>), dim3(g), dim3(256), 0, st, (const float*)dy, lddy,
                           (const float*)y, ldy, (const float*)nullptr, 0, (const float*)nullptr, scale,
                           (const float*)nullptr, 0.f, (float*)dx, lddx, (float*)dres, lddres, (long long)rows, C, relu ? 1 : 0,
                           (const float*)nullptr, 0, 0);
        else if (g_bn_bwd_rows == 2) hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 1, 2>), dim3(g), dim3(256), 0, st, (const float*)dy, lddy,
                           (const float*)y, ldy, (const float*)nullptr, 0, (const float*)nullptr, scale,
                           (const float*)nullptr, 0.f, (float*)dx, lddx, (float*)dres, lddres, (long long)rows, C, relu ? 1 : 0,
                           (const float*)nullptr, 0, 0);
        else hipLaunchKernelGGL((bn_bwd_apply_kernel<float, 1, 1>), dim3(g), dim3(256), 0, st, (const float*)dy, lddy,
                           (const float*)y, ldy, (const float*)nullptr, 0, (const float*)nullptr, scale,
                           (const float*)nullptr, 0.f, (float*)dx, lddx, (float*)dres, lddres, (long long)rows, C, relu ? 1 : 0,
                           (const float*)nullptr, 0, 0);
    }
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_maxpool3x3s2(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int dtype,
                                  void* stream) {
    EESEG_CHECK(x && y, EESEG_ERR_ARG, "maxpool: null pointer");
    EESEG_CHECK(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, EESEG_ERR_ARG, "maxpool: bad output size");
    CHECK_ROWS("maxpool x", x, C, C, dtype);
    CHECK_ROWS("maxpool y", y, C, C, dtype);
    hipStream_t st = (hipStream_t)stream;
    const int epc = 16 / eeseg_dtype_size(dtype);
    const int g = ew_grid((long long)N * Ho * Wo * (C / epc));
    if (dtype == EESEG_BF16)
        hipLaunchKernelGGL((maxpool_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, H, W, C,
                           Ho, Wo);
    else
        hipLaunchKernelGGL((maxpool_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, Ho,
                           Wo);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_maxpool3x3s2_bwd(const void* x, const void* y, const void* dy, void* dx, int N, int H, int W, int C,
                                      int Ho, int Wo, int dtype, void* stream) {
    EESEG_CHECK(x && dy && dx, EESEG_ERR_ARG, "maxpool_bwd: null pointer");
    EESEG_CHECK(Ho == (H + 2 - 3) / 2 + 1 && Wo == (W + 2 - 3) / 2 + 1, EESEG_ERR_ARG, "maxpool_bwd: bad output size");
    CHECK_ROWS("maxpool_bwd x", x, C, C, dtype);
    CHECK_ROWS("maxpool_bwd dy", dy, C, C, dtype);
    CHECK_ROWS("maxpool_bwd dx", dx, C, C, dtype);
    hipStream_t st = (hipStream_t)stream;
    const int epc = 16 / eeseg_dtype_size(dtype);
    const int g = ew_grid((long long)N * H * W * (C / epc));
    if (y != nullptr) {
        CHECK_ROWS("maxpool_bwd y", y, C, C, dtype);
        if (dtype == EESEG_BF16)
            hipLaunchKernelGGL((maxpool_bwd_y_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)y,
                               (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, Ho, Wo);
        else
            hipLaunchKernelGGL((maxpool_bwd_y_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)y,
                               (const float*)dy, (float*)dx, N, H, W, C, Ho, Wo);
        EESEG_LAUNCH_CHECK();
        return EESEG_OK;
    }
    if (dtype == EESEG_BF16)
        hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy,
                           (bf16_t*)dx, N, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (const float*)dy,
                           (float*)dx, N, H, W, C, Ho, Wo);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_sum_hw(const void* x, int ldx, void* y, int N, int HW, int C, float scale, int dtype,
                            void* workspace, int64_t workspace_bytes, void* stream) {
    EESEG_CHECK(x && y && N > 0 && HW > 0, EESEG_ERR_ARG, "sum_hw: bad argument");
    CHECK_ROWS("sum_hw x", x, ldx, C, dtype);
    CHECK_ROWS("sum_hw y", y, C, C, dtype);
    hipStream_t st = (hipStream_t)stream;
    const int epc = 16 / eeseg_dtype_size(dtype);
    const int colblocks = (C / epc + 31) / 32;
    int splits = HW / 256;                      // >= 256 rows per split
    if (splits > 16) splits = 16;
    if (splits >= 2 && workspace && workspace_bytes >= (int64_t)splits * N * C * (int64_t)sizeof(float)) {
        const int rps = (HW + splits - 1) / splits;
        dim3 g(colblocks, N, splits);
        if (dtype == EESEG_BF16) {
            hipLaunchKernelGGL((sum_hw_part_kernel<bf16_t>), g, dim3(32, 8), 0, st, (const bf16_t*)x, ldx, (float*)workspace,
                               N, HW, C, rps);
            hipLaunchKernelGGL((sum_hw_fin_kernel<bf16_t>), dim3((N * C + 255) / 256), dim3(256), 0, st,
                               (const float*)workspace, (bf16_t*)y, N * C, splits, scale);
        } else {
            hipLaunchKernelGGL((sum_hw_part_kernel<float>), g, dim3(32, 8), 0, st, (const float*)x, ldx, (float*)workspace, N,
                               HW, C, rps);
            hipLaunchKernelGGL((sum_hw_fin_kernel<float>), dim3((N * C + 255) / 256), dim3(256), 0, st,
                               (const float*)workspace, (float*)y, N * C, splits, scale);
        }
        EESEG_LAUNCH_CHECK();
        return EESEG_OK;
    }
    dim3 g(colblocks, N);
    if (dtype == EESEG_BF16)
        hipLaunchKernelGGL((sum_hw_kernel<bf16_t>), g, dim3(32, 8), 0, st, (const bf16_t*)x, ldx, (bf16_t*)y, HW, C, scale);
    else
        hipLaunchKernelGGL((sum_hw_kernel<float>), g, dim3(32, 8), 0, st, (const float*)x, ldx, (float*)y, HW, C, scale);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_broadcast_hw(const void* x, void* y, int ldy, int N, int HW, int C, float scale, int accumulate,
                                  int dtype, void* stream) {
    EESEG_CHECK(x && y && N > 0 && HW > 0, EESEG_ERR_ARG, "broadcast_hw: bad argument");
    CHECK_ROWS("broadcast_hw x", x, C, C, dtype);
    CHECK_ROWS("broadcast_hw y", y, ldy, C, dtype);
    hipStream_t st = (hipStream_t)stream;
    const int epc = 16 / eeseg_dtype_size(dtype);
    const int g = ew_grid((long long)N * HW * (C / epc));
    if (dtype == EESEG_BF16)
        hipLaunchKernelGGL((broadcast_hw_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, ldy, N,
                           HW, C, scale, accumulate);
    else
        hipLaunchKernelGGL((broadcast_hw_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (float*)y, ldy, N, HW,
                           C, scale, accumulate);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, const int64_t* step_dev,
                             int64_t index_offset, int dtype, void* stream) {
    EESEG_CHECK(x && y && n > 0 && p >= 0.f && p < 1.f && index_offset >= 0, EESEG_ERR_ARG, "dropout: bad argument");
    const int epc = 16 / eeseg_dtype_size(dtype);
    EESEG_CHECK(n % epc == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, EESEG_ERR_ARG,
                "dropout: n must be a multiple of %d and pointers 16-byte aligned", epc);
    hipStream_t st = (hipStream_t)stream;
    const int g = ew_grid(n / epc);
    if (dtype == EESEG_BF16)
        hipLaunchKernelGGL((dropout_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, (long long)n,
                           p, seed, (const long long*)step_dev, (long long)index_offset);
    else
        hipLaunchKernelGGL((dropout_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (float*)y, (long long)n, p,
                           seed, (const long long*)step_dev, (long long)index_offset);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

namespace {
// rows x row_bytes (16-byte multiples) from a row pitch of src_ld to a row pitch of dst_ld bytes
__global__ __launch_bounds__(256) void copy2d_kernel(const char* __restrict__ src, long long src_ld, char* __restrict__ dst,
                                                     long long dst_ld, long long rows, int chunks) {
    const long long total = rows * chunks;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / chunks;
        const int c = (int)(i - r * chunks);
        *reinterpret_cast<i32x4*>(dst + r * dst_ld + (long long)c * 16) =
            *reinterpret_cast<const i32x4*>(src + r * src_ld + (long long)c * 16);
    }
}
}  // namespace

extern "C" int eeseg_copy2d(const void* src, int64_t src_ld_bytes, void* dst, int64_t dst_ld_bytes, int64_t rows,
                            int64_t row_bytes, void* stream) {
    EESEG_CHECK(src && dst && rows > 0 && row_bytes > 0, EESEG_ERR_ARG, "copy2d: bad argument");
    EESEG_CHECK(row_bytes % 16 == 0 && src_ld_bytes % 16 == 0 && dst_ld_bytes % 16 == 0 && src_ld_bytes >= row_bytes &&
                    dst_ld_bytes >= row_bytes && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && row_bytes / 16 < (1ll << 31),
                EESEG_ERR_ARG, "copy2d: rows, pitches and pointers must be 16-byte multiples / aligned");
    const long long chunks = row_bytes / 16;
    long long blocks = (rows * chunks + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(copy2d_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const char*)src,
                       (long long)src_ld_bytes, (char*)dst, (long long)dst_ld_bytes, (long long)rows, (int)chunks);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_add_inplace(void* y, const void* x, int64_t n, int dtype, void* stream) {
    EESEG_CHECK(x && y && n > 0, EESEG_ERR_ARG, "add_inplace: bad argument");
    const int epc = 16 / eeseg_dtype_size(dtype);
    EESEG_CHECK(n % epc == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, EESEG_ERR_ARG,
                "add_inplace: n must be a multiple of %d and pointers 16-byte aligned", epc);
    hipStream_t st = (hipStream_t)stream;
    const int g = ew_grid(n / epc);
    if (dtype == EESEG_BF16)
        hipLaunchKernelGGL((add_inplace_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (bf16_t*)y, (const bf16_t*)x,
                           (long long)n);
    else
        hipLaunchKernelGGL((add_inplace_kernel<float>), dim3(g), dim3(256), 0, st, (float*)y, (const float*)x,
                           (long long)n);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}
