// Fused bilinear-upsample + per-pixel heads for the early-exit outputs (gfx950).
//
// The exits produce low-resolution logits [N,h,w,ldc] (NHWC fp32, C <= 32 valid
// channels, row = 128 bytes when ldc == 32).  The reference upsamples every exit to
// the input size and stacks [E,B,C,H,W] before the loss / argmax / entropy
// (from_deepv3_new.py:149-155, my_pixelwise_xentropy.py:36-38, seg_metrics.py:13-28,
// eval_br_ent.py:58-60).  These kernels interpolate on the fly instead: a 32-lane
// half wave owns one full-resolution pixel (lane = class), so the 4 neighbour rows
// are 128-byte coalesced loads and softmax / argmax are half-wave DPP reductions.
#include "eeseg_common.h"

int g_ce_span = 1;        // EESEG_OPT_CE_SPAN: 1 = thread-per-span cross-entropy kernels (default), 0 = half-wave-per-pixel

namespace {

constexpr int CMAX = 32;

// torch's area_pixel_compute_source_index(align_corners=False) + guard_index_and_lambda
struct Src { int i0, i1; float l0, l1; };
__device__ __forceinline__ Src src_index(int dst, float scale, int in_size) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    int i0 = (int)floorf(s);
    if (i0 > in_size - 1) i0 = in_size - 1;
    float l1 = s - (float)i0;
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    Src r;
    r.i0 = i0;
    r.i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    r.l1 = l1;
    r.l0 = 1.f - l1;
    return r;
}

__device__ __forceinline__ float half_max(float v) {   // reduce over the 32-lane half wave
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// interpolated logit of class `c` at full-res pixel (y,x) of image n
__device__ __forceinline__ float interp(const float* __restrict__ lr, int ldc, int h, int w, int n, const Src& sy,
                                        const Src& sx, int c) {
    const float* base = lr + (size_t)n * h * w * ldc + c;
    const float v00 = base[((size_t)sy.i0 * w + sx.i0) * ldc];
    const float v01 = base[((size_t)sy.i0 * w + sx.i1) * ldc];
    const float v10 = base[((size_t)sy.i1 * w + sx.i0) * ldc];
    const float v11 = base[((size_t)sy.i1 * w + sx.i1) * ldc];
    return sy.l0 * (sx.l0 * v00 + sx.l1 * v01) + sy.l1 * (sx.l0 * v10 + sx.l1 * v11);
}

// ---------------------------------------------------------------- upsample ----
__global__ __launch_bounds__(256) void upsample_nchw_kernel(const float* __restrict__ lr, int ldc, float* out, int N,
                                                            int C, int h, int w, int H, int W) {
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const long long total = (long long)N * H * W;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const long long t = i / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        const Src sy = src_index(y, sh, h), sx = src_index(x, sw, w);
        float* o = out + ((size_t)n * C * H + y) * W + x;
        for (int c = 0; c < C; ++c) o[(size_t)c * H * W] = interp(lr, ldc, h, w, n, sy, sx, c);
    }
}

// gather form of the transposed interpolation: one thread per low-res (n,c,yi,xi)
__global__ __launch_bounds__(256) void upsample_nchw_bwd_kernel(const float* __restrict__ dout, float* dlr, int ldc,
                                                                int N, int C, int h, int w, int H, int W) {
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const long long total = (long long)N * h * w * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int xi = (int)(t % w); t /= w;
        const int yi = (int)(t % h);
        const int n = (int)(t / h);
        // conservative full-res window that can touch (yi, xi)
        int y_lo = (int)floorf(((float)yi - 1.f + 0.5f) / sh - 0.5f) - 1;
        int y_hi = (int)ceilf(((float)yi + 1.f + 0.5f) / sh - 0.5f) + 1;
        int x_lo = (int)floorf(((float)xi - 1.f + 0.5f) / sw - 0.5f) - 1;
        int x_hi = (int)ceilf(((float)xi + 1.f + 0.5f) / sw - 0.5f) + 1;
        y_lo = max(y_lo, 0); x_lo = max(x_lo, 0);
        y_hi = min(y_hi, H - 1); x_hi = min(x_hi, W - 1);
        const float* src = dout + ((size_t)n * C + c) * H * W;
        float acc = 0.f;
        for (int y = y_lo; y <= y_hi; ++y) {
            const Src sy = src_index(y, sh, h);
            const float wy = (sy.i0 == yi ? sy.l0 : 0.f) + (sy.i1 == yi ? sy.l1 : 0.f);
            if (wy == 0.f) continue;
            float row = 0.f;
            for (int x = x_lo; x <= x_hi; ++x) {
                const Src sx = src_index(x, sw, w);
                const float wx = (sx.i0 == xi ? sx.l0 : 0.f) + (sx.i1 == xi ? sx.l1 : 0.f);
                if (wx != 0.f) row += wx * src[(size_t)y * W + x];
            }
            acc += wy * row;
        }
        dlr[(((size_t)n * h + yi) * w + xi) * ldc + c] = acc;
    }
}

// ------------------------------------------------------ cross entropy fwd ----
// accum (double[2]): [0] += sum over valid pixels of (lse - z[target]); [1] += #valid.
// One half wave per (n, y, x0): all full-res pixels of row y whose left source column is x0
// share the same four low-res logit vectors, which are loaded once per span.
__global__ __launch_bounds__(256) void upsample_ce_fwd_kernel(const float* __restrict__ lr, int ldc,
                                                              const int64_t* __restrict__ target, int N, int C, int h,
                                                              int w, int H, int W, long long ignore_index,
                                                              double* accum) {
    __shared__ float sl[8];
    __shared__ float sc[8];
    const int lane32 = threadIdx.x & 31;
    const int half = threadIdx.x >> 5;                 // 8 half waves per block
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const long long total = (long long)N * H * w;
    float loss_acc = 0.f, cnt_acc = 0.f;
    const bool active = lane32 < C;
    for (long long it = (long long)blockIdx.x * 8 + half; it < total; it += (long long)gridDim.x * 8) {
        const int x0 = (int)(it % w);
        const long long t = it / w;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        int xa = (int)floorf(((float)x0 + 0.5f) / sw - 0.5f) - 1;
        int xb = (int)ceilf(((float)x0 + 1.5f) / sw - 0.5f) + 1;
        xa = (x0 == 0) ? 0 : max(xa, 0);
        xb = min(xb, W - 1);
        const Src sy = src_index(y, sh, h);
        const int x1 = min(x0 + 1, w - 1);
        const float* base = lr + (size_t)n * h * w * ldc + lane32;
        float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
        if (active) {
            v00 = base[((size_t)sy.i0 * w + x0) * ldc];
            v01 = base[((size_t)sy.i0 * w + x1) * ldc];
            v10 = base[((size_t)sy.i1 * w + x0) * ldc];
            v11 = base[((size_t)sy.i1 * w + x1) * ldc];
        }
        const int64_t* trow = target + ((long long)n * H + y) * W;
        for (int x = xa; x <= xb; ++x) {
            const Src sx = src_index(x, sw, w);
            if (sx.i0 != x0) continue;                               // uniform over the half wave
            const long long tg = trow[x];
            if (tg == ignore_index || tg < 0 || tg >= C) continue;   // uniform
            const float z = active ? sy.l0 * (sx.l0 * v00 + sx.l1 * v01) + sy.l1 * (sx.l0 * v10 + sx.l1 * v11)
                                   : -INFINITY;
            const float m = half_max(z);
            const float e = active ? __expf(z - m) : 0.f;
            const float ssum = half_sum(e);
            const float zt = __shfl(z, (int)tg, 32);
            loss_acc += (m + __logf(ssum)) - zt;
            cnt_acc += 1.f;
        }
    }
    // every lane of a half wave carries the same partial; reduce over the 8 halves
    if (lane32 == 0) { sl[half] = loss_acc; sc[half] = cnt_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double l = 0.0, c = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { l += (double)sl[i]; c += (double)sc[i]; }
        if (c > 0.0) {
            atomicAdd(&accum[0], l);
            atomicAdd(&accum[1], c);
        }
    }
}

// ------------------------------------------------------ cross entropy bwd ----
// One half wave per (n, y, x0): the full-res pixels of row y whose left source
// column is x0 (a contiguous span) are reduced in registers into the two columns
// x0 / x0+1, then scattered to the two source rows with 128-byte float atomics.
__global__ __launch_bounds__(256) void upsample_ce_bwd_kernel(const float* __restrict__ lr, int ldc,
                                                              const int64_t* __restrict__ target, int N, int C, int h,
                                                              int w, int H, int W, long long ignore_index,
                                                              const double* __restrict__ accum, float gscale,
                                                              const float* __restrict__ gscale_dev, float* dlr) {
    const int lane32 = threadIdx.x & 31;
    const int half = threadIdx.x >> 5;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const double cnt = accum[1];
    if (!(cnt > 0.0)) return;
    const float g = (float)((double)gscale * (gscale_dev ? (double)gscale_dev[0] : 1.0) / cnt);
    const bool active = lane32 < C;
    const long long total = (long long)N * H * w;
    for (long long it = (long long)blockIdx.x * 8 + half; it < total; it += (long long)gridDim.x * 8) {
        const int x0 = (int)(it % w);
        const long long t = it / w;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        // span of x with floor(src(x)) == x0 : invert src = sw*(x+0.5)-0.5 conservatively, then test
        int xa = (int)floorf(((float)x0 + 0.5f) / sw - 0.5f) - 1;
        int xb = (int)ceilf(((float)x0 + 1.5f) / sw - 0.5f) + 1;
        xa = (x0 == 0) ? 0 : max(xa, 0);      // sources clamped at 0 all land in column 0
        xb = min(xb, W - 1);
        const Src sy = src_index(y, sh, h);
        const int xn = min(x0 + 1, w - 1);
        const float* base = lr + (size_t)n * h * w * ldc + lane32;
        float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;     // the span's four source vectors, loaded once
        if (active) {
            v00 = base[((size_t)sy.i0 * w + x0) * ldc];
            v01 = base[((size_t)sy.i0 * w + xn) * ldc];
            v10 = base[((size_t)sy.i1 * w + x0) * ldc];
            v11 = base[((size_t)sy.i1 * w + xn) * ldc];
        }
        float a0 = 0.f, a1 = 0.f;
        for (int x = xa; x <= xb; ++x) {
            const Src sx = src_index(x, sw, w);
            if (sx.i0 != x0) continue;                               // uniform over the half wave
            const long long tg = target[((long long)n * H + y) * W + x];
            if (tg == ignore_index || tg < 0 || tg >= C) continue;   // uniform
            const float z = active ? sy.l0 * (sx.l0 * v00 + sx.l1 * v01) + sy.l1 * (sx.l0 * v10 + sx.l1 * v11)
                                   : -INFINITY;
            const float m = half_max(z);
            const float e = active ? __expf(z - m) : 0.f;
            const float s = half_sum(e);
            float r = e / s;
            if (lane32 == (int)tg) r -= 1.f;
            if (sx.i1 == sx.i0) {
                a0 += r;                    // clamped right border: both taps on x0
            } else {
                a0 += sx.l0 * r;
                a1 += sx.l1 * r;
            }
        }
        if (!active) continue;
        const int x1 = min(x0 + 1, w - 1);
        float* r0 = dlr + (((size_t)n * h + sy.i0) * w) * ldc + lane32;
        float* r1 = dlr + (((size_t)n * h + sy.i1) * w) * ldc + lane32;
        const float wy0 = (sy.i1 == sy.i0) ? 1.f : sy.l0;
        const float wy1 = (sy.i1 == sy.i0) ? 0.f : sy.l1;
        if (a0 != 0.f) {
            atomicAdd(r0 + (size_t)x0 * ldc, g * wy0 * a0);
            if (wy1 != 0.f) atomicAdd(r1 + (size_t)x0 * ldc, g * wy1 * a0);
        }
        if (a1 != 0.f) {
            atomicAdd(r0 + (size_t)x1 * ldc, g * wy0 * a1);
            if (wy1 != 0.f) atomicAdd(r1 + (size_t)x1 * ldc, g * wy1 * a1);
        }
    }
}

// ------------------------------------------ cross entropy, one THREAD per span ----
// The half-wave-per-pixel kernels above spend ~2/3 of their instructions in the two 32-lane reductions of every
// pixel.  Here a thread owns one span (row y, source column x0: ~H/h pixels that share the same four low-res
// vectors): the y-interpolated vectors P = l0y*v[y0][x0] + l1y*v[y1][x0] and Q (column x0+1) live in registers,
// every pixel is z_c = l0x*P_c + l1x*Q_c and its softmax statistics are plain loops over the classes - no
// cross-lane traffic at all (5-6x fewer wave instructions).  Classes beyond C carry -1e30 (exp -> 0, never NaN).
template <int CP>
__device__ __forceinline__ void span_vectors(const float* __restrict__ lr, int ldc, int w, int C, int n, int h, const Src& sy,
                                             int x0, float (&P)[CP], float (&Q)[CP]) {
    const int x1 = min(x0 + 1, w - 1);
    const float* r0 = lr + ((size_t)n * h + sy.i0) * w * ldc;
    const float* r1 = lr + ((size_t)n * h + sy.i1) * w * ldc;
#pragma unroll
    for (int c4 = 0; c4 < CP / 4; ++c4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(r0 + (size_t)x0 * ldc + 4 * c4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(r1 + (size_t)x0 * ldc + 4 * c4);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(r0 + (size_t)x1 * ldc + 4 * c4);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(r1 + (size_t)x1 * ldc + 4 * c4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool ok = 4 * c4 + k < C;
            P[4 * c4 + k] = ok ? sy.l0 * a[k] + sy.l1 * b[k] : -1e30f;
            Q[4 * c4 + k] = ok ? sy.l0 * a1[k] + sy.l1 * b1[k] : -1e30f;
        }
    }
}

template <int CP>
__global__ __launch_bounds__(256) void upsample_ce_fwd_span_kernel(const float* __restrict__ lr, int ldc,
                                                                   const int64_t* __restrict__ target, int N, int C, int h,
                                                                   int w, int H, int W, long long ignore_index,
                                                                   double* accum) {
    __shared__ float sl[4], sc[4];
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const long long total = (long long)N * H * w;
    float loss_acc = 0.f, cnt_acc = 0.f;
    for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
        const int x0 = (int)(it % w);
        const long long t = it / w;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        int xa = (int)floorf(((float)x0 + 0.5f) / sw - 0.5f) - 1;
        int xb = (int)ceilf(((float)x0 + 1.5f) / sw - 0.5f) + 1;
        xa = (x0 == 0) ? 0 : max(xa, 0);
        xb = min(xb, W - 1);
        const Src sy = src_index(y, sh, h);
        float P[CP], Q[CP];
        span_vectors<CP>(lr, ldc, w, C, n, h, sy, x0, P, Q);
        const int64_t* trow = target + ((long long)n * H + y) * W;
        for (int x = xa; x <= xb; ++x) {
            const Src sx = src_index(x, sw, w);
            if (sx.i0 != x0) continue;
            const long long tg = trow[x];
            if (tg == ignore_index || tg < 0 || tg >= C) continue;
            float z[CP];
            float m = -INFINITY, zt = 0.f;
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                z[c] = sx.l0 * P[c] + sx.l1 * Q[c];
                m = fmaxf(m, z[c]);
                zt = (c == (int)tg) ? z[c] : zt;
            }
            float ssum = 0.f;
#pragma unroll
            for (int c = 0; c < CP; ++c) ssum += __expf(z[c] - m);
            loss_acc += (m + __logf(ssum)) - zt;
            cnt_acc += 1.f;
        }
    }
    loss_acc = wave_sum(loss_acc);
    cnt_acc = wave_sum(cnt_acc);
    if ((threadIdx.x & 63) == 0) { sl[threadIdx.x >> 6] = loss_acc; sc[threadIdx.x >> 6] = cnt_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double l = (double)sl[0] + (double)sl[1] + (double)sl[2] + (double)sl[3];
        const double c = (double)sc[0] + (double)sc[1] + (double)sc[2] + (double)sc[3];
        if (c > 0.0) {
            atomicAdd(&accum[0], l);
            atomicAdd(&accum[1], c);
        }
    }
}

// Backward: a block owns 256 consecutive spans of one image (a few output rows).  Every thread reduces its span into
// the two source columns in registers and parks the 2 x C partial sums in its own LDS slot (plain stores, 49-float
// stride = conflict free).  Then the block turns around: one thread per (source row, column, class) GATHERS the few
// slots that feed it (<= 2 per output row of the block) and issues one class-contiguous global atomic.  No LDS atomics
// (ds_add_f32 measured ~170 cycles per wave instruction here) and ~5x fewer, fully coalesced global float atomics
// than one flush per span.
template <int CP>
__global__ __launch_bounds__(256) void upsample_ce_bwd_span_kernel(const float* __restrict__ lr, int ldc,
                                                                   const int64_t* __restrict__ target, int C, int h, int w,
                                                                   int H, int W, long long ignore_index,
                                                                   const double* __restrict__ accum, float gscale,
                                                                   const float* __restrict__ gscale_dev, float* dlr,
                                                                   int tile_rows) {
    constexpr int SLOT = 2 * CP + 1;
    extern __shared__ float slots[];                        // [256][SLOT]: a0[CP], a1[CP]
    const double cnt = accum[1];
    if (!(cnt > 0.0)) return;
    const float g = (float)((double)gscale * (gscale_dev ? (double)gscale_dev[0] : 1.0) / cnt);
    const int n = blockIdx.y;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int items = H * w;
    const int first = blockIdx.x * 256;
    const int item = first + threadIdx.x;
    {
        float a0[CP], a1[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) { a0[c] = 0.f; a1[c] = 0.f; }
        if (item < items) {
            const int x0 = item % w, y = item / w;
            int xa = (int)floorf(((float)x0 + 0.5f) / sw - 0.5f) - 1;
            int xb = (int)ceilf(((float)x0 + 1.5f) / sw - 0.5f) + 1;
            xa = (x0 == 0) ? 0 : max(xa, 0);
            xb = min(xb, W - 1);
            const Src sy = src_index(y, sh, h);
            float P[CP], Q[CP];
            span_vectors<CP>(lr, ldc, w, C, n, h, sy, x0, P, Q);
            const int64_t* trow = target + ((long long)n * H + y) * W;
            for (int x = xa; x <= xb; ++x) {
                const Src sx = src_index(x, sw, w);
                if (sx.i0 != x0) continue;
                const long long tg = trow[x];
                if (tg == ignore_index || tg < 0 || tg >= C) continue;
                float e[CP];
                float m = -INFINITY;
#pragma unroll
                for (int c = 0; c < CP; ++c) {
                    e[c] = sx.l0 * P[c] + sx.l1 * Q[c];
                    m = fmaxf(m, e[c]);
                }
                float ssum = 0.f;
#pragma unroll
                for (int c = 0; c < CP; ++c) {
                    e[c] = __expf(e[c] - m);
                    ssum += e[c];
                }
                const float inv = 1.f / ssum;
                const bool clamp = sx.i1 == sx.i0;          // right border: both taps on x0
                const float w0 = clamp ? 1.f : sx.l0, w1 = clamp ? 0.f : sx.l1;
#pragma unroll
                for (int c = 0; c < CP; ++c) {
                    const float r = e[c] * inv - ((c == (int)tg) ? 1.f : 0.f);
                    a0[c] += w0 * r;
                    a1[c] += w1 * r;
                }
            }
        }
        float* mine = slots + threadIdx.x * SLOT;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            mine[c] = g * a0[c];
            mine[CP + c] = g * a1[c];
        }
    }
    __syncthreads();
    // gather: output (source row r, column x, class c) <- spans (y, x)[a0] and (y, x-1)[a1] of the block's rows y
    const int y_first = first / w;
    const int y_last = min(H - 1, (first + 255) / w);
    const int ny = y_last - y_first + 1;
    const int base_row = src_index(y_first, sh, h).i0;
    float* wtab = slots + 256 * SLOT;                       // [tile_rows][ny]: y-weight of output row y on source row r
    for (int i = threadIdx.x; i < tile_rows * ny; i += 256) {
        const int r = base_row + i / ny;
        const Src sy = src_index(y_first + i % ny, sh, h);
        wtab[i] = (sy.i0 == r ? ((sy.i1 == sy.i0) ? 1.f : sy.l0) : 0.f) + ((sy.i1 == r && sy.i1 != sy.i0) ? sy.l1 : 0.f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < tile_rows * w * 32; i += 256) {
        const int c = i & 31, x = (i >> 5) % w, rl = (i >> 5) / w;
        if (c >= C || base_row + rl >= h) continue;
        float v = 0.f;
        for (int yl = 0; yl < ny; ++yl) {
            const float wy = wtab[rl * ny + yl];
            if (wy == 0.f) continue;
            const int t0 = (y_first + yl) * w + x - first;   // span (y, x): its a0 lands on column x
            if (t0 >= 0 && t0 < 256) v += wy * slots[t0 * SLOT + c];
            if (x > 0) {                                     // span (y, x-1): its a1 lands on column x
                const int t1 = t0 - 1;
                if (t1 >= 0 && t1 < 256) v += wy * slots[t1 * SLOT + CP + c];
            }
            if (x == w - 1 && t0 >= 0 && t0 < 256) v += wy * slots[t0 * SLOT + CP + c];   // clamped right tap (always 0)
        }
        if (v != 0.f) atomicAdd(dlr + (((size_t)n * h + base_row + rl) * w + x) * ldc + c, v);
    }
}

// ------------------------------------------- per-class probability sums (region losses) ----
// Dice / Jaccard (branchy_seg_losses.py:40-78) are closed forms of three per-image, per-class sums over the
// full-resolution pixels: S_c = sum p_c, I_c = sum p_c [t = c], T_c = #[t = c] (p = softmax of the upsampled
// logits); Focal (:113-131) is a per-pixel sum F = sum -alpha_t (1 - p_t)^gamma log p_t.  One pass computes all of
// them from the low-res logits (same span scheme as the CE kernels).  grid.y = image.
// sums [N][3][CMAX] double (+=), extra [N][2] double (+=): [0] pixels with a label outside [0,C), [1] F.
// FocalLoss(alpha=...) as the reference COMPUTES it (branchy_seg_losses.py:126-129): the [B,H,W] loss map times
// alpha[targets] of shape [B,1,H,W] broadcasts to [B,B,H,W] - every image's loss at a pixel is weighted by the alphas of
// ALL images' labels at that pixel.  Summed over the extra axis that is the per-pixel weight sum_i alpha[t[i][y][x]].
__device__ __forceinline__ float focal_alpha_batch_sum(const float* __restrict__ alpha, const int64_t* __restrict__ target,
                                                       int N, int C, int H, int W, int y, int x) {
    float a = 0.f;
    for (int i = 0; i < N; ++i) {
        const long long t = target[((long long)i * H + y) * W + x];
        if (t >= 0 && t < C) a += alpha[(int)t];
    }
    return a;
}

__global__ __launch_bounds__(256) void class_sums_fwd_kernel(const float* __restrict__ lr, int ldc,
                                                             const int64_t* __restrict__ target, int C, int h, int w,
                                                             int H, int W, float gamma, const float* __restrict__ alpha,
                                                             int alpha_batch_sum, double* sums, double* extra) {
    __shared__ float red[8][4][32];
    __shared__ float red2[8][2];
    const int n = blockIdx.y;
    const int lane32 = threadIdx.x & 31;
    const int half = threadIdx.x >> 5;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int total = H * w;
    const bool active = lane32 < C;
    float aS = 0.f, aI = 0.f, aT = 0.f, aV = 0.f, aF = 0.f;
    for (int it = blockIdx.x * 8 + half; it < total; it += gridDim.x * 8) {
        const int x0 = it % w, y = it / w;
        int xa = (int)floorf(((float)x0 + 0.5f) / sw - 0.5f) - 1;
        int xb = (int)ceilf(((float)x0 + 1.5f) / sw - 0.5f) + 1;
        xa = (x0 == 0) ? 0 : max(xa, 0);
        xb = min(xb, W - 1);
        const Src sy = src_index(y, sh, h);
        const int x1 = min(x0 + 1, w - 1);
        const float* base = lr + (size_t)n * h * w * ldc + lane32;
        float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
        if (active) {
            v00 = base[((size_t)sy.i0 * w + x0) * ldc];
            v01 = base[((size_t)sy.i0 * w + x1) * ldc];
            v10 = base[((size_t)sy.i1 * w + x0) * ldc];
            v11 = base[((size_t)sy.i1 * w + x1) * ldc];
        }
        const int64_t* trow = target + ((long long)n * H + y) * W;
        for (int x = xa; x <= xb; ++x) {
            const Src sx = src_index(x, sw, w);
            if (sx.i0 != x0) continue;
            const long long tg = trow[x];
            const bool labelled = tg >= 0 && tg < C;
            const float z = active ? sy.l0 * (sx.l0 * v00 + sx.l1 * v01) + sy.l1 * (sx.l0 * v10 + sx.l1 * v11)
                                   : -INFINITY;
            const float m = half_max(z);
            const float e = active ? __expf(z - m) : 0.f;
            const float ssum = half_sum(e);
            const float pr = e / ssum;
            aS += pr;
            if (labelled) {
                const bool mine = lane32 == (int)tg;
                aI += mine ? pr : 0.f;
                aT += mine ? 1.f : 0.f;
                if (gamma >= 0.f) {                                   // focal term (uniform over the half wave)
                    const float zt = __shfl(z, (int)tg, 32);
                    const float logq = zt - (m + __logf(ssum));
                    const float q = __expf(logq);
                    const float a = alpha ? (alpha_batch_sum ? focal_alpha_batch_sum(alpha, target, gridDim.y, C, H, W, y, x)
                                                             : alpha[(int)tg]) : 1.f;
                    aF += -a * __powf(fmaxf(1.f - q, 0.f), gamma) * logq;
                }
            } else {
                aV += 1.f;
            }
        }
    }
    red[half][0][lane32] = aS; red[half][1][lane32] = aI; red[half][2][lane32] = aT;
    if (lane32 == 0) { red2[half][0] = aV; red2[half][1] = aF; }
    __syncthreads();
    if (threadIdx.x < 96) {
        const int k = threadIdx.x >> 5, c = threadIdx.x & 31;
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += (double)red[i][k][c];
        if (c < C && t != 0.0) atomicAdd(&sums[((size_t)n * 3 + k) * CMAX + c], t);
    } else if (threadIdx.x < 98) {
        const int k = threadIdx.x - 96;
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) t += (double)red2[i][k];
        if (t != 0.0) atomicAdd(&extra[n * 2 + k], t);
    }
}

// Backward of the above: dL/dz_c = p_c (g_c - sum_k p_k g_k), g_c = gS[n][c] + [t = c] gI[n][c], plus the focal
// term gF a_t [gamma (1-q)^(gamma-1) q log q - (1-q)^gamma] ([c = t] - p_c); transposed interpolation as in the CE
// backward (span reduction in registers, 128-byte float atomics).
__global__ __launch_bounds__(256) void class_sums_bwd_kernel(const float* __restrict__ lr, int ldc,
                                                             const int64_t* __restrict__ target, int C, int h, int w,
                                                             int H, int W, const float* __restrict__ gS,
                                                             const float* __restrict__ gI, const float* __restrict__ gF,
                                                             float gamma, const float* __restrict__ alpha,
                                                             int alpha_batch_sum, float* dlr) {
    const int n = blockIdx.y;
    const int lane32 = threadIdx.x & 31;
    const int half = threadIdx.x >> 5;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const bool active = lane32 < C;
    const float gs = (gS && active) ? gS[n * CMAX + lane32] : 0.f;
    const float gi = (gI && active) ? gI[n * CMAX + lane32] : 0.f;
    const float gf = gF ? gF[0] : 0.f;
    const int total = H * w;
    for (int it = blockIdx.x * 8 + half; it < total; it += gridDim.x * 8) {
        const int x0 = it % w, y = it / w;
        int xa = (int)floorf(((float)x0 + 0.5f) / sw - 0.5f) - 1;
        int xb = (int)ceilf(((float)x0 + 1.5f) / sw - 0.5f) + 1;
        xa = (x0 == 0) ? 0 : max(xa, 0);
        xb = min(xb, W - 1);
        const Src sy = src_index(y, sh, h);
        const int xn = min(x0 + 1, w - 1);
        const float* base = lr + (size_t)n * h * w * ldc + lane32;
        float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
        if (active) {
            v00 = base[((size_t)sy.i0 * w + x0) * ldc];
            v01 = base[((size_t)sy.i0 * w + xn) * ldc];
            v10 = base[((size_t)sy.i1 * w + x0) * ldc];
            v11 = base[((size_t)sy.i1 * w + xn) * ldc];
        }
        float a0 = 0.f, a1 = 0.f;
        for (int x = xa; x <= xb; ++x) {
            const Src sx = src_index(x, sw, w);
            if (sx.i0 != x0) continue;
            const long long tg = target[((long long)n * H + y) * W + x];
            const bool labelled = tg >= 0 && tg < C;
            const float z = active ? sy.l0 * (sx.l0 * v00 + sx.l1 * v01) + sy.l1 * (sx.l0 * v10 + sx.l1 * v11)
                                   : -INFINITY;
            const float m = half_max(z);
            const float e = active ? __expf(z - m) : 0.f;
            const float s = half_sum(e);
            const float pr = e / s;
            const bool mine = labelled && lane32 == (int)tg;
            const float g = gs + (mine ? gi : 0.f);
            const float dot = half_sum(pr * g);
            float r = pr * (g - dot);
            if (gf != 0.f && labelled) {
                const float zt = __shfl(z, (int)tg, 32);
                const float logq = zt - (m + __logf(s));
                const float q = __expf(logq);
                const float omq = fmaxf(1.f - q, 0.f);
                const float a = alpha ? (alpha_batch_sum ? focal_alpha_batch_sum(alpha, target, gridDim.y, C, H, W, y, x)
                                                         : alpha[(int)tg]) : 1.f;
                const float k = gamma == 0.f ? -1.f : gamma * __powf(omq, gamma - 1.f) * q * logq - __powf(omq, gamma);
                r += gf * a * k * ((mine ? 1.f : 0.f) - pr);
            }
            if (sx.i1 == sx.i0) {
                a0 += r;
            } else {
                a0 += sx.l0 * r;
                a1 += sx.l1 * r;
            }
        }
        if (!active) continue;
        const int x1 = min(x0 + 1, w - 1);
        float* r0 = dlr + (((size_t)n * h + sy.i0) * w) * ldc + lane32;
        float* r1 = dlr + (((size_t)n * h + sy.i1) * w) * ldc + lane32;
        const float wy0 = (sy.i1 == sy.i0) ? 1.f : sy.l0;
        const float wy1 = (sy.i1 == sy.i0) ? 0.f : sy.l1;
        if (a0 != 0.f) {
            atomicAdd(r0 + (size_t)x0 * ldc, wy0 * a0);
            if (wy1 != 0.f) atomicAdd(r1 + (size_t)x0 * ldc, wy1 * a0);
        }
        if (a1 != 0.f) {
            atomicAdd(r0 + (size_t)x1 * ldc, wy0 * a1);
            if (wy1 != 0.f) atomicAdd(r1 + (size_t)x1 * ldc, wy1 * a1);
        }
    }
}

// ------------------------------------------- focal loss as a per-pixel MAP (reduction = 'none') ----
// FocalLoss._compute_loss (branchy_seg_losses.py:113-131) returns the [N,H,W] map  -(1 - p_t)^gamma log p_t  and BrSegLoss.forward
// (:24-38) hands it back unreduced for any reduction other than 'mean' / 'sum'.  Same span scheme and arithmetic as the focal term of
// class_sums_fwd_kernel, written out instead of summed.  alpha (:126-129): mode 1 = the pixel's own alpha[t]; mode 2 = the reference's
// broadcast of the [N,H,W] map against alpha[targets] of shape [N,1,H,W]: out[i][j][y][x] = loss[j][y][x] * alpha[t[i][y][x]],
// an [N,N,H,W] tensor.  Pixels labelled outside [0,C) are counted in *void_count (the reference's gather fails on them) and get 0.
__global__ __launch_bounds__(256) void focal_map_fwd_kernel(const float* __restrict__ lr, int ldc, const int64_t* __restrict__ target,
                                                            int C, int h, int w, int H, int W, float gamma,
                                                            const float* __restrict__ alpha, int alpha_mode, float* __restrict__ out,
                                                            int* void_count) {
    const int n = blockIdx.y, N = gridDim.y;
    const int lane32 = threadIdx.x & 31;
    const int half = threadIdx.x >> 5;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int total = H * w;
    const bool active = lane32 < C;
    int voids = 0;
    for (int it = blockIdx.x * 8 + half; it < total; it += gridDim.x * 8) {
        const int x0 = it % w, y = it / w;
        int xa = (int)floorf(((float)x0 + 0.5f) / sw - 0.5f) - 1;
        int xb = (int)ceilf(((float)x0 + 1.5f) / sw - 0.5f) + 1;
        xa = (x0 == 0) ? 0 : max(xa, 0);
        xb = min(xb, W - 1);
        const Src sy = src_index(y, sh, h);
        const int x1 = min(x0 + 1, w - 1);
        const float* base = lr + (size_t)n * h * w * ldc + lane32;
        float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
        if (active) {
            v00 = base[((size_t)sy.i0 * w + x0) * ldc];
            v01 = base[((size_t)sy.i0 * w + x1) * ldc];
            v10 = base[((size_t)sy.i1 * w + x0) * ldc];
            v11 = base[((size_t)sy.i1 * w + x1) * ldc];
        }
        const int64_t* trow = target + ((long long)n * H + y) * W;
        for (int x = xa; x <= xb; ++x) {
            const Src sx = src_index(x, sw, w);
            if (sx.i0 != x0) continue;
            const long long tg = trow[x];
            const bool labelled = tg >= 0 && tg < C;
            const float z = active ? sy.l0 * (sx.l0 * v00 + sx.l1 * v01) + sy.l1 * (sx.l0 * v10 + sx.l1 * v11)
                                   : -INFINITY;
            const float m = half_max(z);
            const float e = active ? __expf(z - m) : 0.f;
            const float ssum = half_sum(e);
            float loss = 0.f;
            if (labelled) {
                const float zt = __shfl(z, (int)tg, 32);
                const float logq = zt - (m + __logf(ssum));
                const float q = __expf(logq);
                loss = -__powf(fmaxf(1.f - q, 0.f), gamma) * logq;
            } else if (lane32 == 0) {
                ++voids;
            }
            if (alpha_mode == 2) {                            // [N(i)][N(n)][H][W]: lane i writes row i
                for (int i = lane32; i < N; i += 32) {
                    const long long ti = target[((long long)i * H + y) * W + x];
                    out[(((size_t)i * N + n) * H + y) * W + x] = (ti >= 0 && ti < C) ? loss * alpha[(int)ti] : 0.f;
                }
            } else if (lane32 == 0) {
                out[((size_t)n * H + y) * W + x] = (alpha_mode == 1 && labelled) ? loss * alpha[(int)tg] : loss;
            }
        }
    }
    if (voids) atomicAdd(void_count, voids);
}

// dlr += d(sum dmap * map) / d(lr): per pixel gf = dmap[n][y][x] (* alpha[t]) or, mode 2, sum_i dmap[i][n][y][x] * alpha[t[i][y][x]];
// then the focal term of class_sums_bwd_kernel with that factor.
__global__ __launch_bounds__(256) void focal_map_bwd_kernel(const float* __restrict__ lr, int ldc, const int64_t* __restrict__ target,
                                                            int C, int h, int w, int H, int W, float gamma,
                                                            const float* __restrict__ alpha, int alpha_mode,
                                                            const float* __restrict__ dmap, float* dlr) {
    const int n = blockIdx.y, N = gridDim.y;
    const int lane32 = threadIdx.x & 31;
    const int half = threadIdx.x >> 5;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const bool active = lane32 < C;
    const int total = H * w;
    for (int it = blockIdx.x * 8 + half; it < total; it += gridDim.x * 8) {
        const int x0 = it % w, y = it / w;
        int xa = (int)floorf(((float)x0 + 0.5f) / sw - 0.5f) - 1;
        int xb = (int)ceilf(((float)x0 + 1.5f) / sw - 0.5f) + 1;
        xa = (x0 == 0) ? 0 : max(xa, 0);
        xb = min(xb, W - 1);
        const Src sy = src_index(y, sh, h);
        const int xn = min(x0 + 1, w - 1);
        const float* base = lr + (size_t)n * h * w * ldc + lane32;
        float v00 = 0.f, v01 = 0.f, v10 = 0.f, v11 = 0.f;
        if (active) {
            v00 = base[((size_t)sy.i0 * w + x0) * ldc];
            v01 = base[((size_t)sy.i0 * w + xn) * ldc];
            v10 = base[((size_t)sy.i1 * w + x0) * ldc];
            v11 = base[((size_t)sy.i1 * w + xn) * ldc];
        }
        float a0 = 0.f, a1 = 0.f;
        for (int x = xa; x <= xb; ++x) {
            const Src sx = src_index(x, sw, w);
            if (sx.i0 != x0) continue;
            const long long tg = target[((long long)n * H + y) * W + x];
            if (!(tg >= 0 && tg < C)) continue;                // (uniform over the half wave)
            float gf;
            if (alpha_mode == 2) {
                gf = 0.f;
                for (int i = 0; i < N; ++i) {
                    const long long ti = target[((long long)i * H + y) * W + x];
                    if (ti >= 0 && ti < C) gf += dmap[(((size_t)i * N + n) * H + y) * W + x] * alpha[(int)ti];
                }
            } else {
                gf = dmap[((size_t)n * H + y) * W + x] * (alpha_mode == 1 ? alpha[(int)tg] : 1.f);
            }
            const float z = active ? sy.l0 * (sx.l0 * v00 + sx.l1 * v01) + sy.l1 * (sx.l0 * v10 + sx.l1 * v11)
                                   : -INFINITY;
            const float m = half_max(z);
            const float e = active ? __expf(z - m) : 0.f;
            const float s = half_sum(e);
            const float pr = e / s;
            const bool mine = lane32 == (int)tg;
            const float zt = __shfl(z, (int)tg, 32);
            const float logq = zt - (m + __logf(s));
            const float q = __expf(logq);
            const float omq = fmaxf(1.f - q, 0.f);
            const float k = gamma == 0.f ? -1.f : gamma * __powf(omq, gamma - 1.f) * q * logq - __powf(omq, gamma);
            const float r = gf * k * ((mine ? 1.f : 0.f) - pr);
            if (sx.i1 == sx.i0) {
                a0 += r;
            } else {
                a0 += sx.l0 * r;
                a1 += sx.l1 * r;
            }
        }
        if (!active) continue;
        const int x1 = min(x0 + 1, w - 1);
        float* r0 = dlr + (((size_t)n * h + sy.i0) * w) * ldc + lane32;
        float* r1 = dlr + (((size_t)n * h + sy.i1) * w) * ldc + lane32;
        const float wy0 = (sy.i1 == sy.i0) ? 1.f : sy.l0;
        const float wy1 = (sy.i1 == sy.i0) ? 0.f : sy.l1;
        if (a0 != 0.f) {
            atomicAdd(r0 + (size_t)x0 * ldc, wy0 * a0);
            if (wy1 != 0.f) atomicAdd(r1 + (size_t)x0 * ldc, wy1 * a0);
        }
        if (a1 != 0.f) {
            atomicAdd(r0 + (size_t)x1 * ldc, wy0 * a1);
            if (wy1 != 0.f) atomicAdd(r1 + (size_t)x1 * ldc, wy1 * a1);
        }
    }
}

// ------------------------------------------------- argmax + TP/FP/FN counts ----
__global__ __launch_bounds__(256) void argmax_confusion_kernel(const float* __restrict__ lr, int ldc,
                                                               const int64_t* __restrict__ target, int N, int C, int h,
                                                               int w, int H, int W, int* counts, int64_t* pred) {
    __shared__ int sc[3 * CMAX];
    for (int i = threadIdx.x; i < 3 * CMAX; i += blockDim.x) sc[i] = 0;
    __syncthreads();
    const int lane32 = threadIdx.x & 31;
    const int half = threadIdx.x >> 5;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const long long total = (long long)N * H * W;
    const bool active = lane32 < C;
    for (long long p = (long long)blockIdx.x * 8 + half; p < total; p += (long long)gridDim.x * 8) {
        const int x = (int)(p % W);
        const long long t = p / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        const Src sy = src_index(y, sh, h), sx = src_index(x, sw, w);
        float z = active ? interp(lr, ldc, h, w, n, sy, sx, lane32) : -INFINITY;
        int idx = lane32;
        // first maximum wins (torch.argmax): on ties keep the smaller index
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
            const float oz = __shfl_xor(z, o);
            const int oi = __shfl_xor(idx, o);
            if (oz > z || (oz == z && oi < idx)) { z = oz; idx = oi; }
        }
        if (lane32 == 0) {
            if (pred) pred[p] = idx;
            if (target) {
                const long long tg = target[p];
                if (tg >= 0 && tg < C) {
                    if (tg == idx) atomicAdd(&sc[idx], 1);
                    else { atomicAdd(&sc[CMAX + idx], 1); atomicAdd(&sc[2 * CMAX + (int)tg], 1); }
                } else {
                    atomicAdd(&sc[CMAX + idx], 1);     // void pixel counts as FP of the prediction (B-9)
                }
            }
        }
    }
    __syncthreads();
    if (counts)
        for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) {
            const int k = i / C, c = i - k * C;
            const int v = sc[k * CMAX + c];
            if (v) atomicAdd(&counts[k * C + c], v);
        }
}

// ---------------------------------------- argmax of two exits -> contingency table ----
// Per image n: hist[n][a][b] = number of pixels where exit A predicts class a and exit B class b (both upsampled
// to H x W on the fly).  Everything the similarity gates need (MSE / NMI / variation of information between the
// label maps of consecutive exits, sim_metrics.py + eval_br_sim.py:41-48) is a function of this C x C table.
__global__ __launch_bounds__(256) void argmax_pair_hist_kernel(const float* __restrict__ lra, const float* __restrict__ lrb,
                                                               int ldc, int C, int h, int w, int H, int W, int* hist) {
    __shared__ int sh_[CMAX * CMAX];
    for (int i = threadIdx.x; i < CMAX * CMAX; i += blockDim.x) sh_[i] = 0;
    __syncthreads();
    const int n = blockIdx.y;
    const int lane32 = threadIdx.x & 31;
    const int half = threadIdx.x >> 5;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int total = H * W;
    const bool active = lane32 < C;
    for (int p = blockIdx.x * 8 + half; p < total; p += gridDim.x * 8) {
        const int x = p % W, y = p / W;
        const Src sy = src_index(y, sh, h), sx = src_index(x, sw, w);
        float za = active ? interp(lra, ldc, h, w, n, sy, sx, lane32) : -INFINITY;
        float zb = active ? interp(lrb, ldc, h, w, n, sy, sx, lane32) : -INFINITY;
        int ia = lane32, ib = lane32;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {            // first maximum wins (torch.argmax)
            const float oa = __shfl_xor(za, o), ob = __shfl_xor(zb, o);
            const int ja = __shfl_xor(ia, o), jb = __shfl_xor(ib, o);
            if (oa > za || (oa == za && ja < ia)) { za = oa; ia = ja; }
            if (ob > zb || (ob == zb && jb < ib)) { zb = ob; ib = jb; }
        }
        if (lane32 == 0) atomicAdd(&sh_[ia * CMAX + ib], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
        const int a = i / C, b = i - a * C;
        const int v = sh_[a * CMAX + b];
        if (v) atomicAdd(&hist[((size_t)n * C + a) * C + b], v);
    }
}

// ---------------------------------------------------------- entropy gate ----
__global__ __launch_bounds__(256) void entropy_map_kernel(const float* __restrict__ lr, int ldc, int N, int C, int h,
                                                          int w, int H, int W, float* emap, const int* n_active) {
    const int lane32 = threadIdx.x & 31;
    const int half = threadIdx.x >> 5;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const float inv_logc = 1.f / logf((float)C);
    if (n_active != nullptr && *n_active < N) N = *n_active;      // progressive inference: leading slots only
    const long long total = (long long)N * H * W;
    const bool active = lane32 < C;
    for (long long p = (long long)blockIdx.x * 8 + half; p < total; p += (long long)gridDim.x * 8) {
        const int x = (int)(p % W);
        const long long t = p / W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        const Src sy = src_index(y, sh, h), sx = src_index(x, sw, w);
        const float z = active ? interp(lr, ldc, h, w, n, sy, sx, lane32) : -INFINITY;
        const float m = half_max(z);
        const float e = active ? expf(z - m) : 0.f;
        const float s = half_sum(e);
        // H = -sum p ln p = ln(s) - sum e*(z-m)/s      (p = e/s)
        const float ez = active ? e * (z - m) : 0.f;
        const float sez = half_sum(ez);
        if (lane32 == 0) emap[p] = (logf(s) - sez / s) * inv_logc;
    }
}

// one block per image: mean of the entropy map, or of its s x s max/min block pool
// (zero padded to a multiple of s, skimage.block_reduce semantics).
__global__ __launch_bounds__(1024) void entropy_reduce_kernel(const float* __restrict__ emap, int H, int W, int pool,
                                                              int s, float tau, float* ent_out, int* flag_out,
                                                              const int* n_active, int less_than) {
    __shared__ double sred[16];
    const int n = blockIdx.x;
    if (n_active != nullptr && n >= *n_active) {            // slot not in flight: no decision
        if (threadIdx.x == 0) {
            ent_out[n] = 0.f;
            if (flag_out) flag_out[n] = 0;
        }
        return;
    }
    const float* e = emap + (size_t)n * H * W;
    double acc = 0.0;
    long long count;
    if (pool == 0 || s <= 1) {
        count = (long long)H * W;
        for (long long i = threadIdx.x; i < count; i += blockDim.x) acc += (double)e[i];
    } else {
        const int Hb = (H + s - 1) / s, Wb = (W + s - 1) / s;
        count = (long long)Hb * Wb;
        for (long long b = threadIdx.x; b < count; b += blockDim.x) {
            const int by = (int)(b / Wb), bx = (int)(b % Wb);
            float v = (pool == 1) ? -INFINITY : INFINITY;
            for (int dy = 0; dy < s; ++dy)
                for (int dx = 0; dx < s; ++dx) {
                    const int y = by * s + dy, x = bx * s + dx;
                    const float f = (y < H && x < W) ? e[(size_t)y * W + x] : 0.f;   // zero padding
                    v = (pool == 1) ? fmaxf(v, f) : fminf(v, f);
                }
            acc += (double)v;
        }
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sred[i];
        const float ent = (float)(t / (double)count);
        ent_out[n] = ent;
        if (flag_out) flag_out[n] = ((ent < tau) == (less_than != 0)) ? 1 : 0;
    }
}

inline int px_grid(long long items) {   // 8 half waves per 256-thread block
    long long b = (items + 7) / 8;
    if (b > 256 * 32) b = 256 * 32;
    if (b < 1) b = 1;
    return (int)b;
}

inline int check_lr(const char* name, const float* lr, int ldc, int N, int C, int h, int w, int H, int W) {
    EESEG_CHECK(lr != nullptr, EESEG_ERR_ARG, "%s: null logits", name);
    EESEG_CHECK(N > 0 && h > 0 && w > 0 && H > 0 && W > 0, EESEG_ERR_ARG, "%s: non-positive shape", name);
    EESEG_CHECK(C >= 1 && C <= CMAX && ldc >= C, EESEG_ERR_ARG, "%s: C=%d (max %d), ldc=%d", name, C, CMAX, ldc);
    return EESEG_OK;
}
#define CHECK_LR(name)                                              \
    do {                                                            \
        int rc_ = check_lr(name, logits_lr, ldc, N, C, h, w, H, W); \
        if (rc_) return rc_;                                        \
    } while (0)

}  // namespace

extern "C" int eeseg_upsample_bilinear_nchw(const float* logits_lr, int ldc, float* out, int N, int C, int h, int w,
                                            int H, int W, void* stream) {
    CHECK_LR("upsample_bilinear_nchw");
    EESEG_CHECK(out != nullptr, EESEG_ERR_ARG, "upsample_bilinear_nchw: null output");
    long long b = ((long long)N * H * W + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    hipLaunchKernelGGL(upsample_nchw_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, logits_lr, ldc, out, N,
                       C, h, w, H, W);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_upsample_bilinear_nchw_bwd(const float* dout, float* dlogits_lr, int ldc, int N, int C, int h,
                                                int w, int H, int W, void* stream) {
    EESEG_CHECK(dout && dlogits_lr && N > 0 && C > 0 && ldc >= C && h > 0 && w > 0 && H > 0 && W > 0, EESEG_ERR_ARG,
                "upsample_bilinear_nchw_bwd: bad argument");
    long long b = ((long long)N * h * w * C + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    hipLaunchKernelGGL(upsample_nchw_bwd_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, dout, dlogits_lr,
                       ldc, N, C, h, w, H, W);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_upsample_ce_fwd(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w,
                                     int H, int W, int64_t ignore_index, double* accum, void* stream) {
    CHECK_LR("upsample_ce_fwd");
    EESEG_CHECK(target && accum && ((uintptr_t)accum & 7) == 0, EESEG_ERR_ARG, "upsample_ce_fwd: bad target/accum");
    if (g_ce_span && ldc % 4 == 0 && ldc >= ((C + 3) & ~3) && (long long)N * H * w < (1ll << 31)) {   // thread-per-span form
        const long long items = (long long)N * H * w;
        const unsigned grid = (unsigned)((items + 255) / 256 > 8192 ? 8192 : (items + 255) / 256);
        hipStream_t st = (hipStream_t)stream;
        if (C <= 8)
            hipLaunchKernelGGL(upsample_ce_fwd_span_kernel<8>, dim3(grid), dim3(256), 0, st, logits_lr, ldc, target, N, C, h, w, H, W, (long long)ignore_index, accum);
        else if (C <= 16)
            hipLaunchKernelGGL(upsample_ce_fwd_span_kernel<16>, dim3(grid), dim3(256), 0, st, logits_lr, ldc, target, N, C, h, w, H, W, (long long)ignore_index, accum);
        else if (C <= 24)
            hipLaunchKernelGGL(upsample_ce_fwd_span_kernel<24>, dim3(grid), dim3(256), 0, st, logits_lr, ldc, target, N, C, h, w, H, W, (long long)ignore_index, accum);
        else
            hipLaunchKernelGGL(upsample_ce_fwd_span_kernel<32>, dim3(grid), dim3(256), 0, st, logits_lr, ldc, target, N, C, h, w, H, W, (long long)ignore_index, accum);
        EESEG_LAUNCH_CHECK();
        return EESEG_OK;
    }
    hipLaunchKernelGGL(upsample_ce_fwd_kernel, dim3(px_grid((long long)N * H * w)), dim3(256), 0, (hipStream_t)stream,
                       logits_lr, ldc, target, N, C, h, w, H, W, (long long)ignore_index, accum);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_upsample_ce_bwd(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w,
                                     int H, int W, int64_t ignore_index, const double* accum, float gscale,
                                     const float* gscale_dev, float* dlogits_lr, void* stream) {
    CHECK_LR("upsample_ce_bwd");
    EESEG_CHECK(target && accum && dlogits_lr, EESEG_ERR_ARG, "upsample_ce_bwd: null pointer");
    {
        // source rows a block of 256 consecutive spans can touch: its output rows (256/w + 2) scaled by h/H, + 2 for
        // the second tap and the rounding of both ends
        const int out_rows = 256 / w + 2;
        const int tile_rows = (int)((double)out_rows * (double)h / (double)H) + 3;
        const int cp = C <= 8 ? 8 : (C <= 16 ? 16 : (C <= 24 ? 24 : 32));
        const size_t lds = ((size_t)256 * (2 * cp + 1) + (size_t)tile_rows * out_rows) * sizeof(float);
        if (g_ce_span && ldc % 4 == 0 && ldc >= ((C + 3) & ~3) && lds <= 64 * 1024 && tile_rows * w <= 4096) {
            const dim3 grid((unsigned)(((long long)H * w + 255) / 256), N);
            hipStream_t st = (hipStream_t)stream;
#define EESEG_CE_BWD(CPV)                                                                                                  \
    hipLaunchKernelGGL(upsample_ce_bwd_span_kernel<CPV>, grid, dim3(256), lds, st, logits_lr, ldc, target, C, h, w, H, W,   \
                       (long long)ignore_index, accum, gscale, gscale_dev, dlogits_lr, tile_rows)
            if (C <= 8) EESEG_CE_BWD(8);
            else if (C <= 16) EESEG_CE_BWD(16);
            else if (C <= 24) EESEG_CE_BWD(24);
            else EESEG_CE_BWD(32);
#undef EESEG_CE_BWD
            EESEG_LAUNCH_CHECK();
            return EESEG_OK;
        }
    }
    hipLaunchKernelGGL(upsample_ce_bwd_kernel, dim3(px_grid((long long)N * H * w)), dim3(256), 0, (hipStream_t)stream,
                       logits_lr, ldc, target, N, C, h, w, H, W, (long long)ignore_index, accum, gscale, gscale_dev,
                       dlogits_lr);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_argmax_confusion(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h,
                                      int w, int H, int W, int32_t* counts, int64_t* pred, void* stream) {
    CHECK_LR("argmax_confusion");
    EESEG_CHECK((counts && target) || pred, EESEG_ERR_ARG, "argmax_confusion: nothing to compute");
    hipLaunchKernelGGL(argmax_confusion_kernel, dim3(px_grid((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream,
                       logits_lr, ldc, target, N, C, h, w, H, W, (int*)counts, pred);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_class_sums_fwd(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w, int H,
                                    int W, float gamma, const float* alpha, int alpha_batch_sum, double* sums, double* extra,
                                    void* stream) {
    CHECK_LR("class_sums_fwd");
    EESEG_CHECK(target && sums && extra, EESEG_ERR_ARG, "class_sums_fwd: null pointer");
    long long blocks = ((long long)H * w + 8 * 8 - 1) / (8 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(class_sums_fwd_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, logits_lr, ldc,
                       target, C, h, w, H, W, gamma, alpha, alpha_batch_sum, sums, extra);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_class_sums_bwd(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w, int H,
                                    int W, const float* gS, const float* gI, const float* gF, float gamma,
                                    const float* alpha, int alpha_batch_sum, float* dlogits_lr, void* stream) {
    CHECK_LR("class_sums_bwd");
    EESEG_CHECK(target && dlogits_lr && (gS || gI || gF), EESEG_ERR_ARG, "class_sums_bwd: null pointer");
    long long blocks = ((long long)H * w + 8 * 8 - 1) / (8 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(class_sums_bwd_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, logits_lr, ldc,
                       target, C, h, w, H, W, gS, gI, gF, gamma, alpha, alpha_batch_sum, dlogits_lr);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_focal_map_fwd(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w, int H, int W,
                                  float gamma, const float* alpha, int alpha_mode, float* out, int32_t* void_count, void* stream) {
    CHECK_LR("focal_map_fwd");
    EESEG_CHECK(target && out && void_count, EESEG_ERR_ARG, "focal_map_fwd: null pointer");
    EESEG_CHECK(gamma >= 0.f && alpha_mode >= 0 && alpha_mode <= 2 && (alpha_mode == 0 || alpha), EESEG_ERR_ARG,
                "focal_map_fwd: gamma >= 0, alpha_mode 0 (none) / 1 (own label) / 2 (reference broadcast, needs alpha)");
    long long blocks = ((long long)H * w + 8 * 8 - 1) / (8 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(focal_map_fwd_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, logits_lr, ldc, target,
                       C, h, w, H, W, gamma, alpha, alpha_mode, out, (int*)void_count);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_focal_map_bwd(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w, int H, int W,
                                  float gamma, const float* alpha, int alpha_mode, const float* dmap, float* dlogits_lr, void* stream) {
    CHECK_LR("focal_map_bwd");
    EESEG_CHECK(target && dmap && dlogits_lr, EESEG_ERR_ARG, "focal_map_bwd: null pointer");
    EESEG_CHECK(gamma >= 0.f && alpha_mode >= 0 && alpha_mode <= 2 && (alpha_mode == 0 || alpha), EESEG_ERR_ARG,
                "focal_map_bwd: gamma >= 0, alpha_mode 0 / 1 / 2 (1, 2 need alpha)");
    long long blocks = ((long long)H * w + 8 * 8 - 1) / (8 * 8);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(focal_map_bwd_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, logits_lr, ldc, target,
                       C, h, w, H, W, gamma, alpha, alpha_mode, dmap, dlogits_lr);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_argmax_pair_hist(const float* logits_a, const float* logits_b, int ldc, int N, int C, int h, int w,
                                      int H, int W, int32_t* hist, void* stream) {
    const float* logits_lr = logits_a;
    CHECK_LR("argmax_pair_hist");
    EESEG_CHECK(logits_b && hist && ((uintptr_t)logits_b & 15) == 0, EESEG_ERR_ARG, "argmax_pair_hist: bad pointer");
    long long blocks = ((long long)H * W + 8 * 16 - 1) / (8 * 16);      // >= 16 pixels per half wave
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(argmax_pair_hist_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, logits_a,
                       logits_b, ldc, C, h, w, H, W, (int*)hist);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

// ------------------------------------------------------------------ SSIM ----
// Structural similarity of two integer LABEL maps (sim_metrics.py:15-37 -> skimage.metrics.structural_similarity with
// its defaults: 7x7 uniform window, sample covariance, K1 = 0.01, K2 = 0.03, mean over the image cropped by 3 pixels
// per side, float64 arithmetic).  Window sums of a, b, a*a, b*b, a*b are exact integers; one thread per output pixel,
// the (16+6)^2 label patch of both maps staged in LDS; per-image sums of S reduced in double.
#define SSIM_T 16
#define SSIM_W 7
__global__ __launch_bounds__(SSIM_T* SSIM_T) void ssim_labels_kernel(const long long* __restrict__ a,
                                                                      const long long* __restrict__ b, int H, int W,
                                                                      double c1, double c2, double* sums) {
    __shared__ int pa[SSIM_T + SSIM_W - 1][SSIM_T + SSIM_W];
    __shared__ int pb[SSIM_T + SSIM_W - 1][SSIM_T + SSIM_W];
    __shared__ double red[SSIM_T * SSIM_T / 64];
    const int n = blockIdx.z;
    const int Ho = H - (SSIM_W - 1), Wo = W - (SSIM_W - 1);       // interior outputs (the crop)
    const int oy0 = blockIdx.y * SSIM_T, ox0 = blockIdx.x * SSIM_T;
    const long long* an = a + (size_t)n * H * W;
    const long long* bn = b + (size_t)n * H * W;
    const int P = SSIM_T + SSIM_W - 1;
    for (int i = threadIdx.x; i < P * P; i += SSIM_T * SSIM_T) {
        const int r = i / P, c = i - r * P;
        const int y = oy0 + r, x = ox0 + c;
        const bool in = y < H && x < W;
        pa[r][c] = in ? (int)an[(size_t)y * W + x] : 0;
        pb[r][c] = in ? (int)bn[(size_t)y * W + x] : 0;
    }
    __syncthreads();
    const int ty = threadIdx.x / SSIM_T, tx = threadIdx.x - ty * SSIM_T;
    double s = 0.0;
    if (oy0 + ty < Ho && ox0 + tx < Wo) {
        int sa = 0, sb = 0, saa = 0, sbb = 0, sab = 0;
#pragma unroll
        for (int r = 0; r < SSIM_W; ++r)
#pragma unroll
            for (int c = 0; c < SSIM_W; ++c) {
                const int va = pa[ty + r][tx + c], vb = pb[ty + r][tx + c];
                sa += va; sb += vb; saa += va * va; sbb += vb * vb; sab += va * vb;
            }
        const double np_ = (double)(SSIM_W * SSIM_W), cov = np_ / (np_ - 1.0);
        const double ux = sa / np_, uy = sb / np_;
        const double vx = cov * (saa / np_ - ux * ux), vy = cov * (sbb / np_ - uy * uy), vxy = cov * (sab / np_ - ux * uy);
        s = ((2.0 * ux * uy + c1) * (2.0 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2));
    }
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < SSIM_T * SSIM_T / 64; ++i) t += red[i];
        atomicAdd(&sums[n], t);
    }
}

__global__ void ssim_finish_kernel(double* sums, int N, double inv_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) sums[i] *= inv_count;
}

extern "C" int eeseg_ssim_labels(const int64_t* labels_a, const int64_t* labels_b, int N, int H, int W, double data_range,
                                 double* ssim_out, void* stream) {
    EESEG_CHECK(labels_a && labels_b && ssim_out, EESEG_ERR_ARG, "ssim_labels: null pointer");
    EESEG_CHECK(N >= 1 && H >= SSIM_W && W >= SSIM_W, EESEG_ERR_ARG, "ssim_labels: image smaller than the 7x7 window");
    EESEG_CHECK(data_range > 0, EESEG_ERR_ARG, "ssim_labels: data_range must be positive");
    hipStream_t st = (hipStream_t)stream;
    EESEG_HIP(hipMemsetAsync(ssim_out, 0, sizeof(double) * N, st));
    const int Ho = H - (SSIM_W - 1), Wo = W - (SSIM_W - 1);
    const double c1 = (0.01 * data_range) * (0.01 * data_range), c2 = (0.03 * data_range) * (0.03 * data_range);
    hipLaunchKernelGGL(ssim_labels_kernel, dim3((Wo + SSIM_T - 1) / SSIM_T, (Ho + SSIM_T - 1) / SSIM_T, N),
                       dim3(SSIM_T * SSIM_T), 0, st, (const long long*)labels_a, (const long long*)labels_b, H, W, c1, c2,
                       ssim_out);
    EESEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(ssim_finish_kernel, dim3((N + 63) / 64), dim3(64), 0, st, ssim_out, N, 1.0 / ((double)Ho * Wo));
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

// ------------------------------------------------ batched progressive early exit ----
//   Images still in flight sit in batch slots 0..n_active-1, order[slot] = their index in the caller's
// batch; both live on the device, so the host enqueues the whole network once and never waits for a gate.
__global__ __launch_bounds__(256) void argmax_exit_kernel(const float* __restrict__ lr, int ldc, int N, int C, int h, int w,
                                                          int H, int W, const int* __restrict__ flags,
                                                          const int* __restrict__ order, const int* n_active,
                                                          int64_t* pred) {
    const int lane32 = threadIdx.x & 31;
    const int half = threadIdx.x >> 5;
    const float sh = (float)h / (float)H, sw = (float)w / (float)W;
    const int n = blockIdx.y;
    if (n >= *n_active || (flags != nullptr && flags[n] == 0)) return;
    int64_t* out = pred + (size_t)order[n] * H * W;
    const int total = H * W;
    const bool active = lane32 < C;
    for (int p = blockIdx.x * 8 + half; p < total; p += gridDim.x * 8) {
        const int x = p % W, y = p / W;
        const Src sy = src_index(y, sh, h), sx = src_index(x, sw, w);
        float z = active ? interp(lr, ldc, h, w, n, sy, sx, lane32) : -INFINITY;
        int idx = lane32;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {            // first maximum wins (torch.argmax)
            const float oz = __shfl_xor(z, o);
            const int oi = __shfl_xor(idx, o);
            if (oz > z || (oz == z && oi < idx)) { z = oz; idx = oi; }
        }
        if (lane32 == 0) out[p] = idx;
    }
}

__global__ void exit_select_kernel(const int* __restrict__ flags, int N, int code, int* n_active, int* order, int* src_slot,
                                   int* exit_idx) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;   // a batch holds tens of images: one lane walks the slots in order
    int n = *n_active;
    if (n > N) n = N;
    int k = 0;
    for (int s = 0; s < n; ++s) {
        const int img = order[s];
        if (flags[s]) {
            exit_idx[img] = code;
        } else {
            order[k] = img;                            // k <= s: never overwrites a slot that is still to be read
            src_slot[k] = s;
            ++k;
        }
    }
    *n_active = k;
}

__global__ __launch_bounds__(256) void gather_images_kernel(const i32x4* __restrict__ x, i32x4* __restrict__ y,
                                                            const int* __restrict__ src_slot, const int* n_active,
                                                            long long chunks) {
    const int k = blockIdx.y;
    if (k >= *n_active) return;
    const i32x4* src = x + (size_t)src_slot[k] * chunks;
    i32x4* dst = y + (size_t)k * chunks;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < chunks; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

extern "C" int eeseg_argmax_exit(const float* logits_lr, int ldc, int N, int C, int h, int w, int H, int W,
                                 const int32_t* flags, const int32_t* order, const int32_t* n_active, int64_t* pred_out,
                                 void* stream) {
    CHECK_LR("argmax_exit");
    EESEG_CHECK(order && n_active && pred_out, EESEG_ERR_ARG, "argmax_exit: null pointer");
    long long blocks = ((long long)H * W + 8 * 16 - 1) / (8 * 16);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(argmax_exit_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, logits_lr, ldc, N, C,
                       h, w, H, W, (const int*)flags, (const int*)order, (const int*)n_active, pred_out);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_exit_select(const int32_t* flags, int N, int code, int32_t* n_active, int32_t* order, int32_t* src_slot,
                                 int32_t* exit_idx, void* stream) {
    EESEG_CHECK(flags && n_active && order && src_slot && exit_idx && N >= 1, EESEG_ERR_ARG, "exit_select: bad argument");
    hipLaunchKernelGGL(exit_select_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)flags, N, code,
                       (int*)n_active, (int*)order, (int*)src_slot, (int*)exit_idx);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_gather_images(const void* x, void* x_out, const int32_t* src_slot, const int32_t* n_active, int N,
                                   int64_t bytes_per_image, void* stream) {
    EESEG_CHECK(x && x_out && src_slot && n_active && N >= 1, EESEG_ERR_ARG, "gather_images: bad argument");
    EESEG_CHECK(bytes_per_image > 0 && bytes_per_image % 16 == 0 && (((uintptr_t)x | (uintptr_t)x_out) & 15) == 0,
                EESEG_ERR_ARG, "gather_images: images must be 16-byte multiples, 16-byte aligned");
    const long long chunks = bytes_per_image / 16;
    long long blocks = (chunks + 256 * 8 - 1) / (256 * 8);
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(gather_images_kernel, dim3((unsigned)blocks, N), dim3(256), 0, (hipStream_t)stream, (const i32x4*)x,
                       (i32x4*)x_out, (const int*)src_slot, (const int*)n_active, chunks);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int64_t eeseg_entropy_gate_workspace(int N, int H, int W) {
    return (int64_t)N * H * W * (int64_t)sizeof(float);
}

extern "C" int eeseg_entropy_gate(const float* logits_lr, int ldc, int N, int C, int h, int w, int H, int W, int pool,
                                  int pool_size, float tau, float* entropy_out, int32_t* exit_flag, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    return eeseg_entropy_gate_active(logits_lr, ldc, N, C, h, w, H, W, pool, pool_size, tau, 1, nullptr, entropy_out,
                                     exit_flag, workspace, workspace_bytes, stream);
}

extern "C" int eeseg_entropy_gate_active(const float* logits_lr, int ldc, int N, int C, int h, int w, int H, int W,
                                         int pool, int pool_size, float tau, int less_than, const int32_t* n_active,
                                         float* entropy_out, int32_t* exit_flag, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
    CHECK_LR("entropy_gate");
    EESEG_CHECK(C >= 2, EESEG_ERR_ARG, "entropy_gate: needs at least 2 classes");
    EESEG_CHECK(entropy_out && workspace && workspace_bytes >= eeseg_entropy_gate_workspace(N, H, W), EESEG_ERR_ARG,
                "entropy_gate: workspace too small");
    EESEG_CHECK(pool >= 0 && pool <= 2 && pool_size >= 1, EESEG_ERR_ARG, "entropy_gate: bad pooling mode");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(entropy_map_kernel, dim3(px_grid((long long)N * H * W)), dim3(256), 0, st, logits_lr, ldc, N, C,
                       h, w, H, W, (float*)workspace, (const int*)n_active);
    EESEG_LAUNCH_CHECK();
    hipLaunchKernelGGL(entropy_reduce_kernel, dim3(N), dim3(1024), 0, st, (const float*)workspace, H, W, pool, pool_size,
                       tau, entropy_out, (int*)exit_flag, (const int*)n_active, less_than);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}
