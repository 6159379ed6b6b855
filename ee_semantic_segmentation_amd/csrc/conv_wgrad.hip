// Convolution weight gradient for gfx950.
//
// GEMM view per filter tap: dW[cout][cin] += sum_pixels dY[pixel][cout] * X[pixel'][cin]
//   (pixel' = tap-shifted input pixel).  K = pixels is the SLOW dimension of both
//   operands in NHWC memory, so the bf16 path feeds the MFMA through the gfx950
//   transposing LDS read (ds_read_b64_tr_b16); the fp32 path (v_mfma_f32_32x32x2_f32
//   takes one k per lane) reads plain dwords.
// Tile: 128 couts x 128 cins x (64 bf16 / 32 f32) pixels per step, 4 waves of
//   64x64; split-K over pixel ranges, fp32 atomics into dW (KRSC, fp32).
#include "eeseg_common.h"

int g_last_wgrad_kernel = 0;        // eeseg_last_kernel(1) (conv_igemm.hip)
int g_last_wgrad_group = 0;         // eeseg_last_kernel(3): problems the last eeseg_conv_wgrad_group call put into ONE launch (0 = it launched them one by one)

namespace {

struct WgP {
    const void* x; const void* dy; float* dw;
    int N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, pad, dil;
    int lddy;
    int M, HWout, co_tiles, ci_tiles, splits, chunk;
    int q64, r64;     // 64 pixels = q64 output rows + r64 columns
    int qk, rk, fast_wrap;   // 128x128 kernel: the same for its K step (64 bf16 / 32 f32 pixels); qk + 1 <= Hout
    uint32_t xbytes, dybytes;
    float* slabs;     // 256x256 kernel: per-block fp32 partial tiles (register layout) instead of atomics, or NULL
    unsigned* bstate; // 256x256 kernel, in-kernel combine (round 4): barrier state; the K splits of a tile meet in the kernel and
                      // each sums its share of the tile over all splits' slabs (fixed order: reproducible, no atomics, no second launch)
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgP p) {
    constexpr int ES = (int)sizeof(T);
    constexpr int KP = 128 / ES;            // pixels per K step (64 / 32)
    constexpr int ROWW = 128 * ES;          // bytes per LDS row = 128 channels
    constexpr int CPRW = ROWW / 16;         // 16-byte chunks per row (16 / 32)
    constexpr int EPC = 16 / ES;
    constexpr int TILE = KP * ROWW;         // 16 KiB
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE];
    char* sDY = smem;                       // [2][KP][ROWW]
    char* sX = smem + 2 * TILE;             // [2][KP][ROWW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int taps = p.R * p.S;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int co_t = bid % p.co_tiles; bid /= p.co_tiles;
    const int ci_t = bid % p.ci_tiles; bid /= p.ci_tiles;
    const int tap = bid % taps;
    const int split = bid / taps;
    const int co0 = co_t * 128, ci0 = ci_t * 128;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    const int dh = tr * p.dil - p.pad, dwv = ts * p.dil - p.pad;

    const int ps = split * p.chunk;
    const int pe = min(p.M, ps + p.chunk);
    const int nk = (pe > ps) ? (pe - ps + KP - 1) / KP : 0;

    // one pixel row per thread (KP rows x TPR threads), 4 chunks interleaved over the row's threads
    // (chunk = t + TPR*i: each load/LDS-store instruction covers a contiguous run per row, conflict
    // free): a single running (n, ho, wo) and one validity test per K step.
    constexpr int TPR = CPRW / 4;            // threads per row (4 bf16 / 8 f32); 256 / TPR == KP rows
    const int lrow = tid / TPR;
    const int cb = tid % TPR;                // chunks cb + TPR*i, i = 0..3
    int pm = ps + lrow;
    int pn = pm / p.HWout;
    int prem = pm - pn * p.HWout;
    int ph = prem / p.Wout;
    int pw = prem - ph * p.Wout;
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dybytes);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.xbytes);
    // per-chunk channel validity (tails of Cout / Cin tiles)
    bool co_ok[4], ci_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        co_ok[i] = (co0 + (cb + TPR * i) * EPC) < p.Cout;
        ci_ok[i] = (ci0 + (cb + TPR * i) * EPC) < p.Cin;
    }

    i32x4 rdyv[4], rxv[4];
    auto load_tile = [&]() {
        const bool in = pm < pe;
        const uint32_t based = in ? (uint32_t)((pm * p.lddy + co0) * ES + cb * 16) : EESEG_OOB;
        const int hi = ph * p.stride + dh, wi = pw * p.stride + dwv;
        const bool ok = in && (unsigned)hi < (unsigned)p.Hin && (unsigned)wi < (unsigned)p.Win;
        const uint32_t basex = ok ? (uint32_t)((((pn * p.Hin + hi) * p.Win + wi) * p.Cin + ci0) * ES + cb * 16)
                                  : EESEG_OOB;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rdyv[i] = __builtin_amdgcn_raw_buffer_load_b128(rdy, (int)(co_ok[i] ? based + i * (TPR * 16) : EESEG_OOB), 0, 0);
            rxv[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(ci_ok[i] ? basex + i * (TPR * 16) : EESEG_OOB), 0, 0);
        }
    };
    // K-step iterator: additions only (KP pixels = qk output rows + rk columns, host-computed).  Block-uniform part:
    // does the step contain a pixel row whose tap-shifted source row is inside the image?  For the atrous convs
    // (|dh| = 12/24/36 on 65 rows) up to 55 % of the steps of the off-centre tap rows multiply only zero padding
    // and are skipped.
    int kpos = 0;                              // the K step the rows currently sit on
    int t_h, t_w;                              // image row / column of that step's first pixel (block-uniform)
    {
        const int rem0 = ps % p.HWout;
        t_h = rem0 / p.Wout;
        t_w = rem0 - t_h * p.Wout;
    }
    auto tile_valid = [&]() {
        int r = t_h, rem = t_w + KP - 1;
        bool ok = false;
        for (;;) {
            ok = ok || ((unsigned)(r * p.stride + dh) < (unsigned)p.Hin);
            if (rem < p.Wout) break;
            rem -= p.Wout;
            r = (r + 1 == p.Hout) ? 0 : r + 1;
        }
        return ok;
    };
    auto step_one = [&]() {
        ++kpos;
        t_w += p.rk; t_h += p.qk;
        if (t_w >= p.Wout) { t_w -= p.Wout; ++t_h; }
        while (t_h >= p.Hout) t_h -= p.Hout;
        pm += KP; pw += p.rk; ph += p.qk;
        if (pw >= p.Wout) { pw -= p.Wout; ++ph; }
        if (p.fast_wrap) {                     // qk + 1 <= Hout: one conditional subtraction wraps the row counter
            if (ph >= p.Hout) { ph -= p.Hout; ++pn; }
        } else {
            while (ph >= p.Hout) { ph -= p.Hout; ++pn; }
        }
    };
    auto seek_valid = [&]() {                  // stay on the current step if it is valid, else move to the next valid one
        while (kpos < nk && !tile_valid()) step_one();
        return kpos;
    };
    auto store_tile = [&](int buf) {
        const int sw = (lrow & 3) << 6;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = lrow * ROWW + (((cb + TPR * i) * 16) ^ sw);
            *reinterpret_cast<i32x4*>(sDY + buf * TILE + o) = rdyv[i];
            *reinterpret_cast<i32x4*>(sX + buf * TILE + o) = rxv[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wr = wave >> 1, wcI = wave & 1;     // cout half, cin half
    int kt = seek_valid();
    const bool any = kt < nk;
    if (any) {
        load_tile();
        store_tile(0);
    }
    __syncthreads();
    int cur = 0;
    while (kt < nk) {
        step_one();
        const int nx = seek_valid();
        const bool has_next = nx < nk;
        if (has_next) load_tile();
        const char* a = sDY + cur * TILE;
        const char* b = sX + cur * TILE;
        if constexpr (ES == 2) {
            // transposing reads: lane l of each 16-lane group addresses row q=(l>>2)&3, 4 channels at 4*(l&3)
            const int q = (lane >> 2) & 3;
            const int chl = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
            const int sw = q << 6;
#pragma unroll
            for (int ks = 0; ks < KP / 16; ++ks) {
                bf16x8 af[2], bfr[2];
                const int row0 = ks * 16 + 8 * (lane >> 5) + q;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int cb = ((wr * 64 + i * 32 + chl) * 2) ^ sw;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(a + row0 * ROWW + cb));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(a + (row0 + 4) * ROWW + cb));
                    union { s16x4 h[2]; bf16x8 v; } u;
                    u.h[0] = lo; u.h[1] = hi;
                    af[i] = u.v;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int cb = ((wcI * 64 + j * 32 + chl) * 2) ^ sw;
                    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(b + row0 * ROWW + cb));
                    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(b + (row0 + 4) * ROWW + cb));
                    union { s16x4 h[2]; bf16x8 v; } u;
                    u.h[0] = lo; u.h[1] = hi;
                    bfr[j] = u.v;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        } else {
            const int fr = lane & 31, fh = lane >> 5;
#pragma unroll 4
            for (int kk = 0; kk < KP / 2; ++kk) {
                const int row = 2 * kk + fh;
                const int sw = (row & 3) << 6;
                float av[2], bv[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    av[i] = *reinterpret_cast<const float*>(a + row * ROWW + (((wr * 64 + i * 32 + fr) * 4) ^ sw));
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    bv[j] = *reinterpret_cast<const float*>(b + row * ROWW + (((wcI * 64 + j * 32 + fr) * 4) ^ sw));
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
        if (has_next) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
        kt = nx;
    }

    if (p.bstate) {
        // ---- in-kernel combine (round 4; full 128 x 128 tiles, the whole grid resident at two blocks per CU): as epilogue C
        // of the 256-tile kernel below - slab in register layout with write-through stores, one barrier per output tile,
        // split s sums elements [s, s+1) * 4096 / splits over all splits in split order and adds them into dW.  No atomics
        // (64 KiB of them per block at ~1.3 TB/s chip-wide were 1/3-2/3 of a call at 4-8 images), reproducible.
        const int tiles = p.co_tiles * p.ci_tiles * taps;
        const int unit = (int)xcd_remap(blockIdx.x, gridDim.x);
        const int tl = unit % tiles;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.slabs, (uint32_t)((long long)tiles * p.splits * 65536ll));
        {
            const uint32_t off = (uint32_t)unit * 65536u + (uint32_t)(wave * 4096 + lane * 4) * 4u;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        union { f32x4 f; i32x4 q; } u;
                        u.f = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                        __builtin_amdgcn_raw_buffer_store_b128(u.q, rs, (int)(off + (uint32_t)(((i * 2 + j) * 4 + g) * 1024)), 0, 16);
                    }
        }
        eeseg_group_barrier(p.bstate, (unsigned)tl, (unsigned)p.splits);
        const int lo = (int)((long long)split * 4096 / p.splits), hi = (int)((long long)(split + 1) * 4096 / p.splits);
        for (int idx = lo + tid; idx < hi; idx += 256) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            const uint32_t eo = (uint32_t)tl * 65536u + (uint32_t)idx * 16u;
            const uint32_t ss = (uint32_t)tiles * 65536u;
            int s = 0;
            for (; s + 4 <= p.splits; s += 4) {
                union { f32x4 f; i32x4 q; } v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u].q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(eo + (uint32_t)(s + u) * ss), 0, 16);
#pragma unroll
                for (int u = 0; u < 4; ++u) a += v[u].f;
            }
            for (; s < p.splits; ++s) {
                union { f32x4 f; i32x4 q; } v;
                v.q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(eo + (uint32_t)s * ss), 0, 16);
                a += v.f;
            }
            const int pl = idx & 63, ijg = (idx >> 6) & 15, pw = idx >> 10;           // lane, (i, j, g), wave of the producer
            const int ai = ijg >> 3, aj = (ijg >> 2) & 1, ag = ijg & 3;
            const int ci = ci0 + (pw & 1) * 64 + aj * 32 + (pl & 31);
            const int co = co0 + (pw >> 1) * 64 + ai * 32 + 8 * ag + 4 * (pl >> 5);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float* dst = p.dw + ((size_t)(co + k) * taps + tap) * p.Cin + ci;
                *dst += a[k];
            }
        }
        return;
    }
    // ---- epilogue: fp32 atomics into dW[co][tap][ci] ---------------------------
    if (!any) return;
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = ci0 + wcI * 64 + j * 32 + fr;
            if (ci >= p.Cin) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int co = co0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                if (co < p.Cout) {
                    float* dst = p.dw + ((size_t)co * taps + tap) * p.Cin + ci;
                    __builtin_amdgcn_global_atomic_fadd_f32(
                        (__attribute__((address_space(1))) float*)dst, acc[i][j][e]);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 256 couts x 256 cins per filter tap, 8 waves (4 along cout x 2 along cin, 64 x 128 outputs per
// wave), one block per CU, bf16.  Same pipeline as conv_big_kernel (conv_igemm.hip): LDS-DMA loads
// in flight across raw barriers, two K tiles (64 pixels each) of LDS split into four 16-KiB half
// tiles (cout halves W0/W1 of dY, cin halves XA/XB of X, 128 contiguous channels each so that every DMA row is two
// whole cache lines; wave (wc, wp) owns couts i*128 + wc*32.. and cins h*128 + wp*64..; [64 pixels][128 channels], 256-byte rows,
// 64-byte XOR swizzle on the DMA source chunk), four quadrant phases per K tile, two wave groups
// half a phase apart.  Fragments come from the transposing LDS read (K = pixels is the row index).
// Split-K over pixel ranges with fp32 atomics into dW; K tiles whose tap-shifted rows are all in the
// padding are skipped (block-uniform).
constexpr int WHT = 64 * 256;              // half tile bytes
constexpr int WB_LDS = 8 * WHT;

#define WB_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define WB_BARRIER() asm volatile("s_barrier" ::: "memory")

// FAST: 64 consecutive pixels span at most two image rows (Wout > 64): the in-loop iterator is straight-line code.
// accumulators of one wave as [cout half i][cin block j][group g][4 floats] for the two MFMA shapes (conv_igemm.hip BigAcc)
template <bool M16> struct WgAcc;
template <> struct WgAcc<false> {
    f32x16 a[2][4];
    __device__ __forceinline__ float get(int i, int j, int g, int e) const { return a[i][j][4 * g + e]; }
    __device__ __forceinline__ f32x4 get4(int i, int j, int g) const {
        return f32x4{a[i][j][4 * g], a[i][j][4 * g + 1], a[i][j][4 * g + 2], a[i][j][4 * g + 3]};
    }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) a[i][j][e] = 0.f;
    }
};
template <> struct WgAcc<true> {
    f32x4 a[2][4][4];
    __device__ __forceinline__ float get(int i, int j, int g, int e) const { return a[i][j][g][e]; }
    __device__ __forceinline__ f32x4 get4(int i, int j, int g) const { return a[i][j][g]; }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) a[i][j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
};
// group g, lane -> (first of the 4 cout rows, cin column) inside a 32 x 32 sub-block
template <bool M16> __device__ __forceinline__ int wg_sub_co(int g, int lane) {
    return M16 ? (g >> 1) * 16 + (lane >> 4) * 4 : 8 * g + 4 * (lane >> 5);
}
template <bool M16> __device__ __forceinline__ int wg_sub_ci(int g, int lane) {
    return M16 ? (g & 1) * 16 + (lane & 15) : (lane & 31);
}

// M16: v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (see conv_big_kernel: same cycles per FLOP, higher clock).  A transposing
// read block is 4 pixel rows x 16 channels per 16-lane group either way; the 16x16 operand wants pixels 8g .. 8g+7 of a 32-pixel
// K step in group g = lane >> 4 (all four groups the same 16 channels), so lanes 0-31 touch rows {tq, 8 + tq} x 32 bytes: the
// source-chunk swizzle gains a 32-byte XOR on row bit 3 to keep those on different banks.
// The body serves two kernels: one weight gradient per launch (vbid / vgrid = blockIdx.x / gridDim.x), and - round 4, the
// per-GPU shards of a data-parallel run - several weight gradients side by side in ONE launch (conv_wgrad_group_kernel:
// vbid / vgrid = the block's index / block count within its own problem, group0 = first barrier group of the problem).
template <bool FAST, bool M16>
__device__ __forceinline__ void wgrad_big_body(const WgP& p, const int vbid, const int vgrid, const unsigned group0) {
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef __attribute__((address_space(3))) s16x4* lds_tr;
    __shared__ __attribute__((aligned(16))) char smem[WB_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int taps = p.R * p.S;
    int bid = xcd_remap(vbid, vgrid);
    const int unit = bid;                      // = ((split * taps + tap) * ci_tiles + ci_t) * co_tiles + co_t
    const int co_t = bid % p.co_tiles; bid /= p.co_tiles;
    const int ci_t = bid % p.ci_tiles; bid /= p.ci_tiles;
    const int tap = bid % taps;
    const int split = bid / taps;
    const int co0 = co_t * 256, ci0 = ci_t * 256;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    const int dh = tr * p.dil - p.pad, dwv = ts * p.dil - p.pad;
    const int ps = split * p.chunk;
    const int pe = min(p.M, ps + p.chunk);
    const int nk_all = (pe > ps) ? (pe - ps + 63) / 64 : 0;

    // DMA role: one wave-instruction = 4 pixel rows x 256 B; per half tile a thread fetches rows drow, drow+32
    const int drow = wave * 4 + (lane >> 4);
    const int lc = ((lane & 15) ^ ((drow & 3) << 2) ^ (M16 ? ((drow >> 3) & 1) << 1 : 0)) * 8;   // local channel of the SOURCE chunk (swizzle)
    int coW[2], ciX[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        coW[i] = (co0 + i * 128 + lc) * 2;        // a half tile = 128 CONTIGUOUS channels: whole 128-byte lines per DMA row
        ciX[i] = (ci0 + i * 128 + lc) * 2;
    }
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(p.dy, p.dybytes);
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.xbytes);

    // K-tile iterator.  It runs once per K tile inside an LDS-read slot, opposite the other wave group's 8 MFMAs
    // (256 cycles): measured with in-kernel cycle stamps, the first version (index -> offset multiplies, nested
    // wrap loops) made that slot 1280 cycles long and the whole K tile 4000 instead of ~2600.  So: every quantity
    // advances by ADDITIONS of block constants (64 pixels = q64 rows + r64 columns; the host guarantees
    // q64 + 1 <= Hout so one conditional correction wraps a counter), no multiplies, no data-dependent loops
    // except the rare skip of all-padding tiles.
    const int st = p.stride;
    const int c_dy = 64 * p.lddy * 2;                                        // dY bytes per K tile
    const int c_x = (p.q64 * st * p.Win + p.r64 * st) * p.Cin * 2;           // X bytes per K tile, no wrap
    const int c_xw = (st * p.Win - p.Wout * st) * p.Cin * 2;                 // correction when the column wraps
    const int c_xh = (p.Hin * p.Win - p.Hout * st * p.Win) * p.Cin * 2;      // correction when the row wraps (next image)
    const int tapc = (dh * p.Win + dwv) * p.Cin * 2;                         // this block's tap shift
    int pm[2], phs[2], pws[2], xoff[2], dyoff[2];   // pixel index, row*stride, column*stride, byte offsets (tap 0,0)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        pm[q] = ps + q * 32 + drow;
        const int n = pm[q] / p.HWout;
        const int rem = pm[q] - n * p.HWout;
        const int h = rem / p.Wout, w = rem - h * p.Wout;
        phs[q] = h * st; pws[q] = w * st;
        xoff[q] = (((n * p.Hin + phs[q]) * p.Win + pws[q]) * p.Cin) * 2;
        dyoff[q] = pm[q] * p.lddy * 2;
    }
    int t_hs, t_w;                             // block-uniform: row*stride / column of the first pixel of K tile kt
    {
        const int rem0 = ps % p.HWout;
        const int h0 = rem0 / p.Wout;
        t_hs = h0 * st;
        t_w = rem0 - h0 * p.Wout;
    }
    const int Hs = p.Hout * st, Ws = p.Wout * st, qs = p.q64 * st, rs = p.r64 * st;
    auto tile_valid = [&]() {                  // any tap-shifted source row of the 64 pixels inside the image?
        if (dh == 0) return true;
        int r = t_hs, rem = t_w + 63;
        bool ok = false;
        for (;;) {
            ok = ok || ((unsigned)(r + dh) < (unsigned)p.Hin);
            if (rem < p.Wout) break;
            rem -= p.Wout;
            r = (r + st == Hs) ? 0 : r + st;
        }
        return ok;
    };
    auto step_rows = [&]() {                   // everything advances by one K tile: additions and selects only
        t_w += p.r64; t_hs += qs;
        const bool tw = t_w >= p.Wout;
        t_w -= tw ? p.Wout : 0; t_hs += tw ? st : 0;
        t_hs -= (t_hs >= Hs) ? Hs : 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            pm[q] += 64; dyoff[q] += c_dy;
            pws[q] += rs; phs[q] += qs; xoff[q] += c_x;
            const bool ww = pws[q] >= Ws;
            pws[q] -= ww ? Ws : 0; phs[q] += ww ? st : 0; xoff[q] += ww ? c_xw : 0;
            const bool hw = phs[q] >= Hs;
            phs[q] -= hw ? Hs : 0; xoff[q] += hw ? c_xh : 0;
        }
    };
    int kt = -1;
    uint32_t voffDY[2] = {EESEG_OOB, EESEG_OOB}, voffX[2] = {EESEG_OOB, EESEG_OOB};
    auto set_offsets = [&](bool live) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const bool in = live && pm[q] < pe;
            voffDY[q] = in ? (uint32_t)dyoff[q] : EESEG_OOB;
            const bool ok = in && (unsigned)(phs[q] + dh) < (unsigned)p.Hin && (unsigned)(pws[q] + dwv) < (unsigned)p.Win;
            voffX[q] = ok ? (uint32_t)(xoff[q] + tapc) : EESEG_OOB;
        }
    };
    auto next_tile = [&]() -> bool {           // general form (prologue): advance to the next K tile that is not all padding
        bool live;
        for (;;) {
            if (kt >= 0) step_rows();          // (kt == -1: the rows already sit on K tile 0)
            ++kt;
            live = kt < nk_all;
            if (!live || tile_valid()) break;
        }
        set_offsets(live);
        return live;
    };
    // FAST form (Wout > 64: a K tile of 64 consecutive pixels crosses at most one row boundary).  The whole position
    // state is block-uniform (scalar registers): pixel index, image row/column, dY and X byte offsets of the tile's
    // first pixel, all advanced by additions.  A thread derives its two rows from it with a handful of selects
    // (row j = q*32 + drow sits before or after the row boundary) - no per-thread running state, no multiplies.
    int um = ps, uA = 0, uDY = 0;              // first pixel of K tile kt: index, X offset (tap 0,0), dY offset
    int jx[2], jd[2];
    if constexpr (FAST) {
        const int n0 = ps / p.HWout;
        uA = (((n0 * p.Hin + t_hs) * p.Win + t_w * st) * p.Cin) * 2;
        uDY = ps * p.lddy * 2;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            jx[q] = (q * 32 + drow) * st * p.Cin * 2;
            jd[q] = (q * 32 + drow) * p.lddy * 2;
        }
    }
    bool need_skip = false;
    // (written with selects and non-short-circuit & | only: no branch, so that the in-loop call stays in the basic
    // block of the MFMAs it is scheduled between)
    auto u_step = [&]() {                      // scalar: one K tile forward
        um += 64; uDY += c_dy; uA += 64 * st * p.Cin * 2;
        t_w += 64;
        const int cw = (t_w >= p.Wout) ? 1 : 0;                 // column wrap
        t_w -= cw ? p.Wout : 0; t_hs += cw ? st : 0; uA += cw ? c_xw : 0;
        const int ch = (t_hs >= Hs) ? 1 : 0;                    // row wrap: next image
        t_hs -= ch ? Hs : 0; uA += ch ? c_xh : 0;
    };
    auto u_valid = [&]() -> int {              // scalar: does the tile touch a source row inside the image?
        const int r2 = (t_hs + st == Hs) ? 0 : t_hs + st;
        const int v1 = ((unsigned)(t_hs + dh) < (unsigned)p.Hin) ? 1 : 0;
        const int v2 = (((p.Wout - t_w) <= 63) ? 1 : 0) & (((unsigned)(r2 + dh) < (unsigned)p.Hin) ? 1 : 0);
        return ((dh == 0) ? 1 : 0) | v1 | v2;
    };
    auto u_offsets = [&](int live) {
        const int wrap_pos = p.Wout - t_w;     // rows j >= wrap_pos belong to the next image row
        const int last_row = (t_hs + st == Hs) ? 1 : 0;
        const int r2 = last_row ? 0 : t_hs + st;
        const int wrapc = c_xw + (last_row ? c_xh : 0);
        const int ok1 = ((unsigned)(t_hs + dh) < (unsigned)p.Hin) ? 1 : 0, ok2 = ((unsigned)(r2 + dh) < (unsigned)p.Hin) ? 1 : 0;
        const int nvalid = live ? pe - um : 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = q * 32 + drow;
            const int seg = (j >= wrap_pos) ? 1 : 0;
            const int w = seg ? j - wrap_pos : t_w + j;
            const int in = (j < nvalid) ? 1 : 0;
            const int ok = in & (seg ? ok2 : ok1) & (((unsigned)(w * st + dwv) < (unsigned)p.Win) ? 1 : 0);
            voffDY[q] = in ? (uint32_t)(uDY + jd[q]) : EESEG_OOB;
            voffX[q] = ok ? (uint32_t)(uA + jx[q] + (seg ? wrapc : 0) + tapc) : EESEG_OOB;
        }
    };
    auto u_first = [&]() -> bool {             // prologue, K tile 0 (or the first one that is not all padding)
        kt = 0;
        while (kt < nk_all && !u_valid()) { u_step(); ++kt; }
        const bool live = kt < nk_all;
        u_offsets(live ? 1 : 0);
        return live;
    };
    auto advance_fast = [&]() -> bool {        // straight-line: schedulable between the MFMAs of the phase it rides in
        u_step();
        ++kt;
        const int live = (kt < nk_all) ? 1 : 0;
        need_skip = (live & (u_valid() ^ 1)) != 0;
        u_offsets(live);
        return live != 0;
    };
    auto advance_skip = [&]() -> bool {        // rare: the next K tile lies entirely in the padding
        while (kt < nk_all && !u_valid()) { u_step(); ++kt; }
        const bool live = kt < nk_all;
        u_offsets(live ? 1 : 0);
        need_skip = false;
        return live;
    };
    auto adv = [&]() -> bool {                 // prologue form of the in-loop advance (skip handled at once)
        bool live = advance_fast();
        if (need_skip) live = advance_skip();
        return live;
    };
    const int w4 = __builtin_amdgcn_readfirstlane(wave) * 4;
    auto dmaW = [&](int s, int i) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_ptr)(smem + (s * 4 + 2 + i) * WHT + (q * 32 + w4) * 256), 16,
                                                     (int)(voffDY[q] + coW[i]), 0, 0, 0);
    };
    uint32_t xbsave[2] = {EESEG_OOB, EESEG_OOB};     // X offsets of the K tile whose XB half is issued one phase later
    auto dmaX = [&](int s, int h) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + (s * 4 + h) * WHT + (q * 32 + w4) * 256), 16,
                                                     (int)(voffX[q] + ciX[h]), 0, 0, 0);
    };
    auto dmaXB_saved = [&](int s) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + (s * 4 + 1) * WHT + (q * 32 + w4) * 256), 16,
                                                     (int)(xbsave[q] + ciX[1]), 0, 0, 0);
    };

    const int wc = wave & 3, wp = wave >> 2;
    const int fr = M16 ? (lane & 15) : (lane & 31), fh = M16 ? (lane >> 4) : (lane >> 5);
    const int tq = (lane >> 2) & 3;                               // row of the 4x16 transposing-read block
    const int chl = M16 ? 4 * (lane & 3) : 16 * ((lane >> 4) & 1) + 4 * (lane & 3);   // first of my 4 channels inside a 32-wide fragment
    const int trow = (8 * fh + tq) * 256;
    // The transposing reads are inline asm: behind the builtin, hipcc (ROCm 7.2) waits vmcnt(0) before every LDS read
    // that follows an LDS-DMA issue, which would drain the load pipeline twice per K tile.  The matching
    // s_waitcnt lgkmcnt(0) (lgk_wait*) names the fragments as in/out operands so no MFMA can be scheduled above it.
    auto rd = [&](const char* base, int lch, bf16x8 (&f)[4]) {
        if constexpr (M16) {
            // f[rb * 2 + k2]: channels lch + 16rb .. +15, pixels 32k2 + 8g .. + 7 (two reads of 4 rows)
            const uint32_t a0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)(
                base + trow + ((lch * 2) ^ (tq << 6) ^ ((fh & 1) << 5))));
            const uint32_t a1 = a0 ^ 32u;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                union { i32x2 h[2]; bf16x8 v; } u;
                const uint32_t a = (ks >> 1) ? a1 : a0;
                if ((ks & 1) == 0) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(u.h[0]) : "v"(a) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(u.h[1]) : "v"(a) : "memory");
                } else {
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(u.h[0]) : "v"(a) : "memory");
                    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9216" : "=v"(u.h[1]) : "v"(a) : "memory");
                }
                f[ks] = u.v;
            }
            return;
        }
        const uint32_t a = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) const char*)(base + trow + ((lch * 2) ^ (tq << 6))));
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            union { i32x2 h[2]; bf16x8 v; } u;
            if (ks == 0) {
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(u.h[0]) : "v"(a) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(u.h[1]) : "v"(a) : "memory");
            } else if (ks == 1) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(u.h[0]) : "v"(a) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:5120" : "=v"(u.h[1]) : "v"(a) : "memory");
            } else if (ks == 2) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(u.h[0]) : "v"(a) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9216" : "=v"(u.h[1]) : "v"(a) : "memory");
            } else {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:12288" : "=v"(u.h[0]) : "v"(a) : "memory");
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:13312" : "=v"(u.h[1]) : "v"(a) : "memory");
            }
            f[ks] = u.v;
        }
    };
    auto lgk_wait4 = [&](bf16x8 (&a)[4]) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : : "memory");
    };
    auto lgk_wait8 = [&](bf16x8 (&a)[4], bf16x8 (&b)[4]) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]),
                     "+v"(b[2]), "+v"(b[3]) : : "memory");
    };
    auto lgk_wait12 = [&](bf16x8 (&a)[4], bf16x8 (&b)[4], bf16x8 (&c)[4]) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]),
                     "+v"(b[2]), "+v"(b[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]) : : "memory");
    };
    WgAcc<M16> A;
    A.zero();
    auto& acc = A.a;
// MFMA number n (0..15) of a 16x16x32 phase (conv_big_kernel's EESEG_M16)
#define EESEG_W16(W_, X0_, X1_, I_, J0_, n_) { \
        constexpr int k2_ = (n_) >> 3, rb_ = ((n_) >> 2) & 1, xs_ = ((n_) >> 1) & 1, cb_ = (n_) & 1; \
        acc[I_][(J0_) + xs_][rb_ * 2 + cb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16( \
            W_[rb_ * 2 + k2_], (xs_ ? X1_ : X0_)[cb_ * 2 + k2_], acc[I_][(J0_) + xs_][rb_ * 2 + cb_], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0); }

    // schedule and wait counts: see conv_big_kernel (phase 1: XB(t+1) | phase 3: XA(t+2), W0(t+2) | phase 4: W1(t+2))
    // The read slots hold only the transposing LDS reads and the counted wait; DMA issues are pinned between the MFMAs
    // (a half tile last read in slot P is refilled in MFMA block P+1: the lagging group has consumed its slot-P reads by
    // then), the K iterator rides in MFMA block 1.  Issue order per iteration t, all for K tile t+2 (stage s):
    //   MFMA block 2: XA, W0 | block 3: W1 | block 4: XB;  waits as in conv_big_kernel.
    bool l0 = FAST ? u_first() : next_tile();
    const bool any = l0;
    dmaX(0, 0); dmaW(0, 0); dmaW(0, 1); dmaX(0, 1);
    bool l1 = FAST ? adv() : next_tile();
    dmaX(1, 0); dmaW(1, 0); dmaW(1, 1); dmaX(1, 1);
    WB_WAIT(8);
    WB_BARRIER();
    const bool lagging = __builtin_amdgcn_readfirstlane(wave) >= 4;
    if (lagging) WB_BARRIER();
    auto dmaW1 = [&](int s, int i, int q) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_ptr)(smem + (s * 4 + 2 + i) * WHT + (q * 32 + w4) * 256), 16,
                                                 (int)(voffDY[q] + coW[i]), 0, 0, 0);
    };
    auto dmaX1 = [&](int s, int h, int q) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + (s * 4 + h) * WHT + (q * 32 + w4) * 256), 16,
                                                 (int)(voffX[q] + ciX[h]), 0, 0, 0);
    };
#define EESEG_WM(a_, b_, c_) { c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
    int s = 0;
    while (l0) {
        const char* sb = smem + s * 4 * WHT;
        bf16x8 w0[4], w1[4], xa0[4], xa1[4], xb0[4], xb1[4];
        rd(sb + 2 * WHT, wc * 32 + chl, w0); rd(sb, wp * 64 + chl, xa0); rd(sb, wp * 64 + 32 + chl, xa1);
        WB_WAIT(10);
        WB_BARRIER();
        lgk_wait12(w0, xa0, xa1);
        bool l2;
        auto mma_phase1 = [&]() {
            __builtin_amdgcn_s_setprio(1);
            if constexpr (M16) {
#pragma unroll
                for (int n = 0; n < 16; ++n) {
                    const int k2 = n >> 3, rb = (n >> 2) & 1, xs = (n >> 1) & 1, cb = n & 1;
                    acc[0][xs][rb * 2 + cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        w0[rb * 2 + k2], (xs ? xa1 : xa0)[cb * 2 + k2], acc[0][xs][rb * 2 + cb], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[ks], xa0[ks], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[ks], xa1[ks], acc[0][1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        };
        if constexpr (FAST) {
            l2 = advance_fast();               // -> K tile t+2, straight-line code between the MFMAs
            mma_phase1();
#pragma unroll
            for (int g8 = 0; g8 < (M16 ? 16 : 8); ++g8) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, M16 ? 5 : 10, 0);
                __builtin_amdgcn_sched_group_barrier(0x004, M16 ? 3 : 6, 0);
            }
            if (need_skip) l2 = advance_skip();
        } else {
            mma_phase1();
            l2 = next_tile();
        }
        WB_BARRIER();
        rd(sb + 3 * WHT, wc * 32 + chl, w1);
        WB_WAIT(8);
        WB_BARRIER();
        lgk_wait4(w1);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (M16) {
            EESEG_W16(w1, xa0, xa1, 1, 0, 0) EESEG_W16(w1, xa0, xa1, 1, 0, 1) dmaX1(s, 0, 0); __builtin_amdgcn_sched_barrier(0);
            EESEG_W16(w1, xa0, xa1, 1, 0, 2) EESEG_W16(w1, xa0, xa1, 1, 0, 3) dmaX1(s, 0, 1); __builtin_amdgcn_sched_barrier(0);
            EESEG_W16(w1, xa0, xa1, 1, 0, 4) EESEG_W16(w1, xa0, xa1, 1, 0, 5) dmaW1(s, 0, 0); __builtin_amdgcn_sched_barrier(0);
            EESEG_W16(w1, xa0, xa1, 1, 0, 6) EESEG_W16(w1, xa0, xa1, 1, 0, 7) dmaW1(s, 0, 1); __builtin_amdgcn_sched_barrier(0);
            EESEG_W16(w1, xa0, xa1, 1, 0, 8) EESEG_W16(w1, xa0, xa1, 1, 0, 9) EESEG_W16(w1, xa0, xa1, 1, 0, 10) EESEG_W16(w1, xa0, xa1, 1, 0, 11)
            EESEG_W16(w1, xa0, xa1, 1, 0, 12) EESEG_W16(w1, xa0, xa1, 1, 0, 13) EESEG_W16(w1, xa0, xa1, 1, 0, 14) EESEG_W16(w1, xa0, xa1, 1, 0, 15)
        } else {
            EESEG_WM(w1[0], xa0[0], acc[1][0]) dmaX1(s, 0, 0); __builtin_amdgcn_sched_barrier(0);
            EESEG_WM(w1[0], xa1[0], acc[1][1]) dmaX1(s, 0, 1); __builtin_amdgcn_sched_barrier(0);
            EESEG_WM(w1[1], xa0[1], acc[1][0]) dmaW1(s, 0, 0); __builtin_amdgcn_sched_barrier(0);
            EESEG_WM(w1[1], xa1[1], acc[1][1]) dmaW1(s, 0, 1); __builtin_amdgcn_sched_barrier(0);
            EESEG_WM(w1[2], xa0[2], acc[1][0]) EESEG_WM(w1[2], xa1[2], acc[1][1])
            EESEG_WM(w1[3], xa0[3], acc[1][0]) EESEG_WM(w1[3], xa1[3], acc[1][1])
        }
        __builtin_amdgcn_s_setprio(0);
        WB_BARRIER();
        rd(sb + WHT, wp * 64 + chl, xb0); rd(sb + WHT, wp * 64 + 32 + chl, xb1);
        WB_BARRIER();
        lgk_wait8(xb0, xb1);
        __builtin_amdgcn_s_setprio(1);
        if constexpr (M16) {
            EESEG_W16(w1, xb0, xb1, 1, 2, 0) EESEG_W16(w1, xb0, xb1, 1, 2, 1) dmaW1(s, 1, 0); __builtin_amdgcn_sched_barrier(0);
            EESEG_W16(w1, xb0, xb1, 1, 2, 2) EESEG_W16(w1, xb0, xb1, 1, 2, 3) dmaW1(s, 1, 1); __builtin_amdgcn_sched_barrier(0);
            EESEG_W16(w1, xb0, xb1, 1, 2, 4) EESEG_W16(w1, xb0, xb1, 1, 2, 5) EESEG_W16(w1, xb0, xb1, 1, 2, 6) EESEG_W16(w1, xb0, xb1, 1, 2, 7)
            EESEG_W16(w1, xb0, xb1, 1, 2, 8) EESEG_W16(w1, xb0, xb1, 1, 2, 9) EESEG_W16(w1, xb0, xb1, 1, 2, 10) EESEG_W16(w1, xb0, xb1, 1, 2, 11)
            EESEG_W16(w1, xb0, xb1, 1, 2, 12) EESEG_W16(w1, xb0, xb1, 1, 2, 13) EESEG_W16(w1, xb0, xb1, 1, 2, 14) EESEG_W16(w1, xb0, xb1, 1, 2, 15)
        } else {
            EESEG_WM(w1[0], xb0[0], acc[1][2]) dmaW1(s, 1, 0); __builtin_amdgcn_sched_barrier(0);
            EESEG_WM(w1[0], xb1[0], acc[1][3]) dmaW1(s, 1, 1); __builtin_amdgcn_sched_barrier(0);
            EESEG_WM(w1[1], xb0[1], acc[1][2]) EESEG_WM(w1[1], xb1[1], acc[1][3])
            EESEG_WM(w1[2], xb0[2], acc[1][2]) EESEG_WM(w1[2], xb1[2], acc[1][3])
            EESEG_WM(w1[3], xb0[3], acc[1][2]) EESEG_WM(w1[3], xb1[3], acc[1][3])
        }
        __builtin_amdgcn_s_setprio(0);
        WB_BARRIER();
        WB_WAIT(10);
        WB_BARRIER();
        __builtin_amdgcn_s_setprio(1);
        if constexpr (M16) {
            EESEG_W16(w0, xb0, xb1, 0, 2, 0) EESEG_W16(w0, xb0, xb1, 0, 2, 1) dmaX1(s, 1, 0); __builtin_amdgcn_sched_barrier(0);
            EESEG_W16(w0, xb0, xb1, 0, 2, 2) EESEG_W16(w0, xb0, xb1, 0, 2, 3) dmaX1(s, 1, 1); __builtin_amdgcn_sched_barrier(0);
            EESEG_W16(w0, xb0, xb1, 0, 2, 4) EESEG_W16(w0, xb0, xb1, 0, 2, 5) EESEG_W16(w0, xb0, xb1, 0, 2, 6) EESEG_W16(w0, xb0, xb1, 0, 2, 7)
            EESEG_W16(w0, xb0, xb1, 0, 2, 8) EESEG_W16(w0, xb0, xb1, 0, 2, 9) EESEG_W16(w0, xb0, xb1, 0, 2, 10) EESEG_W16(w0, xb0, xb1, 0, 2, 11)
            EESEG_W16(w0, xb0, xb1, 0, 2, 12) EESEG_W16(w0, xb0, xb1, 0, 2, 13) EESEG_W16(w0, xb0, xb1, 0, 2, 14) EESEG_W16(w0, xb0, xb1, 0, 2, 15)
        } else {
            EESEG_WM(w0[0], xb0[0], acc[0][2]) dmaX1(s, 1, 0); __builtin_amdgcn_sched_barrier(0);
            EESEG_WM(w0[0], xb1[0], acc[0][3]) dmaX1(s, 1, 1); __builtin_amdgcn_sched_barrier(0);
            EESEG_WM(w0[1], xb0[1], acc[0][2]) EESEG_WM(w0[1], xb1[1], acc[0][3])
            EESEG_WM(w0[2], xb0[2], acc[0][2]) EESEG_WM(w0[2], xb1[2], acc[0][3])
            EESEG_WM(w0[3], xb0[3], acc[0][2]) EESEG_WM(w0[3], xb1[3], acc[0][3])
        }
        __builtin_amdgcn_s_setprio(0);
        WB_BARRIER();
        l0 = l1; l1 = l2; s ^= 1;
    }
#undef EESEG_WM
#undef EESEG_W16
    if (!lagging) WB_BARRIER();
    WB_WAIT(0);
    if (p.bstate) {
        // ---- epilogue C (round 4, default when the whole grid is resident): in-kernel combine.  Every block publishes its
        // partial tile as a slab with write-through (sc1) 16-byte stores, the K splits of the tile meet (eeseg_group_barrier:
        // a counter per tile, no fence), then split s sums elements [s, s+1) * 16384 / splits of the tile over ALL splits'
        // slabs in split order (sc1 loads) and adds them into dW with plain read-modify-writes - it owns them.  Against the
        // atomics of epilogue B (256 blocks x 256 KiB at ~1.3 TB/s = ~49 us per call whatever the layer): the same bytes as
        // plain stores + loads, every block reducing its own share in parallel; bitwise reproducible.
        const int tiles = p.co_tiles * p.ci_tiles * taps;
        const int tl = unit % tiles;                             // = (tap * ci_tiles + ci_t) * co_tiles + co_t
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.slabs, (uint32_t)((long long)tiles * p.splits * 262144ll));
        {
            const uint32_t off = (uint32_t)unit * 262144u + (uint32_t)((wave * 32) * 256 + lane * 4) * 4u;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        union { f32x4 f; i32x4 q; } u;
                        u.f = A.get4(i, j, g);
                        __builtin_amdgcn_raw_buffer_store_b128(u.q, rs, (int)(off + (uint32_t)(((i * 4 + j) * 4 + g) * 1024)), 0, 16);   // aux 16 = sc1
                    }
        }
        eeseg_group_barrier(p.bstate, group0 + (unsigned)tl, (unsigned)p.splits);
        const int lo = (int)((long long)split * 16384 / p.splits), hi = (int)((long long)(split + 1) * 16384 / p.splits);
        const int co_t2 = tl % p.co_tiles, ci_t2 = (tl / p.co_tiles) % p.ci_tiles;      // (= co_t, ci_t of this block)
        for (int idx = lo + tid; idx < hi; idx += 512) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            const uint32_t eo = (uint32_t)tl * 262144u + (uint32_t)idx * 16u;
            const uint32_t ss = (uint32_t)tiles * 262144u;       // bytes between the same tile's slabs of consecutive splits
            int s = 0;
            for (; s + 4 <= p.splits; s += 4) {                   // four loads in flight, added in split order
                union { f32x4 f; i32x4 q; } v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u].q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(eo + (uint32_t)(s + u) * ss), 0, 16);
#pragma unroll
                for (int u = 0; u < 4; ++u) a += v[u].f;
            }
            for (; s < p.splits; ++s) {
                union { f32x4 f; i32x4 q; } v;
                v.q = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(eo + (uint32_t)s * ss), 0, 16);
                a += v.f;
            }
            // slab element idx (f32x4 units) -> producing wave, accumulator group, lane -> (4 couts, cin)
            const int pl = idx & 63, ijg = (idx >> 6) & 31, pw = idx >> 11;
            const int ai = ijg >> 4, aj = (ijg >> 2) & 3, ag = ijg & 3;
            const int ci = ci_t2 * 256 + (aj >> 1) * 128 + (pw >> 2) * 64 + (aj & 1) * 32 + wg_sub_ci<M16>(ag, pl);
            const int co = co_t2 * 256 + ai * 128 + (pw & 3) * 32 + wg_sub_co<M16>(ag, pl);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float* dst = p.dw + ((size_t)(co + k) * taps + tap) * p.Cin + ci;
                *dst += a[k];
            }
        }
        return;
    }
    if (p.slabs) {
        // ---- epilogue A (opt-in, reproducible): partial tile -> slab[unit] in register layout (1 KiB per wave store);
        // wgrad_slab_reduce_kernel sums the splits in a fixed order.  Measured 2-5 % slower than the atomics below on the
        // 3x3 layers (the atomics of one block overlap the K loops of the others; the reduce is one more launch)
        float* ws = p.slabs + (size_t)unit * 65536 + (wave * 32) * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(ws + ((i * 4 + j) * 4 + g) * 256) = A.get4(i, j, g);
        return;
    }
    if (!any) return;

    // ---- epilogue B: fp32 atomics into dW[co][tap][ci] -------------------------
    // (the K splits of a tile finish together and add in the same order; walking the accumulator groups in eight rotated orders
    // per split changed nothing, 117-531 us per call within 1 %: the adds are bound by the atomic rate, not by address conflicts)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int g = e >> 2;
                const int ci = ci0 + (j >> 1) * 128 + wp * 64 + (j & 1) * 32 + wg_sub_ci<M16>(g, lane);
                const int co = co0 + i * 128 + wc * 32 + wg_sub_co<M16>(g, lane) + (e & 3);
                float* dst = p.dw + ((size_t)co * taps + tap) * p.Cin + ci;
                __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)dst, A.get(i, j, g, e & 3));
            }
        }
    }
}

template <bool FAST, bool M16>
__global__ __launch_bounds__(512) void conv_wgrad_big_kernel(WgP p) {
    wgrad_big_body<FAST, M16>(p, (int)blockIdx.x, (int)gridDim.x, 0u);
}

// Several weight gradients in one launch (in-kernel combine only: every block resident).  At a per-GPU shard of a few images a
// single weight gradient has too little K for the chip: 4-9 output tiles x 264 K tiles split 64 ways leave ~4 K tiles per block
// and a combine of 64 slabs per tile (32 us per call whatever the block count, scripts/wgrad_group_probe.py); the three
// gradients of a bottleneck block side by side get ~15 splits each: 4x the K per block, a quarter of the combine traffic, one launch.
constexpr int WG_GROUP_MAX = 4;
struct WgGroup {
    WgP p[WG_GROUP_MAX];
    int start[WG_GROUP_MAX + 1];      // first block of problem i (start[n] = grid)
    unsigned group0[WG_GROUP_MAX];    // first barrier group (= output tile) of problem i
    int n;
};
template <bool FAST, bool M16>
__global__ __launch_bounds__(512) void conv_wgrad_group_kernel(WgGroup g) {
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.start[i + 1]) ++i;
    wgrad_big_body<FAST, M16>(g.p[i], (int)blockIdx.x - g.start[i], g.start[i + 1] - g.start[i], g.group0[i]);
}

// dW[co][tap][ci] += sum over the K splits of one 256x256 tile, slabs in the producing kernel's register layout.
// grid = (tiles x 8, G): one block per producing wave (= 32 KiB of every slab) and per group of splits, 256 threads.
// G > 1 (few output tiles, many splits: tiles x 8 blocks alone would read the slabs at a fraction of the memory rate):
// group g sums splits [g*n/G, (g+1)*n/G) into a second-level slab; a second launch with G = 1 folds those into dW.
// Fixed summation order at both levels: bitwise reproducible.
template <bool M16>
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(WgP p, const float* __restrict__ src, int nsplit,
                                                                float* dst_slabs) {
    const int taps = p.R * p.S;
    const int tiles = p.co_tiles * p.ci_tiles * taps;
    int tl = blockIdx.x >> 3;
    const int wave = blockIdx.x & 7;
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int wc = wave & 3, wp = wave >> 2, fr = lane & 31, fh = lane >> 5;
    const int G = gridDim.y, g = blockIdx.y;
    const int s_begin = (int)((long long)g * nsplit / G), s_end = (int)((long long)(g + 1) * nsplit / G);
    const size_t slice = (size_t)tl * 65536 + (wave * 32) * 256 + lane * 4;
    const float* ws = src + slice;
    const int co_t = tl % p.co_tiles; tl /= p.co_tiles;
    const int ci_t = tl % p.ci_tiles;
    const int tap = tl / p.ci_tiles;
#pragma unroll
    for (int ij = 0; ij < 2; ++ij) {
        const int i = (sub * 2 + ij) >> 2, j = (sub * 2 + ij) & 3;
        f32x4 a[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s = s_begin; s < s_end; ++s) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                a[q] += *reinterpret_cast<const f32x4*>(ws + (size_t)s * tiles * 65536 + ((i * 4 + j) * 4 + q) * 256);
        }
        if (dst_slabs != nullptr) {
            float* wd = dst_slabs + (size_t)g * tiles * 65536 + slice;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(wd + ((i * 4 + j) * 4 + q) * 256) = a[q];
            continue;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ci = ci_t * 256 + (j >> 1) * 128 + wp * 64 + (j & 1) * 32 + wg_sub_ci<M16>(q, lane);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int co = co_t * 256 + i * 128 + wc * 32 + wg_sub_co<M16>(q, lane) + k;
                float* dst = p.dw + ((size_t)co * taps + tap) * p.Cin + ci;
                *dst += a[q][k];
            }
        }
    }
}

int g_wgrad_big = 1;                // eeseg_set_wgrad_big(0|1)
int g_wgrad_big_min_ktiles = 8;     // was 20 with the atomics epilogue: with the in-kernel combine short K ranges pay no 256-KiB atomic tile any more (4 images: 21.82 / 21.78 / 21.68 / 21.72 ms per step for 20 / 12 / 8 / 5; 8 images: 33.9 / 34.1 for 12 / 8)
int g_wgrad_big_blocks = 256;       // EESEG_OPT_WGRAD_BIG_BLOCKS: concurrent blocks the 256x256 wgrad kernel sizes its K split for
int g_wgrad_big_rounds = 8;         // EESEG_OPT_WGRAD_BIG_ROUNDS: at most this many rounds of them
int g_wgrad_m16 = 0;                // eeseg_set_wgrad_big(on | 16): 256x256 kernel on v_mfma_f32_16x16x32_bf16
int g_wgrad_slabs = 0;              // eeseg_set_wgrad_big(on | 4): 4 = combine the K splits through slabs (bitwise reproducible)
int g_wgrad_coop = 1;               // eeseg_set_wgrad_big(on | 8): 8 (default) = combine the K splits INSIDE the kernel when its whole grid is resident
int g_wgrad_group_max_ktiles = 1 << 20;  // eeseg_set_wgrad_group(n): weight gradients of at most this many 64-pixel K tiles may be grouped (0 = never; default: no limit,
                                         // the cost model in eeseg_conv_wgrad_group decides)
int g_wgrad_target_blocks = 0;      // tiles * splits aimed at (eeseg_set_wgrad_target_blocks); 0 = by the cost model below

// Cost model of the 128x128-tile kernel (fitted to scripts/wgrad_sweep.py, MI355X, us): a block takes ~1.0 us per
// 64-pixel K step alone on a CU, ~1.4 us when two share it (2 resident blocks per CU = 512 slots, more blocks queue),
// and every block ends with one fp32 partial tile of float atomics that run at ~1.3 TB/s chip-wide WHATEVER the layer
// size - so the block count that is best for a long K (1024) drowns a short one in atomics (50 us per call).
struct WgPlan { long long splits; double us; };
// max_blocks > 0: the in-kernel combine (all blocks resident, at most max_blocks of them): a block then writes and reads its
// 64-KiB share with plain write-through stores / loads (~4 TB/s chip-wide) and meets its tile's other splits once (~4 us).
static WgPlan plan_wgrad128(long long M, int kp, long long tiles, double tile_bytes, long long max_splits, long long max_blocks = 0) {
    const double ksteps = (double)((M + kp - 1) / kp);
    const double b = max_blocks > 0 ? 2.0 * tile_bytes / 4.0e6 : tile_bytes / 1.3e6;      // us of combine traffic per block
    const double fixed = max_blocks > 0 ? 4.0 : 0.0;
    const long long cap = max_blocks > 0 ? max_blocks : 1536;
    WgPlan best{1, 1e30};
    for (long long sp = 1; sp <= max_splits && tiles * sp <= cap; ++sp) {
        const double blocks = (double)(tiles * sp);
        const double per_step = blocks * 0.7 / 256.0 > 1.0 ? blocks * 0.7 / 256.0 : 1.0;
        const double us = ksteps / (double)sp * per_step + b * blocks + (sp > 1 ? fixed : 0.0);
        if (us < best.us * 0.98) best = WgPlan{sp, us};
    }
    return best;
}

}  // namespace

extern "C" int eeseg_set_wgrad_big_min_ktiles(int n) {
    EESEG_CHECK(n >= 1 && n <= 4096, EESEG_ERR_ARG, "set_wgrad_big_min_ktiles: out of range");
    g_wgrad_big_min_ktiles = n;
    return EESEG_OK;
}

extern "C" int eeseg_set_wgrad_big_grid(int blocks, int rounds) {
    EESEG_CHECK(blocks >= 16 && blocks <= 1024 && rounds >= 1 && rounds <= 64, EESEG_ERR_ARG, "set_wgrad_big_grid: out of range");
    g_wgrad_big_blocks = blocks;
    g_wgrad_big_rounds = rounds;
    return EESEG_OK;
}

extern "C" int64_t eeseg_wgrad_workspace(void) {
    return 2048ll * 65536 * 4;           // tiles * splits <= 2048 slabs of 256 KiB
}

extern "C" int eeseg_set_wgrad_big(int on) {
    g_wgrad_slabs = (on & 4) ? 1 : 0;
    g_wgrad_coop = (on & 8) ? 1 : 0;
    g_wgrad_m16 = (on & 16) ? 1 : 0;
    on &= 3;
    g_wgrad_big = on > 2 ? 2 : on;
    return EESEG_OK;
}

extern "C" int eeseg_set_wgrad_target_blocks(int blocks) {
    EESEG_CHECK(blocks == 0 || (blocks >= 64 && blocks <= 65535), EESEG_ERR_ARG, "set_wgrad_target_blocks: out of range");
    g_wgrad_target_blocks = blocks;
    return EESEG_OK;
}

extern "C" int eeseg_conv_wgrad(const eeseg_wgrad_args* a, void* stream) {
    EESEG_CHECK(a && a->x && a->dy && a->dw, EESEG_ERR_ARG, "conv_wgrad: null pointer");
    const int es = eeseg_dtype_size(a->dtype);
    EESEG_CHECK(es != 0, EESEG_ERR_ARG, "conv_wgrad: bad dtype");
    const int epc = 16 / es;
    EESEG_CHECK(a->Cin % epc == 0 && a->Cout % epc == 0, EESEG_ERR_ARG,
                "conv_wgrad: Cin=%d / Cout=%d must be multiples of %d", a->Cin, a->Cout, epc);
    EESEG_CHECK(a->N > 0 && a->Hin > 0 && a->Win > 0 && a->Hout > 0 && a->Wout > 0, EESEG_ERR_ARG,
                "conv_wgrad: non-positive shape");
    EESEG_CHECK(a->R * a->S >= 1 && a->stride >= 1 && a->dil >= 1, EESEG_ERR_ARG, "conv_wgrad: bad geometry");
    const long long M = (long long)a->N * a->Hout * a->Wout;
    const long long xbytes = (long long)a->N * a->Hin * a->Win * a->Cin * es;
    const long long dybytes = M * a->Cout * es;
    EESEG_CHECK(xbytes < (1ll << 31) && dybytes < (1ll << 31), EESEG_ERR_TOO_LARGE,
                "conv_wgrad: tensor exceeds 2 GiB descriptor range");
    EESEG_CHECK(((uintptr_t)a->x & 15) == 0 && ((uintptr_t)a->dy & 15) == 0 && ((uintptr_t)a->dw & 3) == 0,
                EESEG_ERR_ARG, "conv_wgrad: misaligned pointer");
    hipStream_t st = (hipStream_t)stream;
    const int taps = a->R * a->S;
    if (!a->accumulate)
        EESEG_HIP(hipMemsetAsync(a->dw, 0, (size_t)a->Cout * taps * a->Cin * sizeof(float), st));

    WgP p;
    p.slabs = nullptr;
    p.bstate = nullptr;
    p.x = a->x; p.dy = a->dy; p.dw = a->dw;
    p.N = a->N; p.Hin = a->Hin; p.Win = a->Win; p.Cin = a->Cin;
    p.Hout = a->Hout; p.Wout = a->Wout; p.Cout = a->Cout; p.R = a->R; p.S = a->S;
    p.stride = a->stride; p.pad = a->pad; p.dil = a->dil; p.lddy = a->Cout;
    p.M = (int)M; p.HWout = a->Hout * a->Wout;
    p.co_tiles = (a->Cout + 127) / 128; p.ci_tiles = (a->Cin + 127) / 128;
    p.xbytes = (uint32_t)xbytes; p.dybytes = (uint32_t)dybytes;
    p.q64 = 64 / a->Wout; p.r64 = 64 % a->Wout;
    p.qk = (128 / es) / a->Wout; p.rk = (128 / es) % a->Wout; p.fast_wrap = (p.qk + 1 <= a->Hout) ? 1 : 0;
    if (g_wgrad_big && a->dtype == EESEG_BF16 && a->Cout % 256 == 0 && a->Cin % 256 == 0 && p.q64 + 1 <= a->Hout) {
        // 256x256 tiles, one block per CU: choose the split count that fills whole rounds of 256 blocks best
        p.co_tiles = a->Cout / 256; p.ci_tiles = a->Cin / 256;
        const long long tiles = (long long)p.co_tiles * p.ci_tiles * taps;
        const long long max_splits = (M + 8 * 64 - 1) / (8 * 64);          // at least 8 K tiles per block
        long long best = 1;
        double best_eff = 0.0;
        const long long T = g_wgrad_big_blocks;                              // blocks that run at once (256 = whole chip)
        for (long long sp = 1; sp <= max_splits && tiles * sp <= T * g_wgrad_big_rounds; ++sp) {
            const long long blocks = tiles * sp;
            const double eff = (double)blocks / (double)((blocks + T - 1) / T * T);
            if (eff > best_eff + 0.02) { best_eff = eff; best = sp; }
        }
        long long chunk = (M + best - 1) / best;
        chunk = (chunk + 63) / 64 * 64;
        const long long splits = (M + chunk - 1) / chunk;
        p.splits = (int)splits; p.chunk = (int)chunk;
        // short K ranges (few output tiles -> many splits) leave one block per CU mostly filling and draining its
        // pipeline: measured break-even against the 128x128 kernel (2-3 blocks per CU) at ~20 K tiles per block
        // ... and against the cost model of the 128x128 kernel: few output tiles mean many K splits, i.e. many 256-KiB
        // partial tiles (0.2 us of atomics each) for little MFMA work per block
        bool take_big = chunk / 64 >= g_wgrad_big_min_ktiles;
        if (take_big && g_wgrad_target_blocks == 0) {
            const long long t128 = (long long)((a->Cout + 127) / 128) * ((a->Cin + 127) / 128) * taps;
            const WgPlan alt = plan_wgrad128(M, 64, t128, 65536.0, (M + 4 * 64 - 1) / (4 * 64));
            const double blocks = (double)(tiles * splits);
            const double rounds = (double)((tiles * splits + T - 1) / T);
            // combine of the K splits: 256 KiB of atomics per block (~0.2 us each at the chip-wide atomic rate), or - one
            // resident round, in-kernel combine (epilogue C) - a write-through slab + a share of the reads (~0.07 us) and one barrier
            extern int eeseg_get_option(int);
            const bool coop_ok = g_wgrad_coop && splits >= 2 && tiles * splits <= eeseg_get_option(EESEG_OPT_CONV_CUS) &&
                                 tiles <= EESEG_BARRIER_GROUPS && a->workspace && a->barrier_state;
            const double big_us = rounds * (double)(chunk / 64) * 1.6 + 7.5 + (coop_ok ? 0.07 * blocks + 4.0 : 0.2 * blocks);
            // the 128-tile estimate is optimistic for few-tile layers (measured at 32 x 65 x 65: 1x1 1024->256 model 111 us, kernel
            // 147 us, against 114 us for the 256-tile kernel; 3x3 256->256 model 230, kernel 257): it has to win clearly
            take_big = big_us <= alt.us * 1.3;
        }
        if (g_wgrad_big == 2 || take_big) {
            // second reduce level when tiles x 8 blocks alone could not pull the slabs at the memory rate
            long long G = 512 / (tiles * 8);
            if (G > splits / 2) G = splits / 2;
            if (G < 2) G = 1;
            const long long need = tiles * (splits + (G > 1 ? G : 0)) * 65536ll * 4;
            p.slabs = (splits >= 2 && g_wgrad_slabs && a->workspace && a->workspace_bytes >= need)
                          ? reinterpret_cast<float*>(a->workspace) : nullptr;
            // in-kernel combine: every block of the launch resident (one per CU the plans may count on), one barrier group per tile
            extern int eeseg_get_option(int);
            if (g_wgrad_coop && splits >= 2 && tiles * splits <= eeseg_get_option(EESEG_OPT_CONV_CUS) && tiles <= EESEG_BARRIER_GROUPS &&
                a->workspace && a->workspace_bytes >= tiles * splits * 262144ll && a->barrier_state &&
                ((uintptr_t)a->barrier_state & 127) == 0) {
                p.slabs = reinterpret_cast<float*>(a->workspace);
                p.bstate = reinterpret_cast<unsigned*>(a->barrier_state);
            }
            g_last_wgrad_kernel = EESEG_KERNEL_WGRAD_BIG;
            const bool m16 = g_wgrad_m16 != 0;
            const dim3 grid((unsigned)(tiles * splits));
            if (p.q64 == 0) {
                if (m16) hipLaunchKernelGGL((conv_wgrad_big_kernel<true, true>), grid, dim3(512), 0, st, p);
                else hipLaunchKernelGGL((conv_wgrad_big_kernel<true, false>), grid, dim3(512), 0, st, p);
            } else {
                if (m16) hipLaunchKernelGGL((conv_wgrad_big_kernel<false, true>), grid, dim3(512), 0, st, p);
                else hipLaunchKernelGGL((conv_wgrad_big_kernel<false, false>), grid, dim3(512), 0, st, p);
            }
            auto reduce = [&](dim3 g, const float* src, int n, float* dst) {
                if (m16) hipLaunchKernelGGL(wgrad_slab_reduce_kernel<true>, g, dim3(256), 0, st, p, src, n, dst);
                else hipLaunchKernelGGL(wgrad_slab_reduce_kernel<false>, g, dim3(256), 0, st, p, src, n, dst);
            };
            if (p.bstate) {
                // combined in the kernel: nothing to launch
            } else if (p.slabs && G > 1) {
                float* lvl2 = p.slabs + (size_t)tiles * splits * 65536;
                reduce(dim3((unsigned)(tiles * 8), (unsigned)G), (const float*)p.slabs, (int)splits, lvl2);
                reduce(dim3((unsigned)(tiles * 8), 1), (const float*)lvl2, (int)G, (float*)nullptr);
            } else if (p.slabs) {
                reduce(dim3((unsigned)(tiles * 8), 1), (const float*)p.slabs, (int)splits, (float*)nullptr);
            }
            EESEG_LAUNCH_CHECK();
            return EESEG_OK;
        }
        p.co_tiles = (a->Cout + 127) / 128; p.ci_tiles = (a->Cin + 127) / 128;
    }
    const int kp = 128 / es;
    const long long tiles = (long long)p.co_tiles * p.ci_tiles * taps;
    const long long max_splits = (M + 4 * kp - 1) / (4 * kp); // at least 4 K steps per block
    const double tile_bytes = 4.0 * (a->Cout < 128 ? a->Cout : 128) * (a->Cin < 128 ? a->Cin : 128);
    long long splits;
    bool coop128 = false;
    if (g_wgrad_target_blocks > 0)
        splits = (g_wgrad_target_blocks + tiles - 1) / tiles;
    else if (a->dtype == EESEG_BF16 && tile_bytes >= 32768.0) {
        const WgPlan at = plan_wgrad128(M, kp, tiles, tile_bytes, max_splits);
        splits = at.splits;
        // in-kernel combine instead of atomics: full tiles, the whole grid resident at two blocks per CU, one barrier group per tile
        extern int eeseg_get_option(int);
        const long long cap = 2ll * eeseg_get_option(EESEG_OPT_CONV_CUS);
        if (g_wgrad_coop && a->Cout % 128 == 0 && a->Cin % 128 == 0 && tiles <= EESEG_BARRIER_GROUPS && tiles <= cap && a->workspace &&
            a->barrier_state && ((uintptr_t)a->barrier_state & 127) == 0) {
            const WgPlan cp = plan_wgrad128(M, kp, tiles, tile_bytes, max_splits, cap);
            // (taken whenever it applies: by the model the two forms are within a few percent of each other at every layer shape of
            // the workload, and this one gives bitwise reproducible gradients)
            if (cp.splits >= 2 && a->workspace_bytes >= tiles * cp.splits * 65536ll) {
                splits = cp.splits;
                coop128 = true;
            }
        }
    }
    else
        splits = (1024 + tiles - 1) / tiles;                  // narrow (64x64) tiles and fp32: cheap atomics, many blocks
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long long chunk = (M + splits - 1) / splits;
    chunk = (chunk + kp - 1) / kp * kp;
    splits = (M + chunk - 1) / chunk;
    p.splits = (int)splits; p.chunk = (int)chunk;
    const long long grid = tiles * splits;
    EESEG_CHECK(grid < (1ll << 31), EESEG_ERR_TOO_LARGE, "conv_wgrad: grid too large");
    if (coop128 && splits >= 2) {
        p.slabs = reinterpret_cast<float*>(a->workspace);
        p.bstate = reinterpret_cast<unsigned*>(a->barrier_state);
    }
    g_last_wgrad_kernel = EESEG_KERNEL_WGRAD_128;
    if (a->dtype == EESEG_BF16)
        hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<float>), dim3((unsigned)grid), dim3(256), 0, st, p);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

extern "C" int eeseg_set_wgrad_group(int max_ktiles) {
    EESEG_CHECK(max_ktiles >= 0 && max_ktiles <= (1 << 20), EESEG_ERR_ARG, "set_wgrad_group: out of range");
    g_wgrad_group_max_ktiles = max_ktiles;
    return EESEG_OK;
}

extern "C" int eeseg_conv_wgrad_group(const eeseg_wgrad_args* a, int n, void* stream) {
    EESEG_CHECK(a && n >= 1 && n <= WG_GROUP_MAX, EESEG_ERR_ARG, "conv_wgrad_group: 1 .. %d problems", WG_GROUP_MAX);
    g_last_wgrad_group = 0;
    extern int eeseg_get_option(int);
    const long long T = eeseg_get_option(EESEG_OPT_CONV_CUS);
    // one launch only if EVERY problem is one the 256x256 kernel combines in-kernel, all of them small (few K tiles) and of one kernel
    // instantiation; anything else: the calls one by one, each with its own plan
    bool ok = n >= 2 && g_wgrad_big && g_wgrad_coop && g_wgrad_group_max_ktiles > 0 && g_wgrad_target_blocks == 0;
    long long tiles[WG_GROUP_MAX], kt[WG_GROUP_MAX], sum_tiles = 0;
    double work = 0.0;
    for (int i = 0; ok && i < n; ++i) {
        const eeseg_wgrad_args& q = a[i];
        ok = q.x && q.dy && q.dw && q.dtype == EESEG_BF16 && q.Cout % 256 == 0 && q.Cin % 256 == 0 && q.N > 0 && q.Hout > 0 && q.Wout > 0 &&
             q.R * q.S >= 1 && q.stride >= 1 && q.dil >= 1 && 64 / q.Wout + 1 <= q.Hout && (64 / q.Wout == 0) == (64 / a[0].Wout == 0) &&
             q.workspace == a[0].workspace && q.workspace_bytes == a[0].workspace_bytes && q.barrier_state == a[0].barrier_state &&
             q.workspace && q.barrier_state && ((uintptr_t)q.barrier_state & 127) == 0 &&
             ((uintptr_t)q.x & 15) == 0 && ((uintptr_t)q.dy & 15) == 0 && ((uintptr_t)q.dw & 3) == 0;
        if (!ok) break;
        const long long M = (long long)q.N * q.Hout * q.Wout;
        ok = (long long)q.N * q.Hin * q.Win * q.Cin * 2 < (1ll << 31) && M * q.Cout * 2 < (1ll << 31);
        tiles[i] = (long long)(q.Cout / 256) * (q.Cin / 256) * q.R * q.S;
        kt[i] = (M + 63) / 64;
        ok = ok && kt[i] <= g_wgrad_group_max_ktiles;
        sum_tiles += tiles[i];
        work += (double)tiles[i] * (double)kt[i];
    }
    ok = ok && sum_tiles <= T && sum_tiles <= EESEG_BARRIER_GROUPS;
    WgGroup g{};
    long long blocks = 0, slab_tiles = 0;
    long long sp_of[WG_GROUP_MAX];
    if (ok) {
        // splits: one each, then one more for the problem with the most K tiles per block while its tiles still fit on the chip and a block
        // keeps at least 4 K tiles (all blocks run at once: the launch lasts as long as its longest K range)
        long long used = sum_tiles;
        for (int i = 0; i < n; ++i) sp_of[i] = 1;
        for (;;) {
            int w = -1;
            long long worst = 0;
            for (int i = 0; i < n; ++i) {
                const long long per = (kt[i] + sp_of[i] - 1) / sp_of[i];
                if (per > worst) { worst = per; w = i; }
            }
            if (w < 0 || used + tiles[w] > T || sp_of[w] >= (kt[w] + 3) / 4) break;
            ++sp_of[w];
            used += tiles[w];
        }
        // does one launch pay?  ~1.3 us per K tile of the longest range + one launch's fixed cost, against the single calls each on the
        // whole chip (their K split: one round of blocks, at least 8 K tiles each) with a launch's fixed cost apiece.  Small shards: always
        // (32 us per call whatever its size, scripts/wgrad_group_probe.py); the 16 + 36 + 16 tiles of a layer-4 block at 32 images: no -
        // three splits each leave a fifth of the chip idle where the single calls fill it
        double grouped = 0.0, one_by_one = 0.0;
        for (int i = 0; i < n; ++i) {
            const double per = (double)((kt[i] + sp_of[i] - 1) / sp_of[i]);
            if (per > grouped) grouped = per;
            long long s1 = T / tiles[i];
            if (s1 > kt[i] / 8) s1 = kt[i] / 8;
            if (s1 < 1) s1 = 1;
            one_by_one += 1.3 * (double)((kt[i] + s1 - 1) / s1) + 25.0;
        }
        grouped = 1.3 * grouped + 30.0;
        ok = grouped < one_by_one;
    }
    if (ok) {
        g.n = n;
        for (int i = 0; i < n; ++i) {
            const eeseg_wgrad_args& q = a[i];
            const long long M = (long long)q.N * q.Hout * q.Wout;
            long long sp = sp_of[i];
            long long chunk = (M + sp - 1) / sp;
            chunk = (chunk + 63) / 64 * 64;
            sp = (M + chunk - 1) / chunk;
            WgP& p = g.p[i];
            p.x = q.x; p.dy = q.dy; p.dw = q.dw;
            p.N = q.N; p.Hin = q.Hin; p.Win = q.Win; p.Cin = q.Cin;
            p.Hout = q.Hout; p.Wout = q.Wout; p.Cout = q.Cout; p.R = q.R; p.S = q.S;
            p.stride = q.stride; p.pad = q.pad; p.dil = q.dil; p.lddy = q.Cout;
            p.M = (int)M; p.HWout = q.Hout * q.Wout;
            p.co_tiles = q.Cout / 256; p.ci_tiles = q.Cin / 256;
            p.xbytes = (uint32_t)((long long)q.N * q.Hin * q.Win * q.Cin * 2); p.dybytes = (uint32_t)(M * q.Cout * 2);
            p.q64 = 64 / q.Wout; p.r64 = 64 % q.Wout;
            p.qk = 64 / q.Wout; p.rk = 64 % q.Wout; p.fast_wrap = (p.qk + 1 <= q.Hout) ? 1 : 0;
            p.splits = (int)sp; p.chunk = (int)chunk;
            p.slabs = reinterpret_cast<float*>(q.workspace) + (size_t)slab_tiles * 65536;
            p.bstate = reinterpret_cast<unsigned*>(q.barrier_state);
            g.start[i] = (int)blocks;
            g.group0[i] = (unsigned)(i == 0 ? 0 : g.group0[i - 1] + (unsigned)tiles[i - 1]);
            blocks += tiles[i] * sp;
            slab_tiles += tiles[i] * sp;
        }
        for (int i = n; i <= WG_GROUP_MAX; ++i) g.start[i] = (int)blocks;
        ok = blocks <= T && slab_tiles * 262144ll <= a[0].workspace_bytes;
    }
    if (!ok) {
        for (int i = 0; i < n; ++i) {
            const int rc = eeseg_conv_wgrad(&a[i], stream);
            if (rc != EESEG_OK) return rc;
        }
        g_last_wgrad_group = 0;
        return EESEG_OK;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n; ++i)
        if (!a[i].accumulate)
            EESEG_HIP(hipMemsetAsync(a[i].dw, 0, (size_t)a[i].Cout * a[i].R * a[i].S * a[i].Cin * sizeof(float), st));
    const bool m16 = g_wgrad_m16 != 0;
    const dim3 grid((unsigned)blocks);
    if (g.p[0].q64 == 0) {
        if (m16) hipLaunchKernelGGL((conv_wgrad_group_kernel<true, true>), grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL((conv_wgrad_group_kernel<true, false>), grid, dim3(512), 0, st, g);
    } else {
        if (m16) hipLaunchKernelGGL((conv_wgrad_group_kernel<false, true>), grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL((conv_wgrad_group_kernel<false, false>), grid, dim3(512), 0, st, g);
    }
    g_last_wgrad_kernel = EESEG_KERNEL_WGRAD_BIG;
    g_last_wgrad_group = n;
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}
