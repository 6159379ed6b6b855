// Implicit-GEMM convolution for gfx950 (forward and data-gradient).  Two kernels:
//   conv_big_kernel  (further down): 256 px x 256 cout tile, 8 waves, one block per CU, LDS-DMA loads in flight across
//                    raw barriers - bf16 layers with Cout % 256 == 0 and a stride-1 gather (the default for them);
//   conv_igemm_kernel (below): 128 px x 128/64 cout tile, 4 waves, 2-3 blocks per CU - everything else (fp32 parity
//                    mode, narrow layers, strided data-gradient).
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[pixel][k], k = (tap, cin).
//   A operand = weight tile (rows = cout), B operand = gathered pixel tile, so the
//   accumulator holds 4 consecutive couts per lane/register group -> the epilogue
//   writes whole NHWC rows.
// Tile: 128 pixels x BN couts x 128 bytes of K (64 bf16 / 32 f32 channels of one
//   tap) per step; 256 threads = 4 waves; each wave owns 32x32 MFMA tiles
//   (v_mfma_f32_32x32x16_bf16 or the exact-fp32 v_mfma_f32_32x32x2_f32).
// Staging: global -> registers (buffer_load, out-of-image taps read as zero via
//   an out-of-range offset) -> XOR-swizzled LDS, double buffered, one barrier per
//   K step; taps that no pixel of the tile can see (dilation 12/24/36 on 65x65
//   maps) are skipped entirely.
// Epilogue: acc*scale+shift -> LDS (row major) -> per-channel partial sums for
//   train-mode BatchNorm + coalesced 16-byte row stores (+residual, ReLU).
#include "eeseg_common.h"

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits vmcnt(0): behind a tile's output stores (or
// with the next tile's LDS-DMA in flight) that is a wait for HBM, not for the workgroup.
#define WS_LDS_BARRIER() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); asm volatile("s_barrier" ::: "memory"); }

namespace {

struct ConvP {
    const void* x; const void* w; void* y;
    const float* scale; const float* shift; const void* residual; float* stats;
    int N, Hin, Win, Cin, Hout, Wout, Cout, R, S;
    int smul, off_h, off_w, tstep_h, tstep_w, sdiv;
    int ldy, ldres, relu;
    int M, HWout, n_tiles, m_tiles;
    uint32_t xbytes, wbytes;
    int vec_ok;   // 1: 16-byte row stores allowed (alignment / Cout multiple)
    int tap_inner;   // LINEAR kernels: K order (0 = taps outer / channels inner, 1 = taps inner)
    int tile_begin, ksplit;   // 256x256 kernel: first tile of this launch (MODE 0/1) or of the split tail (MODE 2), K ranges per tile
    int n_split_blocks;       // MODE 2: leading blocks that work on K ranges of the tail tiles
    float* slabs;             // 256x256 kernel: fp32 partial tiles [tile][range][256*256]
    int pointwise;            // 256x256 kernel: 1x1, stride 1, no padding (source pixel = output pixel)
    const int* n_active;      // device count of the leading images that are computed at all (NULL = all)
    const unsigned char* resmask;   // optional 1-bit mask of the residual (one byte per 16-byte chunk of a row, bn_apply_relu_mask)
    int ldmask;                     // bytes per mask row
    // generalised taps (256-tile kernel only): the K loop walks n_gtaps (shift, source) pairs instead of an R x S grid -
    // tap t reads source pixel (h + gdh[t], w + gdw[t]) of the [N,Hin,Win,Cin] tensor that starts goff[t] bytes behind x.
    // One launch then sums several convolutions of different dilation into one output (the ASPP data-gradient).
    int n_gtaps;
    short gdh[32], gdw[32];
    int goff[32];
    // 256-tile kernel, tile order: tiles below cg_limit are walked in groups of 32 = cg cout tiles x (32 / cg) pixel tiles
    // (0 = cout tiles fastest over ALL n_tiles): the 32 CUs of an XCD then share cg weight tiles instead of n_tiles
    int cg, cg_limit;
};

// tile index -> (cout tile, pixel tile) of the 256-tile kernel and its fix-up
__device__ __forceinline__ void big_tile_coords(const ConvP& p, int tile, int& nt, int& mt) {
    if (tile < p.cg_limit) {
        const int blk = tile >> 5, in = tile & 31;
        const int ngrp = p.n_tiles / p.cg;
        const int pg = blk / ngrp, cgi = blk - pg * ngrp;
        const int c = in % p.cg;
        nt = cgi * p.cg + c;
        mt = pg * (32 / p.cg) + in / p.cg;
    } else {
        nt = tile % p.n_tiles;
        mt = tile / p.n_tiles;
    }
}

// residual chunk `q` (8 bf16) AND-ed with mask byte `mb`: element e survives iff bit e is set
__device__ __forceinline__ i32x4 mask_chunk_bf16(i32x4 q, unsigned mb) {
    i32x4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned lo = (mb >> (2 * k)) & 1u, hi = (mb >> (2 * k + 1)) & 1u;
        r[k] = (int)((unsigned)q[k] & ((lo ? 0x0000ffffu : 0u) | (hi ? 0xffff0000u : 0u)));
    }
    return r;
}

// progressive inference: a block whose first output pixel belongs to an image >= *n_active has nothing to do
#define EESEG_ACTIVE_EXIT(first_px) \
    if (p.n_active != nullptr && (long long)(first_px) >= (long long)(*p.n_active) * p.HWout) return

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 Frag;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    typedef f32x4 Frag;
    static __device__ __forceinline__ void run(const Frag& a, const Frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], c, 0, 0, 0);
    }
};

__device__ __forceinline__ bool map_coord(int base, int t, int tstep, int sdiv, int lim, int& out) {
    int v = base + t * tstep;
    if (sdiv > 1) {
        if (v < 0) return false;
        const int q = v / sdiv;
        if (q * sdiv != v) return false;
        v = q;
    }
    out = v;
    return (unsigned)v < (unsigned)lim;
}

constexpr int BM = 128;     // pixels per tile
constexpr int ROWB = 128;   // bytes of K per LDS row

// LINEAR (sdiv == 1): K order = channel chunk OUTER, filter tap INNER, so at any moment every
// resident block streams the same 128-byte channel slice of the activation for all taps (L2-hot:
// ~0.5 MB per image) instead of re-streaming the whole tensor once per tap; gather offsets are
// base[row] + delta[tap] with a per-row valid-tap bit mask.  !LINEAR (strided data-gradient, where
// the source coordinate is not linear in the tap): tap outer, offsets recomputed per tap.
template <typename T, int BN, int PIPE, bool LINEAR>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvP p) {
    constexpr int EPK = ROWB / (int)sizeof(T);        // K elements per step
    constexpr int WAVES_C = BN / 64;                  // waves along cout
    constexpr int WAVES_P = 4 / WAVES_C;              // waves along pixels
    constexpr int TI = 2;                             // 32-wide cout tiles per wave
    constexpr int TJ = BM / (32 * WAVES_P);           // 32-wide pixel tiles per wave
    constexpr int WCH = BN / 32;                      // weight chunks per thread
    constexpr int EPC = 16 / (int)sizeof(T);          // elements per 16-byte chunk
    constexpr int CPR = BN / EPC;                     // chunks per output row
    constexpr int SROW = BN * (int)sizeof(T) + 16;    // staged output row bytes
    constexpr int MAIN_BYTES = 2 * (BM + BN) * ROWB;
    constexpr int EPI_BYTES = BM * SROW + 4 * BN * 2 * 4;
    constexpr int LDS_BYTES = (MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES) + 16;
    typedef typename Mma<T>::Frag Frag;

    __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
    char* sX = smem;                         // [2][BM][ROWB]
    char* sW = smem + 2 * BM * ROWB;         // [2][BN][ROWB]
    unsigned* sMask = reinterpret_cast<unsigned*>(smem + LDS_BYTES - 16);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % p.n_tiles, mt = bid / p.n_tiles;
    const int m0 = mt * BM, n0 = nt * BN;
    EESEG_ACTIVE_EXIT(m0);
    const int taps = p.R * p.S;

    // ---- per-thread gather rows (fixed for the whole K loop) -----------------
    const int lrow = tid >> 3, lchunk = tid & 7;
    // PIPE == 0: tiles are staged by LDS-DMA (buffer_load ... lds): the LDS image is lane-linear, so the XOR
    // swizzle moves to the SOURCE chunk each lane fetches (rows lrow+32j share (row>>1)&7)
    constexpr bool DMA = (PIPE == 0);
    const int gchunk = DMA ? (lchunk ^ ((lrow >> 1) & 7)) : lchunk;
    int hb[4], wb[4], nb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = m0 + lrow + 32 * j;
        if (m < p.M) {
            const int n = m / p.HWout;
            const int rem = m - n * p.HWout;
            const int ho = rem / p.Wout;
            const int wo = rem - ho * p.Wout;
            hb[j] = ho * p.smul + p.off_h;
            wb[j] = wo * p.smul + p.off_w;
            nb[j] = n * p.Hin * p.Win;
        } else {
            hb[j] = -(1 << 28); wb[j] = -(1 << 28); nb[j] = 0;
        }
    }
    // ---- which taps can any pixel of this tile see? ---------------------------
    if (tid == 0) *sMask = 0u;
    __syncthreads();
    unsigned vmask[4] = {0u, 0u, 0u, 0u};        // per gathered row: taps that land inside the image
    {
        unsigned mine = 0u;
        for (int t = 0; t < taps; ++t) {
            const int r = t / p.S, s = t - r * p.S;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int hi, wi;
                if (map_coord(hb[j], r, p.tstep_h, p.sdiv, p.Hin, hi) && map_coord(wb[j], s, p.tstep_w, p.sdiv, p.Win, wi))
                    vmask[j] |= 1u << t;
            }
        }
        mine = vmask[0] | vmask[1] | vmask[2] | vmask[3];
        // OR-reduce inside the wave, then one LDS atomic per wave
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine |= (unsigned)__shfl_xor((int)mine, o);
        if (lane == 0 && mine) atomicOr(sMask, mine);
    }
    __syncthreads();
    const unsigned tapmask = *sMask;
    const int kc_steps = p.Cin / EPK;
    const int nk = __popc(tapmask) * kc_steps;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.xbytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.wbytes);

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    i32x4 ra0[4], rw0[WCH], ra1[4], rw1[WCH];     // two staging register sets (tiles k+1 and k+2)
    // Gather offsets are recomputed only when the filter tap changes; the channel step inside a
    // tap goes through the buffer instruction's scalar offset (no per-K-step VALU address math).
    // An out-of-image row keeps voffset = EESEG_OOB: voffset + soffset stays out of range -> zeros.
    uint32_t voffA[4], voffW[WCH];
    int baseA[4];                 // LINEAR: byte offset of the tap-(0,0) source pixel (may be negative)
    unsigned rest = tapmask;
    int ci = (LINEAR && p.tap_inner) ? 0 : kc_steps - 1;   // tap-outer orders: first next_tile() advances the tap
    int cur_tap = 0, soffA = 0, soffW = 0;
    bool live = nk > 0;

    auto set_tap = [&](int tap) {            // !LINEAR only
        const int r = tap / p.S, s = tap - r * p.S;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int hi = 0, wi = 0;
            const bool ok = map_coord(hb[j], r, p.tstep_h, p.sdiv, p.Hin, hi) &&
                            map_coord(wb[j], s, p.tstep_w, p.sdiv, p.Win, wi);
            voffA[j] = ok ? (uint32_t)(((nb[j] + hi * p.Win + wi) * p.Cin) * (int)sizeof(T) + gchunk * 16) : EESEG_OOB;
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            const int co = n0 + lrow + 32 * j;
            voffW[j] = (co < p.Cout) ? (uint32_t)(((co * taps + tap) * p.Cin) * (int)sizeof(T) + gchunk * 16) : EESEG_OOB;
        }
    };
    if constexpr (LINEAR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) baseA[j] = ((nb[j] + hb[j] * p.Win + wb[j]) * p.Cin) * (int)sizeof(T) + gchunk * 16;
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            const int co = n0 + lrow + 32 * j;
            voffW[j] = (co < p.Cout) ? (uint32_t)((co * taps * p.Cin) * (int)sizeof(T) + gchunk * 16) : EESEG_OOB;
        }
    }
    auto set_tap_linear = [&](int tap) {     // LINEAR: offsets of one tap = base[row] + delta[tap], masked
        const int r = tap / p.S, s = tap - r * p.S;
        const int dtap = ((r * p.tstep_h) * p.Win + s * p.tstep_w) * p.Cin * (int)sizeof(T);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            voffA[j] = (live && ((vmask[j] >> tap) & 1u)) ? (uint32_t)(baseA[j] + dtap) : EESEG_OOB;
    };
    auto next_tile = [&]() {   // block-uniform iterator over the K tiles
        if constexpr (LINEAR) {
            if (p.tap_inner) {                    // channel chunk outer, taps inner
                if (!rest) { rest = tapmask; ++ci; }
                live = live && ci < kc_steps;
                cur_tap = __ffs(rest) - 1;
                rest &= rest - 1;
                set_tap_linear(cur_tap);
            } else if (++ci >= kc_steps) {        // taps outer, channel chunks inner (default)
                ci = 0;
                live = live && rest != 0u;
                cur_tap = rest ? __ffs(rest) - 1 : 0;
                rest &= rest - 1;
                set_tap_linear(cur_tap);
            }
            soffA = ci * ROWB;
            soffW = live ? (cur_tap * p.Cin) * (int)sizeof(T) + ci * ROWB : 0x7FFFFF00;
        } else {
            if (++ci >= kc_steps) {
                ci = 0;
                if (rest) {
                    const int tap = __ffs(rest) - 1;
                    rest &= rest - 1;
                    set_tap(tap);
                } else {       // past the last tile: every load becomes an out-of-range (zero, no traffic) load
#pragma unroll
                    for (int j = 0; j < 4; ++j) voffA[j] = EESEG_OOB;
#pragma unroll
                    for (int j = 0; j < WCH; ++j) voffW[j] = EESEG_OOB;
                }
            }
            soffA = soffW = ci * ROWB;
        }
    };
    auto load_tile = [&](i32x4 (&ra)[4], i32x4 (&rwv)[WCH]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ra[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)voffA[j], soffA, 0);
#pragma unroll
        for (int j = 0; j < WCH; ++j) rwv[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, (int)voffW[j], soffW, 0);
    };
    auto store_tile = [&](int buf, const i32x4 (&ra)[4], const i32x4 (&rwv)[WCH]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = lrow + 32 * j;
            *reinterpret_cast<i32x4*>(sX + buf * BM * ROWB + row * ROWB + ((lchunk ^ ((row >> 1) & 7)) << 4)) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j) {
            const int row = lrow + 32 * j;
            *reinterpret_cast<i32x4*>(sW + buf * BN * ROWB + row * ROWB + ((lchunk ^ ((row >> 1) & 7)) << 4)) = rwv[j];
        }
    };

    auto dma_tile = [&](int buf) {    // one 1-KiB wave-instruction = 8 rows x 128 B, LDS dest = uniform base + lane*16
        typedef __attribute__((address_space(3))) void* lds_ptr;
        const int w8 = __builtin_amdgcn_readfirstlane(wave) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(sX + buf * BM * ROWB + (32 * j + w8) * ROWB), 16,
                                                     (int)voffA[j], soffA, 0, 0);
#pragma unroll
        for (int j = 0; j < WCH; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(sW + buf * BN * ROWB + (32 * j + w8) * ROWB), 16,
                                                     (int)voffW[j], soffW, 0, 0);
    };

    const int wc = wave % WAVES_C, wp = wave / WAVES_C;
    const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;

    auto compute = [&](int cur) {
        const char* xa = sX + cur * BM * ROWB + (wp * 32 * TJ + fr) * ROWB;
        const char* wa = sW + cur * BN * ROWB + (wc * 64 + fr) * ROWB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int coff = ((ks * 2 + fh) ^ fsw) << 4;
            Frag wf[TI], xf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) wf[i] = *reinterpret_cast<const Frag*>(wa + i * 32 * ROWB + coff);
#pragma unroll
            for (int j = 0; j < TJ; ++j) xf[j] = *reinterpret_cast<const Frag*>(xa + j * 32 * ROWB + coff);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) Mma<T>::run(wf[i], xf[j], acc[i][j]);
        }
    };

    // Pipeline: LDS[cur] = tile k, set A = tile k+1 (in flight / landed), set B <- tile k+2 issued
    // before computing tile k, so every global load has two compute phases to land.  Loads and
    // LDS stores are unconditional (tiles past the end are out-of-range zero loads) so the
    // compiler's counted s_waitcnt vmcnt leaves the 8 newest loads in flight.
    if constexpr (DMA) {
        // 1-deep: the DMA of tile k+1 flies during compute(k); __syncthreads() drains vmcnt(0) before the
        // barrier (hipcc treats an in-flight LDS-DMA as a pending LDS write), so no counted waits are needed
        next_tile();
        dma_tile(0);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            next_tile();
            dma_tile(cur ^ 1);
            compute(cur);
            __syncthreads();
            cur ^= 1;
        }
    } else if constexpr (PIPE == 2) {
        next_tile();
        load_tile(ra0, rw0);
        store_tile(0, ra0, rw0);
        next_tile();
        load_tile(ra0, rw0);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; kt += 2) {
            next_tile();
            load_tile(ra1, rw1);
            compute(cur);
            store_tile(cur ^ 1, ra0, rw0);
            __syncthreads();
            if (kt + 1 >= nk) break;
            next_tile();
            load_tile(ra0, rw0);
            compute(cur ^ 1);
            store_tile(cur, ra1, rw1);
            __syncthreads();
        }
    } else {   // PIPE == 1: one tile of prefetch (fewer registers)
        next_tile();
        load_tile(ra0, rw0);
        store_tile(0, ra0, rw0);
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            next_tile();
            load_tile(ra0, rw0);
            compute(cur);
            store_tile(cur ^ 1, ra0, rw0);
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue 1: acc -> (scale, shift) -> LDS staging [pixel][cout] --------
    // (the trailing __syncthreads of the K loop / prologue guarantees nobody still reads sX/sW)
    char* stage = smem;
    float* sRed = reinterpret_cast<float*>(smem + BM * SROW);   // [4 waves][2][BN]
#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = wc * 64 + i * 32 + 8 * g + 4 * fh;   // local cout of element 0
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = n0 + cl + e;
                if (co < p.Cout) {
                    if (p.scale) sc[e] = p.scale[co];
                    if (p.shift) sh[e] = p.shift[co];
                }
            }
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int px = wp * 32 * TJ + j * 32 + fr;
                T v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = from_f32<T>(acc[i][j][4 * g + e] * sc[e] + sh[e]);
                T* dst = reinterpret_cast<T*>(stage + px * SROW) + cl;
                if constexpr (sizeof(T) == 2) {
                    *reinterpret_cast<bf16x4*>(dst) = bf16x4{v[0], v[1], v[2], v[3]};
                } else {
                    *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
                }
            }
        }
    }
    __syncthreads();

    // ---- epilogue 2: row-major read back, residual/ReLU, stats, coalesced stores ----
    const int c = tid % CPR, r0 = tid / CPR;
    constexpr int RSTEP = 256 / CPR;
    const int cg = n0 + c * EPC;                         // first global cout of my chunk
    float s1[EPC], s2[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    T* yout = reinterpret_cast<T*>(p.y);
    const T* res = reinterpret_cast<const T*>(p.residual);
    const bool full = p.vec_ok && (cg + EPC <= p.Cout);
    for (int row = r0; row < BM; row += RSTEP) {
        const int m = m0 + row;
        if (m >= p.M) break;
        union { i32x4 q; T e[EPC]; } u;
        u.q = *reinterpret_cast<const i32x4*>(stage + row * SROW + c * 16);
        T* v = u.e;
        if (res != nullptr || p.relu) {
            if (res != nullptr) {
                if (full) {
                    union { i32x4 q; T e[EPC]; } ur;
                    ur.q = *reinterpret_cast<const i32x4*>(res + (size_t)m * p.ldres + cg);
                    const T* rv = ur.e;
#pragma unroll
                    for (int e = 0; e < EPC; ++e) v[e] = from_f32<T>(to_f32(v[e]) + to_f32(rv[e]));
                } else {
#pragma unroll
                    for (int e = 0; e < EPC; ++e)
                        if (cg + e < p.Cout) v[e] = from_f32<T>(to_f32(v[e]) + to_f32(res[(size_t)m * p.ldres + cg + e]));
                }
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[e] = from_f32<T>(fmaxf(to_f32(v[e]), 0.f));
            }
        }
        if (p.stats) {
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float f = to_f32(v[e]);
                s1[e] += f;
                s2[e] += f * f;
            }
        }
        if (full) {
            *reinterpret_cast<i32x4*>(yout + (size_t)m * p.ldy + cg) = u.q;
        } else {
#pragma unroll
            for (int e = 0; e < EPC; ++e)
                if (cg + e < p.Cout) yout[(size_t)m * p.ldy + cg + e] = v[e];
        }
    }
    if (p.stats) {
        // lanes holding the same chunk index: lane, lane^CPR, lane^2CPR, ...
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
#pragma unroll
            for (int o = CPR; o < 64; o <<= 1) {
                s1[e] += __shfl_xor(s1[e], o);
                s2[e] += __shfl_xor(s2[e], o);
            }
        }
        if constexpr (CPR <= 64) {
            if (lane < CPR) {
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    sRed[(wave * 2 + 0) * BN + lane * EPC + e] = s1[e];
                    sRed[(wave * 2 + 1) * BN + lane * EPC + e] = s2[e];
                }
            }
        }
        WS_LDS_BARRIER();            // LDS only: the tile's output stores keep draining
        if (tid < 2 * BN) {
            const int which = tid / BN, col = tid - which * BN;
            // waves that share a chunk column set: with CPR<=64 every wave covers all chunks
            float t = sRed[(0 * 2 + which) * BN + col] + sRed[(1 * 2 + which) * BN + col] +
                      sRed[(2 * 2 + which) * BN + col] + sRed[(3 * 2 + which) * BN + col];
            if (n0 + col < p.Cout) p.stats[((size_t)mt * 2 + which) * p.Cout + n0 + col] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 256 pixels x 256 couts, 8 waves (4 along cout x 2 along pixels, 64 x 128 outputs per wave), one
// block per CU.  bf16, linear addressing, taps outer.  The K loop keeps LDS-DMA loads in flight
// ACROSS barriers: two K tiles of LDS (2 x 64 KiB), each split into four 16-KiB half tiles
// (pixel halves XA/XB, cout halves W0/W1, rows permuted so that every wave reads its own quadrant
// from each half); a K tile is consumed in four phases (one 32x64 quadrant x K=64 each = 8 MFMAs
// per wave) and a half tile is refilled with the data of K tile t+2 in the phase after its last
// reader, so every load has >= 6 phases to land.  Waits are counted (s_waitcnt vmcnt(N), never 0
// inside the loop) and placed one phase ahead of the first read of the buffer they retire;
// barriers are raw s_barrier (no fence: __syncthreads() would drain the DMA queue).
//
// One 256x256 tile per CU quantizes badly (16 x 65 x 65 pixels = 265 tiles on 256 CUs), so the
// tiles beyond the last full round of 256 are split along K over the idle CUs:
//   MODE 0: whole tile, fused epilogue;   MODE 1: one K range of a tile -> fp32 slab (register
//   layout, 1-KiB coalesced stores), summed by conv_big_fixup_kernel in a fixed order with the
//   same fused epilogue.
constexpr int BIGT = 256;
constexpr int HT = 128 * ROWB;                               // half tile: 128 rows x 128 B
constexpr int BIG_SROW = BIGT * 2 + 16;                      // staged output row (bf16) + pad
constexpr int BIG_EPI = BIGT * BIG_SROW + 8 * 2 * BIGT * 4;  // staging + [8 waves][2][256] stats
constexpr int BIG_LDS = (BIG_EPI > 8 * HT ? BIG_EPI : 8 * HT) + 16;
constexpr int SLAB_FLOATS = BIGT * BIGT;

#define BIG_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define BIG_BARRIER() asm volatile("s_barrier" ::: "memory")

// Accumulators of one wave (64 couts x 128 pixels) as [cout half i][pixel block j][group g][4 floats].  Two MFMA shapes
// (MI355X_MICROARCH.md 'DVFS give-back' item 7: the chip may hold a higher clock on v_mfma_f32_16x16x32_bf16 than on
// 32x32x16 at equal cycles per FLOP; same LDS image, same fragment bytes, same register count):
//   M16 = false: one 32x32 accumulator per (i, j); group g = registers 4g..4g+3: couts 8g + 4*(lane>>5) + e, pixel lane&31
//   M16 = true : four 16x16 accumulators per (i, j); group g = (rb, cb) = (g>>1, g&1): couts 16rb + 4*(lane>>4) + e,
//                pixel 16cb + (lane&15)
template <bool M16> struct BigAcc;
template <> struct BigAcc<false> {
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
template <> struct BigAcc<true> {
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
// element (g, lane) of a wave's 32 x 32 sub-block -> (first of 4 couts, pixel) inside the sub-block
template <bool M16> __device__ __forceinline__ int big_sub_cout(int g, int lane) {
    return M16 ? (g >> 1) * 16 + (lane >> 4) * 4 : 8 * g + 4 * (lane >> 5);
}
template <bool M16> __device__ __forceinline__ int big_sub_px(int g, int lane) {
    return M16 ? (g & 1) * 16 + (lane & 15) : (lane & 31);
}

// SWP (needs M16): software-pipelined K loop - every wave issues the LDS reads of phase p+1 BEFORE the MFMAs of phase p, so a
// fragment read has a whole MFMA block to land and the eight waves run in lockstep (one barrier per phase) instead of two
// groups half a phase apart that take turns reading and multiplying.
template <int MODE, bool M16, bool SWP>
__global__ __launch_bounds__(512) void conv_big_kernel(ConvP p) {
    static_assert(M16 || !SWP, "the software-pipelined loop is written for the 16x16x32 shape");
    typedef bf16_t T;
    typedef Mma<T>::Frag Frag;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // SWP: 8 KiB more for the per-thread tap-visibility masks (4 rows x 512 threads), kept out of the register file in that form
    __shared__ __attribute__((aligned(16))) char smem[BIG_LDS + (SWP ? 8192 : 0)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // MODE 2 (mixed launch): the first p.n_split_blocks blocks are K ranges of the tail tiles (they are dispatched
    // first and finish early), the others are whole tiles 0 .. p.tile_begin-1 - one launch, no idle gap between the two
    const int nsb = (MODE == 2) ? p.n_split_blocks : 0;
    const bool split_blk = (MODE == 1) || (MODE == 2 && (int)blockIdx.x < nsb);
    const int unit = split_blk ? (MODE == 2 ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x))
                               : xcd_remap(blockIdx.x - nsb, gridDim.x - nsb);
    const int nsplit = split_blk ? p.ksplit : 1;
    const int tile_local = split_blk ? unit / nsplit : unit;
    const int split = split_blk ? unit - tile_local * nsplit : 0;
    const int tile = (split_blk || MODE == 0) ? p.tile_begin + tile_local : tile_local;
    int nt, mt;
    big_tile_coords(p, tile, nt, mt);
    const int m0 = mt * BIGT, n0 = nt * BIGT;
    EESEG_ACTIVE_EXIT(m0);
    const int wc = wave & 3, wp = wave >> 2;
    // fragment row / 16-byte K chunk of this lane inside a 32-row group and a K step (16 channels: chunk pair 2ks + fh;
    // M16: 32 channels, chunk quad 4k2 + fh); the XOR swizzle ((row >> 1) & 7) only sees the low 4 row bits
    const int fr = M16 ? (lane & 15) : (lane & 31), fh = M16 ? (lane >> 4) : (lane >> 5), fsw = (fr >> 1) & 7;

    BigAcc<M16> A;
    A.zero();
    auto& acc = A.a;

    {
        const int taps = p.n_gtaps ? p.n_gtaps : p.R * p.S;
        // DMA role: one wave-instruction = 8 rows x 128 B; per half tile a thread fetches rows rr and rr+64
        const int rr = wave * 8 + (lane >> 3);
        const int gchunk = (lane & 7) ^ ((rr >> 1) & 7);    // XOR swizzle on the SOURCE chunk (LDS image is lane-linear)
        // pixel rows [h][q] -> block pixel q*128 + h*64 + rr;  cout rows [i][q] -> ((rr>>5) + 2q)*64 + i*32 + (rr&31)
        // p.pointwise (1x1, stride 1, no padding: the source pixel IS the output pixel) needs no coordinates at all;
        // otherwise one division pair per thread, the other three rows (+64, +128, +192 pixels) follow by carries
        int baseA[4];
        unsigned vmask[4] = {0u, 0u, 0u, 0u};
        unsigned tapmask = 1u;
        if (p.pointwise) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = m0 + (k & 1) * 128 + (k >> 1) * 64 + rr;
                vmask[k] = m < p.M ? 1u : 0u;
                baseA[k] = m * p.Cin * 2 + gchunk * 16;
            }
        } else {
            unsigned* sMask = reinterpret_cast<unsigned*>(smem + BIG_LDS - 16);
            if (tid == 0) *sMask = 0u;
            int hb[4], wb[4], nb[4];
            {
                const int mA = m0 + rr;
                int n = mA / p.HWout;
                const int rem = mA - n * p.HWout;
                int ho = rem / p.Wout;
                int wo = rem - ho * p.Wout;
                const bool carry_ok = p.Wout >= 64;           // +64 pixels wraps at most one row (else: divide again)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {              // rows in pixel order: k = 0, 2, 1, 3
                    const int k = (kk >> 1) | ((kk & 1) << 1);
                    if (kk > 0) {
                        if (carry_ok) {
                            wo += 64;
                            if (wo >= p.Wout) { wo -= p.Wout; ++ho; }
                            if (ho >= p.Hout) { ho -= p.Hout; ++n; }
                        } else {
                            const int m = mA + kk * 64;
                            n = m / p.HWout;
                            const int r2 = m - n * p.HWout;
                            ho = r2 / p.Wout;
                            wo = r2 - ho * p.Wout;
                        }
                    }
                    if (mA + kk * 64 < p.M) {
                        hb[k] = ho * p.smul + p.off_h;
                        wb[k] = wo * p.smul + p.off_w;
                        nb[k] = n * p.Hin * p.Win;
                    } else {
                        hb[k] = -(1 << 28); wb[k] = -(1 << 28); nb[k] = 0;
                    }
                }
            }
            __syncthreads();
            if (p.n_gtaps) {                   // generalised taps: one (h, w) range check per tap and row
                for (int t = 0; t < p.n_gtaps; ++t) {
                    const int dh = p.gdh[t], dw = p.gdw[t];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((unsigned)(hb[k] + dh) < (unsigned)p.Hin && (unsigned)(wb[k] + dw) < (unsigned)p.Win) vmask[k] |= 1u << t;
                }
            } else
            // tap (r, s) is visible from a row iff r is visible along h and s along w: R + S range checks per row
            {
                unsigned hm[4] = {0u, 0u, 0u, 0u}, wm[4] = {0u, 0u, 0u, 0u};
                for (int r = 0; r < p.R; ++r)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((unsigned)(hb[k] + r * p.tstep_h) < (unsigned)p.Hin) hm[k] |= 1u << r;
                for (int s = 0; s < p.S; ++s)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((unsigned)(wb[k] + s * p.tstep_w) < (unsigned)p.Win) wm[k] |= 1u << s;
                for (int r = 0; r < p.R; ++r)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if ((hm[k] >> r) & 1u) vmask[k] |= wm[k] << (r * p.S);
            }
            unsigned mine = vmask[0] | vmask[1] | vmask[2] | vmask[3];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mine |= (unsigned)__shfl_xor((int)mine, o);
            if (lane == 0 && mine) atomicOr(sMask, mine);
            __syncthreads();
            tapmask = *sMask;
            __syncthreads();                   // sMask lies inside the epilogue staging area
#pragma unroll
            for (int k = 0; k < 4; ++k) baseA[k] = ((nb[k] + hb[k] * p.Win + wb[k]) * p.Cin) * 2 + gchunk * 16;
        }
        if constexpr (SWP) {
#pragma unroll
            for (int k = 0; k < 4; ++k) reinterpret_cast<unsigned*>(smem + BIG_LDS)[k * 512 + tid] = vmask[k];   // read back by this thread only
        }
        const int kc_steps = p.Cin / 64;
        const int nk_all = __popc(tapmask) * kc_steps;
        const int k_begin = split_blk ? (int)((long long)split * nk_all / nsplit) : 0;
        const int k_end = split_blk ? (int)((long long)(split + 1) * nk_all / nsplit) : nk_all;
        const int nk = k_end - k_begin;

        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.xbytes);
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.wbytes);

        uint32_t voffA[4], voffW[4], voffWl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int co = n0 + ((rr >> 5) + 2 * (k & 1)) * 64 + (k >> 1) * 32 + (rr & 31);
            voffW[k] = (co < p.Cout) ? (uint32_t)((co * taps * p.Cin) * 2 + gchunk * 16) : EESEG_OOB;
            voffA[k] = EESEG_OOB; voffWl[k] = EESEG_OOB;
        }
        // block-uniform iterator over the K tiles, starting at K tile k_begin.  Two K orders (p.tap_inner):
        //   0: taps outer, channel chunks inner - one gather-offset update per tap;
        //   1: channel chunks outer, taps inner - all taps of one 128-byte channel slice back to back, so a 3x3 conv
        //      re-reads that slice from L2 instead of streaming the whole input tensor once per tap from beyond L2.
        const int nvis = __popc(tapmask);
        const int rcpS = (65536 + p.S - 1) / p.S;           // tap / S for tap < 32 without a division
        unsigned rest = tapmask;
        int ci, cur_tap = 0, dtap = 0, soffA = 0, soffW = 0, remain = nk;
        bool live = true, upd = true;
        auto pop_tap = [&]() {
            cur_tap = rest ? __ffs(rest) - 1 : 0;
            rest &= rest - 1;
            if (p.n_gtaps) {
                dtap = (p.gdh[cur_tap] * p.Win + p.gdw[cur_tap]) * p.Cin * 2 + p.goff[cur_tap];
            } else {
                const int r = (cur_tap * rcpS) >> 16, s = cur_tap - r * p.S;
                dtap = ((r * p.tstep_h) * p.Win + s * p.tstep_w) * p.Cin * 2;
            }
            upd = true;
        };
        if (p.tap_inner) {
            ci = nvis ? k_begin / nvis : 0;
            for (int i = nvis ? k_begin % nvis : 0; i > 0; --i) rest &= rest - 1;
        } else {
            for (int i = k_begin / kc_steps; i > 0; --i) rest &= rest - 1;
            ci = k_begin % kc_steps - 1;
            if (ci >= 0) pop_tap(); else ci = kc_steps - 1;  // mid-tap start: that tap is current; else the first call pops
        }
        auto next_tile = [&]() {
            if (p.tap_inner) {
                if (!rest) { rest = tapmask; ++ci; }
                pop_tap();
            } else if (++ci >= kc_steps) {
                ci = 0;
                pop_tap();
            }
            if (remain-- == 0) { live = false; upd = true; }  // past the last tile: out-of-range (zero, no traffic) loads
            if (upd) {
                upd = false;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned vm = SWP ? reinterpret_cast<const unsigned*>(smem + BIG_LDS)[k * 512 + tid] : vmask[k];
                    voffA[k] = (live && ((vm >> cur_tap) & 1u)) ? (uint32_t)(baseA[k] + dtap) : EESEG_OOB;
                    if (!SWP) voffWl[k] = live ? voffW[k] : EESEG_OOB;     // SWP selects at the issue (4 registers less)
                }
            }
            soffA = ci * ROWB;
            soffW = (cur_tap * p.Cin) * 2 + ci * ROWB;
        };
        const int w8 = __builtin_amdgcn_readfirstlane(wave) * 8;
        auto dmaX = [&](int s, int h) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + (s * 4 + h) * HT + (q * 64 + w8) * ROWB), 16,
                                                         (int)voffA[h * 2 + q], soffA, 0, 0);
        };
        auto dmaW = [&](int s, int i) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(smem + (s * 4 + 2 + i) * HT + (q * 64 + w8) * ROWB), 16,
                                                         (int)voffWl[i * 2 + q], soffW, 0, 0);
        };
        // M16: f[rb * 2 + k2] = rows 16rb .. 16rb+15 of the 32-row group, channels 32k2 .. 32k2+31
        auto rdW = [&](const char* sb, int i, Frag (&f)[4]) {
            const char* a = sb + (2 + i) * HT + (wc * 32 + fr) * ROWB;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                f[ks] = M16 ? *reinterpret_cast<const Frag*>(a + (ks >> 1) * 16 * ROWB + ((((ks & 1) * 4 + fh) ^ fsw) << 4))
                            : *reinterpret_cast<const Frag*>(a + (((ks * 2 + fh) ^ fsw) << 4));
        };
        auto rdX = [&](const char* sb, int h, int jj, Frag (&f)[4]) {
            const char* a = sb + h * HT + (wp * 64 + jj * 32 + fr) * ROWB;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                f[ks] = M16 ? *reinterpret_cast<const Frag*>(a + (ks >> 1) * 16 * ROWB + ((((ks & 1) * 4 + fh) ^ fsw) << 4))
                            : *reinterpret_cast<const Frag*>(a + (((ks * 2 + fh) ^ fsw) << 4));
        };
// MFMA number n (0..15) of a 16x16x32 phase: K step k2 = n>>3 outermost, so the 8 accumulators of the phase are each
// touched once per K step (no back-to-back MFMAs on one accumulator).  The LDS-DMA issues of a phase sit behind MFMA pairs
// (after n = 1, 3, 5, 7); one issue per MFMA (n = 0..3) measured 1-3 % slower, one per three or four MFMAs the same.
#define EESEG_M16(W_, X0_, X1_, I_, J0_, n_) { \
            constexpr int k2_ = (n_) >> 3, rb_ = ((n_) >> 2) & 1, xs_ = ((n_) >> 1) & 1, cb_ = (n_) & 1; \
            acc[I_][(J0_) + xs_][rb_ * 2 + cb_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16( \
                W_[rb_ * 2 + k2_], (xs_ ? X1_ : X0_)[cb_ * 2 + k2_], acc[I_][(J0_) + xs_][rb_ * 2 + cb_], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0); }

        // Two wave groups (waves 0-3 / 4-7 = one wave per SIMD each) run the phase sequence
        //   [refill DMAs | LDS reads | counted wait] barrier [8 MFMAs] barrier
        // half a phase apart (group 1 takes one extra barrier up front, group 0 one at the end), so between any
        // two barriers one group reads LDS while the other owns the MFMA pipes.  Consequences for the waits:
        //   * a half tile last read in phase P is refilled at the start of phase P+2 (by then the lagging
        //     group has consumed its reads);
        //   * the wait that retires the loads of phase Q sits in phase Q-1 after the reads, before the barrier.
        // Issue order per iteration: phase 1: XB(t+1) | phase 3: XA(t+2), W0(t+2) | phase 4: W1(t+2).
        // The read slots hold nothing but LDS reads and the counted wait: every DMA issue and the K iterator sit INSIDE the
        // MFMA blocks, pinned between the MFMAs (the wave has issue slack there; one K tile costs 2 x the sum of the four
        // slot times).  A half tile last read in slot P is refilled in MFMA block P+1: by then the lagging
        // group, half a phase behind, has consumed its slot-P reads too.  Issue order per iteration t (all for K tile t+2,
        // stage s):  MFMA block 2: XA, W0 | block 3: W1 | block 4: XB;  waits (in slot P-1 for the reads of slot P) as before.
        if constexpr (!SWP) {
        next_tile(); dmaX(0, 0); dmaW(0, 0); dmaW(0, 1); dmaX(0, 1);      // K tile 0
        next_tile(); dmaX(1, 0); dmaW(1, 0); dmaW(1, 1); dmaX(1, 1);      // K tile 1
        BIG_WAIT(8);
        BIG_BARRIER();
        const bool lagging = __builtin_amdgcn_readfirstlane(wave) >= 4;
        if (lagging) BIG_BARRIER();
        auto dma1 = [&](const __amdgpu_buffer_rsrc_t& rs, int half, int q, uint32_t voff, int soff, int s) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (s * 4 + half) * HT + (q * 64 + w8) * ROWB), 16,
                                                     (int)voff, soff, 0, 0);
        };
#define EESEG_MM(a_, b_, c_) { Mma<T>::run(a_, b_, c_); __builtin_amdgcn_sched_barrier(0); }
        for (int t = 0; t < nk; ++t) {
            const int s = t & 1;
            const char* sb = smem + s * 4 * HT;
            Frag w0[4], w1[4], xa0[4], xa1[4], xb0[4], xb1[4];
            // phase 1: quadrant (W0, XA)
            rdW(sb, 0, w0); rdX(sb, 0, 0, xa0); rdX(sb, 0, 1, xa1);
            BIG_WAIT(10);                  // W1(t) landed
            BIG_BARRIER();
            __builtin_amdgcn_s_setprio(1);
            if constexpr (M16) {
                EESEG_M16(w0, xa0, xa1, 0, 0, 0) EESEG_M16(w0, xa0, xa1, 0, 0, 1) EESEG_M16(w0, xa0, xa1, 0, 0, 2) EESEG_M16(w0, xa0, xa1, 0, 0, 3)
                EESEG_M16(w0, xa0, xa1, 0, 0, 4) EESEG_M16(w0, xa0, xa1, 0, 0, 5) EESEG_M16(w0, xa0, xa1, 0, 0, 6) EESEG_M16(w0, xa0, xa1, 0, 0, 7)
                EESEG_M16(w0, xa0, xa1, 0, 0, 8) EESEG_M16(w0, xa0, xa1, 0, 0, 9) EESEG_M16(w0, xa0, xa1, 0, 0, 10) EESEG_M16(w0, xa0, xa1, 0, 0, 11)
                EESEG_M16(w0, xa0, xa1, 0, 0, 12) EESEG_M16(w0, xa0, xa1, 0, 0, 13) EESEG_M16(w0, xa0, xa1, 0, 0, 14) EESEG_M16(w0, xa0, xa1, 0, 0, 15)
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    Mma<T>::run(w0[ks], xa0[ks], acc[0][0]);
                    Mma<T>::run(w0[ks], xa1[ks], acc[0][1]);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            next_tile();                   // -> K tile t+2 (in the shadow of the MFMAs just issued)
            BIG_BARRIER();
            // phase 2: quadrant (W1, XA); XA(s), W0(s) were last read in slot 1: refill them between these MFMAs
            rdW(sb, 1, w1);
            BIG_WAIT(8);                   // XB(t) landed
            BIG_BARRIER();
            __builtin_amdgcn_s_setprio(1);
            if constexpr (M16) {
                EESEG_M16(w1, xa0, xa1, 1, 0, 0) EESEG_M16(w1, xa0, xa1, 1, 0, 1) dma1(rx, 0, 0, voffA[0], soffA, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w1, xa0, xa1, 1, 0, 2) EESEG_M16(w1, xa0, xa1, 1, 0, 3) dma1(rx, 0, 1, voffA[1], soffA, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w1, xa0, xa1, 1, 0, 4) EESEG_M16(w1, xa0, xa1, 1, 0, 5) dma1(rw, 2, 0, voffWl[0], soffW, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w1, xa0, xa1, 1, 0, 6) EESEG_M16(w1, xa0, xa1, 1, 0, 7) dma1(rw, 2, 1, voffWl[1], soffW, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w1, xa0, xa1, 1, 0, 8) EESEG_M16(w1, xa0, xa1, 1, 0, 9) EESEG_M16(w1, xa0, xa1, 1, 0, 10) EESEG_M16(w1, xa0, xa1, 1, 0, 11)
                EESEG_M16(w1, xa0, xa1, 1, 0, 12) EESEG_M16(w1, xa0, xa1, 1, 0, 13) EESEG_M16(w1, xa0, xa1, 1, 0, 14) EESEG_M16(w1, xa0, xa1, 1, 0, 15)
            } else {
                EESEG_MM(w1[0], xa0[0], acc[1][0]) dma1(rx, 0, 0, voffA[0], soffA, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_MM(w1[0], xa1[0], acc[1][1]) dma1(rx, 0, 1, voffA[1], soffA, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_MM(w1[1], xa0[1], acc[1][0]) dma1(rw, 2, 0, voffWl[0], soffW, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_MM(w1[1], xa1[1], acc[1][1]) dma1(rw, 2, 1, voffWl[1], soffW, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_MM(w1[2], xa0[2], acc[1][0]) EESEG_MM(w1[2], xa1[2], acc[1][1])
                EESEG_MM(w1[3], xa0[3], acc[1][0]) EESEG_MM(w1[3], xa1[3], acc[1][1])
            }
            __builtin_amdgcn_s_setprio(0);
            BIG_BARRIER();
            // phase 3: quadrant (W1, XB); W1(s) was last read in slot 2
            rdX(sb, 1, 0, xb0); rdX(sb, 1, 1, xb1);
            BIG_BARRIER();
            __builtin_amdgcn_s_setprio(1);
            if constexpr (M16) {
                EESEG_M16(w1, xb0, xb1, 1, 2, 0) EESEG_M16(w1, xb0, xb1, 1, 2, 1) dma1(rw, 3, 0, voffWl[2], soffW, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w1, xb0, xb1, 1, 2, 2) EESEG_M16(w1, xb0, xb1, 1, 2, 3) dma1(rw, 3, 1, voffWl[3], soffW, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w1, xb0, xb1, 1, 2, 4) EESEG_M16(w1, xb0, xb1, 1, 2, 5) EESEG_M16(w1, xb0, xb1, 1, 2, 6) EESEG_M16(w1, xb0, xb1, 1, 2, 7)
                EESEG_M16(w1, xb0, xb1, 1, 2, 8) EESEG_M16(w1, xb0, xb1, 1, 2, 9) EESEG_M16(w1, xb0, xb1, 1, 2, 10) EESEG_M16(w1, xb0, xb1, 1, 2, 11)
                EESEG_M16(w1, xb0, xb1, 1, 2, 12) EESEG_M16(w1, xb0, xb1, 1, 2, 13) EESEG_M16(w1, xb0, xb1, 1, 2, 14) EESEG_M16(w1, xb0, xb1, 1, 2, 15)
            } else {
                EESEG_MM(w1[0], xb0[0], acc[1][2]) dma1(rw, 3, 0, voffWl[2], soffW, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_MM(w1[0], xb1[0], acc[1][3]) dma1(rw, 3, 1, voffWl[3], soffW, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_MM(w1[1], xb0[1], acc[1][2]) EESEG_MM(w1[1], xb1[1], acc[1][3])
                EESEG_MM(w1[2], xb0[2], acc[1][2]) EESEG_MM(w1[2], xb1[2], acc[1][3])
                EESEG_MM(w1[3], xb0[3], acc[1][2]) EESEG_MM(w1[3], xb1[3], acc[1][3])
            }
            __builtin_amdgcn_s_setprio(0);
            BIG_BARRIER();
            // phase 4: quadrant (W0, XB) from registers; XB(s) was last read in slot 3
            BIG_WAIT(10);                  // XA(t+1), W0(t+1) landed
            BIG_BARRIER();
            __builtin_amdgcn_s_setprio(1);
            if constexpr (M16) {
                EESEG_M16(w0, xb0, xb1, 0, 2, 0) EESEG_M16(w0, xb0, xb1, 0, 2, 1) dma1(rx, 1, 0, voffA[2], soffA, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w0, xb0, xb1, 0, 2, 2) EESEG_M16(w0, xb0, xb1, 0, 2, 3) dma1(rx, 1, 1, voffA[3], soffA, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w0, xb0, xb1, 0, 2, 4) EESEG_M16(w0, xb0, xb1, 0, 2, 5) EESEG_M16(w0, xb0, xb1, 0, 2, 6) EESEG_M16(w0, xb0, xb1, 0, 2, 7)
                EESEG_M16(w0, xb0, xb1, 0, 2, 8) EESEG_M16(w0, xb0, xb1, 0, 2, 9) EESEG_M16(w0, xb0, xb1, 0, 2, 10) EESEG_M16(w0, xb0, xb1, 0, 2, 11)
                EESEG_M16(w0, xb0, xb1, 0, 2, 12) EESEG_M16(w0, xb0, xb1, 0, 2, 13) EESEG_M16(w0, xb0, xb1, 0, 2, 14) EESEG_M16(w0, xb0, xb1, 0, 2, 15)
            } else {
                EESEG_MM(w0[0], xb0[0], acc[0][2]) dma1(rx, 1, 0, voffA[2], soffA, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_MM(w0[0], xb1[0], acc[0][3]) dma1(rx, 1, 1, voffA[3], soffA, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_MM(w0[1], xb0[1], acc[0][2]) EESEG_MM(w0[1], xb1[1], acc[0][3])
                EESEG_MM(w0[2], xb0[2], acc[0][2]) EESEG_MM(w0[2], xb1[2], acc[0][3])
                EESEG_MM(w0[3], xb0[3], acc[0][2]) EESEG_MM(w0[3], xb1[3], acc[0][3])
            }
            __builtin_amdgcn_s_setprio(0);
            BIG_BARRIER();
        }
#undef EESEG_MM
        if (!lagging) BIG_BARRIER();
        } else {
        auto dma1s = [&](const __amdgpu_buffer_rsrc_t& rs, int half, int q, uint32_t voff, int soff, int s) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + (s * 4 + half) * HT + (q * 64 + w8) * ROWB), 16,
                                                     (int)voff, soff, 0, 0);
        };
        for (int s0 = 0; s0 < 2; ++s0) {                                  // K tiles 0 and 1, in the loop's issue order
            next_tile();
            dma1s(rx, 0, 0, voffA[0], soffA, s0); dma1s(rx, 0, 1, voffA[1], soffA, s0);
#pragma unroll
            for (int k = 0; k < 4; ++k) dma1s(rw, 2 + (k >> 1), k & 1, live ? voffW[k] : EESEG_OOB, soffW, s0);
            dma1s(rx, 1, 0, voffA[2], soffA, s0); dma1s(rx, 1, 1, voffA[3], soffA, s0);
        }
#define BIG_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define EESEG_M16x4(W_, X0_, X1_, I_, J0_, n_) EESEG_M16(W_, X0_, X1_, I_, J0_, n_) EESEG_M16(W_, X0_, X1_, I_, J0_, (n_) + 1) \
        EESEG_M16(W_, X0_, X1_, I_, J0_, (n_) + 2) EESEG_M16(W_, X0_, X1_, I_, J0_, (n_) + 3)
        // Per wave, DMA issue order per K tile: XA XA W0 W0 W1 W1 XB XB (prologue and loop alike), so with vmcnt retiring in
        // order: W1(t) has 10 younger loads at the top of phase 1, XB(t) 12 at phase 2, {XA, W0}(t+1) 12 at phase 4.
        // A buffer is refilled (tile t+2) in the MFMA block of the phase AFTER the one whose top issued its last reads: every
        // wave passes "s_waitcnt lgkmcnt(0); s_barrier" in between.
        Frag wA[4], wB[4], xa0[4], xa1[4], xb0[4], xb1[4];
        // fragment reads with the lane coordinates made opaque at every use: the LDS addresses are then recomputed (3 VALU
        // ops per read) instead of ~20 loop-invariant addresses living in registers next to 96 fragment + 128 accumulator VGPRs
        auto rd2 = [&](const char* base, int row0, Frag (&f)[4]) {
            int r_ = fr, h_ = fh;
            asm volatile("" : "+v"(r_), "+v"(h_));
            const int sw_ = (r_ >> 1) & 7;
            const char* a = base + (row0 + r_) * ROWB;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                f[ks] = *reinterpret_cast<const Frag*>(a + (ks >> 1) * 16 * ROWB + ((((ks & 1) * 4 + h_) ^ sw_) << 4));
        };
        auto rdW2 = [&](const char* sb, int i, Frag (&f)[4]) { rd2(sb + (2 + i) * HT, wc * 32, f); };
        auto rdX2 = [&](const char* sb, int h, int jj, Frag (&f)[4]) { rd2(sb + h * HT, wp * 64 + jj * 32, f); };
        BIG_WAIT(12);                      // XA(0), W0(0) landed
        BIG_BARRIER();
        rdW2(smem, 0, wA); rdX2(smem, 0, 0, xa0); rdX2(smem, 0, 1, xa1);
        __builtin_amdgcn_sched_barrier(0);
        // one K tile; on entry w0 / xa hold W0(t) / XA(t), on exit w1 holds W0(t+1) and xa XA(t+1) (the caller swaps w0 and w1)
        // single fragment reads, pinned BETWEEN the MFMAs of a block (a ds_read_b128 per MFMA gap is nearly free, MI355X_MICROARCH.md
        // LDS): o0 / o1 = the lane's swizzled 16-byte chunk offsets of the two K halves, rp() = the lane's row in a half tile
        int o0, o1;
        auto rp = [&](const char* base, int row0) -> const char* {
            int r_ = fr, h_ = fh;
            asm volatile("" : "+v"(r_), "+v"(h_));
            const int sw_ = (r_ >> 1) & 7;
            o0 = (h_ ^ sw_) << 4;
            o1 = ((4 + h_) ^ sw_) << 4;
            return base + (row0 + r_) * ROWB;
        };
#define EESEG_RDF(dst_, a_, ks_) { dst_[ks_] = *reinterpret_cast<const Frag*>((a_) + ((ks_) >> 1) * 16 * ROWB + (((ks_) & 1) ? o1 : o0)); \
            __builtin_amdgcn_sched_barrier(0); }
        // one K tile; on entry w0 / xa hold W0(t) / XA(t), on exit w1 holds W0(t+1) and xa XA(t+1) (the caller swaps w0 and w1)
        auto ktile = [&](int t, Frag (&w0)[4], Frag (&w1)[4]) {
            const int s = t & 1;
            const char* sb = smem + s * 4 * HT;
            const char* sn = smem + (s ^ 1) * 4 * HT;
            // ---- phase 1: read W1(t); multiply (W0, XA); refill XA, W0 of this stage
            BIG_WAIT(10);
            BIG_LGKM0();
            BIG_BARRIER();
            {
                const char* aw = rp(sb + 3 * HT, wc * 32);
                next_tile();                   // -> K tile t+2
                __builtin_amdgcn_s_setprio(1);
                EESEG_M16(w0, xa0, xa1, 0, 0, 0) EESEG_RDF(w1, aw, 0) EESEG_M16(w0, xa0, xa1, 0, 0, 1) dma1s(rx, 0, 0, voffA[0], soffA, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w0, xa0, xa1, 0, 0, 2) EESEG_RDF(w1, aw, 1) EESEG_M16(w0, xa0, xa1, 0, 0, 3) dma1s(rx, 0, 1, voffA[1], soffA, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w0, xa0, xa1, 0, 0, 4) EESEG_RDF(w1, aw, 2) EESEG_M16(w0, xa0, xa1, 0, 0, 5) dma1s(rw, 2, 0, (live ? voffW[0] : EESEG_OOB), soffW, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w0, xa0, xa1, 0, 0, 6) EESEG_RDF(w1, aw, 3) EESEG_M16(w0, xa0, xa1, 0, 0, 7) dma1s(rw, 2, 1, (live ? voffW[1] : EESEG_OOB), soffW, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16x4(w0, xa0, xa1, 0, 0, 8) EESEG_M16x4(w0, xa0, xa1, 0, 0, 12)
                __builtin_amdgcn_s_setprio(0);
            }
            // ---- phase 2: read XB(t); multiply (W1, XA); refill W1
            BIG_WAIT(12);
            BIG_LGKM0();
            BIG_BARRIER();
            {
                const char* a0 = rp(sb + HT, wp * 64);
                const char* a1 = a0 + 32 * ROWB;
                __builtin_amdgcn_s_setprio(1);
                EESEG_M16(w1, xa0, xa1, 1, 0, 0) EESEG_RDF(xb0, a0, 0) EESEG_M16(w1, xa0, xa1, 1, 0, 1) EESEG_RDF(xb0, a0, 1)
                dma1s(rw, 3, 0, (live ? voffW[2] : EESEG_OOB), soffW, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w1, xa0, xa1, 1, 0, 2) EESEG_RDF(xb0, a0, 2) EESEG_M16(w1, xa0, xa1, 1, 0, 3) EESEG_RDF(xb0, a0, 3)
                dma1s(rw, 3, 1, (live ? voffW[3] : EESEG_OOB), soffW, s); __builtin_amdgcn_sched_barrier(0);
                EESEG_M16(w1, xa0, xa1, 1, 0, 4) EESEG_RDF(xb1, a1, 0) EESEG_M16(w1, xa0, xa1, 1, 0, 5) EESEG_RDF(xb1, a1, 1)
                EESEG_M16(w1, xa0, xa1, 1, 0, 6) EESEG_RDF(xb1, a1, 2) EESEG_M16(w1, xa0, xa1, 1, 0, 7) EESEG_RDF(xb1, a1, 3)
                EESEG_M16x4(w1, xa0, xa1, 1, 0, 8) EESEG_M16x4(w1, xa0, xa1, 1, 0, 12)
                __builtin_amdgcn_s_setprio(0);
            }
            // ---- phase 3: nothing to read; multiply (W1, XB); refill XB
            BIG_LGKM0();
            BIG_BARRIER();
            __builtin_amdgcn_s_setprio(1);
            EESEG_M16(w1, xb0, xb1, 1, 2, 0) EESEG_M16(w1, xb0, xb1, 1, 2, 1) dma1s(rx, 1, 0, voffA[2], soffA, s); __builtin_amdgcn_sched_barrier(0);
            EESEG_M16(w1, xb0, xb1, 1, 2, 2) EESEG_M16(w1, xb0, xb1, 1, 2, 3) dma1s(rx, 1, 1, voffA[3], soffA, s); __builtin_amdgcn_sched_barrier(0);
            EESEG_M16x4(w1, xb0, xb1, 1, 2, 4) EESEG_M16x4(w1, xb0, xb1, 1, 2, 8) EESEG_M16x4(w1, xb0, xb1, 1, 2, 12)
            __builtin_amdgcn_s_setprio(0);
            // ---- phase 4: read W0(t+1) into w1's registers and XA(t+1); multiply (W0, XB)
            BIG_WAIT(12);
            BIG_BARRIER();
            {
                const char* aw = rp(sn + 2 * HT, wc * 32);
                const char* a0 = sn + (aw - (sn + 2 * HT)) - (wc * 32) * ROWB + (wp * 64) * ROWB;     // same lane row, XA half tile
                const char* a1 = a0 + 32 * ROWB;
                __builtin_amdgcn_s_setprio(1);
                EESEG_M16(w0, xb0, xb1, 0, 2, 0) EESEG_RDF(w1, aw, 0) EESEG_M16(w0, xb0, xb1, 0, 2, 1) EESEG_RDF(w1, aw, 1)
                EESEG_M16(w0, xb0, xb1, 0, 2, 2) EESEG_RDF(w1, aw, 2) EESEG_M16(w0, xb0, xb1, 0, 2, 3) EESEG_RDF(w1, aw, 3)
                EESEG_M16(w0, xb0, xb1, 0, 2, 4) EESEG_RDF(xa0, a0, 0) EESEG_M16(w0, xb0, xb1, 0, 2, 5) EESEG_RDF(xa0, a0, 1)
                EESEG_M16(w0, xb0, xb1, 0, 2, 6) EESEG_RDF(xa0, a0, 2) EESEG_M16(w0, xb0, xb1, 0, 2, 7) EESEG_RDF(xa0, a0, 3)
                EESEG_M16(w0, xb0, xb1, 0, 2, 8) EESEG_RDF(xa1, a1, 0) EESEG_M16(w0, xb0, xb1, 0, 2, 9) EESEG_RDF(xa1, a1, 1)
                EESEG_M16(w0, xb0, xb1, 0, 2, 10) EESEG_RDF(xa1, a1, 2) EESEG_M16(w0, xb0, xb1, 0, 2, 11) EESEG_RDF(xa1, a1, 3)
                EESEG_M16x4(w0, xb0, xb1, 0, 2, 12)
                __builtin_amdgcn_s_setprio(0);
            }
        };
        for (int t = 0; t < nk; t += 2) {
            ktile(t, wA, wB);
            if (t + 1 < nk) ktile(t + 1, wB, wA);
        }
        BIG_LGKM0();
#undef EESEG_M16x4
#undef EESEG_RDF
#undef BIG_LGKM0
        }
#undef EESEG_M16
        BIG_WAIT(0);                       // trailing out-of-range DMAs still write (zeros) into LDS
        BIG_BARRIER();
    }

    if (split_blk) {                       // partial sums of this K range, register layout: 1 KiB per wave store
        if (p.stats && split == 0) {       // the fix-up kernel adds its 32-pixel slices into this tile's two stat rows
            const int which = tid >> 8, col = tid & 255;
            p.stats[((size_t)(2 * mt) * 2 + which) * p.Cout + n0 + col] = 0.f;
            if (2 * mt + 1 < p.m_tiles) p.stats[((size_t)(2 * mt + 1) * 2 + which) * p.Cout + n0 + col] = 0.f;
        }
        float* ws = p.slabs + ((size_t)tile_local * nsplit + split) * SLAB_FLOATS + (wave * 32) * 256 + lane * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(ws + ((i * 4 + j) * 4 + g) * 256) = A.get4(i, j, g);
        return;
    }

    // ---- epilogue 1: acc -> (scale, shift) -> LDS staging [pixel][cout] --------
    // (this kernel only runs layers with Cout % 256 == 0 and 16-byte row stores: no column guards, no scalar tails)
    char* stage = smem;
    float* sRed = reinterpret_cast<float*>(smem + BIGT * BIG_SROW);   // [8 waves][2][256]
    const bool affine = p.scale != nullptr || p.shift != nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = wc * 64 + i * 32 + big_sub_cout<M16>(g, lane);
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (affine) {
                if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n0 + cl);
                if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n0 + cl);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int px = wp * 128 + j * 32 + big_sub_px<M16>(g, lane);
                T v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = from_f32<T>(affine ? A.get(i, j, g, e) * sc[e] + sh[e] : A.get(i, j, g, e));
                // 8-byte XOR on rows 8..15 (mod 16): lanes fr and fr+8 of a 16-lane store group would share banks
                *reinterpret_cast<bf16x4*>(stage + px * BIG_SROW + ((cl * 2) ^ (((px >> 3) & 1) << 3))) = bf16x4{v[0], v[1], v[2], v[3]};
            }
        }
    }
    __syncthreads();

    // ---- epilogue 2: row-major read back, residual/ReLU, stats, coalesced 16-byte stores ----
    const int c = tid & 31, r0 = tid >> 5;
    const int cg = n0 + c * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    T* yout = reinterpret_cast<T*>(p.y);
    const T* res = reinterpret_cast<const T*>(p.residual);
    const bool post = res != nullptr || p.relu;            // block-uniform: the training forward has neither
    i32x4 rq[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {                      // all LDS reads first (the accumulators are dead: registers are free)
        const int row = r0 + 16 * it;
        const i32x4 q = *reinterpret_cast<const i32x4*>(stage + row * BIG_SROW + c * 16);
        rq[it] = ((row >> 3) & 1) ? i32x4{q[2], q[3], q[0], q[1]} : q;
    }
    if (post) {
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int m = m0 + r0 + 16 * it;
            if (m < p.M) {
                union { i32x4 q; T e[8]; } u;
                u.q = rq[it];
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = to_f32(u.e[e]);
                if (res != nullptr) {
                    union { i32x4 q; T e[8]; } ur;
                    ur.q = *reinterpret_cast<const i32x4*>(res + (size_t)m * p.ldres + cg);
                    if (p.resmask) ur.q = mask_chunk_bf16(ur.q, p.resmask[(size_t)m * p.ldmask + (cg >> 3)]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = to_f32(from_f32<T>(f[e] + to_f32(ur.e[e])));
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) u.e[e] = from_f32<T>(f[e]);
                rq[it] = u.q;
            }
        }
    }
    // the stores first (they drain while the statistics are summed), one 64-bit pointer bump per row
    {
        T* yrow = yout + (size_t)(m0 + r0) * p.ldy + cg;
        const size_t ystep = (size_t)16 * p.ldy;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            if (m0 + r0 + 16 * it < p.M) *reinterpret_cast<i32x4*>(yrow) = rq[it];
            yrow += ystep;
        }
    }
    if (p.stats) {                                         // of the values as stored (bf16); packed fp32 math, 2 couts per op
        f32x2 a1[4], a2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a1[k] = f32x2{0.f, 0.f}; a2[k] = f32x2{0.f, 0.f}; }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            if (m0 + r0 + 16 * it < p.M) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned w = (unsigned)rq[it][k];
                    const f32x2 f = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                    a1[k] += f;
                    a2[k] += f * f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s1[2 * k] = a1[k][0]; s1[2 * k + 1] = a1[k][1];
            s2[2 * k] = a2[k][0]; s2[2 * k + 1] = a2[k][1];
        }
    }
    if (p.stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s1[e] += __shfl_xor(s1[e], 32);
            s2[e] += __shfl_xor(s2[e], 32);
        }
        if (lane < 32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sRed[(wave * 2 + 0) * BIGT + lane * 8 + e] = s1[e];
                sRed[(wave * 2 + 1) * BIGT + lane * 8 + e] = s2[e];
            }
        }
        WS_LDS_BARRIER();            // LDS only: the tile's output stores keep draining
        const int which = tid >> 8, col = tid & 255;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += sRed[(w * 2 + which) * BIGT + col];
        // the partial-sum buffer has one row pair per 128 pixels (eeseg_conv_stats_tiles): this tile owns two
        p.stats[((size_t)(2 * mt) * 2 + which) * p.Cout + n0 + col] = t;
        if (2 * mt + 1 < p.m_tiles) p.stats[((size_t)(2 * mt + 1) * 2 + which) * p.Cout + n0 + col] = 0.f;
    }
}

// Fix-up of the K-split tiles: one block per 32-pixel slice of a tile (8 per tile, 4 waves = the 4 cout
// groups): sums the slabs in a fixed order, then the fused epilogue; BN partial sums of the 4 slices that
// share a 128-pixel stat row are combined with fp32 atomics (rows zeroed by the MODE 1 kernel).
template <bool M16>
__global__ __launch_bounds__(256) void conv_big_fixup_kernel(ConvP p) {
    typedef bf16_t T;
    constexpr int SL = 32;                                   // pixels per slice
    __shared__ __attribute__((aligned(16))) char smem[SL * BIG_SROW + 4 * 2 * BIGT * 4];
    const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
    const int tile_local = blockIdx.x >> 3, slice = blockIdx.x & 7;
    const int wp = slice >> 2, j = slice & 3;
    const int tile = p.tile_begin + tile_local;
    int nt, mt;
    big_tile_coords(p, tile, nt, mt);
    const int m0 = mt * BIGT + wp * 128 + j * SL, n0 = nt * BIGT;
    EESEG_ACTIVE_EXIT(mt * BIGT);            // same decision as the blocks that would have filled this tile's slabs
    const int fr = lane & 31, fh = lane >> 5;

    f32x4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* ws = p.slabs + (size_t)tile_local * p.ksplit * SLAB_FLOATS + ((wp * 4 + wc) * 32 + j * 4) * 256 + lane * 4;
    // four K ranges per iteration: 32 independent 16-byte loads in flight per lane (one range at a time made this kernel a
    // chain of ksplit load latencies); the additions keep the order of the ranges
    int s = 0;
    for (; s + 4 <= p.ksplit; s += 4) {
        f32x4 t[4][2][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    t[u][i][g] = *reinterpret_cast<const f32x4*>(ws + (size_t)(s + u) * SLAB_FLOATS + (i * 16 + g) * 256);
        __builtin_amdgcn_sched_barrier(0);     // all 32 loads issued before the first add (else the scheduler trades them for registers)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[i][g] += t[u][i][g];
    }
    for (; s < p.ksplit; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                acc[i][g] += *reinterpret_cast<const f32x4*>(ws + (size_t)s * SLAB_FLOATS + (i * 16 + g) * 256);
    }
    char* stage = smem;
    float* sRed = reinterpret_cast<float*>(smem + SL * BIG_SROW);   // [4 waves][2][256]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = wc * 64 + i * 32 + big_sub_cout<M16>(g, lane);      // the producing kernel's accumulator layout
            T v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = n0 + cl + e;
                const float sc = (p.scale && co < p.Cout) ? p.scale[co] : 1.f;
                const float sh = (p.shift && co < p.Cout) ? p.shift[co] : 0.f;
                v[e] = from_f32<T>(acc[i][g][e] * sc + sh);
            }
            *reinterpret_cast<bf16x4*>(reinterpret_cast<T*>(stage + big_sub_px<M16>(g, lane) * BIG_SROW) + cl) = bf16x4{v[0], v[1], v[2], v[3]};
        }
    __syncthreads();
    const int c = tid & 31, r0 = tid >> 5;
    const int cg = n0 + c * 8;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    T* yout = reinterpret_cast<T*>(p.y);
    const T* res = reinterpret_cast<const T*>(p.residual);
    const bool full = p.vec_ok && (cg + 8 <= p.Cout);
    for (int row = r0; row < SL; row += 8) {
        const int m = m0 + row;
        if (m >= p.M) break;
        union { i32x4 q; T e[8]; } u;
        u.q = *reinterpret_cast<const i32x4*>(stage + row * BIG_SROW + c * 16);
        T* v = u.e;
        if (res != nullptr) {
            if (full) {
                union { i32x4 q; T e[8]; } ur;
                ur.q = *reinterpret_cast<const i32x4*>(res + (size_t)m * p.ldres + cg);
                if (p.resmask) ur.q = mask_chunk_bf16(ur.q, p.resmask[(size_t)m * p.ldmask + (cg >> 3)]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(to_f32(v[e]) + to_f32(ur.e[e]));
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (cg + e < p.Cout) v[e] = from_f32<T>(to_f32(v[e]) + to_f32(res[(size_t)m * p.ldres + cg + e]));
            }
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(fmaxf(to_f32(v[e]), 0.f));
        }
        if (p.stats) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = to_f32(v[e]);
                s1[e] += f;
                s2[e] += f * f;
            }
        }
        if (full) {
            *reinterpret_cast<i32x4*>(yout + (size_t)m * p.ldy + cg) = u.q;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (cg + e < p.Cout) yout[(size_t)m * p.ldy + cg + e] = v[e];
        }
    }
    if (p.stats) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s1[e] += __shfl_xor(s1[e], 32);
            s2[e] += __shfl_xor(s2[e], 32);
        }
        if (lane < 32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sRed[(wc * 2 + 0) * BIGT + lane * 8 + e] = s1[e];
                sRed[(wc * 2 + 1) * BIGT + lane * 8 + e] = s2[e];
            }
        }
        WS_LDS_BARRIER();            // LDS only: the tile's output stores keep draining
        const int col = tid;                                  // 256 threads = 256 couts, both sums
        const int srow = 2 * mt + wp;
        if (n0 + col < p.Cout && srow < p.m_tiles) {
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const float t = sRed[(0 * 2 + which) * BIGT + col] + sRed[(1 * 2 + which) * BIGT + col] +
                                sRed[(2 * 2 + which) * BIGT + col] + sRed[(3 * 2 + which) * BIGT + col];
                atomicAdd(&p.stats[((size_t)srow * 2 + which) * p.Cout + n0 + col], t);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Pointwise (1x1, stride 1, no padding) bf16 convolution for the layers that sit BELOW the bf16 ridge
// (ResNet bottleneck 1x1s at 65x65: K = 256..1024 - HBM-bound).  The 256x256 kernel above runs
// one 160-KiB block per CU, so nothing overlaps a tile's epilogue (accumulators -> LDS -> 16-byte row
// stores, ~4 us) and its first loads: with 4..16 K tiles per tile that fixed cost is half the run time.
// Here a tile is 128 pixels x 256 couts on 4 waves (64 couts x 128 pixels per wave: the same per-wave
// shape, so the same 24 B/clk of LDS fragment reads per MFMA-bound wave) and a block needs 74 KiB of
// LDS and <= 256 VGPRs -> TWO blocks per CU: while one streams its output, the other owns the MFMA pipes
// and keeps loads in flight.  (Tried and dropped: separate rings for activations (6 deep, issued by waves 0-1) and weights
// (2 deep, waves 2-3) so that HBM-latency and L2-latency loads do not share one in-order vmcnt: 256->1024 133 -> 140 us,
// 1024->256 127 us vs 103 on the 256-tile kernel - the weight tile's round trip is then exposed on every K tile.)
// K tile = 32 channels (64-byte LDS rows), three stages of LDS-DMA
// (buffer_load ... lds) in flight across one raw barrier per K tile; XOR swizzle on the SOURCE chunk
// (chunk ^ ((row >> 2) & 3): conflict-free ds_read_b128 for a 64-byte row pitch, MI355X_MICROARCH.md LDS).
constexpr int PW_BM = 128, PW_BN = 256;
constexpr int PW_ROW = 64;                                   // bytes of K per LDS row (32 bf16 channels)
constexpr int PW_XS = PW_BM * PW_ROW, PW_WS = PW_BN * PW_ROW;
constexpr int PW_STAGE = PW_XS + PW_WS;                      // 24 KiB
constexpr int PW_NST = 3;                                    // stages of the two-blocks-per-CU form (template NST of the kernel)
constexpr int PW_SROW = PW_BN * 2 + 16;                      // staged output row (bf16) + pad
constexpr int PW_EPI = PW_BM * PW_SROW + 4 * 2 * PW_BN * 4;  // staging + [4 waves][2][256] stats
constexpr int pw_lds(int nst) { return (PW_EPI > nst * PW_STAGE ? PW_EPI : nst * PW_STAGE) + 16; }

// Round 4: the same kernel for SMALL pixel counts (the 4-8 image shards of a data-parallel run), template parameters:
//   BMT = 64 / 96 / 128 pixels per tile: 4 x 65 x 65 pixels are 67 tiles of 256 (every one a K-split + fix-up launch on
//         the 256-tile kernel) but 177 tiles of 96 - one round of whole tiles on 256 CUs, no slabs, no second launch;
//         the LDS image keeps 128 pixel rows per stage (rows >= BMT are out-of-range loads: zeros, no traffic);
//   TAPS: the K loop walks R x S filter taps (taps outer, 32-channel chunks inner) with per-row tap-visibility masks -
//         the 3x3 / dilated layers of a shard; !TAPS = the pointwise form above, no coordinates at all.
//   NST:  LDS-DMA stages.  3 (72 KiB, two blocks per CU) as above; 6 (144 KiB, ONE block per CU) for launches of at most
//         one block per CU: a lone block streams its K tiles at (bytes in flight) / (L2 latency) - two 22-KiB tiles in
//         flight gave ~40 GB/s per block, i.e. 0.55 us per K tile against 0.18 us of MFMA work (3x3 256->256 at 4 images:
//         46 us); five tiles in flight are what the second block's LDS buys when there is no second block.
//   KW:   wave groups of the deep form (NST > 3 only).  2 = eight waves: waves 4-7 mirror waves 0-3 (same couts, same pixels) and
//         each group multiplies ONE of the two 16-channel k-steps of every K tile, so a SIMD holds two waves whose MFMAs, fragment
//         reads and DMA issues interleave (one wave per SIMD is issue-bound: 765 cycles per K tile for 384 cycles of MFMAs); the two
//         partial accumulator sets are added through LDS after the K loop and waves 4-7 leave before the epilogue.
template <int BMT, bool TAPS, int NST, int KW = 1>
__global__ __launch_bounds__(256 * KW, NST > 3 ? 1 : 2) void conv_pw_kernel(ConvP p) {
    static_assert(KW == 1 || (KW == 2 && NST > 3), "two wave groups: deep form only");
    constexpr int KS = 2 / KW;                               // k-steps of a K tile a wave multiplies
    constexpr int NXQ = 2 / KW, NWQ = 4 / KW;                // LDS-DMA instructions per wave and K tile: activations, weights
    typedef bf16_t T;
    typedef Mma<T>::Frag Frag;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    static_assert(BMT == 64 || BMT == 96 || BMT == 128, "pixel tile");
    constexpr int J = BMT / 32;                              // 32-pixel MFMA column blocks per wave
    __shared__ __attribute__((aligned(16))) char smem[pw_lds(NST)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;                              // cout quarter (and the wave index of everything behind the K loop)
    const int kh = KW == 2 ? wave8 >> 2 : 0;                 // which k-step of a K tile this wave multiplies (KW == 2)
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = bid % p.n_tiles, mt = bid / p.n_tiles;    // cout tiles fastest: the 1..8 blocks that share a pixel tile are neighbours
    const int m0 = mt * BMT, n0 = nt * PW_BN;
    EESEG_ACTIVE_EXIT(m0);
    const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 2) & 3;
#ifdef EESEG_PW_STAMPS      // diagnostic build: wall-clock stamps (100 MHz) per block into the conv workspace, nothing reads them
#define PW_STAMP(i) if (tid == 0 && p.slabs) reinterpret_cast<unsigned long long*>(p.slabs)[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime()
#else
#define PW_STAMP(i)
#endif
    PW_STAMP(0);

    f32x16 acc[2][J];
    {
        // DMA roles: one wave instruction = 16 rows x 64 B (lane -> row lane>>2, chunk slot lane&3).  The first two K
        // tiles are requested before anything else happens in the block (their latency is the longest thing in a tile
        // of 4..40 K tiles), the accumulators are produced by the first K tile's MFMAs (C = 0) instead of 128 moves.
        const int lrow = lane >> 2, slot = lane & 3;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.xbytes);
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, p.wbytes);
        uint32_t voffX[NXQ], voffW[NWQ];
        int baseX[NXQ];                                      // TAPS: byte offset of the tap-(0,0) source pixel of the lane's rows (may be negative)
        unsigned vmask[NXQ];                                 // TAPS: taps that land inside the image, per row
        const int taps = TAPS ? p.R * p.S : 1;
#pragma unroll
        for (int q = 0; q < NXQ; ++q) {
            vmask[q] = 0u;
            const int r = (wave8 * NXQ + q) * 16 + lrow;
            const int m = m0 + r;
            const bool on = r < BMT && m < p.M;
            if constexpr (TAPS) {
                baseX[q] = 0;
                if (on) {
                    const int n = m / p.HWout;
                    const int rem = m - n * p.HWout;
                    const int ho = rem / p.Wout, wo = rem - ho * p.Wout;
                    const int hb = ho * p.smul + p.off_h, wb = wo * p.smul + p.off_w;
                    unsigned hm = 0u, wm = 0u;               // separable: tap (r, s) is visible iff r is along h and s along w
                    for (int rr = 0; rr < p.R; ++rr)
                        if ((unsigned)(hb + rr * p.tstep_h) < (unsigned)p.Hin) hm |= 1u << rr;
                    for (int ss = 0; ss < p.S; ++ss)
                        if ((unsigned)(wb + ss * p.tstep_w) < (unsigned)p.Win) wm |= 1u << ss;
                    for (int rr = 0; rr < p.R; ++rr)
                        if ((hm >> rr) & 1u) vmask[q] |= wm << (rr * p.S);
                    baseX[q] = (((n * p.Hin + hb) * p.Win + wb) * p.Cin) * 2 + ((slot ^ ((r >> 2) & 3)) << 4);
                }
                voffX[q] = EESEG_OOB;
            } else {
                voffX[q] = on ? (uint32_t)(m * p.Cin * 2 + ((slot ^ ((r >> 2) & 3)) << 4)) : EESEG_OOB;
            }
        }
#pragma unroll
        for (int q = 0; q < NWQ; ++q) {
            const int r = (wave8 * NWQ + q) * 16 + lrow;
            voffW[q] = (uint32_t)((n0 + r) * taps * p.Cin * 2 + ((slot ^ ((r >> 2) & 3)) << 4));
        }
        const int nkc = p.Cin / 32;                          // K tiles per tap
        const int nk = taps * nkc;
        // block-uniform K iterator of the issue side (K tiles are issued in order 0, 1, 2, ...): tap it_tap, chunk it_ci
        int it_tap = 0, it_ci = 0;
        auto set_tap = [&](int tap) {                        // TAPS: the lane's gather offsets of one tap
            const int r = tap / p.S, s_ = tap - r * p.S;
            const int dtap = ((r * p.tstep_h) * p.Win + s_ * p.tstep_w) * p.Cin * 2;
#pragma unroll
            for (int q = 0; q < NXQ; ++q) voffX[q] = ((vmask[q] >> tap) & 1u) ? (uint32_t)(baseX[q] + dtap) : EESEG_OOB;
        };
        if constexpr (TAPS) set_tap(0);
        auto issue = [&](int kt, int st) {
            const bool live = kt < nk;                       // past the end: out-of-range loads (zeros, no traffic) keep vmcnt uniform
            const int soffX = TAPS ? it_ci * PW_ROW : kt * PW_ROW;
            const int sW = kt * PW_ROW;                      // weights [cout][tap][cin]: K tile kt = tap * nkc + ci starts kt * 64 bytes into a row
            char* sx = smem + st * PW_STAGE;
#pragma unroll
            for (int q = 0; q < NXQ; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(sx + (wave8 * NXQ + q) * 1024), 16,
                                                         (int)(live ? voffX[q] : EESEG_OOB), soffX, 0, 0);
#pragma unroll
            for (int q = 0; q < NWQ; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(sx + PW_XS + (wave8 * NWQ + q) * 1024), 16,
                                                         (int)(live ? voffW[q] : EESEG_OOB), sW, 0, 0);
            if constexpr (TAPS) {                            // advance to K tile kt + 1
                if (++it_ci == nkc) {
                    it_ci = 0;
                    if (++it_tap < taps) set_tap(it_tap);
                }
            }
        };
        if constexpr (NST > 3) {
        // One block per CU: nothing else hides this block's LDS fragment reads, so the loop is software-pipelined - the
        // fragments of K tile t+1 are read (into the other register set) while tile t is multiplied; every stage is in
        // use (tile t's stage is refilled with tile t+NST as soon as all waves hold its fragments in registers).
#pragma unroll
        for (int q = 0; q < NST; ++q) issue(q, q);            // K tiles 0 .. NST-1 (past the end: out-of-range loads)
        PW_STAMP(1);
        auto rdfr = [&](int st, Frag (&a)[2][KS], Frag (&b)[J][KS]) {
            const char* sb = smem + st * PW_STAGE;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* r = sb + PW_XS + (wave * 64 + i * 32 + fr) * PW_ROW;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) a[i][ks] = *reinterpret_cast<const Frag*>(r + ((((ks + kh) * 2 + fh) ^ fsw) << 4));
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const char* r = sb + (j * 32 + fr) * PW_ROW;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) b[j][ks] = *reinterpret_cast<const Frag*>(r + ((((ks + kh) * 2 + fh) ^ fsw) << 4));
            }
        };
        Frag fa0[2][KS], fb0[J][KS], fa1[2][KS], fb1[J][KS];
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NXQ + NWQ) * (NST - 1)) : "memory");       // K tile 0 landed
        BIG_BARRIER();
        rdfr(0, fa0, fb0);
        int s_cur = 0;                                         // stage of K tile t
#ifdef EESEG_PW_DIAG      // diagnostic build: p.tap_inner bit 0 = no MFMAs (after the first tile), bit 1 = no DMA in the loop, bit 2 = no fragment reads
        const int diag = p.tap_inner;
#else
        constexpr int diag = 0;
#endif
        // One K tile: 4J MFMAs with the 6 DMA issues of tile t+NST and the 4 + 2J fragment reads of tile t+1 pinned BETWEEN
        // them (a lone wave per SIMD issues in order: memory instructions ahead of the MFMA block are serial time, between
        // two MFMAs they ride in the matrix pipe's shadow - diagnostic build: DMA issue 10 us, reads 9 us, MFMAs 14 us of a
        // 30-us loop when issued one after the other)
        auto step = [&](int t, const Frag (&ac)[2][KS], const Frag (&bc)[J][KS], Frag (&an)[2][KS], Frag (&bn)[J][KS], bool first) {
            // K tile t+1 landed (tiles t+2 .. t+NST-1 stay in flight) and this wave holds the fragments of tile t
#ifdef EESEG_PW_CYCLES    // diagnostic build: shader-clock stamps of K tiles 16..19, block 0, per wave, behind the per-block stamps
#define PW_CYC(i) if (blockIdx.x == 0 && lane == 0 && p.slabs && t >= 16 && t < 20) \
    reinterpret_cast<unsigned long long*>(p.slabs)[8192 + ((t - 16) * 8 + wave8) * 8 + (i)] = __builtin_readcyclecounter()
#else
#define PW_CYC(i)
#endif
            PW_CYC(0);
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NXQ + NWQ) * (NST - 2)) : "memory");
            PW_CYC(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PW_CYC(2);
            BIG_BARRIER();                                     // ... as does every wave: tile t's stage is free
            PW_CYC(3);
            const int kt = t + NST;
            const bool live = kt < nk;
            const int soffX = TAPS ? it_ci * PW_ROW : kt * PW_ROW, sW = kt * PW_ROW;
            char* sx = smem + s_cur * PW_STAGE;
            const int s_n = s_cur == NST - 1 ? 0 : s_cur + 1;
            const char* sbn = smem + s_n * PW_STAGE;
            constexpr int NM = 2 * KS * J, NA = 2 * KS, NB = J * KS, NO = NA + NB + NXQ + NWQ;
#pragma unroll
            for (int n = 0; n < NM; ++n) {
                const int ks = n / (2 * J), j = (n / 2) % J, i = n & 1;
                if (first || !(diag & 1)) {
                    if (first && ks == 0) {
                        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ac[i][ks], bc[j][ks], z, 0, 0, 0);
                    } else {
                        Mma<T>::run(ac[i][ks], bc[j][ks], acc[i][j]);
                    }
                }
#pragma unroll
                for (int k = 0; k < NO; ++k) {
                    if (k * NM / NO != n) continue;            // memory op k rides behind MFMA number k * NM / NO
                    // the fragment reads first: they must have landed at the top of the next step (an LDS read issued behind
                    // the last MFMA would be exposed latency), the DMA issues last (nobody waits for them for NST-1 steps)
                    if (k < NA) {
                        const int fi = k / KS, fk = k % KS;
                        if (!(diag & 4))
                            an[fi][fk] = *reinterpret_cast<const Frag*>(sbn + PW_XS + (wave * 64 + fi * 32 + fr) * PW_ROW + ((((fk + kh) * 2 + fh) ^ fsw) << 4));
                    } else if (k < NA + NB) {
                        const int fj = (k - NA) / KS, fk = (k - NA) % KS;
                        if (!(diag & 4))
                            bn[fj][fk] = *reinterpret_cast<const Frag*>(sbn + (fj * 32 + fr) * PW_ROW + ((((fk + kh) * 2 + fh) ^ fsw) << 4));
                    } else if (k < NA + NB + NXQ) {
                        const int q = k - NA - NB;
                        if (!(diag & 2))
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(sx + (wave8 * NXQ + q) * 1024), 16,
                                                                     (int)(live ? voffX[q] : EESEG_OOB), soffX, 0, 0);
                    } else {
                        const int q = k - NA - NB - NXQ;
                        if (!(diag & 2))
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(sx + PW_XS + (wave8 * NWQ + q) * 1024), 16,
                                                                     (int)(live ? voffW[q] : EESEG_OOB), sW, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            PW_CYC(4);
            if constexpr (TAPS) {                              // the issue-side iterator: on to K tile kt + 1
                if (++it_ci == nkc) {
                    it_ci = 0;
                    if (++it_tap < taps) set_tap(it_tap);
                }
            }
            s_cur = s_n;
        };
        step(0, fa0, fb0, fa1, fb1, true);
        for (int t = 1; t < nk; t += 2) {
            step(t, fa1, fb1, fa0, fb0, false);
            if (t + 1 < nk) step(t + 1, fa0, fb0, fa1, fb1, false);
        }
        } else {
#pragma unroll
        for (int q = 0; q < NST - 1; ++q) issue(q, q);        // K tiles 0 .. NST-2 (past the end: out-of-range loads)
        PW_STAMP(1);
        auto ktile = [&](int t, int st, int st2, bool first) {
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(6 * (NST - 2)) : "memory");   // K tile t landed (this wave's pieces): NST-2 younger tiles stay in flight
            BIG_BARRIER();                                     // ... everybody's; and stage (t-1) % NST has no reader left
            issue(t + NST - 1, st2);
            const char* sb = smem + st * PW_STAGE;
            Frag a[2][2], b[J][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* r = sb + PW_XS + (wave * 64 + i * 32 + fr) * PW_ROW;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) a[i][ks] = *reinterpret_cast<const Frag*>(r + (((ks * 2 + fh) ^ fsw) << 4));
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const char* r = sb + (j * 32 + fr) * PW_ROW;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) b[j][ks] = *reinterpret_cast<const Frag*>(r + (((ks * 2 + fh) ^ fsw) << 4));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (first && ks == 0) {
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][ks], b[j][ks], z, 0, 0, 0);
                        } else {
                            Mma<T>::run(a[i][ks], b[j][ks], acc[i][j]);
                        }
                    }
            __builtin_amdgcn_s_setprio(0);
        };
        ktile(0, 0, NST - 1, true);
        int st = 1, st2 = 0;
        for (int t = 1; t < nk; ++t) {
            ktile(t, st, st2, false);
            st = st == NST - 1 ? 0 : st + 1;
            st2 = st2 == NST - 1 ? 0 : st2 + 1;
        }
        }
        PW_STAMP(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // trailing out-of-range DMAs still write (zeros) into LDS
        __syncthreads();
        if constexpr (KW == 2) {
            // the two k-halves of every accumulator meet in LDS (the ring is idle now): waves 4-7 write theirs (24 x 16 bytes per lane,
            // [quarter][register group][lane]) and leave - the hardware barrier counts the waves that are still there -, waves 0-3 add
            f32x4* xb = reinterpret_cast<f32x4*>(smem) + (size_t)wave * (2 * J * 4) * 64 + lane;
            if (kh == 1) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < J; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            xb[((i * J + j) * 4 + g) * 64] = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            }
            __syncthreads();
            if (kh == 1) return;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < J; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 o = xb[((i * J + j) * 4 + g) * 64];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] += o[e];
                    }
            __syncthreads();                                   // (four waves from here on) the exchange area becomes the staging area
        }
    }
    PW_STAMP(3);

    // ---- epilogue 1: acc -> (scale, shift) -> LDS staging [pixel][cout] (as the 256-tile kernel) ----
    char* stage = smem;
    float* sRed = reinterpret_cast<float*>(smem + PW_BM * PW_SROW);   // [4 waves][2][256]
    const bool affine = p.scale != nullptr || p.shift != nullptr;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cl = wave * 64 + i * 32 + 8 * g + 4 * fh;
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if (affine) {
                if (p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n0 + cl);
                if (p.shift) sh = *reinterpret_cast<const f32x4*>(p.shift + n0 + cl);
            }
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int px = j * 32 + fr;
                T v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = from_f32<T>(affine ? acc[i][j][4 * g + e] * sc[e] + sh[e] : acc[i][j][4 * g + e]);
                *reinterpret_cast<bf16x4*>(stage + px * PW_SROW + ((cl * 2) ^ (((px >> 3) & 1) << 3))) = bf16x4{v[0], v[1], v[2], v[3]};
            }
        }
    }
    __syncthreads();
    PW_STAMP(4);

    // ---- epilogue 2: row-major read back, residual / ReLU, coalesced 16-byte stores, BN partial sums ----
    constexpr int NIT = BMT / 8;                               // 8 rows per pass
    const int c = tid & 31, r0 = tid >> 5;
    const int cg = n0 + c * 8;
    T* yout = reinterpret_cast<T*>(p.y);
    const T* res = reinterpret_cast<const T*>(p.residual);
    const bool post = res != nullptr || p.relu;
    i32x4 rq[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int row = r0 + 8 * it;
        const i32x4 q = *reinterpret_cast<const i32x4*>(stage + row * PW_SROW + c * 16);
        rq[it] = ((row >> 3) & 1) ? i32x4{q[2], q[3], q[0], q[1]} : q;
    }
    if (post) {
        i32x4 rr[NIT];
        if (res != nullptr) {                                  // all residual loads in flight before the first use
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int m = m0 + r0 + 8 * it;
                rr[it] = m < p.M ? *reinterpret_cast<const i32x4*>(res + (size_t)m * p.ldres + cg) : i32x4{0, 0, 0, 0};
            }
            if (p.resmask) {
                unsigned mb[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int m = m0 + r0 + 8 * it;
                    mb[it] = m < p.M ? p.resmask[(size_t)m * p.ldmask + (cg >> 3)] : 0u;
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) rr[it] = mask_chunk_bf16(rr[it], mb[it]);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            union { i32x4 q; T e[8]; } u, ur;
            u.q = rq[it];
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = to_f32(u.e[e]);
            if (res != nullptr) {
                ur.q = rr[it];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = to_f32(from_f32<T>(f[e] + to_f32(ur.e[e])));
            }
            if (p.relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) u.e[e] = from_f32<T>(f[e]);
            rq[it] = u.q;
        }
    }
    {
        T* yrow = yout + (size_t)(m0 + r0) * p.ldy + cg;
        const size_t ystep = (size_t)8 * p.ldy;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (m0 + r0 + 8 * it < p.M) *reinterpret_cast<i32x4*>(yrow) = rq[it];
            yrow += ystep;
        }
    }
    PW_STAMP(5);
    if (p.stats) {                                             // of the values as stored (bf16)
        f32x2 a1[4], a2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { a1[k] = f32x2{0.f, 0.f}; a2[k] = f32x2{0.f, 0.f}; }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (m0 + r0 + 8 * it < p.M) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned w = (unsigned)rq[it][k];
                    const f32x2 f = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                    a1[k] += f;
                    a2[k] += f * f;
                }
            }
        }
        float s1[8], s2[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            s1[2 * k] = a1[k][0]; s1[2 * k + 1] = a1[k][1];
            s2[2 * k] = a2[k][0]; s2[2 * k + 1] = a2[k][1];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s1[e] += __shfl_xor(s1[e], 32);
            s2[e] += __shfl_xor(s2[e], 32);
        }
        if (lane < 32) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sRed[(wave * 2 + 0) * PW_BN + lane * 8 + e] = s1[e];
                sRed[(wave * 2 + 1) * PW_BN + lane * 8 + e] = s2[e];
            }
        }
        WS_LDS_BARRIER();            // LDS only: the tile's output stores keep draining
        const int col = tid;                                   // 256 threads = 256 couts, both sums; one stat row per 128 pixels
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const float t = sRed[(0 * 2 + which) * PW_BN + col] + sRed[(1 * 2 + which) * PW_BN + col] +
                            sRed[(2 * 2 + which) * PW_BN + col] + sRed[(3 * 2 + which) * PW_BN + col];
            p.stats[((size_t)mt * 2 + which) * p.Cout + n0 + col] = t;
        }
    }
    PW_STAMP(6);
}

// ---------------------------------------------------------------------------------------------
// Weight-stationary pointwise kernel for Cin = 256 (the layer-3 bottleneck convs that EXPAND: conv3 forward and the
// data-gradient of conv1, 47 calls of a ResNet-101 step).  conv_pw_kernel's K loop is bound by bytes in flight: a block
// holds two K tiles (48 KiB) of LDS-DMA in the air and must re-fetch its 128-KiB weight tile from L2 for every pixel tile
// (192 KiB of DMA per 64 KiB of output).  Here a block is PERSISTENT over the pixel tiles of one cout tile and keeps
// its weights where nothing else competes for space: in REGISTERS, as MFMA A fragments (64 couts x 256 channels per
// wave = 128 VGPRs).  Only the activations stream: 64 pixels x 256 channels = one 32-KiB LDS slot per half tile, two
// slots, so the next half tile's loads are in flight during the whole MFMA + epilogue of the current one, and the
// epilogue stages its 64 x 256 output through the slot it has just consumed (no staging area of its own): 72 KiB of
// LDS, two blocks per CU.  Per half tile: 64 MFMAs per wave (1.1 us), epilogue beside the other block's MFMAs.
constexpr int WS_K = 256;
constexpr int WS_HALF = 64;                        // pixels per half tile
constexpr int WS_ROWB = WS_K * 2;                  // 512-byte LDS rows: X [px][256 ch] and the staged output [px][256 cout]
constexpr int WS_SLOT = WS_HALF * WS_ROWB;         // 32 KiB
constexpr int WS_LDS = 2 * WS_SLOT + 4 * 2 * 256 * 4 + 16;

// RES: the call adds a residual (a data-gradient: no BN partial sums) - the residual registers and the statistics registers
// never live in the same instantiation, which is what keeps the weights from spilling
template <bool RES>
__global__ __launch_bounds__(256, 2) void conv_pws_kernel(ConvP p) {
    typedef bf16_t T;
    typedef Mma<T>::Frag Frag;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    __shared__ __attribute__((aligned(16))) char smem[WS_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr_ = lane & 31, fh_ = lane >> 5, fx_ = fr_ & 15;
#ifdef EESEG_PW_STAMPS      // diagnostic build: wall-clock stamps (100 MHz) of the block's second tile into the conv workspace
    int stamp_tile = 0;
#define WS_STAMP(i) if (tid == 0 && p.slabs && (stamp_tile == 1 || (i) == 0 || (i) == 1 || (i) == 15)) \
        reinterpret_cast<unsigned long long*>(p.slabs)[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime()
#else
#define WS_STAMP(i)
#endif
    WS_STAMP(0);
    // block -> (cout tile, tile sequence): the n_tiles blocks that walk the same pixel tiles at the same time have equal
    // blockIdx % 8, i.e. share an XCD and its L2 under round-robin placement (speed only): X comes from HBM once
    const int per = 8 * p.n_tiles;
    const int grp = blockIdx.x / per, g = blockIdx.x % per;
    const int nt = g >> 3, seq = grp * 8 + (g & 7), nseq = (gridDim.x / per) * 8;
    const int n0 = nt * 256;
    const int n_m = p.m_tiles;                                   // 128-pixel tiles (= BN statistic rows)

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.xbytes);
    // DMA role: one wave instruction = 2 rows x 512 B; instr q of wave w covers rows (w*8+q)*2 + (lane>>5)
    const int drow = lane >> 5, dslot = lane & 31;
    auto issue = [&](int mt, int half, bool live) {
        char* sx = smem + half * WS_SLOT;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = (wave * 8 + q) * 2 + drow;
            const int m = mt * 128 + half * WS_HALF + r;
            const uint32_t voff = (live && m < p.M) ? (uint32_t)(m * WS_ROWB + ((dslot ^ (r & 15)) << 4)) : EESEG_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(sx + (wave * 8 + q) * 1024), 16, (int)voff, 0, 0, 0);
        }
    };
    int mt = seq;
    issue(mt, 0, mt < n_m);
    issue(mt, 1, mt < n_m);

    // the weights of this wave's 64 couts, all of K, as A fragments (read once per block)
    Frag a[2][16];
    {
        const T* w = reinterpret_cast<const T*>(p.w);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const T* wr = w + (size_t)(n0 + wave * 64 + i * 32 + fr_) * WS_K + fh_ * 8;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) a[i][ks] = *reinterpret_cast<const Frag*>(wr + ks * 16);
        }
    }
    WS_STAMP(1);
    const T* res = RES ? reinterpret_cast<const T*>(p.residual) : nullptr;
    float* const stats = RES ? nullptr : p.stats;
    T* yout = reinterpret_cast<T*>(p.y);
    float* sRed = reinterpret_cast<float*>(smem + 2 * WS_SLOT);   // [4 waves][2][256]
    const int c_ = tid & 31, r0_ = tid >> 5;                      // epilogue role: 16-byte chunk c of rows r0 + 8*it

    float stat_half0[2] = {0.f, 0.f};
    for (; mt < n_m; mt += nseq) {
        const int nxt = mt + nseq;
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
            char* slot = smem + half * WS_SLOT;
            // lane coordinates made opaque per half tile: every LDS / global address below is then recomputed here instead
            // of being hoisted out of the tile loop, where ~60 loop-invariant addresses would push the weights out of registers
            int fr = fr_, fh = fh_, fx = fx_, c = c_, r0 = r0_;
            asm volatile("" : "+v"(fr), "+v"(fh), "+v"(fx), "+v"(c), "+v"(r0));
            const int cg = n0 + c * 8;
            WS_STAMP(2 + half * 6);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // this half's X landed (the 16 youngest: the other half's loads + stores)
            BIG_BARRIER();
            WS_STAMP(3 + half * 6);
            // two passes of 32 pixels (32 MFMAs each): the accumulators of a pass are rounded to bf16 at once, so 16
            // registers instead of 32 wait for the staging while the weights hold 128 (no scale / shift here: the training
            // forward and the data-gradient have none, and loading 64 per-lane coefficients made the allocator spill the weights)
            bf16x4 packed[2][2][4];                               // [pixel tile j][cout tile i][register group]
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 acc[2];
                const char* r = slot + (j * 32 + fr) * WS_ROWB;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {                  // 2 k-steps at a time: 2 B fragments live
                    Frag b[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) b[u] = *reinterpret_cast<const Frag*>(r + ((((kk * 2 + u) * 2 + fh) ^ fx) << 4));
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            if (kk == 0 && u == 0) {
                                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[0], z, 0, 0, 0);
                            } else {
                                Mma<T>::run(a[i][kk * 2 + u], b[u], acc[i]);
                            }
                        }
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        T v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = from_f32<T>(acc[i][4 * gq + e]);
                        packed[j][i][gq] = bf16x4{v[0], v[1], v[2], v[3]};
                    }
            }
            WS_STAMP(4 + half * 6);
            // LDS-only barriers from here on (s_waitcnt lgkmcnt(0) + s_barrier): a __syncthreads() also waits vmcnt(0), i.e. for the
            // next half tile's LDS-DMA and the output stores in flight - ~2 us of HBM latency per barrier, five times per half tile
            WS_LDS_BARRIER();                                     // every wave has read its fragments: the slot becomes the staging area
            // ---- stage [px][cout] (16-byte chunks XOR-ed with px & 15: the 512-byte pitch maps every row to the same banks) ----
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int cl = wave * 64 + i * 32 + 8 * gq + 4 * fh;
                    // (the 8-byte half of the chunk is flipped for pixels 8-15 of every 16: a 16-lane store group - pixels fr .. fr + 15 of
                    // one accumulator half - then covers all 32 banks instead of the even 8-byte slots of 16 chunks twice; round 4)
                    char* d = slot + fr * WS_ROWB + ((((cl >> 3) ^ fx) << 4) | ((((cl >> 2) ^ (fr >> 3)) & 1) << 3));
#pragma unroll
                    for (int j = 0; j < 2; ++j) *reinterpret_cast<bf16x4*>(d + j * 32 * WS_ROWB) = packed[j][i][gq];
                }
            WS_LDS_BARRIER();
            WS_STAMP(5 + half * 6);
            // ---- two passes of 4 rows x 16 B per thread: read back, residual (+ mask), ReLU, store, BN partial sums.  The next
            //      half tile's DMA is issued after the second read-back (the slot is free then) and after the residual loads
            //      (vmcnt retires in order: a wait for a load issued behind the DMA would wait for the DMA) ----
            const int mh = mt * 128 + half * WS_HALF;
            f32x2 a1[4], a2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { a1[k] = f32x2{0.f, 0.f}; a2[k] = f32x2{0.f, 0.f}; }
            // the residual rows of BOTH passes are requested first: two dependent HBM round trips (one per pass) were half of this
            // phase; the registers are free here (accumulators, B fragments and the packed tiles are dead)
            i32x4 rr[2][4];
            unsigned mb[2][4];
            if (RES) {
#pragma unroll
                for (int ps = 0; ps < 2; ++ps)
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int m = mh + r0 + 8 * (ps * 4 + it);
                        rr[ps][it] = m < p.M ? *reinterpret_cast<const i32x4*>(res + (size_t)m * p.ldres + cg) : i32x4{0, 0, 0, 0};
                        mb[ps][it] = (p.resmask && m < p.M) ? p.resmask[(size_t)m * p.ldmask + (cg >> 3)] : 0xffu;
                    }
            }
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                i32x4 rq[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int row = r0 + 8 * (ps * 4 + it);
                    const i32x4 q = *reinterpret_cast<const i32x4*>(slot + row * WS_ROWB + ((c ^ (row & 15)) << 4));
                    rq[it] = ((row >> 3) & 1) ? i32x4{q[2], q[3], q[0], q[1]} : q;
                }
                if (ps == 1) {
                    WS_LDS_BARRIER();                             // staging consumed by every thread: the slot is free
                    issue(nxt, half, nxt < n_m);                  // next tile's half into it (out-of-range loads past the end)
                }
                if (RES || p.relu) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        union { i32x4 q; T e[8]; } u, ur;
                        u.q = rq[it];
                        float f[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = to_f32(u.e[e]);
                        if (RES) {
                            ur.q = p.resmask ? mask_chunk_bf16(rr[ps][it], mb[ps][it]) : rr[ps][it];
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] = to_f32(from_f32<T>(f[e] + to_f32(ur.e[e])));
                        }
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) u.e[e] = from_f32<T>(f[e]);
                        rq[it] = u.q;
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int m = mh + r0 + 8 * (ps * 4 + it);
                    if (m < p.M) *reinterpret_cast<i32x4*>(yout + (size_t)m * p.ldy + cg) = rq[it];
                }
                if (!RES && stats) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        if (mh + r0 + 8 * (ps * 4 + it) < p.M) {
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const unsigned w = (unsigned)rq[it][k];
                                const f32x2 f = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                                a1[k] += f;
                                a2[k] += f * f;
                            }
                        }
                    }
                }
            }
            WS_STAMP(6 + half * 6);
            if (!RES && stats) {                                  // one stat row per 128-pixel tile: the first half keeps its sums,
                                                                  // the second adds its own and stores (same thread)
                float s1[8], s2[8];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s1[2 * k] = a1[k][0]; s1[2 * k + 1] = a1[k][1];
                    s2[2 * k] = a2[k][0]; s2[2 * k + 1] = a2[k][1];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s1[e] += __shfl_xor(s1[e], 32);
                    s2[e] += __shfl_xor(s2[e], 32);
                }
                if (lane < 32) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        sRed[(wave * 2 + 0) * 256 + lane * 8 + e] = s1[e];
                        sRed[(wave * 2 + 1) * 256 + lane * 8 + e] = s2[e];
                    }
                }
                WS_LDS_BARRIER();
                const int col = tid;
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    const float t = sRed[(0 * 2 + which) * 256 + col] + sRed[(1 * 2 + which) * 256 + col] +
                                    sRed[(2 * 2 + which) * 256 + col] + sRed[(3 * 2 + which) * 256 + col];
                    if (half == 0) stat_half0[which] = t;         // kept in a register: one store per tile, no read-modify-write
                    else stats[((size_t)mt * 2 + which) * p.Cout + n0 + col] = stat_half0[which] + t;
                }
                WS_LDS_BARRIER();                                 // sRed is rewritten by the next half
            }
            WS_STAMP(7 + half * 6);
        }
#ifdef EESEG_PW_STAMPS
        ++stamp_tile;
#endif
    }
    WS_STAMP(15);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // trailing out-of-range DMAs still write into LDS
}


// ---------------------------------------------------------------------------------------------
// Round 4: the same weight-stationary layer with the two halves of the work on DIFFERENT waves.  In conv_pws_kernel every wave
// alternates between 64 MFMAs and the epilogue of what it has just computed (stage, read back, residual, store, statistics),
// five barriers per half tile, and only the second block of the CU fills the gaps: 107 us per call where the MFMAs need 32 us and
// HBM 63 us.  Here a block has 8 waves, one block per CU: waves 0-3 hold the weights and do nothing but LDS-DMA, MFMA and the
// bf16 staging of 32-pixel sub-tiles; waves 4-7 turn the previous sub-tile's staged output into stores (residual, ReLU, BN
// partial sums).  One LDS-only barrier per sub-tile hands a staging slot over (two slots), the activations arrive through a
// four-slot ring issued three sub-tiles ahead (counted vmcnt on the MFMA waves, whose only vector memory traffic it is), and the
// residual rows of the output waves are requested three rounds ahead into a register ring.  Same MFMAs in the same order, same
// order of every sum as conv_pws_kernel: bit-identical outputs and statistics.
constexpr int W2_SUB = 32;                          // pixels per sub-tile
constexpr int W2_SLOT = W2_SUB * WS_ROWB;           // 16 KiB
constexpr int W2_XS = 4, W2_SS = 2;
constexpr int W2_RED = (W2_XS + W2_SS) * W2_SLOT;   // [4 waves][2][256] floats behind the slots
constexpr int W2_LDS = W2_RED + 4 * 2 * 256 * 4 + 16;

template <bool RES, int NMW, int NOW>
__global__ __launch_bounds__((NMW + NOW) * 64, 1) void conv_pws2_kernel(ConvP p) {
    typedef bf16_t T;
    typedef Mma<T>::Frag Frag;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    __shared__ __attribute__((aligned(16))) char smem[W2_RED + NOW * 2 * 256 * 4 + 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = 8 * p.n_tiles;
    const int grp = blockIdx.x / per, g = blockIdx.x % per;
    const int nt = g >> 3, seq = grp * 8 + (g & 7), nseq = (gridDim.x / per) * 8;
    const int n0 = nt * 256;
    const int n_m = p.m_tiles;                                   // 128-pixel tiles (= BN statistic rows)
    // sub-tile t of this block = rows (t & 3) * 32 ... of the 128-pixel tile seq + (t >> 2) * nseq; T4 of them.  Both wave roles
    // run T4 + 1 rounds (the output waves trail by one) and one closing barrier: the same number of s_barrier on every wave.
    const int T4 = seq < n_m ? 4 * ((n_m - seq + nseq - 1) / nseq) : 0;
    char* const sS = smem + W2_XS * W2_SLOT;
    float* const sRed = reinterpret_cast<float*>(smem + W2_RED);

#ifdef EESEG_PW_CYCLES    // diagnostic build: shader-clock time per phase, summed over the rounds, waves 0 and 4 of every 37th block
    long long cyc_t[6] = {0, 0, 0, 0, 0, 0}, cyc_last = 0;
#define W2_CYC(i) { const long long now_ = (long long)__builtin_readcyclecounter(); if ((i) > 0 || cyc_last) cyc_t[(i)] += now_ - cyc_last; cyc_last = now_; }
#define W2_CYC_OUT(role) if (lane == 0 && p.slabs && blockIdx.x % 37 == 0 && (wave == 0 || wave == NMW)) { \
        long long* d_ = reinterpret_cast<long long*>(p.slabs) + 8192 + ((blockIdx.x / 37) * 2 + (role)) * 8; \
        d_[0] = cyc_t[0]; d_[1] = cyc_t[1]; d_[2] = cyc_t[2]; d_[3] = cyc_t[3]; d_[4] = T4; d_[5] = cyc_t[4]; d_[6] = cyc_t[5]; }
#else
#define W2_CYC(i)
#define W2_CYC_OUT(role)
#endif
#ifdef EESEG_W2_WHATIF    // what-if switches of a diagnostic build (EESEG_W2_VAR): each leaves one piece of a round out
    const int var = p.tap_inner;
#else
    constexpr int var = 0;
#endif
    constexpr int NI = 8 / NMW;                                  // 32-cout blocks per MFMA wave (NMW = 4: 64 couts, 8: 32 couts)
    constexpr int ND = 16 / NMW;                                 // LDS-DMA instructions per MFMA wave and sub-tile
    if (wave < NMW) {
        // ------------------------------- MFMA waves -------------------------------
        const int fr_ = lane & 31, fh_ = lane >> 5, fx_ = fr_ & 15;
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.xbytes);
        // one wave instruction = 2 rows x 512 B; instruction q of wave w covers rows (w*ND+q)*2 + (lane>>5) of the sub-tile
        auto issue = [&](int t) {
            int drow = lane >> 5, dslot = lane & 31;
            asm volatile("" : "+v"(drow), "+v"(dslot));          // recomputed per call: no hoisted addresses beside the weights
            char* sx = smem + (t & 3) * W2_SLOT;
            const int mb = (seq + (t >> 2) * nseq) * 128 + (t & 3) * W2_SUB;
#pragma unroll
            for (int q = 0; q < ND; ++q) {
                const int r = (wave * ND + q) * 2 + drow;
                const int m = mb + r;
                const uint32_t voff = (t < T4 && m < p.M) ? (uint32_t)(m * WS_ROWB + ((dslot ^ (r & 15)) << 4)) : EESEG_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(sx + (wave * ND + q) * 1024), 16, (int)voff, 0, 0, 0);
            }
        };
        Frag a[NI][16];                                           // this wave's couts, all of K, as A fragments
        {
            const T* w = reinterpret_cast<const T*>(p.w);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const T* wr = w + (size_t)(n0 + wave * (NI * 32) + i * 32 + fr_) * WS_K + fh_ * 8;
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) a[i][ks] = *reinterpret_cast<const Frag*>(wr + ks * 16);
            }
        }
        issue(0);
        issue(1);
        issue(2);
#pragma unroll 1
        for (int t = 0; t <= T4; ++t) {
            W2_CYC(0);
            if (NMW == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // sub-tile t landed (t+1, t+2 in flight); the weights are older still
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            W2_CYC(1);
            WS_LDS_BARRIER();                                     // ... for every wave; staging slot t & 1 has been read back
            W2_CYC(2);
            if (!(var & 32)) issue(t + 3);                        // into the slot round t - 1 consumed (zero-fill past the end)
            W2_CYC(3);
            if (t < T4) {
                int fr = fr_, fh = fh_, fx = fx_;
                asm volatile("" : "+v"(fr), "+v"(fh), "+v"(fx));
                const char* r = smem + (t & 3) * W2_SLOT + fr * WS_ROWB;
                f32x16 acc[NI];
                // 2 k-steps at a time, the B fragments of the NEXT pair requested before the MFMAs of this one: the wave is alone
                // (or one of two) on its SIMD's matrix pipe, a read -> wait -> MFMA chain would idle the pipe for the LDS latency
                // eight times per sub-tile
                Frag b[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u) b[0][u] = *reinterpret_cast<const Frag*>(r + (((u * 2 + fh) ^ fx) << 4));
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    if (kk < 7) {
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            b[(kk + 1) & 1][u] = *reinterpret_cast<const Frag*>(r + (((((kk + 1) * 2 + u) * 2 + fh) ^ fx) << 4));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (var & 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int i = 0; i < NI; ++i) {
                            if (kk == 0 && u == 0) {
                                const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[0][0], z, 0, 0, 0);
                            } else {
                                Mma<T>::run(a[i][kk * 2 + u], b[kk & 1][u], acc[i]);
                            }
                        }
                    if (var & 1) __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                W2_CYC(4);
                // stage [px][cout] as bf16 (16-byte chunks XOR-ed with px & 15, as conv_pws_kernel)
                char* slot = sS + (t & 1) * W2_SLOT;
                if (var & 8) { if (acc[0][0] == 1.2345f && acc[NI - 1][3] == 7.f) slot[lane] = 1; }
                else
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        T v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = from_f32<T>(acc[i][4 * gq + e]);
                        const int cl = wave * (NI * 32) + i * 32 + 8 * gq + 4 * fh;
                        *reinterpret_cast<bf16x4*>(slot + fr * WS_ROWB + ((((cl >> 3) ^ fx) << 4) | ((cl & 4) << 1))) = bf16x4{v[0], v[1], v[2], v[3]};
                    }
#ifdef EESEG_PW_CYCLES
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                W2_CYC(5);
#endif
            }
        }
        W2_CYC(0);
        WS_LDS_BARRIER();                                         // the closing round of the output waves
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // trailing out-of-range DMAs still write into LDS
        W2_CYC_OUT(0);
    } else {
        // ------------------------------- output waves -------------------------------
        const int ot = tid - NMW * 64, ow = wave - NMW;
        const int c = ot & 31, r0 = ot >> 5;                      // 16-byte chunk c of rows r0 + RS*it of a sub-tile
        constexpr int RS = NOW * 2, NR = 32 / RS;                 // row stride and rows per thread (NOW = 4: 8 and 4, 8: 16 and 2)
        const int cg = n0 + c * 8;
        const T* res = RES ? reinterpret_cast<const T*>(p.residual) : nullptr;
        float* const stats = RES ? nullptr : p.stats;
        T* yout = reinterpret_cast<T*>(p.y);
        constexpr int RR = RES ? 4 : 1;
        i32x4 rr[RR][NR];                                          // residual rows, requested three rounds before their use
        unsigned mb[RR][NR];
        f32x2 a1[4], a2[4];
        float stat_half0[2] = {0.f, 0.f};
        auto row0 = [&](int t) { return (seq + (t >> 2) * nseq) * 128 + (t & 3) * W2_SUB; };
        auto request = [&](auto kc, int t) {                      // residual (+ mask bytes) of sub-tile t into ring slot K
            constexpr int K = decltype(kc)::value;
            if (RES) {
                const int mh = row0(t);
#pragma unroll
                for (int it = 0; it < NR; ++it) {
                    const int m = mh + r0 + RS * it;
                    const bool ok = t < T4 && m < p.M;
                    rr[K % RR][it] = ok ? *reinterpret_cast<const i32x4*>(res + (size_t)m * p.ldres + cg) : i32x4{0, 0, 0, 0};
                    mb[K % RR][it] = (p.resmask && ok) ? p.resmask[(size_t)m * p.ldmask + (cg >> 3)] : 0xffu;
                }
            }
        };
        // the statistics of a half tile (two sub-tiles) cross the four output waves through sRed one round AFTER they were written
        auto stats_collect = [&](int t_done) {                    // t_done: the sub-tile whose sums sRed holds (odd)
            const int col = ot;
            if (NOW > 4 && col >= 256) return;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                float s = sRed[(0 * 2 + which) * 256 + col] + sRed[(1 * 2 + which) * 256 + col] +
                          sRed[(2 * 2 + which) * 256 + col] + sRed[(3 * 2 + which) * 256 + col];
                if (NOW > 4) s += sRed[(4 * 2 + which) * 256 + col] + sRed[(5 * 2 + which) * 256 + col] +
                                  sRed[(6 * 2 + which) * 256 + col] + sRed[(7 * 2 + which) * 256 + col];
                if ((t_done & 2) == 0) stat_half0[which] = s;     // first half: kept in a register, one store per tile
                else stats[((size_t)(seq + (t_done >> 2) * nseq) * 2 + which) * p.Cout + n0 + col] = stat_half0[which] + s;
            }
        };
        auto round = [&](auto kc, int t) {                        // round t: emits sub-tile t - 1 (ring slot K = (t - 1) & 3)
            constexpr int K = decltype(kc)::value;
            W2_CYC(0);
            WS_LDS_BARRIER();
            W2_CYC(1);
            request(std::integral_constant<int, (K + 3) % 4>{}, t + 2);
            if (t < 1) return;
            const int u = t - 1;
            if (!RES && stats && u >= 2 && (u & 1) == 0) stats_collect(u - 1);
            const char* slot = sS + (u & 1) * W2_SLOT;
            const int mh = row0(u);
            i32x4 rq[NR];
            if (var & 64) return;
#pragma unroll
            for (int it = 0; it < NR; ++it) {
                const int row = r0 + RS * it;
                rq[it] = *reinterpret_cast<const i32x4*>(slot + row * WS_ROWB + ((c ^ (row & 15)) << 4));
            }
            if ((u & 1) == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { a1[k] = f32x2{0.f, 0.f}; a2[k] = f32x2{0.f, 0.f}; }
            }
#ifdef EESEG_PW_CYCLES
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            W2_CYC(2);
#endif
            if (RES || p.relu) {
#pragma unroll
                for (int it = 0; it < NR; ++it) {
                    union { i32x4 q; T e[8]; } v, vr;
                    v.q = rq[it];
                    float f[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = to_f32(v.e[e]);
                    if (RES) {
                        vr.q = p.resmask ? mask_chunk_bf16(rr[K % RR][it], mb[K % RR][it]) : rr[K % RR][it];
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = to_f32(from_f32<T>(f[e] + to_f32(vr.e[e])));
                    }
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v.e[e] = from_f32<T>(f[e]);
                    rq[it] = v.q;
                }
            }
#pragma unroll
            for (int it = 0; it < NR; ++it) {
                const int m = mh + r0 + RS * it;
                if (m < p.M && (!(var & 4) || rq[it][0] == 0x12345)) *reinterpret_cast<i32x4*>(yout + (size_t)m * p.ldy + cg) = rq[it];
            }
            W2_CYC(3);
            if (!RES && stats && !(var & 2)) {
#pragma unroll
                for (int it = 0; it < NR; ++it) {
                    if (mh + r0 + RS * it < p.M) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const unsigned w = (unsigned)rq[it][k];
                            const f32x2 f = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                            a1[k] += f;
                            a2[k] += f * f;
                        }
                    }
                }
                if (u & 1) {                                      // a half tile is complete: its sums go to sRed (read next round)
                    float s1[8], s2[8];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        s1[2 * k] = a1[k][0]; s1[2 * k + 1] = a1[k][1];
                        s2[2 * k] = a2[k][0]; s2[2 * k + 1] = a2[k][1];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s1[e] += __shfl_xor(s1[e], 32);
                        s2[e] += __shfl_xor(s2[e], 32);
                    }
                    if (lane < 32) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            sRed[(ow * 2 + 0) * 256 + lane * 8 + e] = s1[e];
                            sRed[(ow * 2 + 1) * 256 + lane * 8 + e] = s2[e];
                        }
                    }
                }
            }
        };
        request(std::integral_constant<int, 0>{}, 0);
        request(std::integral_constant<int, 1>{}, 1);
        // round t uses ring slot (t - 1) & 3 and requests sub-tile t + 2 into slot (t + 2) & 3: four rounds per trip, T4 is a
        // multiple of 4, so rounds 0 .. T4 are T4 / 4 whole trips and the single round T4 (slot 3)
#pragma unroll 1
        for (int t = 0; t < T4; t += 4) {
            round(std::integral_constant<int, 3>{}, t);
            round(std::integral_constant<int, 0>{}, t + 1);
            round(std::integral_constant<int, 1>{}, t + 2);
            round(std::integral_constant<int, 2>{}, t + 3);
        }
        round(std::integral_constant<int, 3>{}, T4);
        W2_CYC(0);
        WS_LDS_BARRIER();
        if (!RES && stats && T4 > 0) stats_collect(T4 - 1);
        W2_CYC_OUT(1);
    }
}


// ---------------------------------------------------------------------------------------------
// Round 4, what the wave-specialised forms measured (scripts/pws_bench.py, pws_pattern.py, pws2_cycles.py): four MFMA + four
// output waves, eight + four and eight + eight all take the time of conv_pws_kernel, 94-96 us for 135,200 x 256 -> 1024
// without statistics - while the SAME kernel on a 256 -> 256 layer with a residual (one cout tile: every block reads its
// pixels once and writes whole rows) moves 5.1 TB/s.  What the four cout-tile blocks of a pixel tile have in common is the
// activation tile: each of them pulls it through its own CU's vector-memory path (L2 hits, but 277 MB of LDS-DMA per call
// beside 277 MB of stores), and the CUs together move ~5.5 TB/s whatever the source.  So: fewer passes of X.  A block here
// holds the weights of 512 couts - eight waves x 64 couts x 256 channels = 128 VGPRs each, 256 KiB of registers, half the
// CU's file - and X makes two passes instead of four.  Every wave does everything (LDS-DMA, MFMA, staging, read-back,
// stores, statistics), in phases separated by two LDS-only barriers per 32-pixel sub-tile; scheduling unit and BN statistic
// row = 64 pixels (two sub-tiles), so 128 sequences share the 2,113 units of a 32-image batch 17 : 16.
constexpr int P3_SUB = 32;
constexpr int P3_XSLOT = P3_SUB * WS_ROWB;          // 16 KiB: [32 px][256 ch]
constexpr int P3_XS = 4;
constexpr int P3_SROW = 512 * 2;                    // staged output rows: 512 couts
constexpr int P3_STAGE = P3_SUB * P3_SROW;          // 32 KiB
constexpr int P3_RED = P3_XS * P3_XSLOT + P3_STAGE; // [8 waves][2][512] floats
constexpr int P3_LDS = P3_RED + 8 * 2 * 512 * 4 + 16;

template <bool RES>
__global__ __launch_bounds__(512, 1) void conv_pws3_kernel(ConvP p) {
    typedef bf16_t T;
    typedef Mma<T>::Frag Frag;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    __shared__ __attribute__((aligned(16))) char smem[P3_LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr_ = lane & 31, fh_ = lane >> 5, fx_ = fr_ & 15;
#ifdef EESEG_W2_WHATIF    // what-if switches of a diagnostic build (EESEG_W2_VAR): each leaves one piece of a round out
    const int var = p.tap_inner;
#else
    constexpr int var = 0;
#endif
    // block -> (512-cout tile, unit sequence): the blocks that walk the same pixels at the same time share an XCD (blockIdx % 8)
    const int per = 8 * p.n_tiles;                               // n_tiles = Cout / 512 here
    const int grp = blockIdx.x / per, g = blockIdx.x % per;
    const int nt = g >> 3, seq = grp * 8 + (g & 7), nseq = (gridDim.x / per) * 8;
    const int n0 = nt * 512;
    const int n_u = p.m_tiles;                                   // 64-pixel units (= BN statistic rows of this kernel)
    const int T2 = seq < n_u ? 2 * ((n_u - seq + nseq - 1) / nseq) : 0;   // sub-tiles of this block
    auto row0 = [&](int t) { return (seq + (t >> 1) * nseq) * 64 + (t & 1) * P3_SUB; };
    char* const sS = smem + P3_XS * P3_XSLOT;
    float* const sRed = reinterpret_cast<float*>(smem + P3_RED);

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, p.xbytes);
    const uint32_t ybytes = (uint32_t)(((size_t)(p.M - 1) * p.ldy + p.Cout) * 2);
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(p.y, ybytes);
    const __amdgpu_buffer_rsrc_t rres = make_rsrc(RES ? p.residual : p.y, RES ? (uint32_t)(((size_t)(p.M - 1) * p.ldres + p.Cout) * 2) : ybytes);
    // LDS-DMA: one wave instruction = 2 rows x 512 B; instruction q of wave w covers rows (w*2+q)*2 + (lane>>5) of the sub-tile
    auto issue = [&](int t) {
        int drow = lane >> 5, dslot = lane & 31;
        asm volatile("" : "+v"(drow), "+v"(dslot));              // recomputed per call: no hoisted addresses beside the weights
        char* sx = smem + (t & 3) * P3_XSLOT;
        const int mb = row0(t);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r = (wave * 2 + q) * 2 + drow;
            const int m = mb + r;
            const uint32_t voff = (t < T2 && m < p.M && !(var & 32)) ? (uint32_t)(m * WS_ROWB + ((dslot ^ (r & 15)) << 4)) : EESEG_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(sx + (wave * 2 + q) * 1024), 16, (int)voff, 0, 0, 0);
        }
    };
    Frag a[2][16];                                                // this wave's 64 couts, all of K, as A fragments
    {
        const T* w = reinterpret_cast<const T*>(p.w);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const T* wr = w + (size_t)(n0 + wave * 64 + i * 32 + fr_) * WS_K + fh_ * 8;
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) a[i][ks] = *reinterpret_cast<const Frag*>(wr + ks * 16);
        }
    }
    // the weights are awaited HERE, by waits the compiler places itself (an empty asm that reads every fragment): its wait-count pass
    // does not read the counted waits in the inline assembly below and would otherwise keep waiting for "possibly outstanding" weights
    // inside every round's MFMAs - each such wait also drains the LDS-DMA just issued
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) asm volatile("" :: "v"(a[i][ks]));
    asm volatile("" ::: "memory");
    issue(0);
    issue(1);
    issue(2);
    float* const stats = RES ? nullptr : p.stats;
    f32x2 a1[4], a2[4];
    // the sums of a finished unit cross the eight waves through sRed and are stored one round later (behind that round's barrier)
    auto stats_collect = [&](int t_done) {                       // t_done: last sub-tile of the unit (odd)
        const int col = tid;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float s = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) s += sRed[(w8 * 2 + which) * 512 + col];
            stats[((size_t)(seq + (t_done >> 1) * nseq) * 2 + which) * p.Cout + n0 + col] = s;
        }
    };
#pragma unroll 1
    for (int t = 0; t < T2; ++t) {
        // vector-memory operations retire in order; behind the LDS-DMA of sub-tile t (issued in round t - 3) this wave has issued at
        // least: the stores of round t - 3 (4), and the residual requests (4 + 4 mask bytes), LDS-DMA (2) and stores (4) of rounds
        // t - 2 and t - 1 - every one of them unconditionally (rows past the end go out of range, not away).  The statistics stores
        // of every other round are not counted: waiting for a few operations more than needed is safe, for fewer is not.
        // (The residual is requested BEFORE the round's LDS-DMA: its data cannot return before everything older has, and behind the
        // DMA it would pull sub-tile t + 3 in within this round.)
        // (the first three rounds have fewer behind theirs: they wait for everything)
        if (t < 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (!RES) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (p.resmask) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        WS_LDS_BARRIER();                                         // X(t) is there for every wave; the staging area and sRed have been read
        if (!RES && stats && t >= 2 && (t & 1) == 0) stats_collect(t - 1);
        int fr = fr_, fh = fh_, fx = fx_, c = tid & 63, r0 = wave;
        asm volatile("" : "+v"(fr), "+v"(fh), "+v"(fx), "+v"(c));
        const int mh = row0(t);
        const int cg = n0 + c * 8;
        i32x4 rr[4];
        unsigned mb[4];
        if (RES) {                                                // requested now, used behind the MFMAs
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int m = mh + r0 + 8 * it;
                const uint32_t off = m < p.M ? (uint32_t)(((size_t)m * p.ldres + cg) * 2) : EESEG_OOB;
                rr[it] = __builtin_amdgcn_raw_buffer_load_b128(rres, (int)off, 0, 0);
                const int mc = m < p.M ? m : p.M - 1;
                mb[it] = p.resmask ? p.resmask[(size_t)mc * p.ldmask + (cg >> 3)] : 0xffu;
            }
            asm volatile("" ::: "memory");
        }
        issue(t + 3);                                             // into the slot round t - 1 consumed (zero-fill past the end)
        asm volatile("" ::: "memory");
        {
            const char* r = smem + (t & 3) * P3_XSLOT + fr * WS_ROWB;
            f32x16 acc[2];
            Frag b[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u) b[0][u] = *reinterpret_cast<const Frag*>(r + (((u * 2 + fh) ^ fx) << 4));
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {                      // 2 k-steps at a time, the next pair's fragments requested first
                if (kk < 7) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        b[(kk + 1) & 1][u] = *reinterpret_cast<const Frag*>(r + (((((kk + 1) * 2 + u) * 2 + fh) ^ fx) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        if (kk == 0 && u == 0) {
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[0][0], z, 0, 0, 0);
                        } else {
                            Mma<T>::run(a[i][kk * 2 + u], b[kk & 1][u], acc[i]);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // stage [px][512 couts] as bf16 (16-byte chunks XOR-ed with px & 15: the 1-KiB pitch maps every row to the same banks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    T v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = from_f32<T>(acc[i][4 * gq + e]);
                    const int cl = wave * 64 + i * 32 + 8 * gq + 4 * fh;
                    *reinterpret_cast<bf16x4*>(sS + fr * P3_SROW + ((((cl >> 3) ^ fx) << 4) | ((cl & 4) << 1))) = bf16x4{v[0], v[1], v[2], v[3]};
                }
        }
        WS_LDS_BARRIER();
        // ---- read back 4 rows x 16 B per thread (thread = chunk c of rows wave + 8*it), residual, ReLU, store, BN partial sums ----
        i32x4 rq[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = r0 + 8 * it;
            rq[it] = *reinterpret_cast<const i32x4*>(sS + row * P3_SROW + ((c ^ (row & 15)) << 4));
        }
        if ((t & 1) == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { a1[k] = f32x2{0.f, 0.f}; a2[k] = f32x2{0.f, 0.f}; }
        }
        if (RES || p.relu) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                union { i32x4 q; T e[8]; } v, vr;
                v.q = rq[it];
                float f[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = to_f32(v.e[e]);
                if (RES) {
                    vr.q = p.resmask ? mask_chunk_bf16(rr[it], mb[it]) : rr[it];
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = to_f32(from_f32<T>(f[e] + to_f32(vr.e[e])));
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v.e[e] = from_f32<T>(f[e]);
                rq[it] = v.q;
            }
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int m = mh + r0 + 8 * it;
            uint32_t off = m < p.M ? (uint32_t)(((size_t)m * p.ldy + cg) * 2) : EESEG_OOB;
            if (var & 128) off = m < p.M ? (uint32_t)((((size_t)nt * p.M + m) * 512 + c * 8) * 2) : EESEG_OOB;   // every block a stream of its own
            if (var & 4) off = EESEG_OOB;
            __builtin_amdgcn_raw_buffer_store_b128(rq[it], ry, (int)off, 0, 0);
        }
        if (!RES && stats && !(var & 2)) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                if (mh + r0 + 8 * it < p.M) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const unsigned w = (unsigned)rq[it][k];
                        const f32x2 f = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
                        a1[k] += f;
                        a2[k] += f * f;
                    }
                }
            }
            if (t & 1) {                                          // the unit is complete: this wave's sums of its 8 rows go to sRed
                f32x4* d1 = reinterpret_cast<f32x4*>(sRed + (wave * 2 + 0) * 512 + c * 8);
                f32x4* d2 = reinterpret_cast<f32x4*>(sRed + (wave * 2 + 1) * 512 + c * 8);
                d1[0] = f32x4{a1[0][0], a1[0][1], a1[1][0], a1[1][1]};
                d1[1] = f32x4{a1[2][0], a1[2][1], a1[3][0], a1[3][1]};
                d2[0] = f32x4{a2[0][0], a2[0][1], a2[1][0], a2[1][1]};
                d2[1] = f32x4{a2[2][0], a2[2][1], a2[3][0], a2[3][1]};
            }
        }
    }
    WS_LDS_BARRIER();
    if (!RES && stats && T2 > 0) stats_collect(T2 - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // trailing out-of-range DMAs still write into LDS
}

int g_conv_pw_all = 0;        // EESEG_OPT_CONV_PW_ALL: 1 = every eligible pointwise layer on conv_pw_kernel, not only the output-heavy ones
int g_conv_pws = 1;           // EESEG_OPT_CONV_PWS: Cin = 256 expanding pointwise layers on the weight-stationary kernel

int g_conv_big_cus = 256;        // EESEG_OPT_CONV_CUS: CUs a launch may count on (< 256 while RCCL kernels hold some)
int g_last_conv_kernel = 0;    // eeseg_last_kernel(0): which kernel the last eeseg_conv_igemm call launched (EESEG_KERNEL_*)
int g_last_stats_rows = 0;     // eeseg_last_kernel(2): rows of `stats` that call wrote (one per pixel tile of the kernel it chose)
int g_conv_sm = 2;             // EESEG_OPT_CONV_SMALL_M (2 = 1 + the tile choice by rounds for pointwise layers of a few rounds): layers with <= CUs/2 tiles of 256 x 256 (per-GPU shards) run on the 64/96/128-pixel
                               // tile kernel (one round of whole tiles) instead of K-split tiles + fix-up on the 256-tile kernel
int g_conv_sm_deep = 1;        // EESEG_OPT_CONV_SMALL_M_DEEP: launches of <= one block per CU use the six-stage ring
int g_conv_sm_max_nk = 160;    // EESEG_OPT_CONV_SMALL_M_MAX_K: ... when the K loop has at most this many 32-channel tiles (taps x Cin / 32)

// pixel tile of conv_pw_kernel for a small layer: the tile whose busiest CU has the least work (rounds of blocks x (pixels + a
// per-block fixed cost worth ~40 pixels)).  A launch of <= one block per CU runs the six-stage ring alone on its CU; anything bigger
// runs two blocks per CU (72 KiB of LDS each), each of them ~1.5x slower than alone.  wide (EESEG_OPT_CONV_SMALL_M = 2, default): the
// same choice for pointwise layers of a few rounds that are NOT small by the 256-tile count - 4 x 65 x 65 x (256 -> 1024) is 532
// blocks of 128 pixels on 512 slots, i.e. two rounds for 1.04 rounds of work; 708 blocks of 96 pixels are two rounds of a smaller tile
int pick_small_bm(long long M, int n_tiles, int cus, bool wide = false) {
    int best = PW_BM;
    double best_cost = -1.0;
    const int cand[3] = {64, 96, 128};
    for (int k = 0; k < 3; ++k) {
        const int bm = cand[k];
        const long long blocks = ((M + bm - 1) / bm) * n_tiles;
        double cost;
        if (wide) {
            const long long slots = blocks <= cus ? cus : 2ll * cus;
            cost = (double)((blocks + slots - 1) / slots) * (bm + 40) * (blocks <= cus ? 1.0 : 1.5);
        } else {
            if (blocks > 2ll * cus && bm != PW_BM) continue;
            cost = (double)(((blocks + cus - 1) / cus) * (bm + 40));
        }
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = bm; }
    }
    return best;
}

int launch_pw(ConvP& p, long long M, int bm, bool taps, hipStream_t st) {
    p.m_tiles = (int)((M + bm - 1) / bm);
    const dim3 grid((unsigned)(p.m_tiles * p.n_tiles));
    // at most one block per CU: the deep ring (six stages, 144 KiB) instead of room for a second block
    const bool deep = g_conv_sm_deep && (long long)p.m_tiles * p.n_tiles <= g_conv_big_cus;
#define EESEG_LAUNCH_PW(BM_) { \
        if (taps && deep && g_conv_sm_deep == 2 && BM_ <= 96) hipLaunchKernelGGL((conv_pw_kernel<BM_ <= 96 ? BM_ : 96, true, 6, 2>), grid, dim3(512), 0, st, p); \
        else if (deep && g_conv_sm_deep == 2 && BM_ <= 96) hipLaunchKernelGGL((conv_pw_kernel<BM_ <= 96 ? BM_ : 96, false, 6, 2>), grid, dim3(512), 0, st, p); \
        else if (taps && deep) hipLaunchKernelGGL((conv_pw_kernel<BM_, true, 6>), grid, dim3(256), 0, st, p); \
        else if (taps) hipLaunchKernelGGL((conv_pw_kernel<BM_, true, 3>), grid, dim3(256), 0, st, p); \
        else if (deep) hipLaunchKernelGGL((conv_pw_kernel<BM_, false, 6>), grid, dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((conv_pw_kernel<BM_, false, 3>), grid, dim3(256), 0, st, p); }
    if (bm == 64) EESEG_LAUNCH_PW(64)
    else if (bm == 96) EESEG_LAUNCH_PW(96)
    else EESEG_LAUNCH_PW(128)
#undef EESEG_LAUNCH_PW
    g_last_conv_kernel = EESEG_KERNEL_CONV_PW;
    g_last_stats_rows = p.m_tiles;
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}
int g_conv_pw_max_k = 1280;   // EESEG_OPT_CONV_PW_MAX_K: pointwise bf16 layers with Cin <= this use conv_pw_kernel (0 = never)

int g_conv_big_split_min_k = 4;  // EESEG_OPT_CONV_SPLIT_MIN_K: K tiles a K range of a split tail tile holds at least
int g_conv_big_tail_min = 224;   // a last round with at least this many tiles is left unsplit
int g_conv_big_merge = 1;        // K-split tail and full rounds in one launch (EESEG_OPT_CONV_TAIL_MERGE)
int g_conv_big_m16 = 1;          // EESEG_OPT_CONV_MFMA16: 1 (default) = the 256-tile kernel computes with v_mfma_f32_16x16x32_bf16, 0 = 32x32x16 (measured at 32 x 65 x 65: 3-7 % faster on every MFMA-bound layer, scripts/m16_bench.py)
int g_conv_big_swp = 1;          // EESEG_OPT_CONV_SWP: software-pipelined K loop of the 16x16x32 form (default; same bits as the two-group loop, step 102.9 -> 101.8 ms)
int g_conv_big_cg = 0;           // EESEG_OPT_CONV_COUT_GROUP: cout tiles an XCD works on at a time when a layer has more (0 = all; measured neutral: the merged ASPP data-gradient 3.18-3.27 ms for 0/1/2/4/8)

// launch plan for the 256x256 kernel: full rounds of one tile per available CU, then the remainder split along K
int launch_big(ConvP& p, long long M, hipStream_t st, void* workspace, long long workspace_bytes) {
    const int mtiles = (int)((M + BIGT - 1) / BIGT);
    const int tiles = mtiles * p.n_tiles;
    const int cus = g_conv_big_cus;
    const int rounds = tiles / cus, rem = tiles % cus;
    p.cg = 1; p.cg_limit = 0;
    if (g_conv_big_cg > 0 && p.n_tiles > g_conv_big_cg && p.n_tiles % g_conv_big_cg == 0) {
        const int pgs = 32 / g_conv_big_cg;                       // pixel tiles per group
        p.cg = g_conv_big_cg;
        p.cg_limit = mtiles / pgs * pgs * p.n_tiles;              // the ragged last pixel tiles keep the plain order
    }
    const int nk_max = (p.n_gtaps ? p.n_gtaps : p.R * p.S) * (p.Cin / 64);
    int ksplit = 1;
    if (rem > 0 && rem < g_conv_big_tail_min * cus / 256) {
        ksplit = cus / rem;
        if (ksplit > nk_max / g_conv_big_split_min_k) ksplit = nk_max / g_conv_big_split_min_k;   // at least 4 K tiles per range (pipeline fill)
        const long long fit = workspace ? workspace_bytes / ((long long)rem * SLAB_FLOATS * 4) : 0;
        if (ksplit > fit) ksplit = (int)fit;
    }
    const int form = g_conv_big_m16 ? (g_conv_big_swp ? 2 : 1) : 0;    // 0: 32x32x16, 1: 16x16x32, 2: 16x16x32 software-pipelined
#define EESEG_LAUNCH_BIG(MODE_, grid_, args_) { \
        if (form == 2) hipLaunchKernelGGL((conv_big_kernel<MODE_, true, true>), grid_, dim3(512), 0, st, args_); \
        else if (form == 1) hipLaunchKernelGGL((conv_big_kernel<MODE_, true, false>), grid_, dim3(512), 0, st, args_); \
        else hipLaunchKernelGGL((conv_big_kernel<MODE_, false, false>), grid_, dim3(512), 0, st, args_); }
    if (ksplit < 2) {
        p.tile_begin = 0; p.ksplit = 1; p.slabs = nullptr;
        EESEG_LAUNCH_BIG(0, dim3(tiles), p)
        EESEG_LAUNCH_CHECK();
        return EESEG_OK;
    }
    p.slabs = reinterpret_cast<float*>(workspace);
    p.ksplit = ksplit;
    p.tile_begin = rounds * cus;
    p.n_split_blocks = rem * ksplit;
    if (rounds > 0 && g_conv_big_merge) {      // one launch: K-range blocks first, whole tiles behind them
        EESEG_LAUNCH_BIG(2, dim3(rem * ksplit + rounds * cus), p)
    } else {
        if (rounds > 0) {
            ConvP q = p;
            q.tile_begin = 0;
            EESEG_LAUNCH_BIG(0, dim3(rounds * cus), q)
        }
        EESEG_LAUNCH_BIG(1, dim3(rem * ksplit), p)
    }
#undef EESEG_LAUNCH_BIG
    if (form) hipLaunchKernelGGL(conv_big_fixup_kernel<true>, dim3(rem * 8), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv_big_fixup_kernel<false>, dim3(rem * 8), dim3(256), 0, st, p);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

int g_conv_auto_narrow = 0;   // measured neutral at B=4 (that regime is bound by per-kernel fixed costs), off by default
int g_conv_narrow_max = 64;   // layers with Cout <= this use the 128x64 tile (EESEG_OPT_CONV_NARROW_MAX)
int g_conv_linear = 0;   // eeseg_set_option(EESEG_OPT_CONV_TAP_INNER, 0|1)
int g_conv_pipe = 3;     // eeseg_set_option(EESEG_OPT_CONV_PIPE, 0..3): 3 = 256x256 kernel where eligible, LDS-DMA 128x128 elsewhere
                         // (A/B on MI355X, R50 layer set at B=16: conv fwd+dgrad 21.5 -> 17.1 ms per step)

template <typename T, int BN>
int launch(const ConvP& p, hipStream_t st) {
    const int grid = p.m_tiles * p.n_tiles;
    const bool lin = (p.sdiv == 1);            // linear addressing whenever the source coordinate is linear in the tap
    if (lin && (g_conv_pipe == 0 || g_conv_pipe == 3)) {
        hipLaunchKernelGGL((conv_igemm_kernel<T, BN, 0, true>), dim3(grid), dim3(256), 0, st, p);
    } else if (lin && g_conv_pipe == 2) {
        hipLaunchKernelGGL((conv_igemm_kernel<T, BN, 2, true>), dim3(grid), dim3(256), 0, st, p);
    } else if (lin) {
        hipLaunchKernelGGL((conv_igemm_kernel<T, BN, 1, true>), dim3(grid), dim3(256), 0, st, p);
    } else {   // strided data-gradient: register staging, 1-deep (the 2-deep form spills there)
        hipLaunchKernelGGL((conv_igemm_kernel<T, BN, 1, false>), dim3(grid), dim3(256), 0, st, p);
    }
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}

}  // namespace

extern int g_ce_span;     // loss.hip
extern int g_bn_reverse;  // elementwise.hip
extern int g_bn_rows;
extern int g_bn_bwd_rows;
extern int g_bn_nt;
extern int g_colreduce_blocks;

extern int g_last_wgrad_kernel;   // conv_wgrad.hip
extern int g_last_wgrad_group;

extern "C" int eeseg_last_kernel(int which) {
    if (which == 0) return g_last_conv_kernel;
    if (which == 1) return g_last_wgrad_kernel;
    if (which == 2) return g_last_stats_rows;
    if (which == 3) return g_last_wgrad_group;
    return EESEG_ERR_ARG;
}

extern "C" int eeseg_set_option(int key, int value) {
    if (key == EESEG_OPT_CONV_PIPE && (value >= 0 && value <= 3)) {
        g_conv_pipe = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_SWP && (value == 0 || value == 1)) {
        g_conv_big_swp = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_MFMA16 && (value == 0 || value == 1)) {
        g_conv_big_m16 = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_COUT_GROUP && (value == 0 || value == 1 || value == 2 || value == 4 || value == 8)) {
        g_conv_big_cg = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_AUTO_NARROW && (value == 0 || value == 1)) {
        g_conv_auto_narrow = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_NARROW_MAX && value >= 64 && value <= 4096) {
        g_conv_narrow_max = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_SPLIT_MIN_K && value >= 1 && value <= 64) {
        g_conv_big_split_min_k = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_COLREDUCE_BLOCKS && value >= 0 && value <= 65536) {
        g_colreduce_blocks = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_BN_BWD_ROWS && (value == 1 || value == 2 || value == 4)) {
        g_bn_bwd_rows = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_BN_NT && value >= 0 && value <= 3) {
        g_bn_nt = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_BN_ROWS && (value == 1 || value == 2 || value == 4)) {
        g_bn_rows = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_BN_REVERSE && value >= 0 && value <= 3) {
        g_bn_reverse = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_CUS && value >= 32 && value <= 256) {
        g_conv_big_cus = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_TAIL_MERGE && (value == 0 || value == 1)) {
        g_conv_big_merge = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CE_SPAN && (value == 0 || value == 1)) {
        g_ce_span = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_TAIL_MIN && value >= 0 && value <= 256) {
        g_conv_big_tail_min = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_TAP_INNER && (value == 0 || value == 1)) {
        g_conv_linear = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_PW_MAX_K && value >= 0 && value <= 65536) {
        g_conv_pw_max_k = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_PW_ALL && (value == 0 || value == 1)) {
        g_conv_pw_all = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_PWS && value >= 0 && value <= 5) {
        g_conv_pws = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_SMALL_M && value >= 0 && value <= 2) {
        g_conv_sm = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_SMALL_M_DEEP && value >= 0 && value <= 2) {
        g_conv_sm_deep = value;
        return EESEG_OK;
    }
    if (key == EESEG_OPT_CONV_SMALL_M_MAX_K && value >= 0 && value <= 4096) {
        g_conv_sm_max_nk = value;
        return EESEG_OK;
    }
    eeseg_set_error("set_option: unknown key %d / value %d", key, value);
    return EESEG_ERR_ARG;
}

extern "C" int eeseg_get_option(int key) {
    switch (key) {
        case EESEG_OPT_CONV_PIPE: return g_conv_pipe;
        case EESEG_OPT_CONV_TAP_INNER: return g_conv_linear;
        case EESEG_OPT_CONV_NARROW_MAX: return g_conv_narrow_max;
        case EESEG_OPT_CONV_AUTO_NARROW: return g_conv_auto_narrow;
        case EESEG_OPT_CONV_TAIL_MIN: return g_conv_big_tail_min;
        case EESEG_OPT_CE_SPAN: return g_ce_span;
        case EESEG_OPT_CONV_TAIL_MERGE: return g_conv_big_merge;
        case EESEG_OPT_CONV_CUS: return g_conv_big_cus;
        case EESEG_OPT_BN_REVERSE: return g_bn_reverse;
        case EESEG_OPT_BN_ROWS: return g_bn_rows;
        case EESEG_OPT_BN_BWD_ROWS: return g_bn_bwd_rows;
        case EESEG_OPT_BN_NT: return g_bn_nt;
        case EESEG_OPT_COLREDUCE_BLOCKS: return g_colreduce_blocks;
        case EESEG_OPT_CONV_SPLIT_MIN_K: return g_conv_big_split_min_k;
        case EESEG_OPT_CONV_PW_MAX_K: return g_conv_pw_max_k;
        case EESEG_OPT_CONV_PWS: return g_conv_pws;
        case EESEG_OPT_CONV_PW_ALL: return g_conv_pw_all;
        case EESEG_OPT_CONV_COUT_GROUP: return g_conv_big_cg;
        case EESEG_OPT_CONV_MFMA16: return g_conv_big_m16;
        case EESEG_OPT_CONV_SWP: return g_conv_big_swp;
        case EESEG_OPT_CONV_SMALL_M: return g_conv_sm;
        case EESEG_OPT_CONV_SMALL_M_MAX_K: return g_conv_sm_max_nk;
        case EESEG_OPT_CONV_SMALL_M_DEEP: return g_conv_sm_deep;
    }
    eeseg_set_error("get_option: unknown key %d", key);
    return EESEG_ERR_ARG;
}

extern "C" int eeseg_conv_stats_tiles(int N, int Hout, int Wout) {     // upper bound: the smallest pixel tile any kernel uses is 64
    const long long M = (long long)N * Hout * Wout;
    return (int)((M + 63) / 64);
}

extern "C" int64_t eeseg_conv_workspace(void) {
    return 256ll * SLAB_FLOATS * 4;      // tiles of the split round x K ranges <= 256 slabs of 256 KiB
}

extern "C" int eeseg_conv_igemm(const eeseg_conv_args* a, void* stream) {
    EESEG_CHECK(a && a->x && a->w && a->y, EESEG_ERR_ARG, "conv_igemm: null pointer");
    const int es = eeseg_dtype_size(a->dtype);
    EESEG_CHECK(es != 0, EESEG_ERR_ARG, "conv_igemm: bad dtype %d", a->dtype);
    const int epk = 128 / es;
    EESEG_CHECK(a->N > 0 && a->Hin > 0 && a->Win > 0 && a->Hout > 0 && a->Wout > 0 && a->Cout > 0, EESEG_ERR_ARG,
                "conv_igemm: non-positive shape");
    EESEG_CHECK(a->Cin > 0 && a->Cin % epk == 0, EESEG_ERR_ARG, "conv_igemm: Cin=%d must be a multiple of %d",
                a->Cin, epk);
    EESEG_CHECK(a->R >= 1 && a->S >= 1 && a->R * a->S <= 32, EESEG_ERR_ARG, "conv_igemm: R*S=%d unsupported",
                a->R * a->S);
    EESEG_CHECK(a->sdiv >= 1 && a->smul >= 1, EESEG_ERR_ARG, "conv_igemm: bad stride mapping");
    EESEG_CHECK(a->ldy >= a->Cout, EESEG_ERR_ARG, "conv_igemm: ldy < Cout");
    EESEG_CHECK(!a->residual || a->ldres >= a->Cout, EESEG_ERR_ARG, "conv_igemm: ldres < Cout");
    const long long M = (long long)a->N * a->Hout * a->Wout;
    const long long xbytes = (long long)a->N * a->Hin * a->Win * a->Cin * es;
    const long long wbytes = (long long)a->Cout * a->R * a->S * a->Cin * es;
    EESEG_CHECK(xbytes < (1ll << 31) && wbytes < (1ll << 31) && M < (1ll << 31) - 256, EESEG_ERR_TOO_LARGE,
                "conv_igemm: tensor exceeds 2 GiB descriptor range (x=%lld w=%lld bytes)", xbytes, wbytes);
    EESEG_CHECK(((uintptr_t)a->x & 15) == 0 && ((uintptr_t)a->w & 15) == 0, EESEG_ERR_ARG,
                "conv_igemm: x/w must be 16-byte aligned");

    ConvP p;
    p.x = a->x; p.w = a->w; p.y = a->y;
    p.scale = a->scale; p.shift = a->shift; p.residual = a->residual; p.stats = a->stats;
    p.N = a->N; p.Hin = a->Hin; p.Win = a->Win; p.Cin = a->Cin;
    p.Hout = a->Hout; p.Wout = a->Wout; p.Cout = a->Cout; p.R = a->R; p.S = a->S;
    p.smul = a->smul; p.off_h = a->off_h; p.off_w = a->off_w;
    p.tstep_h = a->tstep_h; p.tstep_w = a->tstep_w; p.sdiv = a->sdiv;
    p.ldy = a->ldy; p.ldres = a->ldres; p.relu = a->relu;
    p.M = (int)M; p.HWout = a->Hout * a->Wout;
    p.m_tiles = (int)((M + BM - 1) / BM);
    p.xbytes = (uint32_t)xbytes; p.wbytes = (uint32_t)wbytes;
    const int epc = 16 / es;
    p.tap_inner = g_conv_linear;
#ifdef EESEG_PW_DIAG
    p.tap_inner = getenv("EESEG_PW_DIAG") ? atoi(getenv("EESEG_PW_DIAG")) : 0;
#endif
    p.n_tiles = 0; p.tile_begin = 0; p.ksplit = 1; p.slabs = nullptr; p.n_split_blocks = 0; p.pointwise = 0;
    p.n_active = a->n_active;
    p.n_gtaps = 0;
    if (a->n_taps > 0) {      // generalised taps: R = 1, S = n_taps describe the weight tensor [Cout][n_taps][Cin]
        EESEG_CHECK(a->taps && a->n_taps <= 32 && a->R == 1 && a->S == a->n_taps, EESEG_ERR_ARG,
                    "conv_igemm: tap table needs taps != NULL, n_taps <= 32, R = 1, S = n_taps");
        EESEG_CHECK(g_conv_pipe == 3 && a->dtype == EESEG_BF16 && a->sdiv == 1 && a->smul == 1 && a->Cout % BIGT == 0 &&
                        a->Hin == a->Hout && a->Win == a->Wout && !a->stats,
                    EESEG_ERR_ARG, "conv_igemm: a tap table runs on the 256-tile bf16 kernel only (Cout %% 256 == 0, stride 1)");
        p.n_gtaps = a->n_taps;
        for (int t = 0; t < a->n_taps; ++t) {
            const long long off = a->taps[3 * t + 2];
            EESEG_CHECK(off >= 0 && off % 16 == 0 && off + xbytes < (1ll << 31) && a->taps[3 * t] > -32768 &&
                            a->taps[3 * t] < 32768 && a->taps[3 * t + 1] > -32768 && a->taps[3 * t + 1] < 32768,
                        EESEG_ERR_ARG, "conv_igemm: bad tap table entry %d", t);
            p.gdh[t] = (short)a->taps[3 * t]; p.gdw[t] = (short)a->taps[3 * t + 1]; p.goff[t] = (int)off;
            if ((uint32_t)(off + xbytes) > p.xbytes) p.xbytes = (uint32_t)(off + xbytes);
        }
        p.off_h = 0; p.off_w = 0;
    }
    p.resmask = a->residual_mask; p.ldmask = a->ld_residual_mask;
    EESEG_CHECK(!a->residual_mask || (a->residual && a->dtype == EESEG_BF16 && a->ld_residual_mask >= a->Cout / 8 &&
                                      a->Cout % 256 == 0 && a->sdiv == 1 && g_conv_pipe == 3),
                EESEG_ERR_ARG, "conv_igemm: a residual mask needs a bf16 residual on the 256-tile / pointwise kernels (Cout %% 256 == 0)");
    EESEG_CHECK(!a->n_active || !a->stats, EESEG_ERR_ARG, "conv_igemm: n_active is an inference feature (no BN statistics)");
    p.vec_ok = (((uintptr_t)a->y & 15) == 0) && (a->ldy % epc == 0) &&
               (!a->residual || ((((uintptr_t)a->residual & 15) == 0) && (a->ldres % epc == 0)));
    hipStream_t st = (hipStream_t)stream;
    const bool affine_ok = (((uintptr_t)a->scale | (uintptr_t)a->shift) & 15) == 0;
    if (g_conv_pipe == 3 && a->dtype == EESEG_BF16 && a->sdiv == 1 && a->Cout % BIGT == 0 && p.vec_ok && affine_ok) {
        p.pointwise = a->R * a->S == 1 && a->smul == 1 && a->off_h == 0 && a->off_w == 0 && a->Hin == a->Hout &&
                      a->Win == a->Wout;
        // HBM-bound 1x1 layers whose traffic is mostly OUTPUT (expanding convs, data-gradients that add into a residual):
        // two 128x256 blocks per CU, so one block's epilogue runs beside the other's K loop (measured at 32 x 65 x 65:
        // 256->1024 150 -> 133 us, with a residual 230 -> 145 us, 512->2048 410 -> 370 us; contracting layers such as
        // 1024->256 stay on the 256-tile kernel: 101 vs 106 us, their K loop dominates and the bigger tile re-reads W less)
        if (p.pointwise && g_conv_pws && a->Cin == WS_K && !a->n_active && !a->scale && !a->shift && a->ldy % 8 == 0 &&
            !(a->residual && a->stats) &&
            (a->Cout >= 2 * a->Cin || a->residual) && M >= 512 * 128) {
            // weight-stationary persistent form (only where a block walks >= 4 pixel tiles: loading its 128 KiB of weights into
            // registers costs 6.6 us): grid = whole groups of 8 blocks per cout tile, two blocks per CU
            p.n_tiles = a->Cout / 256;
            const int per = 8 * p.n_tiles;
            int groups = 512 / per;
            if (groups < 1) groups = 1;
            while (groups > 1 && (groups - 1) * 8 >= p.m_tiles) --groups;     // no more sequences than tiles
#ifdef EESEG_PW_STAMPS
            p.slabs = reinterpret_cast<float*>(a->workspace);
#endif
            if (g_conv_pws == 5 && a->Cout % 512 == 0 && (long long)M * a->ldy * 2 < (1ll << 31) &&
                (!a->residual || (long long)M * a->ldres * 2 < (1ll << 31))) {
                // 512 couts per block, every wave in every role: X makes Cout / 512 passes through the CUs instead of Cout / 256
                p.n_tiles = a->Cout / 512;
                p.m_tiles = (M + 63) / 64;                                    // 64-pixel units = statistic rows of this kernel
#ifdef EESEG_W2_WHATIF
                p.tap_inner = getenv("EESEG_W2_VAR") ? atoi(getenv("EESEG_W2_VAR")) : 0;
#endif
                const int per5 = 8 * p.n_tiles;
                int groups5 = g_conv_big_cus / per5;
                if (groups5 < 1) groups5 = 1;
                while (groups5 > 1 && (groups5 - 1) * 8 >= p.m_tiles) --groups5;
                if (a->residual) hipLaunchKernelGGL(conv_pws3_kernel<true>, dim3((unsigned)(groups5 * per5)), dim3(512), 0, st, p);
                else hipLaunchKernelGGL(conv_pws3_kernel<false>, dim3((unsigned)(groups5 * per5)), dim3(512), 0, st, p);
                g_last_conv_kernel = EESEG_KERNEL_CONV_PWS;
                g_last_stats_rows = p.m_tiles;
                EESEG_LAUNCH_CHECK();
                return EESEG_OK;
            }
            if (g_conv_pws >= 2) {                                            // wave-specialised forms: one block per CU
#ifdef EESEG_W2_WHATIF
                p.tap_inner = getenv("EESEG_W2_VAR") ? atoi(getenv("EESEG_W2_VAR")) : 0;
#endif
                groups = g_conv_big_cus / per;
                if (groups < 1) groups = 1;
                while (groups > 1 && (groups - 1) * 8 >= p.m_tiles) --groups;
                const dim3 grid2((unsigned)(groups * per));
                if (g_conv_pws == 4) {                                        // eight MFMA waves of 32 couts + eight output waves
                    if (a->residual) hipLaunchKernelGGL((conv_pws2_kernel<true, 8, 8>), grid2, dim3(1024), 0, st, p);
                    else hipLaunchKernelGGL((conv_pws2_kernel<false, 8, 8>), grid2, dim3(1024), 0, st, p);
                } else if (g_conv_pws == 3) {                                 // eight MFMA waves of 32 couts (two per SIMD) + four output waves
                    if (a->residual) hipLaunchKernelGGL((conv_pws2_kernel<true, 8, 4>), grid2, dim3(768), 0, st, p);
                    else hipLaunchKernelGGL((conv_pws2_kernel<false, 8, 4>), grid2, dim3(768), 0, st, p);
                } else if (a->residual) hipLaunchKernelGGL((conv_pws2_kernel<true, 4, 4>), grid2, dim3(512), 0, st, p);
                else hipLaunchKernelGGL((conv_pws2_kernel<false, 4, 4>), grid2, dim3(512), 0, st, p);
            } else if (a->residual) hipLaunchKernelGGL(conv_pws_kernel<true>, dim3((unsigned)(groups * per)), dim3(256), 0, st, p);
            else hipLaunchKernelGGL(conv_pws_kernel<false>, dim3((unsigned)(groups * per)), dim3(256), 0, st, p);
            g_last_conv_kernel = EESEG_KERNEL_CONV_PWS;
            g_last_stats_rows = p.m_tiles;
            EESEG_LAUNCH_CHECK();
            return EESEG_OK;
        }
        // small M (the 4-images-per-GPU shard of an 8-GPU run: 67 tiles of 256 pixels): on the 256-tile kernel EVERY tile of a
        // contracting layer would be a K-split + fix-up launch; the 128x256 kernel covers the same layer in one launch of
        // whole tiles (measured at 4 x 65 x 65, R101 step: 24.32 -> 23.74 ms; neutral at 8 images)
        const bool small_m = ((M + BIGT - 1) / BIGT) * (a->Cout / BIGT) <= g_conv_big_cus / 2;
        const bool sm = small_m && g_conv_sm;
        if (p.pointwise && a->Cin % 32 == 0 && (a->Cin <= g_conv_pw_max_k || (sm && a->Cin / 32 <= g_conv_sm_max_nk)) &&
            (g_conv_pw_all || a->Cout >= 2 * a->Cin || a->residual || small_m)) {
            p.n_tiles = a->Cout / PW_BN;
#ifdef EESEG_PW_STAMPS
            p.slabs = reinterpret_cast<float*>(a->workspace);
#endif
            // round 4: at small M the pixel tile is chosen so that one round of whole tiles covers the chip (4 x 65 x 65: 96 pixels)
            // (wide: layers of at most 6 rounds of 128-pixel blocks pick their tile by rounds too)
            const bool few = g_conv_sm == 2 && ((M + PW_BM - 1) / PW_BM) * p.n_tiles <= 6ll * g_conv_big_cus;
            return launch_pw(p, M, sm ? pick_small_bm(M, p.n_tiles, g_conv_big_cus, g_conv_sm == 2)
                                      : few ? pick_small_bm(M, p.n_tiles, g_conv_big_cus, true) : PW_BM, false, st);
        }
        // round 4: the 3x3 / dilated layers of a shard on the same kernel with a tap loop - one launch of whole tiles where the
        // 256-tile kernel would run EVERY tile as K ranges + a fix-up launch (4 x 65 x 65, 3x3 256->256: 56 -> see DESIGN.md)
        if (sm && !p.pointwise && a->n_taps == 0 && a->Cin % 32 == 0 && a->R * a->S * (a->Cin / 32) <= g_conv_sm_max_nk) {
            p.n_tiles = a->Cout / PW_BN;
#ifdef EESEG_PW_STAMPS
            p.slabs = reinterpret_cast<float*>(a->workspace);
#endif
            return launch_pw(p, M, pick_small_bm(M, p.n_tiles, g_conv_big_cus, g_conv_sm == 2), true, st);
        }
        p.n_tiles = a->Cout / BIGT;             // p.m_tiles stays the 128-pixel count (stats rows)
        g_last_conv_kernel = EESEG_KERNEL_CONV_BIG;
        g_last_stats_rows = p.m_tiles;
        return launch_big(p, M, st, a->workspace, a->workspace_bytes);
    }
    // 128x64 tiles for thin outputs, and whenever the 128x128 grid would fill less than ~85 % of the
    // 512 resident block slots (small batches: M = 4*65*65 gives only 133 pixel tiles)
    const long long wide_blocks = ((M + BM - 1) / BM) * ((a->Cout + 127) / 128);
    const bool narrow = a->Cout <= g_conv_narrow_max || (g_conv_auto_narrow && wide_blocks < 448 && a->Cout > 64);
    p.n_tiles = narrow ? (a->Cout + 63) / 64 : (a->Cout + 127) / 128;
    g_last_conv_kernel = narrow ? EESEG_KERNEL_CONV_IGEMM_64 : EESEG_KERNEL_CONV_IGEMM_128;
    g_last_stats_rows = p.m_tiles;
    if (a->dtype == EESEG_BF16) return narrow ? launch<bf16_t, 64>(p, st) : launch<bf16_t, 128>(p, st);
    return narrow ? launch<float, 64>(p, st) : launch<float, 128>(p, st);
}
