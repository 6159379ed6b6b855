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

namespace {

struct WgP {
    const void* x; const void* dy; float* dw;
    int N, Hin, Win, Cin, Hout, Wout, Cout, R, S, stride, pad, dil;
    int lddy;
    int M, HWout, co_tiles, ci_tiles, splits, chunk;
    uint32_t xbytes, dybytes;
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
    auto advance = [&](int steps) {
        pm += steps * KP;
        pw += steps * KP;
        while (pw >= p.Wout) { pw -= p.Wout; ph += 1; }
        while (ph >= p.Hout) { ph -= p.Hout; pn += 1; }
    };
    // Block-uniform: does K step `kt` contain a pixel row whose tap-shifted source row is inside the
    // image?  For the atrous convs (|dh| = 12/24/36 on 65 rows) up to 55 % of the steps of the off-centre
    // tap rows multiply only zero padding and are skipped.
    auto step_valid = [&](int kt) {
        const int m_first = ps + kt * KP;
        const int m_last = min(pe, m_first + KP) - 1;
        const int r_first = m_first / p.Wout, r_last = m_last / p.Wout;     // global output-row indices
        for (int rr = r_first; rr <= r_last; ++rr) {
            const int hi = (rr % p.Hout) * p.stride + dh;
            if ((unsigned)hi < (unsigned)p.Hin) return true;
        }
        return false;
    };
    auto next_valid = [&](int kt) {
        while (kt < nk && !step_valid(kt)) ++kt;
        return kt;
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
    int kt = next_valid(0);
    const bool any = kt < nk;
    if (any) {
        advance(kt);
        load_tile();
        store_tile(0);
    }
    __syncthreads();
    int cur = 0;
    while (kt < nk) {
        const int nx = next_valid(kt + 1);
        const bool has_next = nx < nk;
        if (has_next) {
            advance(nx - kt);
            load_tile();
        }
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

int g_wgrad_target_blocks = 1024;   // tiles * splits aimed at (eeseg_set_wgrad_target_blocks)

}  // namespace

extern "C" int eeseg_set_wgrad_target_blocks(int blocks) {
    EESEG_CHECK(blocks >= 64 && blocks <= 65535, EESEG_ERR_ARG, "set_wgrad_target_blocks: out of range");
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
    p.x = a->x; p.dy = a->dy; p.dw = a->dw;
    p.N = a->N; p.Hin = a->Hin; p.Win = a->Win; p.Cin = a->Cin;
    p.Hout = a->Hout; p.Wout = a->Wout; p.Cout = a->Cout; p.R = a->R; p.S = a->S;
    p.stride = a->stride; p.pad = a->pad; p.dil = a->dil; p.lddy = a->Cout;
    p.M = (int)M; p.HWout = a->Hout * a->Wout;
    p.co_tiles = (a->Cout + 127) / 128; p.ci_tiles = (a->Cin + 127) / 128;
    p.xbytes = (uint32_t)xbytes; p.dybytes = (uint32_t)dybytes;
    const int kp = 128 / es;
    const long long tiles = (long long)p.co_tiles * p.ci_tiles * taps;
    long long splits = (g_wgrad_target_blocks + tiles - 1) / tiles;
    const long long max_splits = (M + 4 * kp - 1) / (4 * kp); // at least 4 K steps per block
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long long chunk = (M + splits - 1) / splits;
    chunk = (chunk + kp - 1) / kp * kp;
    splits = (M + chunk - 1) / chunk;
    p.splits = (int)splits; p.chunk = (int)chunk;
    const long long grid = tiles * splits;
    EESEG_CHECK(grid < (1ll << 31), EESEG_ERR_TOO_LARGE, "conv_wgrad: grid too large");
    if (a->dtype == EESEG_BF16)
        hipLaunchKernelGGL((conv_wgrad_kernel<bf16_t>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<float>), dim3((unsigned)grid), dim3(256), 0, st, p);
    EESEG_LAUNCH_CHECK();
    return EESEG_OK;
}
