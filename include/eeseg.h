/* libeeseg - C ABI of the MI355X (gfx950) early-exit DeepLabV3 hot path.
 *
 * The reference (MateusGilbert/ee_semantic_segmentation) has NO FFI layer: every
 * device op is reached implicitly through torch/torchvision.
 * Each entry point below names the reference call site whose implicit device
 * work it replaces (file:line under /root/reference).
 *
 * Conventions
 *  - return 0 on success, <0 = EESEG_ERR_*; eeseg_last_error() gives the text
 *    (thread local).
 *  - the caller owns every buffer (device pointers, 16-byte aligned), passes
 *    explicit shapes, and a hipStream_t as `void* stream`.  No hidden
 *    allocation, no device synchronisation inside any call.
 *  - activations are NHWC ("channels last"): x[n][h][w][c], c contiguous.
 *    `dtype` selects the activation/weight element type of the call:
 *    EESEG_F32 (exact fp32 MFMA, parity mode) or EESEG_BF16 (bf16 MFMA, fp32
 *    accumulate, throughput mode).  Statistics, losses, counters, gradients of
 *    weights and optimizer state are always fp32.
 *  - conv weights are "KRSC": w[cout][r][s][cin], cin contiguous.
 */
#ifndef EESEG_H
#define EESEG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { EESEG_F32 = 0, EESEG_BF16 = 1 };
enum {
    EESEG_OK = 0,
    EESEG_ERR_ARG = -1,       /* bad shape / alignment / unsupported combination */
    EESEG_ERR_HIP = -2,       /* HIP runtime error (launch failed, ...) */
    EESEG_ERR_TOO_LARGE = -3  /* a buffer exceeds the 2 GiB buffer-descriptor range */
};

const char* eeseg_last_error(void);
int eeseg_version(void);
/* tuning switches (process wide).  EESEG_OPT_CONV_PIPE: global-load prefetch depth
 * of the implicit-GEMM conv kernel: 0 = tiles staged by LDS-DMA (buffer_load ... lds, no staging
 * registers / ds_write; 1 K-step in flight); 1 or 2 = K-steps through staging registers; 3 = bf16 stride-1-gather convs
 * with Cout % 256 == 0 use the 256x256-tile kernel (8 waves, LDS-DMA loads in flight across raw barriers, counted vmcnt),
 * everything else as 0 (default). */
enum { EESEG_OPT_CONV_SMALL_M_DEEP = 23 /* 1 (default): a launch of the 128x256 kernel with at most one block per CU uses six LDS-DMA stages (144 KiB, five K tiles in flight) instead of three: a lone block streams at bytes-in-flight / L2 latency; 2 = the same with EIGHT waves for tiles of <= 96 pixels (two wave groups, each multiplies one of the two k-steps of every K tile, partial sums added through LDS): measured neutral (33.9 vs 32.0 us on 3x3 256->256 at 4 images - the loop is bound by the 22 KiB every block pulls from L2 per K tile, not by instruction issue), kept as an A/B switch */,
       EESEG_OPT_CONV_SMALL_M_MAX_K = 22 /* ... when the K loop has at most this many 32-channel tiles, taps x Cin / 32 (default 160: everything of a ResNet shard but the 2048-wide atrous convs, whose long K loops stay on the 256-tile kernel) */,
       EESEG_OPT_CONV_SMALL_M = 21 /* 2 (default) = 1 + pointwise layers of at most 6 rounds of 128-pixel blocks pick their pixel tile by ROUNDS of blocks too (4 x 65 x 65 x 256 -> 1024: 708 blocks of 96 pixels instead of 532 of 128 on 512 slots); 1: bf16 layers with <= CUs / 2 tiles of 256 x 256 (the 4-8 image shards of a data-parallel run) run on the 128x256 kernel with a 64 / 96 / 128 pixel tile chosen so that ONE round of whole tiles covers the chip, 3x3 / dilated layers included (tap loop); 0 = round-3 dispatch (K-split tiles + fix-up launch on the 256-tile kernel) */,
       EESEG_OPT_BN_NT = 20 /* bit 0: bn_apply, bit 1: bn_bwd_apply load the tensors that are dead after the pass (conv output / residual; dy / conv output) with a nontemporal hint (default 0) */,
       EESEG_OPT_CONV_SWP = 19 /* 256x256 conv kernel, 16x16x32 form: 1 (default) = software-pipelined K loop (the LDS reads of phase p+1 sit between the MFMAs of phase p, the eight waves in lockstep, one barrier per phase); 0 = two wave groups half a phase apart; same bits */,
       EESEG_OPT_BN_BWD_ROWS = 18 /* bn_bwd_apply / scale_act_bwd: rows whose loads a thread keeps in flight (1 = default, 2, 4) */,
       EESEG_OPT_CONV_MFMA16 = 17 /* 256x256 conv kernel: 1 (default) = v_mfma_f32_16x16x32_bf16, 0 = v_mfma_f32_32x32x16_bf16 (same tile, same LDS image, same cycles per FLOP; the chip holds a higher clock on the 16x16 shape: 3-7 % faster) */,
       EESEG_OPT_CONV_COUT_GROUP = 16 /* 256x256 conv kernel, layers with more cout tiles than this: the 32 CUs of an XCD work on `value` cout tiles x 32/value pixel tiles at a time (1, 2, 4, 8; default 0 = all cout tiles of few pixel tiles; an A/B switch, measured neutral on the 8-cout-tile layers) */,
       EESEG_OPT_CONV_PW_ALL = 15 /* 1: every eligible pointwise bf16 layer (Cin <= EESEG_OPT_CONV_PW_MAX_K) on the 128x256 kernel; 0 (default): the rule in eeseg_conv_igemm */,
       EESEG_OPT_CONV_PWS = 14 /* 1 (default): expanding pointwise bf16 layers with Cin = 256 run on the weight-stationary persistent kernel (weights in registers, only activations stream); 0 = the 128x256 kernel; 2 / 3 / 4 = the same layer with MFMA waves and output waves (4 + 4, 8 + 4, 8 + 8 waves per block; 2 and 3 bit-identical to 1), 5 = 512 couts per block (Cout % 512 == 0; one statistics row per 64 pixels) - measured alternatives of round 4, all within 10 % of 1 */,
       EESEG_OPT_CONV_PW_MAX_K = 13 /* pointwise (1x1, stride 1) bf16 layers with Cout % 256 == 0 and Cin <= value run on the 128x256 two-blocks-per-CU kernel (default 1280; 0 = never) */,
       EESEG_OPT_CONV_SPLIT_MIN_K = 12 /* 256x256 conv kernel: K tiles per K range of a split tail tile, at least (default 4) */,
       EESEG_OPT_COLREDUCE_BLOCKS = 11 /* column reductions (BN backward sums, channel statistics): blocks aimed at in all (default 512); 0 = up to 1024 row blocks per column block */,
       EESEG_OPT_BN_ROWS = 10 /* bn_apply: rows whose loads a thread keeps in flight (1, 2 = default, 4) */,
       EESEG_OPT_BN_REVERSE = 9 /* bit 0: bn_apply, bit 1: bn_bwd_apply sweep the rows from the end (the part the producing kernel touched last is still in the caches) */,
       EESEG_OPT_CONV_CUS = 8 /* 256x256 conv kernel: CUs a launch may count on when it sizes its rounds and its K-split tail (default 256; lower it while collectives hold CUs) */,
       EESEG_OPT_CONV_TAIL_MERGE = 7 /* 256x256 kernel: K-split tail blocks and full rounds in ONE launch (default 1) */,
       EESEG_OPT_CE_SPAN = 6 /* fused cross entropy: 1 = one thread per span (default), 0 = one half wave per pixel */,
       EESEG_OPT_CONV_TAIL_MIN = 5 /* 256x256 kernel: a last round with fewer tiles than this is split along K (default 224; 0 = never) */,
       EESEG_OPT_CONV_AUTO_NARROW = 4 /* 1: 128x64 tiles when the 128x128 grid underfills the chip (default 0) */,
       EESEG_OPT_CONV_NARROW_MAX = 3 /* layers with Cout <= value use the 128x64 tile (default 64) */,
       EESEG_OPT_CONV_PIPE = 1, EESEG_OPT_CONV_TAP_INNER = 2 /* K order: 0 = taps outer (default), 1 = taps inner (fewer L2 misses,
                                  measured 3% slower end to end on MI355X: the Infinity Cache absorbs the re-reads) */ };
int eeseg_set_option(int key, int value);
int eeseg_get_option(int key);   /* current value, or a negative error code */
/* Which kernel the last eeseg_conv_igemm (which = 0) / eeseg_conv_wgrad (which = 1) call of this process launched - host-side
 * state for measurement code that attributes a timed call to a kernel (bench.py's per-kernel roofline); 0 = none yet.
 * which = 2: the number of `stats` rows the last eeseg_conv_igemm call wrote (depends on the pixel tile of the kernel chosen);
 * which = 3: the number of problems the last eeseg_conv_wgrad_group call put into ONE launch (0 = it issued them one by one). */
enum { EESEG_KERNEL_CONV_IGEMM_128 = 1, EESEG_KERNEL_CONV_IGEMM_64 = 2, EESEG_KERNEL_CONV_BIG = 3 /* 256x256 tile (+ K-split tail, fix-up) */,
       EESEG_KERNEL_CONV_PW = 4 /* 128x256 pointwise */, EESEG_KERNEL_CONV_PWS = 5 /* weight-stationary pointwise */,
       EESEG_KERNEL_WGRAD_128 = 6, EESEG_KERNEL_WGRAD_BIG = 7 };
int eeseg_last_kernel(int which);
/* upper bound on the grid of the column-fixed BatchNorm elementwise kernels (tuning) */
int eeseg_set_ew_grid_cap(int blocks);
/* split-K sizing of the 128x128-tile weight-gradient kernel: number of blocks (tiles x pixel splits) aimed at; 0 (default) =
 * chosen per layer by a cost model (K steps per block vs the fp32 partial tile every block adds with float atomics) */
int eeseg_set_wgrad_target_blocks(int blocks);
/* bf16 weight gradients with Cout % 256 == 0 and Cin % 256 == 0: 1 (default) = 256x256-tile kernel when every block
 * gets at least 20 K tiles (64 pixels each), 2 = always, 0 = never (128x128-tile kernel); +4 = combine the K splits through
 * the workspace slabs (reproducible, 2-5 % slower) instead of fp32 atomics; +8 (default on) = combine them INSIDE the kernel when
 * its whole grid is resident and eeseg_wgrad_args.barrier_state is given (reproducible; round 4); +16 = the 256x256-tile kernel computes with
 * v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (same results up to fp32 summation order; measured neutral, default off) */
int eeseg_set_wgrad_big(int on);
/* K-split sizing of the 256x256 weight-gradient kernel: aim at `blocks` concurrent blocks (default 256 = one per CU) and
 * at most `rounds` rounds of them (default 8).  Fewer blocks = fewer fp32 partial tiles to combine; use with the weight
 * gradient on a side stream so that the other CUs are not idle. */
int eeseg_set_wgrad_big_grid(int blocks, int rounds);
/* the 256x256 weight-gradient kernel is used when every block gets at least this many K tiles (default 8; 20 before the in-kernel combine of round 4) */
int eeseg_set_wgrad_big_min_ktiles(int n);

/* ---------------------------------------------------------------- conv ----
 * Implicit-GEMM convolution.  Replaces F.conv2d reached via torchvision
 * Bottleneck / ASPP / DeepLabHead (from_deepv3_new.py:146-151 -> SURVEY 2.2).
 *
 * y[n,ho,wo,co] = sum_{r,s,ci} x[n, hi, wi, ci] * w[co,r,s,ci]
 *   with hnum = ho*smul + off_h + r*tstep_h ; valid iff hnum % sdiv == 0 and
 *   0 <= hnum/sdiv < Hin (same for w).  Forward conv: smul=stride, off=-pad,
 *   tstep=dilation, sdiv=1.  Data-gradient: x:=dY, w:=transposed weights
 *   [cin][r][s][cout], smul=1, off=+pad, tstep=-dilation, sdiv=stride.
 * Epilogue (all optional): v = acc*scale[co] + shift[co]; v += residual;
 *   v = relu(v); and per-channel partial sums of v, v^2 over each 128-pixel
 *   tile -> stats[tile][2][Cout] (fp32) for train-mode BatchNorm.
 * Cin must be a multiple of 64 (bf16) / 32 (f32).  ldy = output row stride in
 * elements (>= Cout; lets a conv write into a channel slice of a wider tensor).
 */
typedef struct {
    const void* x; const void* w; void* y;
    const float* scale; const float* shift; const void* residual; float* stats;
    int N, Hin, Win, Cin, Hout, Wout, Cout, R, S;
    int smul, off_h, off_w, tstep_h, tstep_w, sdiv;
    int ldy, ldres, relu, dtype;
    void* workspace;              /* optional scratch (NULL = none): lets the 256x256-tile kernel split the K loop of its */
    int64_t workspace_bytes;      /* last, partly filled round of tiles over the idle CUs; eeseg_conv_workspace() bytes suffice */
    const int32_t* n_active;      /* optional DEVICE int32 (NULL = all): only the first *n_active images of the batch are computed -
                                     blocks whose pixels all belong to later images return at once.  Batched progressive
                                     early-exit inference keeps the images still in flight in the leading slots and
                                     their count on the device, so no exit decision is ever read back by the host. */
    const uint8_t* residual_mask; /* optional (NULL = none): the residual is multiplied by a 1-bit mask before it is added - one byte */
    int ld_residual_mask;         /* per 16-byte chunk of a residual row, bit e = element e (the ReLU mask eeseg_bn_apply_relu_mask
                                     writes), ld = bytes per mask row.  Lets the data-gradient of a bottleneck's first conv add the
                                     masked block gradient itself, so BatchNorm backward need not write it (bf16, Cout % 256 == 0). */
    int n_taps;                   /* 0 = a regular R x S convolution.  > 0: GENERALISED taps (<= 32; bf16 256-tile kernel, stride 1, */
    const int32_t* taps;          /* Cout % 256 == 0): HOST array [n_taps][3] = {dh, dw, byte offset}; tap t multiplies weight slice
                                     w[co][t][:] (w = [Cout][n_taps][Cin], pass R = 1, S = n_taps) with source pixel (h + dh, w + dw) of the
                                     [N,Hin,Win,Cin] tensor that starts `byte offset` behind x (zero outside the image).  One launch sums
                                     convolutions of different dilation / different inputs into one output: the four data-gradients of an
                                     ASPP head (torchvision ASPP: 1x1 + three atrous 3x3, from_deepv3_new.py:13,131) write dx once. */
} eeseg_conv_args;
int64_t eeseg_conv_workspace(void);
/* rows to ALLOCATE for `stats` (upper bound over the kernels: one row per 64 pixels); the rows a call actually wrote =
 * eeseg_last_kernel(2) right after it (one row per pixel tile of the kernel it chose; the reduction takes exactly those) */
int eeseg_conv_stats_tiles(int N, int Hout, int Wout);
int eeseg_conv_igemm(const eeseg_conv_args* a, void* stream);

/* Weight gradient: dw[co][r][s][ci] (+)= sum_{n,ho,wo} dy[n,ho,wo,co]*x[n,hi,wi,ci]
 * (forward geometry).  dw is fp32 KRSC; accumulate=0 zero-fills it first.
 * Replaces the conv weight-grad inside `l.mean().backward()` (train_funcs.py:26). */
typedef struct {
    const void* x; const void* dy; float* dw;
    int N, Hin, Win, Cin, Hout, Wout, Cout, R, S;
    int stride, pad, dil, dtype, accumulate;
    void* workspace;              /* optional scratch (NULL = none) for eeseg_set_wgrad_big(.. | 4): the 256x256-tile kernel then combines */
    int64_t workspace_bytes;      /* its K splits through plain-store slabs + a fixed-order reduce (bitwise reproducible) instead of fp32
                                     atomics; eeseg_wgrad_workspace() bytes suffice */
    void* barrier_state;          /* optional (NULL = none; round 4): with `workspace`, lets the 256x256-tile kernel combine its K splits
                                     INSIDE the launch when its whole grid is resident (one block per CU): slabs published write-through,
                                     a barrier per output tile, every split sums its share in split order - no atomics (~49 us per call),
                                     no second launch, bitwise reproducible.  4128+ 32-bit words as eeseg_bn_bwd_coop's (8224: 128 groups),
                                     128-byte aligned, zeroed ONCE by the caller, owned by the weight-gradient calls of one stream. */
} eeseg_wgrad_args;
int64_t eeseg_wgrad_workspace(void);
int eeseg_conv_wgrad(const eeseg_wgrad_args* a, void* stream);
/* Round 4: the weight gradients of up to 4 convolutions in ONE launch (the three convs of a bottleneck block).  Replaces the same
 * autograd step as eeseg_conv_wgrad (torch's conv backward reached from from_deepv3_new.py:146-151), `n` times.  At a per-GPU shard of
 * a few images one weight gradient has too little K to fill the chip - every call costs ~32 us whatever its block count; side by side
 * each problem gets a share of the CUs in proportion to its work, ~4x the K per block, a quarter of the combine traffic.  Taken when
 * EVERY problem qualifies for the 256x256 kernel's in-kernel combine (bf16, Cin and Cout multiples of 256, the same `workspace` and
 * `barrier_state` in all of them, output width on the same side of 64), has at most eeseg_set_wgrad_group() K tiles of 64 pixels
 * (default: no limit; 0 = never) and the launch's longest K range beats the single calls by the library's cost model (small shards:
 * always; the 68 output tiles of a layer-4 block at 32 images: no): otherwise - and that is not an error - the calls are issued one by
 * one exactly as eeseg_conv_wgrad would.  eeseg_last_kernel(3) = problems the last call put into one launch (0 = one by one).  Results are bitwise
 * reproducible; they differ from the single calls' in the last bits (another K split). */
int eeseg_conv_wgrad_group(const eeseg_wgrad_args* a, int n, void* stream);
int eeseg_set_wgrad_group(int max_ktiles);

/* fp32 master weight -> compute-dtype KRSC (`w_fwd` [Cout_pad][R][S][Cin], may be
 * NULL) and transposed CRSK (`w_bwd` [Cin][R][S][Cout_pad], may be NULL; the
 * "weights" of the data-gradient call).  Rows co >= Cout are zero filled.
 * src_krsc: 1 = src is [co][r][s][ci] (torch channels_last parameter), 0 = torch
 * default [co][ci][r][s]. */
int eeseg_pack_weight(const float* src, void* w_fwd, void* w_bwd, int Cout, int Cout_pad, int Cin, int R, int S,
                      int src_krsc, int dtype, void* stream);
/* The same for n weights in ONE launch.  desc_table: device array of n 48-byte records
 * { const float* src; void* w_fwd; void* w_bwd; int32 Cout, Cout_pad, Cin, taps, src_krsc, pad; }. */
int eeseg_pack_weight_multi(const void* desc_table, int n, int dtype, void* stream);
/* fp32 [rows][cols] (row stride ld_src) -> dtype [rows_pad][cols_pad], zero padded
 * (stem weight [64][147] -> [64][192] for the im2col GEMM). */
int eeseg_pack_matrix(const float* src, int rows, int cols, int ld_src, void* dst, int rows_pad, int cols_pad,
                      int dtype, void* stream);

/* Stem: NCHW fp32 image -> im2col rows [N*Ho*Wo][Kpad] (k = (r*S+s)*C+ci, zero
 * padded to Kpad) so the 7x7/s2 stem conv (Cin=3) runs as a GEMM.
 * Replaces conv1 of the torchvision ResNet stem (from_deepv3_new.py:77-79). */
int eeseg_im2col_nchw(const float* x, void* col, int N, int C, int H, int W, int R, int S, int stride,
                      int pad, int Ho, int Wo, int Kpad, int dtype, void* stream);

/* ----------------------------------------------------------- batchnorm ----
 * Train-mode BatchNorm2d split as: conv epilogue partial sums -> reduce ->
 * (all-reduce under SyncBN) -> finalize -> apply(+residual, ReLU).  Replaces
 * F.batch_norm / relu / `out += identity` in torchvision Bottleneck / ASPP
 * (SURVEY 2.2).  count = N*H*W (global count under SyncBN).  mean_invstd[2][C]
 * is saved for backward; scale_shift[2][C] feeds eeseg_bn_apply.  Running stats
 * are updated with `momentum` and the unbiased variance when running_mean != NULL.
 * Column reductions over more than one row block need a workspace of
 * eeseg_colreduce_workspace(rows, C) bytes. */
int64_t eeseg_colreduce_workspace(int64_t rows, int C);
/* partials[tiles][KC] -> sums[KC] (fixed order, double accumulation); an optional
 * workspace of >= 32*KC floats enables the two-level reduction for many tiles */
int eeseg_bn_reduce_partials(const float* partials, int tiles, int KC, float* sums, void* workspace,
                             int64_t workspace_bytes, void* stream);
int eeseg_bn_finalize(const float* sums /*[2][C]*/, double count, const float* gamma, const float* beta,
                      float eps, float momentum, float* running_mean, float* running_var,
                      float* mean_invstd, float* scale_shift, int C, void* stream);
/* reduce + finalize in one launch (local BatchNorm: partials[tiles][2][C] straight from the conv epilogue) */
int eeseg_bn_reduce_finalize(const float* partials, int tiles, double count, const float* gamma, const float* beta,
                             float eps, float momentum, float* running_mean, float* running_var, float* mean_invstd,
                             float* scale_shift, int C, void* stream);
/* eval mode: scale/shift from running stats */
int eeseg_bn_eval_scale_shift(const float* gamma, const float* beta, const float* running_mean,
                              const float* running_var, float eps, float* scale_shift, int C, void* stream);
/* y = act(x*scale[c] + shift[c] (+ residual)); x rows have stride ldx, y rows ldy
 * (elements).  out_dtype may differ from in_dtype (bf16 -> f32 before the
 * classifier).  rows = N*H*W. */
int eeseg_bn_apply(const void* x, int ldx, const float* scale_shift, const void* residual, int ldres,
                   void* y, int ldy, int64_t rows, int C, int relu, int in_dtype, int out_dtype, void* stream);
/* bn_apply with relu = 1 that also writes the ReLU mask of the backward: one byte per 16-byte channel chunk
 * (8 bf16 / 4 fp32 channels), bit e = (stored y[chunk*epc + e] > 0); relu_mask[rows][C/epc].  Layers with a
 * residual input cannot recompute the mask from x alone: with the byte mask their backward streams 1/16 of y. */
int eeseg_bn_apply_relu_mask(const void* x, int ldx, const float* scale_shift, const void* residual, int ldres,
                             void* y, int ldy, void* relu_mask, int64_t rows, int C, int dtype, void* stream);
/* bn_finalize + bn_apply(_relu_mask) in ONE launch: the coefficients are derived inside the apply pass from the [2][C]
 * sums (the arithmetic of eeseg_bn_finalize, bit for bit; mean_invstd / scale_shift / running statistics are written as
 * there).  For the SyncBN path, where the sums come out of a collective and a separate finalize launch would sit on the
 * critical path of every layer (torch BatchNorm2d reached from from_deepv3_new.py:146-151).  relu_mask optional. */
int eeseg_bn_finalize_apply(const void* x, int ldx, const float* sums, double count, const float* gamma, const float* beta,
                            float eps, float momentum, float* running_mean, float* running_var, float* mean_invstd,
                            float* scale_shift, const void* residual, int ldres, void* y, int ldy, void* relu_mask,
                            int64_t rows, int C, int relu, int dtype, void* stream);
/* eeseg_bn_reduce_finalize + eeseg_bn_apply(_relu_mask) in ONE launch (round 4) for small tensors - the per-GPU shards of a
 * data-parallel run: every block reduces the conv-epilogue partial sums of its own 64 (bf16) / 32 (fp32) channels itself, in
 * the summation order of eeseg_bn_reduce_finalize, and applies the coefficients to its rows: bit-identical with the two calls
 * (torch BatchNorm2d forward reached from from_deepv3_new.py:146-151).  eeseg_bn_fwd_fused_ok(rows, C, tiles, dtype): C a
 * multiple of 64 / 32 and at most 320 partial-sum rows. */
int eeseg_bn_fwd_fused_ok(int64_t rows, int C, int tiles, int dtype);
int eeseg_bn_fwd_fused(const void* x, int ldx, const float* partials /*[tiles][2][C]*/, int tiles, double count,
                       const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                       float* running_var, float* mean_invstd, float* scale_shift, const void* residual, int ldres,
                       void* y, int ldy, void* relu_mask, int64_t rows, int C, int relu, int dtype, void* stream);
/* per-channel sums of x and x^2 over rows (tensors that did not come out of the
 * conv epilogue, e.g. the pooled ASPP branch): sums[2][C] */
int eeseg_channel_stats(const void* x, int ldx, int64_t rows, int C, float* sums, int dtype, void* workspace,
                        int64_t workspace_bytes, void* stream);
/* backward, step 1: g = dy * mask; sums[0][c] = sum g, sums[1][c] = sum g*xhat.
 * relu: 0 no activation; 1 mask = (y > 0) read from the stored output; 2 mask recomputed as
 * (x*scale+shift > 0) from scale_shift[2][C] - for layers without a residual input this saves the
 * read of y (one of the 3-4 streamed tensors); 3 `y` points at the byte mask of eeseg_bn_apply_relu_mask and
 * `ldy` is its row length in bytes (same in eeseg_bn_bwd_apply).  sums_copy (optional): a second [2][C] buffer that
 * receives the same sums - under SyncBN one copy stays local (dbeta, dgamma in the gradient arena), the other goes into
 * the collective. */
int eeseg_bn_bwd_reduce(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                        const float* mean_invstd, const float* scale_shift, int64_t rows, int C, int relu,
                        float* sums, float* sums_copy, int dtype, void* workspace, int64_t workspace_bytes, void* stream);
/* backward, step 2: dx = gamma*invstd*(g - sums0/count - xhat*sums1/count);
 * dres (optional) = g.  dgamma = sums1, dbeta = sums0 (taken by the host). */
int eeseg_bn_bwd_apply(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                       const float* mean_invstd, const float* gamma, const float* sums, double count,
                       void* dx, int lddx, void* dres, int lddres, int64_t rows, int C, int relu,
                       const float* scale_shift, int dtype, void* stream);
/* backward, steps 1 + 2 in ONE launch (round 4; torch BatchNorm2d backward reached through loss.mean().backward(),
 * train_funcs.py:26) for tensors small enough that a one-block-per-CU grid keeps its rows in registers across a grid-wide
 * barrier - the per-GPU shards of a data-parallel run (4-8 images of 65 x 65): dy and x are read once instead of twice and
 * two launches disappear.  Same arguments and results as eeseg_bn_bwd_reduce followed by eeseg_bn_bwd_apply (sums[2][C] =
 * (dbeta, dgamma) are written too); the cross-block summation order differs from the two-step form (fixed, run-to-run
 * identical).  eeseg_bn_bwd_coop_ok(rows, C, dtype) says whether the shape fits (C a multiple of 64 bf16 / 32 fp32 channels,
 * <= 18 rows per thread); workspace >= eeseg_bn_bwd_coop_workspace() bytes; barrier_state: 8224 32-bit words, 128-byte
 * aligned, ZEROED once by the caller and then owned by the calls on one stream (the kernel leaves it zeroed; word 8192 is a
 * sticky give-up flag: non-zero = some launch found its grid not co-resident and produced garbage instead of hanging). */
int eeseg_bn_bwd_coop_ok(int64_t rows, int C, int dtype);
int64_t eeseg_bn_bwd_coop_workspace(void);
int eeseg_bn_bwd_coop(const void* dy, int lddy, const void* y, int ldy, const void* x, int ldx,
                      const float* mean_invstd, const float* gamma, const float* scale_shift, double count,
                      float* sums, float* sums_copy, void* dx, int lddx, void* dres, int lddres, int64_t rows,
                      int C, int relu, int dtype, void* workspace, int64_t workspace_bytes, void* barrier_state,
                      void* stream);
/* frozen-BN / plain backward of y = act(x*scale+shift): dx = g*scale, dres = g */
int eeseg_scale_act_bwd(const void* dy, int lddy, const void* y, int ldy, const float* scale,
                        void* dx, int lddx, void* dres, int lddres, int64_t rows, int C, int relu, int dtype,
                        void* stream);

/* --------------------------------------------------- pooling / misc ------- */
/* 3x3 stride-2 pad-1 max pool (ResNet stem, from_deepv3_new.py:77-79); backward
 * routes each window's gradient to its FIRST maximum (torch CPU tie rule).  `y` = the forward output (may be
 * NULL: the window maxima are then recomputed, ~4x the loads). */
int eeseg_maxpool3x3s2(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int dtype, void* stream);
int eeseg_maxpool3x3s2_bwd(const void* x, const void* y, const void* dy, void* dx, int N, int H, int W, int C, int Ho,
                           int Wo, int dtype, void* stream);
/* per-image column sum: y[n][c] = scale * sum_hw x[n][hw][c] (ASPPPooling's
 * AdaptiveAvgPool2d(1) with scale = 1/HW; also the backward of the broadcast) */
int eeseg_sum_hw(const void* x, int ldx, void* y, int N, int HW, int C, float scale, int dtype, void* workspace,
                 int64_t workspace_bytes /* >= 16*N*C*4 enables the split reduction */, void* stream);
/* y[n][hw][c] (+)= scale * x[n][c] into a channel slice of row stride ldy
 * (ASPPPooling's bilinear upsample from 1x1 is a broadcast; with scale = 1/HW
 * and accumulate = 1 it is the backward of the average pool) */
int eeseg_broadcast_hw(const void* x, void* y, int ldy, int N, int HW, int C, float scale, int accumulate,
                       int dtype, void* stream);
/* Dropout(p) with a counter-based hash RNG: keep iff hash(seed', index) >= p with
 * seed' = seed + K*step_dev[0] (step_dev: optional device step counter, so a captured
 * HIP graph draws a fresh mask every replay); kept values are scaled by 1/(1-p).
 * index = index_offset + element position: a data-parallel rank passes rank * n, so the ranks' masks are the slices of
 * the mask ONE process would draw for the whole (rank-major) batch.  Backward = the same call on dy. */
int eeseg_dropout(const void* x, void* y, int64_t n, float p, uint64_t seed, const int64_t* step_dev,
                  int64_t index_offset, int dtype, void* stream);
int eeseg_cast(const void* x, int in_dtype, void* y, int out_dtype, int64_t n, void* stream);
/* y[i] += x[i] (gradient accumulation at residual joins) */
int eeseg_add_inplace(void* y, const void* x, int64_t n, int dtype, void* stream);
/* strided row copy (16-byte multiples): dst[r][0:row_bytes] = src[r][0:row_bytes]; builds the tap-concatenated weights of
 * a generalised-tap conv (eeseg_conv_args.taps) from the per-conv packed weights */
int eeseg_copy2d(const void* src, int64_t src_ld_bytes, void* dst, int64_t dst_ld_bytes, int64_t rows, int64_t row_bytes,
                 void* stream);
/* per-column sum of a [rows][C] matrix -> out[C] fp32 (bias gradient) */
int eeseg_colsum(const void* x, int ldx, int64_t rows, int C, float* out, int dtype, void* workspace,
                 int64_t workspace_bytes, void* stream);

/* ------------------------------------------------ upsample / losses -------
 * logits_lr: [N,h,w,ldc] fp32 NHWC low-resolution logits of ONE exit (C <= 32
 * valid channels, ldc >= C). */
/* out: [N,C,H,W] fp32 NCHW.  Replaces F.interpolate(bilinear,
 * align_corners=False) at from_deepv3_new.py:149,152. */
int eeseg_upsample_bilinear_nchw(const float* logits_lr, int ldc, float* out, int N, int C, int h, int w,
                                 int H, int W, void* stream);
int eeseg_upsample_bilinear_nchw_bwd(const float* dout, float* dlogits_lr, int ldc, int N, int C, int h, int w,
                                     int H, int W, void* stream);
/* Fused upsample + per-pixel cross entropy for one exit
 * (my_pixelwise_xentropy.py:11-14,36-38: CrossEntropyLoss(mean, ignore_index)).
 * accum (double[2], device): [0] += sum of -log p[target] over valid pixels,
 * [1] += #valid.  Labels outside [0,C) are treated as ignored. */
int eeseg_upsample_ce_fwd(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w,
                          int H, int W, int64_t ignore_index, double* accum /*[2]*/, void* stream);
/* dlogits_lr += gscale*gscale_dev[0]/accum[1] * d(sum CE)/d(logits_lr).  The valid
 * count and the optional upstream gradient scalar (gscale_dev, may be NULL) are
 * read from device memory: no host sync.  The caller zeroes dlogits_lr. */
int eeseg_upsample_ce_bwd(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w,
                          int H, int W, int64_t ignore_index, const double* accum, float gscale,
                          const float* gscale_dev, float* dlogits_lr, void* stream);
/* Fused upsample + argmax + TP/FP/FN (seg_metrics.py:13-28 + compute_mIoU.py:16-27):
 * counts[3][C] int32 (+= per call, exact); void pixels (label outside [0,C))
 * count as FP of the predicted class (B-9).  pred (optional) receives the argmax
 * mask [N,H,W] int64; target may be NULL when only pred is wanted. */
int eeseg_argmax_confusion(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w,
                           int H, int W, int32_t* counts, int64_t* pred, void* stream);
/* Region / focal losses (branchy_seg_losses.py:40-131) from low-res logits in one pass per exit: per image and class
 * S = sum p_c, I = sum p_c [t = c], T = #[t = c] over the full-resolution pixels (p = softmax of the upsampled
 * logits) -> sums [N][3][32] double (+=); extra [N][2] double (+=): [0] = pixels labelled outside [0,C),
 * [1] = focal sum  sum -alpha_t (1 - p_t)^gamma log p_t  (gamma < 0: skipped; alpha may be NULL).  Dice / Jaccard are
 * closed forms of (S, I, T) evaluated by the caller.  The backward takes dL/dS = gS [N][32], dL/dI = gI [N][32]
 * and dL/dF = gF[0] as DEVICE pointers (any may be NULL) and accumulates (+=) into dlogits_lr.
 * alpha_batch_sum = 1: the pixel's focal weight is sum_i alpha[t[i][y][x]] over ALL N images instead of the alpha of its
 * own label - what the reference's broadcast of the [B,H,W] loss map against alpha[targets] of shape [B,1,H,W]
 * (branchy_seg_losses.py:126-129) amounts to once the extra axis is summed. */
int eeseg_class_sums_fwd(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w, int H, int W,
                         float gamma, const float* alpha, int alpha_batch_sum, double* sums, double* extra, void* stream);
int eeseg_class_sums_bwd(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w, int H, int W,
                         const float* gS, const float* gI, const float* gF, float gamma, const float* alpha,
                         int alpha_batch_sum, float* dlogits_lr, void* stream);
/* FocalLoss with reduction = 'none' (branchy_seg_losses.py:113-131 through BrSegLoss.forward :24-38): the per-pixel map
 * -(1 - p_t)^gamma log p_t [N,H,W] fp32 of one exit from its low-res logits, and its backward (dlogits_lr += for an upstream
 * gradient dmap of the map's shape).  alpha_mode 0: no alpha; 1: times alpha[t] of the pixel's own label; 2: the reference's
 * broadcast against alpha[targets] of shape [N,1,H,W] - out / dmap are [N,N,H,W] with out[i][j] = map[j] * alpha[t[i]].
 * *void_count (device int32, +=) counts pixels labelled outside [0,C) (the reference's gather fails on them; they get 0). */
int eeseg_focal_map_fwd(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w, int H, int W,
                        float gamma, const float* alpha, int alpha_mode, float* out, int32_t* void_count, void* stream);
int eeseg_focal_map_bwd(const float* logits_lr, int ldc, const int64_t* target, int N, int C, int h, int w, int H, int W,
                        float gamma, const float* alpha, int alpha_mode, const float* dmap, float* dlogits_lr, void* stream);
/* Fused upsample + argmax of TWO exits -> per-image contingency table hist[N][C][C] int32 (+= per call):
 * hist[n][a][b] = pixels where exit A predicts a and exit B predicts b.  The similarity gates between consecutive
 * exits (MSE / NMI / variation of information of the label maps: sim_metrics.py:41-120, eval_br_sim.py:41-48,
 * ee_dnn_op.py:86) are functions of this table, so no label map leaves the device. */
int eeseg_argmax_pair_hist(const float* logits_a, const float* logits_b, int ldc, int N, int C, int h, int w, int H, int W,
                           int32_t* hist, void* stream);
/* ---- batched progressive early-exit inference (ee_dnn_op_ne.py:51-108 made real and batched) -------
 * State on the device: n_active (int32[1]) images are still in flight, occupying batch slots 0..n_active-1;
 * order[slot] = index of that image in the caller's batch.
 * eeseg_entropy_gate_active: eeseg_entropy_gate restricted to the active slots; flag = (entropy < tau) == less_than.
 * eeseg_argmax_exit: upsample + argmax of the slots that leave (flags != NULL: flags[slot] != 0; NULL: all active
 *   slots) written to pred_out[order[slot]] ([B,H,W] int64).
 * eeseg_exit_select: for every active slot with flags[slot]: exit_idx[order[slot]] = code; the others are compacted
 *   to the front: order[k] = their image, src_slot[k] = the slot their features sit in; n_active = their count.
 * eeseg_gather_images: x_out[k] = x[src_slot[k]] for k < *n_active (bytes_per_image each, 16-byte multiples). */
int eeseg_entropy_gate_active(const float* logits_lr, int ldc, int N, int C, int h, int w, int H, int W, int pool,
                              int pool_size, float tau, int less_than, const int32_t* n_active, float* entropy_out,
                              int32_t* exit_flag, void* workspace, int64_t workspace_bytes, void* stream);
int eeseg_argmax_exit(const float* logits_lr, int ldc, int N, int C, int h, int w, int H, int W, const int32_t* flags,
                      const int32_t* order, const int32_t* n_active, int64_t* pred_out, void* stream);
int eeseg_exit_select(const int32_t* flags, int N, int code, int32_t* n_active, int32_t* order, int32_t* src_slot,
                      int32_t* exit_idx, void* stream);
int eeseg_gather_images(const void* x, void* x_out, const int32_t* src_slot, const int32_t* n_active, int N,
                        int64_t bytes_per_image, void* stream);
/* SSIM of two integer label maps [N,H,W] int64 (sim_metrics.py:15-37: skimage.metrics.structural_similarity with its
 * defaults - 7x7 uniform window, sample covariance, K1 0.01, K2 0.03, mean over the map cropped by 3 px per side, float64).
 * ssim_out[N] double, overwritten. */
int eeseg_ssim_labels(const int64_t* labels_a, const int64_t* labels_b, int N, int H, int W, double data_range,
                      double* ssim_out, void* stream);
/* Fused upsample + softmax + normalised entropy (+ s x s max/min block pool) +
 * mean per image (eval_br_ent.py:19-36).  entropy_out[N] fp32; exit_flag[N]
 * int32 = (entropy < tau) stays on device (eval_br_ent.py:60, ee_dnn_op_ne.py:81).
 * pool: 0 none, 1 max, 2 min. */
int64_t eeseg_entropy_gate_workspace(int N, int H, int W);
int eeseg_entropy_gate(const float* logits_lr, int ldc, int N, int C, int h, int w, int H, int W, int pool,
                       int pool_size, float tau, float* entropy_out, int32_t* exit_flag, void* workspace,
                       int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------ lovasz ------
 * Multi-class Lovasz on RAW scores for one exit (branchy_seg_losses.py:154 ->
 * lovaszsoftmax.py:172-200): all N images ranked jointly (per_image=False; per_image=True, lovaszsoftmax.py:165-166, is
 * one call per image).  scores: [N,C,HW] fp32 (full resolution, NCHW), target [N,HW] int64 (labels outside
 * [0,C) or == ignore_index are void).  loss_out[0] = mean of the per-class losses over the classes c with bit c of
 * class_mask set and, when present_only = 1, at least one labelled pixel: classes='present' = (all ones, 1), 'all' =
 * (all ones, 0), a list = (its bits, 0) (lovaszsoftmax.py:185-188; an absent class that is counted contributes its
 * largest |score|).  dscores (optional, [N,C,HW]) = gscale*gscale_dev[0] * d(loss)/d(scores).
 * Ties between equal errors are ordered by pixel index (torch.sort leaves them
 * unspecified); the loss value is tie-invariant. */
int64_t eeseg_lovasz_workspace(int64_t P, int C);
/* Ranking a SUBSET of the label classes (round 4, the class-sharded data-parallel form): n_label_classes (0 = C): labels in
 * [0, n_label_classes) are valid pixels; class_ids (HOST int32[C], NULL = identity): score plane c ranks label class
 * class_ids[c]; norm_classes_dev (NULL = none): DEVICE int32, the number of classes the mean runs over - the call then
 * returns sum_c loss_c / norm over ITS classes.  The per-class losses are independent given all pixels, so rank r ranks
 * classes r, r + world, ... over the gathered batch and the ranks' shares add up to the reference's loss - exact, and no
 * rank sorts more than its share.  eeseg_label_hist: counts[c] = pixels labelled c (which classes are 'present'). */
int eeseg_label_hist(const int64_t* target, int64_t n, int C, int64_t ignore_index, int32_t* counts, void* stream);
int eeseg_lovasz(const float* scores, const int64_t* target, int N, int C, int HW, int64_t ignore_index,
                 float* loss_out, float* dscores, float gscale, const float* gscale_dev, uint64_t class_mask,
                 int present_only, int n_label_classes, const int32_t* class_ids, const int32_t* norm_classes_dev,
                 void* workspace, int64_t workspace_bytes, void* stream);

/* --------------------------------------------------------------- SGD ------
 * torch.optim.SGD(momentum, weight_decay) step, multi-tensor
 * (deepv3_funcs.py:99; train_funcs.py:27).  ptrs: device array of
 * {param, grad, momentum_buf} triples; sizes: device array of element counts;
 * lrs: device array of per-tensor learning rates.  first_step=1 initialises
 * the momentum buffer with the gradient (torch semantics). */
int eeseg_sgd_step(void* const* param_grad_buf /*[n][3]*/, const int64_t* sizes, const float* lrs, int n,
                   float momentum, float weight_decay, float grad_scale, int first_step, void* stream);

/* ------------------------------------------------ input pipeline ------- */
/* The torchvision chain of get_seg_datasets.py:49-86 on decoded uint8 pixels, bit-exact with Pillow:
 *   image : Resize (PIL bilinear, antialiased; 8-bit intermediate between the passes) -> CenterCrop -> ToTensor ->
 *           Normalize(mean, std) -> out [C][Dh][Dw] fp32
 *   target: Resize (PIL forces NEAREST on palette images) -> CenterCrop -> lut[value] -> out [Dh][Dw] int64
 *           (lut = the reference's ToTensor()*255 -> long -> 255 -> void chain applied to 0..255 on the host).
 * The two host helpers reproduce Pillow's tables (Resample.c precompute_coeffs/normalize_coeffs_8bpc; Geometry.c
 * ImagingScaleAffine nearest coordinates, accumulated in double); they do not touch the GPU.
 * eeseg_pil_bilinear_coeffs returns the tap count ksize (call with NULL tables to size them; kk rows have
 * `ksize_capacity` entries, which is also the `*ksize` the device call takes). */
int eeseg_pil_bilinear_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int ksize_capacity);
int eeseg_pil_nearest_index(int in_size, int out_size, int32_t* idx);
int eeseg_preprocess_image_u8(const uint8_t* src, int H, int W, int C, int Hr, int Wr, const int32_t* hbounds,
                              const int32_t* hkk, int hksize, const int32_t* vbounds, const int32_t* vkk, int vksize,
                              int crop_top, int crop_left, int Dh, int Dw, const float* mean, const float* stdv,
                              uint8_t* tmp /* [H][Wr][C] */, float* out, void* stream);
int eeseg_preprocess_label_u8(const uint8_t* src, int H, int W, const int32_t* yidx, const int32_t* xidx, int Hr, int Wr,
                              int crop_top, int crop_left, int Dh, int Dw, const int64_t* lut, int64_t* out, void* stream);

/* ------------------------------------- data-parallel collectives -------
 * The reference has no collective (nn.DataParallel is commented out, train_funcs.py:72-74): these are the exchange
 * steps of the build's own data-parallel training path - gradient buckets, SyncBN (sum x, sum x^2) pairs, the global
 * cross-entropy valid-pixel count, the exact-Lovasz all-gather, the per-exit mIoU counters (eval_mIoU.py:15-40 under DP).
 * RCCL is called DIRECTLY and enqueues ONE kernel on the stream passed in - eagerly or inside a HIP-graph capture
 * alike; there is no hidden stream, no completion event and no polling thread (DESIGN.md section 7 explains why
 * torch.distributed's process group is not used for the data path).  State lives in the communicator handle only.
 * librccl.so.1 is bound at run time (the copy the host process has already mapped, else /opt/rocm/lib).
 *   rendezvous: rank 0 calls eeseg_comm_unique_id and hands the 128 bytes to every rank out of band (the Python host
 *   uses torch.distributed's store / gloo); then EVERY rank calls eeseg_comm_create (collective, blocking) with its
 *   device current (hipSetDevice).  Buffers are device memory of that device; `count` in elements, `bytes` in bytes.
 *   In-place only.  All ranks must issue the same collectives in the same order per communicator. */
enum { EESEG_COMM_F32 = 0, EESEG_COMM_BF16 = 1, EESEG_COMM_F64 = 2, EESEG_COMM_I32 = 3, EESEG_COMM_I64 = 4 };
enum { EESEG_COMM_SUM = 0, EESEG_COMM_AVG = 1, EESEG_COMM_MAX = 2 };
int eeseg_comm_available(int* rccl_version /* optional out */);       /* 0 when librccl could be bound */
int eeseg_comm_unique_id(void* id128);
int eeseg_comm_create(const void* id128, int world, int rank, void** comm_out);
int eeseg_comm_destroy(void* comm);          /* the device must be idle w.r.t. this communicator (and graphs holding its kernels gone) */
int eeseg_comm_info(void* comm, int* world, int* rank, int* device);
int eeseg_comm_check(void* comm);            /* asynchronous RCCL error of the communicator, if any (does not touch the GPU) */
int eeseg_comm_all_reduce(void* comm, void* buf, int64_t count, int dtype, int op, void* stream);
int eeseg_comm_all_gather(void* comm, const void* send, void* recv /* [world][bytes_per_rank] */, int64_t bytes_per_rank,
                          void* stream);
int eeseg_comm_broadcast(void* comm, void* buf, int64_t bytes, int root, void* stream);
/* recv[count_per_rank] = rank r's slice of the element-wise reduction of every rank's send[world * count_per_rank] (the
 * class-sharded exact Lovasz hands each rank the summed low-resolution logit gradients of ITS images) */
int eeseg_comm_reduce_scatter(void* comm, const void* send, void* recv, int64_t count_per_rank, int dtype, int op, void* stream);

#ifdef __cplusplus
}
#endif
#endif
