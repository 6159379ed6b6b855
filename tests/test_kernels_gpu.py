"""GPU parity tests of every libeeseg kernel (through the C ABI) against the CPU
oracle: torch CPU fp32 ops (what the reference reaches through torchvision) and
the golden vectors generated from the importable reference files.

Tolerances: fp32 mode -> 1e-4 relative to the result scale (summation order only);
bf16 mode -> inputs are rounded to bf16 first, 2^-7 relative (bf16 output rounding).
"""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from ee_semantic_segmentation_amd import kernels as K

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "losses_seed*.npz")))


def tol(dtype):
    return 1e-4 if dtype == torch.float32 else 1.6e-2


@pytest.fixture(autouse=True)
def _keep_256_tile_tests_on_the_256_tile_kernel(request):
    """Round 4 sends SMALL layers (<= CUs / 2 tiles of 256 x 256) to the small-M form of the 128x256 kernel
    (EESEG_OPT_CONV_SMALL_M, tests/test_round4_kernels_gpu.py).  The tests of this file that are ABOUT the 256-tile kernel use
    small shapes to stay CPU-checkable, so they switch that dispatch off for their duration."""
    if "256_tile" not in request.node.name or not torch.cuda.is_available():
        yield
        return
    from ee_semantic_segmentation_amd._lib import lib
    lib().eeseg_set_option(21, 0)
    try:
        yield
    finally:
        lib().eeseg_set_option(21, 2)


def rnd(dtype, *shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    t = torch.randn(*shape, generator=g) * scale
    return t.to(dtype).float()     # value exactly representable in `dtype`


def close(got, want, rel, what=""):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-12
    err = (got - want).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e} > {rel})"


def nhwc(t):   # NCHW cpu -> NHWC
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 9, 7, 64, 128, 1, 1, 0, 1),
    (3, 17, 13, 128, 64, 1, 1, 0, 1),
    (2, 13, 11, 64, 128, 3, 1, 1, 1),
    (2, 13, 11, 64, 256, 3, 1, 2, 2),
    (1, 17, 17, 128, 256, 3, 1, 12, 12),     # atrous: whole taps fall outside (tap skipping)
    (2, 15, 14, 64, 128, 3, 2, 1, 1),        # strided
    (2, 12, 12, 128, 256, 1, 2, 0, 1),       # strided 1x1 (downsample)
    (2, 10, 9, 256, 32, 1, 1, 0, 1),         # narrow classifier-like output
    (1, 40, 40, 192, 64, 1, 1, 0, 1),        # stem-GEMM-like K
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case, dtype):
    N, H, W, Cin, Cout, k, s, p, d = case
    if dtype == torch.bfloat16 and Cout % 64:
        pytest.skip("bf16 data-gradient needs K = Cout % 64 == 0; the 21/19-class classifier runs in fp32")
    x = rnd(dtype, N, Cin, H, W, seed=1).requires_grad_(True)
    w = rnd(dtype, Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p, dilation=d)
    gy = rnd(dtype, *y.shape, seed=3)
    y.backward(gy)

    xd = nhwc(x.detach()).to(DEV, dtype)
    wf, wb = K.pack_weight(w.detach().to(DEV), dtype)
    close(wf.permute(0, 3, 1, 2), w, 1e-7, "pack fwd")
    close(wb.permute(3, 0, 1, 2), w, 1e-7, "pack bwd")
    yd, part = K.conv_fwd(xd, wf, s, p, d, want_stats=True)
    close(nchw(yd), y, tol(dtype), "conv fwd")
    # BN partial sums from the epilogue (computed on the stored, rounded output)
    ys = yd.float().reshape(-1, Cout)
    sums = K.reduce_partials(part)
    close(sums[0], ys.sum(0), 1e-4, "stats sum")
    close(sums[1], (ys * ys).sum(0), 1e-4, "stats sumsq")

    gyd = nhwc(gy).to(DEV, dtype)
    dx = K.conv_dgrad(gyd, wb, (H, W), s, p, d)
    close(nchw(dx), x.grad, tol(dtype), "conv dgrad")
    dw = K.conv_wgrad(xd, gyd, k, k, s, p, d)
    close(dw.permute(0, 3, 1, 2), w.grad, 2e-4 if dtype == torch.float32 else 4e-3, "conv wgrad")
    # accumulation variants
    dx2 = K.conv_dgrad(gyd, wb, (H, W), s, p, d, accumulate_into=dx.clone())
    close(nchw(dx2), 2 * x.grad, 2 * tol(dtype), "conv dgrad accumulate")
    dw2 = K.conv_wgrad(xd, gyd, k, k, s, p, d, out=dw.clone(), accumulate=True)
    close(dw2.permute(0, 3, 1, 2), 2 * w.grad, 2e-4 if dtype == torch.float32 else 4e-3, "conv wgrad accumulate")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_conv_epilogue_and_slice_output(dtype):
    N, H, W, Cin, Cout = 2, 11, 9, 64, 128
    x = rnd(dtype, N, Cin, H, W, seed=4)
    w = rnd(dtype, Cout, Cin, 3, 3, seed=5, scale=0.05)
    res = rnd(dtype, N, Cout, H, W, seed=6)
    scale = torch.rand(Cout) + 0.5
    shift = torch.randn(Cout)
    want = F.relu(F.conv2d(x, w, padding=1) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    wf, _ = K.pack_weight(w.to(DEV), dtype, want_bwd=False)
    wide = torch.zeros(N, H, W, 3 * Cout, dtype=dtype, device=DEV)
    out = wide[..., Cout:2 * Cout]
    K.conv_fwd(nhwc(x).to(DEV, dtype), wf, 1, 1, 1, scale=scale.to(DEV), shift=shift.to(DEV),
               residual=nhwc(res).to(DEV, dtype), relu=True, out=out)
    close(nchw(out), want, tol(dtype) * 2, "fused epilogue")
    assert wide[..., :Cout].abs().max().item() == 0 and wide[..., 2 * Cout:].abs().max().item() == 0


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_stem_im2col_gemm(dtype):
    N, H, W = 2, 33, 29
    x = torch.randn(N, 3, H, W, generator=torch.Generator().manual_seed(7))
    w = rnd(dtype, 64, 3, 7, 7, seed=8, scale=0.1)
    want = F.conv2d(x.to(dtype).float(), w, stride=2, padding=3)
    col = K.im2col_nchw(x.to(DEV), 7, 7, 2, 3, 192, dtype)
    wk = w.permute(0, 2, 3, 1).reshape(64, 147).contiguous()
    wm = K.pack_matrix(wk.to(DEV), 64, 192, dtype).view(64, 1, 1, 192)
    y, _ = K.conv_fwd(col, wm)
    close(nchw(y), want, tol(dtype), "stem conv")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 3, 41, 300, 7, 2, 3), (1, 3, 9, 131, 7, 2, 3), (2, 4, 20, 70, 3, 1, 1)],
                         ids=["three_wo_tiles", "ragged", "3x3_s1"])
def test_im2col_matches_unfold(shape, dtype):
    """The LDS-tiled im2col (64 output pixels of one row per block) against torch's unfold, bit exact: several pixel
    tiles per row, a ragged last tile, borders on all four sides; K order (tap, ci), zero padding up to Kpad."""
    N, Cc, H, W, k, stride, pad = shape
    x = torch.randn(N, Cc, H, W, generator=torch.Generator().manual_seed(3))
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    Kd = k * k * Cc
    Kpad = (Kd + 63) // 64 * 64
    col = K.im2col_nchw(x.to(DEV), k, k, stride, pad, Kpad, dtype)
    want = F.unfold(x, k, padding=pad, stride=stride)                    # [N, C*k*k, L], row index (ci, r, s)
    want = want.view(N, Cc, k * k, Ho * Wo).permute(0, 3, 2, 1).reshape(N * Ho * Wo, Kd).to(dtype)
    got = col.reshape(N * Ho * Wo, Kpad).cpu()
    assert torch.equal(got[:, :Kd], want)
    assert torch.count_nonzero(got[:, Kd:]) == 0


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("relu,use_res", [(True, False), (True, True), (False, False)])
def test_batchnorm_train_fwd_bwd(dtype, relu, use_res):
    N, H, W, Cc = 3, 23, 19, 128
    x = rnd(dtype, N, Cc, H, W, seed=9, scale=2.0).add_(0.3).to(dtype).float().requires_grad_(True)
    res = rnd(dtype, N, Cc, H, W, seed=10).requires_grad_(True)
    gamma = (torch.rand(Cc) + 0.5).requires_grad_(True)
    beta = torch.randn(Cc).requires_grad_(True)
    rm, rv = torch.zeros(Cc), torch.ones(Cc)
    y = F.batch_norm(x, rm, rv, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    if use_res:
        y = y + res
    if relu:
        y = F.relu(y)
    gy = rnd(dtype, *y.shape, seed=11)
    y.backward(gy)

    xd = nhwc(x.detach()).to(DEV, dtype)
    sums = K.channel_stats(xd)
    rmd, rvd = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
    cnt = N * H * W
    mi, ss = K.bn_finalize(sums, cnt, gamma.detach().to(DEV), beta.detach().to(DEV), 1e-5, 0.1, rmd, rvd)
    close(rmd, rm, 1e-5, "running mean")
    close(rvd, rv, 1e-5, "running var")
    resd = nhwc(res.detach()).to(DEV, dtype) if use_res else None
    yd = K.bn_apply(xd, ss, residual=resd, relu=relu)
    close(nchw(yd), y, tol(dtype), "bn fwd")
    gyd = nhwc(gy).to(DEV, dtype)
    bs = K.bn_bwd_reduce(gyd, yd if relu else None, xd, mi, relu)
    close(bs[0], beta.grad, 2e-4 if dtype == torch.float32 else 2e-2, "dbeta")
    close(bs[1], gamma.grad, 2e-4 if dtype == torch.float32 else 2e-2, "dgamma")
    dx, dres = K.bn_bwd_apply(gyd, yd if relu else None, xd, mi, gamma.detach().to(DEV), bs, cnt, relu,
                              want_dres=use_res)
    close(nchw(dx), x.grad, 3e-4 if dtype == torch.float32 else 3e-2, "bn dx")
    if use_res:
        close(nchw(dres), res.grad, tol(dtype), "bn dres")
        # byte mask written by the forward (1 bit per element) instead of the stored output: identical results
        yd3, mask = K.bn_apply(xd, ss, residual=resd, relu=True, want_mask=True)
        assert torch.equal(yd3, yd) and mask.dtype == torch.uint8 and mask.shape == (cnt, Cc // (16 // xd.element_size()))
        bs3 = K.bn_bwd_reduce(gyd, mask, xd, mi, relu)
        dx3, dres3 = K.bn_bwd_apply(gyd, mask, xd, mi, gamma.detach().to(DEV), bs3, cnt, relu, want_dres=True)
        assert torch.equal(bs3, bs) and torch.equal(dx3, dx) and torch.equal(dres3, dres)
    elif relu:      # mask recomputed from x*scale+shift instead of reading y: identical results
        bs2 = K.bn_bwd_reduce(gyd, None, xd, mi, relu, scale_shift=ss)
        dx2, _ = K.bn_bwd_apply(gyd, None, xd, mi, gamma.detach().to(DEV), bs2, cnt, relu, scale_shift=ss)
        assert torch.equal(bs2, bs) and torch.equal(dx2, dx)


def test_bn_eval_and_frozen_bwd():
    Cc = 64
    g, b, rm, rv = torch.rand(Cc) + 0.5, torch.randn(Cc), torch.randn(Cc), torch.rand(Cc) + 0.5
    ss = K.bn_eval_scale_shift(g.to(DEV), b.to(DEV), rm.to(DEV), rv.to(DEV), 1e-5)
    x = torch.randn(2, Cc, 5, 6, requires_grad=True)
    y = F.relu(F.batch_norm(x, rm, rv, g, b, training=False, eps=1e-5))
    gy = torch.randn_like(y)
    y.backward(gy)
    yd = K.bn_apply(nhwc(x.detach()).to(DEV), ss, relu=True)
    close(nchw(yd), y, 1e-5, "bn eval")
    dx, _ = K.scale_act_bwd(nhwc(gy).to(DEV), yd, ss[0].contiguous(), True)
    close(nchw(dx), x.grad, 1e-5, "frozen bn bwd")


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_maxpool_fwd_bwd_with_ties(dtype):
    N, Cc, H, W = 2, 64, 17, 21
    x = F.relu(rnd(dtype, N, Cc, H, W, seed=12)).requires_grad_(True)    # many exact ties at 0
    y = F.max_pool2d(x, 3, 2, 1)
    gy = rnd(dtype, *y.shape, seed=13)
    y.backward(gy)
    xd = nhwc(x.detach()).to(DEV, dtype)
    yd = K.maxpool3x3s2(xd)
    close(nchw(yd), y, 0.0 + 1e-7, "maxpool fwd")
    dx = K.maxpool3x3s2_bwd(xd, nhwc(gy).to(DEV, dtype))
    close(nchw(dx), x.grad, tol(dtype), "maxpool bwd")
    dx2 = K.maxpool3x3s2_bwd(xd, nhwc(gy).to(DEV, dtype), yd)          # with the forward output: same routing
    assert torch.equal(dx2, dx)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_gap_broadcast_dropout_cast_add(dtype):
    N, H, W, Cc = 3, 9, 11, 256
    wide = rnd(dtype, N, H, W, 2 * Cc, seed=14).to(DEV, dtype)
    xs = wide[..., Cc:]
    got = K.sum_hw(xs, 1.0 / (H * W))
    close(got, xs.float().mean(dim=(1, 2)), tol(dtype), "gap")
    out = torch.zeros(N, H, W, 3 * Cc, dtype=dtype, device=DEV)
    K.broadcast_hw(got, out[..., Cc:2 * Cc])
    close(out[..., Cc:2 * Cc], got.float().view(N, 1, 1, Cc).expand(N, H, W, Cc), 1e-6, "broadcast")
    K.broadcast_hw(got, out[..., Cc:2 * Cc], scale=0.5, accumulate=True)
    close(out[..., Cc:2 * Cc], 1.5 * got.float().view(N, 1, 1, Cc).expand(N, H, W, Cc), tol(dtype), "broadcast acc")
    assert out[..., :Cc].abs().max().item() == 0
    # dropout
    x = torch.ones(4, 64, 64, 16, dtype=dtype, device=DEV)
    d1, d2, d3 = K.dropout(x, 0.5, 123), K.dropout(x, 0.5, 123), K.dropout(x, 0.5, 124)
    assert torch.equal(d1, d2) and not torch.equal(d1, d3)
    keep = (d1 != 0).float().mean().item()
    assert abs(keep - 0.5) < 0.01 and set(d1.float().unique().tolist()) == {0.0, 2.0}
    # data parallel: rank r passes index_offset = r * n and draws the r-th slice of the whole batch's mask
    half = x[:2].contiguous()
    lo, hi = K.dropout(half, 0.5, 123), K.dropout(half, 0.5, 123, index_offset=half.numel())
    assert torch.equal(torch.cat([lo, hi]), d1)
    # cast / add / colsum
    f = torch.randn(1000, 64, device=DEV)
    close(K.cast(f, torch.bfloat16), f.bfloat16(), 1e-6, "cast")
    a, b = wide.clone(), wide.clone()
    K.add_inplace(a, b)
    close(a, 2 * wide.float(), tol(dtype), "add")
    close(K.colsum(xs), xs.float().reshape(-1, Cc).sum(0), 1e-3 if dtype == torch.float32 else 1e-2, "colsum")


def _lr_from_nchw(y, ldc=32):
    """[N,C,H,W] cpu -> device NHWC [N,H,W,ldc] fp32 zero padded."""
    N, Cc, H, W = y.shape
    lr = torch.zeros(N, H, W, ldc)
    lr[..., :Cc] = y.permute(0, 2, 3, 1)
    return lr.to(DEV)


@pytest.mark.parametrize("hw,HW", [((9, 7), (65, 50)), ((17, 17), (129, 129)), ((5, 8), (5, 8)), ((12, 10), (7, 9))])
def test_upsample_fwd_bwd(hw, HW):
    h, w = hw
    H, W = HW
    N, Cc = 2, 21
    y = torch.randn(N, Cc, h, w, generator=torch.Generator().manual_seed(15)).requires_grad_(True)
    up = F.interpolate(y, size=(H, W), mode="bilinear", align_corners=False)
    g = torch.randn(up.shape, generator=torch.Generator().manual_seed(16))
    up.backward(g)
    lr = _lr_from_nchw(y.detach())
    got = K.upsample_bilinear_nchw(lr, Cc, H, W)
    close(got, up, 2e-6, "upsample")
    dlr = K.upsample_bilinear_nchw_bwd(g.to(DEV), h, w, 32)
    close(dlr[..., :Cc].permute(0, 3, 1, 2), y.grad, 1e-5, "upsample bwd")


@pytest.mark.parametrize("hw,HW,Cc", [((9, 7), (65, 50), 21), ((17, 17), (129, 129), 19), ((6, 5), (6, 5), 5)])
def test_upsample_ce_fwd_bwd(hw, HW, Cc):
    h, w = hw
    H, W = HW
    N = 3
    gen = torch.Generator().manual_seed(17)
    y = (torch.randn(N, Cc, h, w, generator=gen) * 2).requires_grad_(True)
    t = torch.randint(0, Cc + 1, (N, H, W), generator=gen)
    up = F.interpolate(y, size=(H, W), mode="bilinear", align_corners=False)
    loss = F.cross_entropy(up, t, ignore_index=Cc)
    (loss * 0.7).backward()
    lr = _lr_from_nchw(y.detach())
    acc = torch.zeros(2, dtype=torch.float64, device=DEV)
    td = t.to(DEV)
    K.upsample_ce_fwd(lr, Cc, td, H, W, Cc, acc)
    a = acc.cpu()
    assert a[1].item() == (t != Cc).sum().item()
    assert abs(a[0].item() / a[1].item() - loss.item()) < 2e-6 * max(1, abs(loss.item()))
    dlr = torch.zeros_like(lr)
    K.upsample_ce_bwd(lr, Cc, td, H, W, Cc, acc, 0.7, dlr)
    close(dlr[..., :Cc].permute(0, 3, 1, 2), y.grad, 2e-5, "ce bwd")
    assert dlr[..., Cc:].abs().max().item() == 0


def test_multi_exit_ce_edge_cases():
    """The degenerate inputs of the reference's per-exit CrossEntropyLoss(ignore_index=C) (my_pixelwise_xentropy.py:11-14), as torch
    itself answers them: a batch of void pixels only -> NaN loss (0 / 0) with ZERO gradients; exactly one valid pixel; one image of odd
    size; targets with the channel dim kept ([B,1,H,W], B-7)."""
    from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss
    gen = torch.Generator().manual_seed(23)
    for name, shape, fill in [("all void", (2, 2, 4, 9, 11), "void"), ("one valid pixel", (2, 2, 4, 9, 11), "one"),
                              ("one odd image", (3, 1, 5, 37, 53), "rand")]:
        E_, B, Cc, H, W = shape
        y = torch.randn(*shape, generator=gen) * 2
        if fill == "rand":
            t = torch.randint(0, Cc + 1, (B, 1, H, W), generator=gen)
        else:
            t = torch.full((B, 1, H, W), Cc)
            if fill == "one":
                t[1, 0, 4, 7] = 2
        yr = y.clone().requires_grad_(True)
        want = sum(F.cross_entropy(yr[e], t[:, 0], ignore_index=Cc) for e in range(E_))
        want.backward()
        yd = y.clone().to(DEV).requires_grad_(True)
        got = BrXEntropyLoss(ignore_index=Cc, b_reduction="sum", n_exits=E_)(yd, t.to(DEV))
        got.backward()
        if fill == "void":
            assert torch.isnan(want) and torch.isnan(got), (name, want, got)
            assert yr.grad.abs().max().item() == 0 and yd.grad.abs().max().item() == 0, name
        else:
            assert abs(got.item() - want.item()) < 2e-6 * max(1.0, abs(want.item())), (name, got.item(), want.item())
            close(yd.grad, yr.grad, 2e-5, f"ce edge grad {name}")


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_reference_golden_through_hip(path):
    """The golden vectors were produced by the reference's own loss / metric code;
    with h==H, w==W the fused kernels see an identity upsample."""
    g = np.load(path)
    y, t, void = torch.from_numpy(g["y"]), torch.from_numpy(g["t"]), int(g["void"])
    E, B, Cc, H, W = y.shape
    td = t.squeeze(1).contiguous().to(DEV)
    total = 0.0
    for e in range(E):
        lr = _lr_from_nchw(y[e])
        acc = torch.zeros(2, dtype=torch.float64, device=DEV)
        K.upsample_ce_fwd(lr, Cc, td, H, W, void, acc)
        a = acc.cpu()
        total += a[0].item() / a[1].item()
        dlr = torch.zeros_like(lr)
        K.upsample_ce_bwd(lr, Cc, td, H, W, void, acc, 1.0, dlr)
        np.testing.assert_allclose(dlr[..., :Cc].permute(0, 3, 1, 2).cpu().numpy(), g["ce_sum_grad"][e],
                                   rtol=2e-4, atol=2e-7)
        # mIoU counters (compute_mIoU.py) - exact integers
        counts, pred = K.argmax_confusion(lr, Cc, td, H, W, want_pred=True)
        assert torch.equal(pred.cpu(), y[e].argmax(1))
        if e == 0:
            np.testing.assert_array_equal(counts.cpu().numpy().astype(np.float32), g["acc0"])
        c = counts.cpu().numpy().astype(np.float32)
        with np.errstate(invalid="ignore", divide="ignore"):
            miou = float((c[0] / c.sum(0)).sum() / Cc)
        want = float(g["miou"][e])
        assert (np.isnan(miou) and np.isnan(want)) or abs(miou - want) < 1e-6
        # entropy gate (eval_br_ent.py:19-36)
        ent, flag = K.entropy_gate(lr * float(g["gate_scale"]), Cc, H, W, tau=0.5)
        np.testing.assert_allclose(ent.cpu().numpy(), g["entropy"][e], rtol=0, atol=3e-6)
        assert flag.cpu().tolist() == [int(v < 0.5) for v in ent.cpu().tolist()]
    assert abs(total - float(g["ce_sum"])) < 2e-6 * max(1.0, abs(total))


def test_entropy_gate_pooling_matches_oracle():
    from oracle import metrics_ref
    N, Cc, h, w, H, W = 2, 19, 8, 9, 33, 38
    y = torch.randn(N, Cc, h, w, generator=torch.Generator().manual_seed(18)) * 3
    up = F.interpolate(y, size=(H, W), mode="bilinear", align_corners=False)
    lr = _lr_from_nchw(y)
    for pool, s in [(0, 1), (1, 4), (2, 4), (1, 5), (2, 3)]:
        ent, _ = K.entropy_gate(lr, Cc, H, W, tau=0.3, pool=pool, pool_size=s)
        for n in range(N):
            p = metrics_ref.softmax_np(up[n].numpy(), axis=0)
            want = metrics_ref.img_norm_entropy(p, Cc, pool_min=(pool == 2), s=s if pool else 1)
            assert abs(ent[n].item() - want) < 5e-6, (pool, s, n)


def test_sgd_step_matches_torch():
    from ee_semantic_segmentation_amd import _lib
    import ctypes as C
    shapes = [(64, 3, 7, 7), (1000,), (37,), (256, 256, 3, 3)]
    ps = [torch.randn(s) for s in shapes]
    ref = [p.clone().requires_grad_(True) for p in ps]
    opt = torch.optim.SGD([{"params": ref[:2], "lr": 0.01}, {"params": ref[2:], "lr": 0.02}], lr=0.01,
                          momentum=0.9, weight_decay=5e-4)
    dev = [p.clone().to(DEV) for p in ps]
    bufs = [torch.zeros_like(p) for p in dev]
    lrs = torch.tensor([0.01, 0.01, 0.02, 0.02], device=DEV)
    sizes = torch.tensor([p.numel() for p in dev], dtype=torch.int64, device=DEV)
    for step in range(3):
        grads = [torch.randn(s, generator=torch.Generator().manual_seed(100 + step)) for s in shapes]
        for r, g in zip(ref, grads):
            r.grad = g.clone()
        opt.step()
        gd = [g.to(DEV) for g in grads]
        ptrs = torch.tensor([[p.data_ptr(), g.data_ptr(), b.data_ptr()] for p, g, b in zip(dev, gd, bufs)],
                            dtype=torch.int64, device=DEV)
        _lib.check(_lib.lib().eeseg_sgd_step(C.c_void_p(ptrs.data_ptr()), C.c_void_p(sizes.data_ptr()),
                                             C.c_void_p(lrs.data_ptr()), len(dev), 0.9, 5e-4, 1.0, int(step == 0),
                                             C.c_void_p(torch.cuda.current_stream().cuda_stream)), "sgd")
        torch.cuda.synchronize()
        for p, r in zip(dev, ref):
            close(p, r, 1e-6, f"sgd step {step}")


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p) for p in GOLD])
def test_lovasz_matches_reference_golden(path):
    """Raw-logit multi-exit Lovasz (branchy_seg_losses.py:133-159) through the HIP sort+scan
    kernels vs the values and gradients the reference itself produced."""
    from ee_semantic_segmentation_amd.branchy_seg_losses import LovaszSoftmax
    g = np.load(path)
    y, t, void = torch.from_numpy(g["y"]), torch.from_numpy(g["t"]), int(g["void"])
    E = y.shape[0]
    for prev in (False, True):
        yd = y.clone().to(DEV).requires_grad_(True)
        crit = LovaszSoftmax(classes="present", ignore=void, n_branches=E - 1, prev_out=prev)
        loss = crit(yd, t.to(DEV))
        loss.mean().backward()
        want = float(g[f"lovasz_prev{int(prev)}"])
        assert abs(loss.item() - want) < 3e-6 * max(1.0, abs(want)), (loss.item(), want)
        np.testing.assert_allclose(yd.grad.cpu().numpy(), g[f"lovasz_prev{int(prev)}_grad"], rtol=2e-4, atol=2e-7)


def test_lovasz_variants_match_reference_golden():
    """LovaszSoftmax(per_image=True), classes='all', classes=[list] (lovaszsoftmax.py:154-169,185-188) on the HIP sort +
    scan kernels (class mask + present_only flag of eeseg_lovasz; per_image = one ranking per image) vs the values and
    gradients the reference classes produced (tests/golden/lovasz_variants.npz)."""
    from ee_semantic_segmentation_amd.branchy_seg_losses import LovaszSoftmax
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lovasz_variants.npz"))
    variants = {"pi_present": dict(classes="present", per_image=True), "all": dict(classes="all", per_image=False),
                "list": dict(classes=None, per_image=False), "pi_all": dict(classes="all", per_image=True),
                "pi_list_prev": dict(classes=None, per_image=True, prev_out=True)}
    for k in range(3):
        y, t, void = torch.from_numpy(g[f"y{k}"]), torch.from_numpy(g[f"t{k}"]), int(g[f"void{k}"])
        for name, kw in variants.items():
            kw = dict(kw)
            if kw["classes"] is None:
                kw["classes"] = [int(c) for c in g[f"cls{k}"]]
            yd = y.clone().to(DEV).requires_grad_(True)
            loss = LovaszSoftmax(ignore=void, n_branches=y.shape[0] - 1, **kw)(yd, t.to(DEV))
            loss.mean().backward()
            want = float(g[f"{name}{k}"])
            assert abs(loss.item() - want) < 3e-6 * max(1.0, abs(want)), (name, k, loss.item(), want)
            np.testing.assert_allclose(yd.grad.cpu().numpy(), g[f"{name}{k}_grad"], rtol=2e-4, atol=2e-7, err_msg=f"{name}{k}")
    from ee_semantic_segmentation_amd._lib import EesegError
    with pytest.raises(EesegError):
        LovaszSoftmax(classes="some", ignore=void)(y.to(DEV), t.to(DEV))


def test_lovasz_edge_cases():
    from ee_semantic_segmentation_amd.branchy_seg_losses import lovasz_softmax
    from oracle import losses_ref
    gen = torch.Generator().manual_seed(5)
    # only void pixels -> 0 ; one pixel ; a class that is absent ; sizes that are not multiples of the scan block
    y = torch.randn(1, 4, 3, 5, generator=gen)
    t = torch.full((1, 3, 5), 4)
    assert lovasz_softmax(y.to(DEV), t.to(DEV), ignore=4).item() == 0.0
    for shape, C in [((1, 3, 1, 1), 3), ((2, 6, 47, 53), 6), ((1, 5, 70, 64), 5)]:
        y = torch.randn(*shape, generator=gen)
        t = torch.randint(0, C - 1, (shape[0], shape[2], shape[3]), generator=gen)      # class C-1 never present
        t[0, 0, 0] = C                                                                  # one void pixel
        yr = y.clone().requires_grad_(True)
        want = losses_ref.lovasz_softmax(yr, t, ignore=C)
        want.backward()
        yd = y.clone().to(DEV).requires_grad_(True)
        got = lovasz_softmax(yd, t.to(DEV), ignore=C)
        got.backward()
        assert abs(got.item() - want.item()) < 3e-6 * max(1.0, abs(want.item())), shape
        close(yd.grad, yr.grad, 1e-3, f"lovasz grad {shape}")      # gradients are O(1e-4): 1 ulp = 3e-4 rel


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_pack_weight_multi_matches_single(dtype):
    """The one-launch multi-tensor pack (LDS-tiled transpose) == per-tensor packs, incl. odd
    shapes, class padding and a torch-default (KCRS) 3x3 source."""
    import numpy as np
    ws = [torch.randn(70, 64, 3, 3).contiguous(memory_format=torch.channels_last), torch.randn(21, 256, 1, 1),
          torch.randn(64, 128, 3, 3), torch.randn(256, 1280, 1, 1)]
    pads = [None, 32, None, None]
    dev = [w.to(DEV) for w in ws]
    rec = np.zeros((len(ws), 6), dtype=np.int64)
    outs = []
    for i, (w, cp) in enumerate(zip(dev, pads)):
        co, ci, r, s_ = w.shape
        cpad = cp or co
        krsc = 1 if (w.is_contiguous(memory_format=torch.channels_last) or (r == 1 and s_ == 1)) else 0
        wf = torch.full((cpad, r, s_, ci), 7.0, dtype=dtype, device=DEV)
        wb = torch.full((ci, r, s_, cpad), 7.0, dtype=dtype, device=DEV)
        outs.append((wf, wb))
        rec[i] = [w.data_ptr(), wf.data_ptr(), wb.data_ptr(), co | (cpad << 32), ci | ((r * s_) << 32), krsc]
    K.pack_weight_multi(torch.from_numpy(rec).to(DEV), len(ws), dtype)
    for w, cp, (wf, wb) in zip(dev, pads, outs):
        rf, rb = K.pack_weight(w, dtype, cp)
        assert torch.equal(wf, rf) and torch.equal(wb, rb)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("pipe", [0, 1], ids=["lds_dma", "pipe1"])
def test_conv_alternate_pipelines(pipe, dtype):
    """The LDS-DMA staged K loop (EESEG_OPT_CONV_PIPE=0) and the 1-deep register pipeline give
    the same results as the default 2-deep one (padding taps = zero-filled DMA lanes included)."""
    from ee_semantic_segmentation_amd._lib import lib
    old = lib().eeseg_get_option(1)
    lib().eeseg_set_option(1, pipe)
    try:
        for case in [(2, 13, 11, 64, 128, 3, 1, 1, 1), (1, 17, 17, 128, 256, 3, 1, 12, 12), (3, 17, 13, 128, 64, 1, 1, 0, 1),
                     (2, 15, 14, 64, 128, 3, 2, 1, 1)]:
            test_conv_fwd_dgrad_wgrad(case, dtype)
    finally:
        lib().eeseg_set_option(1, old)
    assert lib().eeseg_get_option(1) == 3, "the 256-tile kernel is the default conv pipeline"


BIG_TILE_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil - multi-tile shapes of the 256 px x 256 cout kernel (bf16, Cout % 256 == 0)
    (2, 96, 96, 64, 1024, 1, 1, 0, 1),        # 72 pixel tiles x 4 cout tiles = 288 blocks: a full round + K-split tail
    (2, 33, 33, 128, 256, 3, 1, 12, 12),      # atrous, 9 tiles, tap skipping at the borders
    (1, 40, 52, 256, 512, 3, 1, 2, 2),        # 3x3 dilated (layer3-like), two cout tiles, ragged last pixel tile
]


@pytest.mark.parametrize("case", BIG_TILE_CASES, ids=[str(c) for c in BIG_TILE_CASES])
def test_conv_256_tile_kernel_vs_torch(case):
    """the 256-tile kernel (forward, fused epilogue, data-gradient) and the 256-tile weight-gradient
    kernel against F.conv2d itself at multi-tile shapes - not against the 128-tile kernel."""
    from ee_semantic_segmentation_amd._lib import lib
    assert lib().eeseg_get_option(1) == 3
    N, H, W, Cin, Cout, k, s, p, d = case
    dtype = torch.bfloat16
    x = rnd(dtype, N, Cin, H, W, seed=1).requires_grad_(True)
    w = rnd(dtype, Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p, dilation=d)
    gy = rnd(dtype, *y.shape, seed=3)
    y.backward(gy)
    xd = nhwc(x.detach()).to(DEV, dtype)
    wf, wb = K.pack_weight(w.detach().to(DEV), dtype)
    yd, part = K.conv_fwd(xd, wf, s, p, d, want_stats=True)
    close(nchw(yd), y, tol(dtype), "256-tile conv fwd")
    ys = yd.float().reshape(-1, Cout)
    sums = K.reduce_partials(part)
    close(sums[0], ys.sum(0), 1e-4, "stats sum")
    close(sums[1], (ys * ys).sum(0), 1e-4, "stats sumsq")
    sc = torch.rand(Cout, generator=torch.Generator().manual_seed(1)) + 0.5
    sh = torch.randn(Cout, generator=torch.Generator().manual_seed(2))
    res = rnd(dtype, *y.shape, seed=4)
    y2, _ = K.conv_fwd(xd, wf, s, p, d, scale=sc.to(DEV), shift=sh.to(DEV), residual=nhwc(res).to(DEV, dtype), relu=True)
    want2 = torch.relu(y.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    close(nchw(y2), want2, tol(dtype), "256-tile fused epilogue")
    gyd = nhwc(gy).to(DEV, dtype)
    if Cin % 256 == 0:                     # the data-gradient is a conv with Cout' = Cin: 256-tile eligible
        dx = K.conv_dgrad(gyd, wb, (H, W), s, p, d)
        close(nchw(dx), x.grad, tol(dtype), "256-tile conv dgrad")
    dw = K.conv_wgrad(xd, gyd, k, k, s, p, d)
    close(dw.permute(0, 3, 1, 2), w.grad, 4e-3, "conv wgrad")


def test_bn_finalize_apply_and_dual_output_reduce_equal_the_separate_launches():
    """SyncBN path: eeseg_bn_finalize_apply (finalize inside the apply pass) must equal eeseg_bn_finalize followed by
    eeseg_bn_apply / _relu_mask BIT FOR BIT (outputs, masks, coefficients, running statistics), and eeseg_bn_bwd_reduce's
    optional second output must be a copy of the first."""
    g = torch.Generator().manual_seed(33)
    for dtype in DTYPES:
        for rows_shape, Cc, relu, use_res in [((2, 17, 19), 64, True, True), ((4, 33, 33), 256, True, False),
                                              ((1, 9, 9), 1024, False, False), ((3, 8, 8), 128, True, True)]:
            x = (torch.randn(*rows_shape, Cc, generator=g) * 2 + 0.3).to(DEV, dtype)
            res = torch.randn(*rows_shape, Cc, generator=g).to(DEV, dtype) if use_res else None
            rows = x.numel() // Cc
            xf = x.float().reshape(rows, Cc)
            sums = torch.stack([xf.sum(0), (xf * xf).sum(0)]).contiguous()
            gamma, beta = torch.rand(Cc, device=DEV) + 0.5, torch.randn(Cc, device=DEV)
            rm1, rv1 = torch.randn(Cc, device=DEV), torch.rand(Cc, device=DEV) + 0.5
            rm2, rv2 = rm1.clone(), rv1.clone()
            mi1, ss1 = K.bn_finalize(sums, rows, gamma, beta, 1e-5, 0.1, rm1, rv1)
            if use_res and relu:
                y1, m1 = K.bn_apply(x, ss1, residual=res, relu=True, want_mask=True)
            else:
                y1, m1 = K.bn_apply(x, ss1, residual=res, relu=relu), None
            y2, m2, mi2, ss2 = K.bn_finalize_apply(x, sums, rows, gamma, beta, 1e-5, 0.1, rm2, rv2, residual=res, relu=relu,
                                                   want_mask=m1 is not None)
            assert torch.equal(y1, y2) and torch.equal(mi1, mi2) and torch.equal(ss1, ss2)
            assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2)
            assert (m1 is None and m2 is None) or torch.equal(m1, m2)
            dy = torch.randn(*rows_shape, Cc, generator=g).to(DEV, dtype)
            a = K.bn_bwd_reduce(dy, None, x, mi1, relu, scale_shift=ss1)
            out, cp = torch.empty(2, Cc, device=DEV), torch.empty(2, Cc, device=DEV)
            K.bn_bwd_reduce(dy, None, x, mi1, relu, out=out, scale_shift=ss1, copy=cp)
            assert torch.equal(a, out) and torch.equal(out, cp)


def test_bn_reduce_finalize_fused_equals_two_step():
    g = torch.Generator().manual_seed(21)
    for tiles, Cc in [(1, 64), (37, 256), (529, 2048), (8257, 64)]:
        part = torch.rand(tiles, 2, Cc, generator=g).to(DEV) * 100
        part[:, 1] += 50
        gamma, beta = torch.rand(Cc, device=DEV) + 0.5, torch.randn(Cc, device=DEV)
        rm1, rv1 = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
        rm2, rv2 = rm1.clone(), rv1.clone()
        cnt = tiles * 128
        mi1, ss1 = K.bn_finalize(K.reduce_partials(part), cnt, gamma, beta, 1e-5, 0.1, rm1, rv1)
        mi2, ss2 = K.bn_reduce_finalize(part, cnt, gamma, beta, 1e-5, 0.1, rm2, rv2)
        for a, b, what in [(mi1, mi2, "mean/invstd"), (ss1, ss2, "scale/shift"), (rm1, rm2, "rm"), (rv1, rv2, "rv")]:
            close(b, a, 1e-5, f"{what} tiles={tiles} C={Cc}")


@pytest.mark.parametrize("shape", [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 33, 33, 128, 256, 3, 1, 12, 12),      # atrous: whole taps masked for many pixels, ragged last tile
    (1, 16, 16, 64, 256, 1, 1, 0, 1),         # exactly one 256-pixel tile, one K tile
    (3, 19, 23, 256, 512, 3, 1, 1, 1),        # two cout tiles, M not a multiple of 256 (+ dgrad)
    (1, 9, 9, 64, 256, 3, 1, 2, 2),           # fewer pixels than one tile
    (2, 31, 31, 256, 256, 3, 2, 1, 1),        # strided forward (smul = 2)
    (2, 96, 96, 64, 1024, 1, 1, 0, 1),        # 288 tiles: one full round of 256 + a K-split tail in the SAME launch
])
def test_conv_256_tile_kernel_matches_the_128_tile_kernel(shape):
    """EESEG_OPT_CONV_PIPE=3 (256x256 tile, loads in flight across barriers) accumulates in the same K order
    with the same MFMA as the default kernel: with the split-K tail off (EESEG_OPT_CONV_TAIL_MIN=0) outputs are
    bit identical; with it on (default) the tail tiles sum K ranges in another order -> one bf16 ulp."""
    from ee_semantic_segmentation_amd._lib import lib
    N, H, W, Cin, Cout, k, s, p, d = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV).bfloat16()
    wt = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(DEV)
    wf, wb = K.pack_weight(wt, torch.bfloat16)
    sc = torch.rand(Cout, generator=torch.Generator().manual_seed(1)).to(DEV) + 0.5
    sh = torch.randn(Cout, generator=torch.Generator().manual_seed(2)).to(DEV)
    res = None
    outs = {}
    try:
        for name, pipe, tail, merge, inner, cus in (("ref", 0, 224, 1, 0, 256), ("big", 3, 0, 1, 0, 256),
                                                    ("split", 3, 224, 1, 0, 256), ("split2", 3, 224, 0, 0, 256),
                                                    ("inner", 3, 224, 1, 1, 256), ("cus", 3, 224, 1, 0, 120)):
            lib().eeseg_set_option(1, pipe)
            lib().eeseg_set_option(8, cus)         # CUs the launch plan counts on (rounds / K-split tail sized for them)
            lib().eeseg_set_option(5, tail)
            lib().eeseg_set_option(7, merge)       # tail blocks in the main launch (default) or in their own
            lib().eeseg_set_option(2, inner)       # K order: taps outer (default) / taps inner
            y, part = K.conv_fwd(x, wf, s, p, d, want_stats=True)
            if res is None:
                res = torch.randn(y.shape, generator=g).to(DEV).bfloat16()
            y2, _ = K.conv_fwd(x, wf, s, p, d, scale=sc, shift=sh, residual=res, relu=True)
            dx = K.conv_dgrad(y, wb, (H, W), s, p, d) if s == 1 and Cin % 256 == 0 else None
            torch.cuda.synchronize()
            outs[name] = (y, K.reduce_partials(part), y2, dx)
    finally:
        lib().eeseg_set_option(1, 3)
        lib().eeseg_set_option(5, 224)
        lib().eeseg_set_option(7, 1)
        lib().eeseg_set_option(2, 0)
        lib().eeseg_set_option(8, 256)
    ref = outs["ref"]
    for i, what in ((0, "y"), (2, "fused epilogue"), (3, "dgrad")):      # taps-inner K order: another summation order
        if ref[i] is not None:
            close(outs["inner"][i], ref[i], 8e-3, f"taps-inner {what}")
            close(outs["cus"][i], ref[i], 8e-3, f"plan for 120 CUs {what}")
    close(outs["cus"][1], ref[1], 2e-3, "plan for 120 CUs BN partial sums")
    for i in (0, 2, 3):                    # merged and separate tail launches do the same arithmetic
        if outs["split"][i] is not None:
            assert torch.equal(outs["split"][i], outs["split2"][i])
    assert torch.equal(ref[0], outs["big"][0])
    assert torch.equal(ref[2], outs["big"][2])
    close(outs["big"][1], ref[1], 1e-5, "BN partial sums")
    if ref[3] is not None:
        assert torch.equal(ref[3], outs["big"][3])
    for i, what in ((0, "y"), (2, "fused epilogue"), (3, "dgrad")):
        if ref[i] is not None:
            close(outs["split"][i], ref[i], 8e-3, f"split-K tail {what}")
    close(outs["split"][1], ref[1], 2e-3, "split-K BN partial sums")


PW_CASES = [
    # N, H, W, Cin, Cout - pointwise layers of the two-blocks-per-CU 128x256 kernel (conv_pw_kernel: Cin <= 1280)
    (2, 65, 65, 256, 1024),       # bottleneck conv3: 67 pixel tiles (ragged last one) x 4 cout tiles
    (16, 65, 65, 256, 1024),      # the same on the weight-stationary persistent kernel (Cin = 256, >= 512 tiles): 529 tiles, ragged
    (3, 33, 31, 1024, 256),       # bottleneck conv1: 32 K tiles (contracting: only its residual forms go to the kernel)
    (1, 9, 11, 64, 256),          # fewer pixels than one tile, two K tiles (the pipeline prologue covers the whole K loop)
    (2, 40, 52, 1280, 256),       # ASPP projection: 40 K tiles
]


@pytest.mark.parametrize("case", PW_CASES, ids=[str(c) for c in PW_CASES])
def test_conv_pointwise_kernel_vs_torch(case):
    """conv_pw_kernel (1x1, stride 1, bf16, Cout % 256 == 0, Cin <= EESEG_OPT_CONV_PW_MAX_K) against F.conv2d:
    plain forward + BN partial sums, fused scale/shift/residual/ReLU epilogue, output into a channel slice of a wider
    buffer (ASPP concat), data-gradient with in-place accumulation, and the same calls on the 256-tile kernel
    (EESEG_OPT_CONV_PW_MAX_K = 0) as a cross-check of the dispatch."""
    from ee_semantic_segmentation_amd._lib import lib
    assert lib().eeseg_get_option(13) >= 1280
    N, H, W, Cin, Cout = case
    dtype = torch.bfloat16
    x = rnd(dtype, N, Cin, H, W, seed=1).requires_grad_(True)
    w = rnd(dtype, Cout, Cin, 1, 1, seed=2, scale=Cin ** -0.5).requires_grad_(True)
    y = F.conv2d(x, w)
    gy = rnd(dtype, *y.shape, seed=3)
    y.backward(gy)
    xd = nhwc(x.detach()).to(DEV, dtype)
    wf, wb = K.pack_weight(w.detach().to(DEV), dtype)
    sc = torch.rand(Cout, generator=torch.Generator().manual_seed(1)) + 0.5
    sh = torch.randn(Cout, generator=torch.Generator().manual_seed(2))
    res = rnd(dtype, *y.shape, seed=4)
    want2 = torch.relu(y.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    outs = {}
    variants = [("pw", 1280, 1), ("big", 0, 0)] + ([("pw128", 1280, 0)] if Cin == 256 and N * H * W >= 512 * 128 else [])
    for name, maxk, ws in variants:
        lib().eeseg_set_option(13, maxk)
        lib().eeseg_set_option(14, ws)
        try:
            assert lib().eeseg_get_option(13) == maxk
            yd, part = K.conv_fwd(xd, wf, want_stats=True)
            wide = torch.full((N, H, W, Cout + 256), 7.0, dtype=dtype, device=DEV)
            K.conv_fwd(xd, wf, out=wide[..., 128:128 + Cout])
            y2, _ = K.conv_fwd(xd, wf, scale=sc.to(DEV), shift=sh.to(DEV), residual=nhwc(res).to(DEV, dtype), relu=True)
            outs[name] = (yd, K.reduce_partials(part), wide, y2)
        finally:
            lib().eeseg_set_option(13, 1280)
            lib().eeseg_set_option(14, 1)
    if "pw128" in outs:                 # weight-stationary vs 128x256 kernel: same MFMAs in the same K order
        assert torch.equal(outs["pw"][0], outs["pw128"][0])
        close(outs["pw"][1], outs["pw128"][1], 1e-5, "BN partial sums")
    for name, (yd, sums, wide, y2) in outs.items():
        close(nchw(yd), y, tol(dtype), f"{name} fwd")
        ys = yd.float().reshape(-1, Cout)
        close(sums[0], ys.sum(0), 1e-4, f"{name} stats sum")
        close(sums[1], (ys * ys).sum(0), 1e-4, f"{name} stats sumsq")
        assert torch.equal(wide[..., 128:128 + Cout], yd), f"{name} slice output"
        assert bool((wide[..., :128] == 7.0).all()) and bool((wide[..., 128 + Cout:] == 7.0).all()), f"{name} wrote outside its slice"
        close(nchw(y2), want2, tol(dtype), f"{name} fused epilogue")
    close(outs["pw"][0], outs["big"][0], 8e-3, "pw vs 256-tile kernel")     # (the latter may split K over idle CUs)
    # data-gradient = pointwise conv with the roles of Cin / Cout swapped (eligible when Cin % 256 == 0)
    if Cin % 256 == 0:
        gyd = nhwc(gy).to(DEV, dtype)
        dx = K.conv_dgrad(gyd, wb, (H, W))
        close(nchw(dx), x.grad, tol(dtype), "pw dgrad")
        dx2 = K.conv_dgrad(gyd, wb, (H, W), accumulate_into=dx.clone())
        close(nchw(dx2), 2 * x.grad, 2 * tol(dtype), "pw dgrad accumulate")


@pytest.mark.parametrize("shape", [
    # N, H, W, Cin, Cout, k, stride, pad, dil
    (2, 33, 33, 256, 256, 3, 1, 12, 12),      # atrous: K tiles that are all padding get skipped
    (1, 20, 20, 512, 256, 1, 1, 0, 1),        # 1x1, two cin tiles, M not a multiple of 64
    (3, 19, 23, 256, 512, 3, 1, 1, 1),        # two cout tiles, ragged pixel count
    (1, 12, 12, 256, 256, 3, 1, 2, 2),        # few K tiles per block, 5 image rows per K tile
    (2, 31, 31, 256, 256, 3, 2, 1, 1),        # strided
    (2, 20, 70, 256, 256, 3, 1, 6, 6),        # Wout > 64: the straight-line in-loop iterator (FAST form), atrous skips
    (1, 30, 131, 256, 512, 1, 1, 0, 1),       # FAST form, 1x1, ragged
    (2, 37, 141, 256, 256, 3, 2, 1, 1),       # FAST form, strided (Wout = 71)
])
def test_wgrad_256_tile_kernel_matches_the_128_tile_kernel(shape):
    from ee_semantic_segmentation_amd._lib import lib
    N, H, W, Cin, Cout, k, s, p, d = shape
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, H, W, Cin, generator=g).to(DEV).bfloat16()
    Ho, Wo = K.conv_out_size(H, k, s, p, d), K.conv_out_size(W, k, s, p, d)
    dy = torch.randn(N, Ho, Wo, Cout, generator=g).to(DEV).bfloat16()
    outs = {}
    try:
        # 2 = force the 256x256 kernel (K splits by atomics), 6 = same through slabs; +16 = on v_mfma_f32_16x16x32_bf16
        # (other transposing-read geometry, LDS swizzle, accumulator layout in the atomics / slabs / slab reduce)
        for big in (0, 2, 6, 18, 22):
            lib().eeseg_set_wgrad_big(big)
            K.WGRAD_SLABS = bool(big & 4)
            dw = K.conv_wgrad(x, dy, k, k, s, p, d)
            dw2 = K.conv_wgrad(x, dy, k, k, s, p, d, out=dw.clone(), accumulate=True)
            if big & 4:                         # slab combine sums the K splits in a fixed order: bitwise reproducible
                assert torch.equal(K.conv_wgrad(x, dy, k, k, s, p, d), dw)
            torch.cuda.synchronize()
            outs[big] = (dw, dw2)
    finally:
        lib().eeseg_set_wgrad_big(1)
        K.WGRAD_SLABS = False
    for mode in (2, 6, 18, 22):
        close(outs[mode][0], outs[0][0], 2e-5, f"dw (mode {mode})")
        close(outs[mode][1], 2 * outs[0][0], 2e-5, f"dw accumulate (mode {mode})")
    # and against fp32 torch on the same bf16 inputs
    ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (Cout, Cin, k, k), dy.float().permute(0, 3, 1, 2),
                                      stride=s, padding=p, dilation=d).permute(0, 2, 3, 1)
    close(outs[2][0], ref, 2e-5, "dw vs torch")


def test_img_miou_matches_oracle_and_reference_vectors():
    """Per-image mIoU (compute_mIoU.py:38-63) on the fused argmax+confusion kernel vs the reference-generated
    vectors (tests/golden/img_miou.npz) and, on an upsampled low-res exit, vs the oracle."""
    import os
    import numpy as np
    from ee_semantic_segmentation_amd.compute_mIoU import img_mIoU
    from ee_semantic_segmentation_amd.from_deepv3_new import ExitLogits
    from oracle.metrics_ref import img_mIoU as RefM
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "img_miou.npz"))
    for i, want in enumerate(g["expected"]):
        m = img_mIoU()
        for y, t in zip(g[f"y{i}"], g[f"t{i}"]):
            m(torch.from_numpy(y[None]).to(DEV), torch.from_numpy(t[None]).to(DEV))
        assert abs(m.compute() - want) < 1e-6, (i, m.compute(), want)
    # low-res exit logits upsampled inside the kernel
    gen = torch.Generator().manual_seed(3)
    C, H, W = 19, 65, 97
    lo = torch.zeros(1, 9, 13, 32)
    lo[..., :C] = torch.randn(1, 9, 13, C, generator=gen) * 3
    t = torch.randint(0, C + 1, (1, H, W), generator=gen)
    el = ExitLogits([lo.to(DEV)], C, (H, W))
    m, r = img_mIoU(), RefM()
    m(el, t.to(DEV), 0)
    r(el.stack()[0].cpu().numpy(), t.numpy())
    assert abs(m.compute() - r.compute()) < 1e-6


def test_similarity_gates_from_pair_histogram():
    """MSE / NMI / VI between the label maps of two exits from the fused pair-histogram kernel vs the numpy oracle
    (restated scikit-image formulas: parity unpinned, see oracle/sim_ref.py) + the identities the metrics must obey."""
    import numpy as np
    from ee_semantic_segmentation_amd import sim_metrics as M
    from ee_semantic_segmentation_amd.from_deepv3_new import ExitLogits
    from oracle import sim_ref as R
    gen = torch.Generator().manual_seed(11)
    C, H, W = 21, 65, 97
    los = []
    for k in range(2):
        lo = torch.zeros(1, 9, 13, 32)
        lo[..., :C] = torch.randn(1, 9, 13, C, generator=gen) * 3
        los.append(lo)
    los.append(los[0] + 0.3 * los[1])            # a third exit correlated with the first
    el = ExitLogits([l.to(DEV) for l in los], C, (H, W))
    full = el.stack().cpu().numpy()              # [E,1,C,H,W]
    for ea, eb in ((0, 1), (0, 2), (2, 1)):
        a, b = full[ea], full[eb]
        t = M.pair_table(el, None, ea, eb).cpu().numpy()
        la, lb = R.label_maps(a, b)
        ref_t = np.zeros((C, C)); np.add.at(ref_t, (la.reshape(-1), lb.reshape(-1)), 1)
        assert np.array_equal(t, ref_t)
        kw = dict(exit_a=ea, exit_b=eb)
        assert abs(M.MSE(el, None, **kw) - R.mse(a, b)) < 1e-9
        assert abs(M.NMI(el, None, **kw) - R.nmi(a, b)) < 1e-9
        for ign in ((), (0, 20)):
            assert abs(M.VI(ign)(el, None, **kw) - R.vi(a, b, ign)) < 1e-9
            pair = R.vi_pair(a, b, ign)
            assert abs(M.Seg_comp(True, ign)(el, None, **kw) - pair[1]) < 1e-9
            assert abs(M.Seg_comp(False, ign)(el, None, **kw) - pair[0]) < 1e-9
        # score-tensor and label-map entry points agree with the fused one
        ta, tb = torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)
        assert abs(M.MSE(ta, tb) - R.mse(a, b)) < 1e-9
        assert abs(M.NMI(torch.from_numpy(la).to(DEV), torch.from_numpy(lb).to(DEV)) - R.nmi(a, b)) < 1e-9
    # identities: identical maps -> MSE 0, VI 0, NMI 2
    kw = dict(exit_a=0, exit_b=0)
    assert M.MSE(el, None, **kw) == 0.0 and abs(M.VI()(el, None, **kw)) < 1e-12 and abs(M.NMI(el, None, **kw) - 2.0) < 1e-9


def test_ssim_gate_matches_oracle():
    """sim_metrics.SSIM (sim_metrics.py:15-37) on the device vs the numpy restatement of skimage's
    structural_similarity defaults (oracle/sim_ref.py: parity unpinned, scikit-image is not installed) + identities."""
    from ee_semantic_segmentation_amd import sim_metrics as M
    from ee_semantic_segmentation_amd.eval_br_sim import br_evaluator
    from ee_semantic_segmentation_amd.from_deepv3_new import ExitLogits
    from oracle import sim_ref as R
    gen = torch.Generator().manual_seed(12)
    C, H, W = 21, 70, 101
    los = []
    for k in range(2):
        lo = torch.zeros(2, 9, 13, 32)
        lo[..., :C] = torch.randn(2, 9, 13, C, generator=gen) * 3
        los.append(lo)
    los.append(los[0] + 0.5 * los[1])
    el = ExitLogits([l.to(DEV) for l in los], C, (H, W))
    full = el.stack().cpu().numpy()              # [E,2,C,H,W]
    f = M.SSIM(C - 1)
    for ea, eb in ((0, 1), (0, 2), (2, 1)):
        got = f.device_value(el, el, ea, eb).cpu().numpy()
        for b in range(2):
            want = R.ssim(full[ea][b:b + 1], full[eb][b:b + 1], C - 1)
            assert abs(got[b] - want) < 1e-12, (ea, eb, b, got[b], want)
        # score-tensor and label-map entry points (what the reference passes) agree
        ta, tb = torch.from_numpy(full[ea][:1]).to(DEV), torch.from_numpy(full[eb][:1]).to(DEV)
        la, lb = R.label_maps(full[ea][:1], full[eb][:1])
        assert abs(f(ta, tb) - got[0]) < 1e-15
        assert abs(f(torch.from_numpy(la).to(DEV), torch.from_numpy(lb).to(DEV)) - got[0]) < 1e-15
    assert abs(f.device_value(el, el, 1, 1).cpu().numpy() - 1.0).max() < 1e-15     # identical maps -> 1
    small = torch.randint(0, C, (1, 7, 7), generator=gen).to(DEV)                   # exactly one window
    assert abs(float(K.ssim_labels(small, small, C - 1)[0]) - 1.0) < 1e-15
    # the evaluator accepts metric='ssim' (eval_br_sim.py:20-21)

    class Net(torch.nn.Module):
        n_branches, num_classes = 2, C

        def forward_lowres(self, X):
            return el.lowres

    y = torch.randint(0, C, (2, 1, H, W), generator=gen)
    res = br_evaluator(Net(), 3, C, [(torch.zeros(2, 3, H, W), y)], DEV, "ssim", 0.2)
    assert res["out_gl"] == 2 and res["b2_count"] + res["count_out"] == 2


@pytest.mark.parametrize("k", [0, 1, 2])
def test_unreduced_focal_loss_matches_reference_vectors(k):
    """FocalLoss(reduction='none') - VERDICT r3 missing 4 - on eeseg_focal_map_fwd / _bwd: the stacked per-pixel maps [E,B,H,W]
    (with alpha and the default faithful_alpha: [E,B,B,H,W], the reference's broadcast) and the gradient of (map * weights).sum()
    against vectors produced by the reference class (tests/golden/focal_unreduced.npz: batch sizes 2, 1, 3; gamma 2, 1.5, 0); the
    intended per-pixel alpha form against the oracle; a void label fails like the reference's gather."""
    import os
    import numpy as np
    from ee_semantic_segmentation_amd import branchy_seg_losses as B
    from oracle import losses_ref as L
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "focal_unreduced.npz"))
    y, t = torch.from_numpy(g[f"y{k}"]), torch.from_numpy(g[f"t{k}"])
    E, C, gamma = y.shape[0], y.shape[2], float(g[f"gamma{k}"])
    alpha = torch.linspace(0.5, 1.5, C)
    for name, a in (("plain", None), ("alpha", alpha)):
        yy = y.clone().to(DEV).requires_grad_(True)
        m = B.FocalLoss(alpha=a, gamma=gamma, reduction="none", n_branches=E - 1)(yy, t.to(DEV))
        want = g[f"{name}{k}"]
        assert tuple(m.shape) == want.shape, (name, m.shape, want.shape)
        assert np.abs(m.detach().cpu().numpy() - want).max() <= 2e-5 * np.abs(want).max(), name
        (m * torch.from_numpy(g[f"{name}{k}_w"]).to(DEV)).sum().backward()
        wg = g[f"{name}{k}_grad"]
        assert np.abs(yy.grad.cpu().numpy() - wg).max() <= 2e-4 * np.abs(wg).max() + 1e-8, name
    yy = y.clone().to(DEV).requires_grad_(True)
    m = B.FocalLoss(alpha=alpha, gamma=gamma, reduction="none", n_branches=E - 1, faithful_alpha=False)(yy, t.to(DEV))
    yr = y.clone().requires_grad_(True)
    mr = L.br_focal(yr, t, E, alpha=alpha, gamma=gamma, reduction="none", faithful_alpha=False)
    assert tuple(m.shape) == tuple(mr.shape) == (E,) + tuple(t.squeeze(1).shape)
    assert float((m.detach().cpu() - mr.detach()).abs().max()) <= 2e-5 * float(mr.abs().max())
    wgt = torch.randn(mr.shape, generator=torch.Generator().manual_seed(7))
    (m * wgt.to(DEV)).sum().backward()
    (mr * wgt).sum().backward()
    assert float((yy.grad.cpu() - yr.grad).abs().max()) <= 2e-4 * float(yr.grad.abs().max()) + 1e-8
    tv = t.clone()
    tv[0, 0, 0, 0] = C
    with pytest.raises(RuntimeError):
        B.FocalLoss(gamma=gamma, reduction="none", n_branches=E - 1)(y.to(DEV), tv.to(DEV))


@pytest.mark.parametrize("k", [0, 1, 2])
def test_region_and_focal_losses_match_reference_vectors(k):
    """Dice / Jaccard / Tversky / FocalTversky / Focal on the fused class-sums kernels vs values and gradients produced
    by the reference classes (tests/golden/region_losses.npz), then from LOW-RES exits (upsample fused) vs the oracle."""
    import os
    import numpy as np
    from ee_semantic_segmentation_amd import branchy_seg_losses as B
    from ee_semantic_segmentation_amd.from_deepv3_new import ExitLogits
    from oracle import losses_ref as L
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "region_losses.npz"))
    y = torch.from_numpy(g[f"y{k}"])
    E, Bn, C = y.shape[0], y.shape[1], y.shape[2]
    t, tv = torch.from_numpy(g[f"t{k}"]).to(DEV), torch.from_numpy(g[f"tv{k}"]).to(DEV)
    alpha = torch.linspace(0.5, 1.5, C)
    specs = {
        "dice_mean": (B.DiceLoss(reduction="mean", n_branches=E - 1), t),
        "dice_sum": (B.DiceLoss(reduction="sum", n_branches=E - 1, weights=[0.5 + i for i in range(E)]), t),
        "jaccard_mean": (B.JaccardLoss(reduction="mean", n_branches=E - 1, downgrad_bg=0.3), tv),
        "jaccard_sum0": (B.JaccardLoss(reduction="sum", n_branches=E - 1, downgrad_bg=0.0), tv),
        "tversky": (B.TverskyLoss(alpha=.3, beta=.7, reduction="mean", n_branches=E - 1), t),
        "focal_tversky": (B.FocalTverskyLoss(alpha=.3, beta=.7, gamma=1.5, reduction="sum", n_branches=E - 1), t),
        "focal_mean": (B.FocalLoss(gamma=2, reduction="mean", n_branches=E - 1), t),
        # with alpha the reference broadcasts [B,H,W] x [B,1,H,W] -> [B,B,H,W] (branchy_seg_losses.py:126-129): the default
        # faithful_alpha=True reproduces it for EVERY batch size (the golden vectors hold B = 2, 1, 3)
        "focal_sum_alpha": (B.FocalLoss(alpha=alpha, gamma=1.5, reduction="sum", n_branches=E - 1), t),
    }
    for name, (crit, tt) in specs.items():
        yy = y.clone().to(DEV).requires_grad_(True)
        l = crit(yy, tt)
        want = float(g[f"{name}{k}"])
        assert abs(float(l) - want) <= 2e-5 * max(1.0, abs(want)), (name, float(l), want)
        l.backward()
        got = np.zeros_like(g[f"y{k}"]) if yy.grad is None else yy.grad.cpu().numpy()
        wg = g[f"{name}{k}_grad"]
        assert np.abs(got - wg).max() <= 2e-4 * max(1e-6, np.abs(wg).max()) + 1e-8, name
    # faithful 'mean' vs the oracle's restatement of the broadcast, and the intended per-pixel form (faithful_alpha=False)
    for faithful in (True, False):
        yy = y.clone().to(DEV).requires_grad_(True)
        l = B.FocalLoss(alpha=alpha, gamma=2, reduction="mean", n_branches=E - 1, faithful_alpha=faithful)(yy, t)
        l.backward()
        yr = y.clone().requires_grad_(True)
        lr_ = L.br_focal(yr, t.cpu(), E, alpha=alpha, gamma=2, reduction="mean", faithful_alpha=faithful)
        lr_.backward()
        assert abs(float(l) - float(lr_)) <= 2e-5 * max(1.0, abs(float(lr_))), (faithful, float(l), float(lr_))
        assert (yy.grad.cpu() - yr.grad).abs().max() <= 2e-4 * yr.grad.abs().max() + 1e-9, faithful
    # void labels: only Jaccard accepts them, the others fail like the reference's one_hot / gather
    with pytest.raises(RuntimeError):
        B.DiceLoss(n_branches=E - 1)(y.to(DEV), tv)
    # low-res exits, upsample fused into the kernels, vs the oracle on the materialised stack
    gen = torch.Generator().manual_seed(40 + k)
    H, W = 33, 41
    los = []
    for _ in range(2):
        lo = torch.zeros(2, 5, 6, 32)
        lo[..., :C] = torch.randn(2, 5, 6, C, generator=gen) * 2
        los.append(lo.to(DEV).requires_grad_(True))
    el = ExitLogits(los, C, (H, W))
    tt = torch.randint(0, C, (2, 1, H, W), generator=gen)
    stack = el.stack().detach().cpu()
    pairs = [(B.DiceLoss(n_branches=1), L.br_dice(stack, tt, 2)),
             (B.JaccardLoss(n_branches=1, downgrad_bg=0.5), L.br_jaccard(stack, tt, 2, downgrad_bg=0.5)),
             (B.FocalLoss(alpha=alpha, gamma=2, n_branches=1), L.br_focal(stack, tt, 2, alpha=alpha, gamma=2, faithful_alpha=True)),
             (B.FocalLoss(alpha=alpha, gamma=2, n_branches=1, faithful_alpha=False),
              L.br_focal(stack, tt, 2, alpha=alpha, gamma=2, faithful_alpha=False))]
    for crit, want in pairs:
        l = crit(el, tt.to(DEV))
        assert abs(float(l) - float(want)) <= 2e-5 * max(1.0, abs(float(want))), (type(crit).__name__, float(l), float(want))
        for lo in los:
            lo.grad = None
        l.backward()
        assert all(lo.grad is not None and torch.isfinite(lo.grad).all() and float(lo.grad.abs().sum()) > 0 for lo in los)
        # the low-res gradient (softmax Jacobian + transposed interpolation) against a central difference along
        # sign(grad) of the second exit, the direction in which the first-order change is largest
        d = torch.sign(los[1].grad)
        first_order = float((los[1].grad * d).sum())
        eps = 2e-2
        with torch.no_grad():
            vals = []
            for sgn in (1.0, -1.0):
                los[1] += sgn * eps * d
                vals.append(float(crit(el, tt.to(DEV))))
                los[1] -= sgn * eps * d
        g_fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(first_order - g_fd) <= 3e-2 * abs(g_fd) + 1e-5, (type(crit).__name__, first_order, g_fd)


@pytest.mark.parametrize("hw,HW,Cc", [((65, 65), (513, 513), 21), ((97, 97), (769, 769), 19), ((33, 40), (33, 40), 21),
                                      ((8, 130), (60, 1030), 7)])
def test_ce_thread_per_span_kernels_match_half_wave_kernels(hw, HW, Cc):
    """EESEG_OPT_CE_SPAN: the thread-per-span cross-entropy kernels (default) vs the half-wave-per-pixel ones at the
    training shapes (and a wide map that takes the fallback in the backward: LDS tile too large)."""
    from ee_semantic_segmentation_amd._lib import lib
    (h, w), (H, W) = hw, HW
    N = 2
    gen = torch.Generator().manual_seed(23)
    lr = torch.zeros(N, h, w, 32)
    lr[..., :Cc] = torch.randn(N, h, w, Cc, generator=gen) * 3
    lr = lr.to(DEV)
    t = torch.randint(0, Cc + 1, (N, H, W), generator=gen).to(DEV)
    out = {}
    try:
        for mode in (0, 1):
            lib().eeseg_set_option(6, mode)
            acc = torch.zeros(2, dtype=torch.float64, device=DEV)
            K.upsample_ce_fwd(lr, Cc, t, H, W, Cc, acc)
            dlr = torch.zeros_like(lr)
            K.upsample_ce_bwd(lr, Cc, t, H, W, Cc, acc, 0.9, dlr)
            torch.cuda.synchronize()
            out[mode] = (acc.cpu(), dlr)
    finally:
        lib().eeseg_set_option(6, 1)
    assert out[0][0][1] == out[1][0][1] == float((t != Cc).sum())
    assert abs(out[0][0][0] - out[1][0][0]) <= 2e-6 * abs(out[0][0][0])
    close(out[1][1], out[0][1], 2e-5, "ce bwd span vs half-wave")
    assert out[1][1][..., Cc:].abs().max().item() == 0


@pytest.mark.parametrize("shape", [(2, 33, 31, 256, 1024), (2, 33, 31, 1024, 256), (1, 40, 52, 512, 2048),
                                   (16, 65, 65, 256, 1024)],
                         ids=["pw-256-1024", "big-1024-256", "pw-512-2048", "ws-256-1024"])
def test_dgrad_adds_masked_residual(shape):
    """conv_dgrad(add=(t, mask)): dx = dgrad(dy) + t * mask with the 1-bit ReLU mask of bn_apply (eeseg_conv_args.
    residual_mask), on the pointwise and the 256-tile kernel (incl. its K-split tail), against the two-step form."""
    N, H, W, Cdy, Cdx = shape
    g = torch.Generator().manual_seed(4)
    dy = torch.randn(N, H, W, Cdy, generator=g).to(DEV).bfloat16()
    wt = (torch.randn(Cdy, Cdx, 1, 1, generator=g) * 0.05).to(DEV)          # forward conv Cdx -> Cdy
    _, wb = K.pack_weight(wt, torch.bfloat16)
    c = torch.randn(N, H, W, Cdx, generator=g).to(DEV).bfloat16()
    res = torch.randn(N, H, W, Cdx, generator=g).to(DEV).bfloat16()
    ss = torch.stack([torch.rand(Cdx, generator=g) + 0.5, torch.randn(Cdx, generator=g) * 0.1]).to(DEV)
    y, mask = K.bn_apply(c, ss, residual=res, relu=True, want_mask=True)
    t = torch.randn(N, H, W, Cdx, generator=g).to(DEV).bfloat16()
    want = K.conv_dgrad(dy, wb, (H, W), accumulate_into=(t * (y > 0)).contiguous())
    got = K.conv_dgrad(dy, wb, (H, W), add=(t, mask))
    assert torch.equal(got, want)
    assert float((y > 0).float().mean()) > 0.2 and float((y > 0).float().mean()) < 0.8


def test_identity_block_backward_with_fused_residual_is_bit_identical():
    """engine.Config.fuse_block_residual: the block gradient dout * mask is added by conv1's data-gradient epilogue
    instead of being written by BatchNorm backward and read back - same arithmetic, one tensor pass less.  Both forms
    run on the SAME forward state (two forwards of a small block differ in the last bit of the BatchNorm statistics -
    the K-split tail sums its partial statistics with fp32 atomics - and a ReLU mask then flips here and there)."""
    from ee_semantic_segmentation_amd import engine as E
    from ee_semantic_segmentation_amd.nn_modules import Bottleneck
    cfg = E.Config()
    cfg.compute_dtype = torch.bfloat16
    torch.manual_seed(3)
    blk = Bottleneck(1024, 256, 1, None, 2, cfg=cfg).to(DEV).train()
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 33, 29, 1024, generator=g).to(DEV).bfloat16()
    gy = torch.randn(2, 33, 29, 1024, generator=g).to(DEV).bfloat16()
    _, st = E.bottleneck_fwd(cfg, x, blk, True)
    outs = []
    for fuse in (False, True):
        cfg.fuse_block_residual = fuse
        dx, grads = E.bottleneck_bwd(cfg, st, gy.clone(), blk)
        outs.append([dx.clone()] + [t.clone() for t in grads])
    assert torch.equal(outs[0][0], outs[1][0])            # the input gradient: same kernels, same arithmetic
    for a, b in zip(outs[0][1:], outs[1][1:]):            # weight gradients are summed with fp32 atomics: last-bit noise
        close(a, b, 1e-3, "parameter gradient")


@pytest.mark.parametrize("shape,accumulate", [((2, 33, 31, 512, 256), False), ((1, 65, 65, 1024, 256), True)],
                         ids=["2x33x31-512", "1x65x65-1024-acc"])
def test_aspp_data_gradients_in_one_launch(shape, accumulate):
    """conv_dgrad_multi (eeseg_conv_args.taps: generalised taps on the 256-tile kernel): the data-gradients of a 1x1 and
    three atrous 3x3 convs of ONE input (dilations 12 / 24 / 36: whole taps fall outside the map and are skipped),
    28 taps in one launch, against torch autograd in fp32 on the same bf16 values."""
    N, H, W, Cin, mid = shape
    g = torch.Generator().manual_seed(12)
    geoms = [(1, 0, 1), (3, 12, 12), (3, 24, 24), (3, 36, 36)]
    x = torch.zeros(N, Cin, H, W, requires_grad=True)
    ws = [rnd(torch.bfloat16, mid, Cin, k, k, seed=20 + i, scale=(Cin * k * k) ** -0.5).requires_grad_(False)
          for i, (k, p, d) in enumerate(geoms)]
    dys = [rnd(torch.bfloat16, N, mid, H, W, seed=30 + i) for i in range(4)]
    tot = sum((F.conv2d(x, w, padding=p, dilation=d) * dy).sum() for w, dy, (k, p, d) in zip(ws, dys, geoms))
    tot.backward()
    want = x.grad
    base = rnd(torch.bfloat16, N, Cin, H, W, seed=40) if accumulate else None
    if accumulate:
        want = want + base
    wbs = [K.pack_weight(w.to(DEV), torch.bfloat16)[1] for w in ws]
    wcat = K.concat_tap_weights(wbs)
    assert wcat.shape == (Cin, 28, mid)
    t0 = 0
    for wb in wbs:                                         # the taps of every conv side by side, bit for bit
        tb = wb.shape[1] * wb.shape[2]
        assert torch.equal(wcat[:, t0:t0 + tb], wb.reshape(Cin, tb, mid))
        t0 += tb
    dcc = torch.stack([nhwc(dy).to(DEV, torch.bfloat16) for dy in dys]).contiguous()
    acc = nhwc(base).to(DEV, torch.bfloat16) if accumulate else None
    dx = K.conv_dgrad_multi(dcc, wcat, geoms, accumulate_into=acc)
    close(nchw(dx), want, tol(torch.bfloat16), "merged ASPP data-gradient")
    # and against the four separate launches it replaces (those round to bf16 after every branch)
    sep = None
    for wb, dy, (k, p, d) in zip(wbs, dcc, geoms):
        sep = K.conv_dgrad(dy, wb, (H, W), 1, p, d, accumulate_into=sep)
    if not accumulate:
        close(dx, sep, 2.5e-2, "merged vs separate launches")


def test_conv_256_tile_cout_group_order():
    """EESEG_OPT_CONV_COUT_GROUP: layers with more cout tiles than the group walk their tiles in blocks of
    (group cout tiles) x (32 / group pixel tiles) - only the ORDER of the tiles changes, so every setting gives the same
    bits (full rounds, K-split tail + fix-up, ragged last pixel tiles), and they match torch."""
    from ee_semantic_segmentation_amd._lib import lib
    N, H, W, Cin, Cout = 4, 65, 65, 64, 2048          # 67 pixel tiles x 8 cout tiles = 536 tiles: 2 rounds + a split tail
    x = rnd(torch.bfloat16, N, Cin, H, W, seed=3)
    w = rnd(torch.bfloat16, Cout, Cin, 3, 3, seed=4, scale=(Cin * 9) ** -0.5)
    want = F.conv2d(x.float(), w.float(), padding=2, dilation=2)
    wf, _ = K.pack_weight(w.to(DEV), torch.bfloat16)
    xd = nhwc(x).to(DEV, torch.bfloat16)
    outs = {}
    try:
        for cg in (0, 1, 2, 4, 8):
            assert lib().eeseg_set_option(16, cg) == 0
            outs[cg] = K.conv_fwd(xd, wf, 1, 2, 2)[0].clone()
    finally:
        lib().eeseg_set_option(16, 0)
    close(nchw(outs[4]), want, tol(torch.bfloat16), "cout-grouped tile order vs torch")
    for cg, y in outs.items():
        assert torch.equal(y, outs[0]), cg


SWP_DEFAULT = 1      # default of EESEG_OPT_CONV_SWP (include/eeseg.h)


@pytest.mark.parametrize("case", ["3x3-stats", "1x1-residual-relu", "atrous-tail"])
def test_conv_256_tile_mfma_16x16x32(case):
    """EESEG_OPT_CONV_MFMA16: the 256-tile kernel built on v_mfma_f32_16x16x32_bf16 (four 16x16 accumulators per 32x32
    sub-block, other lane -> (cout, pixel) map in the epilogue, the K-split slabs and the fix-up) against torch and against
    the 32x32x16 build: whole tiles, a K-split tail + fix-up, BN partial sums, residual + ReLU."""
    from ee_semantic_segmentation_amd._lib import lib
    if case == "3x3-stats":
        N, H, W, Cin, Cout, k, pad, dil, res, relu, stats = 5, 65, 65, 128, 256, 3, 2, 2, False, False, True
    elif case == "1x1-residual-relu":
        N, H, W, Cin, Cout, k, pad, dil, res, relu, stats = 17, 65, 65, 1024, 256, 1, 0, 1, True, True, False
    else:
        N, H, W, Cin, Cout, k, pad, dil, res, relu, stats = 2, 33, 33, 256, 512, 3, 12, 12, False, False, True
    x = rnd(torch.bfloat16, N, Cin, H, W, seed=5)
    w = rnd(torch.bfloat16, Cout, Cin, k, k, seed=6, scale=(Cin * k * k) ** -0.5)
    r = rnd(torch.bfloat16, N, Cout, H, W, seed=7) if res else None
    want = F.conv2d(x.float(), w.float(), padding=pad, dilation=dil)
    raw = want.bfloat16().float()
    if res:
        want = want.bfloat16().float() + r.float()
    if relu:
        want = want.clamp_min(0)
    wf, _ = K.pack_weight(w.to(DEV), torch.bfloat16)
    xd = nhwc(x).to(DEV, torch.bfloat16)
    rd = nhwc(r).to(DEV, torch.bfloat16) if res else None
    outs = {}
    try:
        lib().eeseg_set_option(13, 0)                  # keep pointwise layers off the 128x256 kernels for this comparison
        lib().eeseg_set_option(21, 0)                  # ... and small layers off the round-4 small-M form of that kernel
        for m16 in (0, 1, 2):                          # 2 = 16x16x32 with the software-pipelined K loop (EESEG_OPT_CONV_SWP)
            assert lib().eeseg_set_option(17, int(m16 > 0)) == 0 and lib().eeseg_set_option(19, int(m16 == 2)) == 0
            y, part = K.conv_fwd(xd, wf, 1, pad, dil, want_stats=stats, residual=rd, relu=relu)
            assert lib().eeseg_last_kernel(0) == 3                      # the 256-tile kernel ran
            outs[m16] = (y.clone(), None if part is None else K.reduce_partials(part).clone())
    finally:
        lib().eeseg_set_option(17, 1)
        lib().eeseg_set_option(19, SWP_DEFAULT)
        lib().eeseg_set_option(13, 1280)
        lib().eeseg_set_option(21, 2)
    for m16, (y, sums) in outs.items():
        close(nchw(y), want, tol(torch.bfloat16), f"mfma16={m16} vs torch")
        if sums is not None:
            flat = raw.permute(1, 0, 2, 3).reshape(Cout, -1).double()
            close(sums[0].cpu(), flat.sum(1).float(), 2e-3, f"mfma16={m16} sum")
            close(sums[1].cpu(), (flat * flat).sum(1).float(), 2e-3, f"mfma16={m16} sum of squares")
    close(outs[1][0], outs[0][0], 8e-3, "16x16x32 vs 32x32x16")
    assert torch.equal(outs[2][0], outs[1][0]), "software-pipelined loop: same MFMAs in the same order, same bits"

