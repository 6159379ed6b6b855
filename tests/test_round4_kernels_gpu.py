"""GPU parity tests of the round-4 kernels (small-M / per-GPU-shard forms), through the C ABI:

* eeseg_bn_bwd_coop - BatchNorm backward in ONE launch (reduce -> grid barrier -> apply, rows kept in registers) against
  torch CPU autograd (F.batch_norm backward, the op the reference reaches through torchvision's Bottleneck / ASPP) and
  against the two-launch form it replaces;
* the grid barrier's state discipline (left zeroed, no give-up) over many back-to-back launches of uneven grids.

Tolerances as tests/test_kernels_gpu.py: fp32 2-3e-4 of the result scale (summation order), bf16 2-3e-2.
"""
import os
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from ee_semantic_segmentation_amd import kernels as K

DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]


def rnd(dtype, *shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).float()


def close(got, want, rel, what=""):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = want.abs().max().item() + 1e-12
    err = (got - want).abs().max().item()
    assert err <= rel * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (rel {err / scale:.3e} > {rel})"


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("relu,use_res", [(True, False), (True, True), (False, False)])
def test_bn_bwd_one_launch_vs_torch(dtype, relu, use_res):
    N, H, W, Cc = 3, 23, 19, 128
    x = rnd(dtype, N, Cc, H, W, seed=9, scale=2.0).add_(0.3).to(dtype).float().requires_grad_(True)
    res = rnd(dtype, N, Cc, H, W, seed=10).requires_grad_(True)
    gamma = (torch.rand(Cc) + 0.5).requires_grad_(True)
    beta = torch.randn(Cc).requires_grad_(True)
    y = F.batch_norm(x, torch.zeros(Cc), torch.ones(Cc), gamma, beta, training=True, momentum=0.1, eps=1e-5)
    if use_res:
        y = y + res
    if relu:
        y = F.relu(y)
    gy = rnd(dtype, *y.shape, seed=11)
    y.backward(gy)

    xd = nhwc(x.detach()).to(DEV, dtype)
    cnt = N * H * W
    ga = gamma.detach().to(DEV)
    mi, ss = K.bn_finalize(K.channel_stats(xd), cnt, ga, beta.detach().to(DEV), 1e-5, 0.1,
                           torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV))
    resd = nhwc(res.detach()).to(DEV, dtype) if use_res else None
    yd = K.bn_apply(xd, ss, residual=resd, relu=relu)
    gyd = nhwc(gy).to(DEV, dtype)
    assert K.bn_bwd_coop_ok(xd)
    gyd_before = gyd.clone()
    dx, dres, sums = K.bn_bwd_coop(gyd, yd if relu else None, xd, mi, ga, cnt, relu, want_dres=use_res)
    assert torch.equal(gyd, gyd_before)                       # inputs untouched
    t32, t16 = dtype == torch.float32, dtype == torch.bfloat16
    close(sums[0], beta.grad, 2e-4 if t32 else 2e-2, "dbeta")
    close(sums[1], gamma.grad, 2e-4 if t32 else 2e-2, "dgamma")
    close(nchw(dx), x.grad, 3e-4 if t32 else 3e-2, "dx")
    if use_res:
        close(nchw(dres), res.grad, 1e-4 if t32 else 1.6e-2, "dres")
        _, mask = K.bn_apply(xd, ss, residual=resd, relu=True, want_mask=True)      # byte mask instead of the stored output
        dx3, dres3, sums3 = K.bn_bwd_coop(gyd, mask, xd, mi, ga, cnt, relu, want_dres=True)
        assert torch.equal(sums3, sums) and torch.equal(dx3, dx) and torch.equal(dres3, dres)
    elif relu:                                                                      # mask recomputed from x*scale+shift
        dx2, _, sums2 = K.bn_bwd_coop(gyd, None, xd, mi, ga, cnt, relu, scale_shift=ss)
        assert torch.equal(sums2, sums) and torch.equal(dx2, dx)
    assert t16 or t32
    assert K.coop_timeouts() == 0


# rows, C: the BatchNorm tensors of a 4-image (and 8-image) shard at 513 x 513 - every register-cache tier, the re-read tail
# (16 900 x 1024 and 33 800 x 512: 17 rows per thread, 12 cached), ragged row blocks, one channel group (C = 64)
SHARD_SHAPES = [(4 * 65 * 65, 256), (4 * 65 * 65, 1024), (4 * 129 * 129, 64), (4 * 129 * 129, 256),
                (8 * 65 * 65, 512), (130, 64), (64, 512), (4 * 65 * 65, 512)]


@pytest.mark.parametrize("rows,Cc", SHARD_SHAPES, ids=[f"{r}x{c}" for r, c in SHARD_SHAPES])
def test_bn_bwd_one_launch_equals_the_two_launch_form(rows, Cc):
    """bf16, every ReLU-mask source, a channel-slice output: the fused launch against eeseg_bn_bwd_reduce + eeseg_bn_bwd_apply
    (same arithmetic per element, another summation order: sums to 1e-4 of their scale, dx within one bf16 rounding)."""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(rows + Cc)
    x = (torch.randn(rows, Cc, generator=g) * 1.5 + 0.2).to(DEV, dt)
    dy = torch.randn(rows, Cc, generator=g).to(DEV, dt)
    res = torch.randn(rows, Cc, generator=g).to(DEV, dt)
    ga = (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    be = torch.randn(Cc, generator=g).to(DEV)
    mi, ss = K.bn_finalize(K.channel_stats(x), rows, ga, be, 1e-5, 0.1, torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV))
    y, mask = K.bn_apply(x, ss, residual=res, relu=True, want_mask=True)
    assert K.bn_bwd_coop_ok(x)
    for relu, ysrc, kw in [(True, mask, {}), (True, None, {"scale_shift": ss}), (False, None, {}), (True, y, {})]:
        s_ref = K.bn_bwd_reduce(dy, ysrc, x, mi, relu, **kw)
        dx_ref, dres_ref = K.bn_bwd_apply(dy, ysrc, x, mi, ga, s_ref, rows, relu, want_dres=True, **kw)
        wide = torch.zeros(rows, Cc + 64, dtype=dt, device=DEV)            # dx written into a channel slice
        dx, dres, s = K.bn_bwd_coop(dy, ysrc, x, mi, ga, rows, relu, want_dres=True, dx=wide[:, 64:], **kw)
        close(s, s_ref, 1e-4, f"sums relu={relu}")
        assert torch.equal(dres, dres_ref)
        diff = (dx.float() - dx_ref.float()).abs()
        bound = 2 ** -7 * dx_ref.float().abs() + 1e-4 * dx_ref.float().abs().max()
        assert (diff <= bound).all(), f"dx differs by more than one bf16 rounding ({diff.max().item()})"
        assert torch.count_nonzero(wide[:, :64]) == 0
        # run-to-run identical (fixed summation order)
        dx_b, _, s_b = K.bn_bwd_coop(dy, ysrc, x, mi, ga, rows, relu, want_dres=False, **kw)
        assert torch.equal(s_b, s) and torch.equal(dx_b, dx)
    assert K.coop_timeouts() == 0
    st = K.coop_state(x.device)
    torch.cuda.synchronize()
    assert torch.count_nonzero(st).item() == 0                 # the barrier leaves its counters zeroed, no give-up


def test_bn_bwd_one_launch_is_not_offered_for_large_or_odd_tensors():
    big = torch.empty(32 * 65 * 65, 1024, dtype=torch.bfloat16, device=DEV)
    mid = torch.empty(32 * 65 * 65, 256, dtype=torch.bfloat16, device=DEV)      # 33 rows per thread: mostly re-read, measured slower
    odd = torch.empty(1000, 40, dtype=torch.bfloat16, device=DEV)
    assert not K.bn_bwd_coop_ok(big) and not K.bn_bwd_coop_ok(mid) and not K.bn_bwd_coop_ok(odd)


def test_grid_barrier_back_to_back_under_uneven_load():
    """200 fused launches of alternating grid sizes with a memory-bound kernel of another shape in between: every result
    equals the first of its shape (a stale read across the barrier would change the sums), no give-up, counters zeroed."""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    cases = []
    for rows, Cc in [(4 * 65 * 65, 256), (4 * 65 * 65, 1024), (2000, 64)]:
        x = (torch.randn(rows, Cc, generator=g) + 0.1).to(DEV, dt)
        dy = torch.randn(rows, Cc, generator=g).to(DEV, dt)
        ga = (torch.rand(Cc, generator=g) + 0.5).to(DEV)
        mi, ss = K.bn_finalize(K.channel_stats(x), rows, ga, torch.zeros(Cc, device=DEV), 1e-5, 0.1,
                               torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV))
        dx0, _, s0 = K.bn_bwd_coop(dy, None, x, mi, ga, rows, True, scale_shift=ss)
        cases.append((x, dy, ga, mi, ss, rows, dx0.clone(), s0.clone()))
    filler = torch.randn(64 << 20, device=DEV)
    for it in range(200):
        x, dy, ga, mi, ss, rows, dx0, s0 = cases[it % 3]
        if it % 2:
            filler.mul_(1.0001)
        dx, _, s = K.bn_bwd_coop(dy, None, x, mi, ga, rows, True, scale_shift=ss)
        assert torch.equal(s, s0) and torch.equal(dx, dx0), f"iteration {it}"
    assert K.coop_timeouts() == 0


# ------------------------------------------------------------------------------------------------------------------
# conv_pw_kernel<BM, TAPS>: the small-M form of the 128x256 kernel (64 / 96 / 128-pixel tiles, 3x3 / dilated tap loop)
SMALL_M_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad, dil, expected pixel tile
    (4, 65, 65, 256, 256, 3, 1, 2, 2, 96),       # layer-3 conv2 of a 4-image shard: 177 tiles of 96 px, ragged last tile
    (4, 65, 65, 256, 256, 3, 1, 1, 1, 96),       # head 3x3
    (4, 65, 65, 512, 256, 1, 1, 0, 1, 96),       # contracting pointwise
    (2, 33, 31, 256, 256, 3, 1, 4, 4, 64),       # dilation 4, 32 tiles of 64 px, taps partly outside
    (1, 20, 23, 128, 512, 3, 1, 1, 1, 64),       # two cout tiles, Cin = 128 (4 K tiles per tap)
    (2, 65, 65, 64, 256, 3, 2, 1, 1, 64),        # stride 2 forward (gather is linear in the tap): 33 x 33 outputs
    (4, 65, 65, 256, 1024, 1, 1, 0, 1, 96),      # not small by the 256-tile count, but 532 blocks of 128 px = two rounds for 1.04: 96-px tiles
    (8, 65, 65, 256, 256, 3, 1, 2, 2, 0),        # 8 images: 133 tiles of 256 > CUs / 2 -> stays on the 256-tile kernel
]


@pytest.fixture(params=[1, 2], ids=["one_wave_group", "two_wave_groups"])
def deep_form(request):
    """EESEG_OPT_CONV_SMALL_M_DEEP: 1 = four waves, 2 = eight waves (two groups, each one k-step of every K tile, tiles <= 96 pixels)"""
    from ee_semantic_segmentation_amd._lib import lib
    prev = lib().eeseg_get_option(23)
    assert lib().eeseg_set_option(23, request.param) == 0
    yield request.param
    lib().eeseg_set_option(23, prev)


@pytest.mark.parametrize("case", SMALL_M_CASES, ids=[str(c) for c in SMALL_M_CASES])
def test_conv_small_m_kernel_vs_torch(case, deep_form):
    """forward + BN partial sums, fused scale / shift / residual / ReLU, slice output, data-gradient (plain, accumulating,
    masked-residual) of the small-M kernel against F.conv2d / its autograd; the dispatch (kernel id, stats rows) as documented;
    switching the path off (EESEG_OPT_CONV_SMALL_M = 0) gives the round-3 kernels and the same numbers to bf16 rounding."""
    from ee_semantic_segmentation_amd._lib import lib
    N, H, W, Cin, Cout, k, s, p, d, bm = case
    dtype = torch.bfloat16
    x = rnd(dtype, N, Cin, H, W, seed=1).requires_grad_(True)
    w = rnd(dtype, Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5).requires_grad_(True)
    y = F.conv2d(x, w, stride=s, padding=p, dilation=d)
    gy = rnd(dtype, *y.shape, seed=3)
    y.backward(gy)
    Ho, Wo = y.shape[2], y.shape[3]
    M = N * Ho * Wo
    xd = nhwc(x.detach()).to(DEV, dtype)
    wf, wb = K.pack_weight(w.detach().to(DEV), dtype)
    yd, part = K.conv_fwd(xd, wf, s, p, d, want_stats=True)
    if bm:
        assert lib().eeseg_last_kernel(0) == 4 and part.shape[0] == (M + bm - 1) // bm, (lib().eeseg_last_kernel(0), part.shape)
    else:
        assert lib().eeseg_last_kernel(0) == 3
    close(nchw(yd), y, 1.6e-2, "fwd")
    ys = yd.float().reshape(-1, Cout)
    sums = K.reduce_partials(part)
    close(sums[0], ys.sum(0), 1e-4, "stats sum")
    close(sums[1], (ys * ys).sum(0), 1e-4, "stats sumsq")
    mi, ss = K.bn_reduce_finalize(part, M, torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV), 1e-5, 0.1, None, None)
    close(mi[0], ys.mean(0), 1e-3, "mean through bn_reduce_finalize")
    sc = torch.rand(Cout, generator=torch.Generator().manual_seed(1)) + 0.5
    sh = torch.randn(Cout, generator=torch.Generator().manual_seed(2))
    res = rnd(dtype, *y.shape, seed=4)
    wide = torch.full((N, Ho, Wo, Cout + 256), 7.0, dtype=dtype, device=DEV)
    K.conv_fwd(xd, wf, s, p, d, scale=sc.to(DEV), shift=sh.to(DEV), residual=nhwc(res).to(DEV, dtype), relu=True,
               out=wide[..., 128:128 + Cout])
    want2 = torch.relu(y.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1) + res)
    close(nchw(wide[..., 128:128 + Cout].contiguous()), want2, 1.6e-2, "fused epilogue into a slice")
    assert bool((wide[..., :128] == 7.0).all()) and bool((wide[..., 128 + Cout:] == 7.0).all())
    lib().eeseg_set_option(21, 0)
    try:
        yd0, part0 = K.conv_fwd(xd, wf, s, p, d, want_stats=True)
        assert not bm or lib().eeseg_last_kernel(0) != 4 or k == 1
    finally:
        lib().eeseg_set_option(21, 2)
    close(yd, yd0, 8e-3, "small-M kernel vs round-3 dispatch")
    close(K.reduce_partials(part0), sums, 2e-3, "stats vs round-3 dispatch")
    if Cin % 256 == 0 and s == 1:
        gyd = nhwc(gy).to(DEV, dtype)
        dx = K.conv_dgrad(gyd, wb, (H, W), s, p, d)
        if bm:
            assert lib().eeseg_last_kernel(0) == 4
        close(nchw(dx), x.grad, 1.6e-2, "dgrad")
        dx2 = K.conv_dgrad(gyd, wb, (H, W), s, p, d, accumulate_into=dx.clone())
        close(nchw(dx2), 2 * x.grad, 3.2e-2, "dgrad accumulate")
        t = rnd(dtype, N, H, W, Cin, seed=7).to(DEV, dtype)
        mask = torch.randint(0, 256, (N * H * W, Cin // 8), dtype=torch.uint8, device=DEV)
        dx3 = K.conv_dgrad(gyd, wb, (H, W), s, p, d, add=(t, mask))
        bits = ((mask.unsqueeze(-1) >> torch.arange(8, device=DEV, dtype=torch.uint8)) & 1).reshape(N, H, W, Cin).float()
        close(dx3, (dx.float() + t.float() * bits), 1.6e-2, "dgrad + masked residual")


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("rows,Cc,tiles", [(4 * 65 * 65, 256, 177), (4 * 65 * 65, 1024, 133), (130, 64, 3), (4 * 129 * 129, 64, 320),
                                           (8 * 65 * 65, 512, 265)])
def test_bn_fwd_one_launch_is_bit_identical_with_reduce_finalize_plus_apply(dtype, rows, Cc, tiles):
    """eeseg_bn_fwd_fused against eeseg_bn_reduce_finalize + eeseg_bn_apply(_relu_mask): same summation order, same arithmetic
    -> equal bits in y, the ReLU byte mask, mean / invstd, scale / shift and the running statistics (plain, +ReLU, +residual+ReLU+mask,
    output into a channel slice)."""
    g = torch.Generator().manual_seed(rows + Cc + tiles)
    x = (torch.randn(rows, Cc, generator=g) * 1.3 + 0.4).to(DEV, dtype)
    res = torch.randn(rows, Cc, generator=g).to(DEV, dtype)
    part = (torch.randn(tiles, 2, Cc, generator=g) * 50).to(DEV).contiguous()       # any partial sums do: the order is what is tested
    part[:, 1] = part[:, 1].abs() * 4 + 500.0
    ga = (torch.rand(Cc, generator=g) + 0.5).to(DEV)
    be = torch.randn(Cc, generator=g).to(DEV)
    from ee_semantic_segmentation_amd._lib import lib
    assert lib().eeseg_bn_fwd_fused_ok(rows, Cc, tiles, 0 if dtype == torch.float32 else 1)
    for relu, use_res, want_mask in [(False, False, False), (True, False, False), (True, True, True)]:
        rm1, rv1 = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
        rm2, rv2 = rm1.clone(), rv1.clone()
        mi1, ss1 = K.bn_reduce_finalize(part, rows, ga, be, 1e-5, 0.1, rm1, rv1)
        r = res if use_res else None
        if want_mask:
            y1, m1 = K.bn_apply(x, ss1, residual=r, relu=True, want_mask=True)
        else:
            y1, m1 = K.bn_apply(x, ss1, residual=r, relu=relu), None
        wide = torch.zeros(rows, Cc + 64, dtype=dtype, device=DEV)
        y2, m2, mi2, ss2 = K.bn_fwd_fused(x, part, rows, ga, be, 1e-5, 0.1, rm2, rv2, residual=r, relu=relu, out=wide[:, 64:],
                                          want_mask=want_mask)
        assert torch.equal(mi1, mi2) and torch.equal(ss1, ss2), "coefficients"
        assert torch.equal(rm1, rm2) and torch.equal(rv1, rv2), "running statistics"
        assert torch.equal(y1, y2), "output"
        assert m1 is None or torch.equal(m1, m2), "ReLU byte mask"
        assert torch.count_nonzero(wide[:, :64]) == 0


def test_bn_fwd_one_launch_in_the_layer_function_matches_torch(monkeypatch):
    """engine.conv_bn_fwd with the (opt-in) fused launch on a shard-sized layer: conv -> BN(train) -> ReLU against torch CPU."""
    from ee_semantic_segmentation_amd import engine, nn_modules
    monkeypatch.setattr(K, "FUSED_BN_FWD", True)
    torch.manual_seed(3)
    conv = nn_modules.Conv2d(256, 256, 3, padding=2, dilation=2, bias=False).to(DEV)
    bn = nn_modules.BatchNorm2d(256).to(DEV)
    cfg = engine.Config()
    cfg.compute_dtype = torch.bfloat16
    x = torch.randn(4, 33, 31, 256, device=DEV).to(torch.bfloat16)
    y, st = engine.conv_bn_fwd(cfg, x, conv, bn, True)
    ref = F.relu(F.batch_norm(F.conv2d(x.float().cpu().permute(0, 3, 1, 2), conv.weight.detach().bfloat16().float().cpu(),
                                       padding=2, dilation=2), None, None, bn.weight.detach().cpu(), bn.bias.detach().cpu(),
                              training=True, eps=bn.eps))
    close(y.permute(0, 3, 1, 2), ref, 3e-2, "conv -> BN -> ReLU")


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape", [
    # N, H, W, Cin, Cout, k, pad, dil
    (8, 65, 65, 256, 256, 3, 2, 2),        # 9 tiles x 28 splits = 252 blocks: the layer-3 conv2 of the metric's step (at 8 images)
    (8, 65, 65, 1024, 256, 1, 0, 1),       # 4 tiles x 64 splits
    (4, 65, 65, 512, 512, 3, 4, 4),        # 36 tiles x 7 splits, ragged K ranges
    (6, 40, 70, 256, 512, 1, 0, 1),        # Wout > 64 (the FAST iterator), 2 tiles
], ids=str)
@pytest.mark.parametrize("big", [2, 0], ids=["256-tile", "128-tile"])
def test_wgrad_in_kernel_combine_is_reproducible_and_matches_atomics_and_torch(shape, big):
    """conv_wgrad_big_kernel, epilogue C (eeseg_set_wgrad_big(.. | 8), the default): the K splits of an output tile meet inside
    the launch and sum their slabs in split order - against torch's weight gradient, against the atomics epilogue (same MFMAs,
    other summation order of the splits), bit-identical run to run (the atomics are not), accumulate mode, counters left clean."""
    from ee_semantic_segmentation_amd._lib import lib
    N, H, W, Cin, Cout, k, pad, dil = shape
    dt = torch.bfloat16
    x = rnd(dt, N, Cin, H, W, seed=1).requires_grad_(False)
    w = rnd(dt, Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5).requires_grad_(True)
    y = F.conv2d(x, w, padding=pad, dilation=dil)
    gy = rnd(dt, *y.shape, seed=3)
    y.backward(gy)
    xd, gyd = nhwc(x).to(DEV, dt), nhwc(gy).to(DEV, dt)
    try:
        assert lib().eeseg_set_wgrad_big(big | 8) == 0             # 2 = the 256-tile kernel whatever the cost model says, 0 = never
        dw1 = K.conv_wgrad(xd, gyd, k, k, 1, pad, dil)
        assert lib().eeseg_last_kernel(1) == (7 if big else 6)
        dw2 = K.conv_wgrad(xd, gyd, k, k, 1, pad, dil)
        tile = 256 if big else 128
        eligible = (Cout // tile) * (Cin // tile) * k * k <= 128   # one barrier group per output tile (EESEG_BARRIER_GROUPS)
        if eligible:
            assert torch.equal(dw1, dw2), "in-kernel combine: fixed summation order, identical bits run to run"
            acc = dw1.clone()
            K.conv_wgrad(xd, gyd, k, k, 1, pad, dil, out=acc, accumulate=True)
            assert torch.equal(acc, dw1 + dw1)                     # owner adds into what is there
        assert lib().eeseg_set_wgrad_big(big) == 0                 # atomics
        dwa = K.conv_wgrad(xd, gyd, k, k, 1, pad, dil)
    finally:
        lib().eeseg_set_wgrad_big(1 | 8)
    close(dw1.permute(0, 3, 1, 2), w.grad, 4e-3, "vs torch")
    close(dw1, dwa, 1e-5, "vs the atomics epilogue")
    st = K.coop_state(xd.device, "wgrad")
    torch.cuda.synchronize()
    assert torch.count_nonzero(st).item() == 0


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("classes", ["present", "all", [0, 2, 3, 5]], ids=["present", "all", "list"])
def test_lovasz_class_shares_add_up_to_the_whole_loss(classes):
    """eeseg_lovasz with class_ids / n_label_classes / norm_classes_dev (the class-sharded data-parallel form): ranking classes
    r, r + world, ... separately - every pixel still valid, foreign labels as background - and dividing by the number of classes
    of the WHOLE mean gives shares whose sum is the loss of one call over all classes, and whose gradients are that call's
    gradient planes, bit for bit (same keys, same sort, same Jaccard increments per class)."""
    C_, N, H, W = 7, 2, 37, 41
    g = torch.Generator().manual_seed(11)
    scores = torch.randn(N, C_, H, W, generator=g).to(DEV)
    target = torch.randint(0, C_ + 1, (N, H, W), generator=g)
    target[target == 4] = 1                                      # class 4 absent ('present' drops it, 'all' keeps it)
    target = target.to(DEV)
    ignore = C_
    loss, ds = K.lovasz(scores, target, ignore, want_grad=True, classes=classes)
    cand = list(range(C_)) if isinstance(classes, str) else list(classes)
    counts = K.label_hist(target, C_, ignore)
    assert counts.tolist() == [int((target == c).sum()) for c in range(C_)]
    if classes == "present":
        norm = torch.count_nonzero(counts).to(torch.int32).reshape(1)
    else:
        norm = torch.full((1,), len(cand), dtype=torch.int32, device=DEV)
    world = 3
    total = 0.0
    ds_sh = torch.zeros_like(ds)
    for r in range(world):
        own = cand[r::world]
        sub = scores[:, own].contiguous()
        l_r, d_r = K.lovasz(sub, target, ignore, want_grad=True, classes="present" if classes == "present" else "all",
                            n_label_classes=C_, class_ids=own, norm_classes_dev=norm)
        total += float(l_r.item())
        ds_sh[:, own] = d_r
    assert abs(total - float(loss.item())) <= 2e-6 * abs(float(loss.item())), (total, float(loss.item()))
    assert torch.equal(ds_sh, ds)


# ---- wave-specialised weight-stationary pointwise kernel (conv_pws2_kernel, EESEG_OPT_CONV_PWS = 2) ----

@pytest.mark.parametrize("shape", [(16, 65, 65), (17, 64, 67), (32, 65, 65)], ids=str)
def test_wave_specialised_pointwise_is_bit_identical_to_the_single_role_kernel(shape):
    """256 -> 1024 expanding pointwise conv and the 1024 -> 256 data-gradient that adds a masked residual: the kernel with
    MFMA waves and output waves (option 14 = 2: four MFMA waves of 64 couts, 3: eight of 32, 4: eight of 32 and eight output waves; 5: 512 couts per block,
    every wave in every role) issues the same MFMAs in the same order and sums the BN partial statistics
    in the same order as conv_pws_kernel (option 14 = 1) - outputs AND statistics must be equal bit for bit; both against
    torch.  Ragged pixel counts: the last 128-pixel tile is partial, the tile count is not a multiple of the sequences."""
    from ee_semantic_segmentation_amd._lib import lib
    N, H, W = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, H, W, 256, generator=g).to(DEV).bfloat16()
    wt = (torch.randn(1024, 256, 1, 1, generator=g) * 256 ** -0.5).to(DEV)
    wf, _ = K.pack_weight(wt, torch.bfloat16)
    dy = torch.randn(N, H, W, 256, generator=g).to(DEV).bfloat16()
    wt2 = (torch.randn(256, 1024, 1, 1, generator=g) * 0.05).to(DEV)          # forward conv 1024 -> 256: its dgrad expands
    _, wb2 = K.pack_weight(wt2, torch.bfloat16)
    c = torch.randn(N, H, W, 1024, generator=g).to(DEV).bfloat16()
    ss = torch.stack([torch.rand(1024, generator=g) + 0.5, torch.randn(1024, generator=g) * 0.1]).to(DEV)
    yb, mask = K.bn_apply(c, ss, relu=True, want_mask=True)
    t = torch.randn(N, H, W, 1024, generator=g).to(DEV).bfloat16()
    got = {}
    for mode in (1, 2, 3, 4, 5):
        assert lib().eeseg_set_option(14, mode) == 0
        try:
            y, part = K.conv_fwd(x, wf, want_stats=True)
            assert lib().eeseg_last_kernel(0) == 5
            yr, _ = K.conv_fwd(x, wf, relu=True)
            dx = K.conv_dgrad(dy, wb2, (H, W), add=(t, mask))
            dx2 = K.conv_dgrad(dy, wb2, (H, W), accumulate_into=t.clone())
            got[mode] = (y, part.clone(), yr, dx, dx2)
        finally:
            lib().eeseg_set_option(14, 1)
    for mode in (2, 3, 4, 5):
        for a, b, what in zip(got[1], got[mode], ("output", "statistics", "relu output", "dgrad + masked residual", "dgrad accumulate")):
            if mode >= 4 and what == "statistics":       # other rows per thread (mode 5: one row of sums per 64 pixels): another order of the sums
                assert torch.allclose(K.reduce_partials(a), K.reduce_partials(b), rtol=1e-5, atol=1e-3)
            else:
                assert torch.equal(a, b), (mode, what)
    want = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), wt.bfloat16().float()).permute(0, 2, 3, 1)
    assert float((got[2][0].float() - want).abs().max()) < 0.06
    sums = K.reduce_partials(got[2][1])
    ys = got[2][0].float().reshape(-1, 1024)
    assert torch.allclose(sums[0], ys.sum(0), rtol=1e-4, atol=1e-2) and torch.allclose(sums[1], (ys * ys).sum(0), rtol=1e-4, atol=1e-2)
    wantd = torch.nn.functional.conv_transpose2d(dy.float().permute(0, 3, 1, 2), wt2.bfloat16().float()).permute(0, 2, 3, 1)
    assert float((got[2][3].float() - (wantd + t.float() * (yb > 0))).abs().max()) < 0.06


# ---- several weight gradients in one launch (eeseg_conv_wgrad_group) ----

def _wgrad_items(B, H, W, seed=0):
    g = torch.Generator().manual_seed(seed)
    items = []
    for cin, cout, k, d in ((1024, 256, 1, 1), (256, 256, 3, 2), (256, 1024, 1, 1)):
        x = torch.randn(B, H, W, cin, generator=g).to(DEV).bfloat16()
        dy = torch.randn(B, H, W, cout, generator=g).to(DEV).bfloat16()
        items.append((x, dy, k, k, 1, d * (k // 2), d))
    return items


@pytest.mark.parametrize("shape", [(4, 65, 65), (3, 33, 47), (8, 65, 65), (32, 65, 65)], ids=str)
def test_grouped_weight_gradients_match_the_single_calls(shape):
    """The three weight gradients of a bottleneck block (1x1 1024->256, 3x3 dilated 256->256, 1x1 256->1024) in ONE launch
    (eeseg_conv_wgrad_group) against three eeseg_conv_wgrad calls and against torch: same sums, another K split (fp32 rounding
    only); two grouped calls are bitwise equal (in-kernel combine in split order, no atomics); accumulate adds in place."""
    from ee_semantic_segmentation_amd._lib import lib
    B, H, W = shape
    items = _wgrad_items(B, H, W)
    single = [K.conv_wgrad(x, dy, r, s, st, p, d) for x, dy, r, s, st, p, d in items]
    outs = [torch.empty_like(o) for o in single]
    K.conv_wgrad_group([it + (o, False) for it, o in zip(items, outs)])
    assert lib().eeseg_last_kernel(3) == 3, "the three problems did not share a launch"
    assert K.coop_timeouts() == 0
    for o, ref, (x, dy, r, s, st, p, d) in zip(outs, single, items):
        scale = float(ref.abs().max())
        assert float((o - ref).abs().max()) < 2e-5 * scale + 1e-3
        want = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (dy.shape[-1], x.shape[-1], r, s), dy.float().permute(0, 3, 1, 2),
                                           stride=st, padding=p, dilation=d).permute(0, 2, 3, 1)
        assert float((o - want).abs().max()) < 2e-3 * float(want.abs().max())
    outs2 = [torch.empty_like(o) for o in single]
    K.conv_wgrad_group([it + (o, False) for it, o in zip(items, outs2)])
    assert all(torch.equal(a, b) for a, b in zip(outs, outs2)), "grouped weight gradients are not reproducible"
    K.conv_wgrad_group([it + (o, True) for it, o in zip(items, outs2)])
    for a, b in zip(outs, outs2):
        assert torch.allclose(b, 2 * a, rtol=1e-6, atol=1e-6)


def test_weight_gradient_group_falls_back_to_single_calls():
    """Problems that cannot share a launch (a batch too large for the grouping to pay, a layer the 256x256 kernel does not take) are
    issued one by one - bitwise what eeseg_conv_wgrad gives - and eeseg_last_kernel(3) says so."""
    from ee_semantic_segmentation_amd._lib import lib
    items = _wgrad_items(4, 65, 65)                     # 265 K tiles of 64 pixels
    single = [K.conv_wgrad(x, dy, r, s, st, p, d) for x, dy, r, s, st, p, d in items]
    outs = [torch.empty_like(o) for o in single]
    assert lib().eeseg_set_wgrad_group(200) == 0        # ... above the limit set here
    try:
        K.conv_wgrad_group([it + (o, False) for it, o in zip(items, outs)])
    finally:
        lib().eeseg_set_wgrad_group(1 << 20)
    assert lib().eeseg_last_kernel(3) == 0
    assert all(torch.equal(a, b) for a, b in zip(outs, single))
    # the cost model: the 16 + 36 + 16 output tiles of a layer-4 block at 16 images leave the chip part idle when grouped
    g4 = torch.Generator().manual_seed(9)
    big = []
    for cin, cout, k, d in ((2048, 512, 1, 1), (512, 512, 3, 4), (512, 2048, 1, 1)):
        big.append((torch.randn(16, 65, 65, cin, generator=g4).to(DEV).bfloat16(), torch.randn(16, 65, 65, cout, generator=g4).to(DEV).bfloat16(),
                    k, k, 1, d * (k // 2), d))
    bouts = [torch.empty(it[1].shape[-1], it[2], it[3], it[0].shape[-1], device=DEV) for it in big]
    K.conv_wgrad_group([it + (o, False) for it, o in zip(big, bouts)])
    assert lib().eeseg_last_kernel(3) == 0
    del big, bouts
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 33, 33, 64, generator=g).to(DEV).bfloat16()
    dy = torch.randn(2, 33, 33, 64, generator=g).to(DEV).bfloat16()
    small = _wgrad_items(2, 33, 33)[:1] + [(x, dy, 3, 3, 1, 1, 1)]
    single = [K.conv_wgrad(x_, dy_, r, s, st, p, d) for x_, dy_, r, s, st, p, d in small]
    outs = [torch.empty_like(o) for o in single]
    K.conv_wgrad_group([it + (o, False) for it, o in zip(small, outs)])
    assert lib().eeseg_last_kernel(3) == 0
    assert all(torch.allclose(a, b, rtol=1e-5, atol=1e-4) for a, b in zip(outs, single))    # (the 64-channel layer sums with atomics)


def test_engine_queues_a_units_weight_gradients_into_group_launches():
    """engine.Config.queue_wgrad: with the gradient arena the weight gradients of a unit wait for `unit_done` and leave as
    eeseg_conv_wgrad_group calls (bf16, ResNet-50 / 1 exit at 2 x 129 x 129: the 16-image-wide layer-3 / layer-4 blocks share launches);
    the arena after backward equals the arena of the same backward with EESEG_GROUP_WGRAD off to fp32 summation order (another
    K split), and every queue is empty when backward returns."""
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_model_gpu import _inputs, _pair
    from ee_semantic_segmentation_amd._lib import lib
    from ee_semantic_segmentation_amd import engine as E
    from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss
    C, B, img = 21, 2, 129
    net, _ = _pair("deeplabv3_resnet50", 1, img)
    net.cfg.compute_dtype = torch.bfloat16
    net.train()
    net.fused_outputs = True
    net.enable_grad_arena()
    X, y = _inputs(B, C, img, img)
    crit = BrXEntropyLoss(ignore_index=C, b_reduction="sum", n_exits=2)
    calls, grouped = [], []
    orig = E.K.conv_wgrad_group

    def spy(items):
        orig(items)
        calls.append(len(items))
        grouped.append(lib().eeseg_last_kernel(3))

    arenas = []
    E.K.conv_wgrad_group = spy
    try:
        for on in (True, False):
            net.cfg.group_wgrad = on
            net.cfg.arena.flat.zero_()
            torch.manual_seed(3)
            loss = crit(net(X.to(DEV)), y.to(DEV))
            loss.mean().backward()
            net.cfg.run_deferred()
            assert net.cfg._wgrad_queue == []
            arenas.append(net.cfg.arena.flat.clone())
            if on:
                assert calls and max(calls) >= 3 and max(grouped) >= 3, (calls, grouped)
                n_on = len(calls)
            else:
                assert len(calls) == n_on, "the switch still queued weight gradients"
    finally:
        E.K.conv_wgrad_group = orig
    a, b = arenas
    assert torch.isfinite(a).all() and float(b.abs().max()) > 0
    # same forward state is not guaranteed bit for bit between two forwards (a last-bit BatchNorm statistic flips
    # a ReLU mask here and there), so the bar is the one the model parity tests use for repeated bf16 backward passes
    assert float((a - b).norm() / b.norm()) < 2e-2
