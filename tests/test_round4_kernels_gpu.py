"""GPU parity tests of the round-4 kernels (small-M / per-GPU-shard forms), through the C ABI:

* eeseg_bn_bwd_coop - BatchNorm backward in ONE launch (reduce -> grid barrier -> apply, rows kept in registers) against
  torch CPU autograd (F.batch_norm backward, the op the reference reaches through torchvision's Bottleneck / ASPP) and
  against the two-launch form it replaces;
* the grid barrier's state discipline (left zeroed, no give-up) over many back-to-back launches of uneven grids.

Tolerances as tests/test_kernels_gpu.py: fp32 2-3e-4 of the result scale (summation order), bf16 2-3e-2.
"""
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


# rows, C: the BatchNorm tensors of a 4-image (and one 8-image) shard at 513 x 513 - every register-cache tier, the
# re-read tail (33 800 x 1024: 34 rows per thread, 14 cached), ragged row blocks, one channel group (C = 64)
SHARD_SHAPES = [(4 * 65 * 65, 256), (4 * 65 * 65, 1024), (4 * 65 * 65, 2048), (4 * 129 * 129, 64), (4 * 129 * 129, 256),
                (8 * 65 * 65, 1024), (130, 64), (64, 512), (4 * 65 * 65, 512)]


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
    odd = torch.empty(1000, 40, dtype=torch.bfloat16, device=DEV)
    assert not K.bn_bwd_coop_ok(big) and not K.bn_bwd_coop_ok(odd)


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
