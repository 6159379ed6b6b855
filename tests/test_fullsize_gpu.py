"""Full-size checks (BASELINE.json shapes, 513x513, B=16) through size-independent
properties - the CPU oracle cannot run these sizes in seconds:
  * adjoint identities  <conv(x), dy> == <x, dgrad(dy)> == <w, wgrad(x, dy)>  (exact algebra,
    so fwd / dgrad / wgrad kernels are mutually consistent at the real layer shapes);
  * linearity of the conv in its input;
  * BN forward output statistics (zero mean / unit variance per channel before the affine);
  * confusion counters add up to the pixel count; fused CE == CE over the stacked tensor;
  * R101 / 4-exit inference with the entropy gate at a non-square size runs and is finite.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


FULL_SHAPES = [  # (H, W, Cin, Cout, k, stride, pad, dil) at B = 16: the heaviest layers of the bench workload
    (65, 65, 2048, 256, 3, 1, 12, 12),
    (65, 65, 2048, 256, 3, 1, 36, 36),
    (65, 65, 512, 512, 3, 1, 4, 4),
    (65, 65, 1024, 2048, 1, 1, 0, 1),
    (129, 129, 128, 128, 3, 2, 1, 1),
    (129, 129, 64, 256, 1, 1, 0, 1),
]


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 4e-3)], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", FULL_SHAPES, ids=[str(s) for s in FULL_SHAPES])
def test_conv_adjoint_identities_full_size(shape, dtype, tol):
    from ee_semantic_segmentation_amd import kernels as K
    H, W, Cin, Cout, k, s, p, d = shape
    B = 16
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn(B, H, W, Cin, device=DEV, generator=g).to(dtype)
    w = (torch.randn(Cout, Cin, k, k, device=DEV, generator=g) * (Cin * k * k) ** -0.5)
    wf, wb = K.pack_weight(w, dtype)
    y, _ = K.conv_fwd(x, wf, s, p, d)
    dy = torch.randn(y.shape, device=DEV, generator=g).to(dtype)
    dx = K.conv_dgrad(dy, wb, (H, W), s, p, d)
    dw = K.conv_wgrad(x, dy, k, k, s, p, d)
    a = _dot(y, dy)                       # <conv(x), dy>
    b = _dot(x, dx)                       # <x, conv^T(dy)>
    c = _dot(wf, dw)                      # <w, wgrad>  (both KRSC)
    scale = (y.double().norm() * dy.double().norm()).item()
    assert abs(a - b) <= tol * scale, (a, b, scale)
    assert abs(a - c) <= tol * scale, (a, c, scale)
    if dtype == torch.float32:            # linearity in the input
        x2 = torch.randn(x.shape, device=DEV, generator=g)
        y2, _ = K.conv_fwd(x2, wf, s, p, d)
        y12, _ = K.conv_fwd(0.5 * x - 2.0 * x2, wf, s, p, d)
        err = (y12 - (0.5 * y - 2.0 * y2)).abs().max().item()
        assert err <= 1e-4 * y.abs().max().item()


def test_full_size_step_properties():
    from ee_semantic_segmentation_amd import kernels as K
    from ee_semantic_segmentation_amd.compute_mIoU import confusion_counts
    from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3
    from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss
    C, B, img = 21, 4, 513
    torch.manual_seed(0)
    net = branchyDeepv3(None, "deeplabv3_resnet50", 1, img, count_branches=False, num_classes=C).to(DEV).train()
    g = torch.Generator().manual_seed(5)
    X = torch.randn(B, 3, img, img, generator=g).to(DEV)
    y = torch.randint(0, C + 1, (B, 1, img, img), generator=g).to(DEV)
    # BN: normalised conv output has zero mean / unit variance per channel
    col = K.im2col_nchw(X, 7, 7, 2, 3, 192, torch.float32)
    from ee_semantic_segmentation_amd import engine as E
    stem_conv, stem_bn = net.base_model[0][0], net.base_model[0][1]
    yb, st = E.conv_bn_fwd(net.cfg, col, stem_conv, stem_bn, relu=False, x_is_col=True)
    flat = yb.reshape(-1, 64).double()
    assert flat.mean(0).abs().max().item() < 1e-4 and (flat.var(0, unbiased=False) - 1).abs().max().item() < 1e-3
    # fused loss == loss over the materialised [E,B,C,H,W] stack, counters add up
    net.fused_outputs = True
    with torch.no_grad():
        el = net(X)
    crit = BrXEntropyLoss(ignore_index=C, b_reduction="sum", n_exits=2)
    l_fused = crit(el, y).item()
    stack = el.stack()
    assert stack.shape == (2, B, C, img, img) and torch.isfinite(stack).all()
    l_stack = crit(stack, y).item()
    assert abs(l_fused - l_stack) < 1e-5 * max(1.0, abs(l_stack))
    for e in range(2):
        cnt = confusion_counts(el, y, e).cpu()
        tp, fp, fn = cnt[0].sum().item(), cnt[1].sum().item(), cnt[2].sum().item()
        assert tp + fp == B * img * img                         # every pixel predicts exactly one class
        assert tp + fn == int((y < C).sum().item())             # every labelled pixel is TP or FN
        # and they agree with the argmax of the stacked tensor (the fused kernel and the materialising kernel
        # interpolate with differently ordered fp32 FMAs: a near-tie may flip in a handful of the 1M pixels)
        pred = stack[e].argmax(1)
        assert abs(int((pred == y.squeeze(1)).sum().item()) - tp) <= 4


def test_r101_four_exit_inference_gate():
    from ee_semantic_segmentation_amd.eval_br_ent import img_norm_entropy
    from ee_semantic_segmentation_amd.from_deepv3_new import ExitLogits, branchyDeepv3
    C = 19
    torch.manual_seed(1)
    net = branchyDeepv3(None, "deeplabv3_resnet101", 3, 256, count_branches=False, num_classes=C,
                        split_after=["layer3.7", "layer3.15", "layer4.0"], compute_dtype=torch.bfloat16).to(DEV).eval()
    assert net.split_names == ["layer3.7", "layer3.15", "layer4.0"] and net.n_branches == 3
    X = torch.randn(1, 3, 257, 385, device=DEV)
    with torch.no_grad():
        el = ExitLogits(net.forward_lowres(X), C, (257, 385))
    assert len(el) == 4 and all(lr.shape == (1, 33, 49, 32) and torch.isfinite(lr).all() for lr in el.lowres)
    gate = img_norm_entropy(C)
    ents = [gate(el, i).item() for i in range(3)]
    assert all(0.0 <= v <= 1.0 + 1e-5 for v in ents), ents
