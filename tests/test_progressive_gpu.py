"""Batched progressive early-exit inference: branchyDeepv3.forward_progressive - per-image exit flags,
compaction of the surviving images and the n_active-limited conv launches, all on the device - against the
reference-shaped operator that evaluates ONE image at a time and always finishes the backbone
(ee_dnn_op_ne.eval_ee_deeplabv3, ee_dnn_op_ne.py:51-108)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _net(arch, n, C, img, dtype, seed=3):
    from ee_semantic_segmentation_amd.from_deepv3_new import branchyDeepv3
    torch.manual_seed(seed)
    net = branchyDeepv3(None, arch, n, img, count_branches=False, num_classes=C, compute_dtype=dtype).to(DEV)
    g = torch.Generator().manual_seed(seed + 1)
    for m in net.modules():                       # non-trivial running statistics: exits then differ from image to image
        if type(m).__name__ == "BatchNorm2d":
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) * 0.5 + 0.75)
    return net.eval()


def _all_exit_entropies(net, X, C):
    """[E-1, B] gate values of every branch for every image (everything computed)."""
    from ee_semantic_segmentation_amd import kernels as K
    H, W = X.shape[-2:]
    with torch.no_grad():
        lrs = net.forward_lowres(X)
    return torch.stack([K.entropy_gate(lr.contiguous(), C, H, W, 0.0)[0] for lr in lrs[:-1]]).cpu().numpy(), lrs


@pytest.mark.parametrize("arch,n,dtype,size,B", [("deeplabv3_resnet50", 2, torch.float32, (97, 97), 6),
                                                 ("deeplabv3_resnet101", 3, torch.bfloat16, (129, 193), 5)],
                         ids=["r50-3exits-f32", "r101-4exits-bf16"])
def test_batched_progressive_equals_one_image_at_a_time(arch, n, dtype, size, B):
    from ee_semantic_segmentation_amd.ee_dnn_op_ne import eval_ee_deeplabv3
    from ee_semantic_segmentation_amd.eval_br_ent import img_norm_entropy
    C = 19
    H, W = size
    net = _net(arch, n, C, H, dtype)
    X = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(5)).to(DEV)
    X *= torch.linspace(0.3, 2.0, B, device=DEV).view(B, 1, 1, 1)          # spread the entropies over the batch
    ents, _ = _all_exit_entropies(net, X, C)
    # the threshold splits the FIRST gate's values (so the batch leaves through different exits) at the split point that is
    # farthest from every gate value of every exit: the decisions then do not hinge on the last bits of an entropy (in bf16
    # the conv launch plan - K-split tails, pointwise kernel choice - depends on the batch size, so batch 1 and batch B
    # round differently)
    first_gate = np.sort(ents[0])
    cands = 0.5 * (first_gate[:-1] + first_gate[1:])
    tau = float(max(cands, key=lambda t: np.abs(ents - t).min()))
    gap = np.abs(ents - tau).min()
    # ---- one image at a time, everything computed (the reference's semantics) -------------------------------------
    op = eval_ee_deeplabv3(net, img_norm_entropy(C), tau, device=torch.device(DEV))
    want_n, want_mask = [], []
    for b in range(B):
        out = op(X[b])
        want_n.append(out["n"])
        want_mask.append(out["exit"])
        first = next((i for i in range(n) if ents[i, b] < tau), n)
        assert out["n"] == first + 1
    assert len(set(want_n)) >= 2, f"the batch should leave through different exits, got {want_n}"
    # ---- the whole batch, progressively ------------------------------------------------------------------------------
    res = net.forward_progressive(X, tau)
    torch.cuda.synchronize()
    got_n = res["exit"].cpu().tolist()
    assert got_n == want_n, (got_n, want_n, ents.tolist(), tau, gap)
    for b in range(B):
        agree = (res["pred"][b].cpu() == want_mask[b]).float().mean().item()
        if dtype == torch.float32:
            assert agree == 1.0, (b, agree)            # per-image results do not depend on the batch around them
        else:
            assert agree > 0.999, (b, agree)           # bf16: the K-split tail of a conv launch depends on the batch size
    # the single-image operator with stop_at_exit=True runs the same machinery with B = 1
    op2 = eval_ee_deeplabv3(net, img_norm_entropy(C), tau, device=torch.device(DEV), stop_at_exit=True)
    for b in (0, B - 1):
        out = op2(X[b])
        assert out["n"] == want_n[b]
        assert (out["exit"] == want_mask[b]).float().mean().item() > (0.999 if dtype == torch.bfloat16 else 0.9999999)
        assert out["exit_flops"] > 0


def test_conv_launches_skip_inactive_slots():
    """eeseg_conv_args.n_active: the leading images come out exactly as in a full launch, the later slots are not
    written at all (every conv kernel family: 256-tile incl. its K-split tail, pointwise, 128-tile, fp32)."""
    from ee_semantic_segmentation_amd import kernels as K
    g = torch.Generator().manual_seed(2)
    cases = [(8, 33, 33, 256, 256, 3, 1, 2, 2, torch.bfloat16),      # 256-tile kernel, one round + split tail
             (8, 33, 33, 256, 1024, 1, 1, 0, 1, torch.bfloat16),     # pointwise kernel
             (8, 33, 33, 1024, 256, 1, 1, 0, 1, torch.bfloat16),     # 256-tile kernel, pointwise form
             (8, 33, 33, 64, 64, 3, 1, 1, 1, torch.bfloat16),        # 128 x 64 tiles
             (8, 17, 17, 64, 128, 3, 2, 1, 1, torch.float32)]        # fp32, strided
    for (N, H, W, Cin, Cout, k, s, p, d, dtype) in cases:
        x = torch.randn(N, H, W, Cin, generator=g).to(DEV, dtype)
        wf, _ = K.pack_weight((torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(DEV), dtype)
        full, _ = K.conv_fwd(x, wf, s, p, d, relu=True)
        for live in (0, 3, 8):
            na = torch.tensor([live], dtype=torch.int32, device=DEV)
            out = torch.full_like(full, 7.0)
            with K.active_images(na):
                K.conv_fwd(x, wf, s, p, d, relu=True, out=out)
            if live == 0:
                pass
            elif dtype == torch.float32 or live == 8:
                assert torch.equal(out[:live], full[:live]), (Cin, Cout, k, live)
            else:       # bf16: the same tiles, but tiles of the K-split tail may be whole tiles now (other rounding)
                assert (out[:live].float() - full[:live].float()).abs().max().item() <= 8e-3 * full.float().abs().max().item()
            hw = out.shape[1] * out.shape[2]
            tile = 256                                                   # a tile may straddle the boundary
            first_untouched = -(-live * hw // tile) * tile // hw + (1 if (-(-live * hw // tile) * tile) % hw else 0)
            if first_untouched < N:
                assert bool((out[first_untouched:] == 7.0).all()), (Cin, Cout, k, live)


def test_exit_select_and_gather():
    from ee_semantic_segmentation_amd import kernels as K
    B = 7
    flags = torch.tensor([0, 1, 0, 0, 1, 1, 0], dtype=torch.int32, device=DEV)
    n_active = torch.tensor([6], dtype=torch.int32, device=DEV)           # slot 6 is not in flight
    order = torch.tensor([4, 0, 6, 2, 5, 1, 3], dtype=torch.int32, device=DEV)
    src = torch.zeros(B, dtype=torch.int32, device=DEV)
    exit_idx = torch.zeros(B, dtype=torch.int32, device=DEV)
    x = torch.arange(B, device=DEV, dtype=torch.float32).view(B, 1, 1, 1).expand(B, 3, 5, 8).contiguous()
    K.exit_select(flags, 2, n_active, order, src, exit_idx)
    y = K.gather_images(x, src, n_active)
    torch.cuda.synchronize()
    assert n_active.item() == 3
    assert order[:3].tolist() == [4, 6, 2] and src[:3].tolist() == [0, 2, 3]
    assert exit_idx.tolist() == [2, 2, 0, 0, 0, 2, 0]                     # images 0, 5, 1 left through exit 2
    assert y[:3, 0, 0, 0].tolist() == [0.0, 2.0, 3.0]
