"""Diagnostic (GPU): per-parameter gradient error of one train step vs the CPU oracle,
plus extra wgrad unit shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from ee_semantic_segmentation_amd import kernels as K

def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-12)

print("== wgrad unit shapes")
for (N, H, W, Cin, Cout, k, s, p, d) in [(2, 9, 9, 256, 256, 3, 1, 1, 1), (2, 9, 9, 256, 128, 3, 1, 1, 1),
                                         (2, 9, 9, 128, 256, 3, 1, 1, 1), (2, 9, 9, 256, 256, 1, 1, 0, 1),
                                         (2, 9, 9, 512, 256, 3, 1, 36, 36), (2, 13, 13, 256, 256, 3, 1, 2, 2),
                                         (4, 20, 20, 256, 256, 3, 1, 1, 1)]:
    for dtype in (torch.float32, torch.bfloat16):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(N, Cin, H, W, generator=g).to(dtype).float().requires_grad_(True)
        w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(dtype).float().requires_grad_(True)
        y = F.conv2d(x, w, stride=s, padding=p, dilation=d)
        gy = torch.randn(y.shape, generator=g).to(dtype).float()
        y.backward(gy)
        xd = x.detach().permute(0, 2, 3, 1).contiguous().cuda().to(dtype)
        gyd = gy.permute(0, 2, 3, 1).contiguous().cuda().to(dtype)
        dw = K.conv_wgrad(xd, gyd, k, k, s, p, d)
        wg = w.grad.permute(0, 2, 3, 1)
        per_tap = [(r_, s_, round(rel(dw[:, r_, s_], wg[:, r_, s_]) if wg[:, r_, s_].abs().max() > 0 else dw[:, r_, s_].abs().max().item(), 5))
                   for r_ in range(k) for s_ in range(k)]
        print((N, H, W, Cin, Cout, k, s, p, d), str(dtype)[6:], "rel", round(rel(dw, wg), 6), per_tap if rel(dw, wg) > 1e-2 else "")

print("== model step")
from test_model_gpu import _pair, _inputs
from ee_semantic_segmentation_amd.my_pixelwise_xentropy import BrXEntropyLoss
from oracle import losses_ref
for (B, img) in [(2, 65), (4, 97)]:
    C, n = 21, 1
    net, ref = _pair("deeplabv3_resnet50", n, img)
    X, y = _inputs(B, C, img, img)
    ref.train(); out_ref = ref(X)
    out_ref.retain_grad()
    losses_ref.br_xentropy(out_ref, y, ignore_index=C, b_reduction="sum", n_exits=2).mean().backward()
    net.train(); out = net(X.cuda())
    out.retain_grad()
    BrXEntropyLoss(ignore_index=C, b_reduction="sum", n_exits=2)(out, y.cuda()).mean().backward()
    print("B", B, "img", img, "d(stacked logits) rel", rel(out.grad, out_ref.grad), "logits rel", rel(out, out_ref))
    rp = dict(ref.named_parameters())
    rows = sorted(((rel(p.grad, rp[k].grad), k, tuple(p.shape)) for k, p in net.named_parameters()), reverse=True)
    for r, k, s in rows[:12]:
        print(f"{r:.3e} {k} {s}")
    print("...median", rows[len(rows) // 2][0])
    for r, k, s in rows[-12:]:
        print(f"{r:.3e} {k} {s}")
    for k in ["classifier.4.weight", "classifier.4.bias", "classifier.2.weight", "classifier.1.weight",
              "classifier.0.project.0.weight", "classifier.0.convs.4.1.weight", "classifier.0.convs.0.0.weight",
              "base_model.1.1.conv3.weight", "base_model.0.0.weight"]:
        print("  sel", k, f"{rel(dict(net.named_parameters())[k].grad, rp[k].grad):.3e}")
