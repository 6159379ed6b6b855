"""Sanitizer-style probe: the caching allocator's pool is pre-filled with NaN, so any kernel that reads memory it was
not given (out-of-range taps that are not zero-filled, uninitialised scratch, partially written outputs) shows up as
NaN / a large error against torch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
from ee_semantic_segmentation_amd import kernels as K

def poison(gb=6):
    ts = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(gb)]      # 1 GiB each
    small = [torch.full((n,), float("nan"), device="cuda") for n in (1 << 10, 1 << 14, 1 << 18, 1 << 20, 1 << 22) for _ in range(16)]
    del ts, small
poison()
def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()
g = torch.Generator().manual_seed(3)
cases = [(4, 21, 19, 256, 256, 3, 1, 1, 1), (4, 21, 19, 256, 256, 3, 1, 12, 12), (4, 21, 19, 1280, 256, 1, 1, 0, 1),
         (2, 13, 13, 2048, 256, 3, 1, 36, 36), (2, 25, 25, 64, 64, 3, 1, 1, 1), (2, 25, 25, 256, 64, 1, 1, 0, 1),
         (2, 25, 25, 128, 128, 3, 2, 1, 1), (2, 25, 25, 256, 512, 1, 2, 0, 1), (4, 21, 19, 256, 32, 1, 1, 0, 1),
         (3, 17, 23, 512, 512, 3, 1, 2, 2), (1, 9, 9, 1024, 256, 3, 1, 4, 4), (2, 65, 65, 256, 1024, 1, 1, 0, 1),
         (3, 33, 31, 1024, 256, 1, 1, 0, 1), (1, 9, 11, 64, 256, 1, 1, 0, 1), (16, 65, 65, 256, 1024, 1, 1, 0, 1)]
for dtype in (torch.float32, torch.bfloat16):
    for (N, H, W, Cin, Cout, k, s, p, d) in cases:
        if dtype == torch.bfloat16 and Cout % 64:
            continue
        x = torch.randn(N, Cin, H, W, generator=g).to(dtype).float().requires_grad_(True)
        w = (torch.randn(Cout, Cin, k, k, generator=g) * (Cin * k * k) ** -0.5).to(dtype).float().requires_grad_(True)
        y = F.conv2d(x, w, stride=s, padding=p, dilation=d)
        gy = torch.randn(y.shape, generator=g).to(dtype).float()
        y.backward(gy)
        res = []
        for rep in range(3):
            poison(2)
            xd = x.detach().permute(0, 2, 3, 1).contiguous().to("cuda", dtype)
            wf, wb = K.pack_weight(w.detach().cuda(), dtype)
            yd, part = K.conv_fwd(xd, wf, s, p, d, want_stats=True)
            gyd = gy.permute(0, 2, 3, 1).contiguous().to("cuda", dtype)
            dx = K.conv_dgrad(gyd, wb, (H, W), s, p, d)
            dw = K.conv_wgrad(xd, gyd, k, k, s, p, d)
            sums = K.reduce_partials(part)
            res.append((rel(yd.permute(0, 3, 1, 2), y), rel(dx.permute(0, 3, 1, 2), x.grad), rel(dw.permute(0, 3, 1, 2), w.grad),
                        rel(sums[0], yd.float().reshape(-1, Cout).sum(0))))
        worst = [max(r[i] for r in res) for i in range(4)]
        flag = "  <-- BAD" if any((v != v) or v > (2e-4 if dtype == torch.float32 else 2e-2) for v in worst) else ""
        print(str(dtype)[6:], (N, H, W, Cin, Cout, k, s, p, d), "fwd %.1e dgrad %.1e wgrad %.1e stats %.1e" % tuple(worst), flag, flush=True)
