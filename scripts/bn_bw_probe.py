"""Effective HBM bandwidth of the BatchNorm kernels on a conv3-sized tensor (B x 65 x 65 x C bf16) next to plain streaming
references (torch copy = 1 read + 1 write, torch sum = read only).  usage: python scripts/bn_bw_probe.py [B] [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda")
M = B * 65 * 65
T = M * C * 2 / 1e6                      # MB of one bf16 tensor
dy = (torch.randn(M, C, device=dev) * 0.1).bfloat16()
x = torch.randn(M, C, device=dev).bfloat16()
res = torch.randn(M, C, device=dev).bfloat16()
mi = torch.stack([torch.zeros(C, device=dev), torch.ones(C, device=dev)]).contiguous()
ss = torch.stack([torch.ones(C, device=dev), torch.zeros(C, device=dev)]).contiguous()
gamma = torch.ones(C, device=dev)
y, mask = K.bn_apply(x, ss, residual=res, relu=True, want_mask=True)
dx = torch.empty_like(x); out = torch.empty_like(x)
filler = torch.empty(600 << 20, dtype=torch.uint8, device=dev)

def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        filler.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best

sums = K.bn_bwd_reduce(dy, mask, x, mi, True, scale_shift=ss)
rows = [
    ("torch copy (1R + 1W)", lambda: out.copy_(x), 2 * T),
    ("torch sum (1R)", lambda: x.view(torch.int16).sum(), T),
    ("bn_apply + residual + mask (2R + 1W + mask)", lambda: K.bn_apply(x, ss, residual=res, relu=True, want_mask=True, out=out), 3 * T + T / 16),
    ("bn_apply relu (1R + 1W)", lambda: K.bn_apply(x, ss, relu=True, out=out), 2 * T),
    ("bn_bwd_reduce, byte mask (2R + mask)", lambda: K.bn_bwd_reduce(dy, mask, x, mi, True, scale_shift=ss), 2 * T + T / 16),
    ("bn_bwd_apply, byte mask (2R + mask + 1W)", lambda: K.bn_bwd_apply(dy, mask, x, mi, gamma, sums, M, True, dx=dx, scale_shift=ss), 3 * T + T / 16),
    ("bn_bwd_reduce, recomputed mask (2R)", lambda: K.bn_bwd_reduce(dy, None, x, mi, True, scale_shift=ss), 2 * T),
    ("bn_bwd_apply, recomputed mask (2R + 1W)", lambda: K.bn_bwd_apply(dy, None, x, mi, gamma, sums, M, True, dx=dx, scale_shift=ss), 3 * T),
]
print(f"B={B} C={C}: one tensor = {T:.0f} MB")
for name, fn, mb in rows:
    t = timeit(fn)
    print(f"  {name:48s} {t:7.1f} us  {mb / t * 1e-6 * 1e6 / 1e6:5.2f} TB/s" if False else f"  {name:48s} {t:7.1f} us  {mb / t:5.2f} TB/s")
