"""Pointwise weight gradients of layer 3 at the metric's shape (B x 65 x 65): the 128-tile kernel (cost-model split),
the 256-tile kernel with fp32 atomics, and the 256-tile kernel with K-split slabs + fixed-order reduce.
usage: python scripts/wgrad_pw_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best

for H, Cin, Cout, k in ((65, 1024, 256, 1), (65, 256, 1024, 1), (65, 512, 2048, 1), (65, 2048, 512, 1), (65, 256, 256, 3), (65, 512, 512, 3), (65, 2048, 256, 3)):
    x = (torch.randn(B, H, H, Cin, device=dev) * 0.1).bfloat16()
    dy = (torch.randn(B, H, H, Cout, device=dev) * 0.1).bfloat16()
    pad = dil = 2 if k == 3 else 0
    dil = max(dil, 1)
    out = torch.zeros(Cout, k, k, Cin, device=dev)
    fn = lambda: K.conv_wgrad(x, dy, k, k, 1, pad, dil, out=out, accumulate=True)
    row = {}
    K.WGRAD_SLABS = False
    lib().eeseg_set_wgrad_big(0); row["128-tile"] = timeit(fn)
    lib().eeseg_set_wgrad_big(1); row["model"] = timeit(fn)
    lib().eeseg_set_wgrad_big(2); row["256 atomics"] = timeit(fn)
    lib().eeseg_set_wgrad_big(2 | 16); row["256 atomics 16x16x32"] = timeit(fn)
    K.WGRAD_SLABS = True
    lib().eeseg_set_wgrad_big(2 | 4); row["256 slabs"] = timeit(fn)
    K.WGRAD_SLABS = False
    lib().eeseg_set_wgrad_big(1)
    mb = 2 * (x.numel() + dy.numel()) / 1e6
    print(f"{k}x{k} {Cin:5d}->{Cout:5d}  " + "  ".join(f"{n}: {t:7.1f} us" for n, t in row.items()) + f"   (operands {mb:.0f} MB = {mb / 5.5e3 * 1e3:.0f} us at 5.5 TB/s)", flush=True)
