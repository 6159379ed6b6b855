"""The expanding Cin = 256 pointwise layers on the three forms (option 14: 0 = 128x256 kernel, 1 = weight-stationary, 2 / 3 = weight-
stationary with 4 / 8 MFMA waves and 4 output waves), replayed from a HIP graph.
    python scripts/pws_bench.py [images]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ee_semantic_segmentation_amd import kernels as K  # noqa: E402
from ee_semantic_segmentation_amd._lib import lib  # noqa: E402

DEV = "cuda"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(3):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * n) * 1e3


H = W = 65
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, 256, generator=g).to(DEV).bfloat16()
wf, _ = K.pack_weight((torch.randn(1024, 256, 1, 1, generator=g) / 16).to(DEV), torch.bfloat16)
_, wb = K.pack_weight((torch.randn(256, 1024, 1, 1, generator=g) / 32).to(DEV), torch.bfloat16)
t = torch.randn(B, H, W, 1024, generator=g).to(DEV).bfloat16()
_, mask = K.bn_apply(t, torch.stack([torch.ones(1024), torch.zeros(1024)]).to(DEV), relu=True, want_mask=True)
M = B * H * W
mb_fwd = (M * 256 + M * 1024) * 2 / 1e6
mb_dg = (M * 256 + 2 * M * 1024) * 2 / 1e6 + M * 128 / 1e6
for mode in (0, 1, 2, 3, 4, 5):
    lib().eeseg_set_option(14, mode)
    tf = timed(lambda: K.conv_fwd(x, wf, want_stats=True))
    tn = timed(lambda: K.conv_fwd(x, wf))
    td = timed(lambda: K.conv_dgrad(x, wb, (H, W), add=(t, mask)))
    print(f"B {B} option 14 = {mode}: fwd {tn:6.1f} us, fwd + stats {tf:6.1f} us ({mb_fwd / tf * 1e3 / 1e3:.2f} TB/s), dgrad + masked residual {td:6.1f} us "
          f"({mb_dg / td * 1e3 / 1e3:.2f} TB/s)", flush=True)
lib().eeseg_set_option(14, 1)
