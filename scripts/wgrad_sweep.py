"""128-tile weight-gradient kernel: time vs the block count its K split aims at (the fp32 partial tiles it adds with
atomics grow with the block count, not with the layer).  usage: python scripts/wgrad_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib

TARGETS = [128, 256, 384, 512, 768, 1024, 0]        # 0 = the library's cost model
SHAPES = [(65, 65, 1024, 256, 1), (65, 65, 256, 1024, 1), (65, 65, 256, 256, 3), (65, 65, 512, 2048, 1), (65, 65, 512, 512, 3),
          (65, 65, 2048, 256, 3), (129, 129, 64, 64, 3), (129, 129, 64, 256, 1), (65, 65, 128, 512, 1), (65, 65, 128, 128, 3)]

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

lib().eeseg_set_wgrad_big(0)
for B in (4, 16):
    print(f"B={B}  (us per call; columns = target blocks)\n" + "shape".ljust(28) + " | " + " ".join(f"{t:>6d}" for t in TARGETS) + " | default (model picks the kernel)")
    for H, W, Cin, Cout, k in SHAPES:
        x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
        dy = torch.randn(B, H, W, Cout, device="cuda").bfloat16()
        pad = k // 2 if k == 3 else 0
        row = []
        for t in TARGETS:
            lib().eeseg_set_wgrad_target_blocks(t)
            row.append(min(timeit(lambda: K.conv_wgrad(x, dy, k, k, 1, pad, 1)) for _ in range(2)))
        lib().eeseg_set_wgrad_target_blocks(0)
        lib().eeseg_set_wgrad_big(1)
        tb = min(timeit(lambda: K.conv_wgrad(x, dy, k, k, 1, pad, 1)) for _ in range(2))
        lib().eeseg_set_wgrad_big(0)
        print(f"{str((H, W, Cin, Cout, k)):28s} | " + " ".join(f"{t:6.1f}" for t in row) + f" | {tb:6.1f}", flush=True)
lib().eeseg_set_wgrad_big(1)
