"""Start-skew sweep for the 256x256 conv kernel on the short-K layers (GPU): does putting half of the CUs half a
tile out of phase relieve the lock-step load / store bursts?  usage: python scripts/skew_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dtype = torch.bfloat16
SHAPES = [(65, 65, 256, 1024, 1), (65, 65, 1024, 256, 1), (65, 65, 512, 2048, 1), (65, 65, 2048, 512, 1),
          (65, 65, 1024, 2048, 1), (65, 65, 512, 1024, 1), (65, 65, 1024, 512, 1), (65, 65, 2048, 256, 1),
          (65, 65, 1280, 256, 1), (65, 65, 256, 256, 3)]
SKEWS = [0, 150, 300, 450, 600, 900, 1200]

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

lib().eeseg_set_option(9, 64)
print("shape".ljust(28) + " | " + " ".join(f"{s:>6d}" for s in SKEWS) + "   (us; fwd then dgrad; skew in 10-ns ticks)")
for H, W, Cin, Cout, k in SHAPES:
    x = torch.randn(B, H, W, Cin, device="cuda").to(dtype)
    wt = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    wf, wb = K.pack_weight(wt, dtype)
    pad = k // 2
    y, _ = K.conv_fwd(x, wf, 1, pad, 1, want_stats=True)
    gy = torch.randn_like(y)
    rows = {"fwd": [], "dgrad": []}
    for sk in SKEWS:
        lib().eeseg_set_option(8, sk)
        rows["fwd"].append(min(timeit(lambda: K.conv_fwd(x, wf, 1, pad, 1, want_stats=True)) for _ in range(3)))
        rows["dgrad"].append(min(timeit(lambda: K.conv_dgrad(gy, wb, (H, W), 1, pad, 1)) for _ in range(3)))
    lib().eeseg_set_option(8, 0)
    for kind in ("fwd", "dgrad"):
        print(f"{str((H, W, Cin, Cout, k)):22s}{kind:>6s} | " + " ".join(f"{t:6.1f}" for t in rows[kind]), flush=True)
