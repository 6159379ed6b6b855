"""Pointwise conv shapes of the metric's workload: 256-tile kernel vs 128x256 kernel vs weight-stationary kernel.
usage: python scripts/pw_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
shapes = [(65, 256, 1024), (65, 1024, 256), (65, 512, 2048), (65, 2048, 512), (65, 2048, 256), (65, 1280, 256), (65, 1024, 2048),
          (65, 512, 1024), (65, 1024, 512), (65, 128, 512), (129, 64, 256)]
print(f"B={B}: us per call (fwd with BN partial sums | data-gradient-like call with a residual);  HBM floor at 5.5 TB/s")
for hw, cin, cout in shapes:
    x = torch.randn(B, hw, hw, cin, device="cuda").bfloat16()
    wf, _ = K.pack_weight(torch.randn(cout, cin, 1, 1, device="cuda") * 0.05, torch.bfloat16)
    r = torch.randn(B, hw, hw, cout, device="cuda").bfloat16()
    M = B * hw * hw
    row = []
    for name, opts in (("big", {13: 0, 14: 0, 15: 0}), ("pw", {13: 1280, 14: 0, 15: 1}), ("ws", {13: 1280, 14: 1, 15: 1})):
        for k, v in opts.items(): lib().eeseg_set_option(k, v)
        t1 = timeit(lambda: K.conv_fwd(x, wf, want_stats=True))
        t2 = timeit(lambda: K.conv_fwd(x, wf, residual=r))
        row.append(f"{name} {t1:6.1f} | {t2:6.1f}")
    for k, v in {13: 1280, 14: 1, 15: 0}.items(): lib().eeseg_set_option(k, v)
    f1 = 2.0 * M * (cin + cout) / 5.5e12 * 1e6
    print(f"{hw}x{hw} {cin:5d}->{cout:5d}: " + "   ".join(row) + f"   floor {f1:5.1f} | {2.0*M*(cin+2*cout)/5.5e12*1e6:5.1f}", flush=True)
