"""Small-M conv shapes (per-GPU shards) kernel by kernel, back to back (HIP events over 20 launches each):
    python scripts/conv_sm_bench.py [images]     -> fwd / dgrad / wgrad us for: small-M path on (deep / shallow ring), off"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ee_semantic_segmentation_amd import kernels as K  # noqa: E402
from ee_semantic_segmentation_amd._lib import lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dt = torch.bfloat16


def timed(fn, n=20):
    """us per call of `fn`, replayed from a HIP graph of n calls (eager launches are host-bound below ~17 us per call)"""
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


SHAPES = [(65, 256, 256, 3, 2), (65, 1024, 256, 1, 1), (65, 256, 1024, 1, 1), (65, 512, 512, 3, 4), (65, 2048, 512, 1, 1),
          (65, 512, 2048, 1, 1), (65, 1280, 256, 1, 1), (65, 2048, 256, 1, 1), (129, 64, 256, 1, 1), (129, 256, 64, 1, 1)]
print(f"{B} images; us per call: [small-M deep, two wave groups] [small-M deep] [small-M 3-stage] [round-3 dispatch]   (kernel id of the default)")
for hw, Cin, Cout, k, d in SHAPES:
    p = d * (k // 2)
    x = torch.randn(B, hw, hw, Cin, device="cuda").to(dt)
    w = torch.randn(Cout, Cin, k, k, device="cuda") * 0.05
    wf, wb = K.pack_weight(w, dt)
    y, _ = K.conv_fwd(x, wf, 1, p, d, want_stats=True)
    kid = lib().eeseg_last_kernel(0)
    gy = torch.randn_like(y)
    res = {}
    for name, o21, o23 in [("kw2", 2, 2), ("deep", 2, 1), ("3st", 2, 0), ("r3", 0, 0)]:
        lib().eeseg_set_option(21, o21)
        lib().eeseg_set_option(23, o23)
        tf = timed(lambda: K.conv_fwd(x, wf, 1, p, d, want_stats=True))
        td = timed(lambda: K.conv_dgrad(gy, wb, (hw, hw), 1, p, d)) if Cin % 256 == 0 else float("nan")
        res[name] = (tf, td)
    lib().eeseg_set_option(21, 2)
    lib().eeseg_set_option(23, 1)
    tw = timed(lambda: K.conv_wgrad(x, gy, k, k, 1, p, d))
    fl = 2.0 * y.numel() * Cin * k * k
    ideal = max(fl / 2.5e15, 2.0 * (x.numel() + y.numel() + w.numel()) / 8e12) * 1e6
    print(f"{k}x{k} {Cin:4d}->{Cout:4d} d{d} @{hw}: fwd {res['kw2'][0]:6.1f} {res['deep'][0]:6.1f} {res['3st'][0]:6.1f} {res['r3'][0]:6.1f} | "
          f"dgrad {res['kw2'][1]:6.1f} {res['deep'][1]:6.1f} {res['3st'][1]:6.1f} {res['r3'][1]:6.1f} | wgrad {tw:6.1f} | roofline {ideal:5.1f} us  (kernel {kid})",
          flush=True)
