"""diagnostic (EESEG_EXTRA_FLAGS="-DEESEG_W2_WHATIF"): conv_pws3_kernel (option 14 = 5) with one piece left out / changed at a time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib
B = 32
x = torch.randn(B, 65, 65, 256, device="cuda").bfloat16()
wf, _ = K.pack_weight(torch.randn(1024, 256, 1, 1, device="cuda") / 16, torch.bfloat16)
lib().eeseg_set_option(14, 5)


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


names = {0: "everything", 2: "no statistics", 4: "stores dropped (out of range)", 128: "every block writes a stream of its own", 32: "no LDS-DMA traffic (zero fill)",
         4 | 32: "no stores, no DMA", 2 | 4 | 32: "no stores, no DMA, no statistics", 128 | 2: "own stream, no statistics"}
for var, name in names.items():
    os.environ["EESEG_W2_VAR"] = str(var)
    t = timed(lambda: K.conv_fwd(x, wf, want_stats=True))
    print(f"var {var:3d} {name:42s}: {t:6.1f} us", flush=True)
