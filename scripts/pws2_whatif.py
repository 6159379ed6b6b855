"""diagnostic (EESEG_EXTRA_FLAGS="-DEESEG_W2_WHATIF"): conv_pws2_kernel with one piece of a round left out at a time"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
B = 32
x = torch.randn(B, 65, 65, 256, device="cuda").bfloat16()
wf, _ = K.pack_weight(torch.randn(1024, 256, 1, 1, device="cuda") / 16, torch.bfloat16)
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib().eeseg_set_option(14, MODE)


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


names = {0: "everything", 1: "with setprio", 2: "no statistics", 4: "no global stores", 8: "no pack / staging writes", 16: "no MFMAs", 32: "no LDS-DMA",
         64: "output waves idle", 2 | 4: "no statistics, no stores", 8 | 16: "no MFMAs, no staging", 8 | 16 | 32: "MFMA waves: fragment reads only",
         8 | 16 | 32 | 64: "barriers + fragment reads", 2 | 8 | 16: "no stats, no MFMA, no staging"}
for var, name in names.items():
    os.environ["EESEG_W2_VAR"] = str(var)
    t = timed(lambda: K.conv_fwd(x, wf, want_stats=True))
    print(f"var {var:3d} {name:36s}: {t:6.1f} us", flush=True)
