"""256-tile conv kernel, tile order (EESEG_OPT_CONV_COUT_GROUP): time of the layers with 8 cout tiles at the metric's
shape (B=32, 65x65) per setting.  usage: python scripts/cg_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")

def timeit(fn, iters=5):
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

def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.05).bfloat16()

H = W = 65
cases = []
for cin in (2048, 1024):
    geoms = [(1, 0, 1), (3, 12, 12), (3, 24, 24), (3, 36, 36)]
    dcc = rnd(4, B, H, W, 256)
    wcat = rnd(cin, 28, 256)
    acc = rnd(B, H, W, cin)
    cases.append((f"dgrad-multi 256x4->{cin} T28", lambda dcc=dcc, wcat=wcat, acc=acc: K.conv_dgrad_multi(dcc, wcat, geoms, accumulate_into=acc)))
x512 = rnd(B, H, W, 512); w2048 = rnd(2048, 1, 1, 512)
lib().eeseg_set_option(13, 0)         # keep the pointwise layers on the 256-tile kernel for this comparison
cases.append(("1x1 512->2048 (256-tile kernel)", lambda: K.conv_fwd(x512, w2048, 1, 0, 1)))
x2048 = rnd(B, H, W, 2048); w3 = rnd(2048, 3, 3, 256)
for name, fn in cases:
    row = []
    for cg in (0, 1, 2, 4, 8):
        lib().eeseg_set_option(16, cg)
        row.append(timeit(fn))
    print(f"{name:36s} " + "  ".join(f"cg={c}: {t:8.1f} us" for c, t in zip((0, 1, 2, 4, 8), row)), flush=True)
lib().eeseg_set_option(16, 0)
