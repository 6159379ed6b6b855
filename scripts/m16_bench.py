"""256-tile conv kernel: v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 (EESEG_OPT_CONV_MFMA16) on the layer shapes
of the metric's workload, random data.  usage: python scripts/m16_bench.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda")

def timeit(fn, iters=8):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best

def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).bfloat16()

H = 65
for cin, cout, k, d in ((256, 256, 3, 2), (512, 512, 3, 4), (2048, 256, 3, 12), (2048, 256, 3, 36), (1024, 256, 1, 1), (2048, 512, 1, 1), (2048, 256, 1, 1)):
    x = rnd(B, H, H, cin)
    wf, _ = K.pack_weight(torch.randn(cout, cin, k, k, device=dev) * (cin * k * k) ** -0.5, torch.bfloat16)
    pad = d if k == 3 else 0
    fn = lambda: K.conv_fwd(x, wf, 1, pad, d, want_stats=True)
    row = []
    for m16 in (0, 1, 0, 1):
        lib().eeseg_set_option(17, m16); lib().eeseg_set_option(19, 0)
        row.append(timeit(fn))
    lib().eeseg_set_option(17, 1); lib().eeseg_set_option(19, 1)
    swp = min(timeit(fn), timeit(fn))
    lib().eeseg_set_option(19, 1)
    fl = 2.0 * B * H * H * cin * cout * k * k
    print(f"{k}x{k} {cin:5d}->{cout:4d} d{d:<2d}  32x32x16: {min(row[0], row[2]):8.1f} us ({fl / min(row[0], row[2]) / 1e6:6.0f} TF/s)   16x16x32: {min(row[1], row[3]):8.1f} us "
          f"({fl / min(row[1], row[3]) / 1e6:6.0f} TF/s)   ratio {min(row[0], row[2]) / min(row[1], row[3]):.3f}   pipelined: {swp:8.1f} us ({fl / swp / 1e6:6.0f} TF/s)", flush=True)
lib().eeseg_set_option(17, 1)
