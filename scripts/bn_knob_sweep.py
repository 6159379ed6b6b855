"""BatchNorm backward (reduce + apply, byte-mask form) on conv3- / conv1-sized tensors: grid cap (eeseg_set_ew_grid_cap), rows in
flight per thread (EESEG_OPT_BN_ROWS), column-reduction blocks (EESEG_OPT_COLREDUCE_BLOCKS).  usage: python scripts/bn_knob_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ee_semantic_segmentation_amd import kernels as K
from ee_semantic_segmentation_amd._lib import lib
dev = torch.device("cuda")
filler = torch.empty(600 << 20, dtype=torch.uint8, device=dev)

def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        filler.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best

for B, C in ((32, 1024), (32, 256), (32, 2048)):
    M = B * 65 * 65
    dy = (torch.randn(M, C, device=dev) * 0.1).bfloat16()
    x = torch.randn(M, C, device=dev).bfloat16()
    res = torch.randn(M, C, device=dev).bfloat16()
    mi = torch.stack([torch.zeros(C, device=dev), torch.ones(C, device=dev)]).contiguous()
    ss = torch.stack([torch.ones(C, device=dev), torch.zeros(C, device=dev)]).contiguous()
    gamma = torch.ones(C, device=dev)
    y, mask = K.bn_apply(x, ss, residual=res, relu=True, want_mask=True)
    dx = torch.empty_like(x)
    sums = K.bn_bwd_reduce(dy, mask, x, mi, True, scale_shift=ss)
    T = M * C * 2 / 1e6
    print(f"B={B} C={C} ({T:.0f} MB per tensor)")
    for cap in (512, 768, 1024, 1536, 2048):
        for rows in (1, 2, 4):
            lib().eeseg_set_ew_grid_cap(cap); lib().eeseg_set_option(10, rows)
            ta = timeit(lambda: K.bn_bwd_apply(dy, mask, x, mi, gamma, sums, M, True, dx=dx, scale_shift=ss))
            tf = timeit(lambda: K.bn_apply(x, ss, residual=res, relu=True, want_mask=True, out=dx))
            print(f"  apply cap {cap:5d} rows {rows}: bwd {ta:7.1f} us {(3 * T + T / 16) / ta:5.2f} TB/s | fwd(+res,+mask) {tf:7.1f} us {(3 * T + T / 16) / tf:5.2f} TB/s", flush=True)
    lib().eeseg_set_ew_grid_cap(512); lib().eeseg_set_option(10, 2)
    for cb in (256, 512, 1024, 2048, 0):
        lib().eeseg_set_option(11, cb)
        tr = timeit(lambda: K.bn_bwd_reduce(dy, mask, x, mi, True, scale_shift=ss))
        print(f"  reduce blocks {cb:5d}: {tr:7.1f} us {(2 * T + T / 16) / tr:5.2f} TB/s", flush=True)
    lib().eeseg_set_option(11, 512)
